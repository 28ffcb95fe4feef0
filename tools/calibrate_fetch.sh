#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of tools/ubench_fetch.bin's known-byte-count kernels (separate PMC passes, --kernel-trace only)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/calib
$R/tools/ubench_fetch.bin > $R/gpurun_out/calib/known_bytes.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/calib_$c -o ub -- $R/tools/ubench_fetch.bin > /dev/null 2>&1
  cp /tmp/calib_$c/ub_counter_collection.csv $R/gpurun_out/calib/$c.csv
done
ls -la $R/gpurun_out/calib
