#!/bin/bash
# per-pass FETCH/WRITE of the LDE of one matrix, tiled vs untiled intermediates
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/lde_pmc
for t in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    LURKHIP_NTT_TILED=$t rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/lp_${t}_$c -o x -- python $R/tools/lde_commit_pmc.py $1 $2 2 > /dev/null 2>&1
    cp /tmp/lp_${t}_$c/x_counter_collection.csv $R/gpurun_out/lde_pmc/tiled${t}_$c.csv
  done
done
ls $R/gpurun_out/lde_pmc
