#!/bin/bash
# GPU box: BASELINE configs[1] (bench.py --workload poseidon2): the bench line, the kernel statistics of the same command and a PMC
# pass (SQ_INSTS_VALU: the static instruction count bench.py prices its live time with) -> gpurun_out/p2_<tag>/
tag=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/p2_$tag
mkdir -p $out
cd $R
python bench.py --workload poseidon2 > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2_$tag -o p2 -- python bench.py --workload poseidon2 --steps 5 --warmup 1 --no-cpu-baseline > $out/stats.log 2>&1
cp /tmp/p2_$tag/p2_kernel_stats.csv $out/ 2>/dev/null
# one width per pass so that the kernel's instructions divide by that width's permutations
for w in 16 24 32 40; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/p2pmc_${tag}_$w -o p2 -- python - <<PY > /dev/null 2>&1
import torch, lurk_amd
from lurk_amd.poseidon import PoseidonChipset
ctx = lurk_amd.Context(0)
n = 1 << 22
x = torch.randint(0, 2013265921, (n, $w), dtype=torch.int32, device="cuda")
out = torch.empty((n, 8), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
PoseidonChipset(ctx, $w).hash_dev(x, out, n)
ctx.sync()
PY
  cp /tmp/p2pmc_${tag}_$w/p2_counter_collection.csv $out/pmc_w$w.csv 2>/dev/null
done
ls -la $out
