#!/bin/bash
# GPU box: rocprofv3 kernel stats (and optionally the PMC passes) of the bench; results under gpurun_out/prof_<tag>/
#   prof_<tag>/bench_kernel_stats.csv        the DEFAULT command (two proofs in flight: kernels of the two lanes overlap)
#   prof_<tag>/seq_kernel_stats.csv, seq_kernel_trace.csv   --lanes 1 (one proof at a time: a kernel alone on the device)
#   pmc_<tag>_{FETCH_SIZE,WRITE_SIZE,sq}/    PMC passes of --lanes 1 --steps 1 (separate passes, --kernel-trace only)
tag=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o bench -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-pipeline > $out/bench.log 2>&1
cp /tmp/prof_$tag/bench_kernel_stats.csv $out/ 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_seq -o seq -- python bench.py --lanes 1 --steps 4 --warmup 1 --no-cpu-baseline --no-host-pipeline > $out/seq.log 2>&1
cp /tmp/prof_${tag}_seq/seq_kernel_stats.csv /tmp/prof_${tag}_seq/seq_kernel_trace.csv $out/ 2>/dev/null
if [ "$2" = "pmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${tag}_$c -o bench -- python bench.py --lanes 1 --steps 1 --warmup 0 --no-cpu-baseline --no-host-pipeline > /dev/null 2>&1
    mkdir -p $R/gpurun_out/pmc_${tag}_$c && cp /tmp/pmc_${tag}_$c/bench_counter_collection.csv $R/gpurun_out/pmc_${tag}_$c/
  done
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pmc_${tag}_sq -o bench -- python bench.py --lanes 1 --steps 1 --warmup 0 --no-cpu-baseline --no-host-pipeline > /dev/null 2>&1
  mkdir -p $R/gpurun_out/pmc_${tag}_sq && cp /tmp/pmc_${tag}_sq/bench_counter_collection.csv $R/gpurun_out/pmc_${tag}_sq/
fi
ls -la $out
