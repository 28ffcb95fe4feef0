#!/bin/bash
# GPU box: the round-end sequence the driver runs (gpu tests, smoke, default bench), then the rocprofv3 passes of tools/prof_bench.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q > gpurun_out/final/gpu_tests.log 2>&1; tail -2 gpurun_out/final/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
( time python bench.py ) > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -4 gpurun_out/final/bench.err
bash tools/prof_bench.sh final pmc
