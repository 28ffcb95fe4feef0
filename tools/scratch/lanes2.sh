#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
A="--steps 24 --warmup 3 --no-cpu-baseline --no-two-in-flight --no-host-pipeline --shards-per-rank 2"
summ() { grep phase1 | sed 's/.*prove_lanes=\([0-9.]*\).*/\1/' | tr '\n' ' '; echo; }
echo prio1; python tools/scratch/bench_timing.py $A 2>&1 >/dev/null | summ
echo prio0; LURKHIP_LANE_PRIORITY=0 python tools/scratch/bench_timing.py $A 2>&1 >/dev/null | summ
python bench.py $A | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['rank0_step_ms'])"
python -m pytest tests/test_workloads_gpu.py tests/test_host_pipeline_gpu.py tests/test_prover_gpu.py -m gpu -x -q 2>&1 | tail -2
