#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/lanes3
python -m pytest tests/test_host_pipeline_gpu.py tests/test_workloads_gpu.py tests/test_prover_gpu.py -m gpu -x -q 2>&1 | tail -2
A="--no-cpu-baseline --no-two-in-flight --no-host-pipeline"
python bench.py $A > gpurun_out/lanes3/one.json 2>gpurun_out/lanes3/one.err
python bench.py $A --shards-per-rank 2 > gpurun_out/lanes3/two.json 2>gpurun_out/lanes3/two.err
python bench.py $A --shards-per-rank 2 > gpurun_out/lanes3/two_b.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --shards-per-rank 2 $A > gpurun_out/lanes3/two_torchrun.json 2>gpurun_out/lanes3/tr.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/lanes3/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],2), d['config']['proofs_identical_across_steps'], d['config']['grand_sum_is_zero'], {k:round(v,2) for k,v in d['config']['stages_ms'].items()})
    except Exception as e: print(f,'ERR',e)
PY
tail -3 gpurun_out/lanes3/two.err
