#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/sizes
A="--no-cpu-baseline --no-two-in-flight --no-host-pipeline --steps 4 --warmup 1"
for lr in 12 16 18 21; do timeout 600 python bench.py $A --log-rows $lr > gpurun_out/sizes/fib_$lr.json 2>gpurun_out/sizes/fib_$lr.err; done
timeout 600 python bench.py $A --workload lurk-mix --log-rows 16 > gpurun_out/sizes/lurk_16.json 2>gpurun_out/sizes/lurk_16.err
timeout 900 python bench.py $A --log-rows 18 --shards-per-rank 4 > gpurun_out/sizes/fib_18_4shards.json 2>gpurun_out/sizes/fib_18_4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/sizes/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],2), round(d['value']/1e6,2), d['config']['proofs_identical_across_steps'], d['config']['grand_sum_is_zero'])
    except Exception as e: print(f,'ERR',e); print(open(f[:-5]+'.err').read()[-600:])
PY
