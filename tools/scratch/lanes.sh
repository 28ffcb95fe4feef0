#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/lanes; rm -f gpurun_out/lanes/*
A="--steps 30 --warmup 3 --no-cpu-baseline --no-two-in-flight --no-host-pipeline --shards-per-rank 2"
for p in 0 1 -1; do
LURKHIP_LANE_PRIORITY=$p python bench.py $A > gpurun_out/lanes/prio_$p.json 2>/dev/null
done
LURKHIP_LANE_PRIORITY=1 python bench.py $A > gpurun_out/lanes/prio_1b.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/lanes/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],2), [round(x,1) for x in d['config']['rank0_step_ms']])
    except Exception as e: print(f,'ERR',e)
PY
