#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/wl
A="--no-cpu-baseline --no-two-in-flight --no-host-pipeline"
for w in eval-only lurk-mix fib-mix; do python bench.py $A --workload $w > gpurun_out/wl/$w.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/wl/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],2), round(d['value']/1e6,2), {k:round(v,2) for k,v in d['config']['stages_ms'].items()})
    except Exception as e: print(f,'ERR',e)
PY
