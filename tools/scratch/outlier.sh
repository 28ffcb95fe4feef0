#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/outlier
A="--steps 30 --warmup 3 --no-cpu-baseline --no-two-in-flight --no-host-pipeline"
python bench.py $A --no-spans > gpurun_out/outlier/nospans.json 2>/dev/null
python bench.py $A > gpurun_out/outlier/spans.json 2>/dev/null
python -c "import gc, runpy, sys; gc.disable(); sys.argv=['bench.py']+'$A'.split(); runpy.run_path('bench.py', run_name='__main__')" > gpurun_out/outlier/nogc.json 2>/dev/null
python bench.py --steps 30 --warmup 20 --no-cpu-baseline --no-two-in-flight --no-host-pipeline > gpurun_out/outlier/warm20.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/outlier/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],2), [round(x,1) for x in d['config']['rank0_step_ms']])
    except Exception as e: print(f,'ERR',e)
PY
