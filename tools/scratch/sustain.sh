#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for k in 20 60 150 400; do
python bench.py --steps $k --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('two lanes, $k steps:', round(d['ms_per_step'],2), 'sequential pass before:', round(d['config']['sequential']['ms_per_step'],2))"
done
python bench.py --lanes 1 --steps 300 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['config']['rank0_step_ms']; print('one lane 300 steps:', round(d['ms_per_step'],2), 'first 10', [round(x,1) for x in r[:10]], 'last 10', [round(x,1) for x in r[-10:]])"
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | head -30
