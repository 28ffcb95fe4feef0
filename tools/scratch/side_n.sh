#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in 1 2 3 4 2 3 4; do
LURKHIP_SIDE_LANES=$v python bench.py --log-rows 12 --lanes 1 --steps 40 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); s=d['config']['stages_ms']
print('2^12 rows, side lanes $v:', round(d['ms_per_step'],3), {k: round(s[k],2) for k in ('permutation','quotient_all','open')})"
done
for v in 3 4; do
LURKHIP_SIDE_LANES=$v python bench.py --log-rows 12 --lanes 2 --steps 60 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('2^12 rows two in flight, side lanes $v:', round(d['ms_per_step'],3))"
done
