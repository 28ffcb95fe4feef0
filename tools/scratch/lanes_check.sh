#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in 4 1 4 1; do
LURKHIP_SIDE_LANES=$v python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('side lanes env $v: two in flight', round(d['ms_per_step'],2), 'seq', round(d['config']['sequential']['ms_per_step'],2))"
done
