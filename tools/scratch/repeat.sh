#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
one() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],2), 'seq', round(d['config']['sequential']['ms_per_step'],2) if d['config'].get('sequential') else '')"; }
for i in 1 2 3 4 5 6; do python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "plain N=1 run $i"; done
for i in 1 2 3 4; do LURKHIP_CTX_PRIORITY=-1 python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "prio -1 run $i"; done
