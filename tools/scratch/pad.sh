#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in 0 1 2 3 4 5; do
LURKHIP_PAD_STREAMS=$v python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('pad $v: two in flight', round(d['ms_per_step'],2), 'seq', round(d['config']['sequential']['ms_per_step'],2))"
done
