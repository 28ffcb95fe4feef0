#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
one() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],2), 'seq', round(d['config']['sequential']['ms_per_step'],2) if d['config'].get('sequential') else '')"; }
python bench.py --steps 24 --lanes 2 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "lanes 2"
python bench.py --steps 24 --lanes 3 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "lanes 3"
LURKHIP_SIDE_LANE=0 python bench.py --steps 24 --lanes 2 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "lanes 2, no side lane"
LURKHIP_SIDE_LANE=0 python bench.py --steps 24 --lanes 3 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "lanes 3, no side lane"
LURKHIP_SIDE_LANE=0 python bench.py --steps 24 --lanes 4 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "lanes 4, no side lane"
for v in 1 2 3; do LURKHIP_PAD_STREAMS=$v python bench.py --steps 24 --lanes 3 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "lanes 3 pad $v"; done
