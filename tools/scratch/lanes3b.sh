#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
one() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],2), 'seq', round(d['config']['sequential']['ms_per_step'],2))"; }
for i in 1 2; do
python bench.py --steps 24 --lanes 2 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "lanes 2"
python bench.py --steps 24 --lanes 3 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "lanes 3"
python bench.py --steps 24 --lanes 4 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "lanes 4"
done
LURKHIP_PAD_STREAMS=4 python bench.py --steps 24 --lanes 3 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "lanes 3 pad 4"
LURKHIP_PAD_STREAMS=3 python bench.py --steps 24 --lanes 3 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "lanes 3 pad 3"
