#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in 12 17 23 28 34 23 17 28; do
python bench.py --steps 30 --stagger-ms $v --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('stagger $v ms:', round(d['ms_per_step'],2))"
done
