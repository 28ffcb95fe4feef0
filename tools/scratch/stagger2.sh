#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for k in 20 60; do for v in 0 22 0 22; do
python bench.py --steps $k --stagger-ms $v --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('steps $k stagger $v ms:', round(d['ms_per_step'],2))"
done; done
