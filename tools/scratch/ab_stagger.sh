#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for st in 0 10 20 30 0 20; do
  python bench.py --lanes 2 --stagger-ms $st --steps 40 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('stagger $st', round(d['ms_per_step'],3), 'seq', round(c['sequential']['ms_per_step'],3))"
done
