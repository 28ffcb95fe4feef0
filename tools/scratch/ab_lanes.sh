#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for l in 2 3 4 2 3; do
  python bench.py --lanes $l --steps 24 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('lanes $l', round(d['ms_per_step'],3), 'seq', round(c['sequential']['ms_per_step'],3), 'identical', c['proofs_identical_across_steps'])"
done
