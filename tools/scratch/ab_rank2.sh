#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in "" "--shards-per-rank 2" "--shards-per-rank 2 --rank-pipeline"; do
  python bench.py --steps 10 --no-cpu-baseline --no-host-pipeline $v 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('[$v]', round(d['ms_per_step'],3), d['config']['device_pools'])"
done
