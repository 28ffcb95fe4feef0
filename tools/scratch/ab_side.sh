#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for lr in 12 20; do
for v in 1 4 1 4; do
  LURKHIP_SIDE_LANES=$v python bench.py --log-rows $lr --lanes 1 --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); s=d['config']['stages_ms']
print('log_rows $lr side lanes $v', round(d['ms_per_step'],3), {k: round(s[k],2) for k in ('permutation','quotient_all','open','commit_main')}, d['config']['proofs_identical_across_steps'])"
done; done
