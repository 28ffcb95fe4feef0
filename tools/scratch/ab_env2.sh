#!/bin/bash
# same-box A/B of an environment setting on the DEFAULT bench (two proofs in flight): ab_env2.sh "VAR=val" [steps]
R=$GRAFT_REPO_ROOT; cd $R
for v in A B A B; do
  if [ $v = A ]; then e="$1"; else e="LURKHIP_DUMMY=0"; fi
  env $e python bench.py --steps ${2:-10} --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); s=d['config']['stages_ms']
print('$v', '$e', round(d['ms_per_step'],3), 'seq', round(d['config']['sequential']['ms_per_step'],3) if 'sequential' in d['config'] else None, 'lde', round(s['lde'],3), 'leaves', round(s['merkle_leaves'],3))"
done
