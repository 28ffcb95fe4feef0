#!/bin/bash
# same-box A/B of an environment switch on the bench step: ab_env.sh VAR [steps]
R=$GRAFT_REPO_ROOT; cd $R
for v in 1 0 1 0; do
  env $1=$v python bench.py --steps ${2:-10} --no-cpu-baseline --no-two-in-flight --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['stages_ms']
print('$1=$v', round(d['ms_per_step'],3), 'lde', round(s['lde'],3), 'leaves', round(s['merkle_leaves'],3), 'commit_main', round(s['commit_main'],2), 'perm', round(s['commit_perm'],2))"
done
