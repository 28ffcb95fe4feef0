#!/bin/bash
# GPU box: the default bench line under a list of environment settings, alternating, two rounds:  tools/ab_env.sh "A=1" "A=2 B=3" ...
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for e in "" "$@"; do
    env $e python bench.py --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stages_ms']
print('%-40s in-flight %.2f ms  latency %.2f ms  lde %.2f leaves %.2f open %.2f quot %.2f perm %.2f' % ('[$e]', d['ms_per_step'], d.get('proof_latency_ms') or 0, s['lde'], s['merkle_leaves'], s['open'], s['quotient_all'], s['permutation']))"
  done
done
