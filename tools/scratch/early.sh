#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
LURKHIP_EARLY_LEAVES=1 python -m pytest tests/test_commit_gpu.py tests/test_prover_gpu.py tests/test_cpu_step_gpu.py -x -q 2>&1 | tail -2
for v in 0 1 0 1; do
LURKHIP_EARLY_LEAVES=$v python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); s=d['config']['sequential']['stages_ms']
print('early leaves $v: two in flight', round(d['ms_per_step'],2), 'seq', round(d['config']['sequential']['ms_per_step'],2), {k: round(s[k],2) for k in ('commit_main','commit_perm','commit_quotient','lde','merkle_leaves')}, d['config']['proofs_identical_across_steps'])"
done
for v in 0 1; do
LURKHIP_EARLY_LEAVES=$v python bench.py --steps 20 --shards-per-rank 2 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('early leaves $v: N>1 schedule', round(d['ms_per_step'],2), d['config']['proofs_identical_across_steps'])"
done
