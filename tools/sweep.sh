#!/bin/bash
# GPU box: the bench at other heights and workloads (regression sweep: every line must say identical True, verified True)
cd $GRAFT_REPO_ROOT
for a in "--log-rows 16" "--log-rows 18" "--workload eval-only" "--workload lurk-mix" "--log-rows 21 --lanes 1 --steps 4" "--log-rows 22 --lanes 1 --steps 3 --warmup 1"; do
  S=$(date +%s)
  python bench.py $a --no-cpu-baseline --no-host-pipeline 2>/tmp/sweep.err | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); c=d['config']; print('[$a]', round(d['ms_per_step'],2), d.get('proof_latency_ms') and round(d['proof_latency_ms'],2), round(d['value']/1e6,2), 'M/s identical', c['proofs_identical_across_steps'], 'verified', c['gathered_proof_set']['product_verifier']['accepted'], 'compile_s', round(c.get('air_compile_s',0),1))
except Exception as e: print('[$a] FAILED', e)
"
  echo "   wall $(( $(date +%s) - S )) s"; tail -2 /tmp/sweep.err | cut -c1-200
done
