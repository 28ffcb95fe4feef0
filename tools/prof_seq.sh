#!/bin/bash
# GPU box: kernel stats of one-proof-at-a-time bench steps; prints the ms per proof of the kernels matching a pattern
#   tools/prof_seq.sh <tag> "<egrep pattern>" [ENV=VAL ...]
tag=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --lanes 1 --steps 4 --warmup 1 --no-cpu-baseline --no-host-pipeline > /tmp/ps_$tag.log 2>&1
python3 - <<PY
import csv, re
tot = 0.0
for r in csv.DictReader(open('/tmp/ps_$tag/run_kernel_stats.csv')):
    if re.search(r'$pat', r['Name']):
        ms = float(r['TotalDurationNs']) / 1e6 / 5
        tot += ms
        m = re.search(r'(k_\w+(<[^>]*>)?|jit_\w+)', r['Name'])
        print('  $tag %-40s %6.1f calls/proof %8.3f ms/proof' % ((m.group(1) if m else r['Name'])[:40], int(r['Calls']) / 5, ms))
print('  $tag total of the pattern: %.3f ms/proof' % tot)
import json
line = [l for l in open('/tmp/ps_$tag.log') if l.startswith('{')]
if line:
    d = json.loads(line[-1]); print('  $tag step under rocprof: %.2f ms; identical proofs: %s; verifier: %s' % (d['ms_per_step'], d['config']['proofs_identical_across_steps'], d['config']['gathered_proof_set']['product_verifier']['accepted']))
PY
