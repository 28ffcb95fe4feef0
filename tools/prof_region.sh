#!/bin/bash
# GPU box: kernel stats of the bench's TIMED region only (rocprofv3 --selected-regions; bench.py brackets the region with
# roctxProfilerResume / Pause), one proof at a time: the csv divided by the steps is per-proof evidence whose sums reproduce stages_ms.
#   tools/prof_region.sh <tag> [steps] [bench flags ...]      -> gpurun_out/<tag>_region_kernel_stats.csv, gpurun_out/<tag>_region.json
tag=$1; steps=${2:-8}; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d /tmp/pr_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --lanes 1 --steps $steps --warmup 2 --no-cpu-baseline --no-host-pipeline "$@" > /tmp/pr_$tag.log 2>&1
cp /tmp/pr_$tag/run_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/${tag}_region_kernel_stats.csv
grep '^{' /tmp/pr_$tag.log | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_region.json
python3 - <<PY
import csv, json, re
rows = list(csv.DictReader(open('/tmp/pr_$tag/run_kernel_stats.csv')))
steps = $steps
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps
print('  $tag: kernels of the timed region: %.3f ms per proof' % tot)
for r in rows[:28]:
    m = re.search(r'(k_\w+(<[^>]*>)?|jit_\w+|__amd_\w+)', r['Name'])
    print('  %-44s %7.1f calls/proof %8.3f ms/proof' % ((m.group(1) if m else r['Name'])[:44], int(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps))
d = json.loads(open('$GRAFT_REPO_ROOT/gpurun_out/${tag}_region.json').read())
print('  step under rocprof: %.2f ms; stages_ms sum %.2f' % (d['ms_per_step'], sum(v for k, v in d['config']['stages_ms'].items() if k not in ('lde', 'merkle_leaves', 'merkle_levels', 'merkle_top'))))
PY
