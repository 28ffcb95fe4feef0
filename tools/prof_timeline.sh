#!/bin/bash
# GPU box: every launch of ONE one-at-a-time proof at --log-rows N (start offset us, duration us, gap before us, grid, workgroup, kernel)
# -> gpurun_out/timeline_<N>.txt
n=${1:-12}; shift
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl -o run -- python $GRAFT_REPO_ROOT/bench.py --log-rows $n --lanes 1 --steps 4 --warmup 2 --no-cpu-baseline --no-host-pipeline > /tmp/ptl.log 2>&1
python3 - <<PY
import csv, re
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')), r.get('Stream_Id', '')) for r in csv.DictReader(open('/tmp/ptl/run_kernel_trace.csv'))))
def short(n):
    m = re.search(r'(k_\w+(<[^>]*>)?|jit_\w+|__amd_rocclr_\w+)', n)
    return m.group(1) if m else n[:40]
opens = [i for i, r in enumerate(rows) if 'k_pow_grind' in r[2]]
ends = []
for i in opens:
    j = i
    while j + 1 < len(rows) and ('k_gather_openings' in rows[j + 1][2] or 'k_records_canonical' in rows[j + 1][2] or 'copyBuffer' in rows[j + 1][2]):
        j += 1
    ends.append(j)
k = len(ends) - 2
seg = rows[ends[k - 1] + 1: ends[k] + 1]
t0 = seg[0][0]
ce = t0
with open('$GRAFT_REPO_ROOT/gpurun_out/timeline_$n.txt', 'w') as f:
    for s, e, nm, g, wg, st in seg:
        f.write('%9.1f %8.1f %7.1f %8s %5s s%s %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, (s - ce) / 1e3, g, wg, st, short(nm)))
        ce = max(ce, e)
print('launches', len(seg), 'span us', (seg[-1][1] - t0) / 1e3)
PY
