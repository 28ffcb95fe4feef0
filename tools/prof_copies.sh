#!/bin/bash
# GPU box: the blit kernels (__amd_rocclr_copyBuffer / fillBuffer) of one-proof-at-a-time bench steps, with their neighbours in the trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pc -o run -- python $GRAFT_REPO_ROOT/bench.py --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-pipeline > /tmp/pc.log 2>&1
python3 - <<PY
import csv
rows = sorted(csv.DictReader(open('/tmp/pc/run_kernel_trace.csv')), key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
# the last step: from the last k_trace_func-ish kernel cluster backwards is hard; print the copies of the last 45 % of the trace
start = int(n * 0.62)
tot = 0
for i in range(start, n):
    r = rows[i]
    if 'rocclr' in r['Kernel_Name']:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot += d
        prev = rows[i - 1]['Kernel_Name'][:40] if i else ''
        nxt = rows[i + 1]['Kernel_Name'][:40] if i + 1 < n else ''
        print('%8.1f us  %-28s grid %-9s after %-40s before %s' % (d, r['Kernel_Name'][:28], r.get('Grid_Size', r.get('Grid_Size_X', '')), prev, nxt))
print('total us', tot)
PY
