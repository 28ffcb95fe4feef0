#!/bin/bash
# GPU box: where a small proof's time goes -- kernel trace of one-proof-at-a-time steps at --log-rows N (default 12):
# device busy / idle per proof, the gaps over 15 us with the kernels either side, launches and time per kernel name
n=${1:-12}; shift
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/psm -o run -- python $GRAFT_REPO_ROOT/bench.py --log-rows $n --lanes 1 --steps 6 --warmup 2 --no-cpu-baseline --no-host-pipeline > /tmp/psm.log 2>&1
python3 - <<PY
import csv, json, re, collections
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open('/tmp/psm/run_kernel_trace.csv'))))
def short(n):
    m = re.search(r'(k_\w+|jit_\w+|__amd_rocclr_\w+)', n)
    return m.group(1) if m else n[:30]
opens = [i for i, r in enumerate(rows) if 'k_pow_grind' in r[2]]
ends = []
for i in opens:
    j = i
    while j + 1 < len(rows) and 'k_gather_openings' in rows[j + 1][2]:
        j += 1
    ends.append(j)
k = len(ends) - 2
seg = rows[ends[k - 1] + 1: ends[k] + 1]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy, cs, ce, gaps = 0, seg[0][0], seg[0][1], []
prev = seg[0][2]
for s, e, nm in seg[1:]:
    if s > ce:
        busy += ce - cs; gaps.append((s - ce, short(prev), short(nm), (s - t0) / 1e3)); cs, ce = s, e
    else:
        ce = max(ce, e)
    prev = nm
busy += ce - cs
print('proof: span %.3f ms, %d launches, busy %.3f ms, idle %.3f ms in %d gaps (%.3f ms in gaps > 15 us)' % ((t1 - t0) / 1e6, len(seg), busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), sum(g[0] for g in gaps if g[0] > 15000) / 1e6))
for g in sorted(gaps, reverse=True)[:24]:
    print('   gap %6.1f us at %8.1f us  %s -> %s' % (g[0] / 1e3, g[3], g[1], g[2]))
hist = collections.Counter(); [hist.update([min(int(g[0] / 2000), 10)]) for g in gaps]
print('   gap histogram (2 us bins, last = 20+):', [hist[i] for i in range(11)])
by = collections.defaultdict(lambda: [0, 0])
for s, e, nm in seg:
    by[short(nm)][0] += 1; by[short(nm)][1] += e - s
for nm, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:30]:
    print('   %-34s %4d launches %8.1f us' % (nm, c, d / 1e3))
line = [l for l in open('/tmp/psm.log') if l.startswith('{')]
if line:
    d = json.loads(line[-1]); print('step under rocprof %.3f ms' % d['ms_per_step'], {k: round(v, 2) for k, v in d['config']['stages_ms'].items()})
PY
