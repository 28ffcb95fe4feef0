#!/bin/bash
# GPU box: host-side profile (cProfile) of one-proof-at-a-time bench steps at --log-rows N: per-step cumulative milliseconds of the
# functions called once or more per step
n=${1:-12}; steps=${2:-60}
cd $GRAFT_REPO_ROOT
python -m cProfile -o /tmp/host.prof bench.py --log-rows $n --lanes 1 --steps $steps --warmup 4 --no-cpu-baseline --no-host-pipeline > /tmp/host.log 2>&1
python3 - <<PY
import pstats, json
st = pstats.Stats('/tmp/host.prof')
steps = $steps + 4
rows = []
for (f, l, name), (cc, nc, tt, ct, callers) in st.stats.items():
    if nc >= steps and nc % steps == 0 or nc >= steps * 3:
        rows.append((ct / steps * 1e3, tt / steps * 1e3, nc / steps, '%s:%d %s' % (f.split('/')[-1], l, name)))
rows.sort(reverse=True)
for ct, tt, n, nm in rows[:45]:
    print('%8.3f ms cum %8.3f ms own %7.1f calls/step  %s' % (ct, tt, n, nm))
line = [l for l in open('/tmp/host.log') if l.startswith('{')]
if line:
    d = json.loads(line[-1]); print('step %.3f ms' % d['ms_per_step'], {k: round(v, 2) for k, v in d['config']['stages_ms'].items()})
PY
