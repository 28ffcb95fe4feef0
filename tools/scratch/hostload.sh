#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
lscpu | grep -E "Model name|^CPU\(s\)|Thread|MHz" | head -5; cat /proc/loadavg; cat /sys/fs/cgroup/cpu.max
run() { python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],2), 'seq', round(d['config']['sequential']['ms_per_step'],2), d['config']['host'])"; }
run idle
for i in $(seq 1 12); do (timeout 60 python -c "
while True: pass" &) ; done
sleep 1
run with_12_hogs
sleep 45
cat /sys/fs/cgroup/cpu.stat | head -8
