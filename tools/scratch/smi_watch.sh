#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
watch_run() {
  ( python bench.py --steps 250 $2 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],2))" ) &
  pid=$!
  sleep 14
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 1; done
  wait $pid
}
watch_run two_lanes ""
watch_run one_lane "--lanes 1"
