#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in 4 8 2 8 4; do
GPU_MAX_HW_QUEUES=$v python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('GPU_MAX_HW_QUEUES=$v: two in flight', round(d['ms_per_step'],2), 'seq', round(d['config']['sequential']['ms_per_step'],2))"
done
GPU_MAX_HW_QUEUES=8 python bench.py --log-rows 12 --lanes 1 --steps 30 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('small, 8 queues', round(d['ms_per_step'],2))"
python bench.py --log-rows 12 --lanes 1 --steps 30 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('small, default', round(d['ms_per_step'],2))"
