#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
one() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],2), 'seq', round(d['config']['sequential']['ms_per_step'],2) if d['config'].get('sequential') else '')"; }
for v in 0 3 4; do
LURKHIP_PAD_STREAMS=$v LURKHIP_LANE_UNPLACED=1 python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "pad $v unplaced"
LURKHIP_PAD_STREAMS=$v python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "pad $v placed beside"
done
GPU_MAX_HW_QUEUES=2 python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "2 hw queues, placed"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "torchrun N=1 placed"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29572 bench.py --gpus 1 --shards-per-rank 2 --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "torchrun spr2 placed"
