#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
one() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],2), 'seq', round(d['config']['sequential']['ms_per_step'],2) if d['config'].get('sequential') else '')"; }
python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "plain N=1"
python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "plain N=1"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "torchrun N=1"
for v in 0 1 2 3 4 5 6; do
LURKHIP_PAD_STREAMS=$v python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$v bench.py --gpus 1 --shards-per-rank 2 --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "torchrun spr2 pad $v"
done
