#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)"
one() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],2), 'seq', round(d['config']['sequential']['ms_per_step'],2) if d['config'].get('sequential') else '')"; }
for p in -1 1; do
LURKHIP_CTX_PRIORITY=$p python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "prio $p plain N=1"
LURKHIP_CTX_PRIORITY=$p python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "prio $p torchrun N=1"
LURKHIP_CTX_PRIORITY=$p python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29562 bench.py --gpus 1 --shards-per-rank 2 --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "prio $p torchrun spr2"
for v in 1 2 3; do
LURKHIP_CTX_PRIORITY=$p LURKHIP_PAD_STREAMS=$v python bench.py --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | one "prio $p plain N=1 pad $v"
done
done
