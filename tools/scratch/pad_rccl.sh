#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in 0 1 2 3; do
LURKHIP_PAD_STREAMS=$v python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$v bench.py --gpus 1 --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('torchrun world 1, pad $v: two in flight', round(d['ms_per_step'],2), 'seq', round(d['config']['sequential']['ms_per_step'],2))"
done
for v in 0 1 2 3; do
LURKHIP_PAD_STREAMS=$v python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$v bench.py --gpus 1 --shards-per-rank 2 --steps 20 --no-cpu-baseline --no-host-pipeline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('torchrun world 1 spr 2 (the N>1 schedule), pad $v:', round(d['ms_per_step'],2))"
done
