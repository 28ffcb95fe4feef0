#!/bin/bash
# GPU box: the rank schedules of the shards -> ranks step at world 1 under torchrun (RCCL behind the C ABI), same build, same box:
#   one shard a rank: one proof at a time / two proofs in flight (the N = 1 line's schedule) / phase 1 of proof j + 1 under phase 2 of j
#   two shards a rank (the N > 1 default): its shards two at a time / + the pipelined phases
cd $GRAFT_REPO_ROOT
run() { # label, extra flags
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 100)) bench.py --gpus 1 --no-cpu-baseline --no-host-pipeline --steps 16 --warmup 3 "${@:2}" 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
b=json.loads(l[-1]); print('$1', round(b['ms_per_step'],2), 'ms per step;', round(b['value']/1e6,2), 'M eval-steps/s; shards per rank', b['config'].get('shards_per_rank'))"
}
for rep in 1 2; do
run "1 shard, one at a time      " --no-two-in-flight
run "1 shard, two in flight       " 
run "1 shard, pipelined depth 3   " --rank-pipeline --rank-pipeline-depth 3
run "2 shards (N > 1 default)     " --shards-per-rank 2
run "2 shards, pipelined depth 2  " --shards-per-rank 2 --rank-pipeline
run "2 shards, pipelined depth 3  " --shards-per-rank 2 --rank-pipeline --rank-pipeline-depth 3
done
