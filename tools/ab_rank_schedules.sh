#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # label, extra flags
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 100)) bench.py --gpus 1 --no-cpu-baseline --no-host-pipeline --steps 16 --warmup 3 "${@:2}" 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
b=json.loads(l[-1]); print('$1', round(b['ms_per_step'],2), b['config'].get('rank_proofs_in_flight'), b['config'].get('rank_pipeline'))"
}
for rep in 1 2; do
run seq --no-two-in-flight
run lanes2
run pipe2 --rank-pipeline
run pipe3 --rank-pipeline --rank-pipeline-depth 3
run inflight2 --shards-per-rank 1 --rank-in-flight 2
done
