#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in "" "--rank-pipeline" "--rank-pipeline --rank-pipeline-depth 3" "--rank-pipeline --rank-pipeline-depth 3 --rank-pipeline-one-lane" "" "--rank-pipeline --rank-pipeline-depth 3" "--rank-pipeline --rank-pipeline-depth 4 --rank-pipeline-one-lane"; do
  python bench.py --steps 12 --shards-per-rank 2 --no-cpu-baseline --no-host-pipeline $v 2>gpurun_out/ab_rank3.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('[$v]', round(d['ms_per_step'],3), d['config']['proofs_identical_across_steps'], d['config']['grand_sum_is_zero'], d['config']['device_pools']['hipMalloc_calls_in_timed_region'])" || tail -5 gpurun_out/ab_rank3.err
done
