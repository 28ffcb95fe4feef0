#!/bin/bash
# GPU box: quick gate + sequential kernel stats + bench lines (development aid)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_prover_gpu.py tests/test_open_gpu.py tests/test_air_gpu.py tests/test_logup_gpu.py tests/test_commit_gpu.py tests/test_poseidon2_gpu.py -x -q 2>&1 | tail -4
bash tools/prof_seq.sh ab "${1:-jit_|reduce_open|column_dot|k_quotient|k_perm_rows|point_weights|fri_fold|powers|selectors}" | head -40
for m in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d.get('proof_latency_ms'), {k:round(v,2) for k,v in d['config']['stages_ms'].items()})"; done
