#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python bench.py --steps 300 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('big 300 steps', d['ms_per_step'], d['config']['proofs_identical_across_steps'], d['config']['grand_sum_is_zero'])"
python bench.py --log-rows 12 --steps 600 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('small 600 steps two lanes', d['ms_per_step'], d['config']['proofs_identical_across_steps'], d['config']['grand_sum_is_zero'])"
fails=0
for i in $(seq 1 25); do python -m pytest tests/test_prover_gpu.py -x -q -k "fresh_contexts" 2>&1 | tail -1 | grep -q "5 passed" || fails=$((fails+1)); done
echo "fresh-context runs failed: $fails of 25"
fails=0
for i in $(seq 1 6); do python -m pytest tests/test_schedule_switches_gpu.py tests/test_bytecode_gpu.py -x -q 2>&1 | tail -1 | grep -q "passed" || fails=$((fails+1)); done
echo "switch/bytecode runs failed: $fails of 6"
