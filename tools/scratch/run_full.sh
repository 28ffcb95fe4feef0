#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/full
python -m pytest tests -m gpu -x -q > gpurun_out/full/pytest.txt 2>&1; tail -3 gpurun_out/full/pytest.txt
python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-two-in-flight --no-host-pipeline > gpurun_out/full/bench40.json 2> gpurun_out/full/bench40.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/full/bench40.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value']); print(d['config']['rank0_step_ms']); print(d['config']['stages_ms'])
PY
