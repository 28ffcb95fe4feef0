#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/quick3
python -m pytest tests/test_air_gpu.py tests/test_prover_gpu.py tests/test_workloads_gpu.py tests/test_profile_gpu.py tests/test_open_gpu.py tests/test_schedule_switches_gpu.py -m gpu -x -q 2>&1 | tail -2
A="--no-cpu-baseline --no-two-in-flight --no-host-pipeline"
for i in 1 2; do
python bench.py $A > gpurun_out/quick3/b$i.json 2>/dev/null
LURKHIP_JIT_KEEP_LDS_REGS=1 python bench.py $A > gpurun_out/quick3/old$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/quick3/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],2), d['config']['proofs_identical_across_steps'], {k:round(v,2) for k,v in d['config']['stages_ms'].items()})
PY
