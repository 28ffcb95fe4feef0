#!/bin/bash
# GPU box: A/B of library variants built into tools/scratch/*.bin (one box, interleaved runs)
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/ab
./tools/scratch/ubench_perm.bin > gpurun_out/ab/ubench_new.txt 2>&1
./tools/scratch/ubench_perm_declared.bin > gpurun_out/ab/ubench_declared.txt 2>&1
cp lurk_amd/liblurkhip.so /tmp/lib_new.so
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-two-in-flight --no-host-pipeline"
for rep in 1 2; do
  for v in new declared w8; do
    if [ $v = new ]; then cp /tmp/lib_new.so lurk_amd/liblurkhip.so; else cp tools/scratch/liblurkhip_$v.bin lurk_amd/liblurkhip.so; fi
    $B > gpurun_out/ab/bench_${v}_$rep.json 2> gpurun_out/ab/bench_${v}_$rep.err
  done
done
cp /tmp/lib_new.so lurk_amd/liblurkhip.so
python -m pytest tests/test_poseidon2_gpu.py tests/test_commit_gpu.py tests/test_air_gpu.py tests/test_prover_gpu.py -m gpu -x -q > gpurun_out/ab/pytest.txt 2>&1
tail -3 gpurun_out/ab/pytest.txt
cat gpurun_out/ab/ubench_*.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], {k:round(v,2) for k,v in d['config'].get('spans_ms',{}).items()} if 'spans_ms' in d['config'] else '')
    except Exception as e: print(f,'ERR',e)
PY
