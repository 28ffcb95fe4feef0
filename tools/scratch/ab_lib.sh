#!/bin/bash
# GPU box: A/B of library variants tools/scratch/liblurkhip_<v>.bin against the built one; usage: ab_lib.sh <v> [<v> ...]
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/ab
cp lurk_amd/liblurkhip.so /tmp/lib_new.so
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-two-in-flight --no-host-pipeline"
for rep in 1 2; do
  for v in base "$@"; do
    if [ $v = base ]; then cp /tmp/lib_new.so lurk_amd/liblurkhip.so; else cp tools/scratch/liblurkhip_$v.bin lurk_amd/liblurkhip.so; fi
    $B > gpurun_out/ab/bench_${v}_$rep.json 2> gpurun_out/ab/bench_${v}_$rep.err
  done
done
cp /tmp/lib_new.so lurk_amd/liblurkhip.so
python - <<'PY'
import json,glob
rows={}
for f in sorted(glob.glob('gpurun_out/ab/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'ERR',e); continue
    rows[f.split('/')[-1][6:-5]]=dict(step=d['ms_per_step'],median=sorted(d['config']['rank0_step_ms'])[len(d['config']['rank0_step_ms'])//2],**d['config']['stages_ms'])
keys=list(next(iter(rows.values())).keys())
print('%-18s'%''+' '.join('%11s'%k[-11:] for k in rows))
for k in keys: print('%-18s'%k[:18]+' '.join('%11.2f'%rows[r].get(k,0) for r in rows))
PY
