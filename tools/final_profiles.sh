#!/bin/bash
# GPU box: everything profiles/rNN_* is made from, at one state of the tree (then: python tools/summarise_profiles.py <tag> rNN gpurun_out/bench_<tag>.json)
R=$GRAFT_REPO_ROOT; cd $R; tag=${1:-fin}
python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
bash tools/prof_bench.sh $tag pmc > /dev/null 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-two-in-flight --no-host-pipeline > gpurun_out/bench_${tag}_torchrun.json 2> gpurun_out/bench_${tag}_torchrun.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --shards-per-rank 2 --no-cpu-baseline --no-two-in-flight --no-host-pipeline > gpurun_out/bench_${tag}_torchrun_2shards.json 2> gpurun_out/bench_${tag}_torchrun_2shards.err
python bench.py --log-rows 12 --no-host-pipeline > gpurun_out/bench_${tag}_log_rows_12.json 2> gpurun_out/bench_${tag}_log_rows_12.err
python bench.py --profile p3-monty-diffusion --no-cpu-baseline --no-host-pipeline > gpurun_out/bench_${tag}_p3_monty_diffusion.json 2> gpurun_out/bench_${tag}_p3_monty_diffusion.err
python tools/soak.py ${SOAK_ROUNDS:-30} > gpurun_out/soak_$tag.txt 2>&1
tail -5 gpurun_out/soak_$tag.txt
python - <<PY
import json
for f in ['bench_$tag','bench_${tag}_torchrun','bench_${tag}_torchrun_2shards','bench_${tag}_log_rows_12','bench_${tag}_p3_monty_diffusion']:
    try:
        d=json.loads(open('gpurun_out/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'), d.get('cpu_baseline'))
    except Exception as e: print(f,'ERR',e)
PY
