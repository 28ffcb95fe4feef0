#!/bin/bash
# The split prover's per-rank work, timed with the device to itself (bench.py --split-turns), for G = 2, 4, 8 ranks on ONE device:
# usage: tools/split_turns.sh <log_rows> <out dir> [G ...]
set -u
LOG=${1:-18}; OUT=${2:-gpurun_out/turns}; shift 2 || true
GS=${@:-2 4 8}
mkdir -p "$OUT"
for G in $GS; do
  free -g | sed -n 2p > "$OUT/mem_before_g$G.txt"
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port $((29600 + G)) \
    bench.py --gpus $G --split intra --oversubscribe --split-turns --log-rows $LOG --steps 3 --warmup 1 \
    > "$OUT/split_g${G}_2p${LOG}.json" 2> "$OUT/split_g${G}_2p${LOG}.err"
  echo "G=$G rc=$?" >> "$OUT/rc.txt"
  rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2 >> "$OUT/rc.txt"
done
