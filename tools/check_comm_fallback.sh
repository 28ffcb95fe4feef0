#!/bin/bash
# GPU box: the distributed bench at world 1 (two shards, RCCL behind the C ABI) with the loader working and with LURKHIP_RCCL_LIB pointing
# nowhere: the second run must fall back to torch.distributed on every rank (lurk_amd.comm.bring_up) and still produce a verified set
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/comm_fallback
for tag in ok broken; do
  lib=""; [ $tag = broken ] && lib=/nonexistent/librccl.so
  LURKHIP_RCCL_LIB=$lib python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --shards-per-rank 2 --no-cpu-baseline --no-host-pipeline --steps 8 > gpurun_out/comm_fallback/$tag.json 2> gpurun_out/comm_fallback/$tag.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/comm_fallback/$tag.json").read().strip().splitlines()[-1])
c = d["config"]
print("$tag", round(d["ms_per_step"], 2), "|", c["collectives"][:40], "|", c["rccl_library"], "|", c["c_abi_collectives_fallback"], "|", c["gathered_proof_set"]["grand_sum_of_gathered_proofs_is_zero"])
PY
done
