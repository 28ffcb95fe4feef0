import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
g = dist.new_group(backend="gloo")
import lurk_amd
from lurk_amd import poseidon, synth
ctx = lurk_amd.Context(0)
chip = poseidon.PoseidonChipset(ctx, 24)
x = synth.field_elements((1 << 16, 24), seed=1)
def busy():
    chip.hash_batch(x)  # some library work on the ctx stream, host-synchronous
T = {}
def tk(name, t0): T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
roots = np.arange(9, dtype=np.int64).reshape(1, 9)
pin = torch.zeros((1, 9), dtype=torch.int64).pin_memory()
dev = torch.zeros((1, 9), dtype=torch.int64, device="cuda")
out = torch.zeros((1, 1, 9), dtype=torch.int64, device="cuda")
for it in range(30):
    busy()
    t = time.perf_counter(); a = torch.tensor(roots, device="cuda"); tk("tensor_to_cuda", t)
    t = time.perf_counter(); dist.all_gather_into_tensor(out.view(-1), a.view(-1)); tk("all_gather_issue", t)
    t = time.perf_counter(); r = out.cpu(); tk("cpu()", t)
    busy()
    t = time.perf_counter(); c = torch.from_numpy(roots.copy()); o2 = torch.zeros((1, 1, 9), dtype=torch.int64); dist.all_gather_into_tensor(o2.view(-1), c.view(-1), group=g); tk("gloo_all_gather", t)
for k, v in T.items():
    v = v[5:]
    print("%-18s mean %.3f ms  max %.3f ms" % (k, sum(v) / len(v), max(v)))
dist.destroy_process_group()
