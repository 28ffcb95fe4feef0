"""GPU box only: Poseidon2 hash8 throughput per width with inputs resident in HBM (ctx HIP-event timer)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import lurk_amd
from lurk_amd import synth
from lurk_amd.poseidon import PoseidonChipset

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 22)
with lurk_amd.Context(0) as ctx:
    for width in (16, 24, 32, 40):
        chip = PoseidonChipset(ctx, width)
        x = synth.field_elements((n, width), seed=width)
        xd = torch.from_numpy(x.view(np.int32)).cuda()
        od = torch.empty((n, 8), dtype=torch.int32, device="cuda")
        wd = None
        torch.cuda.synchronize()
        for name, fn in (("hash8", lambda: chip.hash_dev(xd, od, n)),):
            fn(); ctx.sync()
            ctx.timer_start()
            for _ in range(5):
                fn()
            ms = ctx.timer_stop() / 5
            alg = n * (width + 8) * 4
            print(f"W={width:2d} {name}: {ms:8.3f} ms  {n / ms * 1e-6:8.2f} Gperm/s  {alg / ms * 1e-6:8.1f} GB/s algorithmic")
        nw = min(n, 1 << 18)
        wd = torch.empty((nw, chip.witness_size()), dtype=torch.int32, device="cuda")
        chip.witness_dev(xd, wd, nw); ctx.sync()
        ctx.timer_start()
        for _ in range(5):
            chip.witness_dev(xd, wd, nw)
        ms = ctx.timer_stop() / 5
        alg = nw * (width + chip.witness_size()) * 4
        print(f"W={width:2d} wide : {ms:8.3f} ms  {nw / ms * 1e-6:8.3f} Grow/s   {alg / ms * 1e-6:8.1f} GB/s algorithmic")
