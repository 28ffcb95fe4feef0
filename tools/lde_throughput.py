"""GPU box only: coset-LDE (blow-up 2) time per matrix shape with input and output resident in HBM."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lurk_amd
from lurk_amd import commit as cm
from lurk_amd import synth

shapes = [(20, 78), (20, 96), (21, 9), (20, 6), (20, 8), (19, 36), (20, 4), (16, 13)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
with lurk_amd.Context(0) as ctx:
    for log_n, w in shapes:
        n = 1 << log_n
        x = synth.field_elements((n, w), seed=w)
        xd = torch.from_numpy(x.view(np.int32)).cuda()
        od = torch.empty((2 * n, w), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        cm.coset_lde_dev(ctx, log_n, w, 1, xd.data_ptr(), od.data_ptr(), lurk_amd.REPR_MONTY)
        ctx.sync()
        ctx.timer_start()
        for _ in range(5):
            cm.coset_lde_dev(ctx, log_n, w, 1, xd.data_ptr(), od.data_ptr(), lurk_amd.REPR_MONTY)
        ms = ctx.timer_stop() / 5
        hbm = 9 * 2 * n * w * 4  # 3 transforms x 3 passes, read + write
        print(f"2^{log_n} x {w:3d}: {ms:7.3f} ms  {n * w / ms * 1e-6:7.2f} Gelem/s  {hbm / ms * 1e-6:7.1f} GB/s (9 r+w passes)")
