"""GPU box: time of the LDE inside one commitment (lurkhip_commit_dev, the prover's path; span "lde") per matrix shape:
   LURKHIP_NTT_FUSED=1 python tools/lde_commit_time.py 19x148 18x114"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lurk_amd
from lurk_amd import commit as cm
from lurk_amd import synth

shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]] or [(20, 78), (19, 148), (18, 114)]
with lurk_amd.Context(0) as ctx:
    for log_n, w in shapes:
        x = synth.field_elements((1 << log_n, w), seed=w)
        xd = torch.from_numpy(x.view(np.int32)).cuda()
        torch.cuda.synchronize()
        for rep in range(2):
            ctx.profile_reset()
            ctx.profile_enable(True)
            for _ in range(5):
                c = cm.commit_dev(ctx, [xd.data_ptr()], [log_n], [w], 1, lurk_amd.REPR_MONTY)
                c.close()
            ctx.sync()
            ctx.profile_enable(False)
        ms, cnt = ctx.profile_read("lde")
        print(f"2^{log_n} x {w:3d}: lde {ms / cnt:7.3f} ms  ({os.environ.get('LURKHIP_NTT_FUSED', '0')} fused, {os.environ.get('LURKHIP_NTT_TILED', '0')} tiled)")
