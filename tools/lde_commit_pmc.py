"""GPU box: one commitment of a single 2^log_n x w device matrix (lurkhip_commit_dev: the LDE path the prover takes, with
chunk-tiled intermediates unless LURKHIP_NTT_TILED=0), for PMC passes over its k_ntt_pass launches:
   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/lde_commit_pmc.py 20 78"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lurk_amd
from lurk_amd import commit as cm
from lurk_amd import synth

log_n, w = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
with lurk_amd.Context(0) as ctx:
    x = synth.field_elements((1 << log_n, w), seed=w)
    xd = torch.from_numpy(x.view(np.int32)).cuda()
    torch.cuda.synchronize()
    for _ in range(reps):
        c = cm.commit_dev(ctx, [xd.data_ptr()], [log_n], [w], 1, lurk_amd.REPR_MONTY)
        ctx.sync()
        c.close()
    ctx.span_begin("x") if False else None
