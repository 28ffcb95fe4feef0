#!/usr/bin/env python3
"""Instruction mix per kernel of a gfx950 assembly listing (hipcc --save-temps): counts of the opcodes that tell where registers
and time go (scratch traffic, global / LDS accesses, products, lane moves, barriers)."""
import re
import subprocess
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
name, body, kernels = None, [], []
for l in lines:
    m = re.match(r"^(_Z\S+):", l)
    if m:
        name, body = m.group(1), []
        continue
    t = l.strip()
    if name and t and not t.startswith((";", ".")):
        body.append(t)
        if t.startswith("s_endpgm"):
            kernels.append((name, body))
            name = None
names = subprocess.run(["c++filt"], input="\n".join(k[0] for k in kernels), capture_output=True, text=True).stdout.split("\n")
KEYS = ("scratch_", "global_load", "global_store", "ds_read", "ds_write", "s_barrier", "v_mad_i64", "v_mul_lo", "v_readfirstlane", "v_writelane",
        "v_readlane", "v_mov_b32", "v_accvgpr", "s_load", "v_mad_u64", "v_lshl_add_u64", "v_add_co", "v_addc")
for (k, body), n in zip(kernels, names):
    n = n.replace("lurkhip::(anonymous namespace)::", "").replace("void ", "")
    if flt and flt not in n:
        continue
    c = Counter()
    for t in body:
        op = t.split()[0]
        for key in KEYS:
            if op.startswith(key):
                c[key] += 1
    print(f"{n[:48]:48s} total={len(body):6d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
