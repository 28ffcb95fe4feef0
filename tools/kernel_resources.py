#!/usr/bin/env python3
"""Per-kernel registers / scratch / occupancy from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr saved to a file).

usage: hipcc ... -c x.hip -Rpass-analysis=kernel-resource-usage 2> res.txt; python tools/kernel_resources.py res.txt [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, n in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    n = n.replace("lurkhip::(anonymous namespace)::", "").replace("void ", "")
    if flt and flt not in n:
        continue
    scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{n[:64]:64s} vgpr={g('VGPRs'):4d} sgpr={g('SGPRs'):4d} scratch={scratch:5d} occ={occ} lds={lds}")
