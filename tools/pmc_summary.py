#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 counter_collection.csv, with every counter as a ratio to SQ_WAVE_CYCLES when present.
usage: python tools/pmc_summary.py <counter_collection.csv> [name filter]"""
import collections
import csv
import re
import sys

d = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    m = re.search(r"(k_\w+(<[^>]*>)?|jit_\w+)", k)
    k = m.group(1) if m else k[:40]
    if flt and flt not in k:
        continue
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in sorted(d):
    c = d[k]
    wc = c.get("SQ_WAVE_CYCLES", 0)
    disp = max(v for (kk, _), v in n.items() if kk == k)
    parts = []
    for name, v in sorted(c.items()):
        parts.append(f"{name.replace('SQ_', '')}={v:.3g}" + (f"({v / wc:.2f})" if wc and name != "SQ_WAVE_CYCLES" else ""))
    print(f"{k:30s} n={disp:3d} " + " ".join(parts))
