#!/usr/bin/env python3
"""Idle time of the device inside the timed steps of a `--lanes 1` bench run, from rocprofv3's kernel trace
(gpurun_out/prof_<tag>/seq_kernel_trace.csv): the union of the kernels' [start, end) intervals against the span of each
step, and the gaps longer than a threshold (host transcript round trips, launch gaps).
   python tools/trace_gaps.py gpurun_out/prof_r03/seq_kernel_trace.csv [steps=4]"""
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed steps are the last `steps` proofs: cut at the k_trace_func launches that open a proof (the first trace kernel after a FRI tail)
opens = [i for i, r in enumerate(rows) if "k_pow_grind" in r[2]]
assert len(opens) >= steps, "not enough proofs in the trace"
# a proof = from the kernel after the previous proof's last launch to its own last launch (the query gather after the grind)
ends = []
for i in opens:
    j = i
    while j + 1 < len(rows) and "k_gather_openings" in rows[j + 1][2]:
        j += 1
    ends.append(j)
res = []
for k in range(len(ends) - steps, len(ends)):
    lo = ends[k - 1] + 1 if k > 0 else 0
    seg = rows[lo:ends[k] + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    busy, cur_s, cur_e, gaps = 0, seg[0][0], seg[0][1], []
    for s, e, name in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, name))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    big = sorted(gaps, reverse=True)[:8]
    res.append((t1 - t0, busy, gaps, big, len(seg)))
for span, busy, gaps, big, n in res:
    g20 = [g for g, _ in gaps if g > 20000]
    print(f"proof: span {span / 1e6:.2f} ms, {n} launches, device busy {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.3f} ms "
          f"({sum(g20) / 1e6:.3f} ms in {len(g20)} gaps > 20 us; the rest in {len(gaps) - len(g20)} short launch gaps)")
    print("   largest gaps (us, next kernel):", [(round(g / 1e3), nm.split("(")[0][-40:]) for g, nm in big[:6]])
