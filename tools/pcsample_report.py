"""Turns tools/pcsample.so's PC offsets into a per-function / per-line histogram:
   python tools/pcsample_report.py <lib.so> <pcs.txt> [--lines N]"""
import collections, subprocess, sys

lib, pcs = sys.argv[1], sys.argv[2]
n_lines = int(sys.argv[sys.argv.index("--lines") + 1]) if "--lines" in sys.argv else 40
addrs = [l.strip() for l in open(pcs) if l.strip()]
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "--obj=" + lib, "-f", "-C"], input="\n".join(addrs) + "\n", capture_output=True, text=True).stdout
funcs, lines = collections.Counter(), collections.Counter()
for blk in out.strip().split("\n\n"):
    ls = blk.splitlines()
    if len(ls) < 2:
        continue
    funcs[ls[-2][:100]] += 1                                   # outermost (non-inlined) function
    lines[ls[1].split("/")[-1] + "  " + ls[0][:60]] += 1       # innermost inlined frame
tot = len(addrs)
print(f"{tot} samples")
for k, v in funcs.most_common(25):
    print(f"{100 * v / tot:6.2f} %  {k}")
print("-- by line (innermost inlined frame)")
for k, v in lines.most_common(n_lines):
    print(f"{100 * v / tot:6.2f} %  {k}")
if "--outer" in sys.argv:
    # samples by the line of the outermost frame (where in the non-inlined function the time goes, inlined callees included)
    outer = collections.Counter()
    for blk in out.strip().split("\n\n"):
        ls = blk.splitlines()
        if len(ls) >= 2:
            outer[ls[-1].split("/")[-1].rsplit(":", 1)[0] + "  " + ls[-2][:40]] += 1
    print("-- by line of the outermost frame")
    for k, v in outer.most_common(n_lines):
        print(f"{100 * v / tot:6.2f} %  {k}")
