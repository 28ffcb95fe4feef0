#!/usr/bin/env python3
"""Static instruction statistics of one kernel in a `hipcc -S --cuda-device-only` listing:
   python tools/kernel_instr_count.py file.s k_row_sponges"""
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r"\n(_Z\w*%s\w*):[^\n]*\n(.*?)\n\.Lfunc_end" % re.escape(name), s, re.S)
if not m:
    sys.exit("kernel not found")
body = m.group(2)
ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.lstrip().startswith((";", "."))]
from collections import Counter

c = Counter(ins)
print(f"{name}: {len(ins)} static instructions; s_nop {c['s_nop']}, v_mad_i64_i32 {c['v_mad_i64_i32']}, v_mul_lo_u32 {c['v_mul_lo_u32']}, s_getpc {c['s_getpc_b64']}")
meta = re.search(r"\.vgpr_count:\s*(\d+)", s[m.end():])
for k, v in c.most_common(12):
    print(f"   {k:28s}{v}")
