#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of one bench run (gpurun_out/<tag>...) into the summaries kept under profiles/.

usage: python tools/summarise_profiles.py <tag> [<prefix> [<bench json>]]     e.g. r02c r02 gpurun_out/bench_ntt10.json
expects gpurun_out/prof_<tag>/bench_kernel_stats.csv, gpurun_out/pmc_<tag>_{FETCH_SIZE,WRITE_SIZE,sq}/bench_counter_collection.csv,
gpurun_out/bench_<tag>.json
"""
import collections
import csv
import json
import re
import shutil
import sys

tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "r02"
bench_json = sys.argv[3] if len(sys.argv) > 3 else f"gpurun_out/bench_{tag}.json"
bench = json.load(open(bench_json))


def kname(s):
    m = re.search(r"(k_lde_[a-z]+<[0-9, a-z]*>|k_[a-z_0-9]+|jit_[a-z_]+)", s)  # the grouped LDE's kernels keep their tile shape
    return m.group(1).replace(" ", "") if m else s[:40]


res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, cnt = collections.Counter(), collections.Counter()
    with open(f"gpurun_out/pmc_{tag}_{c}/bench_counter_collection.csv") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != c:
                continue
            n = kname(row["Kernel_Name"])
            tot[n] += float(row["Counter_Value"])
            cnt[n] += 1
    res[c] = (tot, cnt)
with open(f"profiles/{prefix}_pmc_hbm_per_kernel.csv", "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python bench.py --steps 1 --warmup 0`\n")
    f.write("# units: KB as reported; MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, WRITE_SIZE uncalibrated\n")
    f.write("kernel,launches,FETCH_SIZE_KB,WRITE_SIZE_KB\n")
    for k in sorted(res["FETCH_SIZE"][0], key=lambda k: -(res["FETCH_SIZE"][0][k] + res["WRITE_SIZE"][0][k])):
        f.write(f"\"{k}\",{res['FETCH_SIZE'][1][k]},{res['FETCH_SIZE'][0][k]:.0f},{res['WRITE_SIZE'][0][k]:.0f}\n")  # (quoted: template arguments hold commas)
HASH = ("k_row_sponges", "k_level_digests", "k_level_coop", "k_level", "k_leaves", "k_leaves_coop")  # the hashing launches (bench.py's merkle_leaves + merkle_levels spans; the PMC pass also counts the small FRI-layer trees)
F = sum(res["FETCH_SIZE"][0][k] for k in HASH) * 1024
W = sum(res["WRITE_SIZE"][0][k] for k in HASH) * 1024

tot, dur, cnt, seen = collections.defaultdict(collections.Counter), collections.Counter(), collections.Counter(), set()
with open(f"gpurun_out/pmc_{tag}_sq/bench_counter_collection.csv") as f:
    for row in csv.DictReader(f):
        n = kname(row["Kernel_Name"])
        tot[n][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"])
            dur[n] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            cnt[n] += 1
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"]
with open(f"profiles/{prefix}_pmc_sq_per_kernel.csv", "w") as f:
    f.write("# rocprofv3 --pmc " + " ".join(cols) + " --kernel-trace, `python bench.py --steps 1 --warmup 0`; valu_tinst_s = SQ_INSTS_VALU * 64 lanes / duration (full rate 78.6, half rate 39.3 T lane-instr/s)\n")
    f.write("kernel,launches,ms," + ",".join(cols) + ",valu_tinst_s\n")
    for n in sorted(dur, key=lambda k: -dur[k]):
        v = tot[n]
        rate = v["SQ_INSTS_VALU"] * 64 / (dur[n] * 1e-9) / 1e12 if dur[n] else 0
        f.write(f"\"{n}\",{cnt[n]},{dur[n] / 1e6:.3f}," + ",".join(f"{v[c]:.4g}" for c in cols) + f",{rate:.2f}\n")
hv = sum(tot[k]["SQ_INSTS_VALU"] for k in HASH) * 64 / (sum(dur[k] for k in HASH) * 1e-9) / 1e12
NTT = [k for k in res["FETCH_SIZE"][0] if k.startswith("k_ntt_pass") or k.startswith("k_lde_")]  # every coset-LDE launch (lde.hip + ntt.hip)
hash_lane_insts = sum(tot[k]["SQ_INSTS_VALU"] for k in HASH) * 64
ntt_bytes = (2 * sum(res["FETCH_SIZE"][0][k] for k in NTT) + sum(res["WRITE_SIZE"][0][k] for k in NTT)) * 1024
traffic = {"log_rows": 20, "workload": "fib-mix", "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 0 (profiles/{prefix}_pmc_hbm_per_kernel.csv)",
           "ntt_pass_bytes_per_step": ntt_bytes,
           "correction": "calibrated (profiles/r03_pmc_calibration.json, tools/ubench_fetch.hip): FETCH_SIZE x 2.0 for every access pattern of this repository (a 128-byte line request is tallied as 64 bytes whatever the load width), WRITE_SIZE x 1.0 (exact in 32-byte sectors)",
           "fetch_factor": 2.0, "write_factor": 1.0,
           "ntt_pass_fetch_bytes_reported": sum(res["FETCH_SIZE"][0][k] for k in NTT) * 1024, "ntt_pass_write_bytes_reported": sum(res["WRITE_SIZE"][0][k] for k in NTT) * 1024,
           "merkle_hash_fetch_bytes_reported": F, "merkle_hash_write_bytes_reported": W, "merkle_hash_bytes_per_step": 2 * F + W,
           "merkle_hash_valu_tinst_s": hv, "merkle_hash_valu_lane_insts_per_step": hash_lane_insts,
           "merkle_hash_algorithmic_bytes_per_step": bench["roofline"]["algorithmic_bytes_per_step"],  # the shapes the instruction count belongs to
           "lde_valu_lane_insts_per_step": sum(tot[k]["SQ_INSTS_VALU"] for k in tot if k.startswith("k_ntt_pass") or k.startswith("k_lde_")) * 64,
           "lde_kernels": sorted(NTT), "merkle_hash_mul_class_frac": 0.6, "valu_full_rate_tinst_s": 78.6, "valu_half_rate_tinst_s": 39.3}
# the instruction count belongs to these sources: bench.py recomputes the hash and stops using the count when they have changed
import hashlib
import os

HASH_SOURCES = ["lurk_amd/csrc/merkle.hip", "lurk_amd/csrc/poseidon2_dev.h", "lurk_amd/csrc/p16_coop.h", "lurk_amd/csrc/babybear.h", "lurk_amd/csrc/merkle.h"]  # (the files whose code the hashing kernels contain; commit.h's host-side declarations are not among them: round 6)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for rel in HASH_SOURCES:
    h.update(open(os.path.join(root, rel), "rb").read())
# the whole step: every kernel's VALU instructions (bench.py: roofline.step), tied to every source a kernel of the step is made of
import glob

traffic["step_valu_lane_insts"] = sum(v["SQ_INSTS_VALU"] for v in tot.values()) * 64
STEP_SOURCES = sorted(os.path.relpath(f, root) for pat in ("lurk_amd/csrc/*.hip", "lurk_amd/csrc/*.h", "lurk_amd/csrc/lair/trace_program.h") for f in glob.glob(os.path.join(root, pat)))
hs = hashlib.sha256()
for rel in STEP_SOURCES:
    hs.update(open(os.path.join(root, rel), "rb").read())
traffic["step_kernel_sources"] = STEP_SOURCES
traffic["step_kernel_sources_sha256"] = hs.hexdigest()
traffic["hash_kernel_sources"] = HASH_SOURCES
traffic["hash_kernel_sources_sha256"] = h.hexdigest()
json.dump(traffic, open(f"profiles/{prefix}_pmc_traffic.json", "w"), indent=1)
json.dump(traffic, open("profiles/pmc_traffic.json", "w"), indent=1)  # the copy bench.py reads (labelled static there)
shutil.copy(f"gpurun_out/prof_{tag}/bench_kernel_stats.csv", f"profiles/{prefix}_full_prove_kernel_stats.csv")
shutil.copy(bench_json, f"profiles/{prefix}_bench_full_prove.json")
with open(f"profiles/{prefix}_full_prove_under_rocprof.log", "w") as f:
    f.write("".join(l for l in open(f"gpurun_out/prof_{tag}/bench.log") if l.startswith("{") or "rocprofv3" in l)[:6000])
print("merkle hash: traffic", 2 * F + W, "B/step; VALU", round(hv, 2), "Tinstr/s; NTT passes", ntt_bytes / 1e9, "GB/step")
