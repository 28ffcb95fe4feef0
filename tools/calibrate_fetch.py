#!/usr/bin/env python3
"""Reads gpurun_out/calib/{known_bytes.json,FETCH_SIZE.csv,WRITE_SIZE.csv} (tools/calibrate_fetch.sh on a GPU box) and
prints / returns, per access pattern of tools/ubench_fetch.hip, counter bytes / known bytes."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def calibrate(d=os.path.join(ROOT, "gpurun_out", "calib")):
    known = json.load(open(os.path.join(d, "known_bytes.json")))
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot = collections.defaultdict(list)
        for row in csv.DictReader(open(os.path.join(d, c + ".csv"))):
            if row["Counter_Name"] != c:
                continue
            m = re.search(r"(k_[a-z_0-9]+(<\d+u?>)?)", row["Kernel_Name"])
            tot[m.group(1).replace("u>", ">")].append(float(row["Counter_Value"]) * 1024)  # the counters are reported in KB
        for k, v in tot.items():
            if k in known and (c == "FETCH_SIZE") == k.startswith("k_read"):
                out[k] = {"counter": c, "known_bytes": known[k], "counter_bytes": sum(v) / len(v), "counter_over_known": sum(v) / len(v) / known[k]}
    return out


if __name__ == "__main__":
    res = calibrate(*sys.argv[1:2])
    for k, v in res.items():
        print(f"{k:22s} {v['counter']:10s} known {v['known_bytes'] / 1e6:9.1f} MB  counter {v['counter_bytes'] / 1e6:9.1f} MB  ratio {v['counter_over_known']:.4f}")
