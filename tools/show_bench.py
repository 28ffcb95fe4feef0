#!/usr/bin/env python3
"""Print the headline and stage times of bench.py JSON lines (diagnostic helper): python tools/show_bench.py a.json [b.json ...]"""
import json
import sys

rows = []
for path in sys.argv[1:]:
    with open(path) as f:
        lines = [l for l in f if l.startswith("{")]
    if not lines:
        print(path, "no JSON line")
        continue
    d = json.loads(lines[-1])
    rows.append((path, d))
keys = []
for _, d in rows:
    for k in d["config"].get("stages_ms", {}):
        if k not in keys:
            keys.append(k)
print("%-18s" % "", *["%12s" % p.split("/")[-1][:12] for p, _ in rows])
print("%-18s" % "ms_per_step", *["%12.3f" % d["ms_per_step"] for _, d in rows])
print("%-18s" % "M eval-steps/s", *["%12.3f" % (d["value"] / 1e6) for _, d in rows])
for k in keys:
    print("%-18s" % k, *["%12.3f" % d["config"]["stages_ms"].get(k, float("nan")) for _, d in rows])
for _, d in rows:
    c = d["config"]
    print({k: c.get(k) for k in ("proofs_identical_across_steps", "grand_sum_is_zero", "host_execute_s", "host_flatten_upload_s")},
          "in flight:", c.get("proofs_in_flight"), "sequential ms:", (c.get("sequential") or {}).get("ms_per_step"),
          "(r02 two_in_flight:", (c.get("two_shards_in_flight") or {}).get("ms_per_shard"), ")")
    if c.get("host_pipeline"):
        print("host_pipeline:", c["host_pipeline"])
