#!/bin/bash
# GPU box: A/B of macros of the run-time compiled kernels (LURKHIP_JIT_DEFINES), one bench line per variant.
# usage: tools/ab_jit_defines.sh <out dir> <variant> [<variant> ..]   (variant: "NAME=V,NAME=V" or "-" for the defaults)
out=$1; shift
mkdir -p $out
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then unset LURKHIP_JIT_DEFINES; else export LURKHIP_JIT_DEFINES="$v"; fi
  python bench.py --no-host-pipeline --no-second-profile --no-cpu-baseline ${AB_ARGS:---compile-min-log-rows 13 --steps 10} > $out/v$i.json 2> $out/v$i.err
  python - "$v" $out/v$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    st = d["config"]["sequential"]["stages_ms"] if "sequential" in d["config"] else d.get("stages_ms", {})
    print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 2), "latency", round(d.get("proof_latency_ms") or 0, 2),
          {k: round(st.get(k, 0), 2) for k in ("permutation", "quotient_all", "commit_main", "commit_perm", "open", "trace_all")})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
