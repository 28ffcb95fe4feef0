#!/bin/bash
# GPU box: the round-end sequence the driver runs (gpu tests, smoke, default bench), then the rocprofv3 passes of tools/prof_bench.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q > gpurun_out/final/gpu_tests.log 2>&1; tail -2 gpurun_out/final/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
( time python bench.py ) > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -4 gpurun_out/final/bench.err
bash tools/prof_bench.sh final pmc
# the driver's line must price the hashing with the LIVE rate: refuse to finish when bench.py would take the stale-hash fallback
#  -- the PMC pass above is of this tree, so after tools/summarise_profiles.py this must hold
python - <<'PY'
import hashlib, json, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", ".")
pmc = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
h = hashlib.sha256()
for rel in pmc["hash_kernel_sources"]:
    h.update(open(os.path.join(root, rel), "rb").read())
if h.hexdigest() != pmc["hash_kernel_sources_sha256"]:
    sys.exit("profiles/pmc_traffic.json is of another tree: run tools/summarise_profiles.py on this run's passes before finishing")
print("pmc_traffic.json: hashing kernels' sources match")
PY
