#!/bin/bash
# One command for a machine WITH cargo and network (the build image has neither): produce the upstream vector file from the real
# sphinx / Plonky3 crates and run the loader tests that turn S1 from "unpinned" into pinned.
#   tools/pin_s1.sh [--shard-proof /path/to/lurk-checkout]
set -euo pipefail
here=$(cd "$(dirname "$0")/.." && pwd)
command -v cargo >/dev/null || { echo "pin_s1.sh: cargo not found -- this kit cannot run in the build image (no Rust toolchain)"; exit 2; }
features=()
if [ "${1:-}" = "--shard-proof" ]; then
    sed -i "s|path = \"../../../reference\"|path = \"$2\"|" "$here/tools/upstream_dump/Cargo.toml"
    features=(--features shard-proof)
fi
out="$here/tests/golden/upstream/sphinx_8a39b951.json"
(cd "$here/tools/upstream_dump" && cargo run --release "${features[@]}") > "$out"
echo "wrote $out"
cd "$here"
python -m pytest tests/test_upstream_vectors.py -q                      # CPU oracle against the vectors
python -m pytest tests/test_profile_gpu.py -q -m gpu -k upstream || true  # HIP library (needs a GPU)
