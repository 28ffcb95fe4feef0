"""The library's schedule switches (A/B hooks: side lanes, fused tree sponges, grouped cooperative levels, sixteen-lane rows,
round-1 opening kernels, trace interpreter, NTT tile caps) choose HOW a proof is computed, never WHAT: the same sharded execution
proved in a fresh process under each of them must give the default schedule's proof words."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r"""
import hashlib, sys
import numpy as np
import lurk_amd
from lurk_amd import lair, prover
from lurk_amd.programs import lurk_mix as lm
mix = lm.fib_mix(1 << 10)
top = lair.Toplevel(mix.source, lurk_chips=True)
q = lair.QueryRecord(top)
top.execute(top.func_index(mix.entry), mix.main_args, q)
pv = q.expect_public_values()
with lurk_amd.Context(0) as ctx:
    m = prover.Machine(ctx, top, mix.entry, len(pv))
    m.setup()
    proofs = m.prove(q, lair.ShardingConfig(1 << 9), num_queries=6, pow_bits=3)
    h = hashlib.sha256()
    for p in proofs:
        h.update(np.ascontiguousarray(p.words, dtype=np.uint32).tobytes())
    m.close()
print("PROOF", len(proofs), h.hexdigest())
"""

SWITCHES = [
    {},
    {"LURKHIP_SIDE_LANE": "0"},
    {"LURKHIP_SIDE_LANES": "1"},
    {"LURKHIP_EARLY_SPONGE": "1"},  # the hash chips' row sponge ahead on the hash stream, under the other groups' LDE passes
    {"LURKHIP_MERKLE_FUSED": "0"},
    {"LURKHIP_MERKLE_COOP_GROUP": "1"},
    {"LURKHIP_SPONGE_COOP": "0"},
    {"LURKHIP_DOT_OLD": "1", "LURKHIP_OPENINGS_NO_QUAD": "1"},
    {"LURKHIP_TRACE_INTERPRET": "1", "LURKHIP_NTT_MAX_LOG_R": "7"},
    {"LURKHIP_REDUCE_ROWS": "0"},  # reduced openings through the round-3 kernels (tiles staged in LDS)
    {"LURKHIP_LDE_PADDED": "0"},  # every LDE in its own dense buffer (round 3's layout)
    {"LURKHIP_LDE_PADDED": "2"},  # every height group in one padded buffer, whatever the padding costs
    {"LURKHIP_LDE_PADDED": "2", "LURKHIP_REDUCE_SLICE_W": "32", "LURKHIP_REDUCE_ROWS": "0"},  # every height group in one padded buffer, whatever the padding costs
]


def _run(extra):
    env = dict(os.environ)
    env.update(extra)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-c", SNIPPET], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("PROOF")][-1]
    return line


def test_schedule_switches_do_not_change_the_proof():
    want = _run(SWITCHES[0])
    assert int(want.split()[1]) >= 2
    for extra in SWITCHES[1:]:
        assert _run(extra) == want, extra
