"""A C program compiled against include/lurkhip.h and linked to the library (tests/abi_client.c) must see what the ctypes
mirror sees: ctx_create -> poseidon2_hash8 -> commit -> commitment_open on the same inputs, compared word for word (and the
hashes with the oracle).  Catches an argument-order or type slip between the header and the library that a by-name check of
the exports (tests/test_abi.py) cannot."""
import os
import subprocess

import numpy as np
import pytest

import lurk_amd
from lurk_amd import commit as cm
from lurk_amd.poseidon import PoseidonChipset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2013265921
pytestmark = pytest.mark.gpu


def splitmix_elems(n, state=0x4C55524B):
    out = []
    mask = (1 << 64) - 1
    for _ in range(n):
        state = (state + 0x9E3779B97F4A7C15) & mask
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z ^= z >> 31
        out.append(z % P)
    return out, state


def test_c_client_through_the_header_equals_the_ctypes_mirror(ctx, oracle, tmp_path):
    exe = str(tmp_path / "abi_client")
    lib_dir = os.path.join(ROOT, "lurk_amd")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_client.c"),
                        "-o", exe, "-L" + lib_dir, "-llurkhip", "-Wl,-rpath," + lib_dir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = {ln.split()[0]: [int(x) for x in ln.split()[1:]] for ln in r.stdout.splitlines()}
    # the same through lurk_amd (ctypes)
    n_hash, w, lh0, w0, lh1, w1 = 5, 24, 6, 11, 4, 3
    vals, st = splitmix_elems(n_hash * w)
    pre = np.array(vals, dtype=np.uint32).reshape(n_hash, w)
    dig = PoseidonChipset(ctx, w).hash_batch(pre)
    assert got["hash8"] == dig.reshape(-1).tolist()
    assert np.array_equal(dig, oracle.p2_hash8(w, pre))
    v0, st = splitmix_elems((1 << lh0) * w0, st)
    v1, st = splitmix_elems((1 << lh1) * w1, st)
    m0 = np.array(v0, dtype=np.uint32).reshape(1 << lh0, w0)
    m1 = np.array(v1, dtype=np.uint32).reshape(1 << lh1, w1)
    c = cm.commit(ctx, [m0, m1], log_blowup=1)
    assert got["root"] == got["root_again"] == [int(x) for x in c.root]
    rows, path = c.open(37)
    assert got["rows"] == rows.tolist() and got["path"] == path.reshape(-1).tolist()
    assert got["abi"] == [lurk_amd._native.lib.lurkhip_abi_version()]
    c.close()
