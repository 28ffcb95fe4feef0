"""The C-ABI library loads on a CPU-only box and exports every symbol include/lurkhip.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lurkhip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lurkhip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lurk_amd import _native

    lib = ctypes.CDLL(_native.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 19
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in lurkhip.h but not exported: {missing}"


def test_python_binding_covers_the_header():
    from lurk_amd import _native

    assert sorted(_native.SIGNATURES) == declared_symbols()


def test_abi_version_and_static_queries():
    from lurk_amd import _native

    assert _native.lib.lurkhip_abi_version() >= 1
    assert _native.lib.lurkhip_poseidon2_num_cols(24) == 449
    assert _native.lib.lurkhip_poseidon2_num_cols(32) == 603
    assert _native.lib.lurkhip_poseidon2_num_cols(40) == 755
    assert _native.lib.lurkhip_poseidon2_num_cols(17) < 0


def test_no_cpu_fallback_without_device():
    """On a box without a HIP device the product must fail loudly, not fall back."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import lurk_amd

    with pytest.raises(lurk_amd.LurkHipError) as ei:
        lurk_amd.Context(0)
    assert ei.value.status == -2  # LURKHIP_ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    """Nothing under lurk_amd/ may import, call or link anything under oracle/."""
    pkg = os.path.join(ROOT, "lurk_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"(from|import)\s+oracle\b|oracle/|liblurkoracle", text):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders


def test_trace_group_layout_is_host_only_and_aligned(monkeypatch):
    """lurkhip_trace_group_layout (round 5): the matrices of one height share a buffer whose pitch is a whole number of 128-byte
    lines when that costs at most half more memory; every matrix gets exactly one column range; no device is touched."""
    import ctypes as C

    import numpy as np

    from lurk_amd import _native as N

    monkeypatch.setenv("LURKHIP_SRC_PADDED", "1")  # off by default: then every matrix is dense
    lh = np.array([20, 19, 19, 19, 3, 19, 16, 16, 2, 12], dtype=np.uint32)
    ws = np.array([78, 148, 107, 114, 5, 7, 13, 64, 9, 20], dtype=np.uint32)
    n = len(lh)
    pitch, col = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
    grp, ng = np.full(n, -1, dtype=np.int32), np.zeros(1, dtype=np.int32)
    assert N.lib.lurkhip_trace_group_layout(n, lh.ctypes.data, ws.ctypes.data, pitch.ctypes.data, col.ctypes.data, grp.ctypes.data, ng.ctypes.data) == N.OK
    assert sorted(set(grp.tolist())) == list(range(int(ng[0])))
    for g in range(int(ng[0])):
        idx = [i for i in range(n) if grp[i] == g]
        assert len({int(lh[i]) for i in idx}) == 1 and len({int(pitch[i]) for i in idx}) == 1
        spans = sorted((int(col[i]), int(col[i] + ws[i])) for i in idx)
        assert spans[0][0] == 0 and all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= int(pitch[idx[0]])
        if len(idx) > 1 or int(pitch[idx[0]]) != int(ws[idx[0]]):
            assert int(pitch[idx[0]]) % 32 == 0
    # the four 2^19-row matrices (376 columns) share one buffer of 384; the 2^20 x 78 eval trace gets pitch 96; 20 columns alone
    # would cost 60 % more: dense
    assert len({int(grp[i]) for i in (1, 2, 3, 5)}) == 1 and int(pitch[1]) == 384
    assert int(pitch[0]) == 96 and int(pitch[9]) == 20
    # below 2^5 rows the grouped LDE does not run: dense
    assert int(pitch[4]) == 5 and int(pitch[8]) == 9
    assert N.lib.lurkhip_trace_group_layout(n, None, ws.ctypes.data, pitch.ctypes.data, col.ctypes.data, grp.ctypes.data, ng.ctypes.data) == N.ERR_INVALID_ARG
    monkeypatch.delenv("LURKHIP_SRC_PADDED")
    assert N.lib.lurkhip_trace_group_layout(n, lh.ctypes.data, ws.ctypes.data, pitch.ctypes.data, col.ctypes.data, grp.ctypes.data, ng.ctypes.data) == N.OK
    assert pitch.tolist() == ws.tolist() and int(ng[0]) == n and not col.any()
