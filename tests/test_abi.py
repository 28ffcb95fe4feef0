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
