"""lurk_amd -- MI355X-native proving hot path for Lurk behind the lurkhip C ABI.

Host-side mirror of the reference's operator interfaces for this path
(`Chipset`, `FuncChip::generate_trace`, `MemChip`, `BytesChip`, the commit
stages) over ``liblurkhip.so``.  The HIP library is the product; this package
is the thin harness the tests and ``bench.py`` drive it through.
"""
from ._native import (  # noqa: F401
    LurkHipError,
    REPR_CANONICAL,
    REPR_MONTY,
    LIB_PATH,
)
from .context import Context  # noqa: F401
from .field import P  # noqa: F401
