"""Host-side mirror of the reference's Lair interface over the lurkhip C ABI.

Names follow /root/reference/src/lair/: `Toplevel` (toplevel.rs), `QueryRecord` + `execute_by_name`
(execute.rs:375-417), `FuncChip.from_name / width / layout_sizes / generate_trace`
(func_chip.rs:34-80, trace.rs:72-135), `MemChip` (memory.rs), `BytesChip` (gadgets/bytes/trace.rs),
`Shard` / `ShardingConfig` (execute.rs:77-124,226-241).  Functions are written in the surface syntax of
the reference's `func!` macro and handed over as text.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native as N
from .context import Context, _addr

DEFAULT_SHARD_SIZE = 1 << 22  # execute.rs:233
MEM_TABLE_SIZES = (2, 3, 4, 5, 6, 8)  # execute.rs:243-244


class LairError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"lurkhip status {status}: {message}")
        self.status = status
        self.message = message


def _check(status):
    if status != N.OK:
        raise LairError(status, N.lib.lurkhip_lair_last_error().decode("utf-8", "replace"))


@dataclass(frozen=True)
class LayoutSizes:  # func_chip.rs:11-26
    nonce: int
    input: int
    aux: int
    sel: int
    output: int

    def total(self) -> int:
        return self.nonce + self.input + self.aux + self.sel + self.output


@dataclass(frozen=True)
class ShardingConfig:
    max_shard_size: int = DEFAULT_SHARD_SIZE


class Toplevel:
    """Compiled Lair functions (+ optionally the native Lurk chips for extern_call)."""

    def __init__(self, source: str, lurk_chips: bool = False):
        h = C.c_void_p()
        _check(N.lib.lurkhip_toplevel_new(source.encode(), int(lurk_chips), C.byref(h)))
        self.handle = h

    @classmethod
    def new_pure(cls, source: str) -> "Toplevel":  # toplevel.rs:52-55
        return cls(source, lurk_chips=False)

    @classmethod
    def from_bytecode(cls, blob) -> "Toplevel":
        """A toplevel from compiled functions ("LBC1" words, lurk_amd/csrc/lair/bytecode_io.cpp): what a host with its own
        compiler (the reference's Toplevel::new, toplevel.rs:38-72) hands over instead of source text."""
        words = np.ascontiguousarray(blob, dtype=np.uint32)
        self = cls.__new__(cls)
        h = C.c_void_p()
        _check(N.lib.lurkhip_toplevel_from_bytecode(_addr(words), words.size, C.byref(h)))
        self.handle = h
        return self

    def to_bytecode(self) -> np.ndarray:
        n = N.lib.lurkhip_toplevel_to_bytecode(self.handle, None, 0)
        if n < 0:
            _check(int(n))
        out = np.zeros(n, dtype=np.uint32)
        if N.lib.lurkhip_toplevel_to_bytecode(self.handle, _addr(out), n) != n:
            raise RuntimeError("lurkhip_toplevel_to_bytecode: size changed")
        return out

    def __del__(self):
        if getattr(self, "handle", None) and N is not None:
            N.lib.lurkhip_toplevel_free(self.handle)
            self.handle = None

    def num_funcs(self) -> int:
        return N.lib.lurkhip_toplevel_num_funcs(self.handle)

    def func_index(self, name: str) -> int:
        i = N.lib.lurkhip_toplevel_func_index(self.handle, name.encode())
        if i < 0:
            raise KeyError(f"Func {name} not found")
        return i

    def func_info(self, idx: int) -> dict:
        info = np.zeros(9, dtype=np.uint32)
        _check(N.lib.lurkhip_toplevel_func_info(self.handle, idx, _addr(info)))
        i = [int(x) for x in info]
        return {
            "input_size": i[0],
            "output_size": i[1],
            "partial": bool(i[2]),
            "invertible": bool(i[3]),
            "layout": LayoutSizes(nonce=i[4], input=i[5], output=i[6], aux=i[7], sel=i[8]),
        }

    def execute_by_name(self, name: str, args, queries: "QueryRecord") -> list[int]:
        return self.execute(self.func_index(name), args, queries)

    def execute(self, func_idx: int, args, queries: "QueryRecord") -> list[int]:
        a = np.ascontiguousarray(args, dtype=np.uint32)
        out = np.zeros(max(self.func_info(func_idx)["output_size"], 1), dtype=np.uint32)
        _check(N.lib.lurkhip_execute(queries.handle, func_idx, _addr(a), len(a), _addr(out)))
        return [int(x) for x in out[: self.func_info(func_idx)["output_size"]]]


class QueryRecord:
    def __init__(self, toplevel: Toplevel):
        self.toplevel = toplevel
        h = C.c_void_p()
        _check(N.lib.lurkhip_record_new(toplevel.handle, C.byref(h)))
        self.handle = h

    def __del__(self):
        if getattr(self, "handle", None) and N is not None:
            N.lib.lurkhip_record_free(self.handle)
            self.handle = None

    def clean(self):
        _check(N.lib.lurkhip_record_clean(self.handle))

    def num_func_queries(self, func_idx: int) -> int:
        return int(N.lib.lurkhip_record_count(self.handle, 0, func_idx))

    def num_mem_queries(self, mem_len: int) -> int:
        return int(N.lib.lurkhip_record_count(self.handle, 1, mem_len))

    def num_byte_records(self) -> int:
        return int(N.lib.lurkhip_record_count(self.handle, 3, 0))

    def expect_public_values(self) -> list[int]:
        n = int(N.lib.lurkhip_record_count(self.handle, 2, 0))
        if n < 0:
            raise LairError(N.ERR_INVALID_ARG, "Public values not set")
        out = np.zeros(max(n, 1), dtype=np.uint32)
        _check(N.lib.lurkhip_record_public_values(self.handle, _addr(out)))
        return [int(x) for x in out[:n]]

    def inject_inv_query(self, func_idx: int, inp, out):
        i = np.ascontiguousarray(inp, dtype=np.uint32)
        o = np.ascontiguousarray(out, dtype=np.uint32)
        _check(N.lib.lurkhip_record_inject_inv_query(self.handle, func_idx, _addr(i), len(i), _addr(o), len(o)))


@dataclass
class Shard:  # execute.rs:77-124
    queries: QueryRecord
    index: int = 0
    shard_config: ShardingConfig = ShardingConfig()

    @classmethod
    def new(cls, queries: QueryRecord) -> "Shard":
        return cls(queries)

    def shard(self, config: ShardingConfig) -> list["Shard"]:
        n = int(N.lib.lurkhip_record_num_shards(self.queries.handle, config.max_shard_size))
        return [Shard(self.queries, i, config) for i in range(n)]


class FuncChip:
    def __init__(self, ctx: Context, func_idx: int, toplevel: Toplevel):
        self.ctx = ctx
        self.toplevel = toplevel
        self.func_idx = func_idx
        self.info = toplevel.func_info(func_idx)
        self.layout_sizes: LayoutSizes = self.info["layout"]

    @classmethod
    def from_name(cls, ctx: Context, name: str, toplevel: Toplevel) -> "FuncChip":
        return cls(ctx, toplevel.func_index(name), toplevel)

    def width(self) -> int:
        return self.layout_sizes.total()

    def compile_trace(self, ctx: Context | None = None):
        """Compile this function's trace program to a straight-line row kernel (hiprtc / the code-object cache) and use it on
        the context's device from now on (lurkhip_trace_compile)."""
        ctx = ctx or self.ctx
        s = N.lib.lurkhip_trace_compile(ctx.handle, self.toplevel.handle, self.func_idx)
        if s != N.OK:
            raise LairError(s, N.last_error(ctx.handle))

    def trace_kernel_source(self) -> str:
        n = N.lib.lurkhip_trace_source(self.toplevel.handle, self.func_idx, None, 0)
        if n < 0:
            raise LairError(n, N.lair_last_error() if hasattr(N, "lair_last_error") else "trace source generation failed")
        buf = C.create_string_buffer(n + 1)
        N.lib.lurkhip_trace_source(self.toplevel.handle, self.func_idx, buf, n + 1)
        return buf.value.decode()

    def trace_shape(self, shard: Shard):
        n, h, w = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(N.lib.lurkhip_func_trace_shape(shard.queries.handle, self.func_idx, shard.index, shard.shard_config.max_shard_size, C.byref(n), C.byref(h), C.byref(w)))
        return n.value, h.value, w.value

    def generate_trace(self, shard: Shard, repr: int = N.REPR_CANONICAL) -> np.ndarray:
        _, h, w = self.trace_shape(shard)
        out = np.empty((h, w), dtype=np.uint32)
        s = N.lib.lurkhip_generate_trace_func(self.ctx.handle, self.toplevel.handle, shard.queries.handle, self.func_idx, shard.index, shard.shard_config.max_shard_size, _addr(out), repr)
        if s != N.OK:
            raise LairError(s, N.last_error(self.ctx.handle))
        return out

    def generate_trace_dev(self, shard: Shard, out_dev, repr: int = N.REPR_CANONICAL):
        s = N.lib.lurkhip_generate_trace_func_dev(self.ctx.handle, self.toplevel.handle, shard.queries.handle, self.func_idx, shard.index, shard.shard_config.max_shard_size, _addr(out_dev), repr)
        if s != N.OK:
            raise LairError(s, N.last_error(self.ctx.handle))


def _row_pitch(buf) -> int:
    """Words between the rows of a 2-D device tensor (a column range of an aligned group buffer has stride(0) > its width);
    0 = dense, for anything else."""
    if hasattr(buf, "stride") and hasattr(buf, "dim") and buf.dim() == 2:
        assert buf.stride(1) == 1, "trace matrices are row-major"
        return int(buf.stride(0))
    return 0


class PreparedFuncTrace:
    """Device-resident inputs of one chip's trace (FuncChip: program + per-row arrays + row stream; MemChip: values +
    provide records; BytesChip: the 65536 x 6 lookup records)."""

    def __init__(self, chip, shard: Shard, handle=None):
        self.ctx = chip.ctx
        h = C.c_void_p()
        if handle is not None:  # flattened by prepare_many
            h, s = handle, N.OK
        elif isinstance(chip, FuncChip):
            s = N.lib.lurkhip_func_trace_prepare(self.ctx.handle, chip.toplevel.handle, shard.queries.handle, chip.func_idx, shard.index, shard.shard_config.max_shard_size, C.byref(h))
        elif isinstance(chip, MemChip):
            s = N.lib.lurkhip_mem_trace_prepare(self.ctx.handle, shard.queries.handle, chip.len, C.byref(h))
        else:
            s = N.lib.lurkhip_bytes_trace_prepare(self.ctx.handle, shard.queries.handle, shard.index, C.byref(h))
        if s != N.OK:
            raise LairError(s, N.last_error(self.ctx.handle))
        self.handle = h
        shape = (C.c_uint64 * 5)()
        _check(N.lib.lurkhip_func_trace_shape_of(h, shape))
        self.n_real, self.height, self.width, self.input_bytes, self.stream_words = [int(x) for x in shape]

    @classmethod
    def prepare_many(cls, chips, shard: Shard, n_threads: int = 0):
        """The FuncChips' inputs of one shard flattened together on host threads into page-locked staging and queued for
        upload on the chips' context (lurkhip_func_trace_prepare_many): [PreparedFuncTrace or None (no rows in the shard)]."""
        if not chips:
            return []
        ctx, top = chips[0].ctx, chips[0].toplevel
        idx = (C.c_int32 * len(chips))(*[c.func_idx for c in chips])
        out = (C.c_void_p * len(chips))()
        s = N.lib.lurkhip_func_trace_prepare_many(ctx.handle, top.handle, shard.queries.handle, len(chips), idx, shard.index,
                                                  shard.shard_config.max_shard_size, n_threads, out)
        if s != N.OK:
            raise LairError(s, N.last_error(ctx.handle))
        return [cls(c, shard, handle=C.c_void_p(h)) if h else None for c, h in zip(chips, out)]

    def export(self) -> np.ndarray:
        """The prepared inputs as bytes (lurkhip_func_trace_export): what a process that did not execute the program needs to
        generate this chip's trace (`from_blob`)."""
        size = C.c_uint64()
        _check(N.lib.lurkhip_func_trace_export_size(self.handle, C.byref(size)))
        blob = np.empty(int(size.value), dtype=np.uint8)
        self.ctx.check(N.lib.lurkhip_func_trace_export(self.ctx.handle, self.handle, _addr(blob), int(size.value)))
        return blob

    @classmethod
    def from_blob(cls, ctx, blob: np.ndarray) -> "PreparedFuncTrace":
        """A prepared trace on `ctx` from `export`'s bytes (lurkhip_func_trace_import)."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        h = C.c_void_p()
        ctx.check(N.lib.lurkhip_func_trace_import(ctx.handle, _addr(blob), blob.nbytes, C.byref(h)))
        self = cls.__new__(cls)
        self.ctx, self.handle = ctx, h
        shape = (C.c_uint64 * 5)()
        _check(N.lib.lurkhip_func_trace_shape_of(h, shape))
        self.n_real, self.height, self.width, self.input_bytes, self.stream_words = [int(x) for x in shape]
        return self

    def run(self, out_dev, repr: int = N.REPR_CANONICAL, ctx=None):
        """Launches the trace kernel on `ctx` (default: the context the inputs were uploaded on; another context's stream must
        only be used once that upload has completed)."""
        ctx = ctx or self.ctx
        ctx.check(N.lib.lurkhip_func_trace_run_pitched(ctx.handle, self.handle, _addr(out_dev), _row_pitch(out_dev), repr))

    def run_rows(self, first_row: int, n_rows: int, out_dev, repr: int = N.REPR_CANONICAL, ctx=None):
        """Rows [first_row, first_row + n_rows) of the trace only, into out_dev[n_rows][..]: a rank's block when several ranks prove
        one shard together (lurkhip_func_trace_run_rows)."""
        ctx = ctx or self.ctx
        ctx.check(N.lib.lurkhip_func_trace_run_rows(ctx.handle, self.handle, first_row, n_rows, _addr(out_dev), _row_pitch(out_dev), repr))

    def close(self):
        if self.handle:
            N.lib.lurkhip_func_trace_free(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MemChip:  # memory.rs:18-69
    def __init__(self, ctx: Context, mem_len: int):
        assert mem_len in MEM_TABLE_SIZES
        self.ctx = ctx
        self.len = mem_len

    def width(self) -> int:
        return 4 + self.len

    def generate_trace(self, shard: Shard, repr: int = N.REPR_CANONICAL) -> np.ndarray:
        n, h, w = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(N.lib.lurkhip_mem_trace_shape(shard.queries.handle, self.len, C.byref(n), C.byref(h), C.byref(w)))
        out = np.empty((h.value, w.value), dtype=np.uint32)
        s = N.lib.lurkhip_generate_trace_mem(self.ctx.handle, shard.queries.handle, self.len, _addr(out), repr)
        if s != N.OK:
            raise LairError(s, N.last_error(self.ctx.handle))
        return out


class BytesChip:  # gadgets/bytes/trace.rs
    HEIGHT = 1 << 16

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def width(self) -> int:
        return 13

    def preprocessed_width(self) -> int:
        return 6

    def generate_trace(self, shard: Shard, repr: int = N.REPR_CANONICAL) -> np.ndarray:
        out = np.empty((self.HEIGHT, 13), dtype=np.uint32)
        s = N.lib.lurkhip_generate_trace_bytes(self.ctx.handle, shard.queries.handle, shard.index, _addr(out), repr)
        if s != N.OK:
            raise LairError(s, N.last_error(self.ctx.handle))
        return out

    def generate_preprocessed_trace(self, repr: int = N.REPR_CANONICAL) -> np.ndarray:
        d = self.ctx.malloc(self.HEIGHT * 6 * 4)
        try:
            self.ctx.check(N.lib.lurkhip_trace_bytes_preprocessed_dev(self.ctx.handle, C.c_void_p(d), repr))
            out = np.empty((self.HEIGHT, 6), dtype=np.uint32)
            self.ctx.d2h(out, d)
        finally:
            self.ctx.free(d)
        return out


def entrypoint_trace(queries: QueryRecord) -> np.ndarray:
    """LairChip::Entrypoint trace: one row holding the public values (lair_chip.rs:112-118)."""
    return np.array([queries.expect_public_values()], dtype=np.uint32)
