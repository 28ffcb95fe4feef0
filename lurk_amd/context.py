"""`Context`: one HIP stream + scratch arenas behind the lurkhip C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N


def _addr(buf) -> int:
    """Raw address of a numpy array (host) or a torch tensor (device)."""
    if isinstance(buf, np.ndarray):
        return buf.ctypes.data
    if hasattr(buf, "data_ptr"):
        return buf.data_ptr()
    if isinstance(buf, int):
        return buf
    raise TypeError(f"cannot take the address of {type(buf)!r}")


def as_u32(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a


class Context:
    """Owns a ``lurkhip_ctx``.  ``stream`` may be a raw ``hipStream_t`` value (for
    example ``torch.cuda.current_stream().cuda_stream``) to enqueue on a caller stream; ``priority`` (0 default, > 0 lower) is
    that of the context's own streams."""

    def __init__(self, device: int = 0, stream: int | None = None, priority: int = 0, beside: "Context | None" = None):
        h = C.c_void_p()
        if beside is not None:
            # a context whose stream was MEASURED to run beside `beside`'s (lurkhip_ctx_create_beside): the second proof in flight
            N.check(N.lib.lurkhip_ctx_create_beside(beside.handle, C.byref(h)))
            device = beside.device
        elif stream is None and priority:
            N.check(N.lib.lurkhip_ctx_create_with_priority(device, priority, C.byref(h)))
        elif stream is None:
            N.check(N.lib.lurkhip_ctx_create(device, C.byref(h)))
        else:
            N.check(N.lib.lurkhip_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h)))
        self.handle = h
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            N.lib.lurkhip_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def check(self, status: int):
        N.check(status, self.handle)

    def overlap_probe(self, other: "Context"):
        """(alone_s, both_s): a short busy kernel alone on this context's stream, then one on each context's stream together --
        about equal when the streams sit on different hardware queues, both_s about twice alone_s when they share one."""
        a, b = C.c_double(), C.c_double()
        self.check(N.lib.lurkhip_ctx_overlap_probe(self.handle, other.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def sync(self):
        self.check(N.lib.lurkhip_ctx_sync(self.handle))

    def record_event(self, event=None):
        """An event behind everything queued on this context's stream (created when `event` is None); returns the handle."""
        e = C.c_void_p(event)
        self.check(N.lib.lurkhip_event_record(self.handle, C.byref(e)))
        return e.value

    def wait_event(self, event):
        """Everything queued on this context's stream from now on runs after `event` (no host wait)."""
        self.check(N.lib.lurkhip_event_wait(self.handle, C.c_void_p(event)))

    @staticmethod
    def destroy_event(event):
        N.lib.lurkhip_event_destroy(C.c_void_p(event))

    def timer_start(self):
        self.check(N.lib.lurkhip_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self.check(N.lib.lurkhip_timer_stop(self.handle, C.byref(ms)))
        return float(ms.value)

    def profile_enable(self, on=True):
        """on: False / True (the stage spans) / 2 (also the per-chip detail spans)."""
        self.check(N.lib.lurkhip_profile_enable(self.handle, int(on)))

    def span_begin(self, name: str):
        self.check(N.lib.lurkhip_profile_span_begin(self.handle, name.encode()))

    def span_end(self, name: str):
        self.check(N.lib.lurkhip_profile_span_end(self.handle, name.encode()))

    def profile_reset(self):
        self.check(N.lib.lurkhip_profile_reset(self.handle))

    def profile_read(self, span: str):
        """(total_ms, count) of a named library span since the last reset."""
        ms = C.c_double()
        cnt = C.c_int64()
        self.check(N.lib.lurkhip_profile_read(self.handle, span.encode(), C.byref(ms), C.byref(cnt)))
        return float(ms.value), int(cnt.value)

    def pool_trim(self):
        self.check(N.lib.lurkhip_pool_trim(self.handle))

    def pool_stats(self) -> dict:
        """Allocator accounting of this context (bytes): live, cached, peak of live + cached, hipMalloc calls, OOM retries."""
        out = np.zeros(6, dtype=np.uint64)
        self.check(N.lib.lurkhip_pool_stats(self.handle, out.ctypes.data))
        keys = ("live_bytes", "cached_bytes", "peak_bytes", "mallocs", "oom_retries", "scale_table_bytes")
        return {k: int(v) for k, v in zip(keys, out)}

    def pool_reset_peak(self):
        self.check(N.lib.lurkhip_pool_reset_peak(self.handle))

    def debug_inject_alloc_failures(self, n: int):
        self.check(N.lib.lurkhip_debug_inject_alloc_failures(self.handle, int(n)))

    # raw device memory (hosts without torch)
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self.check(N.lib.lurkhip_malloc(self.handle, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr: int):
        self.check(N.lib.lurkhip_free(self.handle, C.c_void_p(ptr)))

    def h2d(self, dev_ptr: int, host: np.ndarray):
        host = np.ascontiguousarray(host)
        self.check(N.lib.lurkhip_memcpy_h2d(self.handle, C.c_void_p(dev_ptr), host.ctypes.data, host.nbytes))

    def d2h(self, host: np.ndarray, dev_ptr: int):
        assert host.flags["C_CONTIGUOUS"]
        self.check(N.lib.lurkhip_memcpy_d2h(self.handle, host.ctypes.data, C.c_void_p(dev_ptr), host.nbytes))
