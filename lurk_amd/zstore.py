"""Content-addressed interning of Lurk data (the host caller of the hash kernels).

Mirrors the subset of /root/reference/src/core/zstore.rs that turns Lurk data into
Poseidon2 preimages: ZPtr flattening (zstore.rs:184-204), hash3/4/5 memo tables
(zstore.rs:305-333), tuple interning (zstore.rs:335-349) and string / symbol / list /
fun / env interning (zstore.rs:397-511).  Tags: /root/reference/src/core/tag.rs:23-39.

The store is hasher-agnostic: it calls ``hasher.hash(preimage)`` with 24/32/40-lane
preimages, which the product wires to ``lurk_amd.poseidon.Hasher`` (HIP kernels).
"""
from __future__ import annotations

from dataclasses import dataclass

DIGEST_SIZE = 8

# tag.rs:23-39
TAGS = ["U64", "Num", "BigNum", "Comm", "Char", "Str", "Key", "Fun", "Builtin", "Coroutine", "Sym", "Cons", "Env", "Fix", "Err"]
TAG = {name: i for i, name in enumerate(TAGS)}

# state.rs:258-268 / zstore.rs builtin_set()
LURK_PACKAGE = "lurk"
BUILTIN_PACKAGE = "builtin"
USER_PACKAGE = "lurk-user"


@dataclass(frozen=True)
class ZPtr:
    tag: int
    digest: tuple

    def flatten(self) -> list[int]:
        return [self.tag] + [0] * 7 + list(self.digest)


def _digest_from_field(f: int) -> tuple:
    return (f,) + (0,) * 7


class ZStore:
    def __init__(self, hasher):
        self.hasher = hasher
        self.hashes = {}  # preimage tuple -> digest tuple (hashes3/4/5 merged; lengths differ)
        self.str_cache = {}
        self.sym_cache = {}
        self.nil = self.intern_symbol([LURK_PACKAGE, "nil"])
        self.t = self.intern_symbol([LURK_PACKAGE, "t"])

    # --- hashing with memoisation (zstore.rs:305-333)
    def hash(self, preimg: list[int]) -> tuple:
        key = tuple(preimg)
        d = self.hashes.get(key)
        if d is None:
            d = tuple(int(x) for x in self.hasher.hash(list(preimg)))
            self.hashes[key] = d
        return d

    def hash_many(self, preimgs) -> list[tuple]:
        """Memoised digests of many preimages; the misses go to the hasher in one batch per width (SURVEY.md 8f.1:
        level-order hashing of independent nodes instead of one kernel round trip per node)."""
        keys = [tuple(p) for p in preimgs]
        missing = [k for k in dict.fromkeys(keys) if k not in self.hashes]
        if missing:
            if hasattr(self.hasher, "hash_many"):
                digests = self.hasher.hash_many([list(k) for k in missing])
            else:
                digests = [self.hasher.hash(list(k)) for k in missing]
            for k, d in zip(missing, digests):
                self.hashes[k] = tuple(int(x) for x in d)
        return [self.hashes[k] for k in keys]

    def intern_strings(self, strings) -> list[ZPtr]:
        """intern_string for many strings at once, level by level: strings are right-nested (char, tail) pairs
        (zstore.rs:397-413), so the tails of length k of all strings form one independent batch."""
        strings = list(strings)
        todo = [s for s in dict.fromkeys(strings) if s not in self.str_cache]
        # suffixes by length, shortest first; the empty suffix is the null string pointer
        known = {"": self.null(TAG["Str"])}
        known.update(self.str_cache)
        level = 1
        while True:
            batch = sorted({s[len(s) - level:] for s in todo if len(s) >= level} - set(known))
            if not batch and all(len(s) < level for s in todo):
                break
            digests = self.hash_many([self.char(suf[0]).flatten() + known[suf[1:]].flatten() for suf in batch])
            for suf, d in zip(batch, digests):
                known[suf] = ZPtr(TAG["Str"], d)
            level += 1
        for s in todo:
            self.str_cache[s] = known[s]
        return [self.str_cache[s] for s in strings]

    def intern_tuple11(self, tag: int, a: ZPtr, b: ZPtr) -> ZPtr:
        return ZPtr(tag, self.hash(a.flatten() + b.flatten()))

    def intern_tuple110(self, tag: int, a: ZPtr, b: ZPtr, c: ZPtr) -> ZPtr:
        return ZPtr(tag, self.hash(a.flatten() + b.flatten() + list(c.digest)))

    # --- atoms (zstore.rs:93-160)
    @staticmethod
    def null(tag: int) -> ZPtr:
        return ZPtr(tag, (0,) * 8)

    @staticmethod
    def num(f: int) -> ZPtr:
        return ZPtr(TAG["Num"], _digest_from_field(f))

    @staticmethod
    def u64(u: int) -> ZPtr:
        return ZPtr(TAG["U64"], tuple((u >> (8 * i)) & 0xFF for i in range(8)))

    @staticmethod
    def char(c: str) -> ZPtr:
        b = c.encode("utf-8")
        return ZPtr(TAG["Char"], tuple(b) + (0,) * (8 - len(b)))

    @staticmethod
    def big_num(digest) -> ZPtr:
        return ZPtr(TAG["BigNum"], tuple(digest))

    @staticmethod
    def comm(digest) -> ZPtr:
        return ZPtr(TAG["Comm"], tuple(digest))

    # --- compound data (zstore.rs:397-511)
    def intern_string(self, s: str) -> ZPtr:
        z = self.str_cache.get(s)
        if z is None:
            z = self.null(TAG["Str"])
            for c in reversed(s):
                z = self.intern_tuple11(TAG["Str"], self.char(c), z)
            self.str_cache[s] = z
        return z

    def intern_symbol(self, path: list[str], *, keyword: bool = False, builtin: bool = False, coroutine: bool = False) -> ZPtr:
        key = (tuple(path), keyword, builtin, coroutine)
        z = self.sym_cache.get(key)
        if z is not None:
            return z
        if not path:
            z = self.null(TAG["Key"] if keyword else TAG["Sym"])
        else:
            z = self.null(TAG["Sym"])
            for i, s in enumerate(path):
                last = i == len(path) - 1
                if last:
                    tag = TAG["Builtin"] if builtin else TAG["Coroutine"] if coroutine else TAG["Key"] if keyword else TAG["Sym"]
                else:
                    tag = TAG["Sym"]
                z = self.intern_tuple11(tag, self.intern_string(s), z)
        self.sym_cache[key] = z
        return z

    def user_sym(self, name: str) -> ZPtr:
        return self.intern_symbol([USER_PACKAGE, name])

    def builtin_sym(self, name: str) -> ZPtr:
        return self.intern_symbol([LURK_PACKAGE, BUILTIN_PACKAGE, name], builtin=True)

    def intern_cons(self, car: ZPtr, cdr: ZPtr) -> ZPtr:
        return self.intern_tuple11(TAG["Cons"], car, cdr)

    def intern_list(self, xs, tail: ZPtr | None = None) -> ZPtr:
        z = self.nil if tail is None else tail
        for x in reversed(list(xs)):
            z = self.intern_cons(x, z)
        return z

    def intern_empty_env(self) -> ZPtr:
        return self.null(TAG["Env"])

    def intern_fun(self, args: ZPtr, body: ZPtr, env: ZPtr) -> ZPtr:
        return self.intern_tuple110(TAG["Fun"], args, body, env)

    def intern_env(self, sym: ZPtr, val: ZPtr, env: ZPtr) -> ZPtr:
        return self.intern_tuple110(TAG["Env"], sym, val, env)

    # --- commitments: hash3(secret || flatten(payload)) (core/eval_direct.rs `hide`/`commit`)
    def hide(self, secret_digest, payload: ZPtr) -> ZPtr:
        return self.comm(self.hash(list(secret_digest) + payload.flatten()))

    def commit(self, payload: ZPtr) -> ZPtr:
        return self.hide((0,) * 8, payload)


# ------------------------------------------------------------------ level-order batched interning on the device
# Syntax of Lurk data as the reference's parser hands it to intern_syntax (zstore.rs:513-549): plain tuples,
#   ("num", f) ("u64", u) ("char", c) ("bignum", digest) ("comm", digest) ("str", s)
#   ("sym", path, flags)   flags in {"", "keyword", "builtin", "coroutine"}
#   ("list", [xs]) ("improper", [xs], y) ("quote", x)
def syn_num(f): return ("num", int(f))
def syn_u64(u): return ("u64", int(u))
def syn_char(c): return ("char", c)
def syn_str(s): return ("str", s)
def syn_sym(*path, flags=""): return ("sym", tuple(path), flags)
def syn_user(name): return ("sym", (USER_PACKAGE, name), "")
def syn_builtin(name): return ("sym", (LURK_PACKAGE, BUILTIN_PACKAGE, name), "builtin")
def syn_list(*xs): return ("list", tuple(xs))
def syn_improper(xs, y): return ("improper", tuple(xs), y)
def syn_quote(x): return ("quote", x)


class BatchedZStore:
    """ZStore whose hashing runs level by level on the device through the native store of the C ABI
    (lurkhip_zstore_*, lurk_amd/csrc/zstore.cpp): `Batch` collects pending nodes -- whole syntax trees, lists, environments,
    closures, commitments -- and `Batch.run()` interns them with one hash launch per DAG height and preimage width instead of
    one Poseidon2 call per node (zstore.rs:305-349 hashes node by node).  Also the Merkle-DAG side: memoize_dag
    (zstore.rs:569-702), fetch_tuple11 / fetch_tuple110 (zstore.rs:705-718) and the ZDag export of a cached proof."""

    def __init__(self, ctx):
        import ctypes as C

        from . import _native as N

        self._N, self._C, self.ctx = N, C, ctx
        h = C.c_void_p()
        ctx.check(N.lib.lurkhip_zstore_new(ctx.handle, C.byref(h)))
        self.handle = h
        b = self.batch()
        nil, t, quote = b.symbol((LURK_PACKAGE, "nil")), b.symbol((LURK_PACKAGE, "t")), b.symbol((LURK_PACKAGE, BUILTIN_PACKAGE, "quote"), "builtin")
        b.run()
        self.nil, self.t, self.quote = b[nil], b[t], b[quote]

    def close(self):
        if getattr(self, "handle", None):
            self._N.lib.lurkhip_zstore_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != self._N.OK:
            raise RuntimeError(f"lurkhip zstore status {st}: {self._N.lib.lurkhip_zstore_last_error(self.handle).decode()}")

    def batch(self) -> "Batch":
        return Batch(self)

    def intern_syntax_many(self, syntaxes) -> list[ZPtr]:
        b = self.batch()
        ids = [b.syntax(s) for s in syntaxes]
        b.run()
        return [b[i] for i in ids]

    def stats(self) -> dict:
        import numpy as np

        out = np.zeros(6, dtype=np.uint64)
        self._check(self._N.lib.lurkhip_zstore_stats(self.handle, out.ctypes.data_as(self._C.c_void_p)))
        return dict(zip(("hash3", "hash4", "hash5", "launches", "memo_hits", "dag_entries"), (int(x) for x in out)))

    # --- Merkle DAG
    def set_inverse_tables(self, hashes4_inv: dict, hashes5_inv: dict):
        """digest tuple -> preimage (32 / 40 lanes): `record.get_inv_queries("hash4" / "hash5")` in the reference."""
        import numpy as np

        def rows(d, n):
            a = np.zeros((len(d), 8 + n), dtype=np.uint32)
            for i, (dg, pre) in enumerate(d.items()):
                a[i, :8], a[i, 8:] = dg, pre
            return a

        a4, a5 = rows(hashes4_inv, 32), rows(hashes5_inv, 40)
        p = self._C.c_void_p
        self._check(self._N.lib.lurkhip_zstore_set_inverse_tables(self.handle, len(a4), a4.ctypes.data_as(p), len(a5), a5.ctypes.data_as(p)))

    def memoize_dag(self, tag: int, digest):
        import numpy as np

        d = np.asarray(list(digest), dtype=np.uint32)
        self._check(self._N.lib.lurkhip_zstore_memoize_dag(self.handle, tag, d.ctypes.data_as(self._C.c_void_p)))

    def _fetch(self, z: ZPtr):
        import numpy as np

        zp = np.asarray([z.tag] + list(z.digest), dtype=np.uint32)
        out = np.zeros(28, dtype=np.uint32)
        p = self._C.c_void_p
        if self._N.lib.lurkhip_zstore_fetch(self.handle, zp.ctypes.data_as(p), out.ctypes.data_as(p)) != self._N.OK:
            raise KeyError("Data missing from ZStore's DAG")
        kids = [ZPtr(int(out[1 + 9 * k]), tuple(int(x) for x in out[2 + 9 * k:10 + 9 * k])) for k in range(3)]
        return int(out[0]), kids

    def fetch_tuple11(self, z: ZPtr):
        kind, kids = self._fetch(z)
        if kind != 1:
            raise KeyError("Tuple11 data not found on DAG")
        return kids[0], kids[1]

    def fetch_tuple110(self, z: ZPtr):
        kind, kids = self._fetch(z)
        if kind != 2:
            raise KeyError("Tuple110 data not found on DAG")
        return tuple(kids)

    def dag_export(self, roots):
        """ZDag::populate_with_many (cli/zdag.rs:16-55): [(zptr, kind, children)] reachable from `roots`, children first."""
        import numpy as np

        r = np.asarray([[z.tag] + list(z.digest) for z in roots], dtype=np.uint32).reshape(-1, 9)
        p = self._C.c_void_p
        n = self._N.lib.lurkhip_zstore_dag_export(self.handle, len(r), r.ctypes.data_as(p), None, 0)
        if n < 0:
            raise KeyError("Data missing from ZStore's DAG")
        out = np.zeros((max(n, 1), 37), dtype=np.uint32)
        assert self._N.lib.lurkhip_zstore_dag_export(self.handle, len(r), r.ctypes.data_as(p), out.ctypes.data_as(p), n) == n

        def zp(w):
            return ZPtr(int(w[0]), tuple(int(x) for x in w[1:9]))

        return [(zp(e[:9]), int(e[9]), [zp(e[10 + 9 * k:19 + 9 * k]) for k in range({0: 0, 1: 2, 2: 3}[int(e[9])])]) for e in out[:n]]


class Batch:
    """Pending nodes of one level-order interning pass; methods return node handles, `run()` hashes, `batch[handle]` is the
    ZPtr afterwards.  Same construction rules as the one-at-a-time ZStore above (zstore.rs:397-511)."""

    def __init__(self, store: BatchedZStore):
        self.store = store
        self.nodes = []      # (kind, tag, payload)
        self.cache = {}      # structural key -> handle
        self.zptrs = None

    def _add(self, key, kind, tag, payload):
        h = self.cache.get(key)
        if h is None:
            h = len(self.nodes)
            self.nodes.append((kind, tag, payload))
            self.cache[key] = h
        return h

    # atoms
    def atom(self, z: ZPtr): return self._add(("atom", z), 0, z.tag, z.digest)
    def ref(self, z: ZPtr): return self._add(("ref", z), 4, z.tag, z.digest)
    def null(self, tag): return self.atom(ZStore.null(tag))
    def num(self, f): return self.atom(ZStore.num(f))
    def u64(self, u): return self.atom(ZStore.u64(u))
    def char(self, c): return self.atom(ZStore.char(c))
    def big_num(self, d): return self.atom(ZStore.big_num(d))
    def comm(self, d): return self.atom(ZStore.comm(d))

    # compound
    def tuple11(self, tag, a, b): return self._add((1, tag, a, b), 1, tag, (a, b))
    def tuple110(self, tag, a, b, c): return self._add((2, tag, a, b, c), 2, tag, (a, b, c))

    def string(self, s: str):
        z = self.null(TAG["Str"])
        for c in reversed(s):
            z = self.tuple11(TAG["Str"], self.char(c), z)
        return z

    def symbol(self, path, flags=""):
        if not path:
            return self.null(TAG["Key"] if flags == "keyword" else TAG["Sym"])
        z = self.null(TAG["Sym"])
        for i, s in enumerate(path):
            last = i == len(path) - 1
            tag = TAG[{"builtin": "Builtin", "coroutine": "Coroutine", "keyword": "Key"}.get(flags, "Sym")] if last else TAG["Sym"]
            z = self.tuple11(tag, self.string(s), z)
        return z

    def cons(self, a, b): return self.tuple11(TAG["Cons"], a, b)

    def list(self, xs, tail=None):
        z = self.ref(self.store.nil) if tail is None else tail
        for x in reversed(list(xs)):
            z = self.cons(x, z)
        return z

    def quoted(self, x): return self.list([self.ref(self.store.quote), x])
    def empty_env(self): return self.null(TAG["Env"])
    def fun(self, args, body, env): return self.tuple110(TAG["Fun"], args, body, env)
    def fix(self, body, binds, env): return self.tuple110(TAG["Fix"], body, binds, env)
    def env(self, sym, val, env): return self.tuple110(TAG["Env"], sym, val, env)
    def hide(self, secret, payload): return self._add((3, secret, payload), 3, TAG["Comm"], (secret, payload))
    def commit(self, payload): return self.hide(self.ref(ZStore.big_num((0,) * 8)), payload)

    def syntax(self, syn):
        k = syn[0]
        if k == "num": return self.num(syn[1] % 2013265921)
        if k == "u64": return self.u64(syn[1])
        if k == "char": return self.char(syn[1])
        if k == "bignum": return self.big_num(syn[1])
        if k == "comm": return self.comm(syn[1])
        if k == "str": return self.string(syn[1])
        if k == "sym": return self.symbol(syn[1], syn[2])
        if k == "list": return self.list([self.syntax(x) for x in syn[1]])
        if k == "improper": return self.list([self.syntax(x) for x in syn[1]], self.syntax(syn[2]))
        if k == "quote": return self.quoted(self.syntax(syn[1]))
        raise ValueError(f"unsupported syntax {k}")

    def run(self):
        import numpy as np

        st = self.store
        n = len(self.nodes)
        arr = np.zeros((max(n, 1), 10), dtype=np.uint32)
        for i, (kind, tag, payload) in enumerate(self.nodes):
            arr[i, 0], arr[i, 1] = kind, tag
            arr[i, 2:2 + len(payload)] = payload
        out = np.zeros((max(n, 1), 9), dtype=np.uint32)
        p = st._C.c_void_p
        st._check(st._N.lib.lurkhip_zstore_intern_dag(st.handle, n, arr.ctypes.data_as(p), out.ctypes.data_as(p)))
        self.zptrs = [ZPtr(int(r[0]), tuple(int(x) for x in r[1:])) for r in out[:n]]
        return self.zptrs

    def __getitem__(self, handle) -> ZPtr:
        if self.zptrs is None:
            raise RuntimeError("Batch.run() first")
        return self.zptrs[handle]
