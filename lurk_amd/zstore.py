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
