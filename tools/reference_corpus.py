#!/usr/bin/env python3
"""The reference's own evaluator test corpus through this repo's host side (build container only).

/root/reference/src/core/tests/eval_direct.rs holds ~190 cases `test!(name, "lurk code", |z| expected)` (also `test_raw!` with a
closure that builds the input and `test_env!` with one that builds the environment); `run_tests`
(/root/reference/src/core/tests/mod.rs:28-74) evaluates the input under `lurk_main` and compares the result with the expected ZPtr.
This module reads that file WHERE IT LIES at run time -- nothing of it is stored in the repo --, takes the cases apart, and turns
each expected-value closure (a few lines of Rust over ZStore constructors) into Python over this repo's ZStore mirror
(lurk_amd/zstore.py: the same interning, Poseidon2 through the product's host hasher).  tests/test_reference_corpus.py runs every
case through the product's Lair interpreter on the reference's 39 functions (tools/measure_lurk_shape.py: RealLurk) and holds the
results against the expected values: the evaluator, the interpreter, the reader and the store pinned by upstream's own vectors.

A closure this translator cannot express is reported as skipped, with the reason; it is never guessed.
"""
import os
import ast
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import lurk_reference as lr  # noqa: E402

CORPUS = "src/core/tests/eval_direct.rs"


# ------------------------------------------------------------------ taking the file apart
def _strip_comments(src: str) -> str:
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c == '"':  # string literal
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1])
            i = j + 1
        elif src.startswith("//", i):
            while i < n and src[i] != "\n":
                i += 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def _balanced(src: str, start: int, open_ch: str, close_ch: str) -> int:
    """index just past the bracket that closes the one at `start`"""
    depth, i = 0, start
    while True:
        c = src[i]
        if c == '"':
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
        elif c == "'" and re.match(r"'(\\.|[^'\\])'", src[i:]):
            i += len(re.match(r"'(\\.|[^'\\])'", src[i:]).group(0)) - 1
        elif c == open_ch:
            depth += 1
        elif c == close_ch:
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1


def _split_top(src: str, sep: str):
    parts, depth, i, last = [], 0, 0, 0
    while i < len(src):
        c = src[i]
        if c == '"':
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
        elif c == "'" and re.match(r"'(\\.|[^'\\])'", src[i:]):
            i += len(re.match(r"'(\\.|[^'\\])'", src[i:]).group(0)) - 1
        elif c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        elif c == sep and depth == 0:
            parts.append(src[last:i])
            last = i + 1
        i += 1
    parts.append(src[last:])
    return [p.strip() for p in parts]


def _unquote_rust(s: str) -> str:
    assert s[0] == '"' and s[-1] == '"', s[:40]
    body = s[1:-1]
    body = re.sub(r"\\\n\s*", "", body)  # line continuation
    return bytes(body, "utf-8").decode("unicode_escape").encode("latin-1").decode("utf-8")


def cases():
    """[(name, macro, args)] with args the macro's arguments after the name (source text), and the helper functions of the file."""
    src = _strip_comments(lr._read(CORPUS))
    helpers = {}
    for m in re.finditer(r"\bfn (\w+)\((\w+): &mut ZStore<F, LurkChip>\) -> ZPtr<F> \{", src):
        end = _balanced(src, m.end() - 1, "{", "}")
        helpers[m.group(1)] = (m.group(2), src[m.end() - 1:end])
    out = []
    for m in re.finditer(r"^(test|test_raw|test_env)!\(", src, re.M):
        end = _balanced(src, m.end() - 1, "(", ")")
        args = _split_top(src[m.end():end - 1], ",")
        args = [a for a in args if a]
        out.append((args[0], m.group(1), args[1:]))
    return out, helpers


# ------------------------------------------------------------------ the closures in Python
class RList(list):
    def try_into(self):
        return self

    def unwrap(self):
        return self


class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def namespace(z, hasher):
    """The names a closure may use, over the repo's sequential ZStore `z`."""
    from lurk_amd import zstore as zs

    enums = lr.enums()
    P = 2013265921

    class ZP(zs.ZPtr):
        pass

    def wrap(p):
        return p

    def flat(p):
        return RList(p.flatten())

    zs.ZPtr.flatten_r = lambda self: RList(self.flatten())

    class ZPtrNS:
        @staticmethod
        def u64(u):
            return zs.ZStore.u64(int(u))

        @staticmethod
        def num(f):
            return zs.ZStore.num(int(f) % P)

        @staticmethod
        def char(c):
            return zs.ZStore.char(c)

        @staticmethod
        def err(e):
            return zs.ZPtr(zs.TAG["Err"], (int(e),) + (0,) * 7)

        @staticmethod
        def null(tag):
            return zs.ZStore.null(int(tag))

        @staticmethod
        def big_num(d):
            return zs.ZStore.big_num(tuple(int(x) for x in d))

        @staticmethod
        def comm(d):
            return zs.ZStore.comm(tuple(int(x) for x in d))

    class Z:
        def t(self):
            return z.t

        def nil(self):
            return z.nil

        def intern_string(self, s):
            return z.intern_string(s)

        def intern_char(self, c):
            return zs.ZStore.char(c)

        def intern_u64(self, u):
            return zs.ZStore.u64(int(u))

        def intern_symbol_no_lang(self, spec):
            kind, path = spec
            return z.intern_symbol(list(path), keyword=kind == "key", builtin=kind == "builtin")

        def intern_list(self, xs):
            return z.intern_list(list(xs))

        def intern_cons(self, a, b):
            return z.intern_cons(a, b)

        def intern_empty_env(self):
            return z.intern_empty_env()

        def intern_env(self, s, v, e):
            return z.intern_env(s, v, e)

        def intern_fun(self, a, b, e):
            return z.intern_fun(a, b, e)

        def intern_fix(self, a, b, e):
            return z.intern_tuple110(zs.TAG["Fix"], a, b, e)

    class Hasher:
        def hash(self, pre):
            return RList(z.hash([int(x) for x in pre]))

    tag_ns = _NS(**{k: v for k, v in enums["Tag"].items()})
    return {
        "ZPtr": ZPtrNS, "uint": ZPtrNS.u64, "EvalErr": _NS(**enums["EvalErr"]), "Tag": tag_ns,
        "F": _NS(one=lambda: 1, zero=lambda: 0, from_canonical_u8=int, from_canonical_u16=int, from_canonical_u32=int, from_canonical_u64=int,
                 from_canonical_usize=int),
        "Symbol": _NS(key=lambda path: ("key", tuple(path))),
        "user_sym": lambda name: ("sym", (zs.USER_PACKAGE, name)),
        "builtin_sym": lambda name: ("builtin", (zs.LURK_PACKAGE, zs.BUILTIN_PACKAGE, name)),
        "lurk_hasher": lambda: Hasher(), "Vec": _NS(with_capacity=lambda n: RList(), new=lambda: RList()),
        "RList": RList, "with_tag": lambda p, tag: zs.ZPtr(int(tag), p.digest), "_Z": Z,
    }


class Untranslatable(Exception):
    pass


def _expr(e: str) -> str:
    """one Rust expression of the corpus' subset as Python source"""
    toks = re.split(r'("(?:[^"\\]|\\.)*"|\'(?:\\.|[^\'\\])\')', e)  # keep string / char literals apart
    out = []
    for k, t in enumerate(toks):
        if k % 2:  # a literal
            if t[0] == "'":
                body = t[1:-1]
                out.append(repr(bytes(body, "utf-8").decode("unicode_escape")))
            else:
                out.append(repr(_unquote_rust(t)))
            continue
        t = t.replace("::", ".")
        t = re.sub(r"&mut\b", "", t)
        t = t.replace("&", "")
        t = re.sub(r"(?<![\w\)\]])\*(?=\s*[A-Za-z_])", "", t)  # deref
        t = re.sub(r"\[([^\[\];]+);\s*(\d+)\]", r"RList([\1] * \2)", t)  # [x; n]
        t = re.sub(r"\.flatten\(\)", ".flatten_r()", t)
        t = re.sub(r"\b(\d+)(u8|u16|u32|u64|usize)\b", r"\1", t)
        out.append(t)
    return "".join(out)


def to_python(closure: str, helpers, depth=0) -> str:
    """A Python function body `def f(z): ...` (source) for a Rust closure `|z| expr`, `|z| { stmts; expr }` or a helper's name."""
    closure = closure.strip()
    if re.fullmatch(r"\w+", closure):
        if closure not in helpers:
            raise Untranslatable(f"unknown helper {closure}")
        param, body = helpers[closure]
    else:
        m = re.match(r"\|\s*(\w+)\s*\|\s*", closure)
        if not m:
            raise Untranslatable("not a closure")
        param, body = m.group(1), closure[m.end():].strip()
    lines = []
    if body.startswith("{"):
        assert _balanced(body, 0, "{", "}") == len(body), body
        stmts = _split_top(body[1:-1], ";")
    else:
        stmts = [body]
    if not stmts[-1]:
        raise Untranslatable("block without a value")
    for s in stmts[:-1]:
        if not s:
            continue
        m = re.match(r"let\s+(?:mut\s+)?(\w+)(?:\s*:\s*[^=]+)?\s*=\s*(.+)$", s, re.S)
        if m:
            lines.append(f"{m.group(1)} = {_expr(m.group(2))}")
            continue
        m = re.match(r"assert_eq!\((.+)\)$", s, re.S)
        if m:
            a, b = _split_top(m.group(1), ",")[:2]
            lines.append(f"assert ({_expr(a)}) == ({_expr(b)})")
            continue
        m = re.match(r"(\w+)\.tag\s*=\s*(.+)$", s, re.S)
        if m:
            lines.append(f"{m.group(1)} = with_tag({m.group(1)}, {_expr(m.group(2))})")
            continue
        if re.match(r"\w+\.extend\(", s):
            lines.append(_expr(s))
            continue
        raise Untranslatable(f"statement: {s[:60]}")
    lines.append(f"return {_expr(stmts[-1])}")
    src = f"def _f({param if param != '_' else '_unused'}):\n" + "".join("    " + " ".join(ln.split()) + "\n" for ln in lines)
    return src


# What a translated closure may consist of.  The closures are text read from /root/reference -- designated untrusted -- and the
# translation is a handful of regular expressions: before anything is executed the result is parsed and every node checked against
# this list, every name against the namespace the closures are given (plus the function's own parameter and locals), attribute
# access against underscore names; it then runs without builtins (`|z| __import__('os').system(..)` must not run).
_ALLOWED_NODES = (ast.Module, ast.FunctionDef, ast.arguments, ast.arg, ast.Return, ast.Assign, ast.Assert, ast.Expr, ast.Name, ast.Load, ast.Store,
                  ast.Attribute, ast.Call, ast.keyword, ast.Constant, ast.List, ast.Tuple, ast.BinOp, ast.UnaryOp, ast.Compare, ast.BoolOp, ast.Subscript,
                  ast.Slice, ast.Starred, ast.IfExp, ast.Add, ast.Sub, ast.Mult, ast.FloorDiv, ast.Mod, ast.Pow, ast.LShift, ast.RShift, ast.BitOr,
                  ast.BitAnd, ast.BitXor, ast.USub, ast.Not, ast.Invert, ast.Eq, ast.NotEq, ast.Lt, ast.LtE, ast.Gt, ast.GtE, ast.And, ast.Or,
                  ast.ListComp, ast.GeneratorExp, ast.comprehension)


def check_closure_source(src: str, allowed_names):
    try:
        tree = ast.parse(src)
    except SyntaxError as e:
        raise Untranslatable(f"python syntax: {e.msg}: {src!r}") from e
    local = set()
    for node in ast.walk(tree):
        if not isinstance(node, _ALLOWED_NODES):
            raise Untranslatable(f"refused: {type(node).__name__} in a translated closure: {src!r}")
        if isinstance(node, ast.arg):
            local.add(node.arg)
        if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Store):
            local.add(node.id)
    for node in ast.walk(tree):
        if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Load) and node.id not in allowed_names and node.id not in local:
            raise Untranslatable(f"refused: unknown name {node.id!r} in a translated closure: {src!r}")
        if isinstance(node, (ast.Name, ast.arg)) and (node.id if isinstance(node, ast.Name) else node.arg).startswith("__"):
            raise Untranslatable(f"refused: dunder name in a translated closure: {src!r}")
        if isinstance(node, ast.Attribute) and node.attr.startswith("_"):
            raise Untranslatable(f"refused: attribute {node.attr!r} in a translated closure: {src!r}")
        if isinstance(node, ast.FunctionDef) and (node.name != "_f" or node.decorator_list):
            raise Untranslatable(f"refused: a definition other than the closure itself: {src!r}")
    return tree


def compile_closure(closure: str, helpers, z, hasher):
    ns = namespace(z, hasher)
    src = to_python(closure, helpers)
    tree = check_closure_source(src, set(ns))
    code = compile(tree, "<corpus closure>", "exec")
    ns["__builtins__"] = {}
    exec(code, ns)
    zw = ns["_Z"]()
    return lambda: ns["_f"](zw)


if __name__ == "__main__":
    cs, hs = cases()
    print(len(cs), "cases,", len(hs), "helpers:", sorted(hs))
    bad = 0
    for name, macro, args in cs:
        for a in (args[-1:] if macro == "test" else args[1:] if macro == "test_env" else args):
            try:
                to_python(a, hs)
            except Untranslatable as e:
                bad += 1
                print("  skip", name, "--", e)
    print(bad, "closures not translatable")
