"""Run-time reader of the reference's Lurk evaluator (build container only).

The 39 Lair functions of the Lurk machine (/root/reference/src/core/eval_direct.rs:29-74) are Rust `func!` invocations whose
bodies are the surface syntax this repo's front ends already parse (lurk_amd/csrc/lair/parse.cpp), with a few Rust-side
expressions inside: `Tag::*` (tag.rs:23-39), `InternalTag::*` (ingress.rs:85-97), `EvalErr::*` (error.rs:7-30), pointers and
digests of the preallocated symbols (`digests.*_ptr(..)`, `digests.lurk_symbol_digest(..)`, ingress.rs:61-80 over
state.rs:258-318).  This module reads those sources WHERE THEY LIE at run time, resolves the Rust-side expressions to field
constants and returns the functions as text, in memory.  Nothing of it is written anywhere: the repo holds numbers measured
on the result (tests/golden/fib_shape.json), never the programs.

`available()` is False wherever /root/reference does not exist (the GPU box); callers skip.
"""
from __future__ import annotations

import os
import re

REF = os.environ.get("LURK_REFERENCE", "/root/reference")
CORE = os.path.join(REF, "src", "core")


def available() -> bool:
    return os.path.isfile(os.path.join(CORE, "eval_direct.rs"))


def _read(rel: str) -> str:
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def _strip_comments(src: str) -> str:
    return re.sub(r"//[^\n]*", "", src)


def _enum(src: str, name: str) -> dict[str, int]:
    """Variants of a fieldless Rust enum with optional `= n` discriminants."""
    m = re.search(r"pub enum " + name + r"\s*\{(.*?)\}", src, re.S)
    assert m, name
    out, nxt = {}, 0
    for item in _strip_comments(m.group(1)).split(","):
        item = item.strip()
        if not item:
            continue
        if "=" in item:
            k, v = [x.strip() for x in item.split("=")]
            nxt = int(v)
        else:
            k = item
        out[k] = nxt
        nxt += 1
    return out


def _str_array(src: str, name: str) -> list[str]:
    m = re.search(name + r": \[&str; (\d+)\] = \[(.*?)\];", src, re.S)
    assert m, name
    items = re.findall(r'"((?:[^"\\]|\\.)*)"', m.group(2))
    assert len(items) == int(m.group(1)), (name, len(items))
    return items


def enums() -> dict[str, dict[str, int]]:
    tag = _enum(_read("src/core/tag.rs"), "Tag")
    internal = {k: len(tag) + v for k, v in _enum(_read("src/core/ingress.rs"), "InternalTag").items()}  # ingress.rs:93-96
    err = _enum(_read("src/core/error.rs"), "EvalErr")
    return {"Tag": tag, "InternalTag": internal, "EvalErr": err}


def preallocated_symbols() -> list[tuple[str, str]]:
    """[(package kind, name)] in the order SymbolsDigests::new interns them (ingress.rs:36-58): pointer = index + 1."""
    st = _read("src/core/state.rs")
    return [("lurk", n) for n in _str_array(st, "LURK_SYMBOLS")] + [("builtin", n) for n in _str_array(st, "BUILTIN_SYMBOLS")]


def test_widths() -> dict[str, int]:
    """The 39 literals of `test_widths` (eval_direct.rs:2025-2063)."""
    src = _read("src/core/eval_direct.rs")
    out = dict((m.group(1), int(m.group(2))) for m in re.finditer(r'expect_eq\((\w+)\.width\(\), expect!\["(\d+)"\]\);', src))
    assert len(out) == 39, len(out)
    return out


def native_func_order() -> list[str]:
    """Function names in the order of `native_lurk_funcs` (eval_direct.rs:33-73) = chip order of the machine."""
    src = _read("src/core/eval_direct.rs")
    m = re.search(r"fn native_lurk_funcs.*?\{\s*\[(.*?)\]\s*\}", src, re.S)
    return re.findall(r"(\w+)\(", m.group(1))


def _balanced(src: str, open_idx: int) -> int:
    """Index just past the parenthesis matching src[open_idx] == '('; string and char literals skipped."""
    depth, i = 0, open_idx
    while i < len(src):
        c = src[i]
        if c == '"':
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
        elif c == "(":
            depth += 1
        elif c == ")":
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    raise ValueError("unbalanced func!")


def _macro_bodies(rel: str) -> list[str]:
    src = _strip_comments(_read(rel))
    out = []
    for m in re.finditer(r"\bfunc!\(", src):
        end = _balanced(src, m.end() - 1)
        out.append(src[m.end():end - 1].strip())
    return out


class Resolver:
    """Rust-side expressions -> field constants.  `symbol_digest(kind, name)` must return the 8 lanes the ZStore interns the
    symbol to (zstore.rs:409-445); pointers are positions in the preallocation order."""

    def __init__(self, symbol_digest):
        self.enums = enums()
        self.syms = preallocated_symbols()
        self.ptr = {s: i + 1 for i, s in enumerate(self.syms)}  # ingress.rs:63-66
        self.digest = {s: [int(x) for x in symbol_digest(*s)] for s in self.syms}

    def resolve(self, body: str) -> str:
        def arr(m):
            return "[" + ", ".join(str(x) for x in self.digest[("lurk", m.group(1))]) + "]"

        # error-message closures (`assert_eq!(a, b, |_, _| "..".to_string())`, `preimg(f, x, |fs| format!(..))`) decide what a
        # failing execution prints, nothing in the trace: dropped
        body = re.sub(r',\s*\|[^|]*\|\s*"[^"]*"\.to_string\(\)', "", body)
        body = re.sub(r',\s*\|[^|]*\|\s*format!\("[^"]*"(?:,[^()]*(?:\([^()]*\))?)?\)\s*(?=\))', "", body)
        body = re.sub(r'Array\(digests\.lurk_symbol_digest\("([^"]+)"\)\.clone\(\)\)', arr, body)
        body = re.sub(r'digests\.lurk_symbol_ptr\("([^"]+)"\)', lambda m: str(self.ptr[("lurk", m.group(1))]), body)
        # the only closure the sources hand to `match x [cloj] {..}` maps a builtin's name to its pointer
        body, n = re.subn(r"\[\|name\| digests\.builtin_symbol_ptr\(name\)\.to_field\(\)\]", "", body)
        if n:
            body = re.sub(r'"([^"]+)"', lambda m: str(self.ptr[("builtin", m.group(1))]), body)
        for en, table in self.enums.items():
            body = re.sub(r"\b" + en + r"::(\w+)", lambda m, t=table: str(t[m.group(1)]), body)
        left = re.findall(r"digests|::|\"", body)
        if left:
            raise ValueError(f"unresolved Rust-side expression(s) {sorted(set(left))} in: {body[:80]}...")
        return body

    def preallocate_symbols(self) -> str:
        """ingress.rs:99-140: per symbol `Array(digest); store; Const(addr); AssertEq`."""
        lines = ["fn preallocate_symbols(): [0] {"]
        for s in self.syms:
            lines.append("    let arr = [" + ", ".join(map(str, self.digest[s])) + "];")
            lines.append("    let ptr = store(arr);")
            lines.append(f"    let addr = {self.ptr[s]};")
            lines.append("    assert_eq!(ptr, addr);")
        lines.append("    return ()")
        lines.append("}")
        return "\n".join(lines)

    @staticmethod
    def eval_coroutine_expr_native() -> str:
        """The function `eval_coroutine_expr` builds by hand when there are no coroutines (eval_direct.rs:200-217)."""
        return ("fn eval_coroutine_expr(_head, _args_tag, _args, _env): [2] {\n    let zero = 0;\n    let one = 1;\n"
                "    assert_eq!(zero, one);\n    return (zero, zero)\n}")

    def functions(self) -> dict[str, str]:
        """name -> resolved `func!` text of every function of the native Lurk toplevel, in `native_lurk_funcs` order."""
        found = {}
        for rel in ("src/core/eval_direct.rs", "src/core/ingress.rs", "src/core/misc.rs"):
            for body in _macro_bodies(rel):
                name = re.search(r"\bfn\s+(\w+)\s*\(", body).group(1)
                found[name] = self.resolve(body)
        found["preallocate_symbols"] = self.preallocate_symbols()
        found["eval_coroutine_expr"] = self.eval_coroutine_expr_native()
        order = native_func_order()
        missing = [n for n in order if n not in found]
        assert not missing, missing
        return {n: found[n] for n in order}


# ---------------------------------------------------------------------------------------------------------------------------
# A reader for the subset of Lurk's surface syntax the benchmark programs use (src/core/parser/syntax.rs): lists, quote,
# unsuffixed / `u64` integers (both u64: syntax.rs:250-256), `n`-suffixed field elements, strings, characters, symbols.
# Symbols resolve the way State::init_lurk_state does for the user package (state.rs:196-213): a name that is a builtin or
# `nil` / `t` / `&rest` is imported from `.lurk(.builtin)`, anything else lives in `.lurk-user`; `:x` is a keyword.

_TOKEN = re.compile(r"""\s*(?:;[^\n]*(?:\n|$)\s*)*(!\(|\(|\)|'.'|'|"(?:[^"\\]|\\.)*"|[^\s()']+)""")


def read_lurk(text: str):
    """Lurk source -> the syntax tuples of lurk_amd.zstore (`syn_*`)."""
    from lurk_amd import zstore as zs

    builtins = {n for k, n in preallocated_symbols() if k == "builtin"}
    lurk = {n for k, n in preallocated_symbols() if k == "lurk"}
    toks = _TOKEN.findall(text)
    pos = 0

    def atom(t):
        if t.startswith("#0x") or t.startswith("#c0x"):  # big-num / commitment literal: base-p digits, little-endian (parser/syntax.rs:267-290)
            comm = t.startswith("#c0x")
            v, digest = int(t[4 if comm else 3:], 16), []
            for _ in range(8):
                digest.append(v % 2013265921)
                v //= 2013265921
            assert v == 0, "digest literal too big"
            return ("comm" if comm else "bignum", tuple(digest))
        if re.fullmatch(r"\d+(u64)?", t):
            return zs.syn_u64(int(t.removesuffix("u64")))
        if re.fullmatch(r"\d+n", t):
            return zs.syn_num(int(t[:-1]))
        if t.startswith('"'):
            return zs.syn_str(bytes(t[1:-1], "utf-8").decode("unicode_escape"))
        if len(t) == 3 and t[0] == "'" and t[2] == "'":
            return zs.syn_char(t[1])
        if t.startswith(":"):
            return zs.syn_sym(*t[1:].split("."), flags="keyword")
        if t in builtins:
            return zs.syn_builtin(t)
        if t in lurk:
            return zs.syn_sym(zs.LURK_PACKAGE, t)
        return zs.syn_user(t)

    def expr():
        nonlocal pos
        t = toks[pos]
        pos += 1
        if t == "(":
            xs = []
            while toks[pos] != ")":
                if toks[pos] == ".":
                    pos += 1
                    y = expr()
                    assert toks[pos] == ")"
                    pos += 1
                    return zs.syn_improper(xs, y)
                xs.append(expr())
            pos += 1
            return zs.syn_list(*xs)
        if t == "'":
            return zs.syn_quote(expr())
        return atom(t)

    out = expr()
    assert pos == len(toks), "trailing input"
    return out


def repl_forms(text: str):
    """The top-level forms of a REPL script: [(head or None for a bare expression, [argument texts], whole text)] -- `!(def f ..)` is
    ("def", ["f", ".."], "!(def f ..)").  Text in, text out."""
    toks = _TOKEN.findall(re.sub(r";[^\n]*", "", text))
    pos = 0

    def form():
        nonlocal pos
        t = toks[pos]
        pos += 1
        if t in ("(", "!("):
            out = [t]
            while toks[pos] != ")":
                out += form()
            pos += 1
            return out + [")"]
        if t == "'":
            return [t] + form()
        return [t]

    def txt(fm):
        return " ".join(fm).replace("( ", "(").replace(" )", ")").replace("' ", "'")

    def split(fm):
        kids, i, depth, cur = [], 1, 0, []
        while i < len(fm) - 1:
            t = fm[i]
            cur.append(t)
            depth += t in ("(", "!(")
            depth -= t == ")"
            if depth == 0 and t != "'":
                kids.append(cur)
                cur = []
            i += 1
        return kids

    out = []
    while pos < len(toks):
        fm = form()
        if fm[0] == "!(":
            kids = split(fm)
            out.append((txt(kids[0]), [txt(k) for k in kids[1:]], txt(fm)))
        else:
            out.append((None, [], txt(fm)))
    return out


def fold_repl_script(text: str, tail: str = "t") -> str:
    """A REPL script (`!(def ..)`, `!(defrec ..)`, `!(defq x !(transition s args..))`, `!(assert ..)`, `!(assert-eq a b)`, bare
    expressions: src/core/cli/meta.rs) as ONE Lurk expression with the same evaluation work: a definition becomes a `let` / `letrec`
    around everything after it, a chain transition the application `((cdr s) args..)` it performs, an assertion the expression it
    evaluates.  The REPL proves each top-level evaluation on its own; folded, the script is one execution of the same machine --
    which is what a measurement of the chips' relative heights needs (the script's own assertions guard it: the fold evaluates to
    `t` only when every one of them held).  Text in, text out; nothing is stored."""
    toks = _TOKEN.findall(re.sub(r";[^\n]*", "", text))  # (no `;` inside the scripts' string literals)
    pos = 0

    def form():  # one balanced form as a token list
        nonlocal pos
        t = toks[pos]
        pos += 1
        if t in ("(", "!("):
            out = [t]
            while toks[pos] != ")":
                out += form()
            pos += 1
            return out + [")"]
        if t == "'":
            return [t] + form()
        return [t]

    def split(fm):  # the direct children of a parenthesised form
        kids, i, depth, cur = [], 1, 0, []
        while i < len(fm) - 1:
            t = fm[i]
            cur.append(t)
            depth += t in ("(", "!(")
            depth -= t == ")"
            if depth == 0 and t != "'":
                kids.append(cur)
                cur = []
            i += 1
        return kids

    def txt(fm):
        return " ".join(fm).replace("( ", "(").replace(" )", ")").replace("' ", "'")

    def value(fm):  # an expression that may be a `!(transition s args..)` meta form
        if fm[0] == "!(":
            kids = split(fm)
            assert txt(kids[0]) == "transition", txt(kids[0])
            return "((cdr " + value(kids[1]) + ")" + "".join(" " + value(k) for k in kids[2:]) + ")"
        return txt(fm)

    forms = []
    while pos < len(toks):
        forms.append(form())
    out = tail  # (what the fold evaluates to when every assertion held: `t`, or the caller's final expression)
    for fm in reversed(forms):
        if fm[0] == "!(":
            kids = split(fm)
            head = txt(kids[0])
            if head in ("def", "defq"):
                out = f"(let (({txt(kids[1])} {value(kids[2])})) {out})"
            elif head == "defrec":
                out = f"(letrec (({txt(kids[1])} {value(kids[2])})) {out})"
            elif head == "assert":  # a failing assertion ends the evaluation with a keyword instead of the final `t`
                out = f"(if {value(kids[1])} {out} :assertion-failed)"
            elif head == "assert-eq":
                out = f"(if (eq {value(kids[1])} {value(kids[2])}) {out} :assertion-failed)"
            else:
                raise ValueError(f"meta command {head} is not folded")
        else:
            out = f"(begin {txt(fm)} {out})"
    return out


def demo_script(name: str) -> str:
    return _read(os.path.join("demo", name))


def fib_program(arg: int) -> str:
    """The benchmark's program text with its argument (benches/fib.rs:36-44), read from the reference at run time."""
    src = _read("benches/fib.rs")
    m = re.search(r'fn build_lurk_expr.*?format!\(\s*"(.*?)"\s*\)', src, re.S)
    return m.group(1).replace("{arg}", str(arg))
