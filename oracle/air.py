"""ORACLE (test infrastructure, not product code): numeric restatement of the Lair chips' AIR.

Independent of lurk_amd/csrc/lair/air.cpp: it walks the oracle's own bytecode (oracle/lair.py) with a
*numeric* builder, the way the reference's DebugConstraintBuilder does (/root/reference/src/air/debug.rs:209-406):
every `assert_*` appends the value of the asserted polynomial on the given (local, next) row pair, every
send / receive appends (multiplicity, tuple).  Used to check the product's symbolic walk + register program
constraint by constraint and tuple by tuple on arbitrary rows, and -- like the reference's own tests
(/root/reference/src/air/debug.rs:119-158) -- that valid traces satisfy every constraint and that the
send / receive multisets balance.

Follows:
  Func chips      /root/reference/src/lair/air.rs:158-552
  MemChip         /root/reference/src/lair/memory.rs:71-109
  BytesChip       /root/reference/src/gadgets/bytes/trace.rs:117-143
  Entrypoint      /root/reference/src/lair/lair_chip.rs:166-191
  provide/require /root/reference/src/air/builder.rs:42-104; relations /root/reference/src/lair/relations.rs:6-59,
                  /root/reference/src/gadgets/bytes/relation.rs:122-134
  depth gadget    /root/reference/src/gadgets/unsigned/less_than.rs:44-99, /root/reference/src/lair/air.rs:103-133

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.
"""
from __future__ import annotations

from collections import Counter

from .lair import DEPTH_LESS_THAN_SIZE, DEPTH_W, P, Toplevel, inv

CALL_TAG, MEMORY_TAG, BYTE_TAG = 0, 1, 3
INTERACTION_KIND_MEMORY = 1  # sphinx InteractionKind::Memory as usize [UPSTREAM-RECALL]


class Builder:
    """Numeric AirBuilder + LookupBuilder: records constraint values and interactions of one row pair."""

    def __init__(self, local, nxt, prep_local=(), prep_next=(), public=(), sels=(0, 0, 0)):
        self.local, self.next = list(local), list(nxt)
        self.prep_local, self.prep_next = list(prep_local), list(prep_next)
        self.public = list(public)
        self.is_first_row, self.is_last_row, self.is_transition = sels
        self.constraints: list[int] = []
        self.sends: list[tuple[int, list[int]]] = []  # (multiplicity, tuple)
        self.receives: list[tuple[int, list[int]]] = []

    # p3 AirBuilder
    def assert_zero(self, x, cond=None):
        self.constraints.append((x if cond is None else cond * x) % P)

    def assert_eq(self, a, b, cond=None):
        self.assert_zero(a - b, cond)

    def assert_one(self, x, cond=None):
        self.assert_zero(x - 1, cond)

    def assert_bool(self, x, cond=None):
        self.assert_zero(x * (x - 1), cond)

    # LookupBuilder (air/builder.rs)
    def receive(self, values, is_real):
        self.receives.append((is_real % P, [v % P for v in values]))

    def send(self, values, is_real):
        self.sends.append((is_real % P, [v % P for v in values]))

    def provide(self, relation, last_nonce, last_count, is_real):
        self.receive([last_nonce, last_count] + list(relation), is_real)
        self.send([0, 0] + list(relation), is_real)

    def require(self, relation, nonce, prev_nonce, prev_count, count_inv, is_real):
        count = prev_count + 1
        self.assert_one(count * count_inv, is_real)
        self.receive([prev_nonce, prev_count] + list(relation), is_real)
        self.send([nonce, count] + list(relation), is_real)

    def dump(self):
        """The flattening lurkhip_air_eval_rows uses: constraints; then per interaction (sends first)
        multiplicity followed by the tuple."""
        flat = []
        for m, vals in self.sends + self.receives:
            flat.append(m)
            flat += vals
        return list(self.constraints), flat


class ByteAirRecord:  # gadgets/bytes/builder.rs
    def __init__(self):
        self.records = []

    def range_check_u8_pair(self, i1, i2, is_real):
        self.records.append(([BYTE_TAG, 1, i1, i2], is_real))

    def range_check_u8_iter(self, xs, is_real):
        xs = list(xs)
        for i in range(0, len(xs), 2):
            self.range_check_u8_pair(xs[i], xs[i + 1] if i + 1 < len(xs) else 0, is_real)

    def less_than(self, i1, i2, r, is_real):
        self.records.append(([BYTE_TAG, 3, i1, i2, r], is_real))

    def require_all(self, b: Builder, nonce, requires):
        assert len(requires) == len(self.records)
        for (rel, is_real), (pn, pc, ci) in zip(self.records, requires):
            b.require(rel, nonce, pn, pc, ci, is_real)


def _return_idents(blk):
    c = blk["ctrl"]
    if c[0] == "return":
        return [c[1]]
    _, _, cases, uniq, d = c
    out = []
    for x in list(uniq) + ([d] if d is not None else []):
        out += _return_idents(x)
    return out


class FuncAir:
    def __init__(self, top: Toplevel, name: str):
        self.top = top
        self.f = top.funcs[top.index[name]]
        self.fe = top.funcs_e[top.index[name]]
        self.layout = top.layout(self.f)
        self.width = sum(self.layout.values())

    def eval(self, b: Builder):
        f, ls = self.f, self.layout
        loc = b.local
        o_in, o_out = 1, 1 + ls["input"]
        o_aux = o_out + ls["output"]
        o_sel = o_aux + ls["aux"]
        st = {"aux": 0, "out": 0}

        def next_aux():
            assert st["aux"] < ls["aux"], "walk ran past the aux columns"
            v = loc[o_aux + st["aux"]]
            st["aux"] += 1
            return v

        def next_require():
            return (next_aux(), next_aux(), next_aux())

        def return_sel(blk):
            return sum(loc[o_sel + i] for i in _return_idents(blk)) % P

        nonce = loc[0]
        b.assert_eq(b.next[0], nonce + 1, b.is_transition)
        vmap = []  # (is_const, value)
        call_inp = []
        for i in range(f["input_size"]):
            vmap.append((False, loc[o_in + i]))
            call_inp.append(loc[o_in + i])
        toplevel_sel = return_sel(f["body"])
        b.assert_bool(toplevel_sel)
        last_nonce, last_count = next_aux(), next_aux()
        out = [loc[o_out + i] for i in range(f["output_size"])]
        depth = []
        if f["partial"]:
            depth = [next_aux() for _ in range(DEPTH_W)]
            reqs = [next_require() for _ in range(DEPTH_W // 2 + DEPTH_W % 2)]
            rec = ByteAirRecord()
            rec.range_check_u8_iter(depth, toplevel_sel)
            rec.require_all(b, nonce, reqs)
            out = out + depth
        b.provide([CALL_TAG, f["index"]] + call_inp + out, last_nonce, last_count, toplevel_sel)

        def assert_less_than(wit, lhs, rhs, rec, is_real):
            is_equal = 0
            for i in range(DEPTH_W):
                if i > 0:
                    b.assert_eq(lhs[i], rhs[i], is_real * is_equal)
                b.assert_bool(wit[i], is_real)
                is_equal += wit[i]
            b.assert_one(is_equal, is_real)
            b.assert_eq(sum(l * w for l, w in zip(lhs, wit[:DEPTH_W])), wit[DEPTH_W], is_real)
            b.assert_eq(sum(r * w for r, w in zip(rhs, wit[:DEPTH_W])), wit[DEPTH_W + 1], is_real)
            rec.less_than(wit[DEPTH_W], wit[DEPTH_W + 1], 1, is_real)

        def eval_depth(sel, out_list):
            dep = [next_aux() for _ in range(DEPTH_W)]
            wit = [next_aux() for _ in range(DEPTH_LESS_THAN_SIZE)]
            rec = ByteAirRecord()
            assert_less_than(wit, dep, depth, rec, sel)
            rec.require_all(b, nonce, [next_require()])
            out_list += dep

        def eval_op(op, sel):
            k = op[0]
            if k == "assert_ne":
                coeffs = [next_aux() for _ in op[1]]
                acc = sum(c * (vmap[x][1] - vmap[y][1]) for c, x, y in zip(coeffs, op[1], op[2]))
                b.assert_one(acc, sel)
            elif k == "assert_eq":
                for x, y in zip(op[1], op[2]):
                    b.assert_eq(vmap[x][1], vmap[y][1], sel)
            elif k == "contains":
                y = vmap[op[2]][1]
                acc = vmap[op[1][0]][1] - y
                for x in op[1][1:]:
                    aux = next_aux()
                    b.assert_eq(acc * (vmap[x][1] - y), aux, sel)
                    acc = aux
                b.assert_zero(acc, sel)
            elif k == "const":
                vmap.append((True, op[1] % P))
            elif k in ("add", "sub"):
                (ca, a), (cb, c) = vmap[op[1]], vmap[op[2]]
                vmap.append((ca and cb, (a + c if k == "add" else a - c) % P))
            elif k == "mul":
                (ca, a), (cb, c) = vmap[op[1]], vmap[op[2]]
                if ca and cb:
                    vmap.append((True, a * c % P))
                else:  # air.rs:345-358: aux whenever not both operands are constants
                    aux = next_aux()
                    b.assert_eq(a * c, aux, sel)
                    vmap.append((False, aux))
            elif k == "inv":
                ca, a = vmap[op[1]]
                if ca:
                    vmap.append((True, inv(a)))
                else:
                    aux = next_aux()
                    b.assert_one(a * aux, sel)
                    vmap.append((False, aux))
            elif k == "not":
                ca, a = vmap[op[1]]
                if ca:
                    vmap.append((True, 1 if a % P == 0 else 0))
                else:
                    d, x = next_aux(), next_aux()
                    b.assert_zero(a * x, sel)
                    b.assert_one(a * d + x, sel)
                    vmap.append((False, x))
            elif k == "call":
                g = self.top.funcs[op[1]]
                outs = []
                for _ in range(g["output_size"]):
                    o = next_aux()
                    vmap.append((False, o))
                    outs.append(o)
                inp = [vmap[i][1] for i in op[2]]
                rec = next_require()
                if g["partial"]:
                    eval_depth(sel, outs)
                b.require([CALL_TAG, op[1]] + inp + outs, nonce, *rec, sel)
            elif k == "preimg":
                g = self.top.funcs[op[1]]
                inp = []
                for _ in range(g["input_size"]):
                    v = next_aux()
                    vmap.append((False, v))
                    inp.append(v)
                outs = [vmap[i][1] for i in op[2]]
                rec = next_require()
                if g["partial"]:
                    eval_depth(sel, outs)
                b.require([CALL_TAG, op[1]] + inp + outs, nonce, *rec, sel)
            elif k == "store":
                ptr = next_aux()
                vmap.append((False, ptr))
                vals = [vmap[i][1] for i in op[1]]
                rec = next_require()
                b.require([MEMORY_TAG, ptr] + vals, nonce, *rec, sel)
            elif k == "load":
                ptr = vmap[op[2]][1]
                vals = []
                for _ in range(op[1]):
                    o = next_aux()
                    vmap.append((False, o))
                    vals.append(o)
                rec = next_require()
                b.require([MEMORY_TAG, ptr] + vals, nonce, *rec, sel)
            elif k == "range_u8":
                reqs = [next_require() for _ in range((len(op[1]) + 1) // 2)]
                rec = ByteAirRecord()
                rec.range_check_u8_iter([vmap[i][1] for i in op[1]], sel)
                rec.require_all(b, nonce, reqs)
            elif k == "extern":
                chip = self.top.chips[op[1]]
                ins = [vmap[i][1] for i in op[2]]
                wit = [next_aux() for _ in range(chip.witness_size)]
                reqs = [next_require() for _ in range(chip.require_size)]
                for o in eval_chip(b, chip, sel, ins, wit, nonce, reqs):
                    vmap.append((False, o))
            elif k == "emit":
                pass
            else:
                raise AssertionError(k)

        def eval_block(blk, sel):
            for op in blk["ops"]:
                eval_op(op, sel)
            c = blk["ctrl"]
            if c[0] == "return":
                s = loc[o_sel + c[1]]
                for v in c[2]:
                    out_var = loc[o_out + st["out"]]
                    st["out"] += 1
                    b.assert_eq(vmap[v][1], out_var, s)
                return
            kind, _, cases, uniq, d = c
            n_map, saved = len(vmap), dict(st)
            if kind == "choose":
                blocks = list(uniq)
            else:  # Map iteration order: sorted by key (lair/map.rs:19-26)
                blocks = [blk2 for _, blk2 in sorted(cases.items())]
            if d is not None:
                blocks.append(d)
            for blk2 in blocks:
                eval_block(blk2, return_sel(blk2))
                del vmap[n_map:]
                st.update(saved)

        eval_block(f["body"], toplevel_sel)


# ------------------------------------------------------------------ extern chips
def _external_layer(s):
    """p3 Poseidon2ExternalMatrixGeneral: M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] per 4-chunk, then column sums."""
    w = len(s)
    for i in range(0, w, 4):
        x0, x1, x2, x3 = s[i:i + 4]
        s[i:i + 4] = [2 * x0 + 3 * x1 + x2 + x3, x0 + 2 * x1 + 3 * x2 + x3, x0 + x1 + 2 * x2 + 3 * x3, 3 * x0 + x1 + x2 + 2 * x3]
    sums = [sum(s[j] for j in range(k, w, 4)) for k in range(4)]
    for i in range(w):
        s[i] = s[i] + sums[i % 4]


def poseidon2_wide_eval(b: Builder, width, ins, output, cols, is_real):
    """Poseidon2Cols::eval, /root/reference/src/poseidon/wide/air.rs:15-124 (column order wide/columns.rs:16-32)."""
    from . import binding

    rp, diag, ext_rc, int_rc = binding.p2_params(width)
    W = width
    ext_state = lambda r, i: cols[r * W + i]
    ext_sbox = lambda r, i: cols[8 * W + r * W + i]
    int_init = lambda i: cols[16 * W + i]
    int_state0 = lambda r: cols[17 * W + r]
    int_sbox = lambda r: cols[17 * W + (rp - 1) + r]
    state = [is_real * x for x in ins]
    _external_layer(state)

    def external_round(r):
        for i in range(W):
            b.assert_eq(state[i], ext_state(r, i))
            state[i] = ext_state(r, i)
        for i in range(W):
            state[i] = state[i] + is_real * ext_rc[r][i]
        for i in range(W):
            s3 = ext_sbox(r, i)
            b.assert_eq(state[i] * state[i] * state[i], s3)
            state[i] = state[i] * (s3 * s3)
        _external_layer(state)

    for r in range(4):
        external_round(r)
    for r in range(rp):
        if r == 0:
            for i in range(W):
                b.assert_eq(state[i], int_init(i))
                state[i] = int_init(i)
        else:
            b.assert_eq(state[0], int_state0(r - 1))
            state[0] = int_state0(r - 1)
        state[0] = state[0] + is_real * int_rc[r]
        s3 = int_sbox(r)
        b.assert_eq(state[0] * state[0] * state[0], s3)
        state[0] = state[0] * (s3 * s3)
        total = sum(state)
        for i in range(W):
            state[i] = state[i] * diag[i] + total
    for r in range(4, 8):
        external_round(r)
    for st, o in zip(state, output):
        b.assert_eq(st, is_real * o)


def _assert_add(b, lhs, rhs, out, is_real):  # gadgets/unsigned/add.rs:16-58
    base_inv = inv(256)
    carry = 0
    for o, x, y in zip(out, lhs, rhs):
        carry = (x + y + carry - o) * base_inv
        b.assert_bool(carry, is_real)


def eval_chip(b: Builder, chip, is_real, ins, wit, nonce, reqs):
    """LurkChip::eval (core/chipset.rs:122-171): PoseidonChipset::eval (core/poseidon.rs:74-93), U64::eval (core/u64.rs:173-229)."""
    rec = ByteAirRecord()
    name = chip.name
    if name.startswith("hasher"):
        output, cols = wit[:8], wit[8:]
        poseidon2_wide_eval(b, chip.input_size, ins, output, cols, is_real)
        out = list(output)
    elif name in ("u64_add", "u64_sub"):
        result = wit[:8]
        rec.range_check_u8_iter(result, is_real)
        if name == "u64_add":
            _assert_add(b, ins[:8], ins[8:16], result, is_real)
        else:
            _assert_add(b, result, ins[8:16], ins[:8], is_real)
        out = list(result)
    elif name == "u64_mul":  # gadgets/unsigned/mul.rs:66-108,141-164
        carry, result = wit[:8], wit[8:16]
        lhs, rhs = ins[:8], ins[8:16]
        products = [0] * 8
        for i in range(8):
            for j in range(8 - i):
                products[i + j] += lhs[i] * rhs[j]
        carry_prev = 0
        for k in range(8):
            rec.records.append(([BYTE_TAG, 2, carry[k]], is_real))
            b.assert_eq(products[k] + carry_prev, result[k] + carry[k] * 256, is_real)
            carry_prev = carry[k]
        rec.range_check_u8_iter(result, is_real)
        out = list(result)
    elif name == "u64_lessthan":  # gadgets/unsigned/cmp.rs:48-118
        lhs, rhs = ins[:8], ins[8:16]
        is_comp, lhs_limb, rhs_limb, diff_inv, is_lt = wit[:8], wit[8], wit[9], wit[10], wit[11]
        is_equal = 1
        for i in reversed(range(8)):
            b.assert_bool(is_comp[i], is_real)
            is_equal = is_equal - is_comp[i]
            b.assert_eq(lhs[i], rhs[i], is_real * is_equal)
        b.assert_bool(is_equal, is_real)
        b.assert_eq(sum(x * f for x, f in zip(lhs, is_comp)), lhs_limb, is_real)
        b.assert_eq(sum(x * f for x, f in zip(rhs, is_comp)), rhs_limb, is_real)
        b.assert_eq((lhs_limb - rhs_limb) * diff_inv, 1 - is_equal, is_real)
        rec.less_than(lhs_limb, rhs_limb, is_lt, is_real)
        out = [is_lt]
    elif name == "u64_iszero":  # gadgets/unsigned/is_zero.rs:69-92,142-157
        inverses, is_zero = wit[:8], wit[8]
        b.assert_bool(is_zero, is_real)
        lc = 0
        for x, w_ in zip(ins[:8], inverses):
            b.assert_zero(x, is_real * is_zero)
            lc = lc + x * w_
        b.assert_eq(lc, 1 - is_zero, is_real)
        out = [is_zero]
    elif name == "u64_divrem":  # gadgets/unsigned/div_rem.rs:65-110
        a, bw = ins[:8], ins[8:16]
        inverses, qv, carry, qb, r = wit[:8], wit[8:16], wit[16:24], wit[24:32], wit[32:40]
        lw, cw = wit[40:50], wit[50:62]
        b.assert_one(sum(x * w_ for x, w_ in zip(bw, inverses)), is_real)
        rec.range_check_u8_iter(qv, is_real)
        products = [0] * 8
        for i in range(8):
            for j in range(8 - i):
                products[i + j] += qv[i] * bw[j]
        carry_prev = 0
        for k in range(8):
            rec.records.append(([BYTE_TAG, 2, carry[k]], is_real))
            b.assert_eq(products[k] + carry_prev, qb[k] + carry[k] * 256, is_real)
            carry_prev = carry[k]
        rec.range_check_u8_iter(qb, is_real)
        rec.range_check_u8_iter(r, is_real)
        _assert_add(b, r, qb, a, is_real)
        is_equal = 0
        for i in range(8):  # r < b: LessThanWitness<_, 8>
            if i > 0:
                b.assert_eq(r[i], bw[i], is_real * is_equal)
            b.assert_bool(lw[i], is_real)
            is_equal = is_equal + lw[i]
        b.assert_one(is_equal, is_real)
        b.assert_eq(sum(x * f for x, f in zip(r, lw[:8])), lw[8], is_real)
        b.assert_eq(sum(x * f for x, f in zip(bw, lw[:8])), lw[9], is_real)
        rec.less_than(lw[8], lw[9], 1, is_real)
        is_equal = 1
        for i in reversed(range(8)):  # qb <= a: CompareWitness<_, 8>
            b.assert_bool(cw[i], is_real)
            is_equal = is_equal - cw[i]
            b.assert_eq(qb[i], a[i], is_real * is_equal)
        b.assert_bool(is_equal, is_real)
        b.assert_eq(sum(x * f for x, f in zip(qb, cw[:8])), cw[8], is_real)
        b.assert_eq(sum(x * f for x, f in zip(a, cw[:8])), cw[9], is_real)
        b.assert_eq((cw[8] - cw[9]) * cw[10], 1 - is_equal, is_real)
        rec.less_than(cw[8], cw[9], cw[11], is_real)
        b.assert_one(cw[11] + is_equal, is_real)
        out = list(qv) + list(r)
    elif name == "big_num_lessthan":  # gadgets/big_num/cmp.rs:52-135, gadgets/unsigned/field.rs:34-84,120-139
        lhs, rhs = ins[:8], ins[8:16]
        is_equal = 1
        for i in reversed(range(8)):
            b.assert_bool(wit[i], is_real)
            is_equal = is_equal - wit[i]
            b.assert_eq(lhs[i], rhs[i], is_real * is_equal)
        b.assert_bool(is_equal, is_real)
        b.assert_eq(sum(x * f for x, f in zip(lhs, wit[:8])), wit[8], is_real)
        b.assert_eq(sum(x * f for x, f in zip(rhs, wit[:8])), wit[9], is_real)

        def field_to_word(fld, fw):
            is_msb_lt, wd = fw[0], fw[1:5]
            b.assert_bool(is_msb_lt, is_real)
            recomposed = 0
            for i in reversed(range(4)):
                recomposed = recomposed * 256 + wd[i]
            b.assert_eq(fld, recomposed, is_real)
            rec.less_than(wd[3], 0x78, is_msb_lt, is_real)
            when_eq = is_real * (1 - is_msb_lt)
            b.assert_eq(wd[3], 0x78, when_eq)
            for i in range(3):
                b.assert_eq(wd[i], 0, when_eq)
            rec.range_check_u8_iter(wd, is_real)
            return wd

        lwd = field_to_word(wit[8], wit[10:15])
        rwd = field_to_word(wit[9], wit[15:20])
        cw = wit[20:28]
        w_equal = 1
        for i in reversed(range(4)):
            b.assert_bool(cw[i], is_real)
            w_equal = w_equal - cw[i]
            b.assert_eq(lwd[i], rwd[i], is_real * w_equal)
        b.assert_bool(w_equal, is_real)
        b.assert_eq(sum(x * f for x, f in zip(lwd, cw[:4])), cw[4], is_real)
        b.assert_eq(sum(x * f for x, f in zip(rwd, cw[:4])), cw[5], is_real)
        b.assert_eq((cw[4] - cw[5]) * cw[6], 1 - w_equal, is_real)
        rec.less_than(cw[4], cw[5], cw[7], is_real)
        b.assert_eq(is_equal, w_equal, is_real)
        out = [cw[7]]
    else:
        raise NotImplementedError(f"AIR of extern chip {name}")
    rec.require_all(b, nonce, reqs)
    return out


class Poseidon2NarrowAir:
    """Poseidon2Chip (one row per round), /root/reference/src/poseidon/air.rs:21-165; columns poseidon/columns.rs:16-25."""

    def __init__(self, width):
        from . import binding

        self.w = width
        self.rp, self.diag, self.ext_rc, self.int_rc = binding.p2_params(width)
        self.rounds = 8 + self.rp
        self.width = 5 * width + 1 + self.rounds
        self.name = f"Poseidon2[{width}]"

    def round_constants(self):  # poseidon/config.rs:59-72
        return [list(c) for c in self.ext_rc[:4]] + [[c] for c in self.int_rc] + [list(c) for c in self.ext_rc[4:]]

    def eval(self, b: Builder):
        W, R, rp = self.w, self.rounds, self.rp

        def cols(row):
            o = [0, W, W + 1, W + 1 + R, 2 * W + 1 + R, 3 * W + 1 + R, 4 * W + 1 + R, 5 * W + 1 + R]
            return [row[o[i]:o[i + 1]] for i in range(7)]

        inp, (is_init,), rounds, add_rc_c, s3_c, s7_c, out = cols(b.local)
        nxt_in = cols(b.next)[0]
        is_external_first, is_internal, is_external_second = sum(rounds[:4]), sum(rounds[4:4 + rp]), sum(rounds[4 + rp:])
        is_external = is_external_first + is_external_second
        is_linear = is_init + is_external
        b.assert_bool(is_init)
        for f in rounds:
            b.assert_bool(f)
        is_real = is_init + is_internal + is_external
        b.assert_bool(is_real)
        add_rc = list(inp)
        for flag, consts in zip(rounds, self.round_constants()):
            for i, c in enumerate(consts):
                add_rc[i] = add_rc[i] + flag * c
        for got, want in zip(add_rc, add_rc_c):
            b.assert_eq(got, want, is_real)
        for x, s3, s7 in zip(add_rc_c, s3_c, s7_c):
            b.assert_eq(x * x * x, s3)
            b.assert_eq(s3 * s3 * x, s7)
        sbox_result = [is_init * add_rc_c[0] + (is_internal + is_external) * s7_c[0]]
        sbox_result += [(is_init + is_internal) * add_rc_c[i] + is_external * s7_c[i] for i in range(1, W)]
        state = list(sbox_result)
        _external_layer(state)
        for st, o in zip(state, out):
            b.assert_eq(st, o, is_linear)
        total = sum(sbox_result)
        for i in range(W):
            b.assert_eq(sbox_result[i] * self.diag[i] + total, out[i], is_internal)
        is_not_last_round = is_real - rounds[-1]
        for o, ni in zip(out, nxt_in):
            b.assert_eq(o, ni, is_not_last_round)


class MemAir:  # lair/memory.rs:71-109
    def __init__(self, mem_len):
        self.len = mem_len
        self.width = 4 + mem_len

    def eval(self, b: Builder):
        loc, nxt = b.local, b.next
        is_real, ptr_local, last_nonce, last_count = loc[0], loc[1], loc[2], loc[3]
        is_real_next, ptr_next = nxt[0], nxt[1]
        b.assert_bool(is_real)
        is_real_transition = is_real_next * b.is_transition
        b.assert_one(is_real, is_real_transition)
        b.assert_one(ptr_local, b.is_first_row * is_real)
        b.assert_eq(ptr_local + 1, ptr_next, is_real_transition)
        b.provide([MEMORY_TAG, ptr_local] + loc[4:4 + self.len], last_nonce, last_count, is_real)


class BytesAir:  # gadgets/bytes/trace.rs:117-143
    width, prep_width = 13, 6

    def eval(self, b: Builder):
        main, prep = b.local, b.prep_local
        is_real = main[0]
        b.assert_bool(is_real)
        i1, i2 = prep[0], prep[1]
        relations = [
            [BYTE_TAG, 1, i1, i2], [BYTE_TAG, 2, i1 + i2 * 256],
            [BYTE_TAG, 3, i1, i2, prep[2]], [BYTE_TAG, 4, i1, i2, prep[3]],
            [BYTE_TAG, 5, i1, i2, prep[4]], [BYTE_TAG, 6, i1, i2, prep[5]],
        ]
        for k, rel in enumerate(relations):
            b.provide(rel, main[1 + 2 * k], main[2 + 2 * k], is_real)


class EntrypointAir:  # lair/lair_chip.rs:166-191
    def __init__(self, func_idx, num_public_values):
        self.func_idx, self.width = func_idx, num_public_values

    def eval(self, b: Builder):
        pv = b.local[: self.width]
        for a, c in zip(pv, b.public):
            b.assert_eq(a, c)
        b.require([CALL_TAG, self.func_idx] + pv, 0, 0, 0, 1, 1)


def eval_rows(air, local, nxt, prep_local=None, prep_next=None, public=(), sels=None):
    """Row-pair evaluation in the layout of lurkhip_air_eval_rows: (constraints[n][K], interactions[n][T])."""
    cons, inter = [], []
    for i in range(len(local)):
        b = Builder(local[i], nxt[i], prep_local[i] if prep_local is not None else (), prep_next[i] if prep_next is not None else (),
                    public, sels[i] if sels is not None else (0, 0, 0))
        air.eval(b)
        c, t = b.dump()
        cons.append(c)
        inter.append(t)
    return cons, inter


def debug_check(chips_and_traces, public=()):
    """The reference's debug_chip_constraints_and_queries (air/debug.rs:119-206): every constraint vanishes on
    every row (indicator selectors, next row wraps) and sends == receives as multisets.
    chips_and_traces: iterable of (air, main rows, preprocessed rows or None)."""
    sends, receives = Counter(), Counter()
    for air, main, prep in chips_and_traces:
        h = len(main)
        for r in range(h):
            n = (r + 1) % h
            b = Builder(main[r], main[n], prep[r] if prep is not None else (), prep[n] if prep is not None else (), public,
                        (1 if r == 0 else 0, 1 if r == h - 1 else 0, 0 if r == h - 1 else 1))
            air.eval(b)
            bad = [k for k, v in enumerate(b.constraints) if v]
            assert not bad, f"{type(air).__name__}: row {r}: constraints {bad} do not vanish"
            for m, vals in b.sends:
                assert m in (0, 1)
                if m:
                    sends[tuple(vals)] += 1
            for m, vals in b.receives:
                assert m in (0, 1)
                if m:
                    receives[tuple(vals)] += 1
    assert sends == receives, "send / receive multisets differ"
    return sum(sends.values())
