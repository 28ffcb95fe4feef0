"""GPU parity of the AIR programs (symbolic walk in C++ -> register program -> device VM) against the oracle's
numeric AIR, constraint by constraint and tuple by tuple, on real trace rows and on random rows with random
selector values (the constraint polynomials must agree everywhere, not just where they vanish); plus the
device-side debug check of whole traces."""
import numpy as np
import pytest

from lair_helpers import PARTIAL_SRC, load_cases
from lurk_amd import air, field, lair, synth
from lurk_amd.programs import synth_eval as se
from oracle import air as oa
from oracle import lair as ol

pytestmark = pytest.mark.gpu
P = field.P
DEMO = load_cases()[0]["source"]


def rows_for(width, real_rows, seed):
    """real consecutive pairs + random pairs, random selectors"""
    rnd = synth.field_elements((24, width), seed=seed)
    local = [list(r) for r in real_rows[:-1]] + [list(map(int, r)) for r in rnd[:12]]
    nxt = [list(r) for r in real_rows[1:]] + [list(map(int, r)) for r in rnd[12:]]
    sels = synth.field_elements((len(local), 3), seed=seed + 1)
    return np.array(local, dtype=np.uint32), np.array(nxt, dtype=np.uint32), sels


def compare(ctx, chip_air, oracle_air, local, nxt, sels, prep_local=None, prep_next=None, public=()):
    cons, inter = chip_air.eval_rows(ctx, local, nxt, prep_local, prep_next, public, sels)
    ocons, ointer = oa.eval_rows(oracle_air, local.tolist(), nxt.tolist(), prep_local.tolist() if prep_local is not None else None,
                                 prep_next.tolist() if prep_next is not None else None, list(public), [tuple(map(int, s)) for s in sels])
    assert cons.shape[1] == len(ocons[0]) and inter.shape[1] == len(ointer[0])
    assert cons.tolist() == ocons
    assert inter.tolist() == ointer


@pytest.mark.parametrize("src,calls", [(DEMO, [("fib", [9]), ("factorial", [6]), ("even", [7])]), (PARTIAL_SRC, [("top", [9])]),
                                       (se.SOURCE, [("synth_eval", [1, 40, 0])])], ids=["demo", "partial", "synth_eval"])
def test_func_air_matches_oracle(ctx, src, calls):
    top, otop = lair.Toplevel(src), ol.Toplevel(src)
    oq = ol.QueryRecord(otop)
    for name, args in calls:
        ol.execute(otop, name, args, oq)
    for i, f in enumerate(otop.funcs):
        rows, width = ol.generate_trace(otop, f["name"], oq)
        if len(rows) < 2:
            rows = [[0] * width, [0] * width]
        a = air.ChipAir.for_func(top, i)
        assert a.width == width
        local, nxt, sels = rows_for(width, rows[:40], seed=100 + i)
        compare(ctx, a, oa.FuncAir(otop, f["name"]), local, nxt, sels)


def test_mem_bytes_entrypoint_air_match_oracle(ctx):
    for ml in lair.MEM_TABLE_SIZES:
        a = air.ChipAir.for_mem(ml)
        local, nxt, sels = rows_for(4 + ml, [[1, 1, 0, 2] + [5] * ml, [1, 2, 3, 1] + [7] * ml, [0] * (4 + ml)], seed=7 + ml)
        compare(ctx, a, oa.MemAir(ml), local, nxt, sels)
    a = air.ChipAir.for_bytes()
    local, nxt, sels = rows_for(13, [[1] + [3, 4] * 6, [0] * 13], seed=33)
    prep = synth.field_elements((len(local), 6), seed=34)
    compare(ctx, a, oa.BytesAir(), local, nxt, sels, prep, prep[::-1].copy())
    a = air.ChipAir.for_entrypoint(2, 7)
    local, nxt, sels = rows_for(7, [[1, 2, 3, 4, 5, 6, 7], [1, 2, 3, 4, 5, 6, 7]], seed=35)
    compare(ctx, a, oa.EntrypointAir(2, 7), local, nxt, sels, public=[1, 2, 3, 9, 5, 6, 7])
    assert a.num_public_values == 7


def _dev_trace(ctx, chip, shard):
    import torch

    _, h, w = chip.trace_shape(shard)
    t = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    chip.generate_trace_dev(shard, t, repr=1)
    ctx.sync()
    return t, h


def test_check_trace_on_device(ctx):
    import torch

    top = lair.Toplevel(PARTIAL_SRC)
    q = lair.QueryRecord(top)
    top.execute_by_name("top", [11], q)
    shard = lair.Shard.new(q)
    for i in range(top.num_funcs()):
        chip = lair.FuncChip(ctx, i, top)
        t, h = _dev_trace(ctx, chip, shard)
        a = air.ChipAir.for_func(top, i)
        assert a.check_trace(ctx, h, t) == (-1, -1), a.name
        if h >= 4:
            bad = t.clone()
            bad[2, 1] += 1  # an input column of row 2
            row, k = a.check_trace(ctx, h, bad)
            assert row in (1, 2) and k >= 0, (a.name, row, k)
    # a big synthetic trace: 2^16 rows of the bench function
    top = lair.Toplevel(se.SOURCE)
    q = lair.QueryRecord(top)
    top.execute_by_name(se.FUNC, se.args_for_rows(1 << 16), q)
    chip = lair.FuncChip(ctx, top.func_index(se.FUNC), top)
    t, h = _dev_trace(ctx, chip, lair.Shard.new(q))
    assert h == 1 << 16
    a = air.ChipAir.for_func(top, top.func_index(se.FUNC))
    assert a.check_trace(ctx, h, t) == (-1, -1)
    t[40000, 30] += 1
    row, _ = a.check_trace(ctx, h, t)
    assert row == 40000
    del torch


def _machine_chips(ctx, src, entry, args):
    """Product traces (device, Montgomery) + oracle traces of every chip of a small machine."""
    import torch

    top, otop = lair.Toplevel(src), ol.Toplevel(src)
    q, oq = lair.QueryRecord(top), ol.QueryRecord(otop)
    top.execute_by_name(entry, args, q)
    ol.execute(otop, entry, args, oq)
    shard = lair.Shard.new(q)
    chips = []
    pv = q.expect_public_values()
    ep = np.array([pv], dtype=np.uint32)
    chips.append((air.ChipAir.for_entrypoint(top.func_index(entry), len(pv)), oa.EntrypointAir(top.func_index(entry), len(pv)),
                  torch.from_numpy(field.to_monty(ep).view(np.int32)).cuda(), ep.tolist()))
    for i, f in enumerate(otop.funcs):
        rows, _ = ol.generate_trace(otop, f["name"], oq)
        if not rows:
            continue
        t, h = _dev_trace(ctx, lair.FuncChip(ctx, i, top), shard)
        chips.append((air.ChipAir.for_func(top, i), oa.FuncAir(otop, f["name"]), t, rows))
    for ml in lair.MEM_TABLE_SIZES:
        rows = ol.mem_trace(oq, ml)
        t = torch.from_numpy(field.to_monty(np.array(rows, dtype=np.uint32)).view(np.int32)).cuda()
        chips.append((air.ChipAir.for_mem(ml), oa.MemAir(ml), t, rows))
    return chips, pv


def test_permutation_trace_matches_oracle_and_sums_to_zero(ctx):
    import torch

    from oracle import stark as os_

    alpha, beta = (11, 22, 33, 44), (5, 6, 7, 2013265920)
    for src, entry, args in [(DEMO, "fib", [10]), (se.SOURCE, "synth_eval", [1, 30, 0])]:
        chips, pv = _machine_chips(ctx, src, entry, args)
        total = os_.ZERO
        for a, oair_, t, rows in chips:
            h = len(rows)
            out = torch.zeros((h, 4 * a.permutation_width), dtype=torch.int32, device="cuda")
            cs = a.permutation_trace(ctx, h, t, None, alpha + beta, out)
            got = field.from_monty(out.cpu().numpy().view(np.uint32)).reshape(h, a.permutation_width, 4)
            want = os_.permutation_trace(oair_, rows, None, alpha, beta, 1 << a.log_quotient_degree, public=pv)
            assert got.tolist() == [[list(c) for c in r] for r in want], a.name
            assert tuple(int(x) for x in cs) == want[-1][-1]
            total = os_.ef_add(total, tuple(int(x) for x in cs))
        # the machine's lookups balance: the chips' cumulative sums cancel (what the verifier checks)
        assert total == os_.ZERO


def test_permutation_trace_on_extreme_rows(ctx):
    """The LogUp sums are accumulated in 64-bit lanes with deferred reduction (lazy_ef.h): rows filled with p - 1, with uniform
    field elements and with zeros, under challenges whose coefficients sit at the edges of the centred range, must still match
    the oracle's canonical arithmetic bit for bit."""
    import torch

    from oracle import stark as os_

    half = (P - 1) // 2
    challenges = [((half, half + 1, P - 1, 1), (half + 1, half, P - 1, half)), ((11, 22, 33, 44), (5, 6, 7, 2013265920))]
    top, otop = lair.Toplevel(se.SOURCE), ol.Toplevel(se.SOURCE)
    for name in ("synth_eval", "aux"):
        a, oair_ = air.ChipAir.for_func(top, top.func_index(name)), oa.FuncAir(otop, name)
        h = 64
        fills = [np.full((h, a.width), P - 1, dtype=np.uint32), synth.field_elements((h, a.width), seed=77), np.zeros((h, a.width), dtype=np.uint32)]
        for rows in fills:
            t = torch.from_numpy(field.to_monty(rows).view(np.int32)).cuda()
            for alpha, beta in challenges:
                out = torch.zeros((h, 4 * a.permutation_width), dtype=torch.int32, device="cuda")
                try:
                    want = os_.permutation_trace(oair_, rows.tolist(), None, alpha, beta, 1 << a.log_quotient_degree, public=[0] * 64)
                except ZeroDivisionError:
                    continue  # a denominator vanished on this synthetic row: nothing to compare
                a.permutation_trace(ctx, h, t, None, alpha + beta, out)
                got = field.from_monty(out.cpu().numpy().view(np.uint32)).reshape(h, a.permutation_width, 4)
                assert got.tolist() == [[list(c) for c in r] for r in want], (name, alpha)


def test_large_permutation_trace_cumulative_sums_cancel(ctx):
    """2^16 rows of the bench function + its callee + the memory tables: scan over many chunks."""
    import torch

    top = lair.Toplevel(se.SOURCE)
    q = lair.QueryRecord(top)
    top.execute_by_name(se.FUNC, se.args_for_rows(1 << 16), q)
    shard = lair.Shard.new(q)
    ch = [3, 1, 4, 1, 5, 9, 2, 6]
    total = np.zeros(4, dtype=np.uint64)
    pv = q.expect_public_values()
    ep = torch.from_numpy(field.to_monty(np.array([pv], dtype=np.uint32)).view(np.int32)).cuda()
    items = [(air.ChipAir.for_entrypoint(top.func_index(se.FUNC), len(pv)), ep, 1)]
    for i in range(top.num_funcs()):
        t, h = _dev_trace(ctx, lair.FuncChip(ctx, i, top), shard)
        items.append((air.ChipAir.for_func(top, i), t, h))
    for ml in lair.MEM_TABLE_SIZES:
        m = lair.MemChip(ctx, ml).generate_trace(shard, repr=1)
        items.append((air.ChipAir.for_mem(ml), torch.from_numpy(m.view(np.int32)).cuda(), m.shape[0]))
    for a, t, h in items:
        out = torch.zeros((h, 4 * a.permutation_width), dtype=torch.int32, device="cuda")
        total += a.permutation_trace(ctx, h, t, None, ch, out).astype(np.uint64)
    assert not (total % P).any()


def test_quotient_matches_oracle(ctx):
    """main commit -> permutation trace -> commit -> quotient chunks, chip by chip, against the oracle's
    quotient computed from its own LDEs and numeric AIR; then the chunks' commitment LDE against the
    oracle's coset LDE (shift w_Q^-c)."""
    import torch

    from lurk_amd import commit as cm
    from oracle import stark as os_

    perm_alpha, perm_beta, alpha = (11, 22, 33, 44), (5, 6, 7, 2013265920), (1000, 2000, 3000, 4000)
    for src, entry, args in [(DEMO, "fib", [10]), (se.SOURCE, "synth_eval", [1, 13, 0])]:
        chips, pv = _machine_chips(ctx, src, entry, args)
        for a, oair_, t, rows in chips:
            h = len(rows)
            log_n = h.bit_length() - 1
            lqd = a.log_quotient_degree
            assert lqd == 1
            main_c = cm.commit_dev(ctx, [t], [log_n], [a.width], log_blowup=1, repr=1)
            perm = torch.zeros((h, 4 * a.permutation_width), dtype=torch.int32, device="cuda")
            cs = a.permutation_trace(ctx, h, t, None, perm_alpha + perm_beta, perm)
            perm_c = cm.commit_dev(ctx, [perm], [log_n], [4 * a.permutation_width], log_blowup=1, repr=1)
            out = torch.zeros((1 << lqd, h, 4), dtype=torch.int32, device="cuda")
            a.quotient(ctx, log_n, main_c.matrix_dev(0)[0], None, perm_c.matrix_dev(0)[0], perm_alpha + perm_beta, alpha, cs, out, public=pv)
            ctx.sync()
            got = field.from_monty(out.cpu().numpy().view(np.uint32))
            # oracle
            operm = os_.permutation_trace(oair_, rows, None, perm_alpha, perm_beta, 1 << lqd, public=pv)
            main_lde = os_.coset_lde(rows, 1)
            perm_flat = os_.coset_lde(os_.flatten_ef_rows(operm), 1)
            perm_lde = [[tuple(r[4 * j:4 * j + 4]) for j in range(a.permutation_width)] for r in perm_flat]
            want = os_.quotient_chunks(oair_, log_n, main_lde, None, perm_lde, perm_alpha, perm_beta, alpha, operm[-1][-1], public=pv, lqd=lqd)
            assert got.tolist() == [[list(v) for v in chunk] for chunk in want], a.name
            # commit the chunks over their cosets: LDE == oracle coset LDE with shift g / (g w_Q^c) = w_Q^-c
            wq = os_.two_adic_generator(log_n + lqd)
            shifts = [pow(wq, (-c) % (P - 1), P) for c in range(1 << lqd)]
            qc = cm.commit_cosets_dev(ctx, [out[c] for c in range(1 << lqd)], [log_n] * (1 << lqd), [4] * (1 << lqd), shifts, log_blowup=1)
            for c in range(1 << lqd):
                lde_c = qc.lde_host(c)
                want_lde = os_.bit_reverse_rows(os_.coset_lde([list(v) for v in want[c]], 1, shift=shifts[c]))
                assert lde_c.tolist() == want_lde, (a.name, c)
            for c_ in (main_c, perm_c, qc):
                c_.close()


def test_extern_chip_airs_match_oracle(ctx, oracle):
    """Poseidon2 wide AIR (hasher3/4/5) and u64 gadget AIRs, constraint by constraint, on real and random rows."""
    from lair_helpers import U64_SRC
    from test_lair_gpu import oracle_chip_callbacks

    poseidon, witness = oracle_chip_callbacks(oracle)

    def u64(v):
        return [(v >> (8 * i)) & 0xFF for i in range(8)]

    top, otop = lair.Toplevel(U64_SRC, lurk_chips=True), ol.Toplevel(U64_SRC, chips=ol.lurk_chips())
    oq = ol.QueryRecord(otop)
    for name, args in [("u64_ops", u64(5) + u64(7)), ("u64_ops", u64(2**64 - 1) + u64(1)), ("chain", [9, 8, 7, 6, 5, 4, 3, 2]),
                       ("hash5", list(range(40))), ("u64_more", u64(0xFEDCBA9876543210) + u64(0x1234567)),
                       ("big_lt", [1, 2, 3, 4, 5, 6, 7, 8] + [1, 2, 3, 4, 5, 6, 9, 8])]:
        ol.execute(otop, name, args, oq, poseidon=poseidon)
    for i, f in enumerate(otop.funcs):
        rows, width = ol.generate_trace(otop, f["name"], oq, witness=witness)
        a = air.ChipAir.for_func(top, i)
        assert a.width == width and a.max_constraint_degree <= 3
        local, nxt, sels = rows_for(width, rows[:4], seed=300 + i)
        compare(ctx, a, oa.FuncAir(otop, f["name"]), local[:8], nxt[:8], sels[:8])


@pytest.mark.parametrize("seed", [6, 16, 35, 41, 52, 62, 68])
def test_random_programs_func_air_matches_oracle(ctx, seed):
    """AIRs of random programs (tests/lair_random.py): the device evaluator and the oracle's numeric AIR agree constraint by
    constraint and tuple by tuple on real row pairs and on random rows with random selectors; and the oracle's own property
    check (every constraint vanishes on every real row, lookups balance) accepts the machine."""
    import lair_random as lr

    src, calls, _ = lr.program(seed)
    top, otop = lair.Toplevel(src), ol.Toplevel(src)
    oq = ol.QueryRecord(otop)
    for name, args in calls[:1]:
        ol.execute(otop, name, args, oq)
    chips = [(oa.EntrypointAir(otop.index[calls[0][0]], len(oq.public_values)), [list(oq.public_values)], None)]
    for i, f in enumerate(otop.funcs):
        rows, width = ol.generate_trace(otop, f["name"], oq)
        if rows:
            chips.append((oa.FuncAir(otop, f["name"]), rows, None))
        if len(rows) < 2:
            rows = [[0] * width, [0] * width]
        a = air.ChipAir.for_func(top, i)
        assert a.width == width
        local, nxt, sels = rows_for(width, rows[:40], seed=300 + i)
        compare(ctx, a, oa.FuncAir(otop, f["name"]), local, nxt, sels)
    for ml in ol.MEM_TABLE_SIZES:
        chips.append((oa.MemAir(ml), ol.mem_trace(oq, ml), None))
    prep = [[i & 0xFF, i >> 8, int((i & 0xFF) < (i >> 8)), (i & 0xFF) & (i >> 8), (i & 0xFF) ^ (i >> 8), (i & 0xFF) | (i >> 8)] for i in range(1 << 16)]
    chips.append((oa.BytesAir(), ol.bytes_trace(oq), prep))
    assert oa.debug_check(chips, public=oq.public_values) > 0
