"""The representative workloads on the GPU (BASELINE.json configs 3-5 as SURVEY.md 8d states them):
  * fib-mix / lurk-mix machines (lurk_amd/programs/lurk_mix.py: the reference's 39 chip widths) prove and the oracle's
    verifier accepts the proofs: permutation / quotient / openings on a >= 40-chip irregular machine (config 5);
  * config 4: ONE execution with 2^22 eval rows, sharded 8 x 2^19 (`Shard::shard`,
    /root/reference/src/lair/execute.rs:186-216; inclusion rules /root/reference/src/lair/lair_chip.rs:124-139), all shards
    proved on one GPU one at a time; every shard's cumulative sum is non-zero, they only cancel over the whole set, and the
    oracle verifies the set."""
import numpy as np
import pytest

import lurk_amd
from lurk_amd import lair, prover
from lurk_amd.programs import lurk_mix as lm
from oracle import air as oa
from oracle import binding as ob
from oracle import lair as ol
from oracle import stark as os_

pytestmark = pytest.mark.gpu


def oracle_airs(mix, n_public):
    otop = ol.Toplevel(mix.source, chips=ol.lurk_chips())
    airs = [oa.EntrypointAir(otop.index[mix.entry], n_public)]
    airs += [oa.FuncAir(otop, f["name"]) for f in otop.funcs]
    airs += [oa.MemAir(ml) for ml in ol.MEM_TABLE_SIZES]
    airs.append(oa.BytesAir())
    return airs


def run(ctx, mix, shard_size=None, num_queries=8, pow_bits=6, compile_min_log_rows=None):
    top = lair.Toplevel(mix.source, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute_by_name(mix.entry, mix.main_args, q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, mix.entry, len(pv))
    root = m.setup()
    cfg = lair.ShardingConfig(shard_size) if shard_size else None
    if compile_min_log_rows is not None:
        sh0 = (lair.Shard.new(q).shard(cfg) if cfg else [lair.Shard.new(q)])[0]
        prepared = m.prepare_shard(sh0)
        m.compile_airs(prepared, compile_min_log_rows)
        del prepared
    proofs = m.prove(q, cfg, num_queries=num_queries, pow_bits=pow_bits)
    assert m.verify(proofs)  # the product's host verifier (csrc/verify.cpp), before the oracle's
    return m, top, q, root, proofs, pv


@pytest.mark.parametrize("mix", [lm.fib_mix(300), lm.lurk_mix(1 << 12)], ids=["fib-mix", "lurk-mix-2^12"])
def test_mix_machine_proves_and_verifies(ctx, mix):
    m, top, q, root, proofs, pv = run(ctx, mix)
    assert len(proofs) == 1 and len(pv) == 44
    p = proofs[0]
    # every function of the machine has rows, plus entrypoint, the used memory tables and the byte table
    assert len(p.chips) >= top.num_funcs() + 2
    widths = sorted(c.width for c in p.chips)
    # (a real fib run touches hash4 only: 655 columns; mastermind's machine has all three hashers, 815 the widest)
    assert widths[-1] == (655 if mix.name == "fib-mix" else 815) and widths[0] <= 10
    assert prover.grand_sum(proofs) == (0, 0, 0, 0)
    assert os_.verify_machine(oracle_airs(mix, len(pv)), root, [16], [6], proofs, ob.merkle_verify)
    m.close()


def test_sharded_mix_matches_reference_inclusion_rules(ctx):
    """Small sharded run: Entrypoint and memory chips only in shard 0, the byte chip in every shard (real rows in shard 0
    only), a function chip exactly where its row range is non-empty; sums cancel only over the set; oracle accepts."""
    mix = lm.fib_mix(1 << 10)
    m, top, q, root, proofs, pv = run(ctx, mix, shard_size=1 << 7)
    assert len(proofs) == 8
    n_funcs = top.num_funcs()
    for s, p in enumerate(proofs):
        names = {m.chips[c.machine_index][0] for c in p.chips}
        assert ("entrypoint" in names) == (s == 0)
        assert ("mem" in names) == (s == 0)
        assert "bytes" in names
        for c in p.chips:
            kind, arg, _ = m.chips[c.machine_index]
            if kind == "func":
                rows = q.num_func_queries(arg)
                assert rows > s * (1 << 7), (s, arg)  # the chip is included only where its range is non-empty
                assert 1 << c.log_n >= min(1 << 7, rows - s * (1 << 7))
        assert prover.shard_sum(p) != (0, 0, 0, 0), s
    assert prover.grand_sum(proofs) == (0, 0, 0, 0)
    assert prover.grand_sum(proofs[:-1]) != (0, 0, 0, 0)
    assert os_.verify_machine(oracle_airs(mix, len(pv)), root, [16], [6], proofs, ob.merkle_verify)
    m.close()


def test_config4_one_execution_2p22_rows_in_8_shards(ctx):
    """BASELINE.json configs[3] on one GPU: 2^22 eval rows, max_shard_size 2^19, 8 shards proved one at a time (phase-2
    regeneration keeps one shard resident), full FRI parameters; the oracle's verifier accepts the set."""
    mix = lm.fib_mix(1 << 22)
    m, top, q, root, proofs, pv = run(ctx, mix, shard_size=1 << 19, num_queries=100, pow_bits=16, compile_min_log_rows=17)
    assert len(proofs) == 8
    eval_idx = top.func_index("eval")
    for s, p in enumerate(proofs):
        ev = [c for c in p.chips if m.chips[c.machine_index][:2] == ("func", eval_idx)]
        assert len(ev) == 1 and ev[0].log_n == 19 and ev[0].width == 78
        assert prover.shard_sum(p) != (0, 0, 0, 0), s
    assert prover.grand_sum(proofs) == (0, 0, 0, 0)
    assert os_.verify_machine(oracle_airs(mix, len(pv)), root, [16], [6], proofs, ob.merkle_verify)
    m.close()


def test_bench_workload_fib_mix_2p20_one_shard_verifies(ctx):
    """bench.py's default step exactly: fib-mix, ONE shard of 2^20 eval rows, chips of 2^17 rows and
    more on compiled AIR / trace kernels, 100 queries, 16 PoW bits; the oracle's verifier accepts the proof."""
    mix = lm.fib_mix(1 << 20)
    m, top, q, root, proofs, pv = run(ctx, mix, num_queries=100, pow_bits=16, compile_min_log_rows=17)
    assert len(proofs) == 1
    ev = [c for c in proofs[0].chips if m.chips[c.machine_index][:2] == ("func", top.func_index("eval"))]
    assert len(ev) == 1 and ev[0].log_n == 20 and ev[0].width == 78
    assert sum(c.width << c.log_n for c in proofs[0].chips) / (1 << 20) > 280  # the bench's 285 main columns per eval row
    assert prover.grand_sum(proofs) == (0, 0, 0, 0)
    assert os_.verify_machine(oracle_airs(mix, len(pv)), root, [16], [6], proofs, ob.merkle_verify)
    m.close()


def test_config5_lurk_mix_2p18_verifies(ctx):
    """BASELINE.json configs[4] at the bench height (`bench.py --workload lurk-mix`: 2^18 eval rows, all 39 functions, widths
    9 ... 815): full FRI parameters, compiled chips, oracle-verified."""
    mix = lm.lurk_mix(1 << 18)
    m, top, q, root, proofs, pv = run(ctx, mix, num_queries=100, pow_bits=16, compile_min_log_rows=17)
    assert len(proofs) == 1 and len(proofs[0].chips) >= top.num_funcs() + 2
    assert prover.grand_sum(proofs) == (0, 0, 0, 0)
    assert os_.verify_machine(oracle_airs(mix, len(pv)), root, [16], [6], proofs, ob.merkle_verify)
    m.close()


def test_reference_default_shard_2p22_rows_one_shard(ctx):
    """The reference's default SHARD_SIZE (/root/reference/src/lair/execute.rs:231-241: 1 << 22): ONE fib-mix shard of 2^22 eval
    rows on one GPU -- LDE height 2^23, matrices past 4 GiB (64-bit offsets without monkeypatching) -- oracle-verified; the
    allocator's high-water mark is the shard's HBM footprint."""
    mix = lm.fib_mix(1 << 22)
    ctx.pool_trim()
    ctx.pool_reset_peak()
    m, top, q, root, proofs, pv = run(ctx, mix, num_queries=100, pow_bits=16, compile_min_log_rows=17)
    assert len(proofs) == 1
    ev = [c for c in proofs[0].chips if m.chips[c.machine_index][:2] == ("func", top.func_index("eval"))]
    assert len(ev) == 1 and ev[0].log_n == 22
    assert proofs[0].log_max_height >= 23
    peak = ctx.pool_stats()["peak_bytes"]
    main_bytes = sum(4 * (c.width << c.log_n) for c in proofs[0].chips)
    assert main_bytes > 4 << 30                      # the main traces alone exceed 4 GiB
    assert 2 * main_bytes < peak < 200 << 30, peak   # traces + LDEs + permutation / quotient rounds resident, inside one MI355X
    print(f"2^22-row shard: main traces {main_bytes / 2**30:.1f} GiB, pool high-water mark {peak / 2**30:.1f} GiB")
    assert prover.grand_sum(proofs) == (0, 0, 0, 0)
    assert os_.verify_machine(oracle_airs(mix, len(pv)), root, [16], [6], proofs, ob.merkle_verify)
    m.close()
    ctx.pool_trim()


@pytest.mark.parametrize("mix", [lm.fib_mix(1 << 13), lm.lurk_mix(1 << 13)], ids=["fib-mix-2^13", "lurk-mix-2^13"])
def test_sparse_permutation_lde_gives_the_dense_proof(ctx, mix, monkeypatch):
    """Round 5: the permutation traces' identically-zero columns are left out of the LDE (only above 2^22 eligible cells by
    default: the threshold is lowered here so that a mid-sized machine with every chip compiled takes the route on its chips of
    2^11 rows and more).  Same proof words as with the route switched off; the oracle's verifier accepts them."""
    top = lair.Toplevel(mix.source, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute_by_name(mix.entry, mix.main_args, q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, mix.entry, len(pv))
    root = m.setup()
    prepared = m.prepare_shard(lair.Shard.new(q))
    assert m.compile_airs(prepared, 11)  # the compiled permutation kernels mark the columns they compute; chips of 2^11 rows and more take the route
    del prepared
    monkeypatch.setenv("LURKHIP_PERM_SPARSE_LDE", "0")
    dense = m.prove(q, num_queries=6, pow_bits=4, parse=False)
    monkeypatch.setenv("LURKHIP_PERM_SPARSE_LDE", "1")
    monkeypatch.setenv("LURKHIP_PERM_SPARSE_MIN_CELLS", "1")
    sparse = m.prove(q, num_queries=6, pow_bits=4, parse=False)
    assert len(dense) == len(sparse) == 1 and np.array_equal(dense[0], sparse[0])
    proofs = [prover.parse_proof(sparse[0])]
    assert m.verify(proofs)
    assert os_.verify_machine(oracle_airs(mix, len(pv)), root, [16], [6], proofs, ob.merkle_verify)
