"""The RCCL collectives behind the C ABI (csrc/comm.cpp: lurkhip_comm_*, lurkhip_exchange_roots*, lurkhip_reduce_sums*).

RCCL refuses two ranks on one device, and the GPU box of the test tier has one: these tests run the real RCCL calls with a world
of ONE rank (communicator set-up through dlopen'ed librccl, all-gather, all-reduce, the device-side record and lane layouts,
the partition and canonicity checks); tests/test_distributed.py covers world 2 over gloo on the CPU, through the same
`RankStep` that takes the communicator here, and the driver's multi-GPU bench is the first run with world > 1 on RCCL."""
import numpy as np
import pytest

import lurk_amd
from lurk_amd import comm as cm
from lurk_amd import field

pytestmark = pytest.mark.gpu
P = field.P


@pytest.fixture()
def comm(ctx):
    c = cm.Comm(ctx, cm.unique_id(), 0, 1)
    yield c
    c.close()


def test_exchange_roots_sorts_by_shard_index_and_checks_the_partition(ctx, comm):
    rng = np.random.default_rng(5)
    roots = rng.integers(0, P, size=(4, 8), dtype=np.uint32)
    got = comm.exchange_roots([2, 0, 3, 1], roots)
    assert got == [[int(x) for x in roots[i]] for i in (1, 3, 0, 2)]
    with pytest.raises(lurk_amd.LurkHipError, match="partition"):
        comm.exchange_roots([0, 0, 1, 2], roots)
    with pytest.raises(lurk_amd.LurkHipError, match="partition"):
        comm.exchange_roots([0, 1, 2, 7], roots)


def test_reduce_sums_adds_lanes_mod_p(ctx, comm):
    rng = np.random.default_rng(6)
    sums = rng.integers(0, P, size=(37, 4), dtype=np.uint32)
    want = tuple(int(sums[:, c].astype(np.uint64).sum() % P) for c in range(4))
    assert comm.reduce_sums(sums) == want
    assert comm.reduce_sums(np.zeros((0, 4), dtype=np.uint32)) == (0, 0, 0, 0)
    bad = sums.copy()
    bad[3, 1] = P
    with pytest.raises(lurk_amd.LurkHipError, match="canonical"):
        comm.reduce_sums(bad)


def test_device_entry_points(ctx, comm):
    import torch

    rec = torch.arange(3 * cm.RECORD_WORDS, dtype=torch.int32, device="cuda")
    out = torch.zeros_like(rec)
    comm.exchange_roots_dev(rec, 3, out)
    lanes = torch.tensor([5, P + 7, 3 * P, (1 << 40) + 1], dtype=torch.int64, device="cuda")
    total = torch.zeros(4, dtype=torch.int32, device="cuda")
    comm.reduce_sums_dev(lanes, total)
    ctx.sync()
    assert torch.equal(out, rec)
    assert total.cpu().numpy().view(np.uint32).tolist() == [5, 7, 0, ((1 << 40) + 1) % P]


def test_rank_step_through_the_c_abi_collectives(ctx):
    """One rank, two shards: the bench's step (shards.RankStep) with its exchange and its check behind the C ABI gives the proofs and
    the grand sum of the torch / single-process path."""
    from lurk_amd import lair, prover, shards
    from lurk_amd.programs import lurk_mix as lm

    mix = lm.fib_mix(2 << 7)
    top = lair.Toplevel(mix.source, lurk_chips=True)
    queries = lair.QueryRecord(top)
    top.execute_by_name(mix.entry, mix.main_args, queries)
    pv = queries.expect_public_values()
    machine = prover.Machine(ctx, top, mix.entry, len(pv))
    vk = machine.setup()
    all_shards = lair.Shard.new(queries).shard(lair.ShardingConfig(1 << 7))
    assert len(all_shards) == 2
    mine = list(range(len(all_shards)))
    prepared = [machine.prepare_shard(sh) for sh in all_shards]
    c = cm.Comm(ctx, cm.unique_id(), 0, 1)
    try:
        with_comm = shards.RankStep(machine, vk, pv, prepared, mine, 8, 6, device="cpu", comm=c)
        plain = shards.RankStep(machine, vk, pv, prepared, mine, 8, 6, device="cpu")
        a, b = with_comm(), plain()
        assert [w.tolist() for w in a] == [w.tolist() for w in b]
        assert with_comm.roots == plain.roots
        assert with_comm.grand_sums == plain.grand_sums == [(0, 0, 0, 0)]
    finally:
        c.close()


def test_split_vtable_callbacks_at_world_one(ctx, comm):
    """The carrier of the split prover on RCCL (csrc/comm.cpp: lurkhip_comm_split_vtable -- grouped ncclSend / ncclRecv pairs, a device
    all-gather, host all-gather and 64-bit all-reduce staged through the pool): the four callbacks called through the struct the
    library is given, with the one rank this box's RCCL accepts.  What it can show here: librccl's ncclSend / ncclRecv /
    ncclGroupStart / ncclGroupEnd resolve, an empty group is accepted, the own block of the all-to-all lands, the data types are
    the 32- and 64-bit unsigned ones."""
    import ctypes as C

    import torch

    from lurk_amd import split

    sc = split.RcclSplitComm(ctx, comm)
    st = sc.struct
    assert (st.rank, st.world) == (0, 1)
    n = 1000
    send = torch.arange(n, dtype=torch.int32, device="cuda") * 3 + 1
    recv = torch.zeros(n + 5, dtype=torch.int32, device="cuda")
    soff = (C.c_uint64 * 2)(0, n)
    roff = (C.c_uint64 * 2)(5, n + 5)  # (the receive offsets are the receiver's own: the block lands behind five words)
    torch.cuda.synchronize()
    assert st.alltoallv_dev(st.user, send.data_ptr(), soff, recv.data_ptr(), roff, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(recv[5:], send) and int(recv[:5].abs().sum()) == 0
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    assert st.allgather_dev(st.user, send.data_ptr(), out.data_ptr(), n, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, send)
    src = bytes(range(37))
    dst = C.create_string_buffer(37)
    assert st.allgather_host(st.user, src, dst, 37) == 0  # (a length that is no multiple of four: padded inside)
    assert dst.raw == src
    lanes = (C.c_uint64 * 3)(5, (1 << 63) + 9, 0xFFFFFFFF)
    assert st.allreduce_sum_u64_host(st.user, lanes, 3) == 0
    assert list(lanes) == [5, (1 << 63) + 9, 0xFFFFFFFF]
    ctx.sync()
    assert sc.selftest() is None  # (what bench.py --split intra runs on every rank before it trusts the carrier)
