"""The index arithmetic of a commitment made by G ranks together, alone and on the host (lurk_amd/csrc/split_plan.h through
lurkhip_split_plan): ragged column counts -- widths not divisible by G, height groups with fewer columns than ranks --, the three
kinds of row sources, next-row copies.  The two exchanges are run with numpy on synthetic matrices: inside one process for
G = 2, 4, 8, and over gloo (real sends and receives between processes) for G = 2 and 4.

What the plan replaces: nothing in the reference moves data between devices -- `machine.prove` (/root/reference/benches/fib.rs:124)
proves a shard in one process; SURVEY.md 8e (second bullet) is the partitioning this implements."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from lurk_amd import split

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

K_FULL, K_BLOCK, K_QUOTIENT = 0, 1, 2

# (log_n, width, kind, lqd, chunk, n_next): two tall groups with ragged widths, a group narrower than the ranks, quotient chunks of
# both degrees, and matrices below the cut
MATS = [
    (7, 78, K_BLOCK, 0, 0, 0),      # a permutation trace with dead batch columns (RUNS below): only the live ones are exchanged
    (7, 13, K_BLOCK, 0, 0, 0),
    (4, 3, K_BLOCK, 0, 0, 0),       # a group of fewer columns than four ranks: some ranks transform nothing of it
    (6, 4, K_QUOTIENT, 1, 0, 0),
    (6, 4, K_QUOTIENT, 1, 1, 0),
    (5, 4, K_QUOTIENT, 0, 0, 0),    # a chip of constraint degree 2: one chunk, quotient domain = the low coset
    (5, 9, K_FULL, 0, 0, 2),        # e.g. a memory chip's main trace: is_real and ptr are read on the next row
    (7, 5, K_FULL, 0, 0, 1),
    (2, 6, K_FULL, 0, 0, 0),        # below the cut
    (0, 11, K_FULL, 0, 0, 0),
]
MIN_LOG_N = 4
# per matrix: the (first column, width) runs that are not identically zero on every rank; []: all columns
RUNS = [[(0, 5), (9, 20), (40, 38)], [(12, 1)], [], [], [], [], [], [], [], []]


def live_mask(i):
    m = np.zeros(MATS[i][1], dtype=bool)
    for c0, w in RUNS[i] or [(0, MATS[i][1])]:
        m[c0:c0 + w] = True
    return m


def brev(x, bits):
    return int(format(x, f"0{bits}b")[::-1], 2) if bits else 0


def full_matrix(i):
    log_n, w = MATS[i][0], MATS[i][1]
    return ((np.arange((1 << log_n) * w, dtype=np.uint32).reshape(1 << log_n, w) * 7 + 1000003 * i) & 0x7FFFFFFF) * live_mask(i).astype(np.uint32)


def fake_lde(i):
    """any [2N][w] matrix stands for the LDE in exchange B, which only moves data"""
    log_n, w = MATS[i][0], MATS[i][1]
    return ((np.arange((2 << log_n) * w, dtype=np.uint32).reshape(2 << log_n, w) * 13 + 777 * i + 5) & 0x7FFFFFFF) * live_mask(i).astype(np.uint32)


def local_rows(i, rank, log_g):
    """what rank holds of matrix i before exchange A, as split.hip is given it"""
    log_n, w, kind, lqd, chunk, _ = MATS[i]
    full = full_matrix(i)
    G = 1 << log_g
    if kind == K_FULL:
        return full
    if kind == K_BLOCK:
        rows = (1 << log_n) >> log_g
        return full[rank * rows:(rank + 1) * rows]
    # K_QUOTIENT: the quotient kernel's output for the rank's storage rows, row brev(j) of the local buffer = storage row j
    log_l2 = log_n + 1 - log_g
    out = np.zeros((1 << log_l2, w), dtype=np.uint32)
    log_q = log_n + lqd
    held = False
    for j in range(1 << log_l2):
        s = (rank << log_l2) + j
        if s >= (1 << log_q):
            continue
        i_nat = brev(s, log_q)
        if i_nat & ((1 << lqd) - 1) != chunk:
            continue
        held = True
        out[brev(j, log_l2)] = full[i_nat >> lqd]
    return out if held else None


def args():
    return dict(log_heights=[m[0] for m in MATS], widths=[m[1] for m in MATS], kinds=[m[2] for m in MATS], lqds=[m[3] for m in MATS],
                chunks=[m[4] for m in MATS], n_next=[m[5] for m in MATS], runs=RUNS)


def run_pack(jobs, bufs, lin):
    for j in jobs:
        b = bufs[j["buf"]]
        for r in range(j["rows"]):
            lin[j["lin_off"] + r * j["lin_pitch"]: j["lin_off"] + r * j["lin_pitch"] + j["width"]] = b[j["row0"] + r * j["row_stride"], j["col0"]: j["col0"] + j["width"]]


def run_unpack(jobs, bufs, lin):
    for j in jobs:
        b = bufs[j["buf"]]
        for r in range(j["rows"]):
            b[j["row0"] + r * j["row_stride"], j["col0"]: j["col0"] + j["width"]] = lin[j["lin_off"] + r * j["lin_pitch"]: j["lin_off"] + r * j["lin_pitch"] + j["width"]]


def check_rank(plan, rank, log_g, slabs, locals_):
    """the rank's slabs hold its column tile of the exchanged matrices, its row blocks every column (and next-row copy) of its storage rows"""
    G = 1 << log_g
    for gi, g in enumerate(plan["groups"]):
        lo, hi = g["bounds"][rank], g["bounds"][rank + 1]
        assert g["bounds"][0] == 0 and g["bounds"][-1] == g["W"] and all(a <= b for a, b in zip(g["bounds"], g["bounds"][1:]))
        vstart = 0  # the virtual row the tiles are cut from: the live runs of the group's matrices, side by side
        for mat, col_start in g["mats"]:
            w, kind = MATS[mat][1], MATS[mat][2]
            below = MATS[mat][0] < MIN_LOG_N
            for c0, rw in (RUNS[mat] if RUNS[mat] and not below else [(0, w)]):
                a, b = max(lo, vstart), min(hi, vstart + rw)
                if a < b and kind != K_FULL:
                    assert np.array_equal(slabs[gi][:, a - lo:b - lo], full_matrix(mat)[:, c0 + a - vstart:c0 + b - vstart]), (rank, gi, mat)
                vstart += rw
            l2 = (2 << g["log_n"]) >> log_g
            assert np.array_equal(locals_[gi][:, col_start:col_start + w], fake_lde(mat)[rank * l2:(rank + 1) * l2]), (rank, gi, mat)
        assert vstart == g["W"] and g["W"] <= g["W_local"] and g["sparse"] == (g["W"] < g["W_local"])
        for e, (mat, col, vcol, owner) in enumerate(g["extras"]):
            l2 = (2 << g["log_n"]) >> log_g
            assert g["bounds"][owner] <= vcol < g["bounds"][owner + 1]
            assert np.array_equal(locals_[gi][:, g["W_local"] + e], next_copy(mat, col)[rank * l2:(rank + 1) * l2]), (rank, gi, mat, col)
        assert g["local_pitch"] % 32 == 0 and g["local_pitch"] >= g["W_local"] + len(g["extras"])


def next_copy(mat, col):
    """stands for split.hip's k_next_rows output: any function of (matrix, column) the owner of the column can compute"""
    return (fake_lde(mat)[:, col] ^ 0x5A5A5A5A) & 0x7FFFFFFF


def prepare_rank(plan, rank, log_g):
    """this rank's buffers: local sources for exchange A, and (after the stand-in LDE) its tiles and next-row copies for exchange B"""
    src = [local_rows(i, rank, log_g) for i in range(len(MATS))]
    a_send = np.zeros(plan["a_send_off"][-1] if plan["has_a"] else 0, dtype=np.uint32)
    if plan["has_a"]:
        run_pack(plan["a_pack"], src, a_send)
    tiles = [np.ascontiguousarray(fake_lde(t["mat"])[:, t["c0"]:t["c0"] + t["w"]]) for t in plan["tiles"]]
    for e in plan["my_extras"]:
        mat, col, _, owner = plan["groups"][e >> 16]["extras"][e & 0xFFFF]
        assert owner == rank
        tiles.append(next_copy(mat, col).reshape(-1, 1).copy())
    b_send = np.zeros(plan["b_send_off"][-1], dtype=np.uint32)
    run_pack(plan["b_pack"], tiles, b_send)
    return a_send, b_send


def finish_rank(plan, rank, log_g, a_recv, b_recv):
    slabs = [np.zeros((1 << g["log_n"], max(g["slab_w"], 1)), dtype=np.uint32) for g in plan["groups"]]
    if plan["has_a"]:
        run_unpack(plan["a_unpack"], slabs, a_recv)
    locals_ = [np.zeros(((2 << g["log_n"]) >> log_g, g["local_pitch"]), dtype=np.uint32) for g in plan["groups"]]
    run_unpack(plan["b_unpack"], locals_, b_recv)
    check_rank(plan, rank, log_g, slabs, locals_)


@pytest.mark.parametrize("log_g", [1, 2, 3])
def test_exchanges_in_one_process(log_g):
    G = 1 << log_g
    plans = [split.split_plan(G, r, MIN_LOG_N, **args()) for r in range(G)]
    sends = [prepare_rank(plans[r], r, log_g) for r in range(G)]
    for which, key in ((0, "a"), (1, "b")):
        for d in range(G):
            # what every rank sends to d arrives where d expects it: the senders' and the receiver's offsets must agree block by block
            so = [plans[s][f"{key}_send_off"] for s in range(G)]
            ro = plans[d][f"{key}_recv_off"]
            if not so[0]:
                continue
            for s in range(G):
                assert so[s][d + 1] - so[s][d] == ro[s + 1] - ro[s], (key, s, d)
    for d in range(G):
        a_recv = np.concatenate([sends[s][0][plans[s]["a_send_off"][d]:plans[s]["a_send_off"][d + 1]] for s in range(G)]) if plans[d]["has_a"] else None
        b_recv = np.concatenate([sends[s][1][plans[s]["b_send_off"][d]:plans[s]["b_send_off"][d + 1]] for s in range(G)])
        finish_rank(plans[d], d, log_g, a_recv, b_recv)
    # the groups are the cut heights, tallest first; matrices below the cut are in none
    assert [g["log_n"] for g in plans[0]["groups"]] == [7, 6, 5, 4]
    assert sorted(m for g in plans[0]["groups"] for m, _ in g["mats"]) == list(range(8))
    # the narrow group leaves ranks without a column when there are more ranks than columns
    narrow = plans[0]["groups"][3]
    assert narrow["W"] == 3 and (G < 4 or any(a == b for a, b in zip(narrow["bounds"], narrow["bounds"][1:])))


def test_refusals():
    with pytest.raises(Exception):
        split.split_plan(3, 0, 4, **args())  # not a power of two
    with pytest.raises(Exception):
        split.split_plan(8, 0, 2, **args())  # cut below the number of ranks
    with pytest.raises(Exception):
        split.split_plan(4, 4, 4, **args())  # rank outside the world


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
        import test_split_plan as T

        log_g = world.bit_length() - 1
        plan = split.split_plan(world, rank, MIN_LOG_N, **T.args())
        a_send, b_send = T.prepare_rank(plan, rank, log_g)

        def alltoall(send, soff, roff):
            ins = [torch.from_numpy(send[soff[d]:soff[d + 1]].astype(np.int64)) for d in range(world)]
            outs = [torch.empty(roff[s + 1] - roff[s], dtype=torch.int64) for s in range(world)]
            reqs = []
            for k in range(1, world):
                to, frm = (rank + k) % world, (rank - k) % world
                reqs.append(dist.isend(ins[to], to))
                reqs.append(dist.irecv(outs[frm], frm))
            outs[rank].copy_(ins[rank])
            for r in reqs:
                r.wait()
            return np.concatenate([o.numpy() for o in outs]).astype(np.uint32)

        a_recv = alltoall(a_send, plan["a_send_off"], plan["a_recv_off"]) if plan["has_a"] else None
        b_recv = alltoall(b_send, plan["b_send_off"], plan["b_recv_off"])
        T.finish_rank(plan, rank, log_g, a_recv, b_recv)
        q.put((rank, "ok", int(plan["b_send_off"][-1])))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        import traceback

        q.put((rank, traceback.format_exc(), 0))
        raise


@pytest.mark.parametrize("world", [2, 4])
def test_exchanges_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(o[1] == "ok" for o in outs), [o[1] for o in outs if o[1] != "ok"][0]
