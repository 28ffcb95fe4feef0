"""The N > 1 path on CPU: two processes over gloo exchange shard roots and reduce cumulative sums exactly as
bench.py does over RCCL, and end up with identical transcripts."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2013265921


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lurk_amd import shards
    from oracle import stark as os_

    n_shards = 6
    mine = shards.assign_shards(n_shards, world, rank)
    # a deterministic fake root / cumulative sum per shard; sums are built to cancel over all shards
    def root_of(s):
        return [(1000 * s + k) % P for k in range(8)]

    sums = {s: (s + 1, 2 * s + 3, P - 5 * s - 1, 7) for s in range(n_shards - 1)}
    total = np.zeros(4, dtype=np.int64)
    for v in sums.values():
        total = (total + np.array(v)) % P
    sums[n_shards - 1] = tuple(int((P - x) % P) for x in total)
    roots = shards.exchange_roots([root_of(s) for s in mine])
    # the balanced assignment (first shards of an execution are the heaviest) with the indices travelling beside the roots
    balanced = shards.assign_shards_balanced([10, 9, 3, 2, 1, 1], world)
    roots_b = shards.exchange_roots([root_of(s) for s in balanced[rank]], shard_indices=balanced[rank])
    assert roots_b == roots and sorted(balanced[0] + balanced[1]) == list(range(6))
    grand = shards.reduce_cumulative_sums([sums[s] for s in mine])
    # the transcript every rank derives from the gathered roots
    ch = os_.Challenger(os_.default_permute16())
    ch.observe([1, 2, 3, 4, 5, 6, 7, 8])
    ch.observe(0)
    for r in roots:
        ch.observe(r)
    # the proofs of all ranks as a set on rank 0, ordered by shard index (shards.gather_proofs)
    words = [np.full(3 + s, s, dtype=np.uint32) for s in balanced[rank]]
    got = shards.gather_proofs(words, balanced[rank], dst=0)
    if rank == 0:
        assert [int(w[0]) for w in got] == list(range(6)) and [len(w) for w in got] == [3 + s for s in range(6)]
    else:
        assert got is None
    q.put((rank, mine, roots, grand, ch.sample_ext()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_root_exchange_and_grand_sum():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    (r0, mine0, roots0, grand0, ch0), (r1, mine1, roots1, grand1, ch1) = results
    assert mine0 == [0, 2, 4] and mine1 == [1, 3, 5]
    assert roots0 == roots1 == [[(1000 * s + k) % P for k in range(8)] for s in range(6)]
    assert grand0 == grand1 == (0, 0, 0, 0)
    assert ch0 == ch1


def _worker_ragged(rank, world, port, q, n_shards):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lurk_amd import shards

    def root_of(s):
        return [(1000 * s + k) % P for k in range(8)]

    # `Shard::shard` cuts an execution into ceil(rows / max_shard_size) shards (execute.rs:186-216): any number.  The first shards hold
    # every chip, the last ones only the tallest: costs fall with the index
    costs = [10 - s for s in range(n_shards)]
    assignment = shards.assign_shards_balanced(costs, world)
    mine = assignment[rank]
    roots = shards.exchange_roots([root_of(s) for s in mine], shard_indices=mine, n_shards=n_shards)
    # a rank that passes a wrong total is told so (every rank alike: nobody is left waiting in a collective)
    try:
        shards.exchange_roots([root_of(s) for s in mine], shard_indices=mine, n_shards=n_shards + 1)
        wrong_total = "accepted"
    except ValueError:
        wrong_total = "refused"
    sums = {s: (s + 1, 2 * s + 3, P - 5 * s - 1, 7) for s in range(n_shards - 1)}
    total = np.zeros(4, dtype=np.int64)
    for v in sums.values():
        total = (total + np.array(v)) % P
    sums[n_shards - 1] = tuple(int((P - x) % P) for x in total)
    grand = shards.reduce_cumulative_sums([sums[s] for s in mine])  # a rank without shards adds nothing
    words = [np.full(3 + s, s, dtype=np.uint32) for s in mine]
    got = shards.gather_proofs(words, mine, dst=0)
    gathered_ok = (rank != 0 and got is None) or (rank == 0 and [int(w[0]) for w in got] == list(range(n_shards)))
    q.put((rank, assignment, roots, grand, wrong_total, gathered_ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_shards", [(2, 5), (3, 7), (3, 2)], ids=["5-shards-2-ranks", "7-shards-3-ranks", "2-shards-3-ranks"])
def test_ragged_shard_counts_over_gloo(world, n_shards):
    """A shard count that is not a multiple of the rank count (missing 3): counts are gathered first, records padded
    to the largest; a rank may hold no shard at all."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, world, port, q, n_shards)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assignment = results[0][1]
    assert sorted(s for a in assignment for s in a) == list(range(n_shards))
    assert max(len(a) for a in assignment) - min(len(a) for a in assignment) <= 1
    for rank, a, roots, grand, wrong_total, gathered_ok in results:
        assert a == assignment
        assert roots == [[(1000 * s + k) % P for k in range(8)] for s in range(n_shards)]
        assert grand == (0, 0, 0, 0) and wrong_total == "refused" and gathered_ok


def _worker_in_flight(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lurk_amd import shards

    groups = [dist.new_group(backend="gloo") for _ in range(2)]  # one per proof lane, created in the same order on every rank

    class FakeStep:
        """The collectives of a RankStep without the GPU work between them: exchange -> (a pause of a different length on every
        rank and lane, so that the lanes' collectives interleave differently on the two ranks) -> reduce."""

        def __init__(self, lane):
            self.lane, self.calls, self.seen = lane, 0, []

        def __call__(self):
            import time

            j = 2 * self.calls + self.lane  # the proof this call is
            mine = [s for s in range(3) if s % world == rank]
            roots = shards.exchange_roots([[(100 * j + 10 * s + k) % P for k in range(8)] for s in mine], shard_indices=mine, n_shards=3, group=groups[self.lane])
            time.sleep(0.002 * ((rank + 1) * (self.lane + 2) + j % 3))
            total = shards.reduce_cumulative_sums([(j + 1, 0, 0, 0)] if rank == 0 else [(P - j - 1, 0, 0, 0)], group=groups[self.lane])
            self.calls += 1
            self.seen.append((j, roots, total))
            return [np.array([j], dtype=np.uint32)]

    steps = [FakeStep(0), FakeStep(1)]
    order = []
    shards.run_in_flight(steps, 9, on_proofs=lambda j, proofs: order.append((j, int(proofs[0][0]))), stagger_s=0.003)
    q.put((rank, [st.seen for st in steps], sorted(order)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_machine_proofs_in_flight_on_two_process_groups():
    """shards.run_in_flight (round 5): two proof lanes per rank, each issuing its collectives on its own process group from its own
    thread; the lanes interleave differently on the two ranks and every proof still sees its own roots and a zero grand sum."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_in_flight, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, seen, order in results:
        assert order == [(j, j) for j in range(9)]
        assert [j for j, _, _ in seen[0]] == [0, 2, 4, 6, 8] and [j for j, _, _ in seen[1]] == [1, 3, 5, 7]
        for lane in seen:
            for j, roots, total in lane:
                assert roots == [[(100 * j + 10 * s + k) % P for k in range(8)] for s in range(3)] and total == (0, 0, 0, 0)


def test_rccl_loader_reports_instead_of_crashing():
    """with no loadable librccl the communicator entry points must return an error with the loader's message (the
    message used to be built from a second dlerror() call, i.e. from a null pointer).  LURKHIP_RCCL_LIB names THE library to use."""
    import subprocess

    code = ("import sys; sys.path.insert(0, %r)\n"
            "from lurk_amd import comm, _native as N\n"
            "import ctypes as C\n"
            "buf = (C.c_uint8 * 128)()\n"
            "st = N.lib.lurkhip_comm_unique_id(C.addressof(buf))\n"
            "print('status', st); print('error', N.last_error(None)); print('library', N.lib.lurkhip_comm_library())\n") % ROOT
    env = dict(os.environ, LURKHIP_RCCL_LIB="/nonexistent/librccl-not-here.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    assert "status -" in r.stdout and "could not be loaded from LURKHIP_RCCL_LIB=/nonexistent/librccl-not-here.so" in r.stdout and "library None" in r.stdout


def test_rccl_loader_reuses_the_copy_the_process_has_mapped():
    """One RCCL per process: once PyTorch has mapped its bundled librccl, the C ABI binds THAT copy, not a second one by name."""
    import subprocess

    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch\n"
            "mapped = sorted({l.split()[-1] for l in open('/proc/self/maps') if '/librccl.so' in l})\n"
            "from lurk_amd import comm\n"
            "print('mapped', mapped); print('bound', comm.library())\n") % ROOT
    env = {k: v for k, v in os.environ.items() if k != "LURKHIP_RCCL_LIB"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    mapped = eval(r.stdout.split("mapped ", 1)[1].splitlines()[0])
    bound = r.stdout.split("bound ", 1)[1].strip()
    if mapped:
        assert bound.startswith(mapped[0]) and "already mapped" in bound
    else:
        assert "librccl" in bound


def test_single_process_paths_need_no_process_group():
    sys.path.insert(0, ROOT)
    from lurk_amd import shards

    assert shards.assign_shards(5, 1, 0) == [0, 1, 2, 3, 4]
    assert shards.assign_shards_balanced([5, 4, 3, 2, 1, 1], 2) == [[0, 3, 4], [1, 2, 5]]  # loads 8 and 8
    assert shards.assign_shards_balanced([1, 1], 1) == [[0, 1]]
    assert shards.assign_shards_balanced([9, 8, 7, 6, 5, 4, 3, 2, 1], 8) == [[0], [1], [2], [3], [4], [5], [6], [7, 8]]  # nine shards, eight GPUs: the lightest two share a rank
    assert shards.assign_shards_balanced([3, 1], 4) == [[0], [1], [], []]
    assert shards.exchange_roots([[2] * 8, [1] * 8], shard_indices=[1, 0]) == [[1] * 8, [2] * 8]
    assert shards.exchange_roots([[1] * 8, [2] * 8]) == [[1] * 8, [2] * 8]
    assert shards.reduce_cumulative_sums([(1, 2, 3, 4), (P - 1, P - 2, P - 3, P - 4)]) == (0, 0, 0, 0)


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """`python bench.py --gpus N` starts its own N ranks; with fewer than N visible GPUs (none here) it must say so and exit
    non-zero instead of printing a one-rank line."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing to run fewer ranks" in r.stderr and "{" not in r.stdout
    # a launcher whose rank count disagrees with --gpus is an error too
    env2 = dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env2,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "must agree" in r.stderr


class _FakePrepared:
    def __init__(self):
        self.closed = False

    def close(self):
        self.closed = True


class _FakeMachine:
    """What shards.scatter_prepared asks of a Machine, without a GPU: a shard's "kernel inputs" are bytes derived from its index."""

    def __init__(self):
        self.prepared = []

    @staticmethod
    def blobs(i):
        return [(0, None)] + [(mi, np.full(1000 * (i + 1) + 17 * mi, (31 * i + mi) % 251, dtype=np.uint8)) for mi in (1, 4, 9)] if i == 0 else \
            [(mi, np.full(1000 * (i + 1) + 17 * mi, (31 * i + mi) % 251, dtype=np.uint8)) for mi in (1, 4)]

    def prepare_shard(self, shard):
        ps = [_FakePrepared() for _ in self.blobs(shard)]
        self.prepared += ps
        return [(mi, None, 0, None, (p if b is not None else None)) for (mi, b), p in zip(self.blobs(shard), ps)]

    def export_prepared(self, prep):
        i = self._current
        return self.blobs(i)


def _scatter_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lurk_amd import shards

    n_shards = 7
    assignment = shards.assign_shards_balanced([10, 9, 8, 3, 2, 1, 1], world)
    m = _FakeMachine()
    # (the fake needs to know which shard it is exporting: scatter_prepared prepares and exports one shard at a time)
    real_prepare = m.prepare_shard

    def prepare(i):
        m._current = i
        return real_prepare(i)

    m.prepare_shard = prepare
    got = shards.scatter_prepared(m, list(range(n_shards)) if rank == 0 else None, assignment, device="cpu", src=0)
    ok = True
    if rank == 0:
        ok = got == {} and all(p.closed for p in m.prepared)  # rank 0 keeps nothing and releases what it prepared for the others
    else:
        ok = sorted(got) == sorted(assignment[rank])
        for i, entries in got.items():
            want = _FakeMachine.blobs(i)
            ok = ok and [mi for mi, _ in entries] == [mi for mi, _ in want]
            for (_, a), (_, b) in zip(entries, want):
                ok = ok and ((a is None and b is None) or (a is not None and b is not None and np.array_equal(a, b)))
    q.put((rank, bool(ok), sorted(got)))
    dist.barrier()
    dist.destroy_process_group()


def test_three_process_scatter_of_prepared_shards():
    """shards.scatter_prepared over gloo: rank 0 "executed", the other two ranks receive exactly the shards the balanced assignment
    deals them (7 shards on 3 ranks, shard 0 with its entrypoint entry), byte for byte; rank 0 keeps nothing of theirs."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scatter_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in out), out
    assert sorted(i for _, _, mine in out[1:] for i in mine) + [] == sorted(set(range(7)) - set(_assignment0()))


def _assignment0():
    sys.path.insert(0, ROOT)
    from lurk_amd import shards

    return shards.assign_shards_balanced([10, 9, 8, 3, 2, 1, 1], 3)[0]
