"""ONE shard proved by G ranks together produces the words one GPU produces (second bullet; include/lurkhip.h:
lurkhip_*_split; lurk_amd/csrc/split.hip, split_plan.h).

`Shard::shard` (/root/reference/src/lair/execute.rs:186-241) leaves anything below 2^22 rows in one shard, so behind
`machine.prove` (/root/reference/benches/fib.rs:124) seven of eight GPUs would idle.  Here G = 2, 4, 8 processes share the one
GPU of the test box (gloo carries the collectives through host memory: RCCL refuses two ranks on one device; on a multi-GPU node
`RcclSplitComm` keeps the blocks on the devices) and every rank must end with: the verifying key's root, the main root and the
WHOLE proof -- openings, FRI, query answers -- word for word equal to the one-rank prover's, which both verifiers accept."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, spec):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import numpy as np
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
        import lurk_amd
        from lurk_amd import lair, prover, split
        from lurk_amd.programs import lurk_mix as lm

        mix = (lm.fib_mix if spec["mix"] == "fib" else lm.lurk_mix)(spec["rows"])
        ctx = lurk_amd.Context(0)
        top = lair.Toplevel(mix.source, lurk_chips=True)
        qr = lair.QueryRecord(top)
        top.execute_by_name(mix.entry, mix.main_args, qr)
        pv = qr.expect_public_values()
        m = prover.Machine(ctx, top, mix.entry, len(pv))
        vk_root = m.setup()
        prepared = m.prepare_shard(lair.Shard.new(qr))
        if spec.get("compile") is not None:
            m.compile_airs(prepared, spec["compile"])
        traces = m.run_prepared(prepared)
        ctx.sync()
        nq, pow_bits = spec.get("queries", 8), spec.get("pow", 6)
        # one rank
        ch = prover.Challenger(ctx)
        ch.observe(vk_root)
        ch.observe([0])
        handle, root = m.commit_shard(traces)
        ch.observe(root)
        ch.observe(pv)
        ref = m.prove_shard(handle, ch, pv, nq, pow_bits, parse=False)
        m.free_shard(handle)
        # all ranks together
        comm = split.TorchSplitComm(ctx)
        sp = split.SplitProver(m, comm, spec["min_log_n"])
        out = {"rank": rank, "vk_equal": sp.setup() == vk_root}
        words, root2 = sp.prove(traces, pv, nq, pow_bits)
        words2, _ = sp.prove(traces, pv, nq, pow_bits)  # (pooled buffers come back in another order: same words)
        out["root_equal"] = root2 == root
        out["len"] = (int(len(words)), int(len(ref)))
        same = len(words) == len(ref) and bool((words == ref).all())
        out["words_equal"] = same
        out["again_equal"] = len(words2) == len(ref) and bool((words2 == ref).all())
        if not same:
            n = min(len(words), len(ref))
            diff = np.nonzero(words[:n] != ref[:n])[0]
            out["first_diff"] = int(diff[0]) if len(diff) else n
            out["n_diff"] = int(len(diff))
            p = prover.parse_proof(ref)
            out["layout"] = {"n_chips": len(p.chips), "header_end": 10 + 11 * len(p.chips) + len(pv), "log_n": [c.log_n for c in p.chips]}
        out["calls"] = dict(comm.calls)
        out["alltoall_bytes"] = comm.alltoall_bytes
        if rank == 0:
            out["verified"] = bool(m.verify([words]))
            if spec.get("oracle"):
                from test_workloads_gpu import oracle_airs
                from oracle import binding as ob
                from oracle import stark as os_

                out["oracle_verified"] = bool(os_.verify_machine(oracle_airs(mix, len(pv)), vk_root, [16], [6], [prover.parse_proof(words)], ob.merkle_verify))
        q.put(out)
        dist.barrier()
        sp.close()
        del prepared, traces
        m.close()
        ctx.close()
        dist.destroy_process_group()
    except BaseException as e:  # surface the failure instead of a queue timeout
        import traceback

        q.put({"rank": rank, "error": traceback.format_exc()})
        raise e


def _run(world, spec, timeout=600):
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, q, spec)) for r in range(world)]
    for p in procs:
        p.start()
    outs = []
    try:
        for _ in range(world):
            outs.append(q.get(timeout=timeout))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    errors = [o["error"] for o in outs if "error" in o]
    assert not errors, errors[0]
    return sorted(outs, key=lambda o: o["rank"])


def _check(outs):
    for o in outs:
        assert o["vk_equal"], o
        assert o["root_equal"], o
        assert o["words_equal"] and o["again_equal"], {k: v for k, v in o.items() if k != "calls"}
    assert outs[0]["verified"]
    if "oracle_verified" in outs[0]:
        assert outs[0]["oracle_verified"]
    # one all-to-all per commitment the traces of which every rank holds (key, main), two for the permutation and quotient commitments
    assert outs[0]["calls"]["alltoallv"] == 1 + 2 * (1 + 2 + 2)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_fib_mix_split_equals_one_rank(world):
    """fib-mix at 2^12 eval rows on the interpreter kernels: chips from 2^6 rows up are cut, so a dozen heights take part in both
    exchanges; the entrypoint chip (one row: an LDE of two) enters the tree above the ranks' subtrees for G = 4 and 8."""
    _check(_run(world, {"mix": "fib", "rows": 1 << 12, "min_log_n": 6, "oracle": world == 2}))


@pytest.mark.parametrize("world", [2, 8])
def test_fib_mix_2p16_compiled_split_equals_one_rank(world):
    """The verdict's case: fib-mix at 2^16 eval rows with the chips' kernels compiled (the quotient's staged tiles, the row-range
    arguments of stark_kernels.h through hiprtc), cut from 2^10 rows up."""
    _check(_run(world, {"mix": "fib", "rows": 1 << 16, "min_log_n": 10, "compile": 0, "queries": 16, "pow": 8}, timeout=1200))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_lurk_mix_2p13_split_equals_one_rank(world):
    """lurk-mix (all 39 chips, the three hashers, six memory tables) at 2^13 eval rows."""
    _check(_run(world, {"mix": "lurk", "rows": 1 << 13, "min_log_n": 5, "oracle": world == 4}))
