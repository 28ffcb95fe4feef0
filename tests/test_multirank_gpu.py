"""The N > 1 path with real proofs: TWO processes share the one GPU of the test box (gloo carries the two tiny collectives --
RCCL refuses two ranks on one device; on an 8-GPU node the same code runs over backend "nccl"), run `shards.RankStep` -- the
body of bench.py's timed step -- on ONE fib-mix execution cut into 4 shards and dealt by work, send every rank's proofs to
rank 0, and the oracle's machine verifier accepts the GATHERED set: per-rank sums non-zero, total zero, main roots observed in
shard order whatever the assignment (/root/reference/src/lair/execute.rs:186-241, lair_chip.rs:104-139).
Also: `python bench.py --gpus 2` on a box with one GPU must refuse loudly, and `--oversubscribe` must run both ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG_SHARD = 10  # 4 shards of 2^10 eval rows


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
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
        import lurk_amd
        from lurk_amd import lair, prover, shards
        from lurk_amd.programs import lurk_mix as lm

        mix = lm.fib_mix(4 << LOG_SHARD)
        ctx = lurk_amd.Context(0)
        top = lair.Toplevel(mix.source, lurk_chips=True)
        qr = lair.QueryRecord(top)
        top.execute_by_name(mix.entry, mix.main_args, qr)
        pv = qr.expect_public_values()
        m = prover.Machine(ctx, top, mix.entry, len(pv))
        vk_root = m.setup()
        all_shards = lair.Shard.new(qr).shard(lair.ShardingConfig(1 << LOG_SHARD))
        assert len(all_shards) == 4
        assignment = shards.assign_shards_balanced([m.shard_cost(sh) for sh in all_shards], world)
        mine = assignment[rank]
        prepared = [m.prepare_shard(all_shards[i]) for i in mine]
        lane_ctx = prover.lane_context(m)
        step = shards.RankStep(m, vk_root, pv, prepared, mine, num_queries=8, pow_bits=6, device="cpu", lane_ctx=lane_ctx)
        words = step()
        words2 = step()  # a second step gives the same proofs
        same = all(len(a) == len(b) and bool((a == b).all()) for a, b in zip(words, words2))
        # the pipelined schedule of the bench's N > 1 path (shards.run_pipelined): a second machine on its own contexts, phase 1 of
        # proof j + 1 under phase 2 of proof j, collectives on this thread only -- same proofs
        ctx_b = lurk_amd.Context(0)
        m_b = prover.Machine(ctx_b, top, mix.entry, len(pv))
        assert m_b.setup() == vk_root
        prepared_b = [m_b.prepare_shard(all_shards[i]) for i in mine]
        lane_b = prover.lane_context(m_b)
        step_b = shards.RankStep(m_b, vk_root, pv, prepared_b, mine, num_queries=8, pow_bits=6, device="cpu", lane_ctx=lane_b)
        piped = []
        shards.run_pipelined([step, step_b], 3, on_proofs=lambda j, pr: piped.append(pr))
        same = same and len(piped) == 3 and all(len(a) == len(b) and bool((a == b).all()) for pr in piped for a, b in zip(pr, words))
        same = same and step_b.grand_sums[-1] == (0, 0, 0, 0) and step_b.rank_sums[-1] == step.rank_sums[-1]
        # ... and with phase 1 running ahead on its own thread over three machines (shards.run_committed_ahead)
        ctx_c = lurk_amd.Context(0)
        m_c = prover.Machine(ctx_c, top, mix.entry, len(pv))
        assert m_c.setup() == vk_root
        prepared_c = [m_c.prepare_shard(all_shards[i]) for i in mine]
        step_c = shards.RankStep(m_c, vk_root, pv, prepared_c, mine, num_queries=8, pow_bits=6, device="cpu", lane_ctx=prover.lane_context(m_c))
        ahead = []
        shards.run_committed_ahead([step, step_b, step_c], 5, on_proofs=lambda j, pr: ahead.append(pr))
        same = same and len(ahead) == 5 and all(len(a) == len(b) and bool((a == b).all()) for pr in ahead for a, b in zip(pr, words))
        same = same and step_c.grand_sums[-1] == (0, 0, 0, 0)
        gathered = shards.gather_proofs(piped[-1], mine, dst=0)
        out = {"rank": rank, "mine": mine, "rank_sum": step.rank_sums[-1], "grand": step.grand_sums[-1], "same": same, "roots": step.roots}
        if rank == 0:
            from test_workloads_gpu import oracle_airs
            from oracle import binding as ob
            from oracle import stark as os_

            proofs = [prover.parse_proof(w) for w in gathered]
            out["n_gathered"] = len(proofs)
            out["shard_sums"] = [prover.shard_sum(p) for p in proofs]
            out["verified"] = bool(os_.verify_machine(oracle_airs(mix, len(pv)), vk_root, [16], [6], proofs, ob.merkle_verify))
            # the set is order-sensitive: swapping two shards' proofs must break the transcript
            swapped = [proofs[1], proofs[0]] + proofs[2:]
            try:
                os_.verify_machine(oracle_airs(mix, len(pv)), vk_root, [16], [6], swapped, ob.merkle_verify)
                out["swapped_rejected"] = False
            except os_.VerifyError:
                out["swapped_rejected"] = True
            # the product's own verifier on the same gathered set, and on the swapped one
            out["verified"] = out["verified"] and bool(m.verify(gathered))
            try:
                m.verify([gathered[1], gathered[0]] + list(gathered[2:]))
                out["swapped_rejected"] = False
            except prover.VerificationError:
                pass
        q.put(out)
        dist.barrier()
        del prepared, prepared_b
        lane_b.close()
        m_b.close()
        ctx_b.close()
        lane_ctx.close()
        m.close()
        ctx.close()
        dist.destroy_process_group()
    except BaseException as e:  # surface the failure instead of a queue timeout
        import traceback

        q.put({"rank": rank, "error": traceback.format_exc()})
        raise e


def test_two_ranks_on_one_gpu_prove_one_execution_and_the_set_verifies():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for r in results:
        assert "error" not in r, r.get("error")
    results.sort(key=lambda r: r["rank"])
    r0, r1 = results
    assert sorted(r0["mine"] + r1["mine"]) == [0, 1, 2, 3] and len(r0["mine"]) == len(r1["mine"]) == 2
    assert 0 in r0["mine"] or 0 in r1["mine"]
    assert r0["roots"] == r1["roots"] and len(r0["roots"]) == 4       # same transcript prefix on both ranks
    assert r0["rank_sum"] != (0, 0, 0, 0) and r1["rank_sum"] != (0, 0, 0, 0)  # only the total cancels
    assert r0["grand"] == r1["grand"] == (0, 0, 0, 0)
    assert r0["same"] and r1["same"]
    assert r0["n_gathered"] == 4 and all(s != (0, 0, 0, 0) for s in r0["shard_sums"])
    assert r0["verified"] and r0["swapped_rejected"]
    for p in procs:
        assert p.exitcode == 0


def _bench(*extra, timeout=1500):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--log-rows", "12", "--no-cpu-baseline", "--queries", "8",
           "--pow-bits", "6", "--no-compile"] + list(extra)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_refuses_more_ranks_than_gpus_and_runs_them_oversubscribed():
    import torch

    n = torch.cuda.device_count()
    r = _bench("--gpus", str(n + 1))
    assert r.returncode != 0 and "refusing to run fewer ranks" in (r.stderr + r.stdout)
    r = _bench("--gpus", "2", "--oversubscribe", "--rank-pipeline") if n < 2 else _bench("--gpus", "2", "--rank-pipeline")
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["ranks"] == 2 and line["config"]["shards"] == 4
    cfg = line["config"]
    assert cfg["rank_pipeline"]
    assert cfg["grand_sum_is_zero"] and cfg["per_rank_sum_nonzero"] and cfg["proofs_identical_across_steps"]
    gs = cfg["gathered_proof_set"]
    assert gs["shards_gathered_on_rank0"] == 4 and gs["main_roots_match_exchanged_roots_in_shard_order"] and gs["grand_sum_of_gathered_proofs_is_zero"]
    if n < 2:
        assert line["n_gpus"] == n and "oversubscribed" in cfg["process_group"]
    else:
        assert cfg["rccl_world_size"] == 2


def test_bench_eight_ranks_nine_shards_rehearsal():
    """The first 8-GPU run, rehearsed: `bench.py --gpus 8` as the driver starts it, eight ranks on the
    devices present (oversubscribed over gloo when there are fewer than eight), ONE execution cut into NINE shards -- not a multiple
    of the rank count: the ranks hold one or two shards, the root exchange gathers the counts first --, two machine proofs in flight
    per rank on two process groups.  The gathered proof set must be a partition with matching roots and a zero grand sum, and
    every rank's host seconds and peak resident set are in the line."""
    import torch

    n = torch.cuda.device_count()
    extra = ["--gpus", "8", "--shards-total", "9", "--log-rows", "13"] + (["--oversubscribe"] if n < 8 else [])
    r = _bench(*extra, timeout=2400)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert line["ranks"] == 8 and cfg["shards"] == 9 and cfg["rank_proofs_in_flight"] == 2
    assert sorted(s for a in cfg["shard_assignment"] for s in a) == list(range(9)) and sorted(len(a) for a in cfg["shard_assignment"]) == [1] * 7 + [2]
    assert cfg["grand_sum_is_zero"] and cfg["per_rank_sum_nonzero"] and cfg["proofs_identical_across_steps"]
    gs = cfg["gathered_proof_set"]
    assert gs["shards_gathered_on_rank0"] == 9 and gs["main_roots_match_exchanged_roots_in_shard_order"] and gs["grand_sum_of_gathered_proofs_is_zero"]
    assert len(cfg["host_execute_s_per_rank"]) == 8 and len(cfg["end_to_end"]["peak_rss_mb_per_rank"]) == 8 and min(cfg["end_to_end"]["peak_rss_mb_per_rank"]) > 100


def test_bench_n_ranks_also_probe_the_intra_shard_split(monkeypatch):
    """Round 6: with N > 1 the bench's line also carries `config.split_intra` -- the same ranks proving ONE shard together
    (`--split intra`), measured by child processes with their own process group on another port and a time limit, so that the split's
    first contact with a multi-GPU box cannot take the driver's line with it.  Rehearsed here with the ranks sharing the box's device
    (LURKHIP_SPLIT_PROBE_OVERSUB=1): the children rendezvous under torchrun's environment, every rank ends with the same verified proof,
    the bytes of the all-to-alls are in the line."""
    import torch

    n = torch.cuda.device_count()
    monkeypatch.setenv("LURKHIP_SPLIT_PROBE_OVERSUB", "1")
    r = _bench("--gpus", "2", "--log-rows", "13", *(["--oversubscribe"] if n < 2 else []), timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    si = line["config"]["split_intra"]
    assert si is not None and "error" not in si, si
    assert si["scaling"] == "strong" and si["n_gpus"] == 2 and si["proofs_identical_on_all_ranks"] and si["proof_verified"]
    assert si["alltoall_bytes_per_rank_per_step"] > 0 and len(si["per_rank"]) == 2 and si["per_rank"][1]["alltoalls"] == 6
    assert line["scaling"] == "weak"  # (the line itself is still the shards -> ranks measurement)


def test_bench_split_turns_times_each_ranks_work_with_the_device_to_itself():
    """`--split intra --oversubscribe --split-turns`: ranks that share the box's device take turns between the collectives (a token goes
    round, lurk_amd/split.py: TorchSplitComm.begin_turns), so that a rank's segments are timed alone; the line carries `predicted`: the sum
    over the segments of the slowest rank, the collectives at the xGMI link rate, the one-rank prover timed in the same process.  Every
    rank went through the same collectives, the proofs are still the one-rank words."""
    r = _bench("--gpus", "2", "--oversubscribe", "--split", "intra", "--split-turns", "--log-rows", "13", "--split-min-log-rows", "8", "--steps", "2")
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["proofs_identical_on_all_ranks"] and line["proof_verified"]
    p = line["predicted"]
    assert p["segments"] == p["collectives_per_proof"] + 1 == len(p["segments_slowest_rank_ms"])
    assert all(len(b["turn_segments_ms"]) == p["segments"] for b in line["per_rank"]) and len(line["per_rank"]) == 2
    assert 0 < p["mean_rank_total_ms"] <= p["slowest_rank_total_ms"] <= p["compute_ms"] < p["ms_per_proof"]
    assert p["one_rank_ms_per_proof"] > 0 and abs(p["speedup_over_one_rank"] - p["one_rank_ms_per_proof"] / p["ms_per_proof"]) < 0.01
    assert "MODEL" in p["what"]


def test_bench_world_one_two_machine_proofs_in_flight_on_two_rccl_communicators():
    """torchrun with one rank, two shards: the N > 1 schedule on RCCL at world 1 -- two machine proofs in flight on two
    communicators behind the C ABI (csrc/comm.cpp: the counts' all-gather, the records' all-gather, the sums' all-reduce per proof,
    from two host threads), the librccl bound being the copy PyTorch has mapped."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--shards-per-rank", "2", "--steps", "4", "--warmup", "1", "--log-rows", "12", "--no-cpu-baseline",
           "--queries", "8", "--pow-bits", "6", "--no-compile"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert cfg["rank_proofs_in_flight"] == 2 and cfg["shards"] == 2 and cfg["rccl_world_size"] == 1
    assert "c-abi" in cfg["collectives"] and "librccl" in cfg["rccl_library"]
    assert cfg["grand_sum_is_zero"] and cfg["proofs_identical_across_steps"]
    assert cfg["gathered_proof_set"]["grand_sum_of_gathered_proofs_is_zero"]


def test_bench_two_proofs_in_flight_line_is_complete():
    """The N = 1 default schedule (--lanes 2) at a small height: the line carries the sequential measurement beside the headline,
    every proof of the timed region equals the sequential one, the sums cancel."""
    r = _bench("--gpus", "1", "--lanes", "2", "--no-host-pipeline")
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["proofs_in_flight"] == 2 and cfg["sequential"]["ms_per_step"] > 0
    assert cfg["proofs_identical_across_steps"] and cfg["grand_sum_is_zero"]
    assert set(cfg["stages_ms"]) >= {"commit_main", "permutation", "commit_perm", "quotient_all", "commit_quotient", "open", "fri_commit"}
    assert cfg["gathered_proof_set"]["grand_sum_of_gathered_proofs_is_zero"]
    assert line["roofline"]["frac"] > 0 and line["roofline"]["hbm"]["frac"] > 0


@pytest.mark.parametrize("pad", [0, 4])
def test_two_proofs_in_flight_overlap_at_the_bench_height(pad):
    """The default bench command's schedule at its own height (2^20 eval rows).  Whether two proofs in flight overlap at all is
    decided by where the runtime puts the process's streams on its hardware queues (idle streams created in the
    wrong place cost the whole 10 %); this is the tripwire -- on the PLACEMENT (the lanes' streams run beside each other), not on a
    ratio of timings, so that a loaded box cannot fail it.
    `pad` = 4: four idle streams ahead of every context, the placement that read 45 ms before the second lane's stream and the
    side streams were MEASURED into place (lurkhip_ctx_create_beside)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-host-pipeline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    if pad:
        env["LURKHIP_PAD_STREAMS"] = str(pad)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert cfg["proofs_in_flight"] == 2 and cfg["proofs_identical_across_steps"] and cfg["gathered_proof_set"]["product_verifier"]["accepted"]
    assert cfg["device_pools"]["hipMalloc_calls_in_timed_region"] == {"main": 0, "proof_lane1": 0}
    # the placement itself, not its effect on a timing (which a loaded box moves): the lanes' streams were measured to run beside
    # the main one (lurkhip_ctx_overlap_probe: a busy kernel on each stream together takes the time of one)
    assert cfg["lane_placement"] and all(p["streams_run_beside_each_other"] for p in cfg["lane_placement"]), cfg["lane_placement"]
    assert line["proof_latency_ms"] == cfg["sequential"]["ms_per_step"] and line["proofs_in_flight"] == 2
