#!/usr/bin/env python3
"""Benchmark of the Lurk proving hot path on MI355X.

Workload (BASELINE.json configs[2]): one shard whose `eval` chip (width 78,
/root/reference/src/core/eval_direct.rs:2028; the synthetic Lair function of lurk_amd/programs/synth_eval.py stands in
for the evaluator's program text) has 2^20 rows.  The host interpreter runs once before the timed region and the
flattened inputs of every chip (row streams, memory tables, byte-lookup records) are resident in HBM.  One *step* = one
pass of the proving hot path over that shard, i.e. what `machine.prove::<LocalProver>` does after `execute`
(/root/reference/benches/fib.rs:88-124): trace generation of every chip of the machine (eval chip, its callee, the memory
tables, the byte table), main-trace commitment, LogUp permutation traces + commitment, quotient + commitment, openings
at zeta and FRI (100 queries, 16 proof-of-work bits).  Metric: eval-steps (rows of the eval chip) proved per second.

Multi-GPU (--gpus N, launched by torch.distributed.run): shards are independent proofs
(`Shard::shard`, /root/reference/src/lair/execute.rs:186-216): rank r proves shard r (weak scaling); per step the ranks
all-gather their 8-lane main-trace roots (every shard's transcript observes every root before any challenge is drawn)
and all-reduce the extension-field cumulative sums of their chips as 4 x uint64 (the verifier's grand-sum check).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LOG_ROWS = 20
WIDTH = 78
LOG_BLOWUP = 1
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
SPANS = ("trace_func", "commit_main", "permutation", "commit_perm", "quotient_all", "commit_quotient", "open", "fri_commit", "fri_query",
         "lde", "merkle_leaves", "merkle_levels", "merkle_top")


def synthetic_trace(log_rows: int, width: int, seed_offset: int) -> np.ndarray:
    from lurk_amd import synth

    n = 1 << log_rows
    t = synth.field_elements((n, width), seed=synth.SEED + 1 + seed_offset)
    t[:, 0] = np.arange(n, dtype=np.uint32)  # nonce column = row index (src/lair/trace.rs:82-84)
    return t


def cpu_baseline(round_shapes, eval_rows: int):
    """The oracle's commit (OpenMP FFT + Poseidon2-16 Merkle) of the step's three commitment rounds -- main traces, LogUp
    permutation traces, quotient chunks -- on synthetic matrices of exactly the shapes the GPU step commits (the CPU time of
    an LDE + Merkle commit does not depend on the values).  Trace generation, the permutation / quotient arithmetic,
    openings and FRI have no compiled CPU port (the oracle does them in Python), so the CPU figure is an upper bound on what
    the port would reach on the full step; the commits are about two thirds of the GPU step."""
    from oracle import binding as ob

    ob.build()
    cores = os.cpu_count() or 1
    dt, cols = 0.0, 0
    for k, shapes in enumerate(round_shapes):
        mats = [synthetic_trace(lg, w, 100 * k + i) for i, (lg, w) in enumerate(shapes)]
        cols += sum(w << lg for lg, w in shapes)
        t0 = time.perf_counter()
        ldes = [ob.lde(m, LOG_BLOWUP) for m in mats]
        ob.merkle_commit(ldes)
        dt += time.perf_counter() - t0
        del mats, ldes
    return {
        "value": eval_rows / dt,
        "unit": "eval-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"the three commitment rounds of one step (coset LDE x2 + Poseidon2-16 Merkle over {cols / eval_rows:.0f} columns per eval row: main, permutation, quotient), "
                  f"OpenMP over {cores} threads, {dt:.2f} s; the GPU step also does trace generation, the permutation / quotient arithmetic, openings and FRI",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-rows", type=int, default=LOG_ROWS)
    ap.add_argument("--queries", type=int, default=100)
    ap.add_argument("--pow-bits", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spans", action="store_true", help="diagnostic: no per-stage HIP events in the timed region (stages_ms and roofline read 0)")
    ap.add_argument("--no-two-in-flight", action="store_true", help="skip the extra two-shards-in-flight measurement")
    ap.add_argument("--no-compile", action="store_true", help="keep every chip's AIR programs on the interpreter")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # launched by torch.distributed.run (any N, also 1)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and distributed:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)

    import lurk_amd
    from lurk_amd import lair, prover
    from lurk_amd.programs import synth_eval as se

    ctx = lurk_amd.Context(local_rank)
    log_rows, n = args.log_rows, 1 << args.log_rows

    # ---- host side, once: execute the program, flatten every chip's inputs into HBM
    t_host = time.perf_counter()
    top = lair.Toplevel(se.SOURCE)
    queries = lair.QueryRecord(top)
    prog_args = se.args_for_rows(n)
    prog_args[2] = rank  # a different environment per rank: every shard is a different trace
    top.execute(top.func_index(se.FUNC), prog_args, queries)
    pv = queries.expect_public_values()
    machine = prover.Machine(ctx, top, se.FUNC, len(pv))
    vk_root = machine.setup()
    prepared = machine.prepare_shard(lair.Shard.new(queries))
    t_host = time.perf_counter() - t_host
    # once per machine, before the timed region: the big chips' AIR programs compiled to straight-line device code
    t_jit = time.perf_counter()
    compiled = [] if args.no_compile else machine.compile_airs(prepared)
    t_jit = time.perf_counter() - t_jit
    eval_rows = queries.num_func_queries(top.func_index(se.FUNC))
    assert eval_rows == n, (eval_rows, n)
    chips_desc = [f"{air.name}:2^{lg}x{air.width}" for _, air, lg, _, _ in prepared]
    input_bytes = sum(p.input_bytes for *_, p in prepared if p is not None)

    from lurk_amd import shards

    grand_sums = []

    def step():
        traces = machine.run_prepared(prepared)
        handle, root = machine.commit_shard(traces)
        ch = prover.Challenger(ctx)
        ch.observe(vk_root)
        ch.observe([0])
        # every shard's transcript observes every shard's main root (RCCL all-gather of 8 lanes per rank)
        for r in shards.exchange_roots([root], device="cuda" if distributed else "cpu"):
            ch.observe(r)
            ch.observe(pv)
        words = machine.prove_shard(handle, ch, pv, num_queries=args.queries, pow_bits=args.pow_bits, parse=False)
        machine.free_shard(handle)
        # grand-sum check data: the chips' cumulative sums, reduced over all shards (RCCL all-reduce of 4 x int64)
        n_chips = int(words[1])
        cs = [words[10 + 11 * i + 7:10 + 11 * i + 11] for i in range(n_chips)]
        grand_sums.append(shards.reduce_cumulative_sums(cs, device="cuda" if distributed else "cpu"))
        return words

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.profile_reset()
    ctx.profile_enable(not args.no_spans)
    t0 = time.perf_counter()
    words = None
    step_words = []
    for _ in range(args.steps):
        words = step()
        step_words.append(words)  # compared after the timed region: the same shard must give the same proof every step
    fence()
    elapsed = time.perf_counter() - t0
    proofs_identical = all(len(w) == len(step_words[0]) and bool((w == step_words[0]).all()) for w in step_words[1:])
    del step_words
    ctx.profile_enable(False)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    spans = {name: ctx.profile_read(name) for name in SPANS}
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n * args.steps / elapsed

    # dominant kernels: the Merkle hashing launches (k_leaves: leaf sponges; k_level: 2-to-1 compressions + the sponges of
    # the shorter matrices injected at their level).  Algorithmic bytes: a leaf row reads its w*4 bytes and
    # writes a 32-byte digest; a level node reads two digests (64 B) + the injected rows and writes 32 B.  `achieved` =
    # bytes of all those launches in one step / their summed HIP-event time (launch-weighted average).
    def merkle_hash_bytes(mats):
        log_max = max(lg for lg, _ in mats)
        total = (1 << log_max) * (4 * sum(w for lg, w in mats if lg == log_max) + 32)
        launches = 1
        for lvl in range(1, log_max + 1):
            n_par = 1 << (log_max - lvl)
            lh = log_max - lvl
            if 2 * n_par <= 64:
                break  # k_top finishes the tree in one workgroup (latency-bound tail, reported separately)
            total += n_par * (64 + 4 * sum(w for lg, w in mats if lg == lh) + 32)
            launches += 1
        return total, launches

    rounds = [
        [(lg + LOG_BLOWUP, air.width) for _, air, lg, _, _ in prepared],
        [(lg + LOG_BLOWUP, 4 * air.permutation_width) for _, air, lg, _, _ in prepared],
        [(lg + LOG_BLOWUP, 4) for _, air, lg, _, _ in prepared for _ in range(1 << air.log_quotient_degree)],
    ]
    max_lg = max(lg for _, _, lg, _, _ in prepared) + LOG_BLOWUP
    rounds += [[(lf, 8)] for lf in range(max_lg - 1, LOG_BLOWUP - 1, -1)]  # FRI layers
    hash_bytes_step = sum(merkle_hash_bytes(r)[0] for r in rounds)
    hash_launches_step = sum(merkle_hash_bytes(r)[1] for r in rounds)
    hash_ms_step = (spans["merkle_leaves"][0] + spans["merkle_levels"][0]) / args.steps
    achieved = hash_bytes_step / (hash_ms_step * 1e-3) / 1e9 if hash_ms_step > 0 else 0.0
    # second-largest kernel family, the coset LDE passes (k_ntt_pass): every pass reads and writes its matrix once; a size-N
    # transform takes ceil(log N / 7) passes, an LDE with blow-up 2 is three transforms
    lde_bytes_step = 0
    for r in rounds[:3]:
        for lg, w in r:
            log_n = lg - LOG_BLOWUP
            lde_bytes_step += 3 * max(1, -(-log_n // 7)) * 2 * (1 << log_n) * w * 4
    lde_ms_step = spans["lde"][0] / args.steps
    lde_achieved = lde_bytes_step / (lde_ms_step * 1e-3) / 1e9 if lde_ms_step > 0 else 0.0
    traffic, valu = None, None
    try:  # HBM bytes per step of the same kernels from the rocprofv3 PMC passes (profiles/)
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("log_rows") == log_rows:
            traffic = pmc["merkle_hash_bytes_per_step"]
            valu = {"achieved": pmc["merkle_hash_valu_tinst_s"], "peak": pmc["int32_valu_peak_tinst_s"], "unit": "Tinstr/s",
                    "frac": pmc["merkle_hash_valu_tinst_s"] / pmc["int32_valu_peak_tinst_s"],
                    "source": "SQ_INSTS_VALU x 64 lanes / kernel time, profiles/r01_pmc_sq_per_kernel.csv"}
    except Exception:
        pass

    # Extra (N = 1 only, never `value`): two shards proved concurrently on two HIP streams of the one GPU, the way a multi-shard
    # proof keeps the device busy through each shard's latency chains (tree tails, FRI layers, host transcript round trips).
    two_in_flight = None
    if world == 1 and not args.no_two_in_flight:
        try:
            import threading

            ctx2 = lurk_amd.Context(local_rank)
            q2 = lair.QueryRecord(top)
            a2 = se.args_for_rows(n)
            a2[2] = 1
            top.execute(top.func_index(se.FUNC), a2, q2)
            pv2 = q2.expect_public_values()
            m2 = prover.Machine(ctx2, top, se.FUNC, len(pv2))
            vk2 = m2.setup()
            prep2 = m2.prepare_shard(lair.Shard.new(q2))
            if not args.no_compile:
                m2.compile_airs(prep2)  # same programs: served from the in-process code cache

            def one(mach, cx, prep, vk, pvs):
                traces = mach.run_prepared(prep)
                handle, root = mach.commit_shard(traces)
                ch = prover.Challenger(cx)
                ch.observe(vk)
                ch.observe([0])
                ch.observe(root)
                ch.observe(pvs)
                w = mach.prove_shard(handle, ch, pvs, num_queries=args.queries, pow_bits=args.pow_bits, parse=False)
                mach.free_shard(handle)
                return w

            in_flight_ok = []  # the first lane proves the timed region's shard: its proofs must be that proof

            def worker(mach, cx, prep, vk, pvs, k):
                for _ in range(k):
                    w = one(mach, cx, prep, vk, pvs)
                    if mach is machine:
                        in_flight_ok.append(len(w) == len(words) and bool((w == words).all()))
                cx.sync()

            for k in (1, args.steps):  # warm-up pass, then the timed one
                ths = [threading.Thread(target=worker, args=(machine, ctx, prepared, vk_root, pv, k)),
                       threading.Thread(target=worker, args=(m2, ctx2, prep2, vk2, pv2, k))]
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t1
            two_in_flight = {"shards": 2 * args.steps, "ms_per_shard": dt2 / (2 * args.steps) * 1e3, "eval_steps_per_s": 2 * n * args.steps / dt2,
                             "proofs_match_sequential": bool(in_flight_ok) and all(in_flight_ok),
                             "note": "two independent shards on two HIP streams of the same GPU; not the headline value"}
            m2.close()
            ctx2.close()
        except Exception as e:
            two_in_flight = {"error": repr(e)}

    if rank == 0:
        out = {
            "metric": "Lurk eval-steps proved/sec (fib trace)",
            "value": value,
            "unit": "eval-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"fib trace 2^{log_rows} rows x {WIDTH} cols per GPU (eval chip) + the rest of its machine: lair trace-gen, main / LogUp permutation / quotient commits (coset LDE blow-up 2 + Poseidon2-16 Merkle), openings + FRI ({args.queries} queries, {args.pow_bits} PoW bits)"
                + ("; RCCL all-gather of shard roots + all-reduce of cumulative sums" if distributed else ""),
                "chips": chips_desc,
                "stages_ms": {k: v[0] / args.steps for k, v in spans.items() if v[1]},
                "parity": "Poseidon2 / traces / AIR pinned by the reference's vectors and constraint property; commit / LogUp / quotient / FRI bit-exact vs the oracle and accepted by its verifier (upstream parity unpinned)",
                "proof_words": int(len(words)),
                "grand_sum_is_zero": all(g == (0, 0, 0, 0) for g in grand_sums),
                "proofs_identical_across_steps": proofs_identical,
                "hbm_resident_input_bytes": int(input_bytes),
                "host_execute_and_upload_s": t_host,
                "compiled_air_chips": compiled,
                "air_compile_s": t_jit,
                "two_shards_in_flight": two_in_flight,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "Merkle hashing (k_leaves + k_level + k_level_coop), all launches of a step",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "int32_valu": valu,
                "also": {"kernel": "coset LDE passes (k_ntt_pass), all launches of a step", "bound": "hbm", "achieved": lde_achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lde_achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_step": lde_bytes_step, "ms_per_step": lde_ms_step},
                "algorithmic_bytes_per_step": hash_bytes_step,
                "launches_per_step": hash_launches_step,
                "ms_per_step": hash_ms_step,
                "note": "int32-VALU bound, not HBM bound: ceil(w/8) width-16 Poseidon2 permutations (~5.0 k int32 instructions each, ~60 % of them multiply-class) per w*4-byte row; throughput-bound on instruction issue (same speed at 4 and 8 waves per SIMD), i.e. ~0.3 TB/s algorithmic is this kernel's ceiling (DESIGN.md 3.4)",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline([[(lg - LOG_BLOWUP, w) for lg, w in r] for r in rounds[:3]], n)
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the line
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()
    machine.close()
    ctx.close()


if __name__ == "__main__":
    main()
