#!/usr/bin/env python3
"""Benchmark of the Lurk proving hot path on MI355X.

Workload (BASELINE.json configs[2]): one shard = 2^20 queries of a width-78 `eval`
chip (/root/reference/src/core/eval_direct.rs:2028; the synthetic Lair function of
lurk_amd/programs/synth_eval.py stands in for the evaluator's program text), executed once on the host
into a query record whose flattened row stream is resident in HBM before the timed region.  One *step*
= one pass of the hot path over that shard: FuncChip trace generation (row kernel, 2^20 x 78) followed
by the main-trace commitment of `machine.prove` (coset LDE with blow-up 2 + Poseidon2-16 Merkle root).
Metric: eval-steps (rows of the eval chip) per second, whole job.

Multi-GPU (--gpus N, launched by torch.distributed.run): shards are independent
(`Shard::shard`, /root/reference/src/lair/execute.rs:186-216): rank r proves shard r (weak scaling) and
the ranks exchange their 8-lane roots with one RCCL all-gather per step, which is the only
cross-shard data the prover's transcript needs.

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


def build_shard(ctx, log_rows: int, rank: int):
    """Executes the synthetic eval program into 2^log_rows queries and uploads the flattened row stream."""
    from lurk_amd import lair
    from lurk_amd.programs import synth_eval as se

    top = lair.Toplevel(se.SOURCE)
    idx = top.func_index(se.FUNC)
    queries = lair.QueryRecord(top)
    args = se.args_for_rows(1 << log_rows)
    args[2] = rank  # a different environment per rank: every shard is a different trace
    top.execute(idx, args, queries)
    chip = lair.FuncChip(ctx, idx, top)
    assert chip.width() == WIDTH, chip.width()
    shard = lair.Shard.new(queries)
    prepared = lair.PreparedFuncTrace(chip, shard)
    assert prepared.n_real == prepared.height == 1 << log_rows, (prepared.n_real, prepared.height)
    return top, queries, prepared


def synthetic_trace(log_rows: int, width: int, seed_offset: int) -> np.ndarray:
    from lurk_amd import synth

    n = 1 << log_rows
    t = synth.field_elements((n, width), seed=synth.SEED + 1 + seed_offset)
    t[:, 0] = np.arange(n, dtype=np.uint32)  # nonce column = row index (src/lair/trace.rs:82-84)
    return t


def cpu_baseline(sample_log_rows: int):
    """The oracle's commit (OpenMP FFT + Merkle) on a bounded sample of the same workload."""
    from oracle import binding as ob

    ob.build()
    t = synthetic_trace(sample_log_rows, WIDTH, 0)
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    lde = ob.lde(t, LOG_BLOWUP)
    ob.merkle_commit([lde])
    dt = time.perf_counter() - t0
    return {
        "value": (1 << sample_log_rows) / dt,
        "unit": "eval-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"commit (LDE x2 + Poseidon2-16 Merkle) of a 2^{sample_log_rows} x {WIDTH} trace, OpenMP over {cores} threads, {dt:.2f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-rows", type=int, default=LOG_ROWS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log-rows", type=int, default=20)
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and distributed:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)

    import lurk_amd
    from lurk_amd import commit as cm
    from lurk_amd import field

    ctx = lurk_amd.Context(local_rank)
    log_rows, n = args.log_rows, 1 << args.log_rows
    t_host = time.perf_counter()
    top, queries, prepared = build_shard(ctx, log_rows, rank)
    t_host = time.perf_counter() - t_host
    trace = torch.zeros((n, WIDTH), dtype=torch.int32, device="cuda")
    roots = torch.zeros((world, 8), dtype=torch.int32, device="cuda")
    my_root = torch.zeros(8, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    MONTY = 1  # the device-native encoding: no conversion between trace generation and commit

    def step():
        prepared.run(trace, repr=MONTY)
        c = cm.commit_dev(ctx, [trace], [log_rows], [WIDTH], log_blowup=LOG_BLOWUP, repr=MONTY)
        if distributed:
            my_root.copy_(torch.from_numpy(c.root.view(np.int32)))
            dist.all_gather_into_tensor(roots.view(-1), my_root)
        c.close()
        return c.root

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
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    root = None
    for _ in range(args.steps):
        root = step()
    fence()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    spans = {name: ctx.profile_read(name) for name in ("trace_func", "lde", "merkle_leaves", "merkle_levels")}
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n * args.steps / elapsed

    # dominant kernel: Merkle leaf hashing (k_leaves), one launch per step.
    # algorithmic bytes per launch = LDE rows * (w*4 read + 32 written)   (DESIGN.md "Merkle leaves")
    leaf_ms, leaf_cnt = spans["merkle_leaves"]
    leaf_avg_ms = leaf_ms / max(leaf_cnt, 1)
    leaf_bytes = (n << LOG_BLOWUP) * (WIDTH * 4 + 32)
    achieved = leaf_bytes / (leaf_avg_ms * 1e-3) / 1e9 if leaf_avg_ms > 0 else 0.0

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
                "workload": f"fib trace 2^{log_rows} rows x {WIDTH} cols per GPU (eval chip): lair trace-gen (row kernel over the HBM-resident row stream) + main-trace commit = coset LDE (blow-up 2) + Poseidon2-16 Merkle root"
                + ("; RCCL all-gather of shard roots" if distributed else ""),
                "stages_ms": {k: (v[0] / max(v[1], 1)) * (v[1] / args.steps) for k, v in spans.items()},
                "parity": "Poseidon2/trace rows pinned by reference KATs; LDE/Merkle self-verified vs oracle (upstream parity unpinned)",
                "root": [int(x) for x in field.from_monty(root)],
                "row_stream_bytes": prepared.input_bytes,
                "host_execute_s": t_host,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_leaves (Merkle leaf sponge)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "avg_launch_ms": leaf_avg_ms,
                "note": "int32-VALU bound (10 width-16 permutations per 312-byte row), see DESIGN.md",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample_log_rows)
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the line
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
