#!/usr/bin/env python3
"""Benchmark of the Lurk proving hot path on MI355X.

Workload (BASELINE.json configs[2]): a `fib`-shaped Lair machine whose `eval` chip (width 78,
/root/reference/src/core/eval_direct.rs:2028) has 2^20 rows per GPU.  The evaluator's program text cannot be shipped, so
the machine is the shape-matched `fib-mix` of lurk_amd/programs/lurk_mix.py: the 17 function chips a real `(fib N)` touches
(+ 3 memory tables, byte table, entrypoint), each with the reference's name, signature and the shape MEASURED on the
reference's own function in the build container (tests/golden/fib_shape.json, tools/measure_lurk_shape.py: main-trace
width, selectors, lookups = permutation-trace columns exactly; constraints and lookup-tuple words exactly for the four tall
chips), at the measured heights per eval row (5 eval_builtin_expr, 4 eval_binop_num, 4 apply, 2 env_lookup, 1 eval_begin,
1 each u64_add / u64_sub / u64_lessthan, 2 + 2 memory rows per 10 eval rows).  Rounds 1-4 ran SURVEY appendix C's hand
estimate instead: 285 main columns per eval row against the measured 277, but 3.4 permutation-trace columns per main column
fewer than the real machine has (its chips have 2-3 x the lookups the estimate's stand-ins had): this round's step is the
heavier, real one, and its time is not comparable with BENCH_r01-r04.  `--workload eval-only` is round 1's thinner
single-function machine, `--workload lurk-mix` all 39 functions (BASELINE config 5, irregular widths 9 ... 815).

The host interpreter runs once before the timed region and the flattened inputs of every chip (row streams, memory
tables, byte-lookup records) are resident in HBM.  At N = 1 the K timed steps are K independent proofs of the shard with two
in flight on two HIP streams (--lanes 2, the schedule of every multi-shard proof: Machine.prove / prove_lanes); the
one-proof-at-a-time time is measured in the same run and reported as config.sequential (--lanes 1 times that instead).
One *step* = one pass of the proving hot path over one shard per GPU,
i.e. what `machine.prove::<LocalProver>` does after `execute` (/root/reference/benches/fib.rs:88-124): trace generation of
every chip, main-trace commitment, LogUp permutation traces + commitment, quotient + commitment, openings at zeta and FRI
(100 queries, 16 proof-of-work bits).  Metric: eval-steps (rows of the eval chip) proved per second.

Multi-GPU (--gpus N: launched by torch.distributed.run, or -- run as a plain command -- bench.py starts its own N ranks
that way, one per visible GPU, and refuses if there are fewer than N): ONE execution with N * 2^log_rows eval rows on every rank (the
same program, so the same query record), cut by `Shard::shard` (/root/reference/src/lair/execute.rs:186-216; Entrypoint /
memory chips only in shard 0, lair_chip.rs:124-139) into 2 N shards of 2^log_rows / 2 eval rows.  `Shard::shard` cuts every
chip at the same row count, so the first shards hold every chip and the last ones only the eval chip: the shards are dealt to
the ranks by work, two per rank (`shards.assign_shards_balanced`; one shard per rank would leave rank 0 with 2.3x the average),
and a rank proves its two shards on two contexts (`prover.prove_lanes`).  Per step: every rank commits its shards' main traces,
the ranks all-gather (shard index, 8-lane root) records (RCCL; every shard's transcript observes every root, in shard order,
before any challenge is drawn), every rank proves its shards, and the extension-field cumulative sums are all-reduced as
4 x int64: each rank's sum is non-zero, the total is zero.  Per-GPU work stays 2^log_rows eval rows as N grows ("weak").
BASELINE config 4 is `--gpus 8 --log-rows 19` (2^22 rows over 8 GPUs).

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LOG_ROWS = 20
LOG_BLOWUP = 1
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
VALU_FULL_RATE = 78.6   # T lane-instr/s: 256 CUs x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md "Wave scheduling"); = 157.3 TFLOPS fp32 / 2
VALU_HALF_RATE = 39.3   # mul-class instructions (v_mul_lo/hi, v_mad_u64_u32, VOP3 adds, ...) issue at half rate (measured)
SPANS = ("trace_all", "commit_main", "permutation", "commit_perm", "quotient_all", "commit_quotient", "open", "fri_commit", "fri_query",
         "lde", "merkle_leaves", "merkle_levels", "merkle_top")


def synthetic_trace(log_rows: int, width: int, seed_offset: int) -> np.ndarray:
    from lurk_amd import synth

    n = 1 << log_rows
    t = synth.field_elements((n, width), seed=synth.SEED + 1 + seed_offset)
    t[:, 0] = np.arange(n, dtype=np.uint32)  # nonce column = row index (src/lair/trace.rs:82-84)
    return t


CPU_SAMPLE_MAX_LOG_ROWS = 20  # the CPU port proves the bench's own shard up to 2^20 eval rows (about 20 s of CPU work on the box's 16-core quota; rounds 2-3: half height), a taller one cut down to that


def host_info():
    """The host the ranks' launch threads run on: the two-proofs-in-flight schedule needs two of them to keep up with the device,
    and the boxes of the pool differ (the same build reads 40.7 .. 45.3 ms per step over boxes while one proof at a time reads
    45.4 .. 46.4 on all of them)."""
    info = {"cpu_count": os.cpu_count()}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    info["cpu_model"] = line.split(":", 1)[1].strip()
                    break
        info["loadavg"] = [float(x) for x in open("/proc/loadavg").read().split()[:3]]
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        info["cpu_quota_cores"] = None if quota[0] == "max" else round(int(quota[0]) / int(quota[1]), 2)
    except (OSError, ValueError, IndexError):
        pass
    return info


def cpu_baseline(workload: str, chip_shapes, log_rows: int, queries: int, pow_bits: int):
    """The CPU PORT of the WHOLE step (oracle/cpu_prover.py: the oracle's transcript driving oracle/cpu_step.c -- coset LDEs and
    Poseidon2-16 Merkle trees of the three commitment rounds, LogUp permutation traces + running sums, quotient values from C
    evaluators generated out of the oracle's AIR, opened values, reduced openings, the FRI commit phase, proof-of-work, query
    openings; OpenMP, Montgomery arithmetic), every stage checked word for word against oracle/stark.py and the whole proof
    accepted by the oracle's verifier and equal to the HIP prover's (tests/test_cpu_step*.py).  Timed on a BOUNDED SAMPLE: the
    same machine at the bench's own heights (from 2^21 eval rows up: every chip of 2^12 rows and more cut down by the same factor to
    2^20 eval rows), on synthetic traces of
    those shapes (no stage's cost depends on the values; the constraints need not hold for the openings and FRI to be
    well-formed).  `value` = sample eval rows / seconds.  Trace generation of the function chips is oracle/cpu_trace.c (a C row
    loop over flattened query records, checked word for word against the oracle's generator): a small real execution of the
    machine through the oracle's interpreter, its rows repeated up to the sample's heights.  kind "port": the reference prover (Rust, sphinx + Plonky3) cannot be
    built here; a tuned CPU prover (packed AVX-512 field, cache-blocked NTTs) would be several times faster per core."""
    from oracle import air as oa
    from oracle import binding as ob
    from oracle import cpu_prover as cpv
    from oracle import lair as ol
    from oracle import stark as os_
    from lurk_amd.programs import lurk_mix as lm

    ob.build()
    cores = ob.usable_cores()
    shrink = max(0, log_rows - CPU_SAMPLE_MAX_LOG_ROWS)
    mix = lm.fib_mix(64) if workload == "fib-mix" else lm.lurk_mix(64)
    otop = ol.Toplevel(mix.source, chips=ol.lurk_chips())
    n_public = 44
    idx = otop.index[mix.entry]
    airs = [oa.EntrypointAir(idx, n_public)] + [oa.FuncAir(otop, f["name"]) for f in otop.funcs] + [oa.MemAir(ml) for ml in ol.MEM_TABLE_SIZES] + [oa.BytesAir()]
    names = [f"Entrypoint[{idx}]"] + [f"Func[{f['name']}]" for f in otop.funcs] + [f"Mem[{ml}-wide]" for ml in ol.MEM_TABLE_SIZES] + ["CPU"]
    pr = cpv.CpuProver(airs, names, n_public, threads=cores)
    rng = np.random.default_rng(0x4C55524B)
    # trace generation (oracle/cpu_trace.c, checked word for word against the oracle's generator by tests/test_cpu_trace.py): the
    # function chips' rows from flattened query records -- a small REAL execution of the same machine through the oracle's
    # interpreter, its rows repeated up to the sample's heights (rows are independent of one another in trace generation)
    from oracle import cpu_trace as ct

    small_q = ol.QueryRecord(otop)
    ol.execute(otop, mix.entry, list(mix.main_args), small_q,
               poseidon=lambda w_, inp: [int(v) for v in ob.p2_permute(w_, np.array(inp, dtype=np.uint32))[0]])
    t_trace = 0.0
    traces, sample_rows = [], None
    for name, lg, w in chip_shapes:
        mi = names.index(name)
        assert airs[mi].width == w, (name, airs[mi].width, w)
        lgs = lg - shrink if lg >= 12 and name != "CPU" else lg
        mat = None
        if name.startswith("Func["):
            fname = name[5:-1]
            n0, hdr0, hints0, offs0 = ct.flatten(otop, fname, small_q)
            if n0:
                prog = ct.FuncProgram(otop, fname)
                rows_ = 1 << lgs
                reps = -(-rows_ // n0)
                lens = np.diff(offs0.astype(np.int64))
                hdr_t = np.tile(hdr0[:n0], (reps, 1))[:rows_]
                hints_t = np.tile(hints0[:int(offs0[-1])], reps)
                offs_t = np.concatenate([[0], np.cumsum(np.tile(lens, reps))]).astype(np.uint64)[:rows_ + 1]
                hints_t = np.concatenate([hints_t, np.zeros(1, dtype=np.uint32)])
                t_q = time.perf_counter()
                mat = ct.run(prog, rows_, hdr_t, hints_t, offs_t, rows_)
                t_trace += time.perf_counter() - t_q
        if mat is None:
            mat = rng.integers(0, 2013265921, size=(1 << lgs, w), dtype=np.uint32)
        traces.append((mi, mat))
        if name == "Func[eval]":
            sample_rows = 1 << lgs
    i = np.arange(1 << 16)
    i1, i2 = i & 0xFF, i >> 8
    prep_m, pc = pr.setup({len(airs) - 1: np.stack([i1, i2, (i1 < i2).astype(int), i1 & i2, i1 ^ i2, i1 | i2], axis=1).astype(np.uint32)})
    ch = os_.Challenger(os_.default_permute16())
    ch.observe(pc["root"])
    ch.observe(0)
    stages = {}
    t0 = time.perf_counter()
    pr.prove_shard(traces, prep_m, pc, [0] * n_public, ch, num_queries=queries, pow_bits=pow_bits, timings=stages)
    dt = time.perf_counter() - t0 + t_trace
    import ctypes

    avx512 = ctypes.c_int(0)
    perm_ns = float(pr.L.cp2_perm_ns(20000, ctypes.byref(avx512)))  # one core, sixteen states at a time
    gpu_names = {"commit_main": "commit_main", "permutation": "permutation", "commit_perm": "commit_perm", "quotient_all": "quotient_all",
                 "commit_quotient": "commit_quotient", "open": "open", "fri_commit": "fri_commit", "pow": "fri_query", "fri_query": "fri_query"}
    stages_s = {"trace_all": t_trace}
    for k, v in stages.items():
        g = gpu_names.get(k, k)
        stages_s[g] = stages_s.get(g, 0.0) + v
    return {
        "value": sample_rows / dt,
        "unit": "eval-steps/s",
        "cores": cores,
        "kind": "port",
        "seconds": dt,
        "poseidon2_16_permutations_per_s_per_core": 1e9 / perm_ns,
        "poseidon2_16_routine": "AVX-512 packed Montgomery (oracle/cpu_port.c: perm16_v_avx512)" if avx512.value else "auto-vectorised loop (no AVX-512 on this host)",
        "ntt": "row-major radix-2 DIF, cache-blocked: stages in groups whose rows fit L2 (two passes over a 2^20-row matrix)",
        "stages_s": stages_s,
        "sample": f"the WHOLE step (function-chip trace generation from flattened query records, main / permutation / quotient commitments, LogUp rows, quotient, openings, FRI with {queries} queries and {pow_bits} PoW bits) "
                  f"on {'the bench shard itself' if shrink == 0 else 'a shard cut down by 2^' + str(shrink)}: 2^{log_rows - shrink} eval rows of the {workload} machine (a small real execution's rows repeated; memory / byte / entry chips synthetic); oracle/cpu_trace.c + cpu_prover.py + cpu_step.c, OpenMP over "
                  f"{cores} threads, {dt:.1f} s; stage names as in config.stages_s of the GPU line (proof-of-work counted under fri_query; to_montgomery = input conversion); "
                  "not the reference binary (no Rust toolchain): never quote the ratio as 'vs the reference'",
        "evaluator_build_s": pr.build_s,
    }


def usable_cores() -> int:
    """Affinity mask capped by the cgroup CPU quota (the library sizes its staging threads the same way)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


_ROCTX = None


def profiler_region(on: bool):
    """`rocprofv3 --selected-regions` collects only between roctxProfilerResume(0) and roctxProfilerPause(0): the bench brackets its
    TIMED region with them, so that a kernel-stats file divided by --steps is per-proof evidence (set-up, warm-up, the placement
    probes and the verifier stay out: VERDICT round 4, weak 12).  A no-op unless the roctx library loads."""
    global _ROCTX
    if _ROCTX is None:
        import ctypes

        _ROCTX = False
        for name in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1"):
            try:
                _ROCTX = ctypes.CDLL(name)
                _ROCTX.roctxProfilerResume.argtypes = _ROCTX.roctxProfilerPause.argtypes = [ctypes.c_uint64]
                break
            except OSError:
                continue
    if _ROCTX:
        (_ROCTX.roctxProfilerResume if on else _ROCTX.roctxProfilerPause)(0)


def step_sources_changed(pmc) -> bool:
    """True when a file a kernel of the step is made of differs from the tree profiles/pmc_traffic.json's PMC passes ran on (the
    static instruction counts belong to those sources)."""
    import hashlib

    hs = hashlib.sha256()
    for rel in pmc.get("step_kernel_sources", []):
        with open(os.path.join(ROOT, rel), "rb") as fsrc:
            hs.update(fsrc.read())
    return hs.hexdigest() != pmc.get("step_kernel_sources_sha256")


def measured_shape(workload: str):
    """What tools/measure_lurk_shape.py measured on the reference's own functions for the largest `(fib N)` it ran (unpadded rows;
    the fields above count the padded heights the prover works on)."""
    if workload != "fib-mix":
        return None
    from lurk_amd.programs import lurk_mix as lm

    shape = lm.load_shape()
    n = max(shape["fib"], key=int)
    r = shape["fib"][n]
    return {"source": "tests/golden/fib_shape.json", "fib_n": int(n), "eval_rows": r["rows"]["eval"], "main_columns_per_eval_row": r["main_columns_per_eval_row"],
            "func_permutation_columns_per_eval_row": r["func_permutation_columns_per_eval_row"], "widths_reproduced": shape["widths_reproduced"],
            "per_fib_level": shape["fib_per_level"],
            # (round 5) the prover skips permutation batches no row of a wave uses and leaves identically-zero permutation columns out
            # of the LDE: the share of such cells on the real machine (the reference's functions, tools/measure_lookup_sparsity.py) --
            # the stand-in is dialled to it and held at or below it (tests/test_mix_programs.py)
            "dead_permutation_cell_share_real": shape.get("lookup_sparsity", {}).get("dead_cell_share_at_2^20")}


def build_workload(name: str, world: int, log_rows: int, eval_rows: int | None = None):
    """(source, lurk_chips, entry, main args, eval function name, description)."""
    n = eval_rows if eval_rows else 1 << log_rows
    if name == "eval-only":
        from lurk_amd.programs import synth_eval as se

        if world != 1:
            raise SystemExit("--workload eval-only is single-GPU (its callee chip is taller than its eval chip, so it does not shard by eval rows)")
        return se.SOURCE, False, se.FUNC, se.args_for_rows(n), se.FUNC, "eval-only: one width-78 non-partial function + a 9-column callee (round 1's workload)"
    from lurk_amd.programs import lurk_mix as lm

    mix = lm.fib_mix(world * n) if name == "fib-mix" else lm.lurk_mix(world * n)
    desc = ("fib-mix: the 17 function chips of a real (fib N) run at the shapes and heights measured on the reference's own functions "
            "(tests/golden/fib_shape.json), partial eval, u64 extern chips"
            if name == "fib-mix" else "lurk-mix: all 39 Lurk functions at their measured shapes (widths 9 ... 815), 6 memory tables, byte table, entrypoint")
    return mix.source, True, mix.entry, mix.main_args, "eval", desc


def poseidon2_bench(args, world, rank, device_index, distributed):
    """BASELINE configs[1]: 2^log_n standalone Poseidon2-BabyBear hashes (`PoseidonChipset::hash`, /root/reference/src/core/poseidon.rs:30-38;
    `Hasher::hash`, /root/reference/src/core/zstore.rs:241-248) per step through lurkhip_poseidon2_hash8_dev, inputs and digests resident in
    HBM, canonical words (the ZStore's form).  N > 1: every rank hashes its own batch (no collective: the path does not exchange anything)."""
    import torch
    import torch.distributed as dist

    import lurk_amd
    from lurk_amd.poseidon import PoseidonChipset

    P = 2013265921
    ctx = lurk_amd.Context(device_index)
    n = 1 << args.log_n
    dev = torch.device("cuda", device_index)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x4C55524B + rank)  # "LURK" 

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(width, steps, warmup, what="hash"):
        chip = PoseidonChipset(ctx, width)
        x = torch.randint(0, P, (n, width), dtype=torch.int32, device=dev, generator=gen)
        cols = 8 if what == "hash" else 8 + chip.num_cols()
        out = torch.empty((n if what == "hash" else min(n, 1 << 20), cols), dtype=torch.int32, device=dev)
        rows = out.shape[0]
        run = (lambda: chip.hash_dev(x, out, rows)) if what == "hash" else (lambda: chip.witness_dev(x, out, rows))
        torch.cuda.synchronize()
        for _ in range(warmup):
            run()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        fence()
        dt = (time.perf_counter() - t0) / steps
        sample = None
        if what == "hash":  # parity of a sample against the oracle (the checker, after the timed region)
            from oracle import binding as ob

            k = 4096
            xs = x[:k].cpu().numpy().view(np.uint32)
            sample = bool(np.array_equal(out[:k].cpu().numpy().view(np.uint32), ob.p2_hash8(width, xs)))
        del x, out
        return dt, rows, sample

    W = args.p2_width
    dt, _, parity = measure(W, args.steps, args.warmup)
    t_all = torch.tensor([dt], dtype=torch.float64)
    if distributed:
        t_dev = t_all.to(dev) if dist.get_backend() == "nccl" else t_all
        dist.all_reduce(t_dev, op=dist.ReduceOp.MAX)
        dt = float(t_dev.cpu()[0])
    if rank == 0:
        alg_bytes = (W + 8) * 4 * n  # (W + 8) * 4 B per permutation
        hbm = alg_bytes / dt / 1e9
        valu = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_poseidon2.json")) as f:
                pmc = json.load(f)
            per_perm = pmc["valu_lane_insts_per_permutation"][str(W)]
            rate = per_perm * n / dt / 1e12
            valu = {"achieved": rate, "unit": "T lane-instr/s", "peak": VALU_FULL_RATE, "frac": rate / VALU_FULL_RATE, "valu_lane_insts_per_permutation": per_perm,
                    "source": "profiles/pmc_poseidon2.json (rocprofv3 --pmc SQ_INSTS_VALU x 64 lanes / permutations, static) / this run's time (live)"}
        except Exception:
            pass
        others = {}
        for w2 in (16, 24, 32, 40):
            if w2 != W:
                d2, _, ok2 = measure(w2, max(3, args.steps // 4), 1)
                others[f"hash8_w{w2}"] = {"ms_per_step": d2 * 1e3, "permutations_per_s": n / d2, "GBs_algorithmic": (w2 + 8) * 4 * n / d2 / 1e9, "parity_sample": ok2}
        for w2 in (24, 32, 40):  # the wide witness of the hash chips' rows (/root/reference/src/core/poseidon.rs:65-72), 2^20 rows
            d2, rows2, _ = measure(w2, 3, 1, what="witness")
            cols2 = 8 + PoseidonChipset(ctx, w2).num_cols()
            others[f"wide_witness_w{w2}"] = {"rows": rows2, "ms": d2 * 1e3, "rows_per_s": rows2 / d2, "GBs_algorithmic": (w2 + cols2) * 4 * rows2 / d2 / 1e9}
        out = {
            "metric": f"Poseidon2-BabyBear hashes/sec (width {W} -> 8 lanes), 2^{args.log_n} per step", "value": world * n / dt, "unit": "permutations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: standalone Poseidon2-BabyBear, 2^{args.log_n} permutations of width {W} per GPU through lurkhip_poseidon2_hash8_dev "
                                   "(PoseidonChipset::hash, src/core/poseidon.rs:30-38; canonical words, inputs and digests resident in HBM)",
                       "parity_sample_vs_oracle": parity, "other_shapes": others},
            "roofline": {"bound": "int32-valu" if valu else "hbm", "kernel": "poseidon2.hip: the hash8 kernel (one permutation per lane, state in registers)",
                         "achieved": valu["achieved"] if valu else hbm, "peak": VALU_FULL_RATE if valu else HBM_PEAK_GBS, "unit": "T lane-instr/s" if valu else "GB/s",
                         "frac": valu["frac"] if valu else hbm / HBM_PEAK_GBS, "int32_valu": valu,
                         "hbm": {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_step": alg_bytes, "rule": "SURVEY.md 8(d): (W + 8) * 4 B per permutation"},
                         "traffic": None,
                         "note": "instruction-bound: ~1.3 k modular multiplications per width-24 permutation against 128 B of traffic (SURVEY.md 8d); the HBM fraction is reported because the contract asks for it"},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                from oracle import binding as ob
                from lurk_amd import synth

                cores = usable_cores()
                ob.cpu_port_set_threads(cores)
                k = 1 << 22
                xs = synth.field_elements((k, W), seed=7)
                kind_note = "oracle/cpu_port.c: cp_p2_hash8 (sixteen rows per AVX-512 register set, OpenMP over blocks)"
                try:
                    ob.cpu_port_p2_hash8(W, xs[:64])
                    fn = lambda: ob.cpu_port_p2_hash8(W, xs)
                except RuntimeError:
                    k = 1 << 18
                    xs = xs[:k]
                    fn, cores, kind_note = (lambda: ob.p2_hash8(W, xs)), 1, "oracle/poseidon2.c (scalar, one thread: no AVX-512 on this host)"
                fn()
                reps, t0 = 0, time.perf_counter()
                while reps < 3 or time.perf_counter() - t0 < 10.0:
                    fn()
                    reps += 1
                dtc = (time.perf_counter() - t0) / reps
                out["cpu_baseline"] = {"value": k / dtc, "unit": "permutations/s", "cores": cores, "kind": "port",
                                       "sample": f"{reps} passes over 2^{k.bit_length() - 1} width-{W} preimages ({reps * dtc:.1f} s), {kind_note}; a port, not the reference's "
                                                 "`Hasher::hash` (no Rust toolchain): never quote the ratio as 'vs the reference'"}
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


XGMI_LINK_GBS = 153.0  # per link and direction, seven links per GPU (the task statement's figure; MI355X_MICROARCH.md)
SPLIT_COLLECTIVE_LATENCY_US = 50.0  # --split-turns' model: a small RCCL collective with its launch (an assumption, stated in the line)


def split_intra(args, world, rank, device_index, distributed, oversubscribed):
    """`--split intra`: ONE shard of 2^log_rows eval rows proved by all `world` ranks together (strong scaling).  A step = trace
    generation (a cut chip's rows by blocks, one block per rank) + the shard proof made by the ranks together; every rank ends with the same
    proof words, rank 0 verifies them after the timed region.  N = 1 runs the one-rank prover one proof at a time, so that the
    N-rank lines divide by a number measured the same way."""
    import torch
    import torch.distributed as dist

    import lurk_amd
    from lurk_amd import lair, prover, split

    ctx = lurk_amd.Context(device_index)
    if args.profile != "default":
        from lurk_amd.profile import ProtocolProfile

        ProtocolProfile.preset(args.profile).install(ctx)
    log_rows = args.log_rows if args.log_rows is not None else (18 if args.workload == "lurk-mix" else LOG_ROWS)
    n = 1 << log_rows
    source, lurk_chips, entry, main_args, eval_name, workload_desc = build_workload(args.workload, 1, log_rows)
    top = lair.Toplevel(source, lurk_chips=lurk_chips)
    # host side, once, on every rank: every rank holds the shard's kernel inputs and generates its own rows of them
    t0 = time.perf_counter()
    queries = lair.QueryRecord(top)
    top.execute(top.func_index(entry), main_args, queries)
    t_execute = time.perf_counter() - t0
    assert queries.num_func_queries(top.func_index(eval_name)) == n
    pv = queries.expect_public_values()
    machine = prover.Machine(ctx, top, entry, len(pv))
    vk_root = machine.setup()
    prepared = machine.prepare_shard(lair.Shard.new(queries))
    compiled = [] if args.no_compile else machine.compile_airs(prepared, min_log_rows=args.compile_min_log_rows)
    if not args.no_compile and getattr(machine, "compile_failures", None):
        raise SystemExit("bench.py: kernels that were to be compiled were not: " + "; ".join(f"{w}: {e}" for w, e in machine.compile_failures))
    carrier, carrier_note, sp, comm_c = "one rank", None, None, None
    if world > 1:
        scomm = None
        if not oversubscribed and not args.torch_collectives:
            from lurk_amd.comm import bring_up

            comm_c, carrier_note = bring_up(ctx)  # collectively, with a self-test; on failure every rank takes the host route together
            if comm_c is not None:
                scomm, carrier = split.RcclSplitComm(ctx, comm_c), "RCCL behind the C ABI (ncclSend / ncclRecv pairs, device buffers)"
                # the four collectives once with known words, before anything is timed: a carrier that moves a block to the wrong place
                # must not be found out by a proof that does not verify -- every rank goes to the host route together
                bad = scomm.selftest()
                flag = torch.tensor([0 if bad is None else 1], dtype=torch.int64, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if int(flag.item()):
                    carrier_note = f"RCCL carrier's self-test failed ({bad or 'on another rank'})"
                    scomm = None
                    comm_c.close()
                    comm_c = None
        if scomm is None:
            group = None if oversubscribed else dist.new_group(backend="gloo")
            scomm, carrier = split.TorchSplitComm(ctx, group), "torch.distributed gloo through host memory" + (" (several ranks share a device)" if oversubscribed else " (fall-back)")
        sp = split.SplitProver(machine, scomm, args.split_min_log_rows)
        assert sp.setup() == vk_root, "the ranks' verifying key differs from one rank's"

    block_bufs = sp.block_buffers(prepared) if sp is not None else None

    def step():
        if sp is not None:  # every cut chip's trace: this rank's block of rows only
            ctx.span_begin("trace_all")
            blocks = sp.run_prepared_blocks(prepared, block_bufs)
            ctx.span_end("trace_all")
            return sp.prove(blocks, pv, args.queries, args.pow_bits, row_blocks=True)[0]
        ctx.span_begin("trace_all")
        traces = machine.run_prepared(prepared)
        ctx.span_end("trace_all")
        ch = prover.Challenger(ctx)
        ch.observe(vk_root)
        ch.observe([0])
        handle, root = machine.commit_shard(traces)
        ch.observe(root)
        ch.observe(pv)
        w = machine.prove_shard(handle, ch, pv, args.queries, args.pow_bits, parse=False)
        machine.free_shard(handle)
        return w

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    gc.collect()
    gc.freeze()
    fence()
    for _ in range(args.warmup):
        step()
    # a short pass with the stage events on, for the per-rank stage table (outside the timed region)
    ctx.profile_reset()
    ctx.profile_enable(not args.no_spans)
    n_span = max(2, min(4, args.steps))
    for _ in range(n_span):
        step()
    fence()
    ctx.profile_enable(False)
    span_names = SPANS + ("split_exchange_a", "split_exchange_b")
    stages = {k: round(v[0] / n_span, 3) for k, v in ((name, ctx.profile_read(name)) for name in span_names) if v[1]}
    stats0 = np.zeros(3, dtype=np.uint64)
    lurk_amd._native.lib.lurkhip_split_stats(ctx.handle, stats0.ctypes.data, 1)
    fence()
    t0 = time.perf_counter()
    words = None
    for _ in range(args.steps):
        words = step()
    fence()
    dt = time.perf_counter() - t0
    stats = np.zeros(3, dtype=np.uint64)
    lurk_amd._native.lib.lurkhip_split_stats(ctx.handle, stats.ctypes.data, 0)
    per_step = {"alltoall_bytes_sent_before_lde": int(stats[0]) // args.steps, "alltoall_bytes_sent_after_lde": int(stats[1]) // args.steps,
                "alltoalls": int(stats[2]) // args.steps}
    # ---- what the ranks' work would take on devices of their own (--split-turns): the ranks take turns between the collectives
    turns, turn_stages = None, None
    if args.split_turns and sp is not None and isinstance(scomm, split.TorchSplitComm):
        n_turn = 5
        segs = []
        ctx.profile_reset()
        ctx.profile_enable(not args.no_spans)  # (the stage events of a rank that has the device to itself: its kernels' own time)
        for _ in range(n_turn):
            scomm.begin_turns()
            step()
            segs.append(scomm.end_turns())
        assert len({len(x) for x in segs}) == 1, "the proofs of one shard went through different numbers of collectives"
        turns = [min(x[k] for x in segs) for k in range(len(segs[0]))]  # (the fastest of the five passes, segment by segment)
        ctx.profile_enable(False)
        turn_stages = {k: round(v[0] / n_turn, 3) for k, v in ((name, ctx.profile_read(name)) for name in SPANS) if v[1]}
    one_rank_ms, one_rank_stages = None, None
    if args.split_turns and sp is not None:
        # the one-rank prover on the same shard, one proof at a time, alone on the device: what the split proof's time is divided by
        fence()
        if rank == 0:
            sp_saved, ts = sp, []
            sp = None
            ctx.profile_reset()
            ctx.profile_enable(not args.no_spans)
            for _ in range(4):
                ctx.sync()
                t1 = time.perf_counter()
                step()
                ctx.sync()
                ts.append(time.perf_counter() - t1)
            sp = sp_saved
            ctx.profile_enable(False)
            one_rank_stages = {k: round(v[0] / 4, 3) for k, v in ((name, ctx.profile_read(name)) for name in SPANS) if v[1]}
            one_rank_ms = min(ts[1:]) * 1e3
        fence()
    mine = {"rank": rank, "seconds": dt, "stages_ms": stages, "turn_segments_ms": None if turns is None else [round(t * 1e3, 4) for t in turns],
            "turn_stages_ms": None if turns is None else turn_stages, **per_step, "proof_words": int(len(words)), "proof_crc": int(np.bitwise_xor.reduce(words.astype(np.uint32)))}
    if distributed:
        box = [None] * world
        dist.all_gather_object(box, mine)
    else:
        box = [mine]
    if rank == 0:
        t_max = max(b["seconds"] for b in box)
        verified = bool(machine.verify([words]))
        identical = len({(b["proof_words"], b["proof_crc"]) for b in box}) == 1
        sent = max(b["alltoall_bytes_sent_before_lde"] + b["alltoall_bytes_sent_after_lde"] for b in box)
        per_link = sent / max(world - 1, 1)
        line = {
            "metric": "Lurk eval-steps proved/sec (fib trace) at 1/2/4/8 MI355X; bit-exact proof vs CPU", "value": n * args.steps / t_max, "unit": "eval-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_max / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: ONE shard of 2^{log_rows} eval rows (BASELINE configs[2]) proved by {world} rank(s) together (--split intra); "
                            f"{args.queries} FRI queries, {args.pow_bits} PoW bits",
                "workload_detail": workload_desc,
                "split": {"min_log_rows": args.split_min_log_rows, "carrier": carrier, "carrier_note": carrier_note, "oversubscribed": bool(oversubscribed),
                          "note": "chips of at least 2^min_log_rows rows: traces, permutation rows and quotient values by row blocks, one block per rank; per commitment one all-to-all "
                                  "rows -> column tiles, the LDE on the tiles, ONE all-to-all to storage-row blocks (the key's preprocessed traces are whole on every rank: the second "
                                  "only), subtree roots all-gathered; FRI on every rank from the all-gathered reduced openings (DESIGN.md 6)"},
                "chips_compiled": len(compiled), "host_execute_s": t_execute,
            },
            "per_rank": [{k: b[k] for k in ("rank", "seconds", "stages_ms", "alltoall_bytes_sent_before_lde", "alltoall_bytes_sent_after_lde", "alltoalls", "turn_segments_ms")
                          if b.get(k) is not None} for b in box],
            "alltoall_bytes_per_rank_per_step": sent, "alltoall_bytes_per_link_per_step": per_link,
            "xgmi_model_ms_per_step": per_link / (XGMI_LINK_GBS * 1e9) * 1e3 if world > 1 else 0.0,
            "xgmi_model_note": f"bytes one rank sends to ONE peer per step / {XGMI_LINK_GBS} GB/s: the seven links of a GPU carry its seven peers' blocks side by side",
            "proofs_identical_on_all_ranks": identical, "proof_verified": verified,
            "roofline": None, "cpu_baseline": None,
            "note": "roofline / cpu_baseline: see the N = 1 line of the default command (this mode adds no kernel: the same LDE, hashing and AIR kernels on a rank's share)",
        }
        if box[0]["turn_segments_ms"] is not None:
            segs = [b["turn_segments_ms"] for b in box]
            n_seg = len(segs[0])
            assert all(len(x) == n_seg for x in segs)
            compute = sum(max(x[k] for x in segs) for k in range(n_seg))
            n_coll = n_seg - 1
            coll_ms = n_coll * SPLIT_COLLECTIVE_LATENCY_US * 1e-3 + per_link / (XGMI_LINK_GBS * 1e9) * 1e3
            line["predicted"] = {
                "what": "a MODEL from measurements on one device, not a multi-GPU measurement: the ranks took turns between the collectives (a token goes round: "
                        "rank 0 works alone from one collective to its arrival at the next, then rank 1, ...), so a segment's time is what that rank's host and device "
                        "work takes with the device to itself; a G-GPU proof lasts at least the sum over segments of the slowest rank's time, plus the collectives",
                "segments": n_seg, "collectives_per_proof": n_coll,
                "compute_ms": round(compute, 3), "slowest_rank_total_ms": round(max(sum(x) for x in segs), 3), "mean_rank_total_ms": round(sum(sum(x) for x in segs) / len(segs), 3),
                "collectives_model_ms": round(coll_ms, 3),
                "collectives_model": f"{n_coll} collectives x {SPLIT_COLLECTIVE_LATENCY_US} us + the bytes a rank sends to ONE peer per proof / {XGMI_LINK_GBS} GB/s",
                "ms_per_proof": round(compute + coll_ms, 3), "one_rank_ms_per_proof": None if one_rank_ms is None else round(one_rank_ms, 3),
                "speedup_over_one_rank": None if one_rank_ms is None else round(one_rank_ms / (compute + coll_ms), 3),
                "segments_slowest_rank_ms": [round(max(x[k] for x in segs), 3) for k in range(n_seg)],
                "stages_ms_rank0_taking_turns": box[0]["turn_stages_ms"], "stages_ms_one_rank": one_rank_stages,
                "note": "every segment ends with the stream drained, which RCCL's device-side collectives do not need: conservative on that side; the xGMI figure is the "
                        "link rate, not a measured all-to-all",
            }
        if oversubscribed:
            line["note"] += "; the ranks SHARE one device here: ms_per_step says nothing about scaling, only the stage spans, the bytes and the identity of the proofs do"
        print(json.dumps(line))
    if distributed:
        dist.barrier()
    if sp is not None:
        sp.close()
    if comm_c is not None:
        comm_c.close()
    machine.close()
    ctx.close()
    if distributed:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("fib-mix", "eval-only", "lurk-mix", "poseidon2"), default="fib-mix",
                    help="fib-mix (default): BASELINE configs[2], the metric's workload; lurk-mix: configs[4]; poseidon2: configs[1], 2^--log-n standalone Poseidon2 hashes")
    ap.add_argument("--eval-rows", type=int, default=None,
                    help="N = 1: exactly this many eval rows instead of 2^--log-rows -- e.g. --workload lurk-mix --eval-rows 6867: the machine at the heights "
                         "of the reference's demo/mastermind.lurk (tests/golden/fib_shape.json: mastermind), the REPL-sized proof of BASELINE configs[4]")
    ap.add_argument("--log-n", type=int, default=24, help="--workload poseidon2: log2 of the permutations per step")
    ap.add_argument("--p2-width", type=int, default=24, choices=(16, 24, 32, 40), help="--workload poseidon2: the headline width (24 = hash3, the commitment hash)")
    ap.add_argument("--log-rows", type=int, default=None, help="log2 of the eval rows per GPU (default 20; 18 for lurk-mix)")
    ap.add_argument("--queries", type=int, default=100)
    ap.add_argument("--pow-bits", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spans", action="store_true", help="diagnostic: no per-stage HIP events in the timed region (stages_ms and roofline read 0)")
    ap.add_argument("--lanes", type=int, choices=(1, 2, 3, 4), default=2,
                    help="N = 1: proofs in flight during the timed steps (2 = the K steps are K independent proofs of the shard dealt to two HIP "
                         "streams of the GPU, the way Machine.prove / prove_lanes run a multi-shard proof; 1 = one proof at a time)")
    ap.add_argument("--no-two-in-flight", action="store_true", help="(accepted for old command lines; same as --lanes 1)")
    ap.add_argument("--rank-pipeline", action="store_true",
                    help="N > 1 (or --shards-per-rank > 1): phase 1 of machine proof j + 1 under phase 2 of proof j on a second machine (measured on one "
                         "GPU through RCCL at world 1: 47.9 against 46.4 ms per step -- phase 2 already keeps two shards in flight; off by default)")
    ap.add_argument("--rank-in-flight", type=int, default=None,
                    help="N > 1 (or --shards-per-rank > 1): machine proofs IN FLIGHT per rank, each on its own machine, contexts and communicator "
                         "(shards.run_in_flight: a rank never idles at a collective).  Default 2 on one device and over gloo, 1 with world > 1 on RCCL "
                         "(two communicators driven from two threads are unordered between ranks: not validated on a multi-GPU box yet)")
    ap.add_argument("--rank-in-flight-lanes", type=int, choices=(1, 2), default=1,
                    help="with --rank-in-flight >= 2: streams per machine proof (1 = its shards one after the other on the machine's stream; 2 = two of its shards at a time)")
    ap.add_argument("--rank-pipeline-depth", type=int, default=2,
                    help="with --rank-pipeline: machines per rank.  2 = phase 1 of proof j + 1 under phase 2 of proof j, joined once per proof "
                         "(shards.run_pipelined); >= 3 = a committer thread runs phase 1 up to depth - 1 proofs ahead (shards.run_committed_ahead)")
    ap.add_argument("--rank-pipeline-one-lane", action="store_true",
                    help="with --rank-pipeline: phase 2 proves the rank's shards one after the other on ONE lane (two streams busy in all: phase 2 of proof j, phase 1 of proof j + 1) instead of two")
    ap.add_argument("--stagger-ms", type=float, default=None,
                    help="--lanes >= 2: lane k starts its first timed proof k * this many milliseconds late (inside the timed region), so that the lanes run "
                         "out of phase (one lane's commitments under the other's openings and FRI).  Default: one proof's sequential time / lanes, "
                         "measured in the same run (two lanes: half a proof) when there are at least 16 timed steps, else 0; 0 = all lanes start together (2.2-2.7 %% slower, DESIGN.md section 4)")
    ap.add_argument("--no-compile", action="store_true", help="keep every chip's AIR programs on the interpreter")
    ap.add_argument("--compile-min-log-rows", type=int, default=0,
                    help="compile the AIR programs and trace generators of chips from 2^this rows up (default 0 = every chip: build() warms the code-object "
                         "cache for the whole fib-mix machine, and a 2^12-row proof is 5.9 ms with every chip compiled against 6.5-7.0; the library's own default is 2^17)")
    ap.add_argument("--no-second-profile", action="store_true", help="skip the child run under the p3-monty-diffusion protocol profile (ms_per_step_p3_monty_diffusion)")
    ap.add_argument("--no-host-pipeline", action="store_true", help="skip the extra streamed multi-shard measurement (host flatten + upload under the proofs)")
    ap.add_argument("--pipeline-shards", type=int, default=4)
    ap.add_argument("--oversubscribe", action="store_true",
                    help="diagnostic: allow more ranks than visible GPUs (ranks share devices, collectives over gloo); never a scaling number")
    ap.add_argument("--torch-collectives", action="store_true",
                    help="the two collectives of a multi-rank step through torch.distributed instead of the C ABI (lurkhip_exchange_roots / lurkhip_reduce_sums)")
    ap.add_argument("--profile", default="default", help="protocol profile preset (lurkhip_protocol_profile_preset): default, hardened, whole-state-squeeze, p3-monty-diffusion")
    ap.add_argument("--shards-total", type=int, default=None,
                    help="cut the execution into exactly this many shards (any number: 9 on 8 ranks is an ordinary case of the reference's "
                         "ceil(rows / max_shard_size)); the ranks then hold different numbers of shards.  Default: ranks x --shards-per-rank")
    ap.add_argument("--execute-on", choices=("rank0", "all"), default="rank0",
                    help="distributed runs: who runs the host interpreter.  rank0 (default): rank 0 executes the ONE program, flattens every rank's shards "
                         "and sends them to their owners (shards.scatter_prepared; lurkhip_func_trace_export / _import): the other ranks' host seconds and "
                         "resident set do not grow with N.  all: every rank executes the whole program (rounds 2-5)")
    ap.add_argument("--shards-per-rank", type=int, default=None,
                    help="distributed runs: shards of 2^log_rows / k eval rows, k per rank, dealt by work (default 2 when WORLD_SIZE > 1, else 1)")
    ap.add_argument("--split", choices=("shards", "intra"), default="shards",
                    help="how N > 1 ranks share the work.  shards (default): every rank proves its own shards of one N-times-larger execution (weak "
                         "scaling, SURVEY.md 8e first bullet).  intra: ONE shard of 2^log_rows eval rows proved by all ranks together -- column-tile LDEs, "
                         "one all-to-all to row blocks, subtree roots all-gathered (lurk_amd/split.py, csrc/split.hip): strong scaling, the case of "
                         "every execution below the reference's default shard size of 2^22 rows")
    ap.add_argument("--no-split-probe", action="store_true",
                    help="N > 1: do not also measure --split intra in child processes (config.split_intra of the line)")
    ap.add_argument("--split-probe-timeout", type=int, default=150)
    ap.add_argument("--split-turns", action="store_true",
                    help="--split intra with --oversubscribe: after the timed steps, five proofs in which the ranks take turns between the collectives -- each "
                         "rank's segments timed with the device to itself -- and the one-rank prover alone, for `predicted` (a labelled model of the G-GPU proof)")
    ap.add_argument("--split-min-log-rows", type=int, default=12, help="--split intra: chips of at least 2^k rows are cut across the ranks, the shorter ones proved whole by every rank")
    args = ap.parse_args()

    import torch

    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # launched by torch.distributed.run (any N, also 1)
    n_devices = torch.cuda.device_count()
    if not distributed and args.gpus > 1:
        # `python bench.py --gpus N` on its own: start the N ranks here, one per visible GPU, the way the driver's launcher does
        # (round 2 silently ran ONE rank and printed n_gpus 1).  Fewer than N devices is an error unless --oversubscribe asks for
        # the diagnostic run in which several ranks share a device.
        if n_devices < args.gpus and not args.oversubscribe:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_devices} GPU(s) visible to this process; refusing to run fewer ranks than asked "
                             "(--oversubscribe runs N ranks on the devices present over gloo, as a diagnostic, not as a scaling number)")
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE is {world}: the launcher's rank count and --gpus must agree")
    oversubscribed = world > n_devices
    if oversubscribed and not args.oversubscribe:
        raise SystemExit(f"bench.py: {world} ranks but only {n_devices} GPU(s) visible (pass --oversubscribe for the shared-device diagnostic run)")
    if n_devices < 1:
        raise SystemExit("bench.py: no GPU visible; the proving path has no CPU fallback")
    device_index = local_rank % n_devices
    rccl_world_size = None
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if oversubscribed:  # RCCL refuses two ranks on one device: the two tiny collectives go over gloo
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        rccl_world_size = dist.get_world_size()
        assert rccl_world_size == world
    torch.cuda.set_device(device_index)

    import lurk_amd
    from lurk_amd import lair, prover, shards

    if args.workload == "poseidon2":
        return poseidon2_bench(args, world, rank, device_index, distributed)
    if args.split == "intra":
        return split_intra(args, world, rank, device_index, distributed, oversubscribed)
    ctx = lurk_amd.Context(device_index)
    if args.profile != "default":
        from lurk_amd.profile import ProtocolProfile

        ProtocolProfile.preset(args.profile).install(ctx)
    log_rows = args.log_rows if args.log_rows is not None else (18 if args.workload == "lurk-mix" else LOG_ROWS)
    n = 1 << log_rows
    if args.eval_rows:
        if world != 1 or args.shards_per_rank not in (None, 1):
            raise SystemExit("--eval-rows is for one GPU and one shard")
        n, log_rows = args.eval_rows, max(1, (args.eval_rows - 1).bit_length())
        args.no_host_pipeline = args.no_second_profile = True
    dev = "cuda" if distributed and not oversubscribed else "cpu"

    # ---- host side, once: execute the ONE program (on rank 0, or on every rank with --execute-on all), flatten this rank's shards into HBM
    source, lurk_chips, entry, main_args, eval_name, workload_desc = build_workload(args.workload, world, log_rows, args.eval_rows)
    scatter = distributed and world > 1 and args.execute_on == "rank0"
    executes = not scatter or rank == 0
    top = lair.Toplevel(source, lurk_chips=lurk_chips)
    spr = args.shards_per_rank if args.shards_per_rank is not None else (2 if world > 1 else 1)
    assert spr >= 1 and n % spr == 0
    queries, all_shards, host = None, None, None
    if executes:
        t0 = time.perf_counter()
        queries = lair.QueryRecord(top)
        top.execute(top.func_index(entry), main_args, queries)
        t_execute = time.perf_counter() - t0
        eval_rows_total = queries.num_func_queries(top.func_index(eval_name))
        assert eval_rows_total == world * n, (eval_rows_total, world, n)
        host = {"pv": queries.expect_public_values(), "t_execute": t_execute,
                "host_queries": sum(queries.num_func_queries(i) for i in range(top.num_funcs())),
                "host_mem_cells": sum(queries.num_mem_queries(ml) for ml in (2, 3, 4, 5, 6, 8))}
    if scatter:  # the public values first: the machine's entrypoint chip is sized by them
        box = [host["pv"] if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        pv = box[0]
    else:
        pv = host["pv"]
    machine = prover.Machine(ctx, top, entry, len(pv))
    vk_root = machine.setup()
    if executes:
        # Shards.  One GPU: the execution is one shard.  Several: `Shard::shard` cuts every chip at the same row count, so the first
        # shards hold all the chips and the last ones only the eval chip -- with one shard per rank, rank 0 would carry 2.3x the
        # average.  The execution is cut into k = shards_per_rank shards per rank (2^log_rows / k eval rows each) and the shards are
        # dealt to the ranks by their work (shards.assign_shards_balanced: heaviest first, k per rank).
        if args.shards_total:
            # any number of shards (`Shard::shard`: ceil(rows / max_shard_size), /root/reference/src/lair/execute.rs:186-216), e.g. 9 on 8 ranks:
            # the ranks then hold different numbers of shards and the root exchange gathers the counts first (lurkhip_exchange_roots_var)
            shard_rows = -(-world * n // args.shards_total)
            all_shards = lair.Shard.new(queries).shard(lair.ShardingConfig(shard_rows))
            assert len(all_shards) == args.shards_total, f"{len(all_shards)} shards of {shard_rows} rows, {args.shards_total} asked for"
        elif world > 1 or spr > 1 or args.workload != "eval-only":
            all_shards = lair.Shard.new(queries).shard(lair.ShardingConfig(n // spr))
            assert len(all_shards) == world * spr, f"{len(all_shards)} shards for {world} ranks x {spr}: the eval chip must be the tallest"
        else:
            all_shards = [lair.Shard.new(queries)]
        host["shard_cost"] = [float(machine.shard_cost(sh)) for sh in all_shards]
        host["assignment"] = shards.assign_shards_balanced(host["shard_cost"], world)
    if scatter:
        box = [host if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        if rank != 0:
            host = dict(box[0], t_execute=0.0)  # (this rank ran no interpreter)
    t_execute, host_queries, host_mem_cells = host["t_execute"], host["host_queries"], host["host_mem_cells"]
    shard_cost, assignment = host["shard_cost"], host["assignment"]
    n_shards = len(shard_cost)
    mine = assignment[rank]
    if any(not a for a in assignment):
        # decided by EVERY rank from the broadcast assignment: a rank that left alone would leave its peers inside the next collective
        # (scatter_prepared / bring_up) until the process group's timeout
        raise SystemExit(f"{n_shards} shards over {world} ranks: rank(s) {[r for r, a in enumerate(assignment) if not a]} would hold none, nothing to time there")
    t0 = time.perf_counter()
    my_blobs = shards.scatter_prepared(machine, all_shards, assignment, device=dev) if scatter else {}

    def prepare_on(mach, i):
        """shard i's kernel inputs on machine `mach`: from the QueryRecord where this rank has it, else from the bytes rank 0 sent"""
        return mach.prepare_shard(all_shards[i]) if executes else mach.import_prepared(my_blobs[i], pv)

    prepared_all = [prepare_on(machine, i) for i in mine]
    prepared = prepared_all[0]
    t_flatten = time.perf_counter() - t0
    # once per machine, before the timed region: the big chips' AIR programs compiled to straight-line device code
    t_jit = time.perf_counter()
    compiled = []
    if not args.no_compile:
        for pr in prepared_all:
            compiled += [c for c in machine.compile_airs(pr, min_log_rows=args.compile_min_log_rows) if c not in compiled]
    t_jit = time.perf_counter() - t_jit
    # the headline is quoted with every chip on compiled kernels: a chip that silently stayed on the interpreter (hiprtc failure) would
    # make the number mean something else
    if not args.no_compile and getattr(machine, "compile_failures", None):
        raise SystemExit("bench.py: kernels that were to be compiled were not: " + "; ".join(f"{w}: {e}" for w, e in machine.compile_failures))
    chips_desc = [f"{air.name}:2^{lg}x{air.width}" for _, air, lg, _, _ in prepared]
    input_bytes = sum(p.input_bytes for pr in prepared_all for *_, p in pr if p is not None)
    main_cols_per_eval_row = sum(air.width << lg for pr in prepared_all for _, air, lg, _, _ in pr) / n
    perm_cols_per_eval_row = sum(4 * air.permutation_width << lg for pr in prepared_all for _, air, lg, _, _ in pr) / n
    constraints_per_eval_row = sum(air.num_constraints << lg for pr in prepared_all for _, air, lg, _, _ in pr) / n
    multi = world > 1 or n_shards > 1
    # Machine proofs in flight per rank.  Two (each on its own communicator and host thread) is the measured optimum on one device,
    # but across communicators nothing orders the collectives between ranks, and RCCL documents that as a deadlock hazard: the
    # collective kernels block, and a device-wide synchronisation on one thread (a hipMalloc / hipFree of either allocator) can wait
    # on the other communicator's kernel while the peer rank is in the mirror state.  It has only ever run at world 1 on RCCL and over
    # gloo, so with world > 1 on RCCL the default is ONE proof in flight until a multi-GPU box has validated two;
    # --rank-in-flight 2 asks for it explicitly.
    rccl_multi = distributed and world > 1 and not oversubscribed
    in_flight = 1 if (args.rank_pipeline or not multi) else (args.rank_in_flight if args.rank_in_flight is not None else (1 if rccl_multi else 2))
    one_lane = (args.rank_pipeline and args.rank_pipeline_one_lane) or (in_flight >= 2 and args.rank_in_flight_lanes == 1)
    lane_ctx = prover.lane_context(machine) if len(mine) > 1 and not one_lane else None  # the second proving lane of a rank with several shards

    # the step itself lives in lurk_amd/shards.py (RankStep) so that the multi-process tests run exactly what is timed here
    # the two collectives of a step behind the C ABI (lurkhip_exchange_roots / lurkhip_reduce_sums on RCCL, csrc/comm.cpp) whenever the
    # process group is RCCL's: what a Rust host drives; two ranks on one device (the oversubscribed test mode) keep gloo
    comm = None
    rccl_library = None
    cabi_fallback = None  # why the collectives run through torch.distributed although the C-ABI route was asked for
    if distributed and not oversubscribed and not args.torch_collectives:
        from lurk_amd.comm import bring_up
        from lurk_amd.comm import library as rccl_library_of

        # brought up collectively, with a self-test, BEFORE the timed region: if any rank cannot (loader, ncclCommInitRank, a wrong
        # word), every rank falls back to torch.distributed together and the line says why
        comm, cabi_fallback = bring_up(ctx)
        if comm is not None:
            rccl_library = rccl_library_of()  # which librccl the C ABI bound: the copy torch has mapped (one RCCL per process)
    rank_step = shards.RankStep(machine, vk_root, pv, prepared_all, mine, args.queries, args.pow_bits, device=dev, lane_ctx=lane_ctx, comm=comm,
                                n_shards=n_shards)
    grand_sums, rank_sums, host_ms = rank_step.grand_sums, rank_step.rank_sums, rank_step.host_ms

    def step():
        proofs = rank_step()
        return np.concatenate(proofs) if len(proofs) > 1 else proofs[0]

    # Ranks with several shards (the N > 1 schedule): a second machine on its own contexts, so that phase 1 of machine proof j + 1
    # (traces + main commitments: throughput-bound) runs under phase 2 of proof j (latency chains, and a light last shard whose
    # lane idles early) -- shards.run_pipelined.  All collectives stay on this thread, in the same order on every rank.
    pipe = None
    if args.rank_pipeline or in_flight >= 2:
        pipe = {"steps": [rank_step], "ctxs": [], "machines": [], "prepared": [], "comms": []}
        for _ in range((in_flight if in_flight >= 2 else max(2, args.rank_pipeline_depth)) - 1):
            ctx_b = lurk_amd.Context(device_index) if os.environ.get("LURKHIP_LANE_UNPLACED") else lurk_amd.Context(beside=ctx)
            if args.profile != "default":
                from lurk_amd.profile import ProtocolProfile

                ProtocolProfile.preset(args.profile).install(ctx_b)
            machine_b = prover.Machine(ctx_b, top, entry, len(pv))
            assert machine_b.setup() == vk_root
            prepared_b = [prepare_on(machine_b, i) for i in mine]
            if not args.no_compile:
                for pr in prepared_b:
                    machine_b.compile_airs(pr, min_log_rows=args.compile_min_log_rows)
            lane_ctx_b = prover.lane_context(machine_b) if (len(mine) > 1 and not one_lane) else None
            comm_b, group_b = comm, None
            if in_flight >= 2 and distributed:
                # a machine proof in flight issues its collectives from its own thread: its own communicator (created here, in the same
                # order on every rank), on its own context's stream
                if comm is not None:
                    comm_b, why_b = bring_up(ctx_b)
                    if comm_b is None:  # (agreed by every rank) the whole run goes back to torch.distributed, one group per machine proof in flight
                        cabi_fallback = why_b
                        for c_old in [comm] + pipe["comms"]:
                            c_old.close()
                        comm, pipe["comms"], rccl_library = None, [], None
                        for k_, st_ in enumerate(pipe["steps"]):
                            st_.comm = None
                            st_.group = None if k_ == 0 else dist.new_group(backend=None)
                        group_b = dist.new_group(backend=None)
                    else:
                        pipe["comms"].append(comm_b)
                else:
                    group_b = dist.new_group(backend="gloo" if oversubscribed else None)
            pipe["steps"].append(shards.RankStep(machine_b, vk_root, pv, prepared_b, mine, args.queries, args.pow_bits, device=dev, lane_ctx=lane_ctx_b, comm=comm_b,
                                                 n_shards=n_shards, group=group_b))
            pipe["ctxs"] += [c for c in (ctx_b, lane_ctx_b) if c is not None]
            pipe["machines"].append(machine_b)
            pipe["prepared"].append(prepared_b)
        if in_flight >= 2:
            pipe["stagger_s"] = 0.0
            pipe["run"] = lambda steps_, n_, on_proofs=None: shards.run_in_flight(steps_, n_, on_proofs=on_proofs, stagger_s=pipe["stagger_s"])
        else:
            pipe["run"] = shards.run_pipelined if len(pipe["steps"]) == 2 else shards.run_committed_ahead

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # The interpreter's cyclic collector: `import torch` alone leaves 170 k container objects behind, and the first full
    # collection over them is a 40-50 ms pause that landed inside one timed step (always the 14th with 3 warm-up steps: one step
    # of 84-126 ms among 46.9 ms ones).  The set-up's garbage is collected here and what survives is moved to the permanent
    # generation; the collector stays enabled for everything allocated from now on.
    gc.collect()
    gc.freeze()
    fence()  # (the first barrier of a process group sets RCCL up lazily; its aftermath showed up as a 90 ms first timed step)
    for _ in range(args.warmup):
        step()
    if pipe is not None:  # warm the second machine and the pipeline's worker
        pipe["run"](pipe["steps"], len(pipe["steps"]))
        if in_flight >= 2:  # the proofs in flight run out of phase: lane t starts t / K of one machine proof's own time late
            t_w = time.perf_counter()
            step()
            pipe["stagger_s"] = (time.perf_counter() - t_w) / in_flight
        for cx_ in pipe["ctxs"]:
            cx_.sync()
    fence()
    # N = 1, --lanes 2 (default): the K timed steps are K independent proofs of the shard with TWO IN FLIGHT, each lane on its own
    # context (HIP stream, pool, host thread) -- while one proof sits in a latency chain (tree tails, FRI layers, transcript round
    # trips) the other's big kernels fill the device; this is how Machine.prove runs every multi-shard proof and how a rank of the
    # N > 1 bench proves its two shards.  A short one-proof-at-a-time pass comes first, for the stage table, the live roofline
    # spans (a kernel alone on the device) and the `sequential` number reported beside the headline.
    lanes = args.lanes if (world == 1 and spr == 1 and args.lanes >= 2 and not args.no_two_in_flight) else 1
    sequential = None
    seq_spans = None
    lane2 = None
    if lanes >= 2:
        import threading

        # three to five proofs, and for short proofs as many as fill a fifth of a second (at most 40): one hiccup in five 7 ms
        # steps moved a 2^12-row latency by a millisecond
        n_seq_min, n_seq = max(3, min(5, args.steps)), 0
        ctx.profile_reset()
        ctx.profile_enable(not args.no_spans)
        t_q = time.perf_counter()
        while n_seq < n_seq_min or (n_seq < 40 and time.perf_counter() - t_q < 0.2):
            words_seq = step()
            n_seq += 1
        fence()
        dt_seq = time.perf_counter() - t_q
        ctx.profile_enable(False)
        seq_spans = {name: ctx.profile_read(name) for name in SPANS}
        sequential = {"steps": n_seq, "ms_per_step": dt_seq / n_seq * 1e3, "eval_steps_per_s": n * n_seq / dt_seq,
                      "stages_ms": {k: v[0] / n_seq for k, v in seq_spans.items() if v[1]},
                      "note": "one proof at a time on one HIP stream (the headline of rounds 1-2), measured in this run before the timed region"}
        extra_lanes = []
        for _ in range(lanes - 1):
            # a context whose stream is measured to run beside the first lane's (lurkhip_ctx_create_beside); LURKHIP_LANE_UNPLACED=1
            # takes whatever hardware queue the runtime hands out (A/B)
            ctx2 = lurk_amd.Context(device_index) if os.environ.get("LURKHIP_LANE_UNPLACED") else lurk_amd.Context(beside=ctx)
            if args.profile != "default":
                from lurk_amd.profile import ProtocolProfile

                ProtocolProfile.preset(args.profile).install(ctx2)
            m2 = prover.Machine(ctx2, top, entry, len(pv))
            vk2 = m2.setup()
            assert vk2 == vk_root
            prep2 = m2.prepare_shard(all_shards[0])
            if not args.no_compile:
                m2.compile_airs(prep2, min_log_rows=args.compile_min_log_rows)  # same programs: served from the in-process code cache
            extra_lanes.append((m2, ctx2, prep2))
        lane2 = extra_lanes

        def one(mach, cx, prep):
            cx.span_begin("trace_all")
            traces = mach.run_prepared(prep)
            cx.span_end("trace_all")
            handle, root = mach.commit_shard(traces)
            ch = prover.Challenger(cx)
            ch.observe(vk_root)
            ch.observe([0])
            ch.observe(root)
            ch.observe(pv)
            w = mach.prove_shard(handle, ch, pv, num_queries=args.queries, pow_bits=args.pow_bits, parse=False)
            mach.free_shard(handle)
            cs = shards.proof_cumulative_sums(w)
            tot = np.zeros(4, dtype=np.int64)
            for c in cs:
                tot = (tot + np.asarray(c, dtype=np.int64)) % 2013265921
            return w, tuple(int(x) for x in tot)

        # the lanes' offset: by default a proof's sequential time (just measured) divided by the number of lanes
        # (the delay is paid once per timed region: with fewer than 16 timed proofs it costs more than the phase shift returns)
        stagger_ms = args.stagger_ms if args.stagger_ms is not None else (sequential["ms_per_step"] / lanes if args.steps >= 16 else 0.0)

        def run_lanes(k_total, sink):
            """k_total proofs over the two lanes, each lane taking the next proof as it finishes one."""
            nxt = [0]
            lock = threading.Lock()
            errors = []

            def worker(mach, cx, prep, delay_s=0.0):
                try:
                    if delay_s:
                        time.sleep(delay_s)
                    while True:
                        with lock:
                            if nxt[0] >= k_total:
                                break
                            nxt[0] += 1
                        sink.append(one(mach, cx, prep))
                    cx.sync()
                except BaseException as e:  # surfaced after the join
                    errors.append(e)

            ths = [threading.Thread(target=worker, args=(machine, ctx, prepared))]
            ths += [threading.Thread(target=worker, args=l + ((k + 1) * stagger_ms * 1e-3,)) for k, l in enumerate(extra_lanes)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            if errors:
                raise errors[0]

        # placement of the lanes' streams (a query, independent of load): a busy kernel alone on the main stream against one on each
        # lane's stream together -- lanes on a shared hardware queue would read twice the time and never overlap their proofs
        lane_placement = []
        for _, cx_, _ in extra_lanes:
            alone_s, both_s = ctx.overlap_probe(cx_)
            lane_placement.append({"alone_ms": alone_s * 1e3, "both_ms": both_s * 1e3, "streams_run_beside_each_other": bool(both_s < 1.5 * alone_s)})
        run_lanes(2 * lanes, [])  # warm every lane (pools, tables)
        fence()
        for _, cx_, _ in extra_lanes:
            cx_.sync()
    ctx.profile_reset()
    ctx.profile_enable(not args.no_spans)
    if lane_ctx is not None:  # the second proving lane's stages count too
        lane_ctx.profile_reset()
        lane_ctx.profile_enable(not args.no_spans)
    for _, cx_, _ in (lane2 or []):
        cx_.profile_reset()
        cx_.profile_enable(not args.no_spans)
    for cx_ in (pipe["ctxs"] if pipe else []):
        cx_.profile_reset()
        cx_.profile_enable(not args.no_spans)
    # every context that works in the timed region: no device allocation may happen there (a hipMalloc synchronises the device)
    timed_ctxs = [("main", ctx)] + ([("lane", lane_ctx)] if lane_ctx is not None else [])
    if pipe is not None:
        timed_ctxs += [(f"pipe{k}", c) for k, c in enumerate(pipe["ctxs"])]
    if lane2:
        timed_ctxs += [(f"proof_lane{k + 1}", c) for k, (_, c, _) in enumerate(lane2)]
    mallocs_before = {k: c.pool_stats()["mallocs"] for k, c in timed_ctxs}
    profiler_region(True)
    t0 = time.perf_counter()
    words = None
    step_words = []
    step_ms = []
    if lanes >= 2:
        results = []
        run_lanes(args.steps, results)
        for _, cx_, _ in lane2:
            cx_.sync()
        step_words = [w for w, _ in results] + [words_seq]
        words = step_words[0]
        grand_sums += [g for _, g in results]
    elif pipe is not None:
        def keep(j, proofs):
            step_words.append(np.concatenate(proofs) if len(proofs) > 1 else proofs[0])

        pipe["run"](pipe["steps"], args.steps, on_proofs=keep)
        for cx_ in pipe["ctxs"]:
            cx_.sync()
        words = step_words[-1]
    else:
        for _ in range(args.steps):
            t_s = time.perf_counter()
            words = step()
            step_ms.append((time.perf_counter() - t_s) * 1e3)
            step_words.append(words)  # compared after the timed region: the same shard must give the same proof every step
    fence()
    elapsed = time.perf_counter() - t0
    profiler_region(False)
    pool_after = {k: c.pool_stats() for k, c in timed_ctxs}
    pool_report = {"hipMalloc_calls_in_timed_region": {k: v["mallocs"] - mallocs_before[k] for k, v in pool_after.items()},
                   "peak_bytes": {k: v["peak_bytes"] for k, v in pool_after.items()}}
    if pipe is not None:
        # the second machine's records count too (collectives' host time, sums); the gathered-set check below takes the LAST proof
        for other in pipe["steps"][1:]:
            grand_sums += other.grand_sums
            rank_sums += other.rank_sums
            for k_, v_ in other.host_ms.items():
                host_ms[k_] = host_ms.get(k_, 0.0) + v_
        rank_step = pipe["steps"][(args.steps - 1) % len(pipe["steps"])]
    proofs_identical = all(len(w) == len(step_words[0]) and bool((w == step_words[0]).all()) for w in step_words[1:])
    del step_words
    ctx.profile_enable(False)
    rank_ms = elapsed / args.steps * 1e3
    per_rank_ms = [rank_ms]
    all_rank_sums_nonzero = all(s != (0, 0, 0, 0) for s in rank_sums)
    exec_s_per_rank = [t_execute]
    import resource

    rss_mb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0  # this rank's peak resident set (the whole query record lives in it)
    rss_per_rank = [rss_mb]
    if distributed:
        tdev = "cuda" if dev == "cuda" else "cpu"
        t = torch.tensor([elapsed, t_execute, rss_mb], dtype=torch.float64, device=tdev)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank_ms = [float(g[0].item()) / args.steps * 1e3 for g in gathered]
        exec_s_per_rank = [float(g[1].item()) for g in gathered]
        rss_per_rank = [float(g[2].item()) for g in gathered]
        elapsed = max(float(g[0].item()) for g in gathered)
        flag = torch.tensor([1 if all_rank_sums_nonzero else 0], dtype=torch.int64, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        all_rank_sums_nonzero = bool(flag.item())
    # The proofs of the last step as a SET, outside the timed region: every rank's proof words travel to rank 0, which checks
    # that the shard indices are a partition, that proof s carries the main root the ranks all-gathered for shard s (the root
    # every transcript observed in position s), and that the cumulative sums of the gathered proofs cancel.  (The oracle's
    # verifier accepting such a gathered set is tests/test_multirank_gpu.py; the bench does not import the oracle here.)
    last = rank_step.last_proofs
    proof_set = None
    got = shards.gather_proofs(last, mine, dst=0)
    if got is not None:
        tot = np.zeros(4, dtype=np.int64)
        roots_ok = True
        for sidx, w in enumerate(got):
            n_chips, n_public = int(w[1]), int(w[5])
            at = 10 + 11 * n_chips + n_public
            roots_ok = roots_ok and [int(x) for x in w[at:at + 8]] == rank_step.roots[sidx]
            for c in shards.proof_cumulative_sums(w):
                tot = (tot + np.asarray(c, dtype=np.int64)) % 2013265921
        proof_set = {"shards_gathered_on_rank0": len(got), "main_roots_match_exchanged_roots_in_shard_order": bool(roots_ok),
                     "grand_sum_of_gathered_proofs_is_zero": bool((tot == 0).all()), "proof_words_total": int(sum(len(w) for w in got))}
        # ... and the product's own verifier (csrc/verify.cpp, host only: transcript, every Merkle opening, every FRI query, the
        # constraint identity of every chip at zeta, the cumulative sums) on the gathered set -- the reference's `fib-verification`
        # stage (benches/fib.rs:105-133); outside the timed region
        from lurk_amd.profile import ProtocolProfile

        t_v = time.perf_counter()
        try:
            accepted = bool(machine.verify(got, profile=ProtocolProfile.of(ctx)))
            why = None
        except prover.VerificationError as e:
            accepted, why = False, str(e)
        proof_set["product_verifier"] = {"accepted": accepted, "reason": why, "host_verify_s": time.perf_counter() - t_v,
                                         "note": "lurkhip_machine_verify on one host thread, every shard proof of the last step"}

    spans = {name: ctx.profile_read(name) for name in SPANS}
    if pipe is not None:
        for cx_ in pipe["ctxs"]:
            cx_.profile_enable(False)
            for name in SPANS:
                ms, cnt = cx_.profile_read(name)
                spans[name] = (spans[name][0] + ms, spans[name][1] + cnt)

    for _, cx_, _ in (lane2 or []):
        cx_.profile_enable(False)
        for name in SPANS:
            ms, cnt = cx_.profile_read(name)
            spans[name] = (spans[name][0] + ms, spans[name][1] + cnt)
    if lane_ctx is not None:
        lane_ctx.profile_enable(False)
        for name in SPANS:
            ms, cnt = lane_ctx.profile_read(name)
            spans[name] = (spans[name][0] + ms, spans[name][1] + cnt)
    # N > 1: what makes the first multi-GPU run diagnosable from its one line -- every rank's stage spans, the work the assignment
    # gave each rank (Machine.shard_cost, the quantity assign_shards_balanced equalises) and the efficiency that predicts
    per_rank_stages = None
    assignment_cost = None
    if distributed:
        mine_stages = {k: v[0] / args.steps for k, v in spans.items() if v[1]}
        boxes = [None] * world
        dist.all_gather_object(boxes, mine_stages)
        per_rank_stages = boxes
        cost = shard_cost
        per_rank_cost = [sum(cost[i] for i in a_) for a_ in assignment]
        mean_cost = sum(per_rank_cost) / len(per_rank_cost)
        # the rank schedule (two half-shards on two lanes) measured 46.4 ms against 40.9 ms for one shard with two proofs in flight on
        # one GPU (round 3): 0.88 before any communication; the slowest rank sets the step
        sched = 0.88 if world > 1 else 1.0
        assignment_cost = {"per_rank_cost": per_rank_cost, "predicted_imbalance_max_over_mean": max(per_rank_cost) / mean_cost,
                           "schedule_factor_vs_n1": sched, "expected_efficiency_vs_n1": sched * mean_cost / max(per_rank_cost),
                           "note": "cost = Machine.shard_cost (columns x rows of the shard's chips); expected efficiency = schedule factor x mean / max rank cost, "
                                   "before the two collectives (tens of bytes per shard: host_ms below)"}
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n * args.steps / elapsed

    # dominant kernels: the Merkle hashing launches (k_row_sponges: the sponges of the leaves and of the shorter matrices injected
    # at their levels, one launch per tree; k_level_digests / k_level_coop: the 2-to-1 compressions).  Algorithmic bytes: a leaf row reads its w*4 bytes and
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

    def rounds_of(pr):
        rs = [
            [(lg + LOG_BLOWUP, air.width) for _, air, lg, _, _ in pr],
            [(lg + LOG_BLOWUP, 4 * air.permutation_width) for _, air, lg, _, _ in pr],
            [(lg + LOG_BLOWUP, 4) for _, air, lg, _, _ in pr for _ in range(1 << air.log_quotient_degree)],
        ]
        max_lg = max(lg for _, _, lg, _, _ in pr) + LOG_BLOWUP
        # FRI layers of 2^16 leaves and more (smaller trees are not in the hashing spans)
        return rs + [[(lf, 8)] for lf in range(max_lg - 1, 15, -1)]

    rounds_all = [rounds_of(pr) for pr in prepared_all]  # per shard of this rank
    rounds = rounds_all[0]
    hash_bytes_step = sum(merkle_hash_bytes(r)[0] for rs in rounds_all for r in rs)
    hash_launches_step = sum(merkle_hash_bytes(r)[1] for rs in rounds_all for r in rs)
    # the live roofline spans: the kernels alone on the device (the sequential pass when two proofs are in flight in the timed region)
    rs, rs_steps = (seq_spans, sequential["steps"]) if seq_spans is not None else (spans, args.steps)
    hash_ms_step = (rs["merkle_leaves"][0] + rs["merkle_levels"][0]) / rs_steps
    achieved = hash_bytes_step / (hash_ms_step * 1e-3) / 1e9 if hash_ms_step > 0 else 0.0
    # second-largest kernel family, the coset LDE passes (k_ntt_pass).  SURVEY.md 8(d) prices an LDE (blow-up 2) at read 4w +
    # write 8w bytes per trace row = 12 w B/row: that is `algorithmic`.  `pass_traffic` is what the implementation moves:
    # every pass reads and writes its matrix once, a size-N transform takes ceil(log N / NTT_LOG_TILE) passes, an LDE is three
    # transforms.
    lde_alg_bytes, lde_pass_bytes = 0, 0
    ntt_log_tile = int(os.environ.get("LURKHIP_NTT_MAX_LOG_R", "10"))  # rows of the tallest LDS tile: 2^10 (ntt.hip)
    for r in (r for rs in rounds_all for r in rs[:3]):
        for lg, w in r:
            log_n = lg - LOG_BLOWUP
            lde_alg_bytes += 12 * w * (1 << log_n)
            # transfers of the implementation: the grouped route (2^5 .. 2^20 rows) moves a matrix 9 times above 2^10 rows (k_in r + w,
            # k_mid r + 2 w, k_out 2 r + 2 w) and 3 times below (one fused kernel); ntt.hip's three transforms r + w per pass otherwise
            if 5 <= log_n <= 20:
                lde_pass_bytes += (9 if log_n > 10 else 3) * (1 << log_n) * w * 4
            else:
                lde_pass_bytes += 3 * max(1, -(-log_n // ntt_log_tile)) * 2 * (1 << log_n) * w * 4
    lde_ms_step = rs["lde"][0] / rs_steps
    # ... of which transformed: the permutation traces' identically-zero columns are left out of the LDE (their 12 w bytes stay in the
    # algorithmic figure -- a property of the commitment, not of the route); the library says how many cells the last proof transformed
    lde_alg_transformed = lde_alg_bytes
    try:
        pst = np.zeros(2, dtype=np.uint64)
        lurk_amd._native.lib.lurkhip_prover_stats(ctx.handle, pst.ctypes.data)
        if int(pst[0]):
            lde_alg_transformed = lde_alg_bytes - 12 * (int(pst[0]) - int(pst[1])) * len(mine)
    except Exception:
        pass
    lde_alg = lde_alg_bytes / (lde_ms_step * 1e-3) / 1e9 if lde_ms_step > 0 else 0.0
    lde_pass = lde_pass_bytes / (lde_ms_step * 1e-3) / 1e9 if lde_ms_step > 0 else 0.0
    traffic, valu, lde_traffic, lde_valu = None, None, None, None
    try:  # HBM bytes / instruction counts of the same kernels from committed rocprofv3 PMC passes (NOT measured in this run)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("log_rows") == log_rows and pmc.get("workload") == args.workload:
            src = f"profiles (static): {pmc.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE')}; {pmc.get('correction', '')}"
            traffic = {"bytes_per_step": pmc["merkle_hash_bytes_per_step"], "over_algorithmic": pmc["merkle_hash_bytes_per_step"] / hash_bytes_step if hash_bytes_step else None,
                       "source": src,
                       "note": "the re-fetches are L2 misses on lines shared by four consecutive 32-byte chunk loads of a lane's row, about 20 k cycles apart; the counter includes Infinity-Cache hits; the kernel is instruction-bound, not waiting for them (SQ_WAIT_ANY 5 %)"}
            lde_traffic = {"bytes_per_step": pmc["ntt_pass_bytes_per_step"], "over_algorithmic": pmc["ntt_pass_bytes_per_step"] / lde_alg_bytes if lde_alg_bytes else None,
                           "over_pass_model": pmc["ntt_pass_bytes_per_step"] / lde_pass_bytes if lde_pass_bytes else None, "source": src}
            # the LDE is not a pure streaming kernel: 41 butterfly levels of 32-bit modular arithmetic per element are ~8 lane-instructions
            # per byte of its nine transfers (half of them multiply-class, at half rate: ~12 full-rate equivalents) against the machine's
            # 78.6 T / 8 TB/s = 9.8 -- its instruction rate is reported beside its bytes
            if pmc.get("lde_valu_lane_insts_per_step") and lde_ms_step > 0 and not step_sources_changed(pmc):
                li = pmc["lde_valu_lane_insts_per_step"] * len(mine)
                lde_valu = {"valu_lane_insts_per_step": li, "achieved": li / (lde_ms_step * 1e-3) / 1e12, "unit": "T lane-instr/s", "peak": VALU_FULL_RATE,
                            "frac": li / (lde_ms_step * 1e-3) / 1e12 / VALU_FULL_RATE,
                            "lane_insts_per_pass_byte": li / lde_pass_bytes if lde_pass_bytes else None, "machine_balance_insts_per_byte": VALU_FULL_RATE * 1e12 / (HBM_PEAK_GBS * 1e9),
                            "source": "profiles/pmc_traffic.json: SQ_INSTS_VALU x 64 lanes of the k_lde_* / k_ntt_pass launches of one step (static) / the `lde` spans' time of this run (live)"}
            mul_frac = pmc.get("merkle_hash_mul_class_frac", 0.6)
            # instruction-mix ceiling: add-class at the full rate, mul-class at half rate, no overlap between the classes
            ceiling = 1.0 / ((1 - mul_frac) / VALU_FULL_RATE + mul_frac / VALU_HALF_RATE)
            # `achieved` is LIVE: the hashing launches' instruction count is a property of the code and the shapes (deterministic; the
            # committed PMC pass counted it: SQ_INSTS_VALU x 64 lanes over k_row_sponges + the level kernels of one step), their time
            # is measured in THIS run (HIP events on the library's stream, the one-proof-at-a-time pass)
            insts = pmc.get("merkle_hash_valu_lane_insts_per_step")
            # the count is a property of the hashing kernels' code: it is only used while the sources the PMC pass ran on are the
            # sources of this tree (a changed kernel would have been priced with a stale count, silently)
            stale = False
            if pmc.get("hash_kernel_sources_sha256"):
                import hashlib

                hh = hashlib.sha256()
                for rel in pmc.get("hash_kernel_sources", []):
                    with open(os.path.join(ROOT, rel), "rb") as fsrc:
                        hh.update(fsrc.read())
                stale = hh.hexdigest() != pmc["hash_kernel_sources_sha256"]
            if stale:
                insts = None
            if insts and hash_ms_step > 0:
                # (the count is proportional to the words hashed: scaled from the PMC pass's shapes -- one 2^20-row shard -- to this
                # rank's shards by the algorithmic bytes of the hashing launches)
                ref_bytes = pmc.get("merkle_hash_algorithmic_bytes_per_step")
                scale = hash_bytes_step / ref_bytes if ref_bytes else float(len(mine))
                live = insts * scale / (hash_ms_step * 1e-3) / 1e12
                src_v = ("instruction count from the committed PMC pass (profiles/pmc_traffic.json: SQ_INSTS_VALU x 64 lanes of the hashing launches of one step, "
                         "deterministic) / the hashing launches' HIP-event time measured live in this run")
            else:
                live, src_v = pmc["merkle_hash_valu_tinst_s"], ("profiles (static): SQ_INSTS_VALU x 64 lanes / kernel time of the PMC pass"
                                                                + (" -- the hashing kernels' sources have changed since that pass: its instruction count is not applied to this run's time" if stale else ""))
            valu = {"achieved": live, "unit": "T lane-instr/s", "peak_full_rate": VALU_FULL_RATE, "peak_half_rate": VALU_HALF_RATE,
                    "mul_class_frac": mul_frac, "mix_ceiling": ceiling, "frac_of_mix_ceiling": live / ceiling,
                    "frac_of_full_rate": live / VALU_FULL_RATE, "static_pmc_rate": pmc["merkle_hash_valu_tinst_s"],
                    "source": src_v}
    except Exception:
        pass

    # The whole step against the VALU peak: every kernel's SQ_INSTS_VALU x 64 lanes of one step from the
    # committed PMC pass / this run's ms_per_step.  The count belongs to the sources of that pass: flagged when any of them changed.
    step_block = None
    try:
        if pmc.get("log_rows") == log_rows and pmc.get("workload") == args.workload and pmc.get("step_valu_lane_insts"):
            step_block = {"valu_lane_insts_per_step": pmc["step_valu_lane_insts"], "unit": "T lane-instr/s", "peak": VALU_FULL_RATE,
                          "sources_changed_since_pmc_pass": step_sources_changed(pmc),
                          "source": "profiles/pmc_traffic.json: SQ_INSTS_VALU x 64 lanes summed over every kernel of one step (static) / ms_per_step of this run (live, "
                                    "filled in below)"}
    except Exception:
        step_block = None

    # Extra (N = 1 only, never `value`): the host side of the path.  ONE execution of pipeline_shards x 2^log_rows eval rows,
    # sharded; (a) every shard's inputs staged beforehand (the resident-input reference), (b) streamed: a staging thread
    # flattens shard k + 1 on host threads into page-locked memory and uploads it on a second context while this context
    # commits shard k (prover.prove_streamed).  Same proofs; execute is reported separately and is in neither timing.
    host_pipeline = None
    if world == 1 and spr == 1 and not args.no_host_pipeline and args.workload != "eval-only":
        try:
            S = args.pipeline_shards
            src2, _, entry2, args2, _, _ = build_workload(args.workload, S, log_rows)
            t1 = time.perf_counter()
            top2 = lair.Toplevel(src2, lurk_chips=True)
            q2 = lair.QueryRecord(top2)
            top2.execute(top2.func_index(entry2), args2, q2)
            t_exec2 = time.perf_counter() - t1
            pv2 = q2.expect_public_values()
            mach2 = prover.Machine(ctx, top2, entry2, len(pv2))
            mach2.setup()
            cfg2 = lair.ShardingConfig(n)
            shards2 = lair.Shard.new(q2).shard(cfg2)
            t1 = time.perf_counter()
            prepared2 = [mach2.prepare_shard(sh) for sh in shards2]
            t_stage = time.perf_counter() - t1
            if not args.no_compile:
                mach2.compile_airs(prepared2[0])  # same chips as above: served from the code cache
            staged_bytes = sum(p.input_bytes for item in prepared2 for *_, p in item if p is not None)
            ctx_in = lurk_amd.Context(beside=ctx)

            def timed(**kw):
                torch.cuda.synchronize()
                t = time.perf_counter()
                pr = prover.prove_streamed(mach2, q2, cfg2, num_queries=args.queries, pow_bits=args.pow_bits, parse=False, **kw)
                ctx.sync()
                torch.cuda.synchronize()
                return time.perf_counter() - t, pr

            timed(prepared=prepared2, input_ctx=ctx_in)  # warm-up (pools, tables; ctx_in is phase 2's second lane in both variants)
            t_res, ref = timed(prepared=prepared2, input_ctx=ctx_in)
            for item in prepared2:
                for *_, p in item:
                    if p is not None:
                        p.close()
            del prepared2
            st = {}
            timed(input_ctx=ctx_in)  # warm-up (page-locked staging, the second context's pool)
            t_str, got = timed(input_ctx=ctx_in, stats=st)
            same = len(ref) == len(got) and all(len(a) == len(b) and bool((a == b).all()) for a, b in zip(ref, got))
            host_pipeline = {
                "shards": len(shards2), "eval_rows": len(shards2) * n, "host_execute_s": t_exec2,
                "resident_ms_per_shard": t_res / len(shards2) * 1e3, "streamed_ms_per_shard": t_str / len(shards2) * 1e3,
                "streamed_over_resident_rate": t_res / t_str,
                "staging_s_per_shard": st.get("staging_s", 0.0) / len(shards2), "staging_threads": min(32, usable_cores()),
                "first_staging_s_per_shard": t_stage / len(shards2), "staged_bytes_per_shard": staged_bytes // len(shards2),
                "proofs_match_resident": same,
                "note": "one sharded execution, all shards' traces and main commitments resident for phase 2, which proves two shards at a time on two contexts; flatten + upload of shard k+1 run under the commit of shard k; not the headline value",
            }
            mach2.close()
            ctx_in.close()
            del q2, top2
        except Exception as e:
            host_pipeline = {"error": repr(e)}

    # N > 1 on real devices: the same ranks also prove ONE 2^log_rows shard together (--split intra: strong scaling), in CHILD processes
    # -- one per rank, their own process group on another port, a time limit -- so that whatever that path does on its first contact
    # with a multi-GPU box (RCCL send / receive pairs have never run with world > 1) cannot take this line with it.  The child's line
    # goes under config.split_intra.
    split_probe = None
    # (LURKHIP_SPLIT_PROBE_OVERSUB=1: also when the ranks share a device -- the rehearsal of the mechanism in the GPU suite)
    if (distributed and world > 1 and (not oversubscribed or os.environ.get("LURKHIP_SPLIT_PROBE_OVERSUB") == "1") and not args.no_split_probe
            and args.workload in ("fib-mix", "lurk-mix") and world & (world - 1) == 0 and os.environ.get("LURKHIP_BENCH_CHILD") != "1"):
        import subprocess

        fence()
        env = dict(os.environ, LURKHIP_BENCH_CHILD="1", MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 23))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)  # (the children rendezvous among themselves: rank 0's child hosts the store)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--split", "intra", "--steps", "6", "--warmup", "2", "--log-rows", str(log_rows),
               "--workload", args.workload, "--queries", str(args.queries), "--pow-bits", str(args.pow_bits), "--no-cpu-baseline"]
        cmd += ["--oversubscribe"] if oversubscribed else []
        cmd += (["--no-compile"] if args.no_compile else []) + ["--compile-min-log-rows", str(args.compile_min_log_rows)]
        t_probe = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.split_probe_timeout, env=env)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            split_probe = json.loads(lines[-1]) if lines else {"error": f"no line (exit code {r.returncode})", "stderr_tail": r.stderr[-600:]}
        except subprocess.TimeoutExpired:
            split_probe = {"error": f"timed out after {args.split_probe_timeout} s"}
        except Exception as e:  # a reported extra, never a reason to lose the line
            split_probe = {"error": repr(e)}
        split_probe["probe_wall_s"] = time.perf_counter() - t_probe
    if step_block is not None and ms_per_step > 0:
        step_block["achieved"] = step_block["valu_lane_insts_per_step"] * len(mine) / (ms_per_step * 1e-3) / 1e12
        step_block["frac"] = step_block["achieved"] / VALU_FULL_RATE
        step_block["ms_per_step"] = ms_per_step
    if rank == 0:
        out = {
            "metric": "Lurk eval-steps proved/sec (fib trace)",
            "value": value,
            "unit": "eval-steps/s",
            "n_gpus": world if not oversubscribed else n_devices,
            "ranks": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            # two different quantities, under two names: `value` / `ms_per_step` are THROUGHPUT of the timed schedule
            # (proofs_in_flight independent proofs of the shard overlapping on the GPU); `proof_latency_ms` is ONE proof at a time on one
            # stream -- the quantity rounds 1-2 reported as ms_per_step -- measured in the same run before the timed region
            "proofs_in_flight": lanes if world == 1 else (2 if len(mine) > 1 else 1),
            "proof_latency_ms": sequential["ms_per_step"] if sequential else (ms_per_step if lanes < 2 and pipe is None else None),
            "eval_steps_per_s_one_proof_at_a_time": sequential["eval_steps_per_s"] if sequential else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"fib trace 2^{log_rows} eval rows x 78 cols per GPU ({args.workload}) + the rest of its machine: lair trace-gen, main / LogUp permutation / quotient commits (coset LDE blow-up 2 + Poseidon2-16 Merkle), openings + FRI ({args.queries} queries, {args.pow_bits} PoW bits)"
                + (f"; one execution of {world} x 2^{log_rows} eval rows in {n_shards} shards dealt to {world} ranks by work, RCCL all-gather of shard roots + all-reduce of cumulative sums" if distributed else ""),
                "workload_detail": workload_desc,
                "chips": chips_desc,
                "main_columns_per_eval_row": main_cols_per_eval_row,
                "permutation_columns_per_eval_row": perm_cols_per_eval_row,
                "constraint_evaluations_per_eval_row": constraints_per_eval_row,
                "measured_shape": measured_shape(args.workload),
                "stages_ms": sequential["stages_ms"] if sequential else {k: v[0] / args.steps for k, v in spans.items() if v[1]},
                "parity": "Poseidon2 / traces / AIR pinned by the reference's vectors and constraint property; commit / LogUp / quotient / FRI bit-exact vs the oracle and accepted by its verifier (upstream parity unpinned: sphinx / Plonky3 sources absent, tests/golden/upstream/ takes vectors)",
                "proof_words": int(len(words)),
                "split_intra": (None if split_probe is None else
                                {k: split_probe.get(k) for k in ("error", "stderr_tail", "probe_wall_s", "value", "ms_per_step", "scaling", "n_gpus", "per_rank",
                                                                 "alltoall_bytes_per_rank_per_step", "alltoall_bytes_per_link_per_step", "xgmi_model_ms_per_step",
                                                                 "proofs_identical_on_all_ranks", "proof_verified") if k in split_probe}
                                | ({"carrier": split_probe["config"]["split"]["carrier"], "carrier_note": split_probe["config"]["split"]["carrier_note"],
                                    "note": "ONE shard of 2^log_rows eval rows proved by all ranks together (bench.py --split intra, measured by child processes of this run "
                                            "after its timed region): strong scaling -- compare ms_per_step with the N = 1 line's proof_latency_ms"}
                                   if "config" in split_probe else {})),
                "eval_rows": n,
                "workload_detail": (None if args.workload != "lurk-mix" else
                                    ("lurk-mix at the REAL height of demo/mastermind.lurk (--eval-rows 6867: every tall chip at the padded height of the reference's own run, "
                                     "tests/test_mix_programs.py::test_lurk_mix_at_the_real_mastermind_height; the REPL-sized proof of src/core/cli/repl.rs:164-207: proof_latency_ms is "
                                     "the number that matters)" if args.eval_rows else
                                     "lurk-mix scaled to 2^log_rows eval rows with the ingress-side chips at their measured (absolute) sizes: the irregular-width commitment of SURVEY.md 8(d) "
                                     "row 5, NOT the mastermind run's height (that is --eval-rows 6867)")),
                "shards": n_shards,
                "executed_on": ("rank 0 (its shards' kernel inputs sent to their owners: shards.scatter_prepared)" if scatter else "every rank") if distributed else "this process",
                "shards_per_rank": spr,
                "shard_assignment": assignment if world > 1 or spr > 1 else None,
                "rccl_world_size": rccl_world_size if not oversubscribed else None,
                "rccl_library": rccl_library,
                "c_abi_collectives_fallback": cabi_fallback,
                "collectives": None if not distributed else ("c-abi: lurkhip_exchange_roots + lurkhip_reduce_sums on RCCL, the context's stream (csrc/comm.cpp)" if comm is not None else "torch.distributed"),
                "process_group": None if not distributed else ("gloo (oversubscribed diagnostic: ranks share a device; NOT a scaling number)" if oversubscribed else "nccl (RCCL)"),
                "visible_gpus": n_devices,
                "host": host_info(),
                "protocol_profile": args.profile,
                "gathered_proof_set": proof_set,
                "grand_sum_is_zero": all(g == (0, 0, 0, 0) for g in grand_sums),
                "per_rank_sum_nonzero": all_rank_sums_nonzero if world > 1 else None,  # one rank: its own sum is the (zero) total
                "per_rank_ms_per_step": per_rank_ms,
                "per_rank_stages_ms": per_rank_stages,
                "assignment_cost": assignment_cost,
                "rank0_step_ms": [round(x, 3) for x in step_ms],
                "collectives_host_ms_per_step": {k: v / (args.steps + args.warmup) for k, v in host_ms.items()},
                "proofs_identical_across_steps": proofs_identical,
                "hbm_resident_input_bytes": int(input_bytes),
                "device_pools": pool_report,
                "host_execute_s": t_execute,
                "host_execute_s_per_rank": exec_s_per_rank,
                "end_to_end": {
                    "note": "every rank runs the WHOLE program through the host interpreter before proving (same query record on every rank, no broadcast of row streams); that time is outside the timed region and is the end-to-end Amdahl bound",
                    "host_execute_s_wall": max(exec_s_per_rank), "host_execute_s_summed_over_ranks": sum(exec_s_per_rank),
                    "host_execute_s_per_rank": exec_s_per_rank, "peak_rss_mb_per_rank": rss_per_rank,
                    "host_interpreter_eval_rows_per_s": world * n / max(exec_s_per_rank),
                    "host_interpreter_queries": host_queries, "host_interpreter_memory_cells": host_mem_cells,
                    "host_interpreter_queries_per_s": host_queries / t_execute,
                    "eval_steps_per_s_including_execute": world * n / (max(exec_s_per_rank) + ms_per_step * 1e-3),
                },
                "host_flatten_upload_s": t_flatten,
                "compiled_air_chips": compiled,
                "compiled_trace_chips": list(machine.compiled_traces),
                "air_compile_s": t_jit,
                "rank_proofs_in_flight": in_flight,
                "rank_in_flight_stagger_ms": (pipe["stagger_s"] * 1e3 if pipe is not None and in_flight >= 2 else None),
                "rank_pipeline": None if in_flight >= 2 else ("phase 1 of machine proof j + 1 (traces + main commitments, on a second machine's context) under phase 2 of proof j; "
                                  "collectives on one thread in a fixed order" if pipe is not None else None),
                "proofs_in_flight": lanes,
                "lane_stagger_ms": (stagger_ms if lanes >= 2 else None),
                "lane_placement": (lane_placement if lanes >= 2 else None),
                "schedule": (f"the K timed steps are K independent proofs of the shard, {lanes} in flight on {lanes} HIP streams / contexts of the GPU (prove lanes); "
                             "`sequential` is one proof at a time, measured before the timed region; stages_ms / roofline.hbm come from that sequential pass "
                             "(a kernel alone on the device), stages_ms_in_flight from the timed region (spans of the lanes overlap in time)") if lanes >= 2
                            else "one proof at a time on one HIP stream",
                "sequential": sequential,
                "stages_ms_in_flight": {k: v[0] / args.steps for k, v in spans.items() if v[1]} if lanes >= 2 else None,
                "host_pipeline": host_pipeline,
            },
            # The dominant kernels (Merkle hashing) are int32-VALU-issue-bound: the headline fraction is against the instruction-mix
            # ceiling of the VALU; the HBM fraction the contract names is kept beside it (`hbm`).
            "roofline": {
                # (no PMC summary for this workload / size: the live HBM fraction is the headline)
                "bound": "valu" if valu else "hbm",
                "kernel": "Merkle hashing (k_row_sponges + k_level_digests + k_level_coop; trees of 2^16 leaves and more), all launches of a step",
                "achieved": valu["achieved"] if valu else achieved,
                # the guide's number: 78.6 T lane-instr/s = 256 CUs x 4 SIMD-32 x 2.4 GHz, one VALU instruction per two cycles
                # (MI355X_MICROARCH.md "Wave scheduling"); the ceiling of THIS instruction mix is beside it (int32_valu.mix_ceiling)
                "peak": VALU_FULL_RATE if valu else HBM_PEAK_GBS,
                "unit": "T lane-instr/s" if valu else "GB/s",
                "frac": valu["frac_of_full_rate"] if valu else achieved / HBM_PEAK_GBS,
                "int32_valu": valu,
                "hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "note": "algorithmic bytes / summed HIP-event time of the hashing launches, measured live in this run"},
                "traffic": traffic,
                "also": {"kernel": "coset LDE (lde.hip: k_lde_in / k_lde_mid / k_lde_out per height group; ntt.hip for the shapes it does not take): the launches inside the `lde` spans "
                                   "of a step's three commitments",
                         "bound": "hbm",
                         "achieved": lde_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lde_alg / HBM_PEAK_GBS,
                         "time_note": "ms_per_step is the HIP-event time of the `lde` spans (stages_ms.lde): the short height groups' launches run on a side stream UNDER the tall "
                                      "groups', so the same kernels' durations summed by name in a rocprofv3 table (profiles/*_per_proof_kernel_stats.csv) come to more than the span",
                         "algorithmic_bytes_per_step": lde_alg_bytes, "algorithmic_bytes_rule": "SURVEY 8(d): 12 w B per trace row (read 4w, write 8w)",
                         "algorithmic_bytes_transformed": lde_alg_transformed,
                         "achieved_on_transformed": lde_alg_transformed / (lde_ms_step * 1e-3) / 1e9 if lde_ms_step > 0 else 0.0,
                         "frac_on_transformed": (lde_alg_transformed / (lde_ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if lde_ms_step > 0 else 0.0,
                         "transformed_note": "12 w B per row of the columns whose extension is computed: the permutation traces' identically-zero columns (interactions no row of "
                                             "the shard uses) are stored as zeros, not transformed -- the honest numerator for the kernels' own efficiency",
                         "pass_traffic_bytes_per_step": lde_pass_bytes, "pass_traffic_GBs": lde_pass, "ms_per_step": lde_ms_step,
                         "traffic": lde_traffic, "int32_valu": lde_valu,
                         "bound_note": "neither roof is near: the pass traffic runs at pass_traffic_GBs of 8000 and the butterflies at int32_valu.frac of the VALU peak -- the "
                                       "three kernels alternate memory phases and register phases in one or two workgroups a CU (DESIGN.md 3.3)"},
                "step": step_block,
                "algorithmic_bytes_per_step": hash_bytes_step,
                "launches_per_step": hash_launches_step,
                "ms_per_step": hash_ms_step,
                "note": "`frac` = achieved int32 VALU rate / the guide's 78.6 T lane-instr/s full rate (achieved: static instruction count / time measured live); int32_valu.frac_of_mix_ceiling is against the ceiling of this instruction mix (60 % four-cycle multiply-class).  int32-VALU bound: ceil(w/8) width-16 Poseidon2 permutations (4.76 k int32 instructions each, ~56 % of them four-cycle multiply-class) per w*4-byte row; throughput-bound on instruction issue (same speed at 4 and 8 waves per SIMD): k_row_sponges takes 19.9 k SIMD cycles per wave-permutation, the rate of the permutation alone on registers (tools/ubench_perm.hip: 20.8-21.1 k), i.e. ~0.3 TB/s algorithmic is this kernel's ceiling (DESIGN.md 3.4); `frac` = achieved VALU rate / the ceiling of its instruction mix (static PMC pass: profiles/pmc_traffic.json), `hbm.frac` = the HBM fraction measured live",
            },
        }
        if world == 1 and spr == 1 and not args.no_cpu_baseline and args.workload != "eval-only":
            try:
                shapes = [(air.name, lg, air.width) for _, air, lg, _, _ in prepared]
                out["cpu_baseline"] = cpu_baseline(args.workload, shapes, log_rows, args.queries, args.pow_bits)
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the line
                out["cpu_baseline"] = {"error": repr(e)}
        # the same command under the protocol profile whose width-16 permutation has the shape of the reference's (Plonky3's
        # Montgomery diffusion matrix, `p3-monty-diffusion`: lurk_amd/profile.py; the default profile's internal layer is knowingly
        # not sphinx's, csrc/merkle.hip) as a second named field, measured by a child process of this one
        if (world == 1 and spr == 1 and not distributed and args.profile == "default" and args.workload == "fib-mix" and not args.no_second_profile
                and os.environ.get("LURKHIP_BENCH_CHILD") != "1"):
            try:
                import subprocess

                cmd = [sys.executable, os.path.abspath(__file__), "--profile", "p3-monty-diffusion", "--no-cpu-baseline", "--no-host-pipeline", "--steps", str(args.steps),
                       "--warmup", str(args.warmup), "--log-rows", str(log_rows), "--queries", str(args.queries), "--pow-bits", str(args.pow_bits), "--lanes", str(args.lanes)]
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=dict(os.environ, LURKHIP_BENCH_CHILD="1"))
                child = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                out["ms_per_step_p3_monty_diffusion"] = child["ms_per_step"]
                out["proof_latency_ms_p3_monty_diffusion"] = child["proof_latency_ms"]
                out["value_p3_monty_diffusion"] = child["value"]
            except Exception as e:  # a reported extra, never a reason to lose the line
                out["ms_per_step_p3_monty_diffusion"] = None
                out["p3_monty_diffusion_error"] = repr(e)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()
    del prepared, prepared_all
    if pipe is not None:
        del pipe["prepared"]
        for c_ in pipe.get("comms", []):
            c_.close()
        for m_ in pipe["machines"]:
            m_.close()
        for cx_ in reversed(pipe["ctxs"]):
            cx_.close()
    while lane2:
        m2_, ctx2_, prep2_ = lane2.pop()
        del prep2_
        m2_.close()
        ctx2_.close()
    if lane_ctx is not None:
        lane_ctx.close()
    machine.close()
    ctx.close()


if __name__ == "__main__":
    main()
