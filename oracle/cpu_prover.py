"""ORACLE directory (test infrastructure, not product code): the CPU PORT of the whole proving step.

`CpuProver` proves one shard on the host cores the way sphinx's `prove_shard` does [UPSTREAM-RECALL], stage for stage what
bench.py times on the GPU (permutation traces, three commitments, quotient, openings, FRI with proof-of-work and queries):
the transcript is the oracle's (`stark.Challenger`), every data-parallel stage is C (oracle/cpu_step.c, OpenMP, Montgomery)
and the chips' constraint / interaction evaluators are C generated from the oracle's AIR (oracle/cpu_emit.py).  The result is
the structure the oracle's verifier consumes (`wire.Shard`), and `stark.verify_machine` accepts it
(tests/test_cpu_step.py; on the GPU box the same test compares it with the HIP prover's proof field by field).

It is the `cpu_baseline` of bench.py: kind "port" -- neither the reference binary (Rust: not buildable here) nor tuned like
one (scalar field arithmetic per row, no packed AVX-512 field).  Trace generation is its own port (oracle/cpu_trace.c, driven
by oracle/cpu_trace.py from the oracle interpreter's query record); this prover starts from the traces, like `machine.prove`
does after `generate_trace`.  Only tests/ and bench.py's cpu_baseline leg import this."""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
import time

import numpy as np

from . import cpu_emit
from . import stark as os_
from . import wire as ow

HERE = os.path.dirname(os.path.abspath(__file__))
GEN_DIR = os.path.join(HERE, "_gen")
P = os_.P
R = (1 << 32) % P
R_INV = pow(R, P - 2, P)
GEN = 31


def to_m(x):
    return (np.asarray(x, dtype=np.uint64) * np.uint64(R) % np.uint64(P)).astype(np.uint32)


def from_m(x):
    return (np.asarray(x, dtype=np.uint64) * np.uint64(R_INV) % np.uint64(P)).astype(np.uint32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _ef_m(e):
    return np.ascontiguousarray(to_m(np.array(list(e), dtype=np.uint64)))


class CpuProver:
    """airs / names by machine index; `prep` = {machine index: canonical preprocessed trace}."""

    def __init__(self, airs, names, n_public, log_blowup=1, threads=None, verbose=False):
        self.airs, self.names, self.n_public, self.log_blowup = list(airs), list(names), n_public, log_blowup
        chips = [(a, n, a.width, getattr(a, "prep_width", 0)) for a, n in zip(self.airs, self.names)]
        t0 = time.perf_counter()
        units, self.stats = cpu_emit.emit_machine(chips, n_public)
        src = "\n".join(units)
        tag = hashlib.sha256((src + open(os.path.join(HERE, "cpu_step.c")).read() + open(os.path.join(HERE, "cpu_port.c")).read()
                              + open(os.path.join(HERE, "cpu_step.h")).read()).encode()).hexdigest()[:16]
        os.makedirs(GEN_DIR, exist_ok=True)
        so = os.path.join(GEN_DIR, f"cpu_step_{tag}.so")
        if not os.path.exists(so):
            # one translation unit per chip, compiled in parallel (the hash chips' evaluators are ~10^4 statements each)
            from concurrent.futures import ThreadPoolExecutor

            work = os.path.join(GEN_DIR, f"build_{tag}_{os.getpid()}")
            os.makedirs(work, exist_ok=True)
            jobs = []
            for k, u in enumerate(units):
                c_path = os.path.join(work, f"unit{k}.c")
                with open(c_path, "w") as f:
                    f.write(u)
                jobs.append(["gcc", "-O1", "-fPIC", "-I", HERE, "-c", c_path, "-o", c_path[:-2] + ".o"])
            jobs.append(["gcc", "-O2", "-fopenmp", "-fPIC", "-I", HERE, "-c", os.path.join(HERE, "cpu_step.c"), "-o", os.path.join(work, "cpu_step.o")])
            jobs.append(["gcc", "-O2", "-fPIC", "-I", HERE, "-c", os.path.join(HERE, "poseidon2.c"), "-o", os.path.join(work, "poseidon2.o")])
            with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
                for rc in ex.map(subprocess.call, jobs):
                    if rc != 0:
                        raise RuntimeError("gcc failed on a unit of the CPU port")
            tmp = so + f".{os.getpid()}.tmp"
            subprocess.check_call(["gcc", "-shared", "-fopenmp", "-o", tmp] + [j[-1] for j in jobs])
            os.replace(tmp, so)
            import shutil

            shutil.rmtree(work, ignore_errors=True)
        self.build_s = time.perf_counter() - t0
        if verbose:
            print(f"cpu port: {len(chips)} chip evaluators, {len(src) // 1024} KiB of C, built in {self.build_s:.1f} s")
        L = self.L = C.CDLL(so)
        L.cp2_commit.restype = C.c_void_p
        L.cp2_commit.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cp2_tree_open.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.cp2_tree_free.argtypes = [C.c_void_p]
        L.cp2_lde.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]
        L.cp2_perm_width.argtypes = [C.c_int, C.c_int]
        L.cp2_perm_trace.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p]
        L.cp2_quotient.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p]
        L.cp2_open.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.cp2_inv_denoms.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.cp2_reduce.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cp2_fri_fold.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cp2_perm_ns.restype = C.c_double
        L.cp2_perm_ns.argtypes = [C.c_int, C.c_void_p]
        L.cp2_pow_grind.restype = C.c_uint32
        L.cp2_pow_grind.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.cp2_to_monty.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.cp_set_threads.argtypes = [C.c_int]
        if threads:
            L.cp_set_threads(int(threads))

    # ------------------------------------------------------------------ building blocks
    def monty(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        out = np.empty_like(a)
        self.L.cp2_to_monty(_ptr(a), _ptr(out), a.size)
        return out

    def lde(self, mat_m, shift=GEN):
        n, w = mat_m.shape
        out = np.empty((n << self.log_blowup, w), dtype=np.uint32)
        assert self.L.cp2_lde(n.bit_length() - 1, w, self.log_blowup, _ptr(mat_m), _ptr(out), shift % P) == 0
        return out

    def commit(self, ldes):
        """Tree over Montgomery matrices (kept alive by the caller): (handle, canonical root)."""
        n = len(ldes)
        ptrs = (C.c_void_p * n)(*[m.ctypes.data for m in ldes])
        lh = np.array([m.shape[0].bit_length() - 1 for m in ldes], dtype=np.uint32)
        ws = np.array([m.shape[1] for m in ldes], dtype=np.uint32)
        root = np.zeros(8, dtype=np.uint32)
        h = self.L.cp2_commit(n, C.cast(ptrs, C.c_void_p), _ptr(lh), _ptr(ws), _ptr(root))
        return h, [int(x) for x in root], (lh, ws)

    def tree_open(self, tree, shapes, index):
        lh, ws = shapes
        rows = np.zeros(int(ws.sum()), dtype=np.uint32)
        path = np.zeros(8 * int(lh.max()), dtype=np.uint32)
        self.L.cp2_tree_open(tree, index, _ptr(rows), _ptr(path))
        return [int(x) for x in rows] + [int(x) for x in path]

    @staticmethod
    def prover_order(traces):
        """The chips of a shard in the prover's order: by trace height, tallest first, machine order among equals (sphinx sorts
        the shard's chips by height before committing [UPSTREAM-RECALL]; the HIP prover does the same, prover.hip)."""
        return sorted(traces, key=lambda t: -t[1].shape[0])

    # ------------------------------------------------------------------ one shard
    def prove_shard(self, traces, prep, prep_commit, public_values, challenger, num_queries=100, pow_bits=16, timings=None):
        """traces: [(machine index, canonical main trace [N][w])] in machine order; prep: {machine index: Montgomery
        preprocessed trace}; prep_commit: result of `setup()` or None; `challenger`: the oracle's, in the state it has when
        prove_shard starts (vk, pc_start, every shard's main root + public values observed)."""
        L, lb = self.L, self.log_blowup
        tm = timings if timings is not None else {}

        def timed(name):
            class T:
                def __enter__(s):
                    s.t = time.perf_counter()

                def __exit__(s, *a):
                    tm[name] = tm.get(name, 0.0) + time.perf_counter() - s.t
            return T()

        traces = self.prover_order(traces)
        ch = challenger
        prof = ch.profile
        pub_m = to_m(np.array(list(public_values) + [0], dtype=np.uint64))
        mis = [mi for mi, _ in traces]
        with timed("to_montgomery"):
            mains = [self.monty(t) for _, t in traces]
        log_ns = [m.shape[0].bit_length() - 1 for m in mains]
        # ---- main commitment (phase 1 of machine.prove did this; repeated here so that a shard stands alone)
        with timed("commit_main"):
            main_lde = [self.lde(m) for m in mains]
            main_tree, main_root, main_shapes = self.commit(main_lde)
        # ---- permutation traces
        if prof.observe_chip_meta:
            for mi, lg in zip(mis, log_ns):
                ch.observe([lg, self.airs[mi].width, (0 if mi in prep else -1) + 1])
        perm_alpha, perm_beta = ch.sample_ext(), ch.sample_ext()
        pa_m, pb_m = _ef_m(perm_alpha), _ef_m(perm_beta)
        perms, cums = [], []
        with timed("permutation"):
            for mi, m, lg in zip(mis, mains, log_ns):
                pw = L.cp2_perm_width(mi, 2)
                out = np.empty((m.shape[0], 4 * pw), dtype=np.uint32)
                cs = np.zeros(4, dtype=np.uint32)
                pm = prep.get(mi)
                assert L.cp2_perm_trace(mi, lg, _ptr(m), _ptr(pm) if pm is not None else None, _ptr(pub_m), _ptr(pa_m), _ptr(pb_m), 2, _ptr(out), _ptr(cs)) == 0
                perms.append(out)
                cums.append(cs)
        with timed("commit_perm"):
            perm_lde = [self.lde(p) for p in perms]
            perm_tree, perm_root, perm_shapes = self.commit(perm_lde)
        cums_c = [tuple(int(x) for x in from_m(c)) for c in cums]
        ch.observe(perm_root)
        if prof.observe_chip_meta:
            for c in cums_c:
                ch.observe(list(c))
        # ---- quotient
        alpha = ch.sample_ext()
        a_m = _ef_m(alpha)
        chunks = []  # (chip position, chunk index, Montgomery N x 4)
        with timed("quotient_all"):
            for k, (mi, lg) in enumerate(zip(mis, log_ns)):
                n = 1 << lg
                out = np.empty((2, n, 4), dtype=np.uint32)
                pl = prep_commit["lde"][prep_commit["index"][mi]] if mi in prep else None
                assert L.cp2_quotient(mi, lg, _ptr(main_lde[k]), _ptr(pl) if pl is not None else None, _ptr(perm_lde[k]), _ptr(pub_m), _ptr(pa_m), _ptr(pb_m),
                                      _ptr(a_m), _ptr(cums[k]), 1, _ptr(out)) == 0
                chunks.append(out)
        with timed("commit_quotient"):
            quot_lde = []
            for k, lg in enumerate(log_ns):
                wq_inv = pow(os_.two_adic_generator(lg + 1), P - 2, P)
                for c in range(2):
                    quot_lde.append(self.lde(np.ascontiguousarray(chunks[k][c]), shift=pow(wq_inv, c, P)))
            quot_tree, quot_root, quot_shapes = self.commit(quot_lde)
        ch.observe(quot_root)
        zeta = ch.sample_ext()
        # ---- rounds in the prover's order: (tree, shapes, [(lde, log_n, [points])])
        def pts_of(lg):
            return [zeta, os_.ef_scale(zeta, os_.two_adic_generator(lg))]

        rounds = []
        if prep_commit is not None:
            rounds.append((prep_commit["tree"], prep_commit["shapes"], [(l, l.shape[0].bit_length() - 1 - lb, pts_of(l.shape[0].bit_length() - 1 - lb)) for l in prep_commit["lde"]]))
        rounds.append((main_tree, main_shapes, [(l, lg, pts_of(lg)) for l, lg in zip(main_lde, log_ns)]))
        rounds.append((perm_tree, perm_shapes, [(l, lg, pts_of(lg)) for l, lg in zip(perm_lde, log_ns)]))
        rounds.append((quot_tree, quot_shapes, [(l, l.shape[0].bit_length() - 1 - lb, [zeta]) for l in quot_lde]))
        # ---- opened values
        opened = []  # [round][matrix] -> Montgomery [n_pts][w][4]
        with timed("open"):
            for _, _, mats in rounds:
                ro_ = []
                for l, lg, pts in mats:
                    zs = np.ascontiguousarray(np.stack([_ef_m(z) for z in pts]))
                    out = np.empty((len(pts), l.shape[1], 4), dtype=np.uint32)
                    assert L.cp2_open(lg, l.shape[1], _ptr(l), len(pts), _ptr(zs), _ptr(out)) == 0
                    ro_.append(out)
                opened.append(ro_)
            opened_c = [[from_m(o) for o in r] for r in opened]
            if prof.observe_openings:
                for r in opened_c:
                    for o in r:
                        for p in range(o.shape[0]):
                            for v in o[p]:
                                ch.observe([int(x) for x in v])
            alpha_fri = ch.sample_ext()
            af_m = _ef_m(alpha_fri)
            # reduced openings per LDE height
            log_max = max(l.shape[0].bit_length() - 1 for _, _, mats in rounds for l, _, _ in mats)
            ro = {}
            apow = {}
            inv_cache = {}

            def invd(log_h, z):
                key = (log_h, z)
                if key not in inv_cache:
                    out = np.empty((1 << log_h, 4), dtype=np.uint32)
                    L.cp2_inv_denoms(log_h, _ptr(_ef_m(z)), _ptr(out))
                    inv_cache[key] = out
                return inv_cache[key]

            for (_, _, mats), ops in zip(rounds, opened):
                for (l, lg, pts), ys in zip(mats, ops):
                    log_h, w = l.shape[0].bit_length() - 1, l.shape[1]
                    key = 0 if prof.fri_alpha_global else log_h
                    ap = apow.get(key, os_.ONE)
                    a0 = []
                    for _ in pts:
                        a0.append(_ef_m(ap))
                        ap = os_.ef_mul(ap, os_.ef_pow(alpha_fri, w))
                    apow[key] = ap
                    if log_h not in ro:
                        ro[log_h] = np.zeros((1 << log_h, 4), dtype=np.uint32)
                    tabs = [invd(log_h, z) for z in pts]
                    ptrs = (C.c_void_p * len(pts))(*[t.ctypes.data for t in tabs])
                    a0a = np.ascontiguousarray(np.stack(a0))
                    assert L.cp2_reduce(log_h, w, _ptr(l), len(pts), C.cast(ptrs, C.c_void_p), _ptr(ys), _ptr(af_m), _ptr(a0a), _ptr(ro[log_h])) == 0
        # ---- FRI commit phase
        fri_roots, layer_trees, betas = [], [], []
        with timed("fri_commit"):
            cur = ro[log_max]
            for log_size in range(log_max, lb, -1):
                mat = np.ascontiguousarray(cur.reshape(1 << (log_size - 1), 8))
                tree, root, shapes = self.commit([mat])
                layer_trees.append((tree, shapes, mat))
                fri_roots.append(root)
                ch.observe(root)
                beta = ch.sample_ext()
                betas.append(beta)
                out = np.empty((1 << (log_size - 1), 4), dtype=np.uint32)
                L.cp2_fri_fold(log_size, _ptr(cur), _ptr(_ef_m(beta)), _ptr(out))
                if (log_size - 1) in ro:
                    out = ((out.astype(np.uint64) + ro[log_size - 1]) % np.uint64(P)).astype(np.uint32)
                cur = out
            final = from_m(cur)
            assert all((final[i] == final[0]).all() for i in range(final.shape[0])), "the folded codeword is not constant"
            final_poly = tuple(int(x) for x in final[0])
            ch.observe(list(final_poly))
        with timed("pow"):
            st = np.array(ch.state, dtype=np.uint32)
            pend = np.array(list(ch.input) + [0], dtype=np.uint32)
            lane = 0 if prof.challenger_pop_front else prof.challenger_squeeze - 1
            witness = int(L.cp2_pow_grind(_ptr(st), len(ch.input), _ptr(pend), pow_bits, lane))
            assert ch.check_witness(pow_bits, witness), "proof-of-work witness rejected by the transcript"
        with timed("fri_query"):
            indices = [ch.sample_bits(log_max) for _ in range(num_queries)]
            round_openings, layer_openings = [], []
            for tree, shapes, _ in rounds:
                lbm = int(shapes[0].max())
                recs = [self.tree_open(tree, shapes, ix >> (log_max - lbm)) for ix in indices]
                round_openings.append((int(shapes[1].sum()) + 8 * lbm, recs))
            for li, (tree, shapes, _) in enumerate(layer_trees):
                recs = [self.tree_open(tree, shapes, (ix >> li) >> 1) for ix in indices]
                layer_openings.append((8 + 8 * int(shapes[0].max()), recs))
        # ---- the proof, in the structure the verifier reads
        chips = []
        ri = 1 if prep_commit is not None else 0
        for k, (mi, lg) in enumerate(zip(mis, log_ns)):
            air = self.airs[mi]
            pidx = prep_commit["index"][mi] if mi in prep else -1
            cp = ow.Chip(mi, lg, air.width, getattr(air, "prep_width", 0) if mi in prep else 0, perms[k].shape[1], 2, pidx, cums_c[k], name=self.names[mi])
            ef_list = lambda a: [tuple(int(x) for x in v) for v in a]
            cp.opened = {"main": (ef_list(opened_c[ri][k][0]), ef_list(opened_c[ri][k][1])),
                         "perm": (ef_list(opened_c[ri + 1][k][0]), ef_list(opened_c[ri + 1][k][1])),
                         "quotient": [ef_list(opened_c[ri + 2][2 * k + c][0]) for c in range(2)]}
            if pidx >= 0:
                cp.opened["prep"] = (ef_list(opened_c[0][pidx][0]), ef_list(opened_c[0][pidx][1]))
            chips.append(cp)
        for tree, _, _ in layer_trees:
            L.cp2_tree_free(tree)
        for t in (main_tree, perm_tree, quot_tree):
            L.cp2_tree_free(t)
        return ow.Shard(lb, num_queries, pow_bits, log_max, chips, list(public_values), main_root, perm_root, quot_root, fri_roots, final_poly, witness,
                        round_openings, layer_openings, len(prep_commit["lde"]) if prep_commit is not None else 0, query_indices=indices, sibling_only=False)

    def setup(self, prep_traces):
        """Commitment of the preprocessed traces ({machine index: canonical matrix}); its root is the verifying key."""
        idx = {mi: k for k, mi in enumerate(sorted(prep_traces))}
        mats = {mi: self.monty(prep_traces[mi]) for mi in prep_traces}
        ldes = [self.lde(mats[mi]) for mi in sorted(prep_traces)]
        tree, root, shapes = self.commit(ldes)
        return mats, {"tree": tree, "root": root, "shapes": shapes, "lde": ldes, "index": idx}
