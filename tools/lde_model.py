#!/usr/bin/env python3
"""Index-level model of the round-4 coset LDE kernels (lurk_amd/csrc/lde.hip), in numpy, canonical arithmetic.

Not the product and not the oracle: a development aid that mirrors the kernels' decomposition thread by thread -- which
rows a thread holds in registers, which twiddle-table entry each in-register butterfly reads, the bit-reversed hand-over
between the inverse transform's last stage group and the forward transform's first inside the fused kernel -- so that the
index arithmetic is checked on the CPU (against oracle/stark.py) before a GPU minute is spent.  `python tools/lde_model.py`
runs the checks.

Decomposition (n = log2 N rows, blow-up 2):
  n <= 10:  k_small   one tile = the whole column: inverse stages, then per coset scale + forward stages.
  n  > 10:  k_in      inverse DIF, top r1 bits: tiles of rows (t << r2 | lo), strided
            k_mid     inverse DIF, low r2 bits on contiguous rows; the tile then holds the coefficients of one forward
                      first-pass tile in bit-reversed order: per coset scale + forward DIF top r2 bits, strided store
            k_out     forward DIF, low r1 bits on contiguous rows, store into the coset's block of the LDE
A thread is (slot s, column c); it holds U = 2^min(5, r) rows of its column in registers.  Stage group 1 works on rows
s + S j (in-register index j = the top bits of the tile row), group 2 -- after one exchange through LDS -- on rows U s + j.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P = 2013265921
GEN = 31


def fpow(a, e):
    return pow(int(a), int(e), P)


def two_adic_generator(bits):
    r = 0x1A427A41
    for _ in range(bits, 27):
        r = r * r % P
    return r


def bitrev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def geo(log_r):
    log_u = min(5, log_r)
    log_s = log_r - log_u
    return log_u, 1 << log_u, log_s, 1 << log_s


def tile_twiddles(tw, n, log_r, bit_lo, lo):
    """tw_lds[(1 << b) + tl] = TW[((tl << bit_lo) | lo) << (n - bit_lo - b - 1)] for local stage b, tl < 2^b."""
    out = np.zeros(1 << log_r, dtype=np.uint64)
    for b in range(log_r):
        for tl in range(1 << b):
            out[(1 << b) + tl] = tw[((tl << bit_lo) | lo) << (n - bit_lo - b - 1)]
    return out


def butterfly(x, j, k, w):
    a, b = x[j].copy(), x[k].copy()
    x[j] = (a + b) % P
    x[k] = ((a + P - b) % P) * np.uint64(w) % P


def group1(x, s, tw, log_r):
    log_u, U, log_s, S = geo(log_r)
    for g in reversed(range(log_u)):
        for j in range(U):
            if (j >> g) & 1:
                continue
            m = (1 << g) + (j & ((1 << g) - 1))
            butterfly(x, j, j | (1 << g), tw[s + S * m])


def group2(x, tw, log_r):
    log_u, U, log_s, S = geo(log_r)
    for g in reversed(range(log_s)):
        for j in range(U):
            if (j >> g) & 1:
                continue
            butterfly(x, j, j | (1 << g), tw[(1 << g) + (j & ((1 << g) - 1))])


def transform_tile(load, store, tw, log_r, C):
    """group 1 on rows s + S j, exchange, group 2 on rows U s + j: the body of k_in and k_out."""
    log_u, U, log_s, S = geo(log_r)
    lds = np.zeros((1 << log_r, C), dtype=np.uint64)
    for s in range(S):
        x = np.stack([load(s + S * j) for j in range(U)])
        group1(x, s, tw, log_r)
        for j in range(U):
            lds[s + S * j] = x[j]
    for s in range(S):
        x = np.stack([lds[U * s + j] for j in range(U)])
        group2(x, tw, log_r)
        for j in range(U):
            store(U * s + j, x[j])


def fused_tile(load, store, tw_inv_tile, fwd_tw_of_coset, scale_of_coset, log_r, C, n_cosets=2):
    """k_mid / k_small: inverse stages, bit-reversed hand-over in registers, per coset scale + forward stages."""
    log_u, U, log_s, S = geo(log_r)
    lds = np.zeros((1 << log_r, C), dtype=np.uint64)
    coef = {}
    for s in range(S):
        x = np.stack([load(s + S * j) for j in range(U)])
        group1(x, s, tw_inv_tile, log_r)
        for j in range(U):
            lds[s + S * j] = x[j]
    for s in range(S):
        x = np.stack([lds[U * s + j] for j in range(U)])
        group2(x, tw_inv_tile, log_r)
        coef[s] = x  # register j of thread s: local position U s + j
    for q in range(n_cosets):
        twf = fwd_tw_of_coset(q)
        for s in range(S):  # thread s acts as forward slot s2 = bitrev(s); register j becomes j2 = bitrev(j)
            s2 = bitrev(s, log_s)
            y = np.zeros((U, C), dtype=np.uint64)
            for j in range(U):
                j2 = bitrev(j, log_u)
                t2 = s2 + S * j2
                y[j2] = coef[s][j] * scale_of_coset(q, t2) % P
            group1(y, s2, twf, log_r)
            for j2 in range(U):
                lds[s2 + S * j2] = y[j2]
        for s in range(S):
            x = np.stack([lds[U * s + j] for j in range(U)])
            group2(x, twf, log_r)
            for j in range(U):
                store(q, U * s + j, x[j])


def lde_group(mats, log_n, shifts, r1=None):
    """mats: list of N x w uint64 arrays (canonical); shifts: per matrix coset shift.  Returns the 2N x w LDEs, rows in
    bit-reversed order (block q = coset q), as the kernels store them."""
    n = log_n
    N = 1 << n
    widths = [m.shape[1] for m in mats]
    W = sum(widths)
    virt = np.concatenate(mats, axis=1).astype(np.uint64)  # the virtual row
    col_shift = np.concatenate([[s] * w for s, w in zip(shifts, widths)])
    root = two_adic_generator(n)
    root_inv = fpow(root, P - 2)
    half = max(1, N // 2)
    tw_fwd = np.array([fpow(root, i) for i in range(half)], dtype=np.uint64)
    tw_inv = np.array([fpow(root_inv, i) for i in range(half)], dtype=np.uint64)
    w_big = two_adic_generator(n + 1)
    n_inv = fpow(N, P - 2)
    scale = {}
    for sh in set(shifts):
        for q in range(2):
            sq = sh * fpow(w_big, q) % P
            scale[(sh, q)] = np.array([fpow(sq, k) * n_inv % P for k in range(N)], dtype=np.uint64)

    def col_scale(q, k):  # per virtual column: scale of coefficient k on coset q
        return np.array([scale[(int(sh), q)][k] for sh in col_shift], dtype=np.uint64)

    out = np.zeros((2 * N, W), dtype=np.uint64)
    if n <= 10 and r1 is None:
        twi = tile_twiddles(tw_inv, n, n, 0, 0)
        twf = tile_twiddles(tw_fwd, n, n, 0, 0)
        fused_tile(lambda t: virt[t], lambda q, t, v: out.__setitem__(q * N + t, v), twi, lambda q: twf,
                   lambda q, t2: col_scale(q, t2), n, W)
    else:
        if r1 is None:
            r1 = (n + 1) // 2
        r2 = n - r1
        A = np.zeros((N, W), dtype=np.uint64)
        B = [np.zeros((N, W), dtype=np.uint64) for _ in range(2)]
        for lo in range(1 << r2):  # k_in
            tw = tile_twiddles(tw_inv, n, r1, r2, lo)
            transform_tile(lambda t: virt[(t << r2) | lo], lambda t, v: A.__setitem__((t << r2) | lo, v), tw, r1, W)
        twi = tile_twiddles(tw_inv, n, r2, 0, 0)
        for hi in range(1 << r1):  # k_mid
            lo2 = bitrev(hi, r1)
            twf = tile_twiddles(tw_fwd, n, r2, r1, lo2)
            fused_tile(lambda t: A[(hi << r2) | t], lambda q, t, v: B[q].__setitem__((t << r1) | lo2, v), twi, lambda q: twf,
                       lambda q, t2: col_scale(q, (t2 << r1) | lo2), r2, W)
        two = tile_twiddles(tw_fwd, n, r1, 0, 0)
        for q in range(2):  # k_out
            for hi in range(1 << r2):
                transform_tile(lambda t: B[q][(hi << r1) | t], lambda t, v: out.__setitem__(q * N + ((hi << r1) | t), v), two, r1, W)
    res, at = [], 0
    for w in widths:
        res.append(out[:, at:at + w])
        at += w
    return res


def check(log_n, widths, shifts, r1=None, seed=1):
    from oracle import stark as st

    rng = np.random.default_rng(seed)
    N = 1 << log_n
    mats = [rng.integers(0, P, size=(N, w), dtype=np.uint64) for w in widths]
    got = lde_group(mats, log_n, shifts, r1)
    for m, sh, g in zip(mats, shifts, got):
        want = st.bit_reverse_rows(st.coset_lde([[int(v) for v in r] for r in m], 1, sh))
        assert [[int(v) for v in r] for r in g] == want, (log_n, widths, shifts, r1)


if __name__ == "__main__":
    for n in range(0, 8):
        check(n, [3, 2], [GEN, GEN])
    check(6, [1, 2], [GEN, 5])            # two shift classes in one group
    check(7, [2], [GEN])                  # U = 32, S = 4, one tile
    for n, r1 in [(4, 2), (5, 3), (6, 3), (7, 4), (8, 4), (9, 5), (10, 5), (11, 6), (12, 6), (12, 7)]:
        check(n, [2, 1], [GEN, 7], r1)
    print("lde_model: all checks passed")
