"""ORACLE (test infrastructure, not product code): restatement of the reference's in-tree LogUp module, /root/reference/src/logup/
(SURVEY.md 8a row L1).  That module is dead upstream -- `// pub mod logup;` (/root/reference/src/lib.rs:6), nothing calls it,
it has no tests and does not compile as written -- so there is nothing to pin it with: **parity unpinned**; this file states
what the code SAYS, line by line, including the places where it disagrees with itself:

  * `generate_multiplicities_trace` (logup/trace.rs:10-50) takes `max(trace index)` powers of z and then indexes that table
    with the trace index itself (trace.rs:20-36): the largest index is one past the end (a panic upstream).  Here the table
    has max + 1 entries, i.e. the evident intent `z^(trace + 1)`.
  * `generate_permutation_trace` (logup/trace.rs:53-151) stores in column 1 + k the INVERSE 1 / d_k (not m_k / d_k as its
    comment says), accumulates sum_k m_k / d_k into column 0 and turns column 0 into an INCLUSIVE running sum
    (trace.rs:142-148), while `eval_logup_constraints` (logup/air.rs:11-77) constrains an EXCLUSIVE one (s_0 = 0,
    s_{i+1} = s_i + t_i, S = s_{n-1} + t_{n-1}); and the trace orders interactions provides-then-requires
    (trace.rs:74) while the AIR chains requires-then-provides (air.rs:37).  `exclusive=True` / `air_order=True` give the
    variant the constraints accept.
  * d_k = r + sum_j gamma^j v_{k,j} with gamma^0 = 1 (`Interaction::apply`, logup/air.rs:79-109).

Interactions are lists of affine forms over (identity | preprocessed | main) columns (`PairColLC`,
/root/reference/src/air/symbolic/virtual_col.rs:8-13,117-138): here `LC = (terms, constant)` with terms = [(kind, index,
weight)], kind 0 = identity (row index), 1 = preprocessed, 2 = main; an interaction = (values: [LC], is_real: LC or None)."""
from __future__ import annotations

from . import stark as os_

P = os_.P
IDENTITY, PREP, MAIN = 0, 1, 2


def lc_apply(lc, identity, prep_row, main_row):  # virtual_col.rs:123-138
    terms, const = lc
    r = const % P
    for kind, idx, w in terms:
        v = identity if kind == IDENTITY else (prep_row[idx] if kind == PREP else main_row[idx])
        r = (r + v * w) % P
    return r


def multiplicities_trace(multiplicities, z):
    """multiplicities: [(traces: [int], counts: [height][len(traces)])] per provide interaction.  Returns [height][n] of EF."""
    height = len(multiplicities[0][1])
    assert all(len(c) == height for _, c in multiplicities)
    n_pow = max(t for tr, _ in multiplicities for t in tr) + 1  # (upstream: one entry short, trace.rs:20-31)
    zp, cur = [], z
    for _ in range(n_pow):  # z^1, z^2, ... (`powers().skip(1)`)
        zp.append(cur)
        cur = os_.ef_mul(cur, z)
    out = [[os_.ZERO] * len(multiplicities) for _ in range(height)]
    for i, (traces, counts) in enumerate(multiplicities):
        for r in range(height):
            acc = os_.ZERO
            for m, t in zip(counts[r], traces):
                acc = os_.ef_add(acc, os_.ef_scale(zp[t], m % P))
            out[r][i] = acc
    return out


def interaction_denominator(inter, identity, prep_row, main_row, r, gamma):  # logup/air.rs:79-109
    values, _ = inter
    d = r
    gp = os_.ONE
    for j, lc in enumerate(values):
        v = lc_apply(lc, identity, prep_row, main_row)
        d = os_.ef_add(d, os_.ef(v) if j == 0 else os_.ef_scale(gp, v))
        gp = os_.ef_mul(gp, gamma)
    return d


def permutation_trace(identity_col, prep, main, mult, provides, requires, z, r, gamma, exclusive=False):
    """logup/trace.rs:53-151.  Rows [s, 1/d_0, 1/d_1, ...] (0 where is_real evaluates to 0), provides first."""
    height = len(main)
    inters = list(provides) + list(requires)
    rows = []
    for i in range(height):
        pr = prep[i] if prep is not None else ()
        cells = []
        for inter in inters:
            is_real = inter[1]
            if is_real is not None and lc_apply(is_real, identity_col[i], pr, main[i]) == 0:
                cells.append(os_.ZERO)
                continue
            cells.append(interaction_denominator(inter, identity_col[i], pr, main[i], r, gamma))
        total = os_.ZERO
        ms = list(mult[i]) + [os_.ef_neg(z)] * len(requires)
        for k, (c, m) in enumerate(zip(cells, ms)):
            if c == os_.ZERO:
                continue
            cells[k] = os_.ef_inv(c)
            total = os_.ef_add(total, os_.ef_mul(cells[k], m))
        rows.append([total] + cells)
    run = os_.ZERO
    for row in rows:
        t = row[0]
        if exclusive:
            row[0] = run
            run = os_.ef_add(run, t)
        else:
            run = os_.ef_add(run, t)
            row[0] = run
    return rows, run  # run = the LogUp sum of the trace


def eval_constraints(perm_local, perm_next, mult_row, identity, prep_row, main_row, provides, requires, z, r, gamma, final_sum, sels, air_order=True):
    """logup/air.rs:11-77 on one row pair: [per interaction d_k * inv_k - 1 (times is_real), first-row, transition, last-row]
    as EF values.  sels = (is_first_row, is_last_row, is_transition) base values.  air_order: the inverse columns follow the
    AIR's chain(requires, provides) (air.rs:37); False: the trace generator's provides-then-requires."""
    partial, inverses = perm_local[0], perm_local[1:]
    partial_next = perm_next[0]
    if air_order:
        inters = list(requires) + list(provides)
        ms = list(mult_row) + [os_.ef_neg(z)] * len(requires)  # air.rs:39-43: provide multiplicities first, whatever the chain order
    else:
        inters = list(provides) + list(requires)
        ms = list(mult_row) + [os_.ef_neg(z)] * len(requires)
    out, running = [], os_.ZERO
    for inter, m, inv in zip(inters, ms, inverses):
        d = interaction_denominator(inter, identity, prep_row, main_row, r, gamma)
        c = os_.ef_sub(os_.ef_mul(d, inv), os_.ONE)
        if inter[1] is not None:
            real = lc_apply(inter[1], identity, prep_row, main_row)
            out.append(os_.ef_scale(c, real))
            running = os_.ef_add(running, os_.ef_scale(os_.ef_mul(m, inv), real))
        else:
            out.append(c)
            running = os_.ef_add(running, os_.ef_mul(m, inv))
    is_first, is_last, is_trans = sels
    out.append(os_.ef_scale(partial, is_first))
    out.append(os_.ef_scale(os_.ef_sub(os_.ef_add(running, partial), partial_next), is_trans))
    out.append(os_.ef_scale(os_.ef_sub(running, final_sum), is_last))
    return out
