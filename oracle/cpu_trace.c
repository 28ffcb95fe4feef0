/* ORACLE directory (test infrastructure, not product code): CPU PORT of FuncChip trace generation, the `trace_all` stage of
 * bench.py's `cpu_baseline` (the rest of the step is cpu_step.c).
 *
 * Follows /root/reference/src/lair/trace.rs:72-135 (generate_trace), 145-418 (populate_row), 218-254 (inequality / depth
 * witnesses) and the extern chips' witnesses (/root/reference/src/gadgets/unsigned/{add,cmp,is_zero,mul,div_rem}.rs,
 * src/gadgets/big_num/cmp.rs, src/poseidon/wide/trace.rs through poseidon2.c) the way oracle/lair.py: generate_trace states
 * them, with the reference's hash-map lookups taken out of the row loop: oracle/cpu_trace.py walks the oracle's bytecode once
 * per function into the flat program below and flattens a query record into per-row headers and hint streams (the values
 * `populate_row` looks up: callee outputs, preimages, pointers, loaded values, callee depths, require records), exactly the
 * split the GPU path makes between the interpreter and its trace kernels.  Rows are independent: OpenMP over rows.
 * tests/test_cpu_trace.py checks it word for word against oracle/lair.py: generate_trace.  Canonical words in and out.
 *
 * Program (int32 words):
 *   OP_ASSERT_NE n a[n] b[n] | OP_CONTAINS n y a[n] | OP_CONST c | OP_ADD x y | OP_SUB x y | OP_MUL x y emit | OP_INV x emit |
 *   OP_NOT x emit | OP_HINT n (n hint words -> variables and aux) | OP_REQ (2 hint words -> nonce, count, 1 / (count + 1)) |
 *   OP_DEPTH (1 hint word: callee depth -> 4 bytes + 6-lane less-than witness against the row's own depth, then OP_REQ) |
 *   OP_EXTERN kind n_in in[n_in] n_req | OP_RETURN sel | OP_MATCH n_vars vars[] n_cases (key[n_vars] pc)* default_pc */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "field.h"

int or_p2_wide_witness(int width, size_t n, const uint32_t* in, uint32_t* out);
int or_p2_permute(int width, size_t n, const uint32_t* in, uint32_t* out);
int or_p2_num_cols(int width);

enum { OP_ASSERT_NE = 1, OP_CONTAINS, OP_CONST, OP_ADD, OP_SUB, OP_MUL, OP_INV, OP_NOT, OP_HINT, OP_REQ, OP_DEPTH, OP_EXTERN, OP_RETURN, OP_MATCH };
enum { X_HASHER = 1, X_U64_ADD, X_U64_SUB, X_U64_MUL, X_U64_DIVREM, X_U64_LESSTHAN, X_U64_ISZERO, X_BIGNUM_LT };

#define TP ((uint64_t)OR_P)
static inline uint32_t tadd(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + b) % TP); }
static inline uint32_t tsub(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + TP - b) % TP); }
static inline uint32_t tmul(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % TP); }
/* Fermat inverse on Montgomery words (a 64-bit `%` per product made the inversions most of a row's time) */
static inline uint32_t tmm(uint32_t a, uint32_t b) { /* a b 2^-32 mod p */
    const uint64_t t = (uint64_t)a * b;
    const uint32_t m = (uint32_t)t * 0x88000001u;
    const uint32_t u = (uint32_t)(((uint64_t)m * OR_P) >> 32), hi = (uint32_t)(t >> 32);
    const uint32_t r = hi - u;
    return hi < u ? r + OR_P : r;
}
static uint32_t tinv(uint32_t a) {
    const uint32_t R2 = 1172168163u; /* 2^64 mod p */
    uint32_t b = tmm(a % OR_P, R2), r = tmm(1u, R2);
    for (uint32_t e = OR_P - 2; e; e >>= 1) {
        if (e & 1) r = tmm(r, b);
        b = tmm(b, b);
    }
    return tmm(r, 1u);
}
#define INV_TABLE 1024
static uint32_t g_inv[INV_TABLE];
static void inv_init(void) {
    if (g_inv[1]) return;
    for (uint32_t i = 1; i < INV_TABLE; i++) g_inv[i] = tinv(i);
}
static inline uint32_t count_inv(uint32_t c) { return c < INV_TABLE ? g_inv[c] : tinv(c); }

typedef struct {
    uint32_t* row;
    uint32_t aux0, aux;
    uint32_t* m; /* variables */
    uint32_t nm;
    const uint32_t* h; /* hint cursor */
} rowctx;
static inline void push_aux(rowctx* c, uint32_t v) { c->row[c->aux0 + c->aux++] = v; }
static inline void push_req(rowctx* c) {
    const uint32_t nonce = *c->h++, count = *c->h++;
    push_aux(c, nonce);
    push_aux(c, count);
    push_aux(c, count_inv(count + 1));
}
static uint64_t u64_of(const uint32_t* v) {
    uint64_t r = 0;
    for (int i = 0; i < 8; i++) r |= (uint64_t)(v[i] & 0xFF) << (8 * i);
    return r;
}
/* CompareWitness<_, n>: is_comp[n], lhs limb, rhs limb, 1 / (lhs - rhs), is_less_than (unsigned/cmp.rs:13-47) */
static void compare_witness(const uint8_t* x, const uint8_t* y, int n, uint32_t* w) {
    memset(w, 0, sizeof(uint32_t) * (size_t)(n + 4));
    for (int i = n - 1; i >= 0; i--)
        if (x[i] != y[i]) {
            w[i] = 1;
            w[n] = x[i];
            w[n + 1] = y[i];
            w[n + 2] = tinv(tsub(x[i], y[i]));
            w[n + 3] = x[i] < y[i];
            return;
        }
}
static void le_bytes(uint64_t v, int n, uint8_t* out) {
    for (int i = 0; i < n; i++) out[i] = (uint8_t)(v >> (8 * i));
}
/* carries[8] + result[8] of the byte-wise product (unsigned/mul.rs:66-108) */
static void mul_witness(const uint8_t* a, const uint8_t* b, uint32_t* carries, uint32_t* res) {
    uint32_t carry = 0;
    for (int k = 0; k < 8; k++) {
        uint32_t o = carry;
        for (int i = 0; i <= k; i++) o += (uint32_t)a[i] * b[k - i];
        res[k] = o & 0xFF;
        carry = (o >> 8) & 0xFFFF;
        carries[k] = carry;
    }
}

/* returns the number of variables pushed (the chip's `populate_witness` return), witness words appended to aux */
static int extern_witness(rowctx* c, int kind, const uint32_t* in, int n_in) {
    uint32_t w[1024];
    switch (kind) {
        case X_HASHER: {
            const int nw = 8 + or_p2_num_cols(n_in);
            or_p2_wide_witness(n_in, 1, in, w);
            for (int i = 0; i < nw; i++) push_aux(c, w[i]);
            or_p2_permute(n_in, 1, in, c->m + c->nm);
            c->nm += (uint32_t)n_in;
            return n_in;
        }
        case X_U64_ADD:
        case X_U64_SUB: {
            const uint64_t a = u64_of(in), b = u64_of(in + 8), r = kind == X_U64_ADD ? a + b : a - b;
            for (int i = 0; i < 8; i++) {
                const uint32_t v = (uint32_t)((r >> (8 * i)) & 0xFF);
                push_aux(c, v);
                c->m[c->nm++] = v;
            }
            return 8;
        }
        case X_U64_LESSTHAN: {
            uint8_t x[8], y[8];
            le_bytes(u64_of(in), 8, x), le_bytes(u64_of(in + 8), 8, y);
            compare_witness(x, y, 8, w);
            for (int i = 0; i < 12; i++) push_aux(c, w[i]);
            c->m[c->nm++] = w[11];
            return 1;
        }
        case X_U64_ISZERO: {
            const uint64_t a = u64_of(in);
            memset(w, 0, 9 * sizeof(uint32_t));
            for (int i = 0; i < 8; i++)
                if ((a >> (8 * i)) & 0xFF) {
                    w[i] = tinv((uint32_t)((a >> (8 * i)) & 0xFF));
                    break;
                }
            w[8] = a == 0;
            for (int i = 0; i < 9; i++) push_aux(c, w[i]);
            c->m[c->nm++] = w[8];
            return 1;
        }
        case X_U64_MUL: {
            uint8_t x[8], y[8];
            le_bytes(u64_of(in), 8, x), le_bytes(u64_of(in + 8), 8, y);
            mul_witness(x, y, w, w + 8);
            for (int i = 0; i < 16; i++) push_aux(c, w[i]);
            for (int i = 0; i < 8; i++) c->m[c->nm++] = w[8 + i];
            return 8;
        }
        case X_U64_DIVREM: { /* unsigned/div_rem.rs:16-31,65-124 */
            const uint64_t a = u64_of(in), b = u64_of(in + 8);
            if (b == 0) return -1;
            const uint64_t qv = a / b, rem = a % b, qb = qv * b;
            uint8_t bb[8], qq[8], rr[8], qbb[8], aa[8];
            le_bytes(b, 8, bb), le_bytes(qv, 8, qq), le_bytes(rem, 8, rr), le_bytes(qb, 8, qbb), le_bytes(a, 8, aa);
            int at = 0;
            memset(w, 0, 62 * sizeof(uint32_t));
            for (int i = 0; i < 8; i++)
                if (bb[i]) {
                    w[i] = tinv(bb[i]);
                    break;
                }
            at = 8;
            for (int i = 0; i < 8; i++) w[at++] = qq[i];
            mul_witness(qq, bb, w + at, w + at + 8);
            at += 16;
            for (int i = 0; i < 8; i++) w[at++] = rr[i];
            for (int i = 7; i >= 0; i--) /* LessThanWitness<_, 8> of (rem, b): is_comp[8], lhs limb, rhs limb */
                if (rr[i] != bb[i]) {
                    w[at + i] = 1;
                    w[at + 8] = rr[i];
                    w[at + 9] = bb[i];
                    break;
                }
            at += 10;
            compare_witness(qbb, aa, 8, w + at);
            at += 12;
            for (int i = 0; i < at; i++) push_aux(c, w[i]);
            for (int i = 0; i < 8; i++) c->m[c->nm++] = qq[i];
            for (int i = 0; i < 8; i++) c->m[c->nm++] = rr[i];
            return 16;
        }
        case X_BIGNUM_LT: { /* big_num/cmp.rs:13-49 */
            uint32_t l = 0, r = 0;
            int idx = -1;
            for (int i = 7; i >= 0; i--)
                if (in[i] != in[8 + i]) {
                    idx = i, l = in[i], r = in[8 + i];
                    break;
                }
            int at = 0;
            for (int k = 0; k < 8; k++) w[at++] = k == idx;
            w[at++] = l, w[at++] = r;
            const uint32_t vs[2] = {l, r};
            for (int s = 0; s < 2; s++) {
                w[at++] = (vs[s] >> 24) < 0x78;
                for (int i = 0; i < 4; i++) w[at++] = (vs[s] >> (8 * i)) & 0xFF;
            }
            uint8_t x[4], y[4];
            le_bytes(l, 4, x), le_bytes(r, 4, y);
            compare_witness(x, y, 4, w + at);
            const uint32_t lt = w[at + 7];
            at += 8;
            for (int i = 0; i < at; i++) push_aux(c, w[i]);
            c->m[c->nm++] = lt;
            return 1;
        }
    }
    return -1;
}

/* header of row i (hdr_stride words): args[n_in] outs[n_out] provide[2] depth dreq[4] (the last five only for partial functions) */
int cp2_trace_func(const int32_t* prog, int width, int n_in, int n_out, int n_aux, int partial, uint32_t n_rows, uint32_t height,
                   uint32_t nonce_start, const uint32_t* hdr, uint32_t hdr_stride, const uint32_t* hints, const uint64_t* hint_off,
                   uint32_t max_vars, uint32_t* out) {
    inv_init();
    int rc = 0;
    const uint32_t aux0 = 1 + (uint32_t)n_in + (uint32_t)n_out, sel0 = aux0 + (uint32_t)n_aux;
#pragma omp parallel
    {
        uint32_t* m = malloc(sizeof(uint32_t) * (size_t)(max_vars + 64));
#pragma omp for schedule(static)
        for (uint32_t i = 0; i < height; i++) {
            uint32_t* row = out + (size_t)i * (size_t)width;
            memset(row, 0, sizeof(uint32_t) * (size_t)width);
            row[0] = (uint32_t)(((uint64_t)nonce_start + i) % TP); /* every row, padding included (trace.rs:82-84) */
            if (i >= n_rows) continue;
            const uint32_t* h = hdr + (size_t)i * hdr_stride;
            rowctx c = {row, aux0, 0, m, 0, hints + hint_off[i]};
            for (int k = 0; k < n_in; k++) row[1 + k] = m[c.nm++] = h[k];
            for (int k = 0; k < n_out; k++) row[1 + n_in + k] = h[n_in + k];
            push_aux(&c, h[n_in + n_out]);
            push_aux(&c, h[n_in + n_out + 1]);
            uint32_t own_depth = 0;
            if (partial) {
                own_depth = h[n_in + n_out + 2];
                for (int b = 0; b < 4; b++) push_aux(&c, (own_depth >> (8 * b)) & 0xFF);
                for (int r = 0; r < 2; r++) {
                    const uint32_t nonce = h[n_in + n_out + 3 + 2 * r], count = h[n_in + n_out + 4 + 2 * r];
                    push_aux(&c, nonce);
                    push_aux(&c, count);
                    push_aux(&c, count_inv(count + 1));
                }
            }
            const int32_t* pc = prog;
            for (int running = 1; running;) {
                switch (*pc++) {
                    case OP_ASSERT_NE: {
                        const int n = *pc++;
                        int found = 0;
                        for (int k = 0; k < n; k++) {
                            const uint32_t d = tsub(m[pc[k]], m[pc[n + k]]);
                            if (!found && d) {
                                push_aux(&c, tinv(d));
                                found = 1;
                            } else {
                                push_aux(&c, 0);
                            }
                        }
                        pc += 2 * n;
                        break;
                    }
                    case OP_CONTAINS: {
                        const int n = *pc++;
                        const uint32_t y = m[*pc++];
                        uint32_t acc = tsub(m[pc[0]], y);
                        for (int k = 1; k < n; k++) {
                            acc = tmul(acc, tsub(m[pc[k]], y));
                            push_aux(&c, acc);
                        }
                        pc += n;
                        break;
                    }
                    case OP_CONST: m[c.nm++] = (uint32_t)*pc++; break;
                    case OP_ADD: m[c.nm++] = tadd(m[pc[0]], m[pc[1]]), pc += 2; break;
                    case OP_SUB: m[c.nm++] = tsub(m[pc[0]], m[pc[1]]), pc += 2; break;
                    case OP_MUL: {
                        const uint32_t v = tmul(m[pc[0]], m[pc[1]]);
                        m[c.nm++] = v;
                        if (pc[2]) push_aux(&c, v);
                        pc += 3;
                        break;
                    }
                    case OP_INV: {
                        const uint32_t v = tinv(m[pc[0]]);
                        m[c.nm++] = v;
                        if (pc[1]) push_aux(&c, v);
                        pc += 2;
                        break;
                    }
                    case OP_NOT: {
                        const uint32_t a = m[pc[0]], d = a ? tinv(a) : 0, v = a == 0;
                        m[c.nm++] = v;
                        if (pc[1]) {
                            push_aux(&c, d);
                            push_aux(&c, v);
                        }
                        pc += 2;
                        break;
                    }
                    case OP_HINT: {
                        const int n = *pc++;
                        for (int k = 0; k < n; k++) {
                            const uint32_t v = *c.h++;
                            m[c.nm++] = v;
                            push_aux(&c, v);
                        }
                        break;
                    }
                    case OP_REQ: push_req(&c); break;
                    case OP_DEPTH: { /* trace.rs:235-254: callee depth bytes, DepthLessThan witness, its require */
                        const uint32_t d = *c.h++;
                        uint32_t wit[6] = {0, 0, 0, 0, 0, 0};
                        for (int b = 0; b < 4; b++) push_aux(&c, (d >> (8 * b)) & 0xFF);
                        for (int b = 3; b >= 0; b--) {
                            const uint32_t l = (d >> (8 * b)) & 0xFF, r = (own_depth >> (8 * b)) & 0xFF;
                            if (l != r) {
                                wit[b] = 1, wit[4] = l, wit[5] = r;
                                break;
                            }
                        }
                        for (int k = 0; k < 6; k++) push_aux(&c, wit[k]);
                        push_req(&c);
                        break;
                    }
                    case OP_EXTERN: {
                        const int kind = *pc++, n = *pc++;
                        uint32_t in[64];
                        for (int k = 0; k < n; k++) in[k] = m[pc[k]];
                        pc += n;
                        const int n_req = *pc++;
                        if (extern_witness(&c, kind, in, n) < 0) {
#pragma omp atomic write
                            rc = -2;
                            running = 0;
                            break;
                        }
                        for (int k = 0; k < n_req; k++) push_req(&c);
                        break;
                    }
                    case OP_RETURN:
                        row[sel0 + (uint32_t)*pc] = 1;
                        running = 0;
                        break;
                    case OP_MATCH: {
                        const int nv = *pc++;
                        const int32_t* vars = pc;
                        pc += nv;
                        const int nc = *pc++;
                        int target = -1;
                        for (int k = 0; k < nc && target < 0; k++) {
                            int eq = 1;
                            for (int v = 0; v < nv; v++) eq = eq && (uint32_t)pc[k * (nv + 1) + v] == m[vars[v]];
                            if (eq) target = pc[k * (nv + 1) + nv];
                        }
                        if (target < 0) target = pc[nc * (nv + 1)];
                        if (target < 0) {
#pragma omp atomic write
                            rc = -3;
                            running = 0;
                            break;
                        }
                        pc = prog + target;
                        break;
                    }
                    default:
#pragma omp atomic write
                        rc = -4;
                        running = 0;
                }
            }
            if (c.aux > (uint32_t)n_aux || c.nm > max_vars + 64) {
#pragma omp atomic write
                rc = -5;
            }
        }
        free(m);
    }
    return rc;
}
