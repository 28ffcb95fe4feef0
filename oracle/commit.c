/* ORACLE (test infrastructure, not product code).  PARITY UNPINNED for this file:
 * the algorithms below live in third-party crates that are absent from /root/reference
 * (Plonky3 @ a0b92870: p3-dft Radix2DitParallel::coset_lde_batch, p3-fri TwoAdicFriPcs::commit,
 * p3-merkle-tree FieldMerkleTreeMmcs, p3-symmetric PaddingFreeSponge / TruncatedPermutation) and the
 * reference holds no golden commitments.  They restate the published algorithms from
 * memory [UPSTREAM-RECALL]; the call site they model is `machine.prove::<LocalProver>`,
 * /root/reference/benches/fib.rs:114-124.  What is checked: the GPU path against this file, this
 * file's FFT against its own O(N^2) definition, and Merkle openings against the root.
 *
 *   LDE:     for an N x w matrix of evaluations over H = <w_N> (natural order), the (N << b) x w
 *            matrix whose row bitrev(j) is the evaluation at g * w_{N<<b}^j, g = 31,
 *            w_{2^k} = 0x1a427a41^(2^(27-k)).
 *   Merkle:  leaf = sponge(rate 8, overwrite, width-16 Poseidon2) over the concatenated rows of the
 *            tallest matrices; parent = perm(left || right)[0..8]; shorter matrices injected at the level
 *            of their height as compress(compress(l, r), sponge(row)).
 *   Width-16 permutation constants: the reference's BabyBearConfig16 (src/poseidon/config.rs:190-199);
 *            sphinx's RC_16_30 / DiffusionMatrixBabyBear are not in the tree.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "field.h"

void or_p2_permute_with(int width, int rounds_p, const uint32_t* diag, const uint32_t* ext_rc, const uint32_t* int_rc,
                        uint32_t* state);
typedef struct {
    int width;
    int rounds_p;
    const uint32_t* diag;
    const uint32_t* ext_rc;
    const uint32_t* int_rc;
} or_p2_params;
int or_p2_lookup(int width, or_p2_params* out);

#define OR_GEN 31u
#define OR_ROOT27 0x1a427a41u

static uint32_t root_of_unity(int bits) {
    uint32_t r = OR_ROOT27;
    for (int i = bits; i < 27; i++) r = or_mul(r, r);
    return r;
}

static uint32_t bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

/* ---- O(N^2) definition ----------------------------------------------------------------- */
int or_lde_naive(int log_n, int w, int log_blowup, const uint32_t* in, uint32_t* out) {
    const size_t n = (size_t)1 << log_n, m = n << log_blowup;
    const uint32_t wn = root_of_unity(log_n), wm = root_of_unity(log_n + log_blowup);
    const uint32_t wn_inv = or_inv(wn), n_inv = or_inv((uint32_t)(n % OR_P));
    uint32_t* coef = malloc(n * sizeof(uint32_t));
    if (!coef) return -1;
    for (int c = 0; c < w; c++) {
        /* c_k = 1/N sum_i e_i w^-ik */
        for (size_t k = 0; k < n; k++) {
            uint32_t acc = 0, step = or_pow(wn_inv, k), x = 1;
            for (size_t i = 0; i < n; i++) {
                acc = or_add(acc, or_mul(in[i * w + c], x));
                x = or_mul(x, step);
            }
            coef[k] = or_mul(acc, n_inv);
        }
        for (size_t j = 0; j < m; j++) {
            uint32_t pt = or_mul(OR_GEN, or_pow(wm, j));
            uint32_t acc = 0;
            for (size_t k = n; k-- > 0;) acc = or_add(or_mul(acc, pt), coef[k]);
            out[(size_t)bitrev((uint32_t)j, log_n + log_blowup) * w + c] = acc;
        }
    }
    free(coef);
    return 0;
}

/* ---- O(N log N): textbook in-place radix-2 DIT on one column ------------------------------ */
static void fft_inplace(uint32_t* a, int log_n, uint32_t root) {
    const size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev((uint32_t)i, log_n);
        if (i < j) {
            uint32_t t = a[i];
            a[i] = a[j];
            a[j] = t;
        }
    }
    for (int s = 1; s <= log_n; s++) {
        const size_t len = (size_t)1 << s, half = len >> 1;
        uint32_t wlen = root;
        for (int i = s; i < log_n; i++) wlen = or_mul(wlen, wlen);
        for (size_t i = 0; i < n; i += len) {
            uint32_t x = 1;
            for (size_t j = 0; j < half; j++) {
                uint32_t u = a[i + j], v = or_mul(a[i + j + half], x);
                a[i + j] = or_add(u, v);
                a[i + j + half] = or_sub(u, v);
                x = or_mul(x, wlen);
            }
        }
    }
}

int or_lde_fft(int log_n, int w, int log_blowup, const uint32_t* in, uint32_t* out) {
    const size_t n = (size_t)1 << log_n, m = n << log_blowup;
    const int log_m = log_n + log_blowup;
    const uint32_t wn_inv = or_inv(root_of_unity(log_n)), wm = root_of_unity(log_m);
    const uint32_t n_inv = or_inv((uint32_t)(n % OR_P));
    int rc = 0;
#pragma omp parallel for schedule(dynamic)
    for (int c = 0; c < w; c++) {
        uint32_t* buf = malloc(m * sizeof(uint32_t));
        if (!buf) {
            rc = -1;
            continue;
        }
        for (size_t i = 0; i < n; i++) buf[i] = in[i * w + c];
        fft_inplace(buf, log_n, wn_inv);
        uint32_t sh = 1;
        for (size_t i = 0; i < n; i++) {
            buf[i] = or_mul(or_mul(buf[i], n_inv), sh);
            sh = or_mul(sh, OR_GEN);
        }
        for (size_t i = n; i < m; i++) buf[i] = 0;
        fft_inplace(buf, log_m, wm);
        for (size_t j = 0; j < m; j++) out[(size_t)bitrev((uint32_t)j, log_m) * w + c] = buf[j];
        free(buf);
    }
    return rc;
}

/* ---- Merkle -------------------------------------------------------------------------------- */
/* The width-16 permutation of the tree, the sponge and the transcript.  Default: the reference's BabyBearConfig16 tables.
 * or_set_p16 installs the tables of a protocol profile (oracle/stark.py: Profile; include/lurkhip.h:
 * lurkhip_protocol_profile), including the internal layer's scale: y_i = scale * (sum_j x_j + diag_i x_i).  Restated here on
 * its own (canonical arithmetic, one statement per step of the paper's round structure) rather than through the product's
 * tables. */
static int g_p16_custom = 0, g_p16_rounds_p = 13;
static uint32_t g_p16_ext[128], g_p16_int[32], g_p16_diag[16], g_p16_scale = 1;

int or_set_p16(int rounds_p, const uint32_t* ext_rc, const uint32_t* int_rc, const uint32_t* diag, uint32_t scale) {
    if (!ext_rc) {
        g_p16_custom = 0;
        return 0;
    }
    if (rounds_p < 1 || rounds_p > 32 || scale % OR_P == 0) return -1;
    for (int i = 0; i < 128; i++) g_p16_ext[i] = ext_rc[i] % OR_P;
    for (int i = 0; i < rounds_p; i++) g_p16_int[i] = int_rc[i] % OR_P;
    for (int i = 0; i < 16; i++) g_p16_diag[i] = diag[i] % OR_P;
    g_p16_rounds_p = rounds_p;
    g_p16_scale = scale % OR_P;
    g_p16_custom = 1;
    return 0;
}

static uint32_t pow7(uint32_t x) {
    uint32_t x2 = or_mul(x, x), x3 = or_mul(x2, x), x6 = or_mul(x3, x3);
    return or_mul(x6, x);
}

static void p16_external_layer(uint32_t* s) {
    /* M4 = circ(2, 3, 1, 1) on each block of four, then every lane gains the sum of the lanes at its position mod 4 */
    for (int b = 0; b < 16; b += 4) {
        uint32_t x[4] = {s[b], s[b + 1], s[b + 2], s[b + 3]};
        for (int r = 0; r < 4; r++) {
            uint32_t acc = or_add(or_add(x[r], x[r]), or_add(or_add(x[(r + 1) & 3], x[(r + 1) & 3]), x[(r + 1) & 3]));
            acc = or_add(acc, or_add(x[(r + 2) & 3], x[(r + 3) & 3]));
            s[b + r] = acc;
        }
    }
    uint32_t sums[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; i++) sums[i & 3] = or_add(sums[i & 3], s[i]);
    for (int i = 0; i < 16; i++) s[i] = or_add(s[i], sums[i & 3]);
}

static void perm16_custom(uint32_t* s) {
    p16_external_layer(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) s[i] = pow7(or_add(s[i], g_p16_ext[r * 16 + i]));
        p16_external_layer(s);
    }
    for (int r = 0; r < g_p16_rounds_p; r++) {
        s[0] = pow7(or_add(s[0], g_p16_int[r]));
        uint32_t sum = 0;
        for (int i = 0; i < 16; i++) sum = or_add(sum, s[i]);
        for (int i = 0; i < 16; i++) s[i] = or_mul(g_p16_scale, or_add(sum, or_mul(g_p16_diag[i], s[i])));
    }
    for (int r = 4; r < 8; r++) {
        for (int i = 0; i < 16; i++) s[i] = pow7(or_add(s[i], g_p16_ext[r * 16 + i]));
        p16_external_layer(s);
    }
}

static void perm16(uint32_t* s) {
    if (g_p16_custom) {
        perm16_custom(s);
        return;
    }
    or_p2_params p;
    or_p2_lookup(16, &p);
    or_p2_permute_with(16, p.rounds_p, p.diag, p.ext_rc, p.int_rc, s);
}

/* the transcript's permutation: the same one as the tree's */
void or_perm16(uint32_t* state) { perm16(state); }

/* sponge over `count` values produced by get(i) */
typedef struct {
    int n_mats;
    const uint32_t* const* mats;
    const uint32_t* widths;
    size_t row;
} row_src;

static void sponge_rows(const row_src* src, const int* which, int n_which, uint32_t out[8]) {
    uint32_t s[16];
    memset(s, 0, sizeof s);
    int pos = 0;
    for (int k = 0; k < n_which; k++) {
        int m = which[k];
        const uint32_t* row = src->mats[m] + src->row * src->widths[m];
        for (uint32_t c = 0; c < src->widths[m]; c++) {
            s[pos++] = row[c];
            if (pos == 8) {
                perm16(s);
                pos = 0;
            }
        }
    }
    if (pos) perm16(s);
    memcpy(out, s, 32);
}

static void compress(const uint32_t* l, const uint32_t* r, uint32_t out[8]) {
    uint32_t s[16];
    memcpy(s, l, 32);
    memcpy(s + 8, r, 32);
    perm16(s);
    memcpy(out, s, 32);
}

/* digests: caller buffer of (2 * 2^log_max - 1) * 8 words, level 0 first.  Matrices are the LDE
 * matrices (heights 2^log_h[i]).  Returns 0 and the root in root[8]. */
int or_merkle_commit(int n_mats, const uint32_t* const* mats, const uint32_t* log_h, const uint32_t* widths,
                     uint32_t* digests, uint32_t* root) {
    int log_max = 0;
    for (int i = 0; i < n_mats; i++)
        if ((int)log_h[i] > log_max) log_max = (int)log_h[i];
    int* which = malloc(sizeof(int) * (size_t)n_mats);
    const size_t n_leaves = (size_t)1 << log_max;
    /* matrices of equal height keep their given order (stable sort by height, tallest first) */
    int nw = 0;
    for (int i = 0; i < n_mats; i++)
        if ((int)log_h[i] == log_max) which[nw++] = i;
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < n_leaves; r++) {
        row_src src = {n_mats, mats, widths, r};
        sponge_rows(&src, which, nw, digests + r * 8);
    }
    uint32_t* prev = digests;
    for (int l = 1; l <= log_max; l++) {
        const size_t n_par = n_leaves >> l;
        uint32_t* cur = prev + (n_par << 1) * 8;
        nw = 0;
        for (int i = 0; i < n_mats; i++)
            if ((int)log_h[i] == log_max - l) which[nw++] = i;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n_par; i++) {
            uint32_t d[8];
            compress(prev + 2 * i * 8, prev + (2 * i + 1) * 8, d);
            if (nw) {
                uint32_t h[8];
                row_src src = {n_mats, mats, widths, i};
                sponge_rows(&src, which, nw, h);
                compress(d, h, cur + i * 8);
            } else {
                memcpy(cur + i * 8, d, 32);
            }
        }
        prev = cur;
    }
    memcpy(root, prev, 32);
    free(which);
    return 0;
}

/* Verifies an opening: rows = opened rows of every matrix back to back (caller order), path = log_max
 * sibling digests, leaf level first.  Returns 1 when it reproduces `root`. */
int or_merkle_verify(int n_mats, const uint32_t* log_h, const uint32_t* widths, uint64_t index, const uint32_t* rows,
                     const uint32_t* path, const uint32_t* root) {
    int log_max = 0;
    for (int i = 0; i < n_mats; i++)
        if ((int)log_h[i] > log_max) log_max = (int)log_h[i];
    const uint32_t** ptrs = malloc(sizeof(uint32_t*) * (size_t)n_mats);
    int* which = malloc(sizeof(int) * (size_t)n_mats);
    size_t off = 0;
    for (int i = 0; i < n_mats; i++) {
        ptrs[i] = rows + off;
        off += widths[i];
    }
    row_src src = {n_mats, ptrs, widths, 0}; /* every "matrix" is a single opened row */
    int nw = 0;
    for (int i = 0; i < n_mats; i++)
        if ((int)log_h[i] == log_max) which[nw++] = i;
    uint32_t cur[8];
    sponge_rows(&src, which, nw, cur);
    for (int l = 0; l < log_max; l++) {
        uint32_t d[8];
        if (((index >> l) & 1) == 0) compress(cur, path + l * 8, d);
        else compress(path + l * 8, cur, d);
        nw = 0;
        for (int i = 0; i < n_mats; i++)
            if ((int)log_h[i] == log_max - l - 1) which[nw++] = i;
        if (nw) {
            uint32_t h[8];
            sponge_rows(&src, which, nw, h);
            compress(d, h, cur);
        } else {
            memcpy(cur, d, 32);
        }
    }
    free(ptrs);
    free(which);
    return memcmp(cur, root, 32) == 0;
}
