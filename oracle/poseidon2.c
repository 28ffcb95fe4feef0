/* ORACLE (test infrastructure, not product code).
 *
 * CPU restatement of the reference's Poseidon2-over-BabyBear for every width it
 * configures (4..48), in canonical arithmetic:
 *
 *   - parameter selection: /root/reference/src/poseidon/config.rs:157-287
 *     (R_F = 8 for all widths; R_P = 21,12,10,13,18,21,25,30,34,38,42,46);
 *     tables: ../lurk_amd/csrc/p2_params.h (numbers from constants.rs:12-3487);
 *   - layer order: /root/reference/src/poseidon/wide/trace.rs:12-82
 *     (initial external layer; 4 x [add RC, x^7, external layer];
 *      R_P x [add RC to lane 0, x^7 on lane 0, internal layer]; 4 x external round);
 *   - internal layer: /root/reference/src/poseidon/config.rs:109-118
 *     (x_i <- x_i * diag_i + sum_j x_j);
 *   - external layer: third-party p3_poseidon2::Poseidon2ExternalMatrixGeneral
 *     (Plonky3 @ a0b92870, absent from /root/reference): per 4-lane chunk
 *     M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]], then for widths > 4 every
 *     lane i gets += sum of the M4 outputs at lanes j = i mod 4.  Pinned by the
 *     in-tree known-answer digests (tests/test_oracle_kat.py);
 *   - hash = first 8 lanes of the permutation over the preimage, no padding:
 *     /root/reference/src/core/poseidon.rs:30-38,61-63;
 *   - wide witness row: /root/reference/src/poseidon/wide/columns.rs:16-32 and
 *     /root/reference/src/core/poseidon.rs:65-72 (8 output lanes first).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../lurk_amd/csrc/p2_params.h"
#include "field.h"

#define OR_MAX_W 48

typedef struct {
    int width;
    int rounds_p;
    const uint32_t* diag;
    const uint32_t* ext_rc; /* [8][width] */
    const uint32_t* int_rc; /* [rounds_p] */
} or_p2_params;

int or_p2_lookup(int width, or_p2_params* out) {
    for (int i = 0; i < LURK_P2_NUM_WIDTHS; i++) {
        if (LURK_P2_PARAMS[i].width == width) {
            out->width = width;
            out->rounds_p = LURK_P2_PARAMS[i].rounds_p;
            out->diag = LURK_P2_PARAMS[i].diag;
            out->ext_rc = LURK_P2_PARAMS[i].ext_rc;
            out->int_rc = LURK_P2_PARAMS[i].int_rc;
            return 0;
        }
    }
    return -1;
}

static void m4(uint32_t* x) {
    static const uint32_t M[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    uint32_t y[4];
    for (int r = 0; r < 4; r++) {
        uint32_t acc = 0;
        for (int c = 0; c < 4; c++) acc = or_add(acc, or_mul(M[r][c], x[c]));
        y[r] = acc;
    }
    memcpy(x, y, sizeof y);
}

static void external_layer(int w, uint32_t* s) {
    for (int i = 0; i < w; i += 4) m4(s + i);
    /* p3's Poseidon2ExternalMatrixGeneral runs width 4 through the same arm as the larger widths, i.e. the sums are added
     * there too and the layer is 2*M4 [UPSTREAM-RECALL; no in-tree vector pins width 4] */
    uint32_t sums[4] = {0, 0, 0, 0};
    for (int i = 0; i < w; i++) sums[i & 3] = or_add(sums[i & 3], s[i]);
    for (int i = 0; i < w; i++) s[i] = or_add(s[i], sums[i & 3]);
}

static void internal_layer(const or_p2_params* p, uint32_t* s) {
    uint32_t sum = 0;
    for (int i = 0; i < p->width; i++) sum = or_add(sum, s[i]);
    for (int i = 0; i < p->width; i++) s[i] = or_add(or_mul(s[i], p->diag[i]), sum);
}

static inline uint32_t cube(uint32_t x) { return or_mul(or_mul(x, x), x); }

/* The permutation, optionally recording the wide-witness columns.
 * cols (may be NULL) is laid out as Poseidon2Cols (wide/columns.rs:16-32):
 *   external_rounds_state[8][W] | external_rounds_sbox[8][W] |
 *   internal_rounds_state_init[W] | internal_rounds_state0[R_P-1] | internal_rounds_sbox[R_P] */
static void permute_rec(const or_p2_params* p, uint32_t* s, uint32_t* cols) {
    const int w = p->width, rp = p->rounds_p;
    uint32_t* ext_state = cols;
    uint32_t* ext_sbox = cols ? cols + 8 * w : NULL;
    uint32_t* int_init = cols ? cols + 16 * w : NULL;
    uint32_t* int_state0 = cols ? cols + 17 * w : NULL;
    uint32_t* int_sbox = cols ? cols + 17 * w + (rp - 1) : NULL;

    external_layer(w, s);
    for (int half = 0; half < 2; half++) {
        for (int r = half * 4; r < half * 4 + 4; r++) {
            if (cols) memcpy(ext_state + r * w, s, (size_t)w * 4);
            for (int i = 0; i < w; i++) {
                uint32_t x = or_add(s[i], p->ext_rc[r * w + i]);
                uint32_t x3 = cube(x);
                if (cols) ext_sbox[r * w + i] = x3;
                s[i] = or_mul(x, or_mul(x3, x3));
            }
            external_layer(w, s);
        }
        if (half == 0) {
            for (int r = 0; r < rp; r++) {
                if (cols) {
                    if (r == 0) memcpy(int_init, s, (size_t)w * 4);
                    else int_state0[r - 1] = s[0];
                }
                uint32_t x = or_add(s[0], p->int_rc[r]);
                uint32_t x3 = cube(x);
                if (cols) int_sbox[r] = x3;
                s[0] = or_mul(x, or_mul(x3, x3));
                internal_layer(p, s);
            }
        }
    }
}

int or_p2_num_cols(int width) {
    or_p2_params p;
    if (or_p2_lookup(width, &p)) return -1;
    return 16 * width + width + (p.rounds_p - 1) + p.rounds_p;
}

/* explicit-parameter entry (used for the width-16 Merkle permutation whose
 * constants the caller may override) */
void or_p2_permute_with(int width, int rounds_p, const uint32_t* diag, const uint32_t* ext_rc,
                        const uint32_t* int_rc, uint32_t* state) {
    or_p2_params p = {width, rounds_p, diag, ext_rc, int_rc};
    permute_rec(&p, state, NULL);
}

int or_p2_permute(int width, size_t n, const uint32_t* in, uint32_t* out) {
    or_p2_params p;
    if (or_p2_lookup(width, &p)) return -1;
    for (size_t k = 0; k < n; k++) {
        uint32_t s[OR_MAX_W];
        memcpy(s, in + k * width, (size_t)width * 4);
        permute_rec(&p, s, NULL);
        memcpy(out + k * width, s, (size_t)width * 4);
    }
    return 0;
}

int or_p2_hash8(int width, size_t n, const uint32_t* in, uint32_t* out) {
    or_p2_params p;
    if (or_p2_lookup(width, &p)) return -1;
    for (size_t k = 0; k < n; k++) {
        uint32_t s[OR_MAX_W];
        memcpy(s, in + k * width, (size_t)width * 4);
        permute_rec(&p, s, NULL);
        memcpy(out + k * 8, s, 32);
    }
    return 0;
}

/* out row = [8 output lanes | Poseidon2Cols], row stride 8 + num_cols */
int or_p2_wide_witness(int width, size_t n, const uint32_t* in, uint32_t* out) {
    or_p2_params p;
    if (or_p2_lookup(width, &p)) return -1;
    const size_t stride = 8 + (size_t)or_p2_num_cols(width);
    for (size_t k = 0; k < n; k++) {
        uint32_t s[OR_MAX_W];
        memcpy(s, in + k * width, (size_t)width * 4);
        permute_rec(&p, s, out + k * stride + 8);
        memcpy(out + k * stride, s, 32);
    }
    return 0;
}

/* Narrow chip (P5): one row per round, R_F + R_P + 1 rows per permutation, padded with zero rows to a power of two.
 * Row = input[W] | is_init | rounds[R] | add_rc[W] | sbox_deg_3[W] | sbox_deg_7[W] | output[W]
 * (poseidon/columns.rs:16-25); row contents poseidon/columns.rs:28-89, row order and round constants
 * poseidon/trace.rs:14-46 with poseidon/config.rs:59-72 (first external half, internal, second external half). */
int or_p2_narrow_width(int width) {
    or_p2_params p;
    if (or_p2_lookup(width, &p)) return -1;
    return 5 * width + 1 + 8 + p.rounds_p;
}

int or_p2_narrow_trace(int width, size_t n, const uint32_t* in, size_t height, uint32_t* out) {
    or_p2_params p;
    if (or_p2_lookup(width, &p)) return -1;
    const int w = width, rounds = 8 + p.rounds_p;
    const size_t nc = (size_t)or_p2_narrow_width(width);
    if (height < n * (size_t)(rounds + 1)) return -2;
    memset(out, 0, height * nc * 4);
    for (size_t k = 0; k < n; k++) {
        uint32_t s[OR_MAX_W];
        memcpy(s, in + k * w, (size_t)w * 4);
        for (int row = 0; row <= rounds; row++) {
            uint32_t* c = out + (k * (size_t)(rounds + 1) + (size_t)row) * nc;
            uint32_t *c_in = c, *c_flags = c + w, *c_rc = c + w + 1 + rounds, *c_s3 = c_rc + w, *c_s7 = c_s3 + w, *c_out = c_s7 + w;
            memcpy(c_in, s, (size_t)w * 4);
            memcpy(c_rc, s, (size_t)w * 4);
            int external = 1;
            if (row == 0) {
                c_flags[0] = 1;
            } else {
                const int round = row - 1;
                c_flags[1 + round] = 1;
                if (round < 4) {
                    for (int i = 0; i < w; i++) c_rc[i] = or_add(s[i], p.ext_rc[round * w + i]);
                } else if (round < 4 + p.rounds_p) {
                    external = 0;
                    c_rc[0] = or_add(s[0], p.int_rc[round - 4]);
                } else {
                    for (int i = 0; i < w; i++) c_rc[i] = or_add(s[i], p.ext_rc[(round - p.rounds_p) * w + i]);
                }
            }
            for (int i = 0; i < w; i++) {
                c_s3[i] = cube(c_rc[i]);
                c_s7[i] = or_mul(or_mul(c_s3[i], c_s3[i]), c_rc[i]);
            }
            if (row > 0) {
                for (int i = 0; i < w; i++) s[i] = (i == 0 || external) ? c_s7[i] : c_rc[i];
            }
            if (external) external_layer(w, s);
            else internal_layer(&p, s);
            memcpy(c_out, s, (size_t)w * 4);
        }
    }
    return 0;
}

/* ---- field helpers exported for the tests ---- */
uint32_t or_f_inv(uint32_t a) { return or_inv(a); }
uint32_t or_f_mul(uint32_t a, uint32_t b) { return or_mul(a, b); }
void or_ef_mul4(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    or_ef x, y;
    memcpy(x.c, a, 16);
    memcpy(y.c, b, 16);
    or_ef z = or_ef_mul(x, y);
    memcpy(out, z.c, 16);
}
void or_ef_inv4(const uint32_t* a, uint32_t* out) {
    or_ef x;
    memcpy(x.c, a, 16);
    or_ef z = or_ef_inv(x);
    memcpy(out, z.c, 16);
}
