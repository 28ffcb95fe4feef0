/* A C client of the boundary, compiled against include/lurkhip.h and linked to lurk_amd/liblurkhip.so by
 * tests/test_abi_client_gpu.py: ctx_create -> poseidon2_hash8 -> commit -> commitment_open, the results printed as words.  The
 * Python mirror goes through a hand-kept ctypes table (lurk_amd/_native.py); this file goes through the header the way a cgo /
 * bindgen / JNI binding would, so an argument-order or type slip between header and library shows up as a difference between the
 * two outputs.  Test infrastructure: pedantic C99, no dependency but the header. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "lurkhip.h"

#define P 2013265921u

static uint64_t state = 0x4C55524BULL; /* "LURK": the seed of SURVEY.md 8d */
static uint32_t next_elem(void) {      /* splitmix64, reduced mod p */
    uint64_t z = (state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (uint32_t)(z % P);
}

static void print_words(const char* name, const uint32_t* w, size_t n) {
    size_t i;
    printf("%s", name);
    for (i = 0; i < n; i++) printf(" %u", (unsigned)w[i]);
    printf("\n");
}

#define CHECK(call)                                                                         \
    do {                                                                                    \
        int32_t st_ = (call);                                                               \
        if (st_ != LURKHIP_OK) {                                                            \
            fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, lurkhip_last_error(ctx));    \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

int main(void) {
    lurkhip_ctx* ctx = NULL;
    enum { N_HASH = 5, W = 24, LOG_H0 = 6, W0 = 11, LOG_H1 = 4, W1 = 3 };
    uint32_t pre[N_HASH * W], dig[N_HASH * 8];
    uint32_t* m0 = (uint32_t*)malloc(sizeof(uint32_t) * (1u << LOG_H0) * W0);
    uint32_t* m1 = (uint32_t*)malloc(sizeof(uint32_t) * (1u << LOG_H1) * W1);
    const uint32_t* mats[2];
    const uint32_t log_heights[2] = {LOG_H0, LOG_H1}, widths[2] = {W0, W1};
    lurkhip_commitment* c = NULL;
    uint32_t root[8], root2[8], rows[W0 + W1], path[8 * (LOG_H0 + 1)];
    size_t i;
    printf("abi %d\n", (int)lurkhip_abi_version());
    if (lurkhip_ctx_create(0, &ctx) != LURKHIP_OK) {
        fprintf(stderr, "lurkhip_ctx_create: %s\n", lurkhip_last_error(NULL));
        return 2;
    }
    for (i = 0; i < N_HASH * W; i++) pre[i] = next_elem();
    CHECK(lurkhip_poseidon2_hash8(ctx, W, N_HASH, pre, dig, LURKHIP_REPR_CANONICAL));
    print_words("hash8", dig, N_HASH * 8);
    for (i = 0; i < ((size_t)1 << LOG_H0) * W0; i++) m0[i] = next_elem();
    for (i = 0; i < ((size_t)1 << LOG_H1) * W1; i++) m1[i] = next_elem();
    mats[0] = m0;
    mats[1] = m1;
    CHECK(lurkhip_commit(ctx, 2, mats, log_heights, widths, /*log_blowup=*/1, LURKHIP_REPR_CANONICAL, /*keep_coeffs=*/0, &c, root));
    print_words("root", root, 8);
    CHECK(lurkhip_commitment_root(ctx, c, root2, LURKHIP_REPR_CANONICAL));
    print_words("root_again", root2, 8);
    CHECK(lurkhip_commitment_open(ctx, c, /*index=*/37, rows, path, LURKHIP_REPR_CANONICAL));
    print_words("rows", rows, W0 + W1);
    print_words("path", path, 8 * (LOG_H0 + 1));
    CHECK(lurkhip_commitment_free(ctx, c));
    CHECK(lurkhip_ctx_destroy(ctx));
    free(m0);
    free(m1);
    return 0;
}
