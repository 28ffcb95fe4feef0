/* ORACLE directory (test infrastructure, not product code): shared by cpu_step.c and the chip evaluators oracle/cpu_emit.py
 * generates.  Montgomery arithmetic on BabyBear words in [0, p). */
#pragma once
#include <stdint.h>

#define CPS_P 2013265921u
#define CPS_MU 0x88000001u /* p^-1 mod 2^32 */

#ifndef CPS_HAVE_FIELD /* cpu_step.c takes these from cpu_port.c, which it includes */
static inline uint32_t mm(uint32_t a, uint32_t b) { /* a b 2^-32 mod p */
    const uint64_t t = (uint64_t)a * b;
    const uint32_t m = (uint32_t)t * CPS_MU;
    const uint32_t u = (uint32_t)(((uint64_t)m * CPS_P) >> 32), hi = (uint32_t)(t >> 32);
    const uint32_t r = hi - u;
    return hi < u ? r + CPS_P : r;
}
static inline uint32_t madd(uint32_t a, uint32_t b) {
    const uint32_t s = a + b;
    return s >= CPS_P ? s - CPS_P : s;
}
static inline uint32_t msub(uint32_t a, uint32_t b) { return a >= b ? a - b : a + CPS_P - b; }
#endif

/* One chip of the machine: sizes and its two generated evaluators (all values Montgomery).
 * inter: per interaction, sends first: multiplicity, then the tuple (tuple_len[j] words).
 * full:  the constraints of a row pair (cons[n_cons]) and the same interaction block for the local row. */
typedef struct {
    const char* name;
    uint32_t width, prep_width, n_cons, n_sends, n_recvs, inter_words;
    const uint32_t* tuple_len;
    void (*inter)(const uint32_t* loc, const uint32_t* pl, const uint32_t* pub, uint32_t* inter);
    void (*full)(const uint32_t* loc, const uint32_t* nxt, const uint32_t* pl, const uint32_t* pn, const uint32_t* pub, const uint32_t* sel,
                 uint32_t* cons, uint32_t* inter);
} cp_chip;
extern const cp_chip cp_chips[];
extern const int cp_n_chips;
