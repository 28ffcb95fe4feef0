// Micro-program format shared by the host emitter (lair/emit.cpp) and the trace kernel (trace.hip).
//
// One program per Lair function: the function's bytecode flattened into u32 words with
//   - variable degrees resolved on the host, so "does this Mul/Inv/Not own an aux column" is a flag
//     (the reference decides it per row from a degree map, /root/reference/src/lair/trace.rs:291-324);
//   - ops that generate no columns (AssertEq, Emit, Breakpoint, Debug) dropped;
//   - constants and match keys pre-converted to Montgomery form;
//   - blocks laid out back to back, Choose/ChooseMany holding absolute word offsets.
#pragma once
#include <stdint.h>

namespace lair {

// Device-side witness generators for extern chips (core/chipset.rs:28-63), selected by T_EXTERN.
enum ChipKind : uint32_t {
    CHIP_NONE = 0,
    CHIP_HASHER3 = 1,   // Poseidon2 width 24
    CHIP_HASHER4 = 2,   // width 32
    CHIP_HASHER5 = 3,   // width 40
    CHIP_U64_ADD = 4,
    CHIP_U64_SUB = 5,
    CHIP_U64_MUL = 6,
    CHIP_U64_DIVREM = 7,
    CHIP_U64_LESSTHAN = 8,
    CHIP_U64_ISZERO = 9,
    CHIP_BIGNUM_LESSTHAN = 10,
};

enum TraceOp : uint32_t {
    T_CONST = 1,       // [op, value_m]
    T_ADD = 2,         // [op, x, y]
    T_SUB = 3,         // [op, x, y]
    T_MUL = 4,         // [op | aux << 8, x, y]
    T_INV = 5,         // [op | aux << 8, x]
    T_NOT = 6,         // [op | aux << 8, x]
    T_ASSERT_NE = 7,   // [op | n << 8, a[n], b[n]]
    T_CONTAINS = 8,    // [op | n << 8, needle, a[n]]
    T_CALL = 9,        // [op | callee_partial << 8, n_values]   (Call and PreImg)
    T_STORE = 10,      // [op]
    T_LOAD = 11,       // [op, len]
    T_EXTERN = 12,     // [op, chip_kind, n_in, witness_size, require_size, return_size, in[n_in]]
    T_RANGE_U8 = 13,   // [op, num_requires]
    T_RETURN = 14,     // [op, ident]
    T_CHOOSE = 15,     // [op, var, n_cases, default_off (0 = none), (key_m, off)[n_cases]]  keys sorted as u32
    T_CHOOSE_MANY = 16 // [op, n_vars, n_cases, default_off, vars[n_vars], (keys_m[n_vars], off)[n_cases]]
};

constexpr uint32_t TRACE_PROGRAM_MAGIC = 0x4c414952u;  // "LAIR"
// header words
enum TraceHeader : uint32_t {
    TH_MAGIC = 0,
    TH_WIDTH,
    TH_INPUT,
    TH_OUTPUT,
    TH_AUX,
    TH_SEL,
    TH_PARTIAL,
    TH_ENTRY,
    TH_MAX_VARS,
    TH_HASH_LO,  // 64-bit FNV-1a of the program words (these two taken as zero): names the function's compiled trace kernel
    TH_HASH_HI,
    TH_WORDS
};

// per-row descriptor of the variable-length part of the row stream
struct RowMeta {
    uint32_t offset;   // word offset of the row's stream segment
    uint32_t n_hints;  // hint words, followed by 2 * n_requires words, then the depth requires
    uint32_t n_requires;
    uint32_t n_depth_requires;
};

// FNV-1a over the program words, the two hash words of the header taken as zero (never 0: 0 = "no hash")
inline uint64_t trace_program_hash(const uint32_t* prog, uint64_t n_words) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < n_words; i++) {
        const uint32_t w = (i == TH_HASH_LO || i == TH_HASH_HI) ? 0u : prog[i];
        for (int b = 0; b < 4; b++) h = (h ^ ((w >> (8 * b)) & 0xff)) * 0x100000001b3ull;
    }
    return h ? h : 1;
}

}  // namespace lair
