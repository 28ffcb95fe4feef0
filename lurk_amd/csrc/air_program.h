// Register-program format of a chip's AIR, shared by the host lowering (lair/air.cpp) and the device VM
// (air_vm.h).  All lanes of a launch run the same instruction stream (constraint evaluation has no
// data-dependent control flow), so the program is read through the scalar cache and only operands and
// results are per-lane.
//
// Words:
//   header[H_WORDS] | code[2 * n_instr] | consts[n_consts] (Montgomery form)
// Instruction = two words: w0 = op | dst << 8, w1 = a | b << 16, where a/b are 16-bit operands
//   operand = source_type << 13 | index   (index < 8192)
#pragma once
#include <stdint.h>

namespace airp {

constexpr uint32_t MAGIC = 0x50524941u;  // "AIRP"

enum Header : uint32_t {
    H_MAGIC = 0,
    H_N_INSTR,
    H_N_REGS,
    H_N_CONSTS,
    H_N_ASSERTS,        // constraint program: number of ASSERT instructions
    H_N_INTERACTIONS,   // interaction program: number of IBEGIN..IEND groups (sends first, then receives)
    H_N_SENDS,
    H_CODE_OFF,
    H_CONST_OFF,
    H_TOTAL_WORDS,
    H_FIRST_COLUMN,     // interaction program piece: permutation column of its first batch (0 for a whole program);
                        // constraint program piece: index of its first constraint
    H_PAD1,
    H_WORDS  // multiple of 4: the code starts 16-byte aligned
};

enum Op : uint32_t {
    OP_NOP = 0,  // padding: programs are a multiple of four instructions long, 16-byte aligned
    OP_ADD = 1,  // regs[dst] = a + b
    OP_SUB = 2,
    OP_MUL = 3,
    OP_ASSERT = 4,  // a must vanish on every row
    OP_IBEGIN = 5,  // dst = interaction kind (argument index), a = is_send, b = number of values
    OP_IVAL = 6,    // a = next value of the tuple
    OP_IEND = 7,    // a = multiplicity
    // compact interaction pieces (the prover kernels): constant tuple elements are folded into a per-interaction start value
    // (alpha + kind + sum of beta^t * constant, computed once per proof), so IBEGIN's b is the interaction's index in the
    // chip's send-then-receive list and every remaining value names its own position t (power of beta)
    OP_IVALS = 8,   // dst = count, a = first main column, b = t of the first value: `count` consecutive columns, t ascending
    OP_IVALT = 9    // dst = t, a = the value
};

enum Src : uint32_t {
    S_REG = 0,
    S_MAIN = 1,
    S_MAIN_NEXT = 2,
    S_PREP = 3,
    S_PREP_NEXT = 4,
    S_CONST = 5,
    S_PUBLIC = 6,
    S_SEL = 7  // index 0: is_first_row, 1: is_last_row, 2: is_transition
};

constexpr uint32_t MAX_CONSTRAINT_PARTS = 8;  // pieces the constraint program is cut into (one wave each in the quotient kernel)

constexpr uint32_t SRC_SHIFT = 13;
constexpr uint32_t SRC_MASK = (1u << SRC_SHIFT) - 1u;
inline constexpr uint32_t operand(uint32_t type, uint32_t index) { return (type << SRC_SHIFT) | index; }

}  // namespace airp
