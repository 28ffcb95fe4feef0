// Device interpreter of AIR register programs (air_program.h).
//
// One trace / quotient-domain row per lane.  Every lane runs the same instruction stream, so the
// program words are fetched with wave-uniform addresses (scalar loads) and there is no divergence;
// registers live in LDS as regs[reg][lane] (conflict-free: consecutive lanes hit consecutive banks),
// main / preprocessed operands are read straight from the lane's row in global memory (a row is
// contiguous, so the lines of a row are reused across the program through L1/L2).
//
// Replaces the per-row `Air::eval` calls sphinx makes with its debug, interaction and folding builders
// (call sites: /root/reference/src/lair/lair_chip.rs:156-194; constraints /root/reference/src/lair/air.rs).
#pragma once
#include "air_program.h"
#include "babybear.h"

namespace airvm {

struct Sources {
    const uint32_t* main_l;  // the lane's local row (Montgomery)
    const uint32_t* main_n;  // the lane's next row
    const uint32_t* prep_l;
    const uint32_t* prep_n;
    const uint32_t* pub;     // public values (Montgomery), uniform
    uint32_t sel[3];         // is_first_row, is_last_row, is_transition at this lane's point
};

template <class Sink>
__device__ __forceinline__ void run(const uint32_t* __restrict__ prog, const Sources& src, uint32_t* __restrict__ regs,
                                    const uint32_t stride, Sink& sink) {
    const uint32_t n = prog[airp::H_N_INSTR];
    const uint32_t* __restrict__ code = prog + prog[airp::H_CODE_OFF];
    const uint32_t* __restrict__ consts = prog + prog[airp::H_CONST_OFF];
    // if-chains in order of frequency (registers, the row, constants; then ALU ops and tuple values): with one or two waves
    // per SIMD every scalar compare-and-branch level of a balanced switch is exposed latency
    auto fetch = [&](uint32_t o) -> uint32_t {
        const uint32_t idx = o & airp::SRC_MASK, ty = o >> airp::SRC_SHIFT;
        if (ty == airp::S_REG) return regs[idx * stride];
        if (ty == airp::S_MAIN) return src.main_l[idx];
        if (ty == airp::S_CONST) return consts[idx];
        if (ty == airp::S_MAIN_NEXT) return src.main_n[idx];
        if (ty == airp::S_PUBLIC) return src.pub[idx];
        if (ty == airp::S_PREP) return src.prep_l[idx];
        if (ty == airp::S_PREP_NEXT) return src.prep_n[idx];
        return src.sel[idx];
    };
    auto step = [&](const uint32_t w0, const uint32_t w1) {
        const uint32_t op = w0 & 0xffu, dst = w0 >> 8, a = w1 & 0xffffu, b = w1 >> 16;
        if (op == airp::OP_MUL) regs[dst * stride] = bb::mul(fetch(a), fetch(b));
        else if (op == airp::OP_IVAL) sink.ival(fetch(a));
        else if (op == airp::OP_SUB) regs[dst * stride] = bb::sub(fetch(a), fetch(b));
        else if (op == airp::OP_ADD) regs[dst * stride] = bb::add(fetch(a), fetch(b));
        else if (op == airp::OP_IBEGIN) sink.ibegin(dst, a != 0, b);
        else if (op == airp::OP_IEND) sink.iend(fetch(a));
        else if (op == airp::OP_ASSERT) sink.assert_zero(fetch(a));
        else if (op == airp::OP_IVALS) sink.ival_run(src.main_l + a, b, dst);  // dst consecutive main columns from a, positions b..
        else if (op == airp::OP_IVALT) sink.ival_at(fetch(a), dst);
        // else OP_NOP padding
    };
    // the program is padded to a multiple of four instructions: eight words (one s_load_dwordx8) per fetch, so the
    // scalar-load latency is paid once per four instructions instead of once per instruction
    const uint4* __restrict__ code4 = reinterpret_cast<const uint4*>(code);
    for (uint32_t i = 0; i < n; i += 4) {
        const uint4 c0 = code4[i >> 1], c1 = code4[(i >> 1) + 1];
        step(c0.x, c0.y);
        step(c0.z, c0.w);
        step(c1.x, c1.y);
        step(c1.z, c1.w);
    }
}

}  // namespace airvm
