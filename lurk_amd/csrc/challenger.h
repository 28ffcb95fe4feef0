// Host side of the Fiat-Shamir transcript: Poseidon2 width 16 on the CPU and p3's DuplexChallenger.
//
// Replaces (third-party, source absent from /root/reference; [UPSTREAM-RECALL], parity unpinned):
//   p3_challenger::DuplexChallenger<Val, Perm, 16, 8> as configured by sphinx's BabyBearPoseidon2
//   (call sites: machine.config().challenger(), /root/reference/benches/fib.rs:120-124).
// The transcript stays on the host: only 8-lane digests and 4-lane extension challenges cross
// the boundary, a few dozen permutations per proof.  This is protocol state, not a CPU fallback of a kernel:
// the proof-of-work search, the only data-parallel use of the challenger, runs on the device (fri.hip).
//   observe:  clears the output buffer, buffers the value, duplexes when 8 values are buffered
//   sample:   duplexes first if inputs are pending or the output buffer is empty; pops from the END of the
//             output buffer (state[7] first)
//   duplexing: overwrite state[0..k) with the k buffered inputs, permute, output = state[0..squeeze), squeeze = 16 or 8
// (lurkhip_protocol_profile::challenger_squeeze / challenger_pop_front)
#pragma once
#include <stdint.h>

#include <vector>

#include "babybear.h"
#include "commit.h"

namespace lurkhip {

// the same permutation the Merkle kernels run, on a Montgomery-form state
inline void host_perm16(const P16Params& p, uint32_t (&s)[16]) {
    auto m4 = [](uint32_t& x0, uint32_t& x1, uint32_t& x2, uint32_t& x3) {
        uint32_t t01 = bb::add(x0, x1), t23 = bb::add(x2, x3), t0123 = bb::add(t01, t23);
        uint32_t t01123 = bb::add(t0123, x1), t01233 = bb::add(t0123, x3);
        uint32_t y3 = bb::add(t01233, bb::dbl(x0)), y1 = bb::add(t01123, bb::dbl(x2));
        uint32_t y0 = bb::add(t01123, t01), y2 = bb::add(t01233, t23);
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
    };
    auto external = [&]() {
        for (int i = 0; i < 16; i += 4) m4(s[i], s[i + 1], s[i + 2], s[i + 3]);
        uint32_t sums[4];
        for (int k = 0; k < 4; k++) sums[k] = bb::add(bb::add(s[k], s[k + 4]), bb::add(s[k + 8], s[k + 12]));
        for (int i = 0; i < 16; i++) s[i] = bb::add(s[i], sums[i & 3]);
    };
    auto ext_round = [&](int r) {
        for (int i = 0; i < 16; i++) {
            uint32_t x = bb::add(s[i], p.ext_rc[r * 16 + i]);
            s[i] = bb::pow7_from_cube(x, bb::cube(x));
        }
        external();
    };
    external();
    for (int r = 0; r < 4; r++) ext_round(r);
    for (int r = 0; r < p.rounds_p; r++) {
        uint32_t x = bb::add(s[0], p.int_rc[r]);
        s[0] = bb::pow7_from_cube(x, bb::cube(x));
        uint32_t sum = 0;
        for (int i = 0; i < 16; i++) sum = bb::add(sum, s[i]);
        const uint32_t scaled_sum = bb::mul(sum, p.sum_mult);  // diag already carries the layer's scale (commit.h: P16Params)
        for (int i = 0; i < 16; i++) s[i] = bb::add(bb::mul(s[i], p.diag[i]), scaled_sum);
    }
    for (int r = 4; r < 8; r++) ext_round(r);
}

struct Challenger {
    const P16Params* params = nullptr;
    uint32_t state[16] = {};       // Montgomery
    std::vector<uint32_t> input;   // Montgomery
    std::vector<uint32_t> output;  // Montgomery
    // lurkhip_protocol_profile: lanes offered after a permutation (8 or 16), and which end sample() pops
    int squeeze = 8;
    bool pop_front = false;

    void duplexing() {
        for (size_t i = 0; i < input.size(); i++) state[i] = input[i];
        input.clear();
        host_perm16(*params, state);
        output.assign(state, state + squeeze);
    }
    void observe_m(uint32_t v_m) {
        output.clear();
        input.push_back(v_m);
        if (input.size() == 8) duplexing();
    }
    void observe(uint32_t canonical) { observe_m(bb::to_monty(canonical % bb::P)); }
    void observe_digest_m(const uint32_t* d_m) {
        for (int i = 0; i < 8; i++) observe_m(d_m[i]);
    }
    void observe_ef_m(const bb::ef& e) {
        for (int i = 0; i < 4; i++) observe_m(e.c[i]);
    }
    uint32_t sample_m() {
        if (!input.empty() || output.empty()) duplexing();
        uint32_t v;
        if (pop_front) {
            v = output.front();
            output.erase(output.begin());
        } else {
            v = output.back();
            output.pop_back();
        }
        return v;
    }
    // lane of the freshly permuted state the next sample() returns when a permutation is due
    int first_sample_lane() const { return pop_front ? 0 : squeeze - 1; }
    bb::ef sample_ef_m() {
        bb::ef e;
        for (int i = 0; i < 4; i++) e.c[i] = sample_m();
        return e;
    }
    uint32_t sample_bits(int bits) { return bb::from_monty(sample_m()) & ((1u << bits) - 1u); }
    bool check_witness(int bits, uint32_t witness_canonical) {
        observe(witness_canonical);
        return sample_bits(bits) == 0;
    }
};

}  // namespace lurkhip
