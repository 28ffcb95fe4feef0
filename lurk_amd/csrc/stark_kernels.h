// Device side of the prover's AIR kernels: LogUp / folding sinks and the bodies of the permutation-trace and quotient kernels,
// templated on the *runner* that executes a chip's program pieces -- the interpreter (air_vm.h) in the library build, straight-
// line code generated from the same programs when a chip is compiled at run time (jit.cpp).  Device code only.
#pragma once
#include "air_program.h"
#include "air_vm.h"
#include "babybear.h"
#include "lazy_ef.h"

namespace lurkhip {

using bb::ef;

__device__ __forceinline__ ef ef_load(const uint32_t* p) { return ef{{p[0], p[1], p[2], p[3]}}; }

// A compiled chip (jit.cpp) repeats the sinks' code once per interaction: there the long extension-field operations are
// calls, so that a piece stays within reach of the instruction cache; the interpreter has one copy and inlines them.
#ifdef LURKHIP_COMPILED_AIR
#define LURKHIP_SINK_OP __attribute__((noinline))
#else
#define LURKHIP_SINK_OP __forceinline__
#endif
__device__ LURKHIP_SINK_OP ef sink_ef_inv(ef a) { return bb::ef_inv(a); }
__device__ LURKHIP_SINK_OP ef sink_ef_mul(ef a, ef b) { return bb::ef_mul(a, b); }

// Copies rows idx[0..n_rows) of a row-major matrix into an LDS tile with row stride wp: lanes run along a row, so
// every global access is one contiguous w*4-byte segment (the per-lane strided reads the VM would otherwise issue
// thrash L1: a workgroup's rows are hundreds of KB apart from lane to lane).  The (row, column) pairs are dealt to all the
// threads of the workgroup and eight loads go out before the first is awaited: one load per trip, awaited on the spot, made
// the staging a chain of n_rows memory round trips -- the longest phase of the permutation and quotient kernels.
// LURK_STAGE_UR loads go out before the first is awaited (A/B: LURKHIP_JIT_DEFINES=LURK_STAGE_UR=16 for the compiled kernels).
#ifndef LURK_STAGE_UR
#define LURK_STAGE_UR 8
#endif
// How many batches ahead the quotient's interaction waves ask for their permutation-column entries (QuotientSink::batch_live).
#ifndef LURK_QUOT_ENTRY_DEPTH
#define LURK_QUOT_ENTRY_DEPTH 1
#endif
__device__ __forceinline__ void stage_rows(uint32_t* __restrict__ tile, uint32_t wp, const uint32_t* __restrict__ mat, uint32_t w,
                                           const uint32_t* __restrict__ idx, uint32_t n_rows, uint32_t pitch = 0 /* words between rows; 0: w */) {
    if (pitch == 0) pitch = w;
    constexpr int UR = LURK_STAGE_UR;
    const uint32_t total = n_rows * w;  // < 2^24: the quotient e / w below is exact after one correction step
    const float inv_w = 1.0f / (float)w;
    for (uint32_t e0 = threadIdx.x; e0 < total; e0 += blockDim.x * UR) {
        uint32_t v[UR], at[UR];
#pragma unroll
        for (int k = 0; k < UR; k++) {
            const uint32_t e = e0 + (uint32_t)k * blockDim.x;
            const uint32_t ec = e < total ? e : total - 1u;  // past the end: re-read the last word, not stored
            uint32_t r = (uint32_t)((float)ec * inv_w);
            r += (r + 1u) * w <= ec ? 1u : 0u;
            r -= r * w > ec ? 1u : 0u;
            const uint32_t c = ec - r * w;
            at[k] = r * wp + c;
            v[k] = mat[(size_t)idx[r] * pitch + c];
        }
#pragma unroll
        for (int k = 0; k < UR; k++)
            if (e0 + (uint32_t)k * blockDim.x < total) tile[at[k]] = v[k];
    }
}

// Running (numerator, denominator) of the current batch of interactions; shared by the permutation trace
// and the quotient kernel (where entry * den - num is the batch's constraint).
struct LogupAccum {
    const uint32_t* __restrict__ beta_pows;  // centred, 8 words per power (k_ef_powers)
    const uint32_t* __restrict__ starts;     // per interaction: alpha + kind + sum of beta^t * (constant tuple elements)
    LazyEf cur64;
    ef cur, num, den;
    uint32_t in_batch = 0, m_first = 0;
    bool is_send = false;
    __device__ __forceinline__ void begin(uint32_t interaction, bool send) {
        // (wave-uniform, written by an earlier kernel: through the constant address space like the power tables, lazy_ef.h)
        const __attribute__((address_space(4))) uint32_t* sp = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)(starts + 4 * interaction);
        cur64.set(ef{{sp[0], sp[1], sp[2], sp[3]}});
        is_send = send;
    }
    // the tuple element at position t (t = 1 + index in the tuple): += beta^t * v
    __device__ __forceinline__ void value_at(uint32_t v, uint32_t t) {
        int32_t p[8];
        load_w8(p, beta_pows + 8 * t);
        cur64.add_base(v, p);
    }
    // `count` elements from consecutive words at positions t, t + 1, ...; the (scalar) table load of the next power overlaps
    // the current element's arithmetic (the table has max_tuple + 2 entries: one past the last position is valid)
    __device__ __forceinline__ void value_run(const uint32_t* __restrict__ vals, uint32_t t, uint32_t count) {
        int32_t p[8];
        load_w8(p, beta_pows + 8 * t);
        for (uint32_t k = 0; k < count; k++) {
            int32_t q[8];
            load_w8(q, beta_pows + 8 * (t + k + 1));
            cur64.add_base(vals[k], p);
#pragma unroll
            for (int c = 0; c < 8; c++) p[c] = q[c];
        }
    }
    // folds the finished interaction into the batch fraction num / den = sum_i m_i / d_i; returns true when the batch
    // holds `batch` interactions.  Multiplicities are base-field: the first two interactions of a batch cost one
    // extension product (d_1 d_2) and two scalings.
    __device__ __forceinline__ bool end(uint32_t mult, uint32_t batch) {
        cur = cur64.value();
        const uint32_t m = is_send ? mult : bb::neg(mult);
        if (in_batch == 0) {
            m_first = m;
            den = cur;
        } else if (in_batch == 1) {
            num = bb::ef_add(bb::ef_scale(cur, m_first), bb::ef_scale(den, m));
            den = sink_ef_mul(den, cur);
        } else {
            num = bb::ef_add(sink_ef_mul(num, cur), bb::ef_scale(den, m));
            den = sink_ef_mul(den, cur);
        }
        in_batch++;
        return in_batch == batch;
    }
    // the same with the position in the batch known when the code is generated (jit.cpp): the two dead arms are not emitted
    template <int POS>
    __device__ __forceinline__ void end_at(uint32_t mult) {
        cur = cur64.value();
        const uint32_t m = is_send ? mult : bb::neg(mult);
        if (POS == 0) {
            m_first = m;
            den = cur;
        } else if (POS == 1) {
            num = bb::ef_add(bb::ef_scale(cur, m_first), bb::ef_scale(den, m));
            den = sink_ef_mul(den, cur);
        } else {
            num = bb::ef_add(sink_ef_mul(num, cur), bb::ef_scale(den, m));
            den = sink_ef_mul(den, cur);
        }
        in_batch = POS + 1;
    }
    // numerator of the (possibly partial) batch
    __device__ __forceinline__ ef numerator() const { return in_batch == 1 ? bb::ef_from_base(m_first) : num; }
};

struct PermSink {
    LogupAccum acc;
    uint32_t batch;
    uint32_t* out_row;  // [perm_width * 4]
    uint32_t col = 0;
    ef row_sum = bb::ef_zero();
    bool live = true;  // lanes past the last row run along (workgroup barriers) and store nothing
    // Round 5: dead batches.  A row of a Lair function takes one branch and every branch has its own lookups: most interactions
    // of a chip have multiplicity zero on most rows, and those of branches a shard never takes on every row (a real `(fib N)`:
    // 52 of eval_builtin_expr's 78 batch columns, tests/golden/fib_shape.json "lookup_sparsity").  A batch all of whose
    // multiplicities are zero on all 64 rows of the wave has the entry 0 whatever its denominators are: compiled pieces test that
    // first (batch_live, wave-uniform) and skip the fingerprints, the products and the inverse.  `col_live` (optional) receives a 1
    // per column some wave computed: a column nobody marks is identically zero and its LDE need not be computed (prover.hip).
    uint32_t* col_live = nullptr;
    bool marker = false;  // lane 0 of the wave
    __device__ __forceinline__ bool batch_live(uint32_t mults_or) const { return __builtin_amdgcn_ballot_w64(mults_or != 0u) != 0ull; }
    __device__ __forceinline__ void skip_batch() {
        uint4* dst = reinterpret_cast<uint4*>(out_row + 4 * col);
#ifndef LURK_AB_PERM_NO_STORE  // (diagnostic: what the permutation rows' per-lane 16-byte stores cost -- no output)
        if (live) *dst = make_uint4(0u, 0u, 0u, 0u);
#endif
        col++;
    }
    __device__ __forceinline__ void assert_zero(uint32_t) {}
    __device__ __forceinline__ void ibegin(uint32_t, bool send, uint32_t interaction) { acc.begin(interaction, send); }
    __device__ __forceinline__ void ival(uint32_t) {}  // compact pieces carry no plain IVAL
    __device__ __forceinline__ void ival_at(uint32_t v, uint32_t t) { acc.value_at(v, t); }
    __device__ __forceinline__ void ival_run(const uint32_t* vals, uint32_t t, uint32_t count) { acc.value_run(vals, t, count); }
    __device__ __forceinline__ void flush() {
        ef v = acc.in_batch == 1 ? bb::ef_scale(sink_ef_inv(acc.den), acc.m_first) : sink_ef_mul(acc.num, sink_ef_inv(acc.den));
        uint4* dst = reinterpret_cast<uint4*>(out_row + 4 * col);
#ifndef LURK_AB_PERM_NO_STORE
        if (live) *dst = make_uint4(v.c[0], v.c[1], v.c[2], v.c[3]);
#else
        if (live && v.c[0] == 0x7fffffffu) *dst = make_uint4(v.c[0], v.c[1], v.c[2], v.c[3]);  // (keeps the value alive, never stores)
#endif
        if (col_live && marker) col_live[col] = 1u;
        row_sum = bb::ef_add(row_sum, v);
        col++;
        acc.in_batch = 0;
    }
    __device__ __forceinline__ void iend(uint32_t m) {
        if (acc.end(m, batch)) flush();
    }
    // compiled pieces: position in the batch (0, 1, 2 = later) and whether the batch ends here are literals
    template <int POS, bool LAST>
    __device__ __forceinline__ void iend_at(uint32_t m) {
        acc.end_at<POS>(m);
        if (LAST) flush();
    }
};

// Program pieces of a launch: one wave of every workgroup per piece, all over the same 64 staged rows.
constexpr int MAX_VM_PARTS = 16;  // up to 8 constraint pieces + 6 interaction pieces: 1024-thread workgroups at most
struct VmParts {
    const uint32_t* prog[MAX_VM_PARTS];
    uint32_t reg_off[MAX_VM_PARTS];  // word offset of the piece's register file regs[n_regs][64] in LDS
    uint32_t n_parts;
    // Round 6: wave w runs piece piece_of[w].  The waves of a workgroup go to the CU's four SIMDs round robin (wave w -> SIMD w mod 4)
    // and a SIMD issues for one wave at a time, so a workgroup lasts as long as its most loaded SIMD: the host deals the pieces to the
    // waves so that the SIMDs' sums of piece lengths are level (stark.hip: balance_parts) -- in piece order the first SIMD of an
    // eval_builtin_expr quotient workgroup carried 1823 program words against 1104 on the last.
    uint8_t piece_of[MAX_VM_PARTS];
};

struct PermArgs {
    VmParts parts;
    const uint32_t* main;
    const uint32_t* prep;
    const uint32_t* beta_pows;
    const uint32_t* starts;  // per interaction: alpha + kind + sum beta^t * constants (k_interaction_starts)
    uint32_t n, w, pw, perm_w, batch;
    uint32_t* out;
    uint32_t regs_words;  // all register files
    uint32_t wp;
    int staged;
    // Round 5: words between rows of the main trace and of the permutation trace (>= w, 4 perm_w).  Inside the prover both are
    // column ranges of aligned group buffers, so that the LDE's first pass reads whole 128-byte lines.
    uint32_t main_pitch, out_pitch;
    uint32_t* col_live;  // [perm_w - 1] or null: 1 = some wave computed the batch column (PermSink)
};

// Workgroup = 64 rows x n_parts waves: wave j runs interaction piece j (its own permutation columns) on the shared tile;
// the pieces' row sums meet in LDS and wave 0 writes the last column.
template <class Runner>
__device__ __forceinline__ void perm_rows_body(const PermArgs& a) {
    extern __shared__ uint32_t lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t i = blockIdx.x * 64u + lane;
    const bool live = i < a.n;
    const uint32_t ic = live ? i : 0u;
    const uint32_t nx = ic + 1 >= a.n ? 0 : ic + 1;
    const uint32_t* main_l = a.main + (size_t)ic * a.main_pitch;
    uint32_t* tile = lds + a.regs_words;
    uint32_t* idx = tile + (a.staged ? 64u * a.wp : 0u);
    uint32_t* sums = idx + 64;  // [n_parts][64][4]
    if (a.staged) {
        // interactions only read the local row
        if (wave == 0) idx[lane] = ic;
        __syncthreads();
#ifndef LURK_AB_NO_STAGE
        stage_rows(tile, a.wp, a.main, a.w, idx, 64u, a.main_pitch);
#endif
        __syncthreads();
        main_l = tile + lane * a.wp;
    }
    const uint32_t piece = __builtin_amdgcn_readfirstlane((uint32_t)a.parts.piece_of[wave]);
    const uint32_t* prog = a.parts.prog[piece];
    airvm::Sources src{main_l, a.main + (size_t)nx * a.main_pitch, a.prep + (size_t)ic * a.pw, a.prep + (size_t)nx * a.pw, nullptr, {0u, 0u, 0u}};
    PermSink sink{LogupAccum{a.beta_pows, a.starts}, a.batch, a.out + (size_t)ic * a.out_pitch};
    sink.col = prog[airp::H_FIRST_COLUMN];
    sink.live = live;
    sink.col_live = a.col_live;
    sink.marker = lane == 0u;
    Runner::run(prog, piece, src, lds + a.parts.reg_off[piece] + lane, sink);
    if (sink.acc.in_batch) sink.flush();
    if (a.parts.n_parts > 1) {
#pragma unroll
        for (int c = 0; c < 4; c++) sums[(wave * 64u + lane) * 4 + c] = sink.row_sum.c[c];
        __syncthreads();
        if (wave != 0) return;
        for (uint32_t j = 1; j < a.parts.n_parts; j++) sink.row_sum = bb::ef_add(sink.row_sum, ef_load(sums + (j * 64u + lane) * 4));
    }
    if (!live) return;
    // the row's sum goes to the last column; the scan below turns it into the running sum
    uint4* dst = reinterpret_cast<uint4*>(sink.out_row + 4 * (a.perm_w - 1));
    *dst = make_uint4(sink.row_sum.c[0], sink.row_sum.c[1], sink.row_sum.c[2], sink.row_sum.c[3]);
}

// ---------------------------------------------------------------- quotient values
// One row of the quotient domain g * <w_Q>, Q = N << log_quotient_degree, per lane.  The committed LDEs are
// stored in bit-reversed row order, so lane s works on natural index i = bitrev(s): its "local" rows are
// the contiguous storage rows of the launch, its "next" row (i + Q/N) is another storage row.
// Constraint k of the chip (then one per permutation batch column, then the three running-sum constraints) is
// folded as sum_k alpha^(K-1-k) C_k(x), which is sphinx's Horner accumulation `acc = acc * alpha + C_k`
// [UPSTREAM-RECALL: ProverConstraintFolder], and multiplied by 1 / Z_H(x).
struct QuotientArgs {
    VmParts parts;          // pieces [0, n_cons_parts): the constraint program pieces, then the interaction program pieces
    uint32_t n_cons_parts;
    uint32_t n_cons;        // constraints of the chip (the interaction batches' constraints follow them)
    const uint32_t* main;   // LDE matrices, bit-reversed rows, Montgomery
    const uint32_t* prep;
    const uint32_t* perm;   // 4 * perm_w base columns
    const uint32_t* pub;
    const uint32_t* alpha_pows;  // alpha^j, j < k_total
    const uint32_t* beta_pows;
    const uint32_t* starts;  // per interaction: alpha + kind + sum beta^t * constants (k_interaction_starts)
    ef cumulative_sum;
    uint32_t log_n, log_q, w, pw, perm_w, batch, k_total;
    uint32_t main_pitch, prep_pitch, perm_pitch;  // words between rows of the three LDE matrices (>= w, pw, 4 perm_w: lurkhip_commitment::pitch)
    uint32_t zh_inv[4];     // 1 / Z_H(x) for i mod 2^lqd
    uint32_t zh[4];
    uint32_t g_m, wq_m, wn_inv_m;
    const uint32_t* sel;    // [2^log_q][3]: is_first_row, is_last_row, is_transition per (bit-reversed) row, or null: computed per row
    uint32_t regs_words, wp;    // LDS layout (layout_parts)
    int staged;
    uint32_t* out;          // [2^lqd][N][4]
    // Round 5.  The transition constraint of the running sum, (phi(xw) - phi(x) - S(xw)) * (x - w^-1) with S = the sum of the batch
    // columns, needs S on the NEXT row: the kernel used to read the whole next row of the permutation LDE for it (one lane per row,
    // perm_w dependent 16-byte loads: 36 % of a fib step's quotient traffic, and the latency chain `SQ_WAIT_ANY 58 %` pointed at).
    // On a trace the prover built itself phi IS the running sum of S and ends in the cumulative sum c, so as polynomials of degree
    // < N:  S(X) = phi(X) - phi(X / w) + L_0(X) c  (both sides interpolate S on the trace domain; L_0 = the first row's Lagrange
    // basis).  Hence S(xw) = phi(xw) - phi(x) + L_0(xw) c and the constraint's value is -L_0(xw) c (x - w^-1)
    // = -c Z_H(x) / (N w): nothing of the next row is read.  Same field element as the direct evaluation -- the proofs do not change
    // by a bit (tests/test_prover_gpu.py, test_cpu_step_gpu.py) -- but only for an honest running sum: the standalone entry point
    // lurkhip_quotient_dev, which may be handed any matrices, keeps the direct evaluation (honest_running_sum = 0).
    int honest_running_sum;
    ef trans_const;         // -cumulative_sum / (N w_N)
    // Round 5: the chip's cumulative sum may still be on its way (the constraint-folding challenge was drawn on the device, the host
    // has not read the permutation stage back yet): then it is read from here (4 words, Montgomery) and trans_const is
    // cumulative_sum * trans_scale, trans_scale = -1 / (N w_N); null: the two fields above hold the values
    const uint32_t* cumsum_dev;
    uint32_t trans_scale;
    // Round 6, one shard over several ranks (split.hip): the launch covers the n_rows storage rows from s_base on, of which the
    // matrices hold exactly those (main / prep / perm point at storage row s_base); the few columns the AIR reads on the next row
    // (nonce; is_real and ptr of a memory chip) travel as copies `next_off` words after the local row's first column, because the
    // next row's storage row belongs to another rank; values go to out[brev(local row)][4], natural order inside the rank's
    // sub-coset (split_plan.h: rows_of).  split = 0: the whole domain, as before.
    uint32_t split, s_base, n_rows, next_off, log_rows;
    // Round 6: one word per batch column, 0 = the permutation stage found no wave that computed it (stark_kernels.h: PermSink): the
    // column is identically zero, and so are its interactions' multiplicities -- the quotient's interaction waves step over it without
    // asking memory for its entry.  Null: every column is tested on the wave's own values (batch_live), as in round 5.
    const uint32_t* col_live;
};

struct QuotientSink {
    const uint32_t* __restrict__ alpha_pows;  // centred, 8 words per power
    uint32_t k_total;
    LogupAccum acc;
    uint32_t batch;
    const uint32_t* perm_l;
    uint32_t k = 0, col = 0;
    LazyEf folded;
    ef sum_cols;        // sum of the permutation entries this piece's batch columns read (the running-sum constraint needs their total)
    int32_t next_w[8];  // alpha^(K-1-k), loaded one constraint ahead (see LogupAccum::next_pow)
    __device__ __forceinline__ void prime(uint32_t k_start) {
        folded.zero();
        sum_cols = bb::ef_zero();
        seek(k_start);
    }
    __device__ __forceinline__ void seek(uint32_t k_next) {
        k = k_next;
        load_w8(next_w, alpha_pows + 8 * (k_total - 1 - k));
    }
    __device__ __forceinline__ void weight(int32_t (&w)[8]) {
#pragma unroll
        for (int c = 0; c < 8; c++) w[c] = next_w[c];
        // k + 1 <= k_total - 1 except after the last constraint, where index 0 is re-read (valid, unused)
        const uint32_t nk = k + 1 < k_total ? k_total - 2 - k : 0u;
        load_w8(next_w, alpha_pows + 8 * nk);
    }
    __device__ __forceinline__ void assert_zero(uint32_t v) {
        int32_t w[8];
        weight(w);
        folded.add_base(v, w);
        k++;
    }
    // compiled constraint pieces (jit.cpp: emit_constraint_function): a group of `n` constraints that all have the factor `c`
    __device__ __forceinline__ bool cond_live(uint32_t c) const { return __builtin_amdgcn_ballot_w64(c != 0u) != 0ull; }
    __device__ __forceinline__ void skip_asserts(uint32_t n) {
        k += n;
        load_w8(next_w, alpha_pows + 8 * (k < k_total ? k_total - 1 - k : 0u));  // (past the last constraint index 0 is re-read: valid, unused)
    }
    __device__ __forceinline__ void assert_zero_ext(const ef& v) {
        int32_t w[8];
        weight(w);
        folded.add_ext(v, w);
        k++;
    }
    __device__ __forceinline__ void ibegin(uint32_t, bool send, uint32_t interaction) { acc.begin(interaction, send); }
    __device__ __forceinline__ void ival(uint32_t) {}  // compact pieces carry no plain IVAL
    __device__ __forceinline__ void ival_at(uint32_t v, uint32_t t) { acc.value_at(v, t); }
    __device__ __forceinline__ void ival_run(const uint32_t* vals, uint32_t t, uint32_t count) { acc.value_run(vals, t, count); }
    // Round 5: dead batches on the quotient domain.  The batch constraint is entry * prod(d_i) - sum_i m_i prod_{j != i} d_j: where
    // the column's entry AND every multiplicity are zero it is zero whatever the denominators are.  That is the case on the whole
    // coset for a batch of interactions no row of the shard uses (their multiplicities are sums of selector columns that are zero
    // columns, hence zero polynomials, and so is the batch column: 42 % of the permutation cells of a real `(fib N)` shard), and
    // the test below does not rely on it: it looks at the values of THIS wave's 64 points.  Compiled pieces ask before the
    // fingerprints (batch_live) and, on no, only step the constraint index past the batch (skip_batch).
    // The column's entry is asked for one batch AHEAD (prefetch, then at every test): a lane's entries are perm_pitch words apart from
    // its neighbours', every load is a memory round trip of its own, and the test needs the value before anything else of the batch
    // has been issued -- loaded on the spot it exposed that latency once per batch (a dozen times per wave).
    // Round 6: the wait that was left.  A shard's dead columns come in RUNS (the lookups of a never-taken branch are neighbours), a
    // dead batch is a dozen instructions, and each one's test waited for its own entry -- a chain of dependent memory round trips,
    // one per dead column, that the one-ahead request cannot cover (nothing runs between two of them).  The permutation stage already
    // knows which columns nobody computed (col_live, one word per column): their entry is zero without asking memory (a scalar load of
    // the flag), so a run of dead columns is a run of register-only tests.  (Measured first and rejected: more entries in flight -- LURK_QUOT_ENTRY_DEPTH 3 and 6: quotient_all
    // 4.2 -> 4.7 ms; the compiler waits for ALL outstanding loads at the join of the live and dead paths.)
    uint4 next_e = make_uint4(0u, 0u, 0u, 0u), cur_e = make_uint4(0u, 0u, 0u, 0u);
    const uint32_t* col_live = nullptr;
    uint32_t last_col = 0;  // perm_w - 1: the running sum's column, the last of the row
    bool have_e = false;
    __device__ __forceinline__ bool flagged_dead(uint32_t c) const {
        if (col_live == nullptr || c >= last_col) return false;
        const __attribute__((address_space(4))) uint32_t* lp = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)(col_live + c);  // wave-uniform
        return *lp == 0u;
    }
    __device__ __forceinline__ void request(uint32_t c) {
#ifndef LURK_AB_NO_ENTRY  // (diagnostic: what the entries' loads cost -- wrong values)
        if (!flagged_dead(c)) next_e = *reinterpret_cast<const uint4*>(perm_l + 4 * c);  // (past a piece's last batch: the next piece's column or the running sum's -- valid, unused)
#endif
    }
    __device__ __forceinline__ void prefetch() { request(col); }
    __device__ __forceinline__ bool batch_live(uint32_t mults_or) {
        // (a flagged column's entry is zero on the whole coset -- the LDE of a zero column -- and is not read; the multiplicities are
        // still tested on the wave's own points: they vanish on the trace domain, which says nothing about a product of columns elsewhere)
        cur_e = flagged_dead(col) ? make_uint4(0u, 0u, 0u, 0u) : next_e;
        have_e = true;
        request(col + 1);
        return __builtin_amdgcn_ballot_w64((mults_or | cur_e.x | cur_e.y | cur_e.z | cur_e.w) != 0u) != 0ull;
    }
    __device__ __forceinline__ void skip_batch() {
        int32_t w[8];
        weight(w);  // (keeps the one-ahead load of the next constraint's weight going)
        k++;
        col++;
        have_e = false;
    }
    __device__ __forceinline__ void flush() {
        // entry * prod(rlc) - sum_i m_i prod_{j != i} rlc_j
        ef entry = have_e ? ef{{cur_e.x, cur_e.y, cur_e.z, cur_e.w}} : ef_load(perm_l + 4 * col);
        have_e = false;
        sum_cols = bb::ef_add(sum_cols, entry);
        assert_zero_ext(bb::ef_sub(sink_ef_mul(acc.den, entry), acc.numerator()));
        col++;
        acc.in_batch = 0;
    }
    __device__ __forceinline__ void iend(uint32_t m) {
        if (acc.end(m, batch)) flush();
    }
    template <int POS, bool LAST>
    __device__ __forceinline__ void iend_at(uint32_t m) {
        acc.end_at<POS>(m);
        if (LAST) flush();
    }
};

// Workgroup = 64 quotient-domain rows x n_parts waves over one staged tile: waves [0, n_cons_parts) fold the pieces of the chip's
// constraints (piece j from its first constraint's index on), the waves after them the batch constraints of the interaction
// pieces (weights alpha^(K-1-k) at their own k); the partial folds meet in LDS and wave 0 adds the running-sum constraints
// and stores the quotient value.
template <class Runner>
__device__ __forceinline__ void quotient_body(const QuotientArgs& a) {
    extern __shared__ uint32_t regs[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t s_raw = blockIdx.x * 64u + lane;
    const uint32_t q = 1u << a.log_q;
    const bool live = s_raw < (a.split ? a.n_rows : q);
    const uint32_t sl = live ? s_raw : 0u;   // row of the launch's matrices
    const uint32_t s = a.s_base + sl;        // storage row of the quotient domain (s_base = 0 unless split)
    const uint32_t lqd = a.log_q - a.log_n, qd = 1u << lqd;
    const uint32_t i = a.log_q ? (__brev(s) >> (32 - a.log_q)) : 0u;
    const uint32_t i_next = (i + qd) & (q - 1);
    const uint32_t s_next = a.log_q ? (__brev(i_next) >> (32 - a.log_q)) : 0u;
    const uint32_t* main_l = a.main + (size_t)sl * a.main_pitch;
    const uint32_t* main_n = a.split ? main_l + a.next_off : a.main + (size_t)s_next * a.main_pitch;
    uint32_t* tile_l = regs + a.regs_words;
    uint32_t* idx = tile_l + (a.staged ? 64u * a.wp : 0u);
    uint32_t* folds = idx + 64;  // [n_parts][64][4]
    if (a.staged) {
        // only the local rows are staged: the Lair AIRs read one or two columns of the next row (nonce, is_real, ptr),
        // which stay in global memory, and a second tile would halve the waves an LDS-bound CU can hold
        if (wave == 0) idx[lane] = sl;
        __syncthreads();
#ifndef LURK_AB_NO_STAGE  // (diagnostic: what staging the tile costs -- wrong values)
        stage_rows(tile_l, a.wp, a.main, a.w, idx, 64u, a.main_pitch);
#endif
        __syncthreads();
        main_l = tile_l + lane * a.wp;
    }
    // selectors at x = g * w_Q^i (p3 TwoAdicMultiplicativeCoset::selectors_on_coset): the constraint wave needs them
    uint32_t is_first = 0, is_last = 0, is_trans = 0;
    const uint32_t piece = __builtin_amdgcn_readfirstlane((uint32_t)a.parts.piece_of[wave]);  // (VmParts: the pieces are dealt to the waves by length)
    const bool cons_wave = piece < a.n_cons_parts;
    if (cons_wave && a.sel) {
        // (functions of the domain only: a table of the context -- per row they were a power ladder and two Fermat inversions,
        // some 660 instructions beside the two to three thousand of an eval row's constraints)
        const uint32_t* sp = a.sel + 3 * (size_t)s;
        is_first = sp[0];
        is_last = sp[1];
        is_trans = sp[2];
    } else if (cons_wave) {
        const uint32_t x = bb::mul(a.g_m, bb::pow(a.wq_m, i));
        const uint32_t zh = a.zh[i & (qd - 1)];
        is_first = bb::mul(zh, bb::inv(bb::sub(x, bb::R1)));
        const uint32_t x_minus_last = bb::sub(x, a.wn_inv_m);
        is_last = bb::mul(zh, bb::inv(x_minus_last));
        is_trans = x_minus_last;
    }
    // (split: no Lair AIR reads a preprocessed column on the next row -- the host refuses one that does --, and the running sum is honest)
    airvm::Sources src{main_l, main_n, a.prep + (size_t)sl * a.prep_pitch, a.prep + (size_t)(a.split ? sl : s_next) * a.prep_pitch, a.pub, {is_first, is_last, is_trans}};
    const uint32_t* perm_l = a.perm + (size_t)sl * a.perm_pitch;
    const uint32_t* perm_n = a.perm + (size_t)(a.split ? sl : s_next) * a.perm_pitch;
    QuotientSink sink{a.alpha_pows, a.k_total, LogupAccum{a.beta_pows, a.starts}, a.batch, perm_l};
    const uint32_t* prog = a.parts.prog[piece];
    const uint32_t first = prog[airp::H_FIRST_COLUMN];  // constraint piece: its first constraint; interaction piece: its first column
    sink.col = cons_wave ? 0u : first;
    sink.last_col = a.perm_w - 1;
    sink.col_live = a.col_live;
    sink.prime(cons_wave ? first : a.n_cons + first);
    if (!cons_wave) sink.prefetch();
#if defined(LURK_AB_NO_CONS)  // diagnostics (wrong values): the interaction waves alone / the constraint waves alone
    if (!cons_wave)
#elif defined(LURK_AB_NO_INTER)
    if (cons_wave)
#endif
    Runner::run(prog, piece, src, regs + a.parts.reg_off[piece] + lane, sink);
    if (sink.acc.in_batch) sink.flush();
    // every batch column's entry of the local row was read by the piece that owns the column (QuotientSink::flush): the pieces
    // hand their sums over with their folds, so that the local row is read once (round 4: the quotient kernels fetched the
    // permutation LDE three times -- entries, local sum, next sum -- 7.3 GB against 5.4 per fib-mix step)
    uint32_t* sums = folds + a.parts.n_parts * 256u;
    if (wave != 0) {
        const ef f = sink.folded.value();
#pragma unroll
        for (int c = 0; c < 4; c++) {
            folds[(wave * 64u + lane) * 4 + c] = f.c[c];
            sums[(wave * 64u + lane) * 4 + c] = sink.sum_cols.c[c];
        }
    }
    __syncthreads();
    if (wave != 0 || !live) return;
    // running-sum constraints (sphinx eval_permutation_constraints)
    const ef cumulative_sum = a.cumsum_dev ? ef_load(a.cumsum_dev) : a.cumulative_sum;
    const ef trans_const = a.cumsum_dev ? bb::ef_scale(cumulative_sum, a.trans_scale) : a.trans_const;
    ef sum_l = sink.sum_cols;
    for (uint32_t j = 1; j < a.parts.n_parts; j++) sum_l = bb::ef_add(sum_l, ef_load(sums + (j * 64u + lane) * 4));
    const ef phi_l = ef_load(perm_l + 4 * (a.perm_w - 1));
    sink.seek(a.k_total - 3);
    sink.assert_zero_ext(bb::ef_scale(bb::ef_sub(phi_l, sum_l), is_first));
    if (a.honest_running_sum) {
        sink.assert_zero_ext(bb::ef_scale(trans_const, a.zh[i & (qd - 1)]));  // (QuotientArgs: no read of the next row)
    } else {
        ef sum_n = bb::ef_zero();
        for (uint32_t c = 0; c + 1 < a.perm_w; c++) sum_n = bb::ef_add(sum_n, ef_load(perm_n + 4 * c));
        const ef phi_n = ef_load(perm_n + 4 * (a.perm_w - 1));
        sink.assert_zero_ext(bb::ef_scale(bb::ef_sub(bb::ef_sub(phi_n, phi_l), sum_n), is_trans));
    }
    sink.assert_zero_ext(bb::ef_scale(bb::ef_sub(phi_l, cumulative_sum), is_last));
    ef folded = sink.folded.value();
    for (uint32_t j = 1; j < a.parts.n_parts; j++) folded = bb::ef_add(folded, ef_load(folds + (j * 64u + lane) * 4));
    const ef quot = bb::ef_scale(folded, a.zh_inv[i & (qd - 1)]);
    const uint32_t chunk = i & (qd - 1), r = i >> lqd;
    uint4* dst = reinterpret_cast<uint4*>(a.out + ((size_t)chunk * ((size_t)1 << a.log_n) + r) * 4);
    if (a.split) dst = reinterpret_cast<uint4*>(a.out + (size_t)(a.log_rows ? (__brev(sl) >> (32 - a.log_rows)) : 0u) * 4);
    *dst = make_uint4(quot.c[0], quot.c[1], quot.c[2], quot.c[3]);
}

// the library's runner: the interpreter
struct InterpreterRunner {
    template <class Sink>
    static __device__ __forceinline__ void run(const uint32_t* prog, uint32_t /*wave*/, const airvm::Sources& src, uint32_t* regs, Sink& sink) {
        airvm::run(prog, src, regs, 64u, sink);
    }
};

}  // namespace lurkhip
