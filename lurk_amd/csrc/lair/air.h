// Symbolic AIR of the Lair chips and its compilation to a register program for the device VM.
//
// The reference states its constraints as Rust code generic over p3's `AirBuilder` plus Lurk's
// `LookupBuilder` (/root/reference/src/air/builder.rs:34-133); the prover (sphinx, third-party) runs that
// code once with a symbolic builder to collect the interactions and then again, per row of the quotient
// domain, with a folding builder.  Here the same walk is done once per chip on the host by `Builder`
// below (the twin of the symbolic builder); the result -- every asserted polynomial, in assertion order,
// and every send/receive -- is lowered to a straight-line program (air_program.h) that the GPU evaluates
// per row for (a) the debug checker, (b) the permutation trace and (c) the quotient.
//
//   Func chips     /root/reference/src/lair/air.rs:158-552
//   MemChip        /root/reference/src/lair/memory.rs:71-109
//   BytesChip      /root/reference/src/gadgets/bytes/trace.rs:117-143
//   Entrypoint     /root/reference/src/lair/lair_chip.rs:166-191
//   provide/require /root/reference/src/air/builder.rs:42-104, relations /root/reference/src/lair/relations.rs:6-59
#pragma once
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../air_program.h"
#include "lair.h"

namespace lair {

enum NodeKind : uint8_t {
    N_CONST,      // a = canonical value
    N_MAIN,       // a = column (local row)
    N_MAIN_NEXT,  // a = column (next row)
    N_PREP,
    N_PREP_NEXT,
    N_PUBLIC,     // a = index
    N_IS_FIRST,
    N_IS_LAST,
    N_IS_TRANS,
    N_ADD,
    N_SUB,
    N_MUL
};

struct Node {
    NodeKind kind;
    uint32_t a, b;
    uint8_t degree;  // p3 SymbolicExpression::degree_multiple
};

using E = uint32_t;  // node id

struct Interaction {  // sphinx AirInteraction {values, multiplicity, kind}
    bool is_send;
    uint32_t kind;  // InteractionKind as usize (Memory = 1 [UPSTREAM-RECALL])
    std::vector<E> values;
    E mult;
};

constexpr uint32_t INTERACTION_KIND_MEMORY = 1;

struct ChipAir {
    std::string name;
    uint32_t width = 0, prep_width = 0, num_public = 0;
    std::vector<Node> nodes;
    std::vector<E> constraints;           // asserted-zero expressions in assertion order
    std::vector<Interaction> sends;       // declaration order
    std::vector<Interaction> receives;
    uint32_t max_constraint_degree() const;
    // sphinx Chip::new: degree >= 3 once there are interactions; log2_ceil(degree - 1)
    uint32_t log_quotient_degree() const;
    uint32_t num_interactions() const { return (uint32_t)(sends.size() + receives.size()); }
    // number of extension-field columns of the permutation trace: ceil(#interactions / batch) + 1
    uint32_t permutation_width() const;
};

// Hash-consing expression builder with the p3 AirBuilder vocabulary.
class Builder {
   public:
    explicit Builder(ChipAir& air) : air_(air) {}
    E cst(uint32_t canonical);
    E zero() { return cst(0); }
    E one() { return cst(1); }
    E main(uint32_t col) { return leaf(N_MAIN, col, 1); }
    E main_next(uint32_t col) { return leaf(N_MAIN_NEXT, col, 1); }
    E prep(uint32_t col) { return leaf(N_PREP, col, 1); }
    E prep_next(uint32_t col) { return leaf(N_PREP_NEXT, col, 1); }
    E pub(uint32_t i) { return leaf(N_PUBLIC, i, 0); }
    E is_first_row() { return leaf(N_IS_FIRST, 0, 1); }
    E is_last_row() { return leaf(N_IS_LAST, 0, 1); }
    E is_transition() { return leaf(N_IS_TRANS, 0, 0); }
    E add(E a, E b);
    E sub(E a, E b);
    E mul(E a, E b);
    E neg(E a) { return sub(zero(), a); }
    bool is_const(E e, uint32_t* v = nullptr) const;

    // filtered assertions: `cond` is the product of the enclosing when(..) conditions (0xffffffff = none)
    static constexpr E NONE = 0xffffffffu;
    void assert_zero(E x, E cond = NONE);
    void assert_eq(E a, E b, E cond = NONE) { assert_zero(sub(a, b), cond); }
    void assert_one(E x, E cond = NONE) { assert_zero(sub(x, one()), cond); }
    void assert_bool(E x, E cond = NONE) { assert_zero(mul(x, sub(x, one())), cond); }
    E both(E c1, E c2) { return c1 == NONE ? c2 : (c2 == NONE ? c1 : mul(c1, c2)); }

    // LookupBuilder (air/builder.rs:34-133)
    void receive(const std::vector<E>& values, E is_real);
    void send(const std::vector<E>& values, E is_real);
    void provide(const std::vector<E>& relation, E last_nonce, E last_count, E is_real);
    void require(const std::vector<E>& relation, E nonce, E prev_nonce, E prev_count, E count_inv, E is_real);

    ChipAir& air() { return air_; }

   private:
    E leaf(NodeKind k, uint32_t a, uint8_t degree);
    E intern(NodeKind k, uint32_t a, uint32_t b, uint8_t degree);
    ChipAir& air_;
    std::map<std::tuple<uint8_t, uint32_t, uint32_t>, E> memo_;
};

ChipAir build_func_air(const Toplevel& t, const Func& f);
ChipAir build_mem_air(uint32_t len);
ChipAir build_bytes_air();
ChipAir build_entrypoint_air(uint32_t func_idx, uint32_t num_public_values);
ChipAir build_poseidon2_air(uint32_t width);  // the narrow (one row per round) Poseidon2 chip

// Lowered programs (air_program.h): the constraint program asserts every constraint in order, the
// interaction program emits sends (declaration order) then receives.
struct AirPrograms {
    std::vector<uint32_t> constraints;
    std::vector<uint32_t> interactions;
    // The interaction program cut into independent pieces at batch boundaries (batch = 2^log_quotient_degree interactions
    // per permutation column): piece j covers whole columns, its header's H_FIRST_COLUMN says where it starts.  The prover
    // kernels give every piece its own wave over one staged tile of rows, which multiplies the waves a CU can hold.
    // The constraint program cut into independent pieces of consecutive constraints (each piece recomputes the shared
    // subexpressions it needs; header word H_FIRST_COLUMN = index of its first constraint).  One piece for ordinary chips; the
    // Poseidon2 chips' 5-10 k instructions become up to 8 pieces = 8 waves per 64 rows of the quotient kernel (a 2^8-row hash
    // chip is 8 workgroups: with one constraint wave each the launch was a 600 us dependent chain).
    std::vector<std::vector<uint32_t>> constraint_parts;
    std::vector<std::vector<uint32_t>> interaction_parts;
    // the same, cut coarser, for the quotient kernel (measured: it does best with two dozen interactions per wave, the
    // permutation-trace kernel with one dozen)
    std::vector<std::vector<uint32_t>> interaction_parts_coarse;
    // The pieces are *compact* (air_program.h: OP_IVALS / OP_IVALT): constant tuple elements are left out of the programs.
    // const_terms lists them -- (interaction index, position t = 1 + index in the tuple, canonical constant) -- and
    // interaction_kinds the kind of every interaction, so the prover can build the start values alpha + kind + sum beta^t c.
    struct ConstTerm {
        uint32_t interaction, t, value;
    };
    std::vector<ConstTerm> const_terms;
    std::vector<uint32_t> interaction_kinds;
};
AirPrograms lower_air(const ChipAir& air);

}  // namespace lair
