// Per-function compiled trace generators: a Lair function's degree-resolved micro-program (lair/trace_program.h) unrolled
// along its block tree into one straight-line row function, compiled with hiprtc against trace_kernels.h.
//
// The interpreter (trace.hip: trace_row) keeps the row's variable map in per-lane scratch because the program indexes it
// dynamically, reads program words through the scalar cache per operation and branches per operation: 6 T lane-instr/s with
// 78 % of the wave cycles waiting (profiles/r02_pmc_sq_per_kernel.csv).  Here every variable index, hint / require offset and
// aux column is a literal, so the map is a set of SSA values in VGPRs, the hint loads of a block go out together and the
// match arms are plain branches.  Same TraceArgs, grid and LDS tile as k_trace_func, so the host only swaps the function.
//
// Replaces the same reference code as the interpreter: Func/Block/Ctrl/Op::populate_row, /root/reference/src/lair/trace.rs:145-418.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <stdexcept>
#include <sstream>
#include <string>
#include <vector>

#include "jit.h"
#include "lair/trace_program.h"

namespace lurkhip {

namespace {

using namespace lair;

struct Gen {
    const std::vector<uint32_t>& prog;
    std::ostringstream o;
    uint32_t n_in, n_out, n_aux;
    bool partial;
    explicit Gen(const std::vector<uint32_t>& p) : prog(p), n_in(p[TH_INPUT]), n_out(p[TH_OUTPUT]), n_aux(p[TH_AUX]), partial(p[TH_PARTIAL] != 0) {}

    static std::string m(uint32_t i) { return "v" + std::to_string(i); }
    void line(int ind, const std::string& s) { o << std::string((size_t)ind * 4, ' ') << s << "\n"; }
    // a new SSA value for variable index `sp`
    void def(int ind, uint32_t sp, const std::string& expr) { line(ind, "const uint32_t " + m(sp) + " = " + expr + ";"); }

    // the ops of the block at `pc`; sp / h / r / d are the interpreter's cursors, all static along a path of the tree
    void block(uint32_t pc, uint32_t sp, uint32_t h, uint32_t r, uint32_t d, int ind) {
        for (;;) {
            const uint32_t ins = prog.at(pc), op = ins & 0xff, flag = ins >> 8;
            auto U = [](uint32_t x) { return std::to_string(x) + "u"; };
            switch (op) {
                case T_CONST: def(ind, sp++, U(prog.at(pc + 1))); pc += 2; break;
                case T_ADD: def(ind, sp++, "bb::add(" + m(prog.at(pc + 1)) + ", " + m(prog.at(pc + 2)) + ")"); pc += 3; break;
                case T_SUB: def(ind, sp++, "bb::sub(" + m(prog.at(pc + 1)) + ", " + m(prog.at(pc + 2)) + ")"); pc += 3; break;
                case T_MUL:
                    def(ind, sp, "bb::mul_s(" + m(prog.at(pc + 1)) + ", " + m(prog.at(pc + 2)) + ")");
                    if (flag) line(ind, "w.push_aux(" + m(sp) + ");");
                    sp++;
                    pc += 3;
                    break;
                case T_INV:
                    def(ind, sp, "bb::inv(" + m(prog.at(pc + 1)) + ")");
                    if (flag) line(ind, "w.push_aux(" + m(sp) + ");");
                    sp++;
                    pc += 2;
                    break;
                case T_NOT: {
                    const std::string x = m(prog.at(pc + 1));
                    def(ind, sp, x + " ? 0u : bb::R1");
                    if (flag) {
                        line(ind, "w.push_aux(" + x + " ? bb::inv(" + x + ") : 0u);");
                        line(ind, "w.push_aux(" + m(sp) + ");");
                    }
                    sp++;
                    pc += 2;
                    break;
                }
                case T_ASSERT_NE: {
                    // inverse of the first non-zero difference, zeros elsewhere (trace.rs:218-233)
                    const uint32_t n = flag;
                    line(ind, "{");
                    line(ind + 1, "bool found = false;");
                    for (uint32_t i = 0; i < n; i++) {
                        line(ind + 1, "{ const uint32_t diff = bb::sub(" + m(prog.at(pc + 1 + i)) + ", " + m(prog.at(pc + 1 + n + i)) + ");");
                        line(ind + 1, "  const bool hit = !found && diff != 0; w.push_aux(hit ? bb::inv(diff) : 0u); found = found || hit; }");
                    }
                    line(ind, "}");
                    pc += 1 + 2 * n;
                    break;
                }
                case T_CONTAINS: {
                    const uint32_t n = flag;
                    const std::string b = m(prog.at(pc + 1));
                    line(ind, "{");
                    line(ind + 1, "uint32_t acc = bb::sub(" + m(prog.at(pc + 2)) + ", " + b + ");");
                    for (uint32_t i = 1; i < n; i++) {
                        line(ind + 1, "acc = bb::mul_s(acc, bb::sub(" + m(prog.at(pc + 2 + i)) + ", " + b + "));");
                        line(ind + 1, "w.push_aux(acc);");
                    }
                    line(ind, "}");
                    pc += 2 + n;
                    break;
                }
                case T_CALL: {
                    const uint32_t n = prog.at(pc + 1);
                    hinted(ind, n, sp, h);
                    line(ind, "push_require(w, reqs + " + U(2 * r++) + ");");
                    if (flag) {
                        // dependency provenance (trace.rs:235-254): callee depth bytes, DepthLessThan, one depth require
                        line(ind, "{ const uint32_t cd = hints[" + U(h++) + "];");
                        line(ind, "  for (int i = 0; i < 4; i++) w.push_aux_int((cd >> (8 * i)) & 0xff);");
                        line(ind, "  push_depth_less_than(w, cd, own_depth); }");
                        line(ind, "push_require(w, dreqs + " + U(2 * d++) + ");");
                    }
                    pc += 2;
                    break;
                }
                case T_STORE:
                    hinted(ind, 1, sp, h);
                    line(ind, "push_require(w, reqs + " + U(2 * r++) + ");");
                    pc += 1;
                    break;
                case T_LOAD:
                    hinted(ind, prog.at(pc + 1), sp, h);
                    line(ind, "push_require(w, reqs + " + U(2 * r++) + ");");
                    pc += 2;
                    break;
                case T_EXTERN: {
                    const uint32_t kind = prog.at(pc + 1), nin = prog.at(pc + 2), wit = prog.at(pc + 3), nreq = prog.at(pc + 4), nret = prog.at(pc + 5);
                    if (nin > 48 || nret > 48) throw std::runtime_error("extern chip with more than 48 inputs or outputs");
                    std::string ins;
                    for (uint32_t i = 0; i < nin; i++) ins += (i ? ", " : "") + m(prog.at(pc + 6 + i));
                    line(ind, "uint32_t o" + std::to_string(sp) + "[" + std::to_string(std::max(nret, 1u)) + "];");
                    line(ind, "{ const uint32_t in_[" + std::to_string(std::max(nin, 1u)) + "] = {" + ins + "};");
                    line(ind, "  extern_op(w, " + U(kind) + ", in_, o" + std::to_string(sp) + ", " + U(wit) + "); }");
                    const uint32_t base = sp;
                    for (uint32_t i = 0; i < nret; i++) def(ind, sp++, "o" + std::to_string(base) + "[" + std::to_string(i) + "]");
                    for (uint32_t i = 0; i < nreq; i++) line(ind, "push_require(w, reqs + " + U(2 * r++) + ");");
                    pc += 6 + nin;
                    break;
                }
                case T_RANGE_U8: {
                    const uint32_t n = prog.at(pc + 1);
                    for (uint32_t i = 0; i < n; i++) line(ind, "push_require(w, reqs + " + U(2 * r++) + ");");
                    pc += 2;
                    break;
                }
                case T_RETURN:
                    line(ind, "w.put_int(" + U(1 + n_in + n_out + n_aux + prog.at(pc + 1)) + ", 1u);");
                    line(ind, "return;");
                    return;
                case T_CHOOSE: {
                    const uint32_t n = prog.at(pc + 2), dflt = prog.at(pc + 3);
                    const std::string v = m(prog.at(pc + 1));
                    // arms by target block, in first-key order (several keys may share an arm)
                    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> arms;
                    for (uint32_t i = 0; i < n; i++) {
                        const uint32_t key = prog.at(pc + 4 + 2 * i), tgt = prog.at(pc + 4 + 2 * i + 1);
                        size_t a = 0;
                        while (a < arms.size() && arms[a].first != tgt) a++;
                        if (a == arms.size()) arms.push_back({tgt, {}});
                        arms[a].second.push_back(key);
                    }
                    bool first = true;
                    for (const auto& arm : arms) {
                        std::string cond;
                        for (size_t k = 0; k < arm.second.size(); k++) cond += (k ? " || " : "") + v + " == " + U(arm.second[k]);
                        line(ind, std::string(first ? "if (" : "} else if (") + cond + ") {");
                        block(arm.first, sp, h, r, d, ind + 1);
                        first = false;
                    }
                    tail(ind, first, dflt, sp, h, r, d);
                    return;
                }
                case T_CHOOSE_MANY: {
                    const uint32_t nv = prog.at(pc + 1), n = prog.at(pc + 2), dflt = prog.at(pc + 3);
                    const uint32_t vars = pc + 4, table = vars + nv;
                    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> arms;  // target -> case indices
                    for (uint32_t i = 0; i < n; i++) {
                        const uint32_t tgt = prog.at(table + i * (nv + 1) + nv);
                        size_t a = 0;
                        while (a < arms.size() && arms[a].first != tgt) a++;
                        if (a == arms.size()) arms.push_back({tgt, {}});
                        arms[a].second.push_back(i);
                    }
                    bool first = true;
                    for (const auto& arm : arms) {
                        std::string cond;
                        for (size_t c = 0; c < arm.second.size(); c++) {
                            std::string one;
                            for (uint32_t k = 0; k < nv; k++)
                                one += (k ? " && " : "") + m(prog.at(vars + k)) + " == " + U(prog.at(table + arm.second[c] * (nv + 1) + k));
                            if (nv == 0) one = "true";
                            cond += (c ? " || " : "") + ("(" + one + ")");
                        }
                        line(ind, std::string(first ? "if (" : "} else if (") + cond + ") {");
                        block(arm.first, sp, h, r, d, ind + 1);
                        first = false;
                    }
                    tail(ind, first, dflt, sp, h, r, d);
                    return;
                }
                default:
                    throw std::runtime_error("corrupt trace program (op " + std::to_string(op) + ")");
            }
        }
    }
    // the default arm (or nothing: a value outside the cases leaves the rest of the row zero, as the interpreter does)
    void tail(int ind, bool no_arms, uint32_t dflt, uint32_t sp, uint32_t h, uint32_t r, uint32_t d) {
        if (no_arms) {
            if (dflt) block(dflt, sp, h, r, d, ind);
            else line(ind, "return;");
            return;
        }
        if (dflt) {
            line(ind, "} else {");
            block(dflt, sp, h, r, d, ind + 1);
        }
        line(ind, "}");
        line(ind, "return;");
    }
    // n values read from the row's hints: variables + aux columns (Call / PreImg outputs, Store pointer, Load values)
    void hinted(int ind, uint32_t n, uint32_t& sp, uint32_t& h) {
        for (uint32_t i = 0; i < n; i++) {
            line(ind, "const uint32_t h" + std::to_string(sp) + " = hints[" + std::to_string(h++) + "u];");
            def(ind, sp, "bb::to_monty(h" + std::to_string(sp) + ")");
            line(ind, "w.push_aux_int(h" + std::to_string(sp) + ");");
            sp++;
        }
    }

    std::string source() {
        // (run-time compiled code takes the contract-clean multiply-add: jit.cpp, babybear.h LURK_MAD_CARRY_DECLARED 3)
        o << "#ifndef LURK_MAD_CARRY_DECLARED\n#define LURK_MAD_CARRY_DECLARED 3\n#endif\n"
          << "#define LURKHIP_COMPILED_TRACE 1\n#include \"trace_kernels.h\"\nnamespace lurkhip_trace {\n"
          << "__device__ __forceinline__ void jit_row(const TraceArgs& a, const uint32_t row_i, RowWriter& w) {\n";
        // the prologue of trace_row (trace.rs:82-131): nonce, outputs, provide record, depth bytes + 2 requires, inputs
        line(1, "w.put_int(0, a.nonce_start + row_i);");
        line(1, "if (row_i >= a.n_real) return;");
        line(1, "const RowMeta rm = a.meta[row_i];");
        line(1, "const uint32_t* __restrict__ hints = a.stream + rm.offset;");
        line(1, "const uint32_t* __restrict__ reqs = hints + rm.n_hints;");
        line(1, "const uint32_t* __restrict__ dreqs = reqs + 2 * rm.n_requires;");
        line(1, std::string("const uint32_t own_depth = ") + (partial ? "a.depths[row_i]" : "0u") + ";");
        line(1, "(void)dreqs; (void)own_depth; (void)reqs; (void)hints;");
        for (uint32_t i = 0; i < n_out; i++)
            line(1, "w.put_int(" + std::to_string(1 + n_in + i) + "u, a.outputs[(size_t)row_i * " + std::to_string(n_out) + "u + " + std::to_string(i) + "u]);");
        line(1, "w.push_aux_int(a.provides[2 * row_i]);");
        line(1, "w.push_aux_int(a.provides[2 * row_i + 1]);");
        uint32_t d = 0;
        if (partial) {
            line(1, "for (int i = 0; i < 4; i++) w.push_aux_int((own_depth >> (8 * i)) & 0xff);");
            for (int i = 0; i < 2; i++) line(1, "push_require(w, dreqs + " + std::to_string(2 * d++) + "u);");
        }
        for (uint32_t i = 0; i < n_in; i++) {
            line(1, "const uint32_t a" + std::to_string(i) + " = a.args[(size_t)row_i * " + std::to_string(n_in) + "u + " + std::to_string(i) + "u];");
            line(1, "w.put_int(" + std::to_string(1 + i) + "u, a" + std::to_string(i) + ");");
            def(1, i, "bb::to_monty(a" + std::to_string(i) + ")");
        }
        block(prog.at(TH_ENTRY), n_in, 0, 0, d, 1);
        // The row function reaches the kernel body through a functor whose call operator is always_inline, not a lambda: a lambda's
        // operator() is inlined only when the inliner's cost model says so, and for a function with dozens of match arms (the
        // shape-matched eval_builtin_expr: 59 arms, 230 KB of code) it does not -- the row writer then travels as a reference to
        // private memory and the LDS tile as a generic pointer through a real call, and the kernel faults (round 5:
        // HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION on the first launch; every smaller function was inlined and ran).
        o << "}\nstruct JitRow {\n"
          << "    __device__ __forceinline__ void operator()(const TraceArgs& a, uint32_t row_i, RowWriter& w) const { jit_row(a, row_i, w); }\n"
          << "};\n}  // namespace lurkhip_trace\n"
          << "extern \"C\" __global__ __launch_bounds__(64) void jit_trace_staged(lurkhip_trace::TraceArgs a) {\n"
          << "    lurkhip_trace::trace_kernel_body<true>(a, lurkhip_trace::JitRow{});\n}\n"
          << "extern \"C\" __global__ __launch_bounds__(64) void jit_trace_flat(lurkhip_trace::TraceArgs a) {\n"
          << "    lurkhip_trace::trace_kernel_body<false>(a, lurkhip_trace::JitRow{});\n}\n";
        return o.str();
    }
};

// loaded kernels by (device, program hash): a program is compiled and loaded once per process and device
std::mutex g_mu;
std::map<std::pair<int, uint64_t>, TraceJitKernels> g_loaded;

}  // namespace

std::string trace_jit_source(const std::vector<uint32_t>& prog) {
    if (prog.size() < lair::TH_WORDS || prog[lair::TH_MAGIC] != lair::TRACE_PROGRAM_MAGIC) throw std::runtime_error("bad trace program");
    Gen g(prog);
    return g.source();
}

size_t trace_jit_compile_only(const std::vector<uint32_t>& prog, std::string* log) {
    std::vector<char> code;
    try {
        return jit_get_code(trace_jit_source(prog), &code, log) ? code.size() : 0;
    } catch (const std::exception& e) {
        if (log) *log = e.what();
        return 0;
    }
}

bool trace_jit_compile(int device, const std::vector<uint32_t>& prog, std::string* log) {
    const uint64_t h = lair::trace_program_hash(prog.data(), prog.size());
    {
        std::lock_guard<std::mutex> g(g_mu);
        if (g_loaded.count({device, h})) return true;
    }
    std::vector<char> code;
    try {
        if (!jit_get_code(trace_jit_source(prog), &code, log)) return false;
    } catch (const std::exception& e) {
        if (log) *log = e.what();
        return false;
    }
    TraceJitKernels k;
    if (hipModuleLoadData(&k.module, code.data()) != hipSuccess || hipModuleGetFunction(&k.staged, k.module, "jit_trace_staged") != hipSuccess ||
        hipModuleGetFunction(&k.flat, k.module, "jit_trace_flat") != hipSuccess) {
        if (k.module) (void)hipModuleUnload(k.module);
        if (log) *log = "loading the compiled trace module failed";
        return false;
    }
    // the LDS tile of a wide function exceeds the default 64 KiB cap? (never: staged tiles are capped at 64 KiB by the launcher)
    std::lock_guard<std::mutex> g(g_mu);
    g_loaded[{device, h}] = k;
    return true;
}

TraceJitKernels trace_jit_lookup(int device, uint64_t hash) {
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_loaded.find({device, hash});
    return it == g_loaded.end() ? TraceJitKernels{} : it->second;
}

}  // namespace lurkhip
