// Run-time compilation of a chip's AIR program pieces.  See jit.h and DESIGN.md (AIR programs).
//
// The interpreter (air_vm.h) pays scalar decode, branches and LDS register traffic per instruction; for a chip with millions
// of rows it is worth compiling its programs once: every piece becomes one straight-line device function over the same sinks
// (stark_kernels.h), every interpreter register an SSA value the compiler keeps in a VGPR.  The generated kernels take the
// arguments, grid and LDS layout of k_perm_rows / k_quotient, so the host side only swaps the function it launches.
#include "jit.h"

#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <sstream>
#include <vector>

#include "air_program.h"
#include "jit_headers.inc"

namespace lurkhip {

namespace {

// one program (classic 2-word encoding, compact interaction ops included) -> a function template over the sink
// `batch` > 0: the program is an interaction piece that starts at a batch boundary; every IEND then names its position in the
// batch and whether it closes it (sink.iend_at<POS, LAST>), so the sink's batch bookkeeping folds away at compile time
// `skip_dead`: (permutation pieces) a batch's sink operations run only when one of its multiplicities is non-zero on some row of
// the wave (PermSink::batch_live); the arithmetic that feeds them stays outside the branch (SSA values later batches may share)
void emit_function(std::ostringstream& o, const std::vector<uint32_t>& prog, const std::string& name, uint32_t batch, bool skip_dead = false) {
    const uint32_t n = prog[airp::H_N_INSTR];
    const uint32_t* code = prog.data() + prog[airp::H_CODE_OFF];
    const uint32_t* consts = prog.data() + prog[airp::H_CONST_OFF];
    std::map<uint32_t, std::string> reg;  // interpreter register -> the SSA value that currently lives in it
    auto operand = [&](uint32_t op) -> std::string {
        const uint32_t idx = op & airp::SRC_MASK;
        switch (op >> airp::SRC_SHIFT) {
            case airp::S_REG: return reg.at(idx);
            case airp::S_MAIN: return "s.main_l[" + std::to_string(idx) + "]";
            case airp::S_MAIN_NEXT: return "s.main_n[" + std::to_string(idx) + "]";
            case airp::S_PREP: return "s.prep_l[" + std::to_string(idx) + "]";
            case airp::S_PREP_NEXT: return "s.prep_n[" + std::to_string(idx) + "]";
            case airp::S_CONST: return std::to_string(consts[idx]) + "u";
            case airp::S_PUBLIC: return "s.pub[" + std::to_string(idx) + "]";
            default: return "s.sel[" + std::to_string(idx) + "]";
        }
    };
    o << "template <class Sink> __device__ __forceinline__ void " << name << "(const airvm::Sources& s, Sink& sink) {\n";
    uint32_t seen_ends = 0;
    std::ostringstream pending;           // skip_dead: the sink operations of the batch being assembled
    std::vector<std::string> batch_mults;
    std::ostringstream& body = o;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t w0 = code[2 * i], w1 = code[2 * i + 1];
        const uint32_t op = w0 & 0xffu, dst = w0 >> 8, a = w1 & 0xffffu, b = w1 >> 16;
        const std::string t = "t" + std::to_string(i);
        const bool sink_op = op == airp::OP_IBEGIN || op == airp::OP_IVAL || op == airp::OP_IEND || op == airp::OP_IVALS || op == airp::OP_IVALT;
        std::ostringstream& o = (skip_dead && batch && sink_op) ? pending : body;
        switch (op) {
            case airp::OP_ADD: o << "    const uint32_t " << t << " = bb::add(" << operand(a) << ", " << operand(b) << ");\n"; reg[dst] = t; break;
            case airp::OP_SUB: o << "    const uint32_t " << t << " = bb::sub(" << operand(a) << ", " << operand(b) << ");\n"; reg[dst] = t; break;
            case airp::OP_MUL: o << "    const uint32_t " << t << " = bb::mul_s(" << operand(a) << ", " << operand(b) << ");\n"; reg[dst] = t; break;
            case airp::OP_ASSERT: o << "    sink.assert_zero(" << operand(a) << ");\n"; break;
            case airp::OP_IBEGIN: o << "    sink.ibegin(" << dst << "u, " << (a ? "true" : "false") << ", " << b << "u);\n"; break;
            case airp::OP_IVAL: o << "    sink.ival(" << operand(a) << ");\n"; break;
            case airp::OP_IEND:
                if (batch) {
                    // the last batch of the chip may be partial: the kernel body flushes it after the piece
                    const uint32_t pos = seen_ends % batch;
                    const bool last = pos + 1 == batch;
                    o << "    sink.template iend_at<" << (pos < 2 ? pos : 2u) << ", " << (last ? "true" : "false") << ">(" << operand(a) << ");\n";
                    seen_ends++;
                    if (skip_dead) {
                        batch_mults.push_back(operand(a));
                        if (last) {
                            body << "    if (sink.batch_live(";
                            for (size_t k = 0; k < batch_mults.size(); k++) body << (k ? " | " : "") << batch_mults[k];
                            body << ")) {\n" << pending.str() << "    } else {\n        sink.skip_batch();\n    }\n";
                            pending.str("");
                            batch_mults.clear();
                        }
                    }
                } else {
                    o << "    sink.iend(" << operand(a) << ");\n";
                }
                break;
            case airp::OP_IVALS: o << "    sink.ival_run(s.main_l + " << a << ", " << b << "u, " << dst << "u);\n"; break;
            case airp::OP_IVALT: o << "    sink.ival_at(" << operand(a) << ", " << dst << "u);\n"; break;
            default: break;  // OP_NOP padding
        }
    }
    o << pending.str();  // a partial last batch (flushed by the kernel body) runs unconditionally
    o << "}\n";
}

// A constraint piece with dead branches skipped (round 5).  Almost every constraint of a Lair function is `sel * x`, sel the
// selector sum of the block the constraint belongs to, and a row takes one branch: on the quotient domain the selectors of the
// branches a shard never takes are zero polynomials, and sel * x is zero wherever sel is, whatever x is.  Consecutive ASSERTs whose
// root products share a factor form a group: the group's factor is computed first, then -- wave-uniform -- either the group's
// arithmetic and its folds run, or only the constraint index moves on (QuotientSink::cond_live / skip_asserts: the test looks at
// the values of the wave's own 64 points, it assumes nothing about the trace).  Values a later group reads are computed outside the
// branch.  (eval_builtin_expr: 60 return selectors, 5 of them taken by a `(fib N)` run.)
void emit_constraint_function(std::ostringstream& o, const std::vector<uint32_t>& prog, const std::string& name) {
    const uint32_t n = prog[airp::H_N_INSTR];
    const uint32_t* code = prog.data() + prog[airp::H_CODE_OFF];
    const uint32_t* consts = prog.data() + prog[airp::H_CONST_OFF];
    struct Ins {
        uint32_t op = 0;
        std::string a, b;      // operand texts (SSA names or leaves)
        int da = -1, db = -1;  // defining instruction of an SSA operand
        int last_use = -1;
    };
    std::vector<Ins> ins(n);
    std::map<uint32_t, int> reg;  // interpreter register -> defining instruction
    auto leaf = [&](uint32_t op) -> std::string {
        const uint32_t idx = op & airp::SRC_MASK;
        switch (op >> airp::SRC_SHIFT) {
            case airp::S_MAIN: return "s.main_l[" + std::to_string(idx) + "]";
            case airp::S_MAIN_NEXT: return "s.main_n[" + std::to_string(idx) + "]";
            case airp::S_PREP: return "s.prep_l[" + std::to_string(idx) + "]";
            case airp::S_PREP_NEXT: return "s.prep_n[" + std::to_string(idx) + "]";
            case airp::S_CONST: return std::to_string(consts[idx]) + "u";
            case airp::S_PUBLIC: return "s.pub[" + std::to_string(idx) + "]";
            default: return "s.sel[" + std::to_string(idx) + "]";
        }
    };
    auto resolve = [&](uint32_t op, std::string* text, int* def, int user) {
        if ((op >> airp::SRC_SHIFT) == airp::S_REG) {
            *def = reg.at(op & airp::SRC_MASK);
            *text = "t" + std::to_string(*def);
            ins[(size_t)*def].last_use = user;
        } else {
            *text = leaf(op);
        }
    };
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t w0 = code[2 * i], w1 = code[2 * i + 1];
        Ins& x = ins[i];
        x.op = w0 & 0xffu;
        const uint32_t dst = w0 >> 8, a = w1 & 0xffffu, b = w1 >> 16;
        if (x.op == airp::OP_ADD || x.op == airp::OP_SUB || x.op == airp::OP_MUL) {
            resolve(a, &x.a, &x.da, (int)i);
            resolve(b, &x.b, &x.db, (int)i);
            reg[dst] = (int)i;
        } else if (x.op == airp::OP_ASSERT) {
            resolve(a, &x.a, &x.da, (int)i);
        }
    }
    auto emit_ins = [&](uint32_t i, const char* ind) {
        const Ins& x = ins[i];
        const std::string t = "t" + std::to_string(i);
        switch (x.op) {
            case airp::OP_ADD: o << ind << "const uint32_t " << t << " = bb::add(" << x.a << ", " << x.b << ");\n"; break;
            case airp::OP_SUB: o << ind << "const uint32_t " << t << " = bb::sub(" << x.a << ", " << x.b << ");\n"; break;
            case airp::OP_MUL: o << ind << "const uint32_t " << t << " = bb::mul_s(" << x.a << ", " << x.b << ");\n"; break;
            case airp::OP_ASSERT: o << ind << "sink.assert_zero(" << x.a << ");\n"; break;
            default: break;
        }
    };
    // the factors of an ASSERT's root product ({} for a constraint that is not a product: it is evaluated unconditionally)
    auto factors = [&](uint32_t i) -> std::vector<std::pair<std::string, int>> {
        const Ins& x = ins[i];
        if (x.da < 0 || ins[(size_t)x.da].op != airp::OP_MUL) return {};
        const Ins& m = ins[(size_t)x.da];
        return {{m.a, m.da}, {m.b, m.db}};
    };
    o << "template <class Sink> __device__ __forceinline__ void " << name << "(const airvm::Sources& s, Sink& sink) {\n";
    uint32_t lo = 0;  // first instruction not emitted yet
    while (lo < n) {
        // the next group: from lo to the last of a run of consecutive ASSERTs (with the arithmetic between them) that share a factor
        uint32_t first_assert = lo;
        while (first_assert < n && ins[first_assert].op != airp::OP_ASSERT) first_assert++;
        if (first_assert == n) {  // trailing arithmetic / padding
            for (uint32_t i = lo; i < n; i++) emit_ins(i, "    ");
            break;
        }
        std::vector<std::pair<std::string, int>> common = factors(first_assert);
        uint32_t hi = first_assert, count = 1;
        for (uint32_t i = first_assert + 1; i < n && !common.empty(); i++) {
            if (ins[i].op != airp::OP_ASSERT) continue;
            std::vector<std::pair<std::string, int>> keep;
            for (const auto& f : factors(i))
                for (const auto& c : common)
                    if (c.first == f.first) keep.push_back(c);
            if (keep.empty()) break;
            common = keep;
            hi = i;
            count++;
        }
        // a factor that is a literal constant says nothing
        while (!common.empty() && !common[0].first.empty() && isdigit((unsigned char)common[0].first[0])) common.erase(common.begin());
        if (common.empty() || count < 2) {  // nothing to share (or one constraint: the test would cost what it saves)
            for (uint32_t i = lo; i <= first_assert; i++) emit_ins(i, "    ");
            lo = first_assert + 1;
            continue;
        }
        const std::pair<std::string, int> cond = common[0];
        // hoisted: the factor's own arithmetic and everything a later group reads, with what those read inside the span
        std::vector<char> hoist(hi + 1, 0);
        std::vector<int> work;
        if (cond.second >= (int)lo) work.push_back(cond.second);
        for (uint32_t i = lo; i <= hi; i++)
            if (ins[i].op != airp::OP_ASSERT && ins[i].last_use > (int)hi) work.push_back((int)i);
        while (!work.empty()) {
            const int i = work.back();
            work.pop_back();
            if (i < (int)lo || hoist[(size_t)i]) continue;
            hoist[(size_t)i] = 1;
            if (ins[(size_t)i].da >= (int)lo) work.push_back(ins[(size_t)i].da);
            if (ins[(size_t)i].db >= (int)lo) work.push_back(ins[(size_t)i].db);
        }
        for (uint32_t i = lo; i <= hi; i++)
            if (hoist[i]) emit_ins(i, "    ");
        o << "    if (sink.cond_live(" << cond.first << ")) {\n";
        for (uint32_t i = lo; i <= hi; i++)
            if (!hoist[i]) emit_ins(i, "        ");
        o << "    } else {\n        sink.skip_asserts(" << count << "u);\n    }\n";
        lo = hi + 1;
    }
    o << "}\n";
}

void emit_runner(std::ostringstream& o, const std::string& name, const std::vector<std::string>& funcs) {
    o << "struct " << name << " {\n    template <class Sink>\n"
      << "    static __device__ __forceinline__ void run(const uint32_t*, uint32_t wave, const airvm::Sources& src, uint32_t*, Sink& sink) {\n"
      << "        switch (wave) {\n";
    for (size_t j = 0; j < funcs.size(); j++) o << "            case " << j << ": " << funcs[j] << "(src, sink); break;\n";
    o << "            default: break;\n        }\n    }\n};\n";
}

}  // namespace

std::string jit_source(const lair::AirPrograms& prog, uint32_t batch) {
    std::ostringstream o;
    // LURKHIP_JIT_DEFINES="NAME=VALUE,NAME=VALUE" (A/B hook): macros of the embedded headers for the compiled kernels (part of the
    // source, hence of the code cache's key)
    if (const char* defs = getenv("LURKHIP_JIT_DEFINES")) {
        std::string d(defs);
        size_t at = 0;
        while (at < d.size()) {
            size_t end = d.find(',', at);
            if (end == std::string::npos) end = d.size();
            const std::string item = d.substr(at, end - at);
            const size_t eq = item.find('=');
            if (eq != std::string::npos && eq > 0) o << "#define " << item.substr(0, eq) << " " << item.substr(eq + 1) << "\n";
            at = end + 1;
        }
    }
    // Round 6: code compiled at run time, on the target box, by whatever hiprtc is there, takes the CONTRACT-CLEAN spelling of the
    // signed multiply-add (babybear.h: LURK_MAD_CARRY_DECLARED 3, an explicit `vcc` clobber) instead of the library's default, which
    // overwrites an inline-assembly input operand and is only held by an A/B build of the ahead-of-time compiled library
    // (tests/test_mad_ab_gpu.py).  Measured on the fib-mix step, alternating: quotient_all 4.29 / 4.28 ms with the trick, 4.38 / 4.39
    // declared; permutation 1.96 either way; the step inside its noise (47.1 - 47.3 against 47.2 - 47.5 ms).  LURKHIP_JIT_DEFINES
    // can still set it back to 0 for an A/B.
    o << "#ifndef LURK_MAD_CARRY_DECLARED\n#define LURK_MAD_CARRY_DECLARED 3\n#endif\n";
    o << "#define LURKHIP_COMPILED_AIR 1\n#include \"stark_kernels.h\"\nnamespace lurkhip {\n";
    std::vector<std::string> perm, quot;
    for (size_t j = 0; j < prog.interaction_parts.size(); j++) {
        perm.push_back("perm_piece" + std::to_string(j));
        emit_function(o, prog.interaction_parts[j], perm.back(), batch, getenv("LURKHIP_PERM_SKIP_DEAD") == nullptr || atoi(getenv("LURKHIP_PERM_SKIP_DEAD")) != 0);
    }
    for (size_t j = 0; j < prog.constraint_parts.size(); j++) {
        quot.push_back("quot_cons" + std::to_string(j));
        // (opt-in: measured 4.30 -> 4.21 ms for quotient_all on the fib-mix step -- with the constraint pieces cut to 256 instructions
        // they are no longer what a workgroup waits for -- and the stand-in's constraints are not dialled to the real functions'
        // share of dead ones: tools/measure_constraint_sparsity.py, 301 of eval_builtin_expr's 616 real constraints against 516)
        if (getenv("LURKHIP_CONS_SKIP_DEAD") != nullptr && atoi(getenv("LURKHIP_CONS_SKIP_DEAD")) != 0) emit_constraint_function(o, prog.constraint_parts[j], quot.back());
        else emit_function(o, prog.constraint_parts[j], quot.back(), 0);
    }
    for (size_t j = 0; j < prog.interaction_parts_coarse.size(); j++) {
        quot.push_back("quot_piece" + std::to_string(j));
        emit_function(o, prog.interaction_parts_coarse[j], quot.back(), batch, getenv("LURKHIP_QUOT_SKIP_DEAD") == nullptr || atoi(getenv("LURKHIP_QUOT_SKIP_DEAD")) != 0);
    }
    emit_runner(o, "JitPermRunner", perm);
    emit_runner(o, "JitQuotRunner", quot);
    o << "}  // namespace lurkhip\n"
      ;
    // LURKHIP_JIT_WAVES_PER_EU = n (A/B hook): ask the compiler for n waves per SIMD (512 / n VGPRs): a quotient workgroup is up to 14
    // waves, and two of them only share a CU when a wave stays below 73 registers
    const char* wpe = getenv("LURKHIP_JIT_WAVES_PER_EU");
    const std::string attr = wpe && atoi(wpe) > 0 ? "__attribute__((amdgpu_waves_per_eu(" + std::to_string(atoi(wpe)) + "))) " : "";
    o << "extern \"C\" __global__ " << attr << "void jit_perm_rows(lurkhip::PermArgs a) { lurkhip::perm_rows_body<lurkhip::JitPermRunner>(a); }\n"
      << "extern \"C\" __global__ " << attr << "void jit_quotient(lurkhip::QuotientArgs a) { lurkhip::quotient_body<lurkhip::JitQuotRunner>(a); }\n";
    return o.str();
}

namespace {
// compiled code objects by source text: a second context / machine of the same toplevel compiles nothing
std::mutex g_cache_mu;
std::atomic<unsigned> g_tmp_serial{0};  // two threads of one process may compile the same source: distinct temporary files
std::map<std::string, std::vector<char>> g_code_cache;

bool load_module(const std::vector<char>& code, JitKernels* out, std::string* log) {
    JitKernels k;
    if (hipModuleLoadData(&k.module, code.data()) != hipSuccess || hipModuleGetFunction(&k.perm_rows, k.module, "jit_perm_rows") != hipSuccess ||
        hipModuleGetFunction(&k.quotient, k.module, "jit_quotient") != hipSuccess) {
        if (k.module) (void)hipModuleUnload(k.module);
        if (log) *log = "loading the compiled module failed";
        return false;
    }
    *out = k;
    return true;
}
}  // namespace

namespace {
bool compile_source(const std::string& src, std::vector<char>* code, std::string* log) {
    hiprtcProgram p = nullptr;
    if (hiprtcCreateProgram(&p, src.c_str(), "lurkhip_air_jit.hip", kJitHeaderCount, kJitHeaderBodies, kJitHeaderNames) != HIPRTC_SUCCESS) {
        if (log) *log = "hiprtcCreateProgram failed";
        return false;
    }
    // the ROCm include directory provides <hip/hip_runtime.h> for the embedded headers
    const char* rocm = getenv("ROCM_PATH");
    const std::string inc = std::string("-I") + (rocm && *rocm ? rocm : "/opt/rocm") + "/include";
    const char* olevel = getenv("LURKHIP_JIT_OPT");
    const char* opts[] = {"--offload-arch=gfx950", olevel && *olevel ? olevel : "-O3", "-std=c++17", "-ffp-contract=off", inc.c_str()};
    const hiprtcResult r = hiprtcCompileProgram(p, 5, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        (void)hiprtcGetProgramLogSize(p, &n);
        std::string l(n, '\0');
        if (n) (void)hiprtcGetProgramLog(p, &l[0]);
        if (log) *log = std::string(hiprtcGetErrorString(r)) + ": " + l.substr(0, 4000);
        (void)hiprtcDestroyProgram(&p);
        return false;
    }
    size_t n = 0;
    (void)hiprtcGetCodeSize(p, &n);
    code->resize(n);
    (void)hiprtcGetCode(p, code->data());
    (void)hiprtcDestroyProgram(&p);
    // debugging aid: LURKHIP_JIT_DUMP=<prefix> keeps the last generated source and code object (<prefix>.hip / <prefix>.co)
    if (const char* dump = getenv("LURKHIP_JIT_DUMP")) {
        if (FILE* f = fopen((std::string(dump) + ".hip").c_str(), "w")) {
            fwrite(src.data(), 1, src.size(), f);
            fclose(f);
        }
        if (FILE* f = fopen((std::string(dump) + ".co").c_str(), "wb")) {
            fwrite(code->data(), 1, code->size(), f);
            fclose(f);
        }
    }
    return true;
}
}  // namespace

// ---- persistent code-object cache.  A machine's AIR programs are fixed by its toplevel, and compiling a Poseidon2 chip takes
// a minute: compiled code objects are kept on disk keyed by everything that determines them (generated source, the embedded
// headers, compiler options, hiprtc version).  Directory: $LURKHIP_JIT_CACHE (empty = no disk cache), else jit_cache/ next to
// liblurkhip.so -- in-tree, so a cache warmed by the build travels with the library.
namespace {
std::string cache_dir() {
    if (const char* e = getenv("LURKHIP_JIT_CACHE")) return e;
    Dl_info info;
    if (dladdr((const void*)&cache_dir, &info) && info.dli_fname) {
        std::string path = info.dli_fname;
        const size_t slash = path.rfind('/');
        return (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/jit_cache";
    }
    return "";
}
uint64_t fnv1a(const char* p, size_t n, uint64_t h) {
    for (size_t i = 0; i < n; i++) h = (h ^ (unsigned char)p[i]) * 0x100000001b3ull;
    return h;
}
std::string cache_key(const std::string& src) {
    uint64_t h1 = 0xcbf29ce484222325ull, h2 = 0x84222325cbf29ce4ull;
    auto mix = [&](const char* p, size_t n) {
        h1 = fnv1a(p, n, h1);
        h2 = fnv1a(p, n, h2 ^ 0x9e3779b97f4a7c15ull);
    };
    mix(src.data(), src.size());
    for (int i = 0; i < kJitHeaderCount; i++) mix(kJitHeaderBodies[i], strlen(kJitHeaderBodies[i]));
    int maj = 0, min = 0;
    (void)hiprtcVersion(&maj, &min);
    const char* olevel = getenv("LURKHIP_JIT_OPT");
    const std::string tag = "gfx950|" + std::string(olevel && *olevel ? olevel : "-O3") + "|" + std::to_string(maj) + "." + std::to_string(min);
    mix(tag.data(), tag.size());
    char buf[40];
    snprintf(buf, sizeof buf, "%016llx%016llx", (unsigned long long)h1, (unsigned long long)h2);
    return buf;
}
bool disk_load(const std::string& key, std::vector<char>* code) {
    const std::string dir = cache_dir();
    if (dir.empty()) return false;
    FILE* f = fopen((dir + "/" + key + ".hsaco").c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    bool ok = n > 0;
    if (ok) {
        code->resize((size_t)n);
        ok = fread(code->data(), 1, (size_t)n, f) == (size_t)n;
    }
    fclose(f);
    return ok;
}
void disk_store(const std::string& key, const std::vector<char>& code) {
    const std::string dir = cache_dir();
    if (dir.empty()) return;
    (void)mkdir(dir.c_str(), 0755);
    const std::string final_path = dir + "/" + key + ".hsaco", tmp = final_path + ".tmp" + std::to_string((long)getpid()) + "." + std::to_string(g_tmp_serial.fetch_add(1));
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;  // read-only tree: the in-process cache still serves this process
    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), final_path.c_str()) != 0) (void)remove(tmp.c_str());
}
// code object for `src`: in-process cache, then disk, then the compiler
bool get_code(const std::string& src, std::vector<char>* code, std::string* log, bool* from_cache) {
    {
        std::lock_guard<std::mutex> g(g_cache_mu);
        auto it = g_code_cache.find(src);
        if (it != g_code_cache.end()) {
            *code = it->second;
            if (from_cache) *from_cache = true;
            return true;
        }
    }
    const std::string key = cache_key(src);
    bool cached = disk_load(key, code);
    if (!cached) {
        if (!compile_source(src, code, log)) return false;
        disk_store(key, *code);
    }
    if (from_cache) *from_cache = cached;
    std::lock_guard<std::mutex> g(g_cache_mu);
    g_code_cache.emplace(src, *code);
    return true;
}
}  // namespace

bool jit_get_code(const std::string& src, std::vector<char>* code, std::string* log) { return get_code(src, code, log, nullptr); }

size_t jit_compile_only(const lair::AirPrograms& prog, uint32_t batch, std::string* log) {
    std::vector<char> code;
    return get_code(jit_source(prog, batch), &code, log, nullptr) ? code.size() : 0;
}

bool jit_compile(const lair::AirPrograms& prog, uint32_t batch, JitKernels* out, std::string* log) {
    std::vector<char> code;
    if (!get_code(jit_source(prog, batch), &code, log, nullptr)) return false;
    return load_module(code, out, log);
}

void jit_release(JitKernels* k) {
    if (k && k->module) (void)hipModuleUnload(k->module);
    if (k) *k = JitKernels{};
}

}  // namespace lurkhip
