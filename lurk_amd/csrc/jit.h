// Run-time compilation of a chip's AIR program pieces (hiprtc): straight-line device code instead of the interpreter.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "lair/air.h"

namespace lurkhip {

struct JitKernels {
    hipModule_t module = nullptr;
    hipFunction_t perm_rows = nullptr;  // same arguments, grid and LDS as k_perm_rows
    hipFunction_t quotient = nullptr;   // same as k_quotient
};

// C++ source of the two kernels for these programs (stark_kernels.h bodies with a generated runner)
std::string jit_source(const lair::AirPrograms& prog, uint32_t batch);
// compiles for gfx950 and loads the module on the current device; on failure returns false and leaves the log in *log
bool jit_compile(const lair::AirPrograms& prog, uint32_t batch, JitKernels* out, std::string* log);
void jit_release(JitKernels* k);
// compiles without loading (no device needed): code object size in bytes, 0 on failure (reason in *log)
size_t jit_compile_only(const lair::AirPrograms& prog, uint32_t batch, std::string* log);

// code object for `src`: in-process cache, then the on-disk cache, then hiprtc (no device needed)
bool jit_get_code(const std::string& src, std::vector<char>* code, std::string* log);

// ---- per-function compiled trace generators (trace_jit.cpp)
struct TraceJitKernels {
    hipModule_t module = nullptr;
    hipFunction_t staged = nullptr;  // arguments, grid and LDS of k_trace_func<CAP, true>
    hipFunction_t flat = nullptr;    // ... of k_trace_func<CAP, false>
};
std::string trace_jit_source(const std::vector<uint32_t>& prog);
size_t trace_jit_compile_only(const std::vector<uint32_t>& prog, std::string* log);  // code object bytes, 0 on failure
bool trace_jit_compile(int device, const std::vector<uint32_t>& prog, std::string* log);  // compile (or cache) + load on the current device
TraceJitKernels trace_jit_lookup(int device, uint64_t program_hash);                 // {} when the program has not been compiled

}  // namespace lurkhip
