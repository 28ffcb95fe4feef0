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

}  // namespace lurkhip
