"""Prototype (not part of the product): straight-line HIP code generated from the lowered programs of the bench's eval chip,
plugged into the kernel bodies of csrc/stark_kernels.h as a runner.  Writes gpurun_out/jit_eval.hip and compiles it with hipcc to
show compile time and register use (run from the repo root).  Measured once on MI355X with the code object loaded in place of
the interpreter kernels for that chip: quotient 5.7 -> 4.2 ms, permutation rows 3.1 -> 2.4 ms per step; 30 s of compile time
and a 550 KB code object."""
import sys, time, subprocess
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lurk_amd import lair, air
from lurk_amd import _native as N
from lurk_amd.context import _addr
from lurk_amd.programs import synth_eval as se

H_N_INSTR, H_N_REGS, H_N_CONSTS, H_CODE_OFF, H_CONST_OFF, H_FIRST_COLUMN = 1, 2, 3, 7, 8, 10
OPS = {0: 'NOP', 1: 'ADD', 2: 'SUB', 3: 'MUL', 4: 'ASSERT', 5: 'IBEGIN', 6: 'IVAL', 7: 'IEND', 8: 'IVALS', 9: 'IVALT'}

def program(a, which, index):
    n = N.lib.lurkhip_air_program(a.handle, which, index, None, 0)
    out = np.zeros(n, dtype=np.uint32)
    N.lib.lurkhip_air_program(a.handle, which, index, _addr(out), n)
    return out

def gen(prog, name):
    n = int(prog[H_N_INSTR]); code = prog[int(prog[H_CODE_OFF]):]; consts = prog[int(prog[H_CONST_OFF]):]
    lines = [f'template <class Sink> __device__ __forceinline__ void {name}(const airvm::Sources& s, Sink& sink) {{']
    last = {}
    def operand(o):
        idx, ty = o & 0x1fff, o >> 13
        if ty == 0: return last[idx]
        if ty == 1: return f's.main_l[{idx}]'
        if ty == 2: return f's.main_n[{idx}]'
        if ty == 3: return f's.prep_l[{idx}]'
        if ty == 4: return f's.prep_n[{idx}]'
        if ty == 5: return f'{int(consts[idx])}u'
        if ty == 6: return f's.pub[{idx}]'
        return f's.sel[{idx}]'
    for i in range(n):
        w0, w1 = int(code[2 * i]), int(code[2 * i + 1]); op, dst, a, b = w0 & 0xff, w0 >> 8, w1 & 0xffff, w1 >> 16
        o = OPS[op]
        if o in ('ADD', 'SUB', 'MUL'):
            f = {'ADD': 'bb::add', 'SUB': 'bb::sub', 'MUL': 'bb::mul'}[o]
            lines.append(f'    const uint32_t t{i} = {f}({operand(a)}, {operand(b)});'); last[dst] = f't{i}'
        elif o == 'ASSERT': lines.append(f'    sink.assert_zero({operand(a)});')
        elif o == 'IBEGIN': lines.append(f'    sink.ibegin({dst}u, {"true" if a else "false"}, {b}u);')
        elif o == 'IVAL': lines.append(f'    sink.ival({operand(a)});')
        elif o == 'IEND': lines.append(f'    sink.iend({operand(a)});')
        elif o == 'IVALS': lines.append(f'    sink.ival_run(s.main_l + {a}, {b}u, {dst}u);')
        elif o == 'IVALT': lines.append(f'    sink.ival_at({operand(a)}, {dst}u);')
    lines.append('}')
    return '\n'.join(lines)

top = lair.Toplevel(se.SOURCE)
a = air.ChipAir.for_func(top, top.func_index(se.FUNC))
info = np.zeros(16, dtype=np.uint32); N.lib.lurkhip_air_info(a.handle, _addr(info))
n_parts = int(info[14])
src = ['#include <hip/hip_runtime.h>', '#include "stark_kernels.h"', 'namespace lurkhip {']
for j in range(n_parts): src.append(gen(program(a, 2, j), f'perm_piece{j}'))
cases = '\n'.join(f'            case {j}: perm_piece{j}(src, sink); break;' for j in range(n_parts))
src.append(f'''struct JitPermRunner {{
    template <class Sink>
    static __device__ __forceinline__ void run(const uint32_t*, uint32_t wave, const airvm::Sources& src, uint32_t*, Sink& sink) {{
        switch (wave) {{
{cases}
            default: break;
        }}
    }}
}};
}}  // namespace lurkhip
extern "C" __global__ void jit_perm_rows(lurkhip::PermArgs a) {{ lurkhip::perm_rows_body<lurkhip::JitPermRunner>(a); }}
''')
# quotient: cons + coarse pieces
nq = 0
while N.lib.lurkhip_air_program(a.handle, 3, nq, None, 0) > 0: nq += 1
src.append('namespace lurkhip {')
src.append(gen(program(a, 0, 0), 'quot_cons'))
for j in range(nq): src.append(gen(program(a, 3, j), f'quot_piece{j}'))
cases = '            case 0: quot_cons(src, sink); break;\n' + '\n'.join(f'            case {j + 1}: quot_piece{j}(src, sink); break;' for j in range(nq))
src.append(f'''struct JitQuotRunner {{
    template <class Sink>
    static __device__ __forceinline__ void run(const uint32_t*, uint32_t wave, const airvm::Sources& src, uint32_t*, Sink& sink) {{
        switch (wave) {{
{cases}
            default: break;
        }}
    }}
}};
}}  // namespace lurkhip
extern "C" __global__ void jit_quotient(lurkhip::QuotientArgs a) {{ lurkhip::quotient_body<lurkhip::JitQuotRunner>(a); }}
''')
open('gpurun_out/jit_eval.hip', 'w').write('\n'.join(src))
print('source lines', sum(s.count('\n') + 1 for s in src))
t0 = time.time()
r = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-I', 'lurk_amd/csrc', '-x', 'hip', '--cuda-device-only', '-S', '-o', 'gpurun_out/jit_eval.s', 'gpurun_out/jit_eval.hip'], capture_output=True, text=True)
print('compile s', time.time() - t0, r.returncode, r.stderr[-2000:])
