#!/usr/bin/env python3
"""Generates build/ubench_bank.hip: does the VGPR BANK of a v_sad_u16's three sources decide how fast ONE wave can issue it?
(tools_dev/ubench_valu.hip: 4.1 cycles per instruction with two waves per SIMD, 8.7 with one -- the pair kernel's partner wave runs alone whenever
the other waits for its stores.)  8 row operands a[r] = v[A + r], 8 column operands b[c] = v[B + c], 64 accumulators v[ACC + 8 r + c], as in the
kernel's 8 x 8 block; cases differ in A, B, ACC modulo 4.  usage: gen_ubench_bank.py > build/ubench_bank.hip"""
CASES = [("A0_B0_ACC0", 32, 40, 64), ("A0_B1_ACC2", 32, 41, 66), ("A1_B2_ACC0", 33, 42, 64), ("A0_B0_ACC1", 32, 40, 65), ("A0_B1_ACC1", 32, 41, 65),
         ("A2_B0_ACC0", 34, 40, 64)]
print('#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n#include <vector>\n')
print('struct Stamp { unsigned long long c0, c1; };')
for name, A, B, ACC in CASES:
    body = []
    for r in range(8):
        for c in range(8):
            body.append(f"v_sad_u16 v{ACC + 8 * r + c}, v{A + r}, v{B + c}, v{ACC + 8 * r + c}")
    clob = ", ".join(f'"v{i}"' for i in range(32, 132)) + ', "s4", "scc"'
    asm = "\\n\\t".join(["s_movk_i32 s4, 0x400", "1:"] + body + ["s_sub_u32 s4, s4, 1", "s_cmp_lg_u32 s4, 0", "s_cbranch_scc1 1b"])
    print(f'__global__ __launch_bounds__(256) void k_{name}(Stamp *st, unsigned *sink) {{')
    print('    unsigned long long c0 = __builtin_amdgcn_s_memrealtime();')
    print(f'    asm volatile("{asm}" ::: {clob});')
    print('    unsigned long long c1 = __builtin_amdgcn_s_memrealtime();')
    print('    if ((threadIdx.x & 63) == 0) { st[(blockIdx.x * 256 + threadIdx.x) >> 6] = Stamp{c0, c1}; }')
    print('    if (sink && threadIdx.x == 999) sink[0] = 1;\n}')
print('''
int main() {
    Stamp *st; hipMalloc(&st, 4096 * sizeof(Stamp));
    std::vector<Stamp> h(4096);
    const double instr = 1024.0 * 64;
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu;''')
for name, A, B, ACC in CASES:
    print(f'        {{ hipLaunchKernelGGL(k_{name}, dim3(grid), dim3(256), 0, 0, st, (unsigned *)nullptr); hipDeviceSynchronize();')
    print(f'          hipLaunchKernelGGL(k_{name}, dim3(grid), dim3(256), 0, 0, st, (unsigned *)nullptr); hipDeviceSynchronize();')
    print('          hipMemcpy(h.data(), st, grid * 4 * sizeof(Stamp), hipMemcpyDeviceToHost); double s = 0; for (int i = 0; i < grid * 4; ++i) s += (double)(h[i].c1 - h[i].c0);')
    print(f'          printf("%d wave(s) per SIMD  {name:12s}: %.2f ns per v_sad_u16 and wave x 0.1 (100 MHz ticks; x 24 = cycles at 2.4 GHz)\\n", wgs_per_cu, s / (grid * 4) / instr); }}')
print('''    }
    return 0;
}''')
