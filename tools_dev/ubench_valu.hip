// Instruction-issue microbenchmark for gfx950: cycles per wave-instruction of the vector
// instructions the pair kernels are made of (v_add_f64, v_sad_u32/u16/u8, ...), measured with
// every SIMD of the chip busy.  Build + run (on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 tools_dev/ubench_valu.hip -o gpurun_out/ubench_valu && gpurun_out/ubench_valu
// Prints, per instruction: wave-instructions/s over the whole chip, cycles per wave-instruction per
// SIMD at the clock the kernel itself measured (s_memtime / s_memrealtime), and that clock.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

constexpr int ITERS = 2048;     // loop trips
constexpr int UNROLL = 32;      // independent instructions per trip

struct Stamp { uint64_t cyc, rt; };

// 16 independent 64-bit accumulators / 32 independent 32-bit accumulators, all live in registers.
#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

template <int KIND>
__global__ __launch_bounds__(256) void bench(uint32_t *sink, Stamp *stamps, uint32_t seed) {
    const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint32_t a32[16], b32[16], c32[16];
    double a64[16], b64[16], tmp64, tmp64b;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a32[i] = (t * 2654435761u + i * 40503u + seed) | 1u;
        b32[i] = (t * 2246822519u + i * 9176u + seed * 3u) | 1u;
        c32[i] = i;
        a64[i] = 1.0 + (double)((t + i * 17 + seed) & 1023) * 1e-3;
        b64[i] = 0.5 + (double)((t * 3 + i * 5 + seed) & 1023) * 1e-3;
    }
    uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL / 16; ++u) {
            if constexpr (KIND == 0) {          // v_add_f64 with |.| source modifier: acc += |x|
#define M(i) asm volatile("v_add_f64 %0, %0, |%1|" : "+v"(a64[i]) : "v"(b64[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 1) {   // v_fma_f64
#define M(i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a64[i]) : "v"(b64[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 2) {   // v_sad_u32
#define M(i) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(c32[i]) : "v"(a32[i]), "v"(b32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 3) {   // v_sad_u16
#define M(i) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(c32[i]) : "v"(a32[i]), "v"(b32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 4) {   // v_sad_u8
#define M(i) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(c32[i]) : "v"(a32[i]), "v"(b32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 5) {   // v_add_u32
#define M(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c32[i]) : "v"(a32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 6) {   // v_mul_u32_u24
#define M(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c32[i]) : "v"(a32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 7) {   // v_mad_u32_u24
#define M(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(c32[i]) : "v"(a32[i]), "v"(b32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 8) {   // v_pk_add_f32
#define M(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a64[i]) : "v"(b64[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 9) {   // v_pk_fma_f32
#define M(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a64[i]) : "v"(b64[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 10) {  // v_fma_f32
#define M(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c32[i]) : "v"(a32[i]), "v"(b32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 11) {  // v_pk_sub_i16
#define M(i) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(c32[i]) : "v"(a32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 12) {  // v_pk_max_i16
#define M(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(c32[i]) : "v"(a32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 13) {  // v_max_f64
#define M(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a64[i]) : "v"(b64[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 14) {  // v_add3_u32
#define M(i) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(c32[i]) : "v"(a32[i]), "v"(b32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 15) {  // v_msad_u8
#define M(i) asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(c32[i]) : "v"(a32[i]), "v"(b32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 16) {  // v_qsad_pk_u16_u8 (64-bit src0 / accumulator, 32-bit src1)
#define M(i) asm volatile("v_qsad_pk_u16_u8 %0, %0, %1, %0" : "+v"(a64[i]) : "v"(a32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 17) {  // v_mqsad_pk_u16_u8
#define M(i) asm volatile("v_mqsad_pk_u16_u8 %0, %0, %1, %0" : "+v"(a64[i]) : "v"(a32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 18) {  // alternating v_add_f64 / v_sad_u16: do fp64 and int32 issue side by side?
#define M(i) asm volatile("v_add_f64 %0, %0, |%2|\n\tv_sad_u16 %1, %3, %4, %1" : "+v"(a64[i]), "+v"(c32[i]) : "v"(b64[i]), "v"(a32[i]), "v"(b32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 19) {  // v_dot4_u32_u8
#define M(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(c32[i]) : "v"(a32[i]), "v"(b32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 20) {  // v_pk_add_u16
#define M(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(c32[i]) : "v"(a32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 21) {  // v_sub_u32 + v_max_i32 pair (|a-b| the long way)
#define M(i) asm volatile("v_sub_u32 %0, %1, %2\n\tv_max_i32 %0, %0, %1" : "+v"(c32[i]) : "v"(a32[i]), "v"(b32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 22) {  // v_min_u32 (plain VOP2 integer)
#define M(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(c32[i]) : "v"(a32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 23) {  // v_sad_u16 fed by two operands re-used across 4 accumulators (register-bank pattern of the tile kernel)
#define M(i) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(c32[i]) : "v"(a32[i & 3]), "v"(b32[i >> 2]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 24) {  // v_add_f64 d = a - b ; acc += |d|  (the W1 fp64 pair of instructions)
#define M(i) asm volatile("v_add_f64 %1, %2, -%3\n\tv_add_f64 %0, %0, |%1|" : "+v"(a64[i]), "=&v"(tmp64) : "v"(b64[i & 3]), "v"(b64[4 + (i >> 2)]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 25) {  // v_cvt_f64_u32 (the fold's conversion of an integer sum)
#define M(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a64[i]) : "v"(c32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 26) {  // v_min_f64
#define M(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a64[i]) : "v"(b64[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 27) {  // v_swap_b32 (the mirrored tile's exchange)
#define M(i) asm volatile("v_swap_b32 %0, %1" : "+v"(c32[i]), "+v"(a32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 28) {  // v_mov_b32 (zeroing an integer sum)
#define M(i) asm volatile("v_mov_b32 %0, %1" : "=v"(c32[i]) : "v"(a32[i]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 29) {  // one pair of a layer's fold: cvt, min, fma, zero
#define M(i) asm volatile("v_cvt_f64_u32 %2, %1\n\tv_min_f64 %3, %4, %5\n\tv_fmac_f64 %0, %2, %3\n\tv_mov_b32 %1, 0" : "+v"(a64[i]), "+v"(c32[i]), "=&v"(tmp64), "=&v"(tmp64b) : "v"(b64[i & 3]), "v"(b64[4 + (i >> 2)]));
                REP16(M)
#undef M
            } else if constexpr (KIND == 30) {  // v_cvt_f64_u32 between SADs: does the conversion hold the integer pipe?
#define M(i) asm volatile("v_cvt_f64_u32 %0, %1\n\tv_sad_u16 %1, %2, %3, %1\n\tv_sad_u16 %2, %1, %3, %2\n\tv_sad_u16 %3, %1, %2, %3" : "=v"(a64[i]), "+v"(c32[i]), "+v"(a32[i]), "+v"(b32[i]));
                REP16(M)
#undef M
            }
        }
    }
    uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc ^= c32[i] ^ a32[i] ^ b32[i];
        acc ^= (uint32_t)__double_as_longlong(a64[i]) ^ (uint32_t)(__double_as_longlong(b64[i]) >> 32);
    }
    extern __shared__ uint32_t dyn_lds[];
    if (acc == 0x12345678u) { sink[t] = acc; dyn_lds[threadIdx.x] = acc; }
    if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

struct Case { const char *name; void (*fn)(uint32_t *, Stamp *, uint32_t); int instr_per_slot; };

int main(int argc, char **argv) {
    int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
    const int rounds = argc > 2 ? atoi(argv[2]) : 8;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    // 256 threads = 4 waves = one per SIMD per workgroup; residency is pinned by dynamic LDS (160 KiB / waves_per_simd per
    // workgroup), and the grid holds `rounds` full rounds of resident workgroups so that wall time measures the saturated rate
    const int grid = cus * waves_per_simd * rounds;
    const size_t lds = (size_t)(160 * 1024 / waves_per_simd) / 1024 * 1024;
    uint32_t *sink;
    Stamp *stamps;
    CHECK(hipMalloc(&sink, (size_t)grid * 256 * 4));
    CHECK(hipMalloc(&stamps, (size_t)grid * sizeof(Stamp)));
    const Case cases[] = {
        {"v_add_f64 (acc += |x|)", bench<0>, 1}, {"v_fma_f64", bench<1>, 1}, {"v_sad_u32", bench<2>, 1}, {"v_sad_u16", bench<3>, 1},
        {"v_sad_u8", bench<4>, 1}, {"v_add_u32", bench<5>, 1}, {"v_mul_u32_u24", bench<6>, 1}, {"v_mad_u32_u24", bench<7>, 1},
        {"v_pk_add_f32", bench<8>, 1}, {"v_pk_fma_f32", bench<9>, 1}, {"v_fma_f32", bench<10>, 1}, {"v_pk_sub_i16", bench<11>, 1},
        {"v_pk_max_i16", bench<12>, 1}, {"v_max_f64", bench<13>, 1}, {"v_add3_u32", bench<14>, 1}, {"v_msad_u8", bench<15>, 1},
        {"v_qsad_pk_u16_u8", bench<16>, 1}, {"v_mqsad_pk_u16_u8", bench<17>, 1}, {"v_add_f64 + v_sad_u16 alternating", bench<18>, 2},
        {"v_dot4_u32_u8", bench<19>, 1}, {"v_pk_add_u16", bench<20>, 1}, {"v_sub_u32 + v_max_i32", bench<21>, 2}, {"v_min_u32", bench<22>, 1},
        {"v_sad_u16 (4x4 operand reuse)", bench<23>, 1}, {"v_add_f64 sub + v_add_f64 |acc| (W1 fp64 pair)", bench<24>, 2},
        {"v_cvt_f64_u32", bench<25>, 1}, {"v_min_f64", bench<26>, 1}, {"v_swap_b32", bench<27>, 1}, {"v_mov_b32", bench<28>, 1},
        {"fold of one pair: cvt + min + fmac + mov 0", bench<29>, 4}, {"v_cvt_f64_u32 + 3 v_sad_u16", bench<30>, 4},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("device %s, %d CUs, %d waves per SIMD, %d x %d instructions per wave\n", prop.gcnArchName, cus, waves_per_simd, ITERS, UNROLL);
    printf("%-50s %10s %14s %12s %10s\n", "instruction", "ms", "Twave-instr/s", "cyc/instr", "clock GHz");
    for (const Case &c : cases) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(c.fn, dim3(grid), dim3(256), lds, 0, sink, stamps, 1u);   // warm + settle the clock
        CHECK(hipEventRecord(e0));
        const int launches = 4;
        for (int rep = 0; rep < launches; ++rep) hipLaunchKernelGGL(c.fn, dim3(grid), dim3(256), lds, 0, sink, stamps, 2u + rep);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= launches;
        std::vector<Stamp> h(grid);
        CHECK(hipMemcpy(h.data(), stamps, (size_t)grid * sizeof(Stamp), hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0;
        for (auto &s : h) { cyc += (double)s.cyc; rt += (double)s.rt; }
        const double clock_ghz = cyc / rt * 0.1;                    // s_memrealtime ticks at 100 MHz
        const double winstr = (double)grid * 4 * ITERS * UNROLL * c.instr_per_slot;   // wave-instructions in one launch
        // saturated rate: wall time x clock / wave-instructions issued by one SIMD
        const double per_simd_cycles = ms * 1e-3 * clock_ghz * 1e9 / (winstr / (cus * 4.0));
        printf("%-50s %10.4f %14.3f %12.3f %10.3f\n", c.name, ms, winstr / (ms * 1e-3) / 1e12, per_simd_cycles, clock_ghz);
    }
    return 0;
}
