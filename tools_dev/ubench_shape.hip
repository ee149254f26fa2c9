// What would a third wave per SIMD buy the headline kernel?  A model of pair_stream_kernel<K_SAD16>'s tile loop in two shapes:
//   TC = 8: 128 x 128 tiles, four waves, 8 x 8 pairs per lane (64 integer sums + 128 registers of fp64 results), two workgroups per CU  -- the kernel as built
//   TC = 4: 128 x  64 tiles, four waves, 8 x 4 pairs per lane (32 + 64 registers), THREE workgroups per CU                             -- the shape DESIGN section 8 (2) names
// Same ingredients as the kernel: operands read from LDS (ds_read_b64 pairs, a barrier per 8-word chunk), v_sad_u16 into integer sums, a fold into fp64 at the end of a
// tile, the finished tile's results stored DEFERRED -- one group of non-temporal 16-byte stores in front of each 8-word trip of the next tile, the mirrored half after
// 2 x 2 register swaps --, persistent workgroups over a tile list of the upper triangle.  No operand loads from memory (the LDS image is filled once): what is measured
// is the overlap of the store stream with the arithmetic, which is what the shape changes.  Results are not distances; the address pattern and instruction mix are the kernel's.
//   hipcc -O3 --offload-arch=gfx950 tools_dev/ubench_shape.hip -o tools_dev/bin/ubench_shape ;  ubench_shape [n = 98304] [words = 52]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v2f64 __attribute__((ext_vector_type(2)));

__device__ inline int lane_y(int l) { return ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4); }
__device__ inline int lane_x(int l) { return (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4); }
__device__ inline void gstore16_s(uint64_t sbase, uint32_t lane_off, v2f64 val) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(lane_off), "v"(val), "s"(sbase) : "memory");
}

// flags: 1 no stores, 2 no arithmetic
template <int TC, int OCC, int STAG>
__global__ __launch_bounds__(256, OCC) void shape_kernel(double *out, int64_t ld, const int2 *__restrict__ tiles, int ntiles, int words, int flags,
                                                         const uint32_t *__restrict__ seed) {
    constexpr int TR = 8, BI = 128, BJ = 16 * TC;           // tile: 128 rows x (128 | 64) columns
    constexpr int NGD = TR / 2, NGM = TC / 2, NG = NGD + NGM;  // store groups: direct (one row granule each), mirrored (one column granule each)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *sI = lds, *sJ = lds + 16 * BI;                // [16 words][BI] and [16 words][BJ]
    for (int i = threadIdx.x; i < 16 * (BI + BJ); i += 256) lds[i] = seed[i & 4095];
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), l = threadIdx.x & 63;
    auto lane = [&]() __attribute__((always_inline)) -> int { int v = l; asm volatile("" : "+v"(v)); return v; };
    const int wi = (w >> 1) * 64 + 2 * lane_y(l), wj = (w & 1) * (BJ / 2) + 2 * lane_x(l);

    uint32_t part[TR][TC];
    v2f64 acc2[TR][TC / 2];
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int c = 0; c < TC; ++c) { part[r][c] = 0u; acc2[r][c >> 1][c & 1] = 0.0; }
    int pend_g = NG, pend_ti = 0, pend_tj = 0;
    bool pend_mirror = false;

    auto transpose_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int aa = 0; aa < TR / 2; ++aa)
#pragma unroll
            for (int qq = 0; qq < TC / 2; ++qq) {
                const long long bx = __double_as_longlong(acc2[2 * aa][qq][1]), by = __double_as_longlong(acc2[2 * aa + 1][qq][0]);
                int xl = (int)bx, xh = (int)(bx >> 32), yl = (int)by, yh = (int)(by >> 32);
                asm("v_swap_b32 %0, %1" : "+v"(xl), "+v"(yl));
                asm("v_swap_b32 %0, %1" : "+v"(xh), "+v"(yh));
                acc2[2 * aa][qq][1] = __longlong_as_double(((long long)xh << 32) | (unsigned int)xl);
                acc2[2 * aa + 1][qq][0] = __longlong_as_double(((long long)yh << 32) | (unsigned int)yl);
            }
    };
    auto store_group = [&](int g, int gti, int gtj, bool gmirror) __attribute__((always_inline)) {
        if (flags & 1) return;
        if (g >= NGD && !gmirror) return;
        const int lv = lane(), lyv = lane_y(lv), lxv = lane_x(lv);
        const uint32_t ld8 = (uint32_t)ld * 8u;
        const int64_t R0 = (int64_t)BI * gti + 64 * (w >> 1), C0 = (int64_t)BJ * gtj + (BJ / 2) * (w & 1);
#pragma unroll
        for (int gg = 0; gg < NGD; ++gg) {
            if (g == gg) {
                uint64_t sb = reinterpret_cast<uint64_t>(out + (R0 + 16 * gg) * ld + C0);
                asm volatile("s_nop 4" : "+s"(sb));
                const uint32_t vd0 = (uint32_t)(2 * lyv) * ld8 + (uint32_t)(16 * lxv);
#pragma unroll
                for (int par = 0; par < 2; ++par)
#pragma unroll
                    for (int qq = 0; qq < TC / 2; ++qq) gstore16_s(sb, vd0 + (par ? ld8 : 0u) + (uint32_t)(128 * qq), acc2[2 * gg + par][qq]);
            }
        }
#pragma unroll
        for (int gg = 0; gg < NGM; ++gg) {
            if (g == NGD + gg) {
                uint64_t sb = reinterpret_cast<uint64_t>(out + (C0 + 16 * gg) * ld + R0);
                asm volatile("s_nop 4" : "+s"(sb));
                const uint32_t vdm = (uint32_t)(2 * lxv) * ld8 + (uint32_t)(16 * lyv);
#pragma unroll
                for (int par = 0; par < 2; ++par)
#pragma unroll
                    for (int aa = 0; aa < TR / 2; ++aa) gstore16_s(sb, vdm + (par ? ld8 : 0u) + (uint32_t)(128 * aa), acc2[2 * aa + par][gg]);
            }
        }
    };

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int2 tt = tiles[t];
        const int ti = __builtin_amdgcn_readfirstlane(tt.x), tj = __builtin_amdgcn_readfirstlane(tt.y);
        const bool mirror = (int64_t)BJ * tj >= (int64_t)BI * (ti + 1);      // strictly above the diagonal square: also written mirrored
        for (int c0 = 0; c0 < words; c0 += 8) {
            if (STAG != 2) __syncthreads();                  // (the chunk barrier; STAG == 2: none -- how much do the barriers' lock-step cost?)
            if (STAG != 1 || (flags & 2)) {
                if (pend_g < NG) {
                    if (pend_g == NGD && pend_mirror) transpose_acc();
                    store_group(pend_g, pend_ti, pend_tj, pend_mirror);
                    ++pend_g;
                }
            }
            if (!(flags & 2)) {
                const int nw = min(8, words - c0);
#pragma unroll 1
                for (int u = 0; u < nw; u += 2) {
                    if (STAG == 1 && (u == 2 * w || (u == 0 && 2 * w >= nw))) {
                        // STAGGERED: the four waves of a workgroup leave the barrier together; with the group in front of the trip all four
                        // (one per SIMD) queue their eight stores at the CU's store path at once and each waits for everybody's.  Here wave w
                        // issues its group in front of word 2 w of the trip.
                        if (pend_g < NG) {
                            if (pend_g == NGD && pend_mirror) transpose_acc();
                            store_group(pend_g, pend_ti, pend_tj, pend_mirror);
                            ++pend_g;
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int c = (c0 + u + h) & 15;
                        uint32_t av[TR], bv[TC];
#pragma unroll
                        for (int qq = 0; qq < TR / 2; ++qq) {
                            const uint2 v = *reinterpret_cast<const uint2 *>(sI + c * BI + wi + 16 * qq);
                            av[2 * qq] = v.x; av[2 * qq + 1] = v.y;
                        }
#pragma unroll
                        for (int qq = 0; qq < TC / 2; ++qq) {
                            const uint2 v = *reinterpret_cast<const uint2 *>(sJ + c * BJ + wj + 16 * qq);
                            bv[2 * qq] = v.x; bv[2 * qq + 1] = v.y;
                        }
#pragma unroll
                        for (int r = 0; r < TR; ++r)
#pragma unroll
                            for (int cc = 0; cc < TC; ++cc) part[r][cc] = __builtin_amdgcn_sad_u16(av[r], bv[cc], part[r][cc]);
                    }
                }
            }
        }
        // whatever of the previous tile has not left yet, then the fold (overwrites the fp64 results)
        for (; pend_g < NG; ++pend_g) {
            if (pend_g == NGD && pend_mirror) transpose_acc();
            store_group(pend_g, pend_ti, pend_tj, pend_mirror);
        }
        const double s = 1.0 / (double)(ti + tj + 3);
#pragma unroll
        for (int r = 0; r < TR; ++r) {
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) asm volatile("" : "+v"(part[r][cc]));      // (keeps hipcc from converting all 64 sums before the first multiply-add)
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) {
                acc2[r][cc >> 1][cc & 1] = fma((double)part[r][cc], s, 0.25);
                part[r][cc] = 0u;
            }
            __builtin_amdgcn_sched_barrier(0);            // row by row (as the kernel's fold): the converted sums of one row only are live
        }
        pend_g = 0; pend_ti = ti; pend_tj = tj; pend_mirror = mirror;
    }
    for (; pend_g < NG; ++pend_g) {
        if (pend_g == NGD && pend_mirror) transpose_acc();
        store_group(pend_g, pend_ti, pend_tj, pend_mirror);
    }
}

template <int TC, int OCC, int STAG>
static void run(const char *name, double *out, int64_t n, int words, const uint32_t *seed, int cus) {
    constexpr int BJ = 16 * TC;
    std::vector<int2> tiles;
    const int nti = (int)(n / 128), ntj = (int)(n / BJ);
    for (int ti = 0; ti < nti; ++ti)
        for (int tj = ti * (128 / BJ); tj < ntj; ++tj) tiles.push_back(int2{ti, tj});
    int2 *d_tiles;
    CHECK(hipMalloc(&d_tiles, tiles.size() * sizeof(int2)));
    CHECK(hipMemcpy(d_tiles, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
    auto kern = shape_kernel<TC, OCC, STAG>;
    const size_t lds = (size_t)(160 * 1024 / OCC) / 1024 * 1024 - 1024;     // pins OCC workgroups per CU
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = cus * OCC;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int flags : {0, 1, 2}) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, n, d_tiles, (int)tiles.size(), words, flags, seed);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best = ms < best ? ms : best;
        }
        printf("%-46s %-22s %9.3f ms   %7.1f GB/s of output\n", name, flags == 0 ? "stores + arithmetic" : flags == 1 ? "arithmetic only" : "stores only", best,
               flags == 1 ? 0.0 : (double)n * n * 8 / (best * 1e-3) / 1e9);
    }
    CHECK(hipFree(d_tiles));
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 98304;
    const int words = argc > 2 ? atoi(argv[2]) : 52;
    if (n % 128) { fprintf(stderr, "n must be a multiple of 128\n"); return 1; }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    double *out;
    CHECK(hipMalloc(&out, (size_t)n * n * 8));
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (uint32_t)(i * 2654435761u) & 0x00ff00ffu;
    uint32_t *seed;
    CHECK(hipMalloc(&seed, 4096 * 4));
    CHECK(hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    printf("device %s, %d CUs, n = %lld (%.1f GB of output), %d words per pair\n", prop.gcnArchName, prop.multiProcessorCount, (long long)n, (double)n * n * 8 / 1e9, words);
    CHECK(hipMemset(out, 0, (size_t)n * n * 8));
    run<8, 2, 0>("128x128, 8x8/lane, 2 WG/CU", out, n, words, seed, prop.multiProcessorCount);
    run<8, 2, 1>("128x128, 8x8/lane, 2 WG/CU, staggered groups", out, n, words, seed, prop.multiProcessorCount);
    run<8, 2, 2>("128x128, 8x8/lane, 2 WG/CU, no chunk barriers", out, n, words, seed, prop.multiProcessorCount);
    run<4, 3, 0>("128x64, 8x4/lane, 3 WG/CU", out, n, words, seed, prop.multiProcessorCount);
    run<4, 3, 1>("128x64, 8x4/lane, 3 WG/CU, staggered groups", out, n, words, seed, prop.multiProcessorCount);
    run<4, 3, 2>("128x64, 8x4/lane, 3 WG/CU, no chunk barriers", out, n, words, seed, prop.multiProcessorCount);
    run<4, 4, 0>("128x64, 8x4/lane, 4 WG/CU", out, n, words, seed, prop.multiProcessorCount);
    run<4, 4, 1>("128x64, 8x4/lane, 4 WG/CU, staggered groups", out, n, words, seed, prop.multiProcessorCount);
    return 0;
}
