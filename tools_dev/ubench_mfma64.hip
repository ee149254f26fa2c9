// v_mfma_f64_16x16x4_f64 issue rate on gfx950: is the fp64 matrix pipe worth using for the KL dot products?
// Measured (round 2): 49.7-50.1 TFLOP/s = ~100 cycles per MFMA per SIMD -- below the 59-64 TFLOP/s the vector v_fma_f64 stream
// sustains (ubench_valu.hip), and equal to what the KL pair kernel already delivers (0.64 of the 78.6 TFLOP/s peak): no.
//   hipcc -O3 --offload-arch=gfx950 tools_dev/ubench_mfma64.hip -o build/ubench_mfma64 && build/ubench_mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int ITERS = 4096;
__global__ __launch_bounds__(256) void k(double *o, double seed) {
    v4d c[8];
    for (int i = 0; i < 8; ++i) c[i] = v4d{0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-4;
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (s == 12345.678) o[threadIdx.x] = s;
}
// MFMA and vector FMA in one wave: NV independent v_fma_f64 between two MFMAs.  If the matrix pipe and the vector ALU
// ran side by side the time would be max(MFMA, VALU); if fp64 MFMA runs on the vector ALU's own multipliers it is the sum.
template <int NM, int NV>
__global__ __launch_bounds__(256) void kmix(double *o, double seed) {
    v4d c[4];
    for (int i = 0; i < 4; ++i) c[i] = v4d{0, 0, 0, 0};
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = seed * i;
    double a = seed + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-4;
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < NM) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[(i * NV + v) & 7] = fma(acc[(i * NV + v) & 7], b, a);
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 12345.678) o[threadIdx.x] = s;
}
template <int NM, int NV>
static void run_mix(double *o, int cus) {
    const int grid = cus * 8;                  // 2 waves per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kmix<NM, NV>), dim3(grid), dim3(256), 0, 0, o, 1.0);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((kmix<NM, NV>), dim3(grid), dim3(256), 0, 0, o, 2.0 + r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 4;
    const double waves = (double)grid * 4 * ITERS;
    const double flops = waves * (NM * 2.0 * 16 * 16 * 4 + 4.0 * NV * 2.0 * 64);
    printf("mix: %d MFMA + %2d v_fma_f64 per trip: %.3f ms, %.2f TFLOP/s fp64 (MFMA share %.0f %%)\n", NM, 4 * NV, ms,
           flops / (ms * 1e-3) / 1e12, 100.0 * NM * 2048.0 / (NM * 2048.0 + 4.0 * NV * 128.0));
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    double *o; (void)hipMalloc(&o, 4096);
    for (int wps = 1; wps <= 2; ++wps) {
        const int grid = p.multiProcessorCount * wps * 8;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, o, 1.0);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, o, 2.0 + r);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 4;
        const double mfma = (double)grid * 4 * ITERS * 8;            // wave-level MFMA instructions per launch
        const double flops = mfma * 2.0 * 16 * 16 * 4;
        printf("grid %d: %.3f ms, %.2f TFLOP/s fp64, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", grid, ms, flops / (ms * 1e-3) / 1e12,
               ms * 1e-3 * 2.4e9 / (mfma / (p.multiProcessorCount * 4.0)));
    }
    run_mix<4, 0>(o, p.multiProcessorCount);
    run_mix<0, 16>(o, p.multiProcessorCount);
    run_mix<4, 4>(o, p.multiProcessorCount);
    run_mix<4, 8>(o, p.multiProcessorCount);
    run_mix<4, 16>(o, p.multiProcessorCount);
    run_mix<4, 24>(o, p.multiProcessorCount);
    run_mix<2, 24>(o, p.multiProcessorCount);
    return 0;
}
