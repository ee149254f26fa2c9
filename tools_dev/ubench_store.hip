// Store-pattern microbenchmark: how fast can the four waves of a 128 x 128 fp64 tile push their 64 x 1 KiB store instructions
// through a CU, as a function of the access shape?  (pair_stream_kernel measures ~200 cycles per store instruction per wave
// with shape 0.)
//   shape 0: the pair kernel's -- an instruction = 8 rows x 128 B (8 lanes x 16 B per row), direct + mirrored
//   shape 1: row-contiguous -- an instruction = one 1-KiB row segment (64 lanes x 16 B), direct + mirrored
//   shape 2 / 3: 2 rows x 512 B / 4 rows x 256 B (what a wave could reach by exchanging registers between lanes)
//   hipcc -O3 --offload-arch=gfx950 tools_dev/ubench_store.hip -o build/ubench_store && build/ubench_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) v2d gv2d;

template <int SHAPE, bool NT>
__global__ __launch_bounds__(256, 2) void k(double *out, long ld, int tiles_per_wg, int tiles_j, unsigned long long *cyc) {
    extern __shared__ double pin[];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, ly = l >> 3, lx = l & 7;
    v2d v = {1.0 + threadIdx.x, 2.0 + blockIdx.x};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long tile = (long)blockIdx.x * tiles_per_wg + t;
        const long i0 = (tile / tiles_j) * 128, j0 = (tile % tiles_j) * 128 + 128 * (tile / tiles_j + 1);   // above the diagonal
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                long row, col, mrow, mcol;
                if (SHAPE == 0) {
                    row = i0 + (w >> 1) * 64 + 2 * ly + 16 * (r >> 1) + (r & 1); col = j0 + (w & 1) * 64 + 2 * lx + 16 * q;
                    mrow = j0 + (w & 1) * 64 + 2 * lx + 16 * (r >> 1) + (r & 1); mcol = i0 + (w >> 1) * 64 + 2 * ly + 16 * q;
                } else if (SHAPE == 1) {
                    row = i0 + w * 32 + r * 4 + q; col = j0 + 2 * l;
                    mrow = j0 + w * 32 + r * 4 + q; mcol = i0 + 2 * l;
                } else if (SHAPE == 2) {      // 2 rows x 512 B: what a wave's 64 x 64 quadrant allows after an in-register transposition
                    row = i0 + (w >> 1) * 64 + (r * 4 + q) * 2 + (l >> 5); col = j0 + (w & 1) * 64 + 2 * (l & 31);
                    mrow = j0 + (w & 1) * 64 + (r * 4 + q) * 2 + (l >> 5); mcol = i0 + (w >> 1) * 64 + 2 * (l & 31);
                } else if (SHAPE == 6) {      // shape 0's lines with the lane roles exchanged: lx picks the row, ly the 16 bytes of the line
                    row = i0 + (w >> 1) * 64 + 2 * lx + 16 * (r >> 1) + (r & 1); col = j0 + (w & 1) * 64 + 2 * ly + 16 * q;
                    mrow = j0 + (w & 1) * 64 + 2 * lx + 16 * (r >> 1) + (r & 1); mcol = i0 + (w >> 1) * 64 + 2 * ly + 16 * q;
                } else if (SHAPE == 7) {      // direct as shape 0, mirrored as shape 6 (the in-lane 2 x 2 swap of the pair kernel)
                    row = i0 + (w >> 1) * 64 + 2 * ly + 16 * (r >> 1) + (r & 1); col = j0 + (w & 1) * 64 + 2 * lx + 16 * q;
                    mrow = j0 + (w & 1) * 64 + 2 * lx + 16 * (r >> 1) + (r & 1); mcol = i0 + (w >> 1) * 64 + 2 * ly + 16 * q;
                } else if (SHAPE == 8) {      // lane bits interleaved (x0 y0 x1 y1 x2 y2): both halves see 2 rows x 32 B per lane quad
                    const int ex = (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4), ey = ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4);
                    row = i0 + (w >> 1) * 64 + 2 * ey + 16 * (r >> 1) + (r & 1); col = j0 + (w & 1) * 64 + 2 * ex + 16 * q;
                    mrow = j0 + (w & 1) * 64 + 2 * ex + 16 * (r >> 1) + (r & 1); mcol = i0 + (w >> 1) * 64 + 2 * ey + 16 * q;
                } else if (SHAPE == 4) {      // 2 rows x four 128-B lines at scattered granule columns (granule-composed tiles after a
                                              // register exchange of two row bits against the column-granule index)
                    const int g = (l >> 3) & 3, rr2 = l >> 5;
                    row = i0 + (w >> 1) * 64 + (r * 4 + q) * 2 + rr2; col = j0 + (w & 1) * 64 + ((g * 37 + r) % 61) * 16 + 2 * (l & 7);
                    mrow = j0 + (w & 1) * 64 + (r * 4 + q) * 2 + rr2; mcol = i0 + (w >> 1) * 64 + ((g * 37 + r) % 61) * 16 + 2 * (l & 7);
                } else if (SHAPE == 5) {      // 4 rows x two scattered 128-B lines
                    const int g = (l >> 3) & 1, rr4 = l >> 4;
                    row = i0 + (w >> 1) * 64 + (r * 2 + (q >> 1)) * 4 + rr4; col = j0 + (w & 1) * 64 + (((g + 2 * (q & 1)) * 37 + r) % 61) * 16 + 2 * (l & 7);
                    mrow = j0 + (w & 1) * 64 + (r * 2 + (q >> 1)) * 4 + rr4; mcol = i0 + (w >> 1) * 64 + (((g + 2 * (q & 1)) * 37 + r) % 61) * 16 + 2 * (l & 7);
                } else {                      // 4 rows x 256 B
                    row = i0 + (w >> 1) * 64 + (r * 2 + (q >> 1)) * 4 + (l >> 4); col = j0 + (w & 1) * 64 + 32 * (q & 1) + 2 * (l & 15);
                    mrow = j0 + (w & 1) * 64 + (r * 2 + (q >> 1)) * 4 + (l >> 4); mcol = i0 + (w >> 1) * 64 + 32 * (q & 1) + 2 * (l & 15);
                }
                gv2d *p = (gv2d *)(out + row * ld + col), *pm = (gv2d *)(out + mrow * ld + mcol);
                if (NT) { __builtin_nontemporal_store(v, p); __builtin_nontemporal_store(v, pm); }
                else { *p = v; *pm = v; }
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (l == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
    if (v.x == -1.0) pin[threadIdx.x] = v.y;
}

template <int SHAPE, bool NT>
void run(const char *name, double *out, long n, unsigned long long *d_cyc, int grid, int tiles_per_wg) {
    const int tiles_j = (int)(n / 128) / 2;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<SHAPE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NT>), dim3(grid), dim3(256), 65536, 0, out, n, tiles_per_wg, tiles_j, d_cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, NT>), dim3(grid), dim3(256), 65536, 0, out, n, tiles_per_wg, tiles_j, d_cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid * 4);
    (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto x : h) c += (double)x;
    const double bytes = (double)grid * tiles_per_wg * 2 * 128 * 128 * 8;
    printf("%-34s %8.3f ms  %7.1f GB/s  %6.0f cycles per store instruction per wave\n", name, ms, bytes / (ms * 1e-3) / 1e9,
           c / h.size() / (tiles_per_wg * 64.0));
}

int main(int argc, char **argv) {
    const long n = 65536;                       // 34 GB matrix
    double *out; if (hipMalloc(&out, (size_t)n * n * 8) != hipSuccess) { puts("alloc failed"); return 1; }
    unsigned long long *d_cyc; (void)hipMalloc(&d_cyc, 512 * 4 * 8);
    const int grid = argc > 1 ? atoi(argv[1]) : 512, tiles = 60;   // resident workgroups x 60 tiles x 256 KiB (512: 8 GB per launch, HBM-bound; 32-64: the CU store path alone)
    run<0, true>("shape 0 (8 rows x 128 B), nt", out, n, d_cyc, grid, tiles);
    run<0, false>("shape 0 (8 rows x 128 B), plain", out, n, d_cyc, grid, tiles);
    run<1, true>("shape 1 (1 KiB row segment), nt", out, n, d_cyc, grid, tiles);
    run<1, false>("shape 1 (1 KiB row segment), plain", out, n, d_cyc, grid, tiles);
    run<2, true>("shape 2 (2 rows x 512 B), nt", out, n, d_cyc, grid, tiles);
    run<3, true>("shape 3 (4 rows x 256 B), nt", out, n, d_cyc, grid, tiles);
    run<6, true>("shape 6 (shape 0, lane roles exchanged), nt", out, n, d_cyc, grid, tiles);
    run<7, true>("shape 7 (direct 0 + mirrored 6), nt", out, n, d_cyc, grid, tiles);
    run<8, true>("shape 8 (lane bits interleaved), nt", out, n, d_cyc, grid, tiles);
    run<4, true>("shape 4 (2 rows x 4 scattered lines), nt", out, n, d_cyc, grid, tiles);
    run<5, true>("shape 5 (4 rows x 2 scattered lines), nt", out, n, d_cyc, grid, tiles);
    return 0;
}
