// Store-instruction cost of the pair kernel's tile epilogue as a function of the LANE MAP: a wave stores its 64 x 64 fp64
// quadrant as 8 x 8 blocks per lane, lane coordinates (ey, ex) -> rows 2 ey + {0,1} + 16 a, columns 2 ex + {0,1} + 16 q; the
// direct half stores [row][16 B of the line <- ex], the mirrored half [row <- ex][16 B <- ey].  Which lane bits feed ex and ey is
// free for the arithmetic (LDS operand reads broadcast either way) -- this measures what the store path prefers.
//   hipcc -O3 --offload-arch=gfx950 tools_dev/ubench_store_map.hip -o build/ubench_store_map && build/ubench_store_map [workgroups=64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) v2d gv2d;

// spec: 6 entries (lane bit 0..5), each 0..2 = bit of ex, 4..6 = bit of ey
__global__ __launch_bounds__(256, 2) void k(double *out, long ld, int tiles_per_wg, int tiles_j, unsigned long long *cyc, unsigned spec, int mode) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    int ex = 0, ey = 0;
    for (int b = 0; b < 6; ++b) {
        const int t = (spec >> (4 * b)) & 15;
        if ((l >> b) & 1) { if (t < 4) ex |= 1 << t; else ey |= 1 << (t - 4); }
    }
    v2d v = {1.0 + threadIdx.x, 2.0 + blockIdx.x};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long tile = (long)blockIdx.x * tiles_per_wg + t;
        const long i0 = (tile / tiles_j) * 128, j0 = (tile % tiles_j) * 128 + 128 * (tile / tiles_j + 1);
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long row = i0 + (w >> 1) * 64 + 2 * ey + 16 * (r >> 1) + (r & 1), col = j0 + (w & 1) * 64 + 2 * ex + 16 * q;
                const long mrow = j0 + (w & 1) * 64 + 2 * ex + 16 * (r >> 1) + (r & 1), mcol = i0 + (w >> 1) * 64 + 2 * ey + 16 * q;
                gv2d *p = (gv2d *)(out + row * ld + col), *pm = (gv2d *)(out + mrow * ld + mcol);
                if (mode & 1) __builtin_nontemporal_store(v, p);
                if (mode & 2) __builtin_nontemporal_store(v, pm);
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (l == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}

// "Quad-transposed" mirrored half: after a 4 x 4 exchange between the lanes of a quad and the four row-granule tuples, a store instruction
// holds 2 rows x 4 lines, the QUAD lanes in one row but in four different 128-byte lines (16 bytes each, same offset), bits 3-5 the
// 16-byte piece.  spread: distance of the four lines in granules (1: adjacent lines, 37: scattered).
__global__ __launch_bounds__(256, 2) void kq(double *out, long ld, int tiles_per_wg, int tiles_j, unsigned long long *cyc, int spread) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int g = l & 3, rsel = (l >> 2) & 1, piece = l >> 3;
    v2d v = {1.0 + threadIdx.x, 2.0 + blockIdx.x};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long tile = (long)blockIdx.x * tiles_per_wg + t;
        const long i0 = (tile / tiles_j) * 128, j0 = (tile % tiles_j) * 128 + 128 * (tile / tiles_j + 1);
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long mrow = j0 + (w & 1) * 64 + 16 * q + 2 * (r >> 1) + 8 * rsel + (r & 1);
                const long mcol = i0 + (w >> 1) * 64 + (long)16 * ((g * spread) % 61) + 2 * piece;
                __builtin_nontemporal_store(v, (gv2d *)(out + mrow * ld + mcol));
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (l == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}

int main(int argc, char **argv) {
    const long n = 65536;
    double *out; if (hipMalloc(&out, (size_t)n * n * 8) != hipSuccess) { puts("alloc failed"); return 1; }
    unsigned long long *d_cyc; (void)hipMalloc(&d_cyc, 512 * 4 * 8);
    const int grid = argc > 1 ? atoi(argv[1]) : 64, tiles = 60, tiles_j = (int)(n / 128) / 2;
    struct { const char *name; int m[6]; } maps[] = {
        {"x0 x1 x2 y0 y1 y2 (round 3)", {0, 1, 2, 4, 5, 6}},
        {"x0 y0 x1 y1 x2 y2", {0, 4, 1, 5, 2, 6}},
        {"x0 x1 y0 y1 x2 y2", {0, 1, 4, 5, 2, 6}},
        {"x0 y0 x1 x2 y1 y2", {0, 4, 1, 2, 5, 6}},
        {"x0 y0 y1 x1 x2 y2", {0, 4, 5, 1, 2, 6}},
        {"x0 x1 y0 x2 y1 y2", {0, 1, 4, 2, 5, 6}},
        {"x0 y0 x1 y1 y2 x2", {0, 4, 1, 5, 6, 2}},
        {"x2 y2 x1 y1 x0 y0", {2, 6, 1, 5, 0, 4}},
        {"x0 y0 x2 y2 x1 y1", {0, 4, 2, 6, 1, 5}},
        {"x1 y1 x0 y0 x2 y2", {1, 5, 0, 4, 2, 6}},
        {"x0 x1 y0 y1 y2 x2", {0, 1, 4, 5, 6, 2}},
    };
    printf("%d workgroups; cycles per store instruction per wave: direct / mirrored / both (GB/s both)\n", grid);
    for (auto &mp : maps) {
        unsigned spec = 0;
        for (int b = 0; b < 6; ++b) spec |= (unsigned)mp.m[b] << (4 * b);
        double res[3], gbs = 0;
        for (int mode = 1; mode <= 3; ++mode) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, n, tiles, tiles_j, d_cyc, spec, mode);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, n, tiles, tiles_j, d_cyc, spec, mode);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h((size_t)grid * 4);
            (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
            double c = 0; for (auto x : h) c += (double)x;
            const int per_tile = mode == 3 ? 64 : 32;
            res[mode - 1] = c / h.size() / (tiles * (double)per_tile);
            if (mode == 3) gbs = (double)grid * tiles * 2 * 128 * 128 * 8 / (ms * 1e-3) / 1e9;
        }
        printf("%-30s %6.0f / %6.0f / %6.0f   (%7.1f GB/s)\n", mp.name, res[0], res[1], res[2], gbs);
    }
    for (int spread : {1, 37}) {
        hipLaunchKernelGGL(kq, dim3(grid), dim3(256), 0, 0, out, n, tiles, tiles_j, d_cyc, spread);
        hipLaunchKernelGGL(kq, dim3(grid), dim3(256), 0, 0, out, n, tiles, tiles_j, d_cyc, spread);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)grid * 4);
        (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double c = 0; for (auto x : h) c += (double)x;
        printf("quad-transposed mirrored half (quad = 1 row x 4 lines, lines %2d granules apart): %6.0f cycles per store instruction per wave\n", spread,
               c / h.size() / (tiles * 32.0));
    }
    return 0;
}
