// Chebyshev step as an SpMM on a scale-free graph: where do the neighbour-row gathers come from?
//   A  node-major block [n][C], one workgroup per row (the round-1/2 kernel shape): every gather leaves the XCD's L2
//   B  column slices of W doubles, slice-major [C/W][n][W]; workgroups b and b+8 share an XCD, so slice s is
//      walked only by the XCD s % 8 and its n*W*8 bytes (3.2 MB at n = 50k, W = 8) stay in that XCD's 4 MiB L2
//   hipcc -O3 --offload-arch=gfx950 tools_dev/ubench_spmm.hip -o build/ubench_spmm && build/ubench_spmm [n] [C]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LCHUNK = 64;

__device__ inline double row_dot(const int32_t *__restrict__ col, const double *__restrict__ val, int64_t b, int64_t e,
                                 const double *__restrict__ T, int64_t stride, int64_t c, double t) {
    int64_t p = b;
    for (; p + 8 <= e; p += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = T[(int64_t)col[p + u] * stride + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) t = fma(val[p + u], x[u], t);
    }
    if (p < e) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = (p + u < e) ? T[(int64_t)col[p + u] * stride + c] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (p + u < e) t = fma(val[p + u], x[u], t);
    }
    return t;
}

// ---- A: node-major ----
__global__ __launch_bounds__(256) void stepA(const int64_t *rowptr, const int32_t *col, const double *val, int64_t n, int C,
                                             const double *Tc, const double *To, double *Tn, double alpha, double beta, double gamma,
                                             const int32_t *row_first, const double *partial) {
    const int64_t i = blockIdx.x;
    const int64_t b = rowptr[i], e = rowptr[i + 1];
    const int ch0 = row_first[i];
    const int nch = (int)((e - b + LCHUNK - 1) / LCHUNK);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double t = 0.0;
        if (ch0 >= 0) for (int q = 0; q < nch; ++q) t += partial[(int64_t)(ch0 + q) * C + c];
        else t = row_dot(col, val, b, e, Tc, C, c, t);
        Tn[i * C + c] = alpha * t + beta * Tc[i * C + c] + gamma * To[i * C + c];
    }
}
__global__ __launch_bounds__(256) void partA(const int64_t *rowptr, const int32_t *col, const double *val, int C, const double *Tc,
                                             const int32_t *chunk_row, const int64_t *chunk_p0, double *partial) {
    const int64_t q = blockIdx.x;
    const int64_t b = chunk_p0[q], e = min(b + (int64_t)LCHUNK, rowptr[chunk_row[q] + 1]);
    for (int c = threadIdx.x; c < C; c += blockDim.x) partial[q * C + c] = row_dot(col, val, b, e, Tc, C, c, 0.0);
}

// ---- B: slices ----
// element (i, c) of a plane lives at ((c / W) * n + i) * W + c % W.  Workgroup b: XCD class x = b % 8, then (slice group,
// row block) in dispatch order, so an XCD finishes one slice before it starts the next.
template <int W, int NT_HINT>
__global__ __launch_bounds__(256) void stepB(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                             const double *__restrict__ val, int64_t n, int nslices, int rows_per_wg, int n_rb,
                                             const double *__restrict__ Tc, const double *__restrict__ To, double *__restrict__ Tn,
                                             double alpha, double beta, double gamma, const int32_t *__restrict__ row_first,
                                             const double *__restrict__ partial, int n_chunks) {
    const int x = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int sg = local / n_rb, rb = local % n_rb;
    const int slice = sg * 8 + x;
    if (slice >= nslices) return;
    const int g = threadIdx.x / W, w = threadIdx.x % W;
    const double *Tcs = Tc + (int64_t)slice * n * W;
    const double *Tos = To + (int64_t)slice * n * W;
    double *Tns = Tn + (int64_t)slice * n * W;
    const double *ps = partial + (int64_t)slice * n_chunks * W;
    const int64_t r_end = min(n, (int64_t)(rb + 1) * rows_per_wg);
    for (int64_t i = (int64_t)rb * rows_per_wg + g; i < r_end; i += 256 / W) {
        const int64_t b = rowptr[i], e = rowptr[i + 1];
        const int ch0 = row_first[i];
        double t = 0.0;
        if (ch0 >= 0) {
            const int nch = (int)((e - b + LCHUNK - 1) / LCHUNK);
            for (int q = 0; q < nch; ++q) t += ps[(int64_t)(ch0 + q) * W + w];
        } else {
            t = row_dot(col, val, b, e, Tcs, W, w, t);
        }
        double o, cur = Tcs[i * W + w];
        if (NT_HINT) o = __builtin_nontemporal_load(Tos + i * W + w); else o = Tos[i * W + w];
        const double v = alpha * t + beta * cur + gamma * o;
        if (NT_HINT) __builtin_nontemporal_store(v, Tns + i * W + w); else Tns[i * W + w] = v;
    }
}
template <int W>
__global__ __launch_bounds__(256) void partB(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                             const double *__restrict__ val, int64_t n, int nslices, const double *__restrict__ Tc,
                                             const int32_t *__restrict__ chunk_row, const int64_t *__restrict__ chunk_p0,
                                             double *__restrict__ partial, int n_chunks, int n_cb) {
    const int x = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int sg = local / n_cb, cb = local % n_cb;
    const int slice = sg * 8 + x;
    if (slice >= nslices) return;
    const int g = threadIdx.x / W, w = threadIdx.x % W;
    const double *Tcs = Tc + (int64_t)slice * n * W;
    double *ps = partial + (int64_t)slice * n_chunks * W;
    const int q = cb * (256 / W) + g;
    if (q >= n_chunks) return;
    const int64_t b = chunk_p0[q], e = min(b + (int64_t)LCHUNK, rowptr[chunk_row[q] + 1]);
    ps[(int64_t)q * W + w] = row_dot(col, val, b, e, Tcs, W, w, 0.0);
}

static uint64_t rng_state = 88172645463325252ull;
static inline uint64_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 50000;
    const int C = argc > 2 ? atoi(argv[2]) : 512;
    const int m = 5;
    // Barabasi-Albert by the repeated-endpoints list
    std::vector<std::vector<int>> adj((size_t)n);
    std::vector<int> rep;
    for (int v = 0; v < m; ++v) rep.push_back(v);
    for (int64_t v = m; v < n; ++v) {
        std::vector<int> tg;
        while ((int)tg.size() < m) {
            const int t = v == m ? (int)tg.size() : rep[rng() % rep.size()];
            if (std::find(tg.begin(), tg.end(), t) == tg.end()) tg.push_back(t);
        }
        for (int t : tg) { adj[(size_t)v].push_back(t); adj[(size_t)t].push_back((int)v); rep.push_back(t); rep.push_back((int)v); }
    }
    std::vector<int64_t> rowptr((size_t)n + 1, 0);
    std::vector<int32_t> col;
    std::vector<double> val;
    int maxdeg = 0;
    for (int64_t i = 0; i < n; ++i) {
        auto &a = adj[(size_t)i];
        maxdeg = std::max<int>(maxdeg, (int)a.size());
        a.push_back((int)i);
        std::sort(a.begin(), a.end());
        for (int u : a) { col.push_back(u); val.push_back(u == i ? (double)(a.size() - 1) : -1.0); }
        rowptr[(size_t)i + 1] = (int64_t)col.size();
    }
    const int64_t nnz = (int64_t)col.size();
    std::vector<int32_t> row_first((size_t)n, -1), chunk_row;
    std::vector<int64_t> chunk_p0;
    for (int64_t i = 0; i < n; ++i) {
        if (rowptr[i + 1] - rowptr[i] <= 2 * LCHUNK) continue;
        row_first[(size_t)i] = (int32_t)chunk_row.size();
        for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p += LCHUNK) { chunk_row.push_back((int32_t)i); chunk_p0.push_back(p); }
    }
    const int n_chunks = (int)chunk_row.size();
    printf("n=%lld nnz(L)=%lld maxdeg=%d C=%d long-row chunks=%d\n", (long long)n, (long long)nnz, maxdeg, C, n_chunks);

    int64_t *d_rowptr, *d_p0; int32_t *d_col, *d_rf, *d_cr; double *d_val, *T[3], *d_part;
    CK(hipMalloc(&d_rowptr, (n + 1) * 8)); CK(hipMalloc(&d_col, nnz * 4)); CK(hipMalloc(&d_val, nnz * 8));
    CK(hipMalloc(&d_rf, n * 4)); CK(hipMalloc(&d_cr, std::max(1, n_chunks) * 4)); CK(hipMalloc(&d_p0, std::max(1, n_chunks) * 8));
    CK(hipMalloc(&d_part, (size_t)std::max(1, n_chunks) * C * 8));
    for (int t = 0; t < 3; ++t) CK(hipMalloc(&T[t], (size_t)n * C * 8));
    CK(hipMemcpy(d_rowptr, rowptr.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_val, val.data(), nnz * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rf, row_first.data(), n * 4, hipMemcpyHostToDevice));
    if (n_chunks) { CK(hipMemcpy(d_cr, chunk_row.data(), n_chunks * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_p0, chunk_p0.data(), n_chunks * 8, hipMemcpyHostToDevice)); }
    // logical contents: X(i, c) pseudo-random, the same in both layouts
    std::vector<double> hA((size_t)n * C), hB((size_t)n * C), out((size_t)n * C), ref((size_t)n * C);
    auto fill = [&](int W) {
        for (int64_t i = 0; i < n; ++i)
            for (int c = 0; c < C; ++c) {
                const double v = (double)((i * 131 + c * 7919) % 1009) / 1009.0 - 0.5;
                hA[(size_t)(i * C + c)] = v;
                if (W) hB[(size_t)(((int64_t)(c / W) * n + i) * W + c % W)] = v;
            }
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double alpha = 0.01, beta = -0.5, gamma = -1.0;
    const int reps = 20;
    const double gather_gb = (double)nnz * C * 8 / 1e9, plane_gb = (double)n * C * 8 / 1e9;

    fill(0);
    for (int t = 0; t < 3; ++t) CK(hipMemcpy(T[t], hA.data(), (size_t)n * C * 8, hipMemcpyHostToDevice));
    {
        const dim3 blk(std::min(256, (C + 63) / 64 * 64));
        float ms = 0;
        for (int r = -2; r < reps; ++r) {
            if (r == 0) CK(hipEventRecord(e0));
            if (n_chunks) hipLaunchKernelGGL(partA, dim3(n_chunks), blk, 0, 0, d_rowptr, d_col, d_val, C, T[0], d_cr, d_p0, d_part);
            hipLaunchKernelGGL(stepA, dim3((unsigned)n), blk, 0, 0, d_rowptr, d_col, d_val, n, C, T[0], T[1], T[2], alpha, beta, gamma, d_rf, d_part);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        CK(hipMemcpy(ref.data(), T[2], (size_t)n * C * 8, hipMemcpyDeviceToHost));
        printf("A node-major          : %.3f ms/step  gathers %.2f GB (%.2f TB/s)  planes 3 x %.3f GB\n", ms, gather_gb, gather_gb / ms, plane_gb);
    }
    auto runB = [&](auto kern, auto pkern, int W, int rows_per_wg, const char *name) {
        fill(W);
        for (int t = 0; t < 3; ++t) CK(hipMemcpy(T[t], hB.data(), (size_t)n * C * 8, hipMemcpyHostToDevice));
        const int nslices = C / W, n_rb = (int)((n + rows_per_wg - 1) / rows_per_wg), sgroups = (nslices + 7) / 8;
        const int n_cb = (n_chunks + 256 / W - 1) / (256 / W);
        float ms = 0;
        for (int r = -2; r < reps; ++r) {
            if (r == 0) CK(hipEventRecord(e0));
            if (n_chunks) hipLaunchKernelGGL(pkern, dim3(8 * sgroups * n_cb), dim3(256), 0, 0, d_rowptr, d_col, d_val, n, nslices, T[0], d_cr, d_p0, d_part, n_chunks, n_cb);
            hipLaunchKernelGGL(kern, dim3(8 * sgroups * n_rb), dim3(256), 0, 0, d_rowptr, d_col, d_val, n, nslices, rows_per_wg, n_rb, T[0], T[1], T[2], alpha, beta, gamma, d_rf, d_part, n_chunks);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        CK(hipMemcpy(out.data(), T[2], (size_t)n * C * 8, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (int64_t i = 0; i < n; ++i)
            for (int c = 0; c < C; ++c)
                if (out[(size_t)(((int64_t)(c / W) * n + i) * W + c % W)] != ref[(size_t)(i * C + c)]) ++bad;
        printf("B %-20s: %.3f ms/step  (W=%d, %d rows/wg, %d wgs)  mismatches vs A: %lld\n", name, ms, W, rows_per_wg, 8 * sgroups * n_rb, (long long)bad);
    };
    runB(stepB<8, 0>, partB<8>, 8, 256, "W=8 plain");
    runB(stepB<8, 1>, partB<8>, 8, 256, "W=8 nt streams");
    runB(stepB<8, 1>, partB<8>, 8, 64, "W=8 nt 64 rows");
    runB(stepB<8, 1>, partB<8>, 8, 1024, "W=8 nt 1024 rows");
    runB(stepB<16, 0>, partB<16>, 16, 256, "W=16 plain");
    runB(stepB<16, 1>, partB<16>, 16, 256, "W=16 nt streams");
    runB(stepB<4, 1>, partB<4>, 4, 256, "W=4 nt streams");
    runB(stepB<32, 1>, partB<32>, 32, 256, "W=32 nt streams");
    return 0;
}
