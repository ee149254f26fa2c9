// K4 and the reference's own ("heat") signature path:
//   * Chebyshev heat-kernel wavelets psi = exp(-s L) applied to blocks of unit impulses
//     (replaces HSD.calculate_wavelets, model/HSD.py:48-67: PyGSP cheby_op per impulse, or eigh)
//   * per-ring sum / mean of coefficients (MultiHSD.get_triple, model/multiscale_HSD.py:45-61)
//   * ragged per-ring sorted coefficient samples (HSD.get_hierarchical_coeffcients, model/HSD.py:71-83)
//   * exact all-pairs W1 on those ragged samples (Variant A, model/HSD.py:98-114: scipy _cdf_distance)
//   * Variant B exactly as written (model/HSD.py:140-161): q is read from row `startIndex`
// psi is dense N x N, so this path is for graphs whose N^2 fits in HBM (N <= ~150k on 288 GB);
// the histogram path (signature.hip / pair.hip) is the one that scales.
#include "common.hpp"

#include <algorithm>
#include <cmath>

namespace {

// sum over CSR entries [b, e) of val[p] * Tcur[col[p] * C + c], added in entry order; the gathers of eight entries are
// issued together.  Measured on BA N = 50k (config 5, one box, tools_dev/cfg5.py): this form 2.08-2.10 s; one wave per row
// with 16-byte gathers 2.39-2.40 s; one workgroup per row with two columns (16 bytes) per thread 2.36-2.39 s.  The step
// moves (nnz + 2 N) C 8 bytes through the L2-miss path at 3-4 TB/s whatever the access shape: the neighbour rows of a
// scale-free graph are all over the 100 MB plane, so the XCD L2s (4 MB) only keep the hubs.
__device__ inline double gather_row_dot(const int32_t *__restrict__ col, const double *__restrict__ val, int64_t b, int64_t e,
                                        const double *__restrict__ Tcur, int64_t C, int64_t c, double t) {
    int64_t p = b;
    for (; p + 8 <= e; p += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = Tcur[(int64_t)col[p + u] * C + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) t = fma(val[p + u], x[u], t);
    }
    if (p < e) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = (p + u < e) ? Tcur[(int64_t)col[p + u] * C + c] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (p + u < e) t = fma(val[p + u], x[u], t);
    }
    return t;
}

// ---- Laplacian SpMM step of the Chebyshev recurrence -------------------------------------
// T_new = alpha * (L T_cur) + beta * T_cur + gamma * T_old ;  R += ck * T_new
// Dense blocks are node-major [n][C] (C = impulse columns of this block) so that one CSR
// neighbour contributes one contiguous C*8-byte row: every load and store is coalesced.
// One workgroup per node row, threads over the C columns.
__global__ __launch_bounds__(256) void cheby_step_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                        const double *__restrict__ val, int64_t n, int C,
                                                        const double *__restrict__ Tcur, const double *__restrict__ Told,
                                                        double *__restrict__ Tnew, double *__restrict__ R, double alpha,
                                                        double beta, double gamma, double ck, int first,
                                                        const int32_t *__restrict__ row_first, const double *__restrict__ partial) {
    const int64_t i = blockIdx.x;
    const int64_t b = rowptr[i], e = rowptr[i + 1];
    const int chunk0 = row_first ? row_first[i] : -1;
    const int nch = (int)((e - b + HSD_LCHUNK - 1) / HSD_LCHUNK);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double t = 0.0;
        if (chunk0 >= 0) {
            // long row: its chunks were multiplied by cheby_partial_kernel; add them in chunk order
            for (int q = 0; q < nch; ++q) t += partial[(int64_t)(chunk0 + q) * C + c];
        } else {
            t = gather_row_dot(col, val, b, e, Tcur, C, c, t);
        }
        const double cur = Tcur[i * C + c];
        double v = alpha * t + beta * cur;
        if (!first) v += gamma * Told[i * C + c];
        Tnew[i * C + c] = v;
        if (R) R[i * C + c] += ck * v;        // R == nullptr: the caller keeps every T_k and combines them afterwards
    }
}

// partial[chunk][c] = sum over the chunk's CSR entries of val * Tcur[col][c]   (one workgroup per chunk)
__global__ __launch_bounds__(256) void cheby_partial_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                           const double *__restrict__ val, int C, const double *__restrict__ Tcur,
                                                           const int32_t *__restrict__ chunk_row, const int64_t *__restrict__ chunk_p0,
                                                           double *__restrict__ partial) {
    const int64_t q = blockIdx.x;
    const int64_t b = chunk_p0[q], e = min(b + HSD_LCHUNK, rowptr[chunk_row[q] + 1]);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        partial[q * C + c] = gather_row_dot(col, val, b, e, Tcur, C, c, 0.0);
    }
}

// Multi-scale combine: R_s = c_0[s]/2 * T_0 + sum_{k>=1} c_k[s] * T_k for every scale s, from the stored
// planes T_0..T_order (plane stride `stride` doubles).  Same accumulation order as the fused per-step
// update (k ascending, one fma per term), so the results are identical to it bit for bit; the difference is
// traffic: each T_k is read once here instead of S accumulators being read and written at every step.
// coef[k * coef_stride + s] for the S scales of this pass, with the k = 0 row already halved.
template <int S_MAX>
__global__ __launch_bounds__(256) void cheby_combine_kernel(const double *__restrict__ Tall, int64_t stride, int order,
                                                           const double *__restrict__ coef, int coef_stride, int S, int64_t cnt,
                                                           double *__restrict__ R) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cnt) return;
    double acc[S_MAX];
    const double t0 = Tall[idx];
#pragma unroll
    for (int s = 0; s < S_MAX; ++s) acc[s] = s < S ? coef[s] * t0 : 0.0;
    int k = 1;
    for (; k + 4 <= order + 1; k += 4) {
        const double a = Tall[(int64_t)k * stride + idx], b = Tall[(int64_t)(k + 1) * stride + idx];
        const double c = Tall[(int64_t)(k + 2) * stride + idx], d = Tall[(int64_t)(k + 3) * stride + idx];
#pragma unroll
        for (int s = 0; s < S_MAX; ++s)
            if (s < S) {
                acc[s] = fma(coef[(int64_t)k * coef_stride + s], a, acc[s]);
                acc[s] = fma(coef[(int64_t)(k + 1) * coef_stride + s], b, acc[s]);
                acc[s] = fma(coef[(int64_t)(k + 2) * coef_stride + s], c, acc[s]);
                acc[s] = fma(coef[(int64_t)(k + 3) * coef_stride + s], d, acc[s]);
            }
    }
    for (; k <= order; ++k) {
        const double a = Tall[(int64_t)k * stride + idx];
#pragma unroll
        for (int s = 0; s < S_MAX; ++s)
            if (s < S) acc[s] = fma(coef[(int64_t)k * coef_stride + s], a, acc[s]);
    }
#pragma unroll
    for (int s = 0; s < S_MAX; ++s)
        if (s < S) R[(int64_t)s * cnt + idx] = acc[s];
}

// Multi-scale variant: the Chebyshev polynomials T_k(L~) e_c do not depend on the heat scale, only
// the coefficients do, so ONE recurrence serves all S scales: R_s += c_k[s] * T_k for s < S.
// R holds S blocks of n*C doubles.  (MultiHSD runs 8..200 scales: model/multiscale_HSD.py:38,80.)
struct CoefS { double c[32]; };
__global__ __launch_bounds__(256) void cheby_step_multi_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                              const double *__restrict__ val, int64_t n, int C,
                                                              const double *__restrict__ Tcur, const double *__restrict__ Told,
                                                              double *__restrict__ Tnew, double *__restrict__ R, double alpha,
                                                              double beta, double gamma, CoefS ck, int S, int first) {
    const int64_t i = blockIdx.x;
    const int64_t b = rowptr[i], e = rowptr[i + 1];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double t = gather_row_dot(col, val, b, e, Tcur, C, c, 0.0);
        const double cur = Tcur[i * C + c];
        double v = alpha * t + beta * cur;
        if (!first) v += gamma * Told[i * C + c];
        Tnew[i * C + c] = v;
        for (int s = 0; s < S; ++s) R[(int64_t)s * n * C + i * C + c] += ck.c[s] * v;
    }
}

__global__ void cheby_init_multi_kernel(int64_t n, int C, int64_t c0, CoefS half_c0, int S, double *__restrict__ T0,
                                        double *__restrict__ R) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * C) return;
    const int64_t i = idx / C;
    const int c = (int)(idx % C);
    const double v = (i == c0 + c) ? 1.0 : 0.0;
    T0[idx] = v;
    for (int s = 0; s < S; ++s) R[(int64_t)s * n * C + idx] = half_c0.c[s] * v;
}

// sums[(s * nsrc + src) * L + k] = sum over ring_k(src) of threshold(R_s[v * C + c]); one workgroup
// per (source of the block, ring), grid.y = scale; fixed-order tree reduction.
__global__ __launch_bounds__(256) void ring_sum_nodemajor_kernel(const int64_t *__restrict__ offsets /* of the block's first source */,
                                                                const int32_t *__restrict__ members, const double *__restrict__ R,
                                                                int64_t n, int C, int L, double thr, int use_thr,
                                                                double *__restrict__ sums, int64_t nsrc_total, int64_t src_base) {
    __shared__ double red[256];
    const int64_t seg = blockIdx.x;
    const int c = (int)(seg / L);
    const int s = blockIdx.y;
    const int64_t b = offsets[seg], e = offsets[seg + 1];
    const double *Rs = R + (int64_t)s * n * C;
    double acc = 0.0;
    for (int64_t p = b + threadIdx.x; p < e; p += 256) {
        double v = Rs[(int64_t)members[p] * C + c];
        if (use_thr) v = v > thr ? v : 0.0;
        acc += v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[((int64_t)s * nsrc_total + src_base + c) * L + (seg % L)] = red[0];
}

// T0 = unit impulses of nodes c0 .. c0+C-1 ; R = c0/2 * T0
__global__ void cheby_init_kernel(int64_t n, int C, int64_t c0, double half_c0, double *__restrict__ T0, double *__restrict__ R) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * C) return;
    const int64_t i = idx / C;
    const int c = (int)(idx % C);
    const double v = (i == c0 + c) ? 1.0 : 0.0;
    T0[idx] = v;
    R[idx] = half_c0 * v;
}

// out[(c) * ld + i] = threshold(R[i * C + c])   (row c0+c of psi; model/HSD.py:65-66 keeps x iff x > thr)
__global__ void transpose_threshold_kernel(const double *__restrict__ R, int64_t n, int C, int Cvalid, double thr, int use_thr,
                                           double *__restrict__ out, int64_t ld) {
    __shared__ double tile[32][33];
    const int64_t i0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int64_t i = i0 + r;
        const int c = c0 + threadIdx.x;
        tile[r][threadIdx.x] = (i < n && c < C) ? R[i * C + c] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int c = c0 + r;
        const int64_t i = i0 + threadIdx.x;
        if (c < Cvalid && i < n) {
            double v = tile[threadIdx.x][r];
            if (use_thr) v = v > thr ? v : 0.0;
            out[(int64_t)c * ld + i] = v;
        }
    }
}

// ---- ring statistics ----------------------------------------------------------------------
// For sources [r0, r1): sums[(i-r0)*(hop+1)+k] = sum of psi[i, v] over ring_k(i) given the ragged
// member lists.  One workgroup per (source, ring) segment, fixed-order tree reduction.
__global__ __launch_bounds__(256) void ring_sum_kernel(const int64_t *__restrict__ offsets, const int32_t *__restrict__ members,
                                                      const double *__restrict__ psi_rows, int64_t ld, int L,
                                                      double *__restrict__ sums) {
    __shared__ double red[256];
    const int64_t seg = blockIdx.x;
    const int64_t src = seg / L;
    const int64_t b = offsets[seg], e = offsets[seg + 1];
    const double *row = psi_rows + src * ld;
    double s = 0.0;
    for (int64_t p = b + threadIdx.x; p < e; p += 256) s += row[members[p]];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[seg] = red[0];
}

// ---- ragged sorted samples ------------------------------------------------------------------
// values[p] = psi[src, members[p]] sorted ascending inside each (source, ring) segment.
// One workgroup per segment of the list; bitonic network over the segment padded to a power of two with
// +inf.  BUF = 0: the padded segment lives in LDS (host sizes the launch by length class);
// BUF = 1: it lives in a per-workgroup slice of global scratch (rings longer than the LDS holds).
template <int BUF, int NT>
__global__ __launch_bounds__(NT) void ring_gather_sort_kernel(const int64_t *__restrict__ offsets, const int32_t *__restrict__ members,
                                                             const double *__restrict__ psi_rows, int64_t ld, int L,
                                                             const int64_t *__restrict__ seg_list, int64_t n_list,
                                                             double *__restrict__ values, double *__restrict__ scratch, int64_t p2max) {
    extern __shared__ double sh[];
    double *buf = BUF ? scratch + (int64_t)blockIdx.x * p2max : sh;
    for (int64_t idx = blockIdx.x; idx < n_list; idx += gridDim.x) {
        const int64_t seg = seg_list[idx];
        const int64_t src = seg / L;
        const int64_t b = offsets[seg], e = offsets[seg + 1];
        const int len = (int)(e - b);
        int p2 = 1;
        while (p2 < len) p2 <<= 1;
        const double *row = psi_rows + src * ld;
        for (int t = threadIdx.x; t < p2; t += NT) buf[t] = t < len ? row[members[b + t]] : INFINITY;
        __syncthreads();
        for (int k = 2; k <= p2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < p2; t += NT) {
                    const int x = t ^ j;
                    if (x > t) {
                        const double a = buf[t], c = buf[x];
                        const bool up = (t & k) == 0;
                        if ((a > c) == up) { buf[t] = c; buf[x] = a; }
                    }
                }
                __syncthreads();
            }
        for (int t = threadIdx.x; t < len; t += NT) values[b + t] = buf[t];
        __syncthreads();
    }
}

// scipy.stats.wasserstein_distance (_cdf_distance, p = 1) on two ascending samples: walk the merged
// values, integrate |U - V| between consecutive distinct values, U = (#u <= x)/nu
__device__ inline double w1_sorted(const double *__restrict__ u, int nu, const double *__restrict__ v, int nv) {
    int i = 0, j = 0;
    double acc = 0.0;
    double prev = fmin(u[0], v[0]);
    const double inu = (double)nu, inv = (double)nv;
    while (i < nu || j < nv) {
        const bool take_u = (j >= nv) || (i < nu && u[i] <= v[j]);
        const double x = take_u ? u[i] : v[j];
        if (x > prev) {
            acc += fabs((double)i / inu - (double)j / inv) * (x - prev);
            prev = x;
        }
        if (take_u) ++i; else ++j;
    }
    return acc;
}

// Variant A on ragged samples: out[(i - r0) * ld + j] for i in [r0, r1), all j.  One thread per pair.
__global__ __launch_bounds__(256) void pair_ragged_kernel(const int64_t *__restrict__ offsets, const double *__restrict__ values,
                                                         int L, int layer_lo, int layer_hi, int64_t n, int64_t r0, int64_t r1,
                                                         double *__restrict__ out, int64_t ld, int *__restrict__ flag) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = r0 + blockIdx.y;
    if (j >= n || i >= r1) return;
    double d = 0.0;
    if (i != j) {
        // the reference evaluates the pair with the smaller index as the first sample (model/HSD.py:104)
        const int64_t a = i < j ? i : j, b = i < j ? j : i;
        for (int k = layer_lo; k <= layer_hi; ++k) {
            const int64_t ua = offsets[a * L + k], ue = offsets[a * L + k + 1];
            const int64_t va = offsets[b * L + k], ve = offsets[b * L + k + 1];
            if (ue == ua || ve == va) { atomicOr(flag, 1); d = NAN; break; }
            d += w1_sorted(values + ua, (int)(ue - ua), values + va, (int)(ve - va));
        }
    }
    out[(i - r0) * ld + j] = d;
}

// Variant B as written (model/HSD.py:140-161), fallback for rings too large for the wave slots of
// pair_rowquirk_tiled_kernel (heatpair.hip): for the pair (a, b), a < b, layer k < hop:
//   p = psi[a, ring_k(a)],  q = psi[a, ring_k(b)]   (BOTH from row a = startIndex, lines 152/155)
//   d += calculate_distance(p, q, 'wasserstein'): zero-pad to equal length, sort, scipy W1 = mean |P_(r) - Q_(r)|.
// One workgroup per (a, chunk of RQ_B partners): q and its zero pads are sorted together in LDS (bitonic,
// 256 threads), P is the pre-sorted sample of a with its pads placed after its negative values, and the rank
// sum is reduced across the workgroup.  D[a, b] and D[b, a] are written from the one evaluation.
constexpr int RQ_B = 32;
__global__ __launch_bounds__(256) void pair_rowquirk_kernel(const int64_t *__restrict__ offsets, const int32_t *__restrict__ members,
                                                           const double *__restrict__ values /* sorted p samples */,
                                                           const double *__restrict__ psi, int64_t ldpsi, int L, int hop, int64_t n,
                                                           int64_t r0, int64_t r1, double *__restrict__ out, int64_t ld, int64_t a_base) {
    extern __shared__ double sh[];      // q buffer: largest ring rounded up to a power of two
    __shared__ double red[4];
    __shared__ int negcnt, negq;
    const int64_t a = a_base + blockIdx.y;                   // a < r1
    const bool a_row = a >= r0;
    const double *row = psi + a * ldpsi;
    for (int64_t b = (int64_t)blockIdx.x * RQ_B; b < min(n, ((int64_t)blockIdx.x + 1) * RQ_B); ++b) {
        if (b == a && a_row && threadIdx.x == 0) out[(a - r0) * ld + a] = 0.0;
        if (b <= a || (!a_row && (b < r0 || b >= r1))) continue;
        double d = 0.0;
        for (int k = 0; k < hop; ++k) {
            const int64_t pb = offsets[a * L + k], qb = offsets[b * L + k];
            const int np = (int)(offsets[a * L + k + 1] - pb), nq = (int)(offsets[b * L + k + 1] - qb);
            if (np == 0 && nq == 0) continue;                // tools/metrics.py:170-171
            const int Lm = np > nq ? np : nq;
            // only the nq real values are sorted (padded to a power of two with +inf); the zero pads of the
            // alignment are placed analytically: sorted(q + pads) = q's negatives, the pads, the rest (same for p)
            int p2 = 1;
            while (p2 < nq) p2 <<= 1;
            if (threadIdx.x == 0) { negcnt = 0; negq = 0; }
            __syncthreads();
            int mineq = 0;
            for (int t = threadIdx.x; t < p2; t += 256) {
                const double v = t < nq ? row[members[qb + t]] : INFINITY;
                sh[t] = v;
                mineq += v < 0.0;
            }
            if (mineq) atomicAdd(&negq, mineq);
            __syncthreads();
            for (int kk = 2; kk <= p2; kk <<= 1)
                for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                    for (int t = threadIdx.x; t < p2; t += 256) {
                        const int x = t ^ jj;
                        if (x > t) {
                            const double u = sh[t], v = sh[x];
                            const bool up = (t & kk) == 0;
                            if ((u > v) == up) { sh[t] = v; sh[x] = u; }
                        }
                    }
                    __syncthreads();
                }
            const double *p = values + pb;
            int mine = 0;
            for (int t = threadIdx.x; t < np; t += 256) mine += p[t] < 0.0;
            if (mine) atomicAdd(&negcnt, mine);
            __syncthreads();
            const int neg = negcnt, padp = Lm - np, nneg_q = negq, padq = Lm - nq;
            auto P_at = [&](int r) -> double { return r < neg ? p[r] : (r < neg + padp ? 0.0 : p[r - padp]); };
            double acc = 0.0;
            for (int t = threadIdx.x; t < nq; t += 256) acc += fabs(P_at(t < nneg_q ? t : t + padq) - sh[t]);
            for (int r = nneg_q + threadIdx.x; r < nneg_q + padq; r += 256) acc += fabs(P_at(r));    // ranks held by q's pads
            for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
            __syncthreads();
            d += (red[0] + red[1] + red[2] + red[3]) / (double)Lm;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (a_row) out[(a - r0) * ld + b] = d;
            if (b >= r0 && b < r1) out[(b - r0) * ld + a] = d;
        }
    }
}


// tools.metrics.calculate_distance(p, q, metric) for ONE pair of samples (tools/metrics.py:151-192):
// zero-pad the shorter sample (:26-28), sort both ascending (:29-30), then
//   metric 0  scipy.stats.wasserstein_distance on the aligned arrays
//   metric 1  KL_divergence(p, q)            (:74-88; precondition |sum - 1| <= 1e-4 on both, else status 1)
//   metric 2  (KL(p||q) + KL(q||p)) / 2
// One workgroup; both padded samples live in LDS (2 * p2 doubles).
__global__ __launch_bounds__(256) void sample_distance_kernel(const double *__restrict__ p, int np, const double *__restrict__ q,
                                                             int nq, int metric, double *__restrict__ out, int *__restrict__ status) {
    extern __shared__ double sh[];
    const int Lm = np > nq ? np : nq;
    int p2 = 1;
    while (p2 < Lm) p2 <<= 1;
    double *sp = sh, *sq = sh + p2;
    for (int t = threadIdx.x; t < p2; t += 256) {
        sp[t] = t < np ? p[t] : (t < Lm ? 0.0 : INFINITY);
        sq[t] = t < nq ? q[t] : (t < Lm ? 0.0 : INFINITY);
    }
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < p2; t += 256) {
                const int x = t ^ j;
                if (x > t) {
                    const bool up = (t & k) == 0;
                    double a = sp[t], b = sp[x];
                    if ((a > b) == up) { sp[t] = b; sp[x] = a; }
                    a = sq[t]; b = sq[x];
                    if ((a > b) == up) { sq[t] = b; sq[x] = a; }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x != 0) return;
    *status = 0;
    if (Lm == 0) { *out = 0.0; return; }                     // tools/metrics.py:170-171
    if (metric == 0) {
        *out = w1_sorted(sp, Lm, sq, Lm);
        return;
    }
    double s1 = 0.0, s2 = 0.0;
    for (int t = 0; t < Lm; ++t) { s1 += sp[t]; s2 += sq[t]; }
    if (!(fabs(s1 - 1.0) <= 1e-4) || !(fabs(s2 - 1.0) <= 1e-4)) { *status = 1; *out = NAN; return; }
    double kl_pq = 0.0, kl_qp = 0.0;
    for (int t = 0; t < Lm; ++t) {
        kl_pq += sp[t] * log(sp[t] / sq[t]);
        kl_qp += sq[t] * log(sq[t] / sp[t]);
    }
    *out = metric == 1 ? kl_pq : (kl_pq + kl_qp) / 2.0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// One step of the recurrence on a node-major block of Cb columns: long rows first (partial products per
// chunk), then one workgroup per row.
static int launch_cheby_step(hsd_ctx *ctx, int Cb, const double *Tc, const double *To, double *Tn, double *R, double alpha,
                             double beta, double gamma, double ck, int first) {
    const dim3 blk(std::min(256, (Cb + 63) / 64 * 64));
    if (ctx->n_lchunks > 0) {
        if (ctx->lpartial_cols < Cb) {
            HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            hsd_dev_free(ctx->d_lpartial);
            int rc = hsd_dev_alloc(ctx, &ctx->d_lpartial, (size_t)ctx->n_lchunks * Cb);
            if (rc) return rc;
            ctx->lpartial_cols = Cb;
        }
        hipLaunchKernelGGL(cheby_partial_kernel, dim3((unsigned)ctx->n_lchunks), blk, 0, ctx->stream, ctx->d_lrowptr, ctx->d_lcol,
                           ctx->d_lval, Cb, Tc, ctx->d_lchunk_row, ctx->d_lchunk_p0, ctx->d_lpartial);
    }
    hipLaunchKernelGGL(cheby_step_kernel, dim3((unsigned)ctx->n), blk, 0, ctx->stream, ctx->d_lrowptr, ctx->d_lcol, ctx->d_lval, ctx->n,
                       Cb, Tc, To, Tn, R, alpha, beta, gamma, ck, first, ctx->n_lchunks > 0 ? ctx->d_lrow_first : nullptr,
                       ctx->d_lpartial);
    return HSD_OK;
}

static void release_heat(hsd_ctx *ctx) {
    hsd_dev_free(ctx->d_lrow_first);
    hsd_dev_free(ctx->d_lchunk_row);
    hsd_dev_free(ctx->d_lchunk_p0);
    hsd_dev_free(ctx->d_lpartial);
    ctx->n_lchunks = 0;
    ctx->lpartial_cols = 0;
    hsd_dev_free(ctx->d_lrowptr);
    hsd_dev_free(ctx->d_lcol);
    hsd_dev_free(ctx->d_lval);
    hsd_dev_free(ctx->d_hoffsets);
    hsd_dev_free(ctx->d_hmembers);
    hsd_dev_free(ctx->d_hvalues);
    ctx->heat_hop = -1;
}

extern "C" void hsd_heat_release_internal(hsd_ctx *ctx) { release_heat(ctx); }

extern "C" int hsd_set_laplacian(hsd_ctx *ctx, int64_t n, const int64_t *rowptr, const int32_t *col, const double *val) try {
    HSD_CHECK_CTX(ctx);
    if (ctx->n <= 0) return hsd_fail(ctx, HSD_ERR_NO_GRAPH, "upload the graph first");
    if (n != ctx->n || !rowptr || !col || !val) return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_set_laplacian: bad arguments");
    const int64_t nnz = rowptr[n];
    for (int64_t e = 0; e < nnz; ++e)
        if (col[e] < 0 || col[e] >= n) return hsd_fail(ctx, HSD_ERR_INVALID, "Laplacian column out of range");
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hsd_dev_free(ctx->d_lrowptr);
    hsd_dev_free(ctx->d_lcol);
    hsd_dev_free(ctx->d_lval);
    int rc;
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_lrowptr, (size_t)n + 1))) return rc;
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_lcol, (size_t)nnz))) return rc;
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_lval, (size_t)nnz))) return rc;
    HSD_HIP(ctx, hipMemcpyAsync(ctx->d_lrowptr, rowptr, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    HSD_HIP(ctx, hipMemcpyAsync(ctx->d_lcol, col, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HSD_HIP(ctx, hipMemcpyAsync(ctx->d_lval, val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    // chunk lists of the long rows
    hsd_dev_free(ctx->d_lrow_first);
    hsd_dev_free(ctx->d_lchunk_row);
    hsd_dev_free(ctx->d_lchunk_p0);
    hsd_dev_free(ctx->d_lpartial);
    ctx->n_lchunks = 0;
    ctx->lpartial_cols = 0;
    std::vector<int32_t> row_first((size_t)n, -1), chunk_row;
    std::vector<int64_t> chunk_p0;
    const char *el = hsd_hook(H_CHEBY_LONGROWS);
    if (!(el && atoi(el) == 0))
        for (int64_t i = 0; i < n; ++i) {
            const int64_t len = rowptr[i + 1] - rowptr[i];
            if (len <= 2 * HSD_LCHUNK) continue;
            row_first[(size_t)i] = (int32_t)chunk_row.size();
            for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p += HSD_LCHUNK) {
                chunk_row.push_back((int32_t)i);
                chunk_p0.push_back(p);
            }
        }
    if (!chunk_row.empty()) {
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_lrow_first, (size_t)n))) return rc;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_lchunk_row, chunk_row.size()))) return rc;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_lchunk_p0, chunk_p0.size()))) return rc;
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_lrow_first, row_first.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_lchunk_row, chunk_row.data(), chunk_row.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_lchunk_p0, chunk_p0.data(), chunk_p0.size() * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        ctx->n_lchunks = (int)chunk_row.size();
    }
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_heat_wavelets(hsd_ctx *ctx, const double *coeff, int order, double lmax, double threshold, int64_t c0,
                                 int64_t c1, double *out, int64_t ld, int out_on_device) try {
    HSD_CHECK_CTX(ctx);
    if (!ctx->d_lrowptr) return hsd_fail(ctx, HSD_ERR_NO_GRAPH, "hsd_set_laplacian has not been called");
    const int64_t n = ctx->n;
    if (!coeff || order < 1 || !(lmax > 0.0)) return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_heat_wavelets: bad coefficients/order/lmax");
    if (c0 < 0 || c1 > n || c0 > c1) return hsd_fail(ctx, HSD_ERR_INVALID, "column range out of bounds");
    if (c0 == c1) return HSD_OK;
    if (!out || ld < n) return hsd_fail(ctx, HSD_ERR_INVALID, "null out or ld < n");
    // impulse block width: bounded by ~3 GiB of working set (3 recurrence blocks + result)
    int64_t C = std::min<int64_t>(c1 - c0, std::max<int64_t>(32, (((int64_t)3 << 30) / (4 * n * (int64_t)sizeof(double))) / 32 * 32));
    C = std::min<int64_t>(C, 1024);
    double *T[3] = {nullptr, nullptr, nullptr}, *R = nullptr, *d_out_stage = nullptr;
    int rc = HSD_OK;
    for (int t = 0; t < 3 && rc == HSD_OK; ++t) rc = hsd_dev_alloc(ctx, &T[t], (size_t)n * C);
    if (rc == HSD_OK) rc = hsd_dev_alloc(ctx, &R, (size_t)n * C);
    if (rc == HSD_OK && !out_on_device) rc = hsd_dev_alloc(ctx, &d_out_stage, (size_t)C * n);
    const double a1 = lmax / 2.0, a2 = lmax / 2.0;
    for (int64_t b0 = c0; b0 < c1 && rc == HSD_OK; b0 += C) {
        const int Cb = (int)std::min<int64_t>(C, c1 - b0);
        const int64_t cnt = n * Cb;
        hipLaunchKernelGGL(cheby_init_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, n, Cb, b0,
                           0.5 * coeff[0], T[0], R);
        // T1 = (L T0 - a2 T0) / a1 ; R += c1 T1
        rc = launch_cheby_step(ctx, Cb, T[0], T[0], T[1], R, 1.0 / a1, -a2 / a1, 0.0, coeff[1], 1);
        int cur = 1, old = 0;
        for (int k = 2; k <= order; ++k) {
            const int nw = 3 - cur - old;
            // T_k = (2/a1)(L T_{k-1} - a2 T_{k-1}) - T_{k-2} ; R += c_k T_k
            if (rc == HSD_OK) rc = launch_cheby_step(ctx, Cb, T[cur], T[old], T[nw], R, 2.0 / a1, -2.0 * a2 / a1, -1.0, coeff[k], 0);
            old = cur;
            cur = nw;
        }
        double *dst = out_on_device ? out + (b0 - c0) * ld : d_out_stage;
        const int64_t dld = out_on_device ? ld : n;
        hipLaunchKernelGGL(transpose_threshold_kernel, dim3((unsigned)((n + 31) / 32), (unsigned)((Cb + 31) / 32)), dim3(32, 8), 0,
                           ctx->stream, R, n, Cb, Cb, threshold, threshold >= 0.0 ? 1 : 0, dst, dld);
        if (!out_on_device) {
            hipError_t e = hipMemcpy2DAsync(out + (b0 - c0) * ld, (size_t)ld * sizeof(double), d_out_stage, (size_t)n * sizeof(double),
                                            (size_t)n * sizeof(double), (size_t)Cb, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_heat_wavelets: %s", hipGetErrorString(e));
        }
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (rc == HSD_OK && e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_heat_wavelets: %s", hipGetErrorString(e));
    if (rc == HSD_OK && (e = hipGetLastError()) != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_heat_wavelets: %s", hipGetErrorString(e));
    for (int t = 0; t < 3; ++t) hsd_dev_free(T[t]);
    hsd_dev_free(R);
    hsd_dev_free(d_out_stage);
    return rc;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_heat_wavelets_multi(hsd_ctx *ctx, const double *coeff /* [n_scales][order+1] */, int n_scales, int order, double lmax,
                                       double threshold, double *const *outs, int64_t ld) try {
    HSD_CHECK_CTX(ctx);
    if (!ctx->d_lrowptr) return hsd_fail(ctx, HSD_ERR_NO_GRAPH, "hsd_set_laplacian has not been called");
    const int64_t n = ctx->n;
    if (!coeff || order < 1 || !(lmax > 0.0) || n_scales < 1 || n_scales > 4096 || !outs || ld < n)
        return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_heat_wavelets_multi: bad arguments");
    for (int s = 0; s < n_scales; ++s)
        if (!outs[s]) return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_heat_wavelets_multi: null output %d", s);
    const int S = n_scales, SP = std::min(S, 8);
    // every plane T_0..T_order of a block of impulses is kept and combined per scale (see hsd_multiscale_ring_stats);
    // if the planes do not fit, one recurrence per scale
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const int64_t budget = (int64_t)std::min<size_t>(free_b / 2, (size_t)48 << 30);
    int64_t C = (budget / ((int64_t)(order + 1 + SP) * n * (int64_t)sizeof(double))) / 32 * 32;
    if (C < 64) {
        for (int s = 0; s < S; ++s) {
            const int rc = hsd_heat_wavelets(ctx, coeff + (size_t)s * (order + 1), order, lmax, threshold, 0, n, outs[s], ld, 1);
            if (rc) return rc;
        }
        return HSD_OK;
    }
    C = std::min<int64_t>({C, 1024, (n + 31) / 32 * 32});
    const int64_t stride = n * C;
    double *Tall = nullptr, *R = nullptr, *d_coef = nullptr;
    std::vector<double> h_coef((size_t)(order + 1) * S);          // [k][s], k = 0 halved
    for (int k = 0; k <= order; ++k)
        for (int s = 0; s < S; ++s) h_coef[(size_t)k * S + s] = (k == 0 ? 0.5 : 1.0) * coeff[(size_t)s * (order + 1) + k];
    int rc = hsd_dev_alloc(ctx, &Tall, (size_t)(order + 1) * stride);
    if (rc == HSD_OK) rc = hsd_dev_alloc(ctx, &R, (size_t)SP * stride);
    if (rc == HSD_OK) rc = hsd_dev_alloc(ctx, &d_coef, h_coef.size());
    if (rc == HSD_OK) {
        hipError_t e = hipMemcpyAsync(d_coef, h_coef.data(), h_coef.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_heat_wavelets_multi: %s", hipGetErrorString(e));
    }
    const double a1 = lmax / 2.0, a2 = lmax / 2.0;
    auto plane = [&](int t) { return Tall + (size_t)t * stride; };
    for (int64_t b0 = 0; b0 < n && rc == HSD_OK; b0 += C) {
        const int Cb = (int)std::min<int64_t>(C, n - b0);
        const int64_t cnt = n * Cb;
        const unsigned gcnt = (unsigned)((cnt + 255) / 256);
        hipLaunchKernelGGL(cheby_init_kernel, dim3(gcnt), dim3(256), 0, ctx->stream, n, Cb, b0, 0.0, plane(0), R);
        rc = launch_cheby_step(ctx, Cb, plane(0), plane(0), plane(1), nullptr, 1.0 / a1, -a2 / a1, 0.0, 0.0, 1);
        for (int k = 2; k <= order && rc == HSD_OK; ++k)
            rc = launch_cheby_step(ctx, Cb, plane(k - 1), plane(k - 2), plane(k), nullptr, 2.0 / a1, -2.0 * a2 / a1, -1.0, 0.0, 0);
        if (rc) break;
        for (int s0 = 0; s0 < S; s0 += SP) {
            const int sc = std::min(SP, S - s0);
            hipLaunchKernelGGL(cheby_combine_kernel<8>, dim3(gcnt), dim3(256), 0, ctx->stream, Tall, stride, order, d_coef + s0, S, sc, cnt, R);
            for (int s = 0; s < sc; ++s)
                hipLaunchKernelGGL(transpose_threshold_kernel, dim3((unsigned)((n + 31) / 32), (unsigned)((Cb + 31) / 32)), dim3(32, 8), 0,
                                   ctx->stream, R + (size_t)s * cnt, n, Cb, Cb, threshold, threshold >= 0.0 ? 1 : 0,
                                   outs[s0 + s] + b0 * ld, ld);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_heat_wavelets_multi: %s", hipGetErrorString(e));
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (rc == HSD_OK && e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_heat_wavelets_multi: %s", hipGetErrorString(e));
    hsd_dev_free(Tall);
    hsd_dev_free(R);
    hsd_dev_free(d_coef);
    return rc;
} HSD_ABI_GUARD(ctx)

// ragged rings of all nodes resident on the device (offsets + members), shared by the calls below
static int ensure_rings(hsd_ctx *ctx, int hop) {
    if (ctx->heat_hop == hop && ctx->d_hoffsets) return HSD_OK;
    const int64_t n = ctx->n;
    const int L = hop + 1;
    hsd_dev_free(ctx->d_hoffsets);
    hsd_dev_free(ctx->d_hmembers);
    hsd_dev_free(ctx->d_hvalues);
    ctx->heat_hop = -1;
    int32_t *d_sizes = nullptr;
    int rc = hsd_dev_alloc(ctx, &d_sizes, (size_t)n * L);
    if (rc) return rc;
    rc = hsd_launch_khop(ctx, hop, 0, n, false, d_sizes, nullptr, nullptr);
    std::vector<int32_t> sizes((size_t)n * L);
    if (rc == HSD_OK) {
        hipError_t e = hipMemcpyAsync(sizes.data(), d_sizes, sizes.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "ensure_rings: %s", hipGetErrorString(e));
    }
    if (rc) { hsd_dev_free(d_sizes); return rc; }
    ctx->h_hoffsets.assign((size_t)n * L + 1, 0);
    int32_t max_seg = 0;
    for (size_t s = 0; s < sizes.size(); ++s) {
        ctx->h_hoffsets[s + 1] = ctx->h_hoffsets[s] + sizes[s];
        max_seg = std::max(max_seg, sizes[s]);
    }
    ctx->heat_max_seg = max_seg;
    const int64_t total = ctx->h_hoffsets.back();
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_hoffsets, (size_t)n * L + 1)) == HSD_OK) rc = hsd_dev_alloc(ctx, &ctx->d_hmembers, (size_t)total);
    if (rc == HSD_OK) {
        hipError_t e = hipMemcpyAsync(ctx->d_hoffsets, ctx->h_hoffsets.data(), ctx->h_hoffsets.size() * sizeof(int64_t),
                                      hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "ensure_rings: %s", hipGetErrorString(e));
    }
    if (rc == HSD_OK) rc = hsd_launch_khop(ctx, hop, 0, n, false, d_sizes, ctx->d_hoffsets, ctx->d_hmembers);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (rc == HSD_OK && e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "ensure_rings: %s", hipGetErrorString(e));
    hsd_dev_free(d_sizes);
    if (rc == HSD_OK) ctx->heat_hop = hop;
    return rc;
}

extern "C" int hsd_ring_stats(hsd_ctx *ctx, int hop, const double *psi_rows_dev, int64_t ld, int64_t r0, int64_t r1, double *sums,
                              int32_t *sizes) try {
    HSD_CHECK_CTX(ctx);
    if (ctx->n <= 0) return hsd_fail(ctx, HSD_ERR_NO_GRAPH, "no graph uploaded");
    if (hop < 0 || hop >= HSD_MAX_LAYERS || r0 < 0 || r1 > ctx->n || r0 > r1 || !psi_rows_dev || ld < ctx->n || !sums)
        return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_ring_stats: bad arguments");
    if (r0 == r1) return HSD_OK;
    int rc = ensure_rings(ctx, hop);
    if (rc) return rc;
    const int L = hop + 1;
    const int64_t nseg = (r1 - r0) * L;
    double *d_sums = nullptr;
    if ((rc = hsd_dev_alloc(ctx, &d_sums, (size_t)nseg))) return rc;
    // psi_rows_dev holds rows r0..r1 (row 0 <-> source r0): shift the source index
    hipLaunchKernelGGL(ring_sum_kernel, dim3((unsigned)nseg), dim3(256), 0, ctx->stream, ctx->d_hoffsets + r0 * L, ctx->d_hmembers,
                       psi_rows_dev, ld, L, d_sums);
    hipError_t e = hipMemcpyAsync(sums, d_sums, (size_t)nseg * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hsd_dev_free(d_sums);
    if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "hsd_ring_stats: %s", hipGetErrorString(e));
    if (sizes)
        for (int64_t s = 0; s < nseg; ++s) sizes[s] = (int32_t)(ctx->h_hoffsets[(size_t)(r0 * L + s + 1)] - ctx->h_hoffsets[(size_t)(r0 * L + s)]);
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_multiscale_ring_stats(hsd_ctx *ctx, int hop, const double *coeff /* [n_scales][order+1] */, int n_scales,
                                         int order, double lmax, double threshold, int64_t r0, int64_t r1,
                                         double *sums /* host [n_scales][r1-r0][hop+1] */, int32_t *sizes /* host or NULL */) try {
    HSD_CHECK_CTX(ctx);
    if (!ctx->d_lrowptr) return hsd_fail(ctx, HSD_ERR_NO_GRAPH, "hsd_set_laplacian has not been called");
    const int64_t n = ctx->n;
    if (!coeff || order < 1 || !(lmax > 0.0) || n_scales < 1 || n_scales > 4096 || !sums || hop < 0 || hop >= HSD_MAX_LAYERS)
        return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_multiscale_ring_stats: bad arguments (1 <= n_scales <= 4096 per call)");
    if (r0 < 0 || r1 > n || r0 > r1) return hsd_fail(ctx, HSD_ERR_INVALID, "source range out of bounds");
    if (r0 == r1) return HSD_OK;
    int rc = ensure_rings(ctx, hop);
    if (rc) return rc;
    const int L = hop + 1, S = n_scales, SP = std::min(S, 32);          // scales per pass of the combine / fused update
    const int64_t nsrc = r1 - r0;
    // Two schedules.  "store": keep every T_k of the block (order + 1 planes) and combine the scales, 32 per pass,
    // at the end -- per element and step 8 B written instead of S accumulators read and written (16 S B), and the
    // recurrence runs ONCE however many scales there are (main.py:36 uses 200); needs (order + 1 + 32) planes of
    // n * C doubles.  "fused": three rolling planes and R_s updated at every step, one recurrence per 32 scales,
    // for series too long to store (approx=False on a wide spectrum).  Same arithmetic, same results.
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const int64_t budget = (int64_t)std::min<size_t>(free_b / 2, (size_t)48 << 30);
    int64_t C_store = (budget / ((int64_t)(order + 1 + SP) * n * (int64_t)sizeof(double))) / 32 * 32;
    bool store = C_store >= 64;
    if (const char *es = hsd_hook(H_CHEBY_STORE)) store = store && atoi(es) != 0;
    int64_t C = store ? std::min<int64_t>(512, C_store)
                      : std::max<int64_t>(32, std::min<int64_t>(512, (((int64_t)4 << 30) / ((3 + SP) * n * (int64_t)sizeof(double))) / 32 * 32));
    if (const char *ec = hsd_hook(H_CHEBY_C)) C = std::max<int64_t>(32, std::min<int64_t>(C, atoll(ec) / 32 * 32));   // measurement hook
    C = std::min<int64_t>(C, (nsrc + 31) / 32 * 32);
    const int n_planes = store ? order + 1 : 3;
    double *Tall = nullptr, *R = nullptr, *d_sums = nullptr, *d_coef = nullptr;
    const int64_t stride = n * C;
    rc = hsd_dev_alloc(ctx, &Tall, (size_t)n_planes * stride);
    if (rc == HSD_OK) rc = hsd_dev_alloc(ctx, &R, (size_t)SP * n * C);
    if (rc == HSD_OK) rc = hsd_dev_alloc(ctx, &d_sums, (size_t)S * nsrc * L);
    std::vector<double> h_coef((size_t)(order + 1) * S);          // [k][s], k = 0 halved
    for (int k = 0; k <= order; ++k)
        for (int s = 0; s < S; ++s) h_coef[(size_t)k * S + s] = (k == 0 ? 0.5 : 1.0) * coeff[(size_t)s * (order + 1) + k];
    if (rc == HSD_OK && store) {
        rc = hsd_dev_alloc(ctx, &d_coef, h_coef.size());
        if (rc == HSD_OK) {
            hipError_t e = hipMemcpyAsync(d_coef, h_coef.data(), h_coef.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_multiscale_ring_stats: %s", hipGetErrorString(e));
        }
    }
    const double a1 = lmax / 2.0, a2 = lmax / 2.0;
    auto ck = [&](int k, int s0, int sc) {
        CoefS c{};
        for (int s = 0; s < sc; ++s) c.c[s] = h_coef[(size_t)k * S + s0 + s];
        return c;
    };
    auto plane = [&](int t) { return Tall + (size_t)t * stride; };
    for (int64_t b0 = r0; b0 < r1 && rc == HSD_OK; b0 += C) {
        const int Cb = (int)std::min<int64_t>(C, r1 - b0);
        const int64_t cnt = n * Cb;
        const dim3 blk(std::min(256, (Cb + 63) / 64 * 64));
        const unsigned gcnt = (unsigned)((cnt + 255) / 256);
        if (store) {
            // T_k -> plane k; R is only scratch for the init kernel until the combine overwrites it
            hipLaunchKernelGGL(cheby_init_kernel, dim3(gcnt), dim3(256), 0, ctx->stream, n, Cb, b0, 0.0, plane(0), R);
            rc = launch_cheby_step(ctx, Cb, plane(0), plane(0), plane(1), nullptr, 1.0 / a1, -a2 / a1, 0.0, 0.0, 1);
            for (int k = 2; k <= order && rc == HSD_OK; ++k)
                rc = launch_cheby_step(ctx, Cb, plane(k - 1), plane(k - 2), plane(k), nullptr, 2.0 / a1, -2.0 * a2 / a1, -1.0, 0.0, 0);
            if (rc) break;
        }
        for (int s0 = 0; s0 < S; s0 += SP) {
            const int sc = std::min(SP, S - s0);
            if (store) {
                if (sc <= 8)
                    hipLaunchKernelGGL(cheby_combine_kernel<8>, dim3(gcnt), dim3(256), 0, ctx->stream, Tall, stride, order, d_coef + s0, S, sc, cnt, R);
                else
                    hipLaunchKernelGGL(cheby_combine_kernel<32>, dim3(gcnt), dim3(256), 0, ctx->stream, Tall, stride, order, d_coef + s0, S, sc, cnt, R);
            } else {
                hipLaunchKernelGGL(cheby_init_multi_kernel, dim3(gcnt), dim3(256), 0, ctx->stream, n, Cb, b0, ck(0, s0, sc), sc, plane(0), R);
                hipLaunchKernelGGL(cheby_step_multi_kernel, dim3((unsigned)n), blk, 0, ctx->stream, ctx->d_lrowptr, ctx->d_lcol, ctx->d_lval,
                                   n, Cb, plane(0), plane(0), plane(1), R, 1.0 / a1, -a2 / a1, 0.0, ck(1, s0, sc), sc, 1);
                int cur = 1, old = 0;
                for (int k = 2; k <= order; ++k) {
                    const int nw = 3 - cur - old;
                    hipLaunchKernelGGL(cheby_step_multi_kernel, dim3((unsigned)n), blk, 0, ctx->stream, ctx->d_lrowptr, ctx->d_lcol,
                                       ctx->d_lval, n, Cb, plane(cur), plane(old), plane(nw), R, 2.0 / a1, -2.0 * a2 / a1, -1.0,
                                       ck(k, s0, sc), sc, 0);
                    old = cur;
                    cur = nw;
                }
            }
            hipLaunchKernelGGL(ring_sum_nodemajor_kernel, dim3((unsigned)(Cb * L), (unsigned)sc), dim3(256), 0, ctx->stream,
                               ctx->d_hoffsets + b0 * L, ctx->d_hmembers, R, n, Cb, L, threshold, threshold >= 0.0 ? 1 : 0,
                               d_sums + (size_t)s0 * nsrc * L, nsrc, b0 - r0);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_multiscale_ring_stats: %s", hipGetErrorString(e));
    }
    if (rc == HSD_OK) {
        hipError_t e = hipMemcpyAsync(sums, d_sums, (size_t)S * nsrc * L * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_multiscale_ring_stats: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(ctx->stream);
    hsd_dev_free(Tall);
    hsd_dev_free(R);
    hsd_dev_free(d_sums);
    hsd_dev_free(d_coef);
    if (rc == HSD_OK && sizes)
        for (int64_t s = 0; s < nsrc * L; ++s)
            sizes[s] = (int32_t)(ctx->h_hoffsets[(size_t)(r0 * L + s + 1)] - ctx->h_hoffsets[(size_t)(r0 * L + s)]);
    return rc;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_heat_signature_build(hsd_ctx *ctx, int hop, const double *psi_dev, int64_t ld) try {
    HSD_CHECK_CTX(ctx);
    if (ctx->n <= 0) return hsd_fail(ctx, HSD_ERR_NO_GRAPH, "no graph uploaded");
    if (hop < 0 || hop >= HSD_MAX_LAYERS || !psi_dev || ld < ctx->n) return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_heat_signature_build: bad arguments");
    int rc = ensure_rings(ctx, hop);
    if (rc) return rc;
    const int L = hop + 1;
    hsd_dev_free(ctx->d_hvalues);
    const int64_t total = ctx->h_hoffsets.back();
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_hvalues, (size_t)total))) return rc;
    // segments by length class: <= 2048 values (16 KB of LDS, many workgroups per CU), up to what one CU's LDS
    // holds (16384 values), and longer ones through global scratch
    const size_t cap = ctx->prop.maxSharedMemoryPerMultiProcessor ? ctx->prop.maxSharedMemoryPerMultiProcessor : 65536;
    int64_t lds_max_len = 1;
    while (lds_max_len * 2 * (int64_t)sizeof(double) <= (int64_t)cap) lds_max_len <<= 1;
    const char *eg = hsd_hook(H_HEAT_SORT_LDS);       // test hook: smaller LDS classes
    if (eg && atoll(eg) >= 2) lds_max_len = std::min<int64_t>(lds_max_len, atoll(eg));
    const int64_t small_len = std::min<int64_t>(2048, lds_max_len);
    std::vector<int64_t> cls[3];
    int64_t big_max = 0;
    const int64_t nseg = ctx->n * L;
    for (int64_t sgi = 0; sgi < nseg; ++sgi) {
        const int64_t len = ctx->h_hoffsets[(size_t)sgi + 1] - ctx->h_hoffsets[(size_t)sgi];
        if (len <= 0) continue;
        if (len <= small_len) cls[0].push_back(sgi);
        else if (len <= lds_max_len) cls[1].push_back(sgi);
        else { cls[2].push_back(sgi); big_max = std::max(big_max, len); }
    }
    int64_t *d_list = nullptr;
    double *d_scr = nullptr;
    const size_t n_all = cls[0].size() + cls[1].size() + cls[2].size();
    if ((rc = hsd_dev_alloc(ctx, &d_list, n_all))) return rc;
    hipError_t e = hipSuccess;
    size_t at = 0;
    for (int c = 0; c < 3 && e == hipSuccess; ++c) {
        if (cls[c].empty()) continue;
        e = hipMemcpyAsync(d_list + at, cls[c].data(), cls[c].size() * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) break;
        const int64_t cnt = (int64_t)cls[c].size();
        if (c < 2) {
            const size_t lds = (size_t)(c == 0 ? small_len : lds_max_len) * sizeof(double);
            auto kern = ring_gather_sort_kernel<0, 256>;
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess)
                hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(cnt, 1 << 20)), dim3(256), lds, ctx->stream, ctx->d_hoffsets,
                                   ctx->d_hmembers, psi_dev, ld, L, d_list + at, cnt, ctx->d_hvalues, (double *)nullptr, (int64_t)0);
        } else {
            int64_t p2max = 1;
            while (p2max < big_max) p2max <<= 1;
            const int64_t grid = std::min<int64_t>(cnt, 512);
            rc = hsd_dev_alloc(ctx, &d_scr, (size_t)(grid * p2max));
            if (rc) break;
            hipLaunchKernelGGL((ring_gather_sort_kernel<1, 1024>), dim3((unsigned)grid), dim3(1024), 0, ctx->stream, ctx->d_hoffsets,
                               ctx->d_hmembers, psi_dev, ld, L, d_list + at, cnt, ctx->d_hvalues, d_scr, p2max);
        }
        at += cls[c].size();
    }
    if (e == hipSuccess) e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(ctx->stream);          // the host lists stay alive until the copies are done
    hsd_dev_free(d_list);
    hsd_dev_free(d_scr);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "hsd_heat_signature_build: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

static int pair_heat_common(hsd_ctx *ctx, int64_t r0, int64_t r1, double *out, int64_t ld, int out_on_device, double **dst, int64_t *dld) {
    if (!ctx->d_hvalues) return hsd_fail(ctx, HSD_ERR_NO_SIGNATURE, "hsd_heat_signature_build has not been called");
    if (r0 < 0 || r1 > ctx->n || r0 > r1 || !out || ld < ctx->n) return hsd_fail(ctx, HSD_ERR_INVALID, "bad row range / output");
    if (out_on_device) { *dst = out; *dld = ld; return HSD_OK; }
    const size_t need = (size_t)(r1 - r0) * ctx->n * sizeof(double);
    if (ctx->stage_bytes < need) {
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hsd_dev_free(ctx->d_stage);
        int rc = hsd_dev_alloc(ctx, &ctx->d_stage, need / sizeof(double));
        if (rc) return rc;
        ctx->stage_bytes = need;
    }
    *dst = ctx->d_stage;
    *dld = ctx->n;
    return HSD_OK;
}

static int pair_heat_finish(hsd_ctx *ctx, int64_t r0, int64_t r1, double *out, int64_t ld, int out_on_device, const double *dst) {
    if (!out_on_device) {
        HSD_HIP(ctx, hipMemcpy2DAsync(out, (size_t)ld * sizeof(double), dst, (size_t)ctx->n * sizeof(double), (size_t)ctx->n * sizeof(double),
                                      (size_t)(r1 - r0), hipMemcpyDeviceToHost, ctx->stream));
    }
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    HSD_HIP(ctx, hipGetLastError());
    return HSD_OK;
}

extern "C" int hsd_pair_heat_rows(hsd_ctx *ctx, int layer_lo, int layer_hi, int64_t r0, int64_t r1, double *out, int64_t ld,
                                  int out_on_device) try {
    HSD_CHECK_CTX(ctx);
    if (r0 == r1) return HSD_OK;
    double *dst;
    int64_t dld;
    int rc = pair_heat_common(ctx, r0, r1, out, ld, out_on_device, &dst, &dld);
    if (rc) return rc;
    if (layer_lo < 0 || layer_hi > ctx->heat_hop || layer_lo > layer_hi) return hsd_fail(ctx, HSD_ERR_INVALID, "bad layer range");
    HSD_HIP(ctx, hipMemsetAsync(ctx->d_flag, 0, 4 * sizeof(int), ctx->stream));
    // rank-axis tiled kernel (heatpair.hip); HSD_HEAT_TILED=0 keeps the value-axis merge, one thread per pair
    const char *et = hsd_hook(H_HEAT_TILED);
    rc = (et && atoi(et) == 0) ? HSD_ERR_UNSUPPORTED : hsd_launch_pair_ragged_tiled(ctx, layer_lo, layer_hi, r0, r1, dst, dld);
    if (rc != HSD_OK && rc != HSD_ERR_UNSUPPORTED) return rc;
    for (int64_t ra = r0; rc == HSD_ERR_UNSUPPORTED && ra < r1; ra += 32768) {       // gridDim.y limit
        const int64_t re = std::min(r1, ra + 32768);
        hipLaunchKernelGGL(pair_ragged_kernel, dim3((unsigned)((ctx->n + 255) / 256), (unsigned)(re - ra)), dim3(256), 0, ctx->stream,
                           ctx->d_hoffsets, ctx->d_hvalues, ctx->heat_hop + 1, layer_lo, layer_hi, ctx->n, ra, re,
                           dst + (ra - r0) * dld, dld, ctx->d_flag);
    }
    int flag[4] = {0, 0, 0, 0};
    HSD_HIP(ctx, hipMemcpyAsync(flag, ctx->d_flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (flag[0] & 1) return hsd_fail(ctx, HSD_ERR_EMPTY_RING, "Distribution can't be empty.");
    return pair_heat_finish(ctx, r0, r1, out, ld, out_on_device, dst);
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_pair_heat_rowform(hsd_ctx *ctx, const double *psi_dev, int64_t ldpsi, int64_t r0, int64_t r1, double *out,
                                     int64_t ld, int out_on_device) try {
    HSD_CHECK_CTX(ctx);
    if (r0 == r1) return HSD_OK;
    if (!psi_dev || ldpsi < ctx->n) return hsd_fail(ctx, HSD_ERR_INVALID, "bad psi");
    double *dst;
    int64_t dld;
    int rc = pair_heat_common(ctx, r0, r1, out, ld, out_on_device, &dst, &dld);
    if (rc) return rc;
    // one row of psi per workgroup, one partner per wave (heatpair.hip); HSD_HEAT_TILED=0 or rings too large
    // for a wave slot: one workgroup per pair
    const char *et = hsd_hook(H_HEAT_TILED);
    rc = (et && atoi(et) == 0) ? HSD_ERR_UNSUPPORTED : hsd_launch_pair_rowquirk_tiled(ctx, psi_dev, ldpsi, r0, r1, dst, dld);
    if (rc != HSD_OK && rc != HSD_ERR_UNSUPPORTED) return rc;
    if (rc == HSD_ERR_UNSUPPORTED) {
        int p2 = 1;
        while (p2 < ctx->heat_max_seg) p2 <<= 1;
        const size_t lds = (size_t)p2 * sizeof(double);
        const size_t cap = ctx->prop.maxSharedMemoryPerMultiProcessor ? ctx->prop.maxSharedMemoryPerMultiProcessor : 65536;
        if (lds + 256 > cap)
            return hsd_fail(ctx, HSD_ERR_TOO_LARGE, "largest ring has %d members: sorting it needs %zu B of LDS (CU has %zu)", ctx->heat_max_seg, lds, cap);
        HSD_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pair_rowquirk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int64_t a0 = 0; a0 < r1; a0 += 32768) {         // gridDim.y limit
            const int64_t a1 = std::min(r1, a0 + 32768);
            hipLaunchKernelGGL(pair_rowquirk_kernel, dim3((unsigned)((ctx->n + RQ_B - 1) / RQ_B), (unsigned)(a1 - a0)), dim3(256), lds,
                               ctx->stream, ctx->d_hoffsets, ctx->d_hmembers, ctx->d_hvalues, psi_dev, ldpsi, ctx->heat_hop + 1,
                               ctx->heat_hop, ctx->n, r0, r1, dst, dld, a0);
        }
    }
    return pair_heat_finish(ctx, r0, r1, out, ld, out_on_device, dst);
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_sample_distance(hsd_ctx *ctx, const double *p, int64_t np, const double *q, int64_t nq, int metric, double *out) try {
    HSD_CHECK_CTX(ctx);
    if (!out || np < 0 || nq < 0 || (np > 0 && !p) || (nq > 0 && !q)) return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_sample_distance: bad arguments");
    if (metric < HSD_METRIC_WASSERSTEIN || metric > HSD_METRIC_SYMMETRIC_KL) return hsd_fail(ctx, HSD_ERR_UNSUPPORTED, "metric %d is not implemented.", metric);
    const int64_t Lm = std::max(np, nq);
    int p2 = 1;
    while (p2 < Lm) p2 <<= 1;
    const size_t lds = (size_t)2 * p2 * sizeof(double);
    const size_t cap = ctx->prop.maxSharedMemoryPerMultiProcessor ? ctx->prop.maxSharedMemoryPerMultiProcessor : 65536;
    if (lds > cap) return hsd_fail(ctx, HSD_ERR_TOO_LARGE, "samples of %lld values do not fit in LDS", (long long)Lm);
    double *d_p = nullptr, *d_q = nullptr, *d_out = nullptr;
    int rc;
    if ((rc = hsd_dev_alloc(ctx, &d_p, (size_t)np)) || (rc = hsd_dev_alloc(ctx, &d_q, (size_t)nq)) || (rc = hsd_dev_alloc(ctx, &d_out, 1))) {
        hsd_dev_free(d_p); hsd_dev_free(d_q); hsd_dev_free(d_out);
        return rc;
    }
    hipError_t e = hipSuccess;
    if (np) e = hipMemcpyAsync(d_p, p, (size_t)np * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && nq) e = hipMemcpyAsync(d_q, q, (size_t)nq * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(sample_distance_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds, 16));
    int status = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sample_distance_kernel, dim3(1), dim3(256), std::max<size_t>(lds, 16), ctx->stream, d_p, (int)np, d_q, (int)nq, metric, d_out, ctx->d_flag);
        e = hipMemcpyAsync(out, d_out, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&status, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hsd_dev_free(d_p); hsd_dev_free(d_q); hsd_dev_free(d_out);
    if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "hsd_sample_distance: %s", hipGetErrorString(e));
    if (status == 1) return hsd_fail(ctx, HSD_ERR_NOT_NORMALISED, "The sum of probability distribution must be 1.0.");
    return HSD_OK;
} HSD_ABI_GUARD(ctx)
