// K2 (finalise): operand preparation of the all-pairs kernels -- counts -> bin-major operand matrices, the zero-pad layers' forms
// (bin / rank / byte ranges) and thresholds, the granule order (the planner), KL's smoothed histograms -- and the entry points
// hsd_pair_prepare / hsd_pair_operand_dims / hsd_pair_work / hsd_pair_shard.  Split from csrc/pair.hip in round 5 (see pair_internal.hpp);
// the kernels that consume what is prepared here are in csrc/pair.hip.
#include "pair_internal.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <system_error>
#include <thread>
#include <type_traits>

using namespace hsd_pair;

namespace {

// ------------------------------------------------------------------------------------------
// operand preparation (K2 finalise): counts -> bin-major operand matrices
// ------------------------------------------------------------------------------------------
struct PrepArgs {
    const uint32_t *counts;     // [rows][L][B]
    const double *support;      // [B]
    const int32_t *node_bin;    // [n]
    int64_t n, npad;
    int L, B;                   // layers in the counts tensor, bins
    int layer_lo, layer_hi;
    int *flag;                  // flag[0] |= 1 : empty ring met
    int stage_nodes;            // 0: one thread per node reads its row of `counts` from memory; else: workgroups of 256 threads stage the rows of
                                // stage_nodes (<= 64) consecutive nodes in LDS first (counts_row)
};

// The histogram row of (node i, layer k) for the thread that walks it, and i.  A row is B consecutive words, a node's rows are L x B words apart: with one
// thread per node reading its row in place every load of a wave touches 64 different lines (0.59 / 0.63 / 0.34 ms for the three walks of a zero-pad
// prepare at BA N = 100k, B = 241: the 193 MB of two layers were fetched ~20 times over).  Staged launches (PrepArgs::stage_nodes, prep_launch_geometry) copy the
// rows of their nodes into LDS first -- four waves, 32 row-coalesced loads in flight per lane, odd row stride: no bank conflicts -- and threads 0 ..
// stage_nodes - 1 then walk row t out of LDS; the other threads leave behind the barrier.  nullptr: nothing to do for this thread.
// (the node of this thread without its row: launches whose layer needs none, e.g. ring 0)
__device__ inline bool node_of_thread(const PrepArgs &p, int64_t &i) {
    if (p.stage_nodes == 0) i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    else i = (int)threadIdx.x < p.stage_nodes ? (int64_t)blockIdx.x * p.stage_nodes + threadIdx.x : p.n;
    return i < p.n;
}
__device__ inline const uint32_t *counts_row(const PrepArgs &p, int k, int64_t &i) {
    if (p.stage_nodes == 0) {
        i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        return i < p.n ? p.counts + (i * p.L + k) * p.B : nullptr;
    }
    extern __shared__ uint32_t prep_rows[];
    const int stride = p.B | 1;
    const int64_t i0 = (int64_t)blockIdx.x * p.stage_nodes;
    const int nodes = (int)min((int64_t)p.stage_nodes, p.n - i0);
    const int w = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63, nw = (int)blockDim.x >> 6;
    // (16 rows x 2 words per lane in flight: with 8 x 1 the copy ran at 1.2 TB/s -- 16 KB in flight per CU against a ~3 us round trip)
    for (int t0 = w; t0 < nodes; t0 += 16 * nw) {
        for (int b = lane; b < p.B; b += 128) {
            uint32_t v[16][2];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int t = t0 + u * nw;
                const uint32_t *src = p.counts + ((i0 + t) * p.L + k) * p.B;
                v[u][0] = t < nodes ? src[b] : 0u;
                v[u][1] = (t < nodes && b + 64 < p.B) ? src[b + 64] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int t = t0 + u * nw;
                if (t < nodes) {
                    prep_rows[t * stride + b] = v[u][0];
                    if (b + 64 < p.B) prep_rows[t * stride + b + 64] = v[u][1];
                }
            }
        }
    }
    __syncthreads();
    i = i0 + threadIdx.x;
    return (int)threadIdx.x < nodes ? prep_rows + (int)threadIdx.x * stride : nullptr;
}

// launch shape of a kernel that walks histogram rows through counts_row: (grid, block, dynamic LDS) for `ny` layers
struct PrepGeom { dim3 grid, block; size_t lds; };
static PrepGeom prep_launch_geometry(const PrepArgs &p, int ny) {
    if (p.stage_nodes > 0)
        return PrepGeom{dim3((unsigned)((p.n + p.stage_nodes - 1) / p.stage_nodes), (unsigned)ny), dim3(256), (size_t)p.stage_nodes * (size_t)(p.B | 1) * sizeof(uint32_t)};
    return PrepGeom{dim3((unsigned)((p.n + 255) / 256), (unsigned)ny), dim3(256), 0};
}

// W1 raw samples: column block of layer k>=1 holds G[b] = (cc[b]/n) * (s[b+1]-s[b]), b < B-1.
// One thread per (node, layer).
__global__ void prep_l1_kernel(PrepArgs p, double *__restrict__ X, int col_begin_first, int cols_per_layer,
                               double *__restrict__ point) {
    const int k = p.layer_lo + blockIdx.y;
    int64_t i;
    if (k == 0) {
        if (!node_of_thread(p, i)) return;               // ring 0 = {i}: W1 of two point masses = |v_i - v_j|
        point[i] = p.support[p.node_bin[i]];
        return;
    }
    const uint32_t *c = counts_row(p, k, i);
    if (!c) return;
    uint32_t tot = 0;
    #pragma unroll 8
    for (int b = 0; b < p.B; ++b) tot += c[b];
    if (tot == 0) atomicOr(p.flag, 1);
    const int first = p.layer_lo == 0 ? 1 : p.layer_lo;
    const int64_t cb = col_begin_first + (int64_t)(k - first) * cols_per_layer;
    const double nn = (double)tot;
    uint32_t cc = 0;
#pragma unroll 4
    for (int b = 0; b + 1 < p.B; ++b) {
        cc += c[b];
        const double cdf = tot ? (double)cc / nn : 0.0;
        X[(cb + b) * p.npad + i] = cdf * (p.support[b + 1] - p.support[b]);
    }
}

// zero-pad alignment (tools/metrics.py:26-28): both samples padded with zeros to max(n_i, n_j),
// so with tail counts T[b] = #members with value > s[b]:
//   W1 = 1/max(n_i,n_j) * ( |n_i - n_j| * s[0] + sum_b |T_i[b] - T_j[b]| * (s[b+1]-s[b]) )
// columns of layer k>=1: [ n*s[0], T[0]*d[0], ..., T[B-2]*d[B-2] ]  (B columns), inv = 1/n (2 if empty)
template <typename T>
__global__ void prep_zp_kernel(PrepArgs p, T *__restrict__ X, double *__restrict__ inv, double *__restrict__ point,
                               const int *__restrict__ seg_begin, uint32_t *__restrict__ ntot /* [n_seg][npad] ring sizes */) {
    const int k = p.layer_lo + blockIdx.y;
    int64_t i;
    if (k == 0) {
        if (!node_of_thread(p, i)) return;
        point[i] = p.support[p.node_bin[i]];
        return;
    }
    const uint32_t *c = counts_row(p, k, i);
    if (!c) return;
    const int first = p.layer_lo == 0 ? 1 : p.layer_lo;
    const int s = k - first;
    uint32_t tot = 0;
    #pragma unroll 8
    for (int b = 0; b < p.B; ++b) tot += c[b];
    inv[(int64_t)s * p.npad + i] = tot ? 1.0 / (double)tot : 2.0;
    ntot[(int64_t)s * p.npad + i] = tot;
    const int64_t cb = seg_begin[s];
    X[cb * p.npad + i] = (T)((double)tot * p.support[0]);
    uint32_t tail = tot;
#pragma unroll 4
    for (int b = 0; b + 1 < p.B; ++b) {
        tail -= c[b];
        X[(cb + 1 + b) * p.npad + i] = (T)((double)tail * (p.support[b + 1] - p.support[b]));
    }
}

// Rank columns of a zero-pad layer above its threshold (SegDesc): word w of node i holds the (2w)-th and (2w+1)-th largest
// of (member value - theta) over the members with bin index >= first_bin (PACK: two uint16 per word; else one uint32 per
// word), zeros beyond them.  The histogram is walked from the top bin down; only the first `cap` words are kept (tile pairs
// that would need more take the bin columns).
template <bool PACK>
__global__ void prep_rank_kernel(PrepArgs p, const int *__restrict__ seg_rank_col0, const int *__restrict__ seg_rank_cap,
                                 const int *__restrict__ seg_first_bin, const int *__restrict__ seg_suf_row0, uint32_t *__restrict__ X,
                                 uint32_t *__restrict__ SUF) {
    const int first = p.layer_lo == 0 ? 1 : p.layer_lo;
    const int sgi = blockIdx.y;
    const int k = first + sgi;
    if (seg_rank_col0[sgi] < 0) return;                  // (the whole launch row: no thread reaches the barrier of counts_row)
    int64_t i;
    const uint32_t *c = counts_row(p, k, i);
    if (!c) return;
    const int cap = seg_rank_cap[sgi], fb = seg_first_bin[sgi];
    const uint32_t theta = fb > 0 ? (uint32_t)p.support[fb - 1] : 0u;
    uint32_t *col = X + (int64_t)seg_rank_col0[sgi] * p.npad + i;
    // suffix sums (min form of the persistent kernel): suf[w] = sum of ALL the node's rank values from word w on, i.e. the
    // whole sum above theta minus what the first w words hold -- also where the rank columns themselves are cut off at the cap
    uint32_t *suf = SUF + (int64_t)seg_suf_row0[sgi] * p.npad + i;
    uint32_t total = 0;
    #pragma unroll 8
    for (int b = p.B - 1; b >= fb; --b) total += c[b] * ((uint32_t)p.support[b] - theta);
    const int want = PACK ? 2 * cap : cap;           // values to emit
    int r = 0;
    uint32_t lo = 0, prefix = 0;
    suf[0] = total;
    // (eight bins' counts are read together: most bins of a node are empty, and one LDS / memory latency per empty bin was most of this kernel)
    for (int b0 = p.B - 1; b0 >= fb && r < want; b0 -= 8) {
        uint32_t cc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cc[u] = b0 - u >= fb ? c[b0 - u] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (cc[u] == 0u) continue;
            const int b = b0 - u;
            const uint32_t v = (uint32_t)p.support[b] - theta;
            for (uint32_t m = cc[u]; m > 0 && r < want; --m, ++r) {
                prefix += v;
                if (PACK) {
                    if (r & 1) { col[(int64_t)(r >> 1) * p.npad] = lo | (v << 16); suf[(int64_t)((r >> 1) + 1) * p.npad] = total - prefix; }
                    else lo = v;
                } else {
                    col[(int64_t)r * p.npad] = v;
                    suf[(int64_t)(r + 1) * p.npad] = total - prefix;
                }
            }
        }
    }
    if (PACK && (r & 1)) {                            // odd count: upper half stays 0
        col[(int64_t)(r >> 1) * p.npad] = lo;
        suf[(int64_t)((r >> 1) + 1) * p.npad] = total - prefix;
    }
    // (rows beyond the node's own words stay 0: the matrix was cleared)
}

// dst word w = the four uint16 values of source words 2 w, 2 w + 1 as saturated bytes (source words from n_src on count as
// zero).  Saturated bytes are never compared: a tile with such a node does not read the uint8 range.
__global__ void pack_u8_kernel(uint32_t *__restrict__ X, int64_t npad, int src_col, int n_src, int dst_col, int n_dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = blockIdx.y;
    if (i >= npad || w >= n_dst) return;
    const uint32_t a = 2 * w < n_src ? X[(int64_t)(src_col + 2 * w) * npad + i] : 0u;
    const uint32_t b = 2 * w + 1 < n_src ? X[(int64_t)(src_col + 2 * w + 1) * npad + i] : 0u;
    const auto sat = [](uint32_t v) { return v > 255u ? 255u : v; };
    X[(int64_t)(dst_col + w) * npad + i] = sat(a & 0xffffu) | (sat(a >> 16) << 8) | (sat(b & 0xffffu) << 16) | (sat(b >> 16) << 24);
}

// above[(s * n_cand + q) * npad + i] = members of ring layer s of node i with bin index >= cand_bin[q]
__global__ void count_above_kernel(PrepArgs p, const int *__restrict__ cand_bin, int n_cand, uint32_t *__restrict__ above) {
    const int first = p.layer_lo == 0 ? 1 : p.layer_lo;
    const int sgi = blockIdx.y;
    int64_t i;
    const uint32_t *c = counts_row(p, first + sgi, i);
    if (!c) return;
    // candidates ascend: from the top bin down, the bins between two candidates are one straight run of additions
    uint32_t run = 0;
    int hi = p.B;
    for (int q = n_cand - 1; q >= 0; --q) {
        const int lo = min(cand_bin[q], hi);
#pragma unroll 8
        for (int b = hi - 1; b >= lo; --b) run += c[b];
        hi = lo;
        above[((int64_t)sgi * n_cand + q) * p.npad + i] = run;
    }
}

// rmax[s][g] = largest ring size of layer segment s within the 64-node granule g
__global__ void granule_max_kernel(const uint32_t *__restrict__ ntot, int64_t npad, int ngran, uint32_t *__restrict__ rmax) {
    const int g = blockIdx.x, sgi = blockIdx.y;
    uint32_t v = ntot[(int64_t)sgi * npad + (int64_t)g * 64 + threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_down((int)v, o, 64));
    if (threadIdx.x == 0) rmax[(int64_t)sgi * ngran + g] = v;
}

// KL: P = (c + alpha) / (n + alpha B), LP = log P, H = sum P LP.  symmetric: X = [P ; LP], Y = [LP ; P].
__global__ void prep_kl_kernel(PrepArgs p, double alpha, int symmetric, int half_cols /* padded cols of one [P] block */,
                               double *__restrict__ X, double *__restrict__ Y, double *__restrict__ Hpart /* [nl][npad] */) {
    const int kk = blockIdx.y;
    const int k = p.layer_lo + kk;
    int64_t i;
    const uint32_t *c = counts_row(p, k, i);
    if (!c) return;
    uint32_t tot = 0;
    #pragma unroll 8
    for (int b = 0; b < p.B; ++b) tot += c[b];
    if (tot == 0 && alpha == 0.0) atomicOr(p.flag, 2);
    const double den = (double)tot + alpha * (double)p.B;
    double h = 0.0;
#pragma unroll 4
    for (int b = 0; b < p.B; ++b) {
        const double P = ((double)c[b] + alpha) / den;
        const double LP = log(P);
        h += P * LP;
        const int64_t col = (int64_t)kk * p.B + b;
        X[col * p.npad + i] = P;
        Y[col * p.npad + i] = LP;
        if (symmetric) {
            X[(half_cols + col) * p.npad + i] = LP;
            Y[(half_cols + col) * p.npad + i] = P;
        }
    }
    Hpart[(int64_t)kk * p.npad + i] = h;
}

__global__ void sum_layers_kernel(const double *__restrict__ Hpart, int nl, int64_t npad, int64_t n, double *__restrict__ H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int k = 0; k < nl; ++k) s += Hpart[(int64_t)k * npad + i];
    H[i] = s;
}

// KL, the form whose columns are SPARSE where the histograms are (round 5: per-tile column trimming in the persistent kernel).
// With P = (c + alpha) / Z, Z = n + alpha B, and m(b) = log(c(b) + alpha) - log alpha (0 exactly where the bin is empty):
//   sum_b P_i(b) log P_j(b) = [ sum_b c_i(b) m_j(b)  +  n_i log alpha  +  alpha L_j ] / Z_i  -  log Z_j ,   L_j = sum_b log(c_j(b) + alpha)
// so per layer kk the columns are  X = c / Z, Y = m  (a product is non-zero only in bins occupied on BOTH sides), one more column
// X = alpha / Z, Y = L, and -- once for all layers -- X = 1, Y = -sum_kk log Z; the term n_i log alpha / Z_i goes into the entropy term
// H'_i = sum_kk (sum_b P log P - n log alpha / Z).  KL(i || j) = H'_i - X_i . Y_j as before.  symmetric: X = [X ; Y], Y = [Y ; X].
// Adding an exact zero leaves an fp64 sum unchanged, so a kernel that skips the columns empty on either side of a tile pair and one that
// runs over all of them give the same bits.
__global__ void prep_kl_sparse_kernel(PrepArgs p, double alpha, int symmetric, int nl, int half_cols /* padded columns of one half */,
                                      double *__restrict__ X, double *__restrict__ Y, double *__restrict__ Hpart /* [nl][npad] */,
                                      double *__restrict__ LZpart /* [nl][npad]: log Z */) {
    const int kk = blockIdx.y;
    const int k = p.layer_lo + kk;
    int64_t i;
    const uint32_t *c = counts_row(p, k, i);
    if (!c) return;
    uint32_t tot = 0;
    #pragma unroll 8
    for (int b = 0; b < p.B; ++b) tot += c[b];
    const double Z = (double)tot + alpha * (double)p.B, la = log(alpha);
    double h = 0.0, Ls = 0.0;
#pragma unroll 4
    for (int b = 0; b < p.B; ++b) {
        const double lc = log((double)c[b] + alpha);
        const double P = ((double)c[b] + alpha) / Z;
        h += P * (lc - log(Z));
        Ls += lc;
        const int64_t col = (int64_t)kk * p.B + b;
        const double x = (double)c[b] / Z, y = c[b] ? lc - la : 0.0;
        X[col * p.npad + i] = x;
        Y[col * p.npad + i] = y;
        if (symmetric) {
            X[(half_cols + col) * p.npad + i] = y;
            Y[(half_cols + col) * p.npad + i] = x;
        }
    }
    const int64_t ce = (int64_t)nl * p.B + kk;
    X[ce * p.npad + i] = alpha / Z;
    Y[ce * p.npad + i] = Ls;
    if (symmetric) {
        X[(half_cols + ce) * p.npad + i] = Ls;
        Y[(half_cols + ce) * p.npad + i] = alpha / Z;
    }
    Hpart[(int64_t)kk * p.npad + i] = h - (double)tot * la / Z;
    LZpart[(int64_t)kk * p.npad + i] = log(Z);
}

// the column X = 1, Y = -sum_kk log Z (symmetric: mirrored into the second half)
__global__ void kl_logz_column_kernel(const double *__restrict__ LZpart, int nl, int64_t npad, int64_t n, int64_t col, int symmetric, int half_cols,
                                      double *__restrict__ X, double *__restrict__ Y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double g = 0.0;
    for (int k = 0; k < nl; ++k) g += LZpart[(int64_t)k * npad + i];
    X[col * npad + i] = 1.0;
    Y[col * npad + i] = -g;
    if (symmetric) {
        X[(half_cols + col) * npad + i] = -g;
        Y[(half_cols + col) * npad + i] = 1.0;
    }
}

// mask[g][w]: bit q set <=> column 64 w + q of a half is occupied by one of the 16 nodes of granule g (histogram columns: a count > 0; the
// nl + 1 columns behind them: always).  The persistent kernel ORs a tile side's eight granules and ANDs the two sides.
__global__ void kl_mask_kernel(PrepArgs p, int nl, int kwords, int64_t ngran16, unsigned long long *__restrict__ mask) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ngran16 * kwords) return;
    const int64_t g = t / kwords;
    const int w = (int)(t % kwords);
    const int hist_cols = nl * p.B, all_cols = hist_cols + nl + 1;
    unsigned long long m = 0ull;
    for (int q = 0; q < 64; ++q) {
        const int col = 64 * w + q;
        if (col >= all_cols) break;
        bool on = col >= hist_cols;
        if (!on) {
            const int kk = col / p.B, b = col % p.B;
            for (int e = 0; e < 16 && !on; ++e) {
                const int64_t i = 16 * g + e;
                on = i < p.n && p.counts[((int64_t)i * p.L + p.layer_lo + kk) * p.B + b] > 0u;
            }
        }
        if (on) m |= 1ull << q;
    }
    mask[t] = m;
}

// ascending sort of each (node, layer) histogram -- tools/metrics.py:29-30 applied to the
// probability vectors (a monotone map of the counts, so sorting counts is equivalent)
__global__ void sort_counts_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int B, int Bp2) {
    extern __shared__ uint32_t sh[];
    const int64_t row = blockIdx.x;
    for (int t = threadIdx.x; t < Bp2; t += blockDim.x) sh[t] = t < B ? in[row * B + t] : 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= Bp2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < Bp2; t += blockDim.x) {
                const int x = t ^ j;
                if (x > t) {
                    const uint32_t a = sh[t], b = sh[x];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) { sh[t] = b; sh[x] = a; }
                }
            }
            __syncthreads();
        }
    for (int t = threadIdx.x; t < B; t += blockDim.x) out[row * B + t] = sh[t];
}

// per-column maximum of a uint32 bin-major matrix (one workgroup per column)
__global__ __launch_bounds__(256) void colmax_u32_kernel(const uint32_t *__restrict__ X, int64_t npad, uint32_t *__restrict__ out) {
    __shared__ uint32_t red[256];
    // (npad is a multiple of 128 words and the matrix 16-byte aligned: 16-byte loads, four in flight per lane)
    const uint4 *col = reinterpret_cast<const uint4 *>(X + (int64_t)blockIdx.x * npad);
    uint32_t m = 0;
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < npad / 4; i += 256) {
        const uint4 v = col[i];
        m = max(max(m, max(v.x, v.y)), max(v.z, v.w));
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// uint32 columns -> packed uint16 pairs.  Virtual column v of a segment is raw column vsrc[v] with
// its bin width vdiv[v] replaced by the part vmul[v] of it (a column whose values T * width exceed
// 65535 is split into several columns of smaller width: |T_i - T_j| * (d1 + d2) is the sum of both).
__global__ void pack_u16_kernel(const uint32_t *__restrict__ X32, int64_t npad, int n_seg, int stride32, int V,
                                const int *__restrict__ vsrc, const uint32_t *__restrict__ vdiv, const uint32_t *__restrict__ vmul,
                                int stride16, uint32_t *__restrict__ X16) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (i >= npad || s >= n_seg) return;
    // (a column that is not split has vdiv = vmul = 1 -- the usual case --: no 32-bit division, which is ~30 instructions on this machine; the
    // choice is uniform over the wave)
#pragma unroll 4
    for (int c = 0; 2 * c < V; ++c) {
        uint32_t w[2] = {0u, 0u};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int v = 2 * c + h;
            if (v < V) {
                const uint32_t x = X32[((int64_t)s * stride32 + vsrc[v]) * npad + i];
                const uint32_t d = vdiv[v], m = vmul[v];
                w[h] = (d == 1u && m == 1u) ? x : x / d * m;
            }
        }
        X16[((int64_t)s * stride16 + c) * npad + i] = w[0] | (w[1] << 16);
    }
}

int prepare_w1_raw(hsd_ctx *ctx, const PrepArgs &p, dim3 grid, dim3 block, int nl_hist) {
    // scipy on the raw samples: fp64 columns G = CDF x bin width, one block of B - 1 columns per layer >= 1
    const int cpl = p.B - 1;
    constexpr int CH = chunk_cols(K_L1);
    ctx->dcols_alg = nl_hist * cpl;
    ctx->dcols = (int)round_up(std::max(ctx->dcols_alg, 0), CH);
    ctx->elem_bytes = 8;
    const size_t cnt = (size_t)std::max(ctx->dcols, CH) * p.npad;
    { const int rc_x = hsd_pool_alloc(ctx, &ctx->d_X, cnt * sizeof(double)); if (rc_x) return rc_x; }
    HSD_HIP(ctx, hipMemsetAsync(ctx->d_X, 0, cnt * sizeof(double), ctx->stream));
    { const PrepGeom g = prep_launch_geometry(p, (int)grid.y); hipLaunchKernelGGL(prep_l1_kernel, g.grid, g.block, g.lds, ctx->stream, p, (double *)ctx->d_X, 0, cpl, ctx->d_H); }
    ctx->mode = PAIR_L1_F64;
    ctx->n_seg = 0;
    ctx->l1_cols_per_layer = cpl;
    return HSD_OK;
}

int prepare_w1_zeropad(hsd_ctx *ctx, const PrepArgs &p, dim3 grid, dim3 block, int nl_hist) {
    const int64_t n = p.n, npad = p.npad;
    const int B = p.B;
    int rc;
    // HSD_PLAN_VERBOSE: wall time of the stages (synchronising: measurement only)
    const bool vt = hsd_hook(H_PLAN_VERBOSE) != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!vt) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[hsd plan] %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    // the exact integer path needs integral supports and sums that fit uint32:
    //   sum_c |U_i - U_j| <= (sum of member values of ring_k(i)) + (same for j) <= 2 * sum_v value(v)
    bool use_u32 = ctx->support_integral;
    if (use_u32) {
        std::vector<int32_t> nb((size_t)n);
        HSD_HIP(ctx, hipMemcpy(nb.data(), ctx->d_node_bin, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        long double tot = 0.0L;
        for (int64_t i = 0; i < n; ++i) tot += ctx->support[(size_t)nb[(size_t)i]];
        if (2.0L * tot >= 4294967295.0L) use_u32 = false;
    }
    const int CH = use_u32 ? chunk_cols(K_SAD) : chunk_cols(K_ZP);
    const int stride = (int)round_up(B, CH);                 // unpacked bin columns per segment
    const int nsg = std::max(nl_hist, 1);
    ctx->n_seg = nl_hist;
    ctx->elem_bytes = use_u32 ? 4 : 8;
    for (int s = 0; s <= nl_hist; ++s) ctx->seg_begin[s] = s * stride;

    // ---- bin form, unpacked: [ n*s0, T(0)*d(0), ..., T(B-2)*d(B-2) ] per layer, + 1/n and n per node ----
    DevTmp x0, ntot, dseg;
    const size_t cnt0 = (size_t)std::max(nl_hist * stride, CH) * npad;
    HSD_TMP_ALLOC(ctx, x0, cnt0 * ctx->elem_bytes);
    HSD_TMP_ALLOC(ctx, ntot, (size_t)nsg * npad * sizeof(uint32_t));
    HSD_TMP_ALLOC(ctx, dseg, sizeof(ctx->seg_begin));
    HSD_HIP(ctx, hipMemsetAsync(x0.p, 0, cnt0 * ctx->elem_bytes, ctx->stream));
    HSD_HIP(ctx, hipMemsetAsync(ntot.p, 0, (size_t)nsg * npad * sizeof(uint32_t), ctx->stream));
    if ((rc = hsd_op_alloc(ctx, &ctx->d_inv, (size_t)nsg * npad))) return rc;
    {   // padded nodes: inv = 2 (treated as empty rings; never stored) -- filled on the device (no host buffer whose lifetime an
        // asynchronous copy would depend on; seg_begin lives in the context)
        hipLaunchKernelGGL(fill_f64_kernel, dim3((unsigned)(((size_t)nsg * npad + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_inv, (int64_t)nsg * npad, 2.0);
        HSD_HIP(ctx, hipMemcpyAsync(dseg.p, ctx->seg_begin, sizeof(ctx->seg_begin), hipMemcpyHostToDevice, ctx->stream));
        const PrepGeom g = prep_launch_geometry(p, (int)grid.y);
        if (use_u32)
            hipLaunchKernelGGL(prep_zp_kernel<uint32_t>, g.grid, g.block, g.lds, ctx->stream, p, x0.as<uint32_t>(), ctx->d_inv, ctx->d_H, dseg.as<int>(), ntot.as<uint32_t>());
        else
            hipLaunchKernelGGL(prep_zp_kernel<double>, g.grid, g.block, g.lds, ctx->stream, p, x0.as<double>(), ctx->d_inv, ctx->d_H, dseg.as<int>(), ntot.as<uint32_t>());
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    mark("support check + bin columns");
    HsdSegDesc *tab = ctx->seg_tab;
    for (int s = 0; s < nl_hist; ++s) tab[s] = HsdSegDesc{s * stride, B, -1, 0, 0, -1};
    ctx->dcols_alg = nl_hist * B;

    if (!use_u32 || nl_hist == 0) {
        // fp64 fallback (non-integral support or sums beyond 32 bits): bin form only
        ctx->d_X = x0.release();
        ctx->dcols = std::max(nl_hist * stride, CH);
        ctx->mode = use_u32 ? PAIR_SAD_U32 : PAIR_ZP_F64;
    } else {
        // ---- per-column maxima (packing decisions) and per-granule ring maxima (rank form) ----
        const int ncol32 = nl_hist * stride;
        const int ngran = (int)(npad / 64);
        DevTmp cmaxd;
        HSD_TMP_ALLOC(ctx, cmaxd, (size_t)ncol32 * sizeof(uint32_t));
        if ((rc = hsd_op_alloc(ctx, &ctx->d_rmax, (size_t)nl_hist * ngran))) return rc;
        ctx->ngran = ngran;
        hipLaunchKernelGGL(colmax_u32_kernel, dim3((unsigned)ncol32), dim3(256), 0, ctx->stream, x0.as<uint32_t>(), npad, cmaxd.as<uint32_t>());
        std::vector<uint32_t> cmax((size_t)ncol32);
        HSD_HIP(ctx, hipMemcpyAsync(cmax.data(), cmaxd.p, cmax.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));

        mark("column maxima");
        // uint16 packing: two columns per word, one v_sad_u16 covers both.  A column whose values (tail count x bin
        // width) reach 65536 is split into columns of smaller width (|T_i - T_j| (d1 + d2) is the sum of both parts).
        const char *e16 = hsd_hook(H_SAD16);
        bool pack = !(e16 && atoi(e16) == 0);
        std::vector<int> vsrc;
        std::vector<uint32_t> vdiv, vmul;
        for (int r = 0; r < B && pack; ++r) {
            uint32_t mx = 0;                                   // worst layer decides the split of column r
            for (int sgi = 0; sgi < nl_hist; ++sgi) mx = std::max(mx, cmax[(size_t)sgi * stride + r]);
            const double width = r == 0 ? ctx->support[0] : ctx->support[(size_t)r] - ctx->support[(size_t)r - 1];
            const uint32_t wd = (uint32_t)width;
            if (mx < 65536u) { vsrc.push_back(r); vdiv.push_back(1u); vmul.push_back(1u); continue; }
            if (wd == 0) { pack = false; break; }
            const uint32_t tmax = mx / wd;                      // largest count in this column
            if (tmax > 65535u) { pack = false; break; }
            const uint32_t dmax = 65535u / tmax;                // widest part that keeps count * part < 65536
            for (uint32_t left = wd; left > 0;) {
                const uint32_t d = std::min(left, dmax);
                vsrc.push_back(r); vdiv.push_back(wd); vmul.push_back(d);
                left -= d;
            }
        }
        const int V = (int)vsrc.size();
        if (pack && V > 4 * B) pack = false;
        const int bin_words = pack ? (V + 1) / 2 : B;
        const int bin_stride = (int)round_up(bin_words, CH);

        // Threshold and rank columns per layer.  Candidates: theta = 0 (pure rank form), the bin boundaries that fall on a
        // word boundary of the packed bin columns nearest to eighths of the support, and bins 2 / 4 / 8 / 16.  For each, the members above theta of
        // every node -> maxima per 64-node granule -> expected words per tile pair
        //   low words + E[ rank words if they fit the cap, else the upper bin words ]
        // over all granule pairs (exact from the sorted maxima); the cheapest candidate is stored, or none if nothing
        // beats the plain bin form by 3 %.
        const char *er = hsd_hook(H_RANK);
        const int rank_mode = er ? atoi(er) : 2;                // 0: bin form only; 1: theta = 0 only; 2: any threshold
        bool rank_ok = rank_mode != 0;
        if (pack && ctx->support.back() >= 65536.0) rank_ok = false;      // member values must fit a uint16 half
        const auto words_of = [&](uint32_t R) { return pack ? (int64_t)((R + 1) / 2) : (int64_t)R; };
        // first packed word of raw column r (columns are split into virtual ones for packing): word boundary iff even
        std::vector<int> vfirst((size_t)B + 1, V);
        if (pack) { for (int v = V - 1; v >= 0; --v) vfirst[(size_t)vsrc[(size_t)v]] = v; vfirst[(size_t)B] = V; }
        else for (int r = 0; r <= B; ++r) vfirst[(size_t)r] = r;
        std::vector<int> cand;                                  // raw column index r*: columns [0, r*) stay in bin form
        cand.push_back(0);
        if (rank_mode >= 2) {
            // ... and a few cuts close to the bottom of the support: a ring of ~10 members (ring 1 of a scale-free graph)
            // is cheapest with its two or three most common small values in bin form and the rest by rank
            std::vector<int> wanted = {2, 4, 8, 16};
            for (int e8 = 1; e8 < 8; ++e8) wanted.push_back(std::max(1, e8 * B / 8));
            std::sort(wanted.begin(), wanted.end());
            for (int r : wanted) {
                while (r < B && pack && (vfirst[(size_t)r] & 1)) ++r;     // packed: the cut must fall between two words
                if (r < B && r > cand.back()) cand.push_back(r);
            }
        }
        const int NC = (int)cand.size();
        const int ngran16 = (int)(npad / 16);
        std::vector<uint32_t> gmax16;                           // [nl_hist][NC][ngran16]: members above the candidate threshold, per 16-node granule
        if (rank_ok) {
            // column r >= 1 is the interval [s_{r-1}, s_r): members above theta = s_{r*-1} are those in bins >= r*
            DevTmp dcand, dabove, dgm;
            HSD_TMP_ALLOC(ctx, dcand, (size_t)NC * sizeof(int));
            HSD_TMP_ALLOC(ctx, dabove, (size_t)nl_hist * NC * npad * sizeof(uint32_t));
            HSD_TMP_ALLOC(ctx, dgm, (size_t)nl_hist * NC * ngran16 * sizeof(uint32_t));
            HSD_HIP(ctx, hipMemcpyAsync(dcand.p, cand.data(), (size_t)NC * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            HSD_HIP(ctx, hipMemsetAsync(dabove.p, 0, (size_t)nl_hist * NC * npad * sizeof(uint32_t), ctx->stream));
            const PrepGeom g = prep_launch_geometry(p, nl_hist);
            hipLaunchKernelGGL(count_above_kernel, g.grid, g.block, g.lds, ctx->stream, p, dcand.as<int>(), NC,
                               dabove.as<uint32_t>());
            hipLaunchKernelGGL(granule16_max_kernel, dim3((unsigned)((ngran16 + 255) / 256), (unsigned)(nl_hist * NC)), dim3(256), 0, ctx->stream,
                               dabove.as<uint32_t>(), npad, ngran16, dgm.as<uint32_t>());
            gmax16.resize((size_t)nl_hist * NC * ngran16);
            HSD_HIP(ctx, hipMemcpyAsync(gmax16.data(), dgm.p, gmax16.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));      // temporaries leave scope
        }
        mark("members above candidates");
        int col = nl_hist * bin_stride;
        int list_cap = 1;
        std::vector<int> first_bin((size_t)nl_hist, 0);
        std::vector<uint32_t> rmax_sel((size_t)nl_hist * ngran, 0u), rmax16_sel((size_t)nl_hist * ngran16, 0u);
        for (int sgi = 0; sgi < nl_hist; ++sgi) {
            tab[sgi] = HsdSegDesc{sgi * bin_stride, bin_words, -1, 0, 0, -1};
            list_cap += (bin_words + CH - 1) / CH + 2;  // either kernel: chunks of CH words; no form of a layer is longer than its bin form
        }
        // Tile order and thresholds are chosen together.  A tile side of the persistent kernel is eight 16-node granules that
        // are consecutive in the graph's granule order; with R(t) the most members above a layer's threshold among a tile's
        // nodes, a tile pair reads (min form)  low words + ceil(min(R_a, R_b) / 2)  rank words if that fits the cap, else
        // the upper bin words.  For each candidate order -- the natural one and the granules sorted by one (layer, threshold)
        // key -- every layer takes its cheapest threshold (expected words over all tile pairs, exact from the sorted tile
        // maxima); the cheapest total wins.  A layer that does not beat its plain bin form by 3 % keeps that.
        const int64_t G16 = n / 16;                               // whole granules
        struct LayerPlan { int q, cap; double cost; };
        std::vector<int32_t> best_order;
        std::vector<LayerPlan> best_plan((size_t)nl_hist, LayerPlan{-1, 0, (double)bin_words});
        double best_total = (double)nl_hist * bin_words;
        bool best_sorted = false;
        if (rank_ok && bin_words >= 2 && G16 >= 1) {
            const char *es = hsd_hook(H_GRAN_SORT);
            const bool may_sort = !(es && atoi(es) == 0) && G16 >= 16;
            const int64_t NT = std::max<int64_t>(1, G16 / 8);     // tiles of eight granules (a last, shorter group counts as one)
            std::vector<int32_t> order((size_t)G16);
            // Pricing one granule order = for every (layer, candidate threshold) the expected words over all unordered tile pairs.  The minimum of a pair
            // is the smaller of its tiles' maxima, so with the tiles sorted by their word count the k-th smallest (0-based) is the minimum of nt - k pairs:
            // the sum needs the COUNTS per word count only (every term is an integer: the result is the same number whatever the order of summation).
            // gT: the granule maxima transposed to [granule][layer x candidate], so that one pass over an order's granules yields every combination's tile
            // maxima at once (round 5: this pricing was 2.2 of 4.9 ms of a warm hsd_pair_prepare at BA N = 100k -- 24 orders x 24 combinations, each
            // with its own gather and its own std::sort of 781 tile maxima).
            const int ncomb = nl_hist * NC;
            std::vector<int> low_of((size_t)NC), hi_of((size_t)NC), hi_comb((size_t)ncomb);
            int hi_max = 0;
            for (int q = 0; q < NC; ++q) {
                low_of[(size_t)q] = vfirst[(size_t)cand[(size_t)q]] / (pack ? 2 : 1);
                hi_of[(size_t)q] = bin_words - low_of[(size_t)q];
                hi_max = std::max(hi_max, hi_of[(size_t)q]);
            }
            for (int c = 0; c < ncomb; ++c) hi_comb[(size_t)c] = hi_of[(size_t)(c % NC)];
            const int NBK = hi_max + 1;                           // word counts 0 .. hi_w - 1, and one bucket for "hi_w or more" (never fits the cap)
            // A maximum enters the price only through min(words_of(.), hi_w): values are clamped to the smallest one whose word count reaches every
            // hi_w, which makes the table 16-bit (signed: the host's baseline vector max) whenever the bin form has fewer than 16 k words
            const int64_t clamp_to = pack ? 2 * (int64_t)hi_max : (int64_t)hi_max;
            const bool narrow = clamp_to <= 32767;
            std::vector<int16_t> gT16;
            std::vector<uint32_t> gT32;
            if (narrow) gT16.resize((size_t)(G16 + 1) * ncomb, 0); else gT32.resize((size_t)(G16 + 1) * ncomb, 0u);
            for (int c = 0; c < ncomb; ++c) {
                const uint32_t *gm = gmax16.data() + (size_t)c * ngran16;
                for (int64_t g = 0; g <= G16 && g < ngran16; ++g) {
                    if (narrow) gT16[(size_t)g * ncomb + c] = (int16_t)std::min<int64_t>((int64_t)gm[g], clamp_to);
                    else gT32[(size_t)g * ncomb + c] = gm[g];
                }
            }
            auto plan_with = [&](const auto *gT, const std::vector<int32_t> &ord, std::vector<LayerPlan> &plan) -> double {
                using E = std::remove_cv_t<std::remove_reference_t<decltype(gT[0])>>;
                const int64_t nt = (G16 + 7) / 8;
                std::vector<E> tm((size_t)ncomb), gmx((size_t)ncomb, (E)0);            // (per call: candidate orders are evaluated on several host threads)
                std::vector<int32_t> cnt((size_t)ncomb * NBK, 0);
                E *const tmp = tm.data(), *const gmp = gmx.data();
                for (int64_t t = 0; t < nt; ++t) {
                    for (int c = 0; c < ncomb; ++c) tmp[c] = (E)0;
                    for (int64_t e = 8 * t; e < std::min<int64_t>(G16, 8 * t + 8); ++e) {
                        const E *row = gT + (size_t)ord[(size_t)e] * ncomb;
                        for (int c = 0; c < ncomb; ++c) tmp[c] = std::max(tmp[c], row[c]);
                    }
                    for (int c = 0; c < ncomb; ++c) gmp[c] = std::max(gmp[c], tmp[c]);
                    for (int c = 0; c < ncomb; ++c) {
                        const int hi_w = hi_comb[(size_t)c];
                        if (hi_w < 2) continue;
                        const int64_t wr = words_of((uint32_t)tmp[c]);
                        ++cnt[(size_t)c * NBK + (size_t)std::min<int64_t>(wr, hi_w)];
                    }
                }
                double total = 0.0;
                for (int sgi = 0; sgi < nl_hist; ++sgi) {
                    LayerPlan bp{-1, 0, 0.97 * bin_words};
                    for (int q = 0; q < NC; ++q) {
                        const int c = sgi * NC + q;
                        const int low_w = low_of[(size_t)q], hi_w = hi_of[(size_t)q];
                        if (hi_w < 2) continue;
                        uint32_t g_all = (uint32_t)gmx[(size_t)c];
                        if (n % 16) g_all = std::max(g_all, (uint32_t)gT[(size_t)G16 * ncomb + c]);        // (the partial last granule: one-tile kernel only)
                        const int cap = (int)std::min<int64_t>(hi_w - 1, words_of(g_all));
                        // sum over the tiles in ascending order of f(words) x (nt - position): a bucket of c_b tiles at positions k0 .. k0 + c_b - 1
                        int64_t acc = 0, k0 = 0;
                        for (int b = 0; b <= hi_w; ++b) {
                            const int64_t cb = cnt[(size_t)c * NBK + b];
                            if (!cb) continue;
                            int64_t wr = b;
                            if (low_w + wr == 0) wr = 1;
                            const int64_t f = (b < hi_w && wr <= cap) ? wr : hi_w;
                            acc += f * (cb * nt - (cb * k0 + cb * (cb - 1) / 2));
                            k0 += cb;
                        }
                        const double cost = low_w + (double)((long double)acc / ((long double)nt * (nt + 1) / 2));
                        if (cost < bp.cost) bp = LayerPlan{q, cap, cost};
                    }
                    if (bp.q < 0) bp.cost = bin_words;
                    plan[(size_t)sgi] = bp;
                    total += bp.cost;
                }
                return total;
            };
            auto plan_for = [&](const std::vector<int32_t> &ord, std::vector<LayerPlan> &plan) -> double {
                return narrow ? plan_with(gT16.data(), ord, plan) : plan_with(gT32.data(), ord, plan);
            };
            (void)NT;
            for (int64_t g = 0; g < G16; ++g) order[(size_t)g] = (int32_t)g;
            std::vector<LayerPlan> plan((size_t)nl_hist);
            best_total = plan_for(order, plan);
            best_plan = plan;
            best_order = order;
            if (may_sort) {
                // one candidate order per (layer, threshold) key: sorted and priced independently (host threads: 24 orders x 24
                // (layer, threshold) prices took 13 of the 17.6 ms of a warm hsd_pair_prepare at BA N = 100k on one core), then
                // compared in the fixed order of the keys
                const int n_task = nl_hist * NC;
                std::vector<std::vector<int32_t>> t_order((size_t)n_task);
                std::vector<std::vector<LayerPlan>> t_plan((size_t)n_task, std::vector<LayerPlan>((size_t)nl_hist));
                std::vector<double> t_total((size_t)n_task, 0.0);
                const char *ept = hsd_hook(H_PLAN_THREADS);    // cap of the host threads (1: everything on the calling thread)
                int thr_cap = ept ? atoi(ept) : 4;              // (round 5: a task is ~35 us now; more threads cost more to start than they save)
                if (thr_cap < 1) thr_cap = 1;
                const int n_thr = std::max(1, std::min<int>(std::min(n_task, thr_cap), std::max(1, (int)std::thread::hardware_concurrency())));
                // tasks are drawn from a shared counter, so whatever threads exist (possibly none but the caller) finish them all
                std::atomic<int> next_task{0};
                std::atomic<bool> task_failed{false};            // nothing may leave a thread, or the C ABI, as an exception
                auto work = [&]() noexcept {
                    try {
                        for (int task; (task = next_task.fetch_add(1, std::memory_order_relaxed)) < n_task;) {
                            const uint32_t *key = gmax16.data() + (size_t)task * ngran16;        // task = ks * NC + kq
                            std::vector<int32_t> &ord = t_order[(size_t)task];
                            ord.resize((size_t)G16);
                            // stable order by key: a counting sort where the keys are small (member counts), std::stable_sort otherwise -- the same permutation
                            uint32_t kmax = 0;
                            for (int64_t g = 0; g < G16; ++g) kmax = std::max(kmax, key[g]);
                            if (kmax < (1u << 20)) {
                                std::vector<int32_t> at((size_t)kmax + 2, 0);
                                for (int64_t g = 0; g < G16; ++g) ++at[(size_t)key[g] + 1];
                                for (size_t v = 1; v < at.size(); ++v) at[v] += at[v - 1];
                                for (int64_t g = 0; g < G16; ++g) ord[(size_t)at[(size_t)key[g]]++] = (int32_t)g;
                            } else {
                                for (int64_t g = 0; g < G16; ++g) ord[(size_t)g] = (int32_t)g;
                                std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return key[x] < key[y]; });
                            }
                            t_total[(size_t)task] = plan_for(ord, t_plan[(size_t)task]);
                        }
                    } catch (...) { task_failed.store(true); }
                };
                {
                    std::vector<std::thread> pool;
                    try {
                        pool.reserve((size_t)n_thr);
                        const char *eft = hsd_hook(H_PLAN_THREADS_FAIL);       // test hook: creating the t-th thread fails
                        const int fail_at = eft ? atoi(eft) : 0;
                        for (int t = 1; t < n_thr; ++t) {
                            if (t == fail_at) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
                            pool.emplace_back(work);
                        }
                    } catch (...) {
                        // RLIMIT_NPROC / a pids limit: the threads that exist and the caller share the tasks
                    }
                    work();
                    for (auto &th : pool) th.join();
                }
                if (task_failed.load()) return hsd_fail(ctx, HSD_ERR_NOMEM, "hsd_pair_prepare: out of host memory while pricing the granule orders");
                for (int task = 0; task < n_task; ++task)
                    if (t_total[(size_t)task] < 0.99 * best_total) {
                        best_total = t_total[(size_t)task]; best_plan = t_plan[(size_t)task]; best_order = t_order[(size_t)task]; best_sorted = true;
                    }
            }
        }
        mark("order + thresholds (host)");
        ctx->gran_order.clear();
        if (best_sorted) ctx->gran_order = best_order;
        int suf_rows = 1;                                         // row 0: zeros
        const char *e8 = hsd_hook(H_U8);
        const bool u8_ok = !(e8 && atoi(e8) == 0);               // HSD_U8=0: no uint8 ranges (tests: same bits either way)
        std::vector<int> r8_reserved((size_t)nl_hist, -1);
        for (int sgi = 0; sgi < nl_hist; ++sgi) {
            const LayerPlan &bp = best_plan[(size_t)sgi];
            if (bp.q < 0) continue;
            const int low_w = vfirst[(size_t)cand[(size_t)bp.q]] / (pack ? 2 : 1);
            tab[sgi].rank_col0 = col;
            tab[sgi].rank_low = low_w;
            tab[sgi].rank_cap = bp.cap;
            tab[sgi].suf_row0 = suf_rows;
            suf_rows += bp.cap + 1;
            first_bin[(size_t)sgi] = cand[(size_t)bp.q];
            const uint32_t *gm = gmax16.data() + ((size_t)sgi * NC + bp.q) * ngran16;
            std::copy(gm, gm + ngran16, rmax16_sel.begin() + (size_t)sgi * ngran16);
            for (int g = 0; g < ngran; ++g)                        // the one-tile kernel plans from 64-node granules in natural order
                rmax_sel[(size_t)sgi * ngran + g] = std::max(std::max(gm[4 * g], gm[4 * g + 1]), std::max(gm[4 * g + 2], gm[4 * g + 3]));
            col += (int)round_up(low_w + bp.cap, CH);
            if (pack && u8_ok) { r8_reserved[(size_t)sgi] = col; col += (int)round_up(low_w + bp.cap + 1, CH); }
        }
        if ((rc = hsd_op_alloc(ctx, &ctx->d_rmax16, (size_t)nl_hist * ngran16))) return rc;
        ctx->ngran16 = ngran16;
        if ((rc = hsd_op_alloc(ctx, &ctx->d_suf, (size_t)(suf_rows + 1) * npad))) return rc;
        ctx->suf_rows = suf_rows;
        HSD_HIP(ctx, hipMemsetAsync(ctx->d_suf, 0, (size_t)(suf_rows + 1) * npad * sizeof(uint32_t), ctx->stream));
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_rmax, rmax_sel.data(), rmax_sel.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_rmax16, rmax16_sel.data(), rmax16_sel.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));          // rmax_sel / rmax16_sel are locals
        if (hsd_hook(H_PLAN_VERBOSE)) {
            fprintf(stderr, "[hsd plan] %s granule order, expected words per tile pair %.1f (bin form %d per layer):", best_sorted ? "sorted" : "natural", best_total,
                    bin_words);
            for (int sgi = 0; sgi < nl_hist; ++sgi)
                fprintf(stderr, " layer %d: first rank bin %d, low %d, cap %d, %.1f;", sgi, first_bin[(size_t)sgi], tab[sgi].rank_low, tab[sgi].rank_cap,
                        best_plan[(size_t)sgi].cost);
            fprintf(stderr, "\n");
        }
        ctx->list_cap = list_cap;
        ctx->dcols = std::max(col, CH);
        ctx->dcols_alg = nl_hist * bin_words;

        const size_t cntf = (size_t)ctx->dcols * npad;
        { const int rc_x = hsd_pool_alloc(ctx, &ctx->d_X, cntf * sizeof(uint32_t)); if (rc_x) return rc_x; }
        HSD_HIP(ctx, hipMemsetAsync(ctx->d_X, 0, cntf * sizeof(uint32_t), ctx->stream));
        if (pack) {
            DevTmp dv;
            HSD_TMP_ALLOC(ctx, dv, (size_t)V * 3 * sizeof(uint32_t));
            uint32_t *d_vsrc = dv.as<uint32_t>(), *d_vdiv = d_vsrc + V, *d_vmul = d_vdiv + V;
            HSD_HIP(ctx, hipMemcpyAsync(d_vsrc, vsrc.data(), (size_t)V * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            HSD_HIP(ctx, hipMemcpyAsync(d_vdiv, vdiv.data(), (size_t)V * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            HSD_HIP(ctx, hipMemcpyAsync(d_vmul, vmul.data(), (size_t)V * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(pack_u16_kernel, dim3((unsigned)((npad + 255) / 256), (unsigned)nl_hist), dim3(256), 0, ctx->stream,
                               x0.as<uint32_t>(), npad, nl_hist, stride, V, reinterpret_cast<const int *>(d_vsrc), d_vdiv, d_vmul, bin_stride,
                               (uint32_t *)ctx->d_X);
            HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));      // dv leaves scope
            ctx->seg_valid16 = bin_words;
            ctx->mode = PAIR_SAD_U16;
        } else {
            for (int sgi = 0; sgi < nl_hist; ++sgi)
                HSD_HIP(ctx, hipMemcpyAsync((uint32_t *)ctx->d_X + (size_t)sgi * bin_stride * npad, x0.as<uint32_t>() + (size_t)sgi * stride * npad,
                                            (size_t)B * npad * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
            ctx->mode = PAIR_SAD_U32;
        }
        mark("operand matrix + packing");
        for (int s = 0; s <= nl_hist; ++s) ctx->seg_begin[s] = s * bin_stride;
        bool any_rank = false;
        for (int sgi = 0; sgi < nl_hist; ++sgi) any_rank |= tab[sgi].rank_col0 >= 0;
        if (any_rank) {
            DevTmp dr;
            std::vector<int> rc0((size_t)4 * nl_hist);
            for (int sgi = 0; sgi < nl_hist; ++sgi) {
                rc0[(size_t)sgi] = tab[sgi].rank_col0 < 0 ? -1 : tab[sgi].rank_col0 + tab[sgi].rank_low;
                rc0[(size_t)nl_hist + sgi] = tab[sgi].rank_cap;
                rc0[(size_t)2 * nl_hist + sgi] = first_bin[(size_t)sgi];
                rc0[(size_t)3 * nl_hist + sgi] = tab[sgi].suf_row0;
                if (tab[sgi].rank_col0 >= 0 && tab[sgi].rank_low > 0)       // the bin columns below the threshold, once more in front of the ranks
                    HSD_HIP(ctx, hipMemcpyAsync((uint32_t *)ctx->d_X + (size_t)tab[sgi].rank_col0 * npad, (uint32_t *)ctx->d_X + (size_t)tab[sgi].bin_col0 * npad,
                                                (size_t)tab[sgi].rank_low * npad * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
            }
            HSD_TMP_ALLOC(ctx, dr, rc0.size() * sizeof(int));
            HSD_HIP(ctx, hipMemcpyAsync(dr.p, rc0.data(), rc0.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            const PrepGeom g = prep_launch_geometry(p, nl_hist);
            if (pack) hipLaunchKernelGGL(prep_rank_kernel<true>, g.grid, g.block, g.lds, ctx->stream, p, dr.as<int>(), dr.as<int>() + nl_hist, dr.as<int>() + 2 * nl_hist, dr.as<int>() + 3 * nl_hist, (uint32_t *)ctx->d_X, ctx->d_suf);
            else hipLaunchKernelGGL(prep_rank_kernel<false>, g.grid, g.block, g.lds, ctx->stream, p, dr.as<int>(), dr.as<int>() + nl_hist, dr.as<int>() + 2 * nl_hist, dr.as<int>() + 3 * nl_hist, (uint32_t *)ctx->d_X, ctx->d_suf);
            HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));      // dr leaves scope
        }
        mark("rank columns + suffix sums");
        // ---- uint8 ranges (persistent kernel).  Most values of a rank-form range are small: ring 2 of a scale-free graph has a
        // handful of hubs in front of hundreds of ordinary members, and its tail counts fall below 256 a few bins up.  v_sad_u8
        // takes four such columns per instruction where v_sad_u16 takes two.  Per layer: keep the first b8 low-bin words and the
        // first r16 rank words as they are and store the rest two-words-into-one as bytes; a tile may read that range when none
        // of its nodes holds a value >= 256 in the byte part (flag in bit 31 of its granules' maxima).  (b8, r16) by expected
        // words over all tile pairs in the chosen order, exact from the sorted tile maxima of the tiles that qualify.
        for (int sgi = 0; sgi < nl_hist && pack && u8_ok; ++sgi) {
            if (tab[sgi].rank_col0 < 0 || r8_reserved[(size_t)sgi] < 0 || G16 < 8) continue;
            const int low = tab[sgi].rank_low, cap = tab[sgi].rank_cap;
            DevTmp dneed, dgn;
            HSD_TMP_ALLOC(ctx, dneed, (size_t)2 * npad * sizeof(uint32_t));
            HSD_TMP_ALLOC(ctx, dgn, (size_t)2 * ngran16 * sizeof(uint32_t));
            HSD_HIP(ctx, hipMemsetAsync(dneed.p, 0, (size_t)2 * npad * sizeof(uint32_t), ctx->stream));
            hipLaunchKernelGGL(u8_need_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_X, npad, n, tab[sgi],
                               dneed.as<uint32_t>());
            hipLaunchKernelGGL(granule16_max_kernel, dim3((unsigned)((ngran16 + 255) / 256), 2u), dim3(256), 0, ctx->stream, dneed.as<uint32_t>(), npad, ngran16,
                               dgn.as<uint32_t>());
            std::vector<uint32_t> gneed((size_t)2 * ngran16);
            HSD_HIP(ctx, hipMemcpyAsync(gneed.data(), dgn.p, gneed.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const uint32_t *gb8 = gneed.data(), *gr8 = gneed.data() + ngran16;
            uint32_t *gR = rmax16_sel.data() + (size_t)sgi * ngran16;
            const int64_t nt = (G16 + 7) / 8;
            std::vector<uint32_t> tb8((size_t)nt, 0u), tr8((size_t)nt, 0u), tR((size_t)nt, 0u);
            for (int64_t t = 0; t < nt; ++t)
                for (int64_t e = 8 * t; e < std::min<int64_t>(G16, 8 * t + 8); ++e) {
                    const int64_t g = ctx->gran_order.empty() ? e : ctx->gran_order[(size_t)e];
                    tb8[(size_t)t] = std::max(tb8[(size_t)t], gb8[g]);
                    tr8[(size_t)t] = std::max(tr8[(size_t)t], gr8[g]);
                    tR[(size_t)t] = std::max(tR[(size_t)t], gR[g]);
                }
            std::vector<int> cb, cr;
            for (int v : {0, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 64}) if (v < low) cb.push_back(v);
            cb.push_back(low);
            for (int v : {0, 1, 2, 3, 4, 6, 8, 10, 12, 16, 20, 24, 32, 48}) if (v < cap) cr.push_back(v);
            cr.push_back(cap);
            double best_save = 0.0;
            int best_b = -1, best_r = -1;
            // the tiles once in ascending order of their rank words; every (b8, r16) then walks the qualifying ones in that order (the sum's terms are
            // integers: equal word counts may come in any order)
            std::vector<int32_t> by_words((size_t)nt);
            for (int64_t t = 0; t < nt; ++t) by_words[(size_t)t] = (int32_t)t;
            std::stable_sort(by_words.begin(), by_words.end(), [&](int32_t x, int32_t y) { return tR[(size_t)x] < tR[(size_t)y]; });
            std::vector<int32_t> sb8((size_t)nt), sr8((size_t)nt);
            std::vector<int64_t> swr((size_t)nt);
            for (int64_t k2 = 0; k2 < nt; ++k2) {
                const size_t t = (size_t)by_words[(size_t)k2];
                sb8[(size_t)k2] = (int32_t)std::min<uint32_t>(tb8[t], 0x7fffffffu);
                sr8[(size_t)k2] = (int32_t)std::min<uint32_t>(tr8[t], 0x7fffffffu);
                swr[(size_t)k2] = words_of(tR[t]);
            }
            for (int b8 : cb)
                for (int r16 : cr) {
                    const int fixed = b8 + r16 + (low - b8 + 1) / 2;
                    int64_t m = 0;
                    for (int64_t t = 0; t < nt; ++t) m += (int)(sb8[(size_t)t] <= b8) & (int)(sr8[(size_t)t] <= r16);
                    long double save = 0.0L;
                    int64_t k2 = 0;
                    for (int64_t t = 0; t < nt && k2 < m; ++t) {
                        if (!(sb8[(size_t)t] <= b8 && sr8[(size_t)t] <= r16)) continue;
                        int64_t wr = swr[(size_t)t];
                        if (low + wr == 0) wr = 1;
                        if (wr > cap) break;
                        const int64_t c16 = low + wr, c8 = fixed + std::max<int64_t>(wr - r16, 0) / 2 + (std::max<int64_t>(wr - r16, 0) & 1);
                        if (c8 >= 1 && c8 < c16) save += (long double)(c16 - c8) * (long double)(m - k2);
                        ++k2;
                    }
                    const double sv = (double)(save / ((long double)nt * (nt + 1) / 2));
                    if (sv > best_save) { best_save = sv; best_b = b8; best_r = r16; }
                }
            if (best_b < 0 || best_save < 0.5) continue;
            HsdSegDesc &d = tab[sgi];
            d.r8_col0 = r8_reserved[(size_t)sgi];
            d.r8_k16 = best_b + best_r;
            d.r8_fixed = d.r8_k16 + (low - best_b + 1) / 2;
            d.r8_r16 = best_r;
            d.r8_cap = (cap - best_r + 1) / 2;
            uint32_t *X = (uint32_t *)ctx->d_X;
            if (best_b > 0)
                HSD_HIP(ctx, hipMemcpyAsync(X + (size_t)d.r8_col0 * npad, X + (size_t)d.rank_col0 * npad, (size_t)best_b * npad * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
            if (best_r > 0)
                HSD_HIP(ctx, hipMemcpyAsync(X + (size_t)(d.r8_col0 + best_b) * npad, X + (size_t)(d.rank_col0 + low) * npad, (size_t)best_r * npad * sizeof(uint32_t),
                                            hipMemcpyDeviceToDevice, ctx->stream));
            const int nb8 = (low - best_b + 1) / 2;
            if (nb8 > 0)
                hipLaunchKernelGGL(pack_u8_kernel, dim3((unsigned)((npad + 255) / 256), (unsigned)nb8), dim3(256), 0, ctx->stream, X, npad, d.rank_col0 + best_b,
                                   low - best_b, d.r8_col0 + d.r8_k16, nb8);
            if (d.r8_cap > 0)
                hipLaunchKernelGGL(pack_u8_kernel, dim3((unsigned)((npad + 255) / 256), (unsigned)d.r8_cap), dim3(256), 0, ctx->stream, X, npad,
                                   d.rank_col0 + low + best_r, cap - best_r, d.r8_col0 + d.r8_fixed, d.r8_cap);
            for (int g = 0; g < ngran16; ++g)
                if ((int)gb8[g] > best_b || (int)gr8[g] > best_r) gR[g] |= 0x80000000u;
            HSD_HIP(ctx, hipMemcpyAsync(ctx->d_rmax16 + (size_t)sgi * ngran16, gR, (size_t)ngran16 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));      // gR (rmax16_sel) / temporaries
            if (hsd_hook(H_PLAN_VERBOSE))
                fprintf(stderr, "[hsd plan] layer %d uint8 range: %d low-bin + %d rank words stay uint16, %d + up to %d words of bytes; saves %.1f words per tile pair\n", sgi,
                        best_b, best_r, nb8, d.r8_cap, best_save);
        }
    }
    mark("byte ranges");
    if (ctx->mode == PAIR_ZP_F64 || nl_hist == 0 || ctx->list_cap == 0) {
        ctx->list_cap = 1;
        for (int s = 0; s < nl_hist; ++s) ctx->list_cap += (tab[s].bin_words + CH - 1) / CH + 2;
    }
    if ((rc = hsd_op_alloc(ctx, &ctx->d_seg_tab, (size_t)nsg))) return rc;
    HSD_HIP(ctx, hipMemcpyAsync(ctx->d_seg_tab, tab, (size_t)nsg * sizeof(HsdSegDesc), hipMemcpyHostToDevice, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));              // x0 / ntot / dseg leave scope
    return HSD_OK;
}

// ring 0 = {i}: its histogram is the point mass at node_bin[i] -- unless the caller uploaded other counts (hsd_signature_set_counts); flag bit 2 then
__global__ void kl_point_kernel(PrepArgs p, double *__restrict__ point) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const uint32_t *c = p.counts + (int64_t)i * p.L * p.B;
    uint32_t tot = 0;
    #pragma unroll 8
    for (int b = 0; b < p.B; ++b) tot += c[b];
    const int nb = p.node_bin[i];
    if (tot != 1u || c[nb] != 1u) atomicOr(p.flag + 1, 4);
    point[i] = (double)nb;
}

// KL of layer 0 in closed form.  Ring 0 of node i is the point mass at its own bin b_i, so on the smoothed support
// P_i(b) = ([b == b_i] + alpha) / Z, Z = 1 + alpha B, and sum_b P_i log(P_i / P_j) has two values only:
//   0 where b_i == b_j, and  ((1 + alpha) - alpha) (log((1 + alpha) / Z) - log(alpha / Z)) / Z = log((1 + alpha) / alpha) / Z  elsewhere
// -- the same both ways (so the symmetric form adds the same constant), and sorted histograms (kl_sorted) of two point masses are
// equal: nothing to add.  The B dense columns of that layer (a sixth of config 3's arithmetic, a quarter at K = 3) become one compare
// per pair in the tile epilogue, where W1 adds its own ring-0 term |v_i - v_j|.  HSD_KL_POINT=0: layer 0 stays in the dense loop.
int prepare_kl(hsd_ctx *ctx, PrepArgs p, dim3 grid, dim3 block, int nl, int metric, double kl_alpha, int kl_sorted) {
    const int64_t n = p.n, npad = p.npad;
    const int B = p.B, L = p.L;
    const int sym = metric == HSD_METRIC_SYMMETRIC_KL;
    constexpr int CH = chunk_cols(K_DOT);
    ctx->has_point = false;
    ctx->kl_point_c = 0.0;
    bool closed = p.layer_lo == 0 && nl >= 2 && kl_alpha > 0.0 && hsd_hook_int(H_KL_POINT, 1) != 0;
    if (closed) {
        // (d_H holds two vectors for the KL kinds: the entropy term [npad], then the ring-0 values [npad])
        hipLaunchKernelGGL(kl_point_kernel, dim3(grid.x), block, 0, ctx->stream, p, ctx->d_H + npad);
        int flag[4] = {0, 0, 0, 0};
        HSD_HIP(ctx, hipMemcpyAsync(flag, ctx->d_flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        HSD_HIP(ctx, hipMemsetAsync(ctx->d_flag + 1, 0, sizeof(int), ctx->stream));
        if (flag[1] & 4) closed = false;            // ring 0 is not the node itself somewhere (caller-assigned rings): the dense loop
    }
    if (closed) {
        p.layer_lo = 1;
        nl -= 1;
        grid.y = (unsigned)nl;
        if (!kl_sorted) {
            ctx->has_point = true;
            ctx->kl_point_c = (log((1.0 + kl_alpha) / kl_alpha)) / (1.0 + kl_alpha * (double)B);
        }
    }
    // sparse-column form + per-tile column trimming (needs alpha > 0: log alpha; HSD_KL_TRIM=0: the dense form P . log P)
    const bool sparse = kl_alpha > 0.0 && hsd_hook_int(H_KL_TRIM, 1) != 0;
    const int half_alg = sparse ? nl * B + nl + 1 : nl * B;
    const int half = (int)round_up((int64_t)half_alg, CH);
    ctx->dcols = sym ? 2 * half : half;
    ctx->dcols_alg = (sym ? 2 : 1) * half_alg;
    ctx->elem_bytes = 8;
    hsd_op_free(ctx, ctx->d_kl_mask);
    ctx->kl_words = 0;
    ctx->kl_half = sym ? half : 0;
    ctx->kl_half_alg = half_alg;
    const size_t cnt = (size_t)ctx->dcols * npad;
    { const int rc_x = hsd_pool_alloc(ctx, &ctx->d_X, cnt * sizeof(double)); if (rc_x) return rc_x; }
    { const int rc_y = hsd_pool_alloc(ctx, &ctx->d_Y, cnt * sizeof(double)); if (rc_y) return rc_y; }
    HSD_HIP(ctx, hipMemsetAsync(ctx->d_X, 0, cnt * sizeof(double), ctx->stream));
    HSD_HIP(ctx, hipMemsetAsync(ctx->d_Y, 0, cnt * sizeof(double), ctx->stream));
    DevTmp hpart, sorted;
    HSD_TMP_ALLOC(ctx, hpart, (size_t)nl * npad * sizeof(double));
    if (kl_sorted) {
        int Bp2 = 1;
        while (Bp2 < B) Bp2 <<= 1;
        HSD_TMP_ALLOC(ctx, sorted, (size_t)n * L * B * sizeof(uint32_t));
        hipLaunchKernelGGL(sort_counts_kernel, dim3((unsigned)(n * L)), dim3(std::min(256, std::max(64, Bp2 / 2))),
                           (size_t)Bp2 * sizeof(uint32_t), ctx->stream, ctx->d_counts, sorted.as<uint32_t>(), B, Bp2);
        p.counts = sorted.as<uint32_t>();
    }
    if (!sparse) {
        const PrepGeom g = prep_launch_geometry(p, (int)grid.y);
        hipLaunchKernelGGL(prep_kl_kernel, g.grid, g.block, g.lds, ctx->stream, p, kl_alpha, sym, half, (double *)ctx->d_X,
                           (double *)ctx->d_Y, hpart.as<double>());
    } else {
        DevTmp lz;
        HSD_TMP_ALLOC(ctx, lz, (size_t)nl * npad * sizeof(double));
        const PrepGeom g = prep_launch_geometry(p, (int)grid.y);
        hipLaunchKernelGGL(prep_kl_sparse_kernel, g.grid, g.block, g.lds, ctx->stream, p, kl_alpha, sym, nl, half, (double *)ctx->d_X, (double *)ctx->d_Y,
                           hpart.as<double>(), lz.as<double>());
        hipLaunchKernelGGL(kl_logz_column_kernel, dim3((unsigned)((n + 255) / 256)), block, 0, ctx->stream, (const double *)lz.as<double>(), nl, npad, n,
                           (int64_t)nl * B + nl, sym, half, (double *)ctx->d_X, (double *)ctx->d_Y);
        // occupancy masks per 16-node granule; the tile's column list lives in the kernel's chunk-list area as uint16 ids.  Trimming only while two
        // lists of a half's columns fit the persistent workgroup's LDS budget (two workgroups per CU: config 3 has 986 columns per half; ~1 700 is
        // the limit) and the ids fit 16 bits, and the masks' words fit the planning wave's 64 lanes: otherwise the sparse columns are simply multiplied densely
        const int kwords = (half_alg + 63) / 64;
        const int list_cap = (half_alg * 2 + 7) / 8 + 2;        // int2 entries per list buffer: half_alg uint16 ids (+ slack for a whole last chunk)
        // (launch_stream's sum for this kind: two stages of 2 x 16 fp64 columns x 128 nodes, the two lists, the plan ring and descriptions, the
        //  entropy term and -- closed form of ring 0 -- the ring-0 values of two tiles, the k-nearest sweep's thresholds)
        const size_t lds_two = 2 * ((size_t)65536 + (size_t)2 * list_cap * 8 + 16 + 3 * 24 * 4 + 32 + 4 * sizeof(HsdSegDesc) +
                                    (size_t)2 * (ctx->has_point ? 2 : 1) * 256 * 8 + 512);
        if (half_alg <= 65535 && kwords <= 64 && lds_two <= (size_t)160 * 1024 && hsd_hook_int(H_KL_TRIM, 1) != 2) {      // (HSD_KL_TRIM=2: tests)
            const int64_t ngran16 = npad / 16;
            int rc_m;
            if ((rc_m = hsd_op_alloc(ctx, &ctx->d_kl_mask, (size_t)ngran16 * kwords))) return rc_m;
            hipLaunchKernelGGL(kl_mask_kernel, dim3((unsigned)((ngran16 * kwords + 255) / 256)), block, 0, ctx->stream, p, nl, kwords, ngran16, ctx->d_kl_mask);
            ctx->kl_words = kwords;
            ctx->list_cap = list_cap;
        }
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));          // (lz goes)
    }
    hipLaunchKernelGGL(sum_layers_kernel, dim3((unsigned)((n + 255) / 256)), block, 0, ctx->stream, hpart.as<double>(), nl, npad, n, ctx->d_H);
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));              // temporaries leave scope
    ctx->mode = sym ? PAIR_SKL : PAIR_KL;
    ctx->n_seg = 0;
    return HSD_OK;
}

int pair_prepare_impl(hsd_ctx *ctx, int metric, int align, int layer_lo, int layer_hi, double kl_alpha, int kl_sorted) {
    const int64_t n = ctx->n;
    const int L = ctx->hop + 1, B = ctx->n_bins;
    const int64_t npad = round_up(n + HSD_TILE, HSD_TILE);
    ctx->npad = npad;
    ctx->list_cap = 0;
    HSD_HIP(ctx, hipMemsetAsync(ctx->d_flag, 0, 4 * sizeof(int), ctx->stream));
    // rows staged through LDS where 64 (32, 16) of them fit 64 KiB (HSD_PREP_STAGED=0: never)
    int stage_nodes = 0;
    {
        const char *es = hsd_hook(H_PREP_STAGED);
        if (!(es && atoi(es) == 0))
            for (int c : {64, 32, 16})
                if ((size_t)c * (size_t)(B | 1) * sizeof(uint32_t) <= (size_t)64 * 1024) { stage_nodes = c; break; }
    }
    PrepArgs p{ctx->d_counts, ctx->d_support, ctx->d_node_bin, n, npad, L, B, layer_lo, layer_hi, ctx->d_flag, stage_nodes};
    const int nl = layer_hi - layer_lo + 1;
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)nl), block(256);
    int rc;
    // d_H: per-node entropy term (KL) or the ring-0 point values (W1 with layer 0)
    if ((rc = hsd_op_alloc(ctx, &ctx->d_H, (size_t)2 * npad))) return rc;
    HSD_HIP(ctx, hipMemsetAsync(ctx->d_H, 0, (size_t)2 * npad * sizeof(double), ctx->stream));
    if (metric == HSD_METRIC_WASSERSTEIN) {
        const int nl_hist = nl - (layer_lo == 0 ? 1 : 0);
        rc = align == HSD_ALIGN_NONE ? prepare_w1_raw(ctx, p, grid, block, nl_hist) : prepare_w1_zeropad(ctx, p, grid, block, nl_hist);
        ctx->d_Y = nullptr;
    } else {
        rc = prepare_kl(ctx, p, grid, block, nl, metric, kl_alpha, kl_sorted);
    }
    if (rc) return rc;
    int flag[4] = {0, 0, 0, 0};
    HSD_HIP(ctx, hipMemcpyAsync(flag, ctx->d_flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    HSD_HIP(ctx, hipGetLastError());
    // ring-0 point values in use? (W1: d_H is the point array; KL: prepare_kl decided, the point array follows the entropy term)
    if (metric == HSD_METRIC_WASSERSTEIN) ctx->has_point = layer_lo == 0;
    if (metric == HSD_METRIC_WASSERSTEIN && align == HSD_ALIGN_NONE && (flag[0] & 1))
        return hsd_fail(ctx, HSD_ERR_EMPTY_RING, "Distribution can't be empty.");
    if (metric != HSD_METRIC_WASSERSTEIN && (flag[0] & 2))
        return hsd_fail(ctx, HSD_ERR_NOT_NORMALISED, "The sum of probability distribution must be 1.0.");
    return HSD_OK;
}

}  // namespace

extern "C" int hsd_pair_prepare(hsd_ctx *ctx, int metric, int align, int layer_lo, int layer_hi, double kl_alpha,
                                int kl_sorted) try {
    HSD_CHECK_CTX(ctx);
    if (!ctx->d_counts) return hsd_fail(ctx, HSD_ERR_NO_SIGNATURE, "no signature built");
    if (layer_lo < 0 || layer_hi > ctx->hop || layer_lo > layer_hi)
        return hsd_fail(ctx, HSD_ERR_INVALID, "layers [%d,%d] outside [0,%d]", layer_lo, layer_hi, ctx->hop);
    if (metric != HSD_METRIC_WASSERSTEIN && metric != HSD_METRIC_KL && metric != HSD_METRIC_SYMMETRIC_KL)
        return hsd_fail(ctx, HSD_ERR_UNSUPPORTED, "metric %d is not implemented.", metric);
    if (align != HSD_ALIGN_NONE && align != HSD_ALIGN_ZEROPAD) return hsd_fail(ctx, HSD_ERR_INVALID, "bad align %d", align);
    if (!(kl_alpha >= 0.0)) return hsd_fail(ctx, HSD_ERR_INVALID, "kl_alpha must be >= 0");
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hsd_release_operands(ctx);
    // a failure on any path leaves no half-built operands behind (mode stays PAIR_NONE)
    const int rc = pair_prepare_impl(ctx, metric, align, layer_lo, layer_hi, kl_alpha, kl_sorted);
    if (rc != HSD_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        hsd_release_operands(ctx);
    }
    return rc;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_pair_operand_dims(hsd_ctx *ctx, int *n_columns, int *bytes_per_element) try {
    HSD_CHECK_CTX(ctx);
    if (ctx->mode == PAIR_NONE) return hsd_fail(ctx, HSD_ERR_NO_SIGNATURE, "hsd_pair_prepare has not been called");
    if (n_columns) *n_columns = ctx->dcols_alg + (ctx->has_point ? 1 : 0);
    if (bytes_per_element) *bytes_per_element = ctx->elem_bytes;
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_pair_work(hsd_ctx *ctx, double *column_pairs) try {
    HSD_CHECK_CTX(ctx);
    if (!column_pairs) return hsd_fail(ctx, HSD_ERR_INVALID, "null output");
    unsigned long long w = 0;
    HSD_HIP(ctx, hipMemcpyAsync(&w, ctx->d_work, sizeof w, hipMemcpyDeviceToHost, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *column_pairs = (double)w;
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

// Row blocks of EQUAL PLANNED COST for `world` ranks (boundaries on whole 128-node tiles): rank g's rows of the prepared distance.
// The zero-pad kinds read min(R_I, R_J) rank words per tile pair, so a block of hub rows (the first rows of a preferential-attachment
// graph) costs more per row than a block of leaves: with equal heights the first of 8 ranks took 3.27 ms where the others took 2.8 - 2.95
// (BA N = 100k, tools_dev/strong_share.py).  Cost of a row tile = sum over the column tiles of (words the planner would read + 32): the
// 32 stand for what a tile pair costs besides its operand words (folds, stores, hand-over), fitted on those per-rank times.  Kinds whose
// tiles all cost the same (raw fp64 W1, KL) get equal blocks.  Every rank computes the same partition from the same resident data.
extern "C" int hsd_pair_shard(hsd_ctx *ctx, int world, int rank, int64_t *r0, int64_t *r1) try {
    HSD_CHECK_CTX(ctx);
    if (ctx->mode == PAIR_NONE) return hsd_fail(ctx, HSD_ERR_NO_SIGNATURE, "hsd_pair_prepare has not been called");
    if (world < 1 || rank < 0 || rank >= world || !r0 || !r1) return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_pair_shard: bad world / rank %d / %d", world, rank);
    const int64_t n = ctx->n, T = (n + 127) / 128;
    std::vector<double> cost((size_t)T, 1.0);
    const bool seg = (ctx->mode == PAIR_SAD_U16 || ctx->mode == PAIR_SAD_U32) && ctx->d_rmax16 && ctx->n_seg > 0 && ctx->n_seg <= 4 && ctx->ngran16 > 0;
    const char *eb = hsd_hook(H_SHARD_BALANCE);          // 0: equal heights whatever the kind (A/B measurements)
    const bool balanced = seg && world > 1 && !(eb && atoi(eb) == 0);
    if (balanced) {
        const int S = ctx->n_seg, G = ctx->ngran16;
        std::vector<uint32_t> rm((size_t)S * G);
        HSD_HIP(ctx, hipMemcpyAsync(rm.data(), ctx->d_rmax16, rm.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const auto tile_max = [&](int s, const int32_t *gl, int cnt) {
            uint32_t m = 0, wide = 0;
            for (int e = 0; e < cnt; ++e) { const uint32_t v = rm[(size_t)s * G + gl[e]]; m = std::max(m, v & 0x7fffffffu); wide |= v & 0x80000000u; }
            return m | wide;
        };
        // row side: natural tiles (a block's own composition is chosen after the partition); column side: the tiles of the whole graph
        // in the order the plan chose -- what counts is the distribution of their maxima
        std::vector<uint32_t> RI((size_t)S * T), RJ;
        std::vector<int32_t> nat(8);
        for (int64_t t = 0; t < T; ++t) {
            int cnt = 0;
            for (int e = 0; e < 8; ++e) if (8 * t + e < G) nat[(size_t)cnt++] = (int32_t)(8 * t + e);
            for (int s = 0; s < S; ++s) RI[(size_t)s * T + t] = tile_max(s, nat.data(), cnt);
        }
        int64_t TJ = T;
        const char *es = hsd_hook(H_GRAN_SORT);
        if (!ctx->gran_order.empty() && !(es && atoi(es) == 0)) {
            TJ = (int64_t)ctx->gran_order.size() / 8;
            RJ.resize((size_t)S * TJ);
            for (int64_t t = 0; t < TJ; ++t)
                for (int s = 0; s < S; ++s) RJ[(size_t)s * TJ + t] = tile_max(s, ctx->gran_order.data() + 8 * t, 8);
        } else RJ = RI;
        const bool pack = ctx->mode == PAIR_SAD_U16;
        for (int64_t t = 0; t < T; ++t) {
            double c = 0.0;
            for (int64_t u = 0; u < TJ; ++u) {
                int w = 32;
                for (int s = 0; s < S; ++s) {
                    const uint32_t a = RI[(size_t)s * T + t], b = RJ[(size_t)s * TJ + u];
                    w += pack ? choose_segment_form<true>(ctx->seg_tab[s], true, a, b).words : choose_segment_form<false>(ctx->seg_tab[s], true, a, b).words;
                }
                c += w;
            }
            const int64_t rows = std::min<int64_t>(128, n - 128 * t);
            cost[(size_t)t] = c * (double)rows / 128.0;
        }
    } else cost[(size_t)T - 1] *= (double)(n - 128 * (T - 1)) / 128.0;
    std::vector<double> pre((size_t)T + 1, 0.0);
    for (int64_t t = 0; t < T; ++t) pre[(size_t)t + 1] = pre[(size_t)t] + cost[(size_t)t];
    const auto boundary = [&](int g) -> int64_t {           // first tile of rank g
        if (g <= 0) return 0;
        if (g >= world) return T;
        const double want = pre[(size_t)T] * (double)g / (double)world;
        int64_t b = std::lower_bound(pre.begin(), pre.end(), want) - pre.begin();
        // (the tile boundary nearest to the target)
        if (b > 0 && want - pre[(size_t)b - 1] < pre[(size_t)std::min<int64_t>(b, T)] - want) --b;
        return std::min<int64_t>(std::max<int64_t>(b, 0), T);
    };
    int64_t a = boundary(rank), b = boundary(rank + 1);
    if (b < a) b = a;
    *r0 = std::min(n, a * 128);
    *r1 = std::min(n, b * 128);
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

