// ------------------------------------------------------------------------------------------
// Fused k-nearest-neighbour consumer: what KNN_evaluate(dists,
// metric="precomputed") (main.py:29, tools/evaluate.py:61-69) needs from the distance matrix,
// without ever writing the N^2 matrix.
//   1. threshold: D[rows, first M columns] -> tau_i = k-th smallest per row (an upper bound)
//   2. sweep    : the normal tile kernel; its epilogue appends (j, v) with v <= tau_i to row i
//   3. select   : per row the k smallest (value, index) candidates
// Retries with a larger subset / capacity if a candidate list overflowed (heavy ties).
// ------------------------------------------------------------------------------------------
// Pruned k-nearest sweep (raw fp64 W1 -- Variant A -- over the whole graph).  The unpruned sweep evaluates every tile pair: 2 x 721 fp64
// instructions per pair, 215 ms at BA N = 100k, although under the true k-th-nearest thresholds an exact lower bound excludes 96 % of the
// PAIRS (tools_dev/sim/prune_pairlevel.py).  Tiles of 128 consecutive nodes are too mixed for that bound to exclude a single TILE pair; tiles
// of 128 SIMILAR nodes are not (tools_dev/sim/prune_sim_a.py: 47 % of the tile pairs survive the first thresholds, 37 % tightened ones).
// So this consumer -- which writes no matrix and is therefore free to order the nodes as it likes -- works on a node-PERMUTED copy of the
// operands:
//   features   per node, the sum of each layer's columns (|difference| = |mean_i - mean_j| of the ring's values, a lower bound of that
//              layer's W1; any grouping of the columns bounds an L1 distance from below) and the ring-0 value: tk_feature_kernel
//   order      a k-d order of the nodes on those features (host: median splits along the widest feature down to 128-node leaves)
//   operands   Xp[c][position] = X[c][node at that position], padded with zero columns to a whole number of tiles: no strips; the
//              padding positions never enter a list (orig_of < 0)
//   launch 0   the tile pairs (t, t) and (t, t + 1) under thresholds of +inf: every row collects the ~383 nodes next to it in the order --
//              its most similar ones -- and the first compaction leaves a threshold within a small factor of the true one (the sampled
//              columns of the unpruned sweep give the 2 % quantile of a row)
//   launches   the usual anti-diagonal phases, each filtered on the device (tk_filter_kernel) against the thresholds of the moment
// The candidate lists hold ORIGINAL node ids (ties are broken by them), the rows are un-permuted at the end: the result is the one the
// unpruned sweep gives, bit for bit (tests/test_gpu_topk.py).  HSD_TOPK_PRUNE=0 / 1: never / whenever the kind allows.
// ------------------------------------------------------------------------------------------
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

// ---- pruned k-nearest sweep (raw fp64 W1): features, permuted operands, tile filter ---------------------------------------------
// Any partition of the columns gives a lower bound of an L1 distance: sum_c |x_c - y_c| >= sum_g |sum_{c in g} x_c - sum_{c in g} y_c|.
// One group per layer (for the columns G = CDF x bin width that is |mean_i - mean_j| of the ring's values, the classic W1 bound) plus the
// ring-0 point value: F[f][i], f < nl: sum of layer f's columns of node i; f == nl (if H): H[i].
// (G > 1: every layer's columns in G consecutive groups -- a tighter bound, more features).  Two steps so that the sums have one fixed
// order: blockIdx.y = feature x TK_SLICES + slice -> partial sums P[feature][slice][i]; then F[feature][i] = the slices added in order.
constexpr int TK_SLICES = 16;
__global__ void tk_feature_partial_kernel(const double *__restrict__ X, int64_t npad, int64_t n, int cpl, int G, double *__restrict__ P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = blockIdx.y / TK_SLICES, sl = blockIdx.y % TK_SLICES, s = f / G, g = f % G;
    const int c0 = s * cpl + (int)((int64_t)cpl * g / G), c1 = s * cpl + (int)((int64_t)cpl * (g + 1) / G);
    const int a = c0 + (int)((int64_t)(c1 - c0) * sl / TK_SLICES), b = c0 + (int)((int64_t)(c1 - c0) * (sl + 1) / TK_SLICES);
    double t = 0.0;
    for (int c = a; c < b; ++c) t += X[(int64_t)c * npad + i];
    P[((int64_t)f * TK_SLICES + sl) * n + i] = t;
}
__global__ void tk_feature_combine_kernel(const double *__restrict__ P, int64_t n, int nfh, const double *__restrict__ H, double *__restrict__ F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int f = 0; f < nfh; ++f) {
        double t = 0.0;
        for (int sl = 0; sl < TK_SLICES; ++sl) t += P[((int64_t)f * TK_SLICES + sl) * n + i];
        F[(int64_t)f * n + i] = t;
    }
    if (H) F[(int64_t)nfh * n + i] = H[i];
}

// Xp[c][p] = X[c][orig[p]] (0 at padding positions), p < npos positions (stride npos), X with node stride npad; blockIdx.y = column
__global__ void tk_gather_kernel(const double *__restrict__ X, int64_t npad, const int32_t *__restrict__ orig, int64_t npos, double *__restrict__ Xp) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    const int32_t o = orig[p];
    const int64_t c = blockIdx.y;
    Xp[c * npos + p] = o >= 0 ? X[c * npad + o] : 0.0;
}

// thresholds by node -> thresholds by position (0 at padding positions: nothing is at distance <= 0 from them but themselves)
__global__ void tk_permute_tau_kernel(const int32_t *__restrict__ orig, int64_t npos, const double *__restrict__ tau_node, double *__restrict__ tau_pos) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    const int32_t o = orig[p];
    tau_pos[p] = o >= 0 ? tau_node[o] : 0.0;
}

// thresholds of +inf for the positions of real nodes, 0 for padding positions (their rows collect nothing and never hold a tile's threshold open)
__global__ void tk_tau_open_kernel(const int32_t *__restrict__ orig, int64_t npos, double *__restrict__ tau_pos) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < npos) tau_pos[p] = orig[p] >= 0 ? (double)INFINITY : 0.0;
}

// out[orig[p]] = in[p] for the rows p of real nodes (k entries per row)
__global__ void tk_unpermute_kernel(const int32_t *__restrict__ orig, int64_t npos, int k, const int *__restrict__ idx_in, const double *__restrict__ val_in,
                                    int *__restrict__ idx_out, double *__restrict__ val_out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npos * k) return;
    const int64_t p = e / k, q = e % k;
    const int32_t o = orig[p];
    if (o < 0) return;
    idx_out[(int64_t)o * k + q] = idx_in[e];
    val_out[(int64_t)o * k + q] = val_in[e];
}

// The next (value, index) pair of a list in lexicographic order after (last_v, last_i), found by ONE WAVE: every lane scans
// its share of the list, six exchange steps pick the minimum.  ci == nullptr: the position is the index (plain values).
__device__ inline void wave_lex_next(const double *__restrict__ cv, const int *__restrict__ ci, int c, double last_v, int last_i, int lane,
                                     double &ov, int &oi) {
    double bv = INFINITY;
    int bi = 0x7fffffff;
    for (int t = lane; t < c; t += 64) {
        const double v = cv[t];
        const int i = ci ? ci[t] : t;
        const bool after = v > last_v || (v == last_v && i > last_i);
        if (after && (v < bv || (v == bv && i < bi))) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v = __shfl_xor(bv, o, 64);
        const int i = __shfl_xor(bi, o, 64);
        if (v < bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    ov = bv;
    oi = bi;
}

// The same with the list held in registers (E entries per lane, loaded once) as order-preserving integer keys, each lane's
// entries SORTED (odd-even transposition, once): a round is then one comparison of the lanes' heads over six exchange steps
// plus a shift in the winning lane, instead of E lexicographic fp64 comparisons per lane.
__device__ inline unsigned long long topk_key(double v) {            // monotone map of the doubles onto the unsigned integers
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double topk_unkey(unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}
template <int E>
struct WaveList {
    unsigned long long key[E];
    int i[E];
    __device__ inline void load(const double *__restrict__ cv, const int *__restrict__ ci, int c, int lane) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int t = lane + 64 * e;
            key[e] = t < c ? topk_key(cv[t]) : ~0ull;
            i[e] = t < c ? (ci ? ci[t] : t) : 0x7fffffff;
        }
#pragma unroll
        for (int p = 0; p < E; ++p)
#pragma unroll
            for (int e = p & 1; e + 1 < E; e += 2) {
                const bool sw = key[e + 1] < key[e] || (key[e + 1] == key[e] && i[e + 1] < i[e]);
                const unsigned long long k0 = sw ? key[e + 1] : key[e], k1 = sw ? key[e] : key[e + 1];
                const int i0 = sw ? i[e + 1] : i[e], i1 = sw ? i[e] : i[e + 1];
                key[e] = k0; key[e + 1] = k1; i[e] = i0; i[e + 1] = i1;
            }
    }
    // the smallest entry still held by any lane; it is removed
    __device__ inline void pop(double &ov, int &oi) {
        unsigned long long bk = key[0];
        int bi = i[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long xk = __shfl_xor(bk, o, 64);
            const int xi = __shfl_xor(bi, o, 64);
            if (xk < bk || (xk == bk && xi < bi)) { bk = xk; bi = xi; }
        }
        if (bk == key[0] && bi == i[0]) {
#pragma unroll
            for (int e = 0; e + 1 < E; ++e) { key[e] = key[e + 1]; i[e] = i[e + 1]; }
            key[E - 1] = ~0ull;
            i[E - 1] = 0x7fffffff;
        }
        ov = topk_unkey(bk);
        oi = bi;
    }
};

// A list of up to 64 E (value, index) entries SORTED across the wave (round 5): element p = 64 e + lane of a bitonic network over 64 E places -- a
// compare-exchange with a partner in another lane is three cross-lane moves, one in the same lane none; 21 / 28 / 36 / 45 stages for E = 1 / 2 / 4 / 8
// against k rounds of a six-step wave minimum (k = 20: 360 dependent cross-lane moves).  (value key, index) is a total order (the indices of a list are
// distinct), so the first k places hold exactly what k pops of WaveList return, in the same order.
template <int E>
struct WaveSort {
    unsigned long long key[E];
    int i[E];
    __device__ inline void load_and_sort(const double *__restrict__ cv, const int *__restrict__ ci, int c, int lane) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int t = lane + 64 * e;
            key[e] = t < c ? topk_key(cv[t]) : ~0ull;
            i[e] = t < c ? (ci ? ci[t] : t) : 0x7fffffff;
        }
#pragma unroll
        for (int k = 2; k <= 64 * E; k <<= 1)
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j >= 64) {
                    const int je = j >> 6;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if ((e & je) == 0) {
                            const int f = e | je;
                            const bool asc = ((64 * e) & k) == 0;          // (the direction bit of a place: lane bits are below k here)
                            const bool sw = key[f] < key[e] || (key[f] == key[e] && i[f] < i[e]);      // f's entry is the smaller
                            if (sw == asc) {
                                const unsigned long long tk = key[e]; key[e] = key[f]; key[f] = tk;
                                const int ti = i[e]; i[e] = i[f]; i[f] = ti;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const unsigned long long xk = __shfl_xor(key[e], j, 64);
                        const int xi = __shfl_xor(i[e], j, 64);
                        const int pl = 64 * e + lane;
                        const bool lower = (lane & j) == 0, asc = (pl & k) == 0;
                        const bool other_smaller = xk < key[e] || (xk == key[e] && xi < i[e]);
                        // the lower place of an ascending pair keeps the smaller entry
                        if ((lower == asc) == other_smaller) { key[e] = xk; i[e] = xi; }
                    }
                }
            }
    }
};

// the first k places of a sorted list -> out (values as doubles); place p sits in lane p % 64, slot p / 64
template <int E>
__device__ inline void wavesort_emit(const WaveSort<E> &ws, int k, int lane, double *__restrict__ ov, int *__restrict__ oi) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int pl = 64 * e + lane;
        if (pl < k) { ov[pl] = topk_unkey(ws.key[e]); oi[pl] = ws.i[e]; }
    }
}

// tau[row] = the (k+1)-th smallest of the row's M sampled distances, slightly inflated.  One wave per row, k + 1 rounds of
// "next smallest" (k is small against M; a bitonic sort of the sample in LDS took 2.2 ms for 100 000 rows of 1024).
// The row's own node may be among the sampled columns (its distance is an exact 0 and comes first): the (k+1)-th smallest
// keeps tau an upper bound of the k-th nearest OTHER node whether or not it is.  Inflated: the sweep may evaluate a pair
// from the mirrored tile, whose fp64 sums are accumulated in another order (symmetric KL).
// slack: absolute allowance on top (symmetric KL: H_i + H_j - dots cancels, so two evaluation orders of a pair of nearly
// identical nodes differ by ~1e-15 ABSOLUTE, which no relative inflation of a tiny distance covers).
__global__ __launch_bounds__(256) void topk_threshold_kernel(const double *__restrict__ S, int64_t ld, int M, int64_t row0, int64_t rows, int k,
                                                            double slack, double *__restrict__ tau) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const double *sv = S + r * ld;
    double last_v = -INFINITY;
    int last_i = -1;
    if (M <= 1024) {
        WaveList<16> wl;
        wl.load(sv, nullptr, M, lane);
        for (int q = 0; q <= k; ++q) wl.pop(last_v, last_i);
    } else {
        for (int q = 0; q <= k; ++q) wave_lex_next(sv, nullptr, M, last_v, last_i, lane, last_v, last_i);
    }
    if (lane == 0) tau[row0 + r] = last_v + fabs(last_v) * 1e-12 + slack;
}

__global__ void fill_columns_kernel(double *__restrict__ S, int64_t ld, int64_t rows, int c_begin, int c_end, double v) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    for (int c = c_begin; c < c_end; ++c) S[r * ld + c] = v;
}

// The k smallest (value, index) pairs of each row's candidate list, ascending, ties by index: one wave per row, k rounds
// of "next lexicographic minimum" (the lists are short).  flag[0] |= 1 on overflow or on a list shorter than k.
__global__ __launch_bounds__(256) void topk_select_kernel(const int *__restrict__ cnt, const int *__restrict__ cidx,
                                                         const double *__restrict__ cval, int cap, int64_t row0, int64_t rows, int k,
                                                         int *__restrict__ out_idx, double *__restrict__ out_val, int *__restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int64_t lr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lr >= rows) return;
    const int64_t row = row0 + lr;
    const int c = cnt[row];
    if (c > cap || c < k) {
        if (lane == 0) atomicOr(flag, 1);
        return;
    }
    const int *ci = cidx + row * cap;
    const double *cv = cval + row * cap;
    double last_v = -INFINITY;
    int last_i = -1;
    if (c <= 512) {
        double *ov = out_val + lr * (int64_t)k;
        int *oi = out_idx + lr * (int64_t)k;
        if (c <= 64) { WaveSort<1> ws; ws.load_and_sort(cv, ci, c, lane); wavesort_emit(ws, k, lane, ov, oi); }
        else if (c <= 128) { WaveSort<2> ws; ws.load_and_sort(cv, ci, c, lane); wavesort_emit(ws, k, lane, ov, oi); }
        else if (c <= 256) { WaveSort<4> ws; ws.load_and_sort(cv, ci, c, lane); wavesort_emit(ws, k, lane, ov, oi); }
        else { WaveSort<8> ws; ws.load_and_sort(cv, ci, c, lane); wavesort_emit(ws, k, lane, ov, oi); }
        return;
    }
    for (int q = 0; q < k; ++q) {
        wave_lex_next(cv, ci, c, last_v, last_i, lane, last_v, last_i);
        if (lane == 0) { out_idx[lr * (int64_t)k + q] = last_i; out_val[lr * (int64_t)k + q] = last_v; }
    }
}

// Between two phases of the sweep: a row whose list has grown beyond k entries keeps its k smallest (value, index) pairs
// and takes the k-th value as its new threshold -- still an upper bound of its k-th nearest, and every later candidate
// must beat it.  One wave per row.  Lists that overflowed are left alone (the select kernel reports them, the host retries).
__global__ __launch_bounds__(256) void topk_compact_kernel(int *__restrict__ cnt, int *__restrict__ cidx, double *__restrict__ cval, int cap,
                                                          int64_t rows, int k, double slack, double *__restrict__ tau) {
    extern __shared__ double cmp_sh[];                 // per wave: k values, then k indices
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *sel_v = cmp_sh + (size_t)w * k;
    int *sel_i = reinterpret_cast<int *>(cmp_sh + (size_t)4 * k) + (size_t)w * k;
    const int64_t row = (int64_t)blockIdx.x * 4 + w;
    if (row >= rows) return;
    const int c = cnt[row];
    if (c > cap || c <= k) return;
    int *ci = cidx + row * cap;
    double *cv = cval + row * cap;
    double last_v = -INFINITY;
    int last_i = -1;
    if (c <= 512) {
        // (k < c <= 512: sorted across the wave, the first k places into the LDS staging the copy below reads)
        if (c <= 64) { WaveSort<1> ws; ws.load_and_sort(cv, ci, c, lane); wavesort_emit(ws, k, lane, sel_v, sel_i); }
        else if (c <= 128) { WaveSort<2> ws; ws.load_and_sort(cv, ci, c, lane); wavesort_emit(ws, k, lane, sel_v, sel_i); }
        else if (c <= 256) { WaveSort<4> ws; ws.load_and_sort(cv, ci, c, lane); wavesort_emit(ws, k, lane, sel_v, sel_i); }
        else { WaveSort<8> ws; ws.load_and_sort(cv, ci, c, lane); wavesort_emit(ws, k, lane, sel_v, sel_i); }
        __builtin_amdgcn_wave_barrier();
        last_v = sel_v[k - 1];
    } else {
        for (int q = 0; q < k; ++q) {
            wave_lex_next(cv, ci, c, last_v, last_i, lane, last_v, last_i);
            if (lane == 0) { sel_v[q] = last_v; sel_i[q] = last_i; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < k; t += 64) { cv[t] = sel_v[t]; ci[t] = sel_i[t]; }
    if (lane == 0) {
        cnt[row] = k;
        tau[row] = fmin(tau[row], last_v + fabs(last_v) * 1e-12 + slack);   // (inflated like the first threshold: mirrored evaluation order)
    }
}


// k-d order of the nodes on nf features F[f][i]: split the widest feature at the median (rounded to whole leaves), down to 128-node leaves; ties by
// node id.  Node-major copy of the features (one node's features in one cache line); once the ranges are short enough their subtrees go to host
// threads (disjoint ranges of idx: no sharing; a thread that cannot be created leaves its range to the caller, as in hsd_pair_prepare).
void kd_order_host(const std::vector<double> &F, int64_t n, int nf, std::vector<int32_t> &idx) {
    idx.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) idx[(size_t)i] = (int32_t)i;
    std::vector<double> A((size_t)n * nf);
    for (int f = 0; f < nf; ++f)
        for (int64_t i = 0; i < n; ++i) A[(size_t)i * nf + f] = F[(size_t)f * n + i];
    const double *Ap = A.data();
    int32_t *ip = idx.data();
    const auto split = [Ap, ip, nf](int64_t lo, int64_t hi) -> int64_t {       // returns the split point, or -1 for a leaf
        if (hi - lo <= 128) return -1;
        int best = 0;
        double ext = -1.0;
        for (int f = 0; f < nf; ++f) {
            double mn = INFINITY, mx = -INFINITY;
            for (int64_t q = lo; q < hi; ++q) { const double v = Ap[(size_t)ip[q] * nf + f]; mn = std::min(mn, v); mx = std::max(mx, v); }
            if (mx - mn > ext) { ext = mx - mn; best = f; }
        }
        int64_t mid = lo + ((hi - lo) / 2 + 127) / 128 * 128;
        if (mid >= hi) mid = lo + (hi - lo) / 2;
        std::nth_element(ip + lo, ip + mid, ip + hi, [Ap, nf, best](int32_t x, int32_t y) {
            const double a = Ap[(size_t)x * nf + best], b = Ap[(size_t)y * nf + best];
            return a < b || (a == b && x < y);
        });
        return mid;
    };
    const auto subtree = [&split](int64_t lo0, int64_t hi0) noexcept {
        try {
            std::vector<std::pair<int64_t, int64_t>> todo{{lo0, hi0}};
            while (!todo.empty()) {
                const auto [lo, hi] = todo.back();
                todo.pop_back();
                const int64_t mid = split(lo, hi);
                if (mid < 0) continue;
                todo.push_back({lo, mid});
                todo.push_back({mid, hi});
            }
        } catch (...) {}                     // (out of memory for the work list: the range stays as it is -- a worse order, the same results)
    };
    // A shared work list: a range longer than 4 096 nodes is split by whoever takes it and its halves go back on the list; shorter ones are finished
    // on the spot.  (Splitting the top of the tree on one thread first -- four full passes over the nodes -- took 7 of the 10 ms of the setup.)
    const int thr_cap = std::max(1, hsd_hook_int(H_PLAN_THREADS, 16));
    const int n_thr = std::max(1, std::min<int>(thr_cap, std::max(1, (int)std::thread::hardware_concurrency())));
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::pair<int64_t, int64_t>> list{{0, n}};
    int busy = 0;                                    // ranges taken and not yet finished (their halves may still arrive)
    const auto work = [&]() noexcept {
        for (;;) {
            std::pair<int64_t, int64_t> r;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !list.empty() || busy == 0; });
                if (list.empty()) return;            // nothing listed, nobody splitting: done
                r = list.back();
                list.pop_back();
                ++busy;
            }
            if (r.second - r.first <= 4096) subtree(r.first, r.second);
            else {
                const int64_t mid = split(r.first, r.second);
                if (mid >= 0) {
                    std::lock_guard<std::mutex> lk(mu);
                    try { list.push_back({r.first, mid}); list.push_back({mid, r.second}); } catch (...) {}      // (no memory: a worse order, the same results)
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                --busy;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    try {
        pool.reserve((size_t)n_thr);
        for (int t = 1; t < n_thr; ++t) pool.emplace_back(work);
    } catch (...) {}
    work();
    for (auto &th : pool) th.join();
}

// zero-pad kinds: F[s][i] = the sum of layer s's bin-form columns of node i = the sum of the ring's member values (|su_i - su_j| <= S_s(i, j))
template <bool PACK>
__global__ void tk_feature_seg_kernel(const uint32_t *__restrict__ X, int64_t npad, int64_t n, const HsdSegDesc *__restrict__ tab, double *__restrict__ F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const HsdSegDesc d = tab[blockIdx.y];
    double t = 0.0;
    for (int w = 0; w < d.bin_words; ++w) {
        const uint32_t v = X[(int64_t)(d.bin_col0 + w) * npad + i];
        t += PACK ? (double)((v & 0xffffu) + (v >> 16)) : (double)v;
    }
    F[(int64_t)blockIdx.y * n + i] = t;
}

// Up[c][p] = U[c][orig[p]] (0 at padding positions), 32-bit words; blockIdx.y = column / row
__global__ void tk_gather_u32_kernel(const uint32_t *__restrict__ X, int64_t npad, const int32_t *__restrict__ orig, int64_t npos, uint32_t *__restrict__ Xp) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    const int32_t o = orig[p];
    const int64_t c = blockIdx.y;
    Xp[c * npos + p] = o >= 0 ? X[c * npad + o] : 0u;
}

// a row block's positions: orig_ext[npad + q] = orig[pos[q]], q < m (the rest of the block's last tile stays -1)
__global__ void tk_block_orig_kernel(const int32_t *__restrict__ orig, const int32_t *__restrict__ pos, int64_t m, int32_t *__restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < m) out[q] = orig[pos[q]];
}

// members above a layer's threshold per POSITION, recovered from the permuted operands: the non-zero rank values of the stored rank words, and
// "more than they hold" (any value that sends the planner to the bin form) where the suffix behind the last stored word is not empty
template <bool PACK>
__global__ void tk_rnode_kernel(const uint32_t *__restrict__ Xp, const uint32_t *__restrict__ sufp, int64_t npad, const HsdSegDesc *__restrict__ tab,
                                uint32_t *__restrict__ R) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npad) return;
    const HsdSegDesc d = tab[blockIdx.y];
    uint32_t cnt = 0;
    if (d.rank_col0 >= 0 && d.suf_row0 >= 0) {
        for (int w = 0; w < d.rank_cap; ++w) {
            const uint32_t v = Xp[(int64_t)(d.rank_col0 + d.rank_low + w) * npad + p];
            cnt += PACK ? (uint32_t)((v & 0xffffu) != 0u) + (uint32_t)((v >> 16) != 0u) : (uint32_t)(v != 0u);
        }
        if (sufp[(int64_t)(d.suf_row0 + d.rank_cap) * npad + p] > 0u) cnt = PACK ? 2u * (uint32_t)d.rank_cap + 2u : (uint32_t)d.rank_cap + 1u;
    }
    R[(int64_t)blockIdx.y * npad + p] = cnt;
}

// bit 31 of a granule's maximum: one of its 16 positions is too wide for the layer's uint8 range (as hsd_pair_prepare marks them)
__global__ void tk_wide_kernel(const uint32_t *__restrict__ need, int64_t npad, int ngran16, int best_b, int best_r, uint32_t *__restrict__ g16) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngran16) return;
    bool wide = false;
    for (int e = 0; e < 16; ++e) wide = wide || (int)need[(int64_t)16 * g + e] > best_b || (int)need[npad + (int64_t)16 * g + e] > best_r;
    if (wide) g16[g] |= 0x80000000u;
}

// Fo[f][i] = V[f][i] x the layer's factor (1 / ring size; 1 for the ring-0 value): what the nodes are ordered on
__global__ void tk_scale_features_kernel(const double *__restrict__ V, const double *__restrict__ inv, int64_t npad, int64_t n, int n_scaled, double *__restrict__ Fo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = blockIdx.y;
    const double v = V[(int64_t)f * n + i];
    Fo[(int64_t)f * n + i] = f < n_scaled ? v * inv[(int64_t)f * npad + i] : v;
}

// per tile of the node order and feature: [min, max] of V over the tile's nodes and their smallest factor (+inf / -inf / +inf for a tile of padding only);
// 128 threads = one tile, blockIdx.y = feature
__global__ void tk_tile_interval_kernel(const double *__restrict__ V, const double *__restrict__ inv, int64_t npad, int64_t n, int n_scaled, const int32_t *__restrict__ orig,
                                        int nt, double *__restrict__ lo, double *__restrict__ hi, double *__restrict__ sc) {
    __shared__ double part[3][2];
    const int t = blockIdx.x, f = blockIdx.y;
    const int32_t o = orig[(int64_t)t * 128 + threadIdx.x];
    double mn = INFINITY, mx = -INFINITY, s = INFINITY;
    if (o >= 0) {
        mn = mx = V[(int64_t)f * n + o];
        s = f < n_scaled ? inv[(int64_t)f * npad + o] : 1.0;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, d, 64));
        mx = fmax(mx, __shfl_xor(mx, d, 64));
        s = fmin(s, __shfl_xor(s, d, 64));
    }
    if ((threadIdx.x & 63u) == 0u) { part[0][threadIdx.x >> 6] = mn; part[1][threadIdx.x >> 6] = mx; part[2][threadIdx.x >> 6] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo[(int64_t)f * nt + t] = fmin(part[0][0], part[0][1]);
        hi[(int64_t)f * nt + t] = fmax(part[1][0], part[1][1]);
        sc[(int64_t)f * nt + t] = fmin(part[2][0], part[2][1]);
    }
}

// [b0, b1): the row block the sweep is for (b0 == b1 == 0: the whole graph).  The permuted operand set then has a second run of positions behind
// the graph's npad: the block's rows in the order the graph's k-d order gives them (a subsequence of it), padded to whole tiles -- the sweep's
// rows; its columns are the graph's positions.  One setup is kept per prepared distance: the last block asked for.
int tk_prepare(hsd_ctx *ctx, int64_t b0, int64_t b1) {
    if (ctx->tk_ready && ctx->tk_blk0 == b0 && ctx->tk_blk1 == b1) return HSD_OK;
    // (another block's setup, or whatever an earlier, failed attempt allocated -- the graph's node order, the expensive part, is kept: a caller that
    //  walks over row blocks pays ~1 ms per new block at 100k nodes instead of 3-6)
    hsd_tk_release_block(ctx);
    const bool verbose = hsd_hook(H_TOPK_VERBOSE) != nullptr;
    if (verbose) (void)hipStreamSynchronize(ctx->stream);          // (whatever hsd_pair_prepare left in flight is not this setup's time)
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!verbose) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[hsd topk] setup: %-34s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const int64_t n = ctx->n, npad = ctx->npad;
    const bool seg = ctx->mode == PAIR_SAD_U16 || ctx->mode == PAIR_SAD_U32;
    const bool pack = ctx->mode == PAIR_SAD_U16;
    if (npad % 128 != 0) return hsd_fail(ctx, HSD_ERR_INVALID, "pruned k-nearest sweep: operand stride is no whole number of tiles");
    const int64_t bm = b1 - b0, bpad = (bm + 127) / 128 * 128, npos = npad + bpad;     // positions: the graph's order, then the block's rows
    const int nt = (int)(npos / 128), nt_graph = (int)(npad / 128);
    // bound features V[f][i] with their factors S[f][i] (1 for the raw distance and the ring-0 value; 1 / ring size for a zero-pad layer: the
    // layer's distance is at least |V_i - V_j| min(S_i, S_j)); the nodes are ordered on V * S
    int nf = 0, nlay = 0;
    std::vector<double> V, S;                          // (host copies: only for the host order, HSD_TOPK_ORDER=0)
    DevTmp dV;                                         // the features on the device, nf rows of n; the zero-pad layers' factors are ctx->d_inv's rows
    const bool dev_order = hsd_hook_int(H_TOPK_ORDER, 1) != 0;
    if (!seg) {
        const int cpl = ctx->l1_cols_per_layer, nl = cpl > 0 ? ctx->dcols_alg / cpl : 0;
        const int G = std::max(1, std::min(8, hsd_hook_int(H_TOPK_GROUPS, 1)));
        nlay = nl * G;
        nf = nlay + (ctx->has_point ? 1 : 0);
        if (nf < 1) return hsd_fail(ctx, HSD_ERR_INVALID, "pruned k-nearest sweep: nothing to order the nodes by");
        DevTmp dP;
        HSD_TMP_ALLOC(ctx, dV, (size_t)nf * n * sizeof(double));
        HSD_TMP_ALLOC(ctx, dP, (size_t)std::max(nlay, 1) * TK_SLICES * n * sizeof(double));
        if (nlay > 0)
            hipLaunchKernelGGL(tk_feature_partial_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(nlay * TK_SLICES)), dim3(256), 0, ctx->stream,
                               (const double *)ctx->d_X, npad, n, cpl, G, dP.as<double>());
        hipLaunchKernelGGL(tk_feature_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dP.as<double>(), n, nlay,
                           ctx->has_point ? (const double *)ctx->d_H : nullptr, dV.as<double>());
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));           // (dP goes)
    } else {
        nlay = ctx->n_seg;
        nf = nlay + (ctx->has_point ? 1 : 0);
        if (nlay < 1 || !ctx->d_inv) return hsd_fail(ctx, HSD_ERR_INVALID, "pruned k-nearest sweep: no zero-pad layer to order the nodes by");
        HSD_TMP_ALLOC(ctx, dV, (size_t)nf * n * sizeof(double));
        if (pack) hipLaunchKernelGGL(tk_feature_seg_kernel<true>, dim3((unsigned)((n + 255) / 256), (unsigned)nlay), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_X, npad, n, ctx->d_seg_tab, dV.as<double>());
        else hipLaunchKernelGGL(tk_feature_seg_kernel<false>, dim3((unsigned)((n + 255) / 256), (unsigned)nlay), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_X, npad, n, ctx->d_seg_tab, dV.as<double>());
        if (ctx->has_point)
            HSD_HIP(ctx, hipMemcpyAsync(dV.as<double>() + (size_t)nlay * n, ctx->d_H, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (!dev_order) {
        V.resize((size_t)nf * n);
        HSD_HIP(ctx, hipMemcpyAsync(V.data(), dV.p, V.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (seg) {
            S.assign((size_t)nf * n, 1.0);
            for (int s2 = 0; s2 < nlay; ++s2)
                HSD_HIP(ctx, hipMemcpyAsync(S.data() + (size_t)s2 * n, ctx->d_inv + (size_t)s2 * npad, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        }
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    mark("features");
    int rc;
    // the block's rows behind the graph's positions: the positions of the graph's order that hold a node of [b0, b1), ascending (their nodes
    // -> d_tk_orig[npad ...]; the positions themselves -> the host, for the first launch's tile list)
    std::vector<int32_t> block_pos;
    const auto block_positions = [&]() -> int {
        if (bm == 0) return HSD_OK;
        DevTmp dpos;
        HSD_TMP_ALLOC(ctx, dpos, (size_t)bm * sizeof(int32_t));
        const int r = hsd_tk_block_positions_device(ctx, ctx->d_tk_orig, npad, b0, b1, dpos.as<int32_t>());
        if (r) return r;
        hipLaunchKernelGGL(tk_block_orig_kernel, dim3((unsigned)((bm + 255) / 256)), dim3(256), 0, ctx->stream, (const int32_t *)ctx->d_tk_orig, (const int32_t *)dpos.as<int32_t>(), bm,
                           ctx->d_tk_orig + npad);
        block_pos.resize((size_t)bm);
        HSD_HIP(ctx, hipMemcpyAsync(block_pos.data(), dpos.p, (size_t)bm * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        mark("block rows in the graph's order");
        return HSD_OK;
    };
    if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_orig, (size_t)npos))) return rc;
    if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_lo, (size_t)nf * nt))) return rc;
    if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_hi, (size_t)nf * nt))) return rc;
    if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_scale, (size_t)nf * nt))) return rc;
    if (dev_order) {
        // the order on the device (csrc/tkorder.hip), on V x factor (zero-pad layers: the ring's MEAN value); then every tile's feature intervals
        DevTmp dFo;
        const double *Fo = dV.as<double>();
        if (seg) {
            HSD_TMP_ALLOC(ctx, dFo, (size_t)nf * n * sizeof(double));
            hipLaunchKernelGGL(tk_scale_features_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nf), dim3(256), 0, ctx->stream, dV.as<double>(), (const double *)ctx->d_inv,
                               npad, n, nlay, dFo.as<double>());
            Fo = dFo.as<double>();
        }
        HSD_HIP(ctx, hipMemsetAsync(ctx->d_tk_orig, 0xff, (size_t)npos * sizeof(int32_t), ctx->stream));          // padding positions: -1
        if (!ctx->tk_order_ready) {
            if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_order, (size_t)npad))) return rc;
            HSD_HIP(ctx, hipMemsetAsync(ctx->d_tk_order, 0xff, (size_t)npad * sizeof(int32_t), ctx->stream));
            if ((rc = hsd_tk_order_device(ctx, Fo, n, nf, ctx->d_tk_order))) return rc;
            ctx->tk_order_ready = true;
            mark("k-d order (device)");
        }
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_tk_orig, ctx->d_tk_order, (size_t)npad * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
        if ((rc = block_positions())) return rc;
        hipLaunchKernelGGL(tk_tile_interval_kernel, dim3((unsigned)nt, (unsigned)nf), dim3(128), 0, ctx->stream, dV.as<double>(), seg ? (const double *)ctx->d_inv : nullptr,
                           npad, n, seg ? nlay : 0, (const int32_t *)ctx->d_tk_orig, nt, ctx->d_tk_lo, ctx->d_tk_hi, ctx->d_tk_scale);
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));           // (dFo goes)
    } else {
        hsd_op_free(ctx, ctx->d_tk_order);              // (the host path keeps no order: HSD_TOPK_ORDER=0 is an A/B hook)
        ctx->tk_order_ready = false;
        std::vector<int32_t> idx;
        {
            std::vector<double> Fo = V;
            if (seg) for (size_t q = 0; q < Fo.size(); ++q) Fo[q] *= S[q];
            kd_order_host(Fo, n, nf, idx);
        }
        mark("k-d order (host)");
        std::vector<int32_t> orig((size_t)npos, -1);
        std::copy(idx.begin(), idx.end(), orig.begin());
        HSD_HIP(ctx, hipMemcpy(ctx->d_tk_orig, orig.data(), orig.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        if ((rc = block_positions())) return rc;
        if (bm > 0) HSD_HIP(ctx, hipMemcpy(orig.data() + npad, ctx->d_tk_orig + npad, (size_t)bpad * sizeof(int32_t), hipMemcpyDeviceToHost));
        std::vector<double> flo((size_t)nf * nt, INFINITY), fhi((size_t)nf * nt, -INFINITY), fsc((size_t)nf * nt, INFINITY);
        for (int64_t p = 0; p < npos; ++p)
            for (int f = 0; f < nf && orig[(size_t)p] >= 0; ++f) {
                const size_t src = (size_t)f * n + orig[(size_t)p], dst = (size_t)f * nt + p / 128;
                flo[dst] = std::min(flo[dst], V[src]); fhi[dst] = std::max(fhi[dst], V[src]);
                fsc[dst] = std::min(fsc[dst], seg ? S[src] : 1.0);
            }
        HSD_HIP(ctx, hipMemcpy(ctx->d_tk_lo, flo.data(), flo.size() * sizeof(double), hipMemcpyHostToDevice));
        HSD_HIP(ctx, hipMemcpy(ctx->d_tk_hi, fhi.data(), fhi.size() * sizeof(double), hipMemcpyHostToDevice));
        HSD_HIP(ctx, hipMemcpy(ctx->d_tk_scale, fsc.data(), fsc.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const int band_w = std::max(1, std::min(4, hsd_hook_int(H_TOPK_BAND, 1)));      // the first launch: every tile with itself and its band_w successors
    std::vector<int2> band;
    if (bm == 0) {
        for (int t = 0; t < nt; ++t)
            for (int d = 0; d <= band_w && t + d < nt; ++d) band.push_back(make_int2(t, t + d));
    } else {
        // a row block: block tile t (its rows are consecutive members of the graph's order) with the graph's tiles its rows sit in and band_w more
        // on either side -- tile indices RELATIVE to the region (rows: the block's tiles; columns: the graph's), as the region's lists are
        const int bt = (int)(bpad / 128);
        std::vector<int32_t> blo((size_t)bt), bhi((size_t)bt);
        for (int t = 0; t < bt; ++t) {
            const int64_t q0 = (int64_t)t * 128, q1 = std::min<int64_t>(bm, q0 + 128) - 1;
            blo[(size_t)t] = std::max(0, block_pos[(size_t)q0] / 128 - band_w);
            bhi[(size_t)t] = std::min(nt_graph - 1, block_pos[(size_t)q1] / 128 + band_w);
            // Where the block is thin in the graph's order its 128 rows span dozens of tiles, and every row of the tile would collect every column of
            // that span that passes its sampled (2 % quantile) threshold: at most TK_BAND_MAX tiles -- 1 536 columns, below the lists' capacity
            // of 2 048 whatever passes (seen without the cap at BA N = 200k, P = 8: 3 947 entries in the first launch, the list lost, the call
            // repeated unpruned) -- around the tile of the middle row; the rest of the span comes with the phases like every other tile pair.
            constexpr int TK_BAND_MAX = 12;
            if (bhi[(size_t)t] - blo[(size_t)t] + 1 > TK_BAND_MAX) {
                const int mid = block_pos[(size_t)((q0 + q1) / 2)] / 128;
                const int lo = std::max(blo[(size_t)t], std::min(mid - TK_BAND_MAX / 2, bhi[(size_t)t] - TK_BAND_MAX + 1));
                blo[(size_t)t] = lo;
                bhi[(size_t)t] = lo + TK_BAND_MAX - 1;
            }
            for (int c = blo[(size_t)t]; c <= bhi[(size_t)t]; ++c) band.push_back(make_int2(t, c));
        }
        if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_band_lo, (size_t)bt))) return rc;
        if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_band_hi, (size_t)bt))) return rc;
        HSD_HIP(ctx, hipMemcpy(ctx->d_tk_band_lo, blo.data(), blo.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HSD_HIP(ctx, hipMemcpy(ctx->d_tk_band_hi, bhi.data(), bhi.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    ctx->tk_band_w = band_w;
    mark("tile intervals + first-launch list");
    if (hsd_hook(H_TOPK_NOMEM)) return hsd_fail(ctx, HSD_ERR_HIP, "hipMalloc: out of memory (HSD_TOPK_NOMEM rehearsal)");
    if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_H, (size_t)npos))) return rc;
    if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_tmax, (size_t)nt))) return rc;
    if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_count, (size_t)4))) return rc;
    ctx->tk_list_cap = (int)std::min<int64_t>(bm == 0 ? (int64_t)nt * (nt + 1) / 2 : (bpad / 128) * (int64_t)nt_graph, 0x7fffffff);
    { const int rc_l = hsd_pool_alloc(ctx, &ctx->d_tk_list, (size_t)ctx->tk_list_cap * sizeof(int2)); if (rc_l) return rc_l; }
    { const int rc_b = hsd_pool_alloc(ctx, &ctx->d_tk_band, band.size() * sizeof(int2)); if (rc_b) return rc_b; }
    ctx->tk_band_n = (int)band.size();
    HSD_HIP(ctx, hipMemcpy(ctx->d_tk_band, band.data(), band.size() * sizeof(int2), hipMemcpyHostToDevice));
    const dim3 pgrid((unsigned)((npos + 255) / 256));
    if (ctx->has_point)
        hipLaunchKernelGGL(tk_gather_kernel, dim3(pgrid.x, 1u), dim3(256), 0, ctx->stream, (const double *)ctx->d_H, npad, ctx->d_tk_orig, npos, ctx->d_tk_H);
    if (!seg) {
        const int cols = std::max(ctx->dcols, 16);
        if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_X, (size_t)cols * npos))) return rc;
        hipLaunchKernelGGL(tk_gather_kernel, dim3(pgrid.x, (unsigned)cols), dim3(256), 0, ctx->stream, (const double *)ctx->d_X, npad, ctx->d_tk_orig, npos, ctx->d_tk_X);
        if (bm > 0) {
            // (a row block's rectangle names its tiles by the positions' own granule list -- the graph's list ends with the graph)
            const int ngran16 = (int)(npos / 16);
            if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_gl, (size_t)ngran16))) return rc;
            std::vector<int32_t> gl((size_t)ngran16);
            for (int g = 0; g < ngran16; ++g) gl[(size_t)g] = g;
            HSD_HIP(ctx, hipMemcpy(ctx->d_tk_gl, gl.data(), gl.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    } else {
        // the zero-pad kinds' operands: packed words, 1 / ring size per layer, suffix sums -- and, recovered from them, what the tile planner reads:
        // members above each layer's threshold per position -> maxima per 16-position granule (+ the "too wide for the uint8 range" flags) -> per tile
        const int cols = std::max(ctx->dcols, 32), ngran16 = (int)(npos / 16);
        const int64_t sufr = (int64_t)ctx->suf_rows + 1;
        if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_U, (size_t)cols * npos))) return rc;
        if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_inv, (size_t)nlay * npos))) return rc;
        if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_suf, (size_t)sufr * npos))) return rc;
        if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_rmax16, (size_t)nlay * ngran16))) return rc;
        if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_gl, (size_t)ngran16))) return rc;
        if ((rc = hsd_op_alloc(ctx, &ctx->d_tk_tmaxs, (size_t)nlay * nt))) return rc;
        hipLaunchKernelGGL(tk_gather_u32_kernel, dim3(pgrid.x, (unsigned)cols), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_X, npad, ctx->d_tk_orig, npos, ctx->d_tk_U);
        hipLaunchKernelGGL(tk_gather_kernel, dim3(pgrid.x, (unsigned)nlay), dim3(256), 0, ctx->stream, (const double *)ctx->d_inv, npad, ctx->d_tk_orig, npos, ctx->d_tk_inv);
        if (ctx->d_suf) hipLaunchKernelGGL(tk_gather_u32_kernel, dim3(pgrid.x, (unsigned)sufr), dim3(256), 0, ctx->stream, ctx->d_suf, npad, ctx->d_tk_orig, npos, ctx->d_tk_suf);
        else HSD_HIP(ctx, hipMemsetAsync(ctx->d_tk_suf, 0, (size_t)sufr * npos * sizeof(uint32_t), ctx->stream));
        DevTmp dR, dneed;
        HSD_TMP_ALLOC(ctx, dR, (size_t)nlay * npos * sizeof(uint32_t));
        HSD_TMP_ALLOC(ctx, dneed, (size_t)2 * npos * sizeof(uint32_t));
        if (pack) hipLaunchKernelGGL(tk_rnode_kernel<true>, dim3(pgrid.x, (unsigned)nlay), dim3(256), 0, ctx->stream, ctx->d_tk_U, ctx->d_tk_suf, npos, ctx->d_seg_tab, dR.as<uint32_t>());
        else hipLaunchKernelGGL(tk_rnode_kernel<false>, dim3(pgrid.x, (unsigned)nlay), dim3(256), 0, ctx->stream, ctx->d_tk_U, ctx->d_tk_suf, npos, ctx->d_seg_tab, dR.as<uint32_t>());
        hipLaunchKernelGGL(granule16_max_kernel, dim3((unsigned)((ngran16 + 255) / 256), (unsigned)nlay), dim3(256), 0, ctx->stream, dR.as<uint32_t>(), npos, ngran16, ctx->d_tk_rmax16);
        for (int s2 = 0; s2 < nlay && pack; ++s2) {
            const HsdSegDesc &d = ctx->seg_tab[s2];
            if (d.r8_col0 < 0) continue;
            HSD_HIP(ctx, hipMemsetAsync(dneed.p, 0, (size_t)2 * npos * sizeof(uint32_t), ctx->stream));
            hipLaunchKernelGGL(u8_need_kernel, dim3(pgrid.x), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_tk_U, npos, npos, d, dneed.as<uint32_t>());
            hipLaunchKernelGGL(tk_wide_kernel, dim3((unsigned)((ngran16 + 255) / 256)), dim3(256), 0, ctx->stream, dneed.as<uint32_t>(), npos, ngran16, d.r8_k16 - d.r8_r16, d.r8_r16,
                               ctx->d_tk_rmax16 + (size_t)s2 * ngran16);
        }
        std::vector<int32_t> gl((size_t)ngran16);
        for (int g = 0; g < ngran16; ++g) gl[(size_t)g] = g;
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_tk_gl, gl.data(), gl.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(list_tile_max_kernel, dim3((unsigned)(((int64_t)nt * nlay + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_tk_gl, nt, ctx->d_tk_rmax16, ngran16, nlay,
                           ctx->d_tk_tmaxs);
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));          // (gl, the temporaries)
    }
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    mark("allocations + permuted operands");
    ctx->tk_nf = nf; ctx->tk_nt = nt;
    ctx->tk_blk0 = b0; ctx->tk_blk1 = b1; ctx->tk_npos = npos;
    ctx->tk_ready = true;
    return HSD_OK;
}

// *done = false: not applicable (or a list overflowed): the caller runs the unpruned sweep.
// [r0, r1): the rows asked for.  The whole graph is one symmetric, mirrored sweep over the graph's positions.  A ROW BLOCK (a rank's share of
// `--gpus P`) is a rectangle: its rows -- a second run of positions behind the graph's, in the graph's order -- against
// ALL columns, row lists only; a pair inside the block is evaluated from both of its rows, as the matrix's row blocks evaluate it, and every
// rank's work is rows x n pairs under the same pruning (the thresholds, the lists and so the results are those of the unpruned call).
int topk_pruned(hsd_ctx *ctx, int k, int64_t r0, int64_t r1, int32_t *idx_out, double *dist_out, bool *done) {
    *done = false;
    const int64_t n = ctx->n, npad = ctx->npad;
    const bool block = !(r0 == 0 && r1 == n);
    const int64_t bm = block ? r1 - r0 : 0, bpad = (bm + 127) / 128 * 128, npos = npad + bpad;
    const bool seg = ctx->mode == PAIR_SAD_U16 || ctx->mode == PAIR_SAD_U32;
    PairArgs a{};
    a.npad = npos;
    a.dcols = ctx->dcols;
    a.n_seg = seg ? ctx->n_seg : 0;
    a.list_cap = ctx->list_cap;
    a.dcols_valid = ctx->dcols_alg;
    { const char *e = hsd_hook(H_TOPK_DEBUG); a.dbg = e ? atoi(e) : 0; }
    a.rbase = 0;
    if (!block) { a.i_begin = 0; a.i_end = npad; a.j_begin = 0; a.j_end = npad; a.sym = 1; a.mirror = 1; }
    else { a.i_begin = npad; a.i_end = npos; a.j_begin = 0; a.j_end = npad; a.sym = 0; a.mirror = 0; a.self_by_orig = 1; }
    {
        // the persistent kernel must take the region (its LDS budget: at most three per-node vectors)
        if (a.n_seg + (ctx->has_point ? 1 : 0) > 3) return HSD_OK;
        PairArgs probe = a;
        probe.tau = reinterpret_cast<const double *>(ctx);          // (any non-null value: the k-nearest kernels have no output alignment to check)
        if (!stream_waves(ctx, probe)) return HSD_OK;
    }
    int rc = tk_prepare(ctx, block ? r0 : 0, block ? r1 : 0);
    if (rc) return rc;
    if (seg) {
        a.X = a.Y = ctx->d_tk_U;
        a.inv = ctx->d_tk_inv;
        a.seg_tab = reinterpret_cast<const SegDesc *>(ctx->d_seg_tab);
        a.rmax = ctx->d_rmax;
        a.ngran = ctx->ngran;
        a.suf = ctx->d_tk_suf;
        a.gl_ov = ctx->d_tk_gl; a.tmax_ov = ctx->d_tk_tmaxs; a.ntl_ov = ctx->tk_nt;
    } else {
        a.X = a.Y = ctx->d_tk_X;
        // (the fp64 kind has no per-tile maxima, but a row block's rectangle still takes its tile composition from the positions' own list)
        if (block) { a.gl_ov = ctx->d_tk_gl; a.tmax_ov = nullptr; a.ntl_ov = ctx->tk_nt; }
    }
    a.point = ctx->has_point ? ctx->d_tk_H : nullptr;
    a.orig_of = ctx->d_tk_orig;
    // rows of the sweep: positions [p0, p0 + rows); the per-row arrays below hold `rows` entries and are addressed by position (pointers moved
    // back by p0, as the unpruned sweep of a row block does with its node range)
    const int64_t p0 = block ? npad : 0, rows = block ? bpad : npad, out_rows = block ? bm : n;
    // (a row block: rows whose own neighbourhood fell outside the capped first launch enter the first phase under their SAMPLED threshold -- the 2 %
    //  quantile of 1 024 columns -- and meet n / phases columns there: the unpruned sweep's capacity rule, 4 130 entries at N = 200k; seen with
    //  2 048: one row of rank 7 of 8 appended 2 091 in the first phase, lost its list and the call was repeated unpruned)
    const int64_t phases_dflt = npad / 128 <= 1200 ? 5 : 8;
    // (the whole graph takes the same rule since round 5: at BA N = 200k its longest list after the first phase was 1 886 of 2 048 entries -- too close)
    const int64_t cap = std::max<int64_t>(std::max<int64_t>(2048, 512 + 8 * (int64_t)k), 8 * (int64_t)k * n / (1024 * phases_dflt) + 8 * k + 64);
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t oTau = carve((size_t)rows * sizeof(double)), oCnt = carve((size_t)rows * sizeof(int)), oIdx = carve((size_t)rows * cap * sizeof(int));
    const size_t oVal = carve((size_t)rows * cap * sizeof(double)), oOi = carve((size_t)rows * k * sizeof(int)), oOv = carve((size_t)rows * k * sizeof(double));
    const size_t oT16 = carve((size_t)rows * sizeof(uint16_t) + 64), oFi = carve((size_t)out_rows * k * sizeof(int)), oFv = carve((size_t)out_rows * k * sizeof(double));
    const int64_t M = std::min<int64_t>(n, 1024), ldS = (M + 127) / 128 * 128;
    const size_t oS = carve((size_t)out_rows * ldS * sizeof(double)), oTn = carve((size_t)out_rows * sizeof(double));
    if (ctx->scratch_bytes < off) {
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        HSD_HIP(ctx, hipMalloc(&ctx->d_scratch, off));
        ctx->scratch_bytes = off;
    }
    char *base = static_cast<char *>(ctx->d_scratch);
    double *d_tau = reinterpret_cast<double *>(base + oTau), *d_val = reinterpret_cast<double *>(base + oVal), *d_oval = reinterpret_cast<double *>(base + oOv);
    double *d_fval = reinterpret_cast<double *>(base + oFv);
    int *d_cnt = reinterpret_cast<int *>(base + oCnt), *d_idx = reinterpret_cast<int *>(base + oIdx), *d_oidx = reinterpret_cast<int *>(base + oOi);
    int *d_fidx = reinterpret_cast<int *>(base + oFi);
    HSD_HIP(ctx, hipMemsetAsync(d_cnt, 0, (size_t)rows * sizeof(int), ctx->stream));
    HSD_HIP(ctx, hipMemsetAsync(ctx->d_flag, 0, 4 * sizeof(int), ctx->stream));
    // First thresholds, part one: the k-th smallest of 1 024 columns spread evenly over the NATURAL node order (the unpruned sweep's own first
    // thresholds, computed on the untouched operands).  A row unlike everything near it in the k-d order -- a hub, whose distances to the
    // thousands of leaves crowd just above its k-th -- would otherwise start from a threshold that admits most of the graph (zero-pad distance,
    // BA N = 100k: lists of 38 000 entries).  Part two, the tiles next to a row's own, only tightens it.  The raw fp64 distance starts from
    // +inf: its lists stay short without the sample (longest 709 of 2 048 at BA N = 100k, 27 000 soak calls), which would cost it 4.4 ms
    // of arithmetic + 0.8 GB of sample matrix per call (66.7 -> 74.4 ms, measured).
    // A ROW BLOCK samples for both kinds: 128 consecutive block rows span as many of the graph's tiles as the block is thin there (the oldest
    // rows of a scale-free graph are its hubs: dense in a few tiles, one in fifty elsewhere), and under thresholds of +inf a wide first launch
    // overflows the lists (seen: a row tile across 17 tiles = 2 176 candidates per row against a capacity of 2 048; rows whose lists overflowed
    // keep +inf and collect the whole graph).  With sampled thresholds the first launch is filtered like the later ones.
    const bool sample = seg || block;
    if (!sample) hipLaunchKernelGGL(tk_tau_open_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, (const int32_t *)ctx->d_tk_orig + p0, rows, d_tau);
    else {
        // (rows [q0, q1) of the natural order: the whole graph, or the block; S and tauN hold these rows only)
        const int64_t q0 = block ? r0 : 0, q1 = block ? r1 : n;
        double *d_S = reinterpret_cast<double *>(base + oS), *d_tauN = reinterpret_cast<double *>(base + oTn);
        PairArgs nat{};
        nat.X = ctx->d_X; nat.Y = ctx->d_X;
        nat.inv = ctx->d_inv;
        nat.point = ctx->has_point ? ctx->d_H : nullptr;
        nat.npad = npad; nat.dcols = ctx->dcols; nat.n_seg = ctx->n_seg;
        nat.seg_tab = reinterpret_cast<const SegDesc *>(ctx->d_seg_tab);
        nat.rmax = ctx->d_rmax; nat.ngran = ctx->ngran; nat.list_cap = ctx->list_cap;
        nat.dcols_valid = ctx->dcols_alg;
        nat.rbase = q0; nat.i_begin = q0; nat.i_end = q1;
        const int n_sub = (int)((M + 127) / 128);
        const int64_t n_tiles_total = (n + 127) / 128;
        for (int t = 0; t < n_sub; ++t) {
            const int64_t tile = n_sub >= n_tiles_total ? t : (int64_t)t * n_tiles_total / n_sub;
            PairArgs sub = nat;
            sub.j_begin = tile * 128;
            sub.j_end = std::min<int64_t>(n, sub.j_begin + 128);
            sub.out = d_S + (int64_t)t * 128 - sub.j_begin;
            sub.ld = ldS;
            if ((rc = launch_region(ctx, sub, tile_edge(ctx), ctx->stream))) return rc;
            if (sub.j_end - sub.j_begin < 128)
                hipLaunchKernelGGL(fill_columns_kernel, dim3((unsigned)((q1 - q0 + 255) / 256)), dim3(256), 0, ctx->stream, d_S, ldS, q1 - q0, (int)(t * 128 + (sub.j_end - sub.j_begin)),
                                   (int)((t + 1) * 128), (double)INFINITY);
        }
        hipLaunchKernelGGL(topk_threshold_kernel, dim3((unsigned)((q1 - q0 + 3) / 4)), dim3(256), 0, ctx->stream, d_S, ldS, n_sub * 128, (int64_t)0, q1 - q0, k, 0.0, d_tauN);
        hipLaunchKernelGGL(tk_permute_tau_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_tk_orig + p0, rows, d_tauN - q0, d_tau);
    }
    PairArgs b = a;
    b.tau = d_tau - p0; b.cand_cnt = d_cnt - p0; b.cand_idx = d_idx - p0 * cap; b.cand_val = d_val - p0 * cap; b.cand_cap = (int)cap;
    b.tau16 = reinterpret_cast<const uint16_t *>(base + oT16) - p0;
    const int BT = tile_edge(ctx);
    const size_t cmp_lds = (size_t)4 * k * 12;
    // launch 0: every row against the tiles next to its own in the order, under thresholds of +inf (a row block: under the sampled thresholds,
    // and filtered by them); then the first real thresholds
    b.explicit_tiles = reinterpret_cast<const int2 *>(ctx->d_tk_band); b.n_explicit = ctx->tk_band_n;
    if (block) { b.prune_lo = ctx->d_tk_lo; b.prune_hi = ctx->d_tk_hi; b.prune_scale = seg ? ctx->d_tk_scale : nullptr; b.prune_nf = ctx->tk_nf; b.prune_nt = ctx->tk_nt; b.prune_drop_band = 0; }
    if ((rc = launch_region(ctx, b, BT, ctx->stream))) return rc;
    const bool verbose = hsd_hook(H_TOPK_VERBOSE) != nullptr;
    const auto list_stats = [&](const char *when) {
        // HSD_TOPK_VERBOSE: the lists after a launch -- longest, shortest, rows still below k entries (their thresholds cannot tighten)
        if (!verbose) return;
        std::vector<int> hc((size_t)out_rows);
        std::vector<double> ht((size_t)out_rows);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(hc.data(), d_cnt, (size_t)out_rows * sizeof(int), hipMemcpyDeviceToHost);
        (void)hipMemcpy(ht.data(), d_tau, (size_t)out_rows * sizeof(double), hipMemcpyDeviceToHost);
        int64_t mx = 0, at = 0, below = 0;
        for (int64_t q = 0; q < out_rows; ++q) { if (hc[(size_t)q] > mx) { mx = hc[(size_t)q]; at = q; } below += hc[(size_t)q] < k; }
        fprintf(stderr, "[hsd topk] %s: longest list %lld (row position %lld, threshold %.6g), %lld rows below k entries\n", when, (long long)mx, (long long)at, ht[(size_t)at], (long long)below);
    };
    list_stats("first launch");
    hipLaunchKernelGGL(topk_compact_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), cmp_lds, ctx->stream, d_cnt, d_idx, d_val, (int)cap, rows, k, 0.0, d_tau);
    // the rest of the pair space in anti-diagonal phases, each filtered against the thresholds of the moment (tightened after phases 1, 2, 4)
    b.explicit_tiles = nullptr; b.n_explicit = 0;
    b.prune_lo = ctx->d_tk_lo; b.prune_hi = ctx->d_tk_hi; b.prune_scale = seg ? ctx->d_tk_scale : nullptr; b.prune_nf = ctx->tk_nf; b.prune_nt = ctx->tk_nt; b.prune_drop_band = ctx->tk_band_w;
    if (block) { b.prune_band_lo = ctx->d_tk_band_lo; b.prune_band_hi = ctx->d_tk_band_hi; }
    // (measured at BA N = 100k: 2 phases overflow the lists -- the band thresholds meet half the columns at once --, 3 / 4 / 5 / 6 / 8 / 12 / 16 take
    //  66.3 / 65.7 / 65.6 / 66.8 / 68.8 / 72.1 / 74.5 ms with longest lists of 847 / 690 / 296 / 360 / 709 entries; twice the nodes, twice the appends per phase)
    int n_phases = (int)phases_dflt;
    if (const char *ep = hsd_hook(H_TOPK_PHASES)) n_phases = std::max(1, std::min(16, atoi(ep)));
    for (int ph = 0; ph < n_phases; ++ph) {
        b.phase = ph; b.phase_w = 1; b.n_phases = n_phases;
        if ((rc = launch_region(ctx, b, BT, ctx->stream))) return rc;
        list_stats("phase");
        if (ph + 1 < n_phases && ((ph + 1) & ph) == 0)
            hipLaunchKernelGGL(topk_compact_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), cmp_lds, ctx->stream, d_cnt, d_idx, d_val, (int)cap, rows, k, 0.0, d_tau);
    }
    // select (whole graph: the rows of real nodes are the first n positions; a block: its first bm), back to node order, out
    int flag[4] = {0, 0, 0, 0};
    hipLaunchKernelGGL(topk_select_kernel, dim3((unsigned)((out_rows + 3) / 4)), dim3(256), 0, ctx->stream, d_cnt, d_idx, d_val, (int)cap, (int64_t)0, out_rows, k, d_oidx, d_oval,
                       ctx->d_flag);
    // (row q of the result is node orig[p0 + q]; a block's nodes start at r0)
    hipLaunchKernelGGL(tk_unpermute_kernel, dim3((unsigned)((out_rows * k + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_tk_orig + p0, out_rows, k, d_oidx, d_oval,
                       d_fidx - (block ? r0 : 0) * k, d_fval - (block ? r0 : 0) * k);
    HSD_HIP(ctx, hipMemcpyAsync(flag, ctx->d_flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    HSD_HIP(ctx, hipGetLastError());
    if (hsd_hook(H_TOPK_VERBOSE)) {
        std::vector<int> hc((size_t)out_rows);
        (void)hipMemcpy(hc.data(), d_cnt, (size_t)out_rows * sizeof(int), hipMemcpyDeviceToHost);
        int64_t mx = 0;
        for (int v : hc) mx = std::max<int64_t>(mx, v);
        fprintf(stderr, "[hsd topk] pruned sweep%s: capacity %lld, longest list %lld, overflow %d\n", block ? " (row block)" : "", (long long)cap, (long long)mx, flag[0] & 1);
    }
    if (flag[0] & 1) return HSD_OK;                 // a list overflowed (or came up short): the unpruned sweep takes the call
    HSD_HIP(ctx, hipMemcpyAsync(idx_out, d_fidx, (size_t)out_rows * k * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HSD_HIP(ctx, hipMemcpyAsync(dist_out, d_fval, (size_t)out_rows * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *done = true;
    return HSD_OK;
}

}  // namespace

extern "C" int hsd_pair_topk(hsd_ctx *ctx, int k, int64_t r0, int64_t r1, int32_t *idx_out, double *dist_out) try {
    HSD_CHECK_CTX(ctx);
    if (ctx->mode == PAIR_NONE) return hsd_fail(ctx, HSD_ERR_NO_SIGNATURE, "hsd_pair_prepare has not been called");
    if (ctx->mode == PAIR_KL)
        return hsd_fail(ctx, HSD_ERR_UNSUPPORTED, "top-k is offered for the symmetric distances (wasserstein, symmetric_kl)");
    const int64_t n = ctx->n;
    if (r0 < 0 || r1 > n || r0 > r1) return hsd_fail(ctx, HSD_ERR_INVALID, "row range out of bounds");
    if (k < 1 || k > n - 1 || k > 1024) return hsd_fail(ctx, HSD_ERR_INVALID, "k=%d must be in [1, min(n-1, 1024)]", k);
    if (r0 == r1) return HSD_OK;
    if (!idx_out || !dist_out) return hsd_fail(ctx, HSD_ERR_INVALID, "null output");
    const int64_t rows = r1 - r0;
    const bool wass = ctx->mode == PAIR_L1_F64 || ctx->mode == PAIR_SAD_U32 || ctx->mode == PAIR_SAD_U16 || ctx->mode == PAIR_ZP_F64;
    {
        // raw fp64 W1 over the whole graph: the pruned sweep on node-permuted operands (above) where it applies
        const int pr = hsd_hook_int(H_TOPK_PRUNE, -1);
        // (the zero-pad kinds from ~50k nodes (384 tiles) on, or on request: where a FIRST call -- one-time setup included -- is ahead of the unpruned
        //  call.  BA N = 40k: 5.05 ms unpruned against 5.2-5.3 for a first call, 4.1 for every later one; N = 60k: 9.1-9.6 against 7.5-7.8 / 6.1;
        //  N = 80k: 13.8-14.6 against 10.8-11.5 / 8.8; N = 100k: 20.0-20.9 against 15.4 / 12.0; N = 200k: 73 against 33.4 for a later call.  With the node order from host threads (HSD_TOPK_ORDER=0) the first calls were
        //  12.4-13.0 and 20.9-21.5 ms and the default began at 100k nodes.)
        const bool l1 = ctx->mode == PAIR_L1_F64 && ctx->l1_cols_per_layer > 0 && ctx->dcols_alg > 0 && pr != 0;
        const bool zp = (ctx->mode == PAIR_SAD_U16 || ctx->mode == PAIR_SAD_U32) && ctx->n_seg >= 1 && (pr == 1 || (pr != 0 && ctx->npad / 128 >= 384));
        // (a row block -- a rank's share -- takes the pruned sweep too, whatever its bounds: its rows become positions of their own)
        const bool whole = r0 == 0 && r1 == n, blk = !whole;
        if ((l1 || zp) && (whole || blk) && k <= 128 && !hsd_hook(H_TOPK_GEOM)) {
            bool done = false;
            const int rcp = topk_pruned(ctx, k, r0, r1, idx_out, dist_out, &done);
            if (rcp == HSD_ERR_HIP || rcp == HSD_ERR_NOMEM || rcp == HSD_ERR_INTERNAL) {
                // the pruned sweep needs a permuted copy of the operands, a sample matrix and rocprim's temporaries on top of what the unpruned
                // sweep needs: when one of them cannot be had, the unpruned sweep -- which worked before the pruned one existed -- takes the call
                hsd_tk_release(ctx);
                (void)hipGetLastError();
                if (hsd_hook(H_TOPK_VERBOSE)) fprintf(stderr, "[hsd topk] pruned sweep failed (%s): the unpruned sweep takes the call\n", hsd_last_error(ctx));
            } else if (rcp != HSD_OK || done) return rcp;
        }
    }

    PairArgs a{};
    a.X = ctx->d_X;
    a.Y = wass ? ctx->d_X : ctx->d_Y;
    a.H = wass ? nullptr : ctx->d_H;
    a.inv = ctx->d_inv;
    a.point = ctx->has_point ? (wass ? ctx->d_H : ctx->d_H + ctx->npad) : nullptr;
    a.point_c = ctx->kl_point_c;
    a.kmask = wass ? nullptr : ctx->d_kl_mask; a.kwords = ctx->kl_words; a.khalf = ctx->kl_half;
    a.npad = ctx->npad;
    a.dcols = ctx->dcols;
    a.n_seg = ctx->n_seg;
    a.seg_tab = reinterpret_cast<const SegDesc *>(ctx->d_seg_tab);
    a.rmax = ctx->d_rmax;
    a.ngran = ctx->ngran;
    a.list_cap = ctx->list_cap;
    a.work = nullptr;
    a.hmode = ctx->mode == PAIR_SKL ? 1 : 0;
    { const char *e = hsd_hook(H_TOPK_DEBUG); a.dbg = e ? atoi(e) : 0; }     // measurement: 1 = sweep without its candidate epilogue
    a.dcols_valid = (ctx->mode == PAIR_SKL) ? ctx->dcols : ctx->dcols_alg;
    a.rbase = r0;
    a.i_begin = r0; a.i_end = r1;

    const double slack = ctx->mode == PAIR_SKL ? 1e-11 : 0.0;      // W1 is evaluated identically in both orientations of a pair
    int64_t M = std::min<int64_t>(n, std::max<int64_t>(1024, 4 * (int64_t)k));
    // A large sweep runs in phases (anti-diagonal samples of the tile groups, tile_order): after each one every row keeps
    // its k best and tightens its threshold, so that the candidates appended per row fall from ~ k n / M to
    // ~ k n / (M P) + k (1 + 1/2 + .. + 1/(P-1)).  HSD_TOPK_PHASES forces a count (tests).
    int n_phases = 1;
    // HSD_TOPK_GEOM=1 (opt-in, measured): the launches GEOMETRICALLY instead of in equal phases: 1/32, 1/32, 1/16, 1/8, 1/4, 1/2 of the tile groups
    // (residues [0,1), [1,2), [2,4), [4,8), [8,16), [16,32) of the anti-diagonal index mod 32).  What phases save are appends
    // (~0.09 ms per million at BA N = 100k), and appends are (columns of the launch) x k / (columns the thresholds have seen): the
    // loose first thresholds -- from a 1 % sample -- then meet 3 % of the columns instead of 12.5 % (61 instead of 244 appends per
    // row), and every later launch doubles what the thresholds know.  Lists are compacted (0.42 ms) where that saves more in the
    // next launch than it costs -- after the first and the fourth -- and after the fifth: rows whose k-th distance is shared by thousands of
    // columns (leaves of a scale-free graph with identical rings: the integer distances tie exactly) append every one of them, and the
    // last launch alone covers half the columns (without it 4 of 100 000 rows overflowed 2 048 entries and the call was repeated).
    // Zero-pad distance: 20.1 -> 19.4-19.9 ms at BA N = 100k, 71.7 -> 66.3 ms at N = 200k.  NOT the default: rows whose distances to
    // most columns lie in a narrow band just above their k-th (hubs against the leaves) keep a threshold inside that band after a
    // first launch that shows them only 3 % of the columns, and append thousands in the launches that follow -- raw fp64 W1 at BA
    // N = 100k: 1 956 rows overflow 2 048 entries (longest list 7 663) and the call is repeated at twice the price (equal phases:
    // longest list 441).
    bool geometric = false;
    {
        const int64_t t128 = (rows + 127) / 128, tiles = t128 * ((n + 127) / 128);
        const int cus = std::max(1, ctx->prop.multiProcessorCount);
        while (n_phases < 8 && tiles / (2 * n_phases) >= 64 * (int64_t)cus) n_phases *= 2;
        if (const char *ep = hsd_hook(H_TOPK_PHASES)) n_phases = std::max(1, std::min(16, atoi(ep)));
        else if (const char *eg = hsd_hook(H_TOPK_GEOM)) geometric = atoi(eg) != 0;
    }
    struct Launch { int lo, w, mod; bool compact_after; };
    std::vector<Launch> sched;
    if (geometric) {
        sched = {{0, 1, 32, true}, {1, 1, 32, false}, {2, 2, 32, false}, {4, 4, 32, true}, {8, 8, 32, true}, {16, 16, 32, false}};
        n_phases = 32;                                // (the list capacity below is sized by the first launch's share)
    } else if (n_phases == 8 && !hsd_hook(H_TOPK_PHASES)) {
        // eight equal phases with the FIRST one in two halves and a compaction in between: the sampled thresholds are the loosest
        // the sweep ever uses (a 2 % quantile), and half a phase under them appends 122 instead of 244 entries per row before the
        // first tightening (-10^7 appends at BA N = 100k against 0.42 ms for the extra compaction: 19.5 -> 19.25 ms; N = 200k: 72 -> 66 ms).
        // Longest list 776 (zero-pad) / 940 (raw fp64) of 2 177 entries, plain equal phases 416 / 441: rows whose distances crowd just
        // above their k-th tighten less after half a phase.  Compaction otherwise after phases 1, 2, 4 as below.
        sched = {{0, 1, 16, true}, {1, 1, 16, true}, {2, 2, 16, true}, {4, 2, 16, false}, {6, 2, 16, true},
                 {8, 2, 16, false}, {10, 2, 16, false}, {12, 2, 16, false}, {14, 2, 16, false}};
        // (n_phases stays 8: the list capacity below is the one equal phases were validated with)
    } else {
        // equal phases; the thresholds tighten fastest early on: compact after phases 1, 2, 4, ... (phases 5-8 of 8 together append ~k entries)
        for (int ph = 0; ph < n_phases; ++ph) sched.push_back({ph, 1, n_phases, ph + 1 < n_phases && ((ph + 1) & ph) == 0});
    }
    // list capacity: the expected candidates of the first phase (k / M of a row's n / P columns) with room for rows of many
    // equal distances (the integer zero-pad distances tie by the hundred); an overflow anywhere costs a whole retry
    int64_t cap = std::min<int64_t>(n, n_phases > 1 ? std::max<int64_t>(2048, 8 * (int64_t)k * n / (M * n_phases) + 8 * k + 64)
                                                    : std::max<int64_t>(256, 4 * (int64_t)k * n / M + 64));
    if (const char *ec = hsd_hook(H_TOPK_CAP)) cap = std::min<int64_t>(n, std::max<int64_t>(k + 1, atoll(ec)));   // test hook
    int rc = HSD_OK;
    for (int attempt = 0; attempt < 3 && rc == HSD_OK; ++attempt) {
        if (attempt == 1) { M = std::min<int64_t>(n, 4 * M); cap = std::min<int64_t>(n, 8 * cap); }
        if (attempt == 2) { M = std::min<int64_t>(n, 16384); cap = n; }
        M = std::min<int64_t>(M, 16384);
        const int64_t ldS = (M + 127) / 128 * 128;
        // scratch: one grow-only arena per context (allocating gigabytes per call would dominate)
        size_t off = 0;
        auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
        const size_t oS = carve((size_t)rows * ldS * sizeof(double)), oTau = carve((size_t)rows * sizeof(double));
        const size_t oCnt = carve((size_t)rows * sizeof(int)), oIdx = carve((size_t)rows * cap * sizeof(int));
        const size_t oVal = carve((size_t)rows * cap * sizeof(double)), oOi = carve((size_t)rows * k * sizeof(int));
        const size_t oOv = carve((size_t)rows * k * sizeof(double));
        const size_t oT16 = carve((size_t)rows * sizeof(uint16_t) + 64);
        if (ctx->scratch_bytes < off) {
            HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
            ctx->d_scratch = nullptr;
            ctx->scratch_bytes = 0;
            HSD_HIP(ctx, hipMalloc(&ctx->d_scratch, off));
            ctx->scratch_bytes = off;
        }
        char *base = static_cast<char *>(ctx->d_scratch);
        double *d_S = reinterpret_cast<double *>(base + oS), *d_tau = reinterpret_cast<double *>(base + oTau);
        double *d_val = reinterpret_cast<double *>(base + oVal), *d_oval = reinterpret_cast<double *>(base + oOv);
        int *d_cnt = reinterpret_cast<int *>(base + oCnt), *d_idx = reinterpret_cast<int *>(base + oIdx);
        int *d_oidx = reinterpret_cast<int *>(base + oOi);
        hipError_t e = hipMemsetAsync(d_cnt, 0, (size_t)rows * sizeof(int), ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(ctx->d_flag, 0, 4 * sizeof(int), ctx->stream);
        // 1. threshold from M columns taken as 128-node tiles spread evenly over the node order (the
        //    first M nodes would be a biased sample: e.g. the oldest = highest-degree nodes of a
        //    preferential-attachment graph); column t*128 + u of S is node tile_start(t) + u
        const int n_sub = (int)((M + 127) / 128);
        const int64_t n_tiles_total = (n + 127) / 128;
        for (int t = 0; t < n_sub && rc == HSD_OK && e == hipSuccess; ++t) {
            const int64_t tile = n_sub >= n_tiles_total ? t : (int64_t)t * n_tiles_total / n_sub;
            PairArgs sub = a;
            sub.j_begin = tile * 128;
            sub.j_end = std::min<int64_t>(n, sub.j_begin + 128);
            sub.out = d_S + (int64_t)t * 128 - sub.j_begin;      // out[(i - r0) * ld + j] lands in column t*128 + (j - j_begin)
            sub.ld = ldS; sub.sym = 0; sub.mirror = 0; sub.swap = 0; sub.tau = nullptr;
            rc = launch_region(ctx, sub, tile_edge(ctx), ctx->stream);
            if (rc == HSD_OK && sub.j_end - sub.j_begin < 128)   // last, partial tile: pad with +inf
                hipLaunchKernelGGL(fill_columns_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, d_S, ldS, rows,
                                   (int)(t * 128 + (sub.j_end - sub.j_begin)), (int)((t + 1) * 128), (double)INFINITY);
        }
        if (rc == HSD_OK && e == hipSuccess) {
            const int Ms = n_sub * 128;
            hipLaunchKernelGGL(topk_threshold_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, d_S, ldS, Ms, r0, rows, k, slack, d_tau - r0);
        }
        // 2. sweep: square on the diagonal (both rows of a pair are ours) + the blocks left and right of it
        for (size_t li = 0; li < sched.size() && rc == HSD_OK && e == hipSuccess; ++li) {
            const Launch &L = sched[li];
            PairArgs b = a;
            b.out = nullptr; b.ld = 0;
            b.tau = d_tau - r0; b.cand_cnt = d_cnt - r0; b.cand_idx = d_idx - r0 * cap; b.cand_val = d_val - r0 * cap; b.cand_cap = (int)cap;
            b.tau16 = reinterpret_cast<const uint16_t *>(base + oT16) - r0;
            b.phase = L.lo; b.phase_w = L.w; b.n_phases = L.mod;
            b.j_begin = r0; b.j_end = r1; b.sym = 1; b.mirror = 1; b.swap = 0;
            const int BT = tile_edge(ctx);
            rc = launch_region(ctx, b, BT, ctx->stream);
            b.sym = 0; b.mirror = 0;
            if (rc == HSD_OK && r0 > 0) { b.j_begin = 0; b.j_end = r0; rc = launch_region(ctx, b, BT, ctx->stream); }
            if (rc == HSD_OK && r1 < n) { b.j_begin = r1; b.j_end = n; rc = launch_region(ctx, b, BT, ctx->stream); }
            if (rc == HSD_OK && L.compact_after)
                hipLaunchKernelGGL(topk_compact_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), (size_t)4 * k * 12, ctx->stream, d_cnt, d_idx, d_val, (int)cap, rows, k, slack, d_tau);
        }
        // 3. select
        int flag[4] = {0, 0, 0, 0};
        if (rc == HSD_OK && e == hipSuccess) {
            hipLaunchKernelGGL(topk_select_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, d_cnt, d_idx, d_val, (int)cap,
                               (int64_t)0, rows, k, d_oidx, d_oval, ctx->d_flag);
            e = hipMemcpyAsync(flag, ctx->d_flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e == hipSuccess) e = hipGetLastError();
        }
        if (hsd_hook(H_TOPK_VERBOSE) && rc == HSD_OK && e == hipSuccess) {
            std::vector<int> hc((size_t)rows);
            (void)hipMemcpy(hc.data(), d_cnt, (size_t)rows * sizeof(int), hipMemcpyDeviceToHost);
            int64_t mx = 0, over = 0;
            for (int v : hc) { mx = std::max<int64_t>(mx, v); over += v > cap; }
            fprintf(stderr, "[hsd topk] attempt %d: M %lld, phases %d, capacity %lld, longest list %lld, rows over capacity %lld\n", attempt,
                    (long long)M, n_phases, (long long)cap, (long long)mx, (long long)over);
        }
        const bool again = rc == HSD_OK && e == hipSuccess && (flag[0] & 1) && attempt < 2;
        if (rc == HSD_OK && e == hipSuccess && !(flag[0] & 1)) {
            e = hipMemcpyAsync(idx_out, d_oidx, (size_t)rows * k * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(dist_out, d_oval, (size_t)rows * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
        (void)hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "hsd_pair_topk: %s", hipGetErrorString(e));
        if (rc != HSD_OK) return rc;
        if (!again) {
            if (flag[0] & 1) return hsd_fail(ctx, HSD_ERR_TOO_LARGE, "hsd_pair_topk: candidate lists overflowed even at capacity n");
            return HSD_OK;
        }
    }
    return rc;
} HSD_ABI_GUARD(ctx)

