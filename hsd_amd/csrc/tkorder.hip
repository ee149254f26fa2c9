// Node order of the pruned k-nearest sweep (pair.hip: tk_prepare), built on the device.
//
// The order is a k-d order of the nodes on a handful of per-node features: split the widest feature at the
// median, rounded to whole 128-node leaves, down to leaves of at most 128 nodes.  The shape of that tree depends on n alone, so the
// host lists every level's ranges once; the device then does, per level,
//     extents   min / max of every feature over every range (wave reduction + 64-bit atomics on order-preserving bit patterns)
//     keys      (range index, value of the range's widest feature) per position; ranges that are leaves keep their order
//     sort      one rocprim radix sort of (key, node) over all positions: a range is a contiguous block of positions and the range
//               index is the high part of the key, so every node stays inside its range and the range comes out sorted on its feature
// The next level's ranges are the halves.  Ten levels at 100 000 nodes.  ANY permutation of the nodes gives the same k nearest
// (the sweep's bounds are computed from the tiles as they are; tests/test_gpu_topk.py compares with the unpruned sweep bit for bit):
// the value part of a key is the top 44 bits of the double's pattern, ties keep the order they had.
// Reference: none -- the reference forms the whole matrix (model/HSD.py:98-114) and hands it to a k-nearest evaluation (main.py:29).
#include "common.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <vector>

namespace {

constexpr int TKO_VALUE_BITS = 44;
constexpr int TKO_LEAF = 128;

__device__ __forceinline__ unsigned long long ord_bits(double v) {           // unsigned patterns that order like the doubles
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord_value(unsigned long long u) {
    u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
    return __longlong_as_double((long long)u);
}

// index of the range that holds position p: b[0] = 0 < b[1] < ... < b[m] = n
__device__ __forceinline__ int range_of(const int32_t *__restrict__ b, int m, int32_t p) {
    int lo = 0, hi = m;                                // invariant: b[lo] <= p < b[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (b[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = (unsigned long long)__double_as_longlong(__shfl_xor(__longlong_as_double((long long)v), d, 64));
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = (unsigned long long)__double_as_longlong(__shfl_xor(__longlong_as_double((long long)v), d, 64));
        v = o > v ? o : v;
    }
    return v;
}

__global__ void tko_iota_kernel(int32_t *__restrict__ order, int32_t n) {
    const int32_t p = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (p < n) order[p] = p;
}

// mn / mx [range][feature]: extents of the features over the ranges (mn starts at all ones, mx at zero: below / above every pattern)
__global__ void tko_extent_kernel(const double *__restrict__ F, int32_t n, int nf, const int32_t *__restrict__ order, const int32_t *__restrict__ b, int m,
                                  unsigned long long *__restrict__ mn, unsigned long long *__restrict__ mx) {
    const int32_t p = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    const bool valid = p < n;
    // one range for the whole wave (the rule near the root): one atomic pair per wave and feature instead of 64
    const int32_t w0 = (int32_t)(blockIdx.x * blockDim.x + (threadIdx.x & ~63u));
    const int32_t w1 = min(w0 + 63, n - 1);
    if (w0 >= n) return;                                // (whole wave: w0 is wave-uniform)
    const int r0 = range_of(b, m, w0), r1 = range_of(b, m, w1);
    const int r = valid ? (r0 == r1 ? r0 : range_of(b, m, p)) : r0;
    const int32_t node = valid ? order[p] : 0;
    for (int f = 0; f < nf; ++f) {
        const unsigned long long v = valid ? ord_bits(F[(int64_t)f * n + node]) : 0ull;
        if (r0 == r1) {
            const unsigned long long lo = wave_min_u64(valid ? v : ~0ull), hi = wave_max_u64(v);
            if ((threadIdx.x & 63u) == 0u) {
                atomicMin(&mn[(int64_t)r * nf + f], lo);
                atomicMax(&mx[(int64_t)r * nf + f], hi);
            }
        } else if (valid) {
            atomicMin(&mn[(int64_t)r * nf + f], v);
            atomicMax(&mx[(int64_t)r * nf + f], v);
        }
    }
}

__global__ void tko_key_kernel(const double *__restrict__ F, int32_t n, int nf, const int32_t *__restrict__ order, const int32_t *__restrict__ b, int m,
                               const unsigned long long *__restrict__ mn, const unsigned long long *__restrict__ mx, unsigned long long *__restrict__ keys,
                               int32_t *__restrict__ vals) {
    const int32_t p = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (p >= n) return;
    const int r = range_of(b, m, p);
    const int32_t lo = b[r], hi = b[r + 1];
    const int32_t node = order[p];
    unsigned long long low;
    if (hi - lo <= TKO_LEAF) low = (unsigned long long)(p - lo);          // a leaf: as it is
    else {
        int best = 0;
        double ext = -1.0;
        for (int f = 0; f < nf; ++f) {
            const double e = ord_value(mx[(int64_t)r * nf + f]) - ord_value(mn[(int64_t)r * nf + f]);
            if (e > ext) { ext = e; best = f; }
        }
        low = ord_bits(F[(int64_t)best * n + node]) >> (64 - TKO_VALUE_BITS);
    }
    keys[p] = ((unsigned long long)r << TKO_VALUE_BITS) | low;
    vals[p] = node;
}

struct Tmp {
    void *p = nullptr;
    ~Tmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

}  // namespace

// d_F[f][i] (nf rows of n doubles, device) -> d_order[p] = the node at position p (n entries, device).  Work is queued on ctx->stream and
// finished (synchronised) on return.
int hsd_tk_order_device(hsd_ctx *ctx, const double *d_F, int64_t n64, int nf, int32_t *d_order) {
    if (n64 < 1 || n64 > 0x7fffff00ll || nf < 1) return hsd_fail(ctx, HSD_ERR_INVALID, "node order: n=%lld, %d features", (long long)n64, nf);
    const int32_t n = (int32_t)n64;
    // every level's ranges (the tree's shape depends on n alone); a level is listed while one of its ranges is longer than a leaf
    std::vector<int32_t> all;                  // the levels' boundary lists, one after the other
    std::vector<size_t> first;                 // where a level's list starts in `all`
    std::vector<int> count;                    // its number of ranges
    {
        std::vector<int32_t> b{0, n};
        for (;;) {
            bool any = false;
            std::vector<int32_t> nb{0};
            for (size_t s = 0; s + 1 < b.size(); ++s) {
                const int32_t lo = b[s], hi = b[s + 1];
                if (hi - lo > TKO_LEAF) {
                    any = true;
                    int32_t mid = lo + ((hi - lo) / 2 + TKO_LEAF - 1) / TKO_LEAF * TKO_LEAF;
                    if (mid >= hi) mid = lo + (hi - lo) / 2;
                    nb.push_back(mid);
                }
                nb.push_back(hi);
            }
            if (!any) break;
            first.push_back(all.size());
            count.push_back((int)b.size() - 1);
            all.insert(all.end(), b.begin(), b.end());
            b.swap(nb);
        }
    }
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(tko_iota_kernel, grid, block, 0, ctx->stream, d_order, n);
    if (count.empty()) {                       // one leaf
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return HSD_OK;
    }
    const int m_max = *std::max_element(count.begin(), count.end());
    if ((unsigned long long)m_max >= (1ull << (64 - TKO_VALUE_BITS))) return hsd_fail(ctx, HSD_ERR_INVALID, "node order: too many ranges");
    Tmp dB, dMn, dMx, dK0, dK1, dV1, dSort;
    HSD_HIP(ctx, dB.alloc(all.size() * sizeof(int32_t)));
    HSD_HIP(ctx, dMn.alloc((size_t)m_max * nf * sizeof(unsigned long long)));
    HSD_HIP(ctx, dMx.alloc((size_t)m_max * nf * sizeof(unsigned long long)));
    HSD_HIP(ctx, dK0.alloc((size_t)n * sizeof(unsigned long long)));
    HSD_HIP(ctx, dK1.alloc((size_t)n * sizeof(unsigned long long)));
    HSD_HIP(ctx, dV1.alloc((size_t)n * sizeof(int32_t)));
    HSD_HIP(ctx, hipMemcpy(dB.p, all.data(), all.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    std::vector<unsigned> bits(count.size());
    size_t sort_bytes = 0;
    for (size_t l = 0; l < count.size(); ++l) {
        unsigned rb = 1;
        while ((1ll << rb) < (long long)count[l]) ++rb;
        bits[l] = (unsigned)TKO_VALUE_BITS + rb;
        size_t need = 0;
        HSD_HIP(ctx, rocprim::radix_sort_pairs(nullptr, need, dK0.as<unsigned long long>(), dK1.as<unsigned long long>(), dV1.as<int32_t>(), d_order,
                                               (size_t)n, 0u, bits[l], ctx->stream));
        sort_bytes = std::max(sort_bytes, need);
    }
    HSD_HIP(ctx, dSort.alloc(sort_bytes));
    for (size_t l = 0; l < count.size(); ++l) {
        const int32_t *b = dB.as<int32_t>() + first[l];
        const int m = count[l];
        HSD_HIP(ctx, hipMemsetAsync(dMn.p, 0xff, (size_t)m * nf * sizeof(unsigned long long), ctx->stream));
        HSD_HIP(ctx, hipMemsetAsync(dMx.p, 0x00, (size_t)m * nf * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(tko_extent_kernel, grid, block, 0, ctx->stream, d_F, n, nf, (const int32_t *)d_order, b, m, dMn.as<unsigned long long>(),
                           dMx.as<unsigned long long>());
        hipLaunchKernelGGL(tko_key_kernel, grid, block, 0, ctx->stream, d_F, n, nf, (const int32_t *)d_order, b, m, (const unsigned long long *)dMn.as<unsigned long long>(),
                           (const unsigned long long *)dMx.as<unsigned long long>(), dK0.as<unsigned long long>(), dV1.as<int32_t>());
        size_t bytes = sort_bytes;
        HSD_HIP(ctx, rocprim::radix_sort_pairs(dSort.p, bytes, dK0.as<unsigned long long>(), dK1.as<unsigned long long>(), dV1.as<int32_t>(), d_order, (size_t)n, 0u,
                                               bits[l], ctx->stream));
    }
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    HSD_HIP(ctx, hipGetLastError());
    return HSD_OK;
}

namespace {
// position p of the order holds a node of the block [lo, hi)
struct InBlock {
    const int32_t *orig;
    int32_t lo, hi;
    __device__ bool operator()(const int32_t &p) const { const int32_t o = orig[p]; return o >= lo && o < hi; }
};
}  // namespace

// The rows of a block [r0, r1) of nodes as a SUBSEQUENCE of the graph's node order: the positions whose node lies in the block, ascending (one
// rocprim select over the positions; stable).  Consecutive block rows are then as similar as the order makes its neighbours, and 128 of them span a
// contiguous run of the order's tiles -- which is what the pruned k-nearest sweep of a row block pairs them with first (pair_topk.hip).
int hsd_tk_block_positions_device(hsd_ctx *ctx, const int32_t *d_orig, int64_t npos, int64_t r0, int64_t r1, int32_t *d_pos) {
    if (npos < 1 || npos > 0x7fffff00ll || r0 < 0 || r1 <= r0) return hsd_fail(ctx, HSD_ERR_INVALID, "block positions: bad range");
    Tmp dCount, dTmp;
    HSD_HIP(ctx, dCount.alloc(sizeof(size_t)));
    const InBlock pred{d_orig, (int32_t)r0, (int32_t)r1};
    rocprim::counting_iterator<int32_t> first(0);
    size_t bytes = 0;
    HSD_HIP(ctx, rocprim::select(nullptr, bytes, first, d_pos, dCount.as<size_t>(), (size_t)npos, pred, ctx->stream));
    HSD_HIP(ctx, dTmp.alloc(bytes));
    HSD_HIP(ctx, rocprim::select(dTmp.p, bytes, first, d_pos, dCount.as<size_t>(), (size_t)npos, pred, ctx->stream));
    size_t got = 0;
    HSD_HIP(ctx, hipMemcpyAsync(&got, dCount.p, sizeof got, hipMemcpyDeviceToHost, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((int64_t)got != r1 - r0) return hsd_fail(ctx, HSD_ERR_INTERNAL, "block positions: %zu of %lld nodes found in the order", got, (long long)(r1 - r0));
    return HSD_OK;
}
