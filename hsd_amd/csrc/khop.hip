// K1: CSR k-hop frontier sweep.  Replaces tools/hierarchy.py:16-38 (one Python-set BFS per node).
//
// One workgroup per source node, persistent over a strided list of sources.  Three bitmaps (n bits
// each) live in LDS: `cur` accumulates everything reached so far, `prev` is its state at the end of
// the previous level, so ring k = cur & ~prev, and `front` holds ring k-1.  A level is
//   expand : top-down, the frontier (ring k-1, held as a list in a per-workgroup global queue) ORs
//            the bits of all its neighbours into `cur` (LDS atomics, no return value needed); or,
//            when the frontier has more edges than the graph has unvisited nodes, bottom-up: every
//            unvisited node looks for one neighbour in `front`
//   scan   : every thread owns a contiguous range of bitmap words, counts the new bits, a
//            workgroup prefix sum turns counts into output positions, and the new members are
//            emitted IN ASCENDING ORDER -- into the queue (next frontier), into the per-hop value
//            histogram (LDS atomics on bins) and, on request, into the ragged member list.
// Ring membership is integer set arithmetic, so it is bit-exact by construction; the ascending
// order makes the ragged output directly comparable with the sorted reference rings.
#include "common.hpp"

#include <algorithm>

namespace {

constexpr int KH_THREADS = 256;

__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *scratch /* >= 5 words LDS */, uint32_t *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) scratch[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < KH_THREADS / 64; ++i) {
        const uint32_t s = scratch[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();                 // scratch reusable afterwards
    *total = tot;
    return base + inc - v;
}

template <int G /* lanes cooperating on one frontier node */>
__global__ __launch_bounds__(KH_THREADS) void khop_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int n, int hop, int64_t r0, int64_t r1,
    const int32_t *__restrict__ node_bin, int n_bins, uint32_t *__restrict__ counts /* [rows][hop+1][n_bins] or null */,
    int32_t *__restrict__ sizes /* [(r1-r0)][hop+1] or null */, const int64_t *__restrict__ offsets /* or null */,
    int32_t *__restrict__ members /* or null */, int32_t *__restrict__ queue_scratch, int words, int wpt, float avg_deg) {
    extern __shared__ uint32_t lds[];
    uint32_t *cur = lds;
    uint32_t *prev = lds + words;
    const bool bottom_up = avg_deg > 0.0f;               // host decision: the third bitmap exists only then
    uint32_t *front = prev + words;                      // ring k-1 as a bitmap (bottom-up expansion)
    uint32_t *hist = prev + (bottom_up ? 2 : 1) * words; // (hop+1)*n_bins when counts != null
    const int n_hist = counts ? (hop + 1) * n_bins : 0;
    uint32_t *scratch = hist + n_hist;                    // 8 words

    int32_t *queue = queue_scratch + (int64_t)blockIdx.x * n;
    const int tid = threadIdx.x;
    const int w_lo = min(tid * wpt, words), w_hi = min(w_lo + wpt, words);

    for (int64_t src = r0 + blockIdx.x; src < r1; src += gridDim.x) {
        for (int i = tid; i < (bottom_up ? 3 : 2) * words + n_hist; i += KH_THREADS) lds[i] = 0u;
        __syncthreads();
        if (tid == 0) cur[src >> 5] = 1u << (src & 31);
        __syncthreads();

        int qs = 0, qe = 0;
        const int64_t srow = src - r0;
        for (int k = 0; k <= hop; ++k) {
            if (k > 0) {
                // ---- expand ring k-1 ----
                // top-down: every frontier node ORs its neighbours into `cur`.  Once the frontier's edges
                // outnumber what is left of the graph (the last hops of a small-world graph), bottom-up is
                // cheaper: every UNVISITED node looks for one neighbour in the frontier bitmap and stops at
                // the first hit; each thread owns its words of `cur`, so no atomics.
                const float unvisited = (float)(n - qe);
                if (bottom_up && (float)(qe - qs) * avg_deg > 4.0f * unvisited) {
                    for (int w = w_lo; w < w_hi; ++w) {
                        uint32_t unv = ~cur[w];
                        if (w == words - 1 && (n & 31)) unv &= (1u << (n & 31)) - 1u;
                        uint32_t found = 0u;
                        while (unv) {
                            const int b = __ffs(unv) - 1;
                            unv &= unv - 1;
                            const int v = (w << 5) + b;
                            for (int64_t e = rowptr[v], end = rowptr[v + 1]; e < end; ++e) {
                                const int u = col[e];
                                if ((front[u >> 5] >> (u & 31)) & 1u) { found |= 1u << b; break; }
                            }
                        }
                        if (found) cur[w] |= found;
                    }
                } else {
                    const int group = tid / G, gl = tid % G;
                    for (int f = qs + group; f < qe; f += KH_THREADS / G) {
                        const int u = queue[f];
                        const int64_t beg = rowptr[u], end = rowptr[u + 1];
                        for (int64_t e = beg + gl; e < end; e += G) {
                            const int v = col[e];
                            const uint32_t bit = 1u << (v & 31);
                            if (!(cur[v >> 5] & bit)) atomicOr(&cur[v >> 5], bit);
                        }
                    }
                }
                __syncthreads();
            }
            // ---- scan: ring k = cur & ~prev, emitted in ascending order ----
            uint32_t cnt = 0;
            for (int w = w_lo; w < w_hi; ++w) cnt += __popc(cur[w] & ~prev[w]);
            uint32_t total;
            const uint32_t excl = block_exclusive_scan(cnt, scratch, &total);
            if (tid == 0 && sizes) sizes[srow * (hop + 1) + k] = (int32_t)total;
            if (total == 0) {
                // component exhausted: the remaining rings are empty (tools/hierarchy.py:36-37
                // keeps appending empty layers)
                if (sizes && tid == 0)
                    for (int kk = k + 1; kk <= hop; ++kk) sizes[srow * (hop + 1) + kk] = 0;
                break;
            }
            int pos = qe + (int)excl;
            int32_t *mem_out = nullptr;
            if (members) mem_out = members + offsets[srow * (hop + 1) + k] - qe;   // indexed by pos
            for (int w = w_lo; w < w_hi; ++w) {
                const uint32_t c = cur[w];
                uint32_t nw = c & ~prev[w];
                prev[w] = c;
                if (bottom_up) front[w] = nw;
                while (nw) {
                    const int b = __ffs(nw) - 1;
                    nw &= nw - 1;
                    const int v = (w << 5) + b;
                    if (k < hop) queue[pos] = v;          // the last ring is never expanded: no frontier entry
                    if (mem_out) mem_out[pos] = v;
                    if (counts) atomicAdd(&hist[k * n_bins + node_bin[v]], 1u);
                    ++pos;
                }
            }
            qs = qe;
            qe += (int)total;
            __syncthreads();
        }
        if (counts) {
            __syncthreads();
            uint32_t *dst = counts + src * (int64_t)n_hist;
            for (int i = tid; i < n_hist; i += KH_THREADS) dst[i] = hist[i];
        }
        __syncthreads();
    }
}

template <int G>
hipError_t launch_g(hsd_ctx *ctx, int grid, size_t lds_bytes, int hop, int64_t r0, int64_t r1, uint32_t *d_counts,
                    int32_t *d_sizes, const int64_t *d_offsets, int32_t *d_members, int words, int wpt, bool bottom_up) {
    auto kern = khop_kernel<G>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(KH_THREADS), lds_bytes, ctx->stream, ctx->d_rowptr, ctx->d_col, (int)ctx->n,
                       hop, r0, r1, ctx->d_node_bin, ctx->n_bins, d_counts, d_sizes, d_offsets, d_members, ctx->d_queue,
                       words, wpt, bottom_up ? (float)((double)ctx->nnz / (double)std::max<int64_t>(ctx->n, 1)) : 0.0f);
    return hipGetLastError();
}

}  // namespace

int hsd_launch_khop(hsd_ctx *ctx, int hop, int64_t r0, int64_t r1, bool hist, int32_t *d_sizes, const int64_t *d_offsets,
                    int32_t *d_members) {
    const int64_t n = ctx->n;
    // histograms / ring sizes only: the bit-parallel sweep of 4096 sources at a time (khop_ms.hip) once the graph is large
    // enough to fill its planes (a batch costs the same for one source as for 4096: short source ranges stay with the
    // per-source sweep); HSD_KHOP_MS=0 / 1 forces the per-source sweep / the bit-parallel one.  The bit-parallel sweep needs
    // 3 n x 512 B of planes + 2 (hop - 1) n x 512 B of ring planes: if they cannot be allocated (a large graph next to its
    // own [n][hop + 1][bins] histogram tensor) the per-source sweep below, which needs LDS bitmaps only, takes over.
    const auto ms_wanted = [&]() {
        if (const char *em = hsd_hook(H_KHOP_MS)) return atoi(em) != 0;
        return n >= 8192 && hop >= 2 && r1 - r0 >= 2048;
    };
    if (!hist && d_sizes && !d_members && r1 > r0 && ms_wanted()) {
        // ring sizes alone (hsd_khop_sizes: hierarchical degree, SIR labels): the bit-parallel sweep over ONE bin counts them
        int32_t *d_zero = nullptr;
        int rc = hsd_dev_alloc(ctx, &d_zero, (size_t)n);
        if (rc == HSD_OK) {
            hipError_t e = hipMemsetAsync(d_zero, 0, (size_t)n * sizeof(int32_t), ctx->stream);
            if (e == hipSuccess) rc = hsd_launch_khop_ms(ctx, hop, r0, r1, reinterpret_cast<uint32_t *>(d_sizes) - r0 * (hop + 1), d_zero, 1);
            const hipError_t e2 = hipStreamSynchronize(ctx->stream);          // d_zero is read until the sweep has finished
            hsd_dev_free(d_zero);
            if (e != hipSuccess || e2 != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "khop sizes: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        }
        if (rc != HSD_INT_NO_PLANES) return rc;
        (void)hipGetLastError();                          // planes did not fit: per-source sweep
    } else if (hist && !d_sizes && !d_members && hsd_khop_ms_usable(ctx, hop, ctx->n_bins) && ms_wanted()) {
        const int rc = hsd_launch_khop_ms(ctx, hop, r0, r1);
        if (rc != HSD_INT_NO_PLANES) return rc;
        (void)hipGetLastError();
    }
    const int words = (int)((n + 31) / 32);
    const int n_hist = hist ? (hop + 1) * ctx->n_bins : 0;
    const size_t lds_cap = ctx->prop.maxSharedMemoryPerMultiProcessor ? ctx->prop.maxSharedMemoryPerMultiProcessor : 65536;
    // The frontier bitmap of the bottom-up expansion costs LDS, i.e. resident workgroups per CU: it is taken
    // for deep sweeps (hop >= 4: the last hops cover most of a small-world graph) and whenever it is free.
    const size_t lds2 = ((size_t)2 * words + n_hist + 8) * sizeof(uint32_t), lds3 = lds2 + (size_t)words * sizeof(uint32_t);
    const char *eb = hsd_hook(H_KHOP_BOTTOMUP);        // 0 / 1 force it off / on
    bool bottom_up = lds3 <= lds_cap && (hop >= 4 || std::min<size_t>(8, lds_cap / lds3) == std::min<size_t>(8, lds_cap / std::max<size_t>(lds2, 1)));
    if (eb) bottom_up = atoi(eb) != 0 && lds3 <= lds_cap;
    const size_t lds_bytes = bottom_up ? lds3 : lds2;
    if (lds_bytes > lds_cap)
        return hsd_fail(ctx, HSD_ERR_TOO_LARGE,
                        "k-hop sweep needs %zu B of LDS (2 visited bitmaps of %lld bits + %d histogram bins), CU has %zu",
                        lds_bytes, (long long)n, n_hist, lds_cap);
    int per_cu = (int)std::min<size_t>(8, lds_cap / lds_bytes);
    if (per_cu < 1) per_cu = 1;
    int64_t grid = std::min<int64_t>(r1 - r0, (int64_t)ctx->prop.multiProcessorCount * per_cu);
    if (grid < 1) return HSD_OK;
    if (ctx->queue_blocks < grid) {
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hsd_dev_free(ctx->d_queue);
        int rc = hsd_dev_alloc(ctx, &ctx->d_queue, (size_t)grid * (size_t)n);
        if (rc) return rc;
        ctx->queue_blocks = grid;
    }
    int wpt = (words + KH_THREADS - 1) / KH_THREADS;
    wpt |= 1;                                   // odd word stride per thread: no LDS bank conflicts in the scan
    const double avg_deg = n ? (double)ctx->nnz / (double)n : 0.0;
    hipError_t e;
    uint32_t *d_counts = hist ? ctx->d_counts : nullptr;
    if (avg_deg < 12.0)
        e = launch_g<8>(ctx, (int)grid, lds_bytes, hop, r0, r1, d_counts, d_sizes, d_offsets, d_members, words, wpt, bottom_up);
    else if (avg_deg < 48.0)
        e = launch_g<16>(ctx, (int)grid, lds_bytes, hop, r0, r1, d_counts, d_sizes, d_offsets, d_members, words, wpt, bottom_up);
    else
        e = launch_g<64>(ctx, (int)grid, lds_bytes, hop, r0, r1, d_counts, d_sizes, d_offsets, d_members, words, wpt, bottom_up);
    if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "khop launch: %s", hipGetErrorString(e));
    return HSD_OK;
}

static int check_rows(hsd_ctx *ctx, int hop, int64_t r0, int64_t r1) {
    if (ctx->n <= 0) return hsd_fail(ctx, HSD_ERR_NO_GRAPH, "no graph uploaded");
    if (hop < 0 || hop >= HSD_MAX_LAYERS) return hsd_fail(ctx, HSD_ERR_INVALID, "hop=%d out of range [0,%d)", hop, HSD_MAX_LAYERS);
    if (r0 < 0 || r1 > ctx->n || r0 > r1)
        return hsd_fail(ctx, HSD_ERR_INVALID, "row range [%lld,%lld) outside [0,%lld)", (long long)r0, (long long)r1, (long long)ctx->n);
    return HSD_OK;
}

extern "C" int hsd_khop_sizes(hsd_ctx *ctx, int hop, int64_t r0, int64_t r1, int32_t *sizes) try {
    HSD_CHECK_CTX(ctx);
    int rc = check_rows(ctx, hop, r0, r1);
    if (rc) return rc;
    if (r0 == r1) return HSD_OK;
    if (!sizes) return hsd_fail(ctx, HSD_ERR_INVALID, "null sizes");
    const size_t cnt = (size_t)(r1 - r0) * (hop + 1);
    int32_t *d_sizes = nullptr;
    if ((rc = hsd_dev_alloc(ctx, &d_sizes, cnt))) return rc;
    rc = hsd_launch_khop(ctx, hop, r0, r1, false, d_sizes, nullptr, nullptr);
    if (rc == HSD_OK) {
        hipError_t e = hipMemcpyAsync(sizes, d_sizes, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_khop_sizes: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(ctx->stream);
    hsd_dev_free(d_sizes);
    return rc;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_khop_members(hsd_ctx *ctx, int hop, int64_t r0, int64_t r1, const int64_t *offsets, int32_t *members) try {
    HSD_CHECK_CTX(ctx);
    int rc = check_rows(ctx, hop, r0, r1);
    if (rc) return rc;
    if (r0 == r1) return HSD_OK;
    if (!offsets) return hsd_fail(ctx, HSD_ERR_INVALID, "null offsets");
    const size_t n_off = (size_t)(r1 - r0) * (hop + 1) + 1;
    const int64_t total = offsets[n_off - 1];
    if (offsets[0] != 0 || total < 0) return hsd_fail(ctx, HSD_ERR_INVALID, "offsets must start at 0");
    if (total > 0 && !members) return hsd_fail(ctx, HSD_ERR_INVALID, "null members");
    // pass 1: ring sizes only -- the offsets are validated against them BEFORE any member is
    // written, so a wrong offsets array can never turn into an out-of-bounds device store
    int64_t *d_off = nullptr;
    int32_t *d_mem = nullptr, *d_sizes = nullptr;
    if ((rc = hsd_dev_alloc(ctx, &d_sizes, n_off))) return rc;
    std::vector<int32_t> sizes(n_off - 1);
    rc = hsd_launch_khop(ctx, hop, r0, r1, false, d_sizes, nullptr, nullptr);
    if (rc == HSD_OK) {
        hipError_t e = hipMemcpyAsync(sizes.data(), d_sizes, (n_off - 1) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_khop_members: %s", hipGetErrorString(e));
    }
    if (rc == HSD_OK)
        for (size_t i = 0; i + 1 < n_off; ++i)
            if (offsets[i + 1] - offsets[i] != sizes[i]) {
                rc = hsd_fail(ctx, HSD_ERR_INVALID, "offsets disagree with ring sizes at entry %zu (%lld vs %d)", i,
                              (long long)(offsets[i + 1] - offsets[i]), sizes[i]);
                break;
            }
    // pass 2: emit members
    if (rc == HSD_OK) rc = hsd_dev_alloc(ctx, &d_off, n_off);
    if (rc == HSD_OK) rc = hsd_dev_alloc(ctx, &d_mem, (size_t)total);
    if (rc == HSD_OK) {
        hipError_t e = hipMemcpyAsync(d_off, offsets, n_off * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_khop_members: %s", hipGetErrorString(e));
    }
    if (rc == HSD_OK) rc = hsd_launch_khop(ctx, hop, r0, r1, false, d_sizes, d_off, d_mem);
    if (rc == HSD_OK && total) {
        hipError_t e = hipMemcpyAsync(members, d_mem, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hsd_fail(ctx, HSD_ERR_HIP, "hsd_khop_members: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(ctx->stream);
    hsd_dev_free(d_off);
    hsd_dev_free(d_mem);
    hsd_dev_free(d_sizes);
    return rc;
} HSD_ABI_GUARD(ctx)
