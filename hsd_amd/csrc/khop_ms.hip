// K1 + K2, bit-parallel form: the k-hop sweep of 4096 sources at once (replaces tools/hierarchy.py:16-38 and the ring
// histograms behind model/HSD.py:71-94 when only the histograms are wanted; khop.hip stays for ragged member lists).
//
// A batch is 4096 consecutive sources; source s of the batch is bit s % 64 of word s / 64.  Per batch three planes of
// n rows x 64 words (512 B, one wave-wide access) live in HBM:
//   V  every source that has reached the row's vertex so far
//   F  ring k-1 (the frontier), Fn ring k
// and ring k of every vertex, k >= 2, is also written word-major (RT[k][word][vertex]) for the histogram pass.
//   ring 0 / ring 1 : one workgroup per source walks its adjacency row once, writes both histogram rows and sets the
//                     source's bit in V / F of its neighbours (returning atomics: duplicate edges and self loops count once)
//   ring k >= 2     : pull.  One wave per vertex i ORs the frontier rows of its neighbours (the neighbour index is
//                     wave-uniform, every gather is one 512-B row), new = acc & ~V[i].  Rows longer than 2 * HSD_LCHUNK
//                     entries are ORed chunk-wise by a pre-pass.  A vertex already reached by every source of the batch is
//                     skipped (the last hops of a small-world graph).  (A flag per frontier row that lets the pull skip empty
//                     rows -- most rows when the frontier is ring 1 -- was measured: the dependent scalar load in front of
//                     every gather costs more than the rows it saves, 86 -> 119 us per pull at BA N = 100k.)  Work per hop is nnz * n / 64 word ORs whatever the
//                     ring sizes, against sum of ring sizes * mean degree single-bit operations of the per-source sweep.
//   histograms      : one workgroup per (word, hop, vertex share): counts[64 sources][bins] in LDS; blocks of 64 vertices are
//                     transposed across the wave (lane = source) and counted bin by bin with popcounts, then added to the
//                     counts tensor row by row (coalesced).
// Membership is integer set arithmetic: the histograms are bit-identical to khop.hip's (tests/test_gpu_khop_ms.py).
#include "common.hpp"

#include <algorithm>
#include <type_traits>

namespace {

constexpr int MS_WORDS = 64;                 // words per plane row
constexpr int MS_BATCH = MS_WORDS * 64;      // sources per batch
constexpr int MS_ROWS = 16;                  // vertices per workgroup of the pull kernel (one 128-B line of RT per word)

__device__ inline uint64_t valid_word(int nsrc, int w) {
    const int left = nsrc - 64 * w;
    return left >= 64 ? ~0ull : (left <= 0 ? 0ull : ((1ull << left) - 1ull));
}

// rings 0 and 1 of the batch's sources; one workgroup (64 threads) per source
__global__ __launch_bounds__(64) void ms_ring01_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t src0,
                                                      int hop, const int32_t *__restrict__ node_bin, int n_bins,
                                                      uint32_t *__restrict__ counts, unsigned long long *__restrict__ V,
                                                      unsigned long long *__restrict__ F) {
    extern __shared__ uint32_t hist[];
    const int s = blockIdx.x;
    const int64_t src = src0 + s;
    const int w = s >> 6;
    const unsigned long long bit = 1ull << (s & 63);
    for (int b = threadIdx.x; b < n_bins; b += 64) hist[b] = 0u;
    if (threadIdx.x == 0) (void)atomicOr(&V[src * MS_WORDS + w], bit);     // returning: performed before the barrier
    __syncthreads();
    if (hop >= 1)
        for (int64_t p = rowptr[src] + threadIdx.x, e = rowptr[src + 1]; p < e; p += 64) {
            const int u = col[p];
            const unsigned long long old = atomicOr(&V[(int64_t)u * MS_WORDS + w], bit);
            if (!(old & bit)) {                                            // first time this source meets u
                atomicAdd(&hist[node_bin[u]], 1u);
                if (hop >= 2) atomicOr(&F[(int64_t)u * MS_WORDS + w], bit);
            }
        }
    __syncthreads();
    uint32_t *row0 = counts + src * (int64_t)(hop + 1) * n_bins;
    const int own = node_bin[src];
    for (int b = threadIdx.x; b < n_bins; b += 64) {
        row0[b] = b == own ? 1u : 0u;
        if (hop >= 1) row0[n_bins + b] = hist[b];
    }
}

// OR of the frontier rows of one chunk of a long adjacency row: P[chunk][lane]
__global__ __launch_bounds__(256) void ms_partial_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                        const unsigned long long *__restrict__ F, const int32_t *__restrict__ chunk_row,
                                                        const int64_t *__restrict__ chunk_p0, int n_chunks,
                                                        unsigned long long *__restrict__ P) {
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (q >= n_chunks) return;
    const int64_t b = chunk_p0[q], e = min(b + (int64_t)HSD_LCHUNK, rowptr[chunk_row[q] + 1]);
    unsigned long long acc = 0ull;
    int64_t p = b;
    for (; p + 8 <= e; p += 8) {
        unsigned long long x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = F[(int64_t)col[p + u] * MS_WORDS + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc |= x[u];
    }
    for (; p < e; ++p) acc |= F[(int64_t)col[p] * MS_WORDS + lane];
    P[(int64_t)q * MS_WORDS + lane] = acc;
}

// ring k of MS_ROWS vertices: new = (OR of the neighbours' frontier rows) & ~V
__global__ __launch_bounds__(256) void ms_pull_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int n,
                                                     int nsrc, const unsigned long long *__restrict__ F,
                                                     unsigned long long *__restrict__ V, unsigned long long *__restrict__ Fn /* or null */,
                                                     unsigned long long *__restrict__ Fzero /* last hop: plane to clear, or null */,
                                                     unsigned long long *__restrict__ RT, int64_t ldn,
                                                     const int32_t *__restrict__ row_first /* or null */,
                                                     const unsigned long long *__restrict__ P) {
    __shared__ unsigned long long tile[MS_ROWS][MS_WORDS];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i0 = blockIdx.x * MS_ROWS;
    const unsigned long long vmask = valid_word(nsrc, lane);
    for (int r = 0; r < MS_ROWS / 4; ++r) {
        const int i = i0 + wv * (MS_ROWS / 4) + r;              // wave-uniform
        unsigned long long nw = 0ull;
        if (i < n) {
            const unsigned long long vis = V[(int64_t)i * MS_WORDS + lane];
            unsigned long long acc = 0ull;
            if (__any((vis & vmask) != vmask)) {                 // somebody has not been here yet
                const int64_t b = rowptr[i], e = rowptr[i + 1];
                const int ch0 = row_first ? row_first[i] : -1;
                if (ch0 >= 0) {
                    const int nch = (int)((e - b + HSD_LCHUNK - 1) / HSD_LCHUNK);
                    for (int q = 0; q < nch; ++q) acc |= P[(int64_t)(ch0 + q) * MS_WORDS + lane];
                } else {
                    int64_t p = b;
                    for (; p + 8 <= e; p += 8) {
                        unsigned long long x[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = F[(int64_t)col[p + u] * MS_WORDS + lane];
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc |= x[u];
                    }
                    for (; p < e; ++p) acc |= F[(int64_t)col[p] * MS_WORDS + lane];
                }
            }
            nw = acc & ~vis;
            if (Fn) {
                if (nw) V[(int64_t)i * MS_WORDS + lane] = vis | nw;
                Fn[(int64_t)i * MS_WORDS + lane] = nw;
            } else {
                // last hop of the batch: V is dead -- leave it (and the frontier plane nobody reads any more) cleared for the
                // next batch instead of two 50-100 MB memsets between the batches
                if (vis) V[(int64_t)i * MS_WORDS + lane] = 0ull;
                if (Fzero) Fzero[(int64_t)i * MS_WORDS + lane] = 0ull;
            }
        }
        tile[wv * (MS_ROWS / 4) + r][lane] = nw;
    }
    __syncthreads();
    // word-major copy: word j of the 16 vertices is one 128-B line
    const int j = threadIdx.x >> 2, part = threadIdx.x & 3;
    unsigned long long *dst = RT + (int64_t)j * ldn + i0 + part * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = tile[part * 4 + k][j];   // ldn is padded: rows >= n hold zeros
}

// 64 x 64 bit transpose across the wave: lane r holds row r (lo = bits 0..31, hi = bits 32..63); afterwards lane s holds
// column s (bit v = row v's bit s).  Stage k swaps the (row bit k clear, column bit k set) block with its mirror image: a
// lane needs half of its partner's bits, so both 32-bit halves travel in ONE 32-bit exchange per stage.
__device__ inline unsigned long long wave_transpose64(unsigned long long x, int lane) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    {   // k = 32: the partner's low word becomes my high word (or the other way round)
        const uint32_t r = __shfl_xor((lane & 32) ? lo : hi, 32, 64);
        if (lane & 32) lo = r; else hi = r;
    }
    auto stage = [&](auto K, uint32_t m) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        const bool up = (lane & k) != 0;
        // what the partner needs from me, of both halves, packed on complementary bit positions
        const uint32_t send = up ? ((lo & m) | ((hi & m) << k)) : (((lo >> k) & m) | (hi & ~m));
        const uint32_t r = __shfl_xor(send, k, 64);
        if (up) { lo = (lo & ~m) | (r & m); hi = (hi & ~m) | ((r >> k) & m); }
        else    { lo = (lo & m) | ((r & m) << k); hi = (hi & m) | (r & ~m); }
    };
    stage(std::integral_constant<int, 16>{}, 0x0000ffffu);
    stage(std::integral_constant<int, 8>{}, 0x00ff00ffu);
    stage(std::integral_constant<int, 4>{}, 0x0f0f0f0fu);
    stage(std::integral_constant<int, 2>{}, 0x33333333u);
    stage(std::integral_constant<int, 1>{}, 0x55555555u);
    return ((unsigned long long)hi << 32) | lo;
}

// counts[src][hop][bin] += number of vertices of that bin whose ring word has the source's bit.
// One workgroup per (word of 64 sources, vertex share, ring); its waves take blocks of 64 vertices (lane = vertex, one
// coalesced 512-B load).  A block with few non-zero words adds its bits one by one; otherwise the 64 x 64 bit block is
// transposed across the wave (lane = source) and every bin present in the block costs one ballot, one AND, one popcount
// and one LDS add per lane -- independent of how dense the ring is.
constexpr int MS_HIST_THREADS = 1024;          // 16 waves share one LDS table: the per-block chain (load, 6 exchanges, bins) is latency
constexpr int MS_HIST_WAVES = MS_HIST_THREADS / 64;

__device__ inline void ms_hist_block(unsigned long long x, int v, int n, int lane, const int32_t *__restrict__ node_bin, uint32_t *hist, int hs) {
    const unsigned long long nz = __ballot(x != 0ull);
    if (!nz) return;
    const int bin = v < n ? node_bin[v] : 0;
    if (__popcll(nz) <= 20) {                      // sparse ring (ring 2 of a batch): a few single bits
        while (x) {
            const int s = __builtin_ctzll(x);
            x &= x - 1;
            atomicAdd(&hist[s * hs + bin], 1u);
        }
        return;
    }
    const unsigned long long t = wave_transpose64(x, lane);          // lane = source, bit = vertex of the block
    unsigned long long left = nz;                                    // vertices whose bin is still to be counted
    while (left) {
        const int b = __builtin_amdgcn_readlane(bin, __builtin_ctzll(left));
        const unsigned long long same = __ballot(bin == b) & left;
        left &= ~same;
        const uint32_t c = (uint32_t)__popcll(t & same);
        if (c) atomicAdd(&hist[lane * hs + b], c);
    }
}

__global__ __launch_bounds__(MS_HIST_THREADS) void ms_hist_kernel(const unsigned long long *__restrict__ RT, int64_t ldn, int64_t plane_stride,
                                                                 int n, int parts, int hop0, int L, const int32_t *__restrict__ node_bin,
                                                                 int n_bins, int64_t src0, int nsrc, uint32_t *__restrict__ counts) {
    extern __shared__ uint32_t hist[];           // [64][hs], hs odd: the 64 sources of a bin sit in 64 different banks
    const int j = blockIdx.x, part = blockIdx.y, pl = blockIdx.z;
    if (j * 64 >= nsrc) return;
    const int hs = n_bins | 1;
    for (int t = threadIdx.x; t < 64 * hs; t += MS_HIST_THREADS) hist[t] = 0u;
    __syncthreads();
    const unsigned long long *w = RT + (int64_t)pl * plane_stride + (int64_t)j * ldn;
    const int lane = threadIdx.x & 63;
    const int nblk = (n + 63) / 64, step = parts * MS_HIST_WAVES;
    for (int blk = part * MS_HIST_WAVES + (threadIdx.x >> 6); blk < nblk; blk += 4 * step) {
        // four blocks' words requested together
        const int v0 = blk * 64 + lane, v1 = v0 + step * 64, v2 = v1 + step * 64, v3 = v2 + step * 64;
        const unsigned long long x0 = v0 < n ? w[v0] : 0ull, x1 = v1 < n ? w[v1] : 0ull, x2 = v2 < n ? w[v2] : 0ull, x3 = v3 < n ? w[v3] : 0ull;
        ms_hist_block(x0, v0, n, lane, node_bin, hist, hs);
        ms_hist_block(x1, v1, n, lane, node_bin, hist, hs);
        ms_hist_block(x2, v2, n, lane, node_bin, hist, hs);
        ms_hist_block(x3, v3, n, lane, node_bin, hist, hs);
    }
    __syncthreads();
    const int ns = min(64, nsrc - j * 64);
    for (int t = threadIdx.x; t < ns * n_bins; t += MS_HIST_THREADS) {
        const int s = t / n_bins, b = t - s * n_bins;
        const uint32_t c = hist[s * hs + b];
        if (c) atomicAdd(&counts[((src0 + j * 64 + s) * L + hop0 + pl) * (int64_t)n_bins + b], c);
    }
}

}  // namespace

// long adjacency rows, split into chunks for ms_partial_kernel (built once per graph)
static int ensure_adj_chunks(hsd_ctx *ctx) {
    if (ctx->ms_chunks_ready) return HSD_OK;
    const int64_t n = ctx->n;
    std::vector<int32_t> row_first((size_t)n, -1), chunk_row;
    std::vector<int64_t> chunk_p0;
    int64_t p = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t len = ctx->h_degree[(size_t)i];
        if (len > 2 * HSD_LCHUNK) {
            row_first[(size_t)i] = (int32_t)chunk_row.size();
            for (int64_t q = p; q < p + len; q += HSD_LCHUNK) {
                chunk_row.push_back((int32_t)i);
                chunk_p0.push_back(q);
            }
        }
        p += len;
    }
    hsd_dev_free(ctx->d_ms_row_first);
    hsd_dev_free(ctx->d_ms_chunk_row);
    hsd_dev_free(ctx->d_ms_chunk_p0);
    ctx->ms_n_chunks = (int)chunk_row.size();
    if (ctx->ms_n_chunks) {
        int rc;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_ms_row_first, (size_t)n))) return rc;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_ms_chunk_row, chunk_row.size()))) return rc;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_ms_chunk_p0, chunk_p0.size()))) return rc;
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_ms_row_first, row_first.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_ms_chunk_row, chunk_row.data(), chunk_row.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_ms_chunk_p0, chunk_p0.data(), chunk_p0.size() * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));         // the host vectors go out of scope
    }
    ctx->ms_chunks_ready = true;
    return HSD_OK;
}

void hsd_khop_ms_release(hsd_ctx *ctx) {
    if (ctx->ms_stream) (void)hipStreamSynchronize(ctx->ms_stream);    // a launch that failed half way may have left a histogram pass behind
    hsd_dev_free(ctx->d_ms_row_first);
    hsd_dev_free(ctx->d_ms_chunk_row);
    hsd_dev_free(ctx->d_ms_chunk_p0);
    hsd_dev_free(ctx->d_ms_planes);
    ctx->ms_plane_words = 0;
    ctx->ms_n_chunks = 0;
    ctx->ms_chunks_ready = false;
}

// can the bit-parallel sweep serve this histogram build?  (LDS: 64 sources x n_bins counters per workgroup)
bool hsd_khop_ms_usable(hsd_ctx *ctx, int hop, int n_bins) {
    const size_t lds_cap = ctx->prop.maxSharedMemoryPerMultiProcessor ? ctx->prop.maxSharedMemoryPerMultiProcessor : 65536;
    return hop >= 0 && (size_t)64 * (n_bins | 1) * sizeof(uint32_t) <= std::min<size_t>(lds_cap, 128 * 1024);
}

// counts: [.][hop + 1][n_bins] indexed by the GLOBAL source id (the caller offsets the pointer for a row range);
// node_bin: [n] bin of every vertex.  The context's own histogram tensor by default; hsd_khop_sizes passes one bin.
int hsd_launch_khop_ms(hsd_ctx *ctx, int hop, int64_t r0, int64_t r1, uint32_t *counts, const int32_t *node_bin, int n_bins) {
    const int64_t n = ctx->n;
    if (!counts) { counts = ctx->d_counts; node_bin = ctx->d_node_bin; n_bins = ctx->n_bins; }
    const int L = hop + 1, B = n_bins;
    if (r1 <= r0) return HSD_OK;
    int rc = ensure_adj_chunks(ctx);
    if (rc) return rc;
    const int64_t ldn = round_up(n, MS_ROWS);
    const int n_planes = std::max(0, hop - 1);                   // rings 2..hop
    const int64_t plane = n * MS_WORDS, rt_plane = (int64_t)MS_WORDS * ldn;
    // the histogram pass of a batch runs on a second stream beside the next batch's pulls (one is bound by the fabric,
    // the other by the LDS and the vector ALU): two sets of ring planes
    const bool overlap = n_planes > 0 && r1 - r0 > MS_BATCH;
    const int n_rt = overlap ? 2 : 1;
    const size_t words = (size_t)3 * plane + (size_t)n_rt * n_planes * rt_plane + (size_t)std::max(1, ctx->ms_n_chunks) * MS_WORDS;
    if (ctx->ms_plane_words < words) {
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hsd_dev_free(ctx->d_ms_planes);
        ctx->ms_plane_words = 0;
        // (no room for the planes: the caller falls back to the per-source sweep, which needs LDS bitmaps only;
        // HSD_KHOP_MS_NOPLANES=1 rehearses that in the tests)
        if (hsd_hook(H_KHOP_MS_NOPLANES)) return HSD_INT_NO_PLANES;
        if (hipMalloc(reinterpret_cast<void **>(&ctx->d_ms_planes), words * sizeof(*ctx->d_ms_planes)) != hipSuccess) {
            ctx->d_ms_planes = nullptr;
            (void)hipGetLastError();
            return HSD_INT_NO_PLANES;
        }
        ctx->ms_plane_words = words;
    }
    if (overlap && !ctx->ms_stream) {
        HSD_HIP(ctx, hipStreamCreateWithFlags(&ctx->ms_stream, hipStreamNonBlocking));
        for (int t = 0; t < 2; ++t) {
            HSD_HIP(ctx, hipEventCreateWithFlags(&ctx->ms_ev_rings[t], hipEventDisableTiming));
            HSD_HIP(ctx, hipEventCreateWithFlags(&ctx->ms_ev_hist[t], hipEventDisableTiming));
        }
    }
    unsigned long long *V = ctx->d_ms_planes, *Fa = V + plane, *Fb = Fa + plane, *RT0 = Fb + plane;
    unsigned long long *P = RT0 + (size_t)n_rt * n_planes * rt_plane;
    const size_t lds_hist = (size_t)64 * (B | 1) * sizeof(uint32_t);
    if (n_planes) {
        rc = hsd_raise_lds_limit(ctx, reinterpret_cast<const void *>(ms_hist_kernel), lds_hist);
        if (rc) return rc;
    }
    const unsigned n_rb = (unsigned)((n + MS_ROWS - 1) / MS_ROWS);
    int batch = 0;
    for (int64_t b0 = r0; b0 < r1; b0 += MS_BATCH, ++batch) {
        const int nsrc = (int)std::min<int64_t>(MS_BATCH, r1 - b0);
        const int set = overlap ? batch & 1 : 0;
        unsigned long long *RT = RT0 + (size_t)set * n_planes * rt_plane;
        // V and Fa start cleared: by memset for the first batch, afterwards by the previous batch's last pull (which can
        // clear Fa only if it did not read it: the frontier of hop k sits in Fa for even k)
        const bool fa_cleared_by_pull = hop >= 3 && (hop & 1);
        if (batch == 0 || hop < 2) HSD_HIP(ctx, hipMemsetAsync(V, 0, (size_t)(hop >= 2 ? 2 : 1) * plane * sizeof(unsigned long long), ctx->stream));
        else if (!fa_cleared_by_pull) HSD_HIP(ctx, hipMemsetAsync(Fa, 0, (size_t)plane * sizeof(unsigned long long), ctx->stream));
        if (n_planes)                    // rings 2..hop are accumulated by atomics: clear their rows (rings 0 / 1 are stored)
            HSD_HIP(ctx, hipMemsetAsync(counts + (int64_t)b0 * L * B, 0, (size_t)nsrc * L * B * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(ms_ring01_kernel, dim3((unsigned)nsrc), dim3(64), (size_t)B * sizeof(uint32_t), ctx->stream, ctx->d_rowptr,
                           ctx->d_col, b0, hop, node_bin, B, counts, V, Fa);
        if (overlap && batch >= 2) HSD_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ms_ev_hist[set], 0));   // this set's planes have been read
        unsigned long long *F = Fa, *Fn = Fb;
        for (int k = 2; k <= hop; ++k) {
            if (ctx->ms_n_chunks)
                hipLaunchKernelGGL(ms_partial_kernel, dim3((unsigned)((ctx->ms_n_chunks + 3) / 4)), dim3(256), 0, ctx->stream, ctx->d_rowptr,
                                   ctx->d_col, F, ctx->d_ms_chunk_row, ctx->d_ms_chunk_p0, ctx->ms_n_chunks, P);
            hipLaunchKernelGGL(ms_pull_kernel, dim3(n_rb), dim3(256), 0, ctx->stream, ctx->d_rowptr, ctx->d_col, (int)n, nsrc, F, V,
                               k < hop ? Fn : nullptr, (k == hop && fa_cleared_by_pull) ? Fa : nullptr, RT + (size_t)(k - 2) * rt_plane, ldn,
                               ctx->ms_n_chunks ? ctx->d_ms_row_first : nullptr, P);
            std::swap(F, Fn);
        }
        if (n_planes) {
            hipStream_t hs = ctx->stream;
            if (overlap) {
                HSD_HIP(ctx, hipEventRecord(ctx->ms_ev_rings[set], ctx->stream));
                HSD_HIP(ctx, hipStreamWaitEvent(ctx->ms_stream, ctx->ms_ev_rings[set], 0));
                hs = ctx->ms_stream;
            }
            const int words_used = (nsrc + 63) / 64;
            const int parts = std::max(1, std::min(8, 2 * ctx->prop.multiProcessorCount / std::max(1, words_used * n_planes)));
            hipLaunchKernelGGL(ms_hist_kernel, dim3((unsigned)words_used, (unsigned)parts, (unsigned)n_planes), dim3(MS_HIST_THREADS), lds_hist,
                               hs, RT, ldn, rt_plane, (int)n, parts, 2, L, node_bin, B, b0, nsrc, counts);
            if (overlap) HSD_HIP(ctx, hipEventRecord(ctx->ms_ev_hist[set], ctx->ms_stream));
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "khop (bit-parallel) launch: %s", hipGetErrorString(e));
    }
    if (overlap)                         // the caller's stream continues once the last histograms are in
        for (int t = 0; t < std::min(2, batch); ++t) HSD_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ms_ev_hist[t], 0));
    return HSD_OK;
}
