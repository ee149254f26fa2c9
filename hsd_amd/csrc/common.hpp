// Internal definitions shared by the translation units of libhsd_amd.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hsd.h"
#include "hooks.hpp"

struct ncclComm;

// Operand layout of one prepared distance (built by hsd_pair_prepare):
//   bin-major matrices  X[c * npad + i]  so that a tile of 128 nodes x one column is 1 KiB
//   contiguous -- global->LDS staging and the pair kernel's LDS reads are then linear.
enum PairMode {
    PAIR_NONE = 0,
    PAIR_L1_F64 = 1,      // sum_c |X[c][i] - X[c][j]|                     (W1, raw samples)
    PAIR_SAD_U32 = 2,     // sum_k min(inv_k[i], inv_k[j]) * sum_c |U-U|   (W1, zero-pad, exact integers)
    PAIR_ZP_F64 = 3,      // same in fp64 for non-integral supports
    PAIR_KL = 4,          // H[i] - sum_c P[c][i] * LP[c][j]               (mirrored upper triangle)
    PAIR_SKL = 5,         // (H[i] + H[j] - P_i.LP_j - P_j.LP_i) / 2
    PAIR_SAD_U16 = 6,     // as PAIR_SAD_U32 with two uint16 columns per word (v_sad_u16)
};

constexpr int HSD_TILE = 128;      // pair tile edge (nodes)
constexpr int HSD_CHUNK = 16;      // columns staged per LDS pass
constexpr int HSD_MAX_LAYERS = 32;
constexpr int HSD_LCHUNK = 64;     // CSR entries per chunk of a long Laplacian row

// One layer of a zero-pad distance (csrc/pair.hip, SegDesc): its bin-form columns [bin_col0, +bin_words), and -- where
// it pays -- a second column range [rank_col0, ...) holding a copy of the bin columns below the layer's threshold
// (rank_low words) followed by the rank columns of the members above it (at most rank_cap words).  rank_col0 < 0: none.
// suf_row0 >= 0: rows [suf_row0, suf_row0 + rank_cap] of the suffix matrix hold, for word index w, the sum of the layer's rank
// values from word w on (persistent kernel, "min form": a tile pair reads only as many rank words as the SHORTER of its two
// sides needs and takes the rest of the longer side from this sum).
// r8_col0 >= 0 (packed uint16 layers, persistent kernel): a third range holds the rank-form range once more with its SMALL values as
// uint8 (four columns per word, one v_sad_u8): [first r8_b16 low-bin words | first r8_r16 rank words] as uint16 (r8_k16 words
// together), then the remaining low-bin words and the remaining rank words two-into-one as uint8 (r8_fixed words before the rank
// tail, whose length follows the tile pair like the rank words; at most r8_cap).  A tile pair may read it when no node of its two
// sides holds a value >= 256 in a column that is uint8 there (bit 31 of the per-granule maxima d_rmax16 marks granules that do).
struct HsdSegDesc {
    int bin_col0, bin_words, rank_col0, rank_low, rank_cap, suf_row0;
    int r8_col0 = -1, r8_fixed = 0, r8_k16 = 0, r8_r16 = 0, r8_cap = 0, r8_pad = 0;
};

struct hsd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // side streams: the off-diagonal rectangles of a row block run beside its diagonal square (created on first use)
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    std::string err;
    hipDeviceProp_t prop{};

    // graph (CSR, device) + host degree
    int64_t n = 0, nnz = 0;
    int64_t *d_rowptr = nullptr;
    int32_t *d_col = nullptr;
    std::vector<int32_t> h_degree;

    // k-hop scratch: one BFS queue of n ints per resident workgroup
    int32_t *d_queue = nullptr;
    int64_t queue_blocks = 0;

    // bit-parallel k-hop sweep (khop_ms.hip): plane scratch and the chunk lists of long adjacency rows
    unsigned long long *d_ms_planes = nullptr;
    size_t ms_plane_words = 0;
    int32_t *d_ms_row_first = nullptr;   // [n] first chunk of the row, or -1
    int32_t *d_ms_chunk_row = nullptr;
    int64_t *d_ms_chunk_p0 = nullptr;
    int ms_n_chunks = 0;
    bool ms_chunks_ready = false;
    hipStream_t ms_stream = nullptr;     // histogram pass of batch b beside the pulls of batch b + 1
    hipEvent_t ms_ev_rings[2] = {nullptr, nullptr}, ms_ev_hist[2] = {nullptr, nullptr};

    // signature (counts resident on device)
    int hop = -1, n_bins = 0;
    int64_t n_rows_alloc = 0;          // rows allocated (n rounded up to world * rows_per_rank)
    uint32_t *d_counts = nullptr;      // [n_rows_alloc][hop+1][n_bins]
    int32_t *d_node_bin = nullptr;     // [n]
    std::vector<int32_t> h_node_bin;   // host copy of what d_node_bin holds
    double *d_support = nullptr;       // [n_bins]
    std::vector<double> support;
    bool support_integral = false;

    // Operand arena (csrc/context.hip, hsd_pool_*): the prepared operand sets, the k-nearest sweep's permuted copies and hsd_pair_prepare's
    // temporaries come out of a GROW-ONLY pool of device blocks -- a released block stays allocated and serves the next request of about its
    // size.  Before round 5 every hsd_pair_prepare freed and re-allocated ~1 GB (hipFree synchronises the device; the first calls behind such a
    // re-allocation came out 20-25 ms late once in ~25 times): a caller that prepares one distance per scale paid that.
    struct PoolBlock { void *p; size_t bytes; bool used; };
    std::vector<PoolBlock> op_pool;

    // prepared pair operands
    int mode = PAIR_NONE;
    int64_t npad = 0;                  // node stride of the bin-major matrices (multiple of HSD_TILE)
    int dcols = 0;                     // padded column count (multiple of HSD_CHUNK)
    int dcols_alg = 0;                 // columns that carry data (algorithmic width)
    int elem_bytes = 0;
    void *d_X = nullptr;               // fp64 or u32 [dcols][npad]
    void *d_Y = nullptr;               // KL: log P, same shape
    double *d_H = nullptr;             // [2 npad]: KL: sum_c P log P per node, then the ring-0 values; W1: the ring-0 values
    HsdSegDesc seg_tab[HSD_MAX_LAYERS] = {};   // zero-pad: per layer segment, bin-form / rank-form columns
    HsdSegDesc *d_seg_tab = nullptr;
    uint32_t *d_rmax = nullptr;        // zero-pad rank form: largest ring per layer and 64-node granule [n_seg][ngran]
    int ngran = 0;
    // Tile composition of the persistent pair kernel: a tile side is EIGHT 16-node granules (16 fp64 = one 128-byte line of the
    // matrix) named by a granule list; the list of the whole graph is sorted by a rank-form key chosen by hsd_pair_prepare
    // (homogeneous tiles need fewer rank words) or is the natural order.  Lists of row / column ranges are cut out of it.
    uint32_t *d_rmax16 = nullptr;      // members above the layer's threshold, maximum per 16-node granule [n_seg][ngran16]
    int ngran16 = 0;
    std::vector<int32_t> gran_order;   // granules of the nodes [0, n - n % 128) in tile order (empty: natural order)
    uint32_t *d_suf = nullptr;         // suffix sums of the rank columns [suf_rows][npad] (row 0: zeros)
    int suf_rows = 0;
    // (up to three node ranges one after the other: the column side of a row-block launch is [0, r0) + the block's own rows + [r1, n))
    struct GranList { int64_t r[6]; int sorted; int32_t *d_gl; uint32_t *d_tmax; int ntiles; };
    std::vector<GranList> gran_lists;
    int list_cap = 0;                  // entries of the pair kernel's per-tile chunk list (LDS)
    int *d_tile_ctr = nullptr;         // persistent pair kernel: 4 sets of 8 per-XCD tile counters
    unsigned long long *d_work = nullptr;   // operand words x pairs evaluated by the last hsd_pair_rows (hsd_pair_work)
    double *d_inv = nullptr;           // zero-pad: 1/ring size per layer [n_seg][npad]
    int n_seg = 0;                     // zero-pad: number of layer segments
    int seg_valid16 = 0;               // PAIR_SAD_U16: packed data columns per segment
    bool has_point = false;            // W1 with layer 0: ring-0 values live in d_H and add |p_i - p_j|; KL with layer 0 in closed
                                       // form: they follow the entropy term (d_H + npad) and add kl_point_c where they differ
    double kl_point_c = 0.0;
    // KL kinds, sparse-column form (prepare_kl): occupancy of the columns of one half per 16-node granule [npad / 16][kl_words] (nullptr: dense form),
    // where the second half starts (symmetric KL; 0: one half), data columns of a half
    unsigned long long *d_kl_mask = nullptr;
    int kl_words = 0, kl_half = 0, kl_half_alg = 0;
    int seg_begin[HSD_MAX_LAYERS + 1] = {0};   // column range of each segment (chunk aligned)
    int *d_flag = nullptr;             // device error flag (empty ring, ...)

    // heat path: Laplacian CSR, resident ragged rings and sorted coefficient samples
    int64_t *d_lrowptr = nullptr;
    int32_t *d_lcol = nullptr;
    double *d_lval = nullptr;
    // Laplacian rows longer than 2 * HSD_LCHUNK entries are split into chunks whose partial products are
    // computed by separate workgroups (hub rows of scale-free graphs would otherwise serialise a step)
    int32_t *d_lrow_first = nullptr;   // [n] first chunk of the row, or -1
    int32_t *d_lchunk_row = nullptr;   // [n_lchunks]
    int64_t *d_lchunk_p0 = nullptr;    // [n_lchunks] first CSR entry of the chunk
    int n_lchunks = 0;
    double *d_lpartial = nullptr;      // [n_lchunks][lpartial_cols]
    int64_t lpartial_cols = 0;
    int64_t *d_hoffsets = nullptr;
    int32_t *d_hmembers = nullptr;
    double *d_hvalues = nullptr;
    std::vector<int64_t> h_hoffsets;
    int heat_hop = -1;
    int heat_max_seg = 0;

    // comm
    ncclComm *comm = nullptr;
    int world = 1, rank = 0;
    int64_t rows_per_rank = 0;

    // dynamic-LDS limit already raised for a kernel on this context's device (kernel address -> bytes)
    std::unordered_map<const void *, size_t> lds_attr;

    // cached explicit tile orders of the pair kernel (device int2 arrays)
    struct TileList { int tiles_i, tiles_j, sym, group; void *d; int count; };
    std::vector<TileList> tile_lists;

    // pruned k-nearest sweep (raw fp64 W1; csrc/pair.hip, tk_*): a node-permuted copy of the operands (similar nodes adjacent), built on the first
    // hsd_pair_topk of a prepared distance and released with the operands
    int l1_cols_per_layer = 0;         // PAIR_L1_F64: columns of one histogram layer (the layers follow one another)
    bool tk_ready = false;
    int32_t *d_tk_order = nullptr;     // [npad] the graph's node order (k-d order on the bound's features), -1 at padding positions: built once per prepared
    bool tk_order_ready = false;       //        distance and kept when the setup is rebuilt for another row block (the expensive part of the setup)
    int32_t *d_tk_orig = nullptr;      // [tk_npos] original node of a position (the graph's order, then the block's rows), -1 at padding positions
    double *d_tk_X = nullptr;          // [dcols][npad]
    double *d_tk_H = nullptr;          // [npad] ring-0 values in permuted order
    double *d_tk_lo = nullptr, *d_tk_hi = nullptr;   // [tk_nf][tk_nt] feature intervals of the 128-position tiles
    double *d_tk_scale = nullptr;      // [tk_nf][tk_nt] smallest factor of a feature's gap within a tile (zero-pad layers: 1 / ring size)
    // zero-pad kinds: the rest of the permuted operand set
    uint32_t *d_tk_U = nullptr;        // [dcols][npad] packed words
    double *d_tk_inv = nullptr;        // [n_seg][npad]
    uint32_t *d_tk_suf = nullptr;      // [suf_rows + 1][npad]
    uint32_t *d_tk_rmax16 = nullptr;   // [n_seg][npad / 16] members above the layer's threshold per 16-position granule (| bit 31: too wide for the uint8 range)
    int32_t *d_tk_gl = nullptr;        // [npad / 16] the positions' granules in natural order
    uint32_t *d_tk_tmaxs = nullptr;    // [n_seg][tk_nt] per-tile maxima of d_tk_rmax16
    int tk_nf = 0, tk_nt = 0;          // features; tiles of ALL positions (tk_npos / 128: the graph's order, then -- a row block's call -- the block's rows)
    // row block of the last setup (0, 0: the whole graph): its rows follow the graph's positions as a second run of positions, in the order the
    // graph's k-d order gives them (a subsequence of it: 128 consecutive block rows span a contiguous run of the graph's tiles)
    int64_t tk_blk0 = 0, tk_blk1 = 0, tk_npos = 0;
    int32_t *d_tk_band_lo = nullptr, *d_tk_band_hi = nullptr;    // [block tiles] first / last tile of the graph's order the first launch paired a block tile with
    void *d_tk_list = nullptr;         // int2[tk_list_cap]: a launch's surviving tile pairs
    void *d_tk_band = nullptr;         // int2[tk_band_n]: the first launch's tiles (t, t) and (t, t + 1)
    int tk_list_cap = 0, tk_band_n = 0, tk_band_w = 1;
    int *d_tk_count = nullptr;
    uint16_t *d_tk_tmax = nullptr;     // [tk_nt]

    // grow-only scratch arena (top-k candidate lists, threshold sample)
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;

    // staging for host outputs
    double *d_stage = nullptr;
    size_t stage_bytes = 0;
};

int hsd_fail(hsd_ctx *ctx, int status, const char *fmt, ...);
int hsd_guard_status(hsd_ctx *ctx, int status, const char *what) noexcept;
// internal status (never crosses the ABI): the bit-parallel k-hop sweep could not allocate its planes
constexpr int HSD_INT_NO_PLANES = -1000;

#define HSD_HIP(ctx, call)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return hsd_fail((ctx), HSD_ERR_HIP, "%s failed: %s (%s:%d)", #call,              \
                            hipGetErrorString(e_), __FILE__, __LINE__);                      \
    } while (0)

// Function-try-block handlers of every int-returning entry point: nothing leaves the C ABI as a C++ exception (a ctypes caller
// would reach std::terminate with a GPU context open).  Usage:  extern "C" int hsd_x(hsd_ctx *ctx, ...) try { ... } HSD_ABI_GUARD(ctx)
#define HSD_ABI_GUARD(ctx)                                                                   \
    catch (const std::bad_alloc &) { return hsd_guard_status((ctx), HSD_ERR_NOMEM, "out of host memory"); }      \
    catch (const std::exception &e_) { return hsd_guard_status((ctx), HSD_ERR_INTERNAL, e_.what()); }           \
    catch (...) { return hsd_guard_status((ctx), HSD_ERR_INTERNAL, "unknown C++ exception"); }

#define HSD_CHECK_CTX(ctx)                                                                   \
    do {                                                                                     \
        if (!(ctx)) return HSD_ERR_INVALID;                                                  \
        hipError_t e_ = hipSetDevice((ctx)->device);                                         \
        if (e_ != hipSuccess)                                                                \
            return hsd_fail((ctx), HSD_ERR_HIP, "hipSetDevice(%d): %s", (ctx)->device,       \
                            hipGetErrorString(e_));                                          \
    } while (0)

template <typename T>
static inline int hsd_dev_alloc(hsd_ctx *ctx, T **p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HSD_HIP(ctx, hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
    return HSD_OK;
}

template <typename T>
static inline void hsd_dev_free(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

// operand arena (see hsd_ctx::op_pool): a block of at least `bytes`; hsd_pool_free keeps the block for the next request (the caller has synchronised
// the context's stream -- nothing in flight reads it -- as it had to before a hipFree of the same block could return); a pointer that did not come
// from the pool is handed to hipFree.  hsd_pool_trim releases the unused blocks (all = true: every block; hsd_destroy).
int hsd_pool_alloc(hsd_ctx *ctx, void **p, size_t bytes);
void hsd_pool_free(hsd_ctx *ctx, void *p);
void hsd_pool_trim(hsd_ctx *ctx, bool all);

template <typename T>
static inline int hsd_op_alloc(hsd_ctx *ctx, T **p, size_t count) {
    return hsd_pool_alloc(ctx, reinterpret_cast<void **>(p), (count ? count : 1) * sizeof(T));
}
template <typename T>
static inline void hsd_op_free(hsd_ctx *ctx, T *&p) {
    if (p) hsd_pool_free(ctx, p);
    p = nullptr;
}

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// dynamic-LDS limit of a kernel, raised once per context (the attribute is per device)
static inline int hsd_raise_lds_limit(hsd_ctx *ctx, const void *kern, size_t bytes) {
    size_t &have = ctx->lds_attr[kern];
    if (bytes > have) {
        HSD_HIP(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    return HSD_OK;
}

// implemented in the per-kernel translation units
int hsd_launch_khop(hsd_ctx *ctx, int hop, int64_t r0, int64_t r1, bool hist, int32_t *d_sizes,
                    const int64_t *d_offsets, int32_t *d_members);
int hsd_launch_khop_ms(hsd_ctx *ctx, int hop, int64_t r0, int64_t r1, uint32_t *counts = nullptr, const int32_t *node_bin = nullptr, int n_bins = 0);
bool hsd_khop_ms_usable(hsd_ctx *ctx, int hop, int n_bins);
void hsd_khop_ms_release(hsd_ctx *ctx);
void hsd_release_operands(hsd_ctx *ctx);
void hsd_tk_release(hsd_ctx *ctx);            // the pruned k-nearest sweep's whole setup
void hsd_tk_release_block(hsd_ctx *ctx);      // ... all of it but the graph's node order (another row block is about to be set up)
int hsd_tk_order_device(hsd_ctx *ctx, const double *d_F, int64_t n, int nf, int32_t *d_order);      // csrc/tkorder.hip
// positions (of the order d_orig[0 .. npos)) whose node lies in [r0, r1), ascending: d_pos[0 .. r1 - r0) (csrc/tkorder.hip); synchronises
int hsd_tk_block_positions_device(hsd_ctx *ctx, const int32_t *d_orig, int64_t npos, int64_t r0, int64_t r1, int32_t *d_pos);
extern "C" void hsd_heat_release_internal(hsd_ctx *ctx);
int hsd_launch_pair_rowquirk_tiled(hsd_ctx *ctx, const double *psi_dev, int64_t ldpsi, int64_t r0, int64_t r1, double *dst, int64_t dld);
int hsd_launch_pair_ragged_tiled(hsd_ctx *ctx, int layer_lo, int layer_hi, int64_t r0, int64_t r1, double *dst, int64_t dld);
