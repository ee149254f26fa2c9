// Shared between the translation units of the all-pairs path (csrc/pair.hip: tile kernels + launch machinery + hsd_pair_rows;
// csrc/pair_prepare.hip: operand preparation, the planner, hsd_pair_prepare / hsd_pair_shard; csrc/pair_topk.hip: the fused
// k-nearest consumer): the kernels' argument block, the layer-segment descriptions, the small kernels more than one of them
// launches, and the host entry points of the launch machinery.  Round 5 split the former 4 300-line pair.hip along these lines --
// no behaviour change: the bit-identity tests (tests/test_gpu_forms.py, test_gpu_fuzz.py, test_gpu_topk.py) are the proof.
#pragma once

#include "common.hpp"

namespace hsd_pair {

#ifndef HSD_PIN
#define HSD_PIN 1
#endif
#ifndef HSD_STORE_BITS
#define HSD_STORE_BITS "nt"     // cache-policy bits of the persistent kernel's result stores (measured: see DESIGN.md section 5)
#endif
#ifndef HSD_ABLATE
#define HSD_ABLATE 0            // measurement builds only (wrong results): bit 0: no accumulator exchange; bits 2 / 3: only the direct / mirrored half is stored
#endif
#ifndef HSD_STAMPS
#define HSD_STAMPS 0                // 1: the persistent kernel carries its in-kernel cycle stamps (HSD_DEBUG bit 3) and the plain-store switch
#endif                              //    (bit 2) -- a measurement build: a dozen scalar registers the product kernel needs for itself
#ifndef HSD_STORE_SPREAD
#define HSD_STORE_SPREAD 0          // 1 (experiment, round 5): a deferred store group leaves ONE store behind each word of its 8-word trip instead of eight in front of it
#endif
#ifndef HSD_F64_UNROLL
#define HSD_F64_UNROLL 4            // columns per straight-line trip of the fp64 inner loops (2 = the plain two-column loop)
#endif
#if HSD_PIN
#define HSD_PIN_ORDER __builtin_amdgcn_sched_barrier(0)
#else
#define HSD_PIN_ORDER ((void)0)
#endif

// operand columns per LDS stage: 16 fp64 or 32 uint32 columns (the same LDS bytes); segments of
// the operand matrices are padded to this width by hsd_pair_prepare
__host__ __device__ constexpr int chunk_cols(int kind) { return (kind == 1 || kind == 4) /* uint32 kinds */ ? 32 : 16; }
constexpr int NT = 256;             // 4 waves; wave w owns quadrant (w>>1, w&1) of the tile

struct SegDesc;
struct PairArgs {
    const void *X;          // I-side operand, bin-major [dcols][npad]
    const void *Y;          // J-side operand (same as X for the L1 kernels)
    const double *H;        // DOT: per-node entropy term
    const double *inv;      // segmented: [n_seg][npad]
    const double *point;    // optional ring-0 value per node [npad] (W1: adds |p_i - p_j|; DOT: adds point_c where p_i != p_j)
    double point_c;         // DOT: KL of two different point masses on the smoothed support (closed form of layer 0, prepare_kl)
    // DOT, sparse-column form (persistent kernel): occupancy masks of one half's columns per 16-node granule [granule][kwords] -- a tile pair runs only
    // over the columns occupied on BOTH of its sides --, and the column offset of the second half (symmetric KL: the same list once more; 0: none)
    const unsigned long long *kmask;
    int kwords, khalf;
    double *out;
    int64_t ld, npad;
    int64_t rbase;          // out row 0 <-> node rbase
    int64_t i_begin, i_end, j_begin, j_end;
    int tiles_i, tiles_j, ntiles;
    const int2 *tiles;      // optional explicit tile order (super-tile major), ntiles entries
    int n_aux;              // persistent kernel: per-node vectors (1/ring size per layer, ring-0 value, entropy) staged through
                            // LDS with every tile (0: read from global memory where they are needed)
    int sym;                // square region on the diagonal: only tiles on/above it are enumerated
    int mirror;             // also write the transposed tile (the region's mirror image belongs to this call); 2: per tile, bit 30 of its row index
    int swap;               // DOT below the diagonal: value(i,j) = H[j] - X_j . Y_i
    int hmode;              // DOT: 0 -> H[i] - acc ; 1 -> (H[i] + H[j] - acc) / 2
    int dcols, n_seg;
    int dcols_valid;        // columns that carry data (non-segmented kinds)
    const struct SegDesc *seg_tab;   // segmented kinds: per layer, where its bin-form / rank-form columns live
    const uint32_t *rmax;   // [n_seg][ngran] largest ring of the layer within each 64-node granule (rank form)
    int ngran;
    int list_cap;           // entries of the per-tile chunk list in LDS
    unsigned long long *work;   // += operand words x pairs evaluated (hsd_pair_work), or nullptr
    int *tile_ctr;          // persistent kernel: [8] tiles drawn so far from each XCD's run (zeroed before the launch)
    int ctr_slot;           // which of the context's counter sets this region uses (regions of a row block run side by side)
    int phase, n_phases;    // host side: this launch covers the tile groups of phase `phase` of `n_phases` (0 / 0: all of them)
    int phase_w;            // ... of the phases [phase, phase + phase_w) (0 counts as 1): a sweep's later launches cover several
    int dbg;                // HSD_DEBUG ablation bits: 1 = no stores, 2 = no arithmetic
    // fused k-nearest consumer (hsd_pair_topk): instead of storing the tile, every value below its
    // row's threshold tau is appended to that row's candidate list
    // persistent kernel: tile composition (eight 16-node granules per tile side) and the min form of rank segments
    const int32_t *gl_i, *gl_j;     // granule lists of the region's row / column range, 8 entries per list tile
    const uint32_t *tmax_i, *tmax_j;   // [n_seg][ntl]: most members above the segment's threshold among a list tile's nodes
    int ntl_i, ntl_j;               // list tiles per side (row stride of tmax)
    const uint32_t *suf;            // suffix sums of the rank columns [rows][npad] (HsdSegDesc::suf_row0)
    const uint16_t *tau16;          // k-nearest sweep: this launch's thresholds as bfloat16 rounded UP, indexed like tau (persistent kernel)
    unsigned long long *stamps;     // HSD_DEBUG bit 3: per-workgroup lifetimes (measurement), else nullptr
    const double *tau;      // [n] per-node threshold, or nullptr for the normal matrix output
    int *cand_cnt;          // [n] candidates appended so far (may exceed cand_cap: overflow)
    int *cand_idx;          // [n][cand_cap]
    double *cand_val;       // [n][cand_cap]
    int cand_cap;
    // pruned k-nearest sweep (raw fp64 W1, hsd_pair_topk): the operands are a node-PERMUTED copy (similar nodes adjacent), the kernel
    // appends ORIGINAL node ids (orig_of[position]; < 0: a padding position, never a candidate); host side: an explicit tile list
    // instead of tile_order's, and per-tile feature intervals that let launch_region drop tile pairs no row's threshold can reach
    const int32_t *orig_of;
    const int2 *explicit_tiles;     // host side only
    int n_explicit;
    const double *prune_lo, *prune_hi;   // host side only: [prune_nf][prune_nt]
    const double *prune_scale;      // host side only: [prune_nf][prune_nt] smallest factor of the feature's gap within a tile (nullptr: 1)
    int prune_nf, prune_nt, prune_drop_band;
    // ... of a ROW BLOCK (hsd_pair_topk(k, r0, r1), a rank's share): the block's rows are a second run of positions behind the graph's
    // (rows [i_begin, i_end) x columns [j_begin, j_end), not symmetric, not mirrored); a row's own node also sits among the columns, so
    // "not itself" is decided on the ORIGINAL ids (kernel: self_by_orig); the first launch paired row tile t with the column tiles
    // [prune_band_lo[t], prune_band_hi[t]], which the filter of the later launches drops (host side)
    int self_by_orig;
    // pruned sweep: the number of tiles of THIS launch as the tile filter left it in device memory (k-nearest kernel only; nullptr: `ntiles` holds it).
    // The host then need not wait for the count between two launches of a sweep (six blocking 4-byte copies per call before round 5).
    const int *ntiles_dev;
    const int32_t *prune_band_lo, *prune_band_hi;
    const int32_t *gl_ov;           // host side only: the region's granule list and its per-tile maxima given by the caller (permuted operands)
    const uint32_t *tmax_ov;
    int ntl_ov;
};

enum { K_L1 = 0, K_SAD = 1, K_ZP = 2, K_DOT = 3, K_SAD16 = 4 };
constexpr int TILE_MIRROR_FLAG = 1 << 30, TILE_INDEX_MASK = TILE_MIRROR_FLAG - 1;   // explicit tile lists of row-block launches (PairArgs::mirror == 2)

template <int KIND> struct Elem { using type = double; };
template <> struct Elem<K_SAD> { using type = uint32_t; };
template <> struct Elem<K_SAD16> { using type = uint32_t; };      // two uint16 columns per word
template <typename T> struct Vec2;
template <> struct Vec2<double> { using type = double2; };
template <> struct Vec2<uint32_t> { using type = uint2; };

// pointers that arrive inside the argument struct lose their address space; typed global
// pointers make the compiler emit global_store / global_load instead of flat_* (flat accesses
// also tick lgkmcnt and would serialise against the LDS operand reads)
typedef __attribute__((address_space(1))) double gdouble;
typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) v2f64 gdouble2;
typedef __attribute__((address_space(1))) const double gcdouble;
typedef __attribute__((address_space(1))) const uint32_t gcu32;

// One layer ("segment") of a zero-pad distance has exact representations of the same integer
//   S(i, j) = |n_i - n_j| s_0 + sum_b |T_i(b) - T_j(b)| (s_{b+1} - s_b)  =  sum_r |u_i(r) - u_j(r)|
// (area between the two tail-count step functions T(x) = #members > x  =  L1 distance of the descending sorted samples,
// zero padded), and the area may be cut at any value theta and each part taken in either form:
//   bin form   one column per support bin (tail count x bin width), two uint16 columns per word
//   rank form  one column per rank r: (r-th largest member value - theta)+, zeros beyond the members above theta
//              -> ceil(R / 2) words, R = most members above theta among the nodes of the two tiles
// A layer is stored as its bin columns (all of them, contiguous) and, in a second range, a copy of the bin columns below
// theta followed by the rank columns above theta; a tile pair reads whichever range is shorter -- one contiguous run of
// columns either way.  theta = 0 is the pure rank form (ring 1 of a scale-free graph:
// ~10 members against 241 degree bins); ring 2 (hundreds of members, a few dozen of them hubs) keeps the dense low bins
// and takes its sparse tail by rank.  Same integer either way, so the result is bit-identical.
struct SegDesc : HsdSegDesc {};    // columns in words

__host__ __device__ constexpr int stream_chunk_cols(int kind) { return chunk_cols(kind); }

// Lane l < n_seg holds segment l.  RI / RJ: most members above the segment's threshold among the nodes of the tile's two
// sides.  Chooses bin form or (min form) rank form, appends the segment's chunks to the tile's list like plan_segment and
// leaves in smeta[l] the suffix row to add and the side it belongs to (bit 30: the J side), or -1.
// The form a tile pair reads a layer segment in (shared by the kernel's planner and by the host's cost estimate of a row partition):
// first column, words, suffix row | side << 30 (or -1), leading uint16 words.
struct SegChoice { int col0, words, sm, k16; };
template <bool PACK>
__host__ __device__ inline SegChoice choose_segment_form(const HsdSegDesc &sd, bool active, uint32_t RI, uint32_t RJ) {
    SegChoice c{sd.bin_col0, active ? sd.bin_words : 0, -1, 1 << 20};
    // (bit 31 of a side's maximum: one of its nodes holds a value >= 256 where the uint8 range has bytes)
    const bool wide = ((RI | RJ) >> 31) != 0;
    RI &= 0x7fffffffu; RJ &= 0x7fffffffu;
    if (active && sd.rank_col0 >= 0 && sd.suf_row0 >= 0) {
        const uint32_t Rm = RI < RJ ? RI : RJ;
        int wr = PACK ? (int)((Rm + 1) >> 1) : (int)Rm;
        if (sd.rank_low + wr == 0 && sd.rank_cap >= 1) wr = 1;      // a layer always closes with a fold (the longer side's suffix is added there)
        if (wr <= sd.rank_cap) {
            c.col0 = sd.rank_col0;
            c.words = sd.rank_low + wr;
            c.sm = (sd.suf_row0 + wr) | (RI >= RJ ? 0 : (1 << 30));
            if (PACK && sd.r8_col0 >= 0 && !wide) {
                // uint8 range: the rank words beyond the first r8_r16 two-into-one.  The columns end with the uint16 range's
                // (zeros beyond its cap), so the suffix row is capped there too.
                const int over = wr - sd.r8_r16, wr8 = over > 0 ? (over + 1) >> 1 : 0, w8 = sd.r8_fixed + wr8;
                if (wr8 <= sd.r8_cap && w8 >= 1 && w8 < c.words) {
                    const int srow = sd.r8_r16 + 2 * wr8;
                    c.col0 = sd.r8_col0;
                    c.words = w8;
                    c.k16 = sd.r8_k16;
                    c.sm = (sd.suf_row0 + (srow < sd.rank_cap ? srow : sd.rank_cap)) | (RI >= RJ ? 0 : (1 << 30));
                }
            }
        }
    }
    return c;
}


// device temporary of hsd_pair_prepare: released on every exit path
struct DevTmp {
    void *p = nullptr;
    hsd_ctx *ctx = nullptr;
    // (back to the operand arena, not to the driver; the stream is drained first: what hipFree's device-wide synchronisation used to guarantee)
    ~DevTmp() { if (p && ctx) { (void)hipStreamSynchronize(ctx->stream); hsd_pool_free(ctx, p); } else if (p) (void)hipFree(p); }
    template <typename T> T *as() const { return static_cast<T *>(p); }
    int alloc(hsd_ctx *c, size_t bytes) { ctx = c; return hsd_pool_alloc(c, &p, bytes ? bytes : 1); }
    void *release() { void *q = p; p = nullptr; return q; }
};

#define HSD_TMP_ALLOC(ctx, tmp, bytes) do { const int rc_tmp_ = (tmp).alloc((ctx), (bytes)); if (rc_tmp_ != HSD_OK) return rc_tmp_; } while (0)

// ---- host entry points of the launch machinery (csrc/pair.hip) used by the k-nearest consumer and the planner ----
int tile_edge(const hsd_ctx *ctx);
int stream_waves(const hsd_ctx *ctx, const PairArgs &a);
int launch_region(hsd_ctx *ctx, PairArgs a, int BT, hipStream_t stream, bool allow_stream = true);

}  // namespace hsd_pair

// ---- small kernels launched from more than one translation unit (one copy per unit: no relocatable device code) ----
namespace {

// uint8 range of a packed layer (HsdSegDesc::r8_*): what node i needs to stay uint16 in the layer's rank-form range
//   need[(2 s) * npad + i]     = leading low-bin words up to and including the last one that holds a value >= 256
//   need[(2 s + 1) * npad + i] = leading rank words that hold a value >= 256 (the ranks are descending)
__global__ void u8_need_kernel(const uint32_t *__restrict__ X, int64_t npad, int64_t n, const HsdSegDesc d, uint32_t *__restrict__ need) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = 0;
    if (i >= n || d.rank_col0 < 0) return;
    const auto wide = [](uint32_t v) { return (v & 0xffffu) >= 256u || (v >> 16) >= 256u; };
    uint32_t b8 = 0, r8 = 0;
    for (int w = 0; w < d.rank_low; ++w)
        if (wide(X[(int64_t)(d.rank_col0 + w) * npad + i])) b8 = (uint32_t)w + 1u;
    for (int w = 0; w < d.rank_cap; ++w) {
        if (!wide(X[(int64_t)(d.rank_col0 + d.rank_low + w) * npad + i])) break;
        r8 = (uint32_t)w + 1u;
    }
    need[(int64_t)(2 * s) * npad + i] = b8;
    need[(int64_t)(2 * s + 1) * npad + i] = r8;
}

// g16[k][g] = largest of vec[k][16 g .. 16 g + 15] (k: layer x candidate threshold)
__global__ void granule16_max_kernel(const uint32_t *__restrict__ vec, int64_t npad, int ngran16, uint32_t *__restrict__ g16) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (g >= ngran16) return;
    const uint32_t *v = vec + (int64_t)k * npad + (int64_t)g * 16;
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) m = max(m, v[e]);
    g16[(int64_t)k * ngran16 + g] = m;
}

// tmax[s * ntiles + t] = most members above layer s's threshold among the nodes of list tile t (its eight granules)
__global__ void list_tile_max_kernel(const int32_t *__restrict__ gl, int ntiles, const uint32_t *__restrict__ rmax16, int ngran16, int n_seg,
                                     uint32_t *__restrict__ tmax) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (int64_t)ntiles * n_seg) return;
    const int s = (int)(k / ntiles), t = (int)(k % ntiles);
    uint32_t m = 0, wide = 0;                        // bit 31: the granule holds a node too wide for the layer's uint8 range
    for (int e = 0; e < 8; ++e) {
        const uint32_t v = rmax16[(int64_t)s * ngran16 + gl[(int64_t)8 * t + e]];
        m = max(m, v & 0x7fffffffu);
        wide |= v & 0x80000000u;
    }
    tmax[k] = m | wide;
}

__global__ void fill_f64_kernel(double *__restrict__ p, int64_t n, double v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace
