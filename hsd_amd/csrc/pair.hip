// K2 (finalise) + K3: operand preparation and the tiled all-pairs distance kernels.
//
// Replaces the N(N-1)/2 Python iterations of model/HSD.py:98-114 / :140-161 and the per-pair
// numpy work inside scipy's _cdf_distance and tools/metrics.py:18-36,74-88.
//
// Every distance on this path is a sum over columns c of f(X[c][i], Y[c][j]):
//   W1 on raw samples (scipy)      sum_c | G[c][i] - G[c][j] |,  G = CDF * bin width      (fp64)
//   W1 after zero-pad alignment    sum_k min(1/n_ik, 1/n_jk) * sum_c | U[c][i] - U[c][j] |: EXACT unsigned integers,
//                                  two uint16 columns per v_sad_u16 (four uint8 per v_sad_u8 where a layer's values are
//                                  small); a layer has several exact forms -- bins, ranks above a threshold, bytes -- and a
//                                  tile pair reads the shortest (SegDesc, plan_segment*, prepare_w1_zeropad)
//   KL / symmetric KL              H[i] - sum_c P[c][i] * logP[c][j]                       (fp64 FMA)
// so one tiling skeleton serves all of them.  Four waves own a 128 x 128 (or 64 x 64) tile of node pairs; both operand
// blocks are staged CH columns at a time in LDS by direct-to-LDS loads (bin-major storage makes a tile-wide slice of one
// column contiguous), double buffered; each lane accumulates an 8 x 8 (4 x 4) block of pairs in registers.  Only tiles on or
// above the diagonal of a symmetric region are computed; the mirrored tile is written from the same registers.
//
// Two kernels: pair_stream_kernel -- persistent workgroups (two per CU) that walk granule-composed 128 x 128 tiles drawn from
// per-XCD counters, plan the next tile while computing this one and let a finished tile's stores leave under the next tile's
// arithmetic: every region with >= 11 (integer kinds) / 16 (fp64) tiles per resident workgroup -- and pair_kernel, one tile per workgroup: small regions
// (64 x 64 tiles, ~5 workgroups per CU), strips at the end of a range, KL's diagonal tiles.  Both produce the same bits per
// pair.  DESIGN.md section 5 holds the measurements behind each choice.
#include "common.hpp"

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

namespace {

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

__device__ inline int64_t uniform64(int64_t v) {
    // a wave-uniform value that the compiler may have left in vector registers: back to scalar registers
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// position t of the tile order -> tile coordinates
template <bool SCALAR_LOAD = false>
__device__ inline void tile_from_index(const PairArgs &a, int t, int &ti, int &tj) {
    if (a.tiles) {
        // super-tile-major order built by the host: the 2S operand blocks of an S x S group of
        // tiles are fetched from HBM once and then served by the XCD's L2
        // (the coordinates are wave-uniform: keep them in scalar registers, and everything derived from them with them)
        if constexpr (SCALAR_LOAD) {
            // The persistent kernel asks for a tile's coordinates between two tiles, with result stores and operand requests in
            // flight: as a vector load (what hipcc emits next to inline asm with memory clobbers) the answer queues behind all of
            // them and every wave waits for it with vmcnt(0).  The scalar data cache has its own counter and a few hundred cycles.
            unsigned long long tt;
            const int2 *p = a.tiles + __builtin_amdgcn_readfirstlane(t);
            asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) : "s"(p) : "memory");
            ti = (int)(unsigned)tt;
            tj = (int)(unsigned)(tt >> 32);
            return;
        }
        const int2 tt = a.tiles[__builtin_amdgcn_readfirstlane(t)];
        ti = __builtin_amdgcn_readfirstlane(tt.x);
        tj = __builtin_amdgcn_readfirstlane(tt.y);
        return;
    }
    if (a.sym) {
        const int T = a.tiles_i;
        // row-major over the upper triangle: row ti starts at ti*T - ti*(ti-1)/2
        int row = (int)((2.0 * T + 1.0 - sqrt((2.0 * T + 1.0) * (2.0 * T + 1.0) - 8.0 * (double)t)) * 0.5);
        row = max(0, min(T - 1, row));
        while (row > 0 && (int64_t)row * T - (int64_t)row * (row - 1) / 2 > t) --row;
        while ((int64_t)(row + 1) * T - (int64_t)(row + 1) * row / 2 <= t) ++row;
        ti = __builtin_amdgcn_readfirstlane(row);
        tj = __builtin_amdgcn_readfirstlane(row + (t - (int)((int64_t)row * T - (int64_t)row * (row - 1) / 2)));
    } else {
        ti = t / a.tiles_j;
        tj = t % a.tiles_j;
    }
}

__device__ inline void decode_tile(const PairArgs &a, int b, int &ti, int &tj) {
    // XCD-aware order: workgroups b and b+8 share an XCD (round-robin dispatch), so give each
    // XCD a contiguous run of tiles -- consecutive tiles share their I block in that XCD's L2.
    const int nb = a.ntiles;
    const int q = nb >> 3, r = nb & 7, x = b & 7, idx = b >> 3;
    tile_from_index(a, (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx, ti, tj);
}

__device__ inline void glds16(const void *g, void *lds_wave_base) {
    // asynchronous global -> LDS copy, 16 B per lane, LDS address = wave-uniform base + 16 * lane
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// The same with the instruction form spelled out: scalar 64-bit base + one 32-bit per-lane offset, LDS address in M0.
// (Inside the persistent kernel hipcc re-associates base + offset and keeps one 64-bit VECTOR address per instruction
// live across the tile loop -- 16 register pairs, spilled to scratch and reloaded behind s_waitcnt vmcnt(0).)
__device__ inline void glds16_s(uint64_t sbase, uint32_t lane_off, uint32_t lds_addr) {
    // (M0 is a reserved register: hipcc does not accept it as a clobber.  Nothing else in pair_stream_kernel uses M0 -- DS
    // instructions do not on gfx9+, and the kernel contains no LDS-DMA builtin, s_movrel or message instruction.)
    // (s_nop 3, not 0: together with the s_mov that is five wait states between a vector-ALU write of the base -- hipcc reloads
    // spilled scalars with v_readlane right in front of this -- and its use as an address: a hazard hipcc pads only for
    // instructions it emitted itself; csrc/check_m0.py verifies the distance in the disassembly)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(lane_off), "s"(sbase) : "memory");
}

// A 16-byte store per lane with a scalar 64-bit base and a 32-bit per-lane offset, non-temporal or plain.  (Written out:
// the result stores of the persistent kernel then cost no vector address arithmetic.  hipcc does not count them in its own
// vmcnt bookkeeping, like the loads above: whatever it waits for with a counted vmcnt is older than these by program order.)
// The s_nop is part of the contract: a store of more than 64 bits followed within two wait states (gfx940+) by a vector
// write to its data registers is a hazard the hardware does not interlock.  hipcc pads its own stores; it cannot see into
// this one, and it does re-use the four registers it copies a non-adjacent accumulator pair into for the next store
// (measured without the pad: one mirrored entry in ~10^5 carried its neighbour's value).
// The caller owes it a second pad: the base comes straight out of v_readfirstlane (uniform64), and a scalar register written
// by the vector ALU must not be read as a memory address within five wait states -- again a hazard hipcc pads for its own
// instructions only (without it: stores through a stale base, a memory access fault).  store_group pads once per group of
// eight stores that share a base.  csrc/check_m0.py verifies both pads in the disassembly.
template <bool NT>
__device__ inline void gstore16_s(uint64_t sbase, uint32_t lane_off, v2f64 val) {
    if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, %2 " HSD_STORE_BITS "\n\ts_nop 1" ::"v"(lane_off), "v"(val), "s"(sbase) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(lane_off), "v"(val), "s"(sbase) : "memory");
}

// Stage CH columns x BT nodes of one operand: LDS image [CH][BT] of T (linear, as in memory).
// FAST: direct-to-LDS loads (needs 16-B aligned column slices); else through registers.
template <typename T, bool FAST, int BT, int CH>
__device__ inline void stage_side(const T *__restrict__ Xg, int64_t npad, int c0, int64_t n0, T *__restrict__ dst, int tid) {
    if constexpr (FAST) {
        constexpr int LANES_PER_COL = BT * (int)sizeof(T) / 16;      // lanes that cover one column slice
        constexpr int COLS_PER_INSTR = 64 / LANES_PER_COL;
        constexpr int INSTRS = CH / COLS_PER_INSTR;                  // wave instructions per side
        static_assert(INSTRS % (NT / 64) == 0, "stage instructions must divide over the waves");
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
        const int lc = l / LANES_PER_COL, ll = l % LANES_PER_COL;
        // wave-uniform 64-bit base (SGPRs) + one 32-bit per-lane byte offset (1 VGPR)
        const uint32_t lane_off = (uint32_t)lc * (uint32_t)(npad * sizeof(T)) + 16u * ll;
#pragma unroll
        for (int k = 0; k < INSTRS / (NT / 64); ++k) {
            const int ins = k * (NT / 64) + w;
            const char *ub = reinterpret_cast<const char *>(Xg + (int64_t)(c0 + ins * COLS_PER_INSTR) * npad + n0);
            glds16(ub + lane_off, dst + (size_t)ins * COLS_PER_INSTR * BT);
        }
    } else {
#pragma unroll
        for (int p = 0; p < CH * BT / NT; ++p) {
            const int e = p * NT + tid;
            const int c = e / BT, rr = e % BT;
            dst[c * BT + rr] = Xg[(int64_t)(c0 + c) * npad + n0 + rr];
        }
    }
}

template <bool VEC>
__device__ inline void store2(gdouble *p, double v0, double v1, bool ok0, bool ok1) {
    if (VEC && ok0 && ok1) {
        *reinterpret_cast<gdouble2 *>(p) = v2f64{v0, v1};
    } else {
        if (ok0) p[0] = v0;
        if (ok1) p[1] = v1;
    }
}

template <bool VEC>
__device__ inline void store2_full(gdouble *p, double v0, double v1, bool nt) {
    if (VEC) {
        // the matrix is written once and never re-read by this kernel: the non-temporal hint keeps
        // the 8 N^2 bytes of results from displacing the operand slices in L2 (measured +5 %)
        if (nt) __builtin_nontemporal_store(v2f64{v0, v1}, reinterpret_cast<gdouble2 *>(p));
        else *reinterpret_cast<gdouble2 *>(p) = v2f64{v0, v1};
    } else {
        p[0] = v0;
        p[1] = v1;
    }
}

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

// The lanes append their segments' chunks {first column, valid | ends_segment << 8 | segment << 16} to the tile's list, the
// LAST segment first (outer rings are the long ones: the persistent kernel lets the previous tile's result stores leave
// beside the first segment it computes, and every kernel folds the layers in list order).  Returns in every lane the list
// length.
// k16: the first k16 words of the run are uint16 pairs, the rest uint8 quadruples (bits 9-14 of a chunk's second entry: how many
// of ITS words are uint16; every word where the layer has no uint8 range).
template <int CH>
__device__ inline int emit_chunks(int nch, int col0, int words, int l, int2 *lst, int k16 = 1 << 20) {
    int suf = nch;                                   // inclusive suffix sum over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int dn = __shfl_down(suf, o, 64);
        if (l + o < 64) suf += dn;
    }
    int2 *dst = lst + (suf - nch);
    for (int c = 0; c < nch; ++c) {
        const int valid = min(CH, words - c * CH), n16 = max(0, min(valid, k16 - c * CH));
        dst[c] = make_int2(col0 + c * CH, valid | ((c == nch - 1) ? 0x100 : 0) | (n16 << 9) | (l << 16));
    }
    return __shfl(suf, 0, 64);
}

// Lane l < n_seg holds segment l: choose the form of its upper part from R (members above theta, worst node of the tile
// pair), then the lanes append their chunks {first column, valid | ends_segment << 8 | segment << 16} to the tile's list.
// Returns the list length (every lane).
template <int CH, bool PACK>
__device__ inline int plan_segment(const SegDesc &sd, bool active, uint32_t R, int l, int2 *lst) {
    int col0 = sd.bin_col0, words = active ? sd.bin_words : 0;
    if (active && sd.rank_col0 >= 0) {
        const int wr = PACK ? (int)((R + 1) >> 1) : (int)R;
        if (wr <= sd.rank_cap) { col0 = sd.rank_col0; words = sd.rank_low + wr; }     // low bins (copy) + ranks: one contiguous range
    }
    const int nch = (words + CH - 1) / CH;
    return emit_chunks<CH>(nch, col0, words, l, lst);
}

// The pair kernels' lane map: lane bits (x0 y0 x1 y1 x2 y2) -> (ly, lx).  Any bijection serves the arithmetic (the LDS operand
// reads broadcast over the other coordinate either way); the result stores decide.  A store instruction holds its wave for as long as
// its data takes to leave the SIMD, and that grows with the number of 64-byte segments its lane QUADS touch (tools_dev/
// ubench_store_map.hip, cycles per instruction and wave: quad in one segment 102, in two 129, in four 257).  With lx in the low lane
// bits the direct half is at 102 and the mirrored half (there lx picks the row) at 257; interleaved, both halves are at 129.
// Persistent kernel: 15.15 -> 14.74 ms at BA N = 100k (measured at no gain earlier in the round, when the lane exchange and the
// store bursts still hid it); the one-tile kernel takes the same map (ER N = 10k is bound by its stores).
__device__ inline int lane_y(int l) { return ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4); }
__device__ inline int lane_x(int l) { return (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4); }

// A workgroup computes exactly one tile.
template <int KIND, bool FAST, int TR>
__global__ __launch_bounds__(NT, TR == 8 ? 2 : 4) void pair_kernel(const PairArgs a) {
    using T = typename Elem<KIND>::type;
    using T2 = typename Vec2<T>::type;
    constexpr int TC = TR;
    constexpr int WQ = 8 * TR;                       // quadrant edge owned by one wave
    constexpr int BT = 2 * WQ;                       // tile edge: 64 (TR = 4) or 128 (TR = 8)
    constexpr bool SEG = (KIND == K_SAD || KIND == K_ZP || KIND == K_SAD16);
    constexpr int CH = chunk_cols(KIND);             // operand columns per LDS stage
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    T *sbase = reinterpret_cast<T *>(lds_raw);       // stage s: I at s*2*CH*BT, J right behind it
    // SEG kinds: the tile's chunk list {first column, valid | ends_segment << 8 | segment << 16} behind the stages
    int2 *clist = reinterpret_cast<int2 *>(sbase + 4 * CH * BT);
    int *nlist_p = reinterpret_cast<int *>(clist + a.list_cap);

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int ly = lane_y(l), lx = lane_x(l);
    // Lane (ly, lx) of wave w owns rows  wi + 16 (r>>1) + (r&1)  and columns  wj + 16 (c>>1) + (c&1),
    // r, c < TR: adjacent pairs interleaved over the 8 lanes of a group.  With this map a store
    // instruction -- direct (fixed r, column pair) or mirrored (fixed c, row pair) -- has each
    // group of 8 lanes write one full 128-byte line, and every LDS operand read is a
    // conflict-free 16-byte access (8 lanes contiguous, broadcast across the other index).
    const int wi = (w >> 1) * WQ + 2 * ly;
    const int wj = (w & 1) * WQ + 2 * lx;
    const T *__restrict__ XI = reinterpret_cast<const T *>(a.swap ? a.Y : a.X);
    const T *__restrict__ XJ = reinterpret_cast<const T *>(a.swap ? a.X : a.Y);
    gdouble *const out = (gdouble *)a.out;
    gcdouble *const gH = (gcdouble *)a.H;
    gcdouble *const ginv = (gcdouble *)a.inv;
    gcdouble *const gpoint = (gcdouble *)a.point;
    const bool nt = !(a.dbg & 4);                    // HSD_DEBUG bit 2: plain instead of non-temporal stores

    const int t = blockIdx.x;
    if (t >= a.ntiles) return;
    int ti, tj;
    decode_tile(a, t, ti, tj);
    const int64_t i0 = a.i_begin + (int64_t)ti * BT, j0 = a.j_begin + (int64_t)tj * BT;

    int nchunks = a.dcols / CH;
    if constexpr (SEG) {
        if (w == 0) {
            // lane s plans segment s (rmax: per 64-node granule maxima of the members above the segment's threshold)
            SegDesc sd{};
            uint32_t R = 0;
            const bool active = l < a.n_seg;
            if (active) {
                sd = a.seg_tab[l];
                if (sd.rank_col0 >= 0) {
                    const uint32_t *rm = a.rmax + (int64_t)l * a.ngran;
                    for (int64_t g = i0 >> 6; g <= (i0 + BT - 1) >> 6; ++g) R = max(R, rm[g]);
                    for (int64_t g = j0 >> 6; g <= (j0 + BT - 1) >> 6; ++g) R = max(R, rm[g]);
                }
            }
            const int pre = plan_segment<CH, KIND == K_SAD16>(sd, active, R, l, clist);
            if (l == 0) *nlist_p = pre;
        }
        __syncthreads();
        nchunks = __builtin_amdgcn_readfirstlane(*nlist_p);
    }
    auto chunk_col = [&](int k) __attribute__((always_inline)) -> int {
        if constexpr (SEG) return __builtin_amdgcn_readfirstlane(clist[k].x);
        else return k * CH;
    };

    // the first two operand chunks: both LDS stages are free at this point
    if (nchunks > 0) {
        const int c0 = chunk_col(0);
        stage_side<T, FAST, BT, CH>(XI, a.npad, c0, i0, sbase, tid);
        stage_side<T, FAST, BT, CH>(XJ, a.npad, c0, j0, sbase + CH * BT, tid);
        if constexpr (FAST) {
            if (nchunks > 1) {
                const int c1 = chunk_col(1);
                stage_side<T, true, BT, CH>(XI, a.npad, c1, i0, sbase + 2 * CH * BT, tid);
                stage_side<T, true, BT, CH>(XJ, a.npad, c1, j0, sbase + 3 * CH * BT, tid);
            }
        }
    }

    double acc[TR][TC];
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int c = 0; c < TC; ++c) acc[r][c] = 0.0;
    if (SEG && gpoint) {
        // layer kinds: the accumulators start at the ring-0 term (the persistent kernel's rule: same bits)
        double pj[TC];
#pragma unroll
        for (int cc = 0; cc < TC; ++cc) pj[cc] = gpoint[j0 + wj + 16 * (cc >> 1) + (cc & 1)];
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            const double pi = gpoint[i0 + wi + 16 * (r >> 1) + (r & 1)];
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) acc[r][cc] = fabs(pi - pj[cc]);
        }
    }
    T part[TR][TC];
    if constexpr (SEG) {
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
            for (int c = 0; c < TC; ++c) part[r][c] = (T)0;
    }
    unsigned long long work = 0;                     // operand words x pairs this tile evaluates (hsd_pair_work)
    for (int k = 0; k < nchunks; ++k) {
        // columns of this chunk that carry data (the tail of a segment is zero padding)
        int cvalid, seg = 0;
        bool ends_seg = false;
        if constexpr (SEG) {
            const int meta = __builtin_amdgcn_readfirstlane(clist[k].y);
            cvalid = meta & 0xff;
            ends_seg = (meta & 0x100) != 0;
            seg = meta >> 16;
        } else {
            cvalid = min(CH, a.dcols_valid - k * CH);
        }
        work += (unsigned)cvalid;
        if (a.dbg & 2) cvalid = 0;
        if constexpr (FAST) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            __syncthreads();
        }
        const T *sI = sbase + (size_t)(FAST ? (k & 1) : 0) * 2 * CH * BT;
        const T *sJ = sI + CH * BT;
        if constexpr (FAST) {
            if (k > 0 && k + 1 < nchunks) {
                T *nI = sbase + (size_t)((k + 1) & 1) * 2 * CH * BT;
                const int cn = chunk_col(k + 1);
                stage_side<T, true, BT, CH>(XI, a.npad, cn, i0, nI, tid);
                stage_side<T, true, BT, CH>(XJ, a.npad, cn, j0, nI + CH * BT, tid);
            }
        }
        // operands are double-buffered in registers: the LDS reads of column c+1 are in flight
        // while column c is being accumulated
        T av[2][TR], bv[2][TC];
        auto load_ops = [&](int c, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < TR / 2; ++q) {
                const T2 v = *reinterpret_cast<const T2 *>(sI + c * BT + wi + 16 * q);
                av[buf][2 * q] = v.x;
                av[buf][2 * q + 1] = v.y;
            }
#pragma unroll
            for (int q = 0; q < TC / 2; ++q) {
                const T2 v = *reinterpret_cast<const T2 *>(sJ + c * BT + wj + 16 * q);
                bv[buf][2 * q] = v.x;
                bv[buf][2 * q + 1] = v.y;
            }
        };
        auto compute = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < TR; ++r)
#pragma unroll
                for (int cc = 0; cc < TC; ++cc) {
                    if constexpr (KIND == K_L1) {
                        acc[r][cc] += fabs(av[buf][r] - bv[buf][cc]);
                    } else if constexpr (KIND == K_SAD) {
                        part[r][cc] = __usad(av[buf][r], bv[buf][cc], part[r][cc]);
                    } else if constexpr (KIND == K_SAD16) {
                        // |a.lo - b.lo| + |a.hi - b.hi| + acc: two columns per instruction
                        part[r][cc] = __builtin_amdgcn_sad_u16(av[buf][r], bv[buf][cc], part[r][cc]);
                    } else if constexpr (KIND == K_ZP) {
                        part[r][cc] += fabs(av[buf][r] - bv[buf][cc]);
                    } else {
                        acc[r][cc] = fma(av[buf][r], bv[buf][cc], acc[r][cc]);
                    }
                }
        };
        // a chunk that closes a layer needs the rows' / columns' 1/ring-size: request them now, they arrive
        // under the column loop instead of stalling the fold below
        // (4 x 4 micro-tiles only: the 8 x 8 ones have no registers to spare and load them at the fold)
        constexpr bool EARLY_INV = (KIND == K_SAD || KIND == K_SAD16) && TR == 4;
        double vi_[TR], vj_[TC];
        if constexpr (SEG) {
            if (EARLY_INV && ends_seg) {
                gcdouble *invs = ginv + (int64_t)seg * a.npad;
#pragma unroll
                for (int cc = 0; cc < TC; ++cc) vj_[cc] = invs[j0 + wj + 16 * (cc >> 1) + (cc & 1)];
#pragma unroll
                for (int r = 0; r < TR; ++r) vi_[r] = invs[i0 + wi + 16 * (r >> 1) + (r & 1)];
            }
        }
        load_ops(0, 0);
        int c = 0;
        // Eight columns per trip, straight-line: inside the body hipcc's wait-count pass knows how many LDS reads were
        // issued after the ones a column needs and emits counted waits (lgkmcnt(n)); at a loop head it merges the
        // back-edge state and waits for everything -- with two columns per trip that drained the read-ahead every 128
        // instructions (WAIT_ANY 29 % of the wave cycles, profiles/r02a_target).
        // the scheduling barriers pin "request column c+1, THEN accumulate column c": without
        // them hipcc sinks the LDS reads below the arithmetic and waits for them at once
        constexpr int UNR = sizeof(T) == 4 ? 8 : HSD_F64_UNROLL;      // (the fp64 kinds have few registers to spare)
#pragma unroll 1
        for (; UNR > 2 && c + UNR <= cvalid; c += UNR) {
#pragma unroll
            for (int u = 0; u < UNR; u += 2) {
                load_ops(min(c + u + 1, CH - 1), 1);
                HSD_PIN_ORDER;
                compute(0);
                HSD_PIN_ORDER;
                load_ops(min(c + u + 2, CH - 1), 0);
                HSD_PIN_ORDER;
                compute(1);
                HSD_PIN_ORDER;
            }
        }
#pragma unroll 1
        for (; c < cvalid; c += 2) {
            load_ops(min(c + 1, CH - 1), 1);
            HSD_PIN_ORDER;
            compute(0);
            HSD_PIN_ORDER;
            if (c + 1 < cvalid) {
                load_ops(min(c + 2, CH - 1), 0);
                HSD_PIN_ORDER;
                compute(1);
                HSD_PIN_ORDER;
            }
        }
        if constexpr (SEG) {
            if (ends_seg) {
                // sum_c |..| / max(n_i, n_j) == sum * min(1/n_i, 1/n_j); empty rings carry inv = 2
                if constexpr (!EARLY_INV) {
                    gcdouble *invs = ginv + (int64_t)seg * a.npad;
#pragma unroll
                    for (int cc = 0; cc < TC; ++cc) vj_[cc] = invs[j0 + wj + 16 * (cc >> 1) + (cc & 1)];
#pragma unroll
                    for (int r = 0; r < TR; ++r) vi_[r] = invs[i0 + wi + 16 * (r >> 1) + (r & 1)];
                }
#pragma unroll
                for (int r = 0; r < TR; ++r) {
                    const double vi = vi_[r];
#pragma unroll
                    for (int cc = 0; cc < TC; ++cc) {
                        acc[r][cc] = fma((double)part[r][cc], fmin(vi, vj_[cc]), acc[r][cc]);
                        part[r][cc] = (T)0;
                    }
                }
            }
        }
        if constexpr (!FAST) {
            __syncthreads();
            if (k + 1 < nchunks) {
                const int cn = chunk_col(k + 1);
                stage_side<T, false, BT, CH>(XI, a.npad, cn, i0, sbase, tid);
                stage_side<T, false, BT, CH>(XJ, a.npad, cn, j0, sbase + CH * BT, tid);
            }
        }
    }
    if (a.work && tid == 0) atomicAdd(a.work, work * (unsigned long long)(BT * BT));

    const bool diag = a.sym && ti == tj;

    // ---- epilogue arithmetic ----
    if constexpr (KIND == K_DOT) {
        double hj[TC];
#pragma unroll
        for (int cc = 0; cc < TC; ++cc) hj[cc] = gH[j0 + wj + 16 * (cc >> 1) + (cc & 1)];
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            const double hi = gH[i0 + wi + 16 * (r >> 1) + (r & 1)];
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) {
                if (a.hmode == 0) acc[r][cc] = (a.swap ? hj[cc] : hi) - acc[r][cc];
                else acc[r][cc] = 0.5 * (hi + hj[cc] - acc[r][cc]);
            }
        }
    }
    if (!SEG && gpoint) {
        double pj[TC];
#pragma unroll
        for (int cc = 0; cc < TC; ++cc) pj[cc] = gpoint[j0 + wj + 16 * (cc >> 1) + (cc & 1)];
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            const double pi = gpoint[i0 + wi + 16 * (r >> 1) + (r & 1)];
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) {
                if constexpr (KIND == K_DOT) acc[r][cc] += (pi != pj[cc]) ? a.point_c : 0.0;      // ring 0 of KL: two point masses
                else acc[r][cc] += fabs(pi - pj[cc]);
            }
        }
    }

    // ---- stores ----
    if ((a.dbg & 1) && acc[0][0] != -1.0) {
        // ablation: no stores
    } else if (a.tau) {
        // fused top-k: keep (j, v) for row i when v <= tau[i]; on a symmetric region the pair also
        // serves row j.  Diagonal tiles count each unordered pair once (i < j).
        gcdouble *tau = (gcdouble *)a.tau;
        double tj_[TC];
#pragma unroll
        for (int cc = 0; cc < TC; ++cc) {
            const int64_t j = j0 + wj + 16 * (cc >> 1) + (cc & 1);
            tj_[cc] = (a.mirror && j < a.j_end) ? tau[j] : -1.0;
        }
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            const int64_t i = i0 + wi + 16 * (r >> 1) + (r & 1);
            if (i >= a.i_end) continue;
            const double ti_ = tau[i];
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) {
                const int64_t j = j0 + wj + 16 * (cc >> 1) + (cc & 1);
                if (j >= a.j_end || i == j || (diag && i > j)) continue;
                const double v = acc[r][cc];
                const int jo = a.orig_of ? a.orig_of[j] : (int)j, io = a.orig_of ? a.orig_of[i] : (int)i;   // (permuted operands: original ids)
                if (v <= ti_ && jo >= 0) {
                    const int slot = atomicAdd(a.cand_cnt + i, 1);
                    if (slot < a.cand_cap) { a.cand_idx[i * a.cand_cap + slot] = jo; a.cand_val[i * a.cand_cap + slot] = v; }
                }
                if (v <= tj_[cc] && io >= 0) {
                    const int slot = atomicAdd(a.cand_cnt + j, 1);
                    if (slot < a.cand_cap) { a.cand_idx[j * a.cand_cap + slot] = io; a.cand_val[j * a.cand_cap + slot] = v; }
                }
            }
        }
    } else if (!diag && i0 + BT <= a.i_end && j0 + BT <= a.j_end) {
        // interior tile: no masks.  direct: each group of 8 lanes writes one 128-B line of a row
        gdouble *pd = out + (i0 + wi - a.rbase) * a.ld + (j0 + wj);
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            gdouble *row = pd + (int64_t)(16 * (r >> 1) + (r & 1)) * a.ld;
#pragma unroll
            for (int q = 0; q < TC / 2; ++q) store2_full<FAST>(row + 16 * q, acc[r][2 * q], acc[r][2 * q + 1], nt);
        }
        if (a.mirror) {
            // mirrored: same shape with the roles of rows and columns exchanged
            gdouble *pm = out + (j0 + wj - a.rbase) * a.ld + (i0 + wi);
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) {
                gdouble *row = pm + (int64_t)(16 * (cc >> 1) + (cc & 1)) * a.ld;
#pragma unroll
                for (int q = 0; q < TR / 2; ++q) store2_full<FAST>(row + 16 * q, acc[2 * q][cc], acc[2 * q + 1][cc], nt);
            }
        }
    } else {
        // edge or diagonal tile: per-element masks in tile-local 32-bit coordinates
        const int ilim = (int)min((int64_t)BT, a.i_end - i0), jlim = (int)min((int64_t)BT, a.j_end - j0);
        gdouble *tb = out + (i0 - a.rbase) * a.ld + j0;
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            int li = wi + 16 * (r >> 1) + (r & 1);
            asm volatile("" : "+v"(li) :: "memory");      // masks are formed here, not hoisted en bloc
            if (li >= ilim) continue;
            gdouble *row = tb + (int64_t)li * a.ld;
#pragma unroll
            for (int q = 0; q < TC / 2; ++q) {
                const int lj = wj + 16 * q;
                const bool ok0 = lj < jlim && (!diag || li <= lj);
                const bool ok1 = lj + 1 < jlim && (!diag || li <= lj + 1);
                store2<FAST>(row + lj, (diag && li == lj) ? 0.0 : acc[r][2 * q], (diag && li == lj + 1) ? 0.0 : acc[r][2 * q + 1],
                             ok0, ok1);
            }
        }
        if (a.mirror) {
            gdouble *mb = out + (j0 - a.rbase) * a.ld + i0;
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) {
                int lj = wj + 16 * (cc >> 1) + (cc & 1);
                asm volatile("" : "+v"(lj) :: "memory");
                if (lj >= jlim) continue;
                gdouble *row = mb + (int64_t)lj * a.ld;
#pragma unroll
                for (int q = 0; q < TR / 2; ++q) {
                    const int li = wi + 16 * q;
                    const bool ok0 = li < ilim && (!diag || li < lj);
                    const bool ok1 = li + 1 < ilim && (!diag || li + 1 < lj);
                    store2<FAST>(row + li, acc[2 * q][cc], acc[2 * q + 1][cc], ok0, ok1);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// pair_stream_kernel<KIND, 4>: the same arithmetic as a PERSISTENT workgroup that streams through its tiles
// (128 x 128 tiles, wave w owns quadrant (w >> 1, w & 1), one barrier per operand chunk).
//
// Measured on the one-tile kernel above at BA N = 100k (profiles/r02a_*): with the zero-pad distance down to ~137 words
// per pair the waves sat in s_waitcnt / s_barrier for 29 % of their cycles and the vector ALU was busy 62 % of the
// time even with stores disabled: every stall of one workgroup (tile prologue, per-chunk barrier skew, epilogue) leaves the
// SIMD to the one other wave.  Here 512 workgroups stay resident, the next tile's plan (coordinates, form decision, chunk
// list) is resolved one tile ahead by wave 0, and its first operand chunks are requested before the current tile's stores.
// (A one-wave, barrier-free 64 x 64 variant and an eight-wave 8 x 4-pairs-per-lane variant were built and measured:
// 247 vs 219 ms / 31 vs 29 ms, and 1052 vs 226 ms / 26.2 vs 22.3 ms -- profiles/r03_exp_nw8_defer_ab.txt -- both dropped.)
//
// Round 3 -- TILE COMPOSITION.  A tile side is no longer 128 consecutive nodes but EIGHT 16-NODE GRANULES named by a granule
// list (PairArgs::gl_i / gl_j, eight entries per list tile).  16 nodes are 16 fp64 = one 128-byte line of a matrix row, so the
// kernel's stores keep their shape (every 8-lane group writes one whole line) whatever granules make up a tile; the operand
// loads take a per-lane offset (a granule's slice of one column is 64 B of uint32 / 128 B of fp64, the direct-to-LDS
// instruction has per-lane addresses anyway).  hsd_pair_prepare sorts the list by the layers' rank-form key, so that the
// nodes of a tile need about the same number of rank words (BA N = 100k, Variant B: 99 -> ~66 words per pair together with
// the min form below); the natural list reproduces round 2's tiles.  Results do not depend on the composition: every pair
// is evaluated by the same operations in the same order, whichever tile it lands in.
//
// Round 3 -- MIN FORM of a rank segment.  Rank columns hold a node's member values above the layer's threshold in descending
// order, zeros beyond its own count.  With R_I / R_J the largest counts on the two sides of a tile pair, every node of the
// SHORTER side is zero from rank min(R_I, R_J) on, so beyond that point |u_i(r) - u_j(r)| is just the longer side's own value:
// the tile reads only ceil(min(R_I, R_J) / 2) rank words and adds the longer side's precomputed suffix sum
// (HsdSegDesc::suf_row0: one uint32 per node and word index) to the integer accumulator before the layer's fold.  The same
// integer as the bin form and the max form (tests/test_gpu_forms.py compares them bit for bit).
// ------------------------------------------------------------------------------------------
__host__ __device__ constexpr int stream_chunk_cols(int kind) { return chunk_cols(kind); }

// wait until at most `young` of this wave's vector-memory operations are outstanding (they retire in order, so that
// means everything older than the `young` newest has landed); a smaller count than the true one only waits longer
__device__ inline void wait_vm_young(int young) {
    if (young >= 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    else if (young >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (young >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (young >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (young >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

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

template <int CH, bool PACK>
__device__ inline int plan_segment_min(const SegDesc &sd, bool active, uint32_t RI, uint32_t RJ, int l, int2 *lst, int *smeta) {
    const SegChoice c = choose_segment_form<PACK>(sd, active, RI, RJ);
    if (l < 4) smeta[l] = c.sm;
    const int nch = (c.words + CH - 1) / CH;
    return emit_chunks<CH>(nch, c.col0, c.words, l, lst, c.k16);
}

// TOPK: the fused k-nearest consumer's instance -- a tile's values below their row's threshold are appended to that row's
// candidate list instead of being stored (a separate instantiation: the matrix kernels' code is not touched by it).
template <int KIND, int NW, bool TOPK = false>
__global__ __launch_bounds__(64 * NW, 2) void pair_stream_kernel(const PairArgs a) {
    using T = typename Elem<KIND>::type;
    using T2 = typename Vec2<T>::type;
    constexpr int TR = 8, TC = 8, BT = 128;
    constexpr bool SEG = (KIND == K_SAD || KIND == K_ZP || KIND == K_SAD16);
    constexpr bool U32 = sizeof(T) == 4;
    constexpr int CH = stream_chunk_cols(KIND);
    constexpr int STAGE = 2 * CH * BT;               // elements per stage: I block, then J block
    static_assert(NW == 4, "four waves per 128 x 128 tile");
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    T *sbase = reinterpret_cast<T *>(lds_raw);
    int2 *clist = reinterpret_cast<int2 *>(sbase + 2 * STAGE);   // [2][list_cap]: chunk lists of the current and the next tile
    int *nlist = reinterpret_cast<int *>(clist + 2 * a.list_cap);   // [2]: their lengths; [2]: the tile index drawn for the tile after next
    // [3][24] ring (current tile, next, the one after): the tile's granules (0-7: I side, 8-15: J side), per layer segment
    // the most members above its threshold on the I side (16-19) and on the J side (20-23; bit 31: see plan_segment_min)
    int *meta = nlist + 4;
    constexpr int META = 24;
    int *smeta = meta + 3 * META;                    // [2][4]: per layer segment, suffix row | side << 30, or -1
    SegDesc *segd = reinterpret_cast<SegDesc *>(smeta + 8);      // [4]
    // auxiliary per-node vectors of the current and the next tile: [2 tiles][n_aux][I: 128 doubles | J: 128 doubles]
    // (vector v < n_seg: 1/ring size of layer v; then the ring-0 value or the entropy term)
    double *aux = reinterpret_cast<double *>(segd + 4);          // (4 x 24 bytes: still 16-byte aligned)
    constexpr int AUX_STRIDE = 256;                  // doubles per vector: one 1-KiB direct-to-LDS instruction per side
    // suffix sums of the longer side of a rank segment: [2 tiles][n_seg][128 uint32]
    uint32_t *sufb = reinterpret_cast<uint32_t *>(aux + 2 * a.n_aux * AUX_STRIDE);
    // k-nearest sweep: the current tile's 128 row + 128 column thresholds as bfloat16 (rounded up: a looser threshold only appends a
    // few more candidates) -- 512 B is what the workgroup's LDS has left, and fetching the exact fp64 thresholds from global memory
    // in the epilogue cost a memory latency per tile (13 k cycles: every tile holds ~6 true neighbour pairs, none can be skipped)
    uint16_t *tau16b = reinterpret_cast<uint16_t *>(sufb + 2 * a.n_seg * 128);
    // k-nearest sweep of the raw fp64 distance on the pruned sweep's permuted operands (tiles of similar nodes): a WAVE whose 64 x 64 pairs have all
    // passed their row's and their column's threshold cannot hold a candidate any more (the sums only grow) and skips the arithmetic of the
    // tile's remaining chunks -- it keeps issuing its share of the operand requests and meeting the barriers, and its SIMD goes to the other
    // workgroup's wave meanwhile.  (Per workgroup -- all four quadrants dead, three LDS flags -- the same test found 4 % of the time, measured.)
    constexpr bool EARLY = TOPK && KIND == K_L1;
    const bool early_on = EARLY && a.orig_of != nullptr && !(a.dbg & 32);
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) double *)lds_raw);
    auto lds_off = [&](const void *p) __attribute__((always_inline)) -> uint32_t {
        return lds_base + (uint32_t)(reinterpret_cast<const char *>(p) - reinterpret_cast<const char *>(lds_raw));
    };

    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), l = threadIdx.x & 63;
    const int ly = lane_y(l), lx = lane_x(l);
    const int wi = (w >> 1) * 64 + 2 * ly, wj = (w & 1) * 64 + 2 * lx;
    // The lane id as the compiler cannot see through: everything derived from it at a point of use (LDS addresses of the
    // per-tile tables and vectors, request offsets, store offsets) is then formed there, instead of being hoisted out of the
    // tile loop into one more register that lives across the inner loop -- there are two dozen such values and no registers.
    auto lane = [&]() __attribute__((always_inline)) -> int { int v = l; asm volatile("" : "+v"(v)); return v; };
    auto wg_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    const T *__restrict__ XI = reinterpret_cast<const T *>(a.swap ? a.Y : a.X);
    const T *__restrict__ XJ = reinterpret_cast<const T *>(a.swap ? a.X : a.Y);
    gcdouble *const gH = (gcdouble *)a.H;
    gcdouble *const ginv = (gcdouble *)a.inv;
    gcdouble *const gpoint = (gcdouble *)a.point;
#if HSD_STAMPS
    const bool nt = !(a.dbg & 4);
#endif

    // this wave's share: XCD x (workgroups b, b + 8, ... share one) owns a contiguous run of the tile order
    const int x = blockIdx.x & 7, wx = blockIdx.x >> 3, nwx = gridDim.x >> 3;
    const int q = a.ntiles >> 3, rr = a.ntiles & 7;
    const int lo = x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q;
    const int hi = lo + (x < rr ? q + 1 : q);
    // the first three tiles of a workgroup are fixed (lo + wx, + nwx, + 2 nwx); from then on it draws the next free tile of
    // its XCD's run from a counter.  With a fixed share per workgroup the OLDER of the two workgroups of a CU wins every
    // issue arbitration, finishes its share early (measured: half the workgroups ended at 137 ms, the other half at 232 ms)
    // and leaves the CU half empty for the rest of the launch.
    int t = lo + wx;
    if (t >= hi) return;
    int src = x;                                      // (wave 0, lane 0) the XCD whose counter this workgroup draws from; -1: all dry
#if HSD_STAMPS
    const unsigned long long stamp0 = (a.dbg & 8) ? __builtin_amdgcn_s_memrealtime() : 0ull;   // HSD_DEBUG bit 3: lifetime stamps
#endif

    // the descriptions of the layer segments: copied to LDS once, read by the planning lanes tile by tile
    if constexpr (SEG) {
        if (w == 0 && l < 4) {
            SegDesc d{};
            d.rank_col0 = -1; d.suf_row0 = -1;
            if (l < a.n_seg) d = a.seg_tab[l];
            segd[l] = d;
        }
    }
    // Per-lane byte offsets of a tile's operand requests, from its granules (ring slot `ms`): this lane's granule of the I / J
    // side within a column slice (uint32 kinds: two columns per instruction, 4 lanes = 64 B per granule; fp64: one column,
    // 8 lanes = 128 B per granule).  Recomputed per request (two LDS reads): held across the tile they cost four registers
    // of a kernel that has none to spare.
    auto operand_offsets = [&](int ms, uint32_t &oI, uint32_t &oJ) __attribute__((always_inline)) {
        const int *g = meta + ms * META;
        const int lv = lane();
        if constexpr (U32) {
            const uint32_t colpart = (uint32_t)(lv >> 5) * (uint32_t)(a.npad * sizeof(T));
            oI = colpart + (uint32_t)g[(lv & 31) >> 2] * 64u + 16u * (lv & 3);
            oJ = colpart + (uint32_t)g[8 + ((lv & 31) >> 2)] * 64u + 16u * (lv & 3);
        } else {
            oI = (uint32_t)g[lv >> 3] * 128u + 16u * (lv & 7);
            oJ = (uint32_t)g[8 + (lv >> 3)] * 128u + 16u * (lv & 7);
        }
    };
    // requests the first `valid` columns of a chunk (the tail of a layer's last chunk is not fetched) of the tile in ring
    // slot `ms`; returns the number of direct-to-LDS instructions THIS wave issued
    auto stage_chunk = [&](int col, int valid, int st, int ms) __attribute__((always_inline)) -> int {
        // up to LOADS / 2 instructions per side and wave, each 64 lanes x 16 B = COLS columns x (BT nodes x sizeof(T)) bytes
        constexpr int LANES_PER_COL = BT * (int)sizeof(T) / 16;
        constexpr int COLS = 64 / LANES_PER_COL;
        static_assert((CH / COLS) % NW == 0, "stage instructions must divide over the waves");
        uint32_t offI, offJ;
        operand_offsets(ms, offI, offJ);
        asm volatile("" ::: "memory");               // no load / store moves across the request: vm_seq below relies on the order
        const uint32_t dst_lds = lds_base + (uint32_t)(st * STAGE * sizeof(T));
        int issued = 0;
#pragma unroll
        for (int k = 0; k < CH / COLS / NW; ++k) {
            const int ins = k * NW + w;
            if (ins * COLS < valid) {
                glds16_s(reinterpret_cast<uint64_t>(XI + (int64_t)(col + ins * COLS) * a.npad), offI,
                         dst_lds + (uint32_t)(ins * COLS * BT * sizeof(T)));
                glds16_s(reinterpret_cast<uint64_t>(XJ + (int64_t)(col + ins * COLS) * a.npad), offJ,
                         dst_lds + (uint32_t)((CH * BT + ins * COLS * BT) * sizeof(T)));
                issued += 2;
            }
        }
        asm volatile("" ::: "memory");
        return issued;
    };
    // one 1-KiB instruction per vector and side (128 doubles), dealt over the waves, and the suffix sums of the rank
    // segments (512 B: the longer side only); issued in front of the tile's first chunk, so the wait for that chunk covers them
    auto stage_aux = [&](int buf, int ms) __attribute__((always_inline)) -> int {
        int issued = 0;
        const int *g = meta + ms * META;
        const int lv = lane();
        const uint32_t aoffI = (uint32_t)g[lv >> 3] * 128u + 16u * (lv & 7), aoffJ = (uint32_t)g[8 + (lv >> 3)] * 128u + 16u * (lv & 7);
        asm volatile("" ::: "memory");
        for (int v = 0; v < a.n_aux; ++v) {
            if ((v & (NW - 1)) != w) continue;
            // (DOT: the entropy term, then -- closed form of ring 0 -- the ring-0 values; layer kinds: 1/ring size per layer, then the ring-0 values)
            gcdouble *src = v < a.n_seg ? ginv + (int64_t)v * a.npad : (KIND == K_DOT ? (v == a.n_seg ? gH : gpoint) : (gpoint ? gpoint : gH));
            const uint32_t dst = lds_off(aux + (buf * a.n_aux + v) * AUX_STRIDE);
            glds16_s(reinterpret_cast<uint64_t>(src), aoffI, dst);
            glds16_s(reinterpret_cast<uint64_t>(src), aoffJ, dst + 1024u);
            issued += 2;
        }
        if constexpr (SEG && U32) {
            for (int v = 0; v < a.n_seg; ++v) {
                if (((v + 3) & (NW - 1)) != w) continue;          // (the waves that carry no per-node vector first)
                const int sm = __builtin_amdgcn_readfirstlane(smeta[buf * 4 + v]);
                if (sm < 0) continue;
                const uint64_t rowp = reinterpret_cast<uint64_t>(a.suf + (int64_t)(sm & 0x3fffffff) * a.npad);
                const uint32_t dst = lds_off(sufb + (buf * a.n_seg + v) * 128);
                // a 128-node uint32 vector: 32 lanes, 4 per granule
                const uint32_t soff = (uint32_t)g[((sm >> 30) ? 8 : 0) + ((lv & 31) >> 2)] * 64u + 16u * (lv & 3);
                if (lv < 32) glds16_s(rowp, soff, dst);
                issued += 1;
            }
        }
        asm volatile("" ::: "memory");
        return issued;
    };
    // per-tile inputs of the plan, requested two tiles ahead by wave 0: the largest member counts of the two sides (lanes
    // 16-23: segment (l - 16) & 3, I side then J side) and the tile's granule ids (lanes 0-15)
    auto issue_meta = [&](int mti, int mtj) __attribute__((always_inline)) -> int {
        // ONE load per lane from an address the lane selects arithmetically (every source is an array of 32-bit words; lanes
        // with nothing to fetch re-read the first granule id): loads under per-lane conditions are merged by hipcc with a wait
        // after each, i.e. wave 0 -- and behind the next barrier the workgroup -- would stand still for a memory latency (and
        // for every operand request in flight: vmcnt(0)) at the top of every tile.
        int v = 0;
        mti &= TILE_INDEX_MASK;                       // (row-block launches: bit 30 of a tile's row index says "also write it mirrored")
        if (w == 0) {
            const int l = lane();
            const int *p = a.gl_i + (int64_t)8 * mti + (l < 8 ? l : 0);
            if (l >= 8 && l < 16) p = a.gl_j + (int64_t)8 * mtj + (l - 8);
            if constexpr (SEG) {
                if (a.tmax_i) {
                    if (l >= 16 && l < 20 && l - 16 < a.n_seg) p = reinterpret_cast<const int *>(a.tmax_i) + (int64_t)(l - 16) * a.ntl_i + mti;
                    if (l >= 20 && l < 24 && l - 20 < a.n_seg) p = reinterpret_cast<const int *>(a.tmax_j) + (int64_t)(l - 20) * a.ntl_j + mtj;
                }
            }
            v = *p;
        }
        return v;
    };
    // ... and parked in the ring (wave 0) while nothing younger than operand requests is outstanding
    auto park_meta = [&](int v, int ms) __attribute__((always_inline)) {
        if (w == 0) { const int l = lane(); if (l < META) meta[ms * META + l] = v; }
    };
    // decide bin form / rank form per segment from the tile's ring slot and write its chunk list + length and its suffix
    // rows into buffer `buf` (wave 0; the other waves read them behind the next barrier)
    auto plan = [&](int ms, int buf) __attribute__((always_inline)) {
        if constexpr (SEG) {
            if (w != 0) return;
            const int l = lane();
            const uint32_t RI = l < 4 ? (uint32_t)meta[ms * META + 16 + l] : 0u, RJ = l < 4 ? (uint32_t)meta[ms * META + 20 + l] : 0u;
            SegDesc sd = segd[l & 3];
            const int pre = plan_segment_min<CH, KIND == K_SAD16>(sd, l < a.n_seg, RI, RJ, l, clist + buf * a.list_cap, smeta + buf * 4);
            if (l == 0) nlist[buf] = pre;
        }
    };
    auto planned_chunks = [&](int buf) __attribute__((always_inline)) -> int {
        if constexpr (SEG) return __builtin_amdgcn_readfirstlane(nlist[buf]);
        else return (a.dcols_valid + CH - 1) / CH;
    };
    auto chunk_col = [&](int buf, int k) __attribute__((always_inline)) -> int {
        if constexpr (SEG) return __builtin_amdgcn_readfirstlane(clist[buf * a.list_cap + k].x);
        else return k * CH;
    };
    auto chunk_valid = [&](int buf, int k) __attribute__((always_inline)) -> int {
        if constexpr (SEG) return __builtin_amdgcn_readfirstlane(clist[buf * a.list_cap + k].y) & 0xff;
        else return min(CH, a.dcols_valid - k * CH);
    };

    // ---- prologue: plan of the first tile, coordinates of the next two ----
    int ti, tj;
    tile_from_index<true>(a, t, ti, tj);
    int cur = 0;                                      // list buffer of the tile being computed
    int ms = 0;                                       // its slot of the meta ring (next tile: ms + 1, the one after: ms + 2, mod 3)
    auto ring = [](int k) __attribute__((always_inline)) { return k >= 3 ? k - 3 : k; };
    const int t1 = t + nwx, t2 = t + 2 * nwx;
    bool has1 = t1 < hi, has2 = t2 < hi;
    int ti1 = 0, tj1 = 0, ti2 = 0, tj2 = 0;
    if (has1) tile_from_index<true>(a, t1, ti1, tj1);
    if (has2) tile_from_index<true>(a, t2, ti2, tj2);
    park_meta(issue_meta(ti, tj), 0);
    if (has1) park_meta(issue_meta(ti1, tj1), 1);
    plan(0, cur);
    __syncthreads();
    int n0 = planned_chunks(cur);

    int vm_seq = 0;                                   // vector-memory operations issued so far (lower bound: masked stores are not counted)
    int seq_of_stage[2] = {0, 0};                     // vm_seq right after the loads of the chunk that sits in each LDS stage
    int st = 0;                                       // LDS stage of the next chunk to be computed
    int ahead = 0;                                    // chunks of the current tile whose loads are already issued
    vm_seq += stage_aux(cur, ms);
    if (n0 > 0) { vm_seq += stage_chunk(chunk_col(cur, 0), chunk_valid(cur, 0), 0, ms); seq_of_stage[0] = vm_seq; ahead = 1; }
    if (n0 > 1) { vm_seq += stage_chunk(chunk_col(cur, 1), chunk_valid(cur, 1), 1, ms); seq_of_stage[1] = vm_seq; ahead = 2; }
    unsigned long long work = 0;
    bool first_landed = false;
#if HSD_STAMPS
    unsigned long long cyc_store = 0, cyc_loop = 0, cyc_epi = 0, n_tiles = 0;      // HSD_DEBUG bit 3 only
#endif

    // DEFERRED STORES (integer kinds, matrix output).  A finished tile's 64 result stores per wave do not leave in one burst
    // behind its last chunk -- 15 k cycles in which the wave computes nothing, the kernel's largest loss at ~65 words per pair
    // -- but in eight groups of eight beside the column trips of the NEXT tile's first segment (the long one: emit_chunks
    // puts the last layer first).  Nothing extra is live for it: until that segment's fold the next tile only fills its
    // integer sums, the fp64 accumulators still hold the previous tile's results, and the fold overwrites them instead of
    // adding to them.  Whatever has not left by then goes out in front of the fold.
    constexpr bool DEFERS = SEG && U32 && !TOPK;
    int pend_g = 8;                                   // next store group of the previous tile (8: nothing pending)
    int pend_ms = 0;                                  // ... its slot of the meta ring (row / column granules)
    bool pend_mirror = false;
    // the accumulators as register QUADRUPLES (two adjacent columns of a row): a 16-byte store takes one as it is
    v2f64 acc2[TR][TC / 2];
#define ACC(r, c) acc2[r][(c) >> 1][(c) & 1]
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int c = 0; c < TC; ++c) ACC(r, c) = 0.0;
    // The mirrored tile: entry (j, i) = this tile's (i, j).  A lane's 8 x 8 block holds rows 2 ly + {0, 1} + 16 a and columns
    // 2 lx + {0, 1} + 16 q, so the two ROWS of a 2 x 2 sub-block are adjacent columns of the mirrored matrix: exchanging the
    // off-diagonal elements of every 2 x 2 sub-block inside the lane (ACC(2a, 2q+1) <-> ACC(2a+1, 2q): 32 v_swap_b32, no
    // temporaries, no LDS) leaves in quadruple [2a + p][q] the mirrored row of column 2q + p with its columns 2a, 2a + 1 --
    // a 16-byte store again, with lx choosing the row and ly the 16 bytes of the line.  (Round 3 first exchanged whole blocks
    // between lane (ly, lx) and lane (lx, ly): 128 ds_bpermute per tile, 1.1 of 16.6 ms at BA N = 100k by a build without it.
    // The same swap written in C++ made hipcc keep six scratch accesses in the 512-SAD inner loop.)
    auto transpose_acc = [&]() __attribute__((always_inline)) {
#if HSD_ABLATE & 1
        return;                                       // -DHSD_ABLATE=1 (measurement build): the mirrored half without the exchange
#endif
#pragma unroll
        for (int aa = 0; aa < TR / 2; ++aa)
#pragma unroll
            for (int qq = 0; qq < TC / 2; ++qq) {
                const long long bx = __double_as_longlong(ACC(2 * aa, 2 * qq + 1)), by = __double_as_longlong(ACC(2 * aa + 1, 2 * qq));
                int xl = (int)bx, xh = (int)(bx >> 32), yl = (int)by, yh = (int)(by >> 32);
                asm("v_swap_b32 %0, %1" : "+v"(xl), "+v"(yl));
                asm("v_swap_b32 %0, %1" : "+v"(xh), "+v"(yh));
                ACC(2 * aa, 2 * qq + 1) = __longlong_as_double(((long long)xh << 32) | (unsigned int)xl);
                ACC(2 * aa + 1, 2 * qq) = __longlong_as_double(((long long)yh << 32) | (unsigned int)yl);
            }
    };
    // One group of a tile's result stores: g < 4: rows 16 GI[g] + 2 ly + {0, 1} x all four column granules (direct);
    // g >= 4: rows 16 GJ[g - 4] + 2 lx + {0, 1} of the mirrored tile x all four row granules (transpose_acc must have run in
    // between).  One scalar base per group + a 32-bit lane offset, formed here from the lane id (see lane()).  (Copying two
    // accumulators of adjacent rows together for every mirrored store instead of the exchange: hipcc spills 350 registers.)
    auto store_group = [&](int g, int gms, bool gmirror) __attribute__((always_inline)) -> int {
        int GI[4], GJ[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            GI[qq] = __builtin_amdgcn_readfirstlane(meta[gms * META + 4 * (w >> 1) + qq]);
            GJ[qq] = __builtin_amdgcn_readfirstlane(meta[gms * META + 8 + 4 * (w & 1) + qq]);
        }
        // HSD_DEBUG bit 4 (measurement): every tile stores into the first granules' place -- the same store instructions
        // without the HBM write stream (the lines stay in L2)
        if (a.dbg & 16) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) { GI[qq] = 4 * (w >> 1) + qq; GJ[qq] = 8 + 4 * (w & 1) + qq; }
        }
        const int lv = lane(), lyv = lane_y(lv), lxv = lane_x(lv);
        const uint32_t ld8 = (uint32_t)a.ld * 8u;
        if (g >= 4 && !gmirror) return 0;
#if HSD_ABLATE & 4
        if (g >= 4) return 0;                         // measurement build: the direct half only
#endif
#if HSD_ABLATE & 8
        if (g < 4) return 0;                          // measurement build: the mirrored half only
#endif
        auto put = [&](auto NT_) __attribute__((always_inline)) {
            constexpr bool NTS = decltype(NT_)::value;
            const uint32_t vd0 = (uint32_t)(2 * lyv) * ld8 + (uint32_t)(16 * lxv);
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                if (g == gg) {
                    // one scalar base per group (the row granule); the column granule goes into the 32-bit lane offset
                    uint64_t sb = (uint64_t)uniform64(reinterpret_cast<int64_t>(a.out + ((int64_t)16 * GI[gg] - a.rbase) * a.ld));
                    asm volatile("s_nop 4" : "+s"(sb));          // vector-ALU-written scalar -> address: five wait states (see gstore16_s)
#pragma unroll
                    for (int par = 0; par < 2; ++par)
#pragma unroll
                        for (int qq = 0; qq < TC / 2; ++qq)
                            gstore16_s<NTS>(sb, vd0 + (par ? ld8 : 0u) + (uint32_t)(128 * GJ[qq]), acc2[2 * gg + par][qq]);
                }
                if (g == 4 + gg) {
                    // mirrored rows 16 GJ[gg] + 2 lx + par (transpose_acc has run): quadruple [2 aa + par][gg] holds that row's
                    // columns 16 GI[aa] + 2 ly + {0, 1}
                    uint64_t sb = (uint64_t)uniform64(reinterpret_cast<int64_t>(a.out + ((int64_t)16 * GJ[gg] - a.rbase) * a.ld));
                    asm volatile("s_nop 4" : "+s"(sb));
                    const uint32_t vdm = (uint32_t)(2 * lxv) * ld8 + (uint32_t)(16 * lyv);
#pragma unroll
                    for (int par = 0; par < 2; ++par)
#pragma unroll
                        for (int aa = 0; aa < TR / 2; ++aa)
                            gstore16_s<NTS>(sb, vdm + (par ? ld8 : 0u) + (uint32_t)(128 * GI[aa]), acc2[2 * aa + par][gg]);
                }
            }
        };
#if HSD_STAMPS
        if (nt) put(std::true_type{}); else put(std::false_type{});       // HSD_DEBUG bit 2: plain instead of non-temporal stores
#elif defined(HSD_PLAIN_STORES)
        put(std::false_type{});                                           // measurement build
#else
        put(std::true_type{});
#endif
        return 8;
    };

    // a layer's integer sums: zero here, and every fold leaves them at zero again (every layer closes with a fold)
    T part[TR][TC];
    if constexpr (SEG) {
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
            for (int c = 0; c < TC; ++c) part[r][c] = (T)0;
    }
    // A tile's accumulators start at the ring-0 term |p_i - p_j| (or zero): one subtraction and a sign mask per pair instead of
    // zeroing them and adding the term at the end.  Layer kinds: at the tile's first fold (every kernel of those kinds starts
    // from the same value, so they stay bit-identical to each other).
    auto init_acc = [&](int buf) __attribute__((always_inline)) {
        if (gpoint) {
            double pj[TC];
            const int lv = lane(), wi = (w >> 1) * 64 + 2 * lane_y(lv), wj = (w & 1) * 64 + 2 * lane_x(lv);
            const double *ax = aux + (buf * a.n_aux + a.n_seg) * AUX_STRIDE;
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) pj[cc] = ax[128 + wj + 16 * (cc >> 1) + (cc & 1)];
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const double pi = ax[wi + 16 * (r >> 1) + (r & 1)];
#pragma unroll
                for (int cc = 0; cc < TC; ++cc) ACC(r, cc) = fabs(pi - pj[cc]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < TR; ++r)
#pragma unroll
                for (int cc = 0; cc < TC; ++cc) ACC(r, cc) = 0.0;
        }
    };
    while (true) {
        if constexpr (!SEG) {
#pragma unroll
            for (int r = 0; r < TR; ++r)
#pragma unroll
                for (int c = 0; c < TC; ++c) ACC(r, c) = 0.0;
        }
        bool first_fold = true;                       // the tile's first fold sets the accumulators (init_acc)
        // the next tile's plan: its inputs were parked in the ring during the previous tile
        int n1 = 0;
        if (has1) plan(ring(ms + 1), cur ^ 1);
        wg_barrier();                                 // plan of the next tile visible; every wave is done with the previous tile's LDS
        if (has1) n1 = planned_chunks(cur ^ 1);
        bool dead = false;                            // (EARLY) no pair of this wave's quadrant can become a candidate any more
        if constexpr (TOPK) {
            // this tile's thresholds (wave 3; 2 lanes per 16-node granule; the column half only where columns carry thresholds)
            if (w == 3) {
                const int lv = lane();
                const int gsel = lv < 16 ? (lv >> 1) : 8 + ((lv - 16) >> 1);
                const uint32_t toff = (uint32_t)meta[ms * META + (gsel & 15)] * 32u + 16u * (lv & 1);
                asm volatile("" ::: "memory");
                if (lv < (a.mirror ? 32 : 16)) glds16_s(reinterpret_cast<uint64_t>(a.tau16), toff, lds_off(tau16b));
                asm volatile("" ::: "memory");
                vm_seq += 1;
            }
        }
        // ... the inputs of the tile after it are requested now (picked up behind the chunk loop)
        const int meta2 = has2 ? issue_meta(ti2, tj2) : 0;     // (not counted in vm_seq: only operations that are certainly issued are)
        // ... and the tile after that is drawn from the XCD's counter (the answer is picked up at the end of this tile)
        int drawn = 0;
        if (w == 0 && l == 0 && src >= 0) drawn = atomicAdd(a.tile_ctr + src, 1);
        int ahead1 = 0;                               // chunks of the NEXT tile already requested
        bool aux1 = false;                            // ... and its auxiliary vectors (the per-lane offsets are the next tile's from then on)
        const bool diag = a.sym && ti == tj;          // a tile on the diagonal of a symmetric region is computed in full and not mirrored

        if (n0 == 0) {
            // a tile without operand chunks (no histogram layer compared): nothing below waits for its per-node vectors
            if constexpr (DEFERS) {
                for (; pend_g < 8; ++pend_g) {
                    if (pend_g == 4 && pend_mirror) transpose_acc();
                    vm_seq += store_group(pend_g, pend_ms, pend_mirror);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wg_barrier();
            if constexpr (SEG) init_acc(cur);
        }
#if HSD_STAMPS
        const unsigned long long c_l0 = (a.dbg & 8) ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
        for (int k = 0; k < n0; ++k) {
            int cvalid, seg = 0, n16 = 0;
            bool ends_seg = false;
            if constexpr (SEG) {
                const int meta = __builtin_amdgcn_readfirstlane(clist[cur * a.list_cap + k].y);
                cvalid = meta & 0xff;
                ends_seg = (meta & 0x100) != 0;
                n16 = (meta >> 9) & 0x3f;                 // leading words of uint16 pairs; the rest are uint8 quadruples
                seg = meta >> 16;
            } else {
                cvalid = min(CH, a.dcols_valid - k * CH);
            }
            work += (unsigned)cvalid;
            if (a.dbg & 2) cvalid = n16 = 0;
            // chunk k has landed once everything older than the operations issued after its loads has retired
            // (the first chunk of a tile was waited for before the previous tile's stores went out)
            // (... and then the barrier at the top of the tile has already told every wave so: no second one)
            if (!(k == 0 && first_landed)) {
                wait_vm_young(__builtin_amdgcn_readfirstlane(vm_seq - (st ? seq_of_stage[1] : seq_of_stage[0])));
                wg_barrier();                         // every wave's part of chunk k has landed, and chunk k - 1 is consumed
            }
            if constexpr (EARLY) {
                if (dead) cvalid = 0;
            }
            const T *sI = sbase + st * STAGE;
            const T *sJ = sI + CH * BT;
            // the other stage is free (its chunk was consumed in the previous iteration): request the chunk after next
            // of this tile, or -- at the end of the tile -- the first chunks of the next one
            if (k >= 1 || ahead < 2) {
                // (k == 0 with two chunks in flight: nothing to request, both stages are taken)
                if (ahead < n0) {
                    vm_seq += stage_chunk(chunk_col(cur, ahead), chunk_valid(cur, ahead), st ^ 1, ms);
                    seq_of_stage[st ^ 1] = vm_seq; ++ahead;
                } else if (has1 && !aux1) {
                    vm_seq += stage_aux(cur ^ 1, ring(ms + 1));     // the next tile's vectors, then its first chunk
                    aux1 = true;
                    if (n1 > 0) {
                        vm_seq += stage_chunk(chunk_col(cur ^ 1, 0), chunk_valid(cur ^ 1, 0), st ^ 1, ring(ms + 1));
                        seq_of_stage[st ^ 1] = vm_seq; ahead1 = 1;
                    }
                }
            }
            T av[2][TR], bv[2][TC];
            auto load_ops = [&](int c, int buf) __attribute__((always_inline)) {
#pragma unroll
                for (int qq = 0; qq < TR / 2; ++qq) {
                    const T2 v = *reinterpret_cast<const T2 *>(sI + c * BT + wi + 16 * qq);
                    av[buf][2 * qq] = v.x;
                    av[buf][2 * qq + 1] = v.y;
                }
#pragma unroll
                for (int qq = 0; qq < TC / 2; ++qq) {
                    const T2 v = *reinterpret_cast<const T2 *>(sJ + c * BT + wj + 16 * qq);
                    bv[buf][2 * qq] = v.x;
                    bv[buf][2 * qq + 1] = v.y;
                }
            };
            // BYTES: the words are four uint8 columns (the uint8 range of a packed layer) instead of two uint16 columns
            auto compute = [&](int buf, auto BYTES) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int cc = 0; cc < TC; ++cc) {
                        if constexpr (KIND == K_L1) ACC(r, cc) += fabs(av[buf][r] - bv[buf][cc]);
                        else if constexpr (KIND == K_SAD) part[r][cc] = __usad(av[buf][r], bv[buf][cc], part[r][cc]);
                        else if constexpr (KIND == K_SAD16) {
                            if constexpr (decltype(BYTES)::value) part[r][cc] = __builtin_amdgcn_sad_u8(av[buf][r], bv[buf][cc], part[r][cc]);
                            else part[r][cc] = __builtin_amdgcn_sad_u16(av[buf][r], bv[buf][cc], part[r][cc]);
                        } else if constexpr (KIND == K_ZP) part[r][cc] += fabs(av[buf][r] - bv[buf][cc]);
                        else ACC(r, cc) = fma(av[buf][r], bv[buf][cc], ACC(r, cc));
                    }
            };
            constexpr int UNR = sizeof(T) == 4 ? 8 : HSD_F64_UNROLL;
            // the words [c, cend) of the chunk
            auto sweep = [&](int c, const int cend, auto BYTES) __attribute__((always_inline)) {
            if (c >= cend) return;
            load_ops(c, 0);
#pragma unroll 1
            for (; UNR > 2 && c + UNR <= cend; c += UNR) {          // straight-line body: counted LDS waits (see pair_kernel)
                if constexpr (DEFERS) {
                    // one group of the previous tile's results in front of every 8-word trip (~2 100 cycles of arithmetic apart:
                    // the store path never sees a burst; two bursts of 32 in front of the first two chunks: +0.25 of 16.7 ms)
                    if (pend_g < 8) {
                        if (pend_g == 4 && pend_mirror) transpose_acc();
                        vm_seq += store_group(pend_g, pend_ms, pend_mirror);
                        ++pend_g;
                    }
                }
#pragma unroll
                for (int u = 0; u < UNR; u += 2) {
                    // (no clamp: the word after a chunk's last is read and never used -- the J block / the next stage / the
                    // tile lists lie there, inside the workgroup's LDS -- and the trip's eight words share one address register)
                    load_ops(c + u + 1, 1);
                    HSD_PIN_ORDER;
                    compute(0, BYTES);
                    HSD_PIN_ORDER;
                    load_ops(c + u + 2, 0);
                    HSD_PIN_ORDER;
                    compute(1, BYTES);
                    HSD_PIN_ORDER;
                }
            }
            if constexpr (DEFERS) {
                // ... and one in front of the shorter rest of a sweep (four words or more)
                if (c + 4 <= cend && pend_g < 8) {
                    if (pend_g == 4 && pend_mirror) transpose_acc();
                    vm_seq += store_group(pend_g, pend_ms, pend_mirror);
                    ++pend_g;
                }
            }
#pragma unroll 1
            for (; c < cend; c += 2) {
                load_ops(c + 1, 1);
                HSD_PIN_ORDER;
                compute(0, BYTES);
                HSD_PIN_ORDER;
                if (c + 1 < cend) {
                    load_ops(c + 2, 0);
                    HSD_PIN_ORDER;
                    compute(1, BYTES);
                    HSD_PIN_ORDER;
                }
            }
            };
            if constexpr (KIND == K_SAD16) {
                sweep(0, n16, std::false_type{});
                sweep(n16, cvalid, std::true_type{});
            } else {
                sweep(0, cvalid, std::false_type{});
            }
            if constexpr (EARLY) {
                // From chunk 2 on the tile's thresholds are in LDS (wave 3 requested them before this tile's third chunk, whose arrival
                // every wave has waited for).  A pair is alive while its partial sum is at or below its row's or (mirrored tiles) its
                // column's threshold; compared on the high words like the candidate epilogue (a superset: never drops a live pair); the
                // ring-0 term is not in the sums yet, which only makes them smaller.
                if (early_on && k >= 2 && !dead && k + 1 < n0) {
                    auto hi32 = [](double v) __attribute__((always_inline)) -> int { return (int)(__double_as_longlong(v) >> 32); };
                    auto thr_hi = [&](uint32_t h16) __attribute__((always_inline)) -> int { return hi32((double)__uint_as_float(h16 << 16)); };
                    const int lv = lane(), twi = (w >> 1) * 64 + 2 * lane_y(lv), twj = (w & 1) * 64 + 2 * lane_x(lv);
                    const bool tile_mirror = a.mirror && !diag;
                    int cmin[TC];
#pragma unroll
                    for (int cc = 0; cc < TC; ++cc) cmin[cc] = 0x7fffffff;
                    bool live = false;
#pragma unroll
                    for (int r = 0; r < TR; ++r) {
                        int rmin = 0x7fffffff;
#pragma unroll
                        for (int cc = 0; cc < TC; ++cc) {
                            const int h = hi32(ACC(r, cc));
                            rmin = min(rmin, h);
                            cmin[cc] = min(cmin[cc], h);
                        }
                        live = live || rmin <= thr_hi(tau16b[twi + 16 * (r >> 1) + (r & 1)]);
                    }
                    if (tile_mirror) {
#pragma unroll
                        for (int cc = 0; cc < TC; ++cc) live = live || cmin[cc] <= thr_hi(tau16b[128 + twj + 16 * (cc >> 1) + (cc & 1)]);
                    }
                    dead = __builtin_amdgcn_ballot_w64(live) == 0ull;
                }
            }
            if constexpr (SEG) {
                if (ends_seg) {
                    if constexpr (DEFERS) {
                        // whatever of the previous tile has not left yet: the fold below overwrites its values
                        for (; pend_g < 8; ++pend_g) {
                            if (pend_g == 4 && pend_mirror) transpose_acc();
                            vm_seq += store_group(pend_g, pend_ms, pend_mirror);
                        }
                    }
                    // the layer closes: sum_c |..| / max(n_i, n_j) == sum * min(1/n_i, 1/n_j) (empty rings carry inv = 2).  Row by
                    // row, with only the column side's vector held: this is where the kernel's register demand peaks (both
                    // accumulator sets are live), and a spill here is a scratch reload that waits behind every outstanding
                    // operand request.
                    const double *ax = aux + (cur * a.n_aux + seg) * AUX_STRIDE;      // 1/ring size, staged with the tile
                    const int lv = lane(), wi = (w >> 1) * 64 + 2 * lane_y(lv), wj = (w & 1) * 64 + 2 * lane_x(lv);
                    double vj_[TC];
#pragma unroll
                    for (int cc = 0; cc < TC; ++cc) vj_[cc] = ax[128 + wj + 16 * (cc >> 1) + (cc & 1)];
                    // min form: the rank values of the longer side beyond the words both sides were read for
                    int sm = -1;
                    if constexpr (U32) sm = __builtin_amdgcn_readfirstlane(smeta[cur * 4 + seg]);
                    const uint32_t *sb = sufb + (cur * a.n_seg + seg) * 128;
                    if (U32 && sm >= 0 && (sm >> 30)) {
#pragma unroll
                        for (int cc = 0; cc < TC; ++cc) {
                            const T sv = (T)sb[wj + 16 * (cc >> 1) + (cc & 1)];
#pragma unroll
                            for (int r = 0; r < TR; ++r) part[r][cc] += sv;
                        }
                    }
                    const bool row_suffix = U32 && sm >= 0 && !(sm >> 30);
                    // (deferred stores: the previous tile's results have all been issued by now)
                    if (first_fold) init_acc(cur);
                    if (row_suffix) {
#pragma unroll
                        for (int r = 0; r < TR; ++r) {
                            const T sv = (T)sb[wi + 16 * (r >> 1) + (r & 1)];
#pragma unroll
                            for (int cc = 0; cc < TC; ++cc) part[r][cc] += sv;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < TR; ++r) {
                        const double vi = ax[wi + 16 * (r >> 1) + (r & 1)];
#pragma unroll
                        for (int cc = 0; cc < TC; ++cc) {
                            ACC(r, cc) = fma((double)part[r][cc], fmin(vi, vj_[cc]), ACC(r, cc));
                            part[r][cc] = (T)0;
                        }
                        HSD_PIN_ORDER;
                    }
                    first_fold = false;
                }
            }
            st ^= 1;
        }
#if HSD_STAMPS
        const unsigned long long c_l1 = (a.dbg & 8) ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
        // the second chunk of the next tile goes out BEFORE this tile's stores: its loads then do not queue behind them
        if (w == 0 && l == 0) {
            // position of the drawn tile in the order, or -1 when every run is exhausted.  A workgroup whose own XCD's run
            // has run dry moves on to the next XCD's counter (the XCDs do not finish together: 217 .. 232 ms at BA N = 100k);
            // those few draws are waited for on the spot
            int pos = -1;
            while (src >= 0) {
                const int slo = src < rr ? src * (q + 1) : rr * (q + 1) + (src - rr) * q;
                const int shi = slo + (src < rr ? q + 1 : q);
                const int cand = slo + 3 * nwx + drawn;
                if (cand < shi) { pos = cand; break; }
                src = (src + 1) & 7;
                if (src == x) { src = -1; break; }
                drawn = atomicAdd(a.tile_ctr + src, 1);
            }
            nlist[2] = pos;
        }
        wg_barrier();                                 // the stage of the last chunk is free only when every wave has consumed it
        // The inputs of the tile after next have arrived (they were requested a tile's worth of arithmetic ago): into the ring
        // -- behind the barrier, because the slot they take was the PREVIOUS tile's, whose deferred stores the other waves may
        // have been issuing until their last chunk; and before anything else is requested or stored -- hipcc does not see the
        // hand-issued operations below, and a wait for these loads behind them would be a wait for those.
        if (has2) park_meta(meta2, ring(ms + 2));
        if (has1 && !aux1) { vm_seq += stage_aux(cur ^ 1, ring(ms + 1)); aux1 = true; }
        if (ahead1 == 0 && n1 > 0) {
            vm_seq += stage_chunk(chunk_col(cur ^ 1, 0), chunk_valid(cur ^ 1, 0), st, ring(ms + 1));
            seq_of_stage[st] = vm_seq; ahead1 = 1;
        }
        if (ahead1 == 1 && n1 > 1) {
            vm_seq += stage_chunk(chunk_col(cur ^ 1, 1), chunk_valid(cur ^ 1, 1), st ^ 1, ring(ms + 1));
            seq_of_stage[st ^ 1] = vm_seq; ahead1 = 2;
        }
        // A wave can have 63 vector-memory operations outstanding at most, and they retire in order: a wait for operand
        // loads issued BEFORE this tile's 128 stores still needs about half of those stores acknowledged.  So the next
        // tile's first chunk (requested a whole chunk ago: it has landed) is waited for now, while nothing younger than
        // its second chunk is in flight; the next tile then starts computing at once and the stores drain under its first
        // chunk (measured: stores cost 4.4 of 26.5 ms at BA N = 100k, zero-pad, without this).
        first_landed = ahead1 > 0;
        if (first_landed) wait_vm_young(__builtin_amdgcn_readfirstlane(vm_seq - (st ? seq_of_stage[1] : seq_of_stage[0])));

        // ---- epilogue arithmetic ----
        if constexpr (KIND == K_DOT) {
            double hj[TC];
            const int lv = lane(), wi = (w >> 1) * 64 + 2 * lane_y(lv), wj = (w & 1) * 64 + 2 * lane_x(lv);
            const double *ax = aux + (cur * a.n_aux + a.n_seg) * AUX_STRIDE;
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) hj[cc] = ax[128 + wj + 16 * (cc >> 1) + (cc & 1)];
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const double hi_ = ax[wi + 16 * (r >> 1) + (r & 1)];
#pragma unroll
                for (int cc = 0; cc < TC; ++cc) {
                    if (a.hmode == 0) ACC(r, cc) = (a.swap ? hj[cc] : hi_) - ACC(r, cc);
                    else ACC(r, cc) = 0.5 * (hi_ + hj[cc] - ACC(r, cc));
                }
            }
        }
        if (!SEG && gpoint) {
            double pj[TC];
            const int lv = lane(), wi = (w >> 1) * 64 + 2 * lane_y(lv), wj = (w & 1) * 64 + 2 * lane_x(lv);
            const double *ax = aux + (cur * a.n_aux + a.n_seg + (KIND == K_DOT ? 1 : 0)) * AUX_STRIDE;
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) pj[cc] = ax[128 + wj + 16 * (cc >> 1) + (cc & 1)];
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const double pi = ax[wi + 16 * (r >> 1) + (r & 1)];
#pragma unroll
                for (int cc = 0; cc < TC; ++cc) {
                    if constexpr (KIND == K_DOT) ACC(r, cc) += (pi != pj[cc]) ? a.point_c : 0.0;     // ring 0 of KL: two point masses
                    else ACC(r, cc) += fabs(pi - pj[cc]);
                }
            }
        }
        // ---- stores ----
        asm volatile("" ::: "memory");
#if HSD_STAMPS
        const unsigned long long c_s0 = (a.dbg & 8) ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
        // a.mirror == 2: the launch covers a whole row block -- the rectangles beside the diagonal square (stored once) and the
        // square's tiles on or above the diagonal, of which those above it are flagged to be written mirrored as well
        const bool mirror = a.mirror == 2 ? ((ti >> 30) & 1) != 0 : (a.mirror && !diag);
        if constexpr (TOPK) {
          // wave 3's request for the tile's thresholds is older than the next tile's first chunk, which every wave has just waited for
          // (or nothing is left to wait for: the last tile): one barrier makes its arrival known to the other waves
          if (!first_landed) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          wg_barrier();
          if (!((a.dbg & 1) && ACC(0, 0) != -1.0)) {      // HSD_DEBUG bit 0 (measurement): the sweep without its epilogue
            // this tile's granules: the node ids of its rows / columns
            int GI[4], GJ[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                GI[qq] = __builtin_amdgcn_readfirstlane(meta[ms * META + 4 * (w >> 1) + qq]);
                GJ[qq] = __builtin_amdgcn_readfirstlane(meta[ms * META + 8 + 4 * (w & 1) + qq]);
            }
            // keep (j, v) for row i when v <= tau[i]; on a symmetric region the pair also serves row j (a tile on the
            // diagonal holds both orientations of its pairs: row i only, and not i itself).  The thresholds are this launch's
            // (the host tightens them between the phases of a sweep), rounded up to bfloat16, staged with the tile.
            // A tile holds ~6 such pairs among its 16 384 (BA N = 100k, k = 20), so the test that runs for every value must be
            // cheap: distances are non-negative, so the HIGH words of two doubles order like the doubles, and a bfloat16 threshold
            // has an empty low word -- v <= tau implies hi(v) <= hi(tau).  Per lane the smallest high word of each of its 8 rows
            // and 8 columns (112 v_min_i32) against the row's / column's threshold; only a row or column that passes (rare:
            // the branch is skipped when no lane of the wave does) tests its eight values exactly.  (Before: two fp64 compares
            // and two divergent branches per value behind a per-granule vote: 0.45 ms of every 1.8-ms phase.)
            auto bf = [](uint16_t h) __attribute__((always_inline)) -> double { return (double)__uint_as_float((uint32_t)h << 16); };
            // (SIGNED high words: a symmetric-KL value that rounding left at -1e-17 is below every threshold, as it should be)
            auto hi32 = [](double v) __attribute__((always_inline)) -> int { return (int)(__double_as_longlong(v) >> 32); };
            int cmin[TC];
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) cmin[cc] = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                int rmin = 0x7fffffff;
#pragma unroll
                for (int cc = 0; cc < TC; ++cc) {
                    const int h = hi32(ACC(r, cc));
                    rmin = min(rmin, h);
                    cmin[cc] = min(cmin[cc], h);
                }
                const double ti_ = bf(tau16b[wi + 16 * (r >> 1) + (r & 1)]);
                if (rmin <= hi32(ti_)) {
                    const int64_t i = (int64_t)16 * GI[r >> 1] + 2 * ly + (r & 1);
#pragma unroll
                    for (int cc = 0; cc < TC; ++cc) {
                        const int64_t j = (int64_t)16 * GJ[cc >> 1] + 2 * lx + (cc & 1);
                        const double v = ACC(r, cc);
                        if (v <= ti_ && i != j) {
                            const int jo = a.orig_of ? a.orig_of[j] : (int)j;       // (permuted operands: the list holds original node ids)
                            if (jo >= 0) {
                                const int slot = atomicAdd(a.cand_cnt + i, 1);
                                if (slot < a.cand_cap) { a.cand_idx[i * a.cand_cap + slot] = jo; a.cand_val[i * a.cand_cap + slot] = v; }
                            }
                        }
                    }
                }
            }
            if (mirror) {
#pragma unroll
                for (int cc = 0; cc < TC; ++cc) {
                    const double tj_ = bf(tau16b[128 + wj + 16 * (cc >> 1) + (cc & 1)]);
                    if (cmin[cc] <= hi32(tj_)) {
                        const int64_t j = (int64_t)16 * GJ[cc >> 1] + 2 * lx + (cc & 1);
#pragma unroll
                        for (int r = 0; r < TR; ++r) {
                            const int64_t i = (int64_t)16 * GI[r >> 1] + 2 * ly + (r & 1);
                            const double v = ACC(r, cc);
                            if (v <= tj_) {
                                const int io = a.orig_of ? a.orig_of[i] : (int)i;
                                if (io >= 0) {
                                    const int slot = atomicAdd(a.cand_cnt + j, 1);
                                    if (slot < a.cand_cap) { a.cand_idx[j * a.cand_cap + slot] = io; a.cand_val[j * a.cand_cap + slot] = v; }
                                }
                            }
                        }
                    }
                }
            }
            // (vm_seq stays as it is: it must remain a LOWER bound of the operations issued, and these are data dependent)
          }
        } else if ((a.dbg & 1) && ACC(0, 0) != -1.0) {
            // ablation: no stores
        } else if (DEFERS && has1) {
            // the results stay in the accumulators and leave beside the next tile's first segment
            pend_g = 0;
            pend_ms = ms;
            pend_mirror = mirror;
        } else {
            for (int g = 0; g < 8; ++g) {
                if (g == 4 && mirror) transpose_acc();
                vm_seq += store_group(g, ms, mirror);
            }
        }
#if HSD_STAMPS
        if (a.dbg & 8) {
            asm volatile("" ::: "memory");
            const unsigned long long c_s1 = __builtin_amdgcn_s_memtime();
            cyc_loop += c_l1 - c_l0; cyc_epi += c_s0 - c_l1; cyc_store += c_s1 - c_s0; ++n_tiles;
        }
#endif
        if (!has1) break;
        // ---- rotate: next tile becomes current ----
        ti = ti1; tj = tj1; ti1 = ti2; tj1 = tj2;
        n0 = n1; cur ^= 1; ms = ring(ms + 1); ahead = ahead1;
        has1 = has2;
        const int t3 = __builtin_amdgcn_readfirstlane(nlist[2]);
        has2 = t3 >= 0;
        if (has2) tile_from_index<true>(a, t3, ti2, tj2);
    }
    if (a.work && threadIdx.x == 0) atomicAdd(a.work, work * (unsigned long long)(BT * BT));
#if HSD_STAMPS
    if ((a.dbg & 8) && a.stamps && threadIdx.x == 0) {
        // measurement hook: (start, end) of this workgroup in 100 MHz ticks
        unsigned long long *sp = a.stamps + 6 * (size_t)blockIdx.x;
        sp[0] = stamp0;
        sp[1] = __builtin_amdgcn_s_memrealtime();
        sp[2] = cyc_loop; sp[3] = cyc_epi; sp[4] = cyc_store; sp[5] = n_tiles;
    }
#endif
}
#undef ACC

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
};

// W1 raw samples: column block of layer k>=1 holds G[b] = (cc[b]/n) * (s[b+1]-s[b]), b < B-1.
// One thread per (node, layer).
__global__ void prep_l1_kernel(PrepArgs p, double *__restrict__ X, int col_begin_first, int cols_per_layer,
                               double *__restrict__ point) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = p.layer_lo + blockIdx.y;
    if (i >= p.n) return;
    const uint32_t *c = p.counts + ((int64_t)i * p.L + k) * p.B;
    if (k == 0) {               // ring 0 = {i}: W1 of two point masses = |v_i - v_j|
        point[i] = p.support[p.node_bin[i]];
        return;
    }
    uint32_t tot = 0;
    for (int b = 0; b < p.B; ++b) tot += c[b];
    if (tot == 0) atomicOr(p.flag, 1);
    const int first = p.layer_lo == 0 ? 1 : p.layer_lo;
    const int64_t cb = col_begin_first + (int64_t)(k - first) * cols_per_layer;
    const double nn = (double)tot;
    uint32_t cc = 0;
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
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = p.layer_lo + blockIdx.y;
    if (i >= p.n) return;
    const uint32_t *c = p.counts + ((int64_t)i * p.L + k) * p.B;
    if (k == 0) {
        point[i] = p.support[p.node_bin[i]];
        return;
    }
    const int first = p.layer_lo == 0 ? 1 : p.layer_lo;
    const int s = k - first;
    uint32_t tot = 0;
    for (int b = 0; b < p.B; ++b) tot += c[b];
    inv[(int64_t)s * p.npad + i] = tot ? 1.0 / (double)tot : 2.0;
    ntot[(int64_t)s * p.npad + i] = tot;
    const int64_t cb = seg_begin[s];
    X[cb * p.npad + i] = (T)((double)tot * p.support[0]);
    uint32_t tail = tot;
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
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int first = p.layer_lo == 0 ? 1 : p.layer_lo;
    const int sgi = blockIdx.y;
    const int k = first + sgi;
    if (i >= p.n || seg_rank_col0[sgi] < 0) return;
    const uint32_t *c = p.counts + ((int64_t)i * p.L + k) * p.B;
    const int cap = seg_rank_cap[sgi], fb = seg_first_bin[sgi];
    const uint32_t theta = fb > 0 ? (uint32_t)p.support[fb - 1] : 0u;
    uint32_t *col = X + (int64_t)seg_rank_col0[sgi] * p.npad + i;
    // suffix sums (min form of the persistent kernel): suf[w] = sum of ALL the node's rank values from word w on, i.e. the
    // whole sum above theta minus what the first w words hold -- also where the rank columns themselves are cut off at the cap
    uint32_t *suf = SUF + (int64_t)seg_suf_row0[sgi] * p.npad + i;
    uint32_t total = 0;
    for (int b = p.B - 1; b >= fb; --b) total += c[b] * ((uint32_t)p.support[b] - theta);
    const int want = PACK ? 2 * cap : cap;           // values to emit
    int r = 0;
    uint32_t lo = 0, prefix = 0;
    suf[0] = total;
    for (int b = p.B - 1; b >= fb && r < want; --b) {
        const uint32_t v = (uint32_t)p.support[b] - theta;
        for (uint32_t m = c[b]; m > 0 && r < want; --m, ++r) {
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
    if (PACK && (r & 1)) {                            // odd count: upper half stays 0
        col[(int64_t)(r >> 1) * p.npad] = lo;
        suf[(int64_t)((r >> 1) + 1) * p.npad] = total - prefix;
    }
    // (rows beyond the node's own words stay 0: the matrix was cleared)
}

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
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int first = p.layer_lo == 0 ? 1 : p.layer_lo;
    const int sgi = blockIdx.y;
    if (i >= p.n) return;
    const uint32_t *c = p.counts + ((int64_t)i * p.L + first + sgi) * p.B;
    uint32_t run = 0;
    int q = n_cand - 1;
    for (int b = p.B - 1; b >= 0; --b) {
        while (q >= 0 && cand_bin[q] > b) { above[((int64_t)sgi * n_cand + q) * p.npad + i] = run; --q; }
        run += c[b];
    }
    for (; q >= 0; --q) above[((int64_t)sgi * n_cand + q) * p.npad + i] = run;
}

// rmax[s][g] = largest ring size of layer segment s within the 64-node granule g
__global__ void granule_max_kernel(const uint32_t *__restrict__ ntot, int64_t npad, int ngran, uint32_t *__restrict__ rmax) {
    const int g = blockIdx.x, sgi = blockIdx.y;
    uint32_t v = ntot[(int64_t)sgi * npad + (int64_t)g * 64 + threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_down((int)v, o, 64));
    if (threadIdx.x == 0) rmax[(int64_t)sgi * ngran + g] = v;
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

// KL: P = (c + alpha) / (n + alpha B), LP = log P, H = sum P LP.  symmetric: X = [P ; LP], Y = [LP ; P].
__global__ void prep_kl_kernel(PrepArgs p, double alpha, int symmetric, int half_cols /* padded cols of one [P] block */,
                               double *__restrict__ X, double *__restrict__ Y, double *__restrict__ Hpart /* [nl][npad] */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kk = blockIdx.y;
    const int k = p.layer_lo + kk;
    if (i >= p.n) return;
    const uint32_t *c = p.counts + ((int64_t)i * p.L + k) * p.B;
    uint32_t tot = 0;
    for (int b = 0; b < p.B; ++b) tot += c[b];
    if (tot == 0 && alpha == 0.0) atomicOr(p.flag, 2);
    const double den = (double)tot + alpha * (double)p.B;
    double h = 0.0;
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

template <int KIND, bool FAST, int TR>
hipError_t launch_pair(hsd_ctx *ctx, const PairArgs &a, int grid, hipStream_t stream) {
    auto kern = pair_kernel<KIND, FAST, TR>;
    // 2 stages x (I, J) operand chunks, then (segmented kinds) the tile's chunk list + its length
    const size_t lds = (size_t)2 * 2 * chunk_cols(KIND) * (16 * TR) * sizeof(typename Elem<KIND>::type) + (size_t)a.list_cap * sizeof(int2) + 16;
    // the attribute is per device: remembered per context (one context per GPU), not per process
    size_t &attr_lds = ctx->lds_attr[reinterpret_cast<const void *>(kern)];
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_lds = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, stream, a);
    return hipGetLastError();
}

template <int KIND, int TR>
hipError_t launch_kind(hsd_ctx *ctx, const PairArgs &a, int grid, hipStream_t stream) {
    // fast path: 16-B aligned operand slices (direct-to-LDS loads) and 16-B aligned output pairs
    const bool fast = (a.i_begin % 4 == 0) && (a.j_begin % 4 == 0) && (a.npad % 4 == 0) && (a.ld % 2 == 0) &&
                      (reinterpret_cast<uintptr_t>(a.out) % 16 == 0);
    return fast ? launch_pair<KIND, true, TR>(ctx, a, grid, stream) : launch_pair<KIND, false, TR>(ctx, a, grid, stream);
}

int tile_edge(const hsd_ctx *ctx) {
    // 128 x 128 (8 x 8 pairs per lane) halves the LDS operand reads per pair and wins whenever the
    // arithmetic dominates; the short uint32 kernels at small column counts are bound by the store
    // stream and prefer many small independent workgroups (measured in DESIGN.md).  HSD_TILE overrides.
    if (ctx->mode == PAIR_ZP_F64) return 64;      // fallback kind is built for 64 x 64 only
    const char *e = hsd_hook(H_TILE);
    if (e) return atoi(e) == 128 ? 128 : 64;
    if ((ctx->mode == PAIR_SAD_U32 || ctx->mode == PAIR_SAD_U16) && ctx->seg_begin[ctx->n_seg] <= 128) return 64;   // padded bin-form words
    return 128;
}

// Explicit tile order, cached per region shape: S x S groups of tiles (row-major inside a group).  The kernels give
// XCD x the x-th eighth of the list, so the groups are dealt to the XCDs round-robin (XCD x gets groups x, x + 8, ...)
// and each eighth holds one XCD's groups: every XCD then walks a sample of the WHOLE pair space.  With row-major group
// order one XCD owned the top rows and another the bottom-right corner -- equal work while every tile costs the same,
// but with the rank form a tile of two young (low-degree) node blocks costs a fraction of a hub tile, and the XCD that
// owned the cheap corner idled (wave occupancy 73-84 % at BA N = 100k, profiles/r02b).  HSD_XCD_DEAL=0: plain order.
// nullptr (formula order in the kernel) when disabled or on allocation failure.
// cls: 0 = every tile; 1 = interior tiles only (off the diagonal of a symmetric region, not in a partial last row /
// column: the persistent kernel's share); 2 = the others (stored with per-element masks by the one-tile kernel).
// phase / n_phases / phase_w: only the groups whose (group row + group column) % n_phases lies in [phase, phase + phase_w) -- an
// anti-diagonal sample of the pair space in which every block of rows meets phase_w / n_phases of the column blocks (the
// k-nearest consumer tightens its thresholds between such phases).
const int2 *tile_order(hsd_ctx *ctx, int tiles_i, int tiles_j, int sym, int cls, bool part_i, bool part_j, int *count, int phase = 0,
                       int n_phases = 1, int phase_w = 1) {
    const char *e = hsd_hook(H_SUPERTILE);
    const int S = e ? atoi(e) : 8;
    if (S <= 1 && cls == 0 && n_phases <= 1) { if (count) *count = -1; return nullptr; }
    const char *ed = hsd_hook(H_XCD_DEAL);
    const int deal = (ed && atoi(ed) == 0) ? 0 : 1;
    if (n_phases < 1) n_phases = 1;
    if (phase_w < 1) phase_w = 1;
    // (S < 256, phase < 32, n_phases <= 32, phase_w <= 32)
    const int key = (std::max(S, 1) & 0xff) + (phase << 8) + (n_phases << 13) + (phase_w << 19) + (deal << 25) + (cls << 26) + ((int)part_i << 28) + ((int)part_j << 29);
    for (size_t k = 0; k < ctx->tile_lists.size(); ++k) {
        const auto &c = ctx->tile_lists[k];
        if (c.tiles_i == tiles_i && c.tiles_j == tiles_j && c.sym == sym && c.group == key) {
            // least recently used first: a hit moves to the back, so the lists of the call in progress (a phased sweep
            // takes a few dozen) are the last to be evicted
            const hsd_ctx::TileList hit = c;
            ctx->tile_lists.erase(ctx->tile_lists.begin() + (long)k);
            ctx->tile_lists.push_back(hit);
            if (count) *count = hit.count;
            return reinterpret_cast<const int2 *>(hit.d);
        }
    }
    const int G = std::max(S, 1);
    std::vector<int2> order;
    order.reserve(sym ? (size_t)tiles_i * (tiles_i + 1) / 2 : (size_t)tiles_i * tiles_j);
    std::vector<size_t> group_begin;
    for (int sr = 0; sr < tiles_i; sr += G)
        for (int sc = sym ? sr : 0; sc < tiles_j; sc += G) {
            { const int res = (sr / G + sc / G) % n_phases; if (res < phase || res >= phase + phase_w) continue; }
            group_begin.push_back(order.size());
            for (int ti = sr; ti < std::min(sr + G, tiles_i); ++ti)
                for (int tj = std::max(sc, sym ? ti : 0); tj < std::min(sc + G, tiles_j); ++tj) {
                    const bool masked = (sym && ti == tj) || (part_i && ti == tiles_i - 1) || (part_j && tj == tiles_j - 1);
                    if (cls == 0 || (cls == 1) == !masked) order.push_back(make_int2(ti, tj));
                }
        }
    group_begin.push_back(order.size());
    if (deal && group_begin.size() > 9) {
        std::vector<int2> dealt;
        dealt.reserve(order.size());
        const size_t ng = group_begin.size() - 1;
        for (size_t x = 0; x < 8; ++x)
            for (size_t g = x; g < ng; g += 8) dealt.insert(dealt.end(), order.begin() + (long)group_begin[g], order.begin() + (long)group_begin[g + 1]);
        order.swap(dealt);
    }
    if (count) *count = (int)order.size();
    void *d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(order.size(), 1) * sizeof(int2)) != hipSuccess) { if (count) *count = -1; return nullptr; }
    if (!order.empty() && hipMemcpy(d, order.data(), order.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        if (count) *count = -1;
        return nullptr;
    }
    if (ctx->tile_lists.size() >= 256) {
        // (hipFree synchronises the device: rare -- 256 lists are more than any one call hands out)
        (void)hipFree(ctx->tile_lists.front().d);
        ctx->tile_lists.erase(ctx->tile_lists.begin());
    }
    ctx->tile_lists.push_back({tiles_i, tiles_j, sym, key, d, (int)order.size()});
    return reinterpret_cast<const int2 *>(d);
}

template <int KIND, int NW, bool TOPK = false>
hipError_t launch_stream(hsd_ctx *ctx, const PairArgs &a, hipStream_t stream) {
    auto kern = pair_stream_kernel<KIND, NW, TOPK>;
    constexpr int CH = stream_chunk_cols(KIND);
    constexpr int BT = 128;
    // stages, chunk lists + lengths, granule ids + suffix rows of two tiles, per-node vectors, suffix sums
    const size_t lds = (size_t)2 * 2 * CH * BT * sizeof(typename Elem<KIND>::type) + (size_t)2 * a.list_cap * sizeof(int2) + 16 + 3 * 24 * 4 + 32 + 4 * sizeof(HsdSegDesc) +
                       (size_t)2 * a.n_aux * 256 * sizeof(double) + (size_t)2 * a.n_seg * 128 * sizeof(uint32_t) + (TOPK ? 512 : 0);
    // (two workgroups per CU need 2 x lds <= 160 KiB: the integer kinds with three per-node vectors sit ~300 bytes below that -- at 96
    // bytes above it, after the segment descriptions had grown, the k-nearest sweep of BA N = 200k ran at half occupancy: 105 vs 66 ms)
    if (hsd_hook(H_PLAN_VERBOSE) && 2 * lds > (size_t)ctx->prop.maxSharedMemoryPerMultiProcessor)
        fprintf(stderr, "[hsd stream<%d,%d,%d>] %zu bytes of LDS per workgroup: ONE workgroup per CU\n", KIND, NW, (int)TOPK, lds);
    // the attribute is per device: remembered per context (one context per GPU), not per process
    size_t &attr_lds = ctx->lds_attr[reinterpret_cast<const void *>(kern)];
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_lds = lds;
    }
    // persistent grid: two waves per SIMD on every CU (HSD_WAVES_PER_CU overrides), a multiple of 8 so that every XCD
    // gets the same number of workgroups; a workgroup with no tile in its XCD's run exits at once
    const char *ew = hsd_hook(H_WAVES_PER_CU);
    const int per_cu = ew ? std::max(NW, atoi(ew)) : 8;
    const int64_t resident = (int64_t)(per_cu / NW) * ctx->prop.multiProcessorCount / 8 * 8;
    const int grid = (int)std::min<int64_t>(resident, round_up(a.ntiles, 8));
    {
        hipError_t e = hipMemsetAsync(a.tile_ctr, 0, 8 * sizeof(int), stream);
        if (e != hipSuccess) return e;
    }
    if (a.dbg & 8) {
        // HSD_DEBUG bit 3: per-workgroup lifetimes -> stderr (synchronises; measurement only; its own buffer)
        PairArgs b = a;
        unsigned long long *d_st = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&d_st), (size_t)grid * 48) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemset(d_st, 0, (size_t)grid * 48);
        b.stamps = d_st;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, b);
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h6((size_t)grid * 6), h((size_t)grid * 2);
        (void)hipMemcpy(h6.data(), d_st, h6.size() * 8, hipMemcpyDeviceToHost);
        for (int g = 0; g < grid; ++g) { h[2 * g] = h6[6 * g]; h[2 * g + 1] = h6[6 * g + 1]; }
        (void)hipFree(d_st);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int g = 0; g < grid; ++g) if (h[2 * g]) { t0 = std::min(t0, h[2 * g]); t1 = std::max(t1, h[2 * g + 1]); }
        std::vector<double> life, endt;
        for (int g = 0; g < grid; ++g) if (h[2 * g]) { life.push_back((h[2 * g + 1] - h[2 * g]) * 1e-2); endt.push_back((h[2 * g + 1] - t0) * 1e-2); }
        std::sort(endt.begin(), endt.end());
        if (endt.empty()) return hipGetLastError();
        double per_xcd[8] = {0};
        int cnt[8] = {0};
        for (int g = 0; g < grid; ++g) if (h[2 * g]) { per_xcd[g & 7] += (h[2 * g + 1] - t0) * 1e-2; ++cnt[g & 7]; }
        fprintf(stderr, "[hsd stream<%d,%d>] %zu workgroups, span %.1f us; end times us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f; mean end per b%%8:", KIND, NW,
                endt.size(), (t1 - t0) * 1e-2, endt.front(), endt[endt.size() / 10], endt[endt.size() / 2], endt[endt.size() * 9 / 10], endt.back());
        for (int x8 = 0; x8 < 8; ++x8) fprintf(stderr, " %.0f", cnt[x8] ? per_xcd[x8] / cnt[x8] : 0.0);
        double cl = 0, ce = 0, cs = 0, nt = 0;
        for (int g = 0; g < grid; ++g) { cl += (double)h6[6 * g + 2]; ce += (double)h6[6 * g + 3]; cs += (double)h6[6 * g + 4]; nt += (double)h6[6 * g + 5]; }
        if (nt > 0) fprintf(stderr, "; wave-0 cycles per tile: chunk loop %.0f, fold + next-tile requests %.0f, stores %.0f (%.0f tiles)", cl / nt, ce / nt, cs / nt, nt);
        fprintf(stderr, "\n");
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, a);
    return hipGetLastError();
}

// The persistent kernel takes every region whose rows and columns start on a 16-node granule and whose output pairs are
// 16-byte aligned, except the fp64 zero-pad fallback (its two fp64 accumulator sets do not fit 8 x 8 pairs per lane); the
// fused k-nearest consumer runs its own instantiation (TOPK).  HSD_STREAM=0 or an explicit HSD_TILE: the
// one-tile-per-workgroup kernel for every tile (A/B measurements; the tests run both and compare them bit for bit).
int stream_waves(const hsd_ctx *ctx, const PairArgs &a) {
    if (const char *ew = hsd_hook(H_STREAM)) if (atoi(ew) == 0) return 0;
    if (hsd_hook(H_TILE)) return 0;
    if (ctx->mode == PAIR_ZP_F64) return 0;
    const bool aligned = (a.i_begin % 16 == 0) && (a.j_begin % 16 == 0) && (a.npad % 4 == 0) &&
                         (a.tau || ((a.ld % 2 == 0) && a.ld < (int64_t)30000000 && (reinterpret_cast<uintptr_t>(a.out) % 16 == 0)));
    if (!aligned) return 0;
    // a resident workgroup needs a few dozen tiles to amortise its start-up and to level the tail (measured at ER N = 10k,
    // 3160 tiles for 512 workgroups: zero-pad 0.210 ms against 0.178 ms with one 64 x 64 tile per workgroup); HSD_STREAM=1 forces it
    const int64_t ti = (a.i_end - a.i_begin) / 128, tj = (a.j_end - a.j_begin) / 128;
    const int64_t tiles = a.sym ? ti * (ti + 1) / 2 : ti * tj;
    if (tiles <= 0) return 0;
    // (tools_dev/xover.py, end of round 3: the integer kinds cross over at ~11 tiles per workgroup -- zero-pad, BA N = 14k: 0.385 one-tile /
    // 0.358 persistent, N = 16k: 0.500 / 0.376 -- the fp64 kinds within 3 % of each other between 12 and 16 tiles)
    const int64_t per_wg = (ctx->mode == PAIR_SAD_U16 || ctx->mode == PAIR_SAD_U32) ? 11 : 16;
    const char *ew = hsd_hook(H_STREAM);
    if (!(ew && atoi(ew) != 0) && tiles < per_wg * 2 * (int64_t)ctx->prop.multiProcessorCount) return 0;
    return 4;
}

hipError_t launch_one_tile(hsd_ctx *ctx, const PairArgs &a, int BT, hipStream_t stream) {
    const int g = a.ntiles;
    switch (ctx->mode) {
        case PAIR_L1_F64: return BT == 128 ? launch_kind<K_L1, 8>(ctx, a, g, stream) : launch_kind<K_L1, 4>(ctx, a, g, stream);
        case PAIR_SAD_U32: return BT == 128 ? launch_kind<K_SAD, 8>(ctx, a, g, stream) : launch_kind<K_SAD, 4>(ctx, a, g, stream);
        case PAIR_SAD_U16: return BT == 128 ? launch_kind<K_SAD16, 8>(ctx, a, g, stream) : launch_kind<K_SAD16, 4>(ctx, a, g, stream);
        case PAIR_ZP_F64: return launch_kind<K_ZP, 4>(ctx, a, g, stream);      // fallback kind: 64 x 64 only
        case PAIR_KL:
        case PAIR_SKL: return BT == 128 ? launch_kind<K_DOT, 8>(ctx, a, g, stream) : launch_kind<K_DOT, 4>(ctx, a, g, stream);
        default: return hipErrorInvalidValue;
    }
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

// tau16[i] = tau[i] rounded UP to bfloat16 (tau >= 0), i in [lo, hi)
__global__ void tau_bf16_up_kernel(const double *__restrict__ tau, int64_t lo, int64_t hi, uint16_t *__restrict__ tau16) {
    const int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    const double x = tau[i];
    float f = (float)x;
    if ((double)f < x) f = nextafterf(f, INFINITY);
    uint32_t b = __float_as_uint(f) >> 16;
    if (__uint_as_float(b << 16) < f) ++b;             // (positive values: the truncated pattern is below, the next one at or above)
    tau16[i] = (uint16_t)b;
}


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

// Xp[c][p] = X[c][orig[p]] (0 at padding positions); blockIdx.y = column
__global__ void tk_gather_kernel(const double *__restrict__ X, int64_t npad, const int32_t *__restrict__ orig, double *__restrict__ Xp) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npad) return;
    const int32_t o = orig[p];
    const int64_t c = blockIdx.y;
    Xp[c * npad + p] = o >= 0 ? X[c * npad + o] : 0.0;
}

// largest threshold (bfloat16 patterns of non-negative values order like the values) among a tile's 128 positions
__global__ void tk_tilemax_kernel(const uint16_t *__restrict__ tau16, int ntiles, uint16_t *__restrict__ tmax) {
    const int t = blockIdx.x, l = threadIdx.x;          // 64 threads
    if (t >= ntiles) return;
    unsigned m = max((unsigned)tau16[(int64_t)t * 128 + l], (unsigned)tau16[(int64_t)t * 128 + 64 + l]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if (l == 0) tmax[t] = (uint16_t)m;
}

// keeps the tile pairs of a list that some row's threshold can still reach: lower bound of every pair of the two tiles (sum over the
// features of the gap between their intervals) against the larger of the two tiles' largest thresholds.  drop_band > 0: pairs |ti - tj| <= drop_band
// are dropped as well (the first launch of the pruned sweep has already evaluated them).  The order of the output is whatever the
// atomics give: the sweep's results are sets.
__global__ void tk_filter_kernel(const int2 *__restrict__ in, int n_in, const double *__restrict__ lo, const double *__restrict__ hi,
                                 const double *__restrict__ scale, int nf, int nt,
                                 const uint16_t *__restrict__ tmax, int drop_band, int2 *__restrict__ out, int *__restrict__ count) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_in) return;
    const int2 t = in[k];
    const int ti = t.x & TILE_INDEX_MASK, tj = t.y;
    if (drop_band && abs(tj - ti) <= drop_band) return;
    double lb = 0.0;
    for (int f = 0; f < nf; ++f) {
        const double a0 = lo[(int64_t)f * nt + ti], a1 = hi[(int64_t)f * nt + ti], b0 = lo[(int64_t)f * nt + tj], b1 = hi[(int64_t)f * nt + tj];
        // (zero-pad layers: the gap between the rings' value sums, divided by the larger ring -- at least the smaller of the tiles' smallest 1 / ring size)
        lb += fmax(0.0, fmax(a0 - b1, b0 - a1)) * (scale ? fmin(scale[(int64_t)f * nt + ti], scale[(int64_t)f * nt + tj]) : 1.0);
    }
    const unsigned m = max((unsigned)tmax[ti], (unsigned)tmax[tj]);
    const double thr = (double)__uint_as_float(m << 16);
    // (the bound and the kernel's distances are sums of the same terms in different orders: a relative 1e-9 covers the rounding of both)
    if (lb * (1.0 - 1e-9) - 1e-9 <= thr) out[atomicAdd(count, 1)] = t;
}

// thresholds by node -> thresholds by position (0 at padding positions: nothing is at distance <= 0 from them but themselves)
__global__ void tk_permute_tau_kernel(const int32_t *__restrict__ orig, int64_t npos, const double *__restrict__ tau_node, double *__restrict__ tau_pos) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    const int32_t o = orig[p];
    tau_pos[p] = o >= 0 ? tau_node[o] : 0.0;
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

// The granule list of up to three node ranges [r[0], r[1]) + [r[2], r[3]) + [r[4], r[5]) taken one after the other (each begins on a
// granule and is a whole number of 128-node tiles; empty ranges are skipped): the granules of each range in the order
// hsd_pair_prepare chose for the graph (HSD_GRAN_SORT=0: natural order), with the per-tile maxima of the rank segments.  Cached per
// context until the operands are released; false if it cannot be built.
bool gran_list_ranges(hsd_ctx *ctx, const int64_t (&r)[6], hipStream_t stream, hsd_ctx::GranList *out) {
    const char *es = hsd_hook(H_GRAN_SORT);
    const int sorted = (!ctx->gran_order.empty() && !(es && atoi(es) == 0)) ? 1 : 0;      // (the two orders of a range are cached side by side)
    for (size_t k = 0; k < ctx->gran_lists.size(); ++k) {
        const auto &c = ctx->gran_lists[k];
        if (c.sorted == sorted && std::equal(r, r + 6, c.r)) {
            // least recently used first (like tile_lists): a hit moves to the back, so the lists of the call in progress are the last to go
            const hsd_ctx::GranList hit = c;
            ctx->gran_lists.erase(ctx->gran_lists.begin() + (long)k);
            ctx->gran_lists.push_back(hit);
            *out = hit;
            return true;
        }
    }
    std::vector<int32_t> gl;
    int64_t want = 0;
    for (int q = 0; q < 3; ++q) {
        const int64_t g0 = r[2 * q] / 16, g1 = r[2 * q + 1] / 16;
        if (g1 <= g0) continue;
        want += g1 - g0;
        if (sorted) { for (int32_t g : ctx->gran_order) if (g >= g0 && g < g1) gl.push_back(g); }
        else for (int64_t g = g0; g < g1; ++g) gl.push_back((int32_t)g);
        if ((int64_t)gl.size() != want || gl.size() % 8 != 0) return false;       // (every range a whole number of tiles)
    }
    if (gl.empty()) return false;
    hsd_ctx::GranList L{};
    std::copy(r, r + 6, L.r);
    L.sorted = sorted;
    L.ntiles = (int)(gl.size() / 8);
    if (hipMalloc(reinterpret_cast<void **>(&L.d_gl), gl.size() * sizeof(int32_t)) != hipSuccess) return false;
    if (hipMemcpy(L.d_gl, gl.data(), gl.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(L.d_gl); return false; }
    if (ctx->d_rmax16 && ctx->n_seg > 0) {
        const int64_t cnt = (int64_t)L.ntiles * ctx->n_seg;
        if (hipMalloc(reinterpret_cast<void **>(&L.d_tmax), (size_t)cnt * sizeof(uint32_t)) != hipSuccess) { (void)hipFree(L.d_gl); return false; }
        // on the stream of the launch that asked for the list: ordered in front of it (a later call finds the list finished --
        // every hsd_pair_* call joins its streams before it returns)
        hipLaunchKernelGGL(list_tile_max_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, L.d_gl, L.ntiles, ctx->d_rmax16,
                           ctx->ngran16, ctx->n_seg, L.d_tmax);
    }
    // Bounded: a caller that walks many different (r0, r1) row blocks -- the host-output loop of hsd_pair_rows takes one list per staging
    // block -- would otherwise grow the cache without limit.  One call holds at most six lists (a row block's square and two rectangles), far
    // below the bound; hipFree synchronises the device, so a list still read by a launch in flight is finished with before it goes.
    constexpr size_t GRAN_LIST_CACHE = 64;
    while (ctx->gran_lists.size() >= GRAN_LIST_CACHE) {
        (void)hipFree(ctx->gran_lists.front().d_gl);
        (void)hipFree(ctx->gran_lists.front().d_tmax);
        ctx->gran_lists.erase(ctx->gran_lists.begin());
    }
    ctx->gran_lists.push_back(L);
    *out = L;
    return true;
}

// the list of one range [begin, end)
bool gran_list(hsd_ctx *ctx, int64_t begin, int64_t end, hipStream_t stream, hsd_ctx::GranList *out) {
    const int64_t r[6] = {begin, end, 0, 0, 0, 0};
    return gran_list_ranges(ctx, r, stream, out);
}

int launch_region(hsd_ctx *ctx, PairArgs a, int BT, hipStream_t stream, bool allow_stream = true) {
    if (ctx->mode == PAIR_NONE) return hsd_fail(ctx, HSD_ERR_NO_SIGNATURE, "hsd_pair_prepare has not been called");
    if (a.i_end <= a.i_begin || a.j_end <= a.j_begin) return HSD_OK;
    const bool wass = ctx->mode == PAIR_L1_F64 || ctx->mode == PAIR_SAD_U32 || ctx->mode == PAIR_SAD_U16;
    const bool dot = ctx->mode == PAIR_KL || ctx->mode == PAIR_SKL;
    // (KL kinds: the entropy term, + the ring-0 values when layer 0 is taken in closed form)
    const int n_aux = ctx->n_seg + (dot ? (a.point ? 2 : 1) : ((wass ? a.point != nullptr : true) ? 1 : 0));
    // LDS budget of the persistent kernel (2 workgroups per CU): two tiles x three vectors x 2 KiB; deeper hierarchies
    // (more than two compared histogram layers + ring 0) take the one-tile kernel for every tile
    const int nw = (allow_stream && n_aux <= 3) ? stream_waves(ctx, a) : 0;
    if (a.gl_ov && !nw) return hsd_fail(ctx, HSD_ERR_INVALID, "permuted operands need the persistent kernel");
    if (nw) {
        // Whole 128-node tiles of both ranges: the persistent kernel, tiles composed from the ranges' granule lists (a tile on
        // the diagonal of a symmetric region is computed in full and not mirrored: no masks anywhere).  What is left over at
        // the end of a range (< 128 nodes) are strips for the one-tile kernel with its masked stores, in natural order.
        const int64_t i_full = a.i_begin + (a.i_end - a.i_begin) / 128 * 128, j_full = a.j_begin + (a.j_end - a.j_begin) / 128 * 128;
        hsd_ctx::GranList LI{}, LJ{};
        bool ok;
        if (a.gl_ov) {
            // (the pruned k-nearest sweep: one square region over the permuted operands, its list made by tk_prepare)
            LI.d_gl = const_cast<int32_t *>(a.gl_ov); LI.d_tmax = const_cast<uint32_t *>(a.tmax_ov); LI.ntiles = a.ntl_ov;
            LJ = LI;
            ok = a.sym != 0;
        } else {
            ok = gran_list(ctx, a.i_begin, i_full, stream, &LI);
            if (ok) { if (a.sym) LJ = LI; else ok = gran_list(ctx, a.j_begin, j_full, stream, &LJ); }
        }
        if (ok) {
            PairArgs b = a;
            b.i_end = i_full; b.j_end = j_full;
            b.tiles_i = LI.ntiles; b.tiles_j = LJ.ntiles;
            const int64_t nb = a.sym ? (int64_t)b.tiles_i * (b.tiles_i + 1) / 2 : (int64_t)b.tiles_i * b.tiles_j;
            if (nb > 0x7fffffff) return hsd_fail(ctx, HSD_ERR_TOO_LARGE, "too many tiles (%lld)", (long long)nb);
            // The Wasserstein kinds compute a tile on the diagonal of a symmetric region in full inside the persistent kernel
            // (|a - b| sums are the same bits in both orientations and an exact 0 on the diagonal).  KL is mirrored from the
            // upper triangle (model/HSD.py:132-134) and its dot products are not: its diagonal tiles keep the one-tile
            // kernel's masks (natural composition: the KL kinds have no sorted order).
            const bool diag_masked = a.sym && !wass;
            int n_in = -1, n_mask = 0;
            const int2 *l_in = nullptr;
            if (a.explicit_tiles) { l_in = a.explicit_tiles; n_in = a.n_explicit; }       // (the pruned k-nearest sweep's first launch)
            else l_in = tile_order(ctx, b.tiles_i, b.tiles_j, a.sym, diag_masked ? 1 : 0, false, false, &n_in, a.phase, a.n_phases, a.phase_w);
            const int2 *l_mask = diag_masked ? tile_order(ctx, b.tiles_i, b.tiles_j, a.sym, 2, false, false, &n_mask, a.phase, a.n_phases, a.phase_w) : nullptr;
            if (!l_in && n_in < 0 && a.n_phases <= 1 && !diag_masked) {
                // (formula order: HSD_SUPERTILE=1 or the list could not be allocated)
                n_in = (int)nb;
            }
            if (n_in >= 0 && n_mask >= 0) {
                hipError_t e = hipSuccess;
                b.tiles = l_in;
                b.ntiles = n_in;
                b.tile_ctr = ctx->d_tile_ctr + 8 * (a.ctr_slot & 3);
                b.n_aux = n_aux;
                b.gl_i = LI.d_gl; b.gl_j = LJ.d_gl;
                b.tmax_i = LI.d_tmax; b.tmax_j = LJ.d_tmax;
                b.ntl_i = LI.ntiles; b.ntl_j = LJ.ntiles;
                b.suf = a.suf ? a.suf : ctx->d_suf;
                b.stamps = nullptr;
                if (a.tau) {
                    b.dbg &= ~8;                          // (the lifetime stamps are a matrix-launch hook)
                    // this launch's thresholds (they tighten between the phases of a sweep) as bfloat16, rounded up: rows [i_begin,
                    // i_end) carry thresholds; the columns as well when the region is mirrored onto itself
                    hipLaunchKernelGGL(tau_bf16_up_kernel, dim3((unsigned)((a.i_end - a.i_begin + 255) / 256)), dim3(256), 0, stream, a.tau, a.i_begin, a.i_end,
                                       const_cast<uint16_t *>(a.tau16));
                }
                if (a.tau && a.prune_lo && l_in && n_in > 0 && ctx->d_tk_list && n_in <= ctx->tk_list_cap) {
                    // pruned sweep: drop the tile pairs that no row's CURRENT threshold can reach (tk_filter_kernel); the count comes
                    // back to the host -- one small synchronisation per launch of the sweep
                    int kept = 0;
                    hipLaunchKernelGGL(tk_tilemax_kernel, dim3((unsigned)a.prune_nt), dim3(64), 0, stream, a.tau16 + a.i_begin, a.prune_nt, ctx->d_tk_tmax);
                    e = hipMemsetAsync(ctx->d_tk_count, 0, sizeof(int), stream);
                    hipLaunchKernelGGL(tk_filter_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, stream, l_in, n_in, a.prune_lo, a.prune_hi, a.prune_scale, a.prune_nf,
                                       a.prune_nt, ctx->d_tk_tmax, a.prune_drop_band, reinterpret_cast<int2 *>(ctx->d_tk_list), ctx->d_tk_count);
                    if (e == hipSuccess) e = hipMemcpyAsync(&kept, ctx->d_tk_count, sizeof(int), hipMemcpyDeviceToHost, stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(stream);
                    if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "k-nearest tile filter: %s", hipGetErrorString(e));
                    if (hsd_hook(H_TOPK_VERBOSE)) fprintf(stderr, "[hsd topk] pruned launch: %d of %d tile pairs kept\n", kept, n_in);
                    b.tiles = reinterpret_cast<const int2 *>(ctx->d_tk_list);
                    b.ntiles = n_in = kept;
                }
                if (n_in > 0) {
                    if (a.tau)
                        switch (ctx->mode) {
                            case PAIR_L1_F64: e = launch_stream<K_L1, 4, true>(ctx, b, stream); break;
                            case PAIR_SAD_U32: e = launch_stream<K_SAD, 4, true>(ctx, b, stream); break;
                            case PAIR_SAD_U16: e = launch_stream<K_SAD16, 4, true>(ctx, b, stream); break;
                            default: e = launch_stream<K_DOT, 4, true>(ctx, b, stream); break;
                        }
                    else
                        switch (ctx->mode) {
                            case PAIR_L1_F64: e = launch_stream<K_L1, 4>(ctx, b, stream); break;
                            case PAIR_SAD_U32: e = launch_stream<K_SAD, 4>(ctx, b, stream); break;
                            case PAIR_SAD_U16: e = launch_stream<K_SAD16, 4>(ctx, b, stream); break;
                            default: e = launch_stream<K_DOT, 4>(ctx, b, stream); break;
                        }
                }
                if (e == hipSuccess && n_mask > 0) {
                    PairArgs c = a;
                    c.i_end = i_full; c.j_end = j_full;
                    c.tiles_i = b.tiles_i; c.tiles_j = b.tiles_j;
                    c.tiles = l_mask;
                    c.ntiles = n_mask;
                    e = launch_one_tile(ctx, c, 128, stream);
                }
                if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "pair kernel launch: %s", hipGetErrorString(e));
                // strips (a phased sweep splits them like everything else: a strip's nodes must not meet every row of the region
                // under their first, loose thresholds)
                int rc = HSD_OK;
                {
                    PairArgs c = a;
                    if (a.sym) {
                        if (i_full < a.i_end) {
                            // rows of the whole tiles x the last columns (mirrored), then the corner square
                            c.sym = 0; c.mirror = a.mirror;
                            c.i_begin = a.i_begin; c.i_end = i_full; c.j_begin = i_full; c.j_end = a.i_end;
                            if ((rc = launch_region(ctx, c, BT, stream, false))) return rc;
                            c.sym = 1;
                            c.i_begin = i_full; c.i_end = a.i_end; c.j_begin = i_full; c.j_end = a.i_end;
                            if ((rc = launch_region(ctx, c, BT, stream, false))) return rc;
                        }
                    } else {
                        if (j_full < a.j_end) {
                            c.i_begin = a.i_begin; c.i_end = a.i_end; c.j_begin = j_full; c.j_end = a.j_end;
                            if ((rc = launch_region(ctx, c, BT, stream, false))) return rc;
                        }
                        if (i_full < a.i_end) {
                            c.i_begin = i_full; c.i_end = a.i_end; c.j_begin = a.j_begin; c.j_end = j_full;
                            if ((rc = launch_region(ctx, c, BT, stream, false))) return rc;
                        }
                    }
                }
                return HSD_OK;
            }
        }
        // (a list could not be built: fall through to the one-tile kernel over every tile)
    }
    if (BT == 128 && !hsd_hook(H_TILE) && ctx->mode != PAIR_ZP_F64) {
        // few tiles per CU: the small tile's five workgroups per CU overlap their phases better than two large ones
        // (ER N = 10k, raw fp64 W1: 0.321 ms with 64 x 64 tiles against 0.351 ms with 128 x 128)
        const int64_t t128i = (a.i_end - a.i_begin + 127) / 128, t128j = (a.j_end - a.j_begin + 127) / 128;
        if ((a.sym ? t128i * (t128i + 1) / 2 : t128i * t128j) < 16 * (int64_t)ctx->prop.multiProcessorCount) BT = 64;
    }
    a.tiles_i = (int)((a.i_end - a.i_begin + BT - 1) / BT);
    a.tiles_j = (int)((a.j_end - a.j_begin + BT - 1) / BT);
    if (a.tiles_i <= 0 || a.tiles_j <= 0) return HSD_OK;
    const int64_t nb = a.sym ? (int64_t)a.tiles_i * (a.tiles_i + 1) / 2 : (int64_t)a.tiles_i * a.tiles_j;
    if (nb > 0x7fffffff) return hsd_fail(ctx, HSD_ERR_TOO_LARGE, "too many tiles (%lld)", (long long)nb);
    a.ntiles = (int)nb;
    a.tiles = tile_order(ctx, a.tiles_i, a.tiles_j, a.sym, 0, false, false, nullptr);
    if (a.n_phases > 1) {
        int cnt = -1;
        a.tiles = tile_order(ctx, a.tiles_i, a.tiles_j, a.sym, 0, false, false, &cnt, a.phase, a.n_phases, a.phase_w);
        if (cnt < 0) return hsd_fail(ctx, HSD_ERR_HIP, "tile list of a sweep phase could not be allocated");
        a.ntiles = cnt;
        if (cnt == 0) return HSD_OK;
    }
    hipError_t e = launch_one_tile(ctx, a, BT, stream);
    if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "pair kernel launch: %s", hipGetErrorString(e));
    return HSD_OK;
}

// Tile order of a ROW-BLOCK launch: nI row tiles x (nA + nI + nB) column tiles -- the rectangle left of the block's diagonal square, the
// square, the rectangle right of it.  Of the square only the tiles on or above its diagonal are listed, those above it flagged
// (TILE_MIRROR_FLAG in the row index) to be written mirrored as well; the rest as tile_order does: S x S groups dealt round-robin to the XCDs.
const int2 *rowblock_tile_order(hsd_ctx *ctx, int nI, int nA, int nB, int *count) {
    const char *e = hsd_hook(H_SUPERTILE);
    const int G = std::max(e ? atoi(e) : 8, 1);
    const char *ed = hsd_hook(H_XCD_DEAL);
    const int deal = (ed && atoi(ed) == 0) ? 0 : 1;
    const int nJ = nA + nI + nB;
    const int sym_key = 2 | (nA << 2), key = (G & 0xff) | (deal << 25);
    for (size_t k = 0; k < ctx->tile_lists.size(); ++k) {
        const auto &c = ctx->tile_lists[k];
        if (c.tiles_i == nI && c.tiles_j == nJ && c.sym == sym_key && c.group == key) {
            const hsd_ctx::TileList hit = c;
            ctx->tile_lists.erase(ctx->tile_lists.begin() + (long)k);
            ctx->tile_lists.push_back(hit);
            *count = hit.count;
            return reinterpret_cast<const int2 *>(hit.d);
        }
    }
    std::vector<int2> order;
    order.reserve((size_t)nI * (nA + nB) + (size_t)nI * (nI + 1) / 2);
    std::vector<size_t> group_begin;
    for (int sr = 0; sr < nI; sr += G)
        for (int sc = 0; sc < nJ; sc += G) {
            const size_t at = order.size();
            for (int ti = sr; ti < std::min(sr + G, nI); ++ti)
                for (int tj = sc; tj < std::min(sc + G, nJ); ++tj) {
                    const int ts = tj - nA;                           // position inside the square, if there
                    if (ts >= 0 && ts < nI) {
                        if (ts < ti) continue;                        // written by the mirror of (ts, ti)
                        order.push_back(make_int2(ti | (ts > ti ? TILE_MIRROR_FLAG : 0), tj));
                    } else order.push_back(make_int2(ti, tj));
                }
            if (order.size() > at) group_begin.push_back(at);
        }
    group_begin.push_back(order.size());
    if (deal && group_begin.size() > 9) {
        std::vector<int2> dealt;
        dealt.reserve(order.size());
        const size_t ng = group_begin.size() - 1;
        for (size_t x = 0; x < 8; ++x)
            for (size_t g = x; g < ng; g += 8) dealt.insert(dealt.end(), order.begin() + (long)group_begin[g], order.begin() + (long)group_begin[g + 1]);
        order.swap(dealt);
    }
    *count = (int)order.size();
    void *d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(order.size(), 1) * sizeof(int2)) != hipSuccess) { *count = -1; return nullptr; }
    if (!order.empty() && hipMemcpy(d, order.data(), order.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        *count = -1;
        return nullptr;
    }
    if (ctx->tile_lists.size() >= 256) {
        (void)hipFree(ctx->tile_lists.front().d);
        ctx->tile_lists.erase(ctx->tile_lists.begin());
    }
    ctx->tile_lists.push_back({nI, nJ, sym_key, key, d, (int)order.size()});
    return reinterpret_cast<const int2 *>(d);
}

// A row block [ra, re) x [0, n) of a Wasserstein matrix (a multi-GPU rank's share) as ONE persistent launch: the diagonal square
// (tiles on / above its diagonal, mirrored) and the rectangles left and right of it in one tile list over one column-side granule
// list [0, ra) + [ra, i_full) + [i_full, ...).  Three launches side by side (round 3) each kept their own 512 resident workgroups
// and their own tail: at BA N = 100k, P = 8 a rank's block took 3.3 - 4.9 ms of zero-pad distance where its arithmetic is 2.6 ms, and
// raw fp64 W1 52 - 62 ms against 50.6 (tools_dev/strong_share.py).  What is not a whole tile (< 128 rows at the end of the block,
// < 128 columns at the end of the matrix) goes to the one-tile kernel as before.  *done = false: not applicable, nothing was launched.
int launch_rowblock(hsd_ctx *ctx, const PairArgs &base, int64_t ra, int64_t re, bool *done) {
    *done = false;
    const int64_t n = ctx->n;
    const bool wass = ctx->mode == PAIR_L1_F64 || ctx->mode == PAIR_SAD_U32 || ctx->mode == PAIR_SAD_U16;
    { const char *e = hsd_hook(H_ROWBLOCK); if (e && atoi(e) == 0) return HSD_OK; }       // A/B and tests: the three-launch form
    if (!wass || ra % 128 != 0 || (ra == 0 && re == n)) return HSD_OK;
    const int n_aux = ctx->n_seg + (base.point ? 1 : 0);
    if (n_aux > 3) return HSD_OK;
    const int64_t i_full = ra + (re - ra) / 128 * 128, Lb = i_full + (n - i_full) / 128 * 128;
    if (i_full == ra) return HSD_OK;
    PairArgs b = base;
    b.i_begin = ra; b.i_end = i_full; b.j_begin = 0; b.j_end = Lb;
    b.sym = 0; b.mirror = 2; b.swap = 0;
    if (!stream_waves(ctx, b)) return HSD_OK;
    hsd_ctx::GranList LI{}, LJ{};
    const int64_t rj[6] = {0, ra, ra, i_full, i_full, Lb};
    if (!gran_list(ctx, ra, i_full, ctx->stream, &LI) || !gran_list_ranges(ctx, rj, ctx->stream, &LJ)) return HSD_OK;
    const int nI = LI.ntiles, nA = (int)(ra / 128), nB = (int)((Lb - i_full) / 128);
    if (LJ.ntiles != nA + nI + nB) return HSD_OK;
    int cnt = -1;
    const int2 *order = rowblock_tile_order(ctx, nI, nA, nB, &cnt);
    if (!order || cnt <= 0) return HSD_OK;
    b.tiles_i = nI; b.tiles_j = LJ.ntiles;
    b.tiles = order; b.ntiles = cnt;
    b.tile_ctr = ctx->d_tile_ctr;
    b.n_aux = n_aux;
    b.gl_i = LI.d_gl; b.gl_j = LJ.d_gl;
    b.tmax_i = LI.d_tmax; b.tmax_j = LJ.d_tmax;
    b.ntl_i = LI.ntiles; b.ntl_j = LJ.ntiles;
    b.suf = ctx->d_suf;
    b.stamps = nullptr;
    hipError_t e;
    switch (ctx->mode) {
        case PAIR_L1_F64: e = launch_stream<K_L1, 4>(ctx, b, ctx->stream); break;
        case PAIR_SAD_U32: e = launch_stream<K_SAD, 4>(ctx, b, ctx->stream); break;
        default: e = launch_stream<K_SAD16, 4>(ctx, b, ctx->stream); break;
    }
    if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "pair kernel launch: %s", hipGetErrorString(e));
    *done = true;
    // strips for the one-tile kernel (natural order, masked stores): the columns behind the last whole tile, then the rows behind
    // the block's last whole tile against every column (both orientations of a pair give the same bits: no mirroring needed)
    PairArgs c = base;
    c.sym = 0; c.mirror = 0; c.swap = 0;
    const int BT = tile_edge(ctx);
    int rc;
    if (Lb < n) {
        c.i_begin = ra; c.i_end = i_full; c.j_begin = Lb; c.j_end = n;
        if ((rc = launch_region(ctx, c, BT, ctx->stream, false))) return rc;
    }
    if (i_full < re) {
        c.i_begin = i_full; c.i_end = re; c.j_begin = 0; c.j_end = n;
        if ((rc = launch_region(ctx, c, BT, ctx->stream, false))) return rc;
    }
    return HSD_OK;
}

// The square block [ra, re)^2 on the diagonal: tiles on or above it, each also written mirrored.
// (A split into whole-round 128-tiles plus a corner of 64-tiles on side streams was measured and
// dropped: workgroups retire unevenly, so there is no "last round" to fill -- N = 11520 (7.998
// rounds of 512 workgroups) and N = 11648 (8.18 rounds) run at the same VALU utilisation.)
int launch_square(hsd_ctx *ctx, const PairArgs &base, int64_t ra, int64_t re) {
    PairArgs b = base;
    b.i_begin = ra; b.i_end = re; b.j_begin = ra; b.j_end = re;
    b.sym = 1; b.mirror = 1; b.swap = 0;
    return launch_region(ctx, b, tile_edge(ctx), ctx->stream);
}

// per-column maximum of a uint32 bin-major matrix (one workgroup per column)
__global__ __launch_bounds__(256) void colmax_u32_kernel(const uint32_t *__restrict__ X, int64_t npad, uint32_t *__restrict__ out) {
    __shared__ uint32_t red[256];
    const uint32_t *col = X + (int64_t)blockIdx.x * npad;
    uint32_t m = 0;
    for (int64_t i = threadIdx.x; i < npad; i += 256) m = max(m, col[i]);
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
    for (int c = 0; 2 * c < V; ++c) {
        uint32_t w[2] = {0u, 0u};
        for (int h = 0; h < 2; ++h) {
            const int v = 2 * c + h;
            if (v < V) w[h] = X32[((int64_t)s * stride32 + vsrc[v]) * npad + i] / vdiv[v] * vmul[v];
        }
        X16[((int64_t)s * stride16 + c) * npad + i] = w[0] | (w[1] << 16);
    }
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

__global__ void fill_f64_kernel(double *__restrict__ p, int64_t n, double v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
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
        WaveList<8> wl;
        wl.load(cv, ci, c, lane);
        for (int q = 0; q < k; ++q) {
            wl.pop(last_v, last_i);
            if (lane == 0) { out_idx[lr * (int64_t)k + q] = last_i; out_val[lr * (int64_t)k + q] = last_v; }
        }
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
        WaveList<8> wl;
        wl.load(cv, ci, c, lane);
        for (int q = 0; q < k; ++q) {
            wl.pop(last_v, last_i);
            if (lane == 0) { sel_v[q] = last_v; sel_i[q] = last_i; }
        }
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

}  // namespace

// ------------------------------------------------------------------------------------------
namespace {

// device temporary of hsd_pair_prepare: released on every exit path
struct DevTmp {
    void *p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    template <typename T> T *as() const { return static_cast<T *>(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    void *release() { void *q = p; p = nullptr; return q; }
};

#define HSD_TMP_ALLOC(ctx, tmp, bytes) HSD_HIP(ctx, (tmp).alloc(bytes))

int prepare_w1_raw(hsd_ctx *ctx, const PrepArgs &p, dim3 grid, dim3 block, int nl_hist) {
    // scipy on the raw samples: fp64 columns G = CDF x bin width, one block of B - 1 columns per layer >= 1
    const int cpl = p.B - 1;
    constexpr int CH = chunk_cols(K_L1);
    ctx->dcols_alg = nl_hist * cpl;
    ctx->dcols = (int)round_up(std::max(ctx->dcols_alg, 0), CH);
    ctx->elem_bytes = 8;
    const size_t cnt = (size_t)std::max(ctx->dcols, CH) * p.npad;
    HSD_HIP(ctx, hipMalloc(&ctx->d_X, cnt * sizeof(double)));
    HSD_HIP(ctx, hipMemsetAsync(ctx->d_X, 0, cnt * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(prep_l1_kernel, grid, block, 0, ctx->stream, p, (double *)ctx->d_X, 0, cpl, ctx->d_H);
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
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_inv, (size_t)nsg * npad))) return rc;
    {   // padded nodes: inv = 2 (treated as empty rings; never stored) -- filled on the device (no host buffer whose lifetime an
        // asynchronous copy would depend on; seg_begin lives in the context)
        hipLaunchKernelGGL(fill_f64_kernel, dim3((unsigned)(((size_t)nsg * npad + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_inv, (int64_t)nsg * npad, 2.0);
        HSD_HIP(ctx, hipMemcpyAsync(dseg.p, ctx->seg_begin, sizeof(ctx->seg_begin), hipMemcpyHostToDevice, ctx->stream));
        if (use_u32)
            hipLaunchKernelGGL(prep_zp_kernel<uint32_t>, grid, block, 0, ctx->stream, p, x0.as<uint32_t>(), ctx->d_inv, ctx->d_H, dseg.as<int>(), ntot.as<uint32_t>());
        else
            hipLaunchKernelGGL(prep_zp_kernel<double>, grid, block, 0, ctx->stream, p, x0.as<double>(), ctx->d_inv, ctx->d_H, dseg.as<int>(), ntot.as<uint32_t>());
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
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_rmax, (size_t)nl_hist * ngran))) return rc;
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
            hipLaunchKernelGGL(count_above_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nl_hist), block, 0, ctx->stream, p, dcand.as<int>(), NC,
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
            auto plan_for = [&](const std::vector<int32_t> &ord, std::vector<LayerPlan> &plan) -> double {
                double total = 0.0;
                const int64_t nt = (G16 + 7) / 8;
                std::vector<uint32_t> tm((size_t)nt);             // (per call: candidate orders are evaluated on several host threads)
                for (int sgi = 0; sgi < nl_hist; ++sgi) {
                    LayerPlan bp{-1, 0, 0.97 * bin_words};
                    for (int q = 0; q < NC; ++q) {
                        const int low_w = vfirst[(size_t)cand[(size_t)q]] / (pack ? 2 : 1), hi_w = bin_words - low_w;
                        if (hi_w < 2) continue;
                        const uint32_t *gm = gmax16.data() + ((size_t)sgi * NC + q) * ngran16;
                        uint32_t gmx = 0;
                        for (int64_t t = 0; t < nt; ++t) {
                            uint32_t m = 0;
                            for (int64_t e = 8 * t; e < std::min<int64_t>(G16, 8 * t + 8); ++e) m = std::max(m, gm[ord[(size_t)e]]);
                            tm[(size_t)t] = m;
                            gmx = std::max(gmx, m);
                        }
                        if (n % 16) gmx = std::max(gmx, gm[G16]);        // (the partial last granule: one-tile kernel only)
                        std::sort(tm.begin(), tm.begin() + nt);
                        const int cap = (int)std::min<int64_t>(hi_w - 1, words_of(gmx));
                        // E over unordered tile pairs of f(min): the k-th smallest (0-based) is the minimum of nt - k pairs
                        long double acc = 0.0L;
                        for (int64_t k2 = 0; k2 < nt; ++k2) {
                            int64_t wr = words_of(tm[(size_t)k2]);
                            if (low_w + wr == 0) wr = 1;
                            acc += (long double)(wr <= cap ? wr : hi_w) * (long double)(nt - k2);
                        }
                        const double cost = low_w + (double)(acc / ((long double)nt * (nt + 1) / 2));
                        if (cost < bp.cost) bp = LayerPlan{q, cap, cost};
                    }
                    if (bp.q < 0) bp.cost = bin_words;
                    plan[(size_t)sgi] = bp;
                    total += bp.cost;
                }
                return total;
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
                int thr_cap = ept ? atoi(ept) : 16;
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
                            for (int64_t g = 0; g < G16; ++g) ord[(size_t)g] = (int32_t)g;
                            std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return key[x] < key[y]; });
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
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_rmax16, (size_t)nl_hist * ngran16))) return rc;
        ctx->ngran16 = ngran16;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_suf, (size_t)(suf_rows + 1) * npad))) return rc;
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
        HSD_HIP(ctx, hipMalloc(&ctx->d_X, cntf * sizeof(uint32_t)));
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
            const dim3 rgrid((unsigned)((n + 255) / 256), (unsigned)nl_hist);
            if (pack) hipLaunchKernelGGL(prep_rank_kernel<true>, rgrid, block, 0, ctx->stream, p, dr.as<int>(), dr.as<int>() + nl_hist, dr.as<int>() + 2 * nl_hist, dr.as<int>() + 3 * nl_hist, (uint32_t *)ctx->d_X, ctx->d_suf);
            else hipLaunchKernelGGL(prep_rank_kernel<false>, rgrid, block, 0, ctx->stream, p, dr.as<int>(), dr.as<int>() + nl_hist, dr.as<int>() + 2 * nl_hist, dr.as<int>() + 3 * nl_hist, (uint32_t *)ctx->d_X, ctx->d_suf);
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
            std::vector<int64_t> wrs;
            for (int b8 : cb)
                for (int r16 : cr) {
                    const int fixed = b8 + r16 + (low - b8 + 1) / 2;
                    wrs.clear();
                    for (int64_t t = 0; t < nt; ++t)
                        if ((int)tb8[(size_t)t] <= b8 && (int)tr8[(size_t)t] <= r16) wrs.push_back(words_of(tR[(size_t)t]));
                    std::sort(wrs.begin(), wrs.end());
                    long double save = 0.0L;
                    const int64_t m = (int64_t)wrs.size();
                    for (int64_t k2 = 0; k2 < m; ++k2) {
                        int64_t wr = wrs[(size_t)k2];
                        if (low + wr == 0) wr = 1;
                        if (wr > cap) break;
                        const int64_t c16 = low + wr, c8 = fixed + std::max<int64_t>(wr - r16, 0) / 2 + (std::max<int64_t>(wr - r16, 0) & 1);
                        if (c8 >= 1 && c8 < c16) save += (long double)(c16 - c8) * (long double)(m - k2);
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
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_seg_tab, (size_t)nsg))) return rc;
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
    const int half = (int)round_up((int64_t)nl * B, CH);
    ctx->dcols = sym ? 2 * half : half;
    ctx->dcols_alg = (sym ? 2 : 1) * nl * B;
    ctx->elem_bytes = 8;
    const size_t cnt = (size_t)ctx->dcols * npad;
    HSD_HIP(ctx, hipMalloc(&ctx->d_X, cnt * sizeof(double)));
    HSD_HIP(ctx, hipMalloc(&ctx->d_Y, cnt * sizeof(double)));
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
    hipLaunchKernelGGL(prep_kl_kernel, grid, block, 0, ctx->stream, p, kl_alpha, sym, half, (double *)ctx->d_X,
                       (double *)ctx->d_Y, hpart.as<double>());
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
    PrepArgs p{ctx->d_counts, ctx->d_support, ctx->d_node_bin, n, npad, L, B, layer_lo, layer_hi, ctx->d_flag};
    const int nl = layer_hi - layer_lo + 1;
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)nl), block(256);
    int rc;
    // d_H: per-node entropy term (KL) or the ring-0 point values (W1 with layer 0)
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_H, (size_t)2 * npad))) return rc;
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

extern "C" int hsd_pair_rows(hsd_ctx *ctx, int64_t r0, int64_t r1, double *out, int64_t ld, int out_on_device) try {
    HSD_CHECK_CTX(ctx);
    if (ctx->mode == PAIR_NONE) return hsd_fail(ctx, HSD_ERR_NO_SIGNATURE, "hsd_pair_prepare has not been called");
    const int64_t n = ctx->n;
    if (r0 < 0 || r1 > n || r0 > r1) return hsd_fail(ctx, HSD_ERR_INVALID, "row range out of bounds");
    if (r0 == r1) return HSD_OK;
    if (!out || ld < n) return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_pair_rows: null out or ld < n");
    const bool wass = ctx->mode == PAIR_L1_F64 || ctx->mode == PAIR_SAD_U32 || ctx->mode == PAIR_SAD_U16 || ctx->mode == PAIR_ZP_F64;

    PairArgs a{};
    a.X = ctx->d_X;
    a.Y = wass ? ctx->d_X : ctx->d_Y;
    a.H = wass ? nullptr : ctx->d_H;
    a.inv = ctx->d_inv;
    a.point = ctx->has_point ? (wass ? ctx->d_H : ctx->d_H + ctx->npad) : nullptr;
    a.point_c = ctx->kl_point_c;
    a.npad = ctx->npad;
    a.dcols = ctx->dcols;
    a.n_seg = ctx->n_seg;
    a.seg_tab = reinterpret_cast<const SegDesc *>(ctx->d_seg_tab);
    a.rmax = ctx->d_rmax;
    a.ngran = ctx->ngran;
    a.list_cap = ctx->list_cap;
    a.work = ctx->d_work;
    a.hmode = ctx->mode == PAIR_SKL ? 1 : 0;
    { const char *e = hsd_hook(H_DEBUG); a.dbg = e ? atoi(e) : 0; }
    a.dcols_valid = (ctx->mode == PAIR_SKL) ? ctx->dcols : ctx->dcols_alg;
    HSD_HIP(ctx, hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned long long), ctx->stream));

    auto run = [&](double *dout, int64_t dld, int64_t rb, int64_t ra, int64_t re) -> int {
        // rows [ra, re) of the matrix; dout row 0 <-> node rb
        PairArgs b = a;
        b.out = dout;
        b.ld = dld;
        b.rbase = rb;
        b.i_begin = ra; b.i_end = re;
        int rc;
        // The diagonal square (symmetric) and the rectangles left and right of it write disjoint parts of the
        // row block: a multi-GPU rank has all three, and they run side by side (two side streams forked from
        // and joined back into the context's stream) so that their tails overlap.  HSD_SIDE_STREAMS=0: in sequence.
        const char *es = hsd_hook(H_SIDE_STREAMS);
        bool fork = (ra > 0 || re < n) && !(es && atoi(es) == 0);
        if (fork && !ctx->ev_fork) {
            bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess;
            for (int t = 0; t < 2 && ok; ++t)
                ok = hipStreamCreateWithFlags(&ctx->side[t], hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&ctx->ev_join[t], hipEventDisableTiming) == hipSuccess;
            if (!ok) return hsd_fail(ctx, HSD_ERR_HIP, "side streams: %s", hipGetErrorString(hipGetLastError()));
        }
        if (ra > 0 || re < n) {
            bool done = false;
            if ((rc = launch_rowblock(ctx, b, ra, re, &done))) return rc;
            if (done) return HSD_OK;
        }
        if (fork) {
            // the granule lists of the three regions are built on the context's stream BEFORE the fork, so that a side stream
            // never meets a list whose tile maxima are still being computed
            PairArgs probe = b;
            hsd_ctx::GranList tmp{};
            const auto prefetch = [&](int64_t ib, int64_t ie, int64_t jb, int64_t je, int sym) {
                probe.i_begin = ib; probe.i_end = ie; probe.j_begin = jb; probe.j_end = je; probe.sym = sym;
                if (ie <= ib || je <= jb || !stream_waves(ctx, probe)) return;
                (void)gran_list(ctx, ib, ib + (ie - ib) / 128 * 128, ctx->stream, &tmp);
                if (!sym) (void)gran_list(ctx, jb, jb + (je - jb) / 128 * 128, ctx->stream, &tmp);
            };
            prefetch(ra, re, ra, re, 1);
            prefetch(ra, re, 0, ra, 0);
            prefetch(ra, re, re, n, 0);
            HSD_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        }
        if ((rc = launch_square(ctx, b, ra, re))) return rc;
        b.sym = 0; b.mirror = 0;
        const int BT = tile_edge(ctx);
        auto rect = [&](int t, int64_t jb, int64_t je, int swap) -> int {
            b.j_begin = jb; b.j_end = je; b.swap = swap;
            b.ctr_slot = t + 1;
            if (!fork) return launch_region(ctx, b, BT, ctx->stream);
            HSD_HIP(ctx, hipStreamWaitEvent(ctx->side[t], ctx->ev_fork, 0));
            const int r = launch_region(ctx, b, BT, ctx->side[t]);
            if (r) return r;
            HSD_HIP(ctx, hipEventRecord(ctx->ev_join[t], ctx->side[t]));
            HSD_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[t], 0));
            return HSD_OK;
        };
        if (ra > 0 && (rc = rect(0, 0, ra, (ctx->mode == PAIR_KL) ? 1 : 0))) return rc;      // below the diagonal (i > j)
        if (re < n && (rc = rect(1, re, n, 0))) return rc;                                    // above the diagonal (i < j)
        return HSD_OK;
    };

    if (out_on_device) return run(out, ld, r0, r0, r1);

    // host output: compute row blocks into a device staging buffer (<= 1 GiB) and copy out.  The staging rows are
    // padded to whole 128-byte lines (full-line stores in the kernel); the 2-D copy drops the padding.
    const int64_t spitch = round_up(n, 16);
    const int64_t max_rows = std::max<int64_t>(HSD_TILE, ((int64_t)1 << 30) / (spitch * (int64_t)sizeof(double)) / HSD_TILE * HSD_TILE);
    const int64_t blk = std::min<int64_t>(round_up(r1 - r0, 1), max_rows);
    const size_t need = (size_t)blk * spitch * sizeof(double);
    if (ctx->stage_bytes < need) {
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hsd_dev_free(ctx->d_stage);
        int rc = hsd_dev_alloc(ctx, &ctx->d_stage, need / sizeof(double));
        if (rc) return rc;
        ctx->stage_bytes = need;
    }
    for (int64_t ra = r0; ra < r1; ra += blk) {
        const int64_t re = std::min(r1, ra + blk);
        int rc = run(ctx->d_stage, spitch, ra, ra, re);
        if (rc) return rc;
        HSD_HIP(ctx, hipMemcpy2DAsync(out + (ra - r0) * ld, (size_t)ld * sizeof(double), ctx->d_stage, (size_t)spitch * sizeof(double),
                                      (size_t)n * sizeof(double), (size_t)(re - ra), hipMemcpyDeviceToHost, ctx->stream));
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

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
namespace {

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
__global__ void tk_gather_u32_kernel(const uint32_t *__restrict__ X, int64_t npad, const int32_t *__restrict__ orig, uint32_t *__restrict__ Xp) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npad) return;
    const int32_t o = orig[p];
    const int64_t c = blockIdx.y;
    Xp[c * npad + p] = o >= 0 ? X[c * npad + o] : 0u;
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

int tk_prepare(hsd_ctx *ctx) {
    if (ctx->tk_ready) return HSD_OK;
    hsd_tk_release(ctx);                               // (whatever an earlier, failed attempt allocated)
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
    const int nt = (int)(npad / 128);
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
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_orig, (size_t)npad))) return rc;
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_lo, (size_t)nf * nt))) return rc;
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_hi, (size_t)nf * nt))) return rc;
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_scale, (size_t)nf * nt))) return rc;
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
        HSD_HIP(ctx, hipMemsetAsync(ctx->d_tk_orig, 0xff, (size_t)npad * sizeof(int32_t), ctx->stream));          // padding positions: -1
        if ((rc = hsd_tk_order_device(ctx, Fo, n, nf, ctx->d_tk_orig))) return rc;
        mark("k-d order (device)");
        hipLaunchKernelGGL(tk_tile_interval_kernel, dim3((unsigned)nt, (unsigned)nf), dim3(128), 0, ctx->stream, dV.as<double>(), seg ? (const double *)ctx->d_inv : nullptr,
                           npad, n, seg ? nlay : 0, (const int32_t *)ctx->d_tk_orig, nt, ctx->d_tk_lo, ctx->d_tk_hi, ctx->d_tk_scale);
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));           // (dFo goes)
    } else {
        std::vector<int32_t> idx;
        {
            std::vector<double> Fo = V;
            if (seg) for (size_t q = 0; q < Fo.size(); ++q) Fo[q] *= S[q];
            kd_order_host(Fo, n, nf, idx);
        }
        mark("k-d order (host)");
        std::vector<int32_t> orig((size_t)npad, -1);
        std::copy(idx.begin(), idx.end(), orig.begin());
        std::vector<double> flo((size_t)nf * nt, INFINITY), fhi((size_t)nf * nt, -INFINITY), fsc((size_t)nf * nt, INFINITY);
        for (int64_t p = 0; p < n; ++p)
            for (int f = 0; f < nf; ++f) {
                const size_t src = (size_t)f * n + idx[(size_t)p], dst = (size_t)f * nt + p / 128;
                flo[dst] = std::min(flo[dst], V[src]); fhi[dst] = std::max(fhi[dst], V[src]);
                fsc[dst] = std::min(fsc[dst], seg ? S[src] : 1.0);
            }
        HSD_HIP(ctx, hipMemcpy(ctx->d_tk_orig, orig.data(), orig.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HSD_HIP(ctx, hipMemcpy(ctx->d_tk_lo, flo.data(), flo.size() * sizeof(double), hipMemcpyHostToDevice));
        HSD_HIP(ctx, hipMemcpy(ctx->d_tk_hi, fhi.data(), fhi.size() * sizeof(double), hipMemcpyHostToDevice));
        HSD_HIP(ctx, hipMemcpy(ctx->d_tk_scale, fsc.data(), fsc.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const int band_w = std::max(1, std::min(4, hsd_hook_int(H_TOPK_BAND, 1)));      // the first launch: every tile with itself and its band_w successors
    std::vector<int2> band;
    for (int t = 0; t < nt; ++t)
        for (int d = 0; d <= band_w && t + d < nt; ++d) band.push_back(make_int2(t, t + d));
    ctx->tk_band_w = band_w;
    mark("tile intervals + first-launch list");
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_H, (size_t)npad))) return rc;
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_tmax, (size_t)nt))) return rc;
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_count, (size_t)4))) return rc;
    ctx->tk_list_cap = (int)std::min<int64_t>((int64_t)nt * (nt + 1) / 2, 0x7fffffff);
    HSD_HIP(ctx, hipMalloc(&ctx->d_tk_list, (size_t)ctx->tk_list_cap * sizeof(int2)));
    HSD_HIP(ctx, hipMalloc(&ctx->d_tk_band, band.size() * sizeof(int2)));
    ctx->tk_band_n = (int)band.size();
    HSD_HIP(ctx, hipMemcpy(ctx->d_tk_band, band.data(), band.size() * sizeof(int2), hipMemcpyHostToDevice));
    const dim3 pgrid((unsigned)((npad + 255) / 256));
    if (ctx->has_point)
        hipLaunchKernelGGL(tk_gather_kernel, dim3(pgrid.x, 1u), dim3(256), 0, ctx->stream, (const double *)ctx->d_H, npad, ctx->d_tk_orig, ctx->d_tk_H);
    if (!seg) {
        const int cols = std::max(ctx->dcols, 16);
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_X, (size_t)cols * npad))) return rc;
        hipLaunchKernelGGL(tk_gather_kernel, dim3(pgrid.x, (unsigned)cols), dim3(256), 0, ctx->stream, (const double *)ctx->d_X, npad, ctx->d_tk_orig, ctx->d_tk_X);
    } else {
        // the zero-pad kinds' operands: packed words, 1 / ring size per layer, suffix sums -- and, recovered from them, what the tile planner reads:
        // members above each layer's threshold per position -> maxima per 16-position granule (+ the "too wide for the uint8 range" flags) -> per tile
        const int cols = std::max(ctx->dcols, 32), ngran16 = (int)(npad / 16);
        const int64_t sufr = (int64_t)ctx->suf_rows + 1;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_U, (size_t)cols * npad))) return rc;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_inv, (size_t)nlay * npad))) return rc;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_suf, (size_t)sufr * npad))) return rc;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_rmax16, (size_t)nlay * ngran16))) return rc;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_gl, (size_t)ngran16))) return rc;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_tk_tmaxs, (size_t)nlay * nt))) return rc;
        hipLaunchKernelGGL(tk_gather_u32_kernel, dim3(pgrid.x, (unsigned)cols), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_X, npad, ctx->d_tk_orig, ctx->d_tk_U);
        hipLaunchKernelGGL(tk_gather_kernel, dim3(pgrid.x, (unsigned)nlay), dim3(256), 0, ctx->stream, (const double *)ctx->d_inv, npad, ctx->d_tk_orig, ctx->d_tk_inv);
        if (ctx->d_suf) hipLaunchKernelGGL(tk_gather_u32_kernel, dim3(pgrid.x, (unsigned)sufr), dim3(256), 0, ctx->stream, ctx->d_suf, npad, ctx->d_tk_orig, ctx->d_tk_suf);
        else HSD_HIP(ctx, hipMemsetAsync(ctx->d_tk_suf, 0, (size_t)sufr * npad * sizeof(uint32_t), ctx->stream));
        DevTmp dR, dneed;
        HSD_TMP_ALLOC(ctx, dR, (size_t)nlay * npad * sizeof(uint32_t));
        HSD_TMP_ALLOC(ctx, dneed, (size_t)2 * npad * sizeof(uint32_t));
        if (pack) hipLaunchKernelGGL(tk_rnode_kernel<true>, dim3(pgrid.x, (unsigned)nlay), dim3(256), 0, ctx->stream, ctx->d_tk_U, ctx->d_tk_suf, npad, ctx->d_seg_tab, dR.as<uint32_t>());
        else hipLaunchKernelGGL(tk_rnode_kernel<false>, dim3(pgrid.x, (unsigned)nlay), dim3(256), 0, ctx->stream, ctx->d_tk_U, ctx->d_tk_suf, npad, ctx->d_seg_tab, dR.as<uint32_t>());
        hipLaunchKernelGGL(granule16_max_kernel, dim3((unsigned)((ngran16 + 255) / 256), (unsigned)nlay), dim3(256), 0, ctx->stream, dR.as<uint32_t>(), npad, ngran16, ctx->d_tk_rmax16);
        for (int s2 = 0; s2 < nlay && pack; ++s2) {
            const HsdSegDesc &d = ctx->seg_tab[s2];
            if (d.r8_col0 < 0) continue;
            HSD_HIP(ctx, hipMemsetAsync(dneed.p, 0, (size_t)2 * npad * sizeof(uint32_t), ctx->stream));
            hipLaunchKernelGGL(u8_need_kernel, dim3(pgrid.x), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_tk_U, npad, npad, d, dneed.as<uint32_t>());
            hipLaunchKernelGGL(tk_wide_kernel, dim3((unsigned)((ngran16 + 255) / 256)), dim3(256), 0, ctx->stream, dneed.as<uint32_t>(), npad, ngran16, d.r8_k16 - d.r8_r16, d.r8_r16,
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
    ctx->tk_ready = true;
    return HSD_OK;
}

// *done = false: not applicable (or a list overflowed): the caller runs the unpruned sweep
int topk_pruned(hsd_ctx *ctx, int k, int32_t *idx_out, double *dist_out, bool *done) {
    *done = false;
    const int64_t n = ctx->n, npad = ctx->npad;
    const bool seg = ctx->mode == PAIR_SAD_U16 || ctx->mode == PAIR_SAD_U32;
    PairArgs a{};
    a.npad = npad;
    a.dcols = ctx->dcols;
    a.n_seg = seg ? ctx->n_seg : 0;
    a.list_cap = ctx->list_cap;
    a.dcols_valid = ctx->dcols_alg;
    { const char *e = hsd_hook(H_TOPK_DEBUG); a.dbg = e ? atoi(e) : 0; }
    a.rbase = 0;
    a.i_begin = 0; a.i_end = npad; a.j_begin = 0; a.j_end = npad;
    a.sym = 1; a.mirror = 1;
    {
        // the persistent kernel must take the region (its LDS budget: at most three per-node vectors)
        if (a.n_seg + (ctx->has_point ? 1 : 0) > 3) return HSD_OK;
        PairArgs probe = a;
        probe.tau = reinterpret_cast<const double *>(ctx);          // (any non-null value: the k-nearest kernels have no output alignment to check)
        if (!stream_waves(ctx, probe)) return HSD_OK;
    }
    int rc = tk_prepare(ctx);
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
    }
    a.point = ctx->has_point ? ctx->d_tk_H : nullptr;
    a.orig_of = ctx->d_tk_orig;
    const int64_t rows = npad;
    const int64_t cap = std::max<int64_t>(2048, 512 + 8 * (int64_t)k);
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t oTau = carve((size_t)rows * sizeof(double)), oCnt = carve((size_t)rows * sizeof(int)), oIdx = carve((size_t)rows * cap * sizeof(int));
    const size_t oVal = carve((size_t)rows * cap * sizeof(double)), oOi = carve((size_t)rows * k * sizeof(int)), oOv = carve((size_t)rows * k * sizeof(double));
    const size_t oT16 = carve((size_t)rows * sizeof(uint16_t) + 64), oFi = carve((size_t)n * k * sizeof(int)), oFv = carve((size_t)n * k * sizeof(double));
    const int64_t M = std::min<int64_t>(n, 1024), ldS = (M + 127) / 128 * 128;
    const size_t oS = carve((size_t)n * ldS * sizeof(double)), oTn = carve((size_t)n * sizeof(double));
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
    if (!seg) hipLaunchKernelGGL(fill_f64_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, d_tau, rows, (double)INFINITY);
    else {
        double *d_S = reinterpret_cast<double *>(base + oS), *d_tauN = reinterpret_cast<double *>(base + oTn);
        PairArgs nat{};
        nat.X = ctx->d_X; nat.Y = ctx->d_X;
        nat.inv = ctx->d_inv;
        nat.point = ctx->has_point ? ctx->d_H : nullptr;
        nat.npad = npad; nat.dcols = ctx->dcols; nat.n_seg = ctx->n_seg;
        nat.seg_tab = reinterpret_cast<const SegDesc *>(ctx->d_seg_tab);
        nat.rmax = ctx->d_rmax; nat.ngran = ctx->ngran; nat.list_cap = ctx->list_cap;
        nat.dcols_valid = ctx->dcols_alg;
        nat.rbase = 0; nat.i_begin = 0; nat.i_end = n;
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
                hipLaunchKernelGGL(fill_columns_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_S, ldS, n, (int)(t * 128 + (sub.j_end - sub.j_begin)),
                                   (int)((t + 1) * 128), (double)INFINITY);
        }
        hipLaunchKernelGGL(topk_threshold_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, d_S, ldS, n_sub * 128, (int64_t)0, n, k, 0.0, d_tauN);
        hipLaunchKernelGGL(tk_permute_tau_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_tk_orig, rows, d_tauN, d_tau);
    }
    PairArgs b = a;
    b.tau = d_tau; b.cand_cnt = d_cnt; b.cand_idx = d_idx; b.cand_val = d_val; b.cand_cap = (int)cap;
    b.tau16 = reinterpret_cast<const uint16_t *>(base + oT16);
    const int BT = tile_edge(ctx);
    const size_t cmp_lds = (size_t)4 * k * 12;
    // launch 0: every row against the tiles next to its own in the order, under thresholds of +inf; then the first real thresholds
    b.explicit_tiles = reinterpret_cast<const int2 *>(ctx->d_tk_band); b.n_explicit = ctx->tk_band_n;
    if ((rc = launch_region(ctx, b, BT, ctx->stream))) return rc;
    hipLaunchKernelGGL(topk_compact_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), cmp_lds, ctx->stream, d_cnt, d_idx, d_val, (int)cap, rows, k, 0.0, d_tau);
    // the rest of the pair space in anti-diagonal phases, each filtered against the thresholds of the moment (tightened after phases 1, 2, 4)
    b.explicit_tiles = nullptr; b.n_explicit = 0;
    b.prune_lo = ctx->d_tk_lo; b.prune_hi = ctx->d_tk_hi; b.prune_scale = seg ? ctx->d_tk_scale : nullptr; b.prune_nf = ctx->tk_nf; b.prune_nt = ctx->tk_nt; b.prune_drop_band = ctx->tk_band_w;
    // (measured at BA N = 100k: 2 phases overflow the lists -- the band thresholds meet half the columns at once --, 3 / 4 / 5 / 6 / 8 / 12 / 16 take
    //  66.3 / 65.7 / 65.6 / 66.8 / 68.8 / 72.1 / 74.5 ms with longest lists of 847 / 690 / 296 / 360 / 709 entries; twice the nodes, twice the appends per phase)
    int n_phases = ctx->tk_nt <= 1200 ? 5 : 8;
    if (const char *ep = hsd_hook(H_TOPK_PHASES)) n_phases = std::max(1, std::min(16, atoi(ep)));
    for (int ph = 0; ph < n_phases; ++ph) {
        b.phase = ph; b.phase_w = 1; b.n_phases = n_phases;
        if ((rc = launch_region(ctx, b, BT, ctx->stream))) return rc;
        if (ph + 1 < n_phases && ((ph + 1) & ph) == 0)
            hipLaunchKernelGGL(topk_compact_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), cmp_lds, ctx->stream, d_cnt, d_idx, d_val, (int)cap, rows, k, 0.0, d_tau);
    }
    // select (the rows of real nodes are the first n positions), back to node order, out
    int flag[4] = {0, 0, 0, 0};
    hipLaunchKernelGGL(topk_select_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, d_cnt, d_idx, d_val, (int)cap, (int64_t)0, n, k, d_oidx, d_oval,
                       ctx->d_flag);
    hipLaunchKernelGGL(tk_unpermute_kernel, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_tk_orig, n, k, d_oidx, d_oval, d_fidx, d_fval);
    HSD_HIP(ctx, hipMemcpyAsync(flag, ctx->d_flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    HSD_HIP(ctx, hipGetLastError());
    if (hsd_hook(H_TOPK_VERBOSE)) {
        std::vector<int> hc((size_t)n);
        (void)hipMemcpy(hc.data(), d_cnt, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
        int64_t mx = 0;
        for (int v : hc) mx = std::max<int64_t>(mx, v);
        fprintf(stderr, "[hsd topk] pruned sweep: capacity %lld, longest list %lld, overflow %d\n", (long long)cap, (long long)mx, flag[0] & 1);
    }
    if (flag[0] & 1) return HSD_OK;                 // a list overflowed (or came up short): the unpruned sweep takes the call
    HSD_HIP(ctx, hipMemcpyAsync(idx_out, d_fidx, (size_t)n * k * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HSD_HIP(ctx, hipMemcpyAsync(dist_out, d_fval, (size_t)n * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
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
        if ((l1 || zp) && r0 == 0 && r1 == n && k <= 128 && !hsd_hook(H_TOPK_GEOM)) {
            bool done = false;
            const int rcp = topk_pruned(ctx, k, idx_out, dist_out, &done);
            if (rcp != HSD_OK || done) return rcp;
        }
    }

    PairArgs a{};
    a.X = ctx->d_X;
    a.Y = wass ? ctx->d_X : ctx->d_Y;
    a.H = wass ? nullptr : ctx->d_H;
    a.inv = ctx->d_inv;
    a.point = ctx->has_point ? (wass ? ctx->d_H : ctx->d_H + ctx->npad) : nullptr;
    a.point_c = ctx->kl_point_c;
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
