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

// (named, not anonymous: tile_edge / stream_waves / launch_region are what pair_internal.hpp declares for the other translation units)
namespace hsd_pair {

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
// One store of a deferred group at a site of its trip (HSD_STORE_SPREAD, pair_stream_kernel): group g < 4 stores tuple d[g], group 4 + k tuple m[k],
// g >= 8 nothing -- chosen by scalar compares and branches INSIDE the statement, so that the compiler sees no control flow in the straight-line trip.
__device__ __attribute__((always_inline)) inline void spread_store_asm(int g, uint32_t v, uint64_t sb, v2f64 d0, v2f64 d1, v2f64 d2, v2f64 d3, v2f64 m0, v2f64 m1,
                                                                       v2f64 m2, v2f64 m3) {
        asm volatile(
            "s_cmp_gt_u32 %[g], 7\n\ts_cbranch_scc1 9f\n\t"
            "s_cmp_gt_u32 %[g], 3\n\ts_cbranch_scc1 4f\n\t"
            "s_cmp_eq_u32 %[g], 0\n\ts_cbranch_scc1 0f\n\t"
            "s_cmp_eq_u32 %[g], 1\n\ts_cbranch_scc1 1f\n\t"
            "s_cmp_eq_u32 %[g], 2\n\ts_cbranch_scc1 2f\n\t"
            "global_store_dwordx4 %[v], %[d3], %[sb] " HSD_STORE_BITS "\n\ts_nop 1\n\ts_branch 9f\n"
            "0:\n\tglobal_store_dwordx4 %[v], %[d0], %[sb] " HSD_STORE_BITS "\n\ts_nop 1\n\ts_branch 9f\n"
            "1:\n\tglobal_store_dwordx4 %[v], %[d1], %[sb] " HSD_STORE_BITS "\n\ts_nop 1\n\ts_branch 9f\n"
            "2:\n\tglobal_store_dwordx4 %[v], %[d2], %[sb] " HSD_STORE_BITS "\n\ts_nop 1\n\ts_branch 9f\n"
            "4:\n\ts_cmp_eq_u32 %[g], 4\n\ts_cbranch_scc1 5f\n\t"
            "s_cmp_eq_u32 %[g], 5\n\ts_cbranch_scc1 6f\n\t"
            "s_cmp_eq_u32 %[g], 6\n\ts_cbranch_scc1 7f\n\t"
            "global_store_dwordx4 %[v], %[m3], %[sb] " HSD_STORE_BITS "\n\ts_nop 1\n\ts_branch 9f\n"
            "5:\n\tglobal_store_dwordx4 %[v], %[m0], %[sb] " HSD_STORE_BITS "\n\ts_nop 1\n\ts_branch 9f\n"
            "6:\n\tglobal_store_dwordx4 %[v], %[m1], %[sb] " HSD_STORE_BITS "\n\ts_nop 1\n\ts_branch 9f\n"
            "7:\n\tglobal_store_dwordx4 %[v], %[m2], %[sb] " HSD_STORE_BITS "\n\ts_nop 1\n"
            "9:\n\ts_nop 0"
            :
            : [g] "s"(g), [v] "v"(v), [sb] "s"(sb),
              [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [m0] "v"(m0), [m1] "v"(m1), [m2] "v"(m2), [m3] "v"(m3)
            : "memory", "scc");
}

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
    // KL kinds, sparse-column form: the tile pair's column list -- ids (uint16, within one half) of the columns occupied on BOTH sides, in the chunk-list
    // area of LDS: buffer `buf` holds up to 4 x list_cap ids, nlist[buf] their number
    const bool ktrim = KIND == K_DOT && a.kmask != nullptr;
    uint16_t *const klist = reinterpret_cast<uint16_t *>(clist);
    const int kcap = 4 * a.list_cap;
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
    int ntiles = a.ntiles;
    if constexpr (TOPK) {
        // (pruned sweep: the launch's tile count comes from the filter kernel that ran just before on the same stream -- a scalar load)
        if (a.ntiles_dev) ntiles = __builtin_amdgcn_readfirstlane(*a.ntiles_dev);
    }
    const int q = ntiles >> 3, rr = ntiles & 7;
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
        if constexpr (KIND == K_DOT) {
            if (ktrim) {
                // sparse-column form: slot `ins` of the stage holds column list[16 chunk + ins] (+ the second half's offset); wave w takes the slots
                // 4 w .. 4 w + 3, whose ids are one 8-byte LDS read
                static_assert(COLS == 1 && CH == 16 && NW == 4, "one fp64 column per instruction, four per wave and chunk");
                const int buf_ = col >> 20, kq = col & 0xfffff;
                const int cnt = __builtin_amdgcn_readfirstlane(nlist[buf_]), nh = (cnt + CH - 1) / CH;
                const bool second = a.khalf && kq >= nh;
                const int kk = second ? kq - nh : kq;
                const unsigned long long four = *reinterpret_cast<const unsigned long long *>(klist + buf_ * kcap + kk * CH + 4 * w);
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)four), hi = __builtin_amdgcn_readfirstlane((uint32_t)(four >> 32));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ins = 4 * w + k;
                    const int id = (int)(((k < 2 ? lo : hi) >> (16 * (k & 1))) & 0xffffu) + (second ? a.khalf : 0);
                    if (ins < valid) {
                        glds16_s(reinterpret_cast<uint64_t>(XI + (int64_t)id * a.npad), offI, dst_lds + (uint32_t)(ins * BT * sizeof(T)));
                        glds16_s(reinterpret_cast<uint64_t>(XJ + (int64_t)id * a.npad), offJ, dst_lds + (uint32_t)((CH * BT + ins * BT) * sizeof(T)));
                        issued += 2;
                    }
                }
                asm volatile("" ::: "memory");
                return issued;
            }
        }
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
        if constexpr (KIND == K_DOT) {
            if (!ktrim || w != 0) return;
            // lane l < kwords: word l of (OR of the I side's eight granules) AND (OR of the J side's); then the set bits as column ids, lane by lane
            // behind an exclusive prefix of the lanes' counts.  (Plain loads: a KL tile runs for ~10^5 cycles, a memory latency here is nothing.)
            const int l = lane();
            unsigned long long mi = 0ull, mj = 0ull;
            if (l < a.kwords) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    mi |= a.kmask[(int64_t)meta[ms * META + e] * a.kwords + l];
                    mj |= a.kmask[(int64_t)meta[ms * META + 8 + e] * a.kwords + l];
                }
            }
            unsigned long long both = mi & mj;
            const int cnt = __popcll(both);
            int pre = cnt;                             // inclusive prefix over the lanes
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(pre, d, 64);
                if (l >= d) pre += o;
            }
            const int total = __shfl(pre, 63, 64);
            int at = pre - cnt;
            uint16_t *out = klist + buf * kcap;
            while (both) {
                const int q = __builtin_ctzll(both);
                both &= both - 1;
                out[at++] = (uint16_t)(64 * l + q);
            }
            if (l == 0) nlist[buf] = total;
        }
    };
    auto planned_chunks = [&](int buf) __attribute__((always_inline)) -> int {
        if constexpr (SEG) return __builtin_amdgcn_readfirstlane(nlist[buf]);
        else if (ktrim) {
            const int nh = (__builtin_amdgcn_readfirstlane(nlist[buf]) + CH - 1) / CH;
            return a.khalf ? 2 * nh : nh;
        } else return (a.dcols_valid + CH - 1) / CH;
    };
    // (sparse-column form: the "column" of a chunk is its place in the list -- buffer << 20 | chunk --, stage_chunk looks the ids up)
    auto chunk_col = [&](int buf, int k) __attribute__((always_inline)) -> int {
        if constexpr (SEG) return __builtin_amdgcn_readfirstlane(clist[buf * a.list_cap + k].x);
        else if (ktrim) return (buf << 20) | k;
        else return k * CH;
    };
    auto chunk_valid = [&](int buf, int k) __attribute__((always_inline)) -> int {
        if constexpr (SEG) return __builtin_amdgcn_readfirstlane(clist[buf * a.list_cap + k].y) & 0xff;
        else if (ktrim) {
            const int cnt = __builtin_amdgcn_readfirstlane(nlist[buf]), nh = (cnt + CH - 1) / CH;
            const int kk = (a.khalf && k >= nh) ? k - nh : k;
            return min(CH, cnt - kk * CH);
        } else return min(CH, a.dcols_valid - k * CH);
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
    unsigned long long cyc_store = 0, cyc_loop = 0, cyc_epi = 0, n_tiles = 0, cyc_wait = 0, cyc_bar = 0;      // HSD_DEBUG bit 3 only
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

#if HSD_STORE_SPREAD
    // EXPERIMENT (round 5): the eight stores of a deferred group leave one behind each WORD of the group's 8-word trip (64 SADs, ~260 cycles, apart)
    // instead of back to back in front of it: a store holds its wave only against the next vector-memory instruction, so spaced like this none
    // waits, and the SIMD's other wave never runs alone.  Rounds 3 and 4 wrote this in C++ three times: the eight-way choice of the accumulator
    // tuple at each site is control flow inside the straight-line trip, and hipcc answered with 464 / 800 / 1 300 spilled registers.  Here the
    // choice is made INSIDE one asm statement per site (scalar compares and branches around eight store instructions): the compiler sees a
    // statement that reads eight tuples which are live anyway, a scalar base and one lane offset.
    // (trip-local state, so that every value is a straight-line result of v_readfirstlane and lives in scalar registers: sp_g the group this trip
    //  carries (8: none), sp_sb its row granule's base, sp_v the lane part of the offsets -- direct: row 2 ly, bytes 16 lx; mirrored: row 2 lx, bytes
    //  16 ly --, sp_G the byte offsets of the four granules of the OTHER side within a matrix row)
    // a wave-uniform value into a scalar register for good (an asm result of class "s": whatever the compiler thought of the value's uniformity)
    auto to_s = [](uint32_t x) __attribute__((always_inline)) -> uint32_t {
        uint32_t r;
        asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "s"(__builtin_amdgcn_readfirstlane(x)));
        return r;
    };
    auto spread_begin = [&](int &sp_g, uint64_t &sp_sb, uint32_t &sp_v, uint32_t (&sp_G)[4]) __attribute__((always_inline)) {
        if (pend_g == 4 && pend_mirror) transpose_acc();
        if (pend_g >= 4 && !pend_mirror) pend_g = 8;
        const int g = pend_g < 8 ? pend_g : 8;        // (8: nothing pending -- the sites then store nothing; the values below are formed all the same)
        // (the tile's granules straight out of the plan ring, at computed positions: two small arrays chosen between by `mir` became a scratch array
        //  under a dynamic index -- three scratch loads with vmcnt(0) at the top of every trip)
        const bool mir = g >= 4;
        const int gg = g & 3;
        const int own = pend_ms * META + (mir ? 8 + 4 * (w & 1) : 4 * (w >> 1)), other = pend_ms * META + (mir ? 4 * (w >> 1) : 8 + 4 * (w & 1));
        const int rowg = __builtin_amdgcn_readfirstlane(meta[own + gg]);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) sp_G[qq] = to_s((uint32_t)(128 * meta[other + qq]));
        sp_sb = (uint64_t)uniform64(reinterpret_cast<int64_t>(a.out + ((int64_t)16 * rowg - a.rbase) * a.ld));
        asm volatile("s_nop 4" : "+s"(sp_sb));         // vector-ALU-written scalar -> address: five wait states
        const int lv = lane(), lyv = lane_y(lv), lxv = lane_x(lv);
        const uint32_t ld8 = (uint32_t)a.ld * 8u;
        sp_v = mir ? (uint32_t)(2 * lxv) * ld8 + (uint32_t)(16 * lyv) : (uint32_t)(2 * lyv) * ld8 + (uint32_t)(16 * lxv);
        sp_g = (int)to_s((uint32_t)g);
    };
    // site s = 4 par + q of the trip: direct group k stores tuple [2 k + par][q], mirrored group 4 + k tuple [2 q + par][k]
    auto spread_site = [&](auto S_, int sp_g, uint64_t sp_sb, uint32_t sp_v, const uint32_t (&sp_G)[4]) __attribute__((always_inline)) {
        constexpr int S = decltype(S_)::value, par = S >> 2, q = S & 3;
        uint64_t sbs = sp_sb + (uint64_t)((par ? (uint32_t)a.ld * 8u : 0u) + sp_G[q]);
        asm volatile("" : "+s"(sbs));                  // (scalar adds of scalar registers: no hazard towards the address use)
        spread_store_asm(sp_g, sp_v, sbs, acc2[0 + par][q], acc2[2 + par][q], acc2[4 + par][q], acc2[6 + par][q], acc2[2 * q + par][0], acc2[2 * q + par][1],
                         acc2[2 * q + par][2], acc2[2 * q + par][3]);
    };
#endif

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
                cvalid = chunk_valid(cur, k);
            }
            work += (unsigned)cvalid;
            if (a.dbg & 2) cvalid = n16 = 0;
            // chunk k has landed once everything older than the operations issued after its loads has retired
            // (the first chunk of a tile was waited for before the previous tile's stores went out)
            // (... and then the barrier at the top of the tile has already told every wave so: no second one)
            if (!(k == 0 && first_landed)) {
#if HSD_STAMPS
                const unsigned long long c_w0 = (a.dbg & 8) ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
                wait_vm_young(__builtin_amdgcn_readfirstlane(vm_seq - (st ? seq_of_stage[1] : seq_of_stage[0])));
#if HSD_STAMPS
                const unsigned long long c_w1 = (a.dbg & 8) ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
                wg_barrier();                         // every wave's part of chunk k has landed, and chunk k - 1 is consumed
#if HSD_STAMPS
                if (a.dbg & 8) { const unsigned long long c_w2 = __builtin_amdgcn_s_memtime(); cyc_wait += c_w1 - c_w0; cyc_bar += c_w2 - c_w1; }
#endif
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
#if HSD_STORE_SPREAD
                int sp_g = 8;
                uint64_t sp_sb = 0;
                uint32_t sp_v = 0, sp_G[4] = {0u, 0u, 0u, 0u};
                if constexpr (DEFERS && UNR == 8) spread_begin(sp_g, sp_sb, sp_v, sp_G);
                auto two_words = [&](auto U_) __attribute__((always_inline)) {
                    constexpr int u = decltype(U_)::value;
                    load_ops(c + u + 1, 1);
                    HSD_PIN_ORDER;
                    compute(0, BYTES);
                    HSD_PIN_ORDER;
                    if constexpr (DEFERS && UNR == 8) spread_site(std::integral_constant<int, u>{}, sp_g, sp_sb, sp_v, sp_G);
                    load_ops(c + u + 2, 0);
                    HSD_PIN_ORDER;
                    compute(1, BYTES);
                    HSD_PIN_ORDER;
                    if constexpr (DEFERS && UNR == 8) spread_site(std::integral_constant<int, u + 1>{}, sp_g, sp_sb, sp_v, sp_G);
                };
                two_words(std::integral_constant<int, 0>{});
                two_words(std::integral_constant<int, 2>{});
                if constexpr (UNR >= 8) {
                    two_words(std::integral_constant<int, 4>{});
                    two_words(std::integral_constant<int, 6>{});
                }
                if constexpr (DEFERS && UNR == 8) {
                    if (sp_g < 8) { vm_seq += 8; pend_g = sp_g + 1; }
                }
#else
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
#endif
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
                            // (a row block's rows are extra positions behind the graph's: the row's own node is one of the columns)
                            if (jo >= 0 && !(a.self_by_orig && jo == a.orig_of[i])) {
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
        unsigned long long *sp = a.stamps + 8 * (size_t)blockIdx.x;
        sp[0] = stamp0;
        sp[1] = __builtin_amdgcn_s_memrealtime();
        sp[2] = cyc_loop; sp[3] = cyc_epi; sp[4] = cyc_store; sp[5] = n_tiles; sp[6] = cyc_wait; sp[7] = cyc_bar;
    }
#endif
}
#undef ACC

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

// LDS of one persistent workgroup: two stages of two 128-node operand blocks (chunk_bytes = bytes of one node's chunk: 128 either
// kind), chunk lists + lengths of two tiles, the three-slot tile-plan ring, suffix rows, the segment descriptions, the per-node vectors
// and suffix sums of two tiles, the k-nearest sweep's thresholds.
constexpr size_t stream_lds_bytes(size_t chunk_bytes, int list_cap, int n_aux, int n_seg, bool topk) {
    return (size_t)2 * 2 * chunk_bytes * 128 + (size_t)2 * list_cap * sizeof(int2) + 16 + 3 * 24 * 4 + 32 + 4 * sizeof(HsdSegDesc) +
           (size_t)2 * n_aux * 256 * sizeof(double) + (size_t)2 * n_seg * 128 * sizeof(uint32_t) + (topk ? 512 : 0);
}
// BUILD-TIME GUARD of the occupancy the design depends on: two workgroups per CU need 2 x LDS <= 160 KiB.  The headline shape -- two zero-pad
// layers + ring 0, i.e. three per-node vectors and two suffix vectors -- must keep it for the matrix kernel AND the k-nearest sweep with
// chunk lists of up to 2 x (ceil(512 words / 32) + 2) + 1 entries (config 4's graph has 303 degree bins = 152 words per layer; round 3 lost
// the second workgroup to 96 bytes of segment description at exactly this point: 87 -> 105-113 ms with every test green).  The fp64
// kinds (raw W1 with three layers + ring 0: four vectors would not fit and take the one-tile kernel; KL with its two vectors) likewise.
static_assert(2 * stream_lds_bytes(128, 2 * (16 + 2) + 1, 3, 2, true) <= 160 * 1024, "k-nearest sweep of the headline shape: ONE workgroup per CU");
static_assert(2 * stream_lds_bytes(128, 2 * (16 + 2) + 1, 3, 2, false) <= 160 * 1024, "matrix kernel of the headline shape: ONE workgroup per CU");
static_assert(2 * stream_lds_bytes(128, 1, 3, 0, true) <= 160 * 1024, "fp64 kinds with three per-node vectors: ONE workgroup per CU");

template <int KIND, int NW, bool TOPK = false>
hipError_t launch_stream(hsd_ctx *ctx, const PairArgs &a, hipStream_t stream) {
    auto kern = pair_stream_kernel<KIND, NW, TOPK>;
    constexpr int CH = stream_chunk_cols(KIND);
    const size_t lds = stream_lds_bytes(CH * sizeof(typename Elem<KIND>::type), a.list_cap, a.n_aux, a.n_seg, TOPK);
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
        if (hipMalloc(reinterpret_cast<void **>(&d_st), (size_t)grid * 64) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemset(d_st, 0, (size_t)grid * 64);
        b.stamps = d_st;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, b);
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h6((size_t)grid * 8), h((size_t)grid * 2);
        (void)hipMemcpy(h6.data(), d_st, h6.size() * 8, hipMemcpyDeviceToHost);
        for (int g = 0; g < grid; ++g) { h[2 * g] = h6[8 * g]; h[2 * g + 1] = h6[8 * g + 1]; }
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
        double cl = 0, ce = 0, cs = 0, nt = 0, cw = 0, cb = 0;
        for (int g = 0; g < grid; ++g) {
            cl += (double)h6[8 * g + 2]; ce += (double)h6[8 * g + 3]; cs += (double)h6[8 * g + 4]; nt += (double)h6[8 * g + 5];
            cw += (double)h6[8 * g + 6]; cb += (double)h6[8 * g + 7];
        }
        if (nt > 0) fprintf(stderr, "; wave-0 cycles per tile: chunk loop %.0f (of which waiting for a chunk's loads -- in-order behind older stores -- %.0f, at the chunk barrier %.0f), "
                            "fold + next-tile requests %.0f, stores %.0f (%.0f tiles)", cl / nt, cw / nt, cb / nt, ce / nt, cs / nt, nt);
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
// Row block (rows_only): the list's tiles are relative to the region -- row tile ti is tile ti_off + ti of the intervals, column tile tj is
// tj_off + tj --, only the rows carry thresholds, and the first launch's pairs are the column tiles [band_lo[ti], band_hi[ti]] of row tile ti.
__global__ void tk_filter_kernel(const int2 *__restrict__ in, int n_in, const double *__restrict__ lo, const double *__restrict__ hi,
                                 const double *__restrict__ scale, int nf, int nt,
                                 const uint16_t *__restrict__ tmax, int drop_band, int ti_off, int tj_off, int rows_only,
                                 const int32_t *__restrict__ band_lo, const int32_t *__restrict__ band_hi, int2 *__restrict__ out, int *__restrict__ count) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_in) return;
    const int2 t = in[k];
    const int ti = (t.x & TILE_INDEX_MASK) + ti_off, tj = t.y + tj_off;
    if (band_lo) { if (tj >= band_lo[ti - ti_off] && tj <= band_hi[ti - ti_off]) return; }
    else if (drop_band && abs(tj - ti) <= drop_band) return;
    double lb = 0.0;
    for (int f = 0; f < nf; ++f) {
        const double a0 = lo[(int64_t)f * nt + ti], a1 = hi[(int64_t)f * nt + ti], b0 = lo[(int64_t)f * nt + tj], b1 = hi[(int64_t)f * nt + tj];
        // (zero-pad layers: the gap between the rings' value sums, divided by the larger ring -- at least the smaller of the tiles' smallest 1 / ring size)
        lb += fmax(0.0, fmax(a0 - b1, b0 - a1)) * (scale ? fmin(scale[(int64_t)f * nt + ti], scale[(int64_t)f * nt + tj]) : 1.0);
    }
    const unsigned m = rows_only ? (unsigned)tmax[ti] : max((unsigned)tmax[ti], (unsigned)tmax[tj]);
    const double thr = (double)__uint_as_float(m << 16);
    // (the bound and the kernel's distances are sums of the same terms in different orders: a relative 1e-9 covers the rounding of both)
    if (lb * (1.0 - 1e-9) - 1e-9 <= thr) out[atomicAdd(count, 1)] = t;
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

int launch_region(hsd_ctx *ctx, PairArgs a, int BT, hipStream_t stream, bool allow_stream) {
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
            if (!a.sym) {
                // (a row block's sweep: rows and columns are two runs of whole tiles of the same position space; the tmax rows keep the
                //  stride of the whole list -- b.ntl_i / b.ntl_j below)
                ok = a.i_begin % 128 == 0 && a.j_begin % 128 == 0 && i_full == a.i_end && j_full == a.j_end && a.i_end <= (int64_t)a.ntl_ov * 128 &&
                     a.j_end <= (int64_t)a.ntl_ov * 128;
                LI.d_gl += 8 * (a.i_begin / 128); LJ.d_gl += 8 * (a.j_begin / 128);
                if (LI.d_tmax) { LI.d_tmax += a.i_begin / 128; LJ.d_tmax += a.j_begin / 128; }
                LI.ntiles = (int)((a.i_end - a.i_begin) / 128); LJ.ntiles = (int)((a.j_end - a.j_begin) / 128);
            }
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
                if (a.gl_ov) b.ntl_i = b.ntl_j = a.ntl_ov;
                b.suf = a.suf ? a.suf : ctx->d_suf;
                b.stamps = nullptr;
                if (a.tau) {
                    b.dbg &= ~8;                          // (the lifetime stamps are a matrix-launch hook)
                    // this launch's thresholds (they tighten between the phases of a sweep) as bfloat16, rounded up: rows [i_begin,
                    // i_end) carry thresholds; the columns as well when the region is mirrored onto itself
                    hipLaunchKernelGGL(tau_bf16_up_kernel, dim3((unsigned)((a.i_end - a.i_begin + 255) / 256)), dim3(256), 0, stream, a.tau, a.i_begin, a.i_end,
                                       const_cast<uint16_t *>(a.tau16));
                }
                // (a pruned sweep whose filter cannot run would re-evaluate the first launch's tile pairs and append their candidates twice:
                //  refuse -- hsd_pair_topk then runs the unpruned sweep)
                if (a.tau && a.prune_lo && n_in > 0 && !(l_in && ctx->d_tk_list && n_in <= ctx->tk_list_cap))
                    return hsd_fail(ctx, HSD_ERR_INTERNAL, "pruned k-nearest sweep: the tile filter cannot run (%d tile pairs, list of %d)", n_in, ctx->tk_list_cap);
                if (a.tau && a.prune_lo && l_in && n_in > 0 && ctx->d_tk_list && n_in <= ctx->tk_list_cap) {
                    // pruned sweep: drop the tile pairs that no row's CURRENT threshold can reach (tk_filter_kernel); the count comes
                    // back to the host -- one small synchronisation per launch of the sweep
                    int kept = 0;
                    // (largest threshold of every row tile: the region's rows; a row block's tiles sit behind the graph's in the intervals' numbering)
                    const int row_tiles = (int)((a.i_end - a.i_begin) / 128), ti_off = a.sym ? 0 : (int)(a.i_begin / 128), tj_off = a.sym ? 0 : (int)(a.j_begin / 128);
                    hipLaunchKernelGGL(tk_tilemax_kernel, dim3((unsigned)row_tiles), dim3(64), 0, stream, a.tau16 + a.i_begin, row_tiles, ctx->d_tk_tmax + a.i_begin / 128);
                    e = hipMemsetAsync(ctx->d_tk_count, 0, sizeof(int), stream);
                    hipLaunchKernelGGL(tk_filter_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, stream, l_in, n_in, a.prune_lo, a.prune_hi, a.prune_scale, a.prune_nf,
                                       a.prune_nt, ctx->d_tk_tmax, a.prune_drop_band, ti_off, tj_off, a.mirror ? 0 : 1, a.prune_band_lo, a.prune_band_hi,
                                       reinterpret_cast<int2 *>(ctx->d_tk_list), ctx->d_tk_count);
                    // The kernel reads the count the filter left in device memory (PairArgs::ntiles_dev): no host round trip between the launches
                    // of a sweep.  The grid is sized for the unfiltered list (workgroups without a tile leave at once).  HSD_TOPK_VERBOSE still
                    // fetches the count, to print it.
                    kept = n_in;
                    if (hsd_hook(H_TOPK_VERBOSE)) {
                        if (e == hipSuccess) e = hipMemcpyAsync(&kept, ctx->d_tk_count, sizeof(int), hipMemcpyDeviceToHost, stream);
                        if (e == hipSuccess) e = hipStreamSynchronize(stream);
                        if (e == hipSuccess) fprintf(stderr, "[hsd topk] pruned launch: %d of %d tile pairs kept\n", kept, n_in);
                    }
                    if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "k-nearest tile filter: %s", hipGetErrorString(e));
                    b.tiles = reinterpret_cast<const int2 *>(ctx->d_tk_list);
                    b.ntiles = n_in;
                    b.ntiles_dev = ctx->d_tk_count;
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
    if (a.gl_ov || a.orig_of) return hsd_fail(ctx, HSD_ERR_INTERNAL, "permuted operands need the persistent kernel (a tile list could not be built)");
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

}  // namespace hsd_pair

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
    a.kmask = wass ? nullptr : ctx->d_kl_mask; a.kwords = ctx->kl_words; a.khalf = ctx->kl_half;
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
        if (rb != 0) b.dbg &= ~16;      // (measurement hook "no HBM write stream" aims every tile at rows 0..255 of a matrix that starts at node 0: not for a row block)
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

