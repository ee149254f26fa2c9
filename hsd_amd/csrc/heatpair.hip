// All-pairs W1 on RAGGED sorted samples (heat-coefficient signatures) -- Variant A of the reference,
// model/HSD.py:98-114: D[a,b] = sum_k scipy.stats.wasserstein_distance(c_a[k], c_b[k]).
//
// scipy integrates |U - V| over the merged VALUES (a serial, data-dependent merge).  Here the same
// integral is taken over the RANK axis, where nothing depends on the data: with u the longer sample
// (n_u >= n_v, both ascending), on the integer axis [0, n_u n_v) element u_i covers
// [i n_v, (i+1) n_v) and v_j covers [j n_u, (j+1) n_u), so u_i meets at most two v's:
//     W1 = 1/(n_u n_v) * sum_i ( |u_i - v_q| w1 + |u_i - v_{q+1}| w2 ),
//     q = floor(i n_v / n_u),  r = i n_v - q n_u,  w1 = min(n_u - r, n_v),  w2 = n_v - w1
// -- n_u independent terms with exact integer weights, lanes over i, (q, r) advanced incrementally
// (no division in the loop).  Same integral as _cdf_distance, different summation order: agreement
// ~1e-16 relative (tests: rtol 1e-6 as the north star states, observed 1e-15).
//
// Pairs are evaluated four per wave, one per 16-lane DPP row (lanes strided over i, row sums by DPP), from
// samples resident in LDS: every LDS read in the loop is a contiguous 16-lane window.  Inside the square
// [r0, r1)^2 only a < b is evaluated and mirrored.  Two schedules of Variant A:
//   pair_ragged_tt_kernel     both sides are tiles of <= 8 consecutive nodes (16 waves; the next tile
//                             travels in registers) -- when every sample fits half the LDS
//   pair_ragged_tiled_kernel  resident tile of <= 16 nodes + one LDS slot per wave for the streamed node
//                             (8 waves); samples too large for a slot or the tile are read from global
//                             memory by the same code instantiated on generic pointers
// and pair_rowquirk_tiled_kernel for Variant B exactly as written (further down).
#include "common.hpp"

#include <algorithm>

namespace {

typedef const __attribute__((address_space(3))) double *lds_cdp;
typedef const __attribute__((address_space(1))) double *g_cdp;      // pointers carried in a struct are flat unless typed
typedef const __attribute__((address_space(1))) int64_t *g_ci64p;

constexpr int RG_WAVES = 8;      // waves per workgroup
constexpr int RG_MAX_TI = 16;    // resident nodes per workgroup
constexpr int RG_SLOT_REGS = 32; // a streamed node is prefetched into 32 doubles per lane: slot <= 2048 values

struct RaggedArgs {
    const int64_t *offsets;      // [n * L + 1]
    const double *values;        // sorted ascending inside each (node, ring) segment
    const int32_t *tile_begin;   // [n_tiles + 1] first node of each resident tile
    double *out;
    int *flag;
    int64_t n, r0, r1, ld;
    int L, lo, nl;               // layers lo .. lo + nl - 1
    int a_cap, b_cap;            // doubles of LDS: resident tile / one wave slot
    int groups;                  // the b range is split over this many workgroups per tile
};

// x / d and x % d for x <= 64 * d < 2^31, rd ~ 1/d: the float estimate is off by at most one
__device__ __forceinline__ void divmod_small(uint32_t x, uint32_t d, float rd, uint32_t &q, uint32_t &r) {
    q = (uint32_t)((float)x * rd);
    int32_t rr = (int32_t)(x - q * d);
    if (rr < 0) { --q; rr += (int32_t)d; }
    else if ((uint32_t)rr >= d) { ++q; rr -= (int32_t)d; }
    r = (uint32_t)rr;
}

// this lane's share of  sum_i |u_i - v_q| w1 + |u_i - v_{q+1}| w2   (n_u >= n_v >= 1)
template <typename P>
__device__ __forceinline__ double w1_rank_lanes(P U, uint32_t nu, P V, uint32_t nv, uint32_t lane) {
    uint32_t q, r, dq, dr;
    if (nu == nv) { q = lane; r = 0; dq = 64; dr = 0; }
    else {
        const float rd = __builtin_amdgcn_rcpf((float)nu);
        divmod_small(lane * nv, nu, rd, q, r);
        divmod_small(64u * nv, nu, rd, dq, dr);
    }
    double acc = 0.0;
    for (uint32_t i = lane; i < nu; i += 64) {
        const double ui = U[i];
        const uint32_t q1 = min(q + 1, nv - 1);
        const double v0 = V[q], v1 = V[q1];
        const uint32_t w1 = min(nu - r, nv), w2 = nv - w1;
        acc = fma(fabs(ui - v0), (double)w1, acc);
        acc = fma(fabs(ui - v1), (double)w2, acc);
        q += dq;
        r += dr;
        if (r >= nu) { r -= nu; ++q; }
    }
    return acc;
}

// One layer of one pair on the 16 lanes of a DPP row: this lane's share of the rank-axis sum (unscaled) with
// both samples in LDS at sh[ub ..], sh[vb ..], n_u >= n_v (n_u = 0: nothing to do).  Two elements per trip
// (i and i + 16) while both exist, then at most one more.  v_q and v_{q+1} come as one ds_read2_b64:
// position n_v is the next segment or the pad behind the sample (finite, weight 0).
__device__ __forceinline__ double rank_sum_row16(lds_cdp sh, uint32_t ub, uint32_t nu, uint32_t vb, uint32_t nv, uint32_t i0) {
    uint32_t q, r, dq, dr;
    {
        const uint32_t d = max(nu, 1u);
        const float rd = __builtin_amdgcn_rcpf((float)d);
        divmod_small(i0 * nv, d, rd, q, r);
        divmod_small(16u * nv, d, rd, dq, dr);
    }
    double acc0 = 0.0, acc1 = 0.0;
    uint32_t i = i0;
    const lds_cdp up = sh + ub, vp = sh + vb;
    for (; i + 16 < nu; i += 32) {
        uint32_t q2 = q + dq, r2 = r + dr;
        if (r2 >= nu) { r2 -= nu; ++q2; }
        const double u1 = up[i], u2 = up[i + 16];
        const double v10 = vp[q], v11 = vp[q + 1];
        const double v20 = vp[q2], v21 = vp[q2 + 1];
        const uint32_t w11 = min(nu - r, nv), w12 = nv - w11;
        const uint32_t w21 = min(nu - r2, nv), w22 = nv - w21;
        acc0 = fma(fabs(u1 - v10), (double)w11, acc0);
        acc1 = fma(fabs(u2 - v20), (double)w21, acc1);
        acc0 = fma(fabs(u1 - v11), (double)w12, acc0);
        acc1 = fma(fabs(u2 - v21), (double)w22, acc1);
        q = q2 + dq;
        r = r2 + dr;
        if (r >= nu) { r -= nu; ++q; }
    }
    if (i < nu) {
        const double u1 = up[i], v10 = vp[q], v11 = vp[q + 1];
        const uint32_t w11 = min(nu - r, nv), w12 = nv - w11;
        acc0 = fma(fabs(u1 - v10), (double)w11, acc0);
        acc1 = fma(fabs(u1 - v11), (double)w12, acc1);
    }
    return acc0 + acc1;
}

__device__ __forceinline__ int64_t readlane64(int64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// sum over the 16 lanes of a DPP row, result in every lane of the row (call in uniform control flow)
__device__ __forceinline__ double row16_sum(double x) {
    union { double d; int i[2]; } v, w;
    v.d = x;
#define HSD_ROW_STEP(ctrl)                                                       \
    w.i[0] = __builtin_amdgcn_update_dpp(0, v.i[0], ctrl, 0xF, 0xF, false);      \
    w.i[1] = __builtin_amdgcn_update_dpp(0, v.i[1], ctrl, 0xF, 0xF, false);      \
    v.d += w.d;
    HSD_ROW_STEP(0xB1)      // quad_perm [1,0,3,2]
    HSD_ROW_STEP(0x4E)      // quad_perm [2,3,0,1]
    HSD_ROW_STEP(0x141)     // row_half_mirror
    HSD_ROW_STEP(0x140)     // row_mirror
#undef HSD_ROW_STEP
    return v.d;
}

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in order; only the compiler must not move them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(RG_WAVES * 64) void pair_ragged_tiled_kernel(const RaggedArgs A) {
    extern __shared__ double rg_sh[];
    const int nl = A.nl, L = A.L;
    // LDS layout (doubles first, then the uint32 tables)
    double *a_vals = rg_sh;                                    // [a_cap]
    double *b_vals = a_vals + A.a_cap;                         // [RG_WAVES][b_cap]   (both capacities include 2 pad values)
    double *a_rcp = b_vals + (size_t)RG_WAVES * A.b_cap;       // [RG_MAX_TI][nl]   1 / ring size
    double *b_rcp = a_rcp + RG_MAX_TI * nl;                    // [RG_WAVES][nl]
    uint32_t *a_off = reinterpret_cast<uint32_t *>(b_rcp + RG_WAVES * nl);   // [RG_MAX_TI][nl + 1], packed positions
    uint32_t *b_off = a_off + RG_MAX_TI * (nl + 1);            // [RG_WAVES][nl + 1]
    uint32_t *a_total = b_off + RG_WAVES * (nl + 1);           // [1]

    const g_ci64p offs = (g_ci64p)A.offsets;
    const g_cdp vals = (g_cdp)A.values;
    const int tile = blockIdx.x / A.groups, grp = blockIdx.x % A.groups;
    const int64_t a0 = A.tile_begin[tile];
    const int ta = (int)(A.tile_begin[tile + 1] - a0);
    const int64_t per = (A.n + A.groups - 1) / A.groups;
    const int64_t b_lo = (int64_t)grp * per;
    const int64_t b_hi = min(A.n, b_lo + per);
    // inside the square only a <= b is evaluated: columns [r0, a0) belong to earlier tiles.  What is left of
    // [b_lo, b_hi) is  [b_lo, b_lo + len1)  followed by  [start2, b_hi)
    const int64_t len1 = max((int64_t)0, min(b_hi, A.r0) - b_lo);
    const int64_t start2 = max(b_lo, a0);
    const int64_t total = len1 + max((int64_t)0, b_hi - start2);
    if (total <= 0) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- resident tile: packed positions, reciprocal sizes, values --------------------------
    if (tid < ta) {
        uint32_t base = 0;
        for (int t = 0; t < tid; ++t) base += (uint32_t)(offs[(a0 + t) * L + A.lo + nl] - offs[(a0 + t) * L + A.lo]);
        const int64_t g0 = offs[(a0 + tid) * L + A.lo];
        for (int k = 0; k <= nl; ++k) a_off[tid * (nl + 1) + k] = base + (uint32_t)(offs[(a0 + tid) * L + A.lo + k] - g0);
        for (int k = 0; k < nl; ++k) {
            const int64_t sz = offs[(a0 + tid) * L + A.lo + k + 1] - offs[(a0 + tid) * L + A.lo + k];
            a_rcp[tid * nl + k] = sz > 0 ? 1.0 / (double)sz : 0.0;
        }
        if (tid == ta - 1) *a_total = a_off[tid * (nl + 1) + nl];
    }
    __syncthreads();
    const bool a_in_lds = *a_total + 2 <= (uint32_t)A.a_cap;
    if (a_in_lds) {
        if (tid < 2) a_vals[*a_total + tid] = 0.0;
        for (int t = 0; t < ta; ++t) {
            const uint32_t p0 = a_off[t * (nl + 1)], span = a_off[t * (nl + 1) + nl] - p0;
            const g_cdp src = vals + offs[(a0 + t) * L + A.lo];
            for (uint32_t e = tid; e < span; e += RG_WAVES * 64) a_vals[p0 + e] = src[e];
        }
    }
    __syncthreads();

    double *slot = b_vals + (size_t)wave * A.b_cap;
    uint32_t *boff = b_off + wave * (nl + 1);
    double *brcp = b_rcp + wave * nl;
    // Software pipeline over this wave's streamed nodes: the offsets are requested two nodes ahead and the
    // values one node ahead (into registers), so their latency runs under the evaluation of the current node.
    auto b_of = [&](int64_t m) { return m < len1 ? b_lo + m : start2 + (m - len1); };
    auto load_off = [&](int64_t m) -> int64_t { return (m < total && lane <= nl) ? offs[b_of(m) * L + A.lo + lane] : (int64_t)0; };
    double vreg[RG_SLOT_REGS];
    auto load_vals = [&](int64_t off_lane) {
        const int64_t g0 = readlane64(off_lane, 0);
        const uint32_t span = (uint32_t)(readlane64(off_lane, nl) - g0);
        if (span + 2 > (uint32_t)A.b_cap) return;
        const g_cdp src = vals + g0;
#pragma unroll
        for (int j = 0; j < RG_SLOT_REGS; ++j)
            if ((uint32_t)j * 64u < span) {
                const uint32_t e = (uint32_t)j * 64u + (uint32_t)lane;
                vreg[j] = e < span ? src[e] : 0.0;
            }
    };
    int64_t off_cur = load_off(wave), off_nxt = load_off(wave + RG_WAVES);
    load_vals(off_cur);
    for (int64_t m = wave; m < total; m += RG_WAVES) {
        const int64_t b = b_of(m);
        const bool sym = b >= A.r0 && b < A.r1;
        const int64_t gb = readlane64(off_cur, 0);
        const uint32_t bspan = (uint32_t)(readlane64(off_cur, nl) - gb);
        const bool b_in_lds = bspan + 2 <= (uint32_t)A.b_cap;
        wave_lds_sync();                                        // previous slot contents are dead
        if (lane <= nl) boff[lane] = (uint32_t)(off_cur - gb);
        if (b_in_lds) {
            if (lane < 2) slot[bspan + lane] = 0.0;
#pragma unroll
            for (int j = 0; j < RG_SLOT_REGS; ++j)
                if ((uint32_t)j * 64u < bspan) {
                    const uint32_t e = (uint32_t)j * 64u + (uint32_t)lane;
                    if (e < bspan) slot[e] = vreg[j];
                }
        }
        wave_lds_sync();
        if (lane < nl) {
            const uint32_t sz = boff[lane + 1] - boff[lane];
            brcp[lane] = sz > 0 ? 1.0 / (double)sz : 0.0;
        }
        wave_lds_sync();
        off_cur = off_nxt;
        off_nxt = load_off(m + 2 * RG_WAVES);
        if (m + RG_WAVES < total) load_vals(off_cur);

        int t_end = ta;
        if (sym && b < a0 + ta) t_end = (int)(b - a0) + 1;      // a <= b only (a == b: the zero diagonal)
        if (a_in_lds && b_in_lds) {
            // ---- four pairs per wave: the 16 lanes of DPP row g evaluate (a0 + t4 + g, b) -----------
            const lds_cdp sh = (lds_cdp)rg_sh;
            const uint32_t slot_idx = (uint32_t)A.a_cap + (uint32_t)wave * (uint32_t)A.b_cap;
            const uint32_t i0 = (uint32_t)lane & 15u;
            for (int t4 = 0; t4 < t_end; t4 += 4) {
                const int tg = t4 + (lane >> 4);
                const int tt = min(tg, ta - 1);
                const int64_t a = a0 + tg;
                const bool live = tg < t_end && a != b;
                double tot = 0.0;
                bool empty = false;
                for (int k = 0; k < nl; ++k) {
                    const uint32_t pa = a_off[tt * (nl + 1) + k], na = a_off[tt * (nl + 1) + k + 1] - pa;
                    const uint32_t pb = slot_idx + boff[k], nb = boff[k + 1] - boff[k];
                    const double scale = a_rcp[tt * nl + k] * brcp[k];
                    empty |= live && (na == 0 || nb == 0);
                    // longer sample indexed by the lanes; on a tie the node with the smaller index (symmetric result)
                    const bool a_is_u = na > nb || (na == nb && a < b);
                    const uint32_t nu = (live && na != 0 && nb != 0) ? (a_is_u ? na : nb) : 0u, nv = a_is_u ? nb : na;
                    const uint32_t ub = a_is_u ? pa : pb, vb = a_is_u ? pb : pa;
                    const double acc = rank_sum_row16(sh, ub, nu, vb, nv, i0);
                    tot = fma(acc, scale, tot);
                }
                tot = row16_sum(tot);
                if (empty) atomicOr(A.flag, 1);
                if (i0 == 0 && tg < t_end) {
                    const double val = a == b ? 0.0 : (empty ? __builtin_nan("") : tot);
                    A.out[(a - A.r0) * A.ld + b] = val;
                    if (sym && a != b) A.out[(b - A.r0) * A.ld + a] = val;
                }
            }
            continue;
        }
        // ---- a sample outside LDS: one pair per wave on generic pointers -----------------------------
        double res = 0.0;                                       // lane t keeps D[a0 + t, b]
        for (int t = 0; t < t_end; ++t) {
            const int64_t a = a0 + t;
            if (a == b) continue;                               // res of lane t stays 0
            double tot = 0.0;
            bool empty = false;
            for (int k = 0; k < nl; ++k) {
                const uint32_t pa = a_off[t * (nl + 1) + k], na = a_off[t * (nl + 1) + k + 1] - pa;
                const uint32_t pb = boff[k], nb = boff[k + 1] - pb;
                if (na == 0 || nb == 0) { empty = true; break; }
                const double scale = a_rcp[t * nl + k] * brcp[k];
                const bool a_is_u = na > nb || (na == nb && a < b);
                const double *ga = a_in_lds ? a_vals + pa : A.values + offs[a * L + A.lo + k];
                const double *gq = b_in_lds ? slot + pb : A.values + gb + pb;
                const double acc = a_is_u ? w1_rank_lanes(ga, na, gq, nb, (uint32_t)lane) : w1_rank_lanes(gq, nb, ga, na, (uint32_t)lane);
                tot = fma(acc, scale, tot);
            }
            if (empty) {
                if (lane == 0) atomicOr(A.flag, 1);
                tot = __builtin_nan("");
            } else {
                for (int m = 32; m > 0; m >>= 1) tot += __shfl_xor(tot, m);
            }
            if (lane == t) res = tot;
        }
        if (lane < t_end) {
            const int64_t a = a0 + lane;
            A.out[(a - A.r0) * A.ld + b] = res;
            if (sym && a != b) A.out[(b - A.r0) * A.ld + a] = res;
        }
    }
}



// ---- Variant A, tile x tile: both sides resident ---------------------------------------------------
// When every node's sample fits half the LDS, the streamed side needs no per-wave slot: a workgroup of 16
// waves keeps one tile of <= 8 nodes (`a`) resident and walks the other tiles (`b`), each loaded by all 1024
// threads as ONE contiguous copy -- the next tile's values travel in 10 registers per thread while the
// current one is evaluated.  Wave w evaluates b-node (w & 7) against a-nodes 4 (w >> 3) .. +3, one pair per
// DPP row as above.  4 waves per SIMD instead of 2.  Tables (packed positions, 1/size) are precomputed per
// tile on the host.
constexpr int TT_WAVES = 16;
constexpr int TT_NODES = 8;
constexpr int TT_REGS = 10;          // values of the next tile per thread: tile <= 10 * 1024 doubles

// packed[dst[c] + e] = values[src[c] + e], e < len[c]: one workgroup per copy (a node's layers lo..hi)
__global__ __launch_bounds__(256) void tile_pack_kernel(const double *__restrict__ values, const int64_t *__restrict__ src,
                                                       const int64_t *__restrict__ dst, const int64_t *__restrict__ len,
                                                       double *__restrict__ packed) {
    const int64_t c = blockIdx.x, s0 = src[c], d0 = dst[c], n = len[c];
    for (int64_t e = threadIdx.x; e < n; e += 256) packed[d0 + e] = values[s0 + e];
}

struct TTArgs {
    const double *tvals;             // tile-packed samples: tile t at tstart[t], ttotal[t] doubles
    const int64_t *tstart;
    const int32_t *ttotal;
    const int32_t *tile_begin;       // [n_tiles + 1] first node of each tile (tiles cover [0, n), breaks at r0, r1)
    const uint32_t *tpos;            // [n_tiles][TT_NODES][nl + 1] packed positions inside the tile
    const double *trcp;              // [n_tiles][TT_NODES][nl]     1 / ring size (0 for an empty ring)
    double *out;
    int *flag;
    int64_t r0, r1, ld;
    int nl, cap, ta0, ta1, n_tiles, groups;
};

__global__ __launch_bounds__(TT_WAVES * 64) void pair_ragged_tt_kernel(const TTArgs A) {
    extern __shared__ double tt_sh[];
    const int nl = A.nl;
    const int TP = TT_NODES * (nl + 1), TR = TT_NODES * nl;
    double *a_vals = tt_sh, *b_vals = tt_sh + A.cap;
    double *a_rcp = b_vals + A.cap, *b_rcp = a_rcp + TR;
    uint32_t *a_off = reinterpret_cast<uint32_t *>(b_rcp + TR), *b_off = a_off + TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ta = A.ta0 + blockIdx.x / A.groups, grp = blockIdx.x % A.groups;
    const int per = (A.n_tiles + A.groups - 1) / A.groups;
    // b tiles of this workgroup: [tb_lo, tb_hi) minus the tiles of the row range that precede ta
    int tb = grp * per;
    const int tb_hi = min(A.n_tiles, tb + per);
    auto skip = [&](int t) { return t >= A.ta0 && t < ta; };
    while (tb < tb_hi && skip(tb)) ++tb;
    if (tb >= tb_hi) return;

    const g_cdp tv = (g_cdp)A.tvals;
    const int64_t a0 = A.tile_begin[ta];
    const int na_nodes = A.tile_begin[ta + 1] - (int)a0;
    {   // resident tile
        const int64_t s0 = A.tstart[ta];
        const int tot = A.ttotal[ta];
        for (int e = tid; e < tot; e += TT_WAVES * 64) a_vals[e] = tv[s0 + e];
        if (tid < 2) a_vals[tot + tid] = 0.0;
        if (tid < TP) a_off[tid] = A.tpos[(int64_t)ta * TP + tid];
        if (tid < TR) a_rcp[tid] = A.trcp[(int64_t)ta * TR + tid];
    }
    double pre[TT_REGS];
    uint32_t pre_off = 0;
    double pre_rcp = 0.0;
    auto fetch = [&](int t) {          // values and tables of tile t -> registers
        const int64_t s0 = A.tstart[t];
        const int tot = A.ttotal[t];
#pragma unroll
        for (int j = 0; j < TT_REGS; ++j) {
            const int e = j * (TT_WAVES * 64) + tid;
            pre[j] = e < tot ? tv[s0 + e] : 0.0;          // zeros behind the data double as the pads
        }
        if (tid < TP) pre_off = A.tpos[(int64_t)t * TP + tid];
        if (tid < TR) pre_rcp = A.trcp[(int64_t)t * TR + tid];
    };
    fetch(tb);
    const lds_cdp sh = (lds_cdp)tt_sh;
    const uint32_t i0 = (uint32_t)lane & 15u;
    const int bsel = wave & 7, tg = (wave >> 3) * 4 + (lane >> 4);
    while (tb < tb_hi) {
        const int tot_b = A.ttotal[tb];
        __syncthreads();                                    // everyone is done with the previous b tile
#pragma unroll
        for (int j = 0; j < TT_REGS; ++j) {
            const int e = j * (TT_WAVES * 64) + tid;
            if (e < tot_b + 2) b_vals[e] = pre[j];
        }
        if (tid < TP) b_off[tid] = pre_off;
        if (tid < TR) b_rcp[tid] = pre_rcp;
        __syncthreads();
        const int64_t b0 = A.tile_begin[tb];
        const int nb_nodes = A.tile_begin[tb + 1] - (int)b0;
        int tn = tb + 1;
        while (tn < tb_hi && skip(tn)) ++tn;
        if (tn < tb_hi) fetch(tn);                          // in flight during the evaluation below

        const int64_t a = a0 + tg, b = b0 + bsel;
        const bool sym = b >= A.r0 && b < A.r1;
        const bool valid = bsel < nb_nodes && tg < na_nodes && (!sym || a <= b);
        const bool live = valid && a != b;
        const int tt = min(tg, na_nodes - 1), bb = min(bsel, nb_nodes - 1);
        double tot = 0.0;
        bool empty = false;
        for (int k = 0; k < nl; ++k) {
            const uint32_t pa = a_off[tt * (nl + 1) + k], na = a_off[tt * (nl + 1) + k + 1] - pa;
            const uint32_t pb0 = b_off[bb * (nl + 1) + k], nb = b_off[bb * (nl + 1) + k + 1] - pb0;
            const uint32_t pb = (uint32_t)A.cap + pb0;
            const double scale = a_rcp[tt * nl + k] * b_rcp[bb * nl + k];
            empty |= live && (na == 0 || nb == 0);
            const bool a_is_u = na > nb || (na == nb && a < b);
            const uint32_t nu = (live && na != 0 && nb != 0) ? (a_is_u ? na : nb) : 0u, nv = a_is_u ? nb : na;
            const double acc = rank_sum_row16(sh, a_is_u ? pa : pb, nu, a_is_u ? pb : pa, nv, i0);
            tot = fma(acc, scale, tot);
        }
        tot = row16_sum(tot);
        if (empty) atomicOr(A.flag, 1);
        if (i0 == 0 && valid) {
            const double val = a == b ? 0.0 : (empty ? __builtin_nan("") : tot);
            A.out[(a - A.r0) * A.ld + b] = val;
            if (sym && a != b) A.out[(b - A.r0) * A.ld + a] = val;
        }
        tb = tn;
    }
}

// ---- Variant B exactly as written (model/HSD.py:140-161) -----------------------------------------
// pair (a, b), a < b, layer k < hop:  p = psi[a, ring_k(a)],  q = psi[a, ring_k(b)]  (BOTH read from row a =
// `startIndex`, lines 152/155), d += calculate_distance(p, q): zero-pad the shorter list, sort both
// (tools/metrics.py:26-30), scipy W1 of two equal-length samples = mean |P_(r) - Q_(r)|.
// q depends on the PAIR, so every pair sorts: a workgroup owns one row a (the row of psi and the sorted
// sample of a stay in LDS), each of its 16 waves takes one b at a time, gathers q from the LDS row into
// REGISTERS, pads, sorts it there (bitonic over the wave: ds_bpermute exchanges, no LDS slot, no barriers)
// and sums |P_r - Q_r| over ranks.
constexpr int RQ_WAVES = 16;
constexpr int RQ_IDREGS = 8;       // member ids of the next partner travel in 8 registers per lane: <= 512 ids

struct QuirkArgs {
    const int64_t *offsets;
    const int32_t *members;
    const double *values;       // sorted samples psi[a, ring_k(a)]
    const double *psi;
    double *out;
    int64_t n, r0, r1, ld, ldpsi;
    int L, hop;
    int row_in_lds, p_cap, groups;
    int id_cap;                  // member ids of a partner's rings 0 .. hop-1 parked in LDS per wave (0: read from global)
};

__device__ __forceinline__ double wave_sum(double x) {
    union { double d; int i[2]; } v, t;
    v.d = row16_sum(x);
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < 64; l += 16) {
        t.i[0] = __builtin_amdgcn_readlane(v.i[0], l);
        t.i[1] = __builtin_amdgcn_readlane(v.i[1], l);
        s += t.d;
    }
    return s;
}

// Bitonic sort of 64 * E doubles held by one wave in registers: element r * 64 + lane lives in x[r].  Stages
// with a partner in another lane exchange through ds_bpermute (__shfl_xor), stages with a partner >= 64 apart
// stay inside the lane.  Every loop bound is a compile-time constant, so x[] is indexed statically.  For
// E == 1 the network stops at the runtime size p2 (power of two <= 64): short rings skip the later stages.
template <int E>
__device__ __forceinline__ void wave_sort_regs(double (&x)[E], uint32_t lane, uint32_t p2) {
    if constexpr (E == 1) {
        for (uint32_t k = 2; k <= p2; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                const double o = __shfl_xor(x[0], (int)j);
                const bool take_min = ((lane & j) == 0) == ((lane & k) == 0);
                x[0] = ((x[0] > o) == take_min) ? o : x[0];
            }
        return;
    }
#pragma unroll
    for (uint32_t k = 2; k <= 64u * E; k <<= 1) {
#pragma unroll
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64u) {
                const uint32_t jr = j >> 6;
#pragma unroll
                for (int r = 0; r < E; ++r)
                    if (((uint32_t)r & jr) == 0) {
                        const int r2 = r | (int)jr;
                        const bool up = (((uint32_t)r * 64u) & k) == 0;
                        const double a = x[r], c = x[r2];
                        const bool sw = (a > c) == up;          // one compare + selects (fmin/fmax would add NaN canonicalisation)
                        x[r] = sw ? c : a;
                        x[r2] = sw ? a : c;
                    }
            } else {
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const double o = __shfl_xor(x[r], (int)j);
                    const bool take_min = ((lane & j) == 0) == ((((uint32_t)r * 64u + lane) & k) == 0);
                    x[r] = ((x[r] > o) == take_min) ? o : x[r];
                }
            }
        }
    }
}

// One layer of one pair of Variant B as written on one wave: q = psi[a, ring_k(b)] gathered into registers and
// sorted there (only the n_q real values: padded to a power of two with +inf).  The zero pads of the
// alignment are never materialised: sorted(q + pads) is q's negatives, then the pads, then the rest, and the
// same holds for P = the pre-sorted sample of a (`neg` negatives, `padp` pads).  Returns this lane's share of
// sum_r |P_r - Q_r| over the Lm aligned ranks.
template <int E, typename IdPtr>
__device__ __forceinline__ double quirk_layer(IdPtr ids, uint32_t nq, uint32_t Lm,
                                              const double *row_lds, g_cdp row_glb, const double *pk, uint32_t padp, uint32_t neg,
                                              uint32_t lane) {
    double x[E];
    uint32_t negq = 0;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t t = (uint32_t)r * 64u + lane;
        double v = __builtin_inf();
        if (t < nq) {
            const int32_t id = ids[t];
            v = row_lds ? row_lds[id] : row_glb[id];
        }
        x[r] = v;
        negq += (uint32_t)__popcll(__ballot(v < 0.0));
    }
    uint32_t p2 = 1;
    while (p2 < nq) p2 <<= 1;
    wave_sort_regs<E>(x, lane, p2);
    const uint32_t padq = Lm - nq;
    auto P_at = [&](uint32_t pos) -> double { return pos < neg ? pk[pos] : (pos < neg + padp ? 0.0 : pk[pos - padp]); };
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t t = (uint32_t)r * 64u + lane;
        if (t < nq) acc += fabs(P_at(t < negq ? t : t + padq) - x[r]);
    }
    for (uint32_t pos = negq + lane; pos < negq + padq; pos += 64) acc += fabs(P_at(pos));    // ranks held by q's pads
    return acc;
}

__global__ __launch_bounds__(RQ_WAVES * 64) void pair_rowquirk_tiled_kernel(const QuirkArgs A) {
    extern __shared__ double rq_sh[];
    const int hop = A.hop, L = A.L;
    double *row = rq_sh;                                         // [n] when row_in_lds
    double *pv = row + (A.row_in_lds ? A.n : 0);                 // [p_cap] sorted sample of a, layers < hop
    uint32_t *poff = reinterpret_cast<uint32_t *>(pv + A.p_cap);  // [hop + 1]
    uint32_t *pneg = poff + hop + 1;                             // [hop] negatives in each sample of a
    const g_ci64p offs = (g_ci64p)A.offsets;
    const g_cdp vals = (g_cdp)A.values;
    const __attribute__((address_space(1))) int32_t *mem = (const __attribute__((address_space(1))) int32_t *)A.members;
    const int64_t a = blockIdx.x / A.groups;
    const int grp = blockIdx.x % A.groups;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool a_row = a >= A.r0;                                // a < r1 by construction of the grid
    if (a_row && grp == 0 && tid == 0) A.out[(a - A.r0) * A.ld + a] = 0.0;
    // partners of a: every b > a when row a is wanted, else only the wanted rows b in [r0, r1)
    const int64_t lo = a_row ? a + 1 : A.r0, hi = a_row ? A.n : A.r1;
    const int64_t per = (hi - lo + A.groups - 1) / A.groups;
    const int64_t b_lo = lo + (int64_t)grp * per, b_hi = min(hi, b_lo + per);
    if (per <= 0 || b_lo >= b_hi) return;

    const g_cdp prow = (g_cdp)A.psi + a * A.ldpsi;
    const int64_t ga = offs[a * L];
    const uint32_t pspan = (uint32_t)(offs[a * L + hop] - ga);
    if (A.row_in_lds)
        for (int64_t e = tid; e < A.n; e += RQ_WAVES * 64) row[e] = prow[e];
    if (tid <= hop) poff[tid] = (uint32_t)(offs[a * L + tid] - ga);
    if (tid < hop) pneg[tid] = 0;
    for (uint32_t e = tid; e < pspan; e += RQ_WAVES * 64) pv[e] = vals[ga + e];
    __syncthreads();
    for (uint32_t e = tid; e < pspan; e += RQ_WAVES * 64)
        if (pv[e] < 0.0) {
            int k = 0;
            while (e >= poff[k + 1]) ++k;
            atomicAdd(&pneg[k], 1u);
        }
    __syncthreads();

    // Software pipeline over this wave's partners: ring offsets two partners ahead, the member ids of rings
    // 0 .. hop-1 (contiguous in the member list) one partner ahead in registers, parked in a per-wave LDS
    // buffer while the pair is evaluated -- no global latency between the layers of a pair.
    typedef const __attribute__((address_space(3))) int32_t *lds_cip;
    int32_t *ibuf = reinterpret_cast<int32_t *>(pneg + hop) + (size_t)wave * A.id_cap;
    const bool use_buf = A.id_cap > 0;
    auto load_off = [&](int64_t bb) -> int64_t { return (bb < b_hi && lane <= hop) ? offs[bb * L + lane] : (int64_t)0; };
    int32_t idr[RQ_IDREGS];
    auto load_ids = [&](int64_t off_lane) {
        const int64_t g0 = readlane64(off_lane, 0);
        const uint32_t span = (uint32_t)(readlane64(off_lane, hop) - g0);
#pragma unroll
        for (int j = 0; j < RQ_IDREGS; ++j)
            if ((uint32_t)j * 64u < span) {
                const uint32_t e = (uint32_t)j * 64u + (uint32_t)lane;
                idr[j] = e < span ? mem[g0 + e] : 0;
            }
    };
    int64_t off_cur = load_off(b_lo + wave), off_nxt = load_off(b_lo + wave + RQ_WAVES);
    if (use_buf) load_ids(off_cur);
    for (int64_t b = b_lo + wave; b < b_hi; b += RQ_WAVES) {
        double d = 0.0;
        const int64_t off_b = off_cur;
        const int64_t g0 = readlane64(off_b, 0);
        if (use_buf) {
            const uint32_t span = (uint32_t)(readlane64(off_b, hop) - g0);
            wave_lds_sync();                                     // the previous partner's ids are dead
#pragma unroll
            for (int j = 0; j < RQ_IDREGS; ++j)
                if ((uint32_t)j * 64u < span) {
                    const uint32_t e = (uint32_t)j * 64u + (uint32_t)lane;
                    if (e < span) ibuf[e] = idr[j];
                }
            wave_lds_sync();
        }
        off_cur = off_nxt;
        off_nxt = load_off(b + 2 * RQ_WAVES);
        if (use_buf && b + RQ_WAVES < b_hi) load_ids(off_cur);
        for (int k = 0; k < hop; ++k) {
            const uint32_t qrel = (uint32_t)(readlane64(off_b, k) - g0);
            const uint32_t nq = (uint32_t)(readlane64(off_b, k + 1) - g0) - qrel, np = poff[k + 1] - poff[k];
            if (np == 0 && nq == 0) continue;                    // tools/metrics.py:170-171
            const uint32_t Lm = max(np, nq);
            const double *pk = pv + poff[k];
            const uint32_t padp = Lm - np, neg = pneg[k];
            const double *rowp = A.row_in_lds ? row : nullptr;
            double acc;
            if (use_buf) {
                const lds_cip ids = (lds_cip)ibuf + qrel;
                if (nq <= 64) acc = quirk_layer<1>(ids, nq, Lm, rowp, prow, pk, padp, neg, (uint32_t)lane);
                else if (nq <= 128) acc = quirk_layer<2>(ids, nq, Lm, rowp, prow, pk, padp, neg, (uint32_t)lane);
                else if (nq <= 256) acc = quirk_layer<4>(ids, nq, Lm, rowp, prow, pk, padp, neg, (uint32_t)lane);
                else acc = quirk_layer<8>(ids, nq, Lm, rowp, prow, pk, padp, neg, (uint32_t)lane);
            } else {
                const auto ids = mem + g0 + qrel;
                if (nq <= 64) acc = quirk_layer<1>(ids, nq, Lm, rowp, prow, pk, padp, neg, (uint32_t)lane);
                else if (nq <= 128) acc = quirk_layer<2>(ids, nq, Lm, rowp, prow, pk, padp, neg, (uint32_t)lane);
                else if (nq <= 256) acc = quirk_layer<4>(ids, nq, Lm, rowp, prow, pk, padp, neg, (uint32_t)lane);
                else if (nq <= 512) acc = quirk_layer<8>(ids, nq, Lm, rowp, prow, pk, padp, neg, (uint32_t)lane);
                else acc = quirk_layer<16>(ids, nq, Lm, rowp, prow, pk, padp, neg, (uint32_t)lane);
            }
            d += wave_sum(acc) / (double)Lm;
        }
        if (lane == 0) {
            if (a_row) A.out[(a - A.r0) * A.ld + b] = d;
            if (b >= A.r0 && b < A.r1) A.out[(b - A.r0) * A.ld + a] = d;
        }
    }
}

}   // namespace


// Tile x tile schedule of Variant A.  HSD_ERR_UNSUPPORTED when a node's sample does not fit half the LDS.
static int launch_pair_ragged_tt(hsd_ctx *ctx, int layer_lo, int layer_hi, int64_t r0, int64_t r1, double *dst, int64_t dld) {
    const int L = ctx->heat_hop + 1, nl = layer_hi - layer_lo + 1;
    const int64_t n = ctx->n;
    if (ctx->heat_max_seg >= (1 << 24) || n >= ((int64_t)1 << 31)) return HSD_ERR_UNSUPPORTED;
    const std::vector<int64_t> &off = ctx->h_hoffsets;
    auto span = [&](int64_t v) { return off[(size_t)(v * L + layer_lo + nl)] - off[(size_t)(v * L + layer_lo)]; };
    size_t lds_max = ctx->prop.maxSharedMemoryPerMultiProcessor ? ctx->prop.maxSharedMemoryPerMultiProcessor : 65536;
    lds_max = std::min<size_t>(lds_max, 160 * 1024);
    const size_t aux = (size_t)2 * TT_NODES * nl * sizeof(double) + (size_t)2 * TT_NODES * (nl + 1) * sizeof(uint32_t) + 64;
    if (aux + 8192 > lds_max) return HSD_ERR_UNSUPPORTED;
    int64_t cap = std::min<int64_t>((int64_t)((lds_max - aux) / sizeof(double) / 2), (int64_t)TT_REGS * TT_WAVES * 64);
    int64_t max_span = 0;
    for (int64_t v = 0; v < n; ++v) max_span = std::max(max_span, span(v));
    if (max_span + 2 > cap) return HSD_ERR_UNSUPPORTED;
    cap = std::min<int64_t>(cap, std::max<int64_t>(TT_NODES * max_span + 2, 64));
    // tiles over [0, n): consecutive nodes, <= TT_NODES and <= cap - 2 values, never across r0 or r1
    std::vector<int32_t> tile_begin{0};
    int64_t cur = 0;
    int cnt = 0;
    for (int64_t v = 0; v < n; ++v) {
        const int64_t sp = span(v);
        if (cnt > 0 && (cnt == TT_NODES || cur + sp + 2 > cap || v == r0 || v == r1)) {
            tile_begin.push_back((int32_t)v);
            cur = 0;
            cnt = 0;
        }
        cur += sp;
        ++cnt;
    }
    tile_begin.push_back((int32_t)n);
    const int n_tiles = (int)tile_begin.size() - 1;
    // worth it only while a tile holds several nodes (wave w needs b-node w & 7 and a-nodes 4 (w >> 3) ..): with
    // long samples the tiles degenerate to one node each and the streamed-slot schedule is the better one
    if ((int64_t)n_tiles * 5 > n + 10) return HSD_ERR_UNSUPPORTED;
    int ta0 = 0, ta1 = n_tiles;
    for (int t = 0; t < n_tiles; ++t) {
        if (tile_begin[(size_t)t] == r0) ta0 = t;
        if (tile_begin[(size_t)t + 1] == r1) ta1 = t + 1;
    }
    const bool contiguous = layer_lo == 0 && nl == L;       // the resident values are already tile-packed
    std::vector<int64_t> tstart((size_t)n_tiles);
    std::vector<int32_t> ttotal((size_t)n_tiles);
    std::vector<uint32_t> tpos((size_t)n_tiles * TT_NODES * (nl + 1), 0u);
    std::vector<double> trcp((size_t)n_tiles * TT_NODES * nl, 0.0);
    std::vector<int64_t> gsrc, gdst, glen;                   // copy list for the packed buffer (sub-range of layers)
    int64_t packed = 0;
    for (int t = 0; t < n_tiles; ++t) {
        const int64_t v0 = tile_begin[(size_t)t], v1 = tile_begin[(size_t)t + 1];
        tstart[(size_t)t] = contiguous ? off[(size_t)(v0 * L)] : packed;
        uint32_t pos = 0;
        for (int64_t v = v0; v < v1; ++v) {
            const int node = (int)(v - v0);
            for (int k = 0; k <= nl; ++k)
                tpos[((size_t)t * TT_NODES + node) * (nl + 1) + k] = pos + (uint32_t)(off[(size_t)(v * L + layer_lo + k)] - off[(size_t)(v * L + layer_lo)]);
            for (int k = 0; k < nl; ++k) {
                const int64_t sz = off[(size_t)(v * L + layer_lo + k + 1)] - off[(size_t)(v * L + layer_lo + k)];
                trcp[((size_t)t * TT_NODES + node) * nl + k] = sz > 0 ? 1.0 / (double)sz : 0.0;
            }
            if (!contiguous) { gsrc.push_back(off[(size_t)(v * L + layer_lo)]); gdst.push_back(packed + pos); glen.push_back(span(v)); }
            pos += (uint32_t)span(v);
        }
        ttotal[(size_t)t] = (int32_t)pos;
        packed += pos;
    }
    // device copies
    double *d_tvals = nullptr, *d_trcp = nullptr;
    int64_t *d_tstart = nullptr;
    int32_t *d_ttotal = nullptr, *d_tbegin = nullptr;
    uint32_t *d_tpos = nullptr;
    int rc = HSD_OK;
    hipError_t e = hipSuccess;
    auto up = [&](auto **d, const auto &h) {
        if (rc == HSD_OK) rc = hsd_dev_alloc(ctx, d, h.size());
        if (rc == HSD_OK && e == hipSuccess) e = hipMemcpyAsync(*d, h.data(), h.size() * sizeof(h[0]), hipMemcpyHostToDevice, ctx->stream);
    };
    up(&d_tstart, tstart); up(&d_ttotal, ttotal); up(&d_tbegin, tile_begin); up(&d_tpos, tpos); up(&d_trcp, trcp);
    int64_t *d_gsrc = nullptr, *d_gdst = nullptr, *d_glen = nullptr;
    if (rc == HSD_OK && !contiguous) {
        rc = hsd_dev_alloc(ctx, &d_tvals, (size_t)packed);
        up(&d_gsrc, gsrc); up(&d_gdst, gdst); up(&d_glen, glen);
        if (rc == HSD_OK && e == hipSuccess && !gsrc.empty())
            hipLaunchKernelGGL(tile_pack_kernel, dim3((unsigned)gsrc.size()), dim3(256), 0, ctx->stream, ctx->d_hvalues, d_gsrc, d_gdst,
                               d_glen, d_tvals);
    }
    if (rc == HSD_OK && e == hipSuccess) {
        const int rows_tiles = ta1 - ta0;
        // ODD number of groups: workgroups are dealt round-robin over the 8 XCDs, and inside the square the
        // groups of one tile carry very different work (the low column ranges are skipped for late tiles) -- with
        // an even count the light and the heavy workgroups would each land on their own half of the XCDs
        // (measured: 2 groups 79 ms, 1 or 3 groups 54 ms at ER N=10k)
        int groups = std::max(1, (1024 + rows_tiles - 1) / std::max(rows_tiles, 1));
        if (const char *eg = hsd_hook(H_HEAT_GROUPS)) groups = std::max(1, atoi(eg));      // measurement hook
        groups = std::min(groups, std::max(1, n_tiles / 4));
        groups |= 1;
        const size_t lds = (size_t)2 * cap * sizeof(double) + aux;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(pair_ragged_tt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) {
            TTArgs A{contiguous ? ctx->d_hvalues : d_tvals, d_tstart, d_ttotal, d_tbegin, d_tpos, d_trcp, dst, ctx->d_flag, r0, r1, dld,
                     nl, (int)cap, ta0, ta1, n_tiles, groups};
            hipLaunchKernelGGL(pair_ragged_tt_kernel, dim3((unsigned)(rows_tiles * groups)), dim3(TT_WAVES * 64), lds, ctx->stream, A);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);          // host vectors and tables stay alive until here
    else (void)hipStreamSynchronize(ctx->stream);
    hsd_dev_free(d_gsrc); hsd_dev_free(d_gdst); hsd_dev_free(d_glen);
    hsd_dev_free(d_tvals); hsd_dev_free(d_trcp); hsd_dev_free(d_tstart); hsd_dev_free(d_ttotal); hsd_dev_free(d_tbegin); hsd_dev_free(d_tpos);
    if (rc) return rc;
    if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "pair_ragged_tt: %s", hipGetErrorString(e));
    return HSD_OK;
}

// Rows [r0, r1) of the Variant-A matrix on the resident ragged samples.  Returns HSD_ERR_UNSUPPORTED
// (without touching ctx->err) when the shape is outside what the tiled kernel handles, so the caller
// can take the one-thread-per-pair kernel instead.
int hsd_launch_pair_ragged_tiled(hsd_ctx *ctx, int layer_lo, int layer_hi, int64_t r0, int64_t r1, double *dst, int64_t dld) {
    {
        // both sides resident (16 waves per CU) when every sample fits half the LDS; HSD_HEAT_TT=0: streamed slots
        const char *ett = hsd_hook(H_HEAT_TT);
        if (!(ett && atoi(ett) == 0) && !hsd_hook(H_HEAT_BCAP) && !hsd_hook(H_HEAT_ACAP)) {
            const int rtt = launch_pair_ragged_tt(ctx, layer_lo, layer_hi, r0, r1, dst, dld);
            if (rtt != HSD_ERR_UNSUPPORTED) return rtt;
        }
    }
    const int L = ctx->heat_hop + 1, nl = layer_hi - layer_lo + 1;
    const int64_t n = ctx->n;
    if (ctx->heat_max_seg >= (1 << 24) || n >= ((int64_t)1 << 31)) return HSD_ERR_UNSUPPORTED;
    const std::vector<int64_t> &off = ctx->h_hoffsets;
    auto span = [&](int64_t v) { return off[(size_t)(v * L + layer_lo + nl)] - off[(size_t)(v * L + layer_lo)]; };
    int64_t max_span = 1;
    for (int64_t v = 0; v < n; ++v) max_span = std::max(max_span, span(v));
    // LDS budget: whole CU (one workgroup of 8 waves per CU)
    size_t lds_max = ctx->prop.maxSharedMemoryPerMultiProcessor ? ctx->prop.maxSharedMemoryPerMultiProcessor : 65536;
    lds_max = std::min<size_t>(lds_max, 160 * 1024);
    const size_t aux = (size_t)(RG_MAX_TI + RG_WAVES) * nl * sizeof(double) + ((size_t)(RG_MAX_TI + RG_WAVES) * (nl + 1) + 2) * sizeof(uint32_t);
    if (aux + 4096 > lds_max) return HSD_ERR_UNSUPPORTED;
    const int64_t budget = (int64_t)((lds_max - aux - 64) / sizeof(double));
    int64_t b_cap = std::min<int64_t>({max_span + 2, budget * 5 / 8 / RG_WAVES, (int64_t)64 * RG_SLOT_REGS});
    if (const char *e = hsd_hook(H_HEAT_BCAP)) b_cap = std::min<int64_t>(b_cap, atoll(e));
    b_cap = std::max<int64_t>(b_cap, 3);
    int64_t a_cap = budget - b_cap * RG_WAVES;
    if (const char *e = hsd_hook(H_HEAT_ACAP)) a_cap = std::max<int64_t>(3, std::min<int64_t>(a_cap, atoll(e)));   // test hook
    // resident tiles: consecutive nodes packed while they fit
    std::vector<int32_t> tiles;
    tiles.push_back((int32_t)r0);
    int64_t cur = 0;
    int cnt = 0;
    for (int64_t a = r0; a < r1; ++a) {
        const int64_t s = span(a);
        if (cnt > 0 && (cnt == RG_MAX_TI || cur + s + 2 > a_cap)) {
            tiles.push_back((int32_t)a);
            cur = 0;
            cnt = 0;
        }
        cur += s;
        ++cnt;
    }
    tiles.push_back((int32_t)r1);
    const int n_tiles = (int)tiles.size() - 1;
    int groups = (int)std::max<int64_t>(1, (2048 + n_tiles - 1) / n_tiles);
    groups = (int)std::min<int64_t>(groups, std::max<int64_t>(1, n / (4 * RG_WAVES)));
    groups |= 1;        // odd: see launch_pair_ragged_tt (XCD round-robin vs the skipped column ranges)
    int32_t *d_tiles = nullptr;
    int rc = hsd_dev_alloc(ctx, &d_tiles, tiles.size());
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
    const size_t lds = (size_t)(a_cap + b_cap * RG_WAVES) * sizeof(double) + aux;
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(pair_ragged_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) {
        RaggedArgs A{ctx->d_hoffsets, ctx->d_hvalues, d_tiles, dst, ctx->d_flag, n, r0, r1, dld, L, layer_lo, nl,
                     (int)a_cap, (int)b_cap, groups};
        hipLaunchKernelGGL(pair_ragged_tiled_kernel, dim3((unsigned)((int64_t)n_tiles * groups)), dim3(RG_WAVES * 64), lds, ctx->stream, A);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // tiles.data() and d_tiles stay alive until here
    }
    hsd_dev_free(d_tiles);
    if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "pair_ragged_tiled: %s", hipGetErrorString(e));
    return HSD_OK;
}

// Rows [r0, r1) of the Variant-B-as-written matrix.  HSD_ERR_UNSUPPORTED (ctx->err untouched) when a ring of
// the compared layers has more than 1024 members: the caller then takes the one-workgroup-per-pair kernel.
int hsd_launch_pair_rowquirk_tiled(hsd_ctx *ctx, const double *psi_dev, int64_t ldpsi, int64_t r0, int64_t r1, double *dst, int64_t dld) {
    const int hop = ctx->heat_hop, L = hop + 1;
    const int64_t n = ctx->n;
    if (hop < 1 || n >= ((int64_t)1 << 31)) return HSD_ERR_UNSUPPORTED;
    const std::vector<int64_t> &off = ctx->h_hoffsets;
    int64_t max_ring = 1, p_cap = 1;
    for (int64_t v = 0; v < n; ++v) {
        for (int k = 0; k < hop; ++k) max_ring = std::max(max_ring, off[(size_t)(v * L + k + 1)] - off[(size_t)(v * L + k)]);
        p_cap = std::max(p_cap, off[(size_t)(v * L + hop)] - off[(size_t)(v * L)]);
    }
    if (max_ring > 1024) return HSD_ERR_UNSUPPORTED;             // a wave sorts up to 64 * 16 values in registers
    size_t lds_max = ctx->prop.maxSharedMemoryPerMultiProcessor ? ctx->prop.maxSharedMemoryPerMultiProcessor : 65536;
    lds_max = std::min<size_t>(lds_max, 160 * 1024);
    const size_t aux = (size_t)(2 * hop + 2) * sizeof(uint32_t) + 64;
    const int id_cap = p_cap <= 64 * RQ_IDREGS ? (int)round_up(p_cap, 2) : 0;
    const size_t base = (size_t)p_cap * sizeof(double) + aux + (size_t)RQ_WAVES * id_cap * sizeof(int32_t);
    if (base > lds_max) return HSD_ERR_UNSUPPORTED;
    int row_in_lds = base + (size_t)n * sizeof(double) <= lds_max ? 1 : 0;
    if (const char *er = hsd_hook(H_HEAT_ROWLDS)) if (atoi(er) == 0) row_in_lds = 0;      // measurement hook
    const size_t lds = base + (row_in_lds ? (size_t)n * sizeof(double) : 0);
    const int groups = (int)std::max<int64_t>(1, std::min<int64_t>((2048 + r1 - 1) / r1, n / (2 * RQ_WAVES) + 1)) | 1;   // odd, as above
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pair_rowquirk_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) {
        QuirkArgs A{ctx->d_hoffsets, ctx->d_hmembers, ctx->d_hvalues, psi_dev, dst, n, r0, r1, dld, ldpsi, L, hop,
                    row_in_lds, (int)p_cap, groups, id_cap};
        hipLaunchKernelGGL(pair_rowquirk_tiled_kernel, dim3((unsigned)(r1 * groups)), dim3(RQ_WAVES * 64), lds, ctx->stream, A);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "pair_rowquirk_tiled: %s", hipGetErrorString(e));
    return HSD_OK;
}
