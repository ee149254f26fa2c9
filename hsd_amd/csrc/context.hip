// Context, memory, stopwatch, graph upload and resident-matrix checks.
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

int hsd_fail(hsd_ctx *ctx, int status, const char *fmt, ...) {
    if (ctx) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return status;
}


// ---- the environment hooks (csrc/hooks.hpp) ----
const HsdHookInfo hsd_hook_table[H_COUNT] = {
    {"HSD_FORCE_COMM", "1 = a one-rank RCCL communicator (single-GPU test of the real all-gather)"},
    {"HSD_COMM_DRYRUN", "multi-rank control flow without a communicator (the caller exchanges through the host)"},
    {"HSD_CHEBY_LONGROWS", "heat path: 0 = no chunking of hub rows of the Laplacian"},
    {"HSD_CHEBY_STORE", "multi-scale heat path: 0 = fused per-step accumulators instead of stored planes"},
    {"HSD_CHEBY_C", "heat path: cap of the impulse block width"},
    {"HSD_HEAT_SORT_LDS", "coefficient-sample path: LDS length classes of the ring sort"},
    {"HSD_HEAT_TILED", "coefficient-sample kernels: 0 = per-pair fallback kernels"},
    {"HSD_HEAT_GROUPS", "coefficient-sample kernels: column groups per tile"},
    {"HSD_HEAT_TT", "coefficient-sample kernels: 0 = streamed-slot instead of tile x tile schedule"},
    {"HSD_HEAT_BCAP", "coefficient-sample kernels: shrunken LDS capacity of the column-side samples (tests)"},
    {"HSD_HEAT_ACAP", "coefficient-sample kernels: shrunken LDS capacity of the row-side samples (tests)"},
    {"HSD_HEAT_ROWLDS", "coefficient-sample path (Variant B as written): LDS row cache switch"},
    {"HSD_KHOP_MS", "0 | 1 = histogram build by the per-source sweep / by the bit-parallel sweep whatever the graph size"},
    {"HSD_KHOP_BOTTOMUP", "0 | 1 = force the bottom-up expansion of the per-source k-hop sweep off / on"},
    {"HSD_KHOP_MS_NOPLANES", "test hook: rehearses the allocation failure of the bit-parallel sweep's planes"},
    {"HSD_TILE", "one-tile pair kernel with that tile edge (64 | 128); also keeps the persistent kernel off"},
    {"HSD_SUPERTILE", "tile-order group size S (S x S tiles per group, default 8; 1 = formula order)"},
    {"HSD_XCD_DEAL", "0 = tile groups in row-major order instead of dealt round-robin to the XCDs"},
    {"HSD_PLAN_VERBOSE", "hsd_pair_prepare prints the chosen order, thresholds, byte ranges and stage times to stderr"},
    {"HSD_WAVES_PER_CU", "resident waves of the persistent kernel per CU (default 8)"},
    {"HSD_STREAM", "pair kernel: 0 = one tile per workgroup for every tile, 1 = force the persistent kernel also for small regions"},
    {"HSD_GRAN_SORT", "0 = tiles composed in natural granule order instead of the planner's sorted order"},
    {"HSD_ROWBLOCK", "0 = a row block as three launches (square + two rectangles) instead of one persistent launch"},
    {"HSD_SAD16", "0 = zero-pad W1 on plain uint32 columns; read at prepare time"},
    {"HSD_RANK", "zero-pad layers: 0 = bin form only, 1 = whole-layer rank form only, 2 = thresholds (default); read at prepare time"},
    {"HSD_PLAN_THREADS", "cap of the host threads that price the candidate granule orders (1 = the calling thread only)"},
    {"HSD_PLAN_THREADS_FAIL", "test hook: creating the t-th planning thread fails (what RLIMIT_NPROC does)"},
    {"HSD_U8", "0 = packed zero-pad layers without their uint8 ranges; read at prepare time"},
    {"HSD_SHARD_BALANCE", "0 = hsd_pair_shard returns blocks of equal height whatever the kind"},
    {"HSD_KL_POINT", "KL kinds: 0 = layer 0 (ring 0 = the node itself) stays in the dense loop instead of its closed form; read at prepare time"},
    {"HSD_KL_TRIM", "KL kinds: 0 = the dense form (P . log P over every column) instead of the sparse-column form with per-tile column trimming, 2 = the sparse columns multiplied without trimming (what a hierarchy too deep for the column lists' LDS gets); read at prepare time"},
    {"HSD_PREP_STAGED", "0 = the operand-preparation kernels read each node's histogram row in place (one thread per node) instead of staging the rows of 64 nodes in LDS with coalesced loads first; same operands either way"},
    {"HSD_DEBUG", "(-DHSD_DEBUG_HOOKS builds only: `make debug`) ablation bits of the pair kernels: 1 no stores, 2 no arithmetic, 16 stores stay in L2; 4 / 8 only in -DHSD_STAMPS builds"},
    {"HSD_SIDE_STREAMS", "0 = the three regions of a row block in sequence instead of side by side (three-launch form only)"},
    {"HSD_TOPK_DEBUG", "(-DHSD_DEBUG_HOOKS builds only) fused k-nearest consumer: 1 = sweep without its candidate epilogue (timing only: the call then fails); 32 = pruned sweep without the early exit of finished waves (A/B)"},
    {"HSD_TOPK_PRUNE", "fused k-nearest consumer over the whole graph: 0 = never the pruned sweep on node-permuted operands; 1 = whenever the kind allows (default: the raw fp64 distance where the persistent kernel runs, the zero-pad distance from ~50k nodes on)"},
    {"HSD_TOPK_GROUPS", "pruned k-nearest sweep: column groups per layer in the lower bound (1..8, default 1); read when the permuted operands are built"},
    {"HSD_TOPK_BAND", "pruned k-nearest sweep: the first launch evaluates every tile against itself and this many successors in the node order (1..4, default 1); read when the permuted operands are built"},
    {"HSD_TOPK_ORDER", "pruned k-nearest sweep: 0 = the node order from host threads instead of from the device (csrc/tkorder.hip); read when the permuted operands are built"},
    {"HSD_TOPK_PHASES", "fused k-nearest consumer: number of equal phases of the sweep"},
    {"HSD_TOPK_GEOM", "fused k-nearest consumer: 0 | 1 = never / always the geometric launch schedule"},
    {"HSD_TOPK_CAP", "fused k-nearest consumer: capacity of a row's candidate list (overflow -> retry path)"},
    {"HSD_TOPK_VERBOSE", "fused k-nearest consumer: per-attempt list statistics to stderr"},
    {"HSD_TOPK_NOMEM", "test hook: rehearses a failed device allocation inside the pruned sweep's setup (the unpruned sweep must take the call)"},
};

int hsd_guard_status(hsd_ctx *ctx, int status, const char *what) noexcept {
    if (ctx) {
        try { ctx->err = what; } catch (...) { ctx->err.clear(); }      // (the assignment itself may need memory)
    }
    return status;
}

extern "C" int hsd_abi_version(void) { return HSD_ABI_VERSION; }

extern "C" const char *hsd_strerror(int status) {
    switch (status) {
        case HSD_OK: return "ok";
        case HSD_ERR_INVALID: return "invalid argument";
        case HSD_ERR_HIP: return "HIP runtime error";
        case HSD_ERR_NO_GRAPH: return "no graph uploaded";
        case HSD_ERR_NO_SIGNATURE: return "signature or operands not built";
        case HSD_ERR_EMPTY_RING: return "Distribution can't be empty.";
        case HSD_ERR_TOO_LARGE: return "problem too large for this kernel";
        case HSD_ERR_COMM: return "RCCL error";
        case HSD_ERR_NOT_NORMALISED: return "The sum of probability distribution must be 1.0.";
        case HSD_ERR_UNSUPPORTED: return "metric is not implemented";
        case HSD_ERR_NOMEM: return "out of host memory";
        case HSD_ERR_INTERNAL: return "internal error (C++ exception stopped at the ABI)";
        default: return "unknown status";
    }
}

extern "C" const char *hsd_last_error(const hsd_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" void hsd_destroy(hsd_ctx *ctx);

extern "C" int hsd_create(int device, hsd_ctx **out) try {
    if (!out) return HSD_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return HSD_ERR_HIP;
    if (device < 0 || device >= count) return HSD_ERR_INVALID;
    hsd_ctx *ctx = new hsd_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&ctx->d_flag), 4 * sizeof(int)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&ctx->d_work), sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&ctx->d_tile_ctr), 32 * sizeof(int)) != hipSuccess ||
        hipMemset(ctx->d_work, 0, sizeof(unsigned long long)) != hipSuccess) {
        hsd_destroy(ctx);          // releases whatever was created (all members start null)
        return HSD_ERR_HIP;
    }
    if (const char *el = getenv("HSD_HOOKS_LIST")) {
        if (atoi(el) != 0)
            for (int h = 0; h < H_COUNT; ++h) {
                const char *v = hsd_hook((HsdHook)h);
                fprintf(stderr, "[hsd hook] %-22s %-8s %s\n", hsd_hook_table[h].name, v ? v : "-", hsd_hook_table[h].doc);
            }
    }
    *out = ctx;
    return HSD_OK;
} HSD_ABI_GUARD(nullptr)

// the pruned k-nearest sweep's node-permuted operand copy (csrc/pair.hip, tk_prepare): released with the operands, and by tk_prepare
// itself before it builds a new one (a failed attempt leaves nothing behind)
void hsd_tk_release(hsd_ctx *ctx) {
    hsd_tk_release_block(ctx);
    hsd_op_free(ctx, ctx->d_tk_order);
    ctx->tk_order_ready = false;
}

void hsd_tk_release_block(hsd_ctx *ctx) {
    // (a block that goes back to the arena may be handed out again at once: nothing in flight may still read it -- hipFree used to see to that by
    //  synchronising the device; the context's stream is where everything that reads operands runs or is joined)
    if (ctx->stream && ctx->d_tk_orig) (void)hipStreamSynchronize(ctx->stream);
    hsd_op_free(ctx, ctx->d_tk_orig);
    hsd_op_free(ctx, ctx->d_tk_X);
    hsd_op_free(ctx, ctx->d_tk_H);
    hsd_op_free(ctx, ctx->d_tk_lo);
    hsd_op_free(ctx, ctx->d_tk_hi);
    hsd_op_free(ctx, ctx->d_tk_scale);
    hsd_op_free(ctx, ctx->d_tk_U);
    hsd_op_free(ctx, ctx->d_tk_inv);
    hsd_op_free(ctx, ctx->d_tk_suf);
    hsd_op_free(ctx, ctx->d_tk_rmax16);
    hsd_op_free(ctx, ctx->d_tk_gl);
    hsd_op_free(ctx, ctx->d_tk_tmaxs);
    hsd_op_free(ctx, ctx->d_tk_list);
    hsd_op_free(ctx, ctx->d_tk_band);
    hsd_op_free(ctx, ctx->d_tk_count);
    hsd_op_free(ctx, ctx->d_tk_tmax);
    hsd_op_free(ctx, ctx->d_tk_band_lo);
    hsd_op_free(ctx, ctx->d_tk_band_hi);
    ctx->tk_ready = false;
    ctx->tk_blk0 = ctx->tk_blk1 = ctx->tk_npos = 0;
    ctx->tk_list_cap = ctx->tk_band_n = ctx->tk_nf = ctx->tk_nt = 0;
}

void hsd_release_operands(hsd_ctx *ctx) {
    if (ctx->stream && (ctx->d_X || ctx->d_H)) (void)hipStreamSynchronize(ctx->stream);      // (see hsd_tk_release_block)
    hsd_op_free(ctx, ctx->d_X);
    hsd_op_free(ctx, ctx->d_Y);
    hsd_op_free(ctx, ctx->d_H);
    hsd_op_free(ctx, ctx->d_inv);
    hsd_op_free(ctx, ctx->d_seg_tab);
    hsd_op_free(ctx, ctx->d_rmax);
    hsd_op_free(ctx, ctx->d_rmax16);
    hsd_op_free(ctx, ctx->d_suf);
    hsd_op_free(ctx, ctx->d_kl_mask);
    ctx->kl_words = ctx->kl_half = ctx->kl_half_alg = 0;
    hsd_tk_release(ctx);
    for (auto &g : ctx->gran_lists) { hsd_op_free(ctx, g.d_gl); hsd_op_free(ctx, g.d_tmax); }
    ctx->gran_lists.clear();
    ctx->gran_order.clear();
    ctx->suf_rows = 0;
    ctx->mode = PAIR_NONE;
}

// ---- operand arena ----------------------------------------------------------------------------------------------------------
// Best fit among the unused blocks of up to twice the request's own block size (requests are rounded up to 256 KiB below 2 MiB, to 2 MiB above: the few
// hundred small vectors of a prepare share a handful of sizes); otherwise a new block.  When the unused blocks together exceed a quarter of the device's
// memory they go back to the driver.
int hsd_pool_alloc(hsd_ctx *ctx, void **p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 1;
    const size_t gran = bytes < ((size_t)2 << 20) ? ((size_t)256 << 10) : ((size_t)2 << 20);
    const size_t want = (bytes + gran - 1) / gran * gran;
    int best = -1;
    for (size_t k = 0; k < ctx->op_pool.size(); ++k) {
        const auto &b = ctx->op_pool[k];
        if (b.used || b.bytes < bytes || b.bytes > 2 * want) continue;
        if (best < 0 || b.bytes < ctx->op_pool[(size_t)best].bytes) best = (int)k;
    }
    if (best >= 0) {
        ctx->op_pool[(size_t)best].used = true;
        *p = ctx->op_pool[(size_t)best].p;
        return HSD_OK;
    }
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, want);
    if (e != hipSuccess) {
        // the unused blocks may be what stands in the way
        (void)hipGetLastError();
        hsd_pool_trim(ctx, false);
        e = hipMalloc(&q, want);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return hsd_fail(ctx, HSD_ERR_HIP, "hipMalloc(%zu bytes): %s", want, hipGetErrorString(e)); }
    ctx->op_pool.push_back({q, want, true});
    *p = q;
    if (hsd_hook(H_PLAN_VERBOSE)) {
        size_t total = 0;
        for (const auto &b : ctx->op_pool) total += b.bytes;
        fprintf(stderr, "[hsd pool] new block of %zu bytes (asked %zu): %zu blocks, %.1f MB\n", want, bytes, ctx->op_pool.size(), (double)total / 1e6);
    }
    return HSD_OK;
}

void hsd_pool_free(hsd_ctx *ctx, void *p) {
    if (!p) return;
    for (auto &b : ctx->op_pool)
        if (b.p == p) {
            b.used = false;
            size_t idle = 0;
            for (const auto &c : ctx->op_pool) if (!c.used) idle += c.bytes;
            if (idle > ctx->prop.totalGlobalMem / 4) hsd_pool_trim(ctx, false);
            return;
        }
    (void)hipFree(p);                                  // (not from the arena)
}

void hsd_pool_trim(hsd_ctx *ctx, bool all) {
    for (size_t k = 0; k < ctx->op_pool.size();) {
        if (all || !ctx->op_pool[k].used) {
            (void)hipFree(ctx->op_pool[k].p);
            ctx->op_pool.erase(ctx->op_pool.begin() + (long)k);
        } else ++k;
    }
}

extern "C" int hsd_comm_destroy_internal(hsd_ctx *ctx);

extern "C" void hsd_destroy(hsd_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    hsd_comm_destroy_internal(ctx);
    hsd_release_operands(ctx);
    hsd_pool_trim(ctx, true);
    hsd_heat_release_internal(ctx);
    hsd_dev_free(ctx->d_rowptr);
    hsd_dev_free(ctx->d_col);
    hsd_dev_free(ctx->d_queue);
    hsd_khop_ms_release(ctx);
    hsd_dev_free(ctx->d_counts);
    hsd_dev_free(ctx->d_node_bin);
    hsd_dev_free(ctx->d_support);
    hsd_dev_free(ctx->d_flag);
    hsd_dev_free(ctx->d_work);
    hsd_dev_free(ctx->d_tile_ctr);
    hsd_dev_free(ctx->d_stage);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    for (auto &t : ctx->tile_lists) (void)hipFree(t.d);
    ctx->tile_lists.clear();
    for (int t = 0; t < 2; ++t) {
        if (ctx->side[t]) (void)hipStreamDestroy(ctx->side[t]);
        if (ctx->ev_join[t]) (void)hipEventDestroy(ctx->ev_join[t]);
    }
    if (ctx->ms_stream) (void)hipStreamSynchronize(ctx->ms_stream);
    for (int t = 0; t < 2; ++t) {
        if (ctx->ms_ev_rings[t]) (void)hipEventDestroy(ctx->ms_ev_rings[t]);
        if (ctx->ms_ev_hist[t]) (void)hipEventDestroy(ctx->ms_ev_hist[t]);
    }
    if (ctx->ms_stream) (void)hipStreamDestroy(ctx->ms_stream);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int hsd_get_device_info(hsd_ctx *ctx, hsd_device_info *out) try {
    HSD_CHECK_CTX(ctx);
    if (!out) return hsd_fail(ctx, HSD_ERR_INVALID, "null output");
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", ctx->prop.name);
    snprintf(out->arch, sizeof out->arch, "%s", ctx->prop.gcnArchName);
    out->compute_units = ctx->prop.multiProcessorCount;
    out->lds_bytes_per_cu = (int32_t)ctx->prop.maxSharedMemoryPerMultiProcessor;
    out->hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
    out->clock_khz = ctx->prop.clockRate;
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_sync(hsd_ctx *ctx) try {
    HSD_CHECK_CTX(ctx);
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_malloc(hsd_ctx *ctx, size_t bytes, void **dptr) try {
    HSD_CHECK_CTX(ctx);
    if (!dptr) return hsd_fail(ctx, HSD_ERR_INVALID, "null output");
    *dptr = nullptr;
    HSD_HIP(ctx, hipMalloc(dptr, bytes ? bytes : 1));
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_free(hsd_ctx *ctx, void *dptr) try {
    HSD_CHECK_CTX(ctx);
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (dptr) HSD_HIP(ctx, hipFree(dptr));
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_memcpy_d2h(hsd_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes) try {
    HSD_CHECK_CTX(ctx);
    HSD_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_memcpy_h2d(hsd_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes) try {
    HSD_CHECK_CTX(ctx);
    HSD_HIP(ctx, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_memset(hsd_ctx *ctx, void *dptr, int value, size_t bytes) try {
    HSD_CHECK_CTX(ctx);
    HSD_HIP(ctx, hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_timer_start(hsd_ctx *ctx) try {
    HSD_CHECK_CTX(ctx);
    HSD_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_timer_stop(hsd_ctx *ctx, float *elapsed_ms) try {
    HSD_CHECK_CTX(ctx);
    HSD_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HSD_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    HSD_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

// ---- graph ------------------------------------------------------------------------------
extern "C" int hsd_set_graph(hsd_ctx *ctx, int64_t n, const int64_t *rowptr, const int32_t *col) try {
    HSD_CHECK_CTX(ctx);
    if (n <= 0 || !rowptr) return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_set_graph: n=%lld", (long long)n);
    if (n >= (int64_t)1 << 31) return hsd_fail(ctx, HSD_ERR_TOO_LARGE, "node ids must fit int32");
    if (rowptr[0] != 0) return hsd_fail(ctx, HSD_ERR_INVALID, "rowptr[0] must be 0");
    const int64_t nnz = rowptr[n];
    if (nnz > 0 && !col) return hsd_fail(ctx, HSD_ERR_INVALID, "null col");
    std::vector<int32_t> deg((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t d = rowptr[i + 1] - rowptr[i];
        if (d < 0) return hsd_fail(ctx, HSD_ERR_INVALID, "rowptr not monotone at %lld", (long long)i);
        deg[(size_t)i] = (int32_t)d;
    }
    for (int64_t e = 0; e < nnz; ++e)
        if (col[e] < 0 || col[e] >= n)
            return hsd_fail(ctx, HSD_ERR_INVALID, "col[%lld]=%d out of range", (long long)e, col[e]);
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hsd_dev_free(ctx->d_rowptr);
    hsd_dev_free(ctx->d_col);
    hsd_dev_free(ctx->d_queue);
    hsd_khop_ms_release(ctx);
    hsd_dev_free(ctx->d_counts);
    hsd_dev_free(ctx->d_node_bin);
    ctx->h_node_bin.clear();
    hsd_dev_free(ctx->d_support);
    hsd_release_operands(ctx);
    hsd_heat_release_internal(ctx);
    ctx->hop = -1;
    ctx->n_bins = 0;
    ctx->queue_blocks = 0;
    ctx->n = 0;                   // no graph until the upload below has succeeded
    ctx->nnz = 0;
    int rc;
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_rowptr, (size_t)n + 1))) return rc;
    if ((rc = hsd_dev_alloc(ctx, &ctx->d_col, (size_t)nnz))) return rc;
    HSD_HIP(ctx, hipMemcpyAsync(ctx->d_rowptr, rowptr, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    if (nnz)
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_col, col, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n = n;
    ctx->nnz = nnz;
    ctx->h_degree.swap(deg);
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

// ---- resident-matrix checks ---------------------------------------------------------------
__device__ inline double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ inline double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
__device__ inline double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
    return v;
}

// Tile-wise: a workgroup owns a strip of 64 rows and a group of MS_GROUP column tiles (64 columns each).  Inside the
// block's square part (columns [row0, row0 + rows): the entries whose mirror image is also in the block) the tile
// (ti, tj), tj > ti, is handled together with its mirror (tj, ti) -- both are read row-wise (coalesced), the mirror is
// transposed through LDS -- so every entry is read exactly ONCE: an 80 GB matrix costs 80 GB of reads (the round-2 kernel
// walked the mirror as a column, 8 bytes out of every line: 967 GB).  partial[(strip * groups + group) * 5 + q].
constexpr int MS_TILE = 64, MS_GROUP = 16;
__global__ __launch_bounds__(256) void matrix_stats_kernel(const double *__restrict__ D, int64_t rows, int64_t cols,
                                                          int64_t ld, int64_t row0, int64_t sq_tiles, int64_t side_tiles_left,
                                                          int64_t side_tiles, int groups, double *__restrict__ partial) {
    __shared__ double tr[MS_TILE][MS_TILE + 1];
    __shared__ double red[4][5];
    const int64_t ti = blockIdx.x;
    const int tid = threadIdx.x, c = tid & 63, rq = tid >> 6;      // thread: column c, rows rq, rq + 4, ...
    double s = 0.0, mx = -INFINITY, mn = INFINITY, asym = 0.0, diag = 0.0;
    const int64_t i_lo = ti * MS_TILE;
    // column tiles are numbered: [0, side_tiles_left) left of the square, then the sq_tiles tiles of the square (aligned
    // with the row strips), then the tiles right of it
    const int64_t total = side_tiles + sq_tiles;
    const int64_t t_end = total < (int64_t)(blockIdx.y + 1) * MS_GROUP ? total : (int64_t)(blockIdx.y + 1) * MS_GROUP;
    for (int64_t t = (int64_t)blockIdx.y * MS_GROUP; t < t_end; ++t) {
        int64_t j_lo, j_hi;
        bool square = false;
        if (t < side_tiles_left) {
            // (a block wholly outside the square, row0 >= cols, has only left tiles: they end at the last column, not at row0)
            const int64_t left_end = row0 < cols ? row0 : cols;
            j_lo = t * MS_TILE; j_hi = left_end < j_lo + MS_TILE ? left_end : j_lo + MS_TILE;
        }
        else if (t < side_tiles_left + sq_tiles) {
            const int64_t tj = t - side_tiles_left;
            if (tj < ti) continue;                                  // read by the strip that owns its mirror
            square = true;
            j_lo = row0 + tj * MS_TILE; j_hi = row0 + rows < j_lo + MS_TILE ? row0 + rows : j_lo + MS_TILE;
        } else {
            j_lo = row0 + rows + (t - side_tiles_left - sq_tiles) * MS_TILE; j_hi = cols < j_lo + MS_TILE ? cols : j_lo + MS_TILE;
        }
        const int64_t tj = square ? t - side_tiles_left : -1;
        const bool mirror = square && tj > ti;
        if (mirror) {
            // the mirror tile: rows tj*64.., columns row0 + ti*64..; tr[r][c] = D[tj*64 + r][row0 + ti*64 + c]
            __syncthreads();
            for (int r = rq; r < MS_TILE; r += 4) {
                const int64_t i = tj * MS_TILE + r, j = row0 + i_lo + c;
                double v = 0.0;
                const bool ok = i < rows && j < row0 + rows;
                if (ok) {
                    v = D[i * ld + j];
                    s += v; mx = fmax(mx, v); mn = fmin(mn, v);
                }
                tr[r][c] = v;
            }
            __syncthreads();
        }
        for (int r = rq; r < MS_TILE; r += 4) {
            const int64_t i = i_lo + r, j = j_lo + c;
            if (i >= rows || j >= j_hi) continue;
            const double v = D[i * ld + j];
            s += v; mx = fmax(mx, v); mn = fmin(mn, v);
            if (square) {
                if (j == row0 + i) diag = fmax(diag, fabs(v));
                if (mirror) asym = fmax(asym, fabs(v - tr[c][r]));
            }
        }
        if (square && !mirror) {
            // a tile on the diagonal is its own mirror: transpose it through LDS as well
            __syncthreads();
            for (int r = rq; r < MS_TILE; r += 4) {
                const int64_t i = i_lo + r, j = j_lo + c;
                tr[r][c] = (i < rows && j < j_hi) ? D[i * ld + j] : 0.0;
            }
            __syncthreads();
            for (int r = rq; r < MS_TILE; r += 4) {
                const int64_t i = i_lo + r, j = j_lo + c;
                if (i < rows && j < j_hi && c > r) asym = fmax(asym, fabs(tr[r][c] - tr[c][r]));
            }
        }
    }
    s = wave_sum(s); mx = wave_max(mx); mn = wave_min(mn); asym = wave_max(asym); diag = wave_max(diag);
    const int w = tid >> 6;
    if ((tid & 63) == 0) { red[w][0] = s; red[w][1] = diag; red[w][2] = asym; red[w][3] = mx; red[w][4] = mn; }
    __syncthreads();
    if (tid == 0) {
        double *p = partial + ((int64_t)blockIdx.x * groups + blockIdx.y) * 5;
        p[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        p[1] = fmax(fmax(red[0][1], red[1][1]), fmax(red[2][1], red[3][1]));
        p[2] = fmax(fmax(red[0][2], red[1][2]), fmax(red[2][2], red[3][2]));
        p[3] = fmax(fmax(red[0][3], red[1][3]), fmax(red[2][3], red[3][3]));
        p[4] = fmin(fmin(red[0][4], red[1][4]), fmin(red[2][4], red[3][4]));
    }
}

extern "C" int hsd_matrix_stats(hsd_ctx *ctx, const double *dev_block, int64_t rows, int64_t cols, int64_t ld,
                                int64_t row0, double *stats) try {
    HSD_CHECK_CTX(ctx);
    if (!dev_block || !stats || rows <= 0 || cols <= 0 || ld < cols || row0 < 0)
        return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_matrix_stats: bad arguments");
    // the square part: columns [row0, row0 + rows) clipped to the matrix
    const int64_t sq = std::max<int64_t>(0, std::min<int64_t>(rows, cols - row0));
    if (sq < rows && row0 < cols) return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_matrix_stats: the block's square part is cut by the last column");
    const int64_t left = std::min<int64_t>(row0, cols), right = std::max<int64_t>(0, cols - row0 - sq);
    const int64_t tl = (left + MS_TILE - 1) / MS_TILE, trr = (right + MS_TILE - 1) / MS_TILE, ts = (sq + MS_TILE - 1) / MS_TILE;
    const int64_t strips = (rows + MS_TILE - 1) / MS_TILE;
    const int groups = (int)((tl + ts + trr + MS_GROUP - 1) / MS_GROUP);
    if (groups > 65535) return hsd_fail(ctx, HSD_ERR_TOO_LARGE, "hsd_matrix_stats: too many column tiles");
    double *d_part = nullptr;
    const size_t np = (size_t)strips * groups;
    int rc = hsd_dev_alloc(ctx, &d_part, np * 5);
    if (rc) return rc;
    hipLaunchKernelGGL(matrix_stats_kernel, dim3((unsigned)strips, (unsigned)groups), dim3(256), 0, ctx->stream, dev_block, rows, cols, ld,
                       row0, ts, tl, tl + trr, groups, d_part);
    std::vector<double> part(np * 5);
    hipError_t e = hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_part);
    if (e != hipSuccess) return hsd_fail(ctx, HSD_ERR_HIP, "hsd_matrix_stats: %s", hipGetErrorString(e));
    // partial sums added in long double, in a fixed order: a stable checksum
    long double s = 0.0L;
    double diag = 0.0, asym = 0.0, mx = -INFINITY, mn = INFINITY;
    for (size_t r = 0; r < np; ++r) {
        s += part[r * 5];
        diag = std::max(diag, part[r * 5 + 1]);
        asym = std::max(asym, part[r * 5 + 2]);
        mx = std::max(mx, part[r * 5 + 3]);
        mn = std::min(mn, part[r * 5 + 4]);
    }
    stats[0] = (double)s; stats[1] = diag; stats[2] = asym; stats[3] = mx; stats[4] = mn;
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

__global__ __launch_bounds__(256) void matrix_add_kernel(double *__restrict__ dst, int64_t ld_dst, const double *__restrict__ src,
                                                        int64_t ld_src, int64_t cols) {
    const int64_t r = blockIdx.y;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < cols; c += (int64_t)gridDim.x * 256) dst[r * ld_dst + c] += src[r * ld_src + c];
}

extern "C" int hsd_matrix_add(hsd_ctx *ctx, double *dev_dst, int64_t ld_dst, const double *dev_src, int64_t ld_src, int64_t rows,
                              int64_t cols) try {
    HSD_CHECK_CTX(ctx);
    if (!dev_dst || !dev_src || rows < 0 || cols < 0 || ld_dst < cols || ld_src < cols)
        return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_matrix_add: bad arguments");
    if (rows == 0 || cols == 0) return HSD_OK;
    for (int64_t r0 = 0; r0 < rows; r0 += 32768) {           // gridDim.y limit
        const int64_t nr = std::min<int64_t>(32768, rows - r0);
        hipLaunchKernelGGL(matrix_add_kernel, dim3((unsigned)std::min<int64_t>((cols + 255) / 256, 64), (unsigned)nr), dim3(256), 0,
                           ctx->stream, dev_dst + r0 * ld_dst, ld_dst, dev_src + r0 * ld_src, ld_src, cols);
    }
    HSD_HIP(ctx, hipGetLastError());
    return HSD_OK;
} HSD_ABI_GUARD(ctx)
