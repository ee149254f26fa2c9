// K2 (host side): per-node per-hop value histograms, resident on the device.
// The histogram accumulation itself is fused into the k-hop sweep (khop.hip, `counts`).
#include "common.hpp"

#include <algorithm>
#include <cmath>

static int alloc_counts(hsd_ctx *ctx, int hop, int n_bins) {
    const int64_t rows = ctx->world > 1 ? ctx->rows_per_rank * ctx->world : ctx->n;
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hsd_dev_free(ctx->d_counts);
    hsd_release_operands(ctx);
    const size_t cnt = (size_t)rows * (hop + 1) * n_bins;
    int rc = hsd_dev_alloc(ctx, &ctx->d_counts, cnt);
    if (rc) return rc;
    HSD_HIP(ctx, hipMemsetAsync(ctx->d_counts, 0, cnt * sizeof(uint32_t), ctx->stream));
    ctx->n_rows_alloc = rows;
    ctx->hop = hop;
    ctx->n_bins = n_bins;
    return HSD_OK;
}

extern "C" int hsd_signature_build(hsd_ctx *ctx, int hop, const int32_t *node_bin, int n_bins, const double *support,
                                   int64_t r0, int64_t r1) try {
    HSD_CHECK_CTX(ctx);
    if (ctx->n <= 0) return hsd_fail(ctx, HSD_ERR_NO_GRAPH, "no graph uploaded");
    if (hop < 0 || hop >= HSD_MAX_LAYERS) return hsd_fail(ctx, HSD_ERR_INVALID, "hop=%d out of range", hop);
    if (!node_bin || !support || n_bins <= 0) return hsd_fail(ctx, HSD_ERR_INVALID, "hsd_signature_build: bad arguments");
    if (r0 < 0 || r1 > ctx->n || r0 > r1) return hsd_fail(ctx, HSD_ERR_INVALID, "row range out of bounds");
    for (int b = 0; b + 1 < n_bins; ++b)
        if (!(support[b] < support[b + 1])) return hsd_fail(ctx, HSD_ERR_INVALID, "support must be strictly ascending");
    for (int64_t i = 0; i < ctx->n; ++i)
        if (node_bin[i] < 0 || node_bin[i] >= n_bins)
            return hsd_fail(ctx, HSD_ERR_INVALID, "node_bin[%lld]=%d outside [0,%d)", (long long)i, node_bin[i], n_bins);
    int rc;
    // keep the counts tensor if the shape is unchanged (several row blocks may be built in turn)
    const int64_t rows = ctx->world > 1 ? ctx->rows_per_rank * ctx->world : ctx->n;
    if (!ctx->d_counts || ctx->hop != hop || ctx->n_bins != n_bins || ctx->n_rows_alloc != rows) {
        if ((rc = alloc_counts(ctx, hop, n_bins))) return rc;
    } else {
        hsd_release_operands(ctx);
    }
    // the bin of every node and the support stay on the device between builds over the same values (row blocks
    // built in turn, repeated builds): only a change is uploaded
    const bool same_bins = ctx->d_node_bin && ctx->d_support && ctx->h_node_bin.size() == (size_t)ctx->n &&
                           ctx->support.size() == (size_t)n_bins && std::equal(node_bin, node_bin + ctx->n, ctx->h_node_bin.begin()) &&
                           std::equal(support, support + n_bins, ctx->support.begin());
    if (!same_bins) {
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hsd_dev_free(ctx->d_node_bin);
        hsd_dev_free(ctx->d_support);
        ctx->h_node_bin.clear();
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_node_bin, (size_t)ctx->n))) return rc;
        if ((rc = hsd_dev_alloc(ctx, &ctx->d_support, (size_t)n_bins))) return rc;
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_node_bin, node_bin, (size_t)ctx->n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        HSD_HIP(ctx, hipMemcpyAsync(ctx->d_support, support, (size_t)n_bins * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->h_node_bin.assign(node_bin, node_bin + ctx->n);
    }
    ctx->support.assign(support, support + n_bins);
    ctx->support_integral = true;
    for (int b = 0; b < n_bins; ++b)
        if (support[b] < 0 || support[b] != std::floor(support[b]) || support[b] > 4.0e9) ctx->support_integral = false;
    if (r1 > r0) {
        rc = hsd_launch_khop(ctx, hop, r0, r1, true, nullptr, nullptr, nullptr);
        if (rc) return rc;
    }
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_signature_build_degree(hsd_ctx *ctx, int hop, int64_t r0, int64_t r1) try {
    HSD_CHECK_CTX(ctx);
    if (ctx->n <= 0) return hsd_fail(ctx, HSD_ERR_NO_GRAPH, "no graph uploaded");
    // distinct degrees in ascending order = the support; a node's bin is the rank of its degree
    int32_t dmax = 0;
    for (int32_t d : ctx->h_degree) dmax = std::max(dmax, d);
    std::vector<int32_t> rank((size_t)dmax + 1, 0);
    for (int32_t d : ctx->h_degree) rank[(size_t)d] = 1;
    std::vector<double> support;
    for (int32_t d = 0; d <= dmax; ++d)
        if (rank[(size_t)d]) {
            rank[(size_t)d] = (int32_t)support.size();
            support.push_back((double)d);
        }
    std::vector<int32_t> bin((size_t)ctx->n);
    for (int64_t i = 0; i < ctx->n; ++i) bin[(size_t)i] = rank[(size_t)ctx->h_degree[(size_t)i]];
    return hsd_signature_build(ctx, hop, bin.data(), (int)support.size(), support.data(), r0, r1);
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_signature_dims(hsd_ctx *ctx, int *hop, int *n_bins) try {
    HSD_CHECK_CTX(ctx);
    if (!ctx->d_counts) return hsd_fail(ctx, HSD_ERR_NO_SIGNATURE, "no signature built");
    if (hop) *hop = ctx->hop;
    if (n_bins) *n_bins = ctx->n_bins;
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_signature_support(hsd_ctx *ctx, double *support) try {
    HSD_CHECK_CTX(ctx);
    if (!ctx->d_counts) return hsd_fail(ctx, HSD_ERR_NO_SIGNATURE, "no signature built");
    if (!support) return hsd_fail(ctx, HSD_ERR_INVALID, "null output");
    std::copy(ctx->support.begin(), ctx->support.end(), support);
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

static int rows_ok(hsd_ctx *ctx, int64_t r0, int64_t r1) {
    if (!ctx->d_counts) return hsd_fail(ctx, HSD_ERR_NO_SIGNATURE, "no signature built");
    if (r0 < 0 || r1 > ctx->n || r0 > r1) return hsd_fail(ctx, HSD_ERR_INVALID, "row range out of bounds");
    return HSD_OK;
}

extern "C" int hsd_signature_get_counts(hsd_ctx *ctx, int64_t r0, int64_t r1, uint32_t *counts) try {
    HSD_CHECK_CTX(ctx);
    int rc = rows_ok(ctx, r0, r1);
    if (rc) return rc;
    if (r0 == r1) return HSD_OK;
    if (!counts) return hsd_fail(ctx, HSD_ERR_INVALID, "null output");
    const size_t row = (size_t)(ctx->hop + 1) * ctx->n_bins;
    HSD_HIP(ctx, hipMemcpyAsync(counts, ctx->d_counts + (size_t)r0 * row, (size_t)(r1 - r0) * row * sizeof(uint32_t),
                                hipMemcpyDeviceToHost, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_signature_set_counts(hsd_ctx *ctx, int64_t r0, int64_t r1, const uint32_t *counts) try {
    HSD_CHECK_CTX(ctx);
    int rc = rows_ok(ctx, r0, r1);
    if (rc) return rc;
    if (r0 == r1) return HSD_OK;
    if (!counts) return hsd_fail(ctx, HSD_ERR_INVALID, "null input");
    const size_t row = (size_t)(ctx->hop + 1) * ctx->n_bins;
    hsd_release_operands(ctx);
    HSD_HIP(ctx, hipMemcpyAsync(ctx->d_counts + (size_t)r0 * row, counts, (size_t)(r1 - r0) * row * sizeof(uint32_t),
                                hipMemcpyHostToDevice, ctx->stream));
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HSD_OK;
} HSD_ABI_GUARD(ctx)
