/*
 * hsd.h -- C ABI of the MI355X-native hierarchical-structural-distance engine.
 *
 * This is the drop-in boundary.  The reference (Sngunfei/HSD,
 * pure Python) has no FFI of its own; these entry points are what its Python
 * methods bind to through ctypes.  Each declaration cites
 * the reference code (file:line under /root/reference) whose work it replaces.
 *
 * Conventions
 *   - every function returns an int status: HSD_OK (0) or a negative HSD_ERR_*;
 *     hsd_last_error(ctx) returns a human-readable message for the last failure
 *   - the caller allocates every output; the library never returns owned
 *     pointers except the opaque context and hsd_malloc() device blocks
 *   - no exceptions cross the boundary; no torch / numpy types in signatures
 *   - one context per GPU; calls on one context are not re-entrant
 *   - "host" pointers are ordinary process memory, "device" pointers come from
 *     hsd_malloc() on the same context
 *   - node indices are 0-based positions in the reference's node order
 *     (networkx insertion order, model/HSD.py:37, tools/util.py:11-24)
 */
#ifndef HSD_H
#define HSD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: hsd_pair_shard, status codes -10 / -11 (round 4); hsd_pair_topk takes the pruned sweep for row blocks (round 5: same signature) */
#define HSD_ABI_VERSION 2

/* status codes */
#define HSD_OK                    0
#define HSD_ERR_INVALID          -1  /* bad argument                                   */
#define HSD_ERR_HIP              -2  /* HIP runtime failure (message has the detail)   */
#define HSD_ERR_NO_GRAPH         -3  /* hsd_set_graph() has not been called            */
#define HSD_ERR_NO_SIGNATURE     -4  /* signature / operand tensors not built yet      */
#define HSD_ERR_EMPTY_RING       -5  /* scipy: ValueError("Distribution can't be empty.") (model/HSD.py:111) */
#define HSD_ERR_TOO_LARGE        -6  /* problem exceeds an on-chip or 32-bit limit     */
#define HSD_ERR_COMM             -7  /* RCCL failure                                   */
#define HSD_ERR_NOT_NORMALISED   -8  /* tools/metrics.py:49-51 ValueError              */
#define HSD_ERR_UNSUPPORTED      -9  /* tools/metrics.py:167 NotImplementedError       */
#define HSD_ERR_NOMEM           -10  /* host allocation failed inside the library      */
#define HSD_ERR_INTERNAL        -11  /* a C++ exception was stopped at the boundary    */

/* metric names accepted by tools.metrics.calculate_distance that are on the hot path
 * (tools/metrics.py:151-192) */
#define HSD_METRIC_WASSERSTEIN    0
#define HSD_METRIC_KL             1
#define HSD_METRIC_SYMMETRIC_KL   2

/* how two ring samples of unequal length are compared */
#define HSD_ALIGN_NONE            0  /* scipy on the raw samples: Variant A, model/HSD.py:111       */
#define HSD_ALIGN_ZEROPAD         1  /* tools/metrics.py:18-36 zero-pad + sort: Variant B, :157     */

typedef struct hsd_ctx hsd_ctx;

typedef struct hsd_device_info {
    char     name[128];
    char     arch[32];
    int32_t  compute_units;
    int32_t  lds_bytes_per_cu;
    int64_t  hbm_bytes;
    int32_t  clock_khz;
} hsd_device_info;

/* ---- context -------------------------------------------------------------------------- */
int         hsd_abi_version(void);
int         hsd_create(int device, hsd_ctx **out);            /* owns one HIP stream on `device`   */
void        hsd_destroy(hsd_ctx *ctx);
const char *hsd_last_error(const hsd_ctx *ctx);
const char *hsd_strerror(int status);
int         hsd_get_device_info(hsd_ctx *ctx, hsd_device_info *out);
int         hsd_sync(hsd_ctx *ctx);                            /* wait for the context's stream     */

/* device memory owned by the caller (distance matrices stay resident: N=100k -> 80 GB) */
int hsd_malloc(hsd_ctx *ctx, size_t bytes, void **dptr);
int hsd_free(hsd_ctx *ctx, void *dptr);
int hsd_memcpy_d2h(hsd_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);
int hsd_memcpy_h2d(hsd_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int hsd_memset(hsd_ctx *ctx, void *dptr, int value, size_t bytes);

/* HIP-event stopwatch on the context's stream (bench.py times kernels with this) */
int hsd_timer_start(hsd_ctx *ctx);
int hsd_timer_stop(hsd_ctx *ctx, float *elapsed_ms);           /* synchronises */

/* ---- graph ---------------------------------------------------------------------------- */
/* Replaces the dense A / L construction of HSD.__init__ (model/HSD.py:33-34): the graph is
 * uploaded once as CSR.  rowptr[n+1] (int64), col[rowptr[n]] (int32, both directions of every
 * undirected edge, ascending within a row).  Host pointers; copied. */
int hsd_set_graph(hsd_ctx *ctx, int64_t n, const int64_t *rowptr, const int32_t *col);

/* ---- K1: k-hop ring sweep --------------------------------------------------------------
 * Replaces tools.get_node_hierarchical_structure / get_hierarchical_representation
 * (tools/hierarchy.py:16-38) for sources [r0, r1): ring k = nodes first reached at BFS depth k,
 * always hop+1 rings, empty once the component is exhausted. */
int hsd_khop_sizes(hsd_ctx *ctx, int hop, int64_t r0, int64_t r1,
                   int32_t *sizes /* host, (r1-r0)*(hop+1) */);
/* ring members, ascending inside a ring.  offsets[(r1-r0)*(hop+1)+1] is the exclusive prefix
 * sum of the sizes above (computed by the caller); members has offsets[last] entries. */
int hsd_khop_members(hsd_ctx *ctx, int hop, int64_t r0, int64_t r1,
                     const int64_t *offsets /* host */, int32_t *members /* host */);

/* ---- K2: per-node per-hop histogram signatures -----------------------------------------
 * Every node v carries a value support[node_bin[v]]; the signature of source i is, per ring k,
 * the histogram of the values of ring_k(i) over the sorted support -- a lossless encoding of
 * the sample lists the reference builds in get_hierarchical_coeffcients (model/HSD.py:71-83)
 * when the per-node value does not depend on the source (degree signature: value = degree;
 * ring sizes of get_nodes_hierarchical_degree, model/HSD.py:87-94, are the histogram totals).
 * Counts stay resident on the device as uint32 [n][hop+1][n_bins]; rows outside [r0, r1) are
 * untouched (filled by hsd_signature_allgather or hsd_signature_set_counts). */
int hsd_signature_build(hsd_ctx *ctx, int hop, const int32_t *node_bin /* host, n */,
                        int n_bins, const double *support /* host, n_bins, ascending */,
                        int64_t r0, int64_t r1);
/* value = number of neighbours; support = distinct degrees of the graph */
int hsd_signature_build_degree(hsd_ctx *ctx, int hop, int64_t r0, int64_t r1);
int hsd_signature_dims(hsd_ctx *ctx, int *hop, int *n_bins);
int hsd_signature_support(hsd_ctx *ctx, double *support /* host, n_bins */);
int hsd_signature_get_counts(hsd_ctx *ctx, int64_t r0, int64_t r1,
                             uint32_t *counts /* host, (r1-r0)*(hop+1)*n_bins */);
int hsd_signature_set_counts(hsd_ctx *ctx, int64_t r0, int64_t r1, const uint32_t *counts);

/* ---- C1: multi-GPU exchange (one process per GPU, RCCL over xGMI) ----------------------
 * No reference analogue (the closest is the Pool gather, model/HSD.py:128-130).  Rank g owns
 * sources [g*rows_per_rank, min(n, (g+1)*rows_per_rank)), rows_per_rank = ceil(n / world) rounded
 * up to a multiple of 128 (tile-aligned blocks). */
int hsd_comm_unique_id(void *id128 /* 128 bytes, host */);
int hsd_comm_init(hsd_ctx *ctx, int world, int rank, const void *id128);
int hsd_comm_shard(hsd_ctx *ctx, int64_t *r0, int64_t *r1);    /* this rank's row block        */
int hsd_signature_allgather(hsd_ctx *ctx);                      /* one ncclAllGather of counts  */

/* ---- K3: all-pairs distance ------------------------------------------------------------
 * hsd_pair_prepare turns the resident counts into the operand matrices of one distance:
 *   WASSERSTEIN / ALIGN_NONE    sum_{k=layer_lo..layer_hi} scipy.stats.wasserstein_distance on
 *                               the raw ring samples (model/HSD.py:108-111; _cdf_distance p=1)
 *   WASSERSTEIN / ALIGN_ZEROPAD the same through tools.metrics.calculate_distance
 *                               (tools/metrics.py:18-36,186-190): shorter sample zero-padded
 *   KL / SYMMETRIC_KL           tools.metrics.KL_divergence (tools/metrics.py:74-88) on the
 *                               per-ring histograms smoothed as (c + alpha)/(n + alpha*bins);
 *                               kl_sorted != 0 applies calculate_distance's ascending sort
 *                               of both vectors first (tools/metrics.py:29-30)
 * Layers follow the reference: Variant A uses 0..hop (model/HSD.py:108), Variant B 0..hop-1
 * (model/HSD.py:148).  HSD_ERR_EMPTY_RING is returned for ALIGN_NONE if any compared ring of
 * any node is empty, as scipy raises. */
int hsd_pair_prepare(hsd_ctx *ctx, int metric, int align, int layer_lo, int layer_hi,
                     double kl_alpha, int kl_sorted);
/* D[r0:r1, 0:n] -> out[(i - r0) * ld + j], fp64, symmetric, zero diagonal
 * (model/HSD.py:102-112,132-134).  out_on_device != 0: `out` is a device pointer and the call
 * is asynchronous on the context's stream; otherwise `out` is host memory. */
int hsd_pair_rows(hsd_ctx *ctx, int64_t r0, int64_t r1, double *out, int64_t ld,
                  int out_on_device);
/* Fused k-nearest-neighbour consumer of the prepared distance: what
 * KNN_evaluate(dists, labels, metric="precomputed") (main.py:29, tools/evaluate.py:61-69) reads
 * from the matrix -- for every node i in [r0, r1) the k other nodes with the smallest D[i, j],
 * ascending, ties by smaller index -- without writing the N^2 matrix.  Symmetric distances only
 * (wasserstein, symmetric_kl).  idx_out / dist_out: host, (r1-r0) * k. */
int hsd_pair_topk(hsd_ctx *ctx, int k, int64_t r0, int64_t r1, int32_t *idx_out, double *dist_out);
/* Multi-GPU: rank's rows [*r0, *r1) of the PREPARED distance when `world` ranks share the matrix in contiguous row blocks of equal
 * planned cost (boundaries on whole 128-row tiles; every rank computes the same partition).  Equal heights -- hsd_comm_shard, the
 * partition of the sweep -- leave the rank that holds a scale-free graph's hub rows ~15 % behind the others; the blocks still tile
 * [0, n) in rank order, as the Pool's rows do (model/HSD.py:120-130). */
int hsd_pair_shard(hsd_ctx *ctx, int world, int rank, int64_t *r0, int64_t *r1);
/* algorithmic size of the prepared operands: columns per node and bytes per element */
int hsd_pair_operand_dims(hsd_ctx *ctx, int *n_columns, int *bytes_per_element);
/* Work of the last hsd_pair_rows call: operand columns (words) x node pairs the kernel evaluated, counted by the
 * kernel itself.  Below pairs x hsd_pair_operand_dims' columns when tile pairs took the shorter rank form of a
 * zero-pad layer (csrc/pair.hip, SegDesc); the measurement harness prices the vector-ALU rate with it. */
int hsd_pair_work(hsd_ctx *ctx, double *column_pairs);

/* ---- K4 and the reference's own coefficient signatures ----------------------------------
 * psi = exp(-s L) is dense N x N: this path serves graphs whose N^2 fits in HBM; the histogram
 * path above is the one that scales. */
/* Laplacian L = D - A as CSR with explicit diagonal (networkx.laplacian_matrix, model/HSD.py:34);
 * host pointers, copied. */
int hsd_set_laplacian(hsd_ctx *ctx, int64_t n, const int64_t *rowptr, const int32_t *col,
                      const double *val);
/* Rows [c0, c1) of the heat-kernel wavelet matrix by the Chebyshev recurrence of
 * pygsp.filters.approximations.cheby_op applied to the unit impulses (model/HSD.py:49-59), with
 * caller-supplied coefficients coeff[0..order] on [0, lmax]; then the reference's threshold
 * (model/HSD.py:65-66): keep x iff x > threshold, else 0 (threshold < 0 disables it).  A high
 * order reproduces the exact branch (model/HSD.py:61-63) to rounding.  out[(c - c0) * ld + j]. */
int hsd_heat_wavelets(hsd_ctx *ctx, const double *coeff, int order, double lmax, double threshold,
                      int64_t c0, int64_t c1, double *out, int64_t ld, int out_on_device);
/* The same for n_scales heat scales at once: the Chebyshev polynomials do not depend on the scale, so ONE
 * recurrence serves every coefficient set coeff[s][0..order]; outs[s] (host array of n_scales DEVICE pointers,
 * each n x ld doubles) receives the whole matrix psi_s.  Replaces the per-scale loops of main.py:24-28 and
 * model/multiscale_HSD.py:104-110. */
int hsd_heat_wavelets_multi(hsd_ctx *ctx, const double *coeff, int n_scales, int order, double lmax,
                            double threshold, double *const *outs, int64_t ld);

/* MultiHSD.get_triple (model/multiscale_HSD.py:45-61): sums[(i-r0)*(hop+1)+k] = sum of psi[i, v]
 * over ring_k(i) (mean = sum / size, [0, 0] for an empty ring); psi_rows_dev holds rows r0..r1 of
 * psi (device, leading dimension ld); sizes may be NULL. */
int hsd_ring_stats(hsd_ctx *ctx, int hop, const double *psi_rows_dev, int64_t ld, int64_t r0,
                   int64_t r1, double *sums /* host */, int32_t *sizes /* host or NULL */);
/* MultiHSD.embed / parallel_embed fused (model/multiscale_HSD.py:35-61,76-98): the Chebyshev
 * polynomials of the recurrence do not depend on the heat scale, so ONE sweep per block of sources
 * serves all scales (coeff[s][0..order], s < n_scales <= 4096); psi is never materialised.
 * sums[(s * (r1-r0) + (i-r0)) * (hop+1) + k] = sum of threshold(psi_s[i, v]) over ring_k(i). */
int hsd_multiscale_ring_stats(hsd_ctx *ctx, int hop, const double *coeff, int n_scales, int order,
                              double lmax, double threshold, int64_t r0, int64_t r1,
                              double *sums /* host */, int32_t *sizes /* host or NULL */);
/* HSD.get_hierarchical_coeffcients (model/HSD.py:71-83): per node, per ring, the coefficients
 * psi[i, ring_k(i)] -- kept on the device as ragged ascending samples.  psi_dev: full N x N. */
int hsd_heat_signature_build(hsd_ctx *ctx, int hop, const double *psi_dev, int64_t ld);
/* Variant A on those samples (model/HSD.py:98-114): sum_{k=layer_lo..layer_hi} of
 * scipy.stats.wasserstein_distance; HSD_ERR_EMPTY_RING like scipy on an empty ring. */
int hsd_pair_heat_rows(hsd_ctx *ctx, int layer_lo, int layer_hi, int64_t r0, int64_t r1,
                       double *out, int64_t ld, int out_on_device);
/* Variant B exactly as written (model/HSD.py:140-161): layers 0..hop-1, q read from the row of the
 * smaller index, tools.metrics.calculate_distance(.., 'wasserstein') per layer. */
int hsd_pair_heat_rowform(hsd_ctx *ctx, const double *psi_dev, int64_t ldpsi, int64_t r0,
                          int64_t r1, double *out, int64_t ld, int out_on_device);

/* tools.metrics.calculate_distance(p, q, metric) for one pair of samples (tools/metrics.py:151-192):
 * zero-pad alignment + ascending sort, both empty -> 0.0, then scipy W1 / KL_divergence /
 * symmetric KL; HSD_ERR_NOT_NORMALISED where the reference raises ValueError (:49-51).
 * p, q: host pointers. */
int hsd_sample_distance(hsd_ctx *ctx, const double *p, int64_t np, const double *q, int64_t nq,
                        int metric, double *out /* host */);

/* checks on a resident matrix block without moving it to the host:
 * stats[0] = sum of all entries, stats[1] = max |D[i,i]| over the block's diagonal entries,
 * stats[2] = max |D[i,j] - D[j,i]| over the square part [row0,row0+rows)^2, stats[3] = max entry,
 * stats[4] = min entry */
int hsd_matrix_stats(hsd_ctx *ctx, const double *dev_block, int64_t rows, int64_t cols,
                     int64_t ld, int64_t row0, double *stats /* host, 5 */);

/* dst[r, c] += src[r, c] on two resident blocks (rows x cols, leading dimensions ld_dst / ld_src).
 * Replaces the Python accumulation over scales of MultiHSD.parallel_calculate_structural_distance
 * (model/multiscale_HSD.py:112-117) without moving the per-scale matrices to the host. */
int hsd_matrix_add(hsd_ctx *ctx, double *dev_dst, int64_t ld_dst, const double *dev_src, int64_t ld_src,
                   int64_t rows, int64_t cols);

#ifdef __cplusplus
}
#endif
#endif /* HSD_H */
