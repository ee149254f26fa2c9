// C1: the single exchange step of the multi-GPU path -- one ncclAllGather (RCCL over xGMI) of
// the per-rank block of histogram counts.  One process per GPU; the 128-byte unique id is
// created on rank 0 and carried to the other ranks by the host program (bench.py uses a
// torch.distributed store for that; any side channel works).
#include "common.hpp"

#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

static_assert(sizeof(ncclUniqueId) == 128, "hsd.h documents a 128-byte id");

extern "C" int hsd_comm_destroy_internal(hsd_ctx *ctx) try {
    if (ctx && ctx->comm) {
        ncclCommDestroy(reinterpret_cast<ncclComm_t>(ctx->comm));
        ctx->comm = nullptr;
    }
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_comm_unique_id(void *id128) try {
    if (!id128) return HSD_ERR_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return HSD_ERR_COMM;
    memcpy(id128, &id, sizeof id);
    return HSD_OK;
} HSD_ABI_GUARD(nullptr)

static int check_shard(hsd_ctx *ctx, int world, int rank) {
    if (world < 1 || rank < 0 || rank >= world) return hsd_fail(ctx, HSD_ERR_INVALID, "bad world/rank %d/%d", world, rank);
    if (ctx->n <= 0) return hsd_fail(ctx, HSD_ERR_NO_GRAPH, "upload the graph before hsd_comm_init");
    return HSD_OK;
}

// Commits a row partition.  Only called once nothing can fail any more: a context never holds a
// world > 1 partition without the communicator that goes with it.
static void commit_shard(hsd_ctx *ctx, int world, int rank, ncclComm_t comm) {
    hsd_comm_destroy_internal(ctx);
    ctx->comm = reinterpret_cast<ncclComm *>(comm);
    ctx->world = world;
    ctx->rank = rank;
    // blocks of ceil(n / world) rows rounded up to whole 128-node tiles: every rank's block starts
    // tile- and 16-byte aligned (the pair kernel's direct-to-LDS path needs i_begin % 4 == 0)
    ctx->rows_per_rank = world > 1 ? round_up((ctx->n + world - 1) / world, HSD_TILE) : ctx->n;
    // a new partition invalidates the resident counts tensor (its row padding depends on it)
    (void)hipStreamSynchronize(ctx->stream);
    hsd_dev_free(ctx->d_counts);
    hsd_release_operands(ctx);
    ctx->hop = -1;
}

extern "C" int hsd_comm_init(hsd_ctx *ctx, int world, int rank, const void *id128) try {
    HSD_CHECK_CTX(ctx);
    int rc = check_shard(ctx, world, rank);
    if (rc) return rc;
    ncclComm_t comm = nullptr;
    // a one-rank communicator is only created on request (HSD_FORCE_COMM=1): it lets a single-GPU
    // box exercise the real ncclCommInitRank / in-place ncclAllGather path.
    // HSD_COMM_DRYRUN: rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks: the row
    // partition is set, no communicator is made (hsd_signature_allgather then fails; the caller exchanges
    // through the host)
    const bool want_comm = !(world == 1 && !(id128 && hsd_hook(H_FORCE_COMM))) && !hsd_hook(H_COMM_DRYRUN);
    if (want_comm) {
        if (!id128) return hsd_fail(ctx, HSD_ERR_INVALID, "null unique id");
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        // the communicator first; the old partition (and its communicator) stay in place if this fails
        ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
        if (r != ncclSuccess) return hsd_fail(ctx, HSD_ERR_COMM, "ncclCommInitRank: %s", ncclGetErrorString(r));
    }
    commit_shard(ctx, world, rank, comm);
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_comm_shard(hsd_ctx *ctx, int64_t *r0, int64_t *r1) try {
    HSD_CHECK_CTX(ctx);
    if (ctx->n <= 0) return hsd_fail(ctx, HSD_ERR_NO_GRAPH, "no graph uploaded");
    const int64_t rpr = ctx->world > 1 ? ctx->rows_per_rank : ctx->n;
    const int64_t a = std::min<int64_t>(ctx->n, (int64_t)ctx->rank * rpr);
    const int64_t b = std::min<int64_t>(ctx->n, a + rpr);
    if (r0) *r0 = a;
    if (r1) *r1 = b;
    return HSD_OK;
} HSD_ABI_GUARD(ctx)

extern "C" int hsd_signature_allgather(hsd_ctx *ctx) try {
    HSD_CHECK_CTX(ctx);
    if (!ctx->d_counts) return hsd_fail(ctx, HSD_ERR_NO_SIGNATURE, "no signature built");
    if (ctx->world == 1 && !ctx->comm) return HSD_OK;
    if (!ctx->comm) return hsd_fail(ctx, HSD_ERR_COMM, "hsd_comm_init has not been called");
    const size_t row = (size_t)(ctx->hop + 1) * ctx->n_bins;
    const size_t cnt = (size_t)ctx->rows_per_rank * row;
    // in place: rank g's block sits at offset g * cnt of the receive buffer
    uint32_t *recv = ctx->d_counts;
    const uint32_t *send = recv + (size_t)ctx->rank * cnt;
    ncclResult_t r = ncclAllGather(send, recv, cnt, ncclUint32, reinterpret_cast<ncclComm_t>(ctx->comm), ctx->stream);
    if (r != ncclSuccess) return hsd_fail(ctx, HSD_ERR_COMM, "ncclAllGather: %s", ncclGetErrorString(r));
    hsd_release_operands(ctx);
    HSD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HSD_OK;
} HSD_ABI_GUARD(ctx)
