"""Row-block partition of the pair space over the GPUs of one node.

One process per GPU.  The signature tensor is the only thing exchanged: every rank sweeps its
own block of source nodes, ONE all-gather materialises the (N x (hop+1) x bins) counts on every
GPU, then rank g computes D[rows_g, :] with no further communication.

The data plane is the library's own ncclAllGather (RCCL over xGMI, csrc/comm.hip).  `group`
is any object with the torch.distributed collective API; it is used for the control plane only
(unique-id broadcast) -- or, with via="host", also as a fallback data plane that moves the
counts through host memory (gloo), which is what the CPU tests exercise.
"""
from __future__ import annotations

import numpy as np

from .engine import HsdError


def rows_per_rank(n, world):
    rpr = (n + world - 1) // world
    return (rpr + 127) // 128 * 128 if world > 1 else n


def shard_rows(n, world, rank):
    """Rows owned by `rank`: blocks of ceil(n / world) rows rounded up to whole 128-node tiles
    (same rule as hsd_comm_shard); trailing ranks may own fewer rows, or none for tiny graphs."""
    rpr = rows_per_rank(n, world)
    a = min(n, rank * rpr)
    return a, min(n, a + rpr)


def init_comm(engine, dist, rank, world, host_fallback=False):
    """Create the RCCL communicator: rank 0 makes the 128-byte id, `dist` carries it.

    host_fallback=True: if ncclCommInitRank fails on ANY rank (the ranks agree through `dist`), every rank commits the same row
    partition without a communicator instead of raising, and `engine.exchange` says "host": allgather_signature then moves the counts
    through host memory (gloo).  The distance kernels do not care how the counts arrived.
    What the fallback covers: SYMMETRIC failures -- a node without working RCCL, where ncclCommInitRank returns an error on the ranks that call
    it.  It does not cover one hung or dead rank: ncclCommInitRank is collective, so its peers block in it (RCCL's own timeout ends that), and
    no agreement through `dist` can be reached with a rank that is gone.  Not thread-safe: the dry-run commit below toggles a process-wide
    environment hook for the duration of one call (one context per process is the deployment this serves)."""
    import os
    import torch
    idt = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        idt = torch.frombuffer(bytearray(engine.comm_unique_id()), dtype=torch.uint8).clone()
    dist.broadcast(idt, 0)
    err = None
    try:
        engine.comm_init(world, rank, bytes(idt.numpy().tobytes()))
    except HsdError as e:       # ncclCommInitRank failed on this rank (anything else -- a TypeError, a bad argument -- is a bug and propagates)
        if not host_fallback:
            raise
        err = e
    engine.exchange = "rccl"
    if host_fallback:
        bad = torch.tensor([1 if err is not None else 0], dtype=torch.int32)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad[0]):
            old = os.environ.get("HSD_COMM_DRYRUN")
            os.environ["HSD_COMM_DRYRUN"] = "1"                 # partition without a communicator (csrc/comm.hip)
            try:
                engine.comm_init(world, rank, None)
            finally:
                if old is None:
                    os.environ.pop("HSD_COMM_DRYRUN", None)
                else:
                    os.environ["HSD_COMM_DRYRUN"] = old
            engine.exchange = "host"
            engine.exchange_reason = str(err) if err is not None else "ncclCommInitRank failed on another rank"
    return engine.comm_shard()


def allgather_signature(engine, dist=None, rank=0, world=1, via="rccl"):
    """Make every rank hold the counts of all N nodes."""
    if world == 1:
        return
    if via == "rccl" and getattr(engine, "exchange", "rccl") == "rccl":
        engine.signature_allgather()
        return
    # host path: counts of the own block -> gloo all_gather -> upload the other blocks
    import torch
    n = engine.n
    r0, r1 = shard_rows(n, world, rank)
    rpr = rows_per_rank(n, world)
    hop, bins = engine.signature_dims()
    mine = np.zeros((rpr, hop + 1, bins), dtype=np.uint32)
    mine[:r1 - r0] = engine.signature_counts(r0, r1)
    send = torch.from_numpy(mine.view(np.int32))
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    for g in range(world):
        if g == rank:
            continue
        a, b = shard_rows(n, world, g)
        if b > a:
            engine.signature_set_counts(a, b, recv[g].numpy().view(np.uint32)[:b - a])


def distance_row_block(engine, metric, align, layer_lo, layer_hi, rank, world, kl_alpha=0.5, device_out=True):
    """This rank's rows of the distance matrix (DeviceMatrix or numpy), after the exchange: contiguous row blocks of equal planned
    cost (hsd_pair_shard), which tile [0, n) in rank order."""
    engine.pair_prepare(metric, align, layer_lo, layer_hi, kl_alpha, False)
    r0, r1 = engine.pair_shard(world, rank)
    if device_out:
        return engine.pair_rows_device(r0, r1)
    return engine.pair_rows_host(r0, r1)
