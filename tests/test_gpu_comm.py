"""GPU: the RCCL entry points on a single-GPU box -- a one-rank communicator drives the real
ncclCommInitRank and the in-place ncclAllGather of the counts tensor (the 8-GPU run itself is
the driver's; the partition logic is covered by tests/test_distributed_cpu.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_allgather():
    from hsd_amd.engine import Engine
    from hsd_amd.graph import erdos_renyi
    g = erdos_renyi(1200, 0.01, 8)
    os.environ["HSD_FORCE_COMM"] = "1"
    eng = Engine(0)
    try:
        eng.set_graph(g.rowptr, g.col)
        uid = eng.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        eng.comm_init(1, 0, uid)
        assert eng.comm_shard() == (0, g.n_node)
        eng.signature_build_degree(2)
        before = eng.signature_counts()
        eng.signature_allgather()                 # in place, one rank: must leave the tensor intact
        assert np.array_equal(eng.signature_counts(), before)
        eng.pair_prepare(0, 1, 0, 1)
        D = eng.pair_rows_host()
        assert np.array_equal(D, D.T) and D.max() > 0
    finally:
        os.environ.pop("HSD_FORCE_COMM", None)
        eng.close()



WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["HSD_REPO"])
import torch.distributed as dist
import hsd_amd
from hsd_amd import distributed as hdist
from hsd_amd.graph import barabasi_albert

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = barabasi_albert(int(os.environ["HSD_N"]), 4, 17)
eng = hsd_amd.Engine(0)                                   # HSD_COMM_DRYRUN: every rank on GPU 0
eng.set_graph(g.rowptr, g.col)
r0, r1 = hdist.init_comm(eng, dist, rank, world)           # hsd_comm_init(world = 2): the partition bench.py --gpus 2 commits
assert (r0, r1) == hdist.shard_rows(g.n_node, world, rank) == eng.comm_shard()
eng.signature_build_degree(3, r0, r1)                      # this rank's sources only
hdist.allgather_signature(eng, dist, rank, world, via="host")
for name, mode in (("B", (0, 1, 0, 2)), ("A", (0, 0, 0, 3)), ("KL", (1, 0, 0, 3))):
    blk = hdist.distance_row_block(eng, *mode, rank, world, device_out=False)     # rows: hsd_pair_shard (blocks of equal planned cost)
    np.save(os.path.join(os.environ["HSD_OUT"], f"{name}_{rank}.npy"), blk)
    np.save(os.path.join(os.environ["HSD_OUT"], f"{name}_rows_{rank}.npy"), np.array(eng.pair_shard(world, rank)))
    idx, val = eng.pair_topk(9, r0, r1) if name != "KL" else (None, None)
    if idx is not None:
        np.save(os.path.join(os.environ["HSD_OUT"], f"{name}_topk_{rank}.npy"), np.concatenate([idx.astype(np.float64), val], axis=1))
dist.barrier()
eng.close()
dist.destroy_process_group()
'''


def test_two_ranks_run_the_hip_row_block_path(tmp_path, engine):
    """world = 2 through the code path of `bench.py --gpus 2`: hsd_comm_init(world = 2) (HSD_COMM_DRYRUN: no communicator,
    both ranks on this box's one GPU), per-rank sweeps, hsd_amd.distributed.allgather_signature(via="host"), then each rank's
    row block by the HIP kernels -- diagonal square + the rectangles on the side streams, persistent kernel forced -- and its
    k-nearest rows by the pruned sweep of a row block.  The two blocks together are the one-GPU matrix, bit for bit for the Wasserstein variants."""
    import socket
    import subprocess
    import sys
    from conftest import REPO
    from hsd_amd.distributed import shard_rows
    from hsd_amd.graph import barabasi_albert
    n = 2 * 1408 + 37                                        # rows per rank = 1536 (whole tiles); the last block carries a strip
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # (HSD_TOPK_PRUNE=1: the k-nearest rows of a rank's block take the PRUNED sweep -- its rows as positions of their own behind the graph's order,
    #  round 5 -- also for the zero-pad distance at this small size; the verbose log proves it ran)
    env = dict(os.environ, HSD_REPO=REPO, HSD_OUT=str(tmp_path), HSD_N=str(n), HSD_COMM_DRYRUN="1", HSD_STREAM="1", HSD_TOPK_PRUNE="1", HSD_TOPK_VERBOSE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(w)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    log = p.stdout + p.stderr
    assert log.count("pruned sweep (row block)") == 4 and log.count("overflow 0") >= 4 and "the unpruned sweep takes the call" not in log, log[-3000:]
    g = barabasi_albert(n, 4, 17)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    os.environ["HSD_STREAM"] = "0"
    try:
        for name, mode in (("B", (0, 1, 0, 2)), ("A", (0, 0, 0, 3)), ("KL", (1, 0, 0, 3))):
            engine.pair_prepare(*mode)
            D = engine.pair_rows_host()
            spans = [tuple(int(v) for v in np.load(tmp_path / f"{name}_rows_{rank}.npy")) for rank in range(2)]
            assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == n and spans[0][1] % 128 == 0, spans
            for rank in range(2):
                r0, r1 = spans[rank]
                blk = np.load(tmp_path / f"{name}_{rank}.npy")
                assert blk.shape == (r1 - r0, n)
                if name == "KL":
                    # a row block mirrors the upper triangle of ITS rows; KL(i||j) for the columns beyond the block is the same
                    assert np.allclose(blk[:, r1:], D[r0:r1, r1:], rtol=1e-9, atol=1e-12)
                    iu = np.triu_indices(r1 - r0, 1)
                    assert np.allclose(blk[:, r0:r1][iu], D[r0:r1, r0:r1][iu], rtol=1e-9, atol=1e-12)
                else:
                    assert np.array_equal(blk, D[r0:r1]), (name, rank)
                    r0, r1 = shard_rows(n, 2, rank)                    # (the k-nearest rows were asked for the sweep's block)
                    tk = np.load(tmp_path / f"{name}_topk_{rank}.npy")
                    idx, val = tk[:, :9].astype(np.int64), tk[:, 9:]
                    Dm = D[r0:r1].copy()
                    Dm[np.arange(r1 - r0), np.arange(r0, r1)] = np.inf
                    order = np.lexsort((np.broadcast_to(np.arange(n), Dm.shape), Dm), axis=1)[:, :9]
                    assert np.array_equal(idx, order) and np.array_equal(val, np.take_along_axis(Dm, order, 1))
    finally:
        os.environ.pop("HSD_STREAM", None)


@pytest.mark.gpu
def test_pair_shard_tiles_the_matrix_and_favours_hub_rows(engine):
    """hsd_pair_shard: contiguous row blocks of equal PLANNED cost, boundaries on whole 128-row tiles, covering [0, n) in rank order.  On a
    preferential-attachment graph the first rank (the hubs' rows: more rank words per tile pair under the zero-pad distance) gets fewer rows
    than the last; the raw fp64 distance (every tile costs the same) gets equal blocks; HSD_SHARD_BALANCE=0 switches the weighting off."""
    import os
    from hsd_amd.graph import barabasi_albert
    g = barabasi_albert(30000 + 50, 5, 42)
    n = g.n_node
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    engine.pair_prepare(0, 1, 0, 2)
    for world in (1, 2, 3, 8):
        blocks = [engine.pair_shard(world, r) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        for (a, b), (c, d) in zip(blocks, blocks[1:]):
            assert b == c and a <= b and b % 128 == 0
    heights = [b - a for a, b in blocks]
    assert heights[0] < heights[-1] and heights[0] < n / 8 < max(heights), heights
    os.environ["HSD_SHARD_BALANCE"] = "0"
    try:
        flat = [engine.pair_shard(8, r) for r in range(8)]
    finally:
        os.environ.pop("HSD_SHARD_BALANCE", None)
    assert max(b - a for a, b in flat) - min(b - a for a, b in flat) <= 128 + 50
    engine.pair_prepare(0, 0, 0, 3)
    eq = [engine.pair_shard(8, r) for r in range(8)]
    assert max(b - a for a, b in eq) - min(b - a for a, b in eq) <= 128 + 50
    with pytest.raises(ValueError):
        engine.pair_shard(4, 4)
