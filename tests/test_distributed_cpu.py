"""CPU (gloo, world_size 2 and 3): the N > 1 path -- row partition, signature exchange through
hsd_amd.distributed, per-rank row blocks -- with an oracle-backed stand-in for the device engine."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from hsd_amd.distributed import shard_rows

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["HSD_REPO"])
import torch.distributed as dist
from hsd_amd.distributed import shard_rows, allgather_signature, init_comm
from hsd_amd.graph import erdos_renyi
from oracle import hsd_oracle as orc
from oracle import c_oracle as corc

class OracleEngine:
    """Stands in for hsd_amd.Engine on a box without a GPU (test infrastructure only)."""
    def __init__(self, g, hop):
        self.n, self.hop = g.n_node, hop
        self.support = np.unique(g.degree)
        self.counts = np.zeros((self.n, hop + 1, self.support.size), dtype=np.uint32)
        self.g = g
    def build(self, r0, r1):
        rings = [orc.khop_rings_node(self.g.rowptr, self.g.col, i, self.hop) for i in range(r0, r1)]
        _, c = orc.degree_histograms(self.g.degree, rings + [], self.hop + 1) if rings else (None, None)
        if rings:
            self.counts[r0:r1] = c
    def signature_dims(self): return self.hop, self.support.size
    def signature_counts(self, r0, r1): return self.counts[r0:r1].copy()
    def signature_set_counts(self, r0, r1, c): self.counts[r0:r1] = c
    def rows(self, r0, r1): return corc.pair_rows_kl(self.counts, 0, self.hop, 0.5, True, r0, r1)
    # the communicator calls of hsd_amd.Engine: ncclCommInitRank "fails" on the last rank only (HSD_TEST_COMM_FAIL=1); without a
    # communicator (the HSD_COMM_DRYRUN form init_comm falls back to) the partition is committed all the same
    def comm_unique_id(self): return bytes(128)
    def comm_init(self, world, rank, uid):
        self.comm_calls = getattr(self, "comm_calls", 0) + 1
        if os.environ.get("HSD_TEST_COMM_FAIL") and not os.environ.get("HSD_COMM_DRYRUN") and rank == world - 1:
            from hsd_amd.engine import HsdError      # what Engine raises for a failed ncclCommInitRank (status HSD_ERR_COMM); a TypeError would propagate
            raise HsdError("RCCL error: ncclCommInitRank: unhandled system error (test)")
        self.world, self.rank, self.has_comm = world, rank, not os.environ.get("HSD_COMM_DRYRUN")
    def comm_shard(self): return shard_rows(self.n, self.world, self.rank)
    def signature_allgather(self): raise AssertionError("the RCCL all-gather must not be reached after a failed communicator")

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = erdos_renyi(301, 0.03, seed=9)
eng = OracleEngine(g, 2)
if os.environ.get("HSD_TEST_COMM_FAIL"):
    # bench.py's path: init_comm with the host fallback -- every rank must agree on it although only one rank's init failed
    r0, r1 = init_comm(eng, dist, rank, world, host_fallback=True)
    assert (r0, r1) == shard_rows(g.n_node, world, rank) and eng.exchange == "host" and not eng.has_comm and eng.comm_calls == 2
    assert "ncclCommInitRank" in eng.exchange_reason
    via = "rccl"                                   # what bench.py asks for; the engine's state redirects it
else:
    r0, r1 = shard_rows(g.n_node, world, rank)
    via = "host"
eng.build(r0, r1)
assert eng.counts[:r0].sum() == 0 and eng.counts[r1:].sum() == 0
allgather_signature(eng, dist, rank, world, via=via)
block = eng.rows(r0, r1)
np.save(os.path.join(os.environ["HSD_OUT"], f"block_{rank}.npy"), block)
np.save(os.path.join(os.environ["HSD_OUT"], f"counts_{rank}.npy"), eng.counts)
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_rows_cover_everything():
    for n in (1, 7, 128, 301, 10000):
        for world in (1, 2, 3, 8):
            blocks = [shard_rows(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (a, b), (c, d) in zip(blocks, blocks[1:]):
                assert b == c and a <= b
            ideal = (n + world - 1) // world
            assert max(b - a for a, b in blocks) <= (ideal + 127) // 128 * 128
            assert all(a % 128 == 0 for a, b in blocks if b > a)


@pytest.mark.parametrize("world,comm_fail", [(2, False), (3, False), (2, True)])
def test_row_block_partition_gloo(tmp_path, world, comm_fail):
    from hsd_amd.graph import erdos_renyi
    from oracle import hsd_oracle as orc
    from oracle import c_oracle as corc
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + os.getpid() % 2000 + world
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSD_REPO=REPO, HSD_OUT=str(tmp_path), **({"HSD_TEST_COMM_FAIL": "1"} if comm_fail else {}))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()[-2000:]
    g = erdos_renyi(301, 0.03, seed=9)
    rings = orc.khop_rings(g.rowptr, g.col, 2)
    _, counts = orc.degree_histograms(g.degree, rings, 3)
    full = corc.pair_rows_kl(counts, 0, 2, 0.5, True)
    got = np.concatenate([np.load(tmp_path / f"block_{r}.npy") for r in range(world)])
    assert got.shape == full.shape and np.array_equal(got, full)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"counts_{r}.npy").astype(np.int64), counts)
