"""GPU: alternative evaluation forms of the zero-pad distance give identical results.

* rank form vs bin form of a layer, whole or above a threshold (csrc/pair.hip, SegDesc): the same integer sum, so matrices
  must be BIT-identical with the rank form disabled (HSD_RANK=0) or restricted to whole layers (HSD_RANK=1), for packed uint16 and plain uint32 operands, both tile
  shapes, aligned and unaligned row ranges, and against the oracle;
* the fp64 fallback kernel (PAIR_ZP_F64) that serves non-integral supports -- reachable through
  hsd_signature_build with fractional node values -- against the oracle on the raw samples;
* the plain-C client of include/hsd.h runs the hot path end to end as a separate (non-Python) process.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-6, 1e-12


class _Env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update({k: str(v) for k, v in self.kw.items()})

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _scale_free(n, seed=42):
    from hsd_amd.graph import barabasi_albert
    return barabasi_albert(n, 5, seed)


@pytest.mark.parametrize("n", [1500, 3001])
@pytest.mark.parametrize("pack", [1, 0])
def test_rank_form_bit_identical_to_bin_form(engine, n, pack):
    from oracle import c_oracle as corc
    g = _scale_free(n)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    mats = {}
    for rank in (2, 1, 0):                # 2: thresholded (low bins + rank tail), 1: pure rank form only, 0: bin form only
        for tile in (64, 128):
            with _Env(HSD_RANK=rank, HSD_SAD16=pack, HSD_TILE=tile):
                engine.pair_prepare(0, 1, 0, 2)
                mats[(rank, tile)] = engine.pair_rows_host()
                if rank == 2 and tile == 128:
                    work_thr = engine.pair_work()
                if rank == 1 and tile == 128:
                    work_rank = engine.pair_work()
                    cols, _ = engine.pair_operand_dims()
                    # an unaligned row block (register-staged kernel instead of direct-to-LDS loads)
                    blk = engine.pair_rows_host(n // 2 + 1, n // 2 + 131)
                    assert np.array_equal(blk, mats[(1, 128)][n // 2 + 1:n // 2 + 131])
                if rank == 0 and tile == 128:
                    work_bin = engine.pair_work()
    ref = mats[(0, 128)]
    for key, m in mats.items():
        assert np.array_equal(m, ref), key
    # the rank form was really taken: ring 1 of a scale-free graph has ~10 members against >100 degree bins
    assert work_rank < 0.8 * work_bin, (work_rank, work_bin)
    assert work_thr <= work_rank, (work_thr, work_rank)
    rs = corc.RingSamples(g.rowptr, g.col, 3, g.degree.astype(np.float64))
    want, _ = rs.pair_rows(0, 2, corc.ALIGN_ZEROPAD, 0, 48)
    assert np.allclose(ref[:48], want, rtol=RTOL, atol=ATOL), np.abs(ref[:48] - want).max()
    assert np.array_equal(ref, ref.T) and not ref.diagonal().any()


def test_rank_form_topk_and_hop1(engine):
    """The fused k-nearest consumer runs the same tile kernel (chunk list included); K = 1 leaves only layer 0
    (no histogram layer at all: an empty chunk list)."""
    g = _scale_free(2000)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    engine.pair_prepare(0, 1, 0, 2)
    D = engine.pair_rows_host()
    idx, dist = engine.pair_topk(7)
    for i in (0, 5, 1234, 1999):
        d = D[i].copy()
        d[i] = np.inf
        order = np.lexsort((np.arange(len(d)), d))[:7]
        assert np.array_equal(idx[i], order) and np.array_equal(dist[i], d[order])
    engine.signature_build_degree(1)
    engine.pair_prepare(0, 1, 0, 0)
    D0 = engine.pair_rows_host()
    deg = g.degree.astype(np.float64)
    assert np.array_equal(D0, np.abs(deg[:, None] - deg[None, :]))


def test_zeropad_fp64_fallback_matches_oracle(engine):
    """Non-integral support -> the exact integer kernels do not apply; the fp64 zero-pad kernel takes over."""
    from hsd_amd.graph import erdos_renyi
    from oracle import c_oracle as corc
    g = erdos_renyi(900, 0.012, seed=5)
    value = np.round(np.sqrt(g.degree.astype(np.float64) + 0.25), 3)          # fractional, few distinct values
    support = np.unique(value)
    node_bin = np.searchsorted(support, value).astype(np.int32)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build(3, node_bin, support)
    rs = corc.RingSamples(g.rowptr, g.col, 3, value)
    for align, lo, hi in ((1, 0, 2), (1, 1, 3)):
        engine.pair_prepare(0, align, lo, hi)
        cols, ebytes = engine.pair_operand_dims()
        assert ebytes == 8                                                    # the fp64 kind, not uint32
        D = engine.pair_rows_host()
        want, _ = rs.pair_rows(lo, hi, corc.ALIGN_ZEROPAD, 0, 64)
        assert np.allclose(D[:64], want, rtol=RTOL, atol=ATOL), np.abs(D[:64] - want).max()
        assert np.array_equal(D, D.T)
    # and the raw-sample distance on the same fractional support (no empty rings in the layers compared)
    sizes = engine.khop_sizes(3)
    if not (sizes[:, :3] == 0).any():
        engine.pair_prepare(0, 0, 0, 2)
        D = engine.pair_rows_host()
        want, _ = rs.pair_rows(0, 2, corc.ALIGN_NONE, 0, 64)
        assert np.allclose(D[:64], want, rtol=RTOL, atol=ATOL)


def test_plain_c_client_runs_the_hot_path(engine, c_client):
    """tests/c/abi_client.c gpu: k-hop sizes, signature, prepare, rows and the error contract from a C99 program
    (built at collection time, before this process touched the GPU; started as a fresh child process)."""
    out = subprocess.run([c_client, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "abi ok (gpu)" in out.stdout


@pytest.mark.parametrize("nw", [4])
def test_persistent_kernels_bit_identical(engine, nw):
    """pair_stream_kernel (persistent workgroups streaming through their tiles, 4 waves or 1) against the
    one-tile-per-workgroup kernel: same per-pair operation order, so every distance kind must agree bit for bit --
    full matrix (diagonal square, mirrored) and a row block with rectangles left and right of it."""
    g = _scale_free(3001, seed=7)
    n = g.n_node
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    cases = [(0, 1, 0, 2), (0, 0, 0, 3), (1, 0, 0, 3), (2, 0, 0, 3), (0, 1, 1, 3)]     # (metric, align, lo, hi)
    for metric, align, lo, hi in cases:
        with _Env(HSD_STREAM=0):
            engine.pair_prepare(metric, align, lo, hi)
            ref = engine.pair_rows_host()
            blk_ref = engine.pair_rows_host(1024, 2048)
        with _Env(HSD_STREAM=nw):
            engine.pair_prepare(metric, align, lo, hi)
            got = engine.pair_rows_host()
            blk = engine.pair_rows_host(1024, 2048)
            work = engine.pair_work()
        assert np.array_equal(got, ref), (metric, align, lo, hi, np.abs(got - ref).max())
        assert np.array_equal(blk, blk_ref)
        # (a row block evaluates the pairs below the diagonal itself instead of mirroring them: symmetric KL then sums
        # its two halves in the other order)
        assert np.array_equal(blk, ref[1024:2048]) if metric != 2 else np.allclose(blk, ref[1024:2048], rtol=1e-9, atol=1e-15)
        assert work > 0


def test_uint8_ranges_bit_identical(engine, capfd):
    """Packed layers keep a third range with their small values as bytes (v_sad_u8: four columns per instruction); tiles
    whose granules hold a node with a value >= 256 there must not read it, and must still see their own largest ring.  A
    scale-free graph large enough for hubs (ring 2 in the thousands beside rings of a few dozen): the persistent kernel with
    the ranges (the plan must have built one), without them (HSD_U8=0) and the one-tile kernel agree bit for bit -- matrix,
    row block off the tile grid, natural and sorted granule order."""
    g = _scale_free(16000 + 77, seed=11)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    with _Env(HSD_STREAM=0, HSD_U8=0):
        engine.pair_prepare(0, 1, 0, 2)
        ref = engine.pair_rows_host(0, 3000)
        ref2 = engine.pair_rows_host(9000 + 4, 9000 + 1300)
    for sort in ("1", "0"):
        words = {}
        for u8 in ("1", "0"):
            with _Env(HSD_STREAM=1, HSD_U8=u8, HSD_GRAN_SORT=sort, HSD_PLAN_VERBOSE=1):
                capfd.readouterr()
                engine.pair_prepare(0, 1, 0, 2)
                plan = capfd.readouterr().err
                got = engine.pair_rows_host(0, 3000)
                words[u8] = engine.pair_work()
                got2 = engine.pair_rows_host(9000 + 4, 9000 + 1300)
            assert ("uint8 range" in plan) == (u8 == "1"), plan
            assert np.array_equal(got, ref), (sort, u8, np.abs(got - ref).max())
            assert np.array_equal(got2, ref2), (sort, u8)
        assert words["1"] < words["0"], words            # (rows 0..3000 of a preferential-attachment graph are its hubs: few tiles qualify)


def test_plan_does_not_depend_on_host_threads(engine, capfd):
    """hsd_pair_prepare prices its candidate granule orders on host threads.  Capped to the calling thread (HSD_PLAN_THREADS=1) and with the
    creation of the third thread failing (HSD_PLAN_THREADS_FAIL=3: what RLIMIT_NPROC or a pids limit does) the call still returns a status,
    not an exception through the C ABI, and chooses the same plan: same verbose plan line, same operand words, same matrix rows."""
    g = _scale_free(9000 + 40, seed=5)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    seen = {}
    for tag, env in (("default", {}), ("one", {"HSD_PLAN_THREADS": 1}), ("fail", {"HSD_PLAN_THREADS_FAIL": 3}), ("fail_first", {"HSD_PLAN_THREADS_FAIL": 1})):
        with _Env(HSD_STREAM=1, HSD_PLAN_VERBOSE=1, **env):
            capfd.readouterr()
            engine.pair_prepare(0, 1, 0, 2)
            plan = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("[hsd plan]") and "granule order" in ln]
            rows = engine.pair_rows_host(128, 128 + 700)
            seen[tag] = (plan, engine.pair_work(), rows)
    for tag in ("one", "fail", "fail_first"):
        assert seen[tag][0] == seen["default"][0] and seen[tag][0], (tag, seen[tag][0], seen["default"][0])
        assert seen[tag][1] == seen["default"][1]
        assert np.array_equal(seen[tag][2], seen["default"][2])


@pytest.mark.parametrize("n", [777, 5003])
def test_operand_preparation_staged_rows_equal_in_place_rows(engine, n):
    """The operand-preparation kernels walk every node's histogram row; by default workgroups stage the rows of 64 (32, 16) consecutive nodes in LDS
    with coalesced loads first (csrc/pair_prepare.hip: counts_row), HSD_PREP_STAGED=0 reads each row in place.  Same operands: every distance kind's
    matrix rows are bit-identical, for a node count that is not a multiple of the stage (the last workgroup holds fewer nodes)."""
    g = _scale_free(n, seed=11)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    r0, r1 = 0, min(n, 640)
    kinds = [(0, 1, 0, 2, 0.5, False), (0, 0, 0, 3, 0.5, False), (0, 1, 1, 3, 0.5, False), (1, 0, 0, 3, 0.5, False), (2, 0, 0, 3, 0.25, False), (1, 0, 0, 3, 0.5, True)]
    for kind in kinds:
        rows = {}
        for staged in ("1", "0"):
            with _Env(HSD_PREP_STAGED=staged):
                engine.pair_prepare(*kind)
            rows[staged] = (engine.pair_rows_host(r0, r1), engine.pair_operand_dims())
        assert rows["1"][1] == rows["0"][1], kind
        assert np.array_equal(rows["1"][0], rows["0"][0]), (kind, np.abs(rows["1"][0] - rows["0"][0]).max())


def test_persistent_kernel_corner_shapes(engine):
    """Forced persistent kernel on shapes its size heuristic would not pick: a distance with NO histogram layer (K = 1,
    layers 0..0: tiles without operand chunks, only the ring-0 term), and a hierarchy too deep for its LDS budget
    (K = 4 zero-pad: three layers + ring 0 = four per-node vectors -> every tile falls back to the one-tile kernel)."""
    g = _scale_free(2600, seed=3)
    engine.set_graph(g.rowptr, g.col)
    deg = g.degree.astype(np.float64)
    engine.signature_build_degree(1)
    with _Env(HSD_STREAM=1):
        engine.pair_prepare(0, 1, 0, 0)
        D0 = engine.pair_rows_host()
    assert np.array_equal(D0, np.abs(deg[:, None] - deg[None, :]))
    engine.signature_build_degree(4)
    out = {}
    for stream in (0, 1):
        with _Env(HSD_STREAM=stream):
            engine.pair_prepare(0, 1, 0, 3)
            out[stream] = engine.pair_rows_host()
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[0].T)


@pytest.mark.parametrize("r0,r1", [(476, 2352), (4, 2600), (1100, 1996)])
def test_persistent_kernel_row_block_off_the_granule_grid(engine, r0, r1):
    """A row block that starts at a multiple of 4 but not of 64: a 128-node tile then touches THREE 64-node granules of
    the rank-form maxima per side.  Persistent kernel (HSD_STREAM=1), zero-pad distance with rank columns (a
    preferential-attachment graph), against the rows of the full matrix, bit for bit."""
    from hsd_amd.graph import barabasi_albert
    g = barabasi_albert(2675, 3, 363629067)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    engine.pair_prepare(0, 1, 0, 2)
    D = engine.pair_rows_host()
    os.environ["HSD_STREAM"] = "1"
    try:
        M = engine.pair_rows_device(r0, r1)
        block = M.to_host()
        M.free()
        idx, val = engine.pair_topk(9, r0, r1)
    finally:
        os.environ.pop("HSD_STREAM", None)
    assert np.array_equal(block, D[r0:r1])
    for r in (r0, (r0 + r1) // 2, r1 - 1):
        d = D[r].copy()
        d[r] = np.inf
        order = np.lexsort((np.arange(g.n_node), d))[:9]
        assert np.array_equal(order, idx[r - r0]) and np.array_equal(d[order], val[r - r0])


@pytest.mark.parametrize("variant", ["B", "A"])
def test_row_block_as_one_launch(engine, variant):
    """A multi-GPU rank's row block [r0, r1) x [0, n) of a Wasserstein matrix runs as ONE persistent launch (diagonal square above its diagonal
    with per-tile mirror flags + both rectangles, one column-side granule list of three ranges) instead of three launches side by side
    (HSD_ROWBLOCK=0).  Blocks in the middle, at the top and at the bottom of the matrix, with a partial last row tile and a graph whose size
    is not a multiple of 128 (column strip), sorted and natural granule order: both forms equal the rows of the full matrix bit for bit, and
    every entry of the block is written (the block is pre-filled with NaN)."""
    g = _scale_free(6000 + 55, seed=9)
    n = g.n_node
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    mode = (0, 1, 0, 2) if variant == "B" else (0, 0, 0, 3)
    with _Env(HSD_STREAM=0):
        engine.pair_prepare(*mode)
        D = engine.pair_rows_host()
    assert np.array_equal(D, D.T)
    blocks = [(1280, 2816), (0, 1536), (4608, n), (2560, 2560 + 128 * 5 + 77)]
    for sort in ("1", "0"):
        for form in ("1", "0"):
            with _Env(HSD_STREAM=1, HSD_GRAN_SORT=sort, HSD_ROWBLOCK=form):
                engine.pair_prepare(*mode)
                for r0, r1 in blocks:
                    M = engine.upload_matrix(np.full((r1 - r0, (n + 15) // 16 * 16), np.nan))
                    M.row0 = r0
                    M.cols = n
                    try:
                        engine.pair_rows_device(r0, r1, M)
                        got = M.to_host()
                    finally:
                        M.free()
                    assert np.array_equal(got[:, :n], D[r0:r1]), (variant, sort, form, r0, r1, np.nanmax(np.abs(got[:, :n] - D[r0:r1])), int(np.isnan(got[:, :n]).sum()))
                # the same blocks into host memory (through the staging block: another base pointer, leading dimension and row origin)
                r0, r1 = blocks[0]
                assert np.array_equal(engine.pair_rows_host(r0, r1), D[r0:r1]), (variant, sort, form, "host")
