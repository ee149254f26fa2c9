"""GPU: the fused k-nearest consumer (hsd_pair_topk) against a lexicographic (distance, index)
selection on the full matrix produced by the normal kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reference_topk(D, k, r0=0):
    rows, n = D.shape
    idx = np.empty((rows, k), dtype=np.int32)
    val = np.empty((rows, k))
    cols = np.arange(n)
    for r in range(rows):
        d = D[r].copy()
        d[r0 + r] = np.inf                       # self excluded
        order = np.lexsort((cols, d))[:k]
        idx[r], val[r] = order, d[order]
    return idx, val


@pytest.mark.parametrize("n,p,hop,mode,k", [
    (1500, 0.008, 2, (0, 1, 0, 1), 10),          # uint32 zero-pad W1: many exact ties
    (1500, 0.008, 2, (0, 0, 0, 2), 16),          # fp64 raw W1 (needs no empty ring)
    (2100, 0.006, 3, (2, 0, 0, 3), 5),           # symmetric KL
    (700, 0.02, 2, (0, 1, 0, 1), 64),
])
def test_topk_matches_full_matrix(engine, n, p, hop, mode, k):
    from hsd_amd.graph import erdos_renyi, connect_components
    g = connect_components(erdos_renyi(n, p, 21), 21)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(hop)
    engine.pair_prepare(*mode)
    D = engine.pair_rows_host()
    idx, val = engine.pair_topk(k)
    ridx, rval = _reference_topk(D, k)
    assert np.array_equal(val, rval)
    assert np.array_equal(idx, ridx)
    # a row block on its own (multi-GPU entry): rows see all columns
    a, b = n // 2 - 37, n // 2 + 91
    idx2, val2 = engine.pair_topk(k, a, b)
    assert np.array_equal(idx2, ridx[a:b])
    if mode[0] == 2:      # symmetric KL: off-diagonal blocks accumulate the two dot products in the other order
        assert np.allclose(val2, rval[a:b], rtol=1e-10, atol=0)      # H_i + H_j - dot cancels ~2 digits
    else:
        assert np.array_equal(val2, rval[a:b])


def test_topk_all_ties_and_errors(engine):
    """A cycle: every node is automorphic, all distances are 0 -> the candidate lists overflow and
    the retry path (capacity n) must still give the k smallest indices."""
    from hsd_amd.graph import Graph
    n = 300
    src = np.arange(n)
    g = Graph.from_edges(list(range(n)), src, (src + 1) % n)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(2)
    engine.pair_prepare(0, 1, 0, 1)
    idx, val = engine.pair_topk(7)
    assert np.all(val == 0.0)
    for i in range(n):
        assert list(idx[i]) == [j for j in range(n) if j != i][:7]
    with pytest.raises(ValueError):
        engine.pair_topk(0)
    with pytest.raises(ValueError):
        engine.pair_topk(n)
    engine.pair_prepare(1, 0, 0, 2, 0.5, False)
    with pytest.raises(NotImplementedError):
        engine.pair_topk(3)                      # plain KL is asymmetric: not offered


@pytest.mark.parametrize("mode,k", [((0, 1, 0, 1), 10), ((0, 0, 0, 2), 16), ((2, 0, 0, 2), 5)])
@pytest.mark.parametrize("phases", ["1", "3", "8", "geom"])
def test_topk_phased_sweep_on_the_persistent_kernel(engine, mode, k, phases):
    """The large-N schedule at a size the full matrix can check: persistent kernel (HSD_STREAM=1) with its
    candidate epilogue, the sweep cut into phases with the thresholds tightened in between ("geom": the geometric
    schedule a sweep of BA N = 100k takes -- 1/32, 1/32, 1/16, 1/8, 1/4, 1/2 of the tile groups)."""
    import os
    from hsd_amd.graph import erdos_renyi, connect_components
    n = 3301
    g = connect_components(erdos_renyi(n, 0.004, 5), 5)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(2)
    engine.pair_prepare(*mode)
    D = engine.pair_rows_host()
    ridx, rval = _reference_topk(D, k)
    os.environ["HSD_STREAM"] = "1"
    os.environ.update({"HSD_TOPK_GEOM": "1"} if phases == "geom" else {"HSD_TOPK_PHASES": phases})
    try:
        idx, val = engine.pair_topk(k)
        a, b = 1024, 2560                                  # an aligned row block: square + both rectangles, all phased
        idx2, val2 = engine.pair_topk(k, a, b)
    finally:
        os.environ.pop("HSD_STREAM", None)
        os.environ.pop("HSD_TOPK_PHASES", None)
        os.environ.pop("HSD_TOPK_GEOM", None)
    assert np.array_equal(val, rval)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(idx2, ridx[a:b])
    if mode[0] == 2:
        assert np.allclose(val2, rval[a:b], rtol=1e-10, atol=0)
    else:
        assert np.array_equal(val2, rval[a:b])


def test_topk_target_size_without_the_matrix(engine, capfd):
    """BA N = 100 000 (the target graph), zero-pad W1, 20 nearest: the default call -- from ~50k nodes on the PRUNED sweep on node-permuted operands --
    against rows of the full matrix, and against the unpruned phased sweep (HSD_TOPK_PRUNE=0), list for list."""
    import os
    from hsd_amd.graph import barabasi_albert
    n, k = 100000, 20
    g = barabasi_albert(n, 5, 42)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    engine.pair_prepare(0, 1, 0, 2)
    os.environ["HSD_TOPK_VERBOSE"] = "1"
    try:
        capfd.readouterr()
        idx, val = engine.pair_topk(k)
        log = capfd.readouterr().err
    finally:
        os.environ.pop("HSD_TOPK_VERBOSE", None)
    assert "pruned launch" in log and "overflow 0" in log, log[-400:]
    os.environ["HSD_TOPK_PRUNE"] = "0"
    try:
        uidx, uval = engine.pair_topk(k)
    finally:
        os.environ.pop("HSD_TOPK_PRUNE", None)
    assert np.array_equal(uidx, idx) and np.array_equal(uval, val)
    M = engine.pair_rows_device()
    try:
        for r in (0, 1, 127, 128, 4097, n // 2, n - 1):
            d = M.rows_to_host(r, r + 1)[0].copy()
            d[r] = np.inf
            order = np.lexsort((np.arange(n), d))[:k]
            assert np.array_equal(order, idx[r]) and np.array_equal(d[order], val[r])
    finally:
        M.free()


def test_topk_phased_sweep_overflow_retry(engine):
    """Lists far too short for the first phase: the overflow is reported by the selection and the call retries with
    larger capacities (the last attempt holds n entries per row)."""
    import os
    from hsd_amd.graph import erdos_renyi, connect_components
    n, k = 2200, 8
    g = connect_components(erdos_renyi(n, 0.005, 9), 9)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(2)
    engine.pair_prepare(0, 1, 0, 1)
    D = engine.pair_rows_host()
    ridx, rval = _reference_topk(D, k)
    os.environ.update(HSD_STREAM="1", HSD_TOPK_PHASES="4", HSD_TOPK_CAP=str(k + 1))
    try:
        idx, val = engine.pair_topk(k)
    finally:
        for v in ("HSD_STREAM", "HSD_TOPK_PHASES", "HSD_TOPK_CAP"):
            os.environ.pop(v, None)
    assert np.array_equal(val, rval) and np.array_equal(idx, ridx)


_ORACLE_D = {}


@pytest.mark.parametrize("mode,align,lo,hi,stream", [
    ((0, 1, 0, 2), 1, 0, 2, "0"), ((0, 1, 0, 2), 1, 0, 2, "1"),      # Variant B layers, one-tile / persistent kernel
    ((0, 0, 0, 3), 0, 0, 3, "0"), ((0, 0, 0, 3), 0, 0, 3, "1"),      # Variant A layers
])
def test_topk_against_the_oracle_directly(engine, mode, align, lo, hi, stream):
    """hsd_pair_topk against distances computed by the CPU ORACLE (oracle/hsd_oracle.c: scipy's W1 on the ring samples,
    model/HSD.py:98-114 / :140-161 + tools/metrics.py:18-36) -- not against the HIP matrix: the k nearest of every row,
    values within rtol 1e-6, and the same index sets wherever the oracle's k-th and (k+1)-th distances are not a near-tie."""
    import os
    from hsd_amd.graph import barabasi_albert
    from oracle import c_oracle as corc
    n, k, K = 1536 + 40, 12, 3                    # whole tiles + a strip
    g = barabasi_albert(n, 4, 5)
    if (align, lo, hi) not in _ORACLE_D:           # (seconds of CPU: once per distance, shared by the two kernels)
        rs = corc.RingSamples(g.rowptr, g.col, K, g.degree.astype(np.float64))
        _ORACLE_D[(align, lo, hi)] = rs.pair_rows(lo, hi, corc.ALIGN_ZEROPAD if align else corc.ALIGN_NONE, 0, n)[0]
    R = _ORACLE_D[(align, lo, hi)]
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(K)
    os.environ["HSD_STREAM"] = stream
    try:
        engine.pair_prepare(*mode)
        idx, val = engine.pair_topk(k)
    finally:
        os.environ.pop("HSD_STREAM", None)
    R = R.copy()
    np.fill_diagonal(R, np.inf)
    order = np.argsort(R, axis=1, kind="stable")
    kth, nxt = np.take_along_axis(R, order[:, k - 1:k], 1)[:, 0], np.take_along_axis(R, order[:, k:k + 1], 1)[:, 0]
    ref_val = np.take_along_axis(R, order[:, :k], 1)
    assert np.allclose(val, ref_val, rtol=1e-6, atol=1e-12)
    # the values the engine reports are the oracle's distances of the nodes it names
    assert np.allclose(val, np.take_along_axis(R, idx.astype(np.int64), 1), rtol=1e-6, atol=1e-12)
    clear = (nxt - kth) > 1e-9 * np.maximum(1.0, kth)
    assert clear.mean() > 0.5                     # (the comparison below must not be vacuous)
    same = np.array([set(idx[r]) == set(order[r, :k]) for r in np.nonzero(clear)[0]])
    assert same.all(), int((~same).sum())


def _env(**kw):
    import os

    class _E:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in kw}
            os.environ.update({k: str(v) for k, v in kw.items()})

        def __exit__(self, *exc):
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return _E()


@pytest.mark.parametrize("n,k", [(20000 + 77, 20), (17000, 1), (16512, 128)])
def test_pruned_sweep_equals_the_unpruned_one(engine, capfd, n, k):
    """Raw fp64 W1 over the whole graph takes the PRUNED sweep by default once the persistent kernel takes the region (node-permuted operands,
    k-d order, band thresholds, per-launch tile filter): its (distance, index) lists are those of the unpruned sweep bit for bit -- ties by the
    ORIGINAL node id -- and sampled rows equal a lexicographic selection on rows of the matrix.  The verbose log must show that tile pairs were
    in fact dropped."""
    from hsd_amd.graph import barabasi_albert
    g = barabasi_albert(n, 5, 7)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    engine.pair_prepare(0, 0, 0, 3)
    with _env(HSD_TOPK_PRUNE=0):
        ridx, rval = engine.pair_topk(k)
    with _env(HSD_TOPK_VERBOSE=1):
        capfd.readouterr()
        idx, val = engine.pair_topk(k)
        log = capfd.readouterr().err
    kept = [tuple(int(x) for x in ln.split(":")[1].split("tile")[0].split("of")) for ln in log.splitlines() if "pruned launch" in ln]
    assert kept and sum(a for a, b in kept) < (0.9 if k > 64 else 0.7) * sum(b for a, b in kept), log[-600:]
    assert "overflow 0" in log
    assert np.array_equal(val, rval) and np.array_equal(idx, ridx)
    M = engine.pair_rows_device()
    try:
        for row in (0, 129, n // 2, n - 1):
            d = M.rows_to_host(row, row + 1)[0].copy()
            d[row] = np.inf
            order = np.lexsort((np.arange(n), d))[:k]
            assert np.array_equal(order, idx[row]) and np.array_equal(d[order], val[row])
    finally:
        M.free()


@pytest.mark.parametrize("n,k", [(100, 9), (128, 5), (129, 40), (700, 16), (2222, 20)])
def test_pruned_sweep_small_and_ragged(engine, n, k):
    """The pruned sweep forced onto small graphs (HSD_STREAM=1, HSD_TOPK_PRUNE=1): fewer than one tile, exactly one, one node more, sizes that
    are no multiple of 128 (padding positions share the last tile and must never become a neighbour), a star's hub among them."""
    from hsd_amd.graph import Graph, barabasi_albert
    if n == 700:
        # two stars joined at their hubs: hundreds of leaves with identical rings (exact ties, broken by the ORIGINAL node id) and two hubs
        half = n // 2
        src = np.concatenate([[0], np.zeros(half - 1, dtype=np.int64), np.full(n - half - 1, half, dtype=np.int64)])
        dst = np.concatenate([[half], np.arange(1, half), np.arange(half + 1, n)])
        g = Graph.from_edges(list(range(n)), src, dst)
    else:
        g = barabasi_albert(n, 3, 5)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(2)
    # the raw fp64 distance (pruned by default where the persistent kernel runs) and the zero-pad distance (pruned on request), packed and plain words
    for mode, extra in (((0, 0, 0, 2), {}), ((0, 1, 0, 1), {}), ((0, 1, 0, 1), {"HSD_SAD16": 0})):
        with _env(**extra):
            engine.pair_prepare(*mode)
            D = engine.pair_rows_host()
        ridx, rval = _reference_topk(D, k)
        with _env(HSD_STREAM=1, HSD_TOPK_PRUNE=1, **extra):
            engine.pair_prepare(*mode)
            idx, val = engine.pair_topk(k)
        assert np.array_equal(val, rval) and np.array_equal(idx, ridx), (mode, extra)


@pytest.mark.parametrize("mode", [(0, 0, 0, 3), (0, 1, 0, 2)], ids=["raw-fp64", "zero-pad"])
def test_pruned_sweep_node_order_from_the_device_and_from_the_host(engine, capfd, mode):
    """The node order of the pruned sweep is built on the device (csrc/tkorder.hip: per level of the k-d tree extents, keys, one radix sort);
    HSD_TOPK_ORDER=0 keeps the host threads' order.  Any order gives the same lists; the device's must prune about as well as the host's
    (its keys carry 44 bits of a value, ties stay in the order they had)."""
    from hsd_amd.graph import barabasi_albert
    n, k = 24000 + 77, 20
    g = barabasi_albert(n, 5, 31)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    got = {}
    for order in (0, 1):
        with _env(HSD_TOPK_PRUNE=1, HSD_TOPK_ORDER=order, HSD_TOPK_VERBOSE=1):
            engine.pair_prepare(*mode)                        # (the permuted operands go with the operands)
            capfd.readouterr()
            idx, val = engine.pair_topk(k)
            log = capfd.readouterr().err
        kept = [tuple(int(x) for x in ln.split(":")[1].split("tile")[0].split("of")) for ln in log.splitlines() if "pruned launch" in ln]
        assert kept and "overflow 0" in log, log[-500:]
        assert ("k-d order (device)" in log) == (order == 1) and ("k-d order (host)" in log) == (order == 0), log[:800]
        got[order] = (idx, val, sum(a for a, b in kept) / sum(b for a, b in kept))
    with _env(HSD_TOPK_PRUNE=0):
        ridx, rval = engine.pair_topk(k)
    for order in (0, 1):
        assert np.array_equal(got[order][0], ridx) and np.array_equal(got[order][1], rval), order
    assert got[1][2] < 1.15 * got[0][2] + 0.02, (got[0][2], got[1][2])


def test_pruned_sweep_falls_back_when_a_list_overflows(engine, capfd):
    """A node order that does not fit the distance (eight column groups per layer as features: HSD_TOPK_GROUPS=8) leaves loose first thresholds and
    overflowing lists on a scale-free graph: the call must notice, hand over to the unpruned sweep and still return the exact result."""
    from hsd_amd.graph import barabasi_albert
    n, k = 24000, 20
    g = barabasi_albert(n, 5, 11)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    with _env(HSD_TOPK_PRUNE=0):
        engine.pair_prepare(0, 0, 0, 3)
        ridx, rval = engine.pair_topk(k)
    with _env(HSD_TOPK_GROUPS=8, HSD_TOPK_VERBOSE=1):
        engine.pair_prepare(0, 0, 0, 3)                       # (the permuted operands are rebuilt with the operands)
        capfd.readouterr()
        idx, val = engine.pair_topk(k)
        log = capfd.readouterr().err
    assert np.array_equal(val, rval) and np.array_equal(idx, ridx)
    if "overflow 1" in log:
        assert "attempt 0" in log                            # the unpruned sweep ran after it


def test_pruned_sweep_of_the_zero_pad_distance_on_request(engine, capfd):
    """HSD_TOPK_PRUNE=1 extends the pruned sweep to the zero-pad distance (Variant B): a permuted copy of the packed words, the layers' 1 / ring
    sizes, the suffix sums and -- recovered from those -- the per-granule maxima the tile planner reads; bound = gap of the rings' value sums over
    the larger ring; first thresholds from the column sample AND the neighbouring tiles.  Same lists as the unpruned sweep, tile pairs dropped."""
    from hsd_amd.graph import barabasi_albert
    n, k = 24000 + 19, 20
    g = barabasi_albert(n, 5, 23)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    engine.pair_prepare(0, 1, 0, 2)
    ridx, rval = engine.pair_topk(k)                         # default: unpruned for this kind
    with _env(HSD_TOPK_PRUNE=1, HSD_TOPK_VERBOSE=1):
        capfd.readouterr()
        idx, val = engine.pair_topk(k)
        log = capfd.readouterr().err
    kept = [tuple(int(x) for x in ln.split(":")[1].split("tile")[0].split("of")) for ln in log.splitlines() if "pruned launch" in ln]
    assert kept and sum(a for a, b in kept) < 0.9 * sum(b for a, b in kept) and "overflow 0" in log, log[-500:]
    assert np.array_equal(val, rval) and np.array_equal(idx, ridx)


@pytest.mark.parametrize("mode,n,k,bounds", [
    ((0, 0, 0, 3), 30000 + 77, 20, [(0, 3712), (3712, 7500), (11111, 20001), (26000, 30077)]),        # raw fp64: ragged bounds, first / middle / last block
    ((0, 1, 0, 2), 30000 + 77, 20, [(0, 3840), (7680, 11500), (25001, 30077)]),                       # zero-pad (packed uint16 words)
    ((0, 0, 0, 3), 20000, 1, [(2560, 5120)]),
    ((0, 1, 0, 2), 20000, 128, [(17408, 20000)]),
], ids=["raw-fp64", "zero-pad", "raw-k1", "zero-pad-k128"])
def test_pruned_sweep_of_a_row_block(engine, capfd, mode, n, k, bounds):
    """hsd_pair_topk(k, r0, r1) with r1 - r0 < n -- a rank's share of `--gpus P` -- takes the pruned sweep too: the block's rows are a second run of
    positions behind the graph's k-d order (in that order), swept against ALL columns with row lists only.  Same lists as the unpruned call for
    the block and as the rows of the whole-graph call, bit for bit, whatever the bounds; tile pairs are in fact dropped; a row never lists itself
    although its own node is among the columns."""
    from hsd_amd.graph import barabasi_albert
    g = barabasi_albert(n, 5, 13)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    engine.pair_prepare(*mode)
    with _env(HSD_TOPK_PRUNE=0):
        widx, wval = engine.pair_topk(k)                     # the whole graph, unpruned: the reference
    for r0, r1 in bounds:
        with _env(HSD_TOPK_PRUNE=0):
            uidx, uval = engine.pair_topk(k, r0, r1)
        # (HSD_STREAM=1: a block of a few thousand rows of a 30k-node graph is below the size from which the persistent kernel -- the only one that
        #  reads permuted operands -- is chosen by itself; a rank's block of config 4 is 25 000 x 200 000)
        with _env(HSD_TOPK_PRUNE=1, HSD_TOPK_VERBOSE=1, HSD_STREAM=1):
            capfd.readouterr()
            idx, val = engine.pair_topk(k, r0, r1)
            log = capfd.readouterr().err
        assert "pruned sweep (row block)" in log and "overflow 0" in log and "the unpruned sweep takes the call" not in log, log[-800:]
        kept = [tuple(int(x) for x in ln.split(":")[1].split("tile")[0].split("of")) for ln in log.splitlines() if "pruned launch" in ln]
        print(f"rows [{r0}, {r1}): kept {sum(a for a, b in kept)} of {sum(b for a, b in kept)} tile pairs")
        assert kept and sum(a for a, b in kept) < 0.97 * sum(b for a, b in kept), log[-600:]
        assert idx.shape == (r1 - r0, k) and np.array_equal(val, uval) and np.array_equal(idx, uidx), (r0, r1)
        assert np.array_equal(val, wval[r0:r1]) and np.array_equal(idx, widx[r0:r1])
        assert not (idx == np.arange(r0, r1)[:, None]).any()
        with _env(HSD_TOPK_PRUNE=1, HSD_STREAM=1):
            idx2, val2 = engine.pair_topk(k, r0, r1)         # the block's setup is kept: the second call is the same call
        assert np.array_equal(idx2, idx) and np.array_equal(val2, val)
    with _env(HSD_TOPK_PRUNE=1):
        idx, val = engine.pair_topk(k)                       # and back to the whole graph (another setup)
    assert np.array_equal(idx, widx) and np.array_equal(val, wval)


@pytest.mark.parametrize("n,k", [(300, 7), (1000, 16), (2222, 20)])
def test_pruned_sweep_of_a_row_block_small_and_ragged(engine, n, k):
    """Row blocks of the pruned sweep forced onto small graphs (HSD_STREAM=1, HSD_TOPK_PRUNE=1): blocks smaller than a tile, blocks that end inside
    one, all-ties stars, both kinds and plain words -- against a lexicographic selection on the matrix rows."""
    from hsd_amd.graph import Graph, barabasi_albert
    if n == 1000:
        half = n // 2
        src = np.concatenate([[0], np.zeros(half - 1, dtype=np.int64), np.full(n - half - 1, half, dtype=np.int64)])
        dst = np.concatenate([[half], np.arange(1, half), np.arange(half + 1, n)])
        g = Graph.from_edges(list(range(n)), src, dst)
    else:
        g = barabasi_albert(n, 3, 5)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(2)
    for mode, extra in (((0, 0, 0, 2), {}), ((0, 1, 0, 1), {}), ((0, 1, 0, 1), {"HSD_SAD16": 0})):
        with _env(**extra):
            engine.pair_prepare(*mode)
            D = engine.pair_rows_host()
        with _env(HSD_STREAM=1, HSD_TOPK_PRUNE=1, **extra):
            engine.pair_prepare(*mode)
            for r0, r1 in ((0, 50), (128, 256), (n // 3, n // 3 + 200), (n - 37, n), (1, n)):
                r1 = min(r1, n)
                ridx, rval = _reference_topk(D[r0:r1], k, r0)
                idx, val = engine.pair_topk(k, r0, r1)
                assert np.array_equal(val, rval) and np.array_equal(idx, ridx), (mode, extra, r0, r1)


def test_pruned_sweep_hands_over_when_its_setup_cannot_allocate(engine, capfd):
    """The pruned sweep needs a permuted copy of the operands, a sample matrix and rocprim temporaries on top of what the unpruned sweep needs.  When one of
    those cannot be had (HSD_TOPK_NOMEM rehearses the failed allocation half-way through the setup) the call must not fail: the setup is released and the
    unpruned sweep -- which worked before the pruned one existed -- returns the same lists; the next call sets up again."""
    from hsd_amd.graph import barabasi_albert
    n, k = 20000 + 5, 12
    g = barabasi_albert(n, 5, 3)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    engine.pair_prepare(0, 0, 0, 3)
    with _env(HSD_TOPK_PRUNE=0):
        ridx, rval = engine.pair_topk(k)
        bidx, bval = engine.pair_topk(k, 4096, 9000)
    for rows, want in (((0, None), (ridx, rval)), ((4096, 9000), (bidx, bval))):
        with _env(HSD_TOPK_NOMEM=1, HSD_TOPK_VERBOSE=1, HSD_STREAM=1):
            capfd.readouterr()
            idx, val = engine.pair_topk(k, *rows)
            log = capfd.readouterr().err
        assert "the unpruned sweep takes the call" in log and "HSD_TOPK_NOMEM" in log, log[-600:]
        assert np.array_equal(idx, want[0]) and np.array_equal(val, want[1])
        with _env(HSD_TOPK_VERBOSE=1, HSD_STREAM=1):
            capfd.readouterr()
            idx, val = engine.pair_topk(k, *rows)                  # no failure this time: the pruned sweep sets up and runs
            log = capfd.readouterr().err
        assert "pruned launch" in log and "overflow 0" in log and "the unpruned sweep takes the call" not in log
        assert np.array_equal(idx, want[0]) and np.array_equal(val, want[1])
