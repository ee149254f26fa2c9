"""GPU parity: the HIP path (through the C ABI) against the golden fixtures and the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_graph

pytestmark = pytest.mark.gpu

RINGS = np.load(os.path.join(GOLDEN, "rings.npz"))
DEG = np.load(os.path.join(GOLDEN, "dist_degree.npz"))
RTOL, ATOL = 1e-6, 1e-12      # north star: fp64 distances within 1e-6 relative


def _graph(name):
    if name == "er500":
        from hsd_amd.graph import Graph
        return Graph(list(range(500)), DEG["er500/rowptr"], DEG["er500/col"])
    return load_graph(name)


@pytest.mark.parametrize("name", ["PTC", "citeseer", "usa", "cora"])
def test_khop_membership_of_the_larger_shipped_graphs(engine, name):
    """The reference's data files with the awkward line formats (PTC lists every edge twice, citeseer has lines with a third column,
    ids that do not start at 0): node order as networkx gives it, ring members at K = 2 and ring sizes at K = 3 as
    tools.get_hierarchical_representation computed them (tests/golden/generate_golden_rings_more.py), bit for bit; the degree histograms
    of both sweeps (per-source and bit-parallel) are the histograms of those rings."""
    import os
    R = np.load(os.path.join(GOLDEN, "rings_more.npz"))
    g = load_graph(name)
    assert g.nodes == list(R[f"{name}/nodes"]) and g.n_edge == int(R[f"{name}/n_edges"][0])
    engine.set_graph(g.rowptr, g.col)
    off, mem = engine.khop_members(2)
    assert np.array_equal(off, R[f"{name}/K2/offsets"]) and np.array_equal(mem, R[f"{name}/K2/members"])
    assert np.array_equal(engine.khop_sizes(3), R[f"{name}/K3/sizes"])
    support = np.unique(g.degree)
    binof = np.searchsorted(support, g.degree)
    for force in ("0", "1"):
        os.environ["HSD_KHOP_MS"] = force
        try:
            engine.signature_build_degree(2)
        finally:
            os.environ.pop("HSD_KHOP_MS", None)
        counts = engine.signature_counts()
        assert counts.shape == (g.n_node, 3, support.size)
        for v in (0, g.n_node // 2, g.n_node - 1):
            for k in range(3):
                ring = mem[off[v * 3 + k]:off[v * 3 + k + 1]]
                assert np.array_equal(counts[v, k], np.bincount(binof[ring], minlength=support.size)), (name, force, v, k)


@pytest.mark.parametrize("name", ["karate", "barbell", "mkarate", "tree", "robust", "line", "europe"])
@pytest.mark.parametrize("K", [2, 3, 5])
def test_khop_membership_bit_exact(engine, name, K):
    g = load_graph(name)
    engine.set_graph(g.rowptr, g.col)
    off, mem = engine.khop_members(K)
    assert np.array_equal(off, RINGS[f"{name}/K{K}/offsets"])
    assert np.array_equal(mem, RINGS[f"{name}/K{K}/members"])
    sizes = engine.khop_sizes(K)
    assert np.array_equal(sizes.reshape(-1), np.diff(off))
    # a sub-range of sources gives the same rings
    n = g.n_node
    a, b = n // 3, min(n, n // 3 + 5)
    off2, mem2 = engine.khop_members(K, a, b)
    lo, hi = off[a * (K + 1)], off[b * (K + 1)]
    assert np.array_equal(mem2, mem[lo:hi])


@pytest.mark.parametrize("name", ["karate", "tree", "europe", "er500"])
def test_degree_histograms_bit_exact(engine, name):
    from oracle import hsd_oracle as orc
    g = _graph(name)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    counts = engine.signature_counts()
    rings = orc.khop_rings(g.rowptr, g.col, 3)
    support, ref = orc.degree_histograms(g.degree, rings, 4)
    assert np.array_equal(engine.signature_support(), support)
    assert np.array_equal(counts.astype(np.int64), ref)


@pytest.mark.parametrize("name", ["karate", "tree", "europe"])
def test_w1_variant_a_golden(engine, name):
    g = _graph(name)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    engine.pair_prepare(0, 0, 0, 3)
    D = engine.pair_rows_host()
    G = DEG[f"{name}/A_raw"]
    assert np.allclose(D, G, rtol=RTOL, atol=ATOL), np.abs(D - G).max()
    assert np.array_equal(D, D.T) and np.all(np.diag(D) == 0.0)


@pytest.mark.parametrize("name", ["karate", "tree", "europe", "er500"])
def test_w1_variant_b_zeropad_golden(engine, name):
    g = _graph(name)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    engine.pair_prepare(0, 1, 0, 2)
    assert engine.pair_operand_dims()[1] == 4          # exact uint32 path
    D = engine.pair_rows_host()
    G = DEG[f"{name}/B_zeropad"]
    assert np.allclose(D, G, rtol=RTOL, atol=ATOL), np.abs(D - G).max()
    assert np.array_equal(D, D.T) and np.all(np.diag(D) == 0.0)


def test_variant_a_empty_ring_raises(engine):
    g = _graph("er500")
    assert bool(DEG["er500/has_empty_ring"][0])
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    with pytest.raises(ValueError, match="Distribution can't be empty."):
        engine.pair_prepare(0, 0, 0, 3)


@pytest.mark.parametrize("name", ["karate", "tree"])
def test_kl_golden(engine, name):
    g = _graph(name)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    n = g.n_node
    iu = np.triu_indices(n, 1)
    engine.pair_prepare(1, 0, 0, 3, 0.5, False)
    D = engine.pair_rows_host()
    G = DEG[f"{name}/KL_a0.5"]
    assert np.allclose(D[iu], G[iu], rtol=RTOL, atol=ATOL)
    assert np.array_equal(D, D.T)                       # upper triangle mirrored (model/HSD.py:132-134)
    engine.pair_prepare(2, 0, 0, 3, 0.5, False)
    S = engine.pair_rows_host()
    assert np.allclose(S, DEG[f"{name}/SKL_a0.5"], rtol=RTOL, atol=ATOL)
    engine.pair_prepare(1, 0, 0, 3, 0.5, True)          # calculate_distance's sort of both vectors
    Ds = engine.pair_rows_host()
    Gs = DEG[f"{name}/KLsorted_a0.5"]
    assert np.allclose(Ds[iu], Gs[iu], rtol=RTOL, atol=ATOL)


KLM = np.load(os.path.join(GOLDEN, "kl_midsize.npz"))


@pytest.mark.parametrize("name", ["usa", "cora"])
@pytest.mark.parametrize("stream", ["0", "1"])
def test_kl_golden_midsize(engine, name, stream):
    """tests/golden/generate_golden_kl_midsize.py: the reference's own KL_divergence / calculate_distance on sampled pairs and whole
    matrix rows of mid-size graphs -- sizes where the sparse-column form trims columns per tile (usa: 144 bins) -- through both kernels."""
    g = _graph(name)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    pr, rows = KLM[f"{name}/pairs"], KLM[f"{name}/rows"]
    fwd = pr[:, 0] < pr[:, 1]                              # the matrix holds KL(P_i || P_j) for i < j and its mirror below (model/HSD.py:132-134)

    def run(kind, sort):
        engine.pair_prepare(kind, 0, 0, 3, 0.5, sort)
        return engine.pair_rows_host()

    D = _with_env("HSD_STREAM", stream, lambda: run(1, False))
    assert fwd.sum() > 200 and np.allclose(D[pr[fwd, 0], pr[fwd, 1]], KLM[f"{name}/KL"][fwd], rtol=RTOL, atol=ATOL)
    assert np.allclose(D[rows], KLM[f"{name}/rows_KL"], rtol=RTOL, atol=ATOL)
    assert np.array_equal(D, D.T)
    S = _with_env("HSD_STREAM", stream, lambda: run(2, False))
    assert np.allclose(S[pr[:, 0], pr[:, 1]], KLM[f"{name}/SKL"], rtol=RTOL, atol=ATOL)
    assert np.allclose(S[rows], KLM[f"{name}/rows_SKL"], rtol=RTOL, atol=ATOL)
    Ds = _with_env("HSD_STREAM", stream, lambda: run(1, True))
    assert np.allclose(Ds[pr[fwd, 0], pr[fwd, 1]], KLM[f"{name}/KLsorted"][fwd], rtol=RTOL, atol=ATOL)


def _with_env(name, value, fn):
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old


@pytest.mark.parametrize("n,p,K", [(1500, 0.006, 3), (700, 0.02, 2), (5000, 0.002, 3)])
def test_kl_ring0_closed_form_equals_the_dense_layer(engine, n, p, K):
    """Ring 0 of a node is the point mass at its own degree bin, so layer 0 of KL / symmetric KL (tools/metrics.py:74-88 on the smoothed
    histograms) is a two-valued function of `same bin?` -- taken in the tile epilogue instead of B dense columns.  Same matrix as with the
    layer in the dense loop (HSD_KL_POINT=0) up to the rounding of one added constant, B fewer operand columns, for the persistent and the
    one-tile kernel, whole matrix and row block, every alpha; sorted histograms (calculate_distance's np.sort) of two point masses are equal."""
    from hsd_amd.graph import erdos_renyi
    g = erdos_renyi(n, p, 5)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(K)
    B = int(engine.signature_dims()[1])
    for metric in (1, 2):
        for alpha, srt in ((0.5, False), (0.01, False), (2.0, False), (0.5, True)):
            def matrix(stream):
                def run():
                    engine.pair_prepare(metric, 0, 0, K, alpha, srt)
                    cols = engine.pair_operand_dims()[0]
                    return engine.pair_rows_host(), engine.pair_rows_host(n // 3, n // 3 + 200), cols
                return _with_env("HSD_STREAM", stream, run)
            dense, dense_blk, cols_dense = _with_env("HSD_KL_POINT", "0", lambda: matrix("1"))
            for stream in ("1", "0"):
                D, blk, cols = matrix(stream)
                # (columns of a half: the layers' bins + -- sparse-column form, round 5 -- one column per layer and one for all of them)
                f = 2 if metric == 2 else 1
                assert cols_dense == f * ((K + 1) * B + (K + 1) + 1)
                assert cols == f * (K * B + K + 1) + (0 if srt else 1)
                assert np.allclose(D, dense, rtol=1e-12, atol=1e-13), (metric, alpha, srt, stream, np.abs(D - dense).max())
                assert np.allclose(blk, dense_blk, rtol=1e-12, atol=1e-13)
                assert np.array_equal(blk, D[n // 3:n // 3 + 200]) or metric == 2
    # ring 0 alone (layers 0..0) and alpha = 0 keep the dense loop (nothing else to compute / log 0)
    engine.pair_prepare(1, 0, 0, 0, 0.5, False)
    assert engine.pair_operand_dims()[0] == B + 2
    D0 = engine.pair_rows_host()
    deg = g.degree
    want = np.where(deg[:, None] != deg[None, :], np.log(1.5 / 0.5) / (1 + 0.5 * B), 0.0)
    assert np.allclose(D0, want, rtol=1e-12, atol=1e-14)


def test_kl_ring0_closed_form_needs_ring0_to_be_the_node(engine):
    """Caller-assigned counts (hsd_signature_set_counts: the route of caller-assigned hierarchies, model/HSD.py:36 + tests/robust_test/main.py:179)
    whose ring 0 is not the node itself: the closed form does not apply, the layer stays dense -- the same matrix as HSD_KL_POINT=0."""
    from hsd_amd.graph import erdos_renyi
    g = erdos_renyi(900, 0.01, 9)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(2)
    B = int(engine.signature_dims()[1])
    counts = engine.signature_counts().copy()
    counts[17, 0] = counts[17, 1]                  # node 17's "ring 0" is its neighbourhood
    engine.signature_set_counts(0, g.n_node, counts)
    engine.pair_prepare(1, 0, 0, 2, 0.5, False)
    assert engine.pair_operand_dims()[0] == 3 * B + 3 + 1
    D = engine.pair_rows_host()
    ref = _with_env("HSD_KL_POINT", "0", lambda: (engine.pair_prepare(1, 0, 0, 2, 0.5, False), engine.pair_rows_host())[1])
    assert np.array_equal(D, ref)
    engine.signature_build_degree(2)               # back to the graph's own rings: closed form again
    engine.pair_prepare(1, 0, 0, 2, 0.5, False)
    assert engine.pair_operand_dims()[0] == 2 * B + 2 + 1 + 1


@pytest.mark.parametrize("n,p,K,seed", [(2000, 0.004, 3, 1), (3001, 0.003, 2, 2), (777, 0.02, 3, 3)])
def test_w1_vs_c_oracle_medium(engine, n, p, K, seed):
    """Seeded ER graphs (odd sizes, ragged tiles) against the C oracle on sampled rows."""
    from oracle import c_oracle as corc
    from hsd_amd.graph import erdos_renyi
    g = erdos_renyi(n, p, seed)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(K)
    rs = corc.RingSamples(g.rowptr, g.col, K, g.degree.astype(np.float64))
    sizes = engine.khop_sizes(K)
    assert np.array_equal(sizes.reshape(-1), np.diff(rs.offsets))
    rows = [(0, 40), (n // 2 - 3, n // 2 + 30), (n - 25, n)]
    for align, lo, hi in [(1, 0, K - 1), (0, 0, K)]:
        ref_any_empty = bool((sizes[:, lo:hi + 1] == 0).any())
        if align == 0 and ref_any_empty:
            with pytest.raises(ValueError):
                engine.pair_prepare(0, align, lo, hi)
            continue
        engine.pair_prepare(0, align, lo, hi)
        D = engine.pair_rows_host()
        assert np.array_equal(D, D.T) and np.all(np.diag(D) == 0.0)
        for a, b in rows:
            R, _ = rs.pair_rows(lo, hi, align, a, b)
            assert np.allclose(D[a:b], R, rtol=RTOL, atol=ATOL), (align, a, np.abs(D[a:b] - R).max())
            # row-block entry point == slice of the full matrix (multi-GPU row partition)
            Dr = engine.pair_rows_host(a, b)
            assert np.array_equal(Dr, D[a:b])


def test_resident_matrix_and_stats(engine):
    from hsd_amd.graph import erdos_renyi
    g = erdos_renyi(1500, 0.01, 5)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(2)
    engine.pair_prepare(0, 1, 0, 1)
    M = engine.pair_rows_device()
    engine.sync()
    D = M.to_host()
    H = engine.pair_rows_host()
    assert np.array_equal(D, H)
    st = M.stats()
    assert st["max_abs_diag"] == 0.0 and st["max_asym"] == 0.0
    assert np.isclose(st["sum"], H.sum(), rtol=1e-12) and st["max"] == H.max() and st["min"] == 0.0
    M.free()


def test_model_surface_degree(engine):
    """HSD class mirrors model/HSD.py: Variant A / Variant B through the Python surface."""
    import hsd_amd
    g = load_graph("karate")
    m = hsd_amd.HSD(g, "karate", 0, 3, "wasserstein", signature="degree", engine=engine)
    assert m.nodes[:5] == ['2', '1', '3', '4', '5'] and m.n_node == 34
    DA = m.calculate_structural_distance(1.0)
    DB = m.parallel_calculate_HSD(n_workers=10)
    assert np.allclose(DA, DEG["karate/A_raw"], rtol=RTOL, atol=ATOL)
    assert np.allclose(DB, DEG["karate/B_zeropad"], rtol=RTOL, atol=ATOL)
    assert m.distMat is DB
    T = np.load(os.path.join(GOLDEN, "triple_karate.npz"))
    hd = m.get_nodes_hierarchical_degree()
    assert np.array_equal(np.asarray([hd[v] for v in m.nodes]), T["hdeg"])
    # hierarchy dict has the reference's shape: node -> hop+1 lists of node labels
    h = m.hierarchy
    assert h['1'][0] == ['1'] and len(h['1']) == 4
    # a caller-assigned hierarchy is honoured (tests/robust_test/main.py:179)
    m2 = hsd_amd.HSD(g, "karate", 0, 3, "wasserstein", signature="degree", engine=engine)
    m2.hierarchy = h
    assert np.array_equal(m2.calculate_structural_distance(1.0), DA)
    with pytest.raises(NotImplementedError):
        hsd_amd.HSD(g, "karate", 0, 3, "cosine", signature="degree", engine=engine).parallel_calculate_HSD()
    with pytest.raises(TypeError):
        hsd_amd.HSD(g, "karate", 0, 3, None, signature="degree", engine=engine).parallel_calculate_HSD()


def test_two_rank_exchange_on_one_gpu(engine):
    """The multi-GPU flow with both 'ranks' on one device: each builds its own block of sources,
    the counts are exchanged through host memory (the gloo fallback path of
    hsd_amd.distributed), each computes its row block -- W1 and KL (below-diagonal blocks of KL
    use the swapped operand roles)."""
    from hsd_amd.engine import Engine
    from hsd_amd.distributed import shard_rows
    from hsd_amd.graph import erdos_renyi
    g = erdos_renyi(900, 0.012, 4)
    n = g.n_node
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(2)
    full = {}
    for key, args in {"w1b": (0, 1, 0, 1), "kl": (1, 0, 0, 2), "skl": (2, 0, 0, 2)}.items():
        engine.pair_prepare(*args)
        full[key] = engine.pair_rows_host()
    other = Engine(0)
    try:
        ranks = [engine, other]
        for r, e in enumerate(ranks):
            e.set_graph(g.rowptr, g.col)
            a, b = shard_rows(n, 2, r)
            e.signature_build_degree(2, a, b)
        blocks = [e.signature_counts(*shard_rows(n, 2, r)) for r, e in enumerate(ranks)]
        for r, e in enumerate(ranks):
            a, b = shard_rows(n, 2, 1 - r)
            e.signature_set_counts(a, b, blocks[1 - r])
        for key, args in {"w1b": (0, 1, 0, 1), "kl": (1, 0, 0, 2), "skl": (2, 0, 0, 2)}.items():
            rows = []
            for r, e in enumerate(ranks):
                e.pair_prepare(*args)
                rows.append(e.pair_rows_host(*shard_rows(n, 2, r)))
            got = np.concatenate(rows)
            if key == "skl":      # the two dot products are accumulated in the other order below the diagonal
                assert np.allclose(got, full[key], rtol=1e-13, atol=1e-15), key
            else:
                assert np.array_equal(got, full[key]), key
    finally:
        other.close()


@pytest.mark.parametrize("flag", ["0", "1"])
def test_khop_bottom_up_expansion(engine, flag):
    """Deep sweep of a small-world graph: the last hops switch to the bottom-up expansion (every unvisited
    node looks for a neighbour in the frontier bitmap).  Rings, members and histograms must not depend on it."""
    from hsd_amd.graph import barabasi_albert
    from oracle import c_oracle as corc
    from oracle import hsd_oracle as orc
    g = barabasi_albert(3000, 4, 7)
    os.environ["HSD_KHOP_BOTTOMUP"] = flag
    try:
        engine.set_graph(g.rowptr, g.col)
        sizes = engine.khop_sizes(5)
        off, mem = engine.khop_members(5, 100, 140)
        engine.signature_build_degree(5)
        counts = engine.signature_counts(0, 50)
    finally:
        os.environ.pop("HSD_KHOP_BOTTOMUP", None)
    assert np.array_equal(sizes, corc.khop_sizes(g.rowptr, g.col, 5))
    assert sizes[:, 4].max() > 1000                       # the frontier does get large enough to switch
    rings = [orc.khop_rings_node(g.rowptr, g.col, i, 5) for i in range(100, 140)]
    assert np.array_equal(mem, np.concatenate([np.concatenate(r) for r in rings]))
    nr = corc.NodeRingSamples(g.rowptr, g.col, 5, g.degree.astype(np.float64), np.arange(50))
    support = engine.signature_support()
    for v in (0, 7, 49):
        assert np.array_equal(counts[v].astype(np.int64), nr.histogram(v, support))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,row0", [(700, 700, 0), (333, 1500, 128), (64, 64, 0), (200, 1000, 800), (129, 321, 0)])
def test_matrix_stats_against_numpy(engine, rows, cols, row0):
    """hsd_matrix_stats (tile-wise: every entry read once, the mirror transposed through LDS) on a block with planted
    asymmetry and a non-zero diagonal: sum / diagonal / asymmetry of the square part / max / min against numpy."""
    rng = np.random.default_rng(rows * 7 + cols)
    a = rng.standard_normal((rows, cols))
    sq = a[:, row0:row0 + rows]
    sym = 0.5 * (sq + sq.T)                      # symmetric square part ...
    a[:, row0:row0 + rows] = sym
    i, j = rows // 3, rows - 2
    a[i, row0 + j] += 0.375                      # ... with one planted asymmetric pair
    M = engine.upload_matrix(a)
    try:
        s = engine.matrix_stats(M.ptr, rows, cols, M.ld, row0)
    finally:
        M.free()
    sqa = a[:, row0:row0 + rows]
    assert np.isclose(s[0], a.sum(), rtol=1e-11, atol=1e-9)
    assert s[1] == np.abs(np.diag(sqa)).max()
    assert s[2] == np.abs(sqa - sqa.T).max() == 0.375
    assert s[3] == a.max() and s[4] == a.min()


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,row0", [(70, 100, 100), (130, 333, 400), (64, 65, 65)])
def test_matrix_stats_block_without_a_square_part(engine, rows, cols, row0):
    """row0 >= cols: the block has no square part and only 'left' tiles, the last of them partial (cols % 64 != 0).  The matrix is
    uploaded with a padded leading dimension whose padding holds huge values: a tile that ran past the last column would see them."""
    rng = np.random.default_rng(rows + cols)
    a = rng.standard_normal((rows, cols))
    ld = (cols + 15) // 16 * 16 + 16
    padded = np.full((rows, ld), 1e300)
    padded[:, :cols] = a
    ptr = engine.malloc(padded.nbytes)
    try:
        engine.memcpy_h2d(ptr, padded, padded.nbytes)
        s = engine.matrix_stats(ptr, rows, cols, ld, row0)
    finally:
        engine.free(ptr)
    assert np.isclose(s[0], a.sum(), rtol=1e-11, atol=1e-9)
    assert s[1] == 0.0 and s[2] == 0.0
    assert s[3] == a.max() and s[4] == a.min()


def test_engine_close_returns_the_hbm_of_live_matrices():
    """DeviceMatrix allocations are tracked by their Engine: close() frees the ones still alive (hsd_destroy does not know
    about hsd_malloc blocks), and a later free() / finalizer of such a matrix is a no-op."""
    import hsd_amd
    from hsd_amd.graph import erdos_renyi
    g = erdos_renyi(300, 0.03, 3)
    eng = hsd_amd.Engine(0)
    eng.set_graph(g.rowptr, g.col)
    eng.signature_build_degree(2)
    eng.pair_prepare(0, 1, 0, 1)
    M = eng.pair_rows_device()
    eng.sync()
    assert M.ptr and len(eng._matrices) == 1
    eng.close()
    assert M.ptr is None and len(eng._matrices) == 0
    M.free()                                    # no-op, no call into a destroyed context
    del M


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["single_node", "single_edge", "three_isolated", "complete_20", "path_5_deep_hop", "star_300"])
def test_degenerate_graphs(engine, kind):
    """The smallest and most uniform inputs: one node, one edge, no edges at all, a complete graph (every pair tied at 0), a path swept
    deeper than its diameter (empty outer rings), a star.  Ring sizes against the C oracle; Variant B (zero-pad: defined on empty rings)
    against the oracle on every pair; Variant A raises like scipy wherever a compared ring is empty and matches otherwise; the k-nearest
    consumer on the all-ties matrix orders by index."""
    from oracle import c_oracle as corc
    from hsd_amd.graph import Graph
    K = 3
    if kind == "single_node":
        n, src, dst = 1, [], []
    elif kind == "single_edge":
        n, src, dst = 2, [0], [1]
    elif kind == "three_isolated":
        n, src, dst = 3, [], []
    elif kind == "complete_20":
        n = 20
        ii, jj = np.triu_indices(n, 1)
        src, dst = ii, jj
    elif kind == "path_5_deep_hop":
        n, src, dst, K = 5, [0, 1, 2, 3], [1, 2, 3, 4], 6
    else:
        n = 300
        src, dst = np.zeros(n - 1, dtype=np.int64), np.arange(1, n)
    g = Graph.from_edges(list(range(n)), np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64))
    engine.set_graph(g.rowptr, g.col)
    sizes = engine.khop_sizes(K)
    assert np.array_equal(sizes, corc.khop_sizes(g.rowptr, g.col, K))
    engine.signature_build_degree(K)
    rs = corc.RingSamples(g.rowptr, g.col, K, g.degree.astype(np.float64))
    engine.pair_prepare(0, 1, 0, K - 1)
    DB = engine.pair_rows_host()
    RB, _ = rs.pair_rows(0, K - 1, corc.ALIGN_ZEROPAD, 0, n)
    assert DB.shape == (n, n) and np.allclose(DB, RB, rtol=1e-6, atol=1e-12) and np.array_equal(DB, DB.T)
    if (sizes == 0).any():
        with pytest.raises(ValueError, match="Distribution can't be empty."):
            engine.pair_prepare(0, 0, 0, K)
    else:
        engine.pair_prepare(0, 0, 0, K)
        DA = engine.pair_rows_host()
        RA, _ = rs.pair_rows(0, K, corc.ALIGN_NONE, 0, n)
        assert np.allclose(DA, RA, rtol=1e-6, atol=1e-12)
    if kind == "complete_20":
        assert not DB.any()
        engine.pair_prepare(0, 1, 0, K - 1)
        idx, val = engine.pair_topk(5)
        for i in range(n):
            assert list(idx[i]) == [j for j in range(n) if j != i][:5] and not val[i].any()
    if n >= 2:
        engine.pair_prepare(0, 1, 0, K - 1)
        idx, val = engine.pair_topk(1)
        Dm = DB.copy()
        np.fill_diagonal(Dm, np.inf)
        assert np.array_equal(val[:, 0], Dm.min(axis=1))
    else:
        with pytest.raises(ValueError):
            engine.pair_prepare(0, 1, 0, K - 1)
            engine.pair_topk(1)                       # k must be <= n - 1


@pytest.mark.gpu
def test_contexts_return_their_hbm(engine):
    """Create / use / destroy a context several times (graph, sweep, every distance kind, row blocks, the k-nearest consumer, the heat
    path, resident matrices left to Engine.close()): the device's free memory afterwards is what it was before (hipMemGetInfo, asked of
    the HIP runtime directly -- the library has no such entry point)."""
    import ctypes
    import hsd_amd
    from hsd_amd.graph import barabasi_albert
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        engine.sync()
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value

    g = barabasi_albert(9000 + 33, 5, 3)
    kar = load_graph("karate")

    def cycle():
        eng = hsd_amd.Engine(0)
        eng.set_graph(g.rowptr, g.col)
        eng.signature_build_degree(3)
        os.environ["HSD_STREAM"] = "1"
        try:
            for mode in ((0, 1, 0, 2), (0, 0, 0, 3), (1, 0, 0, 3), (2, 0, 0, 3)):
                eng.pair_prepare(*mode)
                M = eng.pair_rows_device(1280, 4096)          # left alive: close() must return it
                eng.pair_rows_host(0, 300)
                if mode[0] != 1:
                    eng.pair_topk(7, 100, 900)
                    eng.pair_topk(5)                          # whole graph: raw fp64 W1 builds its node-permuted operand copy (pruned sweep)
        finally:
            os.environ.pop("HSD_STREAM", None)
        m = hsd_amd.HSD(kar, "karate", 1.0, 3, "wasserstein", engine=eng)
        m.calculate_structural_distance(1.0, approx=True)
        assert M.ptr
        eng.close()

    cycle()                                       # first cycle: code objects, runtime pools
    before = free_bytes()
    for _ in range(3):
        cycle()
    after = free_bytes()
    assert before - after < 32 << 20, (before, after, before - after)


@pytest.mark.parametrize("n,p,K", [(6000, None, 3), (1111, 0.01, 2)], ids=["scale-free", "erdos-renyi"])
def test_kl_sparse_columns_and_tile_trimming(engine, n, p, K):
    """Round 5: the KL kinds' operands in the sparse-column form (X = c / Z, Y = log(c + alpha) - log alpha: zero wherever a bin is empty) and the persistent
    kernel running a tile pair only over the columns occupied on BOTH of its sides.  Skipped products are exact zeros, so the trimmed persistent kernel,
    the same kernel over every column and the one-tile kernel give the same bits; against the dense form (HSD_KL_TRIM=0: P . log P) the difference is the
    rounding of a re-associated sum; the kernel's own column count shows that columns were in fact skipped."""
    from hsd_amd.graph import barabasi_albert, erdos_renyi
    g = erdos_renyi(n, p, 12) if p else barabasi_albert(n, 4, 12)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(K)
    for metric in (1, 2):
        for alpha, srt in ((0.5, False), (0.05, False), (0.5, True)):
            def run(stream):
                def f():
                    engine.pair_prepare(metric, 0, 0, K, alpha, srt)
                    D = engine.pair_rows_host()
                    cols = engine.pair_operand_dims()[0]
                    M = engine.pair_rows_device()
                    try:
                        work = engine.pair_work() / (n * (n - 1) / 2.0)
                    finally:
                        M.free()
                    return D, cols, work
                return _with_env("HSD_STREAM", stream, f)
            trimmed, cols, work = run("1")
            one_tile, _, _ = run("0")
            assert np.array_equal(trimmed, one_tile), (metric, alpha, srt)
            # the same persistent kernel over every column (HSD_KL_TRIM=2: what a hierarchy too deep for the column lists' LDS budget gets)
            untrimmed, ucols, uwork = _with_env("HSD_KL_TRIM", "2", lambda: run("1"))
            assert np.array_equal(trimmed, untrimmed) and ucols == cols and uwork >= work
            if p is None:
                assert work < 0.99 * cols, (work, cols)                     # columns were skipped (3 % at this size; 27 % at config 3: outer rings of a large graph hold few hubs)
            dense, dcols, dwork = _with_env("HSD_KL_TRIM", "0", lambda: run("1"))
            assert dcols < cols and (p is not None or dwork > work)
            assert np.allclose(trimmed, dense, rtol=1e-9, atol=1e-12), np.abs(trimmed - dense).max()
    # alpha = 0 keeps the dense form (log 0): whatever it returns or raises, both settings agree
    def a0():
        try:
            engine.pair_prepare(1, 0, 1, K, 0.0, False)
            return engine.pair_rows_host(0, 8)
        except ValueError as e:
            return str(e)
    r1, r0 = a0(), _with_env("HSD_KL_TRIM", "0", a0)
    assert (isinstance(r1, str) and r1 == r0) or np.array_equal(r1, r0, equal_nan=True)
