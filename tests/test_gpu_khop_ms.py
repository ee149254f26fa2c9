"""Bit-parallel k-hop sweep (csrc/khop_ms.hip) against the per-source sweep (csrc/khop.hip) and the oracle.

Both build the per-node per-hop histograms of tools/hierarchy.py:16-38 rings; membership is set arithmetic, so the
two must agree bit for bit on every graph, hop count and row range."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _counts(engine, hop, ms, r0=0, r1=None, node_bin=None, support=None):
    os.environ["HSD_KHOP_MS"] = ms
    try:
        if node_bin is None:
            engine.signature_build_degree(hop, r0, engine.n if r1 is None else r1)
        else:
            engine.signature_build(hop, node_bin, support, r0, engine.n if r1 is None else r1)
        return engine.signature_counts(r0, r1)
    finally:
        os.environ.pop("HSD_KHOP_MS", None)


def _graphs():
    from hsd_amd.graph import barabasi_albert, erdos_renyi
    from conftest import load_graph
    return {
        "karate": load_graph("karate"),                       # one partial word
        "tree": load_graph("tree"),
        "er700": erdos_renyi(700, 0.012, seed=11),
        "er_sparse": erdos_renyi(3000, 0.0004, seed=5),       # isolated nodes, small components, empty rings
        "ba5000": barabasi_albert(5000, 4, 7),                # two batches, hub rows split into chunks
    }


@pytest.mark.parametrize("name", ["karate", "tree", "er700", "er_sparse", "ba5000"])
@pytest.mark.parametrize("hop", [0, 1, 2, 3, 5])
def test_ms_sweep_equals_per_source_sweep(engine, name, hop):
    g = _graphs()[name]
    engine.set_graph(g.rowptr, g.col)
    a = _counts(engine, hop, "0")
    b = _counts(engine, hop, "1")
    assert a.shape == b.shape == (g.n_node, hop + 1, len(np.unique(g.degree)))
    assert np.array_equal(a, b)
    assert np.array_equal(b[:, 0].sum(axis=1), np.ones(g.n_node, dtype=np.uint64))


def test_ms_sweep_against_oracle_rings(engine):
    from oracle import hsd_oracle as orc
    g = _graphs()["er700"]
    engine.set_graph(g.rowptr, g.col)
    counts = _counts(engine, 3, "1")
    rings = orc.khop_rings(g.rowptr, g.col, 3)
    support, ref = orc.degree_histograms(g.degree, rings, 4)
    assert np.array_equal(engine.signature_support(), support)
    assert np.array_equal(counts.astype(np.int64), ref)


@pytest.mark.parametrize("r0,r1", [(0, 100), (37, 4200), (4096, 5000), (4999, 5000)])
def test_ms_sweep_row_ranges(engine, r0, r1):
    """Row blocks (the multi-GPU partition) that start and end inside words and batches."""
    g = _graphs()["ba5000"]
    engine.set_graph(g.rowptr, g.col)
    full = _counts(engine, 3, "0")
    part = _counts(engine, 3, "1", r0, r1)
    assert np.array_equal(part, full[r0:r1])
    # rows outside the range keep what they held
    assert np.array_equal(engine.signature_counts(), full)


def test_ms_sweep_duplicate_edges_and_self_loops(engine):
    """A CSR handed over through the C ABI may list an edge twice (data/graph/PTC, cox2) or hold a self loop:
    a ring is a set, both sweeps count every member once."""
    rng = np.random.default_rng(3)
    n = 400
    src = rng.integers(0, n, 1500)
    dst = rng.integers(0, n, 1500)
    src = np.concatenate([src, src[:300], np.arange(0, n, 7)])        # duplicates + self loops
    dst = np.concatenate([dst, dst[:300], np.arange(0, n, 7)])
    a = np.concatenate([src, dst])
    b = np.concatenate([dst, src])
    order = np.lexsort((b, a))
    a, b = a[order], b[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(a, minlength=n), out=rowptr[1:])
    engine.set_graph(rowptr, b.astype(np.int32))
    node_bin = (np.arange(n) % 9).astype(np.int32)
    support = np.arange(9, dtype=np.float64)
    for hop in (1, 3):
        x = _counts(engine, hop, "0", node_bin=node_bin, support=support)
        y = _counts(engine, hop, "1", node_bin=node_bin, support=support)
        assert np.array_equal(x, y)


def test_ms_sweep_wide_support_falls_back(engine):
    """More bins than 64 sources' counters fit in LDS: the per-source sweep serves the call."""
    from hsd_amd.graph import erdos_renyi
    g = erdos_renyi(900, 0.01, seed=2)
    engine.set_graph(g.rowptr, g.col)
    node_bin = np.arange(g.n_node, dtype=np.int32)
    support = np.arange(g.n_node, dtype=np.float64)
    x = _counts(engine, 2, "0", node_bin=node_bin, support=support)
    y = _counts(engine, 2, "1", node_bin=node_bin, support=support)
    assert np.array_equal(x, y)


def test_ms_sweep_default_for_large_graphs(engine):
    """N >= 8192 takes the bit-parallel sweep by default; checked against the oracle on sampled nodes."""
    from hsd_amd.graph import barabasi_albert
    from oracle import c_oracle as corc
    g = barabasi_albert(20000, 5, 42)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(4)
    counts = engine.signature_counts()
    ref = _counts(engine, 4, "0")
    assert np.array_equal(counts, ref)
    nodes = np.array([0, 1, 4095, 4096, 12345, 19999])
    nr = corc.NodeRingSamples(g.rowptr, g.col, 4, g.degree.astype(np.float64), nodes)
    support = engine.signature_support()
    for v in nodes:
        assert np.array_equal(counts[v].astype(np.int64), nr.histogram(int(v), support))


@pytest.mark.parametrize("n,hop", [(100000, 3), (50000, 5)])
def test_ms_sweep_full_size_configs(engine, n, hop):
    """The target (BA N = 100 000, K = 3) and config 3 (BA N = 50 000, K = 5) at full size: every count of the
    bit-parallel sweep equals the per-source sweep's, and the ring sizes they imply equal hsd_khop_sizes."""
    from hsd_amd.graph import barabasi_albert
    g = barabasi_albert(n, 5, 42)
    engine.set_graph(g.rowptr, g.col)
    a = _counts(engine, hop, "1")
    sizes = a.sum(axis=2, dtype=np.int64)
    os.environ["HSD_KHOP_MS"] = "0"
    try:
        per_source = engine.khop_sizes(hop, 0, 3000)
    finally:
        os.environ.pop("HSD_KHOP_MS", None)
    assert np.array_equal(sizes[:3000], per_source)
    assert np.array_equal(sizes, engine.khop_sizes(hop))          # default for a graph of this size: one-bin bit-parallel sweep
    b = _counts(engine, hop, "0")
    assert np.array_equal(a, b)
    # every source sees the whole component at most once
    assert sizes.sum(axis=1).max() <= n


@pytest.mark.parametrize("name", ["karate", "er_sparse", "ba5000"])
@pytest.mark.parametrize("hop", [0, 1, 2, 4])
def test_ring_sizes_by_the_one_bin_sweep(engine, name, hop):
    """hsd_khop_sizes (hierarchical degree, SIR labels) through the bit-parallel sweep with a single bin, whole graph and a
    row range, against the per-source sweep and the C oracle."""
    from oracle import c_oracle as corc
    g = _graphs()[name]
    engine.set_graph(g.rowptr, g.col)
    n = g.n_node
    out = {}
    for ms in ("0", "1"):
        os.environ["HSD_KHOP_MS"] = ms
        try:
            out[ms] = (engine.khop_sizes(hop), engine.khop_sizes(hop, n // 3, n - 5))
        finally:
            os.environ.pop("HSD_KHOP_MS", None)
    assert np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1], out["1"][1])
    assert np.array_equal(out["1"][0], corc.khop_sizes(g.rowptr, g.col, hop))
    assert np.array_equal(out["1"][1], out["1"][0][n // 3:n - 5])


def test_ms_sweep_falls_back_when_the_planes_cannot_be_allocated(engine):
    """khop.hip: if the bit-parallel sweep cannot get its planes (3 n x 512 B + ring planes) the per-source sweep takes
    over instead of failing -- histograms and ring sizes, same counts (HSD_KHOP_MS_NOPLANES rehearses the failure)."""
    g = _graphs()["ba5000"]
    engine.set_graph(g.rowptr, g.col)
    ref = _counts(engine, 3, "0")
    sizes_ref = engine.khop_sizes(3)
    engine.set_graph(g.rowptr, g.col)                 # drops the planes a previous test may have left
    os.environ["HSD_KHOP_MS_NOPLANES"] = "1"
    try:
        got = _counts(engine, 3, "1")
        os.environ["HSD_KHOP_MS"] = "1"
        sizes = engine.khop_sizes(3)
    finally:
        os.environ.pop("HSD_KHOP_MS_NOPLANES", None)
        os.environ.pop("HSD_KHOP_MS", None)
    assert np.array_equal(got, ref) and np.array_equal(sizes, sizes_ref)
