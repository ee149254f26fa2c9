"""CPU: the oracle (numpy restatement + C restatement) against the golden vectors that
tests/golden/generate_golden.py produced by importing the reference."""
import json
import os

import numpy as np
import pytest

from oracle import hsd_oracle as orc
from oracle import c_oracle as corc
from conftest import GOLDEN, load_graph

RINGS = np.load(os.path.join(GOLDEN, "rings.npz"))


def _flat_rings(rings):
    off, mem = [0], []
    for layers in rings:
        for layer in layers:
            mem.extend(int(x) for x in layer)
            off.append(len(mem))
    return np.asarray(off, dtype=np.int64), np.asarray(mem, dtype=np.int32)


@pytest.mark.parametrize("name", ["karate", "barbell", "mkarate", "tree", "robust", "line"])
@pytest.mark.parametrize("K", [2, 3, 5])
def test_rings_bit_exact(name, K):
    g = load_graph(name)
    assert [str(x) for x in g.nodes] == [str(x) for x in RINGS[f"{name}/nodes"]]
    off, mem = _flat_rings(orc.khop_rings(g.rowptr, g.col, K))
    assert np.array_equal(off, RINGS[f"{name}/K{K}/offsets"])
    assert np.array_equal(mem, RINGS[f"{name}/K{K}/members"])
    sizes = corc.khop_sizes(g.rowptr, g.col, K)
    assert np.array_equal(sizes.reshape(-1), np.diff(off))


def test_rings_europe_c_sizes():
    g = load_graph("europe")
    for K in (2, 3, 5):
        sizes = corc.khop_sizes(g.rowptr, g.col, K)
        assert np.array_equal(sizes.reshape(-1), np.diff(RINGS[f"europe/K{K}/offsets"]))


def test_wavelets_exact():
    W = np.load(os.path.join(GOLDEN, "wavelets.npz"))
    for name in ("karate", "barbell"):
        L = load_graph(name).laplacian_dense()
        for si, s in enumerate(W[f"{name}/scales"]):
            psi = orc.heat_wavelets_exact(L, s)
            assert np.allclose(psi, W[f"{name}/psi{si}"], rtol=0, atol=1e-13)
            # high-order Chebyshev == exact, low-order within truncation error
            lmax = float(W[f"{name}/lmax"][0]) * 1.01
            pc = orc.heat_wavelets_cheby(L, s, lmax, m=60)
            assert np.abs(pc - W[f"{name}/psi{si}"]).max() < 1e-9


@pytest.mark.parametrize("name", ["karate", "barbell", "mkarate"])
def test_heat_distance_variants(name):
    D = np.load(os.path.join(GOLDEN, "dist_heat.npz"))
    g = load_graph(name)
    rings = orc.khop_rings(g.rowptr, g.col, 3)
    psi = orc.heat_wavelets_exact(g.laplacian_dense(), 1.0)
    coeffs = orc.hierarchical_coefficients(psi, rings)
    assert np.allclose(orc.distance_variant_a(coeffs, 3), D[f"{name}/A"], rtol=1e-9, atol=1e-13)
    assert np.allclose(orc.distance_variant_b_rowquirk(psi, rings, 3), D[f"{name}/B"], rtol=1e-9, atol=1e-13)


def test_metric_micro_vectors():
    with open(os.path.join(GOLDEN, "metrics.json")) as f:
        M = json.load(f)
    for rec in M["w1"]:
        d = orc.calculate_distance(rec["p"], rec["q"], "wasserstein")
        assert abs(d - rec["aligned"]) <= 1e-15 + 1e-13 * abs(rec["aligned"])
        dc = corc.w1_zeropad(rec["p"], rec["q"])
        assert abs(dc - rec["aligned"]) <= 1e-14 + 1e-12 * abs(rec["aligned"])
        if "raw" in rec:
            assert abs(orc.w1_samples(rec["p"], rec["q"]) - rec["raw"]) <= 1e-15 + 1e-13 * abs(rec["raw"])
            assert abs(corc.w1_sorted(rec["p"], rec["q"]) - rec["raw"]) <= 1e-14 + 1e-12 * abs(rec["raw"])
    for rec in M["kl"]:
        assert abs(orc.kl_divergence(rec["p"], rec["q"]) - rec["kl"]) <= 1e-15
        assert abs(orc.kl_divergence(rec["p"], rec["q"], True) - rec["skl"]) <= 1e-15
        assert abs(orc.calculate_distance(rec["p"], rec["q"], "kl") - rec["kl_via_calculate_distance"]) <= 1e-15
    for rec in M["align"]:
        a, b = orc.align(rec["p"], rec["q"], rec["normalized"])
        assert np.array_equal(a, rec["pa"]) and np.array_equal(b, rec["qa"])
    k = M["known"]
    assert k["w1_[1..5]_vs_[15]"] == 12.0 and abs(k["w1_empty_vs_[.2,.5]"] - 0.35) < 1e-15
    assert abs(orc.kl_divergence([.2, .3, .5], [.1, .6, .3]) - k["kl_direct"]) < 1e-16
    assert abs(orc.calculate_distance([.2, .3, .5], [.1, .6, .3], "kl") - k["kl_sorted"]) < 1e-16


def test_metric_errors():
    with open(os.path.join(GOLDEN, "metrics.json")) as f:
        E = json.load(f)["errors"]

    def exc(fn, *a):
        try:
            fn(*a)
        except Exception as e:      # noqa: BLE001
            return [type(e).__name__, str(e)]
        return None
    assert exc(orc.w1_samples, [], [1.0]) == E["scipy_empty"]
    assert exc(orc.calculate_distance, [1.0], [1.0], None) == E["metric_none"]
    assert exc(orc.calculate_distance, [1.0], [1.0], "cosine") == E["metric_unknown"]
    assert exc(orc.calculate_distance, [.2, .3], [.1, .6, .3], "kl") == E["kl_not_normalised"]


@pytest.mark.parametrize("name", ["karate", "tree", "europe", "er500"])
def test_degree_signature_c_oracle(name):
    """C restatement of the degree-signature distances vs scipy / tools.metrics golden matrices."""
    D = np.load(os.path.join(GOLDEN, "dist_degree.npz"))
    if name == "er500":
        rowptr, col = D["er500/rowptr"], D["er500/col"]
    else:
        g = load_graph(name)
        rowptr, col = g.rowptr, g.col
    deg = np.diff(rowptr).astype(np.float64)
    rs = corc.RingSamples(rowptr, col, 3, deg)
    B, _ = rs.pair_rows(0, 2, corc.ALIGN_ZEROPAD)
    assert np.allclose(B, D[f"{name}/B_zeropad"], rtol=1e-12, atol=1e-12)
    A, empty = rs.pair_rows(0, 3, corc.ALIGN_NONE)
    assert empty == bool(D[f"{name}/has_empty_ring"][0])
    if not empty:
        assert np.allclose(A, D[f"{name}/A_raw"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name", ["karate", "tree"])
def test_degree_kl_c_oracle(name):
    D = np.load(os.path.join(GOLDEN, "dist_degree.npz"))
    g = load_graph(name)
    rings = orc.khop_rings(g.rowptr, g.col, 3)
    _, counts = orc.degree_histograms(g.degree, rings, 4)
    KL = corc.pair_rows_kl(counts, 0, 3, 0.5, False)
    SKL = corc.pair_rows_kl(counts, 0, 3, 0.5, True)
    G = D[f"{name}/KL_a0.5"]
    n = G.shape[0]
    iu = np.triu_indices(n, 1)
    assert np.allclose(KL[iu], G[iu], rtol=1e-12, atol=1e-14)          # upper triangle = KL(i || j)
    assert np.allclose(KL.T[iu], G[iu], rtol=1e-12, atol=1e-14)        # mirrored (model/HSD.py:132-134)
    assert np.allclose(SKL, D[f"{name}/SKL_a0.5"], rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("name", ["usa", "cora"])
def test_degree_kl_midsize_c_oracle(name):
    """tests/golden/generate_golden_kl_midsize.py: the reference's KL_divergence on sampled pairs and whole rows of mid-size graphs."""
    M = np.load(os.path.join(GOLDEN, "kl_midsize.npz"))
    g = load_graph(name)
    rings = orc.khop_rings(g.rowptr, g.col, 3)
    _, counts = orc.degree_histograms(g.degree, rings, 4)
    assert counts.shape[2] == int(M[f"{name}/bins"][0])
    rows = M[f"{name}/rows"]
    for sym, key in ((False, "rows_KL"), (True, "rows_SKL")):
        for a, i in enumerate(rows):
            got = corc.pair_rows_kl(counts, 0, 3, 0.5, sym, int(i), int(i) + 1)[0]
            assert np.allclose(got, M[f"{name}/{key}"][a], rtol=1e-12, atol=1e-14)


def test_robust_csv_known_answer():
    """tests/robust_test/robust.csv: BFS + exact heat kernel + threshold + ring sum/mean/var."""
    R = np.load(os.path.join(GOLDEN, "robust_kat.npz"))
    g = load_graph("robust")
    assert [int(x) for x in g.nodes] == [int(x) for x in R["nodes"]]
    rings = orc.khop_rings(g.rowptr, g.col, 2)
    L = g.laplacian_dense()
    for si in (0, 17, 50, 99):
        psi = orc.heat_wavelets_exact(L, R["scales"][si])
        for i in range(g.n_node):
            tr = orc.get_triple(psi, rings[i], i)
            for k in range(3):
                assert abs(tr[2 * k] - R["stats"][i, si, k, 0]) < 1e-12
                assert abs(tr[2 * k + 1] - R["stats"][i, si, k, 1]) < 1e-12
    assert R["csv_max_abs_err"][0] < 6e-9


def test_triple_and_hierarchical_degree():
    T = np.load(os.path.join(GOLDEN, "triple_karate.npz"))
    g = load_graph("karate")
    rings = orc.khop_rings(g.rowptr, g.col, 3)
    assert np.array_equal(np.asarray(orc.hierarchical_degree(rings, 3)), T["hdeg"])
    L = g.laplacian_dense()
    for si, s in enumerate(T["scales"]):
        psi = orc.heat_wavelets_exact(L, s)
        for i in range(g.n_node):
            assert np.allclose(orc.get_triple(psi, rings[i], i), T["emb"][i, si], rtol=0, atol=1e-13)


def test_weighted_graph_oracle_against_the_reference():
    """Weighted Laplacian (3-column edgelist lines; 2-column lines weigh 1): the oracle's wavelets and both distance variants against
    what the reference computed from the same file (tests/golden/generate_golden_weighted.py)."""
    import os
    import numpy as np
    from conftest import GOLDEN, load_graph
    from oracle import hsd_oracle as orc
    W = np.load(os.path.join(GOLDEN, "weighted.npz"))
    g = load_graph("karate_weighted")
    assert np.array_equal(g.laplacian_dense(), W["L"])
    rings = orc.khop_rings(g.rowptr, g.col, 3)
    for si, s in enumerate(W["scales"]):
        psi = orc.heat_wavelets_exact(g.laplacian_dense(), float(s))
        assert np.allclose(psi, W[f"psi{si}"], rtol=0, atol=1e-13)
        assert np.allclose(orc.distance_variant_a(orc.hierarchical_coefficients(psi, rings), 3), W[f"A{si}"], rtol=1e-9, atol=1e-13)
        assert np.allclose(orc.distance_variant_b_rowquirk(psi, rings, 3), W[f"B{si}"], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("name", ["PTC", "citeseer", "usa", "cora"])
def test_rings_of_the_larger_shipped_graphs(name):
    """Node order, ring members (K = 2) and ring sizes (K = 3) of the reference's larger data files (duplicate edge lines, mixed 2- / 3-column
    lines, ids not starting at 0) against tools.get_hierarchical_representation (generate_golden_rings_more.py)."""
    import os
    import numpy as np
    from conftest import GOLDEN, load_graph
    from oracle import c_oracle as corc
    from oracle import hsd_oracle as orc
    R = np.load(os.path.join(GOLDEN, "rings_more.npz"))
    g = load_graph(name)
    assert g.nodes == list(R[f"{name}/nodes"])
    assert np.array_equal(corc.khop_sizes(g.rowptr, g.col, 3), R[f"{name}/K3/sizes"])
    off = R[f"{name}/K2/offsets"]
    mem = R[f"{name}/K2/members"]
    for v in (0, 1, g.n_node // 2, g.n_node - 1):
        rings = orc.khop_rings_node(g.rowptr, g.col, v, 2)
        for k in range(3):
            assert np.array_equal(np.asarray(rings[k], dtype=np.int32), mem[off[v * 3 + k]:off[v * 3 + k + 1]])
