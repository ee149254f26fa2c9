"""CPU: the host-side graph container (hsd_amd/graph.py) -- the CSR the device path is fed with must be the graph networkx.Graph would hold for
the same edgelist (main.py:10-11, model/HSD.py:33-37): node order = first appearance, duplicates collapse with the LAST weight, a self-loop is one
entry, the Laplacian is D - A on weighted degrees -- and the seeded generators of the BASELINE.json configs are deterministic simple graphs."""
import numpy as np
import pytest

from hsd_amd.graph import Graph, barabasi_albert, connect_components, erdos_renyi, read_edgelist


def _is_simple_symmetric(g):
    n = g.n_node
    rows = np.repeat(np.arange(n), np.diff(g.rowptr))
    assert g.rowptr[0] == 0 and g.rowptr[-1] == g.col.shape[0] and np.all(np.diff(g.rowptr) >= 0)
    key = rows * n + g.col
    assert np.all(np.diff(key) > 0)                               # ascending within a row, no duplicate
    assert np.array_equal(np.sort(g.col.astype(np.int64) * n + rows), key)          # every (i, j) has its (j, i)


def test_read_edgelist_is_what_networkx_would_hold(tmp_path):
    p = tmp_path / "g.edgelist"
    p.write_text("# a comment\n"
                 "b a 2.5\n"
                 "\n"
                 "a c\n"
                 "c b 4   # trailing comment\n"
                 "a b 7\n"                                      # the same edge again, other orientation: the last weight wins
                 "d d 3\n"                                      # a self-loop
                 "lonely\n")                                    # fewer than two columns: skipped
    g = read_edgelist(str(p))
    assert g.nodes == ["b", "a", "c", "d"]                       # first appearance
    _is_simple_symmetric(g)
    A = g.adjacency_dense()
    want = np.zeros((4, 4))
    want[0, 1] = want[1, 0] = 7.0
    want[1, 2] = want[2, 1] = 1.0
    want[2, 0] = want[0, 2] = 4.0
    want[3, 3] = 3.0
    assert np.array_equal(A, want)
    assert g.n_edge == 4 and list(g.degree) == [2, 2, 2, 1]
    # L = D - A with the weighted degree on the diagonal (a self-loop adds its weight to D and subtracts it as A: networkx.laplacian_matrix)
    L = g.laplacian_dense()
    assert np.allclose(L, np.diag(A.sum(1)) - A) and np.allclose(L.sum(1), 0.0)
    ints = read_edgelist(str(tmp_path / "g.edgelist"), nodetype=str)
    assert ints.node2idx == {"b": 0, "a": 1, "c": 2, "d": 3}


def test_from_edges_orientation_and_duplicates():
    g = Graph.from_edges(list("xyz"), [0, 1, 2, 1], [1, 0, 1, 2], [1.0, 2.0, 3.0, 5.0])
    _is_simple_symmetric(g)
    assert g.col.tolist() == [1, 0, 2, 1] and g.weight.tolist() == [2.0, 2.0, 5.0, 5.0]
    with pytest.raises(TypeError):
        Graph.coerce([(0, 1)])
    assert Graph.coerce(g) is g


def test_networkx_duck_typing():
    class FakeNx:                                                   # what Graph.from_networkx reads of a networkx.Graph
        def nodes(self):
            return ["2", "1", "3"]

        def edges(self, data=False):
            return [("2", "1", {}), ("1", "3", {"weight": 0.5})]
    g = Graph.coerce(FakeNx())
    assert g.nodes == ["2", "1", "3"] and g.adjacency_dense()[1, 2] == 0.5 and g.adjacency_dense()[0, 1] == 1.0


@pytest.mark.parametrize("n,m", [(50, 1), (2000, 5)])
def test_barabasi_albert_is_seeded_and_simple(n, m):
    g = barabasi_albert(n, m, 7)
    h = barabasi_albert(n, m, 7)
    assert np.array_equal(g.rowptr, h.rowptr) and np.array_equal(g.col, h.col)
    assert not np.array_equal(g.col, barabasi_albert(n, m, 8).col)
    _is_simple_symmetric(g)
    assert g.n_edge == m * (n - m)                                 # every node from the m-th on brings m distinct earlier neighbours
    rows = np.repeat(np.arange(n), np.diff(g.rowptr))
    earlier = np.bincount(rows[g.col < rows], minlength=n)
    assert np.all(earlier[m:] == m) and np.all(earlier[:m] == 0)
    assert not np.any(rows == g.col)


def test_erdos_renyi_density_and_repair():
    n, p = 3000, 10.0 / 3000
    g = erdos_renyi(n, p, seed=3)
    _is_simple_symmetric(g)
    assert np.array_equal(g.col, erdos_renyi(n, p, seed=3).col)
    rows = np.repeat(np.arange(n), np.diff(g.rowptr))
    assert not np.any(rows == g.col)
    expect = p * n * (n - 1) / 2
    assert abs(g.n_edge - expect) < 5 * np.sqrt(expect)
    # every small component gets one edge to the giant one; nothing else changes
    c = connect_components(g, 0)
    _is_simple_symmetric(c)
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components
    before, lab = connected_components(csr_matrix((np.ones(g.col.shape[0]), g.col, g.rowptr), shape=(n, n)), directed=False)
    after, _ = connected_components(csr_matrix((np.ones(c.col.shape[0]), c.col, c.rowptr), shape=(n, n)), directed=False)
    assert after == 1 and c.n_edge == g.n_edge + before - 1
    assert connect_components(c, 0) is c


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_against_networkx_itself(tmp_path, seed):
    """The same edgelist file through networkx as main.py:10-11 reads it (duplicates in both orientations, self-loops, weights, comments):
    node order, adjacency and Laplacian (model/HSD.py:33-37) must agree entry by entry."""
    nx = pytest.importorskip("networkx")
    rng = np.random.default_rng(seed)
    n, m = 40, 220
    lines = ["# header"]
    for _ in range(m):
        u, v = (int(x) for x in rng.integers(0, n, 2))
        if rng.random() < 0.5:
            lines.append(f"n{u} n{v} {rng.integers(1, 9) / 4}")
        else:
            lines.append(f"n{u} n{v}")
    p = tmp_path / "r.edgelist"
    p.write_text("\n".join(lines) + "\n")
    ref = nx.read_edgelist(str(p), create_using=nx.Graph, edgetype=float, data=[("weight", float)])
    g = read_edgelist(str(p))
    assert g.nodes == list(ref.nodes())
    assert np.array_equal(g.adjacency_dense(), nx.to_numpy_array(ref, nodelist=g.nodes, weight="weight"))
    assert np.allclose(g.laplacian_dense(), nx.laplacian_matrix(ref, nodelist=g.nodes).toarray(), rtol=0, atol=1e-12)
    assert g.n_edge == ref.number_of_edges()
    h = Graph.coerce(ref)
    assert h.nodes == g.nodes and np.array_equal(h.col, g.col) and np.array_equal(h.weight, g.weight)
