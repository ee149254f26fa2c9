"""Host-side graph container: edgelist -> CSR in the reference's node order.

The reference hands `networkx.Graph` objects around (main.py:10-11) and fixes
the row/column order of every matrix to networkx insertion order, i.e. the
order in which node labels first appear in the edgelist (model/HSD.py:37,
tools/util.py:11-24).  The device path wants CSR (`rowptr` int64[N+1],
`col` int32[nnz], rows sorted ascending), so this module builds exactly that
order without needing networkx, and also accepts a `networkx.Graph` by duck
typing (`.nodes()`, `.edges(data=...)`).

Synthetic generators (Erdos-Renyi, Barabasi-Albert) are the build's own seeded
generators for the BASELINE.json configs; they are inputs, not reference code.
"""
from __future__ import annotations

import numpy as np


class Graph:
    """Undirected simple graph in CSR form.

    nodes   : list of labels, index = row/column of every matrix (model/HSD.py:37)
    rowptr  : int64[N+1]
    col     : int32[nnz]   neighbour indices, ascending within a row
    weight  : float64[nnz] edge weights (1.0 unless a 3-column edgelist gave one)
    """

    def __init__(self, nodes, rowptr, col, weight=None):
        self.nodes = list(nodes)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        if weight is None:
            weight = np.ones(self.col.shape[0], dtype=np.float64)
        self.weight = np.ascontiguousarray(weight, dtype=np.float64)
        self.n_node = len(self.nodes)
        assert self.rowptr.shape[0] == self.n_node + 1
        self.node2idx = {v: i for i, v in enumerate(self.nodes)}

    # -- construction -----------------------------------------------------
    @classmethod
    def from_edges(cls, nodes, src, dst, weight=None):
        """Build from index pairs (each undirected edge once or twice, any order).

        Duplicate edges collapse (networkx.Graph keeps the last weight) and
        self-loops are kept as a single CSR entry like networkx does.
        """
        n = len(nodes)
        src = np.asarray(src, dtype=np.int64)
        dst = np.asarray(dst, dtype=np.int64)
        if weight is None:
            w = np.ones(src.shape[0], dtype=np.float64)
        else:
            w = np.asarray(weight, dtype=np.float64)
        loops = src == dst
        a = np.concatenate([src, dst[~loops]])
        b = np.concatenate([dst, src[~loops]])
        ww = np.concatenate([w, w[~loops]])
        # later duplicates win (networkx overwrites edge data), whichever orientation they were given in: both directed copies of
        # an edge carry the position of the edge in the input; sort by (key, position), keep the last of every key
        pos = np.arange(src.shape[0])
        order = np.concatenate([pos, pos[~loops]])
        key = a * n + b
        perm = np.lexsort((order, key))
        key_s = key[perm]
        last = np.ones(key_s.shape[0], dtype=bool)
        last[:-1] = key_s[1:] != key_s[:-1]
        keep = perm[last]
        a, b, ww = a[keep], b[keep], ww[keep]          # already sorted by (a, b)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.add.at(rowptr, a + 1, 1)
        np.cumsum(rowptr, out=rowptr)
        return cls(nodes, rowptr, b.astype(np.int32), ww)

    @classmethod
    def from_networkx(cls, g):
        nodes = list(g.nodes())
        idx = {v: i for i, v in enumerate(nodes)}
        src, dst, w = [], [], []
        for u, v, d in g.edges(data=True):
            src.append(idx[u])
            dst.append(idx[v])
            w.append(float(d.get("weight", 1.0)) if isinstance(d, dict) else 1.0)
        return cls.from_edges(nodes, src, dst, w)

    @classmethod
    def coerce(cls, g):
        if isinstance(g, cls):
            return g
        if hasattr(g, "nodes") and hasattr(g, "edges"):
            return cls.from_networkx(g)
        raise TypeError("graph must be hsd_amd.Graph or a networkx.Graph")

    # -- derived ----------------------------------------------------------
    @property
    def degree(self):
        """Number of neighbours per node (unweighted), int32[N]."""
        return np.diff(self.rowptr).astype(np.int32)

    @property
    def n_edge(self):
        loops = int(np.sum(self.col == np.repeat(np.arange(self.n_node), np.diff(self.rowptr))))
        return (self.col.shape[0] + loops) // 2

    def laplacian_csr(self):
        """Combinatorial Laplacian L = D - A as CSR (rowptr, col, val) with the
        diagonal stored explicitly; D is the weighted degree, as
        networkx.laplacian_matrix does (model/HSD.py:34)."""
        n = self.n_node
        deg_w = np.zeros(n, dtype=np.float64)
        rows = np.repeat(np.arange(n), np.diff(self.rowptr))
        np.add.at(deg_w, rows, self.weight)
        r = np.concatenate([rows, np.arange(n)])
        c = np.concatenate([self.col.astype(np.int64), np.arange(n)])
        v = np.concatenate([-self.weight, deg_w])
        # merge a possible self-loop entry with the diagonal
        key = r * n + c
        perm = np.argsort(key, kind="stable")
        key, v = key[perm], v[perm]
        uniq, start = np.unique(key, return_index=True)
        val = np.add.reduceat(v, start)
        rr = uniq // n
        cc = (uniq % n).astype(np.int32)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.add.at(rowptr, rr + 1, 1)
        np.cumsum(rowptr, out=rowptr)
        return rowptr, cc, val

    def laplacian_dense(self):
        rp, c, v = self.laplacian_csr()
        n = self.n_node
        L = np.zeros((n, n), dtype=np.float64)
        L[np.repeat(np.arange(n), np.diff(rp)), c] = v
        return L

    def adjacency_dense(self):
        n = self.n_node
        A = np.zeros((n, n), dtype=np.float64)
        A[np.repeat(np.arange(n), np.diff(self.rowptr)), self.col] = self.weight
        return A


def read_edgelist(path, nodetype=str):
    """Whitespace edgelist -> Graph, node order = first appearance.

    Mirrors `nx.read_edgelist(path, create_using=nx.Graph, edgetype=float,
    data=[('weight', float)])` as called at main.py:10-11: `#` comments and
    blank lines skipped, optional third column = weight, labels kept as
    `nodetype(label)`.
    """
    idx = {}
    nodes = []
    src, dst, w = [], [], []
    seen = {}                  # weight an edge holds so far: a repeated line WITHOUT a weight column leaves it alone (networkx updates the
                               # edge's data with an empty dict), one with a weight overwrites it
    with open(path, "r", encoding="utf-8") as fin:
        for line in fin:
            p = line.find("#")
            if p >= 0:
                line = line[:p]
            s = line.split()
            if len(s) < 2:
                continue
            u, v = nodetype(s[0]), nodetype(s[1])
            for x in (u, v):
                if x not in idx:
                    idx[x] = len(nodes)
                    nodes.append(x)
            src.append(idx[u])
            dst.append(idx[v])
            key = (idx[u], idx[v]) if idx[u] <= idx[v] else (idx[v], idx[u])
            if len(s) > 2:
                seen[key] = float(s[2])
            else:
                seen.setdefault(key, 1.0)
            w.append(seen[key])
    return Graph.from_edges(nodes, src, dst, w)


# ---------------------------------------------------------------------------
# synthetic inputs for the BASELINE.json configs
# ---------------------------------------------------------------------------
def erdos_renyi(n, p, seed=42):
    """G(n, p) by geometric skipping over the strict upper triangle."""
    rng = np.random.default_rng(seed)
    total = n * (n - 1) // 2
    expect = int(total * p)
    chunks = []
    pos = -1
    while True:
        k = max(1024, int(expect * 1.1) + 1024)
        gaps = rng.geometric(p, size=k).astype(np.int64)
        lin = pos + np.cumsum(gaps)
        chunks.append(lin[lin < total])
        if lin[-1] >= total:
            break
        pos = int(lin[-1])
    lin = np.concatenate(chunks)
    # linear index -> (i, j), i < j, row-major over the strict upper triangle
    # row i starts at offset i*n - i*(i+1)/2
    i = (n - 2 - np.floor(np.sqrt(-8.0 * lin + 4.0 * n * (n - 1) - 7) / 2.0 - 0.5)).astype(np.int64)
    # fix possible off-by-one from floating point
    start = i * n - i * (i + 1) // 2
    bad = lin < start
    i[bad] -= 1
    start = i * n - i * (i + 1) // 2
    nxt = (i + 1) * n - (i + 1) * (i + 2) // 2
    bad = lin >= nxt
    i[bad] += 1
    start = i * n - i * (i + 1) // 2
    j = lin - start + i + 1
    return Graph.from_edges(list(range(n)), i, j)


def connect_components(g, seed=0):
    """Join every small component to the largest one with a single random edge (used by the
    weak-scaling bench: G(n, 10/n) has ~n e^-10 isolated nodes, whose empty rings make scipy --
    and therefore Variant A -- raise)."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components
    n = g.n_node
    A = csr_matrix((np.ones(g.col.shape[0], dtype=np.int8), g.col, g.rowptr), shape=(n, n))
    ncomp, lab = connected_components(A, directed=False)
    if ncomp == 1:
        return g
    sizes = np.bincount(lab, minlength=ncomp)
    giant = int(np.argmax(sizes))
    rng = np.random.default_rng(seed)
    giant_nodes = np.nonzero(lab == giant)[0]
    first = np.full(ncomp, -1, dtype=np.int64)
    first[lab[::-1]] = np.arange(n)[::-1]              # smallest node index of every component
    src = first[np.arange(ncomp) != giant]
    dst = giant_nodes[rng.integers(0, giant_nodes.shape[0], size=src.shape[0])]
    rows = np.repeat(np.arange(n), np.diff(g.rowptr))
    keep = rows < g.col                                # each undirected edge once
    return Graph.from_edges(g.nodes, np.concatenate([rows[keep], src]), np.concatenate([g.col[keep].astype(np.int64), dst]))


def barabasi_albert(n, m, seed=42):
    """Preferential attachment: node v >= m attaches to m distinct earlier
    nodes drawn proportionally to degree (repeated-endpoints urn)."""
    rng = np.random.default_rng(seed)
    urn = np.empty(2 * m * (n - m) + m, dtype=np.int64)
    urn[:m] = np.arange(m)          # seed nodes enter the urn once each
    fill = m
    src = np.empty(m * (n - m), dtype=np.int64)
    dst = np.empty(m * (n - m), dtype=np.int64)
    e = 0
    first = True
    for v in range(m, n):
        if first:
            targets = np.arange(m)
            first = False
        else:
            targets = np.unique(urn[rng.integers(0, fill, size=m)])
            while targets.shape[0] < m:
                extra = urn[rng.integers(0, fill, size=m)]
                targets = np.unique(np.concatenate([targets, extra]))
            if targets.shape[0] > m:
                targets = rng.permutation(targets)[:m]
        src[e:e + m] = v
        dst[e:e + m] = targets
        e += m
        urn[fill:fill + m] = targets
        urn[fill + m:fill + 2 * m] = v
        fill += 2 * m
    return Graph.from_edges(list(range(n)), src, dst)
