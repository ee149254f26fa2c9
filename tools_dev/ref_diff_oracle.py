"""dev, BUILD CONTAINER ONLY (imports /root/reference): randomized differential run of the CPU ORACLE (oracle/hsd_oracle.py) against the reference
itself on random small graphs -- trees, sparse / dense random graphs, disconnected ones (empty rings), weighted edges, self-loops:
rings (tools/hierarchy.py:16-38), exact wavelets with the threshold (model/HSD.py:61-66), hierarchical degree, Variant A (model/HSD.py:98-114)
and Variant B as written (:118-161), on heat coefficients and on degree samples; exceptions must agree too (scipy refuses an empty ring).
The committed goldens pin the oracle on the reference's own data files; this widens the pin to graph shapes those files do not have.
usage: python3 tools_dev/ref_diff_oracle.py [seed=1] [graphs=60]"""
import contextlib
import io
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("pygsp", types.ModuleType("pygsp"))
np.float = float

import networkx as nx                                   # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    import tools as ref_tools                           # noqa: E402
    import model as ref_model                           # noqa: E402
from oracle import hsd_oracle as orc                    # noqa: E402
from hsd_amd.graph import Graph                         # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()), np.errstate(all="ignore"):
        try:
            return ("ok", fn(*a, **k))
        except Exception as e:                          # noqa: BLE001
            return ("exc", type(e).__name__)


def random_graph(rng):
    n = int(rng.integers(4, 28))
    kind = rng.choice(["tree", "sparse", "dense", "pieces", "star"])
    G = nx.Graph()
    names = [f"v{i}" for i in rng.permutation(n)]
    G.add_nodes_from(names[: int(rng.integers(0, 3))])             # (some nodes known before any edge: insertion order is the matrix order)
    def add(u, v):
        if rng.random() < 0.4:
            G.add_edge(names[u], names[v], weight=float(rng.integers(1, 9)) / 4)
        else:
            G.add_edge(names[u], names[v])
    if kind == "tree":
        for v in range(1, n):
            add(int(rng.integers(0, v)), v)
    elif kind == "star":
        for v in range(1, n):
            add(0, v)
        add(1, 2)
    else:
        p = {"sparse": 1.5 / n, "dense": 0.4, "pieces": 0.8 / n}[kind]
        for u in range(n):
            for v in range(u + 1, n):
                if rng.random() < p:
                    add(u, v)
        if kind != "pieces":
            for v in range(1, n):                                   # connect
                if not any(G.has_edge(names[v], names[u]) for u in range(v)) and names[v] not in G:
                    add(int(rng.integers(0, v)), v)
        for v in range(n):
            if names[v] not in G:
                if kind == "pieces" and rng.random() < 0.5:
                    G.add_node(names[v])                            # an isolated node
                else:
                    add(int(rng.integers(0, n)), v)
    if rng.random() < 0.15:
        u = names[int(rng.integers(0, n))]
        if u in G:
            G.add_edge(u, u)                                        # a self-loop
    return G


def main(seed, count):
    rng = np.random.default_rng(seed)
    bad = []
    done = 0
    seen = {"variant A refused": 0, "variant B refused": 0, "weighted": 0, "disconnected": 0, "self-loop": 0}
    for t in range(count):
        G = random_graph(rng)
        if G.number_of_nodes() < 2:
            continue
        K = int(rng.integers(1, 5))
        scale = float(rng.choice([0.1, 1.0, 5.0]))
        mine = Graph.coerce(G)
        tag = (t, G.number_of_nodes(), G.number_of_edges(), K, scale)
        assert mine.nodes == list(G.nodes())
        # rings
        hier = quiet(ref_tools.get_hierarchical_representation, G, K)[1]
        rings = orc.khop_rings(mine.rowptr, mine.col, K)
        for i, v in enumerate(mine.nodes):
            for k in range(K + 1):
                if sorted(mine.node2idx[u] for u in hier[v][k]) != list(rings[i][k]):
                    bad.append(("rings", tag, v, k))
        m = ref_model.HSD(G, "g", scale, K, "wasserstein")
        m.hierarchy = hier
        # wavelets (exact branch + threshold), hierarchical degree
        psi_ref = np.asarray(m.calculate_wavelets(scale, approx=False))
        psi = orc.threshold_wavelets(orc.heat_wavelets_exact(mine.laplacian_dense(), scale), mine.n_node)
        if not np.allclose(psi, psi_ref, rtol=1e-9, atol=1e-13):
            bad.append(("wavelets", tag, float(np.abs(psi - psi_ref).max())))
        hd = quiet(m.get_nodes_hierarchical_degree)
        od = orc.hierarchical_degree(rings, K)
        if hd[0] == "ok" and any(list(hd[1][v]) != list(od[i]) for i, v in enumerate(mine.nodes)):
            bad.append(("hdeg", tag))
        # Variant A on heat coefficients: values or the same refusal
        a = quiet(m.calculate_structural_distance, scale, False)
        b = quiet(lambda: orc.distance_variant_a(orc.hierarchical_coefficients(psi, rings), K))
        if a[0] != b[0] or (a[0] == "ok" and not np.allclose(a[1], b[1], rtol=1e-9, atol=1e-13)) or (a[0] == "exc" and a[1] != b[1]):
            bad.append(("variant A", tag, a if a[0] == "exc" else "values", b if b[0] == "exc" else "values"))
        seen["variant A refused"] += a[0] == "exc"
        seen["weighted"] += any("weight" in d for _, _, d in G.edges(data=True))
        seen["disconnected"] += not nx.is_connected(G)
        seen["self-loop"] += nx.number_of_selfloops(G) > 0
        # Variant B as written (row quirk), serial worker
        m.wavelets = psi_ref
        n = mine.n_node
        rows = quiet(lambda: [ref_model.HSD._calculate_worker(m, i) for i in range(n)])
        ob = quiet(orc.distance_variant_b_rowquirk, psi, rings, K)
        seen["variant B refused"] += rows[0] == "exc"
        if rows[0] != ob[0]:
            bad.append(("variant B outcome", tag, rows[:2] if rows[0] == "exc" else "ok", ob[:2] if ob[0] == "exc" else "ok"))
        elif rows[0] == "ok":
            DB = np.zeros((n, n))
            for i in range(n):
                for j in range(i + 1, n):
                    DB[i, j] = DB[j, i] = rows[1][i][j]
            if not np.allclose(DB, ob[1], rtol=1e-9, atol=1e-13):
                bad.append(("variant B", tag, float(np.abs(DB - ob[1]).max())))
        done += 1
    for b in bad[:12]:
        print("MISMATCH", b)
    print(f"ref_diff_oracle: seed {seed}, {done} graphs ({seen}), {len(bad)} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 60))
