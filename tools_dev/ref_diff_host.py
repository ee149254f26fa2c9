"""dev, BUILD CONTAINER ONLY (imports /root/reference): randomized differential run of the host-only helpers against the reference's own
functions -- `.layers` reader, label reader, embeddings CSV writer / reader, distance edgelist writer, scale range, edge filter -- results AND
exceptions must agree; and the SIR labels' host arithmetic (hsd_amd/sir.py: ring sizes x decay weights, the class cut) against tools/SIR.py's per-origin
infection loop on random graphs, with the ring sizes of this tool taken from the CPU oracle in place of hsd_khop_sizes (the GPU test uses the real call).  The graph container is compared with networkx itself in tests/test_graph_cpu.py.  usage: python3 tools_dev/ref_diff_host.py [seed]"""
import contextlib
import io
import os
import random
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("pygsp", types.ModuleType("pygsp"))
np.float = float                                    # the reference predates numpy 1.24

with contextlib.redirect_stdout(io.StringIO()):
    from tools import SIR as r_sir, dataloader as r_dl, hierarchy as r_h, rw as r_rw, util as r_util      # noqa: E402
from hsd_amd import driver, formats, sir                                                     # noqa: E402
from oracle import hsd_oracle as orc                                                         # noqa: E402  (a dev tool may)


class OracleSizes:
    """Stands in for the Engine in THIS TOOL ONLY: ring sizes from the CPU oracle, so that sir.py's arithmetic can be compared without a GPU."""
    def set_graph(self, rowptr, col):
        self.rowptr, self.col = rowptr, col

    def khop_sizes(self, hop):
        return np.array([[len(r) for r in row] for row in orc.khop_rings(self.rowptr, self.col, hop)], dtype=np.int32)


def outcome(fn, *a):
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        try:
            return ("ok", fn(*a))
        except Exception as e:                       # noqa: BLE001
            return ("exc", type(e).__name__)


def main(seed):
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    bad = []
    with tempfile.TemporaryDirectory() as d:
        for _ in range(400):                         # .layers text: ragged rings, empty rings, missing trailing '#', blank lines
            lines = []
            for i in range(rng.randint(1, 6)):
                rings = [str(i)] + [",".join(str(rng.randint(0, 9)) for _ in range(rng.randint(0, 3))) for _ in range(rng.randint(0, 4))]
                lines.append("#".join(rings) + ("#" if rng.random() < 0.8 else ""))
            if rng.random() < 0.2:
                lines.insert(rng.randint(0, len(lines)), "")
            p = os.path.join(d, "h.layers")
            open(p, "w").write("\n".join(lines) + ("\n" if rng.random() < 0.8 else ""))
            hop = rng.randint(0, 5)
            if outcome(r_h.read_hierarchy, p, hop) != outcome(formats.read_hierarchy, p, hop):
                bad.append(("read_hierarchy", open(p).read(), hop))
        for _ in range(400):                         # label files: double blanks, non-integers, blank lines in the middle
            lines = []
            for _i in range(rng.randint(0, 6)):
                c = rng.random()
                lines.append(f"{rng.randint(0, 20)} {rng.randint(0, 5)}" if c < 0.7 else f"{rng.randint(0, 20)}  {rng.randint(0, 5)}" if c < 0.8 else "x y" if c < 0.9 else "")
            p = os.path.join(d, "l.label")
            open(p, "w").write("\n".join(lines) + "\n")
            if outcome(r_dl.read_label, p) != outcome(formats.read_label, p):
                bad.append(("read_label", lines))
        for _ in range(200):                         # embeddings CSV and distance edgelist: text identical, readers agree
            n, dim = rng.randint(1, 5), rng.randint(1, 4)
            nodes = [str(rng.randint(0, 99)) for _ in range(n)]
            vecs = [[rng.choice([rng.uniform(-1e3, 1e3), rng.uniform(-1e-9, 1e-9), float(rng.randint(-3, 3)), 1e20, float("nan")]) for _ in range(dim)] for _ in range(n)]
            p1, p2 = os.path.join(d, "a.csv"), os.path.join(d, "b.csv")
            r_rw.save_vectors(nodes, vecs, p1)
            formats.save_vectors(nodes, vecs, p2)
            if open(p1).read() != open(p2).read():
                bad.append(("save_vectors", open(p1).read(), open(p2).read()))
            D = np.array([[rng.uniform(0, 5) for _ in range(n)] for _ in range(n)])
            names = [str(i) for i in range(n)]
            r_rw.save_distance_edgelist(p1, names, D)
            formats.save_distance_edgelist(p2, names, D)
            if open(p1).read() != open(p2).read():
                bad.append(("save_distance_edgelist",))
            a, b = outcome(r_rw.read_distance, p1, n), outcome(formats.read_distance, p1, n)
            if a[0] != b[0] or (a[0] == "ok" and not np.array_equal(a[1], b[1])):
                bad.append(("read_distance",))
        for _ in range(500):                         # scale range (zero and tiny eigenvalues), edge filter with tied distances
            n = int(nrng.integers(2, 30))
            lam = list(np.abs(nrng.normal(size=n)) * nrng.choice([1e-4, 1, 10]))
            if nrng.random() < 0.5:
                lam[0] = 0.0
            a, b = outcome(r_util.recommend_scale_range, list(lam)), outcome(driver.recommend_scale_range, list(lam))
            if a[0] != b[0] or (a[0] == "ok" and not (a[1][0] == b[1][0] and a[1][1] == b[1][1])):
                bad.append(("recommend_scale_range", lam))
            D = np.round(nrng.random((n, n)) * 6) / 6
            D = np.maximum(D, D.T)
            np.fill_diagonal(D, 0)
            names = [f"n{i}" for i in range(n)]
            ratio = float(nrng.choice([0.05, 0.3, 1.0]))
            sp1, sp2 = os.path.join(d, "k1"), os.path.join(d, "k2")
            a, b = r_util.filter_distance_matrix(D, names, sp1, ratio), driver.filter_distance_matrix(D, names, sp2, ratio)
            if [tuple(x) for x in a] != [tuple(x) for x in b] or open(sp1).read() != open(sp2).read():
                bad.append(("filter_distance_matrix", n, ratio))
    import networkx as nx
    for _ in range(150):                             # SIR influence and labels
        n = int(nrng.integers(5, 60))
        G = nx.gnp_random_graph(n, float(nrng.choice([1.5 / n, 3 / n, 0.3])), seed=int(nrng.integers(1 << 30)))
        G = nx.relabel_nodes(G, {i: str(i) for i in G})
        if G.number_of_edges() == 0:
            continue
        t, ncls = int(nrng.integers(1, 8)), int(nrng.integers(2, 6))
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            ref = r_sir.SIR(G, 1.0, t)
            ref.start()
        mine = sir.SIR(G, 1.0, t, engine=OracleSizes())
        mine.start()
        if ref.label_nodes(ncls) != mine.label_nodes(ncls) or any(abs(mine.influence[v] - ref.influence[v]) > 1e-12 * max(1.0, ref.influence[v]) for v in G):
            bad.append(("SIR", n, t, ncls))
    for b in bad[:10]:
        print("MISMATCH", b)
    print(f"ref_diff_host: seed {seed}, {len(bad)} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 1))
