#!/usr/bin/env python3
"""Golden vectors for WEIGHTED graphs.  RUN ONLY IN THE BUILD CONTAINER (imports /root/reference like generate_golden.py).

main.py:10-11 reads every graph with `data=[('weight', float)]`: a 3-column edgelist line carries a weight, a 2-column line
does not (networkx then uses 1), and `HSD.__init__` (model/HSD.py:33-34) takes the weighted adjacency and Laplacian.  The rings
ignore the weights (tools/hierarchy.py walks neighbours), the heat coefficients do not.  Inputs written here are our own data
(the karate edges of the reference's data file with deterministic weights on two thirds of the lines); outputs are what the
reference computes from them: thresholded exact wavelets, Variant A and Variant B (as written) distance matrices, the
recommended scale range of the weighted Laplacian, get_triple.  Every value is also compared with the oracle.

    python tests/golden/generate_golden_weighted.py
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.modules.setdefault("pygsp", types.ModuleType("pygsp"))

import networkx as nx                                     # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    import tools as ref_tools                             # noqa: E402
    from tools import util as ref_util                    # noqa: E402
    import model as ref_model                             # noqa: E402

from oracle import hsd_oracle as orc                      # noqa: E402
from hsd_amd.graph import read_edgelist                   # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def main():
    gdir = os.path.join(HERE, "graphs")
    src = os.path.join(gdir, "karate.edgelist")
    dst = os.path.join(gdir, "karate_weighted.edgelist")
    lines = []
    for k, line in enumerate(open(src)):
        u, v = line.split()[:2]
        if k % 3 == 2:
            lines.append(f"{u} {v}\n")                                        # no weight on this line: networkx uses 1
        else:
            lines.append(f"{u} {v} {0.5 + ((7 * int(u) + 13 * int(v)) % 9) / 4.0}\n")
    with open(dst, "w") as f:
        f.writelines(lines)

    G = nx.read_edgelist(dst, create_using=nx.Graph, edgetype=float, data=[("weight", float)])       # main.py:10-11
    mine = read_edgelist(dst)
    assert list(G.nodes()) == mine.nodes
    K = 3
    m = ref_model.HSD(G, "karate_weighted", 0, K, "wasserstein")
    assert np.allclose(np.asarray(m.L, dtype=float), mine.laplacian_dense(), rtol=0, atol=0), "weighted Laplacian differs"
    out = {"nodes": np.asarray(mine.nodes), "L": np.asarray(m.L, dtype=float)}
    lam = np.linalg.eigvalsh(np.asarray(m.L, dtype=float))
    out["scale_range"] = np.asarray(ref_util.recommend_scale_range(list(lam)))
    assert tuple(out["scale_range"]) == tuple(orc.recommend_scale_range(list(lam)))
    m.hierarchy = quiet(ref_tools.get_hierarchical_representation, G, K)
    rings = orc.khop_rings(mine.rowptr, mine.col, K)
    for si, s in enumerate((0.3, 1.0)):
        psi = np.asarray(m.calculate_wavelets(s, approx=False), dtype=np.float64)
        opsi = orc.heat_wavelets_exact(mine.laplacian_dense(), s)
        assert np.allclose(opsi, psi, rtol=0, atol=1e-13), np.abs(opsi - psi).max()
        out[f"psi{si}"] = psi
        DA = quiet(m.calculate_structural_distance, s, False)
        m.wavelets = psi
        n = m.n_node
        DB = np.zeros((n, n))
        rows = [ref_model.HSD._calculate_worker(m, i) for i in range(n)]      # the Pool's worker, serially (model/HSD.py:140-161)
        for i in range(n):
            for j in range(i + 1, n):
                DB[i, j] = DB[j, i] = rows[i][j]
        coeffs = orc.hierarchical_coefficients(opsi, rings)
        assert np.allclose(orc.distance_variant_a(coeffs, K), DA, rtol=1e-9, atol=1e-13)
        assert np.allclose(orc.distance_variant_b_rowquirk(opsi, rings, K), DB, rtol=1e-9, atol=1e-13)
        out[f"A{si}"], out[f"B{si}"] = DA, DB
        # MultiHSD.get_triple on the same wavelets (model/multiscale_HSD.py:45-61): an instance without PyGSP's __init__
        mh = ref_model.MultiHSD.__new__(ref_model.MultiHSD)
        ref_model.HSD.__init__(mh, G, "karate_weighted", 0, K, "euclidean")
        mh.hierarchy = m.hierarchy
        out[f"triple{si}"] = np.asarray([mh.get_triple(psi, v) for v in mine.nodes], dtype=np.float64)
        print(f"s = {s}: sum A {DA.sum()!r}, sum B {DB.sum()!r}")
    out["scales"] = np.asarray([0.3, 1.0])
    np.savez_compressed(os.path.join(HERE, "weighted.npz"), **out)
    print("weighted.npz written")


if __name__ == "__main__":
    main()
