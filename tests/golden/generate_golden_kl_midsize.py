#!/usr/bin/env python3
"""Golden vectors of the KL / symmetric-KL distances on MID-SIZE shipped graphs.  RUN ONLY IN THE BUILD CONTAINER.

Round 5 changed how the build evaluates KL (ring 0 in closed form, the sparse-column form X = c / Z, Y = log(c + alpha) - log alpha with
per-tile column lists: DESIGN.md section 5): the KL goldens of rounds 1-4 stop at karate (34 nodes) and tree (127), where no tile ever
trims a column.  This script imports the reference's own tools.metrics.KL_divergence / calculate_distance (tools/metrics.py:74-88, :9-38,
the functions model/HSD.py:140-161 would call with metric "kl") and evaluates them on the build's alpha-smoothed per-ring degree
histograms for sampled pairs of

  usa    N = 1 190, connected, 144 distinct degree bins: wide support, most columns empty in most tiles
  cora   N = 2 708, 78 components (empty rings -> uniform histograms), 37 bins

For each graph, K = 3, alpha = 0.5: `pairs` (600 x 2, i != j), `KL`, `SKL`, `KLsorted` (calculate_distance's sort of both vectors first),
plus `rows` (6 whole rows of the KL and SKL MATRICES -- entry (i, j) = KL(P_min(i,j) || P_max(i,j)), the reference's mirrored upper
triangle, model/HSD.py:132-134 -- against every node: the row sums a tile-trimming error would shift).

    python tests/golden/generate_golden_kl_midsize.py
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

with contextlib.redirect_stdout(io.StringIO()):
    import tools as ref_tools                             # noqa: E402,F401
    from tools import metrics as ref_metrics              # noqa: E402

from oracle import hsd_oracle as orc                      # noqa: E402
from hsd_amd.graph import read_edgelist                   # noqa: E402


def main():
    out = {}
    gdir = os.path.join(HERE, "graphs")
    K, alpha = 3, 0.5
    rng = np.random.default_rng(20260205)
    for name in ("usa", "cora"):
        g = read_edgelist(f"{gdir}/{name}.edgelist")
        n = g.n_node
        rings = orc.khop_rings(g.rowptr, g.col, K)
        support, counts = orc.degree_histograms(g.degree, rings, K + 1)
        P = orc.smoothed_histogram(counts, alpha)                      # (n, K + 1, B), rows sum to 1
        assert np.allclose(P.sum(axis=2), 1.0, rtol=0, atol=1e-12)
        pairs = np.stack([rng.integers(0, n, 640), rng.integers(0, n, 640)], axis=1)
        pairs = pairs[pairs[:, 0] != pairs[:, 1]][:600]
        kl = np.empty(len(pairs))
        skl = np.empty(len(pairs))
        kls = np.empty(len(pairs))
        for t, (i, j) in enumerate(pairs):
            kl[t] = sum(ref_metrics.KL_divergence(P[i, k], P[j, k], False) for k in range(K + 1))
            skl[t] = sum(ref_metrics.KL_divergence(P[i, k], P[j, k], True) for k in range(K + 1))
            kls[t] = sum(ref_metrics.calculate_distance(list(P[i, k]), list(P[j, k]), "kl") for k in range(K + 1))
        rows = np.sort(rng.choice(n, size=6, replace=False))
        rkl = np.zeros((len(rows), n))
        rskl = np.zeros((len(rows), n))
        for a, i in enumerate(rows):
            for j in range(n):
                if j == i:
                    continue
                # the reference fills the upper triangle with KL(P_i || P_j), i < j, and mirrors it (model/HSD.py:132-134): row i of the
                # matrix holds KL(P_min || P_max)
                lo, hi = (i, j) if i < j else (j, i)
                rkl[a, j] = sum(ref_metrics.KL_divergence(P[lo, k], P[hi, k], False) for k in range(K + 1))
                rskl[a, j] = sum(ref_metrics.KL_divergence(P[i, k], P[j, k], True) for k in range(K + 1))
        out[f"{name}/n"] = np.asarray([n])
        out[f"{name}/bins"] = np.asarray([P.shape[2]])
        out[f"{name}/pairs"] = pairs.astype(np.int64)
        out[f"{name}/KL"] = kl
        out[f"{name}/SKL"] = skl
        out[f"{name}/KLsorted"] = kls
        out[f"{name}/rows"] = rows.astype(np.int64)
        out[f"{name}/rows_KL"] = rkl
        out[f"{name}/rows_SKL"] = rskl
        # the oracle's own KL on the same pairs
        for t in range(0, len(pairs), 25):
            i, j = pairs[t]
            od = sum(orc.kl_divergence(P[i, k], P[j, k]) for k in range(K + 1))
            assert abs(od - kl[t]) <= 1e-13 + 1e-12 * abs(kl[t]), (i, j, od, kl[t])
        print(f"{name}: n = {n}, {P.shape[2]} bins, {len(pairs)} pairs: KL sum {kl.sum():.12g}, SKL sum {skl.sum():.12g}, sorted sum {kls.sum():.12g}, "
              f"row sums {rkl.sum():.12g}")
    path = os.path.join(HERE, "kl_midsize.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
