"""dev (round 5): what would a different tile composition give the KL kinds' column trimming?  Config 3's graph (BA N = 50k, K = 5): per 16-node granule the
occupancy of every (layer, bin) column, tiles = 8 consecutive granules of an order, a tile pair multiplies the columns occupied on both sides.
Mean columns per tile pair for the natural order (what the kernel does) and for granules sorted by a few keys.  Needs the GPU (the histograms)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, hsd_amd
n, K = 50000, 5
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col); eng.signature_build_degree(K)
c = eng.signature_counts()[:, 1:, :] > 0                      # [n][K][B]
B = c.shape[2]
ng = n // 16
occ = c[:ng * 16].reshape(ng, 16, K * B).any(axis=1)           # [granule][col]
deg = g.degree[:ng * 16].reshape(ng, 16).mean(axis=1)


def cols_per_pair(order):
    nt = len(order) // 8
    T = occ[order[:nt * 8]].reshape(nt, 8, -1).any(axis=1).astype(np.float32)
    M = T @ T.T
    iu = np.triu_indices(nt, 1)
    per_layer = [float((T[:, k * B:(k + 1) * B] @ T[:, k * B:(k + 1) * B].T)[iu].mean()) for k in range(K)]
    return float(M[iu].mean()), per_layer


print(f"BA N={n} K={K}: {K * B} histogram columns per half")
for name, order in (("natural", np.arange(ng)),
                    ("by mean degree", np.argsort(deg, kind="stable")),
                    ("by occupied columns", np.argsort(occ.sum(axis=1), kind="stable")),
                    ("by occupied columns of ring 1", np.argsort(occ[:, :B].sum(axis=1), kind="stable")),
                    ("by occupied columns of ring 5", np.argsort(occ[:, 4 * B:].sum(axis=1), kind="stable")),
                    ("random", np.random.default_rng(0).permutation(ng))):
    tot, per = cols_per_pair(order)
    print(f"{name:32s} {tot:7.1f} columns per tile pair; per ring " + " ".join(f"{x:6.1f}" for x in per))
