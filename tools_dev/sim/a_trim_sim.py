"""dev (round 5): would Variant A (raw fp64 W1, columns G = CDF x bin width) gain from trimming a layer's columns at the largest occupied bin of a tile pair?
Above a node's largest member the CDF is 1 and G equals the bin width for every node: |G_i - G_j| = 0 exactly.  BA N = 100k, K = 3: per layer the mean
number of columns a tile pair needs (max over its 256 nodes of the largest occupied bin + 1), for the natural granule order and for granules sorted by the key."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, hsd_amd
n, K = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 3
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col); eng.signature_build_degree(K)
c = eng.signature_counts()                                     # [n][K+1][B]
B = c.shape[2]
ng = n // 16
top = np.zeros((K + 1, n), dtype=np.int32)
for k in range(1, K + 1):
    occ = c[:, k, :] > 0
    top[k] = np.where(occ.any(axis=1), B - np.argmax(occ[:, ::-1], axis=1), 0)     # largest occupied bin + 1
gtop = top[:, :ng * 16].reshape(K + 1, ng, 16).max(axis=2)
print(f"BA N={n} K={K}: {B} bins per layer")
for k in range(1, K + 1):
    print(f"layer {k}: per node mean top bin {top[k].mean():.1f}, per granule {gtop[k].mean():.1f}")
    for name, order in (("natural", np.arange(ng)), ("sorted by this layer's key", np.argsort(gtop[k], kind="stable")), ("sorted by layer 1's key", np.argsort(gtop[1], kind="stable"))):
        nt = ng // 8
        t = gtop[k][order[:nt * 8]].reshape(nt, 8).max(axis=1)
        s = np.sort(t)
        # mean over unordered tile pairs of max(t_a, t_b): element s[r] is the max for r pairs (+ itself)
        r = np.arange(nt)
        mean_pair = float((s * (r + 1)).sum() / (nt * (nt + 1) / 2))
        print(f"    {name:28s} mean columns per tile pair {mean_pair:6.1f} of {B - 1}")
