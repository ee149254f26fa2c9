"""Zero-pad layer forms at BA N = 100k (Variant B, ring 2): expected operand words per tile pair for one global
threshold (what hsd_pair_prepare stores today) against several stored thresholds with a per-tile-pair choice.
CPU only (scipy.sparse)."""
import sys, os, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsd_amd.graph import barabasi_albert
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = barabasi_albert(n, 5, 42)
deg = g.degree.astype(np.int64)
A = sp.csr_matrix((np.ones(len(g.col), dtype=np.int32), g.col, g.rowptr), shape=(n, n))
t0 = time.time()
A2 = (A @ A).tocsr(); A2.data[:] = 1
R2 = A2 - A2.multiply(A) ; R2.setdiag(0); R2.eliminate_zeros()           # ring 2 = reached in two steps, not in ring 0/1
print("ring-2 members:", R2.nnz, "mean", R2.nnz / n, f"({time.time() - t0:.1f} s)")
support = np.unique(deg)
B = len(support)
print("bins", B, "max degree", support[-1])
def words_for(ring, name):
    # candidate thresholds: every bin boundary with an even number of bins below it
    tile = 128
    nt = (n + tile - 1) // tile
    sizes = np.asarray(ring.sum(axis=1)).ravel()
    cands = [0] + [int(support[k]) for k in range(2, B, 2)]
    low_words, tmax = [], []
    for th in cands:
        above = np.asarray(ring @ (deg > th).astype(np.int32)).ravel() if th > 0 else sizes
        m = np.zeros(nt * tile, dtype=np.int64); m[:n] = above
        tmax.append(m.reshape(nt, tile).max(axis=1))
        low_words.append((np.searchsorted(support, th, side='right') + 1) // 2 if th > 0 else 0)
    tmax = np.array(tmax); low_words = np.array(low_words)
    bin_words = (B + 1) // 2
    iu = np.triu_indices(nt)
    def cost(sel):
        # per tile pair: min over the selected thresholds (and the bin form) of low + ceil(max(R_i, R_j) / 2)
        best = np.full(len(iu[0]), bin_words, dtype=np.int64)
        for k in sel:
            r = np.maximum(tmax[k][iu[0]], tmax[k][iu[1]])
            best = np.minimum(best, low_words[k] + (r + 1) // 2)
        return best.mean()
    single = [(cost([k]), k) for k in range(len(cands))]
    c1, k1 = min(single)
    print(f"{name}: bin form {bin_words} words; best single threshold {cands[k1]} -> {c1:.1f} words")
    # greedy selection of further thresholds
    sel = [k1]
    for extra in range(3):
        c, k = min((cost(sel + [k]), k) for k in range(len(cands)) if k not in sel)
        sel.append(k)
        print(f"   + threshold {cands[k]:5d}: {c:.1f} words  (thresholds {[cands[x] for x in sel]})")
    print(f"   every threshold available: {cost(range(len(cands))):.1f} words")
words_for(R2, "ring 2")
R1 = A.copy(); R1.data[:] = 1
words_for(R1, "ring 1")

# ---- what would a node order with homogeneous tiles buy?  (granules of 16 nodes keep whole 128-B lines of D) ----
def reorder_sim(ring, name):
    tile = 128
    nt = (n + tile - 1) // tile
    sizes = np.asarray(ring.sum(axis=1)).ravel()
    cands = [0] + [int(support[k]) for k in range(2, B, 2)]
    bin_words = (B + 1) // 2
    iu = np.triu_indices(nt)
    for label, gran in (("generation order", None), ("granules of 16 sorted", 16), ("granules of 64 sorted", 64), ("nodes sorted", 1)):
        best = None
        for th in cands[:80]:
            above = np.asarray(ring @ (deg > th).astype(np.int32)).ravel() if th > 0 else sizes
            m = np.zeros(nt * tile, dtype=np.int64); m[:n] = above
            if gran:
                gm = m.reshape(-1, gran)
                m = gm[np.argsort(gm.max(axis=1), kind='stable')].ravel()
            tm = m.reshape(nt, tile).max(axis=1)
            low = (np.searchsorted(support, th, side='right') + 1) // 2 if th > 0 else 0
            c = np.minimum(bin_words, low + (np.maximum(tm[iu[0]], tm[iu[1]]) + 1) // 2).mean()
            if best is None or c < best[0]: best = (c, th)
        print(f"{name} {label:24s}: {best[0]:.1f} words at threshold {best[1]}")
reorder_sim(R2, "ring 2")
reorder_sim(R1, "ring 1")
