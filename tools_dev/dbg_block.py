import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
from hsd_amd.graph import barabasi_albert
g = barabasi_albert(3001, 5, 7)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
for metric, align, lo, hi in [(0, 1, 0, 2), (0, 0, 0, 3), (1, 0, 0, 3), (2, 0, 0, 3), (0, 1, 1, 3)]:
    eng.pair_prepare(metric, align, lo, hi)
    ref = eng.pair_rows_host()
    blk = eng.pair_rows_host(1024, 2048)
    d = np.argwhere(blk != ref[1024:2048])
    print((metric, align, lo, hi), "differing", len(d), "maxabs", np.abs(blk - ref[1024:2048]).max(), d[:5].tolist(),
          "sym", np.array_equal(ref, ref.T))
    if len(d):
        i, j = d[0]
        print("   blk", blk[i, j], "ref", ref[1024 + i, j], "refT", ref[j, 1024 + i])
