"""dev: the pruned k-nearest sweep (BA N = 100k, raw fp64 W1, k = 20) by phase count and width of the first launch's band.  usage: tk_phases.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, hsd_amd
g = hsd_amd.barabasi_albert(100000, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col); eng.signature_build_degree(3); eng.pair_prepare(0, 0, 0, 3)
ref = eng.pair_topk(20)
for band in ("1", "2", "3"):
    os.environ["HSD_TOPK_BAND"] = band
    eng.pair_prepare(0, 0, 0, 3)
    for ph in ("4", "5", "8"):
        os.environ["HSD_TOPK_PHASES"] = ph
        eng.pair_topk(20)
        ts = []
        for _ in range(3):
            eng.sync(); t0 = time.perf_counter(); r = eng.pair_topk(20); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"band {band} phases {ph:>2}: " + " ".join(f"{t:7.2f}" for t in ts) + f" ms  same: {np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1])}", flush=True)
