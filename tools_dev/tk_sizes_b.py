import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, hsd_amd
for n in (60000, 200000):
    g = hsd_amd.barabasi_albert(n, 5, 42)
    eng = hsd_amd.Engine(0)
    eng.set_graph(g.rowptr, g.col); eng.signature_build_degree(3)
    out = {}
    for pr in ("0", "1"):
        os.environ["HSD_TOPK_PRUNE"] = pr
        eng.pair_prepare(0, 1, 0, 2); eng.sync()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); r = eng.pair_topk(20); ts.append((time.perf_counter() - t0) * 1e3)
        out[pr] = r
        print(f"n={n} HSD_TOPK_PRUNE={pr}: " + " ".join(f"{t:7.2f}" for t in ts) + " ms", flush=True)
    print("   identical:", np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1], out["1"][1]), flush=True)
    eng.close()
