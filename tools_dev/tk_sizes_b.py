"""dev: the k-nearest call of the zero-pad distance by graph size: the default dispatch, the unpruned sweep (HSD_TOPK_PRUNE=0) and the pruned one
forced (=1); first call of a prepared distance and later calls.  usage: tk_sizes_b.py [n ...]   (default 40000 60000 80000)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, hsd_amd
sizes = [int(a) for a in sys.argv[1:]] or [40000, 60000, 80000]
for n in sizes:
    g = hsd_amd.barabasi_albert(n, 5, 42)
    eng = hsd_amd.Engine(0)
    eng.set_graph(g.rowptr, g.col); eng.signature_build_degree(3)
    eng.pair_prepare(0, 1, 0, 2); eng.pair_topk(20)          # (scratch of the context, rocprim's kernels: not what is compared)
    os.environ["HSD_TOPK_PRUNE"] = "1"; eng.pair_prepare(0, 1, 0, 2); eng.pair_topk(20)
    out = {}
    for pr in (None, "0", "1", None, "0", "1"):
        os.environ.pop("HSD_TOPK_PRUNE", None)
        if pr is not None: os.environ["HSD_TOPK_PRUNE"] = pr
        eng.pair_prepare(0, 1, 0, 2); eng.sync()
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); r = eng.pair_topk(20); ts.append((time.perf_counter() - t0) * 1e3)
        out[pr] = r
        print(f"n={n} HSD_TOPK_PRUNE={'(default)' if pr is None else pr}: first {ts[0]:7.2f}, later " + " ".join(f"{t:7.2f}" for t in ts[1:]) + " ms", flush=True)
    print("   identical:", all(np.array_equal(out["0"][q], out[p][q]) for p in (None, "1") for q in (0, 1)), flush=True)
    os.environ.pop("HSD_TOPK_PRUNE", None)
    eng.close()
