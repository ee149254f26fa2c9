"""dev: the pruned k-nearest sweep with its node order from the device (default) and from host threads (HSD_TOPK_ORDER=0): setup by stage
(HSD_TOPK_VERBOSE=1), first call of a prepared distance, later calls, tile pairs kept.  usage: tk_order_ab.py [n=100000] [A|B ...]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, hsd_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
variants = sys.argv[2:] or ["A", "B"]
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col); eng.signature_build_degree(3)
for v in variants:
    mode = (0, 0, 0, 3) if v == "A" else (0, 1, 0, 2)
    os.environ["HSD_TOPK_PRUNE"] = "0"
    eng.pair_prepare(*mode); eng.sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ref = eng.pair_topk(20); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"n={n} {v} unpruned: " + " ".join(f"{t:7.2f}" for t in ts) + " ms", flush=True)
    os.environ["HSD_TOPK_PRUNE"] = "1"
    for order in ("0", "1", "0", "1"):
        os.environ["HSD_TOPK_ORDER"] = order
        eng.pair_prepare(*mode); eng.sync()
        os.environ["HSD_TOPK_VERBOSE"] = "1"
        t0 = time.perf_counter(); r = eng.pair_topk(20); first = (time.perf_counter() - t0) * 1e3
        os.environ.pop("HSD_TOPK_VERBOSE")
        eng.pair_prepare(*mode); eng.sync()
        t0 = time.perf_counter(); r = eng.pair_topk(20); first_quiet = (time.perf_counter() - t0) * 1e3
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); r = eng.pair_topk(20); ts.append((time.perf_counter() - t0) * 1e3)
        same = np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1])
        print(f"n={n} {v} HSD_TOPK_ORDER={order}: first call {first_quiet:7.2f} ms (with the stage log {first:7.2f}), later " + " ".join(f"{t:7.2f}" for t in ts) + f" ms, identical to unpruned: {same}", flush=True)
