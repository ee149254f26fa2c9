"""dev: the one-time setup of the pruned k-nearest sweep by stage (HSD_TOPK_VERBOSE=1).  usage: tk_setup_time.py [A|B]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hsd_amd
v = sys.argv[1] if len(sys.argv) > 1 else "A"
g = hsd_amd.barabasi_albert(100000, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col); eng.signature_build_degree(3)
os.environ["HSD_TOPK_PRUNE"] = "1"
for rep in range(2):
    eng.pair_prepare(*((0, 0, 0, 3) if v == "A" else (0, 1, 0, 2)))
    eng.sync()
    os.environ["HSD_TOPK_VERBOSE"] = "1"
    t0 = time.perf_counter(); eng.pair_topk(20); print("first call of a prepared distance, ms", (time.perf_counter() - t0) * 1e3, flush=True)
    os.environ.pop("HSD_TOPK_VERBOSE")
    t0 = time.perf_counter(); eng.pair_topk(20); print("second call, ms", (time.perf_counter() - t0) * 1e3, flush=True)
