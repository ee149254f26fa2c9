import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/hsd_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import hsd_amd
g = hsd_amd.barabasi_albert(100000, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col); eng.signature_build_degree(3); eng.pair_prepare(0, 0, 0, 3)
os.environ["HSD_TOPK_VERBOSE"] = "1"
t0 = time.perf_counter(); eng.pair_topk(20); print("first call ms", (time.perf_counter() - t0) * 1e3)
