"""dev: how often does a k-nearest call take far longer than its median?  BA N = 50k, zero-pad distance, k = 20: `calls` pruned calls, then as many
unpruned ones (HSD_TOPK_PRUNE=0).  Question behind it: tools that re-prepare the distance in a loop saw one call in ~25 come out 20-25 ms late;
on one prepared distance none of 1 600 did (profiles/r04_tk_stall_rate.txt).  usage: tk_stall_rate.py [calls=400] [n=50000]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, hsd_amd
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col); eng.signature_build_degree(3)
for pr in ("1", "0", "1", "0"):
    os.environ["HSD_TOPK_PRUNE"] = pr
    eng.pair_prepare(0, 1, 0, 2)
    for _ in range(3): eng.pair_topk(20)
    ts = np.empty(calls)
    for c in range(calls):
        t0 = time.perf_counter(); eng.pair_topk(20); ts[c] = (time.perf_counter() - t0) * 1e3
    med = float(np.median(ts))
    late = ts[ts > 2 * med]
    print(f"HSD_TOPK_PRUNE={pr}: {calls} calls, median {med:.2f} ms, p99 {np.percentile(ts, 99):.2f}, max {ts.max():.2f}; later than twice the median: {late.size} ({', '.join(f'{x:.1f}' for x in late[:12])})", flush=True)
