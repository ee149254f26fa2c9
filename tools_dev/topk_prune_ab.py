"""dev: the pruned k-nearest sweep (raw fp64 W1 on node-permuted operands) against the unpruned one: same result, time of a warm call.
usage: topk_prune_ab.py [n=100000] [k=20] [variant=A|B]   (B: the zero-pad distance, pruned only on request)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hsd_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
MODE = (0, 1, 0, 2) if (len(sys.argv) > 3 and sys.argv[3] == "B") else (0, 0, 0, 3)
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
eng.pair_prepare(*MODE)
res = {}
cases = [("0", "1")] + [("1", g) for g in os.environ.get("AB_GROUPS", "1").split(",")]
for prune, groups in cases:
    os.environ["HSD_TOPK_PRUNE"] = prune
    os.environ["HSD_TOPK_GROUPS"] = groups
    eng.pair_prepare(*MODE)                           # (the permuted operands are rebuilt with the operands)
    ts = []
    for rep in range(3):
        if rep == 2: os.environ["HSD_TOPK_VERBOSE"] = "1"
        eng.sync(); t0 = time.perf_counter()
        idx, val = eng.pair_topk(k)
        ts.append((time.perf_counter() - t0) * 1e3)
        os.environ.pop("HSD_TOPK_VERBOSE", None)
    res[prune] = (idx, val) if prune == "0" or "1" not in res else res[prune]
    if prune == "1" and "0" in res:
        print("   identical to the unpruned result:", np.array_equal(res["0"][0], idx) and np.array_equal(res["0"][1], val), flush=True)
    print(f"HSD_TOPK_PRUNE={prune} groups={groups}: calls " + " ".join(f"{t:8.2f}" for t in ts) + " ms", flush=True)
# ground truth: a lexicographic selection on rows of the matrix (the k-nearest kernels share their early-exit logic, so equality between
# the two sweeps alone would not catch a mistake in it)
M = eng.pair_rows_device()
ok = True
for r in (0, 1, 777, 12345 % n, n // 2, n - 2, n - 1):
    d = M.rows_to_host(r, r + 1)[0].copy(); d[r] = np.inf
    order = np.lexsort((np.arange(n), d))[:k]
    for tag in res:
        ok = ok and bool(np.array_equal(order, res[tag][0][r]) and np.array_equal(d[order], res[tag][1][r]))
M.free()
print("sampled rows equal a selection on the matrix rows:", ok, flush=True)
same_i = np.array_equal(res["0"][0], res["1"][0]); same_v = np.array_equal(res["0"][1], res["1"][1])
print("identical indices:", same_i, " identical distances:", same_v, flush=True)
if not same_i:
    bad = np.argwhere((res["0"][0] != res["1"][0]).any(axis=1))[:5].ravel()
    for r in bad:
        print(" row", r, res["0"][0][r], res["1"][0][r], res["0"][1][r], res["1"][1][r])
eng.close()
