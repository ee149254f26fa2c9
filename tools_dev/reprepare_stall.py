"""dev (round 5): the late k-nearest call behind a re-prepare.  Round 4 saw one call in ~25 come out 20-25 ms late, always the second or third call after an
hsd_pair_prepare that had freed and re-allocated ~1 GB of operand sets.  Since round 5 the operand sets live in a grow-only arena
(csrc/context.hip, hsd_pool_*): this script re-prepares `rounds` times and times the first calls behind every prepare.
usage: reprepare_stall.py [rounds=60] [n=50000]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, hsd_amd
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col); eng.signature_build_degree(3)
eng.pair_prepare(0, 1, 0, 2); eng.pair_topk(20)
prep, calls = [], []
for r in range(rounds):
    eng.sync()
    t0 = time.perf_counter(); eng.pair_prepare(0, 1, 0, 2); eng.sync(); prep.append((time.perf_counter() - t0) * 1e3)
    row = []
    for c in range(5):
        t0 = time.perf_counter(); eng.pair_topk(20); row.append((time.perf_counter() - t0) * 1e3)
    calls.append(row)
calls = np.asarray(calls)
med = float(np.median(calls[:, 2:]))
print(f"BA N={n}, zero-pad distance, k=20: {rounds} x (hsd_pair_prepare, 5 x hsd_pair_topk)")
print(f"prepare: median {np.median(prep):.2f} ms, max {max(prep):.2f}")
for c in range(5):
    col = calls[:, c]
    print(f"call {c + 1} after a prepare: median {np.median(col):.2f} ms, p90 {np.percentile(col, 90):.2f}, max {col.max():.2f}; later than median + 10 ms: {int((col > med + 10).sum())} of {rounds}")
