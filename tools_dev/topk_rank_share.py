"""dev (round 5): every rank's k-nearest rows of `--gpus P` -- hsd_pair_topk(k, r0, r1) over the rank's row block -- timed one after the other on ONE
GPU: pruned sweep of a row block (default where it applies) against the unpruned one (HSD_TOPK_PRUNE=0), lists compared bit for bit.  What the
slowest of P identical GPUs would take before any launch skew; not a multi-GPU measurement.
usage: topk_rank_share.py [n=200000] [variants=B] [P=8] [k=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hsd_amd
from hsd_amd.distributed import shard_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
variants = sys.argv[2] if len(sys.argv) > 2 else "B"
P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
k = int(sys.argv[4]) if len(sys.argv) > 4 else 20
g = hsd_amd.barabasi_albert(n, 5, seed=42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)


def timed(r0, r1, reps):
    eng.sync()
    t0 = time.perf_counter()
    first = eng.pair_topk(k, r0, r1)
    t_first = (time.perf_counter() - t0) * 1e3
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = eng.pair_topk(k, r0, r1)
        ts.append((time.perf_counter() - t0) * 1e3)
    return out, t_first, min(ts), sum(ts) / len(ts)


for name in variants:
    mode = (0, 1, 0, 2) if name == "B" else (0, 0, 0, 3)
    eng.pair_prepare(*mode)
    print(f"# BA N={n} m=5 seed=42, K=3, variant {name}, k={k}, P={P}: per-rank hsd_pair_topk(k, r0, r1) on one GPU, end to end (host outputs), ms")
    print(f"# {'rank':>4s} {'rows':>17s} | {'pruned first':>12s} {'best':>8s} {'mean':>8s} | {'unpruned best':>13s} {'mean':>8s} | identical")
    worst_p, worst_u = 0.0, 0.0
    for r in range(P):
        r0, r1 = shard_rows(n, P, r)
        if r1 <= r0:
            continue
        reps = 3 if name == "B" else 2
        os.environ.pop("HSD_TOPK_PRUNE", None)
        (pi, pv), pf, pb, pm = timed(r0, r1, reps)
        os.environ["HSD_TOPK_PRUNE"] = "0"
        (ui, uv), uf, ub, um = timed(r0, r1, max(1, reps - 1))
        os.environ.pop("HSD_TOPK_PRUNE", None)
        same = bool(np.array_equal(pi, ui) and np.array_equal(pv, uv))
        worst_p, worst_u = max(worst_p, pb), max(worst_u, ub)
        print(f"  {r:4d} {r0:8d}..{r1:<8d} | {pf:12.2f} {pb:8.2f} {pm:8.2f} | {ub:13.2f} {um:8.2f} | {same}", flush=True)
    print(f"# slowest rank: pruned {worst_p:.2f} ms, unpruned {worst_u:.2f} ms")
eng.close()
