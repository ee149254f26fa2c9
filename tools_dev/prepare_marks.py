"""hsd_pair_prepare's own phase timings (HSD_PLAN_VERBOSE=1: a stream synchronisation and a line on stderr per phase), third warm call, Variant B at the target.
usage: python tools_dev/prepare_marks.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
for rep in range(3):
    eng.sync(); t0 = time.perf_counter()
    eng.pair_prepare(0, 1, 0, 2); eng.sync()
    print(f"prepare call {rep}: {(time.perf_counter() - t0) * 1e3:.3f} ms", file=sys.stderr, flush=True)
os.environ["HSD_PLAN_VERBOSE"] = "1"
eng.pair_prepare(0, 1, 0, 2); eng.sync()
os.environ.pop("HSD_PLAN_VERBOSE")
for thr in ("1", "4", "8", "16", "32"):
    os.environ["HSD_PLAN_THREADS"] = thr
    ts = []
    for rep in range(4):
        eng.sync(); t0 = time.perf_counter()
        eng.pair_prepare(0, 1, 0, 2); eng.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"HSD_PLAN_THREADS={thr}: " + " ".join(f"{t:.3f}" for t in ts), file=sys.stderr, flush=True)
eng.close()
