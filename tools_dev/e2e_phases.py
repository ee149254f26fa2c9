"""End-to-end phases for a SEQUENCE of graphs through one context (what a service that keeps the context pays per graph): set_graph, k-hop sweep + histograms,
prepare, matrix into an existing block.  usage: python tools_dev/e2e_phases.py [n] [graphs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
graphs = [hsd_amd.barabasi_albert(n, 5, 42 + r) for r in range(reps)]
eng = hsd_amd.Engine(0)
M = None
def t(f):
    eng.sync(); t0 = time.perf_counter(); r = f(); eng.sync(); return (time.perf_counter() - t0) * 1e3, r
for r, g in enumerate(graphs):
    a, _ = t(lambda: eng.set_graph(g.rowptr, g.col))
    b, _ = t(lambda: eng.signature_build_degree(3))
    c, _ = t(lambda: eng.pair_prepare(0, 1, 0, 2))
    if M is None:
        d, M = t(lambda: eng.pair_rows_device())
    else:
        d, _ = t(lambda: eng.pair_rows_device(0, None, M))
    print(f"graph {r}: set_graph {a:7.3f}  sweep + histograms {b:7.3f}  prepare {c:7.3f}  matrix {d:7.3f}   total {a + b + c + d:7.3f} ms", flush=True)
M.free(); eng.close()
