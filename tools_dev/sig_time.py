"""dev: hsd_set_graph / hsd_signature_build_degree wall time, cold and warm, at the target graph"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
g = hsd_amd.barabasi_albert(100000, 5, 42)
eng = hsd_amd.Engine(0)
for rep in range(3):
    eng.sync(); t0 = time.perf_counter()
    eng.set_graph(g.rowptr, g.col)
    eng.sync(); t1 = time.perf_counter()
    eng.signature_build_degree(3)
    eng.sync(); t2 = time.perf_counter()
    eng.signature_build_degree(3)
    eng.sync(); t3 = time.perf_counter()
    print(f"rep {rep}: set_graph {(t1-t0)*1e3:.2f} ms, signature (first) {(t2-t1)*1e3:.2f} ms, signature (again) {(t3-t2)*1e3:.2f} ms", flush=True)
eng.close()
