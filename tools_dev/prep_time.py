import os, sys, time
sys.path.insert(0, "/root/repo")
import hsd_amd
g = hsd_amd.barabasi_albert(100000, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
for rep in range(3):
    eng.sync(); t0 = time.perf_counter()
    eng.pair_prepare(0, 1, 0, 2)
    eng.sync()
    print(f"prepare {rep}: {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
eng.close()
