#!/usr/bin/env python3
"""one fused k-nearest call at BA N = 100k (variant from argv: B | A), for rocprofv3 --kernel-trace --stats"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
variant = sys.argv[1] if len(sys.argv) > 1 else "B"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
eng.pair_prepare(*((0, 1, 0, 2) if variant == "B" else (0, 0, 0, 3)))
for rep in range(3):
    eng.sync(); t0 = time.perf_counter()
    eng.pair_topk(20)
    print(f"{variant} top-20 call {rep}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
eng.close()
