#!/usr/bin/env python3
"""fused k-nearest consumer at BA N = 100k, zero-pad: end-to-end time by number of sweep phases (HSD_TOPK_PHASES)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
os.environ["HSD_TOPK_VERBOSE"] = "1"
for mode, label in (((0, 1, 0, 2), "B"),):
    eng.pair_prepare(*mode)
    for ph in (1, 2, 3, 4, 6, 8, 12, 16):
        os.environ["HSD_TOPK_PHASES"] = str(ph)
        eng.pair_topk(20)
        eng.sync()
        t0 = time.perf_counter()
        eng.pair_topk(20)
        print(label, "phases", ph, f"{(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
eng.close()
