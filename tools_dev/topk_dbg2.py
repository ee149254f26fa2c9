#!/usr/bin/env python3
"""fused k-nearest consumer at BA N = 100k: per-attempt list statistics (HSD_TOPK_VERBOSE) and timing, sorted / natural order."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
os.environ["HSD_TOPK_VERBOSE"] = "1"
for mode, label in (((0, 1, 0, 2), "B"), ((0, 0, 0, 3), "A")):
    eng.pair_prepare(*mode)
    for env in ({}, {"HSD_GRAN_SORT": "0"}, {"HSD_TOPK_DEBUG": "1"}):
        for k in ("HSD_GRAN_SORT", "HSD_TOPK_DEBUG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            eng.pair_topk(20)
        except Exception as e:                      # HSD_TOPK_DEBUG=1: no candidates -> the call fails by design (timing only)
            pass
        eng.sync()
        t0 = time.perf_counter()
        try:
            eng.pair_topk(20)
        except Exception as e:
            pass
        print(label, env, f"{(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
eng.close()
