#!/usr/bin/env python3
"""dev: config 3's KL kernel (BA N = 50k, K = 5) under the library named by HSD_LIB: ms per matrix.  usage: ab_kl.py [n=50000] [reps=5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(5)
for name, mode in (("KL", (1, 0, 0, 5)), ("SKL", (2, 0, 0, 5))):
    eng.pair_prepare(*mode, 0.5, False)
    M = eng.pair_rows_device()
    eng.pair_rows_device(0, None, M)
    ts = []
    for _ in range(reps):
        eng.timer_start()
        eng.pair_rows_device(0, None, M)
        ts.append(eng.timer_stop())
    st = M.stats()
    print(f"{os.path.basename(os.environ.get('HSD_LIB', 'default')):16s} {name:4s} " + " ".join(f"{t:7.2f}" for t in ts) + f" ms  best {min(ts):.2f}  sum {st['sum']!r}", flush=True)
    M.free()
eng.close()
