#!/usr/bin/env python3
"""config 5 alone (MultiHSD, BA N=50 000, 8 scales): device time of one parallel_embed-sized pass."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
from hsd_amd.graph import barabasi_albert
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
# optional: impulse rows [r0, r1) only -- ONE block of config 5 (50 cheby_step dispatches instead of 4 000), what the PMC
# passes of rocprofv3 can afford at full size
r0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
r1 = int(sys.argv[3]) if len(sys.argv) > 3 else n
g = barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
m = hsd_amd.MultiHSD(g, "ba", 3, 8, engine=eng)
m._check_default_hierarchy()
sets = [m._cheby(float(s), True) for s in m.scales]
coeffs = np.stack([c for c, _ in sets])
for rep in range(2):
    eng.sync(); t0 = time.perf_counter()
    eng.multiscale_ring_stats(3, coeffs, m.lmax, 1e-4 / n, r0, r1)
    eng.sync(); dt = time.perf_counter() - t0
print(f"HSD_CHEBY_C={os.environ.get('HSD_CHEBY_C','default')} n={n} rows=[{r0},{r1}) seconds={dt:.4f}")
