"""dev: randomized soak of the bit-parallel sweep against the per-source sweep (counts must be identical).
usage: python3 tools_dev/fuzz_khop.py [seconds=60] [seed=1]"""
import os, sys, time
_tick = [time.time()]
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
from hsd_amd.graph import barabasi_albert, erdos_renyi
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = hsd_amd.Engine(0)
t0, cases = time.time(), 0
def counts(hop, ms, r0, r1, nb=None, sup=None):
    os.environ["HSD_KHOP_MS"] = ms
    if nb is None: eng.signature_build_degree(hop, r0, r1)
    else: eng.signature_build(hop, nb, sup, r0, r1)
    return eng.signature_counts(r0, r1)
while time.time() - t0 < budget:
    kind = rng.integers(0, 3)
    n = int(rng.integers(40, 30000))
    if kind == 0: g = barabasi_albert(n, int(rng.integers(1, 8)), int(rng.integers(1 << 30)))
    elif kind == 1: g = erdos_renyi(n, float(rng.uniform(0.2, 12.0)) / n, seed=int(rng.integers(1 << 30)))
    else: g = barabasi_albert(n, 2, int(rng.integers(1 << 30)))
    eng.set_graph(g.rowptr, g.col)
    for _ in range(3):
        hop = int(rng.integers(0, 7))
        r0 = int(rng.integers(0, n)); r1 = int(rng.integers(r0, n + 1))
        if rng.random() < 0.4: r0, r1 = 0, n
        if rng.random() < 0.5:
            nbins = int(rng.integers(1, 300)); nb = rng.integers(0, nbins, n).astype(np.int32); sup = np.arange(nbins, dtype=np.float64)
        else: nb = sup = None
        a = counts(hop, "0", r0, r1, nb, sup); b = counts(hop, "1", r0, r1, nb, sup)
        if not np.array_equal(a, b):
            print("MISMATCH", dict(kind=int(kind), n=n, hop=hop, r0=r0, r1=r1, bins=None if nb is None else int(nb.max()) + 1), flush=True)
            sys.exit(1)
        cases += 1
        if time.time() - _tick[0] > 60:      # (a line a minute: a silent run is taken to be hung by the GPU box's watchdog)
            _tick[0] = time.time(); print(f"... {cases} cases after {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_khop: {cases} cases identical in {time.time() - t0:.0f} s", flush=True)
