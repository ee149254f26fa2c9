"""dev: the seeded random-graph tests of tests/test_gpu_random.py (every histogram mode and the coefficient-sample kernels
against the numpy oracle) over MANY more seeds than the test suite runs.  usage: python3 tools_dev/soak_random.py [seconds=120] [first_seed=100]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import c_oracle as _c; _c.lib()               # built before the GPU is touched
import test_gpu_random as t
from hsd_amd.engine import Engine
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
eng = Engine(0)
t0, n_a, n_h, tick = time.time(), 0, 0, time.time()
while time.time() - t0 < budget:
    t.test_random_graph_all_modes(eng, seed); n_a += 1
    t.test_random_graph_heat_sample_kernels(eng, seed); n_h += 1
    seed += 1
    if time.time() - tick > 60:         # (a line a minute: a silent run is taken to be hung by the GPU box's watchdog)
        tick = time.time(); print(f"... {n_a} graphs after {time.time() - t0:.0f} s", flush=True)
print(f"soak_random: {n_a} graphs through every histogram mode and {n_h} through the coefficient-sample kernels match the oracle ({time.time() - t0:.0f} s, seeds up to {seed - 1})", flush=True)
