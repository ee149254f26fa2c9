import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
from hsd_amd.graph import barabasi_albert
eng = hsd_amd.Engine(0)
g = barabasi_albert(2675, 3, 363629067)
n = g.n_node
eng.set_graph(g.rowptr, g.col); eng.signature_build_degree(3); eng.pair_prepare(0, 1, 0, 2)
D = eng.pair_rows_host(); cols = np.arange(n)
def check(k, a, b, env):
    os.environ.update(env)
    try: idx, val = eng.pair_topk(k, a, b)
    finally:
        for v in env: os.environ.pop(v, None)
    bad = 0
    for r in range(a, b):
        d = D[r].copy(); d[r] = np.inf
        order = np.lexsort((cols, d))[:k]
        if not (np.array_equal(order, idx[r - a]) and np.array_equal(d[order], val[r - a])): bad += 1
    return bad
for env in ({"HSD_TOPK_PHASES": "3", "HSD_STREAM": "1", "HSD_TOPK_CAP": "75"}, {"HSD_TOPK_PHASES": "3", "HSD_STREAM": "1"}, {"HSD_TOPK_PHASES": "1", "HSD_STREAM": "1", "HSD_TOPK_CAP": "75"},
            {"HSD_TOPK_PHASES": "3", "HSD_TOPK_CAP": "75"}, {"HSD_TOPK_PHASES": "1", "HSD_TOPK_CAP": "75"}, {"HSD_TOPK_PHASES": "1"}, {"HSD_TOPK_PHASES": "3", "HSD_STREAM": "1", "HSD_TOPK_CAP": "75", "HSD_TOPK_VERBOSE": "1"}):
    for (a, b) in ((476, 2350), (0, n), (476, 2352)):
        print(env, (a, b), "bad rows:", check(41, a, b, env), flush=True)
