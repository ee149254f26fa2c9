#!/usr/bin/env python3
"""Symmetry / identity probe of the persistent kernel at full size: sub-matrix on a set of rows against its transpose and
against the one-tile kernel, plus hsd_matrix_stats after each."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
rng = np.random.default_rng(3)
R = np.unique(np.concatenate([rng.integers(0, n, 40), [0, 1, 17, 100, 127, 128, 129, 5000, 5001, 5016, n - 1, n - 2, n - 31, n - 32, n - 33, n - 200]]))
for mode in ((0, 1, 0, 2), (0, 0, 0, 3)):
    eng.pair_prepare(*mode)
    subs = {}
    for label, env in (("one-tile", {"HSD_STREAM": "0"}), ("natural", {"HSD_GRAN_SORT": "0"}), ("sorted", {})):
        for k in ("HSD_STREAM", "HSD_GRAN_SORT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        M = eng.pair_rows_device()
        eng.sync()
        sub = np.stack([M.rows_to_host(int(r), int(r) + 1)[0][R] for r in R])
        st = M.stats()
        M.free()
        subs[label] = sub
        bad = np.argwhere(sub != sub.T)
        print(mode, label, "stats", st, "sub asym entries", len(bad), [(int(R[a]), int(R[b]), sub[a, b], sub[b, a]) for a, b in bad[:6]], flush=True)
    for k in ("natural", "sorted"):
        d = np.argwhere(subs[k] != subs["one-tile"])
        print("   ", k, "vs one-tile differing", len(d), [(int(R[a]), int(R[b]), subs[k][a, b], subs["one-tile"][a, b]) for a, b in d[:6]], flush=True)
eng.close()
