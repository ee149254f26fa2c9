#!/usr/bin/env python3
"""A/B of the persistent pair kernel inside ONE process (environment hooks are read per call): BA N = 100k K = 3,
Variant B (zero-pad) and Variant A (raw fp64); granule order sorted / natural, and the arith-only / store-only ablations.
Sampled rows are compared bit for bit with the first configuration."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd                                                   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
which = sys.argv[2] if len(sys.argv) > 2 else "BA"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
rows = [0, 777, n // 2 + 1, n - 1]
HOOKS = ("HSD_GRAN_SORT", "HSD_DEBUG", "HSD_STREAM")


def run(label, env, M, ref):
    for h in HOOKS:
        os.environ.pop(h, None)
    os.environ.update(env)
    eng.pair_rows_device(0, None, M)
    eng.sync()
    ts = []
    for _ in range(reps):
        eng.timer_start()
        eng.pair_rows_device(0, None, M)
        ts.append(eng.timer_stop())
    words = eng.pair_work() / (n * (n - 1) / 2.0)
    same = None
    if "HSD_DEBUG" not in env:
        got = [M.rows_to_host(r, r + 1).copy() for r in rows]
        if ref is None:
            ref = got
        same = all(np.array_equal(a, b) for a, b in zip(got, ref))
    alg = 8.0 * n * n
    best = min(ts) * 1e-3
    print(f"{label:34s} {' '.join(f'{t:8.3f}' for t in ts)} ms  best {best * 1e3:8.3f}  -> {alg / best / 1e9:7.1f} GB/s alg = {alg / best / 8e12:.3f} of 8 TB/s; "
          f"words/pair {words:.1f}; rows identical: {same}", flush=True)
    for h in HOOKS:
        os.environ.pop(h, None)
    return ref


M = None
for variant in which:
    mode = (0, 1, 0, 2) if variant == "B" else (0, 0, 0, 3)
    os.environ["HSD_PLAN_VERBOSE"] = "1"
    eng.pair_prepare(*mode)
    if M is None:
        M = eng.pair_rows_device()
        eng.sync()
    print(f"== variant {variant}", flush=True)
    ref = None
    ref = run("one-tile kernel (reference)", {"HSD_STREAM": "0"}, M, ref)
    run("persistent, natural order", {"HSD_GRAN_SORT": "0"}, M, ref)
    run("persistent, sorted granules", {}, M, ref)
    run("persistent, natural order", {"HSD_GRAN_SORT": "0"}, M, ref)
    run("persistent, sorted granules", {}, M, ref)
    st = M.stats()
    print("   matrix stats:", st, flush=True)
    run("sorted, no stores", {"HSD_DEBUG": "1"}, M, ref)
    run("sorted, no arithmetic", {"HSD_DEBUG": "2"}, M, ref)
    run("sorted, stores stay in L2", {"HSD_DEBUG": "16"}, M, ref)
M.free()
eng.close()
