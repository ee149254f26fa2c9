#!/usr/bin/env python3
"""Ablations of the persistent Variant-B kernel at BA N = 100k (HSD_STAMPS build: HSD_LIB=build/lib_stamps.so): which part of
the result path costs what.  HSD_DEBUG bits: 1 no stores at all (no exchange either), 2 no arithmetic, 16 stores stay in L2,
32 no exchange (mirrored half stores the direct values), 64 no store instructions (exchange and address arithmetic stay).
Timing only: results of the ablated runs are wrong by construction."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd                                                   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
eng.pair_prepare(0, 1, 0, 2)
M = eng.pair_rows_device()
eng.sync()
for label, dbg in (("all", 0), ("no stores, no exchange (1)", 1), ("no store instructions (64)", 64), ("no exchange (32)", 32),
                   ("no exchange, no store instr (96)", 96), ("stores stay in L2 (16)", 16), ("L2 + no exchange (48)", 48),
                   ("no arithmetic (2)", 2), ("all", 0)):
    os.environ["HSD_DEBUG"] = str(dbg)
    eng.pair_rows_device(0, None, M)
    eng.sync()
    ts = []
    for _ in range(reps):
        eng.timer_start()
        eng.pair_rows_device(0, None, M)
        ts.append(eng.timer_stop())
    print(f"{label:36s} {' '.join(f'{t:7.3f}' for t in ts)}  best {min(ts):7.3f} ms", flush=True)
os.environ.pop("HSD_DEBUG", None)
M.free()
eng.close()
