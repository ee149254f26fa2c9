"""dev: k-hop sweep + histogram time (signature_build_degree) on the BASELINE graph shapes, per-source sweep
(HSD_KHOP_MS=0) against the bit-parallel sweep (HSD_KHOP_MS=1) in one box; the counts must be identical."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
eng = hsd_amd.Engine(0)
which = sys.argv[1:] or ["er10k", "ba50k5", "ba100k", "ba200k"]
shapes = {"er10k": ("er10k K3", lambda: hsd_amd.erdos_renyi(10000, 0.001, 42), 3),
          "ba50k5": ("ba50k K5", lambda: hsd_amd.barabasi_albert(50000, 5, 42), 5),
          "ba100k": ("ba100k K3", lambda: hsd_amd.barabasi_albert(100000, 5, 42), 3),
          "ba200k": ("ba200k K3", lambda: hsd_amd.barabasi_albert(200000, 5, 42), 3)}
for key in which:
    name, make, hop = shapes[key]
    g = make()
    eng.set_graph(g.rowptr, g.col)
    sums = {}
    for ms in ("0", "1"):
        os.environ["HSD_KHOP_MS"] = ms
        eng.signature_build_degree(hop)
        best_wall, best_dev = 1e9, 1e9
        for _ in range(3):
            eng.sync(); t = time.perf_counter(); eng.timer_start(); eng.signature_build_degree(hop); dev = eng.timer_stop()
            best_wall = min(best_wall, time.perf_counter() - t); best_dev = min(best_dev, dev)
        c = eng.signature_counts()
        w = (1 + np.arange(c.shape[2], dtype=np.uint64))[None, None, :] * (1 + np.arange(c.shape[1], dtype=np.uint64))[None, :, None]
        sums[ms] = (int(c.sum()), int((c.astype(np.uint64) * w).sum()), c[::997].copy())
        print(f"{name:12s} HSD_KHOP_MS={ms}  wall {best_wall * 1e3:9.2f} ms  stream {best_dev:9.2f} ms  checksum {sums[ms][0]} {sums[ms][1]}", flush=True)
    same = sums["0"][:2] == sums["1"][:2] and np.array_equal(sums["0"][2], sums["1"][2])
    print(f"{name:12s} identical: {same}", flush=True)
