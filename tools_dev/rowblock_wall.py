"""A rank's row-block call: wall time per call (host work + launches + kernel, back to back without a sync in between, as bench.py's timed loop runs it)
against the device time of the same calls.  usage: python tools_dev/rowblock_wall.py [P] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
eng.pair_prepare(0, 1, 0, 2)
for r in (0, P // 2, P - 1):
    r0, r1 = eng.pair_shard(P, r)
    M = eng.pair_rows_device(r0, r1); eng.sync()
    for _ in range(5):
        eng.pair_rows_device(r0, r1, M)
    eng.sync()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        eng.pair_rows_device(r0, r1, M)
    t_issue = time.perf_counter() - t0
    eng.sync()
    wall = (time.perf_counter() - t0) / K * 1e3
    eng.timer_start()
    for _ in range(K):
        eng.pair_rows_device(r0, r1, M)
    dev = eng.timer_stop() / K
    print(f"P={P} rank {r}: wall {wall:.3f} ms per call (host returned after {t_issue / K * 1e3:.3f} ms per call), device {dev:.3f} ms per call", flush=True)
    M.free()
eng.close()
