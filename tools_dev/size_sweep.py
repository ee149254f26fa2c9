"""dev: the headline kernel (Variant B, zero-pad W1, K = 3) across graph sizes and families: ms, fraction of the 8 TB/s HBM roofline on the
algorithmic bytes (8 N^2 + operands), operand words per pair, and the hipMemsetAsync of the same output beside it.
usage: size_sweep.py [variants=B]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
which = sys.argv[1] if len(sys.argv) > 1 else "B"
eng = hsd_amd.Engine(0)
CASES = [("ba", 20000), ("ba", 40000), ("ba", 70000), ("ba", 100000), ("ba", 150000), ("er", 20000), ("er", 50000), ("er", 100000), ("er", 150000)]
print(f"{'graph':>6} {'n':>7} {'var':>3} {'best ms':>9} {'mean ms':>9} {'of 8 TB/s (mean)':>17} {'words/pair':>11} {'fill ms (mean)':>15} {'kernel/fill':>11}", flush=True)
for kind, n in CASES:
    g = hsd_amd.barabasi_albert(n, 5, 42) if kind == "ba" else hsd_amd.erdos_renyi(n, 10.0 / n, 42)
    eng.set_graph(g.rowptr, g.col)
    eng.signature_build_degree(3)
    for v in which:
        try:
            eng.pair_prepare(*((0, 1, 0, 2) if v == "B" else (0, 0, 0, 3)))
        except ValueError as e:
            print(f"{kind:>6} {n:>7} {v:>3} {e}", flush=True)
            continue
        M = eng.pair_rows_device()
        eng.pair_rows_device(0, None, M)
        reps = 10 if v == "B" else 2
        ts = []
        for _ in range(reps):
            eng.timer_start(); eng.pair_rows_device(0, None, M); ts.append(eng.timer_stop())
        ms = sum(ts) / len(ts)
        words = eng.pair_work() / (n * (n - 1) / 2.0)
        cols, eb = eng.pair_operand_dims()
        alg = 8.0 * n * n + float(eb) * n * cols
        fills = []
        for _ in range(5):
            eng.timer_start(); eng._call("hsd_memset", C.c_void_p(M.ptr), 0, C.c_size_t(M.rows * M.ld * 8)); fills.append(eng.timer_stop())
        fill = sum(fills) / len(fills)
        M.free()
        print(f"{kind:>6} {n:>7} {v:>3} {min(ts):9.3f} {ms:9.3f} {alg / (ms * 1e-3) / 8e12:17.3f} {words:11.1f} {fill:15.3f} {ms / fill:11.2f}", flush=True)
eng.close()
