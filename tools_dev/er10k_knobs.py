"""dev: config 2 (ER N = 10k, K = 3) zero-pad and raw fp64 W1 under the launch knobs that exist (tile edge, persistent kernel, group size,
XCD dealing, resident waves): is anything left on the table at this size?  usage: er10k_knobs.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
eng = hsd_amd.Engine(0)
g = hsd_amd.erdos_renyi(10000, 0.001, 42)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
KNOBS = [{}, {"HSD_TILE": "128"}, {"HSD_TILE": "64"}, {"HSD_STREAM": "1"}, {"HSD_STREAM": "1", "HSD_WAVES_PER_CU": "16"}, {"HSD_SUPERTILE": "4"},
         {"HSD_SUPERTILE": "16"}, {"HSD_XCD_DEAL": "0"}, {"HSD_STREAM": "1", "HSD_SUPERTILE": "4"}, {"HSD_RANK": "0"}, {"HSD_RANK": "1"}]
for name, mode in (("B", (0, 1, 0, 2)), ("A", (0, 0, 0, 3))):
    ref = None
    for env in KNOBS:
        os.environ.update(env)
        try:
            eng.pair_prepare(*mode)
            M = eng.pair_rows_device()
            eng.sync()
            ts = []
            for _ in range(30):
                eng.timer_start(); eng.pair_rows_device(0, None, M); ts.append(eng.timer_stop())
            s = M.stats()["sum"]
            M.free()
        finally:
            for k in env:
                os.environ.pop(k, None)
        ref = s if ref is None else ref
        ts.sort()
        print(f"{name} {str(env):60s} best {ts[0]:.4f}  median {ts[len(ts) // 2]:.4f} ms   sum identical: {s == ref}", flush=True)
eng.close()
