"""dev: where the persistent pair kernel overtakes the one-tile kernel (HSD_STREAM=0 / 1) by graph size.  usage: xover.py [B|A|KL]"""
import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/hsd_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import hsd_amd
eng = hsd_amd.Engine(0)
mode = {"B": (0, 1, 0, 2), "A": (0, 0, 0, 3), "KL": (1, 0, 0, 3)}[sys.argv[1] if len(sys.argv) > 1 else "B"]
for n in (10000, 12000, 14000, 16000):
    for kind in ("ba", "er"):
        g = hsd_amd.barabasi_albert(n, 5, 42) if kind == "ba" else hsd_amd.erdos_renyi(n, 10.0 / n, 42)
        eng.set_graph(g.rowptr, g.col)
        eng.signature_build_degree(3)
        res = []
        for s in ("0", "1"):
            os.environ["HSD_STREAM"] = s
            try:
                eng.pair_prepare(*mode)
            except ValueError:
                res.append(float("nan")); continue
            M = eng.pair_rows_device()
            eng.sync()
            ts = []
            for _ in range(5):
                eng.timer_start(); eng.pair_rows_device(0, None, M); ts.append(eng.timer_stop())
            M.free()
            res.append(min(ts))
        os.environ.pop("HSD_STREAM", None)
        print(f"{kind} n={n}: one-tile {res[0]:.4f} ms, persistent {res[1]:.4f} ms", flush=True)
eng.close()
