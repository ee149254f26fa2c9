import os, sys
sys.path.insert(0, "/root/repo")
import hsd_amd
n = 100000
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
M = None
for sort in ("1", "0", "1", "0"):
    os.environ["HSD_GRAN_SORT"] = sort
    eng.pair_prepare(0, 1, 0, 2)
    if M is None:
        M = eng.pair_rows_device(); eng.sync()
    for dbg in ("0", "2", "16"):
        os.environ["HSD_DEBUG"] = dbg
        eng.pair_rows_device(0, None, M); eng.sync()
        ts = []
        for _ in range(4):
            eng.timer_start(); eng.pair_rows_device(0, None, M); ts.append(eng.timer_stop())
        print(f"sorted={sort} HSD_DEBUG={dbg}: best {min(ts):.3f} ms  ({' '.join(f'{t:.2f}' for t in ts)})", flush=True)
    os.environ.pop("HSD_DEBUG", None)
M.free(); eng.close()
