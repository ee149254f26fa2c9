import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, hsd_amd
eng = hsd_amd.Engine(0)
g = hsd_amd.barabasi_albert(100000, 5, 42)
eng.set_graph(g.rowptr, g.col); eng.signature_build_degree(3); eng.pair_prepare(0, 1, 0, 2)
for dbg, ph in (("0", "8"), ("1", "8"), ("1", "1"), ("0", "1"), ("1", "4"), ("0", "4")):
    os.environ["HSD_TOPK_DEBUG"] = dbg; os.environ["HSD_TOPK_PHASES"] = ph
    try:
        eng.pair_topk(20)
    except Exception as e:
        pass
    eng.sync(); t0 = time.perf_counter()
    for _ in range(3):
        try: eng.pair_topk(20)
        except Exception as e: err = type(e).__name__
    print(f"HSD_TOPK_DEBUG={dbg} phases={ph}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)
