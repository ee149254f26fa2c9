"""dev: every rank's row block of the strong-scaling bench (`bench.py --gpus P`: the SAME matrix in P row blocks) timed one after the
other on ONE GPU -- what the slowest of P identical GPUs would take, i.e. the strong-scaling curve the row-block partition allows before
any communication or launch skew.  usage: strong_share.py [n=100000] [variants=BA] [P list=2,4,8]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
from hsd_amd.distributed import shard_rows
from hsd_amd.engine import DeviceMatrix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
variants = sys.argv[2] if len(sys.argv) > 2 else "BA"
plist = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "2,4,8").split(",")]
g = hsd_amd.barabasi_albert(n, 5, seed=42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
ld = (n + 15) // 16 * 16
out = {}
for name in variants:
    mode = (0, 1, 0, 2) if name == "B" else (0, 0, 0, 3)
    eng.pair_prepare(*mode)
    reps = 10 if name == "B" else 2
    res = {}
    balanced = os.environ.get("SHARE_BALANCED", "1") != "0"     # hsd_pair_shard's blocks (equal planned cost) or equal heights
    for P in ([] if 8.0 * n * ld > 200e9 else [1]) + plist:          # (the whole matrix of N = 200k does not fit one GPU)
        rows = []
        for r in range(P):
            r0, r1 = (eng.pair_shard(P, r) if balanced else shard_rows(n, P, r)) if P > 1 else (0, n)
            M = DeviceMatrix(eng, r1 - r0, n, ld, r0)
            eng.pair_rows_device(r0, r1, M)
            eng.pair_rows_device(r0, r1, M)
            eng.timer_start()
            for _ in range(reps):
                eng.pair_rows_device(r0, r1, M)
            ms = eng.timer_stop() / reps
            words = eng.pair_work() / max(1.0, (r1 - r0) * float(n))
            M.free()
            rows.append(round(ms, 3))
            print(f"variant {name} P={P} rank {r} rows {r0}..{r1}: {ms:.3f} ms  ({words:.1f} operand words per computed pair)", flush=True)
        res[P] = rows
    t1 = res[1][0] if 1 in res else float("nan")
    for P in plist:
        print(f"variant {name} P={P}: slowest rank {max(res[P]):.3f} ms, fastest {min(res[P]):.3f}; speed-up over one GPU {t1 / max(res[P]):.2f} "
              f"(balanced ranks would give {t1 / (sum(res[P]) / P):.2f})", flush=True)
    out[name] = res
print(json.dumps(out))
