"""dev: randomized soak of the fused k-nearest consumer (phased sweep, persistent and one-tile kernels, row blocks)
against a lexicographic selection on the full matrix.  usage: python3 tools_dev/fuzz_topk.py [seconds=90] [seed=1]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
from hsd_amd.graph import barabasi_albert, erdos_renyi, connect_components
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = hsd_amd.Engine(0)
t0, cases = time.time(), 0
while time.time() - t0 < budget:
    n = int(rng.integers(200, 5000))
    seed = int(rng.integers(1 << 30))
    if rng.random() < 0.5:
        gdesc = ('ba', n, int(rng.integers(2, 6)), seed)
        g = barabasi_albert(n, gdesc[2], seed)
    else:
        gdesc = ('er', n, float(rng.uniform(3, 12)) / n, seed)
        g = connect_components(erdos_renyi(n, gdesc[2], seed=seed), seed)
    hop = int(rng.integers(1, 4))
    eng.set_graph(g.rowptr, g.col)
    eng.signature_build_degree(hop)
    mode = [(0, 1, 0, max(hop - 1, 0)), (0, 0, 0, hop), (2, 0, 0, hop)][int(rng.integers(0, 3))]
    try:
        eng.pair_prepare(*mode)
        D = eng.pair_rows_host()
    except ValueError:          # an empty ring under the raw-sample distance
        continue
    cols = np.arange(n)
    for _ in range(2):
        k = int(rng.integers(1, min(n - 1, 70)))
        env = dict(HSD_TOPK_PHASES=str(int(rng.integers(1, 9))))
        if rng.random() < 0.6: env["HSD_STREAM"] = "1"
        if rng.random() < 0.2: env["HSD_TOPK_CAP"] = str(k + int(rng.integers(1, 40)))
        a = int(rng.integers(0, n - 1)) // 4 * 4 if rng.random() < 0.5 else 0
        b = int(rng.integers(a + 1, n + 1)) if a else n
        os.environ.update(env)
        try:
            idx, val = eng.pair_topk(k, a, b)
        except Exception as e:
            print("EXCEPTION", type(e).__name__, str(e)[:120], dict(graph=gdesc, hop=hop, mode=mode, k=k, env=env, rows=(a, b)), flush=True)
            sys.exit(1)
        finally:
            for v in env: os.environ.pop(v, None)
        for r in rng.integers(a, b, 12):
            d = D[r].copy(); d[r] = np.inf
            order = np.lexsort((cols, d))[:k]
            same_idx = np.array_equal(order, idx[r - a])
            same_val = np.array_equal(d[order], val[r - a]) if mode[0] != 2 else np.allclose(d[order], val[r - a], rtol=1e-9, atol=1e-12)   # symmetric KL: H_i + H_j - dots cancels, the two orientations of a pair differ by ~1e-15 absolute
            if not (same_val and (same_idx or mode[0] == 2)):
                print("MISMATCH", dict(graph=gdesc, hop=hop, mode=mode, k=k, env=env, rows=(a, b), row=int(r)), flush=True)
                sys.exit(1)
        cases += 1
print(f"fuzz_topk: {cases} calls exact in {time.time() - t0:.0f} s", flush=True)
