"""dev: randomized soak of the pair kernels: row blocks at arbitrary offsets, persistent / one-tile kernels, tile edges,
forms of the zero-pad layers, tile composition (sorted / natural granule order) -- every combination must reproduce the
rows of a reference matrix (one-tile kernel, bin form), bit for bit for the Wasserstein distances.
usage: python3 tools_dev/fuzz_pair.py [seconds=120] [seed=1] [n_max=4200]      (tests/test_gpu_fuzz.py runs one seed inside -m gpu;
n_max ~12000 brings hubs whose rings exceed a byte: tiles that must not read a layer's uint8 range)"""
import os, sys, time
_tick = [time.time()]
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
from hsd_amd.graph import barabasi_albert, erdos_renyi, connect_components

HOOKS = ("HSD_STREAM", "HSD_TILE", "HSD_RANK", "HSD_SAD16", "HSD_XCD_DEAL", "HSD_SUPERTILE", "HSD_GRAN_SORT", "HSD_U8", "HSD_ROWBLOCK")


def with_env(env, fn):
    os.environ.update(env)
    try:
        return fn()
    finally:
        for v in env:
            os.environ.pop(v, None)


def soak(eng, seed=1, budget_s=120.0, max_blocks=None, n_max=4200):
    """Returns (blocks checked, None) or (blocks, description of the first mismatch)."""
    rng = np.random.default_rng(seed)
    t0, cases = time.time(), 0
    while time.time() - t0 < budget_s and (max_blocks is None or cases < max_blocks):
        n = int(rng.integers(130, n_max))
        gseed = int(rng.integers(1 << 30))
        if rng.random() < 0.6:
            gdesc = ('ba', n, int(rng.integers(2, 6)), gseed); g = barabasi_albert(n, gdesc[2], gseed)
        else:
            gdesc = ('er', n, float(rng.uniform(3, 12)) / n, gseed); g = connect_components(erdos_renyi(n, gdesc[2], seed=gseed), gseed)
        hop = int(rng.integers(1, 4))
        eng.set_graph(g.rowptr, g.col)
        eng.signature_build_degree(hop)
        mode = [(0, 1, 0, max(hop - 1, 0)), (0, 0, 0, hop), (1, 0, 0, hop), (2, 0, 0, hop)][int(rng.integers(0, 4))]
        try:
            ref = with_env({"HSD_STREAM": "0", "HSD_RANK": "0"}, lambda: (eng.pair_prepare(*mode), eng.pair_rows_host())[1])
        except ValueError:
            continue
        exact = mode[0] == 0
        for _ in range(3):
            env = {}
            if rng.random() < 0.7: env["HSD_STREAM"] = "1" if rng.random() < 0.8 else "0"
            if rng.random() < 0.25 and "HSD_STREAM" not in env: env["HSD_TILE"] = "64" if rng.random() < 0.5 else "128"
            if rng.random() < 0.5: env["HSD_RANK"] = str(int(rng.integers(0, 3)))
            if rng.random() < 0.2: env["HSD_SAD16"] = "0"
            if rng.random() < 0.2: env["HSD_XCD_DEAL"] = "0"
            if rng.random() < 0.2: env["HSD_SUPERTILE"] = str(int(rng.choice([1, 4, 16])))
            if rng.random() < 0.3: env["HSD_GRAN_SORT"] = "0"
            if rng.random() < 0.2: env["HSD_U8"] = "0"
            if rng.random() < 0.3: env["HSD_ROWBLOCK"] = "0"       # a row block as three launches instead of one
            gran = int(rng.choice([1, 2, 4, 16, 64, 128, 128, 128]))      # (128: the one-launch row block)
            r0 = int(rng.integers(0, n)) // gran * gran
            r1 = int(rng.integers(r0 + 1, n + 1)) if rng.random() < 0.7 else n

            def run():
                eng.pair_prepare(*mode)                       # forms are chosen at prepare time
                M = eng.pair_rows_device(r0, r1)
                out = M.to_host(); M.free()
                return out
            got = with_env(env, run)
            want = ref[r0:r1]
            if mode[0] == 1:                                  # KL: a row block mirrors the upper triangle of ITS rows; compare the upper part
                iu = np.triu_indices(r1 - r0, 1)
                ok = np.allclose(got[:, r0:r1][iu], want[:, r0:r1][iu], rtol=1e-9, atol=1e-12) and np.allclose(got[:, r1:], want[:, r1:], rtol=1e-9, atol=1e-12)
            else:
                ok = np.array_equal(got, want) if exact else np.allclose(got, want, rtol=1e-9, atol=1e-12)
            if not ok:
                bad = np.argwhere(got != want)
                return cases, dict(graph=gdesc, hop=hop, mode=mode, env=env, rows=(r0, r1), first_bad=bad[:3].tolist(), n_bad=len(bad))
            cases += 1
            if time.time() - _tick[0] > 60:      # (a line a minute: a silent run is taken to be hung by the GPU box's watchdog)
                _tick[0] = time.time(); print(f"... {cases} cases after {time.time() - t0:.0f} s", flush=True)
    return cases, None


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    t0 = time.time()
    cases, bad = soak(hsd_amd.Engine(0), int(sys.argv[2]) if len(sys.argv) > 2 else 1, budget, n_max=int(sys.argv[3]) if len(sys.argv) > 3 else 4200)
    if bad:
        print("MISMATCH", bad, flush=True)
        sys.exit(1)
    print(f"fuzz_pair: {cases} row blocks identical to the reference matrix in {time.time() - t0:.0f} s", flush=True)
