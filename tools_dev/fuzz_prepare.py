"""dev: randomized check of the operand preparation against the C oracle on CUSTOM signatures whose number of bins sweeps the staging regimes of
csrc/pair_prepare.hip (counts_row: 64 / 32 / 16 nodes' histogram rows per workgroup through LDS, or in place for very wide supports), node counts that are not
multiples of the stage, integral and fractional supports (exact-integer path / fp64 fallback), every distance kind.
usage: python3 tools_dev/fuzz_prepare.py [seconds=120] [seed=1]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
from hsd_amd.graph import barabasi_albert, erdos_renyi, connect_components
from oracle import c_oracle as corc
from oracle import hsd_oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
eng = hsd_amd.Engine(0)
t0, cases, worst = time.time(), 0, 0.0
BINS = [3, 17, 64, 130, 241, 255, 256, 300, 511, 600, 1023, 1100]
while time.time() - t0 < budget:
    n = int(rng.integers(150, 1400))
    gseed = int(rng.integers(1 << 30))
    g = barabasi_albert(n, int(rng.integers(2, 5)), gseed) if rng.random() < 0.5 else connect_components(erdos_renyi(n, float(rng.uniform(3, 9)) / n, seed=gseed), gseed)
    hop = int(rng.integers(1, 4))
    B = int(BINS[int(rng.integers(len(BINS)))])
    B = min(B, n)
    integral = rng.random() < 0.7
    steps = rng.integers(1, 6, size=B).astype(np.float64) if integral else rng.uniform(0.25, 3.0, size=B)
    support = np.cumsum(steps)
    node_bin = rng.integers(0, B, size=n).astype(np.int32)
    node_bin[rng.integers(0, n, size=min(n, B))] = np.arange(min(n, B), dtype=np.int32)[:min(n, B)]      # most bins occupied
    value = support[node_bin]
    eng.set_graph(g.rowptr, g.col)
    eng.signature_build(hop, node_bin, support)
    rs = corc.RingSamples(g.rowptr, g.col, hop, value)
    r0 = int(rng.integers(0, n - 64)); r1 = min(n, r0 + int(rng.integers(16, 200)))
    staged = "1" if rng.random() < 0.8 else "0"
    os.environ["HSD_PREP_STAGED"] = staged
    for kind in ("B", "A", "KL", "SKL"):
        if kind == "B":
            hi = max(hop - 1, 0)
            eng.pair_prepare(0, 1, 0, hi); want, _ = rs.pair_rows(0, hi, corc.ALIGN_ZEROPAD, r0, r1)
        elif kind == "A":
            want, empty = rs.pair_rows(0, hop, corc.ALIGN_NONE, r0, r1)
            try:
                eng.pair_prepare(0, 0, 0, hop)
            except ValueError:
                assert empty, ("GPU raised for an empty ring the oracle does not see", n, hop, B)
                continue
            assert not empty, ("oracle sees an empty ring, the GPU did not raise", n, hop, B)
        else:
            counts = eng.signature_counts()
            alpha = float(rng.choice([0.5, 0.1, 1.0]))
            eng.pair_prepare(1 if kind == "KL" else 2, 0, 0, hop, alpha, False)
            want = corc.pair_rows_kl(counts, 0, hop, alpha, kind == "SKL", r0, r1)
        got = eng.pair_rows_host(r0, r1)
        err = np.abs(got - want) / np.maximum(np.abs(want), 1e-12)
        err[np.abs(got - want) <= 1e-12] = 0.0
        worst = max(worst, float(err.max()))
        if err.max() > 1e-6:
            print("MISMATCH", dict(n=n, hop=hop, B=B, integral=integral, kind=kind, staged=staged, gseed=gseed, rows=(r0, r1), err=float(err.max())), flush=True)
            sys.exit(1)
    cases += 1
    if cases % 25 == 0:
        print(f"... {cases} signatures after {time.time() - t0:.0f} s (worst relative error {worst:.2e})", flush=True)
os.environ.pop("HSD_PREP_STAGED", None)
print(f"fuzz_prepare: {cases} custom signatures x 4 distance kinds within 1e-6 of the C oracle (worst {worst:.2e}) in {time.time() - t0:.0f} s")
