"""CPU: the measurement harness's host logic -- workload table, sampling, roofline arithmetic, rank plumbing --
without a GPU (bench.py's GPU legs are exercised by the driver and by tools_dev/rehearse_ranks.sh)."""
import json
import sys

import numpy as np

from conftest import REPO

sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_workloads_are_baseline_configs():
    names = json.load(open(f"{REPO}/BASELINE.json"))["configs"]
    w = bench.WORKLOADS
    assert w["target"]["nodes"] == 100000 and w["target"]["hop"] == 3 and w["target"]["graph"] == "ba"
    assert w["config2"]["nodes"] == 10000 and w["config2"]["p"] == 0.001 and "10 000" in names[1]
    assert w["config3"]["nodes"] == 50000 and w["config3"]["hop"] == 5 and w["config3"]["metric"] == "kl" and "K=5" in names[2]
    assert w["config4"]["nodes"] == 200000 and "200 000" in names[3]
    g = bench.make_graph(dict(graph="er", nodes=2000, p=0.005))
    assert g.n_node == 2000 and abs(g.n_edge - 2000 * 1999 / 2 * 0.005) < 600
    assert bench.variant_params("A", 3) == (0, 0, 3) and bench.variant_params("B", 3) == (1, 0, 2)


def test_sample_nodes_spread_and_deterministic():
    a = bench.sample_nodes(100000, 400)
    assert len(a) == 400 and len(np.unique(a)) == 400 and np.all(np.diff(a) > 0)
    assert np.array_equal(a, bench.sample_nodes(100000, 400)) and a[0] < 5000 and a[-1] > 95000
    assert len(bench.sample_nodes(50, 400)) == 50


def test_scipy_baseline_matches_oracle_distance():
    """The CPU baseline loop is the reference's loop: its distances equal the C oracle's on the same pairs."""
    from oracle import c_oracle as corc
    g = bench.make_graph(dict(graph="ba", nodes=600, m=5))
    for variant in ("A", "B"):
        out = bench.scipy_baseline(g, 3, variant, budget_s=0.5, n_sample=24)
        assert out["pairs"] >= 64 and out["rate"] > 0
        align, lo, hi = bench.variant_params(variant, 3)
        nr = corc.NodeRingSamples(g.rowptr, g.col, 3, g.degree.astype(np.float64), out["nodes"])
        for (i, j), d in list(out["sample"].items())[:40]:
            assert i < j and abs(d - nr.w1(i, j, lo, hi, align)) <= 1e-9 * max(1.0, abs(d))


def test_kernel_roofline_arithmetic():
    res = dict(ebytes=8, cols=721, ev_ms=2200.0, align=0, work=None)
    r = bench.kernel_roofline(100000, 100000, res, 10, "wasserstein")
    assert r["bytes_alg"] == 8.0 * 100000 * 100000 + 8.0 * 100000 * 721
    assert abs(r["kernel_ms"] - 220.0) < 1e-9 and abs(r["achieved"] - r["bytes_alg"] / 0.22 / 1e9) < 0.1
    pairs = 100000 * 100001 / 2.0
    assert abs(r["valu_frac"] - pairs * 721 * 2 / 0.22 / bench.FP64_VALU_PEAK) < 1e-3 and r["limiter"] == "vector ALU issue"
    # the kernel's own work count replaces pairs x columns when given (rank form: fewer words than the bin form)
    res = dict(ebytes=4, cols=245, ev_ms=224.0, align=1, work=pairs * 103.0)
    r = bench.kernel_roofline(100000, 100000, res, 10, "wasserstein")
    assert abs(r["columns_evaluated_per_pair"] - 103.0) < 1e-6 and abs(r["valu_frac"] - pairs * 103 / 0.0224 / bench.SAD_VALU_PEAK) < 1e-3
    # a rank's row block owes its diagonal square once and the rectangles fully
    res = dict(ebytes=8, cols=1000, ev_ms=1.0, align=0, work=None)
    r = bench.kernel_roofline(1000, 250, res, 1, "wasserstein")
    assert abs(r["valu_frac"] - (250 * 251 / 2 + 250 * 750) * 1000 * 2 / 0.001 / bench.FP64_VALU_PEAK) < 1e-4


def test_ranks_without_process_group():
    rk = bench.Ranks(None, 0, 1)
    rk.barrier()
    assert rk.max(3.5) == 3.5 and rk.gather(2.0) == [2.0] and rk.any(True) and not rk.any(False)


def test_traffic_table_is_keyed_like_the_bench():
    t = json.load(open(f"{REPO}/profiles/r02_traffic.json"))
    for key in ("ba_n100000_K3_A", "ba_n100000_K3_B", "ba_n50000_K5_KL", "er_n10000_K3_A", "er_n10000_K3_B"):
        assert abs(t[key]["bytes"] - t[key]["read_bytes"] - t[key]["write_bytes"]) <= 2 and t[key]["write_bytes"] > 0
    assert bench.load_traffic("ba_n100000_K3_A")["source"].startswith("profiles/")
