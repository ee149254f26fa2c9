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


def _detail(n_configs=5, pad=1):
    rf = dict(bound="hbm", achieved=5455.9, peak=8000.0, unit="GB/s", frac=0.682, traffic=83172966707, traffic_over_bytes_alg=1.038,
              traffic_detail=dict(kernels=["pair_stream_kernel<4, 4, false>"] * 40 * pad), kernel_ms=14.681, bytes_alg=80098000000.0, valu_frac=0.4464,
              valu_peak_lane_instr_per_s=39.3e12, limiter="HBM write stream", operand_columns=245, fill_same_bytes=dict(ms=13.3))
    var = dict(variant="B", value=3.4e11, unit="node-pairs/s", ms_per_step=14.7, roofline=rf, parity=dict(ok=True), fused_knn=dict(k=20, note="x" * 3000 * pad))
    return dict(metric=bench.METRIC, value=3.4e11, unit="node-pairs/s", n_gpus=1, steps=20, warmup=5, ms_per_step=14.7, higher_is_better=True, scaling="strong",
                vs_baseline=None, dtype="u16", data="synthetic",
                config=dict(workload="w" * 900 * pad, nodes=100000, edges=499975, hop=3, bins=241, parallelism="row-block x1", rows_of_rank0=[0, 100000]),
                roofline=rf, cpu_baseline=dict(value=17940.6, unit="node-pairs/s", cores=1, kind="port", sample="s" * 2000 * pad),
                cpu_baseline_pool=dict(value=79813.9, unit="node-pairs/s", cores=64, kind="port", sample="p" * 500),
                parity=dict(pairs_checked=79800, max_rel_err=float("nan") if pad > 1 else 4e-15, tol=1e-6, ok=True),
                secondary=dict(var, variant="A"), kernel_ms_per_rank=[14.68], other_kernels=[dict(note="k" * 4000)], phases_s=dict(allgather=0.0, exchange="none"),
                fused_knn=dict(k=20, ms_end_to_end=11.98, exact=True, note="n" * 2000), row_block_rehearsal=dict(note="r" * 5000),
                configs=[dict(config=f"configs[{i}]", workload="c" * 500, variants=[dict(var, variant="A"), var], heat_signature_path=dict(x="h" * 3000))
                         for i in range(n_configs)] + [dict(config="5", error="RuntimeError: " + "e" * 600)],
                device="gfx950:sramecc+:xnack-", cus=256)


def test_contract_line_stays_compact_and_strict(tmp_path, monkeypatch, capsys):
    """The driver reads a tail of stdout: the contract line is < 4 KB whatever was measured, carries the contract's keys with the roofline /
    cpu_baseline / parity objects, is the ONLY stdout line and is strict JSON (a non-finite figure becomes null); everything else goes to
    the detail record."""
    def strict(text):
        def no_constants(name):
            raise ValueError(name)
        return json.loads(text, parse_constant=no_constants)
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    for n_configs, pad in ((5, 1), (40, 8)):
        bench.emit(_detail(n_configs, pad), "d.json")
        cap = capsys.readouterr()
        out = [ln for ln in cap.out.splitlines() if ln.strip()]
        assert len(out) == 1 and out[0].startswith('{"metric"') and len(out[0].encode()) < 4096
        d = strict(out[0])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
            assert k in d
        assert set(d["roofline"]) == set(bench.ROOFLINE_KEYS) and d["roofline"]["frac"] == 0.682 and d["roofline"]["traffic"] == 83172966707
        assert set(d["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"} and len(d["cpu_baseline"]["sample"]) <= 330
        assert d["config"]["nodes"] == 100000 and "model" not in d["config"] and d["parity"]["ok"] is True
        full = strict(open(tmp_path / "d.json").read())
        assert full["row_block_rehearsal"]["note"].startswith("r") and len(full["configs"]) == n_configs + 1 and d["detail"] == "d.json"
        assert "bench detail" in cap.err
        if pad == 1:
            assert [c["config"] for c in d["configs"]][:2] == ["configs[0]", "configs[1]"] and d["configs"][0]["variant"] == "B" and d["configs"][-1]["error"]
        else:
            assert d["parity"]["max_rel_err"] is None and "configs" not in d          # 41 summaries do not fit: dropped, the contract keys stay


def test_contract_line_of_a_multi_rank_run():
    d = dict(_detail(0), n_gpus=8, kernel_ms_per_rank=[3.1] * 8, phases_s=dict(khop_signature_build=0.01, allgather=0.002, exchange="ncclAllGather (RCCL)"))
    line = bench.compact_line(bench._strict(d), None)
    assert line["kernel_ms_per_rank"] == [3.1] * 8 and line["phases_s"]["exchange"].startswith("ncclAllGather")
    assert line["design_ceiling"]["speedup_at_most"] == 4.27 and bench.row_block_ceiling(2) == 1.33 and bench.row_block_ceiling(1) == 1.0
