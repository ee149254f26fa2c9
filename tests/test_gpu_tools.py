"""GPU: hsd_amd.tools (the reference's `tools` helpers on the hot path) against the golden
micro-vectors produced by tools.metrics in the reference."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_graph

pytestmark = pytest.mark.gpu


def test_calculate_distance_micro_vectors(engine):
    from hsd_amd import tools
    with open(os.path.join(GOLDEN, "metrics.json")) as f:
        M = json.load(f)
    for rec in M["w1"]:
        d = tools.metrics.calculate_distance(rec["p"], rec["q"], "wasserstein", engine=engine)
        assert abs(d - rec["aligned"]) <= 1e-14 + 1e-12 * abs(rec["aligned"]), (rec, d)
    for rec in M["kl"]:
        d = tools.calculate_distance(rec["p"], rec["q"], "kl", engine=engine)
        assert abs(d - rec["kl_via_calculate_distance"]) <= 1e-13
        d = tools.calculate_distance(rec["p"], rec["q"], "symmetric_kl", engine=engine)
        assert abs(d - rec["skl_via_calculate_distance"]) <= 1e-13
    k = M["known"]
    assert abs(tools.calculate_distance([], [.2, .5], "wasserstein", engine=engine) - k["w1_empty_vs_[.2,.5]"]) < 1e-15
    assert abs(tools.calculate_distance([.2, .3, .5], [.1, .6, .3], "kl", engine=engine) - k["kl_sorted"]) < 1e-15
    assert tools.calculate_distance([], [], "wasserstein", engine=engine) == 0.0
    E = M["errors"]
    with pytest.raises(TypeError, match="Need to specify a metric."):
        tools.calculate_distance([1.0], [1.0], None, engine=engine)
    with pytest.raises(NotImplementedError, match="cosine metric is not implemented."):
        tools.calculate_distance([1.0], [1.0], "cosine", engine=engine)
    with pytest.raises(ValueError) as ei:
        tools.calculate_distance([.2, .3], [.1, .6, .3], "kl", engine=engine)
    assert [type(ei.value).__name__, str(ei.value)] == E["kl_not_normalised"]


def test_get_hierarchical_representation(engine, capsys):
    from hsd_amd import tools
    R = np.load(os.path.join(GOLDEN, "rings.npz"))
    g = load_graph("karate")
    h = tools.get_hierarchical_representation(g, 3, engine=engine)
    assert "done, number of nodes: 34" in capsys.readouterr().out
    off, mem = R["karate/K3/offsets"], R["karate/K3/members"]
    for i, v in enumerate(g.nodes):
        assert len(h[v]) == 4 and h[v][0] == [v]
        for k in range(4):
            assert [g.node2idx[u] for u in h[v][k]] == list(mem[off[i * 4 + k]:off[i * 4 + k + 1]])


def test_two_models_share_one_engine(engine):
    """A model that finds the shared context holding another model's graph uploads its own again."""
    import hsd_amd
    from conftest import load_graph
    m1 = hsd_amd.HSD(load_graph("karate"), "karate", 1.0, 3, "wasserstein", signature="degree", engine=engine)
    d1 = m1.parallel_calculate_HSD()
    m2 = hsd_amd.HSD(load_graph("barbell"), "barbell", 1.0, 2, "wasserstein", engine=engine)
    w2 = m2.calculate_wavelets(0.5, approx=False)
    assert w2.shape == (m2.n_node, m2.n_node)
    assert np.array_equal(m1.parallel_calculate_HSD(), d1)               # m1 again: rebinds, same result
    assert np.array_equal(m2.calculate_wavelets(0.5, approx=False), w2)  # and m2 again
    assert m1.get_nodes_hierarchical_degree()[m1.nodes[0]][0] == 1


def test_ablation_hooks_are_ignored_by_the_shipped_library(engine):
    """HSD_DEBUG=1 (the pair kernels without their stores) and HSD_TOPK_DEBUG=1 (the sweep without its candidate epilogue: "the call then fails") are
    measurement switches of `make debug` builds; the library the package ships must not look at them."""
    from hsd_amd.graph import barabasi_albert
    g = barabasi_albert(3000, 4, 3)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    engine.pair_prepare(0, 1, 0, 2)
    os.environ["HSD_STREAM"] = "1"
    try:
        D = engine.pair_rows_host()
        idx, val = engine.pair_topk(5)
        os.environ["HSD_DEBUG"] = "3"
        os.environ["HSD_TOPK_DEBUG"] = "1"
        D2 = engine.pair_rows_host()
        idx2, val2 = engine.pair_topk(5)
    finally:
        for k in ("HSD_STREAM", "HSD_DEBUG", "HSD_TOPK_DEBUG"):
            os.environ.pop(k, None)
    assert D.max() > 0 and np.array_equal(D, D2) and np.array_equal(idx, idx2) and np.array_equal(val, val2)


def _check_contract_line(stdout, n_gpus, steps, warmup):
    """stdout of bench.py is the contract line and nothing else: compact (the driver reads a tail of stdout -- round 4's 20 KB line was lost), strict
    JSON, the contract's keys with the roofline / cpu_baseline / parity objects, value consistent with ms_per_step; the detail record holds the rest."""
    import json
    from conftest import REPO
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), lines[:3]
    assert len(lines[0].encode()) < 4096, len(lines[0])

    def no_constants(name):
        raise ValueError(f"{name} is not JSON")
    d = json.loads(lines[0], parse_constant=no_constants)
    detail = json.load(open(os.path.join(REPO, d["detail"])), parse_constant=no_constants)
    assert detail["value"] == d["value"] and detail["roofline"]["kernel_ms"] == d["roofline"]["kernel_ms"] and "phases_s" in detail
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md holds no published number for this metric
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True and d["data"].startswith("synthetic")
    assert "node-pairs" in d["metric"] and d["unit"] == "node-pairs/s" and "workload" in d["config"] and "model" not in d["config"]
    n = d["config"]["nodes"]
    assert abs(d["value"] - n * (n - 1) / 2.0 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.01 < r["frac"] < 1.0 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and isinstance(c["sample"], str)
    assert d["parity"]["ok"] is True and d["parity"]["tol"] == 1e-6
    return d, detail


def test_bench_line_contract(tmp_path):
    """bench.py as the driver runs it (a child process, flags --gpus/--steps/--warmup), on a small custom workload: ONE JSON line on stdout with
    the contract's keys -- metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
    config.workload, the roofline object (bound, achieved, peak, unit, frac, traffic) and the cpu_baseline object (value, unit, cores, kind, sample)
    -- and value consistent with ms_per_step."""
    import subprocess
    import sys
    from conftest import REPO
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--graph", "ba", "--nodes", "20000",
                        "--cpu-seconds", "1.5", "--configs", "none", "--detail", "bench_detail_contract_test.json"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    d, _ = _check_contract_line(p.stdout, 1, 3, 1)
    assert 0.05 < d["roofline"]["frac"] and d["config"]["parallelism"] == "row-block x1" and "design_ceiling" not in d


def test_bench_line_contract_two_ranks(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per GPU) rehearsed on this box's ONE GPU (HSD_COMM_DRYRUN=1: no RCCL
    communicator, the histograms travel through gloo): the same contract line from rank 0 -- now with the CPU baseline (computed on rank 0 before the
    rendezvous), both ranks' kernel times, the exchange named, the row blocks' design ceiling, and the k-nearest rows of every rank's block."""
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # (--configs 4: the branch `--gpus 8` takes by default -- config 4 on the same ranks, a fresh context -- rehearsed on a stand-in graph of 24 000 nodes)
    env = dict(os.environ, HSD_COMM_DRYRUN="1", HSA_ENABLE_IPC_MODE_LEGACY="0", HSD_BENCH_CONFIG4_NODES="24000")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--graph", "ba", "--nodes", "20000", "--cpu-seconds", "1.5",
                        "--configs", "4", "--detail", "bench_detail_contract_test2.json"], capture_output=True, text=True, timeout=900, cwd=str(tmp_path), env=env)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    d, detail = _check_contract_line(p.stdout, 2, 3, 1)
    assert d["scaling"] == "strong" and d["config"]["parallelism"].startswith("row-block x2") and "rehearsal" in d["data"]
    assert len(d["kernel_ms_per_rank"]) == 2 and all(v > 0 for v in d["kernel_ms_per_rank"])
    assert d["phases_s"]["exchange"].startswith("host (gloo)") and d["phases_s"]["allgather"] >= 0
    assert d["design_ceiling"]["speedup_at_most"] == 1.33
    assert d["fused_knn"]["exact"] is True and d["fused_knn"]["ms_end_to_end"] > 0 and len(detail["fused_knn"]["ms_per_rank"]) == 2
    assert d["parity"]["pairs_checked"] > 0                       # rank 0's rows of the CPU sample
    (c4,) = d["configs"]
    assert c4["config"] == "configs[3]" and "error" not in c4 and c4["variant"] == "B" and c4["value"] > 0
    (full4,) = detail["configs"]
    assert "REHEARSAL at N=24000" in full4["workload"] and [v["variant"] for v in full4["variants"]] == ["B", "A"]
    assert all(len(v["kernel_ms_per_rank"]) == 2 and v["fused_knn"]["exact"] for v in full4["variants"])


def test_single_node_structure_and_index_maps(engine, capsys):
    """tools.get_node_hierarchical_structure (tools/hierarchy.py:25-38) is one row of get_hierarchical_representation -- always maxHop + 1
    lists, empty ones once the component is exhausted -- and build_node_idx_map (tools/util.py:11-24) follows the node order of the matrices."""
    from hsd_amd import tools
    g = load_graph("karate")
    full = tools.get_hierarchical_representation(g, 3, engine=engine)
    capsys.readouterr()
    for v in (g.nodes[0], g.nodes[7], g.nodes[-1]):
        one = tools.get_node_hierarchical_structure(g, v, 3, engine=engine)
        assert one == full[v] and one[0] == [v] and len(one) == 4
    line = load_graph("line")
    far = tools.get_node_hierarchical_structure(line, line.nodes[0], line.n_node + 2, engine=engine)
    assert len(far) == line.n_node + 3 and sum(len(r) for r in far) == line.n_node and far[-1] == []
    idx2node, node2idx = tools.build_node_idx_map(g)
    assert [idx2node[i] for i in range(g.n_node)] == g.nodes and all(node2idx[v] == i for i, v in enumerate(g.nodes))
    assert tools.scale_boundary(1.0, 4.0) == (-np.log(0.95) / 2.0, -np.log(0.85) / 2.0)


def test_operand_arena_reuses_its_blocks(engine):
    """hsd_pair_prepare takes its operand sets, the k-nearest sweep's permuted copies and its temporaries from the context's grow-only arena (csrc/context.hip,
    hsd_pool_*): cycling through three distances and their k-nearest setups again and again must not take more device memory than the first cycle did, and
    the results stay what they were."""
    import ctypes
    from hsd_amd.graph import barabasi_albert
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value
    g = barabasi_albert(12000, 5, 21)
    engine.set_graph(g.rowptr, g.col)
    engine.signature_build_degree(3)
    first = {}
    os.environ["HSD_STREAM"] = "1"
    os.environ["HSD_TOPK_PRUNE"] = "1"
    try:
        def cycle(check):
            for name, mode in (("B", (0, 1, 0, 2)), ("A", (0, 0, 0, 3)), ("KL", (2, 0, 0, 3, 0.5, False))):
                engine.pair_prepare(*mode)
                row = engine.pair_rows_host(100, 101)[0]
                idx, val = engine.pair_topk(7) if name != "KL" else (None, None)
                blk = engine.pair_topk(7, 2048, 6000) if name != "KL" else None
                if not check:
                    first[name] = (row, idx, val, blk)
                else:
                    assert np.array_equal(row, first[name][0])
                    if idx is not None:
                        assert np.array_equal(idx, first[name][1]) and np.array_equal(val, first[name][2])
                        assert np.array_equal(blk[0], first[name][3][0]) and np.array_equal(blk[1], first[name][3][1])
        cycle(False)
        cycle(True)
        engine.sync()
        after_two = free_bytes()
        for _ in range(6):
            cycle(True)
        engine.sync()
        assert free_bytes() >= after_two - (64 << 20), (after_two, free_bytes())
    finally:
        os.environ.pop("HSD_STREAM", None)
        os.environ.pop("HSD_TOPK_PRUNE", None)
