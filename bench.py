#!/usr/bin/env python3
"""Headline benchmark: node-pairs/s on the N x N structural-distance matrix (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node P ... bench.py --gpus P --steps K --warmup W

Headline workload (config.workload) = the north-star target, SURVEY.md 8(d) "target": synthetic
Barabasi-Albert N = 100 000, m = 5, seed 42, K = 3, Wasserstein on per-hop degree signatures, the full
symmetric fp64 matrix (80 GB) resident in HBM.  It is the largest single-GPU configuration BASELINE.json
names, and the one its roofline target is quoted on.

  primary    Variant B -- HSD.parallel_calculate_HSD, THE CALL main.py:26 MAKES (model/HSD.py:140-161;
             BASELINE.json's configs are "K=3 Wasserstein via main.py"): tools.metrics.calculate_distance =
             zero-pad alignment + scipy, layers 0..K-1; exact integer arithmetic on the device
  secondary  Variant A -- HSD.calculate_structural_distance (model/HSD.py:98-114), the reference's serial
             self-contained form: scipy wasserstein_distance on the raw ragged ring samples, layers 0..K, fp64
             (2 x 721 fp64 instructions per pair: bound by the vector ALU at 0.05 of the HBM roofline)
  (rounds 1-2 printed A first; --variant A restores that.  Both variants carry the same keys.)

A "step" is one pass of the all-pairs kernel over the whole matrix (this rank's row block of it) with
the signature operands already resident in HBM.  With --gpus P > 1 the SAME matrix is row-block
partitioned over the ranks ("scaling": "strong"; SURVEY 8(e), model/HSD.py:118-137 is the pool it
replaces) after ONE RCCL all-gather of the histogram tensor (outside the timed region, reported under
phases_s); at P = 8 BASELINE's config 4 (N = 200 000) is run the same way and reported under "configs".
--weak keeps round 1's weak-scaling mode (ER N = 10 000 sqrt(P)).

On one GPU the line also carries, under "configs", BASELINE.json's other single-GPU configurations
measured in the same run with the same keys (roofline, cpu_baseline, parity):
  config 2  ER N = 10 000 p = 0.001, K = 3, Wasserstein (both variants + the heat-coefficient path)
  config 3  BA N = 50 000 m = 5, K = 5, KL divergence
  config 5  MultiHSD heat-kernel signatures, 8 scales, BA N = 50 000, K = 3

Keys next to the contract's:
  roofline      HBM roofline of the all-pairs kernel from HIP-event timing of the K launches on the stream
                they run on; valu_frac = the same launches priced against the vector-ALU issue rate (these
                kernels are arithmetic-bound beyond a few dozen operand columns)
  cpu_baseline  the reference's own loop (scipy.stats.wasserstein_distance per layer per pair) timed on this
                host on a bounded sample of pairs of the same matrix; the same pairs are the parity sample
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
# Vector-ALU issue peaks in lane-instructions/s.  fp64 add/fma and the 3-operand integer SADs (v_sad_u32 / v_sad_u16)
# all issue at 16 lanes per clock per SIMD on gfx950 (measured: 4.1 cycles per wave-instruction, tools_dev/ubench_valu.hip,
# profiles/r02_ubench_valu.txt): 256 CU x 4 SIMD x 16 lanes x 2.4 GHz.
FP64_VALU_PEAK = 39.3e12
SAD_VALU_PEAK = 39.3e12

WORKLOADS = {
    "target": dict(graph="ba", nodes=100000, m=5, hop=3, metric="wasserstein",
                   name="north-star target (SURVEY 8(d)): barabasi_albert N=100000 m=5 seed=42, K=3, Wasserstein"),
    "config2": dict(graph="er", nodes=10000, p=0.001, hop=3, metric="wasserstein",
                    name="BASELINE configs[1]: erdos_renyi N=10000 p=0.001 seed=42, K=3, Wasserstein"),
    "config3": dict(graph="ba", nodes=50000, m=5, hop=5, metric="kl",
                    name="BASELINE configs[2]: barabasi_albert N=50000 m=5 seed=42, K=5, KL divergence"),
    "config4": dict(graph="ba", nodes=200000, m=5, hop=3, metric="wasserstein",
                    name="BASELINE configs[3]: scale-free (barabasi_albert m=5 seed=42) N=200000, K=3, Wasserstein, row blocks"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="target", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", default="B", choices=["A", "B"],
                    help="primary variant. A: scipy on raw ring samples, layers 0..K (model/HSD.py:98-114); "
                         "B: tools.metrics.calculate_distance zero-pad alignment, layers 0..K-1 (model/HSD.py:140-161)")
    ap.add_argument("--configs", default=None,
                    help="comma list of the other BASELINE configs measured beside the headline (2,3,5 on one GPU; 4: on eight GPUs the "
                         "row-block matrix, on one GPU the k-nearest consumer only; 'extra': the headline kernel on an Erdos-Renyi graph of the target's size); "
                         "'none' to skip.  Default: extra,2,3,4,5 at --gpus 1, 4 at --gpus 8, none otherwise")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the scipy baseline sample of the headline (0 = skip)")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip what rides beside the timed steps (k-nearest consumer, row-block rehearsal): a profile of this "
                    "run then holds the matrix launches of the all-pairs kernels and nothing else of the same instantiation (tools_dev/prof.sh)")
    ap.add_argument("--detail", default=None, help="file name of the detail record (default bench_detail.json, next to bench.py and under gpurun_out/)")
    ap.add_argument("--weak", action="store_true", help="round-1 weak-scaling mode: ER N = nodes * sqrt(P), per-GPU output fixed")
    # overrides (measurement scripts): a custom graph instead of a named workload
    ap.add_argument("--graph", default=None, choices=["er", "ba"])
    ap.add_argument("--nodes", type=int, default=None)
    ap.add_argument("--mean-degree", type=float, default=10.0)
    ap.add_argument("--hop", type=int, default=None)
    ap.add_argument("--metric", default=None, choices=["wasserstein", "kl", "symmetric_kl"])
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def make_graph(spec):
    import hsd_amd
    n = spec["nodes"]
    if spec["graph"] == "er":
        p = spec.get("p", spec.get("mean_degree", 10.0) / n)
        g = hsd_amd.erdos_renyi(n, p, seed=42)
        if spec.get("connect"):
            # a scaled-up G(n, 10/n) has ~n e^-10 isolated nodes; their empty rings make scipy (and Variant A)
            # raise, so small components are tied to the giant one by one edge each
            from hsd_amd.graph import connect_components
            g = connect_components(g, seed=42)
        return g
    return hsd_amd.barabasi_albert(n, spec.get("m", 5), seed=42)


def variant_params(variant, hop):
    return (0, 0, hop) if variant == "A" else (1, 0, hop - 1)      # (align, layer_lo, layer_hi)


def variant_call(variant):
    return ("HSD.calculate_structural_distance (model/HSD.py:98-114): scipy on raw ring samples" if variant == "A" else
            "HSD.parallel_calculate_HSD, the call main.py:26 makes (model/HSD.py:140-161): zero-pad alignment")


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def sample_nodes(n, count, seed=7):
    """Nodes the CPU baseline / parity sample is taken on: spread over the whole node order (the first rows of a
    preferential-attachment graph are its hubs -- not a representative sample)."""
    rng = np.random.default_rng(seed)
    return np.sort(rng.choice(n, size=min(count, n), replace=False)).astype(np.int64)


# ------------------------------------------------------------------------------------------------
# CPU baseline: the reference's loop on the same signatures, on a bounded sample of pairs
# ------------------------------------------------------------------------------------------------
def _pair_scipy(nr, a, b, align, lo, hi, wd):
    d = 0.0
    for k in range(lo, hi + 1):
        p, q = nr.sample(a, k), nr.sample(b, k)
        if align:                                   # tools/metrics.py:18-36: zero-pad the shorter list, sort both
            L = max(len(p), len(q))
            if L == 0:
                continue                            # tools/metrics.py:170-171: both empty -> 0.0
            p = np.sort(np.concatenate([p, np.zeros(L - len(p))]))
            q = np.sort(np.concatenate([q, np.zeros(L - len(q))]))
        d += wd(p, q)
    return d


def scipy_baseline(g, hop, variant, budget_s, n_sample=400):
    """for i, for j > i over a sample of nodes: sum over layers of scipy.stats.wasserstein_distance (Variant A,
    model/HSD.py:103-112) or tools.metrics.calculate_distance's zero-pad + scipy (Variant B, tools/metrics.py:18-36,
    186-190).  Returns dict(rate, pairs, seconds, sample {(i, j): d}, nodes)."""
    from scipy.stats import wasserstein_distance
    from oracle import c_oracle as corc
    nodes = sample_nodes(g.n_node, n_sample)
    nr = corc.NodeRingSamples(g.rowptr, g.col, hop, g.degree.astype(np.float64), nodes)
    align, lo, hi = variant_params(variant, hop)
    sample = {}
    t0 = time.perf_counter()
    done = False
    # anti-diagonal order over the sampled nodes, so that a run cut short still mixes early and late nodes
    m = len(nodes)
    for s in range(1, 2 * m):
        for ia in range(max(0, s - m + 1), (s + 1) // 2):
            ib = s - ia
            if ib >= m or ia >= ib:
                continue
            a, b = int(nodes[ia]), int(nodes[ib])
            sample[(a, b)] = _pair_scipy(nr, a, b, align, lo, hi, wasserstein_distance)
            if len(sample) % 64 == 0 and time.perf_counter() - t0 > budget_s:
                done = True
                break
        if done:
            break
    dt = time.perf_counter() - t0
    return dict(rate=len(sample) / dt, pairs=len(sample), seconds=dt, sample=sample, nodes=nodes)


_POOL = {}


def _pool_init(rowptr, col, hop, degree, nodes, variant):
    from oracle import c_oracle as corc
    _POOL["nr"] = corc.NodeRingSamples(rowptr, col, hop, degree, nodes)
    _POOL["nodes"], _POOL["variant"], _POOL["hop"] = nodes, variant, hop


def _pool_row(task):
    """One task per source row, as model/HSD.py:122-124 submits them: sampled node ia against the later sampled nodes."""
    from scipy.stats import wasserstein_distance
    ia, count = task
    nr, nodes = _POOL["nr"], _POOL["nodes"]
    align, lo, hi = variant_params(_POOL["variant"], _POOL["hop"])
    done = 0
    for ib in range(ia + 1, min(len(nodes), ia + 1 + count)):
        _pair_scipy(nr, int(nodes[ia]), int(nodes[ib]), align, lo, hi, wasserstein_distance)
        done += 1
    return done


def scipy_pool_baseline(g, hop, variant, budget_s, single_rate):
    """The same loop fanned out over a process pool of os.cpu_count() workers, one task per source row
    (model/HSD.py:118-130).  Tasks are short (about a quarter second of one core each) and the pool is stopped at the
    budget: the rate is pairs finished / elapsed, whatever the per-core rate under full load turns out to be."""
    import multiprocessing as mp
    workers = min(os.cpu_count() or 1, 64)
    per_task = max(8, int(single_rate * 0.25))
    n_tasks = workers * 256
    nodes = sample_nodes(g.n_node, min(g.n_node, max(640, per_task + 64)), seed=11)
    tasks = [(i % (len(nodes) - per_task - 1), per_task) for i in range(n_tasks)]
    ctx = mp.get_context("fork")
    pool = ctx.Pool(workers, initializer=_pool_init, initargs=(g.rowptr, g.col, hop, g.degree.astype(np.float64), nodes, variant))
    done = 0
    try:
        pool.map(_pool_row, [(0, 0)] * (4 * workers), chunksize=1)      # every worker has built its ring samples before the clock starts
        t0 = time.perf_counter()
        for k in pool.imap_unordered(_pool_row, tasks, chunksize=1):
            done += k
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    finally:
        pool.terminate()
        pool.join()
    return dict(rate=done / dt, pairs=done, workers=workers, seconds=dt)


def parity_against(mat, r0, sample, tol=1e-6):
    """GPU matrix entries against the CPU sample {(i, j): d}: rows are fetched once each."""
    rows = {}
    worst = 0.0
    checked = 0
    r1 = r0 + mat.rows
    for (i, j), d in sample.items():
        if not (r0 <= i < r1):
            continue
        if i not in rows:
            rows[i] = mat.rows_to_host(i - r0, i - r0 + 1)[0]
        worst = max(worst, abs(rows[i][j] - d) / max(abs(d), 1e-12))
        checked += 1
    return dict(pairs_checked=checked, max_rel_err=worst, tol=tol, ok=bool(checked > 0 and worst <= tol))


# ------------------------------------------------------------------------------------------------
# timing + roofline of the all-pairs kernel
# ------------------------------------------------------------------------------------------------
def time_pair(eng, mat, r0, r1, steps, warmup, barrier):
    for _ in range(warmup):
        eng.pair_rows_device(r0, r1, mat)
    eng.sync()
    barrier()
    t0 = time.perf_counter()
    eng.timer_start()
    for _ in range(steps):
        eng.pair_rows_device(r0, r1, mat)
    ev_ms = eng.timer_stop()            # HIP events on the stream the kernels run on; synchronises
    eng.sync()
    barrier()
    wall = time.perf_counter() - t0
    return wall, ev_ms


def load_traffic(key):
    """HBM bytes per launch from the PMC passes of this command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, collected in
    their own runs as MI355X_MICROARCH.md prescribes; summaries under profiles/).  None if that workload was not profiled."""
    for name in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                t = json.load(f).get(key)
            if t is not None:
                return dict(t, source=f"profiles/{name}") if isinstance(t, dict) else t
        except OSError:
            pass
    return None


def traffic_fields(key):
    """roofline.traffic = HBM bytes per launch (a number, or None); roofline.traffic_detail = read / write split, kernels, source."""
    t = load_traffic(key) if key else None
    if isinstance(t, dict):
        return t.get("bytes"), t
    return t, None


def fill_reference(eng, mat, kernel_ms):
    """What this GPU's memory system takes for the output alone: hipMemsetAsync over the same row block (80 GB at N = 100k), timed
    the same way right after the kernel.  `peak` stays the 8 TB/s of MI355X_MICROARCH.md; this is the measured ceiling of a pure
    write stream of that size beside it (5.9-6.0 TB/s for 80 GB, 6.6 TB/s for 800 MB)."""
    import ctypes as C
    nbytes = mat.rows * mat.ld * 8
    try:
        eng._call("hsd_memset", C.c_void_p(mat.ptr), 0, C.c_size_t(nbytes))
        eng.sync()
        reps = 3
        eng.timer_start()
        for _ in range(reps):
            eng._call("hsd_memset", C.c_void_p(mat.ptr), 0, C.c_size_t(nbytes))
        ms = eng.timer_stop() / reps
    except Exception as e:          # noqa: BLE001  (a reference figure: never fails the bench line)
        return dict(error=str(e)[:80])
    return dict(call="hipMemsetAsync over the output block", bytes=nbytes, ms=round(ms, 4), GBps=round(nbytes / ms / 1e6, 1),
                kernel_over_fill=round(kernel_ms / ms, 3) if ms > 0 else None)


def kernel_roofline(n, rows, res, steps, metric, traffic_key=None):
    """bytes_alg = 8 * rows * N (each fp64 output written once) + operand matrix read once; beside it
    the vector instructions the launch needs at minimum (fp64 W1: 2 per column pair, integer SAD: 1 per word pair,
    KL: 1 FMA) against the issue peak."""
    bytes_alg = 8.0 * rows * n + float(res["ebytes"]) * n * res["cols"]
    t_kernel = res["ev_ms"] * 1e-3 / steps
    achieved = bytes_alg / t_kernel / 1e9
    # vector instructions of one launch: (operand words x pairs) the kernel itself counted (hsd_pair_work; whole tiles, and
    # fewer words than `cols` where a tile pair took the shorter rank form of a zero-pad layer) x instructions per word pair
    sq = rows * (rows + 1) / 2.0 + rows * (n - rows)            # pairs this rank owes (diagonal square once)
    work = res.get("work") or sq * res["cols"]
    if res["ebytes"] == 4:
        instr, peak = work * 1.0, SAD_VALU_PEAK                  # one v_sad_u32 / v_sad_u16 per word pair
    else:
        instr, peak = work * (2.0 if metric == "wasserstein" else 1.0), FP64_VALU_PEAK   # sub + add|.| per column pair; KL: one FMA
    traffic, detail = traffic_fields(traffic_key)
    return dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                traffic=traffic, traffic_over_bytes_alg=round(traffic / bytes_alg, 3) if traffic else None, traffic_detail=detail,
                kernel_ms=round(t_kernel * 1e3, 4), bytes_alg=bytes_alg,
                valu_frac=round(instr / t_kernel / peak, 4), valu_peak_lane_instr_per_s=peak,
                limiter="vector ALU issue" if instr / peak > bytes_alg / (HBM_PEAK_GBS * 1e9) else "HBM write stream",
                operand_columns=res["cols"], columns_evaluated_per_pair=round(work / sq, 2))


def dtype_of(res, n_bins):
    """Arithmetic type of the kernel: fp64, uint32 SAD, or two uint16 columns per 32-bit word (v_sad_u16)."""
    if res["ebytes"] != 4:
        return "f64"
    point = 1 if res["lo"] == 0 else 0
    packed = (res["cols"] - point) < (res["hi"] - res["lo"] + 1 - point) * n_bins
    return "u16" if packed else "u32"


class Ranks:
    """Control plane of a multi-rank run (gloo): barriers, max over ranks, error agreement."""

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, x):
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def gather(self, x):
        if self.dist is None:
            return [x]
        import torch
        t = torch.zeros(self.world, dtype=torch.float64)
        t[self.rank] = x
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t]

    def any(self, flag):
        return bool(self.max(1.0 if flag else 0.0))


EXTRAS = True           # --no-extras clears it


def run_degree_workload(eng, rk, g, spec, variants, steps, warmup, cpu=None, traffic_prefix=None, dryrun=False):
    """k-hop sweep + histograms (this rank's sources), the exchange, then every variant: prepare, first call, timed
    steps.  Returns dict(variants={name: result}, phases, sweep, n_bins)."""
    from hsd_amd import _lib as L
    from hsd_amd import distributed as hdist
    n, hop, metric = g.n_node, spec["hop"], spec["metric"]
    t0 = time.perf_counter()
    eng.set_graph(g.rowptr, g.col)
    eng.sync()
    t_graph = time.perf_counter() - t0
    # (a communicator that cannot be made -- a node without working RCCL -- must not lose the run: the counts then travel through the
    # host, outside the timed region either way, and the line says so under phases_s.exchange)
    r0, r1 = hdist.init_comm(eng, rk.dist, rk.rank, rk.world, host_fallback=True) if rk.world > 1 else (0, n)
    assert (r0, r1) == hdist.shard_rows(n, rk.world, rk.rank)
    eng.sync()
    t0 = time.perf_counter()
    eng.signature_build_degree(hop, r0, r1)
    eng.sync()
    t_sig = time.perf_counter() - t0
    # the sweep once more under the device stopwatch (buffers exist now), with its algorithmic bytes:
    # sum over sources of sum_{v in ball_{K-1}} (8 + 4 deg v)  +  the histogram rows written
    eng.timer_start()
    eng.signature_build_degree(hop, r0, r1)
    sweep_ms = eng.timer_stop()
    sweep = None
    n_bins = int(eng.signature_dims()[1])
    if rk.rank == 0 and r1 > r0:
        sizes = eng.khop_sizes(hop, r0, min(r1, r0 + 2000))                # sample of sources for the ball sizes
        inner = float(sizes[:, :hop].sum(axis=1).mean())                   # |ball_{K-1}| per source
        bytes_sweep = (r1 - r0) * inner * (8.0 + 4.0 * float(g.degree.mean())) + 4.0 * (r1 - r0) * (hop + 1) * n_bins
        bit_parallel = n >= 8192 and hop >= 2 and r1 - r0 >= 2048 and os.environ.get("HSD_KHOP_MS", "1") != "0"    # the library's own rule (csrc/khop.hip)
        if bit_parallel:
            # the bit-parallel sweep does not expand one BFS per source: its bytes are the frontier rows it gathers and the
            # planes it streams, (hop - 1) pulls per batch of 4096 sources, 512 B per CSR entry and 4 x 512 B per vertex
            moved = (hop - 1) * (float(g.rowptr[-1]) + 4.0 * n) * 512.0 * ((r1 - r0 + 4095) // 4096)
            # priced by the bytes of THIS algorithm (the frontier planes it streams).  SURVEY 8(d)'s per-source figure (one BFS expansion
            # per source) describes a different algorithm: the bit-parallel sweep advances 4096 sources per gathered row, so that figure
            # divided by the time can exceed the HBM peak -- kept only as a labelled equivalent, never as a fraction
            sec = sweep_ms * 1e-3
            sweep = {"kernel": "ms_pull / ms_hist kernels (bit-parallel k-hop sweep, 4096 sources per batch, + per-hop histograms)",
                     "ms": round(sweep_ms, 4), "bytes_alg": moved,
                     "bytes_alg_is": "(hop - 1) x (nnz + 4 n) x 512 B per batch of 4096 sources: the frontier planes gathered and streamed",
                     "achieved_GBps": round(moved / sec / 1e9, 1), "frac_of_hbm": round(moved / sec / 1e9 / HBM_PEAK_GBS, 4),
                     "per_source_bfs_equivalent": {"bytes": bytes_sweep, "GBps": round(bytes_sweep / sec / 1e9, 1),
                                                   "note": "SURVEY 8(d): sum over sources of sum_{v in ball_{K-1}} (8 + 4 deg v) + histogram rows -- the bytes "
                                                           "one BFS per source would move; not this kernel's traffic, so no fraction of the HBM peak is quoted"},
                     "bound": "row gathers out of the Infinity Cache (about 8.6 TB/s, MI355X_MICROARCH.md)"}
        else:
            sweep = {"kernel": "khop_kernel (per-source k-hop sweep + per-hop histograms)", "ms": round(sweep_ms, 4), "bytes_alg": bytes_sweep,
                     "achieved_GBps": round(bytes_sweep / (sweep_ms * 1e-3) / 1e9, 1),
                     "frac_of_hbm": round(bytes_sweep / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "bound": "LDS atomics / latency (the graph is L2-resident), not HBM"}
    t0 = time.perf_counter()
    hdist.allgather_signature(eng, rk.dist, rk.rank, rk.world, via="host" if dryrun else "rccl")   # ONE ncclAllGather over xGMI
    eng.sync()
    t_gather = time.perf_counter() - t0

    metric_code = {"wasserstein": L.METRIC_WASSERSTEIN, "kl": L.METRIC_KL, "symmetric_kl": L.METRIC_SYMMETRIC_KL}[metric]
    out = {}
    for variant in variants:
        align, lo, hi = variant_params(variant, hop) if metric == "wasserstein" else (0, 0, hop)
        err, res, mat = None, None, None
        try:
            t0 = time.perf_counter()
            eng.pair_prepare(metric_code, align, lo, hi, 0.5, False)
            eng.sync()
            t_prep = time.perf_counter() - t0
            cols, ebytes = eng.pair_operand_dims()
            # this rank's rows of THIS distance: contiguous row blocks of equal planned cost (hsd_pair_shard; the sweep's blocks above
            # have equal heights) -- the rank that holds a scale-free graph's hub rows would otherwise be ~15 % behind the others
            p0, p1 = eng.pair_shard(rk.world, rk.rank) if rk.world > 1 else (0, n)
            t0 = time.perf_counter()
            mat = eng.pair_rows_device(p0, p1)          # allocates this rank's row block (80 GB at N = 100k) and fills it
            eng.sync()
            t_first = time.perf_counter() - t0
            # a second prepare + matrix into the existing block: what one more distance matrix costs once the
            # context is warm (buffers, code objects, tile lists exist)
            t0 = time.perf_counter()
            eng.pair_prepare(metric_code, align, lo, hi, 0.5, False)
            eng.pair_rows_device(p0, p1, mat)
            eng.sync()
            t_warm = time.perf_counter() - t0
        except ValueError as e:             # Variant A raises on an empty ring, as scipy does
            err = e
        # an exception on SOME ranks must not leave the others waiting in the barriers of the timed loop
        if rk.any(err is not None):
            if mat is not None:
                mat.free()
            out[variant] = dict(variant=variant, error=str(err) if err else "Distribution can't be empty. (raised on another rank)")
            continue
        wall, ev_ms = time_pair(eng, mat, p0, p1, steps, warmup, rk.barrier)
        work = eng.pair_work()                          # of the last launch
        res = dict(work=work, variant=variant, wall=rk.max(wall), wall_rank=wall, ev_ms=ev_ms, cols=cols, ebytes=ebytes, align=align, lo=lo, hi=hi,
                   t_prep=t_prep, t_first=t_first, t_warm=t_warm, rows=(p0, p1))
        key = f"{traffic_prefix}_{variant}" if traffic_prefix and rk.world == 1 else None
        res["roofline"] = kernel_roofline(n, p1 - p0, res, steps, metric, key)
        res["pairs_per_s"] = n * (n - 1) / 2.0 * steps / res["wall"]
        res["ms_per_step"] = res["wall"] * 1e3 / steps
        res["dtype"] = dtype_of(res, n_bins)
        res["kernel_ms_per_rank"] = [round(v, 4) for v in rk.gather(ev_ms / steps)]
        if cpu and variant in cpu:
            res["parity"] = parity_against(mat, p0, cpu[variant]["sample"])
        st = mat.stats()
        res["matrix_check"] = {k: st[k] for k in ("sum", "max_abs_diag", "max_asym")}
        if rk.world > 1 and metric == "wasserstein" and EXTRAS:
            res["fused_knn"] = fused_knn_rank(eng, rk, mat, n, p0, p1, 20)
        if rk.world == 1 and metric == "wasserstein" and EXTRAS:
            res["fused_knn"] = fused_knn(eng, mat, n, 20)
            if n >= 50000:
                res["row_block_rehearsal"] = row_block_rehearsal(eng, mat, n, res["roofline"]["kernel_ms"], heavy=(variant == "A"))
        # (last: it clears the block the checks above read)
        res["roofline"]["fill_same_bytes"] = fill_reference(eng, mat, res["roofline"]["kernel_ms"])
        mat.free()
        out[variant] = res
    return dict(variants=out, sweep=sweep, n_bins=n_bins, rows=(r0, r1),
                phases=dict(set_graph=round(t_graph, 4), khop_signature_build=round(t_sig, 4), allgather=round(t_gather, 4),
                            exchange=("none (one GPU)" if rk.world == 1 else
                                      "ncclAllGather (RCCL)" if getattr(eng, "exchange", "rccl") == "rccl" and not dryrun else
                                      "host (gloo): " + (getattr(eng, "exchange_reason", "") or "HSD_COMM_DRYRUN rehearsal"))))


def cpu_entry(c, variant, hop, what):
    loop = ("model/HSD.py:140-161 (_calculate_worker): tools.metrics.calculate_distance = zero-pad + sort + "
            "scipy.stats.wasserstein_distance per layer" if variant == "B" else
            "model/HSD.py:103-112 calling scipy.stats.wasserstein_distance per layer")
    return dict(value=round(c["rate"], 1), unit="node-pairs/s", cores=1, kind="port",
                sample=f"{c['pairs']} pairs among {len(c['nodes'])} nodes spread over the node order of {what}, {c['seconds']:.1f} s; "
                       f"loop of {loop} on the same degree signatures (variant {variant}); host: {cpu_model()}, {os.cpu_count()} cores")


def fused_knn(eng, mat, n, k):
    """The consumer of the matrix in the reference is a k-nearest-neighbour evaluation (main.py:29): hsd_pair_topk gives
    every row's k nearest without the matrix.  Timed end to end (host outputs), checked against rows
    of the resident matrix: same distances, ties by index."""
    if n <= k + 1:
        return None
    eng.sync()
    t0 = time.perf_counter()
    eng.pair_topk(k)                                   # first call: scratch allocation; raw fp64 W1: + the one-time node order / permuted operands
    dt_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx, val = eng.pair_topk(k)
    dt = time.perf_counter() - t0
    ok = True
    for r in (0, n // 3, n - 1):
        d = mat.rows_to_host(r, r + 1)[0].copy()
        d[r] = np.inf
        order = np.lexsort((np.arange(n), d))[:k]
        ok = ok and bool(np.array_equal(order, idx[r]) and np.array_equal(d[order], val[r]))
    extra = {}
    if os.environ.get("HSD_TOPK_PRUNE") is None:
        # the same call with the pruning switched off (every tile pair evaluated): what the pruned sweep on node-permuted operands saves, and that both give
        # the same lists.  (Default: pruned for the raw fp64 distance where the persistent kernel runs and for the zero-pad distance from ~50k nodes on.)
        os.environ["HSD_TOPK_PRUNE"] = "0"
        try:
            eng.pair_topk(k)
            eng.sync()
            t0 = time.perf_counter()
            uidx, uval = eng.pair_topk(k)
            u_dt = time.perf_counter() - t0
            extra["unpruned_sweep"] = dict(env="HSD_TOPK_PRUNE=0", ms_end_to_end=round(u_dt * 1e3, 3),
                                           identical_to_the_default_call=bool(np.array_equal(uidx, idx) and np.array_equal(uval, val)))
        except Exception as e:              # noqa: BLE001  (an extra figure: never fails the bench line)
            extra["unpruned_sweep"] = dict(error=str(e)[:100])
        finally:
            os.environ.pop("HSD_TOPK_PRUNE", None)
    return dict(extra, call="hsd_pair_topk", k=k, ms_end_to_end=round(dt * 1e3, 3), ms_first_call=round(dt_first * 1e3, 3), rows_checked_against_matrix=3, exact=ok,
                note="k nearest of every row without forming the matrix; time includes the results' copy to host memory (pruned sweep on node-permuted "
                     "operands where it applies: its one-time setup is in ms_first_call)")


def fused_knn_rank(eng, rk, mat, n, r0, r1, k):
    """The k-nearest consumer on P ranks: every rank asks for the k nearest of ITS rows, hsd_pair_topk(k, r0, r1) -- the pruned sweep of a row
    block (the rank's rows as positions of their own behind the graph's node order; round 5) where it applies.  Timed end to end per rank (host
    outputs), the slowest rank is the figure; rows checked against this rank's resident block of the matrix."""
    if n <= k + 1 or r1 <= r0:
        return None
    try:
        eng.sync()
        rk.barrier()
        t0 = time.perf_counter()
        eng.pair_topk(k, r0, r1)                          # first call: scratch + the one-time setup of the pruned sweep
        dt_first = time.perf_counter() - t0
        rk.barrier()
        t0 = time.perf_counter()
        idx, val = eng.pair_topk(k, r0, r1)
        dt = time.perf_counter() - t0
        ok = True
        for r in sorted({r0, (r0 + r1) // 2, r1 - 1}):
            d = mat.rows_to_host(r - r0, r - r0 + 1)[0].copy()
            d[r] = np.inf
            order = np.lexsort((np.arange(n), d))[:k]
            ok = ok and bool(np.array_equal(order, idx[r - r0]) and np.array_equal(d[order], val[r - r0]))
        err = None
    except Exception as e:                  # noqa: BLE001  (an extra figure: never fails the bench line; the ranks still meet below)
        dt = dt_first = 0.0
        ok, err = False, f"{type(e).__name__}: {e}"[:120]
    per = rk.gather(dt * 1e3)
    bad = rk.any(not ok)
    out = dict(call="hsd_pair_topk(k, r0, r1) on every rank", k=k, ms_end_to_end=round(max(per), 3), ms_per_rank=[round(v, 3) for v in per],
               ms_first_call=round(rk.max(dt_first * 1e3), 3), rows_checked_against_matrix=3 * rk.world, exact=not bad,
               value=n * (n - 1) / 2.0 / (max(per) * 1e-3) if max(per) > 0 else None, unit="node-pairs/s")
    if err:
        out["error"] = err
    return out


def row_block_rehearsal(eng, mat, n, kernel_ms, heavy):
    """What a rank of `bench.py --gpus P` would compute, measured on THIS GPU: for P = 2, 4, 8 every rank's row block of the same matrix
    (hsd_pair_shard: contiguous blocks of equal planned cost; one persistent launch per block) timed alone, one after the other.  The
    slowest block is what P identical GPUs would take before any communication or launch skew -- NOT a multi-GPU measurement (the driver's
    SCALE run is); it is here because a single-GPU box is the only hardware this build ever sees.  Rows 0 and the last row of every block
    are compared with the resident one-GPU matrix, bit for bit."""
    from hsd_amd.engine import DeviceMatrix
    out = {"note": "each rank's row block of the SAME matrix run alone on this one GPU (no communication, no launch skew in it): the time of the "
                   "slowest block is the floor of `bench.py --gpus P`; not a multi-GPU measurement", "one_gpu_ms": round(kernel_ms, 3), "by_P": {}}
    ld = (n + 15) // 16 * 16
    reps = 1 if heavy else 3
    for P in (2, 4, 8):
        per, same = [], True
        for r in range(P):
            r0, r1 = eng.pair_shard(P, r)
            if r1 <= r0:
                per.append(0.0)
                continue
            blk = DeviceMatrix(eng, r1 - r0, n, ld, r0)
            try:
                eng.pair_rows_device(r0, r1, blk)
                eng.timer_start()
                for _ in range(reps):
                    eng.pair_rows_device(r0, r1, blk)
                per.append(eng.timer_stop() / reps)
                for row in (r0, r1 - 1):
                    same = same and bool(np.array_equal(blk.rows_to_host(row - r0, row - r0 + 1), mat.rows_to_host(row, row + 1)))
            finally:
                blk.free()
        out["by_P"][str(P)] = dict(per_rank_ms=[round(v, 3) for v in per], rows_first_rank=list(eng.pair_shard(P, 0)),
                                   slowest_ms=round(max(per), 3), speedup_of_slowest_rank=round(kernel_ms / max(per), 2),
                                   rows_identical_to_one_gpu_matrix=same)
    return out


def variant_summary(res):
    if "error" in res:
        return dict(variant=res["variant"], error=res["error"])
    return dict(variant=res["variant"], call=variant_call(res["variant"]), value=res["pairs_per_s"], unit="node-pairs/s",
                ms_per_step=res["ms_per_step"], dtype=res["dtype"], roofline=res["roofline"], parity=res.get("parity"),
                phases_s=dict(operand_prepare_first=round(res["t_prep"], 4), first_matrix_incl_allocation=round(res["t_first"], 4),
                              warm_prepare_plus_matrix=round(res["t_warm"], 4)),
                matrix_check=res["matrix_check"], fused_knn=res.get("fused_knn"), row_block_rehearsal=res.get("row_block_rehearsal"))


# ------------------------------------------------------------------------------------------------
# the other BASELINE configs (one GPU)
# ------------------------------------------------------------------------------------------------
def heat_path(eng, g, hop):
    """The reference's own signature (heat coefficients, model/HSD.py:48-114) on the same graph: psi by the Chebyshev
    kernel, ragged sorted samples, all-pairs scipy-semantics W1.  Dense psi, so only while N^2 doubles is small."""
    import hsd_amd
    n = g.n_node

    def timed(fn):
        eng.sync()
        t0 = time.perf_counter()
        r = fn()
        eng.sync()
        return r, time.perf_counter() - t0
    try:
        hm = hsd_amd.HSD(g, "bench", 1.0, hop, "wasserstein", engine=eng)
        hm._check_default_hierarchy()
        psi, t_psi = timed(lambda: hm._wavelets_device(1.0, True))
        with psi:
            _, t_samp = timed(lambda: eng.heat_signature_build(hop, psi))
            DA, _ = timed(lambda: eng.pair_heat_rows_device(0, hop))
            with DA:
                _, t_a = timed(lambda: eng.pair_heat_rows_device(0, hop, matrix=DA))
                _, t_b = timed(lambda: eng.pair_heat_rowform_device(psi, matrix=DA))
        pairs = n * (n - 1) / 2.0
        nnzL = g.n_edge * 2 + n
        cblk = min(n, 1024)
        steps_total = 50 * math.ceil(n / cblk)
        bytes_step = 12.0 * nnzL + 24.0 * n * cblk                      # per order and column block
        t_step = t_psi / steps_total
        return {"signature": "heat coefficients psi = exp(-L), Chebyshev order 50, threshold 1e-4/N",
                "wavelets_s": round(t_psi, 4), "sorted_ring_samples_s": round(t_samp, 4),
                "variant_A_pair_s": round(t_a, 4), "variant_A_pairs_per_s": pairs / t_a,
                "variant_B_as_written_pair_s": round(t_b, 4), "variant_B_as_written_pairs_per_s": pairs / t_b,
                "k4": {"kernel": "cheby_step_kernel (one Chebyshev order of a 1024-impulse block)", "ms": round(t_step * 1e3, 4),
                       "bytes_alg": bytes_step, "achieved_GBps": round(bytes_step / t_step / 1e9, 1),
                       "frac_of_hbm": round(bytes_step / t_step / 1e9 / HBM_PEAK_GBS, 4)}}
    except (ValueError, MemoryError) as e:
        return {"error": str(e)}


def run_config2(eng, rk, steps, warmup, cpu_seconds):
    spec = WORKLOADS["config2"]
    g = make_graph(spec)
    cpu = {v: scipy_baseline(g, spec["hop"], v, cpu_seconds / 2) for v in ("A", "B")} if cpu_seconds > 0 else None
    r = run_degree_workload(eng, rk, g, spec, ["A", "B"], steps, warmup, cpu=cpu, traffic_prefix="er_n10000_K3")
    ent = dict(config="configs[1]", workload=spec["name"], nodes=g.n_node, edges=g.n_edge, hop=spec["hop"], bins=r["n_bins"],
               steps=steps, variants=[variant_summary(r["variants"][v]) for v in ("A", "B")],
               other_kernels=[r["sweep"]], phases_s=r["phases"])
    if cpu:
        ent["cpu_baseline"] = {v: cpu_entry(cpu[v], v, spec["hop"], "this graph") for v in ("A", "B")}
    ent["heat_signature_path"] = heat_path(eng, g, spec["hop"])
    return ent


def run_extra_er100k(eng, rk, steps, warmup):
    """NOT a BASELINE configuration: the headline kernel (Variant B) at the target's size on the OTHER graph family -- Erdos-Renyi, mean
    degree 10, N = 100 000: 24 degree bins instead of 241, ~11 operand words per pair instead of ~51.  It shows what the kernel's store
    path reaches when the arithmetic beside it is light (tools_dev/size_sweep.py has the whole size / family table)."""
    spec = dict(graph="er", nodes=100000, p=1e-4, hop=3, metric="wasserstein",
                name="extra (not a BASELINE config): erdos_renyi N=100000 p=0.0001 seed=42, K=3, Wasserstein, variant B")
    g = make_graph(spec)
    r = run_degree_workload(eng, rk, g, spec, ["B"], steps, warmup)
    return dict(config="extra", workload=spec["name"], nodes=g.n_node, edges=g.n_edge, hop=3, bins=r["n_bins"], steps=steps,
                variants=[variant_summary(r["variants"]["B"])], other_kernels=[r["sweep"]], phases_s=r["phases"])


def run_config3(eng, rk, steps, warmup, cpu_seconds):
    """KL on alpha-smoothed per-ring histograms; CPU baseline = the C restatement of
    tools/metrics.py:74-88 on the same histograms (kind "port": the reference's KL has no runnable semantics on ring
    samples), timed on a few rows; the same rows are the parity sample."""
    from oracle import c_oracle as corc
    spec = WORKLOADS["config3"]
    g = make_graph(spec)
    r = run_degree_workload(eng, rk, g, spec, ["KL"], steps, warmup, traffic_prefix="ba_n50000_K5")
    res = r["variants"]["KL"]
    ent = dict(config="configs[2]", workload=spec["name"], nodes=g.n_node, edges=g.n_edge, hop=spec["hop"], bins=r["n_bins"], steps=steps,
               other_kernels=[r["sweep"]], phases_s=r["phases"])
    if "error" in res:
        ent["error"] = res["error"]
        return ent
    ent.update(value=res["pairs_per_s"], unit="node-pairs/s", ms_per_step=res["ms_per_step"], dtype="f64", roofline=res["roofline"],
               matrix_check=res["matrix_check"])
    if cpu_seconds > 0:
        from hsd_amd import _lib as L
        counts = eng.signature_counts()
        n = g.n_node
        rows = sample_nodes(n, 64, seed=3)
        t0 = time.perf_counter()
        ref = {}
        for i in rows:
            ref[int(i)] = corc.pair_rows_kl(counts, 0, spec["hop"], 0.5, False, int(i), int(i) + 1)[0]
            if time.perf_counter() - t0 > cpu_seconds:
                break
        dt = time.perf_counter() - t0
        eng.pair_prepare(L.METRIC_KL, 0, 0, spec["hop"], 0.5, False)
        worst = 0.0
        for i, want in ref.items():
            got = eng.pair_rows_host(i, i + 1)[0]
            # KL(i || j) for j > i, mirrored below the diagonal as model/HSD.py:132-134 does: compare the row's upper part
            worst = max(worst, float(np.max(np.abs(got[i + 1:] - want[i + 1:]) / np.maximum(np.abs(want[i + 1:]), 1e-12))) if i + 1 < n else 0.0)
        pairs = len(ref) * n
        ent["cpu_baseline"] = dict(value=round(pairs / dt, 1), unit="node-pairs/s", cores=1, kind="port",
                                   sample=f"{len(ref)} rows x {n} columns of the same matrix in {dt:.1f} s: C restatement (oracle/hsd_oracle.c) of "
                                          f"tools/metrics.py:74-88 KL_divergence on the alpha = 0.5 smoothed histograms; host: {cpu_model()}")
        ent["parity"] = dict(pairs_checked=int(sum(n - i - 1 for i in ref)), max_rel_err=worst, tol=1e-6, ok=bool(worst <= 1e-6))
    return ent


def run_config4_one_gpu(eng, k=20):
    """configs[3] (BA N = 200 000, K = 3, Wasserstein) is quoted on 8 GPUs because its 320 GB matrix does not fit one.  What ONE
    GPU can do with it: the k-hop sweep and the fused k-nearest consumer (hsd_pair_topk), which evaluates every pair of the
    zero-pad distance and keeps each row's k nearest -- checked against single rows of the matrix computed on request."""
    from hsd_amd import _lib as L
    spec = WORKLOADS["config4"]
    g = make_graph(spec)
    n, hop = g.n_node, spec["hop"]
    eng.set_graph(g.rowptr, g.col)
    eng.signature_build_degree(hop)
    eng.timer_start()
    eng.signature_build_degree(hop)
    sweep_ms = eng.timer_stop()
    eng.pair_prepare(L.METRIC_WASSERSTEIN, 1, 0, hop - 1, 0.5, False)
    eng.pair_topk(k)
    eng.sync()
    t0 = time.perf_counter()
    idx, val = eng.pair_topk(k)
    dt = time.perf_counter() - t0
    ok = True
    for r in (0, n // 2, n - 1):
        d = eng.pair_rows_host(r, r + 1)[0]
        d[r] = np.inf
        order = np.lexsort((np.arange(n), d))[:k]
        ok = ok and bool(np.array_equal(order, idx[r]) and np.array_equal(d[order], val[r]))
    ent = dict(config="configs[3] on ONE GPU", workload=spec["name"] + " -- k-nearest consumer only: the 320 GB matrix is never formed",
               nodes=n, edges=g.n_edge, hop=hop, bins=int(eng.signature_dims()[1]), variant="B", call="hsd_pair_topk", k=k,
               value=n * (n - 1) / 2.0 / dt, unit="node-pairs/s", ms_end_to_end=round(dt * 1e3, 3), khop_sweep_ms=round(sweep_ms, 3),
               rows_checked_against_matrix_rows=3, exact=ok, dtype="u16")
    # Variant A (raw fp64 W1) on the same graph: the pruned sweep on node-permuted operands (first call = with its one-time setup)
    try:
        eng.pair_prepare(L.METRIC_WASSERSTEIN, 0, 0, hop, 0.5, False)
        eng.sync()
        t0 = time.perf_counter()
        eng.pair_topk(k)
        dt_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        idx, val = eng.pair_topk(k)
        dta = time.perf_counter() - t0
        oka = True
        for r in (0, n // 2, n - 1):
            d = eng.pair_rows_host(r, r + 1)[0]
            d[r] = np.inf
            order = np.lexsort((np.arange(n), d))[:k]
            oka = oka and bool(np.array_equal(order, idx[r]) and np.array_equal(d[order], val[r]))
        ent["variant_A"] = dict(call="hsd_pair_topk", k=k, ms_end_to_end=round(dta * 1e3, 3), ms_first_call=round(dt_first * 1e3, 3),
                                value=n * (n - 1) / 2.0 / dta, unit="node-pairs/s", rows_checked_against_matrix_rows=3, exact=oka, dtype="f64")
    except ValueError as e:                # an empty ring under the raw-sample distance
        ent["variant_A"] = dict(error=str(e))
    return ent


def run_config5(eng, cpu_seconds):
    """MultiHSD.parallel_embed (model/multiscale_HSD.py:76-98): 8 log-spaced scales, Chebyshev order 50, threshold
    1e-4/N, [sum, mean] per ring, K = 3 -> N x 64 fp64.  Unit: node signatures per second (N / wall).  The Chebyshev
    recurrence is priced with SURVEY 8(d): bytes per order and impulse block of width C = 12 nnz(L) + 3 * 8 N C."""
    import hsd_amd
    from oracle import hsd_oracle as orc
    spec = dict(graph="ba", nodes=50000, m=5)
    g = make_graph(spec)
    n, hop, S, order = g.n_node, 3, 8, 50
    eng.sync()
    t0 = time.perf_counter()
    m = hsd_amd.MultiHSD(g, "ba50000", hop, S, engine=eng)
    t_init = time.perf_counter() - t0
    t0 = time.perf_counter()
    emb = m.parallel_embed(1, approx=True)
    eng.sync()
    t_first = time.perf_counter() - t0
    # the device part alone, warm: one recurrence for all scales + ring sums
    sets = [m._cheby(float(s), True) for s in m.scales]
    coeffs = np.stack([c for c, _ in sets])
    thr = 1e-4 / n
    eng.sync()
    t0 = time.perf_counter()
    eng.multiscale_ring_stats(hop, coeffs, m.lmax, thr)
    eng.sync()
    t_dev = time.perf_counter() - t0
    nnzL = g.n_edge * 2 + n
    bytes_alg = order * (12.0 * nnzL * math.ceil(n / 512) + 24.0 * n * n)      # all impulse blocks (C = 512), all orders
    ent = dict(config="configs[4]", workload="BASELINE configs[4]: model.multiscale_HSD heat-kernel signatures, 8 scales, "
                                              "barabasi_albert N=50000 m=5 seed=42, K=3, Chebyshev order 50, threshold 1e-4/N",
               nodes=n, edges=g.n_edge, hop=hop, scales=S, value=n / t_dev, unit="node signatures/s (8 scales x 4 rings x [sum, mean])",
               seconds=round(t_dev, 4), dtype="f64",
               roofline=dict(bound="hbm", achieved=round(bytes_alg / t_dev / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=round(bytes_alg / t_dev / 1e9 / HBM_PEAK_GBS, 4), traffic=traffic_fields("ba_n50000_multihsd")[0],
                             traffic_detail=traffic_fields("ba_n50000_multihsd")[1],
                             bytes_alg=bytes_alg, kernel="cheby_step / cheby_combine (K4): order x (12 nnz(L) per block + 24 N C)"),
               phases_s=dict(lmax_estimate_and_setup=round(t_init, 4), parallel_embed_first_call=round(t_first, 4),
                             device_recurrence_and_ring_sums_warm=round(t_dev, 4)))
    if cpu_seconds > 0:
        # CPU baseline: the PyGSP recurrence restated in numpy/scipy.sparse (oracle.cheby_heat_apply; PyGSP itself is
        # absent) on a few unit impulses at once, all 8 scales, + get_triple; also the parity sample
        from scipy.sparse import csr_matrix
        rp, c, v = g.laplacian_csr()
        Lm = csr_matrix((v, c, rp), shape=(n, n))
        nodes = [int(x) for x in sample_nodes(n, 32, seed=5)]
        E = np.zeros((n, len(nodes)))
        for t, u in enumerate(nodes):
            E[u, t] = 1.0
        rings = {u: orc.khop_rings_node(g.rowptr, g.col, u, hop) for u in nodes}
        t0 = time.perf_counter()
        worst = 0.0
        for si, s in enumerate(m.scales):
            cc = orc.cheby_coeff_heat(s, m.lmax, order)
            R = orc.cheby_heat_apply(Lm, cc, m.lmax, E)
            for t, u in enumerate(nodes):
                row = orc.threshold_wavelets(R[:, t], n)
                ref = np.asarray(orc.get_triple(row[None, :], rings[u], 0))
                got = np.asarray(emb[u][si * 2 * (hop + 1):(si + 1) * 2 * (hop + 1)])
                # an entry within rounding of the 1e-4/N threshold may be kept by one side only (each flip moves a
                # ring sum by at most 2e-9 at this N): absolute slack 2e-8
                worst = max(worst, float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-2))))
        dt = time.perf_counter() - t0
        ent["cpu_baseline"] = dict(value=round(len(nodes) / dt, 3), unit="node signatures/s", cores=1, kind="port",
                                   sample=f"{len(nodes)} source nodes x 8 scales in {dt:.1f} s: numpy/scipy.sparse restatement of pygsp cheby_op "
                                          f"(model/HSD.py:55-59) + get_triple (model/multiscale_HSD.py:45-61); host: {cpu_model()}")
        ent["parity"] = dict(nodes_checked=len(nodes), max_rel_err=worst, rel_err_floor=1e-2, tol=1e-6, ok=bool(worst <= 1e-6))
    return ent


# ------------------------------------------------------------------------------------------------
# output: ONE compact contract line on stdout; every other figure goes to bench_detail.json (+ stderr)
# ------------------------------------------------------------------------------------------------
METRIC = "node-pairs/sec on NxN structural-distance matrix"
LINE_LIMIT = 4000               # bytes of the contract line (the driver keeps only a tail of stdout: a 20 KB line was lost in round 4)
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_bytes_alg", "kernel_ms", "bytes_alg", "valu_frac", "limiter")
# row blocks hold full rows: a pair (i, j) with i and j on different ranks is evaluated by both, so P ranks can at best
# reach P / (2 - 1/P) of one GPU's rate before any communication (3.6 at P = 4 ... 4.27 at P = 8; measured 4.3-4.8 per-block on one GPU
# because small blocks also skip the mirrored stores' tail); DESIGN.md section 5 C1
def row_block_ceiling(world):
    return round(world / (2.0 - 1.0 / world), 2)


def _short(text, limit):
    text = str(text)
    return text if len(text) <= limit else text[:limit - 3] + "..."


def _config_summary(c):
    """One number per side configuration: config, value, unit, ms_per_step, frac (+ parity.ok, error)."""
    if "error" in c and "value" not in c and "variants" not in c:
        return dict(config=c.get("config"), error=_short(c["error"], 120))
    out = dict(config=c.get("config"))
    src = c
    if "variants" in c:                                    # degree workloads list both variants: variant B (main.py's call) is the figure
        vs = [v for v in c["variants"] if "error" not in v]
        src = next((v for v in vs if v.get("variant") == "B"), vs[0] if vs else {"error": c["variants"][0].get("error")})
        out["variant"] = src.get("variant")
    if "error" in src:
        return dict(out, error=_short(src["error"], 120))
    out.update(value=src.get("value"), unit=_short(src.get("unit", ""), 40))
    ms = src.get("ms_per_step", src.get("ms_end_to_end"))
    if ms is None and "seconds" in src:
        ms = src["seconds"] * 1e3
    out["ms_per_step"] = ms
    rf = src.get("roofline") or {}
    out["frac"] = rf.get("frac")
    if rf.get("traffic_over_bytes_alg") is not None:
        out["traffic_over_bytes_alg"] = rf["traffic_over_bytes_alg"]
    par = src.get("parity") or c.get("parity")
    if par:
        out["parity_ok"] = par.get("ok")
    if "exact" in src:
        out["exact"] = src["exact"]
    return out


def compact_line(d, detail_path):
    """The contract line: the keys the driver reads, the roofline / cpu_baseline / parity objects with scalar fields only, one
    summary per side configuration; < LINE_LIMIT bytes whatever the run measured (optional keys are dropped first)."""
    cfg = d["config"]
    line = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype")}
    line["data"] = _short(d["data"], 100)
    line["config"] = dict(workload=_short(cfg.get("workload_short") or cfg["workload"], 240), nodes=cfg["nodes"], edges=cfg["edges"], hop=cfg["hop"], bins=cfg["bins"],
                          parallelism=_short(cfg["parallelism"], 80))
    line["roofline"] = {k: d["roofline"].get(k) for k in ROOFLINE_KEYS}
    cb = d.get("cpu_baseline")
    line["cpu_baseline"] = dict(value=cb["value"], unit=cb["unit"], cores=cb["cores"], kind=cb["kind"], sample=_short(cb["sample"], 330)) if cb else None
    line["parity"] = d.get("parity")
    if d["n_gpus"] > 1:
        line["kernel_ms_per_rank"] = d.get("kernel_ms_per_rank")
        ph = d.get("phases_s") or {}
        line["phases_s"] = dict(khop_signature_build=ph.get("khop_signature_build"), allgather=ph.get("allgather"), exchange=_short(ph.get("exchange", ""), 90))
        line["design_ceiling"] = dict(speedup_at_most=row_block_ceiling(d["n_gpus"]),
                                      why="full rows per rank: each off-diagonal pair is evaluated on two ranks")
    optional = []
    pool = d.get("cpu_baseline_pool")
    if pool:
        line["cpu_baseline_pool"] = dict(value=pool["value"], unit=pool["unit"], cores=pool["cores"], kind=pool["kind"])
        optional.append("cpu_baseline_pool")
    sec = d.get("secondary")
    if sec:
        line["secondary"] = _config_summary(dict(sec, config="variant " + str(sec.get("variant"))))
        optional.append("secondary")
    knn = d.get("fused_knn")
    if knn:
        line["fused_knn"] = dict(k=knn.get("k"), ms_end_to_end=knn.get("ms_end_to_end"), exact=knn.get("exact"))
        optional.append("fused_knn")
    line["configs"] = [_config_summary(c) for c in d.get("configs") or []]
    line["device"] = d.get("device")
    line["detail"] = detail_path
    for key in optional + ["configs"]:
        if len(json.dumps(line, allow_nan=False)) < LINE_LIMIT:
            break
        line.pop(key, None)
    return line


def _strict(obj):
    """NaN / inf have no JSON spelling: a figure that came out non-finite is reported as null."""
    if isinstance(obj, float):
        return obj if math.isfinite(obj) else None
    if isinstance(obj, dict):
        return {str(k): _strict(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_strict(v) for v in obj]
    if isinstance(obj, (np.floating, np.integer)):
        return _strict(obj.item())
    return obj


def emit(detail, detail_name=None):
    """bench_detail.json (next to bench.py, and under gpurun_out/ so it travels back from a GPU box) gets everything measured;
    stdout gets exactly ONE line, the compact contract line, last."""
    detail = _strict(detail)
    name = detail_name or "bench_detail.json"
    text = json.dumps(detail, indent=1, allow_nan=False)
    written = None
    for folder in (REPO, os.path.join(REPO, "gpurun_out")):
        try:
            if os.path.isdir(folder):
                with open(os.path.join(folder, name), "w") as f:
                    f.write(text + "\n")
                written = written or name
        except OSError:
            pass
    print("bench detail (also in %s):" % (written or "no file: directory not writable"), file=sys.stderr)
    print(json.dumps(detail, allow_nan=False), file=sys.stderr, flush=True)
    line = json.dumps(compact_line(detail, written), allow_nan=False)
    assert len(line) < LINE_LIMIT and "\n" not in line
    sys.stdout.flush()
    print(line, flush=True)
    return line


# ------------------------------------------------------------------------------------------------
def main():
    args = parse()
    global EXTRAS
    EXTRAS = not args.no_extras
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import hsd_amd

    custom = any(v is not None for v in (args.graph, args.nodes, args.hop, args.metric)) or args.weak
    if custom:
        base = WORKLOADS["config2"] if (args.weak or args.graph in (None, "er")) else WORKLOADS["target"]
        n = args.nodes if args.nodes is not None else (10000 if args.weak else base["nodes"])
        if args.weak:
            n = int(round(n * math.sqrt(world)))
        kind = args.graph or base["graph"]
        spec = dict(graph=kind, nodes=n, hop=args.hop if args.hop is not None else base["hop"], metric=args.metric or "wasserstein")
        if kind == "er":
            spec.update(p=args.mean_degree / n, connect=world > 1)
            spec["name"] = f"erdos_renyi N={n} p={args.mean_degree / n:g} seed=42" + (" + small components joined" if world > 1 else "")
        else:
            spec.update(m=int(args.mean_degree / 2))
            spec["name"] = f"barabasi_albert N={n} m={int(args.mean_degree / 2)} seed=42"
        spec["name"] += f", K={spec['hop']}, {spec['metric']}" + (" (weak scaling: N = nodes * sqrt(P))" if args.weak else "")
        tkey = f"{kind}_n{n}_K{spec['hop']}" if args.mean_degree == 10.0 else None
    else:
        spec = dict(WORKLOADS[args.workload])
        tkey = f"{spec['graph']}_n{spec['nodes']}_K{spec['hop']}"
    g = make_graph(spec)
    n = g.n_node
    wass = spec["metric"] == "wasserstein"
    primary_v = args.variant if wass else "KL"
    variants = [primary_v] + ([("B" if args.variant == "A" else "A")] if wass and not args.no_secondary else [])

    cfg_list = args.configs
    if cfg_list is None:
        cfg_list = "none" if custom or args.workload != "target" else ("extra,2,3,4,5" if world == 1 else ("4" if world == 8 else "none"))
    cfgs = [] if cfg_list == "none" else [c.strip() for c in cfg_list.split(",") if c.strip()]

    # ---- CPU baseline first, on rank 0 of ANY world size (the other ranks wait in the rendezvous below): the process pool
    # forks, which must happen before anything initialises the GPU -- or starts gloo's threads -- in this process ----
    cpu = None
    cpu_pool = None
    if rank == 0 and args.cpu_seconds > 0 and wass:
        cpu = {primary_v: scipy_baseline(g, spec["hop"], primary_v, args.cpu_seconds)}
        for v in variants[1:]:                         # the other variant: a shorter sample, mainly for its parity check
            cpu[v] = scipy_baseline(g, spec["hop"], v, args.cpu_seconds / 3)
        if world == 1:
            cpu_pool = scipy_pool_baseline(g, spec["hop"], primary_v, min(10.0, args.cpu_seconds), cpu[primary_v]["rate"])

    dist = None
    if world > 1:
        # control plane only (unique-id broadcast, barriers, max over ranks); the data-plane
        # exchange is the library's own ncclAllGather over RCCL
        import torch.distributed as dist
        # (gloo's C++ side reports "[Gloo] Rank r is connected to ..." on STDOUT: while the group forms, file descriptor 1 points at stderr, so that
        #  stdout stays what the contract says -- the one JSON line)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            # (a rank that died leaves the others in a collective: five minutes, not gloo's thirty, until they give up and say so)
            import datetime
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=5))
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    rk = Ranks(dist, rank, world)

    # HSD_COMM_DRYRUN=1 rehearses the multi-rank flow on a box with fewer GPUs than ranks: every rank on GPU 0,
    # no RCCL communicator, the signature exchange through the host (gloo) -- never a measurement
    dryrun = bool(os.environ.get("HSD_COMM_DRYRUN"))
    eng = hsd_amd.Engine(0 if dryrun else local_rank)
    info = eng.device_info()
    r = run_degree_workload(eng, rk, g, spec, variants, args.steps, args.warmup, cpu=cpu, traffic_prefix=tkey, dryrun=dryrun)
    prim = r["variants"][primary_v]
    if "error" in prim:
        raise ValueError(prim["error"])

    configs = []
    if rank == 0 and world == 1:
        csteps, cwarm = max(3, min(args.steps, 20)), 2
        for c in cfgs:
            try:
                if c == "2":
                    configs.append(run_config2(eng, rk, max(csteps, 50), 5, min(args.cpu_seconds, 16.0)))
                elif c == "3":
                    configs.append(run_config3(eng, rk, csteps, cwarm, min(args.cpu_seconds, 6.0)))
                elif c == "5":
                    configs.append(run_config5(eng, min(args.cpu_seconds, 10.0)))
                elif c == "4":
                    configs.append(run_config4_one_gpu(eng))
                elif c == "extra":
                    # a fresh context: the headline's k-nearest calls leave gigabytes of scratch behind, and a store-bound kernel whose 80 GB matrix
                    # is allocated around them runs up to 17 % slower
                    eng.close()
                    eng = hsd_amd.Engine(0 if dryrun else local_rank)
                    configs.append(run_extra_er100k(eng, rk, csteps, cwarm))
            except Exception as e:        # noqa: BLE001  (a side configuration must not lose the headline)
                configs.append(dict(config=c, error=f"{type(e).__name__}: {e}"))
    if world > 1 and "4" in cfgs:
        # config 4 on the same ranks: a fresh context (new row partition and communicator for the larger graph)
        eng.close()
        eng = hsd_amd.Engine(0 if dryrun else local_rank)
        spec4 = dict(WORKLOADS["config4"])
        if os.environ.get("HSD_BENCH_CONFIG4_NODES"):
            # rehearsal only (a box with fewer GPUs than ranks cannot hold the 320 GB matrix): the same code path on a smaller graph, labelled as such
            spec4["nodes"] = int(os.environ["HSD_BENCH_CONFIG4_NODES"])
            spec4["name"] += f" -- REHEARSAL at N={spec4['nodes']} (HSD_BENCH_CONFIG4_NODES), not the configuration"
        # (a side configuration must not lose the headline: whatever goes wrong here -- on this rank, or on another one, which this rank then
        #  notices as a collective that times out -- ends up as an error entry, and the contract line is still printed)
        try:
            g4 = make_graph(spec4)
            r4 = run_degree_workload(eng, rk, g4, spec4, ["B", "A"], max(3, min(args.steps, 10)), 2, dryrun=dryrun)
            if rank == 0:
                configs.append(dict(config="configs[3]", workload=spec4["name"], nodes=g4.n_node, edges=g4.n_edge, hop=3, bins=r4["n_bins"],
                                    n_gpus=world, rows_of_rank0=list(r4["variants"]["B"].get("rows", r4["rows"])), scaling="row blocks over all ranks (equal planned cost per rank)",
                                    variants=[dict(variant_summary(r4["variants"][v]), kernel_ms_per_rank=r4["variants"][v].get("kernel_ms_per_rank"))
                                              for v in ("B", "A")],
                                    other_kernels=[r4["sweep"]], phases_s=r4["phases"]))
        except Exception as e:        # noqa: BLE001
            if rank == 0:
                configs.append(dict(config="configs[3]", error=f"{type(e).__name__}: {e}"[:200]))

    if rank == 0:
        sec = None
        if len(variants) > 1:
            sec = variant_summary(r["variants"][variants[1]])
            if cpu and variants[1] in cpu and "error" not in sec:
                sec["cpu_baseline"] = cpu_entry(cpu[variants[1]], variants[1], spec["hop"], "the same matrix")
        detail = {
            "metric": METRIC,
            "value": prim["pairs_per_s"], "unit": "node-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": prim["ms_per_step"], "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": prim["dtype"],
            "data": "synthetic" if not dryrun else "synthetic (HSD_COMM_DRYRUN rehearsal: all ranks on one GPU, not a measurement)",
            "config": {"workload": f"{spec['name']}, degree-histogram signatures, variant {primary_v} ({variant_call(primary_v) if wass else 'KL on alpha-smoothed ring histograms'}; "
                                   f"layers {prim['lo']}..{prim['hi']}), full symmetric fp64 matrix resident in HBM",
                       "workload_short": f"{spec['name']}; degree-histogram signatures, variant {primary_v}, layers {prim['lo']}..{prim['hi']}; full symmetric fp64 matrix in HBM",
                       "nodes": n, "edges": g.n_edge, "hop": spec["hop"], "bins": r["n_bins"],
                       "parallelism": f"row-block x{world}" + (" (contiguous blocks of equal planned cost, hsd_pair_shard)" if world > 1 else ""),
                       "rows_of_rank0": list(prim["rows"])},
            "roofline": prim["roofline"],
            "cpu_baseline": cpu_entry(cpu[primary_v], primary_v, spec["hop"], "the same matrix") if cpu else None,
            "cpu_baseline_pool": dict(value=round(cpu_pool["rate"], 1), unit="node-pairs/s", cores=cpu_pool["workers"], kind="port",
                                      sample=f"{cpu_pool['pairs']} pairs in {cpu_pool['seconds']:.1f} s over a multiprocessing.Pool({cpu_pool['workers']}), "
                                             f"one task per source row as model/HSD.py:118-130 submits them") if cpu_pool else None,
            "parity": prim.get("parity"),
            "secondary": sec,
            "kernel_ms_per_rank": prim["kernel_ms_per_rank"],
            "other_kernels": [k for k in (r["sweep"],) if k],
            "phases_s": dict(r["phases"], operand_prepare_first=round(prim["t_prep"], 4),
                             first_matrix_incl_allocation=round(prim["t_first"], 4),
                             warm_prepare_plus_matrix=round(prim["t_warm"], 4),
                             end_to_end_one_matrix_warm=round(r["phases"]["khop_signature_build"] + r["phases"]["allgather"] + prim["t_warm"], 4)),
            "matrix_check": prim["matrix_check"],
            "fused_knn": prim.get("fused_knn"),
            "row_block_rehearsal": prim.get("row_block_rehearsal"),
            "configs": configs,
            "device": info["arch"], "cus": info["compute_units"],
        }
        emit(detail, args.detail)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
