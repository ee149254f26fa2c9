"""Python surface of the reference's model classes on top of the HIP engine.

`HSD` keeps the constructor signature, attribute names and method names of
model/HSD.py:17-161, `MultiHSD` those of model/multiscale_HSD.py:15-119, so a
caller of the reference (main.py:14-28, tests/robust_test/main.py:173-187) can
switch imports.  The numerical work is done by libhsd_amd.so; there is no CPU
fallback.

Two signatures are available (`signature=`):
  "heat"    the reference's own: per ring, the multiset of heat-kernel wavelet
            coefficients psi[i, ring_k(i)] (model/HSD.py:71-83).  psi is dense N x N,
            so this is for graphs whose N^2 fits in HBM.
  "degree"  the per-hop degree histogram of BASELINE.json's north star: per ring,
            the multiset of member degrees, held losslessly as a histogram over
            the graph's distinct degrees.  Scales to N = 1e5..1e6.
The distance arithmetic applied to either is the reference's (scipy W1 on raw
samples = Variant A; tools.metrics.calculate_distance = Variant B).
"""
from __future__ import annotations

from collections import defaultdict

import numpy as np

from . import _lib as L
from . import heat as _heat
from .engine import DeviceMatrix, Engine
from .graph import Graph

_METRICS = ['l1', 'l2', 'kl', 'symmetric_kl', 'js', 'wasserstein_guass', 'wasserstein', 'hellinger']
_ON_DEVICE = {"wasserstein": L.METRIC_WASSERSTEIN, "kl": L.METRIC_KL, "symmetric_kl": L.METRIC_SYMMETRIC_KL}


def _metric_code(metric):
    """Name validation exactly as tools/metrics.py:160-167."""
    if not metric or not isinstance(metric, str):
        raise TypeError("Need to specify a metric.")
    m = metric.lower()
    if m not in _METRICS:
        raise NotImplementedError("{} metric is not implemented.".format(m))
    if m not in _ON_DEVICE:
        raise NotImplementedError(
            "{} metric is outside the accelerated hot path (wasserstein, kl, symmetric_kl).".format(m))
    return _ON_DEVICE[m]


class HSD(object):

    def __init__(self, graph, graphName, scale, hop, metric, signature="heat", device=0,
                 kl_alpha=0.5, engine=None):
        """
        Hierarchical Structural Distance model (model/HSD.py:19-40).
        :param graph: networkx.Graph or hsd_amd.Graph
        :param graphName: the name of graph
        :param scale: the heat coefficient
        :param hop: k-hop local neighborhoods
        :param metric: 'wasserstein' ('kl' / 'symmetric_kl' on the degree signature)
        """
        if signature not in ("heat", "degree"):
            raise ValueError("signature must be 'heat' or 'degree'")
        self.graph = graph
        self.graphName = graphName
        self.scale = scale
        self.hop = hop
        self.metric = metric
        self.signature = signature
        self.kl_alpha = kl_alpha
        self._g = Graph.coerce(graph)

        self.nodes = list(self._g.nodes)
        self.n_node = len(self.nodes)
        self.idx2node = {i: v for i, v in enumerate(self.nodes)}
        self.node2idx = dict(self._g.node2idx)
        self._hierarchy = None
        self._hierarchy_user = False
        self.wavelets = None            # the reference never assigns it; callers do
        self.distMat = None
        self.lmax = None                # spectral bound used by the Chebyshev recurrence

        self._engine = engine if engine is not None else Engine(device)
        self._lap = None
        self._lap_on_device = False
        self._sig_key = None
        _ = self.engine                 # uploads the graph

    @property
    def engine(self):
        """The Engine (one GPU context).  Several models may share one: a model that finds the context holding
        another model's graph uploads its own again and forgets what it had cached there."""
        e = self._engine
        if getattr(e, "_owner", None) is not self:
            e.set_graph(self._g.rowptr, self._g.col)
            e._owner = self
            self._lap_on_device = False
            self._sig_key = None
        return e

    # dense A and L exist in the reference (model/HSD.py:33-34); here they are built only if read
    @property
    def A(self):
        return self._g.adjacency_dense()

    @property
    def L(self):
        return self._g.laplacian_dense()

    # ---- hierarchy (tools/hierarchy.py:16-38 on the GPU) -----------------
    @property
    def hierarchy(self):
        if self._hierarchy is None:
            self._hierarchy = self._rings_to_dict(*self.engine.khop_members(self.hop))
        return self._hierarchy

    @hierarchy.setter
    def hierarchy(self, value):
        self._hierarchy = value
        self._hierarchy_user = value is not None
        self._sig_key = None

    def _rings_to_dict(self, offsets, members):
        K1 = self.hop + 1
        out = {}
        for i, v in enumerate(self.nodes):
            out[v] = [[self.nodes[u] for u in members[offsets[i * K1 + k]:offsets[i * K1 + k + 1]]]
                      for k in range(K1)]
        return out

    def _check_default_hierarchy(self):
        """A caller-assigned hierarchy (tests/robust_test/main.py:179) is honoured by the degree
        signature; the coefficient path needs it to equal the BFS rings the device computes."""
        if not self._hierarchy_user:
            return
        off, mem = self.engine.khop_members(self.hop)
        ref = self._rings_to_dict(off, mem)
        for v in self.nodes:
            mine = [sorted(self.node2idx[u] for u in layer if u != '') for layer in self._hierarchy[v]]
            theirs = [sorted(self.node2idx[u] for u in layer) for layer in ref[v]]
            if mine[:self.hop + 1] != theirs:
                raise NotImplementedError(
                    "signature='heat' uses the k-hop rings of the graph; a different caller-assigned "
                    "hierarchy is only supported with signature='degree'")

    def init(self):
        """model/HSD.py:43-44 reads a .layers file from a hard-coded path; here the
        hierarchy is computed by the k-hop sweep."""
        self._hierarchy = None
        self._hierarchy_user = False
        _ = self.hierarchy

    def get_nodes_hierarchical_degree(self) -> dict:
        """model/HSD.py:87-94: ring sizes, padded with 0 to length `hop`."""
        if self._hierarchy_user:
            out = {}
            for node, layers in self._hierarchy.items():
                d = [len(level) for level in layers]
                if len(d) < self.hop:
                    d = d + [0] * (self.hop - len(d))
                out[node] = d
            return out
        sizes = self.engine.khop_sizes(self.hop)
        return {v: [int(x) for x in sizes[i]] for i, v in enumerate(self.nodes)}

    # ---- wavelets (model/HSD.py:48-67) ------------------------------------
    def _ensure_laplacian(self):
        if self._lap is None:
            self._lap = self._g.laplacian_csr()
        eng = self.engine
        if not self._lap_on_device:
            eng.set_laplacian(*self._lap)
            self._lap_on_device = True
        return self._lap

    def _cheby(self, scale, approx):
        rp, c, v = self._ensure_laplacian()
        if approx:
            # model/HSD.py:50-53: PyGSP lmax estimate, Heat filter, Chebyshev order 50
            if self.lmax is None:
                self.lmax = _heat.estimate_lmax(rp, c, v, self.n_node)
            return _heat.cheby_coeff_heat(scale, self.lmax, 50), self.lmax
        # model/HSD.py:61-63: exact exp(-s L); here a Chebyshev series run to rounding error
        bound = _heat.spectral_bound(rp, c, v, self.n_node)
        return _heat.exact_coefficients(scale, bound), bound

    def _wavelets_device(self, scale, approx):
        coeff, lmax = self._cheby(scale, approx)
        thr = 1e-4 * 1.0 / self.n_node                  # model/HSD.py:65
        return self.engine.heat_wavelets_device(coeff, lmax, thr)

    def wavelets_for_scales(self, scales, approx=True):
        """Resident psi matrices (DeviceMatrix, caller frees) for several heat scales from ONE Chebyshev
        recurrence: the polynomials T_k(L) e_c do not depend on the scale, only the coefficients do.  Serves the
        5-scale sweep of main.py:24-28 and the sum over scales of model/multiscale_HSD.py:104-117."""
        sets = [self._cheby(float(s), approx) for s in scales]
        lmax = sets[0][1]
        order = max(len(c) for c, _ in sets) - 1
        coeffs = np.zeros((len(sets), order + 1), dtype=np.float64)
        for si, (c, _) in enumerate(sets):
            coeffs[si, :len(c)] = c
        return self.engine.heat_wavelets_multi_device(coeffs, lmax, 1e-4 * 1.0 / self.n_node)      # model/HSD.py:65

    def calculate_wavelets(self, scale, approx=True, device_out=False):
        """psi = exp(-scale L), thresholded at 1e-4/N (model/HSD.py:48-67), rows in self.nodes order.
        device_out=True returns the resident DeviceMatrix (it can be assigned to self.wavelets as it is;
        the caller frees it)."""
        psi = self._wavelets_device(scale, approx)
        if device_out:
            return psi
        try:
            return psi.to_host()
        finally:
            psi.free()

    def get_hierarchical_coeffcients(self, wavelets) -> dict:
        """model/HSD.py:71-83 (spelling kept): node -> hop+1 lists of psi[i, j], j in ring_k(i).
        Host-side view for callers that inspect it; the distance kernels keep their own
        device-resident copy."""
        out = {}
        for i, v in enumerate(self.nodes):
            out[v] = [[wavelets[i, self.node2idx[u]] for u in layer] for layer in self.hierarchy[v]]
        return out

    # ---- signatures ------------------------------------------------------
    def _ensure_degree_signature(self):
        _ = self.engine                 # bind first: a rebinding forgets the cached signature key
        key = ("degree", self.hop, id(self._hierarchy) if self._hierarchy_user else None)
        if self._sig_key == key:
            return
        if self._hierarchy_user:
            # caller-assigned rings (tests/robust_test/main.py:179): histogram them on the host
            # and upload; the pair kernels then run unchanged
            deg = self._g.degree
            support, inv = np.unique(deg, return_inverse=True)
            self.engine.signature_build(self.hop, inv.astype(np.int32), support.astype(np.float64), 0, 0)
            counts = np.zeros((self.n_node, self.hop + 1, support.shape[0]), dtype=np.uint32)
            for i, v in enumerate(self.nodes):
                for k, layer in enumerate(self._hierarchy[v][:self.hop + 1]):
                    idx = [self.node2idx[u] for u in layer if u != '']
                    if idx:
                        counts[i, k] = np.bincount(inv[idx], minlength=support.shape[0])
            self.engine.signature_set_counts(0, self.n_node, counts)
        else:
            self.engine.signature_build_degree(self.hop)
        self._sig_key = key

    def _degree_distance(self, align, layer_lo, layer_hi, device_out=False, kl_sorted=False):
        code = _metric_code(self.metric)
        self._ensure_degree_signature()
        self.engine.pair_prepare(code, align, layer_lo, layer_hi, self.kl_alpha, kl_sorted)
        if device_out:
            return self.engine.pair_rows_device()
        return self.engine.pair_rows_host()

    # ---- distances ---------------------------------------------------------
    def calculate_structural_distance(self, scale=None, approx=False, device_out=False):
        """Variant A (model/HSD.py:98-114): D[i,j] = sum_{k=0..hop} W1(c_i[k], c_j[k]) with
        scipy.stats.wasserstein_distance semantics on the raw ring samples; symmetric, zero
        diagonal, rows/cols in self.nodes order.  An empty ring raises
        ValueError("Distribution can't be empty.") as scipy does."""
        if self.signature == "degree":
            saved = self.metric
            try:
                self.metric = "wasserstein"          # model/HSD.py:111 always calls scipy W1
                return self._degree_distance(L.ALIGN_NONE, 0, self.hop, device_out)
            finally:
                self.metric = saved
        self._check_default_hierarchy()
        psi = self._wavelets_device(self.scale if scale is None else scale, approx)
        try:
            self.engine.heat_signature_build(self.hop, psi)
            if device_out:
                return self.engine.pair_heat_rows_device(0, self.hop)
            return self.engine.pair_heat_rows_host(0, self.hop)
        finally:
            psi.free()

    def parallel_calculate_HSD(self, n_workers=3, device_out=False, kl_sorted=False):
        """Variant B (model/HSD.py:118-161): layers 0..hop-1, per-layer
        tools.metrics.calculate_distance (zero-pad alignment + ascending sort) with
        self.metric; result also stored in self.distMat.  n_workers is accepted for
        signature compatibility; the GPU replaces the process pool.

        signature='heat' reproduces the method as written, including reading BOTH samples from
        row `startIndex` of self.wavelets (model/HSD.py:152,155); self.wavelets must have been
        assigned by the caller, exactly as the reference requires (it raises AttributeError
        otherwise)."""
        code = _metric_code(self.metric)
        if self.hop < 1:
            self.distMat = np.zeros((self.n_node, self.n_node), dtype=float)
            return self.distMat
        if self.signature == "degree":
            self.distMat = self._degree_distance(L.ALIGN_ZEROPAD, 0, self.hop - 1, device_out, kl_sorted=kl_sorted)
            return self.distMat
        if self.wavelets is None:
            raise AttributeError("'HSD' object has no attribute 'wavelets'")
        if code != L.METRIC_WASSERSTEIN:
            # ring coefficients do not sum to 1: tools/metrics.py:49-51 raises for kl / symmetric_kl
            raise ValueError("The sum of probability distribution must be 1.0.")
        self._check_default_hierarchy()
        resident = isinstance(self.wavelets, DeviceMatrix)      # calculate_wavelets(..., device_out=True): no round trip
        psi = self.wavelets if resident else self.engine.upload_matrix(np.asarray(self.wavelets, dtype=np.float64))
        try:
            self.engine.heat_signature_build(self.hop, psi)
            self.distMat = self.engine.pair_heat_rowform_device(psi) if device_out else self.engine.pair_heat_rowform_host(psi)
        finally:
            if not resident:
                psi.free()
        return self.distMat


class MultiHSD(HSD):

    def __init__(self, graph, graphName: str, hop: int, n_scales: int, metric="euclidean", signature="heat",
                 device=0, engine=None):
        """model/multiscale_HSD.py:17-23.  `metric` is stored but unused, as in the reference."""
        super(MultiHSD, self).__init__(graph, graphName, 0, hop, "wasserstein", signature=signature, device=device,
                                       engine=engine)
        self.metric = metric
        self.n_scales = n_scales
        self.scales = None
        self.embeddings = None
        self.init()

    def init(self):
        """model/multiscale_HSD.py:26-31: log-spaced scales from the lmax estimate; the hierarchy is
        computed by the k-hop sweep instead of being read from a hard-coded .layers path."""
        rp, c, v = self._ensure_laplacian()
        self.lmax = _heat.estimate_lmax(rp, c, v, self.n_node)
        self.scales = np.exp(np.linspace(np.log(0.01), np.log(self.lmax * 1.25), self.n_scales))
        self._hierarchy = None
        self._hierarchy_user = False

    def _embed(self, approx=True):
        """[sum, mean] of psi_s[i, ring_k(i)] per scale and ring (get_triple, model/multiscale_HSD.py:45-61)."""
        self._check_default_hierarchy()
        K1 = self.hop + 1
        scales = [float(s) for s in self.scales]
        emb = np.zeros((self.n_node, len(scales), K1, 2), dtype=np.float64)
        thr = 1e-4 * 1.0 / self.n_node                  # model/HSD.py:65
        # One Chebyshev sweep serves every scale (the polynomials are scale-independent, only the
        # coefficients differ), so all coefficient sets are padded to one common order.
        sets = [self._cheby(s, approx) for s in scales]
        lmax = sets[0][1]
        order = max(len(c) for c, _ in sets) - 1
        coeffs = np.zeros((len(scales), order + 1), dtype=np.float64)
        for si, (c, _) in enumerate(sets):
            coeffs[si, :len(c)] = c
        for s0 in range(0, len(scales), 4096):          # the ABI takes up to 4096 scales per call (one recurrence for all)
            sums, sizes = self.engine.multiscale_ring_stats(self.hop, coeffs[s0:s0 + 4096], lmax, thr)
            emb[:, s0:s0 + sums.shape[0], :, 0] = np.transpose(sums, (1, 0, 2))
            mean = np.where(sizes > 0, sums / np.maximum(sizes, 1), 0.0)
            emb[:, s0:s0 + sums.shape[0], :, 1] = np.transpose(mean, (1, 0, 2))
        flat = emb.reshape(self.n_node, -1)
        out = defaultdict(list)
        for i, v in enumerate(self.nodes):
            out[v].extend(float(x) for x in flat[i])
        return out

    def embed(self, approx=True) -> dict:
        """model/multiscale_HSD.py:35-42: node -> list of n_scales * (hop+1) * 2 floats."""
        return self._embed(approx)

    def get_triple(self, wavelets: np.ndarray, node) -> list:
        """model/multiscale_HSD.py:45-61 for one node of a host wavelet matrix."""
        out = []
        i = self.node2idx[node]
        for level in self.hierarchy[node]:
            coeffs = [wavelets[i, self.node2idx[u]] for u in level if u != '']
            out.extend([np.sum(coeffs), np.mean(coeffs)] if len(coeffs) > 0 else [0.0, 0.0])
        return out

    def get_layer_sum(self, wavelets: np.ndarray, node) -> list:
        """model/multiscale_HSD.py:64-73 for one node of a host wavelet matrix: the sum of every ring's coefficients
        (the reference keeps it beside get_triple; its callers there are commented out)."""
        i = self.node2idx[node]
        out = [0] * (self.hop + 1)
        for hop, level in enumerate(self.hierarchy[node]):
            for u in level:
                if u != '':
                    out[hop] += wavelets[i, self.node2idx[u]]
        return out

    def parallel_embed(self, n_workers=1, approx=True) -> dict:
        """model/multiscale_HSD.py:76-98; one GPU pass per scale replaces the pool."""
        self.embeddings = self._embed(approx)
        return self.embeddings

    def parallel_calculate_structural_distance(self, n_workers: int = 1, approx=True):
        """model/multiscale_HSD.py:101-119: sum over scales of the Variant-A matrix
        (approx=True is hard-coded there, :107)."""
        if self.signature != "heat":
            # the degree signature has no scale: the sum over the scales (:112-117) is the one Variant-A matrix added to
            # itself len(scales) times -- computed once and added in the reference's order (same rounding as its loop)
            one = HSD.calculate_structural_distance(self, float(self.scales[0]) if len(self.scales) else 0.0, approx)
            total = np.zeros((self.n_node, self.n_node), dtype=float)
            for _ in self.scales:
                total += one
            return total
        # heat coefficients: one Chebyshev recurrence serves a batch of scales (as many psi matrices as a quarter of
        # the HBM holds), the per-scale matrices are added on the device, one copy to the host at the end
        self._check_default_hierarchy()
        hbm = self.engine.device_info()["hbm_bytes"]
        batch = max(1, int(0.25 * hbm / (8.0 * self.n_node * self.n_node)))
        total = None
        scales = [float(x) for x in self.scales]
        try:
            for s0 in range(0, len(scales), batch):
                psis = self.wavelets_for_scales(scales[s0:s0 + batch], approx)
                try:
                    for psi in psis:
                        self.engine.heat_signature_build(self.hop, psi)
                        D = self.engine.pair_heat_rows_device(0, self.hop)
                        if total is None:
                            total = D
                        else:
                            total.add(D)
                            D.free()
                finally:
                    for psi in psis:
                        psi.free()
            if total is None:
                return np.zeros((self.n_node, self.n_node), dtype=float)
            return total.to_host()
        finally:
            if total is not None:
                total.free()
