"""Counterpart of the reference's experiment driver (main.py:9-53).

The reference script does not run as written (`construct_hierarchy`, `model.laplacian`
and `self.wavelets` do not exist); this module implements the flow it intends:
edgelist -> graph in first-appearance node order -> scale range from the Laplacian spectrum
(tools/util.py:100-125) -> 5-scale sweep -> Variant-B distance matrix per scale -> kNN score on
the precomputed distances (sklearn stays the consumer, as in tools/evaluate.py:61-69).
"""
from __future__ import annotations

import os

import numpy as np

from .formats import read_label, save_edgelist
from .graph import Graph, read_edgelist
from .model import HSD, MultiHSD


def scale_boundary(e1, eN, eta=0.85, gamma=0.95):
    """The GraphWave scale interval the reference uses (tools/util.py:120-125): with g = sqrt(e1 * eN) the
    geometric mean of the two eigenvalues, s_min = -ln(gamma) / g and s_max = -ln(eta) / g."""
    geo_mean = np.sqrt(e1 * eN)
    return -np.log(gamma) / geo_mean, -np.log(eta) / geo_mean


def recommend_scale_range(eignvalues: list):
    """(s_min, s_max) from a Laplacian spectrum (tools/util.py:100-108; the parameter keeps the reference's
    spelling because callers pass it by name): e1 is the smallest eigenvalue above 0.001 -- the smallest
    one of all if none is -- and eN the largest."""
    spectrum = np.sort(np.asarray(eignvalues, dtype=float))
    above = spectrum[spectrum > 0.001]
    e1 = above[0] if above.size else spectrum[0]
    return scale_boundary(e1, spectrum[-1])


def laplacian_extreme_eigenvalues(g: Graph):
    """(smallest eigenvalue above 0.001, largest eigenvalue) of L = D - A: dense eigvalsh like
    main.py:21 for small graphs, ARPACK for large ones."""
    n = g.n_node
    if n <= 3000:
        lam = np.linalg.eigvalsh(g.laplacian_dense())
        return float(next((e for e in lam if e > 0.001), lam[0])), float(lam[-1])
    from scipy.sparse import csr_matrix
    from scipy.sparse.linalg import eigsh
    rp, c, v = g.laplacian_csr()
    L = csr_matrix((v, c, rp), shape=(n, n))
    top = float(eigsh(L, k=1, which="LA", return_eigenvectors=False)[0])
    low = np.sort(eigsh(L, k=min(8, n - 2), sigma=-1e-3, which="LM", return_eigenvectors=False))
    return float(next((e for e in low if e > 0.001), low[-1])), top


def filter_edgelist(edgelist: list, save_path=None, ratio=0.05) -> list:
    """tools/util.py:133-144: the int(len * ratio) + 1 heaviest edges (u, v, weight), heaviest first, equal weights in the order
    they came in; the caller's list is sorted in place, as there; written as an edgelist when `save_path` is given."""
    edgelist.sort(key=lambda e: e[2], reverse=True)
    kept = edgelist[:int(len(edgelist) * ratio) + 1]
    if save_path is not None:
        save_edgelist(save_path, kept)
    return kept


def filter_distance_matrix(dist_mat, nodes: list, save_path=None, ratio=0.05) -> list:
    """tools/util.py:147-159 (same positional order: matrix, nodes, save_path, ratio): the (int(#pairs * ratio) + 1) pairs i < j with
    the LARGEST distance, sorted descending (stable), as (node_i, node_j, distance); saved as `u v d` lines when `save_path` is given."""
    dist_mat = np.asarray(dist_mat)
    assert dist_mat.shape[0] == dist_mat.shape[1], "the distance matrix must be square"
    assert dist_mat.shape[0] == len(nodes), "one row of the distance matrix per node"
    iu, ju = np.triu_indices(len(nodes), 1)
    d = dist_mat[iu, ju]
    keep = int(len(d) * ratio) + 1
    order = np.argsort(-d, kind="stable")[:keep]
    kept = [(nodes[iu[t]], nodes[ju[t]], float(d[t])) for t in order]
    if save_path is not None:
        save_edgelist(save_path, kept)
    return kept


def _knn_score(dists, labels, cv, n_neighbor):
    try:
        from sklearn.model_selection import cross_val_score
        from sklearn.neighbors import KNeighborsClassifier
        from sklearn import utils as sktools
    except ImportError:
        return None
    dists, labels = sktools.shuffle(dists, labels)
    knn = KNeighborsClassifier(weights="uniform", algorithm="auto", n_neighbors=n_neighbor, metric="precomputed", p=2)
    return float(np.mean(cross_val_score(knn, dists, y=labels, cv=cv, scoring="accuracy")))


def base_HSD_Test(graphName, hop=3, metric="wasserstein", data_dir="data", signature="heat", approx=True, engine=None,
                  evaluate=True):
    """main.py:9-34.  Returns (score_max, scale_opt, {scale: distance matrix})."""
    graph = read_edgelist(os.path.join(data_dir, "graph", f"{graphName}.edgelist"))
    label_dict = read_label(os.path.join(data_dir, "label", f"{graphName}.label"))
    model = HSD(graph, graphName, 0, hop, metric, signature=signature, engine=engine)
    labels = [label_dict.get(node) for node in model.nodes]
    e1, en = laplacian_extreme_eigenvalues(graph)
    scale_min, scale_max = scale_boundary(e1, en)
    score_max, scale_opt, mats = 0, 0, {}
    scales = np.linspace(scale_min, scale_max, num=5, dtype=float)
    # psi of all five scales from one Chebyshev recurrence, resident on the device (the assignment to
    # model.wavelets is the one main.py:27 forgets)
    psis = model.wavelets_for_scales(scales, approx=approx) if signature == "heat" else [None] * len(scales)
    try:
        for scale, psi in zip(scales, psis):
            model.scale = scale
            model.wavelets = psi
            dists = model.parallel_calculate_HSD(n_workers=10)
            mats[float(scale)] = dists
            if evaluate and all(l is not None for l in labels):
                score = _knn_score(dists, labels, cv=10, n_neighbor=20)
                if score is not None and score > score_max:
                    score_max, scale_opt = score, float(scale)
    finally:
        model.wavelets = None
        for psi in psis:
            if psi is not None:
                psi.free()
    print(f"max score: {score_max}, optimal scale: {scale_opt}\n")
    return score_max, scale_opt, mats


def multi_HSD_Test(graphName, hop=3, n_scales=200, data_dir="data", engine=None):
    """main.py:36-53 up to the embedding; returns node -> vector (the kNN scoring on the labels is
    the caller's, as it needs two label files)."""
    graph = read_edgelist(os.path.join(data_dir, "graph", f"{graphName}.edgelist"))
    model = MultiHSD(graph, graphName, hop, n_scales, engine=engine)
    return model.parallel_embed(n_workers=10)
