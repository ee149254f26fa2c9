"""The helper functions of the reference's `tools` package that sit on the hot path
(SURVEY.md 8(b)): `get_hierarchical_representation` (tools/hierarchy.py:16-22) and
`metrics.calculate_distance` (tools/metrics.py:151-192), served by the device library."""
from __future__ import annotations

import types

from . import _lib as L
from .engine import Engine
from .graph import Graph

_default_engine = None


def _engine(engine=None):
    global _default_engine
    if engine is not None:
        return engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


def get_hierarchical_representation(graph, maxHop, engine=None):
    """dict node -> list of maxHop+1 lists: layers[0] = [node], layers[k] = nodes first reached at
    depth k, empty lists once the component is exhausted (tools/hierarchy.py:16-38).  Members are
    listed in ascending index order (the reference's order is Python-set order)."""
    g = Graph.coerce(graph)
    eng = _engine(engine)
    eng.set_graph(g.rowptr, g.col)
    offsets, members = eng.khop_members(maxHop)
    K1 = maxHop + 1
    out = {}
    for i, v in enumerate(g.nodes):
        out[v] = [[g.nodes[u] for u in members[offsets[i * K1 + k]:offsets[i * K1 + k + 1]]] for k in range(K1)]
    print(f"done, number of nodes: {len(out)}")            # tools/hierarchy.py:21
    return out


def get_node_hierarchical_structure(graph, node, maxHop: int, engine=None):
    """tools/hierarchy.py:25-38 for ONE node: [[node], ring 1, ..., ring maxHop] (always maxHop + 1 lists, empty once the component
    is exhausted), one k-hop sweep from that source on the device; members in ascending index order."""
    g = Graph.coerce(graph)
    eng = _engine(engine)
    eng.set_graph(g.rowptr, g.col)
    i = g.node2idx[node]
    offsets, members = eng.khop_members(maxHop, i, i + 1)
    return [[g.nodes[u] for u in members[offsets[k]:offsets[k + 1]]] for k in range(maxHop + 1)]


def build_node_idx_map(graph):
    """tools/util.py:11-24: (idx2node, node2idx) in the graph's node order -- the row / column order of every matrix."""
    g = Graph.coerce(graph)
    return {i: v for i, v in enumerate(g.nodes)}, dict(g.node2idx)


_SUPPORTED = ['l1', 'l2', 'kl', 'symmetric_kl', 'js', 'wasserstein_guass', 'wasserstein', 'hellinger']
_CODE = {"wasserstein": L.METRIC_WASSERSTEIN, "kl": L.METRIC_KL, "symmetric_kl": L.METRIC_SYMMETRIC_KL}


def calculate_distance(p, q, metric, engine=None):
    """tools/metrics.py:151-192 for the metrics on the accelerated path."""
    if not metric or not isinstance(metric, str):
        raise TypeError("Need to specify a metric.")
    metric = metric.lower()
    if metric not in _SUPPORTED:
        raise NotImplementedError("{} metric is not implemented.".format(metric))
    if metric not in _CODE:
        raise NotImplementedError("{} metric is outside the accelerated hot path.".format(metric))
    return _engine(engine).sample_distance(list(p), list(q), _CODE[metric])


metrics = types.SimpleNamespace(calculate_distance=calculate_distance)

# the rest of what `from tools import *` gives a caller of the reference on this path (tools/__init__.py:2-4: hierarchy.*, util.*)
from .formats import read_hierarchy, read_label, save_hierarchical_representation          # noqa: E402,F401
from .driver import filter_distance_matrix, filter_edgelist, recommend_scale_range, scale_boundary      # noqa: E402,F401
