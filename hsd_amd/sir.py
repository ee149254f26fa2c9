"""SIR structural-influence labels -- a by-product of the k-hop sweep.

The reference (tools/SIR.py:43-63) runs, per origin, a deterministic breadth-first "infection"
for t rounds and adds e^(-0.5*round) for every node infected in that round, i.e.
    influence(i) = sum_{k=0}^{t-1} |ring_k(i)| * e^(-0.5 k),
so it needs nothing but the ring sizes hsd_khop_sizes already produces on the GPU.
(`p`, the infection probability, is stored but never used by the reference either.)
"""
from __future__ import annotations

import math
from collections import defaultdict

import numpy as np

from .engine import Engine
from .graph import Graph


class SIR:
    def __init__(self, graph, p=1.0, t=25, random_state=42, engine=None, device=0):
        """tools/SIR.py:14-27."""
        self.G = graph
        self.p = p
        self.t = t
        self.random_state = random_state
        self._g = Graph.coerce(graph)
        self.idx2node = {i: v for i, v in enumerate(self._g.nodes)}
        self.node2idx = dict(self._g.node2idx)
        self.influence = defaultdict(int)
        self.engine = engine if engine is not None else Engine(device)

    def start(self):
        """tools/SIR.py:36-39 for all origins at once."""
        if self.t < 1:
            return
        self.engine.set_graph(self._g.rowptr, self._g.col)
        sizes = self.engine.khop_sizes(self.t - 1).astype(np.float64)          # rounds 0 .. t-1
        w = np.array([math.e ** (-0.5 * k) for k in range(self.t)])
        infl = sizes @ w
        for i, v in self.idx2node.items():
            self.influence[v] += float(infl[i])

    def label_nodes(self, n_class):
        """tools/SIR.py:66-81: sort by influence (stable), cut into n_class groups."""
        order = sorted(self.influence, key=self.influence.get)      # ascending influence, ties in insertion order
        width = len(order) / n_class
        labels, current = {}, 1
        for position, node in enumerate(order):
            labels[node] = current
            current += position >= width * current                 # at most one step per node, as in the reference
        return labels


def split_nodes(graph, n_class, infect_prob, t=5, save_path=None, engine=None) -> dict:
    """Influence classes of every node, optionally written as a label file ordered by integer node id
    (what tools/SIR.py:83-93 produces)."""
    labels = get_SIR_labels(graph, n_class, t, infect_prob, engine=engine)
    if save_path:
        lines = [f"{node} {labels[node]}\n" for node in sorted(labels, key=int)]
        with open(save_path, "w", encoding="utf-8") as fout:
            fout.writelines(lines)
    return labels


def get_SIR_labels(g, n: int, t: int, infect_p: float, engine=None) -> dict:
    """Shortcut of tools/SIR.py:95-100: t rounds from every origin, then n influence classes."""
    model = SIR(g, p=infect_p, t=t, engine=engine)
    model.start()
    return model.label_nodes(n)
