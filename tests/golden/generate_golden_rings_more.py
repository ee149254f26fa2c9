#!/usr/bin/env python3
"""More ring-membership goldens from the reference's own data files.  RUN ONLY IN THE BUILD CONTAINER (imports /root/reference).

The first fixture set covers the small graphs; these are the shipped files with the awkward line formats SURVEY 8(d) lists:
PTC (every edge listed twice, once per direction), citeseer (some lines carry a third column), usa and cora (a thousand nodes and more,
ids that do not start at 0).  For each: the node order networkx gives (model/HSD.py:37), tools.get_hierarchical_representation at
K = 2 as sorted member lists, and the ring sizes at K = 3 -- the bit-exact target of the k-hop sweep.

    python tests/golden/generate_golden_rings_more.py
"""
import contextlib
import io
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.modules.setdefault("pygsp", types.ModuleType("pygsp"))

import networkx as nx                                     # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    import tools as ref_tools                             # noqa: E402

from oracle import c_oracle as corc                       # noqa: E402
from hsd_amd.graph import read_edgelist                   # noqa: E402

GRAPHS = ["PTC", "citeseer", "usa", "cora"]


def main():
    gdir = os.path.join(HERE, "graphs")
    out = {}
    for g in GRAPHS:
        shutil.copyfile(f"{REF}/data/graph/{g}.edgelist", f"{gdir}/{g}.edgelist")
        G = nx.read_edgelist(f"{gdir}/{g}.edgelist", create_using=nx.Graph, edgetype=float, data=[("weight", float)])   # main.py:10-11
        mine = read_edgelist(f"{gdir}/{g}.edgelist")
        nodes = list(G.nodes())
        assert nodes == mine.nodes, g
        n2i = {v: i for i, v in enumerate(nodes)}
        out[f"{g}/nodes"] = np.asarray(nodes)
        out[f"{g}/n_edges"] = np.asarray([G.number_of_edges()])
        assert G.number_of_edges() == mine.n_edge, (g, G.number_of_edges(), mine.n_edge)
        with contextlib.redirect_stdout(io.StringIO()):
            h2 = ref_tools.get_hierarchical_representation(G, 2)
            h3 = ref_tools.get_hierarchical_representation(G, 3)
        offs, mem = [0], []
        for v in nodes:
            for layer in h2[v]:
                mem.extend(sorted(n2i[u] for u in layer))
                offs.append(len(mem))
        out[f"{g}/K2/offsets"] = np.asarray(offs, dtype=np.int64)
        out[f"{g}/K2/members"] = np.asarray(mem, dtype=np.int32)
        sizes3 = np.asarray([[len(layer) for layer in h3[v]] for v in nodes], dtype=np.int32)
        out[f"{g}/K3/sizes"] = sizes3
        assert np.array_equal(sizes3, corc.khop_sizes(mine.rowptr, mine.col, 3)), g       # the C oracle agrees with the reference
        print(g, len(nodes), "nodes", G.number_of_edges(), "edges; ring sizes at K = 3 (mean)", sizes3.mean(axis=0).round(1))
    np.savez_compressed(os.path.join(HERE, "rings_more.npz"), **out)
    print("rings_more.npz written", os.path.getsize(os.path.join(HERE, "rings_more.npz")), "bytes")


if __name__ == "__main__":
    main()
