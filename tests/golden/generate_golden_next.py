#!/usr/bin/env python3
"""Golden vectors for the rows SURVEY.md 8(f) marks "next": SIR influence labels, the `.layers`
hierarchy file format, the embeddings CSV / distance edgelist formats, and the scale sweep of the
driver.  RUN ONLY IN THE BUILD CONTAINER (imports the reference); writes next_rows.json.
(The member order inside a ring of the `.layers` text is the iteration order of a Python set of strings, tools/hierarchy.py:33-35, i.e. it
changes from run to run with the interpreter's hash seed; the committed `layers` section is one such run -- the tests compare the reader on
that text and the writer ring by ring as sets.  Every other section is reproduced bit for bit.)"""
import contextlib
import io
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.modules.setdefault("pygsp", types.ModuleType("pygsp"))
np.float = float                                   # the reference predates numpy 1.24 (tools/rw.py:58,78)

import networkx as nx                              # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from tools import SIR as ref_sir               # noqa: E402
    from tools import hierarchy as ref_h           # noqa: E402
    from tools import rw as ref_rw                 # noqa: E402
    from tools import util as ref_util             # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def load(name):
    return nx.read_edgelist(f"{HERE}/graphs/{name}.edgelist", create_using=nx.Graph, edgetype=float, data=[("weight", float)])


def main():
    out = {}
    # ---- SIR influence and labels (tools/SIR.py:13-82) ----
    sir = {}
    for name, t, ncls in [("karate", 25, 4), ("tree", 5, 5), ("barbell", 3, 3), ("europe", 5, 4)]:
        G = load(name)
        m = ref_sir.SIR(G, 1.0, t)
        quiet(m.start)
        labels = m.label_nodes(ncls)
        sir[name] = {"t": t, "n_class": ncls, "influence": {str(k): float(v) for k, v in m.influence.items()},
                     "labels": {str(k): int(v) for k, v in labels.items()}}
    out["sir"] = sir

    # ---- .layers files (tools/hierarchy.py:41-63 writer, :76-99 reader) ----
    layers = {}
    with tempfile.TemporaryDirectory() as d:
        for name, hop_w, hop_r in [("karate", 7, 3), ("tree", 3, 5), ("line", 2, 3)]:
            G = load(name)
            path = os.path.join(d, f"{name}.layers")
            quiet(ref_h.save_hierarchical_representation, G, path, hop_w)
            text = open(path).read()
            parsed = quiet(ref_h.read_hierarchy, path, hop_r)
            layers[name] = {"hop_written": hop_w, "hop_read": hop_r, "text": text,
                            "parsed": {str(k): v for k, v in parsed.items()}}
        # ---- rw formats ----
        rng = np.random.default_rng(5)
        nodes = ["3", "1", "10", "7"]
        vecs = rng.normal(size=(4, 6)).tolist()
        p = os.path.join(d, "v.csv")
        ref_rw.save_vectors(nodes, vecs, p)
        csv_text = open(p).read()
        back = ref_rw.read_vectors(p)
        D = rng.random((4, 4))
        D = (D + D.T) / 2
        np.fill_diagonal(D, 0.0)
        p2 = os.path.join(d, "d.edgelist")
        ref_rw.save_distance_edgelist(p2, ["0", "1", "2", "3"], D)
        el_text = open(p2).read()
        Dback = ref_rw.read_distance(p2, 4)
        out["rw"] = {"nodes": nodes, "vectors": vecs, "csv_text": csv_text, "read_vectors": {k: [float(x) for x in v] for k, v in back.items()},
                     "D": D.tolist(), "edgelist_text": el_text, "read_distance": Dback.tolist()}
    out["layers"] = layers

    # ---- driver: scale range and the 5-scale sweep of main.py:21-25 ----
    drv = {}
    for name in ("karate", "barbell", "mkarate"):
        G = load(name)
        lam = np.linalg.eigvalsh(np.asarray(nx.laplacian_matrix(G).todense(), dtype=float))
        smin, smax = ref_util.recommend_scale_range(list(lam))
        drv[name] = {"scale_min": float(smin), "scale_max": float(smax),
                     "scales": [float(s) for s in np.linspace(smin, smax, num=5)],
                     "lambda_2": float(sorted(x for x in lam if x > 0.001)[0]), "lambda_max": float(lam[-1])}
    out["driver"] = drv
    # top-5 % edge filter (tools/util.py:133-159) on a small symmetric matrix
    rng = np.random.default_rng(9)
    Dm = rng.random((12, 12))
    Dm = (Dm + Dm.T) / 2
    np.fill_diagonal(Dm, 0.0)
    kept = ref_util.filter_distance_matrix(Dm, [str(i) for i in range(12)], None, 0.05)
    out["filter"] = {"D": Dm.tolist(), "kept": [[a, b, float(c)] for a, b, c in kept]}
    # the same with tied distances (a descending sort that keeps equal entries in row-major order), a save path (tools/rw.py:101-116) and
    # the in-place behaviour of filter_edgelist (tools/util.py:133-144)
    Dt = np.round(rng.random((9, 9)) * 4) / 4
    Dt = np.maximum(Dt, Dt.T)
    np.fill_diagonal(Dt, 0.0)
    names = [f"v{i}" for i in range(9)]
    with tempfile.TemporaryDirectory() as d:
        sp = os.path.join(d, "kept.edgelist")
        kept_t = ref_util.filter_distance_matrix(Dt, names, sp, 0.3)
        text = open(sp).read()
    edges = [("a", "b", 0.5), ("c", "d", 2.0), ("e", "f", 0.5), ("g", "h", 2.0), ("i", "j", 1.0), ("k", "l", 0.25), ("m", "n", 2.0)]
    edges_in = [list(e) for e in edges]
    edges_kept = ref_util.filter_edgelist(edges, None, 0.3)
    out["filter_saved"] = {"D": Dt.tolist(), "nodes": names, "ratio": 0.3, "kept": [[a, b, float(c)] for a, b, c in kept_t], "text": text,
                           "edges_in": edges_in, "edges_kept": [list(e) for e in edges_kept], "edges_sorted_in_place": [list(e) for e in edges]}
    # embeddings with a missing value (pandas writes an empty field) and a distance file with a blank line (a ValueError in read_distance)
    with tempfile.TemporaryDirectory() as d:
        pv = os.path.join(d, "nan.csv")
        ref_rw.save_vectors(["5", "6"], [[1.0, float("nan"), 2.5], [0.125, 3.0, float("nan")]], pv)
        nan_text = open(pv).read()
        pd_ = os.path.join(d, "blank.edgelist")
        open(pd_, "w").write("0 1 0.5\n\n1 2 0.25\n")
        try:
            ref_rw.read_distance(pd_, 3)
            blank = "no error"
        except Exception as e:                          # noqa: BLE001
            blank = type(e).__name__
    out["rw_edge_cases"] = {"nan_vectors_text": nan_text, "read_distance_blank_line": blank}

    with open(os.path.join(HERE, "next_rows.json"), "w") as f:
        json.dump(out, f)
    print("wrote next_rows.json", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
