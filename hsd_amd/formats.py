"""On-disk formats of the reference, for interchange with artefacts it
produced: the `.layers` hierarchy file (tools/hierarchy.py:41-99), the embeddings CSV
(tools/rw.py:13-47), the distance edgelist (tools/rw.py:50-98) and label files
(tools/dataloader.py:57-75).  Plain Python -- these are file formats, not the hot path; the
hierarchy itself comes from the GPU sweep.
"""
from __future__ import annotations

import os

import numpy as np

from .graph import Graph


# ---- .layers ----------------------------------------------------------------------------------
def save_hierarchical_representation(graph, file_path: str, hop=7, engine=None):
    """tools/hierarchy.py:41-63: one line per node, `node#a,b#c,d,e#`: rings joined by '#',
    members by ',', a trailing '#'; the line stops at the first empty ring.  Members are written
    in ascending index order (the reference writes Python-set order; readers do not care)."""
    from .engine import Engine
    g = Graph.coerce(graph)
    eng = engine if engine is not None else Engine(0)
    eng.set_graph(g.rowptr, g.col)
    offsets, members = eng.khop_members(hop)
    K1 = hop + 1
    with open(file_path, encoding="utf-8", mode="w+") as fout:
        for i in range(g.n_node):
            record = ""
            for k in range(K1):
                level = members[offsets[i * K1 + k]:offsets[i * K1 + k + 1]]
                if len(level) == 0:
                    break
                record += ",".join(str(g.nodes[u]) for u in level) + "#"
            fout.write(record + "\n")


def read_hierarchy(file_path: str, maxHop: int) -> dict:
    """Reader of the `.layers` format (behaviour of tools/hierarchy.py:76-99).  Every record is cut at
    '#' and the first maxHop + 1 fields become rings (split at ','); fields a record does not have are
    empty rings.  The writer's trailing '#' therefore leaves one empty FIELD behind the last ring, which
    comes back as [''] -- a quirk readers of the reference's files depend on (MultiHSD.get_triple skips
    the '' token), so it is reproduced.  Reading stops at the first blank line."""
    if not os.path.isfile(file_path):
        raise FileNotFoundError(f"path:{file_path}, hierarchy file not exist")
    n_rings = maxHop + 1
    hierarchy = {}
    with open(file_path, encoding="utf-8") as fin:
        for raw in fin:
            record = raw.strip()
            if record == "":
                break
            fields = record.split("#")[:n_rings]
            rings = [f.strip().split(",") for f in fields] + [[] for _ in range(n_rings - len(fields))]
            hierarchy[rings[0][0]] = rings
    print(f"done, number of nodes: {len(hierarchy)}")
    return hierarchy


# ---- embeddings CSV ---------------------------------------------------------------------------
def save_vectors(nodes: list, vectors: list, path: str):
    """tools/rw.py:13-22 (pandas to_csv(header=False, float_format='%.8f'))."""
    with open(path, "w", encoding="utf-8") as f:
        for n, v in zip(nodes, vectors):
            f.write(",".join([str(n)] + ["" if x != x else "%.8f" % float(x) for x in v]) + "\n")       # (NaN: an empty field, as pandas writes it)


def save_vectors_dict(vectors: dict, path: str):
    """tools/rw.py:25-31."""
    save_vectors(list(vectors.keys()), list(vectors.values()), path)


def read_vectors(path: str) -> dict:
    """tools/rw.py:34-47: keys are str(int(first column))."""
    if not os.path.exists(path):
        raise FileNotFoundError
    out = {}
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            s = line.strip().split(",")
            if len(s) < 2:
                continue
            out[str(int(float(s[0])))] = [float(x) for x in s[1:]]
    return out


# ---- distance edgelist ------------------------------------------------------------------------
def save_distance_edgelist(path: str, nodes: list, mat):
    """tools/rw.py:84-98: `u v d` for every i < j (Python float repr)."""
    n = len(mat)
    with open(path, mode="w+", encoding="utf-8") as fout:
        for idx1, node1 in enumerate(nodes):
            for idx2 in range(idx1 + 1, n):
                fout.write(f"{node1} {nodes[idx2]} {float(mat[idx1, idx2])!r}\n")


def read_distance(path: str, n_nodes: int) -> np.ndarray:
    """tools/rw.py:50-65 (node labels are integer indices).  Every line up to the end of the file must be `u v d` separated by single
    blanks: a blank line is a ValueError there (it reads to EOF and unpacks three fields), and so it is here."""
    mat = np.zeros((n_nodes, n_nodes), dtype=float)
    with open(path, mode="r", encoding="utf-8") as fin:
        for line in fin:
            u, v, dist = line.strip().split(" ")
            mat[int(u), int(v)] = mat[int(v), int(u)] = float(dist)
    return mat


def save_edgelist(path: str, edgelist: list):
    """tools/rw.py:101-116: `u v` or `u v w` per edge; tuples shorter than two are skipped, longer than three leave an empty line."""
    with open(path, mode="w+", encoding="utf-8") as fout:
        for edge in edgelist:
            if len(edge) < 2:
                continue
            text = f"{edge[0]} {edge[1]}" if len(edge) == 2 else f"{edge[0]} {edge[1]} {edge[2]}" if len(edge) == 3 else ""
            fout.write(f"{text}\n")


# ---- labels -----------------------------------------------------------------------------------
def read_label(path) -> dict:
    """`node label` per line, separated by one blank (behaviour of tools/dataloader.py:57-75): labels are
    ints, node ids stay strings, reading stops at the first blank line, a missing file is reported on
    stdout and yields {} instead of raising -- and, because the reference returns from a `finally:` block,
    ANY error while parsing (a line with more than one blank, a label that is not an int) is swallowed and the
    labels read so far are returned.  That quirk is kept: files the reference loads (partially) load the same here."""
    labels = {}
    if not os.path.isfile(path):
        print(f"Lable file not exist, path:{path}")
        return labels
    try:
        with open(path, encoding="utf-8") as fin:
            for raw in fin:                 # (a file with invalid UTF-8 raises here, inside the try, as in the reference)
                record = raw.strip()
                if record == "":
                    break
                node, label = record.split(" ")
                labels[node] = int(label)
    except Exception:                       # the reference's `finally: return label_dict`: partial result, no exception
        pass
    return labels
