"""The rows SURVEY.md 8(f) marks "next" -- SIR labels, `.layers` / CSV / edgelist formats, the
driver's scale sweep and top-5 % filter -- against goldens produced by the reference
(tests/golden/generate_golden_next.py)."""
import json
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN, load_graph

with open(os.path.join(GOLDEN, "next_rows.json")) as f:
    G = json.load(f)


# ---------------------------------------------------------------- CPU: formats and host helpers
def test_read_hierarchy_quirks(tmp_path, capsys):
    from hsd_amd.formats import read_hierarchy
    for name, rec in G["layers"].items():
        p = tmp_path / f"{name}.layers"
        p.write_text(rec["text"])
        got = read_hierarchy(str(p), rec["hop_read"])
        assert {str(k): v for k, v in got.items()} == rec["parsed"]
    assert G["layers"]["line"]["parsed"]["0"][3] == ['']          # the trailing-'#' quirk is in the golden
    with pytest.raises(FileNotFoundError):
        read_hierarchy(str(tmp_path / "missing.layers"), 3)


def test_vectors_csv_and_distance_edgelist(tmp_path):
    from hsd_amd import formats
    r = G["rw"]
    p = tmp_path / "v.csv"
    formats.save_vectors(r["nodes"], r["vectors"], str(p))
    assert p.read_text() == r["csv_text"]
    assert formats.read_vectors(str(p)) == r["read_vectors"]
    p2 = tmp_path / "d.edgelist"
    formats.save_distance_edgelist(str(p2), ["0", "1", "2", "3"], np.asarray(r["D"]))
    assert p2.read_text() == r["edgelist_text"]
    assert np.array_equal(formats.read_distance(str(p2), 4), np.asarray(r["read_distance"]))
    assert formats.read_label(os.path.join(GOLDEN, "graphs", "barbell.label"))["1"] >= 0
    assert formats.read_label(str(tmp_path / "none.label")) == {}
    # a missing value is an empty field (pandas), a blank line in a distance file is the reference's ValueError
    e = G["rw_edge_cases"]
    pn = tmp_path / "nan.csv"
    formats.save_vectors(["5", "6"], [[1.0, float("nan"), 2.5], [0.125, 3.0, float("nan")]], str(pn))
    assert pn.read_text() == e["nan_vectors_text"]
    pb = tmp_path / "blank.edgelist"
    pb.write_text("0 1 0.5\n\n1 2 0.25\n")
    assert e["read_distance_blank_line"] == "ValueError"
    with pytest.raises(ValueError):
        formats.read_distance(str(pb), 3)


def test_scale_range_and_filter(tmp_path):
    from hsd_amd import driver
    for name, rec in G["driver"].items():
        g = load_graph(name)
        e1, en = driver.laplacian_extreme_eigenvalues(g)
        assert abs(e1 - rec["lambda_2"]) < 1e-10 and abs(en - rec["lambda_max"]) < 1e-10
        smin, smax = driver.scale_boundary(e1, en)
        assert abs(smin - rec["scale_min"]) < 1e-14 and abs(smax - rec["scale_max"]) < 1e-14
        lam = np.linalg.eigvalsh(g.laplacian_dense())
        assert driver.recommend_scale_range(list(lam)) == (rec["scale_min"], rec["scale_max"])
    D = np.asarray(G["filter"]["D"])
    kept = driver.filter_distance_matrix(D, [str(i) for i in range(12)], None, 0.05)         # (the reference's positional order)
    assert [[a, b, c] for a, b, c in kept] == G["filter"]["kept"]
    # with ties (equal distances keep their row-major order under the descending sort) and saved as an edgelist (tools/rw.py:101-116)
    rec = G["filter_saved"]
    Dt = np.asarray(rec["D"])
    p = tmp_path / "kept.edgelist"
    kept = driver.filter_distance_matrix(Dt, rec["nodes"], str(p), rec["ratio"])
    assert [[a, b, c] for a, b, c in kept] == rec["kept"] and p.read_text() == rec["text"]
    edges = [tuple(e) for e in rec["edges_in"]]
    assert [list(e) for e in driver.filter_edgelist(edges, None, rec["ratio"])] == rec["edges_kept"]
    assert [list(e) for e in edges] == rec["edges_sorted_in_place"]


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["karate", "tree", "barbell", "europe"])
def test_sir_influence_and_labels(engine, name):
    from hsd_amd import sir
    rec = G["sir"][name]
    g = load_graph(name)
    m = sir.SIR(g, 1.0, rec["t"], engine=engine)
    m.start()
    for v, ref in rec["influence"].items():
        assert abs(m.influence[v] - ref) <= 1e-12 * max(1.0, abs(ref)), (v, m.influence[v], ref)
    assert {str(k): v for k, v in m.label_nodes(rec["n_class"]).items()} == rec["labels"]
    assert {str(k): v for k, v in sir.get_SIR_labels(g, rec["n_class"], rec["t"], 1.0, engine=engine).items()} == rec["labels"]


@pytest.mark.gpu
def test_layers_writer_roundtrip(engine, tmp_path, capsys):
    from hsd_amd import formats
    for name, rec in G["layers"].items():
        g = load_graph(name)
        p = tmp_path / f"{name}.layers"
        formats.save_hierarchical_representation(g, str(p), rec["hop_written"], engine=engine)
        mine = formats.read_hierarchy(str(p), rec["hop_read"])
        ref = rec["parsed"]
        assert set(mine) == set(ref)
        for v in ref:
            assert [sorted(x) for x in mine[v]] == [sorted(x) for x in ref[v]]
        # same line structure as the reference's file: number of '#' per line
        assert [ln.count("#") for ln in p.read_text().splitlines()] == [ln.count("#") for ln in rec["text"].splitlines()]


@pytest.mark.gpu
def test_driver_flow_on_karate(engine, tmp_path, capsys):
    """base_HSD_Test (main.py:9-34 as intended): 5 scales from the Laplacian spectrum, Variant B per
    scale; the distance matrices equal the oracle's reading of model/HSD.py:140-161."""
    from hsd_amd import driver
    from oracle import hsd_oracle as orc
    os.makedirs(tmp_path / "graph")
    os.makedirs(tmp_path / "label")
    shutil.copy(os.path.join(GOLDEN, "graphs", "mkarate.edgelist"), tmp_path / "graph" / "mkarate.edgelist")
    shutil.copy(os.path.join(GOLDEN, "graphs", "mkarate.label"), tmp_path / "label" / "mkarate.label")
    score, scale_opt, mats = driver.base_HSD_Test("mkarate", hop=3, metric="wasserstein", data_dir=str(tmp_path),
                                                  approx=False, engine=engine, evaluate=False)
    assert [round(s, 15) for s in mats] == [round(s, 15) for s in G["driver"]["mkarate"]["scales"]]
    g = load_graph("mkarate")
    rings = orc.khop_rings(g.rowptr, g.col, 3)
    s = sorted(mats)[2]
    psi = orc.heat_wavelets_exact(g.laplacian_dense(), s)
    ref = orc.distance_variant_b_rowquirk(psi, rings, 3)
    assert np.allclose(mats[s], ref, rtol=1e-6, atol=1e-12)


def test_read_label_swallows_malformed_lines_like_the_reference(tmp_path):
    from hsd_amd import formats
    # tools/dataloader.py:57-75 returns from `finally:`: a malformed line ends the read, the labels so far come back
    p = tmp_path / "bad.label"
    p.write_text("a 1\nb 2\nc  3\nd 4\n")          # third line has two blanks -> split(" ") yields three fields
    assert formats.read_label(str(p)) == {"a": 1, "b": 2}
    p.write_text("a 1\nb x\nc 3\n")
    assert formats.read_label(str(p)) == {"a": 1}
    # invalid UTF-8 raises inside the file iteration: the reference's `finally: return` swallows that too
    p.write_bytes(b"a 1\nb 2\n\xff\xfe 3\nd 4\n")
    got = formats.read_label(str(p))
    assert isinstance(got, dict) and set(got) <= {"a", "b"}
