"""Pin the oracle (oracle/gnnvc_oracle.c) against outputs of the reference itself.

Golden vectors: tests/golden/README.md.  The reference has no tests of its own
(SURVEY.md §4); these survey-time outputs of the unmodified reference are the
only known answers for the path.
"""
import hashlib
import json

import numpy as np
import pytest

from oracle import oracle_py
from tools import graphgen as gg


@pytest.fixture(scope="module")
def manifest(golden_dir):
    return json.loads((golden_dir / "manifest.json").read_text())


def _ulp_diff(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    return np.abs(ai - bi)


def _check_scores(scores: np.ndarray, gold: np.ndarray, gold_md5: str):
    assert scores.dtype == np.float32 and scores.shape == gold.shape
    if hashlib.md5(scores.tobytes()).hexdigest() == gold_md5:
        return 0
    # host expf may differ in the last ulp on another CPU (tests/golden/README.md)
    d = _ulp_diff(scores, gold)
    assert d.max() <= 1, f"oracle differs from reference by {d.max()} ulp"
    return int((d > 0).sum())


def test_golden_files_intact(golden_dir, manifest):
    for key in ("ex3", "er100k", "hub200k"):
        raw = (golden_dir / manifest[key]["scores_file"]).read_bytes()
        assert hashlib.md5(raw).hexdigest() == manifest[key]["scores_md5"]


def test_readme_three_vertex_graph(oracle_model, golden_dir, manifest):
    spec = manifest["ex3"]
    g = gg.from_edge_list(spec["graph"]["n"], spec["graph"]["edges"], spec["graph"]["weights"])
    assert g.ws == spec["ws"]
    oracle_model.set_weight_scale(g.ws)
    s = oracle_model.scores(g)
    gold = np.fromfile(golden_dir / spec["scores_file"], dtype=np.float32)
    assert [hex(v) for v in gold.view(np.uint32)] == spec["scores_hex"]
    assert _check_scores(s, gold, spec["scores_md5"]) == 0


def test_er100k_bitwise(oracle_model, golden_dir, manifest):
    spec = manifest["er100k"]
    p = spec["graph"]
    g = gg.erdos_renyi(p["n"], p["m"], p["seed"])
    # the regenerated input is byte-identical to the file the reference read
    assert gg.metis_md5(g) == spec["metis_md5"]
    oracle_model.set_weight_scale(g.ws)
    s = oracle_model.scores(g)
    gold = np.fromfile(golden_dir / spec["scores_file"], dtype=np.float32)
    mism = _check_scores(s, gold, spec["scores_md5"])
    assert mism <= 8, f"{mism} scores differ by 1 ulp"
    np.testing.assert_allclose(gold[:4], spec["first_scores"], rtol=0, atol=1e-7)
    assert abs(float(gold.astype(np.float64).sum()) - spec["scores_sum_f64"]) < 1e-4
    # the parallel-aggregation variant is the same arithmetic row by row
    s2 = oracle_model.predict(g, g.x(), parallel_agg=True)[:, 0]
    assert np.array_equal(s2.view(np.uint32), s.view(np.uint32))


def test_hub_graph_bitwise(oracle_model, golden_dir, manifest):
    spec = manifest["hub200k"]
    p = spec["graph"]
    g = gg.hub_graph(p["n"], p["m"], p["hubs"], p["hub_degree"], p["seed"])
    assert g.n_edges == spec["n_edges"]
    assert gg.metis_md5(g) == spec["metis_md5"]
    deg = np.diff(g.rowptr.astype(np.int64))
    assert deg[:3].min() >= 65536  # the planted hubs exercise 65536-term CSR-order sums
    oracle_model.set_weight_scale(g.ws)
    s = oracle_model.scores(g)
    gold = np.fromfile(golden_dir / spec["scores_file"], dtype=np.float32)
    mism = _check_scores(s, gold, spec["scores_md5"])
    assert mism <= 16


def _layer_graph(spec):
    p = spec["graph"]
    if p["kind"] == "edge_list":
        return gg.from_edge_list(p["n"], p["edges"], p["weights"])
    if p["kind"] == "erdos_renyi":
        return gg.erdos_renyi(p["n"], p["m"], p["seed"])
    return gg.hub_graph(p["n"], p["m"], p["hubs"], p["hub_degree"], seed=p["seed"])


@pytest.mark.parametrize("name", ["ex3", "er4k", "hub2k"])
def test_layer_boundary_fixtures(oracle_model, golden_dir, name):
    """Activations after the first and second fused block (N x 16), logits and scores, written by
    tools/make_golden_layers.py from the reference's own layer code with its products in genuine OpenBLAS
    (tests/golden/manifest_layers.json): the oracle reproduces every one bit for bit."""
    man = json.loads((golden_dir / "manifest_layers.json").read_text())
    spec = man["graphs"][name]
    g = _layer_graph(spec)
    assert gg.metis_md5(g) == spec["metis_md5"] and g.ws == spec["ws"]
    oracle_model.set_weight_scale(g.ws)
    for what, f in spec["files"].items():
        raw = (golden_dir / f["file"]).read_bytes()
        assert hashlib.md5(raw).hexdigest() == f["md5"]
        gold = np.frombuffer(raw, dtype=np.float32).reshape(f["shape"])
        got = oracle_model.predict(g, g.x(), stop_after=f["after_layer_index"])
        if what == "scores2":     # (host expf: see _check_scores)
            assert _ulp_diff(np.ascontiguousarray(got), gold).max() <= 1
        else:
            assert np.array_equal(got.view(np.uint32), gold.view(np.uint32)), what
    # and the committed recipe reproduced the survey-time score files when it ran
    assert all(v["equals_recorded_md5"] for v in man["reproduced_survey_scores"].values())


def test_graph_layer_column_layout():
    """The F+1..F+3 quirk (reference src/gnn_inference.cpp:37-40), on the survey's
    probe input: in(u, j) = (u+1)*10 + j on the README graph, ws = 20."""
    g = gg.from_edge_list(3, [(0, 2), (1, 2)], [15, 15, 20])
    for f in (1, 4, 16):
        h = np.array([[(u + 1) * 10 + j for j in range(f)] for u in range(3)], dtype=np.float32)
        out = oracle_py.graph_layer(g, 20.0, h)
        assert out.shape == (3, 2 * f + 3)
        exp = np.zeros((3, 2 * f + 3), dtype=np.float32)
        exp[0, :f] = h[2]
        exp[1, :f] = h[2]
        exp[2, :f] = h[0] + h[1]
        exp[:, f:2 * f] = h
        exp[:, f + 1] = [1, 1, 2]
        exp[:, f + 2] = np.array([15, 15, 20], dtype=np.float32) / np.float32(20)
        exp[:, f + 3] = np.array([20, 20, 30], dtype=np.float32) / np.float32(20)
        assert np.array_equal(out, exp)
        if f == 16:
            assert np.all(out[:, 32:35] == 0)       # never written
            assert np.all(out[:, 16] == h[:, 0])    # only h[0] and h[4..15] survive
            assert np.all(out[:, 20:32] == h[:, 4:])


def test_empty_graph_is_noop(oracle_model):
    """predict is called with N = 0 at the end of every CLI run."""
    g = gg.from_edge_list(0, [], [])
    out = oracle_model.predict(g, np.zeros((0, 1), dtype=np.float32))
    assert out.shape == (0, 1)


def test_isolated_vertices(oracle_model):
    g = gg.from_edge_list(5, [(0, 1)], [20, 30, 40, 50, 120])
    oracle_model.set_weight_scale(g.ws)
    s = oracle_model.scores(g)
    assert s.shape == (5,) and np.all((s > 0) & (s < 1))
    # isolated vertices with different weights still get their own MLP result
    assert len({float(v) for v in s[2:]}) == 3


def test_model_parser_shapes(oracle_model):
    kinds = oracle_model.layer_kinds()
    assert len(kinds) == 21 and kinds.count(1) == 3 and kinds.count(0) == 9
    shapes = [W.shape for W, _ in oracle_model.linear_params()]
    assert shapes == [(5, 32), (32, 32), (32, 16), (35, 32), (32, 32), (32, 16),
                      (35, 32), (32, 16), (16, 1)]
    assert sum(W.size + b.size for W, b in oracle_model.linear_params()) == 6209


def test_metis_roundtrip():
    g = gg.erdos_renyi(500, 2000, 3)
    g2 = gg.parse_metis(gg.metis_text(g))
    for a, b in ((g.rowptr, g2.rowptr), (g.col, g2.col), (g.w, g2.w), (g.nw, g2.nw)):
        assert np.array_equal(a, b)
