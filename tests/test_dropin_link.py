"""Link-level drop-in checks (build container only: needs /root/reference).

oracle/Makefile `make ref` compiles the reference's OWN translation units where
they lie and links them with this repo's host mirror:

  ref_layers.so       reference layers/predict  + our matrix TU  -> the oracle must agree
                      bit for bit with the reference's own code (quirk, sequencing, parser)
  GNN_VC_reflayers    reference driver + reference layers + our matrix TU
  GNN_VC_dropin       reference driver + OUR gnn_inference.cpp + OUR matrix.cpp

with the C ABI served by the oracle-backed test double (no GPU here).  The CLI
known answers (tests/golden/manifest.json) come from the unmodified reference
linked to real OpenBLAS, so matching them also confirms dot() == cblas_sgemm at
every shape the run exercised.
"""
import ctypes as C
import hashlib
import json
import pathlib
import subprocess

import numpy as np
import pytest

from oracle import oracle_py
from tools import graphgen as gg

ROOT = pathlib.Path(__file__).resolve().parent.parent
REF = pathlib.Path("/root/reference/src/gnn_inference.cpp")
OUT = ROOT / "oracle" / "_ref"

pytestmark = pytest.mark.skipif(not REF.exists(), reason="reference sources not mounted here")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", str(ROOT / "oracle"), "ref"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return OUT


@pytest.fixture(scope="module")
def manifest(golden_dir):
    return json.loads((golden_dir / "manifest.json").read_text())


@pytest.fixture(scope="module")
def ref_layers(built):
    L = C.CDLL(str(built / "ref_layers.so"))
    L.ref_predict.argtypes = [C.c_char_p, C.c_int, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    L.ref_graph_layer.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def _ref_predict(L, text, keep, g, x):
    out = np.zeros(g.n * 35 + 8, dtype=np.float32)
    wd = C.c_uint32(0)
    rowptr = np.ascontiguousarray(g.rowptr, dtype=np.uint64)
    L.ref_predict(text.encode(), keep, C.c_float(g.ws), g.n, rowptr.ctypes.data, g.col.ctypes.data,
                  g.w.ctypes.data, x.ctypes.data, out.ctypes.data, C.byref(wd))
    return out[: g.n * wd.value].reshape(g.n, wd.value)


@pytest.mark.parametrize("maker", [
    lambda: gg.from_edge_list(3, [(0, 2), (1, 2)], [15, 15, 20]),
    lambda: gg.erdos_renyi(1000, 5000, 5),
    lambda: gg.rmat(10, 16, 10),
    lambda: gg.hub_graph(20000, 60000, 3, 4096, seed=7),
    lambda: gg.from_edge_list(130, [(0, i) for i in range(1, 40)], list(range(20, 150))),
])
def test_oracle_equals_reference_layer_code(ref_layers, oracle_model, model_text, maker):
    g = maker()
    oracle_model.set_weight_scale(g.ws)
    x = g.x()
    # every prefix that ends a fused stage, the logits, and the scores
    for keep, stop in ((7, 6), (14, 13), (20, 19), (-1, -1)):
        got = _ref_predict(ref_layers, model_text, keep, g, x)
        want = oracle_model.predict(g, x, stop_after=stop)
        assert got.shape == want.shape
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (keep, stop)


def test_reference_graph_layer_quirk(ref_layers):
    g = gg.erdos_renyi(300, 1500, 3)
    rng = np.random.default_rng(0)
    rowptr = np.ascontiguousarray(g.rowptr, dtype=np.uint64)
    for f in (1, 4, 16):
        h = rng.normal(size=(g.n, f)).astype(np.float32)
        out = np.zeros((g.n, 2 * f + 3), dtype=np.float32)
        wd = ref_layers.ref_graph_layer(C.c_float(77.0), g.n, f, rowptr.ctypes.data, g.col.ctypes.data,
                                        g.w.ctypes.data, h.ctypes.data, out.ctypes.data)
        assert wd == 2 * f + 3
        want = oracle_py.graph_layer(g, 77.0, h)
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("maker", [
    lambda: gg.erdos_renyi(3000, 6000, 5),                 # sparse: many degree-1/2 vertices, twins, dominations
    lambda: gg.erdos_renyi(2000, 12000, 6),
    lambda: gg.rmat(11, 4, 10),
    lambda: gg.from_edge_list(9, [(0, 1), (0, 2), (1, 2), (3, 4), (3, 5), (4, 5), (5, 6), (7, 8)],
                              [20, 30, 40, 25, 25, 60, 10, 5, 5]),
    lambda: gg.from_edge_list(6, [(0, 2), (0, 3), (1, 2), (1, 3), (4, 5)], [10, 10, 20, 20, 7, 7]),  # twins 0 and 1
    lambda: gg.erdos_renyi(1500, 2200, 8, lo=1, hi=9),     # small weights: the meta rules fire often
    lambda: gg.erdos_renyi(800, 3000, 9, lo=1, hi=40),
    # u larger than all of its own neighbours: neighborhood_difference's unfiltered tail copies u itself
    lambda: gg.from_edge_list(8, [(0, 7), (1, 7), (7, 6), (6, 2), (6, 3), (2, 3), (4, 5)], [3, 4, 9, 2, 6, 6, 12, 5]),
])
def test_reduction_predicates_match_reference_methods(ref_layers, maker):
    """oracle_reduction_flags against the reference's own is_twin / is_dominating / is_isolated."""
    g = maker()
    ref_layers.ref_reduction_flags.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                               C.c_void_p]
    want = np.zeros(g.n, dtype=np.uint8)
    rowptr = np.ascontiguousarray(g.rowptr, dtype=np.uint64)
    ref_layers.ref_reduction_flags(g.n, rowptr.ctypes.data, g.col.ctypes.data, g.w.ctypes.data, 20,
                                   want.ctypes.data)
    got = oracle_py.reduction_flags(g, 20)
    assert np.array_equal(got & 0x1F, want & 0x1F)
    # the two small-solver rules: the oracle's restatement against the reference's own rule functions, each
    # called on a fresh copy of the graph (they apply the reduction when they fire)
    ref_layers.ref_meta_flags.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    meta = np.zeros(g.n, dtype=np.uint8)
    ref_layers.ref_meta_flags(g.n, rowptr.ctypes.data, g.col.ctypes.data, g.w.ctypes.data, 20, meta.ctypes.data)
    assert np.array_equal(got & 0x60, meta), np.flatnonzero((got & 0x60) != meta)[:10]


@pytest.mark.parametrize("maker", [
    lambda: gg.erdos_renyi(3000, 6000, 5),
    lambda: gg.erdos_renyi(20000, 30000, 6),
    lambda: gg.erdos_renyi(30000, 90000, 7),
    lambda: gg.rmat(13, 4, 3),                             # hubs: the marks budget drops the flags
    lambda: gg.erdos_renyi(1500, 2200, 8, lo=1, hi=9),
    lambda: gg.from_edge_list(6, [(0, 2), (0, 3), (1, 2), (1, 3), (4, 5)], [10, 10, 20, 20, 7, 7]),
])
def test_flagged_reduce_is_the_reference_reduce(ref_layers, maker):
    """host/flagged_reduce.hpp (the consumer of the device flags: skips a first test of a vertex whose flags
    say no rule applies and whose 2-hop neighbourhood is untouched) against the reference's own reduce_graph on
    copies of the same graph: identical reduced graph, cover marks and offset (rc 0)."""
    g = maker()
    L = ref_layers
    L.ref_flagged_reduce_check.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    flags = oracle_py.reduction_flags(g, 20)
    rowptr = np.ascontiguousarray(g.rowptr, dtype=np.uint64)
    stats = np.zeros(4)
    rc = L.ref_flagged_reduce_check(g.n, rowptr.ctypes.data, g.col.ctypes.data, g.w.ctypes.data,
                                    flags.ctypes.data, stats.ctypes.data)
    assert rc == 0, (rc, stats)
    assert stats[3] > 0


def _run_cli(binary, graph_path, out_path):
    r = subprocess.run([str(binary), str(graph_path), str(out_path), "0", "-1", "0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return r.stdout.strip()


@pytest.mark.parametrize("binary", ["GNN_VC_reflayers", "GNN_VC_dropin"])
def test_cli_readme_graph(built, manifest, tmp_path, binary):
    (tmp_path / "ex3.graph").write_text("3 2 10\n15 3\n15 3\n20 1 2\n")
    out = _run_cli(built / binary, tmp_path / "ex3.graph", tmp_path / "ex3.out")
    spec = manifest["ex3"]["cli"]
    assert out.startswith(spec["stdout_prefix"])
    assert out.split(",")[6] == str(spec["final_cost"])
    assert [int(v) for v in (tmp_path / "ex3.out").read_text().split()] == spec["cover"]


@pytest.mark.slow
@pytest.mark.parametrize("binary", ["GNN_VC_reflayers", "GNN_VC_dropin"])
def test_cli_er100k_identical_cover(built, manifest, golden_dir, tmp_path, binary):
    spec = manifest["er100k"]
    p = spec["graph"]
    g = gg.erdos_renyi(p["n"], p["m"], p["seed"])
    text = gg.metis_text(g)
    assert hashlib.md5(text.encode()).hexdigest() == spec["metis_md5"]
    (tmp_path / "er100k.graph").write_text(text)
    out = _run_cli(built / binary, tmp_path / "er100k.graph", tmp_path / "er100k.out")
    fields = out.split(",")
    assert fields[0] == "er100k" and int(fields[1]) == spec["cli"]["final_cost"]
    raw = (tmp_path / "er100k.out").read_bytes()
    assert hashlib.md5(raw).hexdigest() == spec["cli"]["result_md5"]
    cover = np.array(raw.split(), dtype=np.uint8)
    gold = np.unpackbits(np.fromfile(golden_dir / spec["cli"]["cover_bits_file"], dtype=np.uint8))[: p["n"]]
    assert np.array_equal(cover, gold)
    # identical final vertex-cover weight on this integer-weighted graph
    assert int(g.w[cover == 1].astype(np.int64).sum()) == spec["cli"]["final_cost"]


def test_fast_io_driver_is_the_same_program(built, tmp_path):
    """oracle/ref_driver_fastio.cpp: the reference's driver with this repo's METIS reader behind parse_graph and
    unflushed result writes, against the plain drop-in: same stdout cost fields, byte-identical
    result files."""
    (tmp_path / "ex3.graph").write_text("3 2 10\n15 3\n15 3\n20 1 2\n")
    g = gg.erdos_renyi(30000, 150000, 13)
    (tmp_path / "g.graph").write_text(gg.metis_text(g))
    for name in ("ex3", "g"):
        res = []
        for binary in ("GNN_VC_dropin", "GNN_VC_dropin_fastio"):
            out = tmp_path / f"{name}.{binary}.out"
            r = subprocess.run([str(built / binary), str(tmp_path / f"{name}.graph"), str(out), "0", "-1", "0"],
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
            f = r.stdout.strip().split(",")
            res.append((f[0], f[1], f[2], out.read_bytes()))     # name, cost, best seen; (the last field is a time)
        assert res[0] == res[1], name
    assert len(res[0][3]) == 2 * g.n
