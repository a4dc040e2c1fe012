"""GPU parity: the HIP path (through the C ABI) against the oracle.

Bar (north_star + SURVEY.md fact 4): logits — everything up to the final
sigmoid — bit-identical to the reference arithmetic; scores within 1 ulp of the
host-libm sigmoid (the device exp is a restatement, not the host's libm), with
the count of differing scores reported.
"""
import json

import numpy as np
import pytest

from oracle import oracle_py
from tools import graphgen as gg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(model_text):
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    assert e.fused and e.num_stages == 3 and e.num_layers == 21
    yield e
    e.close()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def ulp(a, b):
    return np.abs(bits(a).view(np.int32).astype(np.int64) - bits(b).view(np.int32).astype(np.int64))


def check_forward(engine, oracle_model, g, label=""):
    engine.set_weight_scale(g.ws)
    oracle_model.set_weight_scale(g.ws)
    engine.upload_graph(g)
    scores, logits = engine.forward(g.x())
    if g.n == 0:
        assert scores.shape == (0, 1)
        return
    want_logits = oracle_model.logits(g)
    want_scores = oracle_model.scores(g)
    mism = int((bits(logits[:, 0]) != bits(want_logits)).sum())
    assert mism == 0, f"{label}: {mism}/{g.n} logits differ from the oracle, max ulp " \
                      f"{ulp(logits[:, 0], want_logits).max()}"
    d = ulp(scores[:, 0], want_scores)
    assert d.max() <= 1, f"{label}: device sigmoid off by {d.max()} ulp"
    # the exact-parity route: host sigmoid on device logits == reference scores
    assert np.array_equal(bits(oracle_py.sigmoid(logits[:, 0])), bits(want_scores))
    return int((d > 0).sum())


def test_readme_graph(engine, oracle_model, golden_dir):
    g = gg.from_edge_list(3, [(0, 2), (1, 2)], [15, 15, 20])
    check_forward(engine, oracle_model, g, "ex3")
    _, logits = engine.forward(g.x())
    gold = np.fromfile(golden_dir / "ex3.scores.f32", dtype=np.float32)
    assert ulp(oracle_py.sigmoid(logits[:, 0]), gold).max() <= 1


def test_empty_graph(engine, oracle_model):
    check_forward(engine, oracle_model, gg.from_edge_list(0, [], []), "empty")


@pytest.mark.parametrize("n,m,seed", [(1, 0, 0), (2, 1, 1), (63, 200, 2), (64, 300, 3), (65, 300, 4),
                                      (1000, 5000, 5), (4097, 30000, 6)])
def test_small_random_graphs(engine, oracle_model, n, m, seed):
    if m == 0:
        g = gg.from_edge_list(n, [], [57] * n)
    else:
        g = gg.erdos_renyi(n, min(m, n * (n - 1) // 2), seed)
    check_forward(engine, oracle_model, g, f"er{n}")


def test_isolated_and_ragged(engine, oracle_model):
    # isolated vertices, a star, a path: ragged degrees inside one 64-vertex tile
    edges = [(0, i) for i in range(1, 40)] + [(50 + i, 51 + i) for i in range(20)]
    g = gg.from_edge_list(130, edges, list(range(20, 150)))
    check_forward(engine, oracle_model, g, "ragged")


def test_rmat_scale10(engine, oracle_model):
    g = gg.rmat(10, 16, 10)
    check_forward(engine, oracle_model, g, "rmat10")


def test_hub_rows(engine, oracle_model):
    # CSR-order sums over thousands of neighbours: order-sensitive in fp32
    g = gg.hub_graph(20000, 60000, 3, 4096, seed=7)
    check_forward(engine, oracle_model, g, "hub4096")


def test_er100k_matches_reference_golden(engine, oracle_model, golden_dir):
    spec = json.loads((golden_dir / "manifest.json").read_text())["er100k"]
    p = spec["graph"]
    g = gg.erdos_renyi(p["n"], p["m"], p["seed"])
    check_forward(engine, oracle_model, g, "er100k")
    _, logits = engine.forward(g.x())
    gold = np.fromfile(golden_dir / spec["scores_file"], dtype=np.float32)
    host_scores = oracle_py.sigmoid(logits[:, 0])
    d = ulp(host_scores, gold)
    assert d.max() <= 1 and int((d > 0).sum()) <= 8  # host-expf caveat, tests/golden/README.md


def test_repeatable_and_graph_replacement(engine, oracle_model):
    """predict is called repeatedly with a shrinking graph (reference src/GNN_VC.cpp:171-192)."""
    g1 = gg.erdos_renyi(5000, 30000, 11)
    g2 = gg.erdos_renyi(700, 2000, 12)
    check_forward(engine, oracle_model, g1, "g1")
    a, _ = engine.forward(g1.x())
    b, _ = engine.forward(g1.x())
    assert np.array_equal(bits(a), bits(b))
    check_forward(engine, oracle_model, g2, "g2")
    check_forward(engine, oracle_model, g1, "g1 again")


def test_weight_scale_is_honoured(engine, oracle_model):
    g = gg.erdos_renyi(300, 1200, 13)
    engine.upload_graph(g)
    for ws in (120.0, 200.0, 33.0):
        engine.set_weight_scale(ws)
        oracle_model.set_weight_scale(ws)
        x = (g.w.astype(np.float32) / np.float32(ws)).astype(np.float32)
        _, logits = engine.forward(x)
        assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g, x)))


def test_arbitrary_input_features(engine, oracle_model):
    """in is an argument of predict, not necessarily W/ws."""
    g = gg.erdos_renyi(2000, 9000, 14)
    engine.set_weight_scale(g.ws)
    oracle_model.set_weight_scale(g.ws)
    engine.upload_graph(g)
    rng = np.random.default_rng(0)
    x = rng.normal(size=g.n).astype(np.float32)
    _, logits = engine.forward(x)
    assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g, x)))


# ---------------------------------------------------------------- stages / partitions

def test_stage_api_row_ranges(engine, oracle_model):
    """Stage-by-stage over vertex ranges (the 1-D partition unit) == whole forward."""
    import torch
    g = gg.erdos_renyi(3000, 15000, 15)
    engine.set_weight_scale(g.ws)
    oracle_model.set_weight_scale(g.ws)
    engine.upload_graph(g)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(g.x()).to(dev)
    h1 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
    h2 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
    sc = torch.zeros(g.n, dtype=torch.float32, device=dev)
    lg = torch.zeros(g.n, dtype=torch.float32, device=dev)
    cuts = [0, 1, 64, 1000, 1777, g.n]
    side = torch.cuda.Stream(device=dev)   # a caller-owned stream (NULL would mean the engine's own)
    side.wait_stream(torch.cuda.current_stream())
    engine.set_stream(side.cuda_stream)
    try:
        for st, (src, dst) in enumerate(((x, h1), (h1, h2), (h2, sc))):
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                engine.stage_forward_device(st, lo, hi, src.data_ptr(), dst.data_ptr(),
                                            lg.data_ptr() if st == 2 else 0)
        side.synchronize()
    finally:
        engine.set_stream(None)
    want_h1 = oracle_model.predict(g, g.x(), stop_after=6)
    want_h2 = oracle_model.predict(g, g.x(), stop_after=13)
    assert np.array_equal(bits(h1[:-1].cpu().numpy()), bits(want_h1))
    assert np.array_equal(bits(h2[:-1].cpu().numpy()), bits(want_h2))
    assert np.array_equal(bits(lg.cpu().numpy()), bits(oracle_model.logits(g)))
    assert float(h1[-1].abs().sum()) == 0.0  # pad row untouched


# ---------------------------------------------------------------- layer-level entry points

def test_graph_layer_entry_point(engine):
    g = gg.erdos_renyi(777, 4000, 16)
    engine.set_weight_scale(77.0)
    engine.upload_graph(g)
    rng = np.random.default_rng(1)
    for f in (1, 4, 16):
        h = rng.normal(size=(g.n, f)).astype(np.float32)
        assert np.array_equal(bits(engine.graph_layer(h)), bits(oracle_py.graph_layer(g, 77.0, h)))


def test_linear_relu_sigmoid_entry_points(engine, oracle_model):
    rng = np.random.default_rng(2)
    for W, b in oracle_model.linear_params():
        h = rng.uniform(-3, 3, size=(1237, W.shape[0])).astype(np.float32)
        h[rng.random(h.shape) < 0.2] = 0.0
        assert np.array_equal(bits(engine.linear(h, W, b)), bits(oracle_py.linear_layer(h, W, b)))
    h = rng.normal(size=5000).astype(np.float32)
    h[::7] = -0.0
    assert np.array_equal(bits(engine.relu(h)), bits(oracle_py.relu(h)))
    assert ulp(engine.sigmoid(h), oracle_py.sigmoid(h)).max() <= 1


def test_sgemm_seam(engine):
    """dot() with transposes and beta (reference src/matrix.cpp:106-122)."""
    rng = np.random.default_rng(3)
    A = rng.normal(size=(37, 19)).astype(np.float32)
    B = rng.normal(size=(19, 23)).astype(np.float32)
    zero = np.zeros(23, dtype=np.float32)
    want = oracle_py.linear_layer(A, B, zero)
    assert np.array_equal(bits(engine.sgemm(A, B)), bits(want))
    assert np.array_equal(bits(engine.sgemm(np.ascontiguousarray(A.T), B, trans_a=True)), bits(want))
    assert np.array_equal(bits(engine.sgemm(A, np.ascontiguousarray(B.T), trans_b=True)), bits(want))
    # beta != 0 (training only in the reference; the inference path passes 0): the documented contract is
    # fma(beta, C_old, chain) — oracle_sgemm restates it; bitwise, with and without transposes, odd betas
    C0 = rng.normal(size=(37, 23)).astype(np.float32)
    for beta in (0.5, 1.0, -1.7, 0.3):
        assert np.array_equal(bits(engine.sgemm(A, B, C_in=C0, beta=beta)), bits(oracle_py.sgemm(A, B, C0, beta)))
        assert np.array_equal(bits(engine.sgemm(np.ascontiguousarray(A.T), np.ascontiguousarray(B.T), C_in=C0, beta=beta,
                                                trans_a=True, trans_b=True)),
                              bits(oracle_py.sgemm(np.ascontiguousarray(A.T), np.ascontiguousarray(B.T), C0, beta, True, True)))
    assert np.array_equal(bits(oracle_py.sgemm(A, B)), bits(want))


@pytest.mark.parametrize("name", ["ex3", "er4k", "hub2k"])
def test_layer_boundary_fixtures_on_the_gpu(engine, golden_dir, name):
    """The stage boundaries of the fused forward (h1, h2, logits) against tests/golden/manifest_layers.json:
    outputs of the reference's own layer code with its products in genuine OpenBLAS."""
    import torch
    man = json.loads((golden_dir / "manifest_layers.json").read_text())
    spec = man["graphs"][name]
    p = spec["graph"]
    g = (gg.from_edge_list(p["n"], p["edges"], p["weights"]) if p["kind"] == "edge_list" else
         gg.erdos_renyi(p["n"], p["m"], p["seed"]) if p["kind"] == "erdos_renyi" else
         gg.hub_graph(p["n"], p["m"], p["hubs"], p["hub_degree"], seed=p["seed"]))
    assert gg.metis_md5(g) == spec["metis_md5"]
    engine.set_weight_scale(g.ws)
    engine.upload_graph(g)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(g.x()).to(dev)
    h1 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
    h2 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
    sc = torch.zeros(g.n, dtype=torch.float32, device=dev)
    lg = torch.zeros(g.n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    engine.stage_forward_device(0, 0, g.n, x.data_ptr(), h1.data_ptr())
    engine.stage_forward_device(1, 0, g.n, h1.data_ptr(), h2.data_ptr())
    engine.stage_forward_device(2, 0, g.n, h2.data_ptr(), sc.data_ptr(), lg.data_ptr())
    engine.synchronize()
    gold = {k: np.fromfile(golden_dir / f["file"], dtype=np.float32).reshape(f["shape"]) for k, f in spec["files"].items()}
    assert np.array_equal(bits(h1[:-1].cpu().numpy()), bits(gold["h1"]))
    assert np.array_equal(bits(h2[:-1].cpu().numpy()), bits(gold["h2"]))
    assert np.array_equal(bits(lg.cpu().numpy()), bits(gold["logits"][:, 0]))
    assert ulp(sc.cpu().numpy(), gold["scores2"][:, 0]).max() <= 1


def test_unfused_model_path(model_text, oracle_model):
    """A model that does not match the fused plan runs layer by layer — still on the GPU."""
    import gnn_mwvc_amd as G
    # drop the last stage: 14 layers ending in ReLU (not a fused plan: no sigmoid tail)
    head = model_text.split("Graph_Layer")
    text = "Trunc\n14 Layers\nGraph_Layer" + head[1] + "Graph_Layer" + head[2]
    e = G.Engine(text, device=0)
    try:
        assert not e.fused and e.num_layers == 14 and e.out_width == 16
        g = gg.erdos_renyi(500, 2500, 17)
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        e.upload_graph(g)
        out, _ = e.forward(g.x(), want_logits=False)
        want = oracle_model.predict(g, g.x(), stop_after=13)
        assert np.array_equal(bits(out), bits(want))
    finally:
        e.close()


def test_malformed_graphs_are_rejected(model_text):
    """A column id >= n or broken row pointers would fault the GPU: both entry points refuse them."""
    import torch
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    try:
        g = gg.erdos_renyi(500, 2000, 19)
        e.upload_graph(g)
        bad = gg.CsrGraph(g.n, g.rowptr, g.col.copy(), g.w, g.nw)
        bad.col[7] = g.n + 5
        with pytest.raises(G.GnnvcError) as ei:
            e.upload_graph(bad)
        assert ei.value.code == -1
        rp = g.rowptr.copy()
        rp[10], rp[11] = rp[11], rp[10] + 0
        rp[10] = rp[12] + 1
        with pytest.raises(G.GnnvcError):
            e.upload_graph(gg.CsrGraph(g.n, rp, g.col, g.w, g.nw))
        with pytest.raises(G.GnnvcError):           # no usable graph is left behind
            e.forward(g.x())
        dev = torch.device("cuda:0")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
        colpad = np.zeros(g.nnz + 64, dtype=np.uint32)
        colpad[: g.nnz] = bad.col
        d = [t(g.rowptr.astype(np.uint32)), t(colpad), t(g.w), t(g.nw)]
        with pytest.raises(G.GnnvcError):
            e.attach_graph_device(g.n, g.nnz, *[x.data_ptr() for x in d], keepalive=d)
        e.upload_graph(g)                           # and a good graph still works afterwards
        e.forward(g.x())
    finally:
        e.close()


def test_one_round_trip_hand_off_classes_like_four(model_text, oracle_model):
    """Round 4: the hand-off's checks and what it wants to know about the graph (XCD cuts, tile waste, long rows) are queued together
    and waited for once (classify_hand_off).  The classing passes read row pointers only, so a graph that fails the checks — with
    hub rows, at a size where every pass runs — is refused without harm on both routes, and good graphs of every family are classed
    as before: a hub graph keeps its long and giant rows, a skewed graph its sorted tiles, and the logits are the oracle's."""
    import torch
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    try:
        g = gg.hub_graph(12000, 60000, 3, 5000, seed=31)
        oracle_model.set_weight_scale(g.ws)
        want = oracle_model.logits(g)
        e.set_weight_scale(g.ws)
        e.upload_graph(g)
        assert e.get_info("long_rows") >= 3
        _, lg = e.forward(g.x())
        assert np.array_equal(bits(lg[:, 0]), bits(want))
        # broken row pointers in the middle of the hub rows' range, then a column id beyond n: refused on the host route ...
        rp = g.rowptr.copy()
        rp[100] = rp[-1] + 7
        for bad in (gg.CsrGraph(g.n, rp, g.col, g.w, g.nw), None):
            if bad is None:
                bad = gg.CsrGraph(g.n, g.rowptr, g.col.copy(), g.w, g.nw)
                bad.col[g.nnz // 2] = g.n + 1
            with pytest.raises(G.GnnvcError):
                e.upload_graph(bad)
            with pytest.raises(G.GnnvcError):           # no usable graph is left behind
                e.forward(g.x())
            # ... and on device arrays
            dev = torch.device("cuda:0")
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
            colpad = np.zeros(bad.nnz + 64, dtype=np.uint32)
            colpad[: bad.nnz] = bad.col
            d = [t(bad.rowptr.astype(np.uint32)), t(colpad), t(bad.w), t(bad.nw)]
            with pytest.raises(G.GnnvcError):
                e.attach_graph_device(bad.n, bad.nnz, *[x.data_ptr() for x in d], keepalive=d)
            e.upload_graph(g)                            # a good graph afterwards: classed and scored as before
            assert e.get_info("long_rows") >= 3
            _, lg = e.forward(g.x())
            assert np.array_equal(bits(lg[:, 0]), bits(want))
        for maker, sorted_tiles in ((lambda: gg.rmat(14, 16, 5), True), (lambda: gg.erdos_renyi(30000, 300000, 6), False)):
            g2 = maker()
            oracle_model.set_weight_scale(g2.ws)
            e.set_weight_scale(g2.ws)
            e.set_option("sorted_min_nnz", 0)
            e.upload_graph(g2)
            if sorted_tiles:
                assert e.get_info("sorted_tiles_active") == 1 or e.get_info("long_rows") > 0    # (the skew is seen one way or the other)
            else:
                assert e.get_info("sorted_tiles_active") == 0 and e.get_info("long_rows") == 0
            for rep in range(2):
                _, lg = e.forward(g2.x())
                assert np.array_equal(bits(lg[:, 0]), bits(oracle_model.logits(g2))), rep
    finally:
        e.close()


@pytest.mark.parametrize("route", ["upload", "staged"])
def test_malformed_row_pointers_never_reach_the_early_builders(model_text, oracle_model, route):
    """a large host hand-off classes the graph and starts the flat plan builders from the row pointers BEFORE the
    full validation at its end; those builders read col[] and write the entry arrays at offsets taken from the row pointers.
    With the early path forced onto a small graph, non-monotone / out-of-range inner row pointers must be refused before any
    builder runs, leave no graph behind, and leave the engine usable (an attach after an abandoned staged hand-off included)."""
    import gnn_mwvc_amd as G
    import torch
    g = gg.erdos_renyi(40000, 400000, 77)
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_option("plans_at_handoff", 2)
        e.set_option("handoff_min_entries", 1)
        e.set_weight_scale(g.ws)
        for kind in ("descending", "beyond_nnz", "huge"):
            rp = g.rowptr.copy()
            if kind == "descending":
                rp[20000] = rp[20002] + 5
            elif kind == "beyond_nnz":
                rp[30000:39000] = g.nnz + 12345          # monotone among themselves, but past the end (and back down after)
            else:
                rp[1000] = (1 << 33) + 7                 # does not fit 32 bits: must not pass as its low half
            bad = gg.CsrGraph(g.n, rp, g.col, g.w, g.nw)
            with pytest.raises(G.GnnvcError) as ei:
                if route == "upload":
                    e.upload_graph(bad)
                elif kind == "huge":
                    raise G.GnnvcError(-1, "(the staged arrays are 32-bit: not expressible)")
                else:
                    e.upload_graph_staged(bad, pieces=3)
            assert ei.value.code == -1, kind
            e.n = g.n                                   # (the wrapper's own shape check: let the call reach the engine)
            with pytest.raises(G.GnnvcError):           # no usable graph is left behind
                e.forward(g.x())
        # an attach right after the refused / abandoned hand-off gets plans of its OWN geometry (attach_common resets every
        # open build)
        from tools import graphgen_torch as ggt
        g2 = gg.erdos_renyi(30000, 330000, 78)
        dg = ggt.from_host(g2, torch.device("cuda:0"))
        torch.cuda.synchronize()
        e.set_weight_scale(g2.ws)
        oracle_model.set_weight_scale(g2.ws)
        e.attach_graph_device(dg.n, dg.nnz, dg.rowptr.data_ptr(), dg.col.data_ptr(), dg.w.data_ptr(), dg.nw.data_ptr(), keepalive=dg)
        _, lg = e.forward(g2.x())
        assert np.array_equal(bits(lg[:, 0]), bits(oracle_model.logits(g2)))
        e.set_weight_scale(g.ws)
        if route == "upload":
            e.upload_graph(g)
        else:
            e.upload_graph_staged(g, pieces=3)
        assert e.get_info("lds_table_active") == 1 and e.get_info("compact_gather_active") == 1
        _, lg = e.forward(g.x())
        assert np.array_equal(bits(lg[:, 0]), bits(want))
    finally:
        e.close()


@pytest.mark.parametrize("maker,pieces", [
    (lambda: gg.erdos_renyi(5000, 40000, 23), 1),
    (lambda: gg.erdos_renyi(5000, 40000, 23), 7),
    (lambda: gg.hub_graph(20000, 60000, 3, 4096, seed=7), 3),
    (lambda: gg.from_edge_list(5, [], [20, 30, 40, 50, 60]), 2),        # no edges at all
    (lambda: gg.from_edge_list(0, [], []), 1),                          # the empty graph
])
def test_staged_hand_off_equals_upload(model_text, oracle_model, maker, pieces):
    """gnnvc_graph_staging .. gnnvc_commit_staged_graph (f-1) leaves the engine in the same state
    as gnnvc_upload_graph: same logits, bit for bit, and the staging is reusable for the next graph."""
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    try:
        for g in (maker(), gg.erdos_renyi(300, 900, 29), maker()):      # grow / shrink / grow again
            e.set_weight_scale(g.ws)
            oracle_model.set_weight_scale(g.ws)
            e.upload_graph_staged(g, pieces=pieces)
            scores, logits = e.forward(g.x())
            if g.n == 0:
                assert scores.shape == (0, 1)
                continue
            assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g)))
            e.upload_graph(g)
            _, again = e.forward(g.x())
            assert np.array_equal(bits(again), bits(logits))
    finally:
        e.close()


def test_staged_hand_off_errors(model_text):
    import ctypes as C
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    L, h = e._L, e._h
    try:
        assert L.gnnvc_commit_staged_graph(h) == -4                     # nothing staged
        assert L.gnnvc_staged_columns_ready(h, 0, 1) == -4
        g = gg.erdos_renyi(500, 2000, 19)
        ptr = [C.c_void_p() for _ in range(4)]
        assert L.gnnvc_graph_staging(h, g.n, g.nnz, *[C.byref(p) for p in ptr]) == 0
        assert all(p.value for p in ptr)
        assert L.gnnvc_staged_columns_ready(h, 64, 64) == -1            # pieces out of order
        assert L.gnnvc_staged_columns_ready(h, 0, g.nnz + 1) == -1      # past the end
        u32 = lambda p, k: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(k,))
        u32(ptr[0], g.n + 1)[:] = g.rowptr.astype(np.uint32)
        u32(ptr[1], g.nnz)[:] = g.col
        u32(ptr[1], g.nnz)[7] = g.n + 5                                 # a wild column id
        u32(ptr[2], g.n)[:] = g.w
        u32(ptr[3], g.n)[:] = g.nw
        assert L.gnnvc_staged_columns_ready(h, 0, 64) == 0
        assert L.gnnvc_commit_staged_graph(h) == -1
        assert b"column id" in L.gnnvc_last_error(h)
        with pytest.raises(G.GnnvcError):                               # no usable graph is left behind
            e.n = g.n
            e.forward(g.x())
        assert L.gnnvc_graph_staging(h, g.n, g.nnz, None, None, None, None) == 0
        u32(ptr[0], g.n + 1)[g.n] = g.nnz - 1                           # rowptr[n] != nnz
        assert L.gnnvc_commit_staged_graph(h) == -1
        e.upload_graph_staged(g)                                        # and a good graph still works
        e.forward(g.x())
    finally:
        e.close()


def test_forward_timing_option(model_text, oracle_model):
    """Round 4: a forward records HIP events only when option "forward_timing" asks for them (four records were 5.5 us of a 30 us
    forward): 0, the default — gnnvc_last_forward_ms is a state error; 1 — the total; 2 — the stages too.  Same logits either way;
    the verdict words of the per-graph plans still reach the host (one small kernel instead of copies)."""
    import gnn_mwvc_amd as G
    g = gg.erdos_renyi(60000, 600000, 5)
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)
    e = G.Engine(model_text, device=0)
    try:
        e.set_weight_scale(g.ws)
        e.upload_graph(g)
        _, lg = e.forward(g.x())
        assert np.array_equal(bits(lg[:, 0]), bits(want))
        with pytest.raises(G.GnnvcError):
            e.last_forward_ms()
        e.set_option("forward_timing", 1)
        _, lg = e.forward(g.x())
        assert np.array_equal(bits(lg[:, 0]), bits(want))
        total, stages = e.last_forward_ms()
        assert 0.0 < total < 50.0 and all(s == -1.0 for s in stages)
        e.set_option("forward_timing", 2)
        for rep in range(12):    # (... and on through the table tiles' steady state, where only every eighth forward asks for verdicts)
            _, lg = e.forward(g.x())
            assert np.array_equal(bits(lg[:, 0]), bits(want)), rep
        total, stages = e.last_forward_ms()
        assert len(stages) == 3 and all(s > 0.0 for s in stages) and abs(sum(stages) - total) < 0.05 * total + 0.01
        assert e.get_info("table_tiles_fit_stage1") == 1 and e.get_info("table_tiles_fit_stage2") == 1
        e.set_option("forward_timing", 0)
        _, lg = e.forward(g.x())
        assert np.array_equal(bits(lg[:, 0]), bits(want))
        with pytest.raises(G.GnnvcError):
            e.last_forward_ms()
    finally:
        e.close()


def test_errors(engine):
    import gnn_mwvc_amd as G
    g = gg.erdos_renyi(100, 300, 18)
    engine.upload_graph(g)
    with pytest.raises(G.GnnvcError):
        engine.stage_forward_device(0, 0, g.n + 1, 1, 1)   # row range outside the graph
    with pytest.raises(G.GnnvcError):
        engine.stage_forward_device(7, 0, g.n, 1, 1)       # no such stage


# ---------------------------------------------------------------- host C++ mirror

def test_host_mirror_predict(oracle_model, model_text, tmp_path):
    """gnn::model::predict of the C++ host mirror (same signature as the reference's)
    through libgnnvc_hip.so: scores bit-identical to the oracle's host-libm scores."""
    import pathlib
    import subprocess
    pkg = pathlib.Path(__file__).resolve().parent.parent / "gnn-mwvc_amd"
    tool = pkg / "gnnvc_predict"
    if not tool.exists():
        r = subprocess.run(["make", "-C", str(pkg / "host")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    g = gg.erdos_renyi(5000, 30000, 21)
    (tmp_path / "g.metis").write_text(gg.metis_text(g))
    (tmp_path / "m.txt").write_text(model_text)
    r = subprocess.run([str(tool), str(tmp_path / "m.txt"), str(tmp_path / "g.metis"),
                        str(tmp_path / "s.f32")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(tmp_path / "s.f32", dtype=np.float32)
    oracle_model.set_weight_scale(g.ws)
    assert np.array_equal(bits(got), bits(oracle_model.scores(g)))


# ---------------------------------------------------------------- column-blocked F = 1 stage

@pytest.mark.parametrize("block_cols", [64, 1000, 4096, 1 << 19])
def test_blocked_stage0_is_bit_identical(model_text, oracle_model, block_cols):
    """The column-blocked plan re-buckets the CSR per column block; every row's adds must
    still happen in stored order, so logits stay bit-identical (forced on small graphs)."""
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_option("block_cols", block_cols)
        e.set_option("blocked_stage0", 2)      # also on the skewed sample graphs (off there by default)
        e.set_option("lds_table", 0)           # (the LDS-table plan would take precedence; it has its own test)
        graphs = [gg.erdos_renyi(5000, 40000, 31), gg.hub_graph(20000, 60000, 3, 4096, seed=7),
                  gg.rmat(11, 8, 3), gg.from_edge_list(130, [(0, i) for i in range(1, 40)], list(range(20, 150)))]
        for g in graphs:
            e.set_weight_scale(g.ws)
            oracle_model.set_weight_scale(g.ws)
            e.upload_graph(g)
            want_active = 1 if (g.n + block_cols - 1) // block_cols >= 2 else 0
            # the index is built lazily, on the second forward over the same graph
            _, first = e.forward(g.x())
            assert e.get_info("blocked_stage0_active") == 0
            _, logits = e.forward(g.x())
            assert e.get_info("blocked_stage0_active") == want_active
            assert np.array_equal(bits(first), bits(logits))
            assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g)))
            # stage output alone (h1) and a row sub-range through the stage entry point
            import torch
            dev = torch.device("cuda:0")
            x = torch.from_numpy(g.x()).to(dev)
            h1 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            mid = (g.n // 2) // 64 * 64
            for lo, hi in ((0, mid), (mid, g.n)):
                e.stage_forward_device(0, lo, hi, x.data_ptr(), h1.data_ptr())
            e.synchronize()
            assert np.array_equal(bits(h1[:-1].cpu().numpy()), bits(oracle_model.predict(g, g.x(), stop_after=6)))
    finally:
        e.close()


def test_unsorted_adjacency_falls_back_to_stored_order(model_text, oracle_model):
    """Stored order is the contract.  With neighbour lists in descending order the blocked
    plan would change the add order, so the engine must not use it."""
    import gnn_mwvc_amd as G
    g = gg.erdos_renyi(4000, 30000, 33)
    rp = g.rowptr.astype(np.int64)
    col = g.col.copy()
    for u in range(g.n):
        col[rp[u]:rp[u + 1]] = col[rp[u]:rp[u + 1]][::-1]
    g2 = gg.CsrGraph(g.n, g.rowptr, col, g.w, g.nw)
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_option("block_cols", 256)
        e.set_weight_scale(g2.ws)
        oracle_model.set_weight_scale(g2.ws)
        e.upload_graph(g2)
        e.forward(g2.x())
        _, logits = e.forward(g2.x())      # second forward: the blocked plan is considered and rejected
        assert e.get_info("blocked_stage0_active") == 0
        want = oracle_model.logits(g2)
        assert np.array_equal(bits(logits[:, 0]), bits(want))
        # and the order does matter: the sorted graph gives (slightly) different logits
        oracle_model.set_weight_scale(g.ws)
        assert not np.array_equal(bits(want), bits(oracle_model.logits(g)))
    finally:
        e.close()


# ---------------------------------------------------------------- LDS-table plan of the F = 1 stage

def _dense_graph(n, deg, seed):
    """Every row has ~deg neighbours inside one or two column blocks: long runs per (row, block)."""
    return gg.erdos_renyi(n, n * deg // 2, seed)


@pytest.mark.parametrize("maker,force", [
    (lambda: gg.erdos_renyi(20000, 200000, 61), 1),
    (lambda: gg.erdos_renyi(200000, 900000, 62), 1),                   # three column blocks of 81920
    (lambda: _dense_graph(3000, 300, 63), 1),                          # runs of ~300 entries: longer than a lane, a wave, a 256-entry step
    (lambda: gg.hub_graph(20000, 60000, 3, 4096, seed=7), 2),          # forced on a skewed graph: 4096-entry runs
    (lambda: gg.from_edge_list(130, [(0, i) for i in range(1, 40)], list(range(20, 150))), 1),
    (lambda: gg.from_edge_list(5, [], [20, 30, 40, 50, 60]), 1),       # no entries at all: the plan steps aside
])
def test_lds_table_plan_is_bit_identical(model_text, oracle_model, maker, force):
    """k_lt_*: neighbour values come from byte slices of the weight table held in LDS, entries regrouped by
    column block — every row is still summed in stored order, so the logits do not move by a bit."""
    import torch
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_option("lds_table", force)
        e.set_option("blocked_stage0", 0)
        if force == 2:
            e.set_option("long_row_threshold", 0)      # the plan does not combine with the long-row kernels
        g = maker()
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        e.upload_graph(g)
        _, first = e.forward(g.x())
        assert e.get_info("lds_table_active") == 0     # built lazily, on the second forward over the same graph
        _, logits = e.forward(g.x())
        assert e.get_info("lds_table_active") == (1 if g.nnz else 0)
        assert np.array_equal(bits(first), bits(logits))
        assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g)))
        if not g.nnz:
            return
        assert e.get_info("lds_table_steps") >= 4 * e.get_info("lds_table_chunks")
        # row sub-ranges through the stage entry point: a long one keeps the plan (the chunk that straddles the cut
        # is computed whole), a short one would leave most CUs idle and takes the blocked / plain kernel instead
        dev = torch.device("cuda:0")
        x = torch.from_numpy(g.x()).to(dev)
        h1 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        mid = max(64, (g.n // 8) // 64 * 64) if g.n > 128 else 64
        for lo, hi in ((mid, g.n), (0, mid)):
            e.stage_forward_device(0, lo, hi, x.data_ptr(), h1.data_ptr())
        e.synchronize()
        want_h1 = oracle_model.predict(g, g.x(), stop_after=6)
        assert np.array_equal(bits(h1[:-1].cpu().numpy()), bits(want_h1))
        # an input that is NOT W / ws: the device-side check sends every row down the plain gather
        rng = np.random.default_rng(5)
        x2 = rng.uniform(0.1, 1.0, size=(g.n, 1)).astype(np.float32)
        _, lg2 = e.forward(x2)
        assert e.get_info("lds_table_active") == 1
        assert np.array_equal(bits(lg2[:, 0]), bits(oracle_model.predict(g, x2, stop_after=oracle_model.n_layers - 2)[:, 0]))
        x3 = g.x().copy()
        x3[g.n // 2] = np.nextafter(x3[g.n // 2], np.float32(2))           # one ulp off in one place is enough
        _, lg3 = e.forward(x3)
        assert np.array_equal(bits(lg3[:, 0]), bits(oracle_model.predict(g, x3, stop_after=oracle_model.n_layers - 2)[:, 0]))
        _, back = e.forward(g.x())                                        # and the plan is used again afterwards
        assert np.array_equal(bits(back), bits(logits))
    finally:
        e.close()


@pytest.mark.parametrize("scale,want_bits", [(1, 8), (3, 10), (8, 10), (9, 16), (500, 16), (700, 0)])
def test_lds_table_entry_width_follows_the_weights(model_text, oracle_model, scale, want_bits):
    """Round 4: the LDS-table plan's entries are 8, 10 or 16 bits wide by the graph's largest weight — weights
    beyond a byte (WEIGHT_SCALE is any u32; folds create new weights) keep a table instead of dropping to the column-blocked
    plan; beyond 65 535 the plan steps aside.  Several column blocks of every width, the byte / 10-bit / 16-bit packing of
    odd vertex counts, one fold-like weight above the rest, an input that is no k / ws, forced widths: the oracle's bits."""
    import gnn_mwvc_amd as G
    g0 = gg.erdos_renyi(200003, 2400000, 66)            # 5 / 4 / 3 column blocks at 16 / 10 / 8 bits
    w = (g0.w.astype(np.int64) * scale).astype(np.uint32)
    g = gg.CsrGraph(g0.n, g0.rowptr, g0.col, w, gg.neighbourhood_weights(g0.rowptr, g0.col, w))
    assert int(w.max()) == 120 * scale
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_option("blocked_stage0", 0)
        e.set_option("handoff_min_entries", 1)           # the plan at hand-off: the first forward runs on it
        e.set_weight_scale(g.ws)
        e.upload_graph(g)
        assert e.get_info("lds_table_active") == (1 if want_bits else 0)
        assert e.get_info("lds_table_bits") == want_bits
        for rep in range(3):
            _, lg = e.forward(g.x())
            assert np.array_equal(bits(lg[:, 0]), bits(want)), rep
        if want_bits:
            assert e.get_info("lds_table_last_ok") == 1
            # an input that is no k / ws: the device notices, the tile kernel gathers, the bits are still the oracle's
            x2 = (g.x() * np.float32(0.999)).astype(np.float32)
            _, lg = e.forward(x2)
            assert e.get_info("lds_table_last_ok") == 0
            assert np.array_equal(bits(lg[:, 0]), bits(oracle_model.predict(g, x2, stop_after=oracle_model.n_layers - 2)[:, 0]))
        # a wider table than the weights need gives the same bits (forced widths)
        for forced in (10, 16):
            if want_bits and forced > want_bits:
                e.set_option("lds_table_bits", forced)
                e.upload_graph(g)
                assert e.get_info("lds_table_bits") == forced
                _, lg = e.forward(g.x())
                assert e.get_info("lds_table_last_ok") == 1
                assert np.array_equal(bits(lg[:, 0]), bits(want)), forced
        e.set_option("lds_table_bits", 0)
        if want_bits == 8:
            # one vertex heavier than a byte (what a fold leaves behind, include/reduction_graph.hpp:394-396): ten bits
            w2 = w.copy()
            w2[12345] = 300
            g2 = gg.CsrGraph(g.n, g.rowptr, g.col, w2, gg.neighbourhood_weights(g.rowptr, g.col, w2))
            oracle_model.set_weight_scale(g.ws)          # (the scale stays the ORIGINAL graph's largest weight: x > 1 there)
            e.upload_graph(g2)
            assert e.get_info("lds_table_bits") == 10
            _, lg = e.forward(w2.astype(np.float32) / np.float32(g.ws))
            assert e.get_info("lds_table_last_ok") == 1
            want2 = oracle_model.predict(g2, w2.astype(np.float32) / np.float32(g.ws), stop_after=oracle_model.n_layers - 2)[:, 0]
            assert np.array_equal(bits(lg[:, 0]), bits(want2))
    finally:
        e.close()


def test_lds_table_plan_steps_aside(model_text, oracle_model):
    """Weights above 65 535, descending adjacency lists, long rows: the plan must not be used."""
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_option("blocked_stage0", 0)
        g = gg.erdos_renyi(6000, 40000, 64)
        big_w = gg.CsrGraph(g.n, g.rowptr, g.col, g.w * 700, gg.neighbourhood_weights(g.rowptr, g.col, g.w * 700))
        g7 = gg.erdos_renyi(200000, 800000, 65)         # three column blocks: descending lists visit them backwards
        rp = g7.rowptr.astype(np.int64)
        col = g7.col.copy()
        row = np.repeat(np.arange(g7.n), np.diff(rp))
        k = np.arange(rp[-1]) - rp[row]
        col[: rp[-1]] = g7.col[rp[row + 1] - 1 - k]     # every adjacency list reversed
        unsorted = gg.CsrGraph(g7.n, g7.rowptr, col, g7.w, g7.nw)
        hubs = gg.hub_graph(20000, 60000, 3, 4096, seed=7)
        for gr in (big_w, unsorted, hubs):
            e.set_option("lds_table_skewed", 0)         # (the skewed-graph layout has its own test below)
            e.set_weight_scale(gr.ws)
            oracle_model.set_weight_scale(gr.ws)
            e.upload_graph(gr)
            e.forward(gr.x())
            _, logits = e.forward(gr.x())
            assert e.get_info("lds_table_active") == 0
            assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(gr)))
    finally:
        e.close()


@pytest.mark.parametrize("maker,giant", [
    (lambda: gg.rmat(14, 8, 3), 4096),
    (lambda: gg.rmat(14, 8, 3), 0),                                          # no giant rows: k_long_f1 keeps what the plan leaves
    (lambda: gg.hub_graph(30000, 200000, 3, 6000, seed=9), 4096),            # giant rows beside the plan
    (lambda: gg.chung_lu_hubs(40000, 8.0, 2.2, 2, 3000, seed=4), 2048),
    (lambda: gg.hub_graph(200000, 900000, 4, 30000, seed=3), 16384),         # several column blocks, rows of 30 000 entries in the plan's neighbourhood
])
def test_lds_table_plan_on_skewed_graphs(model_text, oracle_model, maker, giant):
    """The LDS-table plan of the F = 1 stage with the skewed-graph layout: every row below the giant-row threshold dealt
    from the degree-sorted list to slices of equal weight, column blocks of equal entry mass; giant rows beside it.
    Same bits as the oracle — also for inputs that are not W / ws (the device-side check) and for row sub-ranges."""
    import torch
    import gnn_mwvc_amd as G
    g = maker()
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_option("blocked_stage0", 0)
        e.set_option("long_row_threshold", 256)
        e.set_option("giant_row_threshold", giant)
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        e.upload_graph(g)
        want = oracle_model.logits(g)
        for rep in range(3):
            _, logits = e.forward(g.x())
            assert np.array_equal(bits(logits[:, 0]), bits(want)), rep
            assert e.get_info("lds_table_active") == (1 if rep else 0)
        assert e.get_info("lds_table_mapped") == 1
        assert e.get_info("lds_table_steps") >= 4 * e.get_info("lds_table_chunks")
        dev = torch.device("cuda:0")
        x = torch.from_numpy(g.x()).to(dev)
        h1 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        want_h1 = oracle_model.predict(g, g.x(), stop_after=6)
        torch.cuda.synchronize()
        e.stage_forward_device(0, 0, g.n, x.data_ptr(), h1.data_ptr())      # the whole range: the plan
        e.synchronize()
        assert np.array_equal(bits(h1[:-1].cpu().numpy()), bits(want_h1))
        h1.fill_(7.0)
        torch.cuda.synchronize()
        mid = (g.n // 3) // 64 * 64
        for lo, hi in ((mid, g.n), (0, mid)):                              # sub-ranges: the gathering kernels
            e.stage_forward_device(0, lo, hi, x.data_ptr(), h1.data_ptr())
        e.synchronize()
        assert np.array_equal(bits(h1[:-1].cpu().numpy()), bits(want_h1))
        rng = np.random.default_rng(5)
        x2 = rng.uniform(0.1, 1.0, size=(g.n, 1)).astype(np.float32)        # not W / ws: the check sends every row down the plain gather
        _, lg2 = e.forward(x2)
        assert np.array_equal(bits(lg2[:, 0]), bits(oracle_model.predict(g, x2, stop_after=oracle_model.n_layers - 2)[:, 0]))
        _, back = e.forward(g.x())
        assert np.array_equal(bits(back[:, 0]), bits(want))
    finally:
        e.close()


# ---------------------------------------------------------------- compact-table plan of the 16-wide stages

def _oracle_stage(oracle_model, g, stage, h_in):
    """One fused stage from an arbitrary input, with the oracle's layer functions."""
    a = oracle_py.graph_layer(g, g.ws, np.ascontiguousarray(h_in, dtype=np.float32))
    for i, (W, b) in enumerate(oracle_model.linear_params()[3 * stage: 3 * stage + 3]):
        a = oracle_py.linear_layer(a, W, b)
        if not (stage == 2 and i == 2):
            a = oracle_py.relu(a)
    return a


def _sparse_features(n, rng, live, density, strays=0, stray_cols=(5, 9)):
    h = np.zeros((n, 16), dtype=np.float32)
    for c, d in zip(live, density):
        h[:, c] = rng.uniform(0.05, 2.0, n).astype(np.float32) * (rng.random(n) < d)
    for i in rng.choice(n, strays, replace=False):
        h[i, stray_cols[i % len(stray_cols)]] = 1.0 + (i % 7)
    return h



def test_skewed_lds_table_plan_without_a_long_row_list(model_text, oracle_model):
    """fuzz_plans.py case 22174 (round 4): the LDS-table plan's skewed layout leaves the rows from its threshold on to the long-row
    kernels — with the long-row list switched off ("long_row_threshold" 0) nobody wrote them, which only showed when the INPUT
    changed between forwards (the graph's first, plan-less forward had written them for the first input).  The plan now takes
    every row when there is no list."""
    import gnn_mwvc_amd as G
    g = gg.rmat(12, 12, 7)
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)
    x2 = (g.x() * np.float32(0.37)).astype(np.float32)          # no k / ws: the plan steps aside for this input
    x3 = (g.x() * np.float32(2.0)).astype(np.float32)           # 2 k / ws: another input the plan serves
    w2, w3 = oracle_model.logits(g, x2), oracle_model.logits(g, x3)
    for rows in (64, 0):
        e = G.Engine(model_text, device=0)
        try:
            for k, v in (("blocked_min_n", 0), ("long_row_threshold", 0), ("sorted_tiles", 1), ("lds_table", 1),
                         ("lds_table_skewed_rows", rows)):
                e.set_option(k, v)
            e.set_weight_scale(g.ws)
            e.upload_graph(g)
            for rep in range(3):
                _, lg = e.forward(g.x())
                assert np.array_equal(bits(lg[:, 0]), bits(want)), (rows, rep)
            assert e.get_info("lds_table_active") == 1 and e.get_info("long_rows") == 0
            for x, w in ((x2, w2), (x3, w3), (g.x(), want), (x3, w3)):
                _, lg = e.forward(x)
                assert np.array_equal(bits(lg[:, 0]), bits(w)), rows
        finally:
            e.close()


@pytest.mark.parametrize("case", ["clean", "strays", "two_live", "too_many_strays", "negative", "minus_zero", "slot_overflow"])
def test_compact_gather_plan_is_bit_identical(model_text, oracle_model, case):
    """k_c4_*: when at most four feature columns are live, neighbours are read from a 16-byte table swept through
    L2; rows that meet a vertex with stray non-zeros are recomputed from full rows.  Whatever the input, the
    stage output equals the oracle's bit for bit (inputs that do not fit the plan take the plain gather)."""
    import torch
    import gnn_mwvc_amd as G
    rng = np.random.default_rng(len(case))
    g = gg.erdos_renyi(3000, 3000 * 150, 71) if case == "slot_overflow" else gg.erdos_renyi(20000, 200000, 70)
    live, dens = ([0, 3, 7, 11], [1.0, 0.1, 0.5, 1.0])
    strays = {"clean": 0, "strays": 25, "too_many_strays": 400, "slot_overflow": 5}.get(case, 3)
    if case == "two_live":
        live, dens = [2, 13], [1.0, 0.3]
    h = _sparse_features(g.n, rng, live, dens, strays)
    if case == "negative":
        h[17, 0] = -0.25
    if case == "minus_zero":
        h[::3, 3] = -0.0
        h[5, 9] = -0.0                    # a negative zero in a dead column is no stray
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        e.upload_graph(g)
        e.forward(g.x())
        _, logits = e.forward(g.x())                       # second forward: the plans are built
        assert e.get_info("compact_gather_active") == 1
        assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g)))
        dev = torch.device("cuda:0")
        hin = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        hin[: g.n] = torch.from_numpy(h).to(dev)
        stray_vertex = (h[:, [c for c in range(16) if c not in live]] != 0).any(axis=1)
        rp = g.rowptr.astype(np.int64)
        hits = np.concatenate(([0], np.cumsum(stray_vertex[g.col[: rp[-1]]])))
        dirty_row = hits[rp[1:]] > hits[rp[:-1]]
        for stage in (1, 2):
            want = _oracle_stage(oracle_model, g, stage, h)
            for lo, hi in ((0, g.n), (g.n // 4 // 64 * 64, g.n), (0, g.n // 8 // 64 * 64)):   # the last one is too short for the plan
                if stage == 1:
                    out = torch.full((g.n + 1, 16), 7.0, dtype=torch.float32, device=dev)
                    torch.cuda.synchronize()
                    e.stage_forward_device(1, lo, hi, hin.data_ptr(), out.data_ptr())
                    e.synchronize()
                    got = out[lo:hi].cpu().numpy()
                else:
                    out = torch.full((g.n + 1,), 7.0, dtype=torch.float32, device=dev)
                    lg = torch.full((g.n + 1,), 7.0, dtype=torch.float32, device=dev)
                    torch.cuda.synchronize()
                    e.stage_forward_device(2, lo, hi, hin.data_ptr(), out.data_ptr(), lg.data_ptr())
                    e.synchronize()
                    got = lg[lo:hi].cpu().numpy().reshape(-1, 1)
                assert np.array_equal(bits(got), bits(want[lo:hi])), (case, stage, lo, hi)
                if (hi - lo) * 2 >= g.n:      # the plan ran: did the device take the route this case is about?
                    ok, dirty = e.get_info("compact_gather_last_ok"), e.get_info("compact_gather_last_dirty")
                    assert ok == (0 if case in ("too_many_strays", "negative") else 1), (case, ok)
                    if ok:
                        # (with two live columns the two stray columns simply become the other two table columns)
                        assert (dirty > 0) == (strays > 0 and case != "two_live"), (case, dirty)
                        if case != "two_live":
                            # dirty rows = rows with a neighbour that has a non-zero outside the table's columns
                            # (chunks that straddle the ends of [lo, hi) are done whole)
                            assert int(dirty_row[lo:hi].sum()) <= dirty <= int(dirty_row.sum()), (case, lo, hi, dirty)
                            if lo == 0 and hi == g.n:
                                assert dirty == int(dirty_row.sum()), (case, dirty)
                        if case == "slot_overflow":
                            assert dirty > g.n // 4
    finally:
        e.close()


def test_compact_gather_long_runs(model_text, oracle_model):
    """k_c4_agg forced onto a skewed graph (three hubs of degree 4096, no long-row kernels): runs of thousands of
    entries of one row cross lanes, waves' steps and — with small chunks — sit alone in a one-row slice."""
    import torch
    import gnn_mwvc_amd as G
    g = gg.hub_graph(20000, 60000, 3, 4096, seed=7)
    for chunk_rows in (0, 16):
        e = G.Engine(model_text, device=0)
        try:
            e.set_option("blocked_min_n", 0)
            e.set_option("compact_gather", 2)
            e.set_option("lds_table", 2)
            e.set_option("long_row_threshold", 0)
            e.set_option("sorted_tiles", 0)
            e.set_option("plan_chunk_rows", chunk_rows)
            e.set_weight_scale(g.ws)
            oracle_model.set_weight_scale(g.ws)
            e.upload_graph(g)
            e.forward(g.x())
            _, logits = e.forward(g.x())
            assert e.get_info("compact_gather_active") == 1
            assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g)))
            rng = np.random.default_rng(11)
            h = _sparse_features(g.n, rng, [0, 5, 9, 12], [1.0, 0.3, 0.6, 1.0], strays=5)
            dev = torch.device("cuda:0")
            hin = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
            hin[: g.n] = torch.from_numpy(h).to(dev)
            for stage in (1, 2):
                out = torch.full((g.n + 1, 16 if stage == 1 else 1), 7.0, dtype=torch.float32, device=dev)
                lg = torch.full((g.n + 1,), 7.0, dtype=torch.float32, device=dev)
                torch.cuda.synchronize()
                e.stage_forward_device(stage, 0, g.n, hin.data_ptr(), out.data_ptr(), lg.data_ptr() if stage == 2 else 0)
                e.synchronize()
                assert e.get_info("compact_gather_last_ok") == 1
                want = _oracle_stage(oracle_model, g, stage, h)
                got = out[: g.n].cpu().numpy() if stage == 1 else lg[: g.n].cpu().numpy().reshape(-1, 1)
                assert np.array_equal(bits(got), bits(want)), (chunk_rows, stage)
        finally:
            e.close()


@pytest.mark.parametrize("mode", [1])
@pytest.mark.parametrize("maker", [
    lambda: gg.rmat(14, 8, 3),
    lambda: gg.hub_graph(30000, 200000, 3, 6000, seed=9),
    lambda: gg.chung_lu_hubs(40000, 8.0, 2.2, 2, 3000, seed=4),
])
def test_pruned_adjacency_is_bit_identical(model_text, oracle_model, maker, mode):
    """k_prune_*: on skewed graphs the model drives the features of high-degree vertices to zero; the entries that point
    to rows taken to be all zero — the rows found so when the plan is built — are dropped from a second CSR, and every call proves on the device that its input fits (else the full
    adjacency is used).  Whole forwards and single stages — fitting inputs, inputs that break the premise, row ranges —
    equal the oracle bit for bit."""
    import torch
    import gnn_mwvc_amd as G
    g = maker()
    deg = np.diff(g.rowptr.astype(np.int64))
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_option("long_row_threshold", 256)
        e.set_option("sorted_long_row_threshold", 512)
        e.set_option("giant_row_threshold", 4096)
        e.set_option("prune_zero_rows", mode)
        e.set_option("prune_min_drop_percent", 1)
        e.set_option("prune_min_entries", 0)
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        e.upload_graph(g)
        want_logits = oracle_model.logits(g)
        for rep in range(4):                                 # the plans are built in the second forward
            _, logits = e.forward(g.x())
            assert np.array_equal(bits(logits[:, 0]), bits(want_logits)), rep
        built = [e.get_info("pruned_stage1"), e.get_info("pruned_stage2")]
        assert any(built), "no stage of this graph had entries to drop"
        # the vertices whose rows the plan of a stage takes to be zero
        stage_in = {1: oracle_model.predict(g, g.x(), stop_after=6), 2: oracle_model.predict(g, g.x(), stop_after=13)}
        zero_set = {}
        for st in (1, 2):
            if not built[st - 1]:
                continue
            assert e.get_info(f"pruned_last_ok_stage{st}") == 1
            assert 0 < e.get_info(f"pruned_entries_stage{st}") < g.nnz
            if True:
                zero_set[st] = np.flatnonzero(~(stage_in[st] != 0).any(axis=1))
                rp = g.rowptr.astype(np.int64)
                dropped = np.isin(g.col[: rp[-1]], zero_set[st]).sum()
                assert e.get_info(f"pruned_entries_stage{st}") == g.nnz - dropped
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(5)
        for case in ("fits", "one_nonzero", "minus_zero", "all_dense"):
            h = (rng.uniform(0.05, 2.0, (g.n, 16)) * (rng.random((g.n, 16)) < 0.5)).astype(np.float32)
            for st in (1, 2):
                hs = h.copy()
                zs = zero_set.get(st, np.zeros(0, dtype=np.int64))
                if case != "all_dense":
                    hs[zs] = 0.0
                if case == "one_nonzero" and len(zs):
                    hs[zs[len(zs) // 2], 7] = 0.5            # breaks the premise: this call must use the full adjacency
                if case == "minus_zero" and len(zs):
                    hs[zs[::2], 3] = -0.0                    # a zero of either sign is a zero
                hin = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
                hin[: g.n] = torch.from_numpy(hs).to(dev)
                want = _oracle_stage(oracle_model, g, st, hs)
                for lo, hi in ((0, g.n), (g.n // 3 // 64 * 64, g.n // 3 * 2)):
                    out = torch.full((g.n + 1, 16 if st == 1 else 1), 7.0, dtype=torch.float32, device=dev)
                    lg = torch.full((g.n + 1,), 7.0, dtype=torch.float32, device=dev)
                    torch.cuda.synchronize()
                    e.stage_forward_device(st, lo, hi, hin.data_ptr(), out.data_ptr(), lg.data_ptr() if st == 2 else 0)
                    e.synchronize()
                    if built[st - 1] and len(zs):
                        fits = case in ("fits", "minus_zero")
                        assert e.get_info(f"pruned_last_ok_stage{st}") == (1 if fits else 0), (case, st)
                    got = out[lo:hi].cpu().numpy() if st == 1 else lg[lo:hi].cpu().numpy().reshape(-1, 1)
                    assert np.array_equal(bits(got), bits(want[lo:hi])), (case, st, lo, hi)
        e.set_option("prune_zero_rows", 0)                   # and the same forward without the plan
        _, logits = e.forward(g.x())
        assert e.get_info("pruned_stage1") == 0 and e.get_info("pruned_stage2") == 0
        assert np.array_equal(bits(logits[:, 0]), bits(want_logits))
    finally:
        e.close()


@pytest.mark.parametrize("maker", [
    lambda: gg.rmat(14, 8, 3),
    lambda: gg.rmat(15, 16, 5),
    lambda: gg.hub_graph(30000, 200000, 3, 6000, seed=9),
    lambda: gg.chung_lu_hubs(40000, 8.0, 2.2, 2, 3000, seed=4),
])
def test_predicted_pruned_adjacency_is_bit_identical(model_text, oracle_model, maker):
    """Round 4: a large skewed graph gets the first 16-wide stage's pruned adjacency when it is HANDED OVER, from the set of zero
    rows its own weights predict (the reference's driver feeds x = W / ws), and the next stage borrows it in the graph's first
    forward.  A prediction is never trusted: every call proves on the device that the set's rows are all zero in ITS input and
    takes the full adjacency otherwise — so whatever the caller feeds (the driver's input, a scaled one, a negated one, noise)
    every forward equals the oracle bit for bit; a prediction that failed is dropped when the graph comes back."""
    import gnn_mwvc_amd as G
    g = maker()
    rng = np.random.default_rng(5)
    inputs = {"driver": g.x(), "half": (g.x() * np.float32(0.5)).astype(np.float32), "negated": (-g.x()).astype(np.float32),
              "noise": rng.uniform(0.0, 40.0, g.x().shape).astype(np.float32)}
    oracle_model.set_weight_scale(g.ws)
    seen_ok, seen_fail = 0, 0
    for name, x in inputs.items():
        want = oracle_model.predict(g, x, stop_after=oracle_model.n_layers - 2)[:, 0]
        e = G.Engine(model_text, device=0)
        try:
            for k, v in (("blocked_min_n", 0), ("long_row_threshold", 256), ("sorted_long_row_threshold", 512), ("giant_row_threshold", 4096),
                         ("prune_min_entries", 0), ("prune_predict_min_entries", 0), ("filter_min_long_percent", 0), ("prune_min_drop_percent", 1)):
                e.set_option(k, v)
            e.set_weight_scale(g.ws)
            e.upload_graph(g)
            predicted = e.get_info("pruned_predicted_stage1")
            _, lg = e.forward(x)
            assert np.array_equal(bits(lg[:, 0]), bits(want)), (name, "first")
            ok1 = e.get_info("pruned_last_ok_stage1") if predicted else -1
            if predicted:
                assert e.get_info("pruned_borrowed_stage2") == 1 and e.get_info("filtered_stage1") == 0
                seen_ok += ok1 == 1
                seen_fail += ok1 == 0
            for rep in range(3):
                _, lg = e.forward(x)
                assert np.array_equal(bits(lg[:, 0]), bits(want)), (name, rep)
                if predicted and rep == 0:      # the graph came back: a prediction that held stays, one that failed is gone
                    assert e.get_info("pruned_predicted_stage1") == (1 if ok1 == 1 else 0), name
            if name == "driver" and predicted:
                assert ok1 == 1                 # the prediction is made for exactly this input
        finally:
            e.close()
    # without the option the same hand-off builds nothing
    e = G.Engine(model_text, device=0)
    try:
        for k, v in (("blocked_min_n", 0), ("prune_min_entries", 0), ("prune_predict_min_entries", 0), ("filter_min_long_percent", 0),
                     ("prune_predict", 0)):
            e.set_option(k, v)
        e.set_weight_scale(g.ws)
        e.upload_graph(g)
        assert e.get_info("pruned_predicted_stage1") == 0 and e.get_info("pruned_stage1") == 0
    finally:
        e.close()


@pytest.mark.parametrize("min_percent", [1, 101])
@pytest.mark.parametrize("maker", [
    lambda: gg.rmat(14, 8, 3),
    lambda: gg.hub_graph(30000, 200000, 3, 6000, seed=9),
    lambda: gg.chung_lu_hubs(40000, 8.0, 2.2, 2, 3000, seed=4),
])
def test_filtered_first_forward_is_bit_identical(model_text, oracle_model, maker, min_percent):
    """A skewed graph scored ONCE (what the reference's driver does) has no pruned adjacency: its 16-wide stages look every
    entry's target up in the bitmap of THIS input's all-zero rows (k_filter_mark) and fetch the pad row instead, the long rows
    walk lists shortened by a pass in front of them (k_long_lists), and the next stage shortens those lists further when the
    device finds the earlier set still all zero in its input.  min_percent 1: the set always counts as worth it; 101: never —
    the filtered instantiations run, nothing is written, nothing may be read.  Whole first forwards and single stages
    (fitting inputs, inputs that break the earlier stage's set, row ranges) equal the oracle bit for bit."""
    import torch
    import gnn_mwvc_amd as G
    g = maker()
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_option("long_row_threshold", 256)
        e.set_option("sorted_long_row_threshold", 512)
        e.set_option("giant_row_threshold", 4096)
        e.set_option("filter_min_entries", 0)
        e.set_option("filter_min_long_percent", 0)
        e.set_option("filter_min_percent", min_percent)
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        want_logits = oracle_model.logits(g)
        for _ in range(2):                                   # twice a FIRST forward: the graph handed over anew
            e.upload_graph(g)
            _, logits = e.forward(g.x())
            assert e.get_info("graph_uses") == 1
            assert e.get_info("filtered_stage1") == 1 and e.get_info("filtered_stage2") == 1
            assert e.get_info("pruned_stage1") == 0 and e.get_info("pruned_stage2") == 0
            assert e.get_info("long_rows") > 0 and e.get_info("short_lists_stage2") == 1
            assert np.array_equal(bits(logits[:, 0]), bits(want_logits))
        if min_percent == 1:
            assert e.get_info("filter_mass_percent_stage1") >= 1
        # single stages on the once-scored graph (still no plan: stage calls do not count as uses)
        stage_in = {1: oracle_model.predict(g, g.x(), stop_after=6), 2: oracle_model.predict(g, g.x(), stop_after=13)}
        zero1 = np.flatnonzero(~(stage_in[1] != 0).any(axis=1))
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(11)

        def stage(st, hs, lo, hi):
            hin = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
            hin[: g.n] = torch.from_numpy(hs).to(dev)
            out = torch.full((g.n + 1, 16 if st == 1 else 1), 7.0, dtype=torch.float32, device=dev)
            lg = torch.full((g.n + 1,), 7.0, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            e.stage_forward_device(st, lo, hi, hin.data_ptr(), out.data_ptr(), lg.data_ptr() if st == 2 else 0)
            e.synchronize()
            got = out[lo:hi].cpu().numpy() if st == 1 else lg[lo:hi].cpu().numpy().reshape(-1, 1)
            want = _oracle_stage(oracle_model, g, st, hs)
            assert np.array_equal(bits(got), bits(want[lo:hi])), (st, lo, hi)

        third = g.n // 3 // 64 * 64
        for case in ("same_set", "bigger_set", "broken_set", "dense", "after_a_range"):
            h1 = (rng.uniform(0.05, 2.0, (g.n, 16)) * (rng.random((g.n, 16)) < 0.5)).astype(np.float32)
            h1[zero1] = 0.0                                  # stage 1's input: the model's own zero rows
            h2 = (rng.uniform(0.05, 2.0, (g.n, 16)) * (rng.random((g.n, 16)) < 0.5)).astype(np.float32)
            if case != "dense":
                h2[zero1] = 0.0
            if case == "bigger_set":
                h2[rng.random(g.n) < 0.3] = 0.0
                h2[zero1[::2], 5] = -0.0                     # a zero of either sign is a zero
            if case == "broken_set" and len(zero1):
                h2[zero1[len(zero1) // 2], 9] = 0.25         # the earlier stage's lists do not stand for this input
            if case == "after_a_range":
                stage(1, h1, third, 2 * third)               # a call that does not see every row leaves nothing behind
            else:
                stage(1, h1, 0, g.n)
            for lo, hi in ((0, g.n), (third, 2 * third)):
                stage(2, h2, lo, hi)
                if min_percent == 1 and len(zero1) and case != "after_a_range":
                    assert e.get_info("short_lists_stage2") == 1
                if case == "after_a_range":
                    assert e.get_info("short_lists_stage2") == 0
        e.set_option("filter_zero_rows", 0)                  # and the same first forward without any of it
        e.upload_graph(g)
        _, logits = e.forward(g.x())
        assert e.get_info("filtered_stage1") == 0 and e.get_info("filtered_stage2") == 0
        assert np.array_equal(bits(logits[:, 0]), bits(want_logits))
    finally:
        e.close()
    # the same once-scored graph cut into three parts behind one handle: every part filters with the whole input's bitmap and
    # shortens the lists of the long rows IT holds
    m = G.Engine(model_text, devices=[0, 0, 0])
    try:
        for k, v in (("blocked_min_n", 0), ("long_row_threshold", 256), ("sorted_long_row_threshold", 512), ("giant_row_threshold", 4096),
                     ("filter_min_entries", 0), ("filter_min_long_percent", 0), ("filter_min_percent", min_percent)):
            m.set_option(k, v)
        m.set_weight_scale(g.ws)
        m.upload_graph(g)
        _, logits = m.forward(g.x())
        assert np.array_equal(bits(logits[:, 0]), bits(want_logits))
    finally:
        m.close()


def test_side_queue_runs_beside_the_main_stream(model_text):
    """HIP places a new stream on the hardware queue with the fewest users — ties included, which can be the queue of the
    engine's own main stream: the two are then serialised and a forward with long rows takes 1.3 - 1.4 x as long.  Every engine
    probes its side queue at creation and replaces it until a kernel on it finishes while the main stream is still busy
    (gnnvc_get_info "side_queue_runs_beside" / "side_queue_probes"), whatever other engines the process has alive."""
    import gnn_mwvc_amd as G
    alive = []
    try:
        for k in range(6):
            e = G.Engine(model_text, device=0)
            alive.append(e)
            assert 1 <= e.get_info("side_queue_probes") <= 4
            assert e.get_info("side_queue_runs_beside") == 1, k
    finally:
        for e in alive:
            e.close()


def test_side_queue_is_probed_again_for_a_callers_stream(model_text, oracle_model):
    """the probe pairs the side queue with the engine's OWN main stream; a caller's stream (gnnvc_set_stream) may
    share the side queue's hardware queue.  Installing one probes the pair again (and replaces the side queue if they are
    serialised); going back to the engine's stream does too.  Logits unchanged on a graph with long rows."""
    import gnn_mwvc_amd as G
    import torch
    g = gg.hub_graph(30000, 300000, 6, 9000, seed=5)   # long rows: the side queue has work
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)
    e = G.Engine(model_text, device=0)
    try:
        e.set_weight_scale(g.ws)
        e.upload_graph(g)
        for k in range(4):
            mine = torch.cuda.Stream()
            e.set_stream(mine.cuda_stream)
            try:
                assert e.get_info("side_queue_runs_beside") == 1, k
                assert 1 <= e.get_info("side_queue_probes") <= 5
                _, logits = e.forward(g.x())
                assert np.array_equal(bits(logits[:, 0]), bits(want))
            finally:
                e.set_stream(None)
            assert e.get_info("side_queue_runs_beside") == 1
    finally:
        e.close()


def test_compact_gather_plan_steps_aside_for_good(model_text, oracle_model):
    """A graph whose stage inputs never fit the plan (low degrees: more than four live columns): after three forwards in a row
    that the device sent down the gathering kernels, the engine stops queuing the plan's counting, choosing and empty
    launches for that stage of this graph; a new graph starts afresh.  Same logits throughout."""
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        g = gg.erdos_renyi(30000, 120000, 77)              # 8 entries per row
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        e.upload_graph(g)
        want = oracle_model.logits(g)
        off = []
        for rep in range(12):
            _, lg = e.forward(g.x())
            assert np.array_equal(bits(lg[:, 0]), bits(want)), rep
            e.synchronize()
            off.append((e.get_info("compact_gather_off_stage1"), e.get_info("compact_gather_off_stage2")))
        assert e.get_info("compact_gather_active") == 1
        fits = [e.get_info("compact_gather_last_ok")]
        assert off[2] == (0, 0)                              # not before three verdicts are in
        assert off[-1] != (0, 0), off                        # at least one stage gave up
        g2 = gg.erdos_renyi(20000, 200000, 70)               # the graph of the plan's own test: fits
        e.set_weight_scale(g2.ws)
        oracle_model.set_weight_scale(g2.ws)
        e.upload_graph(g2)
        for rep in range(6):
            _, lg = e.forward(g2.x())
            assert np.array_equal(bits(lg[:, 0]), bits(oracle_model.logits(g2))), rep
        assert e.get_info("compact_gather_off_stage1") == 0 and e.get_info("compact_gather_off_stage2") == 0
        assert e.get_info("compact_gather_last_ok") == 1
    finally:
        e.close()


def test_plan_built_inside_the_first_forward(model_text, oracle_model):
    """"compact_first_forward_entries": a graph with that many entries builds the compact-table plan inside its first forward
    and uses it there (what a score-once caller runs); the LDS-table plan follows in the second.  Same logits throughout, and
    the same as with the plan built one forward later."""
    import gnn_mwvc_amd as G
    g = gg.erdos_renyi(20000, 200000, 70)
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)
    for first_too in (1, 0):
        e = G.Engine(model_text, device=0)
        try:
            e.set_option("blocked_min_n", 0)
            e.set_option("compact_first_forward_entries", g.nnz if first_too else 0)
            e.set_weight_scale(g.ws)
            e.upload_graph(g)
            _, lg = e.forward(g.x())
            assert np.array_equal(bits(lg[:, 0]), bits(want))
            assert e.get_info("compact_gather_active") == first_too
            if first_too:
                assert e.get_info("compact_gather_last_ok") == 1      # ... and the device took the plan's route
            assert e.get_info("lds_table_active") == 0
            for rep in range(3):
                _, lg = e.forward(g.x())
                assert np.array_equal(bits(lg[:, 0]), bits(want)), rep
            assert e.get_info("compact_gather_active") == 1 and e.get_info("lds_table_active") == 1
        finally:
            e.close()


@pytest.mark.parametrize("route", ["upload", "staged", "attach"])
def test_plans_built_at_hand_off(model_text, oracle_model, route):
    """"plans_at_handoff" (round 3): what depends on the graph alone exists when the hand-off returns — before any forward —
    on every hand-off route, the first forward runs on it (table columns from the pilot), and the bits are the oracle's.
    With 0 the plans wait for the graph's first forwards as in round 2."""
    import gnn_mwvc_amd as G
    g = gg.erdos_renyi(40000, 400000, 73)
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)

    def hand_over(e):
        if route == "upload":
            e.upload_graph(g)
        elif route == "staged":
            e.upload_graph_staged(g, pieces=3)
        else:
            import torch
            from tools import graphgen_torch as ggt
            dg = ggt.from_host(g, torch.device("cuda:0"))
            torch.cuda.synchronize()
            e.attach_graph_device(dg.n, dg.nnz, dg.rowptr.data_ptr(), dg.col.data_ptr(), dg.w.data_ptr(), dg.nw.data_ptr(), keepalive=dg)

    for handoff in (1, 2, 0):
        e = G.Engine(model_text, device=0)
        try:
            e.set_option("blocked_min_n", 0)              # let the plans apply to a small graph
            e.set_option("plans_at_handoff", handoff)
            e.set_option("handoff_min_entries", 1)
            e.set_option("pilot_rows", 2048)
            e.set_weight_scale(g.ws)
            hand_over(e)
            assert e.get_info("graph_uses") == 0
            built = 1 if handoff else 0
            assert e.get_info("lds_table_active") == built and e.get_info("compact_gather_active") == built, (route, handoff)
            assert (e.get_info("handoff_build_us") > 0) == bool(built)
            _, lg = e.forward(g.x())
            assert np.array_equal(bits(lg[:, 0]), bits(want)), (route, handoff)
            if built:
                assert e.get_info("compact_gather_last_ok") == 1      # the device took the plan's route in the FIRST forward
            for rep in range(3):
                _, lg = e.forward(g.x())
                assert np.array_equal(bits(lg[:, 0]), bits(want)), (route, handoff, rep)
            assert e.get_info("lds_table_active") == 1 and e.get_info("compact_gather_active") == 1
            # the next graph handed to the same engine gets its own plans (and none of this one's)
            g2 = gg.erdos_renyi(30000, 330000, 74)
            e.set_weight_scale(g2.ws)
            oracle_model.set_weight_scale(g2.ws)
            e.upload_graph(g2)
            assert e.get_info("lds_table_active") == built and e.get_info("graph_uses") == 0
            _, lg = e.forward(g2.x())
            assert np.array_equal(bits(lg[:, 0]), bits(oracle_model.logits(g2))), (route, handoff)
            oracle_model.set_weight_scale(g.ws)
        finally:
            e.close()


def test_pilot_choice_never_changes_a_result(model_text, oracle_model):
    """The pilot picks the next stage's table columns from the FIRST rows of a stage.  Here those rows are unlike the rest
    (their weights are tiny, other columns light up): the consumer's choice from all rows differs, the table is rewritten,
    the logits are the oracle's — on the first forward and on the ones after it."""
    import gnn_mwvc_amd as G
    g = gg.erdos_renyi(40000, 400000, 75)
    w = g.w.copy()
    w[:4096] = 1
    g = gg.CsrGraph(g.n, g.rowptr, g.col, w, gg.neighbourhood_weights(g.rowptr, g.col, w))
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)
    for pilot in (2048, 0):
        e = G.Engine(model_text, device=0)
        try:
            e.set_option("blocked_min_n", 0)
            e.set_option("handoff_min_entries", 1)
            e.set_option("pilot_rows", pilot)
            e.set_weight_scale(g.ws)
            e.upload_graph(g)
            for rep in range(3):
                _, lg = e.forward(g.x())
                assert np.array_equal(bits(lg[:, 0]), bits(want)), (pilot, rep)
        finally:
            e.close()


def test_whole_graph_calls_on_a_slice_are_state_errors(model_text):
    """the host-pointer gnnvc_forward on an engine that holds a slice (or an empty slice) must say
    GNNVC_ERR_STATE before it touches a buffer — a sliced engine has no feature buffers of its own."""
    import gnn_mwvc_amd as G
    import torch
    from gnn_mwvc_amd import distributed as D
    g = gg.erdos_renyi(3000, 20000, 9)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a.astype(np.int64)).to(torch.int32).to(dev)
    rp, col, w, nw = t(g.rowptr), t(g.col), t(g.w), t(g.nw)
    e = G.Engine(model_text, device=0)
    try:
        e.set_weight_scale(g.ws)
        for lo, hi in ((1000, 2500), (0, 0), (1200, 1200)):
            sl = D.slice_csr(g.n, rp, col, w, nw, lo, hi)
            torch.cuda.synchronize()
            e.attach_graph_slice(g.n, lo, hi, sl.nnz, sl.rowptr.data_ptr(), sl.col.data_ptr(), sl.w.data_ptr(), sl.nw.data_ptr(),
                                 keepalive=sl)
            e.n = g.n
            with pytest.raises(G.GnnvcError) as err:
                e.forward(g.x())
            assert err.value.code == -4, (lo, hi, err.value)          # GNNVC_ERR_STATE
        # ... also right after the engine held a (smaller) whole graph: its stale buffers must not be written
        small = gg.erdos_renyi(100, 300, 1)
        e.upload_graph(small)
        e.forward(small.x())
        sl = D.slice_csr(g.n, rp, col, w, nw, 500, 2000)
        torch.cuda.synchronize()
        e.attach_graph_slice(g.n, 500, 2000, sl.nnz, sl.rowptr.data_ptr(), sl.col.data_ptr(), sl.w.data_ptr(), sl.nw.data_ptr(), keepalive=sl)
        e.n = g.n
        with pytest.raises(G.GnnvcError) as err:
            e.forward(g.x())
        assert err.value.code == -4
    finally:
        e.close()


def test_derive_commit_after_a_graph_change_is_refused(model_text, oracle_model):
    """a derivation begun against one resident graph cannot be committed against another."""
    import ctypes as C
    import gnn_mwvc_amd as G
    rng = np.random.default_rng(3)
    g0 = gg.erdos_renyi(4000, 24000, 5)
    other = gg.erdos_renyi(900, 3000, 6)
    g1, old_row = _shrunk_graph(g0, rng, 0.7, 20, 10)
    e = G.Engine(model_text, device=0)
    try:
        e.upload_graph(g0)
        L = e._L
        rp = np.ascontiguousarray(g1.rowptr.astype(np.uint32))
        tail = np.zeros(g1.n, dtype=np.uint32)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        assert L.gnnvc_derive_graph_begin(e._h, g1.n, ptr(np.ascontiguousarray(old_row, dtype=np.uint32)), ptr(rp), ptr(tail)) == 0
        e.upload_graph(other)                                # the resident graph changes between begin and commit
        ends = g1.rowptr.astype(np.int64)[1:]
        pieces = [g1.col[en - t: en] for en, t in zip(ends[tail > 0], tail[tail > 0])]
        tails = np.ascontiguousarray(np.concatenate(pieces) if pieces else np.zeros(0, np.uint32), dtype=np.uint32)
        rc = L.gnnvc_derive_graph_commit(e._h, ptr(tails), tails.size, ptr(np.ascontiguousarray(g1.w)), ptr(np.ascontiguousarray(g1.nw)))
        assert rc == -4                                      # GNNVC_ERR_STATE
        e.set_weight_scale(other.ws)                         # ... and the graph that IS resident still scores
        oracle_model.set_weight_scale(other.ws)
        _, lg = e.forward(other.x())
        assert np.array_equal(bits(lg[:, 0]), bits(oracle_model.logits(other)))
    finally:
        e.close()


@pytest.mark.parametrize("maker", [
    lambda: gg.erdos_renyi(30000, 45000, 81),                      # 3 entries per row: one row in twenty is empty, often several in a row
    lambda: gg.from_edge_list(3000, [(i, i + 1) for i in range(0, 2999, 7)] + [(5, j) for j in range(900, 1100)], [20 + i % 100 for i in range(3000)]),   # mostly empty rows, one of 200 entries
    lambda: _dense_graph(3000, 300, 63),                           # runs of a row that cross trips and waves' shares
])
def test_flat_plan_builders_on_ragged_rows(model_text, oracle_model, maker):
    """k_lt_count_flat / k_lt_scatter_flat (plans over consecutive rows): empty rows (the row of an entry is then searched, not
    read off the row-start flags), rows longer than a trip, slices with a handful of entries."""
    import gnn_mwvc_amd as G
    g = maker()
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)
    for chunk_rows in (0, 48):
        e = G.Engine(model_text, device=0)
        try:
            e.set_option("blocked_min_n", 0)
            e.set_option("long_row_threshold", 0)                  # no long-row kernels: the plans hold every row
            e.set_option("giant_row_threshold", 0)
            e.set_option("sorted_tiles", 0)
            e.set_option("compact_first_forward_entries", 1)
            if chunk_rows:
                e.set_option("plan_chunk_rows", chunk_rows)
            e.set_weight_scale(g.ws)
            e.upload_graph(g)
            for rep in range(3):
                _, lg = e.forward(g.x())
                assert np.array_equal(bits(lg[:, 0]), bits(want)), (chunk_rows, rep)
            assert e.get_info("lds_table_active") == 1 and e.get_info("compact_gather_active") == 1
        finally:
            e.close()


def test_compact_gather_plan_rejects_unsorted_lists(model_text, oracle_model):
    """The flat builders check while regrouping that a row's column blocks ascend: with descending adjacency lists over several
    column blocks neither plan may be used (regrouping by block would change the order of the sums)."""
    import gnn_mwvc_amd as G
    g7 = gg.erdos_renyi(200000, 800000, 65)
    rp = g7.rowptr.astype(np.int64)
    row = np.repeat(np.arange(g7.n), np.diff(rp))
    k = np.arange(rp[-1]) - rp[row]
    col = g7.col.copy()
    col[: rp[-1]] = g7.col[rp[row + 1] - 1 - k]                    # every adjacency list reversed
    g = gg.CsrGraph(g7.n, g7.rowptr, col, g7.w, g7.nw)
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_option("blocked_stage0", 0)
        e.set_option("compact_first_forward_entries", 1)
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        e.upload_graph(g)
        want = oracle_model.logits(g)
        for rep in range(3):
            _, lg = e.forward(g.x())
            assert np.array_equal(bits(lg[:, 0]), bits(want)), rep
        assert e.get_info("compact_gather_active") == 0 and e.get_info("lds_table_active") == 0
    finally:
        e.close()


@pytest.mark.parametrize("maker", [
    lambda: gg.erdos_renyi(1933, 7000, 61),          # the reference CLI's later predict calls on ER-100K: 1 933 ...
    lambda: gg.erdos_renyi(7375, 30000, 62),         # ... 7 375 ...
    lambda: gg.erdos_renyi(19675, 90000, 63),        # ... 19 675 vertices
    lambda: gg.erdos_renyi(65, 200, 64),             # a second tile of one vertex
    lambda: gg.erdos_renyi(3, 2, 65),                # fewer vertices than a quad row
    lambda: gg.erdos_renyi(5000, 400000, 66),        # 160 entries a row: forty rounds of four
    lambda: gg.from_edge_list(300, [(0, 1), (5, 299)], [30, 40, 50] * 100),   # nearly every row empty
])
def test_wide_tiles_are_bit_identical(model_text, oracle_model, maker):
    """Round 4: graphs with fewer 64-vertex tiles than the chip has SIMDs (the reference CLI's predict calls 2..n,
    src/GNN_VC.cpp:171-192) run every stage a WORKGROUP per tile — the gather on quads of lanes over four waves, each dense layer's
    outputs a quarter per wave (k_stage_w1 / k_stage_w16).  Same fma chains, same bits: whole forwards, arbitrary inputs, stage
    calls over row ranges, and the option off."""
    import torch
    import gnn_mwvc_amd as G
    g = maker()
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)
    e = G.Engine(model_text, device=0)
    try:
        e.set_weight_scale(g.ws)
        e.upload_graph(g)
        for rep in range(2):
            sc, lg = e.forward(g.x())
            assert np.array_equal(bits(lg[:, 0]), bits(want)), rep
        assert e.get_info("wide_tiles_used") == 1
        rng = np.random.default_rng(3)
        x2 = rng.normal(size=g.n).astype(np.float32)
        x2[rng.integers(0, g.n, 3)] = np.array([np.inf, -0.0, 1e30], dtype=np.float32)
        w2 = oracle_model.logits(g, x2)
        _, lg = e.forward(x2)
        same = (bits(lg[:, 0]) == bits(w2)) | (np.isnan(lg[:, 0]) & np.isnan(w2))
        assert same.all()
        # stage by stage over row ranges on device buffers
        dev = torch.device("cuda:0")
        x = torch.from_numpy(g.x()).to(dev)
        h1 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        h2 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        sc = torch.zeros(g.n, dtype=torch.float32, device=dev)
        lgd = torch.zeros(g.n, dtype=torch.float32, device=dev)
        cuts = sorted({0, min(1, g.n), min(64, g.n), g.n // 3, (2 * g.n) // 3 + 1 if g.n > 3 else g.n, g.n})
        for st, (src, dst) in enumerate(((x, h1), (h1, h2), (h2, sc))):
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                e.stage_forward_device(st, lo, hi, src.data_ptr(), dst.data_ptr(), lgd.data_ptr() if st == 2 else 0)
        e.synchronize()
        assert np.array_equal(bits(lgd.cpu().numpy()), bits(want))
        assert float(h1[-1].abs().sum()) == 0.0 and float(h2[-1].abs().sum()) == 0.0   # pad rows untouched
        e.set_option("wide_tiles", 0)
        e.upload_graph(g)
        _, lg = e.forward(g.x())
        assert np.array_equal(bits(lg[:, 0]), bits(want))
        assert e.get_info("wide_tiles_used") == 0
    finally:
        e.close()


@pytest.mark.parametrize("maker,fits", [
    (lambda: gg.erdos_renyi(100000, 1000000, 1), True),        # BASELINE configs[1]: four live columns
    (lambda: gg.erdos_renyi(60000, 900000, 81), True),         # 30 entries a row
    (lambda: gg.erdos_renyi(65537, 700000, 82), True),         # a ragged last tile
    (lambda: gg.erdos_renyi(80000, 240000, 83), False),        # 6 entries a row: more live columns than a table holds
])
def test_table_tiles_are_bit_identical(model_text, oracle_model, maker, fits):
    """Round 4: on graphs of 50 - 400 K vertices the 16-wide stages of a whole forward gather their neighbours
    from the 16-byte-per-vertex compact table of the input (k_stage_t4: the table sits in every L2, the 64-byte rows do not),
    written by the kernel that produced the input for the columns the previous forward chose; the device decides per forward
    and stage whether the table in place fits (else the gathering kernel behind it runs).  Forward after forward — the first
    ones, which cannot fit yet, included — the logits are the oracle's; an input with other live columns refits or steps
    aside; a graph that never fits stops being offered the tiles."""
    import gnn_mwvc_amd as G
    g = maker()
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)
    e = G.Engine(model_text, device=0)
    try:
        e.set_weight_scale(g.ws)
        e.upload_graph(g)
        assert e.get_info("table_tiles_active") == 1 and e.get_info("compact_gather_active") == 0
        seen = []
        for rep in range(8):
            _, lg = e.forward(g.x())
            assert np.array_equal(bits(lg[:, 0]), bits(want)), rep
            seen.append((e.get_info("table_tiles_fit_stage1"), e.get_info("table_tiles_fit_stage2")))
        if fits:
            assert seen[0] == (0, 0)                        # nobody has chosen columns yet
            assert seen[-1] == (1, 1) and seen[3] == (1, 1), seen
            # another input on the same graph: whatever it makes live, the bits hold (and the tiles come back for the old one)
            rng = np.random.default_rng(9)
            for x2 in ((g.x() * np.float32(0.25)).astype(np.float32), rng.uniform(0.0, 2.0, g.n).astype(np.float32)):
                w2 = oracle_model.predict(g, x2, stop_after=oracle_model.n_layers - 2)[:, 0]
                for rep in range(4):
                    _, lg = e.forward(x2)
                    assert np.array_equal(bits(lg[:, 0]), bits(w2)), rep
            for rep in range(4):
                _, lg = e.forward(g.x())
                assert np.array_equal(bits(lg[:, 0]), bits(want)), rep
            # values that cannot lend the table their sign bit (NaN: every column of such a vertex's rows) and infinities: the
            # vertices around them take their neighbours from the full rows, everything else stays on the table
            x3 = g.x().copy()
            x3[5] = np.nan; x3[77] = np.inf; x3[1234] = -np.inf; x3[4321] = -3.0; x3[999] = 3.0e38; x3[g.n - 1] = np.nan
            w3 = oracle_model.predict(g, x3, stop_after=oracle_model.n_layers - 2)[:, 0]
            assert np.isnan(w3).sum() > 100
            for rep in range(3):
                _, lg = e.forward(x3)
                got = lg[:, 0]
                same = (bits(got) == bits(w3)) | (np.isnan(got) & np.isnan(w3))
                assert same.all(), (rep, int((~same).sum()))
            for rep in range(3):
                _, lg = e.forward(g.x())
                assert np.array_equal(bits(lg[:, 0]), bits(want)), rep
            # (the noise input missed four times in a row: the tiles may have been switched off for this graph by now)
            assert e.get_info("table_tiles_active") == 0 or e.get_info("table_tiles_fit_stage1") == 1
            # another graph on the same engine: its FIRST forward already takes the tiles wherever the last forward with tiles on
            # the previous graph left a choice (here: whatever the inputs above left behind — any choice gives the same bits)
            g2 = gg.erdos_renyi(70000, 800000, 91)
            oracle_model.set_weight_scale(g2.ws)
            want2 = oracle_model.logits(g2)
            e.set_weight_scale(g2.ws)
            e.upload_graph(g2)
            assert e.get_info("table_tiles_active") == 1
            first = None
            for rep in range(4):
                _, lg = e.forward(g2.x())
                assert np.array_equal(bits(lg[:, 0]), bits(want2)), rep
                fit = (e.get_info("table_tiles_fit_stage1"), e.get_info("table_tiles_fit_stage2"))
                first = fit if first is None else first
            assert fit == (1, 1) and first != (0, 0), (first, fit)
            # ... and a third graph straight after: both stages of its first forward
            g3 = gg.erdos_renyi(90000, 1100000, 92)
            oracle_model.set_weight_scale(g3.ws)
            e.set_weight_scale(g3.ws)
            e.upload_graph(g3)
            _, lg = e.forward(g3.x())
            assert np.array_equal(bits(lg[:, 0]), bits(oracle_model.logits(g3)))
            assert (e.get_info("table_tiles_fit_stage1"), e.get_info("table_tiles_fit_stage2")) == (1, 1)
            oracle_model.set_weight_scale(g.ws)
            e.set_weight_scale(g.ws)
            e.upload_graph(g)
            # the same with the gathering kernel always launched behind the tiles (no solo mode)
            e.set_option("table_tiles_solo", 0)
            e.upload_graph(g)
            for rep in range(5):
                _, lg = e.forward(g.x())
                assert np.array_equal(bits(lg[:, 0]), bits(want)), rep
            assert e.get_info("table_tiles_fit_stage1") == 1 and e.get_info("table_tiles_fit_stage2") == 1
            e.set_option("table_tiles_solo", 1)
        else:
            assert all(f == (0, 0) for f in seen)
            assert e.get_info("table_tiles_active") == 0     # four misses in a row: no longer offered
        # stage calls outside a forward never take the tiles (nothing of theirs may be read or written there)
        e.set_option("table_tiles", 0)
        e.upload_graph(g)
        assert e.get_info("table_tiles_active") == 0
        _, lg = e.forward(g.x())
        assert np.array_equal(bits(lg[:, 0]), bits(want))
    finally:
        e.close()


@pytest.mark.parametrize("chunk_rows,overlap,maker", [
    (16, 1, lambda: gg.erdos_renyi(20000, 200000, 72)),     # one row per wave slice: 1250 chunks, five rounds of the grid
    (16, 0, lambda: gg.erdos_renyi(20000, 200000, 72)),     # the same with the last stage's dense layers after, not under, the sums
    (48, 1, lambda: gg.erdos_renyi(20011, 150000, 73)),     # three rows per slice, a ragged last chunk
    (32, 1, lambda: _dense_graph(3000, 300, 74)),           # 300-entry rows: runs longer than a step in a 2-row slice
])
def test_plans_with_many_small_chunks(model_text, oracle_model, chunk_rows, overlap, maker):
    """Both per-graph plans with the chunk size capped ("plan_chunk_rows"): several rounds of the persistent grids,
    a partial last round, slices of one to three rows — what only the 10 M-vertex graph exercises otherwise."""
    import torch
    import gnn_mwvc_amd as G
    g = maker()
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_option("plan_chunk_rows", chunk_rows)
        e.set_option("overlap_dense", overlap)
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        e.upload_graph(g)
        e.forward(g.x())
        _, logits = e.forward(g.x())
        _, again = e.forward(g.x())
        assert e.get_info("lds_table_active") == 1 and e.get_info("compact_gather_active") == 1
        rows = e.get_info("compact_gather_rows_per_chunk")          # (at least 256 chunks: small graphs get smaller ones still)
        assert 16 <= rows <= chunk_rows and e.get_info("compact_gather_chunks") == (g.n + rows - 1) // rows
        want = oracle_model.logits(g)
        assert np.array_equal(bits(logits[:, 0]), bits(want)) and np.array_equal(bits(again[:, 0]), bits(want))
        # the 16-wide stages on an input the compact table certainly takes (the model's own activations may not)
        rng = np.random.default_rng(chunk_rows)
        h = _sparse_features(g.n, rng, [1, 6, 10, 15], [1.0, 0.4, 0.2, 1.0], strays=min(7, g.n // 1024))   # (more than n / 512 stray values and the plan steps aside)
        dev = torch.device("cuda:0")
        hin = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        hin[: g.n] = torch.from_numpy(h).to(dev)
        out = torch.full((g.n + 1, 16), 7.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        e.stage_forward_device(1, 0, g.n, hin.data_ptr(), out.data_ptr())
        e.synchronize()
        assert e.get_info("compact_gather_last_ok") == 1 and e.get_info("compact_gather_last_dirty") > 0
        assert np.array_equal(bits(out[: g.n].cpu().numpy()), bits(_oracle_stage(oracle_model, g, 1, h)))
    finally:
        e.close()


@pytest.mark.parametrize("world,pieces", [(2, 3), (2, 1), (3, 2), (4, 2)])
def test_compact_gather_over_a_ranks_rows(model_text, oracle_model, world, pieces):
    """gnnvc_stage_input_ready: the plan chunked over the rows ONE rank of a P-rank run computes, the table written
    once per stage, the rank's rows computed in pieces — same bits as the oracle for every rank's range."""
    import torch
    import gnn_mwvc_amd as G
    from gnn_mwvc_amd import distributed as D
    g = gg.erdos_renyi(120000, 1200000, 73)   # (enough chunks per piece for the pieces to take the plan)
    rng = np.random.default_rng(world * 10 + pieces)
    h = _sparse_features(g.n, rng, [0, 3, 7, 11], [1.0, 0.1, 0.5, 1.0], strays=12)
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        e.upload_graph(g)
        dev = torch.device("cuda:0")
        hin = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        hin[: g.n] = torch.from_numpy(h).to(dev)
        bounds = D.partition_bounds(g.n, world)
        for stage in (1, 2):
            want = _oracle_stage(oracle_model, g, stage, h)
            for lo, hi in bounds:
                out = torch.full((g.n + 1, 16) if stage == 1 else (g.n + 1,), 7.0, dtype=torch.float32, device=dev)
                lg = torch.full((g.n + 1,), 7.0, dtype=torch.float32, device=dev)
                torch.cuda.synchronize()
                e.stage_input_ready(stage, hin.data_ptr(), lo, hi)
                assert e.get_info("compact_gather_active") == 1
                step = ((hi - lo + pieces - 1) // pieces + 63) // 64 * 64
                for r0 in range(lo, hi, step):
                    e.stage_forward_device(stage, r0, min(r0 + step, hi), hin.data_ptr(), out.data_ptr(),
                                           lg.data_ptr() if stage == 2 else 0)
                e.synchronize()
                got = out[lo:hi].cpu().numpy() if stage == 1 else lg[lo:hi].cpu().numpy().reshape(-1, 1)
                assert np.array_equal(bits(got), bits(want[lo:hi])), (stage, lo, hi)
                assert e.get_info("compact_gather_last_ok") == 1
        # a whole forward afterwards still gives the reference's logits (the range plan is simply not used for it)
        _, logits = e.forward(g.x())
        assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g)))
    finally:
        e.close()


def test_compact_gather_producer_side_statistics(model_text, oracle_model):
    """From the third forward on a graph, the stage kernels themselves count the non-zeros of the rows they write and
    write their compact form (c4_emit), and the next stage skips its two passes over its input.  Same bits, forward
    after forward, also when the input — and with it the set of live columns — changes in between."""
    import gnn_mwvc_amd as G
    g = gg.erdos_renyi(30000, 300000, 72)
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        e.upload_graph(g)
        want = oracle_model.logits(g)
        rng = np.random.default_rng(9)
        x2 = (g.x() * rng.uniform(0.2, 3.0, g.n).astype(np.float32)).astype(np.float32)   # other features, other live columns
        want2 = oracle_model.predict(g, x2, stop_after=oracle_model.n_layers - 2)[:, 0]
        seen_ok = 0
        for i, (x, w) in enumerate([(g.x(), want)] * 4 + [(x2, want2)] * 3 + [(g.x(), want)] * 2):
            _, logits = e.forward(x)
            assert np.array_equal(bits(logits[:, 0]), bits(w)), i
            if i >= 1:
                assert e.get_info("compact_gather_active") == 1
                seen_ok += e.get_info("compact_gather_last_ok")
        assert seen_ok > 0       # the plan really ran on some of these forwards
    finally:
        e.close()


# ---------------------------------------------------------------- long-row path

@pytest.mark.parametrize("thresh,block_cols", [(8, 0), (64, 0), (300, 0), (0, 0), (16, 512), (1000, 2048)])
def test_long_row_path_is_bit_identical(model_text, oracle_model, thresh, block_cols):
    """Rows of degree >= threshold are summed by a workgroup of their own (k_long_*), on a
    second stream beside the tile kernel — same CSR-order add chain, so same bits; also in
    combination with the column-blocked stage 0."""
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("long_row_threshold", thresh)
        e.set_option("sorted_tiles", 0)   # (sorted tiles raise the threshold; covered by their own test)
        if block_cols:
            e.set_option("blocked_min_n", 0)
            e.set_option("block_cols", block_cols)
            e.set_option("blocked_stage0", 2)
        graphs = [gg.hub_graph(20000, 60000, 3, 4096, seed=7), gg.rmat(11, 16, 5),
                  gg.erdos_renyi(3000, 30000, 41),
                  gg.from_edge_list(700, [(0, i) for i in range(1, 700)] + [(1, i) for i in range(2, 300)],
                                    [20 + (i % 101) for i in range(700)])]
        for g in graphs:
            e.set_weight_scale(g.ws)
            oracle_model.set_weight_scale(g.ws)
            e.upload_graph(g)
            deg = np.diff(g.rowptr.astype(np.int64))
            want_long = int((deg >= thresh).sum()) if thresh else 0
            assert e.get_info("long_rows") == want_long
            scores, logits = e.forward(g.x())
            assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g)))
            assert ulp(scores[:, 0], oracle_model.scores(g)).max() <= 1
            # a vertex sub-range through the stage entry point (partitioned execution)
            import torch
            dev = torch.device("cuda:0")
            x = torch.from_numpy(g.x()).to(dev)
            h1 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
            h2 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            mid = (g.n // 3) // 64 * 64
            for lo, hi in ((0, mid), (mid, g.n)):
                e.stage_forward_device(0, lo, hi, x.data_ptr(), h1.data_ptr())
            for lo, hi in ((0, mid), (mid, g.n)):
                e.stage_forward_device(1, lo, hi, h1.data_ptr(), h2.data_ptr())
            e.synchronize()
            assert np.array_equal(bits(h2[:-1].cpu().numpy()), bits(oracle_model.predict(g, g.x(), stop_after=13)))
    finally:
        e.close()


# ---------------------------------------------------------------- giant rows: the chain's bits from a parallel scan

def _seq_sum(v):
    return np.cumsum(np.ascontiguousarray(v, dtype=np.float32), dtype=np.float32)[-1] if len(v) else np.float32(0)


def test_stream_sum_entry_point_is_the_sequential_chain(engine):
    """gnnvc_stream_sum (k_giant_sum on caller data) against acc = acc + v[i] (reference src/gnn_inference.cpp:33-36)
    on streams chosen to hit every branch of csrc/exact_sum.h: ties, parity flips, carries through many binades,
    denormals, huge jumps, overflow, -0.0, negative and non-finite values."""
    rng = np.random.default_rng(7)
    n = 70_000
    streams = []
    v = rng.gamma(2.0, 0.7, n).astype(np.float32); v[rng.random(n) < 0.4] = 0.0; streams.append(v)
    streams.append((rng.integers(20, 121, n) / np.float32(120.0)).astype(np.float32))
    t = (rng.integers(0, 64, n) * np.float32(1 / 16)).astype(np.float32); t[0] = 4096.0; streams.append(t)
    streams.append(np.full(n, 1.0, dtype=np.float32))
    streams.append(np.full(n, 0.1, dtype=np.float32))
    a = np.tile(np.array([1.5, 0.5, 2.5, 1.0], dtype=np.float32) * np.float32(2.0 ** -10), n // 4); a[0] = 8191.0; streams.append(a)
    streams.append((rng.random(n).astype(np.float32) * np.exp2(rng.integers(-149, 20, n)).astype(np.float32)).astype(np.float32))
    streams.append(rng.integers(0, 1 << 20, n).astype(np.uint32).view(np.float32))          # denormals only
    streams.append(rng.integers(0, 1 << 24, n).astype(np.uint32).view(np.float32))          # across the denormal boundary
    j = rng.uniform(0, 1, n).astype(np.float32); j[n // 2] = 3.0e30; streams.append(j)
    b = rng.uniform(0, 1, n).astype(np.float32); b[1000:11000] = 3.0e38; streams.append(b)  # overflows to +inf
    z = rng.uniform(0, 2, n).astype(np.float32); z[::997] = -0.0; streams.append(z)
    g1 = z.copy(); g1[1234] = -5.0; g1[55_555] = -1e-3; streams.append(g1)                  # two negative values
    streams.append(rng.normal(0, 1, n).astype(np.float32))                                  # half negative
    q = z.copy(); q[40_000] = np.nan; streams.append(q)
    i2 = z.copy(); i2[10] = np.inf; i2[20_000] = -np.inf; streams.append(i2)
    streams.append(np.zeros(n, dtype=np.float32))
    bitsr = rng.integers(0, 1 << 31, n).astype(np.uint32); bitsr[(bitsr >> 23) == 255] &= 0x7F000000
    streams.append(bitsr.view(np.float32))
    V = np.stack(streams)
    with np.errstate(all="ignore"):
        want = np.array([_seq_sum(r) for r in V], dtype=np.float32)
    for mode in (0, 2):     # a stream on several waves (segments of 4096 addends with their own parity maps) / on one wave
        got = engine.stream_sum(V, mode=mode)
        same = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), f"mode {mode}: streams {np.nonzero(~same)[0].tolist()} differ: {got[~same]} vs {want[~same]}"
    # ragged lengths around the window size (1024 addends), the segment size (4096) and the lane size (16)
    for ln in (1, 15, 16, 17, 1023, 1024, 1025, 2048, 4095, 4096, 4097, 5000, 8192, 8193, 12289, 300_000):
        r = rng.gamma(2.0, 0.7, (3, ln)).astype(np.float32)
        ws = bits(np.array([_seq_sum(x) for x in r], dtype=np.float32))
        assert np.array_equal(bits(engine.stream_sum(r)), ws), ln
        assert np.array_equal(bits(engine.stream_sum(r, mode=2)), ws), ln
    assert engine.stream_sum(np.zeros((2, 0), dtype=np.float32)).tolist() == [0.0, 0.0]
    # many segments per stream (their maps are fetched 64 at a time): lengths around 64 and 128 segments
    for ln in (64 * 4096 - 1, 64 * 4096 + 1, 65 * 4096, 128 * 4096 + 7, 200_943):
        r = rng.gamma(2.0, 0.7, (2, ln)).astype(np.float32)
        ws = bits(np.array([_seq_sum(x) for x in r], dtype=np.float32))
        assert np.array_equal(bits(engine.stream_sum(r)), ws), ln


@pytest.mark.parametrize("giant,long_t", [(64, 8), (300, 64), (1000, 512), (5000, 512)])
def test_giant_row_path_is_bit_identical(model_text, oracle_model, giant, long_t):
    """Rows of degree >= "giant_row_threshold" are gathered into per-column streams and summed by k_giant_sum; the
    rest of the long rows stay with k_long_*.  Same bits as the chain, whole forwards and vertex sub-ranges."""
    import gnn_mwvc_amd as G
    import torch
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("long_row_threshold", long_t)
        e.set_option("giant_row_threshold", giant)
        e.set_option("sorted_tiles", 0)
        graphs = [gg.hub_graph(20000, 60000, 3, 4096, seed=7), gg.rmat(11, 16, 5),
                  gg.hub_graph(30000, 40000, 2, 20000, seed=8),
                  gg.from_edge_list(700, [(0, i) for i in range(1, 700)] + [(1, i) for i in range(2, 300)],
                                    [20 + (i % 101) for i in range(700)])]
        for g, segments in [(g, s) for g in graphs for s in (0, 1)]:   # one wave per stream / a stream on several waves
            e.set_option("giant_segments", segments)
            e.set_weight_scale(g.ws)
            oracle_model.set_weight_scale(g.ws)
            e.upload_graph(g)
            deg = np.diff(g.rowptr.astype(np.int64))
            assert e.get_info("giant_rows") == int((deg >= max(giant, long_t)).sum())
            if e.get_info("giant_rows"):
                assert (e.get_info("giant_segments") > 1) == (segments == 1 and deg.max() > 4096)
            assert e.get_info("giant_entries") == int(deg[deg >= max(giant, long_t)].sum())
            scores, logits = e.forward(g.x())
            assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g)))
            assert ulp(scores[:, 0], oracle_model.scores(g)).max() <= 1
            dev = torch.device("cuda:0")
            x = torch.from_numpy(g.x()).to(dev)
            h1 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
            h2 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            mid = (g.n // 3) // 64 * 64
            for lo, hi in ((0, mid), (mid, g.n)):
                e.stage_forward_device(0, lo, hi, x.data_ptr(), h1.data_ptr())
            for lo, hi in ((0, mid), (mid, g.n)):
                e.stage_forward_device(1, lo, hi, h1.data_ptr(), h2.data_ptr())
            e.synchronize()
            assert np.array_equal(bits(h2[:-1].cpu().numpy()), bits(oracle_model.predict(g, g.x(), stop_after=13)))
    finally:
        e.close()


def test_giant_rows_with_arbitrary_features(model_text, oracle_model):
    """Stage entry point with inputs the model itself never produces (negative values, -0.0): the giant-row sums
    still are the chain's."""
    import gnn_mwvc_amd as G
    import torch
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("long_row_threshold", 64)
        e.set_option("giant_row_threshold", 1000)
        g = gg.hub_graph(20000, 60000, 3, 4096, seed=7)
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        e.upload_graph(g)
        rng = np.random.default_rng(3)
        for kind in ("mixed_sign", "few_negative"):
            h = rng.gamma(2.0, 0.5, (g.n, 16)).astype(np.float32)
            h[rng.random((g.n, 16)) < 0.5] = 0.0
            if kind == "mixed_sign":
                h *= rng.choice(np.array([-1.0, 1.0], dtype=np.float32), size=h.shape)
            else:
                h[rng.integers(0, g.n, 40), rng.integers(0, 16, 40)] = -3.0
                h[rng.integers(0, g.n, 40), rng.integers(0, 16, 40)] = -0.0
            dev = torch.device("cuda:0")
            hin = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
            hin[:-1] = torch.from_numpy(h).to(dev)
            hout = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            e.stage_forward_device(1, 0, g.n, hin.data_ptr(), hout.data_ptr())
            e.synchronize()
            want = _oracle_stage(oracle_model, g, 1, h)
            assert np.array_equal(bits(hout[:-1].cpu().numpy()), bits(want)), kind
    finally:
        e.close()


def test_the_tolerance_mode_is_gone(model_text):
    """Rounds 1 - 3 carried SURVEY.md §7's tolerance mode ("hub_mode" 1: tree sums for long rows, a few ulp off).  With the
    exact parallel chain as fast as it is the mode bought nothing measurable; round 4 removed it: the option and the
    layer-level mode 1 are refused, there is exactly one way a long row is summed."""
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    try:
        with pytest.raises(G.GnnvcError):
            e.set_option("hub_mode", 1)
        with pytest.raises(G.GnnvcError):
            e.stream_sum(np.ones((2, 100), dtype=np.float32), mode=1)
    finally:
        e.close()


# ---------------------------------------------------------------- one rank's CSR slice

@pytest.mark.parametrize("maker,world,mode,reps", [
    (lambda: gg.erdos_renyi(5000, 40000, 31), 2, "rows", 1),
    (lambda: gg.hub_graph(20000, 60000, 3, 4096, seed=7), 3, "nnz", 1),       # long rows inside slices
    (lambda: gg.hub_graph(30000, 40000, 2, 20000, seed=8), 4, "nnz", 1),      # giant rows inside slices
    (lambda: gg.rmat(11, 16, 5), 3, "rows", 1),
    (lambda: gg.erdos_renyi(100, 300, 5), 3, "rows", 1),                      # a short slice and an empty one
    (lambda: gg.rmat(14, 8, 3), 3, "nnz", 3),                                 # repeated forwards: each slice prunes its adjacency
    (lambda: gg.chung_lu_hubs(40000, 8.0, 2.2, 2, 3000, seed=4), 2, "rows", 3),
])
def test_sliced_engines_equal_the_whole_graph(model_text, oracle_model, maker, world, mode, reps):
    """gnnvc_attach_graph_slice: `world` engines, each holding only its rows' CSR slice (row pointers relative to
    the slice, global column ids), driven stage by stage on shared full-size feature buffers — what the ranks of
    a vertex-partitioned run do between exchanges.  Same bits as the whole graph."""
    import gnn_mwvc_amd as G
    import torch
    from gnn_mwvc_amd import distributed as D
    g = maker()
    dev = torch.device("cuda:0")
    oracle_model.set_weight_scale(g.ws)
    bounds = D.partition_bounds(g.n, world, g.rowptr, mode)
    rp = torch.from_numpy(g.rowptr.astype(np.int64)).to(torch.int32).to(dev)   # (uint32 values fit: small graphs)
    col = torch.from_numpy(g.col.astype(np.int64)).to(torch.int32).to(dev)
    w = torch.from_numpy(g.w.astype(np.int64)).to(torch.int32).to(dev)
    nw = torch.from_numpy(g.nw.astype(np.int64)).to(torch.int32).to(dev)
    engines = []
    try:
        total = 0
        for lo, hi in bounds:
            e = G.Engine(model_text, device=0)
            engines.append(e)
            e.set_option("long_row_threshold", 64)
            e.set_option("giant_row_threshold", 3000)
            if reps > 1:
                e.set_option("prune_min_entries", 0)
                e.set_option("prune_min_drop_percent", 1)
            e.set_weight_scale(g.ws)
            sl = D.slice_csr(g.n, rp, col, w, nw, lo, hi)
            torch.cuda.synchronize()
            e.attach_graph_slice(g.n, lo, hi, sl.nnz, sl.rowptr.data_ptr(), sl.col.data_ptr(), sl.w.data_ptr(),
                                 sl.nw.data_ptr(), keepalive=sl)
            assert e.get_info("slice_rows") == hi - lo and e.get_info("slice_entries") == sl.nnz
            total += sl.nbytes()
        assert total <= (rp.numel() + col.numel() + w.numel() + nw.numel()) * 4 + world * (64 + 1) * 4   # slices partition the CSR
        x = torch.from_numpy(g.x()).to(dev)
        h1 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        h2 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        sc = torch.zeros(g.n, dtype=torch.float32, device=dev)
        lg = torch.zeros(g.n, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        for rep in range(reps):
            if rep:
                for t in (h1, h2, sc, lg):
                    t.fill_(7.0)
                h1[g.n] = 0.0
                h2[g.n] = 0.0
                torch.cuda.synchronize()
            for st, (src, dst, lgt) in enumerate(((x, h1, None), (h1, h2, None), (h2, sc, lg))):
                for e, (lo, hi) in zip(engines, bounds):
                    mid = lo + ((hi - lo) // 2) // 64 * 64
                    for r0, r1 in ((lo, mid), (mid, hi)):       # in two pieces, like a pipelined rank
                        e.stage_forward_device(st, r0, r1, src.data_ptr(), dst.data_ptr(), lgt.data_ptr() if lgt is not None else 0)
                for e in engines:
                    e.synchronize()                              # ("exchange": the buffers are shared here)
            assert np.array_equal(bits(lg.cpu().numpy()), bits(oracle_model.logits(g))), rep
            assert ulp(sc.cpu().numpy(), oracle_model.scores(g)).max() <= 1
        if reps > 1:   # every slice with entries found zero rows among its neighbours and used its pruned adjacency to the end
            pruned = [(e.get_info("pruned_stage1"), e.get_info("pruned_stage2")) for e in engines]
            assert all(any(p) for p in pruned), pruned
            for e, p in zip(engines, pruned):
                for st in (1, 2):
                    if p[st - 1]:
                        assert e.get_info(f"pruned_last_ok_stage{st}") == 1
                        assert e.get_info(f"pruned_entries_stage{st}") < e.get_info("slice_entries")
        # outside its slice an engine has no adjacency, and says so
        e0, (lo0, hi0) = engines[0], bounds[0]
        if hi0 < g.n:
            with pytest.raises(G.GnnvcError):
                e0.stage_forward_device(1, hi0, g.n, h1.data_ptr(), h2.data_ptr())
        with pytest.raises(G.GnnvcError):
            e0.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    finally:
        for e in engines:
            e.close()


# ---------------------------------------------------------------- several devices behind one handle (gnnvc_create_multi)

@pytest.mark.parametrize("maker,devices", [
    (lambda: gg.erdos_renyi(5000, 40000, 31), [0, 0]),
    (lambda: gg.hub_graph(20000, 60000, 3, 4096, seed=7), [0, 0, 0]),          # long rows inside parts
    (lambda: gg.rmat(13, 16, 5), [0, 0, 0, 0]),                                 # skewed: nnz-balanced cuts, sorted tiles per part
    (lambda: gg.erdos_renyi(100, 300, 5), [0] * 8),                             # more parts than 64-row tiles: empty parts
    (lambda: gg.erdos_renyi(3000, 20000, 9), [0]),                              # one device behind the same code path
])
def test_multi_device_handle_equals_single_engine(model_text, oracle_model, maker, devices):
    """gnnvc_create_multi through the C ABI: upload / staged hand-off partition internally, forward exchanges rows device
    to device (all parts on GPU 0 here), host and device-pointer forwards, repeated forwards, graph replacement, N = 0.
    Logits bit-identical to the oracle — what a single engine gives."""
    import gnn_mwvc_amd as G
    import torch
    g = maker()
    e = G.Engine(model_text, devices=devices)
    try:
        assert e.get_info("devices") == len(devices)
        for route in ("upload", "staged"):
            e.set_weight_scale(g.ws)
            oracle_model.set_weight_scale(g.ws)
            (e.upload_graph if route == "upload" else e.upload_graph_staged)(g)
            rows = [e.get_info(f"part_rows_{r}") for r in range(len(devices))]
            entries = [e.get_info(f"part_entries_{r}") for r in range(len(devices))]
            assert sum(rows) == g.n and sum(entries) == g.nnz
            if len(devices) > 1 and g.nnz > 64 * 64 * len(devices):            # the cuts balance the entries
                assert max(entries) < 2.0 * g.nnz / len(devices) + 64 * (g.nnz // g.n + 4096)
            want = oracle_model.logits(g)
            for rep in range(3):
                scores, logits = e.forward(g.x())
                assert np.array_equal(bits(logits[:, 0]), bits(want)), (route, rep)
                assert ulp(scores[:, 0], oracle_model.scores(g)).max() <= 1
            keys, above = e.score_keys()
            ok, oa = oracle_py.score_keys(scores[:, 0])
            assert np.array_equal(bits(keys), bits(ok)) and np.array_equal(above, oa)
        # device-pointer forward (pointers on the first device)
        dev = torch.device("cuda:0")
        x = torch.from_numpy(g.x()).to(dev)
        sc = torch.zeros(g.n, device=dev)
        lg = torch.zeros(g.n, device=dev)
        torch.cuda.synchronize()
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
        assert np.array_equal(bits(lg.cpu().numpy()), bits(want))
        # another graph, then the empty one (the reference's last predict call)
        g2 = gg.erdos_renyi(777, 4000, 3)
        e.set_weight_scale(g2.ws)
        oracle_model.set_weight_scale(g2.ws)
        e.upload_graph(g2)
        _, logits = e.forward(g2.x())
        assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g2)))
        e.upload_graph(gg.from_edge_list(0, [], []))
        s0, _ = e.forward(np.zeros(0, dtype=np.float32))
        assert s0.shape == (0, 1)
        # what reads ONE device's resident graph is refused, loudly
        with pytest.raises(G.GnnvcError) as err:
            e.reduction_flags()
        assert err.value.code == -5
        with pytest.raises(G.GnnvcError):
            e.stage_forward_device(0, 0, 1, x.data_ptr(), sc.data_ptr())
        # malformed graphs are refused by the parts' device-side checks
        bad = gg.erdos_renyi(500, 2000, 2)
        bad.col[7] = 500
        with pytest.raises(G.GnnvcError):
            e.upload_graph(bad)
    finally:
        e.close()


@pytest.mark.parametrize("maker,parts", [
    (lambda: gg.erdos_renyi(60000, 1200000, 37), 8),                  # 40 entries a row: few live columns, the rows travel packed
    (lambda: gg.erdos_renyi(30000, 90000, 38), 5),                    # sparse: many live columns
    (lambda: gg.rmat(14, 16, 9), 6),                                  # skewed, uneven parts
])
def test_multi_handle_exchange_options(model_text, oracle_model, maker, parts):
    """Round 4: the exchange behind gnnvc_create_multi — pieces packed to their live columns, pushed to every peer by one kernel,
    expanded per piece index — in every configuration ("multi_pieces" 1 .. 8, "multi_pack" 0 / 1, "multi_push" 0 / 1): the
    first forward on a graph (which chooses the packing from every part's counts) and the ones after it equal the oracle bit
    for bit, packed rows ship fewer bytes than full ones, and an input that makes other columns live than the ones the packing
    was chosen for (exception lists overflow) is repeated with full rows instead of delivering a wrong row."""
    import gnn_mwvc_amd as G
    g = maker()
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)
    e = G.Engine(model_text, devices=[0] * parts)
    try:
        e.set_weight_scale(g.ws)
        full = None
        for pieces, pack, push in ((0, 1, 1), (1, 1, 1), (3, 1, 0), (8, 1, 1), (2, 0, 1), (4, 0, 0)):
            e.set_option("multi_pieces", pieces)
            e.set_option("multi_pack", pack)
            e.set_option("multi_push", push)
            e.upload_graph(g)
            for rep in range(3):
                _, lg = e.forward(g.x())
                assert np.array_equal(bits(lg[:, 0]), bits(want)), (pieces, pack, push, rep)
            assert e.get_info("multi_pieces") == (pieces if pieces else (1 if parts <= 4 else 4))
            shipped = [e.get_info(f"multi_exchange_bytes_per_peer_stage{s}") for s in (0, 1)]
            if not pack:
                assert e.get_info("multi_packed_stage0") == 0 and e.get_info("multi_packed_stage1") == 0
                full = shipped
            elif full is not None:
                for s in (0, 1):
                    if e.get_info(f"multi_packed_stage{s}"):
                        assert shipped[s] < full[s]
        # another input on the same graph: the packing was chosen for the driver's input; whatever this one makes live travels
        # in the lists or, when they overflow, as full rows in a repeated forward — never as a wrong row
        e.set_option("multi_pieces", 0)
        e.set_option("multi_pack", 1)
        e.set_option("multi_push", 1)
        e.upload_graph(g)
        _, lg = e.forward(g.x())
        assert np.array_equal(bits(lg[:, 0]), bits(want))
        rng = np.random.default_rng(3)
        x2 = rng.uniform(0.0, 3.0, g.n).astype(np.float32)
        want2 = oracle_model.predict(g, x2, stop_after=oracle_model.n_layers - 2)[:, 0]
        for rep in range(2):
            _, lg = e.forward(x2)
            assert np.array_equal(bits(lg[:, 0]), bits(want2)), rep
        _, lg = e.forward(g.x())
        assert np.array_equal(bits(lg[:, 0]), bits(want))
        # a part's share alone (the timing rehearsal): runs, reports its span, and the next complete forward is right again
        e.set_option("multi_only_part", 0)
        e.forward(g.x())
        assert e.get_info("multi_part_span_us_0") > 0
        e.set_option("multi_only_part", -1)
        _, lg = e.forward(g.x())
        assert np.array_equal(bits(lg[:, 0]), bits(want))
        with pytest.raises(G.GnnvcError):
            e.set_option("multi_no_such_option", 1)
    finally:
        e.close()
    single = G.Engine(model_text, device=0)
    try:
        with pytest.raises(G.GnnvcError):          # the exchange's options belong to multi-device handles
            single.set_option("multi_pieces", 2)
    finally:
        single.close()


def test_push_and_unpack_pieces_entry_points(model_text):
    """gnnvc_push_piece / gnnvc_unpack_pieces against numpy: packed pieces of a random sparse 16-column matrix pushed into
    several destination regions by one launch each, the regions of several "peers" expanded by one launch, -0.0 and
    exceptions included; a short list (overflow) raises gnnvc_pack_rows' flag as before."""
    import ctypes as C
    import torch
    import gnn_mwvc_amd as G
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    n, kp, cap = 5000, 8, 512
    feat = (rng.uniform(0.1, 2.0, (n + 1, 16)) * (rng.random((n + 1, 16)) < 0.3)).astype(np.float32)
    live = [1, 3, 4, 8, 9, 15]
    dead = [c for c in range(16) if c not in live]
    feat[:, dead] = 0.0
    feat[rng.integers(0, n, 40), 2] = 0.75          # strays outside the mask: the exception list's
    feat[17, 3] = -0.0
    feat[n] = 0.0
    mask = sum(1 << c for c in live)
    e = G.Engine(model_text, device=0)
    L = e._L
    L.gnnvc_push_piece.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.gnnvc_unpack_pieces.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    try:
        src = torch.from_numpy(feat).to(dev)
        cuts = [0, 1200, 1200, 3333, n]              # (an empty piece among them)
        words = [(cuts[i + 1] - cuts[i]) * kp + 4 + 4 * cap for i in range(4)]
        send = [torch.zeros(w, dtype=torch.float32, device=dev) for w in words]
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        for i in range(4):
            rows = cuts[i + 1] - cuts[i]
            exc = send[i].data_ptr() + rows * kp * 4
            e._check(L.gnnvc_pack_rows(e._h, src.data_ptr(), 16, cuts[i], cuts[i + 1], mask, kp, send[i].data_ptr(), exc, cap, flag.data_ptr()))
        # every piece into three "peers'" regions with one launch each
        peers = [[torch.full((w,), 7.0, dtype=torch.float32, device=dev) for w in words] for _ in range(3)]
        for i in range(4):
            dst = (C.c_void_p * 3)(*[peers[q][i].data_ptr() for q in range(3)])
            e._check(L.gnnvc_push_piece(e._h, send[i].data_ptr(), cuts[i + 1] - cuts[i], kp, cap, 3, dst, None))
        e.synchronize()
        assert int(flag.item()) == 0

        class Piece(C.Structure):
            _fields_ = [("d_region", C.c_void_p), ("row_lo", C.c_uint32), ("row_hi", C.c_uint32)]
        for q in range(3):
            out = torch.full((n + 1, 16), 5.0, dtype=torch.float32, device=dev)
            arr = (Piece * 4)(*[Piece(peers[q][i].data_ptr(), cuts[i], cuts[i + 1]) for i in range(4)])
            e._check(L.gnnvc_unpack_pieces(e._h, arr, 4, cap, 16, mask, kp, out.data_ptr()))
            e.synchronize()
            got = out.cpu().numpy()
            assert np.array_equal(got[:n] == 0, feat[:n] == 0)
            assert np.array_equal(bits(np.where(got[:n] == 0, np.float32(0), got[:n])), bits(np.where(feat[:n] == 0, np.float32(0), feat[:n])))
            assert np.all(got[n] == 5.0)             # nothing outside the pieces' rows is written
        # a list too short for the strays raises the pack flag
        small = torch.zeros(n * kp + 4 + 4 * 4, dtype=torch.float32, device=dev)
        e._check(L.gnnvc_pack_rows(e._h, src.data_ptr(), 16, 0, n, mask, kp, small.data_ptr(), small.data_ptr() + n * kp * 4, 4, flag.data_ptr()))
        e.synchronize()
        assert int(flag.item()) & 2
        with pytest.raises(G.GnnvcError):
            e._check(L.gnnvc_unpack_pieces(e._h, None, 65, cap, 16, mask, kp, src.data_ptr()))
    finally:
        e.close()


@pytest.mark.parametrize("world", [4, 8])
def test_sliced_engines_run_their_range_plans(model_text, oracle_model, world):
    """Per-rank plans for any P (round 3): every rank's engine holds only its slice and runs each stage over its whole row
    range — stage 0 from the LDS-table plan laid out over the slice (byte table made from x: no vertex weights needed),
    the 16-wide stages from the compact-table plan announced per stage.  Same bits as the oracle; the plans really ran."""
    import gnn_mwvc_amd as G
    import torch
    from gnn_mwvc_amd import distributed as D
    g = gg.erdos_renyi(160000, 3200000, 79)     # 40 entries per row: at most four live feature columns
    dev = torch.device("cuda:0")
    oracle_model.set_weight_scale(g.ws)
    want = oracle_model.logits(g)
    bounds = D.partition_bounds(g.n, world)
    t = lambda a: torch.from_numpy(a.astype(np.int64)).to(torch.int32).to(dev)
    rp, col, w, nw = t(g.rowptr), t(g.col), t(g.w), t(g.nw)
    engines = []
    try:
        for lo, hi in bounds:
            e = G.Engine(model_text, device=0)
            engines.append(e)
            e.set_option("blocked_min_n", 0)          # let the plans apply to a small graph
            e.set_weight_scale(g.ws)
            sl = D.slice_csr(g.n, rp, col, w, nw, lo, hi)
            torch.cuda.synchronize()
            e.attach_graph_slice(g.n, lo, hi, sl.nnz, sl.rowptr.data_ptr(), sl.col.data_ptr(), sl.w.data_ptr(), sl.nw.data_ptr(),
                                 keepalive=sl)
        x = torch.from_numpy(g.x()).to(dev)
        h1 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        h2 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        sc = torch.zeros(g.n, dtype=torch.float32, device=dev)
        lg = torch.zeros(g.n, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        for rep in range(4):
            if rep:
                for buf in (h1, h2, sc, lg):
                    buf.fill_(7.0)
                h1[g.n] = 0.0
                h2[g.n] = 0.0
                torch.cuda.synchronize()
            for st, (src, dst, lgt) in enumerate(((x, h1, None), (h1, h2, None), (h2, sc, lg))):
                for e, (lo, hi) in zip(engines, bounds):
                    if st >= 1:
                        e.stage_input_ready(st, src.data_ptr(), lo, hi)
                    e.stage_forward_device(st, lo, hi, src.data_ptr(), dst.data_ptr(), lgt.data_ptr() if lgt is not None else 0)
                for e in engines:
                    e.synchronize()                              # ("exchange": the buffers are shared here)
            assert np.array_equal(bits(lg.cpu().numpy()), bits(want)), rep
        for e, (lo, hi) in zip(engines, bounds):
            assert e.get_info("lds_table_active") == 1 and e.get_info("lds_table_chunks") >= 16, (lo, hi)
            assert e.get_info("compact_gather_active") == 1 and e.get_info("compact_gather_last_ok") == 1, (lo, hi)
        # a rank whose x is NOT k / ws (k <= 255): the byte table cannot hold it, the plan steps aside, same bits as the oracle
        x2 = (g.x() * np.float32(0.37)).astype(np.float32)
        want2 = oracle_model.predict(g, x2, stop_after=oracle_model.n_layers - 2)[:, 0]
        xt = torch.from_numpy(x2).to(dev)
        torch.cuda.synchronize()
        for st, (src, dst, lgt) in enumerate(((xt, h1, None), (h1, h2, None), (h2, sc, lg))):
            for e, (lo, hi) in zip(engines, bounds):
                e.stage_forward_device(st, lo, hi, src.data_ptr(), dst.data_ptr(), lgt.data_ptr() if lgt is not None else 0)
            for e in engines:
                e.synchronize()
        assert np.array_equal(bits(lg.cpu().numpy()), bits(want2))
    finally:
        for e in engines:
            e.close()


def test_stage_input_announcement_does_not_outlive_its_input(model_text, oracle_model):
    """gnnvc_stage_input_ready is matched on (stage, buffer address, row range); the contents of that buffer change
    from one forward to the next.  A caller that announces in one forward and not in the next (same buffers) must
    not be served the old table: any call for another stage ends the announcement."""
    import gnn_mwvc_amd as G
    import torch
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("blocked_min_n", 0)      # let the compact-table plan apply to a small graph
        e.set_option("compact_gather", 2)
        g = _dense_graph(6000, 24, 71)
        e.set_weight_scale(g.ws)
        oracle_model.set_weight_scale(g.ws)
        e.upload_graph(g)
        dev = torch.device("cuda:0")
        h1 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        h2 = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
        rng = np.random.default_rng(5)
        outs = []
        for rnd in range(2):
            x = rng.integers(20, 121, g.n).astype(np.float32) / np.float32(g.ws)     # a different input every forward
            xt = torch.from_numpy(x).to(dev)
            torch.cuda.synchronize()
            e.stage_forward_device(0, 0, g.n, xt.data_ptr(), h1.data_ptr())
            if rnd == 0:
                e.stage_input_ready(1, h1.data_ptr(), 0, g.n)     # announced in the first forward only
                assert e.get_info("compact_gather_active") == 1   # (the plan really is in force: the test is not vacuous)
            e.stage_forward_device(1, 0, g.n, h1.data_ptr(), h2.data_ptr())
            e.synchronize()
            want = oracle_model.predict(g, x.reshape(-1, 1), stop_after=13)
            assert np.array_equal(bits(h2[:-1].cpu().numpy()), bits(want)), rnd
            outs.append(h2[:-1].cpu().numpy().copy())
        assert not np.array_equal(outs[0], outs[1])
    finally:
        e.close()


# ---------------------------------------------------------------- the next graph derived on the device (f-1)

def _shrunk_graph(g, rng, keep_frac, new_vertices, fan):
    """(g1, old_row): what the reference's driver hands predict next — survivors of g renumbered in order, their
    lists filtered, plus `new_vertices` fold vertices (largest ids) each adjacent to ~fan survivors / earlier new ones."""
    keep = rng.random(g.n) < keep_frac
    new_of = np.full(g.n, -1, dtype=np.int64)
    ns = int(keep.sum())
    new_of[keep] = np.arange(ns)
    rp = g.rowptr.astype(np.int64)
    src = np.repeat(np.arange(g.n), np.diff(rp))
    dst = g.col.astype(np.int64)
    m = keep[src] & keep[dst] & (src < dst)
    a, b = new_of[src[m]], new_of[dst[m]]
    ea, eb = [a], [b]
    for k in range(new_vertices):
        v = ns + k
        nb = np.unique(rng.integers(0, v, size=min(fan, v))) if v else np.zeros(0, dtype=np.int64)
        ea.append(nb)
        eb.append(np.full(nb.size, v, dtype=np.int64))
    a, b = np.concatenate(ea), np.concatenate(eb)
    n1 = ns + new_vertices
    key = np.unique(a * n1 + b)
    w = rng.integers(20, 121, size=n1)
    g1 = gg.csr_from_pairs(n1, key // n1, key % n1, w)
    old_row = np.full(n1, 0xFFFFFFFF, dtype=np.uint32)
    old_row[:ns] = np.nonzero(keep)[0]
    return g1, old_row


def _fnv_rows(g):
    out = np.empty(g.n, dtype=np.uint64)
    rp = g.rowptr.astype(np.int64)
    with np.errstate(over="ignore"):
        for u in range(g.n):
            h = np.uint64(1469598103934665603)
            for c in g.col[rp[u]:rp[u + 1]]:
                h = (h ^ np.uint64(c)) * np.uint64(1099511628211)
            out[u] = h
    return out


def test_graph_derived_on_the_device_equals_an_upload(model_text, oracle_model):
    """gnnvc_derive_graph_begin / _commit: the engine builds the next graph from the CSR it already holds, the caller
    ships the row mapping and the tails only.  The result is the graph an upload would have produced: same row hashes,
    same forward bits; chained twice; hubs (long / giant rows) included."""
    import gnn_mwvc_amd as G
    rng = np.random.default_rng(17)
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("long_row_threshold", 64)
        e.set_option("giant_row_threshold", 1000)
        for g0 in (gg.erdos_renyi(4000, 24000, 5), gg.hub_graph(20000, 60000, 3, 4096, seed=7)):
            e.upload_graph(g0)
            g = g0
            for step, (frac, nv, fan) in enumerate(((0.7, 50, 40), (0.5, 7, 300), (1.0, 0, 0))):
                g1, old_row = _shrunk_graph(g, rng, frac, nv, fan)
                tail = e.derive_graph(g1, old_row)
                ns = int((old_row != 0xFFFFFFFF).sum())
                deg1 = np.diff(g1.rowptr.astype(np.int64))
                # survivors keep only new-vertex ids in their tails; new vertices' lists are all tail
                want_tail = np.array([int((g1.col[int(g1.rowptr[u]):int(g1.rowptr[u + 1])] >= ns).sum()) if u < ns else deg1[u]
                                      for u in range(g1.n)], dtype=np.uint32)
                assert np.array_equal(tail, want_tail), step
                if g1.n <= 5000:
                    assert np.array_equal(e.row_hashes(), _fnv_rows(g1)), step
                e.set_weight_scale(g1.ws)
                oracle_model.set_weight_scale(g1.ws)
                _, logits = e.forward(g1.x())
                assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g1))), step
                g = g1
        # inconsistent mappings are refused and leave the resident graph usable
        g1, old_row = _shrunk_graph(g, rng, 0.6, 3, 10)
        bad = old_row.copy()
        bad[1] = bad[0]
        with pytest.raises(G.GnnvcError):
            e.derive_graph(g1, bad)
        bad = old_row.copy()
        bad[0] = g.n + 7
        with pytest.raises(G.GnnvcError):
            e.derive_graph(g1, bad)
        shuffled = old_row.copy()
        shuffled[: 100] = shuffled[: 100][::-1]          # rows mapped to the wrong old rows: survivors exceed the new degree somewhere
        try:
            e.derive_graph(g1, shuffled)
            derived_wrong = True
        except G.GnnvcError:
            derived_wrong = False
        if derived_wrong:                                  # the engine cannot see the caller's lists: the hashes tell
            assert not np.array_equal(e.row_hashes(), _fnv_rows(g1)) or g1.n > 5000
            e.upload_graph(g)
        _, logits = e.forward(g.x()) if not derived_wrong else e.forward(g.x())
        oracle_model.set_weight_scale(g.ws)
        e.set_weight_scale(g.ws)
        _, logits = e.forward(g.x())
        assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g)))
    finally:
        e.close()


# ---------------------------------------------------------------- dense layers: MFMA vs VALU

@pytest.mark.parametrize("mfma", [0, 1, 2])
def test_dense_layers_mfma_and_valu_are_bit_identical(model_text, oracle_model, mfma):
    """fp32 MFMA accumulates k-ordered with one rounding per product, i.e. the same fmaf chain
    as the VALU path and the reference's SGEMM: both variants must reproduce the oracle bits."""
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("mfma_dense", mfma)
        assert e.get_info("mfma_dense") == mfma
        rng = np.random.default_rng(5)
        for g in (gg.erdos_renyi(10000, 80000, 51), gg.rmat(11, 16, 6),
                  gg.from_edge_list(65, [(i, i + 1) for i in range(64)], list(range(20, 85)))):
            e.set_weight_scale(g.ws)
            oracle_model.set_weight_scale(g.ws)
            e.upload_graph(g)
            for x in (g.x(), rng.normal(size=g.n).astype(np.float32)):
                scores, logits = e.forward(x)
                assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g, x)))
                assert ulp(scores[:, 0], oracle_model.scores(g, x)).max() <= 1
    finally:
        e.close()


# ---------------------------------------------------------------- degree-sorted tiles

@pytest.mark.parametrize("mfma", [0, 2])
def test_sorted_tiles_are_bit_identical(model_text, oracle_model, mfma):
    """Tiles taken from a degree-sorted vertex list (skewed graphs) instead of 64 consecutive
    rows: every row is still summed alone, in CSR order — same bits; also over row sub-ranges."""
    import torch
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    try:
        e.set_option("sorted_tiles", 1)
        e.set_option("mfma_dense", mfma)
        e.set_option("sorted_long_row_threshold", 700)
        dev = torch.device("cuda:0")
        for g in (gg.rmat(12, 16, 9), gg.hub_graph(20000, 60000, 3, 4096, seed=7), gg.erdos_renyi(3000, 20000, 61),
                  gg.from_edge_list(130, [(0, i) for i in range(1, 40)], list(range(20, 150)))):
            e.set_weight_scale(g.ws)
            oracle_model.set_weight_scale(g.ws)
            e.upload_graph(g)
            assert e.get_info("sorted_tiles_active") == 1
            scores, logits = e.forward(g.x())
            assert np.array_equal(bits(logits[:, 0]), bits(oracle_model.logits(g)))
            assert ulp(scores[:, 0], oracle_model.scores(g)).max() <= 1
            h1 = torch.zeros((g.n + 1, 16), device=dev)
            h1[:-1] = torch.from_numpy(oracle_model.predict(g, g.x(), stop_after=6)).to(dev)
            h2 = torch.zeros((g.n + 1, 16), device=dev)
            torch.cuda.synchronize()
            cuts = [0, (g.n // 3) // 64 * 64, g.n]
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                e.stage_forward_device(1, lo, hi, h1.data_ptr(), h2.data_ptr())
            e.synchronize()
            assert np.array_equal(bits(h2[:-1].cpu().numpy()), bits(oracle_model.predict(g, g.x(), stop_after=13)))
    finally:
        e.close()


def test_sorted_tiles_auto_decision(model_text):
    """Auto mode sorts only when natural tiles would waste more than half of their rounds."""
    import gnn_mwvc_amd as G
    e = G.Engine(model_text, device=0)
    try:
        e.upload_graph(gg.erdos_renyi(20000, 200000, 71))      # uniform degrees: natural tiles
        assert e.get_info("sorted_tiles_active") == 0 and 100 <= e.get_info("tile_waste_x100") < 200
        g = gg.rmat(13, 16, 72)                                # power-law degrees, but a small graph:
        e.upload_graph(g)                                      # the sort would cost more than it saves
        assert e.get_info("sorted_tiles_active") == 0 and e.get_info("tile_waste_x100") >= 200
        e.set_option("sorted_min_nnz", 0)
        e.upload_graph(g)                                      # without the size gate: sorted tiles
        assert e.get_info("sorted_tiles_active") == 1
    finally:
        e.close()


# ---------------------------------------------------------------- score consumer keys (f-3)

def test_score_keys_match_oracle(engine, oracle_model):
    """min(s, 1 - s) and s > 0.5 from the device scores == the oracle's restatement of the driver's reads."""
    import torch
    g = gg.erdos_renyi(20000, 120000, 41)
    engine.set_weight_scale(g.ws)
    engine.upload_graph(g)
    scores, _ = engine.forward(g.x())
    keys, above = engine.score_keys()
    want_k, want_a = oracle_py.score_keys(scores[:, 0])
    assert np.array_equal(bits(keys), bits(want_k)) and np.array_equal(above, want_a)
    assert 0 < above.sum() < g.n
    # explicit device buffer, edge values: 0.5 exactly, the ends, values whose 1 - s rounds
    s = np.array([0.5, 0.0, 1.0, np.nextafter(np.float32(0.5), np.float32(1)), np.nextafter(np.float32(0.5), np.float32(0)),
                  1e-8, 1 - 1e-7, 0.25, 0.75, 3e-39], dtype=np.float32)
    d = torch.from_numpy(s).to("cuda:0")
    keys, above = engine.score_keys(d.data_ptr(), s.size)
    want_k, want_a = oracle_py.score_keys(s)
    assert np.array_equal(bits(keys), bits(want_k)) and np.array_equal(above, want_a)
    assert above.tolist() == [0, 0, 1, 1, 0, 0, 1, 0, 1, 0]


# ---------------------------------------------------------------- feature-row codec of the exchange

@pytest.mark.parametrize("mask,cap", [(0x0000, 64), (0x080B, 64), (0x0001, 0), (0x8000, 4096), (0x0FF0, 64),
                                      (0x0FFF, 64), (0x0555, 64)])
def test_row_codec_round_trip(engine, mask, cap):
    """gnnvc_column_counts / gnnvc_pack_rows / gnnvc_unpack_rows against numpy: dense columns plus the
    exception list restore every row exactly; an overflowing list (or none) is flagged."""
    import torch
    import gnn_mwvc_amd as G
    from gnn_mwvc_amd import distributed as D
    dev = torch.device("cuda:0")
    rows, lo, hi = 1000, 130, 901
    rng = np.random.default_rng(mask)
    a = rng.normal(size=(rows, 16)).astype(np.float32)
    a[rng.random(a.shape) < 0.3] = 0.0
    a[5, 3] = -0.0                                                # a negative zero is a zero
    live = [c for c in range(16) if mask >> c & 1]
    dead = [c for c in range(16) if not mask >> c & 1]
    a[:, dead] = 0.0
    stray = []                                                    # a few non-zeros outside the dense columns
    if dead and cap:
        for i, r in enumerate(range(lo + 3, hi, 97)):
            c = dead[i % len(dead)]
            a[r, c] = 1.5 + i
            stray.append((r, c))
    kp = max(4, (len(live) + 3) // 4 * 4)
    pk = D.Packing(mask, kp, cap, 0.0)
    feat = torch.from_numpy(a).to(dev)
    assert engine.live_columns(feat.data_ptr(), rows) == sum(1 << c for c in range(16) if np.any(a[:, c] != 0))
    assert [int(v) for v in engine.column_counts(feat.data_ptr(), rows)] == [int((a[:, c] != 0).sum()) for c in range(16)]
    dense_rows = hi - lo + 5                                      # the dense part may be longer than the piece
    region = torch.full((pk.piece_words(dense_rows),), 9.0, dtype=torch.float32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    codec = G.EngineRowCodec(engine)
    torch.cuda.synchronize()   # (torch filled these on its own stream)
    codec.pack(feat, lo, hi, pk, region, dense_rows, flag)
    engine.synchronize()
    assert int(flag.item()) == 0
    p = region.cpu().numpy()
    d = p[: dense_rows * kp].reshape(dense_rows, kp)
    assert np.array_equal(bits(d[: hi - lo, : len(live)]), bits(a[lo:hi][:, live]))
    assert np.all(d[: hi - lo, len(live):] == 0) and np.all(d[hi - lo:] == 9.0)
    exc = p[dense_rows * kp:].view(np.uint32)
    assert exc[0] == len(stray)
    got = {(int(r) + lo, int(c)): v for r, c, v, _ in exc[4: 4 + 4 * len(stray)].reshape(-1, 4)}
    assert got == {(r, c): int(a[r, c:c + 1].view(np.uint32)[0]) for r, c in stray}
    out = torch.full((rows, 16), 5.0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()   # the engine runs on its own stream: torch's fill must have landed first
    codec.unpack(region, dense_rows, lo, hi, pk, out)
    engine.synchronize()
    o = out.cpu().numpy()
    assert np.array_equal(o[lo:hi] == 0, a[lo:hi] == 0)
    assert np.array_equal(bits(np.where(o[lo:hi] == 0, 0, o[lo:hi])), bits(np.where(a[lo:hi] == 0, 0, a[lo:hi])))
    assert not np.signbit(o[lo:hi][o[lo:hi] == 0]).any()          # zeros come back as +0.0f
    assert np.all(o[:lo] == 5.0) and np.all(o[hi:] == 5.0)
    if dead:
        # one more stray than the list holds: bit 1; no list at all: bit 0
        extra = [(r, dead[0]) for r in range(lo, hi) if a[r, dead[0]] == 0][: max(cap - len(stray), 0) + 1]
        for r, c in extra:
            feat[r, c] = 1e-30
        torch.cuda.synchronize()
        codec.pack(feat, lo, hi, pk, region, dense_rows, flag)
        engine.synchronize()
        assert int(flag.item()) == (2 if len(extra) + len(stray) > cap else 0)   # (a 4096-entry list cannot overflow here)
        flag.zero_()
        torch.cuda.synchronize()
        engine.pack_rows(feat.data_ptr(), lo, hi, mask, kp, region.data_ptr(), flag.data_ptr())
        engine.synchronize()
        assert int(flag.item()) == 1
        flag.zero_()
        for r, c in extra + stray:
            feat[r, c] = 0.0
        feat[hi, dead[0]] = 3.0                                   # outside the shipped rows: not this call's business
        torch.cuda.synchronize()
        engine.pack_rows(feat.data_ptr(), lo, hi, mask, kp, region.data_ptr(), flag.data_ptr())
        engine.synchronize()
        assert int(flag.item()) == 0


@pytest.mark.parametrize("world,n,off,size", [(4, 1000, 0, 256), (4, 1000, 64, 128), (3, 700, 192, 64), (2, 130, 0, 128)])
def test_row_codec_unpack_gathered(engine, world, n, off, size):
    """gnnvc_unpack_gathered (all peers of an all-gathered piece in one launch) == per-peer gnnvc_unpack_rows."""
    import torch
    import gnn_mwvc_amd as G
    from gnn_mwvc_amd import distributed as D
    dev = torch.device("cuda:0")
    bounds = D.partition_bounds(n, world)
    per = bounds[0][1] - bounds[0][0]
    rng = np.random.default_rng(n + off)
    a = np.zeros((n + 64, 16), dtype=np.float32)
    a[:n, [0, 5, 11]] = rng.normal(size=(n, 3)).astype(np.float32)
    a[:n:7, 9] = 2.5                                              # travels in the exception lists
    feat = torch.from_numpy(a).to(dev)
    pk = D.Packing(1 << 0 | 1 << 5 | 1 << 11, 4, 64, 0.0)
    pw = pk.piece_words(size)
    buf = torch.zeros(world * pw, dtype=torch.float32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    codec = G.EngineRowCodec(engine)
    torch.cuda.synchronize()
    for r, (lo, hi) in enumerate(bounds):
        r0, r1 = min(lo + off, hi), min(lo + off + size, hi)
        codec.pack(feat, r0, r1, pk, buf[r * pw:(r + 1) * pw], size, flag)
    skip = 1
    one = torch.full((n + 64, 16), 5.0, dtype=torch.float32, device=dev)
    many = one.clone()
    torch.cuda.synchronize()
    codec.unpack_gathered(buf, world, skip, size, per, off, size, n, pk, one)
    for r, (lo, hi) in enumerate(bounds):
        r0, r1 = min(lo + off, hi), min(lo + off + size, hi)
        if r != skip and r1 > r0:
            codec.unpack(buf[r * pw:(r + 1) * pw], size, r0, r1, pk, many)
    engine.synchronize()
    assert int(flag.item()) == 0
    assert torch.equal(one.view(torch.int32), many.view(torch.int32))
    got = one.cpu().numpy()
    for r, (lo, hi) in enumerate(bounds):
        r0, r1 = min(lo + off, hi), min(lo + off + size, hi)
        if r == skip:
            assert np.all(got[lo:hi] == 5.0)
        else:
            assert np.array_equal(got[r0:r1], a[r0:r1])
    assert np.all(got[n:] == 5.0)


def test_row_codec_errors(engine):
    import gnn_mwvc_amd as G
    with pytest.raises(G.GnnvcError):
        engine.pack_rows(1, 0, 10, 0xFFFF, 12, 1, 1)              # 16 dense columns do not fit 12
    with pytest.raises(G.GnnvcError):
        engine.unpack_rows(1, 0, 10, 0x1, 5, 1)                   # packed width not a multiple of 4
    with pytest.raises(G.GnnvcError):
        engine.pack_rows(1, 0, 10, 0x1, 4, 1, 1, width=8)         # only 16-column rows
    with pytest.raises(G.GnnvcError):
        engine.pack_rows(1, 10, 5, 0x1, 4, 1, 1)                  # reversed row range


# ---------------------------------------------------------------- reduction-rule predicates (f-2)

def test_reduction_flags_match_oracle(engine):
    """The vertex-parallel predicate pass against the oracle's restatement of the reference's
    rule predicates (itself checked against the reference's own graph methods on CPU)."""
    for g in (gg.erdos_renyi(100000, 300000, 1), gg.erdos_renyi(50000, 60000, 2), gg.rmat(14, 4, 3),
              gg.hub_graph(20000, 30000, 2, 3000, seed=4),
              gg.erdos_renyi(30000, 45000, 8, lo=1, hi=9),          # small weights: the small-solver rules fire often
              gg.from_edge_list(8, [(0, 7), (1, 7), (7, 6), (6, 2), (6, 3), (2, 3), (4, 5)], [3, 4, 9, 2, 6, 6, 12, 5]),
              gg.from_edge_list(6, [(0, 2), (0, 3), (1, 2), (1, 3), (4, 5)], [10, 10, 20, 20, 7, 7])):
        engine.upload_graph(g)
        got = engine.reduction_flags(20)
        want = oracle_py.reduction_flags(g, 20)
        assert np.array_equal(got, want), [int(((got ^ want) >> b & 1).sum()) for b in range(7)]
        if g.n > 1000:
            assert (got & 0x20).any() and (got & 0x40).any()      # all seven rules are exercised
    assert engine.reduction_flags(3).max() <= 0x7F
