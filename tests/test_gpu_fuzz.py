"""Randomised plan fuzz: random small graphs (Erdős–Rényi, R-MAT, hubs, power-law, dense) x random thresholds and plan
options — long / giant / sorted thresholds, pruned adjacency in both modes, the F = 1 and compact-table plans in their
skewed layouts, giant streams on one or several waves, MFMA or VALU dense layers, small plan chunks — three forwards each,
logits against the oracle bit for bit.  (`scratch/experiments/fuzz_plans.py` is the long form: 3 000 cases, 0 mismatches.)"""
import numpy as np
import pytest

from tools import graphgen as gg

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _graph(rng):
    kind = rng.choice(["er", "rmat", "hub", "chung", "dense"])
    seed = int(rng.integers(1 << 30))
    if kind == "er":
        n = int(rng.integers(2000, 60000))
        return gg.erdos_renyi(n, int(n * rng.uniform(2, 12)), seed)
    if kind == "rmat":
        return gg.rmat(int(rng.integers(11, 16)), int(rng.integers(4, 17)), seed)
    if kind == "hub":
        n = int(rng.integers(5000, 50000))
        return gg.hub_graph(n, int(n * rng.uniform(2, 8)), int(rng.integers(1, 5)), int(rng.integers(300, min(n - 1, 20000))), seed=seed)
    if kind == "chung":
        n = int(rng.integers(5000, 50000))
        return gg.chung_lu_hubs(n, float(rng.uniform(4, 12)), float(rng.uniform(2.0, 2.6)), int(rng.integers(0, 4)),
                                int(rng.integers(300, min(n - 1, 9000))), seed=seed)
    n = int(rng.integers(1500, 4000))
    return gg.erdos_renyi(n, n * int(rng.integers(60, 200)), seed)


def _options(rng):
    o = {"blocked_min_n": 0, "prune_min_entries": 0, "prune_min_drop_percent": int(rng.integers(0, 30))}
    if rng.random() < 0.7:
        o["long_row_threshold"] = int(rng.choice([0, 8, 40, 64, 128, 256, 512]))
    if rng.random() < 0.5:
        o["sorted_long_row_threshold"] = int(rng.choice([64, 256, 512, 1024, 2048]))
    if rng.random() < 0.7:
        o["giant_row_threshold"] = int(rng.choice([0, 64, 300, 1000, 4096, 16384]))
    if rng.random() < 0.4:
        o["giant_row_threshold_f16"] = int(rng.choice([64, 1000, 5000, 65536]))
    o["giant_segments"] = int(rng.choice([-1, 0, 1]))
    o["sorted_tiles"] = int(rng.choice([-1, 0, 1]))
    o["prune_zero_rows"] = int(rng.choice([0, 1, 1]))
    o["prune_class_by_entries_left"] = int(rng.choice([0, 1, 1]))
    o["prune_giant_rows"] = int(rng.choice([0, 1, 1]))
    o["prune_heavy_entries"] = int(rng.choice([1, 1 << 24]))
    o["lds_table"] = int(rng.choice([0, 1, 1]))
    o["lds_table_skewed_rows"] = int(rng.choice([0, 64, 512, 2048, 16384]))
    o["compact_gather"] = int(rng.choice([0, 1, 1]))
    o["mfma_dense"] = int(rng.choice([0, 1, 2]))
    o["overlap_dense"] = int(rng.choice([0, 1]))
    if rng.random() < 0.3:
        o["plan_chunk_rows"] = int(rng.choice([16, 48, 256, 4096]))
    o["table_tiles"] = int(rng.choice([0, 1, 1]))
    o["table_tiles_min_n"] = int(rng.choice([0, 0, 49152]))   # (0: the table tiles on these small graphs too)
    o["table_tiles_solo"] = int(rng.choice([0, 1]))
    o["wide_tiles"] = int(rng.choice([0, 1, 1]))
    o["poison_features"] = 1   # (round 4: a forward starts with NaN patterns in the engine's feature buffers: a row no kernel writes shows)
    return o


@pytest.mark.parametrize("block", range(4))
def test_random_graphs_and_plan_options(model_text, oracle_model, block):
    import gnn_mwvc_amd as G
    for case in range(block * 15, block * 15 + 15):
        rng = np.random.default_rng(90_000 + case)
        g, opts = _graph(rng), _options(rng)
        e = G.Engine(model_text, device=0)
        try:
            for k, v in opts.items():
                e.set_option(k, v)
            e.set_weight_scale(g.ws)
            oracle_model.set_weight_scale(g.ws)
            e.upload_graph(g)
            want = oracle_model.logits(g)
            for rep in range(5):
                _, lg = e.forward(g.x())
                assert np.array_equal(bits(lg[:, 0]), bits(want)), (case, rep, g.n, g.nnz, opts)
            # another input on the plans in force, then the first again: a row that no kernel writes goes unseen while the input stays
            # the same (round 4, fuzz_plans.py case 22174: the skewed LDS-table plan without a long-row list)
            x2 = (g.x() * np.float32(0.37)).astype(np.float32)
            _, lg = e.forward(x2)
            assert np.array_equal(bits(lg[:, 0]), bits(oracle_model.logits(g, x2))), (case, "other input", g.n, g.nnz, opts)
            _, lg = e.forward(g.x())
            assert np.array_equal(bits(lg[:, 0]), bits(want)), (case, "back", g.n, g.nnz, opts)
        finally:
            e.close()


def _oracle_stage(oracle_model, g, stage, h):
    from oracle import oracle_py
    a = oracle_py.graph_layer(g, g.ws, np.ascontiguousarray(h, dtype=np.float32))
    for i, (W, b) in enumerate(oracle_model.linear_params()[3 * stage: 3 * stage + 3]):
        a = oracle_py.linear_layer(a, W, b)
        if not (stage == 2 and i == 2):
            a = oracle_py.relu(a)
    return a


@pytest.mark.parametrize("block", range(3))
def test_random_stage_inputs(model_text, oracle_model, block):
    """The 16-wide stage entry point with plans built from the graph's own forwards and then ARBITRARY inputs: random live
    columns, zero rows that do or do not match what the pruned adjacency was built for, strays, a negative value, -0.0;
    whole range and sub-ranges.  (Long form: scratch/experiments/fuzz_stage_inputs.py — it found the one bug of this kind:
    with the giant threshold at or below the sorted long-row threshold, rows classed "long" by the entries they had left
    were nobody's.)"""
    import torch
    import gnn_mwvc_amd as G
    dev = torch.device("cuda:0")
    for case in range(block * 12, block * 12 + 12):
        rng = np.random.default_rng(70_000 + case)
        g = _graph(rng)
        if g.n > 40000:
            g = gg.rmat(13, 8, int(rng.integers(1 << 30)))
        deg = np.diff(g.rowptr.astype(np.int64))
        opts = {"blocked_min_n": 0, "prune_min_entries": 0, "prune_min_drop_percent": int(rng.integers(0, 20)),
                "long_row_threshold": int(rng.choice([64, 128, 256, 512])), "giant_row_threshold": int(rng.choice([300, 1000, 4096, 16384])),
                "sorted_tiles": int(rng.choice([-1, 0, 1])), "prune_zero_rows": 1,
                "giant_segments": int(rng.choice([0, 1]))}
        e = G.Engine(model_text, device=0)
        try:
            for k, v in opts.items():
                e.set_option(k, v)
            e.set_weight_scale(g.ws)
            oracle_model.set_weight_scale(g.ws)
            e.upload_graph(g)
            for _ in range(3):
                e.forward(g.x())
            for trial in range(2):
                live = rng.choice(16, int(rng.integers(1, 17)), replace=False)
                h = np.zeros((g.n, 16), dtype=np.float32)
                for c in live:
                    h[:, c] = rng.uniform(0.05, 2.0, g.n).astype(np.float32) * (rng.random(g.n) < rng.choice([1.0, 0.5, 0.1, 0.01]))
                mode = rng.choice(["as_built", "degree_zero", "random_zero", "none"])
                if mode == "degree_zero":
                    h[deg >= int(rng.choice([20, 60, 150]))] = 0.0
                elif mode == "random_zero":
                    h[rng.random(g.n) < 0.3] = 0.0
                elif mode == "as_built":
                    real = oracle_model.predict(g, g.x(), stop_after=6 if rng.random() < 0.5 else 13)
                    h[~(real != 0).any(axis=1)] = 0.0
                for i in rng.choice(g.n, int(rng.integers(0, 6)), replace=False):
                    h[i, int(rng.integers(16))] = 1.0 + (i % 5)
                if rng.random() < 0.15:
                    h[int(rng.integers(g.n)), int(rng.integers(16))] = -0.5
                if rng.random() < 0.3:
                    h[::7, int(rng.integers(16))] = -0.0
                hin = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
                hin[: g.n] = torch.from_numpy(h).to(dev)
                for st in (1, 2):
                    want = _oracle_stage(oracle_model, g, st, h)
                    lo = int(rng.integers(0, g.n // 2)) // 64 * 64
                    for a, b in ((0, g.n), (lo, g.n), (0, max(64, lo))):
                        out = torch.full((g.n + 1, 16 if st == 1 else 1), 7.0, dtype=torch.float32, device=dev)
                        lg = torch.full((g.n + 1,), 7.0, dtype=torch.float32, device=dev)
                        torch.cuda.synchronize()
                        e.stage_forward_device(st, a, b, hin.data_ptr(), out.data_ptr(), lg.data_ptr() if st == 2 else 0)
                        e.synchronize()
                        got = out[a:b].cpu().numpy() if st == 1 else lg[a:b].cpu().numpy().reshape(-1, 1)
                        assert np.array_equal(bits(got), bits(want[a:b])), (case, trial, mode, st, a, b, opts)
        finally:
            e.close()


def test_multi_device_handle_fuzz_slice():
    """A slice of scratch/experiments/fuzz_multi.py (random graphs x parts x exchange and plan options, two graphs per handle, the
    input changing between forwards) — with case 232, which found that two attempts were one too few for the packed exchange: a
    stage-0 overflow damages stage 1's input in the first attempt, so stage 1's own overflow shows only in the second."""
    import importlib.util
    import pathlib
    path = pathlib.Path(__file__).resolve().parents[1] / "scratch" / "experiments" / "fuzz_multi.py"
    spec = importlib.util.spec_from_file_location("fuzz_multi", path)
    fm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fm)
    for case in [232] + list(range(40, 52)):
        assert fm.one_case(case) == 0, case
