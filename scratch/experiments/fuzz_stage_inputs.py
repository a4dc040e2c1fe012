"""Randomised fuzz of the 16-wide stage entry point (GPU box): plans built from the graph's own forwards, then ARBITRARY
stage inputs — random live columns, zero rows that do or do not match what the pruned adjacency was built for, strays,
negative values, -0.0 — whole range and a sub-range, against the oracle's layer functions bit for bit.
python scratch/experiments/fuzz_stage_inputs.py [cases=300] [seed0=0]"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import gnn_mwvc_amd as G
from oracle import oracle_py
from tools import graphgen as gg

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
om = oracle_py.OracleModel(G.default_model_text())
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
dev = torch.device("cuda:0")


def oracle_stage(g, stage, h):
    a = oracle_py.graph_layer(g, g.ws, np.ascontiguousarray(h, dtype=np.float32))
    for i, (W, b) in enumerate(om.linear_params()[3 * stage: 3 * stage + 3]):
        a = oracle_py.linear_layer(a, W, b)
        if not (stage == 2 and i == 2):
            a = oracle_py.relu(a)
    return a


bad = 0
t0 = time.time()
for case in range(cases):
    rng = np.random.default_rng(seed0 + case)
    kind = rng.choice(["er", "rmat", "hub", "chung"])
    if kind == "er":
        n = int(rng.integers(3000, 40000)); g = gg.erdos_renyi(n, int(n * rng.uniform(3, 12)), int(rng.integers(1 << 30)))
    elif kind == "rmat":
        g = gg.rmat(int(rng.integers(11, 15)), int(rng.integers(4, 17)), int(rng.integers(1 << 30)))
    elif kind == "hub":
        n = int(rng.integers(5000, 40000))
        g = gg.hub_graph(n, int(n * rng.uniform(2, 8)), int(rng.integers(1, 4)), int(rng.integers(300, min(n - 1, 12000))), seed=int(rng.integers(1 << 30)))
    else:
        n = int(rng.integers(5000, 40000))
        g = gg.chung_lu_hubs(n, float(rng.uniform(4, 12)), float(rng.uniform(2.0, 2.6)), int(rng.integers(0, 3)), int(rng.integers(300, min(n - 1, 6000))), seed=int(rng.integers(1 << 30)))
    deg = np.diff(g.rowptr.astype(np.int64))
    opts = {"blocked_min_n": 0, "prune_min_entries": 0, "prune_min_drop_percent": int(rng.integers(0, 20)),
            "long_row_threshold": int(rng.choice([64, 128, 256, 512])), "giant_row_threshold": int(rng.choice([300, 1000, 4096, 16384])),
            "sorted_tiles": int(rng.choice([-1, 0, 1])), "prune_zero_rows": int(rng.choice([0, 1, 1])),
            "compact_gather": int(rng.choice([0, 1, 1])), "dense_skip_zeros": int(rng.choice([0, 1, 1])), "forward_timing": int(rng.choice([0, 2])),
            "giant_segments": int(rng.choice([0, 1]))}   # (round 4: compact_skewed / compact_passes / prune mode 2 are gone)
    e = G.Engine(G.default_model_text(), device=0)
    try:
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_weight_scale(g.ws); om.set_weight_scale(g.ws)
        e.upload_graph(g)
        for _ in range(3):
            e.forward(g.x())
        for trial in range(3):
            ncols = int(rng.integers(1, 17))
            live = rng.choice(16, ncols, replace=False)
            h = np.zeros((g.n, 16), dtype=np.float32)
            for c in live:
                h[:, c] = rng.uniform(0.05, 2.0, g.n).astype(np.float32) * (rng.random(g.n) < rng.choice([1.0, 0.5, 0.1, 0.01]))
            mode = rng.choice(["as_built", "degree_zero", "random_zero", "none"])
            if mode == "degree_zero":
                h[deg >= int(rng.choice([20, 60, 150]))] = 0.0
            elif mode == "random_zero":
                h[rng.random(g.n) < 0.3] = 0.0
            elif mode == "as_built":   # the rows the real stage inputs have all zero, plus noise elsewhere
                real = om.predict(g, g.x(), stop_after=6 if rng.random() < 0.5 else 13)
                h[~(real != 0).any(axis=1)] = 0.0
            for i in rng.choice(g.n, int(rng.integers(0, 6)), replace=False):
                h[i, int(rng.integers(16))] = 1.0 + (i % 5)
            if rng.random() < 0.15: h[int(rng.integers(g.n)), int(rng.integers(16))] = -0.5
            if rng.random() < 0.3: h[::7, int(rng.integers(16))] = -0.0
            hin = torch.zeros((g.n + 1, 16), dtype=torch.float32, device=dev)
            hin[: g.n] = torch.from_numpy(h).to(dev)
            for st in (1, 2):
                want = oracle_stage(g, st, h)
                lo = int(rng.integers(0, g.n // 2)) // 64 * 64
                for a, b in ((0, g.n), (lo, g.n), (0, max(64, lo))):
                    out = torch.full((g.n + 1, 16 if st == 1 else 1), 7.0, dtype=torch.float32, device=dev)
                    lg = torch.full((g.n + 1,), 7.0, dtype=torch.float32, device=dev)
                    torch.cuda.synchronize()
                    e.stage_forward_device(st, a, b, hin.data_ptr(), out.data_ptr(), lg.data_ptr() if st == 2 else 0)
                    e.synchronize()
                    got = out[a:b].cpu().numpy() if st == 1 else lg[a:b].cpu().numpy().reshape(-1, 1)
                    if not np.array_equal(bits(got), bits(want[a:b])):
                        bad += 1
                        d = np.flatnonzero((bits(got) != bits(want[a:b])).any(axis=1)) + a
                        print(f"MISMATCH case {seed0 + case} {kind} n {g.n} trial {trial} mode {mode} stage {st} range {a}:{b}: rows {d[:5].tolist()} deg {deg[d[:5]].tolist()} opts {opts}", flush=True)
    finally:
        e.close()
    if case % 25 == 24:
        print(f"{case + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("done:", cases, "cases,", bad, "mismatching")
