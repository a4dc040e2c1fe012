"""Randomised fuzz of vertex-partitioned runs on one GPU (GPU box): `world` engines on CSR slices (or on the whole graph with
row ranges), random options, stages driven in pieces on shared buffers, three repetitions, logits against the oracle; and
engines that are handed one graph after another (plan state must reset).
python scratch/experiments/fuzz_slices.py [cases=200] [seed0=0]"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import gnn_mwvc_amd as G
from gnn_mwvc_amd import distributed as D
from oracle import oracle_py
from tools import graphgen as gg

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
om = oracle_py.OracleModel(G.default_model_text())
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
dev = torch.device("cuda:0")


def graph(rng):
    kind = rng.choice(["er", "rmat", "hub", "chung"])
    s = int(rng.integers(1 << 30))
    if kind == "er":
        n = int(rng.integers(300, 30000)); return gg.erdos_renyi(n, int(n * rng.uniform(2, 10)), s)
    if kind == "rmat":
        return gg.rmat(int(rng.integers(10, 15)), int(rng.integers(4, 17)), s)
    if kind == "hub":
        n = int(rng.integers(3000, 30000))
        return gg.hub_graph(n, int(n * rng.uniform(2, 6)), int(rng.integers(1, 4)), int(rng.integers(200, min(n - 1, 12000))), seed=s)
    n = int(rng.integers(3000, 30000))
    return gg.chung_lu_hubs(n, float(rng.uniform(4, 10)), float(rng.uniform(2.0, 2.6)), int(rng.integers(0, 3)), int(rng.integers(200, min(n - 1, 5000))), seed=s)


def options(rng):
    return {"blocked_min_n": 0, "prune_min_entries": 0, "prune_min_drop_percent": int(rng.integers(0, 20)),
            "long_row_threshold": int(rng.choice([0, 40, 64, 256, 512])), "giant_row_threshold": int(rng.choice([0, 300, 1000, 4096, 16384])),
            "sorted_tiles": int(rng.choice([-1, 0, 1])), "prune_zero_rows": int(rng.choice([0, 1, 1, 2])),
            "giant_segments": int(rng.choice([-1, 0, 1])), "compact_gather": int(rng.choice([0, 1])), "lds_table": int(rng.choice([0, 1])),
            # round 4: table tiles (a choice of columns carried from graph to graph), wide tiles, plans at hand-off, poisoned buffers
            "table_tiles": int(rng.choice([0, 1, 1])), "table_tiles_min_n": int(rng.choice([0, 0, 49152])), "table_tiles_solo": int(rng.choice([0, 1])),
            "wide_tiles": int(rng.choice([0, 1, 1])), "plans_at_handoff": int(rng.choice([0, 1, 2])), "poison_features": 1,
            "forward_timing": int(rng.choice([0, 2]))}


bad = 0
t0 = time.time()
for case in range(cases):
    rng = np.random.default_rng(seed0 + case)
    g = graph(rng)
    om.set_weight_scale(g.ws)
    want = om.logits(g)
    opts = options(rng)
    if rng.random() < 0.6:      # ---- partitioned
        world = int(rng.integers(2, 5)); mode = str(rng.choice(["rows", "nnz"])); sliced = bool(rng.random() < 0.7); pieces = int(rng.integers(1, 4))
        bounds = D.partition_bounds(g.n, world, g.rowptr, mode)
        rp = torch.from_numpy(g.rowptr.astype(np.int64)).to(torch.int32).to(dev); col = torch.from_numpy(g.col.astype(np.int64)).to(torch.int32).to(dev)
        w = torch.from_numpy(g.w.astype(np.int64)).to(torch.int32).to(dev); nw = torch.from_numpy(g.nw.astype(np.int64)).to(torch.int32).to(dev)
        engines = []
        try:
            for lo, hi in bounds:
                e = G.Engine(G.default_model_text(), device=0); engines.append(e)
                for k, v in opts.items():
                    e.set_option(k, v)
                e.set_weight_scale(g.ws)
                if sliced:
                    sl = D.slice_csr(g.n, rp, col, w, nw, lo, hi)
                    torch.cuda.synchronize()
                    e.attach_graph_slice(g.n, lo, hi, sl.nnz, sl.rowptr.data_ptr(), sl.col.data_ptr(), sl.w.data_ptr(), sl.nw.data_ptr(), keepalive=sl)
                else:
                    torch.cuda.synchronize()
                    e.attach_graph_device(g.n, g.nnz, rp.data_ptr(), col.data_ptr(), w.data_ptr(), nw.data_ptr(), keepalive=(rp, col, w, nw))
            x = torch.from_numpy(g.x()).to(dev)
            for rep in range(3):
                h1 = torch.full((g.n + 1, 16), 7.0, device=dev); h2 = torch.full((g.n + 1, 16), 7.0, device=dev)
                h1[g.n] = 0.0; h2[g.n] = 0.0
                sc = torch.full((g.n,), 7.0, device=dev); lg = torch.full((g.n,), 7.0, device=dev)
                torch.cuda.synchronize()
                for st, (src, dst, lgt) in enumerate(((x, h1, None), (h1, h2, None), (h2, sc, lg))):
                    for e, (lo, hi) in zip(engines, bounds):
                        cuts = sorted(set([lo, hi] + [lo + ((hi - lo) * k // pieces) // 64 * 64 for k in range(1, pieces)]))
                        for a, b in zip(cuts[:-1], cuts[1:]):
                            if b > a:
                                e.stage_forward_device(st, a, b, src.data_ptr(), dst.data_ptr(), lgt.data_ptr() if lgt is not None else 0)
                    for e in engines:
                        e.synchronize()
                if not np.array_equal(bits(lg.cpu().numpy()), bits(want)):
                    bad += 1
                    d = np.flatnonzero(bits(lg.cpu().numpy()) != bits(want))
                    print(f"MISMATCH case {seed0 + case} partitioned world {world} {mode} sliced {sliced} pieces {pieces} rep {rep} n {g.n}: {len(d)} rows {d[:5].tolist()} opts {opts}", flush=True)
                    break
        finally:
            for e in engines:
                e.close()
    else:                       # ---- one engine, several graphs in a row
        e = G.Engine(G.default_model_text(), device=0)
        try:
            for k, v in opts.items():
                e.set_option(k, v)
            for gi in range(int(rng.integers(2, 6))):
                if gi:
                    g = graph(rng); om.set_weight_scale(g.ws); want = om.logits(g)
                e.set_weight_scale(g.ws)
                (e.upload_graph if rng.random() < 0.5 else e.upload_graph_staged)(g)
                reps = int(rng.integers(1, 7))
                for rep in range(reps):
                    x = g.x() if (rep != 3) else (g.x() * np.float32(0.37)).astype(np.float32)   # (another input in the middle of a longer sequence)
                    w_ = want if rep != 3 else om.logits(g, x)
                    _, lgh = e.forward(x)
                    if not np.array_equal(bits(lgh[:, 0]), bits(w_)):
                        bad += 1
                        print(f"MISMATCH case {seed0 + case} graph {gi} rep {rep} n {g.n} opts {opts}", flush=True)
                        break
        finally:
            e.close()
    if case % 25 == 24:
        print(f"{case + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("done:", cases, "cases,", bad, "mismatching")
