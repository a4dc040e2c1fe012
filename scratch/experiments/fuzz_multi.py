"""Randomised fuzz of the multi-device handle (gnnvc_create_multi, all "devices" = GPU 0): random graphs x parts x exchange
options (pieces, packing, push, announcements) x plan options, both host hand-off routes, several forwards with the INPUT changing
in the middle (another scale, an input that is no k / ws, arbitrary values) and a second graph on the same handle — logits against
the oracle bit for bit.  python scratch/experiments/fuzz_multi.py [cases=100] [seed0=0]"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import numpy as np
import gnn_mwvc_amd as G
from oracle import oracle_py
from tools import graphgen as gg

om = oracle_py.OracleModel(G.default_model_text())
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def graph(rng):
    kind = rng.choice(["er", "rmat", "hub", "chung", "tiny"])
    if kind == "er":
        n = int(rng.integers(2000, 60000)); return kind, gg.erdos_renyi(n, int(n * rng.uniform(2, 12)), int(rng.integers(1 << 30)))
    if kind == "rmat":
        return kind, gg.rmat(int(rng.integers(11, 15)), int(rng.integers(4, 17)), int(rng.integers(1 << 30)))
    if kind == "hub":
        n = int(rng.integers(5000, 40000))
        return kind, gg.hub_graph(n, int(n * rng.uniform(2, 8)), int(rng.integers(1, 5)), int(rng.integers(300, min(n - 1, 20000))), seed=int(rng.integers(1 << 30)))
    if kind == "chung":
        n = int(rng.integers(5000, 40000))
        return kind, gg.chung_lu_hubs(n, float(rng.uniform(4, 12)), float(rng.uniform(2.0, 2.6)), int(rng.integers(0, 4)), int(rng.integers(300, min(n - 1, 9000))), seed=int(rng.integers(1 << 30)))
    n = int(rng.integers(1, 700)); return kind, gg.erdos_renyi(n, int(n * rng.uniform(0, 4)), int(rng.integers(1 << 30)))


def one_case(case, override=None, verbose=False, parts_override=None):
    """-> number of mismatching (graph, forward) pairs of this case (stops at the first unless verbose)"""
    rng = np.random.default_rng(case)
    parts = int(rng.choice([1, 2, 3, 4, 8]))
    opts = {"blocked_min_n": 0, "prune_min_entries": 0, "multi_pieces": int(rng.choice([0, 1, 2, 4])), "multi_pack": int(rng.choice([0, 1, 1])),
            "multi_push": int(rng.choice([0, 1, 1])), "multi_announce": int(rng.choice([-1, 0, 1])),
            "lds_table": int(rng.choice([0, 1, 1])), "compact_gather": int(rng.choice([0, 1, 1])), "prune_zero_rows": int(rng.choice([0, 1, 1])),
            "sorted_tiles": int(rng.choice([-1, 0, 1])), "wide_tiles": int(rng.choice([0, 1, 1])), "forward_timing": int(rng.choice([0, 2]))}
    if rng.random() < 0.5: opts["long_row_threshold"] = int(rng.choice([0, 64, 256, 512]))
    if rng.random() < 0.5: opts["giant_row_threshold"] = int(rng.choice([0, 300, 4096]))
    opts["poison_features"] = 1   # (a row no kernel writes, or no exchange delivers, becomes a NaN in the result)
    if override:
        opts.update(override)
    if parts_override:
        parts = parts_override
    nbad = 0
    e = G.Engine(G.default_model_text(), devices=[0] * parts)
    try:
        for k, v in opts.items():
            e.set_option(k, v)
        for gi in range(2):
            kind, g = graph(rng)
            e.set_weight_scale(g.ws); om.set_weight_scale(g.ws)
            (e.upload_graph if rng.random() < 0.5 else e.upload_graph_staged)(g)
            xs = [g.x(), g.x(), g.x(), (g.x() * np.float32(2.0)).astype(np.float32), (g.x() * np.float32(0.37)).astype(np.float32),
                  g.x(), rng.normal(size=g.n).astype(np.float32), g.x()]
            per = []
            for rep, x in enumerate(xs):
                want = om.logits(g, x)
                _, lg = e.forward(x)
                d = np.flatnonzero((bits(lg[:, 0]) != bits(want)) & ~(np.isnan(lg[:, 0]) & np.isnan(want)))
                per.append(len(d))
                if len(d):
                    nbad += 1
                    if not verbose:
                        print(f"MISMATCH case {case} graph {gi} kind {kind} n {g.n} nnz {g.nnz} parts {parts} forward {rep}: {len(d)} rows, first {d[:5].tolist()} opts {opts}", flush=True)
                        return nbad
            if verbose:
                info = {k: e.get_info(k) for k in ("multi_packed_stage0", "multi_packed_stage1", "multi_packed_columns_stage0", "multi_packed_columns_stage1", "multi_pieces")}
                print(f"   graph {gi} {kind} n {g.n} nnz {g.nnz} parts {parts}: mismatching rows per forward {per} {info}", flush=True)
    finally:
        e.close()
    return nbad


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if len(sys.argv) > 3 and sys.argv[3] == "only":      # fuzz_multi.py 1 CASE only [parts=P] [key=value ...]: the case under variations
        kv = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[4:]}
        po = kv.pop("parts", None)
        print("case", seed0, "override", kv, "parts", po, flush=True)
        one_case(seed0, kv, verbose=True, parts_override=po)
        sys.exit(0)
    bad = 0
    t0 = time.time()
    for case in range(cases):
        bad += 1 if one_case(seed0 + case) else 0
        if case % 20 == 19:
            print(f"{case + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
    print("done:", cases, "cases,", bad, "mismatching")
    sys.exit(1 if bad else 0)
