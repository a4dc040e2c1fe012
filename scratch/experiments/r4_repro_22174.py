"""fuzz_plans case 22174 (R-MAT scale 12, another input after nine forwards): which option / which forward makes it differ."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import gnn_mwvc_amd as G
from oracle import oracle_py
from tools import graphgen as gg
case = 22174
rng = np.random.default_rng(case)
kind = rng.choice(["er", "rmat", "hub", "chung", "dense"])
assert kind == "rmat"
g = gg.rmat(int(rng.integers(11, 16)), int(rng.integers(4, 17)), int(rng.integers(1 << 30)))
print("graph", g.n, g.nnz, flush=True)
base = {'blocked_min_n': 0, 'prune_min_entries': 0, 'prune_min_drop_percent': 19, 'long_row_threshold': 0, 'sorted_long_row_threshold': 256, 'giant_row_threshold': 1000, 'giant_row_threshold_f16': 65536, 'giant_segments': -1, 'sorted_tiles': 1, 'prune_zero_rows': 0, 'prune_class_by_entries_left': 1, 'prune_giant_rows': 0, 'prune_heavy_entries': 1, 'lds_table': 1, 'lds_table_skewed_rows': 64, 'compact_gather': 1, 'mfma_dense': 2, 'overlap_dense': 0, 'filter_zero_rows': 1, 'filter_min_entries': 0, 'filter_min_long_percent': 0, 'filter_min_percent': 101, 'filter_keep_lists': 1, 'long_rows_on_main': 0, 'giant_gather_first': 1, 'side_streams': 1, 'prune_predict': 1, 'prune_predict_min_entries': 0, 'dense_skip_zeros': 0, 'lds_table_bits': 0, 'table_tiles': 1, 'table_tiles_min_n': 0, 'table_tiles_solo': 0, 'wide_tiles': 1, 'forward_timing': 2}
om = oracle_py.OracleModel(G.default_model_text())
om.set_weight_scale(g.ws)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
want = om.logits(g)
x_other = (g.x() * np.float32(0.37)).astype(np.float32)
w_other = om.logits(g, x_other)
deg = np.diff(g.rowptr.astype(np.int64))


def run(tag, change, at=9, reps=14):
    opts = dict(base); opts.update(change)
    e = G.Engine(G.default_model_text(), device=0)
    try:
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_weight_scale(g.ws)
        e.upload_graph(g)
        out = []
        for rep in range(reps):
            if rep == at:
                _, lg = e.forward(x_other)
                d = np.flatnonzero(bits(lg[:, 0]) != bits(w_other))
            else:
                _, lg = e.forward(g.x())
                d = np.flatnonzero(bits(lg[:, 0]) != bits(want))
            out.append(len(d))
            if len(d) and rep == at:
                print(f"   rows {d[:8].tolist()} degrees {deg[d[:8]].tolist()} got {lg[d[:3], 0].tolist()} want {w_other[d[:3]].tolist()}")
        info = {k: e.get_info(k) for k in ("table_tiles_active", "lds_table_active", "compact_gather_active", "sorted_tiles_active", "long_rows", "giant_rows", "pruned_stage1", "pruned_stage2")}
        print(f"{tag:34s} mismatching rows per forward {out}  {info}", flush=True)
    finally:
        e.close()


run("as the fuzz ran it", {})
run("verdict every forward", {"verdict_period": 1})
run("other input at forward 2", {}, at=2)
run("other input at forward 5", {}, at=5)
run("table_tiles 0", {"table_tiles": 0})
run("table_tiles_solo 1", {"table_tiles_solo": 1})
run("lds_table 0", {"lds_table": 0})
run("compact_gather 0", {"compact_gather": 0})
run("sorted_tiles 0", {"sorted_tiles": 0})
run("forward_timing 0", {"forward_timing": 0})
run("wide_tiles 0", {"wide_tiles": 0})
run("filter 0", {"filter_zero_rows": 0})
run("predict 0", {"prune_predict": 0})
run("giant threshold 0", {"giant_row_threshold": 0})
