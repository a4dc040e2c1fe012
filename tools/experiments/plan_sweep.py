"""Robustness sweep: Erdős–Rényi graphs of several sizes and densities, per-graph plans on vs off — same bits?
and how long.  usage: plan_sweep.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
import gnn_mwvc_amd as G  # noqa: E402
from tools import graphgen_torch as ggt  # noqa: E402

dev = torch.device("cuda:0")
for n, deg, lo_w, hi_w in ((2_000_000, 4, 20, 120), (2_000_000, 100, 20, 120), (5_000_000, 10, 1, 255), (3_000_000, 30, 1, 1000),
                           (1_500_000, 20, 20, 120), (4_000_000, 2, 20, 120)):
    g = ggt.erdos_renyi(n, n * deg // 2, 5, dev, lo=lo_w, hi=hi_w)
    x = g.x().contiguous()
    res = {}
    for plans in (0, 1):
        e = G.Engine(G.default_model_text(), device=0)
        if not plans:
            e.set_option("lds_table", 0)
            e.set_option("compact_gather", 0)
            e.set_option("blocked_stage0", 0)
        e.set_weight_scale(g.ws)
        e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
        sc = torch.zeros(g.n, device=dev)
        lg = torch.zeros(g.n, device=dev)
        torch.cuda.synchronize()
        for _ in range(4):
            e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
        e.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
        e.synchronize()
        ms = (time.perf_counter() - t) * 100
        info = {k: e.get_info(k) for k in ("lds_table_active", "blocked_stage0_active", "compact_gather_active", "compact_gather_last_ok",
                                           "compact_gather_last_dirty", "sorted_tiles_active", "long_rows")}
        res[plans] = (ms, lg.clone(), info)
        e.close()
    same = bool(torch.equal(res[0][1].view(torch.int32), res[1][1].view(torch.int32)))
    print(f"n={n} deg={deg} w=[{lo_w},{hi_w}]: plain {res[0][0]:.3f} ms, plans {res[1][0]:.3f} ms, same bits {same}, {res[1][2]}", flush=True)
    del g, x, res
    torch.cuda.empty_cache()
