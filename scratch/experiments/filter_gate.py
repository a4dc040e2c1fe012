"""First forward of large graphs under the filter's device-side bound (filter_min_percent): 50 (default), 65, and no filter."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt
dev = torch.device("cuda", 0)


def run(g, x, opts):
    best = 1e9
    info = {}
    for _ in range(3):
        e = G.Engine(G.default_model_text(), device=0)
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_weight_scale(g.ws)
        e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
        sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
        torch.cuda.synchronize()
        t = time.perf_counter(); e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
        best = min(best, (time.perf_counter() - t) * 1e3)
        info = {k: e.get_info(k) for k in ("filtered_stage1", "filter_mass_percent_stage1", "filter_mass_percent_stage2", "long_entries_percent")}
        e.close()
    return best, info


cases = [("pl", 6_000_000, 14.0, 11), ("pl", 5_000_000, 20.0, 12), ("rmat", 22, 8, 29), ("rmat", 21, 16, 28), ("rmat", 23, 8, 31), ("hubs", 4_000_000, 24.0, 13)]
for c in cases:
    if c[0] == "rmat": g = ggt.rmat(c[1], c[2], c[3], dev)
    elif c[0] == "pl": g = ggt.power_law_hubs(c[1], c[2], 2.1, 8, 65536, c[3], dev)
    else: g = ggt.power_law_hubs(c[1], c[2], 4.0, 6, 200000, c[3], dev)
    x = g.x().contiguous()
    r50, i50 = run(g, x, {})
    r65, _ = run(g, x, {"filter_min_percent": 65})
    r0, _ = run(g, x, {"filter_zero_rows": 0})
    print(f"{c} n {g.n} nnz {g.nnz}: first forward bound 50: {r50:.3f}  bound 65: {r65:.3f}  no filter: {r0:.3f}   {i50}", flush=True)
    del g, x
    torch.cuda.empty_cache()
