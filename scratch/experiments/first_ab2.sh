#!/bin/bash
# long rows' share of the entries + first forward with / without the filter, R-MAT family and the named workloads
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
import torch
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt
dev = torch.device("cuda", 0)
def run(g, x, opts):
    e = G.Engine(G.default_model_text(), device=0)
    for k, v in opts.items(): e.set_option(k, v)
    e.set_weight_scale(g.ws)
    e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
    torch.cuda.synchronize()
    t = time.perf_counter(); e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
    first = (time.perf_counter() - t) * 1e3
    info = {k: e.get_info(k) for k in ("long_entries_percent", "filter_mass_percent_stage1", "filter_mass_percent_stage2", "long_rows")}
    for _ in range(3): e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    t = time.perf_counter()
    for _ in range(5): e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    steady = (time.perf_counter() - t) * 200
    e.close()
    return first, steady, info, lg.clone()
cases = [("rmat", s, ef, 7 + s) for s in (19, 20, 21, 22) for ef in (4, 8, 16)] + [("pl", 1_000_000, 16.0, 5), ("pl", 2_000_000, 10.0, 6), ("pl", 4_000_000, 12.0, 7)]
for c in cases:
    if c[0] == "rmat": g = ggt.rmat(c[1], c[2], c[3], dev)
    else: g = ggt.power_law_hubs(c[1], c[2], 2.1, 8, 65536, c[3], dev)
    x = g.x().contiguous()
    FORCE = {"filter_min_entries": 0}
    run(g, x, FORCE)
    f1, s1, i1, l1 = run(g, x, FORCE)
    f1b, _, _, _ = run(g, x, FORCE)
    f0, s0, i0, l0 = run(g, x, {"filter_zero_rows": 0})
    f0b, _, _, _ = run(g, x, {"filter_zero_rows": 0})
    f5, _, _, _ = run(g, x, {"filter_min_entries": 0, "filter_min_long_percent": 0})
    bad = int((l1.view(torch.int32) != l0.view(torch.int32)).sum())
    print(f"{c} n {g.n} nnz {g.nnz}: first filter(any size) {min(f1, f1b):.3f} filter(any size, any long share) {f5:.3f} nofilter {min(f0, f0b):.3f} steady {s1:.3f}; {i1}; mismatches {bad}", flush=True)
    del g, x
    torch.cuda.empty_cache()
PY
