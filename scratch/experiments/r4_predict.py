"""Round 4: the pruned adjacency built at HAND-OFF from the predicted set (prune_predict) against the filtered gather:
attach, first / second / third forward, steady state, bits against the plain path.  usage: r4_predict.py [workloads...]"""
import sys, time
sys.path.insert(0, ".")
import torch
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt
from bench import WORKLOADS, build_workload
dev = torch.device("cuda", 0)

def run(g, x, opts, want=None):
    e = G.Engine(G.default_model_text(), device=0)
    for k, v in opts.items(): e.set_option(k, v)
    e.set_weight_scale(g.ws)
    torch.cuda.synchronize()
    t = time.perf_counter()
    e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    e.synchronize()
    attach = (time.perf_counter() - t) * 1e3
    sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
    torch.cuda.synchronize()
    early, bad = [], []
    for i in range(3):
        t = time.perf_counter(); e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
        early.append((time.perf_counter() - t) * 1e3)
        if want is not None: bad.append(int((lg.view(torch.int32) != want.view(torch.int32)).sum()))
        if i == 0:
            info = {k: e.get_info(k) for k in ("pruned_predicted_stage1", "pruned_borrowed_stage2", "pruned_last_ok_stage1", "pruned_last_ok_stage2",
                                               "filtered_stage1", "pruned_entries_stage1", "pruned_vertices_stage1", "handoff_build_us")}
    for _ in range(3): e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    t = time.perf_counter()
    for _ in range(10): e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    steady = (time.perf_counter() - t) * 100
    if want is not None: bad.append(int((lg.view(torch.int32) != want.view(torch.int32)).sum()))
    info2 = {k: e.get_info(k) for k in ("pruned_predicted_stage1", "pruned_entries_stage1", "pruned_entries_stage2", "pruned_from_previous_stage2")}
    e.close()
    return attach, early, steady, info, info2, bad, lg.clone()

for name in (sys.argv[1:] or ["rmat22"]):
    g, desc = build_workload(name, ggt, dev)
    x = g.x().contiguous()
    torch.cuda.synchronize()
    print(desc, "n", g.n, "nnz", g.nnz, flush=True)
    _, _, sp, _, _, _, want = run(g, x, {"lds_table": 0, "compact_gather": 0, "prune_zero_rows": 0, "filter_zero_rows": 0})
    print(f"  plain steady {sp:.3f} ms", flush=True)
    for label, opts in (("warm-up", {}), ("predict", {}), ("predict", {}), ("filter (r3)", {"prune_predict": 0}), ("filter (r3)", {"prune_predict": 0}),
                        ("neither", {"prune_predict": 0, "filter_zero_rows": 0, "prune_early_entries": 0}),
                        ("predict, x != W/ws", {"_perturb": 1})):
        xx = x
        w2 = None
        if opts.pop("_perturb", 0):
            xx = (x * 0.5).contiguous()
            _, _, _, _, _, _, w2 = run(g, xx, {"lds_table": 0, "compact_gather": 0, "prune_zero_rows": 0, "filter_zero_rows": 0})
        a, early, st, i1, i2, bad, _ = run(g, xx, opts, want if w2 is None else w2)
        print(f"  {label:20s} attach {a:7.3f}  forwards {early[0]:7.3f} {early[1]:7.3f} {early[2]:7.3f}  steady {st:7.3f}  a+f {a + early[0]:7.3f}  bad {bad}\n      {i1}\n      {i2}", flush=True)
    del g, x, want
    torch.cuda.empty_cache()
