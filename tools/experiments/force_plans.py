"""Force the LDS-table / compact-table plans on a skewed workload and time the forward.  usage: force_plans.py workload"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import gnn_mwvc_amd as G  # noqa: E402
from tools import graphgen_torch as ggt  # noqa: E402

dev = torch.device("cuda:0")
wl = sys.argv[1]
g, _ = bench.build_workload(wl, ggt, dev)
x = g.x().contiguous()
ref = None
for opts in ({}, {"lds_table": 2}, {"lds_table": 2, "compact_gather": 2}):
    e = G.Engine(G.default_model_text(), device=0)
    for k, v in opts.items():
        e.set_option(k, v)
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
    same = True if ref is None else bool(torch.equal(lg.view(torch.int32), ref.view(torch.int32)))
    ref = lg.clone() if ref is None else ref
    print(f"{wl} {opts}: {ms:.3f} ms, stages {[round(v, 3) for v in e.last_forward_ms()[1]]}, lt={e.get_info('lds_table_active')} "
          f"c4={e.get_info('compact_gather_active')} ok={e.get_info('compact_gather_last_ok')}, same bits: {same}")
    e.close()
