"""Do the per-graph plans pay below 2^20 vertices?  usage: min_n_check.py workload"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import gnn_mwvc_amd as G  # noqa: E402
from tools import graphgen_torch as ggt  # noqa: E402

dev = torch.device("cuda:0")
for wl in sys.argv[1:]:
    g, _ = bench.build_workload(wl, ggt, dev)
    for min_n in (1 << 20, 0):
        e = G.Engine(G.default_model_text(), device=0)
        e.set_option("blocked_min_n", min_n)
        e.set_weight_scale(g.ws)
        e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
        x = g.x().contiguous()
        sc = torch.zeros(g.n, device=dev)
        lg = torch.zeros(g.n, device=dev)
        torch.cuda.synchronize()
        for _ in range(4):
            e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
        e.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
        e.synchronize()
        ms = (time.perf_counter() - t) * 50
        print(f"{wl} n={g.n} min_n={min_n}: {ms:.3f} ms/forward, plans lt={e.get_info('lds_table_active')} c4={e.get_info('compact_gather_active')}"
              f" c4ok={e.get_info('compact_gather_last_ok')}")
        e.close()
