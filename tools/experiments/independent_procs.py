"""Several INDEPENDENT processes (no collectives) running forwards with the per-graph plans on ONE GPU: tells a
GPU-sharing problem apart from a problem in the multi-rank driver.  usage: independent_procs.py [workload] [forwards]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import gnn_mwvc_amd as G  # noqa: E402
from tools import graphgen_torch as ggt  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "er3m"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
g, _ = bench.build_workload(wl, ggt, dev)
e = G.Engine(G.default_model_text(), device=0)
e.set_weight_scale(g.ws)
e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
x = g.x().contiguous()
sc = torch.zeros(g.n, device=dev)
lg = torch.zeros(g.n, device=dev)
torch.cuda.synchronize()
t = time.time()
for i in range(k):
    e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
e.synchronize()
print(f"{k} forwards in {time.time() - t:.2f} s, plans {e.get_info('lds_table_active')}{e.get_info('compact_gather_active')}", flush=True)
