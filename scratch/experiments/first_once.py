"""attach + ONE forward of a fresh engine on one of bench.py's workloads (for a rocprofv3 kernel trace: first_timeline.sh)."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt
import bench

dev = torch.device("cuda", 0)
g, desc = bench.build_workload(sys.argv[1], ggt, dev)
x = g.x().contiguous()
sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
torch.cuda.synchronize()
e = G.Engine(G.default_model_text(), device=0)
for kv in sys.argv[2:]:
    e.set_option(kv.split("=")[0], int(kv.split("=")[1]))
e.set_weight_scale(g.ws)
e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
e.synchronize()
e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
e.close()
