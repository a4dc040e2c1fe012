import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt
dev = torch.device("cuda", 0)
n, m, deep = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = ggt.erdos_renyi(n, m, 1, dev)
x = g.x().contiguous()
sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
torch.cuda.synchronize()
e = G.Engine(G.default_model_text(), device=0)
e.set_option("deep_gather_rows", deep)
e.set_weight_scale(g.ws)
e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
for _ in range(50):
    e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
e.synchronize()
e.close()
