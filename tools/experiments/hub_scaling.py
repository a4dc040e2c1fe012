"""How does a stage's time scale with the degree of a single hub row? (long-row kernel)"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import gnn_mwvc_amd as G
from tools import graphgen as gg, graphgen_torch as ggt

dev = torch.device("cuda", 0)
eng = G.Engine(G.default_model_text(), device=0)
for hubs, hd in ((1, 8192), (1, 32768), (1, 131072), (4, 131072), (0, 0)):
    g = gg.hub_graph(400000, 600000, hubs, hd, seed=3) if hubs else gg.erdos_renyi(400000, 600000, 3)
    d = ggt.from_host(g, dev)
    eng.set_weight_scale(d.ws)
    eng.attach_graph_device(d.n, d.nnz, d.rowptr.data_ptr(), d.col.data_ptr(), d.w.data_ptr(), d.nw.data_ptr(), keepalive=d)
    x = d.x().contiguous(); sc = torch.zeros(d.n, device=dev); lg = torch.zeros(d.n, device=dev)
    for _ in range(3):
        eng.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); eng.synchronize()
    tot, st = eng.last_forward_ms()
    print(f"hubs={hubs} deg={hd} long_rows={eng.get_info('long_rows')} total={tot:.3f} ms stages={[round(v,3) for v in st]}",
          f" -> {st[1]*1e6/max(hd,1):.1f} ns per hub neighbour (F=16 stage)")
eng.close()
