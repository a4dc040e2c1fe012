import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import numpy as np
import gnn_mwvc_amd as G
from oracle import oracle_py
from tools import graphgen as gg
text = G.default_model_text()
om = oracle_py.OracleModel(text)
g = gg.erdos_renyi(200, 800, 1)
om.set_weight_scale(g.ws)
want_h1 = om.predict(g, g.x(), stop_after=6)
want_h2 = om.predict(g, g.x(), stop_after=13)
want = om.logits(g)
import torch
dev = torch.device("cuda:0")
for mf in (0, 1):
    e = G.Engine(text, device=0)
    e.set_option("mfma_dense", mf)
    e.set_weight_scale(g.ws); e.upload_graph(g)
    x = torch.from_numpy(g.x()).to(dev)
    h1 = torch.zeros((g.n + 1, 16), device=dev); h2 = torch.zeros((g.n + 1, 16), device=dev)
    sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
    torch.cuda.synchronize()
    e.stage_forward_device(0, 0, g.n, x.data_ptr(), h1.data_ptr()); e.synchronize()
    a = h1[:-1].cpu().numpy()
    print("mfma", mf, "stage0 mismatches", int((a.view(np.uint32) != want_h1.view(np.uint32)).sum()), "of", a.size)
    if mf and (a != want_h1).any():
        bad = np.argwhere(a.view(np.uint32) != want_h1.view(np.uint32))
        print(" first bad", bad[:8].tolist(), a[bad[0][0]][:16], want_h1[bad[0][0]][:16])
    h1c = torch.zeros((g.n + 1, 16), device=dev); h1c[:-1] = torch.from_numpy(want_h1).to(dev)
    e.stage_forward_device(1, 0, g.n, h1c.data_ptr(), h2.data_ptr()); e.synchronize()
    b = h2[:-1].cpu().numpy()
    print("mfma", mf, "stage1 mismatches", int((b.view(np.uint32) != want_h2.view(np.uint32)).sum()), "of", b.size)
    if (b != want_h2).any():
        bad = np.argwhere(b.view(np.uint32) != want_h2.view(np.uint32))
        print(" first bad", bad[:8].tolist(), b[bad[0][0]][:16], want_h2[bad[0][0]][:16])
    h2c = torch.zeros((g.n + 1, 16), device=dev); h2c[:-1] = torch.from_numpy(want_h2).to(dev)
    e.stage_forward_device(2, 0, g.n, h2c.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
    c = lg.cpu().numpy()
    print("mfma", mf, "stage2 mismatches", int((c.view(np.uint32) != want.view(np.uint32)).sum()), "of", c.size, c[:4], want[:4])
    e.close()
