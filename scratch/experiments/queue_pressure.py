"""Does the number of LIVE engines (each holds four HIP streams) slow another engine's forward?  rmat22 / powerlaw1m steady state
with K idle engines alive."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt
import bench
dev = torch.device("cuda", 0)
name = sys.argv[1]
g, _ = bench.build_workload(name, ggt, dev)
x = g.x().contiguous()
sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
idle = []
for k in (0, 1, 2, 4, 8):
    while len(idle) < k:
        idle.append(G.Engine(G.default_model_text(), device=0))
    e = G.Engine(G.default_model_text(), device=0)
    e.set_weight_scale(g.ws)
    e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    for _ in range(4):
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    print(f"{name}: {k} idle engines alive: {(time.perf_counter() - t) * 50:.3f} ms per forward (side queue: {e.get_info('side_queue_probes')} probes, beside {e.get_info('side_queue_runs_beside')})", flush=True)
    e.close()
for e in idle:
    e.close()
