"""A/B of one engine option on a workload: steady-state ms and bitwise logits against the default.
python scratch/experiments/opt_ab.py WORKLOAD key=value [key=value ...]"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt
import bench
dev = torch.device("cuda", 0)
name = sys.argv[1]
opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[2:]}
g, _ = bench.build_workload(name, ggt, dev)
x = g.x().contiguous()


def run(o):
    e = G.Engine(G.default_model_text(), device=0)
    for k, v in o.items():
        e.set_option(k, v)
    e.set_weight_scale(g.ws)
    e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
    t = time.perf_counter(); e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
    first = (time.perf_counter() - t) * 1e3
    first_lg = lg.clone()
    for _ in range(4):
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    ms = (time.perf_counter() - t) * 50
    e.close()
    return first, ms, first_lg, lg.clone(), sc.clone()


for rep in range(2):
    f0, m0, fl0, l0, s0 = run({})
    f1, m1, fl1, l1, s1 = run(opts)
    bad = [int((a.view(torch.int32) != b.view(torch.int32)).sum()) for a, b in ((fl0, fl1), (l0, l1), (s0, s1), (fl0, l0))]
    print(f"{name} {opts}: default first {f0:.3f} steady {m0:.3f} | option first {f1:.3f} steady {m1:.3f} | mismatches first/steady/scores/first-vs-steady {bad}", flush=True)
