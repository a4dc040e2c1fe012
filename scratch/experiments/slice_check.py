"""Two engines, each on its slice (or on the whole graph with row ranges) of a bench workload, against one whole-graph forward.
python scratch/experiments/slice_check.py rmat20 2 nnz"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import torch
import bench
import gnn_mwvc_amd as G
from gnn_mwvc_amd import distributed as D
from tools import graphgen_torch as ggt

name, world, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
opts = dict(kv.split("=") for kv in sys.argv[4:])
dev = torch.device("cuda", 0)
g, _ = bench.build_workload(name, ggt, dev)
x = g.x().contiguous()
ref = G.Engine(G.default_model_text(), device=0)
for k, v in opts.items():
    ref.set_option(k, int(v))
ref.set_weight_scale(g.ws)
ref.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
sc0 = torch.zeros(g.n, device=dev); lg0 = torch.zeros(g.n, device=dev)
torch.cuda.synchronize()
ref.forward_device(x.data_ptr(), sc0.data_ptr(), lg0.data_ptr()); ref.synchronize()
bounds = D.partition_bounds(g.n, world, g.rowptr if mode == "nnz" else None, mode)
print("bounds", bounds, "long", ref.get_info("long_rows"), "giant", ref.get_info("giant_rows"), "sorted", ref.get_info("sorted_tiles_active"))
for sliced in (False, True):
    engines = []
    for lo, hi in bounds:
        e = G.Engine(G.default_model_text(), device=0)
        for k, v in opts.items():
            e.set_option(k, int(v))
        e.set_weight_scale(g.ws)
        if sliced:
            sl = D.slice_csr(g.n, g.rowptr, g.col, g.w, g.nw, lo, hi)
            torch.cuda.synchronize()
            e.attach_graph_slice(g.n, lo, hi, sl.nnz, sl.rowptr.data_ptr(), sl.col.data_ptr(), sl.w.data_ptr(), sl.nw.data_ptr(), keepalive=sl)
        else:
            e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
        engines.append(e)
    h1 = torch.zeros((g.n + 1, 16), device=dev); h2 = torch.zeros((g.n + 1, 16), device=dev)
    sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
    torch.cuda.synchronize()
    for rep in range(2):
        for st, (src, dst, l) in enumerate(((x, h1, None), (h1, h2, None), (h2, sc, lg))):
            for e, (lo, hi) in zip(engines, bounds):
                e.stage_forward_device(st, lo, hi, src.data_ptr(), dst.data_ptr(), l.data_ptr() if l is not None else 0)
            for e in engines:
                e.synchronize()
        bad = (lg.view(torch.int32) != lg0.view(torch.int32))
        print("sliced" if sliced else "ranges", "rep", rep, "logit mismatches", int(bad.sum()), "first", bad.nonzero()[:5].flatten().tolist())
    for e in engines:
        e.close()
