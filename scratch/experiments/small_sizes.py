"""Small graphs (the reference CLI's predict calls 2..n: N = 41 423, 19 675, 7 375, 1 933): forward time and per-kernel times.
python scratch/experiments/small_sizes.py [key=value ...]"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt
dev = torch.device("cuda", 0)
opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[1:]}
for n, m in ((2000, 8000), (7000, 30000), (20000, 100000), (41000, 200000), (60000, 600000), (100000, 1000000), (140000, 1400000), (300000, 3000000), (1000000, 10000000)):
    g = ggt.erdos_renyi(n, m, 1, dev)
    x = g.x().contiguous()
    sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
    torch.cuda.synchronize()
    e = G.Engine(G.default_model_text(), device=0)
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_weight_scale(g.ws)
    e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    for _ in range(20):
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    reps = 300
    t = time.perf_counter()
    for _ in range(reps):
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    back_to_back = (time.perf_counter() - t) / reps * 1e6
    t = time.perf_counter()
    for _ in range(reps):
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
    one_by_one = (time.perf_counter() - t) / reps * 1e6
    e.set_option("kernel_trace", 1)
    e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
    e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
    ks = [(k.split("<")[0].strip("("), round(ms * 1e3, 1)) for k, ms in e.kernel_trace(64)]
    print(f"n {n} m {m}: {back_to_back:.1f} us back to back, {one_by_one:.1f} us forward + sync; kernels (us) {ks}", flush=True)
    e.close()
