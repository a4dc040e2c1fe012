"""Per-kernel times (the engine's own HIP events, option "kernel_trace") of a fresh engine's FIRST, second and a steady
forward on one of bench.py's workloads, with the wall time of each beside them and the plans in force.
python scratch/experiments/first_trace.py WORKLOAD [key=value ...]   (GPU box)"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt
import bench

dev = torch.device("cuda", 0)
name = sys.argv[1]
opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[2:]}
g, desc = bench.build_workload(name, ggt, dev)
x = g.x().contiguous()
sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
torch.cuda.synchronize()
INFO = ("lds_table_active", "compact_gather_active", "pruned_stage1", "pruned_stage2", "filtered_stage1", "filtered_stage2",
        "short_lists_stage1", "short_lists_stage2",
        "sorted_tiles_active", "long_rows", "giant_rows", "plan_build_us", "handoff_build_us")


def one(traced):
    e = G.Engine(G.default_model_text(), device=0)
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_option("kernel_trace", 1 if traced else 0)
    e.set_weight_scale(g.ws)
    t = time.perf_counter()
    e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    e.synchronize()
    print(f"{name} ({desc}) traced={traced}: attach {(time.perf_counter() - t) * 1e3:.3f} ms")
    first = None
    for i in range(4):
        lg.fill_(-1.0)
        t = time.perf_counter()
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
        ms = (time.perf_counter() - t) * 1e3
        if first is None:
            first = lg.clone()
        else:
            bad = int((first.view(torch.int32) != lg.view(torch.int32)).sum())
            print(f"  forward {i} vs forward 0: {bad} logits differ")
        info = {}
        for k in INFO:
            try:
                v = e.get_info(k)
            except Exception:
                continue
            if v:
                info[k] = v
        print(f"  forward {i}: {ms:.3f} ms  {info}")
        if traced and i in (0, 1, 3):
            tot = 0.0
            for kn, kms in e.kernel_trace(512):
                tot += kms
                print(f"      {kn[:70]:70s} {kms:8.3f}")
            print(f"      main-stream kernels: {tot:.3f} ms")
    e.close()


one(False)
one(True)
