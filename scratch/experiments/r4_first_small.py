"""A small graph's FIRST forward (what the reference's driver gets, src/GNN_VC.cpp:171-192) against its later ones: fresh engines
in one process, forward + sync wall time and the kernels' HIP-event times.
python scratch/experiments/r4_first_small.py [key=value ...]"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt
dev = torch.device("cuda", 0)
opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[1:]}
for n, m in ((20000, 100000), (100000, 1000000), (7000, 30000)):
    for trial in range(4):
        g = ggt.erdos_renyi(n, m, 1 + trial, dev)
        x = g.x().contiguous()
        sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
        torch.cuda.synchronize()
        t = time.perf_counter()
        e = G.Engine(G.default_model_text(), device=0)
        create_us = (time.perf_counter() - t) * 1e6
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_weight_scale(g.ws)
        trace = trial == 3
        if trace:
            e.set_option("kernel_trace", 1)
        t = time.perf_counter()
        e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
        e.synchronize()
        attach_us = (time.perf_counter() - t) * 1e6
        ts, issue = [], []
        ks = []
        for rep in range(5):
            t = time.perf_counter()
            e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
            t1 = time.perf_counter()
            e.synchronize()
            ts.append((time.perf_counter() - t) * 1e6)
            issue.append((t1 - t) * 1e6)
            if trace:
                ks.append([(k.split("<")[0].strip("("), round(ms * 1e3, 1)) for k, ms in e.kernel_trace(64)])
        # the reference's pattern: the SAME engine handed the next graph (buffers in place: no allocation in it)
        g2 = ggt.erdos_renyi(n - n // 8, m - m // 8, 50 + trial, dev)
        x2 = g2.x().contiguous()
        torch.cuda.synchronize()
        e.set_weight_scale(g2.ws)
        t = time.perf_counter()
        e.attach_graph_device(g2.n, g2.nnz, g2.rowptr.data_ptr(), g2.col.data_ptr(), g2.w.data_ptr(), g2.nw.data_ptr(), keepalive=g2)
        e.synchronize()
        reattach_us = (time.perf_counter() - t) * 1e6
        t = time.perf_counter()
        e.forward_device(x2.data_ptr(), sc.data_ptr(), lg.data_ptr())
        e.synchronize()
        refirst_us = (time.perf_counter() - t) * 1e6
        print(f"n {n} trial {trial}: next graph on the same engine: attach {reattach_us:.0f} us, its first forward + sync {refirst_us:.1f} us", flush=True)
        print(f"n {n} trial {trial}: create {create_us:.0f} us, attach {attach_us:.0f} us, forward + sync (us) {[round(v, 1) for v in ts]}, "
              f"of which issuing {[round(v, 1) for v in issue]}", flush=True)
        for rep, kk in enumerate(ks[:3]):
            print(f"    forward {rep}: {kk}", flush=True)
        e.close()
