"""A/B: the engine on its own stream vs on a torch stream (power-law graph: side-stream concurrency matters)."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import bench
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "powerlaw1m"
g, _ = bench.build_workload(name, ggt, dev)
x = g.x().contiguous()
sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
torch.cuda.synchronize()

def measure(tag, use_torch_stream, pre=None):
    if pre: pre()
    e = G.Engine(G.default_model_text(), device=0)
    e.set_option("forward_timing", 2)
    e.set_weight_scale(g.ws)
    e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    if use_torch_stream:
        st = torch.cuda.Stream(device=dev)
        e.set_stream(st.cuda_stream)
    ts = []
    for i in range(6):
        torch.cuda.synchronize(); t = time.perf_counter()
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20):
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    ms = (time.perf_counter() - t) * 1e3 / 20
    print(tag, "steady %.3f" % ms, "early", ["%.2f" % v for v in ts], "stages", ["%.3f" % v for v in e.last_forward_ms()[1]], flush=True)
    e.close()

measure("own-stream  ", False)
measure("torch-stream", True)
measure("own-stream 2", False)
def churn():
    a = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(12)]
    del a
    torch.cuda.empty_cache()
measure("own after churn", False, churn)
measure("torch after churn", True)
