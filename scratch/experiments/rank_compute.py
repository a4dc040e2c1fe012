"""Per-rank compute of a P-rank partitioned run, measured on ONE GPU: rank r's engine holds only its CSR slice and runs each
stage over its whole row range (no pieces), with the range plans (LDS table over the slice, compact table announced per stage)
or with the plain kernels.  usage: python scratch/experiments/rank_compute.py [workload] [P] [rank] [key=value ...]"""
import sys
import time
import pathlib

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
import gnn_mwvc_amd as G  # noqa: E402
from gnn_mwvc_amd import distributed as D  # noqa: E402
from tools import graphgen_torch as ggt  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "er10m"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
opts = dict(kv.split("=") for kv in sys.argv[4:])
dev = torch.device("cuda:0")
g, _ = bench.build_workload(wl, ggt, dev)
n = g.n
x = g.x().contiguous()
# reference: the whole graph on one engine
e0 = G.Engine(G.default_model_text(), device=0)
e0.set_weight_scale(g.ws)
e0.attach_graph_device(n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
full = [torch.zeros(n + 64, 16, device=dev), torch.zeros(n + 64, 16, device=dev), torch.zeros(n + 64, device=dev)]
lg = torch.zeros(n + 64, device=dev)
torch.cuda.synchronize()
src = x
for st in range(3):
    e0.stage_forward_device(st, 0, n, src.data_ptr(), full[st].data_ptr(), lg.data_ptr() if st == 2 else 0)
    src = full[st]
e0.synchronize()
e0.close()
bounds = D.partition_bounds(n, P, g.rowptr if wl.startswith(("rmat", "power")) else None, "nnz" if wl.startswith(("rmat", "power")) else "rows")
lo, hi = bounds[rank]
sl = D.slice_csr(n, g.rowptr, g.col, g.w, g.nw, lo, hi)
out = [torch.zeros_like(t) for t in full]
for mode in ("plans", "plain"):
    e = G.Engine(G.default_model_text(), device=0)
    e.set_weight_scale(g.ws)
    for k, v in opts.items():
        e.set_option(k, int(v))
    if mode == "plain":
        e.set_option("lds_table", 0)
        e.set_option("compact_gather", 0)
        e.set_option("blocked_stage0", 0)
    torch.cuda.synchronize()
    e.attach_graph_slice(n, lo, hi, sl.nnz, sl.rowptr.data_ptr(), sl.col.data_ptr(), sl.w.data_ptr(), sl.nw.data_ptr(), keepalive=sl)
    for rep in range(6):
        times = []
        src = x
        for st in range(3):
            torch.cuda.synchronize()
            ts = time.perf_counter()
            if mode == "plans" and st >= 1:
                e.stage_input_ready(st, src.data_ptr(), lo, hi)
            e.stage_forward_device(st, lo, hi, src.data_ptr(), out[st].data_ptr(), lg.data_ptr() if st == 2 else 0)
            e.synchronize()
            times.append((time.perf_counter() - ts) * 1e3)
            src = full[st]
    same = all(torch.equal(out[st][lo:hi].view(torch.int32), full[st][lo:hi].view(torch.int32)) for st in range(3))
    info = {k: e.get_info(k) for k in ("lds_table_active", "lds_table_chunks", "compact_gather_active", "compact_gather_chunks", "compact_gather_last_ok")}
    print(f"{wl} rank {rank} of {P} rows [{lo},{hi}) {mode}: per-stage {[round(t, 3) for t in times]} ms, sum {sum(times):.3f} ms, "
          f"identical={same} {info}", flush=True)
    e.close()
