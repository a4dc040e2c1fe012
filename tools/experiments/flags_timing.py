"""gnnvc_reduction_flags on a large sparse graph: time on the GPU, time of the oracle's restatement on the host
cores, agreement, and how many (vertex, rule) pairs it clears.  usage: flags_timing.py [n] [m]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import gnn_mwvc_amd as G  # noqa: E402
from oracle import oracle_py  # noqa: E402
from tools import graphgen_torch as ggt  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 30_000_000
dev = torch.device("cuda:0")
g = ggt.erdos_renyi(n, m, 7, dev)
e = G.Engine(G.default_model_text(), device=0)
e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
e.reduction_flags(20)
t = time.perf_counter()
for _ in range(3):
    got = e.reduction_flags(20)
gpu_ms = (time.perf_counter() - t) / 3 * 1e3
hg = g.to_host()
t = time.perf_counter()
want = oracle_py.reduction_flags(hg, 20)
cpu_ms = (time.perf_counter() - t) * 1e3
looked = int((np.diff(hg.rowptr.astype(np.int64)) <= 20).sum())
fires = [int((got >> b & 1).sum()) for b in range(7)]
print(f"n={n} m={m}: GPU {gpu_ms:.1f} ms (incl. {n} B copied back), oracle on {oracle_py.num_threads()} threads {cpu_ms:.0f} ms, "
      f"equal {bool(np.array_equal(got, want))}")
print(f"vertices reduce_graph looks at: {looked}; rule fires on {fires}; cleared {100 * (1 - sum(fires) / (7 * looked)):.1f} % of the (vertex, rule) pairs")
