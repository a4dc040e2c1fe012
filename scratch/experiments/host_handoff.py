"""The reference's call pattern through the host ABI on the metric graph: hand-off (upload or staged) + one forward, with the
plans built under the copy (plans_at_handoff 1), after it (early path off: handoff_min_entries huge + build in forward) ..."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import bench
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt

dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "er10m"
g, _ = bench.build_workload(wl, ggt, dev)
hg = g.to_host()
xh = hg.x()
del g
torch.cuda.empty_cache()
sc = np.empty((hg.n, 1), np.float32); lg = np.empty((hg.n, 1), np.float32)
ref = None
for label, opts, staged in (("warm-up", {}, False), ("upload, plans under the copy", {}, False), ("upload, plans in the forwards (handoff 0)", {"plans_at_handoff": 0}, False),
                            ("staged x8, plans under the copy", {}, True), ("staged x8, handoff 0", {"plans_at_handoff": 0}, True),
                            ("upload, plans under the copy", {}, False)):
    e = G.Engine(G.default_model_text(), device=0)
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_weight_scale(hg.ws)
    t0 = time.perf_counter()
    (e.upload_graph_staged(hg, pieces=8) if staged else e.upload_graph(hg))
    e.synchronize()
    t1 = time.perf_counter()
    e.forward(xh, out=(sc, lg))
    t2 = time.perf_counter()
    e.forward(xh, out=(sc, lg))
    t3 = time.perf_counter()
    if ref is None:
        ref = lg.copy()
    print(f"{label:45s} hand-off {1e3*(t1-t0):7.2f} ms (early {e.get_info('handoff_early_us')/1e3:.2f}, build at the end {e.get_info('handoff_build_us')/1e3:.2f}), "
          f"first forward {1e3*(t2-t1):6.2f}, second {1e3*(t3-t2):6.2f}, sum {1e3*(t2-t0):7.2f} ms; same bits {np.array_equal(ref.view(np.uint32), lg.view(np.uint32))}", flush=True)
    e.close()
