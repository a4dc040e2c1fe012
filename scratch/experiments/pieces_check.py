"""One process, no collectives: run every fused stage in the row pieces a P-rank pipelined run would use (plans
active) and compare with the whole-range result.  usage: python scratch/experiments/pieces_check.py [workload] [P] [chunks]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import gnn_mwvc_amd as G  # noqa: E402
from gnn_mwvc_amd import distributed as D  # noqa: E402
from tools import graphgen_torch as ggt  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "er3m"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
g, _ = bench.build_workload(wl, ggt, dev)
e = G.Engine(G.default_model_text(), device=0)
e.set_weight_scale(g.ws)
e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
x = g.x().contiguous()
n = g.n
full = [torch.zeros(n + 64, 16, device=dev), torch.zeros(n + 64, 16, device=dev), torch.zeros(n + 64, device=dev)]
lg = torch.zeros(n + 64, device=dev)
torch.cuda.synchronize()
for rep in range(3):     # the third pass runs with every plan built
    src = x
    for st in range(3):
        e.stage_forward_device(st, 0, n, src.data_ptr(), full[st].data_ptr(), lg.data_ptr() if st == 2 else 0)
        src = full[st]
    e.synchronize()
print("plans:", {k: e.get_info(k) for k in ("lds_table_active", "compact_gather_active")})
bounds = D.partition_bounds(n, P)
per = bounds[0][1] - bounds[0][0]
step = max(64, (per // chunks + 63) // 64 * 64)
out = [torch.zeros_like(t) for t in full]
torch.cuda.synchronize()
src = x
t0 = time.time()
for st in range(3):
    for lo, hi in bounds:
        for off in range(0, per, step):
            r0, r1 = min(lo + off, hi), min(lo + off + min(step, per - off), hi)
            if r1 > r0:
                e.stage_forward_device(st, r0, r1, src.data_ptr(), out[st].data_ptr(), lg.data_ptr() if st == 2 else 0)
    e.synchronize()
    same = torch.equal(out[st][:n].view(torch.int32), full[st][:n].view(torch.int32))
    print(f"stage {st}: {P} x {chunks} pieces, identical to the whole-range run: {same}  ({time.time() - t0:.2f} s)")
    src = full[st]

# per-rank compute of a P-rank run (rank 0's pieces only, no exchange), steady state
lo, hi = bounds[0]
prepare = len(sys.argv) > 4 and sys.argv[4] == "prepare"
if prepare:   # pieces of 256 of the plan's chunks
    e.stage_input_ready(1, full[0].data_ptr(), lo, hi)
    if e.get_info("compact_gather_active"):
        step = 256 * e.get_info("compact_gather_rows_per_chunk")
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    times = []
    src = x
    for st in range(3):
        ts = time.perf_counter()
        if prepare and st >= 1:
            e.stage_input_ready(st, src.data_ptr(), lo, hi)
        for off in range(0, per, step):
            r0, r1 = min(lo + off, hi), min(lo + off + min(step, per - off), hi)
            if r1 > r0:
                e.stage_forward_device(st, r0, r1, src.data_ptr(), out[st].data_ptr(), lg.data_ptr() if st == 2 else 0)
        e.synchronize()
        times.append((time.perf_counter() - ts) * 1e3)
        src = full[st]
same = all(torch.equal(out[st][lo:hi].view(torch.int32), full[st][lo:hi].view(torch.int32)) for st in range(3))
print(f"prepare={prepare} identical={same} plan rows [{lo},{hi}) chunks={e.get_info('compact_gather_chunks')}")
print(f"rank 0 of {P}: per-stage compute {[round(t, 3) for t in times]} ms, sum {sum(times):.3f} ms "
      f"(whole graph on one GPU / P = {4.79 / P:.3f} ms)")
