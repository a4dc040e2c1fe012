"""The skewed-graph compact plan against the plain gathering kernels on a bench workload: bit mismatches of logits and
scores over three forwards, what the device decided, ms per forward.
python scratch/experiments/mapped_check.py rmat22 [key=value ...]"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import torch
import bench
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt

name = sys.argv[1]
opts = dict(kv.split("=") for kv in sys.argv[2:] if "=" in kv)
dev = torch.device("cuda", 0)
g, _ = bench.build_workload(name, ggt, dev)
x = g.x().contiguous()


def run(extra):
    e = G.Engine(G.default_model_text(), device=0)
    for k, v in {**opts, **extra}.items():
        e.set_option(k, int(v))
    e.set_weight_scale(g.ws)
    e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
    torch.cuda.synchronize()
    outs = []
    for rep in range(4):
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
        outs.append((sc.clone(), lg.clone()))
    t0 = time.time()
    for rep in range(10):
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    ms = (time.time() - t0) * 100
    info = {k: e.get_info(k) for k in ("compact_gather_active", "compact_gather_mapped", "compact_gather_last_passes", "compact_gather_last_dirty",
                                       "compact_gather_blocks", "compact_gather_steps", "compact_gather_mapped_rows",
                                       "compact_gather_mapped_entries", "compact_gather_rows_per_chunk", "compact_gather_chunks",
                                       "long_rows", "giant_rows", "plan_build_us", "pruned_stage1", "pruned_stage2", "pruned_bound_stage1",
                                       "pruned_bound_stage2", "pruned_entries_stage1", "pruned_entries_stage2", "pruned_last_ok_stage1",
                                       "pruned_last_ok_stage2")}
    info = {k: v for k, v in info.items() if v}
    e.close()
    return outs, ms, info


print(name, "n", g.n, "nnz", g.nnz)
ref, ms0, i0 = run({"compact_skewed": 0, "prune_zero_rows": 0})
print("plain ", round(ms0, 3), "ms", i0)
variants = [("pruned", {"prune_zero_rows": 1})]
if "--mapped" in sys.argv:
    variants += [("mapped", {"compact_skewed": 1, "prune_zero_rows": 0}), ("pruned+mapped", {"compact_skewed": 1, "prune_zero_rows": 1})]
for tag, extra in variants:
    got, ms1, i1 = run(extra)
    print(tag, round(ms1, 3), "ms", i1)
    for rep, ((s0, l0), (s1, l1)) in enumerate(zip(ref, got)):
        print("   forward", rep, "logit mismatches", int((l0.view(torch.int32) != l1.view(torch.int32)).sum()),
              "score mismatches", int((s0.view(torch.int32) != s1.view(torch.int32)).sum()))
