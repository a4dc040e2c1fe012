"""Where do a pruned and an unpruned forward differ on a small skewed graph?  (GPU box)"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import gnn_mwvc_amd as G
from tools import graphgen as gg

g = gg.chung_lu_hubs(40000, 8.0, 2.2, 2, 3000, seed=4)
deg = np.diff(g.rowptr.astype(np.int64))
dev = torch.device("cuda:0")


def engine(prune, giant, **kw):
    e = G.Engine(G.default_model_text(), device=0)
    for k, v in dict(blocked_min_n=0, long_row_threshold=256, sorted_long_row_threshold=512, giant_row_threshold=giant,
                     prune_min_drop_percent=1, prune_min_entries=0, prune_zero_rows=prune, **kw).items():
        e.set_option(k, v)
    e.set_weight_scale(g.ws)
    e.upload_graph(g)
    return e


def stages(e):
    x = torch.from_numpy(g.x().astype(np.float32)).to(dev).contiguous()
    xs = torch.zeros(g.n + 1, device=dev); xs[: g.n] = x.flatten()
    outs = []
    for rep in range(3):
        h1 = torch.zeros((g.n + 1, 16), device=dev); h2 = torch.zeros((g.n + 1, 16), device=dev)
        sc = torch.zeros(g.n + 1, device=dev); lg = torch.zeros(g.n + 1, device=dev)
        torch.cuda.synchronize()
        e.stage_forward_device(0, 0, g.n, xs.data_ptr(), h1.data_ptr(), 0)
        e.stage_forward_device(1, 0, g.n, h1.data_ptr(), h2.data_ptr(), 0)
        e.stage_forward_device(2, 0, g.n, h2.data_ptr(), sc.data_ptr(), lg.data_ptr())
        e.synchronize()
        outs.append((h1.cpu().numpy(), h2.cpu().numpy(), lg.cpu().numpy()))
    return outs


ref = stages(engine(0, 4096))
for giant, kw in ((4096, {}), (4096, {"prune_giant_rows": 0}), (4096, {"side_streams": 0}), (1 << 30, {})):
    e = engine(1, giant, **kw)
    print(kw)
    got = stages(e)
    print("giant threshold", giant, {k: e.get_info(k) for k in ("giant_rows", "long_rows", "pruned_stage1", "pruned_stage2", "pruned_bound_stage1",
                                                             "pruned_bound_stage2", "pruned_last_ok_stage1", "pruned_last_ok_stage2")})
    for rep in range(3):
        for name, a, b in zip(("h1", "h2", "logits"), ref[rep], got[rep]):
            bad = np.flatnonzero((a.view(np.uint32) != b.view(np.uint32)).reshape(len(a), -1).any(axis=1))
            print("  rep", rep, name, "rows that differ", len(bad), bad[:6], "degrees", deg[bad[:6]] if len(bad) else "",
                  a.reshape(len(a), -1)[bad[:2]].tolist(), b.reshape(len(b), -1)[bad[:2]].tolist())
    if giant == 4096 and not kw:
        # the stream kernel on the giant row's pruned and unpruned streams against numpy's sequential sum
      u = int(np.argmax(deg))
      cols = g.col[g.rowptr[u]: g.rowptr[u + 1]].astype(np.int64)
      for st in (1, 2):
        h1 = ref[2][st - 1]
        b1 = e.get_info(f"pruned_bound_stage{st}")
        keep = cols[deg[cols] < b1]
        np.save(f"gpurun_out/prune_debug_stream_stage{st}.npy", np.ascontiguousarray(h1[keep].T))
        for tag, cc in (("full", cols), ("pruned", keep)):
            vals = np.ascontiguousarray(h1[cc].T)          # 16 streams
            got_s = e.stream_sum(vals)
            want = np.zeros(16, np.float32)
            for v in vals.T:
                want = want + v
            print("  stage", st, "stream_sum", tag, "len", len(cc), "mismatching columns", int((got_s.view(np.uint32) != want.view(np.uint32)).sum()),
                  np.flatnonzero(got_s.view(np.uint32) != want.view(np.uint32)), got_s[:4], want[:4])
    e.close()
