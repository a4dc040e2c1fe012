"""Non-zero density of every feature column after stages 0 and 1 (GPU box): the premise of the
compressed inter-GPU exchange.  usage: python scratch/experiments/column_density.py [workload ...]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import gnn_mwvc_amd as G  # noqa: E402
from tools import graphgen_torch as ggt  # noqa: E402

dev = torch.device("cuda:0")
for wl in sys.argv[1:] or ["er10m"]:
    g, desc = bench.build_workload(wl, ggt, dev)
    e = G.Engine(G.default_model_text(), device=0)
    e.set_weight_scale(g.ws)
    e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    x = g.x().contiguous()
    torch.cuda.synchronize()
    h = [torch.zeros(g.n + 64, 16, device=dev) for _ in range(2)]
    src = x
    print(desc)
    for st in range(2):
        e.stage_forward_device(st, 0, g.n, src.data_ptr(), h[st].data_ptr(), 0)
        e.synchronize()
        c = e.column_counts(h[st].data_ptr(), g.n)
        ref = (h[st][: g.n] != 0).sum(0).cpu().numpy()
        assert (c.astype("int64") == ref).all(), (c, ref)
        print(f"  stage {st}: live mask {e.live_columns(h[st].data_ptr(), g.n):#06x}  counts " +
              " ".join(f"{int(v)}" for v in c))
        print("           density " + " ".join(f"{v / g.n:.4f}" for v in c))
        # entries that point to a vertex with a non-zero in the column (what a stray in that column would dirty), as a share
        # of all entries, for all rows and for the rows below degree 1024 only
        deg = (g.rowptr[1:] - g.rowptr[:-1]).to(torch.float64)[: g.n]
        nzm = (h[st][: g.n] != 0).to(torch.float64)
        print("           entry share " + " ".join(f"{float(v):.4f}" for v in (nzm * deg[:, None]).sum(0) / deg.sum()))
        small = (deg < 1024).to(torch.float64)
        print("           nz vertices of degree >= 1024 " + " ".join(f"{int(v)}" for v in (nzm * (1 - small)[:, None]).sum(0)))
        # is "the row is all zero" a function of the degree?  per degree bucket: vertices, share with a non-zero row, share of all entries
        anynz = (h[st][: g.n] != 0).any(1)
        mx = int(deg[anynz].max()) if bool(anynz.any()) else -1
        print(f"           vertices with a non-zero row: {int(anynz.sum())} of {g.n}; their largest degree {mx}; entries that point to them: "
              f"{float((deg * anynz).sum() / deg.sum()):.4f} of all")
        edges = [0, 1, 2, 4, 8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024, 1 << 30]
        for a, b in zip(edges[:-1], edges[1:]):
            m = (deg >= a) & (deg < b)
            cnt = int(m.sum())
            if cnt:
                print(f"             degree [{a}, {b}): {cnt} vertices, {float((anynz & m).sum()) / cnt:.4f} non-zero, {float((deg * m).sum() / deg.sum()):.4f} of the entries")
        src = h[st]
    e.close()
    del g, h, x
    torch.cuda.empty_cache()
