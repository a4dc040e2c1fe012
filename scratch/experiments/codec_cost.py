"""Time of the exchange codec kernels on one GPU at the metric graph's sizes (10 M rows).
For P ranks a rank packs N/P rows and expands N(P-1)/P rows per exchanged stage."""
import sys

import torch

sys.path.insert(0, ".")
import gnn_mwvc_amd as G  # noqa: E402
from gnn_mwvc_amd import distributed as D  # noqa: E402

dev = torch.device("cuda:0")
n = 10_000_000
e = G.Engine(G.default_model_text(), device=0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
e.set_stream(stream.cuda_stream)
codec = G.EngineRowCodec(e)
feat = torch.zeros(n + 64, 16, device=dev)
feat[:n, 0] = 1.0
feat[:n, 11] = 2.0
feat[: n // 8, 3] = 3.0
feat[::5000, 7] = 4.0
pk = D.choose_packing(codec.column_counts(feat, n), n, 8)
print("packing", pk)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
out = torch.zeros_like(feat)
for world in (2, 4, 8):
    per = n // world
    region = torch.zeros(pk.piece_words(per) * world, device=dev)
    pw = pk.piece_words(per)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for it in range(3):
        ev[0].record(stream)
        codec.pack(feat, 0, per, pk, region[:pw], per, flag)
        ev[1].record(stream)
        for peer in range(1, world):
            codec.unpack(region[:pw], per, peer * per, (peer + 1) * per, pk, out)
        ev[2].record(stream)
        torch.cuda.synchronize()
    print(f"P={world}: pack {per} rows {ev[0].elapsed_time(ev[1]):.3f} ms, expand {per * (world - 1)} rows "
          f"{ev[1].elapsed_time(ev[2]):.3f} ms, flag {int(flag.item())}")
