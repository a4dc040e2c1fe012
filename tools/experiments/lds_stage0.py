"""Prototype driver for tools/experiments/lds_stage0.hip (GPU box): builds the (chunk, block) segment
layout with torch, runs the LDS-table aggregation, checks it bit for bit against a CSR-order
reference and times it.  usage: python tools/experiments/lds_stage0.py [n] [m] [Rc] [Bc]"""
import ctypes as C
import pathlib
import subprocess
import sys

import torch

sys.path.insert(0, ".")
from tools import graphgen_torch as ggt  # noqa: E402

here = pathlib.Path(__file__).resolve().parent
so = here / "lds_stage0.so"
src = here / "lds_stage0.hip"
if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
    import os
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-shared", "-fPIC"] +
                          os.environ.get("LDS_FLAGS", "").split() + ["-o", str(so), str(src)])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10 * n
Rc = int(sys.argv[3]) if len(sys.argv) > 3 else 19536
Bc = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
dev = torch.device("cuda:0")
g = ggt.erdos_renyi(n, m, 10, dev)
L = C.CDLL(str(so))
ver = sys.argv[5] if len(sys.argv) > 5 else "v1"
fn = {"v0": L.lds_agg, "v1": L.lds_agg_v1, "v2": L.lds_agg_v2, "v3": L.lds_agg_v3, "v4": L.lds_agg_v4}[ver]
fn.argtypes = ([C.c_void_p] * 4 + [C.c_float, C.c_void_p] + [C.c_uint32] * 5 + [C.c_void_p]) if ver == 'v4' else \
    ([C.c_void_p] * 3 + [C.c_float, C.c_void_p] + [C.c_uint32] * 5 + [C.c_void_p] + ([C.c_uint32] if ver == 'v3' else []))
nchunks, nblocks = (n + Rc - 1) // Rc, (n + Bc - 1) // Bc
assert Rc <= 32768 and Bc <= 131072
rp = g.rowptr.to(torch.int64)
deg = rp[1:] - rp[:-1]
row = torch.repeat_interleave(torch.arange(n, device=dev), deg)
col = g.col[: g.nnz].to(torch.int64)
seg = (row // Rc) * nblocks + col // Bc
packed = ((row % Rc) << 17) | (col % Bc)
key = (seg << 32) | packed
del row, col
key, _ = torch.sort(key)
entries = (key & 0xFFFFFFFF).to(torch.int32)
seg_sorted = key >> 32
del key
seg_ptr = torch.zeros(nchunks * nblocks + 1, dtype=torch.int64, device=dev)
seg_ptr[1:] = torch.cumsum(torch.bincount(seg_sorted, minlength=nchunks * nblocks), 0)
seg_ptr = seg_ptr.to(torch.int32)
del seg_sorted
wb = torch.zeros(n + 131072 + 64, dtype=torch.uint8, device=dev)
wb[:n] = g.w.to(torch.uint8)
agg = torch.zeros(n, dtype=torch.float32, device=dev)
print(f"n={n} nnz={g.nnz} chunks={nchunks} x {Rc} rows, blocks={nblocks} x {Bc} cols, "
      f"avg entries/segment {g.nnz / (nchunks * nblocks):.0f}, LDS {Rc * 4 + 1024 + Bc} B")
stream = torch.cuda.current_stream().cuda_stream


entries_pad = torch.zeros(g.nnz + 2, dtype=torch.int32, device=dev)
entries_pad[1:-1] = entries


# v4: steps {block, first entry, count <= 2048, 0}, every chunk's list padded to a multiple of 4 (+ 8 spare at the end)
if ver == "v4":
    sp64 = seg_ptr.to(torch.int64)
    cnt = (sp64[1:] - sp64[:-1]).view(nchunks, nblocks)
    nst = torch.clamp((cnt + 2047) // 2048, min=1)                       # steps per (chunk, block)
    per_chunk = (nst.sum(1) + 3) // 4 * 4
    step_ptr = torch.zeros(nchunks + 1, dtype=torch.int64, device=dev)
    step_ptr[1:] = torch.cumsum(per_chunk, 0)
    total = int(step_ptr[-1].item())
    steps = torch.zeros((total + 8, 4), dtype=torch.int32, device=dev)
    flat_nst = nst.view(-1)
    seg_of_step = torch.repeat_interleave(torch.arange(nchunks * nblocks, device=dev), flat_nst)
    first_step_of_seg = torch.cumsum(flat_nst, 0) - flat_nst
    k_in_seg = torch.arange(seg_of_step.numel(), device=dev) - first_step_of_seg[seg_of_step]
    chunk_of_step = seg_of_step // nblocks
    # position of the step inside its chunk's list
    steps_before_chunk = torch.zeros(nchunks + 1, dtype=torch.int64, device=dev)
    steps_before_chunk[1:] = torch.cumsum(nst.sum(1), 0)
    pos = step_ptr[chunk_of_step] + (torch.arange(seg_of_step.numel(), device=dev) - steps_before_chunk[chunk_of_step])
    first = sp64[seg_of_step] + 2048 * k_in_seg
    count = torch.clamp(sp64[seg_of_step + 1] - first, max=2048)
    steps[pos, 0] = (seg_of_step % nblocks).to(torch.int32)
    steps[pos, 1] = first.to(torch.int32)
    steps[pos, 2] = count.to(torch.int32)
    step_ptr32 = step_ptr.to(torch.int32)
    print(f"steps: {total} ({total / nchunks:.0f} per chunk)")


def run():
    if ver == "v4":
        rc = fn(step_ptr32.data_ptr(), steps.data_ptr(), entries_pad[1:].data_ptr() if False else entries.data_ptr(),
                wb.data_ptr(), C.c_float(g.ws), agg.data_ptr(), n, Rc, Bc, nchunks, g.nnz - 1, stream)
    elif ver == "v3":
        rc = fn(seg_ptr.data_ptr(), entries_pad.data_ptr(), wb.data_ptr(), C.c_float(g.ws), agg.data_ptr(), n, Rc, Bc,
                nchunks, nblocks, stream, g.nnz)
    else:
        rc = fn(seg_ptr.data_ptr(), entries.data_ptr(), wb.data_ptr(), C.c_float(g.ws), agg.data_ptr(), n, Rc, Bc,
                nchunks, nblocks, stream)
    assert rc == 0, rc


run()
torch.cuda.synchronize()
# reference: CSR-order sequential fp32 sums, vectorised over rows (k-th neighbour of every row at step k)
x = g.x()
ref = torch.zeros(n, dtype=torch.float32, device=dev)
maxd = int(deg.max().item())
colp = g.col
for k in range(maxd):
    live = deg > k
    idx = (rp[:-1] + k)[live]
    ref[live] = ref[live] + x[colp[idx].to(torch.int64)]
bad = int((ref.view(torch.int32) != agg.view(torch.int32)).sum().item())
print("bit mismatches vs CSR-order sums:", bad)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(3):
    run()
ev[0].record()
for _ in range(10):
    run()
ev[1].record()
torch.cuda.synchronize()
print(f"lds_agg {ver}: {ev[0].elapsed_time(ev[1]) / 10:.3f} ms per pass")
