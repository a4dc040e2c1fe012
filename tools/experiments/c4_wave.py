"""Prototype driver for tools/experiments/c4_wave.hip (GPU box): compact-table F = 16 aggregation, wave-owned rows.
usage: python tools/experiments/c4_wave.py [n] [m] [Rw] [Bc] [grid] [depth]"""
import ctypes as C
import os
import pathlib
import subprocess
import sys

import torch

sys.path.insert(0, ".")
from tools import graphgen_torch as ggt  # noqa: E402

here = pathlib.Path(__file__).resolve().parent
KSUB = int(os.environ.get("KSUB", "4"))
PF = int(os.environ.get("PF", "0"))
so, src = here / f"c4_wave_k{KSUB}_p{PF}.so", here / "c4_wave.hip"
if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-shared", "-fPIC"] +
                          os.environ.get("C4_FLAGS", "").split() + [f"-DKSUB={KSUB}"] + ([f"-DPREFETCH={PF}"] if PF else []) + ["-o", str(so), str(src)])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10 * n
Rc = int(sys.argv[3]) if len(sys.argv) > 3 else 611
Bc = int(sys.argv[4]) if len(sys.argv) > 4 else 131072
grid = int(sys.argv[5]) if len(sys.argv) > 5 else 256
depth = int(sys.argv[6]) if len(sys.argv) > 6 else 1
pace = int(sys.argv[7]) if len(sys.argv) > 7 else 1
CAP = 64 * KSUB
dev = torch.device("cuda:0")
g = ggt.erdos_renyi(n, m, 10, dev)
L = C.CDLL(str(so))
L.c4w_agg.argtypes = [C.c_void_p] * 5 + [C.c_uint32] * 8 + [C.c_void_p]
nchunks, nblocks = (n + Rc - 1) // Rc, (n + Bc - 1) // Bc
assert Rc <= 640 and Bc <= 262144
rp = g.rowptr.to(torch.int64)
deg = rp[1:] - rp[:-1]
row = torch.repeat_interleave(torch.arange(n, device=dev), deg)
col = g.col[: g.nnz].to(torch.int64)
seg = (row // Rc) * nblocks + col // Bc
key = (seg << 32) | ((row % Rc) << 18) | (col % Bc)
del row, col
key, _ = torch.sort(key)
entries = (key & 0xFFFFFFFF).to(torch.int32)
seg_sorted = key >> 32
del key
sp64 = torch.zeros(nchunks * nblocks + 1, dtype=torch.int64, device=dev)
sp64[1:] = torch.cumsum(torch.bincount(seg_sorted, minlength=nchunks * nblocks), 0)
del seg_sorted
cnt = (sp64[1:] - sp64[:-1]).view(nchunks, nblocks)
nst = (cnt + CAP - 1) // CAP                                   # steps per (chunk, block); none for empty segments
per_chunk = torch.clamp((nst.sum(1) + 3) // 4 * 4, min=4)
step_ptr = torch.zeros(nchunks + 2, dtype=torch.int64, device=dev)
step_ptr[1:nchunks + 1] = torch.cumsum(per_chunk, 0)
step_ptr[nchunks + 1] = step_ptr[nchunks]                     # an empty slice for waves past the end
total = int(step_ptr[-1].item())
steps = torch.zeros((total + 8, 4), dtype=torch.int32, device=dev)
flat = nst.view(-1)
seg_of_step = torch.repeat_interleave(torch.arange(nchunks * nblocks, device=dev), flat)
first_step_of_seg = torch.cumsum(flat, 0) - flat
k_in_seg = torch.arange(seg_of_step.numel(), device=dev) - first_step_of_seg[seg_of_step]
chunk_of_step = seg_of_step // nblocks
before = torch.zeros(nchunks + 1, dtype=torch.int64, device=dev)
before[1:] = torch.cumsum(nst.sum(1), 0)
pos = step_ptr[chunk_of_step] + (torch.arange(seg_of_step.numel(), device=dev) - before[chunk_of_step])
first = sp64[seg_of_step] + CAP * k_in_seg
steps[pos, 0] = (seg_of_step % nblocks).to(torch.int32)
steps[pos, 1] = first.to(torch.int32)
steps[pos, 2] = torch.clamp(sp64[seg_of_step + 1] - first, max=CAP).to(torch.int32)
nnz_plan = g.nnz
if depth == 4:   # the lane-contiguous form reads 16 bytes per lane: every (slice, block) segment starts at a multiple of 4
    cpad = (cnt.view(-1) + 3) // 4 * 4
    spp = torch.zeros_like(sp64)
    spp[1:] = torch.cumsum(cpad, 0)
    seg_all = torch.repeat_interleave(torch.arange(nchunks * nblocks, device=dev), cnt.view(-1))
    dest = spp[seg_all] + (torch.arange(g.nnz, device=dev) - sp64[seg_all])
    nnz_plan = int(spp[-1].item())
    padded = torch.zeros(nnz_plan + 8, dtype=torch.int32, device=dev)
    padded[dest] = entries
    entries = padded
    steps[pos, 1] = (spp[seg_of_step] + CAP * k_in_seg).to(torch.int32)
    del seg_all, dest
step_ptr32 = step_ptr.to(torch.int32)
table = torch.zeros((n + 1, 4), dtype=torch.float32, device=dev)
gen = torch.Generator(device=dev)
gen.manual_seed(1)
table[:n] = torch.rand((n, 4), generator=gen, device=dev)
table[:n, 1] *= (torch.rand(n, generator=gen, device=dev) < 0.07)     # like the live columns of h1: two dense, two sparse
table[:n, 3] *= (torch.rand(n, generator=gen, device=dev) < 0.13)
agg = torch.zeros((n, 4), dtype=torch.float32, device=dev)
print(f"n={n} nnz={g.nnz} chunks={nchunks} x {Rc} rows, blocks={nblocks} x {Bc} cols, steps {total} "
      f"({total / nchunks:.0f} per chunk), grid {grid}, depth {depth}, LDS {Rc * 256} B")
stream = torch.cuda.current_stream().cuda_stream


def run():
    rc = L.c4w_agg(step_ptr32.data_ptr(), steps.data_ptr(), entries.data_ptr(), table.data_ptr(), agg.data_ptr(), n, Rc, Bc,
                   nchunks, nnz_plan, grid, depth, nblocks if pace else 0, stream)
    assert rc == 0, rc


run()
torch.cuda.synchronize()
ref = torch.zeros((n, 4), dtype=torch.float32, device=dev)
for k in range(int(deg.max().item())):
    live = deg > k
    idx = (rp[:-1] + k)[live]
    ref[live] = ref[live] + table[g.col[idx].to(torch.int64)]
print("bit mismatches vs CSR-order sums:", int((ref.view(torch.int32) != agg.view(torch.int32)).sum().item()))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(3):
    run()
ev[0].record()
for _ in range(10):
    run()
ev[1].record()
torch.cuda.synchronize()
t = ev[0].elapsed_time(ev[1]) / 10
print(f"c4w_agg: {t:.3f} ms per pass = {g.nnz / t / 1e6:.1f} G gathers/s")


# ---- co-residency experiment (depth 4 form only): the aggregation kernel and a VALU-bound kernel on two streams
if depth == 4 and os.environ.get("CORES"):
    L.valu_burn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    sink = torch.zeros(1 << 22, dtype=torch.float32, device=dev)
    s2 = torch.cuda.Stream(device=dev)
    blocks, iters = 256 * 40, int(os.environ["CORES"])

    def burn(st):
        assert L.valu_burn(sink.data_ptr(), blocks, iters, st.cuda_stream) == 0

    def timed(fn, reps=5):
        torch.cuda.synchronize()
        t0 = __import__("time").perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (__import__("time").perf_counter() - t0) / reps * 1e3

    cur = torch.cuda.current_stream()
    t_agg = timed(run)
    t_burn = timed(lambda: burn(cur))

    def both():
        run()          # current stream
        burn(s2)       # second stream, no dependency

    both()
    # each kernel's own span when both are in flight
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    eb = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ea[0].record(cur)
    run()
    ea[1].record(cur)
    eb[0].record(s2)
    burn(s2)
    eb[1].record(s2)
    torch.cuda.synchronize()
    print(f"in flight together: agg span {ea[0].elapsed_time(ea[1]):.3f} ms, VALU kernel span {eb[0].elapsed_time(eb[1]):.3f} ms, "
          f"VALU kernel ends {ea[0].elapsed_time(eb[1]):.3f} ms after the agg's start")
    t_both = timed(both)
    print(f"agg alone {t_agg:.3f} ms, VALU kernel alone {t_burn:.3f} ms, both on two streams {t_both:.3f} ms "
          f"(sum {t_agg + t_burn:.3f})")
