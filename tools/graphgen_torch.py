"""Device-side synthetic graph construction (torch on the GPU — plumbing, not product).

The metric graph (Erdős–Rényi 10 M vertices / 100 M edges) is
built directly as CSR in HBM: drawing, de-duplicating and sorting 2e8 entries
takes a few seconds on the GPU versus minutes in numpy.  Same construction as
tools/graphgen.erdos_renyi (draw 1.1 m pairs, drop loops, canonicalise,
unique, keep m, symmetrise, sort) but with torch's Philox streams, so the two
generators produce different graphs of the same distribution.
"""
from __future__ import annotations

import dataclasses

import torch

COL_PAD = 64


@dataclasses.dataclass
class DeviceCsr:
    n: int
    nnz: int
    rowptr: torch.Tensor  # int32 [n+1]  (bit pattern of uint32)
    col: torch.Tensor     # int32 [nnz + COL_PAD]
    w: torch.Tensor       # int32 [n]
    nw: torch.Tensor      # int32 [n]   (uint32 wrap-around sum)
    ws: float

    @property
    def n_edges(self) -> int:
        return self.nnz // 2

    def x(self) -> torch.Tensor:
        return self.w.to(torch.float32) / torch.tensor(self.ws, dtype=torch.float32, device=self.w.device)

    def to_host(self):
        """-> tools.graphgen.CsrGraph (for the oracle)."""
        import numpy as np
        from tools.graphgen import CsrGraph
        return CsrGraph(self.n,
                        self.rowptr.cpu().numpy().view(np.uint32).astype(np.uint64),
                        self.col[: self.nnz].cpu().numpy().view(np.uint32).copy(),
                        self.w.cpu().numpy().view(np.uint32).copy(),
                        self.nw.cpu().numpy().view(np.uint32).copy())


def csr_from_unique_pairs(n: int, key: torch.Tensor, w: torch.Tensor) -> DeviceCsr:
    """key = a*n + b with a < b, unique (any order), int64 on the device."""
    dev = key.device
    a = torch.div(key, n, rounding_mode="floor")
    b = key - a * n
    key2 = torch.cat([a * n + b, b * n + a])
    del a, b
    key2, _ = torch.sort(key2)
    src = torch.div(key2, n, rounding_mode="floor")
    col64 = key2 - src * n
    del key2
    counts = torch.bincount(src, minlength=n)
    del src
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rowptr[1:])
    nnz = int(rowptr[-1].item())
    assert nnz < 2**31, "device generator keeps 31-bit row pointers"
    # NW(u) = sum of neighbour weights, uint32 wrap-around
    wsum = torch.zeros(nnz + 1, dtype=torch.int64, device=dev)
    torch.cumsum(w.to(torch.int64)[col64], 0, out=wsum[1:])
    nw = (wsum[rowptr[1:]] - wsum[rowptr[:-1]]) & 0xFFFFFFFF
    del wsum
    nw = torch.where(nw >= 2**31, nw - 2**32, nw).to(torch.int32)
    col = torch.zeros(nnz + COL_PAD, dtype=torch.int32, device=dev)
    col[:nnz] = col64.to(torch.int32)
    del col64
    return DeviceCsr(n, nnz, rowptr.to(torch.int32), col, w.to(torch.int32), nw,
                     float(w.max().item()) if n else 1.0)


def erdos_renyi(n: int, m: int, seed: int, device, lo: int = 20, hi: int = 120) -> DeviceCsr:
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    draw = int(m * 1.1)
    u = torch.randint(0, n, (draw,), generator=gen, device=device, dtype=torch.int64)
    v = torch.randint(0, n, (draw,), generator=gen, device=device, dtype=torch.int64)
    keep = u != v
    u, v = u[keep], v[keep]
    key = torch.unique(torch.minimum(u, v) * n + torch.maximum(u, v))
    del u, v, keep
    if key.numel() > m:
        perm = torch.randperm(key.numel(), generator=gen, device=device)[:m]
        key = key[perm]
        del perm
    w = torch.randint(lo, hi + 1, (n,), generator=gen, device=device, dtype=torch.int64)
    return csr_from_unique_pairs(n, key, w)


def rmat(scale: int, edge_factor: int, seed: int, device, a=0.57, b=0.19, c=0.19) -> DeviceCsr:
    """R-MAT (0.57, 0.19, 0.19, 0.05), duplicates and self-loops removed, symmetrised."""
    n = 1 << scale
    m = n * edge_factor
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    u = torch.zeros(m, dtype=torch.int64, device=device)
    v = torch.zeros(m, dtype=torch.int64, device=device)
    for bit in range(scale):
        r = torch.rand(m, generator=gen, device=device)
        u |= (r >= a + b).to(torch.int64) << bit
        v |= (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int64) << bit
        del r
    keep = u != v
    u, v = u[keep], v[keep]
    key = torch.unique(torch.minimum(u, v) * n + torch.maximum(u, v))
    del u, v, keep
    w = torch.randint(20, 121, (n,), generator=gen, device=device, dtype=torch.int64)
    return csr_from_unique_pairs(n, key, w)


def power_law_hubs(n: int, avg_degree: float, exponent: float, hubs: int, hub_degree: int, seed: int,
                   device) -> DeviceCsr:
    """Chung-Lu power-law background plus `hubs` vertices of exactly `hub_degree`
    distinct neighbours (BASELINE.json config 5)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    ranks = torch.arange(1, n + 1, dtype=torch.float64, device=device)
    p = ranks ** (-1.0 / (exponent - 1.0))
    m = int(n * avg_degree / 2)
    cdf = torch.cumsum(p / p.sum(), 0)
    draw = lambda: torch.searchsorted(cdf, torch.rand(m, generator=gen, device=device, dtype=torch.float64)).clamp_(max=n - 1)
    perm = torch.randperm(n, generator=gen, device=device)
    u, v = perm[draw()], perm[draw()]
    keep = (u != v) & (u >= hubs) & (v >= hubs)
    u, v = u[keep], v[keep]
    hu = torch.arange(hubs, device=device).repeat_interleave(hub_degree)
    hv = torch.cat([torch.randperm(n - hubs, generator=gen, device=device)[:hub_degree] + hubs
                    for _ in range(hubs)]) if hubs else torch.zeros(0, dtype=torch.int64, device=device)
    u = torch.cat([u, hu])
    v = torch.cat([v, hv])
    key = torch.unique(torch.minimum(u, v) * n + torch.maximum(u, v))
    w = torch.randint(20, 121, (n,), generator=gen, device=device, dtype=torch.int64)
    return csr_from_unique_pairs(n, key, w)


def from_host(g, device) -> DeviceCsr:
    """tools.graphgen.CsrGraph -> device CSR."""
    import numpy as np
    col = np.zeros(g.nnz + COL_PAD, dtype=np.uint32)
    col[: g.nnz] = g.col
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(device)
    return DeviceCsr(g.n, g.nnz, t(g.rowptr.astype(np.uint32)), t(col), t(g.w.astype(np.uint32)),
                     t(g.nw.astype(np.uint32)), g.ws)
