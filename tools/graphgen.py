"""Synthetic weighted graphs for parity tests and the bench (numpy, host side).

All graphs are simple and undirected; `CsrGraph.col` holds each vertex's
neighbours sorted ascending — the order `reduction_graph` stores them in after
construction from sorted unique (u<v) pairs (reference
include/reduction_graph.hpp:104-128) — because the forward's fp32 sums follow
that order.  Vertex weights are i.i.d. uniform integers in [20, 120]
(SURVEY.md §8d), drawn AFTER the edges so the streams match the recipes that
produced the golden vectors.

Generators:
  erdos_renyi(n, m, seed)        G(n, m); seed 1, n=1e5, m=1e6 reproduces the
                                 survey's ER-100K graph byte for byte
                                 (METIS text md5 f292fb0e...).
  hub_graph(n, m, hubs, hd)      sparse background + planted high-degree hubs
                                 (n=200000, m=600000, 3 x 65536 reproduces the
                                 survey's hub graph, md5 697c0a3b...).
  rmat(scale, edge_factor, seed) R-MAT (0.57, 0.19, 0.19, 0.05).
  chung_lu_hubs(n, ...)          power-law background + hubs of exact degree.
"""
from __future__ import annotations

import dataclasses
import hashlib
import io

import numpy as np


@dataclasses.dataclass
class CsrGraph:
    n: int
    rowptr: np.ndarray  # uint64, n+1
    col: np.ndarray     # uint32, nnz (sorted within each row)
    w: np.ndarray       # uint32, n
    nw: np.ndarray      # uint32, n  (sum of neighbour weights, uint32 wrap)

    @property
    def nnz(self) -> int:
        return int(self.rowptr[-1])

    @property
    def n_edges(self) -> int:
        return self.nnz // 2

    @property
    def ws(self) -> float:
        """Weight scale the CLI uses: max vertex weight (reference src/GNN_VC.cpp:272-278)."""
        return float(self.w.max()) if self.n else 1.0

    def x(self) -> np.ndarray:
        """Forward input x[u] = (float)W(u) / ws (reference src/GNN_VC.cpp:189-191)."""
        return (self.w.astype(np.float32) / np.float32(self.ws)).astype(np.float32)


def neighbourhood_weights(rowptr: np.ndarray, col: np.ndarray, w: np.ndarray) -> np.ndarray:
    n = len(rowptr) - 1
    if len(col) == 0:
        return np.zeros(n, dtype=np.uint32)
    contrib = w[col].astype(np.uint64)
    cs = np.zeros(len(col) + 1, dtype=np.uint64)
    np.cumsum(contrib, out=cs[1:])
    s = cs[rowptr[1:].astype(np.int64)] - cs[rowptr[:-1].astype(np.int64)]
    return (s & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def csr_from_pairs(n: int, a: np.ndarray, b: np.ndarray, w: np.ndarray) -> CsrGraph:
    """a<b unique pairs -> symmetric CSR with ascending neighbour lists."""
    a = np.asarray(a, dtype=np.int64)
    b = np.asarray(b, dtype=np.int64)
    src = np.concatenate([a, b])
    dst = np.concatenate([b, a])
    order = np.lexsort((dst, src))
    src = src[order]
    dst = dst[order]
    counts = np.bincount(src, minlength=n).astype(np.uint64)
    rowptr = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(counts, out=rowptr[1:])
    col = dst.astype(np.uint32)
    w = np.asarray(w, dtype=np.uint32)
    return CsrGraph(n, rowptr, col, w, neighbourhood_weights(rowptr, col, w))


def erdos_renyi(n: int, m: int, seed: int, lo: int = 20, hi: int = 120) -> CsrGraph:
    rng = np.random.default_rng(seed)
    draw = int(m * 1.1)
    u = rng.integers(0, n, size=draw)
    v = rng.integers(0, n, size=draw)
    keep = u != v
    u, v = u[keep], v[keep]
    key = np.unique(np.minimum(u, v).astype(np.int64) * n + np.maximum(u, v))
    rng.shuffle(key)
    key = np.sort(key[:m])
    w = rng.integers(lo, hi + 1, size=n)
    return csr_from_pairs(n, key // n, key % n, w)


def hub_graph(n: int, m: int, hubs: int, hub_degree: int, seed: int = 5) -> CsrGraph:
    rng = np.random.default_rng(seed)
    u = rng.integers(0, n, size=m)
    v = rng.integers(0, n, size=m)
    hu = np.repeat(np.arange(hubs), hub_degree)
    hv = np.concatenate(
        [rng.choice(np.arange(hubs, n), size=hub_degree, replace=False) for _ in range(hubs)]
    ) if hubs else np.zeros(0, dtype=np.int64)
    u = np.concatenate([u, hu])
    v = np.concatenate([v, hv])
    keep = u != v
    u, v = u[keep], v[keep]
    key = np.unique(np.minimum(u, v).astype(np.int64) * n + np.maximum(u, v))
    w = rng.integers(20, 121, size=n)
    return csr_from_pairs(n, key // n, key % n, w)


def rmat(scale: int, edge_factor: int, seed: int, a=0.57, b=0.19, c=0.19) -> CsrGraph:
    n = 1 << scale
    m = n * edge_factor
    rng = np.random.default_rng(seed)
    u = np.zeros(m, dtype=np.int64)
    v = np.zeros(m, dtype=np.int64)
    for bit in range(scale):
        r = rng.random(m)
        ub = (r >= a + b).astype(np.int64)
        vb = (((r >= a) & (r < a + b)) | (r >= a + b + c)).astype(np.int64)
        u |= ub << bit
        v |= vb << bit
    keep = u != v
    u, v = u[keep], v[keep]
    key = np.unique(np.minimum(u, v) * n + np.maximum(u, v))
    w = rng.integers(20, 121, size=n)
    return csr_from_pairs(n, key // n, key % n, w)


def chung_lu_hubs(n: int, avg_degree: float, exponent: float, hubs: int, hub_degree: int,
                  seed: int) -> CsrGraph:
    """Power-law expected-degree background (Chung-Lu) plus `hubs` vertices of
    exactly `hub_degree` distinct neighbours (BASELINE.json config 5)."""
    rng = np.random.default_rng(seed)
    ranks = np.arange(1, n + 1, dtype=np.float64)
    wts = ranks ** (-1.0 / (exponent - 1.0))
    p = wts / wts.sum()
    m = int(n * avg_degree / 2)
    u = rng.choice(n, size=m, p=p)
    v = rng.choice(n, size=m, p=p)
    perm = rng.permutation(n)          # de-correlate degree from vertex id
    u, v = perm[u], perm[v]
    keep = (u != v) & (u >= hubs) & (v >= hubs)
    u, v = u[keep], v[keep]
    hu = np.repeat(np.arange(hubs), hub_degree)
    hv = np.concatenate(
        [rng.choice(np.arange(hubs, n), size=hub_degree, replace=False) for _ in range(hubs)]
    ) if hubs else np.zeros(0, dtype=np.int64)
    u = np.concatenate([u, hu])
    v = np.concatenate([v, hv])
    key = np.unique(np.minimum(u, v).astype(np.int64) * n + np.maximum(u, v))
    w = rng.integers(20, 121, size=n)
    return csr_from_pairs(n, key // n, key % n, w)


def from_edge_list(n: int, edges, weights) -> CsrGraph:
    """Small hand-written graphs: edges are (u, v) 0-based pairs in any order."""
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    if len(e):
        a = np.minimum(e[:, 0], e[:, 1])
        b = np.maximum(e[:, 0], e[:, 1])
        keep = a != b
        key = np.unique(a[keep] * n + b[keep])
        a, b = key // n, key % n
    else:
        a = b = np.zeros(0, dtype=np.int64)
    return csr_from_pairs(n, a, b, np.asarray(weights, dtype=np.uint32))


# ---------------------------------------------------------------- METIS text

def metis_text(g: CsrGraph) -> str:
    """The reference's input format (README.md:49-62): header `N E 10`, then per
    vertex `weight n1 n2 ...` with 1-based neighbour ids."""
    out = io.StringIO()
    out.write(f"{g.n} {g.n_edges} 10\n")
    rp = g.rowptr.astype(np.int64)
    col1 = (g.col.astype(np.int64) + 1).astype(str)
    w = g.w.astype(str)
    for i in range(g.n):
        out.write(w[i] + " " + " ".join(col1[rp[i]:rp[i + 1]]) + "\n")
    return out.getvalue()


def metis_md5(g: CsrGraph) -> str:
    return hashlib.md5(metis_text(g).encode()).hexdigest()


def parse_metis(text: str) -> CsrGraph:
    """Loader with the semantics of the reference's parse_graph
    (src/GNN_VC.cpp:34-91): only neighbours with a larger id are kept from each
    line, pairs are sorted and de-duplicated, self-loops vanish."""
    lines = text.split("\n")
    n = int(lines[0].split()[0])
    w = np.zeros(n, dtype=np.uint32)
    aa, bb = [], []
    for i in range(n):
        t = lines[1 + i].split()
        w[i] = int(t[0])
        nb = np.asarray(t[1:], dtype=np.int64) - 1
        nb = nb[nb > i]
        aa.append(np.full(len(nb), i, dtype=np.int64))
        bb.append(nb)
    a = np.concatenate(aa) if aa else np.zeros(0, dtype=np.int64)
    b = np.concatenate(bb) if bb else np.zeros(0, dtype=np.int64)
    key = np.unique(a * n + b)
    return csr_from_pairs(n, key // n, key % n, w)
