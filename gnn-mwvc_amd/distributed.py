"""1-D vertex-partitioned forward across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  Rank r holds the CSR SLICE of its rows
[lo_r, hi_r) — row pointers relative to the slice, global column ids, its slice of
W / NW (`slice_csr`, `Engine.attach_graph_slice`): 1 / P of the graph's memory —
and full-size, replicated feature buffers.  It computes those rows of each fused
stage, then the ranks exchange their freshly written rows so that every rank holds
the complete input of the next stage — an all-gather of the N x 16 fp32 feature
matrix after stages 0 and 1 and of the N scores after the last stage.  Each row is
still summed on one GPU in CSR order, so results are bit-identical to the
single-GPU forward.  (A caller that keeps the WHOLE graph on every rank may also
have early stages computed in full everywhere instead of exchanged — `replicate`;
with slices every stage is partitioned.)

There is no reference counterpart (the reference is single-process); the
arithmetic is `gnn::model::predict` (reference src/gnn_inference.cpp:67-81)
unchanged.

Compressed exchange: after the ReLU that ends a stage most of the N x 16 feature matrix is
zero — on the metric graph two columns are non-zero in every row, two in about a tenth of the
rows, five in a handful of rows and seven in none (which, and how dense, depends on the graph).
Once a forward has counted the non-zeros per column, later forwards on the same graph ship each
row as its 4, 8 or 12 densest columns plus a short exception list for non-zeros anywhere else,
and expand on arrival: a quarter of the bytes on the metric graph.  It stays lossless: the pack
step routes every non-zero it does not ship densely into the list, flags a list that overflows,
and a forward whose flag is raised is repeated with full rows
(`partitioned_forward(..., verify=True)`, the default).

The stage executor is injected: on a GPU it is `Engine.stage_forward_device`
(HIP kernels); the CPU tests inject a checker-backed one.  This module itself
never computes a stage and never imports the oracle.
"""
from __future__ import annotations

import dataclasses
from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist

ALIGN = 64  # a wave tile; keeps every rank's tiles full except the last one's tail


def partition_bounds(n: int, world: int, rowptr=None, mode: str = "rows") -> List[Tuple[int, int]]:
    """Contiguous row ranges [lo, hi) per rank, boundaries multiples of ALIGN.

    mode "rows": equal row counts (shards equal-sized -> one all_gather_into_tensor).
    mode "nnz":  equal CSR entries per rank (prefix-sum split of rowptr), for skewed
                 graphs; shards become uneven -> exchanged by direct sends.
    """
    if world <= 0:
        raise ValueError("world must be positive")
    if mode == "rows" or rowptr is None or n == 0:
        per = ((n + world - 1) // world + ALIGN - 1) // ALIGN * ALIGN if n else 0
        cuts = [min(r * per, n) for r in range(world + 1)]
    elif mode == "nnz":
        rp = torch.as_tensor(rowptr).to(torch.int64).cpu()
        nnz = int(rp[n])
        cuts = [0]
        for r in range(1, world):
            target = nnz * r // world
            v = int(torch.searchsorted(rp, torch.tensor([target], dtype=torch.int64)).item())
            v = min(max((v + ALIGN // 2) // ALIGN * ALIGN, cuts[-1]), n)
            cuts.append(v)
        cuts.append(n)
    else:
        raise ValueError(f"unknown partition mode {mode!r}")
    cuts[-1] = n
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


@dataclasses.dataclass
class CsrSlice:
    """Rows [lo, hi) of a CSR graph of n vertices: what one rank of a vertex-partitioned run keeps."""
    n: int                 # vertices of the whole graph (column ids are global)
    lo: int
    hi: int
    rowptr: torch.Tensor   # (hi - lo + 1,) relative to the slice: rowptr[0] = 0, rowptr[-1] = nnz
    col: torch.Tensor      # (nnz + pad,) global column ids, stored order
    w: torch.Tensor        # (hi - lo,)
    nw: torch.Tensor       # (hi - lo,)
    nnz: int

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.rowptr, self.col, self.w, self.nw))


def slice_csr(n: int, rowptr: torch.Tensor, col: torch.Tensor, w: torch.Tensor, nw: torch.Tensor, lo: int, hi: int,
              pad: int = 64) -> CsrSlice:
    """Cut rows [lo, hi) out of a whole CSR (tensors on any device; the slice owns fresh storage, so the whole
    graph can be released afterwards).  `pad` extra column entries follow the slice's last one (the engine's
    kernels read a few entries past a tile's end, GNNVC_COL_PAD)."""
    if not 0 <= lo <= hi <= n:
        raise ValueError(f"rows [{lo}, {hi}) outside a graph of {n} vertices")
    e0, e1 = int(rowptr[lo]), int(rowptr[hi])
    rp = (rowptr[lo: hi + 1].to(torch.int64) - e0).to(rowptr.dtype).contiguous()
    c = torch.zeros(e1 - e0 + pad, dtype=col.dtype, device=col.device)
    c[: e1 - e0] = col[e0:e1]
    return CsrSlice(n, lo, hi, rp, c, w[lo:hi].clone().contiguous(), nw[lo:hi].clone().contiguous(), e1 - e0)


@dataclasses.dataclass(frozen=True)
class Packing:
    """How one stage's output rows travel: `mask` = the columns shipped densely (kp floats per row,
    zero padded), `cap` = exception-list entries per piece for non-zeros in any other column."""
    mask: int
    kp: int
    cap: int
    bytes_per_row: float   # expected, exceptions included (a full row is 64)

    def piece_words(self, dense_rows: int) -> int:
        return dense_rows * self.kp + 4 + 4 * self.cap


EXC_BYTES = 16   # one exception entry: {row, column, value, pad}


def choose_packing(counts: Sequence[int], n: int, pieces: int) -> "Packing | None":
    """Cheapest of kp = 4, 8, 12 dense columns (+ exceptions for the rest) from the per-column
    non-zero counts of a stage's N x 16 output; None if nothing beats three quarters of a full row.
    `pieces` = how many pieces (ranks x pipeline chunks) the rows travel in, for the list size."""
    if n <= 0:
        return None
    order = sorted(range(16), key=lambda c: (-int(counts[c]), c))
    best = None
    for kp in (4, 8, 12):
        dense = order[:kp]
        rest = sum(int(counts[c]) for c in order[kp:])
        cost = 4.0 * kp + EXC_BYTES * rest / n
        if best is None or cost < best[0]:
            best = (cost, kp, dense, rest)
    cost, kp, dense, rest = best
    if cost > 48.0:
        return None
    mask = sum(1 << c for c in dense if counts[c] > 0)
    cap = int(2 * rest / max(pieces, 1)) + 1024     # twice the expected share of a piece, plus slack
    return Packing(mask, kp, cap, cost)


class RowCodec:
    """What the compressed exchange needs from the device side (gnn_mwvc_amd.EngineRowCodec on a
    GPU; the CPU tests inject a torch one).  A piece `region` is a flat fp32 tensor of
    Packing.piece_words(dense_rows) words: dense_rows x kp floats, then the exception list."""

    def column_counts(self, feat: torch.Tensor, n: int) -> List[int]:   # non-zeros per column over rows < n
        raise NotImplementedError

    def pack(self, feat, lo: int, hi: int, pk: Packing, region, dense_rows: int, flag) -> None:
        raise NotImplementedError

    def unpack(self, region, dense_rows: int, lo: int, hi: int, pk: Packing, feat) -> None:
        raise NotImplementedError


@dataclasses.dataclass
class ForwardBuffers:
    """Full-size buffers every rank owns.  Feature buffers have `rows_alloc` rows;
    row n (the gather's pad row) and everything after it stay zero."""
    n: int
    rows_alloc: int
    feat: List[torch.Tensor]   # two (rows_alloc, 16) fp32 ping-pong buffers
    scores: torch.Tensor       # (rows_alloc,) fp32
    logits: torch.Tensor       # (rows_alloc,) fp32
    # compressed exchange
    packed: "torch.Tensor | None" = None       # flat fp32 staging for the pieces (grown on demand)
    flag: "torch.Tensor | None" = None         # int32[1]: a pack step could not ship a row losslessly
    live: "dict[int, Packing | None]" = dataclasses.field(default_factory=dict)   # stage -> packing of its output (None = full rows)

    @staticmethod
    def allocate(n: int, bounds: Sequence[Tuple[int, int]], device) -> "ForwardBuffers":
        world = len(bounds)
        equal = _equal_shard_rows(n, bounds)
        rows = (world * equal if equal else n) + ALIGN
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=device)
        return ForwardBuffers(n, rows, [z(rows, 16), z(rows, 16)], z(rows), z(rows))

    def staging(self, words: int) -> torch.Tensor:
        dev = self.feat[0].device
        if self.flag is None:
            self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        if self.packed is None or self.packed.numel() < words:
            self.packed = torch.zeros(words, dtype=torch.float32, device=dev)
        return self.packed

    def forget_live_columns(self) -> None:
        """Call when the graph (or the input) behind these buffers changes."""
        self.live.clear()


def _equal_shard_rows(n: int, bounds: Sequence[Tuple[int, int]]) -> int:
    """Rows per shard if the partition is the equal-rows one (last shard may be short), else 0."""
    world = len(bounds)
    per = bounds[0][1] - bounds[0][0]
    if per == 0 or per % ALIGN:
        return 0 if world > 1 else max(per, 0)
    for r, (lo, hi) in enumerate(bounds):
        if lo != min(r * per, n) or hi != min((r + 1) * per, n):
            return 0
    return per


def _before_comm(t: torch.Tensor, group=None) -> None:
    """RCCL collectives are ordered on the stream behind the kernels that produced `t`.  gloo is not: handed a device
    tensor (the one-GPU rehearsal, `bench.py --backend gloo`) it reads the memory from the host right away, so the
    stream has to be drained first — otherwise it ships rows a pack or stage kernel is still writing."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        torch.cuda.current_stream(t.device).synchronize()


def exchange_rows(buf: torch.Tensor, bounds: Sequence[Tuple[int, int]], rank: int, n: int,
                  group=None, method: str = "auto") -> None:
    """Make rows [lo_r, hi_r) written by each rank r visible on every rank, in place.
    `buf` is indexed by row along dim 0 (a (rows, 16) feature buffer, a (rows,) score vector or a
    (rows, kp) view of the packed buffer).

    "allgather": equal shards -> one in-place all_gather_into_tensor (RCCL picks the
                 xGMI schedule).
    "p2p":       any shard sizes -> every rank sends its shard straight to each peer
                 and receives theirs, all transfers in one group (full-mesh: one
                 xGMI link per peer, no ring).
    """
    world = len(bounds)
    if world == 1:
        return
    per = _equal_shard_rows(n, bounds)
    if method == "auto":
        method = "allgather" if per else "p2p"
    _before_comm(buf, group)
    if method == "allgather":
        if not per:
            raise ValueError("all_gather_into_tensor needs the equal-rows partition")
        flat = buf[: world * per]
        dist.all_gather_into_tensor(flat, flat[rank * per:(rank + 1) * per], group=group)
        return
    if method != "p2p":
        raise ValueError(f"unknown exchange method {method!r}")
    lo, hi = bounds[rank]
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        plo, phi = bounds[peer]
        if hi > lo:
            ops.append(dist.P2POp(dist.isend, buf[lo:hi], peer, group))
        if phi > plo:
            ops.append(dist.P2POp(dist.irecv, buf[plo:phi], peer, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


@dataclasses.dataclass
class HaloPlan:
    """Which rows cross between which ranks when only REFERENCED rows travel (north_star: "boundary-vertex feature rows").
    Rank r's slice references a set of rows (the column ids of its entries); of a peer's shard it needs exactly the referenced
    ones.  send_idx[q] = the rows of MY shard that peer q's slice references, recv_idx[q] = the rows of q's shard mine does
    (global row ids, ascending, on the buffers' device).  On an Erdős–Rényi graph a rank references ~92 % of all rows and the
    all-gather stays (use_halo False); on a graph with locality (a banded / mesh-like graph cut into contiguous ranges) the
    halo is a sliver of the shards and the exchange's bytes go with it."""
    send_idx: List[torch.Tensor]
    recv_idx: List[torch.Tensor]
    halo_rows_in: int          # rows this rank receives per exchange
    halo_rows_out: int         # rows it sends (summed over peers)
    full_rows_in: int          # what the all-gather would deliver to it: every row of every other shard
    use_halo: bool

    def bytes_in(self, row_bytes: int = 64) -> int:
        return self.halo_rows_in * row_bytes


def build_halo_plan(col: torch.Tensor, nnz: int, bounds: Sequence[Tuple[int, int]], rank: int, n: int, group=None,
                    max_fraction: float = 0.5) -> HaloPlan:
    """Once per graph: every rank finds the rows its slice references (col = its slice's GLOBAL column ids, first nnz entries),
    tells every peer which of that peer's rows it needs, and learns what the peers need of its own.  use_halo = the busiest
    rank would receive at most `max_fraction` of what the all-gather delivers (every rank reaches the same verdict)."""
    world = len(bounds)
    dev = col.device
    empty = torch.zeros(0, dtype=torch.int64, device=dev)
    if world == 1:
        return HaloPlan([empty], [empty], 0, 0, 0, False)
    need = torch.unique(col[:nnz].to(torch.int64)) if nnz else empty
    recv_idx = []
    for q, (plo, phi) in enumerate(bounds):
        recv_idx.append(empty if q == rank else need[(need >= plo) & (need < phi)].contiguous())
    counts = torch.tensor([int(t.numel()) for t in recv_idx], dtype=torch.int64, device=dev)
    table = torch.zeros(world * world, dtype=torch.int64, device=dev)      # table[r * world + q] = rows r needs from q
    _before_comm(counts, group)
    dist.all_gather_into_tensor(table, counts, group=group)
    table = table.view(world, world).cpu()
    send_idx = [empty if q == rank else torch.zeros(int(table[q, rank]), dtype=torch.int64, device=dev) for q in range(world)]
    ops = []
    for q in range(world):
        if q == rank:
            continue
        if recv_idx[q].numel():
            ops.append(dist.P2POp(dist.isend, recv_idx[q], q, group))
        if send_idx[q].numel():
            ops.append(dist.P2POp(dist.irecv, send_idx[q], q, group))
    if ops:
        _before_comm(counts, group)
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    lo, hi = bounds[rank]
    for q in range(world):      # never trust a received index with a gather: it has to name one of MY rows
        if q != rank and send_idx[q].numel() and (int(send_idx[q].min()) < lo or int(send_idx[q].max()) >= hi):
            raise ValueError(f"rank {q} asked rank {rank} for rows outside [{lo}, {hi})")
    rows_in = table.sum(dim=1)                                              # per rank: what it receives
    full_in = torch.tensor([n - (phi - plo) for plo, phi in bounds], dtype=torch.int64)
    worst = float((rows_in.to(torch.float64) / full_in.clamp(min=1).to(torch.float64)).max())
    return HaloPlan(send_idx, recv_idx, int(rows_in[rank]), int(table[:, rank].sum()), int(full_in[rank]), worst <= max_fraction)


def exchange_rows_halo(buf: torch.Tensor, plan: HaloPlan, rank: int, group=None) -> int:
    """exchange_rows for callers that only need the rows their slice references: every rank sends each peer the rows of its
    shard that peer references (gathered into one contiguous message per peer) and scatters what it receives into `buf`.
    Rows nobody here references keep whatever they held — no kernel of this rank reads them.  Returns the bytes received."""
    world = len(plan.send_idx)
    if world == 1:
        return 0
    outs = [buf.index_select(0, plan.send_idx[q]) if (q != rank and plan.send_idx[q].numel()) else None for q in range(world)]
    ins = [torch.empty((int(plan.recv_idx[q].numel()),) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
           if (q != rank and plan.recv_idx[q].numel()) else None for q in range(world)]
    ops = []
    for q in range(world):
        if outs[q] is not None:
            ops.append(dist.P2POp(dist.isend, outs[q], q, group))
        if ins[q] is not None:
            ops.append(dist.P2POp(dist.irecv, ins[q], q, group))
    got = 0
    if ops:
        _before_comm(buf, group)
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for q in range(world):
        if ins[q] is not None:
            buf.index_copy_(0, plan.recv_idx[q], ins[q])
            got += ins[q].numel() * ins[q].element_size()
    return got


def _piece_step(per: int, chunks: int, piece_rows: int) -> int:
    """Rows per piece: `piece_rows` when the caller knows a natural size (a multiple of ALIGN — e.g. 256 chunks of
    the engine's compact-table plan, so that no piece ends inside a chunk), else the shard cut into `chunks`."""
    if piece_rows > 0:
        if piece_rows % ALIGN:
            raise ValueError("piece_rows must be a multiple of ALIGN")
        return piece_rows
    return max(ALIGN, (per // max(chunks, 1) + ALIGN - 1) // ALIGN * ALIGN)


def exchange_rows_pipelined(stage_fn, stage: int, src: torch.Tensor, dst: torch.Tensor, logits,
                            bounds: Sequence[Tuple[int, int]], rank: int, n: int, chunks: int,
                            group=None, piece_rows: int = 0) -> None:
    """Compute this rank's rows of `stage` in `chunks` pieces and all-gather each piece
    asynchronously while the next piece is being computed (equal-rows partition only).

    The collective of piece k is issued right after piece k's kernels were queued: the
    communication stream waits for them through the event torch records at the call, while
    this stream goes on launching piece k + 1 — RCCL's copies and the gather kernels overlap.
    All pieces are waited for before returning (the next stage needs every row).
    """
    world = len(bounds)
    per = _equal_shard_rows(n, bounds)
    lo, hi = bounds[rank]
    step = _piece_step(per, chunks, piece_rows)
    works = []
    for off in range(0, per, step):
        size = min(step, per - off)
        r0, r1 = min(lo + off, hi), min(lo + off + size, hi)
        if r1 > r0:
            stage_fn(stage, r0, r1, src, dst, logits)
        # piece `off` of every rank's (padded) shard: equal sizes, rows past n stay zero
        outs = [dst[r * per + off: r * per + off + size] for r in range(world)]
        _before_comm(dst, group)
        works.append(dist.all_gather(outs, outs[rank], group=group, async_op=True))
    for w in works:
        w.wait()


def exchange_rows_packed(codec: RowCodec, bufs: ForwardBuffers, dst: torch.Tensor, pk: Packing,
                         bounds: Sequence[Tuple[int, int]], rank: int, group=None, method: str = "auto") -> None:
    """exchange_rows for a feature buffer that travels as `pk`: pack this rank's rows into its
    region, exchange the regions, expand every peer's region into `dst`."""
    world = len(bounds)
    if world == 1:
        return
    per = _equal_shard_rows(bufs.n, bounds)
    if method == "auto":
        method = "allgather" if per else "p2p"
    lo, hi = bounds[rank]
    if method == "allgather":
        if not per:
            raise ValueError("all_gather_into_tensor needs the equal-rows partition")
        rows = [per] * world                      # every region the same size (the last shard may be short)
    else:
        rows = [phi - plo for plo, phi in bounds]
    words = [pk.piece_words(r) for r in rows]
    start = [sum(words[:r]) for r in range(world + 1)]
    buf = bufs.staging(start[-1])
    region = lambda r: buf[start[r]: start[r + 1]]
    codec.pack(dst, lo, hi, pk, region(rank), rows[rank], bufs.flag)
    _before_comm(buf, group)
    if method == "allgather":
        dist.all_gather_into_tensor(buf[: start[-1]], region(rank), group=group)
    else:
        ops = []
        for peer in range(world):
            if peer != rank:
                ops.append(dist.P2POp(dist.isend, region(rank), peer, group))
                ops.append(dist.P2POp(dist.irecv, region(peer), peer, group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if method == "allgather" and hasattr(codec, "unpack_gathered"):
        codec.unpack_gathered(buf[: start[-1]], world, rank, per, per, 0, per, bufs.n, pk, dst)
        return
    for peer, (plo, phi) in enumerate(bounds):
        if peer != rank and phi > plo:
            codec.unpack(region(peer), rows[peer], plo, phi, pk, dst)


def exchange_rows_pipelined_packed(codec: RowCodec, stage_fn, stage: int, src: torch.Tensor, dst: torch.Tensor,
                                   bufs: ForwardBuffers, pk: Packing, bounds: Sequence[Tuple[int, int]], rank: int,
                                   chunks: int, group=None, piece_rows: int = 0) -> None:
    """exchange_rows_pipelined with packed rows: piece k is computed, packed and its all-gather
    started; while it travels, piece k - 1 is waited for and expanded and piece k + 1 computed."""
    world = len(bounds)
    per = _equal_shard_rows(bufs.n, bounds)
    lo, hi = bounds[rank]
    step = _piece_step(per, chunks, piece_rows)
    pw = pk.piece_words(step)
    n_pieces = (per + step - 1) // step
    buf = bufs.staging(n_pieces * world * pw)      # [piece][rank][pw]: one contiguous gather per piece

    def finish(work, k, off, size):
        work.wait()
        if hasattr(codec, "unpack_gathered"):     # all peers' regions of the piece in one launch
            codec.unpack_gathered(buf[k * world * pw: (k + 1) * world * pw], world, rank, step, per, off, size, bufs.n,
                                  pk, dst)
            return
        for peer, (plo, phi) in enumerate(bounds):
            r0, r1 = min(plo + off, phi), min(plo + off + size, phi)
            if peer != rank and r1 > r0:
                codec.unpack(buf[(k * world + peer) * pw: (k * world + peer + 1) * pw], step, r0, r1, pk, dst)

    pending = None
    for k, off in enumerate(range(0, per, step)):
        size = min(step, per - off)
        r0, r1 = min(lo + off, hi), min(lo + off + size, hi)
        if r1 > r0:
            stage_fn(stage, r0, r1, src, dst, None)
        mine = buf[(k * world + rank) * pw: (k * world + rank + 1) * pw]
        codec.pack(dst, r0, r1, pk, mine, step, bufs.flag)       # an empty range still clears the list header
        _before_comm(buf, group)
        work = dist.all_gather_into_tensor(buf[k * world * pw: (k + 1) * world * pw], mine, group=group, async_op=True)
        if pending is not None:
            finish(*pending)
        pending = (work, k, off, size)
    if pending is not None:
        finish(*pending)


StageFn = Callable[[int, int, int, torch.Tensor, torch.Tensor, "torch.Tensor | None"], None]


def replicated_stages(world: int, num_stages: int = 3) -> "set[int]":
    """Stages every rank computes in full instead of partitioning + exchanging.

    Stage s may be replicated when every earlier stage is (its input is then complete on every
    rank).  Replicating trades (1 - 1/P) of the stage's compute for one N x 16 all-gather
    (640 MB on the metric graph).  With P ranks a rank's share of that gather crosses P - 1
    xGMI links of ~75 GB/s per direction: ~4.2 ms at P = 2, ~2.1 ms at P = 4, ~1 ms at P = 8,
    against ~2 ms (stage 0) and ~3.8 ms (stage 1) of single-GPU compute:
      P = 2: replicate stages 0 and 1 (only the last stage is partitioned, only scores travel);
      P = 3, 4: replicate stage 0;  P >= 5: partition everything.
    The last stage is never replicated (it would leave nothing to share)."""
    if world <= 1:
        return set()
    if world == 2:
        return set(range(min(2, num_stages - 1)))
    if world <= 4:
        return {0} if num_stages > 1 else set()
    return set()


def plan_replication(world: int, num_stages: int, live: "dict | None" = None) -> "set[int]":
    """replicated_stages(), revised once the packings are known: when every exchanged stage travels in about
    half the bytes of full rows or less, the gathers of the 16-wide stages are cheaper than replicated compute and
    those stages are partitioned.  Stage 0 stays replicated up to 4 ranks: computed in full it takes ~1.1 ms on
    the metric graph (its input x is replicated anyway, and a whole-range call runs the LDS-table plan), which is
    less than a rank's share in pipelined pieces plus the first exchange (measured per-rank compute of the pieces
    alone: 1.43 ms at P = 2, 0.91 ms at P = 4, 0.61 ms at P = 8 — scratch/experiments/pieces_check.py)."""
    if live is not None and all(live.get(st) is not None and live[st].bytes_per_row <= 34.0
                                for st in range(num_stages - 1)):
        return {0} if (1 < world <= 4 and num_stages > 1) else set()
    return replicated_stages(world, num_stages)


def replicate_first_stage(world: int) -> bool:   # kept for callers that only ask about stage 0
    return 0 in replicated_stages(world)


def partitioned_forward(stage_fn: StageFn, num_stages: int, x: torch.Tensor, bufs: ForwardBuffers,
                        bounds: Sequence[Tuple[int, int]], rank: int, group=None,
                        exchange: str = "auto", on_stage=None, gather_logits: bool = True,
                        replicate_stage0: "bool | None" = None,
                        pipeline_chunks: int = 0,
                        replicate: "set[int] | None" = None,
                        codec: "RowCodec | None" = None,
                        verify: bool = True,
                        prepare_fn=None,
                        piece_rows: int = 0,
                        halo: "HaloPlan | None" = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Run every fused stage on this rank's rows and exchange between stages.

    stage_fn(stage, lo, hi, src, dst, logits_or_None) computes rows [lo, hi) of `dst`
    from the full `src`.  Returns (scores[:n], logits[:n]) — complete on every rank,
    like `predict` hands every caller all N scores.
    replicate: stages computed in full on every rank (None = replicated_stages(world));
    replicate_stage0 (legacy switch): True/False forces {0} / {}.
    pipeline_chunks > 1: overlap each feature exchange with the stage's own compute
    (exchange_rows_pipelined; equal-rows partition only).
    on_stage(stage, phase) is an optional hook ("begin" | "computed" | "exchanged")
    used by the bench to drop timing events on the stream.
    codec: enables the compressed exchange.  The first forward on a graph exchanges full rows and
    settles each stage's Packing in bufs.live from its per-column non-zero counts; later forwards
    ship packed rows.  With verify=True (default) the forward ends with exchange_verified(): if any
    rank's exception list overflowed, every rank goes back to full rows and the forward is repeated.
    halo: a HaloPlan with use_halo set (build_halo_plan) — the feature exchanges then ship only the rows the receiving rank's
    slice references (full rows, one message per peer; no packing, no pieces: graphs with locality move little anyway).  The
    scores still go to every rank.
    prepare_fn(stage, src, lo, hi): optional hint called once before a partitioned stage >= 1 — "src is this
    stage's complete input and stays as it is while rows [lo, hi) are computed from it, possibly in pieces"
    (Engine.stage_input_ready: the engine writes its compact table once instead of gathering full rows per piece).  Callers that pass verify=False (a timed loop) must call
    exchange_verified() themselves before trusting the results.
    """
    world = len(bounds)
    lo, hi = bounds[rank]
    if replicate is None:
        replicate = plan_replication(world, num_stages, bufs.live if codec is not None else None) \
            if replicate_stage0 is None else ({0} if replicate_stage0 and num_stages > 1 else set())
    # a stage can only be computed in full if its input is complete everywhere
    rep, ok = set(), True
    for st in range(num_stages - 1):
        ok = ok and st in replicate
        if ok:
            rep.add(st)
    can_pipeline = pipeline_chunks > 1 and world > 1 and _equal_shard_rows(bufs.n, bounds) > 0
    used_codec = False
    pieces = world * (max(pipeline_chunks, 1) if can_pipeline else 1)
    if can_pipeline and piece_rows > 0:
        pieces = world * max(1, -(-_equal_shard_rows(bufs.n, bounds) // piece_rows))
    src = x
    for st in range(num_stages):
        last = st == num_stages - 1
        dst = bufs.scores if last else bufs.feat[st & 1]
        if on_stage:
            on_stage(st, "begin")
        if st in rep and not last:
            stage_fn(st, 0, bufs.n, src, dst, None)       # every row, no exchange
            if on_stage:
                on_stage(st, "computed")
            if codec is not None and world > 1 and st not in bufs.live:
                bufs.live[st] = choose_packing(codec.column_counts(dst, bufs.n), bufs.n, pieces)
        elif not last:
            if prepare_fn is not None and st >= 1:
                prepare_fn(st, src, lo, hi)
            use_halo = halo is not None and halo.use_halo and world > 1
            pk = bufs.live.get(st) if (codec is not None and world > 1 and not use_halo) else None
            pack = pk is not None
            used_codec = used_codec or pack
            if use_halo:
                stage_fn(st, lo, hi, src, dst, None)
                if on_stage:
                    on_stage(st, "computed")
                exchange_rows_halo(dst, halo, rank, group)
            elif can_pipeline:
                if pack:
                    exchange_rows_pipelined_packed(codec, stage_fn, st, src, dst, bufs, pk, bounds, rank,
                                                   pipeline_chunks, group, piece_rows)
                else:
                    exchange_rows_pipelined(stage_fn, st, src, dst, None, bounds, rank, bufs.n,
                                            pipeline_chunks, group, piece_rows)
                if on_stage:
                    on_stage(st, "computed")
            else:
                stage_fn(st, lo, hi, src, dst, None)
                if on_stage:
                    on_stage(st, "computed")
                if pack:
                    exchange_rows_packed(codec, bufs, dst, pk, bounds, rank, group, exchange)
                else:
                    exchange_rows(dst, bounds, rank, bufs.n, group, exchange)
            if codec is not None and world > 1 and st not in bufs.live and not use_halo:
                # every rank now holds the complete, identical dst: count its non-zeros per column
                # (synchronises; first forward on a graph only) and settle how it travels from now on
                bufs.live[st] = choose_packing(codec.column_counts(dst, bufs.n), bufs.n, pieces)
        else:
            if prepare_fn is not None and st >= 1:
                prepare_fn(st, src, lo, hi)
            stage_fn(st, lo, hi, src, dst, bufs.logits)
            if on_stage:
                on_stage(st, "computed")
            exchange_rows(dst, bounds, rank, bufs.n, group, exchange)
            if gather_logits:   # the exact-parity route applies the host sigmoid to the logits
                exchange_rows(bufs.logits, bounds, rank, bufs.n, group, exchange)
        if on_stage:
            on_stage(st, "exchanged")
        src = dst
    if used_codec and verify and not exchange_verified(bufs, group):
        return partitioned_forward(stage_fn, num_stages, x, bufs, bounds, rank, group, exchange, on_stage,
                                   gather_logits, replicate_stage0, pipeline_chunks, replicate, codec, verify, prepare_fn,
                                   piece_rows, halo)
    return bufs.scores[: bufs.n], bufs.logits[: bufs.n]


def exchange_verified(bufs: ForwardBuffers, group=None) -> bool:
    """True if every pack step since the last call shipped its rows losslessly.  Otherwise every rank
    (they all see the same reduced flag) stops packing for these buffers — full rows from now on —
    and the caller repeats the forward."""
    if bufs.flag is None:
        return True
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(bufs.flag, op=dist.ReduceOp.MAX, group=group)
    if int(bufs.flag.item()) == 0:
        return True
    bufs.flag.zero_()
    for st in list(bufs.live):
        bufs.live[st] = None
    return False
