"""world_size > 1 on CPU (gloo): the 1-D vertex partition, the buffer layout and the
inter-stage exchange of gnn-mwvc_amd/distributed.py.  The stage arithmetic is
injected from the oracle here (no GPU in this container); on the GPU box the same
driver runs with Engine.stage_forward_device."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gnn_mwvc_amd  # noqa: F401  (import shim)
from gnn_mwvc_amd import distributed as D
from tools import graphgen as gg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_stage_fn(g, text):
    """stage_fn computing one fused stage with the oracle's layer functions."""
    from oracle import oracle_py
    om = oracle_py.OracleModel(text)
    params = om.linear_params()

    def stage_fn(stage, lo, hi, src, dst, logits):
        h = src[: g.n].numpy().reshape(g.n, -1)
        a = oracle_py.graph_layer(g, g.ws, h)
        for i, (W, b) in enumerate(params[3 * stage: 3 * stage + 3]):
            a = oracle_py.linear_layer(a, W, b)
            if stage == 2 and i == 2:
                if logits is not None:
                    logits[lo:hi] = torch.from_numpy(a[lo:hi, 0].copy())
                a = oracle_py.sigmoid(a)
            else:
                a = oracle_py.relu(a)
        out = torch.from_numpy(np.ascontiguousarray(a[lo:hi]))
        if dst.dim() == 1:
            dst[lo:hi] = out[:, 0]
        else:
            dst[lo:hi] = out
    return stage_fn


def _sliced_oracle_stage_fn(sl, ws, text):
    """stage_fn of a rank that holds only its CSR slice (D.slice_csr): rows outside [lo, hi) have no adjacency
    here, and asking for them is an error — like Engine.stage_forward_device on a sliced engine."""
    from oracle import oracle_py
    om = oracle_py.OracleModel(text)
    params = om.linear_params()
    n, lo, hi = sl.n, sl.lo, sl.hi
    rowptr = np.zeros(n + 1, dtype=np.uint64)
    rowptr[lo: hi + 1] = sl.rowptr.numpy().astype(np.uint64)
    rowptr[hi + 1:] = sl.nnz
    w = np.zeros(n, dtype=np.uint32)
    nw = np.zeros(n, dtype=np.uint32)
    w[lo:hi] = sl.w.numpy()
    nw[lo:hi] = sl.nw.numpy()
    local = gg.CsrGraph(n, rowptr, sl.col.numpy()[: sl.nnz].astype(np.uint32), w, nw)

    def stage_fn(stage, r0, r1, src, dst, logits):
        assert lo <= r0 <= r1 <= hi, f"rows [{r0}, {r1}) are not in this rank's slice [{lo}, {hi})"
        h = src[:n].numpy().reshape(n, -1)
        a = oracle_py.graph_layer(local, ws, h)[r0:r1]
        for i, (W, b) in enumerate(params[3 * stage: 3 * stage + 3]):
            a = oracle_py.linear_layer(a, W, b)
            if stage == 2 and i == 2:
                if logits is not None:
                    logits[r0:r1] = torch.from_numpy(a[:, 0].copy())
                a = oracle_py.sigmoid(a)
            else:
                a = oracle_py.relu(a)
        out = torch.from_numpy(np.ascontiguousarray(a))
        if dst.dim() == 1:
            dst[r0:r1] = out[:, 0]
        else:
            dst[r0:r1] = out
    return stage_fn


class TorchRowCodec(D.RowCodec):
    """CPU stand-in (tests only) for gnn_mwvc_amd.EngineRowCodec: same contract, torch indexing."""

    @staticmethod
    def _split(region, dense_rows, pk):
        dense = region[: dense_rows * pk.kp].view(dense_rows, pk.kp)
        exc = region[dense_rows * pk.kp: pk.piece_words(dense_rows)].view(torch.int32)
        return dense, exc

    def column_counts(self, feat, n):
        return [int((feat[:n, c] != 0).sum()) for c in range(16)]

    def pack(self, feat, lo, hi, pk, region, dense_rows, flag):
        dense, exc = self._split(region, dense_rows, pk)
        cols = [c for c in range(16) if pk.mask >> c & 1]
        other = [c for c in range(16) if not pk.mask >> c & 1]
        exc[:4] = 0
        rows = hi - lo
        if rows <= 0:
            return
        dense[:rows] = 0
        if cols:
            dense[:rows, : len(cols)] = feat[lo:hi][:, cols]
        if other:
            sub = feat[lo:hi][:, other].contiguous()
            nz = (sub != 0).nonzero()
            exc[0] = len(nz)
            m = min(len(nz), pk.cap)
            if m:
                en = exc[4: 4 + 4 * pk.cap].view(-1, 4)
                en[:m, 0] = nz[:m, 0].to(torch.int32)
                en[:m, 1] = torch.tensor(other, dtype=torch.int32)[nz[:m, 1]]
                en[:m, 2] = sub[nz[:m, 0], nz[:m, 1]].view(torch.int32)
                en[:m, 3] = 0
            if len(nz) > pk.cap:
                flag |= 2

    def unpack(self, region, dense_rows, lo, hi, pk, feat):
        dense, exc = self._split(region, dense_rows, pk)
        cols = [c for c in range(16) if pk.mask >> c & 1]
        rows = hi - lo
        feat[lo:hi] = 0
        if cols:
            feat[lo:hi, cols] = dense[:rows, : len(cols)]
        m = min(int(exc[0]), pk.cap)
        en = exc[4: 4 + 4 * pk.cap].view(-1, 4)[:m]
        for rel, col, bits_, _ in en.tolist():
            if rel < rows and col < 16:
                feat[lo + rel, col] = torch.tensor([bits_], dtype=torch.int32).view(torch.float32)[0]


def _codec_worker(rank, world, port, mode, graph_args, chunks, tamper, q):
    """Two forwards with the compressed exchange: the first settles the packings, the second packs."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dataclasses
        import pathlib
        text = (pathlib.Path(__file__).resolve().parent.parent / "gnn-mwvc_amd" / "data" /
                "mwvc_model.txt").read_text()
        g = gg.hub_graph(*graph_args) if len(graph_args) == 5 else gg.erdos_renyi(*graph_args)
        bounds = D.partition_bounds(g.n, world, g.rowptr, mode)
        bufs = D.ForwardBuffers.allocate(g.n, bounds, "cpu")
        x = torch.from_numpy(g.x())
        fn = _oracle_stage_fn(g, text)
        codec = TorchRowCodec()
        kw = dict(replicate=set(), pipeline_chunks=chunks, codec=codec)
        D.partitioned_forward(fn, 3, x, bufs, bounds, rank, **kw)
        learned = dict(bufs.live)
        if tamper and learned[0] is not None:
            # drop the densest column from the dense set: its values must now travel as exceptions ...
            pk = learned[0]
            victim = next(c for c in range(16) if pk.mask >> c & 1)
            big = bufs.n * 16
            bufs.live[0] = dataclasses.replace(pk, mask=pk.mask & ~(1 << victim), cap=big if tamper == "list" else 3)
            # ... and with room for only 3 of them the pack step must raise the flag instead
        for f in bufs.feat:           # stale rows must not be able to pass for exchanged ones
            f[: g.n] = 7.0
        scores, logits = D.partitioned_forward(fn, 3, x, bufs, bounds, rank, **kw)
        pad_ok = all(float(f[g.n:].abs().sum()) == 0.0 for f in bufs.feat)
        q.put((rank, scores.numpy().copy(), logits.numpy().copy(), learned, dict(bufs.live), pad_ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,graph_args,chunks,tamper", [
    (2, "rows", (3000, 30000, 4), 0, ""),
    (2, "rows", (3000, 30000, 4), 3, ""),                 # pipelined pieces, packed
    (3, "nnz", (5000, 20000, 2, 1500, 9), 0, ""),         # uneven shards: packed rows by direct sends
    (2, "rows", (3000, 30000, 4), 0, "list"),             # a dense column demoted: carried by the exception list
    (3, "rows", (3000, 30000, 4), 2, "list"),
    (2, "rows", (3000, 30000, 4), 0, "overflow"),         # list too short -> flagged, forward repeated in full
    (2, "rows", (3000, 30000, 4), 2, "overflow"),
])
def test_compressed_exchange_is_lossless(world, mode, graph_args, chunks, tamper, oracle_model):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_codec_worker, args=(r, world, port, mode, graph_args, chunks, tamper, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = gg.hub_graph(*graph_args) if len(graph_args) == 5 else gg.erdos_renyi(*graph_args)
    oracle_model.set_weight_scale(g.ws)
    want_s, want_l = oracle_model.scores(g), oracle_model.logits(g)
    assert all(r[3] == results[0][3] for r in results)       # every rank settled on the same packings
    learned = results[0][3]
    assert set(learned) == {0, 1}
    assert any(pk is not None for pk in learned.values()), learned   # something was actually packed
    for rank, s, l, _, live_after, pad_ok in results:
        assert pad_ok
        assert np.array_equal(s.view(np.uint32), want_s.view(np.uint32)), f"rank {rank}"
        assert np.array_equal(l.view(np.uint32), want_l.view(np.uint32)), f"rank {rank}"
        if tamper == "overflow":
            assert all(pk is None for pk in live_after.values())   # packing switched off after the hit
        elif not tamper:
            assert live_after == learned


def test_choose_packing():
    n = 1000
    full = [n] * 16
    assert D.choose_packing(full, n, 4) is None                       # nothing to gain
    pk = D.choose_packing([n, 70, 0, 130, 0, 1, 0, 0, 1, 0, 1, n, 0, 0, 0, 0], n, 8)   # the metric graph's stage 0
    assert pk.kp == 4 and pk.mask == (1 << 0 | 1 << 1 | 1 << 3 | 1 << 11) and pk.cap >= 1024
    assert 16.0 <= pk.bytes_per_row < 16.1
    pk = D.choose_packing([n] * 6 + [n // 10] * 2 + [0] * 8, n, 2)    # 6 dense + 2 sparse: 8 dense slots, no exceptions
    assert pk.kp == 8 and bin(pk.mask).count("1") == 8 and pk.bytes_per_row == 32.0
    pk = D.choose_packing([0] * 16, n, 2)
    assert pk.kp == 4 and pk.mask == 0
    assert D.choose_packing([5] * 16, 0, 2) is None


def _worker(rank, world, port, mode, exchange, graph_args, q, replicate=None, chunks=0, sliced=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pathlib
        text = (pathlib.Path(__file__).resolve().parent.parent / "gnn-mwvc_amd" / "data" /
                "mwvc_model.txt").read_text()
        g = gg.hub_graph(*graph_args) if len(graph_args) == 5 else gg.erdos_renyi(*graph_args)
        bounds = D.partition_bounds(g.n, world, g.rowptr, mode)
        bufs = D.ForwardBuffers.allocate(g.n, bounds, "cpu")
        x = torch.from_numpy(g.x())
        if sliced:
            # this rank keeps its rows' slice of the CSR and nothing else of the graph
            lo, hi = bounds[rank]
            sl = D.slice_csr(g.n, torch.from_numpy(g.rowptr.astype(np.int64)), torch.from_numpy(g.col.astype(np.int64)),
                             torch.from_numpy(g.w.astype(np.int64)), torch.from_numpy(g.nw.astype(np.int64)), lo, hi)
            nnz_all = int(g.rowptr[-1])
            assert sl.nnz == int(g.rowptr[hi]) - int(g.rowptr[lo]) and sl.rowptr[0] == 0 and sl.rowptr[-1] == sl.nnz
            assert world == 1 or sl.nnz < nnz_all
            fn, n_ = _sliced_oracle_stage_fn(sl, g.ws, text), g.n
            del g
            g = type("N", (), {"n": n_})()
            scores, logits = D.partitioned_forward(fn, 3, x, bufs, bounds, rank, exchange=exchange, replicate=set(),
                                                   pipeline_chunks=chunks)
        else:
            scores, logits = D.partitioned_forward(_oracle_stage_fn(g, text), 3, x, bufs, bounds, rank,
                                                   exchange=exchange, replicate_stage0=replicate,
                                                   pipeline_chunks=chunks)
        # pad rows of the feature buffers must still be zero (the gather reads row n)
        pad_ok = all(float(f[g.n:].abs().sum()) == 0.0 for f in bufs.feat)
        q.put((rank, scores.numpy().copy(), logits.numpy().copy(), bounds, pad_ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,exchange,graph_args,replicate,chunks,sliced", [
    (2, "rows", "allgather", (3000, 15000, 4), False, 0, False),
    (2, "rows", "p2p", (3000, 15000, 4), None, 0, False),        # whole CSR on every rank: stage 0 replicated at P <= 4
    (3, "nnz", "auto", (5000, 20000, 2, 1500, 9), False, 0, False),  # hub graph: uneven nnz-balanced shards
    (2, "rows", "auto", (100, 300, 5), False, 0, False),         # second rank's shard is short
    (2, "rows", "auto", (3000, 15000, 4), False, 4, False),      # exchanges overlapped with compute, 4 pieces
    (3, "rows", "auto", (1000, 6000, 8), None, 3, False),        # replicated stage 0 + pipelined stage 1
    # every rank holds ONLY its CSR slice: the same bits
    (2, "rows", "auto", (3000, 15000, 4), False, 0, True),
    (3, "nnz", "auto", (5000, 20000, 2, 1500, 9), False, 0, True),   # uneven slices, direct sends
    (2, "rows", "auto", (3000, 15000, 4), False, 3, True),       # sliced + pipelined pieces
    (3, "rows", "auto", (100, 300, 5), False, 0, True),          # a short second slice, an empty third one
])
def test_partitioned_forward_matches_single_process(world, mode, exchange, graph_args, replicate, chunks, sliced,
                                                    oracle_model):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, exchange, graph_args, q, replicate, chunks, sliced))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = gg.hub_graph(*graph_args) if len(graph_args) == 5 else gg.erdos_renyi(*graph_args)
    oracle_model.set_weight_scale(g.ws)
    want_s, want_l = oracle_model.scores(g), oracle_model.logits(g)
    for rank, s, l, bounds, pad_ok in results:
        assert pad_ok
        assert np.array_equal(s.view(np.uint32), want_s.view(np.uint32)), f"rank {rank}"
        assert np.array_equal(l.view(np.uint32), want_l.view(np.uint32)), f"rank {rank}"
        assert bounds[0][0] == 0 and bounds[-1][1] == g.n
        assert all(a[1] == b[0] for a, b in zip(bounds[:-1], bounds[1:]))
        assert all(lo % D.ALIGN == 0 or lo == g.n for lo, _ in bounds)   # (an empty last shard starts at n)


def _banded_graph(n, band, seed):
    """vertex u is adjacent to u +- 1 .. band (a graph with locality: a contiguous range references a sliver of its neighbours')"""
    edges = [(u, u + d) for u in range(n) for d in range(1, band + 1) if u + d < n]
    rng = np.random.default_rng(seed)
    return gg.from_edge_list(n, edges, rng.integers(20, 121, n).tolist())


def _halo_worker(rank, world, port, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pathlib
        text = (pathlib.Path(__file__).resolve().parent.parent / "gnn-mwvc_amd" / "data" / "mwvc_model.txt").read_text()
        g = _banded_graph(4000, 3, 5) if kind == "banded" else gg.erdos_renyi(3000, 30000, 4)
        bounds = D.partition_bounds(g.n, world, g.rowptr, "rows")
        lo, hi = bounds[rank]
        sl = D.slice_csr(g.n, torch.from_numpy(g.rowptr.astype(np.int64)), torch.from_numpy(g.col.astype(np.int64)),
                         torch.from_numpy(g.w.astype(np.int64)), torch.from_numpy(g.nw.astype(np.int64)), lo, hi)
        plan = D.build_halo_plan(sl.col, sl.nnz, bounds, rank, g.n)
        bufs = D.ForwardBuffers.allocate(g.n, bounds, "cpu")
        for f in bufs.feat:          # rows nobody sends this rank must never be read: poison them
            f[: g.n] = float("nan")
        fn = _sliced_oracle_stage_fn(sl, g.ws, text)
        x = torch.from_numpy(g.x())
        scores, logits = D.partitioned_forward(fn, 3, x, bufs, bounds, rank, replicate=set(), halo=plan)
        q.put((rank, scores.numpy().copy(), logits.numpy().copy(), plan.use_halo, plan.halo_rows_in, plan.full_rows_in,
               plan.halo_rows_out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "banded"), (3, "banded"), (2, "er")])
def test_halo_exchange_ships_only_referenced_rows(world, kind, oracle_model):
    """north_star: "boundary-vertex feature rows exchanged".  On a graph with locality (a banded graph cut into contiguous
    ranges) a rank needs a few rows of its neighbours' shards: the halo exchange ships exactly the rows the receiving slice
    references — bytes proportional to the halo, not to N — every other remote row stays poisoned (NaN) here and the scores
    are still the oracle's, bit for bit.  On an Erdős–Rényi graph a rank references nearly every row: the plan says so
    (use_halo False) and the all-gather stays."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = _banded_graph(4000, 3, 5) if kind == "banded" else gg.erdos_renyi(3000, 30000, 4)
    oracle_model.set_weight_scale(g.ws)
    want_s, want_l = oracle_model.scores(g), oracle_model.logits(g)
    for rank, s, l, use_halo, rows_in, full_in, rows_out in results:
        assert np.array_equal(s.view(np.uint32), want_s.view(np.uint32)), f"rank {rank}"
        assert np.array_equal(l.view(np.uint32), want_l.view(np.uint32)), f"rank {rank}"
        if kind == "banded":
            # a rank borders at most two neighbours, 3 rows deep each way
            assert use_halo and 0 < rows_in <= 6 and 0 < rows_out <= 6 and full_in > 1000
        else:
            assert not use_halo and rows_in > 0.8 * full_in


def test_replication_rule():
    assert D.replicated_stages(1) == set()
    assert D.replicated_stages(2) == {0, 1}
    assert D.replicated_stages(4) == {0}
    assert D.replicated_stages(8) == set()
    assert D.replicated_stages(2, num_stages=1) == set()


def test_plan_replication_with_packings():
    good = {0: D.Packing(0x80B, 4, 1024, 16.0), 1: D.Packing(0x603, 4, 1024, 16.0)}
    wide = {0: D.Packing(0xFFF, 12, 1024, 48.0), 1: D.Packing(0x603, 4, 1024, 16.0)}
    for world in (2, 3, 4):
        assert D.plan_replication(world, 3, good) == {0}        # 16-wide stages partitioned, stage 0 in full
        assert D.plan_replication(world, 3, wide) == D.replicated_stages(world)
        assert D.plan_replication(world, 3, {0: good[0]}) == D.replicated_stages(world)   # a stage not measured yet
    assert D.plan_replication(8, 3, good) == set() and D.plan_replication(8, 3, None) == set()
    assert D.plan_replication(1, 3, good) == set()


def test_partition_bounds_properties():
    g = gg.hub_graph(20000, 60000, 3, 4096, seed=7)
    for world in (1, 2, 4, 8):
        for mode in ("rows", "nnz"):
            b = D.partition_bounds(g.n, world, g.rowptr, mode)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == g.n
            assert all(lo <= hi for lo, hi in b)
            assert all(x[1] == y[0] for x, y in zip(b[:-1], b[1:]))
    # nnz mode balances entries better than rows mode on a hub graph
    rp = g.rowptr.astype(np.int64)
    load = lambda bb: max(int(rp[hi] - rp[lo]) for lo, hi in bb)
    assert load(D.partition_bounds(g.n, 4, g.rowptr, "nnz")) <= load(D.partition_bounds(g.n, 4, g.rowptr, "rows"))
    assert D.partition_bounds(0, 4) == [(0, 0)] * 4
