#!/usr/bin/env python3
"""bench.py — GNN forward edges/s on the metric graph (BASELINE.json).

One step = one full forward (3 fused stages: graph layer + dense layers) over the synthetic Erdős–Rényi graph
with 10 M vertices / 100 M undirected edges (weights U[20,120]), device-resident in HBM when the
timed region starts.  N = 1: the engine's whole forward (gnnvc_forward_device).  N > 1: one process per GPU
(torch.distributed, backend nccl = RCCL), the graph 1-D vertex-partitioned — every rank keeps ONLY its rows'
CSR slice (gnnvc_attach_graph_slice) and full-size replicated feature buffers — the feature rows all-gathered
over xGMI between stages (only their live columns once a forward has shown which those are); total work is
fixed, so scaling is "strong".

What the ONE JSON line on rank 0 says (contract in the project brief, read as VERDICT r1 asked):
  value / ms_per_step   the STEADY STATE of a graph that is scored again and again: the engine builds per-graph
                        plans inside a graph's first forwards, the timed region starts after them.
  first_forward_ms      what a caller gets who hands over a fresh graph and scores it once — the reference's own
                        driver does that (src/GNN_VC.cpp:171-192); on graphs of 48 Mi entries and more it builds and
                        uses the 16-wide stages' plan already — plus second / third forward (the second builds the rest
                        of the plans; plan_build_ms = wall time of all builds, the first one running under stage 0's kernels) and
                        plain_forward_ms (steady state, plans switched off).
  roofline              frac = the forward-level fraction of SURVEY.md §8d: (288 E + 300 N) algorithmic bytes /
                        ms_per_step / 8 TB/s.  dominant_kernel: the kernel with the largest share of a forward, its
                        HIP-event time (events on the launch stream, every timed step), its own algorithmic bytes
                        and fraction — which may exceed 1 where a plan serves the gathers from L2 or LDS instead
                        of memory; that is flagged, not hidden.  traffic: measured bytes per forward from the
                        committed PMC summary, only if it was taken from the kernels as they are now (source hash).
  score_once            attach_ms + first_forward_ms = attach_plus_first_forward_ms: everything a caller pays who hands over
                        a graph and scores it once (round 3: the plans one use repays are built at hand-off, the table
                        columns of the first forward come from a pilot), next to the same with plans_at_handoff = 0.
  cpu_baseline          the oracle run the way the reference runs (serial aggregation; products through a discovered
                        OpenBLAS, all cores) ON THE METRIC GRAPH ITSELF, same run (one warm-up + one timed forward, ~50 s;
                        --cpu-sample NxM falls back to a bounded sample of the family), a second labelled line with the
                        row-parallel aggregation its inert OpenMP pragma intended, and the parity checks: EVERY logit of
                        the timed graph, timed configuration, against the oracle (bitwise), every logit against the
                        stage-by-stage path, exact sampled rows.
  workloads             the other configs of BASELINE.json on one GPU (er100k = configs[1], rmat22 = configs[2], rmat24 =
                        configs[3]'s graph unpartitioned, powerlaw1m = configs[4]) and er3m, each on a fresh engine: attach,
                        first forward, steady state (K timed steps after W warm-ups), forward-level fraction, every logit
                        of the steady state AND of a fresh engine's first forward against the oracle.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # Erdős–Rényi: (n, m, seed)
    "er10m": (10_000_000, 100_000_000, 10),   # the metric graph
    "er1m": (1_000_000, 10_000_000, 2),
    "er100k": (100_000, 1_000_000, 1),
    "er20k": (19_675, 98_000, 4),             # the size of the reference CLI's third predict call on ER-100K (src/GNN_VC.cpp:171-192)
    "er3m": (3_000_000, 30_000_000, 3),       # feature table (192 MB) fits the Infinity Cache
    # the other BASELINE.json configs (parity-test cases; selectable here for profiling)
    "rmat24": ("rmat", 24, 16, 24),
    "rmat22": ("rmat", 22, 16, 22),
    "rmat20": ("rmat", 20, 16, 20),
    "powerlaw1m": ("powerlaw", 1_000_000, 16.0, 2.1, 8, 65536, 5),
}


def build_workload(name, ggt, dev):
    spec = WORKLOADS[name]
    if spec[0] == "rmat":
        _, scale, ef, seed = spec
        return ggt.rmat(scale, ef, seed, dev), f"r-mat scale={scale} edge_factor={ef} seed={seed} weights U[20,120]"
    if spec[0] == "powerlaw":
        _, n, deg, expo, hubs, hd, seed = spec
        return (ggt.power_law_hubs(n, deg, expo, hubs, hd, seed, dev),
                f"chung-lu power-law n={n} avg_deg={deg} exp={expo} + {hubs} hubs of degree {hd} seed={seed}")
    n, m, seed = spec
    return ggt.erdos_renyi(n, m, seed, dev), f"erdos-renyi n={n} m={m} seed={seed} weights U[20,120]"


def stage_bytes(stage: int, n: int, nnz: int) -> int:
    """Algorithmic HBM bytes of one fused stage (u32 indices,
    fp32 features, no cache-reuse credit)."""
    if stage == 0:    # nnz*(col 4 + x 4) + N*(rowptr 4 + x 4 + W 4 + NW 4) + N*64 written
        return nnz * 8 + n * 16 + n * 64
    if stage == 1:    # nnz*(4 + 64) + N*(rowptr 4 + own row 64 + W 4 + NW 4) + N*64 written
        return nnz * 68 + n * 76 + n * 64
    return nnz * 68 + n * 76 + n * 4  # last stage writes one score per vertex


def source_hash() -> str:
    """Identity of the kernels a PMC summary was taken from."""
    h = hashlib.sha256()
    for rel in ("gnn-mwvc_amd/csrc/gnnvc_kernels.hip", "gnn-mwvc_amd/csrc/gnnvc_engine.cpp", "gnn-mwvc_amd/csrc/gnnvc_plans.cpp",
                "gnn-mwvc_amd/csrc/gnnvc_engine_state.h", "gnn-mwvc_amd/csrc/exact_sum.h"):
        h.update((ROOT / rel).read_bytes())
    return h.hexdigest()[:16]


def short_kernel(name: str) -> str:
    """'(k_stage_f16<32, 32, 16, false, 2, true, false>)' -> 'k_stage_f16<32,32,16,false,2,true,false>'"""
    return name.strip("()").replace(" ", "")


def kernel_algorithmic_bytes(name: str, n: int, nnz: int, calls_per_forward: float):
    """Algorithmic bytes one forward's launches of `name` stand for (SURVEY.md §8d's per-unit figures x the units
    the kernel processes), or None for kernels that only prepare / decide."""
    if name.startswith("k_c4_agg"):               # the neighbour sums of a 16-wide stage: col id + 64-byte row per entry
        stages = 2 if name.startswith("k_c4_agg<0>") and calls_per_forward >= 2 else 1
        return stages * nnz * 68
    if name.startswith("k_lt_agg"):               # the neighbour sums of the F = 1 stage: col id + 4-byte value per entry
        return nnz * 8
    if name.startswith("k_blk_accumulate"):
        return nnz * 8
    if name.startswith("k_stage_f1<"):            # own values and scalars in, one 64-byte row out (+ the gather when no plan has it)
        return n * 80
    targs = name[name.index("<") + 1:name.rindex(">")].split(",") if "<" in name else []
    # k_stage_f16<N1, N2, N3, SIGMOID, S, MFMA, SORTED, AGGONLY = false, FILTER = false>
    if name.startswith("k_stage_f16<32,32,16") and len(targs) >= 8 and targs[7] == "true":   # aggregate-only: own row, scalars, output row
        return n * (76 + 64)
    if name.startswith("k_stage_f16<32,32,16"):   # gathering feature stage
        return nnz * 68 + n * (76 + 64)
    if name.startswith("k_stage_f16<32,16,1"):    # gathering sigmoid stage
        return nnz * 68 + n * (76 + 4)
    if name.startswith("k_dense_sigmoid"):
        return n * (76 + 4)
    return None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="er10m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline and the oracle parity checks")
    ap.add_argument("--no-variants", action="store_true", help="skip first_forward_ms / plain_forward_ms (N = 1)")
    ap.add_argument("--no-lds-table", action="store_true", help="F = 1 stage without the LDS-table plan")
    ap.add_argument("--no-compact", action="store_true", help="16-wide stages without the compact-table plan")
    ap.add_argument("--no-overlap", action="store_true", help="dense layers after the sums instead of under the next round's")
    ap.add_argument("--cpu-sample", default="",
                    help="CPU baseline on a bounded sample 'NxM' of the workload's family instead of the workload graph itself "
                         "(default: the graph itself — ~50 s of CPU work on the metric graph)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the `workloads` block (er20k, er100k, er3m, rmat22, rmat24, powerlaw1m)")
    ap.add_argument("--workloads", default="er20k,er100k,er3m,rmat22,rmat24,powerlaw1m", help="comma-separated side workloads of the default run")
    ap.add_argument("--no-host-path", action="store_true", help="skip host_path_ms (PCIe-inclusive gnnvc_forward)")
    ap.add_argument("--no-blocked", action="store_true", help="disable the column-blocked F=1 stage (A/B)")
    ap.add_argument("--block-cols", type=int, default=0)
    ap.add_argument("--long-threshold", type=int, default=-1, help="degree at which a row gets its own workgroup")
    ap.add_argument("--giant-threshold", type=int, default=-1, help="degree from which a long row is summed by the parallel scan kernels")
    ap.add_argument("--side-streams", type=int, default=1, help="0 = long / giant rows on the main stream (profiling: standalone kernel times)")
    ap.add_argument("--kernel-trace", type=int, default=1, help="HIP events around every main-stream kernel of the timed forwards")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="any engine option (A/B runs), repeatable")
    ap.add_argument("--sorted-tiles", type=int, default=-1, help="degree-sorted tiles: -1 auto, 0 off, 1 on")
    ap.add_argument("--sorted-long-threshold", type=int, default=0)
    ap.add_argument("--mfma", type=int, default=-1, help="dense layers: 0 VALU, 1 MFMA everywhere, 2 MFMA in the 16-wide stages (default)")
    ap.add_argument("--prepare-input", type=int, default=1,
                    help="N = 2..3: 1 = announce each 16-wide stage's complete input to the engine (compact-table plan "
                         "over the rank's rows), 0 = plain gathering kernels")
    ap.add_argument("--pipeline-chunks", type=int, default=-1,
                    help="N>1: pieces per stage whose all-gather overlaps the next piece's compute (0/1 = off)")
    ap.add_argument("--partition", default="auto", choices=["auto", "rows", "nnz"])
    ap.add_argument("--compress-exchange", type=int, default=1,
                    help="N>1: 1 = ship only the live feature columns between stages (lossless, verified), 0 = full rows")
    ap.add_argument("--halo", type=int, default=-1,
                    help="N>1: 1 = ship only the rows the receiving rank's slice references (halo exchange), 0 = every row to every "
                         "rank, -1 (default) = halo when the busiest rank would receive at most half of the all-gather's rows")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (gloo only to rehearse several ranks on ONE GPU)")
    ap.add_argument("--host-path", action="store_true",
                    help="also time the host-pointer path (PCIe inclusive), reported separately")
    ap.add_argument("--multi-handle", type=int, default=0, metavar="P",
                    help="time the path behind the C ABI's multi-device handle (gnnvc_create_multi: what the reference's one call "
                         "site m.predict gets with GNNVC_DEVICES): P parts on the first P devices of this process — or, with fewer "
                         "devices, several parts per device (a rehearsal: the exchange's bytes and the per-part critical path are "
                         "real, the forward's wall time is not a scaling number)")
    args = ap.parse_args()

    # stdout carries ONE line, the result: whatever libraries print there while the job runs (RCCL's version banner,
    # gloo's connection notes) is sent to stderr instead, and the JSON goes out through the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run "
              f"--nproc-per-node {args.gpus} (WORLD_SIZE={world})", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the engine has no CPU fallback", file=sys.stderr)
        return 2
    dev_index = local_rank % torch.cuda.device_count()   # == local_rank on a real multi-GPU node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # rehearsal on one GPU: GNNVC_BENCH_ONE_RANK_GROUP=1 runs the partitioned path (slices, pieces, packed exchange,
    # RCCL calls on a one-rank group) instead of the single-GPU forward
    multi = world > 1 or bool(os.environ.get("GNNVC_BENCH_ONE_RANK_GROUP"))
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        # --gpus N smoke: the group really has N ranks and a collective crosses all of them
        got = dist.get_world_size()
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)
        if got != world or int(one.item()) != world:
            print(f"bench.py: the process group has {got} ranks / an all-reduce saw {int(one.item())}, expected {world}",
                  file=sys.stderr)
            return 3

    import gnn_mwvc_amd as G
    from gnn_mwvc_amd import distributed as D
    from tools import graphgen_torch as ggt

    t0 = time.time()
    g, workload_desc = build_workload(args.workload, ggt, dev)   # every rank builds the same graph (same Philox stream)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    n, nnz, n_edges, ws = g.n, g.nnz, g.n_edges, g.ws
    csr_bytes_full = (g.rowptr.numel() + g.col.numel() + g.w.numel() + g.nw.numel()) * 4

    if args.multi_handle:
        rc = multi_handle_run(args, dev, g, workload_desc, result_fd)
        return rc

    def make_engine(**opts):
        e = G.Engine(G.default_model_text(), device=dev_index)
        assert e.fused and e.num_stages == 3
        e.set_weight_scale(ws)
        if args.no_blocked:
            e.set_option("blocked_stage0", 0)
        if args.no_lds_table:
            e.set_option("lds_table", 0)
        if args.no_compact:
            e.set_option("compact_gather", 0)
        if args.no_overlap:
            e.set_option("overlap_dense", 0)
        if args.block_cols:
            e.set_option("block_cols", args.block_cols)
        if args.long_threshold >= 0:
            e.set_option("long_row_threshold", args.long_threshold)
        if args.giant_threshold >= 0:
            e.set_option("giant_row_threshold", args.giant_threshold)
        e.set_option("side_streams", args.side_streams)
        if args.mfma >= 0:
            e.set_option("mfma_dense", args.mfma)
        e.set_option("sorted_tiles", args.sorted_tiles)
        if args.sorted_long_threshold > 0:
            e.set_option("sorted_long_row_threshold", args.sorted_long_threshold)
        for kv in args.opt:
            k, _, v = kv.partition("=")
            e.set_option(k, int(v))
        for k, v in opts.items():
            e.set_option(k, v)
        return e

    def attach_whole(e):
        e.attach_graph_device(n, nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)

    x = g.x().contiguous()
    part_mode = args.partition if args.partition != "auto" else \
        ("nnz" if WORKLOADS[args.workload][0] in ("rmat", "powerlaw") else "rows")
    bounds = D.partition_bounds(n, world, g.rowptr if part_mode == "nnz" else None, part_mode)
    lo, hi = bounds[rank]

    eng = make_engine()
    halo_plan = None
    ref_scores = None
    csr_bytes_rank = csr_bytes_full
    t0 = time.time()
    if multi:
        # Reference for the self-check, while the whole graph still exists on this rank: a plain single-GPU forward
        # (setup only).  Then the rank cuts its slice out, lets the whole graph go and attaches the slice: from here on
        # its CSR memory is 1 / P of the graph's.
        ref_scores = torch.zeros(n, dtype=torch.float32, device=dev)
        e0 = make_engine()
        attach_whole(e0)
        e0.forward_device(x.data_ptr(), ref_scores.data_ptr(), 0)
        e0.synchronize()
        e0.close()
        sl = D.slice_csr(n, g.rowptr, g.col, g.w, g.nw, lo, hi, pad=ggt.COL_PAD)
        csr_bytes_rank = sl.nbytes()
        g.rowptr = g.col = g.w = g.nw = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        t0 = time.time()
        eng.attach_graph_slice(n, lo, hi, sl.nnz, sl.rowptr.data_ptr(), sl.col.data_ptr(), sl.w.data_ptr(), sl.nw.data_ptr(),
                               keepalive=sl)
        if args.halo != 0:
            # which rows of the other shards this rank's slice references at all (once per graph; setup)
            halo_plan = D.build_halo_plan(sl.col.to(torch.int64), sl.nnz, bounds, rank, n, max_fraction=0.5 if args.halo < 0 else 2.0)
    else:
        attach_whole(eng)
    eng.synchronize()
    t_attach = time.time() - t0
    # a dedicated HIP stream shared by torch (events, collectives) and the engine;
    # torch's default stream has the NULL handle, which the ABI reads as "engine's own"
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    eng.set_stream(stream.cuda_stream)

    bufs = D.ForwardBuffers.allocate(n, bounds, dev)

    def stage_fn(st, r0, r1, src, dst, lg):
        eng.stage_forward_device(st, r0, r1, src.data_ptr(), dst.data_ptr(),
                                 lg.data_ptr() if lg is not None else 0)

    # HIP events on the launch stream around every stage launch of the timed region
    # (the engine launches on this torch stream, so torch events bracket its kernels)
    stage_evt = [[torch.cuda.Event(enable_timing=True) for _ in range(6)]
                 for _ in range(args.steps)]

    # Per-rank plans (round 3: both plans can be laid out over the rows a rank holds).  Measured on one GPU with rank 0's slice
    # of the metric graph, each stage over the whole slice (scratch/experiments/rank_compute.py; ms per stage, plans vs plain
    # kernels): P = 2: 0.50 / 1.28 / 1.16 vs 1.75 / 2.0 / 2.0;  P = 4: 0.41 / 0.84 / 0.77 vs 0.88 / 1.01 / 0.99;
    # P = 8: 0.36 / 0.67 / 0.64 vs 0.44 / 0.53 / 0.51 — the compact table's fixed passes over ALL N rows (count, compact: ~0.3 ms)
    # stop paying once a rank has an eighth of the rows.  So: up to 4 ranks announce each 16-wide stage's input (compact table
    # over the rank's rows) and cut a stage into pieces of whole 256-chunk rounds (3 at 2 ranks, 2 at 3) or, at 4 ranks, run it
    # as ONE piece (so that stage 0 takes the LDS-table plan over the slice too); beyond 4 ranks: 4 pipelined pieces of the plain
    # kernels, whose all-gathers overlap the next piece's gathers.
    use_prepare = bool(args.prepare_input) and 2 <= world <= 4
    chunks = args.pipeline_chunks if args.pipeline_chunks >= 0 else ((3 if world == 2 else 2 if world == 3 else 1) if use_prepare else 4)
    prepare_fn = (lambda st, src, r0, r1: eng.stage_input_ready(st, src.data_ptr(), r0, r1)) if use_prepare else None
    piece_rows = [0]   # set after the first forward: pieces of 256 of the engine's chunks, so that no piece ends inside one
    fwd_scores = torch.zeros(n, dtype=torch.float32, device=dev)
    fwd_logits = torch.zeros(n, dtype=torch.float32, device=dev)

    def step(k: int | None):
        if not multi:
            # one GPU: the whole forward inside the engine (its own feature buffers; per-stage HIP events on this
            # stream are read back after the timed region)
            eng.forward_device(x.data_ptr(), fwd_scores.data_ptr(), fwd_logits.data_ptr())
            return
        hook = None
        if k is None and os.environ.get("GNNVC_BENCH_TRACE") == "2":
            def hook(st, phase):
                print(f"[rank {rank} +{time.time() - t_start:.2f}s] stage {st} {phase}", file=sys.stderr, flush=True)
        if k is not None:
            ev = stage_evt[k]

            def hook(st, phase):
                if phase == "begin":
                    ev[2 * st].record(stream)
                elif phase == "computed":
                    ev[2 * st + 1].record(stream)
        # every rank holds only its slice: every stage is partitioned (replicate = {}).
        # verify=False: the dead-column check of the compressed exchange is read once, after the loop
        D.partitioned_forward(stage_fn, 3, x, bufs, bounds, rank, on_stage=hook, gather_logits=False,
                              replicate=set(), pipeline_chunks=chunks, codec=codec, verify=False, prepare_fn=prepare_fn,
                              piece_rows=piece_rows[0], halo=halo_plan)

    trace = bool(os.environ.get("GNNVC_BENCH_TRACE"))
    t_start = time.time()

    def mark(what):
        if trace:
            if os.environ.get("GNNVC_BENCH_TRACE") != "2":   # "2": host-side progress only, no extra device syncs
                torch.cuda.synchronize()
            print(f"[rank {rank} +{time.time() - t_start:.1f}s] {what}", file=sys.stderr, flush=True)

    codec = G.EngineRowCodec(eng) if (multi and args.compress_exchange) else None
    mark("setup done")

    def settle():
        if os.environ.get("GNNVC_BENCH_NO_SETTLE"):
            return
        # outside the timed region the ranks are kept in step: the per-graph plans are built inside the second
        # forward (host-synchronous pieces of work of different length on every rank)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()

    def timed_once():
        torch.cuda.synchronize()
        t = time.perf_counter()
        step(None)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3

    # The first three forwards on this graph, timed one by one (host wall clock around a synchronised forward):
    # the first runs on a fresh graph (no plan exists yet — what a score-once caller gets), the second and third
    # build the per-graph plans inside them.  With N > 1 the first one also learns which feature columns are live.
    early_ms = []
    for i in range(3):
        early_ms.append(timed_once())
        settle()
        if i == 0 and codec is not None and use_prepare and args.pipeline_chunks < 0 and eng.get_info("compact_gather_active"):
            piece_rows[0] = 256 * eng.get_info("compact_gather_rows_per_chunk")
    plan_build_ms = eng.get_info("plan_build_us") / 1e3
    mark("plans settled")
    for i in range(args.warmup):
        step(None)
        settle()
        mark(f"warm-up {i} done")
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    exchange_ok = D.exchange_verified(bufs) if codec is not None else True   # no shipped row hid a non-zero

    if not multi:
        # the engine's own HIP events (same stream) around each stage of ONE more forward right after the timed region (option
        # "forward_timing" 2: the timed forwards record no events — four records were 5.5 us of a small forward); the mean step
        # time of the K timed forwards is ms_per_step
        eng.set_option("forward_timing", 2)
        eng.forward_device(x.data_ptr(), fwd_scores.data_ptr(), fwd_logits.data_ptr())
        torch.cuda.synchronize()
        stage_ms = list(eng.last_forward_ms()[1])
        eng.set_option("forward_timing", 0)
    else:
        stage_ms = [sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for ev in stage_evt) / args.steps
                    for i in range(3)]

    ms_per_step = elapsed * 1e3 / args.steps
    edges_per_s = n_edges / (elapsed / args.steps)
    fwd_bytes = sum(stage_bytes(i, n, nnz) for i in range(3))          # = 288 E + 300 N
    fwd_gbs = fwd_bytes / (ms_per_step * 1e-3) / 1e9 / world           # per GPU

    # ---- per-kernel HIP-event times (main-stream kernels) and the dominant kernel: the SAME K steps once more, right
    # after the timed region, with two HIP events around every kernel launch on the launch stream.  (The events are kept
    # out of the timed region itself: between the short rounds of the last stage they cost ~0.1 ms per forward; the traced
    # loop's own time per step is reported next to ms_per_step.)
    kernels, dominant, traced_ms = {}, None, None
    if args.kernel_trace and not multi:
        eng.set_option("kernel_trace", 1)
        step(None)
        torch.cuda.synchronize()
        eng.kernel_trace(1)     # (reading clears the records)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(None)
        torch.cuda.synchronize()
        traced_ms = (time.perf_counter() - t1) * 1e3 / args.steps
        recs = eng.kernel_trace(16384)
        eng.set_option("kernel_trace", 0)
        acc = {}
        for name, ms in recs:
            k = short_kernel(name)
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += ms
        for k, (calls, total) in acc.items():
            kernels[k] = {"launches_per_forward": calls / args.steps, "ms_per_forward": total / args.steps,
                          "ms_per_launch": total / calls}
        if kernels:
            dk = max(kernels, key=lambda k: kernels[k]["ms_per_forward"])
            d = kernels[dk]
            ab = kernel_algorithmic_bytes(dk, n, nnz, d["launches_per_forward"])
            dominant = {"name": dk, **d, "algorithmic_bytes_per_forward": ab}
            if ab:
                gbs = ab / (d["ms_per_forward"] * 1e-3) / 1e9
                dominant.update({"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "exceeds_peak": gbs > HBM_PEAK_GBS})
                if gbs > HBM_PEAK_GBS:
                    dominant["note"] = ("algorithmic bytes / time is above the HBM peak: this kernel does not move those "
                                        "bytes — the plan in force serves its gathers from a table swept through L2 / LDS "
                                        "(DESIGN.md §5); the forward-level fraction and the measured traffic are the bounds")

    traffic, traffic_src = measured_traffic(args.workload) if not multi else (None, None)
    plans = {"compact_gather_f16": bool(eng.get_info("compact_gather_active")),
             "pruned_adjacency": {f"stage{st}": ({"entries_kept": eng.get_info(f"pruned_entries_stage{st}"),
                                                  "zero_row_vertices": eng.get_info(f"pruned_vertices_stage{st}"),
                                                  "last_call_fit": bool(eng.get_info(f"pruned_last_ok_stage{st}"))}
                                                 if eng.get_info(f"pruned_stage{st}") else None) for st in (1, 2)},
             "compact_gather_last": {"fit": bool(eng.get_info("compact_gather_last_ok")), "passes": eng.get_info("compact_gather_last_passes"),
                                     "dirty_rows": eng.get_info("compact_gather_last_dirty"), "column_blocks": eng.get_info("compact_gather_blocks"),
                                     "steps": eng.get_info("compact_gather_steps")},
             "lds_table_stage0": bool(eng.get_info("lds_table_active")),
             "lds_table_skewed_layout": bool(eng.get_info("lds_table_mapped")),
             "compact_gather_switched_off": [bool(eng.get_info("compact_gather_off_stage1")), bool(eng.get_info("compact_gather_off_stage2"))],
             "heavy_tail_share": eng.get_info("heavy_tail_x1000") / 1000.0,
             "giant_stream_segments": eng.get_info("giant_segments"),
             "blocked_stage0": bool(eng.get_info("blocked_stage0_active")),
             "column_blocks": eng.get_info("blocked_blocks"), "block_cols": eng.get_info("block_cols"),
             "mfma_dense": eng.get_info("mfma_dense"),
             "sorted_tiles": bool(eng.get_info("sorted_tiles_active")),
             "table_tiles": [bool(eng.get_info("table_tiles_fit_stage1")), bool(eng.get_info("table_tiles_fit_stage2"))]
             if eng.get_info("table_tiles_active") else None,
             "side_queue_runs_beside": bool(eng.get_info("side_queue_runs_beside")) if eng.get_info("long_rows") else None,
             "natural_tile_waste": eng.get_info("tile_waste_x100") / 100.0,
             "interleaved_tiles": bool(eng.get_info("interleaved_tiles")), "long_rows": eng.get_info("long_rows"),
             "long_row_threshold": eng.get_info("long_row_threshold"), "giant_rows": eng.get_info("giant_rows"),
             "giant_entries": eng.get_info("giant_entries"), "giant_row_threshold": eng.get_info("giant_row_threshold")}
    out = {
        "metric": "GNN forward edges/sec", "value": edges_per_s, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "mode": "exact",
        "config": {"workload": workload_desc, "vertices": n, "edges": n_edges,
                   "graph": args.workload, "partition": f"1d-vertex x{world} ({part_mode}-balanced)",
                   "timed_state": "steady state of a graph scored repeatedly: per-graph plans built in its first two forwards, "
                                  "before the warm-up; see first_forward_ms for a graph scored once",
                   "csr_bytes_per_rank": csr_bytes_rank, "csr_bytes_whole_graph": csr_bytes_full,
                   "exchange": "none" if not multi else "all-gather of the N feature rows (16 fp32, or only their live "
                                                          "columns) after each partitioned stage, N scores at the end"},
        "first_forward_ms": early_ms[0], "second_forward_ms": early_ms[1], "third_forward_ms": early_ms[2],
        # a graph scored ONCE (the reference driver's call pattern, src/GNN_VC.cpp:171-192): edges/s of the first forward on
        # the fresh graph, and its forward-level fraction
        "score_once_value": n_edges / (early_ms[0] * 1e-3),
        "first_forward_roofline_frac": fwd_bytes / (early_ms[0] * 1e-3) / 1e9 / world / HBM_PEAK_GBS,
        "plan_build_ms": plan_build_ms,
        "roofline": {"bound": "hbm",
                     "definition": "forward level (SURVEY.md 8d): (288 E + 300 N) algorithmic bytes / ms_per_step, per GPU",
                     "achieved": fwd_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fwd_gbs / HBM_PEAK_GBS,
                     "forward_bytes": fwd_bytes,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_frac": (traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "dominant_kernel": dominant, "kernels": kernels, "kernel_trace_ms_per_step": traced_ms},
        "stage_ms": stage_ms, "graph_build_s": t_gen, "graph_attach_s": t_attach,
        "plan": plans,
    }

    if multi:
        # self-check: the partitioned result on THIS rank against the plain single-GPU forward of the whole graph that
        # this GPU ran during setup (before it let the whole graph go)
        bad = (bufs.scores[:n].view(torch.int32) != ref_scores.view(torch.int32)).sum().to(torch.int64)
        dist.all_reduce(bad, op=dist.ReduceOp.SUM)
        out["parity"] = {"partitioned_vs_single_gpu_score_bit_mismatches_all_ranks": int(bad.item())}
        per_rank = torch.zeros(world, dtype=torch.int64, device=dev)
        per_rank[rank] = csr_bytes_rank
        dist.all_reduce(per_rank)
        out["config"]["csr_bytes_each_rank"] = [int(v) for v in per_rank.tolist()]
        out["config"]["replicated_stages"] = []
        out["config"]["pipeline_chunks"] = chunks
        out["config"]["piece_rows"] = piece_rows[0]
        out["config"]["compact_table_over_rank_rows"] = bool(use_prepare and eng.get_info("compact_gather_active"))
        out["config"]["halo_exchange"] = None if halo_plan is None else {
            "in_use": bool(halo_plan.use_halo), "rows_received_per_exchange": halo_plan.halo_rows_in,
            "rows_an_all_gather_delivers": halo_plan.full_rows_in, "rows_sent_per_exchange": halo_plan.halo_rows_out}
        out["config"]["compressed_exchange"] = None if codec is None else {
            "verified_lossless": bool(exchange_ok),
            "bytes_per_row_shipped_of_64": {str(st): (round(pk.bytes_per_row, 2) if pk is not None else 64)
                                            for st, pk in sorted(bufs.live.items())},
            "dense_columns": {str(st): (bin(pk.mask).count("1") if pk is not None else 16)
                              for st, pk in sorted(bufs.live.items())}}
        if not exchange_ok:
            out["invalid"] = "an exception list of the compressed exchange overflowed in the timed region"
    if rank == 0 and not multi:
        # what a caller pays who hands a graph over and scores it once (the reference's driver, src/GNN_VC.cpp:171-192)
        out["score_once"] = {"attach_ms": t_attach * 1e3, "first_forward_ms": early_ms[0],
                             "attach_plus_first_forward_ms": t_attach * 1e3 + early_ms[0],
                             "handoff_build_ms": eng.get_info("handoff_build_us") / 1e3,
                             "plans_at_handoff": eng.get_info("plans_at_handoff"),
                             "first_forward_over_steady": early_ms[0] / ms_per_step,
                             "note": "attach = device-side checks + row classes + the plans built at hand-off + every buffer of the "
                                     "forward (a fresh engine: allocations included); device-resident CSR, no PCIe copy in it"}
        if not args.no_variants:
            out.update(forward_variants(make_engine, attach_whole, x, n, dev))
            # hipMalloc on this stack now and then takes 100+ ms for no reason of ours (seen on the first and on the sixth engine
            # of a process): the hand-off numbers are the MEDIAN of the timed engine's and two more fresh engines' (all listed)
            trials = [(out["score_once"]["attach_ms"], out["score_once"]["first_forward_ms"]),
                      (out["fresh_engine_attach_ms"], out["fresh_engine_first_forward_ms"])]
            e4 = make_engine()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            attach_whole(e4)
            e4.synchronize()
            a4 = (time.perf_counter() - t4) * 1e3
            t4 = time.perf_counter()
            e4.forward_device(x.data_ptr(), fwd_scores.data_ptr(), 0)
            e4.synchronize()
            trials.append((a4, (time.perf_counter() - t4) * 1e3))
            e4.close()
            med = sorted(trials, key=lambda t: t[0] + t[1])[1]
            out["score_once_value"] = n_edges / (med[1] * 1e-3)
            out["first_forward_roofline_frac"] = fwd_bytes / (med[1] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["score_once"].update({"attach_ms": med[0], "first_forward_ms": med[1], "attach_plus_first_forward_ms": med[0] + med[1],
                                      "first_forward_over_steady": med[1] / ms_per_step,
                                      "trials_attach_first_ms": [[round(a, 3), round(f, 3)] for a, f in trials]})
        if args.host_path or not args.no_host_path:
            # PCIe-inclusive path (host x in, host scores + logits out); never `value`
            xh = x.cpu().numpy()
            keep = eng.forward(xh)          # the caller's output buffers, reused like the reference's `out` matrix
            t1 = time.perf_counter()
            for _ in range(3):
                eng.forward(xh, out=keep)
            out["host_path_ms"] = (time.perf_counter() - t1) * 1e3 / 3
            del xh, keep
        # The other single-GPU configs are TIMED before the CPU legs: a forward of theirs is a few dozen launches per
        # millisecond, and the launching thread shares this process's CPU quota with the oracle's 16 OpenMP workers and
        # OpenBLAS's spinning pool once those have run (power-law graph: 0.98 ms per forward before them, 1.50 after).
        checks = []
        if not args.no_workloads and args.workload == "er10m":
            out["workloads"] = {}
            for name in [w for w in args.workloads.split(",") if w]:
                out["workloads"][name], chk = side_workload(name, args, dev, make_engine, ggt)
                checks.append((name, chk))
        if not args.no_cpu_baseline:
            out["cpu_baseline"], out["cpu_baseline_openmp_aggregation"], out["parity"], handoff = \
                cpu_baseline(args, dev, eng, ggt, g, x, fwd_logits, make_engine)
            out["score_once"].update(handoff)
            for name, chk in checks:
                out["workloads"][name].update(chk())

    if rank == 0 and plans.get("side_queue_runs_beside") is False:
        out["warning"] = "no side queue beside the main stream was found: the long / giant rows ran serialised with the tile kernels"
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    eng.close()
    if multi:
        dist.destroy_process_group()
    return 0


def multi_handle_run(args, dev, g, desc, result_fd):
    """--multi-handle P: the forward behind ONE handle of the C ABI that drives P devices (gnnvc_create_multi, csrc/gnnvc_multi.cpp)
    — the path the reference's single call site m.predict(x, out, g) (src/GNN_VC.cpp:192) takes with GNNVC_DEVICES set.  Host
    hand-off (gnnvc_upload_graph cuts the graph and gives every device its slice), first forward (chooses the exchange's
    packing), K timed steps of gnnvc_forward_device, every score against a single engine's, the bytes a part ships to one peer per
    exchange, and part 0's critical path — its compute + pack + push + expand with the other parts idle ("multi_only_part")."""
    import numpy as np
    import torch
    import gnn_mwvc_amd as G
    P = args.multi_handle
    ndev = torch.cuda.device_count()
    devices = list(range(P)) if ndev >= P else [i % ndev for i in range(P)]
    n, nnz, n_edges = g.n, g.nnz, g.n_edges
    x = g.x().contiguous()
    ref_sc = torch.zeros(n, dtype=torch.float32, device=dev)
    ref_lg = torch.zeros(n, dtype=torch.float32, device=dev)
    e0 = G.Engine(G.default_model_text(), device=0)
    e0.set_weight_scale(g.ws)
    e0.attach_graph_device(n, nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    e0.forward_device(x.data_ptr(), ref_sc.data_ptr(), ref_lg.data_ptr())
    e0.synchronize()
    e0.close()
    hg = g.to_host()
    e = G.Engine(G.default_model_text(), devices=devices)
    e.set_weight_scale(g.ws)
    for kv in args.opt:
        k, _, v = kv.partition("=")
        e.set_option(k, int(v))
    t = time.perf_counter()
    e.upload_graph(hg)
    e.synchronize()
    upload_ms = (time.perf_counter() - t) * 1e3
    sc = torch.zeros(n, dtype=torch.float32, device=dev)
    lg = torch.zeros(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def once():
        t = time.perf_counter()
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())   # complete when it returns
        return (time.perf_counter() - t) * 1e3

    early = [once() for _ in range(3)]
    first_bad = int((lg.view(torch.int32) != ref_lg.view(torch.int32)).sum())
    for _ in range(args.warmup):
        once()
    t = time.perf_counter()
    for _ in range(args.steps):
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    ms = (time.perf_counter() - t) * 1e3 / args.steps
    bad = int((lg.view(torch.int32) != ref_lg.view(torch.int32)).sum())
    bad_sc = int((sc.view(torch.int32) != ref_sc.view(torch.int32)).sum())
    info = {k: e.get_info(k) for k in ("multi_pieces", "multi_packed_stage0", "multi_packed_stage1", "multi_packed_columns_stage0",
                                       "multi_packed_columns_stage1", "multi_exchange_bytes_per_peer_stage0",
                                       "multi_exchange_bytes_per_peer_stage1", "multi_peer_stores")}
    spans = [e.get_info(f"multi_part_span_us_{r}") / 1e3 for r in range(P)]
    parts = [{"rows": e.get_info(f"part_rows_{r}"), "entries": e.get_info(f"part_entries_{r}")} for r in range(P)]
    # part 0 alone: what ONE device of P would have to do per forward (its three stages in pieces, pack, push to P - 1 peers,
    # expand P - 1 peers' pieces), nothing else on the GPU
    solo_ms = None
    if P > 1:
        e.set_option("multi_only_part", 0)
        for _ in range(2):
            once()
        solo = []
        for _ in range(max(args.steps // 2, 3)):
            once()
            solo.append(e.get_info("multi_part_span_us_0") / 1e3)
        solo_ms = sorted(solo)[len(solo) // 2]
        e.set_option("multi_only_part", -1)
    e.close()
    distinct = len(set(devices))
    fwd_bytes = sum(stage_bytes(i, n, nnz) for i in range(3))
    out = {"metric": "GNN forward edges/sec", "value": n_edges / (ms * 1e-3), "unit": "edges/s", "n_gpus": distinct,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic", "mode": "exact",
           "config": {"workload": desc, "vertices": n, "edges": n_edges, "graph": args.workload,
                      "path": "gnnvc_create_multi: one handle of the C ABI, one host thread per device, packed pieces pushed over the fabric",
                      "parts": P, "devices": devices,
                      "rehearsal": distinct < P,
                      "note": ("several parts share a device: ms_per_step is NOT a scaling number — the exchange's bytes, the bits and "
                               "part 0's critical path are what this run shows") if distinct < P else "one part per device"},
           "first_forward_ms": early[0], "second_forward_ms": early[1], "third_forward_ms": early[2],
           "score_once_value": n_edges / (early[0] * 1e-3),
           "host_upload_ms": upload_ms,
           "roofline": {"bound": "hbm", "definition": "forward level (SURVEY.md 8d): (288 E + 300 N) algorithmic bytes / ms_per_step, per GPU",
                        "achieved": fwd_bytes / (ms * 1e-3) / 1e9 / distinct, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": fwd_bytes / (ms * 1e-3) / 1e9 / distinct / HBM_PEAK_GBS, "traffic": None},
           "multi": {**info, "part_span_ms_last_forward": spans, "parts": parts,
                     "part0_critical_path_ms": solo_ms,
                     "part0_critical_path_note": "part 0's compute + pack + push (to every peer) + expand (every peer's pieces) with the "
                                                 "other parts idle: HIP events on its stream, median"},
           "parity": {"logit_bit_mismatches_vs_single_engine": bad, "score_bit_mismatches_vs_single_engine": bad_sc,
                      "first_forward_logit_bit_mismatches_vs_single_engine": first_bad, "logits_checked": n}}
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    return 0 if (bad == 0 and bad_sc == 0 and first_bad == 0) else 4


def forward_variants(make_engine, attach_whole, x, n, dev):
    """What the headline leaves out (N = 1): a fresh engine's attach + first forward on the same graph — no plan exists,
    the reference driver's call pattern (src/GNN_VC.cpp:171-192 hands predict a new graph every call) — and the steady
    state with the per-graph plans switched off."""
    import torch
    sc = torch.zeros(n, dtype=torch.float32, device=dev)
    res = {}

    def run(e):
        torch.cuda.synchronize()
        t = time.perf_counter()
        e.forward_device(x.data_ptr(), sc.data_ptr(), 0)
        e.synchronize()
        return (time.perf_counter() - t) * 1e3

    e = make_engine()
    torch.cuda.synchronize()
    t = time.perf_counter()
    attach_whole(e)
    e.synchronize()
    res["fresh_engine_attach_ms"] = (time.perf_counter() - t) * 1e3
    e.set_option("forward_timing", 1)
    res["fresh_engine_first_forward_ms"] = run(e)
    res["fresh_engine_first_forward_device_ms"] = e.last_forward_ms()[0]
    e.close()
    e = make_engine(plans_at_handoff=0)      # round 2's timing: plans inside the graph's first two forwards
    torch.cuda.synchronize()
    t = time.perf_counter()
    attach_whole(e)
    e.synchronize()
    a0 = (time.perf_counter() - t) * 1e3
    f0 = run(e)
    res["plans_in_forward"] = {"attach_ms": a0, "first_forward_ms": f0, "attach_plus_first_forward_ms": a0 + f0}
    e.close()
    e = make_engine(lds_table=0, compact_gather=0, prune_zero_rows=0)
    attach_whole(e)
    for _ in range(3):
        run(e)
    ts = sorted(run(e) for _ in range(5))
    res["plain_forward_ms"] = ts[len(ts) // 2]
    e.set_option("forward_timing", 1)
    run(e)
    res["plain_forward_device_ms"] = e.last_forward_ms()[0]
    res["plain_forward_plans"] = {"blocked_stage0": bool(e.get_info("blocked_stage0_active")), "lds_table": False,
                                  "compact_gather": False, "pruned_adjacency": False}
    e.close()
    return res


def measured_traffic(workload: str):
    """Measured memory traffic of one steady-state forward, from the newest committed PMC summary (rocprofv3 --pmc passes of
    tools/pmc_probe.py on that workload: L2->fabric read requests x 128 B, calibrated on a 1 GiB copy in the same
    run, + WRITE_SIZE) — only if that summary was taken from the kernels as they are now (it records a hash of the
    kernel sources); otherwise null: a stale number would not describe what was just timed."""
    prof = sorted((ROOT / "profiles").glob("r*/pmc_summary.json" if workload == "er10m" else f"r*/pmc_summary_{workload}.json"))
    if not prof:
        return None, None
    data = json.loads(prof[-1].read_text())
    meta = data.get("_meta", {})
    if meta.get("source_hash") != source_hash() or "forward_traffic_bytes" not in meta:
        return None, f"{prof[-1].relative_to(ROOT)} is from other kernel sources (hash {meta.get('source_hash')}): not used"
    return float(meta["forward_traffic_bytes"]), str(prof[-1].relative_to(ROOT))


def cpu_baseline(args, dev, eng, ggt, g, x, timed_logits, make_engine):
    """The CPU leg (rank 0, N = 1): baseline timings of the oracle and the parity checks that need it."""
    import numpy as np
    import torch
    import gnn_mwvc_amd as G
    from oracle import oracle_py
    from tools import graphgen as gg

    threads = oracle_py.num_threads()
    blas = oracle_py.use_openblas(threads)
    om = oracle_py.OracleModel(G.default_model_text())

    # ---- parity on the TIMED graph, timed configuration: (a) every logit of the engine's whole forward (plans in force)
    # equals the stage-by-stage path's, (b) exact rows: for sampled vertices the oracle recomputes each stage's row from
    # the device's own stage inputs, in CSR order
    n = g.n
    h1 = torch.zeros((n + 1, 16), dtype=torch.float32, device=dev)
    h2 = torch.zeros((n + 1, 16), dtype=torch.float32, device=dev)
    sc = torch.zeros(n, dtype=torch.float32, device=dev)
    lg = torch.zeros(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    eng.stage_forward_device(0, 0, n, x.data_ptr(), h1.data_ptr())
    eng.stage_forward_device(1, 0, n, h1.data_ptr(), h2.data_ptr())
    eng.stage_forward_device(2, 0, n, h2.data_ptr(), sc.data_ptr(), lg.data_ptr())
    eng.synchronize()
    whole_vs_staged = int((lg.view(torch.int32) != timed_logits.view(torch.int32)).sum().item())
    om.set_weight_scale(g.ws)
    params = om.linear_params()
    oracle_py.set_num_threads(1)     # one-row problems
    rng = np.random.default_rng(123)
    deg = (g.rowptr[1:] - g.rowptr[:-1]).to(torch.int64)
    heavy = torch.topk(deg, min(4, n)).indices.cpu().numpy()
    sample = np.unique(np.concatenate([rng.integers(0, n, 256), [0, n - 1], heavy]))
    stage_in = [x.reshape(-1, 1), h1, h2]
    stage_out = [h1, h2, lg.reshape(-1, 1)]
    row_mismatch = 0
    for u in sample:
        s, t = int(g.rowptr[u]), int(g.rowptr[u + 1])
        nbrs = g.col[s:t].to(torch.int64)
        d = t - s
        star = gg.CsrGraph(d + 1, np.array([0, d] + [d] * d, dtype=np.uint64), np.arange(1, d + 1, dtype=np.uint32),
                           np.array([int(g.w[u]) & 0xFFFFFFFF] + [0] * d, dtype=np.uint32),
                           np.array([int(g.nw[u]) & 0xFFFFFFFF] + [0] * d, dtype=np.uint32))
        for st in range(3):
            src = stage_in[st]
            feats = torch.cat([src[u: u + 1], src[nbrs]]).cpu().numpy().astype(np.float32)
            a = oracle_py.graph_layer(star, g.ws, feats)[:1]
            for i, (W, b) in enumerate(params[3 * st: 3 * st + 3]):
                a = oracle_py.linear_layer(a, W, b)
                if not (st == 2 and i == 2):
                    a = oracle_py.relu(a)
            got = stage_out[st][u].cpu().numpy()
            row_mismatch += int(not np.array_equal(got.view(np.uint32), a[0].view(np.uint32)))
    oracle_py.set_num_threads(threads)
    del h1, h2, sc, lg

    # ---- the CPU baseline: on the workload graph itself, same run (default), or on a bounded sample of its family
    spec = WORKLOADS[args.workload]
    on_workload = not args.cpu_sample
    if on_workload:
        gs, what = g, "the timed graph itself"
    elif spec[0] == "rmat":
        gs, what = ggt.rmat(20, 16, 99, dev), "r-mat scale=20 edge_factor=16 (same generator, seed 99)"
    elif spec[0] == "powerlaw":
        gs, what = ggt.power_law_hubs(1_000_000, 16.0, 2.1, 8, 65536, 99, dev), "the same power-law construction, seed 99"
    else:
        sn, sm = (int(v) for v in args.cpu_sample.split("x"))
        gs, what = ggt.erdos_renyi(sn, sm, 99, dev), f"erdos-renyi n={sn} m={sm} (same generator, seed 99)"
    hg = gs.to_host()
    om.set_weight_scale(hg.ws)
    xh = hg.x()
    reps = 1 if hg.n_edges >= 50_000_000 else 3   # (the shipped-serial variant takes ~25 s per forward on the metric graph)

    def timed(**kw):
        om.predict(hg, xh, stop_after=om.n_layers - 2, **kw)   # warm-up (page faults, thread pools)
        ts, res = [], None
        for _ in range(reps):
            t0 = time.perf_counter()
            res = om.predict(hg, xh, stop_after=om.n_layers - 2, **kw)[:, 0]
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2], res              # median of 3 after 1 warm-up; 1 + 1 on the metric graph

    # The reference's call pattern through the host ABI on a fresh engine — upload (PCIe + hand-off plans), one forward — BEFORE
    # the CPU legs: their OpenMP / OpenBLAS pools keep spinning for a while and the pageable copies of this leg are host work
    # (behind them: first host forward 11.3 ms instead of 7.1)
    handoff, lg3_host = {}, None
    if on_workload:
        e3 = make_engine()
        try:
            t0 = time.perf_counter()
            e3.upload_graph(hg)
            e3.synchronize()
            up = (time.perf_counter() - t0) * 1e3
            out3 = (np.zeros((hg.n, 1), np.float32), np.zeros((hg.n, 1), np.float32))   # (the caller's `out` matrix: exists, is touched)
            t0 = time.perf_counter()
            sc3, lg3 = e3.forward(xh, out=out3)
            ff = (time.perf_counter() - t0) * 1e3
            handoff = {"host_upload_ms": up, "host_first_forward_ms": ff, "host_upload_plus_first_forward_ms": up + ff,
                       "host_upload_handoff_build_ms": e3.get_info("handoff_build_us") / 1e3}
            lg3_host = lg3[:, 0].copy()
            del sc3, lg3
        finally:
            e3.close()
    t_ship, want = timed(as_shipped=True)
    t_omp, want2 = timed(parallel_agg=True)
    if on_workload:
        # every logit of the TIMED graph in the TIMED configuration (plans in force) against the oracle
        got = timed_logits.cpu().numpy()
        if lg3_host is not None:
            handoff["host_first_forward_logit_bit_mismatches_vs_oracle"] = int((lg3_host.view(np.uint32) != want.view(np.uint32)).sum())
    else:
        # parity of the GPU path on the same sample (first forward on a fresh graph)
        eng.set_weight_scale(gs.ws)
        eng.attach_graph_device(gs.n, gs.nnz, gs.rowptr.data_ptr(), gs.col.data_ptr(),
                                gs.w.data_ptr(), gs.nw.data_ptr(), keepalive=gs)
        sc = torch.zeros(gs.n, dtype=torch.float32, device=dev)
        lg = torch.zeros(gs.n, dtype=torch.float32, device=dev)
        eng.forward_device(gs.x().contiguous().data_ptr(), sc.data_ptr(), lg.data_ptr())
        eng.synchronize()
        got = lg.cpu().numpy()
    mism = int((got.view(np.uint32) != want.view(np.uint32)).sum())
    how_timed = "one forward after a warm-up" if reps == 1 else "one forward, median of 3 after a warm-up"
    base = {"value": hg.n_edges / t_ship, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"{what}: {hg.n} vertices / {hg.n_edges} edges, {how_timed}",
            "how": "the reference's call pattern (src/gnn_inference.cpp:20-52): aggregation serial as shipped (its OpenMP pragma "
                   "is inert, Makefile:4,30-32), every linear layer through cblas_sgemm + serial bias add, serial ReLU",
            "sgemm": blas or "internal sequential-k fmaf loops on OpenMP threads (no OpenBLAS found on this host)",
            "seconds": t_ship}
    omp = {"value": hg.n_edges / t_omp, "unit": "edges/s", "cores": threads, "kind": "port",
           "label": "NOT what the reference ships: aggregation rows in parallel (OpenMP dynamic schedule — the variant "
                    "src/gnn_inference.cpp:31's inert pragma intended), dense layers row-parallel; same bits",
           "sample": base["sample"], "seconds": t_omp,
           "same_bits_as_shipped_variant": bool(np.array_equal(want.view(np.uint32), want2.view(np.uint32)))}
    parity = {"oracle_graph": what, "oracle_vertices": hg.n, "logit_bit_mismatches_vs_oracle": mism,
              "every_logit_of_the_timed_graph_checked": bool(on_workload),
              "timed_graph": {"whole_forward_vs_stage_path_logit_bit_mismatches": whole_vs_staged,
                              "sampled_rows": int(len(sample)), "stages_checked": 3,
                              "sampled_row_bit_mismatches_vs_oracle": row_mismatch}}
    return base, omp, parity, handoff


def side_workload(name, args, dev, make_engine, ggt):
    """One of BASELINE.json's other single-GPU configs on a fresh engine: attach, first forward, steady state, fraction of
    the forward-level roofline, and every logit against the oracle's whole forward."""
    import numpy as np
    import torch
    import gnn_mwvc_amd as G
    from oracle import oracle_py
    g, desc = build_workload(name, ggt, dev)
    torch.cuda.synchronize()
    n, nnz = g.n, g.nnz
    x = g.x().contiguous()
    sc = torch.zeros(n, dtype=torch.float32, device=dev)
    lg = torch.zeros(n, dtype=torch.float32, device=dev)
    e = make_engine()
    e.set_weight_scale(g.ws)
    torch.cuda.synchronize()

    def run():
        t = time.perf_counter()
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
        e.synchronize()
        return (time.perf_counter() - t) * 1e3

    t = time.perf_counter()
    e.attach_graph_device(n, nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    e.synchronize()
    attach_ms = (time.perf_counter() - t) * 1e3
    early = [run() for _ in range(3)]
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(args.steps):
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
    e.synchronize()
    ms = (time.perf_counter() - t) * 1e3 / args.steps
    fwd_bytes = sum(stage_bytes(i, n, nnz) for i in range(3))
    traffic, traffic_src = measured_traffic(name)
    traffic1, _ = measured_traffic(name + "_first")
    res = {"workload": desc, "vertices": n, "edges": g.n_edges, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms, "value": g.n_edges / (ms * 1e-3), "unit": "edges/s",
           "attach_ms": attach_ms, "first_forward_ms": early[0], "attach_plus_first_forward_ms": attach_ms + early[0],
           "second_forward_ms": early[1], "third_forward_ms": early[2], "first_forward_over_steady": early[0] / ms,
           # a graph scored ONCE — the reference driver's call pattern (src/GNN_VC.cpp:171-192): edges/s and the forward-level
           # fraction of the FIRST forward on the fresh graph
           "score_once_value": g.n_edges / (early[0] * 1e-3),
           "first_forward_roofline_frac": fwd_bytes / (early[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "roofline_frac": fwd_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "forward_bytes": fwd_bytes,
           # measured bytes of one steady-state forward (profiles/r*/pmc_summary_<workload>.json, only if taken from these sources)
           "traffic": traffic, "traffic_source": traffic_src,
           "traffic_frac": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
           "first_forward_traffic": traffic1,
           "first_forward_traffic_frac": (traffic1 / (early[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic1 else None,
           "plan": {"lds_table": bool(e.get_info("lds_table_active")), "compact_gather": bool(e.get_info("compact_gather_active")),
                    "pruned_stage1": bool(e.get_info("pruned_stage1")), "pruned_stage2": bool(e.get_info("pruned_stage2")),
                    "pruned_stage1_predicted_at_handoff": bool(e.get_info("pruned_predicted_stage1")),
                    "pruned_stage2_from_stage1_entries": bool(e.get_info("pruned_from_previous_stage2")),
                    "sorted_tiles": bool(e.get_info("sorted_tiles_active")), "long_rows": e.get_info("long_rows"),
                    "giant_rows": e.get_info("giant_rows"),
                    # (graphs of 50 - 400 K vertices: the 16-wide stages gather an L2-resident table, DESIGN.md 5 "Table tiles")
                    "table_tiles": [bool(e.get_info("table_tiles_fit_stage1")), bool(e.get_info("table_tiles_fit_stage2"))]
                    if e.get_info("table_tiles_active") else None,
                    "wide_tiles": bool(e.get_info("wide_tiles_used")),   # (small graphs: a workgroup per tile, DESIGN.md 5 "Wide tiles")
                    # (skewed graphs: the long / giant rows' kernels have to run BESIDE the tile kernel — a side queue that shares the
                    # main stream's hardware queue serialises them and nothing else would say so)
                    "side_queue_runs_beside": bool(e.get_info("side_queue_runs_beside")) if e.get_info("long_rows") else None}}
    if res["plan"]["side_queue_runs_beside"] is False:
        res["warning"] = "no side queue beside the main stream was found: this workload's long / giant rows ran serialised"
    # a fresh engine's first-forward logits too (what a score-once caller reads)
    e2 = make_engine()
    e2.set_weight_scale(g.ws)
    e2.attach_graph_device(n, nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    lg1 = torch.zeros(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    e2.forward_device(x.data_ptr(), sc.data_ptr(), lg1.data_ptr())
    e2.synchronize()
    e2.close()
    steady = lg.cpu().numpy()
    # ... and the driver's pattern on a LIVE engine (src/GNN_VC.cpp:171-192: one engine, a new graph every call): the same arrays
    # handed over again — a new graph to the engine: every per-graph state goes, the buffers stay — and scored once
    lg2 = torch.zeros(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    e.attach_graph_device(n, nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
    e.synchronize()
    live_attach = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    e.forward_device(x.data_ptr(), sc.data_ptr(), lg2.data_ptr())
    e.synchronize()
    live_first = (time.perf_counter() - t) * 1e3
    res["live_engine"] = {"attach_ms": live_attach, "first_forward_ms": live_first, "attach_plus_first_forward_ms": live_attach + live_first,
                          "note": "the same graph handed again to the engine that has just scored it (a new graph to the engine: no "
                                  "per-graph state survives, its buffers do) and scored once: what the reference driver's 2nd..nth "
                                  "predict call pays on device-resident arrays"}
    e.close()
    first, live = lg1.cpu().numpy(), lg2.cpu().numpy()
    hg = g.to_host()
    del g, x, sc, lg, lg1, lg2
    torch.cuda.empty_cache()

    def check():
        """every steady-state logit (plans in force) and every first-forward logit against the oracle's whole forward"""
        om = oracle_py.OracleModel(G.default_model_text())
        om.set_weight_scale(hg.ws)
        want = om.predict(hg, hg.x(), stop_after=om.n_layers - 2, parallel_agg=True)[:, 0]
        return {"logit_bit_mismatches_vs_oracle": int((steady.view(np.uint32) != want.view(np.uint32)).sum()),
                "first_forward_logit_bit_mismatches_vs_oracle": int((first.view(np.uint32) != want.view(np.uint32)).sum()),
                "live_engine_first_forward_logit_bit_mismatches_vs_oracle": int((live.view(np.uint32) != want.view(np.uint32)).sum()),
                "logits_checked": int(hg.n)}

    return res, check


if __name__ == "__main__":
    sys.exit(main())
