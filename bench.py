#!/usr/bin/env python3
"""bench.py — GNN forward edges/s on the metric graph (BASELINE.json).

One step = one full forward (3 fused stages: graph layer + dense layers) over
the synthetic Erdős–Rényi graph with 10 M vertices / 100 M undirected edges
(weights U[20,120]), device-resident in HBM when the timed region
starts.  N = 1: the engine's whole forward (gnnvc_forward_device).  N > 1: one
process per GPU (torch.distributed, backend nccl = RCCL), the graph 1-D
vertex-partitioned (gnn-mwvc_amd/distributed.py), the feature rows all-gathered
over xGMI between stages — only their live columns once a forward has shown which
those are; total work is fixed, so scaling is "strong".

Before the W warm-up steps come three untimed setup forwards (the engine builds its per-graph plans inside a
graph's second and third forward; with N > 1 the first of them also learns which feature columns are live).

Prints ONE JSON line on rank 0 (contract in the project brief):
  value      = undirected edges / second, whole job
  roofline   = the dominant stage's algorithmic HBM bytes / its duration (HIP
               events on the launch stream, live) vs 8 TB/s, its kernels by name
  cpu_baseline = the oracle (bit-equal CPU port of the reference path) timed on
               this host on a bounded sample graph of the same distribution
"""
from __future__ import annotations

import argparse
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


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="er10m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lds-table", action="store_true", help="F = 1 stage without the LDS-table plan")
    ap.add_argument("--no-compact", action="store_true", help="16-wide stages without the compact-table plan")
    ap.add_argument("--no-overlap", action="store_true", help="dense layers after the sums instead of under the next round's")
    ap.add_argument("--cpu-sample", default="2000000x20000000",
                    help="n x m of the CPU-baseline sample graph")
    ap.add_argument("--no-blocked", action="store_true", help="disable the column-blocked F=1 stage (A/B)")
    ap.add_argument("--block-cols", type=int, default=0)
    ap.add_argument("--long-threshold", type=int, default=-1, help="degree at which a row gets its own workgroup")
    ap.add_argument("--giant-threshold", type=int, default=-1, help="degree from which a long row is summed by the parallel scan kernels")
    ap.add_argument("--hub-mode", type=int, default=0, help="1 = tolerance mode for long rows (tree sums); never the default")
    ap.add_argument("--side-streams", type=int, default=1, help="0 = long / giant rows on the main stream (profiling: standalone kernel times)")
    ap.add_argument("--sorted-tiles", type=int, default=-1, help="degree-sorted tiles: -1 auto, 0 off, 1 on")
    ap.add_argument("--sorted-long-threshold", type=int, default=0)
    ap.add_argument("--mfma", type=int, default=-1, help="dense layers: 0 VALU, 1 MFMA everywhere, 2 MFMA in the 16-wide stages (default)")
    ap.add_argument("--prepare-input", type=int, default=1,
                    help="N = 2..3: 1 = announce each 16-wide stage's complete input to the engine (compact-table plan "
                         "over the rank's rows), 0 = plain gathering kernels")
    ap.add_argument("--pipeline-chunks", type=int, default=-1,
                    help="N>1: pieces per stage whose all-gather overlaps the next piece's compute (0/1 = off)")
    ap.add_argument("--replicate-stage0", type=int, default=-1,
                    help="N>1: 1/0 forces stage 0 replicated / partitioned; -1 = auto (P=2: stages 0,1; P<=4: stage 0)")
    ap.add_argument("--partition", default="auto", choices=["auto", "rows", "nnz"])
    ap.add_argument("--compress-exchange", type=int, default=1,
                    help="N>1: 1 = ship only the live feature columns between stages (lossless, verified), 0 = full rows")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (gloo only to rehearse several ranks on ONE GPU)")
    ap.add_argument("--host-path", action="store_true",
                    help="also time the host-pointer path (PCIe inclusive), reported separately")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run "
              f"--nproc-per-node {args.gpus} (WORLD_SIZE={world})", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the engine has no CPU fallback", file=sys.stderr)
        return 2
    dev_index = local_rank % torch.cuda.device_count()   # == local_rank on a real multi-GPU node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # rehearsal on one GPU: GNNVC_BENCH_ONE_RANK_GROUP=1 runs the partitioned path (pieces, packed exchange, RCCL
    # calls on a one-rank group) instead of the single-GPU forward
    multi = world > 1 or bool(os.environ.get("GNNVC_BENCH_ONE_RANK_GROUP"))
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import gnn_mwvc_amd as G
    from tools import graphgen_torch as ggt

    t0 = time.time()
    g, workload_desc = build_workload(args.workload, ggt, dev)   # every rank builds the same graph (same Philox stream)
    torch.cuda.synchronize()
    t_gen = time.time() - t0

    eng = G.Engine(G.default_model_text(), device=dev_index)
    assert eng.fused and eng.num_stages == 3
    eng.set_weight_scale(g.ws)
    if args.no_blocked:
        eng.set_option("blocked_stage0", 0)
    if args.no_lds_table:
        eng.set_option("lds_table", 0)
    if args.no_compact:
        eng.set_option("compact_gather", 0)
    if args.no_overlap:
        eng.set_option("overlap_dense", 0)
    if args.block_cols:
        eng.set_option("block_cols", args.block_cols)
    if args.long_threshold >= 0:
        eng.set_option("long_row_threshold", args.long_threshold)
    if args.mfma >= 0:
        eng.set_option("mfma_dense", args.mfma)
    if args.giant_threshold >= 0:
        eng.set_option("giant_row_threshold", args.giant_threshold)
    if args.hub_mode:
        eng.set_option("hub_mode", 1)
    eng.set_option("side_streams", args.side_streams)
    eng.set_option("sorted_tiles", args.sorted_tiles)
    if args.sorted_long_threshold > 0:
        eng.set_option("sorted_long_row_threshold", args.sorted_long_threshold)
    t0 = time.time()
    eng.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(),
                            g.nw.data_ptr(), keepalive=g)
    eng.synchronize()
    t_attach = time.time() - t0   # includes building the column-blocked index (once per graph)
    # a dedicated HIP stream shared by torch (events, collectives) and the engine;
    # torch's default stream has the NULL handle, which the ABI reads as "engine's own"
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    eng.set_stream(stream.cuda_stream)

    x = g.x().contiguous()

    # 1-D vertex partition: equal 64-aligned row blocks (ER is degree-uniform, so this is
    # nnz-balanced too); rank r owns rows [lo, hi).  Buffers, stage sequencing and the
    # inter-stage exchange live in gnn-mwvc_amd/distributed.py.
    from gnn_mwvc_amd import distributed as D
    # skewed graphs (R-MAT, power-law): balance CSR entries, not rows — shards become uneven and are
    # exchanged by direct sends; degree-uniform graphs keep equal shards and the pipelined all-gather
    part_mode = args.partition if args.partition != "auto" else \
        ("nnz" if WORKLOADS[args.workload][0] in ("rmat", "powerlaw") else "rows")
    bounds = D.partition_bounds(g.n, world, g.rowptr if part_mode == "nnz" else None, part_mode)
    lo, hi = bounds[rank]
    bufs = D.ForwardBuffers.allocate(g.n, bounds, dev)

    def stage_fn(st, r0, r1, src, dst, lg):
        eng.stage_forward_device(st, r0, r1, src.data_ptr(), dst.data_ptr(),
                                 lg.data_ptr() if lg is not None else 0)

    # HIP events on the launch stream around every stage launch of the timed region
    # (the engine launches on this torch stream, so torch events bracket its kernels)
    stage_evt = [[torch.cuda.Event(enable_timing=True) for _ in range(6)]
                 for _ in range(args.steps)]

    # pieces per stage: with the compact-table plan over a rank's rows a piece should be a whole number of
    # 256-chunk rounds (3 at 2 ranks, 2 at 3-4 ranks on the metric graph); otherwise 4
    use_prepare = bool(args.prepare_input) and 2 <= world <= 3   # (from 4 ranks on a rank's pieces are too small to gain)
    chunks = args.pipeline_chunks if args.pipeline_chunks >= 0 else ((3 if world == 2 else 2) if use_prepare else 4)
    prepare_fn = (lambda st, src, r0, r1: eng.stage_input_ready(st, src.data_ptr(), r0, r1)) if use_prepare else None
    piece_rows = [0]   # set after the first forward: pieces of 256 of the engine's chunks, so that no piece ends inside one
    fwd_scores = torch.zeros(g.n, dtype=torch.float32, device=dev)
    engine_stage_ms = []

    def step(k: int | None):
        if not multi:
            # one GPU: the whole forward inside the engine (its own feature buffers; per-stage HIP events on this
            # stream are read back after the timed region)
            eng.forward_device(x.data_ptr(), fwd_scores.data_ptr(), 0)
            if k is not None and k == args.steps - 1:
                torch.cuda.synchronize()
                engine_stage_ms.append(eng.last_forward_ms()[1])
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
        # verify=False: the dead-column check of the compressed exchange is read once, after the loop
        D.partitioned_forward(stage_fn, 3, x, bufs, bounds, rank, on_stage=hook, gather_logits=False,
                              replicate_stage0=None if args.replicate_stage0 < 0 else bool(args.replicate_stage0),
                              pipeline_chunks=chunks, codec=codec, verify=False, prepare_fn=prepare_fn,
                              piece_rows=piece_rows[0])

    trace = bool(os.environ.get("GNNVC_BENCH_TRACE"))
    t_start = time.time()

    def mark(what):
        if trace:
            if os.environ.get("GNNVC_BENCH_TRACE") != "2":   # "2": host-side progress only, no extra device syncs
                torch.cuda.synchronize()
            print(f"[rank {rank} +{time.time() - t_start:.1f}s] {what}", file=sys.stderr, flush=True)

    t_start = time.time()
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

    if codec is not None:
        step(None)   # first forward on this graph: full rows, records each stage's live columns (not a timed or warm-up step)
        settle()
        if use_prepare and args.pipeline_chunks < 0 and eng.get_info("compact_gather_active"):
            piece_rows[0] = 256 * eng.get_info("compact_gather_rows_per_chunk")
        mark("learning forward done")
    # setup, like building the graph: the engine builds its per-graph plans inside a graph's second and third
    # forward (host-synchronous, a few ms each) — these forwards come before the W warm-up steps, so that neither
    # the warm-up count nor the timed region decides whether the plans exist
    for i in range(3 if codec is None else 2):
        step(None)
        settle()
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
        # the engine's own HIP events (same stream) around each stage of the last timed forward; the mean step time
        # of the K timed forwards is ms_per_step
        stage_ms = list(engine_stage_ms[0])
    else:
        stage_ms = [sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for ev in stage_evt) / args.steps
                    for i in range(3)]

    ms_per_step = elapsed * 1e3 / args.steps
    edges_per_s = g.n_edges / (elapsed / args.steps)

    rows = hi - lo
    local_nnz = int(g.rowptr[hi].item()) - int(g.rowptr[lo].item())
    dom = max(range(3), key=lambda i: stage_ms[i])
    dom_bytes = stage_bytes(dom, rows, local_nnz)
    achieved = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9
    fwd_bytes = sum(stage_bytes(i, g.n, g.nnz) for i in range(3))
    # the kernels a stage launches under the plans in force (names as in profiles/*/kernel_stats.csv); the first
    # one is the stage's dominant kernel
    lt, blk, c4 = (bool(eng.get_info(k)) for k in ("lds_table_active", "blocked_stage0_active", "compact_gather_active"))
    c4 = c4 and world == 1        # (calls that cover less than half of the rows keep the gathering kernels)
    lt = lt and world <= 4        # (stage 0 is replicated — one whole-range call — up to 4 ranks; pieces take the blocked plan)
    blk = blk or (world > 4 and bool(eng.get_info("blocked_stage0_active")))
    agg_only = "false,2,false,false,true>"
    stage_kernels = [
        (["k_lt_agg", "k_lt_check_x", "k_stage_f1<32,32,16"] if lt else
         ["k_blk_accumulate", "k_stage_f1<32,32,16"] if blk else ["k_stage_f1<32,32,16"]),
        # (inside a whole forward the producing stage kernel counts and compacts: no k_column_counts, and
        # k_c4_compact leaves at once)
        (["k_c4_agg<0>", "k_c4_choose", "k_c4_compact", "k_c4_fix", "k_stage_f16<32,32,16," + agg_only] +
         (["k_column_counts"] if multi else []) if c4 else ["k_stage_f16<32,32,16,false"]),
        # (last stage: the sums one round at a time, k_dense_sigmoid of round k under the sums of round k + 1)
        ([("k_c4_agg<0>" if args.no_overlap else "k_c4_agg<1>"), "k_c4_choose", "k_c4_compact", "k_c4_fix", "k_dense_sigmoid<32,16>"] +
         (["k_column_counts"] if multi else []) if c4 else ["k_stage_f16<32,16,1,true"])]
    kernel_names = [k[0] for k in stage_kernels]

    # (the PMC summary is a single-GPU run of whole-range launches: not comparable with a rank's pieces)
    traffic, traffic_src = measured_traffic(stage_kernels[dom], args.workload) if not multi else (None, None)
    out = {
        "metric": "GNN forward edges/sec", "value": edges_per_s, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload_desc, "vertices": g.n, "edges": g.n_edges,
                   "graph": args.workload, "partition": f"1d-vertex x{world} ({part_mode}-balanced)",
                   "exchange": "none" if not multi else "all-gather of the N feature rows (16 fp32, or only their live "
                                                          "columns) after each partitioned stage, N scores at the end"},
        "roofline": {"bound": "hbm", "kernel": kernel_names[dom], "stage": dom, "stage_kernels": stage_kernels[dom],
                     "note": "achieved = algorithmic bytes of the dominant STAGE (SURVEY.md 8d, 64-byte rows, no cache "
                             "credit) / its HIP-event time; the stage is the kernels listed, the first one dominates",
                     "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": dom_bytes,
                     "kernel_ms": stage_ms[dom],
                     "forward_bytes": fwd_bytes,
                     "forward_frac": fwd_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS / world},
        "stage_ms": stage_ms, "graph_build_s": t_gen, "graph_attach_s": t_attach,
        "plan": {"compact_gather_f16": bool(eng.get_info("compact_gather_active")),
                 "lds_table_stage0": bool(eng.get_info("lds_table_active")),
                 "blocked_stage0": bool(eng.get_info("blocked_stage0_active")),
                 "column_blocks": eng.get_info("blocked_blocks"), "block_cols": eng.get_info("block_cols"),
                 "mfma_dense": eng.get_info("mfma_dense"),
                 "sorted_tiles": bool(eng.get_info("sorted_tiles_active")),
                 "natural_tile_waste": eng.get_info("tile_waste_x100") / 100.0,
                 "interleaved_tiles": bool(eng.get_info("interleaved_tiles")), "long_rows": eng.get_info("long_rows"), "long_row_threshold": eng.get_info("long_row_threshold")},
    }

    if multi:
        # self-check: the partitioned result on THIS rank against a plain single-GPU forward of
        # the same graph on this GPU (outside the timed region)
        ref_sc = torch.zeros(g.n, dtype=torch.float32, device=dev)
        ref_lg = torch.zeros(g.n, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        eng.forward_device(x.data_ptr(), ref_sc.data_ptr(), ref_lg.data_ptr())
        torch.cuda.synchronize()
        bad = (bufs.scores[: g.n].view(torch.int32) != ref_sc.view(torch.int32)).sum().to(torch.int64)
        dist.all_reduce(bad, op=dist.ReduceOp.SUM)
        out["parity"] = {"partitioned_vs_single_gpu_score_bit_mismatches_all_ranks": int(bad.item())}
        out["config"]["replicated_stages"] = \
            sorted(D.plan_replication(world, 3, bufs.live if codec is not None else None)) \
            if args.replicate_stage0 < 0 else ([0] if args.replicate_stage0 else [])
        out["config"]["pipeline_chunks"] = chunks
        out["config"]["piece_rows"] = piece_rows[0]
        out["config"]["compact_table_over_rank_rows"] = bool(use_prepare and eng.get_info("compact_gather_active"))
        out["config"]["compressed_exchange"] = None if codec is None else {
            "verified_lossless": bool(exchange_ok),
            "bytes_per_row_shipped_of_64": {str(st): (round(pk.bytes_per_row, 2) if pk is not None else 64)
                                            for st, pk in sorted(bufs.live.items())},
            "dense_columns": {str(st): (bin(pk.mask).count("1") if pk is not None else 16)
                              for st, pk in sorted(bufs.live.items())}}
        if not exchange_ok:
            out["invalid"] = "an exception list of the compressed exchange overflowed in the timed region"
    if rank == 0 and not multi:
        if args.host_path:
            # PCIe-inclusive path (host x in, host scores + logits out); never `value`
            import numpy as np
            xh = x.cpu().numpy()
            keep = eng.forward(xh)          # the caller's output buffers, reused like the reference's `out` matrix
            t1 = time.perf_counter()
            for _ in range(3):
                eng.forward(xh, out=keep)
            out["host_path_ms"] = (time.perf_counter() - t1) * 1e3 / 3
        if not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline(args, dev, eng, ggt)

    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if multi:
        dist.destroy_process_group()
    return 0


def measured_traffic(kernels, workload: str):
    """HBM/fabric bytes per forward of the dominant stage's kernels from the committed PMC summary of the latest
    round (rocprofv3 --pmc passes of tools/pmc_probe.py on the metric graph; L2->fabric read requests x 128 B,
    calibrated on a 1 GiB copy in the same run, + WRITE_SIZE; steady-state launch of each kernel).  null when the
    summary does not cover every kernel of the stage / the workload."""
    if workload != "er10m":
        return None, None
    prof = sorted((ROOT / "profiles").glob("r*/pmc_summary.json"))
    if not prof:
        return None, None
    data = json.loads(prof[-1].read_text())
    total = 0.0
    for kernel in kernels:
        want = kernel.replace(" ", "")
        hit = [c for name, c in data.items() if name.replace(" ", "").startswith(want) and "traffic_bytes" in c]
        if not hit:
            return None, None
        total += hit[0]["traffic_bytes"]
    return total, str(prof[-1].relative_to(ROOT))


def cpu_baseline(args, dev, eng, ggt):
    """The oracle (bit-equal CPU port of the reference path, as shipped: serial
    aggregation, threaded dense layers) timed on this host, on a bounded sample of
    the same graph family; the GPU logits on the same sample are checked against it."""
    import numpy as np
    import torch
    import gnn_mwvc_amd as G
    from oracle import oracle_py

    sn, sm = (int(v) for v in args.cpu_sample.split("x"))
    gs = ggt.erdos_renyi(sn, sm, 99, dev)
    hg = gs.to_host()
    om = oracle_py.OracleModel(G.default_model_text())
    om.set_weight_scale(hg.ws)
    xh = hg.x()
    om.predict(hg, xh)                         # warm-up (page faults, thread pool)
    times = []
    want = None
    for _ in range(2):
        t0 = time.perf_counter()
        want = om.predict(hg, xh, stop_after=om.n_layers - 2)[:, 0]
        times.append(time.perf_counter() - t0)
    t_cpu = sorted(times)[0]
    # parity of the GPU path on the same sample
    eng.set_weight_scale(gs.ws)
    eng.attach_graph_device(gs.n, gs.nnz, gs.rowptr.data_ptr(), gs.col.data_ptr(),
                            gs.w.data_ptr(), gs.nw.data_ptr(), keepalive=gs)
    sc = torch.zeros(gs.n, dtype=torch.float32, device=dev)
    lg = torch.zeros(gs.n, dtype=torch.float32, device=dev)
    eng.forward_device(gs.x().contiguous().data_ptr(), sc.data_ptr(), lg.data_ptr())
    eng.synchronize()
    got = lg.cpu().numpy()
    mism = int((got.view(np.uint32) != want.view(np.uint32)).sum())
    base = {"value": hg.n_edges / t_cpu, "unit": "edges/s", "cores": oracle_py.num_threads(),
            "kind": "port",
            "sample": f"erdos-renyi n={sn} m={hg.n_edges} (same generator, seed 99), one forward, "
                      f"best of 2 after a warm-up; aggregation serial as the reference ships it, "
                      f"dense layers on {oracle_py.num_threads()} OpenMP threads",
            "seconds": t_cpu}
    parity = {"sample_vertices": sn, "logit_bit_mismatches_vs_oracle": mism}
    return base, parity


if __name__ == "__main__":
    sys.exit(main())
