/*
 * gnnvc.h — C ABI of the MI355X GNN-VC scoring engine (libgnnvc_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of KennethLangedal/GNN-MWVC:
 * the GNN forward `gnn::model::predict` and the layer `forward()`s it runs
 * (reference src/gnn_inference.cpp:20-81, include/gnn_inference.hpp:11-59) plus
 * the OpenBLAS seam `dot()` (reference src/matrix.cpp:106-122).  Plain
 * pointers and sizes only; no C++ or torch types.  Every entry point returns
 * GNNVC_OK (0) or a negative error code, never throws, and treats N = 0 as a
 * successful no-op (the reference calls predict with an empty graph at the end
 * of every run, src/GNN_VC.cpp:192 / SURVEY.md §3.2).
 *
 * There is no CPU fallback behind this ABI: without a HIP device every compute
 * entry point returns GNNVC_ERR_DEVICE.
 *
 * Numerics contract: logits — the input of the final sigmoid —
 * are bit-identical to the reference's on the same inputs (CSR-order fp32
 * neighbour sums, sequential-k fused-multiply-add chains, separately rounded
 * bias adds).  Device-side scores apply 1/(1+exp(-x)) with a device exp that is
 * a restatement of glibc's expf; hosts that need the reference's exact scores
 * apply their own libm to the logits (the C++ wrapper does).
 *
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef GNNVC_H
#define GNNVC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNVC_ABI_VERSION 1

enum {
    GNNVC_OK = 0,
    GNNVC_ERR_INVALID = -1,     /* bad argument / malformed model text */
    GNNVC_ERR_DEVICE = -2,      /* no HIP device, or a HIP call failed */
    GNNVC_ERR_NOMEM = -3,       /* host or device allocation failed */
    GNNVC_ERR_STATE = -4,       /* call out of order (e.g. forward before a graph) */
    GNNVC_ERR_UNSUPPORTED = -5  /* model / size outside what the engine handles */
};

typedef struct gnnvc_engine gnnvc_engine;

int gnnvc_abi_version(void);
const char *gnnvc_strerror(int code);
/* Detail of the last failure on this engine (empty string if none). */
const char *gnnvc_last_error(const gnnvc_engine *e);

/* ---- model ----------------------------------------------------------------
 * gnnvc_create replaces `istream >> gnn::model` (reference
 * src/gnn_inference.cpp:120-139 + src/matrix.cpp:97-104): `model_text` is the
 * reference's text format (`<name> <n> Layers`, then Linear_Layer / Graph_Layer /
 * ReLU_Activation / Sigmoid_Activation records).  `device` is the HIP device
 * ordinal.  The parameters are uploaded once. */
int gnnvc_create(gnnvc_engine **out, const char *model_text, size_t len, int device);
void gnnvc_destroy(gnnvc_engine *e);

/* Several devices behind ONE handle (SURVEY.md §8b's `n_devices`; §5: "one process driving 8 devices").  The reference has one
 * call site, m.predict(x, out, g) on one thread (src/GNN_VC.cpp:192, include/gnn_inference.hpp:50): a drop-in that wants more
 * than one GPU has to partition behind that call, and this handle does.  devices[0 .. n_devices) are HIP ordinals (an ordinal
 * may repeat: several parts then share a device — how a one-GPU machine rehearses the path).  On such a handle
 *   gnnvc_upload_graph / the staged hand-off   cut the graph into n_devices contiguous row ranges of equal entry count (multiples
 *                         of 64 rows) and give device r the CSR slice of its rows (global column ids) — 1 / n_devices of the
 *                         graph's memory each — next to full-size replicated feature buffers;
 *   gnnvc_forward / gnnvc_forward_device        run every stage range by range — every device driven by a host thread of its own —
 *                         and, after the first and second stage, let each device send the rows it computed to every peer: in
 *                         pieces, packed to their live columns (gnnvc_pack_rows: 16 bytes a row on the metric graph; the
 *                         packing is chosen by the graph's first forward and checked lossless on every one), each piece stored
 *                         into the peers' receive buffers by one kernel (gnnvc_push_piece: a direct all-gather, one xGMI link
 *                         per peer, no ring, no host hop) under the next piece's kernels, and expanded there
 *                         (gnnvc_unpack_pieces); the scores (and logits) are assembled on devices[0].  gnnvc_forward_device
 *                         takes pointers on devices[0] and is complete when it returns — on an error too: every device is
 *                         drained first.  Options "multi_pieces" (pieces per part and stage; default 1 up to 4 devices, where
 *                         the per-range plans want whole ranges, 4 beyond), "multi_pack" 0 = full rows, "multi_push" 0 =
 *                         hipMemcpyPeerAsync per peer and piece, "multi_only_part" r = a forward runs part r's share only (a
 *                         timing rehearsal where parts share a device; its results are not complete), "multi_announce" -1|0|1 = a
 *                         part that runs a 16-wide stage as one piece announces the stage's complete input (compact table over
 *                         its rows: gnnvc_stage_input_ready; -1 = up to four devices).  A forward whose exception lists
 *                         overflow is repeated — up to three attempts, each with full rows for the stage that overflowed —
 *                         before it returns.  gnnvc_get_info: "multi_exchange_bytes_per_peer_stage0", "..._stage1" (bytes the
 *                         busiest part's pushes of the last forward wrote to one peer), "multi_packed_stage0", "..._stage1"
 *                         (is the stage's exchange packed), "multi_packed_columns_stage0", "..._stage1", "multi_pieces",
 *                         "multi_peer_stores" (every pair of devices can store into each other's memory),
 *                         "multi_part_span_us_<r>";
 *   gnnvc_set_weight_scale / gnnvc_set_option / gnnvc_synchronize / gnnvc_score_keys / gnnvc_last_forward_ms (total only)
 *                         and the layer-level entry points without a graph (linear, relu, sigmoid, sgemm, stream_sum) work
 *                         as on any handle; gnnvc_get_info adds "devices", "part_rows_<r>", "part_entries_<r>";
 *   the entry points that read ONE device's resident graph (attach_graph_device / _slice, stage_forward_device,
 *                         stage_input_ready, derive_graph_*, graph_row_hashes, reduction_flags, graph_layer_forward) return
 *                         GNNVC_ERR_UNSUPPORTED.
 * Results are those of a single device, bit for bit: a row is summed on one device in stored order.  n_devices = 1 is a
 * single engine behind the same code path. */
int gnnvc_create_multi(gnnvc_engine **out, const char *model_text, size_t len, const int *devices, int n_devices);

/* model::set_weight_scale (reference src/gnn_inference.cpp:83-90): sets
 * graph_layer::WEIGHT_SCALE of every graph layer (default 120). */
int gnnvc_set_weight_scale(gnnvc_engine *e, float ws);

/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of
 * the engine's own stream.  NULL restores the engine's stream.  Call it between forwards.  The engine's side queue (long and
 * giant rows, plan builders) has to sit on another hardware queue than the main stream; it is probed against the new one
 * (~0.3 ms) and replaced if the runtime put the two on the same queue: gnnvc_get_info "side_queue_runs_beside" says whether a
 * queue beside the main stream was found (0: skewed graphs' side work runs serialised — slower, never wrong). */
int gnnvc_set_stream(gnnvc_engine *e, void *hip_stream);
/* The stream the engine launches on at the moment (its own unless gnnvc_set_stream installed another): a caller orders its own
 * work behind the engine's asynchronous calls with it. */
int gnnvc_get_stream(gnnvc_engine *e, void **hip_stream);

/* Tuning options (take effect at the next graph upload / attach):
 *   "plans_at_handoff" 0|1|2  WHEN the per-graph plans below are built.  The reference's driver hands predict a new graph every
 *                         call and scores it once (src/GNN_VC.cpp:171-192), so what depends on the graph alone is built when
 *                         the graph is handed over (upload / staged commit / attach), not inside a later forward: 1 (default) =
 *                         the plans that ONE forward repays (degree-uniform graphs of at least "handoff_min_entries"
 *                         adjacency entries, default 24 Mi: LDS table + compact table; every graph: its tile order and
 *                         every buffer a forward would otherwise allocate), 2 = every plan the graph qualifies for whatever
 *                         it costs (callers who score a graph many times, or hide the build under a copy), 0 = inside the
 *                         graph's first two forwards (round 2's behaviour).  gnnvc_get_info "handoff_build_us" = what it took
 *   "pilot_rows"     n    first use of the compact-table plan on a graph: the producing stage's plain kernel over its first n
 *                         rows (default 65536, 0 = off) picks the next stage's table columns ahead of the real run, which
 *                         then writes the table on its way; the choice is re-made from the counts of all rows, so the pilot
 *                         changes time, never a result
 *   "filter_zero_rows" 0|1  a large skewed graph's FIRST forward (the reference's driver never comes back for a second): its
 *                         16-wide stages look every entry's target up in the bitmap of this input's all-zero rows and fetch the
 *                         pad row instead, the long rows walk lists a pass in front of them shortened — nothing is built, the
 *                         pruned adjacency ("prune_zero_rows") waits for a second forward (default 1; bit-identical either
 *                         way).  "filter_min_entries" (default 48 Mi) / "filter_min_long_percent" (25: the share of the
 *                         entries in long rows) say which graphs, "filter_min_percent" (50) from which share of the entries
 *                         pointing to zero rows the device uses the bitmap, "filter_keep_lists" 0 = every stage shortens the
 *                         full rows.  gnnvc_get_info: "filtered_stage1|2", "short_lists_stage2", "filter_mass_percent_stage1|2"
 *   "long_rows_on_main" -1|0|1  where the long rows' kernels run: the engine has ONE side queue beside its main stream (probed at
 *                         creation to sit on another hardware queue: gnnvc_get_info "side_queue_probes" / "side_queue_runs_beside");
 *                         the giant rows always take it, the long rows join them there (0) or run ahead of the tile kernel on the
 *                         main stream (1); -1 (default) = by the graph: on the main stream when the longest giant row's walk is
 *                         what a stage waits for.  "giant_gather_first" -1|0|1: the giant rows' gather on the main stream ahead
 *                         of the tile kernel (1) or on the side queue with the rest of their chain (0); -1 (default) = on the
 *                         main stream when it is small (at most 16 Mi giant entries) and the long rows are on the side queue
 *   "blocked_stage0" 0|1  column-blocked plan of the F = 1 stage (default 1; results are
 *                         bit-identical either way, it only changes memory traffic)
 *   "lds_table"      0|1|2  LDS-table plan of the F = 1 stage: when the input is x[v] = (float)W(v)/ws with integer
 *                         W(v) <= 65 535 (checked on the device at every forward), the neighbour values are read from a
 *                         table held slice by slice in LDS instead of gathered from memory — a byte per vertex for weights up
 *                         to 255, ten bits (three to a word) up to 1023, sixteen bits beyond (round 4; chosen per graph from
 *                         its largest weight and the weight scale: gnnvc_get_info "lds_table_bits"; "lds_table_bits" 8|10|16
 *                         forces a width, 0 = by the graph; "lds_table_min_chunks": how finely a short row range — a rank's
 *                         slice — is cut at least, default 128).  Default 1 = large graphs, skewed ones in their own layout
 *                         (byte table only); 2 = the consecutive-row layout on any large graph; 0 = off.
 *                         Takes precedence over "blocked_stage0"; bit-identical results
 *   "compact_gather" 0|1|2  compact-table plan of the 16-wide stages: when at most four feature columns carry
 *                         (nearly) all non-zeros of a stage's input — decided on the device at every forward —
 *                         neighbours are read from a 16-byte-per-vertex table swept block by block through L2
 *                         instead of 64-byte rows from memory (default 1 = large, non-skewed graphs); bit-identical
 *   "block_cols"     n    vertices per column block (default 524288 = 2 MiB of x)
 *   "blocked_min_n"  n    graphs with fewer vertices get none of the per-graph plans of the F = 1 stage (default 2^20)
 *   "compact_min_n"  n    ... nor the compact-table plan below the smaller of this (default 2^18) and "blocked_min_n"
 *                         (with default bounds the plans also want entries: 8 Mi for the compact table, 10 per row for the
 *                         F = 1 plans)
 *   "compact_first_forward_entries" n  the plans are built inside a graph's second forward; a graph with at least this
 *                         many adjacency entries builds the compact-table plan inside its FIRST forward, which one use
 *                         repays there (default 48 Mi; 0 = never) — what a caller gets who scores every graph once
 *   "overlap_dense"  0|1  last stage under the compact-table plan: dense layers of one round of the sums on a second
 *                         stream, under the next round's sums (default 1; bit-identical either way)
 *   "plan_chunk_rows" n   cap on the rows per chunk of the LDS-table and compact-table plans (default 0 = what
 *                         fits LDS); smaller chunks mean more rounds of the persistent grids — a test hook
 *   "long_row_threshold" d  rows of degree >= d get a workgroup of their own (0 = off); same CSR-order sums,
 *                         bit-identical results.  Default: by the graph — 256 where at most 16 Ki rows (and one row in
 *                         64) are that long, 512 otherwise; setting it (or "sorted_long_row_threshold") fixes it
 *   "lds_table_skewed" 0|1, "lds_table_skewed_rows" d  the LDS-table plan on SKEWED graphs (default 1): rows below d
 *                         entries dealt to slices of equal weight, column blocks of equal entry mass, giant rows (and
 *                         the rows from d on) beside it; d = 0 (default): 2048 while x (4 N bytes) fits the L2s
 *                         together, else everything below the giant rows.  Bit-identical
 *   "giant_row_threshold" d  long rows of degree >= d (default 16384, 0 = off) are summed by many waves at once:
 *                         their neighbour values are gathered by every CU into per-column streams and each stream's
 *                         sequential fp32 sum is evaluated with a parallel scan that reproduces the chain's
 *                         roundings (csrc/exact_sum.h); bit-identical results
 *   "giant_row_threshold_f16" d  the 16-wide stages send only rows of degree >= d the giant way.  Default: 65536 on graphs
 *                         of 64 Mi adjacency entries and more (sixteen streams per row are three times a long row's
 *                         traffic: worth it for the rows whose add chain a stage would wait for), "giant_row_threshold"
 *                         on smaller ones (there a 65 536-entry chain IS what a stage waits for); setting
 *                         "giant_row_threshold" sets this one too
 *   "giant_segments" -1|0|1  1 = a giant row's streams are cut into segments of 4096 addends summarised by many waves
 *                         at once (k_giant_segmap) and joined by one walk; 0 = one wave walks each stream; -1 (default) =
 *                         segments where the longest stream's walk would be what a stage waits for.  Same bits
 *   "side_streams"   0|1  1 (default) = long / giant rows on side streams beside the tile kernel; 0 = on the main
 *                         stream, one after the other (profiling: standalone kernel times)
 *   "kernel_trace"   0|1  HIP events around every main-stream kernel of a forward (gnnvc_kernel_trace)
 *   "mfma_dense"     0|1|2  dense layers on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, a
 *                         k-ordered fma chain: same bits as the VALU path): 0 = VALU, 1 = MFMA in
 *                         every stage, 2 = MFMA in the 16-wide stages only (default); immediate
 *   "sorted_tiles"   -1|0|1  16-wide stages take their 64-vertex tiles from a degree-sorted
 *                         vertex list instead of 64 consecutive rows (-1 = only when natural tiles
 *                         would spend more than twice the useful gather rounds, the default)
 *   "sorted_min_nnz" n     in auto mode, graphs with fewer adjacency entries keep natural tiles
 *                         (default 4 Mi: the sort costs more than it saves on a graph used once)
 *   "prune_zero_rows" 0|1  pruned adjacency of the 16-wide stages: adjacency entries whose target row is all zero in
 *                         the stage's input are left out of a second CSR (adding a row of zeros changes no bit of a
 *                         sum); EVERY call proves on the device that its input fits, else it uses the full adjacency.
 *                         1 (default) = the set of vertices is the rows found all zero when the plan is built (a
 *                         graph's stage inputs follow from its weights), 0 = off.  Bit-identical results
 *   "prune_predict"  0|1  (round 4) a large skewed graph — at least "prune_predict_min_entries" adjacency entries, default
 *                         48 Mi, "filter_min_long_percent" of them in long rows — gets the first 16-wide stage's pruned
 *                         adjacency when it is HANDED OVER, from the set of zero rows its own weights predict (the reference's
 *                         driver feeds x = W / ws, src/GNN_VC.cpp:189-191), proven per call like any other set; a graph scored
 *                         once runs its first forward on it and the next stage borrows it.  Default 1.  gnnvc_get_info:
 *                         "pruned_predicted_stage1", "pruned_borrowed_stage2"
 *   "dense_skip_zeros" 0|1  (round 4; A/B) the aggregate-only dense kernels of the compact-table plan leave out the first-layer
 *                         terms they know to be zero (default 1; bit-identical)
 *   "forward_timing" 0|1|2  (round 4) HIP events of a forward for gnnvc_last_forward_ms: 0 (default) = none — a record costs the
 *                         stream ~1.8 us, four of them were 5.5 us of a 30 - 80 us forward —, 1 = the forward's first and last
 *                         (total only), 2 = one per stage too
 *   "verdict_period" 1..64  (round 4) once four verdicts of the per-graph plans in a row have changed nothing, only every this-many
 *                         forwards ask for them (default 8; a verdict steers which kernels the NEXT forwards launch, never a result)
 *   "poison_features" 0|1  (tests, fuzz) a whole forward starts by filling the engine's own feature buffers (a multi-device
 *                         handle: every part's) with NaN bit patterns, so that a row no kernel writes — or no exchange delivers —
 *                         shows in the result instead of hiding behind what an earlier forward left there.  Default 0.
 *   "wide_tiles"     0|1  (round 4) graphs with fewer 64-vertex tiles than the chip has SIMDs — the reference CLI's later predict
 *                         calls — run a stage a WORKGROUP per tile (the gather on quads of lanes over four waves, each dense
 *                         layer's outputs a quarter per wave: the same fma chains, the same bits): the F = 1 stage up to
 *                         "wide_tiles_max_n" vertices (default 49 152), the 16-wide stages up to "wide_tiles_max_n_f16"
 *                         (default 131 072) where no plan has the stage; graphs without long rows.  Default 1.
 *                         gnnvc_get_info "wide_tiles_used"
 *   "table_tiles"    0|1  (round 4) graphs of "table_tiles_min_n" (default 49 152) vertices and more whose 16-byte-per-vertex
 *                         table fits "table_tiles_max_bytes" (default 6 MiB: n <= 393 K) and that are outside the compact-table
 *                         plan's range, without long rows: whole forwards gather the 16-wide stages' neighbours from an
 *                         L2-resident table of the input's four live columns, written by the kernel that produced the input
 *                         for the columns the previous forward chose (kept across graphs of one engine); vertices with
 *                         non-zeros elsewhere are flagged and their neighbours fetch those columns from the full rows —
 *                         bit-identical for any input.  Offered from a graph's second forward on, or its first when the
 *                         engine carries a choice over from its previous graph.  Default 1.  "table_tiles_solo" 0|1 (A/B):
 *                         0 = the gathering kernel is always launched behind the tiles.  gnnvc_get_info:
 *                         "table_tiles_active", "table_tiles_fit_stage1|2" (did the last forward's stage run on its table)
 *   "prune_min_entries" n  graphs with fewer adjacency entries do not prune (default 2^20);
 *   "prune_min_drop_percent" p  nor do graphs where less than p % of the entries would go (default 15);
 *   "prune_early_entries" n  skewed graphs with at least n entries build the plan in their first forward (default
 *                         2^26; 0 = always with the other plans, in the second)
 *   "prune_class_by_entries_left" 0|1, "prune_heavy_entries" n, "prune_giant_rows" 0|1  A/B switches of the plan: rows
 *                         classed by the entries they have left (default 1), the entry count from which the tile
 *                         kernel keeps rows of up to "sorted_long_row_threshold" entries (default 2^24; below: 512),
 *                         giant rows pruned too (default 1)
 * gnnvc_get_info keys (further): "pruned_stage1|2", "pruned_entries_stage1|2", "pruned_vertices_stage1|2",
 * "pruned_last_ok_stage1|2" (did the last call of that stage use its pruned adjacency),
 * "compact_gather_last_ok", "compact_gather_last_passes", "compact_gather_last_dirty",
 * "compact_gather_blocks", "compact_gather_steps", "plan_build_us".
 * gnnvc_get_info keys: "compact_gather_active", "compact_gather_chunks", "lds_table_active", "lds_table_chunks", "lds_table_steps", "mfma_dense", "sorted_tiles_active", "tile_waste_x100", "blocked_stage0_active", "blocked_blocks", "block_cols", "long_rows",
 * "long_row_threshold", "giant_rows", "giant_entries", "giant_row_threshold". */
int gnnvc_set_option(gnnvc_engine *e, const char *key, long value);
int gnnvc_get_info(const gnnvc_engine *e, const char *key, long *value);

/* Model introspection (what model::layers holds). */
int gnnvc_num_layers(const gnnvc_engine *e);
/* 1 if the model matches the fused 3-stage plan, 0 if it runs layer by layer. */
int gnnvc_is_fused(const gnnvc_engine *e);
/* Width of the forward's input / output rows. */
int gnnvc_in_width(const gnnvc_engine *e);
int gnnvc_out_width(const gnnvc_engine *e);

/* ---- graph hand-off ---------------------------------------------------------
 * What the forward reads from reduction_graph<uint32_t,uint32_t> (reference
 * include/reduction_graph.hpp: size() :141, begin(u)/end(u) :693-704, D :144,
 * W :147-151, NW :154-158): vertices 0..n-1, adj(u) = col[rowptr[u]..rowptr[u+1])
 * in the graph's stored order (that order is the fp32 summation order), the
 * vertex weights and the neighbourhood weights.  Host pointers; copied to the
 * device.  nnz = rowptr[n] must be < 2^32. */
int gnnvc_upload_graph(gnnvc_engine *e, uint32_t n, const uint64_t *rowptr,
                       const uint32_t *col, const uint32_t *w, const uint32_t *nw);

/* Zero-copy variant for callers whose CSR already lives in device memory
 * (multi-GPU shards, device-side generators): d_rowptr is uint32[n+1], d_col is
 * uint32[nnz + GNNVC_COL_PAD] (the pad entries may hold anything), d_w / d_nw are
 * uint32[n].  The caller keeps them alive until the next attach/upload/destroy. */
#define GNNVC_COL_PAD 64
int gnnvc_attach_graph_device(gnnvc_engine *e, uint32_t n, uint64_t nnz,
                              const uint32_t *d_rowptr, const uint32_t *d_col,
                              const uint32_t *d_w, const uint32_t *d_nw);

/* One rank's SLICE of a vertex-partitioned graph ("each GPU holds its rows' CSR slice with global
 * column ids, its slice of W / NW").  The engine then holds rows [row_lo, row_hi) of a graph of n_global vertices:
 *   d_rowptr_local  uint32[row_hi - row_lo + 1], relative to the slice (first = 0, last = nnz_local)
 *   d_col_local     uint32[nnz_local + GNNVC_COL_PAD], GLOBAL column ids, stored order
 *   d_w_local, d_nw_local  uint32[row_hi - row_lo]
 * — 1 / P of the graph's memory at P ranks; the feature matrices stay full-size and replicated (the caller's).  A
 * sliced engine is driven stage by stage with gnnvc_stage_forward_device on sub-ranges of its slice; the whole-graph
 * entry points (gnnvc_forward*, gnnvc_reduction_flags, gnnvc_graph_layer_forward) return GNNVC_ERR_STATE, and the
 * per-graph plans that index whole graphs stay off.  Results are those of the whole graph: a row is summed on one
 * GPU in stored order.  No reference counterpart (the reference is one process reading one graph,
 * src/gnn_inference.cpp:32-41).  The caller keeps the arrays alive until the next attach / upload / destroy. */
int gnnvc_attach_graph_slice(gnnvc_engine *e, uint32_t n_global, uint32_t row_lo, uint32_t row_hi, uint64_t nnz_local,
                             const uint32_t *d_rowptr_local, const uint32_t *d_col_local, const uint32_t *d_w_local,
                             const uint32_t *d_nw_local);

/* Staged hand-off (the caller is the pack loop over reduction_graph's
 * begin(u)/end(u) ranges, include/reduction_graph.hpp:693-704): instead of building a temporary
 * CSR and passing it to gnnvc_upload_graph, the caller writes the arrays straight into page-locked
 * memory owned by the engine.  Protocol:
 *   1. gnnvc_graph_staging(e, n, 0, &rowptr, NULL, &w, &nw): fill w[n], nw[n] and rowptr[n+1]
 *      (32-bit, rowptr[0] = 0, rowptr[n] = nnz);
 *   2. gnnvc_graph_staging(e, n, nnz, NULL, &col, NULL, NULL): the vertex arrays keep their contents;
 *      fill col[nnz] in any order;
 *   3. optionally, whenever a prefix-contiguous piece [first, first+count) of col is final,
 *      gnnvc_staged_columns_ready(e, first, count) starts its copy while the caller packs on
 *      (pieces must be announced in ascending order without gaps);
 *   4. gnnvc_commit_staged_graph(e) copies the rest, runs the same device-side checks as
 *      gnnvc_upload_graph and makes the graph current.
 * The pointers stay valid until the next gnnvc_graph_staging call with larger sizes.
 * An error from gnnvc_staged_columns_ready (e.g. GNNVC_ERR_INVALID for row pointers that are not monotone from 0 to nnz,
 * which a large hand-off checks before it starts building from them) ends the hand-off: no graph is current, begin again
 * at step 1. */
int gnnvc_graph_staging(gnnvc_engine *e, uint32_t n, uint64_t nnz, uint32_t **rowptr, uint32_t **col,
                        uint32_t **w, uint32_t **nw);
int gnnvc_staged_columns_ready(gnnvc_engine *e, uint64_t first, uint64_t count);
int gnnvc_commit_staged_graph(gnnvc_engine *e);

/* The NEXT graph derived from the one the device already holds.  Between two predict calls of the
 * reference's driver (src/GNN_VC.cpp:171-192) the graph shrinks — vertices leave, lists lose entries, survivors are
 * renumbered in order (include/reduction_graph.hpp:537-587) — and the only additions are the vertices its folds
 * create, which carry the largest ids and therefore sit at the END of their neighbours' lists (:335-398).  So a row of
 * the next graph is: the surviving entries of its old row, in order, renumbered — plus a short tail; a new vertex's
 * row is all tail.  Protocol (the resident graph must have come through gnnvc_upload_graph / the staged calls / a
 * previous derive, i.e. live in engine-owned memory):
 *   1. gnnvc_derive_graph_begin(e, n_new, old_row, rowptr_new, tail): old_row[u] = the row of the RESIDENT graph that
 *      vertex u of the next graph was, or GNNVC_NEW_VERTEX; rowptr_new = the next graph's 32-bit row pointers (n_new + 1).
 *      The device counts every row's surviving old entries and returns tail[u] = degree(u) - survivors: how many
 *      trailing entries of u's list it cannot derive.  An inconsistent mapping (a row with more survivors than its new
 *      degree, an old row claimed twice, ...) is GNNVC_ERR_INVALID and leaves the resident graph as it was;
 *   2. gnnvc_derive_graph_commit(e, tail_cols, n_tail, w, nw): the last tail[u] entries of every row, row after row
 *      (n_tail = their total), and the next graph's W / NW.  The engine assembles the CSR on the device, runs the same
 *      checks as an upload and makes it current.
 * What crosses the bus: 8 bytes per vertex + the tails, instead of 4 bytes per adjacency entry.  What the engine
 * cannot check is that the caller's lists really ARE "survivors + tail" (it never sees them): gnnvc_graph_row_hashes
 * returns a 64-bit FNV-1a hash of every resident row (column ids in stored order) for callers that want to compare. */
#define GNNVC_NEW_VERTEX 0xFFFFFFFFu
int gnnvc_derive_graph_begin(gnnvc_engine *e, uint32_t n_new, const uint32_t *old_row, const uint32_t *rowptr_new, uint32_t *tail);
int gnnvc_derive_graph_commit(gnnvc_engine *e, const uint32_t *tail_cols, uint64_t n_tail, const uint32_t *w, const uint32_t *nw);
int gnnvc_graph_row_hashes(gnnvc_engine *e, uint64_t *hashes);

/* ---- forward ----------------------------------------------------------------
 * gnnvc_forward replaces model::predict (reference src/gnn_inference.cpp:67-81)
 * for host callers: x is n x in_width (n x 1 for the shipped model:
 * x[u] = (float)W(u)/ws, reference src/GNN_VC.cpp:189-191), scores is
 * n x out_width.  `logits` (optional, may be NULL) receives the input of the
 * final sigmoid when the model ends in one. */
int gnnvc_forward(gnnvc_engine *e, const float *x, float *scores, float *logits);

/* Device-resident forward: all pointers are device memory; asynchronous on
 * the engine's stream.  d_logits may be NULL. */
int gnnvc_forward_device(gnnvc_engine *e, const float *d_x, float *d_scores, float *d_logits);

/* One fused stage (graph layer + the dense layers up to the next graph
 * layer) over the vertex range [row_lo, row_hi) — the unit a 1-D
 * vertex-partitioned multi-GPU run executes between feature exchanges.
 * d_in is the full (n + 1) x in_width feature matrix whose LAST row is all
 * zeros (the gather reads it for masked lanes); d_out is the full
 * (n + 1) x out_width matrix of which only rows [row_lo, row_hi) are written.
 * d_logits (stage with a final sigmoid only, may be NULL) likewise.
 * Asynchronous on the engine's stream. */
int gnnvc_num_stages(const gnnvc_engine *e);
int gnnvc_stage_widths(const gnnvc_engine *e, int stage, int *in_width, int *out_width);
int gnnvc_stage_forward_device(gnnvc_engine *e, int stage, uint32_t row_lo, uint32_t row_hi,
                               const float *d_in, float *d_out, float *d_logits);

/* ---- feature-row codec for the exchange between vertex-partitioned GPUs --------
 * No reference counterpart (the reference is single-process).  After the ReLU that ends a fused
 * stage most entries of the N x 16 feature matrix are zero: on the metric graph two columns are
 * non-zero in every row, two in about a tenth of the rows, five in a handful of rows and seven in
 * none (which columns, and how dense, depends on the graph).  A piece of rows [row_lo, row_hi)
 * travels as
 *   d_dense  (row_hi - row_lo) x kp floats: the columns of `mask` in ascending order, zero padded
 *            (kp = 4, 8 or 12), and
 *   d_exc    an exception list for non-zeros in any other column: word 0 = count, then entries of
 *            four words {row - row_lo, column, value bits, 0} from word 4 on, room for exc_cap
 *            entries (4 + 4 * exc_cap words); may be NULL.
 * gnnvc_unpack_rows rebuilds the 16-float rows, writing +0.0f where nothing was shipped — exactly
 * the rows a full exchange would have delivered.  gnnvc_pack_rows ORs into *d_flag: bit 0 if a
 * non-zero fell outside `mask` and there was no list, bit 1 if the list overflowed; the caller then
 * falls back to full rows.  gnnvc_column_counts gives the per-column non-zero counts from which
 * the caller chooses mask, kp and exc_cap (synchronous); the other calls are asynchronous on the
 * engine's stream.  All pointers are device memory. */
int gnnvc_live_columns(gnnvc_engine *e, const float *d_feat, uint32_t rows, uint32_t width, uint32_t *mask);
int gnnvc_column_counts(gnnvc_engine *e, const float *d_feat, uint32_t rows, uint32_t width, uint64_t *counts /*[16]*/);
int gnnvc_pack_rows(gnnvc_engine *e, const float *d_feat, uint32_t width, uint32_t row_lo, uint32_t row_hi,
                    uint32_t mask, uint32_t kp, float *d_dense, uint32_t *d_exc, uint32_t exc_cap, uint32_t *d_flag);
int gnnvc_unpack_rows(gnnvc_engine *e, const float *d_dense, const uint32_t *d_exc, uint32_t exc_cap, uint32_t width,
                      uint32_t row_lo, uint32_t row_hi, uint32_t mask, uint32_t kp, float *d_feat);

/* One all-gathered piece, every peer's region in one launch: d_buf holds `world` regions of
 * piece_words words (dense part of dense_rows rows, then the exception list), region r carrying rows
 * [r * rows_per_rank + row_off, + rows) of rank r (cut at the end of its shard and at n);
 * skip_rank's region (the caller's own rows, already in place) is left alone. */
int gnnvc_unpack_gathered(gnnvc_engine *e, const float *d_buf, uint32_t world, uint32_t skip_rank, uint64_t piece_words,
                          uint32_t dense_rows, uint32_t exc_cap, uint32_t width, uint32_t rows_per_rank, uint32_t row_off,
                          uint32_t rows, uint32_t n, uint32_t mask, uint32_t kp, float *d_feat);

/* One packed piece to several destinations, and several packed pieces expanded, in ONE launch each (round 4; what
 * gnnvc_create_multi's devices exchange).  A piece REGION = rows x kp dense floats (gnnvc_pack_rows' d_dense) directly followed
 * by its exception list (4 + 4 * exc_cap words).
 *   gnnvc_push_piece     stores the dense part and the USED part of the list into each of d_dst[0 .. n_dst) (n_dst <= 64): device
 *                        pointers this engine's device can store to — its own memory, or a peer's once peer access is enabled
 *                        (hipDeviceEnablePeerAccess: the stores then cross the xGMI link to that peer) — on hip_stream (NULL =
 *                        the engine's stream; the region must be complete in that stream's order);
 *   gnnvc_unpack_pieces  expands pieces[0 .. n_pieces) (n_pieces <= 64) — region i holding rows [row_lo, row_hi) — into the
 *                        16-column rows of d_feat exactly as gnnvc_unpack_rows would, on the engine's stream. */
typedef struct gnnvc_piece {
    const float *d_region;
    uint32_t row_lo, row_hi;
} gnnvc_piece;
int gnnvc_push_piece(gnnvc_engine *e, const float *d_region, uint32_t rows, uint32_t kp, uint32_t exc_cap, uint32_t n_dst,
                     float *const *d_dst, void *hip_stream);
int gnnvc_unpack_pieces(gnnvc_engine *e, const gnnvc_piece *pieces, uint32_t n_pieces, uint32_t exc_cap, uint32_t width, uint32_t mask,
                        uint32_t kp, float *d_feat);

/* A multi-GPU rank that computes rows [row_lo, row_hi) of `stage` (1 or 2) in several gnnvc_stage_forward_device
 * calls announces the stage's complete input once: "d_in is final and will not change until those calls are done".
 * The engine then builds the compact-table plan over that row range — once per graph and range —
 * and writes the table for this input, so that each of the calls (if it covers enough rows to fill the GPU) reads
 * its neighbours from the table instead of gathering 64-byte rows.  Purely an optimisation hint: results are
 * bit-identical with or without it, and it does nothing on graphs or inputs the plan does not fit.  The
 * announcement is forgotten at the next gnnvc_stage_input_ready / gnnvc_forward(_device) / graph change. */
int gnnvc_stage_input_ready(gnnvc_engine *e, int stage, const float *d_in, uint32_t row_lo, uint32_t row_hi);

/* ---- reduction-rule candidates ------------------------------------
 * One vertex-parallel pass over the uploaded graph that evaluates which local rules of the
 * reference's reduce_graph (include/mwvc_reductions.hpp:335-380) would fire on each vertex as
 * the graph stands: flags[u] bit r for rule r of its switch — 0 neighborhood_reduction,
 * 1 twin_fold, 2 domination_reduction, 3 isolated_fold, 4 independent_fold, 5 neighbor_meta_reduction,
 * 6 neighborhood_meta_reduction — exact predicates, the last two with the covers of their <= 8-vertex
 * subgraphs enumerated the way include/small_solve.hpp does.  Vertices with D(u) > max_degree get 0
 * (reduce_graph skips them, :344; pass 20).
 * Needs ascending neighbour lists.  The host keeps applying reductions in the reference's own
 * stack order and merely skips clean vertices whose bit is 0. */
int gnnvc_reduction_flags(gnnvc_engine *e, uint32_t max_degree, uint8_t *flags);

/* ---- score consumer keys ------------------------------------------------
 * What the driver's sort and selection loop read from the scores (reference src/GNN_VC.cpp:194-206,
 * 213, 220): keys[u] = std::min(s, 1.0f - s), above_half[u] = s > 0.5f, computed on the device from
 * d_scores (n floats; NULL = the scores the last gnnvc_forward left there) and copied to the host.
 * The comparator itself (its 1e-4 tie band is not a strict weak order) must stay the host's
 * std::sort for the order to stay the reference's. */
int gnnvc_score_keys(gnnvc_engine *e, const float *d_scores, uint32_t n, float *keys, uint8_t *above_half);

/* Wait for everything queued on the engine's stream. */
int gnnvc_synchronize(gnnvc_engine *e);

/* hipEvent timings of the last gnnvc_forward / gnnvc_forward_device on the
 * engine's stream: total and per stage, in milliseconds (waits for them).  A forward records
 * events only when option "forward_timing" asks for them: 0 (the default) -> GNNVC_ERR_STATE,
 * 1 -> the total (stage_ms[] = -1), 2 -> total and stages. */
int gnnvc_last_forward_ms(gnnvc_engine *e, float *total_ms, float *stage_ms, int max_stages);

/* Per-kernel HIP-event timings of the gnnvc_forward_device calls made since the last call of this function, when
 * option "kernel_trace" is 1 (bench.py's roofline block): every kernel those forwards launched on the engine's
 * stream, in launch order — names[i] (static strings: the kernel as written at its launch site) and ms[i];
 * *count = how many there were (may exceed max; at most 16384 are kept).  Reading clears the records.
 * Kernels on the engine's side streams (long / giant rows, overlapped dense rounds) are not listed.  Waits. */
int gnnvc_kernel_trace(gnnvc_engine *e, int max, const char **names, float *ms, int *count);

/* ---- layer-level entry points (host pointers) ------------------------------
 * One call per reference layer forward(); used by the C++ mirror of the
 * reference's layer structs and by the per-layer parity tests. */

/* graph_layer::forward (reference src/gnn_inference.cpp:27-42) on the graph
 * currently uploaded: in is n x f, out is n x (2f+3), ws as set. */
int gnnvc_graph_layer_forward(gnnvc_engine *e, uint32_t f, const float *in, float *out);

/* linear_layer::forward (reference src/gnn_inference.cpp:20-25): out = in*W + bias,
 * in n x k, W k x m row-major, bias m. */
int gnnvc_linear_forward(gnnvc_engine *e, uint32_t n, uint32_t k, uint32_t m,
                         const float *in, const float *W, const float *bias, float *out);

/* ReLU::forward / sigmoid::forward (reference src/gnn_inference.cpp:44-52). */
int gnnvc_relu_forward(gnnvc_engine *e, size_t count, const float *in, float *out);
int gnnvc_sigmoid_forward(gnnvc_engine *e, size_t count, const float *in, float *out);

/* The neighbour sum of graph_layer::forward (reference src/gnn_inference.cpp:33-36) on explicit data:
 * sums[i] = (((0 + v[i][0]) + v[i][1]) + ...) + v[i][len-1], one rounded fp32 add per element, for `streams`
 * rows of `len` floats (host pointers, row-major).  Both modes evaluate that chain with the engine's parallel
 * giant-row kernels and return its exact bits whatever the data holds — the layer-level handle on the path rows of degree
 * >= "giant_row_threshold" take.  mode 0 runs a stream on several waves (segments of 4096 addends, each with its own parity
 * map relative to an estimated binade, used by the final walk only where the exact accumulator confirms it); mode 2 is the
 * same result with one wave walking the whole stream ("giant_segments" 0 for the engine's own rows).  (Rounds 1 - 3 had a
 * tolerance mode 1 — tree sums; it is gone: GNNVC_ERR_INVALID.) */
int gnnvc_stream_sum(gnnvc_engine *e, const float *values, uint32_t streams, uint32_t len, int mode, float *sums);

/* dot() (reference src/matrix.cpp:106-122, the cblas_sgemm seam):
 * C = op(A) * op(B) + beta * C, row-major, op = transpose when the flag is set;
 * m, n, k are the dimensions AFTER op.  Each output is one sequential-k fmaf
 * chain from +0.0f; with beta != 0 the old C is then added as fma(beta, C, acc). */
int gnnvc_sgemm(gnnvc_engine *e, int trans_a, int trans_b, uint32_t m, uint32_t n, uint32_t k,
                const float *A, uint32_t lda, const float *B, uint32_t ldb, float beta,
                float *C, uint32_t ldc);

#ifdef __cplusplus
}
#endif
#endif /* GNNVC_H */
