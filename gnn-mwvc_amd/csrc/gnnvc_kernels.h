// gnnvc_kernels.h — launch interface between the engine (host logic) and the
// gfx950 kernels in gnnvc_kernels.hip.  Internal; the public boundary is
// include/gnnvc.h.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <hip/hip_vector_types.h>

#include <vector>

namespace gnnvc {

// Device view of what the forward reads from reduction_graph
// (reference include/reduction_graph.hpp:141-158,693-704).
struct GraphDev {
    uint32_t n = 0;
    uint64_t nnz = 0;
    const uint32_t *rowptr = nullptr;  // n + 1
    const uint32_t *col = nullptr;     // nnz + GNNVC_COL_PAD (pad content irrelevant)
    const uint32_t *w = nullptr;       // n
    const uint32_t *nw = nullptr;      // n
    // A SLICE of a graph (one rank of a vertex-partitioned run, gnnvc_attach_graph_slice) holds rows
    // [row_base, row_end) only: rowptr / w / nw are then the slice's arrays biased by -row_base so that they are
    // still indexed by global row id (valid for those rows only, row pointers relative to the slice's first
    // entry), col holds the slice's nnz entries with GLOBAL column ids.  row_end == 0: the whole graph.
    uint32_t row_base = 0, row_end = 0;
    // Pruned view of the adjacency for ONE call of a 16-wide stage (set by the engine for that call, see "pruned
    // adjacency" in gnnvc_kernels.hip): the entries whose target row is known to be all zero in this call's input are
    // left out of prp / pcol (n + 1 offsets, the kept entries in CSR order + GNNVC_COL_PAD).  The gathering kernels
    // use it iff *prune_bad == 0 (decided on the device for this very input); rowptr stays the source of degrees.
    const uint32_t *prp = nullptr, *pcol = nullptr, *prune_bad = nullptr;
    // prune_eff != 0: with the pruned adjacency in force a row goes to the tile kernel or to the long-row kernel by the
    // number of entries it has LEFT (not by its degree) — except the giant rows, degree >= eff_giant, which stay with
    // their kernels.  The engine sets it when the tile order of the call was built from those numbers too.
    uint32_t prune_eff = 0, eff_giant = 0xFFFFFFFFu, eff_thresh = 0xFFFFFFFFu;   // eff_thresh: entries left from which a row is the long-row kernel's
    // Filtered gather for ONE call of a 16-wide stage that has no pruned adjacency (yet): zero_bits = bit v set iff row v of this
    // call's input is all zero (n + 1 bits, the pad row's is 0), zero_info = {degrees of those vertices, their number} — both
    // written on the device from this very input (filter_mark) just before the stage's kernels, which redirect an entry whose
    // target's bit is set to the pad row when the set is worth it (filter_worth below).  Null: every entry is gathered.
    const uint32_t *zero_bits = nullptr;
    const unsigned long long *zero_info = nullptr;
    uint32_t zero_min_pct = 50;   // the set is worth a look-up per entry from this share of the entries pointing into it
    // The LONG rows' lists under the filter (k_long_lists in front of k_long_f16): keep_col / keep_cnt = per long row of this call
    // its targets outside the set, in stored order, IN PLACE — row u's at keep_col[rowptr[u] ...], keep_cnt[u] of them — written
    // and walked within the call.  short_col / short_cnt = the lists an earlier stage of the same forward left, which the pass
    // shortens further instead of the full rows iff *short_bad == 0: the set they were made with is all zero in this call's
    // input too (k_filter_mark's verdict, on the device).  cnt pointers are biased like rowptr.
    uint32_t *keep_col = nullptr, *keep_cnt = nullptr;
    const uint32_t *short_col = nullptr, *short_cnt = nullptr, *short_bad = nullptr;
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t lo() const { return row_base; }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t hi() const { return row_end ? row_end : n; }
    bool sliced() const { return row_end != 0 && (row_base != 0 || row_end != n); }
#if defined(__HIPCC__)
    __host__ __device__ bool sliced_dev() const { return row_end != 0 && (row_base != 0 || row_end != n); }
#endif
};

// One fused stage = graph layer (input width F) followed by up to three dense
// layers.  Parameter block layout in device memory (floats):
//   W1[K1 x N1] b1[N1] W2[N1 x N2] b2[N2] W3[N2 x N3] b3[N3],  K1 = 2F + 3.
struct StagePlan {
    int f = 0;                // graph-layer input width: 1 or 16
    int n1 = 0, n2 = 0, n3 = 0;
    int sigmoid_last = 0;     // last activation is a sigmoid (else ReLU)
    size_t param_offset = 0;  // float offset of W1 in the engine's parameter buffer
    int variant = -1;         // index into the compiled instantiations, -1 = none
};

// Degree-sorted tile order of the non-long rows of one row range (device arrays).
struct SortedOrder {
    uint32_t n = 0;                   // entries
    const uint32_t *vertex = nullptr; // vertex id per entry, degrees descending
    const uint4 *meta = nullptr;      // {row begin, row end, W, NW} per entry
};

// Per-kernel timing: while a sink is installed on the calling thread, every kernel the launchers below put on
// sink->stream is bracketed by two HIP events (recs[0 .. used): name = the kernel as written at the launch site).
struct KernelTraceSink {
    struct Rec {
        const char *name;
        hipEvent_t a, b;
    };
    hipStream_t stream = nullptr;
    std::vector<Rec> recs;
    size_t used = 0;
};
void set_kernel_trace(KernelTraceSink *sink);   // nullptr = off

// Returns the instantiation index for a stage shape, or -1.
int stage_variant(int f, int n1, int n2, int n3, int sigmoid_last);

// Fused stage over rows [row_lo, row_hi).  in: F=1 -> n floats; F=16 -> (n+1) x 16
// with a zero last row.  out: (n+1) x n3 rows (n3 = 16) or, for the sigmoid
// stage, out = scores[n], logits optional.
// Producer side of the compact-table plan: a stage kernel whose dense layers run on the VALU can count the
// non-zeros of the rows it writes (64 slots x 17 counters, zeroed by the caller) and write their compact form for
// the columns chosen at the previous forward (spec = that forward's desc), see c4_emit.  All null = off.
struct EmitArgs {
    const uint32_t *spec = nullptr;
    float *table = nullptr;
    unsigned long long *counts = nullptr;
};
constexpr int kEmitCounters = 64 * 17;

hipError_t launch_stage(const StagePlan &sp, const GraphDev &g, float ws, const float *params,
                        const float *in, float *out, float *logits, uint32_t row_lo,
                        uint32_t row_hi, uint32_t long_thresh, bool mfma, const SortedOrder *so,
                        bool interleave, hipStream_t stream, const float *acc4 = nullptr,
                        const uint32_t *c4desc = nullptr, const float *agg16 = nullptr, bool mfma_agg = false,
                        const EmitArgs &emit = EmitArgs(),
                        bool dense_part = true /* false: only the gathering kernel (which leaves at once when the compact-table
                                                  plan applies); the caller launches the sums and the dense kernel itself */,
                        const SortedOrder *so_pruned = nullptr /* g.prune_eff: the tile order by entries left (meta = pruned ranges) */,
                        const float *table_in = nullptr /* acc4: the compact table of THIS stage's input (one pass), so that the dense
                                                           kernel takes a row's own live values from it; null = from the full rows */,
                        const uint32_t *skip_flag = nullptr /* (no acc4) a device word: != 0 = another kernel has done this launch's rows */);
// What a hand-off wants to know about a graph before it accepts and classes it, in ONE pass of the stream and one wait (round 4;
// it used to be four round trips — the checks, nine row pointers, the tiles' lockstep cost, the long rows — 0.2 of a mid-size
// graph's 0.3 ms attach): k_validate_graph, then (all three read row pointers only, so an invalid graph cannot send them astray)
// the row pointers at the eighths of the row range, k_tile_waste and k_find_long, and one kernel that stores the results into
// page-locked memory.  dev_words: 16 words of device scratch.  out_dev (as the device sees it), 18 words: [0] validation flags,
// [1..9] rowptr at the nine cuts, [10..13] k_tile_waste's two 64-bit sums, [14..17] k_find_long's count words.
struct GraphClassArgs {
    bool cuts = false, waste = false, longs = false;
    uint32_t waste_thresh = 0xFFFFFFFFu, heavy_from = 1, long_thresh = 0xFFFFFFFFu;
    uint32_t *long_list = nullptr;   // room for every row the engine holds
};
hipError_t classify_graph(const GraphDev &g, const GraphClassArgs &a, uint32_t *dev_words, uint32_t *out_dev, hipStream_t stream);

// A forward's verdict words (did the LDS-table / compact-table / table-tile plans fit this input: device words the plans' kernels
// leave behind), stored into page-locked host memory by ONE small kernel: out[i] = src[i] ? *src[i] : 0 (out as the device sees it)
struct VerdictWords {
    const uint32_t *src[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
hipError_t write_verdicts(const VerdictWords &vw, uint32_t *out_dev, hipStream_t stream);

// a whole stage (variants 0, 1, 2) on WIDE tiles — a workgroup per 64-vertex tile, the tile's gather and each dense layer's outputs
// split over its four waves (k_stage_w1 / k_stage_w16): graphs with fewer tiles than the chip has SIMDs, no long rows
hipError_t launch_stage_wide(const StagePlan &sp, const GraphDev &g, float ws, const float *params, const float *in, float *out,
                             float *logits, uint32_t row_lo, uint32_t row_hi, hipStream_t stream);
// a 16-wide stage (variant 1 or 2) from the L2-resident compact table of its input (k_stage_t4: graphs of 50 - 400 K vertices, whole
// forwards): table_in = (n + 1) rows of 16 bytes written by the producer of the input (EmitArgs) for the columns in desc_in,
// counts_in = that producer's kEmitCounters; desc_out (another 16 words) receives this forward's choice — the next forward's
// spec — and desc_out[8] = 1 iff the kernel did the rows (the gathering kernel launched behind it takes desc_out + 8 as skip_flag)
hipError_t launch_stage_t4(const StagePlan &sp, const GraphDev &g, float ws, const float *params, const float *in, float *out, float *logits,
                           uint32_t row_lo, uint32_t row_hi, bool interleave, hipStream_t stream, const float *table_in,
                           const unsigned long long *counts_in, unsigned long long *counts_zero /* the other parity's set, cleared here */,
                           const uint32_t *desc_in, uint32_t *desc_out, const EmitArgs &emit,
                           bool solo = false /* no gathering kernel follows: a table that does not fit is handled inside, the slow way */);
// the dense layers + sigmoid of the last stage when its aggregates are ready (compact-table plan)
hipError_t launch_dense_sigmoid(const StagePlan &sp, const GraphDev &g, float ws, const float *params, const float *in, float *out,
                                float *logits, uint32_t row_lo, uint32_t row_hi, const float *acc4, const uint32_t *c4desc,
                                const float *agg16, hipStream_t stream, uint32_t long_thresh = 0xFFFFFFFFu /* rows at least this long are not this kernel's */,
                                const float *table_in = nullptr /* as launch_stage's */);

// Building blocks of the degree-sorted order (the prefix over the few thousand degree classes
// is done on the host).
hipError_t measure_tile_waste(const GraphDev &g, uint32_t row_lo, uint32_t row_hi, uint32_t long_thresh,
                              unsigned long long *sum_max /* [2] */, hipStream_t stream, uint32_t heavy_from = 0xFFFFFFFFu);
// (degree_*: g.rowptr decides a row's class; skip_rowptr != nullptr: rows with skip_rowptr[u + 1] - skip_rowptr[u] >= skip_from
// are left out as well — the giant rows when g.rowptr is a pruned adjacency's)
hipError_t degree_histogram(const GraphDev &g, uint32_t row_lo, uint32_t row_hi, uint32_t long_thresh,
                            uint32_t bins, uint32_t *hist, hipStream_t stream, const uint32_t *skip_rowptr = nullptr,
                            uint32_t skip_from = 0xFFFFFFFFu);
// hist (degree_histogram) -> in place the first slot of every class, heaviest class first; info[0] = rows listed, info[1] = rows
// without entries (device memory, 2 words)
hipError_t degree_starts(uint32_t *hist, uint32_t bins, uint32_t *info, hipStream_t stream);
hipError_t degree_scatter(const GraphDev &g, uint32_t row_lo, uint32_t row_hi, uint32_t long_thresh,
                          uint32_t bins, uint32_t *cursor, uint32_t *vertex, void *meta, hipStream_t stream,
                          const uint32_t *skip_rowptr = nullptr, uint32_t skip_from = 0xFFFFFFFFu);

// Long rows (degree >= thresh): listed once per graph, then one workgroup per row per
// stage (same CSR-order sums).  The tile kernels above skip those rows when given the
// same threshold; pass 0xFFFFFFFF to make them handle every row.
hipError_t find_long_rows(const GraphDev &g, uint32_t thresh, uint32_t *list, uint32_t *count,
                          hipStream_t stream);
hipError_t launch_long_stage(const StagePlan &sp, const GraphDev &g, float ws, const float *params,
                             const float *in, float *out, float *logits, uint32_t row_lo, uint32_t row_hi,
                             const uint32_t *list, uint32_t n_long, uint32_t min_deg, uint32_t max_deg, hipStream_t stream);

// Giant rows (degree >= the giant threshold, a subset of the long rows): the same CSR-order fp32 sums evaluated in
// parallel (exact_sum.h).  meta: uint4[n + 1] {row, first CSR entry, degree, first gather block}, the last entry's
// .w = blocks; off[i]: float offset in `slab` of row i's first stream (16 streams of the degree rounded up to
// giant_window()); agg: float[n x 16].  find_giant_rows fills {row, first entry, degree, 0} in no particular order.
struct GiantRows {
    uint32_t n = 0, blocks = 0;
    const void *meta = nullptr;
    const unsigned long long *off = nullptr;
    float *slab = nullptr, *agg = nullptr;
    // one stream on several waves (k_giant_segsum / k_giant_segmap): n x F x maxseg floats / uint4; null or maxseg <= 1: one wave per stream
    float *segsum = nullptr;
    void *segmap = nullptr;
    uint32_t maxseg = 0;
};
uint32_t giant_segments(uint32_t len);
uint32_t giant_window();
uint32_t giant_block();
hipError_t find_giant_rows(const GraphDev &g, const uint32_t *list, uint32_t n_long, uint32_t thresh, void *meta, uint32_t *count,
                           hipStream_t stream);
hipError_t launch_giant_stage(const StagePlan &sp, const GraphDev &g, float ws, const float *params, const float *in, float *out,
                              float *logits, uint32_t row_lo, uint32_t row_hi, const GiantRows &gr, hipStream_t stream,
                              uint32_t min_deg = 0 /* listed rows below this degree are another kernel's in this stage */, int part = 0);
hipError_t stream_sums(const float *streams_dev, uint32_t streams, uint32_t len, void *meta, unsigned long long *off, float *agg,
                       int mode /* 0 on several waves, 2 on one wave: the same exact sum */, hipStream_t stream, float *segsum = nullptr,
                       void *segmap = nullptr);

// Column-blocked F = 1 stage (bit-identical to launch_stage on stage 0; see the kernels).
// bp: uint32[nblocks * n + 1] block-major entry pointers, colb: uint32[nnz + pad] re-bucketed
// columns, acc: float[n] running sums.  *bad_flag != 0 after the build means the graph's
// rows are not block-monotone and the plan must not be used.
size_t blocked_scan_scratch_elems(size_t n_elems);
hipError_t build_blocked_index(const GraphDev &g, uint32_t wb, uint32_t nblocks, uint32_t long_thresh,
                               uint32_t *bp, uint32_t *colb, uint32_t *scratch, uint32_t *bad_flag,
                               hipStream_t stream);
hipError_t launch_stage0_blocked(const StagePlan &sp, const GraphDev &g, float ws, const float *params,
                                 const float *x, float *out, uint32_t row_lo, uint32_t row_hi,
                                 uint32_t nblocks, const uint32_t *bp, const uint32_t *colb, float *acc,
                                 uint32_t long_thresh, bool mfma, bool interleave, hipStream_t stream,
                                 const EmitArgs &emit = EmitArgs());

// Layer-by-layer kernels (any model; also the layer-level ABI entry points).
hipError_t launch_graph_layer(const GraphDev &g, float ws, uint32_t f, const float *in,
                              float *out, hipStream_t stream);
hipError_t launch_linear(uint32_t n, uint32_t k, uint32_t m, const float *in, const float *W,
                         const float *bias, float *out, hipStream_t stream);
hipError_t launch_relu(size_t count, const float *in, float *out, hipStream_t stream);
hipError_t launch_sigmoid(size_t count, const float *in, float *out, hipStream_t stream);
hipError_t launch_sgemm(int ta, int tb, uint32_t m, uint32_t n, uint32_t k, const float *A,
                        uint32_t lda, const float *B, uint32_t ldb, float beta, float *C,
                        uint32_t ldc, hipStream_t stream);

// LDS-table plan of the F = 1 stage (see the kernels): build steps, then the per-forward launch
// (bits = the width of a table entry: 8, 10 or 16 — lds_table_bits_for(largest k); 0 = too wide for the plan)
uint32_t lds_table_max_rows(uint32_t bits = 8);
uint32_t lds_table_bits_for(uint32_t kmax);
size_t lds_table_bytes_for(uint32_t bits, size_t n);   // bytes of a table over n vertices (padding included)
hipError_t lds_table_wmax(const uint32_t *w, uint32_t n, uint32_t *wmax, hipStream_t stream);   // *wmax = max of w[0 .. n)
// A plan over rows that are not consecutive (skewed graphs, compact-table plan): chunk c holds the rows
// rowmap[c * rows_per_chunk ..] (0xFFFFFFFF = empty slot), its regrouped entries start at first[c] (+ the padding
// slack), and column blocks may have any widths: block b = columns [bstart[b], bstart[b + 1]).  All null (the
// default): consecutive rows, CSR offsets, blocks of block_cols columns.
struct PlanMap {
    const uint32_t *rowmap = nullptr, *first = nullptr, *bstart = nullptr;
    const uint16_t *coarse = nullptr;   // with bstart: the block of column 256 k, for every k (a column's block is that one or, rarely, a later one)
};
uint32_t lds_table_block(uint32_t bits = 8);
uint32_t lds_table_step();
uint32_t lds_table_record_words();
// workgroup steps of the F = 1 plan: one record per (chunk, block, up to 256 entries per slice); a chunk = 16 slices
hipError_t lds_table_wsteps(const GraphDev &g, uint32_t slice_rows, uint32_t nchunks, uint32_t nblocks, const uint32_t *seg_cnt,
                            const uint32_t *step_ptr, uint32_t *step_count, uint32_t *recs, bool write, uint32_t slack,
                            hipStream_t stream, const PlanMap &pm = PlanMap(), uint32_t row_base = 0, uint32_t row_end = 0xFFFFFFFFu,
                            uint32_t bits = 8);
// (chunk0 / chunk1: only chunks [chunk0, chunk1) of the plan — the flat builders, lds_table_is_flat(), work piece by piece, so a
// hand-off can regroup the rows whose column entries have arrived while the rest is still on its way; the steps come last)
bool lds_table_is_flat(uint32_t rows_per_chunk, const PlanMap &pm = PlanMap());
hipError_t lds_table_count(const GraphDev &g, uint32_t rows_per_chunk, uint32_t nchunks, uint32_t nblocks, uint32_t block_cols,
                           uint32_t *seg_cnt, uint32_t *bad, hipStream_t stream, uint32_t row_base = 0,
                           uint32_t row_end = 0xFFFFFFFFu, const PlanMap &pm = PlanMap(), uint32_t chunk0 = 0,
                           uint32_t chunk1 = 0xFFFFFFFFu);
hipError_t lds_table_steps(const GraphDev &g, uint32_t rows_per_chunk, uint32_t nchunks, uint32_t nblocks, const uint32_t *seg_cnt,
                           const uint32_t *step_ptr, uint32_t *step_count, void *steps, bool write, hipStream_t stream,
                           uint32_t row_base = 0, uint32_t row_end = 0xFFFFFFFFu, uint32_t cap = 0 /* entries per step, 0 = 2048 */,
                           uint32_t slack = 0 /* != 0: segments start at multiples of 4, chunk c shifted by c * slack */,
                           uint32_t block_cols = 0 /* the step descriptors' fourth word: the block's first column */,
                           const PlanMap &pm = PlanMap());
hipError_t lds_table_scatter(const GraphDev &g, uint32_t rows_per_chunk, uint32_t nchunks, uint32_t nblocks, uint32_t block_cols,
                             const uint32_t *seg_cnt, uint32_t *entries, hipStream_t stream, uint32_t shift,
                             uint32_t row_base, uint32_t row_end, uint32_t slack, const PlanMap &pm,
                             uint32_t *bad /* |= 2 if a row's blocks do not ascend (the flat walk checks here, not in the count pass) */,
                             uint32_t chunk0 = 0, uint32_t chunk1 = 0xFFFFFFFFu);
// Pruned adjacency (k_prune_*).  The set of vertices whose rows are taken to be all zero, as a bitmap: mark_zero (the rows that
// ARE all zero in feat) or predict_zero_rows (the rows the graph's weights predict, at hand-off).  count = per chunk of 64 entries which are kept (mask) and, scanned in place, how many before
// it (off: chunks + 1 words, off[chunks] = kept entries; scratch as for blocked_scan_scratch_elems(chunks + 1)); fill = the
// kept entries (pcol, sized by the caller from off[chunks]) and the pruned row offsets (prp: rows + 1 words); check =
// *bad |= 1 if a vertex of the set has a non-zero row in feat.
hipError_t prune_mark_zero(const GraphDev &g, const float *feat, uint32_t *heavy_bits, hipStream_t stream);
// the set PREDICTED from the graph alone (hand-off): the vertices whose rows the F = 1 stage sp0 will write as all zero when the
// caller's input is the reference driver's x = W / ws (see k_predict_zero_f1); whole graphs only
hipError_t predict_zero_rows(const StagePlan &sp0, const GraphDev &g, float ws, const float *params, uint32_t *heavy_bits, hipStream_t stream);
// filtered gather (GraphDev::zero_bits / zero_info): bits = n / 32 + 1 words, info = 3 words {degrees, members, verdict}, written
// from feat (16 columns).  prev_bits / prev_info: an earlier stage's set of this forward — info[2] = 0 iff the lists that stage
// left (GraphDev::keep_col) may stand for the adjacency with THIS input (GraphDev::short_bad points to info + 2).
hipError_t filter_mark(const GraphDev &g, const float *feat, uint32_t *bits, unsigned long long *info, hipStream_t stream,
                       const uint32_t *prev_bits = nullptr, const unsigned long long *prev_info = nullptr);
// mass[0] = sum of the degrees of the set's vertices (whole graphs: the entries that point to them, if the adjacency is symmetric;
// 0 on a slice), mass[1] = vertices in the set
// (prev_bits: another set — mass[2] = its vertices outside this one; mass has room for 3 words)
hipError_t prune_mass(const GraphDev &g, const uint32_t *heavy_bits, unsigned long long *mass, hipStream_t stream,
                      const uint32_t *prev_bits = nullptr);
hipError_t prune_count(const GraphDev &g, const uint32_t *heavy_bits, unsigned long long *mask, uint32_t *off, uint32_t *scratch,
                       hipStream_t stream);
hipError_t prune_fill(const GraphDev &g, const unsigned long long *mask, const uint32_t *off, uint32_t *pcol, uint32_t *prp,
                      hipStream_t stream);
hipError_t prune_check(const GraphDev &g, const float *feat, const uint32_t *heavy_bits, uint32_t *bad, hipStream_t stream);
// rows of a degree-sorted list (heaviest first; the first m of it) dealt serpentine to nslices slices of slice_rows slots:
// rowmap[s * slice_rows + t], weight[s] = entries of slice s
hipError_t deal_rows(const GraphDev &g, const uint32_t *sorted_rows, uint32_t m, uint32_t slice_rows, uint32_t nslices,
                     uint32_t *rowmap, uint32_t *weight, hipStream_t stream);
// cand[k] = first row whose CSR offset reaches k * target: column ranges of equal entry mass on a symmetric adjacency
hipError_t mass_bounds(const GraphDev &g, unsigned long long target, uint32_t count, uint32_t *cand, hipStream_t stream);
hipError_t launch_stage0_lds_table(const StagePlan &sp, const GraphDev &g, float ws, const float *params, const float *x,
                                   float *out, uint32_t row_lo, uint32_t row_hi, uint32_t rows_per_chunk,
                                   const uint32_t *step_ptr, const void *steps, const uint32_t *entries,
                                   uint8_t *wbyte /* lds_table_bytes_for(bits, n) bytes: rewritten from x by every launch */,
                                   float *acc, uint32_t *bad, uint32_t long_thresh, bool mfma, bool interleave,
                                   hipStream_t stream, const EmitArgs &emit, uint32_t last_entry,
                                   const uint32_t *rowmap = nullptr /* skewed graphs: the plan's rows, slice by slice */,
                                   uint32_t mapped_chunks = 0,
                                   uint32_t plan_base = 0, uint32_t plan_end = 0xFFFFFFFFu /* the plan's row range (a rank's rows) */,
                                   uint32_t bits = 8 /* width of the table's entries (the plan's blocks were laid out for it) */);

// compact-table plan of the 16-wide stages (see the k_c4_* kernels); the step layout is built with the
// lds_table_* functions per SLICE (rows_per_chunk / compact_slices() rows), compact_step() entries per step,
// segments aligned to 4 entries (slack != 0)
uint32_t compact_max_rows();
uint32_t compact_block();
uint32_t compact_shift();
uint32_t compact_slices();   // a chunk = this many row slices (one per wave); the plan is built per slice
uint32_t compact_step();     // entries per step
// counts: per-column non-zero counts of `in` — 16 counters from column_counts() (count_slots = 1) or the
// producer's kEmitCounters (count_slots = 64: 17 counters per slot, the 17th = rows seen; the table may then
// already hold the rows' compact form for the columns in desc, see EmitArgs)
struct CompactPlan {   // the per-graph part of the plan, as the launches need it
    uint32_t rows_per_chunk = 0, block_cols = 0, nblocks = 0, plan_base = 0, plan_end = 0;
    uint32_t last_entry = 0;             // last index of entries[] a 16-byte read may start at
    uint32_t nslices = 0;                // mapped plans: slices in rowmap
    uint32_t max_passes = 1;             // tables of four columns the device may choose for one input (<= compact_max_passes())
    const uint32_t *step_ptr = nullptr;
    const void *steps = nullptr;
    const uint32_t *entries = nullptr;
    const uint32_t *rowmap = nullptr;    // null: slices of consecutive rows
};
uint32_t compact_max_passes();
// table: max_passes x (n + 1) rows of 16 bytes, acc4: max_passes x n
hipError_t launch_compact_gather(const GraphDev &g, const CompactPlan &cp, const float *in, const unsigned long long *counts,
                                 int count_slots, uint32_t *desc, float *table, float *acc4, uint32_t row_lo, uint32_t row_hi,
                                 uint32_t *dirty_rows, uint32_t dirty_cap, float *agg16, hipStream_t stream,
                                 int what = 3 /* 1 = prepare, 2 = sums, 3 = both */);

hipError_t compact_choose(const unsigned long long *counts, int count_slots, uint32_t rows, uint32_t *desc, uint32_t max_passes,
                          hipStream_t stream);
// the parts of launch_compact_gather's second half, for callers that run them round by round
hipError_t compact_sums(const GraphDev &g, const CompactPlan &cp, uint32_t *desc, const float *table, float *acc4, uint32_t row_lo,
                        uint32_t row_hi, uint32_t *dirty_rows, uint32_t dirty_cap, hipStream_t stream, bool one_round = false);
hipError_t compact_mark(const uint32_t *desc, uint32_t *marks, uint32_t k, hipStream_t stream);
hipError_t compact_fix(const GraphDev &g, const float *in, const uint32_t *desc, const uint32_t *dirty_rows, uint32_t dirty_cap,
                       float *agg16, const uint32_t *marks, hipStream_t stream, uint32_t blocks = 1024);

// The next graph derived from the resident one (see the k_derive_* kernels): new_of is scratch of
// old_g.n words, tail receives per new row the number of entries the device cannot derive, *bad != 0 afterwards means
// the mapping is inconsistent (bit0 old row out of range, bit1 claimed twice, bit2 more survivors than the new degree).
hipError_t derive_tails(const GraphDev &old_g, const uint32_t *old_row, uint32_t n_new, const uint32_t *rowptr_new, uint32_t *new_of,
                        uint32_t *tail, uint32_t *bad, hipStream_t stream);
hipError_t derive_fill(const GraphDev &old_g, const uint32_t *old_row, const uint32_t *new_of, const uint32_t *rowptr_new,
                       const uint32_t *tail_ptr, const uint32_t *tail_cols, uint32_t n_new, uint32_t *col_new, hipStream_t stream);
hipError_t row_hashes(const GraphDev &g, unsigned long long *out, hipStream_t stream);

hipError_t score_keys(const float *scores, size_t n, float *keys, uint8_t *above_half, hipStream_t stream);

// feature-row codec of the inter-GPU exchange (16-column rows; a piece = dense rows + exception list)
hipError_t live_columns(const float *feat, size_t rows, uint32_t *mask, hipStream_t stream);
hipError_t column_counts(const float *feat, size_t rows, unsigned long long *counts, hipStream_t stream);
hipError_t pack_rows(const float *feat, uint32_t row_lo, uint32_t row_hi, uint32_t mask, uint32_t kp, float *dense,
                     uint32_t *exc, uint32_t cap, uint32_t *flag, hipStream_t stream);
hipError_t unpack_rows(const float *dense, const uint32_t *exc, uint32_t cap, uint32_t row_lo, uint32_t row_hi,
                       uint32_t mask, uint32_t kp, float *feat, hipStream_t stream);

hipError_t unpack_gathered(const float *buf, uint32_t world, uint32_t skip, size_t piece_words, uint32_t dense_rows,
                           uint32_t cap, uint32_t per, uint32_t off, uint32_t size, uint32_t n, uint32_t mask, uint32_t kp,
                           float *feat, hipStream_t stream);

// several devices behind one handle: one packed piece (rows x kp dense floats + exception list of room `cap`) stored into n_dst
// (<= 64) destination regions in one launch; the regions of one piece index from up to 64 peers expanded in one launch
struct UnpackPiece {
    const float *region;
    uint32_t row_lo, row_hi;
};
hipError_t push_piece(const float *region, uint32_t rows, uint32_t kp, uint32_t cap, uint32_t n_dst, float *const *dst, hipStream_t stream);
hipError_t unpack_pieces(const UnpackPiece *pieces, uint32_t n_pieces, uint32_t cap, uint32_t mask, uint32_t kp, float *feat, hipStream_t stream);

// Reduction-rule predicates per vertex (one byte each) on the device CSR; see the kernel.
hipError_t launch_reduction_flags(const GraphDev &g, uint32_t max_degree, uint8_t *flags, hipStream_t stream);

// Device-side graph checks after an upload (*flags: bit0 column id out of range, bit1 bad
// row pointers) and the uint64 -> uint32 row-pointer narrowing of the host ABI.
hipError_t validate_graph(const GraphDev &g, uint32_t *flags, hipStream_t stream);
hipError_t validate_rowptr(const GraphDev &g, uint32_t *flags, hipStream_t stream);   // row pointers only (bit1 as validate_graph)
hipError_t narrow_rowptr(const void *in_u64, uint32_t *out, size_t count, hipStream_t stream);

// this zero-fills
// the pad row of a feature matrix: rows [n, n+1) of an (n+1) x width buffer.
// *yes = kernels on b run beside kernels on a (the two streams sit on different hardware queues); ~0.3 ms
hipError_t streams_run_side_by_side(hipStream_t a, hipStream_t b, bool *yes);
hipError_t launch_zero_pad_row(float *buf, uint32_t n, uint32_t width, hipStream_t stream);

}  // namespace gnnvc
