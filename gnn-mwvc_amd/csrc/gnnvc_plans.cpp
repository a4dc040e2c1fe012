// gnnvc_plans.cpp — what the engine builds per GRAPH, and when (libgnnvc_hip.so; see gnnvc_engine_state.h).
//
// The forward reads reduction_graph's rows in stored order (reference include/reduction_graph.hpp:693-704,
// src/gnn_inference.cpp:33-36); everything here is a re-arrangement of those rows that leaves every row's order of addition
// alone: row classes (long / giant rows), the degree-sorted tile order, the LDS-table plan of the F = 1 stage, the
// compact-table plan of the 16-wide stages, the column-blocked index, the pruned adjacency — and the hand-off logic that
// builds what depends on the graph alone while the graph is still arriving (round 3).  No arithmetic of the forward.
#include "gnnvc_engine_state.h"

namespace gnnvc_eng {

// Giant rows of the current graph: those of the long-row list whose degree reaches the giant threshold (in fast
// hub mode: every long row).  Their neighbour values go through a column-major slab, one stream per (row, feature
// column), laid out here on the host — heaviest row first, so the longest streams start first.
int find_giant(gnnvc_engine *e) {
    uint32_t gt = e->opt_giant_thresh ? std::max(e->opt_giant_thresh, e->long_thresh) : 0xFFFFFFFFu;
    if (gt == 0xFFFFFFFFu || e->n_long == 0) return GNNVC_OK;
    HIP_TRY(e, e->gi_meta.reserve((size_t)e->n_long + 1));
    HIP_TRY(e, gnnvc::find_giant_rows(e->g, e->long_list.p, e->n_long, gt, e->gi_meta.p, e->long_count.p, e->stream));
    uint32_t cnt = 0;
    HIP_TRY(e, hipMemcpyAsync(&cnt, e->long_count.p, sizeof cnt, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (cnt == 0) return GNNVC_OK;
    std::vector<uint4> meta((size_t)cnt + 1);
    HIP_TRY(e, hipMemcpy(meta.data(), e->gi_meta.p, (size_t)cnt * sizeof(uint4), hipMemcpyDeviceToHost));
    std::sort(meta.begin(), meta.begin() + cnt, [](const uint4 &a, const uint4 &b) { return a.z != b.z ? a.z > b.z : a.x < b.x; });
    std::vector<unsigned long long> off(cnt);
    const uint32_t win = gnnvc::giant_window(), blk = gnnvc::giant_block();
    uint64_t floats = 0, blocks = 0, entries = 0;
    for (uint32_t i = 0; i < cnt; ++i) {
        const uint64_t lpad = ((uint64_t)meta[i].z + win - 1) / win * win;
        off[i] = floats;
        meta[i].w = (uint32_t)blocks;
        floats += 16 * lpad;
        blocks += (meta[i].z + blk - 1) / blk;
        entries += meta[i].z;
    }
    if (blocks >= 0x7FFFFFFFull) return GNNVC_OK;   // (cannot happen with 32-bit row pointers and a threshold >= 2)
    meta[cnt] = make_uint4(0xFFFFFFFFu, 0u, 0u, (uint32_t)blocks);
    HIP_TRY(e, e->gi_off.reserve(cnt));
    HIP_TRY(e, e->gi_slab.reserve(floats));
    HIP_TRY(e, e->gi_agg.reserve((size_t)cnt * 16));
    e->gi_maxseg = gnnvc::giant_segments(meta[0].z);   // (the list is sorted: its first row is the longest)
    // One stream on several waves pays when the longest stream's walk (~2 ns per addend) is what a stage waits for — the
    // power-law graph: 0.39 ms against ~0.15 ms of gathering, forward 1.56 -> 1.16 ms.  Where the stage is busy gathering
    // anyway (R-MAT-22: 0.32 ms of walk inside a 1.2 ms stage) the extra kernels of the high-priority stream only take
    // slots from the tile kernel: 2.98 -> 3.16 ms.  Auto: on when the walk exceeds half of nnz / 50 G entries per second.
    e->giant_walk_bound = (double)meta[0].z * 2.0e-9 > 0.5 * (double)e->g.nnz / 50.0e9;
    bool segments = e->opt_giant_segments > 0;
    if (e->opt_giant_segments < 0) segments = e->giant_walk_bound;
    if (segments && e->gi_maxseg > 1 && (uint64_t)cnt * 16 * e->gi_maxseg < (1ull << 31)) {
        HIP_TRY(e, e->gi_segsum.reserve((size_t)cnt * 16 * e->gi_maxseg));
        HIP_TRY(e, e->gi_segmap.reserve((size_t)cnt * 16 * e->gi_maxseg));
    } else {
        e->gi_maxseg = 0;
    }
    HIP_TRY(e, hipMemcpy(e->gi_meta.p, meta.data(), ((size_t)cnt + 1) * sizeof(uint4), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->gi_off.p, off.data(), (size_t)cnt * sizeof(unsigned long long), hipMemcpyHostToDevice));
    {
        int rc = ensure_side_streams(e);
        if (rc) return rc;
    }
    e->n_giant = cnt;
    e->giant_blocks = (uint32_t)blocks;
    e->giant_entries = entries;
    e->giant_thresh = gt;
    return GNNVC_OK;
}

// Rows the tile kernels hand to the long-row kernels (per graph).
// The checks of a candidate graph and what find_long wants to know about it, in one pass of the stream and ONE wait (round 4: a
// mid-size graph's attach spent 0.2 of its 0.3 ms in four round trips).  bad = the validation flags (the caller refuses the graph);
// e->pre = the row pointers at the eighths of the row range, the tiles' lockstep cost and the long rows at the first threshold
// find_long will ask for — consumed by the find_long of the same row range that follows, whatever it decides from them.
int classify_hand_off(gnnvc_engine *e, const GraphDev &cand, uint32_t &bad) {
    e->pre = gnnvc_engine::PreClass();
    HIP_TRY(e, e->cls_dev.reserve(16));
    if (!e->cls_pin_dev) {
        HIP_TRY(e, e->cls_pin.reserve(24));
        HIP_TRY(e, hipHostGetDevicePointer(reinterpret_cast<void **>(&e->cls_pin_dev), e->cls_pin.p, 0));
    }
    gnnvc::GraphClassArgs a;
    const uint32_t glo = cand.lo(), ghi = cand.hi();
    const bool classes = !e->stages.empty() && cand.n != 0 && ghi > glo;   // (find_long's own first test)
    gnnvc_engine::PreClass pre;
    pre.lo = glo;
    pre.hi = ghi;
    if (classes) {
        a.cuts = pre.cuts = cand.nnz && ghi - glo >= 4096;
        if (e->opt_sorted != 0 && cand.nnz) {
            a.waste = pre.waste = true;
            a.waste_thresh = pre.waste_thresh = e->opt_long_thresh ? e->opt_long_thresh : 0xFFFFFFFFu;
            a.heavy_from = pre.heavy_from =
                (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, std::max<uint64_t>(1, 4 * cand.nnz / std::max<uint32_t>(ghi - glo, 1)));
        }
        if (e->opt_long_thresh) {
            HIP_TRY(e, e->long_list.reserve(ghi - glo));
            a.longs = pre.longs = true;
            a.long_thresh = pre.long_thresh = e->opt_long_auto ? 256u : e->opt_long_thresh;
            a.long_list = e->long_list.p;
        }
    }
    for (int i = 0; i < 24; ++i) e->cls_pin.p[i] = 0;
    HIP_TRY(e, gnnvc::classify_graph(cand, a, e->cls_dev.p, e->cls_pin_dev, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    const uint32_t *w = e->cls_pin.p;
    bad = w[0];
    if (bad || !classes) return GNNVC_OK;
    for (int k = 0; k < 9; ++k) pre.cut[k] = w[1 + k];
    pre.sums[0] = (unsigned long long)w[10] | ((unsigned long long)w[11] << 32);
    pre.sums[1] = (unsigned long long)w[12] | ((unsigned long long)w[13] << 32);
    for (int k = 0; k < 4; ++k) pre.found[k] = w[14 + k];
    pre.valid = true;
    e->pre = pre;
    return GNNVC_OK;
}

int find_long(gnnvc_engine *e) {
    gnnvc_engine::PreClass pre = e->pre;   // (what classify_hand_off learned, if it ran for THIS hand-off just now)
    if (!e->pre_armed) pre.valid = false;
    e->pre.valid = false;
    e->pre_armed = false;
    e->n_long = 0;
    e->n_giant = 0;
    e->giant_blocks = 0;
    e->giant_entries = 0;
    e->giant_thresh = 0xFFFFFFFFu;
    e->long_thresh = 0xFFFFFFFFu;
    e->thresh_f16 = 0xFFFFFFFFu;
    for (auto &r : e->srt) r.valid = false;   // new graph: any cached tile order is stale
    e->srt_cur = -1;
    const GraphDev &g = e->g;
    e->sorted_wanted = false;
    e->srt_waste = 0.0;
    e->interleave = false;
    if (e->stages.empty() || g.n == 0 || g.hi() <= g.lo()) return GNNVC_OK;
    const uint32_t glo = g.lo(), ghi = g.hi();   // the rows this engine holds (a slice of a partitioned graph, or all)
    const bool have_pre = pre.valid && pre.lo == glo && pre.hi == ghi;
    if (g.nnz && ghi - glo >= 4096) {
        // The natural tile map hands each XCD a contiguous eighth of the rows.  If the eighths hold
        // very different numbers of entries (R-MAT: low ids are the hubs) deal tiles round-robin.
        uint32_t cut[9];
        if (have_pre && pre.cuts) {
            for (int k = 0; k <= 8; ++k) cut[k] = pre.cut[k];
        } else {
            for (int k = 0; k <= 8; ++k)
                HIP_TRY(e, hipMemcpyAsync(&cut[k], g.rowptr + glo + (size_t)((uint64_t)(ghi - glo) * k / 8), sizeof(uint32_t),
                                          hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(e, hipStreamSynchronize(e->stream));
        }
        uint32_t mx = 0;
        for (int k = 0; k < 8; ++k) mx = std::max(mx, cut[k + 1] - cut[k]);
        e->interleave = (double)mx > 1.25 * (double)g.nnz / 8.0;
    }
    const uint32_t base_thresh = e->opt_long_thresh ? e->opt_long_thresh : 0xFFFFFFFFu;
    if (e->opt_sorted != 0 && g.nnz) {
        // lockstep cost of natural 64-row tiles (64 x sum of per-tile maxima) against the useful work
        HIP_TRY(e, e->srt_sum.reserve(2));
        const uint32_t heavy_from = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, std::max<uint64_t>(1, 4 * g.nnz / std::max<uint32_t>(ghi - glo, 1)));
        unsigned long long sums[2] = {0, 0};
        if (have_pre && pre.waste && pre.waste_thresh == base_thresh && pre.heavy_from == heavy_from) {
            sums[0] = pre.sums[0];
            sums[1] = pre.sums[1];
        } else {
            HIP_TRY(e, gnnvc::measure_tile_waste(g, glo, ghi, base_thresh, e->srt_sum.p, e->stream, heavy_from));
            HIP_TRY(e, hipMemcpyAsync(sums, e->srt_sum.p, sizeof sums, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(e, hipStreamSynchronize(e->stream));
        }
        const unsigned long long sum_max = sums[0];
        e->srt_waste = 64.0 * (double)sum_max / (double)g.nnz;
        // below a few million entries a 16-wide stage takes tens of microseconds either way and the
        // sort (two kernels and a host round trip) costs more than it saves on a graph used once
        // ... and where the heaviest row of a tile is short anyway (sparse degree-uniform graphs: Poisson(6) has tiles of maximum
        // ~13 against a mean of 6 — "waste" 2.2 — and loses 35 % to the sorted order's uncoalesced rows and per-row records)
        const double mean_tile_max = (double)sum_max / (double)((ghi - glo + 63) / 64);
        // ... and where the graph has no heavy TAIL: a degree-uniform graph with a few hubs also shows "waste" 2.5 - 4.5, but its tiles'
        // maxima are a few times the mean — short chains, the kernel stays bound by the fabric, and sorting costs 10 - 25 %.  What
        // separates the families (fuzz_large.py, 130 graphs): the share of entries in non-long rows of at least 4 x the mean degree —
        // at most 0.05 there, 0.16 and more on power-law and R-MAT graphs (sorted tiles 1.1 - 3 x faster on those).
        e->srt_tail = (double)sums[1] / (double)g.nnz;
        e->sorted_wanted = e->opt_sorted > 0 ||
                           (e->srt_waste >= 2.0 && g.nnz >= e->opt_sorted_min_nnz && mean_tile_max >= 24.0 && e->srt_tail >= 0.10);
    }
    if (!e->opt_long_thresh) return GNNVC_OK;
    // One list at the base threshold serves every stage.  With degree-sorted tiles the 16-wide
    // tile kernel copes with longer rows, so those stages send only rows >= thresh_f16 long
    // (the others return at once from the long kernel and sit in the sorted tile list instead).
    uint32_t thresh = e->opt_long_thresh;
    e->thresh_f16 = 0xFFFFFFFFu;
    HIP_TRY(e, e->long_list.reserve(ghi - glo));
    HIP_TRY(e, e->long_count.reserve(4));
    uint32_t cnt = 0;
    uint32_t found[4] = {0, 0, 0, 0};   // {rows, -, their entries (64 bits)}
    e->long_entries = 0;
    bool few_long = false;
    if (e->opt_long_auto) {
        // A row of d entries holds its tile for d / 3 gather trips (~1.2 us each): with only a few thousand rows above 256 the
        // graph's stages are as long as those tiles (power-law 1 M: 1.16 -> 1.03 ms with the threshold at 256), so they get
        // workgroups of their own; where a hundred thousand rows sit there (R-MAT-22: 110 K) a workgroup each costs more than
        // the tiles (6.2 vs 3.0 ms) and the threshold stays at 512.
        if (have_pre && pre.longs && pre.long_thresh == 256u) {
            for (int k = 0; k < 4; ++k) found[k] = pre.found[k];
        } else {
            HIP_TRY(e, gnnvc::find_long_rows(g, 256u, e->long_list.p, e->long_count.p, e->stream));
            HIP_TRY(e, hipMemcpyAsync(found, e->long_count.p, sizeof found, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(e, hipStreamSynchronize(e->stream));
        }
        cnt = found[0];
        if (cnt == 0) return GNNVC_OK;   // (no row of 256 entries: none of 512 either)
        few_long = cnt <= 16384u && (uint64_t)cnt * 64 <= (uint64_t)(ghi - glo);   // (few, and the exception among the rows: not a dense graph)
        if (few_long) thresh = 256u;
    }
    if (!few_long) {
        if (!e->opt_long_auto && have_pre && pre.longs && pre.long_thresh == thresh) {
            for (int k = 0; k < 4; ++k) found[k] = pre.found[k];
        } else {
            HIP_TRY(e, gnnvc::find_long_rows(g, thresh, e->long_list.p, e->long_count.p, e->stream));
            HIP_TRY(e, hipMemcpyAsync(found, e->long_count.p, sizeof found, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(e, hipStreamSynchronize(e->stream));
        }
        cnt = found[0];
    }
    e->long_entries = (uint64_t)found[2] | ((uint64_t)found[3] << 32);
    if (cnt == 0) return GNNVC_OK;   // nothing long: the tile kernels keep every row
    {
        int rc = ensure_side_streams(e);
        if (rc) return rc;
    }
    e->n_long = cnt;
    e->long_thresh = thresh;
    // (never above the giant threshold: the rows from there on have their own kernels in every stage)
    const uint32_t gt = e->opt_giant_thresh ? std::max(e->opt_giant_thresh, thresh) : 0xFFFFFFFFu;
    e->thresh_f16 = (e->sorted_wanted && !few_long) ? std::max(thresh, std::min(e->opt_sorted_long_thresh, gt)) : thresh;
    return find_giant(e);
}

// The rows of [lo, hi) below the long-row threshold of the 16-wide stages, heaviest degree class first: vertex[] (+ per
// row {first entry, end, W, NW} in meta[]); listed = how many, zero_rows = how many of them (the list's tail) have no entry.
int sort_by_degree_async(gnnvc_engine *e, uint32_t lo, uint32_t hi, DevBuf<uint32_t> &vertex, DevBuf<uint4> &meta, uint32_t *pin_out /* [2] */,
                         const GraphDev *view = nullptr, const uint32_t *skip_rowptr = nullptr, uint32_t skip_from = 0xFFFFFFFFu,
                         uint32_t class_thresh = 0) {
    // view: the adjacency whose row lengths class the rows (a pruned one: the entries LEFT; then skip_rowptr / skip_from leave
    // out the giant rows, which go by their degree).  Everything is queued on the engine's stream — the scan over the degree
    // classes runs on the device — and pin_out[0] = rows listed, pin_out[1] = rows without entries arrive with the stream.
    const GraphDev &g = view ? *view : e->g;
    const uint32_t lt = class_thresh ? class_thresh : e->thresh_f16;
    const uint32_t bins = lt < 4096u ? lt + 1 : 4096u;
    HIP_TRY(e, e->srt_hist.reserve(4096));
    HIP_TRY(e, e->dev_info.reserve(64));
    HIP_TRY(e, vertex.reserve(hi - lo));
    HIP_TRY(e, meta.reserve(hi - lo));
    HIP_TRY(e, gnnvc::degree_histogram(g, lo, hi, lt, bins, e->srt_hist.p, e->stream, skip_rowptr, skip_from));
    HIP_TRY(e, gnnvc::degree_starts(e->srt_hist.p, bins, e->dev_info.p + 8, e->stream));
    HIP_TRY(e, hipMemcpyAsync(pin_out, e->dev_info.p + 8, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, gnnvc::degree_scatter(g, lo, hi, lt, bins, e->srt_hist.p, vertex.p, meta.p, e->stream, skip_rowptr, skip_from));
    return GNNVC_OK;
}

int sort_by_degree(gnnvc_engine *e, uint32_t lo, uint32_t hi, DevBuf<uint32_t> &vertex, DevBuf<uint4> &meta, uint32_t &listed,
                   uint32_t &zero_rows, const GraphDev *view = nullptr, const uint32_t *skip_rowptr = nullptr,
                   uint32_t skip_from = 0xFFFFFFFFu, uint32_t class_thresh = 0) {
    HIP_TRY(e, e->pin_info.reserve(64));
    int rc = sort_by_degree_async(e, lo, hi, vertex, meta, e->pin_info.p + 8, view, skip_rowptr, skip_from, class_thresh);
    if (rc) return rc;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    listed = e->pin_info.p[8];
    zero_rows = e->pin_info.p[9];
    return GNNVC_OK;
}

// Degree-sorted tile order for rows [lo, hi) of the current graph.  A natural tile of 64
// consecutive rows costs max-degree gather rounds; when the measured cost (64 x sum of
// per-tile maxima) exceeds twice the useful work, tiles are formed from a degree-sorted list
// instead.  Cached per row range; the scan over the degree classes runs on the host.
int ensure_sorted(gnnvc_engine *e, uint32_t lo, uint32_t hi) {
    int slot = -1, empty = -1, lru = 0;
    for (int i = 0; i < gnnvc_engine::kSortedRanges; ++i) {
        const auto &r = e->srt[i];
        if (r.valid && r.lo == lo && r.hi == hi) slot = i;
        if (!r.valid && empty < 0) empty = i;
        if (r.valid && r.stamp < e->srt[lru].stamp) lru = i;
    }
    if (slot >= 0) {
        e->srt[slot].stamp = ++e->srt_clock;
        e->srt_cur = slot;
        return GNNVC_OK;
    }
    const int victim = empty >= 0 ? empty : lru;   // an empty entry, else the least recently used one
    gnnvc_engine::SortedRange &sr = e->srt[victim];
    sr.valid = true;
    sr.use = false;
    sr.lo = lo;
    sr.hi = hi;
    sr.n = 0;
    sr.stamp = ++e->srt_clock;
    e->srt_cur = victim;
    if (!e->sorted_wanted || hi <= lo || e->g.nnz == 0) return GNNVC_OK;
    uint32_t zero_rows = 0;
    int rc = sort_by_degree(e, lo, hi, sr.vertex, sr.meta, sr.n, zero_rows);
    if (rc) return rc;
    sr.use = true;
    return GNNVC_OK;
}

// Column-blocked index of the current graph (stage 0 only).  Not used when the
// model is not fused, the graph is small, or a row's block ids are not monotone.
int build_blocked_impl(gnnvc_engine *e) {
    e->blocked_ready = false;
    e->blocked_tried = true;
    const GraphDev &g = e->g;
    if (!e->opt_blocked || e->stages.empty() || e->stages[0].f != 1 || e->stages[0].variant != 0) return GNNVC_OK;
    if (g.sliced()) return GNNVC_OK;   // (the per-graph plans index whole graphs)
    if (g.n < e->opt_blocked_min_n || g.nnz == 0) return GNNVC_OK;
    if (e->opt_blocked_min_n >= (1u << 20) && g.nnz < (uint64_t)g.n * 10) return GNNVC_OK;   // (as for the LDS-table plan: too few entries per row)
    // skewed graphs gather mostly from a few hot (hub) entries of x that stay cached anyway, and
    // the per-row accumulate passes run in lockstep to each wave's largest count: measured slower
    if (e->sorted_wanted && e->opt_blocked < 2) return GNNVC_OK;
    const uint32_t wb = e->opt_block_cols ? e->opt_block_cols : (512u << 10);  // 2 MiB of x per block
    const uint32_t nb = (g.n + wb - 1) / wb;
    if (nb < 2 || nb > 4096) return GNNVC_OK;
    const size_t elems = (size_t)nb * g.n + 1;
    HIP_TRY(e, e->blk_ptr.reserve(elems));
    HIP_TRY(e, e->blk_col.reserve(g.nnz + GNNVC_COL_PAD));
    HIP_TRY(e, e->blk_scratch.reserve(gnnvc::blocked_scan_scratch_elems(elems)));
    HIP_TRY(e, e->blk_flag.reserve(1));
    HIP_TRY(e, e->blk_acc.reserve(g.n));
    HIP_TRY(e, gnnvc::build_blocked_index(g, wb, nb, e->long_thresh, e->blk_ptr.p, e->blk_col.p,
                                          e->blk_scratch.p, e->blk_flag.p, e->stream));
    HIP_TRY(e, hipMemsetAsync(e->blk_col.p + g.nnz, 0, GNNVC_COL_PAD * sizeof(uint32_t), e->stream));
    uint32_t bad = 1;
    HIP_TRY(e, hipMemcpyAsync(&bad, e->blk_flag.p, sizeof bad, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (bad) return GNNVC_OK;  // rows not block-monotone: the blocked order would differ from CSR order
    e->blk_count = nb;
    e->blk_cols = wb;
    e->blocked_ready = true;
    return GNNVC_OK;
}

// Layout of a plan over a SKEWED graph (LDS-table plan of the F = 1 stage, compact-table plan of the 16-wide stages): the rows
// below `class_thresh` entries that have any are dealt serpentine from the degree-sorted list to slices of equal weight
// (rowmap, first), and the column blocks are cut at equal entry mass (bstart) — on a symmetric adjacency the entries that
// point INTO columns [a, b) are as many as the entries of rows [a, b), so the cuts are read off rowptr (for any other
// adjacency they are a heuristic: nothing but the step fill depends on them); block widths between 256 columns and `maxw`,
// cuts at multiples of `align`.
struct SkewedLayout {
    uint32_t plan_rows = 0, rows = 0, chunks = 0, slice_rows = 0, slices = 0, nblocks = 0;
    uint64_t entries = 0;
    gnnvc::PlanMap pm;
};
int layout_skewed_plan(gnnvc_engine *e, uint32_t class_thresh, uint32_t max_rows, uint32_t nsl, double fill, uint32_t maxw, uint32_t align,
                       DevBuf<uint32_t> &rowmap, DevBuf<uint32_t> &first, DevBuf<uint32_t> &bstart, DevBuf<uint32_t> &weights,
                       SkewedLayout &L) {
    const GraphDev &g = e->g;
    uint32_t listed = 0, zero_rows = 0;
    int rc = sort_by_degree(e, 0, g.n, e->c4_map_vertex, e->c4_map_meta, listed, zero_rows, nullptr, nullptr, 0xFFFFFFFFu, class_thresh);
    if (rc) return rc;
    L.plan_rows = listed - zero_rows;   // (rows without entries keep the zeros their sums are initialised with)
    if (L.plan_rows == 0) return GNNVC_OK;
    uint32_t chunks = (L.plan_rows + max_rows - 1) / max_rows;
    chunks = (chunks + 255u) / 256u * 256u;
    uint32_t rows = (L.plan_rows + chunks - 1) / chunks;
    rows = (rows + nsl - 1) / nsl * nsl;
    chunks = (L.plan_rows + rows - 1) / rows;
    L.rows = rows;
    L.chunks = chunks;
    L.slice_rows = rows / nsl;
    L.slices = chunks * nsl;
    const uint32_t slices = L.slices;
    HIP_TRY(e, e->pin_small.reserve((size_t)slices + 4100));
    HIP_TRY(e, rowmap.reserve((size_t)slices * L.slice_rows));
    HIP_TRY(e, first.reserve((size_t)slices + 1));
    HIP_TRY(e, weights.reserve(slices));
    HIP_TRY(e, gnnvc::deal_rows(g, e->c4_map_vertex.p, L.plan_rows, L.slice_rows, slices, rowmap.p, weights.p, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->pin_small.p, weights.p, slices * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    uint64_t run = 0;
    for (uint32_t c = 0; c < slices; ++c) {
        const uint32_t w = e->pin_small.p[c];
        e->pin_small.p[c] = (uint32_t)run;
        run += w;
    }
    e->pin_small.p[slices] = (uint32_t)run;
    L.entries = run;
    if (run == 0 || run >= (1ull << 31)) {
        L.plan_rows = 0;
        return GNNVC_OK;
    }
    HIP_TRY(e, hipMemcpyAsync(first.p, e->pin_small.p, ((size_t)slices + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));   // pin_small is reused below
    const uint32_t minw = std::max(256u, align);
    const double per_slice = (double)L.entries / slices;
    const double nb_want = std::max(1.0, per_slice / fill);
    const unsigned long long target = (unsigned long long)std::max(1.0, (double)g.nnz / nb_want);
    const uint32_t ncand = (uint32_t)std::min<unsigned long long>(g.nnz / target + 2, 4096);
    HIP_TRY(e, bstart.reserve(4100));
    HIP_TRY(e, gnnvc::mass_bounds(g, target, ncand, bstart.p, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->pin_small.p, bstart.p, ncand * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    std::vector<uint32_t> bs{0u};
    auto cut = [&](uint32_t at) {
        while (at - bs.back() > maxw) bs.push_back(bs.back() + maxw);
        if (at > bs.back()) bs.push_back(at);
    };
    for (uint32_t k = 1; k < ncand; ++k) {
        const uint32_t at = std::min(e->pin_small.p[k], g.n) / align * align;
        if (at >= g.n) break;
        if (at > bs.back() && at - bs.back() >= minw) cut(at);
    }
    if (g.n - bs.back() < minw && bs.size() > 1) bs.pop_back();   // no sliver at the end
    cut(g.n);                                                      // bs.back() == g.n: the end of the last block
    L.nblocks = (uint32_t)bs.size() - 1;
    if (L.nblocks == 0 || L.nblocks > 4096) {
        L.plan_rows = 0;
        return GNNVC_OK;
    }
    std::memcpy(e->pin_small.p, bs.data(), bs.size() * sizeof(uint32_t));
    HIP_TRY(e, hipMemcpyAsync(bstart.p, e->pin_small.p, bs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    {   // the block of every 256th column (the builders look a column's block up instead of searching for it)
        const size_t granules = ((size_t)g.n + 255) / 256 + 1;
        std::vector<uint16_t> coarse(granules);
        uint32_t b = 0;
        for (size_t k = 0; k < granules; ++k) {
            const uint64_t c = (uint64_t)k * 256;
            while (b + 1 < L.nblocks && c >= bs[b + 1]) ++b;
            coarse[k] = (uint16_t)b;
        }
        HIP_TRY(e, e->map_coarse.reserve((granules + 1) / 2));
        HIP_TRY(e, hipMemcpyAsync(e->map_coarse.p, coarse.data(), granules * sizeof(uint16_t), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));   // (coarse lives on this stack frame)
    }
    L.pm.rowmap = rowmap.p;
    L.pm.first = first.p;
    L.pm.bstart = bstart.p;
    L.pm.coarse = reinterpret_cast<const uint16_t *>(e->map_coarse.p);
    return GNNVC_OK;
}

// stage launcher shared by the whole-forward and the per-stage entry points
// LDS-table plan of the current graph's F = 1 stage (kernels: k_lt_*).  Applies when every weight fits a
// byte, adjacency lists ascend, no row is long enough for the long-row kernels and the graph is large and
// not skewed; whether a given forward's input really is W / ws is checked on the device at every launch.
// lt_begin: eligibility, geometry, buffers (on e->stream); leaves lt_pb.open set if there is a plan to build.
int lt_begin(gnnvc_engine *e) {
    e->lt_ready = false;
    e->lt_tried = true;
    e->lt_mapped = false;
    gnnvc_engine::PlanBuild &pb = e->lt_pb;
    pb = gnnvc_engine::PlanBuild();
    const GraphDev &g = e->g;
    if (!e->opt_lds_table || e->stages.empty() || e->stages[0].f != 1 || e->stages[0].variant != 0) return GNNVC_OK;
    // The plan covers the rows this engine holds: the whole graph, or (round 3) the SLICE of one rank of a partitioned run —
    // the byte table is made from each forward's x (replicated on every rank), not from the vertex weights, so a slice has
    // everything the plan needs; the columns are always the whole graph's.
    const uint32_t base = g.lo(), end = g.hi(), held = end - base;
    if (e->empty_slice || held == 0) return GNNVC_OK;
    if (g.n < e->opt_blocked_min_n || g.nnz == 0 || g.nnz >= (1ull << 31)) return GNNVC_OK;
    // (every chunk streams the whole byte table whatever the rows hold: below ~10 entries per row the gathering kernel is as fast —
    // 0.56 vs 0.50 ms per forward on an Erdős–Rényi graph of 1.1 M vertices and 8 entries per row)
    if (e->opt_blocked_min_n >= (1u << 20) && g.nnz < (uint64_t)held * 10) return GNNVC_OK;
    // Skewed graphs (sorted tiles wanted, or long rows present): the plan covers the rows below the giant-row threshold, dealt
    // from the degree-sorted list to slices of equal weight, over column blocks of equal entry mass (layout_skewed_plan); the
    // giant rows keep their kernels.  "lds_table" 2 forces the consecutive-row layout onto such a graph instead (tests).
    const bool skewed = e->sorted_wanted || e->n_long > 0;
    // (on a skewed graph most gathers of x go to hubs, which the L2s hold: the plan pays from 2 M vertices on — R-MAT-22 stage 0
    // 1.07 -> 0.86 ms, R-MAT-20 0.27 -> 0.34 ms; a lowered "blocked_min_n" — tests — lowers this bound too)
    const uint32_t skewed_min_n = e->opt_blocked_min_n < (1u << 20) ? e->opt_blocked_min_n : e->opt_lds_skewed_min_n;
    const bool mapped = skewed && e->opt_lds_table < 2 && e->opt_lds_skewed && g.n >= skewed_min_n && !g.sliced();   // (the skewed layout deals whole graphs)
    if (skewed && !mapped && e->opt_lds_table < 2) return GNNVC_OK;   // long runs would serialise in one thread
    // How wide a table entry has to be (round 4): the largest k = W(v) among the rows held here and the weight scale (the
    // original graph's largest weight, src/GNN_VC.cpp:272-278) decide between a byte, ten bits (three to a word) and sixteen bits
    // per vertex — and with it how many vertices a column block's 80 KiB of LDS hold.  Weights beyond 65 535: no table.  (One
    // host round trip, in a hand-off that has several; the skewed layout's blocks are cut for the byte table only.)
    uint32_t bits = 8;
    {
        HIP_TRY(e, e->lt_bad.reserve(2));
        HIP_TRY(e, e->pin_info.reserve(64));
        HIP_TRY(e, gnnvc::lds_table_wmax(g.w + base, held, e->lt_bad.p + 1, e->stream));
        HIP_TRY(e, hipMemcpyAsync(e->pin_info.p + 13, e->lt_bad.p + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        const uint32_t wmax = e->pin_info.p[13];
        const double scale = (e->ws >= 1.0f && e->ws < 4.0e9f) ? (double)e->ws : 0.0;
        bits = e->opt_lt_bits ? (uint32_t)e->opt_lt_bits : gnnvc::lds_table_bits_for(std::max<uint32_t>(wmax, (uint32_t)scale));
        if (bits == 0 || gnnvc::lds_table_bits_for(wmax) == 0 || gnnvc::lds_table_bits_for(wmax) > bits) return GNNVC_OK;
        if (mapped && bits != 8) return GNNVC_OK;
    }
    const uint32_t bc = gnnvc::lds_table_block(bits);
    uint32_t max_rows = gnnvc::lds_table_max_rows(bits);
    if (e->opt_plan_chunk_rows) max_rows = std::min(max_rows, std::max(16u, e->opt_plan_chunk_rows / 16u * 16u));
    uint32_t nblocks = (g.n + bc - 1) / bc, chunks = 0, rows = 0, slice_rows = 0, slices = 0;
    uint64_t plan_nnz = g.nnz;
    if (mapped) {
        // Rows of this many entries and more stay outside the plan.  A row's entries of one block are one run, folded in order
        // lane after lane — a 16 K-entry row makes every step of its slice several times longer (R-MAT-22: k_lt_agg 0.58 ms with
        // rows below 2048, 1.07 ms with everything below the giant rows), while k_long_f1 gathers such rows at ~100 G entries/s as
        // long as x (4 N bytes) mostly sits in the L2s.  Beyond that (R-MAT-24: 67 MB, k_long_f1 at 34 G entries/s) the plan is
        // the better place for them: forward 13.5 -> 12.8 ms.  "lds_table_skewed_rows" overrides.
        const uint32_t below_giant = e->n_giant ? e->giant_thresh : 16384u;
        const uint32_t want = e->opt_lds_skewed_rows ? e->opt_lds_skewed_rows : ((uint64_t)g.n * 4 > (32ull << 20) ? below_giant : 2048u);
        // (the rows that stay outside are the long-row kernels': with no long-row list — no row reaches its threshold, or the list
        // is switched off — there is nobody to sum them, so the plan takes every row.  Found by fuzz_plans.py case 22174 in round 4:
        // "long_row_threshold" 0 with "lds_table_skewed_rows" 64 left the rows of 64 entries and more unwritten in every forward
        // on the plan — invisible while the input stayed the same, since the graph's first forward had written them.)
        e->lt_plan_thresh = e->n_long == 0 ? 0xFFFFFFFFu : std::max(e->long_thresh, std::min(want, below_giant));
        SkewedLayout L;
        int rc = layout_skewed_plan(e, e->lt_plan_thresh, max_rows, 16u, gnnvc::lds_table_step() * 5.0 / 6.0, bc, 16u, e->lt_rowmap,
                                    e->lt_first, e->lt_bstart, e->lt_stepcnt, L);
        if (rc) return rc;
        if (L.plan_rows == 0) return GNNVC_OK;
        chunks = L.chunks;
        rows = L.rows;
        slice_rows = L.slice_rows;
        slices = L.slices;
        nblocks = L.nblocks;
        plan_nnz = L.entries;
        pb.pm = L.pm;
    } else {
        if (nblocks > 4096) return GNNVC_OK;
        // chunks: a multiple of the 256 CUs, each within the LDS budget; a chunk = 16 slices of rows (one per wave).  Every chunk
        // streams the whole byte table once, so a range of FEW rows (a rank's slice at 8 ranks: 64 full chunks) is not cut finer
        // than the CUs need to be busy at half a chunk's rows each — opt_lt_min_chunks — instead of a full multiple of 256.
        chunks = (held + max_rows - 1) / max_rows;
        if (chunks >= 256u) chunks = (chunks + 255u) / 256u * 256u;
        else chunks = std::max(chunks, std::min(e->opt_lt_min_chunks, (held + 255u) / 256u));
        rows = (held + chunks - 1) / chunks;
        rows = (rows + 15u) / 16u * 16u;
        chunks = (held + rows - 1) / rows;
        slice_rows = rows / 16u;
        slices = chunks * 16u;
    }
    const uint32_t slack = 3u * nblocks + 4u;   // every (slice, block) segment starts at a multiple of 4 entries
    const uint64_t entry_cap = plan_nnz + (uint64_t)slack * slices + 8;
    if (entry_cap >= (1ull << 31)) return GNNVC_OK;
    const size_t table_bytes = gnnvc::lds_table_bytes_for(bits, g.n);
    HIP_TRY(e, e->lt_bytes.reserve(table_bytes));
    HIP_TRY(e, e->lt_segcnt.reserve((size_t)slices * nblocks));
    HIP_TRY(e, e->lt_stepcnt.reserve(std::max(chunks, slices)));
    HIP_TRY(e, e->lt_stepptr.reserve((size_t)chunks + 1));
    HIP_TRY(e, e->lt_entries.reserve(entry_cap));
    HIP_TRY(e, e->blk_acc.reserve(g.n));
    uint32_t *flag = e->lt_bad.p + 1;   // word 0 is the per-forward flag
    HIP_TRY(e, hipMemsetAsync(flag, 0, sizeof(uint32_t), e->stream));
    HIP_TRY(e, hipMemsetAsync(e->lt_bytes.p + table_bytes - 80, 0, 80, e->stream));   // (the pad and the last piece: every forward rewrites the entries in front)
    HIP_TRY(e, hipMemsetAsync(e->lt_entries.p, 0, entry_cap * sizeof(uint32_t), e->stream));   // (pad slots are read, never used)
    if (mapped)   // rows outside the plan (no entries, or giant): their sums stay +0 (never read for the giant ones)
        HIP_TRY(e, hipMemsetAsync(e->blk_acc.p, 0, (size_t)g.n * sizeof(float), e->stream));
    // (the table itself is rewritten from x by every forward)
    e->lt_bits = bits;
    pb.open = true;
    pb.mapped = mapped;
    pb.base = base;
    pb.end = end;
    pb.slice_rows = slice_rows;
    pb.slices = slices;
    pb.chunks = chunks;
    pb.rows = rows;
    pb.nblocks = nblocks;
    pb.bc = bc;
    pb.slack = slack;
    pb.entry_cap = entry_cap;
    pb.plan_nnz = plan_nnz;
    pb.done = 0;
    return GNNVC_OK;
}

// count and regroup the entries of slices [done, upto) on `stream` (flat layouts: any sub-range; mapped ones: all at once)
int lt_advance(gnnvc_engine *e, uint32_t upto, hipStream_t stream) {
    gnnvc_engine::PlanBuild &pb = e->lt_pb;
    if (!pb.open) return GNNVC_OK;
    upto = std::min(upto, pb.slices);
    if (upto <= pb.done) return GNNVC_OK;
    uint32_t *flag = e->lt_bad.p + 1;
    HIP_TRY(e, gnnvc::lds_table_count(e->g, pb.slice_rows, pb.slices, pb.nblocks, pb.bc, e->lt_segcnt.p, flag, stream, pb.base, pb.end,
                                      pb.pm, pb.done, upto));
    HIP_TRY(e, gnnvc::lds_table_scatter(e->g, pb.slice_rows, pb.slices, pb.nblocks, pb.bc, e->lt_segcnt.p, e->lt_entries.p, stream, 17, pb.base,
                                        pb.end, pb.slack, pb.pm, flag, pb.done, upto));
    pb.done = upto;
    return GNNVC_OK;
}

// the step records (they need every slice's counts) and the verdict; on e->stream, behind whatever lt_advance queued there
int lt_finish(gnnvc_engine *e) {
    gnnvc_engine::PlanBuild &pb = e->lt_pb;
    if (!pb.open) return GNNVC_OK;
    pb.open = false;
    const GraphDev &g = e->g;
    uint32_t *flag = e->lt_bad.p + 1;
    const uint32_t chunks = pb.chunks;
    HIP_TRY(e, e->pin_small.reserve((size_t)chunks + 2));
    HIP_TRY(e, gnnvc::lds_table_wsteps(g, pb.slice_rows, chunks, pb.nblocks, e->lt_segcnt.p, nullptr, e->lt_stepcnt.p, nullptr, false, pb.slack,
                                       e->stream, pb.pm, pb.base, pb.end, e->lt_bits));
    HIP_TRY(e, hipMemcpyAsync(e->pin_small.p, e->lt_stepcnt.p, chunks * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->pin_small.p + chunks, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->pin_small.p[chunks]) {   // a weight above 255, or an unsorted row (the regrouping would change the order of its sum): no plan
        e->lt_entries.release();
        return GNNVC_OK;
    }
    std::vector<uint32_t> ptr((size_t)chunks + 1, 0);
    uint64_t total = 0;
    for (uint32_t c = 0; c < chunks; ++c) {
        ptr[c] = (uint32_t)total;
        total += e->pin_small.p[c];
    }
    if (total + 8 >= (1ull << 26)) return GNNVC_OK;
    ptr[chunks] = (uint32_t)total;
    const size_t rec_quads = ((total + 8) * gnnvc::lds_table_record_words() + 3) / 4;   // lt_steps counts in 16-byte units
    HIP_TRY(e, e->lt_steps.reserve(rec_quads));
    std::memcpy(e->pin_small.p, ptr.data(), ptr.size() * sizeof(uint32_t));
    HIP_TRY(e, hipMemcpyAsync(e->lt_stepptr.p, e->pin_small.p, ptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemsetAsync(e->lt_steps.p, 0, rec_quads * sizeof(uint4), e->stream));
    HIP_TRY(e, gnnvc::lds_table_wsteps(g, pb.slice_rows, chunks, pb.nblocks, e->lt_segcnt.p, e->lt_stepptr.p, nullptr,
                                       reinterpret_cast<uint32_t *>(e->lt_steps.p), true, pb.slack, e->stream, pb.pm, pb.base, pb.end, e->lt_bits));
    HIP_TRY(e, hipStreamSynchronize(e->stream));   // pin_small is reused by others
    e->c4_map_meta.release();     // (only the dealing needed the list)
    e->c4_map_vertex.release();
    e->lt_last_entry = (uint32_t)((pb.entry_cap - 4) & ~3ull);
    e->lt_rows = pb.rows;
    e->lt_base = pb.base;
    e->lt_end = pb.end;
    e->lt_chunks = chunks;
    e->lt_blocks = pb.nblocks;
    e->lt_steps_total = (uint32_t)total;
    e->lt_mapped = pb.mapped;
    e->lt_ready = true;
    return GNNVC_OK;
}

int build_lds_table_impl(gnnvc_engine *e) {
    int rc = lt_begin(e);
    if (rc == GNNVC_OK) rc = lt_advance(e, 0xFFFFFFFFu, e->stream);
    if (rc == GNNVC_OK) rc = lt_finish(e);
    return rc;
}

// Compact-table plan of the 16-wide stages (kernels: k_c4_*): the same (chunk, column block, step) layout
// as the LDS-table plan at 2 MiB column blocks.  Whether a forward's input really has at most four live
// columns (per pass) is decided on the device at every launch (k_c4_choose / k_c4_compact).
//
// Degree-uniform graphs: slices of consecutive rows, column blocks of one width, one table per input.
// Skewed graphs (sorted tiles wanted, or long rows present; whole-graph plans only): the rows below the long-row
// threshold are dealt from the degree-sorted list to slices of equal weight, the column blocks are cut at equal
// entry mass (hub columns sit in narrow blocks), and an input may take up to three tables — the long and giant rows
// stay with their own kernels beside the plan.
// c4_begin: eligibility, geometry, buffers (on e->stream); leaves c4_pb.open set if there is a plan to build.
int c4_begin(gnnvc_engine *e, uint32_t base, uint32_t end) {
    e->c4_ready = false;
    e->c4_tried = true;
    e->c4_prepared_stage = -1;
    for (bool &b : e->c4_seeded) b = false;
    gnnvc_engine::PlanBuild &pb = e->c4_pb;
    pb = gnnvc_engine::PlanBuild();
    const GraphDev &g = e->g;
    if (end > g.n) end = g.n;
    if (base >= end) return GNNVC_OK;
    if (base < g.lo() || end > g.hi()) return GNNVC_OK;   // rows this engine does not hold (a slice): no plan
    const uint32_t span = end - base;
    if (!e->opt_compact || e->stages.size() < 2) return GNNVC_OK;
    for (size_t st = 1; st < e->stages.size(); ++st)
        if (e->stages[st].f != 16) return GNNVC_OK;
    // (the 64-byte feature rows outgrow the L2s long before x does: the 16-wide stages' plan pays from a quarter of a million vertices on —
    // ER-1M 0.87 -> 0.58 ms per forward — the F = 1 plans from a million)
    if (g.n < std::min(e->opt_blocked_min_n, e->opt_compact_min_n) || g.nnz == 0 || g.nnz >= (1ull << 31)) return GNNVC_OK;
    // (... and the plan's fixed ~0.1 ms per forward needs entries to earn it back: ~25 ps per entry and stage)
    if (e->opt_blocked_min_n >= (1u << 20) && g.nnz < e->opt_compact_min_nnz) return GNNVC_OK;
    // (degree-uniform graphs only.  Rounds 2 - 3 also carried the plan in a layout for SKEWED graphs — rows dealt to slices of
    // equal weight, column blocks of equal entry mass, up to three tables per input, "compact_skewed" — which tied with the
    // gathering kernels at three passes and lost 2 x to the pruned adjacency: removed in round 4.)
    const bool skewed = e->sorted_wanted || e->n_long > 0;
    if (skewed && e->opt_compact < 2) return GNNVC_OK;
    if (e->n_long > 0) return GNNVC_OK;   // (the long-row kernels write their rows themselves)
    // a chunk = 16 slices (one per wave of the workgroup that sums it); the plan is laid out per slice
    const uint32_t nsl = gnnvc::compact_slices();
    uint32_t max_rows = gnnvc::compact_max_rows();
    if (e->opt_plan_chunk_rows) max_rows = std::min(max_rows, std::max(nsl, e->opt_plan_chunk_rows / nsl * nsl));
    uint32_t plan_rows = span;          // rows the plan sums
    uint64_t range_nnz = g.nnz;         // ... and their entries
    // column blocks: wide enough that a slice brings about 160 entries per block (5/6 of a 192-entry step: room for
    // the spread — a segment of 193 costs a second step that the other waves of the workgroup wait for; measured
    // on the metric graph: 0.70 / 0.78 / 0.83 / 0.88 / 0.93 of a step -> 4.71 / 4.66 / 4.65 / 4.82 / 5.26 ms), but at
    // most 160 K vertices = 2.5 MiB of table, which still sits in an XCD's 4 MiB L2 while its 32 CUs sweep it
    const double fill = gnnvc::compact_step() * 5.0 / 6.0;
    uint32_t chunks = 0, rows = 0, slice_rows = 0, slices = 0, bc = gnnvc::compact_block(), nblocks = 0;
    {
        chunks = (plan_rows + max_rows - 1) / max_rows;
        chunks = (chunks + 255u) / 256u * 256u;
        rows = (plan_rows + chunks - 1) / chunks;
        rows = (rows + nsl - 1) / nsl * nsl;
        chunks = (plan_rows + rows - 1) / rows;
        slice_rows = rows / nsl;
        slices = chunks * nsl;
        if (span != g.n) {   // the range's share of the entries (on a slice g.nnz counts the slice's entries only)
            uint32_t rp[2] = {0, 0};
            HIP_TRY(e, hipMemcpyAsync(&rp[0], g.rowptr + base, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(e, hipMemcpyAsync(&rp[1], g.rowptr + end, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(e, hipStreamSynchronize(e->stream));
            range_nnz = rp[1] - rp[0];
            if (range_nnz == 0) return GNNVC_OK;
        }
        const double per_slice = (double)range_nnz / slices;
        const double want = fill * g.n / std::max(per_slice, 1.0);
        bc = (uint32_t)std::min(160.0 * 1024, std::max(32.0 * 1024, want)) / 1024u * 1024u;
        nblocks = (g.n + bc - 1) / bc;
    }
    if (nblocks > 4096) return GNNVC_OK;
    // every (slice, block) segment starts at a multiple of 4 entries: up to 3 pad entries per segment
    const uint32_t slack = 3u * nblocks + 4u;
    const uint64_t entry_cap = g.nnz + (uint64_t)slack * slices + 8;
    if (entry_cap >= (1ull << 31)) return GNNVC_OK;
    const uint32_t passes = 1u;
    constexpr int kFlagWord = 2 * gnnvc_engine::kDescWords;   // after the descriptors of the two consumer stages
    HIP_TRY(e, e->c4_desc.reserve(kFlagWord + 8));
    HIP_TRY(e, e->c4_counts.reserve(16));
    HIP_TRY(e, e->c4_emit_counts.reserve(gnnvc::kEmitCounters));
    HIP_TRY(e, e->c4_segcnt.reserve((size_t)slices * nblocks));
    HIP_TRY(e, e->c4_stepcnt.reserve(slices));
    HIP_TRY(e, e->c4_stepptr.reserve((size_t)slices + 2));
    HIP_TRY(e, e->c4_entries.reserve(entry_cap));
    HIP_TRY(e, e->c4_table.reserve(((size_t)g.n + 1) * 4 * passes));
    HIP_TRY(e, e->c4_marks.reserve(64));
    HIP_TRY(e, e->c4_acc.reserve((size_t)g.n * 4 * passes));
    e->c4_dirty_cap = g.n;   // rows recomputed from full rows (every row could be one: the dense-only stage kernel never gathers)
    HIP_TRY(e, e->c4_dirty.reserve(e->c4_dirty_cap));
    HIP_TRY(e, e->c4_agg16.reserve((size_t)e->c4_dirty_cap * 16));
    HIP_TRY(e, hipMemsetAsync(e->c4_desc.p, 0, (kFlagWord + 8) * sizeof(uint32_t), e->stream));
    HIP_TRY(e, hipMemsetAsync(e->c4_entries.p, 0, entry_cap * sizeof(uint32_t), e->stream));   // (pad slots are read, never used)
    pb.open = true;
    pb.mapped = false;
    pb.base = base;
    pb.end = end;
    pb.slice_rows = slice_rows;
    pb.slices = slices;
    pb.chunks = chunks;
    pb.rows = rows;
    pb.nblocks = nblocks;
    pb.bc = bc;
    pb.slack = slack;
    pb.entry_cap = entry_cap;
    pb.plan_nnz = range_nnz;
    pb.plan_rows = plan_rows;
    pb.passes = passes;
    pb.done = 0;
    return GNNVC_OK;
}

int c4_advance(gnnvc_engine *e, uint32_t upto, hipStream_t stream) {
    gnnvc_engine::PlanBuild &pb = e->c4_pb;
    if (!pb.open) return GNNVC_OK;
    upto = std::min(upto, pb.slices);
    if (upto <= pb.done) return GNNVC_OK;
    uint32_t *flag = e->c4_desc.p + 2 * gnnvc_engine::kDescWords;
    HIP_TRY(e, gnnvc::lds_table_count(e->g, pb.slice_rows, pb.slices, pb.nblocks, pb.bc, e->c4_segcnt.p, flag, stream, pb.base, pb.end, pb.pm,
                                      pb.done, upto));
    HIP_TRY(e, gnnvc::lds_table_scatter(e->g, pb.slice_rows, pb.slices, pb.nblocks, pb.bc, e->c4_segcnt.p, e->c4_entries.p, stream,
                                        gnnvc::compact_shift(), pb.base, pb.end, pb.slack, pb.pm, flag, pb.done, upto));
    pb.done = upto;
    return GNNVC_OK;
}

int c4_finish(gnnvc_engine *e) {
    gnnvc_engine::PlanBuild &pb = e->c4_pb;
    if (!pb.open) return GNNVC_OK;
    pb.open = false;
    const GraphDev &g = e->g;
    const uint32_t slices = pb.slices;
    uint32_t *flag = e->c4_desc.p + 2 * gnnvc_engine::kDescWords;
    HIP_TRY(e, e->pin_small.reserve((size_t)slices + 4100));
    HIP_TRY(e, gnnvc::lds_table_steps(g, pb.slice_rows, slices, pb.nblocks, e->c4_segcnt.p, nullptr, e->c4_stepcnt.p, nullptr, false, e->stream,
                                      pb.base, pb.end, gnnvc::compact_step(), pb.slack, pb.bc, pb.pm));
    HIP_TRY(e, hipMemcpyAsync(e->pin_small.p, e->c4_stepcnt.p, slices * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->pin_small.p + slices, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->pin_small.p[slices]) {   // an unsorted row: regrouping by block would change the order of its sum
        e->c4_entries.release(); e->c4_table.release(); e->c4_acc.release();
        e->c4_dirty.release(); e->c4_agg16.release();
        return GNNVC_OK;
    }
    std::vector<uint32_t> ptr((size_t)slices + 2, 0);
    uint64_t total = 0;
    for (uint32_t c = 0; c < slices; ++c) {
        ptr[c] = (uint32_t)total;
        total += e->pin_small.p[c];
    }
    if (total + 8 >= (1ull << 31)) return GNNVC_OK;
    ptr[slices] = (uint32_t)total;
    ptr[slices + 1] = (uint32_t)total;   // slice `slices`: the empty one idle waves walk
    HIP_TRY(e, e->c4_steps.reserve(total + 8));
    std::memcpy(e->pin_small.p, ptr.data(), ptr.size() * sizeof(uint32_t));
    HIP_TRY(e, hipMemcpyAsync(e->c4_stepptr.p, e->pin_small.p, ptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemsetAsync(e->c4_steps.p + total, 0, 8 * sizeof(uint4), e->stream));
    HIP_TRY(e, gnnvc::lds_table_steps(g, pb.slice_rows, slices, pb.nblocks, e->c4_segcnt.p, e->c4_stepptr.p, nullptr, e->c4_steps.p, true, e->stream,
                                      pb.base, pb.end, gnnvc::compact_step(), pb.slack, pb.bc, pb.pm));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    e->c4_map_meta.release();     // (only the dealing needed the list)
    e->c4_map_vertex.release();
    e->c4_last_entry = (uint32_t)((pb.entry_cap - 4) & ~3ull);
    e->c4_block = pb.bc;
    e->c4_nblocks = pb.nblocks;
    e->c4_base = pb.base;
    e->c4_end = pb.end;
    e->c4_rows = pb.rows;
    e->c4_chunks = pb.chunks;
    e->c4_nslices = slices;
    e->c4_steps_total = (uint32_t)total;
    e->c4_ready = true;
    return GNNVC_OK;
}

int build_compact_impl(gnnvc_engine *e, uint32_t base, uint32_t end) {
    int rc = c4_begin(e, base, end);
    if (rc == GNNVC_OK) rc = c4_advance(e, 0xFFFFFFFFu, e->stream);
    if (rc == GNNVC_OK) rc = c4_finish(e);
    return rc;
}

gnnvc::CompactPlan compact_plan(const gnnvc_engine *e) {
    gnnvc::CompactPlan cp;
    cp.rows_per_chunk = e->c4_rows;
    cp.block_cols = e->c4_block;
    cp.nblocks = e->c4_nblocks;
    cp.plan_base = e->c4_base;
    cp.plan_end = e->c4_end;
    cp.last_entry = e->c4_last_entry;
    cp.nslices = e->c4_nslices;
    cp.max_passes = 1;
    cp.step_ptr = e->c4_stepptr.p;
    cp.steps = e->c4_steps.p;
    cp.entries = e->c4_entries.p;
    cp.rowmap = nullptr;
    return cp;
}

int build_blocked(gnnvc_engine *e) {
    return timed_build(e, [&] { return build_blocked_impl(e); }, e->opt_blocked && e->g.n >= e->opt_blocked_min_n && e->g.nnz != 0);
}
int build_lds_table(gnnvc_engine *e) {
    return timed_build(e, [&] { return build_lds_table_impl(e); }, e->opt_lds_table && e->g.n >= e->opt_blocked_min_n && e->g.nnz != 0);
}
int build_compact(gnnvc_engine *e, uint32_t base, uint32_t end) {
    return timed_build(e, [&] { return build_compact_impl(e, base, end); },
                       e->opt_compact && e->g.n >= std::min(e->opt_blocked_min_n, e->opt_compact_min_n) && e->g.nnz != 0);
}

// The engine's SIDE queue, made once per engine (gnnvc_create): what runs beside the main stream's kernels — the dense part of a
// round under the next round's sums, the long and giant rows beside the tile kernel, the plan builders of a hand-off beside the
// copy.  ONE side queue, not one per purpose (round 3): with the long rows, the giant rows and the overlapped rounds each on a
// stream of their own, where the runtime placed those streams decided the forward's time — power-law 1 M 0.91 or 1.49 ms, R-MAT-22
// 2.66 or 2.85, R-MAT-20 0.80 or 1.05 - 1.5 — by how many other engines the process had alive
// (scratch/experiments/queue_pressure.py; priorities did not stabilise it).  A HIP stream goes to the hardware queue with the
// fewest users, ties included, which can be the main stream's own queue: then the two are serialised (R-MAT-22 3.44 ms) — so
// the candidate is PROBED (streams_run_side_by_side, 0.3 ms) and replaced until one runs beside the main stream.  Creating a
// stream costs ~10 ms on this stack (scratch/experiments/stream_cost.py): none is made inside a hand-off or a forward.
int ensure_side_streams(gnnvc_engine *e) {
    if (e->aux_stream) return GNNVC_OK;
    hipStream_t tried[4] = {nullptr, nullptr, nullptr, nullptr};
    int n = 0;
    hipStream_t good = nullptr;
    while (n < 4 && !good) {
        HIP_TRY(e, hipStreamCreateWithFlags(&tried[n], hipStreamNonBlocking));
        bool beside = false;
        HIP_TRY(e, gnnvc::streams_run_side_by_side(e->stream, tried[n], &beside));
        if (beside) good = tried[n];
        ++n;
    }
    e->side_probes = n;
    e->side_beside = good != nullptr;
    if (!good) good = tried[n - 1];   // (none passed: a serialised side queue is slow, not wrong)
    for (int i = 0; i < n; ++i)
        if (tried[i] != good) (void)hipStreamDestroy(tried[i]);
    e->aux_stream = e->long_stream = e->giant_stream = good;
    HIP_TRY(e, hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    HIP_TRY(e, hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    HIP_TRY(e, hipEventCreateWithFlags(&e->ev_long, hipEventDisableTiming));
    HIP_TRY(e, hipEventCreateWithFlags(&e->ev_giant, hipEventDisableTiming));
    return GNNVC_OK;
}

// The caller installed another main stream (gnnvc_set_stream): the side queue was probed against the engine's own.  The runtime
// may have put the caller's stream on the side queue's hardware queue — the two would then run one after the other and nothing
// would say so — so the pair is probed again, and the side queue replaced (up to four candidates) if it fails.
// Streams the forward's events refer to are only swapped while nothing of a forward is in flight (the caller's contract for
// gnnvc_set_stream: between forwards).
int reprobe_side_streams(gnnvc_engine *e) {
    if (!e->aux_stream) return GNNVC_OK;   // (no side queue yet: ensure_side_streams will probe against the stream in force)
    bool beside = false;
    HIP_TRY(e, gnnvc::streams_run_side_by_side(e->stream, e->aux_stream, &beside));
    e->side_probes = 1;
    if (!beside) {
        hipStream_t tried[4] = {nullptr, nullptr, nullptr, nullptr};
        int n = 0;
        hipStream_t good = nullptr;
        while (n < 4 && !good) {
            HIP_TRY(e, hipStreamCreateWithFlags(&tried[n], hipStreamNonBlocking));
            bool ok = false;
            HIP_TRY(e, gnnvc::streams_run_side_by_side(e->stream, tried[n], &ok));
            if (ok) good = tried[n];
            ++n;
        }
        e->side_probes += n;
        for (int i = 0; i < n; ++i)
            if (tried[i] != good) (void)hipStreamDestroy(tried[i]);
        if (good) {
            (void)hipStreamSynchronize(e->aux_stream);
            (void)hipStreamDestroy(e->aux_stream);
            e->aux_stream = e->long_stream = e->giant_stream = good;
            beside = true;
        }
    }
    e->side_beside = beside;
    return GNNVC_OK;
}

int ensure_round_events(gnnvc_engine *e, size_t count) {
    {
        int rc = ensure_side_streams(e);
        if (rc) return rc;
    }
    while (e->round_ev.size() < count) {
        hipEvent_t v;
        HIP_TRY(e, hipEventCreateWithFlags(&v, hipEventDisableTiming));
        e->round_ev.push_back(v);
    }
    return GNNVC_OK;
}

// Buffers of the pruned adjacency sized by what the GRAPH allows (the kept entries are at most all of them), so that a plan
// built inside a forward allocates nothing there.
int reserve_prune(gnnvc_engine *e, int stage) {
    gnnvc_engine::PrunePlan &pp = e->prune[stage];
    const GraphDev &g = e->g;
    const uint32_t held = g.hi() - g.lo();
    const size_t chunks = (size_t)((g.nnz + 63) / 64);
    HIP_TRY(e, e->prune_flags.reserve(8));
    HIP_TRY(e, e->pin_info.reserve(64));
    HIP_TRY(e, e->dev_info.reserve(64));
    HIP_TRY(e, pp.heavy.reserve(((size_t)g.n + 31) / 32 + 1));
    if (e->opt_filter) {
        HIP_TRY(e, e->filter_bits[stage].reserve((size_t)g.n / 32 + 1));
        HIP_TRY(e, e->filter_info.reserve(16));
    }
    HIP_TRY(e, e->prune_mask.reserve(std::max<size_t>(chunks, 4)));   // (prune_mass keeps three 64-bit words there: with <= 128 entries two were too few — fuzz_multi.py)
    HIP_TRY(e, e->prune_off.reserve(chunks + 1));
    HIP_TRY(e, e->prune_scratch.reserve(gnnvc::blocked_scan_scratch_elems(chunks + 1)));
    HIP_TRY(e, pp.prp.reserve((size_t)held + 1));
    HIP_TRY(e, pp.pcol.reserve((size_t)g.nnz + GNNVC_COL_PAD));
    if (e->sorted_wanted && e->opt_prune_eff) {
        HIP_TRY(e, pp.svertex.reserve(held));
        HIP_TRY(e, pp.smeta.reserve(held));
        HIP_TRY(e, e->srt_hist.reserve(4096));
    }
    return GNNVC_OK;
}

// Pruned adjacency for consumer stage `stage` (see k_prune_*): built once per graph from the input `in` of the call at hand.
// Two host round trips: the size of the set and what it promises (before any pass over the entries), and what came of it.
int build_prune_impl(gnnvc_engine *e, int stage, const float *in, bool early, bool predicted = false) {
    gnnvc_engine::PrunePlan &pp = e->prune[stage];
    pp.tried = true;
    pp.ready = false;
    pp.predicted = pp.verified = false;
    const GraphDev &g = e->g;
    if (!e->opt_prune || g.n == 0 || e->empty_slice || g.nnz == 0 || g.nnz < e->opt_prune_min_nnz) return GNNVC_OK;   // (no entries: nothing to prune — and no buffers: fuzz_multi.py, a part of rows without entries under "prune_min_entries" 0)
    if (predicted && (g.sliced() || stage != 1 || e->stages.size() < 2)) return GNNVC_OK;
    if (g.nnz >= (1ull << 32)) return GNNVC_OK;
    int rc = reserve_prune(e, stage);
    if (rc) return rc;
    uint32_t *pin = e->pin_info.p;   // [0..3] = {mass lo, mass hi, members lo, members hi}, [4] = kept, [6..7] = listed rows, rows without entries
    if (predicted) {
        // the set the GRAPH predicts (hand-off: there is no input yet), see k_predict_zero_f1
        HIP_TRY(e, gnnvc::predict_zero_rows(e->stages[0], g, e->ws, e->params.p, pp.heavy.p, e->stream));
    } else {
        // the very vertices whose rows are all zero in this input (a graph's stage inputs follow from its weights: the same on
        // every forward; any other input fails the check and is served by the full adjacency).  (Rounds 2 - 3 also offered a
        // degree bound as the set, "prune_zero_rows" 2: it drops fewer entries — R-MAT-22's last stage kept 25 % instead of 14 % —
        // and was slower on every graph measured; removed in round 4.)
        HIP_TRY(e, gnnvc::prune_mark_zero(g, in, pp.heavy.p, e->stream));
    }
    // a cheap look before the passes over the entries: how many vertices the set has, and (whole graphs) their degrees ~ the
    // entries that would go
    unsigned long long *mass_dev = reinterpret_cast<unsigned long long *>(e->prune_mask.p);
    // (the stage before this one, if it has a plan built from THIS graph's set of zero rows: does this stage's set contain it?)
    const gnnvc_engine::PrunePlan *prev = (stage >= 2 && !predicted && e->prune[stage - 1].ready && !g.sliced()) ? &e->prune[stage - 1] : nullptr;
    HIP_TRY(e, gnnvc::prune_mass(g, pp.heavy.p, mass_dev, e->stream, prev ? prev->heavy.p : nullptr));
    HIP_TRY(e, hipMemcpyAsync(pin, mass_dev, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(pin + 10, mass_dev + 2, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    const uint64_t mass = (uint64_t)pin[0] | ((uint64_t)pin[1] << 32), members = (uint64_t)pin[2] | ((uint64_t)pin[3] << 32);
    // A stage's zero rows mostly stay zero rows in the next stage (the model drives the hubs' features to zero and keeps them
    // there): when this stage's set CONTAINS the previous stage's, the entries this plan keeps are among the ones that plan
    // kept, and the passes below walk its 0.14 - 0.27 x nnz entries instead of all of them (R-MAT-22, last stage: 35 M of 128 M).
    const bool from_prev = prev && (((uint64_t)pin[10] | ((uint64_t)pin[11] << 32)) == 0) && prev->kept > 0 && prev->kept < g.nnz;
    GraphDev src = g;   // the adjacency the kept entries are taken from
    if (from_prev) {
        src.rowptr = prev->prp.p - g.lo();
        src.col = prev->pcol.p;
        src.nnz = prev->kept;
    }
    pp.from_prev = from_prev;
    pp.members = members;
    if (members == 0) return GNNVC_OK;   // degree-uniform graphs: every row has a non-zero, nothing to prune
    if (!g.sliced()) {
        if (mass * 100 < g.nnz * (uint64_t)std::min(e->opt_prune_min_drop, 100u) / 2)   // (half the bound: the estimate is exact only for symmetric graphs)
            return GNNVC_OK;
        if (early && mass * 100 < g.nnz * 40ull) {   // in a graph's FIRST forward the build has to pay within that forward: not at a 15 - 25 % cut
            pp.tried = false;                         // (nearly degree-uniform graphs with hubs: first forward 1.1 - 1.2 x) — with the other plans, then
            pp.deferred = true;
            return GNNVC_OK;
        }
    }
    const size_t src_chunks = (size_t)((src.nnz + 63) / 64);
    HIP_TRY(e, gnnvc::prune_count(src, pp.heavy.p, e->prune_mask.p, e->prune_off.p, e->prune_scratch.p, e->stream));
    HIP_TRY(e, hipMemcpyAsync(pin + 4, e->prune_off.p + src_chunks, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    // (pcol holds room for every entry + the pad: reads a little past the kept entries — the gathering kernels' look-ahead —
    // stay inside it whatever the count turns out to be; what they find there is masked, never used as an index)
    HIP_TRY(e, gnnvc::prune_fill(src, e->prune_mask.p, e->prune_off.p, pp.pcol.p, pp.prp.p, e->stream));
    pp.slist = false;
    pp.sn = 0;
    // Rows are classed by the entries they have LEFT.  The tile kernel walks a row's entries three at a time, ~1.2 us a trip:
    // a 1000-entry row holds its tile for ~0.4 ms — nothing beside the ~0.7 ms R-MAT-22's stage spends gathering, but the
    // whole stage on the power-law graph (8 M entries left: 0.57 -> 0.64 ms at 1024; R-MAT-22: 1.35 -> 1.25 ms).
    // (decided from the estimate of the entries left: the count itself arrives with the second round trip)
    const uint64_t kept_est = g.sliced() ? g.nnz : g.nnz - std::min<uint64_t>(mass, g.nnz);
    pp.eff_thresh = kept_est >= e->opt_prune_heavy_entries ? e->thresh_f16 : std::max(e->long_thresh, std::min(e->thresh_f16, 512u));
    const bool slist = e->sorted_wanted && e->opt_prune_eff;
    if (slist) {
        // tiles of the 16-wide stages from the rows sorted by the entries they have LEFT; the giant rows (by degree) are not in it
        GraphDev view = g;
        view.rowptr = pp.prp.p - g.lo();
        view.col = pp.pcol.p;
        rc = sort_by_degree_async(e, g.lo(), g.hi(), pp.svertex, pp.smeta, pin + 6, &view, g.rowptr, e->giant_f16(), pp.eff_thresh);
        if (rc) return rc;
    }
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    const uint32_t kept = pin[4];
    pp.kept = kept;
    if ((uint64_t)kept * 100 > g.nnz * (uint64_t)(100 - std::min(e->opt_prune_min_drop, 100u))) return GNNVC_OK;   // too little to gain
    if (slist) {
        pp.sn = pin[6];
        pp.slist = true;
    }
    pp.ready = true;
    pp.predicted = predicted;
    return GNNVC_OK;
}

// The graph as this call's gathering kernels see it: with the pruned adjacency attached when the stage has one (the
// check of this very input is queued here, ahead of every kernel that reads its verdict).  so_p: the tile order that goes
// with classing the rows by the entries they have left (natural tiles need none).
int gather_view(gnnvc_engine *e, int stage, uint32_t lo, uint32_t hi, const float *in, bool gathering, bool sorted_tiles,
                GraphDev &gv, gnnvc::SortedOrder &so_p, bool matrix_cores, uint32_t long_from) {
    gv = e->g;
    so_p = gnnvc::SortedOrder();
    if (stage < 1 || stage > 3 || e->stages[stage].f != 16 || !e->opt_prune) return GNNVC_OK;
    gnnvc_engine::PrunePlan &pp = e->prune[stage];
    if (!gathering) return GNNVC_OK;   // (a compact-table plan has this call: its kernels do not gather, nothing to build or check)
    // Built with the rest of the plans when the graph is scored a second time — but on a LARGE skewed graph (sorted tiles or
    // long rows: that is where zero rows are found) already the first time its stage runs: the passes cost less than the
    // gathers they save there (first forward R-MAT-22 6.95 -> 6.16 ms, R-MAT-24 34.5 -> 25.9 ms; R-MAT-20 and the power-law
    // graph lose 0.3 - 0.5 ms to the fixed costs, hence the size bound).
    // (round 3: that is the rule with "filter_zero_rows" off.  With it — the default — a skewed graph's first forward builds
    // nothing: its kernels skip the zero rows by looking them up, see below, and the plan is built if the graph comes back.)
    const bool skewed = e->sorted_wanted || e->n_long > 0;
    // (the tile kernel's filtered instantiations are those of the matrix-core variants, the default of the 16-wide stages)
    // ... on graphs whose LONG rows hold a good share of the entries (known since the hand-off): that is where the model zeroes
    // most targets (R-MAT: 73 - 86 % of the entries point to zero rows; power-law graphs and uniform graphs with hubs, 1 - 55 %,
    // lose 0.1 - 0.3 ms of their first forward to the look-ups)
    const bool filter = e->opt_filter && matrix_cores && skewed && e->g.nnz >= e->opt_filter_min_nnz && e->g.nnz < (1ull << 32) &&
                        e->n_long > 0 && e->long_entries * 100ull >= e->g.nnz * (uint64_t)e->opt_filter_min_long_pct;
    const bool early = !filter && skewed && e->opt_prune_early_nnz && e->g.nnz >= e->opt_prune_early_nnz;
    const uint32_t uses_needed = (early && !pp.deferred) ? 1u : 2u;
    e->filtered[stage] = e->short_used[stage] = e->borrowed[stage] = false;
    if (e->short_from >= stage) e->short_from = 0;   // (the stage that left the lists runs again: they are this call's to leave, or nobody's)
    if (pp.predicted && !pp.verified && e->graph_uses >= 2) {
        // A plan built at hand-off from the PREDICTED set has served a forward: did its check pass there?  If the caller's input
        // is not what the prediction assumed, the set is wrong for this graph's stage inputs (which are the same on every
        // forward) and every call would fall back to the full adjacency: the plan goes, and is rebuilt below from the input the
        // stage really sees.  One host round trip per graph, in a forward that builds the next stage's plan anyway.
        uint32_t *pin = e->pin_info.p;
        HIP_TRY(e, hipMemcpyAsync(pin + 12, e->prune_flags.p + stage, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        if (pin[12] != 0u) pp.forget();
        else pp.verified = true;
    }
    if (!pp.tried && e->graph_uses >= uses_needed) {
        const bool first_forward = e->graph_uses < 2;
        int rc = timed_build(e, [&] { return build_prune_impl(e, stage, in, first_forward); },
                             e->opt_prune && e->g.n != 0 && e->g.nnz >= e->opt_prune_min_nnz);
        if (rc) return rc;
    }
    // A graph's FIRST forward, the stage behind one that runs on a predicted plan: it borrows that plan — the model keeps the hubs'
    // rows at zero from stage to stage (R-MAT-22: the earlier set is contained in this stage's), its own check against THIS input
    // decides — until the graph comes back and it gets its own (built from the borrowed plan's kept entries: from_prev).
    const gnnvc_engine::PrunePlan *use = &pp;
    if (!pp.ready && !pp.tried && stage >= 2 && e->prune[stage - 1].ready && e->prune[stage - 1].predicted &&
        e->stages[stage - 1].f == 16)
        use = &e->prune[stage - 1];
    if (!use->ready) {
        // no plan (yet, or the graph has too few zero rows for one): the bitmap of this input's zero rows for the kernels to
        // look into — in a graph's first forward, and later only while a plan may still come (tried && !ready: it was found
        // not worth it, and neither is a look-up per entry)
        if (filter && !pp.tried) {
            const GraphDev &g = e->g;
            HIP_TRY(e, e->filter_bits[stage].reserve((size_t)g.n / 32 + 1));
            HIP_TRY(e, e->filter_info.reserve(16));
            // an earlier stage of this graph left its long rows' lists — of every row this call's long-row kernel will take?
            const int from = e->short_from;
            const uint32_t giant_from = e->giant_f16();
            const bool consume = from >= 1 && from < stage && long_from >= e->short_min && giant_from <= e->short_max;
            // (gv, not g: the verdict on the earlier stage's lists has to be reached with the bound that stage's kernels used)
            gv.zero_min_pct = e->opt_filter_min_pct;
            HIP_TRY(e, gnnvc::filter_mark(gv, in, e->filter_bits[stage].p, e->filter_info.p + 4 * stage, e->stream,
                                          consume ? e->filter_bits[from].p : nullptr, consume ? e->filter_info.p + 4 * from : nullptr));
            gv.zero_bits = e->filter_bits[stage].p;
            gv.zero_info = e->filter_info.p + 4 * stage;
            e->filtered[stage] = true;
            if (e->opt_filter_keep && e->n_long > 0) {
                // the long rows' short lists of this call go where the plan's entries will go once it is built
                HIP_TRY(e, pp.pcol.reserve((size_t)g.nnz + GNNVC_COL_PAD));
                HIP_TRY(e, pp.prp.reserve((size_t)(g.hi() - g.lo()) + 1));
                gv.keep_col = pp.pcol.p;
                gv.keep_cnt = pp.prp.p - g.lo();
                if (consume) {
                    gv.short_col = e->prune[from].pcol.p;
                    gv.short_cnt = e->prune[from].prp.p - g.lo();
                    gv.short_bad = reinterpret_cast<const uint32_t *>(e->filter_info.p + 4 * stage + 2);
                    e->short_used[stage] = true;
                }
                // a call that sees every row the engine holds leaves them for the next 16-wide stage, if there is one
                if (lo == g.lo() && hi == g.hi() && stage + 1 < (int)e->stages.size() && stage + 1 <= 3 && e->stages[stage + 1].f == 16) {
                    e->short_from = stage;
                    e->short_min = long_from;
                    e->short_max = giant_from;
                }
            }
        }
        return GNNVC_OK;
    }
    HIP_TRY(e, gnnvc::prune_check(e->g, in, use->heavy.p, e->prune_flags.p + stage, e->stream));
    gv.prp = use->prp.p - e->g.lo();   // (indexed by global row id, like rowptr)
    gv.pcol = use->pcol.p;
    gv.prune_bad = e->prune_flags.p + stage;
    if (e->opt_prune_eff) {
        const bool whole = lo == e->g.lo() && hi == e->g.hi();
        if (!sorted_tiles) {
            gv.prune_eff = 1;
        } else if (whole && use->slist) {
            gv.prune_eff = 1;
            so_p.n = use->sn;
            so_p.vertex = use->svertex.p;
            so_p.meta = use->smeta.p;
        }
        gv.eff_giant = e->giant_f16();
        gv.eff_thresh = use->eff_thresh;
    }
    e->borrowed[stage] = use != &pp;
    return GNNVC_OK;
}

int reserve_features(gnnvc_engine *e, uint32_t n) {
    if (e->g.sliced()) return GNNVC_OK;   // a slice is driven stage by stage on the caller's replicated buffers
    const size_t rows = (size_t)n + 1;
    HIP_TRY(e, e->x.reserve(rows * (size_t)e->in_width));
    HIP_TRY(e, e->scores.reserve(rows * (size_t)e->out_width));
    HIP_TRY(e, e->logits.reserve(rows * (size_t)e->out_width));
    if (!e->stages.empty()) {
        for (auto &b : e->h) HIP_TRY(e, b.reserve(rows * 16));
        // the pad rows (index n) of the 16-wide feature buffers must read as zero; no kernel ever writes them, so once per
        // graph will do (not two launches per forward: a small graph's forward is a handful of launches in all)
        for (auto &b : e->h) HIP_TRY(e, gnnvc::launch_zero_pad_row(b.p, n, 16, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));   // (a later forward may run on another stream: gnnvc_set_stream)
    }
    return GNNVC_OK;
}

// the front of a multi-device handle keeps the input and the assembled scores / logits of the host-pointer forward on its device
int reserve_multi_front(gnnvc_engine *e, uint32_t n) {
    const size_t rows = (size_t)n + 1;
    HIP_TRY(e, e->x.reserve(rows * (size_t)e->in_width));
    HIP_TRY(e, e->scores.reserve(rows * (size_t)e->out_width));
    HIP_TRY(e, e->logits.reserve(rows * (size_t)e->out_width));
    return GNNVC_OK;
}

int ensure_events(gnnvc_engine *e, size_t count) {
    while (e->ev.size() < count) {
        hipEvent_t v;
        HIP_TRY(e, hipEventCreate(&v));
        e->ev.push_back(v);
    }
    return GNNVC_OK;
}

// What a forward needs that depends on the GRAPH alone, made when the graph is handed over (round 3): the
// reference's driver scores every graph once (src/GNN_VC.cpp:171-192), and a plan built inside a later forward never
// serves such a caller.  Runs behind find_long (which classed the graph) on the engine's stream; complete when it returns.
// Table tiles (k_stage_t4): which graphs, and their buffers.  Nothing is built: the tables are written by the forwards' own
// kernels; the graph only has to be of the size where a 16-byte-per-vertex table fits an L2 and 64-byte rows do not, degree-uniform
// (no long rows, natural tiles) and outside the compact-table plan's range.
int prepare_table_tiles(gnnvc_engine *e) {
    e->t4_ok = false;
    e->t4_unfit_runs = 0;
    e->t4_used = false;
    for (bool &b : e->t4_fit_seen) b = false;
    const GraphDev &g = e->g;
    if (!e->opt_t4 || !e->opt_compact || e->opt_mfma == 1 || e->stages.size() != 3 || g.n == 0 || e->empty_slice || g.sliced()) return GNNVC_OK;
    if (e->stages[0].variant != 0 || e->stages[1].variant != 1 || e->stages[2].variant != 2) return GNNVC_OK;
    if (e->n_long > 0 || e->sorted_wanted || g.nnz == 0) return GNNVC_OK;
    if (g.n < e->opt_t4_min_n || ((uint64_t)g.n + 1) * 16 > e->opt_t4_max_bytes) return GNNVC_OK;
    const bool plan_range = g.n >= std::min(e->opt_blocked_min_n, e->opt_compact_min_n) &&
                            (e->opt_blocked_min_n < (1u << 20) || g.nnz >= e->opt_compact_min_nnz);
    if (plan_range) return GNNVC_OK;   // (the compact-table plan has these graphs)
    for (auto &t : e->t4_table) {
        HIP_TRY(e, t.reserve(((size_t)g.n + 1) * 4));
        HIP_TRY(e, hipMemsetAsync(t.p, 0, ((size_t)g.n + 1) * 4 * sizeof(float), e->stream));   // (row n: the zero row padded slots read)
    }
    for (auto &c : e->t4_counts) {
        HIP_TRY(e, c.reserve(2 * gnnvc::kEmitCounters));
        HIP_TRY(e, hipMemsetAsync(c.p, 0, 2 * gnnvc::kEmitCounters * sizeof(unsigned long long), e->stream));
    }
    // The CHOICE of columns survives a change of graph: the reference's driver scores every graph once (src/GNN_VC.cpp:171-192), its
    // graphs are shrinking versions of one another and which columns the trained model leaves live hardly depends on the graph —
    // so the producers of the new graph's FIRST forward write their tables for the previous graph's last choice (any choice gives
    // the same bits: it only decides how few rows meet a flagged neighbour).  An engine's first qualifying graph starts without.
    if (!e->t4_choice_live) {
        HIP_TRY(e, e->t4_desc.reserve(2 * 2 * 16));
        HIP_TRY(e, hipMemsetAsync(e->t4_desc.p, 0, 2 * 2 * 16 * sizeof(uint32_t), e->stream));
        e->t4_parity = 0;
    }
    e->t4_ok = true;
    return GNNVC_OK;
}

int prepare_plans(gnnvc_engine *e) {
    e->handoff_build_ms = 0.0;
    {
        const int rc = prepare_table_tiles(e);
        if (rc) return rc;
    }
    if (!e->opt_handoff || e->stages.empty() || e->g.n == 0 || e->empty_slice) return GNNVC_OK;
    const GraphDev &g = e->g;
    const double before = e->plan_build_ms;
    const bool skewed = e->sorted_wanted || e->n_long > 0;
    int rc = GNNVC_OK;
    HIP_TRY(e, e->pin_info.reserve(64));
    HIP_TRY(e, e->dev_info.reserve(64));
    // the tile order of the engine's rows: the first forward needs it anyway (whole-graph calls, a rank's whole slice)
    if (e->sorted_wanted) {
        rc = timed_build(e, [&] { return ensure_sorted(e, g.lo(), g.hi()); });
        if (rc) return rc;
    }
    // The per-graph plans.  By default only where ONE use repays the build: degree-uniform graphs from opt_handoff_min_nnz
    // entries on (metric graph: ~3.5 ms of builds against 4.3 ms saved in the very first forward; a skewed graph's F = 1 plan
    // costs 9 - 29 ms to build and saves 0.2 - 2 ms a forward — it keeps waiting for a second forward unless asked for, "2").
    if (e->lt_pb.open || e->c4_pb.open) {   // begun while the column array was arriving (handoff_early): the rest, and the step records
        rc = timed_build(e, [&] {
            int r = lt_advance(e, 0xFFFFFFFFu, e->stream);
            if (r == GNNVC_OK) r = c4_advance(e, 0xFFFFFFFFu, e->stream);
            if (r == GNNVC_OK) r = lt_finish(e);
            if (r == GNNVC_OK) r = c4_finish(e);
            return r;
        });
        if (rc) return rc;
    }
    if (!g.sliced() && (e->opt_handoff >= 2 || (!skewed && g.nnz >= e->opt_handoff_min_nnz))) {
        if (!e->lt_tried) rc = build_lds_table(e);
        if (rc) return rc;
        if (!e->c4_tried && !e->c4_range_mode) rc = build_compact(e);
        if (rc) return rc;
    }
    // pruned adjacency: it needs a stage's INPUT and is built inside a forward — with every buffer it wants already here
    if (skewed && e->opt_prune && g.nnz >= e->opt_prune_min_nnz && g.nnz < (1ull << 32) && true) {
        for (int st = 1; st < (int)e->stages.size() && st < 4; ++st)
            if (e->stages[st].f == 16) {
                rc = reserve_prune(e, st);
                if (rc) return rc;
            }
        // ... except the first 16-wide stage's on the graphs whose first forward would otherwise look every target up (the
        // filtered gather's graphs: large, their long rows hold a good share of the entries): the reference's driver feeds
        // x = W / ws (src/GNN_VC.cpp:189-191), so that stage's zero rows follow from the graph's own weights, and the plan for the
        // PREDICTED set is built here — proven per call like any other.  R-MAT-22: +0.9 ms of hand-off, first forward 3.5 -> 2.8 ms.
        if (e->opt_prune_predict && !g.sliced() && g.nnz >= e->opt_predict_min_nnz && e->n_long > 0 &&
            e->long_entries * 100ull >= g.nnz * (uint64_t)e->opt_filter_min_long_pct && e->stages.size() >= 2 &&
            e->stages[0].variant == 0 && e->stages[1].f == 16 && !e->prune[1].tried) {
            rc = timed_build(e, [&] { return build_prune_impl(e, 1, nullptr, /*early=*/true, /*predicted=*/true); });
            if (rc) return rc;
            if (!e->prune[1].ready) e->prune[1].forget();   // (not worth it by the prediction: the first forward filters, a second one observes)
        }
    }
    if (!g.sliced()) {
        rc = ensure_events(e, e->stages.size() + 1);
        if (rc) return rc;
    }
    e->handoff_build_ms = e->plan_build_ms - before;
    return GNNVC_OK;
}

// per-graph state of the plans: nothing of the previous graph's survives
void reset_graph_state(gnnvc_engine *e) {
    e->wide_used = false;
    e->lt_ready = e->lt_tried = false;
    e->lt_used = e->lt_off = false;
    e->lt_unfit_runs = 0;
    e->c4_ready = e->c4_tried = false;
    e->lt_pb.open = e->c4_pb.open = false;
    for (auto &pp : e->prune) pp.forget();
    e->short_from = 0;
    for (int s = 0; s < 4; ++s) { e->c4_stage_off[s] = false; e->c4_unfit_runs[s] = 0; e->fit_used[s] = false; }
    e->fit_pending = false;
    e->fit_calm = e->fit_skip = 0;
    e->c4_range_mode = false;
    e->c4_prepared_stage = -1;
    e->blocked_ready = false;   // the column-blocked index is built on the graph's SECOND forward:
    e->blocked_tried = false;   // it costs about as much as it saves on one, and the reference's
    e->graph_uses = 0;          // driver uses every graph exactly once (src/GNN_VC.cpp:171-192)
    e->plan_build_ms = 0.0;
    e->early_ms = 0.0;
    for (bool &b : e->c4_seeded) b = false;
}

// A host hand-off (upload / staged) of a large graph: the row pointers and weights are on the device before most of the
// column array is.  That is enough to class the graph (find_long reads row pointers only) and to lay out the flat plans;
// their count / regroup passes then run on the second stream for every slice whose entries have arrived, under the copies
// of the rest ("built at hand-off, on the second stream, as the column pieces arrive").  The column ids have
// not been validated at that point — the builders clamp what they index with them, and a hand-off that fails validation at
// the end throws the plans away.  Called with e->rowptr / w / nw holding the new graph's arrays (stream-ordered).
int handoff_early(gnnvc_engine *e, uint32_t n, uint64_t nnz) {
    if (e->early_open || e->early_declined) return GNNVC_OK;
    e->early_declined = true;
    if (!e->opt_handoff || e->multi || e->stages.empty() || n == 0 || !e->aux_stream) return GNNVC_OK;
    if (e->opt_handoff < 2 && nnz < e->opt_handoff_min_nnz) return GNNVC_OK;
    const auto t0 = std::chrono::steady_clock::now();
    e->have_graph = false;
    e->empty_slice = false;
    e->g = GraphDev{n, nnz, e->rowptr.p, e->col.p, e->w.p, e->nw.p};
    reset_graph_state(e);
    {   // The flat builders read col[] and write the plans' entry arrays at offsets they take from the row pointers: those have
        // to be known good BEFORE anything is classed or begun (the column ids are clamped where they index; the full check of
        // the hand-off still runs at its end).  n + 1 words, and find_long waits for the row pointers anyway.
        HIP_TRY(e, e->blk_flag.reserve(1));
        HIP_TRY(e, gnnvc::validate_rowptr(e->g, e->blk_flag.p, e->stream));
        uint32_t bad = 0;
        HIP_TRY(e, hipMemcpyAsync(&bad, e->blk_flag.p, sizeof bad, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        if (bad) return fail(e, GNNVC_ERR_INVALID, "row pointers are not monotone from 0 to nnz");
    }
    int rc = find_long(e);
    if (rc) return rc;
    const bool skewed = e->sorted_wanted || e->n_long > 0;
    if (!skewed) {
        rc = lt_begin(e);
        if (rc) return rc;
        rc = c4_begin(e, 0, 0xFFFFFFFFu);
        if (rc) return rc;
        // (only the flat builders work piece by piece; anything else waits for the commit)
        if (e->lt_pb.open && !gnnvc::lds_table_is_flat(e->lt_pb.slice_rows, e->lt_pb.pm)) e->lt_pb.open = false, e->lt_tried = false;
        if (e->c4_pb.open && !gnnvc::lds_table_is_flat(e->c4_pb.slice_rows, e->c4_pb.pm)) e->c4_pb.open = false, e->c4_tried = false;
    }
    if (!e->ev_piece) HIP_TRY(e, hipEventCreateWithFlags(&e->ev_piece, hipEventDisableTiming));
    e->early_open = true;
    e->early_declined = false;
    e->early_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return GNNVC_OK;
}


}  // namespace gnnvc_eng
