// gnnvc_engine_state.h — the engine object behind include/gnnvc.h's opaque handle and what the translation units of
// libgnnvc_hip.so share about it.  Internal.
//   gnnvc_engine.cpp  the C ABI: model text, graph hand-off entry points, stage selection and launches, forwards
//   gnnvc_plans.cpp   what is built per GRAPH: row classes (long / giant rows, tile order), the LDS-table, compact-table and
//                     column-blocked plans, the pruned adjacency, and when (hand-off / first forwards)
//   gnnvc_multi.cpp   several devices behind one handle (public ABI only)
//   gnnvc_kernels.hip the gfx950 kernels and their launchers
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/gnnvc.h"
#include "gnnvc_kernels.h"
#include "gnnvc_multi.h"

using gnnvc::GraphDev;
using gnnvc::StagePlan;

namespace gnnvc_eng {


enum LayerKind { kLinear = 0, kGraph = 1, kRelu = 2, kSigmoid = 3 };

struct Layer {
    LayerKind kind;
    uint32_t k = 0, m = 0;          // linear: W is k x m
    std::vector<float> W, bias;     // host copies
    size_t w_off = 0, b_off = 0;    // float offsets in the device parameter buffer
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t count) {
        if (count <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t rc = hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T));
        if (rc == hipSuccess) cap = count;
        return rc;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Page-locked host staging (graph hand-off: the copy engine reads it directly, no bounce buffer).
template <class T>
struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t count) {
        if (count <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(count + count / 8, 64);   // head-room: the driver's graphs shrink
        hipError_t rc = hipHostMalloc(reinterpret_cast<void **>(&p), want * sizeof(T), hipHostMallocDefault);
        if (rc == hipSuccess) cap = want;
        return rc;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};


}  // namespace gnnvc_eng

using gnnvc_eng::DevBuf;
using gnnvc_eng::PinBuf;
using gnnvc_eng::Layer;
using namespace gnnvc_eng;

struct gnnvc_engine {
    int device = 0;
    // gnnvc_create_multi: this handle is the FRONT of several devices — an ordinary engine on devices[0] (model, layer-level
    // entry points, staging memory, the assembled scores) whose graph hand-offs and forwards go to `multi` (gnnvc_multi.cpp)
    gnnvc::MultiState *multi = nullptr;
    std::string name;
    std::vector<Layer> layers;
    std::vector<StagePlan> stages;  // non-empty iff fused
    int in_width = 1, out_width = 1;
    int max_width = 1;
    bool ends_in_sigmoid = false;
    float ws = 120.0f;  // graph_layer::WEIGHT_SCALE default (reference include/gnn_inference.hpp:25)

    hipStream_t own_stream = nullptr, stream = nullptr;
    std::vector<hipEvent_t> ev;  // stage boundaries of the last forward
    int ev_count = 0;
    bool ev_stages = false;                  // the last forward recorded an event per stage (forward_timing 2)

    DevBuf<float> params;
    // graph
    GraphDev g;
    bool have_graph = false;
    DevBuf<uint32_t> rowptr, col, w, nw;
    // staged hand-off (gnnvc_graph_staging .. gnnvc_commit_staged_graph)
    PinBuf<uint32_t> pin_rowptr, pin_col, pin_w, pin_nw;
    PinBuf<uint32_t> pin_small;   // host side of small device<->host round trips
    uint32_t staged_n = 0;
    uint64_t staged_nnz = 0, staged_sent = 0;   // columns [0, staged_sent) are already on their way
    bool staging = false;
    // feature buffers
    DevBuf<float> x, h[2], scores, logits;
    DevBuf<float> scratch[2];  // layer-level entry points / unfused path

    // column-blocked plan of the F = 1 stage (built per graph, see gnnvc_kernels.hip)
    int opt_blocked = 1;            // option "blocked_stage0"
    uint32_t opt_block_cols = 0;    // option "block_cols" (0 = default)
    uint32_t opt_blocked_min_n = 1u << 20;  // below this x fits the L2s anyway
    uint32_t opt_compact_min_n = 1u << 18;  // option "compact_min_n": the compact-table plan's own bound (the smaller of the two counts)
    uint64_t opt_compact_min_nnz = 8u << 20;   // ... and its entries bound (default sizes only)
    uint64_t opt_compact_first_entries = 48ull << 20;   // option "compact_first_forward_entries": graphs of this many entries build
                                               // the plan inside their FIRST forward (0 = never; otherwise it is built in the second).
                                               // Metric graph (200 M entries): first forward 10.96 -> 9.18 ms; ER-3M (60 M): 3.03 -> 2.69;
                                               // ER-1M (20 M): 0.92 -> 1.14
    uint32_t opt_plan_chunk_rows = 0;       // != 0: cap on the rows per chunk of the LDS-table / compact-table plans
    bool blocked_ready = false;
    bool blocked_tried = false;     // build attempted for the current graph
    uint32_t graph_uses = 0;        // stage-0 executions on the current graph
    uint32_t blk_count = 0, blk_cols = 0;
    DevBuf<uint32_t> blk_ptr, blk_col, blk_scratch, blk_flag;
    DevBuf<float> blk_acc;
    // LDS-table plan of the F = 1 stage (same timing as the blocked plan: built on the graph's second forward)
    int opt_lds_table = 1;          // 0 = off, 1 = when it applies, 2 = also on skewed graphs
    bool lt_ready = false, lt_tried = false;
    // (round 4) the device's per-forward verdict on the byte table comes back behind whole forwards like the compact table's: an
    // input that is not k / ws leaves the plan's launches empty and — in the skewed layout, whose rows below the giant ones are
    // all the tile kernel's then — makes the stage several times slower than without the plan; one miss there, three on the
    // consecutive-row layout, switch it off for the graph
    uint32_t lt_bits = 8;           // width of the plan's table entries: 8, 10 or 16 bits per vertex (by the graph's largest weight)
    int opt_lt_bits = 0;            // option "lds_table_bits" (tests, A/B): force a width (0 = by the graph)
    bool lt_used = false, lt_off = false;
    uint32_t lt_unfit_runs = 0;
    bool lt_mapped = false;              // skewed graphs: rows dealt to slices (lt_rowmap), blocks of equal mass, rows below lt_plan_thresh
    uint32_t lt_plan_thresh = 0xFFFFFFFFu;
    DevBuf<uint32_t> lt_rowmap, lt_first, lt_bstart;
    uint32_t opt_lds_skewed_rows = 0;       // (0 = by the size of x) option "lds_table_skewed_rows": rows of at least this many entries stay outside the skewed-graph plan
    uint32_t opt_lds_skewed_min_n = 1u << 21;
    int opt_lds_skewed = 1;              // option "lds_table_skewed": 0 = skewed graphs keep the gathering F = 1 kernels
    uint32_t lt_rows = 0, lt_chunks = 0, lt_blocks = 0, lt_steps_total = 0, lt_last_entry = 0;
    uint32_t lt_base = 0, lt_end = 0;       // the plan's row range: the rows this engine holds when it was built
    // A plan being put together (round 3).  The expensive passes — counting and regrouping a slice's entries by column block —
    // need nothing but that slice's rows, so a hand-off runs them piece by piece on the second stream while the rest of the
    // column array is still crossing the bus (flat layouts: consecutive rows, uniform blocks); begin = eligibility, geometry,
    // buffers; advance = count + regroup the slices up to a given one; finish = the step records (they need every slice's
    // counts) and the verdict.
    struct PlanBuild {
        bool open = false, mapped = false;
        uint32_t base = 0, end = 0, slice_rows = 0, slices = 0, chunks = 0, rows = 0, nblocks = 0, bc = 0, slack = 0, done = 0;
        uint32_t plan_rows = 0, passes = 1;
        uint64_t entry_cap = 0, plan_nnz = 0;
        gnnvc::PlanMap pm;
    };
    PlanBuild lt_pb, c4_pb;
    // a host hand-off in progress whose plans are being built while the column array arrives (handoff_early / handoff_progress)
    bool early_open = false, early_declined = false;
    hipEvent_t ev_piece = nullptr;
    double early_ms = 0.0;      // host time the hand-off spent classing the graph and queuing builds before the commit
    uint32_t opt_lt_min_chunks = 128;       // option "lds_table_min_chunks": a short row range is cut into at least this many chunks
    DevBuf<uint8_t> lt_bytes;
    DevBuf<uint32_t> lt_entries, lt_segcnt, lt_stepptr, lt_stepcnt, lt_bad;
    DevBuf<uint4> lt_steps;
    // compact-table plan of the 16-wide stages (built like the LDS-table plan, on the graph's second forward)
    int opt_compact = 1;            // 0 = off, 1 = when it applies, 2 = also on skewed graphs
    bool c4_ready = false, c4_tried = false;
    uint32_t c4_rows = 0, c4_chunks = 0, c4_steps_total = 0, c4_block = 0, c4_last_entry = 0, c4_nblocks = 0;
    DevBuf<uint32_t> c4_entries, c4_segcnt, c4_stepptr, c4_stepcnt, c4_desc;
    uint32_t c4_nslices = 0;
    DevBuf<uint32_t> c4_map_vertex;   // (scratch of layout_skewed_plan: the LDS-table plan's layout on skewed graphs)
    DevBuf<uint32_t> map_coarse;      // uint16 per 256 columns: their block (scratch of the skewed-graph plan builders)
    DevBuf<uint4> c4_map_meta;
    DevBuf<uint4> c4_steps;
    DevBuf<float> c4_table, c4_acc, c4_agg16;
    DevBuf<uint32_t> c4_marks;            // dirty-row slots handed out after each round of the aggregation grid
    std::vector<hipEvent_t> round_ev;     // "round k's sums are done" (main stream -> aux stream)
    int opt_overlap = 1;                  // last stage: dense layers of round k under the sums of round k + 1
    int opt_dense_skip = 1;               // option "dense_skip_zeros" (A/B): the aggregate-only dense kernels take a clean row's <= 11 non-zero
                                          // first-layer terms from its sums and the input's compact table instead of the 32-term chain (k_dense_f16)
    DevBuf<uint32_t> c4_dirty;
    uint32_t c4_dirty_cap = 0;
    DevBuf<unsigned long long> c4_counts, c4_emit_counts;
    uint32_t c4_base = 0, c4_end = 0;   // the plan's row range: the whole graph, or the rows a multi-GPU rank computes
    bool c4_range_mode = false;         // a driver asked for a range plan (gnnvc_stage_input_ready): no whole-graph plan any more
    int c4_prepared_stage = -1;         // gnnvc_stage_input_ready: the table holds this stage's input ...
    const float *c4_prepared_in = nullptr;   // ... as found at this address
    int c4_fused_for = -1;          // stage whose input statistics (and table) the previous stage kernel of this forward produced
    // pruned adjacency of the 16-wide stages (kernels: k_prune_*), one per consumer stage: built from the input the stage
    // sees the second time the graph is scored; every later call proves on the device that its input still fits
    struct PrunePlan {
        bool tried = false, ready = false, deferred = false;
        // round 4: built at HAND-OFF from the set the graph alone predicts (k_predict_zero_f1: the reference driver's input is
        // x = W / ws, so the first 16-wide stage's zero rows follow from the weights) — proven per call like any other set;
        // verified: a forward has run with it and the host has seen that its check passed (if not, the plan is dropped and
        // rebuilt from the input the stage really sees)
        bool predicted = false, verified = false;
        void forget() { tried = ready = deferred = predicted = verified = false; }
        bool from_prev = false;             // built from the previous stage's kept entries (its set is contained in this one)
        uint64_t kept = 0;                  // entries left
        uint64_t members = 0;               // vertices in the set
        DevBuf<uint32_t> prp, pcol, heavy;
        DevBuf<uint32_t> svertex;            // skewed graphs: the engine's rows below the long-row threshold BY ENTRIES LEFT, heaviest first
        DevBuf<uint4> smeta;                 // ... with their pruned ranges
        uint32_t sn = 0;
        bool slist = false;
        uint32_t eff_thresh = 0xFFFFFFFFu;   // entries left from which a row goes to the long-row kernel
    };
    PrunePlan prune[4];
    DevBuf<uint32_t> prune_flags, prune_scratch, prune_off;   // (off / mask: per chunk of 64 entries, while a plan is built)
    DevBuf<unsigned long long> prune_mask;   // flags: [stage] = this call's verdict (0 = the pruned adjacency applies), [3] = observe
    int opt_prune = 1;               // option "prune_zero_rows": 1 = the rows found all zero when the plan is built (or predicted at hand-off), 0 = off
    uint64_t opt_prune_heavy_entries = 16u << 20;   // option "prune_heavy_entries": from this many entries left, rows up to the sorted threshold stay with the tile kernel
    uint64_t opt_prune_early_nnz = 64u << 20;   // option "prune_early_entries": skewed graphs with at least this many entries build the plan in their first forward (0 = never)
    // option "prune_predict": 1 = large skewed graphs (the ones the filtered gather is offered to) get the first 16-wide stage's
    // pruned adjacency when they are HANDED OVER, from the predicted set — a graph scored once (the reference's driver,
    // src/GNN_VC.cpp:171-192) then runs its first forward on it, and the next stage borrows it until it has its own
    int opt_prune_predict = 1;
    uint64_t opt_predict_min_nnz = 48u << 20;   // option "prune_predict_min_entries"
    int opt_prune_eff = 1;           // option "prune_class_by_entries_left" (A/B): 0 = rows keep the class their degree gives them
    int opt_prune_giant = 1;         // option "prune_giant_rows" (A/B): 0 = the giant rows keep their full streams
    uint64_t opt_prune_min_nnz = 1u << 20;   // option "prune_min_entries": smaller graphs are not worth a plan
    uint32_t opt_prune_min_drop = 15;   // option "prune_min_drop_percent": build only if at least this share of the entries goes
    // Filtered gather: while a skewed graph's 16-wide stage has no pruned adjacency (the graph's first forward: the reference's
    // driver never comes back for a second), its kernels look every entry's target up in the bitmap of THIS input's all-zero
    // rows, written just before them, and fetch the pad row instead (GraphDev::zero_bits; nothing to build, nothing to prove).
    int opt_giant_gather_first = -1; // option "giant_gather_first": the giant rows' gather on the main queue ahead of the tile kernel (1), on the side queue with the rest of their chain (0), -1 = by the graph (launch_side_rows)
    int opt_long_on_main = -1;       // option "long_rows_on_main": -1 = by the graph (launch_side_rows), 0 = beside the giant rows on the side queue, 1 = ahead of the tile kernel
    int side_join = 0;               // what the stage at hand joins on: 0 nothing, 1 the long rows' queue, 2 the giant rows' queue
    int opt_filter = 1;              // option "filter_zero_rows" (A/B): 0 = plain gathers until the plan is there
    // which graphs (measured, scratch/experiments/first_ab2.sh + fuzz_large.py: first forward with / without): R-MAT from ~48 M
    // entries on gains 0.5 - 1.6 ms (R-MAT-22 4.61 -> 3.98, R-MAT-24 19.3 -> 17.7, scale 21 x 16: 2.84 -> 2.30); smaller graphs
    // lose 0.05 - 0.25 ms to the marks and look-ups, power-law graphs (41 - 58 % of the entries point to zero rows) 0.1 ms, nearly
    // uniform graphs with a few hubs (1 - 20 %) 0.2 ms — those have 1 - 3 % of their entries in long rows, R-MAT 35 - 58 %
    uint64_t opt_filter_min_nnz = 48u << 20;   // option "filter_min_entries"
    uint32_t opt_filter_min_long_pct = 25;     // option "filter_min_long_percent": only graphs whose long rows hold this share of the entries
    uint32_t opt_filter_min_pct = 50;   // option "filter_min_percent": the share of the entries that has to point into the set (decided on the device)
    int opt_filter_keep = 1;         // option "filter_keep_lists" (A/B): 0 = every filtered stage walks the whole adjacency
    DevBuf<uint32_t> filter_bits[4];
    DevBuf<unsigned long long> filter_info;   // [4 * stage]: {degrees of the set's vertices, their number, verdict on an earlier stage's lists}
    bool filtered[4] = {false, false, false, false};   // the last call of the stage was offered the bitmap
    bool borrowed[4] = {false, false, false, false};   // the last call of the stage ran on the PREVIOUS stage's predicted plan (gather_view)
    // ... and the targets a filtered stage found outside its set, left per row in prune[stage].pcol / .prp (the buffers of the
    // plan that is not built yet), are the adjacency of the NEXT 16-wide stage of the same forward when the device finds that
    // stage's input to keep the set all zero (GraphDev::keep_col / short_col): short_from = the stage that left them, 0 = none
    int short_from = 0;
    uint32_t short_min = 0, short_max = 0;   // the degrees [min, max) of the rows that have a list
    bool short_used[4] = {false, false, false, false};   // the last call of the stage was offered an earlier stage's lists
    // Does the device keep finding a stage's input unfit for the plan (more than its tables' columns live: low-degree graphs)?
    // Whole forwards copy the verdicts out behind themselves; three misses in a row switch the plan off for that stage of this
    // graph — its counting, choosing and empty launches cost up to 17 % of a forward that then gathers anyway.
    PinBuf<uint32_t> fit_pin;
    hipEvent_t ev_fit = nullptr;
    bool fit_pending = false, fit_used[4] = {false, false, false, false}, c4_stage_off[4] = {false, false, false, false};
    uint32_t c4_unfit_runs[4] = {0, 0, 0, 0};
    int c4_last_desc = 0;           // word offset in c4_desc of the plan's last launch (tests / tools)
    static constexpr int kDescWords = 16;   // per consumer stage (see k_c4_choose); the build flag follows the last stage's

    // Table tiles (round 4; k_stage_t4): graphs too small for the compact-table plan and too large for their feature rows to sit in
    // an L2 (50 - 400 K vertices: BASELINE configs[1]) gather the 16-wide stages' neighbours from the 16-byte compact table of the
    // input — written by the kernel that produces the input, for the columns the previous forward chose — inside whole forwards.
    int opt_t4 = 1;                          // option "table_tiles"
    uint32_t opt_t4_min_n = 49152;           // option "table_tiles_min_n": below, the 64-byte rows fit an XCD's L2 anyway
    uint64_t opt_t4_max_bytes = 6ull << 20;  // option "table_tiles_max_bytes": the table has to (mostly) sit in a 4 MiB L2
    bool t4_ok = false;                      // the current graph qualifies
    bool t4_used = false;                    // the forward whose verdicts are on their way ran with the table tiles offered
    bool t4_fit_seen[4] = {false, false, false, false};   // the stage's table fit in the last forward whose verdict has arrived
    int opt_t4_solo = 1;                     // option "table_tiles_solo" (A/B): 0 = always launch the gathering kernel behind the tiles
    uint32_t t4_unfit_runs = 0;              // forwards in a row whose first 16-wide stage left the launch to the gathering kernel
    int opt_timing = 0;                      // option "forward_timing": 0 = a forward records no events (gnnvc_last_forward_ms is refused), 1 = its first and last, 2 = one per stage too
    uint32_t *fit_dev = nullptr;             // fit_pin as the device sees it (the verdict words are WRITTEN there by one small kernel)
    uint32_t fit_calm = 0;                   // verdicts in a row that changed nothing: from four on, only every eighth forward asks
    uint32_t fit_skip = 0;
    int opt_poison = 0;                      // option "poison_features" (tests, fuzz): a whole forward starts by filling the engine's feature buffers with NaN bit patterns — a row no kernel writes shows in the result instead of hiding behind an earlier forward's values
    uint32_t opt_verdict_period = 8;         // option "verdict_period": calm verdicts are asked for every this-many forwards (1 = always)
    int opt_wide = 1;                        // option "wide_tiles": graphs of up to "wide_tiles_max_n" vertices run their plain stages a workgroup per tile
    uint32_t opt_wide_max_n = 49152;         // the F = 1 stage ("wide_tiles_max_n": where the table tiles start — feeding them from wide tiles was measured slower) ...
    uint32_t opt_wide_max_n16 = 131072;      // ... and the 16-wide stages ("wide_tiles_max_n_f16") up to these many vertices (measured: small_sizes.py)
    bool wide_used = false;
    bool t4_now = false;                     // the forward at hand runs with the table tiles offered
    bool t4_choice_live = false;             // a forward with table tiles has run on this engine: the descriptors hold a choice (kept across graphs)
    uint32_t t4_parity = 0;                  // which of a stage's two descriptors the producers read in the forward at hand
    DevBuf<float> t4_table[2];               // [0]: the table of stage 1's input, [1]: of stage 2's (a stage gathers from one while emitting the other)
    DevBuf<uint32_t> t4_desc;                // [stage - 1][parity][16 words]
    DevBuf<unsigned long long> t4_counts[2]; // the producers' per-column counters: [stage - 1], two sets of kEmitCounters each (by parity, like the descriptors)
    unsigned long long *t4_counts_of(int stage, uint32_t parity) { return t4_counts[stage - 1].p + (size_t)parity * gnnvc::kEmitCounters; }
    uint32_t *t4_desc_of(int stage, uint32_t parity) { return t4_desc.p + ((size_t)(stage - 1) * 2 + parity) * 16; }

    // option "mfma_dense": dense layers on the matrix cores (bit-identical to the VALU path).
    // 0 = VALU everywhere, 1 = MFMA everywhere, 2 = MFMA in the F = 16 stages only (default:
    // the F = 1 stage's first layer has K = 5 and stays on the VALU, and sending its 32
    // activations through LDS just to reach the matrix layout costs more than it saves)
    int opt_mfma = 2;

    // degree-sorted tile order (16-wide stages, skewed graphs); built per row range on demand
    int opt_sorted = -1;               // option "sorted_tiles": -1 auto (by measured waste), 0 off, 1 on
    uint64_t opt_sorted_min_nnz = 4ull << 20;   // auto mode leaves smaller graphs on natural tiles
    uint32_t opt_sorted_long_thresh = 1024;   // long-row threshold of the 16-wide stages when their tiles are sorted
    uint32_t thresh_f16 = 0xFFFFFFFFu;        // rows >= this go to k_long_f16 (>= long_thresh, the list's threshold)
    bool interleave = false;           // deal natural tiles round-robin (work is unevenly spread over the row range)
    bool sorted_wanted = false;        // decided per graph from the measured tile waste
    // A few row ranges are cached: a vertex-partitioned caller alternates between its own rows and (for a replicated
    // stage) the whole graph, or between the pieces of a pipelined stage — each range is sorted once per graph.
    struct SortedRange {
        bool valid = false, use = false;
        uint32_t lo = 0, hi = 0, n = 0;
        uint64_t stamp = 0;
        DevBuf<uint32_t> vertex;
        DevBuf<uint4> meta;
    };
    static constexpr int kSortedRanges = 6;
    SortedRange srt[kSortedRanges];
    int srt_cur = -1;                  // the entry ensure_sorted selected for the call in progress
    uint64_t srt_clock = 0;
    double srt_waste = 0.0, srt_tail = 0.0;
    DevBuf<uint32_t> srt_hist;
    DevBuf<unsigned long long> srt_sum;

    // long rows (degree >= long_thresh): one workgroup each, on aux_stream beside the tile kernel
    uint32_t opt_long_thresh = 512;   // option "long_row_threshold" (0 = off)
    bool opt_long_auto = true;        // no explicit threshold: 256 where few rows are that long, else 512
    uint32_t long_thresh = 0xFFFFFFFFu, n_long = 0;
    DevBuf<uint32_t> long_list, long_count;
    // what classify_hand_off learned about a graph in its one round trip, for the find_long that follows it
    struct PreClass {
        bool valid = false, cuts = false, waste = false, longs = false;
        uint32_t lo = 0, hi = 0, waste_thresh = 0, heavy_from = 0, long_thresh = 0;
        uint32_t cut[9] = {0};
        unsigned long long sums[2] = {0, 0};
        uint32_t found[4] = {0, 0, 0, 0};
    } pre;
    bool pre_armed = false;                  // set by the hand-off that ran classify_hand_off, right before ITS find_long (which clears it): no other find_long — the early hand-off's, a later graph's after a failed attach — may take what was learned about another candidate
    DevBuf<uint32_t> cls_dev;                // classify_graph's 16 words of device scratch
    PinBuf<uint32_t> cls_pin;                // ... and its 24 result words
    uint32_t *cls_pin_dev = nullptr;
    uint64_t long_entries = 0;       // entries of the listed rows
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // (the long rows' and the giant rows' stream handles: the side queue again — ensure_side_streams — with join events of their own)
    hipStream_t long_stream = nullptr;
    hipEvent_t ev_long = nullptr;
    int side_probes = 0;             // streams tried until one ran beside the main stream (info "side_queue_probes")
    bool side_beside = false;        // ... and whether one did (info "side_queue_runs_beside")
    hipStream_t giant_stream = nullptr;  // giant rows: three dependent launches, the side work's long pole -> a high-priority stream of its own
    hipEvent_t ev_giant = nullptr;
    // giant rows (degree >= giant_thresh, a subset of the long rows): CSR-order sums evaluated in parallel (exact_sum.h)
    double plan_build_ms = 0.0;          // host wall time spent building per-graph plans for the current graph (they end in stream syncs)
    int opt_ktrace = 0;                  // option "kernel_trace": HIP events around every main-stream kernel of a forward
    gnnvc::KernelTraceSink ktrace;
    // the next graph derived from the resident one (gnnvc_derive_graph_begin / _commit)
    DevBuf<uint32_t> rowptr2, col2, der_old_row, der_new_of, der_tail, der_tailptr, der_tailcols;
    DevBuf<unsigned long long> hash_buf;
    std::vector<uint32_t> der_tail_host;
    uint32_t der_n_new = 0;
    uint64_t der_nnz_new = 0, der_tail_total = 0;
    bool der_open = false;
    bool empty_slice = false;            // gnnvc_attach_graph_slice with no rows: every stage call is a no-op
    uint32_t opt_giant_thresh = 16384;   // option "giant_row_threshold" (0 = off: k_long_* take every long row)
    int opt_side_streams = 1;            // option "side_streams": 0 = long / giant rows on the main stream, one after the other (profiling)
    uint32_t giant_thresh = 0xFFFFFFFFu, n_giant = 0, giant_blocks = 0;
    uint32_t opt_giant_f16 = 65536;     // option "giant_row_threshold_f16": the 16-wide stages send only rows from this degree on the giant way
    bool giant_f16_auto = true, giant_walk_bound = false;   // (walk_bound: the longest stream's walk is what a stage waits for, find_giant)
    uint32_t giant_f16() const {
        if (!n_giant) return 0xFFFFFFFFu;
        // by the graph: a 65 536-entry row's add chain in k_long_f16 is ~0.26 ms — lost in the stages of a graph with 64 M entries
        // and more (R-MAT-22 2.98 -> 2.88 ms, R-MAT-24 12.6 -> 11.6 ms), what the stages of a smaller one would wait for (R-MAT-20
        // 0.99 -> 1.05 ms, power-law 1.03 -> 1.30 ms)
        if (giant_f16_auto && g.nnz < (64ull << 20)) return giant_thresh;
        return std::max(giant_thresh, opt_giant_f16);
    }   // (16 streams per row: three times a
                                        // long row's traffic — worth it only for the rows whose add chain a stage would wait for)
    uint64_t giant_entries = 0;
    DevBuf<uint4> gi_meta;
    DevBuf<unsigned long long> gi_off;
    DevBuf<float> gi_slab, gi_agg, gi_segsum;   // (segsum / segmap: one stream on several waves, see k_giant_segmap)
    DevBuf<uint4> gi_segmap;
    uint32_t gi_maxseg = 0;
    int opt_giant_segments = -1;  // option "giant_segments": 1 = a stream on several waves, 0 = one wave walks it, -1 = by the graph (default)
    // Plans at hand-off (round 3).  The reference's driver scores every graph exactly once (src/GNN_VC.cpp:171-192), so a plan
    // built inside a graph's second forward never serves it.  What depends on the graph alone is built when the graph is handed
    // over (upload / staged commit / attach): 1 (default) = the plans one use repays (degree-uniform graphs of at least
    // opt_handoff_min_nnz entries: LDS table + compact table; every graph: the tile order and every buffer a forward would
    // otherwise allocate), 2 = every plan whatever its cost (callers who score a graph many times, or hide the build under a
    // copy), 0 = as in round 2 (inside the first two forwards).
    int opt_handoff = 1;
    uint64_t opt_handoff_min_nnz = 24ull << 20;   // (the builds cost ~20 ps per entry and plan, a first forward saves ~40: from ~20 Mi entries on one use repays them)
    // First use of the compact-table plan on a graph: a pilot over the first opt_pilot_rows rows of the producing stage picks
    // the consumer's table columns, so the producer can write the table on its way (see launch_main)
    uint32_t opt_pilot_rows = 65536;
    bool c4_seeded[4] = {false, false, false, false};
    PinBuf<uint32_t> pin_info;   // small device -> host results that outlive the call that asked for them (never reallocated)
    DevBuf<uint32_t> dev_info;
    double handoff_build_ms = 0.0;

    std::string err;
};

namespace gnnvc_eng {

inline int fail(gnnvc_engine *e, int code, const char *fmt, ...) {
    if (e) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        e->err = buf;
    }
    return code;
}

#define HIP_TRY(e, call)                                                                   \
    do {                                                                                   \
        hipError_t rc_ = (call);                                                           \
        if (rc_ != hipSuccess)                                                             \
            return fail((e), rc_ == hipErrorOutOfMemory ? GNNVC_ERR_NOMEM : GNNVC_ERR_DEVICE, \
                        "%s: %s", #call, hipGetErrorString(rc_));                          \
    } while (0)

// entry points that read ONE device's resident graph have no meaning on the front of several devices
#define NOT_ON_MULTI(e, what)                                                                                          \
    do {                                                                                                               \
        if ((e)->multi) return fail((e), GNNVC_ERR_UNSUPPORTED, what " is not available on a multi-device handle (gnnvc_create_multi)"); \
    } while (0)

inline int hip_rc(gnnvc_engine *e, hipError_t rc) {
    if (rc == hipSuccess) return GNNVC_OK;
    return fail(e, rc == hipErrorOutOfMemory ? GNNVC_ERR_NOMEM : GNNVC_ERR_DEVICE, "%s", hipGetErrorString(rc));
}

inline int use_device(gnnvc_engine *e) {
    HIP_TRY(e, hipSetDevice(e->device));
    return GNNVC_OK;
}

// ---- gnnvc_plans.cpp: what is built per graph -------------------------------------------------------------------------
int find_long(gnnvc_engine *e);                              // row classes of a new graph (long / giant rows, tile waste)
int ensure_sorted(gnnvc_engine *e, uint32_t lo, uint32_t hi);
int build_blocked(gnnvc_engine *e);
int build_lds_table(gnnvc_engine *e);
int build_compact(gnnvc_engine *e, uint32_t base = 0, uint32_t end = 0xFFFFFFFFu);
// a plan put together piece by piece (gnnvc_engine::PlanBuild): begin = eligibility, geometry, buffers; advance = count and
// regroup the slices up to a given one on a given stream; finish = the step records and the verdict
int lt_begin(gnnvc_engine *e);
int lt_advance(gnnvc_engine *e, uint32_t upto, hipStream_t stream);
int lt_finish(gnnvc_engine *e);
int c4_begin(gnnvc_engine *e, uint32_t base, uint32_t end);
int c4_advance(gnnvc_engine *e, uint32_t upto, hipStream_t stream);
int c4_finish(gnnvc_engine *e);
gnnvc::CompactPlan compact_plan(const gnnvc_engine *e);
int ensure_side_streams(gnnvc_engine *e);
int reprobe_side_streams(gnnvc_engine *e);
int ensure_round_events(gnnvc_engine *e, size_t count);
int ensure_events(gnnvc_engine *e, size_t count);
int classify_hand_off(gnnvc_engine *e, const GraphDev &cand, uint32_t &bad);
int gather_view(gnnvc_engine *e, int stage, uint32_t lo, uint32_t hi, const float *in, bool gathering, bool sorted_tiles,
                GraphDev &gv, gnnvc::SortedOrder &so_p, bool matrix_cores = true, uint32_t long_from = 0xFFFFFFFFu);
int reserve_features(gnnvc_engine *e, uint32_t n);
int reserve_multi_front(gnnvc_engine *e, uint32_t n);
int prepare_plans(gnnvc_engine *e);                          // everything a forward needs that depends on the graph alone
int prepare_table_tiles(gnnvc_engine *e);                    // does the graph qualify for k_stage_t4, and its buffers
void reset_graph_state(gnnvc_engine *e);
int handoff_early(gnnvc_engine *e, uint32_t n, uint64_t nnz);
// host wall time of a plan build, added to plan_build_ms (what was queued before is drained first: not the plan's cost)
template <class F>
int timed_build(gnnvc_engine *e, F &&f, bool may_build = true) {
    // (may_build false: the caller can already see that f will leave at its first test — a graph too small for the plan — and the
    // wait that keeps earlier work out of the build's time would only stall the forward: five of them made a small graph's SECOND
    // forward 60 us where its third takes 34)
    if (may_build) (void)hipStreamSynchronize(e->stream);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = f();
    e->plan_build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

}  // namespace gnnvc_eng
