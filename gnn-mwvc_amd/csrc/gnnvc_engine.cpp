// gnnvc_engine.cpp — host side of libgnnvc_hip.so: the C ABI of include/gnnvc.h.
//
// Holds the parsed model (the reference's text format, reference
// src/gnn_inference.cpp:120-139), the device copy of the graph view, the
// feature buffers, and sequences the stages like model::predict does
// (reference src/gnn_inference.cpp:67-81).  All arithmetic happens in the HIP
// kernels of gnnvc_kernels.hip; there is no CPU fallback here.
#include "gnnvc_engine_state.h"

namespace {

// ---- model text ------------------------------------------------------------
struct Cursor {
    const char *p, *end;
    bool token(std::string &out) {
        while (p < end && isspace((unsigned char)*p)) ++p;
        if (p >= end) return false;
        const char *s = p;
        while (p < end && !isspace((unsigned char)*p)) ++p;
        out.assign(s, p);
        return true;
    }
    bool size(uint32_t &v) {
        std::string t;
        if (!token(t)) return false;
        char *q = nullptr;
        unsigned long x = strtoul(t.c_str(), &q, 10);
        if (!q || *q) return false;
        v = (uint32_t)x;
        return true;
    }
    // matrix text: "<h> <w>" then h*w numbers (reference src/matrix.cpp:97-104)
    bool matrix(uint32_t &h, uint32_t &w, std::vector<float> &d) {
        if (!size(h) || !size(w)) return false;
        if ((uint64_t)h * w > (1u << 26)) return false;
        d.resize((size_t)h * w);
        std::string t;
        for (auto &v : d) {
            if (!token(t)) return false;
            v = strtof(t.c_str(), nullptr);  // same rounding as istream >> float
        }
        return true;
    }
};

int parse_model(gnnvc_engine *e, const char *text, size_t len) {
    Cursor c{text, text + len};
    std::string t;
    uint32_t n = 0;
    if (!c.token(e->name) || !c.size(n) || !c.token(t))
        return fail(e, GNNVC_ERR_INVALID, "model header: expected '<name> <n> Layers'");
    if (n > 4096) return fail(e, GNNVC_ERR_INVALID, "implausible layer count %u", n);
    for (uint32_t i = 0; i < n; ++i) {
        if (!c.token(t)) return fail(e, GNNVC_ERR_INVALID, "model text ends after %u of %u layers", i, n);
        Layer l;
        if (t == "Linear_Layer") {
            uint32_t bh = 0, bw = 0;
            l.kind = kLinear;
            if (!c.token(t) || !c.matrix(l.k, l.m, l.W))
                return fail(e, GNNVC_ERR_INVALID, "layer %u: malformed weight matrix", i);
            if (!c.token(t) || !c.matrix(bh, bw, l.bias) || bh != 1 || bw != l.m)
                return fail(e, GNNVC_ERR_INVALID, "layer %u: malformed bias", i);
        } else if (t == "Graph_Layer") {
            l.kind = kGraph;
        } else if (t == "ReLU_Activation") {
            l.kind = kRelu;
        } else if (t == "Sigmoid_Activation") {
            l.kind = kSigmoid;
        } else {
            continue;  // the reference's if-chain ignores unknown records
        }
        e->layers.push_back(std::move(l));
    }
    // zero layers is allowed: an engine used only for the layer-level entry points
    return GNNVC_OK;
}

// Widths through the network; decides fused vs layer-by-layer.
int plan_model(gnnvc_engine *e) {
    // input width: a leading graph layer accepts any width; a leading linear fixes it.
    int wd = 1;
    for (const auto &l : e->layers) {
        if (l.kind == kLinear) { wd = (int)l.k; break; }
        if (l.kind == kGraph) { wd = 1; break; }
    }
    // walk back from the first linear through graph layers: k = 2*w + 3
    {
        int first_lin = -1, graphs_before = 0;
        for (size_t i = 0; i < e->layers.size(); ++i) {
            if (e->layers[i].kind == kLinear) { first_lin = (int)i; break; }
            if (e->layers[i].kind == kGraph) ++graphs_before;
        }
        if (first_lin >= 0) {
            int k = (int)e->layers[first_lin].k;
            for (int gq = 0; gq < graphs_before; ++gq) {
                if ((k - 3) % 2 != 0 || k < 5) return fail(e, GNNVC_ERR_INVALID, "inconsistent layer widths");
                k = (k - 3) / 2;
            }
            wd = k;
        }
    }
    e->in_width = wd;
    e->max_width = wd;
    size_t off = 0;
    for (auto &l : e->layers) {
        if (l.kind == kLinear) {
            if ((int)l.k != wd) return fail(e, GNNVC_ERR_INVALID, "linear layer expects %u inputs, gets %d", l.k, wd);
            wd = (int)l.m;
            l.w_off = off; off += l.W.size();
            l.b_off = off; off += l.bias.size();
        } else if (l.kind == kGraph) {
            wd = 2 * wd + 3;
        }
        e->max_width = std::max(e->max_width, wd);
    }
    e->out_width = wd;
    e->ends_in_sigmoid = !e->layers.empty() && e->layers.back().kind == kSigmoid;

    // fused plan: (Graph, Linear, ReLU, Linear, ReLU, Linear, ReLU|Sigmoid)+
    std::vector<StagePlan> st;
    size_t i = 0;
    int f = e->in_width;
    bool ok = !e->layers.empty() && e->layers.size() % 7 == 0;
    while (ok && i < e->layers.size()) {
        const Layer *L = &e->layers[i];
        ok = L[0].kind == kGraph && L[1].kind == kLinear && L[2].kind == kRelu &&
             L[3].kind == kLinear && L[4].kind == kRelu && L[5].kind == kLinear &&
             (L[6].kind == kRelu || L[6].kind == kSigmoid);
        if (!ok) break;
        StagePlan sp;
        sp.f = f;
        sp.n1 = (int)L[1].m; sp.n2 = (int)L[3].m; sp.n3 = (int)L[5].m;
        sp.sigmoid_last = L[6].kind == kSigmoid;
        sp.param_offset = L[1].w_off;
        // the parameter buffer is laid out in layer order, so W1 b1 W2 b2 W3 b3 are contiguous
        sp.variant = gnnvc::stage_variant(sp.f, sp.n1, sp.n2, sp.n3, sp.sigmoid_last);
        const bool last = (i + 7 == e->layers.size());
        ok = sp.variant >= 0 && (int)L[1].k == 2 * f + 3 && (sp.sigmoid_last ? last : true) &&
             (last || sp.n3 == 16);
        st.push_back(sp);
        f = sp.n3;
        i += 7;
    }
    if (ok && !st.empty() && st.back().sigmoid_last) e->stages = std::move(st);
    return GNNVC_OK;
}

int upload_params(gnnvc_engine *e) {
    std::vector<float> flat;
    for (const auto &l : e->layers)
        if (l.kind == kLinear) {
            flat.insert(flat.end(), l.W.begin(), l.W.end());
            flat.insert(flat.end(), l.bias.begin(), l.bias.end());
        }
    HIP_TRY(e, e->params.reserve(flat.size() + 64));
    if (!flat.empty())
        HIP_TRY(e, hipMemcpy(e->params.p, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice));
    return GNNVC_OK;
}

// ---------------------------------------------------------------- one stage call = plan selection, then launches
//
// choose_stage  decides WHAT a call runs: which of the per-graph plans serves the neighbour sums (building a plan the
//               first time it is due), whether the previous stage kernel of this forward already left the column
//               statistics, whether this stage's kernel produces them for the next one, sorted or natural tiles.
//               It launches nothing.
// launch_side_rows / launch_main  put the chosen kernels on the streams.  A new plan variant adds an enumerator, its
//               eligibility rule in choose_stage and its launches in launch_main.
struct StageChoice {
    enum Sums {
        kGather,            // the tile kernel gathers full rows itself (always correct; the fallback of every plan)
        kLdsTable,          // F = 1: byte table in LDS (k_lt_agg), the tile kernel only runs the dense layers
        kBlocked,           // F = 1: column-blocked partial sums (k_blk_accumulate)
        kCompactPrepared,   // F = 16: compact table written by gnnvc_stage_input_ready for exactly this input
        kCompactWhole,      // F = 16: compact table over the whole graph, decided per forward on the device
        kTableTiles         // F = 16, mid-size graphs: the tile kernel gathers the input's L2-resident compact table (k_stage_t4), or — decided
                            // on the device — leaves the launch to the gathering kernel behind it
    } sums = kGather;
    bool mfma = false;          // dense layers of the gathering kernel on the matrix cores
    bool fused_counts = false;  // kCompactWhole: the producing stage kernel of this forward left counts (and perhaps the table)
    bool rounds = false;        // kCompactWhole, last stage: sums one round at a time, dense kernel of round k under round k + 1
    bool emit = false;          // this stage's VALU epilogue counts / compacts its output rows for stage `stage + 1`
    bool emit_t4 = false;       // ... for the table tiles of stage `stage + 1` (k_stage_t4)
    uint32_t long_thresh = 0xFFFFFFFFu;   // rows of at least this degree belong to the side streams
    gnnvc::SortedOrder sorted;  // n != 0: tiles from the degree-sorted list of this row range
};

int choose_stage(gnnvc_engine *e, int stage, uint32_t lo, uint32_t hi, const float *in, const float *out, bool in_forward,
                 StageChoice &c) {
    const bool longs = e->n_long > 0;
    const gnnvc::StagePlan &sp = e->stages[stage];
    // An announcement (gnnvc_stage_input_ready) covers the calls for ITS stage that follow it.  A call for any other
    // stage means the caller has moved on — the next forward has begun, or the announced buffer is about to be
    // rewritten — and a call that writes into the announced buffer ends it too: the table must never outlive its input.
    if (e->c4_prepared_stage != -1 && (stage != e->c4_prepared_stage || out == e->c4_prepared_in)) e->c4_prepared_stage = -1;
    // Producer side of the compact-table plan: inside a whole forward (engine-owned feature buffers nobody else
    // writes) the stage kernel that produces the next 16-wide stage's input also counts its non-zeros and writes
    // its compact rows, so that stage can skip its two passes over the input.  Only the VALU variants emit.
    const int fused_in = in_forward ? e->c4_fused_for : -1;   // is THIS stage's input covered by the previous kernel?
    e->c4_fused_for = -1;
    const bool may_emit = in_forward && e->c4_ready && !e->c4_range_mode && e->c4_base == 0 && e->c4_end == e->g.n && !longs &&
                          (size_t)stage + 1 < e->stages.size() && e->stages[stage + 1].f == 16 && lo == 0 && hi == e->g.n &&
                          e->opt_mfma != 1 && !e->c4_stage_off[stage + 1];
    c.long_thresh = (sp.f == 16) ? e->thresh_f16 : e->long_thresh;
    c.mfma = e->opt_mfma == 1 || (e->opt_mfma == 2 && sp.f == 16);
    const bool t4 = in_forward && e->t4_now && lo == 0 && hi == e->g.n;   // table tiles: whole forwards on a graph that qualifies
    c.emit_t4 = t4 && (size_t)stage + 1 < e->stages.size();
    if (stage == 0) {
        // The LDS-table plan works in chunks of ~19.5 K rows, one workgroup each: a call that covers fewer than
        // three quarters of a GPU's worth of chunks (the pieces of a pipelined multi-GPU run) would leave most CUs
        // idle for the time one chunk takes — such calls use the column-blocked plan instead.
        if (e->graph_uses >= 1 && !e->lt_tried) {
            int rc = build_lds_table(e);
            if (rc) return rc;
        }
        // (a skewed graph's plan sums all of its rows at once: whole-graph calls only)
        // (consecutive-row layout: any call that covers at least three quarters of the chunks a full GPU takes — or of the plan's
        // own, where the plan is a rank's slice of fewer)
        const uint32_t lt_need = std::min(192u, e->lt_chunks - e->lt_chunks / 4u);
        const bool lt_fits = e->lt_ready && !e->lt_off && hi > lo &&
                             (e->lt_mapped ? (lo == 0 && hi == e->g.n)
                                           : (lo >= e->lt_base && hi <= e->lt_end &&
                                              ((hi - 1 - e->lt_base) / e->lt_rows - (lo - e->lt_base) / e->lt_rows + 1) >= lt_need));
        if (e->graph_uses >= 1 && !lt_fits && !e->blocked_tried) {
            int rc = build_blocked(e);
            if (rc) return rc;
        }
        ++e->graph_uses;
        c.sums = lt_fits ? StageChoice::kLdsTable : (e->blocked_ready ? StageChoice::kBlocked : StageChoice::kGather);
        if (lt_fits && in_forward) e->lt_used = true;
        if (lt_fits && e->lt_mapped) c.long_thresh = e->lt_plan_thresh;   // the plan holds every row below the giant ones
        c.emit = may_emit;
        if (c.sums != StageChoice::kGather) return GNNVC_OK;   // (those two bring their own tile order)
    } else if (sp.f == 16 && t4) {
        c.sums = StageChoice::kTableTiles;
    } else if (sp.f == 16) {
        // (the second forward on a graph builds the plan; a graph whose plain 16-wide stages cost well above the build — the
        // option's bound — builds it in its first, which is all a score-once caller ever runs)
        const bool first_too = e->opt_compact_first_entries && e->g.nnz >= e->opt_compact_first_entries && lo == 0 && hi == e->g.n && in_forward;
        if (!e->c4_range_mode && (e->graph_uses >= 2 || first_too) && !e->c4_tried) {
            int rc = build_compact(e);
            if (rc) return rc;
        }
        const bool whole_plan = e->c4_ready && e->c4_base == 0 && e->c4_end == e->g.n && !e->c4_stage_off[stage];
        const bool prepared = e->c4_ready && e->c4_prepared_stage == stage && e->c4_prepared_in == in &&
                              lo >= e->c4_base && hi <= e->c4_end && hi > lo;
        if (prepared && !longs) {
            // gnnvc_stage_input_ready wrote the table for this input: any call that fills at least half the
            // GPU with chunks takes the sums from it (smaller ones would leave most CUs idle for a chunk's time)
            const uint32_t nchunks = (hi - 1 - e->c4_base) / e->c4_rows - (lo - e->c4_base) / e->c4_rows + 1;
            if (nchunks >= 128u) c.sums = StageChoice::kCompactPrepared;
        } else if (!e->c4_range_mode && whole_plan && !longs && (uint64_t)(hi - lo) * 2 >= e->g.n) {
            // worth its fixed cost (count + compact the whole input) only when this call covers most of the rows
            c.sums = StageChoice::kCompactWhole;
            c.fused_counts = fused_in == stage;
            // Last stage: k_c4_agg is a persistent grid that holds nearly all LDS of its CUs but leaves most VALU
            // cycles idle, and the dense-only sigmoid kernel that follows is VALU-bound and needs no LDS.  So the sums
            // are launched one round (256 chunks) at a time and the dense kernel of round k goes to the aux stream,
            // under the sums of round k + 1.  Metric graph: 1.71 -> 1.58 ms.  (Not for the feature stages: their dense
            // kernels store 64-byte rows and slow the co-running sums by more than is gained.)
            c.rounds = e->opt_overlap && sp.variant == 2 && e->opt_mfma != 1;
            c.emit = may_emit;   // this stage's own (aggregate-only, VALU) kernel produces for the next one
        }
    }
    // degree-sorted tiles on skewed graphs (every stage; the list is cached per row range)
    int rc = ensure_sorted(e, lo, hi);
    if (rc) return rc;
    if (e->srt_cur >= 0 && e->srt[e->srt_cur].use) {
        c.sorted.n = e->srt[e->srt_cur].n;
        c.sorted.vertex = e->srt[e->srt_cur].vertex.p;
        c.sorted.meta = e->srt[e->srt_cur].meta.p;
        c.rounds = false;   // (the plan forced onto a graph with sorted tiles: the gathering kernel does the stage)
    }
    return GNNVC_OK;
}

// fork: the long (and giant) rows of this stage beside the tile kernel
int launch_side_rows(gnnvc_engine *e, const GraphDev &gv, int stage, uint32_t lo, uint32_t hi, const float *in, float *out,
                     float *logits, uint32_t thr, bool plain_f1) {
    // One side queue beside the main one (ensure_side_streams).  The giant rows' walk is a latency chain on a few waves and
    // always goes there; the long rows join it — unless that walk is what a stage waits for (find_giant: the power-law graph),
    // then they run ahead of the tile kernel on the main queue instead: power-law 1 M 0.89 ms (long rows beside the giant
    // walk: 1.15), R-MAT-22 2.62 ms (long rows on the main queue: 2.79).
    const bool side = e->opt_side_streams != 0;
    const bool long_on_main = side && (e->opt_long_on_main < 0 ? (e->n_giant != 0 && e->giant_walk_bound) : e->opt_long_on_main != 0);
    const bool side_long = side && !long_on_main;
    hipStream_t s_long = side_long ? e->long_stream : e->stream;
    hipStream_t s_giant = !side ? e->stream : (long_on_main ? e->giant_stream : e->long_stream);
    e->side_join = !side ? 0 : (long_on_main ? 2 : 1);
    // rows from this degree on go the giant way in this stage
    const uint32_t giant_from = e->stages[stage].f == 16 ? e->giant_f16() : e->giant_thresh;
    gnnvc::GiantRows gr;
    bool gather_first = false;
    if (e->n_giant) {
        gr.n = e->n_giant;
        gr.blocks = e->giant_blocks;
        gr.meta = e->gi_meta.p;
        gr.off = e->gi_off.p;
        gr.slab = e->gi_slab.p;
        gr.agg = e->gi_agg.p;
        if (e->gi_maxseg > 1) {
            gr.segsum = e->gi_segsum.p;
            gr.segmap = e->gi_segmap.p;
            gr.maxseg = e->gi_maxseg;
        }
        // The giant rows' gather on the main queue, ahead of the fork (see launch_giant_stage) — where it is small (a throughput
        // kernel that delays the tile kernel by what it takes alone: R-MAT-22's 7.5 M giant entries 0.03 - 0.05 ms, R-MAT-24's
        // 64 M 0.3 - 0.6 ms) and the long rows are on the side queue behind the walk.  Measured: R-MAT-20 0.89 -> 0.83 ms
        // (first forward 1.32 -> 1.20), R-MAT-22 first forward 3.75 -> 3.57 (steady the same); R-MAT-24 9.5 -> 10.8 and
        // power-law 0.84 -> 0.87 the wrong way, hence the two conditions.
        // ... or, however large, in an F = 1 stage that has no plan (a graph's first forward): every long row's chain is on the
        // side queue there and the gather at its head waited for slots for as long as the tile kernel ran (R-MAT-24: 3.1 ms
        // for a kernel that takes 0.6 alone; the stage 6.1 ms against the tile kernel's 3.3)
        gather_first = side && (e->opt_giant_gather_first < 0 ? (!long_on_main && (e->giant_entries <= (16ull << 20) || plain_f1))
                                                               : e->opt_giant_gather_first != 0);
        if (gather_first)
            HIP_TRY(e, gnnvc::launch_giant_stage(e->stages[stage], e->opt_prune_giant ? gv : e->g, e->ws, e->params.p, in, out, logits, lo, hi,
                                                 gr, e->stream, giant_from, /*part=*/1));
    }
    if (side) {
        HIP_TRY(e, hipEventRecord(e->ev_fork, e->stream));
        if (side_long) HIP_TRY(e, hipStreamWaitEvent(e->long_stream, e->ev_fork, 0));
    }
    if (e->n_giant) {   // the heaviest rows: beside the tile kernel
        if (long_on_main) HIP_TRY(e, hipStreamWaitEvent(e->giant_stream, e->ev_fork, 0));
        HIP_TRY(e, gnnvc::launch_giant_stage(e->stages[stage], e->opt_prune_giant ? gv : e->g, e->ws, e->params.p, in, out, logits, lo, hi, gr,
                                             s_giant, giant_from, gather_first ? 2 : 0));
        if (long_on_main) HIP_TRY(e, hipEventRecord(e->ev_giant, e->giant_stream));
    }
    // (with rows classed by the entries they have left, k_long_* takes rows from gv.eff_thresh entries on whatever their degree)
    const uint32_t long_from = gv.prune_eff ? std::min(thr, gv.eff_thresh) : thr;
    if ((e->n_giant < e->n_long || giant_from > e->giant_thresh) && long_from < giant_from)   // (equal: a plan or the giant kernels have every row in between)
        HIP_TRY(e, gnnvc::launch_long_stage(e->stages[stage], gv, e->ws, e->params.p, in, out, logits, lo, hi,
                                            e->long_list.p, e->n_long, thr, giant_from, s_long));
    if (side_long) HIP_TRY(e, hipEventRecord(e->ev_long, e->long_stream));
    return GNNVC_OK;
}

int launch_main(gnnvc_engine *e, const StageChoice &c, const GraphDev &gv, const gnnvc::SortedOrder &so_p, int stage, uint32_t lo,
                uint32_t hi, const float *in, float *out, float *logits) {
    const gnnvc::StagePlan &sp = e->stages[stage];
    gnnvc::EmitArgs emit;
    // (the counters are zeroed only AFTER this stage's own k_c4_choose has read what the previous stage kernel left in them)
    auto arm_emit = [&]() -> int {
        if (!c.emit) return GNNVC_OK;
        HIP_TRY(e, hipMemsetAsync(e->c4_emit_counts.p, 0, gnnvc::kEmitCounters * sizeof(unsigned long long), e->stream));
        emit.spec = e->c4_desc.p + gnnvc_engine::kDescWords * stage;   // consumer stage `stage + 1`: its descriptor words
        emit.table = e->c4_table.p;
        emit.counts = e->c4_emit_counts.p;
        e->c4_fused_for = stage + 1;
        return GNNVC_OK;
    };
    // First use of the compact-table plan on this graph: no earlier forward has said which four columns the NEXT stage's
    // table holds, so this stage's kernel could only count, and the consumer would make a pass of its own over its whole
    // input to write the table (k_c4_compact: 0.25 ms per stage on the metric graph — what a graph scored ONCE pays in full).
    // A pilot — the plain gathering kernel over the first rows of this very stage — makes that choice ahead of the real
    // run, which then writes the table on its way.  The consumer's k_c4_choose still decides from the counts of ALL rows
    // and has the table rewritten if the pilot chose otherwise: the pilot changes time, never a result.
    auto pilot = [&]() -> int {
        if (!c.emit || e->c4_seeded[stage + 1] || !e->opt_pilot_rows) return GNNVC_OK;
        e->c4_seeded[stage + 1] = true;
        const uint32_t rows = std::min(e->opt_pilot_rows, hi - lo);
        if ((uint64_t)rows * 8 > (uint64_t)(hi - lo)) return GNNVC_OK;   // (a graph this small is its own pilot: not worth a launch)
        uint32_t *cons = e->c4_desc.p + gnnvc_engine::kDescWords * stage;   // consumer stage `stage + 1`
        HIP_TRY(e, hipMemsetAsync(e->c4_emit_counts.p, 0, gnnvc::kEmitCounters * sizeof(unsigned long long), e->stream));
        gnnvc::EmitArgs pe;
        pe.spec = e->c4_desc.p + 2 * gnnvc_engine::kDescWords + 4;   // a word that is always 0: count, do not write table rows
        pe.table = e->c4_table.p;
        pe.counts = e->c4_emit_counts.p;
        HIP_TRY(e, gnnvc::launch_stage(sp, e->g, e->ws, e->params.p, in, out, nullptr, lo, lo + rows, 0xFFFFFFFFu, /*mfma=*/false, nullptr,
                                       /*interleave=*/true, e->stream, nullptr, nullptr, nullptr, false, pe));
        HIP_TRY(e, gnnvc::compact_choose(e->c4_emit_counts.p, 64, rows, cons, 1u, e->stream));
        return GNNVC_OK;
    };
    if (c.emit_t4) {   // this stage's epilogue counts its output's columns and writes the next stage's table (for that stage's last choice)
        const int cons = stage + 1;
        emit.spec = e->t4_desc_of(cons, e->t4_parity);
        emit.table = e->t4_table[cons - 1].p;
        emit.counts = e->t4_counts_of(cons, e->t4_parity);   // (cleared by the consumer's launch of the previous forward: k_stage_t4)
    }
    if (c.sums == StageChoice::kTableTiles) {
        uint32_t *d_in = e->t4_desc_of(stage, e->t4_parity), *d_out = e->t4_desc_of(stage, e->t4_parity ^ 1u);
        const bool solo = e->t4_fit_seen[stage] && e->opt_t4_solo;
        HIP_TRY(e, gnnvc::launch_stage_t4(sp, e->g, e->ws, e->params.p, in, out, logits, lo, hi, e->interleave, e->stream,
                                          e->t4_table[stage - 1].p, e->t4_counts_of(stage, e->t4_parity),
                                          e->t4_counts_of(stage, e->t4_parity ^ 1u), d_in, d_out, emit, solo));
        if (solo) return GNNVC_OK;
        // ... and the gathering kernel, which leaves at once when the table tiles did the rows (d_out[8]: decided on the device)
        return hip_rc(e, gnnvc::launch_stage(sp, gv, e->ws, e->params.p, in, out, logits, lo, hi, c.long_thresh, c.mfma, nullptr,
                                             e->interleave, e->stream, nullptr, nullptr, nullptr, false, gnnvc::EmitArgs(), true, nullptr,
                                             nullptr, d_out + 8));
    }
    if (c.sums == StageChoice::kLdsTable || c.sums == StageChoice::kBlocked) {
        int rc = pilot();
        if (rc) return rc;
        rc = arm_emit();
        if (rc) return rc;
    }
    if (c.sums == StageChoice::kLdsTable)
        return hip_rc(e, gnnvc::launch_stage0_lds_table(sp, e->g, e->ws, e->params.p, in, out, lo, hi, e->lt_rows, e->lt_stepptr.p,
                                                        e->lt_steps.p, e->lt_entries.p, e->lt_bytes.p, e->blk_acc.p, e->lt_bad.p,
                                                        c.long_thresh, e->opt_mfma == 1, e->interleave, e->stream, emit,
                                                        e->lt_last_entry, e->lt_mapped ? e->lt_rowmap.p : nullptr,
                                                        e->lt_mapped ? e->lt_chunks : 0u, e->lt_base, e->lt_end, e->lt_bits));
    if (c.sums == StageChoice::kBlocked)
        return hip_rc(e, gnnvc::launch_stage0_blocked(sp, e->g, e->ws, e->params.p, in, out, lo, hi, e->blk_count, e->blk_ptr.p,
                                                      e->blk_col.p, e->blk_acc.p, e->long_thresh, e->opt_mfma == 1, e->interleave,
                                                      e->stream, emit));
    const float *acc4 = nullptr;
    uint32_t *desc = nullptr;
    if (c.sums == StageChoice::kCompactPrepared || c.sums == StageChoice::kCompactWhole) {
        desc = e->c4_desc.p + gnnvc_engine::kDescWords * (stage - 1);
        e->c4_last_desc = gnnvc_engine::kDescWords * (stage - 1);
        if (c.sums == StageChoice::kCompactWhole) e->fit_used[stage] = true;
        acc4 = e->c4_acc.p;
        const bool whole = c.sums == StageChoice::kCompactWhole;
        if (whole && !c.fused_counts) HIP_TRY(e, gnnvc::column_counts(in, e->g.n, e->c4_counts.p, e->stream));
        const bool fused = whole && c.fused_counts;
        // what: 1 = choose the columns + write the table, 2 = the sums; the prepared table only needs its sums, a call
        // that runs its sums round by round (below) only the preparation
        const int what = !whole ? 2 : (c.rounds ? 1 : 3);
        HIP_TRY(e, gnnvc::launch_compact_gather(e->g, compact_plan(e), in, fused ? e->c4_emit_counts.p : e->c4_counts.p, fused ? 64 : 1,
                                                desc, e->c4_table.p, e->c4_acc.p, lo, hi, e->c4_dirty.p, e->c4_dirty_cap,
                                                e->c4_agg16.p, e->stream, what));
    }
    if (c.sums != StageChoice::kLdsTable && c.sums != StageChoice::kBlocked) {
        int rc = pilot();
        if (rc) return rc;
        rc = arm_emit();
        if (rc) return rc;
    }
    const gnnvc::SortedOrder *sop = c.sorted.n ? &c.sorted : nullptr;
    // Wide tiles (k_stage_w*): a graph with fewer tiles than the chip has SIMDs — the reference CLI's later predict calls — and
    // nothing but the plain gather to run (no plan, no long rows, no pruned adjacency, nothing to emit): a workgroup per tile
    if (c.sums == StageChoice::kGather && e->opt_wide && e->g.n <= (sp.f == 16 ? e->opt_wide_max_n16 : e->opt_wide_max_n) && !e->g.sliced() && e->n_long == 0 && !sop &&
        !acc4 && !emit.counts && gv.prune_bad == nullptr && gv.zero_bits == nullptr && sp.variant >= 0 && sp.variant <= 2) {
        e->wide_used = true;
        return hip_rc(e, gnnvc::launch_stage_wide(sp, e->g, e->ws, e->params.p, in, out, logits, lo, hi, e->stream));
    }
    HIP_TRY(e, gnnvc::launch_stage(sp, gv, e->ws, e->params.p, in, out, logits, lo, hi, c.long_thresh, c.mfma, sop,
                                   e->interleave, e->stream, acc4, desc, e->c4_agg16.p, e->opt_mfma == 1, emit,
                                   /*dense_part=*/!c.rounds, so_p.vertex ? &so_p : nullptr,
                                   (acc4 && e->opt_dense_skip) ? e->c4_table.p : nullptr));
    if (!c.rounds) return GNNVC_OK;
    HIP_TRY(e, hipMemsetAsync(desc + 5, 0, sizeof(uint32_t), e->stream));        // dirty-row counter
    HIP_TRY(e, hipMemsetAsync(e->c4_marks.p, 0, sizeof(uint32_t), e->stream));   // marks[0]
    const uint32_t rows = e->c4_rows, c0 = (lo - e->c4_base) / rows, c1 = (hi - 1 - e->c4_base) / rows;
    const uint32_t nrounds = (c1 - c0) / 256u + 1u;
    if (nrounds + 1 > 64) return fail(e, GNNVC_ERR_UNSUPPORTED, "too many rounds of the compact-table plan");
    {
        int rc = ensure_round_events(e, nrounds);
        if (rc) return rc;
    }
    for (uint32_t k = 0; k < nrounds; ++k) {
        const uint32_t ca = c0 + 256u * k, cb = std::min(c1 + 1u, ca + 256u);
        const uint32_t ra = std::max(lo, e->c4_base + ca * rows);
        const uint32_t rb = (uint32_t)std::min<uint64_t>(hi, (uint64_t)e->c4_base + (uint64_t)cb * rows);
        HIP_TRY(e, gnnvc::compact_sums(e->g, compact_plan(e), desc, e->c4_table.p, e->c4_acc.p, ra, rb, e->c4_dirty.p, e->c4_dirty_cap,
                                       e->stream, /*one_round=*/true));
        HIP_TRY(e, gnnvc::compact_mark(desc, e->c4_marks.p, k + 1, e->stream));
        const bool last = k + 1 == nrounds;
        hipStream_t ds = last ? e->stream : e->aux_stream;
        if (!last) {
            HIP_TRY(e, hipEventRecord(e->round_ev[k], e->stream));
            HIP_TRY(e, hipStreamWaitEvent(e->aux_stream, e->round_ev[k], 0));
        }
        HIP_TRY(e, gnnvc::compact_fix(e->g, in, desc, e->c4_dirty.p, e->c4_dirty_cap, e->c4_agg16.p, e->c4_marks.p + k, ds,
                                      /*blocks=*/64));
        HIP_TRY(e, gnnvc::launch_dense_sigmoid(sp, e->g, e->ws, e->params.p, in, out, logits, ra, rb, e->c4_acc.p, desc,
                                               e->c4_agg16.p, ds, 0xFFFFFFFFu, e->opt_dense_skip ? e->c4_table.p : nullptr));
    }
    if (nrounds > 1) {
        HIP_TRY(e, hipEventRecord(e->ev_join, e->aux_stream));
        HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_join, 0));
    }
    return GNNVC_OK;
}

int run_stage(gnnvc_engine *e, int stage, uint32_t lo, uint32_t hi, const float *in, float *out, float *logits,
              bool in_forward = false) {
    StageChoice c;
    int rc = choose_stage(e, stage, lo, hi, in, out, in_forward, c);
    if (rc) return rc;
    const bool longs = e->n_long > 0;
    GraphDev gv;   // (the check behind a pruned adjacency is queued before the fork to the side streams)
    gnnvc::SortedOrder so_p;
    rc = gather_view(e, stage, lo, hi, in, c.sums == StageChoice::kGather, c.sorted.n != 0, gv, so_p, c.mfma, c.long_thresh);
    if (rc) return rc;
    if (longs) {
        rc = launch_side_rows(e, gv, stage, lo, hi, in, out, logits, c.long_thresh,
                              /*plain_f1=*/e->stages[stage].f == 1 && c.sums == StageChoice::kGather);
        if (rc) return rc;
    }
    rc = launch_main(e, c, gv, so_p, stage, lo, hi, in, out, logits);
    if (rc) return rc;
    if (longs && e->side_join) {   // join: the side queue's last kernel of this stage
        if (e->side_join == 1) HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_long, 0));
        else if (e->n_giant) HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_giant, 0));
    }
    return GNNVC_OK;
}

// Layer-by-layer forward on device buffers (any model).
int forward_unfused(gnnvc_engine *e, const float *d_x, float *d_out, float *d_logits) {
    const uint32_t n = e->g.n;
    const size_t cap = ((size_t)n + 1) * (size_t)e->max_width;
    for (auto &s : e->scratch) HIP_TRY(e, s.reserve(cap));
    const float *cur = d_x;
    int wd = e->in_width, pp = 0;
    for (size_t i = 0; i < e->layers.size(); ++i) {
        const Layer &l = e->layers[i];
        const bool last = i + 1 == e->layers.size();
        float *dst = last ? d_out : e->scratch[pp].p;
        switch (l.kind) {
        case kGraph:
            HIP_TRY(e, gnnvc::launch_graph_layer(e->g, e->ws, (uint32_t)wd, cur, dst, e->stream));
            wd = 2 * wd + 3;
            break;
        case kLinear:
            HIP_TRY(e, gnnvc::launch_linear(n, l.k, l.m, cur, e->params.p + l.w_off,
                                            e->params.p + l.b_off, dst, e->stream));
            wd = (int)l.m;
            break;
        case kRelu:
            HIP_TRY(e, gnnvc::launch_relu((size_t)n * wd, cur, dst, e->stream));
            break;
        case kSigmoid:
            if (last && d_logits)
                HIP_TRY(e, hipMemcpyAsync(d_logits, cur, (size_t)n * wd * sizeof(float),
                                          hipMemcpyDeviceToDevice, e->stream));
            HIP_TRY(e, gnnvc::launch_sigmoid((size_t)n * wd, cur, dst, e->stream));
            break;
        }
        cur = dst;
        pp ^= 1;
    }
    return GNNVC_OK;
}

}  // namespace


// ============================================================================ ABI

extern "C" {

int gnnvc_abi_version(void) { return GNNVC_ABI_VERSION; }

const char *gnnvc_strerror(int code) {
    switch (code) {
    case GNNVC_OK: return "ok";
    case GNNVC_ERR_INVALID: return "invalid argument or malformed model";
    case GNNVC_ERR_DEVICE: return "HIP device unavailable or HIP call failed";
    case GNNVC_ERR_NOMEM: return "out of memory";
    case GNNVC_ERR_STATE: return "call out of order";
    case GNNVC_ERR_UNSUPPORTED: return "unsupported model or size";
    default: return "unknown error";
    }
}

const char *gnnvc_last_error(const gnnvc_engine *e) { return e ? e->err.c_str() : ""; }

int gnnvc_create(gnnvc_engine **out, const char *model_text, size_t len, int device) {
    if (!out || !model_text) return GNNVC_ERR_INVALID;
    *out = nullptr;
    gnnvc_engine *e = new (std::nothrow) gnnvc_engine();
    if (!e) return GNNVC_ERR_NOMEM;
    int rc = GNNVC_OK;
    try {
        e->device = device;
        rc = parse_model(e, model_text, len);
        if (rc == GNNVC_OK) rc = plan_model(e);
        if (rc == GNNVC_OK) {
            int count = 0;
            if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
                rc = fail(e, GNNVC_ERR_DEVICE, "no HIP device %d (found %d) — this engine has no CPU path", device, count);
        }
        if (rc == GNNVC_OK) rc = use_device(e);
        if (rc == GNNVC_OK) {
            if (hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess)
                rc = fail(e, GNNVC_ERR_DEVICE, "hipStreamCreate failed");
            e->stream = e->own_stream;
        }
        // (creating a stream takes ~10 ms on this stack: the side queues are made here, once per engine, not inside the first
        // hand-off or forward that wants them)
        if (rc == GNNVC_OK) rc = ensure_round_events(e, 0);
        if (rc == GNNVC_OK) rc = upload_params(e);
        // (the page-locked words a forward's verdicts are written to: a pinned allocation is tens of microseconds — here, not in
        // the first forward that asks)
        if (rc == GNNVC_OK && (e->fit_pin.reserve(8) != hipSuccess ||
                               hipHostGetDevicePointer(reinterpret_cast<void **>(&e->fit_dev), e->fit_pin.p, 0) != hipSuccess))
            rc = fail(e, GNNVC_ERR_NOMEM, "page-locked verdict words");
        if (rc == GNNVC_OK && hipEventCreateWithFlags(&e->ev_fit, hipEventDisableTiming) != hipSuccess)
            rc = fail(e, GNNVC_ERR_DEVICE, "hipEventCreate failed");
    } catch (const std::bad_alloc &) {
        rc = GNNVC_ERR_NOMEM;
    } catch (...) {
        rc = GNNVC_ERR_INVALID;
    }
    if (rc != GNNVC_OK) {
        // keep the message reachable for the caller through stderr: there is no engine to ask
        if (!e->err.empty()) fprintf(stderr, "gnnvc_create: %s\n", e->err.c_str());
        gnnvc_destroy(e);
        return rc;
    }
    *out = e;
    return GNNVC_OK;
}

int gnnvc_create_multi(gnnvc_engine **out, const char *model_text, size_t len, const int *devices, int n_devices) {
    if (!out || !model_text || !devices || n_devices < 1) return GNNVC_ERR_INVALID;
    *out = nullptr;
    gnnvc_engine *e = nullptr;
    int rc = gnnvc_create(&e, model_text, len, devices[0]);
    if (rc != GNNVC_OK) return rc;
    std::string err;
    try {
        rc = gnnvc::multi_create(&e->multi, model_text, len, devices, n_devices, err);
    } catch (const std::bad_alloc &) {
        rc = GNNVC_ERR_NOMEM;
    } catch (...) {
        rc = GNNVC_ERR_INVALID;
    }
    if (rc != GNNVC_OK) {
        if (!err.empty()) fprintf(stderr, "gnnvc_create_multi: %s\n", err.c_str());
        gnnvc_destroy(e);
        return rc;
    }
    (void)hipSetDevice(devices[0]);
    *out = e;
    return GNNVC_OK;
}

void gnnvc_destroy(gnnvc_engine *e) {
    if (!e) return;
    if (e->multi) {
        gnnvc::multi_destroy(e->multi);
        e->multi = nullptr;
    }
    if (e->own_stream) {
        (void)hipSetDevice(e->device);
        (void)hipStreamSynchronize(e->own_stream);
    }
    e->params.release();
    e->rowptr.release(); e->col.release(); e->w.release(); e->nw.release();
    e->x.release(); e->h[0].release(); e->h[1].release();
    e->scores.release(); e->logits.release();
    e->scratch[0].release(); e->scratch[1].release();
    e->blk_ptr.release(); e->blk_col.release(); e->blk_scratch.release(); e->blk_flag.release();
    e->blk_acc.release();
    e->lt_bytes.release(); e->lt_entries.release(); e->lt_segcnt.release(); e->lt_stepptr.release();
    e->lt_stepcnt.release(); e->lt_bad.release(); e->lt_steps.release(); e->lt_rowmap.release(); e->lt_first.release(); e->lt_bstart.release();
    e->c4_entries.release(); e->c4_segcnt.release(); e->c4_stepptr.release(); e->c4_stepcnt.release();
    for (auto &t : e->t4_table) t.release();
    for (auto &t : e->t4_counts) t.release();
    e->t4_desc.release();
    e->c4_desc.release(); e->c4_map_vertex.release(); e->c4_map_meta.release(); e->map_coarse.release(); e->c4_steps.release(); e->c4_table.release(); e->c4_marks.release(); e->c4_acc.release(); e->c4_counts.release();
    e->c4_agg16.release(); e->c4_dirty.release(); e->c4_emit_counts.release();
    for (auto &pp : e->prune) { pp.prp.release(); pp.pcol.release(); pp.heavy.release(); pp.svertex.release(); pp.smeta.release(); }
    e->prune_flags.release(); e->prune_scratch.release(); e->prune_off.release(); e->prune_mask.release();
    for (auto &b : e->filter_bits) b.release();
    e->filter_info.release();
    e->long_list.release(); e->long_count.release(); e->cls_dev.release(); e->cls_pin.release();
    e->gi_meta.release(); e->gi_off.release(); e->gi_slab.release(); e->gi_agg.release(); e->gi_segsum.release(); e->gi_segmap.release();
    e->rowptr2.release(); e->col2.release(); e->der_old_row.release(); e->der_new_of.release(); e->der_tail.release();
    e->der_tailptr.release(); e->der_tailcols.release(); e->hash_buf.release();
    e->pin_rowptr.release(); e->pin_col.release(); e->pin_w.release(); e->pin_nw.release(); e->fit_pin.release();
    if (e->ev_fit) (void)hipEventDestroy(e->ev_fit);
    if (e->ev_piece) (void)hipEventDestroy(e->ev_piece);
    e->pin_small.release();
    e->pin_info.release();
    e->dev_info.release();
    for (auto &r : e->srt) { r.vertex.release(); r.meta.release(); }
    e->srt_hist.release(); e->srt_sum.release();
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->aux_stream) { (void)hipStreamSynchronize(e->aux_stream); (void)hipStreamDestroy(e->aux_stream); }
    if (e->ev_long) (void)hipEventDestroy(e->ev_long);


    if (e->ev_giant) (void)hipEventDestroy(e->ev_giant);
    for (auto v : e->ev) (void)hipEventDestroy(v);
    for (auto v : e->round_ev) (void)hipEventDestroy(v);
    for (auto &r : e->ktrace.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

int gnnvc_set_weight_scale(gnnvc_engine *e, float ws) {
    if (!e) return GNNVC_ERR_INVALID;
    e->ws = ws;
    if (e->multi) return gnnvc::multi_set_weight_scale(e->multi, ws);
    return GNNVC_OK;
}

int gnnvc_set_stream(gnnvc_engine *e, void *hip_stream) {
    if (!e) return GNNVC_ERR_INVALID;
    hipStream_t want = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
    if (want == e->stream) return GNNVC_OK;
    e->stream = want;
    if (e->multi) return GNNVC_OK;
    int rc = use_device(e);
    if (rc) return rc;
    return reprobe_side_streams(e);   // (the side queue has to sit on another hardware queue than THIS stream)
}

int gnnvc_get_stream(gnnvc_engine *e, void **hip_stream) {
    if (!e || !hip_stream) return GNNVC_ERR_INVALID;
    *hip_stream = e->stream;
    return GNNVC_OK;
}

int gnnvc_set_option(gnnvc_engine *e, const char *key, long value) {
    if (!e || !key) return GNNVC_ERR_INVALID;
    const std::string k(key);
    if (k.rfind("multi_", 0) == 0) {   // the exchange of a multi-device handle (gnnvc_multi.cpp): "multi_pieces", "multi_pack", "multi_push", "multi_only_part"
        if (!e->multi) return fail(e, GNNVC_ERR_INVALID, "option '%s' needs a multi-device handle (gnnvc_create_multi)", key);
        const int rc = gnnvc::multi_set_option(e->multi, key, value);
        return rc ? fail(e, rc, "unknown option '%s'", key) : GNNVC_OK;
    }
    if (k == "poison_features") {
        e->opt_poison = value != 0 ? 1 : 0;
        return e->multi ? gnnvc::multi_set_option(e->multi, key, value) : GNNVC_OK;
    }
    if (k == "verdict_period") {
        e->opt_verdict_period = value < 1 ? 1u : (value > 64 ? 64u : (uint32_t)value);
        return e->multi ? gnnvc::multi_set_option(e->multi, key, value) : GNNVC_OK;
    }
    if (k == "forward_timing") {   // (touches nothing a forward has cached)
        e->opt_timing = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
        return e->multi ? gnnvc::multi_set_option(e->multi, key, value) : GNNVC_OK;
    }
    e->short_from = 0;   // (lists a filtered stage left go by the thresholds and variants of the call that wrote them)
    if (k == "blocked_stage0") e->opt_blocked = value < 0 ? 0 : (value > 2 ? 2 : (int)value);   // 2 = also on skewed graphs
    else if (k == "block_cols") e->opt_block_cols = value > 0 ? (uint32_t)value : 0;
    else if (k == "blocked_min_n") e->opt_blocked_min_n = value > 0 ? (uint32_t)value : 0;
    else if (k == "compact_min_n") e->opt_compact_min_n = value > 0 ? (uint32_t)value : 0;
    else if (k == "compact_first_forward_entries") e->opt_compact_first_entries = value > 0 ? (uint64_t)value : 0;
    else if (k == "plan_chunk_rows") e->opt_plan_chunk_rows = value > 0 ? (uint32_t)value : 0;
    else if (k == "overlap_dense") e->opt_overlap = value != 0 ? 1 : 0;
    else if (k == "long_row_threshold") { e->opt_long_thresh = value > 0 ? (uint32_t)value : 0; e->opt_long_auto = false; }
    else if (k == "giant_row_threshold") {   // (an explicit threshold holds for every stage; "giant_row_threshold_f16" afterwards refines it)
        e->opt_giant_thresh = value > 0 ? (uint32_t)std::max<long>(value, 64) : 0;
        e->opt_giant_f16 = e->opt_giant_thresh ? e->opt_giant_thresh : 1u;
        e->giant_f16_auto = false;
    }
    else if (k == "giant_row_threshold_f16") { e->opt_giant_f16 = value > 0 ? (uint32_t)value : 1u; e->giant_f16_auto = false; for (auto &pp : e->prune) pp.forget(); }
    else if (k == "giant_segments") e->opt_giant_segments = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "side_streams") e->opt_side_streams = value != 0 ? 1 : 0;
    else if (k == "kernel_trace") e->opt_ktrace = value != 0 ? 1 : 0;
    else if (k == "compact_gather") e->opt_compact = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
    else if (k == "prune_zero_rows") { e->opt_prune = value > 0 ? 1 : 0; for (auto &pp : e->prune) pp.forget(); }
    else if (k == "prune_class_by_entries_left") { e->opt_prune_eff = value != 0 ? 1 : 0; for (auto &pp : e->prune) pp.forget(); }
    else if (k == "prune_heavy_entries") { e->opt_prune_heavy_entries = value > 0 ? (uint64_t)value : 0; for (auto &pp : e->prune) pp.forget(); }
    else if (k == "prune_early_entries") e->opt_prune_early_nnz = value > 0 ? (uint64_t)value : 0;
    else if (k == "prune_giant_rows") e->opt_prune_giant = value != 0 ? 1 : 0;
    else if (k == "prune_predict") e->opt_prune_predict = value != 0 ? 1 : 0;
    else if (k == "wide_tiles") e->opt_wide = value != 0 ? 1 : 0;
    else if (k == "wide_tiles_max_n") e->opt_wide_max_n = value > 0 ? (uint32_t)value : 0u;
    else if (k == "wide_tiles_max_n_f16") e->opt_wide_max_n16 = value > 0 ? (uint32_t)value : 0u;
    else if (k == "dense_skip_zeros") e->opt_dense_skip = value != 0 ? 1 : 0;
    else if (k == "table_tiles") e->opt_t4 = value != 0 ? 1 : 0;
    else if (k == "table_tiles_solo") e->opt_t4_solo = value != 0 ? 1 : 0;
    else if (k == "table_tiles_min_n") e->opt_t4_min_n = value > 0 ? (uint32_t)value : 0u;
    else if (k == "table_tiles_max_bytes") e->opt_t4_max_bytes = value > 0 ? (uint64_t)value : 0ull;
    else if (k == "prune_predict_min_entries") e->opt_predict_min_nnz = value > 0 ? (uint64_t)value : 0;
    else if (k == "giant_gather_first") e->opt_giant_gather_first = value < 0 ? -1 : (value != 0 ? 1 : 0);
    else if (k == "long_rows_on_main") e->opt_long_on_main = value < 0 ? -1 : (value != 0 ? 1 : 0);
    else if (k == "filter_zero_rows") e->opt_filter = value != 0 ? 1 : 0;
    else if (k == "filter_keep_lists") e->opt_filter_keep = value != 0 ? 1 : 0;
    else if (k == "filter_min_entries") e->opt_filter_min_nnz = value > 0 ? (uint64_t)value : 0;
    else if (k == "filter_min_long_percent") e->opt_filter_min_long_pct = value < 0 ? 0u : (value > 100 ? 101u : (uint32_t)value);
    else if (k == "filter_min_percent") e->opt_filter_min_pct = value < 0 ? 0u : (value > 100 ? 101u : (uint32_t)value);
    else if (k == "prune_min_entries") e->opt_prune_min_nnz = value > 0 ? (uint64_t)value : 0;
    else if (k == "prune_min_drop_percent") e->opt_prune_min_drop = value < 0 ? 0u : (value > 100 ? 100u : (uint32_t)value);
    else if (k == "lds_table_skewed") e->opt_lds_skewed = value != 0 ? 1 : 0;
    else if (k == "lds_table_skewed_rows") e->opt_lds_skewed_rows = value > 0 ? (uint32_t)value : 0u;
    else if (k == "lds_table") e->opt_lds_table = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
    else if (k == "lds_table_min_chunks") e->opt_lt_min_chunks = value > 0 ? (uint32_t)value : 1u;
    else if (k == "lds_table_bits") e->opt_lt_bits = (value == 8 || value == 10 || value == 16) ? (int)value : 0;
    else if (k == "plans_at_handoff") e->opt_handoff = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
    else if (k == "handoff_min_entries") e->opt_handoff_min_nnz = value > 0 ? (uint64_t)value : 0;
    else if (k == "pilot_rows") e->opt_pilot_rows = value > 0 ? (uint32_t)value : 0u;
    else if (k == "sorted_min_nnz") e->opt_sorted_min_nnz = value > 0 ? (uint64_t)value : 0;
    else if (k == "sorted_long_row_threshold") { e->opt_sorted_long_thresh = value > 0 ? (uint32_t)value : 1; e->opt_long_auto = false; }
    else if (k == "mfma_dense") e->opt_mfma = (value >= 0 && value <= 2) ? (int)value : 2;
    else if (k == "sorted_tiles") { e->opt_sorted = value < 0 ? -1 : (value ? 1 : 0); for (auto &r : e->srt) r.valid = false; }
    else return fail(e, GNNVC_ERR_INVALID, "unknown option '%s'", key);
    if (e->multi) return gnnvc::multi_set_option(e->multi, key, value);
    return GNNVC_OK;
}

int gnnvc_get_info(const gnnvc_engine *e, const char *key, long *value) {
    if (!e || !key || !value) return GNNVC_ERR_INVALID;
    const std::string k(key);
    if (k == "devices") *value = e->multi ? gnnvc::multi_devices(e->multi) : 1;
    else if (e->multi && (k.rfind("part_rows_", 0) == 0 || k.rfind("part_entries_", 0) == 0)) {   // "part_rows_<r>", "part_entries_<r>"
        const int r = atoi(k.c_str() + k.rfind('_') + 1);
        uint32_t lo = 0, hi = 0;
        uint64_t en = 0;
        if (gnnvc::multi_part_info(e->multi, r, &lo, &hi, &en) != GNNVC_OK) return GNNVC_ERR_INVALID;
        *value = k[5] == 'r' ? (long)(hi - lo) : (long)en;
    }
    else if (k == "multi_last_forward_us") *value = e->multi ? (long)(gnnvc::multi_last_forward_ms(e->multi) * 1000.0) : 0;
    else if (k.rfind("multi_", 0) == 0) {
        if (!e->multi || !gnnvc::multi_get_info(e->multi, key, value)) return GNNVC_ERR_INVALID;
    }
    else if (k == "compact_gather_active") *value = e->c4_ready ? 1 : 0;
    else if (k == "compact_gather_chunks") *value = e->c4_ready ? (long)e->c4_chunks : 0;
    else if (k == "compact_gather_block_cols") *value = e->c4_ready ? (long)e->c4_block : 0;
    else if (k == "compact_gather_rows_per_chunk") *value = e->c4_ready ? (long)e->c4_rows : 0;
    else if (k == "compact_gather_steps") *value = e->c4_ready ? (long)e->c4_steps_total : 0;
    else if (k == "pruned_stage1" || k == "pruned_stage2") *value = e->prune[k.back() - '0'].ready ? 1 : 0;
    else if (k == "side_queue_probes") *value = e->side_probes;
    else if (k == "side_queue_runs_beside") *value = e->side_beside ? 1 : 0;
    else if (k == "long_entries_percent") *value = e->n_long && e->g.nnz ? (long)(e->long_entries * 100ull / e->g.nnz) : 0;
    else if (k == "filtered_stage1" || k == "filtered_stage2") *value = e->filtered[k.back() - '0'] ? 1 : 0;
    else if (k == "short_lists_stage1" || k == "short_lists_stage2") *value = e->short_used[k.back() - '0'] ? 1 : 0;
    else if (k == "filter_mass_percent_stage1" || k == "filter_mass_percent_stage2") {   // (diagnostic: reads the device's counters back)
        const int st = k.back() - '0';
        *value = -1;
        if (e->filtered[st] && e->filter_info.p && e->g.nnz) {
            unsigned long long info[2] = {0, 0};
            if (hipStreamSynchronize(e->stream) != hipSuccess ||
                hipMemcpy(info, e->filter_info.p + 4 * st, sizeof info, hipMemcpyDeviceToHost) != hipSuccess)
                return GNNVC_ERR_DEVICE;
            *value = (long)(info[0] * 100ull / e->g.nnz);
        }
    }
    else if (k == "pruned_vertices_stage1" || k == "pruned_vertices_stage2") *value = e->prune[k.back() - '0'].ready ? (long)e->prune[k.back() - '0'].members : 0;
    else if (k == "pruned_entries_stage1" || k == "pruned_entries_stage2") *value = e->prune[k.back() - '0'].ready ? (long)e->prune[k.back() - '0'].kept : 0;
    else if (k == "pruned_predicted_stage1") *value = e->prune[1].ready && e->prune[1].predicted ? 1 : 0;
    else if (k == "pruned_borrowed_stage2") *value = e->borrowed[2] ? 1 : 0;
    else if (k == "pruned_from_previous_stage2") *value = e->prune[2].ready && e->prune[2].from_prev ? 1 : 0;
    else if (k == "pruned_last_ok_stage1" || k == "pruned_last_ok_stage2") {
        // did the last call of that stage use its pruned adjacency?  (waits for the stream; tests and tools)
        const int st = k.back() - '0';
        *value = 0;
        if (e->prune[st].ready || e->borrowed[st]) {
            uint32_t bad = 1;
            if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess ||
                hipMemcpy(&bad, e->prune_flags.p + st, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess)
                return GNNVC_ERR_DEVICE;
            *value = bad == 0 ? 1 : 0;
        }
    }
    else if (k == "wide_tiles_used") *value = e->wide_used ? 1 : 0;   // (did any stage since the graph was handed over run on wide tiles)
    else if (k == "table_tiles_active") *value = e->t4_ok ? 1 : 0;
    else if (k == "table_tiles_fit_stage1" || k == "table_tiles_fit_stage2") {   // did the last forward's stage run on the table?  (waits for the stream)
        *value = 0;
        if (e->t4_ok && e->t4_desc.p) {
            uint32_t fit = 0;
            gnnvc_engine *m = const_cast<gnnvc_engine *>(e);
            if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess ||
                hipMemcpy(&fit, m->t4_desc_of(k.back() - '0', e->t4_parity) + 8, sizeof fit, hipMemcpyDeviceToHost) != hipSuccess)
                return GNNVC_ERR_DEVICE;
            *value = (long)fit;
        }
    }
    else if (k == "lds_table_off") *value = e->lt_off ? 1 : 0;
    else if (k == "lds_table_bits") *value = e->lt_ready ? (long)e->lt_bits : 0;
    else if (k == "compact_gather_off_stage1" || k == "compact_gather_off_stage2") *value = e->c4_stage_off[k.back() - '0'] ? 1 : 0;
    else if (k == "compact_gather_blocks") *value = e->c4_ready ? (long)e->c4_nblocks : 0;
    else if (k == "compact_gather_last_ok" || k == "compact_gather_last_dirty" || k == "compact_gather_last_passes" ||
             k == "compact_table_written_by_producer") {
        // what the device decided at the last launch of the plan (waits for the stream; for tests and tools)
        *value = 0;
        if (e->c4_ready && e->c4_desc.p) {
            uint32_t d[8] = {0};
            if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess ||
                hipMemcpy(d, e->c4_desc.p + e->c4_last_desc, sizeof d, hipMemcpyDeviceToHost) != hipSuccess)
                return GNNVC_ERR_DEVICE;
            if (k == "compact_table_written_by_producer") *value = (d[0] && !d[6]) ? 1 : 0;   // (no compaction pass was needed)
            else *value = k == "compact_gather_last_ok" ? (d[0] ? 1 : 0) : (k == "compact_gather_last_passes" ? (long)d[0] : (long)d[5]);
        }
    }
    else if (k == "lds_table_active") *value = e->lt_ready ? 1 : 0;
    else if (k == "lds_table_last_ok") {   // did the last forward's input fit the table?  (waits for the stream; tests and tools)
        *value = 0;
        if (e->lt_ready && e->lt_bad.p) {
            uint32_t bad = 1;
            if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess ||
                hipMemcpy(&bad, e->lt_bad.p, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess)
                return GNNVC_ERR_DEVICE;
            *value = bad == 0 ? 1 : 0;
        }
    }
    else if (k == "lds_table_mapped") *value = e->lt_ready && e->lt_mapped ? 1 : 0;
    else if (k == "lds_table_blocks") *value = e->lt_ready ? (long)e->lt_blocks : 0;
    else if (k == "lds_table_chunks") *value = e->lt_ready ? (long)e->lt_chunks : 0;
    else if (k == "lds_table_steps") *value = e->lt_ready ? (long)e->lt_steps_total : 0;
    else if (k == "blocked_stage0_active") *value = e->blocked_ready ? 1 : 0;
    else if (k == "blocked_blocks") *value = e->blocked_ready ? (long)e->blk_count : 0;
    else if (k == "block_cols") *value = e->blocked_ready ? (long)e->blk_cols : 0;
    else if (k == "long_rows") *value = (long)e->n_long;
    else if (k == "plan_build_us") *value = (long)(e->plan_build_ms * 1000.0);
    else if (k == "handoff_build_us") *value = (long)(e->handoff_build_ms * 1000.0);
    else if (k == "handoff_early_us") *value = (long)(e->early_ms * 1000.0);
    else if (k == "plans_at_handoff") *value = e->opt_handoff;
    else if (k == "graph_uses") *value = (long)e->graph_uses;
    else if (k == "slice_rows") *value = e->empty_slice ? 0 : (long)(e->g.hi() - e->g.lo());
    else if (k == "slice_entries") *value = (long)e->g.nnz;
    else if (k == "giant_rows") *value = (long)e->n_giant;
    else if (k == "giant_segments") *value = e->n_giant ? (long)e->gi_maxseg : 0;
    else if (k == "giant_entries") *value = (long)e->giant_entries;
    else if (k == "giant_row_threshold") *value = e->n_giant ? (long)e->giant_thresh : 0;
    else if (k == "mfma_dense") *value = e->opt_mfma;
    else if (k == "sorted_tiles_active") *value = e->sorted_wanted ? 1 : 0;
    else if (k == "tile_waste_x100") *value = (long)(e->srt_waste * 100.0);
    else if (k == "heavy_tail_x1000") *value = (long)(e->srt_tail * 1000.0);
    else if (k == "interleaved_tiles") *value = e->interleave ? 1 : 0;
    else if (k == "long_row_threshold") *value = e->n_long ? (long)e->long_thresh : 0;
    else return GNNVC_ERR_INVALID;
    return GNNVC_OK;
}

int gnnvc_num_layers(const gnnvc_engine *e) { return e ? (int)e->layers.size() : GNNVC_ERR_INVALID; }
int gnnvc_is_fused(const gnnvc_engine *e) { return e ? (e->stages.empty() ? 0 : 1) : GNNVC_ERR_INVALID; }
int gnnvc_in_width(const gnnvc_engine *e) { return e ? e->in_width : GNNVC_ERR_INVALID; }
int gnnvc_out_width(const gnnvc_engine *e) { return e ? e->out_width : GNNVC_ERR_INVALID; }
int gnnvc_num_stages(const gnnvc_engine *e) { return e ? (int)e->stages.size() : GNNVC_ERR_INVALID; }

int gnnvc_stage_widths(const gnnvc_engine *e, int stage, int *in_width, int *out_width) {
    if (!e || stage < 0 || stage >= (int)e->stages.size()) return GNNVC_ERR_INVALID;
    if (in_width) *in_width = e->stages[stage].f;
    if (out_width) *out_width = e->stages[stage].n3;
    return GNNVC_OK;
}

// Common tail of the host hand-offs: device-side sanity checks of the arrays now in
// e->rowptr/col/w/nw, then make them the engine's graph.
// Slices of an open plan build whose rows' column entries lie inside the first `sent` entries of the column array (rp: the
// HOST copy of the row pointers the hand-off was given).
extern "C++" {
template <class RP>
static uint32_t slices_arrived(const gnnvc_engine::PlanBuild &pb, const RP *rp, uint32_t n, uint64_t sent) {
    uint32_t lo = 0, hi = pb.slices;   // largest s with rp[min(n, base + s * slice_rows)] <= sent
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo + 1) / 2;
        const uint64_t row = std::min<uint64_t>(n, (uint64_t)pb.base + (uint64_t)mid * pb.slice_rows);
        if ((uint64_t)rp[row] <= sent) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
}   // extern "C++"


// the first `sent` column entries are on their way (queued on e->stream): regroup the slices they complete, on the second stream
extern "C++" {
template <class RP>
static int handoff_progress(gnnvc_engine *e, const RP *rp, uint32_t n, uint64_t sent) {
    if (!e->early_open || (!e->lt_pb.open && !e->c4_pb.open)) return GNNVC_OK;
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t lt_to = e->lt_pb.open ? slices_arrived(e->lt_pb, rp, n, sent) : 0u;
    const uint32_t c4_to = e->c4_pb.open ? slices_arrived(e->c4_pb, rp, n, sent) : 0u;
    if ((e->lt_pb.open && lt_to > e->lt_pb.done) || (e->c4_pb.open && c4_to > e->c4_pb.done)) {
        HIP_TRY(e, hipEventRecord(e->ev_piece, e->stream));
        HIP_TRY(e, hipStreamWaitEvent(e->aux_stream, e->ev_piece, 0));
        int rc = lt_advance(e, lt_to, e->aux_stream);
        if (rc == GNNVC_OK) rc = c4_advance(e, c4_to, e->aux_stream);
        if (rc) return rc;
    }
    e->early_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return GNNVC_OK;
}
}   // extern "C++"

static int adopt_uploaded(gnnvc_engine *e, uint32_t n, uint64_t nnz) {
    if (e->early_open) {   // whatever the second stream still regroups has to be done before the plans are finished on this one
        HIP_TRY(e, hipEventRecord(e->ev_piece, e->aux_stream));
        HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_piece, 0));
    }
    {
        const GraphDev cand{n, nnz, e->rowptr.p, e->col.p, e->w.p, e->nw.p};
        uint32_t bad = 0;
        if (e->early_open) {   // (the graph was classed when its row pointers arrived: only the checks are left)
            HIP_TRY(e, gnnvc::validate_graph(cand, e->blk_flag.p, e->stream));
            HIP_TRY(e, hipMemcpyAsync(&bad, e->blk_flag.p, sizeof bad, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(e, hipStreamSynchronize(e->stream));
        } else {
            const int rc0 = classify_hand_off(e, cand, bad);
            if (rc0) return rc0;
        }
        if (bad) {
            e->have_graph = false;
            if (e->early_open) {   // plans begun from this graph's arrays: worthless
                e->early_open = false;
                reset_graph_state(e);
            }
            return fail(e, GNNVC_ERR_INVALID, "%s", (bad & 1u) ? "a column id is not a vertex of this graph (col[i] >= n)"
                                                                 : "row pointers are not monotone from 0 to nnz");
        }
    }
    const bool early = e->early_open;   // the graph was classed and its plans begun while the column array was arriving
    e->early_open = false;
    e->g = GraphDev{n, nnz, e->rowptr.p, e->col.p, e->w.p, e->nw.p};
    e->have_graph = true;
    e->empty_slice = false;
    e->der_open = false;   // (a derivation begun against the previous graph must not be committed against this one)
    int rc = reserve_features(e, n);
    if (rc) return rc;
    if (!early) {
        reset_graph_state(e);
        e->pre_armed = true;
        rc = find_long(e);
        if (rc) return rc;
    }
    return prepare_plans(e);   // ... which is why what depends on the graph alone is built here, not in a later forward
}

int gnnvc_upload_graph(gnnvc_engine *e, uint32_t n, const uint64_t *rowptr, const uint32_t *col,
                       const uint32_t *w, const uint32_t *nw) {
    if (!e) return GNNVC_ERR_INVALID;
    if (n && (!rowptr || !w || !nw)) return fail(e, GNNVC_ERR_INVALID, "null graph arrays");
    int rc = use_device(e);
    if (rc) return rc;
    if (e->multi) {
        if (n && rowptr[n] && !col) return fail(e, GNNVC_ERR_INVALID, "null column array");
        std::string err;
        rc = gnnvc::multi_upload(e->multi, n, rowptr, nullptr, col, w, nw, err);
        (void)hipSetDevice(e->device);
        if (rc) return fail(e, rc, "%s", err.c_str());
        e->have_graph = false;   // (the front itself holds no graph: the parts do)
        return reserve_multi_front(e, n);
    }
    const uint64_t nnz = n ? rowptr[n] : 0;
    if (nnz >= 0xFFFFFFFFull - GNNVC_COL_PAD)
        return fail(e, GNNVC_ERR_UNSUPPORTED, "nnz %llu does not fit 32-bit row pointers", (unsigned long long)nnz);
    if (nnz && !col) return fail(e, GNNVC_ERR_INVALID, "null column array");
    e->der_open = false;
    HIP_TRY(e, e->rowptr.reserve((size_t)n + 1));
    HIP_TRY(e, e->col.reserve(nnz + GNNVC_COL_PAD));
    HIP_TRY(e, e->w.reserve(n));
    HIP_TRY(e, e->nw.reserve(n));
    HIP_TRY(e, e->blk_flag.reserve(1));
    // stream-ordered behind earlier forwards; the uint64 row pointers are staged in the scratch
    // buffer and narrowed on the device, and the sanity checks run there too (a host pass over
    // 2e8 column ids costs more than the copy)
    HIP_TRY(e, e->scratch[0].reserve(((size_t)n + 1) * 2));
    if (n) {
        HIP_TRY(e, hipMemcpyAsync(e->scratch[0].p, rowptr, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(e, gnnvc::narrow_rowptr(e->scratch[0].p, e->rowptr.p, (size_t)n + 1, e->stream));
        HIP_TRY(e, hipMemcpyAsync(e->w.p, w, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(e, hipMemcpyAsync(e->nw.p, nw, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    } else {
        HIP_TRY(e, hipMemsetAsync(e->rowptr.p, 0, sizeof(uint32_t), e->stream));
    }
    e->early_open = e->early_declined = false;
    rc = handoff_early(e, n, nnz);   // (large graphs: classed now, plans begun — see handoff_early)
    if (rc) {
        e->early_open = false;
        reset_graph_state(e);
        return rc;
    }
    if (e->early_open && (e->lt_pb.open || e->c4_pb.open)) {
        // the column array in pieces: while piece k + 1 crosses the bus the second stream regroups the slices piece k completed
        const uint64_t piece = 16ull << 20;
        for (uint64_t at = 0; at < nnz; at += piece) {
            const uint64_t cnt = std::min(piece, nnz - at);
            HIP_TRY(e, hipMemcpyAsync(e->col.p + at, col + at, cnt * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
            rc = handoff_progress(e, rowptr, n, at + cnt);
            if (rc) return rc;
        }
    } else if (nnz) {
        HIP_TRY(e, hipMemcpyAsync(e->col.p, col, nnz * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    }
    HIP_TRY(e, hipMemsetAsync(e->col.p + nnz, 0, GNNVC_COL_PAD * sizeof(uint32_t), e->stream));
    return adopt_uploaded(e, n, nnz);
}

/* ---- staged hand-off --------------------------------------------------------
 * The reference's driver hands predict a freshly mutated graph every call (src/GNN_VC.cpp:171-192)
 * whose adjacency lives in scattered ranges of one big array (include/reduction_graph.hpp:29-36),
 * so the host has to pack it.  Packing straight into page-locked memory owned by the engine saves
 * the allocation + zero fill + page faults of a temporary CSR and the bounce copy of a pageable
 * upload, and finished pieces of the column array go out while the host packs the next. */
int gnnvc_graph_staging(gnnvc_engine *e, uint32_t n, uint64_t nnz, uint32_t **rowptr, uint32_t **col,
                        uint32_t **w, uint32_t **nw) {
    if (!e) return GNNVC_ERR_INVALID;
    if (nnz >= 0xFFFFFFFFull - GNNVC_COL_PAD)
        return fail(e, GNNVC_ERR_UNSUPPORTED, "nnz %llu does not fit 32-bit row pointers", (unsigned long long)nnz);
    int rc = use_device(e);
    if (rc) return rc;
    if (e->staging && e->staged_sent && (n != e->staged_n || nnz != e->staged_nnz))
        return fail(e, GNNVC_ERR_STATE, "staging resized after columns were sent");
    HIP_TRY(e, e->pin_rowptr.reserve((size_t)n + 1));
    HIP_TRY(e, e->pin_w.reserve(n));
    HIP_TRY(e, e->pin_nw.reserve(n));
    HIP_TRY(e, e->pin_col.reserve(nnz));
    e->staged_n = n;
    e->staged_nnz = nnz;
    if (!e->staging) {
        e->staged_sent = 0;
        e->early_open = e->early_declined = false;
    }
    e->staging = true;
    if (rowptr) *rowptr = e->pin_rowptr.p;
    if (col) *col = e->pin_col.p;
    if (w) *w = e->pin_w.p;
    if (nw) *nw = e->pin_nw.p;
    return GNNVC_OK;
}

int gnnvc_staged_columns_ready(gnnvc_engine *e, uint64_t first, uint64_t count) {
    if (!e) return GNNVC_ERR_INVALID;
    if (!e->staging) return fail(e, GNNVC_ERR_STATE, "no graph is being staged");
    if (first != e->staged_sent || first + count > e->staged_nnz)
        return fail(e, GNNVC_ERR_INVALID, "column pieces must be announced in order (expected offset %llu)",
                    (unsigned long long)e->staged_sent);
    if (!count) return GNNVC_OK;
    if (e->multi) {   // (the slices are cut when the whole graph is known: nothing leaves before the commit)
        e->staged_sent = first + count;
        return GNNVC_OK;
    }
    int rc = use_device(e);
    if (rc) return rc;
    if (!e->staged_sent) {   // first piece: the device array is about to be overwritten
        e->have_graph = false;
        e->der_open = false;
        HIP_TRY(e, e->col.reserve(e->staged_nnz + GNNVC_COL_PAD));
    }
    HIP_TRY(e, hipMemcpyAsync(e->col.p + first, e->pin_col.p + first, count * sizeof(uint32_t), hipMemcpyHostToDevice,
                              e->stream));
    e->staged_sent = first + count;
    // Large graphs: the row pointers and weights are final by now (step 1 of the protocol) — they go out, the graph is classed
    // and its plans begun, and from here on every announced piece lets the second stream regroup the slices it completes.
    if (!e->early_open && !e->early_declined && e->opt_handoff && e->staged_n &&
        (e->opt_handoff >= 2 || e->staged_nnz >= e->opt_handoff_min_nnz) && e->pin_rowptr.p[e->staged_n] == e->staged_nnz) {
        const uint32_t n = e->staged_n;
        HIP_TRY(e, e->rowptr.reserve((size_t)n + 1));
        HIP_TRY(e, e->w.reserve(n));
        HIP_TRY(e, e->nw.reserve(n));
        HIP_TRY(e, hipMemcpyAsync(e->rowptr.p, e->pin_rowptr.p, ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(e, hipMemcpyAsync(e->w.p, e->pin_w.p, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(e, hipMemcpyAsync(e->nw.p, e->pin_nw.p, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        rc = handoff_early(e, n, e->staged_nnz);
    }
    if (rc == GNNVC_OK) rc = handoff_progress(e, e->pin_rowptr.p, e->staged_n, e->staged_sent);
    if (rc) {   // (e.g. row pointers the early builders must not see: the hand-off is over, the staging memory can be reused)
        e->staging = false;
        e->staged_sent = 0;
        e->early_open = false;
        reset_graph_state(e);
    }
    return rc;
}

int gnnvc_commit_staged_graph(gnnvc_engine *e) {
    if (!e) return GNNVC_ERR_INVALID;
    if (!e->staging) return fail(e, GNNVC_ERR_STATE, "no graph is being staged");
    int rc = use_device(e);
    if (rc) return rc;
    const uint32_t n = e->staged_n;
    const uint64_t nnz = e->staged_nnz;
    e->staging = false;
    e->have_graph = false;
    if (n && e->pin_rowptr.p[n] != nnz) return fail(e, GNNVC_ERR_INVALID, "staged rowptr[n] differs from the staged nnz");
    if (e->multi) {
        e->staged_sent = 0;
        std::string err;
        rc = gnnvc::multi_upload(e->multi, n, nullptr, e->pin_rowptr.p, e->pin_col.p, e->pin_w.p, e->pin_nw.p, err);
        (void)hipSetDevice(e->device);
        if (rc) return fail(e, rc, "%s", err.c_str());
        return reserve_multi_front(e, n);
    }
    HIP_TRY(e, e->rowptr.reserve((size_t)n + 1));
    HIP_TRY(e, e->col.reserve(nnz + GNNVC_COL_PAD));
    HIP_TRY(e, e->w.reserve(n));
    HIP_TRY(e, e->nw.reserve(n));
    HIP_TRY(e, e->blk_flag.reserve(1));
    if (n && !e->early_open) {   // (an early hand-off sent them with its first piece)
        HIP_TRY(e, hipMemcpyAsync(e->rowptr.p, e->pin_rowptr.p, ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(e, hipMemcpyAsync(e->w.p, e->pin_w.p, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(e, hipMemcpyAsync(e->nw.p, e->pin_nw.p, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    } else if (!n) {
        HIP_TRY(e, hipMemsetAsync(e->rowptr.p, 0, sizeof(uint32_t), e->stream));
    }
    if (nnz > e->staged_sent)
        HIP_TRY(e, hipMemcpyAsync(e->col.p + e->staged_sent, e->pin_col.p + e->staged_sent,
                                  (nnz - e->staged_sent) * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    e->staged_sent = 0;
    HIP_TRY(e, hipMemsetAsync(e->col.p + nnz, 0, GNNVC_COL_PAD * sizeof(uint32_t), e->stream));
    return adopt_uploaded(e, n, nnz);
}

static int attach_common(gnnvc_engine *e, const GraphDev &cand) {
    {   // a column id >= n would send the gather to a wild address: check before accepting the graph — and, in the same pass of
        // the stream, learn what find_long is about to ask (classify_hand_off: one wait instead of four)
        uint32_t bad = 0;
        const int rc0 = classify_hand_off(e, cand, bad);
        if (rc0) return rc0;
        if (bad) {
            e->have_graph = false;
            return fail(e, GNNVC_ERR_INVALID, "%s", (bad & 1u) ? "a column id is not a vertex of this graph (col[i] >= n)"
                                                                 : "row pointers are not monotone from 0 to nnz");
        }
    }
    e->g = cand;
    e->have_graph = true;
    e->empty_slice = false;
    e->der_open = false;
    int rc = reserve_features(e, cand.n);
    if (rc) return rc;
    // (a staged hand-off that opened the early builders may have been abandoned or refused before its commit: nothing of it —
    // open plan builds with the OLD graph's geometry least of all — may reach prepare_plans with this graph)
    reset_graph_state(e);
    e->early_open = e->early_declined = false;
    e->pre_armed = true;
    rc = find_long(e);
    if (rc) return rc;
    return prepare_plans(e);   // ... which is why what depends on the graph alone is built here, not in a later forward
}

/* ---- the next graph derived from the resident one ---------------------------------------------
 * See the k_derive_* kernels: the device keeps the CSR of the last hand-off; the caller says which old row every vertex
 * of the next graph was (or that it is new) and how long its list is now, learns how many trailing entries per row the
 * device cannot derive, ships exactly those, and the engine assembles the next CSR in place of an upload of all of it. */
int gnnvc_derive_graph_begin(gnnvc_engine *e, uint32_t n_new, const uint32_t *old_row, const uint32_t *rowptr_new, uint32_t *tail) {
    if (!e) return GNNVC_ERR_INVALID;
    NOT_ON_MULTI(e, "gnnvc_derive_graph_begin");
    e->der_open = false;
    if (!e->have_graph || e->g.rowptr != e->rowptr.p || e->g.col != e->col.p || e->g.sliced() || e->empty_slice)
        return fail(e, GNNVC_ERR_STATE, "no engine-owned whole graph is resident (hand one over with gnnvc_upload_graph or the staged calls first)");
    if (n_new && (!old_row || !rowptr_new || !tail)) return fail(e, GNNVC_ERR_INVALID, "null arrays");
    int rc = use_device(e);
    if (rc) return rc;
    const uint64_t nnz_new = n_new ? rowptr_new[n_new] : 0;
    if (n_new && rowptr_new[0] != 0) return fail(e, GNNVC_ERR_INVALID, "rowptr_new[0] != 0");
    if (nnz_new >= 0xFFFFFFFFull - GNNVC_COL_PAD) return fail(e, GNNVC_ERR_UNSUPPORTED, "nnz too large");
    HIP_TRY(e, e->der_old_row.reserve(n_new));
    HIP_TRY(e, e->rowptr2.reserve((size_t)n_new + 1));
    HIP_TRY(e, e->der_new_of.reserve(std::max<uint32_t>(e->g.n, 1u)));
    HIP_TRY(e, e->der_tail.reserve(n_new));
    HIP_TRY(e, e->blk_flag.reserve(1));
    if (n_new) {
        HIP_TRY(e, hipMemcpyAsync(e->der_old_row.p, old_row, (size_t)n_new * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(e, hipMemcpyAsync(e->rowptr2.p, rowptr_new, ((size_t)n_new + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    } else {
        HIP_TRY(e, hipMemsetAsync(e->rowptr2.p, 0, sizeof(uint32_t), e->stream));
    }
    HIP_TRY(e, gnnvc::derive_tails(e->g, e->der_old_row.p, n_new, e->rowptr2.p, e->der_new_of.p, e->der_tail.p, e->blk_flag.p, e->stream));
    uint32_t bad = 0;
    if (n_new) HIP_TRY(e, hipMemcpyAsync(tail, e->der_tail.p, (size_t)n_new * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(&bad, e->blk_flag.p, sizeof bad, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (bad)
        return fail(e, GNNVC_ERR_INVALID, "%s", (bad & 1u) ? "old_row[] names a row the resident graph does not have"
                                                 : (bad & 2u) ? "two vertices of the next graph claim the same old row"
                                                              : "a row has more surviving old neighbours than its new degree: not derived from the resident graph");
    e->der_tail_host.assign(tail, tail + n_new);
    uint64_t total = 0;
    for (uint32_t u = 0; u < n_new; ++u) total += tail[u];
    e->der_n_new = n_new;
    e->der_nnz_new = nnz_new;
    e->der_tail_total = total;
    e->der_open = true;
    return GNNVC_OK;
}

int gnnvc_derive_graph_commit(gnnvc_engine *e, const uint32_t *tail_cols, uint64_t n_tail, const uint32_t *w, const uint32_t *nw) {
    if (!e) return GNNVC_ERR_INVALID;
    NOT_ON_MULTI(e, "gnnvc_derive_graph_commit");
    if (!e->der_open) return fail(e, GNNVC_ERR_STATE, "gnnvc_derive_graph_begin first");
    e->der_open = false;
    e->early_open = e->early_declined = false;   // (an abandoned staged hand-off must not pass for this graph's)
    if (!e->have_graph || e->g.rowptr != e->rowptr.p || e->g.col != e->col.p || e->g.sliced() || e->empty_slice)
        return fail(e, GNNVC_ERR_STATE, "the resident graph changed since gnnvc_derive_graph_begin");
    const uint32_t n = e->der_n_new;
    if (n_tail != e->der_tail_total) return fail(e, GNNVC_ERR_INVALID, "expected %llu tail entries, got %llu",
                                                 (unsigned long long)e->der_tail_total, (unsigned long long)n_tail);
    if ((n_tail && !tail_cols) || (n && (!w || !nw))) return fail(e, GNNVC_ERR_INVALID, "null arrays");
    int rc = use_device(e);
    if (rc) return rc;
    std::vector<uint32_t> tptr((size_t)n + 1, 0);
    for (uint32_t u = 0; u < n; ++u) tptr[u + 1] = tptr[u] + e->der_tail_host[u];
    HIP_TRY(e, e->der_tailptr.reserve((size_t)n + 1));
    HIP_TRY(e, e->der_tailcols.reserve(std::max<uint64_t>(n_tail, 1)));
    HIP_TRY(e, e->col2.reserve(e->der_nnz_new + GNNVC_COL_PAD));
    HIP_TRY(e, hipMemcpy(e->der_tailptr.p, tptr.data(), tptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (n_tail) HIP_TRY(e, hipMemcpyAsync(e->der_tailcols.p, tail_cols, n_tail * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, gnnvc::derive_fill(e->g, e->der_old_row.p, e->der_new_of.p, e->rowptr2.p, e->der_tailptr.p, e->der_tailcols.p, n,
                                  e->col2.p, e->stream));
    HIP_TRY(e, hipMemsetAsync(e->col2.p + e->der_nnz_new, 0, GNNVC_COL_PAD * sizeof(uint32_t), e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));   // (the old arrays are about to change hands)
    std::swap(e->rowptr, e->rowptr2);
    std::swap(e->col, e->col2);
    e->have_graph = false;
    HIP_TRY(e, e->w.reserve(n));
    HIP_TRY(e, e->nw.reserve(n));
    if (n) {
        HIP_TRY(e, hipMemcpyAsync(e->w.p, w, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(e, hipMemcpyAsync(e->nw.p, nw, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    }
    return adopt_uploaded(e, n, e->der_nnz_new);
}

int gnnvc_graph_row_hashes(gnnvc_engine *e, uint64_t *hashes) {
    if (!e) return GNNVC_ERR_INVALID;
    NOT_ON_MULTI(e, "gnnvc_graph_row_hashes");
    if (!e->have_graph || e->empty_slice) return fail(e, GNNVC_ERR_STATE, "no graph attached");
    const uint32_t rows = e->g.hi() - e->g.lo();
    if (!rows) return GNNVC_OK;
    if (!hashes) return fail(e, GNNVC_ERR_INVALID, "null output");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, e->hash_buf.reserve(rows));
    HIP_TRY(e, gnnvc::row_hashes(e->g, e->hash_buf.p, e->stream));
    HIP_TRY(e, hipMemcpyAsync(hashes, e->hash_buf.p, (size_t)rows * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return GNNVC_OK;
}

int gnnvc_attach_graph_device(gnnvc_engine *e, uint32_t n, uint64_t nnz, const uint32_t *d_rowptr,
                              const uint32_t *d_col, const uint32_t *d_w, const uint32_t *d_nw) {
    if (!e) return GNNVC_ERR_INVALID;
    NOT_ON_MULTI(e, "gnnvc_attach_graph_device");
    if (n && (!d_rowptr || !d_col || !d_w || !d_nw)) return fail(e, GNNVC_ERR_INVALID, "null device graph arrays");
    if (nnz >= 0xFFFFFFFFull - GNNVC_COL_PAD) return fail(e, GNNVC_ERR_UNSUPPORTED, "nnz too large");
    int rc = use_device(e);
    if (rc) return rc;
    return attach_common(e, GraphDev{n, nnz, d_rowptr, d_col, d_w, d_nw});
}

int gnnvc_attach_graph_slice(gnnvc_engine *e, uint32_t n_global, uint32_t row_lo, uint32_t row_hi, uint64_t nnz_local,
                             const uint32_t *d_rowptr_local, const uint32_t *d_col_local, const uint32_t *d_w_local,
                             const uint32_t *d_nw_local) {
    if (!e) return GNNVC_ERR_INVALID;
    NOT_ON_MULTI(e, "gnnvc_attach_graph_slice");
    if (row_lo > row_hi || row_hi > n_global) return fail(e, GNNVC_ERR_INVALID, "slice [%u, %u) outside a graph of %u vertices", row_lo, row_hi, n_global);
    if (!d_rowptr_local || !d_col_local || (row_hi > row_lo && (!d_w_local || !d_nw_local)))
        return fail(e, GNNVC_ERR_INVALID, "null device graph arrays");
    if (nnz_local >= 0xFFFFFFFFull - GNNVC_COL_PAD) return fail(e, GNNVC_ERR_UNSUPPORTED, "nnz too large");
    int rc = use_device(e);
    if (rc) return rc;
    // the kernels index rowptr / w / nw by GLOBAL row id: bias the slice's arrays so that they can (only rows of the
    // slice are ever touched — every whole-graph pass of the engine walks [row_base, row_end))
    GraphDev cand{n_global, nnz_local, d_rowptr_local - row_lo, d_col_local, d_w_local - row_lo, d_nw_local - row_lo};
    cand.row_base = row_lo;
    cand.row_end = row_hi ? row_hi : 0;
    if (row_lo == 0 && row_hi == n_global) cand.row_end = 0;   // the whole graph after all
    e->der_open = false;
    if (row_hi == 0) {   // an empty slice at the front: nothing to compute, nothing to index
        e->g = GraphDev{n_global, 0, d_rowptr_local, d_col_local, d_w_local, d_nw_local};
        e->g.row_base = 0;
        e->g.row_end = 0;
        e->have_graph = true;
        e->n_long = e->n_giant = 0;
        e->empty_slice = true;
        return GNNVC_OK;
    }
    e->empty_slice = row_hi == row_lo;
    if (e->empty_slice) {
        e->g = cand;
        e->g.row_end = row_hi;
        e->have_graph = true;
        e->n_long = e->n_giant = 0;
        return GNNVC_OK;
    }
    return attach_common(e, cand);
}

int gnnvc_stage_forward_device(gnnvc_engine *e, int stage, uint32_t row_lo, uint32_t row_hi,
                               const float *d_in, float *d_out, float *d_logits) {
    if (!e) return GNNVC_ERR_INVALID;
    NOT_ON_MULTI(e, "gnnvc_stage_forward_device");
    if (!e->have_graph) return fail(e, GNNVC_ERR_STATE, "no graph attached");
    if (stage < 0 || stage >= (int)e->stages.size()) return fail(e, GNNVC_ERR_INVALID, "stage %d out of range", stage);
    if (row_lo > row_hi || row_hi > e->g.n) return fail(e, GNNVC_ERR_INVALID, "row range [%u,%u) outside graph of %u", row_lo, row_hi, e->g.n);
    if (row_lo == row_hi) return GNNVC_OK;
    if (e->empty_slice || row_lo < e->g.lo() || row_hi > e->g.hi())
        return fail(e, GNNVC_ERR_INVALID, "rows [%u,%u) are not in the slice [%u,%u) this engine holds", row_lo, row_hi, e->g.lo(),
                    e->empty_slice ? e->g.lo() : e->g.hi());
    if (!d_in || !d_out) return fail(e, GNNVC_ERR_INVALID, "null feature buffers");
    int rc = use_device(e);
    if (rc) return rc;
    return run_stage(e, stage, row_lo, row_hi, d_in, d_out, d_logits);
}

int gnnvc_forward_device(gnnvc_engine *e, const float *d_x, float *d_scores, float *d_logits) {
    if (!e) return GNNVC_ERR_INVALID;
    if (e->multi) {   // pointers on the first device; complete (every device drained) when it returns
        if (!gnnvc::multi_has_graph(e->multi)) return fail(e, GNNVC_ERR_STATE, "no graph attached");
        if (gnnvc::multi_vertices(e->multi) == 0) return GNNVC_OK;
        if (!d_x || !d_scores) return fail(e, GNNVC_ERR_INVALID, "null feature buffers");
        int rc = use_device(e);
        if (rc) return rc;
        HIP_TRY(e, hipStreamSynchronize(e->stream));   // (whatever produced d_x on this handle's stream)
        std::string err;
        rc = gnnvc::multi_forward_device(e->multi, d_x, d_scores, d_logits, err);
        (void)hipSetDevice(e->device);
        if (rc) return fail(e, rc, "%s", err.c_str());
        return GNNVC_OK;
    }
    if (!e->have_graph) return fail(e, GNNVC_ERR_STATE, "no graph attached");
    const uint32_t n = e->g.n;
    e->ev_count = 0;
    if (e->layers.empty()) return fail(e, GNNVC_ERR_STATE, "engine was created without a model");
    if (n == 0) return GNNVC_OK;
    if (e->g.sliced() || e->empty_slice)
        return fail(e, GNNVC_ERR_STATE, "this engine holds a slice of the graph: run it stage by stage (gnnvc_stage_forward_device)");
    if (!d_x || !d_scores) return fail(e, GNNVC_ERR_INVALID, "null feature buffers");
    int rc = use_device(e);
    if (rc) return rc;
    if (e->stages.empty()) {
        rc = ensure_events(e, 2);
        if (rc) return rc;
        if (e->opt_timing >= 1) HIP_TRY(e, hipEventRecord(e->ev[0], e->stream));
        rc = forward_unfused(e, d_x, d_scores, d_logits);
        if (rc) return rc;
        if (e->opt_timing >= 1) HIP_TRY(e, hipEventRecord(e->ev[1], e->stream));
        e->ev_count = e->opt_timing >= 1 ? 2 : 0;
        e->ev_stages = false;
        return GNNVC_OK;
    }
    const size_t ns = e->stages.size();
    rc = ensure_events(e, ns + 1);
    if (rc) return rc;
    const float *cur = d_x;   // (the pad rows of e->h were zeroed when the graph was handed over: reserve_features)
    e->c4_fused_for = -1;
    e->c4_prepared_stage = -1;
    if (e->fit_pending && hipEventQuery(e->ev_fit) == hipSuccess) {   // the previous forward's verdicts have arrived
        e->fit_pending = false;
        const uint32_t before = e->lt_unfit_runs + e->t4_unfit_runs + e->c4_unfit_runs[1] + e->c4_unfit_runs[2];
        const bool seen1 = e->t4_fit_seen[1], seen2 = e->t4_fit_seen[2];
        if (e->lt_used && e->lt_ready) {
            e->lt_unfit_runs = e->fit_pin.p[2] != 0u ? e->lt_unfit_runs + 1 : 0u;
            if (e->lt_unfit_runs >= (e->lt_mapped ? 1u : 3u)) e->lt_off = true;
        }
        if (e->t4_used && e->t4_ok) {
            // table tiles: a graph whose first 16-wide stage keeps missing (more than four live columns: sparse graphs) stops paying
            // for the counters, the choice and the launches that leave at once — the first forward on a graph always misses (nobody
            // has chosen columns yet), hence four in a row
            e->t4_unfit_runs = e->fit_pin.p[3] == 0u ? e->t4_unfit_runs + 1 : 0u;
            if (e->t4_unfit_runs >= 4u) e->t4_ok = false;
            // a stage whose table fit the last forward seen gets no gathering kernel behind it (k_stage_t4's solo mode copes with
            // the rare forward that does not fit after all)
            e->t4_fit_seen[1] = e->fit_pin.p[3] != 0u;
            e->t4_fit_seen[2] = e->fit_pin.p[4] != 0u;
        }
        for (int s = 1; s <= 2; ++s) {
            if (!e->fit_used[s]) continue;
            // (a stage whose statistics were to come from a producer that itself fell back had no chance: not its miss)
            if (s == 2 && e->fit_used[1] && e->fit_pin.p[0] == 0u) continue;
            e->c4_unfit_runs[s] = e->fit_pin.p[s - 1] == 0u ? e->c4_unfit_runs[s] + 1 : 0u;
            if (e->c4_unfit_runs[s] >= 3u) e->c4_stage_off[s] = true;
        }
        const uint32_t after = e->lt_unfit_runs + e->t4_unfit_runs + e->c4_unfit_runs[1] + e->c4_unfit_runs[2];
        const bool calm = after == 0u && before == 0u && seen1 == e->t4_fit_seen[1] && seen2 == e->t4_fit_seen[2];
        e->fit_calm = calm ? e->fit_calm + 1u : 0u;
    }
    if (!e->fit_pending) {
        for (int s = 0; s < 4; ++s) e->fit_used[s] = false;
        e->lt_used = false;
        e->t4_used = false;
    }
    struct SinkGuard {   // the thread-local sink never outlives this call, whichever way it returns
        ~SinkGuard() { gnnvc::set_kernel_trace(nullptr); }
    } sink_guard;
    if (e->opt_ktrace && e->ktrace.used < 16384) {   // records pile up over forwards until gnnvc_kernel_trace reads them
        e->ktrace.stream = e->stream;
        gnnvc::set_kernel_trace(&e->ktrace);
    }
    // Table tiles are offered from a graph's SECOND forward on — or from its first, when the engine carries a choice of columns
    // over from its previous graph: a fresh engine's first forward on a graph has no table to gather from, and the counting, the
    // choice and the launches that leave at once would only cost the caller who scores the graph once (ER-100K: 0.14 vs 0.11 ms).
    e->t4_now = e->t4_ok && (e->t4_choice_live || e->graph_uses >= 1);
    // Events (option "forward_timing"): none by default — a record costs the stream ~1.8 us, four of them were 5.5 us of a small
    // graph's 31 - 82 us forward (scratch/experiments/r4_gaps.sh) — 1 = the forward's first and last, 2 = one per stage as well.
    // (ev[0] is also what the first-forward plan build on the second stream waits for, below.)
    const bool build_under_stage0 = ns >= 2 && !e->c4_tried && !e->c4_range_mode && e->aux_stream && e->n_long == 0 && !e->sorted_wanted &&
                                    e->opt_compact_first_entries && e->g.nnz >= e->opt_compact_first_entries;
    if (e->opt_timing >= 1 || build_under_stage0) HIP_TRY(e, hipEventRecord(e->ev[0], e->stream));
    if (e->opt_poison)   // (tests, fuzz: rows 0 .. n - 1 of both feature buffers; the pad row n stays zero)
        for (size_t b = 0; b + 1 < ns && b < 2; ++b) HIP_TRY(e, hipMemsetAsync(e->h[b].p, 0xFF, (size_t)n * 16 * sizeof(float), e->stream));
    for (size_t s = 0; s < ns; ++s) {
        const bool last = s + 1 == ns;
        float *dst = last ? d_scores : e->h[s & 1].p;
        rc = run_stage(e, (int)s, 0, n, cur, dst, last ? d_logits : nullptr, /*in_forward=*/true);
        if (rc) break;
        if ((e->opt_timing >= 2 || (last && e->opt_timing == 1)) && hipEventRecord(e->ev[s + 1], e->stream) != hipSuccess) { rc = fail(e, GNNVC_ERR_DEVICE, "hipEventRecord failed"); break; }
        if (s == 0 && build_under_stage0 && !e->c4_tried) {
            // A large graph's first forward: the compact-table plan of the stages to come depends on the graph alone — it is
            // built now, on the second stream, under the kernels of stage 0 that were just queued (the build synchronises with
            // its own stream only; it is complete when it returns).
            // (behind whatever the caller had queued before this forward — it may still be writing the graph's arrays)
            HIP_TRY(e, hipStreamWaitEvent(e->aux_stream, e->ev[0], 0));
            hipStream_t main_stream = e->stream;
            e->stream = e->aux_stream;
            rc = build_compact(e);
            e->stream = main_stream;
            if (rc) break;
        }
        cur = dst;
    }
    gnnvc::set_kernel_trace(nullptr);
    if (rc) return rc;
    if (e->t4_now) {   // (the descriptors this forward's table tiles wrote are the next forward's specs)
        e->t4_parity ^= 1u;
        e->t4_choice_live = true;
    }
    e->ev_count = e->opt_timing >= 1 ? (int)ns + 1 : 0;
    e->ev_stages = e->opt_timing >= 2;
    const bool c4_verdicts = (e->fit_used[1] || e->fit_used[2]) && e->c4_ready && e->c4_desc.p;
    const bool lt_verdict = e->lt_used && e->lt_ready && e->lt_bad.p;
    const bool t4_verdict = e->t4_now && e->t4_desc.p;
    if (t4_verdict && !e->fit_pending) e->t4_used = true;
    if (!e->fit_pending && (c4_verdicts || lt_verdict || t4_verdict)) {   // this forward's verdicts, written out behind it
        // ONE small kernel stores the words into page-locked host memory (round 4: they were up to five 4-byte hipMemcpyAsync, ~20 us
        // of a small graph's forward + wait, scratch/experiments/r4_gaps.sh), and once four verdicts in a row have changed nothing
        // only every eighth forward asks — a verdict steers which kernels the NEXT forwards launch, never a result: every plan
        // proves its input on the device in every call.
        const bool ask = e->fit_calm < 4u || (++e->fit_skip % e->opt_verdict_period) == 0u;
        if (!ask) {
            if (!c4_verdicts) for (int s = 1; s <= 2; ++s) e->fit_used[s] = false;
            return GNNVC_OK;
        }
        if (!e->ev_fit) HIP_TRY(e, hipEventCreateWithFlags(&e->ev_fit, hipEventDisableTiming));
        if (!e->fit_dev) {
            HIP_TRY(e, e->fit_pin.reserve(8));
            HIP_TRY(e, hipHostGetDevicePointer(reinterpret_cast<void **>(&e->fit_dev), e->fit_pin.p, 0));
        }
        gnnvc::VerdictWords vw;
        if (c4_verdicts)
            for (int s = 1; s <= 2; ++s) vw.src[s - 1] = e->c4_desc.p + gnnvc_engine::kDescWords * (s - 1);
        else
            for (int s = 1; s <= 2; ++s) e->fit_used[s] = false;
        if (lt_verdict) vw.src[2] = e->lt_bad.p;
        if (t4_verdict)   // (the parity has flipped: the descriptors this forward wrote are the current ones)
            for (int s = 1; s <= 2; ++s) vw.src[2 + s] = e->t4_desc_of(s, e->t4_parity) + 8;
        HIP_TRY(e, gnnvc::write_verdicts(vw, e->fit_dev, e->stream));
        HIP_TRY(e, hipEventRecord(e->ev_fit, e->stream));
        e->fit_pending = true;
    }
    return GNNVC_OK;
}

int gnnvc_stage_input_ready(gnnvc_engine *e, int stage, const float *d_in, uint32_t row_lo, uint32_t row_hi) {
    if (!e) return GNNVC_ERR_INVALID;
    NOT_ON_MULTI(e, "gnnvc_stage_input_ready");
    if (!e->have_graph) return fail(e, GNNVC_ERR_STATE, "no graph attached");
    if (e->stages.empty()) return fail(e, GNNVC_ERR_UNSUPPORTED, "model is not fused into stages");
    if (stage < 1 || stage >= (int)e->stages.size()) return fail(e, GNNVC_ERR_INVALID, "stage %d has no 16-wide input", stage);
    if (row_lo > row_hi || row_hi > e->g.n) return fail(e, GNNVC_ERR_INVALID, "row range [%u, %u) outside the graph", row_lo, row_hi);
    e->c4_prepared_stage = -1;
    if (e->g.n == 0 || row_lo == row_hi || e->empty_slice) return GNNVC_OK;
    if (!d_in) return fail(e, GNNVC_ERR_INVALID, "null feature buffer");
    int rc = use_device(e);
    if (rc) return rc;
    const bool same_range = e->c4_range_mode && e->c4_tried && e->c4_base == row_lo && e->c4_end == row_hi;
    e->c4_range_mode = true;
    if (!same_range) {
        rc = build_compact(e, row_lo, row_hi);
        if (rc) return rc;
        if (!e->c4_ready) {   // remember what was tried, so that the next forward does not try again
            e->c4_base = row_lo;
            e->c4_end = row_hi;
        }
    }
    if (!e->c4_ready || e->n_long > 0) return GNNVC_OK;   // the plan does not apply to this graph: nothing to prepare
    uint32_t *desc = e->c4_desc.p + gnnvc_engine::kDescWords * (stage - 1);
    e->c4_last_desc = gnnvc_engine::kDescWords * (stage - 1);
    HIP_TRY(e, gnnvc::column_counts(d_in, e->g.n, e->c4_counts.p, e->stream));
    HIP_TRY(e, gnnvc::launch_compact_gather(e->g, compact_plan(e), d_in, e->c4_counts.p, 1, desc, e->c4_table.p, e->c4_acc.p, e->c4_base,
                                            e->c4_end, e->c4_dirty.p, e->c4_dirty_cap, e->c4_agg16.p, e->stream, /*what=*/1));
    e->c4_prepared_stage = stage;
    e->c4_prepared_in = d_in;
    return GNNVC_OK;
}

int gnnvc_forward(gnnvc_engine *e, const float *x, float *scores, float *logits) {
    if (!e) return GNNVC_ERR_INVALID;
    if (e->multi) {
        if (!gnnvc::multi_has_graph(e->multi)) return fail(e, GNNVC_ERR_STATE, "no graph attached");
        const uint32_t n = gnnvc::multi_vertices(e->multi);
        if (n == 0) return GNNVC_OK;
        if (!x || !scores) return fail(e, GNNVC_ERR_INVALID, "null host buffers");
        int rc = use_device(e);
        if (rc) return rc;
        const size_t bytes = (size_t)n * sizeof(float);
        HIP_TRY(e, hipMemcpyAsync(e->x.p, x, bytes, hipMemcpyHostToDevice, e->stream));
        const bool want_logits = logits && e->ends_in_sigmoid;
        rc = gnnvc_forward_device(e, e->x.p, e->scores.p, want_logits ? e->logits.p : nullptr);
        if (rc) return rc;
        HIP_TRY(e, hipMemcpyAsync(scores, e->scores.p, bytes, hipMemcpyDeviceToHost, e->stream));
        if (want_logits) HIP_TRY(e, hipMemcpyAsync(logits, e->logits.p, bytes, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        return GNNVC_OK;
    }
    if (!e->have_graph) return fail(e, GNNVC_ERR_STATE, "no graph attached");
    const uint32_t n = e->g.n;
    if (e->layers.empty()) return fail(e, GNNVC_ERR_STATE, "engine was created without a model");
    if (n == 0) return GNNVC_OK;
    // (before the copy below: a sliced engine has no feature buffers of its own — e->x may be null or sized for an earlier graph)
    if (e->g.sliced() || e->empty_slice)
        return fail(e, GNNVC_ERR_STATE, "this engine holds a slice of the graph: run it stage by stage (gnnvc_stage_forward_device)");
    if (!x || !scores) return fail(e, GNNVC_ERR_INVALID, "null host buffers");
    int rc = use_device(e);
    if (rc) return rc;
    if (e->x.cap < ((size_t)n + 1) * (size_t)e->in_width) return fail(e, GNNVC_ERR_STATE, "feature buffers are not sized for this graph");
    const size_t in_b = (size_t)n * e->in_width * sizeof(float);
    const size_t out_b = (size_t)n * e->out_width * sizeof(float);
    HIP_TRY(e, hipMemcpyAsync(e->x.p, x, in_b, hipMemcpyHostToDevice, e->stream));
    const bool want_logits = logits && e->ends_in_sigmoid;
    rc = gnnvc_forward_device(e, e->x.p, e->scores.p, want_logits ? e->logits.p : nullptr);
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(scores, e->scores.p, out_b, hipMemcpyDeviceToHost, e->stream));
    if (want_logits)
        HIP_TRY(e, hipMemcpyAsync(logits, e->logits.p, out_b, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return GNNVC_OK;
}

int gnnvc_reduction_flags(gnnvc_engine *e, uint32_t max_degree, uint8_t *flags) {
    if (!e) return GNNVC_ERR_INVALID;
    NOT_ON_MULTI(e, "gnnvc_reduction_flags");
    if (!e->have_graph) return fail(e, GNNVC_ERR_STATE, "no graph attached");
    const uint32_t n = e->g.n;
    if (n == 0) return GNNVC_OK;
    if (e->g.sliced() || e->empty_slice) return fail(e, GNNVC_ERR_STATE, "this engine holds a slice of the graph");
    if (!flags) return fail(e, GNNVC_ERR_INVALID, "null flags buffer");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, e->scratch[1].reserve((size_t)n / 4 + 2));   // n bytes
    uint8_t *d = reinterpret_cast<uint8_t *>(e->scratch[1].p);
    HIP_TRY(e, gnnvc::launch_reduction_flags(e->g, max_degree, d, e->stream));
    HIP_TRY(e, hipMemcpyAsync(flags, d, n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return GNNVC_OK;
}

// ---- feature-row codec of the inter-GPU exchange (device pointers, asynchronous except live_columns)
int gnnvc_live_columns(gnnvc_engine *e, const float *d_feat, uint32_t rows, uint32_t width, uint32_t *mask) {
    if (!e || !mask) return GNNVC_ERR_INVALID;
    if (width != 16) return fail(e, GNNVC_ERR_UNSUPPORTED, "the row codec handles 16-column feature rows");
    *mask = 0;
    if (!rows) return GNNVC_OK;
    if (!d_feat) return fail(e, GNNVC_ERR_INVALID, "null feature buffer");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, e->blk_flag.reserve(1));
    HIP_TRY(e, e->pin_small.reserve(16));
    HIP_TRY(e, gnnvc::live_columns(d_feat, rows, e->blk_flag.p, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->pin_small.p, e->blk_flag.p, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    *mask = e->pin_small.p[0];
    return GNNVC_OK;
}

int gnnvc_column_counts(gnnvc_engine *e, const float *d_feat, uint32_t rows, uint32_t width, uint64_t *counts) {
    if (!e || !counts) return GNNVC_ERR_INVALID;
    if (width != 16) return fail(e, GNNVC_ERR_UNSUPPORTED, "the row codec handles 16-column feature rows");
    for (int c = 0; c < 16; ++c) counts[c] = 0;
    if (!rows) return GNNVC_OK;
    if (!d_feat) return fail(e, GNNVC_ERR_INVALID, "null feature buffer");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, e->srt_sum.reserve(16));
    HIP_TRY(e, e->pin_small.reserve(64));
    HIP_TRY(e, gnnvc::column_counts(d_feat, rows, e->srt_sum.p, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->pin_small.p, e->srt_sum.p, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    std::memcpy(counts, e->pin_small.p, 16 * sizeof(uint64_t));
    return GNNVC_OK;
}

static int codec_args(gnnvc_engine *e, uint32_t width, uint32_t row_lo, uint32_t row_hi, uint32_t mask, uint32_t kp) {
    if (width != 16) return fail(e, GNNVC_ERR_UNSUPPORTED, "the row codec handles 16-column feature rows");
    if (row_lo > row_hi) return fail(e, GNNVC_ERR_INVALID, "row range [%u, %u)", row_lo, row_hi);
    if (mask >> 16) return fail(e, GNNVC_ERR_INVALID, "column mask has bits beyond the row width");
    if (kp < 4 || kp > 16 || kp % 4 || (uint32_t)__builtin_popcount(mask) > kp)
        return fail(e, GNNVC_ERR_INVALID, "packed width %u cannot hold %d live columns", kp, __builtin_popcount(mask));
    return GNNVC_OK;
}

int gnnvc_pack_rows(gnnvc_engine *e, const float *d_feat, uint32_t width, uint32_t row_lo, uint32_t row_hi,
                    uint32_t mask, uint32_t kp, float *d_dense, uint32_t *d_exc, uint32_t exc_cap, uint32_t *d_flag) {
    if (!e) return GNNVC_ERR_INVALID;
    int rc = codec_args(e, width, row_lo, row_hi, mask, kp);
    if (rc) return rc;
    if (!d_flag || (row_lo != row_hi && (!d_feat || !d_dense))) return fail(e, GNNVC_ERR_INVALID, "null device buffers");
    rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, gnnvc::pack_rows(d_feat, row_lo, row_hi, mask, kp, d_dense, d_exc, d_exc ? exc_cap : 0, d_flag, e->stream));
    return GNNVC_OK;
}

int gnnvc_unpack_rows(gnnvc_engine *e, const float *d_dense, const uint32_t *d_exc, uint32_t exc_cap, uint32_t width,
                      uint32_t row_lo, uint32_t row_hi, uint32_t mask, uint32_t kp, float *d_feat) {
    if (!e) return GNNVC_ERR_INVALID;
    int rc = codec_args(e, width, row_lo, row_hi, mask, kp);
    if (rc) return rc;
    if (row_lo == row_hi) return GNNVC_OK;
    if (!d_feat || !d_dense) return fail(e, GNNVC_ERR_INVALID, "null device buffers");
    rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, gnnvc::unpack_rows(d_dense, d_exc, d_exc ? exc_cap : 0, row_lo, row_hi, mask, kp, d_feat, e->stream));
    return GNNVC_OK;
}

int gnnvc_unpack_gathered(gnnvc_engine *e, const float *d_buf, uint32_t world, uint32_t skip_rank, uint64_t piece_words,
                          uint32_t dense_rows, uint32_t exc_cap, uint32_t width, uint32_t rows_per_rank, uint32_t row_off,
                          uint32_t rows, uint32_t n, uint32_t mask, uint32_t kp, float *d_feat) {
    if (!e) return GNNVC_ERR_INVALID;
    int rc = codec_args(e, width, 0, rows, mask, kp);
    if (rc) return rc;
    if (!world || !rows) return GNNVC_OK;
    if (!d_buf || !d_feat) return fail(e, GNNVC_ERR_INVALID, "null device buffers");
    if (rows > dense_rows || piece_words < (uint64_t)dense_rows * kp + (exc_cap ? 4 + 4ull * exc_cap : 0) ||
        (uint64_t)row_off + rows > rows_per_rank)
        return fail(e, GNNVC_ERR_INVALID, "piece geometry: %u rows at offset %u of %u per rank, dense part %u rows, %llu words",
                    rows, row_off, rows_per_rank, dense_rows, (unsigned long long)piece_words);
    rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, gnnvc::unpack_gathered(d_buf, world, skip_rank, (size_t)piece_words, dense_rows, exc_cap, rows_per_rank, row_off,
                                      rows, n, mask, kp, d_feat, e->stream));
    return GNNVC_OK;
}

int gnnvc_push_piece(gnnvc_engine *e, const float *d_region, uint32_t rows, uint32_t kp, uint32_t exc_cap, uint32_t n_dst,
                     float *const *d_dst, void *hip_stream) {
    if (!e) return GNNVC_ERR_INVALID;
    if (!n_dst) return GNNVC_OK;
    if (n_dst > 64 || kp < 4 || kp > 16 || kp % 4) return fail(e, GNNVC_ERR_INVALID, "push of %u destinations, packed width %u", n_dst, kp);
    if (!d_region || !d_dst) return fail(e, GNNVC_ERR_INVALID, "null device buffers");
    for (uint32_t i = 0; i < n_dst; ++i)
        if (!d_dst[i]) return fail(e, GNNVC_ERR_INVALID, "null destination %u", i);
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, gnnvc::push_piece(d_region, rows, kp, exc_cap, n_dst, d_dst, hip_stream ? static_cast<hipStream_t>(hip_stream) : e->stream));
    return GNNVC_OK;
}

int gnnvc_unpack_pieces(gnnvc_engine *e, const gnnvc_piece *pieces, uint32_t n_pieces, uint32_t exc_cap, uint32_t width, uint32_t mask,
                        uint32_t kp, float *d_feat) {
    if (!e) return GNNVC_ERR_INVALID;
    int rc = codec_args(e, width, 0, 0, mask, kp);
    if (rc) return rc;
    if (!n_pieces) return GNNVC_OK;
    if (n_pieces > 64 || !pieces || !d_feat) return fail(e, GNNVC_ERR_INVALID, "%u pieces (1 .. 64), null buffers", n_pieces);
    gnnvc::UnpackPiece up[64];
    for (uint32_t i = 0; i < n_pieces; ++i) {
        if (pieces[i].row_lo > pieces[i].row_hi || (pieces[i].row_lo != pieces[i].row_hi && !pieces[i].d_region))
            return fail(e, GNNVC_ERR_INVALID, "piece %u: rows [%u, %u)", i, pieces[i].row_lo, pieces[i].row_hi);
        up[i] = gnnvc::UnpackPiece{pieces[i].d_region, pieces[i].row_lo, pieces[i].row_hi};
    }
    rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, gnnvc::unpack_pieces(up, n_pieces, exc_cap, mask, kp, d_feat, e->stream));
    return GNNVC_OK;
}

int gnnvc_score_keys(gnnvc_engine *e, const float *d_scores, uint32_t n, float *keys, uint8_t *above_half) {
    if (!e) return GNNVC_ERR_INVALID;
    if (!d_scores) {   // the scores the last gnnvc_forward left on the device
        const bool have = e->multi ? gnnvc::multi_has_graph(e->multi) : e->have_graph;
        const uint32_t cur = e->multi ? gnnvc::multi_vertices(e->multi) : e->g.n;
        if (!have || e->out_width != 1) return fail(e, GNNVC_ERR_STATE, "no single-column scores on the device");
        if (n != cur) return fail(e, GNNVC_ERR_INVALID, "n = %u, the current graph has %u vertices", n, cur);
        d_scores = e->scores.p;
    }
    if (!n) return GNNVC_OK;
    if (!keys || !above_half) return fail(e, GNNVC_ERR_INVALID, "null output buffers");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, e->scratch[0].reserve(n));
    HIP_TRY(e, e->scratch[1].reserve((size_t)n / 4 + 2));
    uint8_t *d_cls = reinterpret_cast<uint8_t *>(e->scratch[1].p);
    HIP_TRY(e, gnnvc::score_keys(d_scores, n, e->scratch[0].p, d_cls, e->stream));
    HIP_TRY(e, hipMemcpyAsync(keys, e->scratch[0].p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(above_half, d_cls, n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return GNNVC_OK;
}

int gnnvc_synchronize(gnnvc_engine *e) {
    if (!e) return GNNVC_ERR_INVALID;
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->multi) {
        rc = gnnvc::multi_synchronize(e->multi);
        (void)hipSetDevice(e->device);
        if (rc) return fail(e, rc, "a device of the multi-device handle failed to synchronise");
    }
    return GNNVC_OK;
}

int gnnvc_last_forward_ms(gnnvc_engine *e, float *total_ms, float *stage_ms, int max_stages) {
    if (!e) return GNNVC_ERR_INVALID;
    if (e->multi) {   // (host wall time of the last forward over all devices; no per-stage split)
        const double ms = gnnvc::multi_last_forward_ms(e->multi);
        if (ms <= 0.0) return fail(e, GNNVC_ERR_STATE, "no timed forward yet");
        if (total_ms) *total_ms = (float)ms;
        for (int s = 0; stage_ms && s < max_stages; ++s) stage_ms[s] = 0.0f;
        return GNNVC_OK;
    }
    if (e->ev_count < 2)
        return fail(e, GNNVC_ERR_STATE, e->opt_timing ? "no timed forward yet" : "no timed forward: option forward_timing is 0 (1 = total, 2 = per stage)");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, hipEventSynchronize(e->ev[e->ev_count - 1]));
    if (total_ms) HIP_TRY(e, hipEventElapsedTime(total_ms, e->ev[0], e->ev[e->ev_count - 1]));
    for (int s = 0; stage_ms && s < max_stages && s + 1 < e->ev_count; ++s) {
        if (e->ev_stages) HIP_TRY(e, hipEventElapsedTime(&stage_ms[s], e->ev[s], e->ev[s + 1]));
        else stage_ms[s] = -1.0f;   // (forward_timing 1: the stages' events were not recorded)
    }
    return GNNVC_OK;
}

int gnnvc_kernel_trace(gnnvc_engine *e, int max, const char **names, float *ms, int *count) {
    if (!e || !count) return GNNVC_ERR_INVALID;
    *count = 0;
    if (!e->opt_ktrace) return fail(e, GNNVC_ERR_STATE, "option kernel_trace is off");
    int rc = use_device(e);
    if (rc) return rc;
    const size_t used = e->ktrace.used;
    for (size_t i = 0; i < used; ++i) {
        const auto &r = e->ktrace.recs[i];
        HIP_TRY(e, hipEventSynchronize(r.b));
        if ((int)i < max) {
            float t = 0.0f;
            HIP_TRY(e, hipEventElapsedTime(&t, r.a, r.b));
            if (names) names[i] = r.name;
            if (ms) ms[i] = t;
        }
    }
    *count = (int)used;
    e->ktrace.used = 0;
    return GNNVC_OK;
}

// ---- layer-level entry points (host pointers in, host pointers out) ------------

static int run_host_op(gnnvc_engine *e, size_t in_count, const float *in, size_t out_count,
                       float *out, bool out_is_input,
                       hipError_t (*op)(gnnvc_engine *, const float *, float *, void *), void *ctx) {
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(e, e->scratch[0].reserve(in_count));
    HIP_TRY(e, e->scratch[1].reserve(out_count));
    if (in_count)
        HIP_TRY(e, hipMemcpyAsync(e->scratch[0].p, in, in_count * sizeof(float), hipMemcpyHostToDevice, e->stream));
    if (out_is_input && out_count)
        HIP_TRY(e, hipMemcpyAsync(e->scratch[1].p, out, out_count * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, op(e, e->scratch[0].p, e->scratch[1].p, ctx));
    if (out_count)
        HIP_TRY(e, hipMemcpyAsync(out, e->scratch[1].p, out_count * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return GNNVC_OK;
}

int gnnvc_graph_layer_forward(gnnvc_engine *e, uint32_t f, const float *in, float *out) {
    if (!e) return GNNVC_ERR_INVALID;
    NOT_ON_MULTI(e, "gnnvc_graph_layer_forward");
    if (!e->have_graph) return fail(e, GNNVC_ERR_STATE, "no graph attached");
    if (f == 0) return fail(e, GNNVC_ERR_INVALID, "zero feature width");
    const uint32_t n = e->g.n;
    if (n == 0) return GNNVC_OK;
    if (e->g.sliced() || e->empty_slice) return fail(e, GNNVC_ERR_STATE, "this engine holds a slice of the graph");
    if (!in || !out) return fail(e, GNNVC_ERR_INVALID, "null host buffers");
    struct Ctx { uint32_t f; } ctx{f};
    return run_host_op(e, (size_t)n * f, in, (size_t)n * (2 * f + 3), out, false,
                       [](gnnvc_engine *en, const float *di, float *dout, void *c) {
                           return gnnvc::launch_graph_layer(en->g, en->ws, ((Ctx *)c)->f, di, dout, en->stream);
                       }, &ctx);
}

int gnnvc_linear_forward(gnnvc_engine *e, uint32_t n, uint32_t k, uint32_t m, const float *in,
                         const float *W, const float *bias, float *out) {
    if (!e) return GNNVC_ERR_INVALID;
    if ((size_t)n * m == 0) return GNNVC_OK;
    if (!in || !W || !bias || !out) return fail(e, GNNVC_ERR_INVALID, "null host buffers");
    int rc = use_device(e);
    if (rc) return rc;
    DevBuf<float> dW;
    HIP_TRY(e, dW.reserve((size_t)k * m + m));
    rc = GNNVC_OK;
    hipError_t h1 = hipMemcpy(dW.p, W, (size_t)k * m * sizeof(float), hipMemcpyHostToDevice);
    hipError_t h2 = hipMemcpy(dW.p + (size_t)k * m, bias, (size_t)m * sizeof(float), hipMemcpyHostToDevice);
    if (h1 != hipSuccess || h2 != hipSuccess) {
        dW.release();
        return fail(e, GNNVC_ERR_DEVICE, "parameter upload failed");
    }
    struct Ctx { uint32_t n, k, m; const float *W; } ctx{n, k, m, dW.p};
    rc = run_host_op(e, (size_t)n * k, in, (size_t)n * m, out, false,
                     [](gnnvc_engine *en, const float *di, float *dout, void *c) {
                         auto *x = (Ctx *)c;
                         return gnnvc::launch_linear(x->n, x->k, x->m, di, x->W, x->W + (size_t)x->k * x->m, dout, en->stream);
                     }, &ctx);
    dW.release();
    return rc;
}

int gnnvc_relu_forward(gnnvc_engine *e, size_t count, const float *in, float *out) {
    if (!e) return GNNVC_ERR_INVALID;
    if (!count) return GNNVC_OK;
    if (!in || !out) return fail(e, GNNVC_ERR_INVALID, "null host buffers");
    struct Ctx { size_t n; } ctx{count};
    return run_host_op(e, count, in, count, out, false,
                       [](gnnvc_engine *en, const float *di, float *dout, void *c) {
                           return gnnvc::launch_relu(((Ctx *)c)->n, di, dout, en->stream);
                       }, &ctx);
}

int gnnvc_sigmoid_forward(gnnvc_engine *e, size_t count, const float *in, float *out) {
    if (!e) return GNNVC_ERR_INVALID;
    if (!count) return GNNVC_OK;
    if (!in || !out) return fail(e, GNNVC_ERR_INVALID, "null host buffers");
    struct Ctx { size_t n; } ctx{count};
    return run_host_op(e, count, in, count, out, false,
                       [](gnnvc_engine *en, const float *di, float *dout, void *c) {
                           return gnnvc::launch_sigmoid(((Ctx *)c)->n, di, dout, en->stream);
                       }, &ctx);
}

int gnnvc_stream_sum(gnnvc_engine *e, const float *values, uint32_t streams, uint32_t len, int mode, float *sums) {
    if (!e) return GNNVC_ERR_INVALID;
    if (!streams) return GNNVC_OK;
    if (!sums || (len && !values)) return fail(e, GNNVC_ERR_INVALID, "null host buffers");
    if (mode != 0 && mode != 2) return fail(e, GNNVC_ERR_INVALID, "mode %d (0 = a stream on several waves, 2 = on one wave; the same exact sum)", mode);
    if (len == 0) {
        for (uint32_t i = 0; i < streams; ++i) sums[i] = 0.0f;
        return GNNVC_OK;
    }
    int rc = use_device(e);
    if (rc) return rc;
    const uint32_t win = gnnvc::giant_window();
    const size_t lpad = ((size_t)len + win - 1) / win * win;
    DevBuf<float> slab, agg, segsum;
    DevBuf<uint4> meta, segmap;
    DevBuf<unsigned long long> off;
    const size_t segs = (size_t)streams * gnnvc::giant_segments(len);
    hipError_t h = slab.reserve(lpad * streams);
    if (h == hipSuccess && mode == 0) h = segsum.reserve(segs);
    if (h == hipSuccess && mode == 0) h = segmap.reserve(segs);
    if (h == hipSuccess) h = agg.reserve((size_t)streams * 16);
    if (h == hipSuccess) h = meta.reserve((size_t)streams + 1);
    if (h == hipSuccess) h = off.reserve(streams);
    if (h == hipSuccess) h = hipMemsetAsync(slab.p, 0, lpad * streams * sizeof(float), e->stream);
    if (h == hipSuccess)
        h = hipMemcpy2DAsync(slab.p, lpad * sizeof(float), values, (size_t)len * sizeof(float), (size_t)len * sizeof(float), streams,
                             hipMemcpyHostToDevice, e->stream);
    if (h == hipSuccess) h = gnnvc::stream_sums(slab.p, streams, len, meta.p, off.p, agg.p, mode, e->stream, segsum.p, segmap.p);
    std::vector<float> host((size_t)streams * 16);
    if (h == hipSuccess) h = hipMemcpyAsync(host.data(), agg.p, host.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream);
    if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
    slab.release(); agg.release(); meta.release(); off.release(); segsum.release(); segmap.release();
    if (h != hipSuccess) return fail(e, h == hipErrorOutOfMemory ? GNNVC_ERR_NOMEM : GNNVC_ERR_DEVICE, "stream sum: %s", hipGetErrorString(h));
    for (uint32_t i = 0; i < streams; ++i) sums[i] = host[(size_t)i * 16];
    return GNNVC_OK;
}

int gnnvc_sgemm(gnnvc_engine *e, int trans_a, int trans_b, uint32_t m, uint32_t n, uint32_t k,
                const float *A, uint32_t lda, const float *B, uint32_t ldb, float beta, float *C,
                uint32_t ldc) {
    if (!e) return GNNVC_ERR_INVALID;
    if ((size_t)m * n == 0) return GNNVC_OK;
    if (!C || (k && (!A || !B))) return fail(e, GNNVC_ERR_INVALID, "null host buffers");
    const size_t a_rows = trans_a ? k : m, b_rows = trans_b ? n : k;
    const size_t a_cols = trans_a ? m : k, b_cols = trans_b ? k : n;
    if (lda < a_cols || ldb < b_cols || ldc < n) return fail(e, GNNVC_ERR_INVALID, "leading dimension too small");
    int rc = use_device(e);
    if (rc) return rc;
    // BLAS guarantees only (rows - 1) * ld + cols elements of a matrix with a padded leading dimension: copy exactly
    // that much in either direction (the last row's padding is not the caller's memory)
    const size_t a_elems = a_rows ? (a_rows - 1) * lda + a_cols : 0;
    const size_t b_elems = b_rows ? (b_rows - 1) * ldb + b_cols : 0;
    const size_t c_elems = (size_t)(m - 1) * ldc + n;
    DevBuf<float> dB;
    HIP_TRY(e, dB.reserve(b_elems + 1));
    if (b_elems && hipMemcpy(dB.p, B, b_elems * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        dB.release();
        return fail(e, GNNVC_ERR_DEVICE, "B upload failed");
    }
    struct Ctx { int ta, tb; uint32_t m, n, k, lda, ldb, ldc; float beta; const float *B; } ctx{
        trans_a, trans_b, m, n, k, lda, ldb, ldc, beta, dB.p};
    rc = run_host_op(e, a_elems, A, c_elems, C, true,
                     [](gnnvc_engine *en, const float *dA, float *dC, void *c) {
                         auto *x = (Ctx *)c;
                         return gnnvc::launch_sgemm(x->ta, x->tb, x->m, x->n, x->k, dA, x->lda, x->B, x->ldb,
                                                    x->beta, dC, x->ldc, en->stream);
                     }, &ctx);
    dB.release();
    return rc;
}

}  // extern "C"
