// ref_harness.cpp — C entry points around the REFERENCE's own layer code, for
// tests (oracle/_ref/ref_layers.so).  Compiled against /root/reference/include
// and linked with the reference's src/gnn_inference.cpp (compiled where it lies)
// and this repo's matrix translation unit; see oracle/Makefile.  No reference
// source is copied: this file only calls the reference's public interface.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "gnn_inference.hpp"
#include "mwvc_reductions.hpp"
#include "flagged_reduce.hpp"
#include <chrono>

namespace {
reduction_graph<uint32_t, uint32_t> make_graph(uint32_t n, const uint64_t *rowptr, const uint32_t *col,
                                                const uint32_t *w) {
    std::vector<uint32_t> weights(w, w + n);
    std::vector<std::pair<uint32_t, uint32_t>> edges;
    for (uint32_t u = 0; u < n; ++u)
        for (uint64_t e = rowptr[u]; e < rowptr[u + 1]; ++e)
            if (col[e] > u) edges.push_back({u, col[e]});
    return reduction_graph<uint32_t, uint32_t>(weights, edges);  // sorted by construction
}
}  // namespace

extern "C" {

// reference gnn::model::predict on a CSR graph; out has n floats.
// n_layers_keep < 0 keeps every layer, otherwise truncates the model text's
// layer count (e.g. 20 of 21 drops the final sigmoid -> logits).
int ref_predict(const char *model_text, int n_layers_keep, float ws, uint32_t n, const uint64_t *rowptr,
                const uint32_t *col, const uint32_t *w, const float *x, float *out, uint32_t *out_width) {
    std::string text(model_text);
    if (n_layers_keep >= 0) {
        // header is "<name> <n> Layers": rewrite the count; the parser stops after that many records
        std::istringstream hs(text);
        std::string name, cnt;
        hs >> name >> cnt;
        const size_t pos = text.find(cnt, name.size());
        text.replace(pos, cnt.size(), std::to_string(n_layers_keep));
    }
    gnn::model m;
    std::istringstream is(text);
    is >> m;
    m.set_weight_scale(ws);
    auto g = make_graph(n, rowptr, col, w);
    matrix in(n, 1), res;
    for (uint32_t u = 0; u < n; ++u) in(u, 0) = x[u];
    m.predict(in, res, g);
    *out_width = (uint32_t)res.get_width();
    for (size_t i = 0; i < res.get_height(); ++i)
        for (size_t j = 0; j < res.get_width(); ++j) out[i * res.get_width() + j] = res(i, j);
    return 0;
}

// reference gnn::graph_layer::forward
int ref_graph_layer(float ws, uint32_t n, uint32_t f, const uint64_t *rowptr, const uint32_t *col,
                    const uint32_t *w, const float *in, float *out) {
    auto g = make_graph(n, rowptr, col, w);
    gnn::graph_layer gl;
    gl.WEIGHT_SCALE = ws;
    matrix a(n, f), res;
    for (uint32_t u = 0; u < n; ++u)
        for (uint32_t j = 0; j < f; ++j) a(u, j) = in[(size_t)u * f + j];
    gl.forward(a, res, g);
    for (size_t i = 0; i < res.get_height(); ++i)
        for (size_t j = 0; j < res.get_width(); ++j) out[i * res.get_width() + j] = res(i, j);
    return (int)res.get_width();
}

// Reduction-rule predicates evaluated with the REFERENCE's own graph methods
// (is_twin / is_dominating / is_isolated, include/reduction_graph.hpp:180-224) in the way the rule
// functions call them (include/mwvc_reductions.hpp:131-284); bits as in oracle_reduction_flags.
int ref_reduction_flags(uint32_t n, const uint64_t *rowptr, const uint32_t *col, const uint32_t *w,
                        uint32_t max_degree, uint8_t *flags) {
    auto g = make_graph(n, rowptr, col, w);
    for (uint32_t u = 0; u < n; ++u) {
        uint8_t f = 0;
        const uint32_t d = g.D(u);
        if (d <= max_degree) {
            f |= 0x60;
            if (g.NW(u) <= g.W(u)) f |= 1u << 0;
            if (d > 0) {
                const uint32_t first_neighbor = *(g.end(u) - 1);
                for (auto it = g.begin(first_neighbor); it != g.end(first_neighbor); ++it)
                    if (*it != u && g.is_twin(u, *it)) { f |= 1u << 1; break; }
                for (auto it = g.begin(u); it != g.end(u); ++it) {
                    const uint32_t v = *it;
                    if ((g.W(v) >= g.W(u) && g.is_dominating(u, v)) || (g.W(v) <= g.W(u) && g.is_dominating(v, u))) {
                        f |= 1u << 2;
                        break;
                    }
                }
                uint32_t wmin = g.W(*g.begin(u));
                for (auto it = g.begin(u); it != g.end(u); ++it) wmin = std::min<uint32_t>(wmin, g.W(*it));
                if (g.W(u) >= g.NW(u) - wmin) f |= 1u << 4;
            }
            if (g.is_isolated(u)) f |= 1u << 3;
        }
        flags[u] = f;
    }
    return 0;
}

// The two rules that run the small exact solver (include/mwvc_reductions.hpp:204-252), evaluated by calling the
// REFERENCE's own rule functions — each on a fresh copy of the graph, because they apply the reduction when they
// fire.  flags[u] bit 5 = neighbor_meta_reduction fires on u, bit 6 = neighborhood_meta_reduction fires on u, for the
// vertices reduce_graph would look at (D(u) <= max_degree, :344).  O(n (n + m)): small graphs only.
int ref_meta_flags(uint32_t n, const uint64_t *rowptr, const uint32_t *col, const uint32_t *w, uint32_t max_degree,
                   uint8_t *flags) {
    const auto g0 = make_graph(n, rowptr, col, w);
    for (uint32_t u = 0; u < n; ++u) {
        uint8_t f = 0;
        if (g0.D(u) <= max_degree) {
            {
                auto g = g0;
                vertex_cover<uint32_t, uint32_t> vc(n);
                graph_search<uint32_t> gs(n);
                if (neighbor_meta_reduction(g, vc, gs, u)) f |= 1u << 5;
            }
            {
                auto g = g0;
                vertex_cover<uint32_t, uint32_t> vc(n);
                graph_search<uint32_t> gs(n);
                if (neighborhood_meta_reduction(g, vc, gs, u)) f |= 1u << 6;
            }
        }
        flags[u] = f;
    }
    return 0;
}

// reduce_graph (reference) against gnnvc_host::reduce_graph_flagged (host/flagged_reduce.hpp) on the same graph:
// returns 0 when graph and cover come out identical; stats = {ms reference, ms flagged, tests skipped, tests run}.
int ref_flagged_reduce_check(uint32_t n, const uint64_t *rowptr, const uint32_t *col, const uint32_t *w,
                             const uint8_t *flags, double *stats) {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto ga = make_graph(n, rowptr, col, w);
    auto gb = ga;
    vertex_cover<uint32_t, uint32_t> va(n), vb(n);
    graph_search<uint32_t> gsa(n);
    const auto t0 = clk::now();
    reduce_graph(ga, va, gsa);
    const auto t1 = clk::now();
    graph_search<uint32_t> gsb(n, 8);
    gnnvc_host::flag_state<uint32_t> fs(std::vector<uint8_t>(flags, flags + n), gsb);
    const auto t2 = clk::now();
    gnnvc_host::reduce_graph_flagged(gb, vb, gsb, fs);
    const auto t3 = clk::now();
    stats[0] = ms(t0, t1);
    stats[1] = ms(t2, t3);
    stats[2] = (double)fs.skipped;
    stats[3] = (double)fs.tested;
    if (va.cost != vb.cost || va.S.size() != vb.S.size() || ga.size() != gb.size()) return 1;
    if (va.r1 != vb.r1 || va.r2 != vb.r2 || va.r3 != vb.r3 || va.r4 != vb.r4 || va.r5 != vb.r5 || va.r6 != vb.r6 ||
        va.r7 != vb.r7)
        return 2;
    for (size_t i = 0; i < va.S.size(); ++i)
        if (va.S[i] != vb.S[i]) return 3;
    for (uint32_t u = 0; u < ga.size(); ++u) {
        if (ga.is_active(u) != gb.is_active(u)) return 4;
        if (!ga.is_active(u)) continue;
        if (ga.W(u) != gb.W(u) || ga.NW(u) != gb.NW(u)) return 5;
        if (!std::equal(ga.begin(u), ga.end(u), gb.begin(u), gb.end(u))) return 6;
    }
    return 0;
}

// debugging aid: the flagged loop, but every skipped test is also run on a copy — reports the first skip that
// would have fired: out = {vertex, rule, tests so far}
int ref_flagged_debug(uint32_t n, const uint64_t *rowptr, const uint32_t *col, const uint32_t *w, const uint8_t *flags,
                      uint32_t *out) {
    auto g = make_graph(n, rowptr, col, w);
    vertex_cover<uint32_t, uint32_t> vc(n);
    graph_search<uint32_t> gs(n, 8);
    gnnvc_host::flag_state<uint32_t> fs(std::vector<uint8_t>(flags, flags + n), gs);
    size_t rule = 0, steps = 0;
    while (rule < 7) {
        if (gs.search[rule].empty()) { ++rule; continue; }
        const uint32_t u = gs.pop_search(rule);
        if (u >= g.size() || !g.is_active(u) || g.D(u) > 20) continue;
        ++steps;
        auto run = [&](reduction_graph<uint32_t, uint32_t> &gg, vertex_cover<uint32_t, uint32_t> &vv, graph_search<uint32_t> &ss) {
            switch ((reduction_rules)rule) {
            case reduction_rules::neighborhood_reduction: return neighborhood_reduction(gg, vv, ss, u);
            case reduction_rules::twin_fold: return twin_fold(gg, vv, ss, u);
            case reduction_rules::domination_reduction: return domination_reduction(gg, vv, ss, u);
            case reduction_rules::isolated_fold: return isolated_fold(gg, vv, ss, u);
            case reduction_rules::independent_fold: return independent_fold(gg, vv, ss, u);
            case reduction_rules::neighbor_meta_reduction: return neighbor_meta_reduction(gg, vv, ss, u);
            case reduction_rules::neighborhood_meta_reduction: return neighborhood_meta_reduction(gg, vv, ss, u);
            default: return false;
            }
        };
        if (fs.can_skip(u, rule)) {
            auto g2 = g;
            auto v2 = vc;
            auto s2 = gs;
            if (run(g2, v2, s2)) {
                out[0] = u;
                out[1] = (uint32_t)rule;
                out[2] = (uint32_t)steps;
                return 1;
            }
            continue;
        }
        if (run(g, vc, gs)) {
            fs.drain(g, gs);
            rule = 0;
        }
    }
    return 0;
}

}  // extern "C"
