// flagged_reduce.hpp — the host side of SURVEY.md §8 f-2: reduce_graph's loop consuming the per-vertex
// rule predicates of gnnvc_reduction_flags (include/gnnvc.h).
//
// Compiles only inside the reference tree (it calls the reference's own rule functions from
// include/mwvc_reductions.hpp and keeps their order: same stacks, same pops, same rule = 0 restart after a
// hit, reference :335-380), so the graph and the cover come out exactly as from the reference's reduce_graph.
// The one difference: a test is skipped when the flag pass proved that the rule cannot fire on that vertex
// AND nothing within two hops of the vertex has changed since the flags were computed.
//
// "Changed" is tracked without touching the reference's classes: graph_search takes the number of stacks as
// a constructor argument and push_search() feeds all of them, so one extra stack (index 7, never used as a
// rule) is a log of every vertex the rule functions push — the active neighbours of whatever they removed,
// folded or re-weighted.  After each hit the log is drained: those vertices, their current neighbours and
// the neighbours of those are marked dirty.  Two hops from the pushed vertices are needed because twin_fold
// on u compares u with the OTHER neighbours x of u's last neighbour — their degree, NW(x) and whole
// adjacency list — so a removal next to x, three hops from u, can make the rule fire on u (and the
// reference would find it at u's turn, not x's: skipping u's test would change the order of the folds).
//
//   graph_search<Tn> gs(g.size(), 8);                       // 7 rule stacks + the touch log
//   gnnvc_host::flag_state<Tn> fs(flags /* from gnnvc_reduction_flags(e, 20, ...) */, gs);
//   gnnvc_host::reduce_graph_flagged(g, vc, gs, fs);        // instead of reduce_graph(g, vc, gs)
//   ...
//   g.relable_graph(); fs.drop();                           // vertex ids change: the flags are spent
#pragma once
#include <cstdint>
#include <vector>

#include <mwvc_reductions.hpp>

namespace gnnvc_host {

template <typename Tn>
struct flag_state {
    static constexpr size_t kLog = 7;     // index of the touch-log stack in graph_search
    std::vector<uint8_t> flags;           // bit r: rule r can fire on the vertex (as the graph stood)
    std::vector<bool> dirty;              // something within two hops has changed since
    size_t skipped = 0, tested = 0, marks = 0;

    flag_state(std::vector<uint8_t> f, graph_search<Tn> &gs) : flags(std::move(f)), dirty(flags.size(), false) {
        // the constructor of graph_search filled every stack, the log included, with all vertices: empty it
        while (!gs.search[kLog].empty()) gs.pop_search(kLog);
    }
    void drop() {
        flags.clear();
        dirty.clear();
    }
    bool can_skip(Tn u, size_t rule) const { return u < flags.size() && !dirty[u] && !((flags[u] >> rule) & 1u); }

    template <typename Tw>
    void drain(const reduction_graph<Tn, Tw> &g, graph_search<Tn> &gs) {
        if (flags.empty()) {                       // spent: only keep the log from growing
            while (!gs.search[kLog].empty()) gs.pop_search(kLog);
            return;
        }
        if (marks > 8 * flags.size()) {            // around hubs two hops are most of the graph: stop paying for it
            drop();
            return drain(g, gs);
        }
        while (!gs.search[kLog].empty()) {
            const Tn t = gs.pop_search(kLog);
            if (t < dirty.size()) dirty[t] = true;
            if (t < g.size() && g.is_active(t))
                for (auto it = g.begin(t); it != g.end(t); ++it) {
                    const Tn v = *it;
                    if (v < dirty.size()) dirty[v] = true;
                    if (v < g.size() && g.is_active(v))
                        for (auto jt = g.begin(v); jt != g.end(v); ++jt, ++marks)
                            if (*jt < dirty.size()) dirty[*jt] = true;
                }
        }
    }
};

template <typename Tn, typename Tw>
void reduce_graph_flagged(reduction_graph<Tn, Tw> &g, vertex_cover<Tn, Tw> &vc, graph_search<Tn> &gs, flag_state<Tn> &fs,
                          bool do_critical = false) {
    const size_t rules = flag_state<Tn>::kLog;   // the seven local rules; stack 7 is the log
    bool critical = false;
    do {
        size_t rule = 0;
        while (rule < rules) {
            if (gs.search[rule].empty()) {
                ++rule;
                continue;
            }
            const Tn u = gs.pop_search(rule);
            if (u >= g.size() || !g.is_active(u) || g.D(u) > 20) continue;
            if (fs.can_skip(u, rule)) {
                ++fs.skipped;
                continue;
            }
            ++fs.tested;
            bool found = false;
            switch ((reduction_rules)rule) {
            case reduction_rules::neighborhood_reduction: found = neighborhood_reduction(g, vc, gs, u); break;
            case reduction_rules::twin_fold: found = twin_fold(g, vc, gs, u); break;
            case reduction_rules::domination_reduction: found = domination_reduction(g, vc, gs, u); break;
            case reduction_rules::isolated_fold: found = isolated_fold(g, vc, gs, u); break;
            case reduction_rules::independent_fold: found = independent_fold(g, vc, gs, u); break;
            case reduction_rules::neighbor_meta_reduction: found = neighbor_meta_reduction(g, vc, gs, u); break;
            case reduction_rules::neighborhood_meta_reduction: found = neighborhood_meta_reduction(g, vc, gs, u); break;
            default: break;
            }
            if (found) {
                fs.drain(g, gs);
                rule = 0;
            }
        }
        if (do_critical) {
            critical = reduction_critial_weight(g, vc, gs);
            fs.drain(g, gs);
        }
    } while (critical);
}

}  // namespace gnnvc_host
