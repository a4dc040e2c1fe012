// reduction_graph.hpp — the READ side of the reference's reduction_graph
// (reference include/reduction_graph.hpp:141-158,693-704) that the GNN forward
// consumes: size(), begin(u)/end(u), D(u), W(u), NW(u).  The reference's class
// also carries the reduction machinery (mutators, undo log) — that stays with the
// reference's host code; when this engine is dropped into the reference tree its
// own header is used instead of this one.
//
// This stand-alone version exists so the host mirror and its tests build
// without the reference: a compact CSR, constructed from sorted unique (u < v)
// pairs exactly like the reference's constructor (:104-128), neighbour lists
// ascending, NW(u) the uint32 sum of neighbour weights.
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

template <typename Tn, typename Tw>
class reduction_graph {
  public:
    reduction_graph(const std::vector<Tw> &w, const std::vector<std::pair<Tn, Tn>> &e)
        : weights_(w), nbr_weights_(w.size(), 0), offsets_(w.size() + 1, 0), adj_(e.size() * 2) {
        for (const auto &uv : e) {
            ++offsets_[uv.first + 1];
            ++offsets_[uv.second + 1];
            nbr_weights_[uv.first] += w[uv.second];
            nbr_weights_[uv.second] += w[uv.first];
        }
        for (size_t i = 1; i < offsets_.size(); ++i) offsets_[i] += offsets_[i - 1];
        std::vector<size_t> fill(offsets_.begin(), offsets_.end() - 1);
        for (const auto &uv : e) {
            adj_[fill[uv.first]++] = uv.second;
            adj_[fill[uv.second]++] = uv.first;
        }
    }

    Tn size() const { return (Tn)weights_.size(); }
    // the reference keeps, per vertex, the label it had when it was created (include/reduction_graph.hpp:134-139); this
    // read-only mirror never relabels, so a vertex's original label is its index
    Tn get_org_label(Tn u) const { return u; }
    Tn D(Tn u) const { return (Tn)(offsets_[u + 1] - offsets_[u]); }
    Tw W(Tn u) const { return weights_[u]; }
    Tw NW(Tn u) const { return nbr_weights_[u]; }

    typename std::vector<Tn>::const_iterator begin(Tn u) const { return adj_.begin() + offsets_[u]; }
    typename std::vector<Tn>::const_iterator end(Tn u) const { return adj_.begin() + offsets_[u + 1]; }

  private:
    std::vector<Tw> weights_, nbr_weights_;
    std::vector<size_t> offsets_;
    std::vector<Tn> adj_;
};
