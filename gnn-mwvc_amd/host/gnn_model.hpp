// gnn_model.hpp — gnn::model, the container whose predict() is the hot path
// (reference include/gnn_inference.hpp:38-59, src/gnn_inference.cpp:54-139).
#pragma once
#include <iosfwd>
#include <string>
#include <variant>
#include <vector>

#include "gnn_layers.hpp"

namespace gnn {

// alternative order is part of the interface (std::visit call sites index into it)
using component = std::variant<linear_layer, graph_layer, ReLU, sigmoid>;

class model {
  public:
    model(std::string name = "");

    // Scores of every vertex of g: in is N x 1 (x[u] = W(u) / weight scale), out becomes N x 1.
    // Runs as three fused HIP stages (gnnvc_upload_graph + gnnvc_forward); N = 0 is a no-op —
    // the reference's driver ends every run with a predict on the empty graph.
    void predict(const matrix &in, matrix &out, const reduction_graph<Tn, Tw> &g) const;

    void add_layer(const component &c);
    void set_weight_scale(float ws);   // WEIGHT_SCALE of every graph layer

    // text format: "<name>\n<n> Layers\n" then one record per layer
    friend std::istream &operator>>(std::istream &is, model &m);
    friend std::ostream &operator<<(std::ostream &os, const model &m);

  private:
    // same members, same order as the reference's class (one layout for both header sets)
    std::string name;
    std::vector<component> layers;
    mutable matrix in_copy;   // the reference's ping-pong scratch; here: logits when the host sigmoid is requested
};

std::istream &operator>>(std::istream &is, model &m);
std::ostream &operator<<(std::ostream &os, const model &m);

}  // namespace gnn
