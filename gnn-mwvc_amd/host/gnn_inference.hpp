// gnn_inference.hpp — host mirror of the reference's GNN interface (reference
// include/gnn_inference.hpp:7-59): same namespace, type names, member layout and
// signatures, so `m.predict(x, out, g)` at reference src/GNN_VC.cpp:192 and the
// per-layer forward()s bind to this engine without source changes.  What differs
// is underneath: every forward() runs on the MI355X through include/gnnvc.h.
#pragma once
#include <cstdint>
#include <iosfwd>
#include <string>
#include <variant>
#include <vector>

#include "matrix.hpp"
#include "reduction_graph.hpp"

namespace gnn {

using Tw = uint32_t;
using Tn = uint32_t;

// out = in * W + bias                                   (reference :11-17)
struct linear_layer {
    matrix W, bias;
    linear_layer(size_t dim_in = 0, size_t dim_out = 0, size_t seed = 0);
    void forward(const matrix &in, matrix &out) const;
};

// Message passing; output width = 2 * input width + 3    (reference :19-28).
// Column layout as the reference computes it (not as its comment says): the
// degree / weight / neighbourhood-weight columns land at f+1..f+3.
struct graph_layer {
    float WEIGHT_SCALE = 120.0f;
    void forward(const matrix &in, matrix &out, const reduction_graph<Tn, Tw> &g) const;
};

struct ReLU {
    void forward(const matrix &in, matrix &out) const;
};

struct sigmoid {
    void forward(const matrix &in, matrix &out) const;
};

using component = std::variant<linear_layer, graph_layer, ReLU, sigmoid>;

class model {
  private:
    std::string name;
    std::vector<component> layers;
    mutable matrix in_copy;

  public:
    model(std::string name = "");
    void add_layer(const component &c);

    // The hot path: scores for every vertex of g (N x 1), N = 0 allowed.
    void predict(const matrix &in, matrix &out, const reduction_graph<Tn, Tw> &g) const;

    void set_weight_scale(float ws);

    friend std::ostream &operator<<(std::ostream &os, const model &m);
    friend std::istream &operator>>(std::istream &is, model &m);
};

std::ostream &operator<<(std::ostream &os, const model &m);
std::istream &operator>>(std::istream &is, model &m);

}  // namespace gnn
