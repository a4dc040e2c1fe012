// gnn_layers.hpp — the four layer kinds of the GNN, host side.
//
// Interface-compatible with the reference's layer structs (reference
// include/gnn_inference.hpp:11-36): same names, same data members in the same order, same
// forward() signatures — so code written against the reference binds to these unchanged.
// Every forward() here runs on the MI355X through the C ABI (include/gnnvc.h); see
// gnn_inference.cpp for the bindings.
#pragma once
#include <cstddef>
#include <cstdint>

#include "matrix.hpp"
#include "reduction_graph.hpp"

namespace gnn {

using Tn = uint32_t;  // vertex ids
using Tw = uint32_t;  // vertex weights

// Elementwise max(x, 0); out-of-place.                              -> gnnvc_relu_forward
struct ReLU {
    void forward(const matrix &in, matrix &out) const;
};

// Elementwise 1 / (1 + exp(-x)); out-of-place.                      -> gnnvc_sigmoid_forward
struct sigmoid {
    void forward(const matrix &in, matrix &out) const;
};

// Dense layer: out = in * W + bias (W is in x out, bias 1 x out).   -> gnnvc_linear_forward
// The constructor reproduces the reference's training-time initialisation
// (uniform in +-1/sqrt(in + 1), mt19937(seed)); inference loads W and bias from the model text.
struct linear_layer {
    matrix W, bias;

    linear_layer(size_t dim_in = 0, size_t dim_out = 0, size_t seed = 0);

    void forward(const matrix &in, matrix &out) const;
};

// Message passing over the graph: for an N x F input the output is N x (2F + 3).
// Column layout as the reference COMPUTES it (src/gnn_inference.cpp:33-40), which is what the
// shipped weights were trained with: [0, F) neighbour sums in stored adjacency order,
// [F, 2F) a copy of the input row, then degree, W/WEIGHT_SCALE and NW/WEIGHT_SCALE written at
// columns F+1, F+2, F+3 — over the copy for F > 1, leaving the last three columns zero
// (DESIGN.md §3).                                                   -> gnnvc_graph_layer_forward
struct graph_layer {
    float WEIGHT_SCALE = 120.0f;

    void forward(const matrix &in, matrix &out, const reduction_graph<Tn, Tw> &g) const;
};

}  // namespace gnn
