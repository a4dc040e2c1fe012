// gnn_inference.hpp — umbrella header under the reference's file name, so that
// `#include "gnn_inference.hpp"` in code written against KennethLangedal/GNN-MWVC finds the
// MI355X-backed mirror of its GNN interface:
//   gnn_layers.hpp  linear_layer, graph_layer, ReLU, sigmoid
//   gnn_model.hpp   component, model (predict = the hot path), text (de)serialisation
#pragma once
#include "gnn_layers.hpp"
#include "gnn_model.hpp"
