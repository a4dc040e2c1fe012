// predict_tool.cpp — smallest possible caller of the host mirror, used by the GPU
// tests to exercise gnn::model::predict end to end through libgnnvc_hip.so:
//   gnnvc_predict <model.txt> <graph.metis> <scores.f32>
// Loads the model with gnn::operator>>, reads a METIS graph with the semantics of
// the reference's loader (reference src/GNN_VC.cpp:34-91: keep neighbours > i,
// sort, unique), sets the weight scale to the maximum weight (:272-278), builds
// x = W/ws (:189-191), calls predict and writes the N fp32 scores.
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "gnn_inference.hpp"

int main(int argc, char **argv) {
    if (argc != 4) {
        std::fprintf(stderr, "usage: %s <model.txt> <graph.metis> <scores.f32>\n", argv[0]);
        return 2;
    }
    gnn::model m;
    {
        std::ifstream ms(argv[1]);
        if (!ms) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
        ms >> m;
    }
    std::ifstream gs(argv[2]);
    if (!gs) { std::fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
    std::string line;
    std::getline(gs, line);
    size_t n = 0, e_hdr = 0;
    { std::istringstream hs(line); hs >> n >> e_hdr; }
    std::vector<uint32_t> w(n);
    std::vector<std::pair<uint32_t, uint32_t>> edges;
    edges.reserve(e_hdr);
    for (size_t i = 0; i < n; ++i) {
        std::getline(gs, line);
        std::istringstream ls(line);
        ls >> w[i];
        size_t v;
        while (ls >> v)
            if (v - 1 > i) edges.push_back({(uint32_t)i, (uint32_t)(v - 1)});
    }
    std::sort(edges.begin(), edges.end());
    edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
    reduction_graph<uint32_t, uint32_t> g(w, edges);
    float ws = 0;
    for (size_t i = 0; i < n; ++i) ws = std::max(ws, (float)g.W((uint32_t)i));
    m.set_weight_scale(ws);
    matrix x(n, 1), out;
    for (size_t i = 0; i < n; ++i) x(i, 0) = (float)g.W((uint32_t)i) / ws;
    m.predict(x, out, g);
    // second call on the same model/graph: engines and buffers are reused
    matrix out2;
    m.predict(x, out2, g);
    for (size_t i = 0; i < n; ++i)
        if (out(i, 0) != out2(i, 0)) { std::fprintf(stderr, "predict not repeatable at %zu\n", i); return 3; }
    FILE *f = std::fopen(argv[3], "wb");
    if (!f) return 1;
    for (size_t i = 0; i < out.get_height(); ++i) { float v = out(i, 0); std::fwrite(&v, 4, 1, f); }
    std::fclose(f);
    std::printf("%zu vertices, %zu edges, ws=%g, %zu scores written\n", n, edges.size(), ws, out.get_height());
    return 0;
}
