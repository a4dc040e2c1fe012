// ref_parse_harness.cpp — C entry point around the REFERENCE's own METIS loader (parse_graph,
// reference src/GNN_VC.cpp:34-91) for tests/test_metis_loader.py.  The reference TU is
// included where it lies with its main() renamed; nothing is copied.  Built by
// oracle/Makefile into oracle/_ref/ref_parse.so.
#define main gnnvc_ref_main_unused
#include "/root/reference/src/GNN_VC.cpp"
#undef main

#include <cstdlib>

extern "C" int ref_parse(const char *path, uint32_t *n, uint64_t *m, uint32_t **weights, uint32_t **pairs) {
    test_graph t = parse_graph(path);
    *n = (uint32_t)t.N;
    *m = t.edges.size();
    *weights = static_cast<uint32_t *>(malloc((t.weights.size() + 1) * sizeof(uint32_t)));
    *pairs = static_cast<uint32_t *>(malloc((t.edges.size() + 1) * 2 * sizeof(uint32_t)));
    for (size_t i = 0; i < t.weights.size(); ++i) (*weights)[i] = t.weights[i];
    for (size_t i = 0; i < t.edges.size(); ++i) {
        (*pairs)[2 * i] = t.edges[i].first;
        (*pairs)[2 * i + 1] = t.edges[i].second;
    }
    return 0;
}
