// ref_driver_fastio.cpp — the reference's CLI driver (src/GNN_VC.cpp, compiled where it lies, nothing copied) with
// this repo's file I/O behind it (INTEGRATION.md "fast I/O"):
//   * parse_graph(...) in main() -> gnnvc_host::load_metis (mmap + threads; same edge set, host/metis_loader.hpp);
//   * every `<< endl` -> a newline WITHOUT the flush, so the result file (one line per vertex, :321 and :388-391)
//     is written through the stream's buffer instead of one write system call per vertex.
// How, without touching the reference's text: the preprocessor.  `parse_graph` becomes a function-like macro that
// pastes its first argument token onto a prefix, which tells the DEFINITION `parse_graph(filesystem::path path)` —
// renamed, kept, unused — from the one CALL `parse_graph(graph_path)` in main() — sent to the function below.
// Built by oracle/Makefile (`make ref`) into _ref/GNN_VC_dropin_fastio (ABI test double) and
// _ref/GNN_VC_hip_fastio (the HIP engine); tests/test_dropin_link.py checks identical output.
#include <filesystem>
#include <ostream>
#include <string>

#include "metis_loader.hpp"

struct test_graph;
test_graph gnnvc_fast_parse_graph(std::filesystem::path path);

namespace gnnvc_fastio {
template <class C, class T>
std::basic_ostream<C, T> &newline(std::basic_ostream<C, T> &os) {
    return os.put(os.widen('\n'));
}
}  // namespace gnnvc_fastio

#define endl gnnvc_fastio::newline
#define parse_graph(a) GNNVC_PG_##a )
#define GNNVC_PG_filesystem gnnvc_reference_parse_graph(filesystem
#define GNNVC_PG_graph_path gnnvc_fast_parse_graph(graph_path
#include "/root/reference/src/GNN_VC.cpp"
#undef parse_graph
#undef GNNVC_PG_filesystem
#undef GNNVC_PG_graph_path
#undef endl

test_graph gnnvc_fast_parse_graph(std::filesystem::path path) {
    gnnvc_host::metis_graph mg;
    const std::string err = gnnvc_host::load_metis(path.string(), mg);
    if (!err.empty()) {
        std::cout << "Error opening graph file" << std::endl;   // the reference's message for an unreadable file (:40)
        return {reduction_graph<Tn, Tw>({}, {}), {}, {}, mg.name, 0, 0};
    }
    const size_t N = mg.weights.size(), E = mg.edges.size();
    return {reduction_graph<Tn, Tw>(mg.weights, mg.edges), mg.weights, mg.edges, mg.name, N, E};
}
