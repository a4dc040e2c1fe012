// metis_loader.hpp — fast reader for the reference's graph files.
//
// Same file format and the same resulting edge set as the reference's parse_graph (reference
// src/GNN_VC.cpp:34-91, format README.md:49-62): header `N E [ignored]`, then one line per
// vertex i = 0..N-1: `weight n1 n2 ...` with 1-based neighbour ids, of which only those with
// id - 1 > i are kept; the (i, id - 1) pairs are sorted and de-duplicated (self-loops and
// entries that only the higher endpoint lists vanish, exactly as in the reference).
//
// The reference parses with one stringstream per line (≈ 7 s per 10 M edges); this reader
// maps the file, splits it at line boundaries and parses the chunks on several threads.
// Because chunk k holds the lines of a contiguous vertex range and every kept pair (i, j) has
// i < j, sorting each line's kept neighbours yields the globally sorted unique pair list
// without a global sort.  Differences, by design: the edge count of the header is not
// trusted (the reference pre-sizes its array from it; a wrong E is undefined behaviour
// there); malformed files raise an error instead.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace gnnvc_host {

struct metis_graph {
    std::string name;                                       // file stem, like the reference's test_graph::name
    std::vector<uint32_t> weights;                          // N
    std::vector<std::pair<uint32_t, uint32_t>> edges;       // sorted unique (u < v)
    uint64_t header_edges = 0;                              // E as written in the file
};

// Returns an empty string on success, else an error message.  threads = 0: one per CPU (max 16).
std::string load_metis(const std::string &path, metis_graph &out, unsigned threads = 0);

}  // namespace gnnvc_host

extern "C" {
// C entry point for tests / other languages: arrays are malloc'ed, caller frees with free().
// pairs holds 2 * (*m) uint32 (u0, v0, u1, v1, ...).  Returns 0 on success.
int gnnvc_host_load_metis(const char *path, uint32_t *n, uint64_t *m, uint32_t **weights, uint32_t **pairs,
                          unsigned threads);
}
