// result_writer.hpp — the reference CLI's files, written in one go.
//
// Result file (reference src/GNN_VC.cpp:388-391, README.md:43-47): N lines, line u = "1" if vertex u is in the
// cover, else "0".  The reference writes it with `os << (...) << endl` per vertex — a flush, i.e. a write
// system call, per line: ~1 µs x 10 M vertices on top of the formatting.  Here the text is assembled in
// memory (two bytes per vertex) and handed to the file with a single write.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace gnnvc_host {

// in_cover[u] != 0 <=> vertex u is in the cover.  Returns an empty string on success, else an error message.
std::string write_cover_file(const std::string &path, const uint8_t *in_cover, size_t n);

// the text itself (tests; callers that stream it elsewhere)
std::string cover_text(const uint8_t *in_cover, size_t n);

// The reference's INPUT format written from a CSR (README.md:49-62: header `N E 10`, then per vertex
// `weight n1 n2 ...`, 1-based neighbour ids): the counterpart of load_metis, for tools that hand graphs to the
// CLI.  E = nnz / 2 (a symmetric CSR lists every edge twice).
std::string write_metis_file(const std::string &path, uint32_t n, const uint64_t *rowptr, const uint32_t *col,
                             const uint32_t *w);

}  // namespace gnnvc_host

extern "C" {
// C entry points: 0 on success.
int gnnvc_host_write_cover(const char *path, const uint8_t *in_cover, size_t n);
int gnnvc_host_write_metis(const char *path, uint32_t n, const uint64_t *rowptr, const uint32_t *col, const uint32_t *w);
}
