// result_writer.cpp — see result_writer.hpp.
#include "result_writer.hpp"

#include <cerrno>
#include <cstdio>
#include <cstring>

namespace gnnvc_host {

std::string cover_text(const uint8_t *in_cover, size_t n) {
    std::string text(2 * n, '\n');
    for (size_t u = 0; u < n; ++u) text[2 * u] = in_cover[u] ? '1' : '0';
    return text;
}

std::string write_cover_file(const std::string &path, const uint8_t *in_cover, size_t n) {
    if (n && !in_cover) return "null cover array";
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return "cannot open " + path + ": " + std::strerror(errno);
    const std::string text = cover_text(in_cover, n);
    const size_t done = text.empty() ? 0 : std::fwrite(text.data(), 1, text.size(), f);
    const bool closed = std::fclose(f) == 0;   // (always: a short write must not leak the handle)
    const bool ok = done == text.size() && closed;
    if (!ok) return "short write to " + path;
    return "";
}

namespace {
inline char *put_u32(char *p, uint32_t v) {
    char tmp[10];
    int k = 0;
    do { tmp[k++] = (char)('0' + v % 10u); v /= 10u; } while (v);
    while (k) *p++ = tmp[--k];
    return p;
}
}  // namespace

std::string write_metis_file(const std::string &path, uint32_t n, const uint64_t *rowptr, const uint32_t *col,
                             const uint32_t *w) {
    if (n && (!rowptr || !w)) return "null graph arrays";
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return "cannot open " + path + ": " + std::strerror(errno);
    const uint64_t nnz = n ? rowptr[n] : 0;
    std::string buf;
    buf.reserve(1 << 24);
    buf += std::to_string(n) + " " + std::to_string(nnz / 2) + " 10\n";
    char num[16];
    bool ok = true;
    for (uint32_t u = 0; u < n && ok; ++u) {
        buf.append(num, put_u32(num, w[u]) - num);
        buf.push_back(' ');
        for (uint64_t e = rowptr[u]; e < rowptr[u + 1]; ++e) {
            if (e != rowptr[u]) buf.push_back(' ');
            buf.append(num, put_u32(num, col[e] + 1u) - num);
        }
        buf.push_back('\n');
        if (buf.size() > (8u << 20)) {
            ok = std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
            buf.clear();
        }
    }
    if (ok && !buf.empty()) ok = std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok ? "" : "short write to " + path;
}

}  // namespace gnnvc_host

extern "C" int gnnvc_host_write_metis(const char *path, uint32_t n, const uint64_t *rowptr, const uint32_t *col, const uint32_t *w) {
    if (!path) return -1;
    return gnnvc_host::write_metis_file(path, n, rowptr, col, w).empty() ? 0 : -1;
}

extern "C" int gnnvc_host_write_cover(const char *path, const uint8_t *in_cover, size_t n) {
    if (!path) return -1;
    return gnnvc_host::write_cover_file(path, in_cover, n).empty() ? 0 : -1;
}
