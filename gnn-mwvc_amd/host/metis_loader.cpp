// metis_loader.cpp — see metis_loader.hpp.
#include "metis_loader.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace gnnvc_host {
namespace {

struct Mapped {
    const char *p = nullptr;
    size_t len = 0;
    int fd = -1;
    ~Mapped() {
        if (p && len) munmap(const_cast<char *>(p), len);
        if (fd >= 0) close(fd);
    }
};

inline const char *skip_blank(const char *p, const char *e) {
    while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    return p;
}

// unsigned decimal; returns false if no digit at p
inline bool parse_u64(const char *&p, const char *e, uint64_t &v) {
    if (p >= e || *p < '0' || *p > '9') return false;
    uint64_t x = 0;
    while (p < e && *p >= '0' && *p <= '9') x = x * 10 + (uint64_t)(*p++ - '0');
    v = x;
    return true;
}

struct Chunk {
    uint32_t v_lo = 0, v_hi = 0;            // vertex range
    const char *begin = nullptr;            // first byte of line v_lo
    std::vector<std::pair<uint32_t, uint32_t>> edges;
    std::string err;
};

void parse_chunk(Chunk &c, const char *end, uint32_t n, uint32_t *weights) {
    const char *p = c.begin;
    std::vector<uint32_t> nb;
    for (uint32_t i = c.v_lo; i < c.v_hi; ++i) {
        const char *le = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        if (!le) le = end;
        const char *q = skip_blank(p, le);
        uint64_t w = 0;
        if (parse_u64(q, le, w)) {
            weights[i] = (uint32_t)w;
            nb.clear();
            for (;;) {
                q = skip_blank(q, le);
                uint64_t id;
                if (!parse_u64(q, le, id)) break;
                if (id == 0 || id > n) {
                    c.err = "vertex " + std::to_string(i + 1) + ": neighbour id " + std::to_string(id) + " out of range";
                    return;
                }
                if (id - 1 > i) nb.push_back((uint32_t)(id - 1));   // the reference keeps only e > i (e 0-based)
            }
            if (q < le && *q != '\r') {
                c.err = "vertex " + std::to_string(i + 1) + ": unexpected character in line";
                return;
            }
            std::sort(nb.begin(), nb.end());
            nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
            for (uint32_t v : nb) c.edges.emplace_back(i, v);
        } else {
            weights[i] = 0;   // blank line: the reference's stream extraction leaves the weight at 0
        }
        p = le < end ? le + 1 : end;
        if (p >= end && i + 1 < c.v_hi) {
            // file ended early: remaining vertices have weight 0 and no neighbours (like getline on EOF)
            for (uint32_t r = i + 1; r < c.v_hi; ++r) weights[r] = 0;
            return;
        }
    }
}

}  // namespace

std::string load_metis(const std::string &path, metis_graph &out, unsigned threads) {
    Mapped m;
    m.fd = open(path.c_str(), O_RDONLY);
    if (m.fd < 0) return "cannot open " + path;
    struct stat st;
    if (fstat(m.fd, &st) != 0) return "cannot stat " + path;
    m.len = (size_t)st.st_size;
    if (m.len == 0) return "empty file";
    void *addr = mmap(nullptr, m.len, PROT_READ, MAP_PRIVATE, m.fd, 0);
    if (addr == MAP_FAILED) {
        m.len = 0;
        return "mmap failed for " + path;
    }
    m.p = static_cast<const char *>(addr);
    const char *end = m.p + m.len;

    // header: N E [anything]
    const char *p = m.p;
    const char *le = static_cast<const char *>(memchr(p, '\n', m.len));
    if (!le) le = end;
    uint64_t n64 = 0, e64 = 0;
    p = skip_blank(p, le);
    if (!parse_u64(p, le, n64)) return "header: expected vertex count";
    p = skip_blank(p, le);
    if (!parse_u64(p, le, e64)) return "header: expected edge count";
    if (n64 >= 0xFFFFFFFFull) return "too many vertices for 32-bit ids";
    const uint32_t n = (uint32_t)n64;
    const char *body = le < end ? le + 1 : end;

    out.weights.assign(n, 0);
    out.edges.clear();
    out.header_edges = e64;
    {
        size_t slash = path.find_last_of('/');
        std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
        size_t dot = base.find_last_of('.');
        out.name = dot == std::string::npos ? base : base.substr(0, dot);
    }
    if (n == 0) return "";

    if (threads == 0) threads = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
    if ((size_t)(end - body) < (1u << 20)) threads = 1;

    // cut the body into byte ranges at line boundaries, then give every chunk its vertex range
    // by counting the newlines before it
    std::vector<const char *> cuts(threads + 1);
    cuts[0] = body;
    cuts[threads] = end;
    for (unsigned t = 1; t < threads; ++t) {
        const char *c = body + (size_t)(end - body) * t / threads;
        const char *nl = c < end ? static_cast<const char *>(memchr(c, '\n', (size_t)(end - c))) : nullptr;
        cuts[t] = nl ? nl + 1 : end;
        if (cuts[t] < cuts[t - 1]) cuts[t] = cuts[t - 1];
    }
    std::vector<uint64_t> lines(threads, 0);
    {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                uint64_t cnt = 0;
                for (const char *q = cuts[t]; q < cuts[t + 1];) {
                    const char *nl = static_cast<const char *>(memchr(q, '\n', (size_t)(cuts[t + 1] - q)));
                    if (!nl) break;
                    ++cnt;
                    q = nl + 1;
                }
                lines[t] = cnt;
            });
        for (auto &th : pool) th.join();
    }
    std::vector<Chunk> chunks(threads);
    uint64_t first = 0;
    for (unsigned t = 0; t < threads; ++t) {
        chunks[t].begin = cuts[t];
        chunks[t].v_lo = (uint32_t)std::min<uint64_t>(first, n);
        first += lines[t];
        chunks[t].v_hi = (uint32_t)std::min<uint64_t>(t + 1 == threads ? (uint64_t)n : first, n);
        // the last chunk also owns a final line without a trailing newline (v_hi = n above)
    }
    {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                if (chunks[t].v_lo < chunks[t].v_hi) parse_chunk(chunks[t], end, n, out.weights.data());
            });
        for (auto &th : pool) th.join();
    }
    size_t total = 0;
    for (auto &c : chunks) {
        if (!c.err.empty()) return c.err;
        total += c.edges.size();
    }
    out.edges.reserve(total);
    for (auto &c : chunks) out.edges.insert(out.edges.end(), c.edges.begin(), c.edges.end());
    return "";
}

}  // namespace gnnvc_host

extern "C" int gnnvc_host_load_metis(const char *path, uint32_t *n, uint64_t *m, uint32_t **weights, uint32_t **pairs,
                                     unsigned threads) {
    if (!path || !n || !m || !weights || !pairs) return -1;
    gnnvc_host::metis_graph g;
    const std::string err = gnnvc_host::load_metis(path, g, threads);
    if (!err.empty()) return -1;
    *n = (uint32_t)g.weights.size();
    *m = g.edges.size();
    *weights = static_cast<uint32_t *>(malloc(std::max<size_t>(1, g.weights.size()) * sizeof(uint32_t)));
    *pairs = static_cast<uint32_t *>(malloc(std::max<size_t>(1, g.edges.size()) * 2 * sizeof(uint32_t)));
    if (!*weights || !*pairs) return -3;
    std::copy(g.weights.begin(), g.weights.end(), *weights);
    for (size_t i = 0; i < g.edges.size(); ++i) {
        (*pairs)[2 * i] = g.edges[i].first;
        (*pairs)[2 * i + 1] = g.edges[i].second;
    }
    return 0;
}
