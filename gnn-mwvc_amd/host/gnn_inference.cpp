// gnn_inference.cpp — host mirror of the reference's GNN layer structs and
// gnn::model (reference src/gnn_inference.cpp), bound to the MI355X engine.
//
//   model::predict          -> gnnvc_upload_graph + gnnvc_forward   (reference :67-81)
//   graph_layer::forward    -> gnnvc_graph_layer_forward            (reference :27-42)
//   linear_layer::forward   -> gnnvc_linear_forward                 (reference :20-25)
//   ReLU / sigmoid::forward -> gnnvc_relu_forward / gnnvc_sigmoid_forward (:44-52)
//
// Compiles against this directory's headers or, for the drop-in check, against
// the reference's own include/ (the class layouts are the same).  No arithmetic
// of the forward happens in this file (an opt-in GNNVC_HOST_SIGMOID=1 applies the
// host libm's final sigmoid to the device logits on hosts whose expf differs).
#include <gnn_inference.hpp>  // angle brackets: the -I order decides (reference headers in the drop-in build)

#include <atomic>
#include <condition_variable>
#include <functional>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <mutex>
#include <random>
#include <sstream>
#include <thread>
#include <algorithm>

#include "gnnvc.h"
#include "gnnvc_host.hpp"

// ------------------------------------------------------------------ glue

namespace gnnvc_host {

int device_ordinal() {
    const char *s = std::getenv("GNNVC_DEVICE");
    return s ? std::atoi(s) : 0;
}

std::vector<int> device_list() {
    std::vector<int> out;
    const char *s = std::getenv("GNNVC_DEVICES");
    if (!s || !*s) return out;
    if (std::strchr(s, ',')) {
        for (const char *p = s; *p;) {
            out.push_back(std::atoi(p));
            const char *c = std::strchr(p, ',');
            if (!c) break;
            p = c + 1;
        }
    } else {
        const int n = std::atoi(s);
        for (int i = 0; i < n; ++i) out.push_back(i);
    }
    return out;
}

void check(int rc, const char *what, const gnnvc_engine *e) {
    if (rc == GNNVC_OK) return;
    std::fprintf(stderr, "gnnvc: %s failed: %s%s%s\n", what, gnnvc_strerror(rc),
                 (e && *gnnvc_last_error(e)) ? " — " : "", e ? gnnvc_last_error(e) : "");
    std::abort();  // never fall back to a CPU computation
}

gnnvc_engine *ops_engine() {
    static gnnvc_engine *eng = [] {
        gnnvc_engine *e = nullptr;
        static const char kEmpty[] = "ops 0 Layers\n";
        check(gnnvc_create(&e, kEmpty, sizeof kEmpty - 1, device_ordinal()), "gnnvc_create(ops)");
        return e;
    }();
    return eng;
}

}  // namespace gnnvc_host

using namespace gnn;
using gnnvc_host::check;

namespace {

template <class... Fs>
struct visitor : Fs... {
    using Fs::operator()...;
};
template <class... Fs>
visitor(Fs...) -> visitor<Fs...>;

const float *cdata(const matrix &m) { return (m.get_height() && m.get_width()) ? &*m.begin(0) : nullptr; }
float *mdata(matrix &m) { return (m.get_height() && m.get_width()) ? &*m.begin(0) : nullptr; }

unsigned pack_threads() {
    static const unsigned t = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
    return t;
}

// Persistent helpers for the pack passes: creating a dozen threads three times per predict call
// costs more than packing a mid-sized graph.  run(job) executes job(worker) on every helper and on
// the calling thread (worker 0) and returns when all are done.  Leaked on purpose: the helpers
// sleep on the condition variable until the process exits.
class PackPool {
public:
    static PackPool &get() {
        static PackPool *pool = new PackPool(pack_threads());
        return *pool;
    }
    unsigned size() const { return (unsigned)helpers_.size() + 1; }
    void run(const std::function<void(unsigned)> &job) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &job;
            pending_ = (unsigned)helpers_.size();
            ++generation_;
        }
        wake_.notify_all();
        job(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    explicit PackPool(unsigned t) {
        for (unsigned i = 1; i < t; ++i) helpers_.emplace_back([this, i] { loop(i); });
        for (auto &h : helpers_) h.detach();
    }
    void loop(unsigned worker) {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(unsigned)> *job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                wake_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                job = job_;
            }
            (*job)(worker);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> helpers_;
    std::mutex mu_;
    std::condition_variable wake_, done_;
    const std::function<void(unsigned)> *job_ = nullptr;
    unsigned pending_ = 0;
    unsigned long generation_ = 0;
};

// Run fn(part, begin, end) over [0, n) cut into equal row ranges, one per pool thread.  The cut
// depends only on n, so two passes see the same parts.
template <class Fn>
void parallel_rows(uint32_t n, Fn fn) {
    const unsigned t = pack_threads();
    if (n < (1u << 15) || t == 1) {
        fn(0u, 0u, n);
        return;
    }
    const uint32_t step = (n + t - 1) / t;
    PackPool::get().run([&](unsigned part) {
        const uint64_t lo = (uint64_t)part * step;
        if (lo < n) fn(part, (uint32_t)lo, (uint32_t)std::min<uint64_t>(n, lo + step));
    });
}

// The graph view the forward reads, packed to contiguous CSR through the
// non-mutating accessors (begin(u)/end(u), W, NW — never D(u)/g[u], which move
// the reference's hidden cursor, include/reduction_graph.hpp:144,240-245; the
// accessors used here are const and safe to call from several threads).
// Used by the layer-level graph_layer::forward.
struct PackedGraph {
    std::vector<uint64_t> rowptr;
    std::vector<uint32_t> col, w, nw;
    template <class G>
    explicit PackedGraph(const G &g) {
        const uint32_t n = g.size();
        rowptr.resize((size_t)n + 1);
        w.resize(n);
        nw.resize(n);
        rowptr[0] = 0;
        parallel_rows(n, [&](unsigned, uint32_t lo, uint32_t hi) {
            for (uint32_t u = lo; u < hi; ++u) {
                rowptr[u + 1] = (uint64_t)(g.end(u) - g.begin(u));   // row length, prefix-summed below
                w[u] = g.W(u);
                nw[u] = g.NW(u);
            }
        });
        for (uint32_t u = 0; u < n; ++u) rowptr[u + 1] += rowptr[u];
        col.resize(rowptr[n]);
        parallel_rows(n, [&](unsigned, uint32_t lo, uint32_t hi) {
            for (uint32_t u = lo; u < hi; ++u) std::copy(g.begin(u), g.end(u), col.begin() + rowptr[u]);
        });
    }
};

// predict's hand-off: the same pack, written straight into the engine's
// page-locked staging (no temporary CSR to allocate, zero and fault in; no bounce copy), with the
// column array leaving in pieces while the remaining rows are still being packed.  Returns nnz.
template <class G>
uint64_t hand_off_graph(gnnvc_engine *e, const G &g) {
    const uint32_t n = g.size();
    uint32_t *rowptr = nullptr, *col = nullptr, *w = nullptr, *nw = nullptr;
    check(gnnvc_graph_staging(e, n, 0, &rowptr, nullptr, &w, &nw), "gnnvc_graph_staging", e);
    // lengths, weights; per-part totals -> offsets -> in-place prefix sums
    std::vector<uint64_t> part_sum(pack_threads() + 1, 0);
    rowptr[0] = 0;
    parallel_rows(n, [&](unsigned part, uint32_t lo, uint32_t hi) {
        uint64_t sum = 0;
        for (uint32_t u = lo; u < hi; ++u) {
            const uint64_t len = (uint64_t)(g.end(u) - g.begin(u));
            rowptr[u + 1] = (uint32_t)len;
            sum += len;
            w[u] = g.W(u);
            nw[u] = g.NW(u);
        }
        part_sum[part + 1] = sum;
    });
    for (size_t i = 1; i < part_sum.size(); ++i) part_sum[i] += part_sum[i - 1];
    const uint64_t nnz = part_sum.back();
    if (nnz >= 0xFFFFFFFFull - 64) {
        std::fprintf(stderr, "gnn::model::predict: %llu adjacency entries do not fit the engine's 32-bit row pointers\n",
                     (unsigned long long)nnz);
        std::abort();
    }
    parallel_rows(n, [&](unsigned part, uint32_t lo, uint32_t hi) {
        uint32_t run = (uint32_t)part_sum[part];
        for (uint32_t u = lo; u < hi; ++u) {
            run += rowptr[u + 1];
            rowptr[u + 1] = run;
        }
    });
    check(gnnvc_graph_staging(e, n, nnz, nullptr, &col, nullptr, nullptr), "gnnvc_graph_staging", e);

    // columns: chunks of ~256 K entries claimed in order by the workers; this thread announces the
    // finished prefix to the engine a few chunks at a time (all HIP calls stay on this thread)
    const uint64_t kChunk = 1u << 18, kAnnounce = 1u << 21;
    const unsigned t = pack_threads();
    if (nnz < 4 * kChunk || t == 1) {
        for (uint32_t u = 0; u < n; ++u) std::copy(g.begin(u), g.end(u), col + rowptr[u]);
    } else {
        std::vector<uint32_t> cut{0};   // chunk c = rows [cut[c], cut[c+1])
        while (cut.back() < n) {
            const uint64_t target = (uint64_t)rowptr[cut.back()] + kChunk;
            uint32_t r = (uint32_t)(std::lower_bound(rowptr + cut.back() + 1, rowptr + n, target,
                                                     [](uint32_t a, uint64_t b) { return a < b; }) - rowptr);
            cut.push_back(std::min(n, std::max(r, cut.back() + 1)));
        }
        const size_t chunks = cut.size() - 1;
        std::vector<std::atomic<uint8_t>> done(chunks);
        for (auto &d : done) d.store(0, std::memory_order_relaxed);
        std::atomic<size_t> next{0};
        auto pack_one = [&]() -> bool {
            const size_t c = next.fetch_add(1, std::memory_order_relaxed);
            if (c >= chunks) return false;
            for (uint32_t u = cut[c]; u < cut[c + 1]; ++u) std::copy(g.begin(u), g.end(u), col + rowptr[u]);
            done[c].store(1, std::memory_order_release);
            return true;
        };
        auto work = [&] {
            while (pack_one()) {
            }
        };
        PackPool::get().run([&](unsigned worker) {
            if (worker) return work();
            size_t ready = 0;          // chunks [0, ready) are packed
            uint64_t sent = 0;         // entries already announced
            while (ready < chunks) {
                if (done[ready].load(std::memory_order_acquire)) {
                    ++ready;
                    const uint64_t upto = rowptr[cut[ready]];
                    if (upto - sent >= kAnnounce && ready < chunks) {
                        check(gnnvc_staged_columns_ready(e, sent, upto - sent), "gnnvc_staged_columns_ready", e);
                        sent = upto;
                    }
                } else if (pack_one()) {   // nothing to announce yet: pack a chunk too
                } else {
                    std::this_thread::yield();
                }
            }
        });
    }
    check(gnnvc_commit_staged_graph(e), "gnnvc_commit_staged_graph", e);
    return nnz;
}

// predict's hand-off by DERIVATION (GNNVC_DELTA=1|2, off by default): the device still holds the graph of
// the previous call; vertices are matched through the labels reduction_graph keeps for them (get_org_label,
// include/reduction_graph.hpp:134-139 — stable across relable_graph), the engine derives every row's surviving entries
// itself (gnnvc_derive_graph_begin) and only the tails — the fold vertices appended since — cross the bus.
// What the public read interface of reduction_graph cannot promise is that a label seen again is the SAME vertex (an
// undone fold frees its label for the next one), so mode 1 verifies: the engine's per-row hashes against hashes of the
// host lists — an O(nnz) read, about what the pack costs — and falls back to the full hand-off on any mismatch.
// Mode 2 skips the verification (measurements only).
struct DeltaState {
    std::vector<uint32_t> row_of_org;   // label -> row of the graph now resident on the device
    bool valid = false;
};
std::map<const gnnvc_engine *, DeltaState> g_delta;

template <class G>
void remember_rows(DeltaState &st, const G &g) {
    const uint32_t n = g.size();
    uint32_t top = 0;
    for (uint32_t u = 0; u < n; ++u) top = std::max<uint32_t>(top, g.get_org_label(u));
    st.row_of_org.assign(n ? (size_t)top + 1 : 0, GNNVC_NEW_VERTEX);
    for (uint32_t u = 0; u < n; ++u) st.row_of_org[g.get_org_label(u)] = u;
    st.valid = n != 0;
}

template <class G>
bool derive_hand_off(gnnvc_engine *e, const G &g, DeltaState &st, int mode, uint64_t &nnz_out) {
    if (!st.valid) return false;
    const uint32_t n = g.size();
    std::vector<uint32_t> old_row(n), rowptr((size_t)n + 1, 0), w(n), nw(n), tail(n);
    std::vector<uint64_t> part_sum(pack_threads() + 1, 0);
    parallel_rows(n, [&](unsigned part, uint32_t lo, uint32_t hi) {
        uint64_t sum = 0;
        for (uint32_t u = lo; u < hi; ++u) {
            const uint32_t org = g.get_org_label(u);
            old_row[u] = org < st.row_of_org.size() ? st.row_of_org[org] : GNNVC_NEW_VERTEX;
            const uint64_t len = (uint64_t)(g.end(u) - g.begin(u));
            rowptr[u + 1] = (uint32_t)len;
            sum += len;
            w[u] = g.W(u);
            nw[u] = g.NW(u);
        }
        part_sum[part + 1] = sum;
    });
    for (size_t i = 1; i < part_sum.size(); ++i) part_sum[i] += part_sum[i - 1];
    if (part_sum.back() >= 0xFFFFFFFFull - 64) return false;
    parallel_rows(n, [&](unsigned part, uint32_t lo, uint32_t hi) {
        uint32_t run = (uint32_t)part_sum[part];
        for (uint32_t u = lo; u < hi; ++u) {
            run += rowptr[u + 1];
            rowptr[u + 1] = run;
        }
    });
    if (gnnvc_derive_graph_begin(e, n, old_row.data(), rowptr.data(), tail.data()) != GNNVC_OK) return false;
    std::vector<uint32_t> tails;
    for (uint32_t u = 0; u < n; ++u)
        if (tail[u]) tails.insert(tails.end(), g.end(u) - tail[u], g.end(u));
    if (gnnvc_derive_graph_commit(e, tails.data(), tails.size(), w.data(), nw.data()) != GNNVC_OK) return false;
    if (mode == 1) {
        std::vector<uint64_t> dev(n), host(n);
        if (gnnvc_graph_row_hashes(e, dev.data()) != GNNVC_OK) return false;
        parallel_rows(n, [&](unsigned, uint32_t lo, uint32_t hi) {
            for (uint32_t u = lo; u < hi; ++u) {
                uint64_t h = 1469598103934665603ull;
                for (auto it = g.begin(u); it != g.end(u); ++it) h = (h ^ (uint64_t)*it) * 1099511628211ull;
                host[u] = h;
            }
        });
        if (dev != host) return false;
    }
    nnz_out = part_sum.back();
    return true;
}

// Lossless text of a model in the reference's format (9 significant digits
// round-trip fp32), used to hand the layers to gnnvc_create.
std::string exact_text(const std::string &name, const std::vector<component> &layers) {
    std::ostringstream os;
    os << std::setprecision(std::numeric_limits<float>::max_digits10);
    os << (name.empty() ? std::string("model") : name) << "\n" << layers.size() << " Layers\n";
    auto put = [&](const matrix &m) {
        os << m.get_height() << " " << m.get_width() << "\n";
        for (size_t i = 0; i < m.get_height(); ++i) {
            for (auto it = m.begin(i); it != m.end(i); ++it) os << *it << " ";
            os << "\n";
        }
    };
    for (const auto &l : layers)
        std::visit(visitor{[&](const linear_layer &c) {
                               os << "Linear_Layer\nWeights: ";
                               put(c.W);
                               os << "Bias: ";
                               put(c.bias);
                           },
                           [&](const graph_layer &) { os << "Graph_Layer\n"; },
                           [&](const ReLU &) { os << "ReLU_Activation\n"; },
                           [&](const sigmoid &) { os << "Sigmoid_Activation\n"; }},
                   l);
    return os.str();
}

// One engine per model object, rebuilt when the parameters change.
struct Bound {
    gnnvc_engine *eng = nullptr;
    std::string text;
};
std::mutex g_mu;
std::map<const void *, Bound> g_bound;

gnnvc_engine *engine_for(const void *key, const std::string &name, const std::vector<component> &layers) {
    std::string text = exact_text(name, layers);
    std::lock_guard<std::mutex> lk(g_mu);
    Bound &b = g_bound[key];
    if (b.eng && b.text == text) return b.eng;
    if (b.eng) gnnvc_destroy(b.eng);
    b.eng = nullptr;
    // GNNVC_DEVICES ("GNNVC_DEVICES=n", CLI unchanged): "4" = devices 0 .. 3 behind this one model, "0,0,1" = that
    // very list of ordinals (an ordinal may repeat: a one-GPU machine rehearsing the partitioned path).  Unset: one device.
    const std::vector<int> devs = gnnvc_host::device_list();
    if (devs.empty())
        check(gnnvc_create(&b.eng, text.data(), text.size(), gnnvc_host::device_ordinal()), "gnnvc_create(model)");
    else
        check(gnnvc_create_multi(&b.eng, text.data(), text.size(), devs.data(), (int)devs.size()), "gnnvc_create_multi(model)");
    // GNNVC_OPTIONS="key=value,key=value": engine options (include/gnnvc.h) for a driver that cannot call gnnvc_set_option itself
    if (const char *opts = std::getenv("GNNVC_OPTIONS")) {
        std::string all(opts);
        for (size_t at = 0; at < all.size();) {
            const size_t end = std::min(all.find(',', at), all.size());
            const std::string kv = all.substr(at, end - at);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos && eq > 0)
                check(gnnvc_set_option(b.eng, kv.substr(0, eq).c_str(), std::atol(kv.c_str() + eq + 1)), "gnnvc_set_option(GNNVC_OPTIONS)", b.eng);
            at = end + 1;
        }
    }
    b.text = std::move(text);
    return b.eng;
}

float scale_of(const std::vector<component> &layers) {
    for (const auto &l : layers)
        if (const auto *gl = std::get_if<graph_layer>(&l)) return gl->WEIGHT_SCALE;
    return 120.0f;
}

}  // namespace

// ------------------------------------------------------------------ layers

// Same initialisation as the reference's training-time constructor (:7-18):
// uniform(-lim, lim), lim = 1/sqrt(dim_in + 1), mt19937(seed), weights then bias.
linear_layer::linear_layer(size_t dim_in, size_t dim_out, size_t seed) : W(dim_in, dim_out), bias(1, dim_out) {
    const float lim = 1.0 / std::sqrt(dim_in + 1);
    std::mt19937 gen(seed);
    std::uniform_real_distribution<float> dist(-lim, lim);
    for (size_t i = 0; i < dim_in; ++i)
        for (size_t j = 0; j < dim_out; ++j) W(i, j) = dist(gen);
    for (size_t j = 0; j < dim_out; ++j) bias(0, j) = dist(gen);
}

void linear_layer::forward(const matrix &in, matrix &out) const {
    out.resize(in.get_height(), W.get_width());
    if (!in.get_height()) return;
    gnnvc_engine *e = gnnvc_host::ops_engine();
    check(gnnvc_linear_forward(e, (uint32_t)in.get_height(), (uint32_t)W.get_height(), (uint32_t)W.get_width(),
                               cdata(in), cdata(W), cdata(bias), mdata(out)),
          "gnnvc_linear_forward", e);
}

void graph_layer::forward(const matrix &in, matrix &out, const reduction_graph<Tn, Tw> &g) const {
    out.resize(in.get_height(), in.get_width() * 2 + 3);
    if (!in.get_height()) return;
    gnnvc_engine *e = gnnvc_host::ops_engine();
    PackedGraph pg(g);
    check(gnnvc_set_weight_scale(e, WEIGHT_SCALE), "gnnvc_set_weight_scale", e);
    check(gnnvc_upload_graph(e, g.size(), pg.rowptr.data(), pg.col.data(), pg.w.data(), pg.nw.data()),
          "gnnvc_upload_graph", e);
    check(gnnvc_graph_layer_forward(e, (uint32_t)in.get_width(), cdata(in), mdata(out)),
          "gnnvc_graph_layer_forward", e);
}

void ReLU::forward(const matrix &in, matrix &out) const {
    out.resize(in.get_height(), in.get_width());
    gnnvc_engine *e = gnnvc_host::ops_engine();
    check(gnnvc_relu_forward(e, in.get_height() * in.get_width(), cdata(in), mdata(out)), "gnnvc_relu_forward", e);
}

void sigmoid::forward(const matrix &in, matrix &out) const {
    out.resize(in.get_height(), in.get_width());
    gnnvc_engine *e = gnnvc_host::ops_engine();
    check(gnnvc_sigmoid_forward(e, in.get_height() * in.get_width(), cdata(in), mdata(out)),
          "gnnvc_sigmoid_forward", e);
}

// ------------------------------------------------------------------ model

model::model(std::string name) : name(std::move(name)) {}

void model::add_layer(const component &c) { layers.push_back(c); }

void model::set_weight_scale(float ws) {
    for (auto &l : layers)
        if (auto *gl = std::get_if<graph_layer>(&l)) gl->WEIGHT_SCALE = ws;
}

void model::predict(const matrix &in, matrix &out, const reduction_graph<Tn, Tw> &g) const {
    gnnvc_engine *e = engine_for(this, name, layers);
    const uint32_t n = g.size();
    const int ow = gnnvc_out_width(e);
    out.resize(n, (size_t)ow);
    if (n == 0) return;  // the reference calls predict on the empty graph at the end of every run
    if (in.get_height() != n || (int)in.get_width() != gnnvc_in_width(e)) {
        std::fprintf(stderr, "gnn::model::predict: input is %zux%zu, expected %ux%d\n", in.get_height(),
                     in.get_width(), n, gnnvc_in_width(e));
        std::abort();
    }
    // GNNVC_TRACE=1: one stderr line per call with the wall time of each hand-off step
    static const bool trace = [] {
        const char *v = std::getenv("GNNVC_TRACE");
        return v && *v && *v != '0';
    }();
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    check(gnnvc_set_weight_scale(e, scale_of(layers)), "gnnvc_set_weight_scale", e);
    static const int delta_mode = [] {
        const char *v = std::getenv("GNNVC_DELTA");
        return v && *v ? std::atoi(v) : 0;
    }();
    uint64_t nnz = 0;
    bool derived = false;
    if (delta_mode == 1 || delta_mode == 2) {
        DeltaState &st = g_delta[e];
        derived = derive_hand_off(e, g, st, delta_mode, nnz);
        if (!derived) nnz = hand_off_graph(e, g);
        remember_rows(st, g);
    } else {
        nnz = hand_off_graph(e, g);   // pack + copy, overlapped
    }
    const auto t2 = clk::now();
    // The scores come straight from the device: its sigmoid evaluates glibc's expf algorithm in
    // fp64 (csrc/expf_glibc.h), bit-identical to the host libm on x86-64 hosts with FMA.  Hosts
    // whose libm differs (no FMA, another libc) can set GNNVC_HOST_SIGMOID=1 to have the final
    // 1/(1+expf(-x)) applied by THEIR libm to the bit-exact device logits instead (reference :51).
    static const bool host_sigmoid = [] {
        const char *v = std::getenv("GNNVC_HOST_SIGMOID");
        return v && *v && *v != '0';
    }();
    const bool sig = host_sigmoid && !layers.empty() && std::holds_alternative<sigmoid>(layers.back());
    if (sig) in_copy.resize(n, (size_t)ow);  // the reference's scratch member: holds the logits here
    if (trace) gnnvc_set_option(e, "forward_timing", 2);   // (a forward records no events unless asked to)
    check(gnnvc_forward(e, cdata(in), mdata(out), sig ? mdata(in_copy) : nullptr), "gnnvc_forward", e);
    if (sig) {
        for (size_t i = 0; i < (size_t)n * ow; ++i) {
            const float x = *(in_copy.begin(0) + i);
            *(out.begin(0) + i) = 1.0f / (1.0f + expf(-x));
        }
    }
    if (trace) {
        const auto t3 = clk::now();
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        float dev_ms = 0, st[3] = {0, 0, 0};
        gnnvc_last_forward_ms(e, &dev_ms, st, 3);
        long longs = 0, sorted = 0;
        gnnvc_get_info(e, "long_rows", &longs);
        gnnvc_get_info(e, "sorted_tiles_active", &sorted);
        std::fprintf(stderr,
                     "gnnvc predict n=%u nnz=%llu hand-off=%.3fms%s forward=%.3fms (device %.3fms: %.3f %.3f %.3f; "
                     "long rows %ld, sorted tiles %ld)\n",
                     n, (unsigned long long)nnz, ms(t0, t2), derived ? " (derived on the device)" : "", ms(t2, t3), dev_ms, st[0],
                     st[1], st[2], longs, sorted);
    }
}

// ------------------------------------------------------------------ text format

std::ostream &gnn::operator<<(std::ostream &os, const model &m) {
    os << m.name << std::endl << m.layers.size() << " Layers" << std::endl;
    for (const auto &l : m.layers)
        std::visit(visitor{[&](const linear_layer &c) {
                               os << "Linear_Layer" << std::endl
                                  << "Weights: " << c.W << std::endl
                                  << "Bias: " << c.bias << std::endl
                                  << std::endl;
                           },
                           [&](const graph_layer &) { os << "Graph_Layer" << std::endl << std::endl; },
                           [&](const ReLU &) { os << "ReLU_Activation" << std::endl << std::endl; },
                           [&](const sigmoid &) { os << "Sigmoid_Activation" << std::endl << std::endl; }},
                   l);
    return os;
}

std::istream &gnn::operator>>(std::istream &is, model &m) {
    size_t count = 0;
    std::string word;
    is >> m.name >> count >> word;  // "<name> <n> Layers"
    while (count-- && (is >> word)) {
        if (word == "Linear_Layer") {
            linear_layer l;
            is >> word >> l.W >> word >> l.bias;  // "Weights:" matrix "Bias:" matrix
            m.layers.emplace_back(std::move(l));
        } else if (word == "Graph_Layer") {
            m.layers.emplace_back(graph_layer{});
        } else if (word == "ReLU_Activation") {
            m.layers.emplace_back(ReLU{});
        } else if (word == "Sigmoid_Activation") {
            m.layers.emplace_back(sigmoid{});
        }
    }
    return is;
}
