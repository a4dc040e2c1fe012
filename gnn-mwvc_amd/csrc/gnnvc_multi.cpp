// gnnvc_multi.cpp — several devices behind ONE handle of the C ABI (gnnvc_create_multi, include/gnnvc.h).
//
// SURVEY.md §8b's minimum surface has `gnnvc_create(..., int n_devices)` and §5 "one process driving 8 devices": the
// reference's only call site, m.predict(x, out, g) (src/GNN_VC.cpp:192, include/gnn_inference.hpp:50), is one thread in
// one process, so a drop-in that wants more than one GPU has to partition behind that call.  This file does: the graph
// handed to the front handle is cut into P contiguous row ranges of equal entry count, device r gets
// the CSR slice of its rows (global column ids) and full-size replicated feature buffers, and every device is driven by
// a HOST THREAD OF ITS OWN through the public ABI (round 4: one thread issuing for eight devices was host-bound — a stage
// of a part is a few dozen launches).  A stage of a part is computed in pieces; each piece's rows are PACKED to their
// live columns (gnnvc_pack_rows: 16 bytes a row on the metric graph instead of 64, + a short exception list — lossless,
// verified per forward) and pushed into every peer's receive buffer by ONE kernel whose stores cross the xGMI mesh
// (gnnvc_push_piece: one link per peer, no ring, no host hop, no copy call per peer), on a stream of its own so that it
// runs under the next piece's kernels; a receiver expands the pieces of all its peers with one launch per piece index
// (gnnvc_unpack_pieces).  Parts synchronise through HIP events only; the host threads only tell each other when an
// event has been recorded.  A row is still summed on one device in stored order: the bits are those of the
// single-device engine.  No reference counterpart.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/gnnvc.h"
#include "gnnvc_multi.h"

namespace gnnvc {

namespace {

constexpr int kMaxPieces = 8;
constexpr uint32_t kExcWordsPerEntry = 4;   // {row - row_lo, column, value bits, 0}

template <class T>
struct Buf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t count) {   // (the caller has made the buffer's device current)
        if (count <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t rc = hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T));
        if (rc == hipSuccess) cap = count;
        return rc;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// how the rows of one exchanged stage travel: the `kp` densest columns (mask) + an exception list of room `cap` per piece
struct Packing {
    bool known = false;   // decided for the current graph (by its first forward)
    bool on = false;      // packed; false = full 64-byte rows straight into the peers' feature buffers
    uint32_t mask = 0, kp = 0, cap = 0;
};

struct Part {
    int index = 0, device = 0;
    gnnvc_engine *eng = nullptr;
    hipStream_t stream = nullptr;   // the ENGINE'S OWN stream (its side queue was probed against it, gnnvc_create)
    hipStream_t copy = nullptr;     // pushes: under the next piece's kernels
    uint32_t lo = 0, hi = 0;
    uint64_t nnz = 0;
    int pieces = 1;
    uint32_t cut[kMaxPieces + 1] = {0};   // piece k = rows [cut[k], cut[k + 1])
    Buf<uint32_t> rowptr, col, w, nw;
    Buf<float> x, h[2], scores, logits;
    Buf<float> send[2], recv[2];          // packed pieces of stage s: mine / my peers'
    Buf<uint32_t> flag;                   // [2]: gnnvc_pack_rows' flags of stage 0 / 1
    Buf<uint32_t> flag_host;              // (page-locked) their copy for the host
    uint32_t *flag_pin = nullptr;
    hipEvent_t packed[2][kMaxPieces] = {{nullptr}};   // piece k of stage s is computed (and packed): on `stream`
    hipEvent_t pushed[2][kMaxPieces] = {{nullptr}};   // ... and has reached every peer: on `copy`
    hipEvent_t t0 = nullptr, t1 = nullptr;            // this part's span of the last forward (timing enabled)
    uint64_t counts[2][16] = {{0}};                   // first forward on a graph: non-zeros per column of my rows of stage s
    int rc = GNNVC_OK;                                // of the job at hand
    std::string err;
    float span_ms = 0.0f;
    std::thread th;
};

enum Job { kIdle = 0, kUpload = 1, kForward = 2, kQuit = 3 };

}  // namespace

struct MultiState {
    std::vector<Part> parts;
    uint32_t n = 0;
    uint64_t nnz = 0;
    bool have_graph = false;
    int stages = 0;
    double last_ms = 0.0;
    bool peer_stores = true;     // every pair of distinct devices can store into each other's memory
    // options (gnnvc_set_option on the front handle, keys "multi_*")
    int opt_pieces = 0;          // pieces per part and stage: 0 = by the number of parts
    int opt_pack = 1;            // 0 = full rows always
    int opt_push = 1;            // 0 = hipMemcpyPeerAsync per peer instead of the push kernel
    int opt_only_part = -1;      // >= 0: a forward runs THIS part's share only (timing rehearsal on one device; results are not complete)
    int opt_poison = 0;          // "poison_features": a forward starts by filling every part's feature buffers with NaN bit patterns (tests, fuzz)
    int opt_announce = -1;       // a part announces a 16-wide stage's complete input (compact table over its rows): -1 = up to 4 parts, 0 / 1
    Packing pk[2];
    std::vector<uint64_t> region_words[2];   // [r * kMaxPieces + k]: words of part r's piece k of stage s (dense + list)
    std::vector<uint64_t> region_pre[2];     // prefix over (r, k) in that order
    std::vector<uint64_t> part_words[2];     // all pieces of part r
    // the workers
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t gen = 0;
    int job = kIdle, pending = 0;
    // the job's arguments
    const uint64_t *up_rp64 = nullptr;
    const uint32_t *up_rp32 = nullptr, *up_col = nullptr, *up_w = nullptr, *up_nw = nullptr;
    const float *fw_x = nullptr;
    float *fw_scores = nullptr, *fw_logits = nullptr;
    uint64_t epoch = 0;
    std::atomic<int> failed{0};
    // rec[(r * 2 + s) * kMaxPieces + k] = epoch in which part r recorded pushed[s][k] (host-side order of record and wait)
    std::vector<std::atomic<uint64_t>> rec;
    // a barrier of the workers inside a job (first forward on a graph: the packing is chosen from every part's counts)
    std::mutex bar_mu;
    std::condition_variable bar_cv;
    int bar_count = 0;
    uint64_t bar_gen = 0;
};

namespace {

int hip_fail(std::string &err, hipError_t rc, const char *what) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(rc));
    err = buf;
    return rc == hipErrorOutOfMemory ? GNNVC_ERR_NOMEM : GNNVC_ERR_DEVICE;
}

#define MTRY(call)                                                  \
    do {                                                            \
        hipError_t rc_ = (call);                                    \
        if (rc_ != hipSuccess) return hip_fail(err, rc_, #call);    \
    } while (0)

int part_fail(std::string &err, const Part &p, int rc, const char *what) {
    char buf[512];
    snprintf(buf, sizeof buf, "device %d: %s: %s", p.device, what, p.eng ? gnnvc_last_error(p.eng) : "");
    err = buf;
    return rc;
}

#define PTRY(p, call, what)                                           \
    do {                                                              \
        int rc_ = (call);                                             \
        if (rc_ != GNNVC_OK) return part_fail(err, (p), rc_, what);   \
    } while (0)

// every worker of the job meets here; false = the job has failed somewhere (nobody waits for a part that gave up)
bool barrier(MultiState *m, int parties) {
    std::unique_lock<std::mutex> lk(m->bar_mu);
    const uint64_t g = m->bar_gen;
    if (++m->bar_count == parties) {
        m->bar_count = 0;
        ++m->bar_gen;
        m->bar_cv.notify_all();
    } else {
        while (m->bar_gen == g && !m->failed.load(std::memory_order_acquire)) m->bar_cv.wait_for(lk, std::chrono::milliseconds(2));
    }
    return !m->failed.load(std::memory_order_acquire);
}

void fail_job(MultiState *m) {
    m->failed.store(1, std::memory_order_release);
    std::lock_guard<std::mutex> lk(m->bar_mu);
    m->bar_cv.notify_all();
}

int pieces_for(const MultiState *m) {
    const int P = (int)m->parts.size();
    if (P <= 1) return 1;
    if (m->opt_pieces > 0) return std::min(m->opt_pieces, kMaxPieces);
    // up to 4 parts a part's rows are many enough for the per-rank plans (LDS table over the slice, announced compact table:
    // whole-range calls — measured per-rank compute, metric graph: P = 4 2.02 ms on the plans against 2.88 in plain pieces);
    // beyond, four plain pieces whose pushes run under the next piece's kernels (P = 8: 1.48 plain against 1.67 on the plans)
    return P <= 4 ? 1 : 4;
}

// ---- the packing of one exchanged stage, from the non-zeros per column of ALL rows (every part computes the same choice)
Packing choose_packing(const uint64_t (&counts)[16], uint32_t n, int total_pieces, bool allow) {
    Packing pk;
    pk.known = true;
    if (!allow || n == 0) return pk;
    int order[16];
    for (int c = 0; c < 16; ++c) order[c] = c;
    std::stable_sort(order, order + 16, [&](int a, int b) { return counts[a] > counts[b]; });
    double best_cost = 1e30;
    uint32_t best_kp = 0;
    uint64_t best_rest = 0;
    for (uint32_t kp = 4; kp <= 12; kp += 4) {
        uint64_t rest = 0;
        for (uint32_t i = kp; i < 16; ++i) rest += counts[order[i]];
        const double cost = 4.0 * kp + 16.0 * (double)rest / (double)n;   // bytes per row: dense part + its share of the list
        if (cost < best_cost) {
            best_cost = cost;
            best_kp = kp;
            best_rest = rest;
        }
    }
    if (best_cost > 48.0) return pk;   // (nothing beats three quarters of a full row: ship full rows)
    pk.on = true;
    pk.kp = best_kp;
    for (uint32_t i = 0; i < best_kp; ++i)
        if (counts[order[i]] > 0) pk.mask |= 1u << order[i];
    pk.cap = (uint32_t)std::min<uint64_t>(2 * best_rest / (uint64_t)std::max(total_pieces, 1) + 1024, 1u << 28);   // twice a piece's expected share, plus slack
    return pk;
}

uint64_t piece_words(const Packing &pk, uint32_t rows) { return (uint64_t)rows * pk.kp + 4 + (uint64_t)kExcWordsPerEntry * pk.cap; }

// region of part r's piece k inside the receive buffer of part q (q != r): the pieces of all parts but q in (part, piece) order
uint64_t recv_offset(const MultiState *m, int s, int q, int r, int k) {
    const uint64_t at = m->region_pre[s][(size_t)r * kMaxPieces + k];
    return r > q ? at - m->part_words[s][(size_t)q] : at;
}

void layout_stage(MultiState *m, int s) {
    const size_t P = m->parts.size();
    m->region_words[s].assign(P * kMaxPieces, 0);
    m->region_pre[s].assign(P * kMaxPieces + 1, 0);
    m->part_words[s].assign(P, 0);
    uint64_t at = 0;
    for (size_t r = 0; r < P; ++r)
        for (int k = 0; k < kMaxPieces; ++k) {
            const Part &p = m->parts[r];
            const uint64_t wds = (m->pk[s].on && k < p.pieces) ? piece_words(m->pk[s], p.cut[k + 1] - p.cut[k]) : 0;
            m->region_words[s][r * kMaxPieces + k] = wds;
            m->region_pre[s][r * kMaxPieces + k] = at;
            at += wds;
            m->part_words[s][r] += wds;
        }
    m->region_pre[s][P * kMaxPieces] = at;
}

// ------------------------------------------------------------------ one part's share of a hand-off
int upload_part(MultiState *m, Part &p, std::string &err) {
    const uint32_t n = m->n;
    auto rp = [&](uint32_t u) -> uint64_t { return m->up_rp64 ? m->up_rp64[u] : (uint64_t)m->up_rp32[u]; };
    const uint32_t rows = p.hi - p.lo;
    const uint64_t first = n ? rp(p.lo) : 0;
    // (the slices' device-side checks see relative row pointers: every part checks its own range of the global array)
    for (uint32_t u = p.lo; u < p.hi; ++u)
        if (rp(u) > rp(u + 1)) {
            err = "row pointers are not monotone from 0 to nnz";
            return GNNVC_ERR_INVALID;
        }
    MTRY(hipSetDevice(p.device));
    MTRY(p.rowptr.reserve((size_t)rows + 1));
    MTRY(p.col.reserve((size_t)p.nnz + GNNVC_COL_PAD));
    MTRY(p.w.reserve(std::max<uint32_t>(rows, 1u)));
    MTRY(p.nw.reserve(std::max<uint32_t>(rows, 1u)));
    const size_t frows = (size_t)n + 1;
    MTRY(p.x.reserve(frows));
    MTRY(p.h[0].reserve(frows * 16));
    MTRY(p.h[1].reserve(frows * 16));
    MTRY(p.scores.reserve(frows));
    MTRY(p.logits.reserve(frows));
    MTRY(p.flag.reserve(2));
    if (!p.flag_pin) MTRY(hipHostMalloc(reinterpret_cast<void **>(&p.flag_pin), 4 * sizeof(uint32_t), hipHostMallocDefault));
    std::vector<uint32_t> local((size_t)rows + 1);
    for (uint32_t i = 0; i <= rows; ++i) local[i] = (uint32_t)(rp(p.lo + i) - first);
    MTRY(hipMemcpyAsync(p.rowptr.p, local.data(), ((size_t)rows + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, p.stream));
    if (p.nnz) MTRY(hipMemcpyAsync(p.col.p, m->up_col + first, (size_t)p.nnz * sizeof(uint32_t), hipMemcpyHostToDevice, p.stream));
    MTRY(hipMemsetAsync(p.col.p + p.nnz, 0, GNNVC_COL_PAD * sizeof(uint32_t), p.stream));
    if (rows) {
        MTRY(hipMemcpyAsync(p.w.p, m->up_w + p.lo, (size_t)rows * sizeof(uint32_t), hipMemcpyHostToDevice, p.stream));
        MTRY(hipMemcpyAsync(p.nw.p, m->up_nw + p.lo, (size_t)rows * sizeof(uint32_t), hipMemcpyHostToDevice, p.stream));
    }
    // the pad rows of the 16-wide feature buffers read as zero: no kernel ever writes them, once per graph will do
    for (auto &h : p.h) MTRY(hipMemsetAsync(h.p + (size_t)n * 16, 0, 16 * sizeof(float), p.stream));
    MTRY(hipStreamSynchronize(p.stream));   // (`local` leaves scope; the attach below reads the arrays on the same stream anyway)
    if (n == 0) return GNNVC_OK;
    PTRY(p, gnnvc_attach_graph_slice(p.eng, n, p.lo, p.hi, p.nnz, p.rowptr.p, p.col.p, p.w.p, p.nw.p), "gnnvc_attach_graph_slice");
    return GNNVC_OK;
}

// ------------------------------------------------------------------ one part's share of a forward
// wait (host side) until part r has RECORDED pushed[s][k] in this forward, then make `stream` wait for it
int wait_piece(MultiState *m, Part &me, int r, int s, int k, std::string &err) {
    std::atomic<uint64_t> &cell = m->rec[((size_t)r * 2 + s) * kMaxPieces + k];
    int spins = 0;
    while (cell.load(std::memory_order_acquire) != m->epoch) {
        if (m->failed.load(std::memory_order_acquire)) {
            err = "another part of the forward failed";
            return GNNVC_ERR_STATE;
        }
        if (++spins > 64) std::this_thread::yield();
    }
    MTRY(hipStreamWaitEvent(me.stream, m->parts[(size_t)r].pushed[s][k], 0));
    return GNNVC_OK;
}

// the rows of piece k of stage s that `me` just computed into `out`: on their way to every peer
int ship_piece(MultiState *m, Part &me, int s, int k, float *out, std::string &err) {
    const size_t P = m->parts.size();
    const Packing &pk = m->pk[s];
    const uint32_t r0 = me.cut[k], r1 = me.cut[k + 1], rows = r1 - r0;
    const bool solo = m->opt_only_part >= 0;
    if (pk.on) {
        float *region = me.send[s].p + (m->region_pre[s][(size_t)me.index * kMaxPieces + k] - m->region_pre[s][(size_t)me.index * kMaxPieces]);
        uint32_t *exc = reinterpret_cast<uint32_t *>(region + (size_t)rows * pk.kp);
        PTRY(me, gnnvc_pack_rows(me.eng, out, 16, r0, r1, pk.mask, pk.kp, region, exc, pk.cap, me.flag.p + s), "gnnvc_pack_rows");
        MTRY(hipEventRecord(me.packed[s][k], me.stream));
        MTRY(hipStreamWaitEvent(me.copy, me.packed[s][k], 0));
        if (m->opt_push && m->peer_stores) {
            float *dst[64];
            uint32_t nd = 0;
            for (size_t q = 0; q < P; ++q)
                if ((int)q != me.index) dst[nd++] = m->parts[q].recv[s].p + recv_offset(m, s, (int)q, me.index, k);
            PTRY(me, gnnvc_push_piece(me.eng, region, rows, pk.kp, pk.cap, nd, dst, me.copy), "gnnvc_push_piece");
        } else {
            for (size_t q = 0; q < P; ++q)
                if ((int)q != me.index)
                    MTRY(hipMemcpyPeerAsync(m->parts[q].recv[s].p + recv_offset(m, s, (int)q, me.index, k), m->parts[q].device, region,
                                            me.device, (size_t)m->region_words[s][(size_t)me.index * kMaxPieces + k] * sizeof(float), me.copy));
        }
    } else {
        MTRY(hipEventRecord(me.packed[s][k], me.stream));
        MTRY(hipStreamWaitEvent(me.copy, me.packed[s][k], 0));
        if (!solo)   // full rows, straight into every peer's feature buffer
            for (size_t q = 0; q < P; ++q)
                if ((int)q != me.index)
                    MTRY(hipMemcpyPeerAsync(m->parts[q].h[s & 1].p + (size_t)r0 * 16, m->parts[q].device, out + (size_t)r0 * 16, me.device,
                                            (size_t)rows * 16 * sizeof(float), me.copy));
    }
    MTRY(hipEventRecord(me.pushed[s][k], me.copy));
    m->rec[((size_t)me.index * 2 + s) * kMaxPieces + k].store(m->epoch, std::memory_order_release);
    return GNNVC_OK;
}

int forward_part(MultiState *m, Part &me, std::string &err) {
    const uint32_t n = m->n;
    const size_t P = m->parts.size();
    const int dev0 = m->parts[0].device;
    const bool solo = m->opt_only_part >= 0;
    if (solo && m->opt_only_part != me.index) return GNNVC_OK;
    MTRY(hipSetDevice(me.device));
    MTRY(hipEventRecord(me.t0, me.stream));
    MTRY(hipMemsetAsync(me.flag.p, 0, 2 * sizeof(uint32_t), me.stream));
    MTRY(hipMemcpyPeerAsync(me.x.p, me.device, m->fw_x, dev0, (size_t)n * sizeof(float), me.stream));   // the input, replicated
    for (int s = 0; s < 3; ++s) {
        const float *in = s == 0 ? me.x.p : me.h[(s - 1) & 1].p;
        float *out = s == 2 ? me.scores.p : me.h[s & 1].p;
        if (s > 0 && P > 1) {
            // this stage's input is complete on this device once every peer's pieces of the last stage have landed (and been expanded)
            const Packing &pk = m->pk[s - 1];
            for (int k = 0; k < kMaxPieces; ++k) {
                gnnvc_piece list[64];
                uint32_t nl = 0;
                for (size_t r = 0; r < P; ++r) {
                    const Part &pr = m->parts[r];
                    if ((int)r == me.index || k >= pr.pieces) continue;
                    if (!solo) {
                        int rc = wait_piece(m, me, (int)r, s - 1, k, err);
                        if (rc) return rc;
                    }
                    if (pk.on && pr.cut[k + 1] > pr.cut[k])
                        list[nl++] = gnnvc_piece{me.recv[s - 1].p + recv_offset(m, s - 1, me.index, (int)r, k), pr.cut[k], pr.cut[k + 1]};
                }
                if (nl) PTRY(me, gnnvc_unpack_pieces(me.eng, list, nl, pk.cap, 16, pk.mask, pk.kp, me.h[(s - 1) & 1].p), "gnnvc_unpack_pieces");
            }
        }
        if (me.hi > me.lo && s >= 1 && me.pieces == 1 && (m->opt_announce < 0 ? P <= 4 : m->opt_announce != 0))
            PTRY(me, gnnvc_stage_input_ready(me.eng, s, in, me.lo, me.hi), "gnnvc_stage_input_ready");
        const bool choose = s < 2 && P > 1 && !m->pk[s].known;
        for (int k = 0; k < me.pieces; ++k) {
            const uint32_t r0 = me.cut[k], r1 = me.cut[k + 1];
            if (r1 > r0)
                PTRY(me, gnnvc_stage_forward_device(me.eng, s, r0, r1, in, out, s == 2 ? me.logits.p : nullptr), "gnnvc_stage_forward_device");
            if (s < 2 && P > 1 && !choose) {
                int rc = ship_piece(m, me, s, k, out, err);
                if (rc) return rc;
            }
        }
        if (choose) {
            // A graph's first forward: how this stage's rows travel is chosen from the non-zeros per column of ALL rows — every
            // part counts its own (a pass over the rows it just wrote), the counts meet on the host, every worker makes the same
            // choice — and only then do the pieces leave (packed, but one after the other: the later forwards pipeline)
            uint64_t cnt[16] = {0};
            if (me.hi > me.lo) PTRY(me, gnnvc_column_counts(me.eng, out + (size_t)me.lo * 16, me.hi - me.lo, 16, cnt), "gnnvc_column_counts");
            for (int c = 0; c < 16; ++c) me.counts[s][c] = cnt[c];
            if (!barrier(m, (int)P)) {
                err = "another part of the forward failed";
                return GNNVC_ERR_STATE;
            }
            if (me.index == 0) {
                uint64_t tot[16] = {0};
                int total_pieces = 0;
                for (const Part &p : m->parts) {
                    for (int c = 0; c < 16; ++c) tot[c] += p.counts[s][c];
                    total_pieces += p.pieces;
                }
                m->pk[s] = choose_packing(tot, n, total_pieces, m->opt_pack != 0);
                layout_stage(m, s);
            }
            if (!barrier(m, (int)P)) {
                err = "another part of the forward failed";
                return GNNVC_ERR_STATE;
            }
            if (m->pk[s].on) {
                MTRY(me.send[s].reserve((size_t)m->part_words[s][(size_t)me.index]));
                MTRY(me.recv[s].reserve((size_t)(m->region_pre[s][P * kMaxPieces] - m->part_words[s][(size_t)me.index])));
            }
            if (!barrier(m, (int)P)) {   // (every receive buffer exists before anybody pushes into it)
                err = "another part of the forward failed";
                return GNNVC_ERR_STATE;
            }
            for (int k = 0; k < me.pieces; ++k) {
                int rc = ship_piece(m, me, s, k, out, err);
                if (rc) return rc;
            }
        }
        if (s == 2 && me.hi > me.lo && !solo) {   // the scores (and logits) of these rows, to the caller's arrays on the first device
            const size_t rows = me.hi - me.lo;
            MTRY(hipMemcpyPeerAsync(m->fw_scores + me.lo, dev0, me.scores.p + me.lo, me.device, rows * sizeof(float), me.stream));
            if (m->fw_logits)
                MTRY(hipMemcpyPeerAsync(m->fw_logits + me.lo, dev0, me.logits.p + me.lo, me.device, rows * sizeof(float), me.stream));
        }
    }
    MTRY(hipMemcpyAsync(me.flag_pin, me.flag.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, me.stream));
    for (int s = 0; s < 2; ++s)   // (the span ends when the last push has landed; an event never recorded waits for nothing)
        MTRY(hipStreamWaitEvent(me.stream, me.pushed[s][std::max(me.pieces - 1, 0)], 0));
    MTRY(hipEventRecord(me.t1, me.stream));
    return GNNVC_OK;
}

// whatever happened: nothing of this part's work is left in flight when the job returns (a failed forward must not
// leave kernels and peer copies writing into the caller's arrays behind an error return)
void drain_part(Part &p) {
    if (hipSetDevice(p.device) != hipSuccess) return;
    if (p.copy) (void)hipStreamSynchronize(p.copy);
    if (p.stream) (void)hipStreamSynchronize(p.stream);
    (void)hipGetLastError();
}

void worker(MultiState *m, Part *p) {
    uint64_t seen = 0;
    for (;;) {
        int job;
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->cv_go.wait(lk, [&] { return m->gen != seen; });
            seen = m->gen;
            job = m->job;
        }
        if (job == kQuit) return;
        p->err.clear();
        p->rc = GNNVC_OK;
        if (job == kUpload) p->rc = upload_part(m, *p, p->err);
        else if (job == kForward) p->rc = forward_part(m, *p, p->err);
        if (p->rc != GNNVC_OK) fail_job(m);
        if (job == kForward) {
            drain_part(*p);
            if (p->rc == GNNVC_OK && p->t0 && p->t1 && (m->opt_only_part < 0 || m->opt_only_part == p->index))
                if (hipEventElapsedTime(&p->span_ms, p->t0, p->t1) != hipSuccess) p->span_ms = 0.0f;
        }
        {
            std::lock_guard<std::mutex> lk(m->mu);
            if (--m->pending == 0) m->cv_done.notify_all();
        }
    }
}

// run one job on every worker; the first failure's code and message
int run_job(MultiState *m, int job, std::string &err) {
    m->failed.store(0, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->job = job;
        m->pending = (int)m->parts.size();
        ++m->gen;
    }
    m->cv_go.notify_all();
    {
        std::unique_lock<std::mutex> lk(m->mu);
        m->cv_done.wait(lk, [&] { return m->pending == 0; });
        m->job = kIdle;
    }
    int rc = GNNVC_OK;
    for (const Part &p : m->parts)   // (a part that only gave up because another one failed comes last)
        if (p.rc != GNNVC_OK && p.rc != GNNVC_ERR_STATE) {
            rc = p.rc;
            err = p.err;
            break;
        }
    if (rc == GNNVC_OK)
        for (const Part &p : m->parts)
            if (p.rc != GNNVC_OK) {
                rc = p.rc;
                err = p.err;
                break;
            }
    return rc;
}

}  // namespace

int multi_create(MultiState **out, const char *model_text, size_t len, const int *devices, int n_devices, std::string &err) {
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) {
        err = "gnnvc_create_multi: 1 .. 64 devices";
        return GNNVC_ERR_INVALID;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        err = "no HIP device — this engine has no CPU path";
        return GNNVC_ERR_DEVICE;
    }
    for (int r = 0; r < n_devices; ++r)
        if (devices[r] < 0 || devices[r] >= count) {
            char buf[128];
            snprintf(buf, sizeof buf, "no HIP device %d (found %d)", devices[r], count);
            err = buf;
            return GNNVC_ERR_DEVICE;
        }
    MultiState *m = new (std::nothrow) MultiState();
    if (!m) return GNNVC_ERR_NOMEM;
    m->parts.resize((size_t)n_devices);
    m->rec = std::vector<std::atomic<uint64_t>>((size_t)n_devices * 2 * kMaxPieces);
    for (auto &c : m->rec) c.store(0);
    int rc = GNNVC_OK;
    for (int r = 0; r < n_devices && rc == GNNVC_OK; ++r) {
        Part &p = m->parts[(size_t)r];
        p.index = r;
        p.device = devices[r];
        rc = gnnvc_create(&p.eng, model_text, len, p.device);
        if (rc != GNNVC_OK) {
            err = "creating a per-device engine failed";
            break;
        }
        hipError_t h = hipSetDevice(p.device);
        // the part runs on its engine's OWN stream: the engine's side queue was probed against that one (ADVICE r3 — a stream
        // made here could share the side queue's hardware queue and serialise the long rows with the tile kernel)
        void *own = nullptr;
        if (h == hipSuccess && gnnvc_get_stream(p.eng, &own) != GNNVC_OK) h = hipErrorInvalidValue;
        p.stream = reinterpret_cast<hipStream_t>(own);
        if (h == hipSuccess) h = hipStreamCreateWithFlags(&p.copy, hipStreamNonBlocking);
        for (int s = 0; s < 2 && h == hipSuccess; ++s)
            for (int k = 0; k < kMaxPieces && h == hipSuccess; ++k) {
                h = hipEventCreateWithFlags(&p.packed[s][k], hipEventDisableTiming);
                if (h == hipSuccess) h = hipEventCreateWithFlags(&p.pushed[s][k], hipEventDisableTiming);
            }
        if (h == hipSuccess) h = hipEventCreate(&p.t0);
        if (h == hipSuccess) h = hipEventCreate(&p.t1);
        if (h != hipSuccess) {
            rc = hip_fail(err, h, "stream / event creation");
            break;
        }
        // rows travel device to device: let each device write its peers' memory directly where the fabric allows it (without
        // peer access the pieces go by hipMemcpyPeerAsync, staged by the runtime)
        for (int q = 0; q < r; ++q) {
            const int a = p.device, b = m->parts[(size_t)q].device;
            if (a == b) continue;
            int ok_ab = 0, ok_ba = 0;
            if (hipDeviceCanAccessPeer(&ok_ab, a, b) == hipSuccess && ok_ab && hipDeviceCanAccessPeer(&ok_ba, b, a) == hipSuccess && ok_ba) {
                (void)hipSetDevice(a);
                (void)hipDeviceEnablePeerAccess(b, 0);
                (void)hipSetDevice(b);
                (void)hipDeviceEnablePeerAccess(a, 0);
                (void)hipGetLastError();   // ("already enabled" is not an error worth keeping)
            } else {
                m->peer_stores = false;
            }
        }
    }
    if (rc == GNNVC_OK) {
        m->stages = gnnvc_num_stages(m->parts[0].eng);
        if (m->stages != 3 || !gnnvc_is_fused(m->parts[0].eng)) {
            err = "a multi-device handle runs the fused three-stage model only";
            rc = GNNVC_ERR_UNSUPPORTED;
        }
    }
    if (rc != GNNVC_OK) {
        multi_destroy(m);
        return rc;
    }
    for (Part &p : m->parts) p.th = std::thread(worker, m, &p);
    *out = m;
    return GNNVC_OK;
}

void multi_destroy(MultiState *m) {
    if (!m) return;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->job = kQuit;
        ++m->gen;
    }
    m->cv_go.notify_all();
    for (Part &p : m->parts)
        if (p.th.joinable()) p.th.join();
    for (Part &p : m->parts) {
        (void)hipSetDevice(p.device);
        if (p.copy) (void)hipStreamSynchronize(p.copy);
        if (p.stream) (void)hipStreamSynchronize(p.stream);
        p.rowptr.release(); p.col.release(); p.w.release(); p.nw.release();
        p.x.release(); p.h[0].release(); p.h[1].release(); p.scores.release(); p.logits.release();
        for (auto &b : p.send) b.release();
        for (auto &b : p.recv) b.release();
        p.flag.release();
        if (p.flag_pin) (void)hipHostFree(p.flag_pin);
        for (auto &row : p.packed)
            for (auto &ev : row)
                if (ev) (void)hipEventDestroy(ev);
        for (auto &row : p.pushed)
            for (auto &ev : row)
                if (ev) (void)hipEventDestroy(ev);
        if (p.t0) (void)hipEventDestroy(p.t0);
        if (p.t1) (void)hipEventDestroy(p.t1);
        if (p.copy) (void)hipStreamDestroy(p.copy);
        if (p.eng) gnnvc_destroy(p.eng);   // (owns p.stream)
    }
    delete m;
}

int multi_devices(const MultiState *m) { return m ? (int)m->parts.size() : 0; }
uint32_t multi_vertices(const MultiState *m) { return (m && m->have_graph) ? m->n : 0u; }
bool multi_has_graph(const MultiState *m) { return m && m->have_graph; }
double multi_last_forward_ms(const MultiState *m) { return m ? m->last_ms : 0.0; }

int multi_part_info(const MultiState *m, int part, uint32_t *row_lo, uint32_t *row_hi, uint64_t *entries) {
    if (!m || part < 0 || part >= (int)m->parts.size()) return GNNVC_ERR_INVALID;
    const Part &p = m->parts[(size_t)part];
    if (row_lo) *row_lo = p.lo;
    if (row_hi) *row_hi = p.hi;
    if (entries) *entries = p.nnz;
    return GNNVC_OK;
}

// what part r shipped to ONE peer in the last exchange of stage s: the dense part of every piece + the used part of its list
static uint64_t shipped_bytes(MultiState *m, const Part &p, int s) {
    if (m->parts.size() < 2) return 0;
    if (!m->pk[s].on) return (uint64_t)(p.hi - p.lo) * 16 * sizeof(float);
    uint64_t bytes = 0;
    if (hipSetDevice(p.device) != hipSuccess) return 0;
    for (int k = 0; k < p.pieces; ++k) {
        const uint32_t rows = p.cut[k + 1] - p.cut[k];
        const float *region = p.send[s].p + (m->region_pre[s][(size_t)p.index * kMaxPieces + k] - m->region_pre[s][(size_t)p.index * kMaxPieces]);
        uint32_t used = 0;
        if (hipMemcpy(&used, region + (size_t)rows * m->pk[s].kp, sizeof used, hipMemcpyDeviceToHost) != hipSuccess) return 0;
        bytes += ((uint64_t)rows * m->pk[s].kp + 4 + (uint64_t)kExcWordsPerEntry * std::min(used, m->pk[s].cap)) * sizeof(float);
    }
    return bytes;
}

bool multi_get_info(MultiState *m, const char *key, long *value) {
    const std::string k(key);
    if (k == "multi_pieces") *value = m->parts.empty() ? 0 : m->parts[0].pieces;
    else if (k == "multi_packed_stage0" || k == "multi_packed_stage1") *value = m->pk[k.back() - '0'].known && m->pk[k.back() - '0'].on ? 1 : 0;
    else if (k == "multi_packed_columns_stage0" || k == "multi_packed_columns_stage1") *value = m->pk[k.back() - '0'].on ? (long)m->pk[k.back() - '0'].kp : 16;
    else if (k == "multi_exchange_bytes_per_peer_stage0" || k == "multi_exchange_bytes_per_peer_stage1") {
        // (diagnostic: reads the pieces' list counts back; the busiest part's figure)
        uint64_t most = 0;
        if (m->have_graph && m->pk[k.back() - '0'].known)
            for (const Part &p : m->parts) most = std::max(most, shipped_bytes(m, p, k.back() - '0'));
        (void)hipSetDevice(m->parts[0].device);
        *value = (long)most;
    }
    else if (k == "multi_peer_stores") *value = m->peer_stores && m->opt_push ? 1 : 0;
    else if (k.rfind("multi_part_span_us_", 0) == 0) {
        const int r = atoi(k.c_str() + 19);
        if (r < 0 || r >= (int)m->parts.size()) return false;
        *value = (long)(m->parts[(size_t)r].span_ms * 1000.0f);
    } else return false;
    return true;
}

int multi_set_weight_scale(MultiState *m, float ws) {
    for (Part &p : m->parts) {
        int rc = gnnvc_set_weight_scale(p.eng, ws);
        if (rc) return rc;
    }
    return GNNVC_OK;
}

int multi_set_option(MultiState *m, const char *key, long value) {
    const std::string k(key);
    if (k.rfind("multi_", 0) == 0) {   // the exchange's own options
        if (k == "multi_pieces") m->opt_pieces = value < 0 ? 0 : (int)std::min<long>(value, kMaxPieces);
        else if (k == "multi_pack") m->opt_pack = value != 0 ? 1 : 0;
        else if (k == "multi_push") m->opt_push = value != 0 ? 1 : 0;
        else if (k == "multi_only_part") m->opt_only_part = (value >= 0 && value < (long)m->parts.size()) ? (int)value : -1;
        else if (k == "multi_announce") m->opt_announce = value < 0 ? -1 : (value != 0 ? 1 : 0);
        else return GNNVC_ERR_INVALID;
        if (k == "multi_pieces" || k == "multi_pack") {   // (they shape the pieces and the receive buffers: decided again on the next graph / forward)
            m->pk[0] = m->pk[1] = Packing();
            if (m->have_graph) {
                const int K = pieces_for(m);
                for (Part &p : m->parts) {
                    p.pieces = K;
                    for (int kk = 0; kk <= K; ++kk) {
                        const uint64_t at = p.lo + (uint64_t)(p.hi - p.lo) * kk / K;
                        p.cut[kk] = kk == K ? p.hi : (uint32_t)std::min<uint64_t>(p.hi, std::max<uint64_t>(p.lo, at / 64u * 64u));
                    }
                    for (int kk = 1; kk < K; ++kk) p.cut[kk] = std::max(p.cut[kk], p.cut[kk - 1]);
                }
            }
        }
        return GNNVC_OK;
    }
    if (k == "poison_features") {   // (the parts run stage by stage on THIS driver's buffers)
        m->opt_poison = value != 0 ? 1 : 0;
        return GNNVC_OK;
    }
    for (Part &p : m->parts) {
        int rc = gnnvc_set_option(p.eng, key, value);
        if (rc) return rc;
    }
    return GNNVC_OK;
}

int multi_upload(MultiState *m, uint32_t n, const uint64_t *rowptr64, const uint32_t *rowptr32, const uint32_t *col, const uint32_t *w,
                 const uint32_t *nw, std::string &err) {
    m->have_graph = false;
    m->pk[0] = m->pk[1] = Packing();
    const size_t P = m->parts.size();
    auto rp = [&](uint32_t u) -> uint64_t { return rowptr64 ? rowptr64[u] : (uint64_t)rowptr32[u]; };
    const uint64_t nnz = n ? rp(n) : 0;
    if (nnz >= 0xFFFFFFFFull - GNNVC_COL_PAD) {
        err = "nnz does not fit 32-bit row pointers";
        return GNNVC_ERR_UNSUPPORTED;
    }
    if (n && rp(0) != 0) {
        err = "row pointers are not monotone from 0 to nnz";
        return GNNVC_ERR_INVALID;
    }
    // contiguous row ranges of (nearly) equal entry count, cut at multiples of 64 rows (the tile kernels' unit).  (The cuts are
    // searched in an array not yet known to be monotone — every part checks its own range, upload_part — a broken one yields
    // uneven parts and then the error.)
    std::vector<uint32_t> cut(P + 1, 0);
    cut[P] = n;
    for (size_t r = 1; r < P; ++r) {
        uint32_t at;
        if (nnz == 0) {
            at = (uint32_t)((uint64_t)n * r / P);
        } else {
            const uint64_t target = nnz / P * r + std::min<uint64_t>(r, nnz % P);
            uint32_t a = 0, b = n;   // first row whose offset reaches the target
            while (a < b) {
                const uint32_t mid = a + (b - a) / 2;
                if (rp(mid) >= target) b = mid;
                else a = mid + 1;
            }
            at = a;
        }
        at = at / 64u * 64u;
        cut[r] = std::min(n, std::max(at, cut[r - 1]));
    }
    m->n = n;
    m->nnz = nnz;
    const int K = pieces_for(m);
    for (size_t r = 0; r < P; ++r) {
        Part &p = m->parts[r];
        p.lo = cut[r];
        p.hi = cut[r + 1];
        const uint64_t a = n ? rp(p.lo) : 0, b = n ? rp(p.hi) : 0;
        if (b < a || b - a > nnz) {
            err = "row pointers are not monotone from 0 to nnz";
            return GNNVC_ERR_INVALID;
        }
        p.nnz = b - a;
        p.pieces = K;
        for (int k = 0; k <= K; ++k) {
            const uint64_t at = p.lo + (uint64_t)(p.hi - p.lo) * k / K;
            p.cut[k] = k == K ? p.hi : (uint32_t)std::min<uint64_t>(p.hi, std::max<uint64_t>(p.lo, at / 64u * 64u));
        }
        for (int k = 1; k < K; ++k) p.cut[k] = std::max(p.cut[k], p.cut[k - 1]);
    }
    m->up_rp64 = rowptr64;
    m->up_rp32 = rowptr32;
    m->up_col = col;
    m->up_w = w;
    m->up_nw = nw;
    const int rc = run_job(m, kUpload, err);   // every part copies, checks and attaches its own slice, side by side
    m->up_rp64 = nullptr;
    m->up_rp32 = m->up_col = m->up_w = m->up_nw = nullptr;
    if (rc != GNNVC_OK) return rc;
    m->have_graph = true;
    return GNNVC_OK;
}

int multi_forward_device(MultiState *m, const float *d_x, float *d_scores, float *d_logits, std::string &err) {
    if (!m->have_graph) {
        err = "no graph attached";
        return GNNVC_ERR_STATE;
    }
    if (m->n == 0) return GNNVC_OK;
    if (m->opt_only_part >= 0 && m->parts.size() > 1 && !(m->pk[0].known && m->pk[1].known)) {
        err = "multi_only_part needs a complete forward on this graph first";
        return GNNVC_ERR_STATE;
    }
    const auto t0 = std::chrono::steady_clock::now();
    m->fw_x = d_x;
    m->fw_scores = d_scores;
    m->fw_logits = d_logits;
    int rc = GNNVC_OK;
    // (every repeat takes the packing off at least one of the two exchanged stages, so the third attempt ships full rows everywhere.
    // Two attempts were one too few: a stage-0 overflow hands stage 1 a damaged input in the first attempt, so stage 1's own overflow
    // only shows in the second — fuzz_multi.py case 232 returned that attempt's lossy rows.)
    for (int attempt = 0; attempt < 3; ++attempt) {
        if (m->opt_poison && m->opt_only_part < 0) {   // (tests, fuzz; before any part starts: peers write into these buffers on their own streams)
            for (Part &p : m->parts) {
                if (hipSetDevice(p.device) != hipSuccess) continue;
                for (int b = 0; b < 2; ++b)
                    if (p.h[b].p) (void)hipMemsetAsync(p.h[b].p, 0xFF, (size_t)m->n * 16 * sizeof(float), p.stream);
                (void)hipStreamSynchronize(p.stream);
            }
        }
        ++m->epoch;
        rc = run_job(m, kForward, err);
        if (rc != GNNVC_OK) break;
        // lossless or repeated: a pack step raises its flag when a non-zero fits neither the dense columns nor the list
        bool again = false;
        if (m->opt_only_part < 0)
            for (int s = 0; s < 2; ++s) {
                if (!m->pk[s].on) continue;
                bool over = false;
                for (const Part &p : m->parts) over |= p.flag_pin && p.flag_pin[s] != 0;
                if (over) {   // full rows for this stage from now on (the lists were sized from this graph's own counts: it never fit)
                    m->pk[s].on = false;
                    layout_stage(m, s);
                    again = true;
                }
            }
        if (!again) break;
    }
    if (rc != GNNVC_OK) m->pk[0] = m->pk[1] = Packing();   // (a first forward may have failed between choosing and allocating)
    (void)hipSetDevice(m->parts[0].device);
    m->last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int multi_synchronize(MultiState *m) {
    for (Part &p : m->parts) {
        if (hipSetDevice(p.device) != hipSuccess || hipStreamSynchronize(p.stream) != hipSuccess ||
            (p.copy && hipStreamSynchronize(p.copy) != hipSuccess))
            return GNNVC_ERR_DEVICE;
    }
    return GNNVC_OK;
}

}  // namespace gnnvc
