// gnnvc_multi.cpp — several devices behind ONE handle of the C ABI (gnnvc_create_multi, include/gnnvc.h).
//
// SURVEY.md §8b's minimum surface has `gnnvc_create(..., int n_devices)` and §5 "one process driving 8 devices": the
// reference's only call site, m.predict(x, out, g) (src/GNN_VC.cpp:192, include/gnn_inference.hpp:50), is one thread in
// one process, so a drop-in that wants more than one GPU has to partition behind that call.  This file does: the graph
// handed to the front handle is cut into P contiguous row ranges of equal entry count, device r gets
// the CSR slice of its rows (global column ids) and full-size replicated feature buffers, every stage is computed
// row range by row range with the ordinary engines' gnnvc_stage_forward_device, and after the first and second stage
// each device pushes the rows it just computed into every peer's buffer — a direct all-gather over the xGMI mesh (one
// link per peer; no ring, no host hop).  A row is still summed on one device in stored order: the bits are those of the
// single-device engine.  No reference counterpart.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <new>
#include <vector>

#include "../../include/gnnvc.h"
#include "gnnvc_multi.h"

namespace gnnvc {

namespace {

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

struct Part {
    int device = 0;
    gnnvc_engine *eng = nullptr;
    hipStream_t stream = nullptr;
    uint32_t lo = 0, hi = 0;
    uint64_t nnz = 0;
    Buf<uint32_t> rowptr, col, w, nw;
    Buf<float> x, h[2], scores, logits;
    hipEvent_t done[3] = {nullptr, nullptr, nullptr};   // this part's rows of stage s have reached every peer
};

}  // namespace

struct MultiState {
    std::vector<Part> parts;
    uint32_t n = 0;
    uint64_t nnz = 0;
    bool have_graph = false;
    int stages = 0;
    double last_ms = 0.0;
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
    int rc = GNNVC_OK;
    for (int r = 0; r < n_devices && rc == GNNVC_OK; ++r) {
        Part &p = m->parts[(size_t)r];
        p.device = devices[r];
        rc = gnnvc_create(&p.eng, model_text, len, p.device);
        if (rc != GNNVC_OK) {
            err = "creating a per-device engine failed";
            break;
        }
        hipError_t h = hipSetDevice(p.device);
        if (h == hipSuccess) h = hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking);
        for (int s = 0; s < 3 && h == hipSuccess; ++s) h = hipEventCreateWithFlags(&p.done[s], hipEventDisableTiming);
        if (h != hipSuccess) {
            rc = hip_fail(err, h, "stream / event creation");
            break;
        }
        rc = gnnvc_set_stream(p.eng, p.stream);
        // rows travel device to device: let each device write its peers' memory directly where the fabric allows it (without
        // peer access the copies still work, staged by the runtime)
        for (int q = 0; q < r; ++q) {
            const int a = p.device, b = m->parts[(size_t)q].device;
            if (a == b) continue;
            int ok = 0;
            if (hipDeviceCanAccessPeer(&ok, a, b) == hipSuccess && ok) {
                (void)hipSetDevice(a);
                (void)hipDeviceEnablePeerAccess(b, 0);
                (void)hipSetDevice(b);
                (void)hipDeviceEnablePeerAccess(a, 0);
                (void)hipGetLastError();   // ("already enabled" is not an error worth keeping)
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
    *out = m;
    return GNNVC_OK;
}

void multi_destroy(MultiState *m) {
    if (!m) return;
    for (Part &p : m->parts) {
        (void)hipSetDevice(p.device);
        if (p.stream) (void)hipStreamSynchronize(p.stream);
        if (p.eng) gnnvc_destroy(p.eng);
        p.rowptr.release(); p.col.release(); p.w.release(); p.nw.release();
        p.x.release(); p.h[0].release(); p.h[1].release(); p.scores.release(); p.logits.release();
        for (auto &ev : p.done)
            if (ev) (void)hipEventDestroy(ev);
        if (p.stream) (void)hipStreamDestroy(p.stream);
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

int multi_set_weight_scale(MultiState *m, float ws) {
    for (Part &p : m->parts) {
        int rc = gnnvc_set_weight_scale(p.eng, ws);
        if (rc) return rc;
    }
    return GNNVC_OK;
}

int multi_set_option(MultiState *m, const char *key, long value) {
    for (Part &p : m->parts) {
        int rc = gnnvc_set_option(p.eng, key, value);
        if (rc) return rc;
    }
    return GNNVC_OK;
}

int multi_upload(MultiState *m, uint32_t n, const uint64_t *rowptr64, const uint32_t *rowptr32, const uint32_t *col, const uint32_t *w,
                 const uint32_t *nw, std::string &err) {
    m->have_graph = false;
    const size_t P = m->parts.size();
    auto rp = [&](uint32_t u) -> uint64_t { return rowptr64 ? rowptr64[u] : (uint64_t)rowptr32[u]; };
    const uint64_t nnz = n ? rp(n) : 0;
    if (nnz >= 0xFFFFFFFFull - GNNVC_COL_PAD) {
        err = "nnz does not fit 32-bit row pointers";
        return GNNVC_ERR_UNSUPPORTED;
    }
    if (n) {   // (the slices' device-side checks see relative row pointers: the global array has to be checked here)
        if (rp(0) != 0) {
            err = "row pointers are not monotone from 0 to nnz";
            return GNNVC_ERR_INVALID;
        }
        for (uint32_t u = 0; u < n; ++u)
            if (rp(u) > rp(u + 1)) {
                err = "row pointers are not monotone from 0 to nnz";
                return GNNVC_ERR_INVALID;
            }
    }
    // contiguous row ranges of (nearly) equal entry count, cut at multiples of 64 rows (the tile kernels' unit)
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
    std::vector<uint32_t> local;
    for (size_t r = 0; r < P; ++r) {
        Part &p = m->parts[r];
        p.lo = cut[r];
        p.hi = cut[r + 1];
        const uint32_t rows = p.hi - p.lo;
        const uint64_t first = n ? rp(p.lo) : 0;
        p.nnz = n ? rp(p.hi) - first : 0;
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
        local.resize((size_t)rows + 1);
        for (uint32_t i = 0; i <= rows; ++i) local[i] = (uint32_t)(rp(p.lo + i) - first);
        MTRY(hipMemcpyAsync(p.rowptr.p, local.data(), ((size_t)rows + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, p.stream));
        if (p.nnz) MTRY(hipMemcpyAsync(p.col.p, col + first, (size_t)p.nnz * sizeof(uint32_t), hipMemcpyHostToDevice, p.stream));
        MTRY(hipMemsetAsync(p.col.p + p.nnz, 0, GNNVC_COL_PAD * sizeof(uint32_t), p.stream));
        if (rows) {
            MTRY(hipMemcpyAsync(p.w.p, w + p.lo, (size_t)rows * sizeof(uint32_t), hipMemcpyHostToDevice, p.stream));
            MTRY(hipMemcpyAsync(p.nw.p, nw + p.lo, (size_t)rows * sizeof(uint32_t), hipMemcpyHostToDevice, p.stream));
        }
        MTRY(hipStreamSynchronize(p.stream));   // (`local` is reused by the next part)
    }
    if (n == 0) {
        m->n = 0;
        m->nnz = 0;
        m->have_graph = true;
        return GNNVC_OK;
    }
    for (size_t r = 0; r < P; ++r) {
        Part &p = m->parts[r];
        int rc = gnnvc_attach_graph_slice(p.eng, n, p.lo, p.hi, p.nnz, p.rowptr.p, p.col.p, p.w.p, p.nw.p);
        if (rc != GNNVC_OK) return part_fail(err, p, rc, "gnnvc_attach_graph_slice");
    }
    m->n = n;
    m->nnz = nnz;
    m->have_graph = true;
    return GNNVC_OK;
}

int multi_forward_device(MultiState *m, const float *d_x, float *d_scores, float *d_logits, std::string &err) {
    if (!m->have_graph) {
        err = "no graph attached";
        return GNNVC_ERR_STATE;
    }
    const uint32_t n = m->n;
    if (n == 0) return GNNVC_OK;
    const auto t0 = std::chrono::steady_clock::now();
    const size_t P = m->parts.size();
    const int dev0 = m->parts[0].device;
    // the input, replicated; the pad rows of the 16-wide feature buffers read as zero
    for (Part &p : m->parts) {
        MTRY(hipSetDevice(p.device));
        MTRY(hipMemcpyPeerAsync(p.x.p, p.device, d_x, dev0, (size_t)n * sizeof(float), p.stream));
        for (auto &h : p.h) MTRY(hipMemsetAsync(h.p + (size_t)n * 16, 0, 16 * sizeof(float), p.stream));
    }
    for (int s = 0; s < 3; ++s) {
        for (size_t r = 0; r < P; ++r) {
            Part &p = m->parts[r];
            MTRY(hipSetDevice(p.device));
            if (s > 0)   // this stage's input is complete on this device once every peer's rows of the last stage have landed
                for (size_t q = 0; q < P; ++q)
                    if (q != r) MTRY(hipStreamWaitEvent(p.stream, m->parts[q].done[s - 1], 0));
            const float *in = s == 0 ? p.x.p : p.h[(s - 1) & 1].p;
            float *out = s == 2 ? p.scores.p : p.h[s & 1].p;
            if (p.hi > p.lo) {
                // up to 4 devices a part's rows are many enough for the compact table over them to pay (measured per-rank
                // compute, metric graph: P = 4 0.84 / 0.77 ms per 16-wide stage against 1.01 / 0.99 plain; P = 8: 0.67 against
                // 0.53 — the table's passes over all N rows do not shrink with P): announce the stage's complete input
                if (s >= 1 && P <= 4) {
                    int rc = gnnvc_stage_input_ready(p.eng, s, in, p.lo, p.hi);
                    if (rc != GNNVC_OK) return part_fail(err, p, rc, "gnnvc_stage_input_ready");
                }
                int rc = gnnvc_stage_forward_device(p.eng, s, p.lo, p.hi, in, out, s == 2 ? p.logits.p : nullptr);
                if (rc != GNNVC_OK) return part_fail(err, p, rc, "gnnvc_stage_forward_device");
                const size_t rows = p.hi - p.lo;
                if (s < 2) {   // direct all-gather: the rows just computed, straight into every peer's copy
                    for (size_t q = 0; q < P; ++q)
                        if (q != r)
                            MTRY(hipMemcpyPeerAsync(m->parts[q].h[s & 1].p + (size_t)p.lo * 16, m->parts[q].device,
                                                    p.h[s & 1].p + (size_t)p.lo * 16, p.device, rows * 16 * sizeof(float), p.stream));
                } else {       // the scores (and logits) of these rows, to the caller's arrays on the first device
                    MTRY(hipMemcpyPeerAsync(d_scores + p.lo, dev0, p.scores.p + p.lo, p.device, rows * sizeof(float), p.stream));
                    if (d_logits)
                        MTRY(hipMemcpyPeerAsync(d_logits + p.lo, dev0, p.logits.p + p.lo, p.device, rows * sizeof(float), p.stream));
                }
            }
            MTRY(hipEventRecord(p.done[s], p.stream));
        }
    }
    for (Part &p : m->parts) {
        MTRY(hipSetDevice(p.device));
        MTRY(hipStreamSynchronize(p.stream));
    }
    MTRY(hipSetDevice(dev0));
    m->last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return GNNVC_OK;
}

int multi_synchronize(MultiState *m) {
    for (Part &p : m->parts) {
        if (hipSetDevice(p.device) != hipSuccess || hipStreamSynchronize(p.stream) != hipSuccess) return GNNVC_ERR_DEVICE;
    }
    return GNNVC_OK;
}

}  // namespace gnnvc
