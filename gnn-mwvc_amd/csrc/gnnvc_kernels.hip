// gnnvc_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the GNN-VC forward.
//
// What is computed follows the reference bit for bit:
//   graph_layer::forward   reference src/gnn_inference.cpp:27-42
//   linear_layer::forward  reference src/gnn_inference.cpp:20-25 (+ dot(), src/matrix.cpp:106-122)
//   ReLU / sigmoid         reference src/gnn_inference.cpp:44-52
// How it is computed is MI355X-first: one fused kernel per "stage" (graph
// layer + the dense layers up to the next graph layer), wave64 tiles of 64
// vertices, a quad of lanes per gathered 64-byte feature row, the activations
// of a vertex kept in registers across the dense layers, weights fetched
// through the scalar cache, outputs transposed through LDS so every global
// store is a full 64-byte row.
//
// Compile with -ffp-contract=off: the only fused multiply-adds are the explicit
// __builtin_fmaf calls; every other add / divide is separately rounded.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include "exact_sum.h"
#include "expf_glibc.h"
#include "gnnvc_kernels.h"

namespace gnnvc {

namespace {

#ifndef GNNVC_PHASE_PROBE
#define GNNVC_PHASE_PROBE 0
#endif
#if GNNVC_PHASE_PROBE
// (experiment builds only: scratch/experiments/phase_probe.py) a wave's first lane stamps the 100 MHz wall clock at the phases of a tile
__device__ unsigned long long *gnnvc_probe_buf = nullptr;
__device__ int gnnvc_probe_kind = 0;
__device__ __forceinline__ void probe_mark(int kind, int phase) {
    if (gnnvc_probe_buf && gnnvc_probe_kind == kind && (threadIdx.x & 63) == 0)
        gnnvc_probe_buf[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + phase] = wall_clock64();
}
#define GNNVC_PROBE(kind, phase) probe_mark(kind, phase)
#else
#define GNNVC_PROBE(kind, phase) ((void)0)
#endif

constexpr int kWave = 64;
constexpr int kBlock = 256;              // 4 waves, each wave owns one 64-vertex tile
constexpr int kWavesPerBlock = kBlock / kWave;

// ---- per-kernel timing (bench.py's roofline block): while a sink is installed on this thread, every launch
// on the sink's stream is bracketed by two HIP events of the sink's pool (launches on the side streams are not).
thread_local KernelTraceSink *t_sink = nullptr;

// GNNVC_DEBUG_SYNC=1 in the environment: every launch is announced on stderr and waited for (which kernel a device fault
// belongs to; nothing overlaps any more)
static const bool g_debug_sync = getenv("GNNVC_DEBUG_SYNC") != nullptr;

struct KernelTimer {
    KernelTraceSink::Rec *rec = nullptr;
    hipStream_t stream;
    const char *dbg_name = nullptr;
    KernelTimer(hipStream_t s, const char *name) : stream(s) {
        if (g_debug_sync) {
            dbg_name = name;
            fprintf(stderr, "[gnnvc] launch %s\n", name);
            fflush(stderr);
        }
        KernelTraceSink *k = t_sink;
        if (!k || k->stream != s) return;
        if (k->used == k->recs.size()) {
            KernelTraceSink::Rec fresh{name, nullptr, nullptr};
            if (hipEventCreate(&fresh.a) != hipSuccess || hipEventCreate(&fresh.b) != hipSuccess) return;
            k->recs.push_back(fresh);
        }
        rec = &k->recs[k->used++];
        rec->name = name;
        (void)hipEventRecord(rec->a, s);
    }
    ~KernelTimer() {
        if (rec) (void)hipEventRecord(rec->b, stream);
        if (dbg_name) {
            const hipError_t rc = hipStreamSynchronize(stream);
            fprintf(stderr, "[gnnvc] done   %s: %s\n", dbg_name, hipGetErrorName(rc));
            fflush(stderr);
        }
    }
};
#define GNNVC_LAUNCH(kernel_, grid_, block_, lds_, stream_, ...)              \
    do {                                                                      \
        KernelTimer kt_((stream_), #kernel_);                                 \
        hipLaunchKernelGGL(kernel_, grid_, block_, lds_, stream_, __VA_ARGS__); \
    } while (0)

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in program order; this only stops the
    // compiler from moving LDS accesses across the hand-off point.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// std::max(x, 0.0f) of the reference (src/gnn_inference.cpp:46): (x < 0) ? 0 : x.
__device__ __forceinline__ float relu_ref(float x) { return (x < 0.0f) ? 0.0f : x; }

// 1.0f / (1.0f + expf(-x)) (src/gnn_inference.cpp:51) with glibc's expf restated
// in fp64 (expf_glibc.h): same bits as the host libm the reference links.
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf_glibc(-x)); }

// out[j] = act( fma-chain_k( in[k] * W[k][j] ) + b[j] ): per output one
// sequential-k chain from +0.0f (what cblas_sgemm computes at these sizes),
// then a separately rounded bias add.  W and b are wave-uniform addresses, so
// they are fetched with scalar loads and feed v_fma as SGPR operands.
template <int K, int KUSED, int N, int ACT>
__device__ __forceinline__ void dense(const float (&in)[K], float (&out)[N],
                                      const float *__restrict__ W, const float *__restrict__ b) {
#pragma unroll
    for (int j = 0; j < N; ++j) out[j] = 0.0f;
#pragma unroll
    for (int k = 0; k < KUSED; ++k) {
        const float a = in[k];
#pragma unroll
        for (int j = 0; j < N; ++j) out[j] = __builtin_fmaf(a, W[k * N + j], out[j]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float t = out[j] + b[j];
        out[j] = (ACT == 0) ? relu_ref(t) : t;
    }
}

// XCD-aware tile mapping.  Workgroups are dealt round-robin over the 8 XCDs
// (block b shares an L2 with block b + 8), so XCD-group x = b % 8 walks the
// x-th contiguous eighth of the tile range: rows a graph stores close together
// are gathered through the same L2.
__device__ __forceinline__ uint32_t tile_for_wave(uint32_t ntiles, bool interleave = false) {
    // interleave: tiles dealt round-robin over the blocks (hence over the XCDs) — used when
    // contiguous eighths of the tile range carry very different amounts of work
    if (interleave) return blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint32_t per_xcd = (ntiles + 7u) / 8u;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t slot = (blockIdx.x >> 3) * kWavesPerBlock + (threadIdx.x >> 6);
    return (slot < per_xcd) ? xcd * per_xcd + slot : 0xFFFFFFFFu;
}

// ------------------------------------------------------------------ per-wave LDS region
// One region per wave, reused over the life of a tile:
//   1. column-index stage: the tile's slice of `col` (all entries of its 64 rows),
//      fetched with full-line coalesced loads so every 128-byte line of `col` is
//      requested from L2 exactly once per tile;
//   2. dense-layer input tile: 64 rows x 32 floats in the first layer's k order
//      [aggregate 0..15 | h0 | degree | W/ws | NW/ws | h4..h15], pitch 33;
//   3. output tile: 64 rows x 17 floats, read back row-wise for full-row stores.
#ifndef GNNVC_SORTED_MIX
#define GNNVC_SORTED_MIX 1   // degree-sorted tiles are dispatched alternating between the two ends of the list (A/B: 0)
#endif
constexpr int kRegionFloats = 2112;          // 8448 B per wave, 33 KiB per workgroup
constexpr int kInPitch = 33;                 // 32 inputs in k order; odd pitch: conflict-free ds_read_b32 down a column
constexpr int kOutPitch = 17;
constexpr uint32_t kStageCap = 1792 - 4;     // most col entries a tile may stage (7 x 256 minus alignment slack)
static_assert(kWave * kInPitch <= kRegionFloats && 1792 <= kRegionFloats, "region too small");

// Stage col[c0, c1) of this wave's tile into LDS.  Returns the entry index that
// LDS slot 0 corresponds to (c0 rounded down to a 16-byte boundary).
__device__ __forceinline__ uint32_t stage_cols(const uint32_t *__restrict__ col, uint32_t c0,
                                               uint32_t c1, uint32_t *stage, int lane) {
    const uint32_t base = c0 & ~3u;
    for (uint32_t off = base + 4u * lane; off < c1; off += 256u) {
        const uint4 v = *reinterpret_cast<const uint4 *>(col + off);  // col is padded past nnz
        *reinterpret_cast<uint4 *>(stage + (off - base)) = v;
    }
    return base;
}

// ------------------------------------------------------------------ dense layers on the matrix cores
// fp32 MFMA (v_mfma_f32_32x32x2_f32) accumulates k-ordered, one rounding per product:
// D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) with k0 supplied by lanes 0-31 and k1 by lanes
// 32-63 — bit for bit the sequential fmaf chain the reference's SGEMM computes, so the
// dense layers can run on the matrix pipe without leaving the exact-parity contract.
//
// Orientation: H^T = W^T * X^T.  A = W^T (out-feature j on the lane, k on the lane half),
// B = X^T (vertex on the lane), so a 64-vertex tile is two 32-wide B tiles (vt = 0, 1)
// sharing every A operand.  The 32x32 result holds, on lane (v = l & 31, h = l >> 5),
// features j = (r & 3) + 8 (r >> 2) + 4 h of vertex v in registers r = 0..15 — and that is
// (after one v_permlane32_swap per register pair) exactly the B operand of the NEXT layer in
// ascending-k order, so activations never leave the registers between layers.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int mfma_feat(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// A operands of a layer with weights W[K x N] (row-major): step s, lane (v, h) -> W[2s + h][v]
template <int N, int STEPS>
__device__ __forceinline__ void mfma_load_a(const float *__restrict__ W, int v, int h, float (&a)[16]) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s) a[s] = (v < N) ? W[(2 * s + h) * N + v] : 0.0f;
}

// first layer of a chain: B operand straight from the LDS input tile (row = vertex, col = k)
__device__ __forceinline__ f32x16 mfma_layer_lds(const float *T, int vt, int v, int h, const float (&a)[16]) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float *row = &T[(32 * vt + v) * kInPitch + h];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], row[2 * s], acc, 0, 0, 0);
    return acc;
}

// separately rounded bias add + activation on a result tile (features >= N are padding)
template <int N, int ACT>
__device__ __forceinline__ void mfma_bias_act(f32x16 &d, const float *__restrict__ bias, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if ((r >> 2) * 8 < N) {   // static per r: feature index < N for both halves
            const float t = d[r] + bias[mfma_feat(r, h)];
            d[r] = (ACT == 0) ? relu_ref(t) : t;
        }
    }
}

// next layer: the previous result tile (K = 32 or 16 features) becomes the B operand.
// Registers 4g..4g+3 hold features 8g + {0..3} on the low half and 8g + {4..7} on the high
// half; swapping the halves of (4g, 4g+1) and (4g+2, 4g+3) yields the pairs (8g, 8g+1),
// (8g+4, 8g+5) and (8g+2, 8g+3), (8g+6, 8g+7): steps 4g, 4g+2, 4g+1, 4g+3.
template <int K>
__device__ __forceinline__ f32x16 mfma_layer_acc(const f32x16 &d, const float (&a)[16]) {
    float b[16];
#pragma unroll
    for (int g = 0; g < K / 8; ++g) {
        auto p0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[4 * g + 0]), __float_as_uint(d[4 * g + 1]), false, false);
        auto p1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[4 * g + 2]), __float_as_uint(d[4 * g + 3]), false, false);
        b[4 * g + 0] = __uint_as_float(p0[0]);
        b[4 * g + 2] = __uint_as_float(p0[1]);
        b[4 * g + 1] = __uint_as_float(p1[0]);
        b[4 * g + 3] = __uint_as_float(p1[1]);
    }
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < K / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    return acc;
}

// 16 -> 1 layer on a 16-feature result tile: one k-ordered fma chain per vertex that hops
// between the two lane halves (features 0-3 low, 4-7 high, 8-11 low, 12-15 high).
// Returns the pre-bias chain value, valid on the HIGH half lane of each vertex.
__device__ __forceinline__ float mfma_tail_16to1(const f32x16 &d, const float *__restrict__ W, int h) {
    float w[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) w[r] = W[mfma_feat(r, h)];
    float acc = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_fmaf(d[r], w[r], acc);          // k 0..3   (low half)
    float t = __shfl_xor(acc, 32);
    acc = h ? t : acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_fmaf(d[r], w[r], acc);          // k 4..7   (high half)
    t = __shfl_xor(acc, 32);
    acc = h ? acc : t;
#pragma unroll
    for (int r = 4; r < 8; ++r) acc = __builtin_fmaf(d[r], w[r], acc);          // k 8..11  (low half)
    t = __shfl_xor(acc, 32);
    acc = h ? t : acc;
#pragma unroll
    for (int r = 4; r < 8; ++r) acc = __builtin_fmaf(d[r], w[r], acc);          // k 12..15 (high half)
    return acc;
}

// Layers 2 and 3 of a stage on the matrix cores, given layer-1 result tiles d[vt] (already
// bias+ReLU'ed), ending either in the 16-float output tile in LDS (row = vertex, pitch 17)
// or in the per-vertex logit.
template <int N1, int N2, int N3, bool SIGMOID>
__device__ __forceinline__ void mfma_tail(f32x16 (&d)[2], const float *__restrict__ W2, const float *__restrict__ b2,
                                          const float *__restrict__ W3, const float *__restrict__ b3, float *T,
                                          int v, int h, float (&logit)[2]) {
    static_assert(N1 == 32, "layer-2 input is a full 32-feature tile");
    float a[16];
    mfma_load_a<N2, 16>(W2, v, h, a);
#pragma unroll
    for (int vt = 0; vt < 2; ++vt) {
        d[vt] = mfma_layer_acc<32>(d[vt], a);
        mfma_bias_act<N2, 0>(d[vt], b2, h);
    }
    if constexpr (SIGMOID) {
        static_assert(N2 == 16 && N3 == 1, "sigmoid tail is 16 -> 1");
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) logit[vt] = mfma_tail_16to1(d[vt], W3, h) + b3[0];
    } else {
        static_assert(N3 == 16, "feature stages emit 16 floats per vertex");
        mfma_load_a<N3, N2 / 2>(W3, v, h, a);
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            d[vt] = mfma_layer_acc<N2>(d[vt], a);
            mfma_bias_act<N3, 0>(d[vt], b3, h);
        }
        wave_lds_sync();   // input tile fully consumed: reuse the region as the output tile
#pragma unroll
        for (int vt = 0; vt < 2; ++vt)
#pragma unroll
            for (int r = 0; r < 8; ++r) T[(32 * vt + v) * kOutPitch + mfma_feat(r, h)] = d[vt][r];
    }
}

// ------------------------------------------------------------------ stage, F = 16
// Parameters: W1[35 x N1] b1 W2[N1 x N2] b2 W3[N2 x N3] b3.  Input columns of
// the first dense layer after the reference's column layout (f = 16):
//   0..15 aggregate, 16 = h[0], 17 = degree, 18 = W/ws, 19 = NW/ws,
//   20..31 = h[4..15], 32..34 = +0.0 (never written; their k-terms are exact
//   no-ops in the fma chain and are skipped).
// SORTED: instead of 64 consecutive vertices a tile takes 64 consecutive entries of a
// degree-sorted vertex list (srt_vertex, with {row begin, row end, W, NW} per entry in
// srt_meta).  A tile costs max-degree gather rounds whatever the other lanes do, so on
// skewed graphs tiles of similar degree waste far fewer rounds; rows are then no longer
// contiguous, so the column indices are read straight from global memory.
// Producer side of the compact-table plan (k_c4_*, further down): a stage kernel's VALU epilogue hands over
// this lane's finished output row (already written to its LDS tile row `trow`, non-zero mask `nz`).  The row
// is added to the per-column non-zero counts of the NEXT stage's input — 64 slots of 17 counters (16 columns +
// rows seen) so that the atomics of different blocks rarely meet — and, when the previous forward's choice of
// table columns is at hand (spec[0] != 0), its compact form is written: the consumer then needs neither its
// counting pass nor its compaction pass over the 640 MB it is about to read.
typedef float c4row __attribute__((ext_vector_type(4)));
constexpr int kEmitSlots = 64, kEmitStride = 17;

// (the spec words as values: a kernel that loads them when it starts has them in scalar registers by the time its epilogue runs —
// read there, behind the dense layers, the two dependent scalar loads are a microsecond or two on every wave's critical path)
__device__ __forceinline__ void c4_emit_with(const float *trow, uint32_t nz, uint32_t u, bool mine, int lane, uint32_t spec_ok,
                                             uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3, c4row *__restrict__ table,
                                             unsigned long long *__restrict__ counts) {
    uint32_t my = 0;
#pragma unroll
    for (int cidx = 0; cidx < 16; ++cidx) {
        const uint32_t cnt = (uint32_t)__popcll(__ballot(mine && (nz >> cidx & 1u)));
        if (lane == cidx) my = cnt;
    }
    const uint32_t rows = (uint32_t)__popcll(__ballot(mine));
    if (lane == 16) my = rows;
    if (lane <= 16 && my) atomicAdd(&counts[(blockIdx.x & (kEmitSlots - 1)) * kEmitStride + lane], (unsigned long long)my);
    if (spec_ok == 1u && mine) {   // (a plan of several passes writes its tables in k_c4_compact)
        const float a = trow[s0], b = trow[s1], c = trow[s2], d = trow[s3];
        c4row out = {a == 0.0f ? 0.0f : a, b == 0.0f ? 0.0f : b, c == 0.0f ? 0.0f : c, d == 0.0f ? 0.0f : d};
        // (a value that is not >= 0 — a NaN, or a negative one if the producer ever ends in something else than a ReLU — cannot lend
        // its sign bit: the vertex is flagged in BOTH of the first two values, and a consumer that wants a flagged vertex's table
        // values — k_stage_t4 — takes such a row's neighbours from the full rows)
        const bool odd = !(a >= 0.0f) || !(b >= 0.0f) || !(c >= 0.0f) || !(d >= 0.0f);
        if (odd || (nz & ~((1u << s0) | (1u << s1) | (1u << s2) | (1u << s3))))
            out[0] = __uint_as_float(__float_as_uint(out[0]) | 0x80000000u);   // stray non-zeros: flag the vertex
        if (odd) out[1] = __uint_as_float(__float_as_uint(out[1]) | 0x80000000u);
        table[u] = out;
    }
}
__device__ __forceinline__ void c4_emit(const float *trow, uint32_t nz, uint32_t u, bool mine, int lane,
                                        const uint32_t *__restrict__ spec, c4row *__restrict__ table,
                                        unsigned long long *__restrict__ counts) {
    const uint32_t ok = spec[0];
    c4_emit_with(trow, nz, u, mine, lane, ok, ok == 1u ? spec[1] : 0u, ok == 1u ? spec[2] : 0u, ok == 1u ? spec[3] : 0u,
                 ok == 1u ? spec[4] : 0u, table, counts);
}

// AGGONLY (compact-table plan): every row's aggregate arrives ready-made — four sums in acc4 for clean rows, the
// full 16 in agg16 for dirty ones — and the kernel is only the dense layers; it leaves at once when the device
// found the input unfit for the plan (c4desc[0] == 0), and the gathering variant leaves at once when it was fit.
// FILTER (round 3, a skewed graph's stages while they have no pruned adjacency yet — a graph's first forward, or an input its
// plan does not fit): g.zero_bits says which vertices' rows are all zero in THIS call's input (k_prune_mark_zero of this very
// input, a moment ago), and an entry that points to one of them gathers the pad row instead — an L1 hit, not a request to the
// fabric, which on R-MAT-22 is what 73 - 86 % of the entries would otherwise be.  x + 0 == x: the same sums, bit for bit, with
// nothing to build and nothing to prove (the set is this input's own).
// is the set worth a look-up per entry?  (uniform: zero_min_pct of the entries point into it, if the adjacency is symmetric — the
// degrees of the set's vertices AMONG THE ROWS THIS ENGINE HOLDS against the entries it holds: a slice sees its own share of both)
__device__ __forceinline__ bool filter_worth_info(const GraphDev &g, const unsigned long long *info) {
    return info[0] * 100ull >= g.nnz * (unsigned long long)g.zero_min_pct;
}
__device__ __forceinline__ bool filter_worth(const GraphDev &g) { return g.zero_bits != nullptr && filter_worth_info(g, g.zero_info); }

template <int N1, int N2, int N3, bool SIGMOID, int S, bool MFMA, bool SORTED, bool AGGONLY = false, bool FILTER = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3))) void k_stage_f16(
        GraphDev g, float ws, const float4 *__restrict__ fin, float *__restrict__ fout,
        float *__restrict__ logits, const float *__restrict__ P, uint32_t row_lo,
        uint32_t row_hi, uint32_t long_thresh, const uint32_t *__restrict__ srt_vertex,
        const uint4 *__restrict__ srt_meta, uint32_t n_sorted, int interleave,
        const float4 *__restrict__ acc4, const uint32_t *__restrict__ c4desc, const float4 *__restrict__ agg16,
        const uint32_t *__restrict__ emit_spec, c4row *__restrict__ emit_table, unsigned long long *__restrict__ emit_counts,
        const uint32_t *__restrict__ srt_vertex_p = nullptr, const uint4 *__restrict__ srt_meta_p = nullptr, uint32_t n_sorted_p = 0) {
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][kRegionFloats];
    const int lane = threadIdx.x & 63;
    GNNVC_PROBE((SIGMOID ? 2 : 1), 0);
    float *T = lds[threadIdx.x >> 6];
    uint32_t *stage = reinterpret_cast<uint32_t *>(T);
    if constexpr (AGGONLY) {
        if (c4desc[0] == 0) return;           // uniform: the plan does not apply to this input
    } else {
        if (c4desc && c4desc[0] != 0) return; // uniform: the aggregate-only variant has this launch
    }
    // the entries to gather: the whole adjacency, or (uniform; decided on the device for this input) the pruned one
    const bool pruned = !AGGONLY && g.prune_bad != nullptr && *g.prune_bad == 0u;
    const uint32_t *__restrict__ grp = pruned ? g.prp : g.rowptr;
    const uint32_t *__restrict__ gcol = pruned ? g.pcol : g.col;
    // with the pruned adjacency a row's class (tile kernel / long-row kernel) goes by the entries it has left, and the
    // sorted tiles come from the list built from those numbers (its meta holds the pruned ranges)
    const bool by_left = pruned && g.prune_eff != 0u;
    const bool filt = FILTER && !pruned && filter_worth(g);
    if (SORTED && by_left) {
        srt_vertex = srt_vertex_p;
        srt_meta = srt_meta_p;
        n_sorted = n_sorted_p;
    }
    const uint32_t ntiles = SORTED ? (n_sorted + kWave - 1) / kWave : (row_hi - row_lo + kWave - 1) / kWave;
    // natural order: XCD-contiguous ranges (locality).  Sorted order lists the heaviest tiles
    // first, so they are dealt round-robin instead — consecutive blocks sit on different XCDs
    // and every XCD gets the same mix of heavy and light tiles.
    // (the grid is sized for the longer of the two lists: the same launch serves whichever the device picks)
    const uint32_t grid_tiles = SORTED ? (max(n_sorted, n_sorted_p) + kWave - 1) / kWave : ntiles;
    uint32_t tile = tile_for_wave(grid_tiles, SORTED || interleave);
    if (tile >= ntiles) return;
    if (SORTED && GNNVC_SORTED_MIX) {
        // The list runs from the heaviest rows to rows without entries; dispatched in that order the kernel would first only
        // gather (every wave waiting on the fabric) and at the end only run dense layers (the fabric idle).  Alternating
        // between the two ends gives every workgroup tiles of both kinds at any time; the heavy tiles still start first.
        tile = (tile & 1u) ? ntiles - 1u - (tile >> 1) : (tile >> 1);
    }
    const uint32_t v0 = row_lo + tile * kWave;   // natural order only

    // lane-per-vertex view of the tile
    uint32_t u, rs, re;
    bool mine;
    float f_deg, f_w, f_nw;
    if constexpr (SORTED) {
        const uint32_t slot = tile * kWave + lane;
        mine = slot < n_sorted;                   // long rows and other ranks' rows are not in the list
        const uint32_t sl = mine ? slot : n_sorted - 1;
        u = srt_vertex[sl];
        const uint4 meta = srt_meta[sl];
        if (by_left) {   // the list's meta holds the pruned range; the degree (a feature) comes from the adjacency itself
            rs = meta.x;
            re = mine ? meta.y : rs;
            f_deg = (float)(g.rowptr[u + 1] - g.rowptr[u]);
        } else {
            rs = pruned ? grp[u] : meta.x;
            re = mine ? (pruned ? grp[u + 1] : meta.y) : rs;
            f_deg = (float)(meta.y - meta.x);
        }
        f_w = (float)meta.z / ws;
        f_nw = (float)meta.w / ws;
    } else {
        const uint32_t uu = v0 + lane;
        const bool valid = uu < row_hi;
        u = valid ? uu : row_hi - 1;
        const uint32_t rs_full = g.rowptr[u];
        const uint32_t re_full = valid ? g.rowptr[u + 1] : rs_full;
        // rows of degree >= long_thresh belong to the long-row kernel (k_long_f16): no gather,
        // no store for them here
        rs = pruned ? grp[u] : rs_full;
        const uint32_t re_p = pruned ? (valid ? grp[u + 1] : rs) : re_full;
        mine = valid && (by_left ? (re_p - rs) < g.eff_thresh && (re_full - rs_full) < g.eff_giant : (re_full - rs_full) < long_thresh);
        re = mine ? re_p : rs;
        f_deg = (float)(re_full - rs_full);
        f_w = (float)g.w[u] / ws;
        f_nw = (float)g.nw[u] / ws;
    }

    // natural order: the tile's slice of col, [c0, c1), is contiguous in CSR -> stage it in LDS
    bool staged = false;
    uint32_t sbase = 0;
    if constexpr (!SORTED && !AGGONLY) {
        const uint32_t c0 = __builtin_amdgcn_readfirstlane(rs);
        const uint32_t vend = (v0 + kWave < row_hi) ? v0 + kWave : row_hi;
        const uint32_t c1 = grp[vend];        // wave-uniform: end of the tile's last valid row
        staged = (c1 - c0) <= kStageCap;
        if (staged) sbase = stage_cols(gcol, c0, c1, stage, lane);
    }

    // ---- quad layout: quad q of lanes owns the tile's vertices 16p + q (p = 0..3),
    // lane c of the quad holds floats 4c..4c+3 of a 64-byte feature row.
    const int q = lane >> 2, c = lane & 3;
    uint32_t b[4], e[4], urow[4];
    int okrow[4];
    float4 self[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        b[p] = __shfl(rs, 16 * p + q);
        e[p] = __shfl(re, 16 * p + q);
        urow[p] = __shfl(u, 16 * p + q);          // always a valid vertex id (clamped above)
        okrow[p] = __shfl((int)mine, 16 * p + q);
        self[p] = fin[(size_t)urow[p] * 4 + c];   // own rows: 16 full rows per load
    }
    float4 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t zrow = g.n;  // all-zero pad row: x + 0.0f == x exactly
    if constexpr (AGGONLY) {
        // this lane holds feature columns 4c .. 4c+3: which of them are table columns, of which pass, and which slot
        const uint32_t npass = c4desc[0];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 a0[4] = {acc4[urow[0]], acc4[urow[1]], acc4[urow[2]], acc4[urow[3]]};
        for (uint32_t qp = 0; qp < npass; ++qp) {   // (uniform; 1 on degree-uniform graphs)
            const uint32_t *dc = c4desc + (qp == 0 ? 1 : 4 + 4 * qp);   // columns of pass 0 at [1..4], pass 1 at [8..11], pass 2 at [12..15]
            const uint32_t d0 = dc[0], d1 = dc[1], d2 = dc[2], d3 = dc[3];
            int sel[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t col = 4 * c + t;
                sel[t] = col == d0 ? 0 : col == d1 ? 1 : col == d2 ? 2 : col == d3 ? 3 : -1;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float4 a = qp == 0 ? a0[p] : acc4[(size_t)qp * g.n + urow[p]];
                auto pick = [&](int j, float old) { return j == 0 ? a.x : j == 1 ? a.y : j == 2 ? a.z : j == 3 ? a.w : old; };
                acc[p] = make_float4(pick(sel[0], acc[p].x), pick(sel[1], acc[p].y), pick(sel[2], acc[p].z), pick(sel[3], acc[p].w));
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if ((__float_as_uint(a0[p].x) >> 31) != 0)   // dirty: met a neighbour with stray non-zeros; recomputed from full rows by k_c4_fix
                acc[p] = agg16[(size_t)__float_as_uint(a0[p].y) * 4 + c];
    }
    wave_lds_sync();            // staged indices visible to the whole wave
    GNNVC_PROBE((SIGMOID ? 2 : 1), 2);

    // ---- gather: neighbour rows summed in CSR order, S rows per vertex in flight
    if constexpr (!AGGONLY) {
    while ((b[0] < e[0]) | (b[1] < e[1]) | (b[2] < e[2]) | (b[3] < e[3])) {
        uint32_t idx[4][S];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const uint32_t ee = b[p] + s;
                // staged: ee - sbase < 1792 + S always (ee <= c1 + S - 1); slots past c1 hold junk, masked below
                const uint32_t cv = staged ? stage[ee - sbase] : gcol[ee];
                idx[p][s] = (ee < e[p]) ? cv : zrow;
            }
        if (FILTER && filt) {
            const uint32_t *__restrict__ zb = g.zero_bits;   // (n + 1 bits: the pad row's is 0)
            uint32_t wd[4][S];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int s = 0; s < S; ++s) wd[p][s] = zb[idx[p][s] >> 5];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int s = 0; s < S; ++s) idx[p][s] = (wd[p][s] >> (idx[p][s] & 31u) & 1u) ? zrow : idx[p][s];
        }
        float4 r[4][S];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int s = 0; s < S; ++s) r[p][s] = fin[(size_t)idx[p][s] * 4 + c];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                acc[p].x += r[p][s].x;
                acc[p].y += r[p][s].y;
                acc[p].z += r[p][s].z;
                acc[p].w += r[p][s].w;
            }
            const uint32_t nb = b[p] + S;
            b[p] = nb < e[p] ? nb : e[p];
        }
    }
    }

    GNNVC_PROBE((SIGMOID ? 2 : 1), 3);
    // ---- hand over through the LDS tile: 64 rows x 32 inputs in k order
    wave_lds_sync();  // every lane is done with the index stage before it is overwritten
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float *row = &T[(16 * p + q) * kInPitch];
        row[4 * c + 0] = acc[p].x; row[4 * c + 1] = acc[p].y;
        row[4 * c + 2] = acc[p].z; row[4 * c + 3] = acc[p].w;
        if (c == 0) {
            row[16] = self[p].x;                 // h[0]; h[1..3] are overwritten by degree / weights
        } else {
            float *d = &row[20 + 4 * (c - 1)];   // h[4..15]
            d[0] = self[p].x; d[1] = self[p].y; d[2] = self[p].z; d[3] = self[p].w;
        }
    }
    {   // columns 17..19 come from the lane-per-vertex view (lane L <-> vertex v0 + L)
        float *row = &T[lane * kInPitch];
        row[17] = f_deg; row[18] = f_w; row[19] = f_nw;
    }
    wave_lds_sync();

    const float *W1 = P, *b1 = W1 + 35 * N1;
    const float *W2 = b1 + N1, *b2 = W2 + N1 * N2;
    const float *W3 = b2 + N2, *b3 = W3 + N2 * N3;

    if constexpr (MFMA) {
        // ---- dense layers on the matrix cores (see mfma_* above)
        const int v = lane & 31, h = lane >> 5;
        f32x16 d[2];
        {
            float a[16];
            mfma_load_a<N1, 16>(W1, v, h, a);   // rows 0..31 of W1; rows 32..34 meet exact zeros
#pragma unroll
            for (int vt = 0; vt < 2; ++vt) {
                d[vt] = mfma_layer_lds(T, vt, v, h, a);
                mfma_bias_act<N1, 0>(d[vt], b1, h);
            }
        }
        float logit[2] = {0.f, 0.f};
        mfma_tail<N1, N2, N3, SIGMOID>(d, W2, b2, W3, b3, T, v, h, logit);
        if constexpr (SIGMOID) {
            // the chain value of vertex 32 vt + v sits on the high half lane
#pragma unroll
            for (int vt = 0; vt < 2; ++vt) {
                const uint32_t uv = __shfl(u, 32 * vt + v);
                const int keep = __shfl((int)mine, 32 * vt + v);   // all lanes active: a masked-off source lane reads as 0
                if (h == 1 && keep) {
                    if (logits) logits[uv] = logit[vt];
                    fout[uv] = sigmoid_ref(logit[vt]);
                }
            }
        }
    } else {
        // ---- dense layers on the VALU: one lane per vertex, activations in registers
        float x0[32];
        {
            const float *row = &T[lane * kInPitch];
#pragma unroll
            for (int j = 0; j < 32; ++j) x0[j] = row[j];
        }
        float x1[N1], x2[N2], x3[N3];
        dense<32, 32, N1, 0>(x0, x1, W1, b1);
        dense<N1, N1, N2, 0>(x1, x2, W2, b2);
        dense<N2, N2, N3, SIGMOID ? 1 : 0>(x2, x3, W3, b3);
        if constexpr (SIGMOID) {
            static_assert(N3 == 1, "sigmoid stage ends in one output");
            if (mine) {
                if (logits) logits[u] = x3[0];
                fout[u] = sigmoid_ref(x3[0]);
            }
        } else {
            wave_lds_sync();
            uint32_t nz = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                T[lane * kOutPitch + j] = x3[j];
                nz |= (x3[j] != 0.0f ? 1u : 0u) << j;
            }
            if (emit_counts) c4_emit(&T[lane * kOutPitch], nz, u, mine, lane, emit_spec, emit_table, emit_counts);
        }
    }

    GNNVC_PROBE((SIGMOID ? 2 : 1), 6);
    if constexpr (!SIGMOID) {
        static_assert(N3 == 16, "feature stages emit 16 floats per vertex");
        // the 64 x 16 output tile sits in LDS: every global store instruction writes 16 full rows
        wave_lds_sync();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float *src = &T[(16 * p + q) * kOutPitch + 4 * c];
            const float4 o = make_float4(src[0], src[1], src[2], src[3]);
            if (okrow[p]) reinterpret_cast<float4 *>(fout)[(size_t)urow[p] * 4 + c] = o;
        }
    }
    GNNVC_PROBE((SIGMOID ? 2 : 1), 7);
}

// ------------------------------------------------------------------ dense-only sigmoid stage (no LDS)
// When the last stage's aggregates arrive ready-made (compact-table plan: four sums per clean row, sixteen per
// dirty one) what is left of it is the dense layers and the sigmoid.  This kernel does them one lane per vertex on
// the VALU with NO LDS, so that its workgroups fit on a CU beside k_c4_agg's (which hold nearly all of its LDS
// but half of its wave slots and registers and leave most VALU cycles idle): the engine launches the sums one
// round of the persistent grid at a time and runs this kernel for round k on a second stream, under the sums
// of round k + 1.  Same fma chains, same bits.  (The feature stages keep the LDS-staged kernel: their 64-byte
// row stores would be one lane per row here, and that many partial-line requests slow the co-running sums by
// more than the overlap gains — measured.)
// acc4[u] = four sums (columns c4desc[1..4]) of a clean row, or {sign bit, slot} of a dirty one whose sixteen sums
// sit in agg16[slot]; leaves at once when c4desc[0] == 0 (the gathering k_stage_f16 has the launch then).
// (round 4) The same kernel serves the FEATURE stage (SIGMOID false: 16 outputs per row, handed to full-row stores through a
// per-wave LDS tile, and the producer side of the next stage's table, c4_emit) — and both skip what they know to be zero: a clean
// row's 32 first-layer inputs are its four sums, its own values in the table's four columns (the input's compact table holds
// exactly those: 16 bytes per vertex instead of its 64-byte row), degree and the two weights — at most 11 non-zero terms of the
// 32-term chain, and fma(+-0, w, acc) == acc leaves every other term out bit for bit (what already drops columns 32 - 34).
// Three routes, uniform per wave: (A) no dirty row, no vertex with stray non-zeros: <= 11 terms; (B) dirty rows (their sixteen
// sums from agg16) but no stray vertex: 16 + <= 7 terms; (C) anything else (a stray vertex among the wave's own rows, several
// passes): the full chain from full rows.  Terms run in ascending k in every route: the reference's order.
template <int N>
__device__ __forceinline__ void fma_term(float (&out)[N], float a, const float *__restrict__ Wrow) {
#pragma unroll
    for (int j = 0; j < N; ++j) out[j] = __builtin_fmaf(a, Wrow[j], out[j]);
}
template <int N>
__device__ __forceinline__ void bias_relu(float (&out)[N], const float *__restrict__ b) {
#pragma unroll
    for (int j = 0; j < N; ++j) out[j] = relu_ref(out[j] + b[j]);
}

template <int N1, int N2, int N3, bool SIGMOID>
__global__ __launch_bounds__(kBlock) void k_dense_f16(GraphDev g, float ws, const float4 *__restrict__ fin, float *__restrict__ fout,
                                                      float *__restrict__ logits, const float *__restrict__ P, uint32_t row_lo,
                                                      uint32_t row_hi, const float4 *__restrict__ acc4,
                                                      const uint32_t *__restrict__ c4desc, const float4 *__restrict__ agg16,
                                                      const float4 *table_in /* may alias emit_table: each lane reads its row first */, uint32_t long_thresh,
                                                      const uint32_t *__restrict__ emit_spec, c4row *__restrict__ emit_table,
                                                      unsigned long long *__restrict__ emit_counts) {
    static_assert(SIGMOID ? N3 == 1 : N3 == 16, "a feature stage writes 16 floats per row, the last stage one score");
    __shared__ float lds[SIGMOID ? 1 : kWavesPerBlock][SIGMOID ? 1 : kWave * kOutPitch];
    if (c4desc[0] == 0) return;
    const int lane = threadIdx.x & 63;
    const uint32_t uu = row_lo + blockIdx.x * kBlock + threadIdx.x;
    const uint32_t u = uu < row_hi ? uu : row_hi - 1;
    const uint32_t deg = g.rowptr[u + 1] - g.rowptr[u];
    const bool mine = uu < row_hi && deg < long_thresh;   // (longer rows: the long-row kernels')
    const uint32_t npass = c4desc[0];
    const uint32_t d0 = c4desc[1], d1 = c4desc[2], d2 = c4desc[3], d3 = c4desc[4];
    const float4 a = acc4[u];
    const bool dirty = (__float_as_uint(a.x) >> 31) != 0;   // met a neighbour with stray non-zeros: recomputed from full rows by k_c4_fix
    // this vertex's own values in the table's columns (sign bit of the first: it has non-zeros elsewhere too)
    const float4 t = (table_in != nullptr && npass == 1u) ? table_in[u] : make_float4(-0.0f, 0.f, 0.f, 0.f);
    const bool stray = (__float_as_uint(t.x) >> 31) != 0;
    const float f_deg = (float)deg, f_w = (float)g.w[u] / ws, f_nw = (float)g.nw[u] / ws;
    const float *W1 = P, *b1 = W1 + 35 * N1;
    const float *W2 = b1 + N1, *b2 = W2 + N1 * N2;
    const float *W3 = b2 + N2, *b3 = W3 + N2 * N3;
    float x1[N1];
    if (!__any(stray)) {
        // the own-row terms of the chain: k = 16 (h[0]), 17 .. 19, 16 + d for a table column d >= 4 (h[1..3] are overwritten by
        // degree and weights in the reference's layout); a column slot without a column reads 0xFFFFFFFF
        auto own_terms = [&]() {
            if (d0 == 0u) fma_term<N1>(x1, t.x, W1 + 16 * N1);
            fma_term<N1>(x1, f_deg, W1 + 17 * N1);
            fma_term<N1>(x1, f_w, W1 + 18 * N1);
            fma_term<N1>(x1, f_nw, W1 + 19 * N1);
            if (d0 >= 4u && d0 < 16u) fma_term<N1>(x1, t.x, W1 + (16u + d0) * N1);
            if (d1 >= 4u && d1 < 16u) fma_term<N1>(x1, t.y, W1 + (16u + d1) * N1);
            if (d2 >= 4u && d2 < 16u) fma_term<N1>(x1, t.z, W1 + (16u + d2) * N1);
            if (d3 >= 4u && d3 < 16u) fma_term<N1>(x1, t.w, W1 + (16u + d3) * N1);
        };
#pragma unroll
        for (int j = 0; j < N1; ++j) x1[j] = 0.0f;
        if (!__any(dirty)) {   // (A)
            if (d0 < 16u) fma_term<N1>(x1, a.x, W1 + d0 * N1);
            if (d1 < 16u) fma_term<N1>(x1, a.y, W1 + d1 * N1);
            if (d2 < 16u) fma_term<N1>(x1, a.z, W1 + d2 * N1);
            if (d3 < 16u) fma_term<N1>(x1, a.w, W1 + d3 * N1);
        } else {               // (B)
            const size_t slot = dirty ? (size_t)__float_as_uint(a.y) : 0;
            const float4 g0 = agg16[slot * 4], g1 = agg16[slot * 4 + 1], g2 = agg16[slot * 4 + 2], g3 = agg16[slot * 4 + 3];
            const float gg[16] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w, g3.x, g3.y, g3.z, g3.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float clean = ((uint32_t)k == d0) ? a.x : ((uint32_t)k == d1) ? a.y : ((uint32_t)k == d2) ? a.z : ((uint32_t)k == d3) ? a.w : 0.0f;
                fma_term<N1>(x1, dirty ? gg[k] : clean, W1 + k * N1);
            }
        }
        own_terms();
        bias_relu<N1>(x1, b1);
    } else {                   // (C)
        const float4 h0 = fin[(size_t)u * 4], h1 = fin[(size_t)u * 4 + 1], h2 = fin[(size_t)u * 4 + 2], h3 = fin[(size_t)u * 4 + 3];
        // first-layer inputs in k order: 0..15 aggregate, 16 = h[0], 17 = degree, 18 = W/ws, 19 = NW/ws, 20..31 = h[4..15]
        float x0[32];
#pragma unroll
        for (int k = 0; k < 16; ++k) x0[k] = 0.0f;
        for (uint32_t qp = 0; qp < npass; ++qp) {   // (uniform)
            const uint32_t *dc = c4desc + (qp == 0 ? 1 : 4 + 4 * qp);
            const uint32_t e0 = dc[0], e1 = dc[1], e2 = dc[2], e3 = dc[3];
            const float4 aq = qp == 0 ? a : acc4[(size_t)qp * g.n + u];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                x0[k] = ((uint32_t)k == e0) ? aq.x : ((uint32_t)k == e1) ? aq.y : ((uint32_t)k == e2) ? aq.z : ((uint32_t)k == e3) ? aq.w : x0[k];
        }
        if (__any(dirty)) {
            const size_t slot = dirty ? (size_t)__float_as_uint(a.y) : 0;
            const float4 g0 = agg16[slot * 4], g1 = agg16[slot * 4 + 1], g2 = agg16[slot * 4 + 2], g3 = agg16[slot * 4 + 3];
            const float gg[16] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w, g3.x, g3.y, g3.z, g3.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) x0[k] = dirty ? gg[k] : x0[k];
        }
        x0[16] = h0.x;
        x0[17] = f_deg;
        x0[18] = f_w;
        x0[19] = f_nw;
        x0[20] = h1.x; x0[21] = h1.y; x0[22] = h1.z; x0[23] = h1.w;
        x0[24] = h2.x; x0[25] = h2.y; x0[26] = h2.z; x0[27] = h2.w;
        x0[28] = h3.x; x0[29] = h3.y; x0[30] = h3.z; x0[31] = h3.w;
        dense<32, 32, N1, 0>(x0, x1, W1, b1);   // rows 32..34 of W1 meet exact zeros
    }
    float x2[N2], x3[N3];
    dense<N1, N1, N2, 0>(x1, x2, W2, b2);
    dense<N2, N2, N3, SIGMOID ? 1 : 0>(x2, x3, W3, b3);
    if constexpr (SIGMOID) {
        if (mine) {
            if (logits) logits[u] = x3[0];
            fout[u] = sigmoid_ref(x3[0]);
        }
    } else {
        // the wave's 64 x 16 outputs through LDS: every global store instruction writes 16 full rows
        float *T = lds[threadIdx.x >> 6];
        uint32_t nz = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            T[lane * kOutPitch + j] = x3[j];
            nz |= (x3[j] != 0.0f ? 1u : 0u) << j;
        }
        if (emit_counts) c4_emit(&T[lane * kOutPitch], nz, u, mine, lane, emit_spec, emit_table, emit_counts);
        wave_lds_sync();
        const int q = lane >> 2, c = lane & 3;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int src_lane = 16 * p + q;
            const float *src = &T[src_lane * kOutPitch + 4 * c];
            const float4 o = make_float4(src[0], src[1], src[2], src[3]);
            const uint32_t row = __shfl(u, src_lane);
            if (__shfl((int)mine, src_lane)) reinterpret_cast<float4 *>(fout)[(size_t)row * 4 + c] = o;
        }
    }
}

// ------------------------------------------------------------------ 16-wide stage from an L2-resident compact table
// (round 4; VERDICT r3 #3 — BASELINE configs[1], Erdős–Rényi 100 K / 1 M.)  A graph of 50 - 400 K vertices is too small for
// the compact-table PLAN (its regrouped entries and block sweeps pay from 2^18 vertices and 8 Mi entries on) and too large for
// its 64-byte feature rows to sit in an XCD's 4 MiB L2 (100 K vertices: 6.4 MB): the gathering tile kernel runs at the fabric's
// rate of line requests (2 M entries a stage at ~60 G/s: 31 of the forward's 93 us per 16-wide stage).  The compact table of
// such a graph — the four live columns, 16 bytes a vertex: 1.6 MB — DOES fit every L2, so a tile kernel that gathers table rows
// instead of feature rows is served by L2 hits, with no plan at all: one lane per vertex, S neighbours' 16-byte rows in flight,
// their four values added in stored order (the plain gather's order, column by column); a neighbour whose table row carries the
// flag "has non-zeros in other columns" makes the row DIRTY and the lane re-gathers that row's full 64-byte rows (rare); the
// dense layers as in k_dense_f16 (routes A / B / C).  Who writes the table: the kernel that PRODUCES the stage's input (c4_emit:
// the F = 1 stage's VALU epilogue, or this kernel's own epilogue for the next stage — into a SECOND table, since its other
// workgroups are still gathering from the first), for the columns the previous forward chose; every workgroup re-derives this
// forward's choice from the producer's counters (the arithmetic of k_c4_choose, one table) and runs only if the table in place
// was written for exactly those columns — otherwise it leaves at once and the gathering kernel launched behind it (which
// leaves at once when this one ran: *fit) does the stage.  A graph's first forward therefore gathers the plain way (nobody has
// chosen columns yet) and the steady state takes the table: this is for graphs that are scored again and again.
// desc_in = this stage's descriptor as the producer saw it ([0] = 1: a table was written for columns [1..4]); desc_out (another
// buffer: nobody may still be reading what block 0 writes) = this forward's choice, the next forward's spec; desc_out[8] = fit.
struct T4Choice {
    uint32_t fit, d0, d1, d2, d3;
};
__device__ __forceinline__ T4Choice t4_choose(const unsigned long long *__restrict__ counts, uint32_t n, const uint32_t *__restrict__ desc_in,
                                              uint32_t *__restrict__ desc_out, int lane, bool write) {
    // lane i < 17 ends up with the total of counter i (16 columns + the rows seen), as in k_c4_choose
    unsigned long long part[kEmitStride], mine = 0;
#pragma unroll
    for (int t = 0; t < kEmitStride; ++t) part[t] = counts[t * 64 + lane];
    for (int c = 0; c < kEmitStride; ++c) {
        unsigned long long v = 0;
#pragma unroll
        for (int t = 0; t < kEmitStride; ++t) v += ((t * 64 + lane) % kEmitStride == c) ? part[t] : 0ull;
#pragma unroll
        for (int off = 32; off; off >>= 1)
            v += ((unsigned long long)__shfl_xor((unsigned)(v >> 32), off) << 32) | __shfl_xor((unsigned)v, off);
        if (lane == c) mine = v;
    }
    const unsigned long long rows = ((unsigned long long)__shfl((unsigned)(mine >> 32), 16) << 32) | __shfl((unsigned)mine, 16);
    const bool prev_ok = desc_in[0] == 1u && rows == n;   // (the tiles' own test)
    const uint32_t p1 = desc_in[1], p2 = desc_in[2], p3 = desc_in[3], p4 = desc_in[4];
    T4Choice r = {0u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t np = 0;
    if (rows == n) {
        const unsigned long long c = lane < 16 ? mine : 0ull;
        uint32_t rank = 0;   // how many columns are fuller than column `lane` (ties: lowest index first)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned long long ck = ((unsigned long long)__shfl((unsigned)(c >> 32), k) << 32) | __shfl((unsigned)c, k);
            rank += (ck > c || (ck == c && k < lane)) ? 1u : 0u;
        }
        unsigned long long rest = (lane < 16 && rank >= 4u) ? c : 0ull;
#pragma unroll
        for (int off = 8; off; off >>= 1)
            rest += ((unsigned long long)__shfl_xor((unsigned)(rest >> 32), off) << 32) | __shfl_xor((unsigned)rest, off);
        rest = ((unsigned long long)__shfl((unsigned)(rest >> 32), 0) << 32) | __shfl((unsigned)rest, 0);
        if (rest <= (unsigned long long)n / 512) np = 1;
        const unsigned long long cm = np ? __ballot(lane < 16 && rank < 4u) : 0ull;
        const uint32_t m0 = (uint32_t)cm, m1 = m0 & (m0 - 1), m2 = m1 & (m1 - 1), m3 = m2 & (m2 - 1);
        r.d0 = m0 ? (uint32_t)__builtin_ctz(m0) : 0xFFFFFFFFu;
        r.d1 = m1 ? (uint32_t)__builtin_ctz(m1) : 0xFFFFFFFFu;
        r.d2 = m2 ? (uint32_t)__builtin_ctz(m2) : 0xFFFFFFFFu;
        r.d3 = m3 ? (uint32_t)__builtin_ctz(m3) : 0xFFFFFFFFu;
        r.fit = (np == 1u && prev_ok && p1 == r.d0 && p2 == r.d1 && p3 == r.d2 && p4 == r.d3) ? 1u : 0u;
    }
    if (write && lane == 0) {
        desc_out[0] = np;
        desc_out[1] = r.d0; desc_out[2] = r.d1; desc_out[3] = r.d2; desc_out[4] = r.d3;
        desc_out[5] = 0; desc_out[6] = 0; desc_out[7] = 0;
        desc_out[8] = prev_ok ? 1u : 0u;   // what this launch's tiles did (the gathering kernel behind a non-solo launch skips on it)
    }
    return r;
}

template <int N1, int N2, int N3, bool SIGMOID, int S>
__global__ __launch_bounds__(kBlock) void k_stage_t4(GraphDev g, float ws, const float4 *__restrict__ fin, float *__restrict__ fout,
                                                     float *__restrict__ logits, const float *__restrict__ P, uint32_t row_lo,
                                                     uint32_t row_hi, int interleave, const float4 *__restrict__ table_in,
                                                     const unsigned long long *__restrict__ counts_in,
                                                     unsigned long long *__restrict__ counts_zero,
                                                     const uint32_t *__restrict__ desc_in, uint32_t *__restrict__ desc_out,
                                                     const uint32_t *__restrict__ emit_spec, c4row *__restrict__ emit_table,
                                                     unsigned long long *__restrict__ emit_counts, int solo) {
    // solo: no gathering kernel was launched behind this one (the host saw the previous forward fit and saves the ~6 us an
    // empty launch of that grid costs): if the table does NOT fit this forward after all, every row goes the way dirty rows and
    // stray vertices go — full rows, lane by lane: slow, rare, and the same bits
    static_assert(SIGMOID ? N3 == 1 : N3 == 16, "a feature stage writes 16 floats per row, the last stage one score");
    // (Half tiles — 32 vertices a wave, twice the waves — were built and measured: the gather phase takes the same 9 us, it is bound by
    // the CU's L2 -> L1 line fills, not by a wave's chain; ER-100K 87 us against 81.)
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][kRegionFloats];
    const int lane = threadIdx.x & 63;
    GNNVC_PROBE((SIGMOID ? 12 : 11), 0);
    if (blockIdx.x + 1u == gridDim.x) {
        // the launch's LAST workgroup has no tile: it makes this input's choice of columns — what the producers of the NEXT forward
        // write their tables for — while the others work.  (Every workgroup used to re-derive the choice first and run only if
        // the table in place was written for exactly those columns: 8 of a tile's 28 us on ER-100K, scratch/experiments/
        // phase_probe.py.  Nothing needs that: whatever columns the table holds, a vertex with non-zeros elsewhere is flagged and
        // a row that meets one is summed from full rows — the choice only decides how FEW such rows there are.)
        if (threadIdx.x < 64) t4_choose(counts_in, g.n, desc_in, desc_out, lane, true);
        // ... and clears the counters the NEXT forward's producer of this stage's input adds to (the set this forward's producer
        // used is still being read by the other workgroups; the two alternate with the descriptors — no memset between kernels,
        // which cost 5 us of stream time apiece)
        for (int i = threadIdx.x; i < kEmitSlots * kEmitStride; i += kBlock) counts_zero[i] = 0ull;
        return;
    }
    // The stage's parameters (2 736 floats) go through LDS, fetched once per workgroup with coalesced loads while the tiles'
    // prologues run: a mid-size graph has a wave or two per SIMD, a tile's chain of latencies IS the kernel, and 88 strided
    // 4-byte loads per lane for the matrix cores' A operands and biases were 2.5 us of it wherever they were put (in front of
    // their layers, or first of all — in front of everything else in the in-order load queue).
    static_assert(N1 == 32, "layer-2 input is a full 32-feature tile");
    constexpr int kParams = 35 * N1 + N1 + N1 * N2 + N2 + N2 * N3 + N3;
    constexpr int kParamLoads = (kParams + kBlock - 1) / kBlock;
    __shared__ float wl[kParams];
    float pw[kParamLoads];
#pragma unroll
    for (int i = 0; i < kParamLoads; ++i) {
        const int at = i * kBlock + (int)threadIdx.x;
        pw[i] = at < kParams ? P[at] : 0.0f;
    }
    const float *W1 = wl, *b1 = W1 + 35 * N1;
    const float *W2 = b1 + N1, *b2 = W2 + N1 * N2;
    const float *W3 = b2 + N2, *b3 = W3 + N2 * N3;
    const int mv = lane & 31, mh = lane >> 5;
    // the table in place is this input's: the producer was told to write it (for columns desc_in[1..4]) AND its counters say that
    // it has written all n rows (a producing kernel that does not emit leaves them short: the stage then gathers the plain way)
    uint32_t seen = (uint32_t)counts_in[lane * kEmitStride + 16];
#pragma unroll
    for (int off = 32; off; off >>= 1) seen += __shfl_xor(seen, off);
    const bool fit = desc_in[0] == 1u && seen == g.n;
    GNNVC_PROBE((SIGMOID ? 12 : 11), 1);
    if (!fit && !solo) return;     // uniform: the gathering kernel behind this one has the launch
    const uint32_t d0 = desc_in[1], d1 = desc_in[2], d2 = desc_in[3], d3 = desc_in[4];
    uint32_t es_ok = 0, es0 = 0, es1 = 0, es2 = 0, es3 = 0;   // what THIS kernel's epilogue writes for the next stage (loaded now: c4_emit_with)
    if (!SIGMOID && emit_counts) {   // (one go, unconditionally: nothing waits for these before the epilogue)
        es_ok = emit_spec[0]; es0 = emit_spec[1] & 15u; es1 = emit_spec[2] & 15u; es2 = emit_spec[3] & 15u; es3 = emit_spec[4] & 15u;
    }
    float *T = lds[threadIdx.x >> 6];
    uint32_t *stage = reinterpret_cast<uint32_t *>(T);
    const uint32_t ntiles = (row_hi - row_lo + kWave - 1) / kWave;
    const uint32_t tile_w = tile_for_wave(ntiles, interleave != 0);
    const bool has_tile = tile_w < ntiles;            // (a wave without one still brings its share of the parameters)
    const uint32_t tile = has_tile ? tile_w : 0u;
    const uint32_t v0 = row_lo + tile * kWave;
    const uint32_t uu = v0 + lane;
    const bool mine = has_tile && uu < row_hi;
    const uint32_t u = mine ? uu : row_hi - 1;
    const uint32_t rs = g.rowptr[u], re = mine ? g.rowptr[u + 1] : rs;
    const uint32_t deg = g.rowptr[u + 1] - g.rowptr[u];
    // the tile's slice of col, staged in LDS with full-line loads (as the gathering tile kernels do)
    const uint32_t c0 = __builtin_amdgcn_readfirstlane(g.rowptr[v0]);
    const uint32_t vend = (v0 + kWave < row_hi) ? v0 + kWave : row_hi;
    const uint32_t c1 = g.rowptr[vend];
    const bool staged = (c1 - c0) <= kStageCap;
    uint32_t sbase = 0;
    if (staged) sbase = stage_cols(g.col, c0, c1, stage, lane);
#pragma unroll
    for (int i = 0; i < kParamLoads; ++i) {
        const int at = i * kBlock + (int)threadIdx.x;
        if (at < kParams) wl[at] = pw[i];
    }
    __syncthreads();
    if (!has_tile) return;
    const float4 t = fit ? table_in[u] : make_float4(-0.0f, 0.f, 0.f, 0.f);   // this vertex's own values in the table's columns
    const bool stray = (__float_as_uint(t.x) >> 31) != 0;      // ... and whether it has non-zeros elsewhere (no table: as if)
    const float f_deg = (float)deg, f_w = (float)g.w[u] / ws, f_nw = (float)g.nw[u] / ws;
    wave_lds_sync();
    GNNVC_PROBE((SIGMOID ? 12 : 11), 2);
    const uint32_t zrow = g.n;                                 // table row n: zeros
    c4row acc = {0.0f, 0.0f, 0.0f, 0.0f};
    // A neighbour whose table row is FLAGGED has non-zeros outside the table's columns.  Its four table values are still exact
    // (they are >= 0: the flag sits in the first one's sign bit and is stripped before the add), so the row's four table sums
    // stand; what is missing are its sums in the OTHER columns, and only flagged neighbours contribute to those — x + (+-0) == x
    // and the sums start at +0: leaving the others out changes no bit.  So a lane notes the (up to two) flagged neighbours it met
    // and afterwards adds their full rows' other columns, in stored order: one round of loads for the one row in three hundred
    // that meets one, instead of the row's whole plain gather (which held one wave in five for 6 us at the end of the launch).
    // A row that met three or more, or a neighbour flagged in its SECOND value too (a value that cannot lend its sign bit, see
    // c4_emit_with), is summed from the full rows by the whole wave as before.
    uint32_t nflag = 0, f1st = zrow, f2nd = zrow;
    bool full = false;
    for (uint32_t eb = rs; fit && eb < re; eb += S) {
        c4row tv[S];
        uint32_t id[S];
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) {
            const uint32_t ee = eb + s2;
            const uint32_t cv = staged ? stage[ee - sbase] : g.col[ee];
            id[s2] = (ee < re) ? cv : zrow;
            tv[s2] = reinterpret_cast<const c4row *>(table_in)[id[s2]];
        }
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) {
            const bool fl = (__float_as_uint(tv[s2][0]) >> 31) != 0;
            full |= (__float_as_uint(tv[s2][1]) >> 31) != 0;
            if (fl) {
                f2nd = nflag == 1u ? id[s2] : f2nd;
                f1st = nflag == 0u ? id[s2] : f1st;
                ++nflag;
            }
            acc[0] += __builtin_fabsf(tv[s2][0]);
            acc[1] += tv[s2][1];
            acc[2] += tv[s2][2];
            acc[3] += tv[s2][3];
        }
    }
    full |= nflag > 2u;
    const bool part = !full && nflag != 0u;
    GNNVC_PROBE((SIGMOID ? 12 : 11), 3);
    float gg[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) gg[k] = 0.0f;
    if (!fit) {
        // (uniform) a solo launch whose table is not there after all — the input changed character since the verdict the host
        // went by: every lane sums its own row from the full rows, neighbour by neighbour.  Slow (a tile takes as long as its
        // longest row's chain of 64-byte fetches), correct, and over as soon as the host sees this forward's verdict.
        for (uint32_t e2 = rs; e2 < re; ++e2) {
            const uint32_t cv = staged ? stage[e2 - sbase] : g.col[e2];
            const float4 r0 = fin[(size_t)cv * 4], r1 = fin[(size_t)cv * 4 + 1], r2 = fin[(size_t)cv * 4 + 2], r3 = fin[(size_t)cv * 4 + 3];
            gg[0] += r0.x; gg[1] += r0.y; gg[2] += r0.z; gg[3] += r0.w;
            gg[4] += r1.x; gg[5] += r1.y; gg[6] += r1.z; gg[7] += r1.w;
            gg[8] += r2.x; gg[9] += r2.y; gg[10] += r2.z; gg[11] += r2.w;
            gg[12] += r3.x; gg[13] += r3.y; gg[14] += r3.z; gg[15] += r3.w;
        }
    }
    if (part) {   // (divergent: the few lanes with such a row)
        const float4 *r1 = fin + (size_t)f1st * 4, *r2 = fin + (size_t)f2nd * 4;   // (f2nd = the zero row when there is no second)
        const float4 p0 = r1[0], p1 = r1[1], p2 = r1[2], p3 = r1[3], q0 = r2[0], q1 = r2[1], q2 = r2[2], q3 = r2[3];
        const float pa[16] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w, p3.x, p3.y, p3.z, p3.w};
        const float qa[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) gg[k] = (0.0f + pa[k]) + qa[k];   // (the table's own columns are overwritten with acc below)
    }
    // a row for the full way: its sixteen sums from the full rows, in stored order (the plain gather of that row), by the WAVE:
    // lane 16 j + c fetches column c of the row's neighbours e + j, e + 4 + j, ... — 32 neighbours' rows in flight — and lane c
    // adds them in stored order
    for (unsigned long long left = __ballot(full); left; left &= left - 1) {
        const int L = __ffsll((long long)left) - 1;
        const uint32_t drs = (uint32_t)__shfl((int)rs, L), dre = (uint32_t)__shfl((int)re, L);
        const int j = lane >> 4, cc = lane & 15;
        const float *__restrict__ fsc = reinterpret_cast<const float *>(fin);
        float a = 0.0f;
        for (uint32_t e0 = drs; e0 < dre; e0 += 32) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t ea = e0 + 4 * k + j;
                const uint32_t cv = ea < dre ? (staged ? stage[ea - sbase] : g.col[ea]) : zrow;   // (row n of the features: zeros)
                v[k] = fsc[(size_t)cv * 16 + cc];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) a += __shfl(v[k], 16 * jj + cc);   // (lanes 0..15 hold the row's sums; a slot past the end adds +0)
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float sk = __shfl(a, k);
            if (lane == L) gg[k] = sk;
        }
    }
    GNNVC_PROBE((SIGMOID ? 12 : 11), 4);
    // ---- the first layer's 32 inputs of this lane's vertex, in k order, into the wave's LDS tile: 0..15 the sums (four table
    // columns, or a dirty row's sixteen), 16 = h[0], 17 = degree, 18 = W/ws, 19 = NW/ws, 20..31 = h[4..15].  The vertex's own
    // values come from its table row — a vertex that is not flagged has nothing outside the table's columns — or, flagged (or no
    // table), from its full row.
    float4 h0 = make_float4(0.f, 0.f, 0.f, 0.f), h1 = h0, h2 = h0, h3 = h0;
    if (stray) {
        h0 = fin[(size_t)u * 4]; h1 = fin[(size_t)u * 4 + 1]; h2 = fin[(size_t)u * 4 + 2]; h3 = fin[(size_t)u * 4 + 3];
    }
    const float own_t[4] = {__uint_as_float(__float_as_uint(t.x) & 0x7FFFFFFFu), t.y, t.z, t.w};
    const float own_f[16] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z, h2.w, h3.x, h3.y, h3.z, h3.w};
    wave_lds_sync();       // the index stage is dead: the region becomes the input tile (row = vertex, column = k)
    {
        float *row = &T[lane * kInPitch];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            // a table column's sum from acc (unless the row went the full way), any other column's from the flagged neighbours (+0 if none)
            const float v = ((uint32_t)k == d0) ? acc[0] : ((uint32_t)k == d1) ? acc[1] : ((uint32_t)k == d2) ? acc[2] : ((uint32_t)k == d3) ? acc[3] : gg[k];
            row[k] = (full || !fit) ? gg[k] : v;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k >= 1 && k <= 3) continue;   // h[1..3]: overwritten by degree / weights in the reference's layout
            const float tab = ((uint32_t)k == d0) ? own_t[0] : ((uint32_t)k == d1) ? own_t[1] : ((uint32_t)k == d2) ? own_t[2] : ((uint32_t)k == d3) ? own_t[3] : 0.0f;
            row[16 + k] = stray ? own_f[k] : tab;
        }
        row[17] = f_deg; row[18] = f_w; row[19] = f_nw;
    }
    wave_lds_sync();
    GNNVC_PROBE((SIGMOID ? 12 : 11), 5);
    // ---- the dense layers on the matrix cores (k-ordered fma chains: the VALU's bits), A operands fetched before the gather
    f32x16 d[2];
    {
        float a1[16];
        mfma_load_a<N1, 16>(W1, mv, mh, a1);   // rows 0..31 of W1; rows 32..34 meet exact zeros
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            d[vt] = mfma_layer_lds(T, vt, mv, mh, a1);
            mfma_bias_act<N1, 0>(d[vt], b1, mh);
        }
    }
    {
        float a2[16];
        mfma_load_a<N2, 16>(W2, mv, mh, a2);
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            d[vt] = mfma_layer_acc<32>(d[vt], a2);
            mfma_bias_act<N2, 0>(d[vt], b2, mh);
        }
    }
    GNNVC_PROBE((SIGMOID ? 12 : 11), 6);
    if constexpr (SIGMOID) {
        static_assert(N2 == 16 && N3 == 1, "sigmoid tail is 16 -> 1");
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            const float logit = mfma_tail_16to1(d[vt], W3, mh) + b3[0];   // (valid on the high half lane of vertex 32 vt + v)
            const uint32_t uv = __shfl(u, 32 * vt + mv);
            const int keep = __shfl((int)mine, 32 * vt + mv);
            if (mh == 1 && keep) {
                if (logits) logits[uv] = logit;
                fout[uv] = sigmoid_ref(logit);
            }
        }
    } else {
        float a3[16];
        mfma_load_a<N3, N2 / 2>(W3, mv, mh, a3);
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            d[vt] = mfma_layer_acc<N2>(d[vt], a3);
            mfma_bias_act<N3, 0>(d[vt], b3, mh);
        }
        GNNVC_PROBE((SIGMOID ? 12 : 11), 8);
        wave_lds_sync();   // the input tile is consumed: the region becomes the output tile
#pragma unroll
        for (int vt = 0; vt < 2; ++vt)
#pragma unroll
            for (int r = 0; r < 8; ++r) T[(32 * vt + mv) * kOutPitch + mfma_feat(r, mh)] = d[vt][r];
        wave_lds_sync();
        uint32_t nz = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) nz |= (T[lane * kOutPitch + j] != 0.0f ? 1u : 0u) << j;
        GNNVC_PROBE((SIGMOID ? 12 : 11), 9);
        if (emit_counts) c4_emit_with(&T[lane * kOutPitch], nz, u, mine, lane, es_ok, es0, es1, es2, es3, emit_table, emit_counts);
        GNNVC_PROBE((SIGMOID ? 12 : 11), 10);
        const int q = lane >> 2, c = lane & 3;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int src_lane = 16 * p + q;
            const float *src = &T[src_lane * kOutPitch + 4 * c];
            const float4 o = make_float4(src[0], src[1], src[2], src[3]);
            const uint32_t row = __shfl(u, src_lane);
            if (__shfl((int)mine, src_lane)) reinterpret_cast<float4 *>(fout)[(size_t)row * 4 + c] = o;
        }
    }
    GNNVC_PROBE((SIGMOID ? 12 : 11), 11);
}

// ------------------------------------------------------------------ wide tiles: a tile on FOUR waves (small graphs)
// (round 4; VERDICT r3: the reference CLI's predict calls 2..n score graphs of 41 423, 19 675, 7 375, 1 933 vertices,
// src/GNN_VC.cpp:171-192.)  Below ~25 K vertices the chip has fewer tiles than SIMDs and a kernel takes as long as ONE tile's
// chain on ONE wave — phase stamps (scratch/experiments/phase_probe.py): column ids 2 us, gather 3 us, dense layers 6 - 7 us (96
// dependent 32x32x2 MFMAs, or as many VALU cycles), stores 0.5 us, + ~4 us to launch: 15 us a kernel for 2 000 vertices as for
// 20 000.  Here a WORKGROUP owns the 64-vertex tile: in the gather a quad of lanes owns a vertex (wave q: vertices 16 q .. 16 q +
// 15; four neighbour rows in flight per vertex — a quarter of the rows and half the rounds per wave), and in the dense layers
// lane L of wave q computes a QUARTER of every layer's outputs of vertex L (outputs [PER q, PER q + PER)) with the layers'
// activations exchanged through LDS: every output is still one k-ordered fma chain from +0.0f, then the separately rounded bias
// add — the bits of dense<>().  No long rows, no sorted tiles, no pruned adjacency, no emit: graphs that small have none of it.
template <int K, int KUSED, int N, int PER, int ACT>
__device__ __forceinline__ void dense_cols(const float (&in)[K], float (&out)[PER], const float *__restrict__ W,
                                           const float *__restrict__ b, int col0 /* wave-uniform */) {
#pragma unroll
    for (int j = 0; j < PER; ++j) out[j] = 0.0f;
#pragma unroll
    for (int k = 0; k < KUSED; ++k) {
        const float a = in[k];
#pragma unroll
        for (int j = 0; j < PER; ++j) out[j] = __builtin_fmaf(a, W[k * N + col0 + j], out[j]);
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const float t = out[j] + b[col0 + j];
        out[j] = (ACT == 0) ? relu_ref(t) : t;
    }
}

constexpr int kWidePitch = 33;

// layers 2 and 3 of a wide tile (layer-1 outputs of this lane's vertex, PER1 of them, in o1): shared by both stage kinds
template <int N1, int N2, int N3, bool SIGMOID>
__device__ __forceinline__ void wide_tail(const float (&o1)[N1 / 4], float *A, float *B, const float *__restrict__ W2,
                                          const float *__restrict__ b2, const float *__restrict__ W3, const float *__restrict__ b3,
                                          int lane, int q, uint32_t v0, uint32_t row_hi, float *__restrict__ fout,
                                          float *__restrict__ logits) {
    static_assert(N1 % 4 == 0 && N2 % 4 == 0, "a quarter of a layer's outputs per wave");
    constexpr int P1 = N1 / 4, P2 = N2 / 4;
#pragma unroll
    for (int j = 0; j < P1; ++j) B[lane * kWidePitch + P1 * q + j] = o1[j];
    __syncthreads();
    float x1[N1];
#pragma unroll
    for (int k = 0; k < N1; ++k) x1[k] = B[lane * kWidePitch + k];
    float o2[P2];
    dense_cols<N1, N1, N2, P2, 0>(x1, o2, W2, b2, P2 * q);
#pragma unroll
    for (int j = 0; j < P2; ++j) A[lane * kWidePitch + P2 * q + j] = o2[j];   // (A's inputs were consumed before the barrier above)
    __syncthreads();
    float x2[N2];
#pragma unroll
    for (int k = 0; k < N2; ++k) x2[k] = A[lane * kWidePitch + k];
    if constexpr (SIGMOID) {
        static_assert(N3 == 1, "sigmoid stage ends in one output");
        if (q == 0) {   // (wave-uniform)
            float o3[1];
            dense_cols<N2, N2, 1, 1, 1>(x2, o3, W3, b3, 0);
            const uint32_t u = v0 + lane;
            if (u < row_hi) {
                if (logits) logits[u] = o3[0];
                fout[u] = sigmoid_ref(o3[0]);
            }
        }
    } else {
        static_assert(N3 == 16, "feature stages emit 16 floats per vertex");
        constexpr int P3 = N3 / 4;
        float o3[P3];
        dense_cols<N2, N2, N3, P3, 0>(x2, o3, W3, b3, P3 * q);
#pragma unroll
        for (int j = 0; j < P3; ++j) B[lane * kOutPitch + P3 * q + j] = o3[j];   // (B's layer-1 outputs were consumed before the second barrier)
        __syncthreads();
        const int row = threadIdx.x >> 2, c = threadIdx.x & 3;   // 256 threads: 64 rows x four 16-byte pieces
        const float *src = &B[row * kOutPitch + 4 * c];
        if (v0 + row < row_hi) reinterpret_cast<float4 *>(fout)[(size_t)(v0 + row) * 4 + c] = make_float4(src[0], src[1], src[2], src[3]);
    }
}

template <int N1, int N2, int N3, bool SIGMOID>
__global__ __launch_bounds__(kBlock) void k_stage_w16(GraphDev g, float ws, const float4 *__restrict__ fin, float *__restrict__ fout,
                                                      float *__restrict__ logits, const float *__restrict__ P, uint32_t row_lo,
                                                      uint32_t row_hi) {
    __shared__ float A[kWave * kWidePitch], B[kWave * kWidePitch];
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t v0 = row_lo + blockIdx.x * kWave;
    {   // ---- gather: a quad of lanes per vertex, lane c holds floats 4c .. 4c + 3 of a 64-byte row
        const int vq = 16 * q + (lane >> 2), c = lane & 3;
        const uint32_t uu = v0 + vq;
        const bool valid = uu < row_hi;
        const uint32_t u = valid ? uu : row_hi - 1;
        const uint32_t rs = g.rowptr[u], re_full = g.rowptr[u + 1];
        const uint32_t re = valid ? re_full : rs;
        const float4 self = fin[(size_t)u * 4 + c];
        const float f_w = (float)g.w[u] / ws, f_nw = (float)g.nw[u] / ws;
        const uint32_t zrow = g.n;   // all-zero pad row: x + 0.0f == x exactly
        constexpr int S = 4;
        c4row acc = {0.0f, 0.0f, 0.0f, 0.0f};
        const c4row *__restrict__ fv = reinterpret_cast<const c4row *>(fin);
        uint32_t nxt[S];
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) nxt[s2] = g.col[rs + s2];   // (col is padded past nnz: readable; masked below)
        for (uint32_t e = rs; e < re; e += S) {
            c4row r[S];
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) r[s2] = fv[(size_t)((e + s2 < re) ? nxt[s2] : zrow) * 4 + c];
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) nxt[s2] = g.col[e + S + s2];   // the next round's ids fly with this round's rows
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) acc += r[s2];                  // stored order, one rounded add per neighbour and column
        }
        float *row = &A[vq * kWidePitch];
        row[4 * c + 0] = acc[0]; row[4 * c + 1] = acc[1]; row[4 * c + 2] = acc[2]; row[4 * c + 3] = acc[3];
        if (c == 0) {
            row[16] = self.x;                    // h[0]; h[1..3] are overwritten by degree / weights in the reference's layout
            row[17] = (float)(re_full - rs);
            row[18] = f_w;
            row[19] = f_nw;
        } else {
            float *d = &row[20 + 4 * (c - 1)];   // h[4..15]
            d[0] = self.x; d[1] = self.y; d[2] = self.z; d[3] = self.w;
        }
    }
    __syncthreads();
    const float *W1 = P, *b1 = W1 + 35 * N1;
    const float *W2 = b1 + N1, *b2 = W2 + N1 * N2;
    const float *W3 = b2 + N2, *b3 = W3 + N2 * N3;
    float x0[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) x0[k] = A[lane * kWidePitch + k];
    float o1[N1 / 4];
    dense_cols<32, 32, N1, N1 / 4, 0>(x0, o1, W1, b1, (N1 / 4) * q);   // rows 32..34 of W1 meet exact zeros
    wide_tail<N1, N2, N3, SIGMOID>(o1, A, B, W2, b2, W3, b3, lane, q, v0, row_hi, fout, logits);
}

template <int N1, int N2, int N3>
__global__ __launch_bounds__(kBlock) void k_stage_w1(GraphDev g, float ws, const float *__restrict__ xin, float *__restrict__ fout,
                                                     const float *__restrict__ P, uint32_t row_lo, uint32_t row_hi) {
    __shared__ float A[kWave * kWidePitch], B[kWave * kWidePitch];
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t v0 = row_lo + blockIdx.x * kWave;
    {   // ---- gather: wave q sums vertices 16 q .. 16 q + 15, a QUAD of lanes per vertex: lane j of the quad fetches neighbours e + j and
        // e + 4 + j of a round of eight, and every lane of the quad adds the eight values in stored order (quad broadcasts): the
        // chain of one lane per vertex with four times its loads in flight
        const int vq = 16 * q + (lane >> 2), j = lane & 3;
        const uint32_t uu = v0 + vq;
        const bool valid = uu < row_hi;
        const uint32_t u = valid ? uu : row_hi - 1;
        const uint32_t rs = g.rowptr[u], re_full = g.rowptr[u + 1];
        const uint32_t re = valid ? re_full : rs;
        const float xself = xin[u];
        float agg = 0.0f;
        uint32_t n0 = g.col[rs + j], n1 = g.col[rs + 4 + j];   // (col is padded past nnz: readable; masked below)
        for (uint32_t e = rs; e < re; e += 8) {
            const bool in0 = e + j < re, in1 = e + 4 + j < re;
            const float r0 = xin[in0 ? n0 : u], r1 = xin[in1 ? n1 : u];
            n0 = g.col[e + 8 + j];
            n1 = g.col[e + 12 + j];
            const float a0 = in0 ? r0 : 0.0f, a1 = in1 ? r1 : 0.0f;   // agg is never -0.0f, so + 0.0f is exact
#define GNNVC_QUAD_BCAST(x_, t_) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x_), (t_) * 0x55, 0xf, 0xf, false))
            agg += GNNVC_QUAD_BCAST(a0, 0);
            agg += GNNVC_QUAD_BCAST(a0, 1);
            agg += GNNVC_QUAD_BCAST(a0, 2);
            agg += GNNVC_QUAD_BCAST(a0, 3);
            agg += GNNVC_QUAD_BCAST(a1, 0);
            agg += GNNVC_QUAD_BCAST(a1, 1);
            agg += GNNVC_QUAD_BCAST(a1, 2);
            agg += GNNVC_QUAD_BCAST(a1, 3);
#undef GNNVC_QUAD_BCAST
        }
        if (j == 0) {
            float *row = &A[vq * kWidePitch];
            row[0] = agg;
            row[1] = xself;
            row[2] = (float)(re_full - rs);
            row[3] = (float)g.w[u] / ws;
            row[4] = (float)g.nw[u] / ws;
        }
    }
    __syncthreads();
    const float *W1 = P, *b1 = W1 + 5 * N1;
    const float *W2 = b1 + N1, *b2 = W2 + N1 * N2;
    const float *W3 = b2 + N2, *b3 = W3 + N2 * N3;
    float x0[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) x0[k] = A[lane * kWidePitch + k];
    float o1[N1 / 4];
    dense_cols<5, 5, N1, N1 / 4, 0>(x0, o1, W1, b1, (N1 / 4) * q);
    wide_tail<N1, N2, N3, false>(o1, A, B, W2, b2, W3, b3, lane, q, v0, row_hi, fout, nullptr);
}

// ------------------------------------------------------------------ stage, F = 1
// First dense layer input (f = 1): [aggregate, x, degree, W/ws, NW/ws].
// `ep` / `ecol` say which CSR entries this launch sums: the whole rows
// (ep = g.rowptr, ecol = g.col, acc_in = nullptr) or, in the column-blocked plan,
// the rows' entries of the LAST column block with the partial sums of the earlier
// blocks arriving in acc_in (same add sequence as the unblocked loop).
template <int N1, int N2, int N3, int S, bool MFMA>
__global__ __launch_bounds__(kBlock) void k_stage_f1(
        GraphDev g, float ws, const float *__restrict__ xin, float *__restrict__ fout,
        const float *__restrict__ P, uint32_t row_lo, uint32_t row_hi,
        const uint32_t *__restrict__ ep, const uint32_t *__restrict__ ecol,
        const float *__restrict__ acc_in, uint32_t long_thresh, int interleave,
        const uint32_t *__restrict__ acc_bad,
        const uint32_t *__restrict__ emit_spec, c4row *__restrict__ emit_table, unsigned long long *__restrict__ emit_counts,
        const uint32_t *__restrict__ srt_vertex, const uint4 *__restrict__ srt_meta, uint32_t n_sorted) {
    // acc_bad (LDS-table plan only): *acc_bad == 0 -> acc_in holds the rows' COMPLETE sums and no entry is left
    // to add; != 0 -> the plan did not apply to this input, acc_in is ignored and every entry is gathered here.
    // srt_vertex (skewed graphs): tiles of 64 vertices of similar degree from the degree-sorted list (see
    // k_stage_f16) instead of 64 consecutive rows, so a tile's gather rounds are not set by one heavy row.
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][kRegionFloats];
    const bool acc_full = acc_bad && *acc_bad == 0;
    if (acc_bad && !acc_full) acc_in = nullptr;
    const bool sorted = srt_vertex != nullptr;   // uniform
    const int lane = threadIdx.x & 63;
    float *T = lds[threadIdx.x >> 6];
    uint32_t *stage = reinterpret_cast<uint32_t *>(T);
    const uint32_t ntiles = sorted ? (n_sorted + kWave - 1) / kWave : (row_hi - row_lo + kWave - 1) / kWave;
    uint32_t tile = tile_for_wave(ntiles, sorted || interleave != 0);
    if (tile >= ntiles) return;
    if (sorted && GNNVC_SORTED_MIX) tile = (tile & 1u) ? ntiles - 1u - (tile >> 1) : (tile >> 1);   // (heavy and light tiles side by side: see k_stage_f16)
    const uint32_t v0 = row_lo + tile * kWave;   // natural order only
    uint32_t u, uc, deg, rs, re;
    bool mine, staged = false;
    uint32_t sbase = 0;
    float f_w, f_nw;
    if (sorted) {
        const uint32_t slot = tile * kWave + lane;
        const bool in = slot < n_sorted;
        const uint4 meta = srt_meta[in ? slot : n_sorted - 1];
        u = uc = srt_vertex[in ? slot : n_sorted - 1];
        deg = meta.y - meta.x;
        mine = in && deg < long_thresh;           // the list also holds rows this stage leaves to k_long_f1
        rs = meta.x;
        re = mine ? meta.y : rs;
        f_w = (float)meta.z / ws;
        f_nw = (float)meta.w / ws;
    } else {
        u = v0 + lane;
        const bool valid = u < row_hi;
        uc = valid ? u : row_hi - 1;
        deg = g.rowptr[uc + 1] - g.rowptr[uc];
        mine = valid && deg < long_thresh;        // long rows belong to k_long_f1
        rs = ep[uc];
        re = (mine && !acc_full) ? ep[uc + 1] : rs;
        const uint32_t c0 = __builtin_amdgcn_readfirstlane(rs);
        const uint32_t vend = (v0 + kWave < row_hi) ? v0 + kWave : row_hi;
        const uint32_t c1 = acc_full ? c0 : ep[vend];   // wave-uniform: end of the tile's last valid row
        staged = (c1 - c0) <= kStageCap;
        if (staged) sbase = stage_cols(ecol, c0, c1, stage, lane);
        f_w = (float)g.w[uc] / ws;
        f_nw = (float)g.nw[uc] / ws;
    }
    const float f_deg = (float)deg;
    const float xself = xin[uc];
    float agg = acc_in ? acc_in[uc] : 0.0f;
    wave_lds_sync();

    for (uint32_t eb = rs; eb < re; eb += S) {
        float xs[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t ee = eb + s;
            const uint32_t cv = staged ? stage[ee - sbase] : ecol[ee];
            const float v = xin[(ee < re) ? cv : uc];
            xs[s] = (ee < re) ? v : 0.0f;  // agg is never -0.0f, so + 0.0f is exact
        }
#pragma unroll
        for (int s = 0; s < S; ++s) agg += xs[s];
    }
    float x0[5];
    x0[0] = agg;
    x0[1] = xself;
    x0[2] = f_deg;
    x0[3] = f_w;
    x0[4] = f_nw;
    const float *W1 = P, *b1 = W1 + 5 * N1;
    const float *W2 = b1 + N1, *b2 = W2 + N1 * N2;
    const float *W3 = b2 + N2, *b3 = W3 + N2 * N3;
    float x1[N1];
    dense<5, 5, N1, 0>(x0, x1, W1, b1);   // 5 -> 32 stays on the VALU (K = 5 is not matrix-shaped)
    static_assert(N3 == 16, "feature stages emit 16 floats per vertex");
    const int q = lane >> 2, c = lane & 3;
    wave_lds_sync();  // index stage is dead; reuse the region
    if constexpr (MFMA) {
        // layer-1 activations through LDS into the matrix-core layout, layers 2-3 as MFMA
#pragma unroll
        for (int j = 0; j < N1; ++j) T[lane * kInPitch + j] = x1[j];
        wave_lds_sync();
        const int v = lane & 31, h = lane >> 5;
        f32x16 d[2];
        float a[16];
        mfma_load_a<N2, 16>(W2, v, h, a);
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            d[vt] = mfma_layer_lds(T, vt, v, h, a);
            mfma_bias_act<N2, 0>(d[vt], b2, h);
        }
        mfma_load_a<N3, N2 / 2>(W3, v, h, a);
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            d[vt] = mfma_layer_acc<N2>(d[vt], a);
            mfma_bias_act<N3, 0>(d[vt], b3, h);
        }
        wave_lds_sync();
#pragma unroll
        for (int vt = 0; vt < 2; ++vt)
#pragma unroll
            for (int r = 0; r < 8; ++r) T[(32 * vt + v) * kOutPitch + mfma_feat(r, h)] = d[vt][r];
    } else {
        float x2[N2], x3[N3];
        dense<N1, N1, N2, 0>(x1, x2, W2, b2);
        dense<N2, N2, N3, 0>(x2, x3, W3, b3);
        uint32_t nz = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            T[lane * kOutPitch + j] = x3[j];
            nz |= (x3[j] != 0.0f ? 1u : 0u) << j;
        }
        if (emit_counts) c4_emit(&T[lane * kOutPitch], nz, u, mine, lane, emit_spec, emit_table, emit_counts);
    }
    wave_lds_sync();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t row = __shfl(u, 16 * p + q);   // (= v0 + 16 p + q in natural order)
        const float *src = &T[(16 * p + q) * kOutPitch + 4 * c];
        const float4 o = make_float4(src[0], src[1], src[2], src[3]);
        if (__shfl((int)mine, 16 * p + q)) reinterpret_cast<float4 *>(fout)[(size_t)row * 4 + c] = o;
    }
}

// ------------------------------------------------------------------ long rows
// A row of degree d costs the tile kernels d / S dependent gather rounds, so one
// hub would hold its wave (and the kernel's tail) for milliseconds.  Rows of
// degree >= long_thresh are therefore listed once per graph and each gets a whole
// workgroup: all 256 threads fetch neighbour rows, 256 per round, into a
// double-buffered LDS slab while the next round's loads are in flight, and ONE
// lane per feature column adds the slab's rows in CSR order — the same sequential
// fp32 add chain as everywhere else, so results stay bit-identical.  The row's
// dense layers then run on one lane.  Runs on a second stream beside the tile
// kernel of the same stage (disjoint output rows).
__global__ __launch_bounds__(256) void k_find_long(GraphDev g, uint32_t thresh, uint32_t *__restrict__ list,
                                                   uint32_t *__restrict__ count) {
    // count[0] = rows listed; count[2..3] (one 64-bit word) = their entries.  A workgroup takes 1024 rows a trip (thread t: rows
    // base + 256 j + t) and reserves the slots of ALL their long rows with one atomic (round 4: one per row made 110 K atomics on one
    // word of R-MAT-22's 4 M rows, 0.35 ms for a 0.02 ms pass; one per wave was still 0.19 ms there, 0.40 ms on R-MAT-24, 66 us for
    // the power-law graph's 4 257 rows: same-address atomics are what the pass takes); a block adds its entries with one; the
    // list's order is free (every listed row gets a workgroup of its own).
    __shared__ unsigned long long part[4];
    __shared__ uint32_t wcnt[4][4];     // [j][wave]: long rows among the wave's 64 rows of sub-trip j
    __shared__ uint32_t trip_first;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lo = g.lo(), hi = g.hi();
    unsigned long long mine = 0;
    for (uint64_t base = lo + (uint64_t)blockIdx.x * 1024u; base < hi; base += (uint64_t)gridDim.x * 1024u) {   // (block-uniform)
        uint32_t d[4];
        unsigned long long m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t u = base + 256u * j + threadIdx.x;
            d[j] = u < hi ? g.rowptr[u + 1] - g.rowptr[u] : 0u;
            m[j] = __ballot(u < hi && d[j] >= thresh);
            if (lane == 0) wcnt[j][wave] = (uint32_t)__popcll(m[j]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) total += wcnt[k >> 2][k & 3];
            trip_first = total ? atomicAdd(count, total) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t u = base + 256u * j + threadIdx.x;
            if (u < hi && d[j] >= thresh) {
                uint32_t before = trip_first;   // the long rows of the sub-trips and waves in front of this wave's, then of its lower lanes
                for (int k = 0; k < 4 * j + (int)wave; ++k) before += wcnt[k >> 2][k & 3];
                list[before + (uint32_t)__popcll(m[j] & ((1ull << lane) - 1ull))] = (uint32_t)u;
                mine += d[j];
            }
        }
        __syncthreads();   // (wcnt and trip_first are the next trip's)
    }
#pragma unroll
    for (int off = 32; off; off >>= 1)
        mine += ((unsigned long long)__shfl_xor((unsigned)(mine >> 32), off) << 32) | __shfl_xor((unsigned)mine, off);
    if (lane == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(count + 2), t);
    }
}

// The dense layers of ONE row on one WAVE (round 3): lane j holds output j of a layer, input k arrives from lane k through
// v_readlane, and every output's chain still runs k = 0, 1, ... from +0.0f with one fused multiply-add per term and a separately
// rounded bias add — the order dense<>() uses, hence the same bits.  A long row's tail used to be one LANE walking all
// 2 656 (1 696) multiply-adds one after the other, ~15 - 30 us with the weights coming through a cold scalar cache — longer
// than gathering a 2 000-entry row; a wave does it in about a microsecond.
template <int K, int N, int ACT>
__device__ __forceinline__ float wave_layer(float xin, const float *__restrict__ W, const float *__restrict__ b, int lane) {
    const int j = lane < N ? lane : 0;
    float wv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) wv[k] = W[k * N + j];    // (one coalesced 4 N-byte load per k, all in flight at once)
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k)
        acc = __builtin_fmaf(__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(xin), k)), wv[k], acc);
    const float t = acc + b[j];
    return (ACT == 0) ? relu_ref(t) : t;
}

// x0: lane k holds input k of the first layer (K1 = 32 of the 35 for a 16-wide stage: the last three meet exact zeros; 5 for the
// F = 1 stage); returns this lane's output of the last layer (lanes >= N3: a copy of lane 0's)
template <int K1, int N1, int N2, int N3, bool SIGMOID>
__device__ __forceinline__ float wave_tail(const float *__restrict__ P, int k1_rows, float x0, int lane) {
    const float *W1 = P, *b1 = W1 + k1_rows * N1;
    const float *W2 = b1 + N1, *b2 = W2 + N1 * N2;
    const float *W3 = b2 + N2, *b3 = W3 + N2 * N3;
    const float x1 = wave_layer<K1, N1, 0>(x0, W1, b1, lane);
    const float x2 = wave_layer<N1, N2, 0>(x1, W2, b2, lane);
    return wave_layer<N2, N3, SIGMOID ? 1 : 0>(x2, W3, b3, lane);
}

template <int N1, int N2, int N3, bool SIGMOID>
__device__ __forceinline__ void wave_tail_f16(const GraphDev &g, float ws, const float *__restrict__ fin, float *__restrict__ fout,
                                              float *__restrict__ logits, const float *__restrict__ P, uint32_t u, uint32_t deg,
                                              float agg_lane /* lanes 0..15: the row's aggregate */, int lane) {
    // first-layer inputs in k order: 0..15 aggregate, 16 = h[0], 17 = degree, 18 = W/ws, 19 = NW/ws, 20..31 = h[4..15]
    float x0 = agg_lane;
    if (lane == 16) x0 = fin[(size_t)u * 16];
    if (lane == 17) x0 = (float)deg;
    if (lane == 18) x0 = (float)g.w[u] / ws;
    if (lane == 19) x0 = (float)g.nw[u] / ws;
    if (lane >= 20 && lane < 32) x0 = fin[(size_t)u * 16 + (lane - 16)];
    const float y = wave_tail<32, N1, N2, N3, SIGMOID>(P, 35, x0, lane);
    if constexpr (SIGMOID) {
        if (lane == 0) {
            if (logits) logits[u] = y;
            fout[u] = sigmoid_ref(y);
        }
    } else {
        if (lane < N3) fout[(size_t)u * N3 + lane] = y;
    }
}

template <int N1, int N2, int N3>
__device__ __forceinline__ void wave_tail_f1(const GraphDev &g, float ws, const float *__restrict__ xin, float *__restrict__ fout,
                                             const float *__restrict__ P, uint32_t u, uint32_t deg, float agg, int lane) {
    float x0 = agg;                                   // lane 0
    if (lane == 1) x0 = xin[u];
    if (lane == 2) x0 = (float)deg;
    if (lane == 3) x0 = (float)g.w[u] / ws;
    if (lane == 4) x0 = (float)g.nw[u] / ws;
    const float y = wave_tail<5, N1, N2, N3, false>(P, 5, x0, lane);
    if (lane < N3) fout[(size_t)u * N3 + lane] = y;
}

constexpr int kLongChunk = 256;

// One round of the long-row kernel: the 64 R neighbour rows a workgroup holds in registers (R per quad: rows q + 64 j of the
// round) go through the LDS slab 256 at a time and threads 0..15 add them, column by column, in CSR order.  R was 16 (rounds of
// 1024, 212 registers a wave, two workgroups per CU) until round 3: with the long rows on ONE side queue behind the giant rows
// (see launch_side_rows) their kernel is on a stage's critical path, most of its rows are a few hundred to a few thousand
// entries, and a workgroup's time on those is its start-up (two dependent fetches, a quarter, the dense tail) — with rounds
// of a QUARTER (R = 4: 104 registers, four workgroups per CU) R-MAT-22 went 2.63 -> 2.46 ms, R-MAT-24 10.5 -> 9.5, power-law 1 M
// 0.91 -> 0.86, R-MAT-20 0.94 -> 0.89 (R = 8: 2.52 / 9.85 / 0.87 / 0.91).
constexpr int kLongR = 4;
// first-class vector type (HIP's float4 is a struct: arrays of it that live across loop
// iterations are not promoted to registers)
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Slab layout: column-major, one padded run of kLongChunk neighbours per feature column, so that
// the adder lane of a column reads four consecutive neighbours with one ds_read_b128 (a single wave
// issues one instruction every ~4 cycles: with one LDS read per add the chain ran at ~11 cycles per
// neighbour, with one per four adds it runs at ~5).  The pad of 4 floats spreads the 16 adder lanes
// over all 64 banks (260 % 64 = 4) and the four column groups of the writers over 16-bank strides.
constexpr int kLongStride = kLongChunk + 4;

// (R: neighbour rows per quad and round; `qc` counts the quarters drained so far: the slab buffers alternate with IT, not with
// the quarter's place in its round — a round of R = 4 is ONE quarter)
template <int R>
__device__ __forceinline__ void long_drain_round(const f32x4 (&rows)[R], float (&slab)[2][kLongStride * 16],
                                                 uint32_t left_round, int tid, int q, int c, float &acc, uint32_t &qc) {
#pragma unroll
    for (int sub = 0; sub < R / 4; ++sub) {
        // slab entry k of this quarter = neighbour sub*256 + k of the round  <->  (j = 4 sub + k / 64, q = k % 64)
        if ((uint32_t)(sub * kLongChunk) < left_round) {   // block-uniform
            float *buf = slab[qc & 1u];
            ++qc;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 r = rows[4 * sub + jj];
                const int k = q + 64 * jj;
                buf[(4 * c + 0) * kLongStride + k] = r[0];
                buf[(4 * c + 1) * kLongStride + k] = r[1];
                buf[(4 * c + 2) * kLongStride + k] = r[2];
                buf[(4 * c + 3) * kLongStride + k] = r[3];
            }
            __syncthreads();
            if (tid < 16) {
                const uint32_t left = left_round - sub * kLongChunk;
                const uint32_t cnt = left < (uint32_t)kLongChunk ? left : (uint32_t)kLongChunk;
                const float *col = buf + tid * kLongStride;
                uint32_t k = 0;
                for (; k + 64 <= cnt; k += 64) {   // 16 LDS reads in flight, then 64 ordered adds:
                    f32x4 t[16];                   // the add chain, not the LDS latency, paces the row
#pragma unroll
                    for (int j = 0; j < 16; ++j) t[j] = *reinterpret_cast<const f32x4 *>(&col[k + 4 * j]);
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        acc += t[j][0];
                        acc += t[j][1];
                        acc += t[j][2];
                        acc += t[j][3];
                    }
                }
                for (; k < cnt; ++k) acc += col[k];
            }
            // this buffer is rewritten two quarters later, after the barrier of the next
            // quarter, which the adders reach only when these reads are done
        }
    }
}

// Long rows under the filtered gather (see k_stage_f16<.., FILTER>): it is the add chain (~4 ns a neighbour) that paces a long
// row, not its fetches, so redirecting the known-zero targets to the pad row inside k_long_f16 gains nothing (measured: 0.85 ->
// 0.89 ms on R-MAT-22), and compacting them away inside it costs more registers and barriers than it saves (1.77 ms).  Instead a
// pass of its own in front of it writes every long row's targets OUTSIDE the set, in stored order, IN PLACE (row u's at
// keep_col[rowptr[u] ...], keep_cnt[u] of them): one workgroup per row, 1024 entries a trip, 4 ballots per wave and a 16-entry
// exchange through LDS for the order — nothing sequential but the trips of one row — and k_long_f16<.., FILTER> then walks the
// short list.  The pass reads the row's full list, or the list an EARLIER stage of this forward left (short_col / short_cnt) when
// the device found this input to keep that stage's set all zero (*short_bad == 0, k_filter_mark).
__global__ __launch_bounds__(256) void k_long_lists(GraphDev g, uint32_t row_lo, uint32_t row_hi, const uint32_t *__restrict__ list,
                                                    uint32_t min_deg, uint32_t max_deg) {
    __shared__ uint32_t rc[2][16];
    const uint32_t u = list[blockIdx.x];
    if (u < row_lo || u >= row_hi) return;   // block-uniform, and the same rows as k_long_f16 takes
    const uint32_t rs = g.rowptr[u], deg = g.rowptr[u + 1] - rs;
    if (deg >= max_deg || deg < min_deg || !filter_worth(g)) return;
    const bool shortl = g.short_col != nullptr && *g.short_bad == 0u;
    const uint32_t *__restrict__ src = (shortl ? g.short_col : g.col) + rs;
    const uint32_t len = shortl ? min(g.short_cnt[u], deg) : deg;   // (min, and the clamp below: a list is never longer than its row, an id never beyond the pad row — whatever the words hold)
    uint32_t *__restrict__ dst = g.keep_col + rs;
    const uint32_t *__restrict__ zb = g.zero_bits;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t kept = 0;
    for (uint32_t base = 0, trip = 0; base < len; base += 1024, ++trip) {   // entry base + 256 i + tid: run (i, wave) is the (4 i + wave)-th
        uint32_t v[4], wd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t k = base + 256 * i + tid;
            v[i] = k < len ? min(src[k], g.n) : g.n;   // (past the end: the pad row, whose bit is 0 — masked below)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) wd[i] = zb[v[i] >> 5];
        unsigned long long m[4];
        uint32_t mycnt = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool take = base + 256 * i + tid < len && !(wd[i] >> (v[i] & 31u) & 1u);
            m[i] = __ballot(take);
            mycnt = (lane == i) ? (uint32_t)__popcll(m[i]) : mycnt;
        }
        uint32_t(&cnt)[16] = rc[trip & 1u];
        if (lane < 4) cnt[4 * lane + w] = mycnt;
        __syncthreads();   // (rc alternates: a wave writes cnt of trip t + 2 only after this barrier of trip t + 1, behind every read of trip t)
        uint32_t before[4], run = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((r & 3) == w) before[r >> 2] = run;
            run += cnt[r];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (m[i] >> lane & 1ull) dst[kept + before[i] + (uint32_t)__popcll(m[i] & below)] = v[i];
        kept += run;
    }
    if (tid == 0) g.keep_cnt[u] = kept;
}

// FILTER: the row's list is the one k_long_lists wrote for this call (when the set was worth it: the same test there and here).
// (Most of those lists are short — R-MAT-22: 35 K rows, a few hundred targets left in most: the rounds of a quarter, kLongR,
// were first measured here: first forward 4.03 -> 3.73 ms.)
template <int N1, int N2, int N3, bool SIGMOID, bool FILTER = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_long_f16(
        GraphDev g, float ws, const float4 *__restrict__ fin, float *__restrict__ fout,
        float *__restrict__ logits, const float *__restrict__ P, uint32_t row_lo, uint32_t row_hi,
        const uint32_t *__restrict__ list, uint32_t min_deg, uint32_t max_deg) {
    __shared__ __attribute__((aligned(16))) float slab[2][kLongStride * 16];
    const uint32_t u = list[blockIdx.x];
    if (u < row_lo || u >= row_hi) return;   // block-uniform
    const int tid = threadIdx.x, q = tid >> 2, c = tid & 3;
    const uint32_t deg = g.rowptr[u + 1] - g.rowptr[u];
    if (deg >= max_deg) return;              // a giant row: the k_giant_* kernels have it
    // the entries to gather: the row's whole list, or its pruned one (see k_prune_*; gdeg may be 0)
    const bool pruned = g.prune_bad != nullptr && *g.prune_bad == 0u;
    const bool lists = FILTER && !pruned && g.keep_col != nullptr && deg >= min_deg && filter_worth(g);
    const uint32_t *__restrict__ gcol = pruned ? g.pcol : (lists ? g.keep_col : g.col);
    const uint32_t rs = pruned ? g.prp[u] : g.rowptr[u];
    const uint32_t re = pruned ? g.prp[u + 1] : (lists ? rs + min(g.keep_cnt[u], deg) : g.rowptr[u + 1]);
    const uint32_t gdeg = re - rs, last = gdeg ? re - 1 : rs;   // (col arrays are padded: [rs] is readable)
    // listed for another stage's threshold (or, classed by the entries it has left, short enough): a tile kernel has it here
    if ((pruned && g.prune_eff) ? gdeg < g.eff_thresh : deg < min_deg) return;
    const uint32_t zrow = g.n;
    // Two register sets (A, B) alternate: while one round drains through the slab, the
    // 1024 row fetches of the next round and the column indices of the round after it are
    // in flight, so neither memory latency sits on the sequential add chain.
    constexpr int R = kLongR;
    constexpr uint32_t kRound = 64u * R;
    const uint32_t nrounds = (gdeg + kRound - 1) / kRound;
    uint32_t idx[R], qc = 0;
    f32x4 ra[R], rb[R];
    const f32x4 *__restrict__ fv = reinterpret_cast<const f32x4 *>(fin);
#define GNNVC_FETCH_IDX(rd_)                                              \
    _Pragma("unroll") for (int j = 0; j < R; ++j) {                        \
        const uint32_t e_ = rs + (rd_) * kRound + q + 64 * j;             \
        idx[j] = gcol[e_ < re ? e_ : last]; /* raw: not consumed until the next round */ \
    }
#define GNNVC_FETCH_ROWS(dst_, rd_)                                       \
    _Pragma("unroll") for (int j = 0; j < R; ++j) {                        \
        const uint32_t e_ = rs + (rd_) * kRound + q + 64 * j;             \
        dst_[j] = fv[(size_t)((e_ < re) ? (FILTER ? min(idx[j], zrow) : idx[j]) : zrow) * 4 + c]; \
    }
    float acc = 0.0f;   // threads 0..15: feature column tid
    // Every fetch below is unconditional (past the row's end the indices clamp to its last
    // entry and the rows to the zero row): only then can the compiler prove how many younger
    // loads are outstanding and wait for exactly the round being drained (vmcnt is in-order).
    {
        GNNVC_FETCH_IDX(0u)
        GNNVC_FETCH_ROWS(ra, 0u)
        GNNVC_FETCH_IDX(1u)
        for (uint32_t rd = 0; rd < nrounds; rd += 2) {
            GNNVC_FETCH_ROWS(rb, rd + 1)
            GNNVC_FETCH_IDX(rd + 2)
            long_drain_round<R>(ra, slab, gdeg - rd * kRound, tid, q, c, acc, qc);
            if (rd + 1 >= nrounds) break;
            GNNVC_FETCH_ROWS(ra, rd + 2)
            GNNVC_FETCH_IDX(rd + 3)
            long_drain_round<R>(rb, slab, gdeg - (rd + 1) * kRound, tid, q, c, acc, qc);
        }
    }
#undef GNNVC_FETCH_IDX
#undef GNNVC_FETCH_ROWS
    if (tid >= 64) return;   // the first wave holds the sums (lanes 0..15) and runs the row's dense layers
    wave_tail_f16<N1, N2, N3, SIGMOID>(g, ws, reinterpret_cast<const float *>(fin), fout, logits, P, u, deg, acc, tid);
}

template <int N1, int N2, int N3>
__global__ __launch_bounds__(256) void k_long_f1(
        GraphDev g, float ws, const float *__restrict__ xin, float *__restrict__ fout,
        const float *__restrict__ P, uint32_t row_lo, uint32_t row_hi,
        const uint32_t *__restrict__ list, uint32_t min_deg, uint32_t max_deg) {
    // a round = 256 R neighbours (R per thread); values of round r + 1 and column indices of
    // round r + 2 are in flight while thread 0 adds round r from the LDS slab in CSR order
    constexpr int R = 4;   // (8 until round 3: rounds of 2048; 4: R-MAT-22 2.46 -> 2.44 ms, power-law 0.856 -> 0.834; 2: the same)
    constexpr uint32_t kRound = 256 * R;
    __shared__ __attribute__((aligned(16))) float slab[2][kRound];
    const uint32_t u = list[blockIdx.x];
    if (u < row_lo || u >= row_hi) return;
    const int tid = threadIdx.x;
    const uint32_t rs = g.rowptr[u], re = g.rowptr[u + 1];
    const uint32_t deg = re - rs;
    if (deg < min_deg || deg >= max_deg) return;   // (giant rows: k_giant_*)
    const uint32_t nrounds = (deg + kRound - 1) / kRound;
    uint32_t idx[R];
    float v[R];
#define GNNVC_FETCH_IDX1(rd_)                                             \
    _Pragma("unroll") for (int j = 0; j < R; ++j) {                        \
        const uint32_t e_ = rs + (rd_) * kRound + tid + 256 * j;          \
        idx[j] = g.col[e_ < re ? e_ : re - 1];                            \
    }
#define GNNVC_FETCH_VAL1(rd_)                                             \
    _Pragma("unroll") for (int j = 0; j < R; ++j) v[j] = xin[idx[j]]; /* raw: masked when written to the slab */
    float agg = 0.0f;
    GNNVC_FETCH_IDX1(0u)
    GNNVC_FETCH_VAL1(0u)
    GNNVC_FETCH_IDX1(1u)
    for (uint32_t rd = 0; rd < nrounds; ++rd) {
        float *buf = slab[rd & 1];
#pragma unroll
        for (int j = 0; j < R; ++j) buf[tid + 256 * j] = v[j];   // entries past the row's end are never added
        __syncthreads();
        GNNVC_FETCH_VAL1(rd + 1)      // unconditional (clamped), see k_long_f16
        GNNVC_FETCH_IDX1(rd + 2)
        if (tid == 0) {
            const uint32_t left = deg - rd * kRound;
            const uint32_t cnt = left < kRound ? left : kRound;
            uint32_t k = 0;
            for (; k + 64 <= cnt; k += 64) {
                float t[64];
#pragma unroll
                for (int j = 0; j < 64; ++j) t[j] = buf[k + j];
#pragma unroll
                for (int j = 0; j < 64; ++j) agg += t[j];
            }
            for (; k < cnt; ++k) agg += buf[k];
        }
        // slab[rd & 1] is rewritten in round rd + 2, behind the barrier of round rd + 1
    }
#undef GNNVC_FETCH_IDX1
#undef GNNVC_FETCH_VAL1
    if (tid >= 64) return;   // (thread 0 holds the sum)
    wave_tail_f1<N1, N2, N3>(g, ws, xin, fout, P, u, deg, agg, tid);
}

// ------------------------------------------------------------------ giant rows
// k_long_* still walk a row with ONE add chain: ~4 ns per neighbour, so a hub of 650 K neighbours (R-MAT
// scale 22) holds its stage for 2.6 ms whatever the other 255 CUs do.  Rows of degree >= the giant threshold
// take another route that keeps the chain's bits (exact_sum.h: within a binade an fp32 add depends on the
// accumulator only through the parity of its significand, and such maps compose):
//   k_giant_gather*  every CU: the row's neighbour values are gathered once, 256 neighbours per workgroup,
//                    and written COLUMN-major into a slab — one contiguous stream of floats per (row, feature
//                    column), in stored order (stream c of giant row i at slab + off[i] + c * lpad);
//   k_giant_sum      one wave per stream: windows of 1024 addends, 16 consecutive ones per lane.  A lane folds
//                    its addends into a parity map, a 6-step wave scan composes the 64 maps, and if the window
//                    ends inside the accumulator's binade the new accumulator is m + D[m & 1].  Where it would
//                    carry (or a lane met a negative / non-finite value) the lanes in front are applied, that
//                    lane's 16 addends are added the plain way in fp32, and the window resumes behind it with
//                    the new binade.  A row crosses a binade about log2(degree) times, so nearly every window
//                    is one step: ~1 ns per neighbour and column, columns in parallel;
//   k_giant_dense    the rows' dense layers, one lane per row (as the tail of k_long_*).
// All three skip rows outside [row_lo, row_hi) (vertex-partitioned runs).  (Rounds 1 - 3 carried a tolerance mode beside this —
// lane-strided partial sums and a wave tree, "hub_mode" 1; with the exact path as fast as it is it bought nothing and is gone.)
constexpr int kGiantB = 16;                          // addends per lane and window
static_assert(kGiantB <= xsum::kMaxAppends, "a lane's appends must fit 32 bits before saturate()");
constexpr uint32_t kGiantWin = 64u * kGiantB;        // 1024: streams are padded to a multiple of this
constexpr uint32_t kGiantBlk = 256;                  // neighbours per gather workgroup
constexpr int kGiantRing = 4;                        // windows held in registers: three 4 KB loads in flight per wave.  (8 changes nothing:
                                                     // 0.39 ms either way for a 260 K-addend stream — the wave is bound by issuing its
                                                     // ~550 vector instructions per window, alone on its SIMD, not by the loads)

// meta[i] = {row, first CSR entry, degree, first gather block}, meta[n_giant].w = number of gather blocks
__device__ __forceinline__ uint32_t giant_of_block(const uint4 *__restrict__ meta, uint32_t n_giant, uint32_t b) {
    uint32_t lo = 0, hi = n_giant;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (meta[mid].w <= b) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void k_find_giant(GraphDev g, const uint32_t *__restrict__ list, uint32_t n_long, uint32_t thresh,
                             uint4 *__restrict__ meta, uint32_t *__restrict__ count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_long) return;
    const uint32_t u = list[i], rs = g.rowptr[u], deg = g.rowptr[u + 1] - rs;
    if (deg >= thresh) meta[atomicAdd(count, 1u)] = make_uint4(u, rs, deg, 0u);
}

__global__ __launch_bounds__(256) void k_giant_gather16(GraphDev g, const float4 *__restrict__ fin, float *__restrict__ slab,
                                                        const uint4 *__restrict__ meta, const unsigned long long *__restrict__ off,
                                                        uint32_t n_giant, uint32_t row_lo, uint32_t row_hi, uint32_t min_deg) {
    __shared__ __attribute__((aligned(16))) float tile[16][kGiantBlk + 4];
    const uint32_t i = giant_of_block(meta, n_giant, blockIdx.x);
    const uint4 mt = meta[i];
    if (mt.x < row_lo || mt.x >= row_hi || mt.z < min_deg) return;      // block-uniform (min_deg: this stage leaves shorter rows to k_long_*)
    // the row's entries: all of them, or (pruned adjacency, see k_prune_*) those whose target row may be non-zero — the
    // streams then are shorter, in the same slab region
    const bool pruned = g.prune_bad != nullptr && *g.prune_bad == 0u;
    const uint32_t *__restrict__ gcol = pruned ? g.pcol : g.col;
    const uint32_t first = pruned ? g.prp[mt.x] : mt.y, deg = pruned ? g.prp[mt.x + 1] - first : mt.z;
    const uint32_t j0 = (blockIdx.x - mt.w) * kGiantBlk;
    const uint32_t lpad = (deg + kGiantWin - 1) / kGiantWin * kGiantWin;
    if (j0 >= lpad) return;                            // (block-uniform; only a pruned row has such blocks)
    const int tid = threadIdx.x, q = tid >> 2, c = tid & 3;
    const f32x4 *__restrict__ fv = reinterpret_cast<const f32x4 *>(fin);
    uint32_t idx[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const uint32_t j = j0 + q + 64 * jj;
        idx[jj] = j < deg ? gcol[first + j] : g.n;     // past the row's end: the all-zero pad row
    }
    if (!pruned && filter_worth(g)) {                  // filtered gather (see k_stage_f16): known-zero rows are not fetched
        uint32_t wd[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) wd[jj] = g.zero_bits[idx[jj] >> 5];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) idx[jj] = (wd[jj] >> (idx[jj] & 31u) & 1u) ? g.n : idx[jj];
    }
    f32x4 r[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) r[jj] = fv[(size_t)idx[jj] * 4 + c];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int k = q + 64 * jj;
        tile[4 * c + 0][k] = r[jj][0];
        tile[4 * c + 1][k] = r[jj][1];
        tile[4 * c + 2][k] = r[jj][2];
        tile[4 * c + 3][k] = r[jj][3];
    }
    __syncthreads();
    float *dst = slab + off[i] + j0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int cc = 4 * p + (tid >> 6), k4 = (tid & 63) * 4;
        *reinterpret_cast<f32x4 *>(dst + (size_t)cc * lpad + k4) = *reinterpret_cast<const f32x4 *>(&tile[cc][k4]);
    }
}

__global__ __launch_bounds__(256) void k_giant_gather1(GraphDev g, const float *__restrict__ x, float *__restrict__ slab,
                                                       const uint4 *__restrict__ meta, const unsigned long long *__restrict__ off,
                                                       uint32_t n_giant, uint32_t row_lo, uint32_t row_hi) {
    const uint32_t i = giant_of_block(meta, n_giant, blockIdx.x);
    const uint4 mt = meta[i];
    if (mt.x < row_lo || mt.x >= row_hi) return;
    const uint32_t j = (blockIdx.x - mt.w) * kGiantBlk + threadIdx.x;
    slab[off[i] + j] = j < mt.z ? x[g.col[mt.y + j]] : 0.0f;
}

// one window of a stream: this lane's 16 addends are d[0..3]; valid are those with local index in [lo, hi)
__device__ __forceinline__ void giant_window(const f32x4 (&d)[kGiantB / 4], int lane, int hi, float &acc) {
    int lo = 0;   // per lane: local indices below lo are consumed
    for (int guard = 0; guard < 66; ++guard) {   // every pass consumes at least one lane
        uint32_t E = 0, m = 0;
        const bool ok = xsum::decode_acc(__float_as_uint(acc), E, m);
        xsum::Map run = {0u, 0u};
        bool bad = false;
        if (ok) {
#pragma unroll
            for (int i = 0; i < kGiantB; ++i) {
                const uint32_t vb = (i >= lo && i < hi) ? __float_as_uint(d[i >> 2][i & 3]) : 0u;
                xsum::append<true>(run, vb, E, bad);
            }
            xsum::saturate(run);
        }
        const unsigned long long badmask = __ballot(bad);
        if (!ok || __popcll(badmask) > 4) {
            // the integer route does not apply (negative or non-finite accumulator, or many such addends):
            // what is left of this window the plain way, in stored order
            for (int l = 0; l < 64; ++l) {
                const int llo = __builtin_amdgcn_readlane(lo, l), lhi = __builtin_amdgcn_readlane(hi, l);
#pragma unroll
                for (int i = 0; i < kGiantB; ++i) {
                    const float v = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d[i >> 2][i & 3]), l));
                    if (i >= llo && i < lhi) acc = acc + v;
                }
            }
            return;
        }
        // inclusive scan of the lanes' maps (earlier lanes first)
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            xsum::Map a;
            a.d0 = __shfl_up(run.d0, s);
            a.d1 = __shfl_up(run.d1, s);
            const xsum::Map cmb = xsum::compose(a, run);
            if (lane >= s) run = cmb;
        }
        const uint32_t D = (m & 1u) ? run.d1 : run.d0;
        const unsigned long long stop = __ballot(bad || m + D >= xsum::kCarry);
        if (stop == 0ull) {
            acc = __uint_as_float(xsum::encode_acc(E, m + __builtin_amdgcn_readlane(D, 63)));
            return;
        }
        const int L = __ffsll((long long)stop) - 1;   // uniform: first lane whose end is beyond the binade
        const uint32_t before = L ? __builtin_amdgcn_readlane(D, L - 1) : 0u;
        acc = __uint_as_float(xsum::encode_acc(E, m + before));
        const int llo = __builtin_amdgcn_readlane(lo, L), lhi = __builtin_amdgcn_readlane(hi, L);
#pragma unroll
        for (int i = 0; i < kGiantB; ++i) {
            const float v = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d[i >> 2][i & 3]), L));
            if (i >= llo && i < lhi) acc = acc + v;
        }
        if (L == 63) return;
        if (lane <= L) lo = kGiantB;
    }
}

// One stream on several waves.  The parity maps compose across waves as they do across lanes, given the accumulator's
// binade at the start of each piece.  Segments of kGiantSeg windows:
//   k_giant_segsum   a plain float sum per segment (any order): enough to ESTIMATE the running sum in front of a segment
//   k_giant_segmap   per segment k >= 1: the binade E of that estimate and the segment's composed map relative to E
//   k_giant_sum      walks the segments in order with the EXACT accumulator: where it sits in the segment's binade E and
//                    m + D[m & 1] < 2^24 (addends are >= 0 — a negative or non-finite one marks the segment — so no carry at
//                    the end means none inside), the segment is one step; otherwise (estimate off by a binade, the sum
//                    crosses into the next one — about log2(length) times per stream — or a marked segment) its windows are
//                    walked as before.  The result never depends on the estimate, only the time does (host emulation with
//                    estimates that are right, one off and random: tests/test_exact_sum_host.py).
// (round 4 tried segments of ONE window, with the maps fetched 64 at a time and, in a second form, every window prefetched: the
// walk of the power-law graph's 201 K-entry row went from 0.09 - 0.17 ms to 0.09 - 0.13 and to 0.10 - 0.19 ms — a wave pulling
// its whole stream is bound by its own memory-level parallelism, a wave fetching on demand by a round trip per binade crossing —
// while the segment kernels doubled; four windows per segment stay, the maps now arrive 64 at a time.)
constexpr uint32_t kGiantSeg = 4;   // windows per segment: 4096 addends

// (a block = kGiantSegWaves waves, one segment each: with one-window segments a block per segment would be a hundred thousand
// one-wave workgroups on the power-law graph)
constexpr uint32_t kGiantSegWaves = 4;
__host__ __device__ inline uint32_t giant_seg_blocks(uint32_t maxseg) { return (maxseg + kGiantSegWaves - 1) / kGiantSegWaves; }

__global__ __launch_bounds__(64 * kGiantSegWaves) void k_giant_segsum(const float *__restrict__ slab, const uint4 *__restrict__ meta,
                                                     const unsigned long long *__restrict__ off, uint32_t F, uint32_t maxseg,
                                                     float *__restrict__ segsum, uint32_t row_lo, uint32_t row_hi,
                                                     const uint32_t *__restrict__ prp, const uint32_t *__restrict__ prune_bad, uint32_t min_deg) {
    const uint32_t sb = giant_seg_blocks(maxseg);
    const uint32_t st = blockIdx.x / sb, sg = (blockIdx.x % sb) * kGiantSegWaves + (threadIdx.x >> 6), i = st / F, c = st % F;
    if (sg >= maxseg) return;
    const uint4 mt = meta[i];
    if (mt.x < row_lo || mt.x >= row_hi || mt.z < min_deg) return;
    const int lane = threadIdx.x & 63;
    const bool pruned = prune_bad != nullptr && *prune_bad == 0u;
    const uint32_t len = pruned ? prp[mt.x + 1] - prp[mt.x] : mt.z;
    const uint32_t lpad = (len + kGiantWin - 1) / kGiantWin * kGiantWin, nwin = lpad / kGiantWin;
    const uint32_t w0 = sg * kGiantSeg, w1 = min(nwin, w0 + kGiantSeg);
    if (w0 >= nwin) return;
    const f32x4 *__restrict__ src = reinterpret_cast<const f32x4 *>(slab + off[i] + (size_t)c * lpad) + lane * (kGiantB / 4);
    float part = 0.0f;
    for (uint32_t w = w0; w < w1; ++w) {   // (the streams are padded with zeros up to lpad)
#pragma unroll
        for (int k = 0; k < kGiantB / 4; ++k) {
            const f32x4 q = src[(size_t)w * (kGiantWin / 4) + k];
            part += (q[0] + q[1]) + (q[2] + q[3]);
        }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) part += __shfl_xor(part, s);
    if (lane == 0) segsum[(size_t)st * maxseg + sg] = part;
}

__global__ __launch_bounds__(64 * kGiantSegWaves) void k_giant_segmap(const float *__restrict__ slab, const uint4 *__restrict__ meta,
                                                     const unsigned long long *__restrict__ off, uint32_t F, uint32_t maxseg,
                                                     const float *__restrict__ segsum, uint4 *__restrict__ segmap, uint32_t row_lo,
                                                     uint32_t row_hi, const uint32_t *__restrict__ prp,
                                                     const uint32_t *__restrict__ prune_bad, uint32_t min_deg) {
    const uint32_t sb = giant_seg_blocks(maxseg);
    const uint32_t st = blockIdx.x / sb, sg = (blockIdx.x % sb) * kGiantSegWaves + (threadIdx.x >> 6), i = st / F, c = st % F;
    if (sg == 0 || sg >= maxseg) return;                  // (the first segment is always walked: nothing in front of it to estimate)
    const uint4 mt = meta[i];
    if (mt.x < row_lo || mt.x >= row_hi || mt.z < min_deg) return;
    const int lane = threadIdx.x & 63;
    const bool pruned = prune_bad != nullptr && *prune_bad == 0u;
    const uint32_t len = pruned ? prp[mt.x + 1] - prp[mt.x] : mt.z;
    const uint32_t lpad = (len + kGiantWin - 1) / kGiantWin * kGiantWin, nwin = lpad / kGiantWin;
    const uint32_t w0 = sg * kGiantSeg, w1 = min(nwin, w0 + kGiantSeg);
    if (w0 >= nwin) return;
    // the running sum in front of the segment, to a few parts in 10^5: its binade
    float pre = 0.0f;
    for (uint32_t j = lane; j < sg; j += 64) pre += segsum[(size_t)st * maxseg + j];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) pre += __shfl_xor(pre, s);
    const uint32_t pb = __float_as_uint(pre), pe = (pb >> 23) & 0xFFu;
    const uint32_t E = pe ? pe : 1u;
    bool bad = (pb >> 31) != 0u || pe == 255u;            // (then no accumulator will match: the segment is walked)
    const f32x4 *__restrict__ src = reinterpret_cast<const f32x4 *>(slab + off[i] + (size_t)c * lpad) + lane * (kGiantB / 4);
    xsum::Map total = {0u, 0u};
    for (uint32_t w = w0; w < w1; ++w) {
        f32x4 d[kGiantB / 4];
#pragma unroll
        for (int k = 0; k < kGiantB / 4; ++k) d[k] = src[(size_t)w * (kGiantWin / 4) + k];
        const long long left = (long long)len - (long long)w * kGiantWin - (long long)lane * kGiantB;
        const int hi = left <= 0 ? 0 : (left >= kGiantB ? kGiantB : (int)left);
        xsum::Map run = {0u, 0u};
#pragma unroll
        for (int t = 0; t < kGiantB; ++t) {
            const uint32_t vb = t < hi ? __float_as_uint(d[t >> 2][t & 3]) : 0u;
            xsum::append<true>(run, vb, E, bad);
        }
        xsum::saturate(run);
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {                // inclusive scan of the lanes' maps (earlier lanes first)
            xsum::Map a;
            a.d0 = __shfl_up(run.d0, s);
            a.d1 = __shfl_up(run.d1, s);
            const xsum::Map cmb = xsum::compose(a, run);
            if (lane >= s) run = cmb;
        }
        xsum::Map wt;
        wt.d0 = __builtin_amdgcn_readlane(run.d0, 63);
        wt.d1 = __builtin_amdgcn_readlane(run.d1, 63);
        total = xsum::compose(total, wt);
    }
    const bool any_bad = __ballot(bad) != 0ull;
    if (lane == 0) segmap[(size_t)st * maxseg + sg] = make_uint4(E, total.d0, total.d1, any_bad ? 1u : 0u);
}

__global__ __launch_bounds__(64) void k_giant_sum(const float *__restrict__ slab, const uint4 *__restrict__ meta,
                                                  const unsigned long long *__restrict__ off, uint32_t F, float *__restrict__ agg,
                                                  uint32_t row_lo, uint32_t row_hi, const uint32_t *__restrict__ prp,
                                                  const uint32_t *__restrict__ prune_bad, const uint4 *__restrict__ segmap,
                                                  uint32_t maxseg, uint32_t min_deg) {
    const uint32_t i = blockIdx.x / F, c = blockIdx.x % F;
    const uint4 mt = meta[i];
    if (mt.x < row_lo || mt.x >= row_hi || mt.z < min_deg) return;
    const int lane = threadIdx.x;
    const bool pruned = prune_bad != nullptr && *prune_bad == 0u;   // (the gather kernel wrote the shorter streams)
    const uint32_t len = pruned ? prp[mt.x + 1] - prp[mt.x] : mt.z;
    const uint32_t lpad = (len + kGiantWin - 1) / kGiantWin * kGiantWin;
    const uint32_t nwin = lpad / kGiantWin;
    if (nwin == 0) {   // nothing left of the row: a sum of no addends
        if (lane == 0) agg[(size_t)i * 16 + c] = 0.0f;
        return;
    }
    const f32x4 *__restrict__ src = reinterpret_cast<const f32x4 *>(slab + off[i] + (size_t)c * lpad) + lane * (kGiantB / 4);
    f32x4 buf[kGiantRing][kGiantB / 4];
    // loads are unconditional (clamped to the last window) so that the in-order vmcnt waits cover exactly the window
    // being consumed while the next kGiantRing - 1 stay in flight
#define GNNVC_GIANT_LOAD(slot_, w_)                                                        \
    {                                                                                      \
        const uint32_t ww_ = (w_) < nwin ? (w_) : nwin - 1;                                \
        const f32x4 *p_ = src + (size_t)ww_ * (kGiantWin / 4);                             \
        _Pragma("unroll") for (int k_ = 0; k_ < kGiantB / 4; ++k_) buf[slot_][k_] = p_[k_]; \
    }
#define GNNVC_GIANT_USE(slot_, w_)                                                         \
    {                                                                                      \
        const long long left_ = (long long)len - (long long)(w_) * kGiantWin - (long long)lane * kGiantB; \
        const int hi_ = left_ <= 0 ? 0 : (left_ >= kGiantB ? kGiantB : (int)left_);         \
        giant_window(buf[slot_], lane, hi_, acc);                                          \
    }
    float acc = 0.0f;
    const bool segmented = segmap != nullptr;
    const uint32_t segw = segmented ? kGiantSeg : nwin;      // (no maps: the whole stream is one walk)
    const uint32_t nseg = (nwin + segw - 1) / segw;
    // lane j: the map of segment (sg & ~63) + j — fetched 64 at a time, ahead of their use (round 4: a dependent fetch per
    // segment was a third of the walk's time on the power-law graph's 201 K-entry row)
    uint4 maps = make_uint4(0u, 0u, 0u, 1u);
    for (uint32_t w0 = 0, sg = 0; w0 < nwin; w0 += segw, ++sg) {
        const uint32_t w1 = min(nwin, w0 + segw);
        if (segmented && (sg & 63u) == 0u) {
            const uint32_t j = sg + (uint32_t)lane;
            maps = (j >= 1u && j < nseg) ? segmap[(size_t)blockIdx.x * maxseg + j] : make_uint4(0u, 0u, 0u, 1u);
        }
        if (segmented && sg > 0) {   // the segment in one step, if its map was made for the binade the accumulator is in
            const int src_lane = (int)(sg & 63u);
            const uint4 mp = make_uint4((uint32_t)__builtin_amdgcn_readlane((int)maps.x, src_lane), (uint32_t)__builtin_amdgcn_readlane((int)maps.y, src_lane),
                                        (uint32_t)__builtin_amdgcn_readlane((int)maps.z, src_lane), (uint32_t)__builtin_amdgcn_readlane((int)maps.w, src_lane));
            uint32_t E = 0, m = 0;
            if (mp.w == 0u && xsum::decode_acc(__float_as_uint(acc), E, m) && E == mp.x) {
                const uint32_t D = (m & 1u) ? mp.z : mp.y;
                if (m + D < xsum::kCarry) {
                    acc = __uint_as_float(xsum::encode_acc(E, m + D));
                    continue;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < kGiantRing - 1; ++s) GNNVC_GIANT_LOAD(s, w0 + (uint32_t)s)
        for (uint32_t w = w0; w < w1; w += kGiantRing) {
#pragma unroll
            for (int j = 0; j < kGiantRing; ++j) {
                if (w + j >= w1) break;
                GNNVC_GIANT_LOAD((j + kGiantRing - 1) % kGiantRing, w + j + kGiantRing - 1)
                GNNVC_GIANT_USE(j, w + j)
            }
        }
    }
#undef GNNVC_GIANT_LOAD
#undef GNNVC_GIANT_USE
    if (lane == 0) agg[(size_t)i * 16 + c] = acc;
}

// VARIANT as stage_variant(): 0 = F 1 -> 16 features, 1 = F 16 -> 16 features, 2 = F 16 -> sigmoid
template <int VARIANT>
__global__ __launch_bounds__(64) void k_giant_dense(GraphDev g, float ws, const float *__restrict__ in, float *__restrict__ out,
                                                    float *__restrict__ logits, const float *__restrict__ P,
                                                    const uint4 *__restrict__ meta, uint32_t n_giant,
                                                    const float *__restrict__ agg, uint32_t row_lo, uint32_t row_hi, uint32_t min_deg) {
    const uint32_t i = blockIdx.x;                       // one wave per giant row (wave_tail)
    const int lane = threadIdx.x;
    if (i >= n_giant) return;
    const uint4 mt = meta[i];
    if (mt.x < row_lo || mt.x >= row_hi || mt.z < min_deg) return;
    if constexpr (VARIANT == 0) {
        wave_tail_f1<32, 32, 16>(g, ws, in, out, P, mt.x, mt.z, agg[(size_t)i * 16], lane);
    } else {
        const float a = agg[(size_t)i * 16 + (lane & 15)];
        if constexpr (VARIANT == 1)
            wave_tail_f16<32, 32, 16, false>(g, ws, in, out, nullptr, P, mt.x, mt.z, a, lane);
        else
            wave_tail_f16<32, 16, 1, true>(g, ws, in, out, logits, P, mt.x, mt.z, a, lane);
    }
}

// the chain's sums of explicit streams (gnnvc_stream_sum: the k_giant_sum path on caller data)
__global__ void k_stream_meta(uint4 *meta, unsigned long long *off, uint32_t streams, uint32_t len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > streams) return;
    const uint32_t lpad = (len + kGiantWin - 1) / kGiantWin * kGiantWin;
    meta[i] = make_uint4(0u, 0u, len, 0u);
    if (i < streams) off[i] = (unsigned long long)i * lpad;
}

// ------------------------------------------------------------------ the next graph derived from the resident one (f-1)
// Between two predict calls of the reference's driver the graph only SHRINKS, apart from the vertices its folds
// create (include/reduction_graph.hpp:335-398: a fold appends a vertex with the largest id and puts it at the END
// of its neighbours' lists; removals drop entries; relable_graph :537-587 renumbers the survivors in order).  So
// the next CSR is: for a surviving row, its old entries that survive, in order, renumbered — followed by a short
// TAIL of new-vertex ids; for a new vertex, a list that is all tail.  Given old_row[] (new vertex -> its row in the
// resident graph, or none) the device derives everything but the tails from the CSR it already holds.
constexpr uint32_t kNoVertex = 0xFFFFFFFFu;

__global__ void k_derive_map(const uint32_t *__restrict__ old_row, uint32_t n_new, uint32_t n_old, uint32_t *__restrict__ new_of,
                             uint32_t *__restrict__ bad) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_new) return;
    const uint32_t o = old_row[u];
    if (o == kNoVertex) return;
    if (o >= n_old) { atomicOr(bad, 1u); return; }
    if (atomicExch(&new_of[o], u) != kNoVertex) atomicOr(bad, 2u);   // two new vertices claim the same old row
}

// tail[u] = (degree of u in the new graph) - (old entries of its old row that survive)
__global__ void k_derive_tails(GraphDev g, const uint32_t *__restrict__ old_row, const uint32_t *__restrict__ new_of,
                               const uint32_t *__restrict__ rowptr_new, uint32_t n_new, uint32_t *__restrict__ tail,
                               uint32_t *__restrict__ bad) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_new) return;
    const uint32_t o = old_row[u];
    uint32_t kept = 0;
    if (o != kNoVertex)
        for (uint32_t e = g.rowptr[o]; e < g.rowptr[o + 1]; ++e) kept += new_of[g.col[e]] != kNoVertex ? 1u : 0u;
    const uint32_t deg = rowptr_new[u + 1] - rowptr_new[u];
    if (rowptr_new[u + 1] < rowptr_new[u] || kept > deg) { atomicOr(bad, 4u); tail[u] = 0; return; }
    tail[u] = deg - kept;
}

__global__ void k_derive_fill(GraphDev g, const uint32_t *__restrict__ old_row, const uint32_t *__restrict__ new_of,
                              const uint32_t *__restrict__ rowptr_new, const uint32_t *__restrict__ tail_ptr,
                              const uint32_t *__restrict__ tail_cols, uint32_t n_new, uint32_t *__restrict__ col_new) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_new) return;
    const uint32_t o = old_row[u];
    uint32_t pos = rowptr_new[u];
    if (o != kNoVertex)
        for (uint32_t e = g.rowptr[o]; e < g.rowptr[o + 1]; ++e) {
            const uint32_t c = new_of[g.col[e]];
            if (c != kNoVertex) col_new[pos++] = c;
        }
    for (uint32_t t = tail_ptr[u]; t < tail_ptr[u + 1]; ++t) col_new[pos++] = tail_cols[t];
}

// FNV-1a over a row's column ids in stored order: lets a caller compare the resident graph with its own lists
__global__ void k_row_hashes(GraphDev g, unsigned long long *__restrict__ out) {
    const uint32_t u = g.lo() + blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= g.hi()) return;
    unsigned long long h = 1469598103934665603ull;
    for (uint32_t e = g.rowptr[u]; e < g.rowptr[u + 1]; ++e) h = (h ^ (unsigned long long)g.col[e]) * 1099511628211ull;
    out[u - g.lo()] = h;
}

// ---- LDS-table plan of the F = 1 stage ---------------------------------------------------------
// The F = 1 gather reads 4 bytes per neighbour and is bound by line fills (one 128-byte line per
// neighbour, from HBM or, column-blocked, from L2).  When the input really is x[v] = (float)W(v)/ws
// with W(v) <= 255 — the reference's driver (src/GNN_VC.cpp:189-191) on the usual weight ranges — the
// table of neighbour values is one BYTE per vertex, and an 81920-vertex slice of it fits in LDS beside
// the partial sums of ~19.5 K rows.  Rows are cut into chunks (one 1024-thread workgroup each) of 16
// SLICES (one per wave, sums in the wave's part of LDS), columns into blocks of 81920; the CSR entries of a
// slice are regrouped by block (a row's entries of one block stay adjacent and in order, every (slice, block)
// segment starts at a multiple of 4) and a workgroup STEP is one block with up to 256 entries per slice.
// Per step the workgroup stages the block's byte slice in LDS and every wave folds its own entries into its
// own rows (k_lt_agg).  Blocks ascend and a row's entries of one block are folded in order by one wave, so
// every row is still summed in CSR order: same bits as the plain gather.  What bounds it: every workgroup
// streams the whole byte table once per chunk (512 chunks x 10 MB on the metric graph) from L2 into LDS.
constexpr uint32_t kLtStep = 2048;        // default entries per step of the plan builder

// The table of THIS forward, made from its input alone: entry v = the integer k with x[v] == (float)k / ws bit for bit — the
// expression the kernel's values are made of; bad |= 1 if some x[v] is no such value (the plan then steps aside for this
// forward).  For the reference's driver x[v] = (float)W(v) / ws (src/GNN_VC.cpp:189-191): k = W(v).  Needs no vertex weights,
// so a rank that holds a SLICE of the graph (and the replicated x) can use the plan too.
// BITS (round 4): how wide an entry is — 8 (k <= 255: a byte per vertex), 10 (k <= 1023: three to a 32-bit word) or 16
// (k <= 65535) — chosen per graph from its largest weight, so that weights beyond a byte (WEIGHT_SCALE is any u32,
// include/gnn_inference.hpp:25; folds create new ones, include/reduction_graph.hpp:394-396) keep a table instead of falling
// back to the column-blocked plan.  The 81 920 bytes of LDS a column block takes then hold 81 920 / 61 440 / 40 960 vertices.
__host__ __device__ constexpr uint32_t lt_block_cols(uint32_t bits) { return bits == 8 ? 81920u : (bits == 10 ? 61440u : 40960u); }
__host__ __device__ constexpr uint32_t lt_piece_cols(uint32_t bits) { return bits == 8 ? 16u : (bits == 10 ? 12u : 8u); }   // vertices per 16 bytes
__host__ __device__ constexpr uint32_t lt_kmax(uint32_t bits) { return bits == 8 ? 255u : (bits == 10 ? 1023u : 65535u); }
__host__ __device__ inline size_t lt_table_bytes(uint32_t bits, size_t n) { return (n + lt_piece_cols(bits) - 1) / lt_piece_cols(bits) * 16 + 64; }

// x -> k (0 and a miss if x is no k / ws with k <= KMAX).  t = x * ws = k (1 + eps), |eps| <= 2^-23: rounds to k for k <= 65535
template <uint32_t KMAX>
__device__ __forceinline__ uint32_t lt_k_of(float xv, float ws, bool &miss) {
    const float t = xv * ws;
    const uint32_t k = (t >= 0.0f && t < (float)KMAX + 0.5f) ? (uint32_t)(t + 0.5f) : 0u;
    miss |= __float_as_uint((float)k / ws) != __float_as_uint(xv);
    return k;
}

template <int BITS>
__global__ __launch_bounds__(256) void k_lt_bytes_x(const float *__restrict__ x, float ws, uint32_t n, uint8_t *__restrict__ wb,
                                                    uint32_t *bad) {
    bool miss = false;
    constexpr uint32_t KMAX = lt_kmax(BITS);
    if constexpr (BITS == 8) {
        // four vertices per thread: one 16-byte load, one 4-byte store (an x that is not 16-byte aligned: one vertex per thread)
        const uint32_t quads = (reinterpret_cast<uintptr_t>(x) & 15u) ? 0u : n / 4u;
        for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x) {
            const float4 v = reinterpret_cast<const float4 *>(x)[q];
            const float f[4] = {v.x, v.y, v.z, v.w};
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) packed |= lt_k_of<KMAX>(f[i], ws, miss) << (8 * i);
            reinterpret_cast<uint32_t *>(wb)[q] = packed;
        }
        for (uint32_t v = quads * 4u + blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)   // the tail (or all of it)
            wb[v] = (uint8_t)lt_k_of<KMAX>(x[v], ws, miss);
    } else if constexpr (BITS == 16) {
        const uint32_t pairs = (n + 1u) / 2u;   // one 32-bit word = two vertices
        for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < pairs; q += gridDim.x * blockDim.x) {
            const uint32_t k0 = lt_k_of<KMAX>(x[2 * q], ws, miss);
            const uint32_t k1 = 2 * q + 1 < n ? lt_k_of<KMAX>(x[2 * q + 1], ws, miss) : 0u;
            reinterpret_cast<uint32_t *>(wb)[q] = k0 | (k1 << 16);
        }
    } else {
        const uint32_t words = (n + 2u) / 3u;   // one 32-bit word = three vertices of ten bits
        for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < words; q += gridDim.x * blockDim.x) {
            uint32_t packed = 0;
#pragma unroll
            for (uint32_t i = 0; i < 3; ++i)
                if (3 * q + i < n) packed |= lt_k_of<KMAX>(x[3 * q + i], ws, miss) << (10 * i);
            reinterpret_cast<uint32_t *>(wb)[q] = packed;
        }
    }
    if (__any(miss) && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
}

// *wmax = the largest of w[0 .. n) (how wide the plan's table entries have to be)
__global__ __launch_bounds__(256) void k_lt_wmax(const uint32_t *__restrict__ w, uint32_t n, uint32_t *__restrict__ wmax) {
    uint32_t m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = max(m, w[i]);
#pragma unroll
    for (int off = 32; off; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(wmax, m);
}

// A plan over rows that are NOT consecutive (skewed graphs, compact-table plan): chunk c holds the rows
// rowmap[c * rows_per_chunk ..] (0xFFFFFFFF = empty slot), its regrouped entries start at first[c] (+ the padding
// slack), and column blocks may have any widths: block b = columns [bstart[b], bstart[b + 1]).  All null: consecutive
// rows, CSR offsets, blocks of block_cols columns.
// (struct PlanMap: gnnvc_kernels.h)
__device__ __forceinline__ uint32_t lt_row(const PlanMap &pm, uint32_t c, uint32_t rows_per_chunk, uint32_t r0, uint32_t i) {
    return pm.rowmap ? pm.rowmap[(size_t)c * rows_per_chunk + i] : r0 + i;
}
// block of column `col`, searched from block `from` on (a row's columns ascend)
__device__ __forceinline__ uint32_t lt_block(const PlanMap &pm, uint32_t col, uint32_t block_cols, uint32_t nblocks, uint32_t from) {
    if (!pm.bstart) return col / block_cols;
    if (pm.coarse) {                                // the block of the column's 256-granule, then forward (blocks are >= 256 wide but for remainders)
        uint32_t b = pm.coarse[col >> 8];
        while (b + 1 < nblocks && col >= pm.bstart[b + 1]) ++b;
        return b;
    }
    if (col < pm.bstart[from]) from = 0;            // (unsorted row: found anyway, flagged by the caller)
    if (col < pm.bstart[from + 1]) return from;
    uint32_t lo = from + 1, hi = nblocks - 1;       // last block whose start is <= col
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (pm.bstart[mid] <= col) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ uint32_t lt_block_start(const PlanMap &pm, uint32_t b, uint32_t block_cols) {
    return pm.bstart ? pm.bstart[b] : b * block_cols;
}

// entries of chunk `c` per column block -> seg_cnt[c * nblocks + b]; bad |= 2 if a row's blocks are
// not ascending (unsorted adjacency: the plan would change the order of its sum)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// whole-wave shifts by one lane (DPP wave_shl / wave_shr, zero shifted in)
__device__ __forceinline__ uint32_t lane_next(uint32_t x) {   // lane i <- lane i + 1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t lane_prev(uint32_t x) {   // lane i <- lane i - 1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ float lane_next(float v) { return __uint_as_float(lane_next(__float_as_uint(v))); }

constexpr uint32_t kLtLongRow = 128;    // plan builders: rows of at least this many entries are walked by a whole wave
constexpr uint32_t kLtLongCap = 1280;   // >= the rows of a slice (kLtwSliceRows, kC4SliceRows)
__device__ __forceinline__ unsigned long long lanes_upto(uint32_t lane) {   // bits 0 .. lane
    return lane >= 63u ? ~0ull : ((1ull << (lane + 1u)) - 1ull);
}

// Flat walk of the plan builders (consecutive rows, uniform blocks, at most kLtFlatRows rows a chunk): rp[] = the chunk's
// CSR offsets in LDS.
constexpr uint32_t kLtFlatRows = 2048;
constexpr uint32_t kLtTrips = 4;
// first row of wave w's share: the first row that starts at or after w / nwaves of the chunk's entries
__device__ __forceinline__ uint32_t lt_wave_cut(const uint32_t *rp, uint32_t nr, uint32_t w, uint32_t nwaves) {
    if (w >= nwaves) return nr;
    const uint32_t want = rp[0] + (uint32_t)((uint64_t)(rp[nr] - rp[0]) * w / nwaves);
    uint32_t lo = 0, hi = nr;                          // first i with rp[i] >= want
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rp[mid] >= want) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}
// the row (index in the chunk) that holds entry e: the last i in [lo, hi] with rp[i] <= e
__device__ __forceinline__ uint32_t lt_row_of(const uint32_t *rp, uint32_t lo, uint32_t hi, uint32_t e) {
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (rp[mid] <= e) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
// col / d for d >= 2 with magic = floor(2^32 / d): the estimate is the quotient or one less
__device__ __forceinline__ uint32_t lt_div(uint32_t x, uint32_t d, uint32_t magic) {
    if (d < 2u) return x;
    const uint32_t q = __umulhi(x, magic);
    return q + (x - q * d >= d ? 1u : 0u);
}

__global__ __launch_bounds__(1024) void k_lt_count(GraphDev g, uint32_t rows_per_chunk, uint32_t nblocks, uint32_t block_cols,
                                                   uint32_t *__restrict__ seg_cnt, uint32_t *bad, uint32_t row_base,
                                                   uint32_t row_end, PlanMap pm) {
    __shared__ uint32_t hist[4096];
    __shared__ uint32_t longrow[kLtLongCap];
    __shared__ uint32_t nlong;
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < nblocks; i += 1024) hist[i] = 0;
    if (tid == 0) nlong = 0;
    __syncthreads();
    const uint32_t r0 = min(row_end, row_base + c * rows_per_chunk), r1 = pm.rowmap ? r0 + rows_per_chunk : min(row_end, r0 + rows_per_chunk);
    bool unsorted = false;
    for (uint32_t i = tid; i < r1 - r0; i += 1024) {   // short rows: a thread each; long ones are listed for the waves
        const uint32_t u = lt_row(pm, c, rows_per_chunk, r0, i);
        if (u >= row_end) continue;
        const uint32_t rs = g.rowptr[u], re = g.rowptr[u + 1];
        if (re - rs >= kLtLongRow && i < kLtLongCap) {
            longrow[atomicAdd(&nlong, 1u)] = u;
            continue;
        }
        uint32_t prev = 0, run = 0;
        for (uint32_t e = rs; e < re; ++e) {
            const uint32_t b = lt_block(pm, g.col[e], block_cols, nblocks, prev);
            if (run && b != prev) {
                atomicAdd(&hist[prev], run);
                unsorted |= b < prev;
                run = 0;
            }
            prev = b;
            ++run;
        }
        if (run) atomicAdd(&hist[prev], run);
    }
    __syncthreads();
    for (uint32_t k = wave; k < nlong; k += 16) {      // a long row: 64 entries a trip, one LDS add per run of equal blocks
        const uint32_t u = longrow[k];
        const uint32_t rs = g.rowptr[u], re = g.rowptr[u + 1];
        uint32_t carry = 0;                            // block of the previous trip's last entry (blocks must not descend)
        for (uint32_t e0 = rs; e0 < re; e0 += 64) {
            const uint32_t e = e0 + lane;
            const bool in = e < re;
            const uint32_t b = in ? lt_block(pm, g.col[e], block_cols, nblocks, 0) : 0xFFFFFFFFu;
            uint32_t bp = lane_prev(b);
            if (lane == 0) bp = carry;
            unsorted |= in && b < bp;
            const bool head = in && (lane == 0 || b != bp);   // (a trip's first entry always starts a count of its own)
            const unsigned long long hm = __ballot(head);
            const uint32_t nin = (uint32_t)__popcll(__ballot(in));
            if (head) {
                const unsigned long long later = hm & ~lanes_upto(lane);
                const uint32_t nh = later ? (uint32_t)__builtin_ctzll(later) : nin;
                atomicAdd(&hist[b], nh - lane);
            }
            carry = __shfl(b, (int)(nin - 1u));
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nblocks; i += 1024) seg_cnt[(size_t)c * nblocks + i] = hist[i];
    if (__any(unsorted) && (tid & 63) == 0) atomicOr(bad, 2u);
}

// Where the regrouped entries of chunk c start.  slack == 0: at the chunk's CSR offset (the regrouped array is a
// permutation of col[]).  slack != 0 (compact-table plan: a lane reads 16 bytes = 4 entries at once): every
// (chunk, block) segment starts at a multiple of 4, and chunk c is shifted by c * slack, slack >= 3 * nblocks + 4
// covering the padding of all chunks before it.
__device__ __forceinline__ uint32_t lt_chunk_first(const GraphDev &g, uint32_t c, uint32_t rows_per_chunk, uint32_t row_base,
                                                   uint32_t row_end, uint32_t slack, const uint32_t *mapped_first = nullptr) {
    const uint32_t r0 = (uint32_t)min((uint64_t)row_end, (uint64_t)row_base + (uint64_t)c * rows_per_chunk);
    const uint32_t first = mapped_first ? mapped_first[c] : g.rowptr[r0];
    return slack ? (first + c * slack + 3u) & ~3u : first;
}

// One thread per chunk.  write == 0: steps[c] = number of steps of chunk c (padded to a multiple of 4).
// write != 0: the descriptors {block, first entry, count, 0} at step_ptr[c]...
__global__ __launch_bounds__(256) void k_lt_steps(GraphDev g, uint32_t rows_per_chunk, uint32_t nchunks, uint32_t nblocks,
                                                  const uint32_t *__restrict__ seg_cnt, const uint32_t *__restrict__ step_ptr,
                                                  uint32_t *__restrict__ step_count, uint4 *__restrict__ steps, int write,
                                                  uint32_t row_base, uint32_t row_end, uint32_t cap, uint32_t slack,
                                                  uint32_t block_cols, PlanMap pm) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const uint32_t *cnt = seg_cnt + (size_t)c * nblocks;
    uint32_t first = lt_chunk_first(g, c, rows_per_chunk, row_base, row_end, slack, pm.first);   // a chunk's entries are a CSR range (or its mapped one)
    uint32_t pos = write ? step_ptr[c] : 0, made = 0;
    for (uint32_t b = 0; b < nblocks; ++b) {
        uint32_t left = cnt[b];
        while (left) {
            const uint32_t take = left < cap ? left : cap;
            if (write) steps[pos + made] = make_uint4(b, first, take, lt_block_start(pm, b, block_cols));   // .w: the block's first column
            ++made;
            first += take;
            left -= take;
        }
        if (slack) first = (first + 3u) & ~3u;
    }
    const uint32_t padded = made ? (made + 3u) / 4u * 4u : 4u;
    if (write)
        for (; made < padded; ++made) steps[pos + made] = make_uint4(0u, 0u, 0u, 0u);
    else
        step_count[c] = padded;
}

// regroup the chunk's CSR entries by column block: entries[...] = row_local << 17 | col_local; a row's
// entries of one block are written as one adjacent run, in order (the order among rows is free)
__global__ __launch_bounds__(1024) void k_lt_scatter(GraphDev g, uint32_t rows_per_chunk, uint32_t nblocks, uint32_t block_cols,
                                                     uint32_t shift, const uint32_t *__restrict__ seg_cnt,
                                                     uint32_t *__restrict__ entries, uint32_t row_base, uint32_t row_end,
                                                     uint32_t slack, PlanMap pm) {
    __shared__ uint32_t cursor[4096];
    __shared__ uint32_t longrow[kLtLongCap];   // slot index in the slice
    __shared__ uint32_t nlong;
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r0 = min(row_end, row_base + c * rows_per_chunk), r1 = pm.rowmap ? r0 + rows_per_chunk : min(row_end, r0 + rows_per_chunk);
    if (tid == 0) {   // exclusive scan of <= 4096 counts: a few microseconds, once per graph
        uint32_t run = lt_chunk_first(g, c, rows_per_chunk, row_base, row_end, slack, pm.first);
        for (uint32_t b = 0; b < nblocks; ++b) {
            cursor[b] = run;
            run += seg_cnt[(size_t)c * nblocks + b];
            if (slack) run = (run + 3u) & ~3u;
        }
        nlong = 0;
    }
    __syncthreads();
    for (uint32_t i = tid; i < r1 - r0; i += 1024) {   // short rows: a thread each; long ones are listed for the waves
        const uint32_t u = lt_row(pm, c, rows_per_chunk, r0, i);
        if (u >= row_end) continue;
        const uint32_t rl = i << shift;
        uint32_t e = g.rowptr[u], b = 0;
        const uint32_t end = g.rowptr[u + 1];
        if (end - e >= kLtLongRow && i < kLtLongCap) {
            longrow[atomicAdd(&nlong, 1u)] = i;
            continue;
        }
        while (e < end) {
            b = lt_block(pm, g.col[e], block_cols, nblocks, b);
            uint32_t f = e + 1;
            while (f < end && lt_block(pm, g.col[f], block_cols, nblocks, b) == b) ++f;
            uint32_t pos = atomicAdd(&cursor[b], f - e);
            const uint32_t bs = lt_block_start(pm, b, block_cols);
            for (; e < f; ++e) entries[pos++] = rl | (g.col[e] - bs);
        }
    }
    __syncthreads();
    // A long row, 64 entries a trip.  A row's entries of one block must land as ONE adjacent run, in order: the lane at the
    // head of a run reserves the whole run at once — for the trip's last run, which may go on, after counting how far it
    // goes — and the trips that follow write the rest of that run behind what is already there.
    for (uint32_t k = wave; k < nlong; k += 16) {
        const uint32_t i = longrow[k], u = lt_row(pm, c, rows_per_chunk, r0, i), rl = i << shift;
        const uint32_t rs = g.rowptr[u], re = g.rowptr[u + 1];
        uint32_t cb = 0xFFFFFFFFu, pos_base = 0;      // the run open at the start of a trip: its block, where its next entry goes
        for (uint32_t e0 = rs; e0 < re; e0 += 64) {
            const uint32_t e = e0 + lane;
            const bool in = e < re;
            const uint32_t col = in ? g.col[e] : 0u;
            const uint32_t b = in ? lt_block(pm, col, block_cols, nblocks, 0) : 0xFFFFFFFEu;
            uint32_t bp = lane_prev(b);
            if (lane == 0) bp = cb;
            const bool head = in && b != bp;
            const unsigned long long hm = __ballot(head);
            const uint32_t nin = (uint32_t)__popcll(__ballot(in));
            const uint32_t first_head = hm ? (uint32_t)__builtin_ctzll(hm) : nin;   // lanes in front of it continue the open run
            if (in && lane < first_head) entries[pos_base + lane] = rl | (col - lt_block_start(pm, cb, block_cols));
            pos_base += first_head;
            if (hm == 0ull) continue;                  // (uniform) the whole trip belonged to the open run
            const uint32_t last_h = 63u - (uint32_t)__builtin_clzll(hm);
            const unsigned long long mine = hm & lanes_upto(lane);
            const uint32_t my_head = mine ? 63u - (uint32_t)__builtin_clzll(mine) : 0u;
            uint32_t len = 0;
            if (head) {
                const unsigned long long later = hm & ~lanes_upto(lane);
                len = (later ? (uint32_t)__builtin_ctzll(later) : nin) - lane;
            }
            uint32_t extra = 0;                        // entries of the last run beyond this trip
            if (e0 + 64 < re) {
                const uint32_t bo = __shfl(b, (int)last_h);
                for (uint32_t f0 = e0 + 64; f0 < re; f0 += 64) {
                    const uint32_t f = f0 + lane;
                    const bool in2 = f < re;
                    const bool same = in2 && lt_block(pm, g.col[in2 ? f : re - 1], block_cols, nblocks, 0) == bo;
                    const unsigned long long sm = __ballot(same);
                    const uint32_t lead = ~sm ? (uint32_t)__builtin_ctzll(~sm) : 64u;   // (blocks ascend: the same-block lanes are a prefix)
                    extra += lead;
                    if (lead < 64u) break;
                }
            }
            uint32_t pos = 0;
            if (head) pos = atomicAdd(&cursor[b], len + (lane == last_h ? extra : 0u));
            const uint32_t p = __shfl(pos, (int)my_head);
            if (in && lane >= first_head) entries[p + (lane - my_head)] = rl | (col - lt_block_start(pm, b, block_cols));
            cb = __shfl(b, (int)last_h);
            pos_base = __shfl(pos, (int)last_h) + __shfl(len, (int)last_h);
        }
    }
}

// ... of a plan over consecutive rows and uniform blocks (lt_flat_plan): the chunk's entries are one CSR range — every wave
// walks an equal share of it, 64 entries a trip, coalesced.  Whether a row's blocks ascend is checked by the scatter pass,
// which knows the rows.  Dynamic LDS: nblocks counters.
__global__ __launch_bounds__(256) void k_lt_count_flat(GraphDev g, uint32_t rows_per_chunk, uint32_t nblocks, uint32_t block_cols,
                                                       uint32_t *__restrict__ seg_cnt, uint32_t row_base, uint32_t row_end, uint32_t chunk0) {
    // chunk0: the launch covers chunks chunk0 .. (a hand-off builds the plan piece by piece, as the column array arrives).  A
    // column id may not have been checked yet when this runs (the hand-off validates at the end): its block is clamped, so a
    // malformed graph costs a wrong plan that is then thrown away, never an access outside the counters.
    extern __shared__ uint32_t lt_dyn[];
    uint32_t *hist = lt_dyn;
    const uint32_t c = chunk0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nthreads = blockDim.x, nwaves = nthreads >> 6;   // (a multiple of 64)
    const uint32_t r0 = min(row_end, row_base + c * rows_per_chunk), r1 = min(row_end, r0 + rows_per_chunk);
    for (uint32_t i = tid; i < nblocks; i += nthreads) hist[i] = 0;
    __syncthreads();
    const uint32_t es = g.rowptr[r0], all = g.rowptr[r1] - es;
    const uint32_t ee = es + (uint32_t)((uint64_t)all * (wave + 1u) / nwaves);
    const uint32_t magic = block_cols > 1u ? (uint32_t)(0x100000000ull / block_cols) : 0u;
    for (uint32_t eg = es + (uint32_t)((uint64_t)all * wave / nwaves); eg < ee; eg += 64 * kLtTrips) {   // kLtTrips trips' loads in flight at once
        uint32_t colv[kLtTrips];
#pragma unroll
        for (uint32_t j = 0; j < kLtTrips; ++j) {
            const uint32_t e = eg + 64 * j + lane;
            colv[j] = e < ee ? g.col[e] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < kLtTrips; ++j) {
            const uint32_t e0 = eg + 64 * j, e = e0 + lane;
            if (e0 >= ee) break;                       // (uniform)
            const bool in = e < ee;
            const uint32_t b = in ? min(lt_div(colv[j], block_cols, magic), nblocks - 1u) : 0xFFFFFFFFu;
            const uint32_t bp = lane_prev(b);
            const bool head = in && (lane == 0 || b != bp);   // one LDS add per run of equal blocks (a trip's first entry starts one)
            const unsigned long long hm = __ballot(head);
            const uint32_t nin = min(64u, ee - e0);
            if (head) {
                const unsigned long long later = hm & ~lanes_upto(lane);
                const uint32_t nh = later ? (uint32_t)__builtin_ctzll(later) : nin;
                atomicAdd(&hist[b], nh - lane);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nblocks; i += nthreads) seg_cnt[(size_t)c * nblocks + i] = hist[i];
}

// ... of a plan over consecutive rows and uniform blocks (lt_flat_plan).  Every wave walks an equal share of the chunk's CSR
// range (cut at row boundaries) 64 entries a trip, coalesced.  A run = the adjacent entries of one row in one block; the lane
// at its head reserves it whole — the trip's last run, which may go on, after counting how far it goes.  *bad |= 2 if a
// row's blocks do not ascend.  Dynamic LDS: nblocks cursors, rows + 1 offsets, 64 flags a wave.
__global__ __launch_bounds__(1024) void k_lt_scatter_flat(GraphDev g, uint32_t rows_per_chunk, uint32_t nblocks, uint32_t block_cols,
                                                          uint32_t shift, const uint32_t *__restrict__ seg_cnt,
                                                          uint32_t *__restrict__ entries, uint32_t row_base, uint32_t row_end,
                                                          uint32_t slack, uint32_t *bad, uint32_t stage_cap, uint32_t chunk0) {
    extern __shared__ uint32_t lt_dyn[];
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t range_end;
    uint32_t *cursor = lt_dyn, *rp = lt_dyn + nblocks, *row_flag = rp + rows_per_chunk + 1, *stage = row_flag + blockDim.x;
    const uint32_t c = chunk0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nthreads = blockDim.x, nwaves = nthreads >> 6;   // (a multiple of 64)
    const uint32_t r0 = min(row_end, row_base + c * rows_per_chunk), r1 = min(row_end, r0 + rows_per_chunk);
    const uint32_t first = lt_chunk_first(g, c, rows_per_chunk, row_base, row_end, slack, nullptr);
    bool unsorted = false;
    for (uint32_t i = tid; i <= r1 - r0; i += nthreads) rp[i] = g.rowptr[r0 + i];
    {   // cursor[b] = first + exclusive scan of the (padded) counts: thread t takes blocks [t * per, (t + 1) * per)
        const uint32_t per = (nblocks + nthreads - 1) / nthreads, b0 = tid * per;
        const uint32_t *cnt = seg_cnt + (size_t)c * nblocks;
        uint32_t local = 0;
        for (uint32_t k = 0; k < per && b0 + k < nblocks; ++k) local += slack ? (cnt[b0 + k] + 3u) & ~3u : cnt[b0 + k];
        uint32_t incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if ((int)lane >= off) incl += t;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t run = first + incl - local;
        for (uint32_t w = 0; w < wave; ++w) run += wave_sum[w];
        for (uint32_t k = 0; k < per && b0 + k < nblocks; ++k) {
            cursor[b0 + k] = run;
            run += slack ? (cnt[b0 + k] + 3u) & ~3u : cnt[b0 + k];
        }
        if (tid == nthreads - 1) range_end = run;
        __syncthreads();
    }
    // The chunk's regrouped range is put together in LDS and written out in one coalesced sweep (scattered 4-byte stores
    // were half of this kernel's time); a chunk whose range is longer than the staging area stores directly.
    const uint32_t span = range_end - first;
    const bool staged = span <= stage_cap;             // (uniform over the workgroup)
    if (staged) {
        for (uint32_t i = tid; i < span; i += nthreads) stage[i] = 0;   // (pad slots are read, never used)
        __syncthreads();
    }
#define LT_PUT(pos_, v_) do { if (staged) stage[(pos_) - first] = (v_); else entries[(pos_)] = (v_); } while (0)
    {
        const uint32_t nr = r1 - r0, i0 = lt_wave_cut(rp, nr, wave, nwaves), i1 = lt_wave_cut(rp, nr, wave + 1, nwaves);
        uint32_t ck = 0xFFFFFFFFu, pos_base = 0;       // the run open at the start of a trip: its key, where its next entry goes
        uint32_t from = i0;                            // row of the previous trip's last entry
        const uint32_t ee = rp[i1], magic = block_cols > 1u ? (uint32_t)(0x100000000ull / block_cols) : 0u;
        for (uint32_t eg = rp[i0]; eg < ee; eg += 64 * kLtTrips) {   // kLtTrips trips' loads in flight at once
          uint32_t colv[kLtTrips];
#pragma unroll
          for (uint32_t j = 0; j < kLtTrips; ++j) {
              const uint32_t e = eg + 64 * j + lane;
              colv[j] = e < ee ? g.col[e] : 0u;
          }
#pragma unroll
          for (uint32_t j = 0; j < kLtTrips; ++j) {
            const uint32_t e0 = eg + 64 * j, e = e0 + lane;
            if (e0 >= ee) break;                       // (uniform)
            const bool in = e < ee;
            const uint32_t nin = min(64u, ee - e0);
            const uint32_t col = colv[j];
            const uint32_t b = in ? min(lt_div(col, block_cols, magic), nblocks - 1u) : 0u;   // (clamped: see k_lt_count_flat)
            // Rows of the trip's entries.  `from` holds the entry before the trip (or is the share's first row); without
            // empty rows the rows that start inside the trip are among the next 64: lane l looks at row from + 1 + l, flags
            // the trip position where it starts, and an entry's row = from + the flags up to its position.  An empty row
            // among them (two rows starting at one position) sends the trip to the search instead.
            uint32_t r;
            {
                const uint32_t s0 = rp[min(from + 1u + lane, nr)], s1 = rp[min(from + 2u + lane, nr)];
                const bool inside = s0 >= e0 && s0 < e0 + nin;
                if (__any(inside && s0 == s1)) {
                    r = lt_row_of(rp, from, i1 - 1u, in ? e : e0 + nin - 1u);
                } else {
                    uint32_t *flag = row_flag + wave * 64u;
                    flag[lane] = 0u;
                    wave_lds_sync();
                    if (inside) flag[s0 - e0] = 1u;
                    wave_lds_sync();
                    const unsigned long long starts = __ballot(flag[lane] != 0u);
                    wave_lds_sync();                   // (the next trip clears the flags again)
                    r = from + (uint32_t)__popcll(starts & lanes_upto(lane));
                }
                from = __shfl(r, (int)(nin - 1u));
            }
            const uint32_t key = in ? (r << 12 | b) : 0xFFFFFFFEu;   // (r <= 2048, b < 4096)
            const uint32_t val = (r << shift) | (col - b * block_cols);
            uint32_t kp = lane_prev(key);
            if (lane == 0) kp = ck;
            unsorted |= in && key < kp && (key >> 12) == (kp >> 12);   // the row's blocks must ascend
            const bool head = in && key != kp;
            const unsigned long long hm = __ballot(head);
            const uint32_t first_head = hm ? (uint32_t)__builtin_ctzll(hm) : nin;   // lanes in front of it continue the open run
            if (in && lane < first_head) LT_PUT(pos_base + lane, val);
            pos_base += first_head;
            if (hm == 0ull) continue;                  // (uniform) the whole trip belonged to the open run
            const uint32_t last_h = 63u - (uint32_t)__builtin_clzll(hm);
            const unsigned long long mine = hm & lanes_upto(lane);
            const uint32_t my_head = mine ? 63u - (uint32_t)__builtin_clzll(mine) : 0u;
            uint32_t len = 0;
            if (head) {
                const unsigned long long later = hm & ~lanes_upto(lane);
                len = (later ? (uint32_t)__builtin_ctzll(later) : nin) - lane;
            }
            uint32_t extra = 0;                        // entries of the last run beyond this trip
            const uint32_t bo = __shfl(b, (int)last_h), row_end_e = rp[__shfl(r, (int)last_h) + 1u];
            for (uint32_t f0 = e0 + 64; f0 < row_end_e; f0 += 64) {
                const uint32_t f = f0 + lane;
                const bool in2 = f < row_end_e;
                const bool same = in2 && min(lt_div(g.col[in2 ? f : row_end_e - 1], block_cols, magic), nblocks - 1u) == bo;
                const unsigned long long sm = __ballot(same);
                const uint32_t lead = ~sm ? (uint32_t)__builtin_ctzll(~sm) : 64u;   // (blocks ascend: the same-block lanes are a prefix)
                extra += lead;
                if (lead < 64u) break;
            }
            uint32_t pos = 0;
            if (head) pos = atomicAdd(&cursor[b], len + (lane == last_h ? extra : 0u));
            const uint32_t p = __shfl(pos, (int)my_head);
            if (in && lane >= first_head) LT_PUT(p + (lane - my_head), val);
            ck = __shfl(key, (int)last_h);
            pos_base = __shfl(pos, (int)last_h) + __shfl(len, (int)last_h);
          }
        }
    }
#undef LT_PUT
    if (staged) {
        __syncthreads();
        for (uint32_t i = tid; i < span; i += nthreads) entries[first + i] = stage[i];
    }
    if (__any(unsorted) && lane == 0) atomicOr(bad, 2u);
}

// Skewed graphs: the rows of the degree-sorted list (heaviest first) dealt to `nslices` slices, serpentine — round t
// hands ranks t * nslices .. to the slices left to right, the next round right to left — so that every slice gets the
// same weight to within one row.  rowmap[s * slice_rows + t] = the row (0xFFFFFFFF past the list's end),
// weight[s] = the slice's entries.  One wave per slice.
__global__ __launch_bounds__(256) void k_map_deal(GraphDev g, const uint32_t *__restrict__ sorted_rows, uint32_t m,
                                                  uint32_t slice_rows, uint32_t nslices, uint32_t *__restrict__ rowmap,
                                                  uint32_t *__restrict__ weight) {
    const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= nslices) return;
    uint32_t w = 0;
    for (uint32_t t = lane; t < slice_rows; t += 64) {
        const uint64_t idx = (uint64_t)t * nslices + ((t & 1u) ? nslices - 1u - s : s);
        uint32_t u = 0xFFFFFFFFu;
        if (idx < m) {
            u = sorted_rows[idx];
            w += g.rowptr[u + 1] - g.rowptr[u];
        }
        rowmap[(size_t)s * slice_rows + t] = u;
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) w += __shfl_xor(w, off);
    if (lane == 0) weight[s] = w;
}

// cand[k] = the first row whose CSR offset reaches k * target (k = 0 .. count - 1): on a symmetric adjacency, column
// ranges of equal entry mass
__global__ __launch_bounds__(256) void k_mass_bounds(GraphDev g, unsigned long long target, uint32_t count, uint32_t *__restrict__ cand) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const unsigned long long want = (unsigned long long)g.rowptr[0] + target * k;
    uint32_t lo = 0, hi = g.n;   // first row r with rowptr[r] >= want
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((unsigned long long)g.rowptr[mid] >= want) hi = mid;
        else lo = mid + 1;
    }
    cand[k] = lo;
}


// Workgroup steps of the LDS-table plan.  A chunk = 16 slices of rows, one per wave; a step = one column block
// (whose byte slice the workgroup stages in LDS) with up to 256 entries per slice.  Record (kLtwRec words):
// [0] block, [2 + w] first entry of slice w, [18 + w] its count.  One thread per chunk.
// write == 0: step_count[c] = number of records of chunk c (padded to a multiple of 4).
constexpr uint32_t kLtwRec = 36;
__global__ __launch_bounds__(64) void k_ltw_steps(GraphDev g, uint32_t slice_rows, uint32_t nchunks, uint32_t nblocks,
                                                  const uint32_t *__restrict__ seg_cnt, const uint32_t *__restrict__ step_ptr,
                                                  uint32_t *__restrict__ step_count, uint32_t *__restrict__ recs, int write,
                                                  uint32_t cap, uint32_t slack, uint32_t block_cols, PlanMap pm, uint32_t row_base,
                                                  uint32_t row_end, uint32_t piece_cols) {
    // record words [1] = the block's first column, [34] = the 16-byte piece of the table its last column sits in (piece_cols =
    // vertices per 16 bytes of table: 16 for the byte table; 0 = the compact-table plan, which does not use the word)
    if (piece_cols == 0) piece_cols = 16;
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    uint32_t first[16];
#pragma unroll
    for (uint32_t w = 0; w < 16; ++w) first[w] = lt_chunk_first(g, c * 16 + w, slice_rows, row_base, row_end, slack, pm.first);
    const uint32_t pos = write ? step_ptr[c] : 0;
    uint32_t made = 0, last_block = 0;
    for (uint32_t b = 0; b < nblocks; ++b) {
        uint32_t cnt[16], mx = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) {
            cnt[w] = seg_cnt[(size_t)(c * 16 + w) * nblocks + b];
            mx = max(mx, cnt[w]);
        }
        for (uint32_t t = 0; t * cap < mx; ++t) {
            if (write) {
                uint32_t *rec = recs + (size_t)(pos + made) * kLtwRec;
                rec[0] = b;
                rec[1] = lt_block_start(pm, b, block_cols);
#pragma unroll
                for (uint32_t w = 0; w < 16; ++w) {
                    rec[2 + w] = first[w] + t * cap;
                    rec[18 + w] = cnt[w] > t * cap ? min(cap, cnt[w] - t * cap) : 0u;
                }
                rec[34] = (min(lt_block_start(pm, b + 1, block_cols), g.n) + piece_cols - 1u) / piece_cols;
                rec[35] = 0;
            }
            ++made;
            last_block = b;
        }
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) {
            first[w] += cnt[w];
            if (slack) first[w] = (first[w] + 3u) & ~3u;
        }
    }
    const uint32_t padded = made ? (made + 3u) / 4u * 4u : 4u;
    if (write)
        for (; made < padded; ++made) {
            uint32_t *rec = recs + (size_t)(pos + made) * kLtwRec;
            for (uint32_t i = 0; i < kLtwRec; ++i) rec[i] = 0;
            rec[0] = last_block;
            rec[1] = lt_block_start(pm, last_block, block_cols);
            rec[34] = (min(lt_block_start(pm, last_block + 1, block_cols), g.n) + piece_cols - 1u) / piece_cols;
        }
    else
        step_count[c] = padded;
}

// sums of the rows of chunks [chunk0, chunk0 + gridDim.x) -> agg[row]; nothing if *bad != 0.
// A WAVE owns a slice of the chunk's rows (sums in its own part of LDS) and walks its own entries; the workgroup
// shares the column block's byte slice.  Per step: the lane takes four consecutive entries (one 16-byte load,
// two steps ahead), looks their bytes up, and folds runs of the same row lane-locally, a run that reaches the
// lane's last entry continuing with the next lane's entries (whole-wave DPP shift) — see k_c4_agg, which does
// the same with four floats per row.  Two barriers per step: everyone is done with the slice / the next one
// is in place (it was loaded into registers before the step's sums).
constexpr uint32_t kLtwBlock = 81920;       // bytes of the LDS slice = vertices per column block of the BYTE table (lt_block_cols)
constexpr uint32_t kLtwSliceRows = 1221;    // rows per slice: 16 x 4 B x rows + 1 KiB + 80 KiB <= 160 KiB
// (only the byte table has a look-up table of its 256 values; the wider forms divide — (float)k / ws, the IEEE division that made
// x — which hides under the slice's streaming.  A 1024-entry table for the 10-bit form was measured: its 4 KiB cost 48 rows per
// slice, the metric graph's 512 chunks became 768 and stage 0 took 1.56 ms instead of 1.3.)
__host__ __device__ constexpr uint32_t lt_lut_floats(uint32_t bits) { return bits == 8 ? 256u : 0u; }
__host__ __device__ constexpr uint32_t lt_slice_rows_max(uint32_t) { return kLtwSliceRows; }
constexpr uint32_t kLtwStep = 256;          // entries per slice and step
constexpr uint32_t kLtwShift = 17;
constexpr uint32_t kLtwNoRow = (1u << (32 - kLtwShift)) - 1u;
static_assert(kLtwBlock <= (1u << kLtwShift) && kLtwSliceRows < kLtwNoRow - 1u, "entry fields");
static_assert(kLtwBlock % (16 * 1024) == 0, "slice pieces per thread");

__device__ __forceinline__ float ltw_sel(bool c, float v) { return c ? v : 0.0f; }   // values >= +0: x + 0 == x

template <int BITS>
__global__ __launch_bounds__(1024) void k_lt_agg(const uint32_t *__restrict__ step_ptr, const uint32_t *__restrict__ recs,
                                                 const uint32_t *__restrict__ entries, const uint8_t *__restrict__ wbyte, float ws,
                                                 float *__restrict__ agg, uint32_t n, uint32_t slice_rows, uint32_t chunk0,
                                                 uint32_t last_entry, const uint32_t *__restrict__ bad,
                                                 const uint32_t *__restrict__ rowmap, uint32_t row_base, uint32_t row_end) {
    // (row_base / row_end: the plan's row range — the whole graph, or the rows one rank of a partitioned run holds; n = the
    // number of COLUMNS, i.e. the whole graph's vertices)
    // rowmap != nullptr (skewed graphs): slice s holds the rows rowmap[s * slice_rows ..] (0xFFFFFFFF = none), dealt from
    // the degree-sorted list, and the column blocks have their own widths (record words 1 and 34)
    extern __shared__ __attribute__((aligned(16))) unsigned char lt_smem[];
    if (*bad) return;                                                   // block-uniform
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *A = reinterpret_cast<float *>(lt_smem) + wave * slice_rows;  // this wave's sums
    float *lut = reinterpret_cast<float *>(lt_smem) + 16 * slice_rows;  // 256 / 1024 / 0 floats
    uint8_t *slice = reinterpret_cast<uint8_t *>(lut + lt_lut_floats(BITS));   // kLtwBlock bytes (16-byte aligned: slice_rows * 64 is)
    constexpr uint32_t kPiece = lt_piece_cols(BITS);                    // vertices per 16-byte piece of the table
    const uint32_t chunk = chunk0 + blockIdx.x;
    const uint32_t row0 = row_base + (chunk * 16 + wave) * slice_rows;
    for (uint32_t i = lane; i < slice_rows; i += 64) A[i] = 0.0f;
    if (tid < lt_lut_floats(BITS)) lut[tid] = (float)tid / ws;          // the very expression that makes x (checked per forward)
    const uint32_t st0 = __builtin_amdgcn_readfirstlane(step_ptr[chunk]);
    const int nsteps = (int)(__builtin_amdgcn_readfirstlane(step_ptr[chunk + 1]) - st0);   // a multiple of 4; 8 more records are readable
    const uint32_t last_piece = (n + kPiece - 1u) / kPiece;             // the table is padded beyond this
    constexpr int SW = kLtwBlock / 16 / 1024;
    uint4 e[4];                              // entries of step s in slot s & 3
    uint32_t cnt[4], bk[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        e[a] = uint4{0, 0, 0, 0};
        cnt[a] = 0;
        bk[a] = 0;
    }
    u32x4 sr[SW];
    // descriptor of the step whose entries are loaded next: block, this wave's first entry and count
    uint32_t d_blk = recs[(size_t)st0 * kLtwRec + 1] / kPiece, d_first = recs[(size_t)st0 * kLtwRec + 2 + wave], d_cnt = recs[(size_t)st0 * kLtwRec + 18 + wave];
    uint32_t d_end = recs[(size_t)st0 * kLtwRec + 34];   // (d_blk: the first 16-byte piece of the block's bytes, d_end: its last)
    uint32_t be[4] = {0, 0, 0, 0};
    int d_step = 0;
    for (int u = -4; u < nsteps; u += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = u + j;             // the step whose sums are done now
            {   // entries of step s + 2, descriptor of step s + 3
                const int se = s + 2;
                const int slot = (j + 2) & 3;
                const bool on = se >= 0 && se < nsteps && d_step == se;
                bk[slot] = d_blk;
                be[slot] = d_end;
                cnt[slot] = on ? d_cnt : 0u;
                const uint32_t x = d_first + 4u * lane;                  // first is a multiple of 4
                const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(entries + (x < last_entry ? x : last_entry)));
                e[slot] = uint4{q[0], q[1], q[2], q[3]};
                const int nx = se + 1;
                const uint32_t nxc = (uint32_t)(nx < 0 ? 0 : (nx < nsteps ? nx : nsteps - 1));
                const uint32_t *rec = recs + (size_t)(st0 + nxc) * kLtwRec;
                d_blk = rec[1] / kPiece;     // (wave-uniform values, left in vector registers: nothing waits for
                d_first = rec[2 + wave];     //  them before the next trip)
                d_cnt = rec[18 + wave];
                d_end = rec[34];
                d_step = nx;
            }
            {   // the byte slice of step s + 1's block, into registers
                const uint32_t b = bk[(j + 1) & 3], pe = min(be[(j + 1) & 3], last_piece);
#pragma unroll
                for (int k = 0; k < SW; ++k) {
                    uint32_t piece = b + tid + 1024u * k;    // (pieces past the block's end are never looked at: one clamped address)
                    piece = piece < pe ? piece : pe;
                    sr[k] = reinterpret_cast<const u32x4 *>(wbyte)[piece];
                }
            }
            if (s >= 0 && __builtin_amdgcn_readfirstlane(cnt[j & 3])) {   // sums of step s (the slice of its block is in LDS)
                const int slot = j & 3;
                const uint32_t w[4] = {e[slot].x, e[slot].y, e[slot].z, e[slot].w};
                uint32_t r[4];
                float a[4], val[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool in = 4u * lane + k < cnt[slot];
                    r[k] = in ? (w[k] >> kLtwShift) : kLtwNoRow;
                    const uint32_t cl = in ? (w[k] & ((1u << kLtwShift) - 1u)) : 0u;   // column within the block
                    if constexpr (BITS == 8) {
                        val[k] = lut[slice[cl]];
                    } else if constexpr (BITS == 10) {
                        const uint32_t wd = cl / 3u;
                        val[k] = (float)((reinterpret_cast<const uint32_t *>(slice)[wd] >> (10u * (cl - 3u * wd))) & 1023u) / ws;
                    } else {   // (no table of 65 536 values: the division itself, IEEE like the one that made x)
                        val[k] = (float)reinterpret_cast<const uint16_t *>(slice)[cl] / ws;
                    }
                    a[k] = A[r[k] < slice_rows ? r[k] : 0];
                }
                uint32_t prev3 = lane_prev(r[3]);
                if (lane == 0) prev3 = kLtwNoRow - 1u;
                const bool h0 = r[0] != kLtwNoRow && r[0] != prev3;
                const bool h1 = r[1] != kLtwNoRow && r[1] != r[0];
                const bool h2 = r[2] != kLtwNoRow && r[2] != r[1];
                const bool h3 = r[3] != kLtwNoRow && r[3] != r[2];
                uint32_t nr[4];
                float nv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    nr[k] = lane_next(r[k]);
                    nv[k] = lane_next(val[k]);
                    if (lane == 63) nr[k] = kLtwNoRow;
                }
                float s0 = a[0] + val[0];
                s0 += ltw_sel(r[1] == r[0], val[1]);
                s0 += ltw_sel(r[2] == r[0], val[2]);
                s0 += ltw_sel(r[3] == r[0], val[3]);
                float s1 = a[1] + val[1];
                s1 += ltw_sel(r[2] == r[1], val[2]);
                s1 += ltw_sel(r[3] == r[1], val[3]);
                float s2 = a[2] + val[2];
                s2 += ltw_sel(r[3] == r[2], val[3]);
                float s3 = a[3] + val[3];
                const bool o0 = h0 && r[0] == r[3], o1 = h1 && r[1] == r[3], o2 = h2 && r[2] == r[3], o3 = h3;
                float x = o0 ? s0 : o1 ? s1 : o2 ? s2 : s3;
                bool more = (o0 || o1 || o2 || o3) && r[3] != kLtwNoRow;
                for (;;) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) x += ltw_sel(more && nr[k] == r[3], nv[k]);
                    more = more && nr[3] == r[3];
                    if (!__any(more)) break;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {   // one lane further
                        nr[k] = lane_next(nr[k]);
                        nv[k] = lane_next(nv[k]);
                        if (lane == 63) nr[k] = kLtwNoRow;
                    }
                }
                if (o0) s0 = x;
                if (o1) s1 = x;
                if (o2) s2 = x;
                if (o3) s3 = x;
                if (h0) A[r[0]] = s0;
                if (h1) A[r[1]] = s1;
                if (h2) A[r[2]] = s2;
                if (h3) A[r[3]] = s3;
            }
            __syncthreads();                 // everyone is done with the slice in LDS
#pragma unroll
            for (int k = 0; k < SW; ++k) reinterpret_cast<u32x4 *>(slice)[tid + 1024 * k] = sr[k];
            __syncthreads();                 // the slice of step s + 1 is in place
        }
    }
    for (uint32_t i = lane; i < slice_rows; i += 64) {
        const uint32_t row = rowmap ? rowmap[(size_t)(chunk * 16 + wave) * slice_rows + i] : row0 + i;
        if (row < row_end) agg[row] = A[i];
    }
}

// ---- compact-table plan of the 16-wide stages ---------------------------------------------------
// The 64-byte-row gather is bound by line requests to the fabric (~55 G/s).  After the
// ReLU that ends a stage most feature columns are zero on many graphs (metric graph: two columns dense,
// two at ~10 %, a handful of stray non-zeros elsewhere).  When at most FOUR columns carry (nearly) all
// non-zeros of this forward's input, the neighbours' rows are read from a COMPACT table — 4 floats per
// vertex — with a blocked traversal: rows in chunks of 16 SLICES (one 1024-thread workgroup, one slice per
// wave, 4 sums per row in the wave's part of LDS), columns in blocks of ~128 K vertices = 2 MiB of the
// table, the workgroups of a 256-wide persistent grid sweeping the blocks in the same order at the same pace,
// so that the block being read stays in every XCD's L2 and a gather is an L2 hit instead of a fabric request.
//   k_c4_choose   picks the four fullest columns from the per-column counts (device side, no host sync)
//   k_c4_compact  writes the table; a vertex with a non-zero in any OTHER column gets the sign bit of
//                 its first value set (inputs are >= 0 after ReLU — checked — so the bit is free)
//   k_c4_agg      the sums; a row that met a flagged neighbour is marked DIRTY (sign bit of its first sum)
//   k_stage_f16   takes a clean row's aggregate from the 4 sums (+0.0f elsewhere: a sum of zeros) and
//                 gathers a dirty row the plain way, full rows in CSR order — so every aggregate is
//                 exactly what the plain gather produces; then the dense layers as always.
// A skewed graph's input often has more than four live columns (R-MAT-22: ten after stage 1, seven after stage 2): the
// plan then runs up to kC4MaxPasses passes, each with its own table of four columns and its own four sums per row.
// desc words (16 per consumer stage): [0] passes (0 = the plan does not apply to this input), [1..4] the columns of pass 0
// (ascending over all passes; 0xFFFFFFFF = none), [8..11] pass 1, [12..15] pass 2.
constexpr uint32_t kC4Block = 131072;     // default vertices per column block: 2 MiB of compact rows (the engine sizes the
                                          // blocks so that a slice has about 160 entries per block: one step, 54 lanes busy)
constexpr uint32_t kC4Shift = 18;         // entries: row_local << 18 | col_local (blocks of up to 262144 vertices)
constexpr uint32_t kC4Slices = 16;        // slices per chunk = waves per workgroup
constexpr uint32_t kC4MaxPasses = 3;      // tables of four columns a plan may use for one input
constexpr uint32_t kC4SliceRows = 632;    // rows per slice: 16 x (16 B x rows + dirty bits) <= 160 KiB
constexpr uint32_t kC4MaxRows = kC4Slices * kC4SliceRows;
constexpr int kC4LaneEntries = 3;         // consecutive entries a lane takes per step
constexpr uint32_t kC4Step = 64u * kC4LaneEntries;   // entries per step
constexpr uint32_t kC4DirtyWords = (kC4SliceRows + 31) / 32;
constexpr uint32_t kC4NoRow = (1u << (32 - kC4Shift)) - 1u;   // row field of a slot past the step's end
static_assert(kC4SliceRows < kC4NoRow - 1u, "row field too narrow");

// desc words: [0] passes, [1..4] / [8..11] / [12..15] the chosen columns, [5] dirty-row counter, [6] the table(s) still have to
// be written (k_c4_compact) — 0 when the producing stage kernel already wrote the one table for exactly these columns.
// slots == 1: counts[16] from k_column_counts.  slots == 64: the producer's counters (17 per slot; the 17th
// counts the rows it saw — if that is not n, the producer that ran was not the emitting one and nothing here can
// be trusted for this forward).  desc on entry = the previous forward's choice (the producer's spec).
__global__ __launch_bounds__(64) void k_c4_choose(const unsigned long long *__restrict__ counts, int slots, uint32_t n,
                                                  uint32_t *__restrict__ desc, uint32_t max_passes) {
    // one wave, everything in registers: lane i < 17 ends up with the total of counter i (16 columns + the rows seen)
    const int lane = threadIdx.x;
    unsigned long long mine = 0;
    if (slots == 1) {
        mine = lane < 16 ? counts[lane] : (lane == 16 ? (unsigned long long)n : 0ull);
    } else {
        // slot sl's 17 counters sit at counts[sl * kEmitStride ..]: the 64 lanes read 64 counters a trip (coalesced, all trips
        // in flight at once — a lane walking its counter through the slots one dependent load after the other took 22 us), and
        // counter i's total is gathered from the lanes that hold its pieces
        unsigned long long part[kEmitStride];
#pragma unroll
        for (int t = 0; t < kEmitStride; ++t) {
            const int idx = t * 64 + lane;                 // counts[idx] = counter idx % 17 of slot idx / 17
            part[t] = idx < slots * kEmitStride ? counts[idx] : 0ull;
        }
        // lane L holds pieces of counters (t * 64 + L) % 17, t = 0..16; sum per counter over all lanes and trips
        for (int c = 0; c < kEmitStride; ++c) {
            unsigned long long v = 0;
#pragma unroll
            for (int t = 0; t < kEmitStride; ++t) v += ((t * 64 + lane) % kEmitStride == c) ? part[t] : 0ull;
#pragma unroll
            for (int off = 32; off; off >>= 1)
                v += ((unsigned long long)__shfl_xor((unsigned)(v >> 32), off) << 32) | __shfl_xor((unsigned)v, off);
            if (lane == c) mine = v;
        }
    }
    const unsigned long long rows = ((unsigned long long)__shfl((unsigned)(mine >> 32), 16) << 32) | __shfl((unsigned)mine, 16);
    const bool prev_ok = desc[0] == 1u;
    const uint32_t p1 = desc[1], p2 = desc[2], p3 = desc[3], p4 = desc[4];
    if (rows != n) {   // no (complete) statistics for this input
        if (lane == 0) {
            desc[0] = 0;
            desc[6] = 0;
        }
        return;
    }
    // columns by fullness (ties: lowest index); the plan takes the 4, 8 or 12 fullest — as few as leave at most n / 512
    // stray non-zeros outside them (every stray flags a vertex and dirties that vertex's neighbours, which are then
    // recomputed from full rows: worth it only while they are few) — and no more than max_passes x 4
    const unsigned long long c = lane < 16 ? mine : 0ull;
    uint32_t rank = 0;   // how many columns are fuller than column `lane`
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const unsigned long long ck = ((unsigned long long)__shfl((unsigned)(c >> 32), k) << 32) | __shfl((unsigned)c, k);
        rank += (ck > c || (ck == c && k < lane)) ? 1u : 0u;
    }
    uint32_t np = 0;
#pragma unroll
    for (uint32_t t = 3; t >= 1; --t) {   // the smallest t that fits wins
        unsigned long long rest = (lane < 16 && rank >= 4 * t) ? c : 0ull;
#pragma unroll
        for (int off = 8; off; off >>= 1)
            rest += ((unsigned long long)__shfl_xor((unsigned)(rest >> 32), off) << 32) | __shfl_xor((unsigned)rest, off);
        rest = ((unsigned long long)__shfl((unsigned)(rest >> 32), 0) << 32) | __shfl((unsigned)rest, 0);
        if (t <= max_passes && rest <= (unsigned long long)n / 512) np = t;
    }
    // the chosen columns in ascending order, four per pass: pass 0 at desc[1..4], pass 1 at [8..11], pass 2 at [12..15]
    const bool chosen = lane < 16 && rank < 4 * np;
    const unsigned long long cm = __ballot(chosen);
    const uint32_t pos = (uint32_t)__popcll(cm & ((1ull << lane) - 1ull));
    const uint32_t c0 = (uint32_t)__builtin_ctzll(cm | (1ull << 63));
    const uint32_t m1 = (uint32_t)(cm & (cm - 1)), m2 = m1 & (m1 - 1), m3 = m2 & (m2 - 1);   // (all within bits 0..15)
    const uint32_t c1 = m1 ? (uint32_t)__builtin_ctz(m1) : 64u, c2 = m2 ? (uint32_t)__builtin_ctz(m2) : 64u,
                   c3 = m3 ? (uint32_t)__builtin_ctz(m3) : 64u;
    const bool table_written = slots > 1 && prev_ok && np == 1u && p1 == c0 && p2 == c1 && p3 == c2 && p4 == c3;
    if (lane < 12) desc[lane < 4 ? 1 + lane : 4 + lane] = 0xFFFFFFFFu;   // (slots without a column)
    if (chosen) desc[pos < 4 ? 1 + pos : 4 + pos] = (uint32_t)lane;
    if (lane == 0) {
        desc[0] = np;
        desc[6] = (np && !table_written) ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void k_c4_compact(const float4 *__restrict__ feat, uint32_t n, uint32_t *__restrict__ desc,
                                                    f32x4 *__restrict__ table) {
    if (!desc[0] || !desc[6]) return;   // not fit, or the producing kernel wrote the table already
    // (a block that finds a negative value below clears desc[0] for the kernels that follow)
    // table of pass q at table + q * (n + 1): its four columns (0xFFFFFFFF = none: +0.0f); the sign bit of a vertex's
    // first value says "has non-zeros in columns no pass carries" in EVERY pass's table
    const uint32_t npass = desc[0];
    uint32_t mask = 0;
    for (uint32_t q = 0; q < npass; ++q)
        for (int k = 0; k < 4; ++k) {
            const uint32_t cc = desc[(q == 0 ? 1 : 4 + 4 * q) + k];
            if (cc < 16u) mask |= 1u << cc;
        }
    bool negative = false;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v <= n; v += (size_t)gridDim.x * blockDim.x) {
        uint32_t nz = 0;
        float r[16];
        if (v < n) {
            const float4 q0 = feat[v * 4], q1 = feat[v * 4 + 1], q2 = feat[v * 4 + 2], q3 = feat[v * 4 + 3];
            const float rr[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                r[i] = rr[i];
                nz |= (rr[i] != 0.0f ? 1u : 0u) << i;
                negative |= rr[i] < 0.0f;
            }
        }
        for (uint32_t q = 0; q < npass; ++q) {
            f32x4 out = {0.0f, 0.0f, 0.0f, 0.0f};
            if (v < n) {
                const uint32_t *dc = desc + (q == 0 ? 1 : 4 + 4 * q);
                const float *row = reinterpret_cast<const float *>(feat) + v * 16;   // the four picks: L1 hits
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t cc = dc[k];
                    const float a = cc < 16u ? row[cc] : 0.0f;
                    out[k] = a == 0.0f ? 0.0f : a;                                    // -0.0f -> +0.0f
                }
                if (nz & ~mask) out[0] = __uint_as_float(__float_as_uint(out[0]) | 0x80000000u);   // stray non-zeros: flag the vertex
            }
            table[(size_t)q * ((size_t)n + 1) + v] = out;   // row n: the zero row clamped reads land on
        }
        (void)r;
    }
    if (__any(negative) && (threadIdx.x & 63) == 0) atomicAnd(&desc[0], 0u);
}

__device__ __forceinline__ f32x4 lane_next(f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __uint_as_float(lane_next(__float_as_uint(v[i])));
    return r;
}
// c ? v : +0.0f — every value and sum here is >= +0 (checked inputs, sums from +0), so x + 0 == x bit for bit
__device__ __forceinline__ f32x4 c4_sel(bool c, f32x4 v) {
    return f32x4{c ? v[0] : 0.0f, c ? v[1] : 0.0f, c ? v[2] : 0.0f, c ? v[3] : 0.0f};
}

// The sums.  A WAVE owns a slice of rows: its sums live in its own part of LDS and it walks its own entry
// stream, regrouped per (slice, column block) into steps of <= 256 entries (a row's entries of one block are
// adjacent and in order; a segment starts at a multiple of 4).  A lane takes kC4LaneEntries = 3 CONSECUTIVE entries
// of the step with one 12-byte load, gathers their table rows, and the lane that gathers a row is the lane that
// adds it: runs of the same row are folded lane-locally, a run that reaches the lane's last entry goes on
// with the next lane's entries (one whole-wave DPP shift of its rows and values; more shifts only for
// runs longer than a lane), and every row written in a step is written by exactly one lane — so the LDS
// reads of a lane are issued together.  (Three entries per lane: lanes past a step's end still cost their slot
// in the texture-address path, and 192-entry steps are 83 % full where 256-entry ones are 62 %.)  LDS operations of one wave execute in order, hence no barrier is needed
// for the sums; the one barrier per column block only keeps the 16 waves of a workgroup (and, since all
// workgroups do the same work, the chip) in the same block, which is what keeps the block in L2.
// Loads are unconditional and in the same order on every trip so that the hardware counters are waited on
// exactly: entries two steps ahead, gathers one step ahead.
// (TAG only names the launch in profiles: 0 = all chunks of a call, 1 = one round of a call done round by round)
template <int TAG>
__global__ __launch_bounds__(1024) void k_c4_agg(const uint32_t *__restrict__ step_ptr, const uint4 *__restrict__ steps,
                                                 const uint32_t *__restrict__ entries, const f32x4 *__restrict__ table,
                                                 f32x4 *__restrict__ agg, uint32_t n, uint32_t slice_rows, uint32_t slice0,
                                                 uint32_t slice1, uint32_t nslices, uint32_t last_entry, uint32_t *__restrict__ desc,
                                                 uint32_t *__restrict__ dirty_rows, uint32_t dirty_cap, uint32_t block_cols,
                                                 uint32_t nblocks, uint32_t row_base, uint32_t row_end, uint32_t pass,
                                                 const uint32_t *__restrict__ rowmap) {
    // pass: which of the plan's tables `table` is (desc[0] = how many the device chose for this input; the dirty rows are
    // registered by pass 0 only — the flags are the same in every table).  rowmap != nullptr: slice s holds the rows
    // rowmap[s * slice_rows ..] (0xFFFFFFFF = none) instead of slice_rows consecutive ones (skewed graphs: slices of
    // equal weight dealt from the degree-sorted list).
    extern __shared__ __attribute__((aligned(16))) unsigned char c4_smem[];
    if (desc[0] <= pass) return;                                        // block-uniform
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 *A = reinterpret_cast<f32x4 *>(c4_smem) + wave * slice_rows;  // this wave's sums
    uint32_t *dirty = reinterpret_cast<uint32_t *>(c4_smem + (size_t)kC4Slices * slice_rows * 16) + wave * kC4DirtyWords;
    for (uint32_t base = slice0 + blockIdx.x * kC4Slices; base < slice1; base += gridDim.x * kC4Slices) {
        const bool live = base + wave < slice1;
        const uint32_t wc = live ? base + wave : nslices;               // slice `nslices` is empty (step_ptr has one more entry)
        const uint32_t row0 = row_base + (live ? wc * slice_rows : 0u);
        uint32_t cur_blk = 0;
        for (uint32_t i = lane; i < slice_rows; i += 64) A[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (lane < kC4DirtyWords) dirty[lane] = 0;
        const uint32_t st0 = __builtin_amdgcn_readfirstlane(step_ptr[wc]);
        const int nsteps = (int)(__builtin_amdgcn_readfirstlane(step_ptr[wc + 1]) - st0);   // a multiple of 4 (0 for the empty slice)
        constexpr int E = kC4LaneEntries;
        uint32_t e[4][E];                    // entries of step s in slot s & 3
        f32x4 v[2][E];                       // gathered rows of step s in slot s & 1
        uint32_t cb[4], cnt[4], bk[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            cb[a] = 0;
            bk[a] = 0;
            cnt[a] = 0;
#pragma unroll
            for (int k = 0; k < E; ++k) e[a][k] = 0;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int k = 0; k < E; ++k) v[a][k] = f32x4{0, 0, 0, 0};
        uint4 dsc = steps[st0];              // descriptor of the step whose entries are loaded next
        int dsc_step = 0;
        for (int u = -4; u < nsteps; u += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = u + j;         // the step whose sums are done now
                {   // entries of step s + 2, descriptor of step s + 3
                    const int se = s + 2;
                    const int slot = (j + 2) & 3;
                    const bool on = se >= 0 && se < nsteps && dsc_step == se;
                    const uint32_t first = dsc.y, count = on ? dsc.z : 0u;
                    cb[slot] = dsc.w;                                        // the block's first column
                    bk[slot] = dsc.x;
                    cnt[slot] = count;
                    const uint32_t x0 = first + (uint32_t)E * lane;          // first is a multiple of 4
                    const uint32_t x = x0 < last_entry ? x0 : last_entry;
                    if constexpr (E == 4) {
                        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(entries + x));
#pragma unroll
                        for (int k = 0; k < E; ++k) e[slot][k] = q[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < E; ++k) e[slot][k] = __builtin_nontemporal_load(entries + x + k);
                    }
                    const int nx = se + 1;
                    const int nxc = nx < 0 ? 0 : (nx < nsteps ? nx : (nsteps > 0 ? nsteps - 1 : 0));
                    dsc = steps[st0 + (uint32_t)nxc];
                    dsc_step = nx;
                }
                {   // gather step s + 1
                    const int slot = (j + 1) & 3;
                    const int vs = (j + 1) & 1;
#pragma unroll
                    for (int k = 0; k < E; ++k) {
                        const uint32_t c = cb[slot] + (e[slot][k] & ((1u << kC4Shift) - 1u));
                        const bool in = (uint32_t)E * lane + k < cnt[slot];
                        v[vs][k] = table[(in && c < n) ? c : n];             // row n: the zero row
                    }
                }
                if (s >= 0) {   // sums of step s
                    const int slot = j & 3;
                    const int vs = j & 1;
                    if (cnt[slot])
                        while (cur_blk < bk[slot]) {   // pacing: the workgroup enters a column block together
                            __syncthreads();
                            ++cur_blk;
                        }
                    if (cnt[slot]) {
                        uint32_t r[E];
                        f32x4 a[E], val[E];
#pragma unroll
                        for (int k = 0; k < E; ++k) {
                            r[k] = ((uint32_t)E * lane + k < cnt[slot]) ? (e[slot][k] >> kC4Shift) : kC4NoRow;
                            a[k] = A[r[k] < slice_rows ? r[k] : 0];
                        }
#pragma unroll
                        for (int k = 0; k < E; ++k) {   // a flagged table row (stray non-zeros) dirties the entry's row
                            val[k] = v[vs][k];
                            const uint32_t f = __float_as_uint(val[k][0]);
                            if ((f >> 31) && r[k] != kC4NoRow) atomicOr(&dirty[r[k] >> 5], 1u << (r[k] & 31));
                            val[k][0] = __uint_as_float(f & 0x7FFFFFFFu);
                        }
                        uint32_t prevr = lane_prev(r[E - 1]);
                        if (lane == 0) prevr = kC4NoRow - 1u;
                        bool h[E];           // the entry starts a run
#pragma unroll
                        for (int k = 0; k < E; ++k) h[k] = r[k] != kC4NoRow && r[k] != (k ? r[k - 1] : prevr);
                        uint32_t nr[E];      // the next lane's rows and values
                        f32x4 nv[E];
#pragma unroll
                        for (int k = 0; k < E; ++k) {
                            nr[k] = lane_next(r[k]);
                            nv[k] = lane_next(val[k]);
                            if (lane == 63) nr[k] = kC4NoRow;
                        }
                        // lane-local folds (a row's entries are adjacent: r[2] == r[0] implies r[1] == r[0])
                        f32x4 sum[E];
#pragma unroll
                        for (int i = 0; i < E; ++i) {
                            sum[i] = a[i] + val[i];
#pragma unroll
                            for (int j = i + 1; j < E; ++j) sum[i] += c4_sel(r[j] == r[i], val[j]);
                        }
                        // the run that holds this lane's last entry goes on in the next lane(s); it is this lane's to
                        // finish iff it starts here
                        bool o[E], any_o = false;
                        f32x4 x = sum[E - 1];
#pragma unroll
                        for (int i = E - 1; i >= 0; --i) {
                            o[i] = h[i] && r[i] == r[E - 1];
                            any_o = any_o || o[i];
                            if (o[i]) x = sum[i];
                        }
                        bool more = any_o && r[E - 1] != kC4NoRow;
                        for (;;) {
#pragma unroll
                            for (int k = 0; k < E; ++k) x += c4_sel(more && nr[k] == r[E - 1], nv[k]);
                            more = more && nr[E - 1] == r[E - 1];
                            if (!__any(more)) break;
#pragma unroll
                            for (int k = 0; k < E; ++k) {   // one lane further
                                nr[k] = lane_next(nr[k]);
                                nv[k] = lane_next(nv[k]);
                                if (lane == 63) nr[k] = kC4NoRow;
                            }
                        }
#pragma unroll
                        for (int i = 0; i < E; ++i) {
                            if (o[i]) sum[i] = x;
                            if (h[i]) A[r[i]] = sum[i];
                        }
                    }
                }
            }
        }
        while (cur_blk < nblocks) {          // every wave passes one barrier per block and chunk
            __syncthreads();
            ++cur_blk;
        }
        if (live)
            for (uint32_t i = lane; i < slice_rows; i += 64) {
                const uint32_t row = rowmap ? rowmap[(size_t)wc * slice_rows + i] : row0 + i;
                if (row >= row_end) continue;   // (0xFFFFFFFF: an unused slot of a mapped slice)
                f32x4 a = A[i];
                if (dirty[i >> 5] >> (i & 31) & 1u) {
                    // a dirty row: its aggregate is recomputed from full rows (k_c4_fix) into slot `slot` of the
                    // side buffer; the sums here are not used.  No slot left: the stage kernel gathers it itself.
                    if (pass == 0) {
                        const uint32_t slot = atomicAdd(&desc[5], 1u);
                        if (slot < dirty_cap) dirty_rows[slot] = row;
                        a[0] = __uint_as_float(0x80000000u);
                        a[1] = __uint_as_float(slot < dirty_cap ? slot : 0xFFFFFFFFu);
                    }
                }
                agg[row] = a;
            }
    }
}

// dirty rows (rows with a neighbour that has stray non-zeros): the plain aggregate, full 64-byte rows in CSR
// order, one quad of lanes per row (lane c: columns 4c .. 4c+3) -> agg16[slot]
__global__ __launch_bounds__(256) void k_c4_fix(GraphDev g, const float4 *__restrict__ fin, const uint32_t *__restrict__ desc,
                                                const uint32_t *__restrict__ dirty_rows, uint32_t dirty_cap,
                                                float4 *__restrict__ agg16, const uint32_t *__restrict__ marks) {
    if (!desc[0]) return;
    // marks != nullptr: only the slots handed out between two marks (one round of the aggregation grid)
    const uint32_t first = marks ? min(marks[0], dirty_cap) : 0u;
    const uint32_t count = min(marks ? marks[1] : desc[5], dirty_cap);
    const uint32_t c = threadIdx.x & 3;
    for (uint32_t slot = first + ((blockIdx.x * blockDim.x + threadIdx.x) >> 2); slot < count; slot += (gridDim.x * blockDim.x) >> 2) {
        const uint32_t u = dirty_rows[slot];
        uint32_t e = g.rowptr[u];
        const uint32_t end = g.rowptr[u + 1];
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (; e + 1 < end; e += 2) {   // two rows in flight, added in order
            const float4 r0 = fin[(size_t)g.col[e] * 4 + c], r1 = fin[(size_t)g.col[e + 1] * 4 + c];
            a.x += r0.x; a.y += r0.y; a.z += r0.z; a.w += r0.w;
            a.x += r1.x; a.y += r1.y; a.z += r1.z; a.w += r1.w;
        }
        if (e < end) {
            const float4 r0 = fin[(size_t)g.col[e] * 4 + c];
            a.x += r0.x; a.y += r0.y; a.z += r0.z; a.w += r0.w;
        }
        agg16[(size_t)slot * 4 + c] = a;
    }
}

// marks[k] = the number of dirty-row slots handed out so far (after round k - 1 of the aggregation grid)
__global__ void k_c4_mark(const uint32_t *__restrict__ desc, uint32_t *__restrict__ marks, uint32_t k) { marks[k] = desc[5]; }

// ------------------------------------------------------------------ pruned adjacency (skewed graphs, 16-wide stages)
// On skewed graphs the trained model drives the features of every high-degree vertex to zero: after the first stage
// no vertex of R-MAT-22 above degree 145 has a non-zero among its 16 values (after the second: above 96), and 73 % /
// 86 % of all adjacency entries point to such vertices.  Adding a row of zeros changes no bit of a sum (x + (+-0) == x,
// and the sums start at +0), so those entries need not be gathered at all.  Per graph and consumer stage the engine
// builds a second CSR without the entries whose target has degree >= a bound (taken from the input it sees when the plan
// is built, plus a margin); per call k_prune_check proves on the device, for the input at hand, that every vertex of
// degree >= bound really has an all-zero row — if one does not, the call uses the full adjacency.  Same sums, bit for bit.
// (Default: not a degree bound but the very set of vertices whose rows were all zero in the input the plan was built from —
// a graph's stage inputs are the same on every forward, they follow from its weights — which also catches the zero rows of
// low-degree vertices: R-MAT-22's last stage keeps 14 % of the entries instead of 25 %.)
// heavy_bits: bit v = row v of feat is all zero (one wave per 64 vertices: two words)
__global__ __launch_bounds__(256) void k_prune_mark_zero(const float4 *__restrict__ feat, uint32_t n, uint32_t *__restrict__ heavy_bits) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;   // (the grid covers n rounded up to 64)
    bool zero = false;
    if (u < n) {
        const float4 a = feat[(size_t)u * 4], b = feat[(size_t)u * 4 + 1], c = feat[(size_t)u * 4 + 2], d = feat[(size_t)u * 4 + 3];
        zero = !(a.x != 0.f || a.y != 0.f || a.z != 0.f || a.w != 0.f || b.x != 0.f || b.y != 0.f || b.z != 0.f || b.w != 0.f ||
                 c.x != 0.f || c.y != 0.f || c.z != 0.f || c.w != 0.f || d.x != 0.f || d.y != 0.f || d.z != 0.f || d.w != 0.f);
    }
    const unsigned long long m = __ballot(zero);
    const uint32_t w0 = (u & ~63u) >> 5, words = (n + 31) / 32;
    if ((threadIdx.x & 63) == 0 && w0 < words) heavy_bits[w0] = (uint32_t)m;
    if ((threadIdx.x & 63) == 1 && w0 + 1 < words) heavy_bits[w0 + 1] = (uint32_t)(m >> 32);
}

// heavy_bits: bit v = row v of the F = 1 stage's OUTPUT is PREDICTED to be all zero — from the graph alone, when it is handed
// over (round 4).  The reference's driver feeds x[u] = (float)W(u) / ws (src/GNN_VC.cpp:189-191), so the stage's first-layer
// input of vertex u is [sum of its neighbours' x, x[u], degree, W/ws, NW/ws] with the sum ~ NW/ws (the same weights added
// as integers instead of as rounded floats).  The kernel runs the stage's dense layers on that input and sets the bit when
// every output is below zero BY A MARGIN before the last ReLU (the margin covers the sum's rounding: a row that close to the
// kink is simply left out of the set).  A prediction, never a proof: every call proves on the device that the set's rows are
// all zero in ITS input (k_prune_check) and takes the full adjacency if one is not — an input other than W/ws costs time,
// never a bit.  Vertices without entries are left out (nothing points to them on a symmetric adjacency).
template <int N1, int N2, int N3>
__global__ __launch_bounds__(256) void k_predict_zero_f1(GraphDev g, float ws, const float *__restrict__ P, uint32_t *__restrict__ heavy_bits) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;   // (the grid covers n rounded up to 64)
    const uint32_t uc = u < g.n ? u : g.n - 1;
    const uint32_t deg = g.rowptr[uc + 1] - g.rowptr[uc];
    bool zero = false;
    if (__any(u < g.n && deg != 0)) {   // (uniform per wave: R-MAT's runs of isolated vertices cost nothing)
        float x0[5];
        x0[1] = (float)g.w[uc] / ws;
        x0[2] = (float)deg;
        x0[3] = x0[1];
        x0[4] = (float)g.nw[uc] / ws;
        x0[0] = x0[4];
        const float *W1 = P, *b1 = W1 + 5 * N1;
        const float *W2 = b1 + N1, *b2 = W2 + N1 * N2;
        const float *W3 = b2 + N2, *b3 = W3 + N2 * N3;
        float x1[N1], x2[N2], x3[N3];
        dense<5, 5, N1, 0>(x0, x1, W1, b1);
        dense<N1, N1, N2, 0>(x1, x2, W2, b2);
        dense<N2, N2, N3, 1>(x2, x3, W3, b3);   // (no ReLU: the margin is taken on the pre-activations)
        float top = x3[0], scale = 0.0f;
#pragma unroll
        for (int j = 0; j < N3; ++j) {
            top = x3[j] > top ? x3[j] : top;
            scale = fabsf(x3[j]) > scale ? fabsf(x3[j]) : scale;
        }
#pragma unroll
        for (int k = 0; k < N2; ++k) scale = x2[k] > scale ? x2[k] : scale;
        zero = u < g.n && deg != 0 && top <= -1e-3f * (1.0f + scale);   // (a NaN compares false: not in the set)
    }
    const unsigned long long m = __ballot(zero);
    const uint32_t w0 = (u & ~63u) >> 5, words = (g.n + 31) / 32;
    if ((threadIdx.x & 63) == 0 && w0 < words) heavy_bits[w0] = (uint32_t)m;
    if ((threadIdx.x & 63) == 1 && w0 + 1 < words) heavy_bits[w0 + 1] = (uint32_t)(m >> 32);
}

// Filtered gather (k_stage_f16<.., FILTER>, k_long_f16<.., FILTER>, k_giant_gather16): the bitmap of this input's all-zero rows
// over n + 1 vertices (the pad row's bit is 0) and, in info (zeroed by the launcher), {the degrees of those vertices (rows this
// engine holds), their number}.
// prev_bits (may be null): the set an earlier stage of this forward filtered with (prev_info: its info) — info[2] becomes
// non-zero unless that set was worth filtering with AND every vertex of it has an all-zero row in feat too: only then do the
// lists that stage left (GraphDev::short_col) stand for the adjacency in this call.
__global__ __launch_bounds__(256) void k_filter_mark(GraphDev g, const float4 *__restrict__ feat, uint32_t *__restrict__ bits,
                                                     unsigned long long *__restrict__ info, const uint32_t *__restrict__ prev_bits,
                                                     const unsigned long long *__restrict__ prev_info) {
    // (a block walks groups of 256 rows with the grid's stride: two atomics per BLOCK at the end — one per group of 256 rows
    // made 32 K atomics on two addresses of R-MAT-22's 4 M rows, 0.18 - 0.26 ms for a 0.05 ms pass)
    const uint32_t words = g.n / 32 + 1, groups = (g.n + 1 + 255) / 256;
    unsigned long long deg = 0;
    uint32_t members = 0;
    bool miss = false;
    for (uint32_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const uint32_t u = grp * 256 + threadIdx.x;
        bool zero = false;
        if (u < g.n) {
            const float4 a = feat[(size_t)u * 4], b = feat[(size_t)u * 4 + 1], c = feat[(size_t)u * 4 + 2], d = feat[(size_t)u * 4 + 3];
            zero = !(a.x != 0.f || a.y != 0.f || a.z != 0.f || a.w != 0.f || b.x != 0.f || b.y != 0.f || b.z != 0.f || b.w != 0.f ||
                     c.x != 0.f || c.y != 0.f || c.z != 0.f || c.w != 0.f || d.x != 0.f || d.y != 0.f || d.z != 0.f || d.w != 0.f);
        }
        const unsigned long long m = __ballot(zero);
        const uint32_t w0 = (u & ~63u) >> 5;
        if ((threadIdx.x & 63) == 0 && w0 < words) bits[w0] = (uint32_t)m;
        if ((threadIdx.x & 63) == 1 && w0 + 1 < words) bits[w0 + 1] = (uint32_t)(m >> 32);
        if (zero) {
            ++members;
            if (u >= g.lo() && u < g.hi()) deg += g.rowptr[u + 1] - g.rowptr[u];
        } else if (prev_bits != nullptr && u < g.n && (prev_bits[u >> 5] >> (u & 31) & 1u)) {
            miss = true;
        }
    }
    if (prev_bits != nullptr) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && !filter_worth_info(g, prev_info)) miss = true;
        if (__any(miss) && (threadIdx.x & 63) == 0) atomicOr(reinterpret_cast<unsigned int *>(info + 2), 1u);
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        deg += ((unsigned long long)__shfl_xor((unsigned)(deg >> 32), off) << 32) | __shfl_xor((unsigned)deg, off);
        members += __shfl_xor(members, off);
    }
    __shared__ unsigned long long part[4][2];
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6][0] = deg;
        part[threadIdx.x >> 6][1] = members;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long d = part[0][0] + part[1][0] + part[2][0] + part[3][0];
        const unsigned long long c = part[0][1] + part[1][1] + part[2][1] + part[3][1];
        if (d) atomicAdd(info, d);
        if (c) atomicAdd(info + 1, c);
    }
}

// mass[0] += the degrees of the set's vertices (on a symmetric adjacency: the entries that point to them; whole graphs only —
// a slice does not hold the other rows' degrees), mass[1] += the number of vertices in the set
__global__ __launch_bounds__(256) void k_prune_mass(GraphDev g, const uint32_t *__restrict__ heavy_bits, unsigned long long *__restrict__ mass,
                                                    const uint32_t *__restrict__ prev_bits) {
    // prev_bits (may be null): another stage's set — mass[2] += its vertices that are NOT in this one (0 = this set contains it)
    unsigned long long mine = 0;
    uint32_t members = 0, outside = 0;
    const bool degrees = !g.sliced_dev();
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < g.n; u += gridDim.x * blockDim.x) {
        const bool in = heavy_bits[u >> 5] >> (u & 31) & 1u;
        if (in) {
            if (degrees) mine += g.rowptr[u + 1] - g.rowptr[u];
            ++members;
        } else if (prev_bits && (prev_bits[u >> 5] >> (u & 31) & 1u)) {
            ++outside;
        }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        mine += ((unsigned long long)__shfl_xor((unsigned)(mine >> 32), off) << 32) | __shfl_xor((unsigned)mine, off);
        members += __shfl_xor(members, off);
        outside += __shfl_xor(outside, off);
    }
    __shared__ unsigned long long part[4][3];
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6][0] = mine;
        part[threadIdx.x >> 6][1] = members;
        part[threadIdx.x >> 6][2] = outside;
    }
    __syncthreads();
    if (threadIdx.x == 0) {   // (one set of atomics per block: per wave they queue up behind each other)
        unsigned long long m[3];
        for (int k = 0; k < 3; ++k) m[k] = part[0][k] + part[1][k] + part[2][k] + part[3][k];
        for (int k = 0; k < 3; ++k)
            if (m[k]) atomicAdd(mass + k, m[k]);
    }
}

// Building the pruned CSR: the engine's entries taken FLAT, in chunks of 64 (one wave trip), whatever rows they belong to —
// a skewed graph's work is then balanced by construction (a first version walked 64-row tiles: R-MAT's hubs share a few
// tiles, 19 ms per pass on R-MAT-22).  Kept entries keep their flat order, which is the CSR order of every row.
//   k_prune_chunks   mask[k] = which of chunk k's 64 entries are kept, cnt[k] = how many      (then: exclusive scan of cnt)
//   k_prune_fill     pcol[off[k] + rank] = the kept entries
//   k_prune_offsets  prp[u] = kept entries before row u's first entry  (u = the engine's rows and one past them)
__global__ __launch_bounds__(256) void k_prune_chunks(const uint32_t *__restrict__ col, uint32_t nnz, const uint32_t *__restrict__ heavy_bits,
                                                      unsigned long long *__restrict__ mask, uint32_t *__restrict__ cnt) {
    const uint32_t lane = threadIdx.x & 63, nchunks = (nnz + 63) / 64;
    constexpr uint32_t U = 4;
    for (uint32_t k0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * U; k0 < nchunks; k0 += gridDim.x * 4 * U) {
        uint32_t c[U], w[U];
#pragma unroll
        for (uint32_t t = 0; t < U; ++t) {
            const uint32_t e = (k0 + t) * 64 + lane;
            c[t] = e < nnz ? col[e] : 0u;
        }
#pragma unroll
        for (uint32_t t = 0; t < U; ++t) w[t] = heavy_bits[c[t] >> 5];
#pragma unroll
        for (uint32_t t = 0; t < U; ++t) {
            const uint32_t e = (k0 + t) * 64 + lane;
            const unsigned long long km = __ballot(e < nnz && !(w[t] >> (c[t] & 31) & 1u));
            if (lane == 0 && k0 + t < nchunks) {
                mask[k0 + t] = km;
                cnt[k0 + t] = (uint32_t)__popcll(km);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_prune_fill(const uint32_t *__restrict__ col, uint32_t nnz, const unsigned long long *__restrict__ mask,
                                                    const uint32_t *__restrict__ off, uint32_t *__restrict__ pcol) {
    const uint32_t lane = threadIdx.x & 63, nchunks = (nnz + 63) / 64;
    for (uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6); k < nchunks; k += gridDim.x * 4) {
        const unsigned long long km = mask[k];
        const uint32_t e = k * 64 + lane;
        if (km >> lane & 1ull) pcol[off[k] + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = col[e];
    }
}

// (a slice: rowptr / prp are biased by the caller, indexed by global row id; entries count from the slice's first)
__global__ __launch_bounds__(256) void k_prune_offsets(GraphDev g, const unsigned long long *__restrict__ mask, const uint32_t *__restrict__ off,
                                                       uint32_t nchunks, uint32_t *__restrict__ prp) {
    const uint32_t u = g.lo() + blockIdx.x * blockDim.x + threadIdx.x;
    if (u > g.hi()) return;
    const uint32_t e = g.rowptr[u], k = e >> 6;
    // (e == nnz on a chunk boundary: off[nchunks] = the total, and no mask word to read)
    prp[u] = off[k] + (k < nchunks ? (uint32_t)__popcll(mask[k] & ((1ull << (e & 63u)) - 1ull)) : 0u);
}

// *bad |= 1 if a heavy vertex has a non-zero among its 16 values
__global__ __launch_bounds__(256) void k_prune_check(GraphDev g, const float4 *__restrict__ feat, const uint32_t *__restrict__ heavy_bits,
                                                     uint32_t *__restrict__ bad) {
    bool miss = false;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < g.n; u += gridDim.x * blockDim.x) {
        if (!(heavy_bits[u >> 5] >> (u & 31) & 1u)) continue;
        const float4 a = feat[(size_t)u * 4], b = feat[(size_t)u * 4 + 1], c = feat[(size_t)u * 4 + 2], d = feat[(size_t)u * 4 + 3];
        miss |= a.x != 0.f || a.y != 0.f || a.z != 0.f || a.w != 0.f || b.x != 0.f || b.y != 0.f || b.z != 0.f || b.w != 0.f ||
                c.x != 0.f || c.y != 0.f || c.z != 0.f || c.w != 0.f || d.x != 0.f || d.y != 0.f || d.z != 0.f || d.w != 0.f;
    }
    if (__any(miss) && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
}

// ------------------------------------------------------------------ degree-sorted tile order
// (built once per graph and row range when natural tiles would waste most of their rounds)
// lockstep cost of natural tiles: sum over tiles of the tile's largest (non-long) degree
__global__ __launch_bounds__(256) void k_tile_waste(GraphDev g, uint32_t row_lo, uint32_t row_hi,
                                                    uint32_t long_thresh, unsigned long long *__restrict__ sum_max,
                                                    uint32_t heavy_from) {
    // sum_max[0] += the tiles' largest (non-long) degrees; sum_max[1] += the entries of the non-long rows of at least heavy_from
    // entries (how much of the graph sits in a heavy tail).  One wave = one 64-row tile per trip; per-wave sums, one atomic per
    // block (every wave hitting the same 8 bytes with its own atomic cost 1.9 ms on the metric graph)
    __shared__ unsigned long long part[4], tail[4];
    unsigned long long mine = 0, heavy = 0;
    const uint32_t ntiles = (row_hi - row_lo + 63) / 64;
    for (uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += gridDim.x * 4) {
        const uint32_t u = row_lo + t * 64 + (threadIdx.x & 63);
        uint32_t d = 0;
        if (u < row_hi) {
            d = g.rowptr[u + 1] - g.rowptr[u];
            if (d >= long_thresh) d = 0;
        }
        uint32_t h = d >= heavy_from ? d : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_xor(d, off);
            d = o > d ? o : d;
            h += __shfl_xor(h, off);
        }
        mine += d;
        heavy += h;
    }
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6] = mine;
        tail[threadIdx.x >> 6] = heavy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long s = part[0] + part[1] + part[2] + part[3], h = tail[0] + tail[1] + tail[2] + tail[3];
        if (s) atomicAdd(sum_max, s);
        if (h) atomicAdd(sum_max + 1, h);
    }
}

// Degree class of a row in the sorted lists: the degree itself below 2048, eight degrees to a class above (so that
// thresholds of up to 16 K fit the kernels' 4096 LDS counters; the order inside a class is free)
__device__ __forceinline__ uint32_t deg_class(uint32_t d, uint32_t bins) {
    const uint32_t c = d < 2048u ? d : 2048u + ((d - 2048u) >> 3);
    return c < bins ? c : bins - 1;
}

// histogram of the degree classes over the non-long rows of [row_lo, row_hi)
__global__ __launch_bounds__(256) void k_deg_hist(GraphDev g, uint32_t row_lo, uint32_t row_hi, uint32_t long_thresh,
                                                  uint32_t bins, uint32_t *__restrict__ hist, const uint32_t *__restrict__ skip_rowptr,
                                                  uint32_t skip_from) {
    __shared__ uint32_t local[4096];   // bins <= 4096: a few degree classes take most rows, so count per block first
    for (uint32_t i = threadIdx.x; i < bins; i += blockDim.x) local[i] = 0;
    __syncthreads();
    for (uint32_t u = row_lo + blockIdx.x * blockDim.x + threadIdx.x; u < row_hi; u += gridDim.x * blockDim.x) {
        const uint32_t d = g.rowptr[u + 1] - g.rowptr[u];
        if (skip_rowptr && skip_rowptr[u + 1] - skip_rowptr[u] >= skip_from) continue;
        if (d < long_thresh) atomicAdd(&local[deg_class(d, bins)], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bins; i += blockDim.x)
        if (local[i]) atomicAdd(&hist[i], local[i]);
}

// hist[d] (rows of degree class d) -> in place, the first slot of class d with the classes laid out by DEscending degree;
// info[0] = rows listed, info[1] = rows of class 0 (no entry).  One wave; bins <= 4096.
__global__ __launch_bounds__(64) void k_deg_starts(uint32_t *__restrict__ hist, uint32_t bins, uint32_t *__restrict__ info) {
    const uint32_t lane = threadIdx.x, per = (bins + 63u) / 64u;
    // lane l owns classes [hi - per, hi) counted from the top: lane 0 the heaviest
    const uint32_t top = bins > lane * per ? bins - lane * per : 0u, bot = top > per ? top - per : 0u;
    uint32_t mine = 0;
    for (uint32_t d = bot; d < top; ++d) mine += hist[d];
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if ((int)lane >= off) incl += t;
    }
    const uint32_t zero_rows = hist[0];
    const uint32_t total = __shfl(incl, 63);
    __builtin_amdgcn_s_waitcnt(0);   // (hist[0] above is read before any lane overwrites it: one wave, program order)
    uint32_t run = incl - mine;
    for (uint32_t d = top; d-- > bot;) {
        const uint32_t h = hist[d];
        hist[d] = run;
        run += h;
    }
    if (lane == 0) {
        info[0] = total;
        info[1] = zero_rows;
    }
}

// cursor[d] holds the next free slot of degree class d (classes laid out by DEscending degree,
// so the heaviest tiles are dispatched first); entries carry what the tile kernel needs per vertex
constexpr int kDegScatterRows = 8;   // rows per thread: the two sweeps over the class counters are per BLOCK (4096 counters
                                     // for 256 rows made the pass 0.22 ms on R-MAT-22's 4 M rows)
__global__ __launch_bounds__(256) void k_deg_scatter(GraphDev g, uint32_t row_lo, uint32_t row_hi, uint32_t long_thresh,
                                                     uint32_t bins, uint32_t *__restrict__ cursor,
                                                     uint32_t *__restrict__ vertex, uint4 *__restrict__ meta,
                                                     const uint32_t *__restrict__ skip_rowptr, uint32_t skip_from) {
    // per block: count its rows per class in LDS, reserve one range per class with a single
    // global atomic, then hand out slots from LDS (global atomics: one per class per block)
    __shared__ uint32_t local[4096];
    for (uint32_t i = threadIdx.x; i < bins; i += blockDim.x) local[i] = 0;
    __syncthreads();
    const uint32_t u0 = row_lo + blockIdx.x * (blockDim.x * kDegScatterRows) + threadIdx.x;
    uint32_t rs[kDegScatterRows], re[kDegScatterRows], cls[kDegScatterRows], rank_in_block[kDegScatterRows];
#pragma unroll
    for (int k = 0; k < kDegScatterRows; ++k) {
        const uint32_t u = u0 + k * blockDim.x;
        cls[k] = 0xFFFFFFFFu;
        rs[k] = re[k] = rank_in_block[k] = 0;
        if (u < row_hi) {
            rs[k] = g.rowptr[u];
            re[k] = g.rowptr[u + 1];
            const uint32_t d = re[k] - rs[k];
            const bool skip = skip_rowptr && skip_rowptr[u + 1] - skip_rowptr[u] >= skip_from;
            if (d < long_thresh && !skip) {
                cls[k] = deg_class(d, bins);
                rank_in_block[k] = atomicAdd(&local[cls[k]], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bins; i += blockDim.x)
        if (local[i]) local[i] = atomicAdd(&cursor[i], local[i]);   // now the block's base slot of class i
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kDegScatterRows; ++k) {
        if (cls[k] != 0xFFFFFFFFu) {
            const uint32_t u = u0 + k * blockDim.x, slot = local[cls[k]] + rank_in_block[k];
            vertex[slot] = u;
            meta[slot] = make_uint4(rs[k], re[k], g.w[u], g.nw[u]);
        }
    }
}

// ------------------------------------------------------------------ column-blocked F = 1 aggregation
// The F = 1 gather reads 4 useful bytes per random 128-byte line of x; with x
// (4 N bytes) far beyond the 4 MiB L2 of an XCD every read goes to the fabric.
// Column blocking makes it cache-resident: vertices are cut into blocks of `wb`
// columns (an x slice of 4 wb bytes that stays in every XCD's L2), the CSR is
// re-bucketed once per graph into block-major order (colb, with entry pointers
// bp[k*N + u]), and one launch per block adds that block's entries to a running
// sum per row.  Because every row's entries are visited in their stored order
// (block ids are non-decreasing along a row — checked when the index is built —
// and entries keep their order inside a block), the fp32 add sequence per row is
// exactly the unblocked one: results are bit-identical.

// pass 1: cnt[k*N + u] = entries of row u in block k (buffer pre-zeroed); flags a
// row whose block ids decrease (then the blocked plan is not exact and is not used).
__global__ void k_blk_count(GraphDev g, uint32_t wb, uint32_t long_thresh, uint32_t *__restrict__ cnt,
                            uint32_t *__restrict__ bad) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= g.n) return;
    const uint32_t rs = g.rowptr[u], re = g.rowptr[u + 1];
    if (re - rs >= long_thresh) return;   // long rows are summed by k_long_f1, not per block
    uint32_t kcur = 0, run = 0;
    for (uint32_t e = rs; e < re; ++e) {
        const uint32_t k = g.col[e] / wb;
        if (k != kcur) {
            if (k < kcur) atomicOr(bad, 1u);
            if (run) cnt[(size_t)kcur * g.n + u] = run;
            kcur = k;
            run = 0;
        }
        ++run;
    }
    if (run) cnt[(size_t)kcur * g.n + u] = run;
}

// pass 3: scatter the entries to block-major order (bp = exclusive scan of cnt).
__global__ void k_blk_scatter(GraphDev g, uint32_t wb, uint32_t long_thresh,
                              const uint32_t *__restrict__ bp, uint32_t *__restrict__ colb) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= g.n) return;
    const uint32_t rs = g.rowptr[u], re = g.rowptr[u + 1];
    if (re - rs >= long_thresh) return;
    uint32_t kcur = 0xFFFFFFFFu, dst = 0;
    for (uint32_t e = rs; e < re; ++e) {
        const uint32_t c = g.col[e];
        const uint32_t k = c / wb;
        if (k != kcur) {
            kcur = k;
            dst = bp[(size_t)k * g.n + u];
        }
        colb[dst++] = c;
    }
}

// exclusive scan of uint32 (three-kernel chunked scan; chunk = 256 threads x 16)
constexpr int kScanPer = 16, kScanChunk = 256 * kScanPer;
__global__ __launch_bounds__(256) void k_scan_chunks(uint32_t *__restrict__ data, size_t n,
                                                     uint32_t *__restrict__ sums) {
    __shared__ uint32_t part[256];
    const size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanPer;
    uint32_t v[kScanPer], tot = 0;
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) {
        v[i] = (base + i < n) ? data[base + i] : 0u;
        tot += v[i];
    }
    part[threadIdx.x] = tot;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {   // Hillis-Steele inclusive scan of the 256 partials
        const uint32_t t = (threadIdx.x >= (unsigned)off) ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - tot;      // exclusive prefix of this thread inside the chunk
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) {
        if (base + i < n) data[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == 255) sums[blockIdx.x] = part[255];
}
__global__ void k_scan_add(uint32_t *__restrict__ data, size_t n, const uint32_t *__restrict__ offs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] += offs[i / kScanChunk];
}

// one launch per column block except the last: acc[u] (+)= sum of x over row u's entries in
// block k, in stored order.  A row has only a few entries per block; the first four are
// fetched together (index loads, then x loads, in parallel) so a thread pays two memory
// latencies instead of two per entry.
__global__ __launch_bounds__(256) void k_blk_accumulate(const uint32_t *__restrict__ bpk,
                                                        const uint32_t *__restrict__ colb,
                                                        const float *__restrict__ xin,
                                                        float *__restrict__ acc, uint32_t row_lo,
                                                        uint32_t row_hi, int first) {
    const uint32_t u = row_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= row_hi) return;
    const uint32_t s = bpk[u], t = bpk[u + 1];
    float a = first ? 0.0f : acc[u];
    if (s == t) {
        if (first) acc[u] = a;
        return;
    }
    constexpr int U = 4;
    uint32_t c[U];
    float v[U];
#pragma unroll
    for (int i = 0; i < U; ++i) c[i] = colb[s + i];            // colb is padded past nnz
#pragma unroll
    for (int i = 0; i < U; ++i) v[i] = xin[(s + i < t) ? c[i] : c[0]];
#pragma unroll
    for (int i = 0; i < U; ++i) a += (s + i < t) ? v[i] : 0.0f;  // a is never -0.0f: + 0.0f is exact
    for (uint32_t e = s + U; e < t; ++e) a += xin[colb[e]];
    acc[u] = a;
}

// ------------------------------------------------------------------ reduction-rule predicates (f-2)
// One vertex-parallel pass over the device CSR that evaluates, for every vertex the
// reference's reduce_graph would look at (D(u) <= max_degree, include/mwvc_reductions.hpp:344),
// which of its local rules would fire on the graph as it stands — so the host loop can skip
// the (vast majority of) vertices where nothing fires and keep applying reductions in the
// reference's own order.  Pure predicates, bit for bit the reference's:
//   bit 0 neighborhood_reduction (:131-139)   bit 1 twin_fold (:141-160)
//   bit 2 domination_reduction   (:162-177)   bit 3 isolated_fold (:270-284)
//   bit 4 independent_fold       (:246-268)   bits 5, 6: the small-solver rules, "host decides"
__device__ __forceinline__ bool dev_is_dominating(const GraphDev &g, uint32_t u, uint32_t v) {
    // reduction_graph::is_dominating (include/reduction_graph.hpp:201-224), same merge loop
    const uint32_t us = g.rowptr[u], ue = g.rowptr[u + 1], vs = g.rowptr[v], ve = g.rowptr[v + 1];
    if (ue - us < ve - vs || (uint32_t)(g.w[u] + g.nw[u]) < (uint32_t)(g.w[v] + g.nw[v])) return false;
    uint32_t f1 = us, f2 = vs;
    while (f2 != ve) {
        if (g.col[f2] == u) {
            ++f2;
            if (f2 == ve) break;
        }
        if (f1 == ue || g.col[f2] < g.col[f1]) return false;
        if (!(g.col[f1] < g.col[f2])) ++f2;
        ++f1;
    }
    return true;
}

__device__ __forceinline__ bool dev_is_twin(const GraphDev &g, uint32_t u, uint32_t v) {
    // reduction_graph::is_twin (include/reduction_graph.hpp:180-186)
    const uint32_t us = g.rowptr[u], vs = g.rowptr[v];
    const uint32_t d = g.rowptr[u + 1] - us;
    if (d != g.rowptr[v + 1] - vs || g.nw[u] != g.nw[v] || u == v) return false;
    for (uint32_t i = 0; i < d; ++i)
        if (g.col[us + i] != g.col[vs + i]) return false;
    return true;
}

// What the driver's sort and selection loop read from the scores (reference src/GNN_VC.cpp:194-206,
// 213, 220): key = std::min(s, 1.0f - s) (the second argument only if it is smaller), class = s > 0.5f.
__global__ __launch_bounds__(256) void k_score_keys(const float *__restrict__ scores, size_t n, float *__restrict__ keys,
                                                    uint8_t *__restrict__ above_half) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = scores[i], t = 1.0f - s;
    keys[i] = t < s ? t : s;
    above_half[i] = s > 0.5f ? 1 : 0;
}

// ---- feature-row codec of the inter-GPU exchange ----------------------------------------------
// After the ReLU that ends a stage many of the 16 output columns are zero in every row (dead
// units; which ones depends on the graph).  The ranks exchange only the live columns: rows are
// packed to `kp` floats (the live columns in ascending order, zero-padded), shipped, and expanded
// back to 16 columns with +0.0f in the dead ones — lossless as long as the dead columns really
// are zero, which the pack kernel verifies on every row it ships (flag).

// bit c of *mask: some row of feat[rows x 16] is non-zero in column c
__global__ __launch_bounds__(256) void k_live_columns(const float4 *__restrict__ feat, size_t quads, uint32_t *mask) {
    uint32_t bits = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = feat[i];
        const uint32_t b = (v.x != 0.0f ? 1u : 0u) | (v.y != 0.0f ? 2u : 0u) | (v.z != 0.0f ? 4u : 0u) |
                           (v.w != 0.0f ? 8u : 0u);
        bits |= b << (4 * (i & 3));
    }
    for (int off = 32; off; off >>= 1) bits |= __shfl_xor(bits, off);
    if ((threadIdx.x & 63) == 0 && bits) atomicOr(mask, bits);
}

// counts[c] += rows of feat[rows x 16] that are non-zero in column c
__global__ __launch_bounds__(256) void k_column_counts(const float4 *__restrict__ feat, size_t quads,
                                                       unsigned long long *counts) {
    __shared__ uint32_t part[4][16];
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;   // this lane always sees the same column group (stride % 4 == 0)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = feat[i];
        c0 += v.x != 0.0f;
        c1 += v.y != 0.0f;
        c2 += v.z != 0.0f;
        c3 += v.w != 0.0f;
    }
    for (int off = 32; off >= 4; off >>= 1) {
        c0 += __shfl_xor(c0, off);
        c1 += __shfl_xor(c1, off);
        c2 += __shfl_xor(c2, off);
        c3 += __shfl_xor(c3, off);
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 4) {
        part[wave][4 * lane + 0] = c0;
        part[wave][4 * lane + 1] = c1;
        part[wave][4 * lane + 2] = c2;
        part[wave][4 * lane + 3] = c3;
    }
    __syncthreads();
    if (threadIdx.x < 16) {   // one atomic per column and block (per wave they queued up behind each other)
        const uint32_t t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (t) atomicAdd(&counts[threadIdx.x], (unsigned long long)t);
    }
}

struct ColumnMap {
    uint8_t col[16];   // col[j] = j-th live column (j < k)
};

// dense[(row - row_lo) * kp + j] = feat[row][col[j]] (0 for j >= k), rows [row_lo, row_hi); one thread
// per packed element.  The first four threads of a row also look at its other columns: a non-zero
// there goes to the exception list exc (word 0 = count, entries of 4 words {row - row_lo, column,
// value bits, 0} from word 4 on, room for `cap`) — or, without a list / when it is full, raises flag
// bit 0 / bit 1 and the caller falls back to full rows.
__global__ __launch_bounds__(256) void k_pack_rows(const float *__restrict__ feat, uint32_t row_lo, uint32_t row_hi,
                                                   uint32_t mask, uint32_t k, uint32_t kp, ColumnMap map,
                                                   float *__restrict__ dense, uint32_t *__restrict__ exc, uint32_t cap,
                                                   uint32_t *flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)(row_hi - row_lo) * kp;
    uint32_t bad = 0;
    if (i < total) {
        const uint32_t rel = (uint32_t)(i / kp), j = (uint32_t)(i % kp);
        const size_t row = (size_t)row_lo + rel;
        dense[i] = j < k ? feat[row * 16 + map.col[j]] : 0.0f;
        if (j < 4) {
            const float4 v = reinterpret_cast<const float4 *>(feat)[row * 4 + j];
            const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t c = 4 * j + t;
                if (f[t] != 0.0f && !(mask >> c & 1u)) {
                    if (!exc) {
                        bad |= 1u;
                    } else {
                        const uint32_t slot = atomicAdd(&exc[0], 1u);
                        if (slot < cap) {
                            reinterpret_cast<uint4 *>(exc)[1 + slot] = make_uint4(rel, c, __float_as_uint(f[t]), 0u);
                        } else {
                            bad |= 2u;
                        }
                    }
                }
            }
        }
    }
    for (int off = 32; off; off >>= 1) bad |= __shfl_xor(bad, off);
    if (bad && (threadIdx.x & 63) == 0) atomicOr(flag, bad);
}

// feat[row][c] = dense column ? dense[(row - row_lo) * kp + rank(c)] : +0.0f, rows [row_lo, row_hi);
// one thread per float4
__global__ __launch_bounds__(256) void k_unpack_rows(const float *__restrict__ dense, uint32_t row_lo, uint32_t row_hi,
                                                     uint32_t mask, uint32_t kp, float *__restrict__ feat) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)(row_hi - row_lo) * 4) return;
    const uint32_t rel = (uint32_t)(i >> 2), q = (uint32_t)(i & 3);
    const float *src = dense + (size_t)rel * kp;
    float v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t c = 4 * q + t;
        v[t] = (mask >> c & 1u) ? src[__popc(mask & ((1u << c) - 1u))] : 0.0f;
    }
    reinterpret_cast<float4 *>(feat)[((size_t)row_lo + rel) * 4 + q] = make_float4(v[0], v[1], v[2], v[3]);
}

// All peers' regions of one all-gathered piece in one launch: blockIdx.y = rank whose region
// (buf + rank * piece_words) holds its rows [rank * per + off, rank * per + off + size) cut at the end
// of its shard and at n; `skip` (this rank, whose rows are already in place) is left alone.
__global__ __launch_bounds__(256) void k_unpack_gathered(const float *__restrict__ buf, uint32_t skip, size_t piece_words,
                                                         uint32_t per, uint32_t off, uint32_t size, uint32_t n,
                                                         uint32_t mask, uint32_t kp, float *__restrict__ feat) {
    const uint32_t peer = blockIdx.y;
    if (peer == skip) return;
    const uint64_t lo64 = (uint64_t)peer * per + off;
    const uint64_t hi64 = min(min(lo64 + size, (uint64_t)(peer + 1) * per), (uint64_t)n);
    if (lo64 >= hi64) return;
    const uint32_t row_lo = (uint32_t)lo64, rows = (uint32_t)(hi64 - lo64);
    const float *dense = buf + (size_t)peer * piece_words;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)rows * 4; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t rel = (uint32_t)(i >> 2), q = (uint32_t)(i & 3);
        const float *src = dense + (size_t)rel * kp;
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t c = 4 * q + t;
            v[t] = (mask >> c & 1u) ? src[__popc(mask & ((1u << c) - 1u))] : 0.0f;
        }
        reinterpret_cast<float4 *>(feat)[((size_t)row_lo + rel) * 4 + q] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__global__ __launch_bounds__(256) void k_unpack_gathered_exceptions(const float *__restrict__ buf, uint32_t skip,
                                                                    size_t piece_words, uint32_t dense_rows, uint32_t cap,
                                                                    uint32_t per, uint32_t off, uint32_t size, uint32_t n,
                                                                    uint32_t kp, float *__restrict__ feat) {
    const uint32_t peer = blockIdx.y;
    if (peer == skip) return;
    const uint64_t lo64 = (uint64_t)peer * per + off;
    const uint64_t hi64 = min(min(lo64 + size, (uint64_t)(peer + 1) * per), (uint64_t)n);
    if (lo64 >= hi64) return;
    const uint32_t *exc = reinterpret_cast<const uint32_t *>(buf + (size_t)peer * piece_words + (size_t)dense_rows * kp);
    const uint32_t count = min(exc[0], cap), rows = (uint32_t)(hi64 - lo64);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint4 en = reinterpret_cast<const uint4 *>(exc)[1 + i];
        if (en.x < rows && en.y < 16u) feat[(lo64 + en.x) * 16 + en.y] = __uint_as_float(en.z);
    }
}

// ---- several devices behind one handle (gnnvc_multi.cpp): a packed piece goes to every peer in ONE launch — the stores land
// in the peers' memory over the fabric (peer access enabled) or, where parts share a device, in local memory — and a receiver
// expands the regions of one piece index from all its peers in one launch.
struct PushDst {
    float *p[64];
};
// region = dense_words floats (a multiple of 4) followed by the exception list {count, -, -, -, entries of 4 words}: the dense
// part and the USED part of the list are copied to destination blockIdx.y, 16 bytes per lane
__global__ __launch_bounds__(256) void k_push_piece(const float *__restrict__ src, size_t dense_words, uint32_t cap, PushDst dst) {
    float *__restrict__ out = dst.p[blockIdx.y];
    const float4 *__restrict__ s4 = reinterpret_cast<const float4 *>(src);
    float4 *__restrict__ d4 = reinterpret_cast<float4 *>(out);
    const size_t quads = dense_words / 4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += stride) d4[i] = s4[i];
    const uint4 *__restrict__ e4 = reinterpret_cast<const uint4 *>(src + dense_words);
    uint4 *__restrict__ o4 = reinterpret_cast<uint4 *>(out + dense_words);
    const uint32_t count = min(reinterpret_cast<const uint32_t *>(src + dense_words)[0], cap);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)count + 1; i += stride) o4[i] = e4[i];
}

struct PieceRef {
    const float *region;
    uint32_t row_lo, row_hi;
};
struct PieceList {
    PieceRef d[64];
};
// blockIdx.y = piece: feat rows [row_lo, row_hi) from its dense part (k_unpack_rows' expansion)
__global__ __launch_bounds__(256) void k_unpack_pieces(PieceList pl, uint32_t mask, uint32_t kp, float *__restrict__ feat) {
    const PieceRef pr = pl.d[blockIdx.y];
    const uint32_t rows = pr.row_hi - pr.row_lo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)rows * 4; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t rel = (uint32_t)(i >> 2), q = (uint32_t)(i & 3);
        const float *src = pr.region + (size_t)rel * kp;
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t c = 4 * q + t;
            v[t] = (mask >> c & 1u) ? src[__popc(mask & ((1u << c) - 1u))] : 0.0f;
        }
        reinterpret_cast<float4 *>(feat)[((size_t)pr.row_lo + rel) * 4 + q] = make_float4(v[0], v[1], v[2], v[3]);
    }
}
__global__ __launch_bounds__(256) void k_unpack_pieces_exceptions(PieceList pl, uint32_t cap, uint32_t kp, float *__restrict__ feat) {
    const PieceRef pr = pl.d[blockIdx.y];
    const uint32_t rows = pr.row_hi - pr.row_lo;
    const uint32_t *exc = reinterpret_cast<const uint32_t *>(pr.region + (size_t)rows * kp);
    const uint32_t count = min(exc[0], cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint4 en = reinterpret_cast<const uint4 *>(exc)[1 + i];
        if (en.x < rows && en.y < 16u) feat[((size_t)pr.row_lo + en.x) * 16 + en.y] = __uint_as_float(en.z);   // (never trust a received index with a store)
    }
}

// the exception list of the same piece, applied after k_unpack_rows (stream order)
__global__ __launch_bounds__(256) void k_unpack_exceptions(const uint32_t *__restrict__ exc, uint32_t cap, uint32_t row_lo,
                                                           uint32_t row_hi, float *__restrict__ feat) {
    const uint32_t count = min(exc[0], cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint4 en = reinterpret_cast<const uint4 *>(exc)[1 + i];
        if (en.x < row_hi - row_lo && en.y < 16u)   // never trust a received index with a store
            feat[((size_t)row_lo + en.x) * 16 + en.y] = __uint_as_float(en.z);
    }
}


// ---- the two reduction rules that run the reference's small exact solver (include/mwvc_reductions.hpp:204-252,
// include/small_solve.hpp:44-74), as exact per-vertex predicates.
// minimum-weight vertex cover of k <= 8 nodes by enumeration: a subset is a cover when every node is in it or has
// all of its neighbours in it; int32 sums like the reference's
__device__ inline int32_t dev_small_mwvc(int k, const int32_t (&wt)[9], const uint32_t (&adj)[9]) {
    int32_t best = 0x7FFFFFFF;
    for (uint32_t sset = 0; sset < (1u << k); ++sset) {
        bool ok = true;
        int32_t cost = 0;
        for (int j = 0; j < k; ++j) {
            if (sset >> j & 1u) cost += wt[j];
            else ok &= (sset & adj[j]) == adj[j];
        }
        if (ok && cost < best) best = cost;
    }
    return best;
}

__device__ inline bool dev_has_edge(const GraphDev &g, uint32_t a, uint32_t b) {   // b in adj(a)?  (ascending lists)
    uint32_t lo = g.rowptr[a], hi = g.rowptr[a + 1];
    const uint32_t end = hi;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g.col[mid] < b) lo = mid + 1;
        else hi = mid;
    }
    return lo < end && g.col[lo] == b;
}

// cover weight of the subgraph induced by nodes[0..k) (edges to vertices outside the list do not exist for the solver)
__device__ inline int32_t dev_induced_mwvc(const GraphDev &g, int k, const uint32_t (&nodes)[9]) {
    int32_t wt[9];
    uint32_t adj[9];
    for (int i = 0; i < k; ++i) {
        wt[i] = (int32_t)g.w[nodes[i]];
        adj[i] = 0;
    }
    for (int i = 0; i < k; ++i)
        for (int j = i + 1; j < k; ++j)
            if (dev_has_edge(g, nodes[i], nodes[j])) {
                adj[i] |= 1u << j;
                adj[j] |= 1u << i;
            }
    return dev_small_mwvc(k, wt, adj);
}

// neighborhood_meta_reduction: D(u) <= 8 and W(u) >= NW(u) - MWVC(G[N(u)])
__device__ inline bool dev_rule_neighborhood_meta(const GraphDev &g, uint32_t u) {
    const uint32_t rs = g.rowptr[u], d = g.rowptr[u + 1] - rs;
    if (d > 8u) return false;
    uint32_t nodes[9];
    uint32_t wmax = 0;
    unsigned long long sum = 0;
    for (uint32_t i = 0; i < d; ++i) {
        nodes[i] = g.col[rs + i];
        const uint32_t wv = g.w[nodes[i]];
        wmax = wv > wmax ? wv : wmax;
        sum += wv;
    }
    // NW(u) - MWVC = weight of the heaviest independent set of N(u): between the heaviest neighbour and NW(u).
    // Only when W(u) falls in between does the cover have to be enumerated (weights below 2^31, like the solver's int32).
    const uint32_t wu = g.w[u], nwu = g.nw[u];
    if (sum == nwu && sum < (1ull << 31)) {   // (no wrap-around anywhere: the bounds below are the reference's arithmetic)
        if (wu < wmax) return false;
        if (wu >= nwu) return true;
    }
    const unsigned long long vc = (unsigned long long)(long long)dev_induced_mwvc(g, (int)d, nodes);
    return (unsigned long long)wu >= (unsigned long long)nwu - vc;
}

// neighbor_meta_reduction, with neighborhood_difference's two peculiarities (gives up once 9 elements are written;
// once adj(u) is exhausted the rest of adj(v) is copied without the "not u itself" test)
__device__ inline bool dev_rule_neighbor_meta(const GraphDev &g, uint32_t u) {
    const uint32_t us = g.rowptr[u], ue = g.rowptr[u + 1], du = ue - us, wu = g.w[u];
    for (uint32_t e = us; e < ue; ++e) {
        const uint32_t v = g.col[e];
        const uint32_t vs = g.rowptr[v], ve = g.rowptr[v + 1], dv = ve - vs, wv = g.w[v];
        if (wv <= wu || (dv > du && dv - du > 8u)) continue;
        uint32_t tmp[9];
        uint32_t t = 0, f1 = vs, f2 = us;
        bool gave_up = false;
        while (f1 != ve && f2 != ue) {
            const uint32_t a = g.col[f1], b = g.col[f2];
            if (a < b) {
                if (a != u) {
                    if (t < 9u) tmp[t] = a;
                    ++t;
                    if (t > 8u) {
                        gave_up = true;
                        break;
                    }
                }
                ++f1;
            } else if (b < a) {
                ++f2;
            } else {
                ++f1;
                ++f2;
            }
        }
        if (!gave_up)
            for (; f1 != ve; ++f1) {
                if (t < 9u) tmp[t] = g.col[f1];
                ++t;
            }
        if (t > 8u) continue;
        uint32_t c = 0, wmax = 0;   // Tw arithmetic
        for (uint32_t i = 0; i < t; ++i) {
            const uint32_t wt = g.w[tmp[i]];
            c += wt;
            wmax = wt > wmax ? wt : wmax;
        }
        // c - MWVC = weight of the heaviest independent set of tmp, between its heaviest member and c: the
        // enumeration is needed only when those bounds leave the comparison open (no wrap-around: sums below 2^31)
        if (c < (1u << 30) && wu < (1u << 30)) {
            if (wmax + wu > wv) continue;
            if (c + wu <= wv) return true;
        }
        const uint32_t vc = (uint32_t)dev_induced_mwvc(g, (int)t, tmp);
        if ((uint32_t)(c - vc + wu) <= wv) return true;
    }
    return false;
}

__global__ __launch_bounds__(256) void k_reduction_flags(GraphDev g, uint32_t max_degree,
                                                         uint8_t *__restrict__ flags) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= g.n) return;
    const uint32_t rs = g.rowptr[u], d = g.rowptr[u + 1] - rs;
    uint32_t f = 0;
    if (d <= max_degree) {
        const uint32_t wu = g.w[u], nwu = g.nw[u];
        if (dev_rule_neighbor_meta(g, u)) f |= 32u;
        if (dev_rule_neighborhood_meta(g, u)) f |= 64u;
        if (nwu <= wu) f |= 1u;
        if (d > 0) {
            const uint32_t last = g.col[rs + d - 1];
            for (uint32_t e = g.rowptr[last], ee = g.rowptr[last + 1]; e < ee; ++e) {
                const uint32_t v = g.col[e];
                if (v != u && dev_is_twin(g, u, v)) { f |= 2u; break; }
            }
            bool all = true;
            uint32_t wmin = g.w[g.col[rs]];
            for (uint32_t i = 0; i < d; ++i) {
                const uint32_t v = g.col[rs + i];
                const uint32_t wv = g.w[v];
                if (!(f & 4u) && ((wv >= wu && dev_is_dominating(g, u, v)) || (wv <= wu && dev_is_dominating(g, v, u))))
                    f |= 4u;
                if (all && !dev_is_dominating(g, v, u)) all = false;
                wmin = wv < wmin ? wv : wmin;
            }
            if (all) f |= 8u;
            if (wu >= (uint32_t)(nwu - wmin)) f |= 16u;
        } else {
            f |= 8u;   // all_of over no neighbours
        }
    }
    flags[u] = (uint8_t)f;
}

// ------------------------------------------------------------------ layer-by-layer kernels
// One thread per output element; exact, simple, used for models that do not
// match a fused plan and for the layer-level ABI.
__global__ void k_graph_layer(GraphDev g, float ws, uint32_t f, const float *__restrict__ in,
                              float *__restrict__ out) {
    const uint32_t wd = 2 * f + 3;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)g.n * wd) return;
    const uint32_t u = (uint32_t)(gid / wd), j = (uint32_t)(gid % wd);
    const uint32_t rs = g.rowptr[u], re = g.rowptr[u + 1];
    float v = 0.0f;
    if (j < f) {
        for (uint32_t e = rs; e < re; ++e) v = v + in[(size_t)g.col[e] * f + j];
    } else if (j < 2 * f) {
        v = in[(size_t)u * f + (j - f)];
    }
    // written last in the reference, so they win over the copied columns
    if (j == f + 1) v = (float)(re - rs);
    if (j == f + 2) v = (float)g.w[u] / ws;
    if (j == f + 3) v = (float)g.nw[u] / ws;
    out[gid] = v;
}

__global__ void k_linear(uint32_t n, uint32_t k, uint32_t m, const float *__restrict__ in,
                         const float *__restrict__ W, const float *__restrict__ bias,
                         float *__restrict__ out) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n * m) return;
    const uint32_t i = (uint32_t)(gid / m), j = (uint32_t)(gid % m);
    float acc = 0.0f;
    for (uint32_t kk = 0; kk < k; ++kk)
        acc = __builtin_fmaf(in[(size_t)i * k + kk], W[(size_t)kk * m + j], acc);
    out[gid] = acc + bias[j];
}

__global__ void k_relu(size_t count, const float *__restrict__ in, float *__restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        out[i] = relu_ref(in[i]);
}

__global__ void k_sigmoid(size_t count, const float *__restrict__ in, float *__restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        out[i] = sigmoid_ref(in[i]);
}

__global__ void k_sgemm(int ta, int tb, uint32_t m, uint32_t n, uint32_t k,
                        const float *__restrict__ A, uint32_t lda, const float *__restrict__ B,
                        uint32_t ldb, float beta, float *__restrict__ C, uint32_t ldc) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)m * n) return;
    const uint32_t i = (uint32_t)(gid / n), j = (uint32_t)(gid % n);
    float acc = 0.0f;
    for (uint32_t kk = 0; kk < k; ++kk) {
        const float a = ta ? A[(size_t)kk * lda + i] : A[(size_t)i * lda + kk];
        const float b = tb ? B[(size_t)j * ldb + kk] : B[(size_t)kk * ldb + j];
        acc = __builtin_fmaf(a, b, acc);
    }
    float *c = &C[(size_t)i * ldc + j];
    *c = (beta == 0.0f) ? acc : __builtin_fmaf(beta, *c, acc);
}

// graph sanity on the device (after upload): bit0 = a column id >= n (the gather would read a
// wild address), bit1 = rowptr not monotone / not ending at nnz
__global__ __launch_bounds__(256) void k_validate_graph(GraphDev g, uint32_t *__restrict__ flags) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    uint32_t bad = 0;
    // (four ids a load where the array is 16-byte aligned: one id a thread and trip kept 2 MB in flight — 1.8 TB/s, 0.57 ms of
    // R-MAT-24's hand-off; the pad behind nnz holds zeros or whatever the caller left there — only ids in front of nnz count)
    const size_t quads = (reinterpret_cast<uintptr_t>(g.col) & 15u) ? 0 : g.nnz / 4;
    for (size_t q = gid; q < quads; q += step) {
        const uint4 c = reinterpret_cast<const uint4 *>(g.col)[q];
        if (c.x >= g.n || c.y >= g.n || c.z >= g.n || c.w >= g.n) bad |= 1u;
    }
    for (size_t i = quads * 4 + gid; i < g.nnz; i += step)
        if (g.col[i] >= g.n) bad |= 1u;
    const size_t r0 = g.lo(), r1 = g.hi();   // (a slice holds rows [r0, r1) only; its row pointers start at 0)
    for (size_t u = r0 + gid; u < r1; u += step)
        if (g.rowptr[u] > g.rowptr[u + 1]) bad |= 2u;
    if (gid == 0 && (g.rowptr[r0] != 0 || (uint64_t)g.rowptr[r1] != g.nnz)) bad |= 2u;
    if (bad) atomicOr(flags, bad);
}

// the row pointers alone (a host hand-off has them on the device long before the column array: handoff_early classes the
// graph and starts the flat plan builders from them, and those index the column and entry arrays with what they find here)
__global__ __launch_bounds__(256) void k_validate_rowptr(GraphDev g, uint32_t *__restrict__ flags) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    uint32_t bad = 0;
    const size_t r0 = g.lo(), r1 = g.hi();
    for (size_t u = r0 + gid; u < r1; u += step)
        if (g.rowptr[u] > g.rowptr[u + 1]) bad |= 2u;
    if (gid == 0 && (g.rowptr[r0] != 0 || (uint64_t)g.rowptr[r1] != g.nnz)) bad |= 2u;
    if (bad) atomicOr(flags, bad);
}

// uint64 row pointers (host ABI) -> uint32 (device layout); a value that does not fit saturates, so that the monotone /
// "ends at nnz" checks see it instead of its low half
__global__ void k_narrow_rowptr(const unsigned long long *__restrict__ in, uint32_t *__restrict__ out, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = in[i] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)in[i];
}

__global__ void k_zero_row(float *buf, uint32_t n, uint32_t width) {
    if (threadIdx.x < width) buf[(size_t)n * width + threadIdx.x] = 0.0f;
}

inline unsigned blocks_for(size_t work, unsigned block) {
    return (unsigned)((work + block - 1) / block);
}

}  // namespace

// ---------------------------------------------------------------------- launchers

void set_kernel_trace(KernelTraceSink *sink) { t_sink = sink; }

int stage_variant(int f, int n1, int n2, int n3, int sigmoid_last) {
    if (f == 1 && n1 == 32 && n2 == 32 && n3 == 16 && !sigmoid_last) return 0;
    if (f == 16 && n1 == 32 && n2 == 32 && n3 == 16 && !sigmoid_last) return 1;
    if (f == 16 && n1 == 32 && n2 == 16 && n3 == 1 && sigmoid_last) return 2;
    return -1;
}

// a 16-wide stage from the L2-resident compact table (k_stage_t4); it leaves at once unless the table in place fits this input
// a stage on wide tiles (k_stage_w1 / k_stage_w16: a workgroup per 64-vertex tile; small graphs without long rows)
hipError_t launch_stage_wide(const StagePlan &sp, const GraphDev &g, float ws, const float *params, const float *in, float *out,
                             float *logits, uint32_t row_lo, uint32_t row_hi, hipStream_t stream) {
    if (row_hi <= row_lo) return hipSuccess;
    const dim3 grid((row_hi - row_lo + kWave - 1) / kWave), block(kBlock);
    const float *P = params + sp.param_offset;
    switch (sp.variant) {
    case 0:
        GNNVC_LAUNCH((k_stage_w1<32, 32, 16>), grid, block, 0, stream, g, ws, in, out, P, row_lo, row_hi);
        break;
    case 1:
        GNNVC_LAUNCH((k_stage_w16<32, 32, 16, false>), grid, block, 0, stream, g, ws, reinterpret_cast<const float4 *>(in), out,
                     (float *)nullptr, P, row_lo, row_hi);
        break;
    case 2:
        GNNVC_LAUNCH((k_stage_w16<32, 16, 1, true>), grid, block, 0, stream, g, ws, reinterpret_cast<const float4 *>(in), out, logits, P,
                     row_lo, row_hi);
        break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

#ifndef GNNVC_T4_S
#define GNNVC_T4_S 4   // neighbours' table rows in flight per vertex
#endif
hipError_t launch_stage_t4(const StagePlan &sp, const GraphDev &g, float ws, const float *params, const float *in, float *out, float *logits,
                           uint32_t row_lo, uint32_t row_hi, bool interleave, hipStream_t stream, const float *table_in,
                           const unsigned long long *counts_in, unsigned long long *counts_zero, const uint32_t *desc_in, uint32_t *desc_out,
                           const EmitArgs &emit, bool solo) {
    if (row_hi <= row_lo) return hipSuccess;
    if (sp.f != 16 || (sp.variant != 1 && sp.variant != 2) || !table_in || !counts_in || !counts_zero || counts_zero == counts_in || !desc_in ||
        !desc_out || desc_in == desc_out)
        return hipErrorInvalidValue;
    const uint32_t ntiles = (row_hi - row_lo + kWave - 1) / kWave;
    const uint32_t per_xcd = (ntiles + 7) / 8;
    const uint32_t blocks_per_xcd = (per_xcd + kWavesPerBlock - 1) / kWavesPerBlock;
    const dim3 grid(blocks_per_xcd * 8 + 1), block(kBlock);   // (+ 1: the workgroup that chooses)
    const float *P = params + sp.param_offset;
    const float4 *in4 = reinterpret_cast<const float4 *>(in), *tab = reinterpret_cast<const float4 *>(table_in);
    if (sp.variant == 1)
        GNNVC_LAUNCH((k_stage_t4<32, 32, 16, false, GNNVC_T4_S>), grid, block, 0, stream, g, ws, in4, out, (float *)nullptr, P, row_lo, row_hi,
                     interleave ? 1 : 0, tab, counts_in, counts_zero, desc_in, desc_out, emit.spec, reinterpret_cast<c4row *>(emit.table),
                     emit.counts, solo ? 1 : 0);
    else
        GNNVC_LAUNCH((k_stage_t4<32, 16, 1, true, GNNVC_T4_S>), grid, block, 0, stream, g, ws, in4, out, logits, P, row_lo, row_hi,
                     interleave ? 1 : 0, tab, counts_in, counts_zero, desc_in, desc_out, (const uint32_t *)nullptr, (c4row *)nullptr,
                     (unsigned long long *)nullptr, solo ? 1 : 0);
    return hipGetLastError();
}

// the dense layers + sigmoid of the last stage when its aggregates are ready (compact-table plan): rows [row_lo, row_hi)
hipError_t launch_dense_sigmoid(const StagePlan &sp, const GraphDev &g, float ws, const float *params, const float *in, float *out,
                                float *logits, uint32_t row_lo, uint32_t row_hi, const float *acc4, const uint32_t *c4desc,
                                const float *agg16, hipStream_t stream, uint32_t long_thresh, const float *table_in) {
    if (row_hi <= row_lo) return hipSuccess;
    if (sp.f != 16 || sp.variant != 2 || !acc4 || !c4desc) return hipErrorInvalidValue;
    const dim3 grid((row_hi - row_lo + kBlock - 1) / kBlock), block(kBlock);
    GNNVC_LAUNCH((k_dense_f16<32, 16, 1, true>), grid, block, 0, stream, g, ws, reinterpret_cast<const float4 *>(in), out, logits,
                       params + sp.param_offset, row_lo, row_hi, reinterpret_cast<const float4 *>(acc4), c4desc,
                       reinterpret_cast<const float4 *>(agg16), reinterpret_cast<const float4 *>(table_in), long_thresh,
                       (const uint32_t *)nullptr, (c4row *)nullptr, (unsigned long long *)nullptr);
    return hipGetLastError();
}

hipError_t launch_stage(const StagePlan &sp, const GraphDev &g, float ws, const float *params,
                        const float *in, float *out, float *logits, uint32_t row_lo,
                        uint32_t row_hi, uint32_t long_thresh, bool mfma, const SortedOrder *so,
                        bool interleave, hipStream_t stream, const float *acc4, const uint32_t *c4desc, const float *agg16,
                        bool mfma_agg, const EmitArgs &emit, bool dense_part, const SortedOrder *so_pruned, const float *table_in,
                        const uint32_t *skip_flag) {
    if (row_hi <= row_lo) return hipSuccess;
    const bool sorted = so && so->n > 0;
    const bool with_p = sorted && g.prune_eff && so_pruned && so_pruned->vertex;
    if (g.prune_eff && sorted && !with_p) return hipErrorInvalidValue;   // (classing by entries left needs the matching tile order)
    if (so && so->n == 0 && !with_p) return hipSuccess;   // every row of the range is a long row
    const uint32_t n_p = with_p ? so_pruned->n : 0u;
    const uint32_t ntiles = sorted ? (std::max(so->n, n_p) + kWave - 1) / kWave : (row_hi - row_lo + kWave - 1) / kWave;
    const uint32_t per_xcd = (ntiles + 7) / 8;
    const uint32_t blocks_per_xcd = (per_xcd + kWavesPerBlock - 1) / kWavesPerBlock;
    const dim3 grid(blocks_per_xcd * 8), block(kBlock);
    const float *P = params + sp.param_offset;
    const float4 *in4 = reinterpret_cast<const float4 *>(in);
    const float *nofloat = nullptr;
    const int il = interleave ? 1 : 0;
    switch (sp.variant * 2 + (mfma ? 1 : 0)) {
    case 0:
        GNNVC_LAUNCH((k_stage_f1<32, 32, 16, 4, false>), grid, block, 0, stream, g, ws, in, out, P,
                           row_lo, row_hi, g.rowptr, g.col, nofloat, long_thresh, il, (const uint32_t *)nullptr,
                           emit.spec, reinterpret_cast<c4row *>(emit.table), emit.counts, sorted ? so->vertex : nullptr,
                           sorted ? reinterpret_cast<const uint4 *>(so->meta) : nullptr, sorted ? so->n : 0u);
        break;
    case 1:
        GNNVC_LAUNCH((k_stage_f1<32, 32, 16, 4, true>), grid, block, 0, stream, g, ws, in, out, P,
                           row_lo, row_hi, g.rowptr, g.col, nofloat, long_thresh, il, (const uint32_t *)nullptr,
                           (const uint32_t *)nullptr, (c4row *)nullptr, (unsigned long long *)nullptr,
                           sorted ? so->vertex : nullptr, sorted ? reinterpret_cast<const uint4 *>(so->meta) : nullptr,
                           sorted ? so->n : 0u);
        break;
// neighbour rows in flight per vertex in the 16-wide tile kernel (x 4 vertices per quad).  Natural tiles: 2.  Degree-sorted
// tiles (skewed graphs: rows of a tile have similar, mostly larger degrees): 3 — measured R-MAT-22 6.71 -> 6.53 ms, power-law
// 2.00 -> 1.82 ms per forward; 4 costs a wave per SIMD (power-law stage 1: 0.75 -> 1.6 ms) and natural tiles gain nothing
// from 3 (metric graph, plain kernels: 9.73 vs 9.75 ms).
#ifndef GNNVC_GATHER_S
#define GNNVC_GATHER_S 2
#endif
#ifndef GNNVC_GATHER_S_SORTED
#define GNNVC_GATHER_S_SORTED 3
#endif
#define GNNVC_F16_ARGS(SIG_, MF_, LG_)                                                                              \
                       grid, block, 0, stream, g, ws, in4, out,                                                           \
                       LG_, P, row_lo, row_hi, long_thresh, sorted ? so->vertex : nullptr,                                \
                       sorted ? so->meta : nullptr, sorted ? so->n : 0u, il,                                              \
                       (const float4 *)nullptr, acc4 ? c4desc : skip_flag, (const float4 *)nullptr,                       \
                       (MF_ || SIG_) ? nullptr : emit.spec, reinterpret_cast<c4row *>((MF_ || SIG_) ? nullptr : emit.table), \
                       (MF_ || SIG_) ? nullptr : emit.counts,                                                             \
                       with_p ? so_pruned->vertex : nullptr, with_p ? so_pruned->meta : nullptr, n_p
#define GNNVC_LAUNCH_X(...) GNNVC_LAUNCH(__VA_ARGS__)   // (the argument list above is expanded on the way through)
#define GNNVC_LAUNCH_F16(N2_, N3_, SIG_, MF_, SRT_, S_, LG_) \
    GNNVC_LAUNCH_X((k_stage_f16<32, N2_, N3_, SIG_, S_, MF_, SRT_>), GNNVC_F16_ARGS(SIG_, MF_, LG_))
#define GNNVC_LAUNCH_F16F(N2_, N3_, SIG_, MF_, SRT_, S_, LG_, FLT_) \
    GNNVC_LAUNCH_X((k_stage_f16<32, N2_, N3_, SIG_, S_, MF_, SRT_, false, FLT_>), GNNVC_F16_ARGS(SIG_, MF_, LG_))
    case 2:
        if (sorted) GNNVC_LAUNCH_F16(32, 16, false, false, true, GNNVC_GATHER_S_SORTED, nullptr);
        else GNNVC_LAUNCH_F16(32, 16, false, false, false, GNNVC_GATHER_S, nullptr);
        break;
    case 3:   // (the filtered gather comes with the matrix-core variants only: the default of the 16-wide stages)
        if (g.zero_bits) {
            if (sorted) GNNVC_LAUNCH_F16F(32, 16, false, true, true, GNNVC_GATHER_S_SORTED, nullptr, true);
            else GNNVC_LAUNCH_F16F(32, 16, false, true, false, GNNVC_GATHER_S, nullptr, true);
        } else if (sorted) GNNVC_LAUNCH_F16(32, 16, false, true, true, GNNVC_GATHER_S_SORTED, nullptr);
        else GNNVC_LAUNCH_F16(32, 16, false, true, false, GNNVC_GATHER_S, nullptr);
        break;
    case 4:
        if (sorted) GNNVC_LAUNCH_F16(16, 1, true, false, true, GNNVC_GATHER_S_SORTED, logits);
        else GNNVC_LAUNCH_F16(16, 1, true, false, false, GNNVC_GATHER_S, logits);
        break;
    case 5:
        if (g.zero_bits) {
            if (sorted) GNNVC_LAUNCH_F16F(16, 1, true, true, true, GNNVC_GATHER_S_SORTED, logits, true);
            else GNNVC_LAUNCH_F16F(16, 1, true, true, false, GNNVC_GATHER_S, logits, true);
        } else if (sorted) GNNVC_LAUNCH_F16(16, 1, true, true, true, GNNVC_GATHER_S_SORTED, logits);
        else GNNVC_LAUNCH_F16(16, 1, true, true, false, GNNVC_GATHER_S, logits);
        break;
#undef GNNVC_LAUNCH_F16
#undef GNNVC_LAUNCH_F16F
#undef GNNVC_F16_ARGS
#undef GNNVC_LAUNCH_X
    default:
        return hipErrorInvalidValue;
    }
    if (acc4 && sp.f == 16 && dense_part) {
        // compact-table plan: the aggregate-only variant does the launch when the device found the input fit for
        // it (the gathering variant above has then left at once, and the other way round).  Without gathers to
        // overlap with, the dense layers run faster on the VALU (one lane per vertex, weights from SGPRs) than on
        // the fp32 matrix cores: 6.65 vs 7.16 ms per forward on the metric graph.  It always walks natural tiles
        // (nothing to balance without a gather, and consecutive rows read their sums coalesced).
        const uint32_t nt = (row_hi - row_lo + kWave - 1) / kWave;
        const dim3 grid((((nt + 7) / 8 + kWavesPerBlock - 1) / kWavesPerBlock) * 8);
#define GNNVC_LAUNCH_AGG(N2_, N3_, SIG_, MF_, LG_)                                                        \
        GNNVC_LAUNCH((k_stage_f16<32, N2_, N3_, SIG_, 2, MF_, false, true>), grid, block, 0, stream, g, ws, in4, \
                           out, LG_, P, row_lo, row_hi, long_thresh, (const uint32_t *)nullptr, (const uint4 *)nullptr, \
                           0u, il, reinterpret_cast<const float4 *>(acc4), c4desc, reinterpret_cast<const float4 *>(agg16), \
                           (MF_ || SIG_) ? nullptr : emit.spec, reinterpret_cast<c4row *>((MF_ || SIG_) ? nullptr : emit.table), \
                           (MF_ || SIG_) ? nullptr : emit.counts)
        if (sp.variant == 1) {
            if (mfma_agg) {
                GNNVC_LAUNCH_AGG(32, 16, false, true, nullptr);
            } else {   // (round 4: one lane per row, the terms that are zero left out — k_dense_f16)
                const dim3 dgrid((row_hi - row_lo + kBlock - 1) / kBlock);
                GNNVC_LAUNCH((k_dense_f16<32, 32, 16, false>), dgrid, block, 0, stream, g, ws, in4, out, (float *)nullptr, P, row_lo, row_hi,
                                   reinterpret_cast<const float4 *>(acc4), c4desc, reinterpret_cast<const float4 *>(agg16),
                                   reinterpret_cast<const float4 *>(table_in), long_thresh, emit.spec,
                                   reinterpret_cast<c4row *>(emit.table), emit.counts);
            }
        } else {
            if (mfma_agg) GNNVC_LAUNCH_AGG(16, 1, true, true, logits);
            else return launch_dense_sigmoid(sp, g, ws, params, in, out, logits, row_lo, row_hi, acc4, c4desc, agg16, stream, long_thresh, table_in);
        }
#undef GNNVC_LAUNCH_AGG
    }
    return hipGetLastError();
}

// Do kernels on stream b run BESIDE kernels on stream a?  HIP streams are placed on a small pool of HSA queues by the runtime
// (four per priority; a new stream takes the one with the fewest users, ties included — which can be the queue of the very
// stream it is meant to run beside: then everything on the two is serialised, and nothing tells).  One wave spins ~300 us on
// a, an empty kernel goes to b: if b's is done while a's still spins, they are on different queues.
__global__ void k_spin(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();   // (100 MHz)
    for (int i = 0; i < (1 << 22) && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(16);
}
struct WordList {
    const uint32_t *src[24];
};
__global__ void k_words(WordList wl, uint32_t *__restrict__ out) {
    const int i = threadIdx.x;
    if (i < 24 && wl.src[i]) out[i] = *wl.src[i];
}
hipError_t classify_graph(const GraphDev &g, const GraphClassArgs &a, uint32_t *dev_words, uint32_t *out_dev, hipStream_t stream) {
    hipError_t rc = hipMemsetAsync(dev_words, 0, 16 * sizeof(uint32_t), stream);
    if (rc != hipSuccess) return rc;
    WordList wl;
    for (auto &p : wl.src) p = nullptr;
    wl.src[0] = dev_words;
    if (g.n) GNNVC_LAUNCH(k_validate_graph, dim3(2048), dim3(256), 0, stream, g, dev_words);
    const uint32_t lo = g.lo(), hi = g.hi();
    if (a.cuts && hi > lo)
        for (int k = 0; k <= 8; ++k) wl.src[1 + k] = g.rowptr + lo + (size_t)((uint64_t)(hi - lo) * k / 8);
    if (a.waste && hi > lo) {
        GNNVC_LAUNCH(k_tile_waste, dim3(std::min<unsigned>((hi - lo + 255) / 256, 512u)), dim3(256), 0, stream, g, lo, hi, a.waste_thresh,   // (two same-address atomics a block: 2048 blocks were 40 - 50 us of them)
                     reinterpret_cast<unsigned long long *>(dev_words + 2), a.heavy_from);
        for (int k = 0; k < 4; ++k) wl.src[10 + k] = dev_words + 2 + k;
    }
    if (a.longs && hi > lo && a.long_list) {
        GNNVC_LAUNCH(k_find_long, dim3(std::min<unsigned>((hi - lo + 1023) / 1024, 2048u)), dim3(256), 0, stream, g, a.long_thresh, a.long_list,
                     dev_words + 8);
        for (int k = 0; k < 4; ++k) wl.src[14 + k] = dev_words + 8 + k;
    }
    GNNVC_LAUNCH(k_words, dim3(1), dim3(64), 0, stream, wl, out_dev);
    return hipGetLastError();
}

__global__ void k_verdicts(VerdictWords vw, uint32_t *__restrict__ out) {
    const int i = threadIdx.x;
    if (i < 8) out[i] = vw.src[i] ? *vw.src[i] : 0u;
}
hipError_t write_verdicts(const VerdictWords &vw, uint32_t *out_dev, hipStream_t stream) {
    GNNVC_LAUNCH(k_verdicts, dim3(1), dim3(64), 0, stream, vw, out_dev);
    return hipGetLastError();
}

__global__ void k_nothing() {}

hipError_t streams_run_side_by_side(hipStream_t a, hipStream_t b, bool *yes) {
    *yes = false;
    hipEvent_t ea = nullptr, eb = nullptr;
    hipError_t rc = hipEventCreateWithFlags(&ea, hipEventDisableTiming);
    if (rc == hipSuccess) rc = hipEventCreateWithFlags(&eb, hipEventDisableTiming);
    if (rc == hipSuccess) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, 30000ull);
        rc = hipEventRecord(ea, a);
    }
    if (rc == hipSuccess) {
        hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, b);
        rc = hipEventRecord(eb, b);
    }
    if (rc == hipSuccess) rc = hipEventSynchronize(eb);
    if (rc == hipSuccess) {
        *yes = hipEventQuery(ea) == hipErrorNotReady;
        rc = hipEventSynchronize(ea);
    }
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    (void)hipGetLastError();   // (hipErrorNotReady is sticky for hipGetLastError)
    return rc;
}

hipError_t launch_graph_layer(const GraphDev &g, float ws, uint32_t f, const float *in,
                              float *out, hipStream_t stream) {
    const size_t work = (size_t)g.n * (2 * f + 3);
    if (!work) return hipSuccess;
    GNNVC_LAUNCH(k_graph_layer, dim3(blocks_for(work, 256)), dim3(256), 0, stream, g, ws, f,
                       in, out);
    return hipGetLastError();
}

hipError_t launch_linear(uint32_t n, uint32_t k, uint32_t m, const float *in, const float *W,
                         const float *bias, float *out, hipStream_t stream) {
    const size_t work = (size_t)n * m;
    if (!work) return hipSuccess;
    GNNVC_LAUNCH(k_linear, dim3(blocks_for(work, 256)), dim3(256), 0, stream, n, k, m, in,
                       W, bias, out);
    return hipGetLastError();
}

hipError_t launch_relu(size_t count, const float *in, float *out, hipStream_t stream) {
    if (!count) return hipSuccess;
    const unsigned nb = (unsigned)((count + 255) / 256 < 8192 ? (count + 255) / 256 : 8192);
    GNNVC_LAUNCH(k_relu, dim3(nb), dim3(256), 0, stream, count, in, out);
    return hipGetLastError();
}

hipError_t launch_sigmoid(size_t count, const float *in, float *out, hipStream_t stream) {
    if (!count) return hipSuccess;
    const unsigned nb = (unsigned)((count + 255) / 256 < 8192 ? (count + 255) / 256 : 8192);
    GNNVC_LAUNCH(k_sigmoid, dim3(nb), dim3(256), 0, stream, count, in, out);
    return hipGetLastError();
}

hipError_t launch_sgemm(int ta, int tb, uint32_t m, uint32_t n, uint32_t k, const float *A,
                        uint32_t lda, const float *B, uint32_t ldb, float beta, float *C,
                        uint32_t ldc, hipStream_t stream) {
    const size_t work = (size_t)m * n;
    if (!work) return hipSuccess;
    GNNVC_LAUNCH(k_sgemm, dim3(blocks_for(work, 256)), dim3(256), 0, stream, ta, tb, m, n, k,
                       A, lda, B, ldb, beta, C, ldc);
    return hipGetLastError();
}

// ---- column-blocked stage 0 ----------------------------------------------------------
static hipError_t scan_u32(uint32_t *data, size_t n, uint32_t *scratch, hipStream_t stream) {
    // scratch needs ceil(n/chunk) + ceil(that/chunk) + ... entries (callers reserve n/2048 + 8192)
    if (n == 0) return hipSuccess;
    const size_t chunks = (n + kScanChunk - 1) / kScanChunk;
    GNNVC_LAUNCH(k_scan_chunks, dim3((unsigned)chunks), dim3(256), 0, stream, data, n, scratch);
    if (chunks > 1) {
        hipError_t rc = scan_u32(scratch, chunks, scratch + chunks, stream);
        if (rc != hipSuccess) return rc;
        GNNVC_LAUNCH(k_scan_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, data, n,
                           scratch);
    }
    return hipGetLastError();
}

size_t blocked_scan_scratch_elems(size_t n_elems) { return n_elems / 2048 + 8192; }

hipError_t build_blocked_index(const GraphDev &g, uint32_t wb, uint32_t nblocks, uint32_t long_thresh,
                               uint32_t *bp, uint32_t *colb, uint32_t *scratch, uint32_t *bad_flag,
                               hipStream_t stream) {
    const size_t elems = (size_t)nblocks * g.n + 1;
    hipError_t rc = hipMemsetAsync(bp, 0, elems * sizeof(uint32_t), stream);
    if (rc != hipSuccess) return rc;
    rc = hipMemsetAsync(bad_flag, 0, sizeof(uint32_t), stream);
    if (rc != hipSuccess) return rc;
    const unsigned nb = (g.n + 255) / 256;
    GNNVC_LAUNCH(k_blk_count, dim3(nb), dim3(256), 0, stream, g, wb, long_thresh, bp, bad_flag);
    rc = scan_u32(bp, elems, scratch, stream);
    if (rc != hipSuccess) return rc;
    GNNVC_LAUNCH(k_blk_scatter, dim3(nb), dim3(256), 0, stream, g, wb, long_thresh, bp, colb);
    return hipGetLastError();
}

hipError_t launch_stage0_blocked(const StagePlan &sp, const GraphDev &g, float ws, const float *params,
                                 const float *x, float *out, uint32_t row_lo, uint32_t row_hi,
                                 uint32_t nblocks, const uint32_t *bp, const uint32_t *colb, float *acc,
                                 uint32_t long_thresh, bool mfma, bool interleave, hipStream_t stream, const EmitArgs &emit) {
    if (row_hi <= row_lo) return hipSuccess;
    if (sp.variant != 0) return hipErrorInvalidValue;
    const unsigned nb = (row_hi - row_lo + 255) / 256;
    for (uint32_t k = 0; k + 1 < nblocks; ++k)
        GNNVC_LAUNCH(k_blk_accumulate, dim3(nb), dim3(256), 0, stream, bp + (size_t)k * g.n, colb, x,
                           acc, row_lo, row_hi, k == 0 ? 1 : 0);
    const uint32_t ntiles = (row_hi - row_lo + kWave - 1) / kWave;
    const uint32_t per_xcd = (ntiles + 7) / 8;
    const uint32_t blocks_per_xcd = (per_xcd + kWavesPerBlock - 1) / kWavesPerBlock;
    const uint32_t *ep = bp + (size_t)(nblocks - 1) * g.n;
    const float *acc_in = nblocks > 1 ? acc : nullptr;
    const dim3 grid(blocks_per_xcd * 8), block(kBlock);
    if (mfma)
        GNNVC_LAUNCH((k_stage_f1<32, 32, 16, 4, true>), grid, block, 0, stream, g, ws, x, out,
                           params + sp.param_offset, row_lo, row_hi, ep, colb, acc_in, long_thresh, interleave ? 1 : 0,
                           (const uint32_t *)nullptr, (const uint32_t *)nullptr, (c4row *)nullptr, (unsigned long long *)nullptr,
                           (const uint32_t *)nullptr, (const uint4 *)nullptr, 0u);
    else
        GNNVC_LAUNCH((k_stage_f1<32, 32, 16, 4, false>), grid, block, 0, stream, g, ws, x, out,
                           params + sp.param_offset, row_lo, row_hi, ep, colb, acc_in, long_thresh, interleave ? 1 : 0,
                           (const uint32_t *)nullptr, emit.spec, reinterpret_cast<c4row *>(emit.table), emit.counts,
                           (const uint32_t *)nullptr, (const uint4 *)nullptr, 0u);
    return hipGetLastError();
}

// ---- LDS-table plan of the F = 1 stage ----------------------------------------------------
// kernels that need more than 64 KiB of dynamic LDS must be told so once per device (engines on several devices may
// live in one process; `done` has one bit per device ordinal)
static hipError_t allow_dynamic_lds(const void *func, int bytes, std::atomic<uint64_t> &done) {
    int dev = 0;
    hipError_t rc = hipGetDevice(&dev);
    if (rc != hipSuccess) return rc;
    if (dev >= 0 && dev < 64 && (done.load(std::memory_order_acquire) >> dev & 1u)) return hipSuccess;
    rc = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (rc == hipSuccess && dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
    return rc;
}

// plan builders: the flat kernels serve plans over consecutive rows and uniform blocks
static bool lt_flat_plan(uint32_t rows_per_chunk, const PlanMap &pm) { return !pm.rowmap && !pm.bstart && rows_per_chunk <= kLtFlatRows; }
static uint32_t lt_flat_threads() {
    return 256u;
}

uint32_t lds_table_max_rows(uint32_t bits) { return 16u * lt_slice_rows_max(bits); }
uint32_t lds_table_block(uint32_t bits) { return lt_block_cols(bits); }
uint32_t lds_table_bits_for(uint32_t kmax) { return kmax <= 255u ? 8u : (kmax <= 1023u ? 10u : (kmax <= 65535u ? 16u : 0u)); }
size_t lds_table_bytes_for(uint32_t bits, size_t n) { return lt_table_bytes(bits, n); }
uint32_t lds_table_step() { return kLtwStep; }
uint32_t lds_table_record_words() { return kLtwRec; }

hipError_t lds_table_wsteps(const GraphDev &g, uint32_t slice_rows, uint32_t nchunks, uint32_t nblocks, const uint32_t *seg_cnt,
                            const uint32_t *step_ptr, uint32_t *step_count, uint32_t *recs, bool write, uint32_t slack,
                            hipStream_t stream, const PlanMap &pm, uint32_t row_base, uint32_t row_end, uint32_t bits) {
    if (slack < 3u * nblocks + 4u) return hipErrorInvalidValue;
    if (bits != 8 && bits != 10 && bits != 16) return hipErrorInvalidValue;
    if (row_end > g.n) row_end = g.n;
    GNNVC_LAUNCH(k_ltw_steps, dim3((nchunks + 63) / 64), dim3(64), 0, stream, g, slice_rows, nchunks, nblocks, seg_cnt,
                       step_ptr, step_count, recs, write ? 1 : 0, kLtwStep, slack, lt_block_cols(bits), pm, row_base, row_end,
                       lt_piece_cols(bits));
    return hipGetLastError();
}

hipError_t lds_table_wmax(const uint32_t *w, uint32_t n, uint32_t *wmax, hipStream_t stream) {
    hipError_t rc = hipMemsetAsync(wmax, 0, sizeof(uint32_t), stream);
    if (rc != hipSuccess || !n) return rc;
    GNNVC_LAUNCH(k_lt_wmax, dim3(std::min<unsigned>((n + 255) / 256, 1024u)), dim3(256), 0, stream, w, n, wmax);
    return hipGetLastError();
}

bool lds_table_is_flat(uint32_t rows_per_chunk, const PlanMap &pm) { return lt_flat_plan(rows_per_chunk, pm); }

hipError_t lds_table_count(const GraphDev &g, uint32_t rows_per_chunk, uint32_t nchunks, uint32_t nblocks, uint32_t block_cols,
                           uint32_t *seg_cnt, uint32_t *bad, hipStream_t stream, uint32_t row_base, uint32_t row_end,
                           const PlanMap &pm, uint32_t chunk0, uint32_t chunk1) {
    if (row_end > g.n) row_end = g.n;
    if (nblocks > 4096 || block_cols > (1u << 18) || rows_per_chunk > (1u << 15)) return hipErrorInvalidValue;
    if (chunk1 > nchunks) chunk1 = nchunks;
    if (chunk0 >= chunk1) return hipSuccess;
    if (lt_flat_plan(rows_per_chunk, pm)) {
        GNNVC_LAUNCH(k_lt_count_flat, dim3(chunk1 - chunk0), dim3(lt_flat_threads()), nblocks * sizeof(uint32_t), stream, g, rows_per_chunk,
                     nblocks, block_cols, seg_cnt, row_base, row_end, chunk0);
        return hipGetLastError();
    }
    if (chunk0 != 0 || chunk1 != nchunks) return hipErrorInvalidValue;   // (only the flat builders work piece by piece)
    GNNVC_LAUNCH(k_lt_count, dim3(nchunks), dim3(1024), 0, stream, g, rows_per_chunk, nblocks, block_cols, seg_cnt, bad,
                       row_base, row_end, pm);
    return hipGetLastError();
}

hipError_t lds_table_steps(const GraphDev &g, uint32_t rows_per_chunk, uint32_t nchunks, uint32_t nblocks, const uint32_t *seg_cnt,
                           const uint32_t *step_ptr, uint32_t *step_count, void *steps, bool write, hipStream_t stream,
                           uint32_t row_base, uint32_t row_end, uint32_t cap, uint32_t slack, uint32_t block_cols,
                           const PlanMap &pm) {
    if (row_end > g.n) row_end = g.n;
    if (cap == 0) cap = kLtStep;
    if (slack && slack < 3u * nblocks + 4u) return hipErrorInvalidValue;
    GNNVC_LAUNCH(k_lt_steps, dim3((nchunks + 255) / 256), dim3(256), 0, stream, g, rows_per_chunk, nchunks, nblocks,
                       seg_cnt, step_ptr, step_count, reinterpret_cast<uint4 *>(steps), write ? 1 : 0, row_base, row_end, cap, slack,
                       block_cols, pm);
    return hipGetLastError();
}

hipError_t lds_table_scatter(const GraphDev &g, uint32_t rows_per_chunk, uint32_t nchunks, uint32_t nblocks, uint32_t block_cols,
                             const uint32_t *seg_cnt, uint32_t *entries, hipStream_t stream, uint32_t shift,
                             uint32_t row_base, uint32_t row_end, uint32_t slack, const PlanMap &pm, uint32_t *bad, uint32_t chunk0,
                             uint32_t chunk1) {
    if (block_cols > (1u << shift) || ((uint64_t)rows_per_chunk << shift) > (1ull << 32)) return hipErrorInvalidValue;
    if (slack && slack < 3u * nblocks + 4u) return hipErrorInvalidValue;
    if (row_end > g.n) row_end = g.n;
    if (chunk1 > nchunks) chunk1 = nchunks;
    if (chunk0 >= chunk1) return hipSuccess;
    if (!lt_flat_plan(rows_per_chunk, pm) && (chunk0 != 0 || chunk1 != nchunks)) return hipErrorInvalidValue;
    if (lt_flat_plan(rows_per_chunk, pm)) {
        // staging area for a chunk's regrouped range: its mean length and a margin (entries + pad slots) within the 160 KiB of a
        // CU's LDS; the workgroup as large as it can be, since the area bounds how many of them a CU holds
        static std::atomic<uint64_t> lds_ok{0};
        constexpr size_t kLdsMax = 156u * 1024u;
        hipError_t rc = allow_dynamic_lds(reinterpret_cast<const void *>(k_lt_scatter_flat), (int)kLdsMax, lds_ok);
        if (rc != hipSuccess) return rc;
        const uint64_t mean = nchunks ? g.nnz / nchunks : 0;
        const uint32_t threads = 1024u;
        const size_t fixed = ((size_t)nblocks + rows_per_chunk + 1 + threads) * sizeof(uint32_t);   // cursors, offsets, 64 flags a wave
        uint64_t stage = (mean + mean / 8 + 3ull * nblocks + 64 + 255) / 256 * 256;
        if (fixed + stage * sizeof(uint32_t) > kLdsMax) stage = (kLdsMax - fixed) / sizeof(uint32_t);
        GNNVC_LAUNCH(k_lt_scatter_flat, dim3(chunk1 - chunk0), dim3(threads), fixed + stage * sizeof(uint32_t), stream, g, rows_per_chunk,
                     nblocks, block_cols, shift, seg_cnt, entries, row_base, row_end, slack, bad, (uint32_t)stage, chunk0);
        return hipGetLastError();
    }
    GNNVC_LAUNCH(k_lt_scatter, dim3(nchunks), dim3(1024), 0, stream, g, rows_per_chunk, nblocks, block_cols, shift, seg_cnt,
                       entries, row_base, row_end, slack, pm);
    return hipGetLastError();
}

hipError_t filter_mark(const GraphDev &g, const float *feat, uint32_t *bits, unsigned long long *info, hipStream_t stream,
                       const uint32_t *prev_bits, const unsigned long long *prev_info) {
    hipError_t rc = hipMemsetAsync(info, 0, 3 * sizeof(unsigned long long), stream);
    if (rc != hipSuccess || g.n == 0) return rc;
    GNNVC_LAUNCH(k_filter_mark, dim3(std::min<size_t>(((size_t)g.n + 1 + 255) / 256, 1024)), dim3(256), 0, stream, g,
                 reinterpret_cast<const float4 *>(feat), bits, info, prev_bits, prev_info);
    return hipGetLastError();
}

hipError_t prune_mark_zero(const GraphDev &g, const float *feat, uint32_t *heavy_bits, hipStream_t stream) {
    if (g.n == 0) return hipSuccess;
    GNNVC_LAUNCH(k_prune_mark_zero, dim3((g.n + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const float4 *>(feat), g.n, heavy_bits);
    return hipGetLastError();
}

// mask: one 64-bit word per chunk of 64 entries, off: chunks + 1 words (scanned in place: off[chunks] = kept entries;
// scratch as for blocked_scan_scratch_elems(chunks + 1))
hipError_t predict_zero_rows(const StagePlan &sp0, const GraphDev &g, float ws, const float *params, uint32_t *heavy_bits, hipStream_t stream) {
    if (g.n == 0) return hipSuccess;
    if (sp0.variant != 0 || g.sliced()) return hipErrorInvalidValue;   // (the F = 1 stage 5 -> 32 -> 32 -> 16; a slice lacks the other rows' degrees)
    GNNVC_LAUNCH((k_predict_zero_f1<32, 32, 16>), dim3((g.n + 255) / 256), dim3(256), 0, stream, g, ws, params + sp0.param_offset, heavy_bits);
    return hipGetLastError();
}

hipError_t prune_mass(const GraphDev &g, const uint32_t *heavy_bits, unsigned long long *mass, hipStream_t stream, const uint32_t *prev_bits) {
    hipError_t rc = hipMemsetAsync(mass, 0, 3 * sizeof(unsigned long long), stream);
    if (rc != hipSuccess || g.n == 0) return rc;
    GNNVC_LAUNCH(k_prune_mass, dim3(std::min<unsigned>((g.n + 255) / 256, 2048u)), dim3(256), 0, stream, g, heavy_bits, mass, prev_bits);
    return hipGetLastError();
}

hipError_t prune_count(const GraphDev &g, const uint32_t *heavy_bits, unsigned long long *mask, uint32_t *off, uint32_t *scratch,
                       hipStream_t stream) {
    if (g.nnz == 0 || g.nnz >= (1ull << 32)) return hipErrorInvalidValue;
    const uint32_t nnz = (uint32_t)g.nnz, chunks = (nnz + 63) / 64;
    hipError_t rc = hipMemsetAsync(off + chunks, 0, sizeof(uint32_t), stream);
    if (rc != hipSuccess) return rc;
    GNNVC_LAUNCH(k_prune_chunks, dim3(std::min<unsigned>((chunks + 15) / 16, 16384u)), dim3(256), 0, stream, g.col, nnz, heavy_bits, mask, off);
    return scan_u32(off, (size_t)chunks + 1, scratch, stream);
}

// pcol: the kept entries (+ pad); prp: rows + 1 words for the rows [g.lo(), g.hi()) this engine holds (NOT biased here)
hipError_t prune_fill(const GraphDev &g, const unsigned long long *mask, const uint32_t *off, uint32_t *pcol, uint32_t *prp,
                      hipStream_t stream) {
    if (g.nnz == 0 || g.nnz >= (1ull << 32)) return hipErrorInvalidValue;
    const uint32_t nnz = (uint32_t)g.nnz, chunks = (nnz + 63) / 64, rows = g.hi() - g.lo();
    GNNVC_LAUNCH(k_prune_fill, dim3(std::min<unsigned>((chunks + 3) / 4, 32768u)), dim3(256), 0, stream, g.col, nnz, mask, off, pcol);
    GNNVC_LAUNCH(k_prune_offsets, dim3((rows + 1 + 255) / 256), dim3(256), 0, stream, g, mask, off, chunks, prp - g.lo());
    return hipGetLastError();
}

hipError_t prune_check(const GraphDev &g, const float *feat, const uint32_t *heavy_bits, uint32_t *bad, hipStream_t stream) {
    hipError_t rc = hipMemsetAsync(bad, 0, sizeof(uint32_t), stream);
    if (rc != hipSuccess || g.n == 0) return rc;
    GNNVC_LAUNCH(k_prune_check, dim3(std::min<unsigned>((g.n + 255) / 256, 2048u)), dim3(256), 0, stream, g,
                 reinterpret_cast<const float4 *>(feat), heavy_bits, bad);
    return hipGetLastError();
}

hipError_t deal_rows(const GraphDev &g, const uint32_t *sorted_rows, uint32_t m, uint32_t slice_rows, uint32_t nslices,
                     uint32_t *rowmap, uint32_t *weight, hipStream_t stream) {
    if (!nslices || !slice_rows) return hipErrorInvalidValue;
    GNNVC_LAUNCH(k_map_deal, dim3((nslices + 3) / 4), dim3(256), 0, stream, g, sorted_rows, m, slice_rows, nslices, rowmap, weight);
    return hipGetLastError();
}

hipError_t mass_bounds(const GraphDev &g, unsigned long long target, uint32_t count, uint32_t *cand, hipStream_t stream) {
    if (!count) return hipSuccess;
    GNNVC_LAUNCH(k_mass_bounds, dim3((count + 255) / 256), dim3(256), 0, stream, g, target, count, cand);
    return hipGetLastError();
}

hipError_t launch_stage0_lds_table(const StagePlan &sp, const GraphDev &g, float ws, const float *params, const float *x,
                                   float *out, uint32_t row_lo, uint32_t row_hi, uint32_t rows_per_chunk,
                                   const uint32_t *step_ptr, const void *steps, const uint32_t *entries, uint8_t *wbyte,
                                   float *acc, uint32_t *bad, uint32_t long_thresh, bool mfma, bool interleave,
                                   hipStream_t stream, const EmitArgs &emit, uint32_t last_entry, const uint32_t *rowmap,
                                   uint32_t mapped_chunks, uint32_t plan_base, uint32_t plan_end, uint32_t bits) {
    if (row_hi <= row_lo) return hipSuccess;
    if (bits != 8 && bits != 10 && bits != 16) return hipErrorInvalidValue;
    if (rowmap && bits != 8) return hipErrorInvalidValue;   // (the skewed layout's block starts are multiples of 256 columns, not of a 10-bit piece)
    if (plan_end > g.n) plan_end = g.n;
    if (row_lo < plan_base || row_hi > plan_end) return hipErrorInvalidValue;
    if (rowmap && (row_lo != 0 || row_hi != g.n || mapped_chunks == 0)) return hipErrorInvalidValue;   // (a mapped plan sums all of its rows)
    if (sp.variant != 0 || rows_per_chunk == 0 || rows_per_chunk > 16u * lt_slice_rows_max(bits) || rows_per_chunk % 16u || g.nnz == 0)
        return hipErrorInvalidValue;
    // does this forward's input match the table?  decided on the device: no host round trip
    hipError_t rc = hipMemsetAsync(bad, 0, sizeof(uint32_t), stream);
    if (rc != hipSuccess) return rc;
    const dim3 tgrid(std::min<unsigned>((g.n / 4 + 255) / 256 + 1, 4096u));
    const uint32_t c0 = rowmap ? 0u : (row_lo - plan_base) / rows_per_chunk, c1 = rowmap ? mapped_chunks - 1 : (row_hi - 1 - plan_base) / rows_per_chunk;
    const uint32_t slice_rows = rows_per_chunk / 16u;
    constexpr size_t lds_max = (size_t)16 * kLtwSliceRows * 4 + 1024 + kLtwBlock;
    static_assert(lds_max <= 160 * 1024, "LDS budget of k_lt_agg");
    const size_t lds = (size_t)16 * slice_rows * 4 + lt_lut_floats(bits) * 4 + kLtwBlock;
#define GNNVC_LT_AGG(B_)                                                                                                          \
    {                                                                                                                             \
        GNNVC_LAUNCH(k_lt_bytes_x<B_>, tgrid, dim3(256), 0, stream, x, ws, g.n, wbyte, bad);                                      \
        static std::atomic<uint64_t> lds_ok{0};                                                                                   \
        rc = allow_dynamic_lds(reinterpret_cast<const void *>(k_lt_agg<B_>), (int)lds_max, lds_ok);                               \
        if (rc != hipSuccess) return rc;                                                                                          \
        GNNVC_LAUNCH(k_lt_agg<B_>, dim3(c1 - c0 + 1), dim3(1024), lds, stream, step_ptr, reinterpret_cast<const uint32_t *>(steps), \
                     entries, wbyte, ws, acc, g.n, slice_rows, c0, last_entry, bad, rowmap, rowmap ? 0u : plan_base,               \
                     rowmap ? g.n : plan_end);                                                                                    \
    }
    if (bits == 8) GNNVC_LT_AGG(8)
    else if (bits == 10) GNNVC_LT_AGG(10)
    else GNNVC_LT_AGG(16)
#undef GNNVC_LT_AGG
    const uint32_t ntiles = (row_hi - row_lo + kWave - 1) / kWave;
    const uint32_t per_xcd = (ntiles + 7) / 8;
    const uint32_t blocks_per_xcd = (per_xcd + kWavesPerBlock - 1) / kWavesPerBlock;
    const dim3 grid(blocks_per_xcd * 8), block(kBlock);
    if (mfma)
        GNNVC_LAUNCH((k_stage_f1<32, 32, 16, 4, true>), grid, block, 0, stream, g, ws, x, out, params + sp.param_offset,
                           row_lo, row_hi, g.rowptr, g.col, acc, long_thresh, interleave ? 1 : 0, (const uint32_t *)bad,
                           (const uint32_t *)nullptr, (c4row *)nullptr, (unsigned long long *)nullptr,
                           (const uint32_t *)nullptr, (const uint4 *)nullptr, 0u);
    else
        GNNVC_LAUNCH((k_stage_f1<32, 32, 16, 4, false>), grid, block, 0, stream, g, ws, x, out, params + sp.param_offset,
                           row_lo, row_hi, g.rowptr, g.col, acc, long_thresh, interleave ? 1 : 0, (const uint32_t *)bad,
                           emit.spec, reinterpret_cast<c4row *>(emit.table), emit.counts,
                           (const uint32_t *)nullptr, (const uint4 *)nullptr, 0u);
    return hipGetLastError();
}

// ---- compact-table plan of the 16-wide stages -----------------------------------------------
uint32_t compact_max_rows() { return kC4MaxRows; }
uint32_t compact_block() { return kC4Block; }
uint32_t compact_shift() { return kC4Shift; }
uint32_t compact_slices() { return kC4Slices; }
uint32_t compact_step() { return kC4Step; }
uint32_t compact_max_passes() { return kC4MaxPasses; }

// counts -> desc -> table(s) -> four sums per pass and row of [row_lo, row_hi) (chunks that straddle the ends are done
// whole; a mapped plan always does all of its rows).  `counts` holds the per-column non-zero counts of `in`
// (column_counts, same stream).
hipError_t launch_compact_gather(const GraphDev &g, const CompactPlan &cp, const float *in, const unsigned long long *counts,
                                 int count_slots, uint32_t *desc, float *table, float *acc4, uint32_t row_lo, uint32_t row_hi,
                                 uint32_t *dirty_rows, uint32_t dirty_cap, float *agg16, hipStream_t stream, int what) {
    if (row_hi <= row_lo || g.nnz == 0 || row_lo < cp.plan_base || row_hi > cp.plan_end) return hipErrorInvalidValue;
    if (cp.rows_per_chunk == 0 || cp.rows_per_chunk > kC4MaxRows || cp.rows_per_chunk % kC4Slices) return hipErrorInvalidValue;
    if (cp.max_passes < 1 || cp.max_passes > kC4MaxPasses) return hipErrorInvalidValue;
    if (what & 1) {   // prepare: choose the columns and (unless the producing kernel did) write the table(s)
        GNNVC_LAUNCH(k_c4_choose, dim3(1), dim3(64), 0, stream, counts, count_slots, g.n, desc, cp.max_passes);
        GNNVC_LAUNCH(k_c4_compact, dim3(std::min<unsigned>((g.n + 256) / 256, 4096u)), dim3(256), 0, stream,
                           reinterpret_cast<const float4 *>(in), g.n, desc, reinterpret_cast<f32x4 *>(table));
    }
    if (!(what & 2)) return hipGetLastError();
    hipError_t rc0 = hipMemsetAsync(desc + 5, 0, sizeof(uint32_t), stream);   // dirty-row counter
    if (rc0 != hipSuccess) return rc0;
    rc0 = compact_sums(g, cp, desc, table, acc4, row_lo, row_hi, dirty_rows, dirty_cap, stream);
    if (rc0 != hipSuccess) return rc0;
    return compact_fix(g, in, desc, dirty_rows, dirty_cap, agg16, nullptr, stream);
}

// the choice of table columns alone, from counters that cover `rows` rows (a pilot over the first rows of a stage's output)
hipError_t compact_choose(const unsigned long long *counts, int count_slots, uint32_t rows, uint32_t *desc, uint32_t max_passes,
                          hipStream_t stream) {
    GNNVC_LAUNCH(k_c4_choose, dim3(1), dim3(64), 0, stream, counts, count_slots, rows, desc, max_passes);
    return hipGetLastError();
}

// the sums of the chunks that hold rows [row_lo, row_hi) (the dirty-row counter desc[5] is the caller's to reset): one
// launch per pass the plan allows; the launches of passes the device did not choose (desc[0]) leave at once
hipError_t compact_sums(const GraphDev &g, const CompactPlan &cp, uint32_t *desc, const float *table, float *acc4, uint32_t row_lo,
                        uint32_t row_hi, uint32_t *dirty_rows, uint32_t dirty_cap, hipStream_t stream, bool one_round) {
    if (row_hi <= row_lo || g.nnz == 0 || row_lo < cp.plan_base || row_hi > cp.plan_end) return hipErrorInvalidValue;
    if (cp.rows_per_chunk == 0 || cp.rows_per_chunk > kC4MaxRows || cp.rows_per_chunk % kC4Slices) return hipErrorInvalidValue;
    if (cp.rowmap && one_round) return hipErrorInvalidValue;
    const uint32_t slice_rows = cp.rows_per_chunk / kC4Slices;
    const uint32_t nslices = cp.rowmap ? cp.nslices : ((cp.plan_end - cp.plan_base + cp.rows_per_chunk - 1) / cp.rows_per_chunk) * kC4Slices;
    const uint32_t c0 = cp.rowmap ? 0u : (row_lo - cp.plan_base) / cp.rows_per_chunk;
    const uint32_t c1 = cp.rowmap ? nslices / kC4Slices : (row_hi - 1 - cp.plan_base) / cp.rows_per_chunk + 1;
    constexpr size_t lds_max = (size_t)kC4Slices * (kC4SliceRows * 16 + kC4DirtyWords * 4);
    static_assert(lds_max <= 160 * 1024, "LDS budget of k_c4_agg");
    const size_t lds = (size_t)kC4Slices * ((size_t)slice_rows * 16 + kC4DirtyWords * 4);
    static std::atomic<uint64_t> lds_ok0{0}, lds_ok1{0};
    {
        hipError_t rc = allow_dynamic_lds(reinterpret_cast<const void *>(k_c4_agg<0>), (int)lds_max, lds_ok0);
        if (rc == hipSuccess) rc = allow_dynamic_lds(reinterpret_cast<const void *>(k_c4_agg<1>), (int)lds_max, lds_ok1);
        if (rc != hipSuccess) return rc;
    }
    // a persistent grid of one workgroup per CU: they start together and sweep the column blocks together
    const dim3 grid(std::min<uint32_t>(256u, c1 - c0)), block(1024);
    for (uint32_t pass = 0; pass < cp.max_passes; ++pass) {
        const f32x4 *tq = reinterpret_cast<const f32x4 *>(table) + (size_t)pass * ((size_t)g.n + 1);
        f32x4 *aq = reinterpret_cast<f32x4 *>(acc4) + (size_t)pass * g.n;
        if (one_round)
            GNNVC_LAUNCH(k_c4_agg<1>, grid, block, lds, stream, cp.step_ptr, reinterpret_cast<const uint4 *>(cp.steps), cp.entries, tq, aq,
                               g.n, slice_rows, c0 * kC4Slices, std::min(c1 * kC4Slices, nslices), nslices, cp.last_entry, desc,
                               dirty_rows, dirty_cap, cp.block_cols, cp.nblocks, cp.plan_base, cp.plan_end, pass, cp.rowmap);
        else
            GNNVC_LAUNCH(k_c4_agg<0>, grid, block, lds, stream, cp.step_ptr, reinterpret_cast<const uint4 *>(cp.steps), cp.entries, tq, aq,
                               g.n, slice_rows, c0 * kC4Slices, std::min(c1 * kC4Slices, nslices), nslices, cp.last_entry, desc,
                               dirty_rows, dirty_cap, cp.block_cols, cp.nblocks, cp.plan_base, cp.plan_end, pass, cp.rowmap);
    }
    return hipGetLastError();
}

// marks[k] = dirty-row slots handed out so far
hipError_t compact_mark(const uint32_t *desc, uint32_t *marks, uint32_t k, hipStream_t stream) {
    GNNVC_LAUNCH(k_c4_mark, dim3(1), dim3(1), 0, stream, desc, marks, k);
    return hipGetLastError();
}

// full-row aggregates of the dirty rows: all of them (marks == nullptr) or those of slots [marks[0], marks[1]); `blocks`:
// a small grid when the kernel runs beside the aggregation grid (it strides over what it has to do)
hipError_t compact_fix(const GraphDev &g, const float *in, const uint32_t *desc, const uint32_t *dirty_rows, uint32_t dirty_cap,
                       float *agg16, const uint32_t *marks, hipStream_t stream, uint32_t blocks) {
    GNNVC_LAUNCH(k_c4_fix, dim3(std::min<uint32_t>((dirty_cap + 63) / 64, blocks)), dim3(256), 0, stream, g,
                       reinterpret_cast<const float4 *>(in), desc, dirty_rows, dirty_cap, reinterpret_cast<float4 *>(agg16), marks);
    return hipGetLastError();
}

// ---- degree-sorted tile order ------------------------------------------------------------
hipError_t measure_tile_waste(const GraphDev &g, uint32_t row_lo, uint32_t row_hi, uint32_t long_thresh,
                              unsigned long long *sum_max /* [2]: tile maxima, entries of the heavy rows */, hipStream_t stream,
                              uint32_t heavy_from) {
    hipError_t rc = hipMemsetAsync(sum_max, 0, 2 * sizeof(unsigned long long), stream);
    if (rc != hipSuccess || row_hi <= row_lo) return rc;
    GNNVC_LAUNCH(k_tile_waste, dim3(std::min<unsigned>((row_hi - row_lo + 255) / 256, 512u)), dim3(256), 0, stream, g,
                       row_lo, row_hi, long_thresh, sum_max, heavy_from);
    return hipGetLastError();
}

hipError_t degree_histogram(const GraphDev &g, uint32_t row_lo, uint32_t row_hi, uint32_t long_thresh,
                            uint32_t bins, uint32_t *hist, hipStream_t stream, const uint32_t *skip_rowptr, uint32_t skip_from) {
    hipError_t rc = hipMemsetAsync(hist, 0, bins * sizeof(uint32_t), stream);
    if (rc != hipSuccess || row_hi <= row_lo) return rc;
    const unsigned nb = std::min<unsigned>((row_hi - row_lo + 255) / 256, 2048u);
    GNNVC_LAUNCH(k_deg_hist, dim3(nb), dim3(256), 0, stream, g, row_lo, row_hi, long_thresh, bins, hist, skip_rowptr, skip_from);
    return hipGetLastError();
}

hipError_t degree_starts(uint32_t *hist, uint32_t bins, uint32_t *info, hipStream_t stream) {
    if (bins == 0 || bins > 4096) return hipErrorInvalidValue;
    GNNVC_LAUNCH(k_deg_starts, dim3(1), dim3(64), 0, stream, hist, bins, info);
    return hipGetLastError();
}

hipError_t degree_scatter(const GraphDev &g, uint32_t row_lo, uint32_t row_hi, uint32_t long_thresh,
                          uint32_t bins, uint32_t *cursor, uint32_t *vertex, void *meta, hipStream_t stream,
                          const uint32_t *skip_rowptr, uint32_t skip_from) {
    if (row_hi <= row_lo) return hipSuccess;
    GNNVC_LAUNCH(k_deg_scatter, dim3((row_hi - row_lo + 256 * kDegScatterRows - 1) / (256 * kDegScatterRows)), dim3(256), 0, stream, g, row_lo, row_hi,
                       long_thresh, bins, cursor, vertex, reinterpret_cast<uint4 *>(meta), skip_rowptr, skip_from);
    return hipGetLastError();
}

// ---- long rows --------------------------------------------------------------------------
hipError_t find_long_rows(const GraphDev &g, uint32_t thresh, uint32_t *list, uint32_t *count,
                          hipStream_t stream) {
    hipError_t rc = hipMemsetAsync(count, 0, 4 * sizeof(uint32_t), stream);   // (count: 4 words, see k_find_long)
    if (rc != hipSuccess || g.hi() <= g.lo()) return rc;
    GNNVC_LAUNCH(k_find_long, dim3(std::min<unsigned>((g.hi() - g.lo() + 1023) / 1024, 2048u)), dim3(256), 0, stream, g, thresh, list, count);
    return hipGetLastError();
}

hipError_t derive_tails(const GraphDev &old_g, const uint32_t *old_row, uint32_t n_new, const uint32_t *rowptr_new, uint32_t *new_of,
                        uint32_t *tail, uint32_t *bad, hipStream_t stream) {
    hipError_t rc = hipMemsetAsync(new_of, 0xFF, (size_t)std::max<uint32_t>(old_g.n, 1u) * sizeof(uint32_t), stream);
    if (rc == hipSuccess) rc = hipMemsetAsync(bad, 0, sizeof(uint32_t), stream);
    if (rc != hipSuccess || n_new == 0) return rc;
    const dim3 grid((n_new + 255) / 256), block(256);
    GNNVC_LAUNCH(k_derive_map, grid, block, 0, stream, old_row, n_new, old_g.n, new_of, bad);
    GNNVC_LAUNCH(k_derive_tails, grid, block, 0, stream, old_g, old_row, new_of, rowptr_new, n_new, tail, bad);
    return hipGetLastError();
}

hipError_t derive_fill(const GraphDev &old_g, const uint32_t *old_row, const uint32_t *new_of, const uint32_t *rowptr_new,
                       const uint32_t *tail_ptr, const uint32_t *tail_cols, uint32_t n_new, uint32_t *col_new, hipStream_t stream) {
    if (n_new == 0) return hipSuccess;
    GNNVC_LAUNCH(k_derive_fill, dim3((n_new + 255) / 256), dim3(256), 0, stream, old_g, old_row, new_of, rowptr_new, tail_ptr,
                 tail_cols, n_new, col_new);
    return hipGetLastError();
}

hipError_t row_hashes(const GraphDev &g, unsigned long long *out, hipStream_t stream) {
    if (g.hi() <= g.lo()) return hipSuccess;
    GNNVC_LAUNCH(k_row_hashes, dim3((g.hi() - g.lo() + 255) / 256), dim3(256), 0, stream, g, out);
    return hipGetLastError();
}

hipError_t find_giant_rows(const GraphDev &g, const uint32_t *list, uint32_t n_long, uint32_t thresh, void *meta, uint32_t *count,
                           hipStream_t stream) {
    hipError_t rc = hipMemsetAsync(count, 0, sizeof(uint32_t), stream);
    if (rc != hipSuccess || n_long == 0) return rc;
    GNNVC_LAUNCH(k_find_giant, dim3((n_long + 255) / 256), dim3(256), 0, stream, g, list, n_long, thresh,
                       reinterpret_cast<uint4 *>(meta), count);
    return hipGetLastError();
}

uint32_t giant_window() { return kGiantWin; }
uint32_t giant_block() { return kGiantBlk; }

hipError_t launch_giant_stage(const StagePlan &sp, const GraphDev &g, float ws, const float *params, const float *in, float *out,
                              float *logits, uint32_t row_lo, uint32_t row_hi, const GiantRows &gr, hipStream_t stream, uint32_t min_deg,
                              int part) {
    // part: 0 = everything on `stream`, 1 = the gather only, 2 = everything behind the gather (the engine puts the gather — a
    // throughput kernel, 0.03 ms alone — on the main queue AHEAD of the tile kernel and the walk on the side queue: queued
    // beside the tile kernel it waited 0.2 - 0.3 ms for free slots at the head of the chain a stage waits for)
    if (gr.n == 0 || row_hi <= row_lo) return hipSuccess;
    const uint4 *meta = reinterpret_cast<const uint4 *>(gr.meta);
    const uint32_t F = sp.f == 16 ? 16u : 1u;
    const uint32_t *pr = sp.f == 16 ? g.prp : nullptr, *pb = sp.f == 16 ? g.prune_bad : nullptr;   // (pruned adjacency: 16-wide stages only)
    if (sp.f != 16 && sp.f != 1) return hipErrorInvalidValue;
    if (part != 2) {
        if (sp.f == 16)
            GNNVC_LAUNCH(k_giant_gather16, dim3(gr.blocks), dim3(256), 0, stream, g, reinterpret_cast<const float4 *>(in), gr.slab,
                               meta, gr.off, gr.n, row_lo, row_hi, min_deg);
        else
            GNNVC_LAUNCH(k_giant_gather1, dim3(gr.blocks), dim3(256), 0, stream, g, in, gr.slab, meta, gr.off, gr.n, row_lo, row_hi);
        if (part == 1) return hipGetLastError();
    }
    {
        const bool seg = gr.segsum && gr.segmap && gr.maxseg > 1;   // one stream on several waves (see k_giant_segmap)
        if (seg) {
            const dim3 sgrid(gr.n * F * giant_seg_blocks(gr.maxseg)), sblock(64 * kGiantSegWaves);
            GNNVC_LAUNCH(k_giant_segsum, sgrid, sblock, 0, stream, gr.slab, meta, gr.off, F, gr.maxseg, gr.segsum,
                         row_lo, row_hi, pr, pb, min_deg);
            GNNVC_LAUNCH(k_giant_segmap, sgrid, sblock, 0, stream, gr.slab, meta, gr.off, F, gr.maxseg, gr.segsum,
                         reinterpret_cast<uint4 *>(gr.segmap), row_lo, row_hi, pr, pb, min_deg);
        }
        GNNVC_LAUNCH(k_giant_sum, dim3(gr.n * F), dim3(64), 0, stream, gr.slab, meta, gr.off, F, gr.agg, row_lo, row_hi, pr, pb,
                     seg ? reinterpret_cast<const uint4 *>(gr.segmap) : nullptr, gr.maxseg, min_deg);
    }
    const float *P = params + sp.param_offset;
    const dim3 grid(gr.n), block(64);
    switch (sp.variant) {
    case 0: GNNVC_LAUNCH(k_giant_dense<0>, grid, block, 0, stream, g, ws, in, out, logits, P, meta, gr.n, gr.agg, row_lo, row_hi, min_deg); break;
    case 1: GNNVC_LAUNCH(k_giant_dense<1>, grid, block, 0, stream, g, ws, in, out, logits, P, meta, gr.n, gr.agg, row_lo, row_hi, min_deg); break;
    case 2: GNNVC_LAUNCH(k_giant_dense<2>, grid, block, 0, stream, g, ws, in, out, logits, P, meta, gr.n, gr.agg, row_lo, row_hi, min_deg); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// sums[i] = the sequential fp32 sum of streams[i * lpad .. + len) (lpad = len rounded up to giant_window()); meta needs
// streams + 1 entries, off streams entries
uint32_t giant_segments(uint32_t len) {   // segments of a stream of `len` addends
    const uint32_t nwin = (len + kGiantWin - 1) / kGiantWin;
    return (nwin + kGiantSeg - 1) / kGiantSeg;
}

// mode: 0 = the stream on several waves (segsum / segmap: streams x giant_segments(len) floats / uint4, may be null:
// then as mode 2), 2 = one wave walks the whole stream; the same exact sum either way
hipError_t stream_sums(const float *streams_dev, uint32_t streams, uint32_t len, void *meta, unsigned long long *off, float *agg,
                       int mode, hipStream_t stream, float *segsum, void *segmap) {
    if (!streams) return hipSuccess;
    GNNVC_LAUNCH(k_stream_meta, dim3((streams + 256) / 256), dim3(256), 0, stream, reinterpret_cast<uint4 *>(meta), off, streams, len);
    const uint4 *mt = reinterpret_cast<const uint4 *>(meta);
    const uint32_t *none = nullptr;
    if (mode != 0 && mode != 2) return hipErrorInvalidValue;
    const uint32_t maxseg = giant_segments(len);
    const bool seg = mode == 0 && segsum && segmap && maxseg > 1;
    if (seg) {
        const dim3 sgrid(streams * giant_seg_blocks(maxseg)), sblock(64 * kGiantSegWaves);
        GNNVC_LAUNCH(k_giant_segsum, sgrid, sblock, 0, stream, streams_dev, mt, off, 1u, maxseg, segsum, 0u, 1u, none, none, 0u);
        GNNVC_LAUNCH(k_giant_segmap, sgrid, sblock, 0, stream, streams_dev, mt, off, 1u, maxseg, segsum,
                     reinterpret_cast<uint4 *>(segmap), 0u, 1u, none, none, 0u);
    }
    GNNVC_LAUNCH(k_giant_sum, dim3(streams), dim3(64), 0, stream, streams_dev, mt, off, 1u, agg, 0u, 1u, none, none,
                 seg ? reinterpret_cast<const uint4 *>(segmap) : nullptr, maxseg, 0u);
    return hipGetLastError();
}

hipError_t launch_long_stage(const StagePlan &sp, const GraphDev &g, float ws, const float *params,
                             const float *in, float *out, float *logits, uint32_t row_lo, uint32_t row_hi,
                             const uint32_t *list, uint32_t n_long, uint32_t min_deg, uint32_t max_deg, hipStream_t stream) {
    if (n_long == 0 || row_hi <= row_lo) return hipSuccess;
    const float *P = params + sp.param_offset;
    const dim3 grid(n_long), block(256);
    switch (sp.variant) {
    case 0:
        GNNVC_LAUNCH((k_long_f1<32, 32, 16>), grid, block, 0, stream, g, ws, in, out, P, row_lo, row_hi, list,
                           min_deg, max_deg);
        break;
    case 1:
        if (g.zero_bits && g.keep_col) {
            GNNVC_LAUNCH(k_long_lists, grid, block, 0, stream, g, row_lo, row_hi, list, min_deg, max_deg);
            GNNVC_LAUNCH((k_long_f16<32, 32, 16, false, true>), grid, block, 0, stream, g, ws,
                               reinterpret_cast<const float4 *>(in), out, nullptr, P, row_lo, row_hi, list, min_deg, max_deg);
        } else
            GNNVC_LAUNCH((k_long_f16<32, 32, 16, false>), grid, block, 0, stream, g, ws,
                               reinterpret_cast<const float4 *>(in), out, nullptr, P, row_lo, row_hi, list, min_deg, max_deg);
        break;
    case 2:
        if (g.zero_bits && g.keep_col) {
            GNNVC_LAUNCH(k_long_lists, grid, block, 0, stream, g, row_lo, row_hi, list, min_deg, max_deg);
            GNNVC_LAUNCH((k_long_f16<32, 16, 1, true, true>), grid, block, 0, stream, g, ws,
                               reinterpret_cast<const float4 *>(in), out, logits, P, row_lo, row_hi, list, min_deg, max_deg);
        } else
            GNNVC_LAUNCH((k_long_f16<32, 16, 1, true>), grid, block, 0, stream, g, ws,
                               reinterpret_cast<const float4 *>(in), out, logits, P, row_lo, row_hi, list, min_deg, max_deg);
        break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t live_columns(const float *feat, size_t rows, uint32_t *mask, hipStream_t stream) {
    hipError_t rc = hipMemsetAsync(mask, 0, sizeof(uint32_t), stream);
    if (rc != hipSuccess || rows == 0) return rc;
    const size_t quads = rows * 4;
    const unsigned blocks = (unsigned)std::min<size_t>((quads + 255) / 256, 4096);
    GNNVC_LAUNCH(k_live_columns, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float4 *>(feat), quads, mask);
    return hipGetLastError();
}

hipError_t score_keys(const float *scores, size_t n, float *keys, uint8_t *above_half, hipStream_t stream) {
    if (!n) return hipSuccess;
    GNNVC_LAUNCH(k_score_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, scores, n, keys, above_half);
    return hipGetLastError();
}

hipError_t column_counts(const float *feat, size_t rows, unsigned long long *counts, hipStream_t stream) {
    hipError_t rc = hipMemsetAsync(counts, 0, 16 * sizeof(unsigned long long), stream);
    if (rc != hipSuccess || rows == 0) return rc;
    const size_t quads = rows * 4;
    const unsigned blocks = (unsigned)std::min<size_t>((quads + 255) / 256, 2048);
    GNNVC_LAUNCH(k_column_counts, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float4 *>(feat), quads,
                       counts);
    return hipGetLastError();
}

hipError_t pack_rows(const float *feat, uint32_t row_lo, uint32_t row_hi, uint32_t mask, uint32_t kp, float *dense,
                     uint32_t *exc, uint32_t cap, uint32_t *flag, hipStream_t stream) {
    if (exc) {
        hipError_t rc = hipMemsetAsync(exc, 0, 4 * sizeof(uint32_t), stream);
        if (rc != hipSuccess) return rc;
    }
    if (row_hi <= row_lo) return hipSuccess;
    ColumnMap map{};
    uint32_t k = 0;
    for (uint32_t c = 0; c < 16; ++c)
        if (mask >> c & 1u) map.col[k++] = (uint8_t)c;
    if (k > kp || kp < 4 || kp > 16) return hipErrorInvalidValue;
    const size_t total = (size_t)(row_hi - row_lo) * kp;
    GNNVC_LAUNCH(k_pack_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, feat, row_lo, row_hi,
                       mask, k, kp, map, dense, exc, cap, flag);
    return hipGetLastError();
}

hipError_t unpack_rows(const float *dense, const uint32_t *exc, uint32_t cap, uint32_t row_lo, uint32_t row_hi,
                       uint32_t mask, uint32_t kp, float *feat, hipStream_t stream) {
    if (row_hi <= row_lo) return hipSuccess;
    if (kp < 4 || kp > 16 || (uint32_t)__builtin_popcount(mask & 0xFFFFu) > kp) return hipErrorInvalidValue;
    const size_t total = (size_t)(row_hi - row_lo) * 4;
    GNNVC_LAUNCH(k_unpack_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dense, row_lo, row_hi,
                       mask & 0xFFFFu, kp, feat);
    if (exc && cap) {
        const unsigned blocks = std::min<unsigned>((cap + 255) / 256, 1024);
        GNNVC_LAUNCH(k_unpack_exceptions, dim3(blocks), dim3(256), 0, stream, exc, cap, row_lo, row_hi, feat);
    }
    return hipGetLastError();
}

hipError_t unpack_gathered(const float *buf, uint32_t world, uint32_t skip, size_t piece_words, uint32_t dense_rows,
                           uint32_t cap, uint32_t per, uint32_t off, uint32_t size, uint32_t n, uint32_t mask, uint32_t kp,
                           float *feat, hipStream_t stream) {
    if (!world || !size) return hipSuccess;
    if (kp < 4 || kp > 16 || (uint32_t)__builtin_popcount(mask & 0xFFFFu) > kp || size > dense_rows) return hipErrorInvalidValue;
    const unsigned bx = (unsigned)std::min<size_t>(((size_t)size * 4 + 255) / 256, 2048);
    GNNVC_LAUNCH(k_unpack_gathered, dim3(bx, world), dim3(256), 0, stream, buf, skip, piece_words, per, off, size, n,
                       mask & 0xFFFFu, kp, feat);
    if (cap)
        GNNVC_LAUNCH(k_unpack_gathered_exceptions, dim3(std::min<unsigned>((cap + 255) / 256, 64), world), dim3(256), 0,
                           stream, buf, skip, piece_words, dense_rows, cap, per, off, size, n, kp, feat);
    return hipGetLastError();
}

hipError_t push_piece(const float *region, uint32_t rows, uint32_t kp, uint32_t cap, uint32_t n_dst, float *const *dst, hipStream_t stream) {
    if (!n_dst) return hipSuccess;
    if (n_dst > 64 || kp < 4 || kp > 16 || (kp & 3u)) return hipErrorInvalidValue;
    PushDst pd{};
    for (uint32_t i = 0; i < n_dst; ++i) pd.p[i] = dst[i];
    const size_t dense_words = (size_t)rows * kp;
    // (a modest grid per destination: the stores are paced by the links, and the CUs are wanted by the next piece's kernels)
    const unsigned bx = (unsigned)std::min<size_t>(std::max<size_t>((dense_words / 4 + 255) / 256, 1), 64);
    GNNVC_LAUNCH(k_push_piece, dim3(bx, n_dst), dim3(256), 0, stream, region, dense_words, cap, pd);
    return hipGetLastError();
}

hipError_t unpack_pieces(const UnpackPiece *pieces, uint32_t n_pieces, uint32_t cap, uint32_t mask, uint32_t kp, float *feat, hipStream_t stream) {
    if (!n_pieces) return hipSuccess;
    if (n_pieces > 64 || kp < 4 || kp > 16 || (uint32_t)__builtin_popcount(mask & 0xFFFFu) > kp) return hipErrorInvalidValue;
    PieceList pl{};
    uint32_t most = 0;
    for (uint32_t i = 0; i < n_pieces; ++i) {
        if (pieces[i].row_hi < pieces[i].row_lo) return hipErrorInvalidValue;
        pl.d[i] = PieceRef{pieces[i].region, pieces[i].row_lo, pieces[i].row_hi};
        most = std::max(most, pieces[i].row_hi - pieces[i].row_lo);
    }
    if (!most) return hipSuccess;
    const unsigned bx = (unsigned)std::min<size_t>(((size_t)most * 4 + 255) / 256, 2048);
    GNNVC_LAUNCH(k_unpack_pieces, dim3(bx, n_pieces), dim3(256), 0, stream, pl, mask & 0xFFFFu, kp, feat);
    if (cap)
        GNNVC_LAUNCH(k_unpack_pieces_exceptions, dim3(std::min<unsigned>((cap + 255) / 256, 64), n_pieces), dim3(256), 0, stream, pl, cap, kp, feat);
    return hipGetLastError();
}

hipError_t launch_reduction_flags(const GraphDev &g, uint32_t max_degree, uint8_t *flags, hipStream_t stream) {
    if (g.n == 0) return hipSuccess;
    GNNVC_LAUNCH(k_reduction_flags, dim3((g.n + 255) / 256), dim3(256), 0, stream, g, max_degree, flags);
    return hipGetLastError();
}

hipError_t validate_graph(const GraphDev &g, uint32_t *flags, hipStream_t stream) {
    hipError_t rc = hipMemsetAsync(flags, 0, sizeof(uint32_t), stream);
    if (rc != hipSuccess || g.n == 0) return rc;
    GNNVC_LAUNCH(k_validate_graph, dim3(2048), dim3(256), 0, stream, g, flags);
    return hipGetLastError();
}

hipError_t validate_rowptr(const GraphDev &g, uint32_t *flags, hipStream_t stream) {
    hipError_t rc = hipMemsetAsync(flags, 0, sizeof(uint32_t), stream);
    if (rc != hipSuccess || g.n == 0) return rc;
    GNNVC_LAUNCH(k_validate_rowptr, dim3(1024), dim3(256), 0, stream, g, flags);
    return hipGetLastError();
}

hipError_t narrow_rowptr(const void *in_u64, uint32_t *out, size_t count, hipStream_t stream) {
    if (!count) return hipSuccess;
    GNNVC_LAUNCH(k_narrow_rowptr, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const unsigned long long *>(in_u64), out, count);
    return hipGetLastError();
}

hipError_t launch_zero_pad_row(float *buf, uint32_t n, uint32_t width, hipStream_t stream) {
    GNNVC_LAUNCH(k_zero_row, dim3(1), dim3(64), 0, stream, buf, n, width);
    return hipGetLastError();
}

}  // namespace gnnvc

#if GNNVC_PHASE_PROBE
extern "C" int gnnvc_debug_probe(void *buf, int kind) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(gnnvc::gnnvc_probe_buf), &buf, sizeof buf) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(gnnvc::gnnvc_probe_kind), &kind, sizeof kind) != hipSuccess) return -2;
    return 0;
}
#endif
