// Prototype (experiment, not product): compact-table aggregation with WAVE-OWNED rows.
//
// k_c4_agg (c4_gather.hip) parks every gathered row in LDS and lets the thread at the head of a row's run add
// it, with two workgroup barriers per 2048-entry step: 105 G rows/s, while the same L2-resident gather alone
// runs at 217 G rows/s (tools/microbench/l2_gather.hip).  Here every wave owns a slice of the chunk's rows
// (its sums live in its own part of LDS) and gets its own entry stream, regrouped per (wave slice, column
// block) into steps of <= 256 entries = 4 sub-batches of one entry per lane.  The lane that gathers a row is
// the lane that adds it; a run of entries of the same row sits on adjacent lanes and is folded into the
// run's first lane with whole-wave DPP shifts, in order.  LDS operations of one wave execute in order, so
// no barrier is needed anywhere, and without the parking buffers the sums of ~9800 rows fit per CU:
// 4 sweeps of the table instead of 6 for the metric graph.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return -1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef BURN_PAD
#define BURN_PAD 896
#endif
constexpr int kShift = 18;
#ifndef KSUB
#define KSUB 4
#endif
constexpr int kSub = KSUB;          // sub-batches per step

__device__ __forceinline__ uint32_t shl1(uint32_t x) {   // lane i <- lane i + 1, lane 63 <- 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t shr1(uint32_t x) {   // lane i <- lane i - 1, lane 0 <- 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ f32x4 shl1(f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __uint_as_float(shl1(__float_as_uint(v[i])));
    return r;
}

template <int DEPTH>   // gathers are issued DEPTH steps ahead of their use (1 or 2)
__global__ __launch_bounds__(1024) void k_c4w_agg(const uint32_t *__restrict__ step_ptr, const uint4 *__restrict__ steps,
                                                  const uint32_t *__restrict__ entries, const f32x4 *__restrict__ table,
                                                  f32x4 *__restrict__ agg, uint32_t n, uint32_t Rw, uint32_t Bc, uint32_t last_entry,
                                                  uint32_t nwc, uint32_t nblocks /* 0 = no pacing barriers */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 *A = reinterpret_cast<f32x4 *>(smem) + wave * Rw;              // this wave's sums
    const uint32_t pf_slice = (blockIdx.x >> 3) * 16 + wave;
    float pf_acc = 0.0f, pf_val = 0.0f;
    for (uint32_t base = blockIdx.x * 16; base < nwc; base += gridDim.x * 16) {
        const uint32_t wc = base + wave < nwc ? base + wave : nwc;    // nwc: an empty slice (step_ptr[nwc] == step_ptr[nwc + 1])
        const uint32_t row0 = wc * Rw;
        uint32_t cur_blk = 0;
        for (uint32_t i = lane; i < Rw; i += 64) A[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const uint32_t st0 = __builtin_amdgcn_readfirstlane(step_ptr[wc]);
        const int nsteps = (int)(__builtin_amdgcn_readfirstlane(step_ptr[wc + 1]) - st0);   // a multiple of 4
        uint32_t e[4][kSub];                 // entries of step s in slot s & 3
        f32x4 v[DEPTH + 1][kSub];            // gathered rows of step s in slot s % (DEPTH + 1)  (indexing below is static)
        uint32_t cb[4], cnt[4], bk[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            cb[a] = 0;
            bk[a] = 0;
            cnt[a] = 0;
#pragma unroll
            for (int k = 0; k < kSub; ++k) e[a][k] = 0;
        }
#pragma unroll
        for (int a = 0; a <= DEPTH; ++a)
#pragma unroll
            for (int k = 0; k < kSub; ++k) v[a][k] = f32x4{0, 0, 0, 0};
        uint4 dsc = steps[st0];              // descriptor of the step whose entries are loaded next
        int dsc_step = 0;
        // step s: descriptor loaded at time s - DEPTH - 2, entries at s - DEPTH - 1, gather at s - DEPTH, sums at s.
        // The v ring is indexed statically, so the loop is unrolled over lcm(4, DEPTH + 1) = 4 or 12 steps.
        constexpr int kUnroll = (DEPTH + 1 == 3) ? 12 : 4;
        const int lead = DEPTH + 1;          // entries are loaded this many steps ahead
        for (int u = -kUnroll; u < nsteps; u += kUnroll) {
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) {
                const int s = u + j;         // the step whose sums are done now
                if (s + lead < 0) continue;  // (warm-up: nothing to load yet)
                // ---- entries of step s + lead, descriptor of the one after
                {
                    const int se = s + lead;
                    const int slot = (j + lead) & 3;
                    const bool live = se >= 0 && se < nsteps && dsc_step == se;
                    const uint32_t first = dsc.y, count = live ? dsc.z : 0u;
                    cb[slot] = dsc.x * Bc;
                    bk[slot] = dsc.x;
                    cnt[slot] = count;
#pragma unroll
                    for (int k = 0; k < kSub; ++k) {
                        const uint32_t x = first + 64u * k + lane;
#ifdef PREDICATE
                        if (64u * k + lane < count) e[slot][k] = __builtin_nontemporal_load(&entries[x]);
#else
                        e[slot][k] = __builtin_nontemporal_load(&entries[x < last_entry ? x : last_entry]);
#endif
                    }
#ifdef PREFETCH
                    {   // pull the column block these entries point into towards this XCD's L2 with coalesced reads:
                        // the 512 waves of an XCD (workgroup g runs on XCD g % 8) take one slice each
                        const uint32_t per_slice = ((Bc + 7) / 8 + 511) / 512;             // 128-byte lines per wave
                        const uint32_t ln = pf_slice * per_slice + (lane < per_slice ? lane : 0);
                        const size_t prow = (size_t)(uint32_t)((int)dsc.x + (PREFETCH - 10)) * Bc + (size_t)ln * 8;   // -DPREFETCH=10: the block itself
                        pf_acc += pf_val;                                                  // (the previous one, issued a step ago)
                        pf_val = reinterpret_cast<const float *>(table + (prow < n ? prow : n))[0];
                    }
#endif
                    const int nx = se + 1;
                    const int nxc = nx < 0 ? 0 : (nx < nsteps ? nx : nsteps - 1);
                    dsc = steps[st0 + (uint32_t)nxc];
                    dsc_step = nx;
                }
                // ---- gather step s + DEPTH
                {
                    const int slot = (j + DEPTH) & 3;
                    const int vs = (j + DEPTH) % (DEPTH + 1);
#pragma unroll
                    for (int k = 0; k < kSub; ++k) {
                        const uint32_t c = cb[slot] + (e[slot][k] & ((1u << kShift) - 1u));
                        const bool in = 64u * k + lane < cnt[slot];
#ifdef PREDICATE
                        if (in) v[vs][k] = table[c < n ? c : n];
#else
                        v[vs][k] = table[(in && c < n) ? c : n];
#endif
                    }
                }
                // ---- sums of step s
                if (s >= 0) {
                    const int slot = j & 3;
                    const int vs = j % (DEPTH + 1);
                    // pacing: the 16 waves of the workgroup enter a column block together (every wave passes the
                    // same number of barriers per chunk round: one per block)
                    if (nblocks && cnt[slot])
                        while (cur_blk < bk[slot]) {
                            __syncthreads();
                            ++cur_blk;
                        }
#pragma unroll
                    for (int k = 0; k < kSub; ++k) {
                        if (64u * k >= cnt[slot]) break;                      // uniform
                        const bool in = 64u * k + lane < cnt[slot];
                        const uint32_t row = e[slot][k] >> kShift;
                        const f32x4 val = v[vs][k];
                        const uint32_t prow = shr1(row);
                        const bool head = in && (lane == 0 || prow != row);
                        f32x4 sum = A[row < Rw ? row : 0] + val;
                        // fold the rest of the run (adjacent lanes, same row) into its head, in order
                        f32x4 nv = val;
                        uint32_t nrow = row, nin = in ? 1u : 0u;
                        bool cont = head;
                        for (;;) {
                            nv = shl1(nv);
                            nrow = shl1(nrow);
                            nin = shl1(nin);
                            cont = cont && nin && nrow == row;
                            if (!__any(cont)) break;
                            if (cont) sum += nv;
                        }
                        if (head) A[row] = sum;
                    }
                }
            }
        }
        while (cur_blk < nblocks) {
            __syncthreads();
            ++cur_blk;
        }
        for (uint32_t i = lane; i < Rw && row0 + i < n; i += 64) agg[row0 + i] = A[i];
    }
    if (pf_acc + pf_val == 1.2345e-30f) agg[0][0] = pf_acc;   // keeps the prefetch loads alive
}


// ---- second form: a lane takes FOUR CONSECUTIVE entries of the step (one 16-byte entry load per lane and step
// instead of four 4-byte ones: entry loads cost the texture-address unit about as much per instruction as the
// gathers' lanes), gathers their four rows, and the runs are folded lane-locally; a run that reaches the end
// of the lane continues with the next lane's entries (one whole-wave DPP shift of its 4 rows and 4 values,
// more shifts only for runs longer than a lane).  Rows of a step are ascending, so every row written in a
// step is written by exactly one lane: the four LDS reads of a lane are issued together.
constexpr uint32_t kNoRow = 0x3FFFu;   // row field of a slot past the step's end (never a real row)

__device__ __forceinline__ f32x4 sel(bool c, f32x4 v) {   // c ? v : +0 (all values here are >= +0: x + 0 == x)
    return f32x4{c ? v[0] : 0.0f, c ? v[1] : 0.0f, c ? v[2] : 0.0f, c ? v[3] : 0.0f};
}

__global__ __launch_bounds__(1024) void k_c4v_agg(const uint32_t *__restrict__ step_ptr, const uint4 *__restrict__ steps,
                                                  const uint32_t *__restrict__ entries, const f32x4 *__restrict__ table,
                                                  f32x4 *__restrict__ agg, uint32_t n, uint32_t Rw, uint32_t Bc, uint32_t last_entry,
                                                  uint32_t nwc, uint32_t nblocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 *A = reinterpret_cast<f32x4 *>(smem) + wave * Rw;
    for (uint32_t base = blockIdx.x * 16; base < nwc; base += gridDim.x * 16) {
        const uint32_t wc = base + wave < nwc ? base + wave : nwc;
        const uint32_t row0 = wc * Rw;
        uint32_t cur_blk = 0;
        for (uint32_t i = lane; i < Rw; i += 64) A[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const uint32_t st0 = __builtin_amdgcn_readfirstlane(step_ptr[wc]);
        const int nsteps = (int)(__builtin_amdgcn_readfirstlane(step_ptr[wc + 1]) - st0);   // a multiple of 4
        uint4 e[4];                          // entries of step s in slot s & 3
        f32x4 v[2][4];                       // gathered rows of step s in slot s & 1
        uint32_t cb[4], cnt[4], bk[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            cb[a] = 0;
            bk[a] = 0;
            cnt[a] = 0;
            e[a] = uint4{0, 0, 0, 0};
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[a][k] = f32x4{0, 0, 0, 0};
        uint4 dsc = steps[st0];
        int dsc_step = 0;
        for (int u = -4; u < nsteps; u += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = u + j;
                {   // entries of step s + 2, descriptor of step s + 3
                    const int se = s + 2;
                    const int slot = (j + 2) & 3;
                    const bool live = se >= 0 && se < nsteps && dsc_step == se;
                    const uint32_t first = dsc.y, count = live ? dsc.z : 0u;
                    cb[slot] = dsc.x * Bc;
                    bk[slot] = dsc.x;
                    cnt[slot] = count;
                    const uint32_t x = first + 4u * lane;                    // first is a multiple of 4
                    {
                        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(entries + (x < last_entry ? x : last_entry)));
                        e[slot] = uint4{q[0], q[1], q[2], q[3]};
                    }
                    const int nx = se + 1;
                    const int nxc = nx < 0 ? 0 : (nx < nsteps ? nx : nsteps - 1);
                    dsc = steps[st0 + (uint32_t)nxc];
                    dsc_step = nx;
                }
                {   // gather step s + 1
                    const int slot = (j + 1) & 3;
                    const int vs = (j + 1) & 1;
                    const uint32_t w[4] = {e[slot].x, e[slot].y, e[slot].z, e[slot].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t c = cb[slot] + (w[k] & ((1u << kShift) - 1u));
                        const bool in = 4u * lane + k < cnt[slot];
                        v[vs][k] = table[(in && c < n) ? c : n];
                    }
                }
                if (s >= 0) {   // sums of step s
                    const int slot = j & 3;
                    const int vs = j & 1;
                    if (nblocks && cnt[slot])
                        while (cur_blk < bk[slot]) {
                            __syncthreads();
                            ++cur_blk;
                        }
                    if (cnt[slot]) {
                        const uint32_t w[4] = {e[slot].x, e[slot].y, e[slot].z, e[slot].w};
                        uint32_t r[4];
                        f32x4 a[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            r[k] = (4u * lane + k < cnt[slot]) ? (w[k] >> kShift) : kNoRow;
                            a[k] = A[r[k] < Rw ? r[k] : 0];
                        }
                        const f32x4 *val = v[vs];
                        uint32_t prev3 = shr1(r[3]);
                        if (lane == 0) prev3 = kNoRow - 1;
                        const bool h0 = r[0] != kNoRow && r[0] != prev3;
                        const bool h1 = r[1] != kNoRow && r[1] != r[0];
                        const bool h2 = r[2] != kNoRow && r[2] != r[1];
                        const bool h3 = r[3] != kNoRow && r[3] != r[2];
                        // the next lane's rows and values
                        uint32_t nr[4];
                        f32x4 nv[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            nr[k] = shl1(r[k]);
                            nv[k] = shl1(val[k]);
                            if (lane == 63) nr[k] = kNoRow;
                        }
                        // lane-local folds (rows ascending: equal rows are adjacent)
                        f32x4 s0 = a[0] + val[0];
                        s0 += sel(r[1] == r[0], val[1]);
                        s0 += sel(r[2] == r[0], val[2]);
                        s0 += sel(r[3] == r[0], val[3]);
                        f32x4 s1 = a[1] + val[1];
                        s1 += sel(r[2] == r[1], val[2]);
                        s1 += sel(r[3] == r[1], val[3]);
                        f32x4 s2 = a[2] + val[2];
                        s2 += sel(r[3] == r[2], val[3]);
                        f32x4 s3 = a[3] + val[3];
                        // the run that holds this lane's last entry goes on in the next lane(s); its first entry is here
                        // iff one of this lane's run starts has its row
                        const bool o0 = h0 && r[0] == r[3], o1 = h1 && r[1] == r[3], o2 = h2 && r[2] == r[3], o3 = h3;
                        const bool own = (o0 || o1 || o2 || o3) && r[3] != kNoRow;
                        f32x4 x = o0 ? s0 : o1 ? s1 : o2 ? s2 : s3;
                        bool more = own;
                        for (;;) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) x += sel(more && nr[k] == r[3], nv[k]);
                            more = more && nr[3] == r[3];
                            if (!__any(more)) break;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {   // one lane further
                                nr[k] = shl1(nr[k]);
                                nv[k] = shl1(nv[k]);
                                if (lane == 63) nr[k] = kNoRow;
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
                }
            }
        }
        while (cur_blk < nblocks) {
            __syncthreads();
            ++cur_blk;
        }
        for (uint32_t i = lane; i < Rw && row0 + i < n; i += 64) agg[row0 + i] = A[i];
    }
}

extern "C" int c4w_agg(const uint32_t *step_ptr, const void *steps, const uint32_t *entries, const void *table, void *agg,
                       uint32_t n, uint32_t Rw, uint32_t Bc, uint32_t nwc, uint32_t nnz, uint32_t grid, uint32_t depth, uint32_t nblocks, void *stream) {
    const size_t lds = (size_t)Rw * 16 * 16;
    if (lds > 163840) return -3;
    if (depth == 4) {   // the lane-contiguous form
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_c4v_agg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_c4v_agg, dim3(grid), dim3(1024), lds, (hipStream_t)stream, step_ptr, reinterpret_cast<const uint4 *>(steps),
                           entries, reinterpret_cast<const f32x4 *>(table), reinterpret_cast<f32x4 *>(agg), n, Rw, Bc, nnz - 4, nwc, nblocks);
    } else if (depth == 2) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_c4w_agg<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_c4w_agg<2>, dim3(grid), dim3(1024), lds, (hipStream_t)stream, step_ptr, reinterpret_cast<const uint4 *>(steps),
                           entries, reinterpret_cast<const f32x4 *>(table), reinterpret_cast<f32x4 *>(agg), n, Rw, Bc, nnz - 1, nwc, nblocks);
    } else {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_c4w_agg<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_c4w_agg<1>, dim3(grid), dim3(1024), lds, (hipStream_t)stream, step_ptr, reinterpret_cast<const uint4 *>(steps),
                           entries, reinterpret_cast<const f32x4 *>(table), reinterpret_cast<f32x4 *>(agg), n, Rw, Bc, nnz - 1, nwc, nblocks);
    }
    CK(hipGetLastError());
    return 0;
}

// ---- co-residency experiment: a VALU-bound kernel without (much) LDS beside the aggregation kernel, which
// holds nearly all of a CU's LDS but only half of its wave slots and registers
__global__ __launch_bounds__(256) void k_valu_burn(float *__restrict__ out, int iters, float seed) {
    __shared__ float pad[BURN_PAD];                    // 3.5 KiB: at most one such workgroup fits beside k_c4v_agg's 156.4 KiB
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
    const float b = seed * 0.5f, c = 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], b, c);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (threadIdx.x == 0) pad[0] = s;
    if (s == 1.2345f) out[blockIdx.x * 256 + threadIdx.x] = s + pad[threadIdx.x & 7];
}

extern "C" int valu_burn(float *out, int blocks, int iters, void *stream) {
    hipLaunchKernelGGL(k_valu_burn, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0f);
    CK(hipGetLastError());
    return 0;
}
