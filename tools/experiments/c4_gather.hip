// Prototype (experiment, not product): F = 16 aggregation when only K = 4 feature columns are live.
//
// The neighbours' rows come from a COMPACT table (4 floats per vertex, 16 B) instead of the 64-byte rows.
// Rows are cut into chunks (one 1024-thread workgroup each, 4 sums per row in LDS), columns into blocks of
// 131072 vertices = 2 MiB of the compact table.  All workgroups sweep the blocks in the same order at about
// the same pace, so the block being read stays in every XCD's L2 and a gather is an L2 hit instead of a
// fabric request.  Entries are regrouped per (chunk, block) into steps of <= 2048, as in the LDS-table plan
// of the F = 1 stage; the gathered values of a step are parked in LDS so that the thread at the head of a
// row's run adds the whole run in order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return -1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kStep = 2048;

__global__ __launch_bounds__(1024) void k_c4_agg(const uint32_t *__restrict__ step_ptr, const uint4 *__restrict__ steps,
                                                 const uint32_t *__restrict__ entries, const f32x4 *__restrict__ table,
                                                 f32x4 *__restrict__ agg, uint32_t n, uint32_t Rc, uint32_t Bc, uint32_t nnz,
                                                 uint32_t nchunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4 *acc = reinterpret_cast<f32x4 *>(smem);                       // Rc x 4 floats
    f32x4 *vbuf = acc + Rc;                                             // kStep gathered rows
    uint32_t *ebuf = reinterpret_cast<uint32_t *>(vbuf + kStep);        // kStep entries
    const uint32_t tid = threadIdx.x;
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint32_t row0 = chunk * Rc;
        __syncthreads();
        for (uint32_t i = tid; i < Rc; i += 1024) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const uint32_t st0 = step_ptr[chunk], st1 = step_ptr[chunk + 1];
        const int nsteps = (int)(st1 - st0);
        // time u: process step u (values and entries in LDS), park step u + 1 (gathered during u), gather step u + 2
        // (entries in registers since u - 2), load the entries of step u + 4
        uint32_t ea0 = 0, ea1 = 0, eb0 = 0, eb1 = 0, ec0 = 0, ec1 = 0, ed0 = 0, ed1 = 0;   // entry ring (4 steps)
        f32x4 v0, v1;   // the values gathered for the step that is parked next (one step of latency hiding: L2 hits)
        uint32_t la = 0, lb = 0, lc = 0, ld = 0, lcur = 0, cba = 0, cbb = 0, cbc = 0, cbd = 0;
        uint4 dn = steps[st0];
        v0 = v1 = f32x4{0, 0, 0, 0};
#define C4_STEP(u_, e0_, e1_, l_, cb_, ge0_, ge1_, gcb_)                                   \
        {                                                                                             \
            if ((u_) >= 0) {                                                                          \
                __syncthreads();                                   /* step u parked */                \
                const uint32_t i0_ = tid, i1_ = tid + 1024;                                           \
                const bool in0_ = i0_ < lcur, in1_ = i1_ < lcur;                                      \
                const uint32_t a0_ = ebuf[i0_], a1_ = ebuf[i1_];                                      \
                const uint32_t b0_ = ebuf[(int)i0_ - 1], b1_ = ebuf[i1_ - 1];                         \
                const uint32_t r0_ = a0_ >> 17, r1_ = a1_ >> 17;                                      \
                const bool h0_ = in0_ && (i0_ == 0 || (b0_ >> 17) != r0_);                            \
                const bool h1_ = in1_ && (b1_ >> 17) != r1_;                                          \
                if (h0_) {                                                                            \
                    f32x4 s_ = acc[r0_] + vbuf[i0_];                                                  \
                    for (uint32_t k_ = i0_ + 1; k_ < lcur && (ebuf[k_] >> 17) == r0_; ++k_) s_ += vbuf[k_]; \
                    acc[r0_] = s_;                                                                    \
                }                                                                                     \
                if (h1_) {                                                                            \
                    f32x4 s_ = acc[r1_] + vbuf[i1_];                                                  \
                    for (uint32_t k_ = i1_ + 1; k_ < lcur && (ebuf[k_] >> 17) == r1_; ++k_) s_ += vbuf[k_]; \
                    acc[r1_] = s_;                                                                    \
                }                                                                                     \
                __syncthreads();                                   /* everyone done reading step u */ \
            }                                                                                         \
            /* park step u + 1: its entries (ring slot e_) and the values gathered for it (v_) */      \
            ebuf[tid] = e0_;                                                                          \
            ebuf[tid + 1024] = e1_;                                                                   \
            vbuf[tid] = v0;                                                                           \
            vbuf[tid + 1024] = v1;                                                                   \
            lcur = l_;                                                                                \
            /* gather step u + 2 (entries ge_, column base gcb_) into the value slot just freed */     \
            {                                                                                         \
                const uint32_t c0_ = gcb_ + (ge0_ & 0x1FFFF), c1_ = gcb_ + (ge1_ & 0x1FFFF);          \
                v0 = table[c0_ < n ? c0_ : n];                                                        \
                v1 = table[c1_ < n ? c1_ : n];                                                       \
            }                                                                                         \
            /* load the entries of step u + 5 into the slot just parked */                            \
            {                                                                                         \
                const uint4 dl_ = dn;                                                                 \
                const int nx_ = (u_) + 6;                                                             \
                dn = steps[st0 + (uint32_t)(nx_ > 0 ? nx_ : 0)];                                      \
                const uint32_t x0_ = dl_.y + tid, x1_ = dl_.y + tid + 1024;                           \
                e0_ = entries[x0_ < nnz ? x0_ : nnz];                                                 \
                e1_ = entries[x1_ < nnz ? x1_ : nnz];                                                 \
                l_ = ((u_) + 5 >= 0) ? dl_.z : 0u;                                                    \
                cb_ = dl_.x * Bc;                                                                     \
            }                                                                                         \
        }
        // entry ring slot of step s: s & 3 (a, b, c, d)
        for (int u = -8; u < nsteps; u += 4) {
            C4_STEP(u,     eb0, eb1, lb, cbb, ec0, ec1, cbc)
            C4_STEP(u + 1, ec0, ec1, lc, cbc, ed0, ed1, cbd)
            C4_STEP(u + 2, ed0, ed1, ld, cbd, ea0, ea1, cba)
            C4_STEP(u + 3, ea0, ea1, la, cba, eb0, eb1, cbb)
        }
#undef C4_STEP
        __syncthreads();
        for (uint32_t i = tid; i < Rc && row0 + i < n; i += 1024) agg[row0 + i] = acc[i];
    }
}

extern "C" int c4_agg(const uint32_t *step_ptr, const void *steps, const uint32_t *entries, const void *table, void *agg,
                      uint32_t n, uint32_t Rc, uint32_t Bc, uint32_t nchunks, uint32_t nnz, uint32_t grid, void *stream) {
    const size_t lds = (size_t)Rc * 16 + kStep * 16 + kStep * 4 + 64;
    if (lds > 163840) return -3;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_c4_agg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_c4_agg, dim3(grid), dim3(1024), lds, (hipStream_t)stream, step_ptr, reinterpret_cast<const uint4 *>(steps),
                       entries, reinterpret_cast<const f32x4 *>(table), reinterpret_cast<f32x4 *>(agg), n, Rc, Bc, nnz - 1, nchunks);
    CK(hipGetLastError());
    return 0;
}
