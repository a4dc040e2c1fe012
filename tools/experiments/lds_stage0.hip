// Prototype (experiment, not product): F = 1 aggregation with the table of neighbour values held in LDS.
//
// Rows are cut into chunks of Rc rows (one workgroup each, accumulators in LDS), columns into blocks
// of Bc vertices.  The CSR entries are re-laid out as segments (chunk, block), each sorted by
// (row, col) and packed as (row_local << 17 | col_local).  A workgroup walks its chunk's segments in
// block order; per block it copies the block's slice of the BYTE weight table (W(v) <= 255) into LDS
// and every thread adds lut[w] = (float)w / ws for its entries into acc[row_local] — rows keep their
// CSR order because blocks ascend and a row's entries inside a segment are handled by one thread.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return -1; } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(1024) void k_lds_agg(const uint32_t *__restrict__ seg_ptr, const uint32_t *__restrict__ entries,
                                                  const uint8_t *__restrict__ wbyte, float ws, float *__restrict__ agg,
                                                  uint32_t n, uint32_t Rc, uint32_t Bc, uint32_t nblocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *acc = reinterpret_cast<float *>(smem);                       // Rc floats
    float *lut = acc + Rc;                                              // 256 floats
    uint8_t *slice = reinterpret_cast<uint8_t *>(lut + 256);            // Bc bytes (multiple of 16)
    const uint32_t chunk = blockIdx.x, tid = threadIdx.x;
    const uint32_t row0 = chunk * Rc;
    for (uint32_t i = tid; i < Rc; i += 1024) acc[i] = 0.0f;
    if (tid < 256) lut[tid] = (float)tid / ws;
    const uint32_t *sp = seg_ptr + (size_t)chunk * nblocks;
    for (uint32_t b = 0; b < nblocks; ++b) {
        const uint32_t s0 = sp[b], s1 = sp[b + 1];
        __syncthreads();                                                // previous block's readers are done
        const uint32_t col0 = b * Bc;
        const uint32_t words = ((min(Bc, n - col0) + 15) / 16);        // 16-byte pieces
        for (uint32_t i = tid; i < words; i += 1024)
            reinterpret_cast<uint4 *>(slice)[i] = reinterpret_cast<const uint4 *>(wbyte + col0)[i];
        __syncthreads();
        for (uint32_t e = s0 + tid; e < s1; e += 1024) {
            const uint32_t p = entries[e];
            const uint32_t r = p >> 17;
            if (e > s0 && (entries[e - 1] >> 17) == r) continue;        // not the head of its row's run
            float a = acc[r];
            a += lut[slice[p & 0x1FFFF]];
            for (uint32_t f = e + 1; f < s1; ++f) {                     // the rest of the run, in order
                const uint32_t pn = entries[f];
                if ((pn >> 17) != r) break;
                a += lut[slice[pn & 0x1FFFF]];
            }
            acc[r] = a;
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < Rc && row0 + i < n; i += 1024) agg[row0 + i] = acc[i];
}

// v1: seg_ptr of the chunk staged in LDS, two slice buffers, the next block's slice and the entries of
// the block after next already in registers when a block is processed (global latency off the loop).
__global__ __launch_bounds__(1024) void k_lds_agg_v1(const uint32_t *__restrict__ seg_ptr, const uint32_t *__restrict__ entries,
                                                     const uint8_t *__restrict__ wbyte, float ws, float *__restrict__ agg,
                                                     uint32_t n, uint32_t Rc, uint32_t Bc, uint32_t nblocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *acc = reinterpret_cast<float *>(smem);                       // Rc floats
    float *lut = acc + Rc;                                              // 256 floats
    uint32_t *sp = reinterpret_cast<uint32_t *>(lut + 256);             // nblocks + 1 (padded to 1024)
    uint8_t *slice0 = reinterpret_cast<uint8_t *>(sp + 1024);           // 2 x Bc bytes
    const uint32_t chunk = blockIdx.x, tid = threadIdx.x;
    const uint32_t row0 = chunk * Rc;
    for (uint32_t i = tid; i < Rc; i += 1024) acc[i] = 0.0f;
    if (tid < 256) lut[tid] = (float)tid / ws;
    if (tid <= nblocks) sp[tid] = seg_ptr[(size_t)chunk * nblocks + tid];
    constexpr int SW = 2;   // uint4 per thread per slice: Bc <= 32768
    constexpr int J = 2;    // entries per thread per block held in registers: segments up to 2048 (longer: loop)
    const uint32_t pieces = (Bc + 15) / 16;
    auto load_slice = [&](uint32_t b, uint4 (&r)[SW]) {
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            const uint32_t i = tid + 1024 * k;
            const uint64_t byte = (uint64_t)b * Bc + (uint64_t)i * 16;
            r[k] = (b < nblocks && i < pieces && byte < (uint64_t)n + 48) ? reinterpret_cast<const uint4 *>(wbyte)[byte / 16]
                                                                         : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_slice = [&](uint32_t b, const uint4 (&r)[SW]) {
        uint4 *dst = reinterpret_cast<uint4 *>(slice0 + (size_t)(b & 1) * Bc);
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            const uint32_t i = tid + 1024 * k;
            if (i < pieces) dst[i] = r[k];
        }
    };
    __syncthreads();   // sp visible
    auto load_entries = [&](uint32_t b, uint32_t (&p)[J], uint32_t (&q)[J]) {
        const uint32_t s0 = b < nblocks ? sp[b] : 0, s1 = b < nblocks ? sp[b + 1] : 0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const uint32_t e = s0 + tid + 1024 * j;
            p[j] = e < s1 ? entries[e] : 0xFFFFFFFFu;
            q[j] = (e < s1 && e > s0) ? entries[e - 1] : 0xFFFFFFFFu;   // the entry before it (run-head test)
        }
    };
    uint4 sreg[SW];
    uint32_t pa[J], qa[J], pb[J], qb[J];
    load_slice(0, sreg);
    store_slice(0, sreg);
    load_slice(1, sreg);
    load_entries(0, pa, qa);
    load_entries(1, pb, qb);
    for (uint32_t b = 0; b < nblocks; ++b) {
        __syncthreads();                                                // slice b complete; everyone done with b - 1
        const uint8_t *slice = slice0 + (size_t)(b & 1) * Bc;
        const uint32_t s0 = sp[b], s1 = sp[b + 1];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const uint32_t p = pa[j];
            if (p == 0xFFFFFFFFu) continue;
            const uint32_t r = p >> 17;
            if (qa[j] != 0xFFFFFFFFu && (qa[j] >> 17) == r) continue;   // not the head of its row's run
            float a = acc[r];
            a += lut[slice[p & 0x1FFFF]];
            for (uint32_t f = s0 + tid + 1024 * j + 1; f < s1; ++f) {   // the rest of the run, in order
                const uint32_t pn = entries[f];
                if ((pn >> 17) != r) break;
                a += lut[slice[pn & 0x1FFFF]];
            }
            acc[r] = a;
        }
        for (uint32_t e = s0 + tid + 1024 * J; e < s1; e += 1024) {    // segments longer than the register window
            const uint32_t p = entries[e];
            const uint32_t r = p >> 17;
            if ((entries[e - 1] >> 17) == r) continue;
            float a = acc[r];
            a += lut[slice[p & 0x1FFFF]];
            for (uint32_t f = e + 1; f < s1; ++f) {
                const uint32_t pn = entries[f];
                if ((pn >> 17) != r) break;
                a += lut[slice[pn & 0x1FFFF]];
            }
            acc[r] = a;
        }
        store_slice(b + 1, sreg);          // buffer (b + 1) & 1 was last read in iteration b - 1
        load_slice(b + 2, sreg);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            pa[j] = pb[j];
            qa[j] = qb[j];
        }
        load_entries(b + 2, pb, qb);
    }
    __syncthreads();
    for (uint32_t i = tid; i < Rc && row0 + i < n; i += 1024) agg[row0 + i] = acc[i];
}

// v2: like v1 with rings of four in registers (slices b+2..b+5 and entries b+1..b+4 in flight): a CU needs
// ~100 KB in flight to stream its 20 MB of slices at L2 latency.
__global__ __launch_bounds__(1024) void k_lds_agg_v2(const uint32_t *__restrict__ seg_ptr, const uint32_t *__restrict__ entries,
                                                     const uint8_t *__restrict__ wbyte, float ws, float *__restrict__ agg,
                                                     uint32_t n, uint32_t Rc, uint32_t Bc, uint32_t nblocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *acc = reinterpret_cast<float *>(smem);                       // Rc floats
    float *lut = acc + Rc;                                              // 256 floats
    uint32_t *sp = reinterpret_cast<uint32_t *>(lut + 256);             // nblocks + 1 (padded to 1024)
    uint8_t *slice0 = reinterpret_cast<uint8_t *>(sp + 1024);           // 2 x Bc bytes
    const uint32_t chunk = blockIdx.x, tid = threadIdx.x;
    const uint32_t row0 = chunk * Rc;
    for (uint32_t i = tid; i < Rc; i += 1024) acc[i] = 0.0f;
    if (tid < 256) lut[tid] = (float)tid / ws;
    if (tid <= nblocks) sp[tid] = seg_ptr[(size_t)chunk * nblocks + tid];
    constexpr int SW = 2;   // uint4 per thread per slice: Bc <= 32768
    constexpr int J = 2;    // entries per thread per block held in registers: segments up to 2048 (longer: loop)
    const uint32_t pieces = (Bc + 15) / 16;
    auto load_slice = [&](uint32_t b, uint4 (&r)[SW]) {
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            const uint32_t i = tid + 1024 * k;
            const uint64_t byte = (uint64_t)b * Bc + (uint64_t)i * 16;
            r[k] = (b < nblocks && i < pieces && byte < (uint64_t)n + 48) ? reinterpret_cast<const uint4 *>(wbyte)[byte / 16]
                                                                         : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_slice = [&](uint32_t b, const uint4 (&r)[SW]) {
        uint4 *dst = reinterpret_cast<uint4 *>(slice0 + (size_t)(b & 1) * Bc);
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            const uint32_t i = tid + 1024 * k;
            if (i < pieces) dst[i] = r[k];
        }
    };
    __syncthreads();   // sp visible
    auto load_entries = [&](uint32_t b, uint32_t (&p)[J], uint32_t (&q)[J]) {
        const uint32_t s0 = b < nblocks ? sp[b] : 0, s1 = b < nblocks ? sp[b + 1] : 0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const uint32_t e = s0 + tid + 1024 * j;
            p[j] = e < s1 ? entries[e] : 0xFFFFFFFFu;
            q[j] = (e < s1 && e > s0) ? entries[e - 1] : 0xFFFFFFFFu;   // the entry before it (run-head test)
        }
    };
    uint4 sr0[SW], sr1[SW], sr2[SW], sr3[SW];
    uint32_t p0[J], q0[J], p1[J], q1[J], p2[J], q2[J], p3[J], q3[J];
    load_slice(0, sr0);
    store_slice(0, sr0);
    load_slice(1, sr1);
    load_slice(2, sr2);
    load_slice(3, sr3);
    load_slice(4, sr0);
    load_entries(0, p0, q0);
    load_entries(1, p1, q1);
    load_entries(2, p2, q2);
    load_entries(3, p3, q3);
    auto process = [&](uint32_t b, const uint32_t (&pa)[J], const uint32_t (&qa)[J]) {
        const uint8_t *slice = slice0 + (size_t)(b & 1) * Bc;
        const uint32_t s0 = sp[b], s1 = sp[b + 1];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const uint32_t p = pa[j];
            if (p == 0xFFFFFFFFu) continue;
            const uint32_t r = p >> 17;
            if (qa[j] != 0xFFFFFFFFu && (qa[j] >> 17) == r) continue;   // not the head of its row's run
            float a = acc[r];
            a += lut[slice[p & 0x1FFFF]];
            for (uint32_t f = s0 + tid + 1024 * j + 1; f < s1; ++f) {   // the rest of the run, in order
                const uint32_t pn = entries[f];
                if ((pn >> 17) != r) break;
                a += lut[slice[pn & 0x1FFFF]];
            }
            acc[r] = a;
        }
        for (uint32_t e = s0 + tid + 1024 * J; e < s1; e += 1024) {    // segments longer than the register window
            const uint32_t p = entries[e];
            const uint32_t r = p >> 17;
            if ((entries[e - 1] >> 17) == r) continue;
            float a = acc[r];
            a += lut[slice[p & 0x1FFFF]];
            for (uint32_t f = e + 1; f < s1; ++f) {
                const uint32_t pn = entries[f];
                if ((pn >> 17) != r) break;
                a += lut[slice[pn & 0x1FFFF]];
            }
            acc[r] = a;
        }
    };
    // step(b, E, S): E = entries ring slot of block b, S = slice ring slot holding block b + 1
#define LDS_STEP(b_, pe_, qe_, sr_)                          \
    if ((b_) < nblocks) {                                    \
        __syncthreads();                                     \
        process((b_), pe_, qe_);                             \
        store_slice((b_) + 1, sr_);                          \
        load_slice((b_) + 5, sr_);                           \
        load_entries((b_) + 4, pe_, qe_);                    \
    }
    for (uint32_t b = 0; b < nblocks; b += 4) {
        LDS_STEP(b, p0, q0, sr1)
        LDS_STEP(b + 1, p1, q1, sr2)
        LDS_STEP(b + 2, p2, q2, sr3)
        LDS_STEP(b + 3, p3, q3, sr0)
    }
#undef LDS_STEP
    __syncthreads();
    for (uint32_t i = tid; i < Rc && row0 + i < n; i += 1024) agg[row0 + i] = acc[i];
}

extern "C" int lds_agg_v2(const uint32_t *seg_ptr, const uint32_t *entries, const uint8_t *wbyte, float ws, float *agg,
                          uint32_t n, uint32_t Rc, uint32_t Bc, uint32_t nchunks, uint32_t nblocks, void *stream) {
    if (Bc > 32768 || nblocks + 1 > 1024) return -2;
    const size_t lds = (size_t)Rc * 4 + 1024 + 4096 + 2 * (size_t)(((Bc + 15) / 16) * 16);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds_agg_v2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_lds_agg_v2, dim3(nchunks), dim3(1024), lds, (hipStream_t)stream, seg_ptr, entries, wbyte, ws, agg, n, Rc,
                       Bc, nblocks);
    CK(hipGetLastError());
    return 0;
}

extern "C" int lds_agg_v1(const uint32_t *seg_ptr, const uint32_t *entries, const uint8_t *wbyte, float ws, float *agg,
                          uint32_t n, uint32_t Rc, uint32_t Bc, uint32_t nchunks, uint32_t nblocks, void *stream) {
    if (Bc > 32768 || nblocks + 1 > 1024) return -2;
    const size_t lds = (size_t)Rc * 4 + 1024 + 4096 + 2 * (size_t)(((Bc + 15) / 16) * 16);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds_agg_v1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_lds_agg_v1, dim3(nchunks), dim3(1024), lds, (hipStream_t)stream, seg_ptr, entries, wbyte, ws, agg, n, Rc,
                       Bc, nblocks);
    CK(hipGetLastError());
    return 0;
}

// v3: every prefetch unconditional (clamped addresses, validity by arithmetic) so the compiler can count
// outstanding loads and wait for exactly the one it needs; the entry after each entry is prefetched too,
// so a thread only touches memory inside the loop when its row really has a second entry in the block.
__global__ __launch_bounds__(1024) void k_lds_agg_v3(const uint32_t *__restrict__ seg_ptr, const uint32_t *__restrict__ entries,
                                                     const uint8_t *__restrict__ wbyte, float ws, float *__restrict__ agg,
                                                     uint32_t n, uint32_t Rc, uint32_t Bc, uint32_t nblocks, uint32_t nnz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *acc = reinterpret_cast<float *>(smem);                       // Rc floats
    float *lut = acc + Rc;                                              // 256 floats
    uint32_t *sp = reinterpret_cast<uint32_t *>(lut + 256);             // nblocks + 1 (room for 1024)
    uint8_t *slice0 = reinterpret_cast<uint8_t *>(sp + 1024);           // 2 x 32 KiB
    const uint32_t chunk = blockIdx.x, tid = threadIdx.x;
    const uint32_t row0 = chunk * Rc;
    for (uint32_t i = tid; i < Rc; i += 1024) acc[i] = 0.0f;
    if (tid < 256) lut[tid] = (float)tid / ws;
    if (tid <= nblocks) sp[tid] = seg_ptr[(size_t)chunk * nblocks + tid];
    constexpr int SW = 2;   // uint4 per thread per slice (slices are 32 KiB in LDS whatever Bc <= 32768 is)
    constexpr int J = 2;    // entries per thread per block held in registers: segments up to 2048 (longer: loop)
    const uint64_t last_piece = ((uint64_t)n + 15) / 16;                // table is padded beyond this
    auto load_slice = [&](uint32_t b, uint4 (&r)[SW]) {
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            uint64_t piece = ((uint64_t)b * Bc) / 16 + tid + 1024 * k;
            piece = piece < last_piece ? piece : last_piece;
            r[k] = reinterpret_cast<const uint4 *>(wbyte)[piece];
        }
    };
    auto store_slice = [&](uint32_t b, const uint4 (&r)[SW]) {
        uint4 *dst = reinterpret_cast<uint4 *>(slice0 + (size_t)(b & 1) * 32768);
#pragma unroll
        for (int k = 0; k < SW; ++k) dst[tid + 1024 * k] = r[k];
    };
    __syncthreads();   // sp visible
#ifdef LDS_FAST_ONLY   /* timing experiment: no memory access inside the block loop (wrong sums for long runs) */
#define LDS_RARE(...)
#else
#define LDS_RARE(...) __VA_ARGS__
#endif
    // the entry before, the entry, the entry after (0xFFFFFFFF = none), J per block, four blocks in flight
#define LDS_LOAD_ENTRIES(b_, q_, p_, x_)                                                  \
    {                                                                                     \
        const uint32_t bb_ = (b_) < nblocks ? (b_) : nblocks;                             \
        const uint32_t s0_ = sp[bb_], s1_ = sp[bb_ < nblocks ? bb_ + 1 : nblocks];        \
        _Pragma("unroll") for (int j = 0; j < J; ++j) {                                   \
            const uint32_t e_ = s0_ + tid + 1024 * j;                                     \
            const uint32_t ec_ = e_ < nnz ? e_ : nnz;                                     \
            q_[j] = entries[ec_];     /* raw: validity is worked out when the block is processed, */ \
            p_[j] = entries[ec_ + 1]; /* so nothing here waits for the loads just issued          */ \
            x_[j] = entries[ec_ + 2];                                                     \
        }                                                                                 \
    }
#define LDS_LOAD_SLICE(b_, r_)                                                            \
    _Pragma("unroll") for (int k = 0; k < SW; ++k) {                                      \
        uint64_t piece_ = ((uint64_t)(b_) * Bc) / 16 + tid + 1024 * k;                    \
        piece_ = piece_ < last_piece ? piece_ : last_piece;                               \
        r_[k] = reinterpret_cast<const u32x4 *>(wbyte)[piece_];                           \
    }
#define LDS_STORE_SLICE(b_, r_)                                                           \
    _Pragma("unroll") for (int k = 0; k < SW; ++k)                                        \
        reinterpret_cast<u32x4 *>(slice0 + (size_t)((b_) & 1) * 32768)[tid + 1024 * k] = r_[k];
#define LDS_ADD_RUN(slice_, first_, r_, f0_, s1_)                                         \
    {                                                                                     \
        float a_ = acc[r_];                                                               \
        a_ += lut[slice_[(first_) & 0x1FFFF]];                                            \
        for (uint32_t f_ = (f0_); f_ < (s1_); ++f_) {                                     \
            const uint32_t pn_ = entries[f_ + 1];                                         \
            if ((pn_ >> 17) != (r_)) break;                                               \
            a_ += lut[slice_[pn_ & 0x1FFFF]];                                             \
        }                                                                                 \
        acc[r_] = a_;                                                                     \
    }
#define LDS_PROCESS(b_, q_, p_, x_)                                                       \
    {                                                                                     \
        const uint8_t *slice_ = slice0 + (size_t)((b_) & 1) * 32768;                      \
        const uint32_t s0_ = sp[b_], s1_ = sp[(b_) + 1];                                  \
        _Pragma("unroll") for (int j = 0; j < J; ++j) {                                   \
            const uint32_t pp_ = p_[j];                                                   \
            const uint32_t e_ = s0_ + tid + 1024 * j;                                     \
            if (e_ < s1_) {                                                               \
                const uint32_t r_ = pp_ >> 17;                                            \
                if (!(e_ > s0_ && (q_[j] >> 17) == r_)) {                                 \
                    if (!(e_ + 1 < s1_) || (x_[j] >> 17) != r_) {                         \
                        acc[r_] += lut[slice_[pp_ & 0x1FFFF]];                            \
                    } else {                                                              \
                        LDS_RARE(LDS_ADD_RUN(slice_, pp_, r_, s0_ + tid + 1024 * j + 1, s1_)) \
                    }                                                                     \
                }                                                                         \
            }                                                                             \
        }                                                                                 \
        LDS_RARE(for (uint32_t e_ = s0_ + tid + 1024 * J; e_ < s1_; e_ += 1024) {         \
            const uint32_t pp_ = entries[e_ + 1];                                         \
            const uint32_t r_ = pp_ >> 17;                                                \
            if (e_ > s0_ && (entries[e_] >> 17) == r_) continue;                          \
            LDS_ADD_RUN(slice_, pp_, r_, e_ + 1, s1_)                                     \
        })                                                                                \
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 sr0[SW], sr1[SW], sr2[SW], sr3[SW];
    uint32_t q0[J], p0[J], x0[J], q1[J], p1[J], x1[J], q2[J], p2[J], x2[J], q3[J], p3[J], x3[J];
    LDS_LOAD_SLICE(0, sr0)
    LDS_STORE_SLICE(0, sr0)
    LDS_LOAD_SLICE(1, sr1)
    LDS_LOAD_SLICE(2, sr2)
    LDS_LOAD_SLICE(3, sr3)
    LDS_LOAD_SLICE(4, sr0)
    LDS_LOAD_ENTRIES(0, q0, p0, x0)
    LDS_LOAD_ENTRIES(1, q1, p1, x1)
    LDS_LOAD_ENTRIES(2, q2, p2, x2)
    LDS_LOAD_ENTRIES(3, q3, p3, x3)
#define LDS_STEP(b_, q_, p_, x_, sr_)                        \
    if ((b_) < nblocks) {                                    \
        __syncthreads();                                     \
        LDS_PROCESS((b_), q_, p_, x_)                        \
        LDS_STORE_SLICE((b_) + 1, sr_)                       \
        LDS_LOAD_SLICE((b_) + 5, sr_)                        \
        LDS_LOAD_ENTRIES((b_) + 4, q_, p_, x_)               \
    }
    for (uint32_t b = 0; b < nblocks; b += 4) {
        LDS_STEP(b, q0, p0, x0, sr1)
        LDS_STEP(b + 1, q1, p1, x1, sr2)
        LDS_STEP(b + 2, q2, p2, x2, sr3)
        LDS_STEP(b + 3, q3, p3, x3, sr0)
    }
#undef LDS_STEP
    __syncthreads();
    for (uint32_t i = tid; i < Rc && row0 + i < n; i += 1024) agg[row0 + i] = acc[i];
}

// entries: slot 0 is a pad, entry e lives in slot e + 1, one more pad slot at the end (nnz + 2 slots)
extern "C" int lds_agg_v3(const uint32_t *seg_ptr, const uint32_t *entries, const uint8_t *wbyte, float ws, float *agg,
                          uint32_t n, uint32_t Rc, uint32_t Bc, uint32_t nchunks, uint32_t nblocks, void *stream, uint32_t nnz) {
    if (Bc > 32768 || Bc % 16 || nblocks + 1 > 1024) return -2;
    const size_t lds = (size_t)Rc * 4 + 1024 + 4096 + 2 * 32768;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds_agg_v3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_lds_agg_v3, dim3(nchunks), dim3(1024), lds, (hipStream_t)stream, seg_ptr, entries, wbyte, ws, agg, n, Rc,
                       Bc, nblocks, nnz);
    CK(hipGetLastError());
    return 0;
}

// v4: the production shape.  A chunk walks a list of STEPS (block, first entry, count <= 2048); each step's
// entries are staged in LDS next to the block's slice, so the rare longer runs of a row are followed through
// LDS and the block loop contains no global load that depends on data — every prefetch is unconditional and
// four steps ahead.  Steps are padded (count 0) to a multiple of 4 per chunk.
__global__ __launch_bounds__(1024) void k_lds_agg_v4(const uint32_t *__restrict__ step_ptr /*chunks+1*/,
                                                     const uint4 *__restrict__ steps /*{block, first, count, 0}*/,
                                                     const uint32_t *__restrict__ entries, const uint8_t *__restrict__ wbyte,
                                                     float ws, float *__restrict__ agg, uint32_t n, uint32_t Rc, uint32_t Bc,
                                                     uint32_t nnz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *acc = reinterpret_cast<float *>(smem);                       // Rc floats
    float *lut = acc + Rc;                                              // 256 floats
    uint32_t *ebuf0 = reinterpret_cast<uint32_t *>(lut + 256);          // 2 x 2048 entries
    uint8_t *slice0 = reinterpret_cast<uint8_t *>(ebuf0 + 4096);        // 2 x 32 KiB
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t chunk = blockIdx.x, tid = threadIdx.x;
    const uint32_t row0 = chunk * Rc;
    for (uint32_t i = tid; i < Rc; i += 1024) acc[i] = 0.0f;
    if (tid < 256) lut[tid] = (float)tid / ws;
    const uint32_t st0 = step_ptr[chunk], st1 = step_ptr[chunk + 1];   // multiple of 4 steps, + 8 readable beyond
    const uint64_t last_piece = ((uint64_t)n + 15) / 16;
    constexpr int SW = 2, J = 2;
    // step descriptors travel one step ahead of the loads that need them (dn = descriptor of the next
    // step to LOAD), and each step's length stays in a register ring until the step is processed
#define V4_LOAD(d_, sr_, en_)                                                             \
    {                                                                                     \
        _Pragma("unroll") for (int k = 0; k < SW; ++k) {                                  \
            uint64_t piece_ = ((uint64_t)(d_).x * Bc) / 16 + tid + 1024 * k;              \
            piece_ = piece_ < last_piece ? piece_ : last_piece;                           \
            sr_[k] = reinterpret_cast<const u32x4 *>(wbyte)[piece_];                      \
        }                                                                                 \
        _Pragma("unroll") for (int j = 0; j < J; ++j) {                                   \
            const uint32_t e_ = (d_).y + tid + 1024 * j;                                  \
            en_[j] = entries[e_ < nnz ? e_ : nnz];                                        \
        }                                                                                 \
    }
#define V4_STORE(s_, sr_, en_)                                                            \
    {                                                                                     \
        _Pragma("unroll") for (int k = 0; k < SW; ++k)                                    \
            reinterpret_cast<u32x4 *>(slice0 + (size_t)((s_) & 1) * 32768)[tid + 1024 * k] = sr_[k]; \
        _Pragma("unroll") for (int j = 0; j < J; ++j) ebuf0[((s_) & 1) * 2048 + tid + 1024 * j] = en_[j]; \
    }
    // both entries of a thread advance together: the LDS reads of the two chains are issued back to back
#define V4_PROCESS(s_, len_)                                                              \
    {                                                                                     \
        const uint8_t *slice_ = slice0 + (size_t)((s_) & 1) * 32768;                      \
        const uint32_t *eb_ = ebuf0 + ((s_) & 1) * 2048;                                  \
        const uint32_t i0_ = tid, i1_ = tid + 1024;                                       \
        const bool in0_ = i0_ < (len_), in1_ = i1_ < (len_);                              \
        const uint32_t a0_ = eb_[in0_ ? i0_ : 0], a1_ = eb_[in1_ ? i1_ : 0];              \
        const uint32_t b0_ = eb_[(in0_ && i0_) ? i0_ - 1 : 0], b1_ = eb_[in1_ ? i1_ - 1 : 0]; \
        const uint32_t n0_ = eb_[(in0_ && i0_ + 1 < (len_)) ? i0_ + 1 : 0];               \
        const uint32_t n1_ = eb_[(in1_ && i1_ + 1 < (len_)) ? i1_ + 1 : 0];               \
        const uint32_t r0_ = a0_ >> 17, r1_ = a1_ >> 17;                                  \
        const bool h0_ = in0_ && (i0_ == 0 || (b0_ >> 17) != r0_);                        \
        const bool h1_ = in1_ && (b1_ >> 17) != r1_;                                      \
        const float v0_ = lut[slice_[a0_ & 0x1FFFF]], v1_ = lut[slice_[a1_ & 0x1FFFF]];   \
        const float c0_ = acc[r0_], c1_ = acc[r1_];                                       \
        const bool more0_ = h0_ && i0_ + 1 < (len_) && (n0_ >> 17) == r0_;                \
        const bool more1_ = h1_ && i1_ + 1 < (len_) && (n1_ >> 17) == r1_;                \
        float s0_ = c0_ + v0_, s1_ = c1_ + v1_;                                           \
        if (more0_)                                                                       \
            for (uint32_t k_ = i0_ + 1; k_ < (len_) && (eb_[k_] >> 17) == r0_; ++k_) s0_ += lut[slice_[eb_[k_] & 0x1FFFF]]; \
        if (more1_)                                                                       \
            for (uint32_t k_ = i1_ + 1; k_ < (len_) && (eb_[k_] >> 17) == r1_; ++k_) s1_ += lut[slice_[eb_[k_] & 0x1FFFF]]; \
        if (h0_) acc[r0_] = s0_;                                                          \
        if (h1_) acc[r1_] = s1_;                                                          \
    }
    static_assert(J == 2, "V4_PROCESS is written for two entries per thread");
    u32x4 sr0[SW], sr1[SW], sr2[SW], sr3[SW];
    uint32_t en0[J], en1[J], en2[J], en3[J];
    uint32_t l0, l1, l2, l3;                       // lengths of the steps whose data sit in ring slots 0..3
    uint4 dn;
    dn = steps[st0];     V4_LOAD(dn, sr0, en0) l0 = dn.z;
    V4_STORE(0, sr0, en0)
    dn = steps[st0 + 1]; V4_LOAD(dn, sr1, en1) l1 = dn.z;
    dn = steps[st0 + 2]; V4_LOAD(dn, sr2, en2) l2 = dn.z;
    dn = steps[st0 + 3]; V4_LOAD(dn, sr3, en3) l3 = dn.z;
    uint32_t lcur = l0;                             // length of the step about to be processed
    dn = steps[st0 + 4]; V4_LOAD(dn, sr0, en0) l0 = dn.z;
    dn = steps[st0 + 5];                            // descriptor for the next load, fetched a whole step early
    // V4_STEP(t, ring slot holding step t + 1, its length register)
#define V4_STEP(t_, sr_, en_, l_)                            \
    {                                                        \
        __syncthreads();                                     \
        V4_PROCESS((t_), lcur)                               \
        V4_STORE((t_) + 1, sr_, en_)                         \
        lcur = l_;                                           \
        const uint4 dl_ = dn;                                \
        dn = steps[st0 + (t_) + 6];                          \
        V4_LOAD(dl_, sr_, en_)                               \
        l_ = dl_.z;                                          \
    }
    for (uint32_t t = 0; t < st1 - st0; t += 4) {
        V4_STEP(t, sr1, en1, l1)
        V4_STEP(t + 1, sr2, en2, l2)
        V4_STEP(t + 2, sr3, en3, l3)
        V4_STEP(t + 3, sr0, en0, l0)
    }
#undef V4_STEP
    __syncthreads();
    for (uint32_t i = tid; i < Rc && row0 + i < n; i += 1024) agg[row0 + i] = acc[i];
}

extern "C" int lds_agg_v4(const uint32_t *step_ptr, const void *steps, const uint32_t *entries, const uint8_t *wbyte, float ws,
                          float *agg, uint32_t n, uint32_t Rc, uint32_t Bc, uint32_t nchunks, uint32_t nnz, void *stream) {
    if (Bc > 32768 || Bc % 16) return -2;
    const size_t lds = (size_t)Rc * 4 + 1024 + 16384 + 2 * 32768;
    if (lds > 163840) return -3;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds_agg_v4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_lds_agg_v4, dim3(nchunks), dim3(1024), lds, (hipStream_t)stream, step_ptr,
                       reinterpret_cast<const uint4 *>(steps), entries, wbyte, ws, agg, n, Rc, Bc, nnz);
    CK(hipGetLastError());
    return 0;
}

extern "C" int lds_agg(const uint32_t *seg_ptr, const uint32_t *entries, const uint8_t *wbyte, float ws, float *agg,
                       uint32_t n, uint32_t Rc, uint32_t Bc, uint32_t nchunks, uint32_t nblocks, void *stream) {
    const size_t lds = (size_t)Rc * 4 + 1024 + ((Bc + 15) / 16) * 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds_agg<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_lds_agg<1>, dim3(nchunks), dim3(1024), lds, (hipStream_t)stream, seg_ptr, entries, wbyte, ws, agg, n, Rc,
                       Bc, nblocks);
    CK(hipGetLastError());
    return 0;
}
