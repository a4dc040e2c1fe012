// gather_policy.hip — microbenchmark: what does a random 64-byte row gather cost on
// MI355X under each cache policy of the load instruction?
//   variant 0: global_load_dwordx4 (default policy)
//   variant 1: ... nt
//   variant 2: ... sc1
//   variant 3: ... sc0 sc1
//   variant 4: ... sc0
// Each quad of lanes reads one 64-byte row (16 B per lane) per step, rows chosen by
// a precomputed random index stream; sums are written out so nothing is dead.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_policy gather_policy.hip
// Run:   ./gather_policy [rows_millions=10] [gathers_millions=200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int V>
__device__ __forceinline__ float4 load_row(const float4 *p) {
    float4 r;
    if constexpr (V == 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    if constexpr (V == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    if constexpr (V == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    if constexpr (V == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    if constexpr (V == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
}

// issue U loads back to back, then one wait (memory-level parallelism like the real kernel)
template <int V, int U>
__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ tab, const uint32_t *__restrict__ idx,
                                                 float4 *__restrict__ out, size_t per_quad) {
    const size_t quad = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int c = threadIdx.x & 3;
    const uint32_t *my = idx + quad * per_quad;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = 0; i < per_quad; i += U) {
        float4 r[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const float4 *p = tab + (size_t)my[i + k] * 4 + c;
            if constexpr (V == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[k]) : "v"(p) : "memory");
            if constexpr (V == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r[k]) : "v"(p) : "memory");
            if constexpr (V == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r[k]) : "v"(p) : "memory");
            if constexpr (V == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(r[k]) : "v"(p) : "memory");
            if constexpr (V == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(r[k]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < U; ++k) { acc.x += r[k].x; acc.y += r[k].y; acc.z += r[k].z; acc.w += r[k].w; }
    }
    out[quad * 4 + c] = acc;
}

__global__ void k_fill_idx(uint32_t *idx, size_t n, uint32_t rows) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        idx[i] = (uint32_t)(z % rows);
    }
}
__global__ void k_fill_tab(float *t, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        t[i] = (float)(i % 1000) * 1e-3f;
}

template <int V, int U>
float run(const float4 *tab, const uint32_t *idx, float4 *out, size_t quads, size_t per_quad, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = (int)(quads * 4 / 256);
    hipLaunchKernelGGL((k_gather<V, U>), dim3(blocks), dim3(256), 0, 0, tab, idx, out, per_quad);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_gather<V, U>), dim3(blocks), dim3(256), 0, 0, tab, idx, out, per_quad);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv) {
    const size_t rows = (size_t)(argc > 1 ? atof(argv[1]) : 10) * 1000000;
    const size_t gathers_req = (size_t)(argc > 2 ? atof(argv[2]) : 200) * 1000000;
    // waves per CU in flight (argv[3], default 16): fewer blocks -> fewer resident waves per CU,
    // to see whether the gather rate is limited by requests in flight or by the fabric
    const size_t waves_per_cu = argc > 3 ? (size_t)atoi(argv[3]) : 16;
    const size_t quads = 256 * waves_per_cu * 16 * (argc > 4 ? (size_t)atoi(argv[4]) : 4);   // x oversubscription
    const size_t per_quad = (gathers_req / quads) / 8 * 8;
    const size_t gathers = quads * per_quad;
    float4 *tab, *out; uint32_t *idx;
    // argv[5]: how the table is allocated: 0 = hipMalloc (default), 1 = fine-grained, 3 = uncached (MTYPE UC)
    const unsigned tab_flags = argc > 5 ? (unsigned)atoi(argv[5]) : 0;
    if (tab_flags) {
        CK(hipExtMallocWithFlags((void **)&tab, (rows + 1) * 64, tab_flags));
        printf("table allocated with hipExtMallocWithFlags(%u)\n", tab_flags);
    } else {
        CK(hipMalloc(&tab, (rows + 1) * 64));
    }
    CK(hipMalloc(&idx, gathers * 4)); CK(hipMalloc(&out, quads * 64));
    hipLaunchKernelGGL(k_fill_tab, dim3(4096), dim3(256), 0, 0, (float *)tab, (rows + 1) * 16);
    hipLaunchKernelGGL(k_fill_idx, dim3(4096), dim3(256), 0, 0, idx, gathers, (uint32_t)rows);
    CK(hipDeviceSynchronize());
    printf("table %.0f MB, %zu gathers of 64 B (%.2f GB useful), %zu waves/CU x %zu\n", rows * 64 / 1e6, gathers,
           gathers * 64 / 1e9, waves_per_cu, quads / (256 * waves_per_cu * 16));
    const char *names[] = {"default", "nt", "sc1", "sc0 sc1", "sc0"};
    float ms[5][2];
    ms[0][0] = run<0, 4>(tab, idx, out, quads, per_quad, 3); ms[0][1] = run<0, 8>(tab, idx, out, quads, per_quad, 3);
    ms[1][0] = run<1, 4>(tab, idx, out, quads, per_quad, 3); ms[1][1] = run<1, 8>(tab, idx, out, quads, per_quad, 3);
    ms[2][0] = run<2, 4>(tab, idx, out, quads, per_quad, 3); ms[2][1] = run<2, 8>(tab, idx, out, quads, per_quad, 3);
    ms[3][0] = run<3, 4>(tab, idx, out, quads, per_quad, 3); ms[3][1] = run<3, 8>(tab, idx, out, quads, per_quad, 3);
    ms[4][0] = run<4, 4>(tab, idx, out, quads, per_quad, 3); ms[4][1] = run<4, 8>(tab, idx, out, quads, per_quad, 3);
    for (int v = 0; v < 5; ++v)
        printf("%-8s U=4: %7.3f ms  %6.0f GB/s useful   U=8: %7.3f ms  %6.0f GB/s useful   (%.1f G rows/s)\n", names[v],
               ms[v][0], gathers * 64 / ms[v][0] / 1e6, ms[v][1], gathers * 64 / ms[v][1] / 1e6, gathers / ms[v][1] / 1e6);
    return 0;
}
