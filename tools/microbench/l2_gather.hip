// l2_gather.hip — microbenchmark: random 16-byte row reads from a table small enough to stay in L2
// (the access pattern of k_c4_agg's gather), per cache policy, rows in flight and resident waves.
// One lane = one row per load (64 different lines per wave instruction).
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_gather l2_gather.hip
// Run:   ./l2_gather [table_rows=131072] [gathers_millions=400] [waves_per_cu=16]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int V, int U, int BYTES>
__global__ __launch_bounds__(1024) void k_gather(const float4 *__restrict__ tab, const uint32_t *__restrict__ idx,
                                                  float4 *__restrict__ out, size_t per_thread, uint32_t stride) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)gridDim.x * blockDim.x;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = 0; i < per_thread; i += U) {
        float4 r[U];
        uint32_t ix[U];
#pragma unroll
        for (int k = 0; k < U; ++k) ix[k] = idx[(i + k) * total + t];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const float4 *p = tab + (size_t)ix[k] * stride;
            if constexpr (BYTES == 16) {
                if constexpr (V == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[k]) : "v"(p) : "memory");
                if constexpr (V == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r[k]) : "v"(p) : "memory");
                if constexpr (V == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r[k]) : "v"(p) : "memory");
                if constexpr (V == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(r[k]) : "v"(p) : "memory");
                if constexpr (V == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(r[k]) : "v"(p) : "memory");
            } else {
                r[k] = make_float4(0, 0, 0, 0);
                if constexpr (V == 0) asm volatile("global_load_dword %0, %1, off" : "=v"(r[k].x) : "v"(p) : "memory");
                if constexpr (V == 1) asm volatile("global_load_dword %0, %1, off nt" : "=v"(r[k].x) : "v"(p) : "memory");
                if constexpr (V == 2) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(r[k].x) : "v"(p) : "memory");
                if constexpr (V == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(r[k].x) : "v"(p) : "memory");
                if constexpr (V == 4) asm volatile("global_load_dword %0, %1, off sc0" : "=v"(r[k].x) : "v"(p) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < U; ++k) { acc.x += r[k].x; acc.y += r[k].y; acc.z += r[k].z; acc.w += r[k].w; }
    }
    out[t] = acc;
}

// mode 0: uniformly random rows; mode 1: the 64 lanes of a wave read 64 rows out of 512 consecutive ones
// (8 lines' worth per line of lanes: some lines shared); mode 2: 64 consecutive rows (8 lines per instruction)
__global__ void k_fill_idx(uint32_t *idx, size_t n, uint32_t rows, int mode, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t wave_instr = (i / total) * (total / 64) + (i % total) / 64;   // one id per wave instruction
        auto mix = [](uint64_t z) {
            z *= 0x9E3779B97F4A7C15ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        if (mode == 0) idx[i] = (uint32_t)(mix(i + 1) % rows);
        else if (mode == 1) idx[i] = (uint32_t)((mix(wave_instr + 1) % (rows - 512)) + mix(i + 7) % 512);
        else idx[i] = (uint32_t)((mix(wave_instr + 1) % (rows - 64)) + (i % 64));
    }
}
__global__ void k_fill_tab(float *t, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        t[i] = (float)(i % 1000) * 1e-3f;
}

template <int V, int U, int BYTES>
float run(const float4 *tab, const uint32_t *idx, float4 *out, int blocks, int threads, size_t per_thread, uint32_t stride) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_gather<V, U, BYTES>), dim3(blocks), dim3(threads), 0, 0, tab, idx, out, per_thread, stride);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_gather<V, U, BYTES>), dim3(blocks), dim3(threads), 0, 0, tab, idx, out, per_thread, stride);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 3;
}

int main(int argc, char **argv) {
    const uint32_t rows = argc > 1 ? (uint32_t)atoi(argv[1]) : 131072;
    const size_t gathers_req = (size_t)(argc > 2 ? atof(argv[2]) : 400) * 1000000;
    const int waves_per_cu = argc > 3 ? atoi(argv[3]) : 16;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    const uint32_t stride = argc > 5 ? (uint32_t)atoi(argv[5]) : 1;    // row pitch in 16-byte units
    const int threads = waves_per_cu >= 16 ? 1024 : waves_per_cu * 64, blocks = 256 * (waves_per_cu >= 16 ? waves_per_cu / 16 : 1);
    const size_t total = (size_t)threads * blocks;
    const size_t per_thread = (gathers_req / total) / 8 * 8;
    const size_t gathers = total * per_thread;
    float4 *tab, *out; uint32_t *idx;
    CK(hipMalloc(&tab, ((size_t)rows * stride + 1) * 16));
    CK(hipMalloc(&idx, gathers * 4)); CK(hipMalloc(&out, total * 16));
    hipLaunchKernelGGL(k_fill_tab, dim3(4096), dim3(256), 0, 0, (float *)tab, ((size_t)rows * stride + 1) * 4);
    hipLaunchKernelGGL(k_fill_idx, dim3(4096), dim3(256), 0, 0, idx, gathers, rows, mode, total);
    CK(hipDeviceSynchronize());
    printf("table %u rows x %u B = %.2f MB, %zu gathers, %d waves/CU (%d x %d), index mode %d\n", rows, 16 * stride,
           (double)rows * stride * 16 / 1e6, gathers, waves_per_cu, blocks, threads, mode);
    const char *names[] = {"default", "nt", "sc1", "sc0 sc1", "sc0"};
    float ms[5][3];
#define ROW(V) ms[V][0] = run<V, 4, 16>(tab, idx, out, blocks, threads, per_thread, stride); \
               ms[V][1] = run<V, 8, 16>(tab, idx, out, blocks, threads, per_thread, stride); \
               ms[V][2] = run<V, 8, 4>(tab, idx, out, blocks, threads, per_thread, stride);
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4)
    for (int v = 0; v < 5; ++v)
        printf("%-8s 16 B U=4: %7.3f ms %6.1f G rows/s   U=8: %7.3f ms %6.1f G rows/s   4 B U=8: %7.3f ms %6.1f G rows/s\n", names[v],
               ms[v][0], gathers / ms[v][0] / 1e6, ms[v][1], gathers / ms[v][1] / 1e6, ms[v][2], gathers / ms[v][2] / 1e6);
    return 0;
}
