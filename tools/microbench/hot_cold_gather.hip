// hot_cold_gather.hip — microbenchmark for skewed graphs: a fraction `p` of the 64-byte row gathers goes to a small HOT
// set of rows (R-MAT-22: 54 % of all adjacency entries point to the 35 K vertices of degree >= 512, 2.2 MB of feature
// rows), the rest to uniformly random COLD rows of a table far larger than L2.  Does the hot set stay in the XCDs'
// L2s, and does a cache-policy bit on the COLD loads (tagged entries: the index's top bit says which class) protect it?
//   variant 0: every load default
//   variant 1: cold loads `nt`, hot loads default
//   variant 2: cold loads `sc1`, hot default
//   variant 3: cold loads `sc0 sc1`, hot default
//   variant 4: every load nt
// Build: hipcc --offload-arch=gfx950 -O3 -o hot_cold_gather hot_cold_gather.hip
// Run:   ./hot_cold_gather [rows_millions=4.2] [hot_rows=35000] [gathers_millions=128]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

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
            const uint32_t e = my[i + k];
            const bool hot = e >> 31;
            const float4 *p = tab + (size_t)(e & 0x7FFFFFFFu) * 4 + c;
            r[k] = make_float4(0, 0, 0, 0);
            if (V == 0 || (hot && V != 4)) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[k]) : "v"(p) : "memory");
            } else {
                if constexpr (V == 1 || V == 4) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r[k]) : "v"(p) : "memory");
                if constexpr (V == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r[k]) : "v"(p) : "memory");
                if constexpr (V == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(r[k]) : "v"(p) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < U; ++k) { acc.x += r[k].x; acc.y += r[k].y; acc.z += r[k].z; acc.w += r[k].w; }
    }
    out[quad * 4 + c] = acc;
}

__global__ void k_fill_idx(uint32_t *idx, size_t n, uint32_t rows, uint32_t hot_rows, uint32_t hot_permille) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const bool hot = (z >> 40) % 1000u < hot_permille;
        // hot rows are spread over the table (stride), like hub vertices among the others
        const uint32_t stride = rows / hot_rows;
        idx[i] = hot ? (0x80000000u | (uint32_t)((z % hot_rows) * stride)) : (uint32_t)(z % rows);
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
    const size_t rows = (size_t)((argc > 1 ? atof(argv[1]) : 4.2) * 1000000);
    const uint32_t hot_rows = argc > 2 ? (uint32_t)atoi(argv[2]) : 35000;
    const size_t gathers_req = (size_t)(argc > 3 ? atof(argv[3]) : 128) * 1000000;
    const size_t quads = 256 * 16 * 16 * 4;
    const size_t per_quad = (gathers_req / quads) / 8 * 8;
    const size_t gathers = quads * per_quad;
    float4 *tab, *out; uint32_t *idx;
    CK(hipMalloc(&tab, (rows + 1) * 64));
    CK(hipMalloc(&idx, gathers * 4)); CK(hipMalloc(&out, quads * 64));
    hipLaunchKernelGGL(k_fill_tab, dim3(4096), dim3(256), 0, 0, (float *)tab, (rows + 1) * 16);
    printf("table %.0f MB (%zu rows), hot set %u rows = %.1f MB, %zu gathers of 64 B\n", rows * 64 / 1e6, rows, hot_rows,
           hot_rows * 64 / 1e6, gathers);
    const char *names[] = {"all default", "cold nt", "cold sc1", "cold sc0sc1", "all nt"};
    for (uint32_t p : {0u, 400u, 540u, 800u, 1000u}) {
        hipLaunchKernelGGL(k_fill_idx, dim3(4096), dim3(256), 0, 0, idx, gathers, (uint32_t)rows, hot_rows, p);
        CK(hipDeviceSynchronize());
        float ms[5];
        ms[0] = run<0, 8>(tab, idx, out, quads, per_quad, 3);
        ms[1] = run<1, 8>(tab, idx, out, quads, per_quad, 3);
        ms[2] = run<2, 8>(tab, idx, out, quads, per_quad, 3);
        ms[3] = run<3, 8>(tab, idx, out, quads, per_quad, 3);
        ms[4] = run<4, 8>(tab, idx, out, quads, per_quad, 3);
        printf("hot fraction %.2f:", p / 1000.0);
        for (int v = 0; v < 5; ++v) printf("  %s %.3f ms (%.1f G rows/s)", names[v], ms[v], gathers / ms[v] / 1e6);
        printf("\n");
    }
    return 0;
}
