// cold_row_size.hip — is the fabric's limit on random gathers a limit on REQUESTS or on BYTES?  Uniformly random rows of
// 16 / 32 / 64 bytes from tables of 67 MB (inside the Infinity Cache) and 269 MB / 1.07 GB (beyond it), every lane group of
// 1 / 2 / 4 lanes reading one row with 16-byte loads, 8 rows in flight per group.  If a 32-byte row (the 8 live columns of a
// skewed graph's stage input) moved at twice the rate of a 64-byte one, a compact table would pay even when it misses L2.
// Build: hipcc --offload-arch=gfx950 -O3 -o cold_row_size cold_row_size.hip      Run: ./cold_row_size [gathers_millions=128]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int LANES, int U>   // LANES lanes per row: row = LANES x 16 bytes
__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ tab, const uint32_t *__restrict__ idx,
                                                 float4 *__restrict__ out, size_t per_group) {
    const size_t group = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const int c = threadIdx.x % LANES;
    const uint32_t *my = idx + group * per_group;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = 0; i < per_group; i += U) {
        float4 r[U];
#pragma unroll
        for (int k = 0; k < U; ++k) r[k] = tab[(size_t)my[i + k] * LANES + c];
#pragma unroll
        for (int k = 0; k < U; ++k) { acc.x += r[k].x; acc.y += r[k].y; acc.z += r[k].z; acc.w += r[k].w; }
    }
    out[group * LANES + c] = acc;
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

template <int LANES>
void run(const float4 *tab, uint32_t *idx, float4 *out, size_t gathers_req, size_t table_bytes) {
    const uint32_t rows = (uint32_t)(table_bytes / (16 * LANES));
    const size_t threads = (size_t)256 * 16 * 16 * 16;   // 16 waves per CU
    const size_t groups = threads / LANES;
    const size_t per_group = (gathers_req / groups) / 8 * 8;
    const size_t gathers = groups * per_group;
    hipLaunchKernelGGL(k_fill_idx, dim3(4096), dim3(256), 0, 0, idx, gathers, rows);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = (int)(threads / 256);
    hipLaunchKernelGGL((k_gather<LANES, 8>), dim3(blocks), dim3(256), 0, 0, tab, idx, out, per_group);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_gather<LANES, 8>), dim3(blocks), dim3(256), 0, 0, tab, idx, out, per_group);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ms /= 3;
    printf("  %2d-byte rows: %zu gathers in %.3f ms = %.1f G rows/s, %.2f TB/s of rows\n", 16 * LANES, gathers, ms, gathers / ms / 1e6,
           gathers * 16.0 * LANES / ms / 1e9);
}

int main(int argc, char **argv) {
    const size_t gathers_req = (size_t)(argc > 1 ? atof(argv[1]) : 128) * 1000000;
    float4 *tab, *out; uint32_t *idx;
    const size_t max_bytes = (size_t)1074 << 20;
    CK(hipMalloc(&tab, max_bytes + 64));
    CK(hipMalloc(&idx, (gathers_req + (1 << 20)) * 4)); CK(hipMalloc(&out, (size_t)256 * 16 * 16 * 16 * 16));
    hipLaunchKernelGGL(k_fill_tab, dim3(4096), dim3(256), 0, 0, (float *)tab, max_bytes / 4);
    for (size_t mb : {67, 269, 1074}) {
        printf("table %zu MB:\n", mb);
        run<1>(tab, idx, out, gathers_req, mb << 20);
        run<2>(tab, idx, out, gathers_req, mb << 20);
        run<4>(tab, idx, out, gathers_req, mb << 20);
    }
    return 0;
}
