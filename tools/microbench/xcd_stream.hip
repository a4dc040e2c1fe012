// xcd_stream.hip — microbenchmark: how fast can the 8 XCDs each pull the SAME table through their own L2
// (the sweep of the compact-table plan: every XCD reads every line of the table once per round; after the
// first XCD the line should come from the memory-side cache), compared with 8 XCDs reading disjoint data?
// Workgroup g is assumed to run on XCD g % 8 (round-robin dispatch); slice g / 8 of 32 per XCD.
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_stream xcd_stream.hip
// Run:   ./xcd_stream [table_MB=160] [sweeps=4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

// mode 0: every XCD reads the whole table (slice g/8 of 32 per workgroup); mode 1: XCD x reads only the x-th
// eighth of an 8x larger region (disjoint data, same total bytes through the fabric)
__global__ __launch_bounds__(1024) void k_sweep(const float4 *__restrict__ tab, size_t quads, int sweeps, int mode,
                                                 float4 *__restrict__ out) {
    const uint32_t xcd = blockIdx.x & 7, slice = blockIdx.x >> 3, nslices = gridDim.x >> 3;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int s = 0; s < sweeps; ++s) {
        const float4 *base = tab + (mode ? (size_t)xcd * quads : 0) + (mode ? 0 : 0);
        // interleave slices at 16 KiB granularity so that the workgroups of an XCD walk the table together
        for (size_t blk = slice; blk * 1024 < quads; blk += nslices) {
            const size_t i = blk * 1024 + threadIdx.x;
            if (i < quads) {
                const float4 v = base[i + (size_t)s * 0];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        __syncthreads();
    }
    if (acc.x == 123.456f) out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? (size_t)atoi(argv[1]) : 160;
    const int sweeps = argc > 2 ? atoi(argv[2]) : 4;
    const size_t quads = mb * 1000000 / 16;
    float4 *tab, *out;
    CK(hipMalloc(&tab, quads * 16 * 8));
    CK(hipMalloc(&out, 256 * 1024 * 16));
    CK(hipMemset(tab, 0, quads * 16 * 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k_sweep, dim3(256), dim3(1024), 0, 0, tab, quads, sweeps, mode, out);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("mode %d (%s): %d sweeps of %zu MB per XCD: %.3f ms = %.2f TB/s into the L2s\n", mode,
                   mode ? "disjoint data" : "same table on every XCD", sweeps, mb, ms, 8.0 * quads * 16 * sweeps / ms / 1e9);
        }
    }
    return 0;
}
