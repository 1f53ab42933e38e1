// Experiment / measurement record (not part of the product): what a streaming kernel sustains on the
// box's HBM3E -- copy (1 read + 1 write) and triad (2 reads + 1 write) over arrays far larger than the
// 256 MB Infinity Cache -- to quote beside the 8 TB/s peak that bench.py prices the roofline with.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm.hip -o exp_libs/ubench_hbm && exp_libs/ubench_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_triad(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                               float4 *__restrict__ c, float s, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 x = a[i], y = b[i];
        c[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}
__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ a, float *out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 x = a[i];
        acc += x.x + x.y + x.z + x.w;
    }
    if (acc == 123.456f) out[0] = acc; // never true: keeps the loads
}

int main() {
    const size_t bytes = (size_t)2 << 30; // 2 GiB per array
    const size_t n = bytes / sizeof(float4);
    float4 *a, *b, *c;
    float *o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&c, bytes) != hipSuccess) return 1;
    (void)hipMalloc(&o, 64);
    (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes); (void)hipMemset(c, 0, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 10;
    printf("{");
    for (int grid : {2048, 8192, 32768}) {
        for (int which = 0; which < 3; ++which) {
            for (int w = 0; w < 2; ++w) {
                if (which == 0) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
                else if (which == 1) hipLaunchKernelGGL(k_triad, dim3(grid), dim3(256), 0, 0, a, b, c, 2.0f, n);
                else hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, o, n);
            }
            (void)hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) {
                if (which == 0) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
                else if (which == 1) hipLaunchKernelGGL(k_triad, dim3(grid), dim3(256), 0, 0, a, b, c, 2.0f, n);
                else hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, o, n);
            }
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double moved = (which == 0 ? 2.0 : which == 1 ? 3.0 : 1.0) * (double)bytes * reps;
            printf("\"%s_grid%d_GBps\": %.1f, ", which == 0 ? "copy" : which == 1 ? "triad" : "read", grid, moved / (ms * 1e-3) / 1e9);
        }
    }
    printf("\"array_GiB\": 2, \"reps\": %d}\n", reps);
    return 0;
}
