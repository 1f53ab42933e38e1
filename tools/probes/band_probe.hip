#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double t = -40.0 - 706.0 * (double)i / n;
    double x = exp(t);
    double y = log1p(x);
    out[3 * i] = t; out[3 * i + 1] = x; out[3 * i + 2] = y;
}
int main() {
    const int n = 1 << 20;
    double *d; hipMalloc(&d, n * 24);
    k<<<n / 256, 256>>>(d, n);
    double *h = (double *)malloc(n * 24);
    hipMemcpy(h, d, n * 24, hipMemcpyDeviceToHost);
    long bad = 0; double tmin = 0, tmax = -1e9; int shown = 0;
    for (int i = 0; i < n; ++i) {
        if (h[3 * i + 1] != h[3 * i + 2]) {
            ++bad; if (h[3 * i] < tmin) tmin = h[3 * i]; if (h[3 * i] > tmax) tmax = h[3 * i];
            if (shown < 6) { printf("t=%.6f exp=%a log1p=%a host_exp=%a host_log1p(exp)=%a\n", h[3*i], h[3*i+1], h[3*i+2], exp(h[3*i]), log1p(exp(h[3*i]))); ++shown; }
        }
    }
    printf("bad=%ld of %d  t range of mismatches [%f, %f]\n", bad, n, tmin, tmax);
    long dexp = 0; for (int i = 0; i < n; ++i) dexp += (h[3*i+1] != exp(h[3*i]));
    printf("device exp != host exp: %ld\n", dexp);
    return 0;
}
