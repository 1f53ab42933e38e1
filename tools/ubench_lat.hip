// Experiment (not part of the product): gfx950 latency microbenchmarks behind the step kernel's
// design decisions -- dependent fp64 op latency, VALU->SALU->VALU round trip, LDS round trip,
// workgroup barrier cost.  One workgroup; clock64() around R repetitions of each pattern.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_lat.hip -o exp_libs/ubench_lat && exp_libs/ubench_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define R 2000

__global__ void k_lat(double *out, unsigned long long *cyc, double seed, int nwaves_active) {
    __shared__ double sh[4][64];
    __shared__ int ring[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double x = seed + lane * 1e-3, y = seed * 0.5, z = 1.0000001, w = 0.999999;
    unsigned long long t0, t1;
    int slot = 0;
#define BEGIN() __syncthreads(); t0 = clock64();
#define END()   t1 = clock64(); if (threadIdx.x == 0) cyc[slot] = (t1 - t0); slot++;

    // 0: dependent v_fma_f64 chain
    BEGIN();
    for (int r = 0; r < R; ++r) { x = __fma_rn(x, z, y); x = __fma_rn(x, w, y); x = __fma_rn(x, z, y); x = __fma_rn(x, w, y); }
    asm volatile("" : "+v"(x));
    END();   // 4R ops
    // 1: four independent fma chains
    {
        double a = x, b = x + 1, c = x + 2, d = x + 3;
        BEGIN();
        for (int r = 0; r < R; ++r) { a = __fma_rn(a, z, y); b = __fma_rn(b, w, y); c = __fma_rn(c, z, y); d = __fma_rn(d, w, y); }
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        END(); // 4R ops
        x = a + b + c + d;
    }
    // 2: dependent v_add_f64 chain
    BEGIN();
    for (int r = 0; r < R; ++r) { x = x + y; x = x + z; x = x + y; x = x + w; }
    asm volatile("" : "+v"(x));
    END();
    // 3: dependent v_mul_f64 chain
    BEGIN();
    for (int r = 0; r < R; ++r) { x = x * z; x = x * w; x = x * z; x = x * w; }
    asm volatile("" : "+v"(x));
    END();
    // 4: dependent 32-bit VALU chain (v_add_u32)
    {
        unsigned u = lane;
        BEGIN();
        for (int r = 0; r < R; ++r) { asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(u) : "v"(lane)); }
        END();
        x += u;
    }
    // 5: v_cmp_f64 -> s_and -> v_cndmask (64-bit select = 2 cndmask) round trip, dependent
    BEGIN();
    for (int r = 0; r < R; ++r) {
        const bool c1 = x > y;
        const unsigned long long b = __ballot(c1) & 0x5555555555555555ull;
        x = ((b >> lane) & 1ull) ? x + z : x - w; // depends on the SALU result
        asm volatile("" : "+v"(x));
    }
    END(); // R iterations
    // 6: LDS write -> read round trip (same wave, dependent)
    BEGIN();
    for (int r = 0; r < R; ++r) {
        sh[wave][lane] = x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        x = sh[wave][(lane + 1) & 63] + z;
        asm volatile("" : "+v"(x));
    }
    END();
    // 7: workgroup barrier, all waves arriving together (raw s_barrier behind an LDS wait)
    BEGIN();
    for (int r = 0; r < R; ++r) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    END();
    // 8: LDS pointer chase (ds_read_b32 latency)
    ring[threadIdx.x & 255] = (threadIdx.x + 1) & 255;
    __syncthreads();
    {
        int p = lane;
        BEGIN();
        for (int r = 0; r < R; ++r) { p = ring[p]; asm volatile("" : "+v"(p)); }
        END();
        x += p;
    }
    // 9: v_rsq_f64 + v_rcp_f64 dependent
    BEGIN();
    for (int r = 0; r < R; ++r) { x = __builtin_amdgcn_rsq(x + 2.0); x = __builtin_amdgcn_rcp(x + 2.0); }
    asm volatile("" : "+v"(x));
    END(); // 2R trans + 2R add
    // 10: barrier + LDS write before / read after (producer-consumer round trip between waves)
    BEGIN();
    for (int r = 0; r < R; ++r) {
        sh[wave][lane] = x;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        x = sh[(wave + 1) & 3][lane] + z;
        asm volatile("" : "+v"(x));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    END(); // 2 barriers + LDS round trip per iteration
    // 11: s_memtime cost (clock64 back to back)
    {
        unsigned long long acc = 0;
        BEGIN();
        for (int r = 0; r < R; ++r) { acc += clock64(); }
        END();
        x += (double)(acc & 1);
    }
    // 12: 64-bit cndmask dependent chain via v_cmp (VCC) only, no SALU op in between
    BEGIN();
    for (int r = 0; r < R; ++r) { x = (x > y) ? x + z : x - w; asm volatile("" : "+v"(x)); }
    END();
    // 13: ds_read_b64 x8 issue + wait (batch of 8 independent reads)
    BEGIN();
    for (int r = 0; r < R; ++r) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += sh[k & 3][(lane + k) & 63];
        x += s;
        asm volatile("" : "+v"(x));
    }
    END();
    // 14/15: a wave-uniform forward branch per iteration, never / always jumping over 4 VALU ops
    // (nwaves_active is 1 or 4 at run time, the compiler cannot fold the conditions)
    for (int taken = 0; taken < 2; ++taken) {
        const int thr = taken ? 0 : 100;   // skip when nwaves_active > thr
        BEGIN();
        for (int r = 0; r < R; ++r) {
            x = __fma_rn(x, z, y);
            asm volatile("s_cmp_gt_i32 %1, %2\n\ts_cbranch_scc1 1f\n\tv_add_f64 %0, %0, %0\n\tv_add_f64 %0, %0, %0\n\t"
                         "v_add_f64 %0, %0, %0\n\tv_add_f64 %0, %0, %0\n1:" : "+v"(x) : "s"(nwaves_active), "s"(thr) : "scc");
        }
        END();
    }
    // 16: a divergent if (exec-masked, s_and_saveexec + s_cbranch_execz, never skipping) per iteration
    BEGIN();
    for (int r = 0; r < R; ++r) {
        if (x > (double)lane - 1e300) x = __fma_rn(x, z, y);
        asm volatile("" : "+v"(x));
    }
    END();
    out[threadIdx.x] = x;
}

int main() {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * sizeof(double));
    hipMalloc(&cyc, 32 * sizeof(unsigned long long));
    const char *names[] = {"dep fma_f64 (4R ops)", "4 indep fma_f64 chains (4R ops)", "dep add_f64 (4R)", "dep mul_f64 (4R)",
                           "dep 32-bit valu (4R)", "cmp->ballot->select round trip (R)", "LDS write->read same wave (R)",
                           "s_barrier, 4 waves (R)", "LDS pointer chase (R)", "rsq+rcp f64 dep (R pairs)",
                           "LDS write, barrier, read, barrier (R)", "clock64 back to back (R)", "cmp->VCC->select dep (R)",
                           "8 ds_read_b64 + sum (R)", "fma + uniform branch NOT taken over 4 adds (R)",
                           "fma + uniform branch TAKEN over 4 adds (R)", "divergent if, all lanes in (R)"};
    for (int waves = 1; waves <= 4; waves += 3) {
        hipLaunchKernelGGL(k_lat, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.25, waves);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k_lat, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.25, waves);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(32);
        hipMemcpy(h.data(), cyc, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        printf("== %d wave(s) in the workgroup\n", waves);
        for (int k = 0; k < 17; ++k) printf("%-44s %8.1f cycles per R\n", names[k], (double)h[k] / R);
    }
    return 0;
}
