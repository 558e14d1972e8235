// Microbenchmark of the rasteriser's per-(quad, Gaussian) evaluation in registers only (no LDS, no
// memory): which instructions carry the cost on gfx950?  8 waves/SIMD resident.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize raster_loop.hip -o raster_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(64, 8) void k(float *out, int iters, float mx0, float my0, float a, float b, float c,
                                           float lo, float col0, float col1, float col2) {
    const int lane = threadIdx.x;
    const float px = (lane & 7) + 0.5f, py = (lane >> 3) + 0.5f;
    float T = 1.f, thr = 1.f / 255.f, p0 = 0.f, p1 = 0.f, p2 = 0.f;
    float mx = mx0, my = my0;
    for (int i = 0; i < iters; ++i) {
        mx += 0.37f; my -= 0.21f;                     // stands in for the LDS record changing per entry
        if (mx > 12.f) mx -= 12.f;
        if (my < -4.f) my += 12.f;
        const float dx = mx - px, dy = my - py;
        const float la = fmaf(dx, fmaf(a, dx, b * dy), fmaf(c * dy, dy, lo));
        float alpha;
        if (MODE == 2) alpha = fminf(0.999f, la * 0.01f + 0.5f);            // no exp
        else alpha = fminf(0.999f, __builtin_amdgcn_exp2f(la));
        const float next_T = fmaf(-alpha, T, T);
        float a_eff;
        if (MODE == 1) a_eff = alpha * 0.001f;                              // no compares / select
        else {
            const bool hit = la <= lo && alpha >= thr;
            const bool add = hit && next_T > 1e-4f;
            a_eff = add ? alpha : 0.f;
            if (__ballot(hit && !add)) { asm volatile("" ::: "memory"); thr = (hit && !add) ? __builtin_huge_valf() : thr; }
        }
        const float vis = a_eff * T;
        p0 = fmaf(col0, vis, p0); p1 = fmaf(col1, vis, p1); p2 = fmaf(col2, vis, p2);
        T = fmaf(-a_eff, T, T);
    }
    out[blockIdx.x * 64 + lane] = p0 + p1 + p2 + T + thr;
}
__global__ __launch_bounds__(64, 8) void k2(float *out, int iters, float mx0, float my0, float a, float b, float c,
                                            float lo, float col0, float col1, float col2) {
    const int lane = threadIdx.x;
    const float px = (lane & 7) + 0.5f, py = (lane >> 3) + 0.5f;
    float T = 1.f, thr = 1.f / 255.f, p0 = 0.f, p1 = 0.f, p2 = 0.f;
    float mx = mx0, my = my0;
    for (int i = 0; i < iters; i += 2) {
        mx += 0.37f; my -= 0.21f;
        if (mx > 12.f) mx -= 12.f;
        if (my < -4.f) my += 12.f;
        const float mx2 = mx + 0.37f, my2 = my - 0.21f;
        const float dx = mx - px, dy = my - py, ex = mx2 - px, ey = my2 - py;
        const float la = fmaf(dx, fmaf(a, dx, b * dy), fmaf(c * dy, dy, lo));
        const float lb = fmaf(ex, fmaf(a, ex, b * ey), fmaf(c * ey, ey, lo));
        const float al = fminf(0.999f, __builtin_amdgcn_exp2f(la));
        const float bl = fminf(0.999f, __builtin_amdgcn_exp2f(lb));
        const bool hit1 = la <= lo && al >= thr;
        const bool hit2pre = lb <= lo;
        // entry 1
        const float nT1 = fmaf(-al, T, T);
        const bool add1 = hit1 && nT1 > 1e-4f;
        const float a1 = add1 ? al : 0.f;
        if (__ballot(hit1 && !add1)) { asm volatile("" ::: "memory"); thr = (hit1 && !add1) ? __builtin_huge_valf() : thr; }
        const float v1 = a1 * T;
        p0 = fmaf(col0, v1, p0); p1 = fmaf(col1, v1, p1); p2 = fmaf(col2, v1, p2);
        T = fmaf(-a1, T, T);
        // entry 2
        const bool hit2 = hit2pre && bl >= thr;
        const float nT2 = fmaf(-bl, T, T);
        const bool add2 = hit2 && nT2 > 1e-4f;
        const float a2 = add2 ? bl : 0.f;
        if (__ballot(hit2 && !add2)) { asm volatile("" ::: "memory"); thr = (hit2 && !add2) ? __builtin_huge_valf() : thr; }
        const float v2 = a2 * T;
        p0 = fmaf(col0, v2, p0); p1 = fmaf(col1, v2, p1); p2 = fmaf(col2, v2, p2);
        T = fmaf(-a2, T, T);
        mx = mx2; my = my2;
    }
    out[blockIdx.x * 64 + lane] = p0 + p1 + p2 + T + thr;
}
template <int MODE>
void run(const char *name, float *d) {
    const int iters = 20000, blocks = 256 * 32;   // 32 waves per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 100, 3.f, 3.f, -0.05f, 0.01f, -0.04f, -0.5f, .3f, .5f, .7f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 3.f, 3.f, -0.05f, 0.01f, -0.04f, -0.5f, .3f, .5f, .7f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double evals_per_simd = (double)blocks * iters / 1024.0;
    printf("%-22s %.3f ms  %.1f ns per eval per SIMD (= %.1f cycles at 2.3 GHz)\n", name, ms, ms * 1e6 / evals_per_simd,
           ms * 1e6 / evals_per_simd * 2.3);
}
int main() {
    float *d; hipMalloc(&d, 256 * 32 * 64 * 4);
    run<0>("full body", d); run<1>("no compares/select", d); run<2>("no exp", d);
    {
        const int iters = 20000, blocks = 256 * 32;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k2, dim3(blocks), dim3(64), 0, 0, d, 100, 3.f, 3.f, -0.05f, 0.01f, -0.04f, -0.5f, .3f, .5f, .7f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k2, dim3(blocks), dim3(64), 0, 0, d, iters, 3.f, 3.f, -0.05f, 0.01f, -0.04f, -0.5f, .3f, .5f, .7f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double evals_per_simd = (double)blocks * iters / 1024.0;
        printf("%-22s %.3f ms  %.1f ns per eval per SIMD (= %.1f cycles at 2.3 GHz)\n", "2 entries interleaved", ms,
               ms * 1e6 / evals_per_simd, ms * 1e6 / evals_per_simd * 2.3);
    }
    return 0;
}
