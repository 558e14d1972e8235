// Round 2: the alpha >= 1/255 select of the blend loop without a compare.
//
// In a kernel whose fp32 denormal mode is "flush" (hipcc -fgpu-flush-denormals-to-zero), m = alpha * k with
// k = RU(2^-126 / fl(1/255)) is a normal number iff alpha >= fl(1/255) and is flushed to +0 otherwise, so
// a_eff = m * K (K = 1 / k) is alpha (to an ulp) for a hit and exactly 0 for a miss: two full-rate multiplies
// instead of v_cmp + v_cndmask (which issue at half rate on gfx950).
//
// Part 1 checks the claim bit by bit for every float within +-2^20 ulps of the threshold and a sweep of the
// whole range (0, 1]; part 2 times the old and the new loop body in registers (8 waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fgpu-flush-denormals-to-zero flush_select.hip -o flush_select
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

__global__ void k_check(const float *alphas, int n, float k, float K, float thr, unsigned *bad, float *worst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = alphas[i];
    const float m = a * k;
    const float e = m * K;
    const bool hit = a >= thr;
    if ((m != 0.f) != hit) atomicAdd(bad, 1u);
    if (hit) {
        const float rel = fabsf(e - a) / a;
        if (rel > 1.3e-7f) atomicAdd(bad + 1, 1u);
    } else if (e != 0.f) atomicAdd(bad + 2, 1u);
}

template <int MODE>
__global__ __launch_bounds__(64, 8) void k_loop(float *out, int iters, float mx0, float my0, float a, float b, float c,
                                                float lo, float col0, float col1, float col2, float k, float K) {
    const int lane = threadIdx.x;
    const float px = (lane & 7) + 0.5f, py = (lane >> 3) + 0.5f;
    float T = 1.f, thr = 1.f / 255.f, kq = k, p0 = 0.f, p1 = 0.f, p2 = 0.f;
    float mx = mx0, my = my0;
    for (int i = 0; i < iters; i += 2) {
        mx += 0.37f; my -= 0.21f;
        if (mx > 12.f) mx -= 12.f;
        if (my < -4.f) my += 12.f;
        const float mx2 = mx + 0.37f, my2 = my - 0.21f;
        const float dx = mx - px, dy = my - py, ex = mx2 - px, ey = my2 - py;
        const float la = fmaf(dx, fmaf(a, dx, b * dy), fmaf(c * dy, dy, lo));
        const float lb = fmaf(ex, fmaf(a, ex, b * ey), fmaf(c * ey, ey, lo));
        const float al = __builtin_amdgcn_exp2f(la), bl = __builtin_amdgcn_exp2f(lb);
        if (MODE == 0) {          // round 2's loop: compare, compare, select per evaluation
            const bool hit1 = al >= thr;
            const float nT1 = fmaf(-al, T, T);
            const bool add1 = hit1 && nT1 > 1e-4f;
            const float a1 = add1 ? al : 0.f;
            if (__ballot(hit1 && !add1)) { asm volatile("" ::: "memory"); thr = (hit1 && !add1) ? __builtin_huge_valf() : thr; }
            const float v1 = a1 * T;
            p0 = fmaf(col0, v1, p0); p1 = fmaf(col1, v1, p1); p2 = fmaf(col2, v1, p2);
            T = fmaf(-a1, T, T);
            const bool hit2 = bl >= thr;
            const float nT2 = fmaf(-bl, T, T);
            const bool add2 = hit2 && nT2 > 1e-4f;
            const float a2 = add2 ? bl : 0.f;
            if (__ballot(hit2 && !add2)) { asm volatile("" ::: "memory"); thr = (hit2 && !add2) ? __builtin_huge_valf() : thr; }
            const float v2 = a2 * T;
            p0 = fmaf(col0, v2, p0); p1 = fmaf(col1, v2, p1); p2 = fmaf(col2, v2, p2);
            T = fmaf(-a2, T, T);
        } else {                  // flush select, one stop test per pair
            float a1 = (al * kq) * K, a2 = (bl * kq) * K;
            float nT1 = fmaf(-a1, T, T);
            float nT2 = fmaf(-a2, nT1, nT1);
            if (__ballot(!(nT2 > 1e-4f))) {
                asm volatile("" ::: "memory");
                const bool s1 = !(nT1 > 1e-4f);
                a1 = s1 ? 0.f : a1;
                nT1 = s1 ? T : nT1;
                const float t2 = fmaf(-a2, nT1, nT1);
                const bool s2 = s1 || !(t2 > 1e-4f);
                a2 = s2 ? 0.f : a2;
                nT2 = s2 ? nT1 : t2;
                kq = s2 ? 0.f : kq;
            }
            const float v1 = a1 * T, v2 = a2 * nT1;
            p0 = fmaf(col0, v1, p0); p1 = fmaf(col1, v1, p1); p2 = fmaf(col2, v1, p2);
            p0 = fmaf(col0, v2, p0); p1 = fmaf(col1, v2, p1); p2 = fmaf(col2, v2, p2);
            T = nT2;
        }
        mx = mx2; my = my2;
    }
    out[blockIdx.x * 64 + lane] = p0 + p1 + p2 + T + thr + kq;
}

template <int MODE>
static void run(const char *name, float *d, float k, float K) {
    const int iters = 20000, blocks = 256 * 32;   // 32 waves per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_loop<MODE>, dim3(blocks), dim3(64), 0, 0, d, 100, 3.f, 3.f, -0.05f, 0.01f, -0.04f, -0.5f, .3f, .5f, .7f, k, K);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_loop<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 3.f, 3.f, -0.05f, 0.01f, -0.04f, -0.5f, .3f, .5f, .7f, k, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double evals_per_simd = (double)blocks * iters / 1024.0;
    float h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-28s %.3f ms  %.2f ns per evaluation per SIMD   (lane 0 checksum %.6f)\n", name, ms, ms * 1e6 / evals_per_simd, h[0]);
}

int main() {
    const float thr = 1.0f / 255.0f;
    // k = the smallest float with fl(thr * k) >= 2^-126 in exact arithmetic: round 2^-126 / thr up
    const double kd = ldexp(1.0, -126) / (double)thr;
    float k = (float)kd;
    if ((double)k < kd) k = nextafterf(k, 1.f);
    const float K = (float)(1.0 / (double)k);
    printf("thr %.9g (0x%08x)  k %.9g  K %.9g\n", thr, *(const unsigned *)&thr, k, K);

    const int span = 1 << 20, nsweep = 1 << 22;
    const int n = 2 * span + 1 + nsweep;
    float *h = (float *)malloc(n * sizeof(float));
    unsigned tb; memcpy(&tb, &thr, 4);
    for (int i = 0; i <= 2 * span; ++i) { unsigned u = tb - span + i; memcpy(&h[i], &u, 4); }
    for (int i = 0; i < nsweep; ++i) h[2 * span + 1 + i] = (float)((i + 1) / (double)nsweep);   // (0, 1]
    float *d; unsigned *bad; float *worst;
    hipMalloc(&d, n * sizeof(float)); hipMalloc(&bad, 16); hipMalloc(&worst, 4);
    hipMemset(bad, 0, 16);
    hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3((n + 255) / 256), dim3(256), 0, 0, d, n, k, K, thr, bad, worst);
    unsigned hb[4]; hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
    printf("checked %d floats: hit/flush disagreements %u, |a_eff - alpha| > 1.3e-7 alpha: %u, non-zero a_eff on a miss: %u\n",
           n, hb[0], hb[1], hb[2]);

    float *o; hipMalloc(&o, 256 * 32 * 64 * 4);
    run<0>("cmp + cmp + cndmask (round 2)", o, k, K);
    run<1>("flush select, pair stop test", o, k, K);
    return (hb[0] | hb[1] | hb[2]) ? 1 : 0;
}
