// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_exp_f32 vs v_cndmask on gfx950,
// 8 waves/SIMD resident.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    float2v av = {a, a}, bv = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                p0 = __builtin_elementwise_fma(p0, av, bv); p1 = __builtin_elementwise_fma(p1, av, bv);
                p2 = __builtin_elementwise_fma(p2, av, bv); p3 = __builtin_elementwise_fma(p3, av, bv);
                p4 = __builtin_elementwise_fma(p4, av, bv); p5 = __builtin_elementwise_fma(p5, av, bv);
                p6 = __builtin_elementwise_fma(p6, av, bv); p7 = __builtin_elementwise_fma(p7, av, bv);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
                x4 = __builtin_amdgcn_exp2f(x4); x5 = __builtin_amdgcn_exp2f(x5); x6 = __builtin_amdgcn_exp2f(x6); x7 = __builtin_amdgcn_exp2f(x7);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = x0 > a ? x1 : b; x1 = x1 > a ? x2 : b; x2 = x2 > a ? x3 : b; x3 = x3 > a ? x4 : b;
                x4 = x4 > a ? x5 : b; x5 = x5 > a ? x6 : b; x6 = x6 > a ? x7 : b; x7 = x7 > a ? x0 : b;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                          p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}
template <int MODE>
void run(const char *name, float *d, int per_iter_instr) {
    const int iters = 2000, blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves/SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * per_iter_instr;  // wave-instructions
    double per_simd = winstr / 1024.0;
    printf("%-12s %.3f ms  %.2f ns per wave-instr per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd,
           ms * 1e6 / per_simd * 2.4);
}
int main() {
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", d, 64); run<1>("v_pk_fma_f32", d, 64); run<2>("v_exp_f32", d, 64); run<3>("cmp+cndmask", d, 128);
    return 0;
}
