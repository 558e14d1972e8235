// Microbenchmark: issue rate of the integer multiplies and 64-bit shifts the binning kernels' tile walks use, on gfx950,
// 8 waves/SIMD resident: v_mul_lo_u32 against v_mul_u32_u24 / v_mad_u32_u24, v_lshlrev_b64 against v_lshlrev_b32, v_add_u32 as
// the yardstick.  Build: hipcc --offload-arch=gfx950 -O3 int_rate.hip -o int_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned a, unsigned b) {
    unsigned x[8];
    unsigned long long y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = threadIdx.x + j; y[j] = threadIdx.x + j; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 0) x[j] = x[j] + a;                                   // v_add_u32
                else if (MODE == 1) x[j] = x[j] * a;                              // v_mul_lo_u32
                else if (MODE == 2) x[j] = __umul24(x[j], a);                     // v_mul_u32_u24
                else if (MODE == 3) x[j] = __umul24(x[j], a) + b;                 // v_mad_u32_u24
                else if (MODE == 4) y[j] = y[j] << (a & 63);                      // v_lshlrev_b64
                else x[j] = x[j] << (a & 31);                                     // v_lshlrev_b32
                asm volatile("" : "+v"(x[j]));
            }
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j] + (unsigned)y[j] + (unsigned)(y[j] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, unsigned *d) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 3u, 5u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u, 5u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)blocks * 4 * iters * 64 / 1024.0;
    printf("%-16s %.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e6 / per_simd * 2.4);
}
int main() {
    unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_u32_u24", d); run<3>("v_mad_u32_u24", d);
    run<4>("v_lshlrev_b64", d); run<5>("v_lshlrev_b32", d);
    return 0;
}
