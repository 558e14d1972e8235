// Microbenchmark: G workgroups each add a row of T counters into T global totals with RETURNING atomics (what a
// count kernel would do to get its base inside every tile segment without a per-tile prefix kernel), against the
// per-tile prefix kernel's own traffic.  Build: hipcc --offload-arch=gfx950 -O3 atomic_rows.hip -o atomic_rows
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int MODE>   // 0: returning atomics, rotated start; 1: returning, same order in every workgroup; 2: non-returning
__global__ __launch_bounds__(1024) void k(unsigned *tot, unsigned *base, int T, int spin) {
    // some work first so that the workgroups do not all arrive in the same microsecond (as after a real count pass)
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    const int rot = MODE == 0 ? (int)((blockIdx.x * 2654435761u) % (unsigned)T) : 0;
    for (int t = threadIdx.x; t < T; t += 1024) {
        int tt = t + rot; if (tt >= T) tt -= T;
        const unsigned c = 1u + ((blockIdx.x + tt) & 3u) + (x < 0.f ? 1u : 0u);
        if (MODE == 2) atomicAdd(&tot[tt], c);
        else base[(size_t)blockIdx.x * T + tt] = atomicAdd(&tot[tt], c);
    }
}
template <int MODE>
void run(const char *name, unsigned *tot, unsigned *base, int G, int T) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spin : {0, 2000}) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipMemsetAsync(tot, 0, T * 4, 0);
            hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(1024), 0, 0, tot, base, T, spin);   // warm
            hipMemsetAsync(tot, 0, T * 4, 0);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(1024), 0, 0, tot, base, T, spin);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-28s G=%d T=%d spin=%d: %.1f us\n", name, G, T, spin, best * 1e3f);
    }
}
int main() {
    const int G = 489;
    for (int T : {2040, 8160}) {
        unsigned *tot, *base; hipMalloc(&tot, T * 4); hipMalloc(&base, (size_t)G * T * 4);
        run<0>("returning, rotated", tot, base, G, T);
        run<1>("returning, same order", tot, base, G, T);
        run<2>("non-returning", tot, base, G, T);
        std::vector<unsigned> h(T); hipMemcpy(h.data(), tot, T * 4, hipMemcpyDeviceToHost);
        unsigned long long s = 0; for (unsigned v : h) s += v;
        printf("  checksum %llu\n", s);
        hipFree(tot); hipFree(base);
    }
    return 0;
}
