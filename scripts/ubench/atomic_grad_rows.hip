// Microbenchmark: the backward rasteriser's flush shape -- float atomics into random 64-byte rows of a [N][16] table, four
// rows per wave instruction, ACT of each row's 16 lanes active.  How many rows per second does the chip take, and
// does the rate follow the rows (64-byte requests) or the active lanes?
// Build: hipcc --offload-arch=gfx950 -O3 atomic_grad_rows.hip -o atomic_grad_rows
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ACT, int ROWF>   // ROWF floats per row (16: 64-byte rows, 8: 32-byte rows -> eight rows per instruction)
__global__ __launch_bounds__(64) void k(float *tab, unsigned n_rows, int per_wave) {
    const int lane = threadIdx.x, col = lane % ROWF, sub = lane / ROWF;
    unsigned s = blockIdx.x * 2654435761u + 12345u;
    for (int i = 0; i < per_wave; ++i) {
        s = s * 1664525u + 1013904223u;
        const unsigned r = ((s >> 4) + sub * 7919u) % n_rows;
        if (col < ACT) atomicAdd(tab + (size_t)r * ROWF + col, 1.0f);
    }
}
template <int ACT, int ROWF>
void run(float *tab, unsigned n_rows) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int waves = 32768, per_wave = 64;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<ACT, ROWF>), dim3(waves), dim3(64), 0, 0, tab, n_rows, per_wave);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double rows = (double)waves * per_wave * (64 / ROWF);
    printf("row %2d floats, %2d active lanes per row, %u rows in the table: %.1f us for %.2f M rows = %.1f G rows/s, %.2f TB/s of added bytes\n",
           ROWF, ACT, n_rows, best * 1e3, rows / 1e6, rows / (best * 1e6), rows * ACT * 4 / (best * 1e9));
}
int main() {
    for (unsigned n_rows : {1000000u, 4000000u}) {
        float *tab; hipMalloc(&tab, (size_t)n_rows * 64); hipMemset(tab, 0, (size_t)n_rows * 64);
        run<16, 16>(tab, n_rows);
        run<9, 16>(tab, n_rows);
        run<4, 16>(tab, n_rows);
        run<8, 8>(tab, 2 * n_rows);
        run<1, 16>(tab, n_rows);
        hipFree(tab);
    }
    return 0;
}
