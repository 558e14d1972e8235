// FETCH_SIZE calibration on GATHER patterns (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads half the bytes of a wide
// coalesced streaming read; "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Three kernels with a known byte count each, on a table far larger than the 256 MiB Infinity
// Cache, every record touched at most once:
//   k_stream   float4 streaming read of the whole table                         (the guide's x2 case)
//   k_rec48    one 48-byte record per lane through a random index: 3 x dwordx4  (the rasteriser's staging, round 2)
//   k_aos36    36 bytes per lane from four arrays through a random index: 8 + 12 + 4 + 12 B in seven loads
//              (the rasteriser's staging of round 1)
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv`; scripts/ubench/gather_calib.py
// divides the known bytes by the counter.
//   hipcc --offload-arch=gfx950 -O3 -o gather_calib gather_calib.hip && ./gather_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_stream(const float4 *t, size_t n4, float *out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = t[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ void k_rec48(const float4 *t, const uint32_t *idx, size_t k, float *out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < k; i += (size_t)gridDim.x * blockDim.x) {
        const float4 *r = t + 3 * (size_t)idx[i];
        const float4 a = r[0], b = r[1], c = r[2];
        acc += a.x + b.y + c.z + a.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ void k_aos36(const float *m2, const float *con, const float *op, const float *col, const uint32_t *idx,
                        size_t k, float *out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < k; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = idx[i];
        const float2 m = reinterpret_cast<const float2 *>(m2)[g];
        acc += m.x + m.y + con[3 * g] + con[3 * g + 1] + con[3 * g + 2] + op[g] + col[3 * g] + col[3 * g + 1] + col[3 * g + 2];
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const size_t R = 32u << 20;          // records: 1.5 GiB of 48-byte records
    const size_t K = 8u << 20;           // gathered records per launch (each at most once)
    float4 *table; uint32_t *idx; float *out;
    CK(hipMalloc(&table, R * 48)); CK(hipMalloc(&idx, K * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(table, 0, R * 48));
    // K distinct pseudo-random records: i -> (i * odd) mod R restricted to a stride that keeps them distinct
    uint32_t *h = (uint32_t *)malloc(K * 4);
    for (size_t i = 0; i < K; ++i) h[i] = (uint32_t)(((uint64_t)i * 2654435761ull) % R);   // odd multiplier, R a power of two: a bijection on [0, R)
    CK(hipMemcpy(idx, h, K * 4, hipMemcpyHostToDevice));
    const float *m2 = (const float *)table;                    // four "arrays" inside the big allocation
    const float *con = m2 + 2 * R, *op = con + 3 * R, *col = op + R;   // 8 + 12 + 4 + 12 = 36 B per record index
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, table, R * 3, out);
        hipLaunchKernelGGL(k_rec48, dim3(4096), dim3(256), 0, 0, table, idx, K, out);
        hipLaunchKernelGGL(k_aos36, dim3(4096), dim3(256), 0, 0, m2, con, op, col, idx, K, out);
    }
    CK(hipDeviceSynchronize());
    printf("{\"k_stream\": %zu, \"k_rec48\": %zu, \"k_aos36\": %zu, \"note\": \"known bytes per launch: table / records + 4-byte indices\"}\n",
           R * 48, K * 48 + K * 4, K * 36 + K * 4);
    return 0;
}
