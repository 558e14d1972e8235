// View-dependent colour from spherical-harmonic coefficients.
//
// The reference leaves this as a TODO -- render_gaussians(sh_degree=...) only slices channels
// (mojosplat/render.py:82-87) -- and names gsplat's convention as the target: real SH basis of
// degree <= 4 in the 3DGS sign convention ((-1)^m times the Condon-Shortley-free real harmonics,
// index l*(l+1)+m), evaluated along normalise(mean - camera position), colour = max(sum + 0.5, 0).
//
// HBM bound: 12*K bytes of coefficients per Gaussian (192 B at degree 3) against ~100 flops.
// A lane owns one Gaussian, but its coefficients are a 12*K-byte record -- read lane-by-lane that
// is 64 scattered records per wave instruction.  So a wave stages the 64 records of its Gaussians
// through LDS with fully coalesced loads (consecutive lanes read consecutive 16-byte / 4-byte
// words of the wave's contiguous block), then every lane picks its record up from LDS at an odd
// word stride (no bank conflicts).  The backward pass mirrors it for the v_coeffs store.
// The basis is evaluated once, in a scalar type T: float for values, a 3-direction dual number
// for the derivatives the backward needs (no hand-derived gradient tables to get wrong).
#include "ms_common.hpp"

namespace {

constexpr int kShThreads = 256;          // 4 waves; a wave owns 64 consecutive Gaussians

struct Dual3 {
    float v, dx, dy, dz;
};
__device__ __forceinline__ Dual3 operator+(Dual3 a, Dual3 b) { return {a.v + b.v, a.dx + b.dx, a.dy + b.dy, a.dz + b.dz}; }
__device__ __forceinline__ Dual3 operator-(Dual3 a, Dual3 b) { return {a.v - b.v, a.dx - b.dx, a.dy - b.dy, a.dz - b.dz}; }
__device__ __forceinline__ Dual3 operator*(Dual3 a, Dual3 b) {
    return {a.v * b.v, a.dx * b.v + a.v * b.dx, a.dy * b.v + a.v * b.dy, a.dz * b.v + a.v * b.dz};
}
__device__ __forceinline__ Dual3 operator*(float s, Dual3 a) { return {s * a.v, s * a.dx, s * a.dy, s * a.dz}; }
__device__ __forceinline__ Dual3 operator+(Dual3 a, float s) { return {a.v + s, a.dx, a.dy, a.dz}; }
__device__ __forceinline__ Dual3 operator-(Dual3 a, float s) { return {a.v - s, a.dx, a.dy, a.dz}; }

// b[k], k < (DEG+1)^2, for a UNIT direction (x, y, z).  Zonal/sectorial recurrences after
// P.-P. Sloan, "Efficient Spherical Harmonic Evaluation" (JCGT 2013) -- the formulation gsplat's
// spherical_harmonics uses; the oracle restates the same basis as explicit polynomials.
template <int DEG, class T>
__device__ __forceinline__ void sh_basis(T x, T y, T z, T *b) {
    b[0] = 0.0f * x + 0.2820947917738781f;
    if (DEG < 1) return;
    b[1] = -0.48860251190292f * y;
    b[2] = 0.48860251190292f * z;
    b[3] = -0.48860251190292f * x;
    if (DEG < 2) return;
    const T z2 = z * z;
    const T fTmp0B = -1.092548430592079f * z;
    const T fC1 = x * x - y * y;
    const T fS1 = 2.0f * (x * y);
    b[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
    b[7] = fTmp0B * x;
    b[5] = fTmp0B * y;
    b[8] = 0.5462742152960395f * fC1;
    b[4] = 0.5462742152960395f * fS1;
    if (DEG < 3) return;
    const T fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    const T fTmp1B = 1.445305721320277f * z;
    const T fC2 = x * fC1 - y * fS1;
    const T fS2 = x * fS1 + y * fC1;
    b[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
    b[13] = fTmp0C * x;
    b[11] = fTmp0C * y;
    b[14] = fTmp1B * fC1;
    b[10] = fTmp1B * fS1;
    b[15] = -0.5900435899266435f * fC2;
    b[9] = -0.5900435899266435f * fS2;
    if (DEG < 4) return;
    const T fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    const T fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
    const T fTmp2B = -1.770130769779931f * z;
    const T fC3 = x * fC2 - y * fS2;
    const T fS3 = x * fS2 + y * fC2;
    b[20] = 1.984313483298443f * (z * b[12]) - 1.006230589874905f * b[6];
    b[21] = fTmp0D * x;
    b[19] = fTmp0D * y;
    b[22] = fTmp1C * fC1;
    b[18] = fTmp1C * fS1;
    b[23] = fTmp2B * fC2;
    b[17] = fTmp2B * fS2;
    b[24] = 0.6258357354491763f * fC3;
    b[16] = 0.6258357354491763f * fS3;
}

struct ShArgs {
    int64_t N;
    int K;            // coefficients per Gaussian in the tensor (>= (DEG+1)^2)
    float cx, cy, cz; // camera position (world)
    int clamp;        // colour = max(sum + 0.5, 0) when set, the raw sum otherwise
};

template <int KU>
struct ShCfg {
    static constexpr int R = KU * 3;
    static constexpr int STR = (R & 1) ? R : R + 1;
    static constexpr size_t LDS = (size_t)(kShThreads / 64) * 64 * STR * sizeof(float);
};

// Stage the first KU*3 floats of each of this wave's 64 records into s_rec (odd row stride STR).
template <int KU>
__device__ __forceinline__ void stage_records(const float *__restrict__ coeffs, int64_t g0, int64_t N, int K,
                                              unsigned long long active, float *s_rec) {
    constexpr int R = ShCfg<KU>::R, STR = ShCfg<KU>::STR;
    const int lane = threadIdx.x & 63;
    const int64_t left = N - g0;
    const int rows = left < 64 ? (int)left : 64;
    if (K == KU && rows == 64 && ((uintptr_t)coeffs & 15) == 0) {
        // the wave's block is one contiguous run of 64*R floats starting 16-byte aligned
        // (64*R*4 bytes per wave): float4 loads; a float4 may straddle two records
        const float4 *src = reinterpret_cast<const float4 *>(coeffs + g0 * R);
        constexpr int n4 = 64 * R / 4;
        for (int i = lane; i < n4; i += 64) {
            const int f = i * 4;
            const int ra = f / R, rb = (f + 3) / R;
            if (!(((active >> ra) | (active >> rb)) & 1ull)) continue;
            const float4 v = src[i];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = (f + j) / R, col = (f + j) - row * R;
                s_rec[row * STR + col] = e[j];  // rows of culled Gaussians are never read back
            }
        }
    } else {
        const int n = rows * R;
        for (int i = lane; i < n; i += 64) {
            const int row = i / R, col = i - row * R;
            if (!((active >> row) & 1ull)) continue;
            s_rec[row * STR + col] = coeffs[(g0 + row) * (int64_t)K * 3 + col];
        }
    }
}

template <int DEG, class OutT>
__global__ __launch_bounds__(kShThreads) void k_sh_fwd(ShArgs A, const float *__restrict__ means3d,
                                                       const float *__restrict__ coeffs,
                                                       const int32_t *__restrict__ radii, OutT *__restrict__ colors) {
    constexpr int KU = (DEG + 1) * (DEG + 1);
    constexpr int STR = ShCfg<KU>::STR;
    extern __shared__ float s_all[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *s_rec = s_all + (size_t)w * 64 * STR;
    const int64_t g0 = ((int64_t)blockIdx.x * (kShThreads / 64) + w) * 64;
    if (g0 >= A.N) return;  // whole wave out of range (no block-level barrier below)
    const int64_t g = g0 + lane;
    bool on = g < A.N;
    if (on && radii) {
        const int2 r = reinterpret_cast<const int2 *>(radii)[g];
        on = r.x > 0 && r.y > 0;
    }
    const unsigned long long active = __ballot(on);
    stage_records<KU>(coeffs, g0, A.N, A.K, active, s_rec);
    // wave-private LDS region, written and read by the same wave: the wave's own program order
    // plus an LDS fence is enough, no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (g >= A.N) return;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    if (on) {
        float x = means3d[3 * g] - A.cx, y = means3d[3 * g + 1] - A.cy, z = means3d[3 * g + 2] - A.cz;
        const float inorm = rsqrtf(x * x + y * y + z * z);
        x *= inorm; y *= inorm; z *= inorm;
        float b[KU];
        sh_basis<DEG, float>(x, y, z, b);
        const float *rec = s_rec + lane * STR;
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            r0 += b[k] * rec[3 * k];
            r1 += b[k] * rec[3 * k + 1];
            r2 += b[k] * rec[3 * k + 2];
        }
        if (A.clamp) {
            r0 = fmaxf(r0 + 0.5f, 0.f);
            r1 = fmaxf(r1 + 0.5f, 0.f);
            r2 = fmaxf(r2 + 0.5f, 0.f);
        }
    }
    colors[3 * g] = (OutT)r0;
    colors[3 * g + 1] = (OutT)r1;
    colors[3 * g + 2] = (OutT)r2;
}

// Backward.  v_coeffs[g,k,c] = b_k * v_c (zero rows beyond the used degree and for masked
// Gaussians); v_means3d = J_normalise^T * sum_k sum_c coeff[k,c] v_c grad(b_k).
template <int DEG>
__global__ __launch_bounds__(kShThreads) void k_sh_bwd(ShArgs A, const float *__restrict__ means3d,
                                                       const float *__restrict__ coeffs,
                                                       const int32_t *__restrict__ radii,
                                                       const float *__restrict__ colors_fwd,
                                                       const float *__restrict__ v_colors,
                                                       float *__restrict__ v_coeffs, float *__restrict__ v_means3d) {
    constexpr int KU = (DEG + 1) * (DEG + 1);
    constexpr int STR = ShCfg<KU>::STR;
    extern __shared__ float s_all[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *s_rec = s_all + (size_t)w * 64 * STR;
    const int64_t g0 = ((int64_t)blockIdx.x * (kShThreads / 64) + w) * 64;
    if (g0 >= A.N) return;
    const int64_t g = g0 + lane;
    bool on = g < A.N;
    if (on && radii) {
        const int2 r = reinterpret_cast<const int2 *>(radii)[g];
        on = r.x > 0 && r.y > 0;
    }
    const unsigned long long active = __ballot(on);
    if (v_means3d) {
        stage_records<KU>(coeffs, g0, A.N, A.K, active, s_rec);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    float vc0 = 0.f, vc1 = 0.f, vc2 = 0.f;
    float bval[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) bval[k] = 0.f;
    if (on) {
        vc0 = v_colors[3 * g]; vc1 = v_colors[3 * g + 1]; vc2 = v_colors[3 * g + 2];
        if (A.clamp) {  // max(. + 0.5, 0): no gradient where the forward clamped
            if (!(colors_fwd[3 * g] > 0.f)) vc0 = 0.f;
            if (!(colors_fwd[3 * g + 1] > 0.f)) vc1 = 0.f;
            if (!(colors_fwd[3 * g + 2] > 0.f)) vc2 = 0.f;
        }
        float x = means3d[3 * g] - A.cx, y = means3d[3 * g + 1] - A.cy, z = means3d[3 * g + 2] - A.cz;
        const float inorm = rsqrtf(x * x + y * y + z * z);
        x *= inorm; y *= inorm; z *= inorm;
        if (v_means3d) {
            Dual3 b[KU];
            sh_basis<DEG, Dual3>(Dual3{x, 1.f, 0.f, 0.f}, Dual3{y, 0.f, 1.f, 0.f}, Dual3{z, 0.f, 0.f, 1.f}, b);
            const float *rec = s_rec + lane * STR;
            float ux = 0.f, uy = 0.f, uz = 0.f;
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                bval[k] = b[k].v;
                const float s = rec[3 * k] * vc0 + rec[3 * k + 1] * vc1 + rec[3 * k + 2] * vc2;
                ux += s * b[k].dx; uy += s * b[k].dy; uz += s * b[k].dz;
            }
            // through d = p / |p|: (I - d d^T) u / |p|
            const float dot = ux * x + uy * y + uz * z;
            v_means3d[3 * g] = (ux - dot * x) * inorm;
            v_means3d[3 * g + 1] = (uy - dot * y) * inorm;
            v_means3d[3 * g + 2] = (uz - dot * z) * inorm;
        } else {
            sh_basis<DEG, float>(x, y, z, bval);
        }
    } else if (g < A.N && v_means3d) {
        v_means3d[3 * g] = 0.f; v_means3d[3 * g + 1] = 0.f; v_means3d[3 * g + 2] = 0.f;
    }
    if (!v_coeffs) return;
    // outer products through LDS so the K*12-byte rows leave as coalesced stores
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // all lanes are done reading s_rec
    float *row = s_rec + lane * STR;
#pragma unroll
    for (int k = 0; k < KU; ++k) {
        row[3 * k] = bval[k] * vc0;
        row[3 * k + 1] = bval[k] * vc1;
        row[3 * k + 2] = bval[k] * vc2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t left = A.N - g0;
    const int rows = left < 64 ? (int)left : 64;
    constexpr int R = KU * 3;
    if (A.K == KU && rows == 64 && ((uintptr_t)v_coeffs & 15) == 0) {
        float4 *dst4 = reinterpret_cast<float4 *>(v_coeffs + g0 * R);
        constexpr int n4 = 64 * R / 4;
        for (int i = lane; i < n4; i += 64) {
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = i * 4 + j, r = f / R;
                e[j] = s_rec[r * STR + (f - r * R)];
            }
            dst4[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
        return;
    }
    const int RK = A.K * 3;
    float *dst = v_coeffs + g0 * (int64_t)RK;
    for (int r = 0; r < rows; ++r)  // row by row: no runtime division, each row a coalesced run
        for (int c = lane; c < RK; c += 64) dst[(int64_t)r * RK + c] = c < R ? s_rec[r * STR + c] : 0.f;
}

template <int DEG>
int launch_fwd(const ShArgs &A, const float *means3d, const float *coeffs, const int32_t *radii, void *colors,
               int color_dtype, hipStream_t stream) {
    constexpr int KU = (DEG + 1) * (DEG + 1);
    const unsigned grid = (unsigned)ms::ceil_div(A.N, kShThreads);
    const size_t lds = ShCfg<KU>::LDS;
    if (color_dtype == MS_COLOR_F16) {
        if (lds > 48 * 1024)
            MS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sh_fwd<DEG, _Float16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_sh_fwd<DEG, _Float16>), dim3(grid), dim3(kShThreads), lds, stream, A, means3d, coeffs,
                           radii, (_Float16 *)colors);
    } else {
        if (lds > 48 * 1024)
            MS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sh_fwd<DEG, float>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_sh_fwd<DEG, float>), dim3(grid), dim3(kShThreads), lds, stream, A, means3d, coeffs,
                           radii, (float *)colors);
    }
    MS_LAUNCH_CHECK();
    return MS_OK;
}

template <int DEG>
int launch_bwd(const ShArgs &A, const float *means3d, const float *coeffs, const int32_t *radii,
               const float *colors_fwd, const float *v_colors, float *v_coeffs, float *v_means3d,
               hipStream_t stream) {
    constexpr int KU = (DEG + 1) * (DEG + 1);
    const unsigned grid = (unsigned)ms::ceil_div(A.N, kShThreads);
    const size_t lds = ShCfg<KU>::LDS;
    if (lds > 48 * 1024)
        MS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sh_bwd<DEG>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_sh_bwd<DEG>), dim3(grid), dim3(kShThreads), lds, stream, A, means3d, coeffs, radii,
                       colors_fwd, v_colors, v_coeffs, v_means3d);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

int check_sh(int64_t N, int K, int degree, const char *what) {
    MS_REQUIRE(N >= 0, MS_ERR_INVALID_ARG, "%s: N < 0", what);
    MS_REQUIRE(degree >= 0 && degree <= 4, MS_ERR_INVALID_ARG, "%s: degree %d not in [0, 4]", what, degree);
    MS_REQUIRE(K >= (degree + 1) * (degree + 1) && K <= 1024, MS_ERR_INVALID_ARG,
               "%s: %d coefficients per Gaussian cannot hold degree %d (needs %d)", what, K, degree,
               (degree + 1) * (degree + 1));
    MS_REQUIRE(ms::ceil_div(N > 0 ? N : 1, kShThreads) <= 0x7fffffff, MS_ERR_TOO_LARGE, "%s: N too large", what);
    return MS_OK;
}

}  // namespace

extern "C" int ms_spherical_harmonics_fwd(int64_t N, int K, int degree, const float *means3d, float cam_x,
                                          float cam_y, float cam_z, const float *coeffs, const int32_t *radii,
                                          int add_half_and_clamp, int color_dtype, void *colors, void *stream_) {
    if (int rc = check_sh(N, K, degree, "sh_fwd")) return rc;
    if (N == 0) return MS_OK;
    MS_REQUIRE(means3d && coeffs && colors, MS_ERR_INVALID_ARG, "sh_fwd: null pointer");
    MS_REQUIRE(color_dtype == MS_COLOR_F32 || color_dtype == MS_COLOR_F16, MS_ERR_INVALID_ARG,
               "sh_fwd: bad colour dtype %d", color_dtype);
    MS_REQUIRE(((uintptr_t)radii & 7) == 0, MS_ERR_INVALID_ARG, "sh_fwd: radii must be 8-byte aligned");
    const ShArgs A{N, K, cam_x, cam_y, cam_z, add_half_and_clamp};
    hipStream_t stream = (hipStream_t)stream_;
    switch (degree) {
        case 0: return launch_fwd<0>(A, means3d, coeffs, radii, colors, color_dtype, stream);
        case 1: return launch_fwd<1>(A, means3d, coeffs, radii, colors, color_dtype, stream);
        case 2: return launch_fwd<2>(A, means3d, coeffs, radii, colors, color_dtype, stream);
        case 3: return launch_fwd<3>(A, means3d, coeffs, radii, colors, color_dtype, stream);
        default: return launch_fwd<4>(A, means3d, coeffs, radii, colors, color_dtype, stream);
    }
}

extern "C" int ms_spherical_harmonics_bwd(int64_t N, int K, int degree, const float *means3d, float cam_x,
                                          float cam_y, float cam_z, const float *coeffs, const int32_t *radii,
                                          int add_half_and_clamp, const float *colors_fwd, const float *v_colors,
                                          float *v_coeffs, float *v_means3d, void *stream_) {
    if (int rc = check_sh(N, K, degree, "sh_bwd")) return rc;
    if (N == 0) return MS_OK;
    MS_REQUIRE(means3d && v_colors && (v_coeffs || v_means3d), MS_ERR_INVALID_ARG, "sh_bwd: null pointer");
    MS_REQUIRE(!v_means3d || coeffs, MS_ERR_INVALID_ARG, "sh_bwd: v_means3d needs the coefficients");
    MS_REQUIRE(!add_half_and_clamp || colors_fwd, MS_ERR_INVALID_ARG,
               "sh_bwd: the clamped form needs the forward colours (f32)");
    MS_REQUIRE(((uintptr_t)radii & 7) == 0, MS_ERR_INVALID_ARG, "sh_bwd: radii must be 8-byte aligned");
    const ShArgs A{N, K, cam_x, cam_y, cam_z, add_half_and_clamp};
    hipStream_t stream = (hipStream_t)stream_;
    switch (degree) {
        case 0: return launch_bwd<0>(A, means3d, coeffs, radii, colors_fwd, v_colors, v_coeffs, v_means3d, stream);
        case 1: return launch_bwd<1>(A, means3d, coeffs, radii, colors_fwd, v_colors, v_coeffs, v_means3d, stream);
        case 2: return launch_bwd<2>(A, means3d, coeffs, radii, colors_fwd, v_colors, v_coeffs, v_means3d, stream);
        case 3: return launch_bwd<3>(A, means3d, coeffs, radii, colors_fwd, v_colors, v_coeffs, v_means3d, stream);
        default: return launch_bwd<4>(A, means3d, coeffs, radii, colors_fwd, v_colors, v_coeffs, v_means3d, stream);
    }
}
