// EWA 3D->2D projection for gfx950: one lane per Gaussian, everything in registers.
//
// Semantics = gsplat's fused projection as the reference restates it in
// mojosplat/kernels/projection.mojo:50-257 (cull order near/far -> det -> opacity -> radius
// clip -> viewport; opacity-aware extent; culled rows report zeros), with the camera's own
// near/far planes (the reference's gsplat call passes them, projection.py:391-392).
// Memory-bound: 44 B in + 32 B out per Gaussian.  The view matrix is wave-uniform and is read
// from HBM through the scalar cache; intrinsics arrive as kernel arguments (SGPRs).
#include "project_device.hpp"

namespace {

__global__ __launch_bounds__(256) void k_project_ewa_fwd(
    int64_t N, const float *__restrict__ means3d, const float *__restrict__ scales,
    const float *__restrict__ quats, const float *__restrict__ opacities,
    const float *__restrict__ viewmat, ms::ProjParams P, float *__restrict__ means2d,
    float *__restrict__ conics, float *__restrict__ depths, int32_t *__restrict__ radii) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const ms::ProjOut o = ms::project_one(i, means3d, scales, quats, opacities, viewmat, P);
    reinterpret_cast<float2 *>(means2d)[i] = make_float2(o.m0, o.m1);
    conics[3 * i] = o.c0;
    conics[3 * i + 1] = o.c1;
    conics[3 * i + 2] = o.c2;
    depths[i] = o.d;
    reinterpret_cast<int2 *>(radii)[i] = make_int2(o.r0, o.r1);
}

}  // namespace

extern "C" int ms_project_gaussians_fwd(int64_t N, const float *means3d, const float *scales,
                                        int scales_are_log, const float *quats,
                                        const float *opacities, const float *viewmat, float fx,
                                        float fy, float cx, float cy, int W, int H, float eps2d,
                                        float near_plane, float far_plane, float radius_clip,
                                        float *means2d, float *conics, float *depths,
                                        int32_t *radii, void *stream) {
    MS_REQUIRE(N >= 0, MS_ERR_INVALID_ARG, "project: N < 0");
    if (N == 0) return MS_OK;
    MS_REQUIRE(means3d && scales && quats && viewmat && means2d && conics && depths && radii,
               MS_ERR_INVALID_ARG, "project: null pointer");
    MS_REQUIRE(W > 0 && H > 0 && fx != 0.f && fy != 0.f, MS_ERR_INVALID_ARG,
               "project: bad camera (W=%d H=%d fx=%g fy=%g)", W, H, fx, fy);
    MS_REQUIRE(((uintptr_t)quats & 15) == 0 && ((uintptr_t)means2d & 7) == 0 && ((uintptr_t)radii & 7) == 0,
               MS_ERR_INVALID_ARG, "project: quats must be 16-byte, means2d/radii 8-byte aligned");
    const ms::ProjParams P = ms::make_proj_params(fx, fy, cx, cy, W, H, eps2d, near_plane, far_plane, radius_clip,
                                                  scales_are_log, opacities != nullptr);
    const int64_t grid = ms::ceil_div(N, 256);
    MS_REQUIRE(grid <= 0x7fffffff, MS_ERR_INVALID_ARG, "project: N too large");
    hipLaunchKernelGGL(k_project_ewa_fwd, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, N,
                       means3d, scales, quats, opacities, viewmat, P, means2d, conics, depths, radii);
    MS_LAUNCH_CHECK();
    return MS_OK;
}
