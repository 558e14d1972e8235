// Backward of the EWA projection (project.hip) for gfx950: one lane per Gaussian.
//
// No reference counterpart (the reference renderer is forward-only, mojosplat/render.py:11);
// the chain rule is applied to the forward stated in mojosplat/kernels/projection.mojo:50-257:
//   (v_means2d, v_conics, v_depths) -> v_means3d, v_scales (w.r.t. the log-scales when the
//   forward took log-scales), v_quats (w.r.t. the un-normalised quaternion).
// Culled Gaussians (radii == 0) get zero gradients.  Recomputes the forward intermediates in
// registers instead of storing them: 44 B in + 24 B of upstream grads + 40 B out per Gaussian.
#include "ms_common.hpp"

namespace {

struct ProjBwdParams {
    float fx, fy, cx, cy, eps2d;
    float lim_x_pos, lim_x_neg, lim_y_pos, lim_y_neg;
    int scales_are_log;
};

// ROWS: the upstream gradients come as the backward rasteriser's packed 64-byte rows (rasterize_bwd.hip: mx my ca cb cc
// op c0 c1 c2 c3 ...) instead of v_means2d / v_conics; the kernel then also writes v_opacities and v_colors from the row
// (what k_unpack_grads would have done in a pass of its own).
// ROWS == 2: the rows of the quad-wave rasteriser (rasterize_bwdq.hip) -- RAW sums over the Gaussian's pixels,
//   gx gy s1 s2 s3 m0 c0 c1 c2  =  sum vs dx, sum vs dy, sum vs dx^2, sum vs dx dy, sum vs dy^2, sum vs, colour gradients
// (vs = dL/dsigma of a pixel-Gaussian pair, dx = mean - pixel): with the conic (a, b, c) this kernel recomputes anyway,
// v_mean = (a gx + b gy, b gx + c gy), v_conic = (s1 / 2, s2, s3 / 2), v_opacity = -m0 / opacity.  A Gaussian whose row is
// all zero was never blended (or culled): zero gradients, no radii needed.
template <int ROWS>
__global__ __launch_bounds__(256) void k_project_ewa_bwd(
    int64_t N, const float *__restrict__ means3d, const float *__restrict__ scales,
    const float *__restrict__ quats, const float *__restrict__ viewmat, ProjBwdParams P,
    const int32_t *__restrict__ radii, const float *__restrict__ v_means2d,
    const float *__restrict__ v_conics, const float *__restrict__ v_depths,
    float *__restrict__ v_means3d, float *__restrict__ v_scales, float *__restrict__ v_quats,
    const float *__restrict__ rows, int cdim, float *__restrict__ v_colors, float *__restrict__ v_opacities,
    const float *__restrict__ opacities) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float4 row0 = make_float4(0.f, 0.f, 0.f, 0.f), row1 = row0;
    bool visible;
    if constexpr (ROWS == 1) {
        const float4 *row = reinterpret_cast<const float4 *>(rows + i * 16);
        row0 = row[0]; row1 = row[1];
        const float4 row2 = row[2];
        v_opacities[i] = row1.y;
        const float c[4] = {row1.z, row1.w, row2.x, row2.y};
        for (int k = 0; k < cdim && k < 4; ++k) v_colors[i * cdim + k] = c[k];
    }
    if constexpr (ROWS == 2) {
        const float4 *row = reinterpret_cast<const float4 *>(rows + i * 16);
        row0 = row[0]; row1 = row[1];
        const float c2 = rows[i * 16 + 8];
        const float op = opacities[i];
        v_opacities[i] = row1.y != 0.f ? -row1.y / op : 0.f;
        v_colors[i * 3] = row1.z; v_colors[i * 3 + 1] = row1.w; v_colors[i * 3 + 2] = c2;
        visible = row0.x != 0.f || row0.y != 0.f || row0.z != 0.f || row0.w != 0.f || row1.x != 0.f || row1.y != 0.f;
    } else {
        const int2 rad = reinterpret_cast<const int2 *>(radii)[i];
        visible = rad.x > 0 && rad.y > 0;
    }
    float o_p[3] = {0.f, 0.f, 0.f}, o_s[3] = {0.f, 0.f, 0.f}, o_q[4] = {0.f, 0.f, 0.f, 0.f};
    if (visible) {
        float V[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) V[k] = viewmat[k];
        // ---- recompute forward -----------------------------------------------------------
        const float p0 = means3d[3 * i], p1 = means3d[3 * i + 1], p2 = means3d[3 * i + 2];
        const float x = V[0] * p0 + V[1] * p1 + V[2] * p2 + V[3];
        const float y = V[4] * p0 + V[5] * p1 + V[6] * p2 + V[7];
        const float z = V[8] * p0 + V[9] * p1 + V[10] * p2 + V[11];
        const float4 q4 = reinterpret_cast<const float4 *>(quats)[i];
        const float qn2 = q4.x * q4.x + q4.y * q4.y + q4.z * q4.z + q4.w * q4.w;
        const float inv_norm = 1.0f / sqrtf(qn2);
        const float w = q4.x * inv_norm, qx = q4.y * inv_norm, qy = q4.z * inv_norm, qz = q4.w * inv_norm;
        float R[3][3];
        R[0][0] = 1.f - 2.f * (qy * qy + qz * qz); R[0][1] = 2.f * (qx * qy - w * qz); R[0][2] = 2.f * (qx * qz + w * qy);
        R[1][0] = 2.f * (qx * qy + w * qz); R[1][1] = 1.f - 2.f * (qx * qx + qz * qz); R[1][2] = 2.f * (qy * qz - w * qx);
        R[2][0] = 2.f * (qx * qz - w * qy); R[2][1] = 2.f * (qy * qz + w * qx); R[2][2] = 1.f - 2.f * (qx * qx + qy * qy);
        float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        if (P.scales_are_log) { s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]); }
        float Mx[3][3], cov[3][3], tmp[3][3], cc[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Mx[r][c] = R[r][c] * s[c];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) cov[r][c] = Mx[r][0] * Mx[c][0] + Mx[r][1] * Mx[c][1] + Mx[r][2] * Mx[c][2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) tmp[r][c] = V[4 * r] * cov[0][c] + V[4 * r + 1] * cov[1][c] + V[4 * r + 2] * cov[2][c];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) cc[r][c] = tmp[r][0] * V[4 * c] + tmp[r][1] * V[4 * c + 1] + tmp[r][2] * V[4 * c + 2];
        const float rz = 1.0f / z, rz2 = rz * rz, rz3 = rz2 * rz;
        const float xr = x * rz, yr = y * rz;
        const bool x_free = xr <= P.lim_x_pos && xr >= -P.lim_x_neg;
        const bool y_free = yr <= P.lim_y_pos && yr >= -P.lim_y_neg;
        const float tx = z * fminf(P.lim_x_pos, fmaxf(-P.lim_x_neg, xr));
        const float ty = z * fminf(P.lim_y_pos, fmaxf(-P.lim_y_neg, yr));
        const float J00 = P.fx * rz, J02 = -P.fx * tx * rz2, J11 = P.fy * rz, J12 = -P.fy * ty * rz2;
        float JC[2][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            JC[0][c] = J00 * cc[0][c] + J02 * cc[2][c];
            JC[1][c] = J11 * cc[1][c] + J12 * cc[2][c];
        }
        const float a = JC[0][0] * J00 + JC[0][2] * J02 + P.eps2d;
        const float b = JC[0][1] * J11 + JC[0][2] * J12;
        const float c = JC[1][1] * J11 + JC[1][2] * J12 + P.eps2d;
        const float inv_det = 1.0f / (a * c - b * b);
        const float ka = c * inv_det, kb = -b * inv_det, kc = a * inv_det;  // conic

        // ---- backward --------------------------------------------------------------------
        float vm0, vm1, vka, vkb, vkc;
        if constexpr (ROWS == 2) {
            vm0 = ka * row0.x + kb * row0.y;
            vm1 = kb * row0.x + kc * row0.y;
            vka = 0.5f * row0.z; vkb = 0.5f * row0.w; vkc = 0.5f * row1.x;
        } else {
            vm0 = ROWS ? row0.x : v_means2d[2 * i]; vm1 = ROWS ? row0.y : v_means2d[2 * i + 1];
            vka = ROWS ? row0.z : v_conics[3 * i]; vkb = (ROWS ? row0.w : v_conics[3 * i + 1]) * 0.5f;
            vkc = ROWS ? row1.x : v_conics[3 * i + 2];
        }
        const float vd = v_depths ? v_depths[i] : 0.f;
        // conic = inverse(cov2d): v_cov2d = -K vK K  (K symmetric; off-diagonal grad halved)
        const float t00 = ka * vka + kb * vkb, t01 = ka * vkb + kb * vkc;
        const float t10 = kb * vka + kc * vkb, t11 = kb * vkb + kc * vkc;
        const float g00 = -(t00 * ka + t01 * kb), g01 = -(t00 * kb + t01 * kc);
        const float g10 = -(t10 * ka + t11 * kb), g11 = -(t10 * kb + t11 * kc);
        // cov2d = J cc J^T :  v_cc = J^T G J ;  v_J = G J cc^T + G^T J cc
        const float Jm[2][3] = {{J00, 0.f, J02}, {0.f, J11, J12}};
        const float G[2][2] = {{g00, g01}, {g10, g11}};
        float v_cc[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cidx = 0; cidx < 3; ++cidx)
                v_cc[r][cidx] = Jm[0][r] * (G[0][0] * Jm[0][cidx] + G[0][1] * Jm[1][cidx]) +
                                Jm[1][r] * (G[1][0] * Jm[0][cidx] + G[1][1] * Jm[1][cidx]);
        // JCt[r][c] = (J cc^T)[r][c] ; cc is symmetric up to rounding, keep both forms
        float v_J[2][3];
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) {
            const float jct0 = Jm[0][0] * cc[cidx][0] + Jm[0][1] * cc[cidx][1] + Jm[0][2] * cc[cidx][2];
            const float jct1 = Jm[1][0] * cc[cidx][0] + Jm[1][1] * cc[cidx][1] + Jm[1][2] * cc[cidx][2];
            v_J[0][cidx] = G[0][0] * jct0 + G[0][1] * jct1 + G[0][0] * JC[0][cidx] + G[1][0] * JC[1][cidx];
            v_J[1][cidx] = G[1][0] * jct0 + G[1][1] * jct1 + G[0][1] * JC[0][cidx] + G[1][1] * JC[1][cidx];
        }
        // camera-space mean
        float v_x = P.fx * rz * vm0, v_y = P.fy * rz * vm1;
        float v_z = -(P.fx * x * vm0 + P.fy * y * vm1) * rz2 + vd;
        v_z += -P.fx * rz2 * v_J[0][0] - P.fy * rz2 * v_J[1][1];
        if (x_free) { v_x += -P.fx * rz2 * v_J[0][2]; v_z += 2.f * P.fx * tx * rz3 * v_J[0][2]; }
        else        { v_z += P.fx * tx * rz3 * v_J[0][2]; }
        if (y_free) { v_y += -P.fy * rz2 * v_J[1][2]; v_z += 2.f * P.fy * ty * rz3 * v_J[1][2]; }
        else        { v_z += P.fy * ty * rz3 * v_J[1][2]; }
        // world mean: p = Wv^T v_mean_c
        o_p[0] = V[0] * v_x + V[4] * v_y + V[8] * v_z;
        o_p[1] = V[1] * v_x + V[5] * v_y + V[9] * v_z;
        o_p[2] = V[2] * v_x + V[6] * v_y + V[10] * v_z;
        // world covariance: v_cov = Wv^T v_cc Wv
        float t2[3][3], v_cov[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cidx = 0; cidx < 3; ++cidx)
                t2[r][cidx] = V[r] * v_cc[0][cidx] + V[4 + r] * v_cc[1][cidx] + V[8 + r] * v_cc[2][cidx];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cidx = 0; cidx < 3; ++cidx)
                v_cov[r][cidx] = t2[r][0] * V[cidx] + t2[r][1] * V[4 + cidx] + t2[r][2] * V[8 + cidx];
        // cov = M M^T : v_M = (v_cov + v_cov^T) M ; M = R diag(s)
        float v_M[3][3], v_R[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cidx = 0; cidx < 3; ++cidx)
                v_M[r][cidx] = (v_cov[r][0] + v_cov[0][r]) * Mx[0][cidx] + (v_cov[r][1] + v_cov[1][r]) * Mx[1][cidx] +
                               (v_cov[r][2] + v_cov[2][r]) * Mx[2][cidx];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float vs = R[0][k] * v_M[0][k] + R[1][k] * v_M[1][k] + R[2][k] * v_M[2][k];
            o_s[k] = P.scales_are_log ? vs * s[k] : vs;
#pragma unroll
            for (int r = 0; r < 3; ++r) v_R[r][k] = v_M[r][k] * s[k];
        }
        // rotation -> normalised quaternion -> raw quaternion
        const float vw = 2.f * (qz * (v_R[1][0] - v_R[0][1]) + qy * (v_R[0][2] - v_R[2][0]) + qx * (v_R[2][1] - v_R[1][2]));
        const float vx = 2.f * (qy * (v_R[0][1] + v_R[1][0]) + qz * (v_R[0][2] + v_R[2][0]) + w * (v_R[2][1] - v_R[1][2])) -
                         4.f * qx * (v_R[1][1] + v_R[2][2]);
        const float vy = 2.f * (qx * (v_R[0][1] + v_R[1][0]) + w * (v_R[0][2] - v_R[2][0]) + qz * (v_R[1][2] + v_R[2][1])) -
                         4.f * qy * (v_R[0][0] + v_R[2][2]);
        const float vz = 2.f * (w * (v_R[1][0] - v_R[0][1]) + qx * (v_R[0][2] + v_R[2][0]) + qy * (v_R[1][2] + v_R[2][1])) -
                         4.f * qz * (v_R[0][0] + v_R[1][1]);
        const float dotn = vw * w + vx * qx + vy * qy + vz * qz;
        o_q[0] = (vw - dotn * w) * inv_norm;
        o_q[1] = (vx - dotn * qx) * inv_norm;
        o_q[2] = (vy - dotn * qy) * inv_norm;
        o_q[3] = (vz - dotn * qz) * inv_norm;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v_means3d[3 * i + k] = o_p[k];
        v_scales[3 * i + k] = o_s[k];
    }
    reinterpret_cast<float4 *>(v_quats)[i] = make_float4(o_q[0], o_q[1], o_q[2], o_q[3]);
}

}  // namespace

extern "C" int ms_project_gaussians_bwd(int64_t N, const float *means3d, const float *scales,
                                        int scales_are_log, const float *quats, const float *viewmat,
                                        float fx, float fy, float cx, float cy, int W, int H,
                                        float eps2d, const int32_t *radii, const float *v_means2d,
                                        const float *v_conics, const float *v_depths,
                                        float *v_means3d, float *v_scales, float *v_quats,
                                        void *stream) {
    MS_REQUIRE(N >= 0, MS_ERR_INVALID_ARG, "project_bwd: N < 0");
    if (N == 0) return MS_OK;
    MS_REQUIRE(means3d && scales && quats && viewmat && radii && v_means2d && v_conics && v_means3d &&
                   v_scales && v_quats, MS_ERR_INVALID_ARG, "project_bwd: null pointer");
    MS_REQUIRE(W > 0 && H > 0 && fx != 0.f && fy != 0.f, MS_ERR_INVALID_ARG, "project_bwd: bad camera");
    MS_REQUIRE(((uintptr_t)quats & 15) == 0 && ((uintptr_t)v_quats & 15) == 0 && ((uintptr_t)radii & 7) == 0,
               MS_ERR_INVALID_ARG, "project_bwd: quats/v_quats must be 16-byte, radii 8-byte aligned");
    ProjBwdParams P;
    P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.eps2d = eps2d;
    const float tan_fovx = 0.5f * (float)W / fx, tan_fovy = 0.5f * (float)H / fy;
    P.lim_x_pos = ((float)W - cx) / fx + 0.3f * tan_fovx;
    P.lim_x_neg = cx / fx + 0.3f * tan_fovx;
    P.lim_y_pos = ((float)H - cy) / fy + 0.3f * tan_fovy;
    P.lim_y_neg = cy / fy + 0.3f * tan_fovy;
    P.scales_are_log = scales_are_log;
    const int64_t grid = ms::ceil_div(N, 256);
    MS_REQUIRE(grid <= 0x7fffffff, MS_ERR_INVALID_ARG, "project_bwd: N too large");
    hipLaunchKernelGGL(k_project_ewa_bwd<0>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, N, means3d,
                       scales, quats, viewmat, P, radii, v_means2d, v_conics, v_depths, v_means3d, v_scales,
                       v_quats, nullptr, 0, nullptr, nullptr, nullptr);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// ms_render_bwd: the same backward straight from the backward rasteriser's packed rows (which it also unpacks into
// v_colors / v_opacities)
int ms::project_bwd_from_rows(int64_t N, const float *means3d, const float *scales, int scales_are_log, const float *quats,
                              const float *viewmat, float fx, float fy, float cx, float cy, int W, int H, float eps2d,
                              const int32_t *radii, const float *rows, int CDIM, float *v_means3d, float *v_scales,
                              float *v_quats, float *v_colors, float *v_opacities, void *stream, const float *raw_rows_opacities) {
    if (N == 0) return MS_OK;
    const bool raw = raw_rows_opacities != nullptr;   // the quad-wave rasteriser's raw sums (ROWS == 2)
    MS_REQUIRE(means3d && scales && quats && viewmat && (radii || raw) && rows && v_means3d && v_scales && v_quats && v_colors && v_opacities,
               MS_ERR_INVALID_ARG, "project_bwd: null pointer");
    MS_REQUIRE(!raw || CDIM == 3, MS_ERR_INVALID_ARG, "project_bwd: raw rows carry three channels");
    MS_REQUIRE(W > 0 && H > 0 && fx != 0.f && fy != 0.f && CDIM >= 1 && CDIM <= 4, MS_ERR_INVALID_ARG, "project_bwd: bad camera / channels");
    MS_REQUIRE(((uintptr_t)quats & 15) == 0 && ((uintptr_t)v_quats & 15) == 0 && ((uintptr_t)radii & 7) == 0 && ((uintptr_t)rows & 15) == 0,
               MS_ERR_INVALID_ARG, "project_bwd: quats / v_quats / rows must be 16-byte, radii 8-byte aligned");
    ProjBwdParams P;
    P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.eps2d = eps2d;
    const float tan_fovx = 0.5f * (float)W / fx, tan_fovy = 0.5f * (float)H / fy;
    P.lim_x_pos = ((float)W - cx) / fx + 0.3f * tan_fovx;
    P.lim_x_neg = cx / fx + 0.3f * tan_fovx;
    P.lim_y_pos = ((float)H - cy) / fy + 0.3f * tan_fovy;
    P.lim_y_neg = cy / fy + 0.3f * tan_fovy;
    P.scales_are_log = scales_are_log;
    const int64_t grid = ms::ceil_div(N, 256);
    MS_REQUIRE(grid <= 0x7fffffff, MS_ERR_INVALID_ARG, "project_bwd: N too large");
    if (raw)
        hipLaunchKernelGGL(k_project_ewa_bwd<2>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, N, means3d,
                           scales, quats, viewmat, P, radii, nullptr, nullptr, nullptr, v_means3d, v_scales, v_quats, rows, CDIM,
                           v_colors, v_opacities, raw_rows_opacities);
    else
        hipLaunchKernelGGL(k_project_ewa_bwd<1>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, N, means3d,
                           scales, quats, viewmat, P, radii, nullptr, nullptr, nullptr, v_means3d, v_scales, v_quats, rows, CDIM,
                           v_colors, v_opacities, nullptr);
    MS_LAUNCH_CHECK();
    return MS_OK;
}
