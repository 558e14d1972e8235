// EWA projection of ONE Gaussian, shared by the projection kernel (project.hip) and the fused
// projection + tile-count kernel (binning.hip).  Semantics: see project.hip.
#pragma once
#include "ms_common.hpp"

namespace ms {

struct ProjParams {
    float fx, fy, cx, cy;
    float W, H;
    float eps2d, near_plane, far_plane, radius_clip;
    float lim_x_pos, lim_x_neg, lim_y_pos, lim_y_neg;
    int scales_are_log;
    int has_opacity;
};


struct ProjOut {
    float m0, m1, c0, c1, c2, d;
    int r0, r1;
};

static inline ProjParams make_proj_params(float fx, float fy, float cx, float cy, int W, int H, float eps2d,
                                          float near_plane, float far_plane, float radius_clip,
                                          int scales_are_log, bool has_opacity) {
    ProjParams P;
    P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy;
    P.W = (float)W; P.H = (float)H;
    P.eps2d = eps2d; P.near_plane = near_plane; P.far_plane = far_plane; P.radius_clip = radius_clip;
    const float tan_fovx = 0.5f * (float)W / fx, tan_fovy = 0.5f * (float)H / fy;
    P.lim_x_pos = ((float)W - cx) / fx + 0.3f * tan_fovx;
    P.lim_x_neg = cx / fx + 0.3f * tan_fovx;
    P.lim_y_pos = ((float)H - cy) / fy + 0.3f * tan_fovy;
    P.lim_y_neg = cy / fy + 0.3f * tan_fovy;
    P.scales_are_log = scales_are_log;
    P.has_opacity = has_opacity;
    return P;
}

// Strict fp32 evaluation order (no FMA contraction) so that the only differences against the CPU
// oracle come from expf/logf; radii are integers and flip on 1-ulp changes.
__device__ __forceinline__ ProjOut project_one(int64_t i, const float *__restrict__ means3d,
                                               const float *__restrict__ scales, const float *__restrict__ quats,
                                               const float *__restrict__ opacities,
                                               const float *__restrict__ viewmat, const ProjParams &P) {
#pragma clang fp contract(off)
    float V[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) V[k] = viewmat[k];  // uniform -> s_load

    const float p0 = means3d[3 * i], p1 = means3d[3 * i + 1], p2 = means3d[3 * i + 2];
    const float mx = V[0] * p0 + V[1] * p1 + V[2] * p2 + V[3];
    const float my = V[4] * p0 + V[5] * p1 + V[6] * p2 + V[7];
    const float z = V[8] * p0 + V[9] * p1 + V[10] * p2 + V[11];

    float o_m0 = 0.f, o_m1 = 0.f, o_c0 = 0.f, o_c1 = 0.f, o_c2 = 0.f, o_d = 0.f;
    int o_r0 = 0, o_r1 = 0;

    bool alive = !(z < P.near_plane || z > P.far_plane);
    if (alive) {
        const float4 q4 = reinterpret_cast<const float4 *>(quats)[i];
        float w = q4.x, x = q4.y, y = q4.z, zq = q4.w;
        const float inv_norm = 1.0f / sqrtf(x * x + y * y + zq * zq + w * w);
        w *= inv_norm; x *= inv_norm; y *= inv_norm; zq *= inv_norm;
        const float x2 = x * x, y2 = y * y, z2 = zq * zq;
        const float xy = x * y, xz = x * zq, yz = y * zq, wx = w * x, wy = w * y, wz = w * zq;
        const float R00 = 1.f - 2.f * (y2 + z2), R01 = 2.f * (xy - wz), R02 = 2.f * (xz + wy);
        const float R10 = 2.f * (xy + wz), R11 = 1.f - 2.f * (x2 + z2), R12 = 2.f * (yz - wx);
        const float R20 = 2.f * (xz - wy), R21 = 2.f * (yz + wx), R22 = 1.f - 2.f * (x2 + y2);

        float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
        if (P.scales_are_log) { s0 = expf(s0); s1 = expf(s1); s2 = expf(s2); }

        // M = R diag(s); cov = M M^T
        const float M00 = R00 * s0, M01 = R01 * s1, M02 = R02 * s2;
        const float M10 = R10 * s0, M11 = R11 * s1, M12 = R12 * s2;
        const float M20 = R20 * s0, M21 = R21 * s1, M22 = R22 * s2;
        float cov[3][3];
        cov[0][0] = M00 * M00 + M01 * M01 + M02 * M02;
        cov[0][1] = M00 * M10 + M01 * M11 + M02 * M12;
        cov[0][2] = M00 * M20 + M01 * M21 + M02 * M22;
        cov[1][0] = M10 * M00 + M11 * M01 + M12 * M02;
        cov[1][1] = M10 * M10 + M11 * M11 + M12 * M12;
        cov[1][2] = M10 * M20 + M11 * M21 + M12 * M22;
        cov[2][0] = M20 * M00 + M21 * M01 + M22 * M02;
        cov[2][1] = M20 * M10 + M21 * M11 + M22 * M12;
        cov[2][2] = M20 * M20 + M21 * M21 + M22 * M22;

        // camera-space covariance Wv cov Wv^T
        float tmp[3][3], cc[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                tmp[r][c] = V[4 * r + 0] * cov[0][c] + V[4 * r + 1] * cov[1][c] + V[4 * r + 2] * cov[2][c];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                cc[r][c] = tmp[r][0] * V[4 * c + 0] + tmp[r][1] * V[4 * c + 1] + tmp[r][2] * V[4 * c + 2];

        // pinhole Jacobian with the 1.3x FOV clamp
        const float rz = 1.0f / z, rz2 = rz * rz;
        const float tx = z * fminf(P.lim_x_pos, fmaxf(-P.lim_x_neg, mx * rz));
        const float ty = z * fminf(P.lim_y_pos, fmaxf(-P.lim_y_neg, my * rz));
        const float J00 = P.fx * rz, J02 = -P.fx * tx * rz2;
        const float J11 = P.fy * rz, J12 = -P.fy * ty * rz2;
        // JC = J cc (2x3) with J01 = J10 = 0 (adding the exact zero products changes nothing)
        float JC[2][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            JC[0][c] = J00 * cc[0][c] + J02 * cc[2][c];
            JC[1][c] = J11 * cc[1][c] + J12 * cc[2][c];
        }
        float a = JC[0][0] * J00 + JC[0][2] * J02;
        const float b01 = JC[0][1] * J11 + JC[0][2] * J12;
        const float b10 = JC[1][0] * J00 + JC[1][2] * J02;
        float c = JC[1][1] * J11 + JC[1][2] * J12;
        const float m2x = P.fx * mx * rz + P.cx, m2y = P.fy * my * rz + P.cy;

        a += P.eps2d;
        c += P.eps2d;
        const float det = a * c - b01 * b10;
        alive = det > 0.f;

        float extend = 3.33f;
        if (alive && P.has_opacity) {
            const float op = opacities[i];
            if (op < kAlphaThreshold) {
                alive = false;
            } else {
                extend = fminf(extend, sqrtf(2.0f * logf(op / kAlphaThreshold)));
            }
        }
        if (alive) {
            const float rx = ceilf(extend * sqrtf(a)), ry = ceilf(extend * sqrtf(c));
            if (rx <= P.radius_clip && ry <= P.radius_clip) alive = false;
            if (m2x + rx <= 0.f || m2x - rx >= P.W || m2y + ry <= 0.f || m2y - ry >= P.H) alive = false;
            if (alive) {
                const float inv_det = 1.0f / det;
                o_c0 = c * inv_det;
                o_c1 = -b01 * inv_det;
                o_c2 = a * inv_det;
                o_r0 = (int)rx;
                o_r1 = (int)ry;
                o_m0 = m2x;
                o_m1 = m2y;
                o_d = z;
            }
        }
    }
    return ProjOut{o_m0, o_m1, o_c0, o_c1, o_c2, o_d, o_r0, o_r1};
}

}  // namespace ms
