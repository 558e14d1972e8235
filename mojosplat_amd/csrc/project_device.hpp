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

// The projection of one Gaussian.  Round 3: the chain is written for the instruction count -- the fused projection +
// count kernel was VALU bound in round 2 (1 200 instructions per Gaussian, a third of them this chain):
//   * cov2d = (J W M)(J W M)^T with M = R diag(s): the 2x3 matrix T = J W costs 12 operations, B = T R 18,
//     A = B diag(s) 6 and the three entries of A A^T 9 -- 45 multiply-adds where the reference's order
//     (cov3d = M M^T, W cov3d W^T, J . J^T; projection.mojo:129-198) takes ~160; same quantity, rounded differently
//     at the 1e-7 level;
//   * FMA contraction on; v_rsq_f32 / v_rcp_f32 + one Newton step / v_sqrt_f32 / v_log_f32 instead of the IEEE
//     division and square-root sequences (~10 instructions each) -- on the COVARIANCE side only.
// What stays in the reference's operation order, uncontracted and IEEE: the camera-space mean, 1 / z and means2d.
// A first version that let those take part (v_rcp for 1 / z, contracted products) left 27 pixels of the config-3
// frame beyond 1e-4 of the oracle's with no branch of its walk within 2e-5 of a threshold: an ulp of a mean at
// x ~ 1000 px is 6e-5 px, which moves sigma by 3e-5 for a 3-px Gaussian evaluated 5 px out -- more than the margin
// the parity bar grants -- while an ulp of the conic moves it by 5e-7.  So means2d and the depth are bit for bit the
// oracle's, the conics agree to ~1e-7 relative, and the integer radii agree except where extent * sqrt(cov) lands
// within a few ulp of an integer (a counted handful per 100k Gaussians, as before: libm's expf / logf already
// differed from the GPU's by an ulp).  (Round 2's strict chain, kept behind a build flag through round 3, is gone: git history.)
template <class Idx>
__device__ __forceinline__ float ld_f32(const float *base, Idx i, int stride, int k) {
    // Idx = uint32_t: byte offsets formed in 32 bits against uniform base pointers (saddr + voffset addressing)
    if constexpr (sizeof(Idx) == 4)
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (uint32_t)(4u * ((uint32_t)stride * (uint32_t)i + (uint32_t)k)));
    else
        return base[(int64_t)stride * i + k];
}
template <class Idx>
__device__ __forceinline__ float4 ld_f32x4(const float *base, Idx i) {
    if constexpr (sizeof(Idx) == 4)
        return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + (uint32_t)(16u * (uint32_t)i));
    else
        return reinterpret_cast<const float4 *>(base)[i];
}

// three consecutive floats of an AoS row as ONE 12-byte load (global_load_dwordx3; 4-byte alignment suffices)
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
template <class Idx>
__device__ __forceinline__ F3 ld_f32x3(const float *base, Idx i) {
    if constexpr (sizeof(Idx) == 4)
        return *reinterpret_cast<const F3 *>(reinterpret_cast<const char *>(base) + (uint32_t)(12u * (uint32_t)i));
    else
        return reinterpret_cast<const F3 *>(base)[i];
}

__device__ __forceinline__ float rcp_nr(float x) {   // 1 / x to ~0.5 ulp: v_rcp_f32 + one Newton step
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}

// FLAT (-DMS_PROJECT_FLAT=1, a measurement of round 5, off by default): one verdict at the end instead of the nested exits.
// Config 4's frame gains 2.6-3.8 % with it (0.227 -> 0.219 ms: its depth-cut count kernel), config 3's loses 0.5 % (count
// kernel 27.85 -> 28.35 us, rasteriser +0.5 on lists that differ by a few pairs).  The two forms are NOT interchangeable
// within a build: the contraction of their products differs in the last bit of a conic (a depth-cut frame projected one
// way and its uncut twin the other differed by 2e-3 at a pixel), so every user of project_one takes the build's one form.
#ifndef MS_PROJECT_FLAT
#define MS_PROJECT_FLAT 0
#endif
template <class Idx, bool LOADS_FIRST = false, bool FLAT = (MS_PROJECT_FLAT != 0)>
__device__ __forceinline__ ProjOut project_one(Idx i, const float *__restrict__ means3d,
                                               const float *__restrict__ scales, const float *__restrict__ quats,
                                               const float *__restrict__ opacities,
                                               const float *__restrict__ viewmat, const ProjParams &P) {
#pragma clang fp contract(fast)
    float V[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) V[k] = viewmat[k];  // uniform -> s_load

    const F3 p3 = ld_f32x3(means3d, i);
    const float p0 = p3.x, p1 = p3.y, p2 = p3.z;
    float mx, my, z;
    {   // the camera-space mean: the reference's operation order, uncontracted (see above: means2d and the depth)
#pragma clang fp contract(off)
        mx = V[0] * p0 + V[1] * p1 + V[2] * p2 + V[3];
        my = V[4] * p0 + V[5] * p1 + V[6] * p2 + V[7];
        z = V[8] * p0 + V[9] * p1 + V[10] * p2 + V[11];
    }

    if constexpr (FLAT) {
        // FLAT (round 5): ONE verdict at the end instead of the nested `if (alive)` blocks below.  The nest cost every wave its exits' bookkeeping
        // whether a lane took them or not -- the eight outputs zeroed again at each level (~40 v_mov), an exec-mask save / restore
        // per level -- and put the quaternion, scale and opacity loads BEHIND the tests before them: three dependent memory round
        // trips a slice.  A culled lane now computes on whatever its Gaussian holds (its loads are valid for any index below N;
        // infinities and NaNs it may produce are never selected) and the loads leave together with the mean's.
        const float4 q4 = ld_f32x4(quats, i);
        const F3 s3 = ld_f32x3(scales, i);
        float op = 1.0f;
        if (P.has_opacity) op = ld_f32(opacities, i, 1, 0);   // (uniform)
        bool alive = !(z < P.near_plane || z > P.far_plane);

        float w = q4.x, x = q4.y, y = q4.z, zq = q4.w;
        const float inv_norm = __builtin_amdgcn_rsqf(x * x + y * y + zq * zq + w * w);
        w *= inv_norm; x *= inv_norm; y *= inv_norm; zq *= inv_norm;
        const float x2 = x * x, y2 = y * y, z2 = zq * zq;
        const float xy = x * y, xz = x * zq, yz = y * zq, wx = w * x, wy = w * y, wz = w * zq;
        const float R00 = 1.f - 2.f * (y2 + z2), R01 = 2.f * (xy - wz), R02 = 2.f * (xz + wy);
        const float R10 = 2.f * (xy + wz), R11 = 1.f - 2.f * (x2 + z2), R12 = 2.f * (yz - wx);
        const float R20 = 2.f * (xz - wy), R21 = 2.f * (yz + wx), R22 = 1.f - 2.f * (x2 + y2);

        float s0 = s3.x, s1 = s3.y, s2 = s3.z;
        if (P.scales_are_log) { s0 = expf(s0); s1 = expf(s1); s2 = expf(s2); }

        // pinhole Jacobian with the 1.3x FOV clamp; T = J Wv (2x3)
        float rz, m2x, m2y;
        {   // means2d: IEEE 1 / z and the reference's products, uncontracted -- bit for bit the oracle's
            rz = 1.0f / z;
            m2x = P.fx * mx * rz + P.cx;
            m2y = P.fy * my * rz + P.cy;
        }
        const float tx = z * fminf(P.lim_x_pos, fmaxf(-P.lim_x_neg, mx * rz));
        const float ty = z * fminf(P.lim_y_pos, fmaxf(-P.lim_y_neg, my * rz));
        const float J00 = P.fx * rz, J02 = -(J00 * tx) * rz;
        const float J11 = P.fy * rz, J12 = -(J11 * ty) * rz;
        const float T00 = J00 * V[0] + J02 * V[8], T01 = J00 * V[1] + J02 * V[9], T02 = J00 * V[2] + J02 * V[10];
        const float T10 = J11 * V[4] + J12 * V[8], T11 = J11 * V[5] + J12 * V[9], T12 = J11 * V[6] + J12 * V[10];
        // A = T R diag(s)
        const float A00 = (T00 * R00 + T01 * R10 + T02 * R20) * s0, A01 = (T00 * R01 + T01 * R11 + T02 * R21) * s1,
                    A02 = (T00 * R02 + T01 * R12 + T02 * R22) * s2;
        const float A10 = (T10 * R00 + T11 * R10 + T12 * R20) * s0, A11 = (T10 * R01 + T11 * R11 + T12 * R21) * s1,
                    A12 = (T10 * R02 + T11 * R12 + T12 * R22) * s2;
        const float a = A00 * A00 + A01 * A01 + A02 * A02 + P.eps2d;
        const float b = A00 * A10 + A01 * A11 + A02 * A12;
        const float c = A10 * A10 + A11 * A11 + A12 * A12 + P.eps2d;

        const float det = a * c - b * b;
        alive = alive && det > 0.f;

        float extend = 3.33f;
        if (P.has_opacity) {   // (uniform)
            alive = alive && !(op < kAlphaThreshold);
            extend = fminf(extend, __builtin_amdgcn_sqrtf(2.0f * __logf(op * 255.0f)));   // (a NaN -- op below 1 / 255 -- leaves 3.33; the lane is dead)
        }
        const float rx = ceilf(extend * __builtin_amdgcn_sqrtf(a)), ry = ceilf(extend * __builtin_amdgcn_sqrtf(c));
        alive = alive && !(rx <= P.radius_clip && ry <= P.radius_clip);
        alive = alive && !(m2x + rx <= 0.f || m2x - rx >= P.W || m2y + ry <= 0.f || m2y - ry >= P.H);
        const float inv_det = rcp_nr(det);
        const float o_c0 = alive ? c * inv_det : 0.f, o_c1 = alive ? -b * inv_det : 0.f, o_c2 = alive ? a * inv_det : 0.f;
        const int o_r0 = alive ? (int)rx : 0, o_r1 = alive ? (int)ry : 0;
        const float o_m0 = alive ? m2x : 0.f, o_m1 = alive ? m2y : 0.f, o_d = alive ? z : 0.f;
        return ProjOut{o_m0, o_m1, o_c0, o_c1, o_c2, o_d, o_r0, o_r1};
    } else {
        float o_m0 = 0.f, o_m1 = 0.f, o_c0 = 0.f, o_c1 = 0.f, o_c2 = 0.f, o_d = 0.f;
        int o_r0 = 0, o_r1 = 0;

        bool alive = !(z < P.near_plane || z > P.far_plane);
        // LOADS_FIRST (the depth-cut count kernel, i.e. scenes of millions of Gaussians): the quaternion, the scales and the
        // opacity fetched WITH the mean instead of behind the tests before them -- valid for any index below N; the arithmetic
        // and its nesting are the same, and so are the bits (tests: a depth-cut frame equals its uncut twin).  Config 4's
        // frame 0.2265 -> 0.2215 ms; on config 3's plain kernel the same change costs 0.8 us, so that one keeps the
        // dependent loads.
        float4 q4 = make_float4(0.f, 0.f, 0.f, 1.f);
        F3 s3{0.f, 0.f, 0.f};
        float op_first = 1.0f;
        if constexpr (LOADS_FIRST) {
            q4 = ld_f32x4(quats, i);
            s3 = ld_f32x3(scales, i);
            if (P.has_opacity) op_first = ld_f32(opacities, i, 1, 0);
        }
        if (alive) {
            if constexpr (!LOADS_FIRST) q4 = ld_f32x4(quats, i);
            float w = q4.x, x = q4.y, y = q4.z, zq = q4.w;
            const float inv_norm = __builtin_amdgcn_rsqf(x * x + y * y + zq * zq + w * w);
            w *= inv_norm; x *= inv_norm; y *= inv_norm; zq *= inv_norm;
            const float x2 = x * x, y2 = y * y, z2 = zq * zq;
            const float xy = x * y, xz = x * zq, yz = y * zq, wx = w * x, wy = w * y, wz = w * zq;
            const float R00 = 1.f - 2.f * (y2 + z2), R01 = 2.f * (xy - wz), R02 = 2.f * (xz + wy);
            const float R10 = 2.f * (xy + wz), R11 = 1.f - 2.f * (x2 + z2), R12 = 2.f * (yz - wx);
            const float R20 = 2.f * (xz - wy), R21 = 2.f * (yz + wx), R22 = 1.f - 2.f * (x2 + y2);

            if constexpr (!LOADS_FIRST) s3 = ld_f32x3(scales, i);
            float s0 = s3.x, s1 = s3.y, s2 = s3.z;
            if (P.scales_are_log) { s0 = expf(s0); s1 = expf(s1); s2 = expf(s2); }

            // pinhole Jacobian with the 1.3x FOV clamp; T = J Wv (2x3)
            float rz, m2x, m2y;
            {   // means2d: IEEE 1 / z and the reference's products, uncontracted -- bit for bit the oracle's
    #pragma clang fp contract(off)
                rz = 1.0f / z;
                m2x = P.fx * mx * rz + P.cx;
                m2y = P.fy * my * rz + P.cy;
            }
            const float tx = z * fminf(P.lim_x_pos, fmaxf(-P.lim_x_neg, mx * rz));
            const float ty = z * fminf(P.lim_y_pos, fmaxf(-P.lim_y_neg, my * rz));
            const float J00 = P.fx * rz, J02 = -(J00 * tx) * rz;
            const float J11 = P.fy * rz, J12 = -(J11 * ty) * rz;
            const float T00 = J00 * V[0] + J02 * V[8], T01 = J00 * V[1] + J02 * V[9], T02 = J00 * V[2] + J02 * V[10];
            const float T10 = J11 * V[4] + J12 * V[8], T11 = J11 * V[5] + J12 * V[9], T12 = J11 * V[6] + J12 * V[10];
            // A = T R diag(s)
            const float A00 = (T00 * R00 + T01 * R10 + T02 * R20) * s0, A01 = (T00 * R01 + T01 * R11 + T02 * R21) * s1,
                        A02 = (T00 * R02 + T01 * R12 + T02 * R22) * s2;
            const float A10 = (T10 * R00 + T11 * R10 + T12 * R20) * s0, A11 = (T10 * R01 + T11 * R11 + T12 * R21) * s1,
                        A12 = (T10 * R02 + T11 * R12 + T12 * R22) * s2;
            const float a = A00 * A00 + A01 * A01 + A02 * A02 + P.eps2d;
            const float b = A00 * A10 + A01 * A11 + A02 * A12;
            const float c = A10 * A10 + A11 * A11 + A12 * A12 + P.eps2d;

            const float det = a * c - b * b;
            alive = det > 0.f;

            float extend = 3.33f;
            if (alive && P.has_opacity) {
                float op = op_first;
                if constexpr (!LOADS_FIRST) op = ld_f32(opacities, i, 1, 0);
                if (op < kAlphaThreshold) {
                    alive = false;
                } else {
                    extend = fminf(extend, __builtin_amdgcn_sqrtf(2.0f * __logf(op * 255.0f)));
                }
            }
            if (alive) {
                const float rx = ceilf(extend * __builtin_amdgcn_sqrtf(a)), ry = ceilf(extend * __builtin_amdgcn_sqrtf(c));
                if (rx <= P.radius_clip && ry <= P.radius_clip) alive = false;
                if (m2x + rx <= 0.f || m2x - rx >= P.W || m2y + ry <= 0.f || m2y - ry >= P.H) alive = false;
                if (alive) {
                    const float inv_det = rcp_nr(det);
                    o_c0 = c * inv_det;
                    o_c1 = -b * inv_det;
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
}

}  // namespace ms
