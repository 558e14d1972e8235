// Round 6 prototype: can the forward rasteriser's log2(alpha) come off the matrix pipe?
//
// log2(alpha)(pixel, Gaussian) expanded about the centre of the wave's 8x8 quad is a contraction
//     [pixel x 6 monomials {ux^2, ux uy, uy^2, ux, uy, 1}] . [6 coefficients {a', b', c', D, E, F} x Gaussian]
// (profiles/r05_raster_expanded.md: the expanded form passes the precision gate).  The monomials (u in {-3.5 .. 3.5}) are
// exact in bf16; every fp32 coefficient is split EXACTLY into three bf16 terms (8 + 8 + 8 mantissa bits), K = 18 of 32:
// one v_mfma_f32_16x16x32_bf16 gives 16 Gaussians x 16 pixels, four of them a whole quad, and a 4x4 transpose across the
// lane groups (8 v_permlane32_swap + 8 v_permlane16_swap per 16 Gaussians) brings a pixel's sixteen values into its lane.
//
// Two kernels over the same LDS-resident record stream, one wave per 8x8 quad as in k_rasterize_fwd<NQ = 1>:
//   base : the shipped blend loop (two ds_read_b128 + blue per record, 15 VALU per evaluation)
//   mfma : per 16 records one ds_read_b128 of the A fragment, 4 MFMA, 16 swaps; per record one colour read + 8 VALU
// and a check of the matrix path's log2(alpha) against the direct fp32 form on every (pixel, record).
//
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fgpu-flush-denormals-to-zero raster_mfma.hip -o raster_mfma
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr float kFlushK = 0x1.fep-119f, kTScale = 0x1p+126f, kAlphaOfV = 0x1.010102p+118f, kStop = 1e-4f;
constexpr int kN = 64;   // records resident per wave (4 chunks of 16)

struct Rec { float mx, my, a, b, c, lo, r, g, bl; };   // mean relative to the quad's first pixel centre; a', b', c' (log2 units)

#ifndef WAVES
#define WAVES 8
#endif

// ---- the shipped loop -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64, WAVES) void k_base(const Rec *recs, float *out, int iters) {
    __shared__ float4 sa[kN + 2], sb[kN + 2];
    __shared__ float sc[kN + 2];
    const int lane = threadIdx.x;
    const float px = (float)(lane & 7), py = (float)(lane >> 3);
    {
        const Rec r = recs[(blockIdx.x % 64) * kN + lane];
        sa[lane] = make_float4(r.mx, r.my, r.a, r.b);
        sb[lane] = make_float4(r.c, r.lo, r.r, r.g);
        sc[lane] = r.bl;
    }
    __syncthreads();
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, kq = kFlushK, acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float t = kTScale;
        int base = 0;
        asm volatile("" : "+s"(base));
#pragma unroll 1
        for (int k0 = base; k0 < base + kN; k0 += 2) {
            float4 ra[2], rb[2];
            float rc[2], m[2], v[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { ra[j] = sa[k0 + j]; rb[j] = sb[k0 + j]; rc[j] = sc[k0 + j]; }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float dx = ra[j].x - px, dy = ra[j].y - py;
                const float la = fmaf(dx, fmaf(ra[j].z, dx, ra[j].w * dy), fmaf(rb[j].x * dy, dy, rb[j].y));
                m[j] = __builtin_amdgcn_exp2f(la) * kq;
                asm volatile("" : "+v"(m[j]));
            }
            const float t_in = t;
#pragma unroll
            for (int j = 0; j < 2; ++j) { v[j] = m[j] * t; t = fmaf(v[j], -kAlphaOfV, t); }
            if (__ballot(!(t > kStop * kTScale))) {
                asm volatile("" ::: "memory");
                t = t_in;
                bool dead = false;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float vj = m[j] * t, nt = fmaf(vj, -kAlphaOfV, t);
                    dead = dead || !(nt > kStop * kTScale);
                    v[j] = dead ? 0.f : vj;
                    t = dead ? t : nt;
                }
                kq = dead ? 0.f : kq;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) { p0 += rb[j].z * v[j]; p1 += rb[j].w * v[j]; p2 += rc[j] * v[j]; }
        }
        acc += t * 0x1p-126f;
    }
    out[blockIdx.x * 64 + lane] = p0 + p1 + p2 + acc + kq;
}

// ---- the matrix-pipe form ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack_hi(float even, float odd) {   // bf16 by truncation: (odd's top half) | (even's top half >> 16)
    return (__float_as_uint(odd) & 0xffff0000u) | (__float_as_uint(even) >> 16);
}
// x = t0 + t1 + t2 exactly, each term a bf16 (truncations of the running remainder)
__device__ __forceinline__ void split3(float x, float &t0, float &t1, float &t2) {
    t0 = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    const float r1 = x - t0;
    t1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    t2 = r1 - t1;
}

template <bool CHECK_LA>
__global__ __launch_bounds__(64, WAVES) void k_mfma(const Rec *recs, float *out, int iters, float *la_err) {
    // A fragments: 64 B per record (32 bf16: k = 3 * monomial + term, 18 used, the rest zero for good); colours 16 B
    __shared__ __attribute__((aligned(16))) uint4 sA[kN * 4];
    __shared__ float4 sC[kN];
    const int lane = threadIdx.x;
    const int lx = lane & 7, ly = lane >> 3;
    Rec mine = recs[(blockIdx.x % 64) * kN + lane];
    {   // staging conversion of this lane's record (what the real kernel's staging lane would do per (record, quad))
        const float U = mine.mx - 3.5f, V = mine.my - 3.5f;   // mean relative to the quad's centre
        const float D = -fmaf(2.0f * mine.a, U, mine.b * V), E = -fmaf(2.0f * mine.c, V, mine.b * U);
        const float F = fmaf(U, fmaf(mine.a, U, mine.b * V), fmaf(mine.c * V, V, mine.lo));
        float e[24];
        split3(mine.a, e[0], e[1], e[2]);  split3(mine.b, e[3], e[4], e[5]);   split3(mine.c, e[6], e[7], e[8]);
        split3(D, e[9], e[10], e[11]);     split3(E, e[12], e[13], e[14]);     split3(F, e[15], e[16], e[17]);
#pragma unroll
        for (int k = 18; k < 24; ++k) e[k] = 0.f;
        sA[lane * 4 + 0] = make_uint4(pack_hi(e[0], e[1]), pack_hi(e[2], e[3]), pack_hi(e[4], e[5]), pack_hi(e[6], e[7]));
        sA[lane * 4 + 1] = make_uint4(pack_hi(e[8], e[9]), pack_hi(e[10], e[11]), pack_hi(e[12], e[13]), pack_hi(e[14], e[15]));
        sA[lane * 4 + 2] = make_uint4(pack_hi(e[16], e[17]), 0u, 0u, 0u);
        sA[lane * 4 + 3] = make_uint4(0u, 0u, 0u, 0u);
        sC[lane] = make_float4(mine.r, mine.g, mine.bl, 0.f);
    }
    // B operands: MFMA p covers the quad's pixels 16 p .. 16 p + 15 (rows 2p, 2p + 1); this lane holds column c = lane & 15,
    // k-slice s = lane >> 4 (k = 8 s + j)
    uint4 Bp[4];
    {
        const int c = lane & 15, s = lane >> 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float ux = (float)(c & 7) - 3.5f, uy = (float)(2 * p + (c >> 3)) - 3.5f;
            const float mono[6] = {ux * ux, ux * uy, uy * uy, ux, uy, 1.0f};
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * s + j;
                float val = 0.f;
#pragma unroll
                for (int q = 0; q < 6; ++q) val = (k / 3 == q && k < 18) ? mono[q] : val;
                e[j] = val;
            }
            Bp[p] = make_uint4(pack_hi(e[0], e[1]), pack_hi(e[2], e[3]), pack_hi(e[4], e[5]), pack_hi(e[6], e[7]));
        }
    }
    __syncthreads();
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, kq = kFlushK, acc = 0.f;
    float worst = 0.f;
    const int a_off = (lane & 15) * 4 + (lane >> 4);   // uint4 index of this lane's A slice inside a chunk
    for (int it = 0; it < iters; ++it) {
        float t = kTScale;
        int base = 0;
        asm volatile("" : "+s"(base));
#pragma unroll 1
        for (int ch = base; ch < base + kN / 16; ++ch) {
            const uint4 af = sA[ch * 64 + a_off];
            const bf16x8 A = __builtin_bit_cast(bf16x8, af);
            f32x4 X[4];
#pragma unroll
            for (int p = 0; p < 4; ++p)
                X[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, __builtin_bit_cast(bf16x8, Bp[p]), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            // 4x4 transpose across the lane groups: afterwards X[r][i] of lane L = log2(alpha)(record 4 r + i, pixel L)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(X[p][i]), __float_as_uint(X[p + 2][i]), false, false);
                    X[p][i] = __uint_as_float(r[0]); X[p + 2][i] = __uint_as_float(r[1]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int p = 0; p < 4; p += 2) {
                    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[p][i]), __float_as_uint(X[p + 1][i]), false, false);
                    X[p][i] = __uint_as_float(r[0]); X[p + 1][i] = __uint_as_float(r[1]);
                }
            }
            if constexpr (CHECK_LA) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const Rec r = recs[(blockIdx.x % 64) * kN + ch * 16 + g];
                    const float dx = r.mx - (float)lx, dy = r.my - (float)ly;
                    const float la = fmaf(dx, fmaf(r.a, dx, r.b * dy), fmaf(r.c * dy, dy, r.lo));
                    // error that matters: of alpha where alpha can matter (log2 alpha > -9)
                    const float got = X[g >> 2][g & 3];
                    if (la > -9.f) worst = fmaxf(worst, fabsf(got - la));
                }
            }
#pragma unroll
            for (int g = 0; g < 16; g += 2) {
                float m[2], v[2];
                float4 col[2];
                asm volatile("" ::: "memory");   // (keeps the pair's colour reads here: hoisted to the chunk's head they spill)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    col[j] = sC[ch * 16 + g + j];
                    m[j] = __builtin_amdgcn_exp2f(X[(g + j) >> 2][(g + j) & 3]) * kq;
                    asm volatile("" : "+v"(m[j]));
                }
                const float t_in = t;
#pragma unroll
                for (int j = 0; j < 2; ++j) { v[j] = m[j] * t; t = fmaf(v[j], -kAlphaOfV, t); }
                if (__ballot(!(t > kStop * kTScale))) {
                    asm volatile("" ::: "memory");
                    t = t_in;
                    bool dead = false;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float vj = m[j] * t, nt = fmaf(vj, -kAlphaOfV, t);
                        dead = dead || !(nt > kStop * kTScale);
                        v[j] = dead ? 0.f : vj;
                        t = dead ? t : nt;
                    }
                    kq = dead ? 0.f : kq;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) { p0 += col[j].x * v[j]; p1 += col[j].y * v[j]; p2 += col[j].z * v[j]; }
                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2));   // (or the compiler sinks all sixteen blends to the loop's latch and spills their operands)
            }
        }
        acc += t * 0x1p-126f;
    }
    out[blockIdx.x * 64 + lane] = p0 + p1 + p2 + acc + kq;
    if constexpr (CHECK_LA) la_err[blockIdx.x * 64 + lane] = worst;
}

// ---- the same on v_mfma_f32_32x32x16_bf16 (two K-steps): 32 Gaussians x 32 pixels per MFMA, two pixel halves; a lane pair
// {l, l + 32} then holds 2 pixels x 32 Gaussians and ONE v_permlane32_swap per register pair finishes the transpose: half a
// swap per evaluation instead of one, for 32 accumulators instead of 16 and chunks of 32.
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(64, WAVES) void k_mfma32(const Rec *recs, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) uint4 sA[kN * 4];
    __shared__ float4 sC[kN];
    const int lane = threadIdx.x;
    Rec mine = recs[(blockIdx.x % 64) * kN + lane];
    {
        const float U = mine.mx - 3.5f, V = mine.my - 3.5f;
        const float D = -fmaf(2.0f * mine.a, U, mine.b * V), E = -fmaf(2.0f * mine.c, V, mine.b * U);
        const float F = fmaf(U, fmaf(mine.a, U, mine.b * V), fmaf(mine.c * V, V, mine.lo));
        float e[24];
        split3(mine.a, e[0], e[1], e[2]);  split3(mine.b, e[3], e[4], e[5]);   split3(mine.c, e[6], e[7], e[8]);
        split3(D, e[9], e[10], e[11]);     split3(E, e[12], e[13], e[14]);     split3(F, e[15], e[16], e[17]);
        sA[lane * 4 + 0] = make_uint4(pack_hi(e[0], e[1]), pack_hi(e[2], e[3]), pack_hi(e[4], e[5]), pack_hi(e[6], e[7]));
        sA[lane * 4 + 1] = make_uint4(pack_hi(e[8], e[9]), pack_hi(e[10], e[11]), pack_hi(e[12], e[13]), pack_hi(e[14], e[15]));
        sA[lane * 4 + 2] = make_uint4(pack_hi(e[16], e[17]), 0u, 0u, 0u);
        sA[lane * 4 + 3] = make_uint4(0u, 0u, 0u, 0u);
        sC[lane] = make_float4(mine.r, mine.g, mine.bl, 0.f);
    }
    uint4 Bp[2][2];   // [pixel half][K-step]
    {
        const int c = lane & 31, h = lane >> 5;
#pragma unroll
        for (int P = 0; P < 2; ++P) {
            const int idx = 32 * P + c;
            const float ux = (float)(idx & 7) - 3.5f, uy = (float)(idx >> 3) - 3.5f;
            const float mono[6] = {ux * ux, ux * uy, uy * uy, ux, uy, 1.0f};
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * st + 8 * h + j;
                    float val = 0.f;
#pragma unroll
                    for (int q = 0; q < 6; ++q) val = (k / 3 == q && k < 18) ? mono[q] : val;
                    e[j] = val;
                }
                Bp[P][st] = make_uint4(pack_hi(e[0], e[1]), pack_hi(e[2], e[3]), pack_hi(e[4], e[5]), pack_hi(e[6], e[7]));
            }
        }
    }
    __syncthreads();
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, kq = kFlushK, acc = 0.f;
    const int a_off = (lane & 31) * 4 + (lane >> 5);   // uint4 index of this lane's K-step-0 slice inside a chunk of 32 (+2: step 1)
    for (int it = 0; it < iters; ++it) {
        float t = kTScale;
        int base = 0;
        asm volatile("" : "+s"(base));
#pragma unroll 1
        for (int ch = base; ch < base + kN / 32; ++ch) {
            const bf16x8 A0 = __builtin_bit_cast(bf16x8, sA[ch * 128 + a_off]), A1 = __builtin_bit_cast(bf16x8, sA[ch * 128 + a_off + 2]);
            f32x16 X[2];
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                f32x16 z = {};
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, __builtin_bit_cast(bf16x8, Bp[P][0]), z, 0, 0, 0);
                X[P] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, __builtin_bit_cast(bf16x8, Bp[P][1]), z, 0, 0, 0);
            }
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) {
                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(X[0][r_]), __float_as_uint(X[1][r_]), false, false);
                X[0][r_] = __uint_as_float(r[0]); X[1][r_] = __uint_as_float(r[1]);
            }
#pragma unroll
            for (int g = 0; g < 32; g += 2) {
                float m[2], v[2];
                float4 col[2];
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int gg = g + j;
                    col[j] = sC[ch * 32 + gg];
                    m[j] = __builtin_amdgcn_exp2f(X[(gg >> 2) & 1][4 * (gg >> 3) + (gg & 3)]) * kq;
                    asm volatile("" : "+v"(m[j]));
                }
                const float t_in = t;
#pragma unroll
                for (int j = 0; j < 2; ++j) { v[j] = m[j] * t; t = fmaf(v[j], -kAlphaOfV, t); }
                if (__ballot(!(t > kStop * kTScale))) {
                    asm volatile("" ::: "memory");
                    t = t_in;
                    bool dead = false;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float vj = m[j] * t, nt = fmaf(vj, -kAlphaOfV, t);
                        dead = dead || !(nt > kStop * kTScale);
                        v[j] = dead ? 0.f : vj;
                        t = dead ? t : nt;
                    }
                    kq = dead ? 0.f : kq;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) { p0 += col[j].x * v[j]; p1 += col[j].y * v[j]; p2 += col[j].z * v[j]; }
                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2));
            }
        }
        acc += t * 0x1p-126f;
    }
    out[blockIdx.x * 64 + lane] = p0 + p1 + p2 + acc + kq;
}

// ---- the exact form: v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate), no split: the staging lane only computes D, E, F.
// K = 6 monomials = two instructions per 16 records x 16 pixels (k 0-3: ux^2, ux uy, uy^2, ux; k 4-7: uy, 1, 0, 0); a lane holds
// ONE float of A (record l & 15, k = l >> 4) and of B (k = l >> 4, pixel l & 15) per instruction; the transpose as k_mfma.
__global__ __launch_bounds__(64, WAVES) void k_mfma_f32(const Rec *recs, float *out, int iters, float *la_err) {
    __shared__ float sK[kN * 8];     // coefficients: record r at sK[r * 8 + k]
    __shared__ float4 sC[kN];
    const int lane = threadIdx.x;
    const int lx = lane & 7, ly = lane >> 3;
    Rec mine = recs[(blockIdx.x % 64) * kN + lane];
    {
        const float U = mine.mx - 3.5f, V = mine.my - 3.5f;
        const float D = -fmaf(2.0f * mine.a, U, mine.b * V), E = -fmaf(2.0f * mine.c, V, mine.b * U);
        const float F = fmaf(U, fmaf(mine.a, U, mine.b * V), fmaf(mine.c * V, V, mine.lo));
        *reinterpret_cast<float4 *>(sK + lane * 8) = make_float4(mine.a, mine.b, mine.c, D);
        *reinterpret_cast<float4 *>(sK + lane * 8 + 4) = make_float4(E, F, 0.f, 0.f);
        sC[lane] = make_float4(mine.r, mine.g, mine.bl, 0.f);
    }
    float Bm[4][2];   // [pixel block][k group]
    {
        const int c = lane & 15, kk = lane >> 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float ux = (float)(c & 7) - 3.5f, uy = (float)(2 * p + (c >> 3)) - 3.5f;
            const float m0[4] = {ux * ux, ux * uy, uy * uy, ux}, m1[4] = {uy, 1.0f, 0.f, 0.f};
            Bm[p][0] = kk == 0 ? m0[0] : kk == 1 ? m0[1] : kk == 2 ? m0[2] : m0[3];
            Bm[p][1] = kk == 0 ? m1[0] : kk == 1 ? m1[1] : 0.f;
        }
    }
    __syncthreads();
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, kq = kFlushK, acc = 0.f, worst = 0.f;
    const int a_off = (lane & 15) * 8 + (lane >> 4);   // float index of this lane's k-group-0 coefficient inside a chunk (+4: group 1)
    for (int it = 0; it < iters; ++it) {
        float t = kTScale;
        int base = 0;
        asm volatile("" : "+s"(base));
#pragma unroll 1
        for (int ch = base; ch < base + kN / 16; ++ch) {
            const float A0 = sK[ch * 128 + a_off], A1 = sK[ch * 128 + a_off + 4];
            f32x4 X[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                z = __builtin_amdgcn_mfma_f32_16x16x4f32(A0, Bm[p][0], z, 0, 0, 0);
                X[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1, Bm[p][1], z, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(X[p][i]), __float_as_uint(X[p + 2][i]), false, false);
                    X[p][i] = __uint_as_float(r[0]); X[p + 2][i] = __uint_as_float(r[1]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int p = 0; p < 4; p += 2) {
                    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[p][i]), __float_as_uint(X[p + 1][i]), false, false);
                    X[p][i] = __uint_as_float(r[0]); X[p + 1][i] = __uint_as_float(r[1]);
                }
            }
            if (la_err && it == 0) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const Rec r = recs[(blockIdx.x % 64) * kN + ch * 16 + g];
                    const float dx = r.mx - (float)lx, dy = r.my - (float)ly;
                    const float la = fmaf(dx, fmaf(r.a, dx, r.b * dy), fmaf(r.c * dy, dy, r.lo));
                    if (la > -9.f) worst = fmaxf(worst, fabsf(X[g >> 2][g & 3] - la));
                }
            }
#pragma unroll
            for (int g = 0; g < 16; g += 2) {
                float m[2], v[2];
                float4 col[2];
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    col[j] = sC[ch * 16 + g + j];
                    m[j] = __builtin_amdgcn_exp2f(X[(g + j) >> 2][(g + j) & 3]) * kq;
                    asm volatile("" : "+v"(m[j]));
                }
                const float t_in = t;
#pragma unroll
                for (int j = 0; j < 2; ++j) { v[j] = m[j] * t; t = fmaf(v[j], -kAlphaOfV, t); }
                if (__ballot(!(t > kStop * kTScale))) {
                    asm volatile("" ::: "memory");
                    t = t_in;
                    bool dead = false;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float vj = m[j] * t, nt = fmaf(vj, -kAlphaOfV, t);
                        dead = dead || !(nt > kStop * kTScale);
                        v[j] = dead ? 0.f : vj;
                        t = dead ? t : nt;
                    }
                    kq = dead ? 0.f : kq;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) { p0 += col[j].x * v[j]; p1 += col[j].y * v[j]; p2 += col[j].z * v[j]; }
                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2));
            }
        }
        acc += t * 0x1p-126f;
    }
    out[blockIdx.x * 64 + lane] = p0 + p1 + p2 + acc + kq;
    if (la_err) la_err[blockIdx.x * 64 + lane] = worst;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000, blocks = 256 * 4 * WAVES * 4;
    std::vector<Rec> h(64 * kN);
    srand(7);
    auto U01 = []() { return (float)rand() / (float)RAND_MAX; };
    for (auto &r : h) {
        r.mx = -6.f + 20.f * U01(); r.my = -6.f + 20.f * U01();
        const float a = 0.02f + 2.3f * U01() * U01(), c = 0.02f + 2.3f * U01() * U01();
        const float b = (2.f * U01() - 1.f) * 0.9f * sqrtf(a * c) * 2.f;   // |b'| < 2 sqrt(a' c'): positive definite
        r.a = -a; r.b = b; r.c = -c;
        r.lo = -3.f * U01();
        r.r = U01(); r.g = U01(); r.bl = U01();
    }
    Rec *d_r; float *d_o, *d_o2, *d_e;
    hipMalloc(&d_r, h.size() * sizeof(Rec)); hipMalloc(&d_o, blocks * 64 * 4); hipMalloc(&d_o2, blocks * 64 * 4); hipMalloc(&d_e, blocks * 64 * 4);
    hipMemcpy(d_r, h.data(), h.size() * sizeof(Rec), hipMemcpyHostToDevice);
    // correctness: one pass, both kernels, same pixels
    hipLaunchKernelGGL(k_base, dim3(64), dim3(64), 0, 0, d_r, d_o, 1);
    hipLaunchKernelGGL(k_mfma<true>, dim3(64), dim3(64), 0, 0, d_r, d_o2, 1, d_e);
    std::vector<float> o(64 * 64), o2(64 * 64), e(64 * 64);
    hipMemcpy(o.data(), d_o, o.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(o2.data(), d_o2, o2.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(e.data(), d_e, e.size() * 4, hipMemcpyDeviceToHost);
    float worst = 0.f, dpix = 0.f;
    for (size_t i = 0; i < o.size(); ++i) { worst = fmaxf(worst, e[i]); dpix = fmaxf(dpix, fabsf(o[i] - o2[i])); }
    {
        hipLaunchKernelGGL(k_mfma32, dim3(64), dim3(64), 0, 0, d_r, d_o2, 1);
        std::vector<float> o3(64 * 64);
        hipMemcpy(o3.data(), d_o2, o3.size() * 4, hipMemcpyDeviceToHost);
        float d32 = 0.f;
        for (size_t i = 0; i < o.size(); ++i) d32 = fmaxf(d32, fabsf(o[i] - o3[i]));
        printf("check: max |pixel sum base - mfma32| (255 x colour units): %.3g\n", d32);
    }
    {
        hipLaunchKernelGGL(k_mfma_f32, dim3(64), dim3(64), 0, 0, d_r, d_o2, 1, d_e);
        std::vector<float> o3(64 * 64), e3(64 * 64);
        hipMemcpy(o3.data(), d_o2, o3.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(e3.data(), d_e, e3.size() * 4, hipMemcpyDeviceToHost);
        float d32 = 0.f, w3 = 0.f;
        for (size_t i = 0; i < o.size(); ++i) { d32 = fmaxf(d32, fabsf(o[i] - o3[i])); w3 = fmaxf(w3, e3[i]); }
        printf("check: fp32 matrix form: max |log2(alpha) - direct| where > -9: %.3g ; max |pixel sum base - mfma_f32|: %.3g\n", w3, d32);
    }
    printf("check: max |log2(alpha) mfma - direct| where log2(alpha) > -9: %.3g ; max |pixel sum base - mfma| (255 x colour units): %.3g\n", worst, dpix);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipLaunchKernelGGL(k_base, dim3(blocks), dim3(64), 0, 0, d_r, d_o, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_base, dim3(blocks), dim3(64), 0, 0, d_r, d_o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double evals_per_simd = (double)blocks * iters * kN / 1024.0;
        printf("base  (WAVES=%d) %.3f ms  %.2f ns per evaluation and SIMD\n", WAVES, ms, ms * 1e6 / evals_per_simd);
        hipLaunchKernelGGL(k_mfma<false>, dim3(blocks), dim3(64), 0, 0, d_r, d_o2, 10, d_e);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma<false>, dim3(blocks), dim3(64), 0, 0, d_r, d_o2, iters, d_e);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("mfma  (WAVES=%d) %.3f ms  %.2f ns per evaluation and SIMD\n", WAVES, ms, ms * 1e6 / evals_per_simd);
        hipLaunchKernelGGL(k_mfma32, dim3(blocks), dim3(64), 0, 0, d_r, d_o2, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma32, dim3(blocks), dim3(64), 0, 0, d_r, d_o2, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("mfma32(WAVES=%d) %.3f ms  %.2f ns per evaluation and SIMD\n", WAVES, ms, ms * 1e6 / evals_per_simd);
        hipLaunchKernelGGL(k_mfma_f32, dim3(blocks), dim3(64), 0, 0, d_r, d_o2, 10, (float *)nullptr);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma_f32, dim3(blocks), dim3(64), 0, 0, d_r, d_o2, iters, (float *)nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("mfmaf32(WAVES=%d) %.3f ms  %.2f ns per evaluation and SIMD\n", WAVES, ms, ms * 1e6 / evals_per_simd);
    }
    return 0;
}
