// Tile rasteriser backward for gfx950.
//
// The reference is forward-only (mojosplat/render.py:11 `@torch.no_grad()`, README.md:145);
// this implements the backward of the forward in rasterize.hip with gsplat's semantics
// (rasterize_to_pixels backward, absgrad off): per pixel, walk the tile's list back to front
// starting at the pixel's last contributing intersection, recover T by dividing out (1-alpha),
// and accumulate d/d(mean2d, conic, colour, opacity).
//
// MI355X mapping: same 16x16 block / 8x8-quad-per-wave layout as the forward.  Per Gaussian
// the 64 lanes of a wave are summed with DPP row shifts + row broadcasts (no LDS traffic, no
// ds_bpermute), waves whose quad cannot see the Gaussian skip it on a ballot, the four waves
// of a block meet in an LDS accumulator, and ONE global float atomic per (tile, Gaussian,
// component) leaves the CU at the end of each 256-intersection batch.
#include <stdlib.h>

#include "ms_common.hpp"

namespace {

struct RasterBwdArgs {
    const float *means2d;
    const float *conics;
    const float *colors;
    const float *opacities;
    const float *backgrounds;
    const int32_t *tile_ranges;
    const int32_t *flatten_ids;
    const float *render_alphas;
    const int32_t *last_ids;
    const float *v_render_colors;
    const float *v_render_alphas;
    float *v_means2d;
    float *v_conics;
    float *v_colors;
    float *v_opacities;
    int W, H, ts, tw, nsx, nsub, cdim;
};

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true);
    return v + __int_as_float(moved);
}

// Sum over the 64 lanes; the total is valid in lane 63 only.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    float s = v;
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));  // row_shr:1
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));  // row_shr:2
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x113, 0xf, 0xf, true));  // row_shr:3
    s = dpp_add<0x114, 0xf, 0xe>(s);   // row_shr:4, banks 1-3
    s = dpp_add<0x118, 0xf, 0xc>(s);   // row_shr:8, banks 2-3   -> lane 15 of each row = row sum
    s = dpp_add<0x142, 0xa, 0xf>(s);   // row_bcast:15 into rows 1 and 3
    s = dpp_add<0x143, 0xc, 0xf>(s);   // row_bcast:31 into rows 2 and 3 -> lane 63 = total
    return s;
}

template <int CP>
__global__ __launch_bounds__(256) void k_rasterize_bwd(RasterBwdArgs A) {
    constexpr int NG = 6 + CP;  // mean.xy, conic.abc, opacity, colour[CP]
    __shared__ float4 s_geo[256];
    __shared__ float2 s_con[256];
    __shared__ int s_id[256];
    __shared__ float s_rgb[256 * CP];
    __shared__ float s_acc[256 * NG];

    const int tile = blockIdx.x / A.nsub, sub = blockIdx.x - tile * A.nsub;
    const int tile_y = tile / A.tw, tile_x = tile - tile_y * A.tw;
    const int sub_y = sub / A.nsx, sub_x = sub - sub_y * A.nsx;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int lx = ((wv & 1) << 3) | (lane & 7), ly = ((wv >> 1) << 3) | (lane >> 3);
    const int ox = sub_x * 16 + lx, oy = sub_y * 16 + ly;
    const int X = tile_x * A.ts + ox, Y = tile_y * A.ts + oy;
    const bool inside = ox < A.ts && oy < A.ts && X < A.W && Y < A.H;
    const float px = (float)X + 0.5f, py = (float)Y + 0.5f;
    const int start = A.tile_ranges[2 * tile], end = A.tile_ranges[2 * tile + 1];
    if (end <= start) return;

    const size_t p = inside ? (size_t)Y * A.W + X : 0;
    const float T_final = inside ? 1.0f - A.render_alphas[p] : 1.0f;
    float T = T_final;
    const int bin_final = inside ? A.last_ids[p] : -1;
    float v_out[CP], buffer[CP];
    float bg_dot = 0.f;
#pragma unroll
    for (int k = 0; k < CP; ++k) {
        v_out[k] = (inside && k < A.cdim) ? A.v_render_colors[p * A.cdim + k] : 0.f;
        buffer[k] = 0.f;
        if (A.backgrounds && k < A.cdim) bg_dot += A.backgrounds[k] * v_out[k];
    }
    const float v_a = (inside && A.v_render_alphas) ? A.v_render_alphas[p] : 0.f;

    // highest intersection index any lane of this wave still needs
    int wave_final = bin_final;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wave_final = max(wave_final, __shfl_xor(wave_final, d));

    const int n = end - start;
    for (int b0 = 0; b0 < n; b0 += 256) {
        __syncthreads();
        const int batch_end = end - 1 - b0;      // index handled by t = 0
        const int idx = batch_end - tid;
        if (idx >= start) {
            const int g = A.flatten_ids[idx];
            s_id[tid] = g;
            const float2 m = reinterpret_cast<const float2 *>(A.means2d)[g];
            s_geo[tid] = make_float4(m.x, m.y, A.opacities[g], A.conics[3 * g]);
            s_con[tid] = make_float2(A.conics[3 * g + 1], A.conics[3 * g + 2]);
#pragma unroll
            for (int k = 0; k < CP; ++k)
                if (k < A.cdim) s_rgb[tid * CP + k] = A.colors[(size_t)g * A.cdim + k];
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) s_acc[tid * NG + j] = 0.f;
        __syncthreads();
        const int bs = min(256, batch_end + 1 - start);
        // skip straight to the first intersection some lane of this wave contributed to
        for (int t = max(0, batch_end - wave_final); t < bs; ++t) {
            const int cur = batch_end - t;
            const float4 ge = s_geo[t];
            const float2 co = s_con[t];
            const float dx = ge.x - px, dy = ge.y - py;
            const float sigma = 0.5f * (ge.w * dx * dx + co.y * dy * dy) + co.x * dx * dy;
            const float vis = __expf(-sigma);
            const float alpha = fminf(ms::kMaxAlpha, ge.z * vis);
            const bool valid = inside && cur <= bin_final && sigma >= 0.f && alpha >= ms::kAlphaThreshold;
            if (!__any(valid)) continue;

            float g_out[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) g_out[j] = 0.f;
            if (valid) {
                const float ra = 1.0f / (1.0f - alpha);
                T *= ra;
                const float fac = alpha * T;
                float v_alpha = 0.f;
#pragma unroll
                for (int k = 0; k < CP; ++k) {
                    if (k < A.cdim) {
                        const float c = s_rgb[t * CP + k];
                        g_out[6 + k] = fac * v_out[k];
                        v_alpha += (c * T - buffer[k] * ra) * v_out[k];
                        buffer[k] += c * fac;
                    }
                }
                v_alpha += T_final * ra * v_a;
                v_alpha -= T_final * ra * bg_dot;
                if (ge.z * vis <= ms::kMaxAlpha) {
                    const float v_sigma = -ge.z * vis * v_alpha;
                    g_out[0] = v_sigma * (ge.w * dx + co.x * dy);
                    g_out[1] = v_sigma * (co.x * dx + co.y * dy);
                    g_out[2] = 0.5f * v_sigma * dx * dx;
                    g_out[3] = v_sigma * dx * dy;
                    g_out[4] = 0.5f * v_sigma * dy * dy;
                    g_out[5] = vis * v_alpha;
                }
            }
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                if (j < 6 + A.cdim) {  // wave-uniform
                    const float s = wave_sum_to_lane63(g_out[j]);
                    if (lane == 63) atomicAdd(&s_acc[t * NG + j], s);
                }
            }
        }
        __syncthreads();
        if (idx >= start) {
            const int g = s_id[tid];
            const float *acc = s_acc + tid * NG;
            if (acc[0] != 0.f) atomicAdd(A.v_means2d + 2 * g, acc[0]);
            if (acc[1] != 0.f) atomicAdd(A.v_means2d + 2 * g + 1, acc[1]);
            if (acc[2] != 0.f) atomicAdd(A.v_conics + 3 * g, acc[2]);
            if (acc[3] != 0.f) atomicAdd(A.v_conics + 3 * g + 1, acc[3]);
            if (acc[4] != 0.f) atomicAdd(A.v_conics + 3 * g + 2, acc[4]);
            if (acc[5] != 0.f) atomicAdd(A.v_opacities + g, acc[5]);
#pragma unroll
            for (int k = 0; k < CP; ++k)
                if (k < A.cdim && acc[6 + k] != 0.f) atomicAdd(A.v_colors + (size_t)g * A.cdim + k, acc[6 + k]);
        }
    }
}

// ---- v2 (CDIM <= 4): one wave64 per 16x16 block, four pixels per lane -------------------------
// Same block/quad layout, staging and bbox culling as the forward kernel (rasterize.hip), walked
// back to front.  What differs from v1 above:
//   * the 4 quads of an entry accumulate into per-lane partial sums first, so there is ONE wave
//     reduction (9 values) per (tile, Gaussian), not one per (wave, Gaussian);
//   * conic / mean gradients are linear in five per-lane sums (vs*dx, vs*dy, vs*dx^2, vs*dx*dy,
//     vs*dy^2), so the lane that holds the totals finishes them with the entry's conic;
//   * totals go to a 64-byte row per Gaussian in a packed scratch buffer (f32[N][16]): the batch
//     flush adds 4 whole rows per wave-instruction -- contiguous float atomics run ~17x faster
//     than one-dword-per-row ones on MI355X (MI355X_MICROARCH.md, Global float atomics);
//     k_unpack_grads then adds the rows into the caller's four gradient tensors.
#ifndef MS_BWD_WAVES
#define MS_BWD_WAVES 4
#endif
#ifndef MS_BWD_ABLATE
#define MS_BWD_ABLATE 0
#endif
constexpr int kRow = 16;  // floats per packed gradient row: mx my ca cb cc op c0 c1 c2 c3 - - - - - -

struct RasterBwd2Args {
    RasterBwdArgs a;
    float *packed;  // f32[N][kRow], zeroed by the caller of the kernel
    const int32_t *order;   // blocks, heaviest list first (k_bwd_order), or null: image order
    int nblocks, n_gauss;
    const float4 *records;  // the forward frame's ready-made ms::RasterRecords (3 channels), or null: the four arrays
};

// Launch order of the blocks: heaviest list first (counting sort of the tiles over 128 length buckets, 4 per
// octave; one workgroup).  The backward walks every block's list as deep as its pixels blended, in waves that
// hold 90 VGPRs (5 per SIMD, 5 120 slots for >= 8 160 waves): in image order the heavy centre tiles that start
// late finish alone.  Round 2 measured the same effect on the forward kernel (profiles/r02_raster_waves.md).
__global__ __launch_bounds__(1024) void k_bwd_order(int n_tiles, int nsub, const int32_t *__restrict__ tile_ranges,
                                                     int32_t *__restrict__ order) {
    __shared__ unsigned int s_bkt[128], s_base[128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    auto bucket_of = [](unsigned int c) -> int {
        if (c == 0) return 0;
        const int l = 31 - __clz((int)c);
        const int frac = l >= 2 ? (int)((c >> (l - 2)) & 3u) : (int)((c << (2 - l)) & 3u);
        return 1 + 4 * l + frac;
    };
    if (threadIdx.x < 128) s_bkt[threadIdx.x] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < n_tiles; t += 1024)
        atomicAdd(&s_bkt[bucket_of((unsigned)max(tile_ranges[2 * t + 1] - tile_ranges[2 * t], 0))], 1u);
    __syncthreads();
    if (w == 0) {   // s_base[b] = tiles in heavier buckets
        const unsigned int c0 = s_bkt[127 - 2 * lane], c1 = s_bkt[126 - 2 * lane];
        unsigned int incl = c0 + c1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int o = (unsigned int)__shfl_up((int)incl, d);
            if (lane >= d) incl += o;
        }
        const unsigned int excl = incl - (c0 + c1);
        s_base[127 - 2 * lane] = excl;
        s_base[126 - 2 * lane] = excl + c0;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n_tiles; t += 1024) {
        const unsigned int slot = atomicAdd(&s_base[bucket_of((unsigned)max(tile_ranges[2 * t + 1] - tile_ranges[2 * t], 0))], 1u);
        for (int s_ = 0; s_ < nsub; ++s_) order[slot * nsub + s_] = t * nsub + s_;
    }
}

__device__ __forceinline__ void wave_lds_sync_bwd() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d));
    // every lane holds the maximum: hand it back as a SCALAR, so that everything derived from it
    // (per-quad entry masks, loop bounds) stays in SGPRs and branches on it are scalar branches
    return __builtin_amdgcn_readfirstlane(v);
}

// Sum 8 per-lane values over the 64 lanes with a halving butterfly: at distance 1, 2, 4 each lane
// keeps half of its values and receives the partner's matching half (DPP quad_perm / row shifts),
// so after three steps it holds ONE value -- number (lane & 7) -- summed over its group of 8 lanes;
// three more single-value steps (distance 8, 16, 32) finish.  27 VALU ops instead of 8 x 7 for
// eight separate reductions.  Returns the wave total of v[lane & 7] in every lane.
__device__ __forceinline__ float wave_allreduce8(const float (&v)[8], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    auto xchg1 = [](float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true)); };
    auto xchg2 = [](float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true)); };
    float w[4], x[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (b0 ? v[2 * i + 1] : v[2 * i]) + xchg1(b0 ? v[2 * i] : v[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 2; ++i) x[i] = (b1 ? w[2 * i + 1] : w[2 * i]) + xchg2(b1 ? w[2 * i] : w[2 * i + 1]);
    const float send = b2 ? x[0] : x[1];
    int recv = __builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x104, 0xf, 0x5, true);      // row_shl:4 -> lanes with bit 2 clear
    recv = __builtin_amdgcn_update_dpp(recv, __float_as_int(send), 0x114, 0xf, 0xa, false);     // row_shr:4 -> lanes with bit 2 set
    float y = (b2 ? x[1] : x[0]) + __int_as_float(recv);
    y += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y), 0x128, 0xf, 0xf, true));  // row_ror:8
    y += __shfl_xor(y, 16);
    y += __shfl_xor(y, 32);
    return y;
}

// NQ = quads per wave: 4 = one wave per 16x16 block; 2 = two waves per block (upper / lower strip; each stages the
// list but blends only its quads): the longest wave's work halves and twice as many waves fill the slots.
// REC: stage from the forward frame's ready-made 48-byte records (ms_render_bwd: three 16-byte gathers and no
// arithmetic per entry instead of seven 4-12-byte gathers, the conic's scaling and the bound's logarithm -- the same
// treatment the forward rasteriser got in round 2).  The records hold the conic pre-scaled exactly as s_a / s_b want
// it, log2(opacity) (the opacity is taken back out with one exp2: an ulp beside the caller's value, far inside the
// gradients' tolerance), and the three numbers of the ellipse-vs-quad test.
template <int CP, int NQ, bool REC>
__global__ __launch_bounds__(64, MS_BWD_WAVES) void k_rasterize_bwd_v2(RasterBwd2Args B2) {
    static_assert(!REC || CP == 3, "ready-made records carry three channels");
    const RasterBwdArgs &A = B2.a;
    constexpr int NG = 6 + CP;
    constexpr int kParts = 4 / NQ;
    constexpr float kLog2e = 1.4426950408889634f;
    __shared__ float4 s_a[64];       // mean.x, mean.y, a', b'   (conic pre-scaled by -log2e/2, -log2e)
    __shared__ float4 s_b[64];       // c', opacity, colour0, colour1
    __shared__ float4 s_c[64];       // colour2, colour3, Gaussian id (bits), -
    __shared__ float s_grad[64 * kRow];

    // the kParts waves of a block sit 8 blockIdx apart (dealt round-robin over the 8 XCDs: one XCD, shared L2)
    int wg = blockIdx.x, part = 0;
    if constexpr (kParts > 1) {
        const int j = blockIdx.x >> 3;
        part = j % kParts;
        wg = ((j / kParts) << 3) | (blockIdx.x & 7);
        if (wg >= B2.nblocks) return;
    }
    const int qbase = part * NQ;
    const int item = B2.order ? B2.order[wg] : wg;
    const int tile = item / A.nsub, sub = item - tile * A.nsub;
    const int tile_y = tile / A.tw, tile_x = tile - tile_y * A.tw;
    const int sub_y = sub / A.nsx, sub_x = sub - sub_y * A.nsx;
    const int lane = threadIdx.x;
    const int lx = lane & 7, ly = lane >> 3;
    const int bx = tile_x * A.ts + sub_x * 16, by = tile_y * A.ts + sub_y * 16;
    const int ox = sub_x * 16 + lx, oy = sub_y * 16 + ly;
    const float px0 = (float)(bx + lx) + 0.5f, py0 = (float)(by + ly) + 0.5f;
    const int start = A.tile_ranges[2 * tile], end = A.tile_ranges[2 * tile + 1];
    if (end <= start) return;

    float T[NQ], tb[NQ], v_out[NQ][CP], buf[NQ][CP];
    int binf[NQ], qmax[NQ];
    int wave_final = -1;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int qq = qbase + q;
        const int X = bx + lx + (qq & 1) * 8, Y = by + ly + (qq >> 1) * 8;
        const bool in = (ox + (qq & 1) * 8) < A.ts && (oy + (qq >> 1) * 8) < A.ts && X < A.W && Y < A.H;
        const size_t p = in ? (size_t)Y * A.W + X : 0;
        const float T_final = in ? 1.0f - A.render_alphas[p] : 1.0f;
        T[q] = T_final;
        binf[q] = in ? A.last_ids[p] : -1;
        float bg_dot = 0.f;
#pragma unroll
        for (int k = 0; k < CP; ++k) {
            v_out[q][k] = (in && k < A.cdim) ? A.v_render_colors[p * A.cdim + k] : 0.f;
            buf[q][k] = 0.f;
            if (A.backgrounds && k < A.cdim) bg_dot += A.backgrounds[k] * v_out[q][k];
        }
        const float v_a = (in && A.v_render_alphas) ? A.v_render_alphas[p] : 0.f;
        tb[q] = T_final * (v_a - bg_dot);
        qmax[q] = wave_max_i32(binf[q]);
        wave_final = max(wave_final, qmax[q]);
    }
    if (wave_final < start) return;
    const int hi = min(end - 1, wave_final);  // last intersection anyone in this block blended
    const float fbx = (float)bx + 0.5f, fby = (float)by + 0.5f;

    for (int b0 = start + ((hi - start) & ~63); b0 >= start; b0 -= 64) {
        // ---- stage entries b0 .. b0+63 (ascending); walk them from the back
        const int idx = b0 + lane;
        int mask = 0, g = 0;
        float mx = 0.f, my = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, op = 0.f, col[4] = {0.f, 0.f, 0.f, 0.f};
        float4 st_a = make_float4(0.f, 0.f, 0.f, 0.f), st_b = st_a;   // what goes to s_a / s_b
        if constexpr (REC) {
            if (idx <= hi) {
                constexpr float kInf = __builtin_huge_valf();
                g = min(max(A.flatten_ids[idx], 0), B2.n_gauss - 1);
                const float4 *rec = B2.records + 3 * (size_t)g;
                const float4 ra = rec[0], rb = rec[1], rc = rec[2];
                const float smax = rc.y, nb_c = rc.z, nb_a = rc.w;
                st_a = ra;
                st_b = make_float4(rb.x, __builtin_amdgcn_exp2f(rb.y), rb.z, rb.w);
                col[2] = rc.x;
                if (smax == kInf) {
                    mask = 0xf;   // no bound (not positive definite, or the 0.999 clamp can bind): every quad
                } else if (smax > -kInf) {
                    // exact ellipse-vs-quad test in log2 units on the record, as in the forward kernel (rasterize.hip)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int qq = qbase + q;
                        const float xl = fbx + (float)((qq & 1) * 8) - ra.x, xh = xl + 7.0f;
                        const float yl = fby + (float)((qq >> 1) * 8) - ra.y, yh = yl + 7.0f;
                        const bool in_x = xl <= 0.f && xh >= 0.f, in_y = yl <= 0.f && yh >= 0.f;
                        float best = (in_x && in_y) ? 0.f : 3.0e38f;
                        if (!in_x) {
                            const float dx = xl > 0.f ? xl : xh;
                            const float dy = fminf(fmaxf(nb_c * dx, yl), yh);
                            best = -(ra.z * dx * dx + rb.x * dy * dy + ra.w * dx * dy);
                        }
                        if (!in_y) {
                            const float dy = yl > 0.f ? yl : yh;
                            const float dx = fminf(fmaxf(nb_a * dy, xl), xh);
                            best = fminf(best, -(ra.z * dx * dx + rb.x * dy * dy + ra.w * dx * dy));
                        }
                        mask |= (best <= smax) ? (1 << q) : 0;
                    }
                }
            }
        } else
        if (idx <= hi) {
            g = min(max(A.flatten_ids[idx], 0), B2.n_gauss - 1);
            const float2 m = reinterpret_cast<const float2 *>(A.means2d)[g];
            mx = m.x; my = m.y;
            ca = A.conics[3 * g]; cb = A.conics[3 * g + 1]; cc = A.conics[3 * g + 2];
            op = A.opacities[g];
#pragma unroll
            for (int k = 0; k < CP; ++k)
                if (k < A.cdim) col[k] = A.colors[(size_t)g * A.cdim + k];
            if (op >= ms::kAlphaThreshold) {
                const float det = ca * cc - cb * cb;
                if (det > 0.f && ca > 0.f && cc > 0.f) {
                    // exact ellipse-vs-quad test, as in the forward kernel (rasterize.hip)
                    const float smax = __logf(op * 255.0f) * 1.0001f + 1e-4f;
                    const float nb_c = -cb * __builtin_amdgcn_rcpf(cc), nb_a = -cb * __builtin_amdgcn_rcpf(ca);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int qq = qbase + q;
                        const float xl = fbx + (float)((qq & 1) * 8) - mx, xh = xl + 7.0f;
                        const float yl = fby + (float)((qq >> 1) * 8) - my, yh = yl + 7.0f;
                        const bool in_x = xl <= 0.f && xh >= 0.f, in_y = yl <= 0.f && yh >= 0.f;
                        float best = (in_x && in_y) ? 0.f : 3.0e38f;
                        if (!in_x) {
                            const float dx = xl > 0.f ? xl : xh;
                            const float dy = fminf(fmaxf(nb_c * dx, yl), yh);
                            best = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                        }
                        if (!in_y) {
                            const float dy = yl > 0.f ? yl : yh;
                            const float dx = fminf(fmaxf(nb_a * dy, xl), xh);
                            best = fminf(best, 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy);
                        }
                        mask |= (best <= smax) ? (1 << q) : 0;
                    }
                } else {
                    mask = 0xf;
                }
            }
        }
        unsigned long long Bq[NQ], U = 0ull;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            Bq[q] = __ballot((mask >> q) & 1);
            // entries behind the last one any pixel of this quad blended cannot matter to it
            const int lim = qmax[q] - b0;
            Bq[q] = lim < 0 ? 0ull : (lim >= 63 ? Bq[q] : (Bq[q] & ((2ull << lim) - 1ull)));
            U |= Bq[q];
        }
        wave_lds_sync_bwd();  // the previous batch's flush has read s_grad / s_c
        if (mask) {
            if constexpr (REC) {
                s_a[lane] = st_a;
                s_b[lane] = st_b;
            } else {
                s_a[lane] = make_float4(mx, my, -0.5f * kLog2e * ca, -kLog2e * cb);
                s_b[lane] = make_float4(-0.5f * kLog2e * cc, op, col[0], col[1]);
            }
            s_c[lane] = make_float4(col[2], col[3], __int_as_float(g), 0.f);
        }
        wave_lds_sync_bwd();

        unsigned long long flush = 0ull;
        while (U) {
            const int t = 63 - __clzll((long long)U);
            U &= ~(1ull << t);
            const int cur = b0 + t;
            const float4 ra = s_a[t];
            const float4 rb = s_b[t];
            const float4 rc = s_c[t];
            const float cv[4] = {rb.z, rb.w, rc.x, rc.y};
            float acc[NG];  // Sx Sy S1 S2 S3 op colour[CP]
#pragma unroll
            for (int j = 0; j < NG; ++j) acc[j] = 0.f;
            unsigned long long anyv = 0ull;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!((Bq[q] >> t) & 1ull)) continue;  // wave-uniform
                const int qq = qbase + q;
                const float dx = ra.x - (px0 + (float)((qq & 1) * 8)), dy = ra.y - (py0 + (float)((qq >> 1) * 8));
                const float lp = fmaf(dx, fmaf(ra.z, dx, ra.w * dy), rb.x * dy * dy);  // -sigma*log2(e)
                const float vis = __builtin_amdgcn_exp2f(lp);
                const float ov = rb.y * vis;
                const float alpha = fminf(ms::kMaxAlpha, ov);
                const bool valid = cur <= binf[q] && lp <= 0.f && alpha >= ms::kAlphaThreshold;
                anyv |= __ballot(valid);
                // Branch-free from here on: lanes that do not blend this entry contribute zeros through
                // selects.  (Divergent `if (valid)` blocks made the compiler copy the nine accumulators
                // at every merge -- ~30 v_mov per evaluation -- and an IEEE division cost 12 more.)
                const float ra_ = __builtin_amdgcn_rcpf(1.0f - alpha);
                const float Tn = T[q] * ra_;                  // transmittance in front of this entry
                T[q] = valid ? Tn : T[q];
                const float fac = valid ? alpha * Tn : 0.f;
                float v_alpha = tb[q] * ra_;
#pragma unroll
                for (int k = 0; k < CP; ++k) {
                    acc[6 + k] = fmaf(fac, v_out[q][k], acc[6 + k]);
                    v_alpha = fmaf(fmaf(cv[k], Tn, -buf[q][k] * ra_), v_out[q][k], v_alpha);
                    buf[q][k] = fmaf(cv[k], fac, buf[q][k]);
                }
                const bool unclamped = valid && ov <= ms::kMaxAlpha;   // the 0.999 clamp has no gradient
                const float vs = unclamped ? -ov * v_alpha : 0.f;
                const float sx = vs * dx, sy = vs * dy;
                acc[0] += sx;
                acc[1] += sy;
                acc[2] = fmaf(sx, dx, acc[2]);
                acc[3] = fmaf(sx, dy, acc[3]);
                acc[4] = fmaf(sy, dy, acc[4]);
                acc[5] += unclamped ? vis * v_alpha : 0.f;
            }
            if (anyv == 0ull) continue;
            // raw sums Sx Sy S1 S2 S3 op c0 c1 -> lanes 0..7; remaining colour channel(s) -> lane 63
            const float first8[8] = {acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]};
#if MS_BWD_ABLATE & 2   // (measurement builds only, profiles/r04_bwd_pmc.md: the walk without its wave reductions)
            const float y = first8[0] + first8[1] + first8[2] + first8[3] + first8[4] + first8[5] + first8[6] + first8[7];
            if (lane < 8) s_grad[t * kRow + lane] = y;
            if (lane == 63) s_grad[t * kRow + 8] = acc[8];
#else
            const float y = wave_allreduce8(first8, lane);
            if (lane < 8) s_grad[t * kRow + lane] = y;
#pragma unroll
            for (int j = 8; j < NG; ++j) {
                const float tj = wave_sum_to_lane63(acc[j]);
                if (lane == 63) s_grad[t * kRow + j] = tj;
            }
#endif
            flush |= 1ull << t;
        }
        wave_lds_sync_bwd();
        // ---- flush: 4 packed rows (4 x 64 B contiguous) per wave instruction
        const int colm = lane & 15, rsub = lane >> 4;
#pragma unroll 1
        for (int r4 = 0; r4 < 64; r4 += 4) {
            if (((flush >> r4) & 0xfull) == 0ull) continue;
            const int r = r4 + rsub;
            if (((flush >> r) & 1ull) && colm < NG) {
                const int gid = __float_as_int(s_c[r].z);
                const float *raw = s_grad + r * kRow;
                float val = raw[colm];
                if (colm < 2) {
                    // undo the staging scale: a' = -log2e/2 * ca, b' = -log2e * cb, c' = -log2e/2 * cc
                    const float4 qa = s_a[r];
                    const float ecb = qa.w * (-1.0f / kLog2e);
                    const float diag = (colm == 0 ? qa.z : s_b[r].x) * (-2.0f / kLog2e);   // ca for mean.x, cc for mean.y
                    val = colm == 0 ? diag * raw[0] + ecb * raw[1] : ecb * raw[0] + diag * raw[1];
                } else if (colm == 2 || colm == 4) {
                    val *= 0.5f;
                }
#if MS_BWD_ABLATE & 1   // (measurement builds only: the kernel without its global atomics)
                if (val == 1.2345678e-30f) B2.packed[(size_t)gid * kRow + colm] = val;
#else
                atomicAdd(B2.packed + (size_t)gid * kRow + colm, val);
#endif
            }
        }
    }
}

template <bool OVERWRITE>
__global__ __launch_bounds__(256) void k_unpack_grads(int64_t N, int cdim, const float *__restrict__ packed,
                                                      float *__restrict__ v_means2d, float *__restrict__ v_conics,
                                                      float *__restrict__ v_colors, float *__restrict__ v_opacities) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float4 *row = reinterpret_cast<const float4 *>(packed + i * kRow);
    const float4 r0 = row[0], r1 = row[1], r2 = row[2];
    auto put = [](float &dst, float v) { if (OVERWRITE) dst = v; else dst += v; };
    put(v_means2d[2 * i], r0.x);
    put(v_means2d[2 * i + 1], r0.y);
    put(v_conics[3 * i], r0.z);
    put(v_conics[3 * i + 1], r0.w);
    put(v_conics[3 * i + 2], r1.x);
    put(v_opacities[i], r1.y);
    const float c[4] = {r1.z, r1.w, r2.x, r2.y};
    for (int k = 0; k < cdim && k < 4; ++k) put(v_colors[i * cdim + k], c[k]);
}

}  // namespace

extern "C" int ms_rasterize_to_pixels_3dgs_bwd(
    int64_t N, int64_t M, const float *means2d, const float *conics, const float *colors, int CDIM,
    const float *opacities, const float *backgrounds, int W, int H, int tile_size,
    const int32_t *tile_ranges, const int32_t *flatten_ids, const float *render_alphas,
    const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
    float *v_means2d, float *v_conics, float *v_colors, float *v_opacities, void *workspace,
    size_t workspace_bytes, int overwrite, void *stream) {
    return ms::rasterize_bwd(N, M, means2d, conics, colors, CDIM, opacities, backgrounds, W, H, tile_size, tile_ranges,
                             flatten_ids, render_alphas, last_ids, v_render_colors, v_render_alphas, v_means2d, v_conics,
                             v_colors, v_opacities, workspace, workspace_bytes, overwrite, nullptr, nullptr, stream);
}

int ms::rasterize_bwd(
    int64_t N, int64_t M, const float *means2d, const float *conics, const float *colors, int CDIM,
    const float *opacities, const float *backgrounds, int W, int H, int tile_size,
    const int32_t *tile_ranges, const int32_t *flatten_ids, const float *render_alphas,
    const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
    float *v_means2d, float *v_conics, float *v_colors, float *v_opacities, void *workspace,
    size_t workspace_bytes, int overwrite, const void *records, const int32_t *block_order, void *stream) {
    MS_REQUIRE(N >= 0 && M >= 0 && M <= 0x7fffffffll, MS_ERR_INVALID_ARG, "rasterize_bwd: bad N/M");
    MS_REQUIRE(W > 0 && H > 0 && tile_size > 0, MS_ERR_INVALID_ARG, "rasterize_bwd: bad image/tile size");
    MS_REQUIRE(CDIM >= 1 && CDIM <= 32, MS_ERR_INVALID_ARG, "rasterize_bwd: CDIM %d not in 1..32", CDIM);
    if (N == 0) return MS_OK;
    const bool packed_path = CDIM <= 4 && workspace && workspace_bytes >= (size_t)N * kRow * sizeof(float) &&
                             N <= 0x7fffffffll && M > 0;
    if (overwrite && !packed_path) {
        // the packed path overwrites in its unpack kernel; everywhere else "overwrite" = zero-fill, then add
        MS_REQUIRE(v_means2d && v_conics && v_colors && v_opacities, MS_ERR_INVALID_ARG, "rasterize_bwd: null output");
        hipStream_t s0 = (hipStream_t)stream;
        MS_HIP(hipMemsetAsync(v_means2d, 0, (size_t)N * 2 * sizeof(float), s0));
        MS_HIP(hipMemsetAsync(v_conics, 0, (size_t)N * 3 * sizeof(float), s0));
        MS_HIP(hipMemsetAsync(v_colors, 0, (size_t)N * CDIM * sizeof(float), s0));
        MS_HIP(hipMemsetAsync(v_opacities, 0, (size_t)N * sizeof(float), s0));
    }
    if (M == 0) return MS_OK;
    MS_REQUIRE(means2d && conics && colors && opacities && tile_ranges && flatten_ids && render_alphas &&
                   last_ids && v_render_colors && v_means2d && v_conics && v_colors && v_opacities,
               MS_ERR_INVALID_ARG, "rasterize_bwd: null pointer");
    MS_REQUIRE(((uintptr_t)means2d & 7) == 0, MS_ERR_INVALID_ARG, "rasterize_bwd: means2d must be 8-byte aligned");
    RasterBwdArgs A;
    A.means2d = means2d; A.conics = conics; A.colors = colors; A.opacities = opacities;
    A.backgrounds = backgrounds; A.tile_ranges = tile_ranges; A.flatten_ids = flatten_ids;
    A.render_alphas = render_alphas; A.last_ids = last_ids; A.v_render_colors = v_render_colors;
    A.v_render_alphas = v_render_alphas; A.v_means2d = v_means2d; A.v_conics = v_conics;
    A.v_colors = v_colors; A.v_opacities = v_opacities;
    A.W = W; A.H = H; A.ts = tile_size;
    A.tw = (W + tile_size - 1) / tile_size;
    const int th = (H + tile_size - 1) / tile_size;
    A.nsx = (tile_size + 15) / 16;
    A.nsub = A.nsx * A.nsx;
    A.cdim = CDIM;
    const int64_t blocks = (int64_t)A.tw * th * A.nsub;
    MS_REQUIRE(blocks <= 0x7fffffff, MS_ERR_TOO_LARGE, "rasterize_bwd: too many tiles");
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = (hipStream_t)stream;
    const size_t packed_bytes = (size_t)N * kRow * sizeof(float);
    if (packed_path) {
        // v2: packed 64-byte gradient rows (contiguous float atomics), then unpack
        RasterBwd2Args B2;
        B2.a = A;
        B2.packed = (float *)workspace;
        B2.nblocks = (int)blocks;
        B2.n_gauss = (int)N;
        B2.records = (CDIM == 3 && ((uintptr_t)records & 15) == 0) ? (const float4 *)records : nullptr;
        MS_HIP(hipMemsetAsync(workspace, 0, packed_bytes, st));
        // heaviest blocks first when the workspace has room for the order (ms_rasterize_bwd_workspace_bytes
        // reserves it); two waves per block while the launch is a few rounds at most
        B2.order = nullptr;
        const size_t order_off = ms::align_up(packed_bytes, 256);
        if (block_order && A.nsub == 1) {
            // the forward frame's own heaviest-first order of the same tiles (its count pass left it): no k_bwd_order
            B2.order = block_order;
        } else
        if (workspace_bytes >= order_off + (size_t)blocks * sizeof(int32_t)) {
            int32_t *order = (int32_t *)((char *)workspace + order_off);
            hipLaunchKernelGGL(k_bwd_order, dim3(1), dim3(1024), 0, st, A.tw * th, A.nsub, tile_ranges, order);
            MS_LAUNCH_CHECK();
            B2.order = order;
        }
        // waves per block: ONE (measured at config 3, forward + backward step, with the heaviest-first order:
        // 1 wave 0.698 ms, 2 waves 0.732, 4 waves 0.832 -- each wave stages the list from four arrays and the
        // per-entry reduction + flush does not shrink with the quads; without the order 1 wave: 0.779).
        const dim3 grid2((unsigned)blocks);
#define MS_LAUNCH_BWD(CPV, RECV) hipLaunchKernelGGL((k_rasterize_bwd_v2<CPV, 4, RECV>), grid2, dim3(64), 0, st, B2)
        if (CDIM == 3 && B2.records) MS_LAUNCH_BWD(3, true);
        else if (CDIM <= 3) MS_LAUNCH_BWD(3, false);
        else MS_LAUNCH_BWD(4, false);
#undef MS_LAUNCH_BWD
        MS_LAUNCH_CHECK();
        // overwrite == 2 (ms_render_bwd): the rows stay packed -- the backward projection reads them as they are and
        // writes v_colors / v_opacities itself (project_bwd.hip, k_project_ewa_bwd<true>): one pass over them fewer
        if (overwrite == 2) return MS_OK;
        if (overwrite)
            hipLaunchKernelGGL(k_unpack_grads<true>, dim3((unsigned)ms::ceil_div(N, 256)), dim3(256), 0, st, N, CDIM,
                               (const float *)workspace, v_means2d, v_conics, v_colors, v_opacities);
        else
            hipLaunchKernelGGL(k_unpack_grads<false>, dim3((unsigned)ms::ceil_div(N, 256)), dim3(256), 0, st, N, CDIM,
                               (const float *)workspace, v_means2d, v_conics, v_colors, v_opacities);
        MS_LAUNCH_CHECK();
        return MS_OK;
    }
    if (CDIM <= 3) hipLaunchKernelGGL(k_rasterize_bwd<3>, grid, block, 0, st, A);
    else if (CDIM <= 4) hipLaunchKernelGGL(k_rasterize_bwd<4>, grid, block, 0, st, A);
    else if (CDIM <= 8) hipLaunchKernelGGL(k_rasterize_bwd<8>, grid, block, 0, st, A);
    else if (CDIM <= 16) hipLaunchKernelGGL(k_rasterize_bwd<16>, grid, block, 0, st, A);
    else hipLaunchKernelGGL(k_rasterize_bwd<32>, grid, block, 0, st, A);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// packed gradient rows + room for the launch order of up to 2^18 blocks (any frame up to 8K x 8K at 16-px tiles)
extern "C" size_t ms_rasterize_bwd_workspace_bytes(int64_t N, int CDIM) {
    return (CDIM >= 1 && CDIM <= 4 && N > 0) ? ms::align_up((size_t)N * kRow * sizeof(float), 256) + ((size_t)4 << 18) : 0;
}
