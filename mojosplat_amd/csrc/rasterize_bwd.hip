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

}  // namespace

extern "C" int ms_rasterize_to_pixels_3dgs_bwd(
    int64_t N, int64_t M, const float *means2d, const float *conics, const float *colors, int CDIM,
    const float *opacities, const float *backgrounds, int W, int H, int tile_size,
    const int32_t *tile_ranges, const int32_t *flatten_ids, const float *render_alphas,
    const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
    float *v_means2d, float *v_conics, float *v_colors, float *v_opacities, void *stream) {
    MS_REQUIRE(N >= 0 && M >= 0 && M <= 0x7fffffffll, MS_ERR_INVALID_ARG, "rasterize_bwd: bad N/M");
    MS_REQUIRE(W > 0 && H > 0 && tile_size > 0, MS_ERR_INVALID_ARG, "rasterize_bwd: bad image/tile size");
    MS_REQUIRE(CDIM >= 1 && CDIM <= 32, MS_ERR_INVALID_ARG, "rasterize_bwd: CDIM %d not in 1..32", CDIM);
    if (M == 0 || N == 0) return MS_OK;
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
    if (CDIM <= 3) hipLaunchKernelGGL(k_rasterize_bwd<3>, grid, block, 0, st, A);
    else if (CDIM <= 4) hipLaunchKernelGGL(k_rasterize_bwd<4>, grid, block, 0, st, A);
    else if (CDIM <= 8) hipLaunchKernelGGL(k_rasterize_bwd<8>, grid, block, 0, st, A);
    else if (CDIM <= 16) hipLaunchKernelGGL(k_rasterize_bwd<16>, grid, block, 0, st, A);
    else hipLaunchKernelGGL(k_rasterize_bwd<32>, grid, block, 0, st, A);
    MS_LAUNCH_CHECK();
    return MS_OK;
}
