// Tile rasteriser (forward) for gfx950: front-to-back alpha compositing of each tile's
// depth-sorted Gaussian list.
//
// Semantics: mojosplat/kernels/rasterization.mojo:75-162 (== gsplat rasterize_to_pixels fwd):
// pixel centre (x+0.5, y+0.5); sigma = 0.5(a dx^2 + c dy^2) + b dx dy;
// alpha = min(0.999, o exp(-sigma)); skip if sigma < 0 or alpha < 1/255; stop BEFORE adding
// when T(1-alpha) <= 1e-4; out = pix + T * background.
//
// Mapping: one 256-thread workgroup (4 x wave64) per 16x16 pixel block; each wave owns a
// compact 8x8 pixel quad so that its 64 lanes agree on which Gaussians matter and finish
// together (the per-lane `done` loop exit is a wave-level early-out by construction).  The
// tile's list is staged 256 intersections at a time in LDS *including colours* (the
// reference gathers colours from global memory inside the per-pixel loop,
// rasterization.mojo:154-155); every inner-loop LDS read is a wave-uniform broadcast.
// A block-wide vote (`__syncthreads_and`) stops staging once every pixel is saturated.
#include <hip/hip_fp16.h>

#include "ms_common.hpp"

namespace {

struct RasterArgs {
    const float *means2d;
    const float *conics;
    const void *colors;
    const float *opacities;
    const float *backgrounds;
    const int32_t *tile_ranges;
    const int32_t *flatten_ids;
    float *render_colors;
    float *render_alphas;
    int32_t *last_ids;
    int W, H, ts, tw, nsx, nsub, cdim, tile0;
};

__device__ __forceinline__ float load_color(const float *p) { return *p; }
__device__ __forceinline__ float load_color(const __half *p) { return __half2float(*p); }

// CP = compile-time channel capacity (>= runtime cdim)
template <int CP, typename ColorT>
__global__ __launch_bounds__(256) void k_rasterize_fwd(RasterArgs A) {
    __shared__ float4 s_geo[256];      // mean.x, mean.y, opacity, conic.a
    __shared__ float2 s_con[256];      // conic.b, conic.c
    __shared__ float s_rgb[256 * CP];

    const int bt = blockIdx.x / A.nsub, sub = blockIdx.x - bt * A.nsub;
    const int tile = A.tile0 + bt;
    const int tile_y = tile / A.tw, tile_x = tile - tile_y * A.tw;
    const int sub_y = sub / A.nsx, sub_x = sub - sub_y * A.nsx;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int lx = ((wv & 1) << 3) | (lane & 7), ly = ((wv >> 1) << 3) | (lane >> 3);
    const int ox = sub_x * 16 + lx, oy = sub_y * 16 + ly;  // offset inside the tile
    const int X = tile_x * A.ts + ox, Y = tile_y * A.ts + oy;
    const bool inside = ox < A.ts && oy < A.ts && X < A.W && Y < A.H;
    const float px = (float)X + 0.5f, py = (float)Y + 0.5f;

    const int start = A.tile_ranges[2 * tile], end = A.tile_ranges[2 * tile + 1];
    const ColorT *colors = reinterpret_cast<const ColorT *>(A.colors);

    float T = 1.0f;
    float pix[CP];
#pragma unroll
    for (int k = 0; k < CP; ++k) pix[k] = 0.f;
    int last = 0;
    bool done = !inside;

    for (int b0 = start; b0 < end; b0 += 256) {
        if (__syncthreads_and(done)) break;
        const int idx = b0 + tid;
        if (idx < end) {
            const int g = A.flatten_ids[idx];
            const float2 m = reinterpret_cast<const float2 *>(A.means2d)[g];
            const float ca = A.conics[3 * g], cb = A.conics[3 * g + 1], cc = A.conics[3 * g + 2];
            s_geo[tid] = make_float4(m.x, m.y, A.opacities[g], ca);
            s_con[tid] = make_float2(cb, cc);
#pragma unroll
            for (int k = 0; k < CP; ++k)
                if (k < A.cdim) s_rgb[tid * CP + k] = load_color(colors + (size_t)g * A.cdim + k);
        }
        __syncthreads();
        const int bs = min(256, end - b0);
        for (int t = 0; t < bs && !done; ++t) {
            const float4 ge = s_geo[t];
            const float2 co = s_con[t];
            const float dx = ge.x - px, dy = ge.y - py;
            const float sigma = 0.5f * (ge.w * dx * dx + co.y * dy * dy) + co.x * dx * dy;
            const float alpha = fminf(ms::kMaxAlpha, ge.z * __expf(-sigma));
            if (sigma < 0.f || alpha < ms::kAlphaThreshold) continue;
            const float next_T = T * (1.0f - alpha);
            if (next_T <= ms::kTransmittanceStop) {
                done = true;
                break;
            }
            const float vis = alpha * T;
#pragma unroll
            for (int k = 0; k < CP; ++k)
                if (k < A.cdim) pix[k] += s_rgb[t * CP + k] * vis;
            last = b0 + t;
            T = next_T;
        }
    }
    if (inside) {
        const size_t p = (size_t)Y * A.W + X;
#pragma unroll
        for (int k = 0; k < CP; ++k)
            if (k < A.cdim)
                A.render_colors[p * A.cdim + k] = pix[k] + (A.backgrounds ? T * A.backgrounds[k] : 0.f);
        if (A.render_alphas) A.render_alphas[p] = 1.0f - T;
        if (A.last_ids) A.last_ids[p] = last;
    }
}

template <typename ColorT>
int launch_fwd(const RasterArgs &A, int T_tiles, hipStream_t stream) {
    const dim3 grid((unsigned)(T_tiles * A.nsub)), block(256);
    if (A.cdim == 3) hipLaunchKernelGGL((k_rasterize_fwd<3, ColorT>), grid, block, 0, stream, A);
    else if (A.cdim <= 4) hipLaunchKernelGGL((k_rasterize_fwd<4, ColorT>), grid, block, 0, stream, A);
    else if (A.cdim <= 8) hipLaunchKernelGGL((k_rasterize_fwd<8, ColorT>), grid, block, 0, stream, A);
    else if (A.cdim <= 16) hipLaunchKernelGGL((k_rasterize_fwd<16, ColorT>), grid, block, 0, stream, A);
    else hipLaunchKernelGGL((k_rasterize_fwd<32, ColorT>), grid, block, 0, stream, A);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

}  // namespace

extern "C" int ms_rasterize_to_pixels_3dgs_fwd(int64_t N, int64_t M, const float *means2d,
                                               const float *conics, const void *colors,
                                               int color_dtype, int CDIM, const float *opacities,
                                               const float *backgrounds, int W, int H,
                                               int tile_size, int tile_row_begin,
                                               int tile_row_end, const int32_t *tile_ranges,
                                               const int32_t *flatten_ids, float *render_colors,
                                               float *render_alphas, int32_t *last_ids,
                                               void *stream) {
    MS_REQUIRE(N >= 0 && M >= 0 && M <= 0x7fffffffll, MS_ERR_INVALID_ARG, "rasterize_fwd: bad N/M");
    MS_REQUIRE(W > 0 && H > 0 && tile_size > 0, MS_ERR_INVALID_ARG, "rasterize_fwd: bad image/tile size");
    MS_REQUIRE(CDIM >= 1 && CDIM <= 32, MS_ERR_INVALID_ARG, "rasterize_fwd: CDIM %d not in 1..32", CDIM);
    MS_REQUIRE(color_dtype == MS_COLOR_F32 || color_dtype == MS_COLOR_F16, MS_ERR_INVALID_ARG,
               "rasterize_fwd: unknown colour dtype %d", color_dtype);
    MS_REQUIRE(tile_ranges && render_colors, MS_ERR_INVALID_ARG, "rasterize_fwd: null pointer");
    MS_REQUIRE(M == 0 || (means2d && conics && colors && opacities && flatten_ids), MS_ERR_INVALID_ARG,
               "rasterize_fwd: null input with M > 0");
    MS_REQUIRE(((uintptr_t)means2d & 7) == 0, MS_ERR_INVALID_ARG, "rasterize_fwd: means2d must be 8-byte aligned");
    RasterArgs A;
    A.means2d = means2d; A.conics = conics; A.colors = colors; A.opacities = opacities;
    A.backgrounds = backgrounds; A.tile_ranges = tile_ranges; A.flatten_ids = flatten_ids;
    A.render_colors = render_colors; A.render_alphas = render_alphas; A.last_ids = last_ids;
    A.W = W; A.H = H; A.ts = tile_size;
    A.tw = (W + tile_size - 1) / tile_size;
    const int th = (H + tile_size - 1) / tile_size;
    A.nsx = (tile_size + 15) / 16;
    A.nsub = A.nsx * A.nsx;
    A.cdim = CDIM;
    MS_REQUIRE(tile_row_begin >= 0 && tile_row_begin <= tile_row_end && tile_row_end <= th,
               MS_ERR_INVALID_ARG, "rasterize_fwd: bad tile row band [%d,%d) of %d", tile_row_begin,
               tile_row_end, th);
    A.tile0 = tile_row_begin * A.tw;
    const int band_tiles = (tile_row_end - tile_row_begin) * A.tw;
    if (band_tiles == 0) return MS_OK;
    const int64_t blocks = (int64_t)band_tiles * A.nsub;
    MS_REQUIRE(blocks <= 0x7fffffff, MS_ERR_TOO_LARGE, "rasterize_fwd: too many tiles");
    if (color_dtype == MS_COLOR_F16) return launch_fwd<__half>(A, band_tiles, (hipStream_t)stream);
    return launch_fwd<float>(A, band_tiles, (hipStream_t)stream);
}
