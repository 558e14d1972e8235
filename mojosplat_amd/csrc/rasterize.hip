// Tile rasteriser (forward) for gfx950: front-to-back alpha compositing of each tile's
// depth-sorted Gaussian list.
//
// Semantics: mojosplat/kernels/rasterization.mojo:75-162 (== gsplat rasterize_to_pixels fwd):
// pixel centre (x+0.5, y+0.5); sigma = 0.5(a dx^2 + c dy^2) + b dx dy;
// alpha = min(0.999, o exp(-sigma)); skip if sigma < 0 or alpha < 1/255; stop BEFORE adding
// when T(1-alpha) <= 1e-4; out = pix + T * background.
//
// MI355X mapping (v2).  The v1 kernel (256 threads per 16x16 block, one pixel per lane, like
// the reference's launch at rasterization.mojo:219-220) was bound by LDS *broadcast* reads:
// every wave re-reads every staged Gaussian, 48 LDS cycles per (tile, Gaussian) against 24
// VALU cycles.  Here ONE wave64 owns a whole 16x16 block and every lane carries four pixels
// (the same lane position in each of the four 8x8 quads), so a staged Gaussian is read from
// LDS once per block (12 cycles) and the exp2/blend work of four pixels shares it.
//   * staging: 64 intersections per batch, one per lane; the lane that stages a Gaussian also
//     folds log2(e) into its conic, takes log2(opacity), and computes which of the four quads
//     the alpha >= 1/255 ellipse can touch (exact bounding box of that ellipse + slack) -> a
//     4-bit mask stored with the record;
//   * inner loop: the mask is wave-uniform (readfirstlane -> scalar branch), so quads the
//     Gaussian cannot reach cost nothing; v_exp_f32 is exp2, so alpha = exp2(q + log2 o);
//   * colours sit in LDS with the geometry (the reference gathers them from global memory in
//     the per-pixel loop, rasterization.mojo:154-155) and are only read when some lane of the
//     wave actually blends;
//   * the next batch's gather (ids -> means/conics/opacity/colour, 36 B per intersection) is
//     issued into registers before the current batch is composited; the wave leaves the list
//     as soon as all 256 pixels are saturated (checked per batch of 64);
//   * blockIdx -> tile mapping hands each XCD a contiguous run of tiles so that neighbouring
//     tiles, which share most of their Gaussians, gather through the same L2.
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
    int W, H, ts, tw, nsx, nsub, cdim, tile0, nblocks;
};

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kBatch = 64;

__device__ __forceinline__ float load_color(const float *p) { return *p; }
__device__ __forceinline__ float load_color(const __half *p) { return __half2float(*p); }

// blockIdx -> work item.  Blocks are dealt round-robin over the 8 XCDs (bid % 8 labels the
// XCD group); give each XCD runs of 8 consecutive tiles (neighbours share most of their
// Gaussians -> same L2) while interleaving the runs over the whole image, so the heavy centre
// of a frame is spread evenly over the XCDs.  Bijective for any n; speed only.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int n64 = n & ~63;
    if (bid >= n64) return bid;
    const int xcd = bid & 7, k = bid >> 3;
    return ((k >> 3) << 6) | (xcd << 3) | (k & 7);
}

// CP = compile-time channel capacity (>= runtime cdim); CS = LDS stride of a colour record
template <int CP, typename ColorT>
__global__ __launch_bounds__(64) void k_rasterize_fwd(RasterArgs A) {
    constexpr int CS = (CP == 3) ? 4 : CP;
    __shared__ float4 s_a[kBatch + 1];        // mean.x, mean.y, a', b'
    __shared__ float4 s_b[kBatch + 1];        // c', log2(opacity), quad mask (bits), -
    __shared__ float s_col[kBatch * CS];

    const int item = xcd_remap(blockIdx.x, A.nblocks);
    const int bt = item / A.nsub, sub = item - bt * A.nsub;
    const int tile = A.tile0 + bt;
    const int tile_y = tile / A.tw, tile_x = tile - tile_y * A.tw;
    const int sub_y = sub / A.nsx, sub_x = sub - sub_y * A.nsx;
    const int lane = threadIdx.x;
    const int lx = lane & 7, ly = lane >> 3;
    const int bx = tile_x * A.ts + sub_x * 16, by = tile_y * A.ts + sub_y * 16;  // block origin (pixels)
    const int ox = sub_x * 16 + lx, oy = sub_y * 16 + ly;                        // offset inside the tile (quad 0)

    // pixel q of this lane: quad (q&1, q>>1)
    float px[2], py[2];
    px[0] = (float)(bx + lx) + 0.5f;      px[1] = px[0] + 8.0f;
    py[0] = (float)(by + ly) + 0.5f;      py[1] = py[0] + 8.0f;
    bool inside[4];
    float T[4];            // > 0: transmittance of a live pixel; < 0: -(transmittance) of a finished one
    float pix[4][CP];
    int last[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int X = bx + lx + (q & 1) * 8, Y = by + ly + (q >> 1) * 8;
        inside[q] = (ox + (q & 1) * 8) < A.ts && (oy + (q >> 1) * 8) < A.ts && X < A.W && Y < A.H;
        T[q] = inside[q] ? 1.0f : -1.0f;
        last[q] = 0;
#pragma unroll
        for (int k = 0; k < CP; ++k) pix[q][k] = 0.f;
    }

    const int start = A.tile_ranges[2 * tile], end = A.tile_ranges[2 * tile + 1];
    const ColorT *colors = reinterpret_cast<const ColorT *>(A.colors);
    // quad rectangles (pixel-centre extents) for the staging-time culling test
    const float qx_lo[2] = {(float)bx + 0.5f, (float)bx + 8.5f}, qy_lo[2] = {(float)by + 0.5f, (float)by + 8.5f};

    // registers for the batch being gathered
    float r_mx = 0.f, r_my = 0.f, r_ca = 0.f, r_cb = 0.f, r_cc = 0.f, r_op = 0.f;
    float r_col[CP];
    int r_g = 0;  // Gaussian id of the batch after the one being gathered (ids run one batch ahead)
    auto fetch_id = [&](int b0) {
        const int idx = b0 + lane;
        r_g = idx < end ? A.flatten_ids[idx] : 0;
    };
    auto gather = [&](int b0) {
        const int idx = b0 + lane;
        if (idx < end) {
            const int g = r_g;
            const float2 m = reinterpret_cast<const float2 *>(A.means2d)[g];
            r_mx = m.x; r_my = m.y;
            r_ca = A.conics[3 * g]; r_cb = A.conics[3 * g + 1]; r_cc = A.conics[3 * g + 2];
            r_op = A.opacities[g];
#pragma unroll
            for (int k = 0; k < CP; ++k) r_col[k] = k < A.cdim ? load_color(colors + (size_t)g * A.cdim + k) : 0.f;
        }
    };

    if (start < end) {
        fetch_id(start);
        gather(start);
        fetch_id(start + kBatch);
    }
    for (int b0 = start; b0 < end; b0 += kBatch) {
        __syncthreads();  // single-wave workgroup: orders LDS reads of the previous batch
        if (b0 + lane < end) {
            // which quads can the alpha >= 1/255 ellipse reach?  sigma <= ln(255 o) =: smax;
            // its bounding box is mean +- sqrt(2 smax cov_xx|yy), cov = conic^-1.
            int mask = 0;
            const float det = r_ca * r_cc - r_cb * r_cb;
            if (r_op >= ms::kAlphaThreshold) {
                if (det > 0.f && r_ca > 0.f && r_cc > 0.f) {
                    const float smax2 = 2.0f * __logf(r_op * 255.0f) * 1.0001f + 1e-4f;
                    const float inv = 1.0f / det;
                    const float hx = sqrtf(smax2 * r_cc * inv) * 1.0001f + 0.01f;
                    const float hy = sqrtf(smax2 * r_ca * inv) * 1.0001f + 0.01f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float xl = qx_lo[q & 1], yl = qy_lo[q >> 1];
                        const bool hit = (r_mx + hx >= xl) && (r_mx - hx <= xl + 7.0f) && (r_my + hy >= yl) &&
                                         (r_my - hy <= yl + 7.0f);
                        mask |= hit ? (1 << q) : 0;
                    }
                } else {
                    mask = 0xf;  // not positive definite: no bound, evaluate everywhere
                }
            }
            s_a[lane] = make_float4(r_mx, r_my, -0.5f * kLog2e * r_ca, -kLog2e * r_cb);
            s_b[lane] = make_float4(-0.5f * kLog2e * r_cc, __log2f(r_op), __int_as_float(mask), 0.f);
#pragma unroll
            for (int k = 0; k < CP; ++k) s_col[lane * CS + k] = r_col[k];
        }
        __syncthreads();
        if (b0 + kBatch < end) {  // both in flight while this batch is composited
            gather(b0 + kBatch);
            fetch_id(b0 + 2 * kBatch);
        }

        const int bs = min(kBatch, end - b0);
        float4 na = s_a[0], nb = s_b[0];
        for (int t = 0; t < bs; ++t) {
            const float4 ra = na, rb = nb;
            na = s_a[t + 1];  // next record in flight while this one is composited (slot bs is slack)
            nb = s_b[t + 1];
            const int mask = __builtin_amdgcn_readfirstlane(__float_as_int(rb.z));
            if (mask == 0) continue;
            float col[CP];
            if constexpr (CS == 4 && CP == 3) {
                const float4 c = reinterpret_cast<const float4 *>(s_col)[t];
                col[0] = c.x; col[1] = c.y; col[2] = c.z;
            } else {
#pragma unroll
                for (int k = 0; k < CP; ++k) col[k] = s_col[t * CS + k];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!(mask & (1 << q))) continue;  // wave-uniform
                const float dx = ra.x - px[q & 1], dy = ra.y - py[q >> 1];
                const float pw = dx * (ra.z * dx + ra.w * dy) + rb.x * dy * dy;  // = -sigma*log2(e)
                const float alpha = fminf(ms::kMaxAlpha, __builtin_amdgcn_exp2f(pw + rb.y));
                const bool hit = T[q] > 0.f && pw <= 0.f && alpha >= ms::kAlphaThreshold;
                if (!__any(hit)) continue;
                const float next_T = T[q] * (1.0f - alpha);
                const bool stop = hit && next_T <= ms::kTransmittanceStop;
                const bool add = hit && !stop;
                const float vis = add ? alpha * T[q] : 0.f;
#pragma unroll
                for (int k = 0; k < CP; ++k) pix[q][k] += col[k] * vis;
                last[q] = add ? b0 + t : last[q];
                T[q] = add ? next_T : (stop ? -T[q] : T[q]);
            }
        }
        const bool live = T[0] > 0.f || T[1] > 0.f || T[2] > 0.f || T[3] > 0.f;
        if (!__any(live)) break;
    }

#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!inside[q]) continue;
        const int X = bx + lx + (q & 1) * 8, Y = by + ly + (q >> 1) * 8;
        const size_t p = (size_t)Y * A.W + X;
        const float Tq = fabsf(T[q]);
#pragma unroll
        for (int k = 0; k < CP; ++k)
            if (k < A.cdim)
                A.render_colors[p * A.cdim + k] = pix[q][k] + (A.backgrounds ? Tq * A.backgrounds[k] : 0.f);
        if (A.render_alphas) A.render_alphas[p] = 1.0f - Tq;
        if (A.last_ids) A.last_ids[p] = last[q];
    }
}

template <typename ColorT>
int launch_fwd(const RasterArgs &A, hipStream_t stream) {
    const dim3 grid((unsigned)A.nblocks), block(64);
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
    A.nblocks = (int)blocks;
    if (color_dtype == MS_COLOR_F16) return launch_fwd<__half>(A, (hipStream_t)stream);
    return launch_fwd<float>(A, (hipStream_t)stream);
}
