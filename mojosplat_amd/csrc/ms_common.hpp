// Shared host-side plumbing for libmojosplat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mojosplat_hip.h"

namespace ms {

void set_error(const char *fmt, ...);

#define MS_REQUIRE(cond, code, ...)  \
    do {                             \
        if (!(cond)) {               \
            ms::set_error(__VA_ARGS__); \
            return (code);           \
        }                            \
    } while (0)

#define MS_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            ms::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                   \
            return MS_ERR_HIP;                                                         \
        }                                                                              \
    } while (0)

// after a kernel launch: catches bad launch configurations without synchronising
#define MS_LAUNCH_CHECK() MS_HIP(hipGetLastError())

constexpr float kAlphaThreshold = 1.0f / 255.0f;
constexpr float kMaxAlpha = 0.999f;
constexpr float kTransmittanceStop = 1e-4f;

// Lazy sorting (binning.hip): tiles longer than front_threshold have only front_count[tile] sorted
// entries; the rasteriser appends a tile to redo_list when pixels are still alive at the end of it.
struct LazyLists {
    const int32_t *front_count;
    int32_t *redo_flag, *redo_list, *redo_count;
    int front_threshold;
    const uint64_t *keys;   // the scatter's unsorted (depth_bits << 32 | id) keys, for the clean-up pass
};
void isect_lazy_arrays(void *workspace, int64_t N, int tile_w, int tile_h, LazyLists *out);

// rasterize.hip: ms_rasterize_to_pixels_3dgs_fwd with a separate density hint (ms_render_fwd's
// sync-free frames pass the buffer capacity as M and the previous frame's M as the hint)
int rasterize_fwd(int64_t N, int64_t M, int64_t density_hint, const float *means2d, const float *conics,
                  const void *colors, int color_dtype, int CDIM, const float *opacities, const float *backgrounds,
                  int W, int H, int tile_size, int tile_row_begin, int tile_row_end, const int32_t *tile_ranges,
                  const int32_t *flatten_ids, float *render_colors, float *render_alphas, int32_t *last_ids,
                  const LazyLists *lazy, void *after_raster_event, void *stream);

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace ms
