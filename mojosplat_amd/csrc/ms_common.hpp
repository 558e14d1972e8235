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
// The forward rasteriser's select without a compare (rasterize.hip is compiled with fp32 denormals flushed):
// kFlushK = 255 * 2^-126 is the smallest float k with fl(1/255) * k >= 2^-126, so alpha * k is a normal number
// iff alpha >= fl(1/255) and +0 otherwise.
constexpr float kFlushK = 0x1.fep-119f;
// alpha * kFlushK is never scaled back: the rasteriser carries the transmittance as S = T * kTScale, so that
// (alpha kFlushK) S = 255 alpha T, and divides by 255 (kInv255 = fl(1/255)) where it needs alpha T itself.
constexpr float kTScale = 0x1p+126f, kTUnscale = 0x1p-126f, kInv255 = kAlphaThreshold;
constexpr float kAlphaOfV = 0x1.010102p+118f;   // 2^126 / 255: alpha S from v = 255 alpha T
static_assert((double)kAlphaThreshold * (double)kFlushK >= 0x1p-126 &&
                  (double)(kAlphaThreshold * (1.0f - 0x1p-24f)) * (double)kFlushK < 0x1p-126,
              "flush select: the threshold must sit exactly on the smallest normal number");

// The rasteriser's staged record of one Gaussian (3 channels), as the fused frame's projection kernel
// writes it once per Gaussian and as the rasteriser's own staging computes it from the per-stage arrays:
//   a = (mean.x, mean.y, a', b')   a' = -log2(e)/2 * conic.a, b' = -log2(e) * conic.b
//   b = (c', log2(opacity), r, g)  c' = -log2(e)/2 * conic.c
//   c = (b, s, n_c, n_a)           s = log2(255 o) with slack (+inf: no bound, evaluate everywhere and take the
//                                  generic blend loop; -inf: opacity < 1/255, reaches nothing),
//                                  n_c = -conic.b / conic.c, n_a = -conic.b / conic.a
// log2(alpha)(d) = a' dx^2 + b' dx dy + c' dy^2 + log2(o); a quad can blend the Gaussian iff the minimum of
// -(a' dx^2 + b' dx dy + c' dy^2) over its pixel-centre rectangle is <= s.
struct RasterRecord {
    float4 a, b, c;
};
#ifdef __HIPCC__
__device__ __forceinline__ RasterRecord make_raster_record(float mx, float my, float ca, float cb, float cc,
                                                           float op, float r, float g, float b) {
    constexpr float kLog2e = 1.4426950408889634f;
    RasterRecord o;
    o.a = make_float4(mx, my, -0.5f * kLog2e * ca, -kLog2e * cb);
    o.b = make_float4(-0.5f * kLog2e * cc, __log2f(op), r, g);
    const float det = ca * cc - cb * cb;
    const bool bounded = det > 0.f && ca > 0.f && cc > 0.f && op <= kMaxAlpha;
    // (approximate reciprocals / log: their 1e-7 relative error is far inside the slack)
    // opacity below 1/255 can never blend: -inf, no quad is reached
    const float s = !(op >= kAlphaThreshold) ? -__builtin_huge_valf()
                    : bounded ? (__log2f(op * 255.0f) * 1.0001f + 1.5e-4f) : __builtin_huge_valf();
    o.c = make_float4(b, s, bounded ? -cb * __builtin_amdgcn_rcpf(cc) : 0.f, bounded ? -cb * __builtin_amdgcn_rcpf(ca) : 0.f);
    return o;
}
#endif

// binning.hip: ms_project_isect_count that also writes the rasteriser's records (raster_records: N x 48 B,
// or null; colors3: the frame's colours, 3 channels, f32 or f16)
int project_isect_count(int64_t N, const float *means3d, const float *scales, int scales_are_log,
                        const float *quats, const float *opacities, const float *viewmat, float fx, float fy,
                        float cx, float cy, int W, int H, float eps2d, float near_plane, float far_plane,
                        float radius_clip, int tile_size, int row_begin, int row_end, int tight, float *means2d,
                        float *conics, float *depths, int32_t *radii, void *workspace, size_t workspace_bytes,
                        int32_t *tile_ranges, int64_t *isect_info, int64_t *isect_info_mirror,
                        const void *colors3, int color_dtype, void *raster_records, void *stream, uint32_t cut_stamp = 0,
                        // a PREPARED scene (ms_scene_prepare): bounds of every block of block_size Gaussians, for the band pre-cull
                        const float *block_bounds = nullptr, int block_size = 0);

// Lazy sorting (binning.hip): tiles longer than front_threshold have only front_count[tile] sorted
// entries; the rasteriser appends a tile to redo_list when pixels are still alive at the end of it.
// what a depth-cut frame's clean-up launches need to bring a dropped Gaussian's record back (host-side: far_regen)
struct CutInputs {
    const float *means3d, *scales, *quats, *opacities, *viewmat;
    const void *colors;
    int color_f16;
    float fx, fy, cx, cy;
    int W, H;
    float eps2d, near_plane, far_plane;
    int scales_are_log;
    void *records;
    int tile_size;
    // a band frame: its rows of the binning grid, and -- pre-culled (band_cull) -- the isect workspace that holds its candidate list
    int row_begin, row_end;
    int64_t N;
    const void *isect_workspace;
    int band_cull;
};
struct LazyLists {
    const int32_t *front_count;
    int32_t *redo_flag, *redo_list, *redo_count;
    int front_threshold;
    const uint64_t *keys;   // the scatter's unsorted (depth_bits << 32 | id) keys, for the clean-up pass
    // split frames (rasterize_fwd_split): the lists above belong to 32-px bins; the rasteriser walks
    // per-block lists cut from them and flags the BIN (bin_more: its list was longer than its front)
    const int32_t *bin_more;
    int bin_w;
    int packed;             // key / list words are id << 4 | block bits
    int row_lo, row_hi;     // split frames: the band in 16-px block rows (the clean-up leaves other rows alone)
    int redo_grid;          // workgroups of the clean-up launch (256: an empty launch costs the frame the same whatever their number)
    int redo_sort;          // != 0: the clean-up takes two launches (rasterize.hip, k_redo_sort): stranded bins sorted whole, then a workgroup per block
    // depth-cut frame (cut_stamp != 0): the tiles marked has_far[tile] == cut_stamp own pairs that were never written
    // (depth bits > tau[tile]); the clean-up launches regenerate them for the tiles on the redo list from the frame's
    // 12-byte box records (`lean`, n_lean of them) into the free tail of the key array behind the cut_words[0] entries
    // of the lists, one segment per tile: far_cnt[tile] keys from far_start[tile] on (far_cur: the scatter's cursors).
    uint32_t cut_stamp;
    const uint32_t *tau;    // this frame's cut-offs (isect_lazy_arrays: buffer 0 of the two, T words each)
    uint32_t *tau_next;     // the next frame's, as the sort launch left them: the clean-up launch resets the tiles it redoes
    const uint32_t *has_far;
    const void *lean;
    int64_t n_lean;
    int64_t *cut_words;     // the device record's words 8.. : near pairs, far pairs, log cursor
    uint64_t *log_keys;     // == keys, writable
    uint32_t *far_cnt, *far_cur, *far_start;
    const CutInputs *cut_inputs;   // (a HOST pointer, read when the clean-up launches are enqueued)
    // DEFERRED clean-up (a band frame in flight behind another: ms_render_band_begin / _finish, pipeline.hip): non-null = a
    // device-visible address of a word of the lane's pinned record.  The rasteriser stores 1 there when it puts a bin on the
    // redo list, its launch is NOT followed by the clean-up launches, and the finishing half -- which has waited for the
    // rasteriser -- enqueues them only if the word is set (rasterize_deferred_cleanup).
    int32_t *verdict;
};
// rasterize.hip: the clean-up launches a frame enqueued with lazy.verdict set left out, on `stream`; `key` = that pointer
int rasterize_deferred_cleanup(const void *key, void *stream);
int far_regen(const LazyLists &lazy, int tw, int n_tiles, int64_t cap, void *stream);
// project_bwd.hip: the backward projection straight from the backward rasteriser's packed gradient rows
int project_bwd_from_rows(int64_t N, const float *means3d, const float *scales, int scales_are_log, const float *quats,
                          const float *viewmat, float fx, float fy, float cx, float cy, int W, int H, float eps2d,
                          const int32_t *radii, const float *rows, int CDIM, float *v_means3d, float *v_scales,
                          float *v_quats, float *v_colors, float *v_opacities, void *stream,
                          const float *raw_rows_opacities = nullptr);   // non-null: rasterize_bwdq.hip's raw sums (radii may be null)
void isect_lazy_arrays(void *workspace, int64_t N, int tile_w, int tile_h, LazyLists *out);
bool depth_cut_fits(int64_t N, int tile_w, int tile_h);
// lazily sorted fronts: depth (entries) of a front, LDS room for it, depth buckets from the camera planes
struct FrontParams {
    uint32_t fixed_min;
    int fixed_shift, front_k, front_cap;
};
FrontParams front_params(int tile_size, int lazy, float depth_near, float depth_far, bool merged, bool split);
const int32_t *isect_order_array(const void *workspace, int64_t N, int tile_w, int tile_h);

// rasterize.hip: ms_rasterize_to_pixels_3dgs_fwd with a separate density hint (ms_render_fwd's
// sync-free frames pass the buffer capacity as M and the previous frame's M as the hint).
// records: N ready-made RasterRecords (3 channels) or null; order: the band's tiles of the BINNING grid,
// heaviest list first (isect_order_array; for a split frame these are its 32-px bins), or null;
// clip_row16_begin / _end: only the 16x16 blocks of these 16-px rows are rasterised (-1: all of the band's tiles)
int rasterize_fwd(int64_t N, int64_t M, int64_t density_hint, const float *means2d, const float *conics,
                  const void *colors, int color_dtype, int CDIM, const float *opacities, const float *backgrounds,
                  int W, int H, int tile_size, int tile_row_begin, int tile_row_end, const int32_t *tile_ranges,
                  const int32_t *flatten_ids, float *render_colors, float *render_alphas, int32_t *last_ids,
                  const LazyLists *lazy, const void *records, const int32_t *order, int clip_row16_begin,
                  int clip_row16_end, void *after_raster_event, void *stream,
                  // a differentiable frame: every 8x8 quad leaves the Gaussians that passed its reach test, in list order, for the
                  // backward rasteriser -- quad_lists i32[M * quads per tile], quad_counts i32[tiles * quads per tile] (or null)
                  int32_t *quad_lists = nullptr, int32_t *quad_counts = nullptr,
                  // round 6: zero_bytes of memory every wave of the launch zeroes a slice of on its way (a differentiable frame's
                  // rows of raw gradient sums: the kernel is issue-bound and moves 70 MB in 90 us -- the 64 N bytes ride along
                  // instead of costing the backward a 10-us memset); null: nothing
                  void *zero_mem = nullptr, size_t zero_bytes = 0);

// rasterize_bwd.hip: ms_rasterize_to_pixels_3dgs_bwd with the forward frame's ready-made records (or null) and its
// heaviest-first order of the image's tiles (16-px tiles only; or null: the kernel's own counting sort)
int rasterize_bwd(int64_t N, int64_t M, const float *means2d, const float *conics, const float *colors, int CDIM,
                  const float *opacities, const float *backgrounds, int W, int H, int tile_size,
                  const int32_t *tile_ranges, const int32_t *flatten_ids, const float *render_alphas,
                  const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas, float *v_means2d,
                  float *v_conics, float *v_colors, float *v_opacities, void *workspace, size_t workspace_bytes,
                  int overwrite, const void *records, const int32_t *block_order, void *stream);

// rasterize_bwdq.hip: the backward rasteriser of a 3-channel differentiable frame -- one wave per 8x8 quad walking the
// frame's own lists front to back (fully sorted, or lazily sorted fronts: front_count / front_threshold as the forward
// took them), per-entry sums on the matrix pipe, raw sums into 64-byte rows (zeroed by the caller; finished by
// project_bwd_from_rows with raw_rows_opacities).  ids: Gaussian index per entry, id_stride ints apart; skip_flag: tiles
// to leave alone (or null); order: the binning grid's tiles heaviest first (or null).
int rasterize_bwd_quads(int64_t N, int64_t M, const void *records, const float *backgrounds, int W, int H, int tile_size,
                        const int32_t *tile_ranges, const int32_t *ids, int id_stride, const int32_t *front_count,
                        int front_threshold, const int32_t *skip_flag, const float *render_colors,
                        const float *render_alphas, const float *v_render_colors, const float *v_render_alphas,
                        float *packed_rows, const int32_t *order, void *stream,
                        int tile_row_begin = 0, int tile_row_end = -1,
                        const int32_t *quad_lists = nullptr, const int32_t *quad_counts = nullptr);   // (a band of tile rows; -1: to the last row | the forward's per-quad lists: rasterize_fwd)

// ... and its launch for the tiles the forward's clean-up pass redid (their sorted front ran out with pixels alive): a wave
// per (tile, block, quad) walks the whole-tile sorted ids the forward's k_redo_sort left (redo_flag[tile] == 2); a tile that
// was not sorted that way gets its keys sorted in place by one workgroup first.  Empty on almost every frame.
int rasterize_bwd_redo(int64_t N, int64_t M, const void *records, const float *backgrounds, int W, int H, int tile_size,
                       const int32_t *tile_ranges, uint64_t *keys, const int32_t *ids, const int32_t *redo_list,
                       const int32_t *redo_count, const int32_t *redo_flag, const float *render_colors,
                       const float *render_alphas, const float *v_render_colors, const float *v_render_alphas,
                       float *packed_rows, void *stream, int32_t *count_mirror = nullptr);

// Block lists of a split frame (ms_render_fwd): what the sort kernels of 32-px bins write instead of
// flatten_ids.  Bin `b` with list [start, start + n) owns block_ids[4 start, 4 (start + n)): its block q
// (2x2 per bin, q = 2 dy + dx) gets [4 start + q n, ... + count_q), and block_ranges holds that pair for
// every 16x16 block of the image; bin_more[b] = the list was longer than its sorted front.
struct BlockLists {
    int32_t *block_ranges, *block_ids, *bin_more;
    int tw16, th16;
};
// Sync-free frames (ms_render_fwd): the scans' total pass -- tile_ranges, work lists, launch order, the size record --
// rides in the scatter launch instead of being a serial pass of its own (binning.hip, deferred_total), and the
// caller's hand-off event is recorded right behind that launch.
struct DeferredTotal {
    int band_only;
    int64_t *info, *info_mirror;
    void *sync_event;
    uint32_t cut_stamp;   // != 0: a depth-cut frame; the value its tiles with dropped pairs are marked with (unique per frame)
};
// bits of the `tight` flags that only ms_render_fwd sets (the C entry points mask them off)
constexpr int kTightLean = 64, kTightDeferTotal = 128, kTightDepthCutBuf = 512, kTightKeepArrays = 1024;   // (bit 10: a lean frame writes the projected arrays too)
constexpr int kTightClaimed = 2048;   // (bit 11, round 6: the count pass claims its rows -- binning.hip, k_project_hist's tile_total; every emit of the frame must carry it too)
//   // (bit 9: a depth-cut frame reads its cut-offs from buffer 1)
// DEPTH CUT (sync-free lean frames on plain bins; binning.hip, k_project_hist): pairs behind their tile's cut-off
// are counted but never written; a tile that outlives its list gets them back from the clean-up launch.
// (project_isect_count: cut_stamp; the emit and the rasteriser: DeferredTotal::cut_stamp / LazyLists::cut_stamp)
// binning.hip: ms_isect_tiles_emit_speculative / ms_isect_tiles_emit with the internal flag bits honoured
int isect_emit_speculative(int64_t N, const float *means2d, const int32_t *radii, const float *depths, int tile_size,
                           int tile_w, int tile_h, int row_begin, int row_end, void *workspace, size_t workspace_bytes,
                           const int32_t *tile_ranges, const int64_t *isect_info_dev, int64_t capacity,
                           const int64_t *prev_info_host, int tight, int lazy, float depth_near, float depth_far,
                           uint64_t *sort_keys, int32_t *flatten_ids, const DeferredTotal *defer, void *stream);
int isect_emit_exact(int64_t N, const float *means2d, const int32_t *radii, const float *depths, int tile_size,
                     int tile_w, int tile_h, int row_begin, int row_end, void *workspace, size_t workspace_bytes,
                     const int32_t *tile_ranges, const int64_t *host_info, int tight, int lazy, float depth_near,
                     float depth_far, uint64_t *sort_keys, uint64_t *sort_tmp, int32_t *flatten_ids, void *stream);

// binning.hip: scatter + sorts on the bin grid, writing block lists (info_dev != null: sync-free frame
// against `cap` entries, host_info = the previous frame's record or null)
int isect_emit_bins(int64_t N, const float *means2d, const int32_t *radii, const float *depths, int bin_w, int bin_h,
                    int row_begin, int row_end, void *workspace, size_t workspace_bytes, const int32_t *bin_ranges,
                    const int64_t *host_info, const int64_t *info_dev, int64_t cap, int flags, int lazy, float depth_near,
                    float depth_far, uint64_t *sort_keys, const BlockLists *out, const DeferredTotal *defer, void *stream);

// Split frame: the block lists cut from 32-px bins are rasterised per 16x16 block and cleaned up per
// bin.  Rows are BLOCK rows.
int rasterize_fwd_split(int64_t N, int64_t cap, int64_t density_hint, const float *means2d, const float *conics,
                        const void *colors, int color_dtype, int CDIM, const float *opacities,
                        const float *backgrounds, int W, int H, int block_row_begin, int block_row_end,
                        const int32_t *bin_ranges, const BlockLists *lists, float *render_colors,
                        const LazyLists *lazy, const void *records, const int32_t *order, void *after_raster_event,
                        void *stream);

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace ms
