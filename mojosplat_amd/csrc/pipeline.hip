// ms_render_fwd: the whole device side of render_gaussians (reference mojosplat/render.py:63-101:
// project -> bin/sort -> rasterise) behind ONE C call, so a frame costs one host->library
// transition instead of five and no Python runs between the kernels.
//
// The only host<->device hand-off is the intersection count M (and the per-class tile counts)
// between the counting and the emitting half of the binning stage -- the same place
// gsplat.isect_tiles has its own (reference call site mojosplat/binning.py:73-82).  Everything
// else is enqueued back to back on the caller's stream.
#include <stdlib.h>

#include "ms_common.hpp"
#include <atomic>

#ifdef MS_DIAG
#include <chrono>
#include <stdio.h>
namespace {
// MS_HOST_PROF=1 (diagnostic build): where the HOST time of ms_render_fwd's enqueueing half goes, printed every 1000 frames
struct HostProf {
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long frames = 0;
    bool on = getenv("MS_HOST_PROF") != nullptr;
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void frame_done() {
        if (on && ++frames % 1000 == 0) {
            fprintf(stderr, "[host prof] per frame, us: setup %.2f | mirror lookup %.2f | count launches %.2f | events %.2f | emit launches %.2f | raster launches %.2f | total %.2f\n",
                    t[0] / 1000, t[1] / 1000, t[2] / 1000, t[3] / 1000, t[4] / 1000, t[5] / 1000, t[6] / 1000);
            for (double &x : t) x = 0;
        }
    }
};
thread_local HostProf g_hp;
}  // namespace
#define MS_HP_T(var) const double var = g_hp.on ? HostProf::now() : 0.0
#define MS_HP_ADD(slot, a, b) do { if (g_hp.on) g_hp.t[slot] += (b) - (a); } while (0)
#else
#define MS_HP_T(var) do {} while (0)
#define MS_HP_ADD(slot, a, b) do {} while (0)
#endif

namespace {

struct WsLayout {
    size_t off_isect, isect_bytes, off_means2d, off_conics, off_depths, off_radii, off_ranges, off_info, off_bin_ranges,
        off_bin_more, off_records, off_quad_counts, off_rows, total;
};

WsLayout ws_layout(int64_t N, int tw, int th) {
    WsLayout L;
    size_t o = 0;
    const size_t n = (size_t)(N > 0 ? N : 1), T = (size_t)tw * th;
    L.off_isect = o;   L.isect_bytes = ms_isect_workspace_bytes(N, tw, th); o += ms::align_up(L.isect_bytes, 256);
    L.off_means2d = o; o += ms::align_up(n * 8, 256);
    L.off_conics = o;  o += ms::align_up(n * 12, 256);
    L.off_depths = o;  o += ms::align_up(n * 4, 256);
    L.off_radii = o;   o += ms::align_up(n * 8, 256);
    L.off_ranges = o;  o += ms::align_up(T * 8, 256);
    L.off_info = o;    o += 256;
    L.off_bin_ranges = o; o += ms::align_up(T * 8, 256);   // split frames: ranges / flags of the 32-px bins
    L.off_bin_more = o;   o += ms::align_up(T * 4, 256);
    L.off_records = o;    o += ms::align_up(n * sizeof(ms::RasterRecord), 256);   // the rasteriser's ready-made records
    L.off_quad_counts = o; o += ms::align_up(T * 64 * 4, 256);   // a differentiable frame: entries of every 8x8 quad's list (<= 64 quads a tile: bins of 64 px)
    // round 6: a differentiable frame's rows of raw gradient sums (ms_render_bwd_rows: f32[N][16]) -- the LAST region, so that a
    // caller finds it at ms_render_workspace_bytes() - ms_render_bwd_rows_bytes(N): the frame's rasteriser zeroes it on its way
    // (bit 15 of the record's flag word says so) and ms_render_bwd_rows, handed exactly this address, skips its memset
    L.off_rows = o; o += ms::align_up(n * 16 * sizeof(float), 256);
    L.total = o;
    return L;
}

}  // namespace

// MOJOSPLAT_LAZY_SORT=0 switches lazy sorting off (full per-tile sorts, as the per-stage API does)
static int ms_lazy_enabled() {
    static const int v = [] {
        const char *e = getenv("MOJOSPLAT_LAZY_SORT");
        return e ? atoi(e) != 0 : 1;
    }();
    return v;
}

// A differentiable frame's per-quad lists (rasterize.hip, RasterArgs::quad_lists): MOJOSPLAT_BWD_LISTS=0 switches them off (the
// backward then tests and compacts every tile's list per quad again).  Quads per tile the frame keeps lists for: 4, 16 or
// -- round 6 -- 64 (tiles of 16 / 32 / 64 px).  64 four-byte slots per pair are address space more than memory: a slot is
// written only where a quad is reached (~1.5 of a pair's 64), the rest of the 268 bytes a pair reserves is never touched; what
// it costs is the caching allocator's block (config 5 on 64-px bins: 3.3 GB of the GPU's 288 a frame in flight) --
// MOJOSPLAT_BWD_LISTS_MAX_NQ=16 keeps rounds 5's limit.
static int ms_bwd_lists_enabled() {
    static const int v = [] {
        const char *e = getenv("MOJOSPLAT_BWD_LISTS");
        return e ? atoi(e) != 0 : 1;
    }();
    return v;
}
static int quad_list_nq(int tile_size) {
    if (tile_size <= 0 || tile_size % 16 != 0) return 0;
    const int nsub = (tile_size / 16) * (tile_size / 16);
    static const int max_nq = [] {
        const char *e = getenv("MOJOSPLAT_BWD_LISTS_MAX_NQ");
        const int v = e ? atoi(e) : 64;
        return v == 4 || v == 16 ? v : 64;
    }();
    return 4 * nsub <= max_nq ? 4 * nsub : 0;
}

// Claimed rows (round 6): the count kernel claims its rows with returning atomics, per XCD, and the prefix kernel
// (k_tile_scan_wg) is not launched -- config 3's frame -2.5 us, config 4's -5 (profiles/r06_claimed_rows.md).
// MOJOSPLAT_CLAIMED_ROWS=0: the prefix kernel, as rounds 1-5.
static int ms_bwd_zero_rows_enabled() {
    static const int v = [] {
        const char *e = getenv("MOJOSPLAT_BWD_ZERO_ROWS");
        return e && e[0] == '0' ? 0 : 1;
    }();
    return v;
}

static int ms_claimed_rows_enabled() {
    static const int v = [] {
        const char *e = getenv("MOJOSPLAT_CLAIMED_ROWS");
        return e && e[0] == '0' ? 0 : 1;
    }();
    return v;
}

static int ms_band_cull_enabled() {
    static const int v = [] {
        const char *e = getenv("MOJOSPLAT_BAND_CULL");
        return e ? atoi(e) != 0 : 1;
    }();
    return v;
}

static int ms_order_enabled() {
    static const int v = [] {
        const char *e = getenv("MOJOSPLAT_RASTER_ORDER");
        return e ? atoi(e) != 0 : 1;
    }();
    return v;
}

// Depth cut-offs (binning.hip, k_project_hist): mode 0 = never drop the pairs behind a bin's cut-off, 1 = on frames whose
// predecessor held at least `min_pairs` pairs (6 M), 2 = on every frame that can (measurements, tests).  Read ONCE from
// MOJOSPLAT_DEPTH_CUT / MOJOSPLAT_DEPTH_CUT_MIN_PAIRS; tests and measurements that switch it inside one process call
// ms_config_depth_cut (round 3 called getenv() on every frame, from any host thread, while Python wrote os.environ:
// undefined behaviour in libc).
namespace {
std::atomic<int> g_cut_mode{-1};
std::atomic<long long> g_cut_min_pairs{-1};
void cut_config_init() {
    if (g_cut_mode.load(std::memory_order_acquire) >= 0) return;
    static std::atomic<int> once{0};
    int expected = 0;
    if (once.compare_exchange_strong(expected, 1)) {
        const char *e = getenv("MOJOSPLAT_DEPTH_CUT_MIN_PAIRS");
        const long long n = e ? atoll(e) : -1;
        g_cut_min_pairs.store(n >= 0 ? n : 6000000ll, std::memory_order_relaxed);
        const char *m = getenv("MOJOSPLAT_DEPTH_CUT");
        const int mode = m ? atoi(m) : 1;
        g_cut_mode.store(mode < 0 ? 0 : mode, std::memory_order_release);
    } else {
        while (g_cut_mode.load(std::memory_order_acquire) < 0) {}
    }
}
}  // namespace
static int ms_depth_cut_mode() { cut_config_init(); return g_cut_mode.load(std::memory_order_relaxed); }
static int64_t ms_depth_cut_min_pairs() { cut_config_init(); return (int64_t)g_cut_min_pairs.load(std::memory_order_relaxed); }
extern "C" int ms_config_depth_cut(int mode, long long min_pairs) {
    cut_config_init();
    if (mode >= 0) g_cut_mode.store(mode, std::memory_order_relaxed);
    if (min_pairs >= 0) g_cut_min_pairs.store(min_pairs, std::memory_order_relaxed);
    return MS_OK;
}
// workgroups of the clean-up launch: 256, always.  (Rounds 1-2 launched ONE while recent frames had needed no clean-up;
// measured in round 3, the frame costs the same with 1 ... 256 of them -- 0.1678-0.1681 ms at config 3: what costs is the
// kernel boundary -- while the first frames that DO need the pass ran their hundreds of bins through one workgroup: 7-14 s
// a frame at configs 4 / 5 after a scene swap.  Round 4: 1024 change nothing either -- 70 against 72 ms for that frame: its
// time is the heaviest bin's sixteen blocks, one after the other in one workgroup.)
static int ms_redo_grid() { return 256; }
// MOJOSPLAT_REDO_SORT=0: the clean-up pass in ONE launch always (measurements: the round-3 behaviour of the frames that
// can expect stranded bins -- rasterize.hip, k_redo_sort)
static int ms_redo_sort_enabled() {
    static const int v = [] {
        const char *e = getenv("MOJOSPLAT_REDO_SORT");
        return e ? atoi(e) != 0 : 1;
    }();
    return v;
}
static int ms_merged_sort_enabled() {   // (binning.hip reads the same variable)
    static const int v = [] {
        const char *e = getenv("MOJOSPLAT_MERGED_SORT");
        return e ? atoi(e) != 0 : 1;
    }();
    return v;
}

// MOJOSPLAT_SPLIT=0: bin on the rasteriser's own 16-px tiles instead of 32-px bins cut into block lists
static int ms_split_enabled() {
    static const int v = [] {
        const char *e = getenv("MOJOSPLAT_SPLIT");
        return e ? atoi(e) != 0 : 1;
    }();
    return v;
}

// most bin entries a split frame may hold (4 block-list slots each must fit int32 offsets); the
// environment can lower it so that tests reach the restart on 16-px tiles
static int64_t ms_split_max_entries() {
    static const int64_t v = [] {
        const char *e = getenv("MOJOSPLAT_SPLIT_MAX_ENTRIES");
        const long long n = e ? atoll(e) : 0;
        return (int64_t)(n > 0 && n < 0x1fffffffll ? n : 0x1fffffffll);
    }();
    return v;
}

extern "C" size_t ms_render_workspace_bytes(int64_t N, int tile_w, int tile_h) {
    if (tile_w <= 0 || tile_h <= 0 || (int64_t)tile_w * tile_h >= (1ll << 30)) return 0;
    return ws_layout(N, tile_w, tile_h).total;
}

extern "C" int ms_render_workspace_layout(int64_t N, int tile_w, int tile_h, size_t *offsets) {
    MS_REQUIRE(offsets && tile_w > 0 && tile_h > 0 && (int64_t)tile_w * tile_h < (1ll << 30), MS_ERR_INVALID_ARG,
               "render_workspace_layout: bad argument");
    const WsLayout L = ws_layout(N, tile_w, tile_h);
    offsets[0] = L.off_means2d; offsets[1] = L.off_conics; offsets[2] = L.off_depths;
    offsets[3] = L.off_radii;   offsets[4] = L.off_ranges; offsets[5] = L.total;
    return MS_OK;
}

// How many bins the clean-up pass of the frame that `workspace` belongs to had to redo: host_counts[0] = all of them,
// [1] = those redone for their depth cut-off.  An asynchronous 8-byte copy on `stream` (host_counts: pinned memory; read
// it once an event recorded behind this call has completed).  A frame on cached scratch learns this from the NEXT frame's
// size record (host_info[5]); a differentiable frame owns fresh scratch, so its caller asks here, behind the backward.
extern "C" int ms_render_redo_counts(const void *workspace, size_t workspace_bytes, int64_t N, int tile_w, int tile_h,
                                     int32_t *host_counts, void *stream) {
    MS_REQUIRE(workspace && host_counts && tile_w > 0 && tile_h > 0 && (int64_t)tile_w * tile_h < (1ll << 30) && N >= 0,
               MS_ERR_INVALID_ARG, "render_redo_counts: bad argument");
    const WsLayout L = ws_layout(N, tile_w, tile_h);
    MS_REQUIRE(workspace_bytes >= L.total, MS_ERR_WORKSPACE, "render_redo_counts: workspace %zu < %zu", workspace_bytes, L.total);
    ms::LazyLists ll{};
    ms::isect_lazy_arrays(const_cast<char *>((const char *)workspace) + L.off_isect, N, tile_w, tile_h, &ll);
    MS_HIP(hipMemcpyAsync(host_counts, ll.redo_count, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MS_OK;
}

extern "C" size_t ms_render_isect_bytes(int64_t M, int with_merge_scratch) {
    const size_t m = (size_t)(M > 0 ? M : 1);
    return ms::align_up(m * 8, 256) * (with_merge_scratch ? 2 : 1) + ms::align_up(m * 4, 256);
}

// split frames: keys 8 M | list words 4 M | block lists 16 M
static size_t split_isect_bytes(int64_t M) {
    const size_t m = (size_t)(M > 0 ? M : 1);
    return ms::align_up(m * 8, 256) + ms::align_up(m * 4, 256) + ms::align_up(m * 16, 256);
}

static int render_fwd_impl(const ms_scene *prepared, int restart, int64_t N, const float *means3d, const float *scales, int scales_are_log,
                             const float *quats, const float *opacities, const void *colors,
                             int color_dtype, int CDIM, const float *viewmat, float fx, float fy,
                             float cx, float cy, int W, int H, float eps2d, float near_plane,
                             float far_plane, int tile_size, int tile_row_begin, int tile_row_end,
                             const float *backgrounds, void *workspace, size_t workspace_bytes, void *isect_buf,
                             size_t isect_bytes, int64_t *host_info, int resume,
                             float *render_colors, float *render_alphas, int32_t *last_ids,
                             void **stage_events, void *sync_event, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MS_HP_T(hp_t0);
    // restart: bit 0 = this frame again without the split, bit 1 = without the depth cut
    int no_split = restart & 1;
    const int no_cut = (restart >> 1) & 1;
    const int phase = resume & 0xff;
    const bool defer_cleanup = (resume & MS_RENDER_DEFER_CLEANUP) != 0;   // (the band pair only: its record has the sixteen words)
    MS_REQUIRE(phase >= MS_RENDER_WHOLE && phase <= MS_RENDER_FINISH, MS_ERR_INVALID_ARG, "render_fwd: bad phase %d",
               phase);
    MS_REQUIRE(N >= 0 && W > 0 && H > 0 && tile_size > 0, MS_ERR_INVALID_ARG, "render_fwd: bad sizes");
    MS_REQUIRE(workspace && host_info && render_colors, MS_ERR_INVALID_ARG, "render_fwd: null pointer");
    const int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
    // MS_RENDER_ROWS16: the band is given in rows of 16 pixels; bin the tile rows that cover it, rasterise it alone
    const bool rows16 = (resume & MS_RENDER_ROWS16) != 0 && tile_size != 16;
    MS_REQUIRE(!rows16 || tile_size % 16 == 0, MS_ERR_INVALID_ARG, "render_fwd: MS_RENDER_ROWS16 needs a tile size that is a multiple of 16");
    const int k16 = rows16 ? tile_size / 16 : 1, th16 = (H + 15) / 16;
    int clip0 = -1, clip1 = -1;   // 16-px rows the rasteriser keeps (-1: all of the band's tiles)
    if (rows16) {
        MS_REQUIRE(tile_row_begin >= 0 && tile_row_begin <= tile_row_end && tile_row_end <= th16, MS_ERR_INVALID_ARG,
                   "render_fwd: bad band of 16-px rows [%d,%d) of %d", tile_row_begin, tile_row_end, th16);
        clip0 = tile_row_begin; clip1 = tile_row_end;
        tile_row_begin = clip0 / k16;
        tile_row_end = clip1 > clip0 ? (clip1 + k16 - 1) / k16 : tile_row_begin;
    }
    MS_REQUIRE(tile_row_begin >= 0 && tile_row_begin <= tile_row_end && tile_row_end <= th, MS_ERR_INVALID_ARG,
               "render_fwd: bad tile row band [%d,%d) of %d", tile_row_begin, tile_row_end, th);
    const int r0 = tile_row_begin, r1 = tile_row_end;
    // (bit 4 of host_info[7]: this frame was restarted without the split, see below; its redo must agree)
    if (phase == MS_RENDER_RESUME && (host_info[7] & 16)) no_split = 1;
    // (bit 6: the frame dropped the pairs behind its bins' depth cut-offs -- the exact path cannot finish such a frame:
    // it starts over without the cut)
    if (phase == MS_RENDER_RESUME && (host_info[7] & 64))
        return render_fwd_impl(prepared, restart | 2, N, means3d, scales, scales_are_log, quats, opacities, colors, color_dtype, CDIM, viewmat,
                               fx, fy, cx, cy, W, H, eps2d, near_plane, far_plane, tile_size, rows16 ? clip0 : tile_row_begin,
                               rows16 ? clip1 : tile_row_end, backgrounds, workspace, workspace_bytes, isect_buf, isect_bytes, host_info,
                               (resume & ~0xff) | MS_RENDER_WHOLE, render_colors, render_alphas, last_ids, stage_events,
                               sync_event, stream_);
    const WsLayout L = ws_layout(N, tw, th);
    MS_REQUIRE(workspace_bytes >= L.total, MS_ERR_WORKSPACE, "render_fwd: workspace %zu < %zu", workspace_bytes,
               L.total);
    char *ws = (char *)workspace;
    float *means2d = (float *)(ws + L.off_means2d), *conics = (float *)(ws + L.off_conics);
    float *depths = (float *)(ws + L.off_depths);
    int32_t *radii = (int32_t *)(ws + L.off_radii), *ranges = (int32_t *)(ws + L.off_ranges);
    int64_t *info = (int64_t *)(ws + L.off_info);
    auto mark = [&](int i) {
        if (stage_events && stage_events[i]) (void)hipEventRecord((hipEvent_t)stage_events[i], stream);
    };

    // host_info[7] belongs to the library between the two halves of a frame: bit 0 = emit +
    // rasterise were enqueued speculatively, bit 1 = the large sort class was among them
    // lazy sorting (binning.hip, k_tile_front): not for a differentiable frame that asks for last_ids -- the older
    // backward rasteriser walks the full lists by position; a frame that only keeps render_alphas (round 4: the
    // quad-wave backward, rasterize_bwdq.hip, walks whatever lists the forward walked, front to back) is lazily sorted
    // like any other
    // (bits 1-2 of `lazy`: the front level the caller asked for, see MS_RENDER_FRONT_LEVEL)
    const int lazy = (last_ids || !opacities || CDIM > 4 || (resume & MS_RENDER_FULL_SORT))
                         ? 0 : (ms_lazy_enabled() ? 1 | (((resume >> 9) & 3) << 1) : 0);
    // split frame (binning.hip, emit_block_lists): bin on 32-px bins, rasterise the 16x16-block lists cut
    // from them (a band that starts or ends inside a bin row bins that whole row).  The frame is the same
    // either way.
    // Thin bands (a rank's share of a frame cut 8 ways) stay on 16-px bins: their cost is the walk over all
    // Gaussians and the largest bin's front, which 32-px bins make longer (1080p, 9 rows: 151 vs 163 us).
    // (not for a differentiable frame: its backward walks the binning grid's own lists)
    const bool split = lazy && tile_size == 16 && N > 0 && N < (1ll << 28) && r1 > r0 && !render_alphas &&
                       (r1 - r0 >= 16 || (r0 == 0 && r1 == th)) && !no_split && ms_split_enabled();
    const int bw = (tw + 1) / 2, bh = (th + 1) / 2, b0 = r0 / 2, b1 = (r1 + 1) / 2;
    // the rasteriser's ready-made records (3-channel forward frames): written by the projection kernel
    const bool use_records = CDIM == 3 && opacities && colors;
    const bool aux_frame = render_alphas || last_ids;   // a differentiable frame: ids must stay Gaussian indices
    // ... and, on the lean path, every 8x8 quad leaves the backward its list (quads per tile; 0: no lists): 4 bytes per
    // (pair, quad of its tile) behind the sorted ids of the intersection buffer, the counts in the workspace
    const int list_nq = (render_alphas && !last_ids && use_records && ((uintptr_t)(ws + L.off_records) & 15) == 0 && ms_bwd_lists_enabled())
                            ? quad_list_nq(tile_size) : 0;
    int32_t *quad_counts = list_nq ? (int32_t *)(ws + L.off_quad_counts) : nullptr;
    // ... and its rasteriser zeroes the backward's rows of raw sums while it is at it (MOJOSPLAT_BWD_ZERO_ROWS=0: the
    // backward's own memset, as rounds 4-5)
    void *zero_rows = (render_alphas && !last_ids && use_records && ms_bwd_zero_rows_enabled()) ? (void *)(ws + L.off_rows) : nullptr;
    const size_t zero_rows_bytes = zero_rows ? (size_t)(N > 0 ? N : 1) * 16 * sizeof(float) : 0;
    void *records = use_records ? (void *)(ws + L.off_records) : nullptr;
    // the rasteriser launches its blocks heaviest list first (the count pass leaves the order of the binning
    // grid's tiles in the isect workspace); MOJOSPLAT_RASTER_ORDER=0: image order interleaved over the XCDs
    const int32_t *order = ms_order_enabled() ? ms::isect_order_array(ws + L.off_isect, N, split ? bw : tw, split ? bh : th)
                                              : nullptr;
    // a band that is a rank's share of a frame (under 60 % of the rows of a scene worth the extra pass): the
    // Gaussians that cannot reach it are culled before the projection (binning.hip, k_band_precull).
    // MOJOSPLAT_BAND_CULL=0 switches that off.
    // (only frames whose rasteriser reads the ready-made records: the lists of a culled band hold POSITIONS in the
    // band's candidate list, which index the workspace's dense projected arrays and records, not the caller's)
    // lean frame: nobody reads the projected arrays (the rasteriser and the clean-up pass stage from the records,
    // the scatter kernel from the 16-byte box + depth records the count kernel leaves instead)
    // (the projected arrays are for the OLDER backward -- last_ids frames -- and the caller's intermediates; the quad-wave
    // backward stages from the records and its backward projection takes the raw sums: nothing reads them)
    const int lean = (use_records && ((uintptr_t)records & 15) == 0)
                         ? ms::kTightLean | (last_ids ? ms::kTightKeepArrays : 0) : 0;
    const int cull = ((use_records && !aux_frame && N >= 32768 && N < (1ll << 28) && 10 * (r1 - r0) < 6 * th && ms_band_cull_enabled()) ? 32 : 0) | lean;
    const int bin_flags = 1 | 2 | 4 | ((tw & 1) ? 8 : 0) | ((th & 1) ? 16 : 0) | cull;
    int32_t *bin_ranges = (int32_t *)(ws + L.off_bin_ranges), *bin_more = (int32_t *)(ws + L.off_bin_more);
    ms::LazyLists lazy_lists{};
    if (lazy) ms::isect_lazy_arrays(ws + L.off_isect, N, split ? bw : tw, split ? bh : th, &lazy_lists);
    // (host_info[5] of the record left by the previous frame: the clean-up count it reported, or the buffer size
    // an exact-path frame asked for -- either way "not a quiet run of frames")
    if (lazy) lazy_lists.redo_grid = ms_redo_grid();
    bool speculated = false;
    if (phase == MS_RENDER_WHOLE || phase == MS_RENDER_BEGIN) {
        mark(0);
        // projection + tile counting share one pass over the Gaussians (k_project_hist)
        int64_t prev[8];  // the previous frame's record: a hint for what this frame will need
        for (int k = 0; k < 8; ++k) prev[k] = host_info[k];
        // the size record reaches the host by a zero-copy store from the scan kernel when host_info
        // is mapped pinned memory (what the header asks for); else by a copy
        void *mirror = nullptr;
        MS_HP_T(hp_t1);
        MS_HP_ADD(0, hp_t0, hp_t1);
        if (hipHostGetDevicePointer(&mirror, host_info, 0) != hipSuccess) {
            (void)hipGetLastError();
            mirror = nullptr;
        }
        MS_HP_T(hp_t2);
        MS_HP_ADD(1, hp_t1, hp_t2);
        // Sync-free frame (see below) -- decided here because such a frame also DEFERS the scans' total pass into
        // its scatter launch (binning.hip, deferred_total): the size record then reaches the host behind that launch
        const int64_t cap = split ? (isect_bytes > 768 ? (int64_t)((isect_bytes - 768) / 28) : 0)
                                  : (isect_bytes > 768 ? (int64_t)((isect_bytes - (list_nq ? 768 : 512)) / (12 + 4 * list_nq)) : 0);
        const bool speculate = sync_event && isect_buf && cap > 0 && N > 0;   // (an empty set has null inputs: exact path, M = 0)
        const bool deferred = speculate && mirror && (split ? b1 > b0 : r1 > r0);
        // claimed rows (round 6, MOJOSPLAT_CLAIMED_ROWS=1; binning.hip, k_project_hist's tile_total): the frame's count pass and
        // EVERY emit of the frame -- the speculative one and an exact redo, in this call or a resumed one: bit 14 of the
        // record's flag word carries it there -- must agree on what a histogram row holds
        const int claim_bit = deferred && ms_claimed_rows_enabled() ? ms::kTightClaimed : 0;
        const int64_t claim_flag = claim_bit ? 16384 : 0;
        // the previous frame on this record (same scratch, same grid) had no tile beyond the small sort class:
        // bet that this one has none either (bit 5 of host_info[7]; checked against the size record below)
        const bool bet_light = !split && lazy && prev[0] > 0 && prev[2] + prev[3] + prev[4] == 0 && !(prev[7] & 4);
        // DEPTH CUT-OFFS (binning.hip): the merged sort launch of a whole-grid sync-free frame leaves, per bin, the
        // depth at which its sorted front ended (bits 7 / 8 of host_info[7]: left, and in which of two buffers); the
        // next frame on this record -- a lean one on plain bins with enough pairs to matter -- counts the pairs
        // behind them but never writes or sorts them (bit 6).  Whatever the cut-offs are, the frame is exact: a bin
        // that outlives its list gets its dropped pairs back in the clean-up launches (rasterize.hip, k_far_regen).
        // (not a differentiable frame: it owns fresh scratch, nobody would read what it left)
        const bool leaves_cutoffs = speculate && !split && lazy && !bet_light && r1 > r0 && ms_merged_sort_enabled() &&
                                    ms_depth_cut_mode() != 0 && !aux_frame;
        const int cut_in = (prev[7] & 128) ? (int)((prev[7] >> 8) & 1) : 0, cut_out = (prev[7] & 128) ? 1 - cut_in : 0;
        // (bits 16-31: the grid those cut-offs belong to -- the record may have served another grid since; a frame whose
        // predecessor had to regenerate the pairs of more than a handful of bins takes no cut: its cut-offs are fresh)
        // (round 4: ... and bits 32-47 the BAND -- a rank's share of a frame keeps cut-offs for its own rows and 16-px clip: the
        // sort launch only writes those, and the fronts they come from were judged by the blocks inside the clip.  A band's
        // frame holds a fraction of the pairs: its threshold is the whole frame's scaled by the rows it renders)
        const int64_t band_sig = (r0 == 0 && r1 == th && clip0 < 0) ? 0 :
            (int64_t)(1 + (((unsigned)r0 * 31u + (unsigned)r1 * 131u + (unsigned)(clip0 + 1) * 521u + (unsigned)(clip1 + 1) * 1031u) % 0xfffeu)) << 32;
        const int64_t cut_grid = ((int64_t)((tw & 0xff) | ((th & 0xff) << 8)) << 16) | band_sig;
        const int64_t prev_cut_redos = (prev[7] & 4) ? 0 : ((prev[5] >> 32) & 0x3fffffffll);
        const int64_t min_pairs = ms_depth_cut_min_pairs() * (int64_t)(r1 - r0) / (th > 0 ? th : 1);
        const bool cut = leaves_cutoffs && deferred && lean && !no_cut && (prev[7] & 128) &&
                         (prev[7] & (0xffffffffll << 16)) == cut_grid && prev_cut_redos <= (tw * (r1 - r0)) / 64 + 4 &&
                         (ms_depth_cut_mode() >= 2 || prev[0] >= min_pairs) && ms::depth_cut_fits(N, tw, th);
        static std::atomic<uint32_t> cut_stamps{0};
        uint32_t cut_stamp = 0;
        if (cut)
            while ((cut_stamp = ++cut_stamps) == 0u) {}
        // DEFERRED clean-up (MS_RENDER_DEFER_CLEANUP: ms_render_band_begin, whose finishing half waits for the rasteriser): up to
        // four launches that are empty on almost every frame are left out; the rasteriser says in word 8 of the pinned record
        // whether they are needed after all.  (Such a frame's size record needs no event of its own: the finishing half waits
        // for the band's end.)
        const bool late_cleanup = defer_cleanup && mirror && phase == MS_RENDER_BEGIN && speculate && !split && lazy && !bet_light;
        const ms::DeferredTotal defer{1, info, (int64_t *)mirror, late_cleanup ? nullptr : sync_event, cut_stamp};
        const int defer_bit = (deferred ? ms::kTightDeferTotal : 0) | (cut && cut_in ? ms::kTightDepthCutBuf : 0) | claim_bit;
        const ms::CutInputs cut_inputs{means3d, scales, quats, opacities, viewmat, colors, color_dtype == MS_COLOR_F16 ? 1 : 0, fx, fy, cx, cy, W, H,
                                       eps2d, near_plane, far_plane, scales_are_log, records, tile_size, r0, r1, N, ws + L.off_isect, (cull & 32) ? 1 : 0};
        const int64_t cut_bits = (cut ? 64 : 0) | (leaves_cutoffs ? (128 | (cut_out << 8) | cut_grid) : 0);
        if (int rc = ms::project_isect_count(N, means3d, scales, scales_are_log, quats, opacities, viewmat, fx, fy,
                                             cx, cy, W, H, eps2d, near_plane, far_plane, 0.0f,
                                             split ? 32 : tile_size, split ? b0 : r0, split ? b1 : r1,
                                             /*tight | ranges for the band only (| block masks)=*/(split ? bin_flags : 1 | 2 | cull) | defer_bit,
                                             means2d, conics, depths, radii, ws + L.off_isect, L.isect_bytes,
                                             split ? bin_ranges : ranges, info, (int64_t *)mirror,
                                             use_records ? colors : nullptr, color_dtype, records, stream, cut_stamp,
                                             // (a prepared scene: the band pre-cull skips the blocks that cannot reach the band)
                                             prepared && prepared->block_bounds && (cull & 32) ? prepared->block_bounds : nullptr,
                                             prepared ? prepared->block_size : 0))
            return rc;
        mark(1);
        MS_HP_T(hp_t3);
        MS_HP_ADD(2, hp_t2, hp_t3);
        // (bit 11: the band's Gaussians were pre-culled -- host_info[6] counts the band's candidates on the grid, not all Gaussians')
        host_info[7] = (no_split ? 16 : 0) | ((cull & 32) ? 2048 : 0) | claim_flag;
        if (!mirror) MS_HIP(hipMemcpyAsync(host_info, info, 7 * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
        if (sync_event && !deferred && !late_cleanup) MS_HIP(hipEventRecord((hipEvent_t)sync_event, stream));   // (deferred: behind the scatter launch)
        // Sync-free frame: if the caller's intersection buffer has room for `cap` entries (it was
        // sized by an earlier frame), enqueue emit + rasterise against that capacity NOW and only
        // then wait for the size record -- the GPU never idles on the hand-off.  Every kernel
        // clamps to `cap`, so an overflowing frame writes nothing out of bounds; it is detected
        // in the finishing half and redone on the exact path.
        MS_HP_T(hp_t4);
        MS_HP_ADD(3, hp_t3, hp_t4);
        if (speculate) {
            const int64_t cmax = split ? ms_split_max_entries() : 0x7fffffffll;
            const int64_t c = cap > cmax ? cmax : cap;
            uint64_t *keys = (uint64_t *)isect_buf;
            int32_t *ids = (int32_t *)((char *)isect_buf + ms::align_up((size_t)c * 8, 256));
            if (split) {
                const ms::BlockLists lists{ranges, (int32_t *)((char *)ids + ms::align_up((size_t)c * 4, 256)), bin_more, tw, th};
                if (int rc = ms::isect_emit_bins(N, means2d, radii, depths, bw, bh, b0, b1, ws + L.off_isect,
                                                 L.isect_bytes, bin_ranges, prev, info, c, bin_flags | claim_bit, lazy,
                                                 near_plane, far_plane, keys, &lists, deferred ? &defer : nullptr, stream))
                    return rc;
                mark(2);
                lazy_lists.keys = keys;
                // (a 16-px grid holds ~1.7x the entries of the 32-px bins it was cut from)
                if (int rc = ms::rasterize_fwd_split(N, c, prev[0] > 0 ? prev[0] * 17 / 10 : c, means2d, conics, colors,
                                                     color_dtype, CDIM, opacities, backgrounds, W, H, r0, r1,
                                                     bin_ranges, &lists, render_colors, &lazy_lists, records, order,
                                                     stage_events ? stage_events[3] : nullptr, stream))
                    return rc;
                host_info[7] = 1 | 8 | ((cull & 32) ? 2048 : 0) | claim_flag;
                if (phase == MS_RENDER_BEGIN) return MS_OK;
            } else {
            if (int rc = ms::isect_emit_speculative(N, means2d, radii, depths, tile_size, tw, th, r0, r1,
                                                    ws + L.off_isect, L.isect_bytes, ranges, info, c, prev,
                                                    /*tight=*/(opacities != nullptr ? 1 : 0) | cull | claim_bit,
                                                    lazy | (bet_light ? 8 : 0) | (leaves_cutoffs ? 32 | (cut_out << 4) : 0),
                                                    near_plane, far_plane, keys, ids, deferred ? &defer : nullptr, stream))
                return rc;
            mark(2);
            MS_HP_T(hp_t5);
            MS_HP_ADD(4, hp_t4, hp_t5);
            lazy_lists.keys = keys;
            if (lazy) {
                uint32_t *cutoffs = const_cast<uint32_t *>(lazy_lists.tau);   // (two buffers of tw * th words)
                lazy_lists.tau_next = leaves_cutoffs ? cutoffs + (size_t)cut_out * tw * th : nullptr;
                lazy_lists.tau = cutoffs + (size_t)cut_in * tw * th;
                lazy_lists.cut_stamp = cut_stamp;
                lazy_lists.cut_words = info + 8;
                lazy_lists.log_keys = keys;
                lazy_lists.cut_inputs = &cut_inputs;
                // stranded bins can be expected -- a depth-cut frame, or the previous frame on this record redid some: the
                // clean-up takes two launches (rasterize.hip, k_redo_sort), 2.4 us a frame that quiet frames do not pay
                const int64_t prev_redos = (prev[7] & 4) ? 0 : (prev[5] & 0xffffffffll) + ((prev[5] >> 32) & 0x3fffffffll);
                // (bins of 48 px and more -- nine or sixteen blocks to a bin, lists of tens of thousands -- always: such a frame's
                // FIRST encounter with stranded bins is the 75 ms one)
                // (round 5: a differentiable frame ALWAYS -- it owns fresh scratch, so the record it is handed says nothing about
                // its predecessor, and its backward walks the whole-bin sorted ids k_redo_sort leaves instead of sorting the
                // bin's keys again in global memory: rasterize_bwdq.hip, k_rasterize_bwd_redo)
                lazy_lists.redo_sort = (cut_stamp != 0u || prev_redos > 0 || tile_size >= 48 || aux_frame) && ms_redo_sort_enabled() ? 1 : 0;
                if (late_cleanup) {
                    host_info[8] = 0;
                    lazy_lists.verdict = (int32_t *)((int64_t *)mirror + 8);
                }
            }
            if (int rc = ms::rasterize_fwd(N, c, prev[0] > 0 ? prev[0] : c, means2d, conics, colors, color_dtype, CDIM,
                                           opacities, backgrounds, W, H, tile_size, r0, r1, ranges, ids,
                                           render_colors, render_alphas, last_ids,
                                           // (a light frame has no sorted FRONT a pixel could outlive: no clean-up launch)
                                           lazy && !bet_light ? &lazy_lists : nullptr,
                                           records, order, clip0, clip1, stage_events ? stage_events[3] : nullptr, stream,
                                           list_nq ? (int32_t *)((char *)ids + ms::align_up((size_t)c * 4, 256)) : nullptr, quad_counts,
                                           zero_rows, zero_rows_bytes))
                return rc;
            // (bit 9: the rasteriser was given lazily sorted fronts -- front counts and redo flags are this frame's; bit 10: a
            // lazily sorted frame -- no merge scratch in the exact layout.  ms_render_bwd reads both.)
            host_info[7] = 1 | (prev[3] > 0 ? 2 : 0) | (no_split ? 16 : 0) | (bet_light ? 32 : 0) | cut_bits |
                           (lazy && !bet_light ? 512 : 0) | (lazy ? 1024 : 0) | ((cull & 32) ? 2048 : 0) |
                           (lazy_lists.verdict ? 4096 : 0) |   // (bit 12: the clean-up launches were left to the finishing half)
                           claim_flag |                        // (bit 14: the histogram rows are per-XCD claims)
                           (zero_rows && r1 > r0 ? 32768 : 0) |   // (bit 15: the rasteriser zeroed the backward's rows in the workspace)
                           (list_nq ? 8192 : 0);               // (bit 13: the quads' lists for the backward sit behind the ids)
#ifdef MS_DIAG
            {
                MS_HP_T(hp_t6);
                MS_HP_ADD(5, hp_t5, hp_t6);
                MS_HP_ADD(6, hp_t0, hp_t6);
                g_hp.frame_done();
            }
#endif
            }
        }
        if (phase == MS_RENDER_BEGIN) return MS_OK;
    }
    if (phase != MS_RENDER_RESUME) {
        // the one size hand-off of a frame (on a speculated frame the record is long written: it
        // precedes the emit in stream order)
        if (sync_event) MS_HIP(hipEventSynchronize((hipEvent_t)sync_event));
        else MS_HIP(hipStreamSynchronize(stream));
        speculated = (host_info[7] & 1) != 0;
        if (speculated) {
            const int64_t cap = split ? (int64_t)((isect_bytes - 768) / 28) : (int64_t)((isect_bytes - (list_nq ? 768 : 512)) / (12 + 4 * list_nq));
            const int64_t cmax = split ? ms_split_max_entries() : 0x7fffffffll;
            const int64_t c = cap > cmax ? cmax : cap;
            const int64_t Ms = host_info[0];
            const bool large_ok = lazy || host_info[3] == 0 || (host_info[7] & 2);  // large class sorted iff launched
            // (bit 5: only the short lists were sorted -- holds iff the frame has no heavy tile)
            const bool light_ok = !(host_info[7] & 32) || host_info[2] + host_info[3] + host_info[4] == 0;
            if (Ms > 0 && Ms <= c && (lazy || host_info[4] == 0) && large_ok && light_ok) return MS_OK;  // the common case
            // else: empty scene, overflow or a tile needing the merge path -> exact path below
        }
    } else {
        speculated = (host_info[7] & 1) != 0;  // a redo after growing the buffer: events were already marked
    }
    if (host_info[7] & 64)   // a depth-cut frame that did not fit its buffer: the frame again, every pair written
        return render_fwd_impl(prepared, restart | 2, N, means3d, scales, scales_are_log, quats, opacities, colors, color_dtype, CDIM, viewmat,
                               fx, fy, cx, cy, W, H, eps2d, near_plane, far_plane, tile_size, rows16 ? clip0 : tile_row_begin,
                               rows16 ? clip1 : tile_row_end, backgrounds, workspace, workspace_bytes, isect_buf, isect_bytes, host_info,
                               (resume & ~0xff) | MS_RENDER_WHOLE, render_colors, render_alphas, last_ids,
                               speculated ? nullptr : stage_events, sync_event, stream_);
    const int64_t M = host_info[0], n_xl = host_info[4];
    MS_REQUIRE(M >= 0 && M <= 0x7fffffffll, MS_ERR_TOO_LARGE, "render_fwd: %lld intersections do not fit int32",
               (long long)M);
    if (host_info[6] == 0 && r0 == 0 && r1 == th) {
        // whole-image call with no Gaussian's bounding box on the grid (gsplat's isect_tiles would
        // return an empty list; with tight binning M alone can be 0 while boxes exist): the
        // reference returns a zeros image here, not the background (render.py:73-76).  A band call
        // leaves that rule to the caller, who reads the same count from host_info[6].
        MS_HIP(hipMemsetAsync(render_colors, 0, (size_t)H * W * CDIM * sizeof(float), stream));
        if (!speculated) { mark(2); mark(3); }
        return MS_OK;
    }
    if (split && M > ms_split_max_entries())   // 4 M block-list slots would not fit int32: the frame again, on 16-px tiles
        return render_fwd_impl(prepared, restart | 1, N, means3d, scales, scales_are_log, quats, opacities, colors, color_dtype, CDIM, viewmat,
                               fx, fy, cx, cy, W, H, eps2d, near_plane, far_plane, tile_size, tile_row_begin,
                               tile_row_end, backgrounds, workspace, workspace_bytes, isect_buf, isect_bytes, host_info,
                               (resume & ~0xff) | MS_RENDER_WHOLE, render_colors, render_alphas, last_ids,
                               speculated ? nullptr : stage_events, sync_event, stream_);
    const size_t need = split ? split_isect_bytes(M)
                              : ms_render_isect_bytes(M, n_xl > 0 && !lazy) + (list_nq ? ms::align_up((size_t)(M > 0 ? M : 1) * list_nq * 4, 256) : 0);
    host_info[5] = (int64_t)need;
    host_info[7] |= 4;  // the lists the caller may read back are in the EXACT layout (below)
    host_info[7] = (host_info[7] & ~8192ll) | (list_nq ? 8192 : 0);
    {
        const bool fronts = !split && lazy && host_info[2] + host_info[3] + host_info[4] > 0;
        host_info[7] = (host_info[7] & ~(512ll | 1024ll)) | (fronts ? 512 : 0) | (lazy ? 1024 : 0);
    }
    MS_REQUIRE(isect_buf && isect_bytes >= need, MS_ERR_WORKSPACE,
               "render_fwd: intersection buffer %zu < %zu (grow it and call again with resume=1)", isect_bytes,
               need);
    char *ib = (char *)isect_buf;
    const size_t key_bytes = ms::align_up((size_t)M * 8, 256);
    uint64_t *keys = (uint64_t *)ib;
    const bool merge = n_xl > 0 && !lazy;
    uint64_t *tmp = merge ? (uint64_t *)(ib + key_bytes) : nullptr;
    int32_t *ids = (int32_t *)(ib + key_bytes * (merge ? 2 : 1));
    if (split) {
        const int64_t c = M > 0 ? M : 1;
        const ms::BlockLists lists{ranges, (int32_t *)((char *)ids + ms::align_up((size_t)M * 4, 256)), bin_more, tw, th};
        if (int rc = ms::isect_emit_bins(N, means2d, radii, depths, bw, bh, b0, b1, ws + L.off_isect, L.isect_bytes,
                                         bin_ranges, host_info, nullptr, c, bin_flags | ((host_info[7] & 16384) ? ms::kTightClaimed : 0), lazy, near_plane, far_plane,
                                         keys, &lists, nullptr, stream))
            return rc;
        if (!speculated) mark(2);
        lazy_lists.keys = keys;
        return ms::rasterize_fwd_split(N, c, M * 17 / 10, means2d, conics, colors, color_dtype, CDIM, opacities,
                                       backgrounds, W, H, r0, r1, bin_ranges, &lists, render_colors, &lazy_lists,
                                       records, order,
                                       (!speculated && stage_events) ? stage_events[3] : nullptr, stream);
    }
    if (int rc = ms::isect_emit_exact(N, means2d, radii, depths, tile_size, tw, th, r0, r1, ws + L.off_isect,
                                      L.isect_bytes, ranges, host_info, /*tight=*/(opacities != nullptr ? 1 : 0) | cull | ((host_info[7] & 16384) ? ms::kTightClaimed : 0), lazy, near_plane, far_plane,
                                      keys, tmp, ids, stream))
        return rc;
    if (!speculated) mark(2);
    lazy_lists.keys = keys;
    if (lazy && aux_frame && ms_redo_sort_enabled()) lazy_lists.redo_sort = 1;   // (as on the sync-free path: for the backward's redo launch)
    if (int rc = ms::rasterize_fwd(N, M, M, means2d, conics, colors, color_dtype, CDIM, opacities, backgrounds, W, H,
                                   tile_size, r0, r1, ranges, ids, render_colors, render_alphas, last_ids,
                                   lazy && host_info[2] + host_info[3] + host_info[4] > 0 ? &lazy_lists : nullptr,
                                   records, order, clip0, clip1,
                                   (!speculated && stage_events) ? stage_events[3] : nullptr, stream,
                                   list_nq ? (int32_t *)((char *)ids + ms::align_up((size_t)(M > 0 ? M : 1) * 4, 256)) : nullptr, quad_counts,
                                   zero_rows, zero_rows_bytes))
        return rc;
    host_info[7] = (host_info[7] & ~32768ll) | (zero_rows && r1 > r0 ? 32768 : 0);
    return MS_OK;
}

extern "C" int ms_render_fwd(int64_t N, const float *means3d, const float *scales, int scales_are_log,
                             const float *quats, const float *opacities, const void *colors,
                             int color_dtype, int CDIM, const float *viewmat, float fx, float fy,
                             float cx, float cy, int W, int H, float eps2d, float near_plane,
                             float far_plane, int tile_size, int tile_row_begin, int tile_row_end,
                             const float *backgrounds, void *workspace, size_t workspace_bytes, void *isect_buf,
                             size_t isect_bytes, int64_t *host_info, int resume,
                             float *render_colors, float *render_alphas, int32_t *last_ids,
                             void **stage_events, void *sync_event, void *stream_) {
    return render_fwd_impl(nullptr, 0, N, means3d, scales, scales_are_log, quats, opacities, colors, color_dtype, CDIM, viewmat, fx,
                           fy, cx, cy, W, H, eps2d, near_plane, far_plane, tile_size, tile_row_begin, tile_row_end,
                           backgrounds, workspace, workspace_bytes, isect_buf, isect_bytes, host_info, resume & ~MS_RENDER_DEFER_CLEANUP,
                           render_colors, render_alphas, last_ids, stage_events, sync_event, stream_);
}


// ---- backward of a differentiable frame ---------------------------------------------------------------------------
// ms_render_fwd with render_alphas / last_ids leaves everything the backward needs in the frame's own scratch: the
// projected arrays and the ready-made records in `workspace`, the fully sorted lists in `isect_buf`.  One C call runs
// the backward rasteriser (staging from the records) and the backward projection on it: no Python between the
// kernels, and the layout of the scratch stays the library's own business.
extern "C" size_t ms_render_bwd_workspace_bytes(int64_t N, int CDIM) {
    const size_t n = (size_t)(N > 0 ? N : 1);
    return ms::align_up(ms_rasterize_bwd_workspace_bytes(N, CDIM), 256) + ms::align_up(n * 8, 256) + ms::align_up(n * 12, 256);
}

// The rasteriser half of a differentiable frame's backward -- the whole frame's, or a multi-GPU rank's BAND of tile rows --
// into the packed per-Gaussian rows (f32[N][16]: the quad-wave kernel's raw sums), zeroed here.  Only frames the quad-wave
// kernel takes: three channels, ready-made records, a tile size that is a multiple of 16, the frame's image at hand.
static int render_bwd_rows_impl(int64_t N, int CDIM, int W, int H, int tile_size, int r0, int r1, const float *backgrounds,
                                const void *workspace, size_t workspace_bytes, const void *isect_buf, size_t isect_bytes,
                                const int64_t *host_info, const float *render_colors, const float *render_alphas,
                                const float *v_render_colors, const float *v_render_alphas, float *rows, void *stream_,
                                int32_t *redo_counts_host = nullptr) {
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t M = host_info[0], n_xl = host_info[4];
    int32_t *count_mirror = nullptr;
    if (redo_counts_host) {   // (mapped pinned memory: the device-side address of the same words)
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, redo_counts_host, 0) == hipSuccess) count_mirror = (int32_t *)dp;
        else (void)hipGetLastError();
    }
    MS_REQUIRE(M > 0 && M <= 0x7fffffffll, MS_ERR_TOO_LARGE, "render_bwd: bad intersection count %lld", (long long)M);
    MS_REQUIRE(workspace && isect_buf && render_alphas && render_colors && v_render_colors && rows, MS_ERR_INVALID_ARG,
               "render_bwd: null pointer");
    MS_REQUIRE(CDIM == 3 && tile_size % 16 == 0, MS_ERR_INVALID_ARG, "render_bwd: the quad-wave backward takes 3 channels and 16-px blocks");
    MS_REQUIRE(!(host_info[7] & 8), MS_ERR_INVALID_ARG, "render_bwd: the frame's lists are block lists of a split frame");
    const int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
    const WsLayout L = ws_layout(N, tw, th);
    MS_REQUIRE(workspace_bytes >= L.total, MS_ERR_WORKSPACE, "render_bwd: workspace %zu < %zu", workspace_bytes, L.total);
    const char *ws = (const char *)workspace;
    const int32_t *ranges = (const int32_t *)(ws + L.off_ranges);
    // (bit 13 of the frame's flag word: its quads' lists sit behind the sorted ids -- rasterize.hip, RasterArgs::quad_lists)
    const int list_nq = (host_info[7] & 8192) ? quad_list_nq(tile_size) : 0;
    size_t ids_off, lists_off;
    if (host_info[7] & 4) {
        ids_off = ms::align_up((size_t)M * 8, 256) * (n_xl > 0 && !(host_info[7] & 1024) ? 2 : 1);
        lists_off = ids_off + ms::align_up((size_t)M * 4, 256);
    } else {
        MS_REQUIRE(isect_bytes > 768, MS_ERR_WORKSPACE, "render_bwd: intersection buffer too small");
        int64_t cap = (int64_t)((isect_bytes - (list_nq ? 768 : 512)) / (12 + 4 * list_nq));
        cap = cap > 0x7fffffffll ? 0x7fffffffll : cap;
        ids_off = ms::align_up((size_t)cap * 8, 256);
        lists_off = ids_off + ms::align_up((size_t)cap * 4, 256);
    }
    MS_REQUIRE(ids_off + (size_t)M * 4 <= isect_bytes, MS_ERR_WORKSPACE, "render_bwd: intersection buffer %zu does not hold %lld ids",
               isect_bytes, (long long)M);
    MS_REQUIRE(!list_nq || lists_off + (size_t)M * list_nq * 4 <= isect_bytes, MS_ERR_WORKSPACE,
               "render_bwd: intersection buffer %zu does not hold the quads' lists", isect_bytes);
    const int32_t *ids = (const int32_t *)((const char *)isect_buf + ids_off);
    const int32_t *quad_lists = list_nq ? (const int32_t *)((const char *)isect_buf + lists_off) : nullptr;
    const int32_t *quad_counts = list_nq ? (const int32_t *)(ws + L.off_quad_counts) : nullptr;
    const void *records = (const void *)(ws + L.off_records);
    // (bit 15: the frame's rasteriser zeroed the workspace's own rows -- handed exactly those, the call has nothing to zero)
    if (!((host_info[7] & 32768) && (const char *)rows == ws + L.off_rows))
        MS_HIP(hipMemsetAsync(rows, 0, (size_t)N * 16 * sizeof(float), stream));
    // a lazily sorted frame: the lists are sorted as deep as the forward rasteriser walked them -- the tiles whose
    // front ran out were redone by the forward's clean-up pass and are this call's second launch (normally empty)
    ms::LazyLists ll{};
    ms::isect_lazy_arrays(const_cast<char *>(ws) + L.off_isect, N, tw, th, &ll);
    const bool fronts = (host_info[7] & 512) != 0;
    const int32_t *order = ms_order_enabled() ? ms::isect_order_array(ws + L.off_isect, N, tw, th) : nullptr;
    if (int rc = ms::rasterize_bwd_quads(N, M, records, backgrounds, W, H, tile_size, ranges, ids, 1,
                                         fronts ? ll.front_count : nullptr, ll.front_threshold, fronts ? ll.redo_flag : nullptr,
                                         render_colors, render_alphas, v_render_colors, v_render_alphas, rows, order, stream_, r0, r1,
                                         quad_lists, quad_counts))
        return rc;
    if (fronts)
        if (int rc = ms::rasterize_bwd_redo(N, M, records, backgrounds, W, H, tile_size, ranges,
                                            (uint64_t *)const_cast<void *>(isect_buf), ids, ll.redo_list, ll.redo_count, ll.redo_flag,
                                            render_colors, render_alphas, v_render_colors, v_render_alphas, rows, stream_, count_mirror))
            return rc;
    return MS_OK;
}

extern "C" size_t ms_render_bwd_rows_bytes(int64_t N) { return ms::align_up((size_t)(N > 0 ? N : 1) * 16 * sizeof(float), 256); }

extern "C" int ms_render_bwd_rows(int64_t N, int CDIM, int W, int H, int tile_size, int tile_row_begin, int tile_row_end,
                                  const float *backgrounds, const void *workspace, size_t workspace_bytes, const void *isect_buf,
                                  size_t isect_bytes, const int64_t *host_info, const float *render_colors,
                                  const float *render_alphas, const float *v_render_colors, const float *v_render_alphas,
                                  float *rows, int32_t *redo_counts_host, void *stream) {
    MS_REQUIRE(N >= 0 && W > 0 && H > 0 && tile_size > 0 && host_info && rows, MS_ERR_INVALID_ARG, "render_bwd_rows: bad argument");
    if (N == 0) return MS_OK;
    if (host_info[6] == 0 || host_info[0] == 0) {   // nothing on the grid / no pair in the band: no gradient from this band
        MS_HIP(hipMemsetAsync(rows, 0, (size_t)N * 16 * sizeof(float), (hipStream_t)stream));
        return MS_OK;
    }
    return render_bwd_rows_impl(N, CDIM, W, H, tile_size, tile_row_begin, tile_row_end, backgrounds, workspace, workspace_bytes,
                                isect_buf, isect_bytes, host_info, render_colors, render_alphas, v_render_colors, v_render_alphas,
                                rows, stream, redo_counts_host);
}

extern "C" int ms_render_bwd_finish(int64_t N, const float *means3d, const float *scales, int scales_are_log, const float *quats,
                                    const float *opacities, int CDIM, const float *viewmat, float fx, float fy, float cx, float cy,
                                    int W, int H, float eps2d, const float *rows, float *v_means3d, float *v_scales, float *v_quats,
                                    float *v_opacities, float *v_colors, void *stream) {
    MS_REQUIRE(N >= 0 && CDIM == 3 && v_means3d && v_scales && v_quats && v_opacities && v_colors, MS_ERR_INVALID_ARG,
               "render_bwd_finish: bad argument");
    if (N == 0) return MS_OK;
    MS_REQUIRE(rows && opacities, MS_ERR_INVALID_ARG, "render_bwd_finish: null pointer");
    return ms::project_bwd_from_rows(N, means3d, scales, scales_are_log, quats, viewmat, fx, fy, cx, cy, W, H, eps2d, nullptr, rows, CDIM,
                                     v_means3d, v_scales, v_quats, v_colors, v_opacities, stream, opacities);
}

extern "C" int ms_render_bwd(int64_t N, const float *means3d, const float *scales, int scales_are_log, const float *quats,
                             const float *opacities, const float *colors, int CDIM, const float *viewmat, float fx, float fy,
                             float cx, float cy, int W, int H, float eps2d, int tile_size, const float *backgrounds,
                             const void *workspace, size_t workspace_bytes, const void *isect_buf, size_t isect_bytes,
                             const int64_t *host_info, const float *render_colors, const float *render_alphas,
                             const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
                             float *v_means3d, float *v_scales, float *v_quats, float *v_opacities, float *v_colors,
                             void *bwd_workspace, size_t bwd_workspace_bytes, void *mid_event, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MS_REQUIRE(N >= 0 && W > 0 && H > 0 && tile_size > 0 && CDIM >= 1 && CDIM <= 32, MS_ERR_INVALID_ARG, "render_bwd: bad sizes");
    MS_REQUIRE(host_info && v_means3d && v_scales && v_quats && v_opacities && v_colors, MS_ERR_INVALID_ARG,
               "render_bwd: null pointer");
    if (N == 0) return MS_OK;
    const int64_t M = host_info[0], n_xl = host_info[4];
    if (host_info[6] == 0 || M == 0) {
        // nothing on the grid / no intersections: the frame was the zeros image (or the background), and constant in
        // every input
        MS_HIP(hipMemsetAsync(v_means3d, 0, (size_t)N * 12, stream));
        MS_HIP(hipMemsetAsync(v_scales, 0, (size_t)N * 12, stream));
        MS_HIP(hipMemsetAsync(v_quats, 0, (size_t)N * 16, stream));
        MS_HIP(hipMemsetAsync(v_opacities, 0, (size_t)N * 4, stream));
        MS_HIP(hipMemsetAsync(v_colors, 0, (size_t)N * CDIM * 4, stream));
        return MS_OK;
    }
    MS_REQUIRE(M > 0 && M <= 0x7fffffffll, MS_ERR_TOO_LARGE, "render_bwd: bad intersection count %lld", (long long)M);
    MS_REQUIRE(means3d && scales && quats && opacities && colors && viewmat && workspace && isect_buf && render_alphas &&
                   (last_ids || render_colors) && v_render_colors && bwd_workspace,
               MS_ERR_INVALID_ARG, "render_bwd: null pointer");
    MS_REQUIRE(!(host_info[7] & 8), MS_ERR_INVALID_ARG, "render_bwd: the frame's lists are block lists of a split frame");
    const int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
    const WsLayout L = ws_layout(N, tw, th);
    MS_REQUIRE(workspace_bytes >= L.total, MS_ERR_WORKSPACE, "render_bwd: workspace %zu < %zu", workspace_bytes, L.total);
    MS_REQUIRE(bwd_workspace_bytes >= ms_render_bwd_workspace_bytes(N, CDIM), MS_ERR_WORKSPACE,
               "render_bwd: backward workspace %zu < %zu", bwd_workspace_bytes, ms_render_bwd_workspace_bytes(N, CDIM));
    const char *ws = (const char *)workspace;
    const float *means2d = (const float *)(ws + L.off_means2d), *conics = (const float *)(ws + L.off_conics);
    const int32_t *radii = (const int32_t *)(ws + L.off_radii), *ranges = (const int32_t *)(ws + L.off_ranges);
    // where the frame left its sorted ids (the layout rules of ms_render_fwd: exact, or sized by the buffer's capacity)
    size_t ids_off;
    if (host_info[7] & 4) {
        ids_off = ms::align_up((size_t)M * 8, 256) * (n_xl > 0 && !(host_info[7] & 1024) ? 2 : 1);
    } else {
        MS_REQUIRE(isect_bytes > 768, MS_ERR_WORKSPACE, "render_bwd: intersection buffer too small");
        const int list_nq = (host_info[7] & 8192) ? quad_list_nq(tile_size) : 0;   // (as render_bwd_rows_impl)
        int64_t cap = (int64_t)((isect_bytes - (list_nq ? 768 : 512)) / (12 + 4 * list_nq));
        cap = cap > 0x7fffffffll ? 0x7fffffffll : cap;
        ids_off = ms::align_up((size_t)cap * 8, 256);
    }
    MS_REQUIRE(ids_off + (size_t)M * 4 <= isect_bytes, MS_ERR_WORKSPACE, "render_bwd: intersection buffer %zu does not hold %lld ids",
               isect_bytes, (long long)M);
    const int32_t *ids = (const int32_t *)((const char *)isect_buf + ids_off);
    const void *records = CDIM == 3 ? (const void *)(ws + L.off_records) : nullptr;
    char *bw = (char *)bwd_workspace;
    const size_t rb = ms::align_up(ms_rasterize_bwd_workspace_bytes(N, CDIM), 256);
    float *v_means2d = (float *)(bw + rb), *v_conics = (float *)(bw + rb + ms::align_up((size_t)N * 8, 256));
    // (the rasteriser's packed 64-byte rows go straight into the backward projection, which unpacks v_colors / v_opacities
    // on its way: no k_unpack_grads pass -- whenever the packed path runs: <= 4 channels, M > 0)
    const bool packed_rows = CDIM <= 4 && rb > 0 && M > 0 && N <= 0x7fffffffll;
    // a 3-channel frame whose image the caller still holds: the quad-wave kernel (rasterize_bwdq.hip) walks the frame's own
    // lists front to back from the records and leaves raw sums, which the backward projection finishes
    // (MOJOSPLAT_BWD_QUADS=0: the older kernel, which needs last_ids)
    static const int quads_on = [] { const char *e = getenv("MOJOSPLAT_BWD_QUADS"); return e ? atoi(e) != 0 : 1; }();
    if (render_colors && records && packed_rows && tile_size % 16 == 0 && (quads_on || !last_ids)) {
        if (int rc = render_bwd_rows_impl(N, CDIM, W, H, tile_size, 0, th, backgrounds, workspace, workspace_bytes, isect_buf, isect_bytes,
                                          host_info, render_colors, render_alphas, v_render_colors, v_render_alphas, (float *)bw, stream_))
            return rc;
        if (mid_event) MS_HIP(hipEventRecord((hipEvent_t)mid_event, stream));
        return ms::project_bwd_from_rows(N, means3d, scales, scales_are_log, quats, viewmat, fx, fy, cx, cy, W, H, eps2d, nullptr,
                                         (const float *)bw, CDIM, v_means3d, v_scales, v_quats, v_colors, v_opacities, stream_, opacities);
    }
    MS_REQUIRE(last_ids, MS_ERR_INVALID_ARG, "render_bwd: this frame needs last_ids");
    if (int rc = ms::rasterize_bwd(N, M, means2d, conics, colors, CDIM, opacities, backgrounds, W, H, tile_size, ranges, ids,
                                   render_alphas, last_ids, v_render_colors, v_render_alphas, v_means2d, v_conics, v_colors,
                                   v_opacities, rb ? bw : nullptr, rb, /*overwrite: 2 = leave the rows packed=*/packed_rows ? 2 : 1, records,
                                   // (a whole-image frame on 16-px tiles: its count pass ordered exactly these blocks)
                                   (tile_size == 16 && ms_order_enabled()) ? ms::isect_order_array(ws + L.off_isect, N, tw, th) : nullptr,
                                   stream_))
        return rc;
    if (mid_event) MS_HIP(hipEventRecord((hipEvent_t)mid_event, stream));   // (in-situ timing: between the two stages)
    if (packed_rows)
        return ms::project_bwd_from_rows(N, means3d, scales, scales_are_log, quats, viewmat, fx, fy, cx, cy, W, H, eps2d, radii,
                                         (const float *)bw, CDIM, v_means3d, v_scales, v_quats, v_colors, v_opacities, stream_);
    return ms_project_gaussians_bwd(N, means3d, scales, scales_are_log, quats, viewmat, fx, fy, cx, cy, W, H, eps2d, radii,
                                    v_means2d, v_conics, nullptr, v_means3d, v_scales, v_quats, stream_);
}


// ---- a multi-GPU rank's band of a frame, in two calls ----------------------------------------------------------------
// What distributed.py's asynchronous path did around ms_render_fwd(BEGIN) / (FINISH) in Python -- marshalling five tensors
// and a camera per frame, four event operations, record_stream on every tensor that crosses to the lane, the frame object --
// cost ~100-130 us of host time per band frame against 18-45 us for the library's own enqueue, level with the GPU time of a
// config-5 band at 8 ranks (profiles/r04_band_rehearsal_cfg5.jsonl: host_us_median 146-151).  Here the caller hands over
// structs it builds ONCE per scene / lane (ms_scene, ms_band_lane) and one small per-frame struct; the stream ordering around
// the lane (caller -> lane before the band, lane -> caller after it) happens in here.
static int band_validate(const ms_band_frame *f, const ms_band_lane *lane, const char *who) {
    MS_REQUIRE(f && lane && f->scene, MS_ERR_INVALID_ARG, "%s: null frame / lane / scene", who);
    // (sync_event may be null, as for ms_render_fwd: the frame then waits for its size record by synchronising the lane's stream)
    MS_REQUIRE(lane->workspace && lane->host_info && lane->in_event && lane->out_event,
               MS_ERR_INVALID_ARG, "%s: the lane lacks scratch or an event", who);
    MS_REQUIRE(f->render_colors && f->viewmat, MS_ERR_INVALID_ARG, "%s: null image / view matrix", who);
    MS_REQUIRE(f->W > 0 && f->H > 0 && f->tile_size > 0 && f->row_begin >= 0 && f->row_begin <= f->row_end, MS_ERR_INVALID_ARG,
               "%s: bad sizes / row band", who);
    return MS_OK;
}

static int band_call(const ms_band_frame *f, const ms_band_lane *lane, int phase) {
    const ms_scene *sc = f->scene;
    return render_fwd_impl(sc, 0, sc->N, sc->means3d, sc->scales, sc->scales_are_log, sc->quats, sc->opacities, sc->colors,
                           sc->color_dtype, sc->CDIM, f->viewmat, f->fx, f->fy, f->cx, f->cy, f->W, f->H, f->eps2d, f->near_plane,
                           f->far_plane, f->tile_size, f->row_begin, f->row_end, f->backgrounds, lane->workspace,
                           lane->workspace_bytes, lane->isect_buf, lane->isect_bytes, lane->host_info,
                           phase | (f->flags & ~0xff), f->render_colors, nullptr, nullptr, f->stage_events, lane->sync_event, lane->stream);
}

extern "C" int ms_render_band_begin(const ms_band_frame *f, const ms_band_lane *lane, void *caller_stream) {
    if (int rc = band_validate(f, lane, "render_band_begin")) return rc;
    // everything the band reads was produced on (or before) the caller's stream: the lane waits for it
    MS_HIP(hipEventRecord((hipEvent_t)lane->in_event, (hipStream_t)caller_stream));
    MS_HIP(hipStreamWaitEvent((hipStream_t)lane->stream, (hipEvent_t)lane->in_event, 0));
    if (int rc = band_call(f, lane, MS_RENDER_BEGIN)) return rc;
    // (deferred clean-up: the finishing half waits for the END of the band -- its rasteriser -- before it looks at the verdict)
    if (lane->host_info[7] & 4096) MS_HIP(hipEventRecord((hipEvent_t)lane->out_event, (hipStream_t)lane->stream));
    return MS_OK;
}

extern "C" int ms_render_band_finish(const ms_band_frame *f, const ms_band_lane *lane, void *caller_stream, int resume,
                                     int64_t *status) {
    if (int rc = band_validate(f, lane, "render_band_finish")) return rc;
    MS_REQUIRE(status, MS_ERR_INVALID_ARG, "render_band_finish: null status");
    // the wait for the band's size record, the check, the exact redo if the speculation did not hold (MS_ERR_WORKSPACE:
    // lane->host_info[5] bytes of isect_buf needed -- grow it, update the lane, call again with resume = 1)
    const int64_t *h = lane->host_info;
    // MS_RENDER_DEFER_CLEANUP: the band's clean-up launches -- up to four, empty on almost every frame: 19 us of a 118 us band
    // and as many microseconds of host time -- were not enqueued.  Wait for the band's rasteriser (with another band in flight
    // on the other lane the GPU does not idle meanwhile), read its verdict from the pinned record, enqueue them only if a bin
    // asked for them.
    const bool deferred = !resume && (h[7] & 4096) != 0;
    if (deferred) MS_HIP(hipEventSynchronize((hipEvent_t)lane->out_event));
    if (int rc = band_call(f, lane, resume ? MS_RENDER_RESUME : MS_RENDER_FINISH)) return rc;
    bool settled = false;   // the host has SEEN the lane's work end: nothing to order the caller's stream behind
    if ((h[7] & 4096) && !(h[7] & 4)) {   // (bit 2: the band was redone on the exact path, clean-up launches and all)
        if (h[8] != 0) {
            void *mirror = nullptr;
            MS_HIP(hipHostGetDevicePointer(&mirror, lane->host_info, 0));
            if (int rc = ms::rasterize_deferred_cleanup((int64_t *)mirror + 8, lane->stream)) return rc;
        } else {
            settled = deferred;
        }
    }
    if (!settled) {
        // the band is complete on the lane's stream: whatever the caller enqueues next (the exchange) comes behind it
        MS_HIP(hipEventRecord((hipEvent_t)lane->out_event, (hipStream_t)lane->stream));
        MS_HIP(hipStreamWaitEvent((hipStream_t)caller_stream, (hipEvent_t)lane->out_event, 0));
    }
    status[0] = h[6];                      // Gaussians on the grid (a pre-culled band: of its candidates)
    status[1] = (h[7] & 2048) ? 1 : 0;     // the library pre-culled the band
    status[2] = h[0];                      // pairs in the band
    status[3] = h[7];                      // the frame's flag word
    return MS_OK;
}

// Camera batch: n_lanes views in flight (begin view v + 1 before finishing view v).  See the header.
extern "C" int ms_render_fwd_batch(int C, int64_t N, const float *means3d, const float *scales, int scales_are_log,
                                   const float *quats, const float *opacities, const void *colors, int color_dtype,
                                   int CDIM, const float *viewmats, const float *intrinsics, int W, int H,
                                   float eps2d, float near_plane, float far_plane, int tile_size,
                                   const float *backgrounds, int n_lanes, const ms_view_lane *lanes, int flags,
                                   float *render_colors, int64_t *counts, int *views_done,
                                   size_t *need_isect_bytes, int *need_lane) {
    MS_REQUIRE(C >= 0 && n_lanes >= 1 && n_lanes <= 4 && lanes && views_done && need_isect_bytes && need_lane,
               MS_ERR_INVALID_ARG, "render_fwd_batch: bad batch / lane arguments");
    MS_REQUIRE(C == 0 || (viewmats && intrinsics && render_colors), MS_ERR_INVALID_ARG, "render_fwd_batch: null pointer");
    MS_REQUIRE(*views_done >= 0 && *views_done <= C, MS_ERR_INVALID_ARG, "render_fwd_batch: views_done out of range");
    MS_REQUIRE(W > 0 && H > 0 && tile_size > 0 && CDIM >= 1, MS_ERR_INVALID_ARG, "render_fwd_batch: bad sizes");
    for (int l = 0; l < n_lanes; ++l)
        MS_REQUIRE(lanes[l].workspace && lanes[l].host_info, MS_ERR_INVALID_ARG, "render_fwd_batch: lane %d lacks scratch", l);
    const int th = (H + tile_size - 1) / tile_size;
    const size_t view_floats = (size_t)H * W * CDIM;
    const int mode = flags & (MS_RENDER_FULL_SORT | (3 * MS_RENDER_FRONT_LEVEL));
    auto call = [&](int v, int phase) {
        const ms_view_lane &L = lanes[v % n_lanes];
        const float *K = intrinsics + 4 * (size_t)v;
        return ms_render_fwd(N, means3d, scales, scales_are_log, quats, opacities, colors, color_dtype, CDIM,
                             viewmats + 16 * (size_t)v, K[0], K[1], K[2], K[3], W, H, eps2d, near_plane, far_plane,
                             tile_size, 0, th, backgrounds, L.workspace, L.workspace_bytes, L.isect_buf, L.isect_bytes,
                             L.host_info, phase | mode, render_colors + view_floats * (size_t)v, nullptr, nullptr, nullptr,
                             L.sync_event, L.stream);
    };
    int pending[4] = {-1, -1, -1, -1};
    int first_fail = -1, fail_rc = MS_OK;
    size_t fail_need = 0;
    auto finish = [&](int lane) {
        const int u = pending[lane];
        if (u < 0) return;
        pending[lane] = -1;
        const int rc = call(u, MS_RENDER_FINISH);
        if (rc == MS_OK) {
            if (counts) counts[u] = lanes[lane].host_info[0];
        } else if (first_fail < 0 || u < first_fail) {
            first_fail = u;
            fail_rc = rc;
            fail_need = rc == MS_ERR_WORKSPACE ? (size_t)lanes[lane].host_info[5] : 0;
        }
    };
    for (int v = *views_done; v < C && first_fail < 0; ++v) {
        const int lane = v % n_lanes;
        finish(lane);                       // the view this lane rendered n_lanes views ago
        if (first_fail >= 0) break;
        const int rc = call(v, MS_RENDER_BEGIN);
        if (rc != MS_OK) {
            first_fail = v;
            fail_rc = rc;
            break;
        }
        pending[lane] = v;
    }
    // drain, oldest view first, so that nothing is in flight when the call returns
    for (;;) {
        int lane = -1;
        for (int l = 0; l < n_lanes; ++l)
            if (pending[l] >= 0 && (lane < 0 || pending[l] < pending[lane])) lane = l;
        if (lane < 0) break;
        finish(lane);
    }
    if (first_fail >= 0) {
        *views_done = first_fail;
        *need_lane = first_fail % n_lanes;
        *need_isect_bytes = fail_need;
        return fail_rc;
    }
    *views_done = C;
    return MS_OK;
}
