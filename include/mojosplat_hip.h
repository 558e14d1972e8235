/*
 * mojosplat_hip.h -- C ABI of libmojosplat_hip.so, the MI355X (gfx950) backend for the
 * mojosplat render path: EWA projection -> tile binning/sort -> tile rasteriser.
 *
 * This is the drop-in boundary.  Each entry point replaces one custom-op / external call
 * of the reference (paths relative to the reference repo root):
 *
 *   ms_project_gaussians_fwd          op "project_gaussians", mojosplat/kernels/projection.mojo:260-326,
 *                                     called from mojosplat/projection.py:429-454; and
 *                                     gsplat.fully_fused_projection, projection.py:381-397
 *   ms_isect_tiles_count / _emit      gsplat.isect_tiles, mojosplat/binning.py:73-82
 *   ms_isect_offset_encode            gsplat.isect_offset_encode, mojosplat/binning.py:84
 *   ms_rasterize_to_pixels_3dgs_fwd   op "rasterize_to_pixels_3dgs_fwd",
 *                                     mojosplat/kernels/rasterization.mojo:169-240, called from
 *                                     mojosplat/rasterization.py:169-183; and
 *                                     gsplat.rasterize_to_pixels, rasterization.py:109-122
 *   ms_rasterize_to_pixels_3dgs_bwd,  no reference counterpart (reference is forward-only,
 *   ms_project_gaussians_bwd          render.py:11); gsplat's backward semantics
 *   ms_render_fwd                     the whole of render_gaussians' device work
 *                                     (mojosplat/render.py:63-101) in one call
 *   ms_render_bwd                     the backward of such a frame (no reference counterpart) in one call
 *   ms_render_fwd_batch               the same for C cameras: the camera dimension of the reference's
 *                                     kernels (kernels/projection.mojo:32-37) that its wrappers pin to 1
 *
 * Conventions (same as the reference's op convention, projection.py:438-454):
 *   - destination passing: the caller (PyTorch) owns and pre-allocates every buffer,
 *     including scratch; the library never allocates device memory and keeps no per-frame or
 *     per-scene state.  What it does keep, process-wide and thread-safe: the thread-local
 *     error string, the environment switches it reads once (MOJOSPLAT_LAZY_SORT, MOJOSPLAT_SPLIT,
 *     MOJOSPLAT_SPLIT_MAX_ENTRIES and the measurement knobs listed in
 *     INTEGRATION.md; MOJOSPLAT_DEPTH_CUT / MOJOSPLAT_DEPTH_CUT_MIN_PAIRS too: ms_config_depth_cut changes them in-process),
 *     a counter that stamps depth-cut frames, and a mutex-guarded table of the (device, kernel) pairs whose dynamic-LDS ceiling it has
 *     already raised (hipFuncSetAttribute);
 *   - all pointers are DEVICE pointers unless a parameter says "host"; tensors are
 *     contiguous row-major with the layouts written next to each parameter;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); the per-stage
 *     entry points never synchronise, so they can be captured into a hipGraph (the one
 *     exception, ms_render_fwd, says so below);
 *   - every function returns 0 on success or a non-zero ms_status; no C++ exception
 *     crosses this boundary.  ms_last_error_string() describes the last failure on the
 *     calling thread.
 *   - single camera (C = 1), like every wrapper of the reference (projection.py:431,
 *     rasterization.py:175); ms_render_fwd_batch is the one entry point with C > 1.
 */
#ifndef MOJOSPLAT_HIP_H
#define MOJOSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MS_ABI_VERSION 3   /* 2: ms_render_bwd takes the frame's image (render_colors); 3: ms_render_redo_counts, the band-frame pair, ms_scene_prepare */

typedef enum ms_status {
    MS_OK = 0,
    MS_ERR_INVALID_ARG = 1,   /* null pointer, negative size, unsupported channel count ... */
    MS_ERR_WORKSPACE = 2,     /* workspace too small (ask ms_*_workspace_bytes)           */
    MS_ERR_TOO_LARGE = 3,     /* tile grid does not fit the binning kernels' LDS budget   */
    MS_ERR_HIP = 4            /* a HIP runtime call failed; see ms_last_error_string()    */
} ms_status;

typedef enum ms_color_dtype { MS_COLOR_F32 = 0, MS_COLOR_F16 = 1 } ms_color_dtype;

int ms_version(void);
const char *ms_last_error_string(void);

/* ---------------------------------------------------------------------------------------
 * Projection (EWA 3D -> 2D), gsplat semantics.
 *   in : means3d f32[N,3]; scales f32[N,3] (log-space when scales_are_log != 0, the exp is
 *        fused; linear otherwise, which is what the reference's kernel receives);
 *        quats f32[N,4] wxyz (normalised in-kernel); opacities f32[N] or NULL (NULL = no
 *        opacity cull / no opacity-aware extent); viewmat f32[16] row-major world->camera
 *        IN DEVICE MEMORY; pinhole fx fy cx cy; image W H; eps2d (0.3), near, far,
 *        radius_clip (0).
 *   out: means2d f32[N,2]; conics f32[N,3] (a,b,c of the inverse blurred 2x2 covariance);
 *        depths f32[N]; radii i32[N,2].  Culled Gaussians get radii = 0 and zeros elsewhere.
 * ------------------------------------------------------------------------------------- */
int ms_project_gaussians_fwd(int64_t N, const float *means3d, const float *scales,
                             int scales_are_log, const float *quats, const float *opacities,
                             const float *viewmat, float fx, float fy, float cx, float cy,
                             int W, int H, float eps2d, float near_plane, float far_plane,
                             float radius_clip, float *means2d, float *conics, float *depths,
                             int32_t *radii, void *stream);

/* ---------------------------------------------------------------------------------------
 * Binning.  Two calls around the one unavoidable size hand-off (the number of
 * intersections M is data dependent), exactly where gsplat.isect_tiles has its own.
 *
 * ms_isect_tiles_count: per-tile counts -> exclusive offsets.
 *   in : means2d f32[N,2], radii i32[N,2]; tile_size; tile grid tile_w x tile_h;
 *        [row_begin,row_end) restricts binning to a band of tile rows (multi-GPU
 *        sharding; pass 0,tile_h for the whole image);
 *        workspace of ms_isect_workspace_bytes(...) bytes.
 *   out: tile_ranges i32[tile_h,tile_w,2] = [start,end) into the sorted intersection list
 *        (tiles outside the band get empty ranges);
 *        tiles_per_gauss i32[N] or NULL;
 *        isect_info i64[8] (device): {M, largest per-tile count, #tiles in the medium /
 *        large / merge-fallback sort classes, 0, G_on, 0}.  G_on = number of Gaussians whose
 *        tile box touches the FULL grid, whatever the band: a band-sharded caller applies
 *        the reference's frame-level "no intersections -> zeros image" rule
 *        (mojosplat/render.py:73-76) from it without a collective.  Copy the record to the
 *        host to size flatten_ids and to drive ms_isect_tiles_emit.
 *
 * ms_isect_tiles_emit: scatter (depth,id) keys into their tile segments and depth-sort
 * every segment.  Order inside a tile: ascending (float bits of depth, Gaussian index) --
 * identical to a stable radix sort of (tile<<32 | depth_bits) keys emitted in Gaussian
 * order, i.e. gsplat.isect_tiles(sort=True).
 *   in : as above plus depths f32[N]; the SAME workspace the count call filled;
 *        host_info = the 8 values of isect_info as read by the host; M = host_info[0].
 *   out: flatten_ids i32[M]; isect_ids i64[M] or NULL (sorted keys (tile<<32)|depth_bits).
 *        sort_keys u64[M] and sort_tmp u64[M or 0] are caller-allocated scratch
 *        (sort_tmp is only touched when host_info[4] > 0).
 * ------------------------------------------------------------------------------------- */
size_t ms_isect_workspace_bytes(int64_t N, int tile_w, int tile_h);

int ms_isect_tiles_count(int64_t N, const float *means2d, const int32_t *radii, int tile_size,
                         int tile_w, int tile_h, int row_begin, int row_end, void *workspace,
                         size_t workspace_bytes, int32_t *tiles_per_gauss, int32_t *tile_ranges,
                         int64_t *isect_info, void *stream);

int ms_isect_tiles_emit(int64_t N, const float *means2d, const int32_t *radii,
                        const float *depths, int tile_size, int tile_w, int tile_h,
                        int row_begin, int row_end, void *workspace, size_t workspace_bytes,
                        const int32_t *tile_ranges, const int64_t *host_info, int tight, int lazy,
                        float depth_near, float depth_far, uint64_t *sort_keys,
                        uint64_t *sort_tmp, int32_t *flatten_ids, int64_t *isect_ids,
                        void *stream);

/* ms_project_gaussians_fwd + ms_isect_tiles_count in ONE pass over the Gaussians (same outputs,
 * same semantics; the tile grid is derived from W, H, tile_size).  What ms_render_fwd starts a
 * frame with: the projected means / radii are not re-read and one kernel launch is saved.
 *
 * isect_info_mirror (nullable): a DEVICE-VISIBLE address of pinned host memory (hipHostGetDevicePointer)
 * that receives words 0..6 of the record straight from the kernel -- no copy kernel; read it after
 * an event recorded behind this call has completed.
 * `tight` is a bit set.  Bit 1 (value 2): tile_ranges is written for the tiles of the band only
 * (for a caller whose later stages all stay inside the band).  Bit 0 (needs opacities): TIGHT binning.  gsplat.isect_tiles lists every tile of a
 * Gaussian's bounding box; ~18 % of those pairs (config 3) can never blend because the
 * alpha >= 1/255 ellipse does not reach the tile (box corners, elongated / rotated footprints).
 * Tight mode drops them (per tile row, the x-extent of the ellipse inside the row's band of pixel
 * centres; boxes of more than 64 tiles are kept whole), so every later stage handles fewer
 * intersections and the IMAGE IS UNCHANGED (the dropped pairs are ones the rasteriser would skip).
 * The lists are then no longer gsplat's: tight mode is for callers that only want pixels
 * (ms_render_fwd); M and tile_ranges count the kept pairs, isect_info[6] still counts bounding
 * boxes.  The per-Gaussian reach masks stay in the workspace: the emit that follows a tight count
 * MUST pass tight = 1 (and tight = 0 after ms_isect_tiles_count or a non-tight count).
 * Bit 2 (value 4, with bit 0, even tile_size, N < 2^28): BLOCK MASKS, what ms_render_fwd's split frames
 * bin with.  The tile box of a Gaussian is derived from its gsplat box on the HALF-tile grid (bits 3 / 4:
 * that grid has 2 tile_w - 1 columns / 2 tile_h - 1 rows), reach masks are kept per half-tile cell, and
 * every emitted key carries `id << 4 | blocks` in its low word (blocks: the 2x2 half-tile blocks of
 * the tile, bit 2 dy + dx, that hold exactly what a grid of half-size tiles would list there).  The
 * emit that follows must pass the same bits.
 *
 * lazy != 0 on the emit calls: LAZY SORTING, also for pixel-only callers.  Tiles of more than 1024
 * entries get only their front (about the 1024 nearest entries) selected and sorted; the workspace
 * then holds, per tile, the length of that sorted front, and flatten_ids beyond it is undefined.
 * depth_near / depth_far (> 0, the camera planes every surviving depth lies between) let the
 * selection use fixed depth buckets and skip a pass; 0, 0 = unknown.  Only ms_render_fwd's
 * rasteriser understands such lists (it redoes a tile whose front did not saturate its pixels);
 * pass lazy = 0 for lists that anyone else reads.  Bits 1-2 of `lazy`: front level, fronts 2^level
 * times as deep (up to the 4096 entries of LDS room).  Bit 3 of `lazy` on the SPECULATIVE emit: the caller
 * expects no tile of more than 1024 entries (its previous frame had none) -- only the short lists are
 * sorted, and the caller must redo the frame if the size record then reports such tiles.
 * MOJOSPLAT_LAZY_SORT=0 in the environment makes ms_render_fwd sort fully. */
int ms_project_isect_count(int64_t N, const float *means3d, const float *scales, int scales_are_log,
                           const float *quats, const float *opacities, const float *viewmat, float fx,
                           float fy, float cx, float cy, int W, int H, float eps2d, float near_plane,
                           float far_plane, float radius_clip, int tile_size, int row_begin,
                           int row_end, int tight, float *means2d, float *conics, float *depths,
                           int32_t *radii,
                           void *workspace, size_t workspace_bytes, int32_t *tile_ranges,
                           int64_t *isect_info, int64_t *isect_info_mirror, void *stream);

/* gsplat.isect_offset_encode: from SORTED keys (tile<<32|depth_bits) to per-tile start
 * offsets i32[tile_h*tile_w] (empty tiles inherit the next start; trailing tiles get M). */
int ms_isect_offset_encode(int64_t M, const int64_t *isect_ids_sorted, int tile_w, int tile_h,
                           int32_t *offsets, void *stream);

/* ---------------------------------------------------------------------------------------
 * Rasteriser forward: per tile, front-to-back alpha compositing of its sorted list.
 *   in : means2d f32[N,2], conics f32[N,3], colors [N,CDIM] (f32 or f16, CDIM 1..32),
 *        opacities f32[N], backgrounds f32[CDIM] or NULL, tile_ranges i32[th,tw,2],
 *        flatten_ids i32[M].
 *   out: render_colors f32[H,W,CDIM]; render_alphas f32[H,W] or NULL; last_ids i32[H,W]
 *        or NULL (index into flatten_ids of the last contributing intersection).
 *   Only tiles in rows [tile_row_begin, tile_row_end) are rendered (0, tile_h = whole image;
 *   a band for multi-GPU sharding); output pointers always address the FULL image.
 * ------------------------------------------------------------------------------------- */
int ms_rasterize_to_pixels_3dgs_fwd(int64_t N, int64_t M, const float *means2d,
                                    const float *conics, const void *colors, int color_dtype,
                                    int CDIM, const float *opacities, const float *backgrounds,
                                    int W, int H, int tile_size, int tile_row_begin,
                                    int tile_row_end, const int32_t *tile_ranges,
                                    const int32_t *flatten_ids, float *render_colors,
                                    float *render_alphas, int32_t *last_ids, void *stream);

/* Rasteriser backward (gsplat rasterize_to_pixels backward semantics, absgrad off).
 *   in : forward inputs + render_alphas, last_ids from the forward,
 *        v_render_colors f32[H,W,CDIM], v_render_alphas f32[H,W] or NULL.
 *   out: v_means2d f32[N,2], v_conics f32[N,3], v_colors f32[N,CDIM], v_opacities f32[N];
 *        ACCUMULATED INTO (caller zero-fills, gsplat's convention) when overwrite == 0;
 *        OVERWRITTEN (no zero-fill needed) when overwrite != 0.                            */
int ms_rasterize_to_pixels_3dgs_bwd(int64_t N, int64_t M, const float *means2d,
                                    const float *conics, const float *colors, int CDIM,
                                    const float *opacities, const float *backgrounds, int W,
                                    int H, int tile_size, const int32_t *tile_ranges,
                                    const int32_t *flatten_ids, const float *render_alphas,
                                    const int32_t *last_ids, const float *v_render_colors,
                                    const float *v_render_alphas, float *v_means2d,
                                    float *v_conics, float *v_colors, float *v_opacities,
                                    void *workspace, size_t workspace_bytes, int overwrite,
                                    void *stream);
/* scratch for the packed-gradient path of the backward rasteriser (0 = not needed / not used:
 * without it, or for CDIM > 4, the call still works through the one-atomic-per-component path) */
size_t ms_rasterize_bwd_workspace_bytes(int64_t N, int CDIM);

/* Projection backward: gradients of (means2d, conics, depths) w.r.t. means3d, scales
 * (w.r.t. the log-scales when scales_are_log), quats.  Culled Gaussians get zero grads.
 *   out: v_means3d f32[N,3], v_scales f32[N,3], v_quats f32[N,4] (overwritten).          */
int ms_project_gaussians_bwd(int64_t N, const float *means3d, const float *scales,
                             int scales_are_log, const float *quats, const float *viewmat,
                             float fx, float fy, float cx, float cy, int W, int H, float eps2d,
                             const int32_t *radii, const float *v_means2d,
                             const float *v_conics, const float *v_depths, float *v_means3d,
                             float *v_scales, float *v_quats, void *stream);

/* Backward of a DIFFERENTIABLE ms_render_fwd frame (one that was asked for render_alphas and last_ids; declared
 * below), from the scratch that frame left behind: `workspace` / `isect_buf` / `host_info` exactly as the forward
 * call returned them (same N, image and tile size, untouched since), render_alphas / last_ids its per-pixel records.
 * One call runs the backward rasteriser -- staging from the frame's ready-made records -- and the backward
 * projection: what a training loop's backward() does between two Python statements (the reference has no backward:
 * render.py:11, README.md:145).
 *   in : the forward's inputs; v_render_colors f32[H,W,CDIM], v_render_alphas f32[H,W] or NULL;
 *        render_colors f32[H,W,CDIM]: the image that frame returned, or NULL.  With it a 3-channel frame's backward
 *        rasteriser walks the lists front to back, one wave per 8x8 quad (csrc/rasterize_bwdq.hip), and last_ids may
 *        be NULL; without it the older back-to-front kernel runs and needs last_ids.
 *   out: v_means3d f32[N,3], v_scales f32[N,3] (w.r.t. the log-scales when scales_are_log), v_quats f32[N,4],
 *        v_opacities f32[N], v_colors f32[N,CDIM] -- all OVERWRITTEN.
 *   bwd_workspace: ms_render_bwd_workspace_bytes(N, CDIM) bytes of scratch.
 *   mid_event: NULL, or a hipEvent_t recorded on `stream` between the two stages (in-situ timing). */
/* Depth cut-offs of ms_render_fwd (DEPTH CUT-OFFS below): mode 0 never / 1 from min_pairs pairs on / 2 whenever possible;
 * a negative argument leaves that setting alone.  Process-wide; the defaults come from MOJOSPLAT_DEPTH_CUT /
 * MOJOSPLAT_DEPTH_CUT_MIN_PAIRS, read once.  (For tests and measurements that switch inside one process.) */
int ms_config_depth_cut(int mode, long long min_pairs);

/* ms_render_bwd in two halves, for a multi-GPU rank's differentiable BAND frame (a 3-channel frame at a tile size that is a
 * multiple of 16, rendered with render_alphas over tile rows [tile_row_begin, tile_row_end) -- or the whole frame: 0, tile_h):
 *   ms_render_bwd_rows  : the backward rasteriser on the band's lists -> rows f32[N][16], the per-Gaussian RAW sums of the
 *                         band's pixels (zeroed by the call).  Sums over disjoint bands ADD: a rank's caller all-reduces
 *                         the rows over the ranks ("per-Gaussian grads need one all-reduce") ...
 *   ms_render_bwd_finish: ... and the backward projection turns the summed rows into the gradients, identically on every rank.
 * ms_render_bwd == rows(0, tile_h) + finish.  redo_counts_host (nullable): pinned HOST i32[2] that a lazily sorted frame's
 * redo launch fills with the frame's clean-up counts (as ms_render_redo_counts would, without a copy of its own; valid once an
 * event recorded behind the call has completed; untouched for a fully sorted frame).
 * Round 6: the workspace of ms_render_fwd ends with a region of ms_render_bwd_rows_bytes(N) bytes (at
 * ms_render_workspace_bytes(N, tile_w, tile_h) - ms_render_bwd_rows_bytes(N)) that a differentiable 3-channel frame's
 * rasteriser zeroes on its way (bit 15 of host_info[7] says it did): handed exactly that address as `rows`, the call skips its
 * own memset; any other buffer is zeroed by the call as before.
 * No reference counterpart (render.py:11; README.md:145). */
size_t ms_render_bwd_rows_bytes(int64_t N);
int ms_render_bwd_rows(int64_t N, int CDIM, int W, int H, int tile_size, int tile_row_begin, int tile_row_end,
                       const float *backgrounds, const void *workspace, size_t workspace_bytes, const void *isect_buf,
                       size_t isect_bytes, const int64_t *host_info, const float *render_colors, const float *render_alphas,
                       const float *v_render_colors, const float *v_render_alphas, float *rows, int32_t *redo_counts_host,
                       void *stream);
int ms_render_bwd_finish(int64_t N, const float *means3d, const float *scales, int scales_are_log, const float *quats,
                         const float *opacities, int CDIM, const float *viewmat, float fx, float fy, float cx, float cy,
                         int W, int H, float eps2d, const float *rows, float *v_means3d, float *v_scales, float *v_quats,
                         float *v_opacities, float *v_colors, void *stream);

/* Bins the clean-up pass of a finished ms_render_fwd frame had to redo (lazily sorted fronts that ran out with pixels
 * alive): host_counts i32[2] = {all redone bins, those redone for their depth cut-off}, copied asynchronously on `stream`
 * from the frame's `workspace` (host_counts: pinned HOST memory; valid once an event recorded behind the call has
 * completed).  Frames on cached scratch learn the same number from the NEXT frame's host_info[5]; a differentiable frame
 * owns fresh scratch, so its caller asks here (behind ms_render_bwd) and chooses MS_RENDER_FRONT_LEVEL / MS_RENDER_FULL_SORT
 * for the next step from it.  No reference counterpart (the reference has no backward: render.py:11). */
int ms_render_redo_counts(const void *workspace, size_t workspace_bytes, int64_t N, int tile_w, int tile_h,
                          int32_t *host_counts, void *stream);

size_t ms_render_bwd_workspace_bytes(int64_t N, int CDIM);
int ms_render_bwd(int64_t N, const float *means3d, const float *scales, int scales_are_log, const float *quats,
                  const float *opacities, const float *colors, int CDIM, const float *viewmat, float fx, float fy,
                  float cx, float cy, int W, int H, float eps2d, int tile_size, const float *backgrounds,
                  const void *workspace, size_t workspace_bytes, const void *isect_buf, size_t isect_bytes,
                  const int64_t *host_info, const float *render_colors, const float *render_alphas,
                  const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas, float *v_means3d,
                  float *v_scales, float *v_quats, float *v_opacities, float *v_colors, void *bwd_workspace,
                  size_t bwd_workspace_bytes, void *mid_event, void *stream);

/* ---------------------------------------------------------------------------------------
 * Spherical-harmonic colours (view dependent).  The reference leaves SH evaluation as a TODO
 * that only slices channels (mojosplat/render.py:82-87); this is the operator that belongs
 * there, in gsplat's convention (spherical_harmonics + the "+0.5, clamp at 0" of its
 * rasterization()): real SH of degree <= 4, 3DGS signs, coefficient index l*(l+1)+m.
 *   in : means3d f32[N,3]; camera position (world) cam_x/y/z; coeffs f32[N,K,3] with
 *        K >= (degree+1)^2 (only the first (degree+1)^2 are read); radii i32[N,2] or NULL:
 *        Gaussians with a zero radius are skipped (colour 0, zero gradients);
 *        add_half_and_clamp: colour = max(sum + 0.5, 0) when non-zero, the raw sum otherwise.
 *   out: colors f32|f16[N,3].
 * Backward: v_colors f32[N,3] -> v_coeffs f32[N,K,3] (overwritten; zeros beyond the degree) and
 * / or v_means3d f32[N,3] (overwritten); either may be NULL.  colors_fwd (f32, the forward
 * output) is required with add_half_and_clamp.
 * ------------------------------------------------------------------------------------- */
int ms_spherical_harmonics_fwd(int64_t N, int K, int degree, const float *means3d, float cam_x,
                               float cam_y, float cam_z, const float *coeffs, const int32_t *radii,
                               int add_half_and_clamp, int color_dtype, void *colors, void *stream);
int ms_spherical_harmonics_bwd(int64_t N, int K, int degree, const float *means3d, float cam_x,
                               float cam_y, float cam_z, const float *coeffs, const int32_t *radii,
                               int add_half_and_clamp, const float *colors_fwd, const float *v_colors,
                               float *v_coeffs, float *v_means3d, void *stream);

/* ---------------------------------------------------------------------------------------
 * Whole forward path in one call: replaces the three stage calls that
 * render_gaussians makes (mojosplat/render.py:63-101) when the caller does not need the
 * intermediates.  Projection outputs, tile ranges and the sorted list live in caller-owned
 * scratch:
 *   workspace  : ms_render_workspace_bytes(N, tile_w, tile_h) bytes, fixed per (N, image);
 *   isect_buf  : ms_render_isect_bytes(M, merge) bytes, data dependent (M = host_info[0]; a split
 *                frame -- tile_size 16, plain forward, see below -- counts the entries of its 32-px
 *                bins and needs 28 bytes for each: keys, list words, 4 block lists).  If it is too small
 *                the call returns MS_ERR_WORKSPACE with host_info[5] = bytes needed and all
 *                M-independent work done; grow the buffer and call again with resume = 1.
 *   host_info  : HOST memory (pinned), i64[8]; receives isect_info (see
 *                ms_isect_tiles_count).  This is the only entry point that waits on the
 *                device: with sync_event == NULL one hipStreamSynchronize between count and
 *                emit; with a hipEvent_t in sync_event and an isect_buf sized by an earlier
 *                frame, emit + rasterise are enqueued speculatively against the buffer's
 *                capacity BEFORE the wait, so the GPU never idles (an overflowing frame is
 *                detected and redone on the exact path).
 *   resume     : which half of the frame to run (enum below).  MS_RENDER_WHOLE = everything;
 *                MS_RENDER_RESUME = the redo after MS_ERR_WORKSPACE; MS_RENDER_BEGIN enqueues
 *                the frame (speculatively where it can) and returns WITHOUT waiting, and a
 *                later MS_RENDER_FINISH call with the same arguments does the wait, the check
 *                and, if needed, the exact redo -- so a host that renders independent frames
 *                (several views, or consecutive frames of a pipeline) can enqueue frame k+1
 *                before it waits for frame k's size record.  Every frame in flight needs its
 *                own workspace, isect_buf, host_info and sync_event; host_info[7] is the
 *                library's own slot between BEGIN and FINISH.
 *                The record a frame leaves in host_info is also the NEXT frame's hint when the same
 *                host_info is passed again (what a caller that renders frame after frame does): its M
 *                sizes nothing but launch shapes, and a previous frame without any list beyond the
 *                small sort class (1024 entries) makes the library bet that this one has none either
 *                and launch the short sorts alone -- a lost bet is read off this frame's own record
 *                and the frame redone on the exact path, like an overflowing one.  Pass a zeroed
 *                record to start afresh.
 *   stage_events: NULL, or 4 hipEvent_t recorded on `stream` at: start, after projection,
 *                after binning, after rasterisation (for in-situ kernel timing).
 * LEAN FRAMES.  A plain forward frame with CDIM == 3 keeps NO projected arrays in `workspace`: its rasteriser reads
 * the 48-byte records the count kernel leaves per Gaussian, its scatter kernel a 12-byte (tile box, depth bits, reach
 * mask) record; means2d / conics / depths / radii are only written for frames that are asked for render_alphas or
 * last_ids (what the older backward rasteriser needs) and for other channel counts.
 * DEPTH CUT-OFFS.  A lean sync-free frame on plain bins (tile_size 32 / 64, up to 255 bins a side) -- the whole grid or,
 * since round 4, a band of it (a rank's share of a frame: pre-culled or not, MS_RENDER_ROWS16 or not) --
 * whose predecessor ON THE SAME host_info / workspace / grid / band was such a frame too, and held at least 6 M pairs
 * (MOJOSPLAT_DEPTH_CUT_MIN_PAIRS; a band: that number scaled by its share of the rows), takes that predecessor's per-bin
 * depth cut-offs -- the depth at which each bin's
 * lazily sorted front ended, plus 1/16 octave -- and neither counts into the lists, scatters, sorts nor writes records
 * for the (Gaussian, bin) pairs behind them: host_info[0] still counts every pair (the buffer keeps room for them),
 * the lists hold the near ones.  The frame is exact whatever the cut-offs are: a bin whose pixels outlive its list
 * gets its dropped pairs -- and their Gaussians' records -- back in the clean-up launches (which project the
 * Gaussians again), and its cut-off is lifted for the next frame.  host_info[5] of the next record reports such bins in its
 * high 32 bits (the low 32: bins whose sorted FRONT was too short, as before); bits 6-8 and 16-47 of host_info[7]
 * are the library's bookkeeping for it (this frame took the cut; it left cut-offs, in which of two buffers, for which
 * grid and band).  MOJOSPLAT_DEPTH_CUT=0 in the environment (read once; ms_config_depth_cut changes it in-process) switches it off, =2 takes the cut whatever the
 * previous frame held.  A frame that fails its speculation is started over without the cut.
 * SPLIT FRAMES.  A plain forward frame (no render_alphas / last_ids, CDIM <= 4) at tile_size 16 over the
 * whole image or a band of >= 16 tile rows is binned on 32-px bins with block masks (see `tight`), and
 * the sort kernels cut every bin's sorted list into the lists of its four 16x16 blocks, which is what
 * the rasteriser walks: 40 % fewer (Gaussian, bin) pairs to scatter and sort, the same pairs to
 * rasterise, the same frame bit for bit.  MOJOSPLAT_SPLIT=0 in the environment bins such frames on
 * 16-px tiles directly.
 * [tile_row_begin, tile_row_end) restricts binning and rasterisation to a band of tile rows
 * (0, tile_h = whole image); render_colors always addresses the FULL image.
 * Whole-image calls: no bounding box on the grid yields a ZERO image (reference render.py:73-76),
 * otherwise render_colors f32[H,W,CDIM] = composited colours + T * background.  render_alphas
 * f32[H,W] and last_ids i32[H,W] (both nullable) are the per-pixel records the backward
 * rasteriser needs (see ms_rasterize_to_pixels_3dgs_fwd).
 * ------------------------------------------------------------------------------------- */
enum { MS_RENDER_WHOLE = 0, MS_RENDER_RESUME = 1, MS_RENDER_BEGIN = 2, MS_RENDER_FINISH = 3,
       MS_RENDER_FULL_SORT = 0x100, /* or-ed into `resume`: no lazy sorting for this frame */
       MS_RENDER_FRONT_LEVEL = 0x200, /* x 0..3, or-ed into `resume`: lazily sorted fronts 2^level times as deep
                                        (a caller whose previous frame needed the clean-up pass, host_info[5] > 0) */
       MS_RENDER_ROWS16 = 0x800, /* or-ed into `resume`: [tile_row_begin, tile_row_end) counts rows of 16 pixels whatever
                                   tile_size (a multiple of 16) is: the band is binned on the tile rows that cover
                                   it and rasterised on exactly those 16-px rows -- a multi-GPU rank keeps its band
                                   while the bin size follows the scene (32- / 64-px bins for dense scenes) */
       MS_RENDER_DEFER_CLEANUP = 0x1000 /* ms_band_frame.flags only (ms_render_fwd ignores it): the band's clean-up launches --
                                   up to four, empty on almost every frame -- are not enqueued behind the rasteriser;
                                   ms_render_band_finish waits for the END of the band instead of its size record, reads the
                                   rasteriser's verdict from word 8 of the lane's pinned record (host_info must then be
                                   i64[16]) and enqueues them only when a bin asked for them.  For bands in flight behind
                                   each other on two lanes: the GPU has the other band to run while the host waits. */ };
size_t ms_render_workspace_bytes(int64_t N, int tile_w, int tile_h);
size_t ms_render_isect_bytes(int64_t M, int with_merge_scratch);
int ms_render_fwd(int64_t N, const float *means3d, const float *scales, int scales_are_log,
                  const float *quats, const float *opacities, const void *colors, int color_dtype,
                  int CDIM, const float *viewmat, float fx, float fy, float cx, float cy, int W,
                  int H, float eps2d, float near_plane, float far_plane, int tile_size,
                  int tile_row_begin, int tile_row_end, const float *backgrounds, void *workspace,
                  size_t workspace_bytes,
                  void *isect_buf, size_t isect_bytes, int64_t *host_info, int resume,
                  float *render_colors, float *render_alphas, int32_t *last_ids,
                  void **stage_events, void *sync_event, void *stream);

/* ---------------------------------------------------------------------------------------
 * Camera batch: the same Gaussians from C cameras in one call -> render_colors f32[C,H,W,CDIM].
 * The reference's kernels carry a camera dimension (kernels/projection.mojo:32-37,
 * kernels/rasterization.mojo:66) that every wrapper pins to 1 (projection.py:431,
 * rasterization.py:175); this is the entry point with C > 1.  View v equals what ms_render_fwd
 * renders for camera v, bit for bit (incl. the zeros image for a view with nothing on the grid).
 * MI355X mapping: a frame is a VALU-bound rasteriser behind latency-bound binning kernels and one
 * host wait for its size record, so the batch keeps n_lanes (1..4, normally 2) views IN FLIGHT on
 * as many streams: view v+1's projection / binning overlaps view v's rasteriser and is enqueued
 * (MS_RENDER_BEGIN) before the host waits for view v's record (MS_RENDER_FINISH).  A pass over
 * (camera, Gaussian) in one kernel would save nothing: the projection is VALU bound, not
 * bandwidth bound.
 *   viewmats   : DEVICE f32[C,16], row-major world->camera;  intrinsics: HOST f32[C,4] = fx fy cx cy
 *   lanes      : HOST array of n_lanes scratch sets, each as ms_render_fwd wants them (workspace of
 *                ms_render_workspace_bytes, isect_buf, pinned host_info i64[8], a hipEvent_t, a
 *                hipStream_t).  The caller orders the lanes' streams after its inputs and itself after
 *                the lanes' streams.
 *   flags      : MS_RENDER_FULL_SORT / MS_RENDER_FRONT_LEVEL bits applied to every view (0 = defaults)
 *   counts     : HOST i64[C] or NULL: intersections of each view
 *   views_done : HOST, in: first view to render (0); out: views completed.  On MS_ERR_WORKSPACE the
 *                isect_buf of lane *need_lane is too small for view *views_done (*need_isect_bytes
 *                needed): nothing is in flight any more; grow it and call again with the same arrays.
 * ------------------------------------------------------------------------------------- */
typedef struct ms_view_lane {
    void *workspace;
    size_t workspace_bytes;
    void *isect_buf;
    size_t isect_bytes;
    int64_t *host_info;
    void *sync_event;
    void *stream;
} ms_view_lane;
int ms_render_fwd_batch(int C, int64_t N, const float *means3d, const float *scales, int scales_are_log,
                        const float *quats, const float *opacities, const void *colors, int color_dtype,
                        int CDIM, const float *viewmats, const float *intrinsics, int W, int H, float eps2d,
                        float near_plane, float far_plane, int tile_size, const float *backgrounds, int n_lanes,
                        const ms_view_lane *lanes, int flags, float *render_colors, int64_t *counts,
                        int *views_done, size_t *need_isect_bytes, int *need_lane);

/* ---------------------------------------------------------------------------------------
 * A multi-GPU rank's BAND of a frame, in two calls (round 5).  No reference counterpart: the reference has no distributed
 * path (mojosplat/binning.py:83 is a dead comment); what a band bins is what mojosplat/binning.py:73-84 bins, restricted
 * to the band's tile rows.
 *
 * ms_scene: the Gaussians as ms_render_fwd takes them one by one, built ONCE per scene by the caller -- plus, for a
 * PREPARED scene, the bounds of every block of block_size consecutive Gaussians (ms_scene_prepare): when the Gaussians are
 * stored in a spatially coherent order (the caller sorts them once, e.g. along a Morton curve of the means) those boxes
 * are small, the band pre-cull skips every block that cannot reach the band without reading it, and the count kernel's
 * gathers through the band's candidate list coalesce.  block_bounds == NULL: any order, no bounds (as ms_render_fwd).
 * The bounds must describe the arrays AS THEY ARE: recompute them after the means or scales change.
 *   block_bounds: the buffer ms_scene_prepare fills (ms_scene_block_bounds_bytes of it): f32[n_blocks][8] = {min x, min y, min z
 *   of the block's means, its largest LINEAR scale, max x, max y, max z, 0}, then f32[N][4] = every Gaussian's mean and largest
 *   linear scale -- the band pre-cull's own 16-byte record (one load instead of 24 bytes in two)
 *
 * ms_band_lane: one scratch set (as ms_render_fwd wants it: workspace, isect_buf, pinned host_info i64[8], sync_event) with the
 * hipStream_t the band runs on and two hipEvent_t for the ordering around it; built once per lane, isect_buf / isect_bytes
 * updated when the buffer grows.  One frame at a time per lane; frames on different lanes overlap.
 *
 * ms_render_band_begin : caller_stream -> lane (in_event), then the whole band enqueued on lane->stream (MS_RENDER_BEGIN:
 *                        speculatively, no host wait).
 * ms_render_band_finish: waits for the band's size record, checks it, redoes the band on the exact path if the speculation
 *                        did not hold (MS_ERR_WORKSPACE as ms_render_fwd: grow isect_buf to lane->host_info[5] bytes and call
 *                        again with resume = 1), then lane -> caller_stream (out_event); status i64[4] (HOST) = {Gaussians on
 *                        the grid -- of a pre-culled band: of its candidates --, 1 if the library pre-culled the band, pairs
 *                        in the band, the frame's flag word (host_info[7])}.
 * flags: MS_RENDER_ROWS16 / MS_RENDER_FULL_SORT / MS_RENDER_FRONT_LEVEL bits, as ms_render_fwd's `resume`; MS_RENDER_DEFER_CLEANUP
 *        (the lane's host_info is then i64[16]; when the verdict is clean the finishing half enqueues NOTHING: the host has
 *        seen the band end, so whatever it enqueues on caller_stream afterwards is behind it).
 * render_colors addresses image row 0 (rows outside the band are never touched), as for ms_render_fwd.
 * ------------------------------------------------------------------------------------- */
typedef struct ms_scene {
    int64_t N;
    const float *means3d, *scales;
    int scales_are_log;
    const float *quats, *opacities;
    const void *colors;
    int color_dtype, CDIM;
    const float *block_bounds;   /* NULL: not a prepared scene */
    int block_size;              /* Gaussians per block: a power of two >= 64 */
    int64_t n_blocks;
} ms_scene;
typedef struct ms_band_lane {
    void *workspace;
    size_t workspace_bytes;
    void *isect_buf;
    size_t isect_bytes;
    int64_t *host_info;
    void *sync_event, *stream, *in_event, *out_event;
} ms_band_lane;
typedef struct ms_band_frame {
    const ms_scene *scene;
    const float *viewmat;        /* DEVICE f32[16] */
    float fx, fy, cx, cy;
    int W, H;
    float eps2d, near_plane, far_plane;
    int tile_size, row_begin, row_end, flags;
    const float *backgrounds;
    float *render_colors;
    void **stage_events;         /* NULL, or 4 hipEvent_t as ms_render_fwd's (in-situ kernel timing) */
} ms_band_frame;
size_t ms_scene_block_bounds_bytes(int64_t N, int block_size);
int ms_scene_prepare(int64_t N, const float *means3d, const float *scales, int scales_are_log, int block_size,
                     float *block_bounds, void *stream);
int ms_render_band_begin(const ms_band_frame *frame, const ms_band_lane *lane, void *caller_stream);
int ms_render_band_finish(const ms_band_frame *frame, const ms_band_lane *lane, void *caller_stream, int resume,
                          int64_t *status);

/* Where ms_render_fwd keeps its intermediates inside `workspace` (byte offsets), for callers that
 * go on to differentiate the frame: offsets[0..4] = means2d f32[N,2], conics f32[N,3],
 * depths f32[N], radii i32[N,2], tile_ranges i32[tile_h,tile_w,2]; offsets[5] = total bytes.
 * The sorted Gaussian ids are the i32 array at byte offset align256(8 * capacity) of isect_buf on a
 * frame that ran sync-free (capacity = (isect_bytes - 512) / 12 entries), and at
 * align256(8 * M) * (1 + merge) on one that took the exact path (host_info[7] & 4 after the call;
 * merge = host_info[4] > 0).
 * A differentiable frame (render_alphas given, three channels, tiles of 16 or 32 px) also leaves, behind those ids, the
 * lists of its 8x8 quads for ms_render_bwd (host_info[7] & 8192): q = 4 or 16 quads a tile, 4 q more bytes per entry --
 * capacity = (isect_bytes - 768) / (12 + 4 q) on the sync-free path, + align256(4 q M) bytes on the exact one (the size
 * ms_render_fwd asks for in host_info[5] includes them). */
int ms_render_workspace_layout(int64_t N, int tile_w, int tile_h, size_t *offsets);

/* ms_isect_tiles_emit without the host knowing M: `isect_info_dev` is the DEVICE record written
 * by ms_isect_tiles_count, `capacity` the number of entries sort_keys / flatten_ids can hold.
 * Every kernel clamps to the capacity; the caller must afterwards check (from its own copy of
 * the record) that M <= capacity and that no tile is in the merge-fallback class, and redo the
 * frame with ms_isect_tiles_emit otherwise.  prev_info_host (HOST, nullable) is the record of
 * the previous frame: when its large-class count is 0 the large-class sort is not launched, and
 * the caller must also redo the frame if this frame's large-class count turns out non-zero.
 * Used by ms_render_fwd for sync-free frames. */
int ms_isect_tiles_emit_speculative(int64_t N, const float *means2d, const int32_t *radii,
                                    const float *depths, int tile_size, int tile_w, int tile_h,
                                    int row_begin, int row_end, void *workspace,
                                    size_t workspace_bytes, const int32_t *tile_ranges,
                                    const int64_t *isect_info_dev, int64_t capacity,
                                    const int64_t *prev_info_host, int tight, int lazy,
                                    float depth_near, float depth_far, uint64_t *sort_keys,
                                    int32_t *flatten_ids, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MOJOSPLAT_HIP_H */
