// Tile rasteriser (forward) for gfx950: front-to-back alpha compositing of each tile's
// depth-sorted Gaussian list.
//
// Semantics: mojosplat/kernels/rasterization.mojo:75-162 (== gsplat rasterize_to_pixels fwd):
// pixel centre (x+0.5, y+0.5); sigma = 0.5(a dx^2 + c dy^2) + b dx dy;
// alpha = min(0.999, o exp(-sigma)); skip if sigma < 0 or alpha < 1/255; stop BEFORE adding
// when T(1-alpha) <= 1e-4; out = pix + T * background.
//
// MI355X mapping.  Measured history of this kernel on BASELINE config 3 (profiles/):
//   v1  256 threads per 16x16 block, one pixel per lane (the reference's launch shape,
//       rasterization.mojo:219-220): 368 us, bound by LDS *broadcast* reads -- each of the four
//       waves re-reads every staged Gaussian, 48 LDS cycles per (tile, Gaussian);
//   v2  one wave64 per block, four pixels per lane, per-entry quad mask + scalar branches: 237 us,
//       VALU issue only 40 % busy (ten scalar branches per entry);
//   v3  (this file) per-quad ballot masks and a branch-free blend: 190 us at 80 % VALU busy; with the
//       exact ellipse-vs-quad test at staging (13 % fewer evaluations) 165 us --
//       a wave64 VALU instruction occupies its SIMD for 4 cycles (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU
//       = 4.1), so ~96 M instructions over 1024 SIMDs at 2.3 GHz are 164 us of pure issue time.
//
//   round 2  ready-made 48-byte records from the projection kernel, one wave per 8x8 QUAD (NQ = 1), compacted record
//       streams, heaviest-first launch order: 96-102 us; then the select by underflow (fp32 denormals flushed in this
//       file: alpha * 255 * 2^-126 is normal iff alpha >= 1/255), the transmittance carried as T * 2^126 so that the
//       product is never scaled back, one stop test per pair of records, the blocks of a coarse tile on one XCD:
//       15 VALU instructions per 64-pixel evaluation, 94-96 us (profiles/r02_raster_isa.md).
//   The notes below describe the one-wave-per-block shape (NQ = 4), which the per-stage API and the backward's
//   forward still use for dense lists; the kernel comment further down has the round-2 shape.
//
//   * ONE wave64 owns a 16x16 block; every lane carries four pixels (the same lane position in
//     each of the four 8x8 quads), so a staged Gaussian is read from LDS once per block;
//   * staging: 64 intersections per batch, one per lane.  The staging lane folds log2(e) into
//     the conic, takes log2(opacity), and tests which quads the alpha >= 1/255 ellipse can reach
//     (exact ellipse-vs-rectangle test: clamped 1-D minimisation on the faces nearest the mean).  The wave then votes: B[q] = ballot(entry
//     reaches quad q) -- a wave-uniform 64-bit mask per quad, no LDS;
//   * compositing: each quad walks ITS set bits in order with scalar s_ff1 (front-to-back per
//     pixel is preserved: the four quads are disjoint pixel sets).  Culled (quad, entry) pairs
//     cost nothing; one scalar branch per evaluation (the loop); v_exp_f32 is exp2, so
//     log2(alpha) comes out of one FMA chain; blend/stop are selects, and a finished pixel
//     raises its own alpha threshold to +inf, which folds "still live" into the 1/255 compare;
//   * colours sit in LDS with the geometry (the reference gathers them from global memory in
//     the per-pixel loop, rasterization.mojo:154-155);
//   * the next batch's gather (36 B per intersection from 4 arrays) and the ids of the batch
//     after it are in flight while the current batch is composited; a quad whose 64 pixels are
//     saturated is skipped, and the wave leaves when all four are (checked per batch);
//   * blockIdx -> tile: each XCD (blocks are dealt round-robin over the 8) gets runs of 8
//     consecutive tiles (neighbours share most of their Gaussians -> same L2), interleaved over
//     the whole image so the heavy centre of a frame is spread over all XCDs.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include <type_traits>
#include <algorithm>
#include <mutex>
#include <vector>

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
    int W, H, ts, tw, nsx, nsub, cdim, tile0, nblocks, ngrid, max_isects, n_gauss, parts;   // ngrid: workgroup indices in use (>= nblocks)
    const int32_t *order;  // the band's tiles (order_bins: the 32-px bins of a split frame), heaviest first; null: xcd_remap
    int order_bins, row0, row1;
    const float4 *records; // ready-made ms::RasterRecord per Gaussian (3 channels), or null: stage from the arrays
    ms::LazyLists lazy;   // front_count == nullptr: every list is fully sorted
    // A differentiable frame (round 5): every 8x8 quad leaves the Gaussians that passed its reach test, in list order, up to the
    // batch in which its last pixel stopped -- quad `qd` (block * 4 + quad) of a tile whose list is [start, start + n) writes
    // quad_lists[start * quad_nq + qd * n ...] and its count to quad_counts[tile * quad_nq + qd].  The backward rasteriser
    // (rasterize_bwdq.hip) walks these instead of testing and compacting the tile's whole list again per quad.
    int32_t *quad_lists, *quad_counts;
    int quad_nq;           // quads per tile: 4 * nsub
    float4 *zero_mem;      // round 6: memory the launch zeroes on its way (16-byte units: zero_total of them, zero_per_wave a wave), or null
    unsigned zero_per_wave;
    size_t zero_total;
    int nvb;               // persistent launch (round 6): virtual workgroup indices in all (what a one-block-per-wave launch's grid would be)
};

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kBatch = 64;

#ifdef MS_DIAG
// Diagnostic build only (scripts/diag_build.py; never the shipped library): per-wave stamps of the
// rasteriser -- start / end shader clock, evaluations, batches -- into a buffer of their own that no
// other code reads (MI355X_MICROARCH.md, DVFS give-back (6): stamps never feed an output).
__device__ unsigned long long *g_diag_stamps = nullptr;
#define MS_DIAG_ONLY(...) __VA_ARGS__
#else
#define MS_DIAG_ONLY(...)
#endif

// accumulated 255 * colour and scaled transmittance S = T * 2^126 -> the pixel's channel.  ONE spelling for the
// rasteriser and its clean-up kernel: left to the compiler, either product of `pix / 255 + T bg` may be the one
// that is fused into the addition, and the two kernels must round alike.
__device__ __forceinline__ float finish_channel(float pix255, float S, float bg) {
    const float tb = (S * ms::kTUnscale) * bg;   // (the first product is exact)
    return __builtin_fmaf(pix255, ms::kInv255, tb);
}

// The blend loop's select needs fp32 denormals FLUSHED (ms::kFlushK).  build.py compiles this file with
// -fgpu-flush-denormals-to-zero, which puts that into the kernel descriptor; the forward kernels also set the mode
// field themselves (MODE[5:4] = 0: one scalar instruction per wave), so a build that lost the flag still selects
// correctly, and tests/test_hip_parity.py::test_alpha_threshold_sits_where_the_reference_puts_it walks the threshold.
__device__ __forceinline__ void flush_fp32_denormals() {
    __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);   // hwreg(HW_REG_MODE, offset 4, size 2) = 0
}

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

// ---- the kernel ------------------------------------------------------------------------------
// CP = compile-time channel capacity (>= runtime cdim).  NQ = quads per wave: a 16x16 block belongs to ONE
// workgroup of 4 / NQ waves (NQ = 4: one wave, 4 pixels per lane; 2: two waves, upper / lower 16x8 strip;
// 1: four waves, one 8x8 quad and ONE pixel per lane each).  The waves of a workgroup are fully independent
// (private LDS slice, no workgroup barrier anywhere): they share a workgroup so that they share a CU, i.e.
// the L1 / L2 lines of the list they all stage.
// PACKED: the frame's projection kernel has left one ready-made 48-byte record per Gaussian
// (ms::RasterRecord): staging is three 16-byte gathers, the quad tests and three 16-byte LDS stores.
// The staging lanes vote: B[q] = ballot(entry reaches quad q) -- a wave-uniform 64-bit mask per
// quad.  Each quad then walks ITS set bits in order (front-to-back per pixel is preserved: the
// quads are disjoint pixel sets), so culled (quad, entry) pairs cost nothing, there is one
// scalar branch per evaluation (the loop), and blend/stop are selects.  A finished pixel
// raises its own alpha threshold to +inf, which folds the "still live" test into the
// alpha >= 1/255 compare.
//
// Round 2 (profiles/r02_raster_waves.md, per-wave stamps of scripts/raster_waves.py on config 3): with ~7
// waves resident per SIMD the round-1 kernel issued VALU at the SIMD's rate (58-65 cycles per 64-pixel
// evaluation and SIMD), but a wave ALONE needed ~400 cycles per evaluation (bit scan -> LDS round trip ->
// 7 dependent FMAs -> exp -> compares -> select, strictly serial), its heaviest waves ran 500 evaluations =
// 80 % of the kernel's duration and the last wave started at 60 % of it: mean residency 64 %.  Hence
//   * the walk is software pipelined over two register sets that take turns (loop unrolled by two, no
//     copies): the LDS reads of the next set bit fly while the current record is evaluated;
//   * with ready-made records staging is cheap enough to give every quad its own wave (NQ = 1): the longest
//     wave's work halves and twice as many, shorter waves fill the tail;
//   * blocks are launched heaviest first (A.order, built by the binning stage) instead of in image order.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

#ifndef MS_RASTER_GROUP
#define MS_RASTER_GROUP 2
#endif
#ifndef MS_RASTER_LOCAL_CONST
#define MS_RASTER_LOCAL_CONST 1
#endif
#ifndef MS_RASTER_HALF_STREAMS
#define MS_RASTER_HALF_STREAMS 0   // 1: the measurement variant of profiles/r06_raster_halfquad.md (a stream per 8x4 half-quad)
#endif
#ifndef MS_RASTER_CHAINS
#define MS_RASTER_CHAINS 0     // 1: the measurement variant of profiles/r06_raster_startup.md (waves that walk chains of blocks)
#endif
#ifndef MS_RASTER_EXPANDED
#define MS_RASTER_EXPANDED 0   // 1: the measurement variant of profiles/r05_raster_expanded.md (never the shipped library)
#endif
#ifndef MS_RASTER_UNROLL
#define MS_RASTER_UNROLL 2
#endif
#ifndef MS_RASTER_MINW_AUX
#define MS_RASTER_MINW_AUX 7   // variants that keep per-pixel records: 8 waves per SIMD made them spill (two quads per wave: 5)
#endif
#ifndef MS_RASTER_MINW2
#define MS_RASTER_MINW2 5   // two quads per wave (96 VGPRs)
#endif
#ifndef MS_RASTER_MINW
#define MS_RASTER_MINW 8
#endif
constexpr int kGroup = MS_RASTER_GROUP;   // records evaluated per trip of the blend loop
constexpr int kUnroll = MS_RASTER_UNROLL; // trips unrolled

// One LDS block per (wave, quad): the records of the entries that REACH the quad, compacted in list order.
// The three arrays sit at fixed offsets, so a record's words share one index * 16 B and the blend loop
// reaches kGroup consecutive records from ONE address register through the instructions' offset fields.
// (colours first: a pair's two blues are then within ds_read2_b32's 1 KB offset reach of the same address)
// CP == 3: r, g ride in the second word; blue is a float of its own per slot (with the index in the batch beside
// it for a frame that keeps per-pixel records): a pair's two blues are ONE ds_read_b64, and staging writes 36 B
// per record, not 48 (the LDS array is 73 % busy in this kernel, SQ_LDS_IDX_ACTIVE)
// (measured against the 16-byte colour slot of before: 94.4 / 94.5 us against 94.2 / 94.8 -- no difference in time,
// 25 % less LDS per wave)
template <int CP, bool AUX>
struct RasterStage {
    static constexpr int CS = (CP == 3) ? (AUX ? 2 : 1) : CP;
    static constexpr int kSlots = kBatch + kGroup;      // room for the neutral records that pad a list to a multiple of kGroup
    __attribute__((aligned(16))) float col[(kSlots * CS + 3) & ~3];  // CP == 3: blue (, index in batch)
    float4 a[kSlots];        // mean.x, mean.y, a', b'
    float4 b[kSlots];        // c', log2(opacity), (r, g | index in batch, -)
};

// Round 6: PERSISTENT waves (profiles/r06_raster_startup.md).  A wave that is its own workgroup pays its start-up chain --
// tile range -> ids -> records -> LDS, three dependent round trips -- alone, before its first blend: 48 % of all wave cycles at
// config 2, 28 % on an edge band of config 5, 14 % at config 3 (scripts/raster_waves.py).  A persistent wave walks several
// blocks (virtual workgroup indices vb, vb + grid, ...) and issues the NEXT block's first ids and records where the current
// block's list has none left to prefetch -- into the very registers the within-list prefetch uses, so no register is added:
//   carry = 1: r_g holds the next block's first batch of ids;  2: r_a / r_b / r_c its records and r_g its second batch of ids.
struct RasterNext {
    int carry;            // in: what the previous block of this wave left for THIS block; out: what this block leaves
    int valid;            // a next block exists and its list is not empty
    int start, end;       // the next block's list (as raster_tile will compute it)
};

// a tile's list as the rasteriser walks it: [start, end) of the sorted ids, end_all = where the whole list ends
__device__ __forceinline__ void raster_list_bounds(const RasterArgs &A, const int tile, int &start, int &end, int &end_all) {
    // clamped to the list length the caller vouches for (a sync-free frame passes its buffer capacity)
    end_all = min(A.tile_ranges[2 * tile + 1], A.max_isects);
    start = min(A.tile_ranges[2 * tile], end_all);
    // lazily sorted frame: a heavy tile's list is only sorted up to its front
    end = end_all;
    if (A.lazy.front_count && end_all - start > A.lazy.front_threshold) end = start + min(A.lazy.front_count[tile], end_all - start);
}

// (a band cut at 16-px rows inside coarser tiles: blocks outside it are not this call's)
__device__ __forceinline__ bool raster_block_in_band(const RasterArgs &A, const int tile, const int sub) {
    if (A.order_bins) return true;
    const int by16 = (tile / A.tw) * A.nsx + sub / A.nsx;
    return by16 >= A.row0 && by16 < A.row1;
}

// One wave's share of a 16x16 block: NQ quads of block `sub` of tile `tile`, starting at quad part * NQ.  s_q: the
// wave's NQ staging blocks.
// HALF (round 6, NQ == 1 only): the wave's two 32-lane halves -- rows 0-3 and rows 4-7 of its 8x8 quad -- walk compacted
// streams of their OWN: an entry that reaches only one half costs the other half nothing, and the loop runs for the longer
// of the two streams (profiles/r06_raster_halfquad.md).
template <int CP, typename ColorT, bool AUX, int NQ, bool PACKED, bool LISTS = false, bool PERSIST = false, bool HALF = false>
__device__ __forceinline__ void raster_tile(const RasterArgs &A, const int tile, const int sub, const int part,
                                            RasterStage<CP, AUX> *s_q, const int diag_slot,
                                            float4 &r_a, float4 &r_b, float4 &r_c, int &r_g, RasterNext &nx) {
    static_assert(!PERSIST || (PACKED && !AUX && !LISTS), "persistent waves: the plain kernel on ready-made records");
    static_assert(!HALF || (NQ == 1 && PACKED && !AUX && !LISTS && !PERSIST && CP == 3), "half-quad streams: the plain one-quad kernel");
    constexpr int NS = HALF ? 2 : NQ;   // compacted streams a wave keeps (and votes on)
    static_assert(!LISTS || (PACKED && !AUX), "quad lists: the plain 3-channel kernel on ready-made records");
    static_assert(!PACKED || CP == 3, "ready-made records carry three channels");
    using Stage = RasterStage<CP, AUX>;
    constexpr int CS = Stage::CS;
    MS_DIAG_ONLY(const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime(), diag_c0 = __builtin_amdgcn_s_memtime(); unsigned diag_evals = 0, diag_batches = 0; unsigned long long diag_first = 0;)
    const int qbase = NQ == 4 ? 0 : part * NQ;
    const int tile_y = tile / A.tw, tile_x = tile - tile_y * A.tw;
    const int sub_y = sub / A.nsx, sub_x = sub - sub_y * A.nsx;
    const int have = PERSIST ? nx.carry : 0;   // (wave-uniform) what the wave's previous block fetched of this one's list
    if constexpr (PERSIST) nx.carry = 0;
    if (!raster_block_in_band(A, tile, sub)) {   // a band cut at 16-px rows inside coarser tiles: blocks outside it are not this call's
        if constexpr (LISTS) {   // (nothing of this block is rendered: its quads leave empty lists)
            if ((threadIdx.x & 63) < NQ) A.quad_counts[(size_t)tile * A.quad_nq + sub * 4 + qbase + (threadIdx.x & 63)] = 0;
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    const int lx = lane & 7, ly = lane >> 3;
    const int bx = tile_x * A.ts + sub_x * 16, by = tile_y * A.ts + sub_y * 16;
    const int ox = sub_x * 16 + lx, oy = sub_y * 16 + ly;
    const float px0 = (float)(bx + lx) + 0.5f, py0 = (float)(by + ly) + 0.5f;
#if MS_RASTER_EXPANDED
    const float ux = (float)lx - 3.5f, uy = (float)ly - 3.5f;   // the pixel centre's offset from its quad's centre
#endif

    constexpr float kInf = __builtin_huge_valf();
    // kq: the lane's multiplier of the flush select (ms::kFlushK while the pixel is live, 0 once it has stopped
    // or when it lies outside the image: alpha * 0 = 0, so a finished pixel blends nothing and keeps its T)
    float T[NQ], kq[NQ], pix[NQ][CP];
    int last[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        const int q = qbase + qi;
        const int X = bx + lx + (q & 1) * 8, Y = by + ly + (q >> 1) * 8;
        const bool in = (ox + (q & 1) * 8) < A.ts && (oy + (q >> 1) * 8) < A.ts && X < A.W && Y < A.H;
        T[qi] = ms::kTScale;   // the transmittance is carried as S = T * 2^126 (see the blend loop)
        kq[qi] = in ? ms::kFlushK : 0.f;
        last[qi] = 0;
#pragma unroll
        for (int k = 0; k < CP; ++k) pix[qi][k] = 0.f;
    }

    int start, end, end_all;
    raster_list_bounds(A, tile, start, end, end_all);
    const ColorT *colors = reinterpret_cast<const ColorT *>(A.colors);
    const float fbx = (float)bx + 0.5f, fby = (float)by + 0.5f;
    // (here rather than at the kernel's first line: s_setreg is a scheduling barrier, and up there it cost the
    // headline variant a spilled VGPR -- 8 MB of scratch writes per frame, WRITE_SIZE 25.7 -> 33.6 MB)
    flush_fp32_denormals();

    // what the staging lane holds of its entry: the record's three words (PACKED: r_a / r_b / r_c, the caller's -- a persistent
    // wave carries them from block to block), or the per-stage fields
    float r_ca = 0.f, r_cb = 0.f, r_cc = 0.f, r_op = 0.f;
    float r_col[CP];
    // LISTS: the Gaussian whose record r_a / r_b / r_c hold; how many entries each quad has left so far; whether it still has
    // a live pixel (wave-uniform)
    int g_staged = 0;
    unsigned long long q_at[NQ];   // where the quad's next entry goes (element index into quad_lists: scalar; 64 slots a pair on 64-px bins pass 2^32 from 67 M pairs on)
    bool q_live[NQ];
    if constexpr (LISTS) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            q_live[qi] = __any(kq[qi] != 0.f);
            q_at[qi] = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(start) * (unsigned long long)(unsigned)A.quad_nq +
                       (unsigned long long)(unsigned)(sub * 4 + qbase + qi) * (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(end_all - start);
        }
    }
    unsigned long long q_at0[NQ];
    if constexpr (LISTS) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) q_at0[qi] = q_at[qi];
    }
    auto fetch_id = [&](int b0) {
        const int idx = b0 + lane;
        r_g = idx < end ? A.flatten_ids[idx] : 0;
    };
    auto gather = [&](int b0) {
        const int idx = b0 + lane;
        if (idx < end) {
            // ids outside [0, N) can only come from a sync-free frame that overflowed its buffer (the
            // frame is then redone); clamp so that even that frame never reads out of bounds
            const int g = min(max(r_g, 0), A.n_gauss - 1);
            if constexpr (LISTS) g_staged = g;
            if constexpr (PACKED) {
                const float4 *rec = A.records + 3 * (size_t)g;
                r_a = rec[0]; r_b = rec[1]; r_c = rec[2];
            } else {
                const float2 m = reinterpret_cast<const float2 *>(A.means2d)[g];
                r_a.x = m.x; r_a.y = m.y;
                r_ca = A.conics[3 * g]; r_cb = A.conics[3 * g + 1]; r_cc = A.conics[3 * g + 2];
                r_op = A.opacities[g];
#pragma unroll
                for (int k = 0; k < CP; ++k) r_col[k] = k < A.cdim ? load_color(colors + (size_t)g * A.cdim + k) : 0.f;
            }
        }
    };

    // PERSIST: the next block's first ids go where this list has nothing left to fetch ahead (r_g is free then); its records
    // follow on this list's last batch, once that batch is staged (r_a / r_b / r_c are free then)
    bool rg_next = false;   // r_g holds (or is loading) the next block's first batch of ids
    int rg_age = 0;         // blend loops since they were asked for: >= 1 -> they have landed behind one
    auto fetch_ahead = [&](int b) __attribute__((always_inline)) {
        if (b < end) fetch_id(b);
        else if constexpr (PERSIST) {
            if (nx.valid && !rg_next) {
                const int idx = nx.start + lane;
                r_g = idx < nx.end ? A.flatten_ids[idx] : 0;
                rg_next = true;
                rg_age = 0;
            }
        }
    };
    auto gather_next = [&]() __attribute__((always_inline)) {
        if constexpr (PERSIST) {
            const int idx = nx.start + lane;
            if (idx < nx.end) {
                const int g = min(max(r_g, 0), A.n_gauss - 1);
                const float4 *rec = A.records + 3 * (size_t)g;
                r_a = rec[0]; r_b = rec[1]; r_c = rec[2];
            }
            const int idx2 = idx + kBatch;
            r_g = idx2 < nx.end ? A.flatten_ids[idx2] : 0;
            rg_next = false;
            nx.carry = 2;
        }
    };
    if (start < end) {
        if (have == 0) fetch_id(start);
        if (have <= 1) gather(start);
        if (have <= 1 || start + kBatch >= end) fetch_ahead(start + kBatch);
    }
    for (int b0 = start; b0 < end; b0 += kBatch) {
        // --- stage this batch: reach of the alpha >= 1/255 ellipse -> quad votes -> compacted records in LDS
        // A quad can blend this Gaussian iff min over the quad's pixel-centre rectangle of
        // sigma(p - mean) <= ln(255 o).  sigma is a convex quadratic, so the minimum over a box
        // is 0 if the mean is inside, else it sits on the face(s) nearest the mean: one 1-D
        // clamped minimisation per such face (exact; tighter than the ellipse's bounding box
        // for elongated / rotated Gaussians).  Slack keeps the test conservative w.r.t. the
        // per-pixel alpha >= 1/255 decision.  (In log2 units on the record's a', b', c'.)
        int mask = 0;
        bool npd = false;
        if (b0 + lane < end) {
            if constexpr (!PACKED) {
                const ms::RasterRecord rr = ms::make_raster_record(r_a.x, r_a.y, r_ca, r_cb, r_cc, r_op, 0.f, 0.f, 0.f);
                r_a = rr.a; r_b = rr.b; r_c = rr.c;
            }
            const float smax = r_c.y, nb_c = r_c.z, nb_a = r_c.w;
            if (smax == kInf) {
                mask = 0xf;   // not positive definite, or the 0.999 clamp can bind: no bound, generic loop
                npd = true;
            } else if (smax > -kInf) {
#pragma unroll
                for (int qi = 0; qi < NS; ++qi) {
                    const int q = HALF ? qbase : qbase + qi;
                    const float xl = fbx + (float)((q & 1) * 8) - r_a.x, xh = xl + 7.0f;   // rectangle - mean
                    const float yl = fby + (float)((q >> 1) * 8 + (HALF ? 4 * qi : 0)) - r_a.y, yh = yl + (HALF ? 3.0f : 7.0f);
                    const bool in_x = xl <= 0.f && xh >= 0.f, in_y = yl <= 0.f && yh >= 0.f;
                    float best = (in_x && in_y) ? 0.f : 3.0e38f;
                    if (!in_x) {
                        const float dx = xl > 0.f ? xl : xh;
                        const float dy = fminf(fmaxf(nb_c * dx, yl), yh);
                        best = -(r_a.z * dx * dx + r_b.x * dy * dy + r_a.w * dx * dy);
                    }
                    if (!in_y) {
                        const float dy = yl > 0.f ? yl : yh;
                        const float dx = fminf(fmaxf(nb_a * dy, xl), xh);
                        best = fminf(best, -(r_a.z * dx * dx + r_b.x * dy * dy + r_a.w * dx * dy));
                    }
                    mask |= (best <= smax) ? (1 << qi) : 0;
                }
            }
        }
        unsigned long long B[NS];
#pragma unroll
        for (int qi = 0; qi < NS; ++qi) B[qi] = __ballot((mask >> qi) & 1);
        int n_loop = 0;   // HALF: the longer of the two streams
        if constexpr (HALF) n_loop = max(__popcll(B[0]), __popcll(B[1]));
#if MS_RASTER_EXPANDED
        // (measurement variant, round 5: log2(alpha) as the quadratic EXPANDED about the quad's centre -- five FMAs on two
        // lane constants instead of two subtractions + five: the staging lane leaves a', b', c', D, E, F per reached quad.
        // Batches that need the sigma >= 0 test keep the plain records.)
        const bool expanded = PACKED && __ballot(npd) == 0;
#endif

        wave_lds_sync();  // LDS reads of the previous batch are complete
        // every reached quad gets the record at its rank among the quad's entries (list order is kept); the
        // index in the batch rides along for last_ids; kGroup neutral records (alpha = 0) close each list
        if constexpr (!PACKED) {
            if constexpr (CP == 3) { r_b.z = r_col[0]; r_b.w = r_col[1]; r_c.x = r_col[2]; }
        }
        if constexpr (CP == 3) r_c.y = __int_as_float(lane);
        else r_b.z = __int_as_float(lane);
#pragma unroll
        for (int qi = 0; qi < NS; ++qi) {
            Stage &S = s_q[qi];
            const unsigned long long b = B[qi];
            const int n = __popcll(b);
            if ((mask >> qi) & 1) {
                const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
                if constexpr (LISTS) {
                    if (q_live[qi]) A.quad_lists[q_at[qi] + (unsigned long long)(unsigned)pos] = g_staged;
                }
#if MS_RASTER_EXPANDED
                if (expanded) {
                    const int q = qbase + qi;
                    const float U = r_a.x - ((float)(bx + (q & 1) * 8) + 4.0f), V = r_a.y - ((float)(by + (q >> 1) * 8) + 4.0f);
                    const float D = -fmaf(2.0f * r_a.z, U, r_a.w * V), E = -fmaf(2.0f * r_b.x, V, r_a.w * U);
                    const float F = fmaf(U, fmaf(r_a.z, U, r_a.w * V), fmaf(r_b.x * V, V, r_b.y));
                    S.a[pos] = make_float4(r_a.z, r_a.w, r_b.x, D);
                    S.b[pos] = make_float4(E, F, r_b.z, r_b.w);
                } else {
                    S.a[pos] = r_a;
                    S.b[pos] = r_b;
                }
#else
                S.a[pos] = r_a;
                S.b[pos] = r_b;
#endif
                if constexpr (CP == 3) {
                    if constexpr (AUX) reinterpret_cast<float2 *>(S.col)[pos] = make_float2(r_c.x, r_c.y);
                    else S.col[pos] = r_c.x;
                }
                else {
#pragma unroll
                    for (int k = 0; k < CP; ++k) S.col[pos * CS + k] = r_col[k];
                }
            }
            if constexpr (LISTS) {
                if (q_live[qi]) q_at[qi] += (unsigned long long)(unsigned)n;
            }
            // (HALF: the shorter stream is padded with neutral records up to the longer one's end: the halves walk in lockstep)
            if (HALF ? (n + lane < n_loop + kGroup) : (lane < kGroup)) {
                // (the two constants are made HERE: hoisted out of the batch loop as eight registers of zeros and -inf they
                // were the first thing the persistent kernel spilled)
                float zero = 0.f, ninf = -kInf;
#if MS_RASTER_LOCAL_CONST
                asm volatile("" : "+v"(zero), "+v"(ninf));
#else
                if constexpr (PERSIST) asm volatile("" : "+v"(zero), "+v"(ninf));
#endif
                S.a[n + lane] = make_float4(zero, zero, zero, zero);
                S.b[n + lane] = make_float4(zero, ninf, zero, zero);   // log2(alpha) = -inf: alpha = 0, never a hit
                if constexpr (CP == 3) {
                    if constexpr (AUX) reinterpret_cast<float2 *>(S.col)[n + lane] = make_float2(0.f, 0.f);
                    else S.col[n + lane] = zero;
                }
                else {
#pragma unroll
                    for (int k = 0; k < CP; ++k) S.col[(n + lane) * CS + k] = 0.f;
                }
            }
        }
        wave_lds_sync();
        if (b0 + kBatch < end) {  // next batch's data and the one after's ids fly during compositing
            gather(b0 + kBatch);
            fetch_ahead(b0 + 2 * kBatch);
        } else if constexpr (PERSIST) {
            // the list's last batch is staged: the next block's records fly during its compositing (ids asked for a blend
            // loop ago have landed; a one-batch list's are still in flight -- its records go out behind the loop instead)
            if (rg_next && rg_age >= 1) gather_next();
        }
        MS_DIAG_ONLY(if (diag_batches == 0) diag_first = __builtin_amdgcn_s_memtime() - diag_c0; ++diag_batches;)   // (first batch staged: the start-up chain range -> ids -> records -> LDS is behind the wave)

        // sigma < 0 (skipped by the reference, rasterization.mojo:144) cannot happen for a positive
        // definite conic, and alpha = o exp(-sigma) <= o cannot exceed 0.999 unless o does: only a
        // batch holding such an entry pays for the sigma compare and the clamp
        const bool check_sigma = __ballot(npd) != 0;
        bool any_live = false;
        auto blend_batch = [&](auto check) __attribute__((always_inline)) {
            constexpr bool CHECK = decltype(check)::value;
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                const int q = qbase + qi;
                if (!__any(kq[qi] != 0.f)) continue;  // every pixel of this quad is finished (or outside)
                const float px = px0 + (float)((q & 1) * 8), py = py0 + (float)((q >> 1) * 8);
                const Stage &S = s_q[HALF ? (lane >> 5) : qi];   // (HALF: a per-lane LDS base, the same offsets)
                const int n = HALF ? n_loop : __popcll(B[qi]);
                // kGroup records per trip: their LDS reads go out together, the kGroup log2(alpha) chains and
                // exp2 are independent of each other (and of the T chain), then the blends run in list order
                // one trip: kGroup records from LDS address k0, transmittance t in and out (by value, so that two
                // unrolled trips take turns on two registers instead of copying T back at the loop's end)
                auto trip = [&](const int k0, float t) __attribute__((always_inline)) -> float {
                    float4 ra[kGroup], rb[kGroup];
                    float rc[kGroup][CP == 3 ? 1 : CP];
                    int rt[kGroup];
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) {
                        ra[j] = S.a[k0 + j];
                        rb[j] = S.b[k0 + j];
                        if constexpr (CP == 3) {
                            if constexpr (AUX) {
                                const float2 c2 = reinterpret_cast<const float2 *>(S.col)[k0 + j];
                                rc[j][0] = c2.x; rt[j] = __float_as_int(c2.y);
                            } else rc[j][0] = S.col[(k0 + j) * CS];
                        } else {
#pragma unroll
                            for (int c = 0; c < CP; ++c) rc[j][c] = S.col[(k0 + j) * CS + c];
                            if constexpr (AUX) rt[j] = __float_as_int(rb[j].z);
                        }
                    }
                    // The select without a compare (v_cmp / v_cndmask issue at half the FMA rate on gfx950): this
                    // file is compiled with fp32 denormals flushed, and m = alpha * (255 * 2^-126) is a normal number
                    // iff alpha >= fl(1/255) -- bit for bit the reference's test (scripts/ubench/flush_select.hip
                    // checks every float around the threshold) -- and exactly 0 otherwise (kq = 0: a finished pixel).
                    // m is never scaled back: the transmittance is carried as S = T * 2^126 (exact; S <= 8.5e37), so
                    // v = m * S = 255 alpha T is the blend weight times 255, S (1 - alpha) = S - v * (2^126 / 255) is ONE fma,
                    // and the accumulated colours are divided by 255 once per pixel at the end: six VALU instructions
                    // per evaluation behind the exp2 instead of nine.
                    float m[kGroup], v[kGroup];
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) {
#if MS_RASTER_EXPANDED
                        float la;
                        if constexpr (!CHECK && PACKED) {
                            // a' ux^2 + b' ux uy + c' uy^2 + D ux + E uy + F, Horner: (a, b, c, D) | (E, F, ...)
                            la = fmaf(ux, fmaf(ra[j].x, ux, fmaf(ra[j].y, uy, ra[j].w)), fmaf(uy, fmaf(ra[j].z, uy, rb[j].x), rb[j].y));
                        } else {
                            const float dx = ra[j].x - px, dy = ra[j].y - py;
                            la = fmaf(dx, fmaf(ra[j].z, dx, ra[j].w * dy), fmaf(rb[j].x * dy, dy, rb[j].y));
                        }
#else
                        const float dx = ra[j].x - px, dy = ra[j].y - py;
                        // log2(alpha) = log2(o) - sigma*log2(e), with log2(o) riding in the FMA chain
                        const float la = fmaf(dx, fmaf(ra[j].z, dx, ra[j].w * dy), fmaf(rb[j].x * dy, dy, rb[j].y));
#endif
                        float alpha = __builtin_amdgcn_exp2f(la);
                        if constexpr (CHECK) {
                            alpha = fminf(ms::kMaxAlpha, alpha);
                            alpha = la <= rb[j].y ? alpha : 0.f;                    // sigma >= 0
                        }
                        m[j] = alpha * kq[qi];
                        // (keeps the group's products apart: v_pk_mul_f32 costs 2.2 v_mul_f32 on gfx950 and wants
                        // its operands in register pairs)
                        asm volatile("" : "+v"(m[j]));
                    }
                    // S after each record of the group; it only shrinks along the group, so ONE test of the last
                    // value tells whether any pixel stops inside it
                    const float t_in = t;
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) {
                        MS_DIAG_ONLY(++diag_evals;)
                        v[j] = m[j] * t;
                        t = fmaf(v[j], -ms::kAlphaOfV, t);                               // S (1 - alpha)
                    }
                    // stop BEFORE adding: the pixel is finished.  Happens once per pixel -> rare path, which
                    // redoes the group's chain with the stopping record (and what follows it) taken out.
                    if (__ballot(!(t > ms::kTransmittanceStop * ms::kTScale))) {
                        asm volatile("" ::: "memory");  // keep this a real (rarely taken) scalar branch
                        t = t_in;
                        bool dead = false;
#pragma unroll
                        for (int j = 0; j < kGroup; ++j) {
                            const float vj = m[j] * t;
                            const float nt = fmaf(vj, -ms::kAlphaOfV, t);
                            dead = dead || !(nt > ms::kTransmittanceStop * ms::kTScale);
                            v[j] = dead ? 0.f : vj;
                            t = dead ? t : nt;
                        }
                        kq[qi] = dead ? 0.f : kq[qi];
                    }
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) {
                        if constexpr (CP == 3) {
                            pix[qi][0] += rb[j].z * v[j];
                            pix[qi][1] += rb[j].w * v[j];
                            pix[qi][2] += rc[j][0] * v[j];
                        } else {
#pragma unroll
                            for (int c = 0; c < CP; ++c) pix[qi][c] += rc[j][c] * v[j];
                        }
                        if constexpr (AUX) last[qi] = v[j] != 0.f ? b0 + rt[j] : last[qi];
                    }
                    return t;
                };
                float t = T[qi];
                int k0 = 0;
                if constexpr (kUnroll == 2) {
                    for (; k0 + kGroup < n; k0 += 2 * kGroup) {
                        const float u = trip(k0, t);
                        t = trip(k0 + kGroup, u);
                    }
                }
                for (; k0 < n; k0 += kGroup) t = trip(k0, t);
                T[qi] = t;
                const bool ql = __any(kq[qi] != 0.f);
                if constexpr (LISTS) q_live[qi] = ql;
                any_live = any_live || ql;
            }
        };
        if (check_sigma) blend_batch(std::true_type{});
        else blend_batch(std::false_type{});
        if constexpr (PERSIST) {
            ++rg_age;
            if (rg_next && b0 + kBatch >= end) gather_next();
        }
        if (!any_live) break;
    }
    if constexpr (PERSIST) {
        if (rg_next) nx.carry = 1;   // (the walk ended early: the ids are all the next block gets)
    }
    if constexpr (LISTS) {
        if (lane == 0) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) A.quad_counts[(size_t)tile * A.quad_nq + sub * 4 + qbase + qi] = (int)(q_at[qi] - q_at0[qi]);
        }
    }

#ifdef MS_DIAG
    if (g_diag_stamps && lane == 0) {
        unsigned long long *d = g_diag_stamps + 8 * (size_t)diag_slot;
        d[0] = diag_t0;                               // 100 MHz, chip-wide
        d[1] = __builtin_amdgcn_s_memrealtime();
        d[4] = __builtin_amdgcn_s_memtime() - diag_c0;  // shader cycles of this wave's life
        d[5] = diag_first;                              // ... of which before its first blend loop (0: an empty list)
        d[2] = ((unsigned long long)diag_batches << 32) | diag_evals;
        d[3] = ((unsigned long long)(end_all - start) << 32) | (unsigned)(__builtin_amdgcn_s_getreg((4 << 11) | 20) & 0xf) << 24 | (unsigned)(tile & 0xffffff);
    }
#endif
    int redo_tile = end < end_all ? tile : -1;  // wave-uniform; only heavy tiles of a lazily sorted frame
    // (depth-cut frame: ... and tiles whose list is short of the pairs behind the tile's cut-off)
    bool cut_short = false;
    if (A.lazy.cut_stamp && redo_tile < 0 && A.lazy.has_far[tile] == A.lazy.cut_stamp) { redo_tile = tile; cut_short = true; }
    if (A.lazy.bin_more) {   // split frame: this block's list was cut from the front of its 32-px bin
        const int bin = (tile_y >> 1) * A.lazy.bin_w + (tile_x >> 1);
        if (A.lazy.bin_more[bin]) redo_tile = bin;
    }
    if (redo_tile >= 0) {
        bool alive = false;
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) alive = alive || __any(kq[qi] != 0.f);
        if (alive) {
            // pixels outlived the sorted front: the clean-up kernel redoes the whole tile (and
            // overwrites what is stored below); once per tile, whichever wave gets there first
            if (lane == 0 && atomicExch(&A.lazy.redo_flag[redo_tile], 1) == 0) {
                A.lazy.redo_list[atomicAdd(A.lazy.redo_count, 1)] = redo_tile;
                // (deferred clean-up: the host looks at this word of its pinned record once the launch has ended)
                if (A.lazy.verdict) *(volatile int32_t *)A.lazy.verdict = 1;
                if (cut_short) atomicAdd(A.lazy.redo_count + 1, 1);   // (reported apart: the caller's front depth is not to blame)
            }
        }
    }

#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        const int q = qbase + qi;
        const int X = bx + lx + (q & 1) * 8, Y = by + ly + (q >> 1) * 8;
        const bool in = (ox + (q & 1) * 8) < A.ts && (oy + (q >> 1) * 8) < A.ts && X < A.W && Y < A.H;
        if (!in) continue;
        const size_t p = (size_t)Y * A.W + X;
#pragma unroll
        for (int k = 0; k < CP; ++k)
            if (k < A.cdim)
                A.render_colors[p * A.cdim + k] = finish_channel(pix[qi][k], T[qi], A.backgrounds ? A.backgrounds[k] : 0.f);
        // (round 4: the alphas alone need no AUX variant -- a differentiable frame whose backward is the quad-wave kernel
        // asks for nothing else, and runs the plain kernel)
        if (A.render_alphas) A.render_alphas[p] = 1.0f - T[qi] * ms::kTUnscale;
        if constexpr (AUX) {
            if (A.last_ids) A.last_ids[p] = last[qi];
        }
    }
}

// virtual workgroup index -> (16x16 block of a tile, which of the block's waves); false: nothing to rasterise there
template <int NQ>
__device__ __forceinline__ bool raster_map_block(const RasterArgs &A, const int vb, int &tile, int &sub, int &part, int &slot) {
    // Every wave is a workgroup of its own (wave slots refill one by one; four-wave workgroups measured the same
    // kernel time at 78 % instead of 86 % residency); the 4 / NQ waves of a block sit at index b, b + 8, b + 16,
    // ...: dealt round-robin over the 8 XCDs, they land on ONE XCD back to back and share its L2 for the list they
    // all stage (speed only, never correctness)
    constexpr int kParts = 4 / NQ;
    int wg = vb;   // workgroup index in units of blocks, and which of the block's waves this is
    part = 0;
    if constexpr (kParts > 1) {
        const int j = vb >> 3;
        part = j % kParts;
        wg = ((j / kParts) << 3) | (vb & 7);
        if (wg >= A.ngrid) return false;
    }
    slot = wg * kParts + part;
    // index -> 16x16 block: heaviest first when the binning stage has left an order, else image order
    // interleaved over the XCDs
    if (A.order && A.order_bins) {   // split frame: the order lists 32-px bins, four workgroups (blocks) per bin
        const int e = A.order[wg >> 2], sb = wg & 3;
        const int by16 = 2 * (e / A.lazy.bin_w) + (sb >> 1), bx16 = 2 * (e % A.lazy.bin_w) + (sb & 1);
        if (bx16 >= A.tw || by16 < A.row0 || by16 >= A.row1) return false;   // (uniform per workgroup; no barrier anywhere)
        tile = by16 * A.tw + bx16;
        sub = 0;
    } else if (A.order) {
        // the nsub 16x16 blocks of a coarse tile on ONE XCD (wg & 7 labels it), back to back: they gather the
        // same records, so the tile's list and records cross into that L2 once instead of nsub times
        const int e = ((wg >> 3) / A.nsub) * 8 + (wg & 7);
        if (e >= A.nblocks / A.nsub) return false;   // (uniform per workgroup; the grid is padded to 8 tiles)
        tile = A.order[e];
        sub = (wg >> 3) % A.nsub;
    } else {
        const int item = xcd_remap(wg, A.nblocks);
        const int bt = item / A.nsub;
        sub = item - bt * A.nsub;
        tile = A.tile0 + bt;
    }
    return true;
}

template <int CP, typename ColorT, bool AUX, int NQ, bool PACKED, bool LISTS = false, bool PERSIST = false, bool HALF = false>
__global__ __launch_bounds__(64, (CP <= 4 ? (NQ == 2 ? MS_RASTER_MINW2 : AUX ? MS_RASTER_MINW_AUX : MS_RASTER_MINW) : 1)) void k_rasterize_fwd(RasterArgs A) {
    __shared__ RasterStage<CP, AUX> s_stage[HALF ? 2 : NQ];
    float4 r_a = make_float4(0.f, 0.f, 0.f, 0.f), r_b = r_a, r_c = r_a;
    int r_g = 0;
    RasterNext nx{0, 0, 0, 0};
    if (A.zero_mem) {   // (uniform) this wave's slice of the memory the launch zeroes: stores nobody waits for
        const size_t i0 = (size_t)blockIdx.x * A.zero_per_wave;
        for (unsigned i = threadIdx.x & 63u; i < A.zero_per_wave; i += 64u)
            if (i0 + i < A.zero_total) A.zero_mem[i0 + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (!PERSIST) {
        int tile, sub, part, slot;
        if (!raster_map_block<NQ>(A, (int)blockIdx.x, tile, sub, part, slot)) return;
        raster_tile<CP, ColorT, AUX, NQ, PACKED, LISTS, false, HALF>(A, tile, sub, part, s_stage, slot, r_a, r_b, r_c, r_g, nx);
    } else {
        // A persistent wave: virtual indices blockIdx, blockIdx + grid, ... (the grid is a multiple of 8 x the waves per
        // block, so an index keeps its XCD label and a block's waves stay neighbours).  Odd rounds run through the grid
        // backwards WITHIN the XCD's share: the order is heaviest first, so a wave that drew a heavy block of one round
        // draws a light one of the next.  Launched for frames with a heaviest-first order of whole tiles only (nsub a power
        // of two: 1, 4, 16): the index -> block mapping is then shifts and one load, nothing to keep in scalar registers
        // across a block (the general mapping's divisions spilled 66 of them to vector lanes).
        constexpr int kParts = 4 / NQ;
        const int G = (int)gridDim.x;
        const int sh = A.nsx == 1 ? 0 : A.nsx == 2 ? 2 : 4;   // log2(nsub)
        // -> tile | sub << 24 | part << 28, or -1
        auto block_of = [&](int round) __attribute__((always_inline)) -> int {
            const int w = (int)blockIdx.x;
            const int x = (round & 1) ? ((((G >> 3) - 1 - (w >> 3)) << 3) | (w & 7)) : w;
            const int vb = round * G + x;
            if (vb >= A.nvb) return -1;
            const int j = vb >> 3;
            const int part = kParts > 1 ? j % kParts : 0;
            const int wg = kParts > 1 ? (((j / kParts) << 3) | (vb & 7)) : vb;
            if (wg >= A.ngrid) return -1;
            const int e = (((wg >> 3) >> sh) << 3) | (wg & 7);
            if (e >= (A.nblocks >> sh)) return -1;   // (the grid is padded to 8 tiles)
            return A.order[e] | (((wg >> 3) & ((1 << sh) - 1)) << 24) | (part << 28);
        };
        int cur = block_of(0);
        for (int round = 0; round * G < A.nvb; ++round) {
            // the block after this one: its list's bounds are needed before this block's last batch
            const int nxt = block_of(round + 1);
            nx.valid = 0;
            if (nxt >= 0 && raster_block_in_band(A, nxt & 0xffffff, (nxt >> 24) & 15)) {
                int e_all;
                raster_list_bounds(A, nxt & 0xffffff, nx.start, nx.end, e_all);
                nx.valid = nx.start < nx.end ? 1 : 0;
            }
            if (cur >= 0) {
                const int tile = cur & 0xffffff, sub = (cur >> 24) & 15, part = cur >> 28;
                int slot = 0;
                MS_DIAG_ONLY(slot = (round * G + (int)blockIdx.x);)   // (diagnostic stamps: one slot per (wave, round))
                raster_tile<CP, ColorT, AUX, NQ, PACKED, LISTS, true>(A, tile, sub, part, s_stage, slot, r_a, r_b, r_c, r_g, nx);
            } else nx.carry = 0;
            cur = nxt;
        }
    }
}

// ---- clean-up pass of a lazily sorted frame -----------------------------------------------------
// Tiles on the redo list ran out of sorted front with pixels still alive (never seen on the
// benchmark scenes; forced by tests with stacks of faint Gaussians and by 32-px tiles).  They are
// redone from scratch here WITHOUT ever sorting their whole list: one 256-thread workgroup per tile
// repeatedly (1) selects the next ~kRedoChunk (1024) smallest keys above the last one consumed -- range
// of the eligible 64-bit keys, LDS histogram over 2048 order-preserving buckets, prefix, narrowing
// into a crowded first bucket if need be (keys are distinct, so that terminates) -- (2) ranks them
// inside their buckets, and (3) composites them, one wave per 8x8 quad and one pixel per lane, with
// exactly the arithmetic of k_rasterize_fwd's generic loop, until every pixel is finished or the
// list is exhausted.  Any list length, no scratch beyond LDS.  Slow and simple by design.
// Round 3: ... but not quadratic.  Selecting every chunk from the WHOLE list costs n^2 / 1024 key reads per 16x16
// block: a scene swap that left 262 bins of 26-69 thousand entries short of their fronts (config 4) took 14 SECONDS
// a frame.  Lists of more than kRedoPartMin entries are therefore partitioned once per tile: their entries' indices are
// written in the order of 2 048 key buckets into the tile's slots of the id array (nobody reads the ids of a redone
// tile any more: the rasteriser is done, the clean-up stages from the keys), the buckets' starts stay in LDS, and
// every chunk is then selected from the buckets that hold the next ~1 024 keys alone.
constexpr int kRedoChunk = 1024, kRedoCap = 2048, kRedoNB = 2048, kRedoPartMin = 4096;

// Round 4: TWO launches where the lists are long.  One workgroup per bin walking the bin's 16x16 blocks one after the
// other, every chunk selected and ranked afresh for each block, made the frame after a scene swap the heaviest bin's
// serial time: 16 blocks x 4.4 ms at config 4 on 64-px bins, 70-85 ms a frame.  On frames that can expect stranded bins
// (LazyLists::redo_sort: a depth-cut frame, or the record of the previous frame on this scratch reported redone bins)
// k_redo_sort first SORTS every stranded bin's keys whole, one workgroup per bin, and k_tile_redo then runs one
// workgroup per (bin, 16x16 block), walking the sorted ids like the main kernel walks a front.
// The sort is a sample sort in LDS: 4 095 of the bin's keys (evenly spaced positions) are sorted and used as splitters,
// the entries' indices are written bucket by bucket into the bin's slots of the id array (the partition of the one-launch
// path, with balanced buckets whatever the depth distribution), then windows of whole buckets of at most 4 096 keys are
// brought into LDS, every key ranked inside its bucket, and the Gaussian ids written over the indices.  redo_flag[bin] = 2 says "sorted".
// Keys are distinct, so the order is THE order of the fully sorted path.  A bucket of more than 4 096 keys (the sample
// failed: not seen) leaves the bin to the selection path below, block by block.
constexpr int kSortCap = 4096, kSortThreads = 1024, kSortPer = kSortCap / kSortThreads;
__device__ __forceinline__ void redo_bitonic(uint64_t *s, int P, int tid) {   // P: a power of two <= kSortCap; ends with a barrier
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (P >> 1); i += kSortThreads) {
                const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1)), b = a | j;
                const bool up = (a & k) == 0;
                const uint64_t x = s[a], y = s[b];
                if ((x > y) == up) { s[a] = y; s[b] = x; }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(kSortThreads) void k_redo_sort(RasterArgs A) {
    __shared__ __attribute__((aligned(16))) uint64_t s_key[kSortCap];   // the sorted sample (splitters), then a window's keys
    __shared__ uint32_t s_cnt[kSortCap];
    __shared__ uint32_t s_bstart[kSortCap + 1];
    __shared__ uint32_t s_wred[2 * (kSortThreads / 64)];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n_redo = *A.lazy.redo_count;
    for (int ri = blockIdx.x; ri < n_redo; ri += gridDim.x) {
        const int tile = A.lazy.redo_list[ri];
        const int end_all = min(A.tile_ranges[2 * tile + 1], A.max_isects);
        const int start = min(A.tile_ranges[2 * tile], end_all);
        const int n = end_all - start;
        const uint64_t *kin = A.lazy.keys + start;
        const bool far_tile = A.lazy.cut_stamp && A.lazy.has_far[tile] == A.lazy.cut_stamp;
        const int64_t far_base = far_tile ? min((int64_t)A.lazy.cut_words[0] + (int64_t)A.lazy.far_start[tile], (int64_t)A.max_isects) : 0;
        const int n_far = far_tile ? (int)min((int64_t)A.lazy.far_cnt[tile], (int64_t)A.max_isects - far_base) : 0;
        const uint64_t *fkeys = A.lazy.keys + far_base;
        const int m_all = n + n_far;
        int32_t *ids_near = const_cast<int32_t *>(A.flatten_ids) + start, *ids_far = const_cast<int32_t *>(A.flatten_ids) + far_base;
        auto key_of = [&](int e) __attribute__((always_inline)) { return e < n ? kin[e] : fkeys[e - n]; };
        auto put = [&](int pos, int32_t v) __attribute__((always_inline)) { if (pos < n) ids_near[pos] = v; else ids_far[pos - n] = v; };
        auto get = [&](int pos) __attribute__((always_inline)) { return pos < n ? ids_near[pos] : ids_far[pos - n]; };
        auto pow2_of = [](int c) { int P = 2; while (P < c) P <<= 1; return P; };
        __syncthreads();   // (the previous bin is done with the LDS arrays)
        if (m_all <= 0) continue;   // (uniform)
        if (m_all <= kSortCap) {
            const int P = pow2_of(m_all);
            for (int i = tid; i < P; i += kSortThreads) s_key[i] = i < m_all ? key_of(i) : ~0ull;
            __syncthreads();
            redo_bitonic(s_key, P, tid);
            for (int i = tid; i < m_all; i += kSortThreads) put(i, (int32_t)(uint32_t)s_key[i]);
            if (tid == 0) A.lazy.redo_flag[tile] = 2;
            continue;
        }
        // splitters: kSortCap - 1 keys at evenly spaced positions (distinct positions, distinct keys) + one +inf
        for (int i = tid; i < kSortCap; i += kSortThreads)
            s_key[i] = i < kSortCap - 1 ? key_of((int)(((int64_t)i * m_all) / (kSortCap - 1))) : ~0ull;
        for (int b = tid; b < kSortCap; b += kSortThreads) s_cnt[b] = 0;
        __syncthreads();
        redo_bitonic(s_key, kSortCap, tid);
        auto bucket_of = [&](uint64_t k) __attribute__((always_inline)) {   // the splitters below k: 0 .. kSortCap - 1
            int lo = 0;
#pragma unroll
            for (int step = kSortCap >> 1; step > 0; step >>= 1)
                if (s_key[lo + step - 1] < k) lo += step;
            return lo;
        };
        // (four keys of a thread in flight: one load per trip left the pass waiting on memory 117 times for a 120 000-entry bin)
        auto each_key4 = [&](auto &&fn) __attribute__((always_inline)) {
            for (int e0 = tid; e0 < m_all; e0 += 4 * kSortThreads) {
                uint64_t k4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * kSortThreads;
                    k4[u] = e < m_all ? key_of(e) : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * kSortThreads;
                    if (e < m_all) fn(e, k4[u]);
                }
            }
        };
        each_key4([&](int, uint64_t k) { atomicAdd(&s_cnt[bucket_of(k)], 1u); });
        __syncthreads();
        unsigned int biggest = 0;
        {   // exclusive scan of the counters: thread t owns buckets kSortPer t .. kSortPer t + kSortPer - 1
            constexpr int NW = kSortThreads / 64;
            unsigned int c[kSortPer], sum = 0;
#pragma unroll
            for (int j = 0; j < kSortPer; ++j) { c[j] = s_cnt[kSortPer * tid + j]; sum += c[j]; biggest = max(biggest, c[j]); }
            unsigned int incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned int o = (unsigned int)__shfl_up((int)incl, d);
                if (lane >= d) incl += o;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) biggest = max(biggest, (unsigned int)__shfl_xor((int)biggest, d));
            if (lane == 63) s_wred[w] = incl;
            if (lane == 0) s_wred[NW + w] = biggest;
            __syncthreads();
            unsigned int run = incl - sum;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                if (ww < w) run += s_wred[ww];
                biggest = max(biggest, s_wred[NW + ww]);
            }
#pragma unroll
            for (int j = 0; j < kSortPer; ++j) { s_bstart[kSortPer * tid + j] = run; s_cnt[kSortPer * tid + j] = run; run += c[j]; }
            if (tid == kSortThreads - 1) s_bstart[kSortCap] = run;
        }
        __syncthreads();
        if (biggest > (unsigned int)kSortCap) continue;   // (uniform) the selection path takes this bin
        each_key4([&](int e, uint64_t k) { put((int)atomicAdd(&s_cnt[bucket_of(k)], 1u), e); });
        __threadfence_block();
        __syncthreads();   // (the splitters are done with: s_key now holds one window after the other)
        for (int cb = 0; cb < kSortCap;) {   // (uniform)
            // the largest ce >= cb whose buckets cb .. ce hold at most kSortCap keys together (every bucket alone does)
            const unsigned int w0 = s_bstart[cb];
            int lo_b = cb, hi_b = kSortCap - 1;
            while (lo_b < hi_b) {
                const int mid = (lo_b + hi_b + 1) >> 1;
                if (s_bstart[mid + 1] - w0 <= (unsigned int)kSortCap) lo_b = mid; else hi_b = mid - 1;
            }
            const int cnt = (int)(s_bstart[lo_b + 1] - w0);
            if (cnt > 0) {
                // every key ranked inside its own bucket (~m / 4 096 keys each): a bitonic network over 2 048-key windows cost 65 us
                // a window, 2.4 of the launch's 3.0 ms at config 4
                uint64_t kk[kSortPer];
#pragma unroll
                for (int e = 0; e < kSortPer; ++e) {
                    const int i = e * kSortThreads + tid;
                    kk[e] = i < cnt ? key_of(get((int)w0 + i)) : ~0ull;
                    if (i < cnt) s_key[i] = kk[e];
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < kSortPer; ++e) {
                    const int i = e * kSortThreads + tid;
                    if (i < cnt) {
                        int lo = cb, hi = lo_b;   // the bucket of position w0 + i: the last b with s_bstart[b] <= w0 + i
                        while (lo < hi) {
                            const int mid = (lo + hi + 1) >> 1;
                            if (s_bstart[mid] <= w0 + (unsigned int)i) lo = mid; else hi = mid - 1;
                        }
                        const int beg = (int)(s_bstart[lo] - w0), en = (int)(s_bstart[lo + 1] - w0);
                        int r = 0, j = beg;   // (two keys a read: the pass is bound by its LDS instructions)
                        if ((j & 1) && j < en) { r += s_key[j] < kk[e] ? 1 : 0; ++j; }
                        for (; j + 1 < en; j += 2) {
                            const ulonglong2 two = *reinterpret_cast<const ulonglong2 *>(&s_key[j]);
                            r += (two.x < kk[e] ? 1 : 0) + (two.y < kk[e] ? 1 : 0);
                        }
                        if (j < en) r += s_key[j] < kk[e] ? 1 : 0;
                        put((int)w0 + beg + r, (int32_t)(uint32_t)kk[e]);   // (every get() of this window came before the barrier)
                    }
                }
                __syncthreads();
            }
            cb = lo_b + 1;
        }
        if (tid == 0) A.lazy.redo_flag[tile] = 2;
    }
}

template <int CP, typename ColorT>
__global__ __launch_bounds__(256) void k_tile_redo(RasterArgs A) {
    __shared__ uint64_t s_key[kRedoCap];
    __shared__ uint32_t s_cnt[kRedoNB];
    __shared__ uint32_t s_bstart[kRedoNB + 1];   // a partitioned list: where each key bucket's indices begin
    __shared__ unsigned long long s_red[16];
    __shared__ int s_sel[4];                 // b*, F, first bucket, first bucket's count
    __shared__ float4 s_pa[256], s_pb[256];  // staged entries: mean.x mean.y a' b' | c' log2(o) - -
    __shared__ float s_pc[256 * CP];
    __shared__ int s_alive[4];
    __shared__ unsigned long long s_hit[16];   // [quad][staging wave]: the staged entries that reach the quad
    flush_fp32_denormals();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lx = lane & 7, ly = lane >> 3;
    const ColorT *colors = reinterpret_cast<const ColorT *>(A.colors);
    const int n_redo = *A.lazy.redo_count;
    constexpr float kInf = __builtin_huge_valf();

    // (two launches -- see k_redo_sort: an item is one 16x16 block of a stranded bin; else a bin with all its blocks)
    const bool two = A.lazy.redo_sort != 0;
    const int n_items = two ? n_redo * A.nsub : n_redo;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ri = two ? item / A.nsub : item;
        const int sub_lo = two ? item - ri * A.nsub : 0, sub_hi = two ? sub_lo + 1 : A.nsub;
        const int tile = A.lazy.redo_list[ri];
        const int tile_y = tile / A.tw, tile_x = tile - tile_y * A.tw;
        const int end_all = min(A.tile_ranges[2 * tile + 1], A.max_isects);
        const int start = min(A.tile_ranges[2 * tile], end_all);
        const int n = end_all - start;
        const uint64_t *kin = A.lazy.keys + start;
        // depth-cut frame: the tile's dropped pairs are back in a segment behind the lists (k_far_regen, the launches
        // before this one); every pass over "the tile's keys" below takes both
        const bool far_tile = A.lazy.cut_stamp && A.lazy.has_far[tile] == A.lazy.cut_stamp;
        const int64_t far_base = far_tile ? min((int64_t)A.lazy.cut_words[0] + (int64_t)A.lazy.far_start[tile], (int64_t)A.max_isects) : 0;
        const int n_far = far_tile ? (int)min((int64_t)A.lazy.far_cnt[tile], (int64_t)A.max_isects - far_base) : 0;
        const uint64_t *fkeys = A.lazy.keys + far_base;
        // A long list is partitioned once (see above): perm = the entries' indices (0 .. n: the list; n ..: the
        // regenerated segment) in key-bucket order, in the id slots of the same two ranges
        const int m_all = n + n_far;
        // (two launches: k_redo_sort left the bin's ids fully sorted in those slots -- or, had its sample failed, untouched:
        // then every block selects from the whole list, the slots being shared with the bin's other workgroups)
        const bool sorted = two && A.lazy.redo_flag[tile] == 2;
        const bool part = !two && !A.lazy.packed && m_all > kRedoPartMin;
        int32_t *perm_near = const_cast<int32_t *>(A.flatten_ids) + start;
        int32_t *perm_far = const_cast<int32_t *>(A.flatten_ids) + far_base;
        auto key_of = [&](int e) __attribute__((always_inline)) { return e < n ? kin[e] : fkeys[e - n]; };
        unsigned long long pk_min = 0ull;
        int pshift = 0;
        if (part) {
            unsigned long long mn = ~0ull, mx = 0ull;
            for (int e = tid; e < m_all; e += 256) {
                const unsigned long long k = key_of(e);
                mn = k < mn ? k : mn; mx = k > mx ? k : mx;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                const unsigned long long omn = __shfl_xor(mn, d), omx = __shfl_xor(mx, d);
                mn = omn < mn ? omn : mn;
                mx = omx > mx ? omx : mx;
            }
            __syncthreads();   // (the previous tile is done with the LDS arrays)
            if (lane == 0) { s_red[w] = mn; s_red[4 + w] = mx; }
            for (int b = tid; b < kRedoNB; b += 256) s_cnt[b] = 0;
            __syncthreads();
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) { mn = s_red[ww] < mn ? s_red[ww] : mn; mx = s_red[4 + ww] > mx ? s_red[4 + ww] : mx; }
            pk_min = mn;
            const unsigned long long span = mx - mn;
            pshift = max(0, (span ? 64 - __clzll((long long)span) : 0) - 11);
            for (int e = tid; e < m_all; e += 256) atomicAdd(&s_cnt[(unsigned int)((key_of(e) - pk_min) >> pshift)], 1u);
            __syncthreads();
            {   // exclusive scan of the counters: thread t owns buckets 8 t .. 8 t + 7
                unsigned int c[8], sum = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) { c[j] = s_cnt[8 * tid + j]; sum += c[j]; }
                unsigned int incl = sum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned int o = (unsigned int)__shfl_up((int)incl, d);
                    if (lane >= d) incl += o;
                }
                __syncthreads();
                if (lane == 63) s_red[12 + w] = incl;
                __syncthreads();
                unsigned int run = incl - sum;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww)
                    if (ww < w) run += (unsigned int)s_red[12 + ww];
#pragma unroll
                for (int j = 0; j < 8; ++j) { s_bstart[8 * tid + j] = run; s_cnt[8 * tid + j] = run; run += c[j]; }
                if (tid == 255) s_bstart[kRedoNB] = run;
            }
            __syncthreads();
            for (int e = tid; e < m_all; e += 256) {
                const int pos = (int)atomicAdd(&s_cnt[(unsigned int)((key_of(e) - pk_min) >> pshift)], 1u);
                if (pos < n) perm_near[pos] = e; else perm_far[pos - n] = e;
            }
            __threadfence_block();
            __syncthreads();
        }
        int w0 = 0, w1 = m_all;   // (a partitioned list: the slice of perm the current chunk is selected from)
        auto each_key = [&](auto &&fn) __attribute__((always_inline)) {
            if (part) {
                for (int i = w0 + tid; i < w1; i += 256) fn(key_of(i < n ? perm_near[i] : perm_far[i - n]));
            } else {
                for (int i = tid; i < n; i += 256) fn(kin[i]);
                for (int i = tid; i < n_far; i += 256) fn(fkeys[i]);
            }
        };
        // the next frame keeps every pair of a tile that outlived its front (k_tile_front, which ran before this
        // launch, has just set the cut-off where that front ended)
        if (A.lazy.tau_next && tid == 0) A.lazy.tau_next[tile] = 0xffffffffu;
        for (int sub = sub_lo; sub < sub_hi; ++sub) {
            const int sub_y = sub / A.nsx, sub_x = sub - sub_y * A.nsx;
            if (A.lazy.packed) {   // a bin at the edge of a band: only the block rows inside it (uniform)
                const int brow = tile_y * A.nsx + sub_y;
                if (brow < A.lazy.row_lo || brow >= A.lazy.row_hi) continue;
            } else {               // likewise a coarse tile of a band cut at 16-px rows
                const int brow = tile_y * A.nsx + sub_y;
                if (brow < A.row0 || brow >= A.row1) continue;
            }
            const int q = w;  // one quad per wave
            const int ox = sub_x * 16 + lx + (q & 1) * 8, oy = sub_y * 16 + ly + (q >> 1) * 8;
            const int X = tile_x * A.ts + ox, Y = tile_y * A.ts + oy;
            const bool in = ox < A.ts && oy < A.ts && X < A.W && Y < A.H;
            const float px = (float)X + 0.5f, py = (float)Y + 0.5f;
            float T = ms::kTScale, kq = in ? ms::kFlushK : 0.f, pix[CP];   // k_rasterize_fwd's flush select and scaled state
#pragma unroll
            for (int k = 0; k < CP; ++k) pix[k] = 0.f;
            // one staged entry (this thread's slot of the 256) -- `pre`: its record, already in registers -- and the blend of
            // the m staged entries into this wave's quad
            auto stage_entry = [&](unsigned int word, const float4 *pre) __attribute__((always_inline)) -> int {
                const int g = min(max((int)(A.lazy.packed ? word >> 4 : word), 0), A.n_gauss - 1);
                // opacity below 1/255 can never blend: log2 -> -inf keeps alpha at 0
                // (split frames: nor can an entry that is not on this block's list)
                const bool listed = !A.lazy.packed || ((word >> sub) & 1u);
                if constexpr (CP == 3) {
                    if (A.records) {
                        // the frame's ready-made records: the same staged values, and the only per-Gaussian
                        // data a pre-culled band frame can be asked for by list id (ids are positions in
                        // the band's candidate list there, not indices into the caller's arrays)
                        const float4 *rec = A.records + 3 * (size_t)g;
                        const float4 ra = pre ? pre[0] : rec[0], rb = pre ? pre[1] : rec[1], rc = pre ? pre[2] : rec[2];
                        // which of the block's four 8x8 quads the entry can reach at all: the forward kernel's
                        // exact ellipse-vs-quad test on the record's three numbers (log2 units) -- a wave skips
                        // what cannot blend in its quad (a 64-px bin's list is ~94 % misses for any one quad)
                        int hit = 0;
                        const float smax = rc.y, nb_c = rc.z, nb_a = rc.w;
                        if (smax == kInf) {
                            hit = 0xf;
                        } else if (smax > -kInf) {
                            const float fbx = (float)(tile_x * A.ts + sub_x * 16) + 0.5f, fby = (float)(tile_y * A.ts + sub_y * 16) + 0.5f;
#pragma unroll
                            for (int qq = 0; qq < 4; ++qq) {
                                const float xl = fbx + (float)((qq & 1) * 8) - ra.x, xh = xl + 7.0f;
                                const float yl = fby + (float)((qq >> 1) * 8) - ra.y, yh = yl + 7.0f;
                                const bool in_x = xl <= 0.f && xh >= 0.f, in_y = yl <= 0.f && yh >= 0.f;
                                float best = (in_x && in_y) ? 0.f : 3.0e38f;
                                if (!in_x) {
                                    const float ddx = xl > 0.f ? xl : xh;
                                    const float ddy = fminf(fmaxf(nb_c * ddx, yl), yh);
                                    best = -(ra.z * ddx * ddx + rb.x * ddy * ddy + ra.w * ddx * ddy);
                                }
                                if (!in_y) {
                                    const float ddy = yl > 0.f ? yl : yh;
                                    const float ddx = fminf(fmaxf(nb_a * ddy, xl), xh);
                                    best = fminf(best, -(ra.z * ddx * ddx + rb.x * ddy * ddy + ra.w * ddx * ddy));
                                }
                                hit |= (best <= smax) ? (1 << qq) : 0;
                            }
                        }
                        s_pa[tid] = ra;
                        s_pb[tid] = make_float4(rb.x, rc.y > -kInf && listed ? rb.y : -kInf, __int_as_float(hit), 0.f);
                        s_pc[tid * CP] = rb.z; s_pc[tid * CP + 1] = rb.w; s_pc[tid * CP + 2] = rc.x;
                        return hit;
                    }
                }
                {
                    const float2 m = reinterpret_cast<const float2 *>(A.means2d)[g];
                    const float ca = A.conics[3 * g], cb = A.conics[3 * g + 1], cc = A.conics[3 * g + 2];
                    const float op = A.opacities[g];
                    s_pa[tid] = make_float4(m.x, m.y, -0.5f * kLog2e * ca, -kLog2e * cb);
                    s_pb[tid] = make_float4(-0.5f * kLog2e * cc, op >= ms::kAlphaThreshold && listed ? __log2f(op) : -kInf, __int_as_float(0xf), 0.f);
#pragma unroll
                    for (int k = 0; k < CP; ++k)
                        s_pc[tid * CP + k] = k < A.cdim ? load_color(colors + (size_t)g * A.cdim + k) : 0.f;
                }
                return 0xf;
            };
            // (which staged entries reach which quad: 64-bit masks, one per quad and staging wave -- a wave walks the set bits of
            // its quad's four instead of testing 256 entries of which a 64-px bin's list lets ~6 % through)
            auto publish_hits = [&](int hit) __attribute__((always_inline)) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const unsigned long long hm = __ballot((hit >> qq) & 1);
                    if (lane == 0) s_hit[qq * 4 + w] = hm;
                }
            };
            auto blend = [&]() __attribute__((always_inline)) {
                if (!__any(kq != 0.f)) return;   // (a wave whose pixels are all finished has nothing to blend)
#pragma unroll 1
                for (int ww = 0; ww < 4; ++ww) {
                  const unsigned long long hm_v = s_hit[w * 4 + ww];
                  unsigned long long hm = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(hm_v >> 32)) << 32) |
                                          (unsigned int)__builtin_amdgcn_readfirstlane((int)hm_v);
                  while (hm) {
                    const int t = ww * 64 + __builtin_ctzll(hm);
                    hm &= hm - 1ull;
                    const float4 rb = s_pb[t];
                    const float4 ra = s_pa[t];
                    const float dx = ra.x - px, dy = ra.y - py;
                    const float la = fmaf(dx, fmaf(ra.z, dx, ra.w * dy), fmaf(rb.x * dy, dy, rb.y));
                    // k_rasterize_fwd's generic (CHECK) arithmetic, operation for operation
                    float alpha = fminf(ms::kMaxAlpha, __builtin_amdgcn_exp2f(la));
                    alpha = la <= rb.y ? alpha : 0.f;
                    const float mj = alpha * kq;
                    const float vj = mj * T;
                    const float nt = fmaf(vj, -ms::kAlphaOfV, T);
                    const bool dead = !(nt > ms::kTransmittanceStop * ms::kTScale);
                    const float vis = dead ? 0.f : vj;
                    kq = dead ? 0.f : kq;
#pragma unroll
                    for (int k = 0; k < CP; ++k) pix[k] += s_pc[t * CP + k] * vis;
                    T = dead ? T : nt;
                  }
                }
            };
            if (sorted) {
                // the bin's ids in depth order (k_redo_sort): staged 256 at a time like a chunk below, the next 256 records on
                // their way while these are blended; done when the block's four quads are, or the list is
                auto word_at = [&](int i) __attribute__((always_inline)) { return (unsigned int)(i < n ? perm_near[i] : perm_far[i - n]); };
                bool ahead = false;
                if constexpr (CP == 3) ahead = A.records != nullptr;
                float4 nx[3] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
                auto fetch = [&](int i) __attribute__((always_inline)) {
                    if (ahead && i < m_all) {
                        const int g = min(max((int)word_at(i), 0), A.n_gauss - 1);
                        const float4 *rec = A.records + 3 * (size_t)g;
                        nx[0] = rec[0]; nx[1] = rec[1]; nx[2] = rec[2];
                    }
                };
                fetch(tid);
                for (int e0 = 0; e0 < m_all; e0 += 256) {
                    const bool wa = __any(kq != 0.f);
                    if (lane == 0) s_alive[w] = wa ? 1 : 0;
                    __syncthreads();   // (the previous 256 are blended; the waves' flags are in)
                    if (!(s_alive[0] | s_alive[1] | s_alive[2] | s_alive[3])) break;   // (uniform)
                    publish_hits(e0 + tid < m_all ? stage_entry(word_at(e0 + tid), ahead ? nx : nullptr) : 0);
                    __syncthreads();
                    fetch(e0 + 256 + tid);
                    blend();
                }
            }
            unsigned long long lower = 0ull;   // keys consumed so far are <= lower (exclusive bound once !first)
            bool first = true;
            for (; !sorted;) {
                // ---- (1) next chunk: window [wlo, whi] over eligible keys
                if (!first && lower == ~0ull) break;   // nothing can follow the largest key (and lower + 1 would wrap)
                unsigned long long wlo = first ? 0ull : lower + 1ull, whi = ~0ull;
                if (part) {
                    // the buckets the next chunk can come from: the one the last consumed key lies in (its remaining
                    // keys pass the `wlo` filter) and as many after it as hold kRedoChunk keys (or all that are left)
                    const int cb = first ? 0 : (int)min((unsigned long long)(kRedoNB - 1), (lower - pk_min) >> pshift);
                    const unsigned int want_to = s_bstart[cb + 1] + (unsigned int)kRedoChunk;
                    int lo_b = cb, hi_b = kRedoNB - 1;   // smallest ce >= cb with s_bstart[ce + 1] >= want_to (else the last bucket)
                    while (lo_b < hi_b) {
                        const int mid = (lo_b + hi_b) >> 1;
                        if (s_bstart[mid + 1] >= want_to) hi_b = mid; else lo_b = mid + 1;
                    }
                    w0 = (int)s_bstart[cb];
                    w1 = (int)s_bstart[lo_b + 1];
                }
                int bstar = -1, F = 0;
                unsigned long long kmin = 0ull;
                int shift = 0;
                bool none = false;
                for (;;) {
                    unsigned long long mn = ~0ull, mx = 0ull;
                    unsigned int cnt = 0;
                    each_key([&](unsigned long long k) {
                        if (k >= wlo && k <= whi) { mn = k < mn ? k : mn; mx = k > mx ? k : mx; ++cnt; }
                    });
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) {
                        const unsigned long long omn = __shfl_xor(mn, d), omx = __shfl_xor(mx, d);
                        mn = omn < mn ? omn : mn;
                        mx = omx > mx ? omx : mx;
                        cnt += (unsigned int)__shfl_xor((int)cnt, d);
                    }
                    __syncthreads();   // previous users of s_red / s_cnt / s_sel are done
                    if (lane == 0) { s_red[w] = mn; s_red[4 + w] = mx; s_red[8 + w] = cnt; }
                    for (int b = tid; b < kRedoNB; b += 256) s_cnt[b] = 0;
                    if (tid == 0) { s_sel[0] = -1; s_sel[1] = 0; s_sel[2] = -1; s_sel[3] = 0; }
                    __syncthreads();
                    unsigned long long total = 0;
                    mn = ~0ull; mx = 0ull;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        mn = s_red[ww] < mn ? s_red[ww] : mn;
                        mx = s_red[4 + ww] > mx ? s_red[4 + ww] : mx;
                        total += s_red[8 + ww];
                    }
                    if (total == 0) { none = true; break; }   // uniform
                    kmin = mn;
                    const unsigned long long span = mx - mn;
                    const int bits = span ? 64 - __clzll((long long)span) : 0;
                    shift = max(0, bits - 11);
                    each_key([&](unsigned long long k) {
                        if (k >= wlo && k <= whi) atomicAdd(&s_cnt[(unsigned int)((k - kmin) >> shift)], 1u);
                    });
                    __syncthreads();
                    // exclusive scan of 2048 counters: thread t owns buckets 8t .. 8t+7
                    unsigned int c[8], sum = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { c[j] = s_cnt[8 * tid + j]; sum += c[j]; }
                    unsigned int incl = sum;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const unsigned int o = (unsigned int)__shfl_up((int)incl, d);
                        if (lane >= d) incl += o;
                    }
                    __syncthreads();
                    if (lane == 63) s_red[12 + w] = incl;
                    __syncthreads();
                    unsigned int run = incl - sum;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww)
                        if (ww < w) run += (unsigned int)s_red[12 + ww];
                    const unsigned int want = total < (unsigned long long)kRedoChunk ? (unsigned int)total : (unsigned int)kRedoChunk;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned int e = run, i2 = run + c[j];
                        if (c[j] && e == 0) { s_sel[2] = 8 * tid + j; s_sel[3] = (int)c[j]; }   // first non-empty bucket
                        if (c[j] && e < want && i2 >= want) {
                            if (i2 <= (unsigned int)kRedoCap) { s_sel[0] = 8 * tid + j; s_sel[1] = (int)i2; }
                            else { s_sel[0] = 8 * tid + j - 1; s_sel[1] = (int)e; }
                        }
                        s_cnt[8 * tid + j] = e;
                        run = i2;
                    }
                    __syncthreads();
                    bstar = s_sel[0]; F = s_sel[1];
                    if (F > 0) break;
                    // the first bucket alone overflows the chunk: narrow the window into it and retry
                    const unsigned long long blo = kmin + ((unsigned long long)s_sel[2] << shift);
                    wlo = blo > wlo ? blo : wlo;
                    const unsigned long long bhi = shift ? blo + ((1ull << shift) - 1ull) : blo;
                    whi = bhi < whi ? bhi : whi;
                }
                if (none) break;
                // gather + rank
                each_key([&](unsigned long long k) {
                    if (k >= wlo && k <= whi) {
                        const int b = (int)((k - kmin) >> shift);
                        if (b <= bstar) s_key[atomicAdd(&s_cnt[b], 1u)] = k;
                    }
                });
                __syncthreads();
                unsigned long long kk[kRedoCap / 256];
                int dest[kRedoCap / 256];
#pragma unroll
                for (int e = 0; e < kRedoCap / 256; ++e) {
                    const int i = e * 256 + tid;
                    dest[e] = -1;
                    if (i < F) {
                        kk[e] = s_key[i];
                        const int b = (int)((kk[e] - kmin) >> shift);
                        const int beg = b ? (int)s_cnt[b - 1] : 0, en = (int)s_cnt[b];
                        int r = 0;
                        for (int j = beg; j < en; ++j) r += s_key[j] < kk[e] ? 1 : 0;
                        dest[e] = beg + r;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < kRedoCap / 256; ++e)
                    if (dest[e] >= 0) s_key[dest[e]] = kk[e];
                __syncthreads();
                lower = s_key[F - 1];
                first = false;
                // ---- (2) composite the chunk, 256 entries staged at a time
                for (int e0 = 0; e0 < F; e0 += 256) {
                    __syncthreads();
                    publish_hits(e0 + tid < F ? stage_entry((unsigned int)s_key[e0 + tid], nullptr) : 0);
                    __syncthreads();
                    blend();
                }
                // ---- (3) anyone still alive?
                const bool wa = __any(kq != 0.f);
                __syncthreads();
                if (lane == 0) s_alive[w] = wa ? 1 : 0;
                __syncthreads();
                if (!(s_alive[0] | s_alive[1] | s_alive[2] | s_alive[3])) break;
            }
            if (in) {
                const size_t p = (size_t)Y * A.W + X;
#pragma unroll
                for (int k = 0; k < CP; ++k)
                    if (k < A.cdim) A.render_colors[p * A.cdim + k] = finish_channel(pix[k], T, A.backgrounds ? A.backgrounds[k] : 0.f);
                if (A.render_alphas) A.render_alphas[p] = 1.0f - T * ms::kTUnscale;   // (a differentiable frame keeps them)
            }
            __syncthreads();
        }
    }
}

// MOJOSPLAT_RASTER_PARTS=1|2|4 forces the waves per 16x16 block (measurements); 0 / unset: the rule below
static int raster_parts_override() {
    static const int v = [] {
        const char *e = getenv("MOJOSPLAT_RASTER_PARTS");
        const int n = e ? atoi(e) : 0;
        return (n == 1 || n == 2 || n == 4) ? n : 0;
    }();
    return v;
}

// MOJOSPLAT_RASTER_HALF=1: one-quad waves keep a compacted stream per 8x4 half (A/B: profiles/r06_raster_halfquad.md)
static bool raster_half() {
    static const bool v = [] {
        const char *e = getenv("MOJOSPLAT_RASTER_HALF");
        return e && e[0] == '1';
    }();
    return v;
}

// MOJOSPLAT_RASTER_CHAIN=k: blocks a rasteriser wave walks one after the other (1: one block per wave, as rounds 2-5;
// A/B: profiles/r06_raster_startup.md)
static int raster_chain() {
    static const int v = [] {
        const char *e = getenv("MOJOSPLAT_RASTER_CHAIN");
        const int k = e ? atoi(e) : 1;
        return k >= 1 && k <= 16 ? k : 1;
    }();
    return v;
}

static unsigned redo_grid(const RasterArgs &A) { return (unsigned)(A.lazy.redo_grid >= 1 && A.lazy.redo_grid <= 4096 ? A.lazy.redo_grid : 64); }

// The clean-up launches behind a lazily sorted frame's rasteriser: empty on almost every frame.
template <int CP, typename ColorT>
void cleanup_launches(const RasterArgs &A, hipStream_t stream) {
    if constexpr (CP <= 4) {
        // (depth-cut frame: the pairs -- and records -- the redone bins are short of, first: binning.hip, k_far_regen)
        if (A.lazy.cut_stamp) (void)ms::far_regen(A.lazy, A.tw, A.tw * ((A.H + A.ts - 1) / A.ts), (int64_t)A.max_isects, stream);
        // (frames that can expect stranded bins: their keys sorted whole first, then a workgroup per 16x16 block)
        if (A.lazy.redo_sort) hipLaunchKernelGGL(k_redo_sort, dim3(512), dim3(kSortThreads), 0, stream, A);
        hipLaunchKernelGGL((k_tile_redo<CP, ColorT>), dim3(A.lazy.redo_sort ? 1024u : redo_grid(A)), dim3(256), 0, stream, A);
    }
}

// Deferred clean-up: what a frame whose caller will look at the rasteriser's verdict itself (LazyLists::verdict) would have
// launched, kept per verdict word -- a lane's pinned record, one frame at a time -- until the finishing half asks for it.
struct CleanupMemo {
    const void *key;
    RasterArgs A;
    ms::CutInputs cut;
    void (*launch)(const RasterArgs &, hipStream_t);
};
std::mutex g_cleanup_mu;
std::vector<CleanupMemo> g_cleanup;

void stash_cleanup(const RasterArgs &A, void (*launch)(const RasterArgs &, hipStream_t)) {
    CleanupMemo m{A.lazy.verdict, A, A.lazy.cut_inputs ? *A.lazy.cut_inputs : ms::CutInputs{}, launch};
    std::lock_guard<std::mutex> lock(g_cleanup_mu);
    for (auto &e : g_cleanup)
        if (e.key == m.key) { e = m; return; }
    g_cleanup.push_back(m);
}

template <int CP, typename ColorT>
void launch_cp(const RasterArgs &A_in, hipStream_t stream, void *after_raster_event) {
    RasterArgs A = A_in;
    A.zero_mem = nullptr; A.zero_per_wave = 0; A.zero_total = 0;
    const bool aux = A.last_ids != nullptr;   // (the per-entry index bookkeeping; render_alphas alone costs the plain kernel one store)
    // parts single-wave workgroups per block, the block count rounded up to the 8 XCDs
    const dim3 grid(A.parts > 1 ? (unsigned)(((A.ngrid + 7) / 8) * 8 * A.parts) : (unsigned)A.ngrid);
    if (A_in.zero_mem) {
        A.zero_mem = A_in.zero_mem; A.zero_total = A_in.zero_total;
        A.zero_per_wave = (unsigned)((A_in.zero_total + grid.x - 1) / grid.x);
    }
#define MS_LAUNCH_RASTER(AUXV, NQV, PK) \
    hipLaunchKernelGGL((k_rasterize_fwd<CP, ColorT, AUXV, NQV, PK>), grid, dim3(64), 0, stream, A)
#define MS_LAUNCH_RASTER_NQ(AUXV, PK)                          \
    do {                                                       \
        if (A.parts == 2) MS_LAUNCH_RASTER(AUXV, 2, PK);       \
        else if (A.parts == 4) MS_LAUNCH_RASTER(AUXV, 1, PK);  \
        else MS_LAUNCH_RASTER(AUXV, 4, PK);                    \
    } while (0)
    bool done = false;
#if MS_RASTER_CHAINS   // (measurement builds, profiles/r06_raster_startup.md: -DMS_RASTER_CHAINS=1; never the shipped library)
    if constexpr (CP == 3) {
        if (A.records && !aux && !A.quad_lists && A.order && !A.order_bins && (A.nsx == 1 || A.nsx == 2 || A.nsx == 4) && (raster_chain() > 1 || getenv("MOJOSPLAT_RASTER_CHAIN_KERNEL"))) {   // (the second: the chain kernel walking one block, to tell its code from its schedule)
            // persistent waves (round 6): as many as the chip holds at once, each walking blocks index, index + grid, ...
            RasterArgs P = A;
            P.nvb = (int)grid.x;
            // CHAINS of blocks: a wave walks `chain` blocks -- index, index + grid, ... with grid = indices / chain -- and the
            // waves are still dispatched one by one as slots come free.  (All waves resident at once and each walking a
            // fixed share of ALL blocks was measured first: the static shares end 25 % apart -- 125 us against 96 at
            // config 3, resident waves falling from 60 % of the kernel on: profiles/r06_raster_startup.md.)
            const int chain = raster_chain();
            auto launch_p = [&](auto kernel, int parts) {
                const int unit = 8 * parts;
                int g = ((P.nvb + chain - 1) / chain + unit - 1) / unit * unit;
                if (g <= 0) g = unit;
                hipLaunchKernelGGL(kernel, dim3((unsigned)g), dim3(64), 0, stream, P);
            };
            if (A.parts == 2) launch_p(k_rasterize_fwd<CP, ColorT, false, 2, true, false, true>, 2);
            else if (A.parts == 4) launch_p(k_rasterize_fwd<CP, ColorT, false, 1, true, false, true>, 4);
            else launch_p(k_rasterize_fwd<CP, ColorT, false, 4, true, false, true>, 1);
            done = true;
        }
    }
#endif
    if constexpr (CP == 3) {
        if (done) {
        } else
#if MS_RASTER_HALF_STREAMS   // (measurement builds, profiles/r06_raster_halfquad.md: -DMS_RASTER_HALF_STREAMS=1 + MOJOSPLAT_RASTER_HALF=1)
        if (A.records && !aux && !A.quad_lists && A.parts == 4 && raster_half()) {
            // one quad a wave, its two halves walking streams of their own
            hipLaunchKernelGGL((k_rasterize_fwd<CP, ColorT, false, 1, true, false, false, true>), grid, dim3(64), 0, stream, A);
            done = true;
        } else
#endif
        if (A.records) {
            if (aux) MS_LAUNCH_RASTER_NQ(true, true);
            else if (A.quad_lists) {   // a differentiable frame that leaves its quads' lists for the backward
                if (A.parts == 2) hipLaunchKernelGGL((k_rasterize_fwd<CP, ColorT, false, 2, true, true>), grid, dim3(64), 0, stream, A);
                else if (A.parts == 4) hipLaunchKernelGGL((k_rasterize_fwd<CP, ColorT, false, 1, true, true>), grid, dim3(64), 0, stream, A);
                else hipLaunchKernelGGL((k_rasterize_fwd<CP, ColorT, false, 4, true, true>), grid, dim3(64), 0, stream, A);
            }
            else MS_LAUNCH_RASTER_NQ(false, true);
            done = true;
        }
    }
    if (!done) {
        if (aux) MS_LAUNCH_RASTER_NQ(true, false);
        else MS_LAUNCH_RASTER_NQ(false, false);
    }
#undef MS_LAUNCH_RASTER_NQ
#undef MS_LAUNCH_RASTER
    // in-situ timing of THIS kernel: the caller's event goes between it and the clean-up launch
    if (after_raster_event) (void)hipEventRecord((hipEvent_t)after_raster_event, stream);
    if constexpr (CP <= 4) {
        // lazily sorted frame: redo the (normally zero) tiles whose front did not saturate them
        // (measured by leaving it out: the launch costs the frame 2.4 us -- 0.1826 -> 0.1802 ms at config 3 -- although
        // rocprofv3 shows the empty kernel at 4.5 us)
        if (A.lazy.front_count) {
            if (A.lazy.verdict) stash_cleanup(A, &cleanup_launches<CP, ColorT>);   // (deferred: rasterize_deferred_cleanup)
            else cleanup_launches<CP, ColorT>(A, stream);
        }
    }
}

template <typename ColorT>
int launch_fwd(const RasterArgs &A, hipStream_t stream, void *after_raster_event) {
    if (A.cdim == 3) launch_cp<3, ColorT>(A, stream, after_raster_event);
    else if (A.cdim <= 4) launch_cp<4, ColorT>(A, stream, after_raster_event);
    else if (A.cdim <= 8) launch_cp<8, ColorT>(A, stream, after_raster_event);
    else if (A.cdim <= 16) launch_cp<16, ColorT>(A, stream, after_raster_event);
    else launch_cp<32, ColorT>(A, stream, after_raster_event);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

template <typename ColorT>
void launch_redo(const RasterArgs &A, hipStream_t stream) {
    if (A.cdim == 3) hipLaunchKernelGGL((k_tile_redo<3, ColorT>), dim3(redo_grid(A)), dim3(256), 0, stream, A);
    else hipLaunchKernelGGL((k_tile_redo<4, ColorT>), dim3(redo_grid(A)), dim3(256), 0, stream, A);
}

}  // namespace

int ms::rasterize_deferred_cleanup(const void *key, void *stream) {
    CleanupMemo m;
    {
        std::lock_guard<std::mutex> lock(g_cleanup_mu);
        auto it = std::find_if(g_cleanup.begin(), g_cleanup.end(), [&](const CleanupMemo &e) { return e.key == key; });
        MS_REQUIRE(it != g_cleanup.end(), MS_ERR_INVALID_ARG, "rasterize_deferred_cleanup: no frame left its clean-up for this record");
        m = *it;
    }
    if (m.A.lazy.cut_inputs) m.A.lazy.cut_inputs = &m.cut;   // (the frame's own copy: the original lived on the enqueuing call's stack)
    m.A.lazy.verdict = nullptr;
    m.launch(m.A, (hipStream_t)stream);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

namespace {
// Waves per 16x16 block.  One wave per block leaves a one-round launch (<= 8192 wave slots on
// the chip) with a long tail: light tiles retire, their slots stay empty and the heavy tiles
// finish at a fraction of the SIMD's issue rate.  Splitting blocks over 2 or 4 waves (each
// stages the whole list but blends only its quads) refills the slots; the duplicated staging
// only pays while staging is cheap against blending: with ready-made records (three 16-byte gathers
// per entry, no arithmetic) every quad gets its own wave; staging from the per-stage arrays (seven
// gathers + the folding arithmetic) 4 waves up to 150 entries per block, 2 up to 1 500, else 1.
int choose_parts(int64_t blocks, int64_t density_hint, bool records, bool plain32 = false) {
    if (const int forced = raster_parts_override()) return forced;
    // (plain forward frames on 32- / 64-px bins whose size record counts 150-600 pairs per 16x16 block -- config 3: 241,
    // config 5: 303 -- run 2 % faster with two waves a block, each staging the bin's list once for two quads: 0.1656
    // against 0.1686 ms and 0.4570 against 0.4670 ms in three alternating pairs of runs each; lighter frames (config 2:
    // 24 a block) and denser ones (config 4: 1 200) keep a wave per quad: 0.0750 against 0.0787, 0.2459 against 0.2514 --
    // and so do 64-px bins on images of fewer than 16 384 blocks: 1 M Gaussians at l = -3 on 1080p, 302 a block, 0.1639
    // with four waves against 0.1669 with two; and a multi-GPU rank's band -- an eighth of config 3: 1 020 blocks -- wants
    // every wave it can get: 28 us against 48.  Hence the block counts: a whole 1080p frame on 32-px bins, 4K on 64)
    if (records && plain32 && blocks > 0 && density_hint / blocks >= 150 && density_hint / blocks <= 600) return 2;
    if (records) return 4;
    if (blocks >= 16384) return 1;   // many-round launches gain nothing
    const int64_t per_block = density_hint / blocks;
    return per_block > 1500 ? 1 : per_block > 150 ? 2 : 4;
}
}  // namespace

// density_hint: intersections the band is expected to hold (the exact M when the caller knows it, the
// previous frame's M on a sync-free frame) -- only steers how many waves share a 16x16 block.
int ms::rasterize_fwd(int64_t N, int64_t M, int64_t density_hint, const float *means2d, const float *conics,
                      const void *colors, int color_dtype, int CDIM, const float *opacities,
                      const float *backgrounds, int W, int H, int tile_size, int tile_row_begin,
                      int tile_row_end, const int32_t *tile_ranges, const int32_t *flatten_ids,
                      float *render_colors, float *render_alphas, int32_t *last_ids,
                      const ms::LazyLists *lazy, const void *records, const int32_t *order, int clip_row16_begin,
                      int clip_row16_end, void *after_raster_event, void *stream, int32_t *quad_lists, int32_t *quad_counts,
                      void *zero_mem, size_t zero_bytes) {
    MS_REQUIRE(N >= 0 && M >= 0 && M <= 0x7fffffffll, MS_ERR_INVALID_ARG, "rasterize_fwd: bad N/M");
    MS_REQUIRE(!zero_mem || (((uintptr_t)zero_mem & 15) == 0 && (zero_bytes & 15) == 0), MS_ERR_INVALID_ARG,
               "rasterize_fwd: the memory to zero must be 16-byte aligned and a multiple of 16 bytes");
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
    A.order = order;
    A.order_bins = 0;
    // the 16-px block rows to rasterise: the band's tiles, or the caller's clip inside them
    A.row0 = clip_row16_begin >= 0 ? clip_row16_begin : 0;
    A.row1 = clip_row16_begin >= 0 ? clip_row16_end : 0x7fffffff;
    A.records = (CDIM == 3 && ((uintptr_t)records & 15) == 0) ? (const float4 *)records : nullptr;
    if (lazy) A.lazy = *lazy;
    else A.lazy = ms::LazyLists{nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0x7fffffff, 64};
    A.W = W; A.H = H; A.ts = tile_size;
    A.tw = (W + tile_size - 1) / tile_size;
    const int th = (H + tile_size - 1) / tile_size;
    A.nsx = (tile_size + 15) / 16;
    A.nsub = A.nsx * A.nsx;
    A.cdim = CDIM;
    // (the quads' lists for the backward: the plain 3-channel kernel on ready-made records only)
    const bool lists = quad_lists && quad_counts && A.records && !last_ids;
    A.quad_lists = lists ? quad_lists : nullptr;
    A.quad_counts = lists ? quad_counts : nullptr;
    A.quad_nq = 4 * A.nsub;
    A.zero_mem = (float4 *)zero_mem; A.zero_total = zero_mem ? zero_bytes / 16 : 0; A.zero_per_wave = 0;
    MS_REQUIRE(tile_row_begin >= 0 && tile_row_begin <= tile_row_end && tile_row_end <= th,
               MS_ERR_INVALID_ARG, "rasterize_fwd: bad tile row band [%d,%d) of %d", tile_row_begin,
               tile_row_end, th);
    A.tile0 = tile_row_begin * A.tw;
    const int band_tiles = (tile_row_end - tile_row_begin) * A.tw;
    if (band_tiles == 0) {
        if (after_raster_event) (void)hipEventRecord((hipEvent_t)after_raster_event, (hipStream_t)stream);
        return MS_OK;
    }
    const int64_t blocks = (int64_t)band_tiles * A.nsub;
    MS_REQUIRE(blocks <= 0x7fffffff, MS_ERR_TOO_LARGE, "rasterize_fwd: too many tiles");
    A.parts = choose_parts(blocks, density_hint, A.records != nullptr, ((tile_size == 32 && blocks >= 8000) || (tile_size == 64 && blocks >= 16384)) && !last_ids && lazy != nullptr);
    A.nblocks = (int)blocks;
    A.ngrid = (int)blocks;
    if (order) A.ngrid = ((band_tiles + 7) / 8) * 8 * A.nsub;   // tiles dealt over the 8 XCDs, each with its nsub blocks
    A.max_isects = (int)M;
    MS_REQUIRE(N > 0 || M == 0, MS_ERR_INVALID_ARG, "rasterize_fwd: M > 0 with N == 0");
    A.n_gauss = (int)(N < 0x7fffffffll ? (N > 0 ? N : 1) : 0x7fffffffll);
    if (color_dtype == MS_COLOR_F16) return launch_fwd<__half>(A, (hipStream_t)stream, after_raster_event);
    return launch_fwd<float>(A, (hipStream_t)stream, after_raster_event);
}

#ifdef MS_DIAG
extern "C" int ms_diag_set_stamps(void *device_buffer) {
    MS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_diag_stamps), &device_buffer, sizeof(void *)));
    return MS_OK;
}
#endif

extern "C" int ms_rasterize_to_pixels_3dgs_fwd(int64_t N, int64_t M, const float *means2d,
                                               const float *conics, const void *colors,
                                               int color_dtype, int CDIM, const float *opacities,
                                               const float *backgrounds, int W, int H,
                                               int tile_size, int tile_row_begin,
                                               int tile_row_end, const int32_t *tile_ranges,
                                               const int32_t *flatten_ids, float *render_colors,
                                               float *render_alphas, int32_t *last_ids,
                                               void *stream) {
    return ms::rasterize_fwd(N, M, M, means2d, conics, colors, color_dtype, CDIM, opacities, backgrounds, W, H,
                             tile_size, tile_row_begin, tile_row_end, tile_ranges, flatten_ids, render_colors,
                             render_alphas, last_ids, nullptr, nullptr, nullptr, -1, -1, nullptr, stream);
}

int ms::rasterize_fwd_split(int64_t N, int64_t cap, int64_t density_hint, const float *means2d, const float *conics,
                            const void *colors, int color_dtype, int CDIM, const float *opacities,
                            const float *backgrounds, int W, int H, int block_row_begin, int block_row_end,
                            const int32_t *bin_ranges, const ms::BlockLists *lists, float *render_colors,
                            const ms::LazyLists *lazy, const void *records, const int32_t *order,
                            void *after_raster_event, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MS_REQUIRE(N > 0 && cap > 0 && cap <= 0x1fffffffll, MS_ERR_INVALID_ARG, "rasterize_fwd_split: bad N/cap");
    MS_REQUIRE(W > 0 && H > 0 && CDIM >= 1 && CDIM <= 4, MS_ERR_INVALID_ARG, "rasterize_fwd_split: bad sizes");
    MS_REQUIRE(color_dtype == MS_COLOR_F32 || color_dtype == MS_COLOR_F16, MS_ERR_INVALID_ARG,
               "rasterize_fwd_split: unknown colour dtype %d", color_dtype);
    MS_REQUIRE(means2d && conics && colors && opacities && bin_ranges && lists && lists->block_ranges &&
                   lists->block_ids && lists->bin_more && render_colors && lazy && lazy->keys,
               MS_ERR_INVALID_ARG, "rasterize_fwd_split: null pointer");
    MS_REQUIRE(((uintptr_t)means2d & 7) == 0, MS_ERR_INVALID_ARG, "rasterize_fwd_split: means2d must be 8-byte aligned");
    const int tw16 = (W + 15) / 16, th16 = (H + 15) / 16, bw = (W + 31) / 32;
    const int r0 = block_row_begin, r1 = block_row_end;
    MS_REQUIRE(r0 >= 0 && r0 <= r1 && r1 <= th16, MS_ERR_INVALID_ARG, "rasterize_fwd_split: bad block rows [%d,%d) of %d",
               r0, r1, th16);
    if (r1 == r0) {
        if (after_raster_event) (void)hipEventRecord((hipEvent_t)after_raster_event, stream);
        return MS_OK;
    }
    RasterArgs A;
    A.means2d = means2d; A.conics = conics; A.colors = colors; A.opacities = opacities;
    A.backgrounds = backgrounds; A.tile_ranges = lists->block_ranges; A.flatten_ids = lists->block_ids;
    A.render_colors = render_colors; A.render_alphas = nullptr; A.last_ids = nullptr;
    A.order = order;
    A.order_bins = order ? 1 : 0; A.row0 = r0; A.row1 = r1;
    A.quad_lists = nullptr; A.quad_counts = nullptr; A.quad_nq = 4;   // (a split frame is never differentiable)
    A.zero_mem = nullptr; A.zero_per_wave = 0; A.zero_total = 0;
    A.records = (CDIM == 3 && ((uintptr_t)records & 15) == 0) ? (const float4 *)records : nullptr;
    A.lazy = *lazy;
    A.lazy.front_count = nullptr;   // block lists are walked to their end; the bin's flag says whether that was all
    A.lazy.bin_more = lists->bin_more;
    A.lazy.bin_w = bw;
    A.lazy.packed = 0;
    A.W = W; A.H = H; A.ts = 16; A.tw = tw16; A.nsx = 1; A.nsub = 1; A.cdim = CDIM;
    A.tile0 = r0 * tw16;
    const int64_t blocks = (int64_t)(r1 - r0) * tw16;
    A.parts = choose_parts(blocks, density_hint, A.records != nullptr);
    // with an order: four workgroups per 32-px bin of the band (those outside the image or the band leave at once)
    A.nblocks = order ? 4 * ((r1 + 1) / 2 - r0 / 2) * bw : (int)blocks;
    A.ngrid = A.nblocks;
    A.max_isects = (int)(4 * cap);
    A.n_gauss = (int)(N < 0x7fffffffll ? N : 0x7fffffffll);
    if (int rc = color_dtype == MS_COLOR_F16 ? launch_fwd<__half>(A, stream, after_raster_event)
                                              : launch_fwd<float>(A, stream, after_raster_event))
        return rc;
    // clean-up per BIN, from the scatter's unsorted keys (words id << 4 | blocks)
    RasterArgs B = A;
    B.tile_ranges = bin_ranges; B.flatten_ids = nullptr;
    B.ts = 32; B.tw = bw; B.nsx = 2; B.nsub = 4;
    B.max_isects = (int)cap;
    B.lazy.packed = 1;
    B.lazy.row_lo = r0;
    B.lazy.row_hi = r1;
    if (color_dtype == MS_COLOR_F16) launch_redo<__half>(B, stream);
    else launch_redo<float>(B, stream);
    MS_LAUNCH_CHECK();
    return MS_OK;
}
