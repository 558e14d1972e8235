// Tile binning + per-tile depth sort for gfx950.
//
// Replaces gsplat.isect_tiles / isect_offset_encode as the reference calls them
// (mojosplat/binning.py:73-84) and the [start,end) conversion at binning.py:88-100.
// Result contract (bit-exact): intersections ordered by (tile id, float bits of depth,
// Gaussian index) -- what a stable radix sort of (tile<<32 | depth_bits) keys emitted in
// Gaussian order yields.
//
// MI355X design: instead of a 6-pass global LSD radix sort over 64-bit keys (24 B x M of HBM
// traffic per pass), the tile id is resolved by ONE bucket pass whose histograms live in LDS
// (the tile grid of a 1080p frame is 8 160 counters = 32 KB, a 4K frame 130 KB -- both fit the
// 160 KB LDS of a CU), and depth order is established inside each tile segment by an LDS
// counting sort on the depth bits of (depth_bits<<32 | gaussian) keys.  No global atomics
// anywhere: LDS atomics only, so the HBM side sees M x 8 B written once, read once, and M x 4 B
// of ids written.
//
//   k_isect_hist      G workgroups, each histograms a contiguous chunk of Gaussians into LDS
//   k_project_hist    the same fused with the projection itself (what ms_render_fwd runs)
//   k_tile_scan_wg    per tile: exclusive prefix over the G partial counts; on grids up to 4 096 tiles its last
//                     workgroup to arrive also runs the total pass (tile_scan_total) in the same launch.  Round 6: NOT
//                     launched by ms_render_fwd's sync-free frames any more -- k_project_hist claims its stretches of the
//                     tiles' segments with returning atomics (tile_total; profiles/r06_claimed_rows.md); the per-stage
//                     entry points and MOJOSPLAT_CLAIMED_ROWS=0 still use it
//   k_tile_scan_total one workgroup: exclusive scan over tiles -> tile_ranges, M, work lists, launch order, size
//                     record (a launch of its own on larger grids and for empty bands)
//   k_isect_scatter   same chunks; LDS cursors hand out slots inside each tile segment
//   k_tile_sort_small one 256-thread workgroup per tile with <= 1024 entries (16 KB LDS)
//   k_tile_sort_list  work lists of tiles with <= 8192 (74 KB LDS, two per CU) and <= 16384
//                     entries (136 KB); fixed-size persistent grids on sync-free frames
//   k_xl_*            tiles beyond that: LDS-sorted 16384-runs + binary-search merge rounds
//   k_tile_front      lazily sorted frames (ms_render_fwd): only the nearest ~1024 entries of a heavy tile
//                     are selected and sorted; the rasteriser flags tiles whose pixels outlive them.
//                     <false, true> (plain bins): ONE launch over the band's tiles, heaviest first, that also
//                     sorts the short lists whole (instead of k_tile_front + k_tile_sort_small)
//   chunk_of_block    workgroup -> chunk of the count / scatter kernels: XCD-contiguous (partial-write merging)
//   <PACK> / <SPLIT>  kernel variants of split frames: 32-px bins whose keys carry the 4 block bits of
//                     their entry and whose sorted lists leave the sort kernels as four 16x16-block
//                     lists (emit_block_lists)
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

#include "project_device.hpp"

#ifdef MS_DIAG
// Diagnostic build only (python -m mojosplat_amd.csrc.build --diag; never the shipped library): per-workgroup phase
// stamps of the binning kernels (100 MHz chip-wide clock) into a buffer of their own that no other code reads
// (scripts/bin_phases.py).  Kernel k, workgroup b, stamp j -> g_diag_bin[(k * 1024 + b) * 8 + j].
__device__ unsigned long long *g_diag_bin = nullptr;
#define MS_BIN_STAMP(k, j) do { if (g_diag_bin && threadIdx.x == 0 && blockIdx.x < 1024) g_diag_bin[((k) * 1024 + blockIdx.x) * 8 + (j)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MS_BIN_STAMP(k, j) do { } while (0)
#endif

namespace {

#ifndef MS_CHUNK
#define MS_CHUNK 2048
#endif
#ifndef MS_PH_WAVES
#define MS_PH_WAVES 8
#endif
constexpr int kHistThreads = 1024;
#ifndef MS_CLAIM_GROUPS
#define MS_CLAIM_GROUPS 4   // (config 3: count kernel + scatter 52.4 us with four arrays -- a pair of XCDs each -- against 53.1 with eight and 56.2 with two)
#endif
constexpr int kClaimGroups = MS_CLAIM_GROUPS;   // claimed rows: counter arrays per tile grid -- one per XCD (8), per pair (4), ...
constexpr int kClaimShift = kClaimGroups == 8 ? 0 : kClaimGroups == 4 ? 1 : kClaimGroups == 2 ? 2 : 3;
static_assert(kClaimGroups == 8 || kClaimGroups == 4 || kClaimGroups == 2 || kClaimGroups == 1, "claim groups: a power of two <= 8");
constexpr int kMaxG = 512;
constexpr int kSmallCapDecl = 1024, kMediumCapDecl = 8192, kLargeCapDecl = 16384;  // per-tile sort classes
constexpr int kCoopThreshold = 32; // boxes touching more tiles are walked by a whole wave
constexpr int kBandCull = 32;      // bit of the `tight` flags: the band's Gaussians were pre-culled (k_band_precull)
constexpr int kLightBucket = 32;   // length bucket (4 per octave) of the longest LIGHT list: 255 entries
constexpr int kLightCap = 256;     // ... sorted by ONE wave (sort_segment_lds<64, 4>), eight lists to a workgroup
constexpr int kLean = 64;          // ... the frame keeps LeanRecs instead of the projected arrays (internal: ms_render_fwd)
constexpr int kDeferTotal = 128;   // ... the scans' total pass rides in the scatter launch (internal: sync-free frames)
static_assert(kLean == ms::kTightLean && kDeferTotal == ms::kTightDeferTotal, "internal flag bits out of sync");
constexpr size_t kMaxLds = 160 * 1024 - 4096;   // dynamic LDS of the binning kernels: the CU's 160 KB less their static blocks (segment prefix of a band's candidate list, wave totals)

struct Grid {
    int ts, tw, th, row_begin, row_end;
    // block masks (split mode of ms_render_fwd, pipeline.hip): for boxes of at most 16 tiles the reach
    // mask is kept at HALF-tile granularity and every emitted entry carries the 4 bits of its tile's
    // half-tile blocks next to the Gaussian index (id << 4 | blocks): the sorted lists of 32-px bins can
    // then be split into the per-16x16-block lists the rasteriser wants, without testing anything again
    int pack, cw, ch;   // cw x ch: the half-tile (block) grid of the image
};

// What k_project_hist leaves per position for a LEAN frame (ms_render_fwd's plain 3-channel forward frames, whose
// rasteriser reads the ready-made records and whose projected arrays nobody reads): the Gaussian's tile box on
// the binning grid, clamped to the band -- x0 | y0 << 16, x1 | y1 << 16 -- the bits of its depth, and the tile
// count n of the box (0: culled, nothing to emit) with the block-mask mode's edge flags in bits 28-31.  16 bytes
// instead of means2d + conics + depth + radii (32): the scatter kernel reads ONE dwordx4 per Gaussian and redoes
// no box arithmetic.  (Grids beyond 65 535 tiles a side keep the full arrays.)
struct LeanRec {
    uint32_t xy0, xy1, depth_bits, n_edges;
};
// ... and on plain bins of grids up to 255 tiles a side (LEAN == 2) ONE 12-byte record carries the reach mask as
// well: box = x0 | y0 << 8 | w << 16 | h << 24, the depth bits, the low 32 bits of the reach mask (boxes of 33-64
// tiles keep every tile: count and scatter kernel agree by construction, the image does not depend on it) -- 12
// bytes written and 12 read per Gaussian instead of 16 + 8.
struct Lean12 {
    uint32_t box, depth_bits, mask32;
};

// Tight binning: which tiles of a Gaussian's box can its alpha >= 1/255 ellipse reach at all?
// bit (y - y0) * w + (x - x0) of the result, for boxes of at most 64 tiles (larger boxes, and
// Gaussians without a usable bound, keep every tile: all ones).
//
// E = {d : ca dx^2 + 2 cb dx dy + cc dy^2 <= 2 s}, s = ln(255 o) with the rasteriser's slack.  For a
// tile row, the pixel centres span dy in [yl, yh] (relative to the mean); E restricted to that
// band is convex, so its x-extent is an interval [xmin, xmax]: the upper root
// u(dy) = (-cb dy + sqrt(2 ca s - det dy^2)) / ca is concave with its maximum at the ellipse's
// rightmost point dy = -cb/cc * X (X = sqrt(2 cc s / det)), clamped into the band; the lower root
// mirrors it.  A tile is reached iff its pixel-centre interval meets [xmin, xmax].  Two square
// roots per tile ROW instead of a rectangle test per TILE, no divergent inner loop.  The interval is
// padded by 0.01 px so rounding can only keep a pair too many, never drop one the rasteriser
// would blend.  Computed ONCE per Gaussian by the count kernel and stored (8 B): the scatter kernel
// reads it back, so the two agree by construction.
__device__ __forceinline__ unsigned long long reach_mask(float mx, float my, float ca, float cb, float cc,
                                                        float opacity, int x0, int x1, int y0, int y1, int ts) {
    const int w = x1 - x0, h = y1 - y0;
    const float det = ca * cc - cb * cb;
    if (w * h > 64 || w * h <= 1 || !(det > 0.f) || !(ca > 0.f) || !(cc > 0.f) || !(opacity >= ms::kAlphaThreshold))
        return ~0ull;
    const float s2 = 2.0f * (__logf(opacity * 255.0f) * 1.0001f + 1e-4f);   // 2 s
    const float inv_det = __builtin_amdgcn_rcpf(det), inv_ca = __builtin_amdgcn_rcpf(ca);
    const float Y = __builtin_amdgcn_sqrtf(ca * s2 * inv_det);                                // |dy| extent of E
    const float dy_right = -(cb * __builtin_amdgcn_rcpf(cc)) * __builtin_amdgcn_sqrtf(cc * s2 * inv_det);            // dy of E's rightmost point
    const float fts = (float)ts, span = (float)(ts - 1), inv_ts = __builtin_amdgcn_rcpf(fts);
    unsigned long long mask = 0;
    for (int r = 0; r < h; ++r) {
        const float yl = (float)((y0 + r) * ts) + 0.5f - my, yh = yl + span;
        const float lo = fmaxf(yl, -Y), hi = fminf(yh, Y);
        if (lo > hi + 0.01f) continue;                                      // the band misses E
        const float dyr = fminf(fmaxf(dy_right, lo), hi), dyl = fminf(fmaxf(-dy_right, lo), hi);
        const float xmax = (-cb * dyr + __builtin_amdgcn_sqrtf(fmaxf(ca * s2 - det * dyr * dyr, 0.f))) * inv_ca + 0.01f;
        const float xmin = (-cb * dyl - __builtin_amdgcn_sqrtf(fmaxf(ca * s2 - det * dyl * dyl, 0.f))) * inv_ca - 0.01f;
        // tiles whose centre interval [tx*ts + 0.5 - mx, + span] meets [xmin, xmax]
        const float base = 0.5f - mx;
        // (approximate reciprocals / roots: their error is orders of magnitude inside the 0.01 px padding)
        int ta = (int)ceilf((xmin - base - span) * inv_ts - 1e-3f), tb = (int)floorf((xmax - base) * inv_ts + 1e-3f);
        ta = max(ta, x0); tb = min(tb, x1 - 1);
        if (ta > tb) continue;
        const unsigned long long rowbits = (tb - ta + 1 >= 64) ? ~0ull : ((1ull << (tb - ta + 1)) - 1ull);
        mask |= rowbits << (r * w + (ta - x0));
    }
    return mask;
}

// Workgroup -> chunk of the count / scatter kernels.  A tile's segment of the key buffer is laid out chunk by
// chunk (the prefix over the partial histograms), each chunk contributing an entry or two: with chunk = blockIdx
// the eight XCDs (blocks are dealt round-robin over them; their L2s do not share lines) interleave at 16-byte
// granularity inside every 64-byte line, and each line leaves as several partial writes (k_isect_scatter: 96 % of
// its write requests were 32-byte partials, profiles/r01_pmc_binning.md).  Giving XCD x the x-th contiguous
// EIGHTH of the chunks makes one L2 own whole stretches of every segment, so lines are completed before they are
// written back.  Any bijection is correct (both kernels use the same one); the XCD placement is speed only.
__device__ __forceinline__ int chunk_of_block(int b, int G) {
    const int n8 = G >> 3, rem = G & 7, x = b & 7, k = b >> 3;
    return x * n8 + min(x, rem) + k;
}

// Which POSITIONS a workgroup of the count / scatter kernels walks (round 5).  Rounds 1-4 gave workgroup g the g-th
// contiguous chunk of the Gaussians.  That is fine for a scene stored in random order -- every chunk is a sample of the
// whole scene -- and wrong for one stored in a spatially coherent order (Morton-sorted checkpoints are common): a chunk is
// then one cell of space, and the work of a cell ranges from nothing (off screen, behind the camera, behind its bins'
// depth cut-offs) to several times the mean (near the camera: large boxes, every pair kept).  With one or two workgroups
// per CU and no dynamic scheduling the launch ends with its heaviest chunk: measured on the Morton-sorted config 5,
// k_isect_scatter 32 -> 93 us and k_project_hist 122 -> 140 (profiles/r04_morton_probe.jsonl).  Now the positions are DEALT:
// slices of kDeal consecutive positions go round-robin over the workgroups, slice s to workgroup s mod G, so every
// workgroup samples the whole array -- kHistThreads / kDeal slices per trip of its loop, sixteen at the default of one
// slice per wave -- whatever its order.  Loads and stores stay coalesced per wave (64 consecutive positions: 768 B of
// means, 3 KB of records).  Which workgroup counts a position and which scatters it must agree (a workgroup's histogram row
// is its stretch of every tile's segment): both kernels -- and a band's candidate walk -- go through this one mapping.
#ifndef MS_DEAL
#define MS_DEAL 64
#endif
#ifndef MS_PH_TICKETS   // k_project_hist: a workgroup's slices go to its waves by ticket (0: two fixed slices a wave, as Deal lays them out)
#define MS_PH_TICKETS 1
#endif
constexpr int kDeal = MS_DEAL;
static_assert(kDeal >= 64 && kDeal <= 1024 && (kDeal & (kDeal - 1)) == 0, "a slice is a power-of-two number of waves");
struct Deal {
    int64_t n, first, stride;
    int iters;
    // n_pos positions, workgroup wg of G (every wave of the workgroup; wave-uniform members)
    __device__ __forceinline__ Deal(int64_t n_pos, int wg, int G) {
        constexpr int kWavesPerSlice = kDeal / 64, kParts = kHistThreads / kDeal;
        const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int part = w / kWavesPerSlice, v = w - part * kWavesPerSlice;
        n = n_pos;
        first = ((int64_t)part * G + wg) * kDeal + (int64_t)v * 64;          // trip 0: slice part * G + wg
        stride = (int64_t)kParts * G * kDeal;                                 // = G * kHistThreads positions per trip of the grid
        iters = (int)((n_pos + stride - 1) / stride);                         // uniform over the grid
    }
    __device__ __forceinline__ int64_t base(int it) const { return first + (int64_t)it * stride; }   // this wave's 64 positions of trip `it`
};

__device__ __forceinline__ int clampi(float v, int lo, int hi) {
    if (!(v > (float)lo)) return lo;  // also NaN
    if (v >= (float)hi) return hi;
    return (int)v;
}

// gsplat tile bounding box: min inclusive, max exclusive, clamped to the grid, then to the band
__device__ __forceinline__ bool tile_bbox(float2 m, int2 r, const Grid &g, int &x0, int &x1,
                                          int &y0, int &y1) {
    const float fts = (float)g.ts;
    float trx, try_, tx, ty;
    if ((g.ts & (g.ts - 1)) == 0) {
        // a power-of-two tile size (the usual case): x * (1 / ts) IS x / ts, bit for bit, and four IEEE
        // divisions (~10 VALU instructions each) leave the per-Gaussian path of the binning kernels
        const float inv = 1.0f / fts;
        trx = (float)r.x * inv; try_ = (float)r.y * inv;
        tx = m.x * inv; ty = m.y * inv;
    } else {
        trx = (float)r.x / fts; try_ = (float)r.y / fts;
        tx = m.x / fts; ty = m.y / fts;
    }
    x0 = clampi(floorf(tx - trx), 0, g.tw);
    x1 = clampi(ceilf(tx + trx), 0, g.tw);
    y0 = clampi(floorf(ty - try_), 0, g.th);
    y1 = clampi(ceilf(ty + try_), 0, g.th);
    const bool on_grid = x1 > x0 && y1 > y0;  // before the band clamp: touches some tile of the frame
    y0 = max(y0, g.row_begin);
    y1 = min(y1, g.row_end);
    y1 = max(y1, y0);
    return on_grid;
}

// Per-workgroup count of Gaussians whose box touches the full grid (whatever the band): lets a
// band-sharded caller apply the frame-level "nothing intersects -> zeros image" rule locally.
__device__ __forceinline__ void count_on_grid(bool on_grid, unsigned int *s_on_grid) {
    const unsigned long long b = __ballot(on_grid);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(s_on_grid, (unsigned int)__popcll(b));
}

// The tile box of one Gaussian for either mode.  Block-mask mode (g.pack) derives it from the gsplat box
// on the half-tile grid -- the lists must hold exactly what 16-px tiles would -- and reports in `edges`
// which outer half-columns / half-rows of the box lie outside that finer box (bit 0 left, 1 right, 2 top,
// 3 bottom).
template <bool PACK>
__device__ __forceinline__ bool bin_box(float2 m, int2 r, const Grid &g, int &x0, int &x1, int &y0, int &y1,
                                        int &edges) {
    edges = 0;
    if constexpr (!PACK) return tile_bbox(m, r, g, x0, x1, y0, y1);
    const Grid gc{g.ts >> 1, g.cw, g.ch, 0, g.ch, 0, 0, 0};
    int cx0, cx1, cy0, cy1;
    const bool on = tile_bbox(m, r, gc, cx0, cx1, cy0, cy1);
    x0 = cx0 >> 1; x1 = (cx1 + 1) >> 1;
    y0 = cy0 >> 1; y1 = (cy1 + 1) >> 1;
    if (!on) { x1 = x0; y1 = y0; }
    y0 = max(y0, g.row_begin);
    y1 = min(y1, g.row_end);
    y1 = max(y1, y0);
    edges = (cx0 > 2 * x0 ? 1 : 0) | (cx1 < 2 * x1 ? 2 : 0) | (cy0 > 2 * y0 ? 4 : 0) | (cy1 < 2 * y1 ? 8 : 0);
    return on;
}

// the blocks (2x2 per tile, bit = 2*dy + dx) of box tile (c, r) that lie inside the finer box
__device__ __forceinline__ int edge_blocks(int edges, int c, int r, int w, int h) {
    int q = 0xf;
    if ((edges & 1) && c == 0) q &= 0xa;
    if ((edges & 2) && c == w - 1) q &= 0x5;
    if ((edges & 4) && r == 0) q &= 0xc;
    if ((edges & 8) && r == h - 1) q &= 0x3;
    return q;
}

// cell mask of a box of w x h tiles (2w x 2h cells) with the edge half-columns / half-rows removed
__device__ __forceinline__ unsigned long long clip_cells(unsigned long long m, int edges, int w, int h) {
    const int cwid = 2 * w, rows = 2 * h;
    if (cwid * rows < 64) m &= (1ull << (cwid * rows)) - 1ull;
    if (edges & 4) m &= ~((1ull << cwid) - 1ull);
    if (edges & 8) m &= (1ull << ((rows - 1) * cwid)) - 1ull;
    if (edges & 3) {
        unsigned long long p = 1ull;   // bit 0 of every row
        for (int sh = cwid; sh < 64; sh <<= 1) p |= p << sh;
        if (edges & 1) m &= ~p;
        if (edges & 2) m &= ~(p << (cwid - 1));
    }
    return m;
}

// One step of a chunk walk: every lane of the workgroup brings the tile box of ONE Gaussian
// (index gi, n = 0 if it has none); F(local_tile, gaussian_index) is called for
// every tile of every box.  Small boxes are walked by their own lane, big ones by the whole wave.
// Must be reached by all lanes of the wave (ballot / shuffles inside).
// `payload`: a word of the box's OWNER that f receives as a fourth argument (the lean scatter's depth bits: a
// big box is walked by the whole wave, whose other lanes know the owner's index but not its registers).
// Boxes of more than kCoopThreshold tiles, queued per WORKGROUP (round 5).  Such a box is walked by a whole wave, a round
// of 64 tiles at a time, each round an LDS round trip.  A scene in random order gives every wave one or two of them; a
// spatially coherent order puts the boxes next to the camera into a handful of waves -- Morton-sorted config 3: one slice
// of 64 positions holds 11 boxes of up to 756 tiles, 45 rounds, and its wave ran 12 us behind the other fifteen of its
// workgroup in the count kernel and again in the scatter kernel (per-workgroup stamps: profiles/r05_bin_phases_morton.txt).
// With a queue the owner only leaves a 32-byte entry in LDS; after the workgroup's walk its sixteen waves drain the queue
// round-robin.  Entries that find the queue full (a scene of nothing but huge footprints: every wave is equally loaded
// then) are walked on the spot, as before.
struct BigQ {
    uint32_t *ent;      // LDS: cap entries of 8 words: x0 | y0 << 16, x1 | y1 << 16, index, payload, mask lo, mask hi, edges, -
    unsigned int *count;
    unsigned int cap;   // 0: no queue
};

template <bool PACK, bool PAYLOAD, class F>
__device__ __forceinline__ void walk_boxes_impl(int gi, int x0, int x1, int y0, int y1, int n, int edges,
                                                const Grid &g, unsigned long long mask, F &&f, uint32_t payload,
                                                const BigQ *Q = nullptr) {
    auto call = [&](int t, int64_t i, int q, uint32_t pl) __attribute__((always_inline)) {
        if constexpr (PAYLOAD) f(t, i, q, pl);
        else f(t, i, q);
    };
    const int lane = threadIdx.x & 63;
    const int64_t i = gi;   // this lane's Gaussian (its index in the arrays; a band's candidate list maps to it)
    const bool big = n > kCoopThreshold;
    if (n > 0 && !big) {
        // only the reached tiles: the trip count is popcount, not the box area
        const int w = x1 - x0;
        const float inv_w = __builtin_amdgcn_rcpf((float)w);   // (1 ulp is far inside the 0.5 / w margin below)
        unsigned int m;
        const bool cells = PACK && n <= 16;   // `mask` is per half-tile cell (2w cells per row): fold to tiles
        if (cells) {
            const int cw = 2 * w, h = y1 - y0;
            const unsigned int rowmask = cw >= 32 ? 0xffffffffu : ((1u << cw) - 1u);
            m = 0;
            for (int r = 0; r < h; ++r) {
                unsigned int t = ((unsigned int)(mask >> (2 * r * cw)) | (unsigned int)(mask >> ((2 * r + 1) * cw))) & rowmask;
                t = (t | (t >> 1)) & 0x55555555u;   // bit 2j: either cell column of tile j ...
                t = (t | (t >> 1)) & 0x33333333u;   // ... compressed to the low w bits
                t = (t | (t >> 2)) & 0x0f0f0f0fu;
                t = (t | (t >> 4)) & 0x00ff00ffu;
                t = (t | (t >> 8)) & 0x0000ffffu;
                m |= t << (r * w);
            }
        } else {
            m = (unsigned int)mask & (n >= 32 ? 0xffffffffu : ((1u << n) - 1u));   // n <= kCoopThreshold = 32
        }
        while (m) {
            const int k = __ffs((int)m) - 1;
            m &= m - 1;
            const int r = (int)(((float)k + 0.5f) * inv_w);   // k / w, exact for k < 64
            const int c = k - r * w;
            int q;
            if (cells) {
                const int cw = 2 * w, b00 = 2 * r * cw + 2 * c;
                q = (int)((mask >> b00) & 3ull) | ((int)((mask >> (b00 + cw)) & 3ull) << 2);
            } else {
                q = PACK ? edge_blocks(edges, c, r, w, y1 - y0) : 0xf;
            }
            call((y0 + r - g.row_begin) * g.tw + x0 + c, i, q, payload);
        }
    }
    // Round 5: the owner's registers come over by v_readlane (the owner is wave-uniform: a scalar out of the ballot) and a
    // tile's row by a float reciprocal -- rounds 1-4 fetched them with eight ds_bpermute round trips (~100 cycles each, one
    // after the other) and divided by w in integers (~40 instructions a tile).  It mattered once scenes came in a spatially
    // coherent order: the 237 boxes of more than 32 tiles that config 3 holds sit next to the camera, a Morton curve puts
    // up to 45 of them into ONE wave's 64 positions (12 in the given order), and that wave's ~1 100 cycles per box were
    // the tail of the count kernel and of the scatter kernel (+5 us each; profiles/r05_morton_prof_cfg3.txt).
    bool queued = false;
    if (Q && Q->cap && big) {
        const unsigned int slot = atomicAdd(Q->count, 1u);
        if (slot < Q->cap) {
            uint32_t *e = Q->ent + 8u * slot;
            e[0] = (uint32_t)x0 | ((uint32_t)y0 << 16); e[1] = (uint32_t)x1 | ((uint32_t)y1 << 16);
            e[2] = (uint32_t)gi; e[3] = payload;
            e[4] = (uint32_t)mask; e[5] = (uint32_t)(mask >> 32); e[6] = (uint32_t)edges;
            queued = true;
        }
    }
    unsigned long long bigmask = __ballot(big && !queued);
    while (bigmask) {
        const int src = __ffsll((long long)bigmask) - 1;
        bigmask &= bigmask - 1;
        const int bx0 = __builtin_amdgcn_readlane(x0, src), bx1 = __builtin_amdgcn_readlane(x1, src);
        const int by0 = __builtin_amdgcn_readlane(y0, src), by1 = __builtin_amdgcn_readlane(y1, src);
        const unsigned long long bm = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mask >> 32), src) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)(mask & 0xffffffffu), src);
        const int be = PACK ? __builtin_amdgcn_readlane(edges, src) : 0;
        const int64_t bi = __builtin_amdgcn_readlane(gi, src);
        const uint32_t bp = PAYLOAD ? (uint32_t)__builtin_amdgcn_readlane((int)payload, src) : 0u;
        const int w = bx1 - bx0, cnt = w * (by1 - by0);
        if (w <= 256 && cnt <= 65536) {   // (uniform) k / w by reciprocal: exact -- (k + 0.5) / w is >= 0.5 / w off any integer, the product's error < 1e-4
            const float inv_w = __builtin_amdgcn_rcpf((float)w);
            for (int k = lane; k < cnt; k += 64) {
                const int r = (int)(((float)k + 0.5f) * inv_w), c = k - r * w;
                if (cnt > 64 || ((bm >> k) & 1ull))
                    call((by0 + r - g.row_begin) * g.tw + bx0 + c, bi, PACK ? edge_blocks(be, c, r, w, by1 - by0) : 0xf, bp);
            }
        } else {
            for (int k = lane; k < cnt; k += 64) {
                const int r = k / w, c = k % w;
                call((by0 + r - g.row_begin) * g.tw + bx0 + c, bi, PACK ? edge_blocks(be, c, r, w, by1 - by0) : 0xf, bp);
            }
        }
    }
}

template <bool PACK, class F>
__device__ __forceinline__ void walk_boxes(int gi, int x0, int x1, int y0, int y1, int n, int edges,
                                           const Grid &g, unsigned long long mask, F &&f, const BigQ *Q = nullptr) {
    walk_boxes_impl<PACK, false>(gi, x0, x1, y0, y1, n, edges, g, mask, f, 0u, Q);
}
template <bool PACK, class F>
__device__ __forceinline__ void walk_boxes(int gi, int x0, int x1, int y0, int y1, int n, int edges,
                                           const Grid &g, unsigned long long mask, F &&f, uint32_t payload, const BigQ *Q = nullptr) {
    walk_boxes_impl<PACK, true>(gi, x0, x1, y0, y1, n, edges, g, mask, f, payload, Q);
}

// The queued boxes of a workgroup, its waves taking them round-robin: f(tile, index, blocks[, payload]) for every tile of
// every box, as walk_boxes would have called it.  ALL threads of the workgroup (a barrier inside).
template <bool PACK, bool PAYLOAD, class F>
__device__ __forceinline__ void drain_big_boxes(const BigQ &Q, const Grid &g, F &&f) {
    if (!Q.cap) return;   // (uniform)
    __syncthreads();      // every push is in
    const unsigned int total = min(*Q.count, Q.cap);
    const int lane = threadIdx.x & 63;
    const unsigned int nw = blockDim.x >> 6;
    for (unsigned int e = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); e < total; e += nw) {
        const uint32_t *q = Q.ent + 8u * e;   // (one address for the wave: an LDS broadcast)
        const int bx0 = (int)(q[0] & 0xffffu), by0 = (int)(q[0] >> 16), bx1 = (int)(q[1] & 0xffffu), by1 = (int)(q[1] >> 16);
        const int64_t bi = (int64_t)(int32_t)q[2];
        const uint32_t bp = q[3];
        const unsigned long long bm = (unsigned long long)q[4] | ((unsigned long long)q[5] << 32);
        const int be = (int)q[6];
        const int w = bx1 - bx0, cnt = w * (by1 - by0);
        const float inv_w = __builtin_amdgcn_rcpf((float)max(w, 1));
        const bool by_rcp = w <= 256 && cnt <= 65536;   // (see walk_boxes_impl)
        for (int k = lane; k < cnt; k += 64) {
            const int r = by_rcp ? (int)(((float)k + 0.5f) * inv_w) : k / w, c = k - r * w;
            if (cnt > 64 || ((bm >> k) & 1ull)) {
                const int t = (by0 + r - g.row_begin) * g.tw + bx0 + c;
                const int qb = PACK ? edge_blocks(be, c, r, w, by1 - by0) : 0xf;
                if constexpr (PAYLOAD) f(t, bi, qb, bp);
                else f(t, bi, qb);
            }
        }
    }
}

// Walk every (Gaussian, tile) pair of one chunk; F(local_tile, gaussian_index).
// Small boxes are walked by their own lane, big ones by the whole wave.
// A band's candidate list (multi-GPU ranks, k_band_precull): segment g of `ids` (capacity seg_cap, the chunk of
// Gaussians workgroup g of the pre-cull looked at) holds seg_count[g] survivors, compacted.  The count and
// scatter kernels walk the concatenation of the segments ("positions") in steps of kHistThreads, step s going to
// workgroup s mod G -- the same assignment in both kernels, so a workgroup's histogram row matches what it
// scatters -- and find a position's Gaussian through the prefix over the <= kMaxG segment counts (LDS).
struct Candidates {
    const int32_t *ids;     // null: every Gaussian, position == index
    const int32_t *seg_count;
    int n_segs;
    int64_t seg_cap;
};

struct CandMap {
    uint32_t *pref;   // LDS: pref[g] = survivors in segments < g; pref[n_segs] = total
    __device__ __forceinline__ void build(const Candidates &c) {   // all threads of the workgroup; ends with a barrier
        // n_segs <= kMaxG <= blockDim: one segment per thread, a wave scan + the wave totals
        __shared__ uint32_t s_wtot[16];
        const int t = threadIdx.x, lane = t & 63, w = t >> 6;
        const uint32_t v = t < c.n_segs ? (uint32_t)c.seg_count[t] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wtot[w] = incl;
        __syncthreads();
        uint32_t run = 0;
        for (int ww = 0; ww < w; ++ww) run += s_wtot[ww];
        if (t < c.n_segs) pref[t + 1] = run + incl;
        if (t == 0) pref[0] = 0;
        __syncthreads();
    }
    __device__ __forceinline__ int64_t gaussian(const Candidates &c, int64_t pos) const {
        int lo = 0, hi = c.n_segs;   // largest g with pref[g] <= pos
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if ((int64_t)pref[mid] <= pos) lo = mid; else hi = mid;
        }
        return (int64_t)c.ids[(int64_t)lo * c.seg_cap + (pos - (int64_t)pref[lo])];
    }
};

template <bool PACK, class F>
__device__ __forceinline__ void for_each_isect(int64_t i0, int64_t i1, const float *means2d,
                                               const int32_t *radii, const unsigned long long *masks,
                                               const Grid &g,
                                               int32_t *tiles_per_gauss, unsigned int *s_on_grid, Candidates cand, int G, F &&f) {
    __shared__ uint32_t s_pref[kMaxG + 1];
    CandMap map{s_pref};
    (void)i0;
    if (cand.ids) {   // positions of the candidate list
        map.build(cand);
        i1 = (int64_t)s_pref[cand.n_segs];
    }
    // (i1: the number of positions -- Gaussians, or a band's candidates; dealt over the workgroups, see Deal)
    const Deal deal(i1, chunk_of_block(blockIdx.x, G), G);
    for (int it = 0; it < deal.iters; ++it) {
        const int64_t j = deal.base(it) + (threadIdx.x & 63);
        int x0 = 0, x1 = 0, y0 = 0, y1 = 0, n = 0, edges = 0, gi = 0;
        bool on_grid = false;
        unsigned long long mask = ~0ull;
        if (j < i1) {
            // (a band's candidates: the projected arrays are indexed by POSITION in the candidate list -- the
            // count kernel wrote them that way, coalesced -- and so are the entries it emits)
            const int64_t i = j;
            gi = (int)i;
            const int2 r = reinterpret_cast<const int2 *>(radii)[i];
            if (r.x > 0 && r.y > 0) {
                const float2 m = reinterpret_cast<const float2 *>(means2d)[i];
                on_grid = bin_box<PACK>(m, r, g, x0, x1, y0, y1, edges);
                n = (x1 - x0) * (y1 - y0);
                if (masks && (n > 1 || PACK)) mask = masks[i];
            }
            if (tiles_per_gauss) tiles_per_gauss[i] = n;
        }
        if (s_on_grid) count_on_grid(on_grid, s_on_grid);
        walk_boxes<PACK>(gi, x0, x1, y0, y1, n, edges, g, mask, f);
    }
}

// Multi-GPU bands: which Gaussians can reach the band's rows [y_lo, y_hi) px (and the image's columns) at all?
// A rank that renders 1/8 of the frame otherwise projects and walks ALL Gaussians (config 5: 190 of the ~330 us
// of a band frame).  Conservative test from the mean and the largest scale alone: the camera-space covariance
// has lambda_max <= ||Wv||^2 smax^2, the Jacobian's rows have |J_y|^2 = (fy / z)^2 (1 + v^2) with v the
// FOV-clamped my / z, so cov2d_yy + eps <= that product + eps, and the tile box's radius is
// ceil(extend sqrt(.)) <= 3.33 sqrt(.) + 1.  Everything is padded (0.1 % + 2 px); what is dropped here
// has no tile in the band, what is kept goes through the exact projection.  Survivors are compacted per wave
// (one atomic per wave); their order only moves the scatter's slots, never the sorted lists.
// Round 5, PREPARED scenes (ms_scene_prepare: the Gaussians stored in a spatially coherent order, with the bounding box of
// the means and the largest scale of every block of 2^block_shift of them): a workgroup first judges the blocks its chunk
// touches -- one lane per block: the box's eight corners through the view matrix; behind the near plane -> keep; else the
// screen rectangle of the corners (a perspective image of a convex set in front of the camera is the hull of its corners')
// padded by the radius bound below taken at the box's nearest depth, its largest scale and the FOV clamp's limit -- and its
// waves then skip the loads of every block that cannot reach the band: at config 5 cut 8 ways the centre band skips 54 %
// of the 120 MB this pass streams, an edge band 92 % (the pass was 32 of a band frame's 148 us whatever the band).
constexpr int kMaxBlocksPerChunk = 4096;
#ifndef MS_PRECULL_SUB
#define MS_PRECULL_SUB 4
#endif
#ifndef MS_PRECULL_SUB_PREPARED   // (a prepared scene's sub-step is ONE 16-byte load: four registers)
#define MS_PRECULL_SUB_PREPARED 4
#endif
// (8 waves a SIMD: at 98 scalar registers the kernel held 7, i.e. ONE 1 024-thread workgroup a CU and two rounds of its 512)
template <bool PREPARED>   // a prepared scene: block verdicts first, the survivors' 16-byte pre-cull records instead of means + scales
__global__ __launch_bounds__(kHistThreads, 8) void k_band_precull(int64_t N, const float *__restrict__ means3d,
                                                               const float *__restrict__ scales,
                                                               const float *__restrict__ viewmat, ms::ProjParams P,
                                                               float y_lo, float y_hi, int64_t chunk,
                                                               int32_t *__restrict__ cand,
                                                               int32_t *__restrict__ seg_count,
                                                               const float *__restrict__ block_bounds, int block_shift) {
    constexpr int kSub = PREPARED ? MS_PRECULL_SUB_PREPARED : MS_PRECULL_SUB;
    __shared__ uint32_t s_w[2][kSub * 16];   // (two buffers, alternating: ONE barrier a round -- a wave writes a buffer again only
                                             // behind the next round's barrier, which every wave reaches after it has read this one)
    __shared__ unsigned char s_blk[kMaxBlocksPerChunk];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    MS_BIN_STAMP(4, 0);
    float V[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) V[k] = viewmat[k];
    // ||Wv||_2^2 <= the largest Gershgorin row sum of Wv^T Wv (exactly 1 for a rotation)
    float lam = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float rowsum = 0.f;
#pragma unroll
        for (int b = 0; b < 3; ++b) rowsum += fabsf(V[a] * V[b] + V[4 + a] * V[4 + b] + V[8 + a] * V[8 + b]);
        lam = fmaxf(lam, rowsum);
    }
    const int64_t i0 = (int64_t)blockIdx.x * chunk, i1 = min(N, i0 + chunk);
    int32_t *seg = cand + i0;
    uint32_t written = 0;   // survivors of this segment so far (uniform)
    const int64_t blk0 = PREPARED ? (i0 >> block_shift) : 0;
    // (a prepared scene's buffer: the bounds of its ceil(N / block) blocks, then a 16-byte pre-cull record per Gaussian)
    // (16-byte aligned: the buffer is, and a block's bounds are 32 bytes)
    const float4 *__restrict__ cull = PREPARED ? reinterpret_cast<const float4 *>(__builtin_assume_aligned(block_bounds + 8 * (((N - 1) >> block_shift) + 1), 16)) : nullptr;
    const bool blocks = PREPARED && i1 > i0 && ((i1 - 1) >> block_shift) - blk0 < kMaxBlocksPerChunk;   // (uniform)
    if (blocks) {
        const int nb = (int)(((i1 - 1) >> block_shift) - blk0) + 1;
        const float vlx = fmaxf(P.lim_x_pos, P.lim_x_neg), vly = fmaxf(P.lim_y_pos, P.lim_y_neg);
        int any_kept = 0;
        for (int b = threadIdx.x; b < nb; b += kHistThreads) {
            const float *q = block_bounds + 8 * (blk0 + b);
            const float lo[3] = {q[0], q[1], q[2]}, hi[3] = {q[4], q[5], q[6]}, smax = q[3];
            float zmin = 3.0e38f, zmax = -3.0e38f, xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
            bool odd = false;   // a corner at or behind the near plane, or not a number: no screen bound
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float p0 = (c & 1) ? hi[0] : lo[0], p1 = (c & 2) ? hi[1] : lo[1], p2 = (c & 4) ? hi[2] : lo[2];
                const float mx = V[0] * p0 + V[1] * p1 + V[2] * p2 + V[3];
                const float my = V[4] * p0 + V[5] * p1 + V[6] * p2 + V[7];
                const float z = V[8] * p0 + V[9] * p1 + V[10] * p2 + V[11];
                zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
                odd = odd || !(z > 0.5f * P.near_plane);
                const float rz = __builtin_amdgcn_rcpf(fmaxf(z, 1e-12f));
                const float xs = P.fx * mx * rz + P.cx, ys = P.fy * my * rz + P.cy;
                xmin = fminf(xmin, xs); xmax = fmaxf(xmax, xs);
                ymin = fminf(ymin, ys); ymax = fmaxf(ymax, ys);
            }
            bool keep = true;
            if (zmax < P.near_plane * 0.999f || zmin > P.far_plane * 1.001f) keep = false;   // every mean fails the exact depth test
            else if (!odd) {
                const float rz = __builtin_amdgcn_rcpf(zmin), s2 = lam * smax * smax;
                const float fxz = P.fx * rz, fyz = P.fy * rz;
                const float rx = 3.33f * __builtin_amdgcn_sqrtf(fxz * fxz * (1.0f + vlx * vlx) * s2 + P.eps2d) * 1.002f + 3.0f;
                const float ry = 3.33f * __builtin_amdgcn_sqrtf(fyz * fyz * (1.0f + vly * vly) * s2 + P.eps2d) * 1.002f + 3.0f;
                // (negated comparisons: a NaN anywhere keeps the block for the per-Gaussian test to judge)
                keep = !(ymax + ry < y_lo || ymin - ry > y_hi || xmax + rx < 0.f || xmin - rx > P.W);
            }
            s_blk[b] = keep ? 1 : 0;
            any_kept |= keep ? 1 : 0;
        }
        // (a chunk none of whose blocks can reach the band -- 212 of the centre band's 512 at config 5, more at an edge -- is
        // done: its three empty rounds cost 8 us of sixteen waves that the other lane's kernels can use)
        if (!__syncthreads_or(any_kept)) {
            if (threadIdx.x == 0) seg_count[blockIdx.x] = 0;
            MS_BIN_STAMP(4, 1);
            MS_BIN_STAMP(4, 2);
            return;
        }
    }
    MS_BIN_STAMP(4, 1);
    // kSub sub-steps of 1024 Gaussians share one pair of barriers (the pass is a chain of load -> test -> count
    // round trips, not a bandwidth problem: LDS-staged 16-byte loads made it 1.5x SLOWER)
    int round = 0;
    for (int64_t base = i0; base < i1; base += (int64_t)kSub * kHistThreads) {
        bool keep[kSub];
        unsigned long long bal[kSub];
        // ALL the sub-steps' loads first, unconditionally, on an index that is always valid (a lane past the chunk's end or in
        // a rejected block fetches the chunk's first Gaussian: one cached line): a load inside `if (live)` cannot be hoisted
        // above the previous sub-step's test, and a wave then pays one memory round trip per sub-step -- measured with
        // per-workgroup stamps, a workgroup whose 9 766 Gaussians all survive the block verdicts took 22 us for its ten
        // steps whether the other workgroups were loading or not (scripts/precull_phases.py).
        bool live[kSub];
        float p0[kSub], p1[kSub], p2[kSub], s0[kSub], s1[kSub], s2_[kSub];
#pragma unroll
        for (int k = 0; k < kSub; ++k) {
            const int64_t i = base + (int64_t)k * kHistThreads + threadIdx.x;
            live[k] = i < i1 && (!blocks || s_blk[(min(i, i1 - 1) >> block_shift) - blk0]);
            const int64_t j = live[k] ? i : i0;
            if constexpr (PREPARED) {   // mean + largest linear scale in ONE 16-byte load
                const float4 r = cull[j];
                p0[k] = r.x; p1[k] = r.y; p2[k] = r.z; s0[k] = r.w; s1[k] = 0.f; s2_[k] = 0.f;
            } else {
                p0[k] = means3d[3 * j]; p1[k] = means3d[3 * j + 1]; p2[k] = means3d[3 * j + 2];
                s0[k] = scales[3 * j]; s1[k] = scales[3 * j + 1]; s2_[k] = scales[3 * j + 2];
            }
        }
#pragma unroll
        for (int k = 0; k < kSub; ++k) {
            keep[k] = false;
            {
                const float mx = V[0] * p0[k] + V[1] * p1[k] + V[2] * p2[k] + V[3];
                const float my = V[4] * p0[k] + V[5] * p1[k] + V[6] * p2[k] + V[7];
                const float z = V[8] * p0[k] + V[9] * p1[k] + V[10] * p2[k] + V[11];
                if (live[k] && !(z < P.near_plane || z > P.far_plane)) {   // the exact projection's own depth cull
                    float sm = s0[k];
                    if constexpr (!PREPARED) {
                        sm = fmaxf(s0[k], fmaxf(s1[k], s2_[k]));
                        if (P.scales_are_log) sm = __expf(sm);
                    }
                    const float rz = __builtin_amdgcn_rcpf(z);
                    const float u = fminf(P.lim_x_pos, fmaxf(-P.lim_x_neg, mx * rz)), v = fminf(P.lim_y_pos, fmaxf(-P.lim_y_neg, my * rz));
                    const float s2 = lam * sm * sm;
                    const float fxz = P.fx * rz, fyz = P.fy * rz;
                    const float rx = 3.33f * __builtin_amdgcn_sqrtf(fxz * fxz * (1.0f + u * u) * s2 + P.eps2d) * 1.001f + 2.0f;
                    const float ry = 3.33f * __builtin_amdgcn_sqrtf(fyz * fyz * (1.0f + v * v) * s2 + P.eps2d) * 1.001f + 2.0f;
                    const float xs = P.fx * mx * rz + P.cx, ys = P.fy * my * rz + P.cy;
                    // (negated comparisons: a NaN anywhere keeps the Gaussian for the exact path to judge)
                    keep[k] = !(ys + ry < y_lo || ys - ry > y_hi || xs + rx < 0.f || xs - rx > P.W);
                }
            }
            bal[k] = __ballot(keep[k]);
        }
        uint32_t *sw = s_w[round & 1];
        ++round;
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < kSub; ++k) sw[k * 16 + w] = (uint32_t)__popcll(bal[k]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSub; ++k) {   // list order = Gaussian order: sub-step by sub-step, wave by wave
            // the sixteen waves' counts of the sub-step: one per lane, a prefix across the lanes, this wave's offset and the
            // total out of it by readlane (every lane summing all sixteen words itself was ~60 instructions a sub-step --
            // more than the test: the pass is bound by what it issues between its round trips, not by its bytes)
            uint32_t incl = lane < 16 ? sw[k * 16 + lane] : 0u;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= d) incl += o;
            }
            const int wu = __builtin_amdgcn_readfirstlane(w);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
            const uint32_t before = written + (wu > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)incl, wu - 1) : 0u);
            if (keep[k])
                seg[before + __builtin_amdgcn_mbcnt_hi((unsigned)(bal[k] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal[k], 0u))] =
                    (int32_t)(base + (int64_t)k * kHistThreads + threadIdx.x);
            written += total;
        }
    }
    if (threadIdx.x == 0) seg_count[blockIdx.x] = (int32_t)written;
    MS_BIN_STAMP(4, 2);
#ifdef MS_DIAG
    if (g_diag_bin && threadIdx.x == 0 && blockIdx.x < 1024) g_diag_bin[(4 * 1024 + blockIdx.x) * 8 + 3] = written;
#endif
}

// Fused projection + tile counting (the first two kernels of a frame in one): every lane projects
// its Gaussian (project_device.hpp), stores the projected record, and counts the tiles of its box
// in the workgroup's LDS histogram.  Same chunking as k_isect_hist / k_isect_scatter.
// LEAN (ms_render_fwd's plain 3-channel forward frames): nobody reads the projected arrays of such a frame -- the
// rasteriser stages from the ready-made records, the scatter kernel wants the tile box and the depth -- so the
// kernel stores one 16-byte LeanRec per position instead of means2d / conics / depths / radii (32 bytes).
template <bool PACK, int LEAN, bool CUT = false>   // LEAN: 0 = the projected arrays, 1 = LeanRec + reach mask, 2 = Lean12 (plain bins); CUT: a depth-cut frame (LEAN == 2)
__global__ __launch_bounds__(kHistThreads, MS_PH_WAVES) void k_project_hist(
    int64_t N, const float *__restrict__ means3d, const float *__restrict__ scales,
    const float *__restrict__ quats, const float *__restrict__ opacities, const float *__restrict__ viewmat,
    ms::ProjParams P, Grid g, int64_t chunk, float *__restrict__ means2d, float *__restrict__ conics,
    float *__restrict__ depths, int32_t *__restrict__ radii, uint32_t *__restrict__ hist,
    uint32_t *__restrict__ wg_on_grid, unsigned long long *__restrict__ masks,
    const void *__restrict__ colors, int color_f16, float4 *__restrict__ rec, Candidates cand,
    LeanRec *__restrict__ lean, uint32_t *__restrict__ wg_depth, const uint32_t *__restrict__ tau,
    uint32_t *__restrict__ wg_far, uint32_t *__restrict__ has_far, uint32_t cut_stamp, LeanRec *__restrict__ near_recs,
    int keep_arrays, int64_t near_cap, unsigned int big_q_off, unsigned int big_q_cap,
    // Round 6 (claimed rows): tile_total != null -> the workgroup does not leave its counts as a histogram row for a prefix
    // kernel; it CLAIMS its stretch of every tile's segment with one returning atomic per tile it holds pairs of, from the
    // counters of its XCD (eight arrays of T_local: row[t] = atomicAdd(tile_total[xcd][t], count)).  Which workgroup comes
    // first within an XCD's stretch is irrelevant -- every list is sorted by (depth bits, index) keys afterwards -- and the
    // XCDs keep contiguous stretches of every segment (chunk_of_block); a tile's count is the sum of its eight counters, the
    // stretch of XCD x starts behind those before it (the scatter kernel's cursors: tile_prefix_lds).  k_tile_scan_wg -- 4 MB
    // read, 4 MB written and a kernel boundary: 7.6 us of config 3's frame -- is not launched.  Workgroup 0 zeroes the
    // counters first and publishes the frame's stamp in *zero_flag; a workgroup reads the flag before its first claim (by
    // then ~20 us old): nothing is carried from frame to frame.
    uint32_t *__restrict__ tile_total, unsigned long long *__restrict__ zero_flag, unsigned long long zero_stamp) {
    extern __shared__ uint32_t s_cnt[];  // T_local tile counters + the on-grid counter (+ a lean frame's depth range) [+ T_local cut-offs + T_local flag bytes] [+ the queue of big boxes]
    const int T_local = (g.row_end - g.row_begin) * g.tw;
    unsigned int &s_on_grid = s_cnt[T_local];
    __shared__ unsigned int s_big_n;
    if (threadIdx.x == 0) s_big_n = 0;   // (published by the barrier below)
    const BigQ bigq{reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_cnt) + big_q_off), &s_big_n, big_q_cap};
    MS_BIN_STAMP(0, 0);
    // DEPTH CUT (tau != null; LEAN == 2 only): a pair whose depth bits exceed its tile's cut-off -- where the PREVIOUS
    // frame's sorted front of that tile ended, with a margin -- is FAR: it is not counted into the tile's list, the tile
    // is marked (has_far[tile] = the frame's stamp), and the scatter kernel never sees it: that kernel's time follows the
    // number of RECORDS it walks, not of pairs (its scattered stores and LDS atomics cost per wave instruction, however
    // few lanes are live), so the Gaussians that keep a pair at all leave it a record of their own -- box, depth bits,
    // the mask of their NEAR tiles, their index -- compacted per workgroup (near_recs; wg_far[kMaxG + g] of them).
    // Should a pixel outlive a marked tile's list, the clean-up launch regenerates the tile's far pairs from the box
    // records every Gaussian still writes (rasterize.hip, k_far_regen).
    __shared__ unsigned int s_near_n;
    __shared__ unsigned int s_next_slice;   // (the workgroup's slices handed to its waves as they come free: below)
    if (threadIdx.x == 0) { s_near_n = 0; s_next_slice = 0; }
    uint32_t *s_tau = s_cnt + T_local + 4;
    unsigned char *s_farflag = reinterpret_cast<unsigned char *>(s_tau + T_local);
    uint32_t far_pairs = 0;
    static_assert(!CUT || LEAN == 2, "depth-cut frames keep 12-byte box records");
    if constexpr (CUT)
        for (int t = threadIdx.x; t < T_local; t += kHistThreads) { s_tau[t] = tau[g.row_begin * g.tw + t]; s_farflag[t] = 0; }
    for (int t = threadIdx.x; t < T_local; t += kHistThreads) s_cnt[t] = 0;
    if (threadIdx.x == 0) { s_on_grid = 0; s_cnt[T_local + 1] = 0xffffffffu; s_cnt[T_local + 2] = 0u; s_cnt[T_local + 3] = 0u; }
    // lean frames: the range of the depth bits of everything this workgroup's Gaussians can emit, for k_tile_front's
    // buckets (the scatter kernel -- a chain of round trips -- otherwise ends with this reduction and its barriers)
    uint32_t dmin = 0xffffffffu, dmax = 0u;
    if (tile_total && blockIdx.x == 0) {
        // write-through stores (the XCDs' L2s are not coherent with each other), drained by the storing wave itself before the
        // barrier, then the flag: the hand-off pattern of k_tile_scan_wg
        for (int t = threadIdx.x; t < kClaimGroups * T_local; t += kHistThreads) __hip_atomic_store(&tile_total[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tile_total && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(zero_flag, zero_stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ uint32_t s_pref[kMaxG + 1];
    CandMap map{s_pref};
    const int wg = chunk_of_block(blockIdx.x, gridDim.x);   // this workgroup's histogram row (and share of the positions: Deal)
    (void)chunk;
    int64_t i1 = N;
    if (cand.ids) {   // positions of the band's candidate list
        map.build(cand);
        i1 = (int64_t)s_pref[cand.n_segs];
    }
    const Deal deal(i1, wg, (int)gridDim.x);
    const uint32_t lane_u = threadIdx.x & 63u;
    // what a (box, tile) pair does to the histogram: plainly, and on a depth-cut frame (a pair behind its tile's cut-off is
    // counted as far, its tile marked) -- for the boxes a wave walks on the spot and for those the workgroup queues
    auto count_plain = [&](int t, int64_t, int) __attribute__((always_inline)) { atomicAdd(&s_cnt[t], 1u); };
    auto count_cut = [&](int t, int64_t, int, uint32_t db) __attribute__((always_inline)) {
        if (db <= s_tau[t]) { atomicAdd(&s_cnt[t], 1u); dmin = min(dmin, db); dmax = max(dmax, db); }
        else { ++far_pairs; s_farflag[t] = 1; }
    };
    // WHICH wave of the workgroup takes which of its slices is decided as the waves come free (an LDS ticket per slice): the
    // slices' costs differ -- a few large boxes, a slice of culled Gaussians -- and with two fixed slices per wave the
    // workgroup waited 4.7 us of its 20 at the loop's barrier for its slowest wave (profiles/r05_bin_phases_given.txt).
    // The workgroup's SET of slices is Deal's, so its histogram row -- and what the scatter kernel does with it -- is the same.
    constexpr bool kTickets = kDeal == 64 && MS_PH_TICKETS != 0;
    const int n_slices = deal.iters * (kHistThreads / kDeal);
    for (int it = 0;; ++it) {
        int64_t base;
        if constexpr (kTickets) {
            int k = 0;
            if (lane_u == 0) k = (int)atomicAdd(&s_next_slice, 1u);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= n_slices) break;
            base = ((int64_t)k * (int)gridDim.x + wg) * kDeal;   // slice k of this workgroup: Deal's (trip k / 16, part k % 16)
        } else {
            if (it >= deal.iters) break;
            base = deal.base(it);   // (wave-uniform: this wave's slice of 64 positions)
        }
        const int64_t j = base + lane_u;
        int x0 = 0, x1 = 0, y0 = 0, y1 = 0, n = 0, edges = 0, gi = 0;
        bool on_grid = false;
        unsigned long long mask = ~0ull;
        uint32_t dbits_of_step = 0u;   // (depth cut: this lane's depth bits)
        // depth-cut frame: the rasteriser's record of a Gaussian is only written once it is known to keep a pair (48 of
        // the ~120 bytes this kernel moves per Gaussian; the clean-up launch writes the record of a dropped Gaussian it
        // brings back: rasterize.hip, k_far_regen) -- what it is made of waits in these
        bool rec_pending = false;
        int64_t rec_b0 = 0;        // (a band's gathered candidates: the whole arrays and the Gaussian's index)
        uint32_t rec_src = 0u;
        float rec_m0 = 0.f, rec_m1 = 0.f, rec_c0 = 0.f, rec_c1 = 0.f, rec_c2 = 0.f;
        auto write_record = [&](int64_t b0, uint32_t rec_src, float m0, float m1, float c0, float c1, float c2) __attribute__((always_inline)) {
            // the rasteriser's staged record, ready made (ms::RasterRecord, ms_common.hpp): three 16-byte
            // words per visible Gaussian instead of seven scattered 4-12-byte gathers + arithmetic per
            // (tile, Gaussian) pair.  Same expressions as the rasteriser's own staging: same bits.
            const float *opac_b = opacities ? opacities + b0 : nullptr;
            float col[3];
            if (color_f16) {
                const __half *c = reinterpret_cast<const __half *>(colors) + 3 * b0;
                col[0] = __half2float(c[3 * rec_src]); col[1] = __half2float(c[3 * rec_src + 1]); col[2] = __half2float(c[3 * rec_src + 2]);
            } else {
                const float *c = reinterpret_cast<const float *>(colors) + 3 * b0;
                const ms::F3 c3 = ms::ld_f32x3(c, rec_src);
                col[0] = c3.x; col[1] = c3.y; col[2] = c3.z;
            }
            const ms::RasterRecord r = ms::make_raster_record(m0, m1, c0, c1, c2, ms::ld_f32(opac_b, rec_src, 1, 0), col[0], col[1], col[2]);
            // (stores likewise: the step's slice of the output + a 32-bit byte offset)
            char *rb = reinterpret_cast<char *>(rec + 3 * base) + 48u * lane_u;
            *reinterpret_cast<float4 *>(rb) = r.a;
            *reinterpret_cast<float4 *>(rb + 16) = r.b;
            *reinterpret_cast<float4 *>(rb + 32) = r.c;
        };
        if (j < i1) {
            // A band's candidates: the INPUT arrays are gathered through the candidate list; everything this
            // kernel writes -- and every index the later stages see -- is the POSITION j in that list, so the
            // projected arrays stay dense and their stores coalesced (scattered 4-48-byte stores by Gaussian
            // index cost 3.5x the projection itself).  Positions grow with the Gaussian index, so the
            // (depth bits, id) order of the sorted lists is the same either way.
            // (32-bit indices against uniform base pointers -- the step's own slice of the inputs, or the whole arrays
            // for a candidate's gather: one address register per load instead of 64-bit arithmetic per lane)
            const bool gather = cand.ids != nullptr;
            const uint32_t src = gather ? (uint32_t)map.gaussian(cand, j) : lane_u;
            const int64_t b0 = gather ? 0 : base;
            const float *opac_b = opacities ? opacities + b0 : nullptr;
            const int64_t i = j;
            gi = (int)i;
            // (a depth-cut frame -- a scene of millions of Gaussians -- fetches everything with the mean: project_device.hpp)
            const ms::ProjOut o = ms::project_one<uint32_t, CUT>(src, means3d + 3 * b0, scales + 3 * b0, quats + 4 * b0, opac_b, viewmat, P);
            // (a lean frame whose caller reads the projected arrays all the same -- a differentiable frame: its backward
            // and its intermediates -- writes both: the scatter kernel then still walks 12-byte records)
            bool arrays = LEAN == 0;
            if constexpr (LEAN != 0) arrays = keep_arrays != 0;
            if (arrays) {
                reinterpret_cast<float2 *>(means2d)[i] = make_float2(o.m0, o.m1);
                conics[3 * i] = o.c0;
                conics[3 * i + 1] = o.c1;
                conics[3 * i + 2] = o.c2;
                depths[i] = o.d;
                reinterpret_cast<int2 *>(radii)[i] = make_int2(o.r0, o.r1);
            }
#ifdef MS_ABL_NOREC   // (measurement builds, profiles/r06_project_hist_isa.md: never the shipped library)
            if (false) {
#else
            if (rec && o.r0 > 0 && o.r1 > 0) {
#endif
                if constexpr (CUT) {
                    rec_pending = true;
                    rec_b0 = b0; rec_src = src;
                    rec_m0 = o.m0; rec_m1 = o.m1; rec_c0 = o.c0; rec_c1 = o.c1; rec_c2 = o.c2;
                } else {
                    write_record(b0, src, o.m0, o.m1, o.c0, o.c1, o.c2);
                }
            }
            if (o.r0 > 0 && o.r1 > 0) {
                on_grid = bin_box<PACK>(make_float2(o.m0, o.m1), make_int2(o.r0, o.r1), g, x0, x1, y0, y1, edges);
                n = (x1 - x0) * (y1 - y0);
#ifdef MS_ABL_NOMASK
                if (false) {
#else
                if (masks) {
#endif
                    if (LEAN == 2 && n > 32) mask = ~0ull;   // (a 12-byte record has room for 32 tiles' worth)
                    else
                    if (PACK && n <= 16 && n > 0)   // per half-tile cell (the 16x16 blocks of a 32-px bin)
                        mask = clip_cells(reach_mask(o.m0, o.m1, o.c0, o.c1, o.c2, ms::ld_f32(opac_b, src, 1, 0), 2 * x0, 2 * x1,
                                                     2 * y0, 2 * y1, g.ts >> 1), edges, x1 - x0, y1 - y0);
                    else
                        mask = reach_mask(o.m0, o.m1, o.c0, o.c1, o.c2, ms::ld_f32(opac_b, src, 1, 0), x0, x1, y0, y1, g.ts);
                    if constexpr (LEAN != 2)
                        *reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(masks + base) + 8u * lane_u) = mask;
                }
            }
            dbits_of_step = __float_as_uint(o.d);
            if constexpr (LEAN != 0) {
                // (a depth-cut frame: only Gaussians that keep a pair stretch the range the sort kernel's buckets cover)
                if (n > 0 && !CUT) { dmin = min(dmin, __float_as_uint(o.d)); dmax = max(dmax, __float_as_uint(o.d)); }
            }
            // (a depth-cut frame's scatter kernel walks the compacted records of the Gaussians that keep a pair, and its
            // clean-up launches project what they bring back themselves: nobody would read a box record -- 12 bytes a
            // Gaussian, a sixth of what this kernel moves, not written)
            if constexpr (LEAN == 2 && !CUT) {
                uint32_t *q = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(reinterpret_cast<Lean12 *>(lean) + base) + 12u * lane_u);
                const uint32_t box = n > 0 ? ((uint32_t)x0 | ((uint32_t)y0 << 8) | ((uint32_t)(x1 - x0) << 16) | ((uint32_t)(y1 - y0) << 24)) : 0u;
                q[0] = box; q[1] = __float_as_uint(o.d); q[2] = (uint32_t)mask;
            }
            if constexpr (LEAN == 1)
                *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(lean + base) + 16u * lane_u) = make_uint4((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)x1 | ((uint32_t)y1 << 16),
                                                                __float_as_uint(o.d), (uint32_t)n | ((uint32_t)edges << 28));
        }
        count_on_grid(on_grid, &s_on_grid);
        if constexpr (CUT) {
            // (only pairs that are kept stretch the depth range the sort kernel's buckets cover)
            uint32_t near32 = 0u;
            if (n > 0 && n <= kCoopThreshold) {   // this lane's own box: the reached tiles, bit by bit (as walk_boxes)
                const int w = x1 - x0;
                const float inv_w = __builtin_amdgcn_rcpf((float)w);
                unsigned int m = (unsigned int)mask & (n >= 32 ? 0xffffffffu : ((1u << n) - 1u));
                while (m) {
                    const int k = __ffs((int)m) - 1;
                    m &= m - 1;
                    const int r = (int)(((float)k + 0.5f) * inv_w);
                    const int t = (y0 + r - g.row_begin) * g.tw + x0 + (k - r * w);
                    if (dbits_of_step <= s_tau[t]) {
                        atomicAdd(&s_cnt[t], 1u);
                        near32 |= 1u << k;
                    } else {
                        ++far_pairs;
                        s_farflag[t] = 1;
                    }
                }
                if (near32) { dmin = min(dmin, dbits_of_step); dmax = max(dmax, dbits_of_step); }
            }
            // larger boxes: the whole wave (db is the OWNER's depth there); the scatter kernel applies the cut-offs itself
            walk_boxes<PACK>(gi, x0, x1, y0, y1, n > kCoopThreshold ? n : 0, edges, g, mask, count_cut, dbits_of_step, &bigq);
            // (boxes the whole wave walked: kept whatever the cut-offs say -- their owner does not know)
            const bool keep = near32 != 0u || n > kCoopThreshold;
            if (rec_pending && keep) write_record(rec_b0, rec_src, rec_m0, rec_m1, rec_c0, rec_c1, rec_c2);
            const unsigned long long kb = __ballot(keep);
            if (kb) {
                const int lane = threadIdx.x & 63;
                unsigned int wbase = 0;
                if (lane == 0) wbase = atomicAdd(&s_near_n, (unsigned int)__popcll(kb));
                wbase = (unsigned int)__builtin_amdgcn_readfirstlane((int)wbase);
                if (keep) {
                    const unsigned int pos = wbase + (unsigned int)__popcll(kb & ((1ull << lane) - 1ull));
                    const uint32_t box = (uint32_t)x0 | ((uint32_t)y0 << 8) | ((uint32_t)(x1 - x0) << 16) | ((uint32_t)(y1 - y0) << 24);
                    reinterpret_cast<uint4 *>(near_recs)[(int64_t)wg * near_cap + pos] = make_uint4(box, (uint32_t)gi, dbits_of_step, near32);
                }
            }
        } else {
#ifndef MS_ABL_NOWALK
            walk_boxes<PACK>(gi, x0, x1, y0, y1, n, edges, g, mask, count_plain, &bigq);
#endif
        }
    }
    // (the boxes of more than kCoopThreshold tiles that the waves queued: all sixteen waves take them in turn)
    if constexpr (CUT) drain_big_boxes<PACK, true>(bigq, g, count_cut);
    else drain_big_boxes<PACK, false>(bigq, g, count_plain);
    if constexpr (CUT) {
        {   // this workgroup's far pairs (the frame's size record counts them: the buffer keeps room for them)
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) far_pairs += (uint32_t)__shfl_xor((int)far_pairs, d);
            if ((threadIdx.x & 63) == 0 && far_pairs) atomicAdd(&s_cnt[T_local + 3], far_pairs);
        }
    }
    if constexpr (LEAN != 0) {
        if (wg_depth) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, d));
                dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, d));
            }
            if ((threadIdx.x & 63) == 0) { atomicMin(&s_cnt[T_local + 1], dmin); atomicMax(&s_cnt[T_local + 2], dmax); }
        }
    }
    MS_BIN_STAMP(0, 1);
    if (tile_total && threadIdx.x == 0) {
        // (workgroup 0 is dispatched first and zeroes before anything else: the flag is there long before a workgroup gets here)
        while (__hip_atomic_load(zero_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != zero_stamp) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    MS_BIN_STAMP(0, 2);
    uint32_t *row = hist + (size_t)wg * T_local;
    if (tile_total) {
        for (int t = threadIdx.x; t < T_local; t += kHistThreads) {
            const uint32_t c = s_cnt[t];
            // (the counters of this workgroup's XCD -- blockIdx & 7, the eighth of the rows chunk_of_block gives it: the XCDs keep
            // contiguous stretches of every tile's segment, which the scatter kernel's write merging lives on)
            row[t] = c ? __hip_atomic_fetch_add(&tile_total[(size_t)((blockIdx.x & 7) >> kClaimShift) * T_local + t], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
    } else
    for (int t = threadIdx.x; t < T_local; t += kHistThreads) row[t] = s_cnt[t];
    if (threadIdx.x == 0) wg_on_grid[wg] = s_on_grid;
    if (threadIdx.x == 0 && blockIdx.x == 0) wg_on_grid[kMaxG] = 0;   // k_tile_scan_wg's arrival ticket
    if constexpr (LEAN != 0) {
        if (wg_depth && threadIdx.x == 0) { wg_depth[2 * blockIdx.x] = s_cnt[T_local + 1]; wg_depth[2 * blockIdx.x + 1] = s_cnt[T_local + 2]; }
    }
    if constexpr (CUT) {
        {
            if (threadIdx.x == 0) { wg_far[wg] = s_cnt[T_local + 3]; wg_far[kMaxG + wg] = s_near_n; }
            for (int t = threadIdx.x; t < T_local; t += kHistThreads)   // the tiles this workgroup dropped pairs of
                if (s_farflag[t]) has_far[g.row_begin * g.tw + t] = cut_stamp;   // (every writer stores the same value)
        }
    }
    MS_BIN_STAMP(0, 3);
}

__global__ __launch_bounds__(kHistThreads) void k_isect_hist(
    int64_t N, const float *__restrict__ means2d, const int32_t *__restrict__ radii, Grid g,
    int64_t chunk, uint32_t *__restrict__ hist, int32_t *__restrict__ tiles_per_gauss,
    uint32_t *__restrict__ wg_on_grid) {
    extern __shared__ uint32_t s_cnt[];  // T_local tile counters + the on-grid counter
    const int T_local = (g.row_end - g.row_begin) * g.tw;
    unsigned int &s_on_grid = s_cnt[T_local];
    for (int t = threadIdx.x; t < T_local; t += kHistThreads) s_cnt[t] = 0;
    if (threadIdx.x == 0) s_on_grid = 0;
    __syncthreads();
    const int wg = chunk_of_block(blockIdx.x, gridDim.x);
    (void)chunk;
    for_each_isect<false>(0, N, means2d, radii, nullptr, g, tiles_per_gauss, &s_on_grid, Candidates{nullptr, nullptr, 0, 0},
                   (int)gridDim.x, [&](int t, int64_t, int) { atomicAdd(&s_cnt[t], 1u); });
    __syncthreads();
    uint32_t *row = hist + (size_t)wg * T_local;
    for (int t = threadIdx.x; t < T_local; t += kHistThreads) row[t] = s_cnt[t];
    if (threadIdx.x == 0) wg_on_grid[wg] = s_on_grid;
    if (threadIdx.x == 0 && blockIdx.x == 0) wg_on_grid[kMaxG] = 0;   // k_tile_scan_wg's arrival ticket
}

// One workgroup: exclusive scan of the per-tile counts over the FULL grid (tiles outside
// the band count 0), tile_ranges, totals and the work lists of over-sized tiles.  Each wave owns
// a contiguous run of tiles and walks it 64 tiles at a time (coalesced loads / int2 stores,
// wave-level scans), so the kernel is three dependent memory round trips long.
// The clean-up counts a frame leaves for the next one (redo_count[0..1]) live in the isect workspace at an offset that follows the
// PLAN's grid -- the tile grid, or a split frame's 32-px bins -- so a frame whose predecessor on this workspace planned another
// grid (lazy -> full sorts on 16-px tiles: split -> not split) would read a word of some other array as "bins redone".  The total
// pass signs what it resets (redo_count[2]) and only reports counts that carry its own grid's signature.
__device__ __forceinline__ int redo_signature(const Grid &g) { return (int)(0x80000000u | ((unsigned)g.tw & 0x7fffu) | (((unsigned)g.th & 0xffffu) << 15)); }

struct ScanTotalArgs {
    Grid g;
    const uint32_t *tile_count;
    int32_t *tile_ranges, *medium_list, *large_list, *xl_list;
    const uint32_t *wg_on_grid;
    int G;
    int32_t *redo_flag, *redo_count;
    int band_only;
    int64_t *info, *info_mirror;
    int32_t *order;
    uint32_t *ticket;   // k_tile_scan_wg: arrival counter of its workgroups (zeroed by the frame's count kernel)
    // depth-cut frame (cut_stamp != 0): far pairs per count workgroup (G of them) for the size record; the cut-offs the
    // scatter kernel holds the boxes of more than 32 tiles against
    const uint32_t *wg_far;
    const uint32_t *tau;
    uint32_t cut_stamp;
    const LeanRec *near_recs;   // the records that keep a pair, compacted per count workgroup (wg_far[kMaxG + g] of them from g * near_cap on)
    int64_t near_cap;
    uint32_t *far_zero;   // 2 T words the clean-up launches count the regenerated pairs in: zeroed by the total pass
    // claimed rows (round 6, MOJOSPLAT_CLAIMED_ROWS=1): the count kernel's workgroups claimed their stretches of every tile's
    // segment from eight counter arrays, one per XCD (xtot[x * T_local + t]): a tile's count is their sum (the total pass
    // writes it to tile_count), and the stretch of XCD x starts behind those of the XCDs before it
    const uint32_t *xtot;
};

// a tile's count in the deferred passes: the prefix kernel's, or the sum of the eight XCDs' claims
__device__ __forceinline__ uint32_t tile_count_of(const ScanTotalArgs &A, int t, int T_local) {
    if (!A.xtot) return A.tile_count[t];
    uint32_t c = 0;
#pragma unroll
    for (int x = 0; x < kClaimGroups; ++x) c += A.xtot[(size_t)x * T_local + t];
    return c;
}

// HANDOFF: the counts were written by other workgroups of the SAME launch (k_tile_scan_wg's last workgroup runs
// this): agent-scope loads; a launch of its own reads them plainly.
template <bool HANDOFF>
__device__ __forceinline__ void tile_scan_total(const ScanTotalArgs &A) {
    const Grid &g = A.g;
    const uint32_t *__restrict__ tile_count = A.tile_count;
    int32_t *__restrict__ tile_ranges = A.tile_ranges;
    int32_t *__restrict__ medium_list = A.medium_list, *__restrict__ large_list = A.large_list,
            *__restrict__ xl_list = A.xl_list;
    const uint32_t *__restrict__ wg_on_grid = A.wg_on_grid;
    const int G = A.G, band_only = A.band_only;
    int32_t *__restrict__ redo_flag = A.redo_flag, *__restrict__ redo_count = A.redo_count;
    int64_t *__restrict__ info = A.info, *__restrict__ info_mirror = A.info_mirror;
    int32_t *__restrict__ order = A.order;
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned int s_nmedium, s_nlarge, s_nxl, s_max, s_on_grid;
    // heaviest-first order of the band's tiles for the rasteriser's launch (order[0 .. band tiles)): counting
    // sort over 128 buckets of the list length (4 per octave); ties in whatever order the atomics fall
    __shared__ unsigned int s_bkt[128], s_base[128];
    auto bucket_of = [](unsigned int c) -> int {
        if (c == 0) return 0;
        const int l = 31 - __clz((int)c);
        const int frac = l >= 2 ? (int)((c >> (l - 2)) & 3u) : (int)((c << (2 - l)) & 3u);
        return 1 + 4 * l + frac;
    };
    if (threadIdx.x < 128) s_bkt[threadIdx.x] = 0;
    const int band0 = g.row_begin * g.tw, band1 = g.row_end * g.tw;
    // band_only: tile_ranges is written for the band's tiles alone (a caller whose later stages all
    // stay inside the band -- ms_render_fwd on a multi-GPU rank -- saves the walk over the rest)
    const int t_lo = band_only ? band0 : 0, t_hi = band_only ? band1 : g.tw * g.th;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int per_wave = ((t_hi - t_lo + 15) / 16 + 63) & ~63;   // tiles per wave, a multiple of 64
    const int w0 = t_lo + w * per_wave, w1 = min(t_hi, w0 + per_wave);
    if (threadIdx.x == 0) { s_nmedium = 0; s_nlarge = 0; s_nxl = 0; s_max = 0; s_on_grid = 0; }
    // (loads whose values are only needed at the end go out first: the pass is a chain of dependent round trips)
    const int prev_redo = (threadIdx.x == 0 && redo_count[2] == redo_signature(g)) ? *redo_count : 0;
    unsigned int on_grid_part = (int)threadIdx.x < G ? wg_on_grid[threadIdx.x] : 0u;   // G <= kMaxG <= blockDim
    // (s_bkt is zeroed above and first added to after the __syncthreads() between the two passes)
    auto count_of = [&](int t) -> unsigned int {
        if (!(t < w1 && t >= band0 && t < band1)) return 0u;
        if constexpr (HANDOFF) return __hip_atomic_load(&tile_count[t - band0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return tile_count[t - band0];
    };
    // pass 1: this wave's total.  The first kPre steps' counts are loaded in one go (independent loads:
    // one memory round trip instead of one per step) and kept in registers for pass 2 -- the kernel is
    // a single workgroup and nothing but latency.  kPre * 16 * 64 = 16 384 tiles are covered that way
    // (1080p: 8 steps per wave); larger grids take the remaining steps with a load each.
    constexpr int kPre = 16;
    unsigned int pre[kPre];
    unsigned long long sum = 0;
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        pre[k] = count_of(w0 + k * 64 + lane);
        sum += pre[k];
    }
    for (int t = w0 + kPre * 64 + lane; t < w1; t += 64) sum += count_of(t);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    if (lane == 0) s_wave[w] = sum;
    __syncthreads();
    unsigned long long run = 0, grand = 0;
#pragma unroll
    for (int ww = 0; ww < 16; ++ww) {
        if (ww < w) run += s_wave[ww];
        grand += s_wave[ww];
    }
    {   // a wave sum, then one LDS atomic per wave
        unsigned int v = on_grid_part;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += (unsigned int)__shfl_xor((int)v, d);
        if (lane == 0 && v) atomicAdd(&s_on_grid, v);
    }
    // pass 2: offsets, ranges, classes
    unsigned int lmax = 0;
    auto step = [&](int tb, unsigned int c) {
        const int t = tb + lane;
        unsigned int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int o = (unsigned int)__shfl_up((int)incl, d);
            if (lane >= d) incl += o;
        }
        if (t < w1) {
            // offsets saturate at INT32_MAX; the host rejects M > INT32_MAX before emitting
            const unsigned long long b = run + (incl - c), e = run + incl;
            reinterpret_cast<int2 *>(tile_ranges)[t] =
                make_int2((int32_t)min(b, 0x7fffffffull), (int32_t)min(e, 0x7fffffffull));
            redo_flag[t] = 0;
            lmax = max(lmax, c);
            if (c > (unsigned)kLargeCapDecl) xl_list[atomicAdd(&s_nxl, 1u)] = t;
            else if (c > (unsigned)kMediumCapDecl) large_list[atomicAdd(&s_nlarge, 1u)] = t;
            else if (c > (unsigned)kSmallCapDecl) medium_list[atomicAdd(&s_nmedium, 1u)] = t;
            if (order && t >= band0 && t < band1) atomicAdd(&s_bkt[bucket_of(c)], 1u);
        }
        run += (unsigned long long)__shfl((int)incl, 63);  // step total (< 2^31 per 64 tiles by int32 M limit)
    };
#pragma unroll
    for (int k = 0; k < kPre; ++k)
        if (w0 + k * 64 < w1) step(w0 + k * 64, pre[k]);   // wave-uniform
    for (int tb = w0 + kPre * 64; tb < w1; tb += 64) step(tb, count_of(tb + lane));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) lmax = max(lmax, (unsigned int)__shfl_xor((int)lmax, d));
    if (lane == 0) atomicMax(&s_max, lmax);
    __syncthreads();
    if (order) {
        if (w == 0) {   // s_base[b] = tiles in heavier buckets (descending order): suffix sums over 128 buckets
            const unsigned int c0 = s_bkt[127 - 2 * lane], c1 = s_bkt[126 - 2 * lane];   // lane 0 owns the heaviest two
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
        // where the LIGHT lists (bucket <= kLightBucket: at most 255 entries) begin in the order: k_tile_front sorts
        // those eight to a workgroup
        if (threadIdx.x == 0 && A.wg_on_grid) const_cast<uint32_t *>(A.wg_on_grid)[kMaxG + 1] = s_base[kLightBucket];
        __syncthreads();
        auto place = [&](int t, unsigned int c) {
            if (t < w1 && t >= band0 && t < band1) order[atomicAdd(&s_base[bucket_of(c)], 1u)] = t;
        };
#pragma unroll
        for (int k = 0; k < kPre; ++k)
            if (w0 + k * 64 < w1) place(w0 + k * 64 + lane, pre[k]);
        for (int tb = w0 + kPre * 64; tb < w1; tb += 64) place(tb + lane, count_of(tb + lane));
    }
    if (threadIdx.x == 0) {
        info[0] = (int64_t)grand;
        info[1] = (int64_t)s_max;
        info[2] = (int64_t)s_nmedium;
        info[3] = (int64_t)s_nlarge;
        info[4] = (int64_t)s_nxl;
        // tiles the PREVIOUS frame on this workspace had to redo (lazy sorting): low 32 bits = because their sorted
        // front was too short, high 32 bits = because their depth cut-off was (redo_count[1] counts those)
        {
            const int prev_cut = min(max(redo_count[1], 0), max(prev_redo, 0));
            // (bit 62: the previous frame's clean-up launches regenerated a pair they had no room for / more than were counted)
            info[5] = (int64_t)(prev_redo - prev_cut) + ((int64_t)(prev_cut & 0x3fffffff) << 32) + (prev_cut > 0 && info[10] != 0 ? (1ll << 62) : 0);
        }
        redo_count[0] = 0;
        redo_count[1] = 0;
        redo_count[2] = redo_signature(g);
        info[6] = (int64_t)s_on_grid;  // Gaussians whose tile box touches the FULL grid (band-independent)
        info[7] = 0;
        if (info_mirror) {
            // a second copy straight into the caller's pinned host memory (zero-copy store): saves the
            // copy kernel and the two pipeline bubbles around it; visible to the host once an event
            // recorded after this kernel has completed.  Word 7 belongs to the host side.
#pragma unroll
            for (int k = 0; k < 7; ++k) info_mirror[k] = info[k];
            __threadfence_system();   // (measured free: 12.9 us with and without)
        }
    }
}

// hist[g][t] -> exclusive prefix over g (in place); tile_count[t] = sum over g.
// kScanTiles tiles per workgroup (round 2, second session: 64 tiles per workgroup left a 1080p frame's 2 040 bins to 32
// workgroups, i.e. 32 CUs moving 8 MB): thread = (tile, slice of G / 64 partial rows); the slices of a tile are
// combined by two shuffles inside the wave and one LDS step across the 16 waves.
// The workgroup that arrives LAST (a ticket zeroed by the frame's count kernel; release / acquire fences at
// agent scope around it) goes on to run the total pass (tile_scan_total) in the same launch: one kernel
// boundary and one launch latency fewer on the frame's critical path.
// (kScanTiles = 16 for grids up to 4 096 tiles; larger ones -- 8 160 tiles of 16 px at 1080p: 17 MB of partial
// rows -- take 64 tiles per workgroup: 64-byte row pieces waste half of every 128-byte line once the rows no
// longer sit in the caches, measured 29 us against 17.)
template <int kScanTiles>
__global__ __launch_bounds__(1024) void k_tile_scan_wg(int G, int T_local,
                                                       uint32_t *__restrict__ hist,
                                                       uint32_t *__restrict__ tile_count, ScanTotalArgs A) {
    constexpr int kScanSlices = 1024 / kScanTiles;
    __shared__ uint32_t s_part[16][kScanTiles];
    __shared__ int s_last;
    MS_BIN_STAMP(1, 0);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int tl = threadIdx.x & (kScanTiles - 1), sl = threadIdx.x / kScanTiles;
    const int t = blockIdx.x * kScanTiles + tl;
    constexpr int kRows = kMaxG / kScanSlices;   // partial rows per slice (8)
    const int per = (G + kScanSlices - 1) / kScanSlices;
    const int g0 = sl * per, g1 = min(G, g0 + per);
    uint32_t v[kRows];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const int gg = g0 + k;
        v[k] = (k < per && gg < g1 && t < T_local) ? hist[(size_t)gg * T_local + t] : 0u;
        sum += v[k];
    }
    // the 64 / kScanTiles slices of a tile inside this wave sit kScanTiles lanes apart: inclusive prefix over them
    uint32_t incl = sum;
#pragma unroll
    for (int d = kScanTiles; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += o;
    }
    if (lane >= 64 - kScanTiles) s_part[w][tl] = incl;   // the wave's total for tile tl
    MS_BIN_STAMP(1, 1);
    __syncthreads();
    MS_BIN_STAMP(1, 2);
    uint32_t run = incl - sum, total = 0;
#pragma unroll
    for (int ww = 0; ww < 16; ++ww) {
        const uint32_t p = s_part[ww][tl];
        if (ww < w) run += p;
        total += p;
    }
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const int gg = g0 + k;
        if (k < per && gg < g1 && t < T_local) {
            hist[(size_t)gg * T_local + t] = run;
            run += v[k];
        }
    }
    // The hand-off to the last workgroup is the tile counts alone (the prefixes are for the next KERNEL): agent-scope
    // stores (write-through: the XCDs' L2s are not coherent with each other), DRAINED by the storing wave itself --
    // an explicit s_waitcnt vmcnt(0): a workgroup-scope release fence only waits on lgkmcnt in this mode, so without
    // it the ticket could become visible on another XCD before the counts have landed -- then the barrier, then the
    // ticket; the total pass reads the counts with agent-scope loads.  A full agent-scope fence here would write back
    // every dirty line of the 4 MB of prefixes first: measured 61 us.
    if (!A.ticket) {   // the total pass is a launch of its own
        if (sl == 0 && t < T_local) tile_count[t] = total;
        MS_BIN_STAMP(1, 3);
        return;
    }
    if (sl == 0 && t < T_local) __hip_atomic_store(&tile_count[t], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have been acknowledged
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    tile_scan_total<true>(A);
}

// The total pass alone (an empty band has no per-tile prefix to take)
__global__ __launch_bounds__(1024) void k_tile_scan_total(ScanTotalArgs A) { tile_scan_total<false>(A); }

// Exclusive prefix over the T tile counts in LDS: s[t] = sum of count[0 .. t), every thread of the (1024-thread)
// workgroup; -> the grand total (64-bit; offsets saturate at INT32_MAX: the host rejects M > INT32_MAX before it
// trusts any of them).  Three barriers whatever T is: counts -> LDS (coalesced), each thread sums a run of
// ceil(T / 1024) consecutive entries, wave scans + the 16 wave totals, each thread writes its run back.
// add: an optional row of T words added to the prefix as it is written back (the scatter kernel's histogram row:
// its loads are issued before the first barrier when a thread's run is at most four entries).
// xtot / x_mine (claimed rows): the counts are the sums of the eight XCDs' claims, and what is added to the prefix is the
// row's claim plus the claims of the XCDs before x_mine.
__device__ __forceinline__ unsigned long long tile_prefix_lds(const uint32_t *__restrict__ count, int T, uint32_t *s,
                                                              const uint32_t *__restrict__ add = nullptr,
                                                              const uint32_t *__restrict__ xtot = nullptr, int x_mine = 0) {
    __shared__ unsigned long long s_wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (T + kHistThreads - 1) / kHistThreads;
    const int j0 = min(T, tid * per), j1 = min(T, j0 + per);
    uint32_t addv[4] = {0u, 0u, 0u, 0u};
    const bool pre = add && per <= 4;
    auto add_of = [&](int j) __attribute__((always_inline)) -> uint32_t {
        uint32_t v = add[j];
        if (xtot)
            for (int x = 0; x < x_mine; ++x) v += xtot[(size_t)x * T + j];
        return v;
    };
    if (pre) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (j0 + k < j1) addv[k] = add_of(j0 + k);
    }
    if (xtot) {
        for (int t = tid; t < T; t += kHistThreads) {
            uint32_t c = 0;
#pragma unroll
            for (int x = 0; x < kClaimGroups; ++x) c += xtot[(size_t)x * T + t];
            s[t] = c;
        }
    } else
    for (int t = tid; t < T; t += kHistThreads) s[t] = count[t];
    __syncthreads();
    unsigned long long sum = 0;
    for (int j = j0; j < j1; ++j) sum += s[j];
    unsigned long long incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wtot[w] = incl;
    __syncthreads();
    unsigned long long run = incl - sum, grand = 0;
#pragma unroll
    for (int ww = 0; ww < 16; ++ww) {
        if (ww < w) run += s_wtot[ww];
        grand += s_wtot[ww];
    }
    if (pre) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (j0 + k < j1) {
                const uint32_t v = s[j0 + k];
                s[j0 + k] = (uint32_t)min(run, 0x7fffffffull) + addv[k];
                run += v;
            }
    } else {
        for (int j = j0; j < j1; ++j) {
            const uint32_t v = s[j];
            s[j] = (uint32_t)min(run, 0x7fffffffull) + (add ? add_of(j) : 0u);
            run += v;
        }
    }
    __syncthreads();
    return grand;
}

// The total pass of the scans, DEFERRED into the scatter launch (sync-free frames of ms_render_fwd): as a pass of
// its own -- one workgroup, a chain of dependent round trips -- it cost 9 of the scan kernel's 12.9 us on the
// frame's critical path.  Here every scatter workgroup takes the exclusive prefix over the tile counts itself
// (tile_prefix_lds: 8 KB of L2-resident counts, three barriers), and two EXTRA workgroups of the same launch
// write what the later kernels and the host want while the others scatter:
//   which == 0   tile_ranges, the zeroed redo flags, the work lists of over-sized tiles, the size record (+ mirror)
//   which == 1   the heaviest-first order of the band's tiles (counting sort over 128 length buckets)
// Same results as tile_scan_total (whose comments apply), bit for bit.
__device__ __forceinline__ void deferred_total(int which, const ScanTotalArgs &A, uint32_t *s, int T_local) {
    const Grid &g = A.g;
    const int band0 = g.row_begin * g.tw, band1 = g.row_end * g.tw;
    const int tid = threadIdx.x, lane = tid & 63;
    if (which == 1) {
        if (!A.order) return;
        __shared__ unsigned int s_bkt[128], s_base[128];
        auto bucket_of = [](unsigned int c) -> int {
            if (c == 0) return 0;
            const int l = 31 - __clz((int)c);
            const int frac = l >= 2 ? (int)((c >> (l - 2)) & 3u) : (int)((c << (2 - l)) & 3u);
            return 1 + 4 * l + frac;
        };
        if (tid < 128) s_bkt[tid] = 0;
        __syncthreads();
        for (int t = tid; t < T_local; t += kHistThreads) atomicAdd(&s_bkt[bucket_of(tile_count_of(A, t, T_local))], 1u);
        __syncthreads();
        if (tid < 64) {   // s_base[b] = tiles in heavier buckets: suffix sums, lane 0 owns the heaviest two
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
        if (tid == 0 && A.wg_on_grid) const_cast<uint32_t *>(A.wg_on_grid)[kMaxG + 1] = s_base[kLightBucket];   // (see tile_scan_total)
        __syncthreads();
        for (int t = tid; t < T_local; t += kHistThreads)
            A.order[atomicAdd(&s_base[bucket_of(tile_count_of(A, t, T_local))], 1u)] = band0 + t;
        return;
    }
    if (A.far_zero)   // (depth-cut frame: the clean-up launches' per-tile counts and cursors; every tile of the grid)
        for (int t = tid; t < 2 * A.g.tw * A.g.th; t += kHistThreads) A.far_zero[t] = 0u;
    __shared__ unsigned int s_nmedium, s_nlarge, s_nxl, s_max, s_on_grid, s_far;
    if (tid == 0) { s_nmedium = 0; s_nlarge = 0; s_nxl = 0; s_max = 0; s_on_grid = 0; s_far = 0; }
    const int prev_redo = (tid == 0 && A.redo_count[2] == redo_signature(A.g)) ? *A.redo_count : 0;
    unsigned int on_grid_part = tid < A.G ? A.wg_on_grid[tid] : 0u;   // G <= kMaxG <= blockDim
    unsigned int far_part = (A.wg_far && tid < A.G) ? A.wg_far[tid] : 0u;   // depth-cut frame: the far log's length
    const unsigned long long grand = tile_prefix_lds(A.tile_count, T_local, s, nullptr, A.xtot);   // (its barriers also publish the zeros above)
    if (tid == 0) s[T_local] = (uint32_t)min(grand, 0x7fffffffull);    // the end of the last tile (the LDS block's spare words)
    __syncthreads();
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        on_grid_part += (unsigned int)__shfl_xor((int)on_grid_part, d);
        far_part += (unsigned int)__shfl_xor((int)far_part, d);
    }
    if (lane == 0 && on_grid_part) atomicAdd(&s_on_grid, on_grid_part);
    if (lane == 0 && far_part) atomicAdd(&s_far, far_part);
    unsigned int lmax = 0;
    for (int t = tid; t < T_local; t += kHistThreads) {
        const uint32_t b = s[t], e = s[t + 1], c = e - b;
        const int at = band0 + t;
        reinterpret_cast<int2 *>(A.tile_ranges)[at] = make_int2((int32_t)b, (int32_t)e);
        A.redo_flag[at] = 0;
        if (A.xtot) const_cast<uint32_t *>(A.tile_count)[t] = c;   // (claimed rows: the counts, for whoever reads them after this launch)
        lmax = max(lmax, c);
        // (a saturated prefix makes c meaningless; such a frame is rejected by the host on info[0])
        if (c > (unsigned)kLargeCapDecl) A.xl_list[atomicAdd(&s_nxl, 1u)] = at;
        else if (c > (unsigned)kMediumCapDecl) A.large_list[atomicAdd(&s_nlarge, 1u)] = at;
        else if (c > (unsigned)kSmallCapDecl) A.medium_list[atomicAdd(&s_nmedium, 1u)] = at;
    }
    if (!A.band_only) {   // the rest of the grid: empty ranges at 0 before the band, at M behind it
        const int32_t m = (int32_t)min(grand, 0x7fffffffull);
        for (int t = tid; t < g.tw * g.th; t += kHistThreads)
            if (t < band0 || t >= band1) {
                reinterpret_cast<int2 *>(A.tile_ranges)[t] = t < band0 ? make_int2(0, 0) : make_int2(m, m);
                A.redo_flag[t] = 0;
            }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) lmax = max(lmax, (unsigned int)__shfl_xor((int)lmax, d));
    if (lane == 0) atomicMax(&s_max, lmax);
    __syncthreads();
    if (tid == 0) {
        // (a depth-cut frame reports ALL its pairs -- the lists' and the far log's: the buffer has to hold both -- and
        // leaves the split in two device-only words behind the record for the clean-up pass)
        const int prev_cut = min(max(A.redo_count[1], 0), max(prev_redo, 0));   // (see tile_scan_total)
        int64_t rec[7] = {(int64_t)grand + (int64_t)s_far, (int64_t)s_max, (int64_t)s_nmedium, (int64_t)s_nlarge, (int64_t)s_nxl,
                          (int64_t)(prev_redo - prev_cut) + ((int64_t)(prev_cut & 0x3fffffff) << 32) +
                              (prev_cut > 0 && A.info[10] != 0 ? (1ll << 62) : 0), (int64_t)s_on_grid};
#pragma unroll
        for (int k = 0; k < 7; ++k) A.info[k] = rec[k];
        A.info[7] = 0;
        A.info[8] = (int64_t)grand;
        A.info[9] = (int64_t)s_far;
        A.info[10] = 0;   // (the clean-up launch's cursor into the regenerated far pairs)
        A.redo_count[0] = 0;
        A.redo_count[1] = 0;
        A.redo_count[2] = redo_signature(A.g);
        if (A.info_mirror) {
#pragma unroll
            for (int k = 0; k < 7; ++k) A.info_mirror[k] = rec[k];
            __threadfence_system();
        }
    }
}

// LEAN: per-position LeanRecs (+ reach masks) instead of means2d / radii / depths.  DEFER: the launch carries the
// scans' total pass (deferred_total; gridDim = G + 2), and the tile starts come from the workgroup's own prefix
// over the tile counts instead of tile_ranges.
#ifndef MS_SCATTER_WAVES
#define MS_SCATTER_WAVES 8   // (round 5: 64 registers for the 12-byte-record variants, two workgroups a CU -- the kernel had grown
                             // to 72 and ran its grid in two rounds; the 16-byte-record variants would spill 40-90 registers: left alone)
#endif
template <bool PACK, int LEAN, bool DEFER>
__global__ __launch_bounds__(kHistThreads, (LEAN == 2 ? MS_SCATTER_WAVES : 4)) void k_isect_scatter(
    int64_t N, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
    const float *__restrict__ depths, const unsigned long long *__restrict__ masks,
    const LeanRec *__restrict__ lean, Grid g, int64_t chunk, const uint32_t *__restrict__ hist,
    const int32_t *__restrict__ tile_ranges, int64_t M, uint64_t *__restrict__ keys,
    uint32_t *__restrict__ wg_depth, Candidates cand, int G, ScanTotalArgs A, unsigned int big_q_off, unsigned int big_q_cap) {
    extern __shared__ uint32_t s_cur[];
    const int T_local = (g.row_end - g.row_begin) * g.tw;
    const int band0 = g.row_begin * g.tw;
    __shared__ unsigned int s_big_n;
    if (threadIdx.x == 0) s_big_n = 0;   // (published by the barriers of the cursors' set-up)
    const BigQ bigq{reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_cur) + big_q_off), &s_big_n, big_q_cap};
    MS_BIN_STAMP(2, 0);
    if constexpr (DEFER) {
        if ((int)blockIdx.x >= G) {   // (uniform per workgroup)
            deferred_total((int)blockIdx.x - G, A, s_cur, T_local);
            MS_BIN_STAMP(2, 4);
            return;
        }
    }
    const int wg = chunk_of_block(blockIdx.x, G);   // the histogram row (and share of the positions: Deal) the count kernel gave this index
    const uint32_t *row = hist + (size_t)wg * T_local;
    (void)chunk;
    int64_t n_pos = N;
    if constexpr (LEAN != 0) {
        if (cand.ids) {   // positions of the band's candidate list (dealt as the count kernel dealt them)
            __shared__ uint32_t s_pref[kMaxG + 1];
            CandMap map{s_pref};
            map.build(cand);
            n_pos = (int64_t)s_pref[cand.n_segs];
        }
    }
    const Deal deal(n_pos, wg, G);
    // The kernel is a chain of round trips (waves sat in s_waitcnt 71 % of their life), so everything the first
    // kAhead steps read -- all a workgroup has at 2 048 Gaussians per chunk -- is on its way before the cursors are set
    // up: the records, and the reach masks (unconditionally: which boxes need theirs is only known from the record).
    // Larger chunks (N / kMaxG Gaussians: 12 steps at 6 M) go kAhead steps at a time, the next kAhead records loaded
    // while these are walked: a wave cannot tell a record's arrival from the completion of the scattered key stores
    // issued since (one counter, in order), so every wait for records also drains the stores -- once per kAhead steps
    // instead of once per step (config 4: 131 -> 58 us).
    constexpr int kAhead = 4;
    LeanRec r_q[kAhead], r_nx[kAhead];
    unsigned long long m_q[kAhead], m_nx[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) { r_q[k] = r_nx[k] = LeanRec{0u, 0u, 0u, 0u}; m_q[k] = m_nx[k] = ~0ull; }
    // depth-cut frame (A.cut_stamp; LEAN == 2 && DEFER only): the count kernel did not count the pairs behind their
    // tile's cut-off, and left the Gaussians that keep a pair records of their own, compacted per workgroup -- box, index,
    // depth bits, the mask of the NEAR tiles of a box of up to 32 tiles; larger boxes, walked by the whole wave, are
    // held against the cut-offs here.  This workgroup walks those records alone.
    bool cut = false;
    if constexpr (LEAN == 2 && DEFER) cut = A.cut_stamp != 0u;
    // trip `it` of this thread's walk -> the position it reads, or -1: the dealt positions (Deal), or -- a depth-cut frame,
    // whatever the walk above would have been -- this workgroup's compacted records, kHistThreads at a time
    int n_trips = deal.iters;
    int64_t cut_i0 = 0, cut_i1 = 0;
    if (cut) {
        cut_i0 = (int64_t)wg * A.near_cap;
        cut_i1 = cut_i0 + (int64_t)A.wg_far[kMaxG + wg];
        n_trips = (int)((cut_i1 - cut_i0 + kHistThreads - 1) / kHistThreads);
    }
    auto pos_of = [&](int it) __attribute__((always_inline)) -> int64_t {
        if (it >= n_trips) return -1;
        if (cut) {
            const int64_t j = cut_i0 + (int64_t)it * kHistThreads + threadIdx.x;
            return j < cut_i1 ? j : -1;
        }
        const int64_t j = deal.base(it) + (threadIdx.x & 63);
        return j < n_pos ? j : -1;
    };
    // (LEAN == 2: a 12-byte record, kept as its three words while it waits -- three registers a record in flight, not
    // six: at 2 x kAhead records the kernel must stay within the 64 registers that let two workgroups share a CU --
    // and unpacked into the 16-byte form when its turn comes)
    auto load_rec = [&](int64_t j, LeanRec &r, unsigned long long &m) __attribute__((always_inline)) {
        if constexpr (LEAN == 2) {
            if (cut) {
                r = A.near_recs[j];   // box, Gaussian, depth bits, near mask
            } else {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(reinterpret_cast<const Lean12 *>(lean) + j);
                r.xy0 = q[0];
                r.xy1 = (uint32_t)j;
                r.depth_bits = q[1];
                r.n_edges = q[2];
            }
        } else {
            r = lean[j];
            if (masks) m = masks[j];
        }
    };
    auto unpack_rec = [&](LeanRec &r, unsigned long long &m, int64_t &gauss) __attribute__((always_inline)) {
        if constexpr (LEAN == 2) {
            const uint32_t box = r.xy0, mk = r.n_edges;
            gauss = (int64_t)r.xy1;   // (a depth-cut frame's compacted records: not the position)
            const uint32_t x0 = box & 0xffu, y0 = (box >> 8) & 0xffu, w = (box >> 16) & 0xffu, h = box >> 24;
            r.xy0 = x0 | (y0 << 16);
            r.xy1 = (x0 + w) | ((y0 + h) << 16);
            r.n_edges = w * h;
            m = w * h > 32u ? ~0ull : (unsigned long long)mk;
        }
    };
    if constexpr (LEAN != 0) {
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int64_t j = pos_of(k);
            if (j >= 0) load_rec(j, r_nx[k], m_nx[k]);
        }
    }
    MS_BIN_STAMP(2, 1);
    uint32_t *s_tau = s_cur + T_local + 4;   // (depth-cut frame: the cut-offs, for the boxes the whole wave walks)
    if (cut)
        for (int t = threadIdx.x; t < T_local; t += kHistThreads) s_tau[t] = A.tau[band0 + t];
    if constexpr (DEFER) {
        (void)tile_prefix_lds(A.tile_count, T_local, s_cur, row, A.xtot, (int)(blockIdx.x & 7) >> kClaimShift);   // (its last barrier publishes the cursors, and a cut frame's cut-offs)
        MS_BIN_STAMP(2, 2);
    } else {
        for (int t = threadIdx.x; t < T_local; t += kHistThreads) {
            uint32_t c = (uint32_t)tile_ranges[2 * (band0 + t)] + row[t];
            if (A.xtot)
                for (int x = 0; x < ((int)(blockIdx.x & 7) >> kClaimShift); ++x) c += A.xtot[(size_t)x * T_local + t];
            s_cur[t] = c;
        }
        __syncthreads();
    }
    // depth bits (order preserving for the positive depths that survive) of everything this workgroup
    // emits: k_tile_front spreads its buckets over the frame's range
    uint32_t dmin = 0xffffffffu, dmax = 0u;
    if constexpr (LEAN != 0) {
        auto emit = [&](int t, int64_t i, int q, uint32_t db) __attribute__((always_inline)) {
            const uint32_t low = PACK ? (((uint32_t)i << 4) | (uint32_t)q) : (uint32_t)i;
            const uint64_t key = ((uint64_t)db << 32) | low;
            const uint32_t slot = atomicAdd(&s_cur[t], 1u);
            if ((int64_t)slot < M) keys[slot] = key;
        };
        auto emit_cut = [&](int t, int64_t i, int q, uint32_t db) __attribute__((always_inline)) { if (db <= s_tau[t]) emit(t, i, q, db); };
        for (int it0 = 0; it0 < n_trips; it0 += kAhead) {
#pragma unroll
            for (int k = 0; k < kAhead; ++k) { r_q[k] = r_nx[k]; m_q[k] = m_nx[k]; }
            if (it0 + kAhead < n_trips) {   // (uniform)
#pragma unroll
                for (int k = 0; k < kAhead; ++k) {
                    const int64_t jn = pos_of(it0 + kAhead + k);
                    if (jn >= 0) load_rec(jn, r_nx[k], m_nx[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < kAhead; ++k) {
                if (it0 + k >= n_trips) break;   // (uniform)
                LeanRec r = r_q[k];
                unsigned long long mk = m_q[k];
                const int64_t j = pos_of(it0 + k);
                int64_t gi = j;   // the Gaussian (the list id): the position, unless the record names it
                unpack_rec(r, mk, gi);
                int x0 = 0, x1 = 0, y0 = 0, y1 = 0, n = 0, edges = 0;
                unsigned long long mask = ~0ull;
                if (j >= 0) {
                    n = (int)(r.n_edges & 0x0fffffffu);
                    edges = (int)(r.n_edges >> 28);
                    x0 = (int)(r.xy0 & 0xffffu); y0 = (int)(r.xy0 >> 16);
                    x1 = (int)(r.xy1 & 0xffffu); y1 = (int)(r.xy1 >> 16);
                    if (masks && (n > 1 || (PACK && n > 0) || cut)) mask = mk;
                }
                const uint32_t dbits = r.depth_bits;
                if (n > 0) { dmin = min(dmin, dbits); dmax = max(dmax, dbits); }
                if constexpr (!PACK) {
                    // Boxes of up to 32 tiles, each lane its own -- FOUR pairs a pass: the cursor of a pair comes back
                    // from LDS ~130 cycles after it is asked for, and a loop that asks, waits and stores pair by pair
                    // spends its life in that wait (measured: the kernel's time followed the number of RECORDS, not of
                    // pairs); four cursors in flight per wait, a quarter of the passes.
                    const int w = x1 - x0;
                    const float inv_w = __builtin_amdgcn_rcpf((float)max(w, 1));
                    unsigned int m = (n > 0 && n <= kCoopThreshold) ? ((unsigned int)mask & (n >= 32 ? 0xffffffffu : ((1u << n) - 1u))) : 0u;
                    const uint64_t key = ((uint64_t)dbits << 32) | (uint32_t)gi;
                    while (m) {
                        int tt[4];
                        bool vv[4];
                        uint32_t slot[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            vv[u] = m != 0u;
                            const int k = vv[u] ? __ffs((int)m) - 1 : 0;
                            m &= m - 1u;   // (0 stays 0)
                            const int r = (int)(((float)k + 0.5f) * inv_w);   // k / w, exact for k < 64
                            tt[u] = (y0 + r - g.row_begin) * g.tw + x0 + (k - r * w);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (vv[u]) slot[u] = atomicAdd(&s_cur[tt[u]], 1u);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (vv[u] && (int64_t)slot[u] < M) keys[slot[u]] = key;
                    }
                    // larger boxes: the whole wave (a depth-cut frame holds them against the cut-offs here)
                    if (cut)
                        walk_boxes<PACK>((int)gi, x0, x1, y0, y1, n > kCoopThreshold ? n : 0, edges, g, mask, emit_cut, dbits, &bigq);
                    else
                        walk_boxes<PACK>((int)gi, x0, x1, y0, y1, n > kCoopThreshold ? n : 0, edges, g, mask, emit, dbits, &bigq);
                } else {
                    walk_boxes<PACK>((int)gi, x0, x1, y0, y1, n, edges, g, mask, emit, dbits, &bigq);
                }
            }
        }
        // (the boxes of more than kCoopThreshold tiles that the waves queued: all sixteen waves take them in turn)
        if (cut) drain_big_boxes<PACK, true>(bigq, g, emit_cut);
        else drain_big_boxes<PACK, true>(bigq, g, emit);
    } else {
        for_each_isect<PACK>(0, N, means2d, radii, masks, g, nullptr, nullptr, cand, G, [&](int t, int64_t i, int q) {
            const uint32_t slot = atomicAdd(&s_cur[t], 1u);
            const uint32_t low = PACK ? (((uint32_t)i << 4) | (uint32_t)q) : (uint32_t)i;
            const uint32_t dbits = __float_as_uint(depths[i]);
            dmin = min(dmin, dbits);
            dmax = max(dmax, dbits);
            const uint64_t key = ((uint64_t)dbits << 32) | low;
            if ((int64_t)slot < M) keys[slot] = key;
        });
    }
    MS_BIN_STAMP(2, 3);
#ifdef MS_DIAG
    if constexpr (LEAN != 0) { __syncthreads(); MS_BIN_STAMP(2, 4); }   // (when the LAST wave is done)
#endif
    if (wg_depth) {
        uint32_t *s_depth = s_cur + T_local + 1;   // (the 16 spare bytes every binning kernel's LDS block ends with)
        __syncthreads();                           // (the cursors are done with)
        if (threadIdx.x == 0) { s_depth[0] = 0xffffffffu; s_depth[1] = 0u; }
        __syncthreads();
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, d));
            dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, d));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_depth[0], dmin); atomicMax(&s_depth[1], dmax); }
        __syncthreads();
        if (threadIdx.x == 0) { wg_depth[2 * blockIdx.x] = s_depth[0]; wg_depth[2 * blockIdx.x + 1] = s_depth[1]; }
    }
}

}  // namespace
#include "sort_device.hpp"
namespace {

constexpr int kSmallCap = SortCfg<256, 4>::CAP;     // 1024: one 256-thread workgroup, 12 KB LDS
constexpr int kMediumCap = SortCfg<1024, 8>::CAP;   // 8192: 1024 threads, 80 KB LDS
constexpr int kLargeCap = SortCfg<1024, 16>::CAP;   // 16384: 1024 threads, 144 KB LDS

template <bool SPLIT>
__global__ __launch_bounds__(256) void k_tile_sort_small(const int32_t *__restrict__ tile_ranges,
                                                         const uint64_t *__restrict__ keys,
                                                         int32_t *__restrict__ flatten_ids,
                                                         int64_t *__restrict__ isect_ids, int64_t cap,
                                                         int tile_lo, ms::BlockLists blocks, int bin_w) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SortCfg<256, 4>::LDS];
    const int tile = tile_lo + blockIdx.x;   // the band's tiles: ranges outside it are empty (or unwritten)
    const int start = tile_ranges[2 * tile], n = tile_ranges[2 * tile + 1] - start;
    const bool skip = n <= 0 || (int64_t)start + n > cap;   // beyond cap: speculative overflow
    if (SPLIT && (skip || n > kSmallCap)) {
        // split frame: empty ranges for every bin this kernel does not sort (the front kernel, launched after
        // it, overwrites those of the heavy bins)
        if (threadIdx.x < 4) {
            const int by = tile / bin_w, bx = tile - by * bin_w;
            const int x = 2 * bx + (threadIdx.x & 1), y = 2 * by + (threadIdx.x >> 1);
            if (x < blocks.tw16 && y < blocks.th16)
                reinterpret_cast<int2 *>(blocks.block_ranges)[y * blocks.tw16 + x] = make_int2(0, 0);
        }
        if (threadIdx.x == 0) blocks.bin_more[tile] = 0;
    }
    if (skip || n > kSmallCap) return;
    sort_segment_lds<256, 4>(smem, keys, start, n, tile, flatten_ids, isect_ids, nullptr, SPLIT ? &blocks : nullptr,
                             bin_w);
}

template <int E>
__global__ __launch_bounds__(1024, (E <= 8 ? 8 : 4)) void k_tile_sort_list(const int32_t *__restrict__ list,
                                                         const int32_t *__restrict__ tile_ranges,
                                                         const uint64_t *__restrict__ keys,
                                                         int32_t *__restrict__ flatten_ids,
                                                         int64_t *__restrict__ isect_ids,
                                                         const int64_t *__restrict__ n_list_dev,
                                                         int n_list_host, int64_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    // the list length is either known to the host or read from the count pass's device record
    // (sync-free frames launch a fixed grid and stride over the list)
    const int n_list = n_list_dev ? (int)*n_list_dev : n_list_host;
    for (int li = blockIdx.x; li < n_list; li += gridDim.x) {
        const int tile = list[li];
        const int start = tile_ranges[2 * tile], n = tile_ranges[2 * tile + 1] - start;
        if ((int64_t)start + n <= cap)
            sort_segment_lds<1024, E>(smem_dyn, keys, start, n, tile, flatten_ids, isect_ids, nullptr);
        __syncthreads();  // LDS is reused by the next list entry
    }
}

// ---- lazy sorting (fused path) ----------------------------------------------------------------
// The rasteriser stops reading a tile's list once its 256 pixels are saturated -- after a few
// hundred entries even where the list holds tens of thousands (measured: at most 459 entries of any
// tile at config 3, 309 of lists up to 30 510 long at config 4; 3.9 % of all entries there).  So a
// HEAVY tile (> 1024 entries) does not get its list sorted: only its FRONT -- the ~kFrontK nearest
// entries -- is selected and sorted; the rasteriser is told how long the sorted front is and raises
// a flag in the (so far unseen) case that pixels are still alive at its end, and a clean-up kernel
// redoes exactly those tiles.
//   pass A: min / max of the depth bits;  pass B: LDS histogram over 2048 order-preserving buckets;
//   scan; b* = first bucket whose inclusive prefix reaches kFrontK;  pass C: keys of buckets <= b*
//   go to LDS, rank themselves inside their bucket (as in sort_segment_lds) and leave as ids.
// Three streams over the tile's keys (fresh from the scatter, mostly L2 / Infinity-Cache hits)
// instead of a full sort; tiles of any length, no merge scratch, no host knowledge of sizes.
#ifndef MS_MERGED_WAVES
#define MS_MERGED_WAVES 6   // waves per SIMD the merged sort kernel is compiled for (8: spills; 6 measured best, profiles/r02_merged_sort.md)
#endif
constexpr int kFrontThreads = 512;           // more workgroups in flight than with 1024 (heavy tiles are latency bound)
static_assert(kFrontNB == 4 * kFrontThreads, "four buckets per thread in the scan");
constexpr size_t kFrontLds = (size_t)kFrontCap * 8 + (size_t)kFrontNB * 4 + 64 * 4 + 16;

// MERGED (plain bins of a lazily sorted frame): ONE launch sorts every list of the band -- block b takes tile
// order[b] (the count pass's heaviest-first order), sorts it whole if it is short (sort_segment_lds<512, 2>, what
// k_tile_sort_small does with 256 threads) and selects + sorts its front otherwise.  The heavy tiles' latency
// chains (two streams over their keys, seven barriers) start first and the short sorts fill the chip beside
// and behind them, instead of a front launch (480 workgroups at config 3: 16 us) FOLLOWED by a small-sort
// launch (10 us).  No list of heavy tiles, no host knowledge of their number.
template <bool SPLIT, bool MERGED>
__global__ __launch_bounds__(kFrontThreads, (MERGED ? MS_MERGED_WAVES : 1)) void k_tile_front(const int32_t *__restrict__ medium,
                                                     const int32_t *__restrict__ large,
                                                     const int32_t *__restrict__ xl,
                                                     const int64_t *__restrict__ info_dev, int nm_host, int nl_host,
                                                     int nx_host, const int32_t *__restrict__ tile_ranges,
                                                     const uint64_t *__restrict__ keys,
                                                     int32_t *__restrict__ flatten_ids,
                                                     int32_t *__restrict__ front_count, int64_t cap,
                                                     uint32_t fixed_min, int fixed_shift, int front_k,
                                                     ms::BlockLists blocks, int bin_w,
                                                     const uint32_t *__restrict__ wg_depth, int n_wg,
                                                     const int32_t *__restrict__ order, int n_order, int front_cap,
                                                     const uint32_t *__restrict__ n_big_dev,
                                                     uint32_t *__restrict__ tau, const uint32_t *__restrict__ tau_now,
                                                     const uint32_t *__restrict__ has_far, uint32_t cut_stamp) {
    static_assert(!(SPLIT && MERGED), "block lists are cut by the two-launch path");
    // DEPTH CUT-OFF of the NEXT frame (tau, merged launch only): a tile whose list was sorted whole keeps every pair next
    // time (0xffffffff) -- unless this frame's list was itself cut short (the tile is marked) and sufficed: then this
    // frame's cut-off (tau_now) stands; a heavy tile's cut-off is where its sorted front ended, plus a quarter of the
    // front's depth span.  (A tile that outlives its list after all is reset by the clean-up launch.)
    auto next_cut_whole = [&](int tile) {
        if (tau) tau[tile] = (cut_stamp && has_far[tile] == cut_stamp) ? tau_now[tile] : 0xffffffffu;
    };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    uint64_t *s_out = reinterpret_cast<uint64_t *>(smem_dyn);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_out + front_cap);   // front_cap <= kFrontCap keys of LDS room
    uint32_t *s_red = s_cnt + kFrontNB;          // 64 words of reduction scratch
    int *s_sel = reinterpret_cast<int *>(s_red + 64);  // [0] = b*, [1] = F
    constexpr int THREADS = kFrontThreads, NW = kFrontThreads / 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nm = info_dev ? (int)info_dev[2] : nm_host, nl = info_dev ? (int)info_dev[3] : nl_host;
    const int nx = info_dev ? (int)info_dev[4] : nx_host;
    const int total = MERGED ? n_order : nm + nl + nx;
    if ((int)blockIdx.x >= total) return;   // (uniform; a sync-free launch is sized for the worst case)
    const int n_big = (MERGED && n_big_dev) ? min((int)*n_big_dev, total) : -1;
    if constexpr (MERGED) {
        // LIGHT lists (at most 255 entries; the order lists them last, from index n_big on): a whole 512-thread
        // workgroup per list spends 4 us of barriers and round trips on each -- 907 of config 3's 2 040 bins, which
        // only find a slot once the heavy bins have finished (profiles/r03_bin_phases.txt) -- so they are sorted
        // EIGHT to a workgroup, one per wave, wave-synchronously (no workgroup barrier on this path).
        if (n_big >= 0 && (int)blockIdx.x >= n_big) {
            const int e = n_big + 8 * ((int)blockIdx.x - n_big) + w;
            if (e < total) {
                const int tile = order[e];
                const int start = tile_ranges[2 * tile], n = tile_ranges[2 * tile + 1] - start;
                if (n > 0 && n <= kLightCap && (int64_t)start + n <= cap)
                    sort_segment_lds<64, kLightCap / 64>(smem_dyn + (size_t)w * SortCfg<64, kLightCap / 64>::LDS, keys, start, n, tile,
                                                         flatten_ids, nullptr, nullptr);
                if (lane == 0) next_cut_whole(tile);
            }
            return;
        }
    }
#ifdef MS_DIAG
    if (g_diag_bin && tid == 0) g_diag_bin[(3 * 1024 + (blockIdx.x & 1023)) * 8 + (blockIdx.x >> 10) * 3] = __builtin_amdgcn_s_memrealtime();
#endif
    // The frame's own depth range (per-workgroup min / max left by the scatter) beats the camera planes:
    // the scene fills a fraction of (near, far), and buckets that are several times finer make the
    // ranking inside a bucket as many times shorter.  (MERGED: only a block that meets a heavy tile needs it.)
    bool range_known = false;
    auto frame_range = [&]() {
      if (wg_depth && !range_known) {
        uint32_t lo = 0xffffffffu, hi = 0u;
        for (int j = tid; j < n_wg; j += THREADS) {
            lo = min(lo, wg_depth[2 * j]);
            hi = max(hi, wg_depth[2 * j + 1]);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, d));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, d));
        }
        if (lane == 0) { s_red[w] = lo; s_red[16 + w] = hi; }
        __syncthreads();
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) { lo = min(lo, s_red[ww]); hi = max(hi, s_red[16 + ww]); }
        __syncthreads();
        if (hi >= lo) {
            const uint32_t span = hi - lo;
            const int bits = span ? 32 - __clz(span) : 0;
            fixed_min = lo;
            fixed_shift = max(0, bits - kFrontLogNB);
        }
      }
      range_known = true;
    };
    if constexpr (!MERGED) frame_range();
    for (int li = blockIdx.x; li < total; li += gridDim.x) {
        const int tile = MERGED ? order[li] : li < nm ? medium[li] : (li < nm + nl ? large[li - nm] : xl[li - nm - nl]);
        const int start = tile_ranges[2 * tile], n = tile_ranges[2 * tile + 1] - start;
        if constexpr (MERGED) {
            if (n <= kSmallCap) {   // a short list: sorted whole (uniform per workgroup)
                if (n > 0 && (int64_t)start + n <= cap)
                    sort_segment_lds<kFrontThreads, kSmallCap / kFrontThreads>(smem_dyn, keys, start, n, tile, flatten_ids,
                                                                               nullptr, nullptr);
                if (tid == 0) next_cut_whole(tile);
                __syncthreads();   // LDS is reused by the next list entry
#ifdef MS_DIAG
                if (g_diag_bin && tid == 0) {
                    unsigned long long *d = g_diag_bin + (3 * 1024 + (blockIdx.x & 1023)) * 8 + (blockIdx.x >> 10) * 3;
                    d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = (unsigned long long)n;
                }
#endif
                continue;
            }
            frame_range();
        }
        if ((int64_t)start + n > cap) {   // speculative overflow: the frame is redone (uniform)
            if (SPLIT) {       // ... but its rasteriser must find ranges it can walk
                if (tid < 4) {
                    const int by = tile / bin_w, bx = tile - by * bin_w;
                    const int x = 2 * bx + (tid & 1), y = 2 * by + (tid >> 1);
                    if (x < blocks.tw16 && y < blocks.th16)
                        reinterpret_cast<int2 *>(blocks.block_ranges)[y * blocks.tw16 + x] = make_int2(0, 0);
                }
                if (tid == 0) blocks.bin_more[tile] = 0;
            }
            continue;
        }
        const int F = front_select_lds<THREADS, (SPLIT ? 8 : 4)>(keys + start, n, s_out, s_cnt, s_red, s_sel, fixed_min, fixed_shift,
                                                                 front_k, front_cap);
        if constexpr (SPLIT) emit_block_lists<THREADS, kFrontCap / THREADS>(s_out, F, n, start, tile, bin_w, blocks, s_cnt);
        else
            for (int i = tid; i < F; i += THREADS) flatten_ids[start + i] = (int32_t)(uint32_t)s_out[i];
        if (tid == 0) front_count[tile] = F;
        if (MERGED && tid == 0 && tau) {
            if (F >= n) next_cut_whole(tile);   // (the front is the whole list)
            else {
                // (the margin: 1/16 of an octave of depth -- 2^19 steps of the float's bits, +4.4 % -- beyond the bucket that
                // completed the front; a margin taken from the front's own depth span is at the mercy of the list's
                // nearest entry: one Gaussian at the camera's feet, in every bin of config 4, made it two octaves)
                const uint32_t edge = (uint32_t)s_sel[2];
                const unsigned long long cut = (unsigned long long)edge + (1ull << 19);
                tau[tile] = (uint32_t)(cut > 0xffffffffull ? 0xffffffffull : cut);
            }
        }
        __syncthreads();   // LDS is reused by the next list entry
#ifdef MS_DIAG
        if (g_diag_bin && tid == 0) {
            unsigned long long *d = g_diag_bin + (3 * 1024 + (blockIdx.x & 1023)) * 8 + (blockIdx.x >> 10) * 3;
            d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = (unsigned long long)n;
        }
#endif
    }
}

// XL tiles: sort runs of kLargeCap in place ...
__global__ __launch_bounds__(1024) void k_xl_chunk_sort(const int32_t *__restrict__ xl_list,
                                                        const int32_t *__restrict__ tile_ranges,
                                                        uint64_t *__restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    const int tile = xl_list[blockIdx.y];
    const int start = tile_ranges[2 * tile], n = tile_ranges[2 * tile + 1] - start;
    const int c0 = blockIdx.x * kLargeCap;
    if (c0 >= n) return;
    const int cn = min(kLargeCap, n - c0);
    sort_segment_lds<1024, 16>(smem_dyn, keys, start + c0, cn, tile, nullptr, nullptr, keys);
}

// ... then merge neighbouring sorted runs of length L.  Keys inside a tile are distinct
// (the Gaussian index is part of the key), so rank = own position + lower_bound in partner.
__global__ __launch_bounds__(256) void k_xl_merge(const int32_t *__restrict__ xl_list,
                                                  const int32_t *__restrict__ tile_ranges,
                                                  const uint64_t *__restrict__ src,
                                                  uint64_t *__restrict__ dst, int L) {
    const int tile = xl_list[blockIdx.y];
    const int start = tile_ranges[2 * tile], n = tile_ranges[2 * tile + 1] - start;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const uint64_t key = src[start + e];
    const int run = e / L, pos = e - run * L;
    const int pair_base = (run & ~1) * L;
    const int pb = (run ^ 1) * L;            // partner run begin
    int cnt = 0;
    if (pb < n) {
        const int pn = min(L, n - pb);
        int lo = 0, hi = pn;                 // lower_bound(partner, key)
        const uint64_t *p = src + start + pb;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (p[mid] < key) lo = mid + 1; else hi = mid;
        }
        cnt = lo;
    }
    dst[start + pair_base + pos + cnt] = key;
}

__global__ __launch_bounds__(256) void k_xl_extract(const int32_t *__restrict__ xl_list,
                                                    const int32_t *__restrict__ tile_ranges,
                                                    const uint64_t *__restrict__ src,
                                                    int32_t *__restrict__ flatten_ids,
                                                    int64_t *__restrict__ isect_ids) {
    const int tile = xl_list[blockIdx.y];
    const int start = tile_ranges[2 * tile], n = tile_ranges[2 * tile + 1] - start;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const uint64_t k = src[start + e];
    flatten_ids[start + e] = (int32_t)(uint32_t)k;
    if (isect_ids) isect_ids[start + e] = ((int64_t)tile << 32) | (int64_t)(k >> 32);
}

__global__ __launch_bounds__(256) void k_offset_encode(int64_t M, const int64_t *__restrict__ ids,
                                                       int T, int32_t *__restrict__ offsets) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M) return;
    const int cur = (int)(ids[idx] >> 32);
    if (idx == 0)
        for (int t = 0; t <= cur && t < T; ++t) offsets[t] = 0;
    else {
        const int prev = (int)(ids[idx - 1] >> 32);
        for (int t = prev + 1; t <= cur && t < T; ++t) offsets[t] = (int32_t)idx;
    }
    if (idx == M - 1)
        for (int t = cur + 1; t < T; ++t) offsets[t] = (int32_t)M;
}

// ---- host side ----------------------------------------------------------------------
struct Plan {
    int G;
    int64_t chunk;
    int64_t near_cap;   // depth-cut frames: the stride of a count workgroup's compacted records (>= what any workgroup walks, band candidates included)
    int T, T_local;
    size_t lds_bytes;
    size_t off_xtot;   // claimed rows: eight arrays of T counters, one per XCD
    size_t off_hist, off_count, off_medium, off_large, off_xl, off_on_grid, off_mask, off_front, off_redo_flag,
        off_redo_list, off_redo_count, off_depth_wg, off_order, off_tau, off_has_far, off_wg_far, off_far_seg, off_cand_count, off_cand, off_lean, off_near, total;
};

bool make_plan(int64_t N, int tw, int th, int row_begin, int row_end, Plan &p) {
    p.T = tw * th;
    p.T_local = (row_end - row_begin) * tw;
    p.lds_bytes = (size_t)p.T_local * 4 + 16;  // + the on-grid counter of the count kernels
    int64_t G = ms::ceil_div(N > 0 ? N : 1, MS_CHUNK);
    // scenes too small to give every CU a workgroup at MS_CHUNK Gaussians each (100k: 49 workgroups on 256 CUs,
    // k_project_hist 20.5 us against 32 for ten times as many): one step of kHistThreads Gaussians per workgroup
    if (G < 256) G = ms::ceil_div(N > 0 ? N : 1, kHistThreads) < 256 ? ms::ceil_div(N > 0 ? N : 1, kHistThreads) : 256;
    G = G < 1 ? 1 : (G > kMaxG ? kMaxG : G);
    if (p.lds_bytes > 64 * 1024 && G > 256) G = 256;
    p.chunk = ms::ceil_div(N > 0 ? N : 1, G);
    p.G = (int)ms::ceil_div(N > 0 ? N : 1, p.chunk);
    p.near_cap = ms::ceil_div(ms::ceil_div(N > 0 ? N : 1, (int64_t)kHistThreads), (int64_t)p.G) * kHistThreads;
    // workspace carve-up is sized for the FULL grid so one buffer serves any band
    size_t o = 0;
    p.off_hist = o;   o += ms::align_up((size_t)kMaxG * p.T * 4, 256);
    p.off_count = o;  o += ms::align_up((size_t)p.T * 4, 256);
    p.off_xtot = o;   o += ms::align_up((size_t)8 * p.T * 4, 256);
    p.off_medium = o; o += ms::align_up((size_t)p.T * 4, 256);
    p.off_large = o;  o += ms::align_up((size_t)p.T * 4, 256);
    p.off_xl = o;     o += ms::align_up((size_t)p.T * 4, 256);
    p.off_on_grid = o; o += ms::align_up((size_t)kMaxG * 4 + 64, 256);   // + the scan kernel's arrival ticket (word kMaxG)
    p.off_front = o;      o += ms::align_up((size_t)p.T * 4, 256);  // lazy sorting: sorted-front length per tile
    p.off_redo_flag = o;  o += ms::align_up((size_t)p.T * 4, 256);  //   tiles whose front did not saturate them
    p.off_redo_list = o;  o += ms::align_up((size_t)p.T * 4, 256);
    p.off_redo_count = o; o += 256;
    p.off_depth_wg = o;   o += ms::align_up((size_t)kMaxG * 8, 256);   // per-workgroup depth-bit min / max (k_isect_scatter)
    p.off_order = o;      o += ms::align_up((size_t)p.T * 4, 256);     // the band's tiles, heaviest first (rasteriser launch order)
    p.off_tau = o;        o += ms::align_up((size_t)p.T * 8, 256);     // depth cut: per tile, the depth bits behind which a pair is "far" -- two buffers of T words:
                                                                       //   a frame reads the one the previous frame's sort kernel wrote and writes the other
    p.off_has_far = o;    o += ms::align_up((size_t)p.T * 4, 256);     //   this frame's tiles that own far pairs
    p.off_wg_far = o;     o += ms::align_up((size_t)kMaxG * 8, 256);   //   far pairs per count workgroup | records that keep a pair per count workgroup
    p.off_far_seg = o;    o += ms::align_up((size_t)p.T * 12, 256);    //   clean-up: regenerated far pairs per tile -- count, cursor (zeroed by the frame's total pass), start
    // (the only N-dependent block comes last: everything above -- the clean-up count among it, which a caller
    // reads back one frame later -- stays where it is when the scene grows or shrinks on a fixed grid)
    p.off_cand_count = o; o += ms::align_up((size_t)kMaxG * 4, 256);   // band pre-cull: survivors per segment
    p.off_mask = o;   o += ms::align_up((size_t)(N > 0 ? N : 1) * 8, 256);  // tight binning: reach masks
    p.off_cand = o;   o += ms::align_up((size_t)(N > 0 ? N : 1) * 4, 256);  // band pre-cull: candidate list
    p.off_lean = o;   o += ms::align_up((size_t)(N > 0 ? N : 1) * sizeof(LeanRec), 256);  // lean frames: box + depth per position
    p.off_near = o;   o += ms::align_up((size_t)((N > 0 ? N : 1) + (int64_t)kHistThreads * (kMaxG + 1)) * sizeof(LeanRec), 256);  // depth-cut frames: the records that keep a pair, compacted per count workgroup (G strides of near_cap)
    p.total = o;
    return p.lds_bytes <= kMaxLds;
}

int check_grid(int tile_size, int tw, int th, int row_begin, int row_end) {
    MS_REQUIRE(tile_size > 0 && tw > 0 && th > 0, MS_ERR_INVALID_ARG, "isect: bad tile grid");
    MS_REQUIRE((int64_t)tw * th < (1ll << 30), MS_ERR_TOO_LARGE, "isect: tile grid too large");
    MS_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= th, MS_ERR_INVALID_ARG,
               "isect: bad row band [%d,%d) of %d", row_begin, row_end, th);
    return MS_OK;
}

// entries of a workgroup's queue of big boxes (BigQ) behind `used` bytes of dynamic LDS: up to 256 (8 KB), fewer where the
// counters of a large grid leave less room, none on grids whose tile coordinates do not fit 16 bits
unsigned int big_queue_cap(size_t used, int tile_w, int tile_h) {
    if (tile_w > 0xffff || tile_h > 0xffff || used >= kMaxLds) return 0u;
    const size_t room = (kMaxLds - used) / 32;
    return (unsigned int)(room < 256 ? room : 256);
}

// raise a kernel's dynamic-LDS ceiling to the full 160 KB: once per (device, kernel) -- the attribute
// belongs to the function as loaded on the CURRENT device -- remembered in a small table under a mutex
// (host threads rendering on several devices of one process share it)
template <class K>
int allow_big_lds(K kernel) {
    struct Seen { int dev; const void *fn; };
    static std::mutex mu;
    static std::vector<Seen> seen;
    const void *f = reinterpret_cast<const void *>(kernel);
    int dev = -1;
    MS_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    for (const Seen &e : seen)
        if (e.dev == dev && e.fn == f) return MS_OK;
    MS_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
    seen.push_back({dev, f});
    return MS_OK;
}

}  // namespace

namespace {
// per-tile prefix over the partial histograms -> tile_ranges, M, work lists
int count_tail(const Plan &p, const Grid &g, char *ws, uint32_t *hist, uint32_t *count, int32_t *medium,
               int32_t *large, int32_t *xl, const uint32_t *wg_on_grid, int n_wg, int32_t *tile_ranges,
               int64_t *isect_info, int band_only, int64_t *info_mirror, hipStream_t stream, bool defer_total = false) {
    ScanTotalArgs A{g, count, tile_ranges, medium, large, xl, wg_on_grid, n_wg, (int32_t *)(ws + p.off_redo_flag),
                    (int32_t *)(ws + p.off_redo_count), band_only, isect_info, info_mirror, (int32_t *)(ws + p.off_order),
                    (uint32_t *)wg_on_grid + kMaxG, nullptr, nullptr, 0u, nullptr, 0, nullptr, nullptr};
    if (defer_total && p.T_local > 0) {
        // the per-tile prefix over the partial rows alone: the total pass rides in the scatter launch (deferred_total)
        A.ticket = nullptr;
        if (p.T_local <= 4096)
            hipLaunchKernelGGL(k_tile_scan_wg<16>, dim3((unsigned)ms::ceil_div(p.T_local, 16)), dim3(1024), 0, stream,
                               p.G, p.T_local, hist, count, A);
        else
            hipLaunchKernelGGL(k_tile_scan_wg<64>, dim3((unsigned)ms::ceil_div(p.T_local, 64)), dim3(1024), 0, stream,
                               p.G, p.T_local, hist, count, A);
        MS_LAUNCH_CHECK();
        return MS_OK;
    }
    if (p.T_local > 0) {
        // (the last workgroup to arrive runs the total pass)
        if (p.T_local <= 4096)
            hipLaunchKernelGGL(k_tile_scan_wg<16>, dim3((unsigned)ms::ceil_div(p.T_local, 16)), dim3(1024), 0, stream,
                               p.G, p.T_local, hist, count, A);
        else {
            // (and the total pass in a launch of its own: 17 us in two launches, 24 in one, on 8 160 tiles)
            ScanTotalArgs A0 = A;
            A0.ticket = nullptr;
            hipLaunchKernelGGL(k_tile_scan_wg<64>, dim3((unsigned)ms::ceil_div(p.T_local, 64)), dim3(1024), 0, stream,
                               p.G, p.T_local, hist, count, A0);
            hipLaunchKernelGGL(k_tile_scan_total, dim3(1), dim3(1024), 0, stream, A0);
        }
    } else {
        A.ticket = nullptr;
        hipLaunchKernelGGL(k_tile_scan_total, dim3(1), dim3(1024), 0, stream, A);
    }
    MS_LAUNCH_CHECK();
    return MS_OK;
}
}  // namespace

// How deep the lazily sorted fronts of a frame are and how their depth buckets are laid out (k_tile_front).  `lazy`: bits 1-2 = the front level the caller asked for.
ms::FrontParams ms::front_params(int tile_size, int lazy, float depth_near, float depth_far, bool merged, bool split) {
    ms::FrontParams fp;
    // all surviving depths lie in the camera's (near, far): fixed order-preserving buckets
    fp.fixed_min = 0;
    fp.fixed_shift = -1;
    if (depth_near > 0.f && depth_far > depth_near) {
        uint32_t lo, hi;
        memcpy(&lo, &depth_near, 4);
        memcpy(&hi, &depth_far, 4);
        const uint32_t span = hi - lo;
        int bits = 0;
        while (bits < 32 && (span >> bits)) ++bits;
        fp.fixed_min = lo;
        fp.fixed_shift = bits > kFrontLogNB ? bits - kFrontLogNB : 0;
    }
    // a tile of (tile_size/16)^2 blocks needs that many times the front of one block (up to the LDS room)
    const int blocks_per_tile = ((tile_size + 15) / 16) * ((tile_size + 15) / 16);
    // split frames: 1280 of a 32-px bin's nearest entries saturate its four blocks on the BASELINE
    // scenes (1024: one bin of config 3 falls short; 2048: +9 us).  A caller that saw the clean-up pass
    // run asks for deeper fronts (lazy bits 1-2: front level, x2 each, up to the LDS room).
    static const int front_env = [] { const char *e = getenv("MOJOSPLAT_FRONT_K"); return e ? atoi(e) : 0; }();
    // plain 32-px bins: 1536 (config 3: 2048 costs +5 us per frame, 1024 sends one bin to the clean-up pass)
    int front_k = split ? 1280 : blocks_per_tile == 4 ? 1536 : min(kFrontK * blocks_per_tile, 2048);
    if (front_env >= 256 && front_env <= kFrontCap && blocks_per_tile > 1) front_k = front_env;   // (measurements)
    fp.front_k = min(front_k << ((lazy >> 1) & 3), kFrontCap);   // (3072: 1-3 % slower, no fewer clean-ups on the BASELINE scenes)
    // LDS room for the selected keys: the front plus the bucket that completes it; the merged launch
    // keeps it small (2048 keys: 25 KB per workgroup, four 512-thread workgroups per CU)
    fp.front_cap = merged ? min(kFrontCap, max(2048, (fp.front_k * 4 / 3 + 511) & ~511)) : kFrontCap;
    return fp;
}

// Where the lazy-sorting bookkeeping lives inside an isect workspace (for ms_render_fwd's rasteriser).
void ms::isect_lazy_arrays(void *workspace, int64_t N, int tile_w, int tile_h, ms::LazyLists *out) {
    Plan p;
    make_plan(N, tile_w, tile_h, 0, tile_h, p);
    char *ws = (char *)workspace;
    out->front_count = (int32_t *)(ws + p.off_front);
    out->redo_flag = (int32_t *)(ws + p.off_redo_flag);
    out->redo_list = (int32_t *)(ws + p.off_redo_list);
    out->redo_count = (int32_t *)(ws + p.off_redo_count);
    out->front_threshold = kFrontK;
    out->keys = nullptr;  // the caller knows where it put them
    out->bin_more = nullptr;
    out->bin_w = 0;
    out->packed = 0;
    out->row_lo = 0;
    out->row_hi = 0x7fffffff;
    out->redo_grid = 256;
    out->redo_sort = 0;
    out->cut_stamp = 0;   // (a depth-cut frame's caller sets the stamp, the cut-off buffers, the record's words and the log)
    out->tau = (uint32_t *)(ws + p.off_tau);   // (buffer 0; buffer 1 follows T words on: the caller picks)
    out->tau_next = nullptr;
    out->has_far = (const uint32_t *)(ws + p.off_has_far);
    out->lean = ws + p.off_lean;
    out->n_lean = N;
    out->cut_words = nullptr;
    out->log_keys = nullptr;
    out->far_cnt = (uint32_t *)(ws + p.off_far_seg);
    out->far_cur = out->far_cnt + p.T;
    out->far_start = out->far_cnt + 2 * (size_t)p.T;
    out->cut_inputs = nullptr;
    out->verdict = nullptr;
    out->cut_inputs = nullptr;
}

// Can a frame on this grid take the depth cut (12-byte box records, room for the cut-offs in the count kernel's LDS)?
bool ms::depth_cut_fits(int64_t N, int tile_w, int tile_h) {
    Plan p;
    if (!make_plan(N, tile_w, tile_h, 0, tile_h, p)) return false;
    return tile_w <= 255 && tile_h <= 255 && N > 0 && N < (1ll << 28) &&
           p.lds_bytes + (size_t)p.T_local * 5 + 16 <= kMaxLds;
}

namespace {
// Depth-cut frame (binning.hip, k_project_hist): the pairs behind a tile's cut-off were counted but never written.
// When the rasteriser put tiles that own such pairs on the redo list, two launches -- between the rasteriser and
// k_tile_redo, empty otherwise: every workgroup reads the redo count and leaves -- walk the frame's 12-byte box
// records again: PASS 0 counts the dropped pairs of THOSE tiles (far_cnt), PASS 1 gives every such tile a segment of
// the free tail of the key array behind the lists (far_start: a prefix over the redo list that every workgroup takes
// for itself; the size record counted the dropped pairs, so the buffer has room for all of them) and writes the keys.
// The same pairs the scatter kernel would have written: box, reach mask and depth bits are the record's.
// PASS 1 also writes the rasteriser's record of every Gaussian it brings back that the count kernel kept none of
// (a depth-cut frame's k_project_hist only writes the records of Gaussians that keep a pair): the same inputs through
// the same functions in the same translation unit -- project_one, make_raster_record -- as that kernel.
struct RegenProject {
    const float *means3d, *scales, *quats, *opacities, *viewmat;
    const void *colors;
    int color_f16;
    ms::ProjParams P;
    float4 *rec;
    Grid g;
    Candidates cand;   // a pre-culled band: positions of its candidate list (ids: null otherwise)
};
constexpr int kRegenThreads = 512, kRegenMaxTiles = 65536;   // (512: a band's candidate map wants a thread per segment)
template <int PASS>
__global__ __launch_bounds__(kRegenThreads) void k_far_regen(ms::LazyLists Z, int tw, int n_tiles, int64_t cap, RegenProject R, int lds_tiles) {
    const int n_redo = min(*Z.redo_count, n_tiles);
    if (n_redo <= 0) return;
    __shared__ uint32_t s_bits[kRegenMaxTiles / 32];
    __shared__ uint32_t s_wtot[kRegenThreads / 64];
    __shared__ int s_any;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < kRegenMaxTiles / 32; i += kRegenThreads) s_bits[i] = 0u;
    if (tid == 0) s_any = 0;
    __syncthreads();
    auto marked = [&](int ri) {
        const int tile = ri < n_redo ? Z.redo_list[ri] : -1;
        return (tile >= 0 && tile < n_tiles && Z.has_far[tile] == Z.cut_stamp) ? tile : -1;
    };
    for (int ri = tid; ri < n_redo; ri += kRegenThreads) {
        const int tile = marked(ri);
        if (tile >= 0) {
            atomicOr(&s_bits[tile >> 5], 1u << (tile & 31));
            s_any = 1;
        }
    }
    __syncthreads();
    if (!s_any) return;
    if constexpr (PASS == 1) {
        // far_start[tile] = the dropped pairs of the marked tiles before it on the redo list (the same values in every workgroup)
        uint32_t running = 0;
        for (int base = 0; base < n_redo; base += kRegenThreads) {
            const int tile = marked(base + tid);
            const uint32_t c = tile >= 0 ? Z.far_cnt[tile] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= d) incl += o;
            }
            if (lane == 63) s_wtot[w] = incl;
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int ww = 0; ww < kRegenThreads / 64; ++ww) {
                if (ww < w) before += s_wtot[ww];
                total += s_wtot[ww];
            }
            if (tile >= 0) Z.far_start[tile] = running + before + incl - c;
            running += total;
            __syncthreads();
        }
        // (the count kernel counted every pair it dropped: what comes back cannot exceed that)
        if (blockIdx.x == 0 && tid == 0 && (int64_t)running > Z.cut_words[1])
            atomicAdd(reinterpret_cast<unsigned long long *>(Z.cut_words + 2), 1ull);
        __threadfence_block();
        __syncthreads();
    }
    const int64_t near = Z.cut_words[0];
    __shared__ uint32_t s_pref[kMaxG + 1];
    CandMap cmap{s_pref};
    int64_t n_items = Z.n_lean;
    if (R.cand.ids) {   // (uniform)
        cmap.build(R.cand);
        n_items = (int64_t)s_pref[R.cand.n_segs];
    }
    // the pairs a Gaussian has behind the cut-offs of the marked bins: fn(bin) for each, -> whether there was one
    // (box, reach mask and depth bits as the count kernel had them -- it keeps no box records on a depth-cut frame:
    // the same functions on the same inputs, in the same translation unit)
    struct Seen { ms::ProjOut o; uint32_t db, mk, src; int x0, y0, bw, bh; };
    auto visit = [&](int64_t j, Seen &v, auto &&fn) __attribute__((always_inline)) {
        v.db = 0u; v.mk = 0u; v.x0 = 0; v.y0 = 0; v.bw = 0; v.bh = 0;
        // (a pre-culled band: j is a POSITION in its candidate list -- what the lists, the keys and the records are indexed by)
        v.src = R.cand.ids ? (uint32_t)cmap.gaussian(R.cand, j) : (uint32_t)j;
        v.o = ms::project_one<uint32_t>(v.src, R.means3d, R.scales, R.quats, R.opacities, R.viewmat, R.P);
        if (v.o.r0 > 0 && v.o.r1 > 0) {
            int bx1, by1, be;
            (void)bin_box<false>(make_float2(v.o.m0, v.o.m1), make_int2(v.o.r0, v.o.r1), R.g, v.x0, bx1, v.y0, by1, be);
            v.bw = bx1 - v.x0; v.bh = by1 - v.y0;
            if (v.bw * v.bh > 0) {
                v.mk = v.bw * v.bh > 32 ? 0xffffffffu
                                        : (uint32_t)reach_mask(v.o.m0, v.o.m1, v.o.c0, v.o.c1, v.o.c2, ms::ld_f32(R.opacities, v.src, 1, 0), v.x0, bx1, v.y0, by1, R.g.ts);
                v.db = __float_as_uint(v.o.d);
            }
        }
        const int n = v.bw * v.bh;
        bool any = false;
        for (int r = 0, k = 0; r < v.bh; ++r) {
            for (int c = 0; c < v.bw; ++c, ++k) {
                if (n <= 32 && !((v.mk >> k) & 1u)) continue;   // (the count / scatter kernels' rule: walk_boxes)
                const int tile = (v.y0 + r) * tw + v.x0 + c;
                if (!((s_bits[tile >> 5] >> (tile & 31)) & 1u) || v.db <= Z.tau[tile]) continue;
                fn(tile);
                any = true;
            }
        }
        return any;
    };
    // Round 4: the counters of a grid of up to kRegenLdsTiles bins are kept in LDS, a workgroup adding its sums to the global
    // ones once per bin: every pair's own global atomic -- six million of them on the ~300 words of a config-4 frame's
    // stranded bins, all in one L2 channel -- made each of the two launches 6-7 ms.
    extern __shared__ uint32_t s_dyn[];   // lds_tiles > 0: [n_tiles] pair counts / cursors (+ [n_tiles] segment bases, PASS 1)
    const bool in_lds = lds_tiles > 0;
    const int64_t j0 = (int64_t)blockIdx.x * kRegenThreads + tid, jstep = (int64_t)gridDim.x * kRegenThreads;
    Seen v;
    if (in_lds) {
        for (int t = tid; t < n_tiles; t += kRegenThreads) s_dyn[t] = 0u;
        __syncthreads();
        for (int64_t j = j0; j < n_items; j += jstep) (void)visit(j, v, [&](int tile) { atomicAdd(&s_dyn[tile], 1u); });
        __syncthreads();
        if constexpr (PASS == 0) {
            for (int t = tid; t < n_tiles; t += kRegenThreads)
                if (s_dyn[t]) atomicAdd(&Z.far_cnt[t], s_dyn[t]);
            return;
        } else {
            uint32_t *s_base = s_dyn + n_tiles;   // this workgroup's stretch of each bin's segment
            for (int t = tid; t < n_tiles; t += kRegenThreads) {
                const uint32_t c = s_dyn[t];
                if (c) { s_base[t] = atomicAdd(&Z.far_cur[t], c); s_dyn[t] = 0u; }
            }
            __syncthreads();
        }
    } else if constexpr (PASS == 0) {
        for (int64_t j = j0; j < n_items; j += jstep) (void)visit(j, v, [&](int tile) { atomicAdd(&Z.far_cnt[tile], 1u); });
        return;
    }
    if constexpr (PASS == 1) {
        const uint32_t *s_base = s_dyn + n_tiles;
        for (int64_t j = j0; j < n_items; j += jstep) {
            const bool brought_back = visit(j, v, [&](int tile) {
                const uint32_t slot = in_lds ? s_base[tile] + atomicAdd(&s_dyn[tile], 1u) : atomicAdd(&Z.far_cur[tile], 1u);
                const int64_t pos = near + (int64_t)Z.far_start[tile] + (int64_t)slot;
                if (pos < cap) Z.log_keys[pos] = ((uint64_t)v.db << 32) | (uint32_t)j;
                // (cannot happen while this pass and the count kernel agree on every Gaussian's depth bits, box and
                // reach mask -- they are the same inlined functions, but two instantiations: the frame's record
                // says so if they ever do not, instead of a pair silently lost: cut_words[2], reported one frame
                // later in bit 62 of host_info[5])
                else atomicAdd(reinterpret_cast<unsigned long long *>(Z.cut_words + 2), 1ull);
            });
            const int n = v.bw * v.bh;
            if (brought_back && n <= kCoopThreshold && R.rec) {
                // did the count kernel keep a pair of this Gaussian (then it wrote the record itself)?
                bool kept = false;
                for (int r = 0, k = 0; r < v.bh; ++r)
                    for (int c = 0; c < v.bw; ++c, ++k)
                        if (((v.mk >> k) & 1u) && v.db <= Z.tau[(v.y0 + r) * tw + v.x0 + c]) kept = true;
                if (!kept) {
                    const uint32_t src = v.src;
                    float col[3];
                    if (R.color_f16) {
                        const __half *cp = reinterpret_cast<const __half *>(R.colors);
                        col[0] = __half2float(cp[3 * src]); col[1] = __half2float(cp[3 * src + 1]); col[2] = __half2float(cp[3 * src + 2]);
                    } else {
                        const ms::F3 c3 = ms::ld_f32x3(reinterpret_cast<const float *>(R.colors), src);
                        col[0] = c3.x; col[1] = c3.y; col[2] = c3.z;
                    }
                    const ms::RasterRecord rr = ms::make_raster_record(v.o.m0, v.o.m1, v.o.c0, v.o.c1, v.o.c2, ms::ld_f32(R.opacities, src, 1, 0), col[0], col[1], col[2]);
                    float4 *out = R.rec + 3 * j;
                    out[0] = rr.a; out[1] = rr.b; out[2] = rr.c;
                }
            }
        }
    }
}

constexpr int kRegenLdsTiles = 16384;   // (2 x 64 KB of counters next to the 8 KB of marks)

}  // namespace

// The two launches (rasterize.hip runs them between a depth-cut frame's rasteriser and its clean-up kernel).
int ms::far_regen(const ms::LazyLists &lazy, int tw, int n_tiles, int64_t cap, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MS_REQUIRE(lazy.cut_inputs && n_tiles <= kRegenMaxTiles, MS_ERR_INVALID_ARG, "far_regen: a depth-cut frame without its inputs");
    const ms::CutInputs &I = *lazy.cut_inputs;
    const int th_ = n_tiles / (tw > 0 ? tw : 1);
    MS_REQUIRE(I.row_begin >= 0 && I.row_begin <= I.row_end && I.row_end <= th_, MS_ERR_INVALID_ARG, "far_regen: bad band");
    Candidates cand{nullptr, nullptr, 0, 0};
    if (I.band_cull) {   // the band's candidate list, where the frame's count launch left it
        Plan p;
        MS_REQUIRE(I.isect_workspace && make_plan(I.N, tw, th_, I.row_begin, I.row_end, p), MS_ERR_INVALID_ARG, "far_regen: a pre-culled band without its workspace");
        const char *ws = (const char *)I.isect_workspace;
        cand = Candidates{(const int32_t *)(ws + p.off_cand), (const int32_t *)(ws + p.off_cand_count), p.G, p.chunk};
    }
    const RegenProject R{I.means3d, I.scales, I.quats, I.opacities, I.viewmat, I.colors, I.color_f16,
                         ms::make_proj_params(I.fx, I.fy, I.cx, I.cy, I.W, I.H, I.eps2d, I.near_plane, I.far_plane, 0.0f, I.scales_are_log,
                                              I.opacities != nullptr),
                         (float4 *)I.records,
                         // (the band's rows: the count kernel clamped every box to them, and the reach masks' bits count from there)
                         Grid{I.tile_size, tw, th_, I.row_begin, I.row_end, 0, 0, 0}, cand};
    // (empty launches on almost every frame -- every workgroup reads the redo count and leaves; the frame costs the same
    // whatever their number: pipeline.hip, ms_redo_grid)
    const unsigned grid = 512u;
    const int lds_tiles = n_tiles <= kRegenLdsTiles ? n_tiles : 0;
    if (lds_tiles > 4096) {   // (more than the 64 KB a launch may take unasked, next to the kernel's 8 KB of marks)
        static std::mutex mu;
        static std::vector<int> devices_set;
        int dev = -1;
        MS_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        if (std::find(devices_set.begin(), devices_set.end(), dev) == devices_set.end()) {
            MS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_far_regen<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kRegenLdsTiles * 4));
            MS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_far_regen<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kRegenLdsTiles * 8));
            devices_set.push_back(dev);
        }
    }
    hipLaunchKernelGGL(k_far_regen<0>, dim3(grid), dim3(kRegenThreads), (size_t)lds_tiles * 4, stream, lazy, tw, n_tiles, cap, R, lds_tiles);
    hipLaunchKernelGGL(k_far_regen<1>, dim3(grid), dim3(kRegenThreads), (size_t)lds_tiles * 8, stream, lazy, tw, n_tiles, cap, R, lds_tiles);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// The band's tiles, heaviest list first, as the count pass leaves them (order[0 .. band tiles), absolute tile ids).
const int32_t *ms::isect_order_array(const void *workspace, int64_t N, int tile_w, int tile_h) {
    Plan p;
    make_plan(N, tile_w, tile_h, 0, tile_h, p);
    return (const int32_t *)((const char *)workspace + p.off_order);
}

#ifdef MS_DIAG
extern "C" int ms_diag_set_bin_stamps(void *device_buffer) {
    MS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_diag_bin), &device_buffer, sizeof(void *)));
    return MS_OK;
}
#endif

// ---- prepared scenes -------------------------------------------------------------------------------------------------
// Bounds of every block of `block_size` consecutive Gaussians: f32[n_blocks][8] = min x y z of the means, the largest
// LINEAR scale | max x y z, 0.  One wave per block.  (For a scene whose caller has stored it in a spatially coherent
// order -- scene_order.prepare_scene sorts along a Morton curve -- so that the boxes are small; any order is correct.)
namespace {
__global__ __launch_bounds__(256) void k_block_bounds(int64_t N, const float *__restrict__ means3d, const float *__restrict__ scales,
                                                      int scales_are_log, int block_size, float *__restrict__ out,
                                                      float4 *__restrict__ cull) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t i0 = b * block_size, i1 = min(N, i0 + block_size);
    if (i0 >= N) return;   // (uniform per wave)
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, sm = 0.f;
    for (int64_t i = i0 + lane; i < i1; i += 64) {
        float p[3], raw = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p[k] = means3d[3 * i + k];
            lo[k] = fminf(lo[k], p[k]); hi[k] = fmaxf(hi[k], p[k]);
            float sc = scales[3 * i + k];
            raw = fmaxf(raw, sc);
            if (scales_are_log) sc = expf(sc);
            sm = fmaxf(sm, sc);
        }
        // the band pre-cull's own record of the Gaussian: its mean and its largest LINEAR scale, the very value that pass
        // computes from the scales (k_band_precull) -- 16 bytes in one load instead of 24 in two
        cull[i] = make_float4(p[0], p[1], p[2], scales_are_log ? __expf(raw) : raw);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], d));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d));
        }
        sm = fmaxf(sm, __shfl_xor(sm, d));
    }
    if (lane == 0) {
        float *q = out + 8 * b;
        q[0] = lo[0]; q[1] = lo[1]; q[2] = lo[2]; q[3] = sm * 1.00001f;   // (expf vs the kernels' __expf: an ulp of slack)
        q[4] = hi[0]; q[5] = hi[1]; q[6] = hi[2]; q[7] = 0.f;
    }
}
}  // namespace

// (the buffer: f32[n_blocks][8] bounds, then f32[N][4] pre-cull records -- mean, largest linear scale -- of the Gaussians)
extern "C" size_t ms_scene_block_bounds_bytes(int64_t N, int block_size) {
    if (N <= 0 || block_size < 64 || (block_size & (block_size - 1))) return 0;
    return (size_t)ms::ceil_div(N, block_size) * 8 * sizeof(float) + (size_t)N * 4 * sizeof(float);
}

extern "C" int ms_scene_prepare(int64_t N, const float *means3d, const float *scales, int scales_are_log, int block_size,
                                float *block_bounds, void *stream) {
    MS_REQUIRE(N >= 0 && block_size >= 64 && (block_size & (block_size - 1)) == 0, MS_ERR_INVALID_ARG,
               "scene_prepare: the block size must be a power of two >= 64");
    if (N == 0) return MS_OK;
    MS_REQUIRE(means3d && scales && block_bounds, MS_ERR_INVALID_ARG, "scene_prepare: null pointer");
    const int64_t nb = ms::ceil_div(N, block_size);
    MS_REQUIRE(nb <= 0x7fffffffll, MS_ERR_TOO_LARGE, "scene_prepare: too many blocks");
    hipLaunchKernelGGL(k_block_bounds, dim3((unsigned)ms::ceil_div(nb, 4)), dim3(256), 0, (hipStream_t)stream, N, means3d, scales,
                       scales_are_log, block_size, block_bounds, reinterpret_cast<float4 *>(block_bounds + 8 * nb));
    MS_LAUNCH_CHECK();
    return MS_OK;
}

extern "C" size_t ms_isect_workspace_bytes(int64_t N, int tile_w, int tile_h) {
    Plan p;
    if (tile_w <= 0 || tile_h <= 0 || (int64_t)tile_w * tile_h >= (1ll << 30)) return 0;
    make_plan(N, tile_w, tile_h, 0, tile_h, p);
    return p.total;
}

extern "C" int ms_isect_tiles_count(int64_t N, const float *means2d, const int32_t *radii,
                                    int tile_size, int tile_w, int tile_h, int row_begin,
                                    int row_end, void *workspace, size_t workspace_bytes,
                                    int32_t *tiles_per_gauss, int32_t *tile_ranges,
                                    int64_t *isect_info, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MS_REQUIRE(N >= 0, MS_ERR_INVALID_ARG, "isect_count: N < 0");
    if (int rc = check_grid(tile_size, tile_w, tile_h, row_begin, row_end)) return rc;
    MS_REQUIRE(workspace && tile_ranges && isect_info && (N == 0 || (means2d && radii)),
               MS_ERR_INVALID_ARG, "isect_count: null pointer");
    MS_REQUIRE(((uintptr_t)means2d & 7) == 0 && ((uintptr_t)radii & 7) == 0 && ((uintptr_t)tile_ranges & 7) == 0,
               MS_ERR_INVALID_ARG, "isect_count: means2d/radii/tile_ranges must be 8-byte aligned");
    Plan p;
    const bool fits = make_plan(N, tile_w, tile_h, row_begin, row_end, p);
    MS_REQUIRE(fits, MS_ERR_TOO_LARGE,
               "isect_count: band of %d tiles needs %zu B of LDS (max %zu); bin in row bands",
               p.T_local, p.lds_bytes, kMaxLds);
    MS_REQUIRE(workspace_bytes >= p.total, MS_ERR_WORKSPACE, "isect_count: workspace %zu < %zu",
               workspace_bytes, p.total);
    char *ws = (char *)workspace;
    uint32_t *hist = (uint32_t *)(ws + p.off_hist);
    uint32_t *count = (uint32_t *)(ws + p.off_count);
    int32_t *medium = (int32_t *)(ws + p.off_medium);
    int32_t *large = (int32_t *)(ws + p.off_large), *xl = (int32_t *)(ws + p.off_xl);
    uint32_t *on_grid = (uint32_t *)(ws + p.off_on_grid);
    const Grid g{tile_size, tile_w, tile_h, row_begin, row_end, 0, 0, 0};

    if (p.T_local > 0) {
        if (p.lds_bytes > 48 * 1024)
            if (int rc = allow_big_lds(k_isect_hist)) return rc;
        hipLaunchKernelGGL(k_isect_hist, dim3(p.G), dim3(kHistThreads), p.lds_bytes, stream, N, means2d,
                           radii, g, p.chunk, hist, tiles_per_gauss, on_grid);
        MS_LAUNCH_CHECK();
    } else if (N > 0) {  // empty band: nothing to histogram, but the frame-level on-grid count is still owed
        hipLaunchKernelGGL(k_isect_hist, dim3(p.G), dim3(kHistThreads), p.lds_bytes, stream, N, means2d, radii, g,
                           p.chunk, hist, tiles_per_gauss, on_grid);
        MS_LAUNCH_CHECK();
    }
    return count_tail(p, g, ws, hist, count, medium, large, xl, on_grid, N > 0 ? p.G : 0, tile_ranges, isect_info,
                      /*band_only=*/0, nullptr, stream);
}

// Projection + tile counting in one pass over the Gaussians (what a frame starts with): the
// outputs of ms_project_gaussians_fwd AND of ms_isect_tiles_count, without re-reading the
// projected means / radii and with one kernel launch fewer.
extern "C" int ms_project_isect_count(int64_t N, const float *means3d, const float *scales, int scales_are_log,
                                      const float *quats, const float *opacities, const float *viewmat,
                                      float fx, float fy, float cx, float cy, int W, int H, float eps2d,
                                      float near_plane, float far_plane, float radius_clip, int tile_size,
                                      int row_begin, int row_end, int tight, float *means2d, float *conics,
                                      float *depths,
                                      int32_t *radii, void *workspace, size_t workspace_bytes,
                                      int32_t *tile_ranges, int64_t *isect_info, int64_t *isect_info_mirror,
                                      void *stream_) {
    return ms::project_isect_count(N, means3d, scales, scales_are_log, quats, opacities, viewmat, fx, fy, cx, cy, W, H,
                                   eps2d, near_plane, far_plane, radius_clip, tile_size, row_begin, row_end, tight & 63,
                                   means2d, conics, depths, radii, workspace, workspace_bytes, tile_ranges, isect_info,
                                   isect_info_mirror, nullptr, 0, nullptr, stream_, 0u);
}

int ms::project_isect_count(int64_t N, const float *means3d, const float *scales, int scales_are_log,
                            const float *quats, const float *opacities, const float *viewmat,
                            float fx, float fy, float cx, float cy, int W, int H, float eps2d,
                            float near_plane, float far_plane, float radius_clip, int tile_size,
                            int row_begin, int row_end, int tight, float *means2d, float *conics,
                            float *depths,
                            int32_t *radii, void *workspace, size_t workspace_bytes,
                            int32_t *tile_ranges, int64_t *isect_info, int64_t *isect_info_mirror,
                            const void *colors3, int color_dtype, void *raster_records, void *stream_, uint32_t cut_stamp,
                            const float *block_bounds, int block_size) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!opacities || !colors3) raster_records = nullptr;
    int block_shift = 0;
    if (block_bounds) {
        MS_REQUIRE(block_size >= 64 && (block_size & (block_size - 1)) == 0, MS_ERR_INVALID_ARG,
                   "project_isect_count: a prepared scene's block size must be a power of two >= 64 (got %d)", block_size);
        while ((1 << block_shift) < block_size) ++block_shift;
    }
    MS_REQUIRE(N >= 0 && W > 0 && H > 0 && tile_size > 0 && fx != 0.f && fy != 0.f, MS_ERR_INVALID_ARG,
               "project_isect_count: bad sizes / camera");
    const int tile_w = (W + tile_size - 1) / tile_size, tile_h = (H + tile_size - 1) / tile_size;
    if (int rc = check_grid(tile_size, tile_w, tile_h, row_begin, row_end)) return rc;
    MS_REQUIRE(workspace && tile_ranges && isect_info &&
                   (N == 0 || (means3d && scales && quats && viewmat && means2d && conics && depths && radii)),
               MS_ERR_INVALID_ARG, "project_isect_count: null pointer");
    MS_REQUIRE(((uintptr_t)quats & 15) == 0 && ((uintptr_t)means2d & 7) == 0 && ((uintptr_t)radii & 7) == 0 &&
                   ((uintptr_t)tile_ranges & 7) == 0,
               MS_ERR_INVALID_ARG, "project_isect_count: quats 16-byte, means2d/radii/tile_ranges 8-byte aligned");
    Plan p;
    const bool fits = make_plan(N, tile_w, tile_h, row_begin, row_end, p);
    MS_REQUIRE(fits, MS_ERR_TOO_LARGE, "project_isect_count: band of %d tiles needs %zu B of LDS (max %zu)",
               p.T_local, p.lds_bytes, kMaxLds);
    MS_REQUIRE(workspace_bytes >= p.total, MS_ERR_WORKSPACE, "project_isect_count: workspace %zu < %zu",
               workspace_bytes, p.total);
    char *ws = (char *)workspace;
    uint32_t *hist = (uint32_t *)(ws + p.off_hist);
    uint32_t *count = (uint32_t *)(ws + p.off_count);
    int32_t *medium = (int32_t *)(ws + p.off_medium);
    int32_t *large = (int32_t *)(ws + p.off_large), *xl = (int32_t *)(ws + p.off_xl);
    uint32_t *on_grid = (uint32_t *)(ws + p.off_on_grid);
    // block masks (bit 2): entries carry the 2x2 half-tile blocks they reach; bits 3 / 4: the half-tile
    // grid has an odd number of columns / rows (2 tile_w - 1, 2 tile_h - 1)
    const bool pack = (tight & 4) && (tight & 1) && opacities && (tile_size & 1) == 0 && N < (1ll << 28);
    const Grid g{tile_size, tile_w, tile_h, row_begin, row_end, pack ? 1 : 0, 2 * tile_w - ((tight >> 3) & 1),
                 2 * tile_h - ((tight >> 4) & 1)};
    unsigned long long *masks = ((tight & 1) && opacities) ? (unsigned long long *)(ws + p.off_mask) : nullptr;
    const ms::ProjParams P = ms::make_proj_params(fx, fy, cx, cy, W, H, eps2d, near_plane, far_plane, radius_clip,
                                                  scales_are_log, opacities != nullptr);
    // bit 5 of `tight`: the band is a rank's share of a frame -- pre-cull the Gaussians that cannot reach it
    Candidates cand{nullptr, nullptr, 0, 0};
    if ((tight & kBandCull) && N >= (1ll << 28)) tight &= ~kBandCull;   // (the candidates' gather indexes with 32-bit byte offsets)
    if ((tight & kBandCull) && N > 0) {
        int32_t *seg_count = (int32_t *)(ws + p.off_cand_count);
        hipLaunchKernelGGL(block_bounds ? k_band_precull<true> : k_band_precull<false>, dim3(p.G), dim3(kHistThreads), 0, stream, N,
                           means3d, scales, viewmat, P, (float)(row_begin * tile_size) - 1.0f, (float)(row_end * tile_size) + 1.0f,
                           p.chunk, (int32_t *)(ws + p.off_cand), seg_count, block_bounds, block_shift);
        MS_LAUNCH_CHECK();
        cand = Candidates{(const int32_t *)(ws + p.off_cand), seg_count, p.G, p.chunk};
    }
    // claimed rows (round 6; ms::kTightClaimed, set by ms_render_fwd for sync-free frames whose total pass rides in the scatter
    // launch): no prefix kernel behind this one (profiles/r06_claimed_rows.md)
    static std::atomic<unsigned long long> claim_stamps{0x5ca1ab1e00000000ull};   // (never a value fresh memory is likely to hold)
    const bool claimed = (tight & ms::kTightClaimed) && (tight & kDeferTotal) && p.T_local > 0;
    {   // also for N == 0 (one workgroup that walks nothing): the histogram row and the on-grid slot the
        // scans read must exist
        // lean frame: LeanRecs instead of the projected arrays (the caller vouches that nobody reads those)
        const bool lean = (tight & kLean) && raster_records && masks && tile_w <= 0xffff && tile_h <= 0xffff && N < (1ll << 28);
        const bool lean12 = lean && !pack && tile_w <= 255 && tile_h <= 255;
        auto kernel = pack ? (lean ? k_project_hist<true, 1> : k_project_hist<true, 0>)
                           : (lean12 ? (cut_stamp ? k_project_hist<false, 2, true> : k_project_hist<false, 2>)
                                     : lean ? k_project_hist<false, 1> : k_project_hist<false, 0>);
        // depth-cut frame: per-tile cut-offs beside the counters (the sort kernel of the previous frame left them)
        const bool cut = cut_stamp != 0u;   // (bit 9 of `tight`: which of the two cut-off buffers to read)
        MS_REQUIRE(!cut || (lean12 && (tight & kDeferTotal)), MS_ERR_INVALID_ARG,
                   "project_isect_count: a depth-cut frame must be a lean sync-free frame on plain bins");
        size_t lds = p.lds_bytes + (cut ? (size_t)p.T_local * 5 + 16 : 0);
        MS_REQUIRE(lds <= kMaxLds, MS_ERR_TOO_LARGE, "project_isect_count: %d tiles with cut-offs need %zu B of LDS", p.T_local, lds);
        // the workgroup's queue of big boxes behind everything else, as far as the LDS has room (BigQ: 32-byte entries with
        // 16-bit tile coordinates)
        const unsigned int big_q_off = (unsigned int)ms::align_up(lds, 16);
        const unsigned int big_q_cap = big_queue_cap(big_q_off, tile_w, tile_h);
        lds = (size_t)big_q_off + (size_t)big_q_cap * 32;
        if (lds > 48 * 1024)
            if (int rc = allow_big_lds(kernel)) return rc;
        hipLaunchKernelGGL(kernel, dim3(p.G), dim3(kHistThreads), lds, stream, N, means3d, scales,
                           quats, opacities, viewmat, P, g, p.chunk, means2d, conics, depths, radii, hist, on_grid, masks,
                           colors3, color_dtype == MS_COLOR_F16 ? 1 : 0, (float4 *)raster_records, cand,
                           (LeanRec *)(ws + p.off_lean), (uint32_t *)(ws + p.off_depth_wg),
                           cut ? (const uint32_t *)(ws + p.off_tau) + (size_t)((tight >> 9) & 1) * p.T : nullptr, (uint32_t *)(ws + p.off_wg_far),
                           (uint32_t *)(ws + p.off_has_far), cut_stamp, (LeanRec *)(ws + p.off_near), (tight & ms::kTightKeepArrays) ? 1 : 0, p.near_cap,
                           big_q_off, big_q_cap, claimed ? (uint32_t *)(ws + p.off_xtot) : nullptr,
                           // (the flag: two words of the 256-byte block the clean-up counts start -- words 0-2 -- 8-byte aligned)
                           (unsigned long long *)(ws + p.off_redo_count + 64), claimed ? ++claim_stamps : 0ull);
        MS_LAUNCH_CHECK();
    }
    if (claimed) return MS_OK;   // (the rows are exclusive offsets already, `count` the tile counts: no prefix kernel)
    return count_tail(p, g, ws, hist, count, medium, large, xl, on_grid, p.G, tile_ranges, isect_info,
                      (tight & 2) ? 1 : 0, isect_info_mirror, stream, (tight & kDeferTotal) != 0);
}

namespace {
// Shared by the exact emit (host knows M and the class counts) and the speculative one (it does
// not: `info_dev` is the count pass's device record, `cap` the capacity of the key/id buffers).
int emit_impl(int64_t N, const float *means2d, const int32_t *radii, const float *depths, int tight, int lazy,
              float depth_near, float depth_far, int tile_size,
              int tile_w, int tile_h, int row_begin, int row_end, void *workspace, size_t workspace_bytes,
              const int32_t *tile_ranges, const int64_t *host_info, const int64_t *info_dev, int64_t cap,
              uint64_t *sort_keys, uint64_t *sort_tmp, int32_t *flatten_ids, int64_t *isect_ids,
              hipStream_t stream, const ms::BlockLists *blocks = nullptr, const ms::DeferredTotal *defer = nullptr) {
    const ms::BlockLists bl = blocks ? *blocks : ms::BlockLists{nullptr, nullptr, nullptr, 0, 0};
    const uint32_t cut_stamp = defer ? defer->cut_stamp : 0u;   // != 0: a depth-cut frame (k_project_hist)
    // bits 4-5 of `lazy` (ms_render_fwd only): the merged sort launch leaves the NEXT frame's cut-offs in buffer (lazy >> 4) & 1;
    // a depth-cut frame reads its own from the other one
    const int tau_out = (lazy >> 4) & 1;
    const bool write_tau = (lazy & 32) != 0;
    Plan p;
    const bool fits = make_plan(N, tile_w, tile_h, row_begin, row_end, p);
    MS_REQUIRE(fits, MS_ERR_TOO_LARGE, "isect_emit: band too large for LDS");
    MS_REQUIRE(workspace_bytes >= p.total, MS_ERR_WORKSPACE, "isect_emit: workspace %zu < %zu",
               workspace_bytes, p.total);
    char *ws = (char *)workspace;
    const uint32_t *hist = (const uint32_t *)(ws + p.off_hist);
    const int32_t *medium = (const int32_t *)(ws + p.off_medium);
    const int32_t *large = (const int32_t *)(ws + p.off_large), *xl = (const int32_t *)(ws + p.off_xl);
    const bool pack = (tight & 4) && (tight & 1) && (tile_size & 1) == 0 && N < (1ll << 28);
    const Grid g{tile_size, tile_w, tile_h, row_begin, row_end, pack ? 1 : 0, 2 * tile_w - ((tight >> 3) & 1),
                 2 * tile_h - ((tight >> 4) & 1)};
    const unsigned long long *masks = (tight & 1) ? (const unsigned long long *)(ws + p.off_mask) : nullptr;
    const bool spec = info_dev != nullptr;
    const int64_t max_count = spec ? 0 : host_info[1], n_medium = spec ? 0 : host_info[2],
                  n_large = spec ? 0 : host_info[3], n_xl = spec ? 0 : host_info[4];

    {
        const bool lean = (tight & kLean) && masks && tile_w <= 0xffff && tile_h <= 0xffff && N < (1ll << 28);
        const bool lean12 = lean && !pack && tile_w <= 255 && tile_h <= 255;
        const bool deferred = defer && p.T_local > 0;
        ScanTotalArgs A{};
        // (claimed rows: the count pass of this frame claimed its rows per XCD -- the cursors of either scatter form add the
        // stretches of the XCDs before the workgroup's own)
        const uint32_t *xtot = (tight & ms::kTightClaimed) && p.T_local > 0 ? (const uint32_t *)(ws + p.off_xtot) : nullptr;
        A.xtot = xtot;
        if (deferred) {
            A = ScanTotalArgs{g, (const uint32_t *)(ws + p.off_count), const_cast<int32_t *>(tile_ranges), (int32_t *)(ws + p.off_medium),
                              (int32_t *)(ws + p.off_large), (int32_t *)(ws + p.off_xl), (const uint32_t *)(ws + p.off_on_grid), p.G,
                              (int32_t *)(ws + p.off_redo_flag), (int32_t *)(ws + p.off_redo_count), defer->band_only,
                              defer->info, defer->info_mirror, (int32_t *)(ws + p.off_order), nullptr,
                              nullptr, nullptr, 0u, nullptr, 0, nullptr, xtot};
            if (defer->cut_stamp) {
                MS_REQUIRE(lean12 && lazy, MS_ERR_INVALID_ARG, "isect emit: a depth-cut frame must be a lean, lazily sorted frame on plain bins");
                A.wg_far = (const uint32_t *)(ws + p.off_wg_far);
                A.tau = (const uint32_t *)(ws + p.off_tau) + (size_t)(1 - tau_out) * p.T;
                A.cut_stamp = defer->cut_stamp;
                A.far_zero = (uint32_t *)(ws + p.off_far_seg);
                A.near_recs = (const LeanRec *)(ws + p.off_near);
                A.near_cap = p.near_cap;
            }
        }
        size_t scatter_lds = p.lds_bytes + (A.cut_stamp ? (size_t)p.T_local * 4 + 16 : 0);
        const unsigned int big_q_off = (unsigned int)ms::align_up(scatter_lds, 16);
        const unsigned int big_q_cap = (lean ? big_queue_cap(big_q_off, tile_w, tile_h) : 0u);   // (the record walks queue; the array walk does not)
        scatter_lds = (size_t)big_q_off + (size_t)big_q_cap * 32;
        auto pick = [&](auto packc, auto leanc) {
            constexpr bool PK = decltype(packc)::value;
            constexpr int LN = decltype(leanc)::value;
            return deferred ? k_isect_scatter<PK, LN, true> : k_isect_scatter<PK, LN, false>;
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        auto kernel = pack ? (lean ? pick(std::true_type{}, I1{}) : pick(std::true_type{}, I0{}))
                           : (lean12 ? pick(std::false_type{}, I2{}) : lean ? pick(std::false_type{}, I1{}) : pick(std::false_type{}, I0{}));
        if (scatter_lds > 48 * 1024)
            if (int rc = allow_big_lds(kernel)) return rc;
        hipLaunchKernelGGL(kernel, dim3(p.G + (deferred ? 2 : 0)), dim3(kHistThreads), scatter_lds, stream, N, means2d, radii, depths, masks,
                           (const LeanRec *)(ws + p.off_lean), g, p.chunk, hist, tile_ranges, cap, sort_keys,
                           // (a lean frame's count kernel has left the per-workgroup depth ranges already)
                           lazy && !lean ? (uint32_t *)(ws + p.off_depth_wg) : nullptr,
                           (tight & kBandCull) ? Candidates{(const int32_t *)(ws + p.off_cand), (const int32_t *)(ws + p.off_cand_count), p.G, p.chunk}
                                               : Candidates{nullptr, nullptr, 0, 0}, p.G, A, big_q_off, big_q_cap);
        MS_LAUNCH_CHECK();
        // the size record is complete once the scatter launch is: the caller's hand-off event goes here
        if (deferred && defer->sync_event) MS_HIP(hipEventRecord((hipEvent_t)defer->sync_event, stream));
    }

    static_assert(kSmallCap == kSmallCapDecl && kMediumCap == kMediumCapDecl && kLargeCap == kLargeCapDecl,
                  "sort class thresholds out of sync");
    // Split frames: the small-class kernel goes FIRST and writes empty block ranges for every bin it does not
    // sort itself; the front kernel, launched after it, overwrites those of the heavy bins it reaches.  The
    // rasteriser must never walk unwritten ranges, whatever a sync-free frame turns out to hold.
    if (bl.block_ids && p.T_local > 0) {
        hipLaunchKernelGGL(k_tile_sort_small<true>, dim3((unsigned)p.T_local), dim3(256), 0, stream, tile_ranges,
                           sort_keys, flatten_ids, isect_ids, cap, row_begin * tile_w, bl, tile_w);
        MS_LAUNCH_CHECK();
    }
    if (lazy) {
        // heavy tiles: sorted front only (k_tile_front); everything else as usual
        static_assert(kFrontK == kSmallCapDecl, "the front kernel takes over exactly where the small class ends");
        // plain bins: one merged launch over the band's tiles, heaviest first (MOJOSPLAT_MERGED_SORT=0: the two
        // launches, for measurements)
        static const bool merged_env = [] { const char *e = getenv("MOJOSPLAT_MERGED_SORT"); return !e || atoi(e) != 0; }();
        // no tile beyond the small class (known on the exact path; the caller's bet on a speculative frame,
        // bit 3 of `lazy`): the 256-thread small-sort launch alone -- 7.8 us at config 2 against 14.6 for the merged
        // launch, whose 512-thread workgroups are sized for the fronts
        const bool no_heavy = !bl.block_ids && (spec ? (lazy & 8) != 0 : n_medium + n_large + n_xl == 0);
        const bool merged = merged_env && !bl.block_ids && p.T_local > 0 && !no_heavy;
        if (!no_heavy && (merged || spec || n_medium + n_large + n_xl > 0)) {
            auto front = bl.block_ids ? k_tile_front<true, false> : merged ? k_tile_front<false, true> : k_tile_front<false, false>;
            if (int rc = allow_big_lds(front)) return rc;
            const int64_t heavy = n_medium + n_large + n_xl;
            // (sync-free frame: the previous frame's heavy-tile count sizes the launch; the kernel strides over
            // the list, so a low guess only costs parallelism)
            int64_t guess = 1024;
            if (spec && host_info) {
                const int64_t prev_heavy = host_info[2] + host_info[3] + host_info[4];
                guess = prev_heavy > 0 ? prev_heavy + prev_heavy / 4 + 16 : 1024;
            }
            const unsigned grid = spec ? (unsigned)min((int64_t)min(p.T, 1024), guess)
                                       : (unsigned)(heavy < 1024 ? heavy : 1024);
            const ms::FrontParams fp = ms::front_params(tile_size, lazy, depth_near, depth_far, merged, bl.block_ids != nullptr);
            const uint32_t fixed_min = fp.fixed_min;
            const int fixed_shift = fp.fixed_shift, front_k = fp.front_k, front_cap = fp.front_cap;
            size_t front_lds = kFrontLds - (size_t)(kFrontCap - front_cap) * 8;
            // (merged launch: room for eight waves' private sorts of light lists)
            const bool light = merged;
            if (light && front_lds < 8 * SortCfg<64, kLightCap / 64>::LDS) front_lds = 8 * SortCfg<64, kLightCap / 64>::LDS;
            hipLaunchKernelGGL(front, dim3(merged ? (unsigned)p.T_local : grid), dim3(kFrontThreads), front_lds, stream,
                               medium, large, xl,
                               spec ? info_dev : nullptr, (int)n_medium, (int)n_large, (int)n_xl, tile_ranges,
                               sort_keys, flatten_ids, (int32_t *)(ws + p.off_front), cap, fixed_min, fixed_shift,
                               front_k, bl, tile_w, (const uint32_t *)(ws + p.off_depth_wg), p.G,
                               (const int32_t *)(ws + p.off_order), p.T_local, front_cap,
                               light ? (const uint32_t *)(ws + p.off_on_grid) + kMaxG + 1 : nullptr,
                               merged && write_tau ? (uint32_t *)(ws + p.off_tau) + (size_t)tau_out * p.T : nullptr,
                               (const uint32_t *)(ws + p.off_tau) + (size_t)(1 - tau_out) * p.T, (const uint32_t *)(ws + p.off_has_far), cut_stamp);
            MS_LAUNCH_CHECK();
        }
        if (merged) return MS_OK;
        if (p.T_local > 0 && !bl.block_ids)
            hipLaunchKernelGGL(k_tile_sort_small<false>, dim3((unsigned)p.T_local), dim3(256), 0, stream, tile_ranges,
                               sort_keys, flatten_ids, isect_ids, cap, row_begin * tile_w, bl, tile_w);
        MS_LAUNCH_CHECK();
        return MS_OK;
    }
    // (Running the size classes on two streams was measured: the fork/join events cost more
    // than the overlap buys -- bin stage 176 us forked vs 160 us in order.)
    hipStream_t big = stream;
    const size_t medium_lds = SortCfg<1024, 8>::LDS, large_lds = SortCfg<1024, 16>::LDS;
    if (spec || n_medium > 0) {
        if (int rc = allow_big_lds(k_tile_sort_list<8>)) return rc;
        const unsigned grid = spec ? (unsigned)min(p.T, 2048) : (unsigned)n_medium;
        hipLaunchKernelGGL(k_tile_sort_list<8>, dim3(grid), dim3(1024), medium_lds, big, medium, tile_ranges,
                           sort_keys, flatten_ids, isect_ids, spec ? info_dev + 2 : nullptr, (int)n_medium, cap);
        MS_LAUNCH_CHECK();
    }
    // sync-free frames cannot know whether the large class is populated; the previous frame's
    // record (still in the caller's host_info when it is passed) says whether to bother: a wrong
    // guess is caught by the caller's post-check like any other speculation failure
    const bool launch_large = spec ? (host_info == nullptr || host_info[3] > 0) : n_large > 0;
    if (launch_large) {
        if (int rc = allow_big_lds(k_tile_sort_list<16>)) return rc;
        const unsigned grid = spec ? 256u : (unsigned)n_large;
        hipLaunchKernelGGL(k_tile_sort_list<16>, dim3(grid), dim3(1024), large_lds, big, large, tile_ranges,
                           sort_keys, flatten_ids, isect_ids, spec ? info_dev + 3 : nullptr, (int)n_large, cap);
        MS_LAUNCH_CHECK();
    }
    if (p.T_local > 0)
        hipLaunchKernelGGL(k_tile_sort_small<false>, dim3((unsigned)p.T_local), dim3(256), 0, stream, tile_ranges,
                           sort_keys, flatten_ids, isect_ids, cap, row_begin * tile_w, bl, tile_w);
    MS_LAUNCH_CHECK();
    if (n_xl > 0) {  // exact mode only: a speculative frame with XL tiles is redone by the caller
        if (int rc = allow_big_lds(k_xl_chunk_sort)) return rc;
        const unsigned chunks = (unsigned)ms::ceil_div(max_count, kLargeCap);
        hipLaunchKernelGGL(k_xl_chunk_sort, dim3(chunks, (unsigned)n_xl), dim3(1024), large_lds, stream,
                           xl, tile_ranges, sort_keys);
        MS_LAUNCH_CHECK();
        uint64_t *src = sort_keys, *dst = sort_tmp;
        const unsigned eblocks = (unsigned)ms::ceil_div(max_count, 256);
        for (int64_t L = kLargeCap; L < max_count; L <<= 1) {
            hipLaunchKernelGGL(k_xl_merge, dim3(eblocks, (unsigned)n_xl), dim3(256), 0, stream, xl,
                               tile_ranges, src, dst, (int)L);
            MS_LAUNCH_CHECK();
            uint64_t *t = src; src = dst; dst = t;
        }
        hipLaunchKernelGGL(k_xl_extract, dim3(eblocks, (unsigned)n_xl), dim3(256), 0, stream, xl,
                           tile_ranges, src, flatten_ids, isect_ids);
        MS_LAUNCH_CHECK();
    }
    return MS_OK;
}
}  // namespace

extern "C" int ms_isect_tiles_emit(int64_t N, const float *means2d, const int32_t *radii,
                                   const float *depths, int tile_size, int tile_w, int tile_h,
                                   int row_begin, int row_end, void *workspace,
                                   size_t workspace_bytes, const int32_t *tile_ranges,
                                   const int64_t *host_info, int tight, int lazy, float depth_near,
                                   float depth_far, uint64_t *sort_keys,
                                   uint64_t *sort_tmp, int32_t *flatten_ids, int64_t *isect_ids,
                                   void *stream_) {
    MS_REQUIRE(N >= 0 && host_info, MS_ERR_INVALID_ARG, "isect_emit: bad N / host_info");
    if (int rc = check_grid(tile_size, tile_w, tile_h, row_begin, row_end)) return rc;
    const int64_t M = host_info[0], n_xl = host_info[4];
    MS_REQUIRE(M >= 0 && M <= 0x7fffffffll, MS_ERR_TOO_LARGE,
               "isect_emit: %lld intersections do not fit int32 indices", (long long)M);
    if (M == 0) return MS_OK;
    MS_REQUIRE(workspace && tile_ranges && means2d && radii && depths && sort_keys && flatten_ids,
               MS_ERR_INVALID_ARG, "isect_emit: null pointer");
    MS_REQUIRE(n_xl == 0 || sort_tmp || lazy, MS_ERR_INVALID_ARG, "isect_emit: sort_tmp required (XL tiles)");
    MS_REQUIRE(!lazy || !isect_ids, MS_ERR_INVALID_ARG, "isect_emit: lazy lists carry no isect_ids");
    return emit_impl(N, means2d, radii, depths, tight & 63, lazy & 15, depth_near, depth_far, tile_size, tile_w, tile_h,
                     row_begin, row_end, workspace, workspace_bytes, tile_ranges, host_info, nullptr, M, sort_keys, sort_tmp,
                     flatten_ids, isect_ids, (hipStream_t)stream_);
}

extern "C" int ms_isect_tiles_emit_speculative(int64_t N, const float *means2d, const int32_t *radii,
                                               const float *depths, int tile_size, int tile_w, int tile_h,
                                               int row_begin, int row_end, void *workspace,
                                               size_t workspace_bytes, const int32_t *tile_ranges,
                                               const int64_t *isect_info_dev, int64_t capacity,
                                               const int64_t *prev_info_host, int tight, int lazy,
                                               float depth_near, float depth_far, uint64_t *sort_keys,
                                               int32_t *flatten_ids, void *stream_) {
    MS_REQUIRE(N >= 0 && isect_info_dev && capacity > 0 && capacity <= 0x7fffffffll, MS_ERR_INVALID_ARG,
               "isect_emit_speculative: bad N / info / capacity");
    if (int rc = check_grid(tile_size, tile_w, tile_h, row_begin, row_end)) return rc;
    MS_REQUIRE(workspace && tile_ranges && means2d && radii && depths && sort_keys && flatten_ids,
               MS_ERR_INVALID_ARG, "isect_emit_speculative: null pointer");
    return emit_impl(N, means2d, radii, depths, tight & 63, lazy & 15, depth_near, depth_far, tile_size, tile_w, tile_h,
                     row_begin, row_end, workspace, workspace_bytes, tile_ranges, prev_info_host, isect_info_dev, capacity,
                     sort_keys, nullptr, flatten_ids, nullptr, (hipStream_t)stream_);
}

int ms::isect_emit_speculative(int64_t N, const float *means2d, const int32_t *radii, const float *depths,
                               int tile_size, int tile_w, int tile_h, int row_begin, int row_end, void *workspace,
                               size_t workspace_bytes, const int32_t *tile_ranges, const int64_t *isect_info_dev,
                               int64_t capacity, const int64_t *prev_info_host, int tight, int lazy, float depth_near,
                               float depth_far, uint64_t *sort_keys, int32_t *flatten_ids,
                               const ms::DeferredTotal *defer, void *stream_) {
    MS_REQUIRE(N >= 0 && isect_info_dev && capacity > 0 && capacity <= 0x7fffffffll, MS_ERR_INVALID_ARG,
               "isect_emit_speculative: bad N / info / capacity");
    if (int rc = check_grid(tile_size, tile_w, tile_h, row_begin, row_end)) return rc;
    MS_REQUIRE(workspace && tile_ranges && means2d && radii && depths && sort_keys && flatten_ids,
               MS_ERR_INVALID_ARG, "isect_emit_speculative: null pointer");
    return emit_impl(N, means2d, radii, depths, tight, lazy, depth_near, depth_far, tile_size, tile_w, tile_h,
                     row_begin, row_end, workspace, workspace_bytes, tile_ranges, prev_info_host, isect_info_dev, capacity,
                     sort_keys, nullptr, flatten_ids, nullptr, (hipStream_t)stream_, nullptr, defer);
}

int ms::isect_emit_exact(int64_t N, const float *means2d, const int32_t *radii, const float *depths, int tile_size,
                         int tile_w, int tile_h, int row_begin, int row_end, void *workspace, size_t workspace_bytes,
                         const int32_t *tile_ranges, const int64_t *host_info, int tight, int lazy, float depth_near,
                         float depth_far, uint64_t *sort_keys, uint64_t *sort_tmp, int32_t *flatten_ids, void *stream_) {
    MS_REQUIRE(N >= 0 && host_info, MS_ERR_INVALID_ARG, "isect_emit: bad N / host_info");
    if (int rc = check_grid(tile_size, tile_w, tile_h, row_begin, row_end)) return rc;
    const int64_t M = host_info[0], n_xl = host_info[4];
    MS_REQUIRE(M >= 0 && M <= 0x7fffffffll, MS_ERR_TOO_LARGE,
               "isect_emit: %lld intersections do not fit int32 indices", (long long)M);
    if (M == 0) return MS_OK;
    MS_REQUIRE(workspace && tile_ranges && means2d && radii && depths && sort_keys && flatten_ids,
               MS_ERR_INVALID_ARG, "isect_emit: null pointer");
    MS_REQUIRE(n_xl == 0 || sort_tmp || lazy, MS_ERR_INVALID_ARG, "isect_emit: sort_tmp required (XL tiles)");
    return emit_impl(N, means2d, radii, depths, tight, lazy, depth_near, depth_far, tile_size, tile_w, tile_h,
                     row_begin, row_end, workspace, workspace_bytes, tile_ranges, host_info, nullptr, M, sort_keys, sort_tmp,
                     flatten_ids, nullptr, (hipStream_t)stream_);
}

int ms::isect_emit_bins(int64_t N, const float *means2d, const int32_t *radii, const float *depths, int bin_w,
                        int bin_h, int row_begin, int row_end, void *workspace, size_t workspace_bytes,
                        const int32_t *bin_ranges, const int64_t *host_info, const int64_t *info_dev, int64_t cap,
                        int flags, int lazy, float depth_near, float depth_far, uint64_t *sort_keys,
                        const ms::BlockLists *out, const ms::DeferredTotal *defer, void *stream_) {
    MS_REQUIRE(N > 0 && cap > 0 && cap <= 0x1fffffffll && (info_dev || host_info) && (lazy & 1), MS_ERR_INVALID_ARG,
               "isect_emit_bins: bad N / capacity / info (block lists come from lazily sorted bins only)");
    if (int rc = check_grid(32, bin_w, bin_h, row_begin, row_end)) return rc;
    MS_REQUIRE(workspace && bin_ranges && means2d && radii && depths && sort_keys && out && out->block_ranges &&
                   out->block_ids && out->bin_more,
               MS_ERR_INVALID_ARG, "isect_emit_bins: null pointer");
    return emit_impl(N, means2d, radii, depths, flags, lazy | 1, depth_near, depth_far, 32, bin_w, bin_h, row_begin,
                     row_end, workspace, workspace_bytes, bin_ranges, host_info, info_dev, cap, sort_keys, nullptr,
                     nullptr, nullptr, (hipStream_t)stream_, out, info_dev ? defer : nullptr);
}

extern "C" int ms_isect_offset_encode(int64_t M, const int64_t *isect_ids_sorted, int tile_w,
                                      int tile_h, int32_t *offsets, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MS_REQUIRE(M >= 0 && tile_w > 0 && tile_h > 0 && offsets, MS_ERR_INVALID_ARG,
               "offset_encode: bad argument");
    MS_REQUIRE(M <= 0x7fffffffll, MS_ERR_TOO_LARGE, "offset_encode: M does not fit int32");
    const int T = tile_w * tile_h;
    if (M == 0) {
        MS_HIP(hipMemsetAsync(offsets, 0, (size_t)T * 4, stream));
        return MS_OK;
    }
    MS_REQUIRE(isect_ids_sorted, MS_ERR_INVALID_ARG, "offset_encode: null ids");
    hipLaunchKernelGGL(k_offset_encode, dim3((unsigned)ms::ceil_div(M, 256)), dim3(256), 0, stream, M,
                       isect_ids_sorted, T, offsets);
    MS_LAUNCH_CHECK();
    return MS_OK;
}
