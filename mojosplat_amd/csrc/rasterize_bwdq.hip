// Tile rasteriser backward, round 4: one wave64 per 8x8 QUAD, walking FRONT TO BACK, per-entry sums on the matrix pipe.
//
// No reference counterpart (mojosplat/render.py:11 `@torch.no_grad()`, README.md:145); gsplat's backward semantics as in
// rasterize_bwd.hip, whose kernels stay behind the per-stage entry point.  This kernel is what ms_render_bwd runs on a
// 3-channel differentiable frame; it needs the frame's image next to its alphas, and no last_ids.
//
// Why another shape (profiles/r04_bwd_pmc.md): k_rasterize_bwd_v2 issues 191 M vector instructions per config-3 frame for
// 1.3 M walked (block, entry) pairs -- ~50 per (quad, entry) evaluation plus ~45 for the wave reduction of nine sums
// and ~20 of walk -- at 3.9 waves per SIMD: 281 us; and it walks by POSITION from last_ids back, so its forward had to
// sort every list in full (250 us of forward against the inference frame's 165).  Here
//   * the walk is the FORWARD kernel's (rasterize.hip, raster_tile): per-quad compacted record streams in LDS, two records
//     per trip, exp2 with log2(opacity) in the FMA chain, the alpha >= 1/255 select by underflow, the transmittance
//     carried as S = T * 2^126, one stop test per pair.  It walks whatever lists the forward walked -- lazily sorted
//     fronts on the binning rule's grid included -- so a differentiable frame IS an inference frame that keeps its alphas.
//     Walking front to back needs no division to recover T and no per-entry index compare; what lies BEHIND an entry
//     comes from the frame's own output:
//         sum_{s > t} Cd_s w_s = v . (C - T_final bg) - sum_{s <= t} Cd_s w_s,      Cd_s = colour_s . v   (v = dL/dC of the pixel)
//     so that   dL/dalpha_t (1 - alpha_t) = Cd_t T_t + Q_{t-1},   Q_t = T_final (v_a - bg . v) - v . (C - T_final bg) + sum_{s <= t} Cd_s w_s
//     is one FMA on a running scalar per pixel;
//   * the nine per-entry sums over the quad's 64 pixels are a GEMM: with x, y the pixel's offset from the quad centre,
//         [sum vs, sum vs x, sum vs y, sum vs x^2, sum vs xy, sum vs y^2] = VS[entry][pixel] . MONO[pixel][6]
//         [sum w v_r, sum w v_g, sum w v_b]                               = W[entry][pixel] . V[pixel][3]
//     (vs = dL/dsigma of the pair, w = alpha T).  Every lane leaves vs and 255 w of eight entries in an LDS tile
//     Y[16 rows][64 pixels], each value as TWO bf16 terms in one dword (hi = the float cut to 8 mantissa bits, lo = the
//     exact remainder cut likewise: 16 significant bits, 2^-17 relative), and four v_mfma_f32_16x16x32_bf16 (K = 64
//     pixels x 2 terms, fp32 accumulation) with the per-lane constants MONO | V_hi | V_lo as the B operand turn the tile
//     into the 8 x 9 sums.  MONO is exact in bf16 (half-integers up to 3.5 and their products); dL/dC is split like the A
//     values.  [The first cut used v_mfma_f32_16x16x4_f32 -- exact, sixteen per tile: 307 us.  An fp32 MFMA runs at the
//     vector rate and, measured, does not overlap the vector pipe's work here; the bf16 form costs a quarter of the
//     issue time and six vector instructions per evaluation for the split: 274 us.]
//   * eight lanes move the moments from the quad centre to the Gaussian's mean (dx = u - x, u = mean - centre), and the
//     totals leave as ONE 64-byte row per (quad, entry): contiguous float atomics, two instructions per eight entries
//     (2.5 M rows at config 3; the memory side takes 20.8 G rows / s whatever their fill: scripts/ubench/atomic_grad_rows.hip).
//     The backward projection finishes means / conics / opacity from the raw sums (project_bwd.hip, ROWS = 2).
#include <stdlib.h>

#include <type_traits>

#include "ms_common.hpp"

namespace {

#ifndef MS_BWDQ_BATCH
#define MS_BWDQ_BATCH 64
#endif
constexpr int kBatch = MS_BWDQ_BATCH;   // list entries staged per round (<= 64: one per lane)
constexpr int kGroup = 2;      // records per trip of the walk
constexpr int kTile = 8;       // entries per matrix tile (rows 0-7 of Y: vs, rows 8-15: 255 w)
constexpr int kYStride = 68;   // floats per row of Y: the A fragments are read as ds_read_b128 at (row, 16 g + 4 i)
constexpr int kRowQ = 16;      // floats per packed gradient row: gx gy s1 s2 s3 m0 c0 c1 c2 - ...
constexpr int kSlots = kTile + kBatch + kGroup;
#ifndef MS_BWDQ_ABLATE
#define MS_BWDQ_ABLATE 0
#endif
#ifndef MS_BWDQ_PREFETCH
#define MS_BWDQ_PREFETCH 0
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x as two bf16 terms, hi in the low half and lo in the high half of a dword: hi = x with its low 16 mantissa bits cut,
// lo = (x - hi) cut likewise (x - hi is exact); hi + lo carries 16 significant bits of x
__device__ __forceinline__ unsigned split_bf16(float x) {
    const unsigned xb = __float_as_uint(x);
    const float lo = x - __uint_as_float(xb & 0xffff0000u);
    return __builtin_amdgcn_perm(__float_as_uint(lo), xb, 0x07060302u);
}

struct BwdQArgs {
    const float4 *records;          // ms::RasterRecord per Gaussian
    const int32_t *tile_ranges;     // [tiles][2]
    const int32_t *ids;             // the sorted lists: Gaussian index per entry (id_stride = 1), or the low words of sorted keys (2)
    int id_stride;
    const int32_t *front_count;     // lazily sorted frame: entries of a heavy tile's sorted front (or null)
    int front_threshold;
    const int32_t *skip_flag;       // tiles the forward's clean-up pass redid: left to the fallback launch (or null)
    const float *render_colors, *render_alphas, *v_render_colors, *v_render_alphas, *backgrounds;
    float *packed;                  // f32[N][kRowQ], zeroed by the caller
    const int32_t *order;           // the binning grid's tiles, heaviest first (or null: image order)
    int W, H, ts, tw, nsx, nsub, ntiles, ngrid, max_isects, n_gauss;
    int tile0;                      // first tile of the band the launch covers (image-order launches; an order lists its own tiles)
    // Round 5: the forward's per-quad lists (rasterize.hip, RasterArgs::quad_lists) -- the Gaussians that passed quad qd's reach
    // test, in list order, up to the batch in which the quad's last pixel stopped: the walk stages 64 of THEM a round, no
    // fetch of the tile's other entries, no test, no compaction.  Null: the tile's list, tested per quad as the forward did.
    const int32_t *quad_lists, *quad_counts;
    int quad_nq;
};

struct BwdQStage {
    float4 a[kSlots];               // mean.x, mean.y, a', b'
    float4 b[kSlots];               // c', log2(opacity), r, g
    float2 c[kSlots];               // b, Gaussian index (bits)
    __attribute__((aligned(16))) float y[16 * kYStride];
    __attribute__((aligned(16))) float e[kTile * 16];
#ifdef MS_BWDQ_LDS_PAD   // (measurement builds: extra LDS per wave, to hold fewer waves per CU)
    float pad[MS_BWDQ_LDS_PAD];
#endif
};

// Every wave is a workgroup of its own and the LDS executes one wave's instructions in order: what a lane stores is there
// for any lane's later load without a wait.  What is needed is that the COMPILER keeps the order (it reasons per lane and
// would move a lane's load above another lane's store): a compiler barrier, no s_waitcnt (MS_BWDQ_FENCE=1: the fences of
// the forward kernel's wave_lds_sync, for comparison).
__device__ __forceinline__ void wave_lds_sync_q() {
#if defined(MS_BWDQ_FENCE) && MS_BWDQ_FENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#endif
}

// One wave: quad `q` (0..3) of 16x16 block `sub` of tile `tile`.
template <bool LISTS = false>
__device__ __forceinline__ void bwd_quad(const BwdQArgs &A, const int tile, const int sub, const int q, BwdQStage &S) {
    constexpr float kInf = __builtin_huge_valf();
    const int lane = threadIdx.x & 63;
    const int lx = lane & 7, ly = lane >> 3;
    const int tile_y = tile / A.tw, tile_x = tile - tile_y * A.tw;
    const int sub_y = sub / A.nsx, sub_x = sub - sub_y * A.nsx;
    const int qx0 = tile_x * A.ts + sub_x * 16 + (q & 1) * 8, qy0 = tile_y * A.ts + sub_y * 16 + (q >> 1) * 8;
    if (qx0 >= A.W || qy0 >= A.H) return;
    const int ox = sub_x * 16 + (q & 1) * 8 + lx, oy = sub_y * 16 + (q >> 1) * 8 + ly;
    const int X = qx0 + lx, Y = qy0 + ly;
    const bool in = ox < A.ts && oy < A.ts && X < A.W && Y < A.H;
    const float px = (float)X + 0.5f, py = (float)Y + 0.5f;
    const float qcx = (float)qx0 + 4.0f, qcy = (float)qy0 + 4.0f;   // the quad's centre; pixel centres sit at -3.5 .. 3.5 from it

    const int end_all = min(A.tile_ranges[2 * tile + 1], A.max_isects);
    const int start = min(A.tile_ranges[2 * tile], end_all);
    int end = end_all;
    if (A.front_count && end_all - start > A.front_threshold) end = start + min(A.front_count[tile], end_all - start);
    if (end <= start) return;

    // ---- the pixel: dL/dC, what the frame left of it, the running scalar Q
    float vo[3] = {0.f, 0.f, 0.f}, Q = 0.f;
    {
        const size_t p = in ? (size_t)Y * A.W + X : 0;
        float Tf = 1.0f, va = 0.f, C[3] = {0.f, 0.f, 0.f};
        if (in) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { vo[k] = A.v_render_colors[p * 3 + k]; C[k] = A.render_colors[p * 3 + k]; }
            Tf = 1.0f - A.render_alphas[p];
            if (A.v_render_alphas) va = A.v_render_alphas[p];
        }
        float bg_dot = 0.f, ptot = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float bg = A.backgrounds ? A.backgrounds[k] : 0.f;
            bg_dot += bg * vo[k];
            ptot += vo[k] * (C[k] - Tf * bg);
        }
        Q = 255.0f * (Tf * (va - bg_dot) - ptot);    // carried times 255, like the blend weights v = 255 alpha T
    }
    float kq = in ? ms::kFlushK : 0.f;
    float t = ms::kTScale;

    // ---- B operand of the four matrix instructions.  K = 128 = 64 pixels x 2 bf16 terms (hi, lo of the A value), pixel
    // p = 16 i + 4 g + m at k = 8 g + 2 m + term of instruction i; lane l = (g = l >> 4, column j = l & 15) holds, per
    // instruction, four dwords: (F(p, hi) | F(p, lo) << 16), m = 0..3, with
    //   j 0-5 : MONO_j(p) for both terms (exact in bf16: half-integers up to 3.5 and their products)
    //   j 6-8 : hi(dL/dC / 255) for both terms            -> sum (w_hi + w_lo) v_hi
    //   j 9-11: lo(dL/dC / 255) for the hi term, 0 for lo -> sum w_hi v_lo      (colour gradient = column 6+c plus 9+c)
    // Round 5: built through a table in LDS -- every lane writes the thirteen packed words of ITS pixel (row j of Y, column
    // p = lane; row 12 = the zeros of columns 12-15) and then reads the four pixels of each of its (instruction, k-group)
    // cells as ONE ds_read_b128 of row j: ~50 instructions and one LDS round trip per wave.  Round 4 built the sixteen
    // words by a per-lane `if (j == ...)` chain, which the compiler turned into sixteen copies of a divergent branch
    // tree with an LDS wait in each: ~1 500 instructions per wave, a quarter of everything the kernel issued at config 3
    // (32 640 waves of ~6 200 instructions each; profiles/r05_bwd_quads.md).
    unsigned Bd[4][4];
    {
        unsigned *tab = reinterpret_cast<unsigned *>(S.y);   // 16 rows of kYStride words: free until the first tile is stored
        const float x = (float)lx - 3.5f, y = (float)ly - 3.5f;
        auto both = [](float f) { const unsigned b = __float_as_uint(f); return (b >> 16) | (b & 0xffff0000u); };
        tab[0 * kYStride + lane] = both(1.0f);
        tab[1 * kYStride + lane] = both(x);
        tab[2 * kYStride + lane] = both(y);
        tab[3 * kYStride + lane] = both(x * x);
        tab[4 * kYStride + lane] = both(x * y);
        tab[5 * kYStride + lane] = both(y * y);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float c = vo[k] * ms::kInv255;
            const unsigned c_hi = __float_as_uint(c) & 0xffff0000u;
            tab[(6 + k) * kYStride + lane] = (c_hi >> 16) | c_hi;                                     // hi(dL/dC) for both terms
            tab[(9 + k) * kYStride + lane] = __float_as_uint(c - __uint_as_float(c_hi)) >> 16;       // lo(dL/dC) for the hi term
        }
        tab[12 * kYStride + lane] = 0u;
        wave_lds_sync_q();
        const int g = lane >> 4, j = min(lane & 15, 12);
        const u32x4 *row = reinterpret_cast<const u32x4 *>(tab + j * kYStride + 4 * g);   // (16-byte aligned: kYStride % 4 == 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 v = row[4 * i];   // pixels 16 i + 4 g + (0..3)
            Bd[i][0] = v[0]; Bd[i][1] = v[1]; Bd[i][2] = v[2]; Bd[i][3] = v[3];
        }
        wave_lds_sync_q();
    }

    __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);   // fp32 denormals flushed: the walk selects by underflow (rasterize.hip)

    // ---- staging registers (as the forward's PACKED path)
    // Loads are UNCONDITIONAL on a clamped index (lanes past the list's end fetch its last entry and are ignored at
    // staging): an exec-masked load has to merge with the register's old value, and the compiler did that with a copy
    // right behind the load -- every batch waited out its gather's full latency there.
    float4 r_a, r_b, r_c;
    int id_next, g_staged;   // the list word fetched for the NEXT gather; the Gaussian index that belongs to r_a / r_b / r_c
    auto fetch_id = [&](int b0) {
        const int idx = min(b0 + lane, end - 1);
        id_next = A.ids[(size_t)idx * A.id_stride];
    };
    auto gather = [&]() {
        // ids outside [0, N) can only come from a frame that overflowed its buffer (it is redone): never out of bounds
        g_staged = min(max(id_next, 0), A.n_gauss - 1);
        const float4 *rec = A.records + 3 * (size_t)g_staged;
        r_a = rec[0]; r_b = rec[1]; r_c = rec[2];
    };

    const float xl0 = qcx - 3.5f, yl0 = qcy - 3.5f;   // the quad's first pixel centre
    int fill = 0;                                      // entries of the current matrix tile already evaluated (even)
    const int y_lane = lane * 4;                       // byte offset of this pixel's column in a row of Y
    const int a_frag = ((lane & 15) * kYStride + 4 * (lane >> 4)) * 4;   // + 64 i bytes for instruction i

    // the matrix tile -> sums -> rows.  n: entries of the tile (<= kTile), tbase: stream slot of the tile's entry 0
    auto fire = [&](const int n, const int tbase) __attribute__((always_inline)) {
#if MS_BWDQ_ABLATE & 2   // (measurement builds: the walk alone)
        if (n > 64) S.e[0] = 0.f;
        return;
#endif
        wave_lds_sync_q();
        const char *yb = reinterpret_cast<const char *>(S.y) + a_frag;
        u32x4 af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const u32x4 *>(yb + 64 * i);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const u32x4 b0 = {Bd[i][0], Bd[i][1], Bd[i][2], Bd[i][3]}, b1 = {Bd[i + 1][0], Bd[i + 1][1], Bd[i + 1][2], Bd[i + 1][3]};
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, b0), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i + 1]), __builtin_bit_cast(bf16x8, b1), acc1, 0, 0, 0);
        }
        const f32x4 D = acc0 + acc1;
        // D[row = 4 (lane >> 4) + r][col = lane & 15]: rows 0-7 hold the entries' moment sums in columns 0-5, rows 8-15
        // the same entries' colour sums in columns 6-8 (+ their low-order parts in 9-11)
        {
            const int g = lane >> 4, j = lane & 15;
            const bool keep = g < 2 ? j < 6 : (j >= 6 && j < 12);
            if (keep) {
                float *eb = S.e + (4 * (g & 1)) * 16 + j;
#pragma unroll
                for (int r = 0; r < 4; ++r) eb[r * 16] = D[r];
            }
        }
        wave_lds_sync_q();
#if MS_BWDQ_ABLATE & 4   // (measurement builds: the walk and the matrix instructions, no moment shift, no rows)
        return;
#endif
        // eight lanes move the entries' moments from the quad centre to the Gaussians' means (dx = u - x, dy = v - y) ...
        if (lane < n) {
            const float4 e0 = *reinterpret_cast<const float4 *>(S.e + lane * 16);
            const float4 e1 = *reinterpret_cast<const float4 *>(S.e + lane * 16 + 4);
            const float4 e2 = *reinterpret_cast<const float4 *>(S.e + lane * 16 + 8);
            const float c0 = e1.z + e2.y, c1 = e1.w + e2.z, c2 = e2.x + e2.w;
            const float4 ra = S.a[tbase + lane];
            const float gid = S.c[tbase + lane].y;
            const float u = ra.x - qcx, v = ra.y - qcy;
            const float m0 = e0.x, mx = e0.y, my = e0.z, mxx = e0.w, mxy = e1.x, myy = e1.y;
            const float gx = fmaf(u, m0, -mx), gy = fmaf(v, m0, -my);
            const float s1 = fmaf(u, fmaf(u, m0, -2.0f * mx), mxx);
            const float s2 = fmaf(u, fmaf(v, m0, -my), fmaf(-v, mx, mxy));
            const float s3 = fmaf(v, fmaf(v, m0, -2.0f * my), myy);
            // (an entry that no pixel of the quad blended -- and the neutral record that pads a stream -- adds nothing)
            const bool nz = m0 != 0.f || c0 != 0.f || c1 != 0.f || c2 != 0.f;
            *reinterpret_cast<float4 *>(S.e + lane * 16) = make_float4(gx, gy, s1, s2);
            *reinterpret_cast<float4 *>(S.e + lane * 16 + 4) = make_float4(s3, m0, c0, c1);
            *reinterpret_cast<float4 *>(S.e + lane * 16 + 8) = make_float4(c2, 0.f, 0.f, 0.f);
            S.e[lane * 16 + 12] = gid;
            S.e[lane * 16 + 13] = nz ? 1.0f : 0.f;
        }
        wave_lds_sync_q();
        // ... and the rows leave four at a time: 36-byte pieces of four 64-byte rows per atomic instruction
        {
            const int col = lane & 15, rs = lane >> 4;
#pragma unroll
            for (int i = 0; i < kTile / 4; ++i) {
                const int slot = 4 * i + rs;
                if (slot < n && col < 9) {
                    const float *row = S.e + slot * 16;
                    if (row[13] != 0.f) {
                        const unsigned gi = (unsigned)__float_as_int(row[12]);
#if MS_BWDQ_ABLATE & 1   // (measurement builds: no global atomics)
                        if (row[col] == 1.2345678e-30f) A.packed[(size_t)gi * kRowQ + col] = 1.f;
#else
                        atomicAdd(reinterpret_cast<float *>(reinterpret_cast<char *>(A.packed) + (gi * (kRowQ * 4u) + (unsigned)col * 4u)), row[col]);
#endif
                    }
                }
            }
        }
        // (the next fire's stores into e come behind these loads in the LDS queue; Y is free since the fragments were read)
        wave_lds_sync_q();
    };

    // LISTS: the walk's bounds become [0, count) of the quad's own list, whose entries are Gaussian indices
    const int32_t *qlist = nullptr;
    int walk_lo = start, walk_hi = end;
    if (LISTS) {
        const int qd = sub * 4 + q;
        qlist = A.quad_lists + ((size_t)start * A.quad_nq + (size_t)qd * (size_t)(end_all - start));
        walk_lo = 0;
        walk_hi = min(max(A.quad_counts[(size_t)tile * A.quad_nq + qd], 0), end_all - start);
        if (walk_hi <= 0) return;
    }
    auto fetch_next = [&](int b0) __attribute__((always_inline)) {
        if constexpr (LISTS) id_next = qlist[min(b0 + lane, walk_hi - 1)];
        else fetch_id(b0);
    };
    fetch_next(walk_lo);
    gather();
    fetch_next(walk_lo + kBatch);
    bool live = true;
    for (int b0 = walk_lo; b0 < walk_hi && live; b0 += kBatch) {
        // --- the quad test of the forward kernel (exact ellipse-vs-rectangle, in log2 units on the record)
        bool reach = false, npd = false;
        if constexpr (LISTS) {   // (every entry of the quad's list passed it in the forward)
            reach = b0 + lane < walk_hi && (kBatch == 64 || lane < kBatch);
            npd = reach && r_c.y == kInf;
        } else
        if (b0 + lane < end && (kBatch == 64 || lane < kBatch)) {
            const float smax = r_c.y, nb_c = r_c.z, nb_a = r_c.w;
            if (smax == kInf) {
                reach = true;
                npd = true;
            } else if (smax > -kInf) {
                const float xl = xl0 - r_a.x, xh = xl + 7.0f;
                const float yl = yl0 - r_a.y, yh = yl + 7.0f;
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
                reach = best <= smax;
            }
        }
        const unsigned long long B = __ballot(reach);
        const int n = __popcll(B);
        const bool check_sigma = __ballot(npd) != 0;
        const int tbase0 = kTile - fill;   // stream slot of the open tile's entry 0: the carried entries sit right below kTile

        wave_lds_sync_q();   // the previous batch's reads of the stream are complete
        if (reach) {
            const int pos = kTile + __builtin_amdgcn_mbcnt_hi((unsigned)(B >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)B, 0u));
            S.a[pos] = r_a;
            S.b[pos] = r_b;
            S.c[pos] = make_float2(r_c.x, __int_as_float(g_staged));
        }
        if (lane < kGroup) {
            S.a[kTile + n + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            S.b[kTile + n + lane] = make_float4(0.f, -kInf, 0.f, 0.f);   // alpha = 0: blends nothing, earns nothing
            S.c[kTile + n + lane] = make_float2(0.f, 0.f);
        }
        wave_lds_sync_q();
        if (b0 + kBatch < walk_hi) {
            gather();
            fetch_next(b0 + 2 * kBatch);
        }

        int tbase = tbase0;
        auto walk = [&](auto check) __attribute__((always_inline)) {
            constexpr bool CHECK = decltype(check)::value;
            float4 ra[kGroup], rb[kGroup];
            float blue[kGroup];
            auto load_pair = [&](const int k) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < kGroup; ++j) {
                    ra[j] = S.a[k + j];
                    rb[j] = S.b[k + j];
                    blue[j] = S.c[k + j].x;
                }
            };
#if MS_BWDQ_PREFETCH
            if (n > 0) load_pair(kTile);
#endif
            for (int k0 = kTile; k0 < kTile + n; k0 += kGroup) {
#if !MS_BWDQ_PREFETCH
                load_pair(k0);
#endif
                float alpha[kGroup], m[kGroup], v[kGroup], tin[kGroup], cd[kGroup];
                bool clamped[kGroup];
#pragma unroll
                for (int j = 0; j < kGroup; ++j) {
                    const float dx = ra[j].x - px, dy = ra[j].y - py;
                    const float la = fmaf(dx, fmaf(ra[j].z, dx, ra[j].w * dy), fmaf(rb[j].x * dy, dy, rb[j].y));
                    float al = __builtin_amdgcn_exp2f(la);
                    clamped[j] = false;
                    if constexpr (CHECK) {
                        clamped[j] = al > ms::kMaxAlpha;                   // the 0.999 clamp has no gradient
                        al = fminf(ms::kMaxAlpha, al);
                        al = la <= rb[j].y ? al : 0.f;                     // sigma >= 0
                    }
                    alpha[j] = al;
                    m[j] = al * kq;
                    asm volatile("" : "+v"(m[j]));
                    cd[j] = fmaf(rb[j].z, vo[0], fmaf(rb[j].w, vo[1], blue[j] * vo[2]));
                }
#if MS_BWDQ_PREFETCH
                // the record registers are free: the next pair's loads fly while this pair's chain runs (the slots behind
                // the stream's end hold the neutral records or stale ones: never evaluated)
                load_pair(min(k0 + kGroup, kSlots - kGroup));
#endif
                const float t_in = t;
#pragma unroll
                for (int j = 0; j < kGroup; ++j) {
                    tin[j] = t;
                    v[j] = m[j] * t;
                    t = fmaf(v[j], -ms::kAlphaOfV, t);
                }
                if (__ballot(!(t > ms::kTransmittanceStop * ms::kTScale))) {
                    asm volatile("" ::: "memory");
                    t = t_in;
                    bool dead = false;
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) {
                        const float vj = m[j] * t;
                        const float nt = fmaf(vj, -ms::kAlphaOfV, t);
                        dead = dead || !(nt > ms::kTransmittanceStop * ms::kTScale);
                        tin[j] = t;
                        v[j] = dead ? 0.f : vj;
                        m[j] = dead ? 0.f : m[j];
                        t = dead ? t : nt;
                    }
                    kq = dead ? 0.f : kq;
                }
                // gradients of the pair: vs = dL/dsigma (= -alpha dL/dalpha), 255 w = v
                float vs[kGroup];
#pragma unroll
                for (int j = 0; j < kGroup; ++j) {
                    const float num = fmaf(cd[j], tin[j] * (255.0f * ms::kTUnscale), Q);   // 255 (1 - alpha) dL/dalpha
                    Q = fmaf(cd[j], v[j], Q);
                    const float r1 = __builtin_amdgcn_rcpf(1.0f - alpha[j]);
                    float am = m[j] * (-ms::kAlphaOfV * ms::kInv255);                  // -alpha / 255 (0: not blended)
                    if constexpr (CHECK) am = clamped[j] ? 0.f : am;
                    vs[j] = am * (num * r1);
                }
                // into the matrix tile: rows slot, slot + 1 (vs) and 8 + slot, 9 + slot (255 w), this pixel's column
                {
                    const int slot = k0 - tbase;
                    char *yb = reinterpret_cast<char *>(S.y) + slot * (kYStride * 4) + y_lane;
                    *reinterpret_cast<unsigned *>(yb) = split_bf16(vs[0]);
                    *reinterpret_cast<unsigned *>(yb + kYStride * 4) = split_bf16(vs[1]);
                    *reinterpret_cast<unsigned *>(yb + 8 * kYStride * 4) = split_bf16(v[0]);
                    *reinterpret_cast<unsigned *>(yb + 9 * kYStride * 4) = split_bf16(v[1]);
                }
                if (k0 + kGroup - tbase == kTile) {
                    fire(kTile, tbase);
                    tbase += kTile;
                }
            }
        };
        if (check_sigma) walk(std::true_type{});
        else walk(std::false_type{});
        // what is left of the open tile stays in Y; its entries' records move below kTile for the next batch's numbering
        const int n_pad = (n + kGroup - 1) / kGroup * kGroup;
        fill = kTile + n_pad - tbase;
        live = __any(kq != 0.f);
        if (fill > 0 && live && b0 + kBatch < walk_hi) {
            wave_lds_sync_q();
            float4 ca, cb; float2 cc;
            if (lane < fill) { ca = S.a[tbase + lane]; cb = S.b[tbase + lane]; cc = S.c[tbase + lane]; }
            wave_lds_sync_q();
            if (lane < fill) { S.a[kTile - fill + lane] = ca; S.b[kTile - fill + lane] = cb; S.c[kTile - fill + lane] = cc; }
        } else if (fill > 0) {
            fire(fill, tbase);
            fill = 0;
        }
    }
}

#ifndef MS_BWDQ_WAVES
#define MS_BWDQ_WAVES 5
#endif
template <bool LISTS>
__global__ __launch_bounds__(64, MS_BWDQ_WAVES) void k_rasterize_bwd_quads(BwdQArgs A) {
    __shared__ BwdQStage s_stage;
    // as k_rasterize_fwd with one quad per wave: the four waves of a block sit 8 blockIdx apart (one XCD, one L2), the
    // blocks of a coarse tile back to back on that XCD
    const int j = blockIdx.x >> 3;
    const int part = j & 3;
    const int wg = ((j >> 2) << 3) | (blockIdx.x & 7);
    if (wg >= A.ngrid) return;
    int tile, sub;
    if (A.order) {
        const int e = ((wg >> 3) / A.nsub) * 8 + (wg & 7);
        if (e >= A.ntiles) return;
        tile = A.order[e];
        sub = (wg >> 3) % A.nsub;
    } else {
        if (wg >= A.ntiles * A.nsub) return;
        tile = wg / A.nsub;
        sub = wg - tile * A.nsub;
        tile += A.tile0;
    }
    if (A.skip_flag && A.skip_flag[tile]) return;
    bwd_quad<LISTS>(A, tile, sub, part, s_stage);
}

// ---- the tiles the forward's clean-up pass redid ------------------------------------------------------------------
// A lazily sorted frame's rasteriser stops at the end of a heavy tile's sorted front; a tile whose pixels outlive it
// is redone by the forward's clean-up pass (rasterize.hip) and left out by the launch above (skip_flag).  Round 5: a
// differentiable frame's clean-up pass is the two-launch one -- k_redo_sort sorts every stranded tile's keys WHOLE in LDS
// (bitonic up to 4 096 keys, sample sort beyond) and leaves the Gaussian ids in depth order in the tile's slots of the id
// array, redo_flag[tile] = 2 -- so this launch has nothing to sort: one WAVE per (stranded tile, 16x16 block, quad), dealt
// over the whole grid, walks those ids with the main launch's walk (no front: the whole list, until its pixels are done).
// Round 4 sorted the tile's keys again here, in global memory, one 512-thread workgroup per tile on a grid of 64, and
// walked the tile's blocks one after the other: exact, but unbounded -- a stranded 64-px bin of 100 000 keys is ~300 bitonic
// passes over 800 KB and then 64 serial walks.  That path remains for a tile whose sample sort gave up (redo_flag == 1:
// not seen) and for frames whose forward ran the one-launch clean-up (MOJOSPLAT_REDO_SORT=0).
constexpr int kRedoThreads = 256;

struct BwdRedoArgs {
    BwdQArgs a;            // ids: the frame's id array (stride 1)
    uint64_t *keys;
    const int32_t *redo_list, *redo_count, *redo_flag;
    int32_t *count_mirror;   // null, or a device-visible address of pinned host memory: receives redo_count[0..1] (the frame's caller learns from it)
};

__global__ __launch_bounds__(kRedoThreads) void k_rasterize_bwd_redo(BwdRedoArgs R) {
    __shared__ BwdQStage s_stage[kRedoThreads / 64];
    const int n_redo = *R.redo_count;
    if (R.count_mirror && blockIdx.x == 0 && threadIdx.x == 0) {
        // (a zero-copy store into the caller's pinned memory: visible to the host once an event behind this launch has
        // completed -- round 5's first cut read the two words back with a copy of their own: 4.5 us of copy kernel per step)
        R.count_mirror[0] = n_redo;
        R.count_mirror[1] = R.redo_count[1];
        __threadfence_system();
    }
    if (n_redo <= 0) return;
    const int tid = threadIdx.x, w = tid >> 6;
    // (1) the tiles whose ids are sorted: a wave per (tile, block, quad)
    {
        const int per_tile = R.a.nsub * 4;
        const int64_t total = (int64_t)n_redo * per_tile, stride = (int64_t)gridDim.x * (kRedoThreads / 64);
        for (int64_t item = (int64_t)blockIdx.x * (kRedoThreads / 64) + w; item < total; item += stride) {
            const int ri = (int)(item / per_tile), qi = (int)(item - (int64_t)ri * per_tile);
            const int tile = R.redo_list[ri];
            if (R.redo_flag[tile] != 2) continue;
            bwd_quad(R.a, tile, qi >> 2, qi & 3, s_stage[w]);
        }
    }
    // (2) the others: the tile's (depth bits << 32 | index) keys sorted IN PLACE -- the frame is finished, nothing else
    // reads them; a sorting network whose compare-exchanges all point the same way, so that the virtual +inf padding up to
    // a power of two never moves -- and the indices taken from the keys' low words.  Slow and simple.
    BwdQArgs B = R.a;
    B.ids = reinterpret_cast<const int32_t *>(R.keys);
    B.id_stride = 2;
    for (int ri = blockIdx.x; ri < n_redo; ri += gridDim.x) {
        const int tile = R.redo_list[ri];
        if (R.redo_flag[tile] == 2) continue;   // (uniform)
        const int end_all = min(R.a.tile_ranges[2 * tile + 1], R.a.max_isects);
        const int start = min(R.a.tile_ranges[2 * tile], end_all);
        const int n = end_all - start;
        uint64_t *k = R.keys + start;
        __syncthreads();
        for (int size = 2; (size >> 1) < n; size <<= 1) {
            // first step of a merge: i against its mirror image inside the block of `size`
            for (int t = tid; t < (n + 1) / 2 + size; t += kRedoThreads) {   // (t indexes pairs; the bound only needs to cover them)
                const int blk = t / (size >> 1), off = t - blk * (size >> 1);
                const int i = blk * size + off, j = blk * size + size - 1 - off;
                if (j < n) {
                    const uint64_t a = k[i], b = k[j];
                    if (a > b) { k[i] = b; k[j] = a; }
                }
            }
            __syncthreads();
            for (int stride = size >> 2; stride > 0; stride >>= 1) {
                for (int t = tid; t < (n + 1) / 2 + stride; t += kRedoThreads) {
                    const int i = 2 * stride * (t / stride) + (t % stride), j = i + stride;
                    if (j < n) {
                        const uint64_t a = k[i], b = k[j];
                        if (a > b) { k[i] = b; k[j] = a; }
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();
        for (int qi = w; qi < R.a.nsub * 4; qi += kRedoThreads / 64) bwd_quad(B, tile, qi >> 2, qi & 3, s_stage[w]);
        __syncthreads();
    }
}

}  // namespace

// host side: ms::rasterize_bwd_quads (declared in ms_common.hpp)
int ms::rasterize_bwd_quads(int64_t N, int64_t M, const void *records, const float *backgrounds, int W, int H, int tile_size,
                            const int32_t *tile_ranges, const int32_t *ids, int id_stride, const int32_t *front_count,
                            int front_threshold, const int32_t *skip_flag, const float *render_colors,
                            const float *render_alphas, const float *v_render_colors, const float *v_render_alphas,
                            float *packed_rows, const int32_t *order, void *stream, int tile_row_begin, int tile_row_end,
                            const int32_t *quad_lists, const int32_t *quad_counts) {
    MS_REQUIRE(N > 0 && M > 0 && M <= 0x7fffffffll && N <= 0x3ffffffll, MS_ERR_INVALID_ARG, "rasterize_bwd_quads: bad N/M");   // (rows are addressed by 32-bit byte offsets)
    MS_REQUIRE(W > 0 && H > 0 && tile_size > 0 && tile_size % 16 == 0, MS_ERR_INVALID_ARG,
               "rasterize_bwd_quads: the tile size must be a multiple of 16");
    MS_REQUIRE(records && tile_ranges && ids && render_colors && render_alphas && v_render_colors && packed_rows,
               MS_ERR_INVALID_ARG, "rasterize_bwd_quads: null pointer");
    MS_REQUIRE(((uintptr_t)records & 15) == 0 && ((uintptr_t)packed_rows & 63) == 0, MS_ERR_INVALID_ARG,
               "rasterize_bwd_quads: records must be 16-byte, rows 64-byte aligned");
    BwdQArgs A;
    A.records = (const float4 *)records;
    A.tile_ranges = tile_ranges; A.ids = ids; A.id_stride = id_stride;
    A.front_count = front_count; A.front_threshold = front_threshold; A.skip_flag = skip_flag;
    A.render_colors = render_colors; A.render_alphas = render_alphas; A.v_render_colors = v_render_colors;
    A.v_render_alphas = v_render_alphas; A.backgrounds = backgrounds;
    A.packed = packed_rows; A.order = order;
    A.W = W; A.H = H; A.ts = tile_size;
    A.tw = (W + tile_size - 1) / tile_size;
    const int th = (H + tile_size - 1) / tile_size;
    A.nsx = tile_size / 16;
    A.nsub = A.nsx * A.nsx;
    // a band of tile rows (a multi-GPU rank's differentiable band frame: ranges and order exist for its tiles only)
    if (tile_row_end < 0) tile_row_end = th;
    MS_REQUIRE(tile_row_begin >= 0 && tile_row_begin <= tile_row_end && tile_row_end <= th, MS_ERR_INVALID_ARG,
               "rasterize_bwd_quads: bad tile row band [%d,%d) of %d", tile_row_begin, tile_row_end, th);
    if (tile_row_end == tile_row_begin) return MS_OK;
    A.tile0 = tile_row_begin * A.tw;
    const int64_t tiles = (int64_t)A.tw * (tile_row_end - tile_row_begin);
    MS_REQUIRE(tiles * A.nsub * 4 <= 0x7fffffff, MS_ERR_TOO_LARGE, "rasterize_bwd_quads: too many tiles");
    A.ntiles = (int)tiles;
    A.ngrid = order ? (int)(((tiles + 7) / 8) * 8 * A.nsub) : (int)(tiles * A.nsub);
    A.max_isects = (int)M;
    A.n_gauss = (int)N;
    // workgroup index space: ngrid blocks padded to a multiple of 8, four waves each
    // (the tree variant of this walk -- one wave per 16x16 block, the nine sums by v_permlane32/16_swap + DPP, one row per
    // block -- was built and measured in round 4: exact to 2e-6, 310 us against this kernel's 274; commit 215461d)
    const unsigned grid = (unsigned)(((A.ngrid + 7) / 8) * 8 * 4);
    A.quad_lists = quad_lists && quad_counts ? quad_lists : nullptr;
    A.quad_counts = quad_lists && quad_counts ? quad_counts : nullptr;
    A.quad_nq = 4 * A.nsub;
    if (A.quad_lists) hipLaunchKernelGGL(k_rasterize_bwd_quads<true>, dim3(grid), dim3(64), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(k_rasterize_bwd_quads<false>, dim3(grid), dim3(64), 0, (hipStream_t)stream, A);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

int ms::rasterize_bwd_redo(int64_t N, int64_t M, const void *records, const float *backgrounds, int W, int H, int tile_size,
                           const int32_t *tile_ranges, uint64_t *keys, const int32_t *ids, const int32_t *redo_list,
                           const int32_t *redo_count, const int32_t *redo_flag, const float *render_colors,
                           const float *render_alphas, const float *v_render_colors, const float *v_render_alphas,
                           float *packed_rows, void *stream, int32_t *count_mirror) {
    MS_REQUIRE(N > 0 && M > 0 && M <= 0x7fffffffll && N <= 0x3ffffffll, MS_ERR_INVALID_ARG, "rasterize_bwd_redo: bad N/M");
    MS_REQUIRE(W > 0 && H > 0 && tile_size > 0 && tile_size % 16 == 0, MS_ERR_INVALID_ARG,
               "rasterize_bwd_redo: the tile size must be a multiple of 16");
    MS_REQUIRE(records && tile_ranges && keys && ids && redo_list && redo_count && redo_flag && render_colors && render_alphas &&
                   v_render_colors && packed_rows, MS_ERR_INVALID_ARG, "rasterize_bwd_redo: null pointer");
    BwdRedoArgs R;
    BwdQArgs &A = R.a;
    A.quad_lists = nullptr; A.quad_counts = nullptr; A.quad_nq = 4;   // (a redone tile's lists are the clean-up pass's: walked whole)
    A.records = (const float4 *)records;
    A.tile_ranges = tile_ranges;
    A.ids = ids;
    A.id_stride = 1;
    A.front_count = nullptr; A.front_threshold = 0; A.skip_flag = nullptr;
    A.render_colors = render_colors; A.render_alphas = render_alphas; A.v_render_colors = v_render_colors;
    A.v_render_alphas = v_render_alphas; A.backgrounds = backgrounds;
    A.packed = packed_rows; A.order = nullptr;
    A.W = W; A.H = H; A.ts = tile_size;
    A.tw = (W + tile_size - 1) / tile_size;
    A.nsx = tile_size / 16;
    A.nsub = A.nsx * A.nsx;
    A.ntiles = A.tw * ((H + tile_size - 1) / tile_size);
    A.ngrid = 0;
    A.tile0 = 0;
    A.max_isects = (int)M;
    A.n_gauss = (int)N;
    R.keys = keys; R.redo_list = redo_list; R.redo_count = redo_count; R.redo_flag = redo_flag;
    R.count_mirror = count_mirror;
    // (every workgroup leaves at once on the frames with an empty redo list -- almost all: what the launch costs then is its
    // kernel boundary, whatever the grid)
    hipLaunchKernelGGL(k_rasterize_bwd_redo, dim3(1024), dim3(kRedoThreads), 0, (hipStream_t)stream, R);
    MS_LAUNCH_CHECK();
    return MS_OK;
}
