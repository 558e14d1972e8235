// Per-tile sorting in LDS for the binning kernels (binning.hip): the counting sort of a whole list and the
// selection + sort of a heavy list's front.  Device code only.
#pragma once
#include "ms_common.hpp"

namespace {

// ---- per-tile sort in LDS -------------------------------------------------------------
// Keys are (depth_bits << 32 | gaussian); inside a tile they are distinct.
//
// Main path: ONE counting pass on the depth bits.  (depth_bits - min) >> shift maps the tile's
// depth span order-preservingly onto B ~ n buckets (float bits are monotone in depth for the
// positive depths that survive projection); an LDS histogram + scan + scatter puts every key
// into its bucket, and inside a bucket -- 0..3 keys typically -- every key counts the keys
// smaller than itself (full 64-bit compare) and moves to that rank.  ~6 barriers per tile instead of the O(log^2 n) of a
// bitonic network.  If some bucket is crowded (many identical depths) the tile falls back to
// the bitonic network below, which is oblivious to the key distribution.
template <int THREADS>
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *s, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < (P >> 1); i += THREADS) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo + j;
                const bool up = (lo & k) == 0;
                const uint64_t a = s[lo], b = s[hi];
                if ((a > b) == up) { s[lo] = b; s[hi] = a; }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

constexpr int kMaxBuckets = 2048;  // bucket-array cap (8 KB of counters: two 8192-key workgroups fit a CU)
constexpr int kBucketFallback = 512;  // a bucket this crowded (identical depths en masse) sends the tile to the bitonic path

// Split frames: a bin's sorted list, still in LDS (low word of s_out[i] = id << 4 | block bits), leaves
// as the four lists of its 16x16 blocks (ms::BlockLists).  Thread t holds entries e * THREADS + t: ballot
// + prefix count per (round, wave, block), one wave-wide scan per block over the E * THREADS / 64 counts,
// then every entry drops into its place.  Order is kept; nothing is tested again.
// s_w: 260 words of scratch.  Called by every thread of the workgroup (two barriers inside).
template <int THREADS, int E>
__device__ __forceinline__ void emit_block_lists(const uint64_t *s_out, int F, int n, int start, int bin, int bin_w,
                                                 const ms::BlockLists &out, uint32_t *s_w) {
    constexpr int NW = THREADS / 64, NI = E * NW;
    static_assert(NI <= 64 && NW >= 4, "one wave scans the counts of one block");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int base = 4 * start;
    uint32_t ranks[E];   // 4 x 8 bits: the entry's rank among its wave's entries of each block
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * THREADS + tid;
        const uint32_t word = i < F ? (uint32_t)s_out[i] : 0u;
        ranks[e] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned long long m = __ballot((word >> b) & 1u);
            ranks[e] |= __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) << (8 * b);
            if (lane == 0) s_w[b * 64 + e * NW + w] = (uint32_t)__popcll(m);
        }
    }
    __syncthreads();
    if (w < 4) {
        const uint32_t v = lane < NI ? s_w[w * 64 + lane] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += o;
        }
        if (lane < NI) s_w[w * 64 + lane] = incl - v;
        if (lane == 63) s_w[256 + w] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * THREADS + tid;
        if (i >= F) break;
        const uint32_t word = (uint32_t)s_out[i];
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if ((word >> b) & 1u)
                out.block_ids[base + b * n + (int)s_w[b * 64 + e * NW + w] + (int)((ranks[e] >> (8 * b)) & 0xffu)] =
                    (int32_t)(word >> 4);
    }
    if (tid < 4) {
        const int by = bin / bin_w, bx = bin - by * bin_w;
        const int x = 2 * bx + (tid & 1), y = 2 * by + (tid >> 1);
        if (x < out.tw16 && y < out.th16)
            reinterpret_cast<int2 *>(out.block_ranges)[y * out.tw16 + x] =
                make_int2(base + tid * n, base + tid * n + (int)s_w[256 + tid]);
    }
    if (tid == 0) out.bin_more[bin] = F < n ? 1 : 0;
}

template <int THREADS, int E>
struct SortCfg {
    static constexpr int CAP = THREADS * E;
    static constexpr int NB = 2 * CAP <= kMaxBuckets ? 2 * CAP : kMaxBuckets;  // ~2 buckets per key while LDS allows
    static constexpr size_t LDS = (size_t)CAP * 8 + (size_t)NB * 4 + 64 * 4;
};

template <int THREADS, int E>
__device__ __forceinline__ void sort_segment_lds(unsigned char *smem, const uint64_t *__restrict__ keys_in,
                                                 int start, int n, int tile,
                                                 int32_t *__restrict__ flatten_ids,
                                                 int64_t *__restrict__ isect_ids,
                                                 uint64_t *__restrict__ keys_out,
                                                 const ms::BlockLists *blocks = nullptr, int bin_w = 0) {
    using Cfg = SortCfg<THREADS, E>;
    constexpr int NW = THREADS / 64;
    uint64_t *s_out = reinterpret_cast<uint64_t *>(smem);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_out + Cfg::CAP);
    uint32_t *s_red = s_cnt + Cfg::NB;  // 64 words of reduction scratch
    // THREADS == 64: ONE WAVE sorts the segment on its own (smem is the wave's private slice; wave-level
    // synchronisation only): several short lists per workgroup, one per wave, no workgroup barrier anywhere
    constexpr bool WAVE = THREADS == 64;
    const int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, lane = tid & 63, w = WAVE ? 0 : tid >> 6;
    auto sync = [&]() __attribute__((always_inline)) {
        if constexpr (WAVE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            __syncthreads();
        }
    };

    // 1. keys -> registers; min / max of the depth bits
    uint64_t k[E];
    uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * THREADS + tid;
        k[e] = i < n ? keys_in[start + i] : ~0ull;
        if (i < n) {
            const uint32_t hi = (uint32_t)(k[e] >> 32);
            kmin = min(kmin, hi);
            kmax = max(kmax, hi);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, d));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d));
    }
    if (lane == 0) { s_red[w] = kmin; s_red[16 + w] = kmax; }
    const int B = min(Cfg::NB, max(64, 2 * next_pow2(n)));
    for (int b = tid; b < B; b += THREADS) s_cnt[b] = 0;
    sync();
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) { kmin = min(kmin, s_red[ww]); kmax = max(kmax, s_red[16 + ww]); }
    const uint32_t span = kmax - kmin;
    const int bits = span ? 32 - __clz(span) : 0;
    const int shift = max(0, bits - (31 - __clz(B)));

    // 2. histogram
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (e * THREADS + tid < n) atomicAdd(&s_cnt[((uint32_t)(k[e] >> 32) - kmin) >> shift], 1u);
    sync();

    // 3. exclusive scan of the B counters (each lane owns `per` consecutive ones)
    const int per = (B + THREADS - 1) / THREADS;   // <= NB / THREADS, a power of two or 1
    uint32_t local[Cfg::NB / THREADS > 0 ? Cfg::NB / THREADS : 1];
    uint32_t sum = 0, cmax = 0;
#pragma unroll
    for (int q = 0; q < (int)(sizeof(local) / 4); ++q) {
        const int b = tid * per + q;
        local[q] = (q < per && b < B) ? s_cnt[b] : 0u;
        sum += local[q];
        cmax = max(cmax, local[q]);
    }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += o;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, d));
    sync();  // s_red reuse
    if (lane == 63) s_red[w] = incl;
    if (lane == 0) s_red[16 + w] = cmax;
    sync();
    uint32_t run = incl - sum;
    cmax = 0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
        if (ww < w) run += s_red[ww];
        cmax = max(cmax, s_red[16 + ww]);
    }
#pragma unroll
    for (int q = 0; q < (int)(sizeof(local) / 4); ++q) {
        const int b = tid * per + q;
        if (q < per && b < B) { s_cnt[b] = run; run += local[q]; }
    }
    sync();

    // 4. scatter into buckets (s_cnt[b] becomes the END of bucket b)
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (e * THREADS + tid < n) {
            const uint32_t pos = atomicAdd(&s_cnt[((uint32_t)(k[e] >> 32) - kmin) >> shift], 1u);
            s_out[pos] = k[e];
        }
    sync();

    // 5. finish.  Every key ranks itself inside its bucket (reads only: k independent LDS loads
    //    for a bucket of k keys, no divergent dependent chains), barrier, then drops into place.
    //    Keys are distinct, so ranks are a permutation.  A crowded bucket (many identical depths)
    //    sends the tile to the oblivious network instead.
    if (WAVE || cmax <= (uint32_t)kBucketFallback) {   // (a wave's list is shorter than any crowded bucket)
        int dest[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            dest[e] = -1;
            if (e * THREADS + tid < n) {
                const uint32_t b = ((uint32_t)(k[e] >> 32) - kmin) >> shift;
                const int beg = b ? (int)s_cnt[b - 1] : 0, end = (int)s_cnt[b];
                int r = 0;
                for (int j = beg; j < end; ++j) r += s_out[j] < k[e] ? 1 : 0;
                dest[e] = beg + r;
            }
        }
        sync();
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (dest[e] >= 0) s_out[dest[e]] = k[e];
    } else if constexpr (!WAVE) {
        const int P = max(2, next_pow2(n));
        for (int i = n + tid; i < P; i += THREADS) s_out[i] = ~0ull;
        sync();
        bitonic_sort_lds<THREADS>(s_out, P);
    }
    sync();
    if (blocks) {   // split frame: block lists instead of the bin's own list
        if constexpr (E * (THREADS / 64) <= 64 && THREADS >= 256)   // (only the small class sorts bins)
            emit_block_lists<THREADS, E>(s_out, n, n, start, tile, bin_w, *blocks, s_cnt);
        return;
    }
    for (int i = tid; i < n; i += THREADS) {
        const uint64_t v = s_out[i];
        if (flatten_ids) flatten_ids[start + i] = (int32_t)(uint32_t)v;
        if (isect_ids) isect_ids[start + i] = ((int64_t)tile << 32) | (int64_t)(v >> 32);
        if (keys_out) keys_out[start + i] = v;
    }
}


// ---- lazily sorted fronts: selection + sort of the nearest keys of a heavy list (k_tile_front) --------
// pass A: min / max of the depth bits (skipped when the caller knows a range);  pass B: LDS histogram over kFrontNB
// order-preserving buckets;  scan;  b* = first bucket whose inclusive prefix reaches front_k;  pass C: keys of buckets
// <= b* go to LDS, rank themselves inside their bucket (as in sort_segment_lds) and drop into place.
// -> F, the number of sorted keys left in s_out[0, F) (<= front_cap; possibly 0: >= front_cap entries at one depth).
// Every thread of the (THREADS-thread) workgroup; s_cnt: kFrontNB words, s_red: 64 words, s_sel: 4 words of LDS
// (s_sel[2] / s_sel[3] = the depth bits at which the selected front ends / the list begins, for the depth cut-off of the
// next frame).
constexpr int kFrontK = 1024;
constexpr int kFrontCap = 4096;   // LDS room for selected keys (32 KB)
constexpr int kFrontNB = 2048;    // buckets
constexpr int kFrontLogNB = 11;
static_assert((1 << kFrontLogNB) == kFrontNB, "bucket count");

template <int THREADS, int kLoads>
__device__ __forceinline__ int front_select_lds(const uint64_t *__restrict__ kin, int n, uint64_t *s_out, uint32_t *s_cnt,
                                                uint32_t *s_red, int *s_sel, uint32_t fixed_min, int fixed_shift, int front_k,
                                                int front_cap) {
    constexpr int NW = THREADS / 64;
    constexpr int kRank = kFrontCap / THREADS;   // keys a thread ranks
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // A. depth-bit range: the camera's [near, far] when the caller knows it (every surviving
    //    depth lies inside, so the pass over the keys is saved), else the tile's own min / max
    uint32_t kmin = fixed_min;
    int shift = fixed_shift;
    for (int b = tid; b < kFrontNB; b += THREADS) s_cnt[b] = 0;
    if (fixed_shift < 0) {
        uint32_t kmax = 0u;
        kmin = 0xffffffffu;
        for (int i = tid; i < n; i += THREADS) {
            const uint32_t d = (uint32_t)(kin[i] >> 32);
            kmin = min(kmin, d);
            kmax = max(kmax, d);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, d));
            kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d));
        }
        if (lane == 0) { s_red[w] = kmin; s_red[16 + w] = kmax; }
        __syncthreads();
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) { kmin = min(kmin, s_red[ww]); kmax = max(kmax, s_red[16 + ww]); }
        const uint32_t span = kmax - kmin;
        const int bits = span ? 32 - __clz(span) : 0;
        shift = max(0, bits - kFrontLogNB);
    } else {
        __syncthreads();
    }
    // B. histogram
    auto bucket_of = [&](uint64_t k) -> int {
        const uint32_t d = (uint32_t)(k >> 32);
        const uint32_t b = (d > kmin ? d - kmin : 0u) >> shift;
        return (int)min(b, (uint32_t)(kFrontNB - 1));
    };
    // (both passes over the keys fetch kLoads keys per thread before touching them: one memory round trip
    // per kLoads * THREADS keys instead of one per THREADS -- the kernel lasts as long as its largest
    // list takes, and that was 2 x n / THREADS dependent round trips)
    for (int i0 = 0; i0 < n; i0 += kLoads * THREADS) {
        uint32_t d[kLoads];
#pragma unroll
        for (int u = 0; u < kLoads; ++u) {
            const int i = i0 + u * THREADS + tid;
            d[u] = (uint32_t)(kin[min(i, n - 1)] >> 32);   // (unconditional: the loads of a batch issue together)
        }
#pragma unroll
        for (int u = 0; u < kLoads; ++u)
            if (i0 + u * THREADS + tid < n) atomicAdd(&s_cnt[bucket_of((uint64_t)d[u] << 32)], 1u);
    }
    __syncthreads();
    // scan: thread t owns kBpt consecutive buckets
    constexpr int kBpt = kFrontNB / THREADS;
    uint32_t c[kBpt], sum = 0;
#pragma unroll
    for (int j = 0; j < kBpt; ++j) { c[j] = s_cnt[kBpt * tid + j]; sum += c[j]; }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads();   // s_red reuse
    if (lane == 63) s_red[w] = incl;
    if (tid == 0) { s_sel[0] = -1; s_sel[1] = 0; s_sel[3] = 0; }
    __syncthreads();
    uint32_t run = incl - sum;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww)
        if (ww < w) run += s_red[ww];
    // b* = the bucket whose inclusive prefix first reaches kFrontK (n > kFrontK, so it exists);
    // if that would overflow the LDS room, stop one bucket earlier (possibly with nothing:
    // >= kFrontCap entries at one depth -- the clean-up kernel takes such a tile)
    // (front_k: kFrontK per 16x16 block of the tile; a tile shorter than that is selected whole)
    const uint32_t want = min((uint32_t)front_k, (uint32_t)n);
#pragma unroll
    for (int j = 0; j < kBpt; ++j) {
        const uint32_t e = run, i = run + c[j];   // exclusive / inclusive prefix of bucket kBpt * tid + j
        if (c[j] && e < want && i >= want) {
            if (i <= (uint32_t)front_cap) { s_sel[0] = kBpt * tid + j; s_sel[1] = (int)i; }
            else { s_sel[0] = kBpt * tid + j - 1; s_sel[1] = (int)e; }
        }
        if (c[j] && e == 0u) s_sel[3] = kBpt * tid + j;   // the first bucket that holds anything (exactly one thread)
        s_cnt[kBpt * tid + j] = e;
        run = i;
    }
    __syncthreads();
    const int bstar = s_sel[0], F = s_sel[1];
    // (for the caller: the depth bits at which bucket b* ends, i.e. where this front ends -- s_sel[2], saturating -- and
    // those at which the list's first occupied bucket begins -- s_sel[3])
    if (tid == 0) {
        const unsigned long long edge = (unsigned long long)kmin + (((unsigned long long)(bstar + 1)) << shift);
        const unsigned long long first = (unsigned long long)kmin + (((unsigned long long)s_sel[3]) << shift);
        s_sel[2] = (int)(uint32_t)(edge > 0xffffffffull ? 0xffffffffull : edge);
        s_sel[3] = (int)(uint32_t)(first > 0xffffffffull ? 0xffffffffull : first);
    }
    // C. select
    for (int i0 = 0; i0 < n; i0 += kLoads * THREADS) {
        uint64_t k[kLoads];
#pragma unroll
        for (int u = 0; u < kLoads; ++u) {
            const int i = i0 + u * THREADS + tid;
            k[u] = kin[min(i, n - 1)];
        }
#pragma unroll
        for (int u = 0; u < kLoads; ++u) {
            const int b = bucket_of(k[u]);
            if (i0 + u * THREADS + tid < n && b <= bstar)
                s_out[atomicAdd(&s_cnt[b], 1u)] = k[u];   // s_cnt[b] becomes the END of bucket b
        }
    }
    __syncthreads();
    // rank inside the bucket; keys are distinct, so ranks are a permutation
    uint64_t kk[kRank];
    int dest[kRank];
#pragma unroll
    for (int e = 0; e < kRank; ++e) {
        const int i = e * THREADS + tid;
        dest[e] = -1;
        if (i < F) {
            kk[e] = s_out[i];
            const int b = bucket_of(kk[e]);
            const int beg = b ? (int)s_cnt[b - 1] : 0, end = (int)s_cnt[b];
            // buckets > b* kept their exclusive prefix in s_cnt, so s_cnt[b-1] is the end of b-1 for every b <= b*
            int r = 0;
            for (int j = beg; j < end; ++j) r += s_out[j] < kk[e] ? 1 : 0;
            dest[e] = beg + r;
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kRank; ++e)
        if (dest[e] >= 0) s_out[dest[e]] = kk[e];
    __syncthreads();
    return F;
}

}  // namespace
