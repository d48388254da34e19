// Stage 2: depth order + per-tile duplication + tile sort + tile ranges.
// New design (the reference renderer has no tiles: it walks each depth-sorted Gaussian's
// integer bbox, DR:582-600).  A tile's list = the Gaussians whose reference bbox intersects
// the tile, in the reference's depth order, so compositing a tile list reproduces DR:582-667
// for the pixels of that tile.
//
// All sizes that depend on the data (D = number of duplicates) stay on the device: grids are
// sized from capacities and kernels read D from saved.counters, so the whole forward is
// free of host synchronisation and can be captured in a hipGraph.
#include "fgs_internal.h"

namespace {

// exclusive scan of one value per thread over a 256-thread block; returns block total in *tot
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *tot) {
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(s, o, 64);
        if ((int)lane >= o) s += t;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    uint32_t pre = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t x = wsum[w];
        pre += (w < (int)wave) ? x : 0u;
        all += x;
    }
    *tot = all;
    return pre + s - v;
}

// tile count of the r-th Gaussian (depth order) of image b
__device__ __forceinline__ uint32_t sorted_count(uint32_t i, uint32_t total, uint32_t N,
                                                 const uint32_t *__restrict__ order,
                                                 const uint32_t *__restrict__ tile_count, uint32_t *gid) {
    if (i >= total) { *gid = 0; return 0; }
    const uint32_t b = i / N;
    const uint32_t g = b * N + order[i];
    *gid = g;
    return tile_count[g];
}

__global__ __launch_bounds__(256) void k_dup_blocksum(uint32_t total, uint32_t N,
                                                      const uint32_t *__restrict__ order,
                                                      const uint32_t *__restrict__ tile_count,
                                                      uint32_t *__restrict__ bsum) {
    uint32_t gid, tot;
    const uint32_t c = sorted_count(blockIdx.x * 256 + threadIdx.x, total, N, order, tile_count, &gid);
    block_exclusive_scan_256(c, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// single block: in-place exclusive scan of bsum[0..n), total -> counters[0] (clamped), overflow flag
__global__ __launch_bounds__(256) void k_dup_scan_bsum(uint32_t n, uint32_t *__restrict__ bsum,
                                                       uint32_t *__restrict__ counters, uint32_t dcap) {
    unsigned long long carry = 0;
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? bsum[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan_256(v, &tot);
        if (i < n) bsum[i] = (uint32_t)carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counters[0] = carry > dcap ? dcap : (uint32_t)carry;
        counters[1] = carry > dcap ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void k_dup_emit(uint32_t total, uint32_t N, uint32_t tiles, uint32_t tiles_x,
                                                  uint32_t dcap, const uint32_t *__restrict__ order,
                                                  const uint32_t *__restrict__ tile_count,
                                                  const float *__restrict__ rec,
                                                  const uint32_t *__restrict__ bsum,
                                                  const uint32_t *__restrict__ layer, uint32_t layers,
                                                  uint32_t *__restrict__ dup_off,
                                                  uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t gid, tot;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t c = sorted_count(i, total, N, order, tile_count, &gid);
    uint32_t off = bsum[blockIdx.x] + block_exclusive_scan_256(c, &tot);
    if (i < total) dup_off[gid] = off;
    // Wave-cooperative emission: the wave walks over its Gaussians (uniform loop, parameters broadcast with
    // v_readlane) and the 64 lanes write each Gaussian's duplicates side by side -- coalesced runs instead of
    // 64 scattered 4-byte stores per instruction.
    uint32_t tx0 = 0, ty0 = 0, w = 1, kbase = 0;
    if (c != 0) {
        const uint32_t bbx = __float_as_uint(rec[(size_t)gid * FGS_REC_FLOATS + R_BBX]);
        const uint32_t bby = __float_as_uint(rec[(size_t)gid * FGS_REC_FLOATS + R_BBY]);
        tx0 = (bbx & 0xFFFFu) / FGS_TILE;
        ty0 = (bby & 0xFFFFu) / FGS_TILE;
        w = ((bbx >> 16) - 1) / FGS_TILE - tx0 + 1;
        kbase = ((gid / N) * layers + (layer ? layer[gid] : 0u)) * tiles;
    }
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long m = __ballot(c != 0);
    while (m) {
        const int g = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint32_t cg = __builtin_amdgcn_readlane(c, g), og = __builtin_amdgcn_readlane(off, g);
        const uint32_t wg = __builtin_amdgcn_readlane(w, g), txg = __builtin_amdgcn_readlane(tx0, g);
        const uint32_t tyg = __builtin_amdgcn_readlane(ty0, g), kb = __builtin_amdgcn_readlane(kbase, g);
        const uint32_t idg = __builtin_amdgcn_readlane(gid, g);
        const float rw = 1.0f / (float)wg;
        for (uint32_t t = lane; t < cg; t += 64) {
            uint32_t r = (uint32_t)(((float)t + 0.5f) * rw);  // t / wg (t < 2^20, exact after the fix-up below)
            if (r * wg > t) --r;
            if ((r + 1) * wg <= t) ++r;
            const uint32_t o = og + t;
            if (o < dcap) {
                keys[o] = kb + (tyg + r) * tiles_x + txg + (t - r * wg);
                vals[o] = idg;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Direct binning (single layer, <= BIN_MAX_TILES tiles per image): a counting sort straight from the bboxes, in
// place of "emit (tile key, id) pairs, then two stable radix passes over them".
//   k_bin_count    one block per BIN_G consecutive depth ranks of one image: per-tile counts in LDS
//                  -> cnt[image][block][tile]; also the block's number of duplicates -> bsum;
//   k_bin_scan     one thread per (image, tile): exclusive scan of its column over the blocks (in place), column
//                  total = list length; its last block scans bsum (duplicate offsets, total D -> counters);
//   k_tile_pre/post turn the lengths into [start, end) ranges (and launch order, depth segments);
//   k_bin_scatter  one wave per block of depth ranks walks its Gaussians IN ORDER and drops each id at
//                  start[tile] + cnt[image][block][tile] + (ids this block already put into the tile); also writes
//                  dup_off[g] = first duplicate slot of Gaussian g (emission order: image, depth rank, tile row, tile
//                  column -- the gradient-row addressing of the backward).
// Every list comes out in depth order, exactly as the stable sort produced it, with one scattered 4-byte store
// per duplicate instead of four (two passes x key + payload) and no key traffic at all
// (emit 0.052 + sort 0.148 + ranges 0.012 ms -> offsets/count/scan 0.030 + scatter 0.080 ms at config 3).
constexpr uint32_t BIN_G = FGS_BIN_G;    // depth ranks per binning block
constexpr uint32_t BIN_MAX_TILES = FGS_BIN_MAX_TILES; // LDS counters per block (16 KB)

struct TileRect { uint32_t tx0, ty0, w, cnt; };
__device__ __forceinline__ TileRect tile_rect(const float *__restrict__ rec, uint32_t gid, uint32_t cnt) {
    TileRect r = {0, 0, 1, cnt};
    if (cnt) {
        const uint32_t bbx = __float_as_uint(rec[(size_t)gid * FGS_REC_FLOATS + R_BBX]);
        const uint32_t bby = __float_as_uint(rec[(size_t)gid * FGS_REC_FLOATS + R_BBY]);
        r.tx0 = (bbx & 0xFFFFu) / FGS_TILE;
        r.ty0 = (bby & 0xFFFFu) / FGS_TILE;
        r.w = ((bbx >> 16) - 1) / FGS_TILE - r.tx0 + 1;
    }
    return r;
}

__global__ __launch_bounds__(256) void k_bin_count(uint32_t N, uint32_t tiles, uint32_t tiles_x, uint32_t bpi,
                                                   const uint32_t *__restrict__ order,
                                                   const uint32_t *__restrict__ tile_count,
                                                   const float *__restrict__ rec, uint32_t *__restrict__ cnt,
                                                   uint32_t *__restrict__ bsum) {
    __shared__ uint32_t hist[BIN_MAX_TILES];
    __shared__ uint32_t btot;
    const uint32_t b = blockIdx.x / bpi, blk = blockIdx.x - b * bpi;
    for (uint32_t t = threadIdx.x; t < tiles; t += 256) hist[t] = 0;
    if (threadIdx.x == 0) btot = 0;
    __syncthreads();
    const uint32_t r = blk * BIN_G + threadIdx.x;
    uint32_t mine = 0;
    if (threadIdx.x < BIN_G && r < N) {
        const uint32_t gid = b * N + order[b * N + r];
        const TileRect q = tile_rect(rec, gid, tile_count[gid]);
        mine = q.cnt;
        const uint32_t h = q.cnt / q.w;
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < q.w; ++x) atomicAdd(&hist[(q.ty0 + y) * tiles_x + q.tx0 + x], 1u);
    }
    // duplicates of this block of depth ranks: the block sums of the duplicate-offset scan (k_bin_scan's last block
    // scans them, k_bin_scatter adds the ranks' own prefix)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if ((threadIdx.x & 63u) == 0) atomicAdd(&btot, mine);
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < tiles; t += 256) cnt[((size_t)b * bpi + blk) * tiles + t] = hist[t];
    if (threadIdx.x == 0) bsum[blockIdx.x] = btot;
}

__global__ __launch_bounds__(256) void k_bin_scan(uint32_t B, uint32_t tiles, uint32_t bpi, uint32_t *__restrict__ cnt,
                                                  uint32_t *__restrict__ lens, uint32_t *__restrict__ bsum,
                                                  uint32_t *__restrict__ counters, uint32_t dcap) {
    if (blockIdx.x == gridDim.x - 1) {
        // last block: in-place exclusive scan of the B * bpi block sums of k_bin_count (duplicate offsets in depth
        // order, image-major), total -> counters[0] (clamped), overflow flag -> counters[1]
        const uint32_t n = B * bpi;
        unsigned long long carry = 0;
        for (uint32_t base = 0; base < n; base += 256) {
            const uint32_t i = base + threadIdx.x;
            const uint32_t v = i < n ? bsum[i] : 0u;
            uint32_t tot;
            const uint32_t ex = block_exclusive_scan_256(v, &tot);
            if (i < n) bsum[i] = (uint32_t)carry + ex;
            carry += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            counters[0] = carry > dcap ? dcap : (uint32_t)carry;
            counters[1] = carry > dcap ? 1u : 0u;
        }
        return;
    }
    // thread = one (image, tile) column of cnt[image][block][tile]; walks the blocks serially (every load of a
    // wave is one contiguous run of tiles), exclusive scan in place, column total = list length
    const uint32_t col = blockIdx.x * 256 + threadIdx.x;
    if (col >= B * tiles) return;
    const uint32_t b = col / tiles, t = col - b * tiles;
    uint32_t *p = cnt + (size_t)b * bpi * tiles + t;
    uint32_t run = 0;
    for (uint32_t k0 = 0; k0 < bpi; k0 += 16) {  // 16 independent loads in flight, then the serial adds
        uint32_t v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (k0 + i < bpi) ? p[(size_t)(k0 + i) * tiles] : 0u;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (k0 + i < bpi) p[(size_t)(k0 + i) * tiles] = run;
            run += v[i];
        }
    }
    lens[col] = run;
}

// NW waves per block of BIN_G depth ranks (NW = 8 when the slot tables of eight waves fit in LDS, i.e. <= 1024 tiles;
// else one wave): wave w owns ranks [w, w+1) * BIN_G / NW of the block.  The walk over a wave's Gaussians is inherently serial
// (each one bumps the counters of its tiles), so what matters is how many such walks run side by side: the four
// waves first count their own duplicates per tile, turn the counts into per-wave start slots (in rank order), and
// then walk independently.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_bin_scatter(uint32_t N, uint32_t tiles, uint32_t tiles_x, uint32_t bpi,
                                                         uint32_t dcap, const uint32_t *__restrict__ order,
                                                         const uint32_t *__restrict__ tile_count,
                                                         const float *__restrict__ rec,
                                                         const uint32_t *__restrict__ cnt,
                                                         const uint32_t *__restrict__ ranges,
                                                         uint32_t *__restrict__ dup_ids,
                                                         const uint32_t *__restrict__ bsum,
                                                         uint32_t *__restrict__ dup_off) {
    constexpr uint32_t T_MAX = NW == 1 ? BIN_MAX_TILES : 1024;  // NW > 1: frames of <= 1024 tiles
    constexpr uint32_t WG = BIN_G / NW;  // ranks per wave
    __shared__ uint32_t run[NW][T_MAX];  // next free slot of every tile list, per wave
    // consecutive rank blocks append to neighbouring list slots: keep them on one XCD so that the 4-byte entries
    // merge into full lines in its L2 (the write side was 6x the list bytes without the remap)
    const uint32_t lb = fgs_xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t b = lb / bpi, blk = lb - b * bpi;
    // (a) duplicate offsets of this block's depth ranks: dup_off[g] = first gradient-row / emission slot of
    //     Gaussian g = scanned block sum + prefix of the tile counts inside the block (rank order)
    for (uint32_t r0 = 0; r0 < BIN_G; r0 += 64 * NW) {
        __shared__ uint32_t carry_sh;
        if (threadIdx.x == 0 && r0 == 0) carry_sh = bsum[lb];
        const uint32_t r = blk * BIN_G + r0 + threadIdx.x;
        uint32_t g = 0, c = 0;
        if (r0 + threadIdx.x < BIN_G && r < N) { g = b * N + order[b * N + r]; c = tile_count[g]; }
        // inclusive scan over the block's 64 * NW threads (wave scans + serial sum of the wave totals)
        __shared__ uint32_t wtot[NW];
        uint32_t x = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if ((int)(threadIdx.x & 63u) >= o) x += y;
        }
        __syncthreads();
        if ((threadIdx.x & 63u) == 63u) wtot[threadIdx.x >> 6] = x;
        __syncthreads();
        uint32_t pre = carry_sh, all = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            pre += (w < (int)(threadIdx.x >> 6)) ? wtot[w] : 0u;
            all += wtot[w];
        }
        if (r0 + threadIdx.x < BIN_G && r < N) dup_off[g] = pre + x - c;
        __syncthreads();
        if (threadIdx.x == 0) carry_sh += all;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = NW > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0u;
    // this lane's Gaussian of each 64-rank batch of the wave (WG / 64 batches)
    constexpr int NB = WG >= 64 ? WG / 64 : 1;
    static_assert(WG % 64 == 0 || WG == 32, "whole 64-rank batches per wave, or half a wave of ranks");
    uint32_t gid[NB];
    TileRect q[NB];
    uint32_t inv[NB];  // ceil(2^18 / w): (t * inv) >> 18 is t / w or one more for t < 2^12 (fixed up below)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const uint32_t r = blk * BIN_G + wave * WG + i * 64 + lane;
        gid[i] = 0; q[i] = TileRect{0, 0, 1, 0};
        if (i * 64 + lane < WG && r < N) {
            gid[i] = b * N + order[b * N + r];
            q[i] = tile_rect(rec, gid[i], tile_count[gid[i]]);
        }
        inv[i] = ((1u << 18) + q[i].w - 1u) / q[i].w;
    }
    if (NW > 1) {
        for (uint32_t t = threadIdx.x; t < tiles; t += 64 * NW)
#pragma unroll
            for (int w = 0; w < NW; ++w) run[w][t] = 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NB; ++i) {  // this wave's duplicates per tile
            const uint32_t h = q[i].cnt / q[i].w;
            for (uint32_t y = 0; y < h; ++y)
                for (uint32_t x = 0; x < q[i].w; ++x) atomicAdd(&run[wave][(q[i].ty0 + y) * tiles_x + q[i].tx0 + x], 1u);
        }
        __syncthreads();
    }
    for (uint32_t t = threadIdx.x; t < tiles; t += 64 * NW) {
        uint32_t base = ranges[2 * ((size_t)b * tiles + t)] + cnt[((size_t)b * bpi + blk) * tiles + t];
#pragma unroll
        for (int w = 0; w < NW; ++w) {  // counts -> start slots, waves in rank order
            const uint32_t c = NW > 1 ? run[w][t] : 0u;
            run[w][t] = base;
            base += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        unsigned long long m = __ballot(q[i].cnt != 0);
        while (m) {  // the wave's Gaussians one after the other, in depth order; lanes = tiles of the current one
            const int g = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t cg = __builtin_amdgcn_readlane(q[i].cnt, g), wg = __builtin_amdgcn_readlane(q[i].w, g);
            const uint32_t txg = __builtin_amdgcn_readlane(q[i].tx0, g), tyg = __builtin_amdgcn_readlane(q[i].ty0, g);
            const uint32_t idg = __builtin_amdgcn_readlane(gid[i], g);
            const uint32_t ivg = __builtin_amdgcn_readlane(inv[i], g);
            for (uint32_t t = lane; t < cg; t += 64) {  // one trip for up to 64 tiles
                // t / wg: the reciprocal is rounded up, so the estimate is exact for wg <= 64 tile columns and at
                // most one too high beyond (t < 4096 <= 2^18 / 64); one compare makes it exact for any frame
                uint32_t y = __umul24(t, ivg) >> 18;
                y -= (y * wg > t) ? 1u : 0u;
                const uint32_t tile = (tyg + y) * tiles_x + txg + (t - y * wg);
                const uint32_t pos = run[wave][tile];  // tiles of one Gaussian are distinct: plain read-modify-write
                run[wave][tile] = pos + 1;
                if (pos < dcap) dup_ids[pos] = idg;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__global__ __launch_bounds__(256) void k_tile_ranges(const uint32_t *__restrict__ counters,
                                                     const uint32_t *__restrict__ keys,
                                                     uint32_t *__restrict__ ranges) {
    const uint32_t D = counters[0];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < D; i += gridDim.x * 256) {
        const uint32_t k = keys[i];
        if (i == 0 || keys[i - 1] != k) ranges[2 * k] = i;
        if (i == D - 1 || keys[i + 1] != k) ranges[2 * k + 1] = i + 1;
    }
}

// Per-tile tables from the list lengths, in two multi-block launches (one 1024-thread block used to do all of it:
// 18 us at 8 192 tiles, 470 us at the 131 072 (image, plane, tile) lists of the batched ASM renderer):
//   ranges      [start, end) of every list (direct binning: from `lens`; radix path: already known);
//   seg_off / seg_tile  the backward's work units: a list is cut into depth segments of seg_len entries; exclusive
//               scan of ceil(len / seg_len) over the tiles, the unit -> tile table, the unit count in
//               seg_off[ntiles] and counters[2]; counters[4..5] = the split this forward runs with;
//   tile_order  launch order of the forward kernels, longest lists first (LPT: the dispatcher hands out workgroups
//               in blockIdx order, so heavy tiles start early and light ones fill the tail): a counting sort into
//               64 quarter-octave length buckets; the order inside a bucket is arbitrary -- scheduling only.
// k_tile_pre: per block of TO_TILES tiles, the sums of (length, units) and the bucket histogram.
// k_tile_post: every block adds up the sums / histograms of the blocks before it (a few hundred values), scans its
// own tiles and writes their rows of all tables.
constexpr uint32_t TO_TILES = 1024;  // tiles per block (256 threads x 4 consecutive tiles)

__device__ __forceinline__ uint32_t length_bucket(uint32_t len) {
    // 63 - quarter-octave of the length: longer lists get smaller bucket numbers (4 buckets per power of two)
    if (len == 0) return 63u;
    const uint32_t lg = 31u - (uint32_t)__clz((int)len);
    const uint32_t frac = lg >= 2u ? (len >> (lg - 2u)) & 3u : (len << (2u - lg)) & 3u;
    const uint32_t q = lg * 4u + frac;  // 0 .. 127
    return q >= 62u ? 0u : 62u - q;
}

__device__ __forceinline__ uint32_t tile_len(const uint32_t *__restrict__ lens, const uint32_t *__restrict__ ranges,
                                             uint32_t t) {
    if (lens) return lens[t];
    const uint2 r = reinterpret_cast<const uint2 *>(ranges)[t];
    return r.y - r.x;
}

__global__ __launch_bounds__(256) void k_tile_pre(uint32_t ntiles, const uint32_t *__restrict__ ranges,
                                                  const uint32_t *__restrict__ lens, uint32_t seg_len,
                                                  unsigned long long *__restrict__ pre64,
                                                  uint32_t *__restrict__ bhist) {
    __shared__ uint32_t hist[64];
    __shared__ unsigned long long wsum[4];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long v = 0;
    const uint32_t t0 = blockIdx.x * TO_TILES + threadIdx.x * 4u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t t = t0 + k;
        if (t < ntiles) {
            const uint32_t len = tile_len(lens, ranges, t);
            v += ((unsigned long long)len << 32) | ((len + seg_len - 1) / seg_len);
            atomicAdd(&hist[length_bucket(len)], 1u);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) pre64[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    if (threadIdx.x < 64) bhist[blockIdx.x * 64 + threadIdx.x] = hist[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_tile_post(uint32_t ntiles, uint32_t *__restrict__ ranges,
                                                   const uint32_t *__restrict__ lens,
                                                   uint32_t *__restrict__ tile_order, uint32_t *__restrict__ seg_off,
                                                   uint32_t *__restrict__ seg_tile, uint32_t *__restrict__ counters,
                                                   uint32_t seg_len, uint32_t fwd_variant,
                                                   const unsigned long long *__restrict__ pre64,
                                                   const uint32_t *__restrict__ bhist) {
    __shared__ uint32_t bucket_pos[64];         // next slot of every bucket for this block's tiles
    __shared__ unsigned long long wsum[4], carry_sh, total_sh;
    __shared__ uint32_t col_before[64], col_total[64];
    const uint32_t nblk = gridDim.x, blk = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // (1) sums of the blocks before this one (and of all blocks)
    {
        unsigned long long before = 0, all = 0;
        for (uint32_t k = threadIdx.x; k < nblk; k += 256) {
            const unsigned long long p = pre64[k];
            all += p;
            if (k < blk) before += p;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { before += __shfl_down(before, o, 64); all += __shfl_down(all, o, 64); }
        __shared__ unsigned long long wb[4], wa[4];
        if (lane == 0) { wb[wave] = before; wa[wave] = all; }
        __syncthreads();
        if (threadIdx.x == 0) { carry_sh = (wb[0] + wb[1]) + (wb[2] + wb[3]); total_sh = (wa[0] + wa[1]) + (wa[2] + wa[3]); }
    }
    // (2) bucket columns: thread (bucket b = tid & 63, quarter q = tid >> 6) sums rows q, q + 4, ...
    {
        uint32_t before = 0, all = 0;
        const uint32_t b = threadIdx.x & 63u;
        for (uint32_t k = threadIdx.x >> 6; k < nblk; k += 4) {
            const uint32_t h = bhist[k * 64 + b];
            all += h;
            if (k < blk) before += h;
        }
        __shared__ uint32_t pb[4][64], pa[4][64];
        pb[threadIdx.x >> 6][b] = before; pa[threadIdx.x >> 6][b] = all;
        __syncthreads();
        if (threadIdx.x < 64) {
            col_before[b] = (pb[0][b] + pb[1][b]) + (pb[2][b] + pb[3][b]);
            col_total[b] = (pa[0][b] + pa[1][b]) + (pa[2][b] + pa[3][b]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (int i = 0; i < 64; ++i) { bucket_pos[i] = run + col_before[i]; run += col_total[i]; }
        }
    }
    // (3) this block's tiles: exclusive scan of (length, units) in tile order
    const uint32_t t0 = blk * TO_TILES + threadIdx.x * 4u;
    uint32_t len[4];
    unsigned long long v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        len[k] = (t0 + k < ntiles) ? tile_len(lens, ranges, t0 + k) : 0u;
        v += ((unsigned long long)len[k] << 32) | ((len[k] + seg_len - 1) / seg_len);
    }
    unsigned long long x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long y = __shfl_up(x, o, 64);
        if (lane >= (uint32_t)o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    unsigned long long off64 = carry_sh + x - v;
#pragma unroll
    for (int w = 0; w < 4; ++w) off64 += (w < (int)wave) ? wsum[w] : 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t t = t0 + k;
        if (t >= ntiles) break;
        const uint32_t n = (len[k] + seg_len - 1) / seg_len, off = (uint32_t)off64, lstart = (uint32_t)(off64 >> 32);
        if (lens) { ranges[2 * t] = lstart; ranges[2 * t + 1] = lstart + len[k]; }
        if (seg_off) {
            seg_off[t] = off;
            for (uint32_t u = 0; u < n; ++u) seg_tile[off + u] = t;
        }
        tile_order[atomicAdd(&bucket_pos[length_bucket(len[k])], 1u)] = t;
        off64 += ((unsigned long long)len[k] << 32) | n;
    }
    if (blk == nblk - 1 && threadIdx.x == 0 && seg_off) {
        seg_off[ntiles] = (uint32_t)total_sh;
        counters[2] = (uint32_t)total_sh;
        // the split this forward runs with: the backward kernels read it from here, not from their launch
        counters[4] = seg_len;
        counters[5] = fwd_variant;
    }
}

// scratch of the two kernels above: (sums, histograms) per block of TO_TILES tiles
static int launch_tile_tables(uint32_t ntiles, uint32_t *ranges, const uint32_t *lens, uint32_t *tile_order,
                              uint32_t *seg_off, uint32_t *seg_tile, uint32_t *counters, uint32_t seg_len,
                              uint32_t fwd_variant, uint32_t *scratch_words, hipStream_t st) {
    const uint32_t nblk = (ntiles + TO_TILES - 1) / TO_TILES;
    unsigned long long *pre64 = reinterpret_cast<unsigned long long *>(scratch_words);
    uint32_t *bhist = scratch_words + 2 * (size_t)nblk;
    hipLaunchKernelGGL(k_tile_pre, dim3(nblk), dim3(256), 0, st, ntiles, ranges, lens, seg_len, pre64, bhist);
    FGS_LAUNCH_CHECK("k_tile_pre");
    hipLaunchKernelGGL(k_tile_post, dim3(nblk), dim3(256), 0, st, ntiles, ranges, lens, tile_order, seg_off, seg_tile,
                       counters, seg_len, fwd_variant, pre64, bhist);
    FGS_LAUNCH_CHECK("k_tile_post");
    return FGS_OK;
}

__global__ __launch_bounds__(256) void k_count_pairs(uint32_t total, const float *__restrict__ rec,
                                                     const uint32_t *__restrict__ tile_count,
                                                     unsigned long long *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long p = 0;
    if (i < total && tile_count[i] != 0) {
        const uint32_t bbx = __float_as_uint(rec[(size_t)i * FGS_REC_FLOATS + R_BBX]);
        const uint32_t bby = __float_as_uint(rec[(size_t)i * FGS_REC_FLOATS + R_BBY]);
        p = (unsigned long long)((bbx >> 16) - (bbx & 0xFFFFu)) * ((bby >> 16) - (bby & 0xFFFFu));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) p += __shfl_down(p, o, 64);
    if ((threadIdx.x & 63u) == 0 && p) atomicAdd(out, p);
}

}  // namespace

int fgs_launch_binning(const FgsPlan &p, char *saved, char *scratch, hipStream_t st) {
    const uint32_t B = p.d.batch, N = p.d.num_gaussians, total = B * N;
    const uint32_t nblk = (total + 255) / 256;
    uint32_t *keys0 = reinterpret_cast<uint32_t *>(scratch + p.s_keys0);
    uint32_t *keys1 = reinterpret_cast<uint32_t *>(scratch + p.s_keys1);
    uint32_t *vals0 = reinterpret_cast<uint32_t *>(scratch + p.s_vals0);
    uint32_t *vals1 = reinterpret_cast<uint32_t *>(scratch + p.s_vals1);
    uint32_t *hist = reinterpret_cast<uint32_t *>(scratch + p.s_hist);
    uint32_t *bsum = reinterpret_cast<uint32_t *>(scratch + p.s_bsum);
    uint32_t *depth_key = reinterpret_cast<uint32_t *>(saved + p.L.depth_key);
    uint32_t *tile_count = reinterpret_cast<uint32_t *>(saved + p.L.tile_count);
    uint32_t *order = reinterpret_cast<uint32_t *>(saved + p.L.order);
    uint32_t *counters = reinterpret_cast<uint32_t *>(saved + p.L.counters);
    uint32_t *ranges = reinterpret_cast<uint32_t *>(saved + p.L.ranges);
    uint32_t *dup_ids = reinterpret_cast<uint32_t *>(saved + p.L.dup_ids);
    const float *rec = reinterpret_cast<const float *>(saved + p.L.rec);
    const uint32_t dcap = (uint32_t)p.L.dup_capacity;

    // (1) canonical depth order per image
    fgs_stage_begin(ST_DEPTH_SORT, st);
    // keys straight from the projection's depth_key (read-only), payload = index inside the image, generated by the
    // first pass; per-pass prefix formed inside the downsweep: 8 launches (was 13)
    uint32_t *ks, *vs;
    int rc = fgs_launch_radix_sort(keys0, vals0, keys1, vals1, order, &ks, &vs, N, nullptr, N, N, B, 32, hist, st,
                                   depth_key, N);
    if (rc) return rc;
    fgs_stage_end(ST_DEPTH_SORT, st);
    const uint32_t ntiles_all = B * (uint32_t)p.layers * (uint32_t)p.tiles;
    uint32_t *tile_order = reinterpret_cast<uint32_t *>(saved + p.L.tile_order);
    uint32_t *seg_off = p.L.seg_capacity ? reinterpret_cast<uint32_t *>(saved + p.L.seg_off) : nullptr;
    uint32_t *seg_tile = p.L.seg_capacity ? reinterpret_cast<uint32_t *>(saved + p.L.seg_tile) : nullptr;
    if (p.direct_binning) {
        // direct binning: counting sort straight from the bboxes (see k_bin_count): 4 launches
        const uint32_t bpi = (N + BIN_G - 1) / BIN_G;
        uint32_t *cnt = keys0;               // [B][tiles][bpi], fits: keys0 holds >= Dcap words
        uint32_t *lens = keys1;              // [B * tiles]
        uint32_t *dup_off = reinterpret_cast<uint32_t *>(saved + p.L.dup_off);
        fgs_stage_begin(ST_DUP_EMIT, st);
        hipLaunchKernelGGL(k_bin_count, dim3(B * bpi), dim3(256), 0, st, N, (uint32_t)p.tiles, (uint32_t)p.L.tiles_x,
                           bpi, order, tile_count, rec, cnt, bsum);
        FGS_LAUNCH_CHECK("k_bin_count");
        hipLaunchKernelGGL(k_bin_scan, dim3((ntiles_all + 255) / 256 + 1), dim3(256), 0, st, B, (uint32_t)p.tiles, bpi,
                           cnt, lens, bsum, counters, dcap);
        FGS_LAUNCH_CHECK("k_bin_scan");
        fgs_stage_end(ST_DUP_EMIT, st);
        fgs_stage_begin(ST_TILE_RANGES, st);
        if ((rc = launch_tile_tables(ntiles_all, ranges, lens, tile_order, seg_off, seg_tile, counters,
                                     (uint32_t)p.L.seg_len, (uint32_t)p.fwd_variant, vals1, st)))
            return rc;
        fgs_stage_end(ST_TILE_RANGES, st);
        fgs_stage_begin(ST_TILE_SORT, st);
        // eight waves x 32 depth ranks per block when the per-wave slot tables fit (<= 1024 tiles): the walk over a
        // wave's Gaussians is serial, so shorter walks, more of them (four waves x 64: 0.061 -> 0.054 ms at config 3)
        if ((uint32_t)p.tiles <= 1024)
            hipLaunchKernelGGL(k_bin_scatter<8>, dim3(B * bpi), dim3(512), 0, st, N, (uint32_t)p.tiles,
                               (uint32_t)p.L.tiles_x, bpi, dcap, order, tile_count, rec, cnt, ranges, dup_ids, bsum,
                               dup_off);
        else
            hipLaunchKernelGGL(k_bin_scatter<1>, dim3(B * bpi), dim3(64), 0, st, N, (uint32_t)p.tiles,
                               (uint32_t)p.L.tiles_x, bpi, dcap, order, tile_count, rec, cnt, ranges, dup_ids, bsum,
                               dup_off);
        FGS_LAUNCH_CHECK("k_bin_scatter");
        fgs_stage_end(ST_TILE_SORT, st);
        return FGS_OK;
    }
    fgs_stage_begin(ST_DUP_EMIT, st);
    // (2) duplicate offsets (exclusive scan over Gaussians in depth order, image-major)
    hipLaunchKernelGGL(k_dup_blocksum, dim3(nblk), dim3(256), 0, st, total, N, order, tile_count, bsum);
    FGS_LAUNCH_CHECK("k_dup_blocksum");
    hipLaunchKernelGGL(k_dup_scan_bsum, dim3(1), dim3(256), 0, st, nblk, bsum, counters, dcap);
    FGS_LAUNCH_CHECK("k_dup_scan_bsum");
    // (3) emit (tile key, gaussian id) in depth order
    hipLaunchKernelGGL(k_dup_emit, dim3(nblk), dim3(256), 0, st, total, N, (uint32_t)p.tiles,
                       (uint32_t)p.L.tiles_x, dcap, order, tile_count, rec, bsum,
                       p.layers > 1 ? reinterpret_cast<const uint32_t *>(saved + p.s_layer) : nullptr,
                       (uint32_t)p.layers, reinterpret_cast<uint32_t *>(saved + p.L.dup_off), keys0, vals0);
    FGS_LAUNCH_CHECK("k_dup_emit");
    fgs_stage_end(ST_DUP_EMIT, st);
    fgs_stage_begin(ST_TILE_SORT, st);
    // (4) stable sort by tile key only
    rc = fgs_launch_radix_sort(keys0, vals0, keys1, vals1, dup_ids, &ks, &vs, 0, counters, dcap, 0, 1,
                               p.tile_key_bits, hist, st);
    if (rc) return rc;
    fgs_stage_end(ST_TILE_SORT, st);
    fgs_stage_begin(ST_TILE_RANGES, st);
    // (5) per-tile [start,end)
    hipError_t e = hipMemsetAsync(ranges, 0, (size_t)ntiles_all * 2 * sizeof(uint32_t), st);
    if (e != hipSuccess) { fgs_set_error("memset ranges: %s", hipGetErrorString(e)); return FGS_ELAUNCH; }
    uint32_t rgrid = (dcap + 255) / 256;
    if (rgrid > 2048) rgrid = 2048;
    hipLaunchKernelGGL(k_tile_ranges, dim3(rgrid), dim3(256), 0, st, counters, ks, ranges);
    FGS_LAUNCH_CHECK("k_tile_ranges");
    // (vals0 is free again: the sort's final payload went to dup_ids, its last keys are in `ks`)
    if ((rc = launch_tile_tables(ntiles_all, ranges, nullptr, tile_order, seg_off, seg_tile, counters,
                                 (uint32_t)p.L.seg_len, (uint32_t)p.fwd_variant, ks == keys0 ? keys1 : keys0, st)))
        return rc;
    fgs_stage_end(ST_TILE_RANGES, st);
    return FGS_OK;
}

int fgs_launch_count_pairs(const FgsPlan &p, const char *saved, uint64_t *out, hipStream_t st) {
    const uint32_t total = p.d.batch * p.d.num_gaussians;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(uint64_t), st);
    if (e != hipSuccess) { fgs_set_error("memset pairs: %s", hipGetErrorString(e)); return FGS_ELAUNCH; }
    hipLaunchKernelGGL(k_count_pairs, dim3((total + 255) / 256), dim3(256), 0, st, total,
                       reinterpret_cast<const float *>(saved + p.L.rec),
                       reinterpret_cast<const uint32_t *>(saved + p.L.tile_count),
                       reinterpret_cast<unsigned long long *>(out));
    FGS_LAUNCH_CHECK("k_count_pairs");
    return FGS_OK;
}
