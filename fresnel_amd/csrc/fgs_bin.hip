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
//   k_tile_order   turns the lengths into [start, end) ranges (and launch order, depth segments);
//   k_bin_scatter  one wave per block of depth ranks walks its Gaussians IN ORDER and drops each id at
//                  start[tile] + cnt[image][block][tile] + (ids this block already put into the tile); also writes
//                  dup_off[g] = first duplicate slot of Gaussian g (emission order: image, depth rank, tile row, tile
//                  column -- the gradient-row addressing of the backward) and its share of the unit -> tile table.
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

// NW waves per block of BIN_G depth ranks (NW = 4 when the counters of four waves fit in LDS, i.e. <= 1024 tiles):
// wave w owns ranks [w, w+1) * BIN_G / NW of the block.  The walk over a wave's Gaussians is inherently serial
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
                                                         uint32_t *__restrict__ dup_off,
                                                         const uint32_t *__restrict__ seg_off,
                                                         uint32_t *__restrict__ seg_tile) {
    constexpr uint32_t T_MAX = NW == 1 ? BIN_MAX_TILES : BIN_MAX_TILES / NW;
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
    // (b) unit -> tile table of this block's share of the image's tiles (the backward's work-unit list)
    if (seg_tile) {
        const uint32_t tpb = (tiles + bpi - 1) / bpi;
        for (uint32_t k = threadIdx.x; k < tpb; k += 64 * NW) {
            const uint32_t t = blk * tpb + k;
            if (t < tiles) {
                const uint32_t bt = b * tiles + t, o0 = seg_off[bt], o1 = seg_off[bt + 1];
                for (uint32_t u = o0; u < o1; ++u) seg_tile[u] = bt;
            }
        }
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = NW > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0u;
    // this lane's Gaussian of each 64-rank batch of the wave (WG / 64 batches)
    constexpr int NB = WG / 64;
    static_assert(WG % 64 == 0, "whole 64-rank batches per wave");
    uint32_t gid[NB];
    TileRect q[NB];
    uint32_t inv[NB];  // ceil(2^18 / w): (t * inv) >> 18 is t / w or one more for t < 2^12 (fixed up below)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const uint32_t r = blk * BIN_G + wave * WG + i * 64 + lane;
        gid[i] = 0; q[i] = TileRect{0, 0, 1, 0};
        if (r < N) {
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
#ifndef FGS_EXP_NOSTORE
                if (pos < dcap) dup_ids[pos] = idg;
#else
                if (pos == 0xFFFFFFF0u) dup_ids[pos] = idg;
#endif
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

// Launch order of the composite kernels: tiles with the longest lists first (LPT scheduling:
// the hardware dispatcher hands workgroups to CUs in blockIdx order, so heavy tiles start
// early and light ones fill the tail).  Single-block counting sort into 64 length buckets;
// the order inside a bucket is arbitrary -- it affects scheduling only, never results.
//
// The same block also cuts every list into depth segments of FGS_SEG entries (the backward's work
// units): exclusive scan of ceil(len / FGS_SEG) over the tiles -> seg_off, the unit -> tile map
// seg_tile, and the unit count in seg_off[ntiles] and counters[2].
__global__ __launch_bounds__(1024) void k_tile_order(uint32_t ntiles, uint32_t *__restrict__ ranges,
                                                     const uint32_t *__restrict__ lens,
                                                     uint32_t *__restrict__ tile_order,
                                                     uint32_t *__restrict__ seg_off, uint32_t *__restrict__ seg_tile,
                                                     uint32_t *__restrict__ counters, uint32_t seg_len,
                                                     uint32_t fwd_variant) {
    __shared__ uint32_t hist[64];
    __shared__ uint32_t maxc;
    __shared__ unsigned long long wsum64[16];
    __shared__ unsigned long long carry64;
    // list lengths are read from global memory once and kept in LDS (when they fit) for the four sweeps below
    constexpr uint32_t LCAP = 8192;
    __shared__ uint32_t lcnt[LCAP];
    const bool cached = ntiles <= LCAP;
    // list lengths: from the sorted keys' ranges (radix path) or straight from the direct binning (`lens`, in
    // which case the [start, end) ranges are produced here by the sweep at the end)
    auto len_of = [&](uint32_t t) -> uint32_t {
        return cached ? lcnt[t] : (lens ? lens[t] : ranges[2 * t + 1] - ranges[2 * t]);
    };
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) maxc = 1;
    __syncthreads();
    uint32_t mymax = 0;
    for (uint32_t t = threadIdx.x; t < ntiles; t += 1024) {
        uint32_t len;
        if (lens) len = lens[t];
        else { const uint2 r = reinterpret_cast<const uint2 *>(ranges)[t]; len = r.y - r.x; }
        if (cached) lcnt[t] = len;
        mymax = max(mymax, len);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mymax = max(mymax, (uint32_t)__shfl_xor((int)mymax, o, 64));
    if ((threadIdx.x & 63u) == 0) atomicMax(&maxc, mymax);  // one LDS atomic per wave
    __syncthreads();
    const float scale = 63.0f / (float)maxc;  // bucket = 63 - floor(len * 63 / max), in float: scheduling only
    for (uint32_t t = threadIdx.x; t < ntiles; t += 1024) {
        const uint32_t cnt = len_of(t);
        atomicAdd(&hist[63u - min(63u, (uint32_t)((float)cnt * scale))], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 64; ++i) { const uint32_t h = hist[i]; hist[i] = run; run += h; }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < ntiles; t += 1024) {
        const uint32_t cnt = len_of(t);
        const uint32_t pos = atomicAdd(&hist[63u - min(63u, (uint32_t)((float)cnt * scale))], 1u);
        tile_order[pos] = t;
    }
    if (!seg_off && !lens) return;
    // One sweep over the tiles scans (list length, depth segments) packed in 64 bits: ranges (direct binning) and
    // the depth-segment units.  Every thread owns a run of consecutive tiles, so the block needs ONE scan (a wave
    // scan and sixteen wave sums) however many tiles there are.
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t per = (ntiles + 1023) / 1024, t0 = threadIdx.x * per, t1 = min(ntiles, t0 + per);
    unsigned long long v = 0;
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t len = len_of(t);
        v += ((unsigned long long)len << 32) | ((len + seg_len - 1) / seg_len);
    }
    unsigned long long x = v;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long y = __shfl_up(x, o, 64);
        if (lane >= (uint32_t)o) x += y;
    }
    if (lane == 63) wsum64[wave] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 16; ++i) { const unsigned long long h = wsum64[i]; wsum64[i] = run; run += h; }
        carry64 = run;
    }
    __syncthreads();
    unsigned long long off64 = wsum64[wave] + x - v;
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t len = len_of(t), n = (len + seg_len - 1) / seg_len;
        const uint32_t off = (uint32_t)off64, lstart = (uint32_t)(off64 >> 32);
        if (lens) { ranges[2 * t] = lstart; ranges[2 * t + 1] = lstart + len; }
        if (seg_off) {
            seg_off[t] = off;
            if (seg_tile) for (uint32_t k = 0; k < n; ++k) seg_tile[off + k] = t;  // (direct binning: k_bin_scatter fills it)
        }
        off64 += ((unsigned long long)len << 32) | n;
    }
    if (threadIdx.x == 0 && seg_off) {
        seg_off[ntiles] = (uint32_t)carry64;
        counters[2] = (uint32_t)carry64;
        // the split this forward runs with: the backward kernels read it from here, not from their launch
        counters[4] = seg_len;
        counters[5] = fwd_variant;
    }
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
        hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, st, ntiles_all, ranges, lens, tile_order, seg_off,
                           (uint32_t *)nullptr, counters, (uint32_t)p.L.seg_len, (uint32_t)p.fwd_variant);
        FGS_LAUNCH_CHECK("k_tile_order");
        fgs_stage_end(ST_TILE_RANGES, st);
        fgs_stage_begin(ST_TILE_SORT, st);
        if ((uint32_t)p.tiles <= BIN_MAX_TILES / 4)
            hipLaunchKernelGGL(k_bin_scatter<4>, dim3(B * bpi), dim3(256), 0, st, N, (uint32_t)p.tiles,
                               (uint32_t)p.L.tiles_x, bpi, dcap, order, tile_count, rec, cnt, ranges, dup_ids, bsum,
                               dup_off, seg_off, seg_tile);
        else
            hipLaunchKernelGGL(k_bin_scatter<1>, dim3(B * bpi), dim3(64), 0, st, N, (uint32_t)p.tiles,
                               (uint32_t)p.L.tiles_x, bpi, dcap, order, tile_count, rec, cnt, ranges, dup_ids, bsum,
                               dup_off, seg_off, seg_tile);
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
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, st, ntiles_all, ranges, (const uint32_t *)nullptr,
                       tile_order, seg_off, seg_tile, counters, (uint32_t)p.L.seg_len, (uint32_t)p.fwd_variant);
    FGS_LAUNCH_CHECK("k_tile_order");
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
