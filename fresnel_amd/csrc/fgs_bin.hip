// Stage 2: depth order + per-tile lists (mask binning; emit + tile sort as the fallback) + tile tables.
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
                                                  uint32_t tile_w, uint32_t *__restrict__ dup_off,
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
        tx0 = (bbx & 0xFFFFu) / tile_w;
        ty0 = (bby & 0xFFFFu) / FGS_TILE;
        w = ((bbx >> 16) - 1) / tile_w - tx0 + 1;
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

struct TileRect { uint32_t tx0, ty0, w, cnt; };
__device__ __forceinline__ TileRect tile_rect(const float *__restrict__ rec, uint32_t gid, uint32_t cnt, uint32_t tile_w) {
    TileRect r = {0, 0, 1, cnt};
    if (cnt) {
        const uint32_t bbx = __float_as_uint(rec[(size_t)gid * FGS_REC_FLOATS + R_BBX]);
        const uint32_t bby = __float_as_uint(rec[(size_t)gid * FGS_REC_FLOATS + R_BBY]);
        r.tx0 = (bbx & 0xFFFFu) / tile_w;
        r.ty0 = (bby & 0xFFFFu) / FGS_TILE;
        r.w = ((bbx >> 16) - 1) / tile_w - r.tx0 + 1;
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------------
// Direct binning = MASK BINNING (single layer, <= FGS_BIN_MAX_TILES tiles, <= FGS_MASK_MAX_LINES tile columns + rows per
// image), in place of "emit (tile key, id) pairs, then stable radix passes over them".  (Round 1's counting sort --
// per-block tile counters in LDS, a column scan over the blocks, and a scatter whose waves walked their Gaussians one
// after the other -- took 97 us at config 3, 54 of them in the serial walk; this one 81.)
// A tile's list is the set of depth ranks whose tile rectangle [tx0, tx1] x [ty0, ty1] contains the tile:
//     list(x, y) = { r : x in [tx0_r, tx1_r] }  AND  { r : y in [ty0_r, ty1_r] }
// i.e. the bit-and of a per-tile-COLUMN and a per-tile-ROW bit mask over the depth ranks.  So:
//   k_mask_build  one block per 256 consecutive depth ranks of one image: 64-lane ballots give, for every tile
//                 column and row, the 64-bit word of each of its four waves -> masks[image][line][rank word]
//                 (line = column x, or tiles_x + row y); also the block's duplicate count -> bsum;
//   k_mask_count  one wave per (image, tile): popcount of (column mask & row mask) = list length; its last block
//                 scans bsum (duplicate offsets, total D -> counters);
//   k_tile_pre/post ranges, depth segments, launch order (unchanged);
//   k_mask_emit   one wave per (image, tile): lanes take consecutive rank words, a wave scan of their popcounts
//                 gives every lane its run of list slots; the set bits (depth ranks, ascending) are parked in LDS
//                 and the wave then turns them into ids (order[rank]) and stores the list with the lanes side by side
//                 -- coalesced stores, independent gathers; extra blocks write dup_off.
// No walk over Gaussians is serial any more (the scatter's was: 54 of the 97 us list building at config 3), nothing
// depends on the tile count fitting in LDS, and lists come out in depth order by construction.
constexpr uint32_t MB_RANKS = FGS_BIN_G; // depth ranks per k_mask_build block (four rank words)
static_assert(MB_RANKS == 256, "k_mask_build: one 256-thread block = four 64-bit rank words per line");
constexpr uint32_t TO_TILES = FGS_TILE_TABLE_TILES;     // tiles per block of the tile-table kernels (k_tile_pre / k_tile_post below)
__device__ __forceinline__ uint32_t length_bucket(uint32_t len);
constexpr uint32_t MB_MAX_LINES = FGS_MASK_MAX_LINES;  // tile columns + rows the build kernel keeps in LDS

// order[b][i] = i: the "depth order" of a consumer that does not composite in depth order (FgsPlan.depth_ordered == false)
__global__ __launch_bounds__(256) void k_index_order(uint32_t total, uint32_t N, uint32_t *__restrict__ order) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < total) order[i] = i % N;
}

// plane id of every depth rank (layered lists): keys of the stable pass that groups `order` by plane
__global__ __launch_bounds__(256) void k_plane_keys(uint32_t total, uint32_t N, const uint32_t *__restrict__ sorted_idx,
                                                    const uint32_t *__restrict__ layer, uint32_t *__restrict__ keys) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < total) keys[i] = layer[(i / N) * N + sorted_idx[i]];
}

__global__ __launch_bounds__(256) void k_mask_build(uint32_t N, uint32_t tiles_x, uint32_t tiles_y, uint32_t bpi,
                                                    uint32_t w64p, const uint32_t *__restrict__ order,
                                                    const uint32_t *__restrict__ tile_count,
                                                    const float *__restrict__ rec,
                                                    unsigned long long *__restrict__ masks,
                                                    uint32_t *__restrict__ bsum, uint32_t nrb, uint32_t layers,
                                                    const uint32_t *__restrict__ plane_keys,
                                                    uint32_t *__restrict__ plane_start, uint32_t tile_w) {
    __shared__ unsigned long long sm[MB_MAX_LINES][4];
    __shared__ uint32_t wtot[4];
    if (blockIdx.x >= nrb) {
        // layered lists (ASM depth planes): `order` is grouped by plane (stable pass over the plane ids after the depth
        // sort); plane_start[b][p] = first rank of plane p = lower bound of p in the image's sorted plane keys
        const uint32_t i = (blockIdx.x - nrb) * 256 + threadIdx.x, per = layers + 1;
        const uint32_t B = nrb / bpi;
        if (i < B * per) {
            const uint32_t b = i / per, pl = i - b * per;
            const uint32_t *k = plane_keys + (size_t)b * N;
            uint32_t lo = 0, hi = N;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (k[mid] < pl) lo = mid + 1; else hi = mid;
            }
            plane_start[i] = lo;
        }
        return;
    }
    const uint32_t b = blockIdx.x / bpi, blk = blockIdx.x - b * bpi;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t r = blk * MB_RANKS + threadIdx.x;
    uint32_t cnt = 0, tx0 = 0xFFFFu, tx1 = 0, ty0 = 0xFFFFu, ty1 = 0;
    if (r < N) {
        const uint32_t gid = b * N + order[b * N + r];
        const TileRect q = tile_rect(rec, gid, tile_count[gid], tile_w);
        cnt = q.cnt;
        if (cnt) { tx0 = q.tx0; tx1 = q.tx0 + q.w - 1u; ty0 = q.ty0; ty1 = q.ty0 + cnt / q.w - 1u; }
    }
    for (uint32_t x = 0; x < tiles_x; ++x) {
        const unsigned long long m = __ballot(x >= tx0 && x <= tx1);
        if (lane == 0) sm[x][wave] = m;
    }
    for (uint32_t y = 0; y < tiles_y; ++y) {
        const unsigned long long m = __ballot(y >= ty0 && y <= ty1);
        if (lane == 0) sm[tiles_x + y][wave] = m;
    }
    uint32_t mine = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if (lane == 0) wtot[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = (wtot[0] + wtot[1]) + (wtot[2] + wtot[3]);
    const uint32_t lines = tiles_x + tiles_y;
    for (uint32_t l = threadIdx.x; l < lines; l += 256) {
        ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(masks + ((size_t)b * lines + l) * w64p + blk * 4u);
        dst[0] = make_ulonglong2(sm[l][0], sm[l][1]);
        dst[1] = make_ulonglong2(sm[l][2], sm[l][3]);
        if (blk == bpi - 1)  // padding words of the line (w64p is a multiple of 8 words)
            for (uint32_t w = bpi * 4u; w < w64p; ++w) masks[((size_t)b * lines + l) * w64p + w] = 0ull;
    }
}

// WPL consecutive 64-bit words of (column mask & row mask) for this lane
template <int WPL>
__device__ __forceinline__ void mask_words(const unsigned long long *__restrict__ col,
                                           const unsigned long long *__restrict__ row, uint32_t w0,
                                           unsigned long long (&m)[WPL]) {
    if constexpr (WPL == 1) {
        m[0] = col[w0] & row[w0];
    } else {
        const ulonglong2 *c2 = reinterpret_cast<const ulonglong2 *>(col + w0);
        const ulonglong2 *r2 = reinterpret_cast<const ulonglong2 *>(row + w0);
        ulonglong2 cv[WPL / 2], rv[WPL / 2];
#pragma unroll
        for (int k = 0; k < WPL / 2; ++k) { cv[k] = c2[k]; rv[k] = r2[k]; }
#pragma unroll
        for (int k = 0; k < WPL / 2; ++k) { m[2 * k] = cv[k].x & rv[k].x; m[2 * k + 1] = cv[k].y & rv[k].y; }
    }
}

// a list's depth-rank range [r_lo, r_hi): the whole image, or the ranks of the list's layer (depth plane)
struct RankRange { uint32_t b, t, r_lo, r_hi; };
__device__ __forceinline__ RankRange rank_range(uint32_t list, uint32_t N, uint32_t tiles, uint32_t layers,
                                                const uint32_t *__restrict__ plane_start) {
    RankRange r;
    const uint32_t bl = list / tiles;  // b * layers + layer
    r.t = list - bl * tiles;
    r.b = bl / layers;
    r.r_lo = 0; r.r_hi = N;
    if (layers > 1) {
        const uint32_t pl = bl - r.b * layers;
        r.r_lo = plane_start[r.b * (layers + 1) + pl];
        r.r_hi = plane_start[r.b * (layers + 1) + pl + 1];
    }
    return r;
}
// the word w of a (column & row) mask restricted to the ranks [r_lo, r_hi)
__device__ __forceinline__ unsigned long long clip_word(unsigned long long m, uint32_t w, uint32_t r_lo, uint32_t r_hi) {
    const uint32_t lo = w * 64u, hi = lo + 64u;
    if (hi <= r_lo || lo >= r_hi) return 0ull;
    if (r_lo > lo) m &= ~0ull << (r_lo - lo);
    if (r_hi < hi) m &= ~0ull >> (hi - r_hi);
    return m;
}

template <int WPL>
__global__ __launch_bounds__(256) void k_mask_count(uint32_t B, uint32_t N, uint32_t layers,
                                                    const uint32_t *__restrict__ plane_start, uint32_t tiles,
                                                    uint32_t tiles_x, uint32_t lines,
                                                    uint32_t w64p, uint32_t nrb,
                                                    const unsigned long long *__restrict__ masks,
                                                    uint32_t *__restrict__ lens, uint32_t *__restrict__ bsum,
                                                    uint32_t *__restrict__ counters, uint32_t dcap) {
    if (blockIdx.x == gridDim.x - 1) {
        // last block: in-place exclusive scan of the block sums of k_mask_build (duplicate offsets in depth order,
        // image-major), total -> counters[0] (clamped), overflow flag -> counters[1]
        unsigned long long carry = 0;
        for (uint32_t base = 0; base < nrb; base += 256) {
            const uint32_t i = base + threadIdx.x;
            const uint32_t v = i < nrb ? bsum[i] : 0u;
            uint32_t tot;
            const uint32_t ex = block_exclusive_scan_256(v, &tot);
            if (i < nrb) bsum[i] = (uint32_t)carry + ex;
            carry += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            counters[0] = carry > dcap ? dcap : (uint32_t)carry;
            counters[1] = carry > dcap ? 1u : 0u;
        }
        return;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile >= B * layers * tiles) return;
    const RankRange rr = rank_range(tile, N, tiles, layers, plane_start);
    const uint32_t b = rr.b, y = rr.t / tiles_x, x = rr.t - y * tiles_x;
    const unsigned long long *col = masks + ((size_t)b * lines + x) * w64p;
    const unsigned long long *row = masks + ((size_t)b * lines + tiles_x + y) * w64p;
    uint32_t c = 0;
    const uint32_t w_end = min(w64p, (rr.r_hi + 63u) / 64u);
    for (uint32_t w0 = (rr.r_lo / 64u) / (64u * WPL) * (64u * WPL) + lane * WPL; w0 < w_end; w0 += 64u * WPL) {
        unsigned long long m[WPL];
        mask_words<WPL>(col, row, w0, m);
#pragma unroll
        for (int k = 0; k < WPL; ++k) c += (uint32_t)__popcll(clip_word(m[k], w0 + k, rr.r_lo, rr.r_hi));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if (lane == 0) lens[tile] = c;
    // (Folding k_tile_pre in here -- two global atomics per tile into the per-1024-tile sums and bucket histograms --
    // made this 8 us kernel take 102 us at config 3: same-address global atomics serialise at ~50 ns each.)
}

constexpr uint32_t ME_CAP = 2048;       // list entries a wave parks in LDS per flush (k_mask_emit)
#ifndef FGS_GROUP_MIN_LISTS
#define FGS_GROUP_MIN_LISTS 4096u  // from this many lists of <= 16 rank words on: sixteen lanes per list (k_mask_*_group)
#endif
#ifndef FGS_EMIT_BLOCK_MAX_LISTS
#define FGS_EMIT_BLOCK_MAX_LISTS 8192u  // up to this many lists per launch: one block per list (k_mask_emit_block)
#endif

// (Staging `order` in LDS per block of eight tiles, to turn the sparse ids gather into coalesced loads, was tried: no
// faster at config 3 -- 43.7 us -- and slower on long lists, 150 vs 110 us on the decoder-like scene: the per-lane
// 4-byte stores it needs cost more than the gather it removes.)
template <int WPL>
__global__ __launch_bounds__(256) void k_mask_emit(uint32_t B, uint32_t N, uint32_t layers,
                                                   const uint32_t *__restrict__ plane_start, uint32_t tiles,
                                                   uint32_t tiles_x,
                                                   uint32_t lines, uint32_t w64p, uint32_t nrb, uint32_t bpi,
                                                   uint32_t dcap, const unsigned long long *__restrict__ masks,
                                                   const uint32_t *__restrict__ order,
                                                   const uint32_t *__restrict__ tile_count,
                                                   const uint32_t *__restrict__ ranges,
                                                   const uint32_t *__restrict__ bsum,
                                                   uint32_t *__restrict__ dup_ids, uint32_t *__restrict__ dup_off) {
    __shared__ uint32_t park[4][ME_CAP];
    if (blockIdx.x < nrb) {
        // duplicate offsets of one block of depth ranks: dup_off[g] = first gradient-row / emission slot of Gaussian
        // g = scanned block sum + prefix of the tile counts inside the block (rank order)
        const uint32_t b = blockIdx.x / bpi, blk = blockIdx.x - b * bpi;
        const uint32_t r = blk * MB_RANKS + threadIdx.x;
        uint32_t g = 0, c = 0, tot;
        if (r < N) { g = b * N + order[b * N + r]; c = tile_count[g]; }
        const uint32_t ex = block_exclusive_scan_256(c, &tot);
        if (r < N) dup_off[g] = bsum[blockIdx.x] + ex;
        return;
    }
    // consecutive tiles' lists are adjacent in dup_ids: an XCD gets a contiguous run of tiles (whole images at B = 8)
    const uint32_t ntb = gridDim.x - nrb;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tile = fgs_xcd_remap(blockIdx.x - nrb, ntb) * 4u + wave;
    if (tile >= B * layers * tiles) return;
    const RankRange rr = rank_range(tile, N, tiles, layers, plane_start);
    const uint32_t b = rr.b, y = rr.t / tiles_x, x = rr.t - y * tiles_x;
    const unsigned long long *col = masks + ((size_t)b * lines + x) * w64p;
    const unsigned long long *row = masks + ((size_t)b * lines + tiles_x + y) * w64p;
    const uint32_t *ord = order + (size_t)b * N;
    uint32_t *pk = park[wave];
    uint32_t base = ranges[2 * tile];  // next list slot
    const uint32_t w_end = min(w64p, (rr.r_hi + 63u) / 64u);
    for (uint32_t c0 = (rr.r_lo / 64u) / (64u * WPL) * (64u * WPL); c0 < w_end; c0 += 64u * WPL) {
        const uint32_t w0 = c0 + lane * WPL;
        unsigned long long m[WPL];
#pragma unroll
        for (int k = 0; k < WPL; ++k) m[k] = 0ull;
        if (w0 < w_end) {
            mask_words<WPL>(col, row, w0, m);
#pragma unroll
            for (int k = 0; k < WPL; ++k) m[k] = clip_word(m[k], w0 + k, rr.r_lo, rr.r_hi);
        }
        uint32_t cl = 0;
#pragma unroll
        for (int k = 0; k < WPL; ++k) cl += (uint32_t)__popcll(m[k]);
        uint32_t inc = cl;  // inclusive wave scan of the lanes' entry counts
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += v;
        }
        const uint32_t total = __builtin_amdgcn_readlane(inc, 63);
        const uint32_t ex = inc - cl;
        // flush windows of ME_CAP entries (one window unless a tile collects more than ME_CAP of these ranks)
        for (uint32_t win = 0; win < total; win += ME_CAP) {
            // (1) park the ranks of this window's entries in list order
            uint32_t e = ex;
#pragma unroll
            for (int k = 0; k < WPL; ++k) {
                unsigned long long mm = m[k];
                const uint32_t rank0 = (w0 + k) * 64u;
                while (mm) {
                    const uint32_t bit = (uint32_t)__ffsll((long long)mm) - 1u;
                    mm &= mm - 1ull;
                    if (e - win < ME_CAP) pk[e - win] = rank0 + bit;  // unsigned: e < win wraps past ME_CAP
                    ++e;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // (2) ranks -> ids, lanes side by side
            const uint32_t n = min(ME_CAP, total - win);
            for (uint32_t i = lane; i < n; i += 256) {
                uint32_t rk[4], id[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) rk[u] = (i + 64u * u < n) ? pk[i + 64u * u] : 0u;
#pragma unroll
                for (int u = 0; u < 4; ++u) id[u] = ord[rk[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t pos = base + win + i + 64u * u;
                    if (i + 64u * u < n && pos < dcap) dup_ids[pos] = b * N + id[u];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        base += total;
    }
}

// SHORT lists over few rank words (the ASM renderer's (image, plane, tile) lists: a plane's share of the depth ranks is
// ~10 words, a list ~13 entries; 131 072 lists at 8 images): SIXTEEN LANES per list, four lists per wave -- a wave per
// list left 50+ lanes idle in every instruction.  Same algorithm per group: the 16 lanes take consecutive rank words,
// a 16-lane scan gives each lane its run of list slots, bits parked in the group's LDS region, ids gathered side by side.
constexpr uint32_t MG_CAP = 512;  // list entries a 16-lane group parks per flush

__global__ __launch_bounds__(256) void k_mask_count_group(uint32_t B, uint32_t N, uint32_t layers,
                                                          const uint32_t *__restrict__ plane_start, uint32_t tiles,
                                                          uint32_t tiles_x, uint32_t lines, uint32_t w64p, uint32_t nrb,
                                                          const unsigned long long *__restrict__ masks,
                                                          uint32_t *__restrict__ lens, uint32_t *__restrict__ bsum,
                                                          uint32_t *__restrict__ counters, uint32_t dcap) {
    if (blockIdx.x == gridDim.x - 1) {  // scan of k_mask_build's block sums (as in k_mask_count)
        unsigned long long carry = 0;
        for (uint32_t base = 0; base < nrb; base += 256) {
            const uint32_t i = base + threadIdx.x;
            const uint32_t v = i < nrb ? bsum[i] : 0u;
            uint32_t tot;
            const uint32_t ex = block_exclusive_scan_256(v, &tot);
            if (i < nrb) bsum[i] = (uint32_t)carry + ex;
            carry += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            counters[0] = carry > dcap ? dcap : (uint32_t)carry;
            counters[1] = carry > dcap ? 1u : 0u;
        }
        return;
    }
    const uint32_t glane = threadIdx.x & 15u;
    const uint32_t list = blockIdx.x * 16u + (threadIdx.x >> 4);
    const bool valid = list < B * layers * tiles;
    uint32_t c = 0;
    if (valid) {
        const RankRange rr = rank_range(list, N, tiles, layers, plane_start);
        const uint32_t y = rr.t / tiles_x, x = rr.t - y * tiles_x;
        const unsigned long long *col = masks + ((size_t)rr.b * lines + x) * w64p;
        const unsigned long long *row = masks + ((size_t)rr.b * lines + tiles_x + y) * w64p;
        const uint32_t w_end = min(w64p, (rr.r_hi + 63u) / 64u);
        for (uint32_t w = rr.r_lo / 64u + glane; w < w_end; w += 16u)
            c += (uint32_t)__popcll(clip_word(col[w] & row[w], w, rr.r_lo, rr.r_hi));
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (valid && glane == 0) lens[list] = c;
}

__global__ __launch_bounds__(256) void k_mask_emit_group(uint32_t B, uint32_t N, uint32_t layers,
                                                         const uint32_t *__restrict__ plane_start, uint32_t tiles,
                                                         uint32_t tiles_x, uint32_t lines, uint32_t w64p, uint32_t nrb,
                                                         uint32_t bpi, uint32_t dcap,
                                                         const unsigned long long *__restrict__ masks,
                                                         const uint32_t *__restrict__ order,
                                                         const uint32_t *__restrict__ tile_count,
                                                         const uint32_t *__restrict__ ranges,
                                                         const uint32_t *__restrict__ bsum,
                                                         uint32_t *__restrict__ dup_ids, uint32_t *__restrict__ dup_off) {
    __shared__ uint32_t park[16][MG_CAP];
    if (blockIdx.x < nrb) {  // duplicate offsets of one block of depth ranks (as in k_mask_emit)
        const uint32_t b = blockIdx.x / bpi, blk = blockIdx.x - b * bpi;
        const uint32_t r = blk * MB_RANKS + threadIdx.x;
        uint32_t g = 0, c = 0, tot;
        if (r < N) { g = b * N + order[b * N + r]; c = tile_count[g]; }
        const uint32_t ex = block_exclusive_scan_256(c, &tot);
        if (r < N) dup_off[g] = bsum[blockIdx.x] + ex;
        return;
    }
    const uint32_t ntb = gridDim.x - nrb;
    const uint32_t glane = threadIdx.x & 15u, grp = threadIdx.x >> 4;
    const uint32_t list = fgs_xcd_remap(blockIdx.x - nrb, ntb) * 16u + grp;
    if (list >= B * layers * tiles) return;  // whole groups leave together
    const RankRange rr = rank_range(list, N, tiles, layers, plane_start);
    const uint32_t b = rr.b, y = rr.t / tiles_x, x = rr.t - y * tiles_x;
    const unsigned long long *col = masks + ((size_t)b * lines + x) * w64p;
    const unsigned long long *row = masks + ((size_t)b * lines + tiles_x + y) * w64p;
    const uint32_t *ord = order + (size_t)b * N;
    uint32_t *pk = park[grp];
    uint32_t base = ranges[2 * list];  // next list slot
    const uint32_t w_end = min(w64p, (rr.r_hi + 63u) / 64u);
    // every bound below is the same for the 16 lanes of a group, so a group stays converged (groups of one wave may
    // diverge from each other: they share nothing)
    for (uint32_t c0 = rr.r_lo / 64u; c0 < w_end; c0 += 16u) {
        const uint32_t w = c0 + glane;
        const unsigned long long m = w < w_end ? clip_word(col[w] & row[w], w, rr.r_lo, rr.r_hi) : 0ull;
        const uint32_t cl = (uint32_t)__popcll(m);
        uint32_t inc = cl;  // inclusive scan over the group's lanes
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o, 16);
            if ((int)glane >= o) inc += v;
        }
        const uint32_t total = __shfl(inc, 15, 16);
        const uint32_t ex = inc - cl;
        for (uint32_t win = 0; win < total; win += MG_CAP) {
            uint32_t e = ex;
            unsigned long long mm = m;
            while (mm) {
                const uint32_t bit = (uint32_t)__ffsll((long long)mm) - 1u;
                mm &= mm - 1ull;
                if (e - win < MG_CAP) pk[e - win] = w * 64u + bit;  // unsigned: e < win wraps past MG_CAP
                ++e;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t n = min(MG_CAP, total - win);
            for (uint32_t i = glane; i < n; i += 64) {
                uint32_t rk[4], id[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) rk[u] = (i + 16u * u < n) ? pk[i + 16u * u] : 0u;
#pragma unroll
                for (int u = 0; u < 4; ++u) id[u] = ord[rk[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t pos = base + win + i + 16u * u;
                    if (i + 16u * u < n && pos < dcap) dup_ids[pos] = b * N + id[u];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        base += total;
    }
}

// The same emission with ONE BLOCK (four waves) per list, for launches of few, long lists (config 3 on 32 x 16 tiles: 4096
// lists of ~1000 entries; decoder-like: ~2500): the 256 lanes take consecutive rank words, a block scan (wave scans + the
// four wave totals) gives each lane its run of list slots, and the bit walk, the `order` gather and the stores of a list
// are spread over four waves instead of one -- a list's critical path is a quarter as long and the launch has four times
// the waves to hide the gather latency with.
template <int WPL>
__global__ __launch_bounds__(256) void k_mask_emit_block(uint32_t B, uint32_t N, uint32_t layers,
                                                         const uint32_t *__restrict__ plane_start, uint32_t tiles,
                                                         uint32_t tiles_x, uint32_t lines, uint32_t w64p, uint32_t nrb,
                                                         uint32_t bpi, uint32_t dcap,
                                                         const unsigned long long *__restrict__ masks,
                                                         const uint32_t *__restrict__ order,
                                                         const uint32_t *__restrict__ tile_count,
                                                         const uint32_t *__restrict__ ranges,
                                                         const uint32_t *__restrict__ bsum,
                                                         uint32_t *__restrict__ dup_ids, uint32_t *__restrict__ dup_off) {
    constexpr uint32_t CAP = 4u * ME_CAP;
    __shared__ uint32_t park[CAP];
    __shared__ uint32_t wtot[4];
    if (blockIdx.x < nrb) {  // duplicate offsets of one block of depth ranks (as in k_mask_emit)
        const uint32_t b = blockIdx.x / bpi, blk = blockIdx.x - b * bpi;
        const uint32_t r = blk * MB_RANKS + threadIdx.x;
        uint32_t g = 0, c = 0, tot;
        if (r < N) { g = b * N + order[b * N + r]; c = tile_count[g]; }
        const uint32_t ex = block_exclusive_scan_256(c, &tot);
        if (r < N) dup_off[g] = bsum[blockIdx.x] + ex;
        return;
    }
    const uint32_t ntb = gridDim.x - nrb;  // = number of lists
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tile = fgs_xcd_remap(blockIdx.x - nrb, ntb);
    const RankRange rr = rank_range(tile, N, tiles, layers, plane_start);
    const uint32_t b = rr.b, y = rr.t / tiles_x, x = rr.t - y * tiles_x;
    const unsigned long long *col = masks + ((size_t)b * lines + x) * w64p;
    const unsigned long long *row = masks + ((size_t)b * lines + tiles_x + y) * w64p;
    const uint32_t *ord = order + (size_t)b * N;
    uint32_t base = ranges[2 * tile];  // next list slot
    const uint32_t w_end = min(w64p, (rr.r_hi + 63u) / 64u);
    for (uint32_t c0 = (rr.r_lo / 64u) / (256u * WPL) * (256u * WPL); c0 < w_end; c0 += 256u * WPL) {
        const uint32_t w0 = c0 + tid * WPL;
        unsigned long long m[WPL];
#pragma unroll
        for (int k = 0; k < WPL; ++k) m[k] = 0ull;
        if (w0 < w_end) {
            mask_words<WPL>(col, row, w0, m);
#pragma unroll
            for (int k = 0; k < WPL; ++k) m[k] = clip_word(m[k], w0 + k, rr.r_lo, rr.r_hi);
        }
        uint32_t cl = 0;
#pragma unroll
        for (int k = 0; k < WPL; ++k) cl += (uint32_t)__popcll(m[k]);
        uint32_t inc = cl;  // inclusive wave scan of the lanes' entry counts
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += v;
        }
        if (lane == 63u) wtot[wave] = inc;
        __syncthreads();
        uint32_t wpre = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) {
            const uint32_t t = wtot[w];
            wpre += w < wave ? t : 0u;
            total += t;
        }
        const uint32_t ex = wpre + inc - cl;
        // flush windows of CAP entries (one window unless a tile collects more than CAP of these ranks)
        for (uint32_t win = 0; win < total; win += CAP) {
            uint32_t e = ex;
#pragma unroll
            for (int k = 0; k < WPL; ++k) {
                unsigned long long mm = m[k];
                const uint32_t rank0 = (w0 + k) * 64u;
                while (mm) {
                    const uint32_t bit = (uint32_t)__ffsll((long long)mm) - 1u;
                    mm &= mm - 1ull;
                    if (e - win < CAP) park[e - win] = rank0 + bit;  // unsigned: e < win wraps past CAP
                    ++e;
                }
            }
            __syncthreads();
            const uint32_t n = min(CAP, total - win);
            for (uint32_t i = tid; i < n; i += 1024) {
                uint32_t rk[4], id[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) rk[u] = (i + 256u * u < n) ? park[i + 256u * u] : 0u;
#pragma unroll
                for (int u = 0; u < 4; ++u) id[u] = ord[rk[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t pos = base + win + i + 256u * u;
                    if (i + 256u * u < n && pos < dcap) dup_ids[pos] = b * N + id[u];
                }
            }
            __syncthreads();
        }
        base += total;
        __syncthreads();  // wtot is rewritten by the next chunk
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
// (TO_TILES: 256 threads x 4 consecutive tiles)

__device__ __forceinline__ uint32_t length_bucket(uint32_t len) {
    // 63 - quarter-octave of the length: longer lists get smaller bucket numbers (4 buckets per power of two)
    if (len == 0) return 63u;
    const uint32_t lg = 31u - (uint32_t)__clz((int)len);
    const uint32_t frac = lg >= 2u ? (len >> (lg - 2u)) & 3u : (len << (2u - lg)) & 3u;
    const uint32_t q = lg * 4u + frac;  // 0 .. 127
    return q >= 62u ? 0u : 62u - q;
}

// XCD groups of the launch order (ngroups = 8, blend forward only): group g = the tiles whose index lies in the g-th of the
// eight contiguous ranges fgs_xcd_remap deals to XCD g -- for a batch of 8 images that is one image per XCD, for fewer
// images a band of tile rows.  tile_order then holds group 0's tiles heavy-first, then group 1's, ...; the forward reads
// tile_order[fgs_xcd_remap(blockIdx.x, ntiles)], so XCD g walks ITS group heavy-first and the records (`rec`, 48 B per
// Gaussian, 1.5 MB per config-3 image) its tiles gather stay in that XCD's 4 MB L2 instead of every L2 streaming the
// records of all images (12.6 MB): the counting sort simply runs over 64 x ngroups columns, column = 64 g + bucket.
__device__ __forceinline__ uint32_t tile_group(uint32_t t, uint32_t ntiles, uint32_t ngroups) {
    if (ngroups == 1u) return 0u;
    const uint32_t q = ntiles >> 3, r = ntiles & 7u, split = r * (q + 1u);
    return t < split ? t / (q + 1u) : r + (t - split) / (q ? q : 1u);
}
constexpr uint32_t TO_MAX_COLS = 512;  // 64 length buckets x up to 8 XCD groups

__device__ __forceinline__ uint32_t tile_len(const uint32_t *__restrict__ lens, const uint32_t *__restrict__ ranges,
                                             uint32_t t) {
    if (lens) return lens[t];
    const uint2 r = reinterpret_cast<const uint2 *>(ranges)[t];
    return r.y - r.x;
}

__global__ __launch_bounds__(256) void k_tile_pre(uint32_t ntiles, const uint32_t *__restrict__ ranges,
                                                  const uint32_t *__restrict__ lens, uint32_t seg_len,
                                                  unsigned long long *__restrict__ pre64,
                                                  uint32_t *__restrict__ bhist, uint32_t ngroups) {
    __shared__ uint32_t hist[TO_MAX_COLS];
    __shared__ unsigned long long wsum[4];
    const uint32_t ncols = 64u * ngroups;
    for (uint32_t i = threadIdx.x; i < ncols; i += 256) hist[i] = 0;
    __syncthreads();
    unsigned long long v = 0;
    const uint32_t t0 = blockIdx.x * TO_TILES + threadIdx.x * 4u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t t = t0 + k;
        if (t < ntiles) {
            const uint32_t len = tile_len(lens, ranges, t);
            v += ((unsigned long long)len << 32) | ((len + seg_len - 1) / seg_len);
            atomicAdd(&hist[64u * tile_group(t, ntiles, ngroups) + length_bucket(len)], 1u);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) pre64[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    for (uint32_t i = threadIdx.x; i < ncols; i += 256) bhist[blockIdx.x * ncols + i] = hist[i];
}

__global__ __launch_bounds__(256) void k_tile_post(uint32_t ntiles, uint32_t *__restrict__ ranges,
                                                   const uint32_t *__restrict__ lens,
                                                   uint32_t *__restrict__ tile_order, uint32_t *__restrict__ seg_off,
                                                   uint32_t *__restrict__ seg_tile, uint32_t *__restrict__ counters,
                                                   uint32_t seg_len, uint32_t fwd_variant,
                                                   const unsigned long long *__restrict__ pre64,
                                                   const uint32_t *__restrict__ bhist, uint32_t ngroups) {
    __shared__ uint32_t bucket_pos[TO_MAX_COLS];  // next slot of every (group, bucket) column for this block's tiles
    __shared__ unsigned long long wsum[4], carry_sh, total_sh;
    __shared__ uint32_t col_before[TO_MAX_COLS], col_total[TO_MAX_COLS];
    const uint32_t ncols = 64u * ngroups;
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
    // (2) bucket columns: thread (bucket b = tid & 63, quarter q = tid >> 6) sums rows q, q + 4, ... of every group's
    // column 64 g + b
    {
        const uint32_t b = threadIdx.x & 63u;
        __shared__ uint32_t pb[4][TO_MAX_COLS], pa[4][TO_MAX_COLS];
        if (ngroups == 1u) {
            uint32_t before = 0, all = 0;
            for (uint32_t k0 = threadIdx.x >> 6; k0 < nblk; k0 += 32) {  // eight loads in flight (rows k0, k0 + 4, ...)
                uint32_t h[8];
#pragma unroll
                for (uint32_t u = 0; u < 8; ++u) h[u] = k0 + 4u * u < nblk ? bhist[(size_t)(k0 + 4u * u) * 64u + b] : 0u;
#pragma unroll
                for (uint32_t u = 0; u < 8; ++u) {
                    all += h[u];
                    if (k0 + 4u * u < blk) before += h[u];
                }
            }
            pb[threadIdx.x >> 6][b] = before; pa[threadIdx.x >> 6][b] = all;
        } else {
            // 64 x ngroups columns, few rows (the blend path has <= a few dozen blocks of 1024 tiles): a thread owns the
            // columns tid, tid + 256 and walks the rows, all of a round's loads in flight together
            for (uint32_t col = threadIdx.x; col < ncols; col += 256) { pb[1][col] = pb[2][col] = pb[3][col] = 0; pa[1][col] = pa[2][col] = pa[3][col] = 0; }
            uint32_t before[2] = {0, 0}, all[2] = {0, 0};
            // many blocks (the layered lists of the splat renderers): when a group is a whole number of blocks, column (g, bucket) is
            // non-zero in group g's rows only -- an eighth of the rows to walk
            const bool aligned = ngroups == 8u && (ntiles & 7u) == 0u && ((ntiles >> 3) % TO_TILES) == 0u && ncols == 512u;
            const uint32_t rpg = aligned ? nblk / 8u : nblk;  // rows per group
            const uint32_t r0[2] = {aligned ? (threadIdx.x >> 6) * rpg : 0u, aligned ? ((threadIdx.x >> 6) + 4u) * rpg : 0u};
            for (uint32_t k0 = 0; k0 < rpg; k0 += 4) {
                uint32_t h[2][4];
#pragma unroll
                for (uint32_t c = 0; c < 2; ++c)
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const uint32_t col = threadIdx.x + 256u * c;
                        h[c][u] = (k0 + u < rpg && col < ncols) ? bhist[(size_t)(r0[c] + k0 + u) * ncols + col] : 0u;
                    }
#pragma unroll
                for (uint32_t c = 0; c < 2; ++c)
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        all[c] += h[c][u];
                        if (r0[c] + k0 + u < blk) before[c] += h[c][u];
                    }
            }
#pragma unroll
            for (uint32_t c = 0; c < 2; ++c)
                if (threadIdx.x + 256u * c < ncols) { pb[0][threadIdx.x + 256u * c] = before[c]; pa[0][threadIdx.x + 256u * c] = all[c]; }
        }
        __syncthreads();
        for (uint32_t col = threadIdx.x; col < ncols; col += 256) {
            col_before[col] = (pb[0][col] + pb[1][col]) + (pb[2][col] + pb[3][col]);
            col_total[col] = (pa[0][col] + pa[1][col]) + (pa[2][col] + pa[3][col]);
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // exclusive scan of the column totals: one wave, ncols / 64 columns per lane
            const uint32_t per = ncols / 64u;  // = ngroups
            uint32_t mine = 0;
            for (uint32_t i = 0; i < per; ++i) mine += col_total[threadIdx.x * per + i];
            uint32_t x = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(x, o, 64);
                if (threadIdx.x >= (uint32_t)o) x += y;
            }
            uint32_t run = x - mine;
            for (uint32_t i = 0; i < per; ++i) {
                const uint32_t col = threadIdx.x * per + i;
                bucket_pos[col] = run + col_before[col];
                run += col_total[col];
            }
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
        tile_order[atomicAdd(&bucket_pos[64u * tile_group(t, ntiles, ngroups) + length_bucket(len[k])], 1u)] = t;
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
                              uint32_t fwd_variant, uint32_t *scratch_words, hipStream_t st, uint32_t ngroups = 1) {
    const uint32_t nblk = (ntiles + TO_TILES - 1) / TO_TILES;
    unsigned long long *pre64 = reinterpret_cast<unsigned long long *>(scratch_words);
    uint32_t *bhist = scratch_words + 2 * (size_t)nblk;  // [nblk][64 * ngroups]
    // (k_tile_post walking all lengths itself instead of this launch: 21.6 us against 4.8 + 9.2 at config 3 -- the
    // bucket histogram's same-address LDS atomics)
    hipLaunchKernelGGL(k_tile_pre, dim3(nblk), dim3(256), 0, st, ntiles, ranges, lens, seg_len, pre64, bhist, ngroups);
    FGS_LAUNCH_CHECK("k_tile_pre");
    hipLaunchKernelGGL(k_tile_post, dim3(nblk), dim3(256), 0, st, ntiles, ranges, lens, tile_order, seg_off, seg_tile,
                       counters, seg_len, fwd_variant, pre64, bhist, ngroups);
    FGS_LAUNCH_CHECK("k_tile_post");
    return FGS_OK;
}

__global__ __launch_bounds__(256) void k_count_pairs(uint32_t total, const float *__restrict__ rec,
                                                     const uint32_t *__restrict__ tile_count,
                                                     unsigned long long *__restrict__ out) {
    // grid-stride, ONE device-scope atomic per block (they serialise at ~50 ns each: one per wave of a 262 144-Gaussian
    // launch took 52 us)
    unsigned long long p = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        if (tile_count[i] != 0) {
            const uint32_t bbx = __float_as_uint(rec[(size_t)i * FGS_REC_FLOATS + R_BBX]);
            const uint32_t bby = __float_as_uint(rec[(size_t)i * FGS_REC_FLOATS + R_BBY]);
            p += (unsigned long long)((bbx >> 16) - (bbx & 0xFFFFu)) * ((bby >> 16) - (bby & 0xFFFFu));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) p += __shfl_down(p, o, 64);
    __shared__ unsigned long long wsum[4];
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = p;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        if (t) atomicAdd(out, t);
    }
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
    uint32_t *ks = nullptr, *vs = nullptr;
    const bool layered_direct = p.direct_binning && p.layers > 1;
    int rc = FGS_OK;
    uint32_t *plane_keys = nullptr;  // sorted plane ids per image (layered direct binning)
    if (!p.depth_ordered) {
        // splat renderers: a sum over a list does not depend on its order -- no depth sort (8 launches, 40-50 us whatever the size).
        // Layered lists: ONE stable pass over the plane ids, payload = index (2 launches instead of 8 + 1 + 2), so `order` lists plane
        // 0's Gaussians in index order, then plane 1's, ...; otherwise `order` is the identity.
        if (layered_direct) {
            uint32_t pbits = 0;
            while ((1u << pbits) < (uint32_t)p.layers) ++pbits;
            uint32_t *vs2;
            if ((rc = fgs_launch_radix_sort(keys0, vals0, keys1, vals1, order, &plane_keys, &vs2, N, nullptr, N, N, B, pbits, hist, st,
                                            reinterpret_cast<const uint32_t *>(saved + p.s_layer), N, nullptr, 0, p.d.sort_mode >> 1)))
                return rc;
        } else {
            hipLaunchKernelGGL(k_index_order, dim3(nblk), dim3(256), 0, st, total, N, order);
            FGS_LAUNCH_CHECK("k_index_order");
        }
    } else if ((rc = fgs_launch_radix_sort(keys0, vals0, keys1, vals1, layered_direct ? nullptr : order, &ks, &vs, N, nullptr, N, N,
                                           B, 32, hist, st, depth_key, N,
                                           (layered_direct || !(p.d.sort_mode & 1)) ? nullptr : reinterpret_cast<const uint32_t *>(saved + p.s_keybits),
                                           (N + 255u) / 256u, p.d.sort_mode >> 1))) {
        return rc;
    }
    if (p.depth_ordered && layered_direct) {
        // group the depth order by layer (depth plane) with ONE more stable pass over the plane ids: `order` then lists
        // plane 0's Gaussians in depth order, then plane 1's, ... and a (plane, tile) list is a rank RANGE of the masks
        uint32_t *kfree = ks == keys0 ? keys1 : keys0, *vfree = vs == vals0 ? vals1 : vals0;
        hipLaunchKernelGGL(k_plane_keys, dim3(nblk), dim3(256), 0, st, total, N, vs,
                           reinterpret_cast<const uint32_t *>(saved + p.s_layer), kfree);
        FGS_LAUNCH_CHECK("k_plane_keys");
        uint32_t pbits = 0;
        while ((1u << pbits) < (uint32_t)p.layers) ++pbits;
        uint32_t *ks2, *vs2;
        if ((rc = fgs_launch_radix_sort(kfree, vs, ks, vfree, order, &ks2, &vs2, N, nullptr, N, N, B, pbits, hist, st, nullptr, 0,
                                        nullptr, 0, p.d.sort_mode >> 1)))
            return rc;
        plane_keys = ks2;
    }
    fgs_stage_end(ST_DEPTH_SORT, st);
    const uint32_t ntiles_all = B * (uint32_t)p.layers * (uint32_t)p.tiles;
    uint32_t *tile_order = reinterpret_cast<uint32_t *>(saved + p.L.tile_order);
    uint32_t *seg_off = p.L.seg_capacity ? reinterpret_cast<uint32_t *>(saved + p.L.seg_off) : nullptr;
    uint32_t *seg_tile = p.L.seg_capacity ? reinterpret_cast<uint32_t *>(saved + p.L.seg_tile) : nullptr;
    if (p.direct_binning) {
        // mask binning (see k_mask_build): 5 launches, none of them serial over Gaussians
        const uint32_t bpi = (N + MB_RANKS - 1) / MB_RANKS, nrb = B * bpi;
        const uint32_t tiles_x = (uint32_t)p.L.tiles_x, tiles_y = (uint32_t)p.L.tiles_y, lines = tiles_x + tiles_y;
        const uint32_t w64p = fgs_mask_words(N), layers = (uint32_t)p.layers;
        // masks [B][lines][w64p] and list lengths [B * layers * tiles] in the two sort buffers the (plane) keys are NOT in
        unsigned long long *masks = reinterpret_cast<unsigned long long *>(plane_keys == keys0 ? keys1 : keys0);
        uint32_t *lens = vals0;
        uint32_t *plane_start = reinterpret_cast<uint32_t *>(scratch + p.s_plane);  // [B][layers + 1]
        uint32_t *dup_off = reinterpret_cast<uint32_t *>(saved + p.L.dup_off);
        const uint32_t ntb = (ntiles_all + 3) / 4;  // four tiles (waves) per block
        // rank words a list spans: all of them, or its plane's share
        const uint32_t wspan = layers > 1 ? (w64p + layers - 1) / layers + 2 : w64p;
        const int wpl = wspan <= 64 ? 1 : (wspan <= 128 ? 2 : (wspan <= 256 ? 4 : 8));
        const uint32_t npb = layers > 1 ? (B * (layers + 1) + 255) / 256 : 0u;  // blocks that find the plane ranges
        fgs_stage_begin(ST_DUP_EMIT, st);
        hipLaunchKernelGGL(k_mask_build, dim3(nrb + npb), dim3(256), 0, st, N, tiles_x, tiles_y, bpi, w64p, order,
                           tile_count, rec, masks, bsum, nrb, layers, plane_keys, plane_start, (uint32_t)p.tile_w);
        FGS_LAUNCH_CHECK("k_mask_build");
#define FGS_MASK_COUNT(W) hipLaunchKernelGGL(k_mask_count<W>, dim3(ntb + 1), dim3(256), 0, st, B, N, layers, plane_start, \
                                             (uint32_t)p.tiles, tiles_x, lines, w64p, nrb, masks, lens, bsum, counters, dcap)
        // many short lists over <= 16 rank words each (the ASM renderer's depth planes): 16 lanes per list
        const bool grouped = wspan <= 16u && ntiles_all >= FGS_GROUP_MIN_LISTS;
        const uint32_t ngb = (ntiles_all + 15) / 16;  // sixteen lists per block
        if (grouped)
            hipLaunchKernelGGL(k_mask_count_group, dim3(ngb + 1), dim3(256), 0, st, B, N, layers, plane_start, (uint32_t)p.tiles,
                               tiles_x, lines, w64p, nrb, masks, lens, bsum, counters, dcap);
        else if (wpl == 1) FGS_MASK_COUNT(1); else if (wpl == 2) FGS_MASK_COUNT(2); else if (wpl == 4) FGS_MASK_COUNT(4); else FGS_MASK_COUNT(8);
#undef FGS_MASK_COUNT
        FGS_LAUNCH_CHECK("k_mask_count");
        fgs_stage_end(ST_DUP_EMIT, st);
        fgs_stage_begin(ST_TILE_RANGES, st);
        if ((rc = launch_tile_tables(ntiles_all, ranges, lens, tile_order, seg_off, seg_tile, counters,
                                     (uint32_t)p.L.seg_len, (uint32_t)p.fwd_variant, vals1, st, (uint32_t)p.order_groups)))
            return rc;
        fgs_stage_end(ST_TILE_RANGES, st);
        fgs_stage_begin(ST_TILE_SORT, st);
#define FGS_MASK_EMIT(W) hipLaunchKernelGGL(k_mask_emit<W>, dim3(nrb + ntb), dim3(256), 0, st, B, N, layers, plane_start, \
                                            (uint32_t)p.tiles, tiles_x, lines, w64p, nrb, bpi, dcap, masks, order,         \
                                            tile_count, ranges, bsum, dup_ids, dup_off)
#define FGS_MASK_EMIT_BLOCK(W) hipLaunchKernelGGL(k_mask_emit_block<W>, dim3(nrb + ntiles_all), dim3(256), 0, st, B, N, layers, \
                                                  plane_start, (uint32_t)p.tiles, tiles_x, lines, w64p, nrb, bpi, dcap, masks, \
                                                  order, tile_count, ranges, bsum, dup_ids, dup_off)
        // few lists (<= 8192: the blend path's frames; the ASM renderer's (image, plane, tile) lists are many and short)
        // over more than 128 rank words (N > 8192 per image): a block per list, a quarter of the rank words per wave.
        // Measured: config 3 (4096 lists of ~1000 entries) 40 -> 28 us, decoder-like (~2500 entries) 119 -> 38 us;
        // config 2 (N = 8192, ~130 entries per list) 13 -> 17 us, so it keeps the wave-per-list kernel.
        if (grouped) {
            hipLaunchKernelGGL(k_mask_emit_group, dim3(nrb + ngb), dim3(256), 0, st, B, N, layers, plane_start, (uint32_t)p.tiles,
                               tiles_x, lines, w64p, nrb, bpi, dcap, masks, order, tile_count, ranges, bsum, dup_ids, dup_off);
        } else if (ntiles_all <= FGS_EMIT_BLOCK_MAX_LISTS && wpl >= 4) {
            if (wpl == 4) FGS_MASK_EMIT_BLOCK(1); else FGS_MASK_EMIT_BLOCK(2);
        } else if (wpl == 1) FGS_MASK_EMIT(1); else if (wpl == 2) FGS_MASK_EMIT(2); else if (wpl == 4) FGS_MASK_EMIT(4); else FGS_MASK_EMIT(8);
#undef FGS_MASK_EMIT
#undef FGS_MASK_EMIT_BLOCK
        FGS_LAUNCH_CHECK("k_mask_emit");
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
                       (uint32_t)p.layers, (uint32_t)p.tile_w, reinterpret_cast<uint32_t *>(saved + p.L.dup_off), keys0, vals0);
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
                                 (uint32_t)p.L.seg_len, (uint32_t)p.fwd_variant, ks == keys0 ? keys1 : keys0, st,
                                 (uint32_t)p.order_groups)))
        return rc;
    fgs_stage_end(ST_TILE_RANGES, st);
    return FGS_OK;
}

int fgs_launch_count_pairs(const FgsPlan &p, const char *saved, uint64_t *out, hipStream_t st) {
    const uint32_t total = p.d.batch * p.d.num_gaussians;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(uint64_t), st);
    if (e != hipSuccess) { fgs_set_error("memset pairs: %s", hipGetErrorString(e)); return FGS_ELAUNCH; }
    uint32_t cgrid = (total + 255) / 256;
    if (cgrid > 64) cgrid = 64;
    hipLaunchKernelGGL(k_count_pairs, dim3(cgrid), dim3(256), 0, st, total,
                       reinterpret_cast<const float *>(saved + p.L.rec),
                       reinterpret_cast<const uint32_t *>(saved + p.L.tile_count),
                       reinterpret_cast<unsigned long long *>(out));
    FGS_LAUNCH_CHECK("k_count_pairs");
    return FGS_OK;
}
