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

__global__ __launch_bounds__(256) void k_sort_init(uint32_t total, uint32_t N,
                                                   const uint32_t *__restrict__ depth_key,
                                                   uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    keys[i] = depth_key[i];
    vals[i] = i % N;
}

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
__global__ __launch_bounds__(1024) void k_tile_order(uint32_t ntiles, const uint32_t *__restrict__ ranges,
                                                     uint32_t *__restrict__ tile_order,
                                                     uint32_t *__restrict__ seg_off, uint32_t *__restrict__ seg_tile,
                                                     uint32_t *__restrict__ counters) {
    __shared__ uint32_t hist[64];
    __shared__ uint32_t maxc;
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    // list lengths are read from global memory once and kept in LDS (when they fit) for the four sweeps below
    constexpr uint32_t LCAP = 8192;
    __shared__ uint32_t lcnt[LCAP];
    const bool cached = ntiles <= LCAP;
    auto len_of = [&](uint32_t t) -> uint32_t { return cached ? lcnt[t] : ranges[2 * t + 1] - ranges[2 * t]; };
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) maxc = 1;
    __syncthreads();
    uint32_t mymax = 0;
    for (uint32_t t = threadIdx.x; t < ntiles; t += 1024) {
        const uint2 r = reinterpret_cast<const uint2 *>(ranges)[t];
        if (cached) lcnt[t] = r.y - r.x;
        mymax = max(mymax, r.y - r.x);
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
    if (!seg_off) return;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < ntiles; base += 1024) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t n = t < ntiles ? (len_of(t) + FGS_SEG - 1) / FGS_SEG : 0u;
        uint32_t x = n;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (int i = 0; i < 16; ++i) { const uint32_t h = wsum[i]; wsum[i] = run; run += h; }
        }
        __syncthreads();
        const uint32_t off = carry + wsum[wave] + x - n;
        if (t < ntiles) {
            seg_off[t] = off;
            for (uint32_t k = 0; k < n; ++k) seg_tile[off + k] = t;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = off + n;
        __syncthreads();
    }
    if (threadIdx.x == 0) { seg_off[ntiles] = carry; counters[2] = carry; }
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
    hipLaunchKernelGGL(k_sort_init, dim3(nblk), dim3(256), 0, st, total, N, depth_key, keys0, vals0);
    FGS_LAUNCH_CHECK("k_sort_init");
    uint32_t *ks, *vs;
    int rc = fgs_launch_radix_sort(keys0, vals0, keys1, vals1, order, &ks, &vs, N, nullptr, N, N, B, 32, hist, st);
    if (rc) return rc;
    fgs_stage_end(ST_DEPTH_SORT, st);
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
    hipError_t e = hipMemsetAsync(ranges, 0, (size_t)B * p.layers * p.tiles * 2 * sizeof(uint32_t), st);
    if (e != hipSuccess) { fgs_set_error("memset ranges: %s", hipGetErrorString(e)); return FGS_ELAUNCH; }
    uint32_t rgrid = (dcap + 255) / 256;
    if (rgrid > 2048) rgrid = 2048;
    hipLaunchKernelGGL(k_tile_ranges, dim3(rgrid), dim3(256), 0, st, counters, ks, ranges);
    FGS_LAUNCH_CHECK("k_tile_ranges");
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, st, B * (uint32_t)p.layers * (uint32_t)p.tiles, ranges,
                       reinterpret_cast<uint32_t *>(saved + p.L.tile_order),
                       p.L.seg_capacity ? reinterpret_cast<uint32_t *>(saved + p.L.seg_off) : nullptr,
                       p.L.seg_capacity ? reinterpret_cast<uint32_t *>(saved + p.L.seg_tile) : nullptr, counters);
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
