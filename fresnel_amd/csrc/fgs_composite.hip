// Stage 3: front-to-back per-pixel alpha compositing, forward and backward.
// Replaces the reference's per-Gaussian Python loop (DR:582-667) and its autograd replay.
//
// Work decomposition (CDNA4, wave64): ONE WAVE owns one 16x16 tile of one image.  The tile is
// cut into four 8x8 sub-tiles; lane l owns pixel (l&7, l>>3) of EACH sub-tile, so a wave
// instruction covers exactly one sub-tile.  For every Gaussian of the tile's depth-sorted list
// the wave tests the Gaussian's integer bbox against the four sub-tile rectangles with scalar
// code and runs the per-pixel blend only for the sub-tiles it touches (wave-uniform branches:
// no divergence, 8x8 work granularity).  The list is staged through LDS 64 records at a time
// (lane j fetches record j with three 16-byte loads; the loop reads each record back once as
// three wave-uniform ds_read_b128 broadcasts and keeps it in registers across the sub-tiles).
//
// Reference semantics kept exactly: rectangular bbox support (DR:594-600), integer pixel
// coordinates (DR:603-607), alpha clamp [0, 0.99] (DR:647), additive accumulated alpha with
// T = 1 - A (DR:650-658), no early termination.
//
// Bound: VALU + transcendental (about 25 VALU + 1 v_exp_f32 per Gaussian-pixel forward, ~60
// backward); HBM traffic is the 52-byte id+record gather per duplicate plus per-pixel state.
#include <type_traits>
#include "fgs_internal.h"
#include "fgs_wave.h"

#if (defined(FGS_BWD_DYN_LDS) || defined(FGS_BWD_WIDE_WAVES) || defined(FGS_PHASE_SCAN) || defined(FGS_CKPT_NT) || defined(FGS_PHASE_WAVE_BLOCKS)) && !defined(FGS_EXPERIMENT_BUILD)
#error "work-split / timing switches of this unit are for experiment builds: python -m fresnel_amd.build --define ... (sets FGS_EXPERIMENT_BUILD; fgs_version() then says so)"
#endif

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr float FGS_SATURATION_EPS = 2.98023223876953125e-8f;  // 2^-25: 1 - T rounds to 1.0f below it
constexpr float NEG_HALF_LOG2E = -0.72134752044448170368f;  // exp(-m/2) = exp2(m * this)
constexpr float ALPHA_MAX = 0.99f;                          // DR:647
constexpr float INV_ALPHA_MAX = 1.0f / 0.99f;
constexpr float PHASE_KAPPA = 2.0f * 3.14159f;              // DR:642 uses the literal 3.14159

// cos / sin of x = kappa * pd, pd in [0, 0.5], i.e. x in [0, pi]: polynomials in y = x - pi/2 (|y| <= pi/2), all plain FMAs.  The
// hardware v_cos_f32 / v_sin_f32 (__cosf / __sinf) are only good to ~1e-5 absolute, which showed up as 1e-5 image error and > 1e-4
// gradient error at phase_amplitude 0.6.  Round 5: MINIMAX coefficients (fit on [0, (pi/2)^2] in y^2 with the constant term held at
// 1; approximation error 4.8e-9 / 2.4e-10, fp32 evaluation error 1.3e-7 / 1.2e-7 over the interval -- that of the Taylor
// polynomials of degree 11 / 12 they replace) of degree 9 / 10: one FMA less per evaluation, three per list entry of the backward.
__device__ __forceinline__ float phase_cos(float x) {
    const float y = x - 1.57079632679489661923f, y2 = y * y;  // cos(x) = -sin(y),  sin(y) = y (1 + y^2 p(y^2))
    float p = 2.6089300414764439e-6f;
    p = p * y2 - 1.9811111665926607e-4f;
    p = p * y2 + 8.3330881539788702e-3f;
    p = p * y2 - 1.6666660461917662e-1f;
    return -(y + y * (y2 * p));
}
__device__ __forceinline__ float phase_sin(float x) {
    const float y = x - 1.57079632679489661923f, y2 = y * y;  // sin(x) = cos(y) = 1 + y^2 p(y^2)
    float p = -2.6077082524225105e-7f;
    p = p * y2 + 2.4761885011562826e-5f;
    p = p * y2 - 1.3888403485377396e-3f;
    p = p * y2 + 4.1666640726419513e-2f;
    p = p * y2 - 4.9999999549521157e-1f;
    return 1.0f + y2 * p;
}
constexpr int CH = 64;                                      // records per LDS chunk (one per lane)

struct TileCtx {
    uint32_t tile, b, tx, ty, X0, Y0, start, end;
};

__device__ __forceinline__ TileCtx tile_ctx_of(uint32_t tile, uint32_t tiles, uint32_t tiles_x,
                                               const uint32_t *__restrict__ ranges, uint32_t tile_w = FGS_TILE) {
    TileCtx c;
    c.tile = tile;
    c.b = c.tile / tiles;
    const uint32_t t = c.tile - c.b * tiles;
    c.ty = t / tiles_x;
    c.tx = t - c.ty * tiles_x;
    c.X0 = c.tx * tile_w;
    c.Y0 = c.ty * FGS_TILE;
    c.start = ranges[2 * c.tile];
    c.end = ranges[2 * c.tile + 1];
    return c;
}

__device__ __forceinline__ TileCtx tile_ctx(uint32_t tiles, uint32_t tiles_x,
                                            const uint32_t *__restrict__ tile_order,
                                            const uint32_t *__restrict__ ranges, uint32_t tile_w = FGS_TILE) {
    return tile_ctx_of(tile_order ? tile_order[blockIdx.x] : blockIdx.x, tiles, tiles_x, ranges, tile_w);
}


// Row-split forward of the blend path (FgsDims.saturation_skip and fwd_variant < 0; the default is the depth-split
// k_blend_fwd_parts below, the phase path has its own kernels, k_phase_fwd / k_phase_bwd).  FWD_WAVES waves per tile: with 2,
// wave w owns sub-tiles 2w and 2w+1, the upper / lower half of the tile (the forward has no cross-lane reduction, so the
// split is free and halves the serial length of the longest lists); 1 when the launch has enough tiles to fill the chip
// several times over.  All waves of a block share the LDS-staged chunk of the list.
//
// Per-record work is decided once, at staging time, in parallel over the chunk: stage_decode (fgs_wave.h) leaves a flags
// word (touched sub-tiles, ...) and 32 pixel bits; in the list loop a lane turns its column / row bit into an all-ones /
// zero mask with v_bfe_i32 and and-s it onto G -- no per-pixel compare / select (issue costs: DESIGN.md section 4).
// SKIP: FgsDims.saturation_skip (separate instantiation).

template <int FWD_WAVES, bool SKIP>
__global__ __launch_bounds__(64 * FWD_WAVES) void k_composite_fwd(
    uint32_t tiles, uint32_t tiles_x, uint32_t W, uint32_t H, float bg0, float bg1, float bg2,
    const uint32_t *__restrict__ tile_order, const uint32_t *__restrict__ ranges,
    const uint32_t *__restrict__ dup_ids, const float *__restrict__ rec,
    float *__restrict__ pix_state, float *__restrict__ out_rgb,
    float *__restrict__ out_depth, const uint32_t *__restrict__ seg_off, float *__restrict__ seg_ckpt, float t_eps) {
    // records per LDS chunk: one per thread, but never more than a depth segment on the blend path
    constexpr int FCH = (64 * FWD_WAVES > FGS_SEG) ? FGS_SEG : 64 * FWD_WAVES;
    static_assert(FGS_SEG % FCH == 0, "segment boundaries must fall on chunk boundaries");
    __shared__ float4 sh0[FCH], sh1[FCH], sh2[FCH];
    const TileCtx c = tile_ctx(tiles, tiles_x, tile_order, ranges);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = FWD_WAVES > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0u;
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    constexpr int NS = 4 / FWD_WAVES;  // sub-tiles per wave
    // running transmittance T (w = alpha T; T -= w)
    float T[NS], Cr[NS], Cg[NS], Cb[NS], Dm[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { T[s] = 1.0f; Cr[s] = 0; Cg[s] = 0; Cb[s] = 0; Dm[s] = 0; }
    // sub-tile rows / columns per wave: 4 sub-tiles = 2 x 2, 2 = one row of two, 1 = a single sub-tile
    constexpr int NR = NS == 4 ? 2 : 1, NC = NS == 1 ? 1 : 2;
    const uint32_t row0 = NS == 4 ? 0u : (NS == 2 ? wave : wave >> 1), col0 = NS == 1 ? (wave & 1u) : 0u;
    // this lane's column / row bit in the staged pixel bits
    const uint32_t shx = lx + 8u * col0, shy = 16u + ly + 8u * row0;
    uint32_t alive = 15u;  // sub-tiles still being composited (saturation_skip)
    uint32_t live_segments = (c.end - c.start + FGS_SEG - 1) / FGS_SEG;
    float fx0 = (float)(c.X0 + lx + 8u * col0), fx1 = (float)(c.X0 + lx + 8u * col0 + 8u), fy0 = (float)(c.Y0 + ly + 8u * row0);
    asm("" : "+v"(fx0), "+v"(fx1), "+v"(fy0));  // hoisted for good: no v_cvt in the list loop
    for (uint32_t base = c.start; base < c.end; base += FCH) {
        const uint32_t n = min((uint32_t)FCH, c.end - base);
        if (base != c.start && ((base - c.start) % FGS_SEG) == 0) {
            // state in front of this depth segment: the backward work unit (tile, segment) restarts from it
            float *ck = seg_ckpt + ((size_t)seg_off[c.tile] + (base - c.start) / FGS_SEG) * (5 * 256) + lane;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t sg = wave * NS + s;
                ck[(0 * 4 + sg) * 64] = Cr[s]; ck[(1 * 4 + sg) * 64] = Cg[s]; ck[(2 * 4 + sg) * 64] = Cb[s];
                ck[(3 * 4 + sg) * 64] = 1.0f - T[s]; ck[(4 * 4 + sg) * 64] = Dm[s];
            }
        }
        if (SKIP && ((base - c.start) % FGS_SEG) == 0) {
            // FgsDims.saturation_skip: at every segment boundary, sub-tiles whose transmittance (as the backward
            // will see it, 1 - A) is below t_eps for all 64 pixels stop being composited; when the whole tile is
            // saturated the list walk ends and the number of live segments is left for the backward
            uint32_t aw = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float acc = 1.0f - T[s];
                asm("" : "+v"(acc));  // keep 1 - (1 - T): what the checkpoint stores, not T itself
                if (__ballot(1.0f - acc >= t_eps) != 0ull) aw |= 1u << (wave * NS + s);
            }
            alive = aw;
            if (!__syncthreads_or(aw != 0u)) { live_segments = (base - c.start) / FGS_SEG; break; }
        }
        if (threadIdx.x < n) {
            const uint32_t gid = dup_ids[base + threadIdx.x];
            const float4 *r = reinterpret_cast<const float4 *>(rec + (size_t)gid * FGS_REC_FLOATS);
            float4 q0 = r[0], q1 = r[1], q2 = r[2];
            q0.z *= NEG_HALF_LOG2E; q0.w *= NEG_HALF_LOG2E; q1.x *= NEG_HALF_LOG2E;
            const uint32_t bbx = __float_as_uint(q2.z), bby = __float_as_uint(q2.w);
            uint32_t flags, bits;  // flags: touched sub-tiles (none when the opacity is negative); bits: pixel masks
            stage_decode(c.X0, c.Y0, bbx, bby, q1.y, flags, bits);
            q2.z = __uint_as_float(bits); q2.w = __uint_as_float(flags);
            sh0[threadIdx.x] = q0; sh1[threadIdx.x] = q1; sh2[threadIdx.x] = q2;
        }
        __syncthreads();
        for (uint32_t j = 0; j < n; ++j) {
            const float4 q0 = sh0[j], q1 = sh1[j], q2 = sh2[j];
            {
                // Non-phase blend.  Most list entries touch only one or two of a wave's sub-tiles, so nothing is
                // precomputed beyond the row terms; bbox membership = the lane's column / row bit of the staged
                // pixel bits as an all-ones / zero mask (v_bfe_i32) and-ed onto G (no compare / select).
                const uint32_t msk = __builtin_amdgcn_readfirstlane(__float_as_uint(q2.w)) & (SKIP ? alive : 15u);
                if (!(msk & (((1u << NS) - 1u) << (wave * NS)))) continue;
                const uint32_t bits = __float_as_uint(q2.z);
#pragma unroll
                for (int row = 0; row < NR; ++row) {
                    if (NS == 4 && !((msk >> (2 * row)) & 3u)) continue;
                    const float dy = (NS == 4 && row) ? fy0 + 8.0f - q0.y : fy0 - q0.y;
                    const float bdy = q0.w * dy, cyy = (q1.x * dy) * dy;
                    const uint32_t my = (uint32_t)__builtin_amdgcn_sbfe((int)bits, shy + 8u * row, 1);
#pragma unroll
                    for (int col = 0; col < NC; ++col) {
                        const int s = NC * row + col;  // index into this wave's state
                        const uint32_t sg = wave * NS + s;
                        if (NS > 1 && !((msk >> sg) & 1u)) continue;  // scalar branch: sub-tile not touched
                        const float dx = (col ? fx1 : fx0) - q0.x;
                        const float t = q0.z * dx + bdy;
                        const uint32_t mk = my & (uint32_t)__builtin_amdgcn_sbfe((int)bits, shx + 8u * col, 1);
                        const float G = __uint_as_float(__float_as_uint(__builtin_amdgcn_exp2f(t * dx + cyy)) & mk);
                        const float alpha = fminf(G * q1.y, 0.99f);  // opacity >= 0 here: no lower clamp needed
                        const float w = alpha * T[s];
                        Cr[s] += w * q1.z; Cg[s] += w * q1.w; Cb[s] += w * q2.x; Dm[s] += w * q2.y;
                        T[s] -= w;
                    }
                }
                continue;
            }
        }
        __syncthreads();
    }
    // live-segment count for the backward, parked in the (otherwise unused) checkpoint slot of segment 0
    if (SKIP && threadIdx.x == 0 && c.end > c.start)
        reinterpret_cast<uint32_t *>(seg_ckpt + (size_t)seg_off[c.tile] * (5 * 256))[0] = live_segments;
    const size_t HW = (size_t)W * H;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t sg = wave * NS + s;
        const uint32_t px = c.X0 + 8u * (sg & 1) + lx, py = c.Y0 + 8u * (sg >> 1) + ly;
        if (px < W && py < H) {
            const size_t o = (size_t)py * W + px;
            float *ps = pix_state + (size_t)c.b * 6 * HW + o;
            const float Af = 1.0f - T[s];
            ps[0] = Cr[s]; ps[HW] = Cg[s]; ps[2 * HW] = Cb[s]; ps[3 * HW] = Af; ps[4 * HW] = Dm[s];
            const float T = 1.0f - Af;  // (as the backward will see it)
            float *img = out_rgb + (size_t)c.b * 3 * HW + o;
            img[0] = fminf(fmaxf(Cr[s] + T * bg0, 0.0f), 1.0f);
            img[HW] = fminf(fmaxf(Cg[s] + T * bg1, 0.0f), 1.0f);
            img[2 * HW] = fminf(fmaxf(Cb[s] + T * bg2, 0.0f), 1.0f);
            out_depth[(size_t)c.b * HW + o] = Dm[s];
        }
    }
}

// Depth-split forward of the blend path.  Alpha compositing is associative: a list cut into parts can be
// composited part by part from T = 1 and the partial results composed afterwards,
//     (C, T) o (C', T') = (C + T C', T T').
// NP waves per tile; wave p composites list part p (a whole number of FGS_SEG segments, ALL FOUR sub-tiles per
// lane), staging its own 64-record chunks in wave-private LDS -- no block barrier inside the list walk.  Compared
// with splitting a tile by sub-tile rows (k_composite_fwd) every list entry is read from LDS by ONE wave instead of
// all of them and its row terms are formed once, at the same serial length per wave.
// Checkpoints: a wave stores its LOCAL state at the segment boundaries inside its part.  After the walk wave 0
// composes the parts in order and leaves the ABSOLUTE state in the checkpoint slot of every part's first segment;
// the backward unit of a later segment of part p > 0 re-bases its local checkpoint with that slot
// (k_composite_bwd, `fwd_parts`).  Hand-over between the waves goes through the checkpoint slots themselves
// (the slot of the next part's first segment; the last part uses the tile's slot 0, which nothing else reads).
struct BlendState { float Cr, Cg, Cb, T, D; };
__device__ __forceinline__ BlendState compose(const BlendState &a, const BlendState &b) {
    return {a.Cr + a.T * b.Cr, a.Cg + a.T * b.Cg, a.Cb + a.T * b.Cb, a.T * b.T, a.D + a.T * b.D};
}
// [5][NS][64] slot (NS = 4 sub-tiles of a 16 x 16 tile, 8 of a 32 x 16 tile), this lane's cell; PL = NS * 64
// Checkpoints are a write-once / read-once stream (307 MB per config-3 step against 12.6 MB of records).  Non-temporal
// stores / loads for them (FGS_CKPT_NT=1) were measured in round 3 and are OFF: WRITE_SIZE of the forward ROSE from 450
// to 547 MB per launch (the nt stores reach the fabric as more, smaller writes), the backward got 1 % slower, and what
// actually evicted `rec` from L2 was the launch order, not the checkpoint stream (see order_groups above: FETCH_SIZE
// 330 -> 62 MB).
#ifndef FGS_CKPT_NT
#define FGS_CKPT_NT 0
#endif
__device__ __forceinline__ void nt_store(float *p, float v) {
#if FGS_CKPT_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ float nt_load(const float *p) {
#if FGS_CKPT_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <int PL = 256>
__device__ __forceinline__ void ckpt_store(float *ck, const BlendState &v) {
    nt_store(ck, v.Cr); nt_store(ck + PL, v.Cg); nt_store(ck + 2 * PL, v.Cb); nt_store(ck + 3 * PL, 1.0f - v.T); nt_store(ck + 4 * PL, v.D);
}
template <int PL = 256>
__device__ __forceinline__ BlendState ckpt_load(const float *ck) {
    return {nt_load(ck), nt_load(ck + PL), nt_load(ck + 2 * PL), 1.0f - nt_load(ck + 3 * PL), nt_load(ck + 4 * PL)};
}

// WIDE = 1: 32 x 16 tiles (FgsSavedLayout.tile_w = 32; the backward then has eight sub-tiles per lane and ~0.6x as many
// list entries, reductions and gradient rows).  The forward keeps its 16 x 16 register budget -- holding eight sub-tiles per
// lane here cost 99 VGPRs, 4-5 waves per SIMD and +5 ... 35 % -- by giving every list part TWO waves, one per 16 x 16 half
// of the tile: a wave stages the part's records with the flags of ITS half and skips the entries that do not touch it, so
// its work is that of the 16 x 16 tile's list plus one LDS read and a branch per skipped entry.
template <int NP, int WIDE>
__global__ __launch_bounds__(64 * NP * (1 + WIDE)) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_blend_fwd_parts(  // 64 VGPRs: -2.6 %
    uint32_t tiles, uint32_t tiles_x, uint32_t W, uint32_t H, float bg0, float bg1, float bg2,
    const uint32_t *__restrict__ tile_order, const uint32_t *__restrict__ ranges,
    const uint32_t *__restrict__ dup_ids, const float *__restrict__ rec, float *__restrict__ pix_state,
    float *__restrict__ out_rgb, float *__restrict__ out_depth, const uint32_t *__restrict__ seg_off,
    float *__restrict__ seg_ckpt, uint32_t seg_len, uint32_t order_groups) {
    constexpr int NW = NP * (1 + WIDE);                   // waves per block
    constexpr int TSX = WIDE ? 4 : 2;                     // sub-tile columns of the whole tile
    constexpr int PL = 2 * TSX * 64, SLOT = 5 * PL;       // floats per checkpoint plane / slot: [5][2 TSX][64]
    __shared__ float4 sh0[NW][64], sh1[NW][64], sh2[NW][64];
    // launch order: tile_order holds the tiles of XCD group 0 heavy-first, then group 1's, ... (fgs_bin.hip tile_group);
    // blocks are dealt round-robin over the XCDs, so XCD g walks group g -- one image (or band of tile rows) per L2
    const uint32_t slot_in_order = order_groups > 1u ? fgs_xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const TileCtx c = tile_ctx_of(tile_order[slot_in_order], tiles, tiles_x, ranges, 8u * TSX);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t part = WIDE ? wave >> 1 : wave, half = WIDE ? wave & 1u : 0u;
    const uint32_t X0 = c.X0 + 16u * half;                // this wave's 16 x 16 half of the tile
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    const uint32_t nseg = (c.end - c.start + seg_len - 1) / seg_len;
    const uint32_t spp = (nseg + NP - 1) / NP;  // segments per part
    const uint32_t first_seg = part * spp;
    const bool active = first_seg < nseg;
    const uint32_t pstart = c.start + first_seg * seg_len;
    const uint32_t pend = active ? min(c.end, pstart + spp * seg_len) : pstart;
    // this lane's cells of a checkpoint slot: sub-tile (row, 2 half + col) of the tile's 2 x TSX grid
    float *slot0 = seg_ckpt + (size_t)seg_off[c.tile] * SLOT + 2u * half * 64u + lane;
    auto cell = [](int s) { return ((s >> 1) * TSX + (s & 1)) * 64; };
    float T[4], Cr[4], Cg[4], Cb[4], Dm[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { T[s] = 1.0f; Cr[s] = 0; Cg[s] = 0; Cb[s] = 0; Dm[s] = 0; }
    const uint32_t shx = lx, shy = 16u + ly;  // this lane's column bit in cbits / row bit in flags
    float fx0 = (float)(X0 + lx), fx1 = (float)(X0 + lx + 8u), fy0 = (float)(c.Y0 + ly);
    asm("" : "+v"(fx0), "+v"(fx1), "+v"(fy0));
    for (uint32_t base = pstart; base < pend; base += 64) {
        const uint32_t n = min(64u, pend - base);
        if (base != pstart && ((base - c.start) % seg_len) == 0) {  // LOCAL state in front of this segment
            float *ck = slot0 + (size_t)((base - c.start) / seg_len) * SLOT;
#pragma unroll
            for (int s = 0; s < 4; ++s) ckpt_store<PL>(ck + cell(s), BlendState{Cr[s], Cg[s], Cb[s], T[s], Dm[s]});
        }
        // staging COMPACTS the chunk: only the records that touch this wave's 16 x 16 pixels are parked (in list order),
        // so the list loop below has no skip test -- on 32 x 16 tiles ~15 % of a list's entries touch only the other half
        float4 q0, q1, q2;
        uint32_t flags = 0, cbits = 0;
        if (lane < n) {
            const uint32_t gid = dup_ids[base + lane];
            const float4 *r = reinterpret_cast<const float4 *>(rec + (size_t)gid * FGS_REC_FLOATS);
            q0 = r[0]; q1 = r[1]; q2 = r[2];
            stage_decode_w<2>(X0, c.Y0, __float_as_uint(q2.z), __float_as_uint(q2.w), q1.y, flags, cbits);
        }
        const unsigned long long tmask = __ballot((flags & 15u) != 0u);
        const uint32_t nt = (uint32_t)__popcll(tmask);
        if (flags & 15u) {
            const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(tmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tmask, 0u));
            q0.z *= NEG_HALF_LOG2E; q0.w *= NEG_HALF_LOG2E; q1.x *= NEG_HALF_LOG2E;
            q2.z = __uint_as_float(cbits); q2.w = __uint_as_float(flags);
            // alpha = min(G op, 0.99) = 0.99 clamp01(G op / 0.99): the list loop forms a' = clamp01(G op') with the
            // FREE clamp modifier of v_mul instead of a v_min (4.3 issue cycles on gfx950), the 0.99 rides on the
            // colours / depth (w c = (a' T)(0.99 c)) and on the transmittance update (T -= 0.99 a' T, one v_fmac with a
            // literal): one instruction and ~10 % of the pass's issue cycles less
            q1.y = q1.y / ALPHA_MAX;
            q1.z *= ALPHA_MAX; q1.w *= ALPHA_MAX; q2.x *= ALPHA_MAX; q2.y *= ALPHA_MAX;
            sh0[wave][slot] = q0; sh1[wave][slot] = q1; sh2[wave][slot] = q2;
        }
        __builtin_amdgcn_wave_barrier();  // wave-private LDS: one wave's LDS instructions execute in order
        for (uint32_t j = 0; j < nt; ++j) {
            const uint32_t fl = __builtin_amdgcn_readfirstlane(__float_as_uint(sh2[wave][j].w));  // stage_decode_w flags
            const uint32_t msk = fl & 15u;
            const float4 q0 = sh0[wave][j], q1 = sh1[wave][j], q2 = sh2[wave][j];
            // (Skipping the lane masks for entries whose bbox covers the tile, or the min for opacities <= 0.98, behind
            // wave-uniform branches -- what pays in the backward -- made this loop 10 % SLOWER: 0.68 -> 0.75 ms at config 3;
            // its passes are too short to amortise a branch.)
            const uint32_t cbits = __float_as_uint(q2.z), rbits = __float_as_uint(q2.w);
            // the lane's two column masks once per entry (3.3 sub-tile passes per entry on average: -2.8 %)
            const uint32_t mxc[2] = {(uint32_t)__builtin_amdgcn_sbfe((int)cbits, shx, 1), (uint32_t)__builtin_amdgcn_sbfe((int)cbits, shx + 8u, 1)};
            // ... and its two column offsets (3.3 passes per entry: one subtraction per pass would be more)
            float dxc[2] = {fx0 - q0.x, fx1 - q0.x};
            asm("" : "+v"(dxc[0]), "+v"(dxc[1]));
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                if (!((msk >> (2 * row)) & 3u)) continue;
                const float dy = row ? fy0 + 8.0f - q0.y : fy0 - q0.y;
                const float bdy = q0.w * dy, cyy = (q1.x * dy) * dy;
                const uint32_t my = (uint32_t)__builtin_amdgcn_sbfe((int)rbits, shy + 8u * row, 1);
#pragma unroll
                for (int col = 0; col < 2; ++col) {
                    const int s = 2 * row + col;
                    if (!((msk >> s) & 1u)) continue;  // scalar branch: sub-tile not touched
                    const float dx = dxc[col];
                    const float t = q0.z * dx + bdy;
                    float G = __builtin_amdgcn_exp2f(t * dx + cyy);
                    const uint32_t mk = my & mxc[col];
                    G = __uint_as_float(__float_as_uint(G) & mk);
                    // alpha / 0.99 (opacity >= 0 here); v_med3(x, 0, 1) of a product folds into the product's clamp modifier
                    const float a1 = __builtin_amdgcn_fmed3f(G * q1.y, 0.0f, 1.0f);
                    const float w = a1 * T[s];  // (alpha T) / 0.99
                    Cr[s] += w * q1.z; Cg[s] += w * q1.w; Cb[s] += w * q2.x; Dm[s] += w * q2.y;
                    T[s] = fmaf(w, -ALPHA_MAX, T[s]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- hand the part results over and compose them (the part-0 wave of each half) ----
    const uint32_t last = nseg ? (nseg - 1) / spp : 0u;
    if (NP > 1 && nseg > spp) {  // more than one part
        if (active && part != 0) {
            float *ck = (part == last) ? slot0 : slot0 + (size_t)((part + 1) * spp) * SLOT;
#pragma unroll
            for (int s = 0; s < 4; ++s) ckpt_store<PL>(ck + cell(s), BlendState{Cr[s], Cg[s], Cb[s], T[s], Dm[s]});
        }
        __threadfence_block();
        __syncthreads();
        if (part != 0) return;
        // part 0: its own result is the absolute state in front of part 1
        {
            float *ck = slot0 + (size_t)spp * SLOT;
#pragma unroll
            for (int s = 0; s < 4; ++s) ckpt_store<PL>(ck + cell(s), BlendState{Cr[s], Cg[s], Cb[s], T[s], Dm[s]});
        }
        for (uint32_t pp = 1; pp <= last; ++pp) {
            const float *src = (pp == last) ? slot0 : slot0 + (size_t)((pp + 1) * spp) * SLOT;
            float *dst = slot0 + (size_t)((pp + 1) * spp) * SLOT;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const BlendState acc = compose(BlendState{Cr[s], Cg[s], Cb[s], T[s], Dm[s]}, ckpt_load<PL>(src + cell(s)));
                Cr[s] = acc.Cr; Cg[s] = acc.Cg; Cb[s] = acc.Cb; T[s] = acc.T; Dm[s] = acc.D;
                if (pp != last) ckpt_store<PL>(dst + cell(s), acc);
            }
        }
    } else if (part != 0) {
        return;
    }
    const size_t HW = (size_t)W * H;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint32_t px = X0 + 8u * (s & 1) + lx, py = c.Y0 + 8u * (s >> 1) + ly;
        if (px < W && py < H) {
            const size_t o = (size_t)py * W + px;
            float *ps = pix_state + (size_t)c.b * 6 * HW + o;
            const float Tf = T[s];
            nt_store(ps, Cr[s]); nt_store(ps + HW, Cg[s]); nt_store(ps + 2 * HW, Cb[s]); nt_store(ps + 3 * HW, 1.0f - Tf);
            nt_store(ps + 4 * HW, Dm[s]);  // (plane 5, Phi, belongs to the phase path: nobody reads it on this one)
            float *img = out_rgb + (size_t)c.b * 3 * HW + o;
            nt_store(img, fminf(fmaxf(Cr[s] + Tf * bg0, 0.0f), 1.0f));
            nt_store(img + HW, fminf(fmaxf(Cg[s] + Tf * bg1, 0.0f), 1.0f));
            nt_store(img + 2 * HW, fminf(fmaxf(Cb[s] + Tf * bg2, 0.0f), 1.0f));
            nt_store(out_depth + (size_t)c.b * HW + o, Dm[s]);
        }
    }
}

// Backward, front-to-back.  For pixel p and Gaussian i:
//   dL/dalpha_i = T_i q_i - S_i / (1 - alpha_i),  q_i = gI.c_i + gD d_i,
//   S_i = T_fin (gI.bg) + sum_{j>i} w_j q_j = T_fin (gI.bg) + Total - sum_{j<=i} w_j q_j,
// with Total = gI.C_acc + gD.D_acc known from the forward's saved per-pixel state, so the
// sweep runs in the SAME order as the forward and T_i is recomputed exactly (no division-
// based transmittance recovery, no saved per-pair state).
//
// Gradient accumulation is atomic-free and deterministic: each lane sums its (up to four)
// pixels' contributions to the ten per-Gaussian sums, the wave adds them up through LDS in a fixed
// order (wave_sum10_addtid, fgs_wave.h) and ten lanes store ONE 48-byte row at the duplicate's
// emission slot (dup_off[gaussian] + index of this tile inside the Gaussian's tile rectangle).  A
// Gaussian's rows are contiguous and k_project_bwd sums them in a fixed order.
//
// Work unit = (tile, depth segment of FGS_SEG list entries): the unit restarts from the forward's
// per-pixel checkpoint in front of its segment, so units are independent and of bounded length.
// Six waves per SIMD (80 VGPRs): the loop is bound by VALU issue with dependent chains in every pass (exp -> alpha -> w
// -> S -> rcp -> dalpha), and the sixth wave buys 4 % (1.37 -> 1.31 ms at config 3); a seventh needs spills and loses 35 %.
// NSX = sub-tile columns of the tile: 2 (16 x 16 tiles) or 4 (32 x 16 tiles: eight sub-tiles per lane, ~0.6x as many list
// entries, reductions and gradient rows; 5 waves per SIMD instead of 6 -- the loop is issue-bound, not latency-bound).
#ifndef FGS_BWD_WIDE_WAVES
#define FGS_BWD_WIDE_WAVES 5  /* waves per SIMD of k_composite_bwd<4>; re-measured in round 3 under the clause scheduler: see DESIGN_LOG.md 10.3 */
#endif
#ifndef FGS_BWD_DYN_LDS
#define FGS_BWD_DYN_LDS 0  /* experiment builds: unused dynamic LDS per block, to cap the waves per SIMD below what the registers allow */
#endif
template <int NSX>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NSX == 2 ? 6 : FGS_BWD_WIDE_WAVES, NSX == 2 ? 6 : FGS_BWD_WIDE_WAVES))) void k_composite_bwd(
    uint32_t tiles, uint32_t tiles_x, uint32_t W, uint32_t H, float bg0, float bg1, float bg2, uint32_t dcap,
    const uint32_t *__restrict__ counters, const uint32_t *__restrict__ seg_off,
    const uint32_t *__restrict__ seg_tile, const float *__restrict__ seg_ckpt, const uint32_t *__restrict__ ranges,
    const uint32_t *__restrict__ dup_ids, const float *__restrict__ rec, const uint32_t *__restrict__ dup_off,
    const float *__restrict__ pix_state, const float *__restrict__ g_rgb, const float *__restrict__ g_depth,
    float *__restrict__ grad_rows, float t_eps) {
    __shared__ float4 sh0[CH], sh1[CH], sh2[CH];
    __shared__ uint32_t she[CH];
    __shared__ __attribute__((aligned(16))) float red[10 * FGS_RED_PITCH];  // per-lane partial sums of one list entry
    // work unit = (tile, depth segment): blockIdx.x indexes the unit list built by k_tile_order; the grid is
    // sized from the capacity, surplus blocks leave at once
    constexpr int NS = 2 * NSX, PL = NS * 64, SLOT = 5 * PL;  // sub-tiles per tile, floats per checkpoint plane / slot
    const uint32_t num_units = counters[2];
    if (blockIdx.x >= num_units) return;
    // units are listed tile by tile: an XCD gets a contiguous run of them, so neighbouring tiles -- which share
    // Gaussians, and write adjacent 40-byte gradient rows -- meet in one L2
    const uint32_t unit = fgs_xcd_remap(blockIdx.x, num_units);
    // the split the forward ran with, as the forward recorded it (k_tile_order): segment length, and the number of
    // list parts of the depth-split forward (> 1: checkpoints inside a later part are part-local)
    const uint32_t seg_len = counters[4];
    const uint32_t fwd_parts = (int32_t)counters[5] > 1 ? counters[5] : 0u;
    const uint32_t unit_tile = seg_tile[unit];
    const uint32_t seg = unit - seg_off[unit_tile];
    TileCtx c = tile_ctx_of(unit_tile, tiles, tiles_x, ranges, 8u * NSX);
    uint32_t rebase_seg = seg;  // first segment of this segment's list part if its checkpoint is part-local
    if (fwd_parts > 1) {
        const uint32_t nseg = (c.end - c.start + seg_len - 1) / seg_len;
        const uint32_t spp = (nseg + fwd_parts - 1) / fwd_parts, first = seg / spp * spp;
        if (first != 0) rebase_seg = first;
    }
    c.start += seg * seg_len;
    c.end = min(c.end, c.start + seg_len);
    const uint32_t lane = threadIdx.x;
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    const size_t HW = (size_t)W * H;
    if (t_eps > 0.0f) {
        // FgsDims.saturation_skip: the forward stopped walking this tile's list after `live` segments (count
        // parked in the checkpoint slot of segment 0); the entries of a dead segment get all-zero gradient rows
        const uint32_t live = reinterpret_cast<const uint32_t *>(seg_ckpt + (size_t)seg_off[unit_tile] * SLOT)[0];
        if (seg >= live) {
            for (uint32_t i = c.start + lane; i < c.end; i += 64) {
                const uint32_t gid = dup_ids[i];
                const float4 q2 = reinterpret_cast<const float4 *>(rec + (size_t)gid * FGS_REC_FLOATS)[2];
                const uint32_t bbx = __float_as_uint(q2.z), bby = __float_as_uint(q2.w);
                const uint32_t tx0 = (bbx & 0xFFFFu) / (8u * NSX), tx1 = ((bbx >> 16) - 1) / (8u * NSX);
                const uint32_t ty0 = (bby & 0xFFFFu) / FGS_TILE;
                const uint32_t e = dup_off[gid] + (c.ty - ty0) * (tx1 - tx0 + 1) + (c.tx - tx0);
                if (e < dcap) {
                    float2 *row = reinterpret_cast<float2 *>(grad_rows + (size_t)e * FGS_BLEND_ROW_FLOATS);
#pragma unroll
                    for (int q = 0; q < FGS_BLEND_ROW_FLOATS / 2; ++q) row[q] = make_float2(0.0f, 0.0f);
                }
            }
            return;
        }
    }
    // T: running transmittance (w = alpha T, T -= w: one instruction less per pixel than T = 1 - A, A += w)
    // S: T_fin (gI.bg) + sum over not-yet-visited w q
    float gr[NS], gg[NS], gb[NS], gd[NS], S[NS], T[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t px = c.X0 + 8u * (s % NSX) + lx, py = c.Y0 + 8u * (s / NSX) + ly;
        gr[s] = gg[s] = gb[s] = gd[s] = S[s] = 0.0f;
        T[s] = 1.0f;
        if (px < W && py < H) {
            const size_t o = (size_t)py * W + px;
            const float *ps = pix_state + (size_t)c.b * 6 * HW + o;
            const float Cr = ps[0], Cg = ps[HW], Cb = ps[2 * HW], Af = ps[3 * HW];
            const float Tf = 1.0f - Af;
            const float pr = Cr + Tf * bg0, pg = Cg + Tf * bg1, pb = Cb + Tf * bg2;
            const float *gi = g_rgb + (size_t)c.b * 3 * HW + o;
            gr[s] = (pr >= 0.0f && pr <= 1.0f) ? gi[0] : 0.0f;  // clamp backward, closed interval
            gg[s] = (pg >= 0.0f && pg <= 1.0f) ? gi[HW] : 0.0f;
            gb[s] = (pb >= 0.0f && pb <= 1.0f) ? gi[2 * HW] : 0.0f;
            gd[s] = g_depth[(size_t)c.b * HW + o];
            S[s] = Tf * (gr[s] * bg0 + gg[s] * bg1 + gb[s] * bg2) + (gr[s] * Cr + gg[s] * Cg + gb[s] * Cb) +
                   gd[s] * ps[4 * HW];
        }
        if (seg) {
            // restart from the forward's state in front of this segment: T, and S minus the part of Total that
            // belongs to the list entries before it.  After a depth-split forward (k_blend_fwd_parts, fwd_parts
            // waves per tile) the checkpoint of a segment inside part p > 0 is local to that part and is re-based
            // with the absolute state kept in the slot of the part's first segment.
            BlendState st = ckpt_load<PL>(seg_ckpt + (size_t)unit * SLOT + s * 64 + lane);
            if (rebase_seg != seg) {
                const size_t slot = (size_t)unit - seg + rebase_seg;
                st = compose(ckpt_load<PL>(seg_ckpt + slot * SLOT + s * 64 + lane), st);
            }
            T[s] = st.T;
            S[s] -= gr[s] * st.Cr + gg[s] * st.Cg + gb[s] * st.Cb + gd[s] * st.D;
        }
    }
    const uint32_t shx = lx, shy = 16u + ly;  // this lane's column bit in the staged column bits / row bit in the flags
    uint32_t alive = (1u << NS) - 1u;  // sub-tiles still composited (FgsDims.saturation_skip; 16 x 16 tiles only)
    if (t_eps > 0.0f) {
        alive = 0u;
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (__ballot(T[s] >= t_eps) != 0ull) alive |= 1u << s;  // T = 1 - A of the checkpoint, as in the forward
    }
    float fx0 = (float)(c.X0 + lx), fy0 = (float)(c.Y0 + ly);
    asm("" : "+v"(fx0), "+v"(fy0));  // hoisted for good: no v_cvt in the list loop
    for (uint32_t base = c.start; base < c.end; base += CH) {
        const uint32_t n = min((uint32_t)CH, c.end - base);
        if (lane < n) {
            const uint32_t gid = dup_ids[base + lane];
            const float4 *r = reinterpret_cast<const float4 *>(rec + (size_t)gid * FGS_REC_FLOATS);
            float4 q2 = r[2];
            const uint32_t bbx = __float_as_uint(q2.z), bby = __float_as_uint(q2.w);
            const uint32_t bx0 = bbx & 0xFFFFu, bx1 = bbx >> 16, by0 = bby & 0xFFFFu;
            const uint32_t tx0 = bx0 / (8u * NSX), tx1 = (bx1 - 1) / (8u * NSX), ty0 = by0 / FGS_TILE;
            she[lane] = dup_off[gid] + (c.ty - ty0) * (tx1 - tx0 + 1) + (c.tx - tx0);
            float4 q0 = r[0], q1 = r[1];
            // m = a dx^2 + (b+c) dx dy + d dy^2 >= 0 everywhere (so G <= 1): positive definite, with a margin
            // that covers the fp32 rounding of m
            const bool conic_ok = q0.z > 0.0f && q1.x > 0.0f && 3.996f * q0.z * q1.x > q0.w * q0.w;
            uint32_t flags, cbits;
            stage_decode_w<NSX>(c.X0, c.Y0, bbx, bby, q1.y, flags, cbits, conic_ok);
            q2.z = __uint_as_float(cbits); q2.w = __uint_as_float(flags);
            q0.z *= NEG_HALF_LOG2E; q0.w *= NEG_HALF_LOG2E; q1.x *= NEG_HALF_LOG2E;  // conic in exp2 units
            // alpha = 0.99 a', a' = clamp01(G op / 0.99) (free clamp modifier instead of a v_min, as in the forward); the
            // list loop works with w' = a' T = w / 0.99 and colours c' = 0.99 c, so that w q = w' q'; its sums come out
            // as 0.99 x (moments, sum dG) and 1 / 0.99 x (colour, depth) and k_row_sum puts the factors back
            q1.y = q1.y / ALPHA_MAX;
            q1.z *= ALPHA_MAX; q1.w *= ALPHA_MAX; q2.x *= ALPHA_MAX; q2.y *= ALPHA_MAX;
            sh0[lane] = q0; sh1[lane] = q1; sh2[lane] = q2;
        }
        __syncthreads();
        for (uint32_t j = 0; j < n; ++j) {
            const float4 q0 = sh0[j], q1 = sh1[j], q2 = sh2[j];
            const uint32_t fl = __builtin_amdgcn_readfirstlane(__float_as_uint(q2.w));  // stage_decode_w flags
            const uint32_t msk = fl & alive;
            const uint32_t cbits = __float_as_uint(q2.z), rbits = __float_as_uint(q2.w);  // column bits; row bits at 16+
            const float ca = q0.z, cbc = q0.w, cd = q1.x, op = q1.y;  // conic pre-multiplied by K = -log2(e)/2
            // terms shared by the sub-tiles of a column / row, formed once per list entry
            const float dxa = fx0 - q0.x, dya = fy0 - q0.y, dyb = dya + 8.0f;
            float bdya = cbc * dya, bdyb = cbc * dyb, cyya = (cd * dya) * dya, cyyb = (cd * dyb) * dyb;
            asm("" : "+v"(bdya), "+v"(bdyb), "+v"(cyya), "+v"(cyyb));  // keep the row terms: do not recompute them per sub-tile
            // bbox membership: the lane's column / row bits of the staged pixel bits become all-ones / zero masks
            // (v_bfe_i32) and zero G with a bit-and -- no per-pixel compare / select (issue costs: DESIGN.md).
            // per-lane partial sums over this lane's (up to four) pixels: moments of dL/dG about the Gaussian's mean, sum dG
            // {dx, dy, dx^2, dx dy, dy^2, 1}; the ln2 * opacity and K factors of the chain through m' = K m are
            // applied once per Gaussian in k_project_bwd.  (Raw moments about the tile origin -- one FMA each with
            // per-lane constants -- were 4 % faster but lose the second moments of sub-pixel Gaussians to
            // cancellation in fp32: up to 4e-2 on dL/dscale in the randomized sweeps.)
            float v_mx = 0, v_my = 0, v_ca = 0, v_cbc = 0, v_cd = 0, v_op = 0, v_r = 0, v_g = 0, v_b = 0, v_d = 0;
            {   // ONE code path for every kind of entry; what differs is handled by a wave-uniform branch around two
                // ops: `clamp` (flag bit 4 clear: opacity > 0.98 or a doubtful conic): the clamp-gradient select.
                // (Four specialised instantiations of this loop body made the compiler carry T and S through eight
                // v_mov per list entry and were 5 % slower.)
                uint32_t cflag = fl & 256u;  // clear: `clamp`.  Kept as a scalar and tested where it is used: as a bool the
                asm("" : "+s"(cflag));       // compiler materialised it through a v_cndmask / v_cmp pair per list entry
                // the lane masks of every entry, also of those whose bbox covers the tile (all bits set): skipping the four
                // v_bfe behind a branch needed four v_mov of the default and was 0.9 % slower
                uint32_t mxs[NSX];
                float dxs[NSX];
#pragma unroll
                for (int col = 0; col < NSX; ++col) {
                    mxs[col] = (uint32_t)__builtin_amdgcn_sbfe((int)cbits, shx + 8u * col, 1);
                    dxs[col] = dxa + 8.0f * col;
                }
                const uint32_t my0 = (uint32_t)__builtin_amdgcn_sbfe((int)rbits, shy, 1), my1 = (uint32_t)__builtin_amdgcn_sbfe((int)rbits, shy + 8u, 1);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (!((msk >> s) & 1u)) continue;  // scalar branch: sub-tile not touched
                    // G is zeroed outside the bbox: alpha, w and every gradient term below then vanish by themselves
                    const int col = s % NSX, row = s / NSX;
                    const uint32_t mk = mxs[col] & (row ? my1 : my0);
                    const float dx = dxs[col], dy = row ? dyb : dya;
                    const float t = ca * dx + (row ? bdyb : bdya);
                    const float Gu = __builtin_amdgcn_exp2f(t * dx + (row ? cyyb : cyya));
                    const float G = __uint_as_float(__float_as_uint(Gu) & mk);
                    const float a1 = __builtin_amdgcn_fmed3f(G * op, 0.0f, 1.0f);  // alpha / 0.99: v_mul ... clamp
                    const float w = a1 * T[s];  // w / 0.99
                    const float q = gr[s] * q1.z + gg[s] * q1.w + gb[s] * q2.x + gd[s] * q2.y;  // 0.99 q
                    S[s] -= w * q;
                    // 0.99 dL/dalpha: 0.99 / (1 - alpha) = 1 / (1 / 0.99 - a')
                    float dalpha = T[s] * q - S[s] * __builtin_amdgcn_rcpf(INV_ALPHA_MAX - a1);
                    T[s] = fmaf(w, -ALPHA_MAX, T[s]);
                    uint32_t cf = cflag;
                    asm("" : "+s"(cf));  // an opaque scalar per use: one s_cmp + branch, nothing on the vector side
                    if (!cf) dalpha = select_lt(a1, 1.0f, dalpha);  // the clamp binds where G op / 0.99 reaches 1
                    const float dG = dalpha * G;
                    v_op += dG;
                    const float dmx = dG * dx, dmy = dG * dy;
                    v_mx += dmx; v_my += dmy;
                    v_ca += dmx * dx; v_cbc += dmx * dy; v_cd += dmy * dy;
                    v_r += w * gr[s]; v_g += w * gg[s]; v_b += w * gb[s]; v_d += w * gd[s];
                }
            }
            // ---- reduce the ten sums over the 64 lanes (wave_sum10_addtid, fgs_wave.h) and store them straight
            // into this duplicate's gradient row: no atomics, fixed order, bitwise reproducible ----
            {
                const float vals[10] = {v_mx, v_my, v_ca, v_cbc, v_cd, v_op, v_r, v_g, v_b, v_d};
                const float tot = wave_sum10_addtid(red, vals, lane);
                const uint32_t kk = lane >> 2, e = she[j];
                // (16-float rows: all sixteen quads' last lanes store, lanes >= 40 a zero -- one whole 64-byte line)
                if ((lane & 3u) == 3u && lane < 4u * FGS_BLEND_ROW_FLOATS && e < dcap) grad_rows[(size_t)e * FGS_BLEND_ROW_FLOATS + kk] = tot;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Phase-blending path (BASELINE config 4).  alpha_i depends on the running weighted-mean phase
// Phi_{i-1} (DR:629-667): a true recurrence per pixel, a serial cos / divide chain -- latency-bound, so the unit of work is
// small: ONE WAVE PER 8 x 8 SUB-TILE (four per 16 x 16 tile, in one block so that they share the tile's list through the
// CU's caches), and -- round 4 -- the four waves no longer share anything else:
//   * every wave scans the tile's list 64 entries at a time, tests the bboxes against ITS sub-tile and parks only the
//     records that touch it, COMPACTED and in list order, in wave-private LDS (ballot + mbcnt).  No block barrier in the list
//     walk (round 3: one per 64 entries, the four waves of a tile waiting for the slowest), no scalar mask walk over
//     untouched entries;
//   * checkpoints (A, Phi) for the backward are taken every FGS_PHASE_CKPT TOUCHED entries of a scan block (round 3: every 8
//     list positions that held at least one touched entry -- at ~40 % touched entries that was 7.9 checkpoints per 64
//     entries for 3.2 entries each; now ceil(25 / 8) ~ 3.6): half the checkpoint traffic and twice the entries per
//     restart.  Slot of group g of the scan block at list offset o: start / 8 + o / 8 + g + tile (g < 8: the capacity
//     of fgs_make_plan is unchanged), plane w = A of sub-tile w, plane 4 + w = Phi.
// bbox membership is a per-lane bit-and with masks from the staged column / row bits (v_bfe_i32), as on the blend path.
#ifndef FGS_PHASE_SCAN
#define FGS_PHASE_SCAN 64  /* list entries per scan block (a multiple of FGS_PHASE_CKPT, <= 64) */
#endif
static_assert(FGS_PHASE_SCAN <= 64 && FGS_PHASE_SCAN % FGS_PHASE_CKPT == 0, "scan block");
#ifndef FGS_PHASE_WAVE_BLOCKS
#define FGS_PHASE_WAVE_BLOCKS 0  /* 1 (experiment): every sub-tile wave is a workgroup of its own -- see phase_wave_block */
#endif
// FGS_PHASE_WAVE_BLOCKS: grid = 4 x (tiles rounded up to 8) single-wave blocks.  Blocks are dealt round-robin over the 8 XCDs;
// block L serves launch slot (L / 32) * 8 + L % 8 as sub-tile wave (L / 8) % 4, so the four waves of a tile land on ONE XCD
// (they share the tile's list through its L2) and consecutive slots of the heavy-first order spread over all eight.
__device__ __forceinline__ bool phase_wave_block(uint32_t total_slots, uint32_t &slot, uint32_t &wave) {
    const uint32_t L = blockIdx.x;
    slot = (L >> 5) * 8u + (L & 7u);
    wave = (L >> 3) & 3u;
    return slot < total_slots;
}
struct PhaseRec {                 // one compacted list entry in wave-private LDS
    float4 a[FGS_PHASE_SCAN];     // u, v, conic a, conic b + c
    float4 b[FGS_PHASE_SCAN];     // conic d, opacity, colour r, g
    float4 c[FGS_PHASE_SCAN];     // colour b, depth, phase, pixel bits (bits 0-7: columns sx + i inside the bbox, 8-15: rows sy + i)
};

// scan step shared by the two kernels: lane j < n looks at list entry base + j; returns the ballot of the entries that touch
// the sub-tile at (sx, sy) and parks those, compacted, in `st`.  K = factor on the conic (forward: exp2 units)
template <bool FWD>
__device__ __forceinline__ unsigned long long phase_scan(PhaseRec &st, uint32_t *rows, uint32_t gid, bool have, uint32_t sx,
                                                         uint32_t sy, const TileCtx &c,
                                                         const float *__restrict__ rec, const float *__restrict__ phase,
                                                         const uint32_t *__restrict__ dup_off) {
    float4 q0, q1, q2;
    uint32_t bits = 0, e = 0;
    if (have) {
        const float4 *r = reinterpret_cast<const float4 *>(rec + (size_t)gid * FGS_REC_FLOATS);
        q0 = r[0]; q1 = r[1]; q2 = r[2];
        const uint32_t bbx = __float_as_uint(q2.z), bby = __float_as_uint(q2.w);
        const int x0 = (int)(bbx & 0xFFFFu), x1 = (int)(bbx >> 16), y0 = (int)(bby & 0xFFFFu), y1 = (int)(bby >> 16);
        const int lx0 = max(x0 - (int)sx, 0), lx1 = min(x1 - (int)sx, 8), ly0 = max(y0 - (int)sy, 0), ly1 = min(y1 - (int)sy, 8);
        const uint32_t cm = lx1 > lx0 ? ((1u << (lx1 - lx0)) - 1u) << lx0 : 0u;
        const uint32_t rm = ly1 > ly0 ? ((1u << (ly1 - ly0)) - 1u) << ly0 : 0u;
        bits = (cm && rm) ? (cm | (rm << 8)) : 0u;
        if (!FWD) {  // this duplicate's emission slot: its four gradient rows (one per sub-tile wave) are 4 e .. 4 e + 3 (k_project_bwd)
            const uint32_t tx0 = (uint32_t)x0 / FGS_TILE, tx1 = (uint32_t)(x1 - 1) / FGS_TILE, ty0 = (uint32_t)y0 / FGS_TILE;
            e = dup_off[gid] + (c.ty - ty0) * (tx1 - tx0 + 1) + (c.tx - tx0);
        }
    }
    const unsigned long long touched = __ballot(bits != 0u);
    if (bits) {
        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(touched >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)touched, 0u));
        // conic in exp2 units, forward AND backward (round 5: the backward's re-run and reverse sweep then evaluate G with the forward's
        // own operation order -- and one multiply less per evaluation)
        q0.z *= NEG_HALF_LOG2E; q0.w *= NEG_HALF_LOG2E; q1.x *= NEG_HALF_LOG2E;
        st.a[slot] = q0; st.b[slot] = q1;
        st.c[slot] = make_float4(q2.x, q2.y, phase[gid], __uint_as_float(bits));
        if (!FWD) rows[slot] = e;
    }
    __builtin_amdgcn_wave_barrier();  // wave-private LDS: one wave's LDS instructions execute in order
    return touched;
}

__global__ __launch_bounds__(FGS_PHASE_WAVE_BLOCKS ? 64 : 256) void k_phase_fwd(
    uint32_t tiles, uint32_t tiles_x, uint32_t W, uint32_t H, float bg0, float bg1, float bg2, float amp,
    const uint32_t *__restrict__ tile_order, const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ dup_ids,
    const float *__restrict__ rec, const float *__restrict__ phase, float *__restrict__ pix_state,
    float *__restrict__ phase_ckpt, float *__restrict__ out_rgb, float *__restrict__ out_depth, uint32_t total_slots) {
    constexpr int PCK = FGS_PHASE_CKPT;
    __shared__ PhaseRec st4[FGS_PHASE_WAVE_BLOCKS ? 1 : 4];
#if FGS_PHASE_WAVE_BLOCKS
    uint32_t pslot, wave;
    if (!phase_wave_block(total_slots, pslot, wave)) return;
    const TileCtx c = tile_ctx_of(tile_order[pslot], tiles, tiles_x, ranges);
    const uint32_t lane = threadIdx.x & 63u;
    PhaseRec &st = st4[0];
#else
    const TileCtx c = tile_ctx(tiles, tiles_x, tile_order, ranges);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    PhaseRec &st = st4[wave];
#endif
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    const uint32_t sx = c.X0 + 8u * (wave & 1u), sy = c.Y0 + 8u * (wave >> 1);
    float fpx = (float)(sx + lx), fpy = (float)(sy + ly);
    asm("" : "+v"(fpx), "+v"(fpy));  // hoisted for good: no v_cvt in the list loop
    float A = 0.0f, Ph = 0.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f, Dm = 0.0f;
    const float base_amp = 1.0f - amp;
    constexpr uint32_t SCAN = FGS_PHASE_SCAN;
    uint32_t gid_next = (lane < SCAN && c.start + lane < c.end) ? dup_ids[c.start + lane] : 0u;
    for (uint32_t base = c.start; base < c.end; base += SCAN) {
        const uint32_t gid = gid_next;
        const bool have = lane < SCAN && base + lane < c.end;
        if (lane < SCAN && base + SCAN + lane < c.end) gid_next = dup_ids[base + SCAN + lane];  // the next block's ids travel under this one's work
        const unsigned long long touched = phase_scan<true>(st, nullptr, gid, have, sx, sy, c, rec, phase, nullptr);
        const uint32_t nt = (uint32_t)__popcll(touched);
        for (uint32_t j0 = 0; j0 < nt; j0 += PCK) {
            const size_t slot = (size_t)(c.start / PCK) + (base - c.start + j0) / PCK + c.tile;
            float *ck = phase_ckpt + slot * 512 + lane;
            ck[wave * 64] = A; ck[(4 + wave) * 64] = Ph;
            const uint32_t m = min((uint32_t)PCK, nt - j0);
#pragma unroll
            for (int k = 0; k < PCK; ++k) {
                if (k >= (int)m) break;  // wave-uniform
                const float4 q0 = st.a[j0 + k], q1 = st.b[j0 + k], q2 = st.c[j0 + k];
                const uint32_t bits = __float_as_uint(q2.w);
                const uint32_t mk = (uint32_t)__builtin_amdgcn_sbfe((int)bits, lx, 1) & (uint32_t)__builtin_amdgcn_sbfe((int)bits, 8u + ly, 1);
                const float dx = fpx - q0.x, dy = fpy - q0.y;
                const float mm = (q0.z * dx) * dx + (q0.w * dx) * dy + (q1.x * dy) * dy;  // conic in exp2 units
                float alpha = __builtin_amdgcn_exp2f(mm) * q1.y;
                float pd = fabsf(q2.z - Ph);
                pd = fminf(pd, 1.0f - pd);
                alpha *= base_amp + amp * phase_cos(pd * PHASE_KAPPA);
                alpha = __builtin_amdgcn_fmed3f(alpha, 0.0f, ALPHA_MAX);
                alpha = __uint_as_float(__float_as_uint(alpha) & mk);   // zero outside the bbox: w = pc = 0, nothing moves
                const float w = alpha * (1.0f - A);
                Cr += w * q1.z; Cg += w * q1.w; Cb += w * q2.x; Dm += w * q2.y;
                A += w;
                const float pc = w / fmaxf(A, 1e-6f);
                Ph = Ph * (1.0f - pc) + q2.z * pc;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the next scan block overwrites the parked records
    }
    const uint32_t px = sx + lx, py = sy + ly;
    if (px < W && py < H) {
        const size_t HW = (size_t)W * H, o = (size_t)py * W + px;
        float *ps = pix_state + (size_t)c.b * 6 * HW + o;
        ps[0] = Cr; ps[HW] = Cg; ps[2 * HW] = Cb; ps[3 * HW] = A; ps[4 * HW] = Dm; ps[5 * HW] = Ph;
        const float T = 1.0f - A;
        float *img = out_rgb + (size_t)c.b * 3 * HW + o;
        img[0] = fminf(fmaxf(Cr + T * bg0, 0.0f), 1.0f);
        img[HW] = fminf(fmaxf(Cg + T * bg1, 0.0f), 1.0f);
        img[2 * HW] = fminf(fmaxf(Cb + T * bg2, 0.0f), 1.0f);
        out_depth[(size_t)c.b * HW + o] = Dm;
    }
}

// (Round 5 measured and removed three experiment switches of this kernel -- parking G / the interference factor in the re-run, checkpoint
// groups of sixteen, the re-run interleaved with the sweep: profiles/r05_ab_phase_variants.txt; the code of the first two is in commit c134f2a.)
// Backward of the phase recurrence (the reference cannot backprop this path at all, §0.6 -- the contract is the
// exact adjoint of the forward recurrence DR:629-667).  The recurrence cannot be inverted back-to-front, so the wave walks
// its sub-tile's touched entries in REVERSE checkpoint groups: re-runs the forward inside a group from the group's (A, Phi)
// checkpoint, keeping (A_{i-1}, Phi_{i-1}) of its entries in registers (both loops over a group are fully unrolled), then
// sweeps the group back-to-front with the per-pixel adjoints Abar (init -gI.bg) and Phibar.  Every touched (entry, sub-tile)
// gets its OWN gradient row (row 4 e + w; k_project_bwd repeats the integer test and never reads the rows of untouched
// sub-tiles): no cross-wave reduction.
__global__ __launch_bounds__(FGS_PHASE_WAVE_BLOCKS ? 64 : 256) void k_phase_bwd(
    uint32_t tiles, uint32_t tiles_x, uint32_t W, uint32_t H, float bg0, float bg1, float bg2, float amp,
    uint32_t dcap, const uint32_t *__restrict__ tile_order, const uint32_t *__restrict__ ranges,
    const uint32_t *__restrict__ dup_ids, const float *__restrict__ rec, const float *__restrict__ phase,
    const uint32_t *__restrict__ dup_off, const float *__restrict__ pix_state,
    const float *__restrict__ phase_ckpt, const float *__restrict__ g_rgb, const float *__restrict__ g_depth,
    float *__restrict__ grad_rows, uint32_t total_slots) {
    constexpr int PCK = FGS_PHASE_CKPT;
    constexpr int NWB = FGS_PHASE_WAVE_BLOCKS ? 1 : 4;
    __shared__ PhaseRec st4[NWB];
    __shared__ uint32_t rows4[NWB][FGS_PHASE_SCAN];
    __shared__ __attribute__((aligned(16))) float red4[NWB][11 * FGS_RED_PITCH];  // reduction scratch, one per wave
#if FGS_PHASE_WAVE_BLOCKS
    uint32_t pslot, wave;
    if (!phase_wave_block(total_slots, pslot, wave)) return;
    const TileCtx c = tile_ctx_of(tile_order[pslot], tiles, tiles_x, ranges);
    const uint32_t lane = threadIdx.x & 63u;
    PhaseRec &st = st4[0];
    uint32_t *rows = rows4[0];
    float *red = red4[0];
#else
    const TileCtx c = tile_ctx(tiles, tiles_x, tile_order, ranges);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    PhaseRec &st = st4[wave];
    uint32_t *rows = rows4[wave];
    float *red = red4[wave];
#endif
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    const size_t HW = (size_t)W * H;
    const uint32_t sx = c.X0 + 8u * (wave & 1u), sy = c.Y0 + 8u * (wave >> 1);
    const uint32_t px = sx + lx, py = sy + ly;
    float fpx = (float)px, fpy = (float)py;
    asm("" : "+v"(fpx), "+v"(fpy));
    float gr = 0.0f, gg = 0.0f, gb = 0.0f, gd = 0.0f, Abar = 0.0f, Pbar = 0.0f;
    if (px < W && py < H) {
        const size_t o = (size_t)py * W + px;
        const float *ps = pix_state + (size_t)c.b * 6 * HW + o;
        const float Tf = 1.0f - ps[3 * HW];
        const float pr = ps[0] + Tf * bg0, pg = ps[HW] + Tf * bg1, pb = ps[2 * HW] + Tf * bg2;
        const float *gi = g_rgb + (size_t)c.b * 3 * HW + o;
        gr = (pr >= 0.0f && pr <= 1.0f) ? gi[0] : 0.0f;  // clamp backward, closed interval
        gg = (pg >= 0.0f && pg <= 1.0f) ? gi[HW] : 0.0f;
        gb = (pb >= 0.0f && pb <= 1.0f) ? gi[2 * HW] : 0.0f;
        gd = g_depth[(size_t)c.b * HW + o];
        Abar = -(gr * bg0 + gg * bg1 + gb * bg2);  // d/dA of (1 - A) * bg
    }
    const float base_amp = 1.0f - amp, neg_amp_kappa = -(amp * PHASE_KAPPA);
    const uint32_t total = c.end - c.start;
    constexpr uint32_t SCAN = FGS_PHASE_SCAN;
    const uint32_t nblocks = (total + SCAN - 1u) / SCAN;
    uint32_t gid_next = 0u;
    if (nblocks) {
        const uint32_t i = c.start + (nblocks - 1u) * SCAN + lane;
        if (lane < SCAN && i < c.end) gid_next = dup_ids[i];
    }
    for (uint32_t bi = nblocks; bi-- > 0;) {
        const uint32_t base = c.start + bi * SCAN;
        const uint32_t gid = gid_next;
        const bool have = lane < SCAN && base + lane < c.end;
        if (bi && lane < SCAN) gid_next = dup_ids[base - SCAN + lane];  // the block in front of this one: always full
        const unsigned long long touched = phase_scan<false>(st, rows, gid, have, sx, sy, c, rec, phase, dup_off);
        const uint32_t nt = (uint32_t)__popcll(touched);
        const uint32_t ngroups = (nt + PCK - 1) / PCK;
        for (uint32_t g = ngroups; g-- > 0;) {
            const uint32_t j0 = g * PCK;
            const uint32_t m = min((uint32_t)PCK, nt - j0);
            const size_t slot = (size_t)(c.start / PCK) + (bi * SCAN + j0) / PCK + c.tile;
            const float *ck = phase_ckpt + slot * 512 + lane;
            float Af = ck[wave * 64], Pf = ck[(4 + wave) * 64];
            // ---- forward re-run of the group: park (A_{i-1}, Phi_{i-1}) ----
            float sA[PCK], sP[PCK];
#pragma unroll
            for (int k = 0; k < PCK; ++k) {
                sA[k] = 0.0f; sP[k] = 0.0f;
                if (k >= (int)m) continue;  // wave-uniform
                const float4 q0 = st.a[j0 + k], q1 = st.b[j0 + k], q2 = st.c[j0 + k];
                const uint32_t bits = __float_as_uint(q2.w);
                const uint32_t mk = (uint32_t)__builtin_amdgcn_sbfe((int)bits, lx, 1) & (uint32_t)__builtin_amdgcn_sbfe((int)bits, 8u + ly, 1);
                sA[k] = Af; sP[k] = Pf;
                const float dx = fpx - q0.x, dy = fpy - q0.y;
                const float mm = (q0.z * dx) * dx + (q0.w * dx) * dy + (q1.x * dy) * dy;
                float alpha = __builtin_amdgcn_exp2f(mm) * q1.y;  // (conic staged in exp2 units)
                float pd = fabsf(q2.z - Pf);
                pd = fminf(pd, 1.0f - pd);
                alpha *= base_amp + amp * phase_cos(pd * PHASE_KAPPA);
                alpha = __builtin_amdgcn_fmed3f(alpha, 0.0f, ALPHA_MAX);
                alpha = __uint_as_float(__float_as_uint(alpha) & mk);
                const float w = alpha * (1.0f - Af);
                Af += w;
                const float pc = w / fmaxf(Af, 1e-6f);
                Pf = Pf * (1.0f - pc) + q2.z * pc;
            }
            // ---- reverse sweep of the group ----
#pragma unroll
            for (int k = PCK - 1; k >= 0; --k) {
                if (k >= (int)m) continue;  // wave-uniform
                const float4 q0 = st.a[j0 + k], q1 = st.b[j0 + k], q2 = st.c[j0 + k];
                const uint32_t e = rows[j0 + k];
                const uint32_t bits = __float_as_uint(q2.w);
                const uint32_t mk = (uint32_t)__builtin_amdgcn_sbfe((int)bits, lx, 1) & (uint32_t)__builtin_amdgcn_sbfe((int)bits, 8u + ly, 1);
                const float ca = q0.z, cbc = q0.w, cd = q1.x, op = q1.y, ph = q2.z;
                const float Aprev = sA[k], Pprev = sP[k];
                const float dx = fpx - q0.x, dy = fpy - q0.y;
                const float dphi = ph - Pprev;
                const float pd0 = fabsf(dphi);
                const float pd = fminf(pd0, 1.0f - pd0);
                // G zeroed outside the bbox: raw, alpha, w, pc and every gradient term of this pixel then vanish by themselves
                const float mm = (ca * dx) * dx + (cbc * dx) * dy + (cd * dy) * dy;
                const float G = __uint_as_float(__float_as_uint(__builtin_amdgcn_exp2f(mm)) & mk);
                const float inter = base_amp + amp * phase_cos(pd * PHASE_KAPPA);
                // (parking G and `inter` of the re-run instead of recomputing them costs 14 VGPRs = one wave per SIMD and
                // was 7 % slower in round 2: this kernel lives on occupancy)
                const float Gop = G * op, Gint = G * inter;
                const float raw = Gop * inter;
                const float alpha = __builtin_amdgcn_fmed3f(raw, 0.0f, ALPHA_MAX);
                const float T = 1.0f - Aprev;
                const float w = alpha * T;
                const float Ai = Aprev + w;
                const float rA = __builtin_amdgcn_rcpf(fmaxf(Ai, 1e-6f));
                const float pc = w * rA;
                const float Pb = __uint_as_float(__float_as_uint(Pbar) & mk);  // Phibar reaches this entry only inside its bbox
                float v_ph = Pb * pc;
                const float pcbar = Pb * dphi;
                // d pc/d w, direct (1/A_i) plus through A_i (-w/A_i^2), equals A_{i-1}/A_i^2: kept in that cancellation-free
                // form (the two parts cancel to ~0 for a pixel's first contribution)
                // (below the 1e-6 clamp of A_i, pc = w / 1e-6: d pc/d w = 1 / 1e-6 = rA and no dependence on A_{i-1}; selected per lane
                // -- a divergent branch cost a save / restore of the exec mask per entry)
                const float rA2 = rA * rA;
                float dpc_dw, dpc_dA;
                select2_ge(Ai, 1e-6f, Aprev * rA2, w * rA2, rA, dpc_dw, dpc_dA);  // (one compare, both selects adjacent: issue costs)
                const float wbar = Abar + (gr * q1.z + gg * q1.w + gb * q2.x + gd * q2.y) + pcbar * dpc_dw;
                const float v_r = w * gr, v_g = w * gg, v_b = w * gb, v_d = w * gd;
                const float abar = wbar * T;
                Abar = (Abar - pcbar * dpc_dA) - wbar * alpha;  // (outside the bbox: w = alpha = pcbar = 0, Abar unchanged)
                // clamp passes the gradient on the closed interval [0, 0.99]: exactly where clamping changed nothing (alpha == raw;
                // one compare instead of two)
                const float rbar = select_eq(alpha, raw, abar);
                const float v_op = rbar * Gint;
                const float pdbar = (rbar * Gop) * neg_amp_kappa * phase_sin(PHASE_KAPPA * pd);
                // d pd/d dphi = sign(1/2 - |dphi|) sign(dphi), 0 on either tie: the sign of x = (1/2 - |dphi|) dphi, formed by saturating
                // x 2^100 to [-1, 1] (exactly +-1 for |x| >= 2^-100, exactly 0 for x = 0; four compares and four selects before round 5).
                // (pd0 < 1 - pd0 is pd0 < 1/2 for every fp32 pd0 in [0, 1]: 1 - pd0 rounds to >= 1/2 below one half and is exact above.)
                const float t_ph = pdbar * __builtin_amdgcn_fmed3f(((0.5f - pd0) * dphi) * 1.2676506e30f, -1.0f, 1.0f);
                v_ph += t_ph;
                Pbar -= Pb * pc + t_ph;  // = Pb (1 - pc) - pd0bar sg inside the bbox, unchanged outside (Pb = G = 0 there)
                const float dm = rbar * raw;  // -2 dL/dm: the -1/2 rides on the per-Gaussian sums (k_project_bwd)
                const float dmx = dm * dx, dmy = dm * dy;
                // slots 0 / 1: the FIRST MOMENTS of dL/dm; k_project_bwd turns their per-Gaussian sums into those of dL/dm' (m' = K m, the
                // blend path's convention: x 1 / K = -2 ln2, once per Gaussian since round 5) and forms dL/d(u, v) = -K conic_sym (moments)
                // in double (the same chain as the blend backward's rows)
                const float vals[11] = {dmx, dmy, dmx * dx, dmx * dy, dmy * dy, v_op, v_r, v_g, v_b, v_d, v_ph};  // (slots 0-4: moments of -2 dL/dm, rescaled in k_project_bwd)
                const float tot = wave_sum11_addtid(red, vals, lane);
                if ((lane & 3u) == 3u && lane < 44u && e < dcap)
                    grad_rows[((size_t)e * 4 + wave) * FGS_GROW_FLOATS + (lane >> 2)] = tot;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the next scan block overwrites the parked records
    }
}

}  // namespace

// Kernel variants are template instantiations selected by the plan (fgs_make_plan: FgsPlan.fwd_parts / fwd_waves,
// a pure function of the FgsDims; FgsDims.fwd_variant overrides for A/B runs and the variant-agreement tests).
int fgs_launch_composite_fwd(const FgsPlan &p, const float *phase, char *saved, float *out_rgb,
                             float *out_depth, hipStream_t st) {
    const uint32_t grid = (uint32_t)p.d.batch * p.tiles;
    const uint32_t *ranges = reinterpret_cast<const uint32_t *>(saved + p.L.ranges);
    const uint32_t *dup_ids = reinterpret_cast<const uint32_t *>(saved + p.L.dup_ids);
    const uint32_t *tile_order = reinterpret_cast<const uint32_t *>(saved + p.L.tile_order);
    const float *rec = reinterpret_cast<const float *>(saved + p.L.rec);
    float *pix = reinterpret_cast<float *>(saved + p.L.pix_state);
    const uint32_t *seg_off = reinterpret_cast<const uint32_t *>(saved + p.L.seg_off);
    float *seg_ckpt = reinterpret_cast<float *>(saved + p.L.seg_ckpt);
    if (p.d.use_phase) {  // one wave per 8 x 8 sub-tile, four per block
        hipLaunchKernelGGL(k_phase_fwd, dim3(FGS_PHASE_WAVE_BLOCKS ? (grid + 7u) / 8u * 32u : grid), dim3(FGS_PHASE_WAVE_BLOCKS ? 64 : 256), 0, st,
                           (uint32_t)p.tiles, (uint32_t)p.L.tiles_x, (uint32_t)p.d.width,
                           (uint32_t)p.d.height, p.d.background[0], p.d.background[1], p.d.background[2], p.d.phase_amplitude,
                           tile_order, ranges, dup_ids, rec, phase, pix, reinterpret_cast<float *>(saved + p.L.phase_ckpt), out_rgb,
                           out_depth, grid);
        FGS_LAUNCH_CHECK("k_phase_fwd");
        return FGS_OK;
    }
    const float t_eps = p.d.saturation_skip ? FGS_SATURATION_EPS : 0.0f;
    const int fw = p.fwd_waves;
    if (const int np = p.fwd_parts) {
#define FGS_PARTS_LAUNCH(NP, WD)                                                                              \
    hipLaunchKernelGGL((k_blend_fwd_parts<NP, WD>), dim3(grid), dim3(64 * NP * (1 + WD)), 0, st, (uint32_t)p.tiles, \
                       (uint32_t)p.L.tiles_x, (uint32_t)p.d.width, (uint32_t)p.d.height, p.d.background[0],  \
                       p.d.background[1], p.d.background[2], tile_order, ranges, dup_ids, rec, pix, out_rgb,  \
                       out_depth, seg_off, seg_ckpt, (uint32_t)p.L.seg_len, (uint32_t)p.order_groups)
        if (p.tile_w == 32) {
            if (np == 1) FGS_PARTS_LAUNCH(1, 1); else if (np == 2) FGS_PARTS_LAUNCH(2, 1); else if (np == 4) FGS_PARTS_LAUNCH(4, 1);
            else FGS_PARTS_LAUNCH(8, 1);
        } else {
            if (np == 1) FGS_PARTS_LAUNCH(1, 0); else if (np == 2) FGS_PARTS_LAUNCH(2, 0); else if (np == 4) FGS_PARTS_LAUNCH(4, 0);
            else if (np == 8) FGS_PARTS_LAUNCH(8, 0); else FGS_PARTS_LAUNCH(16, 0);
        }
#undef FGS_PARTS_LAUNCH
        FGS_LAUNCH_CHECK("k_blend_fwd_parts");
        return FGS_OK;
    }
#define FGS_FWD_LAUNCH(FW, SK)                                                                                \
    hipLaunchKernelGGL((k_composite_fwd<FW, SK>), dim3(grid), dim3(64 * FW), 0, st, (uint32_t)p.tiles,       \
                       (uint32_t)p.L.tiles_x, (uint32_t)p.d.width, (uint32_t)p.d.height, p.d.background[0],  \
                       p.d.background[1], p.d.background[2], tile_order, ranges, dup_ids, rec, pix, out_rgb,  \
                       out_depth, seg_off, seg_ckpt, t_eps)
    if (t_eps > 0.0f) {  // FgsDims.saturation_skip: separate instantiation, the default path carries no trace of it
        if (fw == 1) FGS_FWD_LAUNCH(1, true); else if (fw == 4) FGS_FWD_LAUNCH(4, true); else FGS_FWD_LAUNCH(2, true);
    } else {
        if (fw == 1) FGS_FWD_LAUNCH(1, false); else if (fw == 4) FGS_FWD_LAUNCH(4, false); else FGS_FWD_LAUNCH(2, false);
    }
#undef FGS_FWD_LAUNCH
    FGS_LAUNCH_CHECK("k_composite_fwd");
    return FGS_OK;
}

int fgs_launch_composite_bwd(const FgsPlan &p, const float *phase, const char *saved, char *scratch,
                             const float *g_rgb, const float *g_depth, float *g_phase, hipStream_t st) {
    (void)g_phase;  // dL/dphase travels in the gradient rows and is written by k_project_bwd
    const uint32_t grid = (uint32_t)p.d.batch * p.tiles;
    if (p.d.use_phase) {
        hipLaunchKernelGGL(k_phase_bwd, dim3(FGS_PHASE_WAVE_BLOCKS ? (grid + 7u) / 8u * 32u : grid), dim3(FGS_PHASE_WAVE_BLOCKS ? 64 : 256), 0, st, (uint32_t)p.tiles,
                           (uint32_t)p.L.tiles_x, (uint32_t)p.d.width, (uint32_t)p.d.height, p.d.background[0],
                           p.d.background[1], p.d.background[2], p.d.phase_amplitude, (uint32_t)p.L.dup_capacity,
                           reinterpret_cast<const uint32_t *>(saved + p.L.tile_order),
                           reinterpret_cast<const uint32_t *>(saved + p.L.ranges),
                           reinterpret_cast<const uint32_t *>(saved + p.L.dup_ids),
                           reinterpret_cast<const float *>(saved + p.L.rec), phase,
                           reinterpret_cast<const uint32_t *>(saved + p.L.dup_off),
                           reinterpret_cast<const float *>(saved + p.L.pix_state),
                           reinterpret_cast<const float *>(saved + p.L.phase_ckpt), g_rgb, g_depth,
                           reinterpret_cast<float *>(scratch + p.s_grows), grid);
        FGS_LAUNCH_CHECK("k_phase_bwd");
        return FGS_OK;
    }
    const uint32_t ugrid = (uint32_t)p.L.seg_capacity;  // surplus blocks exit at once (measured: free)
#define FGS_BWD_LAUNCH(NSXV)                                                                                  \
    hipLaunchKernelGGL(k_composite_bwd<NSXV>, dim3(ugrid), dim3(64), FGS_BWD_DYN_LDS, st, (uint32_t)p.tiles, (uint32_t)p.L.tiles_x, \
                       (uint32_t)p.d.width, (uint32_t)p.d.height, p.d.background[0], p.d.background[1], \
                       p.d.background[2], (uint32_t)p.L.dup_capacity, \
                       reinterpret_cast<const uint32_t *>(saved + p.L.counters), \
                       reinterpret_cast<const uint32_t *>(saved + p.L.seg_off), \
                       reinterpret_cast<const uint32_t *>(saved + p.L.seg_tile), \
                       reinterpret_cast<const float *>(saved + p.L.seg_ckpt), \
                       reinterpret_cast<const uint32_t *>(saved + p.L.ranges), \
                       reinterpret_cast<const uint32_t *>(saved + p.L.dup_ids), \
                       reinterpret_cast<const float *>(saved + p.L.rec), \
                       reinterpret_cast<const uint32_t *>(saved + p.L.dup_off), \
                       reinterpret_cast<const float *>(saved + p.L.pix_state), g_rgb, g_depth, \
                       reinterpret_cast<float *>(scratch + p.s_grows), \
                       p.d.saturation_skip ? FGS_SATURATION_EPS : 0.0f)
    if (p.tile_w == 32) FGS_BWD_LAUNCH(4); else FGS_BWD_LAUNCH(2);
#undef FGS_BWD_LAUNCH
    FGS_LAUNCH_CHECK("k_composite_bwd");
    return FGS_OK;
}
