// BUILD EXPERIMENT (FGS_ASM_ROWS=1; off by default -- measured slower, see make_asm_plan in fgs_asm.hip; parity-tested on the
// GPU in round 3: tests/test_hip_asm.py, tests/test_parity_domain.py green with it).
// Row transforms of the angular-spectrum renderer FUSED with the splat (forward) and with its adjoint (backward); included
// by fgs_asm.hip inside its anonymous namespace.  Replaces, for power-of-two widths 64 ... 512 (config 5: 512),
//     forward   k_asm_splat (writes field[b][p][c][y][x]) + rocFFT 1-D rows (read + write of the same 0.8 GB at 8 images)
//     backward  rocFFT 1-D inverse rows (read + write) + k_asm_splat<true> (reads the field gradient)
// by ONE kernel per direction that keeps a band of 8 rows x 3 channels x W points in LDS: the forward splats a tile row's
// (image, plane, tile) lists into registers, parks one 8-row half at a time in LDS, transforms the 24 lines along x and stores
// the row spectra -- the plane fields themselves never exist in HBM; the backward loads the column-transformed gradient
// spectra, inverse-transforms the rows in LDS, hands every wave its tiles' field gradients in registers and walks the lists
// (reference: the per-plane fft2 / ifft2 of DR:1286-1313 with the splat of DR:1233-1283 in front of it).
// One write instead of write + read + write of the plane data per direction: -1.6 GB of the 8.0 GB config 5 moves per step.
//
// PLANES WITHOUT GAUSSIANS are skipped here and in the column kernels (k_colfft_fwd / k_colfft_bwd), as the reference
// skips them (DR:1302): their blocks leave at once, their spectra are neither written nor read.
//
// Block = (tile row, plane, image), 1024 threads = 16 waves; wave w owns tiles w, w + 16 of the row (whole lists: this
// path is for launches of many short lists, fgs_asm.hip decides).  LDS: x[W][32] float2, point-major, 24 lines per point,
// the line slot rotated by rows_rot(point) so that all three access patterns are at the 2-way minimum of a 64-lane 8-byte
// access: (lx, ly) lanes parking / fetching pixels, the butterflies, and consecutive-frequency lanes against the
// bit-reversed point order of the in-place transform (without the rotation: 8-way and 64-way; searched numerically).
#pragma once

constexpr int ROWS_NT = 1024;     // threads per block
constexpr int ROWS_LINES = 24;    // 8 rows x 3 channels
constexpr int ROWS_PITCH = 32;    // float2 slots per point
constexpr int ROWS_MAX_LOGW = 9;  // W <= 512: 128 KB of LDS, two tiles per wave
// per-wave list staging of the two kernels, in float4 units (it overlays the transform buffer, which is therefore never
// smaller than sixteen of them): four float4 per record + sub-tile masks [+ gradient-row slots + reduction scratch]
constexpr int ROWS_STG_FWD = 4 * ACH + ACH / 4;
constexpr int ROWS_STG_BWD = 4 * ACH + ACH / 2 + (12 * FGS_RED_PITCH + 3) / 4 + 1;
template <int LOGW, int STG>
constexpr int rows_lds_float2() {  // float2 elements of the shared buffer
    return ((1 << LOGW) * ROWS_PITCH > 16 * STG * 2) ? (1 << LOGW) * ROWS_PITCH : 16 * STG * 2;
}

__device__ __forceinline__ int rows_slot(int r, int line) {
    return r * ROWS_PITCH + ((line + 8 * (r & 7) + ((r >> 3) & 7) + 4 * (r >> 6)) & (ROWS_PITCH - 1));
}

// an (image, plane) pair -- or, with `tiles` = lists per image, an image -- with no list entry at all: seg_off is the
// exclusive scan of every list's depth-segment count in key order (key = (b P + p) T + t; k_tile_post writes it on both
// list-building paths, [lists + 1] entries), so a key range without entries is a range without units.  (NOT `ranges`: the
// radix path leaves the ranges of empty lists zeroed.)
__device__ __forceinline__ bool asm_plane_empty(const uint32_t *__restrict__ seg_off, uint32_t bp, uint32_t tiles) {
    return seg_off[(size_t)(bp + 1u) * tiles] == seg_off[(size_t)bp * tiles];
}

template <int LOGW>
__device__ __forceinline__ void rows_twiddles(float2 *tw) {
    constexpr int W = 1 << LOGW;
    for (int n = threadIdx.x; n < W / 2; n += ROWS_NT) {
        float sn, cs;
        sincospif(-2.0f * (float)n / (float)W, &sn, &cs);
        tw[n] = make_float2(cs, sn);
    }
}

template <int LOGW>
__global__ __launch_bounds__(ROWS_NT) void k_asm_splat_rows(
    uint32_t tiles, uint32_t tiles_x, uint32_t P, uint32_t H, const uint32_t *__restrict__ ranges,
    const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ dup_ids, const float *__restrict__ rec,
    const float *__restrict__ ccs, float2 *__restrict__ field) {
    constexpr int W = 1 << LOGW, TXN = W / 16, TPW = (TXN + 15) / 16;
    static_assert(LOGW >= 6 && LOGW <= ROWS_MAX_LOGW, "row-fused splat: widths 64 ... 512");
    __shared__ __attribute__((aligned(16))) float2 xs[rows_lds_float2<LOGW, ROWS_STG_FWD>()];
    __shared__ float2 tw[W / 2];
    const uint32_t ty = blockIdx.x, p = blockIdx.y, b = blockIdx.z, bp = b * P + p;
    if (asm_plane_empty(seg_off, bp, tiles)) return;
    rows_twiddles<LOGW>(tw);
    const uint32_t lane = threadIdx.x & 63u, lx = lane & 7u, ly = lane >> 3;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- phase 0: the lists of this wave's tiles, all four sub-tiles per lane, accumulators in registers ----
    // (staging overlays the transform buffer: 4 KB + masks per wave)
    float4 *stg = reinterpret_cast<float4 *>(xs) + (size_t)wave * ROWS_STG_FWD;
    float4 *sh0 = stg, *sh1 = stg + ACH, *sh2 = stg + 2 * ACH, *sh3 = stg + 3 * ACH;
    uint32_t *shm = reinterpret_cast<uint32_t *>(stg + 4 * ACH);
    float re[TPW][4][3], im[TPW][4][3];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c) { re[ti][s][c] = 0.0f; im[ti][s][c] = 0.0f; }
        const uint32_t tx = wave + 16u * ti;
        if (tx >= (uint32_t)TXN) continue;  // wave-uniform
        const uint32_t key = bp * tiles + ty * tiles_x + tx;
        const uint32_t X0 = tx * FGS_TILE, Y0 = ty * FGS_TILE;
        const float fx0 = (float)(X0 + lx), fy0 = (float)(Y0 + ly);
        const uint32_t start = ranges[2 * key], end = ranges[2 * key + 1];
        for (uint32_t base = start; base < end; base += ACH) {
            const uint32_t n = min((uint32_t)ACH, end - base);
            if (lane < n) {
                const uint32_t gid = dup_ids[base + lane];
                const float4 *r = reinterpret_cast<const float4 *>(rec + (size_t)gid * FGS_REC_FLOATS);
                const float4 q0 = r[0], q1 = r[1], q2 = r[2];
                const uint32_t bbx = __float_as_uint(q2.z), bby = __float_as_uint(q2.w);
                const uint32_t bx0 = bbx & 0xFFFFu, bx1 = bbx >> 16, by0 = bby & 0xFFFFu, by1 = bby >> 16;
                shm[lane] = subtile_mask(X0, Y0, bx0, bx1, by0, by1);
                const float4 *pz = reinterpret_cast<const float4 *>(ccs + (size_t)gid * 8);
                const float4 z0 = pz[0], z1 = pz[1];
                sh0[lane] = make_float4(q0.x, q0.y, q0.z * NEG_HALF_LOG2E, q0.w * NEG_HALF_LOG2E);
                sh1[lane] = make_float4(q1.x * NEG_HALF_LOG2E, q1.y, __uint_as_float(bx0 | ((bx1 - bx0) << 16)),
                                        __uint_as_float(by0 | ((by1 - by0) << 16)));
                sh2[lane] = z0;
                sh3[lane] = make_float4(z1.x, z1.y, 0.0f, 0.0f);
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = 0; j < n; ++j) {
                const float4 q0 = sh0[j], q1 = sh1[j], q2 = sh2[j], q3 = sh3[j];
                const uint32_t msk = __builtin_amdgcn_readfirstlane(shm[j]);
                const uint32_t bbx = __float_as_uint(q1.z), bby = __float_as_uint(q1.w);
                const float ca = q0.z, cbc = q0.w, cd = q1.x, op = q1.y;
                const float cc[3] = {q2.x, q2.y, q2.z}, cs[3] = {q2.w, q3.x, q3.y};
                const float dxs[2] = {fx0 - q0.x, fx0 + 8.0f - q0.x}, dys[2] = {fy0 - q0.y, fy0 + 8.0f - q0.y};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (!((msk >> s) & 1u)) continue;
                    const uint32_t px = X0 + 8u * (s & 1) + lx, py = Y0 + 8u * (s >> 1) + ly;
                    const bool in = (px - (bbx & 0xFFFFu)) < (bbx >> 16) && (py - (bby & 0xFFFFu)) < (bby >> 16);
                    const float dx = dxs[s & 1], dy = dys[s >> 1];
                    const float m = (ca * dx) * dx + (cbc * dx) * dy + (cd * dy) * dy;  // K m
                    const float G = in ? __builtin_amdgcn_exp2f(m) : 0.0f;
                    const float a = G * op;  // amplitude, DR:1270-1271
#pragma unroll
                    for (int c = 0; c < 3; ++c) { re[ti][s][c] += a * cc[c]; im[ti][s][c] += a * cs[c]; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();  // every wave is done with its staging: the buffer becomes the transform tile
    auto X = [&](int r, int line) -> float2 & { return xs[rows_slot(r, line)]; };
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            const uint32_t tx = wave + 16u * ti;
            if (tx >= (uint32_t)TXN) continue;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int px = (int)(tx * FGS_TILE + 8u * s2 + lx);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    // compile-time register indices: the half is selected, not indexed
                    const float vr = half ? re[ti][2 + s2][c] : re[ti][s2][c], vi = half ? im[ti][2 + s2][c] : im[ti][s2][c];
                    X(px, (int)ly * 3 + c) = make_float2(vr, vi);
                }
            }
        }
        __syncthreads();
        lds_fft_core<LOGW, ROWS_LINES, ROWS_NT, false>(X, tw);  // natural order in, bit-reversed order out
        const uint32_t y0 = ty * FGS_TILE + 8u * half;
        for (int idx = threadIdx.x; idx < ROWS_LINES * W; idx += ROWS_NT) {
            const int line = idx >> LOGW, kx = idx & (W - 1);
            const int row = line / 3, c = line - 3 * row;
            field[(((size_t)bp * 3 + c) * H + (y0 + row)) * W + kx] = X(bitrev<LOGW>(kx), line);
        }
        __syncthreads();
    }
}

// Backward: field holds the gradient spectra after the column pass (k_colfft_bwd); inverse row transform in LDS, then
// the splat adjoint of this tile row's lists (one gradient row per (tile, Gaussian) duplicate, as k_asm_splat<true>).
template <int LOGW>
__global__ __launch_bounds__(ROWS_NT) void k_asm_rows_splat_bwd(
    uint32_t tiles, uint32_t tiles_x, uint32_t P, uint32_t H, uint32_t dcap, const uint32_t *__restrict__ ranges,
    const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ dup_ids, const float *__restrict__ rec,
    const float *__restrict__ ccs,
    const uint32_t *__restrict__ dup_off, const float2 *__restrict__ field, float *__restrict__ grad_rows) {
    constexpr int W = 1 << LOGW, TXN = W / 16, TPW = (TXN + 15) / 16;
    __shared__ __attribute__((aligned(16))) float2 xs[rows_lds_float2<LOGW, ROWS_STG_BWD>()];
    __shared__ float2 tw[W / 2];
    const uint32_t ty = blockIdx.x, p = blockIdx.y, b = blockIdx.z, bp = b * P + p;
    if (asm_plane_empty(seg_off, bp, tiles)) return;
    rows_twiddles<LOGW>(tw);
    const uint32_t lane = threadIdx.x & 63u, lx = lane & 7u, ly = lane >> 3;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto X = [&](int r, int line) -> float2 & { return xs[rows_slot(r, line)]; };
    float re[TPW][4][3], im[TPW][4][3];  // field gradient at this lane's pixels
    for (int half = 0; half < 2; ++half) {
        const uint32_t y0 = ty * FGS_TILE + 8u * half;
        for (int idx = threadIdx.x; idx < ROWS_LINES * W; idx += ROWS_NT) {
            const int line = idx >> LOGW, kx = idx & (W - 1);
            const int row = line / 3, c = line - 3 * row;
            X(bitrev<LOGW>(kx), line) = field[(((size_t)bp * 3 + c) * H + (y0 + row)) * W + kx];
        }
        __syncthreads();  // (also orders the twiddles in front of the first transform)
        lds_fft_core<LOGW, ROWS_LINES, ROWS_NT, true>(X, tw);  // bit-reversed order in, natural order out, e^+, unnormalised
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            const uint32_t tx = wave + 16u * ti;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int px = (int)(tx * FGS_TILE + 8u * s2 + lx);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float2 g = tx < (uint32_t)TXN ? X(px, (int)ly * 3 + c) : make_float2(0.0f, 0.0f);
                    if (half) { re[ti][2 + s2][c] = g.x; im[ti][2 + s2][c] = g.y; }
                    else { re[ti][s2][c] = g.x; im[ti][s2][c] = g.y; }
                }
            }
        }
        __syncthreads();
    }
    // ---- the splat adjoint: staging + reduction scratch overlay the transform buffer ----
    float4 *stg = reinterpret_cast<float4 *>(xs) + (size_t)wave * ROWS_STG_BWD;
    float4 *sh0 = stg, *sh1 = stg + ACH, *sh2 = stg + 2 * ACH, *sh3 = stg + 3 * ACH;
    uint32_t *shm = reinterpret_cast<uint32_t *>(stg + 4 * ACH), *she = shm + ACH;
    float *red = reinterpret_cast<float *>(stg + 4 * ACH + ACH / 2);  // 16-byte aligned (wave_sum_addtid scratch)
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const uint32_t tx = wave + 16u * ti;
        if (tx >= (uint32_t)TXN) continue;
        const uint32_t key = bp * tiles + ty * tiles_x + tx;
        const uint32_t X0 = tx * FGS_TILE, Y0 = ty * FGS_TILE;
        const float fx0 = (float)(X0 + lx), fy0 = (float)(Y0 + ly);
        const uint32_t start = ranges[2 * key], end = ranges[2 * key + 1];
        for (uint32_t base = start; base < end; base += ACH) {
            const uint32_t n = min((uint32_t)ACH, end - base);
            if (lane < n) {
                const uint32_t gid = dup_ids[base + lane];
                const float4 *r = reinterpret_cast<const float4 *>(rec + (size_t)gid * FGS_REC_FLOATS);
                const float4 q0 = r[0], q1 = r[1], q2 = r[2];
                const uint32_t bbx = __float_as_uint(q2.z), bby = __float_as_uint(q2.w);
                const uint32_t bx0 = bbx & 0xFFFFu, bx1 = bbx >> 16, by0 = bby & 0xFFFFu, by1 = bby >> 16;
                shm[lane] = subtile_mask(X0, Y0, bx0, bx1, by0, by1);
                const uint32_t tx0 = bx0 / FGS_TILE, tx1 = (bx1 - 1) / FGS_TILE, ty0 = by0 / FGS_TILE;
                she[lane] = dup_off[gid] + (ty - ty0) * (tx1 - tx0 + 1) + (tx - tx0);
                const float4 *pz = reinterpret_cast<const float4 *>(ccs + (size_t)gid * 8);
                const float4 z0 = pz[0], z1 = pz[1];
                sh0[lane] = make_float4(q0.x, q0.y, q0.z * NEG_HALF_LOG2E, q0.w * NEG_HALF_LOG2E);
                sh1[lane] = make_float4(q1.x * NEG_HALF_LOG2E, q1.y, __uint_as_float(bx0 | ((bx1 - bx0) << 16)),
                                        __uint_as_float(by0 | ((by1 - by0) << 16)));
                sh2[lane] = z0;
                sh3[lane] = make_float4(z1.x, z1.y, 0.0f, 0.0f);
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = 0; j < n; ++j) {
                const float4 q0 = sh0[j], q1 = sh1[j], q2 = sh2[j], q3 = sh3[j];
                const uint32_t msk = __builtin_amdgcn_readfirstlane(shm[j]);
                const uint32_t bbx = __float_as_uint(q1.z), bby = __float_as_uint(q1.w);
                const float ca = q0.z, cbc = q0.w, cd = q1.x, op = q1.y;
                const float cc[3] = {q2.x, q2.y, q2.z}, cs[3] = {q2.w, q3.x, q3.y};
                float v_u = 0, v_v = 0, v_ca = 0, v_cbc = 0, v_cd = 0, v_op = 0;
                float v_cc[3] = {0, 0, 0}, v_cs[3] = {0, 0, 0};
                const float dxs[2] = {fx0 - q0.x, fx0 + 8.0f - q0.x}, dys[2] = {fy0 - q0.y, fy0 + 8.0f - q0.y};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (!((msk >> s) & 1u)) continue;
                    const uint32_t px = X0 + 8u * (s & 1) + lx, py = Y0 + 8u * (s >> 1) + ly;
                    const bool in = (px - (bbx & 0xFFFFu)) < (bbx >> 16) && (py - (bby & 0xFFFFu)) < (bby >> 16);
                    const float dx = dxs[s & 1], dy = dys[s >> 1];
                    const float m = (ca * dx) * dx + (cbc * dx) * dy + (cd * dy) * dy;
                    const float G = in ? __builtin_amdgcn_exp2f(m) : 0.0f;
                    const float a = G * op;
                    float da = 0.0f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        da += cc[c] * re[ti][s][c] + cs[c] * im[ti][s][c];
                        v_cc[c] += a * re[ti][s][c]; v_cs[c] += a * im[ti][s][c];
                    }
                    v_op += da * G;
                    const float dm = -0.5f * (da * op) * G;
                    v_ca += dm * dx * dx; v_cbc += dm * dx * dy; v_cd += dm * dy * dy;
                    const float dmk = (da * op) * G * 0.69314718055994530942f;  // -0.5 / K = ln 2 (unscaled conic, see k_asm_splat)
                    v_u -= dmk * (2.0f * ca * dx + cbc * dy);
                    v_v -= dmk * (cbc * dx + 2.0f * cd * dy);
                }
                const float vals[12] = {v_u, v_v, v_ca, v_cbc, v_cd, v_op, v_cc[0], v_cc[1], v_cc[2], v_cs[0], v_cs[1], v_cs[2]};
                const float tot = wave_sum_addtid<12>(red, vals, lane);
                const uint32_t e = she[j];
                if ((lane & 3u) == 3u && lane < 48u && e < dcap) grad_rows[(size_t)e * FGS_GROW_FLOATS + (lane >> 2)] = tot;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}
