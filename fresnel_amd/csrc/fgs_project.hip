// Stage 1: 3D -> 2D projection, 2D covariance, radius, visibility, fp64 bbox, conic.
// Replaces DR:98-120, DR:123-195, DR:452-487, DR:541-543, DR:578-579, DR:594-597 of the
// reference (scripts/models/differentiable_renderer.py) for a whole batch in one launch.
//
// HBM-bound streaming kernel: one thread per Gaussian, 56 B in (pos/scale/quat/color/opacity
// as the reference's AoS tensors -- consecutive lanes read consecutive 12/16-B rows, so each
// wave's loads are fully coalesced), 56 B out (48-B record + depth key + tile count).
//
// "Canonical fp32": every operation below that feeds an INTEGER decision (visibility,
// bbox, depth key) is individually rounded, in the association order documented in
// DESIGN.md -- this file is compiled with -ffp-contract=off and uses IEEE division/sqrt,
// so those decisions are bit-identical to the CPU oracle's.
#include "fgs_internal.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float clamp_min(float x, float lo) { return x < lo ? lo : x; }  // NaN-propagating
__device__ __forceinline__ float clamp_max(float x, float hi) { return x > hi ? hi : x; }
__device__ __forceinline__ float sgnf(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }

struct Proj {
    float xc, yc, zc, dep;
    float w, x, y, z, nrm;   // normalised quaternion + raw norm
    float Rc[3][3];          // view_rot @ R
    float M[3][3];           // Rc * diag(s)
    float S[3][3];           // cov3d
    float zs, z2, J00, J02, J11, J12;
    float T0[3], T1[3];
    float a, b, c, d, u, v;
};

__device__ __forceinline__ void project_one(const float *__restrict__ V, float fx, float fy, float cx,
                                            float cy, const float p[3], const float s[3],
                                            const float q[4], Proj &o) {
    o.xc = ((V[0] * p[0] + V[1] * p[1]) + V[2] * p[2]) + V[3];
    o.yc = ((V[4] * p[0] + V[5] * p[1]) + V[6] * p[2]) + V[7];
    o.zc = ((V[8] * p[0] + V[9] * p[1]) + V[10] * p[2]) + V[11];
    o.dep = -o.zc;
    float nrm = __fsqrt_rn(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
    o.nrm = nrm;
    if (nrm < 1e-12f) nrm = 1e-12f;
    const float w = __fdiv_rn(q[0], nrm), x = __fdiv_rn(q[1], nrm), y = __fdiv_rn(q[2], nrm),
                z = __fdiv_rn(q[3], nrm);
    o.w = w; o.x = x; o.y = y; o.z = z;
    float R[3][3];
    R[0][0] = (1.0f - (2.0f * y) * y) - (2.0f * z) * z;
    R[0][1] = (2.0f * x) * y - (2.0f * w) * z;
    R[0][2] = (2.0f * x) * z + (2.0f * w) * y;
    R[1][0] = (2.0f * x) * y + (2.0f * w) * z;
    R[1][1] = (1.0f - (2.0f * x) * x) - (2.0f * z) * z;
    R[1][2] = (2.0f * y) * z - (2.0f * w) * x;
    R[2][0] = (2.0f * x) * z - (2.0f * w) * y;
    R[2][1] = (2.0f * y) * z + (2.0f * w) * x;
    R[2][2] = (1.0f - (2.0f * x) * x) - (2.0f * y) * y;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            o.Rc[i][j] = (V[4 * i] * R[0][j] + V[4 * i + 1] * R[1][j]) + V[4 * i + 2] * R[2][j];
            o.M[i][j] = o.Rc[i][j] * s[j];
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o.S[i][j] = (o.M[i][0] * o.M[j][0] + o.M[i][1] * o.M[j][1]) + o.M[i][2] * o.M[j][2];
    o.zs = clamp_min(fabsf(o.zc), 0.01f) * sgnf(o.zc + 1e-8f);
    o.z2 = o.zs * o.zs;
    o.J00 = __fdiv_rn(fx, -o.zs);
    o.J02 = __fdiv_rn(fx * o.xc, o.z2);
    o.J11 = __fdiv_rn(fy, o.zs);
    o.J12 = __fdiv_rn(fy * o.yc, o.z2);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        o.T0[j] = o.J00 * o.S[0][j] + o.J02 * o.S[2][j];
        o.T1[j] = o.J11 * o.S[1][j] + o.J12 * o.S[2][j];
    }
    o.a = o.T0[0] * o.J00 + o.T0[2] * o.J02;
    o.b = o.T0[1] * o.J11 + o.T0[2] * o.J12;
    o.c = o.T1[0] * o.J00 + o.T1[2] * o.J02;
    o.d = o.T1[1] * o.J11 + o.T1[2] * o.J12;
    o.u = __fdiv_rn(fx * o.xc, -o.zs) + cx;
    o.v = __fdiv_rn(fy * (-o.yc), -o.zs) + cy;
}

__global__ __launch_bounds__(256) void k_project(
    int32_t N, int32_t W, int32_t H, int32_t num_cameras, float max_radius,
    const float *__restrict__ cams, const float *__restrict__ pos, const float *__restrict__ scale,
    const float *__restrict__ quat, const float *__restrict__ color, const float *__restrict__ opacity,
    float *__restrict__ rec, uint32_t *__restrict__ depth_key, uint32_t *__restrict__ tile_count,
    uint32_t *__restrict__ layer, int32_t num_planes, float plane_near, float plane_far, int32_t tile_w,
    uint32_t *__restrict__ key_bits, uint32_t *__restrict__ zero_words, uint32_t zero_count) {
    // the depth sort's hand-off words (fgs_sort.hip, sort_mode bits 1-2 = 3) must read "not published" when its passes start: this
    // launch precedes them on the stream, so it clears them on the side (<= 1 store per thread)
    for (uint32_t i = (blockIdx.y * gridDim.x + blockIdx.x) * 256u + threadIdx.x; i < zero_count; i += gridDim.x * gridDim.y * 256u)
        zero_words[i] = 0u;
    // grid (blocks per image, B): a block never straddles two images, so that its share of the image's key statistics
    // (below) is one record
    const int32_t n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    uint32_t key = 0xFFFFFFFFu;
    if (n < N) {
        const int32_t idx = b * N + n;
        const float *__restrict__ cam = cams + (num_cameras > 1 ? b : 0) * FGS_CAMERA_FLOATS;
        const float p[3] = {pos[3 * idx], pos[3 * idx + 1], pos[3 * idx + 2]};
        const float s[3] = {scale[3 * idx], scale[3 * idx + 1], scale[3 * idx + 2]};
        const float4 q4 = reinterpret_cast<const float4 *>(quat)[idx];
        const float q[4] = {q4.x, q4.y, q4.z, q4.w};
        const float fx = cam[16], fy = cam[17], cx = cam[18], cy = cam[19], nearp = cam[20], farp = cam[21];
        Proj o;
        project_one(cam, fx, fy, cx, cy, p, s, q, o);

        // radius, DR:471-485
        const float tr = o.a + o.d;
        float det = o.a * o.d - o.b * o.c;
        det = clamp_min(det, 1e-6f);
        const float disc = clamp_min(tr * tr - 4.0f * det, 0.0f);
        const float lam = __fdiv_rn(tr + __fsqrt_rn(disc), 2.0f);
        float r = 3.0f * __fsqrt_rn(clamp_min(lam, 1e-6f));
        r = clamp_max(r, max_radius);
        // visibility, DR:541-543
        const bool vis = (o.dep > nearp) && (o.dep < farp) && (o.u + r > 0.0f) && (o.u - r < (float)W) &&
                         (o.v + r > 0.0f) && (o.v - r < (float)H);
        // bbox in fp64, DR:594-597
        int32_t x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        if (vis) {
            const double ud = (double)o.u, vd = (double)o.v, rd = (double)r;
            double e;
            e = trunc(ud - rd); x0 = e < 0.0 ? 0 : (e > (double)W ? W : (int32_t)e);
            e = trunc(ud + rd) + 1.0; x1 = e > (double)W ? W : (e < 0.0 ? 0 : (int32_t)e);
            e = trunc(vd - rd); y0 = e < 0.0 ? 0 : (e > (double)H ? H : (int32_t)e);
            e = trunc(vd + rd) + 1.0; y1 = e > (double)H ? H : (e < 0.0 ? 0 : (int32_t)e);
        }
        uint32_t ntiles = 0;
        if (vis && x0 < x1 && y0 < y1)
            ntiles = (uint32_t)(((x1 - 1) / tile_w - x0 / tile_w + 1) * ((y1 - 1) / FGS_TILE - y0 / FGS_TILE + 1));
        // inverse of cov + 1e-4 I, DR:578-579 (closed form)
        const float ar = o.a + 1e-4f, dr = o.d + 1e-4f;
        const float detr = ar * dr - o.b * o.c;
        const float ia = __fdiv_rn(dr, detr);
        const float ibc = __fdiv_rn(-o.b, detr) + __fdiv_rn(-o.c, detr);
        const float id = __fdiv_rn(ar, detr);

        float4 *out = reinterpret_cast<float4 *>(rec + (size_t)idx * FGS_REC_FLOATS);
        out[0] = make_float4(o.u, o.v, ia, ibc);
        out[1] = make_float4(id, opacity[idx], color[3 * idx], color[3 * idx + 1]);
        out[2] = make_float4(color[3 * idx + 2], o.dep, __uint_as_float((uint32_t)x0 | ((uint32_t)x1 << 16)),
                             __uint_as_float((uint32_t)y0 | ((uint32_t)y1 << 16)));
        key = vis ? fgs_float_key(o.dep) : 0xFFFFFFFFu;
        depth_key[idx] = key;
        tile_count[idx] = ntiles;
        if (layer) {
            // nearest depth plane, DR:1106 (torch.linspace) + DR:1147-1148 (first argmin of |depth - plane|)
            const float step = (plane_far - plane_near) / (float)(num_planes - 1);
            int32_t best = 0;
            float bd = 3.4e38f;
            for (int32_t k = 0; k < num_planes; ++k) {
                const float pk = (k < num_planes / 2) ? plane_near + step * (float)k
                                                      : plane_far - step * (float)(num_planes - 1 - k);
                const float dd = fabsf(o.dep - pk);
                if (dd < bd) { bd = dd; best = k; }
            }
            layer[idx] = (uint32_t)best;
        }
    }
    // Which bits of the depth keys VARY over an image's visible Gaussians decides how many radix passes its depth sort needs
    // (fgs_sort.hip: the keys are compressed to those bits -- zone-snapped depths, BASELINE config 4, differ in 3 bits and sort
    // in one pass instead of four).  Per block: OR and AND of the visible keys and the visible / culled flags, three words;
    // every block of the sort folds the <= N / 256 records of its image.  No atomics, no extra launch.
    {
        const bool visk = key != 0xFFFFFFFFu;
        uint32_t vor = visk ? key : 0u, vand = key;  // (a culled key is all ones: neutral for the AND)
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { vor |= __shfl_xor(vor, o2, 64); vand &= __shfl_xor(vand, o2, 64); }
        const uint32_t fl = (__ballot(visk) != 0ull ? 2u : 0u) | (__ballot(n < N && !visk) != 0ull ? 1u : 0u);
        __shared__ uint32_t wv[4][3];
        const uint32_t wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63u) == 0) { wv[wave][0] = vor; wv[wave][1] = vand; wv[wave][2] = fl; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t *o3 = key_bits + ((size_t)b * gridDim.x + blockIdx.x) * 4;
            o3[0] = (wv[0][0] | wv[1][0]) | (wv[2][0] | wv[3][0]);
            o3[1] = (wv[0][1] & wv[1][1]) & (wv[2][1] & wv[3][1]);
            o3[2] = (wv[0][2] | wv[1][2]) | (wv[2][2] | wv[3][2]);
            o3[3] = 0u;  // (the record's fourth word: nobody reads it, but `saved` holds no uninitialised words)
        }
    }
}

// Gradient-row reduction + projection backward (autograd of DR:98-195 + DR:578-579).
// Blend path, first half of the projection backward: sum every Gaussian's contiguous 40-byte gradient rows (moments
// of dL/dG about the mean, colour and depth sums; k_composite_bwd) in a fixed order -- FOUR lanes per Gaussian in
// DEPTH-RANK order (rows are laid out in emission order, so neighbouring lanes read neighbouring rows), lane `sub`
// takes rows sub, sub + 4, ..., two rows in flight per lane; moments in double; two quad shuffles combine the partial
// sums.  The totals get the factors of the chain through m' = K m, G = exp2(m'), alpha = G opacity (first moments x
// ln2 opacity, second moments x K ln2 opacity) and go to sums[input index][12].  Streaming kernel: ~40 VGPRs, full occupancy.
__global__ __launch_bounds__(256) void k_row_sum(int32_t total, int32_t N, uint32_t dcap,
                                                 const uint32_t *__restrict__ order,
                                                 const uint32_t *__restrict__ dup_off,
                                                 const uint32_t *__restrict__ tile_count,
                                                 const float *__restrict__ grad_rows, const float *__restrict__ rec,
                                                 float *__restrict__ sums) {
    const int32_t tid = blockIdx.x * 256 + threadIdx.x;
    const int32_t ri = tid >> 2;
    const uint32_t sub = threadIdx.x & 3u;
    const bool live = ri < total;
    float acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int32_t idx = 0;
    if (live) {
        idx = ri / N * N + (int32_t)order[ri];
        const uint32_t off = dup_off[idx];
        uint32_t cnt = tile_count[idx];
        if (off >= dcap) cnt = 0; else if (cnt > dcap - off) cnt = dcap - off;
        double m[6] = {0, 0, 0, 0, 0, 0};
        const float *base = grad_rows + (size_t)off * FGS_BLEND_ROW_FLOATS;
        uint32_t k = sub;
#if FGS_BLEND_ROW_FLOATS == 16
        // 64-byte rows (build experiment): three aligned 16-byte loads per row
        auto add_row = [&](const float4 a, const float4 b, const float4 c) {
            m[0] += a.x; m[1] += a.y; m[2] += a.z; m[3] += a.w; m[4] += b.x; m[5] += b.y;
            acc[6] += b.z; acc[7] += b.w; acc[8] += c.x; acc[9] += c.y;
        };
        for (; k + 4 < cnt; k += 8) {
            const float4 *r = reinterpret_cast<const float4 *>(base + (size_t)k * 16);
            const float4 *r2 = reinterpret_cast<const float4 *>(base + (size_t)(k + 4) * 16);
            const float4 a = r[0], b = r[1], c = r[2], a2 = r2[0], b2 = r2[1], c2 = r2[2];
            add_row(a, b, c); add_row(a2, b2, c2);
        }
        if (k < cnt) {
            const float4 *r = reinterpret_cast<const float4 *>(base + (size_t)k * 16);
            add_row(r[0], r[1], r[2]);
        }
#else
        for (; k + 4 < cnt; k += 8) {  // rows k and k + 4
            const float2 *r = reinterpret_cast<const float2 *>(base + (size_t)k * FGS_BLEND_ROW_FLOATS);
            const float2 *r2 = reinterpret_cast<const float2 *>(base + (size_t)(k + 4) * FGS_BLEND_ROW_FLOATS);
            const float2 a = r[0], bq = r[1], cq = r[2], dq = r[3], eq = r[4];
            const float2 a2 = r2[0], bq2 = r2[1], cq2 = r2[2], dq2 = r2[3], eq2 = r2[4];
            m[0] += a.x; m[1] += a.y; m[2] += bq.x; m[3] += bq.y; m[4] += cq.x; m[5] += cq.y;
            acc[6] += dq.x; acc[7] += dq.y; acc[8] += eq.x; acc[9] += eq.y;
            m[0] += a2.x; m[1] += a2.y; m[2] += bq2.x; m[3] += bq2.y; m[4] += cq2.x; m[5] += cq2.y;
            acc[6] += dq2.x; acc[7] += dq2.y; acc[8] += eq2.x; acc[9] += eq2.y;
        }
        if (k < cnt) {
            const float2 *r = reinterpret_cast<const float2 *>(base + (size_t)k * FGS_BLEND_ROW_FLOATS);
            const float2 a = r[0], bq = r[1], cq = r[2], dq = r[3], eq = r[4];
            m[0] += a.x; m[1] += a.y; m[2] += bq.x; m[3] += bq.y; m[4] += cq.x; m[5] += cq.y;
            acc[6] += dq.x; acc[7] += dq.y; acc[8] += eq.x; acc[9] += eq.y;
        }
#endif
        // k_composite_bwd works with alpha / 0.99 and 0.99 x colours: its moment and sum-dG rows carry a factor 0.99, its
        // colour / depth rows 1 / 0.99
        const double ia = 1.0 / (double)0.99f;
        const double hp = 0.69314718055994530942 * (double)rec[(size_t)idx * FGS_REC_FLOATS + R_OP] * ia;
        acc[0] = (float)(hp * m[0]); acc[1] = (float)(hp * m[1]);
        acc[2] = (float)(-0.72134752044448170368 * hp * m[2]); acc[3] = (float)(-0.72134752044448170368 * hp * m[3]);
        acc[4] = (float)(-0.72134752044448170368 * hp * m[4]); acc[5] = (float)(m[5] * ia);
        acc[6] *= 0.99f; acc[7] *= 0.99f; acc[8] *= 0.99f; acc[9] *= 0.99f;
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        acc[k] += __shfl_xor(acc[k], 1, 64);
        acc[k] += __shfl_xor(acc[k], 2, 64);
    }
    if (!live || sub != 0) return;
    float4 *o = reinterpret_cast<float4 *>(sums + (size_t)idx * 12);  // by input index: the adjoint kernel runs in input order
    o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    o[2] = make_float4(acc[8], acc[9], 0.0f, 0.0f);
}

// One thread per Gaussian, walked in DEPTH-RANK order so that the rows read by neighbouring
// lanes are neighbours in memory (rows are laid out in emission order).  Sums the Gaussian's
// contiguous 48-byte gradient rows in a fixed order (deterministic), then chains
// (dL/dmean2d, dL/dconic, dL/ddepth) to (dL/dpos, dL/dscale, dL/dquat), recomputing the forward
// intermediates from the inputs (cheaper than saving ~60 floats per Gaussian).
// MODE 1 (ASM): rows come from the angular-spectrum splat backward (k_asm_splat<BWD>, fgs_asm.hip) and hold the MOMENTS of
//   t = dL/da G about the Gaussian's mean (round 4): slots 0-1 the first moments (sum t dx, sum t dy), 2-4 the second moments
//   (sum t dx^2, sum t dx dy, sum t dy^2), 5 the zeroth (sum t = dL/dopacity before the chain), 6-8 dL/d(c cos phi)[3],
//   9-11 dL/d(c sin phi)[3].  The chain through a = G opacity and m is applied HERE, once per Gaussian, in double:
//   dL/dconic = -1/2 opacity (second moments), dL/d(u, v) = 1/2 opacity conic_sym (first moments).  This row format is a
//   contract between the two translation units.  Colour and phase gradients are formed here and no gradient flows through
//   depth (the depth only selects the plane, DR:1147-1148).
// MODE 2 (WaveFieldRenderer): ASM rows widened to 16 floats, slot 12 = dL/ddepth (amplitude-weighted depth map)
// PRESUM: the rows were already summed by k_row_sum (blend path): ONE thread per Gaussian reads its ten totals from
// `grad_rows` (= the per-Gaussian sums, [input index][12] floats) -- all 64 lanes of a wave run the double-precision adjoint
// instead of every fourth, and the streaming part no longer runs at this kernel's 3 waves per SIMD (160 VGPRs).
template <int MODE, bool PRESUM = false>
__global__ __launch_bounds__(256) void k_project_bwd(
    int32_t total, int32_t N, int32_t num_cameras, uint32_t dcap, const float *__restrict__ cams,
    const float *__restrict__ pos, const float *__restrict__ scale, const float *__restrict__ quat,
    const uint32_t *__restrict__ depth_key, const uint32_t *__restrict__ order,
    const uint32_t *__restrict__ dup_off, const uint32_t *__restrict__ tile_count,
    const float *__restrict__ grad_rows, float *__restrict__ g_pos, float *__restrict__ g_scale,
    float *__restrict__ g_quat, float *__restrict__ g_color, float *__restrict__ g_opacity,
    float *__restrict__ g_phase, const float *__restrict__ color, const float *__restrict__ phase,
    int32_t phase_channels, uint32_t rows_per_dup, const float *__restrict__ rec) {
    // four lanes per Gaussian: lane `sub` sums rows sub, sub+4, ... (neighbouring lanes read
    // neighbouring 48-byte rows), then two quad shuffles combine the partial sums in a fixed order
    const int32_t tid = blockIdx.x * 256 + threadIdx.x;
    const int32_t ri = PRESUM ? tid : tid >> 2;
    const uint32_t sub = PRESUM ? 0u : threadIdx.x & 3u;
    const bool live = ri < total;
    const int32_t b = live ? ri / N : 0;
    // PRESUM: Gaussians in INPUT order -- every load and store of this kernel is then coalesced (k_row_sum scattered
    // its totals to sums[input index]); in depth-rank order the ~28 scattered 4-12 byte accesses per Gaussian made
    // the adjoint take 29 us at config 3
    const int32_t idx = live ? (PRESUM ? ri : b * N + (int32_t)order[ri]) : 0;
    float gp[3] = {0, 0, 0}, gs[3] = {0, 0, 0}, gq[4] = {0, 0, 0, 0};
    constexpr bool ASM = MODE != 0;
    constexpr int ROWF = MODE == 2 ? 16 : FGS_GROW_FLOATS;
    float acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (PRESUM) {
        if (live) {
            const float4 *r = reinterpret_cast<const float4 *>(grad_rows + (size_t)idx * 12);
            const float4 a = r[0], bq = r[1], cq = r[2];
            acc[0] = a.x; acc[1] = a.y; acc[2] = a.z; acc[3] = a.w; acc[4] = bq.x; acc[5] = bq.y; acc[6] = bq.z;
            acc[7] = bq.w; acc[8] = cq.x; acc[9] = cq.y;
        }
    } else if (live) {
        // rows_per_dup gradient rows per duplicate (4 on the phase path: one per sub-tile wave), contiguous
        const uint32_t cnt = tile_count[idx] * rows_per_dup, off = dup_off[idx] * rows_per_dup;
        if (MODE == 0) {  // (the blend path's rows are summed by k_row_sum: PRESUM)
            // phase path: FOUR rows per duplicate, one per 8x8 sub-tile wave of k_phase_bwd, which writes
            // only the rows of sub-tiles the bbox touches -- the same integer test decides here which rows exist
            // (lane `sub` of the quad owns sub-tile `sub` of every duplicate)
            const uint32_t bbx = __float_as_uint(rec[(size_t)idx * FGS_REC_FLOATS + R_BBX]);
            const uint32_t bby = __float_as_uint(rec[(size_t)idx * FGS_REC_FLOATS + R_BBY]);
            const uint32_t x0 = bbx & 0xFFFFu, x1 = bbx >> 16, y0 = bby & 0xFFFFu, y1 = bby >> 16;
            const uint32_t tx0 = x0 / FGS_TILE, ty0 = y0 / FGS_TILE, tw = (x1 - 1) / FGS_TILE - tx0 + 1;
            uint32_t tx = 0, ty = 0;  // tile of duplicate k / 4, walked without a division
            for (uint32_t k = sub; k < cnt && off + k < dcap * rows_per_dup; k += 4) {
                const uint32_t sx = (tx0 + tx) * FGS_TILE + 8u * (sub & 1u), sy = (ty0 + ty) * FGS_TILE + 8u * (sub >> 1);
                if (++tx == tw) { tx = 0; ++ty; }
                if (x1 <= sx || x0 >= sx + 8u || y1 <= sy || y0 >= sy + 8u) continue;  // row never written
                const float4 *r = reinterpret_cast<const float4 *>(grad_rows + (size_t)(off + k) * ROWF);
                const float4 a = r[0], bq = r[1], cq = r[2];
                acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
                acc[4] += bq.x; acc[5] += bq.y; acc[6] += bq.z; acc[7] += bq.w;
                acc[8] += cq.x; acc[9] += cq.y; acc[10] += cq.z; acc[11] += cq.w;
            }
            // k_phase_bwd leaves the moments of -2 dL/dm in slots 0-4 (round 5: the constants ride on the per-Gaussian sums, not on every
            // list entry); the chain below takes the first moments of dL/dm' (m' = K m: x -1/2 x 1/K = ln 2) and dL/dconic (x -1/2)
            acc[0] *= 0.69314718055994530942f; acc[1] *= 0.69314718055994530942f;
            acc[2] *= -0.5f; acc[3] *= -0.5f; acc[4] *= -0.5f;
        } else {
            // rows sub, sub + 4, ...: two rows in flight per lane, added in the same order as one at a time
            const uint32_t room = dcap * rows_per_dup, lim = off >= room ? 0u : min(cnt, room - off);
            auto add_row = [&](const float4 &a, const float4 &bq, const float4 &cq, float extra) {
                if (MODE == 2) acc[12] += extra;
                acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
                acc[4] += bq.x; acc[5] += bq.y; acc[6] += bq.z; acc[7] += bq.w;
                acc[8] += cq.x; acc[9] += cq.y; acc[10] += cq.z; acc[11] += cq.w;
            };
            uint32_t k = sub;
            for (; k + 4 < lim; k += 8) {
                const float4 *r = reinterpret_cast<const float4 *>(grad_rows + (size_t)(off + k) * ROWF);
                const float4 *r2 = reinterpret_cast<const float4 *>(grad_rows + (size_t)(off + k + 4) * ROWF);
                const float4 a = r[0], bq = r[1], cq = r[2], a2 = r2[0], bq2 = r2[1], cq2 = r2[2];
                const float e = MODE == 2 ? r[3].x : 0.0f, e2 = MODE == 2 ? r2[3].x : 0.0f;
                add_row(a, bq, cq, e);
                add_row(a2, bq2, cq2, e2);
            }
            if (k < lim) {
                const float4 *r = reinterpret_cast<const float4 *>(grad_rows + (size_t)(off + k) * ROWF);
                const float4 a = r[0], bq = r[1], cq = r[2];
                add_row(a, bq, cq, MODE == 2 ? r[3].x : 0.0f);
            }
        }
    }
    if constexpr (!PRESUM) {
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            acc[k] += __shfl_xor(acc[k], 1, 64);
            acc[k] += __shfl_xor(acc[k], 2, 64);
        }
    }
    if (!live || sub != 0) return;
    const float4 s0 = make_float4(acc[0], acc[1], acc[2], acc[3]);
    const float4 s1 = make_float4(acc[4], acc[5], acc[6], acc[7]);
    const float4 s2 = make_float4(acc[8], acc[9], acc[10], acc[11]);
    const float g_mean[2] = {s0.x, s0.y};
    // ASM / wave rows (k_asm_splat<BWD>) hold the MOMENTS of t = dL/da G about the mean -- slots 0-1 first, 2-4 second,
    // 5 zeroth -- and the chain through a = G op, m is applied here: dL/dconic = -1/2 op (second moments)
    const float op_asm = ASM ? rec[(size_t)idx * FGS_REC_FLOATS + R_OP] : 0.0f;
    const float kcon = ASM ? -0.5f * op_asm : 1.0f;
    const float g_conic[3] = {kcon * s0.z, kcon * s0.w, kcon * s1.x};
    float g_depth = s2.y;
    g_opacity[idx] = s1.y;
    if (!ASM) {
        if (g_phase) g_phase[idx] = s2.z;
        g_color[3 * idx] = s1.z; g_color[3 * idx + 1] = s1.w; g_color[3 * idx + 2] = s2.x;
    } else {
        g_depth = MODE == 2 ? acc[12] : 0.0f;
        const float dcc[3] = {s1.z, s1.w, s2.x}, dcs[3] = {s2.y, s2.z, acc[11]};
        float gph = 0.0f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float ph = phase_channels == 3 ? phase[3 * idx + ch] : phase[idx];
            float sn, cs;
            sincosf(ph, &sn, &cs);
            const float col = color[3 * idx + ch];
            g_color[3 * idx + ch] = dcc[ch] * cs + dcs[ch] * sn;
            const float gp1 = col * (dcs[ch] * cs - dcc[ch] * sn);
            if (phase_channels == 3) g_phase[3 * idx + ch] = gp1; else gph += gp1;
        }
        if (phase_channels != 3) g_phase[idx] = gph;
    }
    if (depth_key[idx] != 0xFFFFFFFFu) {
        // Projection adjoint in DOUBLE precision, recomputing the projection from the fp32 inputs.  The kernel is
        // bound by the gradient-row reads, so this costs nothing measurable, and it keeps the chain through the
        // regularised 2x2 covariance inverse accurate for needle / disc Gaussians (scale ratios of 100:1 and
        // more), where an fp32 chain -- this one earlier, and the reference's autograd -- loses all digits.
        const float *__restrict__ Vf = cams + (num_cameras > 1 ? b : 0) * FGS_CAMERA_FLOATS;
        double V[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) V[i][j] = Vf[4 * i + j];
        const double fx = Vf[16], fy = Vf[17];
        const double p[3] = {pos[3 * idx], pos[3 * idx + 1], pos[3 * idx + 2]};
        const double s[3] = {scale[3 * idx], scale[3 * idx + 1], scale[3 * idx + 2]};
        const float4 q4 = reinterpret_cast<const float4 *>(quat)[idx];
        const double q0[4] = {q4.x, q4.y, q4.z, q4.w};
        const double xc = V[0][0] * p[0] + V[0][1] * p[1] + V[0][2] * p[2] + V[0][3];
        const double yc = V[1][0] * p[0] + V[1][1] * p[1] + V[1][2] * p[2] + V[1][3];
        const double zc = V[2][0] * p[0] + V[2][1] * p[1] + V[2][2] * p[2] + V[2][3];
        const double nrm = sqrt(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);
        // (one reciprocal per denominator instead of a double division per use: 21 divisions -> 3)
        const double inn = 1.0 / (nrm < 1e-12 ? 1e-12 : nrm);
        const double w = q0[0] * inn, x = q0[1] * inn, y = q0[2] * inn, z = q0[3] * inn;
        const double R[3][3] = {{1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y},
                                {2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x},
                                {2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y}};
        double Rc[3][3], M[3][3], S[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                Rc[i][j] = V[i][0] * R[0][j] + V[i][1] * R[1][j] + V[i][2] * R[2][j];
                M[i][j] = Rc[i][j] * s[j];
            }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) S[i][j] = M[i][0] * M[j][0] + M[i][1] * M[j][1] + M[i][2] * M[j][2];
        const double az = fabs(zc);
        const double sg = (double)sgnf((float)zc + 1e-8f);  // same sign convention as the fp32 forward
        const double zs = (az < 0.01 ? 0.01 : az) * sg;
        const double izs = 1.0 / zs, iz2 = izs * izs, iz3 = iz2 * izs;
        const double J[2][3] = {{-fx * izs, 0.0, fx * xc * iz2}, {0.0, fy * izs, fy * yc * iz2}};
        double T[2][3], C2[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) T[i][j] = J[i][0] * S[0][j] + J[i][1] * S[1][j] + J[i][2] * S[2][j];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) C2[i][j] = T[i][0] * J[j][0] + T[i][1] * J[j][1] + T[i][2] * J[j][2];
        const double ar = C2[0][0] + (double)1e-4f, dr = C2[1][1] + (double)1e-4f, cb = C2[0][1], cc = C2[1][0];
        const double rdet = 1.0 / (ar * dr - cb * cc);
        const double Y[2][2] = {{dr * rdet, -cb * rdet}, {-cc * rdet, ar * rdet}};
        const double GY[2][2] = {{g_conic[0], g_conic[1]}, {g_conic[1], g_conic[2]}};
        double tmp[2][2], G2[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) tmp[i][j] = Y[0][i] * GY[0][j] + Y[1][i] * GY[1][j];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) G2[i][j] = -(tmp[i][0] * Y[j][0] + tmp[i][1] * Y[j][1]);
        // dL/dSigma = J^T G2 J
        double GS[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2) acc += J[a2][i] * G2[a2][b2] * J[b2][j];
                GS[i][j] = acc;
            }
        // dL/dJ = G2 (J S^T) + G2^T (J S)
        double JS[2][3], JSt[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                JS[i][j] = J[i][0] * S[0][j] + J[i][1] * S[1][j] + J[i][2] * S[2][j];
                JSt[i][j] = J[i][0] * S[j][0] + J[i][1] * S[j][1] + J[i][2] * S[j][2];
            }
        double GJ[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                GJ[i][j] = (G2[i][0] * JSt[0][j] + G2[i][1] * JSt[1][j]) + (G2[0][i] * JS[0][j] + G2[1][i] * JS[1][j]);
        double gu = g_mean[0], gv = g_mean[1];
        if (ASM) {
            // dL/d(u, v) = -dL/dm (2 a dx + (b + c) dy, (b + c) dx + 2 d dy) summed = 1/2 op conic_sym (first moments)
            const double h = 0.5 * (double)op_asm, cbc = Y[0][1] + Y[1][0];
            gu = h * (2.0 * Y[0][0] * (double)g_mean[0] + cbc * (double)g_mean[1]);
            gv = h * (cbc * (double)g_mean[0] + 2.0 * Y[1][1] * (double)g_mean[1]);
        } else if (MODE == 0 && phase_channels) {
            // MODE 0 reuses `phase_channels` as a flag: slots 0/1 of the rows hold the first moments
            // M = sum dm' (dx, dy) in exp2 units (k_composite_bwd); dL/d(u,v) = -K conic_sym M, K = -log2(e)/2
            const double kc = 0.72134752044448170368, cbc = Y[0][1] + Y[1][0];
            gu = kc * (2.0 * Y[0][0] * (double)g_mean[0] + cbc * (double)g_mean[1]);
            gv = kc * (cbc * (double)g_mean[0] + 2.0 * Y[1][1] * (double)g_mean[1]);
        }
        const double gxc = GJ[0][2] * fx * iz2 + gu * (-fx * izs);
        const double gyc = GJ[1][2] * fy * iz2 + gv * (fy * izs);
        const double gzs = GJ[0][0] * fx * iz2 + GJ[0][2] * (-2.0 * fx * xc * iz3) + GJ[1][1] * (-fy * iz2) +
                           GJ[1][2] * (-2.0 * fy * yc * iz3) + gu * (fx * xc * iz2) + gv * (-fy * yc * iz2);
        const double dzs = (az >= 0.01 ? (double)sgnf((float)zc) : 0.0) * sg;
        const double gpc[3] = {gxc, gyc, gzs * dzs - (double)g_depth};
#pragma unroll
        for (int j = 0; j < 3; ++j) gp[j] = (float)(V[0][j] * gpc[0] + V[1][j] * gpc[1] + V[2][j] * gpc[2]);
        // Sigma = M M^T -> dM = (GS + GS^T) M ; M = Rc diag(s)
        double GRc[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double gm = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) gm += (GS[i][k] + GS[k][i]) * M[k][j];
                acc += gm * Rc[i][j];
                GRc[i][j] = gm * s[j];
            }
            gs[j] = (float)acc;
        }
        double GR[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) GR[i][j] = V[0][i] * GRc[0][j] + V[1][i] * GRc[1][j] + V[2][i] * GRc[2][j];
        double gh[4];
        gh[0] = -2 * z * GR[0][1] + 2 * y * GR[0][2] + 2 * z * GR[1][0] - 2 * x * GR[1][2] - 2 * y * GR[2][0] + 2 * x * GR[2][1];
        gh[1] = 2 * y * GR[0][1] + 2 * z * GR[0][2] + 2 * y * GR[1][0] - 4 * x * GR[1][1] - 2 * w * GR[1][2] + 2 * z * GR[2][0] + 2 * w * GR[2][1] - 4 * x * GR[2][2];
        gh[2] = -4 * y * GR[0][0] + 2 * x * GR[0][1] + 2 * w * GR[0][2] + 2 * x * GR[1][0] + 2 * z * GR[1][2] - 2 * w * GR[2][0] + 2 * z * GR[2][1] - 4 * y * GR[2][2];
        gh[3] = -4 * z * GR[0][0] - 2 * w * GR[0][1] + 2 * x * GR[0][2] + 2 * w * GR[1][0] - 4 * z * GR[1][1] + 2 * y * GR[1][2] + 2 * x * GR[2][0] + 2 * y * GR[2][1];
        const double qh[4] = {w, x, y, z};
        if (nrm >= 1e-12) {
            const double dot = qh[0] * gh[0] + qh[1] * gh[1] + qh[2] * gh[2] + qh[3] * gh[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) gq[i] = (float)((gh[i] - qh[i] * dot) * inn);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) gq[i] = (float)(gh[i] * 1e12);
        }
    }
    g_pos[3 * idx] = gp[0]; g_pos[3 * idx + 1] = gp[1]; g_pos[3 * idx + 2] = gp[2];
    g_scale[3 * idx] = gs[0]; g_scale[3 * idx + 1] = gs[1]; g_scale[3 * idx + 2] = gs[2];
    reinterpret_cast<float4 *>(g_quat)[idx] = make_float4(gq[0], gq[1], gq[2], gq[3]);
}

}  // namespace

int fgs_launch_project(const FgsPlan &p, const float *cams, const float *pos, const float *scale,
                       const float *quat, const float *color, const float *opacity, char *saved,
                       hipStream_t st, int num_planes, float plane_near, float plane_far, uint32_t *zero_words,
                       uint32_t zero_count) {
    const dim3 grid((unsigned)((p.d.num_gaussians + 255) / 256), (unsigned)p.d.batch);
    hipLaunchKernelGGL(k_project, grid, dim3(256), 0, st, p.d.num_gaussians, p.d.width,
                       p.d.height, p.d.num_cameras, p.d.max_radius, cams, pos, scale, quat, color, opacity,
                       reinterpret_cast<float *>(saved + p.L.rec),
                       reinterpret_cast<uint32_t *>(saved + p.L.depth_key),
                       reinterpret_cast<uint32_t *>(saved + p.L.tile_count),
                       num_planes > 1 ? reinterpret_cast<uint32_t *>(saved + p.s_layer) : nullptr, num_planes,  // one plane: layer 0
                       plane_near, plane_far, p.tile_w, reinterpret_cast<uint32_t *>(saved + p.s_keybits), zero_words,
                       zero_words ? zero_count : 0u);
    FGS_LAUNCH_CHECK("k_project");
    return FGS_OK;
}

int fgs_launch_project_bwd(const FgsPlan &p, const float *cams, const float *pos, const float *scale,
                           const float *quat, const char *saved, const float *grad_rows, float *g_pos,
                           float *g_scale, float *g_quat, float *g_color, float *g_opacity, float *g_phase,
                           hipStream_t st, float *row_sums) {
    const int32_t total = p.d.batch * p.d.num_gaussians;
    const int grid = (int)(((size_t)total * 4 + 255) / 256);
    const uint32_t *depth_key = reinterpret_cast<const uint32_t *>(saved + p.L.depth_key);
    const uint32_t *order = reinterpret_cast<const uint32_t *>(saved + p.L.order);
    const uint32_t *dup_off = reinterpret_cast<const uint32_t *>(saved + p.L.dup_off);
    const uint32_t *tile_count = reinterpret_cast<const uint32_t *>(saved + p.L.tile_count);
    const float *rec = reinterpret_cast<const float *>(saved + p.L.rec);
    if (!p.d.use_phase) {
        if (!row_sums) { fgs_set_error("fgs_launch_project_bwd: the blend path needs the row-sum scratch"); return FGS_EINVAL; }
        // blend path: streaming row sums at full occupancy, then one thread per Gaussian for the adjoint
        hipLaunchKernelGGL(k_row_sum, dim3(grid), dim3(256), 0, st, total, p.d.num_gaussians,
                           (uint32_t)p.L.dup_capacity, order, dup_off, tile_count, grad_rows, rec, row_sums);
        FGS_LAUNCH_CHECK("k_row_sum");
        hipLaunchKernelGGL((k_project_bwd<0, true>), dim3((total + 255) / 256), dim3(256), 0, st, total,
                           p.d.num_gaussians, p.d.num_cameras, (uint32_t)p.L.dup_capacity, cams, pos, scale, quat,
                           depth_key, order, dup_off, tile_count, row_sums, g_pos, g_scale, g_quat, g_color, g_opacity,
                           g_phase, nullptr, nullptr, 1, 1u, rec);
        FGS_LAUNCH_CHECK("k_project_bwd");
        return FGS_OK;
    }
    hipLaunchKernelGGL(k_project_bwd<0>, dim3(grid), dim3(256), 0, st, total, p.d.num_gaussians,
                       p.d.num_cameras, (uint32_t)p.L.dup_capacity, cams, pos, scale, quat, depth_key, order, dup_off,
                       tile_count, grad_rows, g_pos, g_scale, g_quat, g_color, g_opacity, g_phase, nullptr, nullptr,
                       1, 4u, rec);  // phase path: four rows per duplicate (rec: row geometry); slots 0 / 1 = first moments
    FGS_LAUNCH_CHECK("k_project_bwd");
    return FGS_OK;
}

int fgs_launch_asm_project_bwd(const FgsPlan &p, const float *cams, const float *pos, const float *scale,
                               const float *quat, const float *color, const float *phase, int phase_channels,
                               const char *saved, const float *grad_rows, float *g_pos, float *g_scale,
                               float *g_quat, float *g_color, float *g_opacity, float *g_phase, hipStream_t st,
                               bool wave_rows) {
    const int32_t total = p.d.batch * p.d.num_gaussians;
    const int grid = (int)(((size_t)total * 4 + 255) / 256);
    if (wave_rows) {
        hipLaunchKernelGGL(k_project_bwd<2>, dim3(grid), dim3(256), 0, st, total, p.d.num_gaussians,
                           p.d.num_cameras, (uint32_t)p.L.dup_capacity, cams, pos, scale, quat,
                           reinterpret_cast<const uint32_t *>(saved + p.L.depth_key),
                           reinterpret_cast<const uint32_t *>(saved + p.L.order),
                           reinterpret_cast<const uint32_t *>(saved + p.L.dup_off),
                           reinterpret_cast<const uint32_t *>(saved + p.L.tile_count), grad_rows, g_pos, g_scale,
                           g_quat, g_color, g_opacity, g_phase, color, phase, phase_channels, 1u,
                           reinterpret_cast<const float *>(saved + p.L.rec));
        FGS_LAUNCH_CHECK("k_wave_project_bwd");
        return FGS_OK;
    }
    hipLaunchKernelGGL(k_project_bwd<1>, dim3(grid), dim3(256), 0, st, total, p.d.num_gaussians,
                       p.d.num_cameras, (uint32_t)p.L.dup_capacity, cams, pos, scale, quat,
                       reinterpret_cast<const uint32_t *>(saved + p.L.depth_key),
                       reinterpret_cast<const uint32_t *>(saved + p.L.order),
                       reinterpret_cast<const uint32_t *>(saved + p.L.dup_off),
                       reinterpret_cast<const uint32_t *>(saved + p.L.tile_count), grad_rows, g_pos, g_scale,
                       g_quat, g_color, g_opacity, g_phase, color, phase, phase_channels, 1u,
                       reinterpret_cast<const float *>(saved + p.L.rec));
    FGS_LAUNCH_CHECK("k_asm_project_bwd");
    return FGS_OK;
}
