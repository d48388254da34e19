// Wave64 cross-lane reductions and list-staging helpers for gfx950, shared by the composite and ASM kernels.
#pragma once

// Staging-time (per lane, parallel over the chunk) decode of a record's bbox against a 16x16 tile
// at (X0, Y0): bit s = sub-tile s (8x8, s = 2*row + col) intersects the bbox [x0,x1) x [y0,y1).
__device__ __forceinline__ uint32_t subtile_mask(uint32_t X0, uint32_t Y0, uint32_t x0, uint32_t x1, uint32_t y0,
                                                 uint32_t y1) {
    const uint32_t cx0 = (x1 > X0 && x0 < X0 + 8u) ? 1u : 0u, cx1 = (x1 > X0 + 8u && x0 < X0 + 16u) ? 1u : 0u;
    const uint32_t ry0 = (y1 > Y0 && y0 < Y0 + 8u) ? 1u : 0u, ry1 = (y1 > Y0 + 8u && y0 < Y0 + 16u) ? 1u : 0u;
    return (cx0 & ry0) | ((cx1 & ry0) << 1) | ((cx0 & ry1) << 2) | ((cx1 & ry1) << 3);
}

// Staging-time decode (per lane, parallel over the chunk) of a record against the 16x16 tile at (X0, Y0), for the
// non-phase composite kernels:
//   flags  bits 0-3: sub-tile s (8x8, s = 2*row + col) intersects the bbox -- all clear when the opacity is
//                    negative (alpha clamps to 0 with zero gradient, DR:646: the record contributes nothing);
//          bit 4:    opacity <= 0.98 and `conic_ok` (the caller's check that the quadratic form is positive
//                    definite with a margin), so alpha = min(G op, 0.99) cannot bind: G <= 1 up to rounding.
//                    For a regularised inverse covariance that came out indefinite in fp32 (needles, edge-on
//                    discs) G can exceed 1.0102 and the clamp of DR:647 does bind -- those keep the clamped path;
//          bit 5:    the bbox covers the whole tile (no per-pixel membership test needed)
//   bits   bit i (i < 16): pixel column X0 + i lies in [x0, x1);  bit 16 + i: pixel row Y0 + i lies in [y0, y1).
// In the list loop a lane turns its column / row bit into an all-ones / zero mask with one v_bfe_i32.
__device__ __forceinline__ void stage_decode(uint32_t X0, uint32_t Y0, uint32_t bbx, uint32_t bby, float op,
                                             uint32_t &flags, uint32_t &bits, bool conic_ok = true) {
    const uint32_t x0 = bbx & 0xFFFFu, x1 = bbx >> 16, y0 = bby & 0xFFFFu, y1 = bby >> 16;
    flags = (op >= 0.0f ? subtile_mask(X0, Y0, x0, x1, y0, y1) : 0u) | ((op <= 0.98f && conic_ok) ? 16u : 0u) |
            ((x0 <= X0 && x1 >= X0 + 16u && y0 <= Y0 && y1 >= Y0 + 16u) ? 32u : 0u);
    const int lx0 = max((int)x0 - (int)X0, 0), lx1 = min((int)x1 - (int)X0, 16);
    const int ly0 = max((int)y0 - (int)Y0, 0), ly1 = min((int)y1 - (int)Y0, 16);
    const uint32_t xm = lx1 > lx0 ? ((1u << (lx1 - lx0)) - 1u) << lx0 : 0u;
    const uint32_t ym = ly1 > ly0 ? ((1u << (ly1 - ly0)) - 1u) << ly0 : 0u;
    bits = xm | (ym << 16);
}

// The same for a tile of NSX x 2 sub-tiles (16 or 32 pixels wide), depth-split forward and blend backward:
//   flags  bits 0 .. 2 NSX - 1: sub-tile s = row * NSX + col touched (none when the opacity is negative);
//          bit 8: the alpha clamp cannot bind (opacity <= 0.98 and `conic_ok`); bits 16-31: pixel row Y0 + i in [y0, y1);
//   cbits  bit i (i < 8 NSX): pixel column X0 + i lies in [x0, x1).
template <int NSX>
__device__ __forceinline__ void stage_decode_w(uint32_t X0, uint32_t Y0, uint32_t bbx, uint32_t bby, float op,
                                               uint32_t &flags, uint32_t &cbits, bool conic_ok = true) {
    constexpr int TW = 8 * NSX;
    const int x0 = (int)(bbx & 0xFFFFu), x1 = (int)(bbx >> 16), y0 = (int)(bby & 0xFFFFu), y1 = (int)(bby >> 16);
    const int lx0 = max(x0 - (int)X0, 0), lx1 = min(x1 - (int)X0, TW);
    const int ly0 = max(y0 - (int)Y0, 0), ly1 = min(y1 - (int)Y0, 16);
    cbits = lx1 > lx0 ? (uint32_t)(((1ull << (lx1 - lx0)) - 1ull) << lx0) : 0u;
    const uint32_t ym = ly1 > ly0 ? ((1u << (ly1 - ly0)) - 1u) << ly0 : 0u;
    uint32_t touched = 0;
    if (op >= 0.0f) {
#pragma unroll
        for (int c = 0; c < NSX; ++c) {
            const uint32_t cm = (cbits >> (8 * c)) & 0xFFu;
            if (cm && (ym & 0xFFu)) touched |= 1u << c;
            if (cm && (ym >> 8)) touched |= 1u << (NSX + c);
        }
    }
    flags = touched | ((op <= 0.98f && conic_ok) ? 256u : 0u) | (ym << 16);
}

// (x < lim) ? v : 0.  The compare and the select are kept ADJACENT in one asm block: a v_cndmask reading VCC
// straight after the v_cmp that wrote it issues in ~2.6 cycles on gfx950, any other VCC-reading v_cndmask in
// 14-23 (scratch/ubench/valu3.hip, valu4.hip).
__device__ __forceinline__ float select_lt(float x, float lim, float v) {
    float o;
    asm("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, 0, %3, vcc" : "=v"(o) : "v"(x), "v"(lim), "v"(v) : "vcc");
    return o;
}

// (a == b) ? v : 0, compare and select adjacent (see select_lt)
__device__ __forceinline__ float select_eq(float a, float b, float v) {
    float o;
    asm("v_cmp_eq_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, 0, %3, vcc" : "=v"(o) : "v"(a), "v"(b), "v"(v) : "vcc");
    return o;
}
// x >= lim ? (a, b) : (c, 0): ONE compare, both selects right behind it
__device__ __forceinline__ void select2_ge(float x, float lim, float a, float b, float c, float &o1, float &o2) {
    asm("v_cmp_ge_f32 vcc, %2, %3\n\tv_cndmask_b32 %0, %6, %4, vcc\n\tv_cndmask_b32 %1, 0, %5, vcc"
        : "=&v"(o1), "=&v"(o2) : "v"(x), "v"(lim), "v"(a), "v"(b), "v"(c) : "vcc");
}

// Quad sum of one value: lanes with (lane & 3) == 3 end up with the sum over their quad.
__device__ __forceinline__ void quad_sum1(float &a) {
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n" : "+v"(a));
}

// Sum NV per-lane values over the 64 lanes of a wave through LDS, transposed (NV <= 16): every lane parks its NV
// partial sums; lane 4 k + p then adds the 16 partials of part p of value k (four ds_read_b128, 15 adds) and two DPP
// steps fold the four parts -- ~NV LDS stores + 15 plain adds + 2 DPP adds per call, against NV x 6 DPP adds (4.3 issue
// cycles each on gfx950) for a full DPP tree.  One wave's LDS instructions execute in order, so no barrier is needed;
// the scratch is private to the calling wave.
// ---- parking with ds_write_addtid_b32 ----------------------------------------------------------------------------
// Layout: value k at dword 68 k, lane l's partial at 68 k + l (ds_write_addtid_b32: address = M0 + offset + 4 lane,
// no address VGPR, 2 store-path cycles per instruction instead of 4-6, MI355X_MICROARCH.md "LDS").  Lane 4 k + p then
// reads the 16 partials [16 p, 16 p + 16) of value k with four ds_read_b128; with the 68-dword pitch the 16 lanes of
// every ds_read_b128 group touch 16 different 4-bank slots (banks 4 k + 16 p + m mod 64), so both sides are
// conflict-free (the [80]-pitch layout above cost 29 % of the LDS cycles in conflicts).  M0 is saved and restored.
typedef __attribute__((address_space(3))) float fgs_lds_float;
#define FGS_RED_PITCH 68
__device__ __forceinline__ void addtid_park10(uint32_t lds_base, const float (&v)[10]) {
    uint32_t m0_save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %11\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %1 offset:0\n\tds_write_addtid_b32 %2 offset:272\n\t"
                 "ds_write_addtid_b32 %3 offset:544\n\tds_write_addtid_b32 %4 offset:816\n\t"
                 "ds_write_addtid_b32 %5 offset:1088\n\tds_write_addtid_b32 %6 offset:1360\n\t"
                 "ds_write_addtid_b32 %7 offset:1632\n\tds_write_addtid_b32 %8 offset:1904\n\t"
                 "ds_write_addtid_b32 %9 offset:2176\n\tds_write_addtid_b32 %10 offset:2448\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(m0_save)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
                   "v"(v[9]), "s"(lds_base)
                 : "memory");
}

__device__ __forceinline__ void addtid_park11(uint32_t lds_base, const float (&v)[11]) {
    uint32_t m0_save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %12\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %1 offset:0\n\tds_write_addtid_b32 %2 offset:272\n\t"
                 "ds_write_addtid_b32 %3 offset:544\n\tds_write_addtid_b32 %4 offset:816\n\t"
                 "ds_write_addtid_b32 %5 offset:1088\n\tds_write_addtid_b32 %6 offset:1360\n\t"
                 "ds_write_addtid_b32 %7 offset:1632\n\tds_write_addtid_b32 %8 offset:1904\n\t"
                 "ds_write_addtid_b32 %9 offset:2176\n\tds_write_addtid_b32 %10 offset:2448\n\t"
                 "ds_write_addtid_b32 %11 offset:2720\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(m0_save)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
                   "v"(v[9]), "v"(v[10]), "s"(lds_base)
                 : "memory");
}

// twelve / thirteen values (ASM / wave splat backward).  ONE asm block per variant: M0 must not be live across
// compiler-scheduled code (every DS instruction reads it).
#define FGS_PARK12_BODY                                                                              \
    "ds_write_addtid_b32 %1 offset:0\n\tds_write_addtid_b32 %2 offset:272\n\t"                       \
    "ds_write_addtid_b32 %3 offset:544\n\tds_write_addtid_b32 %4 offset:816\n\t"                     \
    "ds_write_addtid_b32 %5 offset:1088\n\tds_write_addtid_b32 %6 offset:1360\n\t"                   \
    "ds_write_addtid_b32 %7 offset:1632\n\tds_write_addtid_b32 %8 offset:1904\n\t"                   \
    "ds_write_addtid_b32 %9 offset:2176\n\tds_write_addtid_b32 %10 offset:2448\n\t"                  \
    "ds_write_addtid_b32 %11 offset:2720\n\tds_write_addtid_b32 %12 offset:2992\n\t"
__device__ __forceinline__ void addtid_park12(uint32_t lds_base, const float (&v)[12]) {
    uint32_t m0_save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %13\n\ts_nop 0\n\t" FGS_PARK12_BODY "s_mov_b32 m0, %0"
                 : "=&s"(m0_save)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
                   "v"(v[9]), "v"(v[10]), "v"(v[11]), "s"(lds_base)
                 : "memory");
}
__device__ __forceinline__ void addtid_park13(uint32_t lds_base, const float (&v)[13]) {
    uint32_t m0_save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %14\n\ts_nop 0\n\t" FGS_PARK12_BODY
                 "ds_write_addtid_b32 %13 offset:3264\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_save)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
                   "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "s"(lds_base)
                 : "memory");
}
#undef FGS_PARK12_BODY

// Sum ten per-lane values over the wave; returns, in lanes with (lane & 3) == 3 and lane < 40, the total of value
// lane >> 2 (other lanes: junk).  `red` = FGS_RED_PITCH * 10 floats of LDS private to the calling wave, 16-B aligned.
__device__ __forceinline__ float wave_sum_addtid_finish(const float *red, uint32_t lane, uint32_t nv) {
    __builtin_amdgcn_wave_barrier();
    float tot = 0.0f;
    if (lane < 4u * nv) {
        const float4 *src = reinterpret_cast<const float4 *>(red + FGS_RED_PITCH * (lane >> 2) + 16u * (lane & 3u));
        const float4 s0 = src[0], s1 = src[1], s2 = src[2], s3 = src[3];
        tot = ((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w)) +
              (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
        quad_sum1(tot);
    }
    __builtin_amdgcn_wave_barrier();
    return tot;
}
__device__ __forceinline__ float wave_sum10_addtid(float *red, const float (&v)[10], uint32_t lane) {
    addtid_park10((uint32_t)(uintptr_t)(fgs_lds_float *)red, v);
    return wave_sum_addtid_finish(red, lane, 10u);
}
// NV = 12 | 13 values (ASM / wave splat backward); `red` = FGS_RED_PITCH * NV floats
template <int NV>
__device__ __forceinline__ float wave_sum_addtid(float *red, const float (&v)[NV], uint32_t lane) {
    static_assert(NV == 12 || NV == 13, "park helpers written for 12 / 13 values");
    if constexpr (NV == 12) addtid_park12((uint32_t)(uintptr_t)(fgs_lds_float *)red, v);
    else addtid_park13((uint32_t)(uintptr_t)(fgs_lds_float *)red, v);
    return wave_sum_addtid_finish(red, lane, (uint32_t)NV);
}
// eleven values (phase backward); `red` = FGS_RED_PITCH * 11 floats
__device__ __forceinline__ float wave_sum11_addtid(float *red, const float (&v)[11], uint32_t lane) {
    addtid_park11((uint32_t)(uintptr_t)(fgs_lds_float *)red, v);
    return wave_sum_addtid_finish(red, lane, 11u);
}
