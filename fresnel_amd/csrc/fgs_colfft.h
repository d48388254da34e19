// In-LDS radix-4 FFT down the columns of a tile, shared by the column-fused transforms of the angular-spectrum renderer
// (fgs_asm.hip) and the plain column pass of fgs_fft2_exec (fgs_fft.hip).
#pragma once
#include <hip/hip_runtime.h>

// (FMAs written out: the butterflies must cost the same whatever the including unit's -ffp-contract setting -- fgs_asm.hip has it off)
__device__ __forceinline__ float2 fgs_cmul(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -(a.x * b.y))); }  // a conj(b)

// 8-point butterflies in registers = three radix-2 stages (block sizes M, M/2, M/4) on the points i + k M/8, k = 0 ... 7, in place
// (a[k] <- the value of point i + k M/8 after the pass).  t1 = w_M^i, t2 = w_M^(2i), t4 = w_M^(4i); UNIT: i = 0, all three are 1
// (the M = 8 pass) and the twelve complex multiplications by them are left out.
// w_M^(i + k M/8) = w_M^i w_8^k, w_(M/2)^(i + k' M/8) = w_M^(2i) w_4^k', w_(M/4)^i = w_M^(4i).
template <bool UNIT>
__device__ __forceinline__ void oct_dif(float2 (&a)[8], float2 t1, float2 t2, float2 t4) {  // forward, decimation in frequency
    constexpr float S = 0.70710678118654752440f;
    float2 s[4], d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[k] = cadd(a[k], a[k + 4]); d[k] = csub(a[k], a[k + 4]); }
    d[1] = make_float2(S * (d[1].x + d[1].y), S * (d[1].y - d[1].x));   // w_8
    d[2] = make_float2(d[2].y, -d[2].x);                                 // w_8^2 = -i
    d[3] = make_float2(S * (d[3].y - d[3].x), -S * (d[3].x + d[3].y));  // w_8^3
    if (!UNIT) {
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = fgs_cmul(d[k], t1);
    }
    float2 b[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float2 *v = h ? d : s;
        const float2 e02 = csub(v[0], v[2]), e13 = csub(v[1], v[3]), r13 = make_float2(e13.y, -e13.x);  // w_4 = -i
        b[4 * h + 0] = cadd(v[0], v[2]);
        b[4 * h + 1] = cadd(v[1], v[3]);
        b[4 * h + 2] = UNIT ? e02 : fgs_cmul(e02, t2);
        b[4 * h + 3] = UNIT ? r13 : fgs_cmul(r13, t2);
    }
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const float2 df = csub(b[2 * qd], b[2 * qd + 1]);
        a[2 * qd] = cadd(b[2 * qd], b[2 * qd + 1]);
        a[2 * qd + 1] = UNIT ? df : fgs_cmul(df, t4);
    }
}

template <bool UNIT>
__device__ __forceinline__ void oct_dit(float2 (&a)[8], float2 t1, float2 t2, float2 t4) {  // inverse, decimation in time
    constexpr float S = 0.70710678118654752440f;
    float2 b[8];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const float2 v = UNIT ? a[2 * qd + 1] : cmulc(a[2 * qd + 1], t4);
        b[2 * qd] = cadd(a[2 * qd], v);
        b[2 * qd + 1] = csub(a[2 * qd], v);
    }
    float2 s[4], d[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float2 *o = h ? d : s;
        const float2 v2 = UNIT ? b[4 * h + 2] : cmulc(b[4 * h + 2], t2), v3t = UNIT ? b[4 * h + 3] : cmulc(b[4 * h + 3], t2);
        const float2 v3 = make_float2(-v3t.y, v3t.x);                    // conj(w_4) = +i
        o[0] = cadd(b[4 * h + 0], v2); o[2] = csub(b[4 * h + 0], v2);
        o[1] = cadd(b[4 * h + 1], v3); o[3] = csub(b[4 * h + 1], v3);
    }
    if (!UNIT) {
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = cmulc(d[k], t1);
    }
    d[1] = make_float2(S * (d[1].x - d[1].y), S * (d[1].x + d[1].y));   // conj(w_8)
    d[2] = make_float2(-d[2].y, d[2].x);                                 // +i
    d[3] = make_float2(-S * (d[3].x + d[3].y), S * (d[3].x - d[3].y));  // conj(w_8^3)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = cadd(s[k], d[k]);
        a[k + 4] = csub(s[k], d[k]);
    }
}

// Block barrier that orders LDS traffic only.  __syncthreads() is a fence as well: with buffer loads into LDS in flight (the next
// tile of k_colfft_fwd) the compiler puts s_waitcnt vmcnt(0) in front of it and the prefetch stops being one.
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// In-LDS FFT of TC independent lines of N = 2^LOGN points, element (point r, line col) = X(r, col) -- an accessor, so that
// callers choose the LDS layout (plain x[N][TC] for the column transforms); all NT threads of the block; tw = w_N^n, n < N/2, in LDS.
// INV = false: forward (e^-), decimation in frequency, natural order in, bit-reversed order out.  INV = true: inverse
// (e^+), decimation in time, bit-reversed order in, natural order out.  Two radix-2 stages per pass over the data
// (block sizes M and M/2), one more radix-2 pass when LOGN is odd.  (Consecutive lanes = the TC columns of a row, then
// the next butterfly: a 32-lane group reads two whole rows, conflict-free except in the last pass.)
template <int LOGN, int TC, int NT, bool INV, bool OUTER = false, bool INNER = false, bool LDSB = false, class Acc>
__device__ __forceinline__ void lds_fft_core(Acc X, const float2 *tw) {
    constexpr int N = 1 << LOGN;
    auto sync = [] { if (LDSB) lds_only_barrier(); else __syncthreads(); };  // LDSB: see lds_only_barrier
    auto pair_pass = [&]() {  // block size 2: point 2k = a + b, point 2k + 1 = a - b
#pragma unroll 1
        for (int idx = threadIdx.x; idx < (N / 2) * TC; idx += NT) {
            const int col = idx % TC, k = idx / TC;
            const float2 a = X(2 * k, col), b = X(2 * k + 1, col);
            X(2 * k, col) = cadd(a, b); X(2 * k + 1, col) = csub(a, b);
        }
        sync();
    };
    auto quad_pass = [&](int M) {  // block sizes M and M / 2 on the points i, i + M/4, i + M/2, i + 3M/4
        const int Q = M / 4, step = N / M;
#pragma unroll 1
        for (int idx = threadIdx.x; idx < (N / 4) * TC; idx += NT) {
            const int col = idx % TC, q = idx / TC;
            const int i = q % Q, p0 = (q / Q) * M + i, p1 = p0 + Q, p2 = p0 + 2 * Q, p3 = p0 + 3 * Q;
            const float2 w1 = tw[i * step], w2 = tw[2 * i * step];  // w_M^i, w_M^(2i) = w_(M/2)^i
            const float2 a0 = X(p0, col), a1 = X(p1, col), a2 = X(p2, col), a3 = X(p3, col);
            if (!INV) {
                const float2 s02 = cadd(a0, a2), s13 = cadd(a1, a3), d02 = csub(a0, a2), d13 = csub(a1, a3);
                const float2 u2 = fgs_cmul(d02, w1), u3 = fgs_cmul(make_float2(d13.y, -d13.x), w1);  // w_M^(i + M/4) = -i w_M^i
                X(p0, col) = cadd(s02, s13);
                X(p1, col) = fgs_cmul(csub(s02, s13), w2);
                X(p2, col) = cadd(u2, u3);
                X(p3, col) = fgs_cmul(csub(u2, u3), w2);
            } else {
                const float2 t1 = cmulc(a1, w2), t3 = cmulc(a3, w2);
                const float2 r0 = cadd(a0, t1), r1 = csub(a0, t1), r2 = cadd(a2, t3), r3 = csub(a2, t3);
                const float2 v2 = cmulc(r2, w1), v3t = cmulc(r3, w1);
                const float2 v3 = make_float2(-v3t.y, v3t.x);  // conj(-i w_M^i) = +i conj(w_M^i)
                X(p0, col) = cadd(r0, v2);
                X(p2, col) = csub(r0, v2);
                X(p1, col) = cadd(r1, v3);
                X(p3, col) = csub(r1, v3);
            }
        }
        sync();
    };
    // THREE radix-2 stages per pass (block sizes M, M/2, M/4) on the eight points i + k M/8: with NT = N TC / 8 threads
    // every thread does exactly one 8-point butterfly per pass, and a 512-point column needs three passes (and barriers)
    // instead of five.  w_M^(i + k M/8) = w_M^i w_8^k, w_(M/2)^(i + k' M/8) = w_M^(2i) w_4^k', w_(M/4)^i = w_M^(4i).
    auto oct_pass = [&](int M) {
        const int E = M / 8, step = N / M;
#pragma unroll 1
        for (int idx = threadIdx.x; idx < (N / 8) * TC; idx += NT) {
            const int col = idx % TC, q = idx / TC;
            const int i = q % E, b0 = (q / E) * M + i;
            const float2 t1 = tw[i * step], t2 = tw[2 * i * step], t4 = tw[4 * i * step];
            float2 a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = X(b0 + k * E, col);
            if (!INV) oct_dif<false>(a, t1, t2, t4); else oct_dit<false>(a, t1, t2, t4);
#pragma unroll
            for (int k = 0; k < 8; ++k) X(b0 + k * E, col) = a[k];
        }
        sync();
    };
    constexpr int NOCT = LOGN / 3, REM = LOGN - 3 * NOCT;  // radix-8 passes from the top, then one radix-4 / radix-2 pass
    // OUTER: the block-size-N pass (first forward, last inverse) is the caller's, in registers; INNER: so is the block-size-8 pass
    // (last forward, first inverse; LOGN a multiple of three, see lds_fft_inner_in_registers)
    constexpr int O_LO = OUTER ? 1 : 0, O_HI = (INNER && REM == 0) ? NOCT - 1 : NOCT;
    if (!INV) {
#pragma unroll
        for (int o = O_LO; o < O_HI; ++o) oct_pass(1 << (LOGN - 3 * o));
        if (REM == 2) quad_pass(4);
        if (REM == 1) pair_pass();
    } else {
        if (REM == 2) quad_pass(4);
        if (REM == 1) pair_pass();
#pragma unroll
        for (int o = O_HI - 1; o >= O_LO; --o) oct_pass(1 << (LOGN - 3 * o));
    }
}

// whether the block-size-8 pass can be the caller's as well (it is an oct pass at all, and not the same pass as the outer one)
template <int LOGN>
constexpr bool lds_fft_inner_in_registers() { return LOGN % 3 == 0 && LOGN >= 6; }

// The same on a plain x[N][TC] tile (column-fused transforms: TC contiguous columns per row, conflict-free by construction).
template <int LOGN, int TC, int NT, bool INV, bool OUTER = false, bool INNER = false, bool LDSB = false>
__device__ __forceinline__ void lds_fft_columns(float2 (*x)[TC], const float2 *tw) {
    lds_fft_core<LOGN, TC, NT, INV, OUTER, INNER, LDSB>([x](int r, int col) -> float2 & { return x[r][col]; }, tw);
}

template <int LOGN>
__device__ __forceinline__ int bitrev(int r) { return (int)(__brev((unsigned)r) >> (32 - LOGN)); }

