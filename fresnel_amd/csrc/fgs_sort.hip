// Stable LSD radix sort (8-bit digits) of uint32 (key, payload) pairs -- the build's own
// design; the reference has no per-tile sort and its Vulkan GPURadixSort
// (src/core/compute/radix_sort.cpp:172-242) scatters through atomicAdd and is NOT stable.
//
// Used twice per forward:
//   (1) depth order: B independent segments of N keys = order-preserving depth bits
//       (replaces torch.argsort(depths), DR:527, with the canonical stable order of
//       depth ascending, ties by original index);
//   (2) tile binning: ONE segment of D duplicates (D known only on the device), key =
//       image*T + tile.  Duplicates are emitted in depth order, so stability alone keeps
//       every tile's list depth-sorted and only ceil(log2(B*T)/8) passes are needed.
//
// Per pass: upsweep (per-block digit histogram) -> per-digit wave scan -> downsweep (each
// block re-walks its contiguous range in rounds of 256 keys; a key's rank inside the round
// comes from wave64 ballots: 8 ballots give the mask of lanes holding the same digit,
// popcount below the lane gives the stable rank -- no atomics, no order dependence).
// HBM-bound: 8 B read in upsweep+downsweep and 8 B written per element per pass.
//
// KEY COMPRESSION (depth sort only, round 4; BASELINE config 4's "depth-zone sort keys"; selected by FgsDims.sort_mode = 1).  Only the key bits that VARY over a
// segment's visible Gaussians can change their order.  k_project leaves, per block, the OR and the AND of the visible keys;
// every block of the sort folds its segment's records into the mask m = OR ^ AND and sorts by the COMPRESSED key
//     pext(key, m)  for a visible Gaussian,    1 << popcount(m)  for a culled one (key 0xFFFFFFFF: behind all visible ones),
// 8 bits per pass -- order-preserving among the visible keys (the dropped bits are equal in all of them), culled ones last
// and among themselves in index order (stable), i.e. the SAME permutation as sorting the full keys.  A segment needs
// ceil((popcount(m) + [any culled]) / 8) passes; the blocks of the later passes leave at once, and the last LIVE pass of a
// segment writes the payload to its final place.  Zone-snapped depths (8 zones: 3 varying bits) sort in ONE pass instead of
// four; ordinary depths in (0.5, 4) vary in ~25 bits and keep their four.  All 8 launches still happen (how many passes a
// segment needs is known on the device only) -- a dead pass costs a launch of blocks that read ~100 words and exit -- and
// every live kernel pays ~1-1.7 us for folding the records.  Measured (profiles/r04_ab_sort_key_compression.txt): config 4
// depth sort 38 -> 27 us, but config 2 39 -> 52 and config 3 53 -> 68 (with the record loads overlapped with the key loads:
// 37 -> 33, 39 -> 47, 53 -> 60).  Hence a MODE the caller selects when it knows its depths are quantised (the training
// harness with --use_fresnel_zones, bench.py --workload config4), never a default: any data sorts correctly in either mode.
#include <type_traits>
#include "fgs_internal.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;

struct SegInfo {
    uint32_t begin, end;  // this block's element range (absolute indices)
};

__device__ __forceinline__ SegInfo block_range(uint32_t seg_len, const uint32_t *seg_len_dev,
                                               uint32_t seg_capacity, uint32_t seg_stride) {
    uint32_t len = seg_len_dev ? *seg_len_dev : seg_len;
    if (len > seg_capacity) len = seg_capacity;
    const uint32_t bps = gridDim.x;
    uint32_t per = (len + bps - 1) / bps;
    per = (per + RS_THREADS * 4 - 1) / (RS_THREADS * 4) * (RS_THREADS * 4);
    const uint32_t seg0 = blockIdx.y * seg_stride;
    uint32_t b = blockIdx.x * per, e = b + per;
    if (b > len) b = len;
    if (e > len) e = len;
    return {seg0 + b, seg0 + e};
}


// Stable rank of a lane among the wave's lanes that hold the same BITS-bit digit (valid lanes only): `rank` = how many of them sit in
// lower lanes, `count` = how many there are.  One ballot per digit bit; the running match mask is kept as two 32-bit halves per lane
// and narrowed by m &= ~(ballot ^ sext(bit)) -- four VALU instructions per bit (v_bfe_i32, v_cmp, two v_bitop3_b32).
// Written as `m &= bset ? bm : ~bm` on a 64-bit mask the compiler spends eleven per bit on it (compare, select, compare again for
// the ballot, 64-bit select built from a cndmask and a 64-bit add, two xor, two and: 110 instructions per 64 keys), and the
// ranking is what a pass of the in-LDS sort costs.
template <int BITS>
__device__ __forceinline__ void match_rank(uint32_t digit, bool valid, uint32_t &rank, uint32_t &count) {
    const unsigned long long bv = __ballot(valid);
    uint32_t mlo = (uint32_t)bv, mhi = (uint32_t)(bv >> 32);
#pragma unroll
    for (int bit = 0; bit < BITS; ++bit) {
        const uint32_t sel = (uint32_t)__builtin_amdgcn_sbfe((int)digit, bit, 1);  // all ones where the bit is set
        const unsigned long long bm = __ballot((int)sel < 0);
        // m & ~(bm ^ sel) as ONE three-input bit operation per half (truth table 0x90: m set and bm == sel); spelled with & ~ ^ the
        // compiler collects the mismatches with shifts, xors and or3s instead: six instructions per bit, this is four
        mlo = __builtin_amdgcn_bitop3_b32(mlo, (uint32_t)bm, sel, 0x90);
        mhi = __builtin_amdgcn_bitop3_b32(mhi, (uint32_t)(bm >> 32), sel, 0x90);
    }
    rank = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
    count = (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi);
}

// ---- key compression (see the header) ------------------------------------------------------------------------------------
struct KeyPlan {
    uint32_t live;      // passes this segment needs (>= 1)
    uint32_t sh[8];     // source bit of digit bit i of the current pass
    uint32_t vmask;     // digit bits that exist in this pass
    uint32_t cull;      // digit of a culled key in this pass
};

// every thread of the block returns the plan of (segment, pass); `kb` = 12 + 12 words of LDS
__device__ __forceinline__ KeyPlan key_plan(const uint32_t *__restrict__ bits, uint32_t nrec, uint32_t seg, uint32_t pass,
                                            uint32_t *kb) {
    uint32_t vor = 0u, vand = 0xFFFFFFFFu, fl = 0u;
    for (uint32_t i = threadIdx.x; i < nrec; i += RS_THREADS) {
        const uint4 r = reinterpret_cast<const uint4 *>(bits)[(size_t)seg * nrec + i];
        vor |= r.x; vand &= r.y; fl |= r.z;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { vor |= __shfl_xor(vor, o, 64); vand &= __shfl_xor(vand, o, 64); fl |= __shfl_xor(fl, o, 64); }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) { kb[3 * wave] = vor; kb[3 * wave + 1] = vand; kb[3 * wave + 2] = fl; }
    __syncthreads();
    if (threadIdx.x < 8) {
        uint32_t o = 0u, a = 0xFFFFFFFFu, f = 0u;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) { o |= kb[3 * w]; a &= kb[3 * w + 1]; f |= kb[3 * w + 2]; }
        const uint32_t m = (f & 2u) ? (o ^ a) : 0u;  // no visible key: nothing varies
        const uint32_t nbits = (uint32_t)__popc(m), cull = f & 1u;
        const uint32_t idx = 8u * pass + threadIdx.x;
        uint32_t mm = m;
        for (uint32_t j = 0; j < idx && mm; ++j) mm &= mm - 1u;
        kb[12 + threadIdx.x] = (idx < nbits) ? (uint32_t)__ffs((int)mm) - 1u : 32u;
        if (threadIdx.x == 0) {
            const uint32_t need = (nbits + cull + 7u) / 8u;
            kb[20] = need < 1u ? 1u : (need > 4u ? 4u : need);  // (33 bits -- every key bit varies and some keys are culled -- is the full key: key_digit)
            kb[21] = (cull && nbits >= 8u * pass && nbits < 8u * pass + 8u) ? 1u << (nbits - 8u * pass) : 0u;
        }
    }
    __syncthreads();
    KeyPlan kp;
    kp.live = kb[20];
    kp.cull = kb[21];
    kp.vmask = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t p = kb[12 + i];
        kp.sh[i] = p & 31u;
        kp.vmask |= (p < 32u ? 1u : 0u) << i;
    }
    return kp;
}

// (a segment whose keys vary in more than 24 bits needs four passes either way: the key's own bytes then, same permutation)
__device__ __forceinline__ uint32_t key_digit(const KeyPlan &kp, uint32_t key, uint32_t pass) {
    if (kp.live >= 4u) return (key >> (8u * pass)) & 0xFFu;  // block-uniform
    uint32_t d = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) d |= ((key >> kp.sh[i]) & 1u) << i;
    return key == 0xFFFFFFFFu ? kp.cull : (d & kp.vmask);
}

template <bool COMPRESSED>
__global__ __launch_bounds__(RS_THREADS) void k_radix_upsweep(
    const uint32_t *__restrict__ keys, uint32_t seg_len, const uint32_t *__restrict__ seg_len_dev,
    uint32_t seg_capacity, uint32_t seg_stride, uint32_t shift, uint32_t dmask, uint32_t *__restrict__ hist,
    const uint32_t *__restrict__ key_bits, uint32_t key_recs, uint32_t pass) {
    __shared__ uint32_t h[256];
    __shared__ uint32_t kb[24];
    KeyPlan kp;
    if (COMPRESSED) {
        kp = key_plan(key_bits, key_recs, blockIdx.y, pass, kb);
        if (pass >= kp.live) return;  // this segment is sorted already
    }
    h[threadIdx.x] = 0;
    __syncthreads();
    const SegInfo r = block_range(seg_len, seg_len_dev, seg_capacity, seg_stride);
    for (uint32_t i = r.begin + threadIdx.x; i < r.end; i += 4 * RS_THREADS) {  // four loads in flight
        uint32_t k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = i + u * RS_THREADS < r.end ? keys[i + u * RS_THREADS] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * RS_THREADS < r.end) atomicAdd(&h[COMPRESSED ? key_digit(kp, k[u], pass) : (k[u] >> shift) & dmask], 1u);
    }
    __syncthreads();
    // layout: hist[(seg*256 + digit) * bps + blk]
    hist[((size_t)blockIdx.y * 256 + threadIdx.x) * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

// One wave per (segment, digit): exclusive scan along the block axis, digit total to dtot.
// (Key compression never comes here: it requires bps <= 64, the fused scan.  Were it to, the rows of segments whose upsweep
// blocks left early -- pass >= live -- would be unwritten; harmless only because their downsweep blocks leave early too.)
__global__ __launch_bounds__(256) void k_radix_scan(uint32_t *__restrict__ hist, uint32_t *__restrict__ dtot,
                                                    uint32_t bps, uint32_t num_rows) {
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);  // seg*256 + digit
    if (row >= num_rows) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *p = hist + (size_t)row * bps;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < bps; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t v = i < bps ? p[i] : 0u;
        uint32_t s = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(s, o, 64);
            if ((int)lane >= o) s += t;
        }
        if (i < bps) p[i] = carry + s - v;
        carry += __shfl(s, 63, 64);
    }
    if (lane == 0) dtot[row] = carry;
}

template <bool COMPRESSED>
__global__ __launch_bounds__(RS_THREADS) void k_radix_downsweep(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t seg_len,
    const uint32_t *__restrict__ seg_len_dev, uint32_t seg_capacity, uint32_t seg_stride, uint32_t shift,
    uint32_t dmask, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ dtot, uint32_t idx_mod,
    const uint32_t *__restrict__ key_bits, uint32_t key_recs, uint32_t pass, uint32_t *__restrict__ vals_final) {
    // dtot == nullptr: `hist` holds the RAW per-block digit counts and this block forms its own prefix (sum over the
    // blocks before it, total over all of them) -- no k_radix_scan launch; used when a segment has few blocks (the
    // depth sort, every launch of which sits at the launch floor).
    // vals_in == nullptr: the payload is the element's index modulo idx_mod (first pass of the depth sort).
    // (Accumulating the NEXT pass's per-block digit counts here, one global atomic per key at its destination block --
    // which would make the upsweep launches of passes 1-3 unnecessary -- was tried: the atomics cost 19 us per pass,
    // 10 -> 30 us per downsweep at config 3, against 5 us for the upsweep they replace.)
    __shared__ uint32_t run_off[256];
    __shared__ uint32_t wcnt[RS_WAVES][256];
    __shared__ uint32_t wtot[RS_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    __shared__ uint32_t kb[24];
    KeyPlan kp;
    if (COMPRESSED) {
        kp = key_plan(key_bits, key_recs, blockIdx.y, pass, kb);
        if (pass >= kp.live) return;                        // this segment is sorted already
        if (pass + 1u == kp.live) vals_out = vals_final;    // its last live pass: the payload goes to its final place
    }
    const SegInfo r = block_range(seg_len, seg_len_dev, seg_capacity, seg_stride);
    const uint32_t seg0 = blockIdx.y * seg_stride;
    constexpr uint32_t ROUND = RS_THREADS * 4;
    uint32_t key[4], val[4];
    bool valid[4];
    // Rounds of 1024 keys: wave w owns the contiguous 256 keys [base + 256 w, +256) and walks them
    // in four 64-key sub-rounds.  The first round's keys are requested BEFORE the prefix below is formed, so the
    // two memory round trips overlap.
    auto load_round = [&](uint32_t base) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t i = base + wave * 256u + it * 64u + lane;
            valid[it] = i < r.end;
            key[it] = valid[it] ? keys_in[i] : 0u;
            val[it] = valid[it] ? (vals_in ? vals_in[i] : i % idx_mod) : 0u;
        }
    };
    load_round(r.begin);
    // exclusive scan of the 256 digit totals of this segment: wave scans + the four wave totals
    {
        uint32_t t, mine;
        if (dtot) {
            t = dtot[blockIdx.y * 256 + tid];
            mine = hist[((size_t)blockIdx.y * 256 + tid) * gridDim.x + blockIdx.x];
        } else {
            // (all the loads of a batch are issued before the first is used: the plain loop waited for every load in turn,
            // 16 serialised L2 round trips of the 10 us this kernel took at config 3)
            const uint32_t *row = hist + ((size_t)blockIdx.y * 256 + tid) * gridDim.x;
            const uint32_t bps = gridDim.x, me = blockIdx.x;
            t = 0; mine = 0;
            auto sum_vectors = [&](auto nv_tag) {  // rows of whole 16-byte groups (the histogram base is 256-byte aligned)
                constexpr uint32_t NV = decltype(nv_tag)::value;
                const uint4 *row4 = reinterpret_cast<const uint4 *>(row);
                uint4 v[NV];
#pragma unroll
                for (uint32_t k = 0; k < NV; ++k) v[k] = 4u * k < bps ? row4[k] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (uint32_t k = 0; k < NV; ++k) {
                    t += v[k].x + v[k].y + v[k].z + v[k].w;
                    mine += (4u * k < me ? v[k].x : 0u) + (4u * k + 1u < me ? v[k].y : 0u) +
                            (4u * k + 2u < me ? v[k].z : 0u) + (4u * k + 3u < me ? v[k].w : 0u);
                }
            };
            if ((bps & 3u) == 0u && bps <= 16u) {
                sum_vectors(std::integral_constant<uint32_t, 4>{});
            } else if ((bps & 3u) == 0u) {  // bps <= 64
                sum_vectors(std::integral_constant<uint32_t, 16>{});
            } else {
                for (uint32_t k0 = 0; k0 < bps; k0 += 8) {
                    uint32_t c[8];
#pragma unroll
                    for (uint32_t u = 0; u < 8; ++u) c[u] = k0 + u < bps ? row[k0 + u] : 0u;
#pragma unroll
                    for (uint32_t u = 0; u < 8; ++u) {
                        mine += k0 + u < me ? c[u] : 0u;
                        t += c[u];
                    }
                }
            }
        }
        uint32_t s = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(s, o, 64);
            if ((int)lane >= o) s += v;
        }
        if (lane == 63) wtot[wave] = s;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) wcnt[w][tid] = 0;
        __syncthreads();
        uint32_t pre = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) pre += (w < (int)wave) ? wtot[w] : 0u;
        run_off[tid] = seg0 + pre + (s - t) + mine;
    }
    __syncthreads();
    // Ranking against the wave's OWN running digit counters in LDS (a wave's LDS operations execute in program
    // order, so read-count-then-bump needs no barrier); one block barrier per round then turns the four waves'
    // counts into global positions.
    for (uint32_t base = r.begin; base < r.end; base += ROUND) {
        uint32_t lrank[4], dig[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t digit = COMPRESSED ? key_digit(kp, key[it], pass) : (key[it] >> shift) & dmask;
            dig[it] = digit;
            uint32_t rank, count;
            match_rank<8>(digit, valid[it], rank, count);
            const uint32_t prev = valid[it] ? wcnt[wave][digit] : 0u;
            lrank[it] = prev + rank;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (valid[it] && rank == 0) wcnt[wave][digit] = prev + count;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (valid[it]) {
                const uint32_t digit = dig[it];
                uint32_t pre = 0;
#pragma unroll
                for (int w = 0; w < RS_WAVES; ++w) pre += (w < (int)wave) ? wcnt[w][digit] : 0u;
                const uint32_t dst = run_off[digit] + pre + lrank[it];
                keys_out[dst] = key[it];
                vals_out[dst] = val[it];
            }
        }
        __syncthreads();
        {
            uint32_t tot = 0;
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) { tot += wcnt[w][tid]; wcnt[w][tid] = 0; }
            run_off[tid] += tot;
        }
        if (base + ROUND < r.end) load_round(base + ROUND);
        __syncthreads();
    }
}

// ---- FUSED PASS (round 5): one launch per radix pass, no histogram hand-off between blocks -------------------------------------
// The depth sort's launches sit at the launch floor (8 launches of ~4.9 us for 16 x 8192 keys that move 1 MB): what a pass costs is
// its launch boundaries, not its bytes.  Here every block of a segment forms the segment's digit histogram BY ITSELF -- the digits
// in front of its own key range (`pre`) and from its range on (`rest`) -- by reading ALL keys of the segment (<= 64 K keys = 256 KB
// from L2, ONE LDS atomic per key; a wave whose 64 keys share a digit -- quantised depths -- adds once), so the upsweep launch and
// its histogram round trip disappear without any inter-block communication (no flags, no spinning, nothing that can hang: the
// redundant reads are the price, ~1 us per pass at 8 192 keys per segment, ~3 us at 32 768).  Digits are BITS = 11 wide: three
// passes over 32-bit keys instead of four (2048 bins: `pre`, `rest` / running offsets and the four waves' rank counters = 48 KB of
// LDS).  Ranking and scatter are k_radix_downsweep's (wave64 ballots, stable, no atomics in the scatter).
// Depth sort: 8 launches -> 3; with key compression a dead pass is one launch of blocks that read ~100 words and leave.
template <int BITS>
struct KeyPlanW {
    uint32_t live;        // passes this segment needs (>= 1)
    uint32_t sh[BITS];    // source bit of digit bit i of the current pass
    uint32_t vmask;       // digit bits that exist in this pass
    uint32_t cull;        // digit of a culled key in this pass
    uint32_t plain;       // != 0: the compressed key would need more passes than the plain one has: the key's own bits then
};

template <int BITS>
__device__ __forceinline__ KeyPlanW<BITS> key_plan_w(const uint32_t *__restrict__ bits, uint32_t nrec, uint32_t seg, uint32_t pass,
                                                     uint32_t *kb /* 12 + BITS + 3 words */) {
    constexpr uint32_t MAXP = (32u + BITS - 1u) / BITS;
    uint32_t vor = 0u, vand = 0xFFFFFFFFu, fl = 0u;
    for (uint32_t i = threadIdx.x; i < nrec; i += RS_THREADS) {
        const uint4 r = reinterpret_cast<const uint4 *>(bits)[(size_t)seg * nrec + i];
        vor |= r.x; vand &= r.y; fl |= r.z;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { vor |= __shfl_xor(vor, o, 64); vand &= __shfl_xor(vand, o, 64); fl |= __shfl_xor(fl, o, 64); }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) { kb[3 * wave] = vor; kb[3 * wave + 1] = vand; kb[3 * wave + 2] = fl; }
    __syncthreads();
    if (threadIdx.x < (uint32_t)BITS) {
        uint32_t o = 0u, a = 0xFFFFFFFFu, f = 0u;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) { o |= kb[3 * w]; a &= kb[3 * w + 1]; f |= kb[3 * w + 2]; }
        const uint32_t m = (f & 2u) ? (o ^ a) : 0u;  // no visible key: nothing varies
        const uint32_t nbits = (uint32_t)__popc(m), cull = f & 1u;
        const uint32_t idx = (uint32_t)BITS * pass + threadIdx.x;
        uint32_t mm = m;
        for (uint32_t j = 0; j < idx && mm; ++j) mm &= mm - 1u;
        kb[12 + threadIdx.x] = (idx < nbits) ? (uint32_t)__ffs((int)mm) - 1u : 32u;
        if (threadIdx.x == 0) {
            const uint32_t need = (nbits + cull + BITS - 1u) / BITS;
            kb[12 + BITS] = need < 1u ? 1u : (need > MAXP ? MAXP : need);
            kb[13 + BITS] = (cull && nbits >= BITS * pass && nbits < BITS * pass + BITS) ? 1u << (nbits - BITS * pass) : 0u;
            kb[14 + BITS] = need > MAXP ? 1u : 0u;  // (8-bit digits: 33 bits -- every key bit varies and some keys are culled)
        }
    }
    __syncthreads();
    KeyPlanW<BITS> kp;
    kp.live = kb[12 + BITS];
    kp.cull = kb[13 + BITS];
    kp.plain = kb[14 + BITS];
    kp.vmask = 0u;
#pragma unroll
    for (int i = 0; i < BITS; ++i) {
        const uint32_t p = kb[12 + i];
        kp.sh[i] = p & 31u;
        kp.vmask |= (p < 32u ? 1u : 0u) << i;
    }
    return kp;
}

template <int BITS>
__device__ __forceinline__ uint32_t key_digit_w(const KeyPlanW<BITS> &kp, uint32_t key, uint32_t pass) {
    if (kp.plain) return (key >> (BITS * pass)) & ((1u << BITS) - 1u);  // block-uniform
    uint32_t d = 0u;
#pragma unroll
    for (int i = 0; i < BITS; ++i) d |= ((key >> kp.sh[i]) & 1u) << i;
    return key == 0xFFFFFFFFu ? kp.cull : (d & kp.vmask);
}

// HANDOFF (sort_mode bits 1-2 = 3; VERDICT r4 item 4c, "the look-back pass actually built and timed"): instead of counting the whole
// segment, a block counts its OWN keys, publishes the NB counts (count + 1 per word, write-through agent-scope stores into this pass's
// region of `handoff`, which k_project cleared) and collects the other blocks' words by polling them with agent-scope loads.  A word
// carries its own "published" mark (non-zero), so no flag and no fence are needed (a single dword store is atomic).  The wait is
// BOUNDED: a block that has not seen a word after FGS_HANDOFF_SPINS polls -- its producer is not resident (a grid larger than the
// chip, another stream's kernels holding the CUs) -- stops waiting and counts the whole segment itself, the plain fused path: every
// block terminates whatever the others do.
#ifndef FGS_HANDOFF_SPINS
#define FGS_HANDOFF_SPINS 20000u
#endif
template <int BITS, bool COMPRESSED, bool HANDOFF = false>
__global__ __launch_bounds__(RS_THREADS) void k_radix_fused(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint32_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, uint32_t seg_len, uint32_t seg_stride, uint32_t shift, uint32_t dmask, uint32_t idx_mod,
    const uint32_t *__restrict__ key_bits, uint32_t key_recs, uint32_t pass, uint32_t *__restrict__ vals_final,
    uint32_t *__restrict__ handoff = nullptr /* this pass's region: [segment][block][NB] words */) {
    constexpr uint32_t NB = 1u << BITS, DPT = NB / RS_THREADS;  // bins; bins per thread
    __shared__ uint32_t pre[NB];               // digits of the segment's keys in front of this block's range
    __shared__ uint32_t rest[NB];              // ... from this block's range on; then the running output offset per digit
    __shared__ uint32_t wcnt[RS_WAVES][NB];    // the four waves' running digit counters of a round
    __shared__ uint32_t wtot[RS_WAVES];
    __shared__ uint32_t kb[12 + BITS + 3];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    KeyPlanW<BITS> kp;
    if (COMPRESSED) {
        kp = key_plan_w<BITS>(key_bits, key_recs, blockIdx.y, pass, kb);
        if (pass >= kp.live) return;                        // this segment is sorted already
        if (pass + 1u == kp.live) vals_out = vals_final;    // its last live pass: the payload goes to its final place
    }
    auto digit_of = [&](uint32_t key) -> uint32_t {
        if constexpr (COMPRESSED) return key_digit_w<BITS>(kp, key, pass);
        else return (key >> shift) & dmask;
    };
    const SegInfo r = block_range(seg_len, nullptr, seg_len, seg_stride);
    const uint32_t seg0 = blockIdx.y * seg_stride, seg_end = seg0 + seg_len;
    constexpr uint32_t ROUND = RS_THREADS * 4;
    uint32_t key[4], val[4];
    bool valid[4];
    auto load_round = [&](uint32_t base) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t i = base + wave * 256u + it * 64u + lane;
            valid[it] = i < r.end;
            key[it] = valid[it] ? keys_in[i] : 0u;
            val[it] = valid[it] ? (vals_in ? vals_in[i] : i % idx_mod) : 0u;
        }
    };
#pragma unroll
    for (uint32_t k = 0; k < DPT; ++k) {
        pre[tid + k * RS_THREADS] = 0u; rest[tid + k * RS_THREADS] = 0u;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) wcnt[w][tid + k * RS_THREADS] = 0u;
    }
    __syncthreads();
    bool counted = false;
    if constexpr (HANDOFF) {
        __shared__ volatile uint32_t gave_up;
        if (tid == 0) gave_up = 0u;
        // (1) this block's own digit counts -> `rest`
        for (uint32_t i0 = r.begin + tid; i0 - tid < r.end; i0 += ROUND) {
            uint32_t k4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k4[u] = i0 + u * RS_THREADS < r.end ? keys_in[i0 + u * RS_THREADS] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool v = i0 + u * RS_THREADS < r.end;
                const uint32_t d = digit_of(k4[u]);
                const unsigned long long act = __ballot(v);
                if (act == 0ull) continue;
                const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
                if (__ballot(v && d == d0) == act) {
                    if (lane == 0) atomicAdd(&rest[d0], (uint32_t)__popcll(act));
                } else if (v) {
                    atomicAdd(&rest[d], 1u);
                }
            }
        }
        __syncthreads();
        // (2) publish them, (3) collect the other blocks': every thread owns DPT digits
        uint32_t *mine = handoff + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NB;
        uint32_t own[DPT], before[DPT], total[DPT];
#pragma unroll
        for (uint32_t k = 0; k < DPT; ++k) {
            own[k] = rest[tid + k * RS_THREADS];
            __hip_atomic_store(&mine[tid + k * RS_THREADS], own[k] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            before[k] = 0u; total[k] = own[k];
        }
        bool ok = true;
        for (uint32_t b = 0; b < gridDim.x && ok; ++b) {
            if (b == blockIdx.x) continue;
            const uint32_t *theirs = handoff + ((size_t)blockIdx.y * gridDim.x + b) * NB;
#pragma unroll
            for (uint32_t k = 0; k < DPT; ++k) {
                uint32_t w = 0u, spins = 0u;
                while ((w = __hip_atomic_load(&theirs[tid + k * RS_THREADS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                    if (++spins > FGS_HANDOFF_SPINS || gave_up) { ok = false; break; }  // bounded: see the header
                    __builtin_amdgcn_s_sleep(2);
                }
                if (!ok) break;
                total[k] += w - 1u;
                if (b < blockIdx.x) before[k] += w - 1u;
            }
        }
        if (!ok) gave_up = 1u;
        __syncthreads();
        if (!gave_up) {
#pragma unroll
            for (uint32_t k = 0; k < DPT; ++k) {
                pre[tid + k * RS_THREADS] = before[k];
                rest[tid + k * RS_THREADS] = total[k] - before[k];
            }
            counted = true;
        } else {  // some producer is not running: count the whole segment here (below)
#pragma unroll
            for (uint32_t k = 0; k < DPT; ++k) { pre[tid + k * RS_THREADS] = 0u; rest[tid + k * RS_THREADS] = 0u; }
        }
        __syncthreads();
    }
    // ---- the segment's digit histogram, split at this block's first key (r.begin is a multiple of 1024: a wave instruction's
    // 64 consecutive keys lie on one side) ----
    for (uint32_t i0 = seg0 + tid; !counted && i0 - tid < seg_end; i0 += ROUND) {
        uint32_t k4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k4[u] = i0 + u * RS_THREADS < seg_end ? keys_in[i0 + u * RS_THREADS] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * RS_THREADS;
            const bool v = i < seg_end;
            const uint32_t d = digit_of(k4[u]);
            const unsigned long long act = __ballot(v);
            if (act == 0ull) continue;  // wave-uniform
            uint32_t *h = (__builtin_amdgcn_readfirstlane(i) < r.begin) ? pre : rest;
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);  // (lane 0 of the wave: valid whenever any lane is)
            if (__ballot(v && d == d0) == act) {
                if (lane == 0) atomicAdd(&h[d0], (uint32_t)__popcll(act));  // the wave's 64 keys share one digit (quantised depths)
            } else if (v) {
                atomicAdd(&h[d], 1u);
            }
        }
    }
    load_round(r.begin);  // (this block's first keys travel under the scan below)
    __syncthreads();
    // ---- exclusive scan over the NB digit totals; running output offset of every digit for THIS block ----
    {
        uint32_t t[DPT], p[DPT], loc = 0;
#pragma unroll
        for (uint32_t k = 0; k < DPT; ++k) { p[k] = pre[tid * DPT + k]; t[k] = p[k] + rest[tid * DPT + k]; loc += t[k]; }
        uint32_t sc = loc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(sc, o, 64);
            if ((int)lane >= o) sc += v;
        }
        if (lane == 63) wtot[wave] = sc;
        __syncthreads();  // (every thread has read its bins of `rest`)
        uint32_t run = seg0 + (sc - loc);
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) run += (w < (int)wave) ? wtot[w] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < DPT; ++k) { rest[tid * DPT + k] = run + p[k]; run += t[k]; }
    }
    __syncthreads();
    uint32_t *run_off = rest;
    for (uint32_t base = r.begin; base < r.end; base += ROUND) {
        uint32_t lrank[4], dig[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t digit = digit_of(key[it]);
            dig[it] = digit;
            uint32_t rank, count;
            match_rank<BITS>(digit, valid[it], rank, count);
            const uint32_t prev = valid[it] ? wcnt[wave][digit] : 0u;
            lrank[it] = prev + rank;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (valid[it] && rank == 0) wcnt[wave][digit] = prev + count;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (valid[it]) {
                const uint32_t digit = dig[it];
                uint32_t pr = 0;
#pragma unroll
                for (int w = 0; w < RS_WAVES; ++w) pr += (w < (int)wave) ? wcnt[w][digit] : 0u;
                const uint32_t dst = run_off[digit] + pr + lrank[it];
                keys_out[dst] = key[it];
                vals_out[dst] = val[it];
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < DPT; ++k) {
            const uint32_t d = tid + k * RS_THREADS;
            uint32_t tot = 0;
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) { tot += wcnt[w][d]; wcnt[w][d] = 0; }
            run_off[d] += tot;
        }
        if (base + ROUND < r.end) load_round(base + ROUND);
        __syncthreads();
    }
}

// ---- WHOLE SEGMENT IN LDS (round 5) ---------------------------------------------------------------------------------------------
// What the A/B of the fused passes showed (profiles/r05_ab_sort_passes.txt): a depth-sort pass costs its launch boundary and two
// memory round trips, not its bytes -- 16 x 8192 keys are 1 MB.  A segment of at most 8192 (key, payload) pairs fits the LDS of ONE
// compute unit twice over (2 x 64 KB of pairs + 16 KB of rank counters), so one 16-wave block per segment runs ALL the passes there:
// one launch instead of eight, one read of the keys and one write of the order, no histogram in memory.  A pass is
// k_radix_downsweep's: wave w owns the w-th contiguous slice of the segment (<= 8 chunks of 64 keys, held in registers), ranks its
// keys chunk by chunk with wave64 ballots against its own row of digit counters (stable, no atomics), 256 threads turn the 16 x 256
// counters into each wave's output base per digit (prefix over the waves + prefix over the digits), every wave scatters its keys into
// the other LDS buffer.  Four block barriers per pass.  Zone keys (COMPRESSED): the key is compressed ONCE, when it is loaded -- pext(key, varying bits),
// a culled key 1 << popcount -- and the passes that the compressed keys need run on plain 8-bit digits of it.
constexpr int SL_THREADS = 1024, SL_WAVES = SL_THREADS / 64, SL_CAP = 8192, SL_CH = SL_CAP / SL_THREADS;

template <bool COMPRESSED>
__global__ __launch_bounds__(SL_THREADS) void k_sort_segment_lds(
    const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals_in, uint32_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, uint32_t seg_len, uint32_t seg_stride, uint32_t passes, uint32_t idx_mod,
    const uint32_t *__restrict__ key_bits, uint32_t key_recs) {
    __shared__ uint32_t ks[2][SL_CAP], vs[2][SL_CAP];
    __shared__ uint32_t wcnt[SL_WAVES][256];
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t kb[3 * SL_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t seg0 = blockIdx.x * seg_stride;
    const uint32_t nch = (seg_len + SL_THREADS - 1u) / SL_THREADS;  // chunks per wave (host-checked: <= SL_CH)
    const uint32_t w0 = wave * nch * 64u;                           // wave w owns the keys [w0, w0 + 64 nch) of the segment
    uint32_t key[SL_CH], val[SL_CH];
    bool valid[SL_CH];
#pragma unroll
    for (int c = 0; c < SL_CH; ++c) {
        const uint32_t i = w0 + (uint32_t)c * 64u + lane;
        valid[c] = (uint32_t)c < nch && i < seg_len;
        key[c] = valid[c] ? keys[seg0 + i] : 0u;
        // (the depth sort's payload: the index inside the image -- seg0 is a multiple of idx_mod there, no division)
        val[c] = valid[c] ? (vals_in ? vals_in[seg0 + i] : (idx_mod == seg_stride ? i : (seg0 + i) % idx_mod)) : 0u;
    }
    if (COMPRESSED) {
        uint32_t vor = 0u, vand = 0xFFFFFFFFu, fl = 0u;
        for (uint32_t i = tid; i < key_recs; i += SL_THREADS) {
            const uint4 r = reinterpret_cast<const uint4 *>(key_bits)[(size_t)blockIdx.x * key_recs + i];
            vor |= r.x; vand &= r.y; fl |= r.z;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { vor |= __shfl_xor(vor, o, 64); vand &= __shfl_xor(vand, o, 64); fl |= __shfl_xor(fl, o, 64); }
        if (lane == 0) { kb[3 * wave] = vor; kb[3 * wave + 1] = vand; kb[3 * wave + 2] = fl; }
        __syncthreads();
        uint32_t o = 0u, a = 0xFFFFFFFFu, f = 0u;
#pragma unroll
        for (int w = 0; w < SL_WAVES; ++w) { o |= kb[3 * w]; a &= kb[3 * w + 1]; f |= kb[3 * w + 2]; }
        const uint32_t m = (f & 2u) ? (o ^ a) : 0u;  // no visible key: nothing varies
        const uint32_t nbits = (uint32_t)__popc(m), cull = f & 1u;
        const uint32_t need = (nbits + cull + 7u) / 8u;
        passes = need < 1u ? 1u : (need > 4u ? 4u : need);
        if (need <= 4u) {  // (33 bits -- every key bit varies and some keys are culled -- is the key itself, as in key_digit)
            uint32_t d[SL_CH];
#pragma unroll
            for (int c = 0; c < SL_CH; ++c) d[c] = 0u;
            uint32_t mm = m, i = 0u;
            while (mm) {  // block-uniform: the varying bits, lowest first
                const uint32_t pos = (uint32_t)__ffs((int)mm) - 1u;
#pragma unroll
                for (int c = 0; c < SL_CH; ++c) d[c] |= ((key[c] >> pos) & 1u) << i;
                ++i; mm &= mm - 1u;
            }
#pragma unroll
            for (int c = 0; c < SL_CH; ++c) key[c] = key[c] == 0xFFFFFFFFu ? 1u << nbits : d[c];  // culled: behind every visible key
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) wcnt[wave][lane + 64u * k] = 0u;  // (the wave's own row: its LDS operations execute in order)
    for (uint32_t p = 0; p < passes; ++p) {
        const uint32_t shift = 8u * p;
        uint32_t lrank[SL_CH], dig[SL_CH];
        // (ranking all chunks first and walking the counters afterwards -- the LDS chain out of the ballots' way -- is SLOWER: 32 against
        //  29.5 us at 16 x 8192 keys, profiles/r05_ab_sort_lds.txt: the waves of a SIMD fall into step and wait together)
#pragma unroll
        for (int c = 0; c < SL_CH; ++c) {
            if ((uint32_t)c >= nch) continue;  // block-uniform
            const uint32_t digit = (key[c] >> shift) & 0xFFu;
            dig[c] = digit;
            uint32_t rank, count;
            match_rank<8>(digit, valid[c], rank, count);
            const uint32_t prev = valid[c] ? wcnt[wave][digit] : 0u;
            lrank[c] = prev + rank;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (valid[c] && rank == 0) wcnt[wave][digit] = prev + count;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        __syncthreads();
        // digit `tid` (threads 0 ... 255): the waves' counts -> their exclusive prefix over the waves + the digit's exclusive prefix over the
        // digits, written back over the counts: the scatter then reads ONE word per key (its wave's base for its digit)
        uint32_t cw[SL_WAVES], acc = 0u, sc = 0u;
        if (tid < 256u) {
#pragma unroll
            for (int w = 0; w < SL_WAVES; ++w) cw[w] = wcnt[w][tid];
#pragma unroll
            for (int w = 0; w < SL_WAVES; ++w) { const uint32_t x = cw[w]; cw[w] = acc; acc += x; }
            sc = acc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(sc, o, 64);
                if ((int)lane >= o) sc += v;
            }
            if (lane == 63u) wtot[wave] = sc;
        }
        __syncthreads();
        if (tid < 256u) {
            uint32_t base = sc - acc;
#pragma unroll
            for (int w = 0; w < 3; ++w) base += (w < (int)wave) ? wtot[w] : 0u;
#pragma unroll
            for (int w = 0; w < SL_WAVES; ++w) wcnt[w][tid] = base + cw[w];
        }
        __syncthreads();
        const bool last = p + 1u == passes;
        uint32_t *kd = ks[p & 1u], *vd = vs[p & 1u];
#pragma unroll
        for (int c = 0; c < SL_CH; ++c) {
            if (valid[c]) {  // (false for c >= nch)
                const uint32_t d = dig[c];
                const uint32_t dst = wcnt[wave][d] + lrank[c];
                if (last) {
                    vals_out[seg0 + dst] = val[c];
                    if (!COMPRESSED && keys_out) keys_out[seg0 + dst] = key[c];
                } else {
                    kd[dst] = key[c]; vd[dst] = val[c];
                }
            }
        }
        if (last) break;
        __syncthreads();  // the pass's output is in place; the counters are free
#pragma unroll
        for (int k = 0; k < 4; ++k) wcnt[wave][lane + 64u * k] = 0u;
#pragma unroll
        for (int c = 0; c < SL_CH; ++c) {
            const uint32_t i = w0 + (uint32_t)c * 64u + lane;
            if (valid[c]) { key[c] = kd[i]; val[c] = vd[i]; }
        }
    }
}

// blocks per segment of the fused pass: ~2048 keys of its own per block (ranking costs ~6x the histogram per key), <= 16 (every
// block reads the whole segment), <= 4096 blocks per launch
uint32_t fused_blocks_per_seg(uint32_t seg_len, uint32_t num_segs) {
    uint32_t bps = (seg_len + 2047u) / 2048u;
    if (bps > 16u) bps = 16u;
    while (bps > 1u && (size_t)bps * num_segs > 4096u) bps >>= 1;
    return bps < 1u ? 1u : bps;
}

uint32_t dmask_of(uint32_t p, uint32_t width, uint32_t key_bits) {
    const uint32_t left = key_bits - p * width;
    return (1u << (left < width ? left : width)) - 1u;
}

}  // namespace

int fgs_launch_radix_sort(uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_alt, uint32_t *vals_alt,
                          uint32_t *vals_final, uint32_t **keys_sorted, uint32_t **vals_sorted,
                          uint32_t seg_len, const uint32_t *seg_len_dev, uint32_t seg_capacity,
                          uint32_t seg_stride, uint32_t num_segs, uint32_t key_bits, uint32_t *hist,
                          hipStream_t st, const uint32_t *keys_first, uint32_t index_payload_mod,
                          const uint32_t *key_stats, uint32_t key_recs, int pass_mode) {
    // pass_mode (FgsDims.sort_mode >> 1): 1 = the fused pass (one launch per pass, every block recounting its segment, 11-bit digits)
    // for any host-known segment length up to 64 K keys | 2 = the same with 8-bit digits | 3 = 8-bit digits with the hand-off (A/B runs,
    // agreement tests).  Measured (profiles/r05_ab_sort_passes.txt): at 16 x 8 192 keys the fused passes take 45.6 us (11-bit) / 55.5
    // (8-bit) against 42.0 for the two-launch passes, at 8 x 32 768 keys 100.9 / 127.8 against 53.8 -- every block counting the WHOLE
    // segment's digits with LDS atomics costs more than the launch boundary it saves (~2.8 us); a pass is two memory round trips
    // whichever way it is launched.  (Until the LDS sort below, automatic used the 11-bit fused pass for segments of <= 4096 keys.)
    // Round 5, later: 0 = automatic now means the whole-segment LDS sort (k_sort_segment_lds: ONE launch for all passes) for
    // host-known segments of at most 8192 keys, the two-launch passes above that | 4 = the two-launch passes for any size | 5 = the LDS
    // sort where it applies (as automatic).  profiles/r05_ab_sort_lds.txt.
    const bool host_len = !seg_len_dev && seg_len == seg_capacity && key_bits >= 1u;
    if (host_len && seg_len <= (uint32_t)SL_CAP && (pass_mode == 0 || pass_mode == 5) && (vals_in || index_payload_mod)) {
        const bool compressed = key_stats != nullptr;
        if (compressed && !(keys_first && index_payload_mod && vals_final && key_recs && key_bits == 32u)) {
            fgs_set_error("radix sort: key compression needs 32-bit keys in keys_first, an index payload and vals_final");
            return FGS_EINVAL;
        }
        const uint32_t *kin = keys_first ? keys_first : keys_in;
        uint32_t *kdst = compressed ? nullptr : (keys_first ? keys_in : keys_alt);
        uint32_t *vdst = vals_final ? vals_final : vals_alt;
        const uint32_t *vsrc = index_payload_mod ? nullptr : vals_in;
        const uint32_t passes = (key_bits + 7u) / 8u, imod = index_payload_mod ? index_payload_mod : 1u;
        if (compressed)
            hipLaunchKernelGGL(k_sort_segment_lds<true>, dim3(num_segs), dim3(SL_THREADS), 0, st, kin, vsrc, kdst, vdst, seg_len, seg_stride,
                               passes, imod, key_stats, key_recs);
        else
            hipLaunchKernelGGL(k_sort_segment_lds<false>, dim3(num_segs), dim3(SL_THREADS), 0, st, kin, vsrc, kdst, vdst, seg_len, seg_stride,
                               passes, imod, (const uint32_t *)nullptr, 0u);
        FGS_LAUNCH_CHECK("k_sort_segment_lds");
        *keys_sorted = kdst;
        *vals_sorted = vdst;
        return FGS_OK;
    }
    const bool fused = host_len && pass_mode >= 1 && pass_mode <= 3 && seg_len <= 65536u;
    // (pass_mode 3, the hand-off form: a compressed segment that leaves a pass early publishes nothing for it, and nobody of that
    //  segment waits either -- all its blocks leave together; region p belongs to pass p alone, so no word is ever reused in a call)
    if (fused) {
        const bool compressed = key_stats != nullptr;
        if (compressed && !(keys_first && index_payload_mod && vals_final && key_recs && key_bits == 32u)) {
            fgs_set_error("radix sort: key compression needs 32-bit keys in keys_first, an index payload and vals_final");
            return FGS_EINVAL;
        }
        const uint32_t bits = (pass_mode >= 2 || key_bits <= 8u) ? 8u : 11u;  // (plane ids, <= 5 bits: the small histogram)
        // hand-off form: the depth sort of a forward whose projection cleared `hist` (fgs_api.hip); needs its four pass regions
        const bool handoff = pass_mode == 3 && hist != nullptr && key_bits == 32u && keys_first != nullptr;
        const uint32_t passes = (key_bits + bits - 1u) / bits, width = (key_bits + passes - 1u) / passes;
        const dim3 grid(fused_blocks_per_seg(seg_len, num_segs), num_segs);
        const uint32_t *kin = keys_first ? keys_first : keys_in;
        uint32_t *vin = vals_in, *kout = keys_first ? keys_in : keys_alt, *kspare = keys_alt;
        for (uint32_t p = 0; p < passes; ++p) {
            uint32_t *vdst = (p == passes - 1 && vals_final) ? vals_final : (vin == vals_in ? vals_alt : vals_in);
            if (p == 0 && index_payload_mod && !(passes == 1 && vals_final)) vdst = vals_in;  // vals_in is free: nothing to read
            const uint32_t *vsrc = (p == 0 && index_payload_mod) ? nullptr : vin;
            const uint32_t shift = p * width, dmask = dmask_of(p, width, key_bits), imod = index_payload_mod ? index_payload_mod : 1u;
#define FGS_FUSED_LAUNCH(B_, C_)                                                                                             \
    hipLaunchKernelGGL((k_radix_fused<B_, C_>), grid, dim3(RS_THREADS), 0, st, kin, vsrc, kout, vdst, seg_len, seg_stride, shift, \
                       dmask, imod, key_stats, key_recs, p, vals_final)
            if (handoff) {
                uint32_t *region = hist + (size_t)p * num_segs * 16u * 256u;
                if (compressed)
                    hipLaunchKernelGGL((k_radix_fused<8, true, true>), grid, dim3(RS_THREADS), 0, st, kin, vsrc, kout, vdst, seg_len, seg_stride,
                                       shift, dmask, imod, key_stats, key_recs, p, vals_final, region);
                else
                    hipLaunchKernelGGL((k_radix_fused<8, false, true>), grid, dim3(RS_THREADS), 0, st, kin, vsrc, kout, vdst, seg_len, seg_stride,
                                       shift, dmask, imod, key_stats, key_recs, p, vals_final, region);
            } else if (bits == 8u) { if (compressed) FGS_FUSED_LAUNCH(8, true); else FGS_FUSED_LAUNCH(8, false); }
            else            { if (compressed) FGS_FUSED_LAUNCH(11, true); else FGS_FUSED_LAUNCH(11, false); }
#undef FGS_FUSED_LAUNCH
            FGS_LAUNCH_CHECK("k_radix_fused");
            uint32_t *next_out = (p == 0 && keys_first) ? kspare : const_cast<uint32_t *>(kin);
            kin = kout; kout = next_out;
            vin = vdst;
        }
        *keys_sorted = compressed ? nullptr : const_cast<uint32_t *>(kin);  // (compressed: see the end of this function)
        *vals_sorted = compressed ? vals_final : vin;
        return FGS_OK;
    }
    const uint32_t bps = fgs_radix_blocks_per_seg(seg_capacity, num_segs);
    // key compression (depth sort): full 32-bit keys read from keys_first, the index as payload, a final place for the payload
    const bool compressed = key_stats != nullptr;
    if (compressed && !(keys_first && index_payload_mod && vals_final && key_recs && !seg_len_dev && key_bits == 32u)) {
        fgs_set_error("radix sort: key compression needs 32-bit keys in keys_first, an index payload and vals_final");
        return FGS_EINVAL;
    }
    uint32_t *dtot = hist + (size_t)num_segs * 256 * bps;
    const uint32_t passes = (key_bits + 7) / 8;
    // equal digit widths over the passes (13 key bits -> 7 + 6, not 8 + 5): fewer bins per pass means longer
    // runs of neighbouring destinations in the scatter
    const uint32_t width = passes ? (key_bits + passes - 1) / passes : 8;
    // keys_first: the first pass reads its keys from there (read-only) instead of keys_in; index_payload_mod != 0:
    // the first pass generates the payload (element index modulo it) instead of reading vals_in
    const uint32_t *kin = keys_first ? keys_first : keys_in;
    uint32_t *vin = vals_in, *kout = keys_first ? keys_in : keys_alt, *vout = vals_alt;
    uint32_t *kspare = keys_alt;
    const bool fused_scan = bps <= 64;
    if (passes == 0) {
        if (vals_final && vals_final != vals_in) {
            // single possible key: already sorted; move the payload where the caller wants it
            hipError_t e = hipMemcpyAsync(vals_final, vals_in, (size_t)seg_stride * (num_segs - 1) * 4 + (size_t)seg_capacity * 4,
                                          hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) { fgs_set_error("radix copy: %s", hipGetErrorString(e)); return FGS_ELAUNCH; }
            vin = vals_final;
        }
        *keys_sorted = const_cast<uint32_t *>(kin); *vals_sorted = vin;
        return FGS_OK;
    }
    (void)vout;
    for (uint32_t p = 0; p < passes; ++p) {
        uint32_t *vdst = (p == passes - 1 && vals_final) ? vals_final : (vin == vals_in ? vals_alt : vals_in);
        if (p == 0 && index_payload_mod && !(passes == 1 && vals_final)) vdst = vals_in;  // vals_in is free: nothing to read
        const dim3 grid(bps, num_segs);
        if (compressed)
            hipLaunchKernelGGL(k_radix_upsweep<true>, grid, dim3(RS_THREADS), 0, st, kin, seg_len, seg_len_dev, seg_capacity,
                               seg_stride, 0u, 0u, hist, key_stats, key_recs, p);
        else
            hipLaunchKernelGGL(k_radix_upsweep<false>, grid, dim3(RS_THREADS), 0, st, kin, seg_len, seg_len_dev, seg_capacity,
                               seg_stride, p * width, dmask_of(p, width, key_bits), hist, (const uint32_t *)nullptr, 0u, p);
        FGS_LAUNCH_CHECK("k_radix_upsweep");
        const uint32_t rows = num_segs * 256;
        if (!fused_scan) {
            hipLaunchKernelGGL(k_radix_scan, dim3((rows + 3) / 4), dim3(256), 0, st, hist, dtot, bps, rows);
            FGS_LAUNCH_CHECK("k_radix_scan");
        }
        const uint32_t *vsrc = (p == 0 && index_payload_mod) ? nullptr : vin;
        if (compressed)
            hipLaunchKernelGGL(k_radix_downsweep<true>, grid, dim3(RS_THREADS), 0, st, kin, vsrc, kout, vdst, seg_len, seg_len_dev,
                               seg_capacity, seg_stride, 0u, 0u, hist, fused_scan ? (const uint32_t *)nullptr : dtot,
                               index_payload_mod, key_stats, key_recs, p, vals_final);
        else
            hipLaunchKernelGGL(k_radix_downsweep<false>, grid, dim3(RS_THREADS), 0, st, kin, vsrc, kout, vdst, seg_len, seg_len_dev,
                               seg_capacity, seg_stride, p * width, dmask_of(p, width, key_bits), hist,
                               fused_scan ? (const uint32_t *)nullptr : dtot, index_payload_mod ? index_payload_mod : 1u,
                               (const uint32_t *)nullptr, 0u, p, (uint32_t *)nullptr);
        FGS_LAUNCH_CHECK("k_radix_downsweep");
        // ping-pong: the buffer just read becomes the next output, except a read-only first-pass source
        uint32_t *next_out = (p == 0 && keys_first) ? kspare : const_cast<uint32_t *>(kin);
        kin = kout; kout = next_out;
        vin = vdst;
    }
    // Key compression: a segment stops after its own number of live passes, so NO buffer holds every segment's sorted keys
    // (the pass-4 buffer has stale data for segments that needed fewer) and the payload is in vals_final: hand back no key
    // pointer rather than one that looks valid (the depth sort's caller reads `order` only).
    *keys_sorted = compressed ? nullptr : const_cast<uint32_t *>(kin);
    *vals_sorted = compressed ? vals_final : vin;
    return FGS_OK;
}

