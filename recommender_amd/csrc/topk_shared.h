// Wave-level top-k primitives shared by the stand-alone select (topk.hip) and the fused score + select tile kernel
// (tower.hip): order-preserving keys, the sorted register-resident list, batched entrants.  See topk.hip for the scheme.
#pragma once
#include "reco_common.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t float_key(float v)
{
    if (v != v) return 0xFFFFFFFFu;          // NaN ranks first (numpy sorts NaN last ascending; torch.topk: largest)
    v += 0.f;                                // -0.0 -> +0.0 so that equal scores have equal keys
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float key_float(uint32_t k)
{
    if (k == 0xFFFFFFFFu) return __uint_as_float(0x7FC00000u);
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

__device__ __forceinline__ u64 shfl_up1(u64 v, int lane)
{
    const int src = lane > 0 ? lane - 1 : 0;
    const uint32_t lo = __shfl((uint32_t)v, src);
    const uint32_t hi = __shfl((uint32_t)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 bcast(u64 v, int src)
{
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}

// v of lane (l ^ mask).  Partners inside a row of 16 lanes come through DPP (a VALU move, a few cycles) instead of an LDS
// cross-lane round trip: quad_perm for 1 and 2, a rotation by 8 for 8, and for 4 a shift left into banks 0/2 plus a shift
// right into banks 1/3 (row_shl:n = lane i reads lane i + n, row_shr:n = lane i reads lane i - n; bank b = lanes 4b..4b+3).
__device__ __forceinline__ uint32_t shfl_xor32(uint32_t v, int mask)
{
    const int x = (int)v;
    switch (mask) {
        case 1: return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);     // quad_perm [1,0,3,2]
        case 2: return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
        case 4: {
            const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);           // row_shl:4 -> banks 0, 2
            return (uint32_t)__builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);        // row_shr:4 -> banks 1, 3
        }
        case 8: return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false);     // row_ror:8
        default: return (uint32_t)__shfl_xor(x, mask);
    }
}

__device__ __forceinline__ u64 shfl_xor64(u64 v, int mask)
{
    const uint32_t lo = shfl_xor32((uint32_t)v, mask);
    const uint32_t hi = shfl_xor32((uint32_t)(v >> 32), mask);
    return ((u64)hi << 32) | lo;
}

// one compare-exchange stage of a descending bitonic network: the lower lane of each pair keeps the larger key
__device__ __forceinline__ u64 cmpx_desc(u64 v, int j, int lane, bool descending)
{
    const u64 o = shfl_xor64(v, j);
    const bool lower = (lane & j) == 0;
    return (lower == descending) ? (v > o ? v : o) : (v < o ? v : o);
}

// bitonic sort of one key per lane, descending across the wave (21 stages)
__device__ __forceinline__ u64 wave_sort_desc(u64 v, int lane)
{
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) v = cmpx_desc(v, j, lane, (lane & k) == 0);
    return v;
}

// Sorted (descending) list of 64 * SLOTS keys across the wave.  0 = empty.
template <int SLOTS>
struct WaveList {
    u64 e[SLOTS];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) e[s] = 0;
    }
    // x is wave-uniform
    __device__ __forceinline__ void insert(u64 x, int lane)
    {
        u64 carry = x;   // element entering slot s (at position pos, or at lane 0 when carried over)
        bool placed = false;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            if (!placed) {
                const int pos = __popcll(__ballot(e[s] > x));   // e[s] sorted descending: prefix of greater keys
                if (pos < 64) {
                    const u64 up = shfl_up1(e[s], lane);
                    const u64 last = bcast(e[s], 63);
                    e[s] = lane < pos ? e[s] : (lane == pos ? x : up);
                    carry = last;
                    placed = true;
                }
            } else {
                const u64 up = shfl_up1(e[s], lane);
                const u64 last = bcast(e[s], 63);
                e[s] = lane == 0 ? carry : up;
                carry = last;
            }
        }
    }
    // c: 64 keys sorted descending across the lanes.  The list becomes the best 64 * SLOTS of (list U c):
    // max(list[r], c_padded[last - r]) holds that set as a bitonic sequence, log2 merge stages sort it.
    __device__ __forceinline__ void merge_sorted(u64 c, int lane)
    {
        const int src = 63 - lane;
        const uint32_t lo = __shfl((uint32_t)c, src), hi = __shfl((uint32_t)(c >> 32), src);
        const u64 r = ((u64)hi << 32) | lo;
        e[SLOTS - 1] = e[SLOTS - 1] > r ? e[SLOTS - 1] : r;
        if (SLOTS == 2) {
            const u64 a = e[0], b = e[SLOTS - 1];
            e[0] = a > b ? a : b;
            e[SLOTS - 1] = a > b ? b : a;
        }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
#pragma unroll
            for (int j = 32; j > 0; j >>= 1) e[s] = cmpx_desc(e[s], j, lane, true);
    }
    __device__ __forceinline__ u64 kth(int k) const   // key at rank k-1 (wave-uniform)
    {
        u64 r = 0;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (((k - 1) >> 6) == s) r = bcast(e[s], (k - 1) & 63);
        return r;
    }
};

// Candidates are not inserted as they appear: keys that beat the current k-th key are collected in a per-wave PENDING ring in
// LDS and only when 64 are pending -- or the row segment ends -- are they sorted and merged with the list as two sorted
// sequences; a handful of keys are inserted serially instead.  A serial insert costs ~40 instructions per key and slot;
// batched, an entrant costs ~5.  The threshold is the k-th key as of the last flush, so a few more keys than strictly necessary
// are collected; the merge drops them.
constexpr int SERIAL_INSERTS_MAX = 4;

// ring: RING 64-bit slots of LDS owned by the wave.  A candidate is stored at ring[n + (candidates in lower lanes)] by a plain
// ds_write (nothing is waited for; compacting them into a pending REGISTER with ds_permute pushes cost two LDS round trips per
// offer that held a candidate), and a flush sorts the pending keys by RANK: every lane counts the pending keys above its own
// through broadcast LDS reads -- independent reads and compares instead of the 21 dependent cross-lane stages of a bitonic
// network -- and stores its key at its rank.  LDS operations of one wave execute in issue order, so no barrier is needed
// between the phases.  (6,038 x 3,533, k = 10: 0.037 -> 0.030 ms; 256 x 1M, k = 100: 0.318 -> 0.288 ms, same-box A/B.)
struct RingPending {
    int n;       // wave-uniform
    __device__ __forceinline__ void clear() { n = 0; }
};

template <int SLOTS, int RING>
__device__ __forceinline__ void flush_ring(WaveList<SLOTS>& L, RingPending& P, u64* ring, u64& thr, int k, int lane)
{
    static_assert(RING <= 64, "one pending key per lane at most");
    if (P.n == 0) return;
    const int n = P.n;
    P.n = 0;
    const u64 mine = lane < n ? ring[lane] : 0;
    if (n <= SERIAL_INSERTS_MAX) {
        for (int i = 0; i < n; ++i) {
            const u64 x = bcast(mine, i);
            if (x > thr) {               // thr may have risen since the key was collected
                L.insert(x, lane);
                thr = L.kth(k);
            }
        }
        return;
    }
    int rank = 0;
    for (int j = 0; j + 1 < n; j += 2) {                    // two keys per broadcast ds_read_b128
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 o = *reinterpret_cast<const u64x2*>(ring + j);
        rank += (o.x > mine) + (o.y > mine);
    }
    if (n & 1) rank += ring[n - 1] > mine;
    // keys are distinct (they carry the position), so the ranks of the n pending keys are 0 .. n-1; the other lanes (key 0,
    // rank n) keep their own slot
    ring[lane < n ? rank : lane] = mine;
    const u64 sorted = lane < RING ? ring[lane] : 0;
    L.merge_sorted(sorted, lane);
    thr = L.kth(k);
}

template <int SLOTS, int RING>
__device__ __forceinline__ void offer_ring(WaveList<SLOTS>& L, RingPending& P, u64* ring, u64 key, bool valid, u64& thr, int k, int lane)
{
    bool cand = valid && key > thr;
    u64 m = __ballot(cand);
    int c = __popcll(m);
    if (c == 0) return;
    if (P.n + c > RING) {
        flush_ring<SLOTS, RING>(L, P, ring, thr, k, lane);
        cand = cand && key > thr;
        m = __ballot(cand);
        c = __popcll(m);
        if (c == 0) return;
        if (c > RING) {              // RING < 64 only: more candidates in one offer than the ring holds -> straight into the list
            L.merge_sorted(wave_sort_desc(cand ? key : 0, lane), lane);
            thr = L.kth(k);
            return;
        }
    }
    const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (cand) ring[P.n + below] = key;
    P.n += c;
}

}  // namespace
