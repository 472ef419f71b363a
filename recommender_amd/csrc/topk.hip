// K-d: wavefront-level per-row top-k select, gfx950.
//
// Replaces the reference's full sorts for K << N:
//   np.argsort(scores)[::-1][:k]        src/api/core/inference_engine.py:166
//   torch.topk(pred, 10) + torch.take   src/ncf/ncf.ipynb cell 8
//   np.argsort(p)[-k:][::-1]            src/autorec/utils/helper.py:335
//   list.sort(reverse=True)[:k]         src/api/core/inference_engine.py:294,341
// with the seen-item mask of helper.py:313 fused into the read.
//
// One wave streams one row segment with coalesced loads and keeps the best K (<= 128) candidates
// SORTED IN REGISTERS, rank r in lane r & 63 (slot r >> 6).  Every element is turned into a 64-bit
// key  (order-preserving float bits << 32) | ~position, so "descending score, ascending position
// among equals" is a single unsigned compare and ties are deterministic.  An element
// enters the list only if its key beats the current K-th key (one v_cmp + ballot per 64 elements in
// the steady state: the kernel is HBM-bound, 4 N bytes per row); entrants are collected 64 at a time, bitonic-sorted across the
// wave and merged with the list instead of being inserted one by one.  Long rows are split over several
// waves; a second launch merges the per-segment lists the same way.
#include "reco_common.h"
#include "topk_shared.h"

// Long-row scan shape, from same-box A/B runs on 256 x 1M, k = 100 (library variants through RECO_HIP_LIB):
//   float4 loads in flight per lane x waves the rows are split into -> ms
//   4 x 4096: 0.329   6 x 3072: 0.314   5 x 3072: 0.323   3 x 4096: 0.325   4 x 3072: 0.351   6 x 2560: 0.339
//   6 x 4096: 0.413 (over 128 VGPRs: a fourth wave per SIMD no longer fits)   8 x 4096: 0.554   8 x 2048: 0.485
// The list upkeep of a segment's warm-up (k ln(n/k) entrants), not the streaming, is what is left: fewer, longer segments win
// as long as three waves per SIMD keep the loads covered.  Masked scans carry the mask registers too and stay at 4 x 4096.
#ifndef RECO_TOPK_UN
#define RECO_TOPK_UN 6
#endif
#ifndef RECO_TOPK_UN_MASKED
#define RECO_TOPK_UN_MASKED 4
#endif
// Catalogue-length rows (one segment of a few thousand elements: 6,040 x 3,706 AutoRec / protocol rows) are another regime:
// six float4 per lane cover 1,536 elements an iteration, so a 3,706-element row wastes a third of its last iteration and has
// nothing to overlap with.  Same-box A/B (round 3), 6,038 x 3,533 unmasked / 6,040 x 3,706 masked by the rating matrix, ms:
//   loads in flight 6 (4 masked): 0.0416 / 0.0479   4: 0.0384 / 0.0478   3: 0.0362 / 0.0450   2: 0.0370 / 0.0480   1: 0.0358 / 0.0460
// Then the wave timeline of the lane kernel (below) showed what such launches are short of: waves per SIMD.  6,040 one-wave rows
// are 5.9 waves per SIMD, the kernel held 102 / 130 VGPRs (4 / 3 waves per SIMD: two rounds).  With ONE float4 in flight per
// lane and the register budget of six waves per SIMD (no spill; two in flight spill under the mask) the whole grid is resident
// at once: 0.0362 / 0.0455 -> 0.0292 / 0.0348 ms.
// Later in round 3, the opposite trade, measured and rejected: a whole row put in flight at once with LDS-direct loads (one wave per
// workgroup, 1 KB of LDS per 256 elements, offers out of LDS: one memory round trip per row instead of fourteen) -- correct, and
// slower: 0.0294 -> 0.0429 / 0.0372 -> 0.0542 ms.  Ten such rows fit a CU where this kernel keeps twenty-four waves: the scan is
// short of instruction issue slots, not of bytes in flight.
#ifndef RECO_TOPK_UN_SHORT
#define RECO_TOPK_UN_SHORT 1  // segments up to RECO_TOPK_SHORT_SEG elements
#endif
#ifndef RECO_TOPK_UN_SHORT_MASKED
#define RECO_TOPK_UN_SHORT_MASKED 1
#endif
#ifndef RECO_TOPK_SHORT_SEG
#define RECO_TOPK_SHORT_SEG 8192
#endif
#ifndef RECO_TOPK_WAVES
#define RECO_TOPK_WAVES 3072
#endif
#ifndef RECO_TOPK_RING
#define RECO_TOPK_RING 64    // pending candidates per wave before a flush
#endif
#ifndef RECO_TOPK_LANE_SELECT
#define RECO_TOPK_LANE_SELECT 1   // 0: the one-pass lane kernel for every short-row shape (A/B)
#endif
#ifndef RECO_TOPK_WAVES_MASKED
#define RECO_TOPK_WAVES_MASKED 4096
#endif

namespace {

struct TopkArgs {
    const float* scores;
    int64_t rows, N, ld;
    int k;
    const uint8_t* exclude;
    int64_t ld_ex;
    const float* seen;      // nullable: float matrix whose entries > 0 mark seen items (train_mask = train_mat > 0, helper.py:272)
    int64_t ld_seen;
    const int64_t* row_ids; // nullable: output row r selects row row_ids[r] of scores / exclude / seen
    int64_t src_rows;       // rows of the source matrices when row_ids is given (ids are clamped into [0, src_rows))
    int32_t* oob;           // nullable: count of clamped row ids
    int splits;
    int64_t seg;            // elements per split (multiple of 64)
    u64* partial;           // [rows][splits][k] when splits > 1
    int32_t* out_ids;
    float* out_vals;
};

// VEC = 4: rows are 16-byte aligned (ld % 4 == 0): each lane loads float4s, two in flight per iteration (512 elements per
// wave and iteration); VEC = 1: any alignment, one float per lane and load, four in flight.
// MASKED = false: no mask operand at all (the mask registers that travel with the scores cost the unmasked scan a third of
// its speed on long rows: 0.354 -> 0.475 ms on 256 x 1M when they were carried unconditionally).
#ifndef RECO_TOPK_SHORT_OCC
#define RECO_TOPK_SHORT_OCC 6          // waves per SIMD the short-segment scans are compiled for (see RECO_TOPK_UN_SHORT)
#endif
#ifndef RECO_TOPK_SHORT_OCC_MASKED
#define RECO_TOPK_SHORT_OCC_MASKED 6
#endif
template <int SLOTS, int VEC, bool MASKED, bool SHORT>
__global__ __launch_bounds__(256, SHORT ? (MASKED ? RECO_TOPK_SHORT_OCC_MASKED : RECO_TOPK_SHORT_OCC) : 1) void topk_scan_kernel(TopkArgs a)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= a.rows * a.splits) return;
    const int64_t row = w / a.splits;
    const int split = (int)(w % a.splits);
    const int64_t s0 = split * a.seg;
    const int64_t s1 = s0 + a.seg < a.N ? s0 + a.seg : a.N;
    int64_t srow = row;
    if (a.row_ids) {
        srow = a.row_ids[row];
        if (srow < 0 || srow >= a.src_rows) {
            if (a.oob && lane == 0 && split == 0) atomicAdd(a.oob, 1);
            srow = srow < 0 ? 0 : a.src_rows - 1;
        }
    }
    const float* src = a.scores + srow * a.ld;
    const uint8_t* ex = (MASKED && a.exclude) ? a.exclude + srow * a.ld_ex : nullptr;
    const float* sf = (MASKED && a.seen) ? a.seen + srow * a.ld_seen : nullptr;
    constexpr bool masked = MASKED;

    __shared__ __attribute__((aligned(16))) u64 s_ring[4][64];       // pending candidates, one ring per wave
    u64* ring = s_ring[threadIdx.x >> 6];
    WaveList<SLOTS> L;
    L.clear();
    RingPending P;
    P.clear();
    u64 thr = 0;
    // (Measured and dropped in round 2: a float pre-filter "score >= k-th score" in front of the key build, per 256-element block
    // 0.354 -> 0.383 ms and per 64-element offer 0.354 -> 0.353 ms on 256 x 1M, k = 100: the key build is not what bounds the scan.
    // Round 3, the fused sink's float pre-compare (no key, no exact compare unless some lane reaches the k-th score) in front of every
    // offer, same-box A/B: 6,038 x 3,533 0.0308 -> 0.0326 ms, masked 6,040 x 3,706 0.0373 -> 0.0389, 2,048 x 65,536 0.2084 -> 0.2134,
    // 256 x 1M 0.2530 -> 0.2356: three of four shapes lose -- the extra branch per element costs more than the keys it saves.)
    if constexpr (VEC == 4) {
        constexpr int UN = SHORT ? (MASKED ? RECO_TOPK_UN_SHORT_MASKED : RECO_TOPK_UN_SHORT) : (MASKED ? RECO_TOPK_UN_MASKED : RECO_TOPK_UN);
        // the masks travel with the scores: mk > 0 marks an excluded element.  (Read where they are applied -- after the scores
        // have arrived -- the mask loads were an exposed round trip per block: 6,040 x 3,706 with the float rating matrix as
        // the mask took 0.080 ms against 0.040 ms without a mask.)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const bool sf2 = sf && a.ld_seen % 2 == 0 && (uintptr_t)a.seen % 8 == 0;      // rows 8-byte aligned: two float2 per 4 elements
        const bool ex4 = ex && a.ld_ex % 4 == 0 && (uintptr_t)a.exclude % 4 == 0;      // rows 4-byte aligned: one dword per 4 elements
        auto load = [&](int64_t base, f32x4* v, f32x4* mk) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t n0 = base + u * 256 + 4 * lane;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                const bool whole = n0 + 3 < s1;
                if (whole) v[u] = *reinterpret_cast<const f32x4*>(src + n0);
                else {
                    if (n0 + 0 < s1) v[u].x = src[n0 + 0];
                    if (n0 + 1 < s1) v[u].y = src[n0 + 1];
                    if (n0 + 2 < s1) v[u].z = src[n0 + 2];
                }
                if constexpr (!MASKED) continue;
                f32x4 m = {0.f, 0.f, 0.f, 0.f};
                if (sf) {
                    if (sf2 && whole) {
                        const f32x2 lo = *reinterpret_cast<const f32x2*>(sf + n0), hi = *reinterpret_cast<const f32x2*>(sf + n0 + 2);
                        m = f32x4{lo.x, lo.y, hi.x, hi.y};
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (n0 + t < s1) m[t] = sf[n0 + t];
                    }
                }
                if (ex) {
                    if (ex4 && whole) {
                        const uint32_t b = *reinterpret_cast<const uint32_t*>(ex + n0);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if ((b >> (8 * t)) & 0xffu) m[t] = 1.f;
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (n0 + t < s1 && ex[n0 + t]) m[t] = 1.f;
                    }
                }
                mk[u] = m;
            }
        };
        constexpr int UM = MASKED ? UN : 1;
        f32x4 v[UN], vn[UN], mk[UM], mkn[UM];
#pragma unroll
        for (int u = 0; u < UM; ++u) mk[u] = mkn[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        load(s0, v, mk);
        for (int64_t base = s0; base < s1; base += 256 * UN) {
            if (base + 256 * UN < s1) load(base + 256 * UN, vn, mkn);  // next iteration's loads fly under this one's compares
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int64_t n = base + u * 256 + 4 * lane + t;
                    const bool ok = n < s1;
                    float x = v[u][t];
                    if constexpr (MASKED) {
                        if (mk[u][t] > 0.f) x = x * 0.f - 1e10f;         // helper.py:313: pred*(~seen) - seen*1e10
                    }
                    const u64 key = ((u64)float_key(x) << 32) | (uint32_t)~(uint32_t)n;
                    offer_ring<SLOTS, RECO_TOPK_RING>(L, P, ring, key, ok, thr, a.k, lane);
                }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                v[u] = vn[u];
                if constexpr (MASKED) mk[u] = mkn[u];
            }
        }
    } else {
        for (int64_t base = s0; base < s1; base += 256) {
            float v[4];
            bool ok[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t n = base + t * 64 + lane;
                ok[t] = n < s1;
                v[t] = ok[t] ? src[n] : 0.f;
                if (masked && ok[t] && ((ex && ex[n]) || (sf && sf[n] > 0.f))) v[t] = v[t] * 0.f - 1e10f;   // helper.py:313
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t n = base + t * 64 + lane;
                const u64 key = ((u64)float_key(v[t]) << 32) | (uint32_t)~(uint32_t)n;
                offer_ring<SLOTS, RECO_TOPK_RING>(L, P, ring, key, ok[t], thr, a.k, lane);
            }
        }
    }
    flush_ring<SLOTS, RECO_TOPK_RING>(L, P, ring, thr, a.k, lane);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int r = s * 64 + lane;
        if (r < a.k) {
            if (a.splits > 1) {
                a.partial[(row * a.splits + split) * a.k + r] = L.e[s];
            } else {
                a.out_ids[row * a.k + r] = (int32_t)~(uint32_t)L.e[s];
                a.out_vals[row * a.k + r] = key_float((uint32_t)(L.e[s] >> 32));
            }
        }
    }
}

template <int SLOTS>
__global__ __launch_bounds__(256) void topk_merge_kernel(TopkArgs a)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    // every segment's list arrives sorted (rank order), so each 64-rank chunk of it is merged as it is: no re-sort, and a
    // chunk whose best key does not beat the current k-th key is skipped after one readlane
    const u64* src = a.partial + row * a.splits * a.k;
    WaveList<SLOTS> L;
    L.clear();
    u64 thr = 0;
    for (int sp = 0; sp < a.splits; ++sp)
        for (int c0 = 0; c0 < a.k; c0 += 64) {
            const u64 key = c0 + lane < a.k ? src[(int64_t)sp * a.k + c0 + lane] : 0;
            if (bcast(key, 0) <= thr) break;            // the rest of this segment is smaller still
            L.merge_sorted(key, lane);
            thr = L.kth(a.k);
        }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int r = s * 64 + lane;
        if (r < a.k) {
            a.out_ids[row * a.k + r] = (int32_t)~(uint32_t)L.e[s];
            a.out_vals[row * a.k + r] = key_float((uint32_t)(L.e[s] >> 32));
        }
    }
}

// Many short rows (the 99-negative protocol: 115,056 rows of 100 scores, ncf.ipynb cell 8): one LANE per row instead of one
// wave.  A wave-wide sort network costs ~500 instructions for such a row whatever its length; here a lane keeps its row's best K
// keys sorted in 2 K registers and inserts an element with K compares + 4 K selects, branch-free (~50 instructions per element
// and 64 rows per instruction).  Rows reach their lanes through an LDS transpose: the block's 256 rows are loaded 32 columns at
// a time with coalesced 128-byte row segments, stored [row][36] (conflict-free ds_read_b128 by the owning lane).
// Same order as the wave kernels: descending key, ascending position among equals (a later equal key never displaces).
template <int K, int VEC>
__global__ __launch_bounds__(256) void topk_lane_kernel(TopkArgs a)
{
    constexpr int SLAB = 32, LSTR = 36;
    __shared__ __attribute__((aligned(16))) float s_t[256 * LSTR];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 256;
    const int64_t row = r0 + tid;
    uint32_t lk[K];
    int32_t li[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { lk[j] = 0u; li[j] = -1; }      // 0 is below every real key (float_key(-inf) = 0x007fffff)

    // slab c0 + 32 is in flight into registers while slab c0 is consumed out of LDS
    constexpr int NV = VEC == 4 ? 8 : 32;
    float st[NV * VEC];
    auto slab_load = [&](int64_t c0) {
        if constexpr (VEC == 4) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int i = tid + 256 * s;
                const int64_t rr = r0 + (i >> 3), n = c0 + 4 * (i & 7);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (rr < a.rows) {
                    const float* src = a.scores + rr * a.ld + n;
                    if (n + 3 < a.N) v = *reinterpret_cast<const f32x4*>(src);
                    else {
                        if (n + 0 < a.N) v.x = src[0];
                        if (n + 1 < a.N) v.y = src[1];
                        if (n + 2 < a.N) v.z = src[2];
                    }
                }
                st[4 * s + 0] = v.x; st[4 * s + 1] = v.y; st[4 * s + 2] = v.z; st[4 * s + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int i = tid + 256 * s;
                const int64_t rr = r0 + (i >> 5), n = c0 + (i & 31);
                st[s] = (rr < a.rows && n < a.N) ? a.scores[rr * a.ld + n] : 0.f;
            }
        }
    };
    auto slab_store = [&]() {
        if constexpr (VEC == 4) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int i = tid + 256 * s;
                *reinterpret_cast<f32x4*>(s_t + (i >> 3) * LSTR + 4 * (i & 7)) = f32x4{st[4 * s], st[4 * s + 1], st[4 * s + 2], st[4 * s + 3]};
            }
        } else {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int i = tid + 256 * s;
                s_t[(i >> 5) * LSTR + (i & 31)] = st[s];
            }
        }
    };

    slab_load(0);
    for (int64_t c0 = 0; c0 < a.N; c0 += SLAB) {
        slab_store();
        __syncthreads();
        if (c0 + SLAB < a.N) slab_load(c0 + SLAB);
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            if (c0 + 4 * c4 >= a.N) break;                               // uniform: the row ends inside this slab
            const f32x4 v = *reinterpret_cast<const f32x4*>(s_t + tid * LSTR + 4 * c4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t n = c0 + 4 * c4 + t;
                const uint32_t key = n < a.N ? float_key(v[t]) : 0u;     // past the row's end: never enters
                const int32_t pos = (int32_t)n;
#pragma unroll
                for (int j = K - 1; j >= 1; --j) {
                    const bool here = key > lk[j], above = key > lk[j - 1];
                    lk[j] = above ? lk[j - 1] : (here ? key : lk[j]);
                    li[j] = above ? li[j - 1] : (here ? pos : li[j]);
                }
                const bool first = key > lk[0];
                lk[0] = first ? key : lk[0];
                li[0] = first ? pos : li[0];
            }
        }
        __syncthreads();
    }
    if (row < a.rows) {
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j < a.k) {
                a.out_ids[row * a.k + j] = li[j];
                a.out_vals[row * a.k + j] = key_float(lk[j]);
            }
    }
}

// Short rows with 16-byte aligned starts (the protocol's 100 scores per row): select, then order.  The insertion above spends
// 4 K selects per element on carrying the ids along; here a lane first finds the VALUES of the K best keys with one
// v_med3_u32 per list slot (new l[j] = med3(key, l[j-1], l[j]) is the sorted insertion of a key into a descending list), then
// marks the elements that reach the K-th key in a per-lane bit mask, and only those ~K elements go through the id-carrying
// insertion, in position order (so equal keys keep the ascending-position rule): ~3,200 instead of ~5,600 instructions per 64 rows.
// What that alone bought was 0.036 -> 0.034 ms: the wave timeline (tools/lane_trace.py) showed that ONE wave issues a VALU
// instruction every ~8 cycles, a SIMD one every 4 -- it takes two waves per SIMD to use the ALU -- and with whole rows in LDS
// (25.6 KB per wave) six waves fit a CU where the grid brings seven to ten.  So a row goes through LDS in SEGMENTS of 52 columns
// (13.3 KB per wave: every wave of the grid is resident at once): the key list runs on across segments, a segment's marked
// elements are those that reach the K-th key SO FAR, and they are inserted into the one (key, id) list -- the result is the
// top K of (top K of the earlier segments + this segment's entrants), in position order as before: 0.0356 -> 0.0253 ms
// (115,056 x 100, k = 10, same-box A/B); DESIGN section 3 K-d has the timeline.
// One wave per workgroup; its rows come in through LDS-direct loads (no staging registers, no barrier anywhere), 13 sixteen-byte
// slots per row and segment -- an odd stride, so a lane's float4 reads of its own row are conflict-free.  Keys are written back
// over the scores so the later passes skip float_key.
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

#ifdef RECO_DIAG_LANE_TRACE         // diagnostic builds only: per-wave start / end time and placement (tools/lane_trace.py)
static __device__ unsigned long long g_lane_trace[3 * 4096];
#endif

constexpr int LSEL_SLOTS = 13;                 // float4 slots per row and segment
constexpr int LSEL_COLS = 4 * LSEL_SLOTS;      // 52 columns

template <int K>
__global__ __launch_bounds__(64) void topk_lane_select_kernel(TopkArgs a)
{
    __shared__ __attribute__((aligned(16))) float s_rows[64 * LSEL_COLS];    // [64 rows][13 slots]
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifdef RECO_DIAG_LANE_TRACE
    const unsigned long long trace_t0 = wall_clock64();
#endif
    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int n4 = (int)(a.N >> 2);
    uint32_t* krow = reinterpret_cast<uint32_t*>(s_rows) + lane * LSEL_COLS;

    uint32_t lv[K];                  // the K best key values so far (pass 1)
    uint32_t lk[K];                  // the K best (key, id) so far (pass 3)
    int32_t li[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { lv[j] = 0u; lk[j] = 0u; li[j] = -1; }       // 0 is below every real key (float_key(-inf) = 0x007fffff)

    for (int c_lo = 0; c_lo < n4; c_lo += LSEL_SLOTS) {
        const int ns = n4 - c_lo < LSEL_SLOTS ? n4 - c_lo : LSEL_SLOTS;       // float4 slots of this segment
        {
            typedef const __attribute__((address_space(1))) void* gptr_t;
            typedef __attribute__((address_space(3))) void* lptr_t;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the previous segment's LDS reads are done
#pragma unroll
            for (int i = 0; i < LSEL_SLOTS; ++i) {
                const int p = i * 64 + lane;                                  // LDS slot p = (row, col): the lane's 16 bytes land there
                const int rr = (p * 5042) >> 16;                              // p / 13 for p < 6,570
                int col = p - LSEL_SLOTS * rr;
                if (col >= ns) col = ns - 1;                                  // slots past the row's end repeat the last one (never read)
                int64_t row = r0 + rr;
                if (row >= a.rows) row = a.rows - 1;                          // the ragged last workgroup: a valid row, result unused
                __builtin_amdgcn_global_load_lds((gptr_t)(a.scores + row * a.ld + 4 * (c_lo + col)), (lptr_t)(s_rows + i * 256), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }

        // pass 1: the K best key VALUES so far (with multiplicity), keys written back over the scores
        f32x4 cur = *reinterpret_cast<const f32x4*>(krow);
#pragma unroll
        for (int c = 0; c < LSEL_SLOTS; ++c) {
            if (c < ns) {                                                    // uniform
                f32x4 nxt = cur;
                if (c + 1 < ns) nxt = *reinterpret_cast<const f32x4*>(krow + 4 * (c + 1));
                u32x4 kk;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t key = float_key(cur[t]);
                    kk[t] = key;
#pragma unroll
                    for (int j = K - 1; j >= 1; --j) lv[j] = umed3(key, lv[j - 1], lv[j]);
                    lv[0] = key > lv[0] ? key : lv[0];
                }
                *reinterpret_cast<u32x4*>(krow + 4 * c) = kk;
                cur = nxt;
            }
        }
        uint32_t thr = lv[0];
#pragma unroll
        for (int j = 1; j < K; ++j) thr = (j < a.k) ? lv[j] : thr;           // the k-th best key so far

        // pass 2: which elements of this segment reach it.  A mask word is shifted left as it fills (one v_cmp + one add-with-carry
        // per element), so the EARLIEST element of a word sits in its highest bit: element e of the segment is bit 31 - e of m0
        // (e < 32) or bit 51 - e of m1
        uint32_t m0 = 0u, m1 = 0u;
#pragma unroll
        for (int c = 0; c < LSEL_SLOTS; ++c) {
            if (c < ns) {                                                    // uniform
                const u32x4 kk = *reinterpret_cast<const u32x4*>(krow + 4 * c);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (c < 8) m0 = m0 + m0 + (kk[t] >= thr ? 1u : 0u);
                    else m1 = m1 + m1 + (kk[t] >= thr ? 1u : 0u);
                }
            } else {
                if (c < 8) m0 <<= 4;
                else m1 <<= 4;
            }
        }

        // pass 3: the marked elements, in position order, through the id-carrying insertion (keys by v_med3, ids by selects)
        for (;;) {
            const uint32_t any = m0 | m1;
            if (__ballot(any != 0u) == 0ull) break;
            const uint32_t mw = m0 ? m0 : m1;
            const int lz = any ? __builtin_clz(mw) : 0;
            const int pl = m0 ? lz : 32 + lz - (64 - LSEL_COLS);             // position inside the segment (m1 holds 20 elements)
            const uint32_t clr = ~(0x80000000u >> lz);
            if (m0) m0 &= clr;
            else m1 &= any ? clr : 0u;
            const uint32_t key = any ? krow[any ? pl : 0] : 0u;              // 0 never enters
            const int32_t pos = 4 * c_lo + pl;
#pragma unroll
            for (int j = K - 1; j >= 1; --j) {
                const bool here = key > lk[j], above = key > lk[j - 1];
                li[j] = above ? li[j - 1] : (here ? pos : li[j]);
                lk[j] = umed3(key, lk[j - 1], lk[j]);
            }
            li[0] = key > lk[0] ? pos : li[0];
            lk[0] = key > lk[0] ? key : lk[0];
        }
    }
#ifdef RECO_DIAG_LANE_TRACE
    if (lane == 0 && blockIdx.x < 4096) {
        g_lane_trace[3 * blockIdx.x + 0] = trace_t0;
        g_lane_trace[3 * blockIdx.x + 1] = wall_clock64();
        g_lane_trace[3 * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |      // HW_ID
                                           ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);  // XCC_ID
    }
#endif
    const int64_t row = r0 + lane;
    if (row < a.rows) {
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j < a.k) {
                a.out_ids[row * a.k + j] = li[j];
                a.out_vals[row * a.k + j] = key_float(lk[j]);
            }
    }
}

int choose_splits(int64_t rows, int64_t N, int k, bool masked)
{
    // enough waves to cover the loads (3-4 per SIMD) while keeping segments long enough to amortise the list: every segment
    // pays its own k ln(segment / k) entrants, so fewer, longer segments do less work
    int64_t want = ceil_div64(masked ? RECO_TOPK_WAVES_MASKED : RECO_TOPK_WAVES, rows);
    int64_t max_by_len = N / 4096 > 0 ? N / 4096 : 1;
    int64_t s = want < max_by_len ? want : max_by_len;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    (void)k;
    return (int)s;
}

}  // namespace

// Folds `splits` sorted partial lists per row (keys as written by a scan segment or by the fused tile kernel; 0 = empty).
int launch_topk_merge(const unsigned long long* partial, int64_t rows, int splits, int k, int32_t* out_ids, float* out_vals, hipStream_t s)
{
    if (rows <= 0) return RECO_OK;
    RECO_REQUIRE(partial && out_ids && out_vals && splits >= 1 && k >= 1 && k <= 128, "topk merge: bad arguments");
    TopkArgs a{nullptr, rows, 0, 0, k, nullptr, 0, nullptr, 0, nullptr, rows, nullptr, splits, 0, const_cast<u64*>(partial), out_ids, out_vals};
    const int64_t mblocks = ceil_div64(rows, 4);
    RECO_REQUIRE(mblocks < (1ll << 31), "topk merge grid too large");
    if (k <= 64) hipLaunchKernelGGL(topk_merge_kernel<1>, dim3((unsigned)mblocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(topk_merge_kernel<2>, dim3((unsigned)mblocks), dim3(256), 0, s, a);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

size_t topk_workspace_bytes(int64_t rows, int64_t N, int k)
{
    if (rows <= 0 || N <= 0 || k <= 0) return 0;
    const int a = choose_splits(rows, N, k, false), b = choose_splits(rows, N, k, true);
    const int splits = a > b ? a : b;                 // the caller may or may not bring a mask
    return splits > 1 ? align_up((size_t)rows * splits * k * sizeof(u64), 256) : 0;
}

int launch_topk_rows(const float* scores, int64_t rows, int64_t N, int64_t ld, int k,
                     const uint8_t* exclude, int64_t ld_ex, int32_t* out_ids, float* out_vals,
                     void* ws, size_t ws_bytes, hipStream_t s)
{
    return launch_topk_rows_ex(scores, rows, N, ld, k, exclude, ld_ex, nullptr, 0, nullptr, rows, nullptr, out_ids, out_vals,
                               ws, ws_bytes, s);
}

int launch_topk_rows_ex(const float* scores, int64_t rows, int64_t N, int64_t ld, int k,
                        const uint8_t* exclude, int64_t ld_ex, const float* seen, int64_t ld_seen,
                        const int64_t* row_ids, int64_t src_rows, int32_t* oob, int32_t* out_ids, float* out_vals,
                        void* ws, size_t ws_bytes, hipStream_t s)
{
    if (rows <= 0) return RECO_OK;
    RECO_REQUIRE(k >= 1 && k <= 128, "topk: k must be in [1,128], got %d", k);
    RECO_REQUIRE(N >= k, "topk: k=%d exceeds the number of candidates %lld", k, (long long)N);
    RECO_REQUIRE(N < (1ll << 32) - 1, "topk: row too long");
    RECO_REQUIRE(scores && out_ids && out_vals, "topk: null pointer");
    RECO_REQUIRE(!seen || ld_seen >= N, "topk: ld_seen smaller than N");
    RECO_REQUIRE(!row_ids || src_rows > 0, "topk: row ids given but the source has no rows");
    TopkArgs a{scores, rows, N, ld, k, exclude, ld_ex, seen, ld_seen, row_ids, src_rows, oob, 1, 0, nullptr, out_ids, out_vals};
    if (!exclude && !seen && !row_ids && k <= 16 && N <= 512 && rows >= 32768) {        // many short rows: a lane per row (topk_lane_kernel)
        const int64_t lblocks = ceil_div64(rows, 256);
        RECO_REQUIRE(lblocks < (1ll << 31), "topk grid too large");
        const bool v4 = ld % 4 == 0 && (uintptr_t)scores % 16 == 0;
        ProfScope prof(s, RECO_PROF_TOPK);
        if (RECO_TOPK_LANE_SELECT && v4 && N % 4 == 0 && N <= 160) {         // rows of float4s: select-then-order, <= 4 segments
            const int64_t sblocks = ceil_div64(rows, 64);
            RECO_REQUIRE(sblocks < (1ll << 31), "topk grid too large");
            if (k <= 10) hipLaunchKernelGGL(topk_lane_select_kernel<10>, dim3((unsigned)sblocks), dim3(64), 0, s, a);
            else hipLaunchKernelGGL(topk_lane_select_kernel<16>, dim3((unsigned)sblocks), dim3(64), 0, s, a);
            RECO_CHECK_LAUNCH();
            return RECO_OK;
        }
        if (k <= 10) {
            if (v4) hipLaunchKernelGGL((topk_lane_kernel<10, 4>), dim3((unsigned)lblocks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((topk_lane_kernel<10, 1>), dim3((unsigned)lblocks), dim3(256), 0, s, a);
        } else {
            if (v4) hipLaunchKernelGGL((topk_lane_kernel<16, 4>), dim3((unsigned)lblocks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((topk_lane_kernel<16, 1>), dim3((unsigned)lblocks), dim3(256), 0, s, a);
        }
        RECO_CHECK_LAUNCH();
        return RECO_OK;
    }
    a.splits = choose_splits(rows, N, k, exclude != nullptr || seen != nullptr);
    if (a.splits > 1) {
        const size_t need = (size_t)rows * a.splits * k * sizeof(u64);
        if (ws == nullptr || ws_bytes < need) {
            reco_set_error("topk: workspace too small (%zu < %zu bytes)", ws_bytes, need);
            return RECO_ENOMEM;
        }
        a.partial = static_cast<u64*>(ws);
    }
    a.seg = ceil_div64(ceil_div64(N, a.splits), 64) * 64;
    const int64_t waves = rows * a.splits;
    const int64_t blocks = ceil_div64(waves, 4);
    RECO_REQUIRE(blocks < (1ll << 31), "topk grid too large");
    const bool vec4 = N >= 1024 && ld % 4 == 0 && (uintptr_t)scores % 16 == 0;      // segment starts are multiples of 64 elements;
    // short rows keep whole 64-element chunks per offer (N = 100: 0.10 ms scalar vs 0.16 ms with quarter-filled chunks)
    const bool masked = exclude != nullptr || seen != nullptr;
    const dim3 grid((unsigned)blocks), block(256);
    const bool shortseg = vec4 && a.seg <= RECO_TOPK_SHORT_SEG;      // catalogue-length rows: fewer loads in flight, less tail waste
#define RECO_TOPK_LAUNCH(SLOTS, VEC)                                                                        \
    do {                                                                                                    \
        if (shortseg && VEC == 4) {                                                                         \
            if (masked) hipLaunchKernelGGL((topk_scan_kernel<SLOTS, VEC, true, VEC == 4>), grid, block, 0, s, a);    \
            else hipLaunchKernelGGL((topk_scan_kernel<SLOTS, VEC, false, VEC == 4>), grid, block, 0, s, a);          \
        } else {                                                                                            \
            if (masked) hipLaunchKernelGGL((topk_scan_kernel<SLOTS, VEC, true, false>), grid, block, 0, s, a);       \
            else hipLaunchKernelGGL((topk_scan_kernel<SLOTS, VEC, false, false>), grid, block, 0, s, a);             \
        }                                                                                                   \
    } while (0)
    {
        ProfScope prof(s, RECO_PROF_TOPK);
        if (k <= 64) {
            if (vec4) RECO_TOPK_LAUNCH(1, 4);
            else RECO_TOPK_LAUNCH(1, 1);
        } else {
            if (vec4) RECO_TOPK_LAUNCH(2, 4);
            else RECO_TOPK_LAUNCH(2, 1);
        }
    }
#undef RECO_TOPK_LAUNCH
    RECO_CHECK_LAUNCH();
    if (a.splits > 1) {
        const int64_t mblocks = ceil_div64(rows, 4);
        ProfScope prof(s, RECO_PROF_TOPK_MERGE);
        if (k <= 64)
            hipLaunchKernelGGL(topk_merge_kernel<1>, dim3((unsigned)mblocks), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(topk_merge_kernel<2>, dim3((unsigned)mblocks), dim3(256), 0, s, a);
        RECO_CHECK_LAUNCH();
    }
    return RECO_OK;
}

#ifdef RECO_DIAG_LANE_TRACE
extern "C" int reco_debug_lane_trace(unsigned long long* host_out, int words)
{
    if (words > 3 * 4096) words = 3 * 4096;
    RECO_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_lane_trace), (size_t)words * sizeof(unsigned long long)));
    return RECO_OK;
}
#endif
