// Pieces shared by the two tower translation units (tower.hip: exact fp32 MFMA; tower_split.hip: fp32 operands split
// into bf16 pieces on the bf16 MFMA): argument blocks, LDS staging of tiles and pair groups, small device helpers.
#pragma once
#include "reco_common.h"

namespace {

// ReLU as ONE VALU instruction.  fmaxf() on an MFMA result makes hipcc emit a canonicalising v_max in
// front of the real one (2 instructions), and beside f32 MFMAs every VALU instruction costs MFMA issue
// time.  max on the raw bits is exact: negative floats (and -0.0) are negative integers -> +0.0,
// non-negative floats are unchanged.  (A positive NaN stays NaN, a negative NaN becomes 0.)
__device__ __forceinline__ float relu1(float v)
{
    return __int_as_float(max(__float_as_int(v), 0));
}

__device__ __forceinline__ f32x4 relu4(f32x4 v)
{
    f32x4 r;
    r.x = relu1(v.x); r.y = relu1(v.y); r.z = relu1(v.z); r.w = relu1(v.w);
    return r;
}

__device__ __forceinline__ f32x16 bias16(const float* s_b, int h)
{
    // accumulator register r = 4 rq + jj  <-  image [rq][h][jj]
    f32x16 c;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(s_b + (rq * 2 + h) * 4);
        c[4 * rq + 0] = v.x; c[4 * rq + 1] = v.y; c[4 * rq + 2] = v.z; c[4 * rq + 3] = v.w;
    }
    return c;
}

__device__ __forceinline__ float dot16_relu(const f32x16 acc, const float* s_w, int h, float part)
{
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(s_w + (rq * 2 + h) * 4);
        part = fmaf(w.x, relu1(acc[4 * rq + 0]), part);
        part = fmaf(w.y, relu1(acc[4 * rq + 1]), part);
        part = fmaf(w.z, relu1(acc[4 * rq + 2]), part);
        part = fmaf(w.w, relu1(acc[4 * rq + 3]), part);
    }
    return part;
}

__device__ __forceinline__ float epilogue(float logit, int act, float lo, float hi)
{
    if (act == RECO_ACT_RATING) return (1.f / (1.f + __expf(-logit))) * (hi - lo) + lo;
    return logit;
}

__device__ __forceinline__ int64_t clamp_id(int64_t id, int64_t n, int32_t* oob)
{
    if (id < 0 || id >= n) {
        if (oob) atomicAdd(oob, 1);
        return id < 0 ? 0 : n - 1;
    }
    return id;
}

// ---------------------------------------------------------------------------------------------
// tile kernels: scores for (user chunk) x (item tile); persistent, one workgroup per CU
// ---------------------------------------------------------------------------------------------
// Each workgroup loads the weight image once and then takes tiles from the work queue below.
// A tile is TUB users x TI items.  The item side of a tile (Q rows, GMF item rows) lives in one of
// two LDS buffers: while the waves run the MFMAs of tile i out of buffer i & 1, the global loads of
// tile i + 1 are in flight into registers, are written to the other buffer after the last MFMA and
// published by the single __syncthreads() per tile.  Wave w owns users w, w + (waves), ... of the tile.
// The exact kernel (tower.hip) stages the tile's P rows in LDS the same way and runs 12 or 8 waves with its own
// THREADS / users-per-tile; the split kernel (tower_split.hip) uses the 512-thread / 32-user defaults below and keeps a
// user's P row in 64 VGPRs across the tile's item groups (its 62-KB weight image leaves no LDS for a P tile).
constexpr int TI = 64;         // items per tile (2 MFMA column groups)
#ifndef RECO_TUB
#define RECO_TUB 32
#endif
constexpr int TUB = RECO_TUB;   // users per tile (4 per wave) -- default, see above
constexpr int QS = H1 + 4;     // padded LDS row stride of the Q tile: conflict-free ds_read_b128 by 16 rows
constexpr int TILE_THREADS = 512;

struct TileArgs {
    const float* img;
    const float* P;        // [U,128]  row u <-> users[u]
    const float* Q;        // [N,128]  row n <-> cands[n]
    const float* EUg;      // GMF user table
    const float* EIg;      // GMF item table
    const int64_t* users;  // nullable
    const int64_t* cands;  // nullable
    int64_t user_base, n_users, n_items;
    int64_t U, N;
    float* out;
    int64_t ld_out;
    float bp;
    int act;
    float lo, hi;
    int32_t* oob;
    unsigned int* next_tile;   // work queue: tiles beyond the first gridDim.x are handed out through this counter (zeroed
                               // before the launch); null = static round-robin
    const float* l1;           // exact kernels: device words {s, 1/s}, the power-of-two scale P and Q were written with (below)
    int reserve_cus;           // host side only: the persistent grid leaves this many CUs without a workgroup (tile_grid)
};

// Persistent tile kernels launch one workgroup per CU.  A handle can keep some CUs free (reco_tower_set_reserved_cus): the tile
// kernels take 152 of a CU's 160 KB of LDS, so a kernel of another stream -- RCCL's all-gather of the previous step's lists --
// otherwise finds no CU until a tile launch ends.  Work is handed out through a queue, so fewer workgroups only take more tiles each.
inline unsigned tile_grid(int64_t tiles, int n_cu, int reserve_cus)
{
    int n = n_cu - (reserve_cus > 0 ? reserve_cus : 0);
    if (n < 1) n = 1;
    return (unsigned)(tiles < n ? tiles : n);
}

// ---- layer 1 as ONE instruction per element -------------------------------------------------------------------------------
// x1 = relu(p + q) is an add and a max, 128 of the ~236 VALU instructions a wave issues per 160 MFMAs -- and beside f32 MFMAs
// every VALU instruction costs MFMA issue time.  v_add_f32 with the CLAMP modifier is add + clamp to [0, 1] in one instruction,
// and clamp IS relu when nothing exceeds 1.  So the exact kernels work on s * x1 with s a power of two such that
// s * (p + q) < 1 for every pair the handle (or the call) can form: the projection GEMMs write P and Q multiplied by s
// (gemm.hip out_scale), the biases b2 / b3 are multiplied by s and the MLP half of the predict weights by 1/s when the weight
// image is copied into LDS.  Scaling by a power of two commutes with every rounding on the way (and with relu), so the scores
// are bit-identical to the unscaled computation, short of terms ~2^-100 below the largest possible activation underflowing.
// s comes from a bound, not from the data of one call: |p_f| <= |W1u_f|_1 max|user rows| + |b1_f|, |q_f| <= |W1i_f|_1 max|item
// rows| (l1scale_kernel: once per handle over its tables, per call when the rows are handed in).  Non-finite table entries are
// left out of the maxima: a pair that touches one gets a finite but meaningless score where the reference has inf / NaN;
// every other pair is unaffected.  Measured: cfg2 tile kernel 3.245 -> 3.155 ms with the add alone.
struct Layer1Scale { float s, inv; };

__device__ __forceinline__ Layer1Scale layer1_scale(const float* l1) { return Layer1Scale{l1[0], l1[1]}; }

// {s, 1/s} for a bound b >= 0 on p + q: s = 2^-k with 2^k > b
__device__ __forceinline__ Layer1Scale layer1_scale_for(float b)
{
    int e = (int)((__float_as_uint(b) >> 23) & 255u);        // b < 2^(e - 126)
    int se = 253 - e;                                        // biased exponent of s = 2^-(e - 126)
    // s <= 2^24: a smaller s than the bound allows is always safe for the clamp, and tables of zeros / 1e-30 must not push
    // s * b2 to overflow or w_predict / s into the denormals
    // s >= 2^-80: one absurd row (|p + q| beyond 2^80, i.e. 1e24) must not push every OTHER row's scaled activations, s * b2 and
    // their products towards the denormals; such a row's own activations saturate at 2^80 instead (a finite, meaningless score
    // for the pairs that touch it, like the non-finite entries above; every other pair stays exact)
    se = se < 127 - 80 ? 127 - 80 : (se > 127 + 24 ? 127 + 24 : se);
    return Layer1Scale{__uint_as_float((unsigned)se << 23), __uint_as_float((unsigned)(254 - se) << 23)};
}

__device__ __forceinline__ f32x4 add_clamp4(f32x4 p, f32x4 q)   // clamp(p + q, 0, 1) per element, one instruction each
{
    f32x4 r;
    asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(r.x) : "v"(p.x), "v"(q.x));
    asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(r.y) : "v"(p.y), "v"(q.y));
    asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(r.z) : "v"(p.z), "v"(q.z));
    asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(r.w) : "v"(p.w), "v"(q.w));
    return r;
}

// the weight image into LDS (tower without a third layer: that block is left out), biases and MLP predict weights rescaled
template <bool L3, int THREADS>
__device__ __forceinline__ void load_image_scaled(float* s_img, const float* img, int tid, Layer1Scale sc)
{
    for (int i = tid; i < IMG_SIZE / 4; i += THREADS) {
        if (!(L3 || i < IMG_W3 / 4 || i >= IMG_B2 / 4)) continue;
        f32x4 v = reinterpret_cast<const f32x4*>(img)[i];
        if (i >= IMG_B2 / 4 && i < IMG_WPM / 4) v = v * sc.s;            // b2, b3
        else if (i >= IMG_WPM / 4 && i < IMG_WPG / 4) v = v * sc.inv;    // predict weights over the (scaled) last MLP layer
        reinterpret_cast<f32x4*>(s_img)[lds_img<L3>(4 * i) / 4] = v;
    }
}

// Tile hand-out for the persistent tile kernels.  Workgroup b starts on tile b; every further tile comes from an atomic
// counter, so a workgroup slowed down by a co-resident kernel (the RCCL gather of the previous step) or by a ragged
// tile simply takes fewer tiles.  The index for the tile after next is fetched by thread 0 at the top of an iteration
// and published through one of two LDS slots behind the iteration's only barrier.
__device__ __forceinline__ void tile_queue_fetch(const TileArgs& a, int tid, int64_t fallback, long long* slot)
{
    if (tid == 0) *slot = a.next_tile ? (long long)gridDim.x + (long long)atomicAdd(a.next_tile, 1u) : (long long)fallback;
}

template <int G, int THREADS = TILE_THREADS>
struct TileStage {
    static constexpr int NQ = (TI * (H1 / 4) + THREADS - 1) / THREADS;      // float4 of Q per thread (4; 3 at 768 threads)
    static constexpr int NG = G > 0 ? (TI * (G / 4) + THREADS - 1) / THREADS : 0;
    f32x4 q[NQ];
    f32x4 g[NG > 0 ? NG : 1];
};

template <int G, int THREADS = TILE_THREADS>
__device__ __forceinline__ void tile_stage_load(const TileArgs& a, int64_t n0, bool count_oob, int tid, TileStage<G, THREADS>& st)
{
#pragma unroll
    for (int s = 0; s < TileStage<G, THREADS>::NQ; ++s) {
        const int i = tid + s * THREADS;
        const int64_t n = n0 + (i >> 5);
        st.q[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < TI * (H1 / 4) && n < a.N) st.q[s] = reinterpret_cast<const f32x4*>(a.Q + n * H1)[i & 31];
    }
    if constexpr (G > 0) {
#pragma unroll
        for (int s = 0; s < TileStage<G, THREADS>::NG; ++s) {
            const int i = tid + s * THREADS;
            const int r = i / (G / 4), c = i % (G / 4);
            const int64_t n = n0 + r;
            st.g[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < TI && n < a.N) {
                const int64_t id = clamp_id(a.cands ? a.cands[n] : n, a.n_items, (c == 0 && count_oob) ? a.oob : nullptr);
                st.g[s] = reinterpret_cast<const f32x4*>(a.EIg + id * G)[c];
            }
        }
    }
}

template <int G, int THREADS = TILE_THREADS>
__device__ __forceinline__ void tile_stage_store(float* s_q, float* s_g, int tid, const TileStage<G, THREADS>& st, float q_scale = 1.f)
{
#pragma unroll
    for (int s = 0; s < TileStage<G, THREADS>::NQ; ++s) {
        const int i = tid + s * THREADS;
        if (i < TI * (H1 / 4)) *reinterpret_cast<f32x4*>(s_q + (i >> 5) * QS + (i & 31) * 4) = st.q[s] * q_scale;
    }
    if constexpr (G > 0) {
#pragma unroll
        for (int s = 0; s < TileStage<G, THREADS>::NG; ++s) {
            const int i = tid + s * THREADS;
            const int r = i / (G / 4), c = i % (G / 4);
            if (r < TI) *reinterpret_cast<f32x4*>(s_g + r * (G + 4) + c * 4) = st.g[s];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pair kernel: forward(user[B], item[B])
// ---------------------------------------------------------------------------------------------
struct PairArgs {
    const float* img;
    const float* P;        // [n_users,128]
    const float* Q;        // [n_items,128]
    const float* EUg;
    const float* EIg;
    const int64_t* user;   // nullable: identity (row b of P / EUg)
    const int64_t* item;   // nullable: identity
    int64_t n_users, n_items, B;
    float* out;
    float bp;
    int act;
    float lo, hi;
    int32_t* oob;
    const float* l1;       // exact kernel: device words {s, 1/s} (TileArgs::l1)
};

// Per wave: 32 pairs at a time, fully wave-private (no workgroup barrier in the loop).  The P/Q/GMF rows of the NEXT 32
// pairs are in flight into registers (two whole 512-B rows per wave instruction) while the MFMAs of the current 32 run
// out of the wave's LDS staging tile; LDS operations of one wave execute in issue order, so staging writes and operand
// reads need no barrier between them.
template <int G>
struct PairStage {
    static constexpr int NG = G > 0 ? (32 * (G / 4)) / 64 : 0;   // float4 of each GMF table per lane
    f32x4 p[16], q[16];
    f32x4 gu[NG > 0 ? NG : 1], gi[NG > 0 ? NG : 1];
};

template <int G>
__device__ __forceinline__ void pair_stage_load(const PairArgs& a, int64_t base, int lane, PairStage<G>& st)
{
    const int j = lane & 31, h = lane >> 5;
    int64_t my_u = 0, my_i = 0;   // ids of the 32 pairs live in lanes 0..31 (mirrored in 32..63)
    const int64_t b = base + j;
    if (b < a.B) {
        my_u = clamp_id(a.user ? a.user[b] : b, a.n_users, h == 0 ? a.oob : nullptr);
        my_i = clamp_id(a.item ? a.item[b] : b, a.n_items, h == 0 ? a.oob : nullptr);
    }
    // Pair lists often come in runs of one user (the 99-negative protocol: 100 candidates per user, inference_engine
    // predict_batch: one user): when all 32 pairs of the group share their user, the user's rows are loaded ONCE (one float4
    // per lane) instead of 32 times -- 40 % of the group's L2 traffic.  Wave-uniform test, identical results.
    const int64_t u_first = __shfl(my_u, 0);
    const bool one_user = __ballot(my_u != u_first) == 0;
    if (one_user) {
        const f32x4 prow = reinterpret_cast<const f32x4*>(a.P + u_first * H1)[j];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            st.p[it] = prow;
            st.q[it] = reinterpret_cast<const f32x4*>(a.Q + __shfl(my_i, it * 2 + h) * H1)[j];
        }
    } else {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int pl = it * 2 + h;
            const int64_t u = __shfl(my_u, pl);
            const int64_t i = __shfl(my_i, pl);
            st.p[it] = reinterpret_cast<const f32x4*>(a.P + u * H1)[j];
            st.q[it] = reinterpret_cast<const f32x4*>(a.Q + i * H1)[j];
        }
    }
    if constexpr (G > 0) {
        constexpr int C = G / 4;
        static_assert(64 % C == 0, "a lane keeps one float4 column of the GMF rows across the stage registers");
        if (one_user) {
            const f32x4 grow = reinterpret_cast<const f32x4*>(a.EUg + u_first * G)[lane % C];
#pragma unroll
            for (int it = 0; it < PairStage<G>::NG; ++it) {
                const int e = it * 64 + lane;
                st.gu[it] = grow;                                   // e % C == lane % C: 64 is a multiple of C
                st.gi[it] = reinterpret_cast<const f32x4*>(a.EIg + __shfl(my_i, e / C) * G)[e % C];
            }
        } else {
#pragma unroll
            for (int it = 0; it < PairStage<G>::NG; ++it) {
                const int e = it * 64 + lane;
                const int pl = e / C, c = e % C;
                const int64_t u = __shfl(my_u, pl);
                const int64_t i = __shfl(my_i, pl);
                st.gu[it] = reinterpret_cast<const f32x4*>(a.EUg + u * G)[c];
                st.gi[it] = reinterpret_cast<const f32x4*>(a.EIg + i * G)[c];
            }
        }
    }
}

template <int G>
__device__ __forceinline__ void pair_stage_store(float* s_x, float* s_gp, int lane, const PairStage<G>& st)
{
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int it = 0; it < 16; ++it)
        *reinterpret_cast<f32x4*>(s_x + (it * 2 + h) * QS + 4 * j) = relu4(st.p[it] + st.q[it]);
    if constexpr (G > 0) {
        constexpr int C = G / 4;
#pragma unroll
        for (int it = 0; it < PairStage<G>::NG; ++it) {
            const int e = it * 64 + lane;
            *reinterpret_cast<f32x4*>(s_gp + (e / C) * (G + 4) + 4 * (e % C)) = st.gu[it] * st.gi[it];
        }
    }
}

// ---- half-row staging for the exact pair kernel -------------------------------------------------------------------------
// The layer-1 activations of a group are staged one K half (64 of 128 features) at a time: [32 pairs][HS] per wave, 8.7 KB
// instead of 16.9 KB, so that 8 waves share one weight image in LDS (2 waves per SIMD).  Register group `it` (0..7) of a half
// holds float4 column c = lane & 15 of rows 4 it + (lane >> 4): a wave instruction reads 4 rows x 256 contiguous bytes.
constexpr int HS = 64 + 4;     // padded LDS row stride of a half row: conflict-free ds_read_b128 by 8 rows

// ids are carried as int32 (tables have fewer than 2^31 rows, checked in reco_tower_create): one ds_bpermute per shuffle
__device__ __forceinline__ void pair_ids(const PairArgs& a, int64_t base, int lane, int& my_u, int& my_i)
{
    const int j = lane & 31, h = lane >> 5;
    my_u = 0; my_i = 0;           // ids of the 32 pairs live in lanes 0..31 (mirrored in 32..63)
    const int64_t b = base + j;
    if (b < a.B) {
        my_u = (int)clamp_id(a.user ? a.user[b] : b, a.n_users, h == 0 ? a.oob : nullptr);
        my_i = (int)clamp_id(a.item ? a.item[b] : b, a.n_items, h == 0 ? a.oob : nullptr);
    }
}

// the same in two steps, so that the load can be issued a whole group before its result is needed
__device__ __forceinline__ void pair_ids_raw(const PairArgs& a, int64_t base, int lane, int64_t& ru, int64_t& ri)
{
    const int64_t b = base + (lane & 31);
    ru = b; ri = b;                                   // identity when no id list is given; also the value past the end (clamped later)
    if (b < a.B) {
        if (a.user) ru = a.user[b];
        if (a.item) ri = a.item[b];
    }
}

__device__ __forceinline__ void pair_ids_clamp(const PairArgs& a, int64_t base, int lane, int64_t ru, int64_t ri, int& my_u, int& my_i)
{
    const int j = lane & 31, h = lane >> 5;
    my_u = 0; my_i = 0;
    if (base + j < a.B) {
        my_u = (int)clamp_id(ru, a.n_users, h == 0 ? a.oob : nullptr);
        my_i = (int)clamp_id(ri, a.n_items, h == 0 ? a.oob : nullptr);
    }
}

// Row addresses.  BIG = false (every table has fewer than 2^23 rows, i.e. under 4 GiB of 512-byte rows): a row's byte offset
// fits 32 bits, so a load is "uniform 64-bit base + 32-bit lane offset" -- the saddr form of global_load, one v_mad_u32 per load
// instead of a sign extension, a 64-bit shift and a 64-bit add (about 100 VALU instructions per group of 32 pairs, each of which
// costs MFMA issue time).  BIG = true keeps full 64-bit offsets (the 10 M-user tables of cfg5).
template <bool BIG>
__device__ __forceinline__ const f32x4* row_chunk(const float* table, int row, int row_floats, int chunk)
{
    if constexpr (BIG) return reinterpret_cast<const f32x4*>(table + (int64_t)row * row_floats) + chunk;
    else {
        const uint32_t off = (uint32_t)row * (uint32_t)(row_floats * 4) + (uint32_t)(chunk * 16);
        return reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(table) + off);
    }
}

template <bool BIG>
__device__ __forceinline__ void pair_load_half(const PairArgs& a, int my_u, int my_i, int lane, int half, f32x4* p, f32x4* q)
{
    const int c = 16 * half + (lane & 15), r = lane >> 4;
    const int u_first = __shfl(my_u, 0);
    if (__ballot(my_u != u_first) == 0) {            // the whole group is one user: its row once (see pair_stage_load)
        const f32x4 prow = *row_chunk<BIG>(a.P, u_first, H1, c);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            p[it] = prow;       // (keeping only one copy and branching again at the store was tried: the partially written
                                //  array stops being promoted to registers and the kernel takes 80 bytes of scratch per lane)
            q[it] = *row_chunk<BIG>(a.Q, __shfl(my_i, 4 * it + r), H1, c);
        }
    } else {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            p[it] = *row_chunk<BIG>(a.P, __shfl(my_u, 4 * it + r), H1, c);
            q[it] = *row_chunk<BIG>(a.Q, __shfl(my_i, 4 * it + r), H1, c);
        }
    }
}

template <bool CLAMP>
__device__ __forceinline__ void pair_store_half(float* s_x, int lane, const f32x4* p, const f32x4* q)
{
    const int c = lane & 15, r = lane >> 4;
#pragma unroll
    for (int it = 0; it < 8; ++it)     // CLAMP: = s * relu(p + q), P / Q pre-scaled
        *reinterpret_cast<f32x4*>(s_x + (4 * it + r) * HS + 4 * c) = CLAMP ? add_clamp4(p[it], q[it]) : relu4(p[it] + q[it]);
}

// ---- quarter-row staging (tower_pairs_q_kernel) --------------------------------------------------------------
// One K quarter (32 of 128 features) of P and Q per register set: a wave instruction reads 8 rows x 128 contiguous bytes,
// register group `it` (0..3) holds float4 column c = lane & 7 of rows 8 it + (lane >> 3).  Staged as [32 pairs][QSP] per wave.
constexpr int QSP = 32 + 4;

template <bool BIG>
__device__ __forceinline__ void pair_load_quarter(const PairArgs& a, int my_u, int my_i, int lane, int quarter, f32x4* p, f32x4* q)
{
    const int c = lane & 7, r = lane >> 3;
    const float* Pq = a.P + 32 * quarter;          // the quarter is part of the uniform base (see pair_load_half)
    const float* Qq = a.Q + 32 * quarter;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        p[it] = *row_chunk<BIG>(Pq, __shfl(my_u, 8 * it + r), H1, c);
        q[it] = *row_chunk<BIG>(Qq, __shfl(my_i, 8 * it + r), H1, c);
    }
}

template <bool CLAMP>
__device__ __forceinline__ void pair_store_quarter(float* s_x, int lane, const f32x4* p, const f32x4* q)
{
    const int c = lane & 7, r = lane >> 3;
#pragma unroll
    for (int it = 0; it < 4; ++it)
        *reinterpret_cast<f32x4*>(s_x + (8 * it + r) * QSP + 4 * c) = CLAMP ? add_clamp4(p[it], q[it]) : relu4(p[it] + q[it]);
}

template <int G, bool BIG>
__device__ __forceinline__ void pair_load_gmf(const PairArgs& a, int my_u, int my_i, int lane, f32x4* gu, f32x4* gi)
{
    constexpr int C = G / 4, NG = (32 * C) / 64;
    static_assert(64 % C == 0, "a lane keeps one float4 column of the GMF rows across the stage registers");
#pragma unroll
    for (int it = 0; it < NG; ++it) {
        const int e = it * 64 + lane;
        gu[it] = *row_chunk<BIG>(a.EUg, __shfl(my_u, e / C), G, e % C);
        gi[it] = *row_chunk<BIG>(a.EIg, __shfl(my_i, e / C), G, e % C);
    }
}

// ---- the GMF half of the predict layer, summed the SAME way in every exact kernel -----------------------------------------
// so that forward(u, i) (pair kernels) and score_matrix()[u, i] / score_topk (tile kernels) return the same bits:
//   per float4 column c of the G-wide rows:  s_c = fma(uw.w, i.w, fma(uw.z, i.z, fma(uw.y, i.y, uw.x * i.x))),  uw = u (*) w_predict[:G]
//   per half h (lane >> 5) of the row:       gmf_h = ((s_c0 + s_c0+1) + ...) + s_c0+G/8-1   in column order
// Tile kernels hold uw per user (one multiply per user, not per pair) and read the item row from LDS; pair kernels compute s_c in
// the lane that still holds column c of both rows in its staging registers and stage ONE float per (pair, column) -- [32][G/4]
// floats per wave instead of the [32][G+4] products -- which the MFMA body's lane adds up.
__device__ __forceinline__ float gmf_col(f32x4 uw, f32x4 i)
{
    float s = uw.x * i.x;
    s = fmaf(uw.y, i.y, s);
    s = fmaf(uw.z, i.z, s);
    return fmaf(uw.w, i.w, s);
}

// tile kernels: grow = this pair's item row half in LDS, gu = the user's row half times the predict weights
template <int G>
__device__ __forceinline__ float gmf_half_tile(const float* grow, const f32x4* gu)
{
    float part = 0.f;
    if constexpr (G > 0) {
        part = gmf_col(gu[0], *reinterpret_cast<const f32x4*>(grow));
#pragma unroll
        for (int c = 1; c < G / 8; ++c) part = part + gmf_col(gu[c], *reinterpret_cast<const f32x4*>(grow + 4 * c));
    }
    return part;
}

// pair kernels, staging side: lane holds float4 column e % C of rows e / C (e = it * 64 + lane) of both tables; wpg = predict weights
// of the GMF half (LDS).  s_gs [32][C] floats.
template <int G>
__device__ __forceinline__ void pair_store_gmf_cols(float* s_gs, int lane, const f32x4* gu, const f32x4* gi, const float* wpg)
{
    constexpr int C = G / 4, NG = (32 * C) / 64;
    const f32x4 w = *reinterpret_cast<const f32x4*>(wpg + 4 * (lane % C));       // e % C == lane % C: 64 is a multiple of C
#pragma unroll
    for (int it = 0; it < NG; ++it) s_gs[it * 64 + lane] = gmf_col(gu[it] * w, gi[it]);
}

// pair kernels, MFMA-body side: pair j's half h
template <int G>
__device__ __forceinline__ float gmf_half_pair(const float* s_gs, int j, int h)
{
    constexpr int C = G / 4;
    static_assert((C / 2) % 4 == 0, "a half row's column sums are read as float4");
    const f32x4* r = reinterpret_cast<const f32x4*>(s_gs + j * C + h * (C / 2));
    f32x4 v = r[0];
    float part = ((v.x + v.y) + v.z) + v.w;
#pragma unroll
    for (int n = 1; n < C / 8; ++n) {
        v = r[n];
        part = (((part + v.x) + v.y) + v.z) + v.w;
    }
    return part;
}

template <int G>
__device__ __forceinline__ void pair_store_gmf(float* s_gp, int lane, const f32x4* gu, const f32x4* gi)
{
    constexpr int C = G / 4, NG = (32 * C) / 64;
#pragma unroll
    for (int it = 0; it < NG; ++it) {
        const int e = it * 64 + lane;
        *reinterpret_cast<f32x4*>(s_gp + (e / C) * (G + 4) + 4 * (e % C)) = gu[it] * gi[it];
    }
}

}  // namespace
