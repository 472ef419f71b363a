// K-b: the fused scoring tower for gfx950 (MI355X), fp32 MFMA.
//
// Replaces, per (user, item) pair, the reference's chain
//   4x Embedding -> mul -> cat -> [Linear, ReLU] x3 -> cat -> Linear        src/helpers/ncf_model.py:205-249
// and the Hybrid head  mul -> cat -> [Linear, ReLU] x2 -> cat -> Linear     src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:167-184
//
// Math.  Layer 1 acts on a concatenation, so it splits exactly:
//   W1 [eu | ei] + b1 = (W1[:, :dm] eu + b1) + W1[:, dm:] ei = P[u] + Q[i]
// P and Q are per-user / per-item 128-vectors (computed by the GEMM kernel in gemm.hip); per pair
// layer 1 is then add + ReLU.  Layers 2 (128->64) and 3 (64->32) run on v_mfma_f32_32x32x2_f32 with
// the PAIR on the MFMA column (lane & 31) and the FEATURE on the accumulator rows, so the 32x32
// accumulator of layer 2 is, register for register, the B operand of layer 3: nothing moves
// between layers, no LDS round trip, no HBM traffic.  Per 32 pairs a wave issues 128 + 32 MFMAs.
//
// MFMA operand maps (guide section 3):  A: lane l holds A[i = l & 31][kslot = l >> 5]
//                                       B: lane l holds B[kslot = l >> 5][j = l & 31]
//                                       D: lane l reg r holds D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31]
// The k index carried by (step, kslot) is free as long as A and B agree; we pick it so that each
// lane's operands for 4 consecutive steps are 16 contiguous bytes (one ds_read_b128 / one float4):
//   layer 2, quad q (0..15), step jj (0..3), half h:   k = 8 q + 4 h + jj
//   layer 3, quad q' (0..7), step jj, half h:           k = 32 (q' >> 2) + 8 (q' & 3) + 4 h + jj
//                                                      (= the feature held by acc[q' >> 2][4 (q' & 3) + jj])
#include "tower_shared.h"
#include "topk_shared.h"

namespace {

// ---------------------------------------------------------------------------------------------
// weight image packing
// ---------------------------------------------------------------------------------------------
struct PackArgs {
    const float* W2;   // [64,128]
    const float* b2;   // [64]
    const float* W3;   // [32,64] or null
    const float* b3;   // [32] or null
    const float* wp;   // [G + mlp_out]
    int G;
    int has_l3;
};

__global__ void pack_image_kernel(PackArgs a, float* __restrict__ img)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= IMG_SIZE) return;
    float v = 0.f;
    if (idx < IMG_W3) {  // W2 image [t][q][lane][jj]
        const int e = idx - IMG_W2;
        const int jj = e & 3, lane = (e >> 2) & 63, q = (e >> 8) & 15, t = e >> 12;
        v = a.W2[(32 * t + (lane & 31)) * H1 + 8 * q + 4 * (lane >> 5) + jj];
    } else if (idx < IMG_B2) {  // W3 image [q'][lane][jj]
        if (a.has_l3) {
            const int e = idx - IMG_W3;
            const int jj = e & 3, lane = (e >> 2) & 63, q = e >> 8;
            v = a.W3[(lane & 31) * H2 + 32 * (q >> 2) + 8 * (q & 3) + 4 * (lane >> 5) + jj];
        }
    } else if (idx < IMG_B3) {  // b2 in accumulator order [t][rq][h][jj]
        const int e = idx - IMG_B2;
        const int jj = e & 3, h = (e >> 2) & 1, rq = (e >> 3) & 3, t = e >> 5;
        v = a.b2[32 * t + jj + 8 * rq + 4 * h];
    } else if (idx < IMG_WPM) {  // b3 in accumulator order [rq][h][jj]
        if (a.has_l3) {
            const int e = idx - IMG_B3;
            const int jj = e & 3, h = (e >> 2) & 1, rq = e >> 3;
            v = a.b3[jj + 8 * rq + 4 * h];
        }
    } else if (idx < IMG_WPG) {  // predict weights of the MLP half, accumulator order
        const int e = idx - IMG_WPM;
        const int jj = e & 3, h = (e >> 2) & 1;
        if (a.has_l3) {
            if (e < 32) v = a.wp[a.G + jj + 8 * (e >> 3) + 4 * h];
        } else {
            const int rq = (e >> 3) & 3, t = e >> 5;
            v = a.wp[a.G + 32 * t + jj + 8 * rq + 4 * h];
        }
    } else {  // predict weights of the GMF half
        const int e = idx - IMG_WPG;
        if (e < a.G) v = a.wp[e];
    }
    img[idx] = v;
}

// ---------------------------------------------------------------------------------------------
// the per-group MFMA body, shared by the tile and the pair kernel
// ---------------------------------------------------------------------------------------------
// Layers 2 (+3) and the MLP half of the predict layer for the 32 pairs of this wave.
// x1.raw(q) issues the LDS read of this lane's 4 layer-1 inputs [8q + 4h .. 8q + 4h + 3] for its pair and
// x1.finish(q, raw) turns them into activations relu(P + Q) (identity when the staged tile already holds them).
template <bool L3, typename X1>
__device__ __forceinline__ float tower_mlp_part(const float* s_img, int lane, X1 x1)
{
    const int h = lane >> 5;
    // A wave inside the MFMA body outranks its SIMD's other waves while they stage tiles, claim work or keep lists: same-box A/B
    // (RECO_HIP_LIB variants) cfg2 tile kernel -0.4 %, cfg5 -0.7 %, 11.5 M-pair list -1.6 %, Hybrid head tile kernel -0.9 %
    // (at three waves per SIMD; with two it lost 0.4 %).
    __builtin_amdgcn_s_setprio(1);
    f32x16 acc0 = bias16(s_img + lds_img<L3>(IMG_B2), h);
    f32x16 acc1 = bias16(s_img + lds_img<L3>(IMG_B2) + 32, h);
    const f32x4* w2 = reinterpret_cast<const f32x4*>(s_img + IMG_W2) + lane;
    // software-pipelined by one quad: the LDS reads (and the add+ReLU) of quad q+1 are issued ahead of
    // the 8 MFMAs of quad q, so no MFMA waits on an LDS round trip
    auto r0 = x1.raw(0);
    f32x4 x = x1.finish(0, r0);
    f32x4 a0 = w2[0];
    f32x4 a1 = w2[16 * 64];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        auto xraw = r0;                         // (type only: overwritten below whenever it is used)
        f32x4 a0n = a0, a1n = a1;
        if (q + 1 < 16) {                       // issue quad q+1's three ds_read_b128 first ...
            x1.pre(q + 1);                      // (pair kernel: the second K half is staged before quad 8 is read)
            xraw = x1.raw(q + 1);
#ifndef RECO_ABL_NO_W2_READS
            a0n = w2[(q + 1) * 64];
            a1n = w2[(16 + q + 1) * 64];
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, x.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, x.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, x.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, x.y, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 xn = x;
#ifdef RECO_ABL_NO_X1_VALU
        if (q + 1 < 16) xn = x1.passthrough(xraw);
#else
        if (q + 1 < 16) xn = x1.finish(q + 1, xraw);   // ... their data is back 4 MFMAs (256 cycles) later
#endif
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, x.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, x.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, x.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, x.w, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        x = xn; a0 = a0n; a1 = a1n;
    }
    // independent work of the caller (the GMF half of the predict layer) fills the wait for layer 2's last accumulators.
    // The two halves of the predict layer are summed separately and added at the end, in every kernel that uses this body: the
    // tile kernels (GMF half here) and the pair-list kernel (GMF half after the call) then agree to the bit.
    const float pre = x1.between();
    float part = 0.f;
#ifdef RECO_ABL_NO_L3
    return acc0[0] + acc1[0];
#endif
    if constexpr (L3) {
        f32x16 acc3 = bias16(s_img + lds_img<L3>(IMG_B3), h);
        const f32x4* w3 = reinterpret_cast<const f32x4*>(s_img + IMG_W3) + lane;
        f32x4 a = w3[0];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f32x4 an = a;
            if (q + 1 < 8) an = w3[(q + 1) * 64];   // one quad ahead, as in layer 2
            __builtin_amdgcn_sched_barrier(0);
            const int r = 4 * (q & 3);
            if (q < 4) {
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, relu1(acc0[r + 0]), acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, relu1(acc0[r + 1]), acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, relu1(acc0[r + 2]), acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, relu1(acc0[r + 3]), acc3, 0, 0, 0);
            } else {
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, relu1(acc1[r + 0]), acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, relu1(acc1[r + 1]), acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, relu1(acc1[r + 2]), acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, relu1(acc1[r + 3]), acc3, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            a = an;
        }
        __builtin_amdgcn_s_setprio(0);
        part = dot16_relu(acc3, s_img + lds_img<L3>(IMG_WPM), h, part);
    } else {
        __builtin_amdgcn_s_setprio(0);
        part = dot16_relu(acc0, s_img + lds_img<L3>(IMG_WPM), h, part);
        part = dot16_relu(acc1, s_img + lds_img<L3>(IMG_WPM) + 32, h, part);
    }
    return pre + part;
}

// THREADS x TUBX: 768 threads x 24 users per tile (3 waves per SIMD at <= 168 VGPRs) for every shape.  The Hybrid head's 64-wide
// GMF item tile fits beside 24 P rows because the LDS copy of the weight image leaves the (absent) third layer out
// (lds_img<L3>, reco_common.h): 512 x 16 -> 768 x 24 took its tile kernel from 2.905 to 2.80 ms.
template <int G, bool L3, int THREADS, int TUBX>
__global__ __launch_bounds__(THREADS, THREADS / 256) void tower_tile_kernel(TileArgs a)
{
    constexpr int TILE_THREADS = THREADS;      // shadow the 512-thread defaults of tower_shared.h
    constexpr int TUB = TUBX;
    constexpr int GS = G + 4;
    constexpr int BUF = TI * QS + (G > 0 ? TI * GS : 0);      // floats per item-side buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_img = smem;
    float* s_buf = smem + LDS_IMG_SIZE<L3>;
    // P rows of the tile's users: staged in LDS with the item side and read per quad by a broadcast ds_read_b128 (in 64 VGPRs
    // they cost a wave per SIMD and a global reload at every user switch)
    constexpr int NP = (TUB * (H1 / 4) + TILE_THREADS - 1) / TILE_THREADS;   // float4 of P per thread (2; 1 at 768 x 24)
    float* s_pbuf = s_buf + 2 * BUF;                            // [2][TUB][H1]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const int64_t tiles_n = (a.N + TI - 1) / TI;
    const int64_t tiles = tiles_n * ((a.U + TUB - 1) / TUB);

    load_image_scaled<L3, TILE_THREADS>(s_img, a.img, tid, layer1_scale(a.l1));

    f32x4 pst[NP];
    auto p_stage_load = [&](int64_t t) {
        const int64_t ub = (t / tiles_n) * TUB;
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const int i = tid + s * TILE_THREADS;
            const int64_t u = ub + (i >> 5);
            pst[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < TUB * (H1 / 4) && u < a.U) pst[s] = reinterpret_cast<const f32x4*>(a.P + u * H1)[i & 31];
        }
    };
    auto p_stage_store = [&](int buf) {
#pragma unroll
        for (int s = 0; s < NP; ++s)
            if (tid + s * TILE_THREADS < TUB * (H1 / 4))
                reinterpret_cast<f32x4*>(s_pbuf + buf * TUB * H1)[tid + s * TILE_THREADS] = pst[s];
    };

#ifdef RECO_DIAG_CLOCK
    const unsigned long long diag_c0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    __shared__ long long s_next[2];
    TileStage<G, THREADS> st;
    int64_t tile = blockIdx.x;
    tile_queue_fetch(a, tid, tile + gridDim.x, &s_next[0]);
    if (tile < tiles) {
        tile_stage_load<G, THREADS>(a, (tile % tiles_n) * TI, tile / tiles_n == 0, tid, st);
        tile_stage_store<G, THREADS>(s_buf, s_buf + TI * QS, tid, st);
        p_stage_load(tile);
        p_stage_store(0);
    }
    __syncthreads();
    int64_t next = s_next[0];

    for (int it = 0; tile < tiles; ++it) {
        const float* s_q = s_buf + (it & 1) * BUF;
        const float* s_g = s_q + TI * QS;
        const int64_t n0 = (tile % tiles_n) * TI;
        const int64_t u0 = (tile / tiles_n) * TUB;
        const int g_n = a.N - n0 > 32 ? TI / 32 : 1;           // the ragged last tile: no MFMAs for a column group past the catalog
        // (handing those half-cost tiles out LAST for all user blocks, so that the launch ends on finer work units, was measured
        // same-box on cfg2 -- 252 half tiles behind 13,860 whole ones: 3.152 -> 3.168 ms, four alternating runs; rejected)
        tile_queue_fetch(a, tid, next + gridDim.x, &s_next[(it + 1) & 1]);     // the tile after next
        if (next < tiles) { tile_stage_load<G, THREADS>(a, (next % tiles_n) * TI, next / tiles_n == 0, tid, st); p_stage_load(next); }
        const float* s_p = s_pbuf + (it & 1) * TUB * H1;

        for (int uu = wave; uu < TUB; uu += TILE_THREADS / 64) {
            const int64_t u = u0 + uu;
            if (u >= a.U) break;
            f32x4 gu[G > 0 ? G / 8 : 1];
            if constexpr (G > 0) {
                const int64_t uid = clamp_id(a.users ? a.users[u] : a.user_base + u, a.n_users,
                                             (lane == 0 && n0 == 0) ? a.oob : nullptr);
                const f32x4* gr = reinterpret_cast<const f32x4*>(a.EUg + uid * G + h * (G / 2));
                const float* wpg = s_img + lds_img<L3>(IMG_WPG) + h * (G / 2);
#pragma unroll
                for (int c = 0; c < G / 8; ++c)   // fold the predict weights into the user row once per user
                    gu[c] = gr[c] * *reinterpret_cast<const f32x4*>(wpg + 4 * c);
            }
#pragma unroll 1
            for (int g = 0; g < g_n; ++g) {
                // keep the (loop-invariant) weight-image reads inside the loop: hoisted, they would pin ~200 VGPRs
                asm volatile("" ::: "memory");
                const float* qrow = s_q + (g * 32 + j) * QS + 4 * h;
                struct PQ { f32x4 q, p; };
                struct X1TileLds {
                    const float* qrow;
                    const float* prow;     // this user's P row in LDS (+ 4 h)
                    __device__ __forceinline__ PQ raw(int q) const
                    {
                        return PQ{*reinterpret_cast<const f32x4*>(qrow + 8 * q), *reinterpret_cast<const f32x4*>(prow + 8 * q)};
                    }
                    __device__ __forceinline__ void pre(int) const {}
                    __device__ __forceinline__ f32x4 finish(int, const PQ& v) const { return add_clamp4(v.p, v.q); }   // = s * relu(p + q), P / Q pre-scaled
                    __device__ __forceinline__ f32x4 passthrough(const PQ& v) const { return v.q; }
                    const float* grow;     // this pair's GMF item row half in LDS
                    const f32x4* gu;       // the user's GMF row half, predict weights folded in
                    __device__ __forceinline__ float between() const { return gmf_half_tile<G>(grow, gu); }
                };
                const float part = tower_mlp_part<L3>(s_img, lane, X1TileLds{qrow, s_p + uu * H1 + 4 * h, s_g + (g * 32 + j) * GS + h * (G / 2), gu});
                const float total = part + __shfl_xor(part, 32) + a.bp;
                const int64_t n = n0 + g * 32 + j;
                if (h == 0 && n < a.N) a.out[u * a.ld_out + n] = epilogue(total, a.act, a.lo, a.hi);
            }
        }
        if (next < tiles) {
            float* d_q = s_buf + ((it + 1) & 1) * BUF;
            tile_stage_store<G, THREADS>(d_q, d_q + TI * QS, tid, st);
            p_stage_store((it + 1) & 1);
        }
        __syncthreads();   // tile i fully consumed, tile i + 1 fully staged
        tile = next;
        next = s_next[(it + 1) & 1];
    }
#ifdef RECO_DIAG_CLOCK
    if (tid == 0 && (blockIdx.x & 63) == 0) {
        const unsigned long long c = __builtin_amdgcn_s_memtime() - diag_c0, r = __builtin_amdgcn_s_memrealtime() - diag_r0;
        printf("block %u: %llu cycles, %llu x10ns -> %.0f MHz\n", blockIdx.x, c, r, (double)c / (double)r * 100.0);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Fused score + per-user top-k: the score rows never reach HBM (north_star (d)).
// ---------------------------------------------------------------------------------------------
// Same tile body as tower_tile_kernel, different hand-out and a different sink.  A wave owns users w, w + WAVES of a user block and
// keeps, per user, the best k scores so far SORTED IN REGISTERS (topk_shared.h: 64-bit keys, entrants batched 64 at a time); the
// 64 scores of a tile row (two MFMA column groups) are offered to that list as they leave the accumulators.
// Hand-out: every USER BLOCK has its own counter of item tiles.  A workgroup stays on its block -- its lists stay live -- and
// pulls that block's item tiles one at a time; when the block runs dry it writes its lists out as one of the block's V partial
// lists ("visitor slots", claimed by an atomic) and moves on: to a block nobody has started (a global counter), else to a block
// in progress that still has tiles and a free slot.  Balance is at
// single-tile granularity (several workgroups can drain one block: cfg5 has 43 user blocks for 256 CUs), while a list is flushed
// only when a workgroup leaves a block -- once or twice per launch.  A merge launch (topk_merge_kernel) folds the <= V sorted
// partial lists of every user; slots nobody wrote are zero = empty (the buffer is cleared per launch).
// Results are bit-identical to score_matrix + reco_topk_rows: the scores are the same and the keys carry the position.
struct TopkFuse {
    int k;
    int v_max;                  // visitor slots (partial lists) per user block, <= 1024
    unsigned int* fresh;        // [1] user blocks handed out beyond the first gridDim.x, zeroed per launch
    unsigned int* next_tile;    // [user blocks] item-tile counter, zeroed per launch
    unsigned int* visits;       // [user blocks] visitor counter, zeroed per launch
    u64* partial;               // [U][v_max][k], zeroed per launch
};

template <int G, bool L3, int THREADS, int TUBX, int SLOTS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void tower_tile_topk_kernel(TileArgs a, TopkFuse f)
{
    constexpr int TILE_THREADS = THREADS;
    constexpr int TUB = TUBX;
    constexpr int WAVES = THREADS / 64;
    constexpr int UPW = TUB / WAVES;                            // users per wave (2)
    static_assert(UPW == 2 && TI == 64, "the fused sink is written for two users per wave and two column groups per tile");
    constexpr int GS = G + 4;
    constexpr int BUF = TI * QS + (G > 0 ? TI * GS : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_img = smem;
    float* s_buf = smem + LDS_IMG_SIZE<L3>;
    constexpr int NP = (TUB * (H1 / 4) + TILE_THREADS - 1) / TILE_THREADS;
    // The workgroup stays with one user block for many tiles, so the block's P rows are staged ONCE per block (single buffer,
    // an extra barrier at the switch) and the space of the second buffer holds the pending-candidate rings of the block's users.
    float* s_pbuf = s_buf + 2 * BUF;                            // [TUB][H1]
    u64* s_ring = reinterpret_cast<u64*>(s_pbuf + TUB * H1);     // [TUB][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const unsigned tiles_n = (unsigned)((a.N + TI - 1) / TI);
    const unsigned blocks_u = (unsigned)((a.U + TUB - 1) / TUB);

    load_image_scaled<L3, TILE_THREADS>(s_img, a.img, tid, layer1_scale(a.l1));

    // ---- hand-out (wave 0; every value below is wave-uniform): the tile fetched last belongs to (q_blk, q_v) ----
    constexpr int LIST_SHIFT = 10;                             // published as block << 10 | visitor slot
    __shared__ unsigned s_blkv[2], s_t[2];                     // published tile: block / slot, item tile (~0u: none)
    const unsigned v_max = (unsigned)f.v_max;
    unsigned q_blk = 0, q_v = 0;
    bool q_have = false;                                       // the workgroup holds a visitor slot in q_blk
    bool q_started = false, q_done = false;
    auto one = [&](unsigned* p) -> unsigned {                  // one atomic increment per wave, result in a scalar
        unsigned r = 0;
        if (lane == 0) r = atomicAdd(p, 1u);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)r);
    };
    auto try_block = [&](unsigned b, unsigned& t) {            // join block b if it has a free partial list and a tile left
        const unsigned v = one(&f.visits[b]);
        if (v >= v_max) return false;                          // every partial list of that block is taken
        const unsigned tt = one(&f.next_tile[b]);
        if (tt >= tiles_n) return false;                       // drained meanwhile: the claimed list stays empty (zero keys)
        q_blk = b; q_v = v; q_have = true; t = tt;
        return true;
    };
    // Help with a block in progress.  Optional -- whoever holds a block drains it -- so the search is bounded: the blocks handed
    // out last are the ones still running at the end.  64 blocks are probed per round trip, one per lane (probed one after the
    // other by a single thread, 252 drained blocks cost every workgroup ~0.25 ms at the end of a 3.3 ms launch).
    auto help = [&](unsigned& t) {
        const unsigned window = blocks_u < 256u ? blocks_u : 256u;
        for (unsigned base = 0; base < window && !q_have; base += 64) {
            const unsigned i = base + lane;
            const unsigned b = blocks_u - 1u - (blockIdx.x + i) % window;
            bool ok = false;
            if (i < window)
                ok = __hip_atomic_load(&f.next_tile[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tiles_n &&
                     __hip_atomic_load(&f.visits[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v_max;
            u64 m = __ballot(ok);
            while (m != 0 && !q_have) {
                const int l = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                try_block((unsigned)__builtin_amdgcn_readlane((int)b, l), t);
            }
        }
    };
    // Claiming a tile is split in two so that the counter's round trip is never waited for: the atomic is issued at the top of a
    // tile (fetch_issue) and its result looked at after the tile's MFMAs (fetch).
    unsigned t_raw = 0;                                        // lane 0
    auto fetch_issue = [&]() { if (q_have && lane == 0) t_raw = atomicAdd(&f.next_tile[q_blk], 1u); };
    auto fetch = [&](int slot) {                               // wave 0: publish the next tile in LDS slot `slot`
        unsigned t = ~0u;
        if (q_have) {                                          // 1. stay on the block: its lists are live in this workgroup
            t = (unsigned)__builtin_amdgcn_readfirstlane((int)t_raw);
            if (t >= tiles_n) { t = ~0u; q_have = false; }
        }
        if (!q_have && !q_done) {
            if (!q_started) {                                  // 2a. the workgroup's own first block
                q_started = true;
                if (blockIdx.x < blocks_u) try_block(blockIdx.x, t);
            }
            while (!q_have) {                                  // 2b. a block nobody has been handed yet
                const unsigned b = gridDim.x + one(f.fresh);
                if (b >= blocks_u) break;
                try_block(b, t);
            }
            if (!q_have) help(t);                              // 3. a block in progress
            if (!q_have) q_done = true;                        // nothing left that this workgroup could join: stop looking
        }
        if (lane == 0) {
            s_blkv[slot] = (q_blk << LIST_SHIFT) | q_v;
            s_t[slot] = q_have ? t : ~0u;
        }
    };

    f32x4 pst[NP];
    auto p_stage_load = [&](unsigned blk) {
        const int64_t ub = (int64_t)blk * TUB;
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const int i = tid + s * TILE_THREADS;
            const int64_t u = ub + (i >> 5);
            pst[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < TUB * (H1 / 4) && u < a.U) pst[s] = reinterpret_cast<const f32x4*>(a.P + u * H1)[i & 31];
        }
    };
    auto p_stage_store = [&]() {
#pragma unroll
        for (int s = 0; s < NP; ++s)
            if (tid + s * TILE_THREADS < TUB * (H1 / 4))
                reinterpret_cast<f32x4*>(s_pbuf)[tid + s * TILE_THREADS] = pst[s];
    };

    // ---- per-wave top-k state: one sorted list per owned user ----
    WaveList<SLOTS> L0, L1;
    RingPending P0, P1;
    u64* const ring0 = s_ring + wave * 64;                      // rings of this wave's two users
    u64* const ring1 = s_ring + (wave + WAVES) * 64;
    u64 thr0 = 0, thr1 = 0;
    L0.clear(); L1.clear(); P0.clear(); P1.clear();
    unsigned list_blkv = ~0u;                                  // block / slot the lists belong to (~0u: nothing collected yet)
    auto write_lists = [&]() {                                 // this wave's two users of block list_blkv -> their partial lists
        const unsigned blk = list_blkv >> LIST_SHIFT, v = list_blkv & ((1u << LIST_SHIFT) - 1u);
        flush_ring<SLOTS, 64>(L0, P0, ring0, thr0, f.k, lane);
        flush_ring<SLOTS, 64>(L1, P1, ring1, thr1, f.k, lane);
#pragma unroll
        for (int ui = 0; ui < UPW; ++ui) {
            const int64_t u = (int64_t)blk * TUB + wave + ui * WAVES;
            if (u >= a.U) continue;
            u64* dst = f.partial + (u * f.v_max + v) * f.k;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int r = s * 64 + lane;
                if (r < f.k) dst[r] = ui == 0 ? L0.e[s] : L1.e[s];
            }
        }
        L0.clear(); L1.clear(); P0.clear(); P1.clear();
        thr0 = 0; thr1 = 0;
    };

    TileStage<G, THREADS> st;
    if (wave == 0) { fetch(0); fetch_issue(); fetch(1); }
    __syncthreads();
    unsigned cur_blkv = s_blkv[0], cur_t = s_t[0];
    if (cur_t != ~0u) {
        tile_stage_load<G, THREADS>(a, (int64_t)cur_t * TI, (cur_blkv >> LIST_SHIFT) == 0, tid, st);
        tile_stage_store<G, THREADS>(s_buf, s_buf + TI * QS, tid, st);
        p_stage_load(cur_blkv >> LIST_SHIFT);
        p_stage_store();
    }
    __syncthreads();
    unsigned nxt_blkv = s_blkv[1], nxt_t = s_t[1];

    for (int it = 0; cur_t != ~0u; ++it) {
        const float* s_q = s_buf + (it & 1) * BUF;
        const float* s_g = s_q + TI * QS;
        const int64_t n0 = (int64_t)cur_t * TI;
        const int64_t u0 = (int64_t)(cur_blkv >> LIST_SHIFT) * TUB;
        if (wave == 0) fetch_issue();                           // the tile after next: claimed now, published below
        if (nxt_t != ~0u) { tile_stage_load<G, THREADS>(a, (int64_t)nxt_t * TI, (nxt_blkv >> LIST_SHIFT) == 0, tid, st); }
        const float* s_p = s_pbuf;
        if (cur_blkv != list_blkv) {                           // the workgroup moved to another block (or holds its first tile)
            if (list_blkv != ~0u) write_lists();
            list_blkv = cur_blkv;
        }

        // ONE copy of the tile body for both users of the wave (the loop is not unrolled: two copies of 160 MFMAs plus the list
        // code were ~56 KB of instructions): the body always works on L0 / P0 / thr0 and the two users' states trade places
        // after every pass, so after the two passes each is back where it started
#pragma unroll 1
        for (int ui = 0; ui < UPW; ++ui) {
            const int uu = wave + ui * WAVES;
            const int64_t u = u0 + uu;
            if (u < a.U) {
            f32x4 gu[G > 0 ? G / 8 : 1];
            if constexpr (G > 0) {
                const int64_t uid = clamp_id(a.users ? a.users[u] : a.user_base + u, a.n_users, (lane == 0 && n0 == 0) ? a.oob : nullptr);
                const f32x4* gr = reinterpret_cast<const f32x4*>(a.EUg + uid * G + h * (G / 2));
                const float* wpg = s_img + lds_img<L3>(IMG_WPG) + h * (G / 2);
#pragma unroll
                for (int c = 0; c < G / 8; ++c) gu[c] = gr[c] * *reinterpret_cast<const f32x4*>(wpg + 4 * c);
            }
            float held[2] = {0.f, 0.f};                        // the two column groups' scores of pair j
#pragma unroll 1
            for (int g = 0; g < TI / 32; ++g) {
                if (n0 + g * 32 >= a.N) break;                   // the ragged last tile: no MFMAs for a group past the catalog
                asm volatile("" ::: "memory");
                const float* qrow = s_q + (g * 32 + j) * QS + 4 * h;
                struct PQ { f32x4 q, p; };
                struct X1TileLds {
                    const float* qrow;
                    const float* prow;
                    __device__ __forceinline__ PQ raw(int q) const
                    {
                        return PQ{*reinterpret_cast<const f32x4*>(qrow + 8 * q), *reinterpret_cast<const f32x4*>(prow + 8 * q)};
                    }
                    __device__ __forceinline__ void pre(int) const {}
                    __device__ __forceinline__ f32x4 finish(int, const PQ& v) const { return add_clamp4(v.p, v.q); }   // = s * relu(p + q), P / Q pre-scaled
                    __device__ __forceinline__ f32x4 passthrough(const PQ& v) const { return v.q; }
                    const float* grow;     // this pair's GMF item row half in LDS
                    const f32x4* gu;       // the user's GMF row half, predict weights folded in
                    __device__ __forceinline__ float between() const { return gmf_half_tile<G>(grow, gu); }
                };
                const float part = tower_mlp_part<L3>(s_img, lane, X1TileLds{qrow, s_p + uu * H1 + 4 * h, s_g + (g * 32 + j) * GS + h * (G / 2), gu});
                const float sc = epilogue(part + __shfl_xor(part, 32) + a.bp, a.act, a.lo, a.hi);
                if (g == 0) held[0] = sc;
                else held[1] = sc;
            }
            {
                // lanes 0..31 offer group 0's scores (items n0 + j), lanes 32..63 group 1's (items n0 + 32 + j): item n0 + lane
                const float x = h == 0 ? held[0] : held[1];
                // one float compare decides the common case (no score of the tile row reaches the k-th key's score; NaN
                // on either side passes and is settled by the exact keys)
                const float tf = key_float((uint32_t)(thr0 >> 32));
                const int64_t n = n0 + lane;
                if (__ballot(n < a.N && !(x < tf)) != 0) {
                    const u64 key = ((u64)float_key(x) << 32) | (uint32_t)~(uint32_t)n;
                    offer_ring<SLOTS, 64>(L0, P0, ui == 0 ? ring0 : ring1, key, n < a.N, thr0, f.k, lane);
                }
            }
            }
            { const WaveList<SLOTS> tl = L0; L0 = L1; L1 = tl; }
            { const RingPending tp = P0; P0 = P1; P1 = tp; }
            { const u64 tt = thr0; thr0 = thr1; thr1 = tt; }
        }
        if (nxt_t != ~0u) {
            float* d_q = s_buf + ((it + 1) & 1) * BUF;
            tile_stage_store<G, THREADS>(d_q, d_q + TI * QS, tid, st);
        }
        if (wave == 0) fetch(it & 1);                          // slot it & 1 was read as `nxt` two barriers ago
        __syncthreads();   // tile fully consumed, next tile fully staged, the tile after next published
        if (nxt_t != ~0u && (nxt_blkv >> LIST_SHIFT) != (cur_blkv >> LIST_SHIFT)) {     // the workgroup moves to another user block:
            p_stage_load(nxt_blkv >> LIST_SHIFT);                                        // its P rows replace the old block's
            p_stage_store();
            __syncthreads();
        }
        cur_blkv = nxt_blkv; cur_t = nxt_t;
        nxt_blkv = s_blkv[it & 1]; nxt_t = s_t[it & 1];
    }
    if (list_blkv != ~0u) write_lists();
}

// Pair-list kernel.  WAVES waves share the weight image; every wave is private otherwise (no workgroup barrier in the loop):
// 32 pairs at a time, the rows of the NEXT group in flight in registers while the MFMAs of the current one run out of the wave's
// LDS staging area.  The layer-1 activations are staged one K half at a time (tower_shared.h): the first half before quad 0,
// the second -- into the same 8.7 KB -- right before quad 8 is read (LDS operations of one wave execute in issue order, so the
// overwrite needs no barrier).  With half rows 8 waves fit next to the image (148 KB): 2 waves per SIMD.
// CLAMP: P and Q were written pre-scaled and layer 1 is the clamped add (tower_shared.h); without it (a.l1 null: row blocks
// handed in per call, where a bound over the rows would cost a pass of its own) layer 1 is the plain add + max.
template <int G, bool L3, int WAVES, bool BIG, bool CLAMP>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void tower_pairs_kernel(PairArgs a)
{
    constexpr int GC = G / 4;                                    // GMF column sums per pair (tower_shared.h)
    constexpr int WAVE_FLOATS = 32 * HS + 32 * GC;
    constexpr int NG = G > 0 ? (32 * (G / 4)) / 64 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_img = smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    float* s_x = smem + LDS_IMG_SIZE<L3> + wave * WAVE_FLOATS;   // [32][HS]  relu(P+Q), one K half of this wave's pairs
    float* s_gp = s_x + 32 * HS;                          // [32][GC]  gmf column sums

    load_image_scaled<L3, 64 * WAVES>(s_img, a.img, tid, CLAMP ? layer1_scale(a.l1) : Layer1Scale{1.f, 1.f});
    __syncthreads();

    const int64_t ngroups = (a.B + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    int64_t grp = (int64_t)blockIdx.x * WAVES + wave;
    f32x4 pf[8], qf[8], ps[8], qs[8];                     // first / second K half of the next group's P and Q rows
    f32x4 gu[NG], gi[NG];
    int nu = 0, ni = 0;                                   // the next group's ids (lanes 0..31)
    // The ids run TWO groups ahead of the MFMAs: a group's row loads are issued from ids that are already in registers, and
    // the id loads of the group after it fly under a whole group of MFMAs.  (Loaded where they are needed, each id load was an
    // exposed global round trip at the top of every group: the row addresses depend on it.)
    int64_t ru = 0, ri = 0;                               // raw ids of the group after next (lanes 0..31)
    if (grp < ngroups) {
        pair_ids(a, grp * 32, lane, nu, ni);
        pair_load_half<BIG>(a, nu, ni, lane, 0, pf, qf);
        pair_load_half<BIG>(a, nu, ni, lane, 1, ps, qs);
        if constexpr (G > 0) pair_load_gmf<G, BIG>(a, nu, ni, lane, gu, gi);
        if (grp + stride < ngroups) pair_ids_raw(a, (grp + stride) * 32, lane, ru, ri);
    }
    for (; grp < ngroups; grp += stride) {
        const int64_t base = grp * 32;
        const bool more = grp + stride < ngroups;
        pair_store_half<CLAMP>(s_x, lane, pf, qf);
        if constexpr (G > 0) pair_store_gmf_cols<G>(s_gp, lane, gu, gi, s_img + lds_img<L3>(IMG_WPG));
        __builtin_amdgcn_wave_barrier();
        // the second half (ps / qs) still belongs to THIS group: the next group's ids wait in other registers until it is stored
        int nu2 = 0, ni2 = 0;
        if (more) {
            pair_ids_clamp(a, (grp + stride) * 32, lane, ru, ri, nu2, ni2);     // loaded one group ago
            pair_load_half<BIG>(a, nu2, ni2, lane, 0, pf, qf);             // in flight under the MFMAs
            if constexpr (G > 0) pair_load_gmf<G, BIG>(a, nu2, ni2, lane, gu, gi);
            if (grp + 2 * stride < ngroups) pair_ids_raw(a, (grp + 2 * stride) * 32, lane, ru, ri);
        }
        const float* xrow = s_x + j * HS + 4 * h;
        struct X1Pair {
            const PairArgs& a;
            const float* xrow;
            float* s_x;
            f32x4* ps;
            f32x4* qs;
            int nu, ni;
            int lane;
            bool more;
            __device__ __forceinline__ void pre(int q) const
            {
                if (q == 8) {                 // quads 0..7 have all been read: stage the second half, refill its registers
                    pair_store_half<CLAMP>(s_x, lane, ps, qs);
                    __builtin_amdgcn_wave_barrier();
                    if (more) pair_load_half<BIG>(a, nu, ni, lane, 1, ps, qs);
                }
            }
            __device__ __forceinline__ f32x4 raw(int q) const { return *reinterpret_cast<const f32x4*>(xrow + 8 * (q & 7)); }
            __device__ __forceinline__ f32x4 finish(int, f32x4 v) const { return v; }
            __device__ __forceinline__ f32x4 passthrough(f32x4 v) const { return v; }
            __device__ __forceinline__ float between() const { return 0.f; }
        };
        float part = tower_mlp_part<L3>(s_img, lane, X1Pair{a, xrow, s_x, ps, qs, nu2, ni2, lane, more});
        if constexpr (G > 0) part = gmf_half_pair<G>(s_gp, j, h) + part;    // summed on its own, as in the tile kernels
        const float total = part + __shfl_xor(part, 32) + a.bp;
        const int64_t b = base + j;
        if (h == 0 && b < a.B) a.out[b] = epilogue(total, a.act, a.lo, a.hi);
        __builtin_amdgcn_wave_barrier();
    }
}

// Quarter-staged form of the pair-list kernel (GMF width <= 32).  The layer-1 activations are staged one K QUARTER at a time.
// LOOK = 2: the rows travel in two register sets of one quarter each (64 VGPRs in flight instead of 128): quarter c is stored
// from set c & 1 right before quad 4 c is read, and the set is refilled with quarter c + 2 (of this group or the next).
// LOOK = 1: one register set (32 VGPRs in flight), refilled with quarter c + 1 -- few enough registers for three waves per SIMD.
template <int G, bool L3, int WAVES, bool BIG, bool CLAMP, int LOOK>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void tower_pairs_q_kernel(PairArgs a)
{
    constexpr int GC = G / 4;                                    // GMF column sums per pair (tower_shared.h)
    constexpr int WAVE_FLOATS = 32 * QSP + 32 * GC;
    constexpr int NG = G > 0 ? (32 * (G / 4)) / 64 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_img = smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    float* s_x = smem + LDS_IMG_SIZE<L3> + wave * WAVE_FLOATS;   // [32][QSP]  one K quarter of this wave's layer-1 activations
    float* s_gp = s_x + 32 * QSP;                                // [32][GC]   gmf column sums

    load_image_scaled<L3, 64 * WAVES>(s_img, a.img, tid, CLAMP ? layer1_scale(a.l1) : Layer1Scale{1.f, 1.f});
    __syncthreads();

    const int64_t ngroups = (a.B + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    int64_t grp = (int64_t)blockIdx.x * WAVES + wave;
    f32x4 p0[4], q0[4], p1[LOOK == 2 ? 4 : 1], q1[LOOK == 2 ? 4 : 1];   // register sets 0 / 1
    f32x4 gu[NG], gi[NG];
    int nu = 0, ni = 0;
    int64_t ru = 0, ri = 0;                               // raw ids of the group after next (lanes 0..31)
    if (grp < ngroups) {
        pair_ids(a, grp * 32, lane, nu, ni);
        pair_load_quarter<BIG>(a, nu, ni, lane, 0, p0, q0);
        if constexpr (LOOK == 2) pair_load_quarter<BIG>(a, nu, ni, lane, 1, p1, q1);
        if constexpr (G > 0) pair_load_gmf<G, BIG>(a, nu, ni, lane, gu, gi);
        if (grp + stride < ngroups) pair_ids_raw(a, (grp + stride) * 32, lane, ru, ri);
    }
    for (; grp < ngroups; grp += stride) {
        const int64_t base = grp * 32;
        const bool more = grp + stride < ngroups;
        pair_store_quarter<CLAMP>(s_x, lane, p0, q0);                      // quarter 0
        if constexpr (G > 0) pair_store_gmf_cols<G>(s_gp, lane, gu, gi, s_img + lds_img<L3>(IMG_WPG));
        __builtin_amdgcn_wave_barrier();
        pair_load_quarter<BIG>(a, nu, ni, lane, LOOK, p0, q0);             // quarter 2 (1) of THIS group
        int nu2 = 0, ni2 = 0;
        if (more) {
            pair_ids_clamp(a, (grp + stride) * 32, lane, ru, ri, nu2, ni2);     // loaded one group ago
            if constexpr (G > 0) pair_load_gmf<G, BIG>(a, nu2, ni2, lane, gu, gi);
            if (grp + 2 * stride < ngroups) pair_ids_raw(a, (grp + 2 * stride) * 32, lane, ru, ri);
        }
        const float* xrow = s_x + j * QSP + 4 * h;
        struct X1Quarter {
            const PairArgs& a;
            const float* xrow;
            float* s_x;
            f32x4* p0; f32x4* q0; f32x4* p1; f32x4* q1;
            int nu, ni, nu2, ni2;
            int lane;
            bool more;
            __device__ __forceinline__ void pre(int q) const
            {
                if constexpr (LOOK == 2) {
                    if (q == 4) {                 // quads 0..3 have all been read: quarter 1 in, quarter 3 on its way
                        pair_store_quarter<CLAMP>(s_x, lane, p1, q1);
                        __builtin_amdgcn_wave_barrier();
                        pair_load_quarter<BIG>(a, nu, ni, lane, 3, p1, q1);
                    } else if (q == 8) {          // quarter 2 in, the NEXT group's quarter 0 on its way
                        pair_store_quarter<CLAMP>(s_x, lane, p0, q0);
                        __builtin_amdgcn_wave_barrier();
                        if (more) pair_load_quarter<BIG>(a, nu2, ni2, lane, 0, p0, q0);
                    } else if (q == 12) {         // quarter 3 in, the next group's quarter 1 on its way
                        pair_store_quarter<CLAMP>(s_x, lane, p1, q1);
                        __builtin_amdgcn_wave_barrier();
                        if (more) pair_load_quarter<BIG>(a, nu2, ni2, lane, 1, p1, q1);
                    }
                } else {
                    if (q == 4 || q == 8) {       // quarter q / 4 in, the following one on its way
                        pair_store_quarter<CLAMP>(s_x, lane, p0, q0);
                        __builtin_amdgcn_wave_barrier();
                        pair_load_quarter<BIG>(a, nu, ni, lane, q / 4 + 1, p0, q0);
                    } else if (q == 12) {         // quarter 3 in, the NEXT group's quarter 0 on its way
                        pair_store_quarter<CLAMP>(s_x, lane, p0, q0);
                        __builtin_amdgcn_wave_barrier();
                        if (more) pair_load_quarter<BIG>(a, nu2, ni2, lane, 0, p0, q0);
                    }
                }
            }
            __device__ __forceinline__ f32x4 raw(int q) const { return *reinterpret_cast<const f32x4*>(xrow + 8 * (q & 3)); }
            __device__ __forceinline__ f32x4 finish(int, f32x4 v) const { return v; }
            __device__ __forceinline__ f32x4 passthrough(f32x4 v) const { return v; }
            __device__ __forceinline__ float between() const { return 0.f; }
        };
        float part = tower_mlp_part<L3>(s_img, lane, X1Quarter{a, xrow, s_x, p0, q0, p1, q1, nu, ni, nu2, ni2, lane, more});
        if constexpr (G > 0) part = gmf_half_pair<G>(s_gp, j, h) + part;
        const float total = part + __shfl_xor(part, 32) + a.bp;
        const int64_t b = base + j;
        if (h == 0 && b < a.B) a.out[b] = epilogue(total, a.act, a.lo, a.hi);
        __builtin_amdgcn_wave_barrier();
        nu = nu2; ni = ni2;
    }
}

// ---------------------------------------------------------------------------------------------
// GMF-only model (model_name == 'GMF', ncf_model.py:232-234): score = w . (eu * ei) + b
// ---------------------------------------------------------------------------------------------
struct GmfArgs {
    const float* EUg;
    const float* EIg;
    const float* wp;
    const int64_t* user;
    const int64_t* item;
    int64_t n_users, n_items, B;
    int G;
    float* out;
    float bp;
    int act;
    float lo, hi;
    int32_t* oob;
};

// 8 lanes per pair, each lane one float4 chunk per step: rows are read as whole 128-B lines.
__global__ __launch_bounds__(256) void gmf_pairs_kernel(GmfArgs a)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t b = t >> 3;
    const int sub = threadIdx.x & 7;
    float part = 0.f;
    if (b < a.B) {
        const int64_t u = clamp_id(a.user ? a.user[b] : b, a.n_users, sub == 0 ? a.oob : nullptr);
        const int64_t i = clamp_id(a.item ? a.item[b] : b, a.n_items, sub == 0 ? a.oob : nullptr);
        for (int c = sub; c < a.G / 4; c += 8) {
            const f32x4 gu = reinterpret_cast<const f32x4*>(a.EUg + u * a.G)[c];
            const f32x4 gi = reinterpret_cast<const f32x4*>(a.EIg + i * a.G)[c];
            const f32x4 w = reinterpret_cast<const f32x4*>(a.wp)[c];
            part = fmaf(gu.x * gi.x, w.x, part);
            part = fmaf(gu.y * gi.y, w.y, part);
            part = fmaf(gu.z * gi.z, w.z, part);
            part = fmaf(gu.w * gi.w, w.w, part);
        }
    }
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    part += __shfl_xor(part, 4);
    if (b < a.B && sub == 0) a.out[b] = epilogue(part + a.bp, a.act, a.lo, a.hi);
}

// matrix form: a block covers 256 items x 16 users; the 16 user rows are staged in LDS, each thread keeps one item and walks
// the users (its item row is re-read through L1 per user).
struct GmfTileArgs {
    const float* EUg;
    const float* EIg;
    const float* wp;
    const int64_t* users;
    const int64_t* cands;
    int64_t user_base, n_users, n_items, U, N;
    int G;
    float* out;
    int64_t ld_out;
    float bp;
    int act;
    float lo, hi;
    int32_t* oob;
};

__global__ __launch_bounds__(256) void gmf_tile_kernel(GmfTileArgs a)
{
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t u0 = (int64_t)blockIdx.y * 16;
    extern __shared__ __attribute__((aligned(16))) float s_u[];   // [16][G] raw user rows (the product is rounded before the
                                                                  // predict weight multiplies it, as in the reference)
    for (int i = threadIdx.x; i < 16 * a.G; i += 256) {
        const int64_t u = u0 + i / a.G;
        float v = 0.f;
        if (u < a.U) {
            const int64_t uid = clamp_id(a.users ? a.users[u] : a.user_base + u, a.n_users,
                                         (i % a.G == 0 && blockIdx.x == 0) ? a.oob : nullptr);
            v = a.EUg[uid * a.G + i % a.G];
        }
        s_u[i] = v;
    }
    __syncthreads();
    if (n >= a.N) return;
    const int64_t id = clamp_id(a.cands ? a.cands[n] : n, a.n_items, blockIdx.y == 0 ? a.oob : nullptr);
    const f32x4* gi = reinterpret_cast<const f32x4*>(a.EIg + id * a.G);
    // Same summation as gmf_pairs_kernel, so that forward(u, i) and score_matrix()[u, i] of one model carry the same bits: eight
    // chains (chunk c feeds chain c % 8, as the pair kernel's eight lanes do), folded pairwise like its xor-1 / 2 / 4 shuffles.
    for (int uu = 0; uu < 16 && u0 + uu < a.U; ++uu) {
        float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < a.G / 4; c0 += 8) {
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const int c = c0 + l;
                if (c >= a.G / 4) break;
                const f32x4 g = gi[c];
                const f32x4 gu = *reinterpret_cast<const f32x4*>(s_u + uu * a.G + 4 * c);
                const f32x4 w = reinterpret_cast<const f32x4*>(a.wp)[c];
                p[l] = fmaf(gu.x * g.x, w.x, p[l]);
                p[l] = fmaf(gu.y * g.y, w.y, p[l]);
                p[l] = fmaf(gu.z * g.z, w.z, p[l]);
                p[l] = fmaf(gu.w * g.w, w.w, p[l]);
            }
        }
        const float part = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        a.out[(u0 + uu) * a.ld_out + n] = epilogue(part + a.bp, a.act, a.lo, a.hi);
    }
}

// Layer-1 weight norms, once per handle: norms[f] = |W1u_f|_1, norms[H1 + f] = |W1i_f|_1, norms[2 H1 + f] = |b1_f|
// (W1 [H1, 2 dm] row-major, user half first; thread f owns output feature f).
__global__ __launch_bounds__(H1) void l1norms_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int dm, float* norms)
{
    const int f = threadIdx.x;
    float lu = 0.f, li = 0.f;
    for (int k = 0; k < dm; ++k) { lu += fabsf(W1[(int64_t)f * 2 * dm + k]); li += fabsf(W1[(int64_t)f * 2 * dm + dm + k]); }
    norms[f] = lu;
    norms[H1 + f] = li;
    norms[2 * H1 + f] = fabsf(b1[f]);
}

// {s, 1/s} from the bound  max_f (|W1u_f|_1 mu + |b1_f|) + max_f |W1i_f|_1 mi  on p + q (tower_shared.h), mu / mi = the largest finite
// |element| of the user / item rows.  ONE launch (it runs per call for the Hybrid head, whose rows are that call's latents):
// workgroups [0, gu) scan the user rows, the rest the item rows, each folds its maximum into scratch[0] / scratch[1] (a float's
// bit pattern) with one atomic; the workgroup that draws the last ticket (scratch[2]) reads both maxima back and writes
// the scale.  scratch[0..2] are zeroed by the launcher.  Only atomics carry data between workgroups.
__global__ __launch_bounds__(256) void l1scale_kernel(const float* __restrict__ u_rows, int64_t u_elems, const float* __restrict__ i_rows,
                                                      int64_t i_elems, unsigned gu, unsigned gi, const float* __restrict__ norms,
                                                      unsigned int* scratch, float* out)
{
    __shared__ unsigned s_ticket;
    __shared__ float s_red[2][2];
    __shared__ float s_max[4];
    const int tid = threadIdx.x;
    const bool user_side = blockIdx.x < gu;
    const unsigned b = user_side ? blockIdx.x : blockIdx.x - gu, g = user_side ? gu : gi;
    float m = 0.f;
    if (b < g) {
        const float* x = user_side ? u_rows : i_rows;
        const int64_t n = user_side ? u_elems : i_elems;
        const int64_t n4 = n / 4;
        for (int64_t i = (int64_t)b * 256 + tid; i < n4; i += (int64_t)g * 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float av = fabsf(v[c]);
                if (av > m && av < __builtin_inff()) m = av;
            }
        }
        for (int64_t i = n4 * 4 + (int64_t)b * 256 + tid; i < n; i += (int64_t)g * 256) {
            const float av = fabsf(x[i]);
            if (av > m && av < __builtin_inff()) m = av;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    }
    if ((tid & 63) == 0) s_max[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {                                            // one atomic per workgroup, then its ticket
        m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        if (m > 0.f) atomicMax(scratch + (user_side ? 0 : 1), __float_as_uint(m));
        __threadfence();
        s_ticket = atomicAdd(scratch + 2, 1u);
    }
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;                     // workgroup-uniform
    __threadfence();
    float a = 0.f, c = 0.f;
    if (tid < H1) {
        const float mu = __uint_as_float(__hip_atomic_load(scratch + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const float mi = __uint_as_float(__hip_atomic_load(scratch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        a = norms[tid] * mu + norms[2 * H1 + tid];
        c = norms[H1 + tid] * mi;
        a = (a == a && a < __builtin_inff()) ? a : 0.f;
        c = (c == c && c < __builtin_inff()) ? c : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a = fmaxf(a, __shfl_xor(a, o)); c = fmaxf(c, __shfl_xor(c, o)); }
    static_assert(H1 == 128, "two waves hold the H1 features");
    if ((tid & 63) == 0 && tid < H1) { s_red[tid >> 6][0] = a; s_red[tid >> 6][1] = c; }
    __syncthreads();
    if (tid == 0) {
        const float ma = fmaxf(s_red[0][0], s_red[1][0]), mc = fmaxf(s_red[0][1], s_red[1][1]);
        const Layer1Scale sc = layer1_scale_for((ma + mc) * 1.001f);      // margin for the roundings of this estimate
        out[0] = sc.s;
        out[1] = sc.inv;
    }
}

// *stale = 1 when the scale the tables need NOW (fresh) is smaller than the one the handle was built with (stored)
__global__ void scale_compare_kernel(const float* stored, const float* fresh, int32_t* stale)
{
    if (threadIdx.x == 0 && fresh[0] < stored[0]) *stale = 1;
}

template <int G, bool L3>
int launch_tile_t(const TileArgs& a, hipStream_t s)
{
    // with the P rows in LDS the kernel needs 143 VGPRs: three waves per SIMD (768 threads, 24 users per tile, LDS 152 KB)
    // measured 3.35 -> 3.32 ms; the Hybrid head (64-wide GMF tile) has LDS room for 16 users per tile: 512 threads
    constexpr int THREADS = 768, TUBX = 24;
    constexpr size_t lds = (size_t)(LDS_IMG_SIZE<L3> + 2 * (TI * QS + (G > 0 ? TI * (G + 4) : 0)) + 2 * TUBX * H1) * sizeof(float);
    static_assert(lds <= 160 * 1024, "tile kernel LDS budget");
    static bool attr[RECO_MAX_DEVICES] = {};
    if (int rc = reco_allow_lds(reinterpret_cast<const void*>(&tower_tile_kernel<G, L3, THREADS, TUBX>), lds, attr)) return rc;
    static int n_cu = 0;                                  // one GPU model per node: the CU count is the same on every device
    if (n_cu == 0) {
        int dev = 0;
        RECO_CHECK_HIP(hipGetDevice(&dev));
        RECO_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        if (n_cu <= 0) n_cu = 256;
    }
    const int64_t tiles = ceil_div64(a.N, TI) * ceil_div64(a.U, TUBX);
    const unsigned grid = tile_grid(tiles, n_cu, a.reserve_cus);      // one resident workgroup per CU
    hipLaunchKernelGGL((tower_tile_kernel<G, L3, THREADS, TUBX>), dim3(grid), dim3(THREADS), lds, s, a);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

template <int G, bool L3, int SLOTS>
int launch_tile_topk_t(const TileArgs& a, const TopkFuse& f, hipStream_t s)
{
    constexpr int THREADS = 768, TUBX = 24;     // as launch_tile_t
    // image | two item-side buffers | one block of P rows | one 64-key ring per user (the same total as launch_tile_t's)
    constexpr size_t lds = (size_t)(LDS_IMG_SIZE<L3> + 2 * (TI * QS + (G > 0 ? TI * (G + 4) : 0)) + TUBX * H1) * sizeof(float) +
                           (size_t)TUBX * 64 * sizeof(unsigned long long);
    static_assert(lds <= 160 * 1024, "tile kernel LDS budget");
    static bool attr[RECO_MAX_DEVICES] = {};
    if (int rc = reco_allow_lds(reinterpret_cast<const void*>(&tower_tile_topk_kernel<G, L3, THREADS, TUBX, SLOTS>), lds, attr)) return rc;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        RECO_CHECK_HIP(hipGetDevice(&dev));
        RECO_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        if (n_cu <= 0) n_cu = 256;
    }
    const int64_t tiles = ceil_div64(a.N, TI) * ceil_div64(a.U, TUBX);
    const unsigned grid = tile_grid(tiles, n_cu, a.reserve_cus);      // one resident workgroup per CU
    hipLaunchKernelGGL((tower_tile_topk_kernel<G, L3, THREADS, TUBX, SLOTS>), dim3(grid), dim3(THREADS), lds, s, a, f);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

template <int G, bool L3, bool BIG, bool CLAMP>
int launch_pairs_tb(const PairArgs& a, hipStream_t s)
{
    // 8 waves per workgroup when their half-row staging areas fit next to the image (G <= 32: 148 KB), else 4 (six waves of the
    // Hybrid shape fit since its LDS image dropped layer 3, but measured slower: 1M pair rows 1.11 -> 1.19 ms)
    if constexpr (G <= 32) {
        // Quarter-staged form, one quarter in flight: 168 VGPRs, so twelve waves (three per SIMD) share the image (152.5 KB).
        // Same-box A/B against the half-staged kernel below (RECO_HIP_LIB variants, 11.5 M protocol pairs / random users):
        //   half rows, 8 waves (256 VGPRs)        1.90 / 1.98 ms
        //   quarters, two in flight, 8 waves      1.88 / 1.96 ms   (194 VGPRs)
        //   quarters, one in flight, 8 waves      1.87 / 1.96 ms
        //   quarters, one in flight, 12 waves     1.79 / 1.90 ms   <- this
        // Tables of 2^23 rows or more (64-bit row offsets: 12 waves would spill 60 VGPRs) run it with 8 waves: 9 M users x 100 k
        // items, 11.5 M pairs from HBM-resident tables (tools/big_pairs_bench.py): half rows 2.56 / 2.16 ms (random users / runs of
        // 100 per user), quarters two in flight 2.38 / 1.99, one in flight 2.34 / 1.93 <- this, 12 waves 2.80 / 2.21.
        // The Hybrid shape (its 64-wide GMF rows fill the LDS at 8 waves; measured level) stays on the half-staged kernel.
        constexpr int WAVES = BIG ? 8 : 12, LOOK = 1;
        constexpr size_t lds = (size_t)(LDS_IMG_SIZE<L3> + WAVES * (32 * QSP + 32 * (G / 4))) * sizeof(float);
        static_assert(lds <= 160 * 1024, "pair kernel LDS budget");
        static bool attr[RECO_MAX_DEVICES] = {};
        if (int rc = reco_allow_lds(reinterpret_cast<const void*>(&tower_pairs_q_kernel<G, L3, WAVES, BIG, CLAMP, LOOK>), lds, attr)) return rc;
        const int64_t nchunks = ceil_div64(a.B, 32 * WAVES);
        const unsigned grid = (unsigned)(nchunks < 256 ? nchunks : 256);   // persistent: one workgroup per CU
        hipLaunchKernelGGL((tower_pairs_q_kernel<G, L3, WAVES, BIG, CLAMP, LOOK>), dim3(grid), dim3(64 * WAVES), lds, s, a);
        RECO_CHECK_LAUNCH();
        return RECO_OK;
    } else {
        constexpr int WAVES = G <= 32 ? 8 : 4;
        constexpr size_t lds = (size_t)(LDS_IMG_SIZE<L3> + WAVES * (32 * HS + 32 * (G / 4))) * sizeof(float);
        static_assert(lds <= 160 * 1024, "pair kernel LDS budget");
        static bool attr[RECO_MAX_DEVICES] = {};
        if (int rc = reco_allow_lds(reinterpret_cast<const void*>(&tower_pairs_kernel<G, L3, WAVES, BIG, CLAMP>), lds, attr)) return rc;
        const int64_t nchunks = ceil_div64(a.B, 32 * WAVES);
        const unsigned grid = (unsigned)(nchunks < 256 ? nchunks : 256);   // persistent: one workgroup per CU
        hipLaunchKernelGGL((tower_pairs_kernel<G, L3, WAVES, BIG, CLAMP>), dim3(grid), dim3(64 * WAVES), lds, s, a);
        RECO_CHECK_LAUNCH();
        return RECO_OK;
    }
}

template <int G, bool L3>
int launch_pairs_t(const PairArgs& a, hipStream_t s)
{
    // 32-bit row offsets (saddr loads) whenever every table is under 2^23 rows; 64-bit offsets otherwise (tower_shared.h)
    const bool big = a.n_users >= (1ll << 23) || a.n_items >= (1ll << 23);
    if (a.l1) return big ? launch_pairs_tb<G, L3, true, true>(a, s) : launch_pairs_tb<G, L3, false, true>(a, s);
    return big ? launch_pairs_tb<G, L3, true, false>(a, s) : launch_pairs_tb<G, L3, false, false>(a, s);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host-side entry points used by api.hip
// ---------------------------------------------------------------------------------------------
int tower_pack_image(const float* W2, const float* b2, const float* W3, const float* b3, const float* wp,
                     int G, int has_l3, float* img, hipStream_t s)
{
    PackArgs p{W2, b2, W3, b3, wp, G, has_l3};
    hipLaunchKernelGGL(pack_image_kernel, dim3((IMG_SIZE + 255) / 256), dim3(256), 0, s, p, img);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

int tower_launch_tile(int G, int has_l3, const float* img, const float* P, const float* Q,
                      const float* EUg, const float* EIg, const int64_t* users, const int64_t* cands,
                      int64_t user_base, int64_t n_users, int64_t n_items, int64_t U, int64_t N,
                      float* out, int64_t ld_out, float bp, int act, float lo, float hi, int32_t* oob,
                      unsigned int* next_tile, const float* l1, int reserve_cus, hipStream_t s)
{
    if (U <= 0 || N <= 0) return RECO_OK;
    RECO_REQUIRE(l1, "tower tile kernel: the layer-1 scale is missing");
    TileArgs a{img, P, Q, EUg, EIg, users, cands, user_base, n_users, n_items, U, N, out, ld_out, bp, act, lo, hi, oob, next_tile, l1, reserve_cus};
    if (G == 32 && has_l3) return launch_tile_t<32, true>(a, s);
    if (G == 0 && has_l3) return launch_tile_t<0, true>(a, s);
    if (G == 64 && !has_l3) return launch_tile_t<64, false>(a, s);
    reco_set_error("tower tile kernel: unsupported configuration G=%d has_l3=%d", G, has_l3);
    return RECO_EINVAL;
}

int tower_launch_pairs(int G, int has_l3, const float* img, const float* P, const float* Q,
                       const float* EUg, const float* EIg, const int64_t* user, const int64_t* item,
                       int64_t n_users, int64_t n_items, int64_t B, float* out, float bp, int act,
                       float lo, float hi, int32_t* oob, const float* l1, hipStream_t s)
{
    if (B <= 0) return RECO_OK;      // l1 null: P and Q are unscaled, layer 1 is add + max
    PairArgs a{img, P, Q, EUg, EIg, user, item, n_users, n_items, B, out, bp, act, lo, hi, oob, l1};
    if (G == 32 && has_l3) return launch_pairs_t<32, true>(a, s);
    if (G == 0 && has_l3) return launch_pairs_t<0, true>(a, s);
    if (G == 64 && !has_l3) return launch_pairs_t<64, false>(a, s);
    reco_set_error("tower pair kernel: unsupported configuration G=%d has_l3=%d", G, has_l3);
    return RECO_EINVAL;
}

int gmf_launch_pairs(const float* EUg, const float* EIg, const float* wp, const int64_t* user,
                     const int64_t* item, int64_t n_users, int64_t n_items, int64_t B, int G, float* out,
                     float bp, int act, float lo, float hi, int32_t* oob, hipStream_t s)
{
    if (B <= 0) return RECO_OK;
    GmfArgs a{EUg, EIg, wp, user, item, n_users, n_items, B, G, out, bp, act, lo, hi, oob};
    const int64_t blocks = ceil_div64(B * 8, 256);
    RECO_REQUIRE(blocks < (1ll << 31), "gmf pair grid too large");
    hipLaunchKernelGGL(gmf_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

int gmf_launch_tile(const float* EUg, const float* EIg, const float* wp, const int64_t* users,
                    const int64_t* cands, int64_t user_base, int64_t n_users, int64_t n_items, int64_t U,
                    int64_t N, int G, float* out, int64_t ld_out, float bp, int act, float lo, float hi,
                    int32_t* oob, hipStream_t s)
{
    if (U <= 0 || N <= 0) return RECO_OK;
    GmfTileArgs a{EUg, EIg, wp, users, cands, user_base, n_users, n_items, U, N, G, out, ld_out, bp, act, lo, hi, oob};
    const int64_t by = ceil_div64(U, 16);
    RECO_REQUIRE(by <= 65535, "gmf tile: too many users per call (%lld); chunk the call", (long long)U);
    hipLaunchKernelGGL(gmf_tile_kernel, dim3((unsigned)ceil_div64(N, 256), (unsigned)by), dim3(256),
                       (size_t)16 * G * sizeof(float), s, a);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

int tower_launch_l1norms(const float* W1, const float* b1, int dm, float* norms, hipStream_t s)
{
    hipLaunchKernelGGL(l1norms_kernel, dim3(1), dim3(H1), 0, s, W1, b1, dm, norms);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

// out[0..1] = {s, 1/s} for layer 1 over the given user / item rows (tower_shared.h); norms: tower_launch_l1norms of the handle's
// layer 1; scratch: 3 words inside [clear, clear + clear_bytes), the region zeroed here first (a caller whose workspace head holds
// other per-launch counters has them cleared by the same memset node).  One memset node and one launch on `s`; nothing is read back.
int tower_launch_l1scale(const float* norms, const float* u_rows, int64_t u_elems, const float* i_rows, int64_t i_elems,
                         unsigned int* scratch, float* out, void* clear, size_t clear_bytes, hipStream_t s)
{
    RECO_REQUIRE((char*)clear <= (char*)scratch && (char*)(scratch + 3) <= (char*)clear + clear_bytes, "internal: l1 scratch outside the cleared region");
    RECO_CHECK_HIP(hipMemsetAsync(clear, 0, clear_bytes, s));
    auto side_blocks = [](int64_t elems) -> unsigned {
        if (elems <= 0) return 0u;
        const int64_t blocks = ceil_div64(elems, 256 * 16);
        return (unsigned)(blocks < 512 ? blocks : 512);
    };
    const unsigned gu = side_blocks(u_elems), gi = side_blocks(i_elems);
    const unsigned grid = gu + gi > 0 ? gu + gi : 1u;
    hipLaunchKernelGGL(l1scale_kernel, dim3(grid), dim3(256), 0, s, u_rows, u_elems, i_rows, i_elems, gu, gi, norms, scratch, out);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

int tower_launch_scale_compare(const float* stored, const float* fresh, int32_t* stale, hipStream_t s)
{
    hipLaunchKernelGGL(scale_compare_kernel, dim3(1), dim3(64), 0, s, stored, fresh, stale);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

// ---- fused score + top-k (tower_tile_topk_kernel) ----
int tower_tile_users_per_block(int G) { (void)G; return 24; }

// counters: [fresh | pad to 64 words | next_tile[blocks] | visits[blocks]], zeroed by the caller together with `partial`
size_t tower_topk_counter_bytes(int G, int64_t U)
{
    return align_up((size_t)(64 + 2 * ceil_div64(U, tower_tile_users_per_block(G))) * sizeof(unsigned int), 256);
}

int tower_launch_tile_topk(int G, int has_l3, const float* img, const float* P, const float* Q,
                           const float* EUg, const float* EIg, const int64_t* users, const int64_t* cands,
                           int64_t user_base, int64_t n_users, int64_t n_items, int64_t U, int64_t N,
                           float bp, int act, float lo, float hi, int32_t* oob,
                           int k, int v_max, unsigned int* counters, unsigned long long* partial,
                           const float* l1, int reserve_cus, hipStream_t s)
{
    if (U <= 0 || N <= 0) return RECO_OK;
    RECO_REQUIRE(l1, "fused top-k tile kernel: the layer-1 scale is missing");
    RECO_REQUIRE(k >= 1 && k <= 128 && v_max >= 1 && v_max <= 1024, "fused top-k: k=%d / v_max=%d out of range", k, v_max);
    RECO_REQUIRE(N < (1ll << 31) && ceil_div64(U, 16) < (1ll << 21), "fused top-k: problem too large for one launch");
    TileArgs a{img, P, Q, EUg, EIg, users, cands, user_base, n_users, n_items, U, N, nullptr, 0, bp, act, lo, hi, oob, nullptr, l1, reserve_cus};
    const int64_t blocks = ceil_div64(U, tower_tile_users_per_block(G));
    TopkFuse f{k, v_max, counters, counters + 64, counters + 64 + blocks, partial};
    if (k <= 64) {
        if (G == 32 && has_l3) return launch_tile_topk_t<32, true, 1>(a, f, s);
        if (G == 0 && has_l3) return launch_tile_topk_t<0, true, 1>(a, f, s);
        if (G == 64 && !has_l3) return launch_tile_topk_t<64, false, 1>(a, f, s);
    } else {
        if (G == 32 && has_l3) return launch_tile_topk_t<32, true, 2>(a, f, s);
        if (G == 0 && has_l3) return launch_tile_topk_t<0, true, 2>(a, f, s);
        if (G == 64 && !has_l3) return launch_tile_topk_t<64, false, 2>(a, f, s);
    }
    reco_set_error("fused top-k tile kernel: unsupported configuration G=%d has_l3=%d", G, has_l3);
    return RECO_EINVAL;
}
