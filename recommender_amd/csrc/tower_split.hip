// K-b, split-operand form: the same tower as tower.hip, with every fp32 MFMA operand split into three bf16 pieces and
// the products formed on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  Opt-in (RECO_PREC_SPLIT_BF16X3); the exact fp32
// MFMA kernels of tower.hip stay the default.
//
// Why: on gfx950 the bf16 MFMA retires 16x the FLOPs of the fp32 MFMA per cycle.  An fp32 value x is EXACTLY
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (8 + 8 + 8 significand bits,
// round-to-nearest each, residuals exact in fp32), so  w * x = sum of nine bf16 x bf16 products, each exact in the
// MFMA's fp32 multiplier.  We keep the six whose magnitude can reach 2^-24 |w x|:
//     hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid
// and drop mid*lo, lo*mid, lo*lo (<= 2^-24, 2^-24, 2^-32 relative).  What remains is an fp32 dot product whose only
// other rounding is the fp32 accumulation, as in the exact kernel: measured deviation from an fp64 reference is the
// same as a plain fp32 fmaf chain's (tools/split_probe.hip: 6.3e-7 vs 6.7e-7 at |score| <= 2.9), and the kernel passes
// the same parity gates (tests/test_gpu_parity.py).  Inf/NaN: an Inf operand has hi = Inf, mid = NaN -> NaN where the
// exact kernel would give Inf; finite inputs only is the contract of this mode.
//
// Cost per 32 pairs and wave: 96 + 24 bf16 MFMAs (32 cycles each) instead of 128 + 32 fp32 MFMAs (64 cycles each),
// plus ~11 VALU instructions per pair of elements for the splitting of the activations (weights are split once, when
// the image is packed).  The splitting of k-step s+1 is interleaved, 5 instructions per MFMA, with the MFMAs of
// k-step s: an MFMA holds the SIMD's issue port for 8 of its 32 cycles, the rest is free for VALU.  Measured
// (tools/split_probe.hip): the chip runs this instruction mix at 1.74-1.89 GHz (power), 44-48 cycles per MFMA slot.
//
// Operand maps (v_mfma_f32_32x32x16_bf16): lane l = (r = l & 31, h = l >> 5) holds A[row r][k = 8h + e] and
// B[k = 8h + e][col r], e = 0..7; D as in tower.hip.  k-step s of layer 2 carries features 16 s + 8 h + e; k-step
// s' = 2 t + g of layer 3 carries the features held by registers 8 g .. 8 g + 7 of layer 2's accumulator t, so the
// accumulators are layer 3's B operand without any lane movement, exactly as in the exact kernel.
#include "tower_shared.h"

#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// split image: [W2: t(2) s(8) piece(3) lane(64)] [W3: s'(4) piece(3) lane(64)] in 16-byte units (8 bf16), then the
// fp32 tail of the exact image (b2, b3, predict weights), copied verbatim.  W3 is absent when the tower has no layer 3.
constexpr int SPL_W2_U = 2 * 8 * 3 * 64;
constexpr int SPL_W3_U = 4 * 3 * 64;
constexpr int SPL_TAIL = IMG_SIZE - IMG_B2;                         // 224 floats
constexpr int TAIL_B2 = 0, TAIL_B3 = IMG_B3 - IMG_B2, TAIL_WPM = IMG_WPM - IMG_B2, TAIL_WPG = IMG_WPG - IMG_B2;
constexpr int spl_w_units(bool l3) { return SPL_W2_U + (l3 ? SPL_W3_U : 0); }
constexpr int spl_bytes(bool l3) { return spl_w_units(l3) * 16 + SPL_TAIL * 4; }

__device__ __forceinline__ unsigned pk_bf16(float a, float b)       // one v_cvt_pk_bf16_f32 (RNE)
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

__device__ __forceinline__ void split3(float w, unsigned short* o)
{
    const unsigned a = pk_bf16(w, 0.f) & 0xffffu;
    const float r = w - __uint_as_float(a << 16);
    const unsigned b = pk_bf16(r, 0.f) & 0xffffu;
    const float q = r - __uint_as_float(b << 16);
    o[0] = (unsigned short)a; o[1] = (unsigned short)b; o[2] = (unsigned short)(pk_bf16(q, 0.f) & 0xffffu);
}

struct PackSplitArgs {
    const float* W2;     // [64,128]
    const float* W3;     // [32,64] or null
    const float* img;    // the exact image (for its fp32 tail)
    int has_l3;
};

__global__ void pack_split_kernel(PackSplitArgs a, unsigned char* __restrict__ out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (unit, element) of the hi piece
    const int units = a.has_l3 ? (2 * 8 + 4) * 64 : 2 * 8 * 64;     // (t,s) or s' blocks x lanes
    unsigned short* w = reinterpret_cast<unsigned short*>(out);
    if (idx < units * 8) {
        const int e = idx & 7, lane = (idx >> 3) & 63, blk = idx >> 9;
        const int r = lane & 31, h = lane >> 5;
        float v;
        size_t base;                                                 // unit index of the hi piece
        if (blk < 16) {
            const int t = blk >> 3, s = blk & 7;
            v = a.W2[(32 * t + r) * H1 + 16 * s + 8 * h + e];
            base = (size_t)((t * 8 + s) * 3) * 64 + lane;
        } else {
            const int s = blk - 16, t = s >> 1, g = s & 1;
            v = a.W3[r * H2 + 32 * t + (e & 3) + 8 * (2 * g + (e >> 2)) + 4 * h];
            base = (size_t)SPL_W2_U + (size_t)(s * 3) * 64 + lane;
        }
        unsigned short o[3];
        split3(v, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) w[(base + (size_t)c * 64) * 8 + e] = o[c];
    }
    if (idx < SPL_TAIL)
        reinterpret_cast<float*>(out + (size_t)spl_w_units(a.has_l3) * 16)[idx] = a.img[IMG_B2 + idx];
}

// ---------------------------------------------------------------------------------------------
// the split body
// ---------------------------------------------------------------------------------------------
struct Frag3 { bf16x8 hi, mid, lo; };
struct Build { u32x4 H, M, L; };

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ Frag3 finish(const Build& bd)
{
    Frag3 f;
    f.hi = __builtin_bit_cast(bf16x8, bd.H); f.mid = __builtin_bit_cast(bf16x8, bd.M); f.lo = __builtin_bit_cast(bf16x8, bd.L);
    return f;
}

// The same splitting as nine micro-op groups, so that a slot can mix groups of two different element pairs: a slot's
// instructions are then mostly independent of each other instead of one three-deep dependent chain .
struct Pair9 { float xa, xb, ta, tb; };
template <bool RELU>
__device__ __forceinline__ void mop(Build& bd, Pair9& st, int pr, int m, float pre_a, float pre_b)
{
    switch (m) {
    case 0: st.xa = pre_a; st.xb = pre_b; break;                                   // (the provider's add, if any)
    case 1: if (RELU) { st.xa = relu1(st.xa); st.xb = relu1(st.xb); } break;
    case 2: bd.H[pr] = pk_bf16(st.xa, st.xb); break;
    case 3: st.ta = lo_f(bd.H[pr]); st.tb = hi_f(bd.H[pr]); break;
    case 4: st.xa -= st.ta; st.xb -= st.tb; break;
    case 5: bd.M[pr] = pk_bf16(st.xa, st.xb); break;
    case 6: st.ta = lo_f(bd.M[pr]); st.tb = hi_f(bd.M[pr]); break;
    case 7: st.xa -= st.ta; st.xb -= st.tb; break;
    default: bd.L[pr] = pk_bf16(st.xa, st.xb); break;
    }
}
struct W6 { bf16x8 h0, m0, l0, h1, m1, l1; };
__device__ __forceinline__ W6 load_w2(const bf16x8* s_w, int s, int lane)
{
    const bf16x8* a = s_w + ((0 * 8 + s) * 3) * 64 + lane;
    const bf16x8* b = s_w + ((1 * 8 + s) * 3) * 64 + lane;
    return W6{a[0], a[64], a[128], b[0], b[64], b[128]};
}
struct W3r { bf16x8 h, m, l; };
__device__ __forceinline__ W3r load_w3(const bf16x8* s_w, int s, int lane)
{
    const bf16x8* a = s_w + SPL_W2_U + (s * 3) * 64 + lane;
    return W3r{a[0], a[64], a[128]};
}

// kept cross term t of one accumulator, smallest magnitudes first
__device__ __forceinline__ f32x16 term(int t, const bf16x8& wh, const bf16x8& wm, const bf16x8& wl, const Frag3& x, f32x16 acc)
{
    switch (t) {
    case 0: return MFMA_BF16(wl, x.hi, acc);
    case 1: return MFMA_BF16(wh, x.lo, acc);
    case 2: return MFMA_BF16(wm, x.mid, acc);
    case 3: return MFMA_BF16(wm, x.hi, acc);
    case 4: return MFMA_BF16(wh, x.mid, acc);
    default: return MFMA_BF16(wh, x.hi, acc);
    }
}

struct X8 { f32x4 a, b; };

// Layers 2 (+3) and the MLP half of the predict layer for the 32 pairs of this wave.
// x1.raw(s) issues the LDS reads of this lane's 8 layer-1 inputs [16 s + 8 h .. + 7] of its pair, x1.pre(s, raw, e) is
// element e before the ReLU (P + Q), X1::RELU says whether the ReLU is still to be applied (false when the staged tile
// already holds activations).
//
// Schedule: the MFMAs of k-step s run beside the splitting of k-step s + 1's operands.  A "slot" is one MFMA plus the
// VALU groups listed by slot6(): two element pairs advance together, so the instructions of a slot are mostly
// independent of each other; six slots take two pairs through all nine groups.
template <bool L3, typename X1>
__device__ __forceinline__ float split_mlp_part(const bf16x8* s_w, const float* s_tail, int lane, X1 x1)
{
    const int h = lane >> 5;
    f32x16 acc0 = bias16(s_tail + TAIL_B2, h);
    f32x16 acc1 = bias16(s_tail + TAIL_B2 + 32, h);

    Build bd;
    Pair9 st[4];
    // slot k (0..5) for the element pairs A, B:   k even: g(A) g(B) g'(A)    k odd: g(B) g'(A) g'(B)
    auto slot6 = [&](auto relu_tag, int k, int A, int B, auto pre) {
        constexpr bool R = decltype(relu_tag)::value;
        constexpr int g0[6] = {0, 1, 3, 4, 6, 7}, g1[6] = {0, 2, 3, 5, 6, 8}, g2[6] = {1, 2, 4, 5, 7, 8};
        if ((k & 1) == 0) {
            mop<R>(bd, st[A], A, g0[k], pre(A, 0), pre(A, 1));
            mop<R>(bd, st[B], B, g1[k], pre(B, 0), pre(B, 1));
            mop<R>(bd, st[A], A, g2[k], 0.f, 0.f);
        } else {
            mop<R>(bd, st[B], B, g0[k], 0.f, 0.f);
            mop<R>(bd, st[A], A, g1[k], 0.f, 0.f);
            mop<R>(bd, st[B], B, g2[k], 0.f, 0.f);
        }
    };
    using relu_x1 = std::integral_constant<bool, X1::RELU != 0>;
    using relu_yes = std::integral_constant<bool, true>;

    X8 q = x1.raw(0);
    {
        auto pre0 = [&](int pr, int w) { return x1.pre(0, q, 2 * pr + w); };
#pragma unroll
        for (int k = 0; k < 6; ++k) { slot6(relu_x1{}, k, 0, 1, pre0); slot6(relu_x1{}, k, 2, 3, pre0); }   // k-step 0: nothing to hide behind
    }
    Frag3 f = finish(bd);
    W6 w = load_w2(s_w, 0, lane);
    q = x1.raw(1);
    FENCE();
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const W6 wn = load_w2(s_w, s + 1, lane);
        X8 qn = q;
        if (s + 2 < 8) {
            x1.stage(s + 2);                 // (quarter-staged pair kernel: k-steps 2c, 2c + 1 are staged right before raw(2c))
            qn = x1.raw(s + 2);
        }
        FENCE();
        auto pre = [&](int pr, int wh) { return x1.pre(s + 1, q, 2 * pr + wh); };
#pragma unroll
        for (int i = 0; i < 12; ++i) {       // slot i: MFMA (accumulator i & 1, term i >> 1)
            if (i & 1) acc1 = term(i >> 1, w.h1, w.m1, w.l1, f, acc1);
            else acc0 = term(i >> 1, w.h0, w.m0, w.l0, f, acc0);
            slot6(relu_x1{}, i % 6, 2 * (i / 6), 2 * (i / 6) + 1, pre);
            FENCE();
        }
        f = finish(bd); w = wn; q = qn;
    }
    float part = 0.f;
    if constexpr (L3) {
        // k-step 7: accumulator 0's six terms first, then accumulator 1's beside the split of acc0[0..7] (layer 3, k-step 0)
        W3r w3 = load_w3(s_w, 0, lane);
        FENCE();
#pragma unroll
        for (int i = 0; i < 6; ++i) { acc0 = term(i, w.h0, w.m0, w.l0, f, acc0); FENCE(); }
        {
            auto pre = [&](int pr, int wh) { return acc0[2 * pr + wh]; };
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                acc1 = term(i, w.h1, w.m1, w.l1, f, acc1);
                slot6(relu_yes{}, i, 0, 1, pre);
                slot6(relu_yes{}, i, 2, 3, pre);
                FENCE();
            }
        }
        f = finish(bd);
        f32x16 acc3 = bias16(s_tail + TAIL_B3, h);
#pragma unroll
        for (int s3 = 0; s3 < 4; ++s3) {
            W3r wn3 = w3;
            if (s3 + 1 < 4) wn3 = load_w3(s_w, s3 + 1, lane);
            FENCE();
            auto pre = [&](int pr, int wh) {
                const int r = 8 * ((s3 + 1) & 1) + 2 * pr + wh;
                return s3 + 1 < 2 ? acc0[r] : acc1[r];
            };
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                acc3 = term(i, w3.h, w3.m, w3.l, f, acc3);
                if (s3 + 1 < 4) { slot6(relu_yes{}, i, 0, 1, pre); slot6(relu_yes{}, i, 2, 3, pre); }
                FENCE();
            }
            if (s3 + 1 < 4) f = finish(bd);
            w3 = wn3;
        }
        part = dot16_relu(acc3, s_tail + TAIL_WPM, h, part);
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (i & 1) acc1 = term(i >> 1, w.h1, w.m1, w.l1, f, acc1);
            else acc0 = term(i >> 1, w.h0, w.m0, w.l0, f, acc0);
        }
        part = dot16_relu(acc0, s_tail + TAIL_WPM, h, part);
        part = dot16_relu(acc1, s_tail + TAIL_WPM + 32, h, part);
    }
    return part;
}

__device__ __forceinline__ float x8_get(const X8& q, int e)
{
    return e < 4 ? q.a[e] : q.b[e - 4];
}

// ---------------------------------------------------------------------------------------------
// tile kernel (same tiling, staging and persistence as tower_tile_kernel)
// ---------------------------------------------------------------------------------------------
template <int G, bool L3>
__global__ __launch_bounds__(TILE_THREADS, 2) void tower_tile_split_kernel(TileArgs a)
{
    constexpr int GS = G + 4;
    constexpr int BUF = TI * QS + (G > 0 ? TI * GS : 0);
    constexpr int WU = spl_w_units(L3);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bf16x8* s_w = reinterpret_cast<const bf16x8*>(smem);
    const float* s_tail = smem + WU * 4;
    float* s_buf = smem + WU * 4 + SPL_TAIL;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const int64_t tiles_n = (a.N + TI - 1) / TI;
    const int64_t tiles = tiles_n * ((a.U + TUB - 1) / TUB);

    for (int i = tid; i < spl_bytes(L3) / 16; i += TILE_THREADS)
        reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(a.img)[i];

    __shared__ long long s_next[2];
    TileStage<G> st;
    int64_t tile = blockIdx.x;
    tile_queue_fetch(a, tid, tile + gridDim.x, &s_next[0]);
    if (tile < tiles) {
        tile_stage_load<G>(a, (tile % tiles_n) * TI, tile / tiles_n == 0, tid, st);
        tile_stage_store<G>(s_buf, s_buf + TI * QS, tid, st);
    }
    __syncthreads();
    int64_t next = s_next[0];

    for (int it = 0; tile < tiles; ++it) {
        const float* s_q = s_buf + (it & 1) * BUF;
        const float* s_g = s_q + TI * QS;
        const int64_t n0 = (tile % tiles_n) * TI;
        const int64_t u0 = (tile / tiles_n) * TUB;
        const int g_n = a.N - n0 > 32 ? TI / 32 : 1;           // the ragged last tile: no MFMAs for a column group past the catalog
        tile_queue_fetch(a, tid, next + gridDim.x, &s_next[(it + 1) & 1]);     // the tile after next
        if (next < tiles) tile_stage_load<G>(a, (next % tiles_n) * TI, next / tiles_n == 0, tid, st);

        for (int uu = wave; uu < TUB; uu += TILE_THREADS / 64) {
            const int64_t u = u0 + uu;
            if (u >= a.U) break;
            f32x4 p[16];                                  // p[2 s], p[2 s + 1] = P[u][16 s + 8 h .. + 7]
            const f32x4* prow = reinterpret_cast<const f32x4*>(a.P + u * H1 + 8 * h);
#pragma unroll
            for (int s = 0; s < 8; ++s) { p[2 * s] = prow[4 * s]; p[2 * s + 1] = prow[4 * s + 1]; }
            // the user's GMF row, pre-multiplied by the predict weights: registers for G = 32; for G = 64 (Hybrid head) the
            // 32 VGPRs would spill, so the wave parks the row in its own LDS slot (same-wave LDS ops execute in order)
            constexpr bool GU_LDS = G > 32;
            f32x4 gu[(G > 0 && !GU_LDS) ? G / 8 : 1];
            float* s_gu = s_buf + 2 * BUF + wave * (G > 0 ? G : 1);
            if constexpr (G > 0) {
                const int64_t uid = clamp_id(a.users ? a.users[u] : a.user_base + u, a.n_users,
                                             (lane == 0 && n0 == 0) ? a.oob : nullptr);
                const f32x4* gr = reinterpret_cast<const f32x4*>(a.EUg + uid * G + h * (G / 2));
                const float* wpg = s_tail + TAIL_WPG + h * (G / 2);
#pragma unroll
                for (int c = 0; c < G / 8; ++c) {
                    const f32x4 v = gr[c] * *reinterpret_cast<const f32x4*>(wpg + 4 * c);
                    if constexpr (GU_LDS) { if (j == 0) *reinterpret_cast<f32x4*>(s_gu + h * (G / 2) + 4 * c) = v; }
                    else gu[c] = v;
                }
            }
#pragma unroll 1
            for (int g = 0; g < g_n; ++g) {
                asm volatile("" ::: "memory");            // keep the weight reads inside the loop (see tower.hip)
                const float* qrow = s_q + (g * 32 + j) * QS + 8 * h;
                struct X1Tile {
                    const float* qrow;
                    const f32x4* p;
                    __device__ __forceinline__ X8 raw(int s) const
                    {
                        return X8{*reinterpret_cast<const f32x4*>(qrow + 16 * s), *reinterpret_cast<const f32x4*>(qrow + 16 * s + 4)};
                    }
                    enum { RELU = 1 };
                    __device__ __forceinline__ void stage(int) const {}
                    __device__ __forceinline__ float pre(int s, const X8& q, int e) const
                    {
                        return (e < 4 ? p[2 * s][e] : p[2 * s + 1][e - 4]) + x8_get(q, e);
                    }
                };
                float part = split_mlp_part<L3>(s_w, s_tail, lane, X1Tile{qrow, p});
                if constexpr (G > 0) {
                    const float* grow = s_g + (g * 32 + j) * GS + h * (G / 2);
#pragma unroll
                    for (int c = 0; c < G / 8; ++c) {
                        const f32x4 gi = *reinterpret_cast<const f32x4*>(grow + 4 * c);
                        f32x4 gv;
                        if constexpr (GU_LDS) gv = *reinterpret_cast<const f32x4*>(s_gu + h * (G / 2) + 4 * c);
                        else gv = gu[c];
                        part = fmaf(gv.x, gi.x, part);
                        part = fmaf(gv.y, gi.y, part);
                        part = fmaf(gv.z, gi.z, part);
                        part = fmaf(gv.w, gi.w, part);
                    }
                }
                const float total = part + __shfl_xor(part, 32) + a.bp;
                const int64_t n = n0 + g * 32 + j;
                if (h == 0 && n < a.N) a.out[u * a.ld_out + n] = epilogue(total, a.act, a.lo, a.hi);
            }
        }
        if (next < tiles) {
            float* d_q = s_buf + ((it + 1) & 1) * BUF;
            tile_stage_store<G>(d_q, d_q + TI * QS, tid, st);
        }
        __syncthreads();
        tile = next;
        next = s_next[(it + 1) & 1];
    }
}

// ---------------------------------------------------------------------------------------------
// pair kernel (same staging as tower_pairs_kernel)
// ---------------------------------------------------------------------------------------------
template <int G, bool L3>
__global__ __launch_bounds__(256, 1) void tower_pairs_split_kernel(PairArgs a)
{
    constexpr int GS = G + 4;
    constexpr int WAVE_FLOATS = 32 * QS + (G > 0 ? 32 * GS : 0);
    constexpr int WU = spl_w_units(L3);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bf16x8* s_w = reinterpret_cast<const bf16x8*>(smem);
    const float* s_tail = smem + WU * 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    float* s_x = smem + WU * 4 + SPL_TAIL + wave * WAVE_FLOATS;
    float* s_gp = s_x + 32 * QS;

    for (int i = tid; i < spl_bytes(L3) / 16; i += 256)
        reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(a.img)[i];
    __syncthreads();

    const int64_t ngroups = (a.B + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t grp = (int64_t)blockIdx.x * 4 + wave;
    PairStage<G> st;
    if (grp < ngroups) pair_stage_load<G>(a, grp * 32, lane, st);
    for (; grp < ngroups; grp += stride) {
        const int64_t base = grp * 32;
        pair_stage_store<G>(s_x, s_gp, lane, st);
        __builtin_amdgcn_wave_barrier();
        if (grp + stride < ngroups) pair_stage_load<G>(a, (grp + stride) * 32, lane, st);
        const float* xrow = s_x + j * QS + 8 * h;
        struct X1Pair {
            const float* xrow;
            __device__ __forceinline__ X8 raw(int s) const
            {
                return X8{*reinterpret_cast<const f32x4*>(xrow + 16 * s), *reinterpret_cast<const f32x4*>(xrow + 16 * s + 4)};
            }
            enum { RELU = 0 };                          // the staged tile holds relu(P + Q)
            __device__ __forceinline__ void stage(int) const {}
            __device__ __forceinline__ float pre(int, const X8& q, int e) const { return x8_get(q, e); }
        };
        float part = split_mlp_part<L3>(s_w, s_tail, lane, X1Pair{xrow});
        if constexpr (G > 0) {
            const float* grow = s_gp + j * GS + h * (G / 2);
            const float* wpg = s_tail + TAIL_WPG + h * (G / 2);
#pragma unroll
            for (int c = 0; c < G / 8; ++c) {
                const f32x4 gp = *reinterpret_cast<const f32x4*>(grow + 4 * c);
                const f32x4 w = *reinterpret_cast<const f32x4*>(wpg + 4 * c);
                part = fmaf(gp.x, w.x, part);
                part = fmaf(gp.y, w.y, part);
                part = fmaf(gp.z, w.z, part);
                part = fmaf(gp.w, w.w, part);
            }
        }
        const float total = part + __shfl_xor(part, 32) + a.bp;
        const int64_t b = base + j;
        if (h == 0 && b < a.B) a.out[b] = epilogue(total, a.act, a.lo, a.hi);
        __builtin_amdgcn_wave_barrier();
    }
}

// Quarter-staged form (tables under 2^23 rows), as tower_pairs_q_kernel in tower.hip: relu(P + Q) is staged one K quarter (two
// k-steps of 16 features) at a time, LOOK register sets of one quarter each carry the rows.  The whole-row kernel above holds
// 128 VGPRs of rows and runs ONE wave per SIMD; this one runs two.
template <int G, bool L3, int WAVES, int LOOK>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void tower_pairs_split_q_kernel(PairArgs a)
{
    constexpr int GS = G + 4;
    constexpr int WAVE_FLOATS = 32 * QSP + (G > 0 ? 32 * GS : 0);
    constexpr int NG = G > 0 ? (32 * (G / 4)) / 64 : 1;
    constexpr int WU = spl_w_units(L3);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bf16x8* s_w = reinterpret_cast<const bf16x8*>(smem);
    const float* s_tail = smem + WU * 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    float* s_x = smem + WU * 4 + SPL_TAIL + wave * WAVE_FLOATS;
    float* s_gp = s_x + 32 * QSP;

    for (int i = tid; i < spl_bytes(L3) / 16; i += 64 * WAVES)
        reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(a.img)[i];
    __syncthreads();

    const int64_t ngroups = (a.B + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    int64_t grp = (int64_t)blockIdx.x * WAVES + wave;
    f32x4 p0[4], q0[4], p1[LOOK == 2 ? 4 : 1], q1[LOOK == 2 ? 4 : 1];
    f32x4 gu[NG], gi[NG];
    int nu = 0, ni = 0;
    int64_t ru = 0, ri = 0;
    if (grp < ngroups) {
        pair_ids(a, grp * 32, lane, nu, ni);
        pair_load_quarter<false>(a, nu, ni, lane, 0, p0, q0);
        if constexpr (LOOK == 2) pair_load_quarter<false>(a, nu, ni, lane, 1, p1, q1);
        if constexpr (G > 0) pair_load_gmf<G, false>(a, nu, ni, lane, gu, gi);
        if (grp + stride < ngroups) pair_ids_raw(a, (grp + stride) * 32, lane, ru, ri);
    }
    for (; grp < ngroups; grp += stride) {
        const int64_t base = grp * 32;
        const bool more = grp + stride < ngroups;
        pair_store_quarter<false>(s_x, lane, p0, q0);                      // quarter 0
        if constexpr (G > 0) pair_store_gmf<G>(s_gp, lane, gu, gi);
        __builtin_amdgcn_wave_barrier();
        pair_load_quarter<false>(a, nu, ni, lane, LOOK, p0, q0);
        int nu2 = 0, ni2 = 0;
        if (more) {
            pair_ids_clamp(a, (grp + stride) * 32, lane, ru, ri, nu2, ni2);
            if constexpr (G > 0) pair_load_gmf<G, false>(a, nu2, ni2, lane, gu, gi);
            if (grp + 2 * stride < ngroups) pair_ids_raw(a, (grp + 2 * stride) * 32, lane, ru, ri);
        }
        const float* xrow = s_x + j * QSP + 8 * h;
        struct X1Quarter {
            const PairArgs& a;
            const float* xrow;
            float* s_x;
            f32x4* p0; f32x4* q0; f32x4* p1; f32x4* q1;
            int nu, ni, nu2, ni2;
            int lane;
            bool more;
            __device__ __forceinline__ void stage(int s) const       // before raw(s); s = 2, 4, 6 open quarters 1, 2, 3
            {
                if (s & 1) return;
                if constexpr (LOOK == 2) {
                    if (s == 2) {
                        pair_store_quarter<false>(s_x, lane, p1, q1);
                        __builtin_amdgcn_wave_barrier();
                        pair_load_quarter<false>(a, nu, ni, lane, 3, p1, q1);
                    } else if (s == 4) {
                        pair_store_quarter<false>(s_x, lane, p0, q0);
                        __builtin_amdgcn_wave_barrier();
                        if (more) pair_load_quarter<false>(a, nu2, ni2, lane, 0, p0, q0);
                    } else if (s == 6) {
                        pair_store_quarter<false>(s_x, lane, p1, q1);
                        __builtin_amdgcn_wave_barrier();
                        if (more) pair_load_quarter<false>(a, nu2, ni2, lane, 1, p1, q1);
                    }
                } else {
                    pair_store_quarter<false>(s_x, lane, p0, q0);
                    __builtin_amdgcn_wave_barrier();
                    if (s < 6) pair_load_quarter<false>(a, nu, ni, lane, s / 2 + 1, p0, q0);
                    else if (more) pair_load_quarter<false>(a, nu2, ni2, lane, 0, p0, q0);
                }
            }
            __device__ __forceinline__ X8 raw(int s) const
            {
                return X8{*reinterpret_cast<const f32x4*>(xrow + 16 * (s & 1)), *reinterpret_cast<const f32x4*>(xrow + 16 * (s & 1) + 4)};
            }
            enum { RELU = 0 };                          // the staged quarter holds relu(P + Q)
            __device__ __forceinline__ float pre(int, const X8& q, int e) const { return x8_get(q, e); }
        };
        float part = split_mlp_part<L3>(s_w, s_tail, lane, X1Quarter{a, xrow, s_x, p0, q0, p1, q1, nu, ni, nu2, ni2, lane, more});
        if constexpr (G > 0) {
            const float* grow = s_gp + j * GS + h * (G / 2);
            const float* wpg = s_tail + TAIL_WPG + h * (G / 2);
#pragma unroll
            for (int c = 0; c < G / 8; ++c) {
                const f32x4 gp = *reinterpret_cast<const f32x4*>(grow + 4 * c);
                const f32x4 w = *reinterpret_cast<const f32x4*>(wpg + 4 * c);
                part = fmaf(gp.x, w.x, part);
                part = fmaf(gp.y, w.y, part);
                part = fmaf(gp.z, w.z, part);
                part = fmaf(gp.w, w.w, part);
            }
        }
        const float total = part + __shfl_xor(part, 32) + a.bp;
        const int64_t b = base + j;
        if (h == 0 && b < a.B) a.out[b] = epilogue(total, a.act, a.lo, a.hi);
        __builtin_amdgcn_wave_barrier();
        nu = nu2; ni = ni2;
    }
}

template <int G, bool L3>
int launch_tile_split_t(const TileArgs& a, hipStream_t s)
{
    constexpr size_t lds = (size_t)spl_bytes(L3) + (size_t)(2 * (TI * QS + (G > 0 ? TI * (G + 4) : 0)) + (G > 32 ? 8 * G : 0)) * sizeof(float);
    static_assert(lds <= 160 * 1024, "split tile kernel LDS budget");
    static bool attr[RECO_MAX_DEVICES] = {};
    if (int rc = reco_allow_lds(reinterpret_cast<const void*>(&tower_tile_split_kernel<G, L3>), lds, attr)) return rc;
    static int n_cu = 0;                                  // one GPU model per node: the CU count is the same on every device
    if (n_cu == 0) {
        int dev = 0;
        RECO_CHECK_HIP(hipGetDevice(&dev));
        RECO_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        if (n_cu <= 0) n_cu = 256;
    }
    const int64_t tiles = ceil_div64(a.N, TI) * ceil_div64(a.U, TUB);
    const unsigned grid = tile_grid(tiles, n_cu, a.reserve_cus);
    hipLaunchKernelGGL((tower_tile_split_kernel<G, L3>), dim3(grid), dim3(TILE_THREADS), lds, s, a);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

template <int G, bool L3>
int launch_pairs_split_t(const PairArgs& a, hipStream_t s)
{
    // Quarter-staged form for tables under 2^23 rows and GMF width <= 32 (same-box A/B, 11.5 M protocol pairs: whole rows, one
    // wave per SIMD 1.666 ms; quarters, two in flight, two waves per SIMD 1.28 ms; one in flight 1.26 ms <- this).  The Hybrid
    // shape needs 256 VGPRs + spill in this form and keeps the whole-row kernel.
    if constexpr (G <= 32) {
        if (a.n_users < (1ll << 23) && a.n_items < (1ll << 23)) {
            constexpr int WAVES = 8, LOOK = 1;
            constexpr size_t ldsq = (size_t)spl_bytes(L3) + (size_t)(WAVES * (32 * QSP + (G > 0 ? 32 * (G + 4) : 0))) * sizeof(float);
            static_assert(ldsq <= 160 * 1024, "split pair kernel LDS budget");
            static bool attrq[RECO_MAX_DEVICES] = {};
            if (int rc = reco_allow_lds(reinterpret_cast<const void*>(&tower_pairs_split_q_kernel<G, L3, WAVES, LOOK>), ldsq, attrq)) return rc;
            const int64_t nch = ceil_div64(a.B, 32 * WAVES);
            const unsigned gridq = (unsigned)(nch < 256 ? nch : 256);
            hipLaunchKernelGGL((tower_pairs_split_q_kernel<G, L3, WAVES, LOOK>), dim3(gridq), dim3(64 * WAVES), ldsq, s, a);
            RECO_CHECK_LAUNCH();
            return RECO_OK;
        }
    }
    constexpr size_t lds = (size_t)spl_bytes(L3) + (size_t)(4 * (32 * QS + (G > 0 ? 32 * (G + 4) : 0))) * sizeof(float);
    static_assert(lds <= 160 * 1024, "split pair kernel LDS budget");
    static bool attr[RECO_MAX_DEVICES] = {};
    if (int rc = reco_allow_lds(reinterpret_cast<const void*>(&tower_pairs_split_kernel<G, L3>), lds, attr)) return rc;
    const int64_t nchunks = ceil_div64(a.B, 128);
    const unsigned grid = (unsigned)(nchunks < 256 ? nchunks : 256);
    hipLaunchKernelGGL((tower_pairs_split_kernel<G, L3>), dim3(grid), dim3(256), lds, s, a);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host-side entry points used by api.hip
// ---------------------------------------------------------------------------------------------
size_t tower_split_image_bytes(int has_l3) { return (size_t)spl_bytes(has_l3 != 0); }

int tower_pack_split_image(const float* W2, const float* W3, int has_l3, const float* img, void* out, hipStream_t s)
{
    PackSplitArgs p{W2, has_l3 ? W3 : nullptr, img, has_l3};
    const int threads = (has_l3 ? (2 * 8 + 4) : 2 * 8) * 64 * 8;
    hipLaunchKernelGGL(pack_split_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, p, static_cast<unsigned char*>(out));
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

int tower_launch_tile_split(int G, int has_l3, const void* img, const float* P, const float* Q,
                            const float* EUg, const float* EIg, const int64_t* users, const int64_t* cands,
                            int64_t user_base, int64_t n_users, int64_t n_items, int64_t U, int64_t N,
                            float* out, int64_t ld_out, float bp, int act, float lo, float hi, int32_t* oob,
                            unsigned int* next_tile, int reserve_cus, hipStream_t s)
{
    if (U <= 0 || N <= 0) return RECO_OK;
    TileArgs a{static_cast<const float*>(img), P, Q, EUg, EIg, users, cands, user_base, n_users, n_items, U, N, out, ld_out,
               bp, act, lo, hi, oob, next_tile, nullptr, reserve_cus};
    if (G == 32 && has_l3) return launch_tile_split_t<32, true>(a, s);
    if (G == 0 && has_l3) return launch_tile_split_t<0, true>(a, s);
    if (G == 64 && !has_l3) return launch_tile_split_t<64, false>(a, s);
    reco_set_error("split tile kernel: unsupported configuration G=%d has_l3=%d", G, has_l3);
    return RECO_EINVAL;
}

int tower_launch_pairs_split(int G, int has_l3, const void* img, const float* P, const float* Q,
                             const float* EUg, const float* EIg, const int64_t* user, const int64_t* item,
                             int64_t n_users, int64_t n_items, int64_t B, float* out, float bp, int act,
                             float lo, float hi, int32_t* oob, hipStream_t s)
{
    if (B <= 0) return RECO_OK;
    PairArgs a{static_cast<const float*>(img), P, Q, EUg, EIg, user, item, n_users, n_items, B, out, bp, act, lo, hi, oob};
    if (G == 32 && has_l3) return launch_pairs_split_t<32, true>(a, s);
    if (G == 0 && has_l3) return launch_pairs_split_t<0, true>(a, s);
    if (G == 64 && !has_l3) return launch_pairs_split_t<64, false>(a, s);
    reco_set_error("split pair kernel: unsupported configuration G=%d has_l3=%d", G, has_l3);
    return RECO_EINVAL;
}
