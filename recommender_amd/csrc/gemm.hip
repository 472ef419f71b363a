// K-c: Linear layer  C = act(A W^T + bias)  as an fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces nn.Linear + {Sigmoid, ReLU, clamp} in
//   AutoRec.forward                      src/autorec/utils/model.py:25-33  (+ clamp inference_engine.py:232)
//   Hybrid AutoEncoder encode / decode   src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:78-125
// and computes the layer-1 projections P / Q of the scoring tower (tower.hip).
//
// Both operands are K-contiguous ([M,K] activations, [N,K] nn.Linear weights), so both are staged
// the same way: 128-B row segments -> LDS tile [rows][32 + 4 pad], read back as one ds_read_b128
// per lane per 4 MFMA steps (k = 8 q + 4 h + jj, the same free k-permutation as tower.hip).
// Exact fp32: each output is a k-ordered fmaf chain (guide: FP32-input MFMA numerics).
//
// SPLIT = true (RECO_LINEAR_SPLIT_BF16X3) is the same kernel on v_mfma_f32_32x32x16_bf16: while a tile is staged, every
// fp32 element of A and W is split exactly into three bf16 pieces (hi + mid + lo, see tower_split.hip) that go to three
// LDS images [rows][32 bf16 + pad]; per 16-wide k-step the six cross terms down to 2^-24 relative are accumulated in
// fp32.  Loads, tiling, split-K, transposed A and the epilogue are shared with the exact form.
#include "reco_common.h"

#include <cstdlib>
#include <cstring>

namespace {

constexpr int BK = 32;
constexpr int LS = BK + 4;   // padded LDS row stride (floats): 16 rows x b128 hit 16 distinct 4-bank slots
constexpr int LSB = 40;      // split form: row stride in bf16 (80 B): 16 rows x b128 hit 16 distinct 4-bank slots (5 i mod 16)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b)       // one v_cvt_pk_bf16_f32 (RNE)
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// two fp32 values -> (hi, mid, lo) packed bf16 pairs; the residuals are exact
__device__ __forceinline__ void split_pair(float a, float b, unsigned& H, unsigned& M, unsigned& L)
{
    H = pk_bf16(a, b);
    const float ra = a - __uint_as_float(H << 16), rb = b - __uint_as_float(H & 0xffff0000u);
    M = pk_bf16(ra, rb);
    L = pk_bf16(ra - __uint_as_float(M << 16), rb - __uint_as_float(M & 0xffff0000u));
}

struct LinArgs {
    const float* A;
    int64_t lda;
    const int64_t* a_rows;
    int64_t a_row_limit;
    int64_t M;
    const float* W;
    int64_t ldw;
    const float* bias;
    int64_t N, K;
    int act;
    float lo, hi;
    float* C;
    int64_t ldc;
    int32_t* oob;
    int splits;            // split-K: block b works on K slice b / tiles of k_per_split (multiple of BK) columns
    int64_t k_per_split;
    float* partial;        // [splits][M][N] raw accumulators when splits > 1
    int a_trans;           // A is stored [K, M] (row stride lda): C = act(A^T W^T + bias)
    int bias_rows;         // bias indexed by output row m instead of column n
    int m_fast;              // tile order: 1 = walk the tile grid along m first (linear_kernel)
    const float* out_scale;  // nullable device word: every output is multiplied by it after bias + activation (the tower's layer-1
                             // projections are written pre-scaled by a power of two, tower_shared.h layer1_scale)
};

__device__ __forceinline__ float apply_act(float v, int act, float lo, float hi)
{
    switch (act) {
        case RECO_ACT_RELU: return fmaxf(v, 0.f);
        case RECO_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        case RECO_ACT_CLAMP: return fminf(fmaxf(v, lo), hi);
        case RECO_ACT_RATING: return (1.f / (1.f + __expf(-v))) * (hi - lo) + lo;
        default: return v;
    }
}

// transposed-A staging: chunk position XOR of a tile row (store_tile / fragment reads of linear_kernel)
__device__ __forceinline__ int at_swz(int row) { return (row >> 4) & 3; }

// 4 consecutive k of one row.  VEC = 4 / 2 / 1: widest load the row alignment allows (K = 3706 rows are only 8-B aligned).
template <int VEC, bool GUARD>
__device__ __forceinline__ f32x4 load_k4(const float* row, int64_t k, int64_t K)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!GUARD || k + 3 < K) {
        if constexpr (VEC == 4) {
            v = *reinterpret_cast<const f32x4*>(row + k);
        } else if constexpr (VEC == 2) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 lo = *reinterpret_cast<const f32x2*>(row + k), hi = *reinterpret_cast<const f32x2*>(row + k + 2);
            v = f32x4{lo.x, lo.y, hi.x, hi.y};
        } else {
            v = f32x4{row[k], row[k + 1], row[k + 2], row[k + 3]};
        }
    } else {
        if (k + 0 < K) v.x = row[k + 0];
        if (k + 1 < K) v.y = row[k + 1];
        if (k + 2 < K) v.z = row[k + 2];
    }
    return v;
}

// block = 256 threads = 4 waves as 2 (M) x 2 (N); wave tile = (BM/2) x (BN/2) = TM x TN MFMA tiles of 32x32.
// K is consumed in BK = 32 tiles through a double-buffered LDS image: the global loads of tile t+1 are issued before
// the MFMAs of tile t and written to the other buffer after them -- one barrier per tile, no branch in the steady state
// (rows past M / N are clamped to the last row and discarded at the store; only the ragged last K tile is guarded).
// DMAT: 16-byte aligned rows of the exact form are staged with global_load_lds_dwordx4 -- straight into LDS, no staging registers
// (32 VGPRs: the single-buffered form then fits 128 and four workgroups share a CU instead of three), no ds_write, and in the
// double-buffered form the next tile is in flight under the MFMAs without occupying a register.  A lane's 16 bytes land at the
// wave's LDS window + 16 lane (tools/lds_dma_probe.hip), so rows cannot be padded; instead the 16-byte chunks of a row are
// XOR-swizzled with bits 1..3 of the row number: the lane that owns LDS position p of row R fetches global chunk
// p ^ ((R >> 1) & 7), and a fragment read of chunk c goes to position c ^ ((R >> 1) & 7) -- 16 rows x b128 then hit 16 distinct
// 4-bank slots, as with the padded layout.  The ragged last K tile still goes through registers (zero fill), into the same layout.
// Same-box A/B (tools/gemm_variant_ab.sh): AutoRec decoder 0.243 -> 0.234 ms, split-K encoder 0.237 -> 0.232, 4096^3 1.12 -> 1.095;
// single-buffered launches with fewer than 12 K tiles lose (a workgroup has nothing in flight while it waits: the Hybrid
// encoders' K = 256 / 128 layers +1.5 %) and keep the register path.
#ifndef RECO_GEMM_ONE_OCC
#define RECO_GEMM_ONE_OCC 4        // workgroups per CU the single-buffered LDS-direct form is compiled for (same-box A/B, AutoRec decoder:
                                   // 3: 0.2211-0.2214 ms   4: 0.2181-0.2224   5: 0.2173-0.2184 -- inside the run-to-run spread)
#endif
#ifndef RECO_GEMM_AT_OCC
#define RECO_GEMM_AT_OCC 4         // the 64 x 64 transposed-A form is compiled for four workgroups per CU (82 VGPRs, no AGPR copy of the
                                   // accumulators): Hybrid dense item encoder 0.1365 -> 0.1316 ms same-box.  The 64 x 128 form would spill
                                   // 52 bytes per lane at that budget and gains nothing (AutoRec dense encoder 0.2354 -> 0.2353 ms); three K
                                   // slices instead of four (696 workgroups, all resident at three per CU) lose: 0.2354 -> 0.2417 ms.
#endif
#ifdef RECO_DIAG_GEMM_TRACE         // diagnostic builds only: phase time stamps of every workgroup's thread 0 (tools/gemm_trace.py)
static __device__ unsigned long long g_gemm_trace[8 * 8192];
#define GEMM_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_gemm_trace[8 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define GEMM_STAMP(i) do { } while (0)
#endif

template <int BM, int BN, int VEC, bool AT, bool SPLIT, bool ONE, bool DMAT>
__global__ __launch_bounds__(256, DMAT ? (ONE ? RECO_GEMM_ONE_OCC : 2) : ((AT && !SPLIT && BM * BN <= 4096) ? RECO_GEMM_AT_OCC : 1)) void linear_kernel(LinArgs a)
{
    GEMM_STAMP(0);
    static_assert(!DMAT || (!AT && !SPLIT && VEC == 4), "LDS-direct staging: plain exact form with 16-byte aligned rows only");
    constexpr bool DMA = DMAT;
    constexpr int LSD = DMA ? BK : LS;          // LDS row stride of the exact form
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RA = BM / 32, RB = BN / 32;   // float4 per thread per operand per K tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: [buf][A tile BM*LS | B tile BN*LS] floats;  split form: [buf][piece(3)][A tile BM*LSB | B tile BN*LSB] bf16
    constexpr int BUF = SPLIT ? 3 * (BM + BN) * LSB / 2 : (BM + BN) * LSD;     // in floats
    constexpr int PIECE = (BM + BN) * LSB;                                      // bf16 per piece image
    // split form and transposed A: ONE LDS image, so more workgroups share a CU and one's staging (global loads, operand
    // splitting / transposing, LDS writes) overlaps the others' MFMAs (measured: transposed encoder 0.30 -> 0.26 ms, split
    // 4096^3 0.93 -> 0.76 ms); the plain exact form double-buffers inside the workgroup (4096^3: 1.09 vs 1.21 ms single)
    // ONE: the plain exact form with a short K (<= 768 per slice, e.g. the AutoRec decoder's 500): a tile is then mostly
    // prologue and epilogue, and three (with LDS-direct staging: four) single-buffered workgroups per CU overlap those better than two double-buffered ones
    // (decoder 0.250 -> 0.243 ms, the Hybrid encoders' six launches 0.357 -> 0.308 ms; 864-wide slices of the AutoRec encoder's
    // split K were 1.6 % slower this way, and at K = 4096 the double buffer wins 1.09 vs 1.21 ms)
    constexpr int NBUF = (SPLIT || AT || ONE) ? 1 : 2;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;

    const int64_t tiles_n = (a.N + BN - 1) / BN;
    const int64_t tiles_m = (a.M + BM - 1) / BM;
    const int64_t tiles = tiles_n * tiles_m;
    const int split = (int)(blockIdx.x / tiles);
    // Which output tile this workgroup computes.  Consecutive workgroups go to the 8 XCDs round-robin (speed only, never
    // correctness) and every XCD has its own 4-MB L2.  Walking the tile grid along a dimension whose tile count is a multiple of
    // 8 therefore gives XCD x every eighth operand slab of that dimension -- the same few slabs for the whole launch, L2-resident
    // -- while the other operand's slab is shared by all eight at the same time.  The launcher picks that dimension as the fast
    // one (m_fast).  AutoRec decoder, 48 x 29 tiles of 128 x 128 (operands swapped so that R_hat lands in [users, items]):
    // n fastest 0.244 ms, m fastest 0.231; the unswapped 29 x 48 form: n fastest 0.230, m fastest 0.243.  Contiguous tile ranges
    // per XCD and Triton-style grouped orders (groups of 2-8 tile rows / columns, with and without the per-XCD ranges) were
    // measured too: each helps one orientation by the same 5 % and costs the other 13-19 %.
    int64_t tile = blockIdx.x % tiles;
    if (a.m_fast) tile = (tile % tiles_m) * tiles_n + tile / tiles_m;
    const int64_t m0 = (tile / tiles_n) * BM;
    const int64_t n0 = (tile % tiles_n) * BN;
    const int64_t kb = split * a.k_per_split;                                   // this block's K slice [kb, ke)
    const int64_t ke = kb + a.k_per_split < a.K ? kb + a.k_per_split : a.K;

    // staging assignment: thread -> (row r = tid/8 + 32*s, float4 column c = tid%8)
    const int sr = tid >> 3, sc_pos = tid & 7;
    const int sc = DMA ? (sc_pos ^ ((sr >> 1) & 7)) : sc_pos;      // the GLOBAL chunk this thread fetches (DMA: swizzled)
    const int sw_f = DMA ? ((i >> 1) & 7) : 0;                     // swizzle key of this lane's fragment rows
    const float* arow[RA];
    const float* wrow[RB];
#pragma unroll
    for (int s = 0; s < RA; ++s) {
        if constexpr (AT) { arow[s] = nullptr; continue; }
        int64_t m = m0 + sr + 32 * s;
        if (m >= a.M) m = a.M - 1;
        int64_t r = m;
        if (a.a_rows) {
            r = a.a_rows[m];
            if (r < 0 || r >= a.a_row_limit) {
                if (a.oob && sc == 0 && n0 == 0 && m0 + sr + 32 * s < a.M) atomicAdd(a.oob, 1);
                r = r < 0 ? 0 : a.a_row_limit - 1;
            }
        }
        arow[s] = a.A + r * a.lda + kb + 4 * sc;
    }
#pragma unroll
    for (int s = 0; s < RB; ++s) {
        int64_t n = n0 + sr + 32 * s;
        if (n >= a.N) n = a.N - 1;
        wrow[s] = a.W + n * a.ldw + kb + 4 * sc;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int64_t Kl = ke - kb;                                                 // local K (pointers are already offset by kb)
    const int64_t nt = (Kl + BK - 1) / BK;
    const bool ragged = (Kl % BK) != 0;
    f32x4 ra[RA], rb[RB];
    // transposed A ([K, M] storage): a k-row of the tile is BM contiguous floats.  Thread -> (m block of 4 = tc, RA consecutive
    // k rows starting at RA * tkb): RA coalesced float4 loads along m, then the 4 x RA block is written k-major per m row
    // (one ds_write_b128 / b64 per row) -- the same [m][k] image as the plain path, transposed in registers.
    constexpr int CPR = BM / 4;
    const int tc = tid % CPR, tkb = tid / CPR;
    const bool at_vec = AT && (a.lda % 4 == 0) && ((uintptr_t)a.A % 16 == 0) && (m0 + BM <= a.M);
    const bool at_vec2 = AT && (a.lda % 2 == 0) && ((uintptr_t)a.A % 8 == 0) && (m0 % 2 == 0) && (m0 + BM <= a.M);   // lda = 3706

    auto load_a_trans = [&](int64_t k0, bool guard_k) {
#pragma unroll
        for (int s = 0; s < RA; ++s) {
            const int64_t k = k0 + RA * tkb + s;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (!guard_k || k < Kl) {
                const float* src = a.A + (kb + k) * a.lda + m0 + 4 * tc;
                if (at_vec) {
                    v = *reinterpret_cast<const f32x4*>(src);
                } else if (at_vec2) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 p0 = *reinterpret_cast<const f32x2*>(src), p1 = *reinterpret_cast<const f32x2*>(src + 2);
                    v = f32x4{p0.x, p0.y, p1.x, p1.y};
                } else {
                    const int64_t left = a.M - (m0 + 4 * tc);
                    if (left > 0) v.x = src[0];
                    if (left > 1) v.y = src[1];
                    if (left > 2) v.z = src[2];
                    if (left > 3) v.w = src[3];
                }
            }
            ra[s] = v;
        }
    };
    auto load_tile = [&](int64_t t) {
        const int64_t k0 = t * BK;
        if (ragged && t == nt - 1) {
            if constexpr (AT) load_a_trans(k0, true);
            else {
#pragma unroll
                for (int s = 0; s < RA; ++s) ra[s] = load_k4<VEC, true>(arow[s], k0, Kl - 4 * sc);
            }
#pragma unroll
            for (int s = 0; s < RB; ++s) rb[s] = load_k4<VEC, true>(wrow[s], k0, Kl - 4 * sc);
        } else {
            if constexpr (AT) load_a_trans(k0, false);
            else {
#pragma unroll
                for (int s = 0; s < RA; ++s) ra[s] = load_k4<VEC, false>(arow[s], k0, Kl);
            }
#pragma unroll
            for (int s = 0; s < RB; ++s) rb[s] = load_k4<VEC, false>(wrow[s], k0, Kl);
        }
    };
    auto store_tile = [&](int buf) {
        if constexpr (SPLIT) {
            unsigned short* base = reinterpret_cast<unsigned short*>(smem + buf * BUF);
            // a thread's RA (RB) register groups hold 4 (AT with BM = 64: 2) consecutive k of one row each
            auto put4 = [&](unsigned short* row, f32x4 v) {
                unsigned h0, m0, l0, h1, m1, l1;
                split_pair(v.x, v.y, h0, m0, l0);
                split_pair(v.z, v.w, h1, m1, l1);
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2*>(row) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(row + PIECE) = u32x2{m0, m1};
                *reinterpret_cast<u32x2*>(row + 2 * PIECE) = u32x2{l0, l1};
            };
            if constexpr (AT) {
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    unsigned short* row = base + (4 * tc + i4) * LSB + RA * tkb;
                    if constexpr (RA == 4) {
                        put4(row, f32x4{ra[0][i4], ra[1][i4], ra[2][i4], ra[3][i4]});
                    } else {
                        unsigned h, m, l;
                        split_pair(ra[0][i4], ra[1][i4], h, m, l);
                        *reinterpret_cast<unsigned*>(row) = h;
                        *reinterpret_cast<unsigned*>(row + PIECE) = m;
                        *reinterpret_cast<unsigned*>(row + 2 * PIECE) = l;
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < RA; ++s) put4(base + (sr + 32 * s) * LSB + 4 * sc, ra[s]);
            }
#pragma unroll
            for (int s = 0; s < RB; ++s) put4(base + BM * LSB + (sr + 32 * s) * LSB + 4 * sc, rb[s]);
            return;
        }
        float* s_a = smem + buf * BUF;
        float* s_b = s_a + BM * LSD;
        if constexpr (AT) {   // [m][k] image from k-major registers: row 4 tc + i gets element i of each of the RA loads
            // Consecutive lanes write rows FOUR apart at the same k chunk: with the padded stride alone 16 lanes land in 4 distinct
            // 16-byte slots (36 x 4 tc mod 64 banks = 16 tc mod 64), a 4-way conflict on every write of the transposition.  The
            // chunk position is therefore XORed with bits 4..5 of the row (at_swz): writes and fragment reads are then both
            // conflict-free (brute-forced over stride x shift x mask, tools/lds_swizzle_search.py).
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int row = 4 * tc + i4;
                if constexpr (RA == 4) {
                    float* d = s_a + row * LS + 4 * (tkb ^ at_swz(row));
                    *reinterpret_cast<f32x4*>(d) = f32x4{ra[0][i4], ra[1][i4], ra[2][i4], ra[3][i4]};
                } else {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    float* d = s_a + row * LS + 4 * ((tkb >> 1) ^ at_swz(row)) + 2 * (tkb & 1);
                    *reinterpret_cast<f32x2*>(d) = f32x2{ra[0][i4], ra[1][i4]};
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < RA; ++s) *reinterpret_cast<f32x4*>(s_a + (sr + 32 * s) * LSD + 4 * sc_pos) = ra[s];
        }
#pragma unroll
        for (int s = 0; s < RB; ++s) *reinterpret_cast<f32x4*>(s_b + (sr + 32 * s) * LSD + 4 * sc_pos) = rb[s];
    };
    // DMA: tile t into the LDS image; a full tile goes straight from global memory, the ragged last one through registers
    auto stage_dma = [&](int64_t t, int buf, bool wait) {
        if (ragged && t == nt - 1) {
            load_tile(t);
            store_tile(buf);
        } else {
            const int64_t k0 = t * BK;
            typedef const __attribute__((address_space(1))) void* gptr_t;
            typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
            for (int s = 0; s < RA; ++s)
                __builtin_amdgcn_global_load_lds((gptr_t)(arow[s] + k0), (lptr_t)(smem + buf * BUF + (32 * s + 8 * wave) * BK), 16, 0, 0);
#pragma unroll
            for (int s = 0; s < RB; ++s)
                __builtin_amdgcn_global_load_lds((gptr_t)(wrow[s] + k0), (lptr_t)(smem + buf * BUF + BM * BK + (32 * s + 8 * wave) * BK), 16, 0, 0);
        }
        if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    if constexpr (DMA) {
        stage_dma(0, 0, true);
    } else {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    GEMM_STAMP(1);

    for (int64_t t = 0; t < nt; ++t) {
#ifdef RECO_ABL_GEMM_NO_LOAD       // timing-only ablation: every K tile re-stages the first one's registers
        if (t + 1 < nt && a.K < 0) load_tile(t + 1);
#else
        if constexpr (!DMA) { if (t + 1 < nt) load_tile(t + 1); }
        else if constexpr (NBUF == 2) { if (t + 1 < nt) stage_dma(t + 1, (int)((t + 1) & 1), false); }   // in flight under the MFMAs
#endif
        if constexpr (SPLIT) {
            const unsigned short* pa = reinterpret_cast<const unsigned short*>(smem + (t & (NBUF - 1)) * BUF) + (wm * (BM / 2) + i) * LSB + 8 * h;
            const unsigned short* pb = reinterpret_cast<const unsigned short*>(smem + (t & (NBUF - 1)) * BUF) + BM * LSB + (wn * (BN / 2) + i) * LSB + 8 * h;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) fa[tm][c] = *reinterpret_cast<const bf16x8*>(pa + c * PIECE + tm * 32 * LSB + 16 * ks);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) fb[tn][c] = *reinterpret_cast<const bf16x8*>(pb + c * PIECE + tn * 32 * LSB + 16 * ks);
                }
                // six cross terms, smallest first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
                constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][ta[term]], fb[tn][tb[term]], acc[tm][tn], 0, 0, 0);
            }
        } else {
        const float* s_a = smem + (t & (NBUF - 1)) * BUF + (wm * (BM / 2) + i) * LSD + ((DMA || AT) ? 0 : 4 * h);
        const float* s_b = smem + (t & (NBUF - 1)) * BUF + BM * LSD + (wn * (BN / 2) + i) * LSD + (DMA ? 0 : 4 * h);
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 fa[TM], fb[TN];
            const int fo = DMA ? 4 * ((2 * q + h) ^ sw_f) : 8 * q;      // chunk 2 q + h of the row (DMA: at its swizzled position)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                if constexpr (AT) fa[tm] = *reinterpret_cast<const f32x4*>(s_a + tm * 32 * LSD + 4 * ((2 * q + h) ^ at_swz(wm * (BM / 2) + tm * 32 + i)));
                else fa[tm] = *reinterpret_cast<const f32x4*>(s_a + tm * 32 * LSD + fo);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) fb[tn] = *reinterpret_cast<const f32x4*>(s_b + tn * 32 * LSD + fo);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[tm][jj], fb[tn][jj], acc[tm][tn], 0, 0, 0);
        }
        }
        if constexpr (DMA && NBUF == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (DMA) {
            __syncthreads();                       // every wave is done reading the image
            if (t + 1 < nt) stage_dma(t + 1, 0, true);
        } else if constexpr (NBUF == 1) {
            __syncthreads();                       // every wave is done reading the image
            if (t + 1 < nt) store_tile(0);
        } else {
            if (t + 1 < nt) store_tile((t + 1) & 1);
        }
        __syncthreads();
    }

    GEMM_STAMP(2);
    // epilogue: D[row = (r&3) + 8 (r>>2) + 4 h][col = lane & 31]
    const float osc = a.out_scale ? a.out_scale[0] : 1.f;
    const bool brows = a.bias && a.bias_rows, bcols = a.bias && !a.bias_rows;
    float bvn[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int64_t n = n0 + wn * (BN / 2) + tn * 32 + i;
        bvn[tn] = (bcols && n < a.N) ? a.bias[n] : 0.f;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        // per-row bias (the AutoRec decoder run with swapped operands): this lane's rows four at a time, not once per element
        // (loaded inside the element loop it cost the swapped decoder 13 %: 0.265 against 0.235 ms)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float brow[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int64_t m = m0 + wm * (BM / 2) + tm * 32 + jj + 8 * rq + 4 * h;
                brow[jj] = (brows && m < a.M) ? a.bias[m] : 0.f;
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int64_t n = n0 + wn * (BN / 2) + tn * 32 + i;
                if (n >= a.N) continue;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int r = 4 * rq + jj;
                    const int64_t m = m0 + wm * (BM / 2) + tm * 32 + jj + 8 * rq + 4 * h;
                    if (m >= a.M) continue;
#ifdef RECO_ABL_GEMM_NO_STORE      // timing-only ablation: the stores stay in the code but never execute
                    if (acc[tm][tn][r] != 12345.678f) continue;
#endif
                    if (a.splits > 1) a.partial[((int64_t)split * a.M + m) * a.N + n] = acc[tm][tn][r];
                    else a.C[m * a.ldc + n] = apply_act(acc[tm][tn][r] + (brows ? brow[jj] : bvn[tn]), a.act, a.lo, a.hi) * osc;
                }
            }
        }
    }
    GEMM_STAMP(3);
#ifdef RECO_DIAG_GEMM_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 8192)
        g_gemm_trace[8 * blockIdx.x + 4] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
#endif
}

#ifdef RECO_DIAG_GEMM_TRACE
extern "C" int reco_debug_gemm_trace(unsigned long long* host_out, int words)
{
    if (words > 8 * 8192) words = 8 * 8192;
    RECO_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_trace), (size_t)words * sizeof(unsigned long long)));
    return RECO_OK;
}
#endif

// split-K epilogue: slices are added in slice order (deterministic), then bias + activation
__global__ __launch_bounds__(256) void splitk_reduce_kernel(LinArgs a)
{
    const int64_t total = a.M * a.N;
    const float osc = a.out_scale ? a.out_scale[0] : 1.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        float v = a.partial[e];
        for (int sp = 1; sp < a.splits; ++sp) v += a.partial[(int64_t)sp * total + e];
        const int64_t m = e / a.N, n = e - m * a.N;
        a.C[m * a.ldc + n] = apply_act(v + (a.bias ? a.bias[a.bias_rows ? m : n] : 0.f), a.act, a.lo, a.hi) * osc;
    }
}

template <int BM, int BN, int VEC, bool AT, bool SPLIT, bool ONE, bool DMAT = false>
int launch_k(const LinArgs& a, hipStream_t s)
{
    constexpr size_t lds = SPLIT ? (size_t)3 * (BM + BN) * LSB * 2 : (size_t)((AT || ONE) ? 1 : 2) * (BM + BN) * LS * sizeof(float);
    static bool attr[RECO_MAX_DEVICES] = {};
    if (int rc = reco_allow_lds(reinterpret_cast<const void*>(&linear_kernel<BM, BN, VEC, AT, SPLIT, ONE, DMAT>), lds, attr)) return rc;
    const int64_t blocks = ceil_div64(a.M, BM) * ceil_div64(a.N, BN) * a.splits;
    RECO_REQUIRE(blocks < (1ll << 31), "linear grid too large");
    {
        ProfScope prof(s, RECO_PROF_LINEAR);
        hipLaunchKernelGGL((linear_kernel<BM, BN, VEC, AT, SPLIT, ONE, DMAT>), dim3((unsigned)blocks), dim3(256), lds, s, a);
    }
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

template <int BM, int BN, int VEC, bool AT, bool SPLIT>
int launch_v(const LinArgs& a, hipStream_t s)
{
    int rc;
    if constexpr (!AT && !SPLIT && VEC == 4) {              // 16-byte aligned rows: LDS-direct staging (see linear_kernel)
        if (a.k_per_split > 768) rc = launch_k<BM, BN, VEC, AT, SPLIT, false, true>(a, s);
        else if (a.k_per_split >= 12 * BK) rc = launch_k<BM, BN, VEC, AT, SPLIT, true, true>(a, s);
        else rc = launch_k<BM, BN, VEC, AT, SPLIT, true, false>(a, s);
    } else if constexpr (!AT && !SPLIT) {
        rc = a.k_per_split <= 768 ? launch_k<BM, BN, VEC, AT, SPLIT, true>(a, s) : launch_k<BM, BN, VEC, AT, SPLIT, false>(a, s);
    } else rc = launch_k<BM, BN, VEC, AT, SPLIT, false>(a, s);
    if (rc) return rc;
    if (a.splits > 1) {
        const int64_t rb = ceil_div64(a.M * a.N, 256);
        {
            ProfScope prof(s, RECO_PROF_LINEAR_REDUCE);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(rb < 4096 ? rb : 4096)), dim3(256), 0, s, a);
        }
        RECO_CHECK_LAUNCH();
    }
    return RECO_OK;
}

template <int BM, int BN, bool SPLIT>
int launch_t(const LinArgs& a, int vec, hipStream_t s)
{
    if (a.a_trans) {
        if (vec == 4) return launch_v<BM, BN, 4, true, SPLIT>(a, s);
        if (vec == 2) return launch_v<BM, BN, 2, true, SPLIT>(a, s);
        return launch_v<BM, BN, 1, true, SPLIT>(a, s);
    }
    if (vec == 4) return launch_v<BM, BN, 4, false, SPLIT>(a, s);
    if (vec == 2) return launch_v<BM, BN, 2, false, SPLIT>(a, s);
    return launch_v<BM, BN, 1, false, SPLIT>(a, s);
}

template <bool SPLIT>
int launch_tile(const LinArgs& a, int bm, int bn, int vec, hipStream_t s)
{
    if (bm == 128 && bn == 128) return launch_t<128, 128, SPLIT>(a, vec, s);
    if (bm == 128) return launch_t<128, 64, SPLIT>(a, vec, s);
    if (bn == 128) return launch_t<64, 128, SPLIT>(a, vec, s);
    return launch_t<64, 64, SPLIT>(a, vec, s);
}

// rough utilisation model: larger tiles amortise LDS traffic better, but the grid must cover the 256 CUs
double tile_score(int64_t M, int64_t N, int bm, int bn, double tile_eff)
{
    const double blocks = (double)(ceil_div64(M, bm) * ceil_div64(N, bn));
    const double rounds = (double)ceil_div64((int64_t)blocks, 256);
    const double useful = ((double)M * (double)N) / (blocks * bm * bn);     // padding waste of ragged edges
    return tile_eff * useful * blocks / (rounds * 256.0);
}

}  // namespace

// K slices so that a small output grid still fills the chip; 0 workspace -> no split
static int choose_splits(int64_t blocks, int64_t K)
{
    if (blocks >= 384 || K < 1024) return 1;
    int64_t sp = ceil_div64(768, blocks);
    const int64_t by_k = K / 512;
    if (sp > by_k) sp = by_k;
    if (sp > 8) sp = 8;
    return sp < 1 ? 1 : (int)sp;
}

size_t linear_workspace_bytes(int64_t M, int64_t N, int64_t K)
{
    if (M <= 0 || N <= 0) return 0;
    // worst case over the tile shapes: the smallest grid (128x128) takes the most slices
    const int sp = choose_splits(ceil_div64(M, 128) * ceil_div64(N, 128), K);
    return sp > 1 ? align_up((size_t)sp * M * N * sizeof(float), 256) : 0;
}

int launch_linear_f32(const float* A, int64_t lda, const int64_t* a_rows, int64_t a_row_limit, int64_t M,
                      const float* W, int64_t ldw, const float* bias, int64_t N, int64_t K,
                      int act, float lo, float hi, float* C, int64_t ldc, int32_t* oob,
                      void* ws, size_t ws_bytes, hipStream_t s, int flags, const float* out_scale)
{
    if (M <= 0 || N <= 0) return RECO_OK;
    const bool at = (flags & RECO_LINEAR_A_TRANSPOSED) != 0;
    RECO_REQUIRE(K > 0, "linear: K must be positive");
    RECO_REQUIRE(A && W && C, "linear: null pointer");
    RECO_REQUIRE(lda >= (at ? M : K) && ldw >= K && ldc >= N, "linear: leading dimension smaller than the row");
    RECO_REQUIRE(!(at && a_rows), "linear: row gather and transposed A cannot be combined");
    LinArgs a{A, lda, a_rows, a_rows ? a_row_limit : M, M, W, ldw, bias, N, K, act, lo, hi, C, ldc, oob, 1, K, nullptr,
              at ? 1 : 0, (flags & RECO_LINEAR_BIAS_PER_ROW) ? 1 : 0, 0, out_scale};
    int vec = 1;
    const bool a4 = at || (lda % 4 == 0 && (uintptr_t)A % 16 == 0), a2 = at || (lda % 2 == 0 && (uintptr_t)A % 8 == 0);
    if (a4 && ldw % 4 == 0 && (uintptr_t)W % 16 == 0) vec = 4;
    else if (a2 && ldw % 2 == 0 && (uintptr_t)W % 8 == 0) vec = 2;
    // transposed A: the A operand is the expensive one to stage (transposed in registers, LDS writes four rows apart), so of two
    // shapes with the same grid the one with FEWER A rows per tile wins: AutoRec's dense encoder 0.287 ms at 128 x 64, 0.260 at
    // 64 x 128 (same-box, forced shapes); the plain form is indifferent (0.219 / 0.217)
    const double s11 = tile_score(M, N, 128, 128, 1.00), s12 = tile_score(M, N, 128, 64, at ? 0.80 : 0.85);
    const double s21 = tile_score(M, N, 64, 128, 0.85), s22 = tile_score(M, N, 64, 64, 0.70);
    int bm = 64, bn = 64;
    if (s11 >= s12 && s11 >= s21 && s11 >= s22) { bm = 128; bn = 128; }
    else if (s12 >= s21 && s12 >= s22) { bm = 128; bn = 64; }
    else if (s21 >= s22) { bm = 64; bn = 128; }
    // Two corrections the grid-fill model above cannot see, from forcing every shape on the product's GEMMs (tools/gemm_shape_exp.py,
    // ms per call at 128x128 / 128x64 / 64x128 / 64x64, table in DESIGN section 3 K-c):
    //  (a) a large A against a narrow output (the Hybrid user encoder: A = the 89-MB rating matrix, N = 256): every tile column
    //      re-reads A, so the shapes with the fewest tile columns win -- 0.134 / 0.165 / 0.128 / 0.151;
    //  (b) a large output behind a short K (the decoders): a tile is mostly prologue and epilogue, and smaller tiles put more of
    //      them in flight -- K = 500: 0.2225 / 0.2109 / 0.2166 / 0.2191; K = 256: 0.134 / 0.130 / 0.130 / 0.117.
    const int64_t tiles128 = ceil_div64(M, 128) * ceil_div64(N, 128);
    if (!at && N <= 256 && M * K * 4 >= (32ll << 20) && bn == 64) {
        bn = 128;
        bm = s11 >= s21 ? 128 : 64;
    } else if (!at && tiles128 > 1024 && K <= 768) {
        if (K <= 320) { bm = 64; bn = 64; }
        else { bm = 128; bn = 64; }
    }
#ifdef RECO_DIAG_GEMM_TILE         // diagnostic builds only: force the tile shape
    if (const char* e = getenv("RECO_GEMM_TILE")) { bm = atoi(e); const char* x = strchr(e, 'x'); bn = x ? atoi(x + 1) : bm; }
#endif
    if (ws != nullptr) {
        int sp = choose_splits(ceil_div64(M, bm) * ceil_div64(N, bn), K);
#ifdef RECO_DIAG_GEMM_TILE         // diagnostic builds only: force the number of K slices
        if (const char* e = getenv("RECO_GEMM_SPLITS")) sp = atoi(e);
#endif
        while (sp > 1 && (size_t)sp * M * N * sizeof(float) > ws_bytes) --sp;
        if (sp > 1) {
            a.splits = sp;
            a.k_per_split = ceil_div64(ceil_div64(K, sp), BK) * BK;
            a.splits = (int)ceil_div64(K, a.k_per_split);            // drop empty trailing slices
            a.partial = static_cast<float*>(ws);
            if (a.splits <= 1) { a.splits = 1; a.k_per_split = K; a.partial = nullptr; }
        }
    }
    {   // fast dimension of the tile order: the one whose tile count is a multiple of 8 (linear_kernel), else the longer one
        const int64_t tm = ceil_div64(M, bm), tn = ceil_div64(N, bn);
        const bool gm = tm % 8 == 0, gn = tn % 8 == 0;
        a.m_fast = (gm != gn) ? (gm ? 1 : 0) : (tm > tn ? 1 : 0);
    }
    if (flags & RECO_LINEAR_SPLIT_BF16X3) return launch_tile<true>(a, bm, bn, vec, s);
    return launch_tile<false>(a, bm, bn, vec, s);
}
