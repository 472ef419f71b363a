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
#include "reco_common.h"

namespace {

constexpr int BK = 32;
constexpr int LS = BK + 4;   // padded LDS row stride (floats): 16 rows x b128 hit 16 distinct 4-bank slots

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

template <bool VEC4>
__device__ __forceinline__ f32x4 load_k4(const float* row, int64_t k, int64_t K)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row == nullptr) return v;
    if (VEC4 && k + 3 < K) return *reinterpret_cast<const f32x4*>(row + k);
    if (k + 0 < K) v.x = row[k + 0];
    if (k + 1 < K) v.y = row[k + 1];
    if (k + 2 < K) v.z = row[k + 2];
    if (k + 3 < K) v.w = row[k + 3];
    return v;
}

// block = 256 threads = 4 waves as 2 (M) x 2 (N); wave tile = (BM/2) x (BN/2) = TM x TN MFMA tiles of 32x32
template <int BM, int BN, bool VEC4>
__global__ __launch_bounds__(256) void linear_kernel(LinArgs a)
{
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RA = BM / 32, RB = BN / 32;   // float4 per thread per operand per K tile
    __shared__ __attribute__((aligned(16))) float s_a[BM * LS];
    __shared__ __attribute__((aligned(16))) float s_b[BN * LS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;

    const int64_t tiles_n = (a.N + BN - 1) / BN;
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM;
    const int64_t n0 = (int64_t)(blockIdx.x % tiles_n) * BN;

    // staging assignment: thread -> (row r = tid/8 + 32*s, float4 column c = tid%8)
    const int sr = tid >> 3, sc = tid & 7;
    const float* arow[RA];
    const float* wrow[RB];
#pragma unroll
    for (int s = 0; s < RA; ++s) {
        const int64_t m = m0 + sr + 32 * s;
        arow[s] = nullptr;
        if (m < a.M) {
            int64_t r = m;
            if (a.a_rows) {
                r = a.a_rows[m];
                if (r < 0 || r >= a.a_row_limit) {
                    if (a.oob && sc == 0 && n0 == 0) atomicAdd(a.oob, 1);
                    r = r < 0 ? 0 : a.a_row_limit - 1;
                }
            }
            arow[s] = a.A + r * a.lda;
        }
    }
#pragma unroll
    for (int s = 0; s < RB; ++s) {
        const int64_t n = n0 + sr + 32 * s;
        wrow[s] = n < a.N ? a.W + n * a.ldw : nullptr;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    f32x4 ra[RA], rb[RB];
#pragma unroll
    for (int s = 0; s < RA; ++s) ra[s] = load_k4<VEC4>(arow[s], 4 * sc, a.K);
#pragma unroll
    for (int s = 0; s < RB; ++s) rb[s] = load_k4<VEC4>(wrow[s], 4 * sc, a.K);

    for (int64_t k0 = 0; k0 < a.K; k0 += BK) {
        __syncthreads();   // everyone is done reading the previous tile
#pragma unroll
        for (int s = 0; s < RA; ++s) *reinterpret_cast<f32x4*>(s_a + (sr + 32 * s) * LS + 4 * sc) = ra[s];
#pragma unroll
        for (int s = 0; s < RB; ++s) *reinterpret_cast<f32x4*>(s_b + (sr + 32 * s) * LS + 4 * sc) = rb[s];
        __syncthreads();
        if (k0 + BK < a.K) {   // prefetch the next tile into registers while this one is consumed
#pragma unroll
            for (int s = 0; s < RA; ++s) ra[s] = load_k4<VEC4>(arow[s], k0 + BK + 4 * sc, a.K);
#pragma unroll
            for (int s = 0; s < RB; ++s) rb[s] = load_k4<VEC4>(wrow[s], k0 + BK + 4 * sc, a.K);
        }
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                fa[tm] = *reinterpret_cast<const f32x4*>(s_a + (wm * (BM / 2) + tm * 32 + i) * LS + 8 * q + 4 * h);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                fb[tn] = *reinterpret_cast<const f32x4*>(s_b + (wn * (BN / 2) + tn * 32 + i) * LS + 8 * q + 4 * h);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[tm][jj], fb[tn][jj], acc[tm][tn], 0, 0, 0);
        }
    }

    // epilogue: D[row = (r&3) + 8 (r>>2) + 4 h][col = lane & 31]
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int64_t n = n0 + wn * (BN / 2) + tn * 32 + i;
        if (n >= a.N) continue;
        const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < a.M) a.C[m * a.ldc + n] = apply_act(acc[tm][tn][r] + bv, a.act, a.lo, a.hi);
            }
        }
    }
}

template <int BM, int BN>
int launch_t(const LinArgs& a, bool vec4, hipStream_t s)
{
    const int64_t blocks = ceil_div64(a.M, BM) * ceil_div64(a.N, BN);
    RECO_REQUIRE(blocks < (1ll << 31), "linear grid too large");
    if (vec4)
        hipLaunchKernelGGL((linear_kernel<BM, BN, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((linear_kernel<BM, BN, false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

}  // namespace

int launch_linear_f32(const float* A, int64_t lda, const int64_t* a_rows, int64_t a_row_limit, int64_t M,
                      const float* W, int64_t ldw, const float* bias, int64_t N, int64_t K,
                      int act, float lo, float hi, float* C, int64_t ldc, int32_t* oob, hipStream_t s)
{
    if (M <= 0 || N <= 0) return RECO_OK;
    RECO_REQUIRE(K > 0, "linear: K must be positive");
    RECO_REQUIRE(A && W && C, "linear: null pointer");
    RECO_REQUIRE(lda >= K && ldw >= K && ldc >= N, "linear: leading dimension smaller than the row");
    LinArgs a{A, lda, a_rows, a_rows ? a_row_limit : M, M, W, ldw, bias, N, K, act, lo, hi, C, ldc, oob};
    const bool vec4 = (lda % 4 == 0) && (ldw % 4 == 0) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0);
    // fill the 256 CUs: fall back to 64x64 tiles when 128x128 would leave most of the chip idle
    const int64_t big = ceil_div64(M, 128) * ceil_div64(N, 128);
    if (big >= 200) return launch_t<128, 128>(a, vec4, s);
    return launch_t<64, 64>(a, vec4, s);
}
