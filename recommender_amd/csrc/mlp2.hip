// Two consecutive narrow Linear layers in ONE launch:  C = act3( act2( X W2^T + b2 ) W3^T + b3 )   (gfx950, fp32 MFMA)
//
// The Hybrid model's encoders and decoders are three-layer MLPs whose wide layer touches the rating matrix and whose two
// narrow layers are tiny (src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:78-106: 256 -> 128 -> 64 after the first
// encoder layer, 64 -> 128 -> 256 before the last decoder layer).  As GEMM launches of their own the narrow layers are all
// prologue and epilogue (K = 256 / 128: 11-15 us each for 0.4 GFLOP) and their [rows, 128] intermediate goes through HBM.
// Here a workgroup takes 32 rows of X: the rows are staged in LDS once, wave w computes output features [32 w, 32 w + 32) of
// the first layer with the ROW on the MFMA column, ReLU'd results go to a second LDS tile, and the second layer runs out of
// that tile -- the intermediate never leaves the CU.  Weights are read straight from global memory (L2-resident: 160 KB),
// sixteen k-quads ahead of their MFMAs.
// Numerics: every output is the k-ordered fmaf chain of gemm.hip's linear_kernel without split-K (same MFMA, same k order
// k = 8 q + 4 h + jj), so the results equal two reco_linear_f32 calls bit for bit.
#include "reco_common.h"

namespace {

struct Mlp2Args {
    const float* X;        // [M, K1], row stride ldx
    int64_t ldx, M;
    int K1, N2, N3;
    const float* W2;       // [N2, K1] nn.Linear layout
    const float* b2;
    const float* W3;       // [N3, N2]
    const float* b3;
    int act2, act3;
    float lo, hi;
    float* C;              // [M, N3], row stride ldc
    int64_t ldc;
};

__device__ __forceinline__ float mlp2_act(float v, int act, float lo, float hi)
{
    switch (act) {
        case RECO_ACT_RELU: return fmaxf(v, 0.f);
        case RECO_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        case RECO_ACT_CLAMP: return fminf(fmaxf(v, lo), hi);
        case RECO_ACT_RATING: return (1.f / (1.f + __expf(-v))) * (hi - lo) + lo;
        default: return v;
    }
}

// acc[32 features x 32 rows] += W[f0 + i, 0..K) . x[row j, 0..K): A from global memory (lane i, k-slot h: 4 consecutive k per
// quad), B from the LDS tile.  AHEAD quads of A are in flight (an L2 hit is ~0.8 us away, a quad's four MFMAs take 0.11 us);
// the first AHEAD are requested by mlp2_prefetch BEFORE the barrier that publishes the tile, so they travel under the staging.
constexpr int MLP2_AHEAD = 16;

__device__ __forceinline__ void mlp2_prefetch(const float* __restrict__ wrow, int nq, f32x4* a)
{
#pragma unroll
    for (int p = 0; p < MLP2_AHEAD; ++p) a[p] = p < nq ? *reinterpret_cast<const f32x4*>(wrow + 8 * p) : f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ f32x16 mlp2_block(const float* __restrict__ wrow, const float* xrow, int nq, f32x4* a, f32x16 acc)
{
    // Whole groups of AHEAD quads run without a branch: the group's LDS reads are issued up front and the refills of the weight
    // window sit between the MFMAs, so nothing but the accumulator chain itself orders the 4 AHEAD MFMAs (with a guard per quad
    // every quad paid a full L2 round trip: 525 cycles per 4 MFMAs on the workgroup timeline, tools/mlp2_trace.py).
    int q0 = 0;
    for (; q0 + MLP2_AHEAD <= nq; q0 += MLP2_AHEAD) {
        f32x4 x[MLP2_AHEAD];
#pragma unroll
        for (int p = 0; p < MLP2_AHEAD; ++p) x[p] = *reinterpret_cast<const f32x4*>(xrow + 8 * (q0 + p));
#pragma unroll
        for (int p = 0; p < MLP2_AHEAD; ++p) {
            const f32x4 av = a[p];
            // the refill is unconditional (past the end it re-reads the last quad): a guard here is a branch, and behind a branch
            // the compiler waits for EVERY outstanding load (s_waitcnt vmcnt(0) after each refill: the window was no window)
            int qn = q0 + p + MLP2_AHEAD;
            qn = qn < nq ? qn : nq - 1;
            a[p] = *reinterpret_cast<const f32x4*>(wrow + 8 * qn);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, x[p].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, x[p].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, x[p].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, x[p].w, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);          // keep the refill here (left alone the scheduler sinks all AHEAD of them to the loop's end)
        }
    }
#pragma unroll
    for (int p = 0; p < MLP2_AHEAD; ++p) {                              // the ragged last group (K not a multiple of 8 AHEAD)
        const int q = q0 + p;
        if (q < nq) {                                                   // uniform
            const f32x4 av = a[p];
            const f32x4 x = *reinterpret_cast<const f32x4*>(xrow + 8 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, x.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, x.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, x.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, x.w, acc, 0, 0, 0);
        }
    }
    return acc;
}

#ifdef RECO_DIAG_MLP2_TRACE         // diagnostic builds only: phase time stamps of every workgroup's wave 0 (tools/mlp2_trace.py)
static __device__ unsigned long long g_mlp2_trace[8 * 1024];
#define MLP2_STAMP(i) do { if (tid == 0 && blockIdx.x < 1024) g_mlp2_trace[8 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define MLP2_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(256) void mlp2_kernel(Mlp2Args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LX = a.K1 + 4, L2 = a.N2 + 4;                 // padded row strides: 16 rows x b128 hit distinct 4-bank slots
    float* s_x = smem;                                      // [32][LX]
    float* s_2 = smem + 32 * LX;                            // [32][L2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * 32;

    MLP2_STAMP(0);
    f32x4 aw[MLP2_AHEAD];
    if (32 * wave < a.N2) mlp2_prefetch(a.W2 + (int64_t)(32 * wave + j) * a.K1 + 4 * h, a.K1 / 8, aw);
    // stage the 32 rows (rows past M repeat the last one; their results are not stored)
    const int c4 = a.K1 / 4;
    for (int e = tid; e < 32 * c4; e += 256) {
        const int r = e / c4, c = e - r * c4;
        int64_t m = m0 + r;
        if (m >= a.M) m = a.M - 1;
        *reinterpret_cast<f32x4*>(s_x + r * LX + 4 * c) = *reinterpret_cast<const f32x4*>(a.X + m * a.ldx + 4 * c);
    }
    __syncthreads();
    MLP2_STAMP(1);

    // first layer: wave w -> features [32 w, 32 w + 32) (and + 128, ... for wider layers)
    for (int f0 = 32 * wave; f0 < a.N2; f0 += 128) {
        const float* wrow = a.W2 + (int64_t)(f0 + j) * a.K1 + 4 * h;
        if (f0 >= 128) mlp2_prefetch(wrow, a.K1 / 8, aw);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = mlp2_block(wrow, s_x + j * LX + 4 * h, a.K1 / 8, aw, acc);
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};           // bias after the chain, as linear_kernel's epilogue does
            if (a.b2) bv = *reinterpret_cast<const f32x4*>(a.b2 + f0 + 8 * rq + 4 * h);
            f32x4 v;
            v.x = mlp2_act(acc[4 * rq + 0] + bv.x, a.act2, a.lo, a.hi); v.y = mlp2_act(acc[4 * rq + 1] + bv.y, a.act2, a.lo, a.hi);
            v.z = mlp2_act(acc[4 * rq + 2] + bv.z, a.act2, a.lo, a.hi); v.w = mlp2_act(acc[4 * rq + 3] + bv.w, a.act2, a.lo, a.hi);
            *reinterpret_cast<f32x4*>(s_2 + j * L2 + f0 + 8 * rq + 4 * h) = v;
        }
    }
    MLP2_STAMP(2);
    if (32 * wave < a.N3) mlp2_prefetch(a.W3 + (int64_t)(32 * wave + j) * a.N2 + 4 * h, a.N2 / 8, aw);
    __syncthreads();
    MLP2_STAMP(3);

    // second layer: wave w -> features [32 w, 32 w + 32), [32 w + 128, ...)
    for (int f0 = 32 * wave; f0 < a.N3; f0 += 128) {
        const float* wrow = a.W3 + (int64_t)(f0 + j) * a.N2 + 4 * h;
        if (f0 >= 128) mlp2_prefetch(wrow, a.N2 / 8, aw);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = mlp2_block(wrow, s_2 + j * L2 + 4 * h, a.N2 / 8, aw, acc);
        const int64_t m = m0 + j;
        if (m < a.M) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (a.b3) bv = *reinterpret_cast<const f32x4*>(a.b3 + f0 + 8 * rq + 4 * h);
                f32x4 v;
                v.x = mlp2_act(acc[4 * rq + 0] + bv.x, a.act3, a.lo, a.hi); v.y = mlp2_act(acc[4 * rq + 1] + bv.y, a.act3, a.lo, a.hi);
                v.z = mlp2_act(acc[4 * rq + 2] + bv.z, a.act3, a.lo, a.hi); v.w = mlp2_act(acc[4 * rq + 3] + bv.w, a.act3, a.lo, a.hi);
                *reinterpret_cast<f32x4*>(a.C + m * a.ldc + f0 + 8 * rq + 4 * h) = v;
            }
        }
    }
    MLP2_STAMP(4);
#ifdef RECO_DIAG_MLP2_TRACE
    if (tid == 0 && blockIdx.x < 1024)
        g_mlp2_trace[8 * blockIdx.x + 5] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
#endif
}

}  // namespace

extern "C" int reco_mlp2_supported(int64_t K1, int64_t N2, int64_t N3)
{
    return K1 >= 8 && K1 <= 512 && K1 % 8 == 0 && N2 >= 32 && N2 <= 512 && N2 % 32 == 0 && N3 >= 32 && N3 <= 1024 && N3 % 32 == 0 &&
           (32 * (K1 + 4) + 32 * (N2 + 4)) * sizeof(float) <= 96 * 1024;
}

extern "C" int reco_mlp2_f32(const float* X, int64_t ldx, int64_t M, int64_t K1,
                             const float* W2, const float* b2, int64_t N2, int act2,
                             const float* W3, const float* b3, int64_t N3, int act3, float lo, float hi,
                             float* C, int64_t ldc, void* stream)
{
    if (M <= 0) return RECO_OK;
    RECO_REQUIRE(X && W2 && W3 && C, "mlp2: null pointer");
    RECO_REQUIRE(reco_mlp2_supported(K1, N2, N3), "mlp2: unsupported widths %lld -> %lld -> %lld (K1 a multiple of 8 up to 512, N2 / N3 "
                 "multiples of 32)", (long long)K1, (long long)N2, (long long)N3);
    RECO_REQUIRE(act2 >= RECO_ACT_NONE && act2 <= RECO_ACT_RATING && act3 >= RECO_ACT_NONE && act3 <= RECO_ACT_RATING, "mlp2: unknown activation");
    RECO_REQUIRE(ldx >= K1 && ldx % 4 == 0 && ldc >= N3 && ldc % 4 == 0, "mlp2: row strides must cover the rows and be multiples of 4");
    RECO_REQUIRE((uintptr_t)X % 16 == 0 && (uintptr_t)W2 % 16 == 0 && (uintptr_t)W3 % 16 == 0 && (uintptr_t)C % 16 == 0 &&
                 (uintptr_t)b2 % 16 == 0 && (uintptr_t)b3 % 16 == 0, "mlp2: X, W2, b2, W3, b3 and C must be 16-byte aligned (rows move as float4)");
    const int64_t blocks = ceil_div64(M, 32);
    RECO_REQUIRE(blocks < (1ll << 31), "mlp2 grid too large");
    Mlp2Args a{X, ldx, M, (int)K1, (int)N2, (int)N3, W2, b2, W3, b3, act2, act3, lo, hi, C, ldc};
    const size_t lds = (size_t)(32 * (K1 + 4) + 32 * (N2 + 4)) * sizeof(float);
    static bool attr[RECO_MAX_DEVICES] = {};
    if (int rc = reco_allow_lds(reinterpret_cast<const void*>(&mlp2_kernel), 96 * 1024, attr)) return rc;
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope prof(s, RECO_PROF_ENCODER);
        hipLaunchKernelGGL(mlp2_kernel, dim3((unsigned)blocks), dim3(256), lds, s, a);
    }
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

#ifdef RECO_DIAG_MLP2_TRACE
extern "C" int reco_debug_mlp2_trace(unsigned long long* host_out, int words)
{
    if (words > 8 * 1024) words = 8 * 1024;
    RECO_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_mlp2_trace), (size_t)words * sizeof(unsigned long long)));
    return RECO_OK;
}
#endif
