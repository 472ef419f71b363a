// K-c': Linear layer over a SPARSE (CSR) input,  C = act(X W^T + bias)  with X given as CSR -- gfx950.
//
// "Masked-observed handling" for the AutoRec / Hybrid encoders: their input is the rating matrix (or its transpose), 4.5 %
// dense on ML-1M, and the reference feeds the unobserved entries as literal zeros to a dense Linear
//   AutoRec.forward      src/autorec/utils/model.py:31-33  (X row i = item i's ratings, inference_engine.py:230)
//   AutoEncoder.encode   src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:78-89,116-117
// A zero contributes exactly nothing to an fp32 sum, so only the observed entries are visited:
//   C[m, :] = act( sum over the nonzeros (k, x) of row m, ascending k:  x * Wt[k, :]  + bias )
// with Wt = W^T stored [K, N] row-major (one transposed copy per weight version, reco_transpose_f32), i.e. one contiguous
// N-float row per nonzero.  The sum is an fmaf chain in ascending k -- the dense kernel's chain with the zero terms left out.
//
// Work is cut into SEGMENTS of at most seg_len nonzeros of one row (table built once per matrix by the host side: popular
// items have thousands of ratings, most have a few dozen), one 256-thread workgroup per segment and 512 columns; a thread owns
// two adjacent columns.  The segment's (k, x) pairs are staged in LDS, then UNROLL weight rows are in flight per thread.  A row
// with one segment is finished in place; rows with several segments leave partial sums that a second launch adds in segment
// order (deterministic) before bias and activation.  Bound: L2 / Infinity-Cache bandwidth, nnz * N * 4 bytes of weight rows.
#include "reco_common.h"

namespace {

constexpr int CSR_SEG_MAX = 256;
constexpr int CSR_UNROLL = 16;

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct CsrArgs {
    const int64_t* row_ptr;      // [M + 1]
    const int32_t* col;          // [nnz]  ascending within a row
    const float* val;            // [nnz]
    int64_t M, K;
    const int64_t* row_seg_ptr;  // [M + 1] first segment of each row (every row has at least one, possibly empty, segment)
    const int32_t* seg_row;      // [S]
    const int64_t* seg_begin;    // [S] offset of the segment's first nonzero in col / val
    int64_t S;
    int seg_len;
    const float* Wt;             // [K, N], row stride ldwt
    int64_t ldwt;
    const float* bias;           // [N] or null
    int64_t N;
    int act;
    float lo, hi;
    float* C;
    int64_t ldc;
    float* partial;              // [S, N] partial sums of multi-segment rows
    int vec2;                    // Wt rows can be read as float2 (N even, ldwt even, 8-byte aligned base)
};

__device__ __forceinline__ float csr_act(float v, int act, float lo, float hi)
{
    switch (act) {
        case RECO_ACT_RELU: return fmaxf(v, 0.f);
        case RECO_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        case RECO_ACT_CLAMP: return fminf(fmaxf(v, lo), hi);
        case RECO_ACT_RATING: return (1.f / (1.f + __expf(-v))) * (hi - lo) + lo;
        default: return v;
    }
}

__global__ __launch_bounds__(256) void csr_linear_kernel(CsrArgs a)
{
    __shared__ int32_t s_col[CSR_SEG_MAX];
    __shared__ float s_val[CSR_SEG_MAX];
    const int tid = threadIdx.x;
    const int64_t seg = blockIdx.x;
    const int64_t row = a.seg_row[seg];
    const int64_t b = a.seg_begin[seg];
    const int64_t row_end = a.row_ptr[row + 1];
    const int n_nz = (int)((b + a.seg_len < row_end ? b + a.seg_len : row_end) - b);
    if (tid < n_nz) {
        int32_t k = a.col[b + tid];
        s_col[tid] = k < 0 ? 0 : (k >= a.K ? (int32_t)(a.K - 1) : k);        // never read outside Wt
        s_val[tid] = a.val[b + tid];
    }
    __syncthreads();

    const int64_t n = (int64_t)blockIdx.y * 512 + 2 * tid;
    if (n >= a.N) return;
    const bool two = n + 1 < a.N;
    float acc0 = 0.f, acc1 = 0.f;
    for (int j = 0; j < n_nz; j += CSR_UNROLL) {
        f32x2 w[CSR_UNROLL];
#pragma unroll
        for (int u = 0; u < CSR_UNROLL; ++u) {
            const int jj = j + u < n_nz ? j + u : n_nz - 1;
            const float* src = a.Wt + (int64_t)s_col[jj] * a.ldwt + n;
            if (a.vec2) w[u] = *reinterpret_cast<const f32x2*>(src);
            else { w[u].x = src[0]; w[u].y = two ? src[1] : 0.f; }
        }
#pragma unroll
        for (int u = 0; u < CSR_UNROLL; ++u)
            if (j + u < n_nz) {                  // uniform over the workgroup
                const float x = s_val[j + u];
                acc0 = fmaf(x, w[u].x, acc0);
                acc1 = fmaf(x, w[u].y, acc1);
            }
    }
    const bool single = a.row_seg_ptr[row + 1] - a.row_seg_ptr[row] == 1;
    if (single) {
        a.C[row * a.ldc + n] = csr_act(acc0 + (a.bias ? a.bias[n] : 0.f), a.act, a.lo, a.hi);
        if (two) a.C[row * a.ldc + n + 1] = csr_act(acc1 + (a.bias ? a.bias[n + 1] : 0.f), a.act, a.lo, a.hi);
    } else {
        a.partial[seg * a.N + n] = acc0;
        if (two) a.partial[seg * a.N + n + 1] = acc1;
    }
}

// rows with several segments: partial sums added in segment order, then bias + activation
__global__ __launch_bounds__(256) void csr_reduce_kernel(CsrArgs a)
{
    const int64_t row = blockIdx.x;
    const int64_t s0 = a.row_seg_ptr[row], s1 = a.row_seg_ptr[row + 1];
    if (s1 - s0 <= 1) return;
    for (int64_t n = (int64_t)blockIdx.y * 256 + threadIdx.x; n < a.N; n += (int64_t)gridDim.y * 256) {
        float v = a.partial[s0 * a.N + n];
        for (int64_t s = s0 + 1; s < s1; ++s) v += a.partial[s * a.N + n];
        a.C[row * a.ldc + n] = csr_act(v + (a.bias ? a.bias[n] : 0.f), a.act, a.lo, a.hi);
    }
}

// out[c, r] = in[r, c] through a padded 32 x 32 LDS tile: coalesced on both sides
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int64_t rows, int64_t cols, int64_t ldi,
                                                        float* __restrict__ out, int64_t ldo)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < rows && c < cols) ? in[r * ldi + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < rows) out[c * ldo + r] = tile[tx][ty + 8 * i];
    }
}

}  // namespace

size_t csr_linear_workspace_bytes(int64_t S, int64_t N)
{
    return (S > 0 && N > 0) ? align_up((size_t)S * N * sizeof(float), 256) : 0;
}

int launch_csr_linear(const int64_t* row_ptr, const int32_t* col, const float* val, int64_t M, int64_t K,
                      const int64_t* row_seg_ptr, const int32_t* seg_row, const int64_t* seg_begin, int64_t S, int seg_len,
                      const float* Wt, int64_t ldwt, const float* bias, int64_t N, int act, float lo, float hi,
                      float* C, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t s)
{
    if (M <= 0 || N <= 0) return RECO_OK;
    RECO_REQUIRE(row_ptr && row_seg_ptr && seg_row && seg_begin && Wt && C, "csr_linear: null pointer");
    RECO_REQUIRE(K > 0 && S >= M, "csr_linear: every row needs at least one segment (S=%lld, M=%lld)", (long long)S, (long long)M);
    RECO_REQUIRE(seg_len >= 1 && seg_len <= CSR_SEG_MAX, "csr_linear: seg_len must be in [1,%d], got %d", CSR_SEG_MAX, seg_len);
    RECO_REQUIRE(ldwt >= N && ldc >= N, "csr_linear: leading dimension smaller than the row");
    RECO_REQUIRE(S < (1ll << 31) && M < (1ll << 31), "csr_linear: grid too large");
    const size_t need = S > M ? (size_t)S * N * sizeof(float) : 0;          // partial sums only exist when some row is split
    if (need && (!ws || ws_bytes < need)) {
        reco_set_error("csr_linear: workspace too small (%zu < %zu bytes)", ws_bytes, need);
        return RECO_ENOMEM;
    }
    CsrArgs a{row_ptr, col, val, M, K, row_seg_ptr, seg_row, seg_begin, S, seg_len, Wt, ldwt, bias, N, act, lo, hi, C, ldc,
              static_cast<float*>(ws), (N % 2 == 0 && ldwt % 2 == 0 && (uintptr_t)Wt % 8 == 0) ? 1 : 0};
    const unsigned by = (unsigned)ceil_div64(N, 512);
    hipLaunchKernelGGL(csr_linear_kernel, dim3((unsigned)S, by), dim3(256), 0, s, a);
    RECO_CHECK_LAUNCH();
    if (S > M) {
        hipLaunchKernelGGL(csr_reduce_kernel, dim3((unsigned)M, (unsigned)ceil_div64(N, 256)), dim3(256), 0, s, a);
        RECO_CHECK_LAUNCH();
    }
    return RECO_OK;
}

int launch_transpose(const float* in, int64_t rows, int64_t cols, int64_t ldi, float* out, int64_t ldo, hipStream_t s)
{
    if (rows <= 0 || cols <= 0) return RECO_OK;
    RECO_REQUIRE(in && out && ldi >= cols && ldo >= rows, "transpose: bad arguments");
    const int64_t bx = ceil_div64(cols, 32), by = ceil_div64(rows, 32);
    RECO_REQUIRE(by <= 65535 && bx < (1ll << 31), "transpose: matrix too large");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)bx, (unsigned)by), dim3(256), 0, s, in, rows, cols, ldi, out, ldo);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}
