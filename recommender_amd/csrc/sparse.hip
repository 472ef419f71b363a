// K-c': Linear layer over a SPARSE (CSR) input,  C = act(X W^T + bias)  with X given as CSR -- gfx950.
//
// "Masked-observed handling" for the AutoRec / Hybrid encoders: their input is the rating matrix (or its transpose), 4.5 %
// dense on ML-1M, and the reference feeds the unobserved entries as literal zeros to a dense Linear
//   AutoRec.forward      src/autorec/utils/model.py:31-33  (X row i = item i's ratings, inference_engine.py:230)
//   AutoEncoder.encode   src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:78-89,116-117
// A zero contributes exactly nothing to an fp32 sum, so only the observed entries are visited:
//   C[m, :] = act( sum over the nonzeros (k, x) of row m, ascending k:  x * Wt[k, :]  + bias )
// with Wt = W^T stored [K, N] row-major (one transposed copy per weight version, reco_transpose_f32), i.e. one contiguous
// N-float row per nonzero.  Summation order: inside a SEGMENT (below) an fmaf chain in ascending k starting from 0; a row that is
// ONE segment is therefore the dense kernel's k-ordered chain with the zero terms left out, bit for bit.  A row cut into several
// segments (one per k range it has entries in, and one per 256 entries) sums each segment on its own and then adds the partial
// sums in segment order with plain float adds -- a different rounding sequence, within the fp32 gate of the dense layer but not
// bit-identical to it.  No reference fixture pins this path bit for bit; its parity tests compare against the dense layer with a
// tolerance and reserve equality for single-segment rows.
//
// Work is cut into SEGMENTS: at most 256 consecutive nonzeros of one row whose k lie in ONE of eight k ranges (table built
// once per matrix by the host side; the ranges hold equal shares of the nonzeros).  One 256-thread workgroup per segment and
// 512 columns; a thread owns two adjacent columns.  The segment's (k, x) pairs are staged in LDS, then UNROLL weight rows are
// in flight per thread.  A row with one segment is finished in place; rows with several segments leave partial sums that a
// second launch adds in segment order -- ascending k, deterministic -- before bias and activation.
// XCD placement: workgroup b runs the segments in slots [4 b, 4 b + 4) of the placement table order[], and the host fills the
// table so that every slot of workgroup b holds a segment of k range b % 8.
// Workgroups b and b + 8 share an XCD (observed round-robin placement; speed only, never correctness), so every XCD walks one
// eighth of Wt (1.5 MB of the 12-MB AutoRec encoder weight) and finds it in its own 4-MB L2 instead of the Infinity Cache.
// Bound: L2 bandwidth, nnz * N * 4 bytes of weight rows.
#include "reco_common.h"

// Weight rows in flight per wave (same-box A/B, round 3; AutoRec encoder N = 500 / Hybrid user and item encoders N = 256, ms):
//   8 columns per lane:  2: 0.1338   3: 0.1332   4: 0.1318   6: 0.1330   8: 0.1375   12: 0.150   16: 0.174
//   4 columns per lane:  8: 0.0760 / 0.111   16: 0.0775 / 0.1071   24: 0.090 / 0.119   32: 0.097 / 0.128
#ifndef RECO_CSR_U8
#define RECO_CSR_U8 4      // weight rows in flight per wave, 8 columns per lane (N > 256)
#endif
#ifndef RECO_CSR_U4
#define RECO_CSR_U4 16     // 4 columns per lane (N <= 256)
#endif

namespace {

constexpr int CSR_SEG_MAX = 256;
constexpr int CSR_UNROLL = 16;
constexpr int CSR_SEGS_PER_WG = RECO_CSR_SEGS_PER_WG;     // slots [4 b, 4 b + 4) of the placement table belong to workgroup b

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct CsrArgs {
    const int64_t* row_ptr;      // [M + 1]
    const int32_t* col;          // [nnz]  ascending within a row
    const float* val;            // [nnz]
    int64_t M, K;
    const int64_t* row_seg_ptr;  // [M + 1] first segment of each row (a row without nonzeros has none)
    const int32_t* seg_row;      // [S]
    const int64_t* seg_begin;    // [S] offset of the segment's first nonzero in col / val
    const int64_t* seg_end;      // [S] one past its last nonzero (at most CSR_SEG_MAX further)
    int64_t S;
    const int32_t* order;        // nullable [n_slots]: slot s runs segment order[s] (< 0: none); null: segment s
    int64_t n_slots;
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
    int c_vec4;                  // C rows can be written as float4
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

// Generic form (any N, any alignment): the workgroup's 256 threads take the group's segments one after another.
__global__ __launch_bounds__(256) void csr_linear_kernel(CsrArgs a)
{
    __shared__ int32_t s_col[CSR_SEG_MAX];
    __shared__ float s_val[CSR_SEG_MAX];
    const int tid = threadIdx.x;
    const int64_t n = (int64_t)blockIdx.y * 512 + 2 * tid;
    const bool two = n + 1 < a.N;
    for (int w = 0; w < CSR_SEGS_PER_WG; ++w) {
        const int64_t slot = (int64_t)blockIdx.x * CSR_SEGS_PER_WG + w;
        if (slot >= a.n_slots) break;
        const int64_t seg = a.order ? (int64_t)a.order[slot] : slot;
        if (seg < 0 || seg >= a.S) continue;             // padding of the placement table (uniform over the workgroup)
        const int64_t row = a.seg_row[seg];
        const int64_t b = a.seg_begin[seg];
        int64_t e = a.seg_end[seg];
        if (e > b + CSR_SEG_MAX) e = b + CSR_SEG_MAX;    // never beyond the LDS staging arrays
        const int n_nz = e > b ? (int)(e - b) : 0;
        __syncthreads();                                 // the previous segment's readers are done with the staging arrays
        if (tid < n_nz) {
            int32_t k = a.col[b + tid];
            s_col[tid] = k < 0 ? 0 : (k >= a.K ? (int32_t)(a.K - 1) : k);        // never read outside Wt
            s_val[tid] = a.val[b + tid];
        }
        __syncthreads();
        if (n >= a.N) continue;
        float acc0 = 0.f, acc1 = 0.f;
        for (int j = 0; j < n_nz; j += CSR_UNROLL) {
            f32x2 w2[CSR_UNROLL];
#pragma unroll
            for (int u = 0; u < CSR_UNROLL; ++u) {
                const int jj = j + u < n_nz ? j + u : n_nz - 1;
                const float* src = a.Wt + (int64_t)s_col[jj] * a.ldwt + n;
                if (a.vec2) w2[u] = *reinterpret_cast<const f32x2*>(src);
                else { w2[u].x = src[0]; w2[u].y = two ? src[1] : 0.f; }
            }
#pragma unroll
            for (int u = 0; u < CSR_UNROLL; ++u)
                if (j + u < n_nz) {                  // uniform over the workgroup
                    const float x = s_val[j + u];
                    acc0 = fmaf(x, w2[u].x, acc0);
                    acc1 = fmaf(x, w2[u].y, acc1);
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
}

// Fast form (N a multiple of 4, 16-byte aligned weight rows): one WAVE per segment, a lane owns CPL = 4 or 8 adjacent
// columns.  Everything about a nonzero is wave-uniform -- its k, its value, the address of its weight row -- so it lives in
// scalar registers (s_load of col / val, scalar address arithmetic) and the weight row is fetched with "scalar base + lane
// offset" loads: no LDS, no barrier, no vector instruction outside the loads and the FMAs.  UNROLL rows (16 KB) are in flight
// per wave.  Same fmaf chain per column as the generic form: bit-identical results.
template <int CPL>
__global__ __launch_bounds__(256) void csr_linear_wave_kernel(CsrArgs a)
{
    constexpr int U = CPL == 8 ? RECO_CSR_U8 : RECO_CSR_U4;
    constexpr int NV = CPL / 4;                              // float4 per lane and nonzero
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t slot = (int64_t)blockIdx.x * CSR_SEGS_PER_WG + wave;
    if (slot >= a.n_slots) return;
    const int64_t seg = a.order ? (int64_t)a.order[slot] : slot;
    if (seg < 0 || seg >= a.S) return;
    const int64_t row = a.seg_row[seg];
    const int64_t b = a.seg_begin[seg];
    const int64_t e = a.seg_end[seg];
    const int64_t n = (int64_t)blockIdx.y * (64 * CPL) + (int64_t)lane * CPL;
    bool ok[NV];
    f32x4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { ok[v] = n + 4 * v + 3 < a.N; acc[v] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int64_t j = b; j < e; j += U) {
        f32x4 w[U][NV];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t jj = j + u < e ? j + u : e - 1;
            int64_t k = a.col[jj];                           // uniform: a scalar load
            k = k < 0 ? 0 : (k >= a.K ? a.K - 1 : k);        // never read outside Wt
            const float* wrow = a.Wt + k * a.ldwt + n;
#pragma unroll
            for (int v = 0; v < NV; ++v) w[u][v] = ok[v] ? *reinterpret_cast<const f32x4*>(wrow + 4 * v) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (j + u < e) {                                 // uniform
                const float x = a.val[j + u];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    acc[v].x = fmaf(x, w[u][v].x, acc[v].x);
                    acc[v].y = fmaf(x, w[u][v].y, acc[v].y);
                    acc[v].z = fmaf(x, w[u][v].z, acc[v].z);
                    acc[v].w = fmaf(x, w[u][v].w, acc[v].w);
                }
            }
    }
    const bool single = a.row_seg_ptr[row + 1] - a.row_seg_ptr[row] == 1;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (!ok[v]) continue;
        const int64_t c = n + 4 * v;
        if (single) {
            f32x4 r = acc[v];
            if (a.bias) { const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + c); r = r + bv; }
            r.x = csr_act(r.x, a.act, a.lo, a.hi); r.y = csr_act(r.y, a.act, a.lo, a.hi);
            r.z = csr_act(r.z, a.act, a.lo, a.hi); r.w = csr_act(r.w, a.act, a.lo, a.hi);
            float* dst = a.C + row * a.ldc + c;
            if (a.c_vec4) *reinterpret_cast<f32x4*>(dst) = r;
            else { dst[0] = r.x; dst[1] = r.y; dst[2] = r.z; dst[3] = r.w; }
        } else {
            *reinterpret_cast<f32x4*>(a.partial + seg * a.N + c) = acc[v];          // N % 4 == 0: aligned
        }
    }
}

// rows with several segments: partial sums added in segment order, then bias + activation; rows without any (no nonzero): bias
__global__ __launch_bounds__(256) void csr_reduce_kernel(CsrArgs a)
{
    const int64_t row = blockIdx.x;
    const int64_t s0 = a.row_seg_ptr[row], s1 = a.row_seg_ptr[row + 1];
    if (s1 - s0 == 1) return;
    for (int64_t n = (int64_t)blockIdx.y * 256 + threadIdx.x; n < a.N; n += (int64_t)gridDim.y * 256) {
        float v = 0.f;
        if (s1 > s0) v = a.partial[s0 * a.N + n];
        // eight partial sums in flight, added in segment order (a popular row has dozens: one dependent load per add was the
        // launch's tail)
        for (int64_t s = s0 + 1; s < s1; s += 8) {
            float p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = s + u < s1 ? a.partial[(s + u) * a.N + n] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s + u < s1) v += p[u];                      // uniform per row
        }
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
                      const int64_t* row_seg_ptr, const int32_t* seg_row, const int64_t* seg_begin, const int64_t* seg_end,
                      int64_t S, const int32_t* order, int64_t n_order, int single_segment_rows,
                      const float* Wt, int64_t ldwt, const float* bias, int64_t N, int act, float lo, float hi,
                      float* C, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t s)
{
    if (M <= 0 || N <= 0) return RECO_OK;
    RECO_REQUIRE(row_ptr && row_seg_ptr && Wt && C, "csr_linear: null pointer");
    RECO_REQUIRE(S == 0 || (seg_row && seg_begin && seg_end), "csr_linear: segment table missing");
    RECO_REQUIRE(K > 0 && S >= 0, "csr_linear: bad sizes");
    RECO_REQUIRE(ldwt >= N && ldc >= N, "csr_linear: leading dimension smaller than the row");
    RECO_REQUIRE(!order || n_order >= 0, "csr_linear: bad placement table");
    const int64_t n_slots = order ? n_order : S;
    const int64_t grid = ceil_div64(n_slots, CSR_SEGS_PER_WG);
    RECO_REQUIRE(grid < (1ll << 31) && M < (1ll << 31), "csr_linear: grid too large");
    // partial sums exist unless the caller states that every row has exactly one segment
    const size_t need = single_segment_rows ? 0 : (size_t)S * N * sizeof(float);
    if (need && (!ws || ws_bytes < need)) {
        reco_set_error("csr_linear: workspace too small (%zu < %zu bytes)", ws_bytes, need);
        return RECO_ENOMEM;
    }
    CsrArgs a{row_ptr, col, val, M, K, row_seg_ptr, seg_row, seg_begin, seg_end, S, order, n_slots, Wt, ldwt, bias, N, act, lo, hi,
              C, ldc, static_cast<float*>(ws), (N % 2 == 0 && ldwt % 2 == 0 && (uintptr_t)Wt % 8 == 0) ? 1 : 0,
              (ldc % 4 == 0 && (uintptr_t)C % 16 == 0) ? 1 : 0};
    // wave-per-segment form when weight rows (and the partial rows) can move as float4
    const bool fast = N % 4 == 0 && ldwt % 4 == 0 && (uintptr_t)Wt % 16 == 0 && (!bias || (uintptr_t)bias % 16 == 0) &&
                      (single_segment_rows || (uintptr_t)ws % 16 == 0);
    if (grid > 0) {
        ProfScope prof(s, RECO_PROF_CSR);
        if (fast && N > 256)
            hipLaunchKernelGGL(csr_linear_wave_kernel<8>, dim3((unsigned)grid, (unsigned)ceil_div64(N, 512)), dim3(256), 0, s, a);
        else if (fast)
            hipLaunchKernelGGL(csr_linear_wave_kernel<4>, dim3((unsigned)grid, 1), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(csr_linear_kernel, dim3((unsigned)grid, (unsigned)ceil_div64(N, 512)), dim3(256), 0, s, a);
        RECO_CHECK_LAUNCH();
    }
    if (!single_segment_rows) {
        ProfScope prof(s, RECO_PROF_CSR_REDUCE);
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
