// K-a: embedding row gathers, gfx950.  HBM-bound: 8 + 8 D bytes per row (id + row read + row write).
//
// Replaces nn.Embedding.__call__ / mul / cat      src/helpers/ncf_model.py:207-211,220-224
//
// A row of D floats is moved by D/4 consecutive lanes as float4, so a wave instruction touches
// whole 128-B lines of at most 64 / (D/4) rows (D = 32: 8 rows x 128 B; D = 128: 2 rows x 512 B)
// and the write side is fully coalesced.  Each lane loads its own row's id; the 8 / 32 lanes of a
// row hit one address, which the texture addresser coalesces.  The scoring kernels in tower.hip gather straight into LDS;
// these stand-alone forms exist for callers that need the materialised tensors and for the
// HBM-roofline measurement of the gather itself.
#include "reco_common.h"

// Loads in flight per thread x workgroups, same-box A/B on HBM-resident tables (tools/ab_run.sh; fused 4-table gather of 2^22 pairs /
// gather_rows D = 128, ms):  4 x 4096 (round 2): 2.001 / 0.832    8 x 4096: 2.282 / 0.886    2 x 4096: 1.941 / 0.772    3 x 4096: 1.969 / 0.785
//   1 x 4096: 1.821 / 0.772    1 x 8192: 1.859 / 0.789    1 x 3072: 2.046 / 0.822    1 x 1024: 2.497 / 0.839    **1 x 2048: 1.800 / 0.767**
// One float4 per thread with every CU's 32 wave slots filled (8 workgroups of 256 per CU) beats deeper per-thread unrolling: the
// unrolled loads of a thread lie a whole grid sweep apart and scatter the write stream over as many DRAM pages.
// Non-temporal LOADS of the table rows (the stores already are) lose: 1.798 -> 1.933 ms fused (an item row is read four times at
// 4 M pairs over 1 M items and the caches were serving the repeats), gather_rows unchanged.
#ifndef RECO_GATHER_UNROLL
#define RECO_GATHER_UNROLL 1
#endif
#ifndef RECO_GATHER_BLOCKS
#define RECO_GATHER_BLOCKS 2048
#endif

namespace {

__device__ __forceinline__ int64_t clamp_row(int64_t id, int64_t n, int32_t* oob, bool count)
{
    if (id < 0 || id >= n) {
        if (oob && count) atomicAdd(oob, 1);
        return id < 0 ? 0 : n - 1;
    }
    return id;
}

template <bool POW2>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, int64_t rows, int dim4, int sh,
                                                          const int64_t* __restrict__ idx, int64_t B,
                                                          float* __restrict__ out, int32_t* oob)
{
    // UNROLL independent float4 per thread in flight (ids first, then rows, then the streaming stores); see the sweep above
    constexpr int UNROLL = RECO_GATHER_UNROLL;
    const int64_t total = B * dim4;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x; t0 < total; t0 += stride * UNROLL) {
        int64_t r[UNROLL];
        int c[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t t = t0 + u * stride;
            const int64_t b = t < total ? (POW2 ? t >> sh : t / dim4) : 0;
            c[u] = (int)((t < total ? t : 0) - b * dim4);
            r[u] = t < total ? clamp_row(idx[b], rows, oob, c[u] == 0) : 0;
        }
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = reinterpret_cast<const f32x4*>(table)[r[u] * dim4 + c[u]];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t t = t0 + u * stride;
            if (t < total) __builtin_nontemporal_store(v[u], reinterpret_cast<f32x4*>(out) + t);
        }
    }
}

struct FusedArgs {
    const float* Eu_g; const float* Ei_g; const float* Eu_m; const float* Ei_m;
    int64_t n_users, n_items;
    int g4, m4;                  // float4 per GMF row / per MLP row
    const int64_t* user; const int64_t* item;
    int64_t B;
    float* gmf_prod; float* mlp_in;
    int64_t ldg4, ldm4;          // output row strides in float4
    int32_t* oob;
};

// per pair: g4 float4 of gmf product + 2*m4 float4 of [user | item] MLP rows; work items are laid
// out so that consecutive lanes write consecutive float4 of the outputs.
// POW2: both row widths are powers of two (the shipped 32 / 128): pair and column come from a shift and a mask instead
// of a 64-bit division per float4 -- at 5 TB/s the index arithmetic is otherwise a visible share of the kernel.
template <bool POW2>
__global__ __launch_bounds__(256) void neumf_gather_kernel(FusedArgs a, int sh_m, int sh_g)
{
    constexpr int UNROLL = RECO_GATHER_UNROLL;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (a.mlp_in) {
        const int per_pair = 2 * a.m4;
        const int64_t total = a.B * per_pair;
        for (int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x; t0 < total; t0 += stride * UNROLL) {
            const f32x4* src[UNROLL];
            int64_t dst[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t t = t0 + u * stride;
                const bool ok = t < total;
                const int64_t b = ok ? (POW2 ? t >> sh_m : t / per_pair) : 0;
                const int c = ok ? (POW2 ? (int)(t & (per_pair - 1)) : (int)(t - b * per_pair)) : 0;
                dst[u] = ok ? b * a.ldm4 + c : -1;
                if (c < a.m4) src[u] = reinterpret_cast<const f32x4*>(a.Eu_m) + clamp_row(a.user[b], a.n_users, a.oob, ok && c == 0) * a.m4 + c;
                else src[u] = reinterpret_cast<const f32x4*>(a.Ei_m) + clamp_row(a.item[b], a.n_items, a.oob, ok && c == a.m4) * a.m4 + (c - a.m4);
            }
            f32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = *src[u];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (dst[u] >= 0) __builtin_nontemporal_store(v[u], reinterpret_cast<f32x4*>(a.mlp_in) + dst[u]);
        }
    }
    if (a.gmf_prod) {
        const int64_t total = a.B * a.g4;
        for (int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x; t0 < total; t0 += stride * UNROLL) {
            const f32x4 *su[UNROLL], *si[UNROLL];
            int64_t dst[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t t = t0 + u * stride;
                const bool ok = t < total;
                const int64_t b = ok ? (POW2 ? t >> sh_g : t / a.g4) : 0;
                const int c = ok ? (POW2 ? (int)(t & (a.g4 - 1)) : (int)(t - b * a.g4)) : 0;
                dst[u] = ok ? b * a.ldg4 + c : -1;
                su[u] = reinterpret_cast<const f32x4*>(a.Eu_g) + clamp_row(a.user[b], a.n_users, a.oob, ok && c == 0 && !a.mlp_in) * a.g4 + c;
                si[u] = reinterpret_cast<const f32x4*>(a.Ei_g) + clamp_row(a.item[b], a.n_items, a.oob, ok && c == 0 && !a.mlp_in) * a.g4 + c;
            }
            f32x4 vu[UNROLL], vi[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { vu[u] = *su[u]; vi[u] = *si[u]; }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (dst[u] >= 0) __builtin_nontemporal_store(vu[u] * vi[u], reinterpret_cast<f32x4*>(a.gmf_prod) + dst[u]);
        }
    }
}

// Any row width / alignment (an embedding of factor_num = 10, a view at an odd column): one float per thread.  Off the hot path --
// every shipped model has widths that are multiples of 4 -- but nn.Embedding takes any width and so do these entry points.
__global__ __launch_bounds__(256) void gather_rows_any_kernel(const float* __restrict__ table, int64_t rows, int dim,
                                                              const int64_t* __restrict__ idx, int64_t B, float* __restrict__ out, int32_t* oob)
{
    const int64_t total = B * dim;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t b = t / dim;
        const int c = (int)(t - b * dim);
        out[t] = table[clamp_row(idx[b], rows, oob, c == 0) * dim + c];
    }
}

struct AnyArgs {
    const float* Eu_g; const float* Ei_g; const float* Eu_m; const float* Ei_m;
    int64_t n_users, n_items;
    int dg, dm;
    const int64_t* user; const int64_t* item;
    int64_t B;
    float* gmf_prod; float* mlp_in;
    int64_t ldg, ldm;            // output row strides in floats
    int32_t* oob;
};

__global__ __launch_bounds__(256) void neumf_gather_any_kernel(AnyArgs a)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (a.mlp_in) {
        const int per_pair = 2 * a.dm;
        const int64_t total = a.B * per_pair;
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
            const int64_t b = t / per_pair;
            const int c = (int)(t - b * per_pair);
            a.mlp_in[b * a.ldm + c] = c < a.dm ? a.Eu_m[clamp_row(a.user[b], a.n_users, a.oob, c == 0) * a.dm + c]
                                               : a.Ei_m[clamp_row(a.item[b], a.n_items, a.oob, c == a.dm) * a.dm + (c - a.dm)];
        }
    }
    if (a.gmf_prod) {
        const int64_t total = a.B * a.dg;
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
            const int64_t b = t / a.dg;
            const int c = (int)(t - b * a.dg);
            const bool count = c == 0 && !a.mlp_in;
            a.gmf_prod[b * a.ldg + c] = a.Eu_g[clamp_row(a.user[b], a.n_users, a.oob, count) * a.dg + c] *
                                        a.Ei_g[clamp_row(a.item[b], a.n_items, a.oob, count) * a.dg + c];
        }
    }
}

static int log2_exact(int v)
{
    int s = 0;
    while ((1 << s) < v) ++s;
    return (v > 0 && (1 << s) == v) ? s : -1;
}

unsigned stream_grid(int64_t work_items)
{
    const int64_t blocks = ceil_div64(work_items, 256);
    return (unsigned)(blocks < RECO_GATHER_BLOCKS ? (blocks > 0 ? blocks : 1) : RECO_GATHER_BLOCKS);
}

}  // namespace

extern "C" int reco_gather_rows(const float* table, int64_t rows, int dim, const int64_t* idx, int64_t B,
                                float* out, int32_t* oob, void* stream)
{
    if (B <= 0) return RECO_OK;
    RECO_REQUIRE(table && idx && out, "gather_rows: null pointer");
    RECO_REQUIRE(dim > 0, "gather_rows: dim must be positive, got %d", dim);
    RECO_REQUIRE(rows > 0, "gather_rows: empty table");
    ProfScope prof((hipStream_t)stream, RECO_PROF_GATHER);
    if (dim % 4 != 0 || (uintptr_t)table % 16 != 0 || (uintptr_t)out % 16 != 0) {      // any width / alignment: the scalar form
        hipLaunchKernelGGL(gather_rows_any_kernel, dim3(stream_grid(B * dim)), dim3(256), 0, (hipStream_t)stream, table, rows, dim, idx, B, out, oob);
        RECO_CHECK_LAUNCH();
        return RECO_OK;
    }
    const int sh = log2_exact(dim / 4);
    if (sh >= 0)
        hipLaunchKernelGGL(gather_rows_kernel<true>, dim3(stream_grid(B * (dim / 4))), dim3(256), 0, (hipStream_t)stream,
                           table, rows, dim / 4, sh, idx, B, out, oob);
    else
        hipLaunchKernelGGL(gather_rows_kernel<false>, dim3(stream_grid(B * (dim / 4))), dim3(256), 0, (hipStream_t)stream,
                           table, rows, dim / 4, 0, idx, B, out, oob);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

extern "C" int reco_neumf_gather(const float* Eu_g, const float* Ei_g, const float* Eu_m, const float* Ei_m,
                                 int64_t n_users, int64_t n_items, int dg, int dm,
                                 const int64_t* user, const int64_t* item, int64_t B,
                                 float* gmf_prod, int64_t ld_gmf, float* mlp_in, int64_t ld_mlp,
                                 int32_t* oob, void* stream)
{
    if (B <= 0) return RECO_OK;
    RECO_REQUIRE(user && item, "neumf_gather: null id pointer");
    RECO_REQUIRE(!gmf_prod || (Eu_g && Ei_g && dg > 0), "neumf_gather: bad GMF tables / dg=%d", dg);
    RECO_REQUIRE(!mlp_in || (Eu_m && Ei_m && dm > 0), "neumf_gather: bad MLP tables / dm=%d", dm);
    RECO_REQUIRE(n_users > 0 && n_items > 0, "neumf_gather: empty tables");
    if (ld_gmf == 0) ld_gmf = dg;
    if (ld_mlp == 0) ld_mlp = 2 * dm;
    RECO_REQUIRE((!gmf_prod || ld_gmf >= dg) && (!mlp_in || ld_mlp >= 2 * dm),
                 "neumf_gather: bad output strides %lld / %lld", (long long)ld_gmf, (long long)ld_mlp);
    const bool vec4 = (!gmf_prod || (dg % 4 == 0 && ld_gmf % 4 == 0)) && (!mlp_in || (dm % 4 == 0 && ld_mlp % 4 == 0)) &&
                      (uintptr_t)gmf_prod % 16 == 0 && (uintptr_t)mlp_in % 16 == 0 && (uintptr_t)Eu_g % 16 == 0 &&
                      (uintptr_t)Ei_g % 16 == 0 && (uintptr_t)Eu_m % 16 == 0 && (uintptr_t)Ei_m % 16 == 0;
    if (!vec4) {                                         // any width / alignment: the scalar form
        AnyArgs any{Eu_g, Ei_g, Eu_m, Ei_m, n_users, n_items, dg, dm, user, item, B, gmf_prod, mlp_in, ld_gmf, ld_mlp, oob};
        ProfScope prof((hipStream_t)stream, RECO_PROF_GATHER);
        hipLaunchKernelGGL(neumf_gather_any_kernel, dim3(stream_grid(B * (mlp_in ? 2 * dm : dg))), dim3(256), 0, (hipStream_t)stream, any);
        RECO_CHECK_LAUNCH();
        return RECO_OK;
    }
    FusedArgs a{Eu_g, Ei_g, Eu_m, Ei_m, n_users, n_items, dg / 4, dm / 4, user, item, B, gmf_prod, mlp_in, ld_gmf / 4, ld_mlp / 4, oob};
    const int64_t work = B * (mlp_in ? 2 * (dm / 4) : dg / 4);
    const int sh_m = mlp_in ? log2_exact(2 * (dm / 4)) : 0, sh_g = gmf_prod ? log2_exact(dg / 4) : 0;
    ProfScope prof((hipStream_t)stream, RECO_PROF_GATHER);
    if (sh_m >= 0 && sh_g >= 0)
        hipLaunchKernelGGL(neumf_gather_kernel<true>, dim3(stream_grid(work)), dim3(256), 0, (hipStream_t)stream, a, sh_m, sh_g);
    else
        hipLaunchKernelGGL(neumf_gather_kernel<false>, dim3(stream_grid(work)), dim3(256), 0, (hipStream_t)stream, a, 0, 0);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}
