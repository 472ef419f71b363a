// Diagnostic microbenchmark (not part of the product): what rate does v_mfma_f32_32x32x2_f32 sustain on this chip
// under the tile kernel's conditions (2 accumulator chains, 1 or 2 waves per SIMD, LDS reads + VALU between)?
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o build/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: bare MFMA, 2 chains; 1: + 3 ds_read_b128 + 8 VALU per 8 MFMAs; 2: single chain
__global__ void probe(float* out, unsigned long long* stamps, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    f32x4 a0 = {1.f, 0.5f, 0.25f, 0.125f}, a1 = {0.5f, 0.25f, 1.f, 2.f}, x = {1.f, 1.f, 1.f, 1.f};
    f32x4 p = {0.1f, 0.2f, 0.3f, 0.4f};
    const f32x4* w = reinterpret_cast<const f32x4*>(lds) + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 xn = x, a0n = a0, a1n = a1;
        if (MODE == 1) {
            const int o = (it & 15) * 64;
            xn = w[o]; a0n = w[o + 1024]; a1n = w[o + 2048];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, x.x, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, x.x, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, x.y, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, x.y, acc0, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, x.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, x.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, x.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, x.y, acc1, 0, 0, 0);
        }
        if (MODE == 1) {
            __builtin_amdgcn_sched_barrier(0);
            xn.x = fmaxf(xn.x + p.x, 0.f); xn.y = fmaxf(xn.y + p.y, 0.f); xn.z = fmaxf(xn.z + p.z, 0.f); xn.w = fmaxf(xn.w + p.w, 0.f);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, x.z, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, x.z, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, x.w, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, x.w, acc0, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, x.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, x.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, x.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, x.w, acc1, 0, 0, 0);
        }
        if (MODE == 1) { __builtin_amdgcn_sched_barrier(0); x = xn; a0 = a0n; a1 = a1n; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int threads, int iters)
{
    const int blocks = 256;
    float* out; unsigned long long* st;
    hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&st, 16 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, st, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0; for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; rt += h[2 * b + 1]; }
    cyc /= blocks; rt /= blocks;
    const double waves_per_simd = threads / 256.0;
    const double mfma_per_simd = 8.0 * iters * waves_per_simd;
    const double flops = 4096.0 * 8.0 * iters * (threads / 64) * blocks;
    printf("%-28s thr %4d  %.3f ms  %.1f TFLOP/s  cycles/MFMA/SIMD %.1f  clock %.0f MHz\n", name, threads, ms, flops / ms / 1e9,
           cyc / mfma_per_simd, cyc / rt * 100.0);
    hipFree(out); hipFree(st);
}


// flexible probe: per iteration 8 MFMAs (2 chains) with NDS ds_read_b128 and NV VALU (add+max pairs count as 2) placed by PLACE
//   PLACE 0: all ds_reads first, all VALU after the 4th MFMA (the tile kernel's schedule)
//   PLACE 1: ds_reads first, one VALU after each MFMA
//   PLACE 2: everything (ds_reads, waitcnt, VALU) before the 8 MFMAs
template <int NDS, int NV, int PLACE, bool FEED>
__global__ void probe2(float* out, unsigned long long* stamps, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float a[8] = {1.f, 0.5f, 0.25f, 0.125f, 0.5f, 0.25f, 1.f, 2.f};
    float x[4] = {1.f, 1.f, 1.f, 1.f};
    float v[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
    const f32x4* w = reinterpret_cast<const f32x4*>(lds) + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 ld[NDS > 0 ? NDS : 1];
        const int o = (it & 15) * 64;
#pragma unroll
        for (int d = 0; d < NDS; ++d) ld[d] = w[o + 1024 * d];
        __builtin_amdgcn_sched_barrier(0);
        if (PLACE == 2) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k & 7] = fmaxf(v[k & 7] + (NDS > 0 ? ld[k % (NDS > 0 ? NDS : 1)][k & 3] : 0.25f), 0.f);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float bx = FEED ? v[m & 3] : x[m & 3];
            if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bx, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bx, acc0, 0, 0, 0);
            if (PLACE == 1) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = m * NV / 8; k < (m + 1) * NV / 8; ++k)
                    v[4 + (k & 3)] = fmaxf(v[4 + (k & 3)] + (NDS > 0 ? ld[k % (NDS > 0 ? NDS : 1)][k & 3] : 0.25f), 0.f);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (PLACE == 0 && m == 3) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < NV; ++k) v[4 + (k & 3)] = fmaxf(v[4 + (k & 3)] + (NDS > 0 ? ld[k % (NDS > 0 ? NDS : 1)][k & 3] : 0.25f), 0.f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PLACE != 2) { v[0] = v[4]; v[1] = v[5]; v[2] = v[6]; v[3] = v[7]; }
        if (NDS > 0 && NV == 0) asm volatile("" :: "v"(ld[0]), "v"(ld[NDS - 1]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NDS, int NV, int PLACE, bool FEED>
void run2(int threads, int iters)
{
    const int blocks = 256;
    float* out; unsigned long long* st;
    hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&st, 16 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe2<NDS, NV, PLACE, FEED>), dim3(blocks), dim3(threads), 0, 0, out, st, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 4096.0 * 8.0 * iters * (threads / 64) * blocks;
    printf("ds %d valu %2d place %d feed %d  thr %4d  %.3f ms  %.1f TFLOP/s (%.1f%% of 155.3)\n", NDS, NV, PLACE, (int)FEED, threads, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 155.3 * 100);
    hipFree(out); hipFree(st);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// the tile kernel's quad: 3 ds_read_b128, then x = relu(p + q) for 4 values feeding the next 8 MFMAs
//   VAR 0: 4 v_add + 4 v_max;  VAR 1: 2 v_pk_add_f32 + 4 v_max;  VAR 2: 2 v_pk_add_f32 clamp (relu for values <= 1)
template <int VAR, int NDS>
__global__ void probe3(float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)(i & 7) * 0.0625f;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    f32x4 a0 = {1.f, 0.5f, 0.25f, 0.125f}, a1 = {0.5f, 0.25f, 1.f, 2.f}, x = {0.5f, 0.5f, 0.5f, 0.5f};
    const f32x4 p = {0.1f, 0.2f, 0.3f, 0.4f};
    const f32x4* w = reinterpret_cast<const f32x4*>(lds) + lane;
    for (int it = 0; it < iters; ++it) {
        const int o = (it & 15) * 64;
        f32x4 xr = w[o], a0n = a0, a1n = a1;
        if (NDS == 3) { a0n = w[o + 1024]; a1n = w[o + 2048]; }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, x.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, x.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, x.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, x.y, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 xn;
        if (VAR == 0) {
            xn.x = fmaxf(xr.x + p.x, 0.f); xn.y = fmaxf(xr.y + p.y, 0.f); xn.z = fmaxf(xr.z + p.z, 0.f); xn.w = fmaxf(xr.w + p.w, 0.f);
        } else if (VAR == 1) {
            f32x2 lo = {xr.x, xr.y}, hi = {xr.z, xr.w}, plo = {p.x, p.y}, phi = {p.z, p.w}, rlo, rhi;
            asm volatile("v_pk_add_f32 %0, %2, %3\n\tv_pk_add_f32 %1, %4, %5" : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(plo), "v"(hi), "v"(phi));
            xn.x = fmaxf(rlo.x, 0.f); xn.y = fmaxf(rlo.y, 0.f); xn.z = fmaxf(rhi.x, 0.f); xn.w = fmaxf(rhi.y, 0.f);
        } else {
            f32x2 lo = {xr.x, xr.y}, hi = {xr.z, xr.w}, plo = {p.x, p.y}, phi = {p.z, p.w}, rlo, rhi;
            asm volatile("v_pk_add_f32 %0, %2, %3 clamp\n\tv_pk_add_f32 %1, %4, %5 clamp\n\ts_nop 1" : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(plo), "v"(hi), "v"(phi));
            xn.x = rlo.x; xn.y = rlo.y; xn.z = rhi.x; xn.w = rhi.y;
        }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, x.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, x.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, x.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, x.w, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        x = xn; a0 = a0n; a1 = a1n;
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VAR, int NDS>
void run3(int threads, int iters)
{
    const int blocks = 256;
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe3<VAR, NDS>), dim3(blocks), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 4096.0 * 8.0 * iters * (threads / 64) * blocks;
    float h0; hipMemcpy(&h0, out, 4, hipMemcpyDeviceToHost);
    printf("quad var %d ds %d  thr %4d  %.3f ms  %.1f TFLOP/s (%.1f%%)  chk %g\n", VAR, NDS, threads, ms, flops / ms / 1e9, flops / ms / 1e9 / 155.3 * 100, h0);
    hipFree(out);
}

int main()
{
    const int iters = 20000;
    run<0>("bare 2 chains", 256, iters);
    run<0>("bare 2 chains", 512, iters);
    run<2>("bare 1 chain", 256, iters);
    run<2>("bare 1 chain", 512, iters);
    run<1>("lds+valu 2 chains", 256, iters);
    run<1>("lds+valu 2 chains", 512, iters);
    for (int thr : {256, 512}) {
        run3<0, 3>(thr, iters); run3<1, 3>(thr, iters); run3<2, 3>(thr, iters); run3<0, 1>(thr, iters); run3<2, 1>(thr, iters);
        run2<0, 0, 0, false>(thr, iters);
        run2<3, 0, 0, false>(thr, iters);
        run2<0, 8, 0, false>(thr, iters);
        run2<0, 8, 0, true>(thr, iters);
        run2<0, 8, 1, true>(thr, iters);
        run2<0, 8, 2, true>(thr, iters);
        run2<3, 8, 0, true>(thr, iters);
        run2<3, 8, 1, true>(thr, iters);
        run2<3, 8, 2, true>(thr, iters);
        run2<0, 16, 1, false>(thr, iters);
        run2<0, 4, 0, true>(thr, iters);
    }
    return 0;
}
