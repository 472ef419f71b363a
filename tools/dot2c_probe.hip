// Is v_dot2c_f32_bf16 with S1 = {-1, 0} / {0, -1} the exact residual x - bf16(x)?  (tower_split.hip sub_lo / sub_hi)
// Compares it with the two-step form (extract the piece with a shift / mask, v_sub_f32) over 2^28 bit patterns spread over the
// whole fp32 range, both signs, and all three splitting levels.   hipcc -O3 --offload-arch=gfx950 tools/dot2c_probe.hip -o /tmp/dot2c_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned pk_bf16(float a, float b)
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// {-1, 0} must not be written as a literal: the compiler turns 0x0000bf80 into the inline constant -1.0, which the instruction
// reads as 0xbf800000 = {0, -1} (tower_split.hip neg_lo_const)
__device__ __forceinline__ float sub_lo(float x, unsigned pk)
{
    unsigned k = 0x0000bf80u;
    asm volatile("" : "+s"(k));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, k), x, false);
}
__device__ __forceinline__ float sub_hi(float x, unsigned pk)
{
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, 0xbf800000u), x, false);
}

// counts[0] mismatches among normal residuals, counts[1] mismatches where the two-step residual is denormal or zero-vs-denormal,
// counts[2] values tested, counts[3] first mismatching input bits
__global__ void probe(unsigned long long* counts, unsigned stride)
{
    unsigned long long bad = 0, bad_den = 0, n = 0;
    const unsigned first = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long bits = first; bits < (1ull << 32); bits += (unsigned long long)stride * gridDim.x * blockDim.x) {
        const unsigned ua = (unsigned)bits, ub = ua * 2654435761u + 12345u;          // two different values per packed pair
        float xa = __uint_as_float(ua), xb = __uint_as_float(ub);
        if (!(fabsf(xa) < 3e38f) || !(fabsf(xb) < 3e38f)) continue;                   // finite inputs only (the mode's contract)
        float ya = xa, yb = xb;
#pragma unroll
        for (int level = 0; level < 2; ++level) {
            const unsigned P = pk_bf16(xa, xb);          // same pieces on both paths as long as the residuals agree
            const float ra = xa - __uint_as_float(P << 16), rb = xb - __uint_as_float(P & 0xffff0000u);
            const unsigned Q = pk_bf16(ya, yb);
            const float sa = sub_lo(ya, Q), sb = sub_hi(yb, Q);
            n += 2;
            if (__float_as_uint(ra) != __float_as_uint(sa) || __float_as_uint(rb) != __float_as_uint(sb)) {
                const bool den = (fabsf(ra) < 1.1754944e-38f && __float_as_uint(ra) != __float_as_uint(sa)) ||
                                 (fabsf(rb) < 1.1754944e-38f && __float_as_uint(rb) != __float_as_uint(sb));
                if (den) ++bad_den; else { if (!bad) atomicCAS(&counts[3], 0ull, (unsigned long long)ua | ((unsigned long long)level << 32) | (1ull << 40)); ++bad; }
            }
            xa = ra; xb = rb; ya = sa; yb = sb;
        }
    }
    atomicAdd(&counts[0], bad); atomicAdd(&counts[1], bad_den); atomicAdd(&counts[2], n);
}

int main()
{
    unsigned long long* d; unsigned long long h[4] = {0, 0, 0, 0};
    CK(hipMalloc(&d, sizeof h)); CK(hipMemset(d, 0, sizeof h));
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, d, 16u);        // every 16th bit pattern: 2^28 pairs x 2 levels
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("{\"tested\": %llu, \"mismatch_normal\": %llu, \"mismatch_denormal_residual\": %llu, \"first_bad\": \"0x%llx\"}\n", h[2], h[0], h[1], h[3]);
    return h[0] ? 1 : 0;
}
