// Where does global_load_lds_dwordx4 put each lane's 16 bytes?  Expectation (gfx950): LDS base (M0, wave-uniform) + lane * 16.
// Each of 4 waves loads its own 1 KB (lane l of wave w fetches floats [1024 w + 4 l', +4) with l' = a permutation of l) into its own
// LDS window; the whole image is then copied out and compared on the host.   hipcc -O3 --offload-arch=gfx950 tools/lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void probe(const float* g, float* out)
{
    __shared__ __attribute__((aligned(16))) float s[1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int src = lane ^ 5;                                   // a lane permutation: the LDS position must follow the LANE, not the address
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 256 * wave + 4 * src),
                                     (__attribute__((address_space(3))) void*)(s + 256 * wave), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = s[i];
}

int main()
{
    float h[1024], o[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *dg, *dout;
    CK(hipMalloc(&dg, sizeof h)); CK(hipMalloc(&dout, sizeof o));
    CK(hipMemcpy(dg, h, sizeof h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, dg, dout);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                const float want = (float)(256 * w + 4 * (l ^ 5) + e);       // lane l's 16 bytes at window + 16 l
                if (o[256 * w + 4 * l + e] != want) { if (bad < 5) printf("wave %d lane %d elem %d: %g, expected %g\n", w, l, e, o[256 * w + 4 * l + e], want); ++bad; }
            }
    printf("{\"mismatches\": %d}\n", bad);
    return bad ? 1 : 0;
}
