// Probe: layers 2/3 of the tower with fp32 operands split into three bf16 pieces (hi + mid + lo, RNE each, residuals
// exact) and products formed on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Six of the nine cross terms are
// kept (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid): the dropped ones are <= 2^-24 relative to |w||x| each.
// Measures (a) the rate of the body beside the splitting VALU work and (b) the deviation from an fp64 host
// reference, next to the same figure for plain fp32 fmaf chains.  Build:
//   hipcc --offload-arch=gfx950 -O3 -o build/split_probe tools/split_probe.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int H1 = 128, H2 = 64, H3 = 32;
constexpr int QS = H1 + 4;
// image (in 16-byte units per lane): W2 [t 2][s 8][piece 3][lane 64], W3 [s 4][piece 3][lane 64]
constexpr int W2_U = 2 * 8 * 3 * 64, W3_U = 4 * 3 * 64;
constexpr int IMG_BYTES = (W2_U + W3_U) * 16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned pk_bf16(float a, float b)
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ float relu1(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

struct Frag3 { bf16x8 hi, mid, lo; };

// 8 fp32 values -> three bf16x8 fragments
__device__ __forceinline__ Frag3 split8(const float* x)
{
    u32x4 H, M, L;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p], b = x[2 * p + 1];
        const unsigned h = pk_bf16(a, b);
        const float ra = a - lo_f(h), rb = b - hi_f(h);
        const unsigned m = pk_bf16(ra, rb);
        const float sa = ra - lo_f(m), sb = rb - hi_f(m);
        H[p] = h; M[p] = m; L[p] = pk_bf16(sa, sb);
    }
    Frag3 f;
    f.hi = __builtin_bit_cast(bf16x8, H); f.mid = __builtin_bit_cast(bf16x8, M); f.lo = __builtin_bit_cast(bf16x8, L);
    return f;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

__device__ __forceinline__ f32x16 six(const bf16x8* w /* hi, mid, lo at stride 64 */, const Frag3& x, f32x16 acc)
{
    const bf16x8 wh = w[0], wm = w[64], wl = w[128];
    acc = MFMA(wl, x.hi, acc);     // small terms first
    acc = MFMA(wh, x.lo, acc);
    acc = MFMA(wm, x.mid, acc);
    acc = MFMA(wm, x.hi, acc);
    acc = MFMA(wh, x.mid, acc);
    acc = MFMA(wh, x.hi, acc);
    return acc;
}

struct Args {
    const void* img;
    const float* P; const float* Q; const float* b2; const float* b3; const float* wp;
    int U; float* out; int iters; int groups; unsigned long long* diag;
};

// ---- explicitly scheduled body -------------------------------------------------------------------------------
// One "slot" = one MFMA followed by ~5 VALU instructions of the NEXT k-step's operand split, fenced so the
// compiler keeps the interleave: a bf16 MFMA occupies the matrix pipe for 32 cycles, in which the wave can issue
// those VALU instructions for free.
struct Pair3 { float xa, xb, ra, rb; };          // running state of one element pair through the three chunks
struct Build { u32x4 H, M, L; };

#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void chunk0(Build& bd, Pair3& st, int pr, float a, float b)
{
    st.xa = a; st.xb = b;
    bd.H[pr] = pk_bf16(a, b);
}
__device__ __forceinline__ void chunk1(Build& bd, Pair3& st, int pr)
{
    st.ra = st.xa - lo_f(bd.H[pr]); st.rb = st.xb - hi_f(bd.H[pr]);
    bd.M[pr] = pk_bf16(st.ra, st.rb);
}
__device__ __forceinline__ void chunk2(Build& bd, Pair3& st, int pr)
{
    const float sa = st.ra - lo_f(bd.M[pr]), sb = st.rb - hi_f(bd.M[pr]);
    bd.L[pr] = pk_bf16(sa, sb);
}
__device__ __forceinline__ Frag3 finish(const Build& bd)
{
    Frag3 f;
    f.hi = __builtin_bit_cast(bf16x8, bd.H); f.mid = __builtin_bit_cast(bf16x8, bd.M); f.lo = __builtin_bit_cast(bf16x8, bd.L);
    return f;
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
    const bf16x8* a = s_w + W2_U + (s * 3) * 64 + lane;
    return W3r{a[0], a[64], a[128]};
}

// the six kept cross terms for accumulator `acc`, term index t (small terms first)
__device__ __forceinline__ f32x16 term(int t, const bf16x8& wh, const bf16x8& wm, const bf16x8& wl, const Frag3& x, f32x16 acc)
{
    switch (t) {
    case 0: return MFMA(wl, x.hi, acc);
    case 1: return MFMA(wh, x.lo, acc);
    case 2: return MFMA(wm, x.mid, acc);
    case 3: return MFMA(wm, x.hi, acc);
    case 4: return MFMA(wh, x.mid, acc);
    default: return MFMA(wh, x.hi, acc);
    }
}

__global__ __launch_bounds__(512, 2) void split_kernel(Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16x8* s_w = reinterpret_cast<bf16x8*>(smem);
    float* s_q = reinterpret_cast<float*>(smem + IMG_BYTES);
    float* s_b = s_q + 64 * QS;      // b2[64] b3[32] wp[32] in accumulator order [t][rq][h][jj]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    for (int i = tid; i < IMG_BYTES / 16; i += blockDim.x) reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(a.img)[i];
    for (int i = tid; i < 64 * 32; i += blockDim.x)
        *reinterpret_cast<f32x4*>(s_q + (i >> 5) * QS + (i & 31) * 4) = reinterpret_cast<const f32x4*>(a.Q)[i];
    if (tid < 128) {
        const int e = tid & 31, jj = e & 3, hh = (e >> 2) & 1, rq = e >> 3, t = tid >> 5;
        const int f = jj + 8 * rq + 4 * hh;
        s_b[tid] = t < 2 ? a.b2[32 * t + f] : (t == 2 ? a.b3[f] : a.wp[f]);
    }
    __syncthreads();

    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < a.iters; ++it)
    for (int u = blockIdx.x * (blockDim.x >> 6) + wave; u < a.U; u += gridDim.x * (blockDim.x >> 6)) {
        float p[64];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(a.P + (size_t)u * H1 + 16 * s + 8 * h);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(a.P + (size_t)u * H1 + 16 * s + 8 * h + 4);
            p[8 * s + 0] = v0.x; p[8 * s + 1] = v0.y; p[8 * s + 2] = v0.z; p[8 * s + 3] = v0.w;
            p[8 * s + 4] = v1.x; p[8 * s + 5] = v1.y; p[8 * s + 6] = v1.z; p[8 * s + 7] = v1.w;
        }
        for (int gg = 0; gg < a.groups; ++gg) {
            const int g = gg & 1;
            asm volatile("" ::: "memory");
            f32x16 acc0, acc1, acc3;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(s_b + (rq * 2 + h) * 4);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(s_b + 32 + (rq * 2 + h) * 4);
                const f32x4 c3 = *reinterpret_cast<const f32x4*>(s_b + 64 + (rq * 2 + h) * 4);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { acc0[4 * rq + jj] = c0[jj]; acc1[4 * rq + jj] = c1[jj]; acc3[4 * rq + jj] = c3[jj]; }
            }
            const float* qrow = s_q + (32 * g + j) * QS + 8 * h;
            // prologue: operands of k-step 0
            f32x4 q0 = *reinterpret_cast<const f32x4*>(qrow), q1 = *reinterpret_cast<const f32x4*>(qrow + 4);
            Build bd; Pair3 st[4];
            {
                const float x[8] = {relu1(p[0] + q0.x), relu1(p[1] + q0.y), relu1(p[2] + q0.z), relu1(p[3] + q0.w),
                                    relu1(p[4] + q1.x), relu1(p[5] + q1.y), relu1(p[6] + q1.z), relu1(p[7] + q1.w)};
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) { chunk0(bd, st[pr], pr, x[2 * pr], x[2 * pr + 1]); chunk1(bd, st[pr], pr); chunk2(bd, st[pr], pr); }
            }
            Frag3 f = finish(bd);
            W6 w = load_w2(s_w, 0, lane);
            q0 = *reinterpret_cast<const f32x4*>(qrow + 16); q1 = *reinterpret_cast<const f32x4*>(qrow + 20);
            FENCE();
            // layer 2, k-steps 0..6: slot i = MFMA (accumulator i & 1, term i >> 1) + chunk (i % 3) of pair (i / 3) of step s+1
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                const W6 wn = load_w2(s_w, s + 1, lane);
                f32x4 qn0 = q0, qn1 = q1;
                if (s + 2 < 8) { qn0 = *reinterpret_cast<const f32x4*>(qrow + 16 * (s + 2)); qn1 = *reinterpret_cast<const f32x4*>(qrow + 16 * (s + 2) + 4); }
                FENCE();
                const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    if (i & 1) acc1 = term(i >> 1, w.h1, w.m1, w.l1, f, acc1);
                    else acc0 = term(i >> 1, w.h0, w.m0, w.l0, f, acc0);
                    const int pr = i / 3, c = i % 3;
                    if (c == 0) chunk0(bd, st[pr], pr, relu1(p[8 * (s + 1) + 2 * pr] + qv[2 * pr]), relu1(p[8 * (s + 1) + 2 * pr + 1] + qv[2 * pr + 1]));
                    else if (c == 1) chunk1(bd, st[pr], pr);
                    else chunk2(bd, st[pr], pr);
                    FENCE();
                }
                f = finish(bd); w = wn; q0 = qn0; q1 = qn1;
            }
            // k-step 7: acc0's six terms first, then acc1's six beside the split of acc0[0..7] (layer 3, step 0)
            W3r w3 = load_w3(s_w, 0, lane);
            FENCE();
#pragma unroll
            for (int i = 0; i < 6; ++i) { acc0 = term(i, w.h0, w.m0, w.l0, f, acc0); FENCE(); }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                acc1 = term(i, w.h1, w.m1, w.l1, f, acc1);
#pragma unroll
                for (int cc = 2 * i; cc < 2 * i + 2; ++cc) {
                    const int pr = cc / 3, c = cc % 3;
                    if (c == 0) chunk0(bd, st[pr], pr, relu1(acc0[2 * pr]), relu1(acc0[2 * pr + 1]));
                    else if (c == 1) chunk1(bd, st[pr], pr);
                    else chunk2(bd, st[pr], pr);
                }
                FENCE();
            }
            f = finish(bd);
            // layer 3: k-step s3 uses registers 8 (s3 & 1) .. + 7 of acc0 (s3 < 2) / acc1
#pragma unroll
            for (int s3 = 0; s3 < 4; ++s3) {
                W3r wn3 = w3;
                if (s3 + 1 < 4) wn3 = load_w3(s_w, s3 + 1, lane);
                FENCE();
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    acc3 = term(i, w3.h, w3.m, w3.l, f, acc3);
                    if (s3 + 1 < 4) {
#pragma unroll
                        for (int cc = 2 * i; cc < 2 * i + 2; ++cc) {
                            const int pr = cc / 3, c = cc % 3, r = 8 * ((s3 + 1) & 1) + 2 * pr;
                            if (c == 0) chunk0(bd, st[pr], pr, relu1(s3 + 1 < 2 ? acc0[r] : acc1[r]), relu1(s3 + 1 < 2 ? acc0[r + 1] : acc1[r + 1]));
                            else if (c == 1) chunk1(bd, st[pr], pr);
                            else chunk2(bd, st[pr], pr);
                        }
                    }
                    FENCE();
                }
                if (s3 + 1 < 4) f = finish(bd);
                w3 = wn3;
            }
            float part = 0.f;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(s_b + 96 + (rq * 2 + h) * 4);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) part = fmaf(wv[jj], relu1(acc3[4 * rq + jj]), part);
            }
            part += __shfl_xor(part, 32);
            if (h == 0) a.out[(size_t)u * 64 + 32 * g + j] = part;
        }
    }
    if (blockIdx.x == 0 && tid == 0) { a.diag[0] = __builtin_amdgcn_s_memtime() - c0; a.diag[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

static unsigned short bf16_rne(float f)
{
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static void split3(float w, unsigned short* o)
{
    o[0] = bf16_rne(w); float r = w - bf16_f(o[0]);
    o[1] = bf16_rne(r); r -= bf16_f(o[1]);
    o[2] = bf16_rne(r);
}

int main(int argc, char** argv)
{
    const int U = argc > 1 ? atoi(argv[1]) : 256 * 8 * 32;
    const int iters = argc > 2 ? atoi(argv[2]) : 1;
    srand(1);
    auto rnd = [](float s) { return s * (2.f * rand() / RAND_MAX - 1.f); };
    std::vector<float> W2(H2 * H1), b2(H2), W3(H3 * H2), b3(H3), wp(H3), P((size_t)U * H1), Q(64 * H1);
    for (auto& v : W2) v = rnd(0.18f);
    for (auto& v : W3) v = rnd(0.25f);
    for (auto& v : b2) v = rnd(0.1f);
    for (auto& v : b3) v = rnd(0.1f);
    for (auto& v : wp) v = rnd(0.3f);
    for (auto& v : P) v = rnd(2.f);
    for (auto& v : Q) v = rnd(2.f);

    std::vector<unsigned short> img(IMG_BYTES / 2);
    for (int t = 0; t < 2; ++t) for (int s = 0; s < 8; ++s) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
        unsigned short o[3]; split3(W2[(32 * t + (l & 31)) * H1 + 16 * s + 8 * (l >> 5) + e], o);
        for (int c = 0; c < 3; ++c) img[((((t * 8 + s) * 3 + c) * 64 + l) * 8) + e] = o[c];
    }
    for (int s = 0; s < 4; ++s) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
        const int t = s >> 1, g = s & 1, hh = l >> 5;
        const int f = 32 * t + (e & 3) + 8 * (2 * g + (e >> 2)) + 4 * hh;
        unsigned short o[3]; split3(W3[(l & 31) * H2 + f], o);
        for (int c = 0; c < 3; ++c) img[(size_t)W2_U * 8 + (((s * 3 + c) * 64 + l) * 8) + e] = o[c];
    }
    void* d_img; float *dP, *dQ, *db2, *db3, *dwp, *dout;
    CK(hipMalloc(&d_img, IMG_BYTES)); CK(hipMemcpy(d_img, img.data(), IMG_BYTES, hipMemcpyHostToDevice));
    CK(hipMalloc(&dP, P.size() * 4)); CK(hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dQ, Q.size() * 4)); CK(hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&db2, 256)); CK(hipMemcpy(db2, b2.data(), 256, hipMemcpyHostToDevice));
    CK(hipMalloc(&db3, 128)); CK(hipMemcpy(db3, b3.data(), 128, hipMemcpyHostToDevice));
    CK(hipMalloc(&dwp, 128)); CK(hipMemcpy(dwp, wp.data(), 128, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, (size_t)U * 64 * 4));
    const size_t lds = IMG_BYTES + 64 * QS * 4 + 128 * 4;
    CK(hipFuncSetAttribute((const void*)split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int groups = argc > 3 ? atoi(argv[3]) : 2;
    const int threads = argc > 4 ? atoi(argv[4]) : 512;
    unsigned long long* ddiag; CK(hipMalloc(&ddiag, 16));
    Args a{d_img, dP, dQ, db2, db3, dwp, U, dout, iters, groups, ddiag};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(split_kernel, dim3(256), dim3(threads), lds, 0, a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double pairs = (double)U * 32 * groups * iters;
        printf("split bf16x3: %.3f ms  %.3f G pairs/s (layers 2+3 only; exact-f32 tile kernel: 6.1 G pairs/s)\n", ms, pairs / ms / 1e6);
    }
    unsigned long long dg[2]; CK(hipMemcpy(dg, ddiag, 16, hipMemcpyDeviceToHost));
    { const double mhz = (double)dg[0] / ((double)dg[1] / 100.0); const double bodies = (double)(U / (256 * (threads / 64))) * groups * iters;
      printf("shader clock %.0f MHz; %.0f cycles per 32-pair body per wave (120 MFMAs -> %.1f cycles per MFMA slot)\n", mhz, dg[0] / bodies, dg[0] / bodies / 120); }
    std::vector<float> out((size_t)U * 64);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    // fp64 reference and plain-fp32 fmaf reference on a sample of users
    double max_split = 0, max_f32 = 0, max_ref = 0;
    for (int u = 0; u < U; u += U / 64 + 1) for (int i = 0; i < 64; ++i) {
        double x1[H1], x2[H2], x3[H3]; float y1[H1], y2[H2], y3[H3];
        for (int f = 0; f < H1; ++f) { float v = P[(size_t)u * H1 + f] + Q[i * H1 + f]; v = v > 0 ? v : 0; x1[f] = v; y1[f] = v; }
        for (int o = 0; o < H2; ++o) { double s = b2[o]; float t = b2[o]; for (int f = 0; f < H1; ++f) { s += (double)W2[o * H1 + f] * x1[f]; t = fmaf(W2[o * H1 + f], y1[f], t); } x2[o] = s > 0 ? s : 0; y2[o] = t > 0 ? t : 0; }
        for (int o = 0; o < H3; ++o) { double s = b3[o]; float t = b3[o]; for (int f = 0; f < H2; ++f) { s += (double)W3[o * H2 + f] * x2[f]; t = fmaf(W3[o * H2 + f], y2[f], t); } x3[o] = s > 0 ? s : 0; y3[o] = t > 0 ? t : 0; }
        double s = 0; float t = 0; for (int f = 0; f < H3; ++f) { s += (double)wp[f] * x3[f]; t = fmaf(wp[f], y3[f], t); }
        max_split = fmax(max_split, fabs(out[(size_t)u * 64 + i] - s)); max_f32 = fmax(max_f32, fabs(t - s)); max_ref = fmax(max_ref, fabs(s));
    }
    printf("max |score| %.3f   max abs dev vs fp64:  split %.3e   plain fp32 %.3e\n", max_ref, max_split, max_f32);
    return 0;
}
