// K-b'': the Hybrid head over PAIR ROWS -- every pair brings its own latents -- with layer 1 inside the kernel (gfx950, fp32 MFMA).
//
// Replaces, per pair b of a batch, the reference's
//   gmf = zu[b] * zi[b];  m = relu(W3 relu(W0 [zu[b] | zi[b]] + b0) + b3);  logit = w_o [gmf | m] + b_o
//                                                   src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:167-184
// as called by HybridAutoRecNCF.forward (:197-208).  Nothing can be factorised here (a pair's rows are used once), so all
// 49,472 FLOP per pair are executed: until round 3 layer 1 ran as two GEMMs that wrote P, Q [B,128] to HBM for the pair kernel to
// read back (1.01 ms per 2^20 pairs, 0.33 of the fp32 MFMA peak).  Here a wave keeps 32 pairs on the MFMA column through BOTH
// layers: X1^T[128 x 32] = W0 X0^T accumulates in four 32x32 accumulators whose registers are, one for one, the B operand of
// layer 2 (the same accumulator-order trick as layers 2 -> 3 of tower.hip: k = 8 q + 4 h + jj holds for both), so the only
// memory traffic is the 512 B of latents per pair.
//
// k order of layer 1.  The concatenation [zu | zi] is staged one QUARTER at a time, and a quarter holds the SAME 16 latent
// features of both sides: quarter c = [zu[16c .. 16c+16) | zi[16c .. 16c+16)].  The k index an MFMA step carries is free as long
// as A and B agree, and the W0 image is packed in that order (pack_rows_image_kernel).  With both sides of a feature in LDS at
// once the GMF product costs no loads at all: the lane that reads zu[f..f+4) for quad q' reads zi[f..f+4) at quad q' + 2.
#include "tower_shared.h"

namespace {

constexpr int RW_W1 = 4 * 16 * 256;          // W0 image [t(4)][qg(16)][lane(64)][4] floats (64 KB)
constexpr int RW_B1 = 128;                   // b0 in accumulator order [t(4)][rq(4)][h(2)][4]
constexpr int RW_SIZE = RW_W1 + RW_B1;

__global__ void pack_rows_image_kernel(const float* __restrict__ W1, const float* __restrict__ b1, float* __restrict__ img1)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= RW_SIZE) return;
    float v;
    if (idx < RW_W1) {
        const int jj = idx & 3, lane = (idx >> 2) & 63, qg = (idx >> 8) & 15, t = idx >> 12;
        const int c = qg >> 2, kq = 8 * (qg & 3) + 4 * (lane >> 5) + jj;          // position inside staged quarter c
        const int col = kq < 16 ? 16 * c + kq : 64 + 16 * c + (kq - 16);          // column of W0 [128, 64 user | 64 item]
        v = W1[(32 * t + (lane & 31)) * 128 + col];
    } else {
        const int e = idx - RW_W1;
        const int jj = e & 3, h = (e >> 2) & 1, rq = (e >> 3) & 3, t = e >> 5;
        v = b1[32 * t + jj + 8 * rq + 4 * h];
    }
    img1[idx] = v;
}

struct RowsArgs {
    const float* img;      // the tower's packed image (layer 2 = the head's second Linear, biases, predict weights)
    const float* img1;     // RW_SIZE floats, above
    const float* zu;       // [B, 64]
    const float* zi;       // [B, 64]
    int64_t B;
    float* out;
    float bp;
    int act;
    float lo, hi;
};

// one quarter (16 features of both sides) of the wave's 32 pairs: lane -> rows (lane >> 2) + 16 it, float4 chunk lane & 3
__device__ __forceinline__ void rows_load_quarter(const RowsArgs& a, int64_t base, int lane, int c, f32x4* xu, f32x4* xi)
{
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int64_t row = base + (lane >> 2) + 16 * it;
        if (row >= a.B) row = a.B - 1;                     // the ragged last group reads a valid row; its score is not stored
        const int64_t off = row * 64 + 16 * c + 4 * (lane & 3);
        xu[it] = *reinterpret_cast<const f32x4*>(a.zu + off);
        xi[it] = *reinterpret_cast<const f32x4*>(a.zi + off);
    }
}

__device__ __forceinline__ void rows_store_quarter(float* s_x, int lane, const f32x4* xu, const f32x4* xi)
{
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        float* d = s_x + ((lane >> 2) + 16 * it) * QSP + 4 * (lane & 3);
        *reinterpret_cast<f32x4*>(d) = xu[it];
        *reinterpret_cast<f32x4*>(d + 16) = xi[it];
    }
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void tower_rows_kernel(RowsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_w1 = smem;
    float* s_img = smem + RW_SIZE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    float* s_x = s_img + LDS_IMG_SIZE<false> + wave * (32 * QSP);      // [32][QSP] one quarter of this wave's pairs

    for (int i = tid; i < RW_SIZE / 4; i += 64 * WAVES) reinterpret_cast<f32x4*>(s_w1)[i] = reinterpret_cast<const f32x4*>(a.img1)[i];
    load_image_scaled<false, 64 * WAVES>(s_img, a.img, tid, Layer1Scale{1.f, 1.f});
    __syncthreads();

    const f32x4* w1 = reinterpret_cast<const f32x4*>(s_w1) + lane;
    const f32x4* w2 = reinterpret_cast<const f32x4*>(s_img + IMG_W2) + lane;
    const float* wpg = s_img + lds_img<false>(IMG_WPG) + 4 * h;
    const float* xrow = s_x + j * QSP + 4 * h;

    const int64_t ngroups = (a.B + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    int64_t grp = (int64_t)blockIdx.x * WAVES + wave;
    f32x4 xu[2], xi[2];
    if (grp < ngroups) rows_load_quarter(a, grp * 32, lane, 0, xu, xi);
    for (; grp < ngroups; grp += stride) {
        const int64_t base = grp * 32;
        const bool more = grp + stride < ngroups;
        __builtin_amdgcn_s_setprio(1);
        f32x16 acc1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc1[t] = bias16(s_w1 + RW_W1 + 32 * t, h);
        float gm = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // quarter c into LDS (every read of quarter c - 1 has been issued: LDS operations of a wave execute in order),
            // the following quarter -- of this group or the next -- on its way
            rows_store_quarter(s_x, lane, xu, xi);
            __builtin_amdgcn_wave_barrier();
            if (c < 3) rows_load_quarter(a, base, lane, c + 1, xu, xi);
            else if (more) rows_load_quarter(a, (grp + stride) * 32, lane, 0, xu, xi);
            f32x4 u4[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(xrow + 8 * q);
                f32x4 aw[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) aw[t] = w1[(t * 16 + 4 * c + q) * 64];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[t][jj], x[jj], acc1[t], 0, 0, 0);
                // GMF half of the predict layer: quads 0, 1 hold the user's features 16 c + 8 q + 4 h .. + 3, quads 2, 3 the item's
                if (q < 2) u4[q] = x;
                else {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wpg + 16 * c + 8 * (q - 2));
                    const float s = gmf_col(u4[q - 2] * w, x);
                    gm = (c == 0 && q == 2) ? s : gm + s;
                }
            }
        }
        // layer 2: the B operand is relu(acc1), register for register (k = 8 q + 4 h + jj = the feature acc1[q >> 2][4 (q & 3) + jj] holds)
        f32x16 acc2a = bias16(s_img + lds_img<false>(IMG_B2), h);
        f32x16 acc2b = bias16(s_img + lds_img<false>(IMG_B2) + 32, h);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x4 a0 = w2[q * 64], a1 = w2[(16 + q) * 64];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float b = relu1(acc1[q >> 2][4 * (q & 3) + jj]);
                acc2a = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[jj], b, acc2a, 0, 0, 0);
                acc2b = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[jj], b, acc2b, 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        float part = dot16_relu(acc2a, s_img + lds_img<false>(IMG_WPM), h, 0.f);
        part = dot16_relu(acc2b, s_img + lds_img<false>(IMG_WPM) + 32, h, part);
        part = gm + part;
        const float total = part + __shfl_xor(part, 32) + a.bp;
        const int64_t b = base + j;
        if (h == 0 && b < a.B) a.out[b] = epilogue(total, a.act, a.lo, a.hi);
    }
}

}  // namespace

size_t tower_rows_image_bytes() { return (size_t)RW_SIZE * sizeof(float); }

int tower_pack_rows_image(const float* W1, const float* b1, float* img1, hipStream_t s)
{
    hipLaunchKernelGGL(pack_rows_image_kernel, dim3((RW_SIZE + 255) / 256), dim3(256), 0, s, W1, b1, img1);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}

// zu / zi [B, 64], 16-byte aligned; img = the tower image of a (dg = 64, dm = 64, two-layer) head, img1 = tower_pack_rows_image
int tower_launch_rows(const float* img, const float* img1, const float* zu, const float* zi, int64_t B, float* out,
                      float bp, int act, float lo, float hi, hipStream_t s)
{
    if (B <= 0) return RECO_OK;
    constexpr int WAVES = 8;
    constexpr size_t lds = (size_t)(RW_SIZE + LDS_IMG_SIZE<false> + WAVES * 32 * QSP) * sizeof(float);
    static_assert(lds <= 160 * 1024, "rows kernel LDS budget");
    static bool attr[RECO_MAX_DEVICES] = {};
    if (int rc = reco_allow_lds(reinterpret_cast<const void*>(&tower_rows_kernel<WAVES>), lds, attr)) return rc;
    RowsArgs a{img, img1, zu, zi, B, out, bp, act, lo, hi};
    const int64_t nchunks = ceil_div64(B, 32 * WAVES);
    const unsigned grid = (unsigned)(nchunks < 256 ? nchunks : 256);     // persistent: one workgroup per CU
    hipLaunchKernelGGL(tower_rows_kernel<WAVES>, dim3(grid), dim3(64 * WAVES), lds, s, a);
    RECO_CHECK_LAUNCH();
    return RECO_OK;
}
