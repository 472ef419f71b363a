// Shared declarations for the gfx950 scoring engine (internal; the public C ABI is include/reco_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/reco_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void reco_set_error(const char* fmt, ...);

#define RECO_CHECK_HIP(expr)                                                                        \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            reco_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RECO_EHIP;                                                                       \
        }                                                                                           \
    } while (0)

#define RECO_REQUIRE(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            reco_set_error(__VA_ARGS__); \
            return RECO_EINVAL;          \
        }                                \
    } while (0)

#define RECO_CHECK_LAUNCH() RECO_CHECK_HIP(hipGetLastError())

// Measurement hook (api.hip; reco_profile_* in reco_hip.h): a hipEvent pair recorded on the launch stream around ONE kernel
// launch, when the calling thread armed the hook.  `kind` is one of RECO_PROF_*.  Costs a thread-local load when disarmed.
struct ProfScope {
    hipStream_t s;
    int slot;
    ProfScope(hipStream_t stream, int kind);
    ~ProfScope();
};

// Raises a kernel's dynamic-LDS limit once per device (the attribute is per device; `done` is the caller's
// function-local table, one per kernel instantiation).
constexpr int RECO_MAX_DEVICES = 64;
static inline int reco_allow_lds(const void* kernel, size_t bytes, bool* done)
{
    int dev = 0;
    RECO_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= RECO_MAX_DEVICES || !done[dev]) {
        RECO_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        if (dev >= 0 && dev < RECO_MAX_DEVICES) done[dev] = true;
    }
    return RECO_OK;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- tower geometry (fixed by the reference's shipped configs) -------------------------------
// layer 1 (factorised into P/Q) -> H1 = 128; layer 2: 128 -> 64 (MFMA); layer 3: 64 -> 32 (MFMA, NeuMF only)
constexpr int H1 = 128;
constexpr int H2 = 64;
constexpr int H3 = 32;

// Packed weight image, in floats.  Laid out so that every MFMA A-operand fetch is one conflict-free
// ds_read_b128 at ((block * 64 + lane) * 4): see tower.hip pack_image_kernel for the index maps.
constexpr int IMG_W2 = 0;                    // [t(2)][q(16)][lane(64)][4]
constexpr int IMG_W3 = IMG_W2 + 2 * 16 * 256;  // [q'(8)][lane(64)][4]
constexpr int IMG_B2 = IMG_W3 + 8 * 256;      // [t(2)][rq(4)][h(2)][4]
constexpr int IMG_B3 = IMG_B2 + 64;           // [rq(4)][h(2)][4]
constexpr int IMG_WPM = IMG_B3 + 32;          // predict weights, MLP half, in accumulator order (32 or 64)
constexpr int IMG_WPG = IMG_WPM + 64;         // predict weights, GMF half [G]
constexpr int IMG_SIZE = IMG_WPG + 64;        // 10464 floats = 41,856 B
// The copy of the image in LDS leaves the layer-3 weights out for a tower without a third layer (the Hybrid head): 8 KB that
// let its tile kernel hold 24 users' P rows and run three waves per SIMD like the NeuMF shapes.
template <bool L3> constexpr int lds_img(int off) { return (L3 || off < IMG_W3) ? off : off - (IMG_B2 - IMG_W3); }
template <bool L3> constexpr int LDS_IMG_SIZE = lds_img<L3>(IMG_SIZE);   // 10464 / 8416 floats

// internal launchers (defined in the .hip files, called from api.hip)
struct TowerImage;  // opaque here

int launch_linear_f32(const float* A, int64_t lda, const int64_t* a_rows, int64_t a_row_limit, int64_t M,
                      const float* W, int64_t ldw, const float* bias, int64_t N, int64_t K,
                      int act, float lo, float hi, float* C, int64_t ldc, int32_t* oob,
                      void* ws, size_t ws_bytes, hipStream_t s, int flags = 0, const float* out_scale = nullptr);
size_t linear_workspace_bytes(int64_t M, int64_t N, int64_t K);

int launch_topk_rows(const float* scores, int64_t rows, int64_t N, int64_t ld, int k,
                     const uint8_t* exclude, int64_t ld_ex, int32_t* out_ids, float* out_vals,
                     void* ws, size_t ws_bytes, hipStream_t s);
int launch_topk_rows_ex(const float* scores, int64_t rows, int64_t N, int64_t ld, int k,
                        const uint8_t* exclude, int64_t ld_ex, const float* seen, int64_t ld_seen,
                        const int64_t* row_ids, int64_t src_rows, int32_t* oob, int32_t* out_ids, float* out_vals,
                        void* ws, size_t ws_bytes, hipStream_t s);
size_t topk_workspace_bytes(int64_t rows, int64_t N, int k);
int launch_topk_merge(const unsigned long long* partial, int64_t rows, int splits, int k, int32_t* out_ids, float* out_vals, hipStream_t s);

// sparse.hip
int launch_csr_linear(const int64_t* row_ptr, const int32_t* col, const float* val, int64_t M, int64_t K,
                      const int64_t* row_seg_ptr, const int32_t* seg_row, const int64_t* seg_begin, const int64_t* seg_end,
                      int64_t S, const int32_t* order, int64_t n_order, int single_segment_rows,
                      const float* Wt, int64_t ldwt, const float* bias, int64_t N, int act, float lo, float hi,
                      float* C, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t s);
size_t csr_linear_workspace_bytes(int64_t S, int64_t N);
int launch_transpose(const float* in, int64_t rows, int64_t cols, int64_t ldi, float* out, int64_t ldo, hipStream_t s);
