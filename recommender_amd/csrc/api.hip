// extern "C" surface of libreco_hip.so (declared in include/reco_hip.h): argument checking,
// the packed tower handle, workspace carving and kernel sequencing.  No torch types, no hidden
// allocation outside reco_tower_create, no synchronisation outside create / status.
#include "reco_common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

// launchers defined in tower.hip
int tower_pack_image(const float* W2, const float* b2, const float* W3, const float* b3, const float* wp,
                     int G, int has_l3, float* img, hipStream_t s);
int tower_launch_tile(int G, int has_l3, const float* img, const float* P, const float* Q,
                      const float* EUg, const float* EIg, const int64_t* users, const int64_t* cands,
                      int64_t user_base, int64_t n_users, int64_t n_items, int64_t U, int64_t N,
                      float* out, int64_t ld_out, float bp, int act, float lo, float hi, int32_t* oob,
                      unsigned int* next_tile, const float* l1, int reserve_cus, hipStream_t s);
int tower_launch_l1norms(const float* W1, const float* b1, int dm, float* norms, hipStream_t s);
int tower_launch_l1scale(const float* norms, const float* u_rows, int64_t u_elems, const float* i_rows, int64_t i_elems,
                         unsigned int* scratch, float* out, void* clear, size_t clear_bytes, hipStream_t s);
int tower_launch_scale_compare(const float* stored, const float* fresh, int32_t* stale, hipStream_t s);
int tower_launch_pairs(int G, int has_l3, const float* img, const float* P, const float* Q,
                       const float* EUg, const float* EIg, const int64_t* user, const int64_t* item,
                       int64_t n_users, int64_t n_items, int64_t B, float* out, float bp, int act,
                       float lo, float hi, int32_t* oob, const float* l1, hipStream_t s);
// launchers defined in tower_split.hip (RECO_PREC_SPLIT_BF16X3)
size_t tower_split_image_bytes(int has_l3);
int tower_pack_split_image(const float* W2, const float* W3, int has_l3, const float* img, void* out, hipStream_t s);
int tower_launch_tile_split(int G, int has_l3, const void* img, const float* P, const float* Q,
                            const float* EUg, const float* EIg, const int64_t* users, const int64_t* cands,
                            int64_t user_base, int64_t n_users, int64_t n_items, int64_t U, int64_t N,
                            float* out, int64_t ld_out, float bp, int act, float lo, float hi, int32_t* oob,
                            unsigned int* next_tile, int reserve_cus, hipStream_t s);
int tower_launch_pairs_split(int G, int has_l3, const void* img, const float* P, const float* Q,
                             const float* EUg, const float* EIg, const int64_t* user, const int64_t* item,
                             int64_t n_users, int64_t n_items, int64_t B, float* out, float bp, int act,
                             float lo, float hi, int32_t* oob, hipStream_t s);
// launchers defined in tower_rows.hip (the Hybrid head over pair rows, layer 1 inside the kernel)
size_t tower_rows_image_bytes();
int tower_pack_rows_image(const float* W1, const float* b1, float* img1, hipStream_t s);
int tower_launch_rows(const float* img, const float* img1, const float* zu, const float* zi, int64_t B, float* out,
                      float bp, int act, float lo, float hi, hipStream_t s);
int tower_tile_users_per_block(int G);
size_t tower_topk_counter_bytes(int G, int64_t U);
int tower_launch_tile_topk(int G, int has_l3, const float* img, const float* P, const float* Q,
                           const float* EUg, const float* EIg, const int64_t* users, const int64_t* cands,
                           int64_t user_base, int64_t n_users, int64_t n_items, int64_t U, int64_t N,
                           float bp, int act, float lo, float hi, int32_t* oob,
                           int k, int v_max, unsigned int* counters, unsigned long long* partial,
                           const float* l1, int reserve_cus, hipStream_t s);
int gmf_launch_pairs(const float* EUg, const float* EIg, const float* wp, const int64_t* user,
                     const int64_t* item, int64_t n_users, int64_t n_items, int64_t B, int G, float* out,
                     float bp, int act, float lo, float hi, int32_t* oob, hipStream_t s);
int gmf_launch_tile(const float* EUg, const float* EIg, const float* wp, const int64_t* users,
                    const int64_t* cands, int64_t user_base, int64_t n_users, int64_t n_items, int64_t U,
                    int64_t N, int G, float* out, int64_t ld_out, float bp, int act, float lo, float hi,
                    int32_t* oob, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void reco_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* reco_last_error(void) { return g_err; }
extern "C" int reco_abi_version(void) { return 5; }

// ---------------------------------------------------------------------------------------------
// measurement hook: hipEvent pairs around kernel launches, armed per host thread (no state shared between threads)
// ---------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>
namespace {
struct ProfState {
    std::vector<hipEvent_t> ev;   // 2 per launch
    std::vector<int> kind;        // per launch
    int used = 0;
    unsigned mask = ~0u;          // kinds that are bracketed (bit k = RECO_PROF_* value k)
};
thread_local ProfState t_prof;
}  // namespace

extern "C" int reco_profile_enable_kinds(int max_launches, unsigned kind_mask);
extern "C" int reco_profile_enable(int max_launches) { return reco_profile_enable_kinds(max_launches, ~0u); }

extern "C" int reco_profile_enable_kinds(int max_launches, unsigned kind_mask)
{
    ProfState& p = t_prof;
    p.mask = kind_mask;
    for (hipEvent_t e : p.ev) (void)hipEventDestroy(e);
    p.ev.clear();
    p.kind.clear();
    p.used = 0;
    for (int i = 0; i < 2 * max_launches; ++i) {
        hipEvent_t e;
        RECO_CHECK_HIP(hipEventCreate(&e));
        p.ev.push_back(e);
    }
    p.kind.assign(max_launches > 0 ? max_launches : 0, 0);
    return RECO_OK;
}

extern "C" int reco_profile_collect_kinds(float* ms_out, int32_t* kind_out, int capacity, int* n_out)
{
    RECO_REQUIRE(ms_out && n_out, "profile_collect: null pointer");
    ProfState& p = t_prof;
    int n = 0;
    for (int i = 0; i < p.used && n < capacity; ++i, ++n) {
        RECO_CHECK_HIP(hipEventSynchronize(p.ev[2 * i + 1]));
        RECO_CHECK_HIP(hipEventElapsedTime(&ms_out[n], p.ev[2 * i], p.ev[2 * i + 1]));
        if (kind_out) kind_out[n] = p.kind[i];
    }
    *n_out = n;
    p.used = 0;
    return RECO_OK;
}

// the tower kernels' launches only (the form bench.py's headline region has used since round 1)
extern "C" int reco_profile_collect(float* ms_out, int capacity, int* n_out)
{
    RECO_REQUIRE(ms_out && n_out, "profile_collect: null pointer");
    ProfState& p = t_prof;
    int n = 0;
    for (int i = 0; i < p.used && n < capacity; ++i) {
        if (p.kind[i] != RECO_PROF_TOWER) continue;
        RECO_CHECK_HIP(hipEventSynchronize(p.ev[2 * i + 1]));
        RECO_CHECK_HIP(hipEventElapsedTime(&ms_out[n], p.ev[2 * i], p.ev[2 * i + 1]));
        ++n;
    }
    *n_out = n;
    p.used = 0;
    return RECO_OK;
}

ProfScope::ProfScope(hipStream_t st, int kind) : s(st), slot(-1)
{
    ProfState& p = t_prof;
    if ((size_t)(2 * p.used + 1) < p.ev.size() && ((p.mask >> kind) & 1u)) {
        slot = p.used++;
        p.kind[slot] = kind;
        (void)hipEventRecord(p.ev[2 * slot], s);
    }
}

ProfScope::~ProfScope()
{
    if (slot < 0) return;
    ProfState& p = t_prof;
    if ((size_t)(2 * slot + 1) < p.ev.size()) (void)hipEventRecord(p.ev[2 * slot + 1], s);
}

// ---------------------------------------------------------------------------------------------
// Linear / AutoRec
// ---------------------------------------------------------------------------------------------
extern "C" int reco_linear_f32(const float* A, int64_t lda, const int64_t* a_rows, int64_t a_row_limit, int64_t M,
                               const float* W, int64_t ldw, const float* bias, int64_t N, int64_t K,
                               int act, float lo, float hi, float* C, int64_t ldc, int flags,
                               void* ws, size_t ws_bytes, void* stream)
{
    RECO_REQUIRE(act >= RECO_ACT_NONE && act <= RECO_ACT_RATING, "linear: unknown activation %d", act);
    RECO_REQUIRE((flags & ~7) == 0, "linear: unknown flags %d", flags);
    return launch_linear_f32(A, lda, a_rows, a_row_limit, M, W, ldw, bias, N, K, act, lo, hi, C, ldc, nullptr,
                             ws, ws_bytes, (hipStream_t)stream, flags);
}

extern "C" size_t reco_linear_workspace_bytes(int64_t M, int64_t N, int64_t K) { return linear_workspace_bytes(M, N, K); }

extern "C" size_t reco_autorec_workspace_bytes(int64_t B, int64_t n_in, int64_t hidden)
{
    const size_t a = linear_workspace_bytes(B, hidden, n_in), b = linear_workspace_bytes(B, n_in, hidden);
    return a > b ? a : b;
}

extern "C" int reco_autorec_forward(const float* X, int64_t B, const float* We, const float* be,
                                    const float* Wd, const float* bd, int64_t n_in, int64_t hidden,
                                    int act, float lo, float hi, int flags, float* hidden_ws, float* out,
                                    void* ws, size_t ws_bytes, void* stream)
{
    if (B <= 0) return RECO_OK;
    RECO_REQUIRE((flags & ~RECO_LINEAR_SPLIT_BF16X3) == 0, "autorec_forward: only RECO_LINEAR_SPLIT_BF16X3 is a valid flag, got %d", flags);
    RECO_REQUIRE(X && We && be && Wd && bd && hidden_ws && out, "autorec_forward: null pointer");
    RECO_REQUIRE(n_in > 0 && hidden > 0, "autorec_forward: bad dimensions");
    RECO_REQUIRE(act == RECO_ACT_NONE || act == RECO_ACT_CLAMP, "autorec_forward: act must be NONE or CLAMP");
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_linear_f32(X, n_in, nullptr, 0, B, We, n_in, be, hidden, n_in, RECO_ACT_SIGMOID, 0.f, 0.f,
                               hidden_ws, hidden, nullptr, ws, ws_bytes, s, flags);
    if (rc) return rc;
    return launch_linear_f32(hidden_ws, hidden, nullptr, 0, B, Wd, hidden, bd, n_in, hidden, act, lo, hi, out, n_in,
                             nullptr, ws, ws_bytes, s, flags);
}

extern "C" size_t reco_csr_linear_workspace_bytes(int64_t S, int64_t N) { return csr_linear_workspace_bytes(S, N); }

extern "C" int reco_csr_linear_f32(const int64_t* row_ptr, const int32_t* col, const float* val, int64_t M, int64_t K,
                                   const int64_t* row_seg_ptr, const int32_t* seg_row, const int64_t* seg_begin,
                                   const int64_t* seg_end, int64_t S, const int32_t* order, int64_t n_order,
                                   int single_segment_rows, const float* Wt, int64_t ldwt, const float* bias, int64_t N,
                                   int act, float lo, float hi, float* C, int64_t ldc, void* ws, size_t ws_bytes, void* stream)
{
    RECO_REQUIRE(act >= RECO_ACT_NONE && act <= RECO_ACT_RATING, "csr_linear: unknown activation %d", act);
    return launch_csr_linear(row_ptr, col, val, M, K, row_seg_ptr, seg_row, seg_begin, seg_end, S, order, n_order,
                             single_segment_rows, Wt, ldwt, bias, N, act, lo, hi, C, ldc, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int reco_transpose_f32(const float* in, int64_t rows, int64_t cols, int64_t ld_in, float* out, int64_t ld_out, void* stream)
{
    return launch_transpose(in, rows, cols, ld_in, out, ld_out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// top-k
// ---------------------------------------------------------------------------------------------
extern "C" size_t reco_topk_workspace_bytes(int64_t rows, int64_t N, int k) { return topk_workspace_bytes(rows, N, k); }

extern "C" int reco_topk_rows(const float* scores, int64_t rows, int64_t N, int64_t ld, int k,
                              const uint8_t* exclude, int64_t ld_ex, int32_t* out_ids, float* out_vals,
                              void* ws, size_t ws_bytes, void* stream)
{
    RECO_REQUIRE(ld >= N, "topk: ld smaller than N");
    RECO_REQUIRE(!exclude || ld_ex >= N, "topk: ld_ex smaller than N");
    return launch_topk_rows(scores, rows, N, ld, k, exclude, ld_ex, out_ids, out_vals, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int reco_topk_rows_seen(const float* scores, int64_t src_rows, int64_t N, int64_t ld, int k,
                                   const int64_t* row_ids, int64_t rows, const float* seen, int64_t ld_seen,
                                   int32_t* out_ids, float* out_vals, int32_t* oob, void* ws, size_t ws_bytes, void* stream)
{
    RECO_REQUIRE(ld >= N, "topk_rows_seen: ld smaller than N");
    RECO_REQUIRE(src_rows >= 0 && rows >= 0, "topk_rows_seen: negative row count");
    return launch_topk_rows_ex(scores, row_ids ? rows : src_rows, N, ld, k, nullptr, 0, seen, ld_seen, row_ids, src_rows, oob,
                               out_ids, out_vals, ws, ws_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// tower handle
// ---------------------------------------------------------------------------------------------
struct reco_tower {
    reco_tower_desc_t d;
    bool mfma;        // false: GMF-only model (no MLP branch)
    int G;
    int has_l3;
    float bp;
    float* img;       // packed weight image [IMG_SIZE]
    void* img_split;  // bf16x3 image (RECO_PREC_SPLIT_BF16X3 only)
    float* img1;      // layer-1 image of the pair-row kernel (tower_rows.hip): exact Hybrid head shape (dg = dm = 64, two layers) only
    float* P;         // [n_users,128] layer-1 user projection (+ b1)
    float* Q;         // [n_items,128] layer-1 item projection
    float* l1;        // device words {s, 1/s}: the power of two layer 1 is scaled by (tower_shared.h), fixed by the handle's tables;
                      // null in split mode and for handles without tables (then per call, resolve_l1)
    float* l1n;       // [3 H1] layer-1 weight norms behind that scale (tower.hip l1norms_kernel); null in split mode and without an MLP branch
    int device;
    int reserve_cus;  // CUs the persistent tile kernels leave without a workgroup (reco_tower_set_reserved_cus); 0 = none
};

// Layer-1 scale words {s, 1/s} (tower_shared.h).  Inside a 256-byte block: floats at words kL1Word, kL1Word + 1, three scratch
// words (two maxima, one ticket) behind them.  The handle owns such a block when it has tables; calls over row blocks handed in per call
// compute theirs into the head of the call workspace (word 0 of that head is the tile work-queue counter).
static const int kL1Word = 16, kL1ScratchWord = 20, kL1CheckScratchWord = 24, kL1CheckWord = 28;

static int tower_project(const reco_tower* t, const float* u_rows, const int64_t* u_idx, int64_t u_limit, int64_t U,
                         const float* i_rows, const int64_t* i_idx, int64_t i_limit, int64_t N,
                         float* P, float* Q, int32_t* oob, const float* l1, hipStream_t s)
{
    const int dm = t->d.dm;
    if (U > 0) {
        int rc = launch_linear_f32(u_rows, dm, u_idx, u_limit, U, t->d.W[0], 2 * dm, t->d.b[0], H1, dm, RECO_ACT_NONE,
                                   0.f, 0.f, P, H1, oob, nullptr, 0, s, 0, l1);
        if (rc) return rc;
    }
    if (N > 0) {
        int rc = launch_linear_f32(i_rows, dm, i_idx, i_limit, N, t->d.W[0] + dm, 2 * dm, nullptr, H1, dm,
                                   RECO_ACT_NONE, 0.f, 0.f, Q, H1, oob, nullptr, 0, s, 0, l1);
        if (rc) return rc;
    }
    return RECO_OK;
}

// the handle's packed weights live on the device that was current in reco_tower_create
static int check_device(const reco_tower* t, const char* who)
{
    int dev = -1;
    RECO_CHECK_HIP(hipGetDevice(&dev));
    RECO_REQUIRE(dev == t->device, "%s: handle belongs to device %d but device %d is current", who, t->device, dev);
    return RECO_OK;
}

static int run_pairs(const reco_tower* t, const float* P, const float* Q, const float* EUg, const float* EIg,
                     const int64_t* user, const int64_t* item, int64_t n_users, int64_t n_items, int64_t B, float* out,
                     int32_t* oob, const float* l1, hipStream_t s)
{
    const reco_tower_desc_t& d = t->d;
    if (t->img_split)
        return tower_launch_pairs_split(t->G, t->has_l3, t->img_split, P, Q, EUg, EIg, user, item, n_users, n_items, B,
                                        out, t->bp, d.act, d.lo, d.hi, oob, s);
    return tower_launch_pairs(t->G, t->has_l3, t->img, P, Q, EUg, EIg, user, item, n_users, n_items, B, out, t->bp,
                              d.act, d.lo, d.hi, oob, l1, s);
}

extern "C" void reco_tower_destroy(reco_tower_t* t)
{
    if (!t) return;
    (void)hipFree(t->img);
    (void)hipFree(t->img_split);
    (void)hipFree(t->img1);
    (void)hipFree(t->P);
    (void)hipFree(t->Q);
    (void)hipFree(t->l1);
    (void)hipFree(t->l1n);
    delete t;
}

static int scratch_lookup(int device);   // no-alloc entry point's static scratch (end of this file)

extern "C" int reco_tower_create(const reco_tower_desc_t* d, void* stream, reco_tower_t** out)
{
    RECO_REQUIRE(d && out, "tower_create: null argument");
    *out = nullptr;
    RECO_REQUIRE(d->dg >= 0 && d->dm >= 0 && (d->dg > 0 || d->dm > 0), "tower_create: model has neither GMF nor MLP branch");
    RECO_REQUIRE(d->dg % 4 == 0 && d->dm % 4 == 0, "tower_create: dg=%d / dm=%d must be multiples of 4", d->dg, d->dm);
    RECO_REQUIRE(d->n_users >= 0 && d->n_items >= 0, "tower_create: negative table size");
    RECO_REQUIRE(d->n_users < (1ll << 31) && d->n_items < (1ll << 31), "tower_create: tables of 2^31 rows or more are not supported");
    RECO_REQUIRE(d->w_predict && d->b_predict, "tower_create: predict layer missing");
    RECO_REQUIRE(d->act == RECO_ACT_NONE || d->act == RECO_ACT_RATING, "tower_create: act must be NONE or RATING");
    RECO_REQUIRE(d->precision == RECO_PREC_F32 || d->precision == RECO_PREC_SPLIT_BF16X3,
                 "tower_create: unknown precision %d", d->precision);
    const bool tables = d->n_users > 0 && d->n_items > 0;
    if (tables && d->dg > 0) RECO_REQUIRE(d->user_gmf && d->item_gmf, "tower_create: GMF tables missing");
    if (tables && d->dm > 0) RECO_REQUIRE(d->user_mlp && d->item_mlp, "tower_create: MLP tables missing");
    RECO_REQUIRE((uintptr_t)d->user_gmf % 16 == 0 && (uintptr_t)d->item_gmf % 16 == 0 && (uintptr_t)d->user_mlp % 16 == 0 &&
                 (uintptr_t)d->item_mlp % 16 == 0, "tower_create: embedding tables must be 16-byte aligned (rows are read as float4)");

    hipStream_t s = (hipStream_t)stream;
    int device = -1;
    RECO_CHECK_HIP(hipGetDevice(&device));
    if (int rc_s = scratch_lookup(device)) return rc_s;
    reco_tower* t = new (std::nothrow) reco_tower();
    if (!t) { reco_set_error("tower_create: out of host memory"); return RECO_ENOMEM; }
    t->d = *d;
    t->mfma = d->dm > 0;
    t->G = d->dg;
    t->has_l3 = d->n_layers == 3;
    t->device = device;

    if (t->mfma) {
        const bool neumf = d->n_layers == 3 && d->layer_out[0] == H1 && d->layer_out[1] == H2 && d->layer_out[2] == H3 &&
                           (d->dg == 32 || d->dg == 0);
        const bool hybrid = d->n_layers == 2 && d->layer_out[0] == H1 && d->layer_out[1] == H2 && d->dg == 64;
        if (!neumf && !hybrid) {
            reco_set_error("tower_create: unsupported tower (dg=%d dm=%d layers=%d widths %d/%d/%d); the MFMA path covers "
                           "NeuMF/MLP factor 32 x 3 layers (2*dm->128->64->32, dg in {0,32}) and the Hybrid head (2*dm->128->64, dg=64)",
                           d->dg, d->dm, d->n_layers, d->layer_out[0], d->layer_out[1], d->layer_out[2]);
            delete t;
            return RECO_EINVAL;
        }
        for (int l = 0; l < d->n_layers; ++l)
            if (!d->W[l] || !d->b[l]) { reco_set_error("tower_create: layer %d weights missing", l); delete t; return RECO_EINVAL; }
    }

    int rc = RECO_OK;
    hipError_t e = hipSuccess;
    if (t->mfma) e = hipMalloc(&t->img, IMG_SIZE * sizeof(float));
    const bool split = t->mfma && d->precision == RECO_PREC_SPLIT_BF16X3;
    if (e == hipSuccess && split) e = hipMalloc(&t->img_split, tower_split_image_bytes(t->has_l3));
    const bool rows_kernel = t->mfma && !split && d->n_layers == 2 && d->dg == 64 && d->dm == 64;
    if (e == hipSuccess && rows_kernel) e = hipMalloc(&t->img1, tower_rows_image_bytes());
    if (e == hipSuccess && t->mfma && tables) e = hipMalloc(&t->P, (size_t)d->n_users * H1 * sizeof(float));
    if (e == hipSuccess && t->mfma && tables) e = hipMalloc(&t->Q, (size_t)d->n_items * H1 * sizeof(float));
    if (e == hipSuccess && t->mfma && tables && !split) e = hipMalloc(&t->l1, 256);
    if (e == hipSuccess && t->mfma && !split) e = hipMalloc(&t->l1n, 3 * H1 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(&t->bp, d->b_predict, sizeof(float), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) {
        reco_set_error("tower_create: %s", hipGetErrorString(e));
        reco_tower_destroy(t);
        return e == hipErrorOutOfMemory ? RECO_ENOMEM : RECO_EHIP;
    }
    if (t->mfma) {
        rc = tower_pack_image(d->W[1], d->b[1], t->has_l3 ? d->W[2] : nullptr, t->has_l3 ? d->b[2] : nullptr,
                              d->w_predict, t->G, t->has_l3, t->img, s);
        if (!rc && t->img1) rc = tower_pack_rows_image(d->W[0], d->b[0], t->img1, s);
        if (!rc && split)
            rc = tower_pack_split_image(d->W[1], t->has_l3 ? d->W[2] : nullptr, t->has_l3, t->img, t->img_split, s);
        if (!rc && t->l1n) rc = tower_launch_l1norms(d->W[0], d->b[0], d->dm, t->l1n, s);
        if (!rc && tables && t->l1)       // the layer-1 scale of every pair these tables can form, before anything is projected
            rc = tower_launch_l1scale(t->l1n, d->user_mlp, d->n_users * d->dm, d->item_mlp, d->n_items * d->dm,
                                      reinterpret_cast<unsigned int*>(t->l1) + kL1ScratchWord, t->l1,
                                      reinterpret_cast<unsigned int*>(t->l1) + kL1ScratchWord, 3 * sizeof(unsigned int), s);
        if (!rc && tables)
            rc = tower_project(t, d->user_mlp, nullptr, 0, d->n_users, d->item_mlp, nullptr, 0, d->n_items, t->P, t->Q, nullptr, t->l1, s);
    }
    if (!rc) {
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) { reco_set_error("tower_create: %s", hipGetErrorString(e)); rc = RECO_EHIP; }
    }
    if (rc) { reco_tower_destroy(t); return rc; }
    *out = t;
    return RECO_OK;
}

// Debug aid for the one way a handle can go wrong SILENTLY: the exact kernels clamp layer 1 at 1/s, s fixed at creation from a bound
// over the tables; a write through param.data (no version counter moves, so no caller-side cache notices) that makes the embeddings
// larger would saturate activations instead of merely using stale projections.  Recomputes the bound over the tables' CURRENT
// contents (one pass over the MLP tables) and raises *stale when the handle's scale no longer covers it.
extern "C" int reco_tower_check_scale(reco_tower_t* t, int32_t* stale, void* stream)
{
    RECO_REQUIRE(t && stale, "check_scale: null argument");
    if (int rc_dev = check_device(t, "check_scale")) return rc_dev;
    if (!t->l1) return RECO_OK;                 // GMF-only, split mode, or a handle without tables: nothing is clamped against a stored scale
    hipStream_t s = (hipStream_t)stream;
    const reco_tower_desc_t& d = t->d;
    unsigned int* blk = reinterpret_cast<unsigned int*>(t->l1);
    float* fresh = reinterpret_cast<float*>(blk + kL1CheckWord);
    if (int rc = tower_launch_l1scale(t->l1n, d.user_mlp, d.n_users * d.dm, d.item_mlp, d.n_items * d.dm,
                                      blk + kL1CheckScratchWord, fresh, blk + kL1CheckScratchWord, 3 * sizeof(unsigned int), s)) return rc;
    return tower_launch_scale_compare(t->l1 + 0, fresh, stale, s);
}

extern "C" int reco_tower_set_reserved_cus(reco_tower_t* t, int cus)
{
    RECO_REQUIRE(t, "set_reserved_cus: null handle");
    RECO_REQUIRE(cus >= 0 && cus <= 128, "set_reserved_cus: %d is outside [0, 128]", cus);
    t->reserve_cus = cus;
    return RECO_OK;
}

extern "C" int reco_tower_score_pairs(reco_tower_t* t, const int64_t* user, const int64_t* item, int64_t B,
                                      float* out, int32_t* oob, void* stream)
{
    RECO_REQUIRE(t, "score_pairs: null handle");
    if (int rc_dev = check_device(t, "score_pairs")) return rc_dev;
    if (B <= 0) return RECO_OK;
    RECO_REQUIRE(user && item && out, "score_pairs: null pointer");
    RECO_REQUIRE(t->d.n_users > 0 && t->d.n_items > 0, "score_pairs: handle was created without embedding tables");
    hipStream_t s = (hipStream_t)stream;
    const reco_tower_desc_t& d = t->d;
    if (!t->mfma)
        return gmf_launch_pairs(d.user_gmf, d.item_gmf, d.w_predict, user, item, d.n_users, d.n_items, B, d.dg, out,
                                t->bp, d.act, d.lo, d.hi, oob, s);
    ProfScope prof(s, RECO_PROF_TOWER);
    return run_pairs(t, t->P, t->Q, d.user_gmf, d.item_gmf, user, item, d.n_users, d.n_items, B, out, oob, t->l1, s);
}

// the scale for a call: the handle's own, or -- a handle without tables -- computed over the row blocks of this call into `head`.
// *head_cleared: the launch zeroed the whole head first, the tile work-queue counter (word 0) included, so the first tile launch
// of the call needs no memset of its own.
static int resolve_l1(const reco_tower* t, const float* u_rows, int64_t U, const float* i_rows, int64_t N, unsigned int* head,
                      const float** l1, bool* head_cleared, hipStream_t s)
{
    *l1 = nullptr;
    *head_cleared = false;
    if (!t->mfma || t->img_split) return RECO_OK;        // GMF-only models have no layer 1; the split mode runs unscaled
    if (t->l1) { *l1 = t->l1; return RECO_OK; }
    RECO_REQUIRE(head, "internal: no workspace head for the layer-1 scale");
    float* out = reinterpret_cast<float*>(head + kL1Word);
    *l1 = out;
    *head_cleared = true;
    // the whole 256-byte head in one fill (a length that is not a multiple of 16 bytes costs the runtime a second fill kernel)
    return tower_launch_l1scale(t->l1n, u_rows, U * t->d.dm, i_rows, N * t->d.dm, head + kL1ScratchWord, out, head, 256, s);
}

extern "C" size_t reco_tower_rows_workspace_bytes(int64_t B)
{
    return B > 0 ? 2 * align_up((size_t)B * H1 * sizeof(float), 256) : 0;
}

// the pair-row kernel with layer 1 inside (tower_rows.hip): exact Hybrid head whose GMF and MLP rows are the same latents
static bool rows_fused(const reco_tower* t, const float* u_gmf, const float* i_gmf, const float* u_mlp, const float* i_mlp)
{
    return t->img1 && u_gmf == u_mlp && i_gmf == i_mlp && u_mlp && i_mlp;
}

extern "C" size_t reco_tower_rows_workspace_bytes_for(const reco_tower_t* t, int gmf_rows_are_mlp_rows, int64_t B)
{
    if (t && t->img1 && gmf_rows_are_mlp_rows) return 0;
    return reco_tower_rows_workspace_bytes(B);
}

extern "C" int reco_tower_score_rows(reco_tower_t* t, const float* u_gmf, const float* i_gmf,
                                     const float* u_mlp, const float* i_mlp, int64_t B, float* out,
                                     void* ws, size_t ws_bytes, void* stream)
{
    RECO_REQUIRE(t, "score_rows: null handle");
    if (int rc_dev = check_device(t, "score_rows")) return rc_dev;
    if (B <= 0) return RECO_OK;
    RECO_REQUIRE(out, "score_rows: null output");
    hipStream_t s = (hipStream_t)stream;
    const reco_tower_desc_t& d = t->d;
    if (d.dg > 0) RECO_REQUIRE(u_gmf && i_gmf, "score_rows: GMF rows missing");
    RECO_REQUIRE((uintptr_t)u_gmf % 16 == 0 && (uintptr_t)i_gmf % 16 == 0 && (uintptr_t)u_mlp % 16 == 0 && (uintptr_t)i_mlp % 16 == 0,
                 "score_rows: row blocks must be 16-byte aligned (rows are read as float4)");
    if (!t->mfma)
        return gmf_launch_pairs(u_gmf, i_gmf, d.w_predict, nullptr, nullptr, B, B, B, d.dg, out, t->bp, d.act, d.lo,
                                d.hi, nullptr, s);
    RECO_REQUIRE(u_mlp && i_mlp, "score_rows: MLP rows missing");
    if (rows_fused(t, u_gmf, i_gmf, u_mlp, i_mlp)) {      // no projections, no workspace: the latents are the only traffic
        ProfScope prof(s, RECO_PROF_TOWER);
        return tower_launch_rows(t->img, t->img1, u_mlp, i_mlp, B, out, t->bp, d.act, d.lo, d.hi, s);
    }
    const size_t half = align_up((size_t)B * H1 * sizeof(float), 256);
    if (!ws || ws_bytes < 2 * half) { reco_set_error("score_rows: workspace too small (%zu < %zu)", ws_bytes, 2 * half); return RECO_ENOMEM; }
    float* P = static_cast<float*>(ws);
    float* Q = reinterpret_cast<float*>(static_cast<char*>(ws) + half);
    // These rows are not the handle's tables and each is used once: a bound over them would be a pass of its own (measured:
    // 1M Hybrid rows 1.00 -> 1.18 ms), so the projections stay unscaled and the pair kernel takes its add + max form.
    int rc = tower_project(t, u_mlp, nullptr, 0, B, i_mlp, nullptr, 0, B, P, Q, nullptr, nullptr, s);
    if (rc) return rc;
    ProfScope prof(s, RECO_PROF_TOWER);
    return run_pairs(t, P, Q, u_gmf, i_gmf, nullptr, nullptr, B, B, B, out, nullptr, nullptr, s);
}

// ---- matrix / top-k forms ----------------------------------------------------------------------
// Head of every call workspace (256 bytes): word 0 = tile work-queue counter (cleared per launch); words kL1Word.. = the
// call's layer-1 scale when the handle has none of its own (resolve_l1).
static const size_t kQueueBytes = 256;       // resolve_l1 clears exactly this much
static size_t proj_bytes(int64_t rows) { return align_up((size_t)(rows > 0 ? rows : 0) * H1 * sizeof(float), 256); }

extern "C" size_t reco_tower_matrix_workspace_bytes(int64_t U, int64_t N) { return kQueueBytes + proj_bytes(U) + proj_bytes(N); }

static int score_matrix_impl(reco_tower* t, const int64_t* users, int64_t user_base, int64_t U,
                             const int64_t* cands, int64_t N, const float* Qready, const float* l1,
                             float* out, int64_t ld_out, float* P, float* Q, unsigned int* queue, bool queue_clean, int32_t* oob,
                             hipStream_t s)
{
    const reco_tower_desc_t& d = t->d;
    if (!t->mfma)
        return gmf_launch_tile(d.user_gmf, d.item_gmf, d.w_predict, users, cands, user_base, d.n_users, d.n_items, U, N,
                               d.dg, out, ld_out, t->bp, d.act, d.lo, d.hi, oob, s);
    // layer-1 projections for exactly these users / candidates, inside the call -- except where the handle already holds them:
    // a user RANGE is a row block of the P the handle projected at creation (every output row of that GEMM is its own k-ordered
    // chain, so a per-call projection of the same rows has the same bits: it was 9 us + a launch gap per step for nothing)
    const float* Pready = (!users && t->P) ? t->P + user_base * H1 : nullptr;
    const float* urows = users ? d.user_mlp : d.user_mlp + user_base * d.dm;
    int rc = tower_project(t, urows, users, d.n_users, Pready ? 0 : U, d.item_mlp, cands, d.n_items, Qready ? 0 : N, P, Q, oob, l1, s);
    if (rc) return rc;
    if (Pready) P = const_cast<float*>(Pready);
    if (queue && !queue_clean) RECO_CHECK_HIP(hipMemsetAsync(queue, 0, sizeof(unsigned int), s));
    ProfScope prof(s, RECO_PROF_TOWER);
    if (t->img_split)
        return tower_launch_tile_split(t->G, t->has_l3, t->img_split, P, Qready ? Qready : Q, d.user_gmf, d.item_gmf, users,
                                       cands, user_base, d.n_users, d.n_items, U, N, out, ld_out, t->bp, d.act, d.lo, d.hi,
                                       oob, queue, t->reserve_cus, s);
    return tower_launch_tile(t->G, t->has_l3, t->img, P, Qready ? Qready : Q, d.user_gmf, d.item_gmf, users, cands,
                             user_base, d.n_users, d.n_items, U, N, out, ld_out, t->bp, d.act, d.lo, d.hi, oob, queue, l1, t->reserve_cus, s);
}

static int check_ranges(const reco_tower* t, const int64_t* users, int64_t user_base, int64_t U, const int64_t* cands,
                        int64_t N, const char* who)
{
    RECO_REQUIRE(t->d.n_users > 0 && t->d.n_items > 0, "%s: handle was created without embedding tables", who);
    if (!users) RECO_REQUIRE(user_base >= 0 && user_base + U <= t->d.n_users, "%s: user range [%lld, %lld) outside [0, %lld)",
                             who, (long long)user_base, (long long)(user_base + U), (long long)t->d.n_users);
    if (!cands) RECO_REQUIRE(N <= t->d.n_items, "%s: N=%lld exceeds the catalog (%lld items)", who, (long long)N,
                             (long long)t->d.n_items);
    return RECO_OK;
}

extern "C" int reco_tower_score_matrix(reco_tower_t* t, const int64_t* users, int64_t user_base, int64_t U,
                                       const int64_t* cands, int64_t N, float* out, int64_t ld_out,
                                       int32_t* oob, void* ws, size_t ws_bytes, void* stream)
{
    RECO_REQUIRE(t, "score_matrix: null handle");
    if (int rc_dev = check_device(t, "score_matrix")) return rc_dev;
    if (U <= 0 || N <= 0) return RECO_OK;
    RECO_REQUIRE(out && ld_out >= N, "score_matrix: bad output");
    int rc = check_ranges(t, users, user_base, U, cands, N, "score_matrix");
    if (rc) return rc;
    float *P = nullptr, *Q = nullptr;
    unsigned int* queue = nullptr;
    if (t->mfma) {
        const size_t need = kQueueBytes + proj_bytes(U) + proj_bytes(N);
        if (!ws || ws_bytes < need) { reco_set_error("score_matrix: workspace too small (%zu < %zu)", ws_bytes, need); return RECO_ENOMEM; }
        queue = static_cast<unsigned int*>(ws);
        P = reinterpret_cast<float*>(static_cast<char*>(ws) + kQueueBytes);
        Q = reinterpret_cast<float*>(static_cast<char*>(ws) + kQueueBytes + proj_bytes(U));
    }
    const float* l1 = nullptr;
    bool clean = false;
    if ((rc = resolve_l1(t, t->d.user_mlp, t->d.n_users, t->d.item_mlp, t->d.n_items, queue, &l1, &clean, (hipStream_t)stream))) return rc;
    const float* Qready = (t->mfma && !cands && N == t->d.n_items && t->Q) ? t->Q : nullptr;       // the whole catalogue: the handle's own projection
    return score_matrix_impl(t, users, user_base, U, cands, N, Qready, l1, out, ld_out, P, Q, queue, clean, oob, (hipStream_t)stream);
}

// user-chunk size for the fused score + top-k path
static const size_t kScoreChunkBytes = (size_t)1 << 30;   // ~1 GiB of score rows per chunk
static const int64_t kTopkExtraRows = 8192;               // bound of rows*splits - rows in topk.hip choose_splits

// leading dimension of the internal score rows: whole 128-byte lines, so that a wave's 32-score store is one line and the
// select reads rows as float4 (N = 3533 rows would otherwise start at odd 4-byte offsets)
static int64_t score_ld(int64_t N) { return (N + 31) / 32 * 32; }

// Whole user blocks per chunk: the exact tile kernels work on blocks of 24 users (12 waves x 2), the split-mode kernel on 32
// (8 waves x 4), and a chunk that ends inside a block leaves part of that block's waves idle in EVERY chunk (1,024 users x 1M
// items went in chunks of 268 = 11 blocks + 4 users: 46 block passes instead of 43; in chunks of 240 the step measured
// 158.1 -> 157.6 ms).  Multiples of 48 are whole blocks of 24 and whole wave rounds (8 users) of the split kernel.
static int64_t whole_blocks(int64_t c) { return c >= 48 ? c - c % 48 : c; }

static int64_t default_chunk(int64_t U, int64_t N)
{
    if (N <= 0) return U;
    int64_t c = (int64_t)(kScoreChunkBytes / ((size_t)score_ld(N) * sizeof(float)));
    if (c < 32) c = 32;
    return c < U ? whole_blocks(c) : U;
}

static size_t topk_chunk_bytes(int64_t Uc, int64_t N, int k)
{
    return kQueueBytes + proj_bytes(N) + proj_bytes(Uc) + align_up((size_t)Uc * score_ld(N) * sizeof(float), 256) +
           align_up((size_t)(Uc + kTopkExtraRows) * k * sizeof(uint64_t), 256);
}

// ---- fused form: tower_tile_topk_kernel keeps every user's running top-k in registers, the score rows never reach HBM ----
// Partial lists ("visitor slots", tower.hip) per user block: a workgroup that joins a block takes one.  Many blocks need few
// (about three visits per CU over a launch).  Few blocks -- fewer than half the CUs: a request for a handful of users over a
// long catalog, or 1,024 users x 1M items = 43 blocks -- need one per CU, because whichever blocks are still running at the end
// must be able to take in every workgroup that has run out of work (with 18 slots two of those 43 blocks ended 5 % late).
// Lists nobody wrote are cleared and skipped, so the slots are capped where that clearing would show: N / (8 k).
static const int kFuseVisitorsMin = 4, kFuseVisitorsMax = 256;
static const size_t kFusePartialBytes = (size_t)256 << 20;   // user chunks are sized so that the partial lists stay below this

static int fuse_visitors(int G, int64_t Uc, int64_t N, int k)
{
    const int64_t blocks = (Uc + tower_tile_users_per_block(G) - 1) / tower_tile_users_per_block(G);
    int64_t v = (3 * 256 + blocks - 1) / (blocks > 0 ? blocks : 1);
    if (blocks < 128) v = kFuseVisitorsMax;
    const int64_t cap = N / (8 * (int64_t)k);
    if (v > cap) v = cap;
    return (int)(v < kFuseVisitorsMin ? kFuseVisitorsMin : (v > kFuseVisitorsMax ? kFuseVisitorsMax : v));
}

static size_t fused_partial_bytes(int64_t Uc, int v, int k) { return align_up((size_t)Uc * v * k * sizeof(uint64_t), 256); }

static size_t fused_chunk_bytes(int G, int64_t Uc, int64_t N, int k)
{
    return kQueueBytes + proj_bytes(N) + proj_bytes(Uc) + tower_topk_counter_bytes(G, Uc) + fused_partial_bytes(Uc, fuse_visitors(G, Uc, N, k), k) + 256;
}

static int64_t fused_default_chunk(int64_t U, int k)
{
    // large chunks run at the minimum list count; small ones hold <= ~768 lists per 24 users, 19 MB at k = 128
    int64_t c = (int64_t)(kFusePartialBytes / ((size_t)kFuseVisitorsMin * k * sizeof(uint64_t)));
    if (c < 32) c = 32;
    return c < U ? whole_blocks(c) : U;
}

// Only the exact MFMA tower has the fused kernel (GMF-only models and the split mode keep score rows in the workspace).
static bool fused_topk_available(const reco_tower* t) { return t->mfma && !t->img_split; }

// reco_tower_score_topk itself takes the fused form where it measured faster: mask-free calls whose score rows would not fit one
// user chunk (1,024 x 1M, k = 100: 157.6 -> 154.3 ms, five tile launches + five scans against one launch + a merge; where one
// chunk suffices the two-kernel form is 0-5 % ahead, DESIGN section 3, fused kernel).
// Not for the Hybrid head shape (GMF width 64): its fused kernel needs ~210 VGPRs, spills 39-45 of them at three waves per SIMD
// and runs 8-11 % behind the two-kernel form on every shape measured (tools/hybrid_fused_ab.py; at 512 threads x 16 users it
// does not spill and is still 5-15 % behind).  reco_tower_score_topk_fused still forces it.
static bool fused_topk_preferred(const reco_tower* t, const uint8_t* exclude, int64_t U, int64_t N)
{
    return fused_topk_available(t) && t->G <= 32 && !exclude && U > default_chunk(U, N);
}

extern "C" size_t reco_tower_topk_workspace_bytes(int64_t U, int64_t N, int k)
{
    if (U <= 0 || N <= 0 || k <= 0) return 0;
    // enough for either form and either user-block size: the handle and the mask are not known here
    const int64_t Uf = fused_default_chunk(U, k);
    const size_t a = topk_chunk_bytes(default_chunk(U, N), N, k), b = fused_chunk_bytes(32, Uf, N, k), c = fused_chunk_bytes(64, Uf, N, k);
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

extern "C" size_t reco_tower_topk_fused_workspace_bytes(const reco_tower_t* t, int64_t U, int64_t N, int k)
{
    if (!t || !fused_topk_available(t) || U <= 0 || N <= 0 || k <= 0) return 0;
    return fused_chunk_bytes(t->G, fused_default_chunk(U, k), N, k);
}

static int score_topk_fused(reco_tower_t* t, const int64_t* users, int64_t user_base, int64_t U, const int64_t* cands, int64_t N,
                            int k, int32_t* out_ids, float* out_scores, int32_t* oob, void* ws, size_t ws_bytes, hipStream_t s,
                            const char* who)
{
    const reco_tower_desc_t& d = t->d;
    // Largest user chunk the workspace admits.  The number of partial lists per block is NOT monotone in the chunk size (few
    // blocks ask for one list per CU, fuse_visitors), so the chunk is sized against the MINIMUM list count and the lists then
    // take whatever room is left, up to what fuse_visitors asks for -- a bounded workspace (the 32 MiB no-alloc scratch) no
    // longer skips feasible chunk sizes.
    auto fixed_bytes = [&](int64_t uc) { return kQueueBytes + proj_bytes(N) + proj_bytes(uc) + tower_topk_counter_bytes(t->G, uc); };
    auto min_bytes = [&](int64_t uc) { return fixed_bytes(uc) + fused_partial_bytes(uc, kFuseVisitorsMin, k) + 256; };
    int64_t Uc = fused_default_chunk(U, k);
    while (Uc > 1 && min_bytes(Uc) > ws_bytes) Uc = whole_blocks(Uc / 2);
    if (!ws || min_bytes(Uc) > ws_bytes) {
        reco_set_error("%s: workspace too small (%zu bytes; one user needs %zu)", who, ws_bytes, min_bytes(1));
        return RECO_ENOMEM;
    }
    char* w = static_cast<char*>(ws) + kQueueBytes;
    float* Q = reinterpret_cast<float*>(w); w += proj_bytes(N);
    float* P = reinterpret_cast<float*>(w); w += proj_bytes(Uc);
    unsigned int* counters = reinterpret_cast<unsigned int*>(w);
    const size_t counter_bytes = tower_topk_counter_bytes(t->G, Uc);       // whole chunks and the last, shorter one share this split
    int64_t list_room = (int64_t)((ws_bytes - fixed_bytes(Uc) - 256) / ((size_t)k * sizeof(uint64_t)));   // partial lists that fit (their block is padded to 256 B)
    if (list_room > Uc * fuse_visitors(t->G, Uc, N, k)) list_room = Uc * fuse_visitors(t->G, Uc, N, k);
    if (list_room > Uc * (int64_t)kFuseVisitorsMax) list_room = Uc * (int64_t)kFuseVisitorsMax;
    unsigned long long* partial = reinterpret_cast<unsigned long long*>(w + counter_bytes);
    const bool cached_q = !cands && N == d.n_items && t->Q;     // as the unfused form below
    int rc = RECO_OK;
    const float* l1 = nullptr;
    bool head_cleared = false;                                  // unused here: the fused form clears its own counters per chunk
    if ((rc = resolve_l1(t, d.user_mlp, d.n_users, d.item_mlp, d.n_items, static_cast<unsigned int*>(ws), &l1, &head_cleared, s))) return rc;
    if (cached_q) Q = t->Q;
    else if ((rc = tower_project(t, nullptr, nullptr, 0, 0, d.item_mlp, cands, d.n_items, N, nullptr, Q, oob, l1, s))) return rc;
    for (int64_t c0 = 0; c0 < U; c0 += Uc) {
        const int64_t n = U - c0 < Uc ? U - c0 : Uc;
        const int64_t* ul = users ? users + c0 : nullptr;
        const float* urows = ul ? d.user_mlp : d.user_mlp + (user_base + c0) * d.dm;
        const float* Pc = P;
        if (!ul && t->P) Pc = t->P + (user_base + c0) * H1;                      // a user range: rows of the handle's own projection
        else if ((rc = tower_project(t, urows, ul, d.n_users, n, nullptr, nullptr, 0, 0, P, nullptr, oob, l1, s))) return rc;
        // hand-out counters and partial lists start at zero (a list nobody wrote reads as empty)
        int v = fuse_visitors(t->G, n, N, k);                                    // a short last chunk may take more lists per block
        if ((int64_t)v * n > list_room) v = (int)(list_room / n);                // >= kFuseVisitorsMin: the chunk was sized for it
        RECO_CHECK_HIP(hipMemsetAsync(counters, 0, counter_bytes + fused_partial_bytes(n, v, k), s));
        {
            ProfScope prof(s, RECO_PROF_TOWER);
            rc = tower_launch_tile_topk(t->G, t->has_l3, t->img, Pc, Q, d.user_gmf, d.item_gmf, ul, cands, user_base + c0, d.n_users,
                                        d.n_items, n, N, t->bp, d.act, d.lo, d.hi, oob, k, v, counters, partial, l1, t->reserve_cus, s);
        }
        if (rc) return rc;
        if ((rc = launch_topk_merge(partial, n, v, k, out_ids + c0 * k, out_scores + c0 * k, s))) return rc;
    }
    return RECO_OK;
}

static int score_topk_impl(reco_tower_t* t, const int64_t* users, int64_t user_base, int64_t U,
                           const int64_t* cands, int64_t N, int k, const uint8_t* exclude, int64_t ld_ex,
                           int32_t* out_ids, float* out_scores, int32_t* oob, void* ws, size_t ws_bytes, hipStream_t s,
                           const char* who, const float* seen = nullptr, int64_t ld_seen = 0)
{
    RECO_REQUIRE(!seen || ld_seen >= N, "%s: ld_seen smaller than N", who);
    RECO_REQUIRE(!(seen && exclude), "%s: give the seen items as a byte mask or as the rating matrix, not both", who);
    RECO_REQUIRE(out_ids && out_scores, "%s: null output", who);
    RECO_REQUIRE(k >= 1 && k <= 128, "%s: k must be in [1,128], got %d", who, k);
    RECO_REQUIRE(N >= k, "%s: k=%d exceeds the number of candidates %lld", who, k, (long long)N);
    RECO_REQUIRE(!exclude || ld_ex >= N, "%s: ld_ex smaller than N", who);
    int rc = check_ranges(t, users, user_base, U, cands, N, who);
    if (rc) return rc;

    if (!seen && fused_topk_preferred(t, exclude, U, N))
        return score_topk_fused(t, users, user_base, U, cands, N, k, out_ids, out_scores, oob, ws, ws_bytes, s, who);

    // largest user chunk the workspace admits
    int64_t Uc = default_chunk(U, N);
    while (Uc > 1 && topk_chunk_bytes(Uc, N, k) > ws_bytes) Uc = whole_blocks(Uc / 2);
    if (!ws || topk_chunk_bytes(Uc, N, k) > ws_bytes) {
        reco_set_error("%s: workspace too small (%zu bytes; one user needs %zu)", who, ws_bytes, topk_chunk_bytes(1, N, k));
        return RECO_ENOMEM;
    }
    char* w = static_cast<char*>(ws);
    unsigned int* queue = t->mfma ? reinterpret_cast<unsigned int*>(w) : nullptr; w += kQueueBytes;
    float* Q = reinterpret_cast<float*>(w); w += proj_bytes(N);
    float* P = reinterpret_cast<float*>(w); w += proj_bytes(Uc);
    const int64_t ldN = score_ld(N);
    float* scores = reinterpret_cast<float*>(w); w += align_up((size_t)Uc * ldN * sizeof(float), 256);
    void* tws = w;
    const size_t tws_bytes = align_up((size_t)(Uc + kTopkExtraRows) * k * sizeof(uint64_t), 256);

    const reco_tower_desc_t& d = t->d;
    // item-side projection once per call; calls over the whole catalogue reuse the projection of all items that the handle
    // packed at creation instead (the serving path, inference_engine.py:149-166, and every full-catalogue batch step).
    const bool cached_q = t->mfma && !cands && N == d.n_items && t->Q;
    const float* l1 = nullptr;
    bool clean = false;
    if ((rc = resolve_l1(t, d.user_mlp, d.n_users, d.item_mlp, d.n_items, queue, &l1, &clean, s))) return rc;
    if (cached_q) Q = t->Q;
    else if (t->mfma) {
        rc = tower_project(t, nullptr, nullptr, 0, 0, d.item_mlp, cands, d.n_items, N, nullptr, Q, oob, l1, s);
        if (rc) return rc;
    }
    for (int64_t c0 = 0; c0 < U; c0 += Uc) {
        const int64_t n = U - c0 < Uc ? U - c0 : Uc;
        rc = score_matrix_impl(t, users ? users + c0 : nullptr, user_base + c0, n, cands, N, Q, l1, scores, ldN, P, Q, queue, clean && c0 == 0, oob, s);
        if (rc) return rc;
        rc = launch_topk_rows_ex(scores, n, N, ldN, k, exclude ? exclude + c0 * ld_ex : nullptr, ld_ex, seen ? seen + c0 * ld_seen : nullptr,
                                 ld_seen, nullptr, n, nullptr, out_ids + c0 * k, out_scores + c0 * k, tws, tws_bytes, s);
        if (rc) return rc;
    }
    return RECO_OK;
}

extern "C" int reco_tower_score_topk_fused(reco_tower_t* t, const int64_t* users, int64_t user_base, int64_t U,
                                           const int64_t* cands, int64_t N, int k, int32_t* out_ids, float* out_scores,
                                           int32_t* oob, void* ws, size_t ws_bytes, void* stream)
{
    RECO_REQUIRE(t, "score_topk_fused: null handle");
    if (int rc_dev = check_device(t, "score_topk_fused")) return rc_dev;
    RECO_REQUIRE(fused_topk_available(t), "score_topk_fused: only the exact MFMA tower has the fused kernel (this handle is %s); "
                 "use reco_tower_score_topk", t->mfma ? "in split bf16x3 mode" : "GMF-only");
    if (U <= 0) return RECO_OK;
    const char* who = "score_topk_fused";
    RECO_REQUIRE(out_ids && out_scores, "%s: null output", who);
    RECO_REQUIRE(k >= 1 && k <= 128, "%s: k must be in [1,128], got %d", who, k);
    RECO_REQUIRE(N >= k, "%s: k=%d exceeds the number of candidates %lld", who, k, (long long)N);
    if (int rc = check_ranges(t, users, user_base, U, cands, N, who)) return rc;
    return score_topk_fused(t, users, user_base, U, cands, N, k, out_ids, out_scores, oob, ws, ws_bytes, (hipStream_t)stream, who);
}

extern "C" int reco_tower_score_topk(reco_tower_t* t, const int64_t* users, int64_t user_base, int64_t U,
                                     const int64_t* cands, int64_t N, int k,
                                     const uint8_t* exclude, int64_t ld_ex,
                                     int32_t* out_ids, float* out_scores, int32_t* oob,
                                     void* ws, size_t ws_bytes, void* stream)
{
    RECO_REQUIRE(t, "score_topk: null handle");
    if (int rc_dev = check_device(t, "score_topk")) return rc_dev;
    if (U <= 0) return RECO_OK;
    return score_topk_impl(t, users, user_base, U, cands, N, k, exclude, ld_ex, out_ids, out_scores, oob, ws, ws_bytes,
                           (hipStream_t)stream, "score_topk");
}

// ---- matrix / top-k forms over row blocks handed in per call (the Hybrid head: tables = this call's encoder latents) ----
// A shallow copy of the handle with the caller's row blocks in place of the tables runs the same code path: the projections
// are recomputed per call inside the workspace anyway, nothing is allocated and nothing synchronises.
static int rows_view(const reco_tower_t* t, const float* u_gmf, const float* i_gmf, const float* u_mlp, const float* i_mlp,
                     int64_t U, int64_t N, reco_tower* v, const char* who)
{
    RECO_REQUIRE(t, "%s: null handle", who);
    if (int rc_dev = check_device(t, who)) return rc_dev;
    RECO_REQUIRE(U >= 0 && N >= 0 && U < (1ll << 31) && N < (1ll << 31), "%s: bad row counts", who);
    if (t->d.dg > 0) RECO_REQUIRE(u_gmf && i_gmf, "%s: GMF rows missing", who);
    if (t->d.dm > 0) RECO_REQUIRE(u_mlp && i_mlp, "%s: MLP rows missing", who);
    RECO_REQUIRE((uintptr_t)u_gmf % 16 == 0 && (uintptr_t)i_gmf % 16 == 0 && (uintptr_t)u_mlp % 16 == 0 && (uintptr_t)i_mlp % 16 == 0,
                 "%s: row blocks must be 16-byte aligned (rows are read as float4)", who);
    *v = *t;
    v->d.user_gmf = u_gmf; v->d.item_gmf = i_gmf; v->d.user_mlp = u_mlp; v->d.item_mlp = i_mlp;
    v->d.n_users = U; v->d.n_items = N;
    v->P = nullptr; v->Q = nullptr;        // the handle's cached projections and layer-1 scale belong to its own tables
    v->l1 = nullptr;
    return RECO_OK;
}

extern "C" int reco_tower_score_matrix_rows(reco_tower_t* t, const float* u_gmf, const float* i_gmf,
                                            const float* u_mlp, const float* i_mlp, int64_t U, int64_t N,
                                            float* out, int64_t ld_out, void* ws, size_t ws_bytes, void* stream)
{
    reco_tower v;
    if (int rc = rows_view(t, u_gmf, i_gmf, u_mlp, i_mlp, U, N, &v, "score_matrix_rows")) return rc;
    if (U <= 0 || N <= 0) return RECO_OK;
    return reco_tower_score_matrix(&v, nullptr, 0, U, nullptr, N, out, ld_out, nullptr, ws, ws_bytes, stream);
}

extern "C" int reco_tower_score_topk_rows(reco_tower_t* t, const float* u_gmf, const float* i_gmf,
                                          const float* u_mlp, const float* i_mlp, int64_t U, int64_t N, int k,
                                          const uint8_t* exclude, int64_t ld_ex, int32_t* out_ids, float* out_scores,
                                          void* ws, size_t ws_bytes, void* stream)
{
    reco_tower v;
    if (int rc = rows_view(t, u_gmf, i_gmf, u_mlp, i_mlp, U, N, &v, "score_topk_rows")) return rc;
    if (U <= 0) return RECO_OK;
    return reco_tower_score_topk(&v, nullptr, 0, U, nullptr, N, k, exclude, ld_ex, out_ids, out_scores, nullptr, ws, ws_bytes, stream);
}

// the ranking evaluation's form: seen items are read off the float rating matrix itself (entry > 0), as reco_topk_rows_seen does
extern "C" int reco_tower_score_topk_rows_seen(reco_tower_t* t, const float* u_gmf, const float* i_gmf,
                                               const float* u_mlp, const float* i_mlp, int64_t U, int64_t N, int k,
                                               const float* seen, int64_t ld_seen, int32_t* out_ids, float* out_scores,
                                               void* ws, size_t ws_bytes, void* stream)
{
    reco_tower v;
    if (int rc = rows_view(t, u_gmf, i_gmf, u_mlp, i_mlp, U, N, &v, "score_topk_rows_seen")) return rc;
    if (U <= 0) return RECO_OK;
    return score_topk_impl(&v, nullptr, 0, U, nullptr, N, k, nullptr, 0, out_ids, out_scores, nullptr, ws, ws_bytes, (hipStream_t)stream,
                           "score_topk_rows_seen", seen, ld_seen);
}

// ---------------------------------------------------------------------------------------------
// score_topk for callers that cannot allocate (HIP graph capture, tight serving loops): the call workspace is a
// file-scope device array instead of a caller buffer.  Every GPU the library runs on holds its own copy of the array
// (at its own address) until the process ends.  The array's NAME is a host-side stand-in, not a device address: host
// code only ever uses the per-device pointer that hipGetSymbolAddress returned, looked up once per device in
// reco_tower_create (the library's allocating set-up call) with that device current.
// ---------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[RECO_NOALLOC_SCRATCH_BYTES];
static void* g_scratch_ptr[RECO_MAX_DEVICES] = {};      // device -> address of that device's copy of `scratch`
static std::mutex g_scratch_mu;

// called from reco_tower_create with the handle's device current
static int scratch_lookup(int device)
{
    RECO_REQUIRE(device >= 0 && device < RECO_MAX_DEVICES, "device index %d out of range", device);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device]) return RECO_OK;
    void* ptr = nullptr;
    RECO_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    RECO_REQUIRE(ptr != nullptr, "hipGetSymbolAddress returned a null address for the static scratch array");
    g_scratch_ptr[device] = ptr;
    return RECO_OK;
}

extern "C" size_t reco_noalloc_scratch_bytes(void) { return RECO_NOALLOC_SCRATCH_BYTES; }

extern "C" int reco_tower_score_topk_noalloc(reco_tower_t* t, const int64_t* users, int64_t user_base, int64_t U,
                                             const int64_t* cands, int64_t N, int k,
                                             const uint8_t* exclude, int64_t ld_ex,
                                             int32_t* out_ids, float* out_scores, int32_t* oob, void* stream)
{
    RECO_REQUIRE(t, "score_topk_noalloc: null handle");
    if (int rc_dev = check_device(t, "score_topk_noalloc")) return rc_dev;
    if (U <= 0) return RECO_OK;
    RECO_REQUIRE(N > 0 && k >= 1 && k <= 128, "score_topk_noalloc: need N > 0 and k in [1,128], got N=%lld k=%d", (long long)N, k);
    const size_t one_user = topk_chunk_bytes(1, N, k);
    RECO_REQUIRE(one_user <= (size_t)RECO_NOALLOC_SCRATCH_BYTES,
                 "score_topk_noalloc: N=%lld candidates at k=%d need %zu bytes of scratch per user chunk, the static scratch "
                 "holds %zu; use reco_tower_score_topk with a caller workspace", (long long)N, k, one_user,
                 (size_t)RECO_NOALLOC_SCRATCH_BYTES);
    void* ptr = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        ptr = g_scratch_ptr[t->device];
    }
    RECO_REQUIRE(ptr != nullptr, "score_topk_noalloc: no scratch address recorded for device %d", t->device);
    // the array keeps the previous call's bytes.  What must start at zero is cleared on `stream` before every launch that
    // reads it: the tile queue counter (score_matrix_impl), or -- when the call takes the fused form -- the hand-out counters
    // and the partial lists (score_topk_fused); every other region is written before it is read within the call
    return score_topk_impl(t, users, user_base, U, cands, N, k, exclude, ld_ex, out_ids, out_scores, oob, ptr,
                           (size_t)RECO_NOALLOC_SCRATCH_BYTES, (hipStream_t)stream, "score_topk_noalloc");
}
