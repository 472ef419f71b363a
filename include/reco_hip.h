/*
 * reco_hip.h -- C ABI of the MI355X (gfx950) scoring engine.
 *
 * This is the drop-in boundary for the reference's forward/scoring hot path
 * (SURVEY.md section 8b).  The reference itself has no native code and no FFI:
 * every entry point below replaces a chain of stock ATen calls made from the
 * reference's Python, cited per function as <file>:<line> relative to the
 * reference root.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - floats are fp32, contiguous, row-major; Linear weights are [out, in] as nn.Linear stores them;
 *   - embedding tables and row blocks handed to a TOWER are 16-byte aligned and their widths multiples of 4 (rows move as
 *     float4; anything torch allocates qualifies); a violation is RECO_EINVAL, never a misaligned access.  The stand-alone
 *     gathers (K-a) take any width and alignment, as nn.Embedding does: widths that are not multiples of 4 or misaligned
 *     pointers run a one-float-per-thread form of the same kernels;
 *   - ids (users, items, candidate lists) are int64 as the reference's torch LongTensors; returned
 *     top-k positions are int32;
 *   - `stream` is a hipStream_t passed as void*; work is only enqueued, never synchronised,
 *     unless the comment says "synchronises";
 *   - return 0 on success, a negative RECO_E* code otherwise; reco_last_error() gives the
 *     thread-local message;
 *   - ids out of range never fault and never assert on the device: they are clamped to the nearest
 *     valid row and counted in the call's `oob` argument (device int32, nullable = do not count; the caller zeroes it
 *     and reads it back when it wants nn.Embedding's IndexError, src/api/core/inference_engine.py:17-31).  The counter
 *     belongs to the call, not to a handle: concurrent callers never see each other's ids;
 *   - NaN conventions: ReLU is max on the raw bits (one VALU instruction beside the MFMAs): a NaN with the sign bit set
 *     becomes +0 where torch.relu would propagate it; NaN inputs are outside the parity contract;
 *   - the exact tower kernels evaluate layer 1 on activations multiplied by a power of two s (one clamped add per element instead
 *     of add + max; scores keep their bits: DESIGN.md section 3 K-b).  s is derived in reco_tower_create from a bound over the
 *     handle's MLP tables and first-layer weights (per call for the *_rows forms), so -- like the cached projections -- it
 *     belongs to the table contents at creation: rebuild the handle after changing them.  Non-finite table entries are left
 *     out of that bound, and s is kept within [2^-80, 2^24]: a pair that touches a non-finite entry, or whose layer-1
 *     pre-activation exceeds 2^80, gets a finite, meaningless score where the reference has inf / NaN / a huge value; every
 *     other pair is unaffected by such rows;
 *   - nothing allocates except *_create (the packed-model handles) -- scratch memory is
 *     brought in by the caller as `ws` / `ws_bytes` (query the size with *_workspace_bytes);
 *   - mutable state outside the handles: the thread-local error string, the thread-local measurement hook
 *     (reco_profile_*, armed per host thread), and the per-device address of the static scratch array of
 *     reco_tower_score_topk_noalloc (looked up once under a lock).  Nothing else is shared between threads.
 */
#ifndef RECO_HIP_H
#define RECO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RECO_OK          0
#define RECO_EINVAL     -1   /* bad argument / unsupported shape  -> ValueError   */
#define RECO_EHIP       -2   /* HIP runtime failure               -> RuntimeError */
#define RECO_ENOMEM     -3   /* workspace too small / hipMalloc   -> MemoryError  */
#define RECO_EINDEX     -4   /* id out of range (see reco_*_status) -> IndexError */

/* epilogue applied to the logit / GEMM output */
#define RECO_ACT_NONE     0
#define RECO_ACT_RELU     1
#define RECO_ACT_SIGMOID  2
#define RECO_ACT_CLAMP    3   /* clamp(x, lo, hi)            src/api/core/inference_engine.py:232 */
#define RECO_ACT_RATING   4   /* sigmoid(x)*(hi-lo)+lo       src/ncf/evaluate_ncf_rating.py:144-146 */

int         reco_abi_version(void);
const char* reco_last_error(void);

/* ------------------------------------------------------------------------------------------
 * K-a  embedding gathers (HBM-bound)
 * replaces nn.Embedding.__call__            src/helpers/ncf_model.py:207-208,220-221
 * ---------------------------------------------------------------------------------------- */

/* out[b, :] = table[idx[b], :]   any dim > 0 (dim % 4 == 0 and 16-byte aligned pointers: the float4 form).  Out-of-range
 * ids are clamped and counted in *oob (device int32, nullable). */
int reco_gather_rows(const float* table, int64_t rows, int dim, const int64_t* idx, int64_t B,
                     float* out, int32_t* oob, void* stream);

/* Fused 4-table NeuMF gather: the tensors the reference materialises before its MLP,
 *   gmf_prod[b, :]  = Eu_g[user[b]] * Ei_g[item[b]]                  ncf_model.py:207-211
 *   mlp_in[b, :]    = [Eu_m[user[b]] | Ei_m[item[b]]]  (user half first) ncf_model.py:220-224
 * Any dg / dm > 0 (the float4 form needs dg, dm and the strides to be multiples of 4 and 16-byte aligned pointers; anything
 * else runs one float per thread).  Either output may be NULL to skip that half.  ld_gmf / ld_mlp are the
 * outputs' row strides in floats (0 = dense: dg and 2*dm), so a product can be
 * written straight into a column block of a wider buffer (the predict layer's concat, ncf_model.py:240). */
int reco_neumf_gather(const float* Eu_g, const float* Ei_g, const float* Eu_m, const float* Ei_m,
                      int64_t n_users, int64_t n_items, int dg, int dm,
                      const int64_t* user, const int64_t* item, int64_t B,
                      float* gmf_prod, int64_t ld_gmf, float* mlp_in, int64_t ld_mlp,
                      int32_t* oob, void* stream);

/* ------------------------------------------------------------------------------------------
 * K-c  Linear layer as an fp32 MFMA GEMM with fused epilogue
 *   C[m, n] = act( sum_k A[row(m), k] * W[n, k] + bias[n] )
 * replaces nn.Linear + Sigmoid/ReLU/clamp   src/autorec/utils/model.py:25-33,
 *                                           src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:78-106
 * a_rows (nullable, int64[M]) gathers A rows: row(m) = a_rows[m], else m.  lda/ldw/ldc are row
 * strides in floats (so W may be a column block of a wider matrix).
 * ws (nullable) enables split-K for small output grids with a long K (AutoRec encoder, Hybrid
 * encoders): K slices are reduced in slice order, so results stay bitwise reproducible.
 * ---------------------------------------------------------------------------------------- */
#define RECO_LINEAR_A_TRANSPOSED 1   /* A is stored [K, M] with row stride lda: the item-vector batch X = R^T of
                                       inference_engine.py:230 / helper.py:283 is read straight out of R [users, items] */
#define RECO_LINEAR_BIAS_PER_ROW 2   /* bias[m] instead of bias[n]: R_hat^T = Wd H^T + bd, no transposed copy */
#define RECO_LINEAR_SPLIT_BF16X3 4   /* multiply as three bf16 pieces per fp32 operand on the bf16 MFMA, fp32 accumulate (see RECO_PREC_SPLIT_BF16X3) */
size_t reco_linear_workspace_bytes(int64_t M, int64_t N, int64_t K);
int reco_linear_f32(const float* A, int64_t lda, const int64_t* a_rows, int64_t a_row_limit, int64_t M,
                    const float* W, int64_t ldw, const float* bias, int64_t N, int64_t K,
                    int act, float lo, float hi, float* C, int64_t ldc, int flags,
                    void* ws, size_t ws_bytes, void* stream);

/* Linear layer over a SPARSE input given as CSR -- the "masked-observed" form of the AutoRec / Hybrid encoders, whose input
 * is the 4.5 %-dense rating matrix (or its transpose) with unobserved entries fed as literal zeros:
 *   src/autorec/utils/model.py:31-33 with X from src/api/core/inference_engine.py:230 / src/api/core/data_loader.py:96-103;
 *   src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:78-89 (first encoder layer).
 *   C[m, :] = act( sum over the nonzeros (k, x) of row m in ascending k of  x * Wt[k, :]  + bias )
 * row_ptr [M+1] int64, col [nnz] int32 (ascending within a row; values outside [0, K) are clamped, never read out of range),
 * val [nnz]; Wt = W^T [K, N] row-major with row stride ldwt (reco_transpose_f32 of the nn.Linear weight, once per weight
 * version).  Zeros contribute nothing to an fp32 sum: a row that is ONE segment is the dense layer's k-ordered fmaf chain without
 * its zero terms, bit for bit; a row cut into several segments sums each on its own and adds the partial sums in segment order
 * (plain adds: another rounding sequence, same fp32 gate).  The segment table cuts the rows into pieces of consecutive
 * nonzeros, one wave each; their length is the caller's choice (any length is computed correctly; the Python side uses whole
 * rows up to 256 nonzeros and pieces of 96 once a row is longer: a wave walks its segment one weight row at a time, so the
 * longest segment bounds the launch):
 *   row_seg_ptr [M+1] int64   segments of row m are row_seg_ptr[m] .. row_seg_ptr[m+1]-1 (a row without nonzeros has none),
 *   seg_row [S] int32, seg_begin / seg_end [S] int64 (offsets into col / val);
 * a row with one segment is finished in place, the partial sums of the others are added in segment order by a second launch
 * (deterministic; pass single_segment_rows = 1 when EVERY row has exactly one segment: no second launch, no workspace).
 * order (nullable, int32 [n_order]): the placement table.  Workgroup b runs the segments order[4 b .. 4 b + 3] (negative =
 * none; RECO_CSR_SEGS_PER_WG = 4 slots per workgroup, one per wave).  The host side uses it for XCD placement: segments are cut
 * at the boundaries of eight k ranges and every slot of workgroup b holds a segment of range b % 8 -- workgroups b and b + 8
 * share an XCD on MI355X, so each XCD keeps one eighth of Wt in its own L2 (speed only, never correctness).  NULL: slot = segment.
 * ws: reco_csr_linear_workspace_bytes(S, N). */
#define RECO_CSR_SEGS_PER_WG 4
size_t reco_csr_linear_workspace_bytes(int64_t S, int64_t N);
int reco_csr_linear_f32(const int64_t* row_ptr, const int32_t* col, const float* val, int64_t M, int64_t K,
                        const int64_t* row_seg_ptr, const int32_t* seg_row, const int64_t* seg_begin, const int64_t* seg_end,
                        int64_t S, const int32_t* order, int64_t n_order, int single_segment_rows,
                        const float* Wt, int64_t ldwt, const float* bias, int64_t N,
                        int act, float lo, float hi, float* C, int64_t ldc, void* ws, size_t ws_bytes, void* stream);

/* out[c, r] = in[r, c]  (weight layout for reco_csr_linear_f32; no counterpart in the reference) */
int reco_transpose_f32(const float* in, int64_t rows, int64_t cols, int64_t ld_in, float* out, int64_t ld_out, void* stream);

/* Two consecutive narrow Linear layers in one launch:  C = act3( act2( X W2^T + b2 ) W3^T + b3 )
 * replaces the two narrow layers of the Hybrid model's encoders / decoders (Linear+ReLU, Linear)
 *   src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:78-89 (encoder.3, encoder.6), :94-106 (decoder.0, decoder.3)
 * whose [rows, N2] intermediate stays in LDS.  X [M, K1] (row stride ldx), W2 [N2, K1], W3 [N3, N2] as nn.Linear stores them;
 * b2 / b3 nullable; lo / hi serve both activations.  Widths: reco_mlp2_supported(K1, N2, N3) != 0 (K1 a multiple of 8 up to 512,
 * N2 and N3 multiples of 32, both row tiles within 96 KB of LDS); everything 16-byte aligned, ldx / ldc multiples of 4.
 * Results equal two reco_linear_f32 calls bit for bit (same k-ordered chain, bias added after it). */
int reco_mlp2_supported(int64_t K1, int64_t N2, int64_t N3);
int reco_mlp2_f32(const float* X, int64_t ldx, int64_t M, int64_t K1,
                  const float* W2, const float* b2, int64_t N2, int act2,
                  const float* W3, const float* b3, int64_t N3, int act3, float lo, float hi,
                  float* C, int64_t ldc, void* stream);

/* AutoRec forward: out = act( sigmoid(X We^T + be) Wd^T + bd )        src/autorec/utils/model.py:31-33
 * X [B, n_in], We [hidden, n_in], Wd [n_in, hidden], hidden_ws [B, hidden] (caller scratch),
 * out [B, n_in].  act = RECO_ACT_NONE (the module's forward) or RECO_ACT_CLAMP with lo/hi
 * (the engines' clamp(1,5), inference_engine.py:232, helper.py:289). */
size_t reco_autorec_workspace_bytes(int64_t B, int64_t n_in, int64_t hidden);
int reco_autorec_forward(const float* X, int64_t B, const float* We, const float* be,
                         const float* Wd, const float* bd, int64_t n_in, int64_t hidden,
                         int act, float lo, float hi, int flags /* 0 or RECO_LINEAR_SPLIT_BF16X3 */,
                         float* hidden_ws, float* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K-b  fused scoring tower (NeuMF, MLP-only, GMF-only, Hybrid head)
 * ---------------------------------------------------------------------------------------- */
typedef struct reco_tower reco_tower_t;

typedef struct reco_tower_desc {
    /* embedding tables (borrowed, must outlive the handle).  NULL when the branch is absent.
     * NeuMF: src/helpers/ncf_model.py:61-70.  Hybrid head: the "tables" are the encoder
     * latents zu [U,64] / zi [I,64] for BOTH gmf and mlp (evaluate_hybridautorecncf_rating.py:167-173). */
    const float* user_gmf;   /* [n_users, dg] */
    const float* item_gmf;   /* [n_items, dg] */
    const float* user_mlp;   /* [n_users, dm] */
    const float* item_mlp;   /* [n_items, dm] */
    int64_t n_users, n_items;
    int32_t dg;              /* GMF width: 32 (NeuMF factor_num) or 64 (Hybrid latent); 0 = no GMF branch */
    int32_t dm;              /* MLP embedding width per side: 128 (NeuMF) or 64 (Hybrid); 0 = no MLP branch */
    /* MLP tower, nn.Linear layout [out,in]:  ncf_model.py:84-99 (MLP_layers.{1,4,7}),
     * hybrid: ncf.mlp.{0,3}.  n_layers = 3 (2dm->128->64->32) or 2 (2dm->128->64). ReLU after every layer. */
    int32_t n_layers;
    int32_t layer_out[3];    /* output width of each layer; the MFMA path requires {128, 64, 32} / {128, 64} */
    const float* W[3];
    const float* b[3];
    /* predict layer: [gmf half | mlp half]  ncf_model.py:113,240-246; hybrid ncf.output */
    const float* w_predict;  /* [dg + mlp_out] */
    const float* b_predict;  /* [1] */
    int32_t act;             /* RECO_ACT_NONE (logit, ncf_model.py:246-249) or RECO_ACT_RATING */
    float lo, hi;
    int32_t precision;       /* RECO_PREC_F32 (default) or RECO_PREC_SPLIT_BF16X3, see below */
} reco_tower_desc_t;

/* How layers 2/3 of the MFMA path multiply.  Both accumulate in fp32 and meet the same parity gate (1e-5 relative
 * against the reference's fp32 and fp64 outputs).
 *   RECO_PREC_F32           fp32 operands on v_mfma_f32_32x32x2_f32: every product is the exact fp32 product.
 *   RECO_PREC_SPLIT_BF16X3  each fp32 operand is split exactly into three bf16 pieces (hi + mid + lo) and the six
 *                           cross terms down to 2^-24 relative are summed on v_mfma_f32_32x32x16_bf16; the dropped
 *                           terms are below fp32 rounding.  Finite inputs only (an Inf operand yields NaN).  No
 *                           counterpart in the reference, which computes in fp32 on the CPU. */
#define RECO_PREC_F32           0
#define RECO_PREC_SPLIT_BF16X3  1

/* Packs the weights into the MFMA/LDS image and precomputes the layer-1 projections
 *   P[u] = W1[:, :dm] Eu_m[u] + b1,  Q[i] = W1[:, dm:] Ei_m[i]        (exact split of ncf_model.py:224-227 layer 1)
 * Allocates device memory; synchronises `stream`. */
int  reco_tower_create(const reco_tower_desc_t* desc, void* stream, reco_tower_t** out);
void reco_tower_destroy(reco_tower_t* t);

/* Debug aid, NO COUNTERPART IN THE REFERENCE.  The layer-1 scale s (Conventions) is fixed when the handle is created; a later
 * in-place write to the MLP tables or the first-layer weights that no caller-side cache can see (torch: param.data, which moves no
 * version counter) leaves the handle with stale projections AND, if values grew, with activations clamped at 1/s.  This call
 * recomputes the bound over the tables' current contents (one pass over both MLP tables, a few small launches on `stream`) and
 * sets *stale (device int32, never cleared here) to 1 when the handle's s no longer covers it -- rebuild the handle then.
 * Handles without a stored scale (GMF-only, split mode, no tables) report nothing.  Calls on one handle must not overlap. */
int reco_tower_check_scale(reco_tower_t* t, int32_t* stale, void* stream);

/* Multi-GPU helper, NO COUNTERPART IN THE REFERENCE (which is single-device).  The persistent tile kernels of score_matrix /
 * score_topk launch one workgroup per CU and each takes 152 of the CU's 160 KB of LDS, so a kernel of another stream -- RCCL's
 * all-gather of the previous step's top-k lists (recommender_amd/dist.py) -- finds no CU until a tile launch ends.  With
 * cus > 0 the tile kernels of this handle launch (CUs - cus) workgroups; tiles come from a work queue, so results do not change.
 * 0 (the default) = every CU.  cus in [0, 128].  Applies to calls issued after it returns; not synchronised with calls that
 * run concurrently on other host threads. */
int reco_tower_set_reserved_cus(reco_tower_t* t, int cus);

/* forward(user, item) -> score[B]           src/helpers/ncf_model.py:188-249
 * Out-of-range ids never fault: they are clamped and counted in *oob (see Conventions). */
int reco_tower_score_pairs(reco_tower_t* t, const int64_t* user, const int64_t* item, int64_t B,
                           float* out, int32_t* oob, void* stream);

/* Score B pairs whose embedding rows are handed over directly (row b of each matrix belongs to
 * pair b) -- the Hybrid forward, where every pair brings its own encoder latents
 * (evaluate_hybridautorecncf_rating.py:197-208).  u_gmf/i_gmf [B,dg], u_mlp/i_mlp [B,dm];
 * for the Hybrid head pass the same latent matrix for gmf and mlp.  The handle may have been
 * created without tables (n_users = n_items = 0).
 * Hybrid head shape (dg = dm = 64, two layers) in RECO_PREC_F32 with u_gmf == u_mlp and i_gmf == i_mlp (what
 * HybridAutoRecNCF.forward passes): ONE kernel runs both MLP layers and the GMF product out of the 512 B of latents per pair
 * (tower_rows.hip); no projection buffers, ws may be NULL -- reco_tower_rows_workspace_bytes_for() then returns 0.  Any other
 * call first writes the layer-1 projections [B,128] x 2 into ws (reco_tower_rows_workspace_bytes(B)). */
size_t reco_tower_rows_workspace_bytes(int64_t B);
size_t reco_tower_rows_workspace_bytes_for(const reco_tower_t* t, int gmf_rows_are_mlp_rows, int64_t B);
int reco_tower_score_rows(reco_tower_t* t, const float* u_gmf, const float* i_gmf,
                          const float* u_mlp, const float* i_mlp, int64_t B, float* out,
                          void* ws, size_t ws_bytes, void* stream);

/* Full (user x candidate) score matrix: out[u, n] = score(users[u], cands[n])
 * replaces the per-user loop around model(user_t, item_t)  src/api/core/inference_engine.py:149-164
 * users NULL -> user_base + u;  cands NULL -> n.  out row stride ld_out (floats).
 * Explicit id lists (users / cands given) are projected through layer 1 inside the call; a user RANGE and the whole catalogue
 * (cands NULL, N = n_items) read the rows the handle projected at creation (same bits: every row is its own k-ordered chain). */
size_t reco_tower_matrix_workspace_bytes(int64_t U, int64_t N);
int reco_tower_score_matrix(reco_tower_t* t, const int64_t* users, int64_t user_base, int64_t U,
                            const int64_t* cands, int64_t N, float* out, int64_t ld_out,
                            int32_t* oob, void* ws, size_t ws_bytes, void* stream);

/* score_matrix + per-user top-k (K-d), user-chunked so the score rows live only in `ws`.
 * replaces model(...) + np.argsort(scores)[::-1][:k]       inference_engine.py:162-166
 * exclude (nullable, uint8 [U, N], row stride ld_ex) applies helper.py:313:
 *   score = score*(1-seen) - seen*1e10.
 * out_ids [U,k] int32 = POSITIONS in the candidate list (== item ids when cands is NULL),
 * out_scores [U,k]; order: descending score, ascending position among equals.
 * k <= 128.  Mask-free calls on the exact MFMA tower (NeuMF / MLP shapes) whose score rows exceed one 1 GiB user chunk run as
 * reco_tower_score_topk_fused below (same results, one launch instead of one per chunk); the workspace size covers both. */
size_t reco_tower_topk_workspace_bytes(int64_t U, int64_t N, int k);
int reco_tower_score_topk(reco_tower_t* t, const int64_t* users, int64_t user_base, int64_t U,
                          const int64_t* cands, int64_t N, int k,
                          const uint8_t* exclude, int64_t ld_ex,
                          int32_t* out_ids, float* out_scores, int32_t* oob,
                          void* ws, size_t ws_bytes, void* stream);

/* The same selection with NO score rows in memory: one kernel scores and keeps every user's running top-k in registers
 * (tower_tile_topk_kernel), a second folds the few partial lists per user.  Mask-free calls on the exact MFMA tower only
 * (a GMF-only or split-mode handle returns RECO_EINVAL); results are bit-identical to reco_tower_score_topk.
 * replaces the same reference lines (inference_engine.py:162-166; SURVEY section 7 "the score row never reaches HBM").
 * Trade (measured): HBM writes per launch fall from the score rows (85 MB at 6038 x 3533) to the lists and
 * the workspace from U x N floats to a few lists per user; where the score rows fit ONE user chunk (1 GiB) the two-kernel form
 * is 1-6 % faster (the list upkeep shares the SIMDs with the MFMAs; a request for a handful of users over a long catalog loses
 * more; the Hybrid head shape loses 8-11 % everywhere), where they do not (1,024 x 1M: five chunks) this form is level or ahead
 * at half the workspace -- and reco_tower_score_topk takes it by itself for such mask-free calls on the NeuMF / MLP shapes.
 * This entry point forces it for any shape.  The workspace size depends on the handle (its user-block size). */
size_t reco_tower_topk_fused_workspace_bytes(const reco_tower_t* t, int64_t U, int64_t N, int k);
int reco_tower_score_topk_fused(reco_tower_t* t, const int64_t* users, int64_t user_base, int64_t U,
                                const int64_t* cands, int64_t N, int k,
                                int32_t* out_ids, float* out_scores, int32_t* oob,
                                void* ws, size_t ws_bytes, void* stream);

/* score_matrix / score_topk over row blocks handed in per call: out[u, n] = score(row u of the user blocks, row n of the item
 * blocks) for all U x N combinations -- the Hybrid model's full-catalog evaluation, where the "tables" are this call's encoder
 * latents (replaces the O(blocks x items) full-model loop of hybridautorecncf.ipynb cell 13 around
 * evaluate_hybridautorecncf_rating.py:197-208).  u_gmf/i_gmf [U|N, dg], u_mlp/i_mlp [U|N, dm]; the handle may have been created
 * without tables.  Workspace sizes: reco_tower_matrix_workspace_bytes(U, N) / reco_tower_topk_workspace_bytes(U, N, k).
 * Nothing is allocated, nothing synchronises; top-k ids are row numbers of the item blocks. */
int reco_tower_score_matrix_rows(reco_tower_t* t, const float* u_gmf, const float* i_gmf,
                                 const float* u_mlp, const float* i_mlp, int64_t U, int64_t N,
                                 float* out, int64_t ld_out, void* ws, size_t ws_bytes, void* stream);
int reco_tower_score_topk_rows(reco_tower_t* t, const float* u_gmf, const float* i_gmf,
                               const float* u_mlp, const float* i_mlp, int64_t U, int64_t N, int k,
                               const uint8_t* exclude, int64_t ld_ex, int32_t* out_ids, float* out_scores,
                               void* ws, size_t ws_bytes, void* stream);

/* reco_tower_score_topk_rows with the ranking evaluation's seen-item rule (src/autorec/utils/helper.py:272,313 as used by the
 * Hybrid evaluation, hybridautorecncf.ipynb cell 13): `seen` (nullable) is the float rating matrix itself, [U, N] with row stride
 * ld_seen; an entry > 0 marks the item as seen (score*(1-seen) - seen*1e10) -- no mask tensor is built. */
int reco_tower_score_topk_rows_seen(reco_tower_t* t, const float* u_gmf, const float* i_gmf,
                                    const float* u_mlp, const float* i_mlp, int64_t U, int64_t N, int k,
                                    const float* seen, int64_t ld_seen, int32_t* out_ids, float* out_scores,
                                    void* ws, size_t ws_bytes, void* stream);

/* The same call for callers that cannot allocate or bring a workspace (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Identical results to
 * reco_tower_score_topk; the call workspace is a static device array of RECO_NOALLOC_SCRATCH_BYTES inside the library
 * (one copy per GPU, held until the process ends).  Calls no hipMalloc / hipFree / allocating routine.
 * Limits: the candidates of ONE user chunk must fit, i.e. reco_tower_topk_workspace_bytes(1, N, k) <=
 * RECO_NOALLOC_SCRATCH_BYTES (about 48,000 candidates at k = 128, 63,000 at k = 10) -- beyond that the call returns
 * RECO_EINVAL and launches nothing; any number of users (they are chunked to fit).
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) RACE; serialise
 * them or use reco_tower_score_topk with separate workspaces. */
#define RECO_NOALLOC_SCRATCH_BYTES (32u << 20)
size_t reco_noalloc_scratch_bytes(void);
int reco_tower_score_topk_noalloc(reco_tower_t* t, const int64_t* users, int64_t user_base, int64_t U,
                                  const int64_t* cands, int64_t N, int k,
                                  const uint8_t* exclude, int64_t ld_ex,
                                  int32_t* out_ids, float* out_scores, int32_t* oob, void* stream);

/* ------------------------------------------------------------------------------------------
 * K-d  per-row top-k select (wavefront-level)
 * replaces np.argsort(...)[::-1][:k] (inference_engine.py:166), torch.topk (ncf.ipynb cell 8),
 *          np.argsort(p)[-k:][::-1] (helper.py:335), list.sort(reverse=True)[:k] (inference_engine.py:294,341)
 * scores [rows, N] with row stride ld.  NaN ranks first (as numpy/torch).  k <= 128, k <= N.
 * ---------------------------------------------------------------------------------------- */
size_t reco_topk_workspace_bytes(int64_t rows, int64_t N, int k);
int reco_topk_rows(const float* scores, int64_t rows, int64_t N, int64_t ld, int k,
                   const uint8_t* exclude, int64_t ld_ex,
                   int32_t* out_ids, float* out_vals, void* ws, size_t ws_bytes, void* stream);

/* The AutoRec / Hybrid ranking evaluation's form of the same select (src/autorec/utils/helper.py:272,313,335):
 *   train_mask = train_mat > 0;  predictions = predictions*(~train_mask) - train_mask*1e10;  argsort(...)[-k:][::-1]
 * `seen` (nullable) is the float rating matrix itself, [src_rows, N] with row stride ld_seen: an entry > 0 marks the item as
 * seen -- no mask tensor is built.  row_ids (nullable, int64 [rows]) selects the users: output row r is the top-k of source
 * row row_ids[r] (ids outside [0, src_rows) are clamped and counted in *oob); NULL = every source row in order (rows ignored).
 * Workspace: reco_topk_workspace_bytes(number of output rows, N, k). */
int reco_topk_rows_seen(const float* scores, int64_t src_rows, int64_t N, int64_t ld, int k,
                        const int64_t* row_ids, int64_t rows, const float* seen, int64_t ld_seen,
                        int32_t* out_ids, float* out_vals, int32_t* oob, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement hook (no counterpart in the reference): when armed, the library brackets the launch of each
 * roofline kernel with a hipEvent pair recorded on the launch stream, so that a caller can time the KERNEL
 * (not the call) without a profiler.  The hook belongs to the calling HOST THREAD: reco_profile_enable(n)
 * allocates n event pairs for the calling thread (n = 0 disarms and frees them) and only launches issued by
 * that thread are bracketed; other threads are unaffected and nothing is shared between threads.
 * What is bracketed (kind):
 *   RECO_PROF_TOWER   tower tile / tile+top-k / pair kernel of reco_tower_score_*
 *   RECO_PROF_GATHER  reco_gather_rows, reco_neumf_gather
 *   RECO_PROF_LINEAR  the GEMM kernel of reco_linear_f32 / reco_autorec_forward (and of the tower's layer-1 projections);
 *                     a split-K reduce launch is RECO_PROF_LINEAR_REDUCE
 *   RECO_PROF_CSR     the segment kernel of reco_csr_linear_f32 (its reduce launch: RECO_PROF_CSR_REDUCE)
 *   RECO_PROF_TOPK    the scan / lane kernel of reco_topk_rows* (its merge launch: RECO_PROF_TOPK_MERGE)
 *   RECO_PROF_ENCODER the fused layer-2/3 kernel of reco_mlp2_f32
 * reco_profile_collect_kinds synchronises the recorded events and returns per-launch milliseconds and kinds in launch order
 * (host pointers; kind_out nullable), then rearms.  reco_profile_collect returns the RECO_PROF_TOWER launches only.
 * ---------------------------------------------------------------------------------------- */
#define RECO_PROF_TOWER          1
#define RECO_PROF_GATHER         2
#define RECO_PROF_LINEAR         3
#define RECO_PROF_LINEAR_REDUCE  4
#define RECO_PROF_CSR            5
#define RECO_PROF_CSR_REDUCE     6
#define RECO_PROF_TOPK           7
#define RECO_PROF_TOPK_MERGE     8
#define RECO_PROF_ENCODER        9
int reco_profile_enable(int max_launches);                                    /* every kind */
int reco_profile_enable_kinds(int max_launches, unsigned kind_mask);          /* only kinds k with bit (1u << k) set: an event pair
                                                                                 costs a few microseconds of stream time */
int reco_profile_collect(float* ms_out /*host*/, int capacity, int* n_out /*host*/);
int reco_profile_collect_kinds(float* ms_out /*host*/, int32_t* kind_out /*host, nullable*/, int capacity, int* n_out /*host*/);

#ifdef __cplusplus
}
#endif
#endif /* RECO_HIP_H */
