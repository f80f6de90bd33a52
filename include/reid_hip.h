/*
 * reid_hip.h -- C-ABI of libreid_hip.so, the MI355X (gfx950) kernels of the
 * PRCV2025REID hot path.
 *
 * The reference (LingmaFuture/PRCV2025REID) is pure Python on PyTorch and has no
 * FFI of its own; every entry point below therefore cites the
 * reference *Python* code whose arithmetic it replaces.  The boundary a user of
 * the reference sees is the Python surface in prcv2025reid_amd/model.py, which
 * calls these functions through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, a negative reid_status otherwise;
 *     reid_last_error() returns a thread-local message for the last failure;
 *   - all pointers are DEVICE pointers unless named host_*; nothing is allocated
 *     or synchronised inside a call; `stream` is a hipStream_t passed as void*;
 *   - matrices are row-major with explicit leading dimensions in ELEMENTS;
 *   - bf16 = upper 16 bits of IEEE fp32 (round-to-nearest-even on conversion).
 */
#ifndef REID_HIP_H
#define REID_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    REID_OK = 0,
    REID_ERR_ARG = -1,      /* bad shape / null pointer / unsupported size */
    REID_ERR_LAUNCH = -2,   /* hipLaunch / hipGetLastError failed          */
    REID_ERR_DEVICE = -3    /* not a gfx950 device                         */
} reid_status;

/* REID_BF16 = the 16-bit format of the library flavor (see reid_flavor); REID_F16 = IEEE half WHATEVER the flavor, with saturating
 * conversion (a finite value beyond +-65504 is stored as +-65504): accepted only where an entry point says so -- for tensors that are
 * not MFMA operands, where half's 11 significant bits cost nothing (the residual-branch output between a GEMM and the add + LayerNorm). */
typedef enum { REID_BF16 = 0, REID_F32 = 1, REID_F16 = 2 } reid_dtype;

typedef enum {
    REID_ACT_NONE = 0,
    REID_ACT_GELU_ERF = 1,    /* nn.GELU() exact, models/mer_lora.py:257             */
    REID_ACT_QUICK_GELU = 2,  /* HF CLIP text MLP x*sigmoid(1.702x)                   */
    REID_ACT_RELU = 3,        /* models/model.py:42                                   */
    REID_ACT_DGELU_ERF = 4,   /* out = acc * gelu'(aux)        (backward of 1)        */
    REID_ACT_DQUICK_GELU = 5, /* out = acc * quick_gelu'(aux)  (backward of 2)        */
    REID_ACT_DRELU = 6,       /* out = acc * (aux > 0)                                */
    REID_ACT_MUL_AUX = 7,     /* out = acc * aux               (backward of 8: aux = the saved derivative)          */
    REID_ACT_GELU_ERF_DSAVE = 8 /* out = gelu(acc), C2 = gelu'(acc) instead of the pre-activation: the backward
                                 * product is then one multiply per element (7) instead of erfc + exp            */
} reid_act;

const char* reid_last_error(void);
int reid_version(void);
/* 16-bit operand format of this build: 0 = bf16 (libreid_hip.so), 1 = IEEE f16 (libreid_hip_f16.so, same sources
 * compiled with -DREID_FLAVOR_F16).  Every `bf16` / REID_BF16 in this header means "the 16-bit format of the flavor". */
int reid_flavor(void);
/* Checks that device `dev` is gfx950 (MI355X).  */
int reid_check_device(int dev);
/* Experiment knobs (tools/ and the A/B harnesses only; the product path never calls this): sets the cached value of
 * knob `name` ("GEMM_TILE", "GEMM_DBG", "GEMM_PERSIST", "GEMM_STAGGER", ... = the REID_<name> environment variables,
 * which are read ONCE per process); value -1 restores "not set" (the built-in default). */
int reid_set_knob(const char* name, int value);

/* ------------------------------------------------------------------------------------------
 * MER GEMM:  C = epilogue( A . B^T  +  A2 . B2^T + bias )      bf16 MFMA, fp32 accumulate
 *
 * Replaces MERLinear.forward (models/mer_lora.py:80-99: shared_linear(x) + lora_B(lora_A(x))*s)
 * and, with A2 = NULL, every plain nn.Linear on the path (models/clip_backbone.py:284,311;
 * HF CLIP text q/k/v/out/fc1/fc2) and their dX backward products.
 *   A  [M, K]  bf16 activations          B  [N, K]  bf16 weight (nn.Linear layout)
 *   A2 [M, *]  bf16 low-rank activations B2 [N, K2] bf16 (scaled LoRA up-projection)
 *   The low-rank pair extends the K loop by K2 (multiple of 32) columns.  With
 *   k2_group_n > 0 output columns [g*k2_group_n, (g+1)*k2_group_n) read A2 columns
 *   [g*K2, (g+1)*K2)  (fused q|k|v projection: one adapter set per projection).
 * Epilogue, in order: + bias[n]; + R (residual, r_dtype; row index m, or m % r_period when
 *   r_period > 0, e.g. the position embedding of models/clip_backbone.py:273); C2 = value
 *   (pre-activation, optional); activation `act` (for the D* forms `aux` [M, ldaux] bf16 holds
 *   the saved pre-activation); row-modality mask (mask_r > 0: column c is kept only if
 *   (c % K_mask_period)/mask_r == img_mod[m / rows_per_img], the LoRA routing of
 *   mer_lora.py:96 for a batch that mixes modalities); * alpha; store as c_dtype at row
 *   (m / c_group) * c_group_stride + m % c_group + c_row_off when c_group > 0 (patch rows ->
 *   token rows behind the CLS slot, models/clip_backbone.py:269-270), else row m.
 * Requirements: K % 64 == 0 or K % 32 == 0, K2 % 32 == 0, N % 4 == 0, 16-byte aligned rows.
 * ------------------------------------------------------------------------------------------ */
#define REID_GEMM_MAX_GROUPS 8
typedef struct {
    const void* A; const void* B; const void* A2; const void* B2;
    const float* bias;
    const void* R; const void* aux;
    void* C; void* C2;
    const int32_t* img_mod;
    int32_t M, N, K, K2;
    int32_t lda, ldb, lda2, ldb2, ldr, ldaux, ldc, ldc2;
    int32_t k2_group_n;
    int32_t act, c_dtype, c2_dtype, r_dtype;
    int32_t r_period;
    int32_t mask_r, mask_period, rows_per_img;
    int32_t c_group, c_group_stride, c_row_off;
    float alpha;
    const float* row_scale;   /* optional f32 [rows / rows_per_img]: out = R + row_scale[row / rows_per_img] * (A.B^T + bias)  (DropPath,
                                 clip_backbone.py:137-141: per-sample branch scale 0 or 1/keep); NULL = 1 */
    /* Row groups (MERLinear routing of mer_lora.py:80-99 for a batch packed modality by modality): with n_row_groups > 0 the rows
     * [row_group_end[g-1], row_group_end[g]) (row_group_end[-1] = 0; the last end must equal M) use weight matrix number
     * row_group_b[g] of a stack: B + row_group_b[g] * b_group_stride (16-bit elements).  The stack holds the MERGED weights
     * W + (alpha/r) B_mu A_mu of each modality (reid_merge_lora), so shared_linear(x) + lora_B(lora_A(x)) * s is ONE GEMM without
     * a K extension.  Row tiles never straddle two groups. */
    int32_t n_row_groups;
    int32_t row_group_end[REID_GEMM_MAX_GROUPS];
    int32_t row_group_b[REID_GEMM_MAX_GROUPS];
    int64_t b_group_stride;
} reid_gemm_args;
int reid_mer_gemm(const reid_gemm_args* host_args, void* stream);

/* ------------------------------------------------------------------------------------------
 * Reduce-over-rows GEMM:  C[P, Q] = beta*C + alpha * X[M, P]^T . Y[M, Q]   (fp32 out)
 * Weight-gradient products of the backward pass: dB = dY^T T and dA = U^T X for LoRA
 * (autograd of models/mer_lora.py:48), dW = dY^T X for unfrozen linears.
 * X, Y bf16; partial sums over row slabs are combined with fp32 atomics (C must hold
 * beta*C on entry; this call zero-fills C itself when beta == 0).
 * ------------------------------------------------------------------------------------------ */
int reid_gemm_tn(const void* X, const void* Y, float* C, int32_t M, int32_t P, int32_t Q,
                 int32_t ldx, int32_t ldy, int32_t ldc, float alpha, float beta, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (eps 1e-5; models/clip_backbone.py:35-36,73-75,82,210,280).
 *   fwd: y = (x-mean)*rstd*gamma+beta; x f32 [rows, ld]; y bf16 and/or f32; saves mean/rstd.
 *        row_index != NULL gathers x rows (CLS rows, clip_backbone.py:281: LN is per-row, so
 *        only the rows that are used get normalised).
 *   bwd: dx = dres + rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma;  dy bf16 or f32.
 *        writes dx f32 and optional bf16 copy; dgamma/dbeta (f32 [cols], atomically
 *        accumulated) only when non-NULL.  row_index != NULL scatters into rows of dx.
 *        bf16_row_scale != NULL: the 16-bit copy is dx * bf16_row_scale[row / rows_per_img] (the gradient entering a
 *        DropPath-scaled residual branch); the f32 dx is unscaled.
 *        dx_dtype: REID_F32, or REID_F16 = the residual-stream gradient (dres read AND dx written) in IEEE half, saturating:
 *        12 instead of 16 bytes per element; the caller keeps the stream inside half's range by scaling the loss.
 * ------------------------------------------------------------------------------------------ */
int reid_layernorm_fwd(const float* x, int32_t ldx, const int32_t* row_index, const float* gamma,
                       const float* beta, void* y_bf16, float* y_f32, int32_t ldy, float* mean, float* rstd,
                       int32_t rows, int32_t cols, float eps, void* stream);
/* Residual add fused into the LayerNorm that follows it (models/clip_backbone.py:76 -> :82, :83 -> next block's :73):
 *   x_out = x + row_scale[row / rows_per_img] * y        (x f32, y = the 16-bit branch output of reid_mer_gemm; row_scale NULL = 1:
 *                                                         DropPath, clip_backbone.py:137-141);  h = LayerNorm(x_out) in 16 bits,
 *   mean / rstd of x_out saved for the backward pass.  x_out may alias x.  y_dtype: REID_BF16 (the flavor's format) or REID_F16
 *   (IEEE half whatever the flavor: what reid_mer_gemm stores with c_dtype = REID_F16). */
int reid_add_layernorm_fwd(const float* x, int32_t ldx, const void* y_bf16, int32_t y_dtype, int32_t ldy, const float* row_scale, int32_t rows_per_img,
                           float* x_out, int32_t ldxo, const float* gamma, const float* beta, void* h_bf16, int32_t ldh,
                           float* mean, float* rstd, int32_t rows, int32_t cols, float eps, void* stream);
int reid_layernorm_bwd(const void* dy, int32_t dy_dtype, int32_t lddy, const float* x, int32_t ldx,
                       const int32_t* row_index, const float* gamma, const float* mean, const float* rstd,
                       const void* dres, void* dx, int32_t dx_dtype, void* dx_bf16, int32_t lddx,
                       float* dgamma, float* dbeta, int32_t rows, int32_t cols,
                       const float* bf16_row_scale, int32_t rows_per_img, void* stream);

/* ------------------------------------------------------------------------------------------
 * Patch extraction (im2col of the k=s=16 conv, models/patch_embeds.py:45-76):
 *   images f32 [n_img, 3, H, W] -> patches bf16 [n_img*(H/16)*(W/16), cin*256]; cin == 1 first
 *   averages the three channels (patch_embeds.py:63-65).  The conv itself is reid_mer_gemm.
 * cls rows: x[img*tokens + 0] = cls + pos[0]   (models/clip_backbone.py:269-273)
 * ------------------------------------------------------------------------------------------ */
int reid_patch_im2col(const float* images, void* patches, int32_t n_img, int32_t H, int32_t W,
                      int32_t patch, int32_t cin, void* stream);
int reid_cls_rows(const float* cls, const float* pos0, float* x, int32_t ldx, int32_t n_img,
                  int32_t tokens, int32_t cols, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-head attention, head_dim 64, sequences of S <= 224 tokens held on chip
 * (MERMultiheadAttention SDPA call, models/mer_lora.py:166-190, scale 1/8; HF CLIP text
 *  attention with causal + key-padding mask; nn.MultiheadAttention of FeatureFusion,
 *  models/model.py:152-155).
 *   qkv bf16 [n_seq*S, ld] with q at column 0, k at column d, v at column 2d (d = heads*64).
 *   key_mask uint8 [n_seq, S] (1 = attend) or NULL; causal != 0 adds key <= query.
 *   out bf16 [n_seq*S, ldo]; lse f32 [n_seq, heads, S] (natural-log-sum-exp of scaled scores).
 *   bwd writes dqkv bf16 in the qkv layout.
 *   q_tiles > 0: only the first q_tiles 32-row query tiles of every sequence are evaluated (all keys take part): rows
 *   of out / lse beyond them are left untouched, dout is assumed zero there and dQ is written as zero -- the last ViT block,
 *   whose output is used at the class token only (clip_backbone.py:281).
 * ------------------------------------------------------------------------------------------ */
int reid_attn_fwd(const void* qkv, int32_t ld, const uint8_t* key_mask, void* out, int32_t ldo, float* lse,
                  int32_t n_seq, int32_t S, int32_t heads, int32_t causal, int32_t q_tiles, void* stream);
int reid_attn_bwd(const void* qkv, int32_t ld, const uint8_t* key_mask, const void* out, const void* dout,
                  int32_t ldo, const float* lse, void* dqkv, int32_t lddqkv, float* delta_ws,
                  int32_t n_seq, int32_t S, int32_t heads, int32_t causal, int32_t q_tiles, void* stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise helpers.
 * ------------------------------------------------------------------------------------------ */
int reid_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
int reid_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream);
/* Batched fp32 -> bf16 repack of many small matrices that live in one fp32 arena (the LoRA parameters):
 * table[e] = {src_off, rows, cols, dst_off, dstT_off} (int64, element offsets; a negative dst offset skips that copy).
 * dst gets the same-layout bf16 copy at dst_off and the TRANSPOSED bf16 copy at dstT_off.  One launch per step. */
int reid_pack_bf16_table(const float* src, void* dst_bf16, const int64_t* table, int32_t n_entries, void* stream);
/* Adapter gradients of one MERLinear whose output width is N = 768 (autograd of LoRAAdapter, mer_lora.py:40-49), both products that
 * read the output cotangent dY in ONE pass:
 *   U[m, :]  = mask_modality(dY[m, :] . B) * scale      (16-bit [M, Rp]; column c kept iff c / mask_r == img_mod[m / rows_per_img];
 *                                                        BT = B^T [Rp, N] 16-bit; the operand of dA = U^T x, reid_gemm_tn)
 *   dB      += dY^T . T                                  (fp32 [N, Rp], atomics; T = the forward's masked x . A^T, 16-bit [M, Rp])
 * Rp must be 32.  A wider cotangent (fc1: 3072 columns) is handled as column blocks of 768, one launch each on the same stream:
 * u_mode bit 0 = add the fp32 partial sums u_partial [M, 32] of the earlier blocks to this block's, bit 1 = store the sum to
 * u_partial instead of finishing U (mask, scale, 16-bit); the last block has bit 1 clear.  u_mode = 0: a 768-column linear, u_partial unused.
 * Other shapes: reid_mer_gemm (U) + reid_gemm_tn (dB).
 * T must be modality-masked (T[m, c] = 0 unless c / mask_r == img_mod[m / rows_per_img]) -- it is how reid_mer_gemm's mask epilogue
 * produces it: with rows_per_img >= 32 and mask_r a divisor of 16 one workgroup streams one image (one modality) and touches only the
 * aligned 16-column group of adapter columns that holds the modality's; elsewhere dB stays as it was. */
int reid_lora_bwd_fused(const void* dY, int32_t lddy, const void* T, int32_t ldt, const void* BT, int32_t ldbt, void* U, int32_t ldu,
                        float* dB, int32_t lddb, const int32_t* img_mod, int32_t rows_per_img, int32_t mask_r, int32_t M, int32_t N,
                        int32_t Rp, float scale, float* u_partial, int32_t u_mode, void* stream);
/* The other adapter gradient of a MERLinear, dA += U^T . X (autograd of lora_A, mer_lora.py:40-49), as one pass over the linear's INPUT
 * X [M, K] (16-bit; K a multiple of 768): dA[g Rp + c, k] += sum_m U[m, g Rp + c] X[m, k] for the n_groups (1, or 3 for the fused q|k|v
 * projection) adapter groups of Rp = 32 columns of U [M, n_groups Rp] (16-bit, modality-masked as reid_lora_bwd_fused writes it).
 * dA: fp32 [n_groups Rp, K] (row stride ldda), accumulated with atomics.  One workgroup per (image, 768-column block): rows_per_img >= 32,
 * mask_r a divisor of 16; other shapes: reid_gemm_tn(U, X). */
int reid_lora_da_fused(const void* X, int32_t ldx, const void* U, int32_t ldu, float* dA, int32_t ldda, const int32_t* img_mod,
                       int32_t rows_per_img, int32_t mask_r, int32_t M, int32_t K, int32_t Rp, int32_t n_groups, void* stream);

/* Merged MER-LoRA weights (mer_lora.py:80-99): for every table entry e and modality mu < nmod
 *     W_eff[e][mu] = W_e + scaling * Bcat_e[:, mu r : (mu+1) r] . Acat_e[g Rp + mu r : g Rp + (mu+1) r, :]      (16-bit, rounded once)
 * and its transpose -- the operands of reid_mer_gemm's row-group form.  table[e] = {W pointer (f32 [N, K] contiguous), arena
 * offset of Acat [G*Rp, K], arena offset of Bcat [N, Rp], destination offset of W_eff [nmod][N][K], destination offset of
 * W_eff^T [nmod][K][N] (negative: skip), N, K, G} (int64; offsets in elements; N, K, N/G multiples of 64; G = projections fused
 * along N, each with its own adapter rows g Rp ...).  max_tiles >= (N/64)(K/64) of every entry.  One launch per optimizer step.
 * Ranks: r <= 64 per modality and nmod * r <= Rp (up to 64 adapter rows are staged at once, more one modality at a time: same results). */
int reid_merge_lora_table(const int64_t* table, int32_t n_entries, int32_t max_tiles, const float* arena, void* weff,
                          int32_t Rp, int32_t r, int32_t nmod, float scaling, void* stream);
/* dst[r, :] = src[index[r], :] (f32, cols % 4 == 0);  scatter_add is the adjoint. */
int reid_gather_rows_f32(const float* src, int32_t lds, const int32_t* index, float* dst, int32_t ldd,
                         int32_t rows, int32_t cols, void* stream);
/* Text-tower input (HF CLIPTextEmbeddings, models/clip_backbone.py:307): out[b*T + t, :] = tok[ids[b, t], :] + pos[t, :]
 * (f32 tables [vocab, D] / [>= T, D], ids int64 [B, T]; ids outside the vocabulary are clamped). */
int reid_embed_tokens(const float* tok, const float* pos, const int64_t* ids, float* out, int32_t B, int32_t T, int32_t D,
                      int32_t vocab, void* stream);
/* out[index[r], :] += src[r, :] (f32; rows with an index outside [0, out_rows) are skipped): gradient of an embedding
 * lookup (HF CLIPTextEmbeddings.token_embedding) when the text tower trains. */
int reid_scatter_add_rows_f32(const float* src, int32_t lds, const int32_t* index, float* out, int32_t ldo,
                              int32_t rows, int32_t cols, int32_t out_rows, void* stream);

/* ------------------------------------------------------------------------------------------
 * BN-neck (BNNeck.forward, models/model.py:208-224): BatchNorm1d(D) -> 8*L2-normalise.
 *   training != 0: batch statistics over `rows` (biased variance), running stats updated with
 *   momentum 0.1 / unbiased variance; else running statistics.  The classifier GEMM that
 *   follows is reid_mer_gemm on the bf16 copy.
 *   With ext_sum/ext_sqsum/ext_count non-NULL the statistics are taken from those buffers
 *   (data-parallel SyncBN: the caller all-reduces them between the two phases).
 *   fwd outputs: y f32 [rows, D] (norm 8 per row), y_bf16 optional, saved xhat-free state:
 *   mean[D], invstd[D], rnorm[rows].
 * ------------------------------------------------------------------------------------------ */
int reid_bnneck_stats(const float* x, int32_t ldx, int32_t rows, int32_t D, float* sum, float* sqsum, void* stream);
int reid_bnneck_fwd(const float* x, int32_t ldx, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, const float* sum, const float* sqsum, float count, int32_t training,
                    float* y, void* y_bf16, int32_t ldy, float* mean, float* invstd, float* rnorm,
                    int32_t rows, int32_t D, float eps, float momentum, float scale, void* stream);
/* bwd phase 1: per-row L2-normalise backward -> dz (grad wrt BN output), and column sums
 *   sum_dz[D], sum_dz_xhat[D] (all-reduced by the caller under data parallelism);
 * bwd phase 2: dx = gamma*invstd*(dz - sum_dz/count - xhat*sum_dz_xhat/count); dgamma = sum_dz_xhat; dbeta = sum_dz */
int reid_bnneck_bwd_p1(const float* dy, int32_t lddy, const float* x, int32_t ldx, const float* gamma,
                       const float* beta, const float* mean, const float* invstd, const float* rnorm,
                       float* dz, float* sum_dz, float* sum_dz_xhat, int32_t rows, int32_t D, float scale,
                       void* stream);
int reid_bnneck_bwd_p2(const float* dz, const float* x, int32_t ldx, const float* gamma, const float* mean,
                       const float* invstd, const float* sum_dz, const float* sum_dz_xhat, float count,
                       int32_t training, float* dx, int32_t lddx, int32_t rows, int32_t D, void* stream);

/* ------------------------------------------------------------------------------------------
 * Cross entropy with label smoothing (nn.CrossEntropyLoss(label_smoothing=0.1), models/model.py:290,
 * 529-549): mean over rows with valid[r] != 0 and 0 <= label < C.
 *   loss_sum[0] += sum of row losses, loss_sum[1] += number of rows used (both f32, zeroed by caller);
 *   dlogits = (softmax - smoothed one-hot) * grad_scale for used rows, 0 otherwise (grad_scale =
 *   ce_weight / global_count, read from device memory so no host sync is needed).
 * ------------------------------------------------------------------------------------------ */
int reid_ce_ls_fwd(const float* logits, int32_t ld, const int64_t* labels, const uint8_t* valid, int32_t rows,
                   int32_t C, float smoothing, float* row_loss, float* loss_sum, void* stream);
int reid_ce_ls_bwd(const float* logits, int32_t ld, const int64_t* labels, const uint8_t* valid, int32_t rows,
                   int32_t C, float smoothing, const float* grad_scale, float* dlogits, int32_t lddl, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused SDM loss (sdm_loss_stable, models/sdm_loss.py:13-149) for ALL modality pairs of a step (models/model.py:586-622
 * calls it once per non-vis modality against vis):
 *   q [P*N, D] f32 = the P query sides stacked (rows [p*N, (p+1)*N) = pair p), g [Mg, D] f32 = the shared vis side;
 *   q_label [N] (the batch labels, shared by the pairs), g_label [Mg]; q_valid [P*N] / g_valid [Mg] uint8 or NULL (rows that
 *   take part, models/model.py:570,595); positives y[i,j] = (q_label[i] == g_label[j]) (models/model.py:605);
 *   tau clamped to [0.15, 0.5] (:28); unit vectors with eps 1e-8 (:31-32); S = q^ g^T / tau clamped to +-20 (:86,94).
 *   result[2p]   = 0.5 * (mean over rows with a positive of CE(q->g) + mean over such columns of CE(g->q))  (:34-70,121-123)
 *   result[2p+1] = 1 if pair p has any positive (it contributes to the modality mean), else 0 and result[2p] = 0 (:105-106).
 * S is never materialised: each tile of it exists only in MFMA accumulators (v_mfma_f32_32x32x2_f32 on the fp32 unit
 * vectors, exact fp32) and is reduced to per-tile row / column partial sums that a second small launch adds in a fixed order;
 * the backward pass recomputes the tiles.  No N x Mg array exists in fwd or bwd.
 *   ws (reid_sdm_ws_floats(P, N, Mg, D) floats, kept from fwd to bwd):
 *     unit vectors (P N + Mg) D | {lse, #pos} per row and column 2 P (N + Mg) | 4 P sums | P (N + Mg) loss terms |
 *     max( per-tile partials 4 (tiles_n P N + tiles_m P Mg),  gradient accumulators (P N + Mg) D )      [tiles of 64 or 128]
 *   bwd: dq [P*N, D] and dg [Mg, D] (+= into f32) for the upstream gradients gscale[p] (device array [P]).
 * Requirements: D % 32 == 0, 32 <= D <= 1024.
 * ------------------------------------------------------------------------------------------ */
int reid_sdm_fwd(const float* q, int32_t ldq, const float* g, int32_t ldg, const int64_t* q_label,
                 const int64_t* g_label, const uint8_t* q_valid, const uint8_t* g_valid, int32_t P, int32_t N, int32_t Mg,
                 int32_t D, float tau, float* ws, float* result, void* stream);
int reid_sdm_bwd(const float* q, int32_t ldq, const float* g, int32_t ldg, const int64_t* q_label,
                 const int64_t* g_label, const uint8_t* q_valid, const uint8_t* g_valid, int32_t P, int32_t N, int32_t Mg,
                 int32_t D, float tau, float* ws, const float* gscale, float* dq, int32_t lddq, float* dg,
                 int32_t lddg, void* stream);
int64_t reid_sdm_ws_floats(int32_t P, int32_t N, int32_t Mg, int32_t D);

/* ------------------------------------------------------------------------------------------
 * Retrieval (train.py:499 + :463; tools/eval_mm_protocol.py:50-53,401-423,622-625):
 *   sim = Q . G^T on L2-normalised rows, per-query top-k in (score desc, index asc) order.
 *   Q [Nq, D], G [Ng, D] bf16 copies drive a tiled MFMA GEMM with an on-chip candidate filter;
 *   survivors are re-scored in fp32 from Qf/Gf so the order equals the fp32 oracle's.
 *   exclude_q/exclude_g (int32 ids, or NULL): entries with equal non-negative id get -1e9
 *   (same-image mask, eval_mm_protocol.py:421-422).
 *   out_idx int32 [Nq, k], out_score f32 [Nq, k].  ws: reid_topk_ws_bytes().
 *   Nq <= 128 with k <= 16, D = 256 or 512 and a gallery of at least 256 k rows: the candidate filter is ONE pass of a query-resident
 *   scan kernel (the 16-bit gallery streamed once, the queries as MFMA operands in registers, the bar of each query taken from the
 *   scan's own running maxima) instead of the sample / threshold / tiled filter launches -- same candidate lists contract, same
 *   results, same workspace.  A query whose candidate list overflows is marked out_idx[q, 0] = -2 for reid_cosine_topk_exact(_slots).
 * ------------------------------------------------------------------------------------------ */
int64_t reid_topk_ws_bytes(int32_t Nq, int32_t Ng, int32_t k);
/* 1 when reid_cosine_topk takes the query-resident scan for this shape (then also the faster form for 2-4 queries). */
int32_t reid_topk_scan_ok(int32_t Nq, int32_t Ng, int32_t D, int32_t k);
int reid_cosine_topk(const void* Q_bf16, const void* G_bf16, const float* Qf, const float* Gf,
                     int32_t Nq, int32_t Ng, int32_t D, int32_t k, const int32_t* exclude_q,
                     const int32_t* exclude_g, void* ws, int32_t* out_idx, float* out_score, void* stream);
/* Exact fp32 pass for the queries reid_cosine_topk flagged with out_idx[q][0] == -2 (candidate list
 * overflow, e.g. thousands of near-duplicates); scratch holds Nq*Ng floats; other queries are untouched. */
int reid_cosine_topk_exact(const float* Qf, const float* Gf, int32_t Nq, int32_t Ng, int32_t D, int32_t k,
                           const int32_t* exclude_q, const int32_t* exclude_g, float* scratch, int32_t* out_idx,
                           float* out_score, void* stream);
/* The same exact pass without reading the flags back: the flagged queries are compacted on the device into `slots` (int32 [1 + n_slots]:
 * count, then query ids) and EVERY listed query is resolved by the one call -- list entry e uses scratch row e (scratch: n_slots * Ng
 * floats; touched only for flagged queries).  n_slots >= Nq guarantees that no -2 marker survives the call; with a smaller list the
 * queries beyond it keep their marker and slots[0] (> n_slots) says so. */
int reid_cosine_topk_exact_slots(const float* Qf, const float* Gf, int32_t Nq, int32_t Ng, int32_t D, int32_t k,
                                 const int32_t* exclude_q, const int32_t* exclude_g, int32_t n_slots, int32_t* slots,
                                 float* scratch, int32_t* out_idx, float* out_score, void* stream);
/* The reference's one-query-at-a-time form (tools/eval_mm_protocol.py:401-455: sim = q @ G.T; argsort) for a handful of
 * queries: ONE pass over the fp32 gallery (Ng*D*4 bytes, HBM-bound) instead of the batched pipeline's launch chain.  Same
 * fp32 scores and the same (score desc, index asc) lists as reid_cosine_topk.  Allowed when reid_topk_stream_ok() returns 1
 * (Nq <= 4, k <= 32, D a multiple of 256 up to 1024).  ws: reid_topk_stream_ws_bytes(k) bytes owned by this entry point: ZERO before the
 * first call and not written by anyone else between calls (its last 256 bytes hold the arrival counter of the one-query form, in which
 * the workgroup whose partial list arrives last merges all lists -- one launch per call; every call leaves the counter at zero). */
int32_t reid_topk_stream_ok(int32_t Nq, int32_t Ng, int32_t D, int32_t k);
int64_t reid_topk_stream_ws_bytes(int32_t k);
int reid_cosine_topk_stream(const float* Qf, const float* Gf, int32_t Nq, int32_t Ng, int32_t D, int32_t k,
                            const int32_t* exclude_q, const int32_t* exclude_g, void* ws, int32_t* out_idx,
                            float* out_score, void* stream);
/* fp32 C[M,N] = act(alpha * op(A).op(B) + bias[n]) + beta*C on the vector ALU with arbitrary element strides
 * (A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]); the small exact GEMMs of the head
 * (models/model.py:57-77,152-162) and of the SDM loss. */
int reid_sgemm(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int64_t sam, int64_t sak,
               int64_t sbk, int64_t sbn, int32_t ldc, float alpha, float beta, const float* bias, int32_t act,
               void* stream);
/* ------------------------------------------------------------------------------------------
 * Small fp32 pieces of the head: SemanticDisentanglementModule.forward (models/model.py:57-77) and
 * FeatureFusion.forward (:113-183) on [B,512] / [B,M<=8,512] tensors.
 *   reid_eltwise_f32: op 0 out=x+alpha*y, 1 relu(x), 2 relu' (x pre-activation, y = dy), 3 erf-GELU(x),
 *                     4 y*GELU'(x), 5 x*y, 6 nan_to_num(x, 0, 1e4, -1e4) (model.py:165)
 *   reid_small_attn_fwd/bwd: softmax(q k^T / 8 + key-padding mask) v over S <= 8 tokens, head_dim 64, fp32
 *                     (nn.MultiheadAttention, model.py:152-155); qkv [n_seq*S, ld] = q|k|v; probs [n_seq, heads, 8, 8] saved;
 *                     drop (optional, same layout as probs): attention-dropout multipliers 0 or 1/keep applied to the
 *                     probabilities after the softmax (nn.MultiheadAttention(dropout=0.1), model.py:35,95)
 *   reid_masked_mean: out[b,:] = sum_m mask[b,m] x[b,m,:] / max(sum_m mask[b,m], 1) (model.py:168-178); backward != 0:
 *                     x = dout [B,D], out = dx [B,M,D]
 * ------------------------------------------------------------------------------------------ */
int reid_eltwise_f32(int32_t op, const float* x, const float* y, float* out, int64_t n, float alpha, void* stream);
int reid_small_attn_fwd(const float* qkv, int32_t ld, const uint8_t* key_mask, const float* drop, float* out, int32_t ldo,
                        float* probs, int32_t n_seq, int32_t S, int32_t heads, void* stream);
int reid_small_attn_bwd(const float* qkv, int32_t ld, const float* probs, const float* drop, const float* dout, int32_t ldo,
                        float* dqkv, int32_t lddqkv, int32_t n_seq, int32_t S, int32_t heads, void* stream);
int reid_masked_mean(const float* x, const float* mask, float* out, int32_t B, int32_t M, int32_t D, int32_t backward,
                     void* stream);
/* L2-normalise rows (F.normalize, train.py:442): x f32 [rows, D] -> y f32 and/or bf16. */
int reid_l2norm_rows(const float* x, int32_t ldx, float* y, void* y_bf16, int32_t ldy, int32_t rows, int32_t D,
                     float eps, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer step of the training driver: train.py:85-96 (_sanitize_grads),
 * :975-1047 (gradient norm, adaptive / fixed clip, optimizer.step()) with torch.optim.AdamW semantics
 * (decoupled weight decay, bias correction, no amsgrad), as three sync-free launches over a DEVICE table of
 * reid_opt_entry records (one per trainable flat fp32 buffer, 16-byte aligned; g == NULL skips the entry):
 *   reid_opt_sumsq  non-finite gradient entries := 0 in place; per-workgroup partial sums of g^2 and counts -> ws
 *   reid_opt_clip   state[1] = ||g||_2 (fixed summation order), state[3] = max_norm, state[2] = clip coefficient
 *                   min(1, max_norm / (||g|| + 1e-6)) (torch.nn.utils.clip_grad_norm_), state[4] = #non-finite;
 *                   adaptive != 0: train.py:981-1001 -- when record != 0 the norm is appended to the history
 *                   (state[5] = count, state[6..15] = last ten), max_norm = min(3, max(0.5, 1.15 * percentile70(last ten)))
 *                   once more than ten norms were recorded, else 1.0; adaptive == 0: max_norm = fixed_max_norm
 *   reid_opt_adamw  p, exp_avg, exp_avg_sq updated with gradient g * (*coef) (coef may be NULL = 1); step counts from 1;
 *                   step == 0: the step counter and bias corrections live in state[16..18] and advance on the device, and
 *                   reid_opt_clip's record < 0 means "record when (device batch counter state[19]) % (-record) == 0" -- the
 *                   forms a captured HIP graph replays;
 *                   zero_grad != 0 clears g in the same pass (optimizer.zero_grad at the next accumulation window)
 * ------------------------------------------------------------------------------------------ */
typedef struct reid_opt_entry {
    float* p; float* g; float* exp_avg; float* exp_avg_sq;
    int64_t n;
    float lr, weight_decay;
} reid_opt_entry;
int32_t reid_opt_entry_bytes(void);
int32_t reid_opt_ws_floats(int32_t n_entries);
int32_t reid_opt_state_floats(void);
int reid_opt_sumsq(const void* table, int32_t n_entries, float* ws, void* stream);
int reid_opt_clip(const float* ws, int32_t n_entries, float* state, int32_t adaptive, float fixed_max_norm,
                  int32_t record, void* stream);
int reid_opt_adamw(const void* table, int32_t n_entries, const float* coef, float beta1, float beta2, float eps,
                   int32_t step, int32_t zero_grad, float* state, void* stream);

/* ------------------------------------------------------------------------------------------
 * AP / CMC from fp32 similarity rows: the metric half of rank_and_metrics,
 * tools/eval_mm_protocol.py:401-469 (and _reid_map, train.py:450-479) without sorting the gallery.
 *   scores  f32 [nq, ld] (ld % 4 == 0, 16-byte aligned): cosine similarities of nq queries against Ng gallery rows
 *   g_pid   i32 [Ng]; g_img i32 [Ng] image ids (NULL or -1 = none)
 *   q_pid   i32 [nq]; q_slot i32 [nq] = row of the query's pid in the CSR below, -1 if the pid has no gallery row
 *   q_excl  i32 [nq, 4] image ids whose gallery rows are ignored for that query (-1 = unused; NULL = no masking)
 *   csr_off i32 [n_pid + 1], csr_idx i32 [Ng]: gallery rows grouped by pid
 * outputs per query: ap f64 (0 when there is no positive), rank1 i32 = rank of the best positive (CMC@k = rank1 <= k),
 *   npos i32 = number of unmasked positives (0: the reference skips the query; -1: more than 8192, not evaluated).
 *   max_pos = length of the longest CSR row (sizes the LDS list of positives).
 * Ranking rule: score descending, gallery index ascending on ties (a stable descending argsort); masked rows rank last.
 * ------------------------------------------------------------------------------------------ */
int reid_rank_metrics(const float* scores, int64_t ld, const int32_t* g_pid, const int32_t* g_img,
                      const int32_t* q_pid, const int32_t* q_slot, const int32_t* q_excl, const int32_t* csr_off,
                      const int32_t* csr_idx, int32_t nq, int32_t Ng, int32_t max_pos, double* ap, int32_t* rank1,
                      int32_t* npos, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* REID_HIP_H */
