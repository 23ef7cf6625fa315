/* C ABI of libfourm_hip.so — the MI355X (gfx950) kernels behind the 4M training hot path.
 *
 * The upstream project (apple/ml-4m) is pure PyTorch and has no FFI of its own; every entry point
 * below therefore names the chain of upstream torch ops it replaces (file:line under
 * /root/reference).  INTEGRATION.md shows the ctypes binding a maintainer would add upstream.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; the caller owns all memory,
 *     including workspaces; no function allocates, frees or synchronises;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream), 0 = null stream;
 *   - bf16 buffers are raw uint16 bit patterns; leading dimensions (ld*) are in ELEMENTS;
 *   - return value: 0 = launched, <0 = rejected (-1 bad argument, -2 launch failure); the reason is
 *     available from fm_last_error() (thread-local, valid until the next call on that thread);
 *   - functions are re-entrant; one host thread per device is the intended use.
 */
#ifndef FOURM_HIP_H
#define FOURM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FM_ABI_VERSION 8
int fm_abi_version(void);
const char* fm_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM (bf16 operands, fp32 accumulate on the MFMA matrix cores)
 * ---------------------------------------------------------------------------------------------- */
enum fm_epilogue {
    FM_EPI_BF16 = 0,     /* out(bf16)  = bf16(acc + bias)                                            */
    FM_EPI_GELU = 1,     /* out(bf16)  = gelu(bf16(acc + bias)); out2(bf16, optional) = pre-activation */
    FM_EPI_RESIDUAL = 2, /* out(f32)   = res(f32) + bf16(acc + bias)      (out may alias res)         */
    FM_EPI_SWIGLU = 3,   /* W -> g, W2 -> u: out(bf16)[m][h] = silu(g)*u; out2(bf16, optional)[m][h] = g, [m][Hp+h] = u */
    FM_EPI_F32 = 4,      /* out(f32)   = acc + bias (+ res(f32) when given; no rounding)              */
    FM_EPI_TANH = 5,     /* out(bf16)  = tanh(bf16(acc + bias))                                       */
    FM_EPI_SWIGLU_BWD = 6, /* acc = d(silu(g)*u); res(bf16,(M,2*Hp)) = g|u saved by FM_EPI_SWIGLU;
                              out(bf16,(M,2*Hp)) = dg | du.  Columns [roundup4(N), Hp) of each half are not written */
    FM_EPI_GELU_BWD = 7  /* acc = d(gelu(pre)); res(bf16) = pre saved by FM_EPI_GELU; out(bf16) = d(pre) */
};

typedef struct fm_gemm_group {  /* one entry per row segment (modality) in grouped mode */
    const void* W;              /* NT: weight-like operand of this group; TN: unused               */
    void* out;                  /* TN: fp32 accumulator of this group;    NT: unused               */
    int32_t N, K, ldw, pad_;
} fm_gemm_group;

/* out[m][n] = sum_k X[m][k] * W[n][k]  (+ epilogue).
 * Replaces nn.Linear forward under bf16 autocast — fourm/models/fm_utils.py:116-126,137-144,155-157,
 * 190-194 — and, with a transposed weight shadow, its input gradient.
 * K % 64 == 0 (zero padded), ldw/ldx % 8 == 0, ldo % 4 == 0 and ldo >= roundup4(N): a lane stores 4
 * consecutive features, so for N % 4 != 0 the columns [N, roundup4(N)) of out receive don't-care values
 * (FM_EPI_SWIGLU writes zeros there instead).
 * Grouped mode (groups != NULL): rows of X are segmented in FM_SEG_ROWS-row tiles, tile_group[tile] selects
 * the group (or -1 = skip); W/N/K/ldw come from the group record, max_N bounds the launch and K (an upper bound
 * of the groups' K, 0 = unknown) picks the tile configuration. */
typedef struct fm_gemm_nt_args {
    const void* W; const void* W2; const void* X;
    void* out; void* out2; const void* res; const void* bias; const void* bias2;
    int32_t M, N, K, ldw, ldx, ldo, ldo2, ldr, Hp, epilogue;
    const fm_gemm_group* groups; const int32_t* tile_group; int32_t max_N, pad_;
    /* Row range in DEVICE memory (ABI 7; both or neither, NULL = rows [0, M)): the launch covers rows [*row0_dev, *row0_dev +
     * roundup(*m_dev, FM_SEG_ROWS)) of X / out - one modality head's segment as fm_segment_rows laid it out (seg_start[h], seg_count[h];
     * pad rows zero) - with M an upper bound.  One dense launch per head then replaces the grouped launch without a host read of the
     * row counts (FourM.forward_logits, fm.py:521-545).  FM_EPI_BF16 without bias, ldo % 64 == 0, out 128-byte aligned, N % 8 == 0,
     * K % 64 == 0; other arguments are rejected. */
    const int32_t* m_dev; const int32_t* row0_dev;
    /* Scratch for split-K (ABI 8; optional, NULL = never split): a dense FM_EPI_BF16 launch whose output has too few tiles for the chip
     * (small M x N, long K: the convolutions of the DiVAE UNet at its coarse levels) is cut into up to 16 K-slices that write fp32 partial
     * tiles here (slices x M x roundup4(N) x 4 bytes are needed; a smaller buffer means fewer slices or none), followed by one reduction
     * pass (sum of the slices + bias -> bf16).  The buffer is dead when the call's work has run; calls on one stream may share it. */
    void* splitk_ws; int64_t splitk_ws_bytes;
    /* Implicit 3 x 3 convolution (ABI 8; conv_C = 0: off).  X is then a bf16 feature map in rows, (B, conv_H >> conv_up, conv_W >> conv_up, conv_C)
     * with row stride ldx, read at (y >> conv_up, x >> conv_up) (conv_up = 1: the nearest x2 up-sampling in front of the convolution); the launch
     * computes out[(b, oy, ox)][n] = sum over tap, c of in[b][oy stride + tap / 3 - 1][ox stride + tap % 3 - 1][c] W[n][tap conv_C + c] (zero
     * padding) on a (conv_Ho, conv_Wo) output grid: M = B conv_Ho conv_Wo, K = 9 conv_C, conv_C % 64 == 0, conv_stride 1 | 2 - exactly
     * fm_unet_im2col + the plain launch (bit-identical), without writing the 9 x expanded rows.  FM_EPI_BF16 (may be split over K like any
     * small launch) or FM_EPI_F32, dense, no residual / second output. */
    int32_t conv_C, conv_H, conv_W, conv_Ho, conv_Wo, conv_stride, conv_up, conv_pad_;
} fm_gemm_nt_args;
int fm_gemm_nt(const fm_gemm_nt_args* args, void* stream);
/* tile configuration of fm_gemm_nt, for A/B measurements (table in csrc/gemm.hip): low byte 0-8 = fixed
 * configuration, 9 = automatic (default); +256 = raise the wave priority around the MFMA clusters;
 * bits 16-19 / 20-23 (when non-zero) = the configurations "automatic" picks for K < 1536 / K >= 1536 (and reading
 * epilogues); bits 24-27 (when non-zero) = its choice for the SwiGLU epilogue. */
void fm_set_gemm_nt_config(int cfg);
int fm_get_gemm_nt_config(void);
/* Compute units the persistent GEMM grids leave free (default 0; env FOURM_RESERVED_CUS): set by the data-parallel wrapper so that
 * RCCL's kernels find CUs while a gradient bucket is exchanged under the backward GEMMs (upstream: DDP overlap, run_training_4m.py:512). */
void fm_set_reserved_cus(int n);
int fm_get_reserved_cus(void);

/* out[n][k] += sum_r A[r][n] * B[r][k]   (fp32 atomic accumulation, reduction split over blocks).
 * Replaces the weight-gradient matmul autograd runs for nn.Linear (dW = dY^T X).
 * Any R >= 1: reduction rows >= R are masked inside the kernel (A / B need no padding rows); lda/ldb % 8 == 0; a_cols/b_cols = number of
 * readable columns of A/B (0 = lda/ldb).  splits <= 0 picks a split count; force_tr: -1 = library
 * default, 0 = 2-byte LDS gathers, 1 = ds_read_b64_tr_b16 transpose reads.
 * Grouped mode: group g reduces rows [seg_start[g], seg_start[g] + roundup64(seg_count[g])) into
 * groups[g].out with N = groups[g].N. */
typedef struct fm_gemm_tn_args {
    const void* A; const void* B; void* out;
    int32_t R, N, K, lda, ldb, ldo, a_cols, b_cols, splits, force_tr;
    const fm_gemm_group* groups; const int32_t* seg_start; const int32_t* seg_count;
    int32_t n_groups, max_N, max_R, pad_;
} fm_gemm_tn_args;
int fm_gemm_tn(const fm_gemm_tn_args* args, void* stream);
/* A list of dense weight-gradient GEMMs in ONE launch (all dW of a transformer layer: fourm/models/fm_utils.py Block /
 * DecoderBlock backward as autograd runs it, one addmm per nn.Linear): job i does out_i[n][k] += sum_{r<R_i} A_i[r][n] * B_i[r][k]
 * with fm_gemm_tn's operand rules.  The tiles of all jobs share one grid of one workgroup per CU: whole-row reductions while a
 * full round of tiles remains, one main + tail cut of the last partial round (csrc/gemm.hip, gemm_tn_multi_kernel).
 * n_jobs <= FM_TN_MAX_JOBS; the job array is read on the host at call time. */
#define FM_TN_MAX_JOBS 16
typedef struct fm_gemm_tn_job {
    const void* A; const void* B; void* out;
    int32_t R, N, K, lda, ldb, ldo, a_cols, b_cols;
} fm_gemm_tn_job;
int fm_gemm_tn_multi(const fm_gemm_tn_job* jobs, int n_jobs, void* stream);
/* tile schedule of fm_gemm_tn (table in csrc/gemm.hip): 0 = K-step 32, two workgroups per CU; 1 = K-step 64,
 * one workgroup per CU, ping-pong schedule (half the workgroups -> half the atomic epilogue traffic); 3 = as 1, and
 * fm_gemm_tn_multi falls back from its 256 x 256 tiles to 128 x 256. */
void fm_set_gemm_tn_config(int cfg);
int fm_get_gemm_tn_config(void);
void fm_set_tn_transpose_read(int on);
int fm_get_tn_transpose_read(void);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (eps inside the sqrt, fp32 statistics) — F.layer_norm as used by
 * fourm/models/fm_utils.py:93-108 (and nn.LayerNorm in the *_gelu factories, fm.py:853).
 * x: f32 (R, D); y: bf16 (default) or f32; b may be NULL; mean/rstd (R) f32 optional outputs.
 * row_map (optional): y row r is written to row row_map[r] (skipped when negative).
 * D % 4 == 0, D <= 2048. */
int fm_layernorm_fwd(const void* x, int ldx, const void* w, const void* b, void* y, int ldy, int y_is_f32,
                     void* mean, void* rstd, const int32_t* row_map, int R, int D, float eps, void* stream);
/* The same with the residual add in front: row = x + delta (delta: bf16 (R, D), the output of the Linear that precedes the norm,
 * fm_utils.py:332-333, 363-365); the sum is also written to x_out (f32, the new residual stream; may not alias x). */
int fm_layernorm_fwd_res(const void* x, int ldx, const void* delta, int ldd, void* x_out, int ldxo, const void* w, const void* b,
                         void* y, int ldy, int y_is_f32, void* mean, void* rstd, const int32_t* row_map, int R, int D, float eps,
                         void* stream);
/* dx(f32) = dres + LN'(dy);  dw += sum_r dy*xhat;  db += sum_r dy  (fp32 atomics; dw/db may be NULL).
 * dy: bf16, read at row dy_row_map[r] when a map is given (negative = zero gradient).
 * dres (optional, may alias dx) is the residual-stream gradient flowing around the norm.
 * dx_bf16 (optional) receives a bf16 copy of dx (operand of the next GEMM). */
int fm_layernorm_bwd(const void* dy, int lddy, const int32_t* dy_row_map, const void* x, int ldx, const void* w,
                     const void* mean, const void* rstd, const void* dres, void* dx, int lddx, void* dx_bf16,
                     int lddxbf, void* dw, void* db, int R, int D, void* stream);
/* The same for a bias-free norm whose bf16 output h = bf16(xhat * w) (R, D; what fm_layernorm_fwd wrote, kept for the weight-gradient
 * GEMM of the Linear it feeds) is still in memory: xhat = h / w is rebuilt from the 2-byte h instead of the 4-byte x (14 instead of
 * 16 bytes per element; xhat carries h's bf16 rounding, relative 2^-9).  x is read only for 4-column chunks holding a weight of
 * magnitude < 1e-20.  No row map, no bias gradient. */
int fm_layernorm_bwd_h(const void* dy, int lddy, const void* h, int ldh, const void* x, int ldx, const void* w,
                       const void* mean, const void* rstd, const void* dres, void* dx, int lddx, void* dx_bf16,
                       int lddxbf, void* dw, int R, int D, void* stream);

/* Per-head LayerNorm of q / k for qk_norm models (NormAttention / NormCrossAttention q_norm, k_norm:
 * fourm/models/fm_utils.py:235-236,244-245,279-280,290-291).  x, y: bf16 rows of H heads x 64 contiguous features
 * (row stride ldx / ldy, e.g. the q or k column block of the fused qkv buffer); w, b: f32 (64) (b may be NULL);
 * stats: f32 (R*H, 2) = (mean, rstd) kept for the backward.  fp32 arithmetic, bf16 result (autocast). */
int fm_headnorm_fwd(const void* x, int ldx, const void* w, const void* b, void* y, int ldy, void* stats, int R, int H, float eps,
                    void* stream);
/* dx (bf16) = LayerNorm backward of dy w.r.t. the head vectors; dw / db (f32 (64), may be NULL) are accumulated. */
int fm_headnorm_bwd(const void* dy, int lddy, const void* x, int ldx, const void* w, const void* stats, void* dx, int lddx, void* dw,
                    void* db, int R, int H, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Masked multi-head attention, head_dim 64 — fourm/models/fm_utils.py:160-180 (Attention) and
 * :197-219 (CrossAttention).  Element (b, t, h, d) of Q lives at Q[(b*Nq + t)*ldq + h*64 + d]
 * (K, V with Nk; O, dO, dQ like Q; dK, dV like K), so q/k/v can alias one fused qkv buffer.
 * A blocked score is replaced by -finfo(bf16).max (upstream semantics: fully blocked rows attend
 * uniformly).  stat_m / stat_l: (B, H, Nq) f32 row max and row sum, written by fwd, read by bwd. */
enum fm_mask_kind {
    FM_MASK_NONE = 0,
    FM_MASK_KEYPAD = 1,   /* kpad (B, Nk) uint8, 1 = blocked                  — fm.py:388                   */
    FM_MASK_DECODER = 2,  /* k >= cs[b][q] (or k > q if causal)  ||  modq[b][q] != modk[b][k] — fm.py:440-475 */
    FM_MASK_DENSE = 3     /* dense (B, Nq, Nk) uint8, 1 = blocked                                            */
};
typedef struct fm_attn_args {
    const void* Q; const void* K; const void* V; void* O;
    void* stat_m; void* stat_l;
    int32_t ldq, ldk, ldv, ldo;
    int32_t B, H, Nq, Nk, head_dim, mask_kind;
    float scale; int32_t causal;
    const void* kpad; const int32_t* cs; const int16_t* modq; const int16_t* modk; const void* dense;
    /* backward only */
    const void* dO; void* dQ; void* dK; void* dV;
    int32_t lddo, lddq, lddk, lddv;
    int32_t force_tr;         /* -1 = library default, 0 = 2-byte LDS gathers, 1 = transpose reads */
    int32_t kv_batch_rows;    /* forward only: rows between two samples in K / V (0 = Nk).  A K/V cache of capacity T holds sample b's
                                 keys at rows [b*T, b*T + Nk): incremental decoding attends to the Nk filled rows without copying */
    int32_t zero_attn;        /* allow_zero_attn (fm_utils.py:28-30, :171-172): softmax over the scores AND one extra zero logit whose
                                 probability is dropped (p_k = e^{s_k} / (1 + sum_j e^{s_j})): a query may attend to nothing */
} fm_attn_args;
int fm_attn_fwd(const fm_attn_args* args, void* stream);
int fm_attn_bwd(const fm_attn_args* args, void* stream);   /* any Nq, Nk (single pass up to 512 rows, 256-row chunks above) */
void fm_set_attn_transpose_read(int on);
int fm_get_attn_transpose_read(void);

/* ------------------------------------------------------------------------------------------------
 * Token selection + embedding  (encoder: fm.py:245-277,338-390 + encoder_embeddings.py forward()s;
 * decoder: fm.py:279-336,392-438 + decoder_embeddings.py forward_embed()s)
 * ---------------------------------------------------------------------------------------------- */
#define FM_MAX_MODS 24
enum fm_mod_kind {
    FM_KIND_TOK = 0,      /* grid of discrete tokens      (ImageToken{En,De}coderEmbedding)   */
    FM_KIND_PATCH = 1,    /* raw pixels, patch projection (ImageEncoderEmbedding)             */
    FM_KIND_SEQ = 2,      /* token sequence               (Sequence{En,De}coderEmbedding)     */
    FM_KIND_SEQ_EMB = 3   /* dense-embedding sequence     (SequenceEmbEncoderEmbedding)       */
};
typedef struct fm_mod_desc {
    const void* ids;       /* TOK/SEQ: int64 or int32 ids; PATCH: f32 pixels (C,H,W); SEQ_EMB: f32 (n_pos, orig_dim) */
    const void* mask;      /* uint8 (B, mask_stride): input_mask (encoder) / target_mask (decoder), 1 = masked        */
    const void* dam;       /* int32 (B, mask_stride): decoder_attention_mask (decoder only)                           */
    const void* table;     /* f32 (vocab, D) token_emb.weight                                                         */
    const void* pos;       /* f32 (pos_rows, D) pos_emb                                                               */
    const void* mod_emb;   /* f32 (D)                                                                                 */
    const void* proj_bias; /* SEQ_EMB: f32 (D) emb_proj.bias                                                          */
    int32_t L;             /* positions contributed to the concatenation (decoder sequences: tensor_len - 1)         */
    int32_t kind, ids_are_i64, mod_id;
    int32_t max_len;       /* decoder sequences: position ids >= max_len collapse to 0 (decoder_embeddings.py:128)   */
    int32_t shifted;       /* decoder sequences: input j, target j+1, mask[j] | mask[j+1]   (fm.py:309-319)           */
    int32_t mask_stride, id_stride;   /* elements per sample in mask/dam resp. ids                                    */
    int32_t patch, channels, grid_w, orig_dim;
    int32_t head_index;    /* decoder: index of this modality's head (stable, independent of the shuffled order)    */
    int32_t pad_;
} fm_mod_desc;

typedef struct fm_select_desc {
    fm_mod_desc mods[FM_MAX_MODS];   /* in concatenation order */
    int32_t n_mods, batch, dim, n_keep, n_reg, total_len, is_decoder;
    int32_t raw;             /* 0: select n_keep positions (stable partition), masked slots zeroed — forward_mask_encoder/decoder;
                                1: every position in place, nothing zeroed — cat_encoder_tensors / cat_decoder_tensors (fm.py:245-336);
                                2: as 1 for ONE modality's own forward() / forward_embed() (decoder grid tokens embed their ids) */
    const void* reg_tokens;  /* f32 (n_reg, D)                      */
    const void* mask_token;  /* f32 (D), decoder                    */
    /* outputs, Nt = n_reg + n_keep rows per sample */
    void* tokens;       /* f32 (B, Nt, D)  gathered token rows (zero where masked)                 */
    void* emb;          /* f32 (B, Nt, D)  pos_emb + mod_emb (zero where masked)                   */
    void* x0;           /* f32 (B, Nt, D)  optional: tokens + emb                                  */
    void* out_mask;     /* uint8 (B, Nt)   1 = masked slot                                         */
    void* out_mod;      /* int16 (B, Nt)   modality id, -1 where masked / register                 */
    void* slot_mod;     /* int32 (B, Nt)   index into mods[] (-1 masked, -2 register)              */
    void* slot_src;     /* int32 (B, Nt)   token id (TOK/SEQ) or position inside the modality      */
    void* slot_pos;     /* int32 (B, Nt)   pos_emb row used                                        */
    void* target_ids;   /* int64 (B, Nt)   decoder                                                 */
    void* out_cs;       /* int32 (B, Nt)   decoder: cumsum of the kept decoder_attention_mask      */
    void* out_mod_pre;  /* int16 (B, Nt)   decoder: modality id *before* pads are set to -1        */
    void* out_mod_index;/* int32 (B, Nt)   decoder: head index, -1 where masked                    */
    void* patch_rows;   /* bf16 (B*Nt, patch_ld)  optional: pixels of the kept PATCH slots, zero elsewhere  */
    void* seqemb_rows;  /* bf16 (B*Nt, seqemb_ld) optional: embeddings of the kept SEQ_EMB slots            */
    int32_t patch_ld, seqemb_ld;
    int32_t rows_f32;   /* 1: patch_rows / seqemb_rows are f32 (fp32 verification path)                     */
    int32_t pad2_;
} fm_select_desc;
int fm_select_embed(const fm_select_desc* desc, void* stream);

typedef struct fm_embed_bwd_mod {
    void* d_table;      /* f32 (vocab, D) or NULL */
    void* d_pos;        /* f32 (pos_rows, D) for learned position embeddings, NULL for fixed sin-cos */
    void* d_mod_emb;    /* f32 (D) */
    void* d_proj_bias;  /* SEQ_EMB: f32 (D) */
    int32_t kind, has_padding_idx, padding_idx, pad_;
} fm_embed_bwd_mod;
typedef struct fm_embed_bwd_desc {
    fm_embed_bwd_mod mods[FM_MAX_MODS];
    const void* dx;          /* f32 (B, Nt, lddx): gradient w.r.t. tokens + emb */
    const void* slot_mod; const void* slot_src; const void* slot_pos;
    void* d_mask_token;      /* decoder */
    void* d_reg_tokens;      /* encoder, (n_reg, D) */
    int32_t n_mods, batch, dim, Nt, lddx, is_decoder;
} fm_embed_bwd_desc;
int fm_embed_bwd(const fm_embed_bwd_desc* desc, void* stream);

/* dense (B, M, M) uint8 mask from the compressed form (FourM.adapt_decoder_attention_mask, fm.py:440-475) */
int fm_dense_decoder_mask(const int32_t* cs, const int16_t* mod, void* out, int B, int M, int causal, int use_cs,
                          int use_sep, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Heads: segment rows by modality, cross-entropy  (fm.py:573-637)
 * ---------------------------------------------------------------------------------------------- */
enum fm_loss_type { FM_LOSS_MOD = 0, FM_LOSS_TOKEN = 1 };
#define FM_SEG_ROWS 256   /* row alignment of the per-modality segments = the row tile of the grouped GEMMs */
/* Buckets rows 0..R-1 by head_of_row (-1 = no head) into FM_SEG_ROWS-aligned segments of a padded row
 * space of Rp rows (Rp % FM_SEG_ROWS == 0, Rp >= roundup(R) + FM_SEG_ROWS*(n_heads-1)).  perm[pr] = source row or
 * -1; row_to_padded[r] = padded row or -1; tile_group[pr/FM_SEG_ROWS] = head or -1. */
int fm_segment_rows(const int32_t* head_of_row, int R, int n_heads, int32_t* seg_start, int32_t* seg_count,
                    int32_t* perm, int32_t* row_to_padded, int32_t* tile_group, int Rp, void* stream);
int fm_gather_rows(const void* src, int ld_src, const int32_t* perm, void* dst, int ld_dst, int Rp, int D, void* stream);
/* logits: bf16 (Rp, ldl) produced by the grouped fm_gemm_nt (16-byte aligned, ldl % 8 == 0).
 * write_grad == 0 (forward): one streaming pass; writes row_loss (Rp), row_lse (Rp), head_loss (n_heads) and
 *   total_loss (1), all f32.
 * write_grad == 1 (backward): the logits are replaced in place by d(total)/d(logits) * grad_scale[0] (bf16; pad
 *   rows and the columns up to roundup64(vocab) zeroed) using the row_lse of the forward call. */
int fm_cross_entropy(void* logits, int ldl, const int32_t* perm, const int32_t* tile_group, const int64_t* target_ids,
                     const int32_t* vocab, const int32_t* seg_start, const int32_t* seg_count, const void* grad_scale,
                     int loss_type, int n_heads, int Rp, int max_vocab, void* row_loss, void* row_lse, void* head_loss,
                     void* total_loss, int write_grad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise / reductions
 * ---------------------------------------------------------------------------------------------- */
/* fm_swiglu_bwd: Hp and the leading dims multiples of 8, buffers 16-byte aligned (16-byte accesses). */
int fm_swiglu_bwd(const void* da, int ldda, const void* gu, int ldgu, void* dgu, int lddgu, int R, int H, int Hp, void* stream);
int fm_gelu_bwd(const void* dh, int lddh, const void* pre, int ldp, void* dpre, int lddp, int R, int H, int Hp, void* stream);
/* One launch refreshing many bf16 weight shadows from their fp32 masters (what autocast's per-call weight
 * casts amount to, fm_utils.py Linear layers under torch.autocast).  Job i covers the 64x64 tiles
 * [tile_start_i, tile_start_{i+1}) of its (rows, cols) source, tiles_i = ceil(rows/64)*ceil(cols/64);
 * transpose == 0: dst[r][c] = bf16(src[r][c]);  transpose == 1: dst[c][r] = bf16(src[r][c]).
 * col_scale (optional, fp32[cols]): the source is multiplied by col_scale[c] first - a LayerNorm weight folded into the Linear
 * that consumes the normalised rows (DecoderBlock.context_norm -> cross_attn.kv, fm_utils.py:364: every decoder block normalises
 * the SAME context, so x_hat is computed once and gamma_l lives in the weight image).  dst_f32 != 0: fp32 destination (the
 * fp32 verification mode keeps the same launch sequence).
 * Destination padding is not written.  descs: DEVICE pointer, sorted by tile_start (first = 0). */
typedef struct fm_shadow_desc {
    const void* src; void* dst;
    int32_t ld_src, ld_dst, rows, cols, transpose, tile_start;
    const void* col_scale;
    int32_t dst_f32, reserved;
} fm_shadow_desc;
int fm_shadow_refresh(const fm_shadow_desc* descs, int n_descs, int total_tiles, void* stream);
/* Gradient of a folded LayerNorm weight (see col_scale above).  With W' = W diag(gamma) used in the forward and
 * dWp = dL/dW' (rows x cols, fp32, leading dimension ld_dwp) produced by the ordinary weight-gradient GEMM:
 *     gW[r][c]  += dWp[r][c] * gamma[c]                (dL/dW,     accumulated like every gradient)
 *     ggamma[c] += sum_r dWp[r][c] * W[r][c]           (dL/dgamma, fp32 atomics)
 * W, gW: contiguous (rows x cols) fp32.  Job table passed by value (graph-capturable, no upload). */
#define FM_FOLD_MAX_JOBS 48
typedef struct fm_fold_grad_job {
    const void* dWp; const void* W; const void* gamma; void* gW; void* ggamma;
    int32_t rows, cols, ld_dwp, reserved;
} fm_fold_grad_job;
int fm_fold_colscale_grad(const fm_fold_grad_job* jobs, int n_jobs, void* stream);

/* bf16 weight shadows: dst[r][c] = src[r][c], columns [cols, ld_dst) zero /
 * dst[c][r] = src[r][c], columns [rows, dst_cols) zero (dst_cols <= ld_dst) */
int fm_cast_pad(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, void* stream);
int fm_transpose_cast_pad(const void* src, int ld_src, void* dst, int ld_dst, int dst_cols, int rows, int cols, void* stream);
int fm_colsum(const void* dy, int ldy, void* db, int R, int N, void* stream);            /* db[n] += sum_r dy[r][n] */
int fm_f32_to_bf16(const void* src, void* dst, int64_t n, void* stream);
/* dst(f32)[i] = scale * src(bf16)[i]: unpacks a gradient bucket that travelled in bf16 (optional wire format of the exchange) */
int fm_bf16_to_f32_scaled(const void* src, void* dst, int64_t n, float scale, void* stream);
/* out = x + delta over n contiguous elements (x, out f32; delta bf16): the residual add of fm_utils.py:332-333 on its own */
int fm_add_bf16_f32(const void* x, const void* delta, void* out, int64_t n, void* stream);
/* Stochastic depth (DropPath, fourm/models/fm_utils.py:64-87): x[r][:] *= scale[r / rows_per_sample] in place on a bf16 (R, N) tile of
 * row stride ld (scale f32 per sample = mask / keep_prob).  The forward scales a residual branch's output, the backward the bf16
 * gradient copy that enters the branch.  N, ld multiples of 8. */
int fm_scale_rows_bf16(void* x, int ld, const void* scale, int rows_per_sample, int R, int N, void* stream);

/* torch.optim.AdamW update on a contiguous fp32 range (fourm/utils/optim_factory.py:239-240);
 * grad_mult: optional device scalar multiplied into the gradient (clipping).
 * hyper: optional DEVICE float[4] = {lr, weight_decay, 1 - beta1^step, sqrt(1 - beta2^step)} that overrides the scalar arguments:
 * a launch captured in a hipGraph keeps following the schedule (the host refreshes those 16 bytes before each replay).
 * sumsq: optional device scalar that receives += sum g^2 of the RAW gradients of this range (round 5: with no clipping the gradient norm of
 * NativeScaler / get_grad_norm_ rides on the update's own pass over the gradients instead of a separate 4 B/param read). */
int fm_adamw(void* p, const void* g, void* m, void* v, int64_t n, float lr, float beta1, float beta2, float eps,
             float weight_decay, int64_t step, const void* grad_mult, const void* hyper, void* sumsq, void* stream);
/* AdamW on weight MATRICES with their plain bf16 shadow (the W operand of y = x W^T) rewritten in the same streaming pass: the
 * update already reads and writes every master weight, the bf16 copy costs 2 B/param on top instead of a separate 6 B/param pass.
 * One launch walks a device table of jobs in tiles of FM_ADAMW_CHUNK (8192) consecutive elements; job i owns tiles
 * [tile_start_i, tile_start_{i+1}); p, g, m, v: contiguous fp32 (rows, cols); dst_plain (optional): bf16 [r][c], row stride ld_plain.
 * dst_t / ld_t are reserved (transposed shadows are refreshed by fm_shadow_refresh).  All jobs of a launch share the
 * hyper-parameters and the step count.  Same arithmetic as fm_adamw. */
#define FM_ADAMW_CHUNK 8192
typedef struct fm_adamw_job {
    void* p; const void* g; void* m; void* v; void* dst_plain; void* dst_t;
    int32_t rows, cols, ld_plain, ld_t, tile_start, pad_;
} fm_adamw_job;
int fm_adamw_shadow(const fm_adamw_job* jobs, int n_jobs, int total_tiles, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int64_t step, const void* grad_mult, const void* hyper, void* sumsq, void* stream);
int fm_sumsq(const void* x, int64_t n, void* out, void* stream);                         /* out[0] += sum x^2 */
int fm_clip_coef(const void* sumsq, float max_norm, void* norm_out, void* coef_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Generation step: token sampling + MaskGIT commit (csrc/sample.hip) — GenerationSampler.top_k_top_p_filtering / sample_tokens /
 * select_tokens_batched and the scatter updates of maskgit_step_batched, fourm/models/generate.py:332-420,650-661.
 * One workgroup per logits row: temperature (0 = argmax), top_k (0 = off; entries strictly below the k-th largest logit are
 * dropped), top_p (0 or 1 = off; an entry survives iff the temperature-1 probability mass of strictly larger logits is <= top_p),
 * then one multinomial draw by inverse CDF in index order from uniforms[row] in [0, 1).  Deterministic: fixed-polynomial exp with
 * separately rounded fp32 operations, integer radix selects, a fixed summation tree (oracle/sample_oracle.py is bit-identical).
 * out_ids: int64 (R); out_prob: f32 (R) = probability of the drawn token after filtering at the sampling temperature. */
int fm_sample_tokens(const void* logits, int ld, int logits_are_f32, int R, int V, float temperature, int top_k, float top_p,
                     const void* uniforms, void* out_ids, void* out_prob, void* stream);
/* Per sample b: the num_select entries of prob (B, N) with the largest value (ties: lower index first), in that order, go to
 * top_idx (B, num_select) and are committed: tensor[b][mod_pos[b][i]] = samples[b][i] (int64 or int32 tensor of row length L),
 * input_mask[b][pos] = 0, target_mask[b][pos] = 1 (bool / uint8). */
/* Classifier-free guidance on logits (generate.py:684, :718): out = base + weight * (cond - uncond) in fp32, one rounding per operation;
 * base = uncond (accumulate == 0) or the current out (accumulate != 0: further weighted conditions).  uncond / cond: bf16 or f32 (R, V)
 * with row strides ldu / ldc; out: f32 (R, ldo). */
int fm_guidance_combine(const void* uncond, int ldu, int uncond_is_f32, const void* cond, int ldc, int cond_is_f32, float weight,
                        void* out, int ldo, int R, int V, int accumulate, void* stream);
int fm_maskgit_commit(const void* prob, const void* samples, const int32_t* mod_pos, int B, int N, int num_select, void* tensor,
                      int tensor_is_i64, int L, void* input_mask, void* target_mask, int32_t* top_idx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp32 VERIFICATION path (csrc/fp32_verify.hip): the floating-point kernels above once more with fp32 activations and weights
 * and no bf16 rounding, so that the launch sequence of the engine can be checked against the upstream fp32 model at fp32
 * tolerances.  Plain kernels, not the hot path.  Same conventions; every activation pointer is f32.
 * ---------------------------------------------------------------------------------------------- */
/* out[m][n] (+)= sum_k X[m*sxm + k*sxk] * W[n*swn + k*swk]  + epilogue (fm_epilogue without the *_BWD forms; no rounding).
 * Fully strided operands: NT (sxk = swk = 1), dX = dY W (swn = 1, swk = ldw) and dW = dY^T X (sxm = swn = 1, accumulate = 1)
 * are one kernel.  Grouped NT: groups + tile_group + seg_rows as in fm_gemm_nt (groups[g].W f32).  Grouped TN: groups[g].out /
 * .N with seg_start / seg_count (device), max_N = largest group N, n_groups; reduces rows [seg_start[g], + seg_count[g]). */
typedef struct fm_gemm_f32_args {
    const void* X; const void* W; const void* W2; void* out; void* out2; const void* res; const void* bias; const void* bias2;
    int64_t sxm, sxk, swn, swk;
    int32_t M, N, K, ldo, ldo2, ldr, Hp, epilogue, accumulate, max_N, seg_rows, n_groups;
    const fm_gemm_group* groups; const int32_t* tile_group; const int32_t* seg_start; const int32_t* seg_count;
} fm_gemm_f32_args;
int fm_gemm_f32(const fm_gemm_f32_args* args, void* stream);
/* fm_attn_args with f32 Q / K / V / O (/ dO, dQ, dK, dV); blocked scores are replaced by -finfo(float32).max; stat_m / stat_l
 * unused.  The backward ACCUMULATES into dK and dV (the caller zeroes them). */
int fm_attn_f32_fwd(const fm_attn_args* args, void* stream);
int fm_attn_f32_bwd(const fm_attn_args* args, void* stream);
int fm_layernorm_bwd_f32(const void* dy, int lddy, const int32_t* dy_row_map, const void* x, int ldx, const void* w, const void* mean,
                         const void* rstd, const void* dres, void* dx, int lddx, void* dx2, int lddx2, void* dw, void* db, int R, int D,
                         void* stream);
int fm_headnorm_f32_fwd(const void* x, int ldx, const void* w, const void* b, void* y, int ldy, void* stats, int R, int H, float eps, void* stream);
int fm_headnorm_f32_bwd(const void* dy, int lddy, const void* x, int ldx, const void* w, const void* stats, void* dx, int lddx, void* dw, void* db,
                        int R, int H, void* stream);
int fm_swiglu_bwd_f32(const void* da, int ldda, const void* gu, int ldgu, void* dgu, int lddgu, int R, int H, int Hp, void* stream);
int fm_gelu_bwd_f32(const void* dh, int lddh, const void* pre, int ldp, void* dpre, int lddp, int R, int H, void* stream);
int fm_colsum_f32(const void* dy, int ldy, void* db, int R, int N, void* stream);
int fm_cross_entropy_f32(void* logits, int ldl, const int32_t* perm, const int32_t* tile_group, const int64_t* target_ids,
                         const int32_t* vocab, const int32_t* seg_start, const int32_t* seg_count, const void* grad_scale, int loss_type,
                         int n_heads, int padded_rows, int max_vocab, void* row_loss, void* row_lse, void* head_loss, void* total_loss,
                         int write_grad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * VQ tokenizer front end (fourm/vq/vqvae.py:302-331)
 * ---------------------------------------------------------------------------------------------- */
/* out[(b*G + g)][c*P*P + py*P + px] (bf16, row stride ld_out, pad zero) <- img (B, C, H, W) f32: the operand
 * of the patch-projection GEMM that replaces Conv2d(k = s = P)  (vq/models/vit_models.py:402-405,482) */
int fm_vq_patchify(const void* img, void* out, int ld_out, int B, int C, int H, int W, int P, void* stream);
int fm_l2norm_rows(const void* x, int ldx, void* y, int ldy, int R, int D, void* stream);   /* F.normalize(p=2, dim=-1) */
/* Cosine-similarity nearest code (quantize_lucid.py:394-407): tokens[r] = argmax_c <zn[r], En[c]> with the
 * first maximum winning; quant (optional, f32 (B, D, tokens_per_image)) = embed[tokens].  z: f32 (R, ldz),
 * l2-normalised in-kernel when normalize_latents; codes_normalized / embed: f32 (K, D).  D == 32.
 * ws_val / ws_idx: (R, splits) f32 / int32 scratch. */
int fm_vq_assign(const void* z, int ldz, const void* codes_normalized, const void* embed, int K, int D, int R,
                 int tokens_per_image, int normalize_latents, void* ws_val, void* ws_idx, int splits, int64_t* tokens,
                 void* quant, void* stream);
/* Codebook training statistics (CosineSimCodebook.forward, training branch, quantize_lucid.py:409-419): bins[k] = number of
 * latents assigned to code k, sums[k][:] = sum of those latents after L2 normalisation.  z: f32 (R, ldz); tokens: int64 (R);
 * bins f32 (K), sums f32 (K, D): zeroed here, accumulated with fp32 atomics.  With a synchronised codebook the caller all-reduces
 * bins and sums over the ranks between the two calls (:411, :419).  D <= 64. */
int fm_vq_code_stats(const void* z, int ldz, const int64_t* tokens, int R, int D, int K, void* bins, void* sums, void* stream);
/* EMA update (quantize_lucid.py:413, :421-425): cluster_size = cluster_size * decay + bins * (1 - decay);
 * embed = embed * decay + target * (1 - decay), target = l2norm(sums / bins) where bins > 0, l2norm(embed) elsewhere. */
int fm_vq_ema_update(const void* bins, const void* sums, void* embed, void* cluster_size, int K, int D, float decay, void* stream);
/* Euclidean codebook (EuclideanCodebook, quantize_lucid.py:181-301; VectorQuantize(use_cosine_sim=False), i.e. norm_codes=False):
 *   fm_vq_code_bias:   bias[k] = -|embed[k]|^2 / 2;
 *   fm_vq_assign_bias: fm_vq_assign with the score of code c starting at code_bias[c] (NULL: 0): with `codes` = embed (not normalised),
 *       normalize_latents = 0 and that bias the arg-max is the nearest code in Euclidean distance (:272-280);
 *   fm_vq_code_stats_raw: fm_vq_code_stats on the latents as they are (no L2 normalisation; :283-289);
 *   fm_vq_ema_update_euclid: cluster_size and embed_avg EMAs, then embed = embed_avg / Laplace-smoothed cluster size (:286-296).
 *       total_scratch: one DEVICE float (zeroed here). */
int fm_vq_code_bias(const void* embed, int K, int D, void* bias, void* stream);
int fm_vq_assign_bias(const void* z, int ldz, const void* codes, const void* code_bias, const void* embed, int K, int D, int R,
                      int tokens_per_image, int normalize_latents, void* ws_val, void* ws_idx, int splits, int64_t* tokens,
                      void* quant, void* stream);
int fm_vq_code_stats_raw(const void* z, int ldz, const int64_t* tokens, int R, int D, int K, void* bins, void* sums, void* stream);
int fm_vq_ema_update_euclid(const void* bins, const void* sums, void* embed, void* embed_avg, void* cluster_size, void* total_scratch,
                            int K, int D, float decay, float eps, void* stream);

/* Tokenizer training path (fourm/vq/vqvae.py:454-481 VQVAE, vq/models/vit_models.py:504-648 ViTDecoder,
 * vq/quantizers/quantize_lucid.py:533-541).
 *   fm_vq_unpatchify: rows f32 (B * G, ld_rows) with features ordered (c, py, px) -> img f32 (B, C, H, W)  (the rearrange behind
 *       ViTDecoder.out_proj, vit_models.py:640-643; the inverse of fm_vq_patchify)
 *   fm_vq_latent_grad: the quantizer's backward and commitment value.  z f32 (R, ldz) latents, tokens (R), embed f32 (K, D);
 *       dquant f32 (R, ld_dquant) or NULL: gradient w.r.t. the quantised rows (straight-through estimator);  grad_loss: DEVICE f32
 *       scalar d(objective)/d(code_loss) or NULL;  dz (optional) = dquant + grad_loss * w * 2 (z - embed[token]) / (R D);
 *       commit_value (optional, f32 scalar the caller zeroed) += w * mean((z - embed[token])^2).  D <= 64.
 *   fm_tanh_bwd_f32: dx = dy * (1 - t * t) on f32 (R, N) tiles of row stride ld (Mlp with act_layer = Tanh, vit_models.py:494-496)
 *   fm_embed_rows_f32: out[r] = table[idx[r]] (f32 rows; F.embedding) */
int fm_vq_unpatchify(const void* rows, int ld_rows, void* img, int B, int C, int H, int W, int P, void* stream);
int fm_vq_latent_grad(const void* z, int ldz, const void* embed, const int64_t* tokens, const void* dquant, int ld_dquant, const void* grad_loss,
                      float commitment_weight, void* dz, int ld_dz, void* commit_value, int R, int D, void* stream);
int fm_tanh_bwd_f32(const void* dy, const void* t, void* dx, int R, int N, int ld, void* stream);
int fm_embed_rows_f32(const void* table, const int64_t* idx, void* out, int ld_out, int R, int D, void* stream);
/* fp32 products on the bf16 matrix cores: out (R, ldo >= 3 K) bf16 = [hi | hi | lo] of x f32 (R, K) (weight_order = 0: the activation
 * operand) or [hi | lo | hi] (weight_order = 1: the weight operand), hi = bf16(x), lo = bf16(x - hi); apply_tanh: x <- tanh(x) first.
 * ONE bf16 NT GEMM over the 3 K columns then accumulates hi hi + hi lo + lo hi in fp32: ~2^-16 relative (the tokenizer's fp32 tail at
 * inference, vit_models.py:494-496). */
int fm_split3_bf16(const void* x, int ldx, void* out, int ldo, int R, int K, int weight_order, int apply_tanh, void* stream);
/* Input variants of the tokenizer (VQ.prepare_input, vq/vqvae.py:269-286) folded into the patch gather: fm_vq_patchify with
 *   labels != NULL: class maps int64 (B, H, W) embedded by cls_emb f32 (n_labels, C) (semantic segmentation, n_labels; img unused);
 *   scale / shift: DEVICE float[C] (or both NULL): value = scale[c] * v + shift[c] (undo_std: 2 * denormalize(x) - 1).
 * fm_vq_cls_emb_bwd: d cls_emb[label] += the bf16 gradient of the patch rows (B * G, ld), fp32 atomics (the embedding's backward).
 * fm_vq_latent_grad_normalized: fm_vq_latent_grad for norm_latents = True (quantize_lucid.py:525-527): x = l2norm(z) enters the
 *   commitment term, dz carries the backward of the normalisation. */
int fm_vq_patchify_ex(const void* img, const int64_t* labels, const void* cls_emb, const void* scale, const void* shift, void* out, int ld_out,
                      int B, int C, int H, int W, int P, void* stream);
int fm_vq_cls_emb_bwd(const void* d_patches, int ld, const int64_t* labels, void* d_cls_emb, int B, int C, int H, int W, int P, void* stream);
int fm_vq_latent_grad_normalized(const void* z, int ldz, const void* embed, const int64_t* tokens, const void* dquant, int ld_dquant, const void* grad_loss,
                                 float commitment_weight, void* dz, int ld_dz, void* commit_value, int R, int D, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data side: input / target masks of an image-like modality (UnifiedMasking.image_mask, fourm/data/masking.py:237-266)
 * noise: f32 (B, L) uniform draws (the caller's RNG); input_budget / target_budget: int32 (B) (target_budget NULL = "None":
 * every non-input position is a target).  With ids = stable argsort(noise[b]):  input_mask[b][i] = ids[i] >= input_budget[b],
 * target_mask[b][i] = !(input_budget <= ids[i] < input_budget + target_budget)  (uint8, 1 = masked out), and
 * decoder_attention_mask (int32) = 0 except the number of targets at the first target position.  L <= 4096. */
int fm_image_mask(const void* noise, const int32_t* input_budget, const int32_t* target_budget, int B, int L, void* input_mask,
                  void* target_mask, int32_t* decoder_attention_mask, void* stream);

/* Compact host-to-device batch format: uint8 pixels, uint16 ids and bit-packed masks cross PCIe, the mod_dict tensors
 * the model takes are rebuilt on the device.
 *   fm_unpack_image_u8: (B, H, W, C) uint8 -> (B, C, H, W) f32 = ((x / 255) - mean[c]) / std[c]  (to_tensor + normalize,
 *       fourm/data/modality_transforms.py:215-218; mean / std are HOST arrays of C floats; bit-identical to the float pipeline)
 *   fm_unpack_ids_u16: n uint16 -> int64
 *   fm_unpack_mask_bits: row b = ceil(L / 8) bytes, bit (i & 7) of byte (i >> 3) -> uint8 / bool (B, L)
 *   fm_decoder_attention_from_target: image-like modalities' decoder_attention_mask (target count at the first target position,
 *       fourm/data/masking.py:262-264) from the unpacked target_mask (1 = masked out) */
int fm_unpack_image_u8(const void* src, void* dst, int B, int H, int W, int C, const float* mean, const float* stdv, void* stream);
int fm_unpack_ids_u16(const void* src, int64_t* dst, int64_t n, void* stream);
int fm_unpack_mask_bits(const void* bits, void* dst, int B, int L, void* stream);
int fm_decoder_attention_from_target(const void* target_mask, int32_t* decoder_attention_mask, int B, int L, void* stream);

/* Token budgets of a batch (UnifiedMasking.input_token_budget / target_token_budget, fourm/data/masking.py:181-234), one sample per
 * thread.  The caller draws:  main_draws f32 (B, T, M) - the Dirichlet sample of try t;  extra_draws f32 (B, T, E, M) - the
 * sample_n(diff) draws of try t (the first diff = n - sum(floor(p * n)) rows are used, E >= M covers every case).
 *   budget = floor(p * n) + bincount(argmax(extra[:diff])), clamped to max_tokens[m]; the first try with budget >= min_tokens everywhere is
 *   taken, else the last one (tries (B), optional: the number of tries consumed; T + 1 = none fitted).
 * Target budgets: pass input_budget int32 (B, M) and is_img uint8 (M): the clamp becomes max(min_tokens, max_tokens - input_budget) for
 * image-like modalities (:218-219).  Input budgets: input_budget = NULL.  M <= 64. */
int fm_token_budgets(const void* main_draws, const void* extra_draws, const int32_t* num_tokens, const int32_t* min_tokens,
                     const int32_t* max_tokens, const void* is_img, const int32_t* input_budget, int B, int T, int E, int M,
                     int32_t* budget, int32_t* tries, void* stream);

/* Span masking of sequence modalities (fourm/data/masking.py: sequence_mask :345-445 after tokenisation, sequence_token_mask :268-343,
 * sequence_emb_mask_span :448-516; simple_span_masking :58-91, chunk_span_masking :94-127), one wave per sample.
 *   ids (B, ld_ids) int32 with len (B): the token ids upstream has after tokenising and appending [EOS] (vocab_offset is added, :286);
 *   unit (B, ld_ids) or NULL: index of the chunk a token belongs to (chunk_span_masking: one mask decision per chunk, truncation by whole
 *     chunks; chunks must be non-empty);  NULL = one decision per token (truncation to max_tokens tokens);
 *   noise f32 (B, T, ld_noise): row t = the torch.rand vector of try t, indexed by token (or chunk);  keep_prob f64 (B): the first keep
 *     probability (sample_uniform / 1.0 / random.choice), multiplied by 0.9 per retry while the input is over its budget (:405-408); a unit
 *     is kept iff noise <= (float)keep_prob;  after T tries everything is masked (the limit keep_prob -> 0);
 *   input_budget (B);  target_budget (B) or NULL (= None; an entry < 0 = None for that sample);  r_choice (B) or NULL: the integer that
 *     replaces np.random.randint (taken modulo its argument, :425);  sentinel_ids[k] = sentinel_to_id[k].
 * Outputs, each (B, 2 * (max_tokens + 1)): tensor int32 (pad_id where empty), input_mask / target_mask uint8 (1 = masked out),
 * decoder_attention_mask int32;  tries (B) optional: draws consumed (T + 1 = they ran out: everything masked), -1 = more spans than sentinel ids (upstream raises KeyError).
 * Embedding mode (emb != NULL; sequence_emb_mask_span): emb f32 (B, emb_rows, emb_dim) with len (B); ids / unit / tensor / target_budget
 * unused; outputs are (B, max_tokens): input_mask, target_mask (all 1), decoder_attention_mask (all 0), src int32 (the source row of every
 * position, -1 = zero row) and emb_out f32 (B, max_tokens, emb_dim). */
typedef struct fm_span_mask_args {
    const int32_t* ids; const int32_t* len; const int32_t* unit; const void* noise; const double* keep_prob; const int32_t* r_choice;
    const int32_t* input_budget; const int32_t* target_budget; const int32_t* sentinel_ids;
    const void* emb; void* emb_out; int32_t* src;
    int32_t* tensor; void* input_mask; void* target_mask; int32_t* decoder_attention_mask; int32_t* tries;
    int32_t B, ld_ids, T, ld_noise, n_sentinels, max_tokens, vocab_offset, pad_id, emb_rows, emb_dim;
} fm_span_mask_args;
int fm_span_mask(const fm_span_mask_args* p, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Diffusion detokenizer (DiVAE decoder), inference (csrc/unet.hip).  Feature maps are rows (B * H * W, C) bf16 ("NHWC"): a 1 x 1
 * convolution is fm_gemm_nt on the rows; a 3 x 3 convolution is fm_unet_im2col + fm_gemm_nt with K = 9 C.
 * Replaces fourm/vq/models/unet/unet.py (ResBlock :163-274, AttentionBlock :277-322, QKVAttentionLegacy :345-374, Upsample /
 * Downsample :103-160, PatchedUNetCondCat :693-744), nn.py:23-25 / :120-140, and the element-wise half of
 * fourm/vq/scheduling/scheduling_{ddim,ddpm}.py (step, _threshold_sample).
 * ---------------------------------------------------------------------------------------------- */
/* out[(b, oy, ox)][tap * (C1 + C2) + c] (bf16, row stride ldo; columns [ksize^2 (C1 + C2), kpad) zero) <- the ksize x ksize neighbourhood
 * (padding ksize / 2, stride 1 | 2) of the channel concatenation [src1 | src2] on an (H, W) grid: src1 (B, H >> up1, W >> up1, C1) is read
 * at (y >> up1, x >> up1) (up1 = 1: the nearest x2 up-sampling of Upsample), src2 (B, H2, W2, C2) at (y H2 / H, x W2 / W) (nearest: the skip
 * tensor of an output block, or the conditioning under a finer grid); src2 = NULL, C2 = 0: one source.  ksize = 1: the plain concatenation.
 * The weight operand of the GEMM behind it is conv.weight.permute(0, 2, 3, 1).reshape(Cout, ksize^2 (C1 + C2)). */
int fm_unet_im2col(const void* src1, int ld1, int C1, const void* src2, int ld2, int C2, int H2, int W2, void* out, int ldo, int kpad,
                   int B, int H, int W, int ksize, int stride, int up1, void* stream);
/* y = [silu] GroupNorm_groups(x + add[b][c]) * w + b over rows (B, HW, C): fp32 statistics per (sample, group) over HW x C / groups
 * values (two passes), fp32 arithmetic, bf16 result (GroupNorm32, nn.py:23-25; ``add`` f32 (B, >= C) with row stride ld_add, or NULL: the
 * timestep embedding a ResBlock adds in front of out_layers, unet.py:270).  stats: f32 scratch of B * groups * 2 * (ceil(HW / 32) + 1) values (per-chunk partial sums,
 * added in a fixed order: the result is bit-reproducible).  C <= 1024. */
int fm_groupnorm_nhwc(const void* x, int ldx, const void* add, int ld_add, const void* w, const void* b, void* y, int ldy, void* stats, int B, int HW,
                      int C, int groups, float eps, int silu, void* stream);
int fm_add_bf16(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int64_t rows, int C, void* stream);   /* out = a + b, bf16 rows */
int fm_silu_f32_to_bf16(const void* x, void* y, int64_t n, void* stream);                                                 /* y = bf16(x * sigmoid(x)) */
/* out[b] = [cos(t_b f_i) | sin(t_b f_i)], f_i = exp(-ln(max_period) i / (dim / 2)); t f32 (B), out bf16 (B, ldo)   (nn.py:120-140) */
int fm_timestep_embedding(const void* t, void* out, int ldo, int B, int dim, float max_period, void* stream);
/* Spatial self-attention of AttentionBlock with QKVAttentionLegacy: qkv bf16 (B * T, ld) whose columns are, per head, [q | k | v] of ch
 * channels each; weights = softmax_fp32(q k^T / sqrt(ch)); out bf16 (B * T, ldo) with a head's ch channels side by side. */
int fm_unet_attention(const void* qkv, int ld, void* out, int ldo, int B, int T, int heads, int ch, void* stream);
/* Scheduler step, element-wise (f32, n = B * per_sample values):
 *   fm_diffusion_x0:    x0 = c0 * sample + c1 * model_output                     (v-prediction: c0 = sqrt(a_t), c1 = -sqrt(1 - a_t); ...)
 *   fm_quantile_abs:    out[b] = torch.quantile(|x[b]|, q) (linear interpolation) - radix select, one workgroup per sample
 *   fm_diffusion_step:  x0' = quantile ? clamp(x0, -s, s) / s with s = clamp(quantile[b], 1, sample_max_value)   (_threshold_sample)
 *                           : clip_range > 0 ? clamp(x0, -clip_range, clip_range) : x0;
 *                       out = k0 * x0' + k1 * sample + k2 * model_output + k3 * noise   (noise may be NULL);  x0_out (optional) = x0' */
int fm_diffusion_x0(const void* sample, const void* model_output, float c0, float c1, void* x0, int64_t n, void* stream);
int fm_quantile_abs(const void* x, int B, int64_t n, float q, void* out, void* stream);
int fm_diffusion_step(const void* x0, const void* quantile, float sample_max_value, float clip_range, const void* sample, const void* model_output,
                      const void* noise, float k0, float k1, float k2, float k3, void* out, void* x0_out, int B, int64_t per_sample, void* stream);

/* ---- lab (not part of the drop-in surface) ------------------------------------------------------------------------------------------
 * Experiment knobs of the GEMM kernels, used by tools/gemm_lab.cpp, the Python tools and the FOURM_NT3_LAB environment switch only: key 0 / 1
 * staggered workgroup start (groups, step), 2 gemm_nt3 mode override, 3 gemm_nt3 ablation flags (gemm_args.h NTArgs::lab), others reserved.
 * Process-global, not thread-safe, no effect on results except where an ablation flag says so.  Keys outside [0, 16) are ignored. */
void fm_lab_set(int key, int value);

#ifdef __cplusplus
}
#endif
#endif /* FOURM_HIP_H */
