/*
 * peneo_hip.h — C ABI of libpeneo_hip.so, the MI355X (gfx950) device library behind the
 * PEneo forward/backward hot path.
 *
 * The upstream reference (ZeningLin/PEneo) is pure Python; it has no FFI.  The boundary a
 * maintainer swaps is the nn.Module contract of model/modeling_peneo.py:108-175 and
 * model/peneo_decoder.py:338-443.  Each entry point below names the reference code whose
 * device work it replaces.  INTEGRATION.md shows the ctypes stub that binds them.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - sizes are explicit; no hidden allocation: workspaces are passed in by the caller;
 *   - `stream` is a hipStream_t passed as void*; all work is asynchronous on it;
 *   - return 0 on success, a negative PENEO_ERR_* otherwise; peneo_last_error() gives the
 *     (thread-local) message;
 *   - "dtype" selects the storage type of activations/weights: PENEO_BF16 (MFMA bf16 inputs,
 *     fp32 accumulate; the throughput mode) or PENEO_F32 (exact fp32 MFMA; the parity mode).
 *     Gradients of parameters, losses, logits, LayerNorm statistics and softmax
 *     log-sum-exps are always fp32.
 */
#ifndef PENEO_HIP_H
#define PENEO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PENEO_OK 0
#define PENEO_ERR_INVALID (-1)
#define PENEO_ERR_LAUNCH (-2)

#define PENEO_F32 0
#define PENEO_BF16 1

#define PENEO_ACT_NONE 0
#define PENEO_ACT_GELU 1 /* exact erf GELU (transformers RobertaIntermediate) */
#define PENEO_ACT_SILU 2 /* nn.SiLU (model/peneo_decoder.py:217,220,235) */

#define PENEO_MAX_HEADS 8 /* pair-classifier heads (the reference has 5) */

typedef void* peneo_stream_t;

int peneo_version(void);
const char* peneo_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Dense contraction:  C[m,n] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
 *   a_kmajor != 0 : A is [M, K] row-major (leading dim lda >= K)   else A is [K, M] (lda >= M)
 *   b_kmajor != 0 : B is [N, K] row-major (ldb >= K)               else B is [K, N] (ldb >= N)
 * so  forward  y = x W^T      -> (a_kmajor=1, b_kmajor=1)            [nn.Linear everywhere]
 *     dgrad    dx = dy W      -> (1, 0)
 *     wgrad    dW = dy^T x    -> (0, 0)
 * Epilogue order:  v = alpha*acc (+bias[n]) ; preact <- v ; v = act(v) ; v *= act'(grad_src[m,n]) ;
 *                  dropout ; v += residual[m,n] ; v += C[m,n] if accumulate ; C <- v
 * Replaces torch.nn.Linear / F.linear call sites: modeling_layoutlmv3.py:292-294,335-360,
 * RobertaSelfOutput/Intermediate/Output, peneo_decoder.py:126,213-222 and their autograd.
 * ------------------------------------------------------------------------------------------ */
struct peneo_pair_dz_args;
typedef struct peneo_gemm_epilogue {
  const float* bias;        /* [N] fp32 or NULL */
  int act;                  /* PENEO_ACT_* applied after bias */
  void* preact;             /* optional [M, ld_preact] (dtype c_dtype): value before act */
  int64_t ld_preact;
  const void* grad_src;     /* optional [M, ld_grad] (dtype c_dtype): multiply by act'(grad_src) */
  int64_t ld_grad;
  int grad_act;             /* which act' */
  const void* residual;     /* optional [M, ld_res] (dtype c_dtype) added last */
  int64_t ld_res;
  float alpha;              /* 0 is read as 1 */
  int accumulate;           /* C += result (c_dtype must be PENEO_F32) */
  float drop_p;             /* dropout probability (0 = off); keep mask = f(drop_seed, m*N+n) */
  uint32_t drop_seed;
  /* pair-head backward fusion (K12 bwd): when set, the tile is the first-layer pre-activation z = acc + bias of
   * rows = pairs, columns = [head, hidden]; the epilogue stores dz (see peneo_pair_dz) instead of z and adds the
   * dW2 / db1 partial sums to `pair_dz_ws` (same workspace layout as peneo_pair_dz, row = m-tile index mod 256).
   * No other epilogue option may be combined with it; split_k must be 1. */
  const struct peneo_pair_dz_args* pair_dz;
  float* pair_dz_ws;
  /* wgrad fusion (a_kmajor = 0, b_kmajor = 0, bf16): a_colsum[m] += sum_k A(m, k) -- with A = dy [tokens, M] this is the
   * bias gradient db = dy^T 1 of the same nn.Linear (autograd of modeling_layoutlmv3.py:292-294 and the Roberta dense
   * layers), taken from the A tiles the weight-gradient product already holds in LDS (one extra MFMA against a ones
   * operand in the workgroups of the first tile column; fp32 atomics, one per row and split slice).  [M] fp32 or NULL. */
  float* a_colsum;
} peneo_gemm_epilogue;

size_t peneo_gemm_workspace_bytes(int M, int N, int K, int split_k);
int peneo_gemm(int dtype, int a_kmajor, int b_kmajor, int M, int N, int K,
               const void* A, int64_t lda, const void* B, int64_t ldb,
               void* C, int64_t ldc, int c_dtype, const peneo_gemm_epilogue* ep,
               int split_k, void* workspace, size_t workspace_bytes, peneo_stream_t stream);

/* Up to 4 independent bf16 GEMMs C_i = op(A_i) op(B_i) of ONE operand layout in ONE launch (tiles of all problems side by
 * side, full K per tile, no split-k, no epilogue besides fp32 accumulate).  Operands must satisfy the LDS-DMA conditions
 * of peneo_gemm's bf16 path (16-byte aligned bases and row strides).  Used for the four weight gradients
 * dW = dy^T x of an encoder layer (autograd of the nn.Linear calls at modeling_layoutlmv3.py:292-294,335-360). */
typedef struct peneo_gemm_problem {
  int M, N, K;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  int accumulate;           /* C += A B (c_dtype must be PENEO_F32) */
  const peneo_gemm_epilogue* ep;   /* optional fused epilogue of THIS problem (bias, activation, pre-activation store, x act'(src),
                                    * dropout, residual: as for peneo_gemm; no pair_dz, no a_colsum); NULL = none.  Round 4: LiLT's
                                    * text and layout streams (modeling_lilt.py:269-429, H = 768 and H / 4 = 192) run each pair of
                                    * same-role Linear layers as one launch -- a [4096, 192] x [192, 576] GEMM alone is all fixed cost */
} peneo_gemm_problem;
int peneo_gemm_group(int dtype, int a_kmajor, int b_kmajor, int c_dtype, const peneo_gemm_problem* problems, int n,
                     peneo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise plumbing
 * ------------------------------------------------------------------------------------------ */
int peneo_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, peneo_stream_t stream);
/* Many contiguous fp32 -> bf16 casts in one launch: the bf16 working copies of the fp32 master weights that every nn.Linear of
 * the reference would read (modeling_layoutlmv3.py:292-294,335-360), refreshed after an optimizer step.  `table_dev` and the
 * two chunk maps (chunk c = elements [chunk_index[c] * peneo_cast_multi_chunk_elems(), ...) of item chunk_item[c]) live in
 * device memory and are built once per set of tensors. */
typedef struct peneo_cast_item { const float* src; void* dst; int64_t numel; } peneo_cast_item;
int peneo_cast_multi_chunk_elems(void);
int peneo_cast_multi(const peneo_cast_item* table_dev, const int32_t* chunk_item_dev, const int32_t* chunk_index_dev, int n_chunks,
                     peneo_stream_t stream);
/* dst[r, c] = src[r, c] for a strided 2-D block (row strides in elements), with optional dropout */
int peneo_copy2d(int dtype, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int64_t cols,
                 float drop_p, uint32_t drop_seed, peneo_stream_t stream);
/* batched variant: logical row r of side X lives at (r / X_rpb) * X_bstride + (r % X_rpb) * ld_X elements
 * (X_rpb = 0: r * ld_X).  Used for the CLS / visual-token crop of model/modeling_peneo.py:138-163. */
int peneo_copy_rows(int dtype, const void* src, int64_t src_rpb, int64_t src_bstride, int64_t ld_src,
                    void* dst, int64_t dst_rpb, int64_t dst_bstride, int64_t ld_dst, int64_t rows, int64_t cols,
                    float drop_p, uint32_t drop_seed, peneo_stream_t stream);
/* LiLT (modeling_lilt.py:269-429) sums text and layout attention scores (BiACM).  With per-head concatenation
 *   out[r, h*(da+db) + c] = c < da ? scale_a * a[r, h*da + c] : scale_b * b[r, h*db + (c - da)]
 * of q (pre-scaled by 1/sqrt(d), 1/sqrt(d_l)), k and v, one attention call over head dim da+db yields both
 * context streams; peneo_head_split is the inverse (used for the context and for dq/dk/dv). */
int peneo_head_concat(int dtype, const void* a, int64_t lda, int da, float scale_a, const void* b, int64_t ldb, int db,
                      float scale_b, void* out, int64_t ldo, int64_t rows, int nh, peneo_stream_t stream);
int peneo_head_split(int dtype, const void* in, int64_t ldi, void* a, int64_t lda, int da, float scale_a, void* b,
                     int64_t ldb, int db, float scale_b, int64_t rows, int nh, peneo_stream_t stream);
/* out[n] (+)= sum_m x[m, n]   (bias gradients) */
int peneo_colsum(int dtype, const void* x, int64_t ldx, int64_t M, int64_t N, float* out, int accumulate,
                 peneo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (modeling_layoutlmv3.py:225,930,1113; Roberta*Output.LayerNorm).
 * Row r of the logical [rows, H] matrix lives at  (r / rpb) * bstride + (r % rpb) * H  elements
 * (rpb = 0 means contiguous), separately for input and output, so that sub-ranges of a
 * [B, T, H] buffer can be normalised in place.
 * ------------------------------------------------------------------------------------------ */
int peneo_layernorm_fwd(int dtype, const void* x, int64_t x_rpb, int64_t x_bstride,
                        void* y, int64_t y_rpb, int64_t y_bstride,
                        const float* gamma, const float* beta, float eps,
                        float* mean, float* rstd, int64_t rows, int H,
                        float drop_p, uint32_t drop_seed, peneo_stream_t stream);
/* dx may alias dy.  dgamma/dbeta (fp32, [H]) are accumulated into.
 * dx_dropped (may be NULL): a second, contiguous [rows, H] output = dx through the dropout mask (drop2_p, drop2_seed,
 * element index r*H + c) of the GEMM that produced this LayerNorm's input — the gradient that GEMM's dgrad/wgrad need,
 * without a separate masking pass.
 * dx_colsum (may be NULL; fp32 [H], accumulated into): the column sums over the rows of dx_dropped (of dx when dx_dropped is
 * NULL), summed in fp32 before the rounding to the output dtype = the BIAS gradient of the Linear that produced the
 * LayerNorm's input (modeling_layoutlmv3.py:482-529: dense -> dropout -> + residual -> LayerNorm), without a column-sum launch. */
int peneo_layernorm_bwd(int dtype, const void* dy, int64_t dy_rpb, int64_t dy_bstride,
                        const void* x, int64_t x_rpb, int64_t x_bstride,
                        void* dx, int64_t dx_rpb, int64_t dx_bstride,
                        const float* gamma, const float* mean, const float* rstd,
                        float* dgamma, float* dbeta, int64_t rows, int H,
                        float drop_p, uint32_t drop_seed, void* dx_dropped, float drop2_p, uint32_t drop2_seed,
                        float* dx_colsum, peneo_stream_t stream);
/* The same backward with the parameter gradients left as per-workgroup partial sums, partials[P][2][H] (fp32; row 0 of a
 * pair = gamma, row 1 = beta), P = peneo_layernorm_bwd_partial_rows(dtype, rows, H); 0 means this dtype / row length has
 * no such form (use peneo_layernorm_bwd).  The caller reduces them with peneo_colsum over [P, 2H] whenever it likes (the
 * model does it on its weight-gradient stream): no same-address atomics at the end of the kernel, one row per half-wave.
 * All row bases 16-byte aligned. */
int64_t peneo_layernorm_bwd_partial_rows(int dtype, int64_t rows, int H);
int peneo_layernorm_bwd_partial(int dtype, const void* dy, int64_t dy_rpb, int64_t dy_bstride,
                                const void* x, int64_t x_rpb, int64_t x_bstride,
                                void* dx, int64_t dx_rpb, int64_t dx_bstride,
                                const float* gamma, const float* mean, const float* rstd,
                                float* partials, int64_t partial_rows, int64_t rows, int H,
                                float drop_p, uint32_t drop_seed, void* dx_dropped, float drop2_p, uint32_t drop2_seed,
                                peneo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K1 — text embeddings (modeling_layoutlmv3.py:131-227, modeling_lilt.py:75-130,160-210)
 * ------------------------------------------------------------------------------------------ */
/* RoBERTa position ids: cumsum(ids != pad) * (ids != pad) + pad   (:164-181) */
int peneo_position_ids(const int64_t* input_ids, int B, int S, int64_t pad_id, int32_t* pos_ids,
                       peneo_stream_t stream);
/* out[b,s,:] = word[ids] + type0 + pos[pid] (+ cat(x[l],y[t],x[r],y[b],h[b-t],w[r-l]) if spatial) ;
 * returns PENEO_ERR_INVALID through `status` (device int, may be NULL) when a coordinate is outside
 * [0, max_2d) — the reference raises IndexError (:133,138-141).
 * clip_hw != 0 clips the h/w lookups to [0, max_2d-1] (LayoutLMv3); LiLT does not clip.         */
typedef struct peneo_embed_tables {
  const float* word;   /* [vocab, H] */
  const float* type0;  /* [H] row 0 of token_type_embeddings */
  const float* pos;    /* [max_pos, H] */
  const float* x; const float* y; const float* h; const float* w; /* [max_2d, coord|shape]; NULL = no spatial part */
  int coord_size; int shape_size; int max_2d; int vocab; int max_pos;
} peneo_embed_tables;
int peneo_embed_text_fwd(int dtype, const int64_t* input_ids, const int32_t* pos_ids, const int64_t* bbox,
                         const peneo_embed_tables* tab, int B, int S, int H, int clip_hw,
                         void* out, int64_t out_rpb, int64_t out_bstride, int32_t* status, peneo_stream_t stream);
/* scatter-add of d_out into fp32 table gradients (dense, like nn.Embedding(sparse=False));
 * rows equal to pad_id are skipped for word/pos (padding_idx semantics). */
typedef struct peneo_embed_grads {
  float* word; float* type0; float* pos; float* x; float* y; float* h; float* w;
} peneo_embed_grads;
int peneo_embed_text_bwd(int dtype, const void* d_out, int64_t rpb, int64_t bstride,
                         const int64_t* input_ids, const int32_t* pos_ids, const int64_t* bbox,
                         const peneo_embed_grads* g, int coord_size, int shape_size, int max_2d,
                         int B, int S, int H, int clip_hw, int64_t pad_id, peneo_stream_t stream);
/* spatial-only embedding for LiLT: out[b,s,:] = cat(x[l],y[t],x[r],y[b],h[b-t],w[r-l]) */

/* ------------------------------------------------------------------------------------------
 * K2 — patch embedding (modeling_layoutlmv3.py:51-84, 910-931)
 * ------------------------------------------------------------------------------------------ */
/* image [B, C, Hi, Wi] fp32 -> patches [B * (Hi/16) * (Wi/16), C*256] (k = c*256 + py*16 + px) */
int peneo_im2col_patch16(int dtype, const float* image, int B, int C, int Hi, int Wi, void* patches,
                         peneo_stream_t stream);
/* vis[b,0,:] = cls + pos[0]; vis[b,1+p,:] = proj[b*np+p,:] + pos[1+p]  */
int peneo_visual_assemble_fwd(int dtype, const void* proj, const float* cls, const float* pos, int B, int np, int H,
                              void* vis, peneo_stream_t stream);
/* d_proj <- d_vis[:,1:]; d_cls += sum_b d_vis[b,0]; d_pos += sum_b d_vis[b] */
int peneo_visual_assemble_bwd(int dtype, const void* d_vis, int B, int np, int H, void* d_proj, float* d_cls,
                              float* d_pos, peneo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K4 — relative-position bias (modeling_layoutlmv3.py:586-676), computed once per forward.
 * lut[|delta|] holds the *unsigned* part of relative_position_bucket (the host fills it with
 * the reference's own fp32 formula so bucket indices are bit-exact); the sign adds nb/2.
 * ------------------------------------------------------------------------------------------ */
/* Per-token inputs of peneo_relpos_buckets and the attention key mask over the concatenated text + visual sequence
 * (T = S + nv; modeling_layoutlmv3.py:1052-1080, 586-676): one launch.  int32 [B, T] outputs, each may be NULL;
 * attention_mask int64 [B, S] (NULL = all ones), bbox int64 [B, S, 4], vx / vy int32 [nv] grid coordinates of the patches. */
int peneo_relpos_inputs(const int64_t* attention_mask, const int64_t* bbox, const int32_t* vx, const int32_t* vy, int B, int S,
                        int nv, int32_t* key_mask, int32_t* pos, int32_t* xs, int32_t* ys, peneo_stream_t stream);
int peneo_relpos_buckets(const int32_t* pos, const int32_t* xs, const int32_t* ys, int B, int T,
                         const uint8_t* lut1, int lut1_len, int half1,
                         const uint8_t* lut2, int lut2_len, int half2,
                         uint8_t* bk1, uint8_t* bkx, uint8_t* bky, peneo_stream_t stream);
/* bias[b,h,i,j] = scale * (w1[h,bk1] + wx[h,bkx] + wy[h,bky]); any of the three may be NULL.
 * Output layout [B, nh, T, Tp] with Tp = peneo_attn_padded_len(T); columns j >= T and keys with
 * key_mask[b,j] == 0 (may be NULL) hold -1e30, which is how the attention kernels see the padding mask. */
int peneo_relpos_bias_fwd(int dtype, const uint8_t* bk1, const uint8_t* bkx, const uint8_t* bky,
                          const float* w1, int bins1, const float* wx, const float* wy, int bins2,
                          float scale, int B, int nh, int T, int Tp, const int32_t* key_mask, void* bias,
                          peneo_stream_t stream);
/* dw*[h,bin] += scale * sum_{b,i,j in bin} g[b,h,i,j]   (g rows have stride ldg >= T) */
/* Bias-table gradients from L per-layer bf16 dS^T buffers (layout of peneo_attn_bwd's ds_out, `layer_stride`
 * elements apart): dS is summed over the layers in fp32, then binned.  The bucket maps are the TRANSPOSED ones,
 * bkT[b, j, i] = bucket(i, j) (peneo_relpos_buckets on negated positions yields exactly that), stored with row
 * stride Tp: uint8 [B, T, Tp] (columns >= T are ignored), so that 8 buckets are one aligned 8-byte load. */
int peneo_relpos_bias_bwd_layers(const void* ds, int L, int64_t layer_stride, const uint8_t* bk1_t, const uint8_t* bkx_t,
                                 const uint8_t* bky_t, float* dw1, int bins1, float* dwx, float* dwy, int bins2,
                                 float scale, int B, int nh, int T, int Tp, peneo_stream_t stream);
int peneo_relpos_bias_bwd(const float* g, int64_t ldg, const uint8_t* bk1, const uint8_t* bkx, const uint8_t* bky,
                          float* dw1, int bins1, float* dwx, float* dwy, int bins2,
                          float scale, int B, int nh, int T, peneo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K6 — attention core (modeling_layoutlmv3.py:365-404; the "cogview" softmax :308-321 is exactly
 * softmax).  q/k/v are rows of a [B*T, ld] matrix, head h at columns h*d..;
 *   scores = scale * q.k + bias[b,h,i,j] + key_bias[b,j]      (both optional; -1e30 = masked key)
 * Operands whose reduction index must be contiguous are passed as per-head transposed copies
 * [B, nh, DP, Tp] (DP = peneo_attn_padded_dim(d), Tp = peneo_attn_padded_len(T)) made by
 * peneo_head_transpose: vt for the forward; kt, qt and dot (= d_out transposed) for the backward.
 * `lse` [B, nh, T] fp32 is the per-row log-sum-exp in log2 units (an opaque forward->backward buffer).
 *
 * Dropout on the attention probabilities (modeling_layoutlmv3.py:396-399): the keep bits of one call are made by
 * peneo_attn_drop_words - a pure function of (seed, word index), Bernoulli(1 - p) per bit with p realised to 2^-17 - into
 * `words`, uint32 [B * nh][n_query_blocks][n_key_slots] (peneo_attn_drop_words_dims), and handed to the
 * forward and to the backward of that call (`drop_words`; NULL with drop_p = 0).  One dword holds the 32 queries of a
 * block for one key; the key slots of a 32-key block are ordered as the forward kernel's accumulator registers see them,
 * so the kernels read whole select masks instead of hashing per element (csrc/attention.hip, top).
 * ------------------------------------------------------------------------------------------ */
int peneo_attn_padded_len(int T);
int peneo_attn_padded_dim(int d);
int peneo_head_transpose(int dtype, const void* src, int64_t ld, int B, int nh, int T, int d, void* dst,
                         peneo_stream_t stream);
/* v / vt: bf16 takes V in place (row stride ld_qk, like q and k) and reads V^T with the hardware transpose read, vt may
 * be NULL; fp32 needs vt, the per-head transposed copy from peneo_head_transpose (v is then ignored). */
int peneo_attn_fwd(int dtype, const void* q, const void* k, const void* v, int64_t ld_qk, const void* vt,
                   int B, int nh, int T, int d, float scale, const void* bias, int64_t bias_ld,
                   const float* key_bias, void* out, int64_t ld_out, float* lse, float drop_p, const uint32_t* drop_words,
                   peneo_stream_t stream);
void peneo_attn_drop_words_dims(int T, int* n_query_blocks, int* n_key_slots);
int64_t peneo_attn_drop_words_count(int B, int nh, int T);   /* = B * nh * n_query_blocks * n_key_slots */
int peneo_attn_drop_words(uint32_t* words, int B, int nh, int T, float drop_p, uint32_t seed, peneo_stream_t stream);
/* dq/dk/dv share the q/k/v layout (ld_dqkv).  g_bias (fp32 [B, nh, T, bias_ld], may be NULL) is
 * accumulated with dS so the bias-table gradient can be reduced once per step.
 * `delta` is a [B, nh, T] fp32 scratch.
 * Implementations: with dtype bf16 and `ds_out` and/or `dq_accum` the single-pass kernel runs (S / dP computed once,
 * dK / dV in registers; kt / qt / dot are not read and may be NULL).  dQ then comes from a second kernel that multiplies
 * the stored dS^T slab with K (ds_out given, dq_accum NULL: the faster form) or from fp32 atomics into `dq_accum`
 * (fp32 scratch [B*T, nh*d], overwritten).  Otherwise (fp32, or both NULL) the dQ kernel + the dK/dV kernel run and
 * need the per-head transposed copies kt / qt / dot from peneo_head_transpose.
 * ds_out (single-pass only, may be NULL): bf16 [B, nh, T keys, Tp queries] receives this layer's dS^T (key-major,
 * unscaled); with one such buffer per layer the bias-table gradient is reduced once per step by
 * peneo_relpos_bias_bwd_layers instead of a read-modify-write of g_bias in every layer. */
int peneo_attn_bwd(int dtype, const void* q, const void* k, const void* v, int64_t ld_qkv,
                   const void* kt, const void* qt, const void* dot,
                   const void* out, const void* d_out, int64_t ld_out, const float* lse,
                   int B, int nh, int T, int d, float scale, const void* bias, int64_t bias_ld, const float* key_bias,
                   void* dq, void* dk, void* dv, int64_t ld_dqkv, float* g_bias, float* delta, float* dq_accum,
                   void* ds_out, float drop_p, const uint32_t* drop_words, peneo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K11 + K12 (+ K13) — handshaking + the pair-classifier heads + class-weighted CE, fused
 * (model/peneo_decoder.py:149-177, 231-292, 315-336, 355-428; model/custom_loss.py:189-202).
 *
 * ab [B, N, 2D]: a_i = W_c[:, :D] h_i  (cols 0..D) and b_j = W_c[:, D:] h_j + bias_c (cols D..2D),
 * so that  combine_fc(cat(h_i, h_j)) = a_i + b_j.  For every packed upper-triangular pair
 * p(i,j) = i*N - i(i-1)/2 + (j-i):  x = SiLU(a_i + b_j);  per head: SiLU(W1_h x + b1_h) -> W2_h . + b2_h.
 * ------------------------------------------------------------------------------------------ */
typedef struct peneo_pair_heads_desc {
  int num_heads;                       /* <= PENEO_MAX_HEADS */
  int D;                               /* decoder hidden size (multiple of 32) */
  int classes[PENEO_MAX_HEADS];        /* 2,3,3,3,3 */
  const void* w_packed;                /* from peneo_pair_heads_pack (both layers, MFMA fragment order) */
  const float* b1;                     /* [num_heads * D] */
  const float* b2;                     /* [sum classes] */
  float drop_p;                        /* K12 dropout between the two classifier layers (model/peneo_decoder.py:261); 0 = off.
                                          keep(b, p, n) is a pure function of (drop_seed, document, pair, hidden column):
                                          the backward entry points regenerate it from the same two numbers.  Realised as
                                          round(p * 2^16) / 2^16 with the matching 1 / (1 - p) scale */
  uint32_t drop_seed;
} peneo_pair_heads_desc;

size_t peneo_pair_heads_packed_bytes(int dtype, int num_heads, int D);
/* w1[h] : [D, D] row-major fp32 (nn.Linear weight), w2[h] : [classes[h], D] fp32;
 * w1 / w2 / classes are HOST arrays (of device pointers / ints) with num_heads entries */
int peneo_pair_heads_pack(int dtype, const float* const* w1, const float* const* w2, const int* classes,
                          int num_heads, int D, void* packed, peneo_stream_t stream);

typedef struct peneo_pair_loss {
  const int64_t* tags[PENEO_MAX_HEADS];   /* [B, P] label maps (data/collator.py:170-204); NULL = no loss */
  const float* class_weight[PENEO_MAX_HEADS]; /* [classes[h]] fp32 device */
  float* partials;                        /* [peneo_pair_loss_partials(B, N)][32] fp32 workspace: one row per workgroup =
                                             sum_p w[tag]*nll [8] | sum_p w[tag] [8] | sum_p dlogits [16]; reduced by
                                             peneo_loss_finish (deterministic, no atomics) */
  float* dlogits[PENEO_MAX_HEADS];        /* optional [B, P, classes[h]] fp32: w[tag] * (softmax - onehot) (unnormalised) */
} peneo_pair_loss;
int64_t peneo_pair_loss_partials(int B, int N);

/* logits[h] : [B, P, classes[h]] fp32, contiguous, P = N (N + 1) / 2 (may be NULL when only the loss is wanted) */
int peneo_pair_heads_fwd(int dtype, const void* ab, int B, int N, const peneo_pair_heads_desc* desc,
                         float* const* logits /* host array of num_heads device pointers, or NULL */,
                         const peneo_pair_loss* loss, peneo_stream_t stream);

/* The same launch for a train step whose backward is peneo_pair_bwd_saved (bf16, D = 384: peneo_pair_save_supported).  It walks the
 * pairs in the backward's blocks of 8 rows i x 16 columns j and leaves what the backward would otherwise rebuild:
 *   act    [peneo_pair_save_bytes(B, N, num_heads, D)]: per document, block, 32-unit slab of the num_heads * D hidden units and
 *          group of 32 pairs one 2 KiB record: the classifiers' pre-activations z = W1_h x + b1_h as f16 [32 pairs][32 units], a
 *          unit the K12 dropout drops stored as -30000 (SiLU and SiLU' of that are 0: the backward needs no mask and runs no dropout
 *          chain) - what autograd under autocast saves of model/peneo_decoder.py:253-271, 2 bytes per pair and hidden unit;
 *   x_rows [B * peneo_pair_bwd_rows(N), D] bf16: x = SiLU(a_i + b_j) in block order (the B operand of dW1 = dz^T x).
 * Logits, loss partial rows and dlogits are those of peneo_pair_heads_fwd bit for bit (same arithmetic per pair, indexed by the
 * packed pair index); `loss->partials` has peneo_pair_loss_partials_save(B, N) rows here.  16-byte aligned buffers.
 * RANGE: z is stored as f16 and a dropped unit as -30000 + W1 x, so the saved form assumes |W1 x + b1| < 30000 for every pair and
 * hidden unit (then a dropped z stays below -20 where SiLU and SiLU' are 0 in f16, and nothing reaches the f16 limit 65504: an
 * overflow to -inf would give -inf * 0 = NaN in peneo_pair_bwd_saved).  With LayerNorm-ed inputs and initializer_range-scale
 * weights |z| is O(10); a caller that cannot bound it runs peneo_pair_heads_fwd + peneo_pair_bwd_fused (the recomputing form has
 * no such limit; PEneoDecoder.save_pair_act = False / PENEO_PAIR_SAVE=0). */
int peneo_pair_save_supported(int dtype, int D, int num_heads);
size_t peneo_pair_save_bytes(int B, int N, int num_heads, int D);
int64_t peneo_pair_loss_partials_save(int B, int N);
int peneo_pair_heads_fwd_save(int dtype, const void* ab, int B, int N, const peneo_pair_heads_desc* desc,
                              float* const* logits, const peneo_pair_loss* loss, void* act, void* x_rows,
                              peneo_stream_t stream);

/* --- building blocks of the chunked backward (rows i0..i1 of the pair triangle = pairs
 *     p(i0,i0) .. p(i1,i1)-1 of one document) ------------------------------------------- */
/* x[p - p0, :] = SiLU(a_i + b_j)  [npairs, D];  pre (may be NULL) receives a_i + b_j itself, the `grad_src` that lets the
 * dx GEMM epilogue apply SiLU' (peneo_gemm_epilogue.grad_src / grad_act) */
int peneo_pair_x_fwd(int dtype, const void* ab_doc, int N, int D, int i0, int i1, void* x, void* pre, peneo_stream_t stream);
/* du = dx * SiLU'(a_i + b_j)  (or du = dx when `premultiplied`: the GEMM epilogue already applied the factor);
 * d_ab_doc[i, :D] += sum_j du ; d_ab_doc[j, D:] += sum_i du   (fp32 [N, 2D]) */
int peneo_pair_x_bwd(int dtype, const void* ab_doc, int N, int D, int i0, int i1, const void* dx, float* d_ab_doc,
                     int premultiplied, peneo_stream_t stream);
/* For the [npairs, nh*D] pre-activations z of all heads' first layers (in place):
 *   y = SiLU(z);  dy[p, h*D+k] = sum_c scale_h * dlogits_h[p, c] * w2_h[c, k];  dz = dy * SiLU'(z)  (written over z)
 * and the reductions over pairs needed by the parameter gradients are accumulated into `workspace`
 * [peneo_pair_dz_workspace_bytes / (16 * nh*D)][4 * nh*D] fp32 (zero it once per step; sum its rows with peneo_colsum at the end):
 *   row block c (c = 0..2): sum_p scale_h*dlogits_h[p,c] * y[p, :]   -> dW2_h[c, :] = block[c][h*D : (h+1)*D]
 *   row block 3            : sum_p dz[p, :]                          -> db1
 * scale[h] (= d(loss) * loss_ratio_h / den_h) is a device vector. */
typedef struct peneo_pair_dz_args {
  int num_heads; int D; int classes[PENEO_MAX_HEADS];
  const float* dlogits[PENEO_MAX_HEADS];  /* [npairs, classes[h]] (already offset to the chunk) */
  const float* w2[PENEO_MAX_HEADS];       /* [classes[h], D] fp32 */
  const float* scale;                     /* [num_heads] device fp32 */
  float drop_p;                           /* the forward's K12 dropout (peneo_pair_heads_desc): y and dz are masked and scaled */
  uint32_t drop_seed;
  int drop_doc;                           /* chunked entry points: document index b and packed index p of the chunk's first */
  int64_t drop_pair0;                     /* pair, i.e. what the forward hashed (peneo_pair_bwd_fused walks all of them itself) */
} peneo_pair_dz_args;
size_t peneo_pair_dz_workspace_bytes(int num_heads, int D);
int peneo_pair_dz(int dtype, void* z_inout, int64_t npairs, const peneo_pair_dz_args* args, float* workspace,
                  peneo_stream_t stream);

/* dz of the pairs of rows i0..i1 of one document WITHOUT x or z in memory (bf16 only): x = SiLU(a_i + b_j) is rebuilt in
 * registers, z = x W1cat^T + b1 comes off the matrix cores with lane = hidden column, and the same kernel turns it into
 * dz [npairs, nh*D] and adds the dW2 / db1 column sums to `workspace` (layout of peneo_pair_dz; only its first 256 rows
 * are touched, as by the pair_dz epilogue of peneo_gemm: 256 rows suffice for a workspace these two share).  `w_packed` is the
 * fragment-packed weight buffer of peneo_pair_heads_pack (its second-layer fragments are not used here).
 * With x_out / pre_out it also leaves x = SiLU(a_i + b_j) and a_i + b_j for the dW1 / dx GEMMs (no peneo_pair_x_fwd pass).
 * Replaces peneo_pair_x_fwd + the first-layer GEMM + peneo_pair_dz of the chunked backward
 * (the autograd graph through model/peneo_decoder.py:269-336). */
int peneo_pair_dz_fused(int dtype, const void* ab_doc, int N, int D, int i0, int i1, const void* w_packed, const float* b1,
                        const peneo_pair_dz_args* args, void* dz, float* workspace,
                        void* x_out /* optional [npairs, D]: what peneo_pair_x_fwd would write */,
                        void* pre_out /* given together with x_out */, peneo_stream_t stream);

/* Decoder backward through the pair space in ONE kernel (bf16; D / 16 in {2, 4, 8, 24, 32}): for all pairs (i, j), i <= j, of
 * all B documents it rebuilds x = SiLU(a_i + b_j) in registers, computes z = x W1cat^T + b1 and dz (as peneo_pair_dz_fused),
 * du = dz W1cat on the matrix cores with the accumulator kept in registers over all hidden units, and
 *   d_ab[b, i, :D] = sum_j du * SiLU'(a_i + b_j),  d_ab[b, j, D:] = sum_i du * SiLU'(a_i + b_j)   (fp32, overwritten; per-block
 *   partial rows + one reduction launch, no atomics),
 * plus the dW2 / db1 column sums into `workspace` (layout of peneo_pair_dz, first 256 rows).  It leaves dz [B * rows, nh*D]
 * and x [B * rows, D] (bf16) for the one remaining GEMM dW1cat = dz^T x.  rows = peneo_pair_bwd_rows(N): the pairs are
 * walked in blocks of 8 rows i x 16 columns j of the triangle (blocks on the diagonal / the last row of blocks carry
 * pairs outside it: their dz rows are zero, their x rows finite), both buffers in that block order.
 * args->dlogits[h] is the whole [B, P, classes[h]] map, args->scale the device vector of peneo_loss_finish.
 * Replaces, per document, peneo_pair_x_fwd + peneo_pair_dz_fused + the du GEMM (grad_src epilogue) + peneo_pair_x_bwd:
 * the autograd graph through model/peneo_decoder.py:149-177 and :231-292 except the first-layer weight gradient.
 * `w_packed`: peneo_pair_bwd_pack (per 32-unit slab the B-operand fragments of both products).
 * peneo_pair_bwd_supported: 1 when the fused launch exists for (dtype, D) AND its LDS image fits `num_heads` heads (the per-column
 * table grows with the head count: D = 512 holds 5 heads, not 6); num_heads <= 0 asks about (dtype, D) only.  On 0 a caller runs
 * the per-document chain (peneo_pair_x_fwd / peneo_pair_dz_fused / peneo_gemm / peneo_pair_x_bwd). */
int peneo_pair_bwd_supported(int dtype, int D, int num_heads);
int64_t peneo_pair_bwd_rows(int N);
size_t peneo_pair_bwd_packed_bytes(int num_heads, int D);
int peneo_pair_bwd_pack(const float* const* w1 /* host array of num_heads device pointers to [D, D] fp32 */, int num_heads, int D,
                        void* packed, peneo_stream_t stream);
size_t peneo_pair_bwd_partial_bytes(int B, int N, int D);   /* `partials`: per-block partial rows of d_a / d_b (scratch) */
int peneo_pair_bwd_fused(int dtype, const void* ab, int B, int N, int D, const void* w_packed, const float* b1,
                         const peneo_pair_dz_args* args, void* dz, void* x, float* d_ab, float* workspace,
                         float* partials, peneo_stream_t stream);

/* peneo_pair_bwd_fused for a forward that ran as peneo_pair_heads_fwd_save (bf16, D = 384): `act` holds the pre-activation z of
 * every pair and hidden unit (dropout applied), so this launch computes y = SiLU(z) and dz = (g W2 / (1 - p)) SiLU'(z) from it, runs
 * neither the first-layer product (24 of the 51 MFMAs per 32 pairs x 32 units) nor a dropout chain, and leaves the same dz
 * [B * rows, nh*D], d_ab and workspace sums (x_rows was written by the forward).  `w_packed`: peneo_pair_bwd_pack.  Results differ
 * from peneo_pair_bwd_fused by the f16 rounding of the saved z (what autograd under autocast would have saved; 5e-4 relative on dz). */
int peneo_pair_bwd_saved(int dtype, const void* ab, int B, int N, int D, const void* w_packed,
                         const peneo_pair_dz_args* args, const void* act, void* dz, float* d_ab, float* workspace,
                         float* partials, peneo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K13 — loss finish: reduces the per-workgroup partial rows of peneo_pair_heads_fwd:
 *   loss_h = num_h / den_h ; total = sum_h ratio_h * loss_h ; scale_h = ratio_h / den_h (the factor the
 *   backward applies to the un-normalised dlogits) ; dl_sum[c] = sum_p dlogits[p, c] (second-layer bias grads).
 * out: [num_heads + 1] = per-head losses then the total.  (model/peneo_decoder.py:315-336,375-428)
 * inv_den (optional, [num_heads]) = 1 / den_h: the factor for a gradient arriving on a per-head loss itself.
 * ------------------------------------------------------------------------------------------ */
int peneo_loss_finish(const float* partials, int64_t n_partials, const float* ratio, int num_heads, int total_classes,
                      float* out, float* scale, float* dl_sum, float* inv_den, peneo_stream_t stream);
/* K13, OHEM branch ("next" row f.3): CrossEntropyLossOHEM.forward with num_hard_positive / num_hard_negative != -1
 * (model/custom_loss.py:204-288) as executed by the reference, over the n = B*P flattened pairs of one head:
 * per-pair weighted CE, positives (tag != 0) and negatives sorted by descending loss (stable: equal losses keep the
 * flattened order), k = min(count, num_hard); k <= 0 or k == count keeps every element of the class, otherwise the kept
 * elements are sorted[idx[:k]] (the reference indexes the sorted array with unsorted positions; reproduced).
 * out8 = [loss, sum of kept losses, k_pos + k_neg, n_pos, n_neg, k_pos, k_neg, 0]; dlogits (optional, [n, C], the
 * un-normalised w_y (softmax - onehot) written by peneo_pair_heads_fwd) is zeroed on the dropped pairs in place and
 * dl_sum[C] receives its column sums.  Everything stays on the device (no host read-back of the counts).
 * Workspace: peneo_ohem_workspace_bytes(n) bytes, 256-byte aligned. */
size_t peneo_ohem_workspace_bytes(int64_t n);
int peneo_ohem_ce(const float* logits, const int64_t* tags, const float* class_weight, int64_t n, int C,
                  int num_hard_positive, int num_hard_negative, float* dlogits, float* out8, float* dl_sum,
                  void* workspace, size_t workspace_bytes, peneo_stream_t stream);
/* per-head out8 rows [num_heads, 8] -> out[num_heads + 1] (losses, then sum_h ratio_h loss_h), scale_h = ratio_h / den_h,
 * inv_den_h = 1 / den_h: the OHEM counterpart of peneo_loss_finish */
int peneo_ohem_finish(const float* out8, const float* ratio, int num_heads, float* out, float* scale, float* inv_den,
                      peneo_stream_t stream);
/* stand-alone weighted CE on materialised logits [rows, C] (used by the unfused parity path) */
int peneo_weighted_ce(const float* logits, const int64_t* tags, const float* class_weight, int64_t rows, int C,
                      float* num, float* den, float* dlogits, peneo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K14 — decode front end (model/peneo_decoder.py:98-114): softmax -> argmax/max -> compaction of
 * the non-zero tags of one [P, C] map into (i, j, tag, score) spots in increasing p order.
 * count: device int (number found, may exceed max_spots; only max_spots are stored).
 * ------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------
 * Optimizer step ("next" row f.4): fused multi-tensor AdamW with per-tensor learning rate and weight decay,
 * i.e. the four parameter groups of PEneoTrainer.create_optimizer (pipeline/trainer.py:275-330: decoder
 * parameters at lr x peneo_downstream_speedup_ratio, no decay on biases / LayerNorm) in ONE launch.
 * Arithmetic of torch.optim.AdamW (decoupled decay, bias-corrected moments).  The table and the two chunk maps
 * (chunk c covers elements [chunk_index[c] * peneo_adamw_chunk_elems(), ...) of tensor chunk_tensor[c]) live in
 * device memory and are built once.  `step` counts from 1; a tensor whose own count differs (state resumed from a
 * checkpoint with per-parameter steps, a parameter that joined later) carries the difference in `step_offset`:
 * its bias corrections use step + step_offset (which must be >= 1).
 * ------------------------------------------------------------------------------------------ */
typedef struct peneo_adamw_tensor {
  float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
  int64_t numel; float lr; float weight_decay;
  int32_t step_offset; int32_t reserved;
} peneo_adamw_tensor;
int peneo_adamw_chunk_elems(void);
int peneo_adamw_step(const peneo_adamw_tensor* table_dev, const int32_t* chunk_tensor_dev, const int32_t* chunk_index_dev,
                     int n_chunks, float beta1, float beta2, float eps, int step, peneo_stream_t stream);
/* Global gradient-norm clipping inside the step (the reference trains through HF Trainer.train(), start/run_rfund.py:307-321,
 * whose default max_grad_norm = 1.0 clips every step with torch.nn.utils.clip_grad_norm_): peneo_grad_sqnorm leaves
 * sum_t |grad_t|^2 over all tensors of the table as peneo_grad_sqnorm_slots() partial sums in sqnorm_dev (device fp64 array,
 * zeroed by the call, stream-ordered, no host sync; their sum is the squared norm); peneo_adamw_step_clip then uses
 * grad * min(1, max_grad_norm / (sqrt(sum) + 1e-6)) in place of grad.  The gradient tensors are not modified.
 * sqnorm_dev == NULL: no clipping (== peneo_adamw_step). */
int peneo_grad_sqnorm_slots(void);
int peneo_grad_sqnorm(const peneo_adamw_tensor* table_dev, const int32_t* chunk_tensor_dev, const int32_t* chunk_index_dev,
                      int n_chunks, double* sqnorm_dev, peneo_stream_t stream);
int peneo_adamw_step_clip(const peneo_adamw_tensor* table_dev, const int32_t* chunk_tensor_dev, const int32_t* chunk_index_dev,
                          int n_chunks, float beta1, float beta2, float eps, int step, const double* sqnorm_dev,
                          float max_grad_norm, peneo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Composite stages: ONE call enqueues every kernel of an encoder layer (bf16 path) -- the per-kernel entry points above
 * called back to back from C++ (csrc/stages.hip), so that a layer costs the host one FFI call instead of 7 (forward) or
 * ~25 (backward).  Replaces LayoutLMv3Layer.forward and its autograd (modeling_layoutlmv3.py:482-529, :335-404; Roberta
 * SelfOutput / Intermediate / Output).  Every buffer is the caller's; rows = B * T; all matrices contiguous:
 *   x, att, h1, a, h2, out [rows, H]; qkv [rows, 3H] (q | k | v); zi, inter [rows, I]; lse [B, nh, T]; m*, r* [rows] (LayerNorm
 *   statistics); Wqkv [3H, H], Wo [H, H], Wi [I, H], Wo2 [H, I] bf16 working copies; biases / LayerNorm parameters fp32.
 * zi (GELU pre-activation) may be NULL in a forward that no backward follows.  Dropout: p_hidden with seed_o / seed_o2 on the
 * two dense + residual sites (mask = f(seed, element), regenerated by the backward), p_attn with `drop_words`
 * (peneo_attn_drop_words).  bias / key_bias as for peneo_attn_fwd.
 * ------------------------------------------------------------------------------------------ */
typedef struct peneo_encoder_layer {
  const void* Wqkv; const float* bqkv; const void* Wo; const float* bo; const float* g1; const float* b1;
  const void* Wi; const float* bi; const void* Wo2; const float* bo2; const float* g2; const float* b2;
  const void* bias; int64_t bias_ld; const float* key_bias; const uint32_t* drop_words;
  const void* x; void* qkv; void* att; float* lse; void* h1; float* m1; float* r1; void* a; void* zi; void* inter; void* h2;
  float* m2; float* r2;
  int32_t B, T, H, nh, I, reserved;
  float eps, attn_scale, p_hidden, p_attn;
  uint32_t seed_o, seed_o2;
} peneo_encoder_layer;
/* Backward: gradients of the activations (caller's scratch, all written) and of the parameters.  d_dense1 / d_dense2 are
 * only used with p_hidden > 0.  dw* fp32 [out, in] are overwritten; db*, dg* fp32 are ACCUMULATED into (zero them first);
 * ds_out bf16 [B, nh, T, Tp] receives this layer's dS^T (peneo_attn_bwd); delta [B, nh, T] fp32 scratch. */
typedef struct peneo_encoder_layer_grads {
  const void* d_out; void* d_x;
  void* d_h2; void* d_dense2; void* d_zi; void* d_a; void* d_h1; void* d_dense1; void* d_att; void* dqkv; float* delta; void* ds_out;
  float* dwqkv; float* dbqkv; float* dwo; float* dbo; float* dg1; float* db1; float* dwi; float* dbi; float* dwo2; float* dbo2;
  float* dg2; float* db2;
} peneo_encoder_layer_grads;
/* sizeof of the structs shared with a binding: 0 peneo_gemm_epilogue, 1 peneo_encoder_layer, 2 peneo_encoder_layer_grads */
size_t peneo_struct_bytes(int which);
/* split-k workspace of the layer's GEMMs: which = 0 forward, 1 backward main stream (dgrads), 2 backward side stream (wgrads) */
size_t peneo_encoder_layer_workspace_bytes(int rows, int H, int I, int which);
int peneo_encoder_layer_fwd(const peneo_encoder_layer* layer, void* out, void* workspace, size_t workspace_bytes,
                            peneo_stream_t stream);
/* The activation-gradient chain runs on `stream`; the parameter-gradient work (weight-gradient GEMMs, bias column sums) on
 * `side_stream` behind HIP events recorded on `stream` (the FFN / output-projection part starts with the attention backward,
 * the QKV part behind it).  The call does NOT join the two streams: the caller waits for `side_stream` before anything
 * reads dw* / db* (side_stream NULL or == stream: everything in order on one stream). */
int peneo_encoder_layer_bwd(const peneo_encoder_layer* layer, const peneo_encoder_layer_grads* grads, void* ws_main,
                            size_t ws_main_bytes, void* ws_side, size_t ws_side_bytes, peneo_stream_t stream,
                            peneo_stream_t side_stream);

/* K13 input side ("next" row f.2): dense label maps [B, P] int64 from n sparse spots (b, i, j, tag) — replaces the
 * host loop of HandshakingTaggingScheme.spots2shaking_tag4batch (model/peneo_decoder.py:35-73, called per head by
 * data/collator.py:156-204) and the 5.2 MB/document host-to-device copy of its result.  Last spot wins, like the
 * host loop; *status is set to 1 when a spot lies outside [0, B) x [0, N)^2. */
int peneo_spots_to_tags(const int32_t* spots_bijt, int n_spots, int B, int N, int64_t* tags, int32_t* status,
                        peneo_stream_t stream);
int peneo_spots_compact(const float* logits, int64_t P, int C, int N, int32_t* spots_ijt, float* scores,
                        int32_t* count, int max_spots, peneo_stream_t stream);

/* ---- diagnostics: NOT part of the thread-safety contract -------------------------------------------------------------
 * Process-wide switch of the forward GEMM tile choice (gemm_big.hip): 0 = always the 128 x 128 kernel, 1 = the calibrated
 * choice (default; also set by PENEO_GEMM_BIG), 256 / 384 / 128 = force one big-tile shape where its constraints hold.  A plain
 * global: set it while no peneo_gemm call is in flight on any thread.  Used by the kernel tests and tools/ only. */
void peneo_gemm_set_big_mode(int mode);
/* The same for the persistent stream-k launch (gemm_sk.hip; first choice of peneo_gemm for large bf16 problems with a k-major A):
 * 0 = off (the tiled kernels above), 1 = where the problem is large enough (default; also set by PENEO_GEMM_SK), 128 / 256 = force
 * the 256 x 128 / 256 x 256 tile wherever the kernel's constraints hold.  The launch keeps one fp32 slab per workgroup and a flag
 * word per workgroup in a buffer the library allocates per (device, stream) at the first call -- not during a stream capture: run
 * the captured sequence once eagerly first (a capture that meets a missing buffer falls back to the tiled kernels). */
void peneo_gemm_set_sk_mode(int mode);
/* tools/ only.  peneo_gemm_sk_set_prof: device buffer of [1024][16] uint64 that receives one lane's s_memrealtime stamps (100 MHz) at
 * the stations of every workgroup's range of a persistent launch (0 start, 1 stream primed, 2 first unit landed, 3 / 4 slab publish,
 * 5 / 6 / 7 flag wait and acquire, 8 / 9 last whole-tile epilogue, 10 / 11 finisher epilogue, 12 stores drained); NULL = off.
 * peneo_gemm_sk_set_max_groups: at most this many workgroups per persistent launch (a multiple of 8; 0 = one per CU). */
void peneo_gemm_sk_set_prof(void* stamps);
void peneo_gemm_sk_set_max_groups(int n);

#ifdef __cplusplus
}
#endif
#endif /* PENEO_HIP_H */
