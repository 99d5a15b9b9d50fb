/*
 * m3p_hip.h — C ABI of libm3p_hip.so: the MI355X (gfx950) kernels behind M3P's
 * pre-training hot path (TransformerModel.jointfwd / predict, Adam, clip).
 *
 * The reference (microsoft/M3P) has no native code: every entry point below replaces
 * a chain of ATen ops launched from Python at the cited reference file:line
 * (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes binding
 * a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers + sizes, no torch / C++ types in any signature;
 *   - every pointer is DEVICE memory owned by the caller (no ownership transfer, the
 *     library never allocates); activations are bf16 unless noted, parameters and
 *     parameter gradients are fp32 ("master" precision), statistics fp32;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued asynchronously;
 *   - return value: 0 = enqueued OK; <0 = M3P_E* (bad shape / alignment, nothing was
 *     launched); >0 = a hipError_t.  Nothing throws across the ABI;
 *   - the library never allocates and keeps no per-call state: re-entrant (the autograd engine calls from its
 *     own thread).  What IS process-global: idempotent one-time initialisation (the CU count, the dynamic-LDS
 *     attribute of each kernel), the two schedule setters m3p_set_persistent_grid / m3p_set_tile_queue and the developer
 *     switch m3p_debug_set_variant (A/B runs of kernel generations; never called by the product).
 */
#ifndef M3P_HIP_H
#define M3P_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M3P_API __attribute__((visibility("default")))

#define M3P_OK 0
#define M3P_EINVAL (-1)
#define M3P_ENOTIMPL (-2)
#define M3P_ENOMEM (-3)   /* an internal scratch allocation failed */

/* library / build identification: returns a static string "m3p_hip <ver> gfx950" */
M3P_API const char* m3p_version(void);

/* ------------------------------------------------------------------------------------
 * Dense contractions on MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate.
 * ---------------------------------------------------------------------------------- */

/* epilogue selectors for m3p_gemm_nt_bf16 */
enum {
  M3P_EPI_NONE = 0,          /* C = alpha*acc                                            */
  M3P_EPI_BIAS = 1,          /* C = alpha*acc + bias[n]; cols n < scale_cols are *= scale */
  M3P_EPI_BIAS_GELU = 2,     /* u = acc + bias -> out2 (bf16); C = gelu_erf(u)           */
  M3P_EPI_BIAS_DROP_RES = 3, /* C = dropout(acc + bias) + aux                            */
  M3P_EPI_RES = 4,           /* C = alpha*acc + aux                                      */
  M3P_EPI_DGELU = 5,         /* C = acc * gelu_erf'(aux); colsum[n] += sum_m C (optional) */
  M3P_EPI_MUL = 6,           /* C = acc * aux;            colsum[n] += sum_m C (optional) */
  /* round 4 - the three below run on the eight-wave 256 x 256 kernel only: M, N multiples of 256, M >= 1024, N >= 512,
   * K % 64 == 0, 16-byte aligned bases; anything else returns M3P_EINVAL */
  M3P_EPI_MULQ = 7,          /* C = acc * decode(aux); colsum[n] += sum_m C (optional).  aux = uint8 [M * N]: the one-byte
                                gelu_erf' codes of this [M, N] in the GEMM's fragment order (written by M3P_EPI_BIAS_GELUQ
                                or m3p_gelu_fwd_gq): the FFN data gradient through GELU, autograd of transformer.py:223-225 */
  M3P_EPI_BIAS_GELUQ = 8,    /* u = acc + bias (fp32, NOT stored); C = gelu_erf(u); out2 = uint8 [M * N]: gelu_erf'(u) as one-byte
                                codes in the fragment order M3P_EPI_MULQ reads (bias required): the FFN's lin1 + activation of
                                transformer.py:223 (gelu :48-56) with what backward needs of u kept in a byte */
  M3P_EPI_BIAS_LSE = 9       /* C = acc + bias (M3P_EPI_BIAS without alpha / column scale) and, for the cross-entropy over the
                                N columns (PredLayer, transformer.py:104-117): out2 = float2 [N / 64][M], the (maximum, sum of
                                exp(x - maximum)) of every row over each 64-column block, columns >= ld_out2 (= V, the valid
                                vocabulary) left out; m3p_ce_lse_from_blocks folds them into the rows' log-sum-exp */
};

typedef struct M3PEpilogue {
  const float* bias;  /* [N] fp32 or NULL                                             */
  const void* aux;    /* bf16 [M, ld_aux]: residual (DROP_RES, RES) / pre-activation (DGELU); uint8 codes (MULQ) */
  void* out2;         /* bf16 [M, ld_out2]: pre-activation output (BIAS_GELU); uint8 codes (BIAS_GELUQ); float2 block
                         statistics (BIAS_LSE, with ld_out2 = the number of valid columns)    */
  float* colsum;      /* fp32 [N] accumulated with atomics, or NULL (DGELU)            */
  int32_t ld_aux;
  int32_t ld_out2;
  int32_t scale_cols; /* BIAS: number of leading output columns multiplied by `scale`   */
  float scale;
  float alpha;        /* accumulator multiplier (NONE, BIAS, RES); 0 is treated as 1    */
  uint32_t seed;      /* dropout stream key (DROP_RES)                                 */
  uint32_t thresh24;  /* round(p * 2^24); 0 = no dropout.  Element i of a stream is dropped iff the low (i even) / high (i odd)
                         16 bits of hash32(i >> 1, seed) are < thresh24 >> 8 (csrc/common.hpp; NumPy twin m3p_amd/rng.py)  */
  float inv_keep;     /* 1 / (1 - p)                                                   */
  const float* descale_a;  /* m3p_gemm_nt_fp8 only: device scalars, the accumulators are multiplied by    */
  const float* descale_b;  /* (*descale_a) * (*descale_b) (NULL = 1) before the epilogue                  */
  /* round 6 - BIAS_GELUQ and MULQ only: the epilogue also leaves an 8-BIT COPY of C for the fp8 product that consumes it
   * (lin2 forward reads gelu(u), the dx1 data gradient reads dU): out8 [M, ld_out8] bytes, row-major,
   * = sat(C_fp32 * (*scale8)) in e4m3 (out8_bf8 = 0) or e5m2 (1), and *amax8 is raised to max |C| (atomic max; the
   * caller zeroes it) - what m3p_quant_fp8 would make of C in a pass of its own.  NULL = no copy.  ld_out8 % 16 == 0,
   * 16-byte aligned base.  BIAS_GELUQ (an activation) takes out8_bf8 = 0 only, MULQ (a gradient) out8_bf8 = 1 only. */
  void* out8;
  const float* scale8;
  float* amax8;
  int32_t ld_out8;
  int32_t out8_bf8;
} M3PEpilogue;

/* C[M,N] (bf16, row pitch ldc) = epilogue( A[M,K] (bf16, pitch lda) x W[N,K]^T (bf16, pitch ldw) ).
 * The nn.Linear shape: replaces F.linear at transformer.py:178-181 (q/k/v_lin as one
 * N=3d GEMM), :208 (out_lin), :223/:225 (FFN lin1/lin2, with :56 gelu and :226 dropout
 * and the residual adds of :951-957 folded into the epilogue), :257 (region
 * projection), :111 (tied vocabulary projection) and, with transposed weight copies,
 * every data-gradient GEMM of their backward.
 * Requires K % 64 == 0, lda/ldw % 8 == 0, ldc % 4 == 0, 16-byte aligned bases. */
M3P_API int m3p_gemm_nt_bf16(const void* A, int lda, const void* W, int ldw, void* C, int ldc,
                             int M, int N, int K, int epilogue, const M3PEpilogue* ep, void* stream);

/* The same contraction with 8-bit operands on the K = 128 MFMA (BASELINE configs[3], "fp8 MFMA GEMMs"):
 * C[M,N] (bf16) = epilogue( (*ep->descale_a) * (*ep->descale_b) * A8[M,K] x W8[N,K]^T ), fp32 accumulate.
 * A8: OCP fp8 e4m3, or bf8 e5m2 when a_is_bf8 (gradients); W8: fp8 e4m3; row pitches lda / ldw in BYTES.
 * Operands come from m3p_quant_fp8 (per-tensor scale, delayed: m3p_amd/fp8.py).  Epilogues NONE, BIAS, BIAS_DROP_RES,
 * RES, DGELU.  Requires M % 256 == 0, N % 256 == 0, K % 128 == 0, lda/ldw % 16 == 0, ldc % 8 == 0, 16-byte bases. */
M3P_API int m3p_gemm_nt_fp8(const void* A8, int lda, int a_is_bf8, const void* W8, int ldw, void* C, int ldc,
                            int M, int N, int K, int epilogue, const M3PEpilogue* ep, void* stream);

/* Per-tensor quantisation for the above: dst[r, c] (8-bit, pitch ld_dst bytes) = sat(src[r, c] * (*scale)) for a bf16
 * matrix [rows, cols] (pitch ld_src elements), format e4m3 (bf8 = 0, saturates at +-448) or e5m2 (bf8 = 1, +-57344),
 * round to nearest even; *amax (fp32, device) is raised to max |src| (atomic max over the launch; the caller zeroes it).
 * scale NULL = 1.  cols % 8 == 0, ld_src % 8 == 0, ld_dst % 8 == 0. */
M3P_API int m3p_quant_fp8(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, const float* scale,
                          float* amax, int bf8, void* stream);

/* The same for many contiguous matrices in one launch (the layers' weights, once per optimizer step): desc = n_desc rows
 * of five int64 {src bf16*, dst uint8*, const float* scale, float* amax or 0, elements / 8}, device memory; e4m3. */
M3P_API int m3p_quant_fp8_batch(const long long* desc, int n_desc, int blocks_per_matrix, void* stream);

/* Stream-K variant with fp32 accumulate output: Cf[M,N] (fp32, pitch ldc) += alpha * A Wᵀ.
 * For few-tile / very-long-K problems — the data gradient of the tied vocabulary
 * projection (autograd of transformer.py:111: M = n_pred, N = d, K = V_pad). */
M3P_API int m3p_gemm_nt_streamk_f32(const void* A, int lda, const void* W, int ldw, float* C, int ldc,
                                    int M, int N, int K, float alpha, void* stream);
/* Same with the second operand given as W[K, N] (row = contraction index): Cf += alpha * A W.
 * This is how the vocabulary data gradient dH = dlogits . E reads the embedding matrix E[V, d]
 * in place (no transposed copy).  K may exceed the number of valid rows of W (V padded to a
 * multiple of 64): rows >= k_valid are not read - the matching columns of A must be zeros. */
M3P_API int m3p_gemm_nn_streamk_f32(const void* A, int lda, const void* W, int ldw, int k_valid, float* C,
                                    int ldc, int M, int N, int K, float alpha, void* stream);
/* The same product for whole-tile shapes on the four-wave kernel (round 4): Cf[M,N] (fp32) += alpha * A[M,K] x W[K,N], A read as
 * an NT activation panel (K contiguous), W through transposing LDS reads, (tile, K-chunk) segments dealt to the CUs, partial
 * tiles folded through the caller's weight-gradient workspace (m3p_gemm_wgrad_workspace_bytes) - no atomics.  ALL K rows of W
 * are read (unlike k_valid above): where A holds zero columns, W must still hold finite numbers.  M % 256 == 0, N % 256 == 0,
 * K % 64 == 0, K >= 4096, lda / ldw % 8 == 0; anything else returns M3P_ENOTIMPL (use the stream-K form). */
M3P_API int m3p_gemm_nn_w4_f32(const void* A, int lda, const void* W, int ldw, float* C, int ldc, int M, int N, int K, float alpha,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Weight gradient: dW[N,K] (fp32, pitch lddw) += alpha * sum_m dY[m,n] * X[m,k]
 * (dY bf16 [M,N] pitch lddy, X bf16 [M,K] pitch ldx).  Accumulates with fp32 atomics
 * (split over M to fill 256 CUs), so dW must hold the running gradient (zero after
 * zero_grad).  Replaces autograd's addmm-backward for every nn.Linear weight above.
 * Requires N % 16 == 0 ... see source; lddy/ldx % 8 == 0.
 * workspace (optional, device memory, 16-byte aligned, >= m3p_gemm_wgrad_workspace_bytes()): scratch for the
 * four-wave kernel's partial tiles (one 256-KB slot per CU, folded into dW by a reduce kernel in slot order, no atomics:
 * bit-reproducible).  It belongs to the caller - the library never allocates - and must not be shared by launches that can
 * run concurrently (different streams).  NULL: partial tiles are added to dW with atomics instead.
 * (Round 5 built the reduction INTO the launch - write-through partials, an arrival counter per tile, every chunk's
 * workgroup summing a 1/C stripe in slot order - correct and slower: DESIGN.md section 4, profiles/r05_wgrad_inkernel_*.) */
M3P_API size_t m3p_gemm_wgrad_workspace_bytes(void);
M3P_API int m3p_gemm_wgrad_bf16(const void* dY, int lddy, const void* X, int ldx, float* dW, int lddw,
                                int M, int N, int K, float alpha, void* workspace, size_t workspace_bytes,
                                void* stream);
/* The same product STORED instead of accumulated: dW = alpha * dY^T X, for a gradient the caller knows to be all zeros (the
 * first product of a step into it).  Only for shapes the four-wave kernel deals out as whole tiles (N, K multiples of 256,
 * M % 64 == 0, at least four 256 x 256 tiles per CU - the tied vocabulary matrix, autograd of transformer.py:111): every tile
 * is then flushed exactly once and a plain store replaces 4-byte atomic read-modify-writes of a 768-MB matrix no cache holds.
 * Anything else returns M3P_ENOTIMPL and the caller accumulates with m3p_gemm_wgrad_bf16. */
M3P_API int m3p_gemm_wgrad_store_bf16(const void* dY, int lddy, const void* X, int ldx, float* dW, int lddw, int M, int N, int K,
                                      float alpha, void* workspace, size_t workspace_bytes, void* stream);
/* Two weight gradients over the same M rows in ONE launch of the four-wave kernel (and one reduction): dWa += dYa^T Xa,
 * dWb += dYb^T Xb.  The attention sub-layer's out_lin (768 x 768: 9 output tiles) and q/k/v (27 tiles) gradients of
 * MultiHeadAttention (transformer.py:178-181, :208) together fill the 256 CUs as evenly as one FFN gradient does - apart they
 * end in two end-of-kernel flushes and two reductions.  Falls back to two m3p_gemm_wgrad_bf16 calls when the shapes are not
 * whole 256 x 256 tiles, M % 64 != 0, the tiles of both together exceed half the CUs, or there is no workspace. */
M3P_API int m3p_gemm_wgrad_pair_bf16(const void* dYa, int lddya, const void* Xa, int ldxa, float* dWa, int lddwa, int Na, int Ka,
                                     const void* dYb, int lddyb, const void* Xb, int ldxb, float* dWb, int lddwb, int Nb, int Kb,
                                     int M, float alpha, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * LayerNorm (eps = 1e-12 in the reference: transformer.py:244,660,694,709)
 * ---------------------------------------------------------------------------------- */

/* y[r,:] = LN(x[r,:]) * gamma + beta, then * rowmask[r] if rowmask != NULL
 * (the `tensor *= mask` of transformer.py:958).  x,y bf16 [rows,d]; gamma,beta fp32;
 * mean,rstd fp32 [rows] saved for backward.  d % 4 == 0, d <= 2048. */
M3P_API int m3p_layernorm_fwd(const void* x, const float* gamma, const float* beta, const uint8_t* rowmask,
                              void* y, float* mean, float* rstd, int rows, int d, float eps, void* stream);

/* Backward of the above.  dy = dy_a (+ dy_b if not NULL), multiplied by rowmask[r] if given.
 *   dx        : bf16 [rows,d] gradient wrt the LayerNorm input (always written)
 *   dx_drop   : if not NULL, also dx * keep/(1-p) with the dropout stream (seed,thresh24)
 *               indexed by r*d+c — i.e. the gradient pushed through the dropout that
 *               produced the residual branch (transformer.py:951 / :226)
 *   dgamma,dbeta : fp32 [d], accumulated with atomics
 *   dbias_drop: if not NULL fp32 [d] += column sums of dx_drop (the bias gradient of the
 *               Linear feeding that dropout: out_lin.bias / lin2.bias). */
M3P_API int m3p_layernorm_bwd(const void* dy_a, const void* dy_b, const void* x, const float* gamma,
                              const float* mean, const float* rstd, const uint8_t* rowmask,
                              void* dx, void* dx_drop, float* dgamma, float* dbeta, float* dbias_drop,
                              int rows, int d, uint32_t seed, uint32_t thresh24, float inv_keep, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused multi-head self-attention (transformer.py:149-210, self-attention branch)
 * ---------------------------------------------------------------------------------- */

/* qkv bf16 [B*S, 3*H*dh] token-major (q | k | v column blocks, q already scaled by
 * 1/sqrt(dh) — M3P_EPI_BIAS scale_cols); keylen int32 [B]: keys >= keylen[b] are masked
 * to -inf (get_masks, transformer.py:59-78, non-causal).  Softmax in fp32, dropout on the
 * probabilities with stream (seed, thresh24) indexed ((b*H+h)*S+q)*S+key, context = P v.
 * ctx bf16 [B*S, H*dh] token-major; lse fp32 [B,H,S] = log-sum-exp saved for backward.
 * dh in {32, 64}, S <= 512. */
M3P_API int m3p_attn_fwd(const void* qkv, const int32_t* keylen, void* ctx, float* lse, uint64_t* keepmask,
                         int B, int S, int H, int dh, uint32_t seed, uint32_t thresh24, float inv_keep,
                         void* stream);
/* keepmask (nullable): if given and thresh24 != 0, the forward also leaves the dropout keep
 * bits, [B*H][nt][nt][4] 64-bit words with nt = ceil(S/16); word [qb][t][r] bit l is the
 * keep decision of (query 16qb + (l & 15), key 16t + 4(l >> 4) + r).  m3p_attn_bwd tests these bits
 * (it needs them whenever dropout is on). */

/* Backward: given dctx (bf16 [B*S, H*dh]) writes dqkv (bf16 [B*S, 3*H*dh]; the q block is
 * multiplied by qscale = 1/sqrt(dh) so it is the gradient of the *unscaled* projection)
 * and, if dbias_qkv != NULL, accumulates (atomics) the column sums of the q and v blocks of
 * dqkv into the fp32 [3*H*dh] bias gradient of the fused q/k/v projection.  The k block of
 * the bias gradient is left untouched: it is identically 0 (softmax is invariant to a
 * per-query shift of the scores; the reference's value there is fp32 noise).  Scores are recomputed from
 * qkv + lse (nothing S x S is ever stored).  S <= 384.  With dropout on (thresh24 != 0) keepmask - the words m3p_attn_fwd
 * left - is REQUIRED (M3P_EINVAL without; round 6: the instantiations that re-drew the decisions from (seed, thresh24) served
 * tests only); seed is unused. */
M3P_API int m3p_attn_bwd(const void* qkv, const int32_t* keylen, const void* ctx, const void* dctx,
                         const float* lse, const uint64_t* keepmask, void* dqkv, float* dbias_qkv, int B, int S,
                         int H, int dh, float qscale, uint32_t seed, uint32_t thresh24, float inv_keep,
                         void* stream);

/* Decoder inference (transformer.py:149-210 with the key / value cache of :187-195): every query row
 * attends a prefix of a cached key / value sequence.  q bf16 [B*Tq, ld_q] (scaled and biased by the projection's
 * epilogue); key j of sequence b, head h at kv + b*kv_bstride + j*ld_kv + h*dh, its value H*dh elements further
 * (bf16; element strides, multiples of 8); klen int32 [B] or NULL (= Lk keys everywhere); causal != 0: query t sees keys
 * 0 .. pos0 + t only (pos0 = tokens cached before this call).  fp32 softmax, no dropout (inference), ctx bf16
 * [B*Tq, H*dh].  dh in {32, 64}, Lk <= 1024. */
M3P_API int m3p_attn_query_fwd(const void* q, int ld_q, const void* kv, long long kv_bstride, int ld_kv,
                               const int32_t* klen, void* ctx, int B, int Tq, int H, int dh, int Lk, int causal,
                               int pos0, void* stream);

/* The same attention for the TRAINING of the causal / cross-attention sub-layers (teacher forcing: Tq = the whole target
 * sequence, pos0 = 0): dropout on the probabilities (stream (seed, thresh24) indexed ((b*H + h)*Tq + t)*Lk + key) and the
 * log-sum-exp lse fp32 [B, H, Tq] kept for backward.  B*H*Tq*Lk < 2^32. */
M3P_API int m3p_attn_rows_fwd(const void* q, int ld_q, const void* kv, long long kv_bstride, int ld_kv, const int32_t* klen,
                              void* ctx, float* lse, int B, int Tq, int H, int dh, int Lk, int causal, int pos0,
                              uint32_t seed, uint32_t thresh24, float inv_keep, void* stream);
/* Backward: dq bf16 [B*Tq, ld_dq] = gradient of the UNSCALED query projection (the scaled gradient x qscale, like
 * m3p_attn_bwd); the key / value gradients are ADDED (fp32 atomics) into dkv [B, Lk, 2*H*dh] (keys | values per position),
 * which the caller zeroes. */
M3P_API int m3p_attn_rows_bwd(const void* q, int ld_q, const void* kv, long long kv_bstride, int ld_kv, const int32_t* klen,
                              const void* dctx, const float* lse, void* dq, int ld_dq, float* dkv, int B, int Tq, int H,
                              int dh, int Lk, int causal, int pos0, float qscale, uint32_t seed, uint32_t thresh24,
                              float inv_keep, void* stream);

/* ------------------------------------------------------------------------------------
 * Input assembly of jointfwd (transformer.py:901-943) and its backward
 * ---------------------------------------------------------------------------------- */

/* fp32 -> bf16 (the region features / any fp32 input that feeds a GEMM). n % 4 == 0. */
M3P_API int m3p_cast_f32_bf16(const float* src, void* dst, long long n, void* stream);

/* Builds the encoder input h[B*S, d] (bf16, batch-major rows b*S+s, S = R+T):
 *   image rows s<R : LN_img(img_proj[s*B+b] + W_loc loc[s,b] + b_loc) -> dropout(seed_img)
 *                    (BertImageEmbeddings.forward, transformer.py:257-268; img_proj is the
 *                    W_img x + b GEMM output in the caller's sequence-major row order)
 *   token rows     : emb[tok[t,b]]                                       (:913)
 *   then + pos[s], * (s < totlen[b]), LN_emb, dropout(seed_emb)          (:936-943)
 * Saves z (LN_emb input) / e (LN_img input) and the LayerNorm statistics for backward.
 * tok int64 (T,B); emb bf16 [V,d]; pos,w_loc,b_loc,gammas,betas fp32; loc fp32 (R,B,5).
 * img_rows (nullable): bf16 [B*R, d], row b*R + r - the image rows to use INSTEAD of computing them
 * (jointfwd with refine_image=True, transformer.py:905-906: m3p_embed_image_rows_fwd -> AoA refiner ->
 * here); e / mean_img / rstd_img are then left untouched. */
M3P_API int m3p_embed_assemble_fwd(const int64_t* tok, const void* emb_bf16, const float* pos,
                                   const void* img_proj, const float* loc, const float* w_loc,
                                   const float* b_loc, const float* g_img, const float* be_img,
                                   const float* g_emb, const float* be_emb, const int32_t* totlen,
                                   void* h, void* z, float* mean_emb, float* rstd_emb, void* e,
                                   float* mean_img, float* rstd_img, int B, int T, int R, int d,
                                   uint32_t seed_img, uint32_t seed_emb, uint32_t thresh24, float inv_keep,
                                   const void* img_rows, void* stream);

/* The image rows alone: img_rows[b*R + r, :] = dropout(LN_img(img_proj[r*B+b] + W_loc loc[r,b] + b_loc))
 * (BertImageEmbeddings.forward, transformer.py:247-269) as their own bf16 [B*R, d] tensor, batch-major - the
 * input of the AoA refiner (refine_image=True).  Saves e / mean_img / rstd_img like m3p_embed_assemble_fwd. */
M3P_API int m3p_embed_image_rows_fwd(const void* img_proj, const float* loc, const float* w_loc, const float* b_loc,
                                     const float* g_img, const float* be_img, void* e, float* mean_img,
                                     float* rstd_img, void* img_rows, int B, int R, int d, uint32_t seed_img,
                                     uint32_t thresh24, float inv_keep, void* stream);

/* Backward of the above given dh[B*S,d].  Accumulates (fp32 atomics) into the parameter
 * gradients d_g_emb,d_be_emb [d], d_pos [S,d] (first S rows of position_embeddings),
 * d_emb [V,d] (scatter-add over token ids, rows of pad_index skipped like
 * nn.Embedding(padding_idx)), d_g_img,d_be_img,d_b_img,d_b_loc [d], d_w_loc [d,5]; writes
 * de [R*B,d] (bf16, gradient of img_proj, rows s*B+b) for the W_img weight-gradient GEMM.
 * dz_scratch: bf16 [B*S,d] workspace.
 * d_tok_rows (optional, bf16 [T*B,d]): when set, the token rows' gradients are WRITTEN there (row t*B+b, zero rows
 * for pad / masked tokens) instead of being scatter-added into d_emb - data parallelism exchanges these rows between
 * ranks and applies them with m3p_scatter_add_token_rows (the 768-MB dense matrix is then reduced early, see
 * m3p_amd/distributed.py).
 * phase: 0 = everything.  With the AoA refiner between the image rows and the assembly the two halves run
 * separately: 1 = LN_emb / position / token part only - leaves the gradient of the image rows in
 * dz_scratch[b*S + r]; the caller takes it through the refiner's backward, writes the result back to the same
 * rows, then 2 = image part only (dropout + LN_img + location embedding -> de and their parameter gradients). */
M3P_API int m3p_embed_assemble_bwd(const void* dh, const void* z, const float* mean_emb, const float* rstd_emb,
                                   const float* g_emb, const void* e, const float* mean_img,
                                   const float* rstd_img, const float* g_img, const int64_t* tok,
                                   const int32_t* totlen, const float* loc, void* dz_scratch, void* de,
                                   float* d_g_emb, float* d_be_emb, float* d_pos, float* d_emb,
                                   void* d_tok_rows, float* d_g_img, float* d_be_img, float* d_b_img, float* d_b_loc,
                                   float* d_w_loc, int B, int T, int R, int d, int pad_index,
                                   uint32_t seed_img, uint32_t seed_emb, uint32_t thresh24, float inv_keep,
                                   int phase, void* stream);

/* ------------------------------------------------------------------------------------
 * Heads (TransformerModel.predict, transformer.py:1183-1214; PredLayer :104-117)
 * ---------------------------------------------------------------------------------- */

/* dst[i,:] = src[idx[i],:] — the boolean-mask gather of :1208 with precomputed row indices */
M3P_API int m3p_gather_rows(const void* src, const int32_t* idx, void* dst, int n, int d, void* stream);
/* dst[idx[i],:] += src[i,:] (idx unique) — its backward */
M3P_API int m3p_scatter_add_rows(const void* src, const int32_t* idx, void* dst, int n, int d, void* stream);
/* dst[ids[i],:] (fp32 [V,d]) += rows[i,:] (bf16 [n,d], row pitch ld_rows elements); ids may repeat (atomics), rows with
 * ids[i] == pad_index are skipped (nn.Embedding(padding_idx), transformer.py:21-26).  The embedding-lookup gradient of token
 * rows gathered from the data-parallel ranks (m3p_embed_assemble_bwd with d_tok_rows); the pitch lets the exchange carry each
 * row's id in extra columns behind it - one collective instead of two. */
M3P_API int m3p_scatter_add_token_rows(const void* rows, int ld_rows, const int64_t* ids, float* dst, int n, int d,
                                       int pad_index, void* stream);

/* ----------------------------------------------------------------------------------
 * ITM head: transformer.py:546-558 (BertPooler: tanh(dense(hidden[:, 0]))) followed by
 * :1194-1197 (seq_relationship Linear(d, 1)); position 0 of a joint sequence is the first
 * image region.  The d x d products run on the GEMMs above:
 *   forward   pre = m3p_gemm_nt_bf16(h0 [B,d], W1 [d,d], M3P_EPI_BIAS b1)            (bf16 [B,d])
 *             m3p_itm_score_fwd: pooled = tanh(pre) (fp32 [B,d], saved), scores = w2 . pooled + b2
 *   backward  m3p_itm_score_bwd(dscores [B]): dpre = dscore w2 (1 - pooled^2) as bf16 [B,d] and
 *             transposed [d, ldt >= B]; ACCUMULATES db1, dw2 [d], db2 [1] (fp32)
 *             dW1 += m3p_gemm_wgrad_bf16(dY = dpre16, X = h0);  dh0 = m3p_gemm_wgrad_bf16(dY = dpreT16, X = W1)
 * ---------------------------------------------------------------------------------- */
M3P_API int m3p_itm_score_fwd(const void* pre, const float* w2, const float* b2, float* pooled, float* scores, int B,
                              int d, void* stream);
M3P_API int m3p_itm_score_bwd(const float* dscores, const float* pooled, const float* w2, void* dpre16, void* dpreT16,
                              int ldt, float* db1, float* dw2, float* db2, int B, int d, void* stream);

/* F.cross_entropy over bf16 logits [n_rows, ld] (V valid columns), per-row loss to
 * row_loss, loss_sum += loss_scale * sum(row losses) if loss_sum != NULL, and IN PLACE
 * logits <- (softmax - onehot(target)) * grad_scale, padding columns [V, ld) zeroed.
 * ld % 8 == 0, logits 16-byte aligned. */
M3P_API int m3p_ce_fwd_bwd(void* logits, int ld, int n_rows, int V, const int64_t* target, float* row_loss,
                           float* loss_sum, float loss_scale, float grad_scale, void* stream);

/* The same cross-entropy with the column sums of the gradient - the gradient of the output bias (PredLayer.proj.bias,
 * transformer.py:111) - produced by the pass that writes the gradient, instead of a second pass over it: colsum[c]
 * (fp32 [ld], OVERWRITTEN) = sum over rows of the rounded bf16 gradient; row_lse [n_rows] receives the log-sum-exp.
 * workspace: m3p_ce_colsum_workspace_bytes(ld, n_rows) bytes owned by the caller (16-byte aligned). */
/* Log-sum-exp and loss of every row from the 64-column block statistics the vocabulary projection left behind
 * (M3P_EPI_BIAS_LSE): row_lse[r] = log sum_c exp(logit[r, c]), row_loss[r] = row_lse[r] - logits[r, target[r]].
 * stats float2 [n_blocks][n_rows]; scratch: float2 [32][n_rows] (caller-owned).  Replaces the first pass of m3p_ce_fwd_bwd_colsum;
 * follow with m3p_ce_bwd_colsum. */
M3P_API int m3p_ce_lse_from_blocks(const void* stats, int n_blocks, int n_rows, const void* logits, int ld, const int64_t* target,
                                   float* row_loss, float* row_lse, void* scratch, void* stream);
/* The second half of m3p_ce_fwd_bwd_colsum alone: logits <- grad_scale * (softmax - onehot) in place given the rows' log-sum-exp,
 * colsum [ld] <- column sums of the rounded gradient.  Same workspace. */
M3P_API int m3p_ce_bwd_colsum(void* logits, int ld, int n_rows, int V, const int64_t* target, const float* row_lse,
                              float grad_scale, float* colsum, void* workspace, size_t workspace_bytes, void* stream);
M3P_API size_t m3p_ce_colsum_workspace_bytes(int ld, int n_rows);
M3P_API int m3p_ce_fwd_bwd_colsum(void* logits, int ld, int n_rows, int V, const int64_t* target, float* row_loss,
                                  float* row_lse, float grad_scale, float* colsum, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* out[c] += scale * sum_r x[r,c] for c < ncols (x bf16 [n, ld]); scale read from the
 * device scalar *scale_ptr (NULL = 1): the vocabulary-bias gradient colsum(dlogits). */
M3P_API int m3p_colsum_bf16(const void* x, int ld, int n, int ncols, float* out, const float* scale_ptr,
                            void* stream);

/* ------------------------------------------------------------------------------------
 * Optimizer path (Trainer.optimize xtrainer.py:205-243; Adam.step optim.py:45-86)
 * ---------------------------------------------------------------------------------- */

/* *out += sum(g^2) in double (clip_grad_norm_, xtrainer.py:225). n % 4 == 0. */
M3P_API int m3p_sumsq_f32(const float* g, long long n, double* out, void* stream);

/* The same over n_ranges pieces base[starts[r] .. starts[r] + counts[r]) of one buffer (starts / counts: HOST arrays, element
 * units, counts % 4 == 0, pieces 16-byte aligned) in one launch: the sharded data-parallel step (m3p_amd/distributed.py, zero1)
 * owns a piece of every gradient bucket.  clip_grad_norm_, xtrainer.py:225. */
M3P_API int m3p_sumsq_ranges_f32(const float* base, const long long* starts, const long long* counts, int n_ranges, double* out,
                                 void* stream);

/* One fused pass over a flat fp32 range:  g' = g * grad_scale * clip_coef,
 *   clip_coef = min(1, max_norm / (sqrt(*gnorm_sq) * grad_scale + 1e-6))   (if gnorm_sq && max_norm > 0)
 *   m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p -= wd*lr*p;  p -= step_size * m / (sqrt(v) + eps)
 * step_size = lr * sqrt(1-b2^t)/(1-b1^t) is computed by the caller (optim.py:78-80).
 * Also refreshes the bf16 working copy w16 (if not NULL) and zeroes g (if zero_grad). */
M3P_API int m3p_adam_step(float* p, float* g, float* m, float* v, void* w16, long long n, float lr, float beta1,
                          float beta2, float eps, float weight_decay, float step_size, const double* gnorm_sq,
                          float max_norm, float grad_scale, int zero_grad, void* stream);
/* The same update over n_ranges pieces [starts[r], starts[r] + counts[r]) (elements, multiples of 4) of the SAME flat arenas in
 * one launch; step_sizes / zero_grad per piece (pieces of parameters with different update counts have different bias
 * corrections; a piece whose gradient the next step overwrites need not be zeroed).  starts / counts / step_sizes / zero_grad
 * are HOST arrays, read before the call returns.  p, g, m, v, w16 are the arenas' bases. */
M3P_API int m3p_adam_step_ranges(float* p, float* g, float* m, float* v, void* w16, const long long* starts,
                                 const long long* counts, const float* step_sizes, const int* zero_grad, int n_ranges,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, const double* gnorm_sq,
                                 float max_norm, float grad_scale, void* stream);

/* h = gelu_erf(u) elementwise (transformer.py:56 applied to the lin1 output :223-224), bf16,
 * n % 8 == 0.  Used instead of M3P_EPI_BIAS_GELU when the GEMM is persistent.
 * dh (nullable, may alias u): also writes gelu_erf'(u) in bf16, which the backward FFN dgrad
 * then applies with M3P_EPI_MUL instead of recomputing the derivative in the GEMM epilogue. */
M3P_API int m3p_gelu_fwd(const void* u, void* h, void* dh, long long n, void* stream);
/* The same activation for the persistent-GEMM FFN (transformer.py:223-225, gelu :48-56) leaving what backward needs in ONE
 * byte per element: h (bf16 [M, N], contiguous) = gelu_erf(u), gq (uint8, M * N bytes) = gelu_erf'(u) quantised to the
 * grid (code - 27) / 200 (0 and 1 exact, |error| <= 2.5e-3), stored in the fragment order of the eight-wave NT GEMM's 256 x 256 tiles
 * so that the FFN data gradient m3p_gemm_nt_bf16(..., M3P_EPI_MULQ, aux = gq) multiplies by it without a layout change.
 * u itself is not needed after this call.  M % 256 == 0, N % 256 == 0; u, h contiguous and 16-byte aligned. */
M3P_API int m3p_gelu_fwd_gq(const void* u, void* h, void* gq, int M, int N, void* stream);
/* The same pass also writing h8 = e4m3(scale * h) (uint8 [n]) and raising *amax to max |h| (nullable): the operand of the
 * fp8 lin2 product without a second pass over h (m3p_quant_fp8 of h gives the same bytes). */
M3P_API int m3p_gelu_fwd_q8(const void* u, void* h, void* h8, long long n, const float* scale, float* amax, void* stream);

/* du = dy * gelu_erf'(u) elementwise (backward of the GELU inside BertPredictionHeadTransform,
 * transformer.py:595-606, for the masked-region classification head), bf16, n % 8 == 0. */
M3P_API int m3p_gelu_bwd(const void* dy, const void* u, void* du, long long n, void* stream);

/* Masked-region feature regression loss (F.mse_loss of xtrainer.py:2346 on the gathered masked rows):
 * row_sq[i] = sum_j (pred[i][j] - tgt[i][j])^2 (the caller divides the total by rows*cols) and
 * dpred[i][j] = 2 (pred - tgt) * grad_scale.  pred/dpred bf16 (pitch ld_pred), tgt fp32 (pitch ld_tgt). */
M3P_API int m3p_mse_fwd_bwd(const void* pred, int ld_pred, const float* tgt, int ld_tgt, void* dpred, float* row_sq,
                            int rows, int cols, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------
 * AoA image refiner (AoA_Refiner_Core, transformer.py:287-422): the elementwise passes its GEMMs,
 * LayerNorms, attention and GELU (all the encoder's kernels) leave over.
 * ---------------------------------------------------------------------------------- */

/* y[r][c] = (res ? res[r][c] : 0) + dropout(x[r][c]) on a [rows, cols] bf16 view (pitches ldx / ldres / ldy,
 * cols % 4 == 0).  The keep decision of element (r, c) is m3p_keep(r * rng_ld + rng_col0 + c, seed, thresh24)
 * (thresh24 = 0: no dropout), so the two halves of torch.cat([x, q], -1) can be dropped by two calls into one
 * [rows, 2d] buffer (rng_ld = 2d, rng_col0 = 0 / d) and backward regenerates the same mask.  y may alias x.
 * SublayerConnection.forward (:392-394), MultiHeadedDotAttention's dropout_aoa (:366), TransformerFFN (:226). */
M3P_API int m3p_dropout_rows(const void* x, int ldx, const void* res, int ldres, void* y, int ldy, int rows, int cols,
                             uint32_t rng_ld, uint32_t rng_col0, uint32_t seed, uint32_t thresh24, float inv_keep,
                             void* stream);

/* nn.GLU of the AoA layer (:317): ab bf16 [rows, 2d] (pitch ld_ab) -> y[rows, d] = ab[:, :d] * sigmoid(ab[:, d:]) */
M3P_API int m3p_glu_fwd(const void* ab, int ld_ab, void* y, int rows, int d, void* stream);
/* dab[:, :d] = dy * sigmoid(b);  dab[:, d:] = dy * a * sigmoid(b) * (1 - sigmoid(b))   (dab pitch = ld_ab) */
M3P_API int m3p_glu_bwd(const void* ab, int ld_ab, const void* dy, void* dab, int rows, int d, void* stream);

/* Batched form of m3p_transpose_bf16: desc = n_desc x {src ptr, dst ptr, rows, cols, ld_src,
 * ld_dst} as int64 in device memory; max_tiles >= max over matrices of ceil(rows/64)*ceil(cols/64). */
M3P_API int m3p_transpose_batch_bf16(const long long* desc, int n_desc, int max_tiles, void* stream);

/* dst[c, r] = src[r, c] (bf16): transposed weight copies for the data-gradient GEMMs */
M3P_API int m3p_transpose_bf16(const void* src, void* dst, int rows, int cols, int ld_src, int ld_dst,
                               void* stream);

/* Size of the persistent GEMM grids (process-wide; 0 = one workgroup per CU, the default; a multiple of 8: tiles are dealt
 * out per XCD).  Data parallelism sets num_CUs - r so that RCCL's r channels find free CUs while a GEMM runs - a persistent
 * workgroup fills its CU's LDS and registers, nothing co-resides with it (m3p_amd/distributed.py). */
M3P_API int m3p_set_persistent_grid(int workgroups);
/* Dynamic tile queues for the persistent NT GEMM (process-wide; counters = NULL switches back to the static schedule).
 * counters: n_slots x 8 int32 in device memory, zeroed by the caller once; every eligible m3p_gemm_nt_bf16 launch (eight-wave
 * kernel, more output tiles than workgroups, K >= 512) takes the next slot - one counter per XCD - and its workgroups pop
 * their output tiles from per-XCD queues instead of a fixed round-robin share, so a CU that runs slower (a collective's kernel
 * resident beside the GEMM under data parallelism) takes fewer tiles instead of stretching the launch.  The ring binds to the
 * stream of the first queued launch after this call and is cleared by a memset on that stream when it wraps; launches on any
 * other stream take the static schedule (never a slot another stream's kernel may still be popping from).  Slot hand-out is
 * serialised inside the library; call this setter with no GEMM of the old ring in flight.  Results are the static schedule's. */
M3P_API int m3p_set_tile_queue(int32_t* counters, int n_slots);

/* Which kernel a product of this shape WOULD run on, given the process-wide switches above (no launch, no device access): the
 * dispatch tables in csrc/gemm.hip (nt_plan / wgrad_plan) are the single place that decides, the launchers switch on the same
 * value.  Tests use it to assert that a model-level parity case really exercised the kernel family the benchmark runs on
 * (tests/test_model_parity.py[tiles]).  Returns an M3P_KERN_* id, or M3P_EINVAL for a shape the entry point would refuse.
 * m3p_gemm_wgrad_plan assumes 16-byte aligned operands with pitches that are multiples of 8. */
enum {
  M3P_KERN_NT_SKINNY = 1,        /* M <= 128: fragments straight from global memory                       */
  M3P_KERN_NT_W8 = 2,            /* eight waves, 256 x 256 tile, 128 x 64 per wave                         */
  M3P_KERN_NT_W8_QUEUE = 3,      /* the same with per-XCD dynamic tile queues (m3p_set_tile_queue)         */
  M3P_KERN_NT_W4 = 4,            /* four waves, 256 x 256 tile, 128 x 128 per wave (K >= 2048)             */
  M3P_KERN_NT_RING = 5,          /* eight waves, 256 x 128 tile, three-stage ring: ragged shapes           */
  M3P_KERN_NT_128 = 6,           /* 128 x 128, two stages: M < 1024                                        */
  M3P_KERN_WGRAD_W4_CHUNKS = 10, /* four waves, one (tile, M-chunk) segment per workgroup + ordered reduce kernel */
  M3P_KERN_WGRAD_W4_TILES = 11,  /* four waves, whole tiles round-robin (the vocabulary matrix)            */
  M3P_KERN_WGRAD_RING = 12,      /* stream-K 256 x 128 with fp32 atomics: ragged N / K                     */
  M3P_KERN_WGRAD_128 = 13        /* 128 x 128 split-M: small M                                             */
};
M3P_API int m3p_gemm_nt_plan(int M, int N, int K, int epilogue);
M3P_API int m3p_gemm_wgrad_plan(int M, int N, int K);

/* Developer switch (process-wide, NEVER called by the product; tools/ab_*.py and M3P_VARIANT in m3p_amd/lib.py use it for A/B
 * runs of kernel generations): low byte 0 = force the 128 x 128 kernels, 1 = the tables above (default), 2 = four-wave NT
 * wherever it applies, 3 = ring kernels, 6 = eight-wave NT, 7 = round 1's choice, 9 = default without the skinny kernel; the
 * higher bits are ablation flags of the timeline builds.  Every launch reads it: set it only with no GEMM call in flight. */
M3P_API void m3p_debug_set_variant(int v);
/* The same kind of switch for the attention kernels: bit 0 = the two-phase backward (scores recomputed in both phases,
 * rounds 1-4) instead of the one-pass form for the M3P sequence (tools/ab_attn.py). */
M3P_API void m3p_debug_attn_variant(int v);

/* ------------------------------------------------------------------------------------
 * Host-glue kernels (csrc/glue.hip): index / mask / loss arithmetic the reference does with chains of elementwise
 * tensor ops around the hot path; one launch each, no host reads.
 * ---------------------------------------------------------------------------------- */
/* get_masks (transformer.py:59-78) for the prefix mask of jointfwd (:917-919): totlen[b] = lengths[b] (+ lengths_b[b]
 * when given), rowmask[b*S + s] = s < totlen[b] */
M3P_API int m3p_seq_masks(const int64_t* lengths, const int64_t* lengths_b, int B, int S, int32_t* totlen,
                          uint8_t* rowmask, void* stream);
/* The boolean gather of :1208 as row numbers: the k-th True entry (flat order i = t * inner + b) of mask [n_mask] ->
 * rows[k] = (soff + t * s0 + b * s1) / d, the row of that position in the [*, d] buffer under a strided (T, B, d) view
 * (element strides s0, s1, storage offset soff).  n_rows = the caller's count of True entries (host knowledge); a
 * shorter mask leaves the tail at row 0.  n_mask <= 2^20. */
M3P_API int m3p_mask_to_rows(const uint8_t* mask, int n_mask, int inner, long long s0, long long s1, long long soff,
                             int d, int32_t* rows, int n_rows, void* stream);
/* out[(i0 * n1 + i1), :] (bf16, contiguous) = in[i0 * s0 + i1 * s1 + (0 .. cols)] (fp32): the region features through
 * the transposed view the model is handed (transformer.py:897-898) without materialising it.  cols, s0, s1 % 4 == 0 */
M3P_API int m3p_cast_rows_f32_bf16(const float* in, long long s0, long long s1, int n0, int n1, int cols, void* out,
                                   void* stream);
/* out (bf16) = g[0] * in (bf16, or fp32 when in_is_f32), g a DEVICE scalar (an upstream gradient); n % 4 == 0 */
M3P_API int m3p_scale_bf16_dev(const void* in, int in_is_f32, const float* g, void* out, long long n, void* stream);
/* dst += g[0] * src (fp32), g a device scalar */
M3P_API int m3p_axpy_dev_f32(float* dst, const float* src, const float* g, long long n, void* stream);
/* ITM loss of xtrainer.py:2357-2372 on the device: scores fp32 [n_groups * sample_n], pos int64 [n_groups];
 * loss[0] = w_ce * CE(scores.view(-1, sample_n), pos) + w_bce * BCEWithLogits(scores, one_hot(pos)) (both means);
 * dscores = d loss / d scores */
M3P_API int m3p_itm_loss_fwd_bwd(const float* scores, const int64_t* pos, int n_groups, int sample_n, float w_ce,
                                 float w_bce, float* loss, float* dscores, void* stream);

/* ------------------------------------------------------------------------------------
 * Hardware-semantics probes (used by tests/test_hw_probes.py only): each fills `out`
 * with what the instruction delivered so the test can compare with the documented map.
 * ---------------------------------------------------------------------------------- */
M3P_API int m3p_probe_mfma_16x16x32(const void* a_rowmajor_16x32, const void* b_colmajor_32x16,
                                    float* d_16x16, int* d_rowcol, void* stream);
/* v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales: a, w row-major 8-bit [16,128] (a: fp8 e4m3, or bf8 e5m2
 * when a_is_bf8; w: fp8 e4m3), d fp32 [16,16] = a w^T written through the 16x16 result map */
M3P_API int m3p_probe_mfma_fp8_16x16x128(const void* a_16x128, const void* w_16x128, float* d_16x16, int a_is_bf8,
                                         void* stream);
M3P_API int m3p_probe_tr16(const void* tile_bf16_64x16, void* out_64x4, void* stream);
/* v_permlane16_swap_b32 x, y with (x, y) = (lane, 100 + lane): out[lane] = x, out[64 + lane] = y afterwards - the odd 16-lane rows
 * of x trade places with the even rows of y (what the persistent attention backward's widened row stores rest on) */
M3P_API int m3p_probe_permlane16_swap(void* out_2x64_u32, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* M3P_HIP_H */
