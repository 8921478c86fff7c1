/*
 * cirrank.h - C ABI of libcirrank.so, the MI355X (gfx950) kernel library behind the stage-II
 * candidate re-ranking forward path of Cuberick-Orion/Candidate-Reranking-CIR.
 *
 * The reference has no native code and no FFI: its "operators" are PyTorch-eager op sequences
 * inside three Python files.  Each entry point below replaces one such sequence; the citation
 * names the reference lines whose arithmetic it computes (paths relative to the reference's src/).
 * The Python host (candidate_reranking_cir_amd/) binds these with ctypes; INTEGRATION.md shows
 * the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (tensor.data_ptr()) - except cir_wgrad_grouped's descriptor array -; the caller owns all buffers; the
 *     library allocates nothing, never synchronises the device, and keeps no state besides the
 *     kernel-selection overrides of cir_set_tuning (default: none) and the cached CU count;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     work is enqueued on it and nowhere else;
 *   - leading dimensions / strides are in ELEMENTS of the tensor they describe - except the "split8" operand rows of ABI v14
 *     (cir_split8, cir_gemm_split8, cir_layernorm_split8, cir_attention_split8), which are BYTE rows of three segments and take their
 *     leading dimensions / strides in BYTES where the argument name says so;
 *   - dtype codes: CIR_BF16 / CIR_F16 for 16-bit activations and weights (fp32 accumulate
 *     everywhere), CIR_F32 for fp32 tensors - and, since ABI v11, as the OPERAND type of cir_gemm_bias_act, cir_attention
 *     and cir_patchify: the "exact" precision mode, every tensor fp32 like the reference's (validate_stage2.py:140-141:
 *     model.float()), products on the f32-input MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2_f32: IEEE fmaf chains, 157.3 TFLOP/s
 *     peak = 1/16 of the 16-bit rate); the RESIDUAL STREAM (every x + sublayer(x) and the
 *     LayerNorm outputs that feed one) is either fp32 or fp16 ("stream dtype": CIR_F32 / CIR_F16),
 *     independently of the operand type - sums are formed in fp32 and rounded on the store (a GEMM with an fp16
 *     residual rounds its result to the stream type before the residual joins it: two roundings there);
 *   - return 0 on success, a negative CIR_E* code for an argument error detected before launch,
 *     or a positive hipError_t from the launch.  Nothing throws or aborts across the ABI.
 *   - functions are re-entrant and may be called from any host thread.
 */
#ifndef CIRRANK_H
#define CIRRANK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CIR_ABI_VERSION 15

enum { CIR_BF16 = 0, CIR_F16 = 1, CIR_F32 = 2 };
enum { CIR_ACT_NONE = 0, CIR_ACT_GELU = 1, CIR_ACT_RELU = 2 };
enum {
    CIR_OK = 0,
    CIR_EINVAL = -1,   /* null pointer / non-positive extent */
    CIR_ESHAPE = -2,   /* extent not supported by the kernels (see each function) */
    CIR_EALIGN = -3,   /* pointer or leading dimension not 16-byte aligned */
    CIR_EDTYPE = -4    /* dtype code not supported here */
};

int cir_version(void);
const char* cir_strerror(int code);

/*
 * Kernel-selection overrides for tests and A/B measurements (the library reads NO environment
 * variables).  Every knob defaults to 0 = automatic choice; a process that never calls this
 * function always gets the automatic choice.  Returns CIR_EINVAL for an unknown knob or a
 * value outside the knob's range.  Not part of any reference interface.
 *   CIR_TUNE_GEMM_TILE        0 auto | 128 | 256 : force the 128x128 or the persistent 256x256 GEMM
 *                             kernel where the shape allows it (cir_gemm_bias_act)
 *   CIR_TUNE_GEMM_GROUP_W     0 auto | 1..64    : n-panels per raster group of the 256x256 kernel
 *   CIR_TUNE_ATTN_SHARED_MAX  0 auto (608) | -1 never | 32..608 : largest padded key count for which
 *                             cir_attention stages a head's K/V once per workgroup in LDS; -2: as 0, and cir_attention_split8
 *                             runs on the f32-input MFMA (attn_f32_kernel) instead of its three-term fp16 form
 */
enum { CIR_TUNE_GEMM_TILE = 0, CIR_TUNE_GEMM_GROUP_W = 1, CIR_TUNE_ATTN_SHARED_MAX = 2 };
int cir_set_tuning(int knob, int value);

/*
 * C[b] = act(A[b] * W[b]^T + bias[b]) (+ residual[b]),   b = 0..batch-1
 *   A (M,K) 16-bit row-major (lda), W (N,K) 16-bit row-major "torch Linear" layout (ldw),
 *   bias fp32 (N) or NULL, residual (M,N) (ldr) or NULL (added after the activation),
 *   C (M,N) (ldc): out_dtype = in_dtype (operand copy), CIR_F32 or CIR_F16 (residual stream; also from
 *   bf16 operands).  res_dtype: CIR_F32 (with a C in the operand type or fp32) or CIR_F16 (only with an fp16 C; an fp16 C
 *   from bf16 operands takes ONLY an fp16 residual - CIR_EDTYPE otherwise).  C may alias residual.
 *   Requirements: K % 64 == 0, N % 16 == 0, 16-byte aligned rows.
 *   in_dtype CIR_F32 (ABI v11, "exact" mode): A, W, residual and C all fp32 (out_dtype and res_dtype must be CIR_F32), K % 32 == 0,
 *   one fmaf chain per output starting at the bias (K ascending within each group of 16, see csrc/gemm.hip), GELU evaluated as
 *   written in the reference: 0.5 x (1 + erf(x / sqrt 2)) with erff.
 *   Rounding: one rounding of the fp32 result to the type of C - except an fp16 residual-stream C WITH a residual, which
 *   rounds A*W^T + bias to fp16 first, adds the residual in fp32 and rounds again.  Both tile sizes (128 x 128, persistent
 *   256 x 256; picked from M*N*batch) start their accumulators at the bias, walk K in the same order and share the epilogue
 *   arithmetic: the result does not depend on the tile choice, i.e. not on the batch size, in any output type (ABI v8).
 * Replaces every nn.Linear on the path: vit.py:35-41,72,84; med.py:158-168,250-251,319-333;
 * nlvr_encoder.py:150-168,250-264,383-396; blip_stage2.py:50-54 (first layer); the erf GELU is
 * ACT2FN['gelu'] (nlvr_encoder.py:376-379) / nn.GELU (vit.py:26).
 */
int cir_gemm_bias_act(const void* A, int64_t lda, int64_t strideA,
                      const void* W, int64_t ldw, int64_t strideW,
                      const float* bias, int64_t strideBias,
                      const void* residual, int res_dtype, int64_t ldr, int64_t strideR,
                      void* C, int64_t ldc, int64_t strideC,
                      int64_t M, int N, int K, int batch,
                      int act, int in_dtype, int out_dtype, void* stream);

/*
 * C = act(LayerNorm(X; gamma, beta, eps) * W^T + b) WITHOUT a LayerNorm pass (ABI v12): the GEMM reads the raw fp16 residual-stream
 * rows X (M, K) (ldx; K = the whole normalised row) and the caller hands over the affine folded into the weight:
 *     Wg (N, K) = W diag(gamma) rounded to fp16 (ldw),   colsum[n] = sum_k Wg[n,k] (fp32, of the ROUNDED values),
 *     bias[n] = b[n] + sum_k W[n,k] beta[k] (fp32)
 * so that  C[m,n] = act( rstd_m * (sum_k X[m,k] Wg[n,k] - mean_m * colsum[n]) + bias[n] ),  C (M, N) fp16 (ldc).
 * mean_m / rstd_m (biased variance E[x^2] - mean^2 in fp32, eps added before the root) are formed inside the kernel from the same
 * fragments it multiplies (csrc/gemm256.hip, LNF): no normalised copy of X is written or read.  act CIR_ACT_NONE | CIR_ACT_GELU;
 * dtype CIR_F16 only (CIR_EDTYPE otherwise: the rows are the fp16 stream).  K % 128 == 0, N % 16 == 0, N >= 64, 16-byte aligned rows.
 * The result does not depend on M (always the 256 x 256 kernel; a row's statistics are summed in one fixed order).
 * Replaces norm1 + attn.qkv and norm2 + mlp.fc1 of every ViT block: vit.py:107-109 with :72 / :36-37.
 */
int cir_gemm_ln_bias_act(const void* X, int64_t ldx, const void* Wg, int64_t ldw, const float* colsum, const float* bias,
                         void* C, int64_t ldc, int64_t M, int N, int K, float eps, int act, int dtype, void* stream);

/*
 * y = act(x),  hi = fp16(y),  lo = fp16(y - hi)   (ABI v13):  x (rows, cols) fp32 with leading dimension ldx; hi, lo (rows, cols) fp16
 *   with leading dimension ldo; hi2 (optional, same layout) receives a second copy of hi.
 *   act CIR_ACT_NONE | CIR_ACT_GELU (0.5 y (1 + erf(y / sqrt 2)), erff) | CIR_ACT_RELU.  cols % 8 == 0, ldx % 4 == 0, ldo % 8 == 0, 16-byte aligned.
 *   The operand split of the "text32" precision mode: with hi = buf, lo = buf + cols, hi2 = buf + 2 cols, ldo = 3 cols a row of buf is
 *   [hi | lo | hi]; against weight rows [W_hi | W_hi | W_lo] ONE cir_gemm_bias_act of depth 3 cols on fp16 operands forms
 *   A_hi W_hi^T + A_lo W_hi^T + A_hi W_lo^T in one fp32 accumulator: ~20 significant bits of a fp32 Linear at 3/16 of the f32-input MFMA's
 *   cost.  The MFMA keeps fp16 subnormals, so lo is stored unscaled.  No reference counterpart (the reference's fp32 Linear is what the
 *   three products approximate).
 */
int cir_split16(const float* x, int64_t ldx, void* hi, void* lo, void* hi2, int64_t ldo, int64_t rows, int cols, int act, void* stream);

/*
 * "split8" operand rows (ABI v14, round 6): y = act(x) as three byte segments per row,
 *     [ hi: cols x fp16 = fp16(y) | lo8: cols x e4m3 = e4m3((y - hi) 2^12) | hi8: cols x e4m3 = e4m3(hi) ]      = 4 cols bytes,
 *   both e4m3 terms clamped to +-448 first (OCP e4m3 has no infinity; the conversion does not saturate).  x (rows, cols) fp32 with leading
 *   dimension ldx (elements); out rows ldo_bytes apart.  act as cir_split16, GELU by an erf good to 1.5e-7.  cols % 16 == 0, 16-byte aligned.
 *   The A operand of cir_gemm_split8.  No reference counterpart (see there).
 */
int cir_split8(const float* x, int64_t ldx, void* out, int64_t ldo_bytes, int64_t rows, int cols, int act, void* stream);

/*
 * cir_attention on fp32 tensors (the CIR_F32 form: fp32 products, fp32 softmax) with the context written as split8 rows (ABI v14):
 *   out row (b1, b0, query) = [H*64 x fp16 | H*64 x e4m3 lo | H*64 x e4m3 hi] at out + b1 o_s1_bytes + b0 o_s0_bytes + query o_rs_bytes.
 *   Replaces BertSelfAttention.forward's core (nlvr_encoder.py:140-222, med.py:158-240) + the operand split in front of the attention
 *   output projection (nlvr_encoder.py:230-236) in the "text32" precision mode.  Strides of q / k / v in elements (% 4), of out in bytes (% 16).
 */
int cir_attention_split8(const float* q, int64_t q_s1, int64_t q_s0, int64_t q_rs, const float* k, int64_t k_s1, int64_t k_s0, int64_t k_rs,
                         const float* v, int64_t v_s1, int64_t v_s0, int64_t v_rs, const float* mask, int64_t m_s1, int64_t m_s0,
                         void* out, int64_t o_s1_bytes, int64_t o_s0_bytes, int64_t o_rs_bytes, int B1, int B0, int H, int Lq, int Lk,
                         float scale, void* stream);

/*
 * cir_layernorm on fp32 stream rows with the operand copy written as split8 rows (ABI v14): y_stream (fp32, may be NULL) as cir_layernorm
 *   writes it (bit for bit), y_split (batch, rows) rows of 4 cols bytes [fp16 | e4m3 lo | e4m3 hi] ld_split_bytes / stride_split_bytes apart.
 *   x / residual / y_stream strides per batch item in elements; gamma / beta (cols) or per batch item (strideG).  cols % 16 == 0, cols <= 1024.
 *   Replaces LayerNorm + the operand split in front of the text side's qkv and fc1 Linears (nlvr_encoder.py:262-264 / 469-476 -> :105-107 / 399).
 */
int cir_layernorm_split8(const float* x, int64_t strideX, const float* residual, int64_t strideR, const float* gamma, const float* beta,
                         int64_t strideG, float* y_stream, int64_t strideY, void* y_split, int64_t ld_split_bytes, int64_t stride_split_bytes,
                         int64_t rows, int cols, int batch, float eps, void* stream);

/*
 * C[b] = act(A[b] W[b]^T + bias[b]) (+ residual[b]) for a fp32 Linear whose operands travel as split8 rows (ABI v14; replaces the
 *   torch.nn.Linear calls of the text side - nlvr_encoder.py:105-107 / 230-236 / 399 / 411, med.py the same - in the "text32" precision
 *   mode, where the reference computes in fp32: validate_stage2.py:140-141):
 *     A (batch, M, K) rows [A_hi fp16 | e4m3(A_lo 2^12) | e4m3(A_hi)]           (cir_split8 / cir_layernorm_split8 / cir_attention_split8 /
 *                                                                                 this function with out_split = 1)
 *     W (batch, N, K) rows [W_hi fp16 | e4m3(W_hi 2^w_exp_hi) | e4m3(W_lo 2^w_exp_lo)],  W_hi = fp16(W), W_lo = W - W_hi
 *   ONE launch forms  A_hi W_hi^T  on the fp16 MFMA  +  A_lo W_hi^T + A_hi W_lo^T  on the block-scaled fp8 MFMA
 *   (v_mfma_scale_f32_16x16x128_f8f6f4, E8M0 scales 2^-12, 2^-w_exp_hi, 2^-w_exp_lo: twice the fp16 rate) in one fp32 accumulator that
 *   starts at the bias: the corrections are 2^-11 of the result and keep 4 significant bits per factor -> ~2^-16 of the result, against
 *   2^-11 for one fp16 product; cost 2 K / 64 K-tiles (three fp16 products: 3 K / 64; the f32-input MFMA: 16 K / 64).
 *   lda / ldw / strides in BYTES (rows are 4 K bytes long), K % 256 == 0, N % 16 == 0, 16-byte aligned.
 *   out_split 0: C (batch, M, N) fp32, ldc / strideC in elements, residual fp32 or NULL (only with CIR_ACT_NONE);
 *   out_split 1: C = split8 rows of act(.) (the next GEMM's operand: fc1 -> fc2), ldc / strideC in BYTES (>= 4 N), N % 64 == 0, no residual.
 *   The 128 x 128 and the persistent 256 x 256 kernel are chosen as in cir_gemm_bias_act and give the same bits.
 */
int cir_gemm_split8(const void* A, int64_t lda_bytes, int64_t strideA_bytes, const void* W, int64_t ldw_bytes, int64_t strideW_bytes,
                    const float* bias, int64_t strideBias, const float* residual, int64_t ldr, int64_t strideR,
                    void* C, int64_t ldc, int64_t strideC, int64_t M, int N, int K, int batch, int act, int out_split,
                    int w_exp_hi, int w_exp_lo, void* stream);

/*
 * y[b] = LayerNorm(x[b] (+ residual[b]); gamma[b], beta[b], eps) over the last dimension.
 *   x, residual (rows, cols) in x_dtype (CIR_F32 or CIR_F16: the residual stream); gamma/beta fp32 (cols);
 *   outputs: y_stream in y_stream_dtype (CIR_F32 / CIR_F16: the copy that feeds the next residual) and/or
 *   y16 (16-bit operand copy in dtype16); either may be NULL.  fp32 statistics.  cols % 4 == 0, cols <= 1024.
 *   Strides select the per-batch slice (stride 0 = shared).  Replaces nn.LayerNorm at vit.py:107-110,192
 *   (eps 1e-6) and med.py:253 / nlvr_encoder.py:252-264,395 (eps 1e-12), including the twin LayerNormA/B.
 */
int cir_layernorm(const void* x, int x_dtype, int64_t strideX, const void* residual, int64_t strideR,
                  const float* gamma, const float* beta, int64_t strideG,
                  void* y_stream, int y_stream_dtype, void* y16, int64_t strideY,
                  int64_t rows, int cols, int batch, float eps, int dtype16, void* stream);

/*
 * out = softmax(q k^T * scale + mask) v per (batch item, head), head_dim fixed at 64.
 *   Batch items are indexed (b1, b0), b1 < B1, b0 < B0; element (b1,b0,row,h,d) of q lives at
 *   q + b1*q_s1 + b0*q_s0 + row*q_rs + h*64 + d (same scheme for k, v, out); mask is an additive
 *   fp32 key mask (Lk) per item at mask + b1*m_s1 + b0*m_s0, or NULL.  kv_index (B1 int64, or NULL):
 *   item b1 reads its K/V from row kv_index[b1] of k / v instead of row b1 - the candidates of a
 *   query are then attended straight out of a per-image K/V bank (the cross-query reuse of
 *   SURVEY section 8(f)-1) with no gather copy.  16-bit in/out, fp32 softmax - or, dtype CIR_F32 (ABI v11), fp32 q / k / v / out
 *   with fp32 operands in both products (csrc/attention_f32.hip).  Replaces BertSelfAttention.forward (nlvr_encoder.py:140-222, med.py:158-240: the
 *   transpose_for_scores / matmul / scale / mask / softmax / matmul / merge-heads sequence) and
 *   Attention.forward's core (vit.py:73-83).
 */
int cir_attention(const void* q, int64_t q_s1, int64_t q_s0, int64_t q_rs,
                  const void* k, int64_t k_s1, int64_t k_s0, int64_t k_rs,
                  const void* v, int64_t v_s1, int64_t v_s0, int64_t v_rs,
                  const float* mask, int64_t m_s1, int64_t m_s0, const int64_t* kv_index,
                  void* out, int64_t o_s1, int64_t o_s0, int64_t o_rs,
                  int B1, int B0, int H, int Lq, int Lk, float scale, int dtype, void* stream);

/*
 * Cross-attention of ONE query row per (branch, head) over the tokens of each candidate, K / V projections folded out of
 * the token side (last fusion layer, where only the two CLS rows are used: nlvr_encoder.py:906-908 with :321-344):
 *   out[t][r][:] = sum_j softmax_j(qp[t][r] . x[t][j] * scale) x[t][j]     r < 32 rows, j < Lk keys, width D
 * x (T, Lk, D) 16-bit tokens (item stride x_s1, rows contiguous; with x_index (T int64, or NULL) item t reads the tokens of
 * row x_index[t] of an index-feature bank instead - the cross-query reuse of SURVEY section 8(f)-1), qp (T, 32, D) =
 * W_k^T q per row (rows beyond 2H: any finite values), out (T, 32, D).  The caller forms qp with a 64 x D GEMM per (branch, head) and applies W_v (+ b_v) to
 * `out` with a D x 64 GEMM (cir_gemm_bias_act, batched); the key bias is constant over j and drops out of the softmax.
 * D in {128, 256, .., 768}; no key mask.  Replaces, for that layer, the 4 D x D K|V projection of every candidate token
 * plus BertSelfAttention.forward's cross branch (nlvr_encoder.py:150-168, 183-217).
 */
int cir_cls_cross_attention(const void* x, int64_t x_s1, const int64_t* x_index, const void* qp, void* out, int T, int Lk, int D,
                            float scale, int dtype, void* stream);

/*
 * Cross-attention of ALL caption tokens of both branches over a candidate's image tokens with the key and value projections folded out of
 * the token side (ABI v11, round 5; csrc/xattn_fold.hip) - the two-branch BertLayer's crossattention.self{0,1} (nlvr_encoder.py:150-168,
 * 183-217, called :321-344) without ever forming K = X W_k^T or V = X W_v^T:
 *   out[t][tok][b][h 64 + d] = sum_f W_v[b][h 64 + d][f] (sum_j softmax_j((q[b][t L + tok][h] W_k[b][h]) . x[t][j] * scale) x[t][j][f]) + b_v[b][h 64 + d]
 * (the key bias is constant over j and drops out of the softmax; the rows of the softmax sum to 1, so the value bias adds as it is).
 * q (2, T L, D) 16-bit: the cross-attention QUERY projection incl. its bias (branch stride q_sb, row stride q_rs); x (T, N, D) 16-bit tokens
 * (item stride x_s1, rows contiguous); wkt / wvp (2, D, D each, branch stride w_sb): key.weight / value.weight re-ordered per MFMA fragment, one
 * contiguous KiB (64 lanes x 8 values; lane = 16 g + i) per wave-load -
 *   wkt[b] block ((u 2 + t) 12 + h) 2 + s : lane (g, i) holds key.weight[64 h + 32 s + 8 g + j][32 u + 8 (i >> 2) + 4 t + (i & 3)],  j < 8
 *   wvp[b] block (u 4 + e) 12 + h         : lane (g, i) holds value.weight[64 h + 16 e + i][32 u + (j < 4 ? 4 g + j : 16 + 4 g + j - 4)]
 * (u < D / 32 feature units, t < 2, s < 2, e < 4, h < 12; candidate_reranking_cir_amd/ops.py: fold_pack_key / fold_pack_value);
 * bv (2, D) fp32 value bias; out (T, L, 2, D)-shaped through strides.
 * D = 768, H = 12, L <= 32, N <= 608 (N <= 224: 48 query rows per wave, csrc/xattn_fold.hip; 225 .. 608 - the 384-px geometry's 577 tokens -: 16 rows per
 * wave, csrc/xattn_fold16.hip) (CIR_ESHAPE otherwise: use cir_gemm_bias_act + cir_attention).  key_mask (ABI v14; NULL = none): additive fp32
 * mask (T, N), rows mask_stride apart, shared by both branches - the (1 - attention_mask) * finfo.min of padded candidate token sets
 * (nlvr_encoder.py:863-868): logits = scores * scale + mask; masks below -2e38 are clamped there (all-masked rows stay uniform).  16-bit operands, fp32
 * accumulation and softmax; Q' = q W_k and C' = P X are rounded to the operand type where the projected path rounds K and V.
 * 614 MFLOP per (candidate, both branches) instead of 969 at 197 keys (3.0 instead of 5.7 GFLOP at 577), and no (T N, 4 D) K|V tensor.
 */
int cir_cross_attention_folded(const void* q, int64_t q_sb, int64_t q_rs, const void* x, int64_t x_s1, const void* wkt, const void* wvp, int64_t w_sb,
                               const float* bv, const float* key_mask, int64_t mask_stride, void* out, int64_t o_st, int64_t o_sr, int64_t o_sb,
                               int T, int L, int N, int D, int H, float scale, int dtype, void* stream);

/*
 * BertEmbeddings.forward (nlvr_encoder.py:68-91, med.py:87-110):
 *   y[r] = LayerNorm(word[ids[r]] + pos[r % L]) for r < rows; fp32 tables, outputs as cir_layernorm
 *   (y_stream in CIR_F32 / CIR_F16, y16 in dtype16).
 */
int cir_embed_layernorm(const int64_t* ids, const float* word, const float* pos,
                        const float* gamma, const float* beta, void* y_stream, int y_stream_dtype, void* y16,
                        int64_t rows, int L, int cols, int vocab, float eps, int dtype16, void* stream);

/*
 * timm PatchEmbed's im2col (vit.py:182; Conv2d k=p s=p == GEMM over flattened patches):
 *   patches[(b*gh+py)*gw+px][c*p*p + ky*p + kx] = image[b][c][py*p+ky][px*p+kx]
 *   image fp32 or 16-bit (img_dtype), patches 16-bit (dtype16) - or fp32 from an fp32 image (dtype16 = CIR_F32, ABI v11).
 */
int cir_patchify(const void* image, int img_dtype, void* patches, int dtype16,
                 int B, int C, int H, int Wd, int patch, void* stream);

/*
 * Token assembly (vit.py:184-187): x[b][0] = cls + pos[0]; x[b][1+i] = proj[b*P+i] + pos[1+i].
 *   proj (B*P, D) (patch-embed GEMM output incl. bias) and x in stream_dtype (CIR_F32 / CIR_F16),
 *   cls (D), pos (P+1, D) fp32 parameters.  D % 8 == 0.
 */
int cir_vit_assemble(const void* proj, const float* cls, const float* pos, void* x, int stream_dtype,
                     int B, int P, int D, void* stream);

/*
 * y (M, N) fp32 = x (M, K) 16-bit * W (N, K) 16-bit ^T + bias, for tiny N (<= 8): the last
 * Linear of cls_head (blip_stage2.py:53, 134-136).  K % 8 == 0.
 */
int cir_small_linear(const void* x, int64_t ldx, const void* W, const float* bias, float* y,
                     int64_t M, int N, int K, int dtype, void* stream);

/*
 * dst[i] = convert(src[index[i]]) for rows of `row_elems` elements (index NULL = identity).
 * Replaces the per-query candidate gather torch.stack(itemgetter(*names)(name_to_feat))
 * (validate_stage2.py:115, 251, 266), the `.expand(K, ...)` of z_t / ids to the K candidates
 * (blip_stage2.py:118-121) and dtype conversion at the boundary.  row_elems % 8 == 0; dtype pairs:
 * f32->f32/bf16/f16, bf16->bf16/f32, f16->f16/f32.  Indices are clamped to [0, src_rows).
 */
int cir_gather_rows(const void* src, int src_dtype, const int64_t* index, void* dst, int dst_dtype,
                    int64_t n_rows, int64_t row_elems, int64_t src_rows, void* stream);

/*
 * Per-row descending argsort (validate_stage2.py:53,174,188: torch.argsort(..., descending=True)):
 *   idx[q][j] = index of the j-th largest logit of row q, ties broken by lower index.  K <= 8192
 *   (also ranks a whole index for stage I: validate.py:58, 203 sort ascending distances = descending -distance).
 */
int cir_topk_desc(const float* logits, int64_t* idx, int Q, int K, void* stream);

/*
 * fp32 y (M,N) = x (M,K) W(N,K)^T + bias (mode 0), 1 - x W^T (mode 1) or x W^T - 1 (mode 2, the exact
 * negative of mode 1, so that cir_topk_desc ranks by ascending distance): stage I's 256-d heads
 * vision_proj / text_proj (blip_stage1.py:42-43, 58, 83) and its distance matrix
 * `1 - predicted_features @ index_features.T` (validate.py:57, 202).  x rows at stride ldx.
 */
int cir_linear_f32(const float* x, int64_t ldx, const float* W, const float* bias, float* y,
                   int64_t M, int N, int K, int mode, void* stream);

/* y = x / max(||x||_2, 1e-12) row-wise, fp32 (F.normalize at blip_stage1.py:58, 83). */
int cir_l2_normalize(const float* x, float* y, int64_t rows, int cols, void* stream);


/* ------------------------------------------------------------------------------------------------------------------------
 * Training-mode operators: what BLIP_NLVR.img_txt_fusion in train() mode and its backward need
 * besides the operators above (blip_stage2.py:65-99 driven by stage2_train.py:202-216; dropout nlvr_encoder.py:86-90, 207,
 * 250-264, 397).  The dense layers' forward and dgrad run on cir_gemm_bias_act (dgrad over a transposed weight copy, cir_transpose16),
 * the weight gradients on cir_bmm.
 * Dropout is counter-based: element i of a launch is kept iff hash(seed, i) >= p, scaled by 1 / (1 - p) (stand-alone operators: splitmix64 of
 * (seed, flat index); fused operators: the pair hash described at cir_residual_layernorm_train); the backward
 * operators regenerate the same mask from the same (seed, p) - no mask tensor exists.  None of these is on the inference path.
 */
/* dst[b][c][r] = src[b][r][c], 16-bit elements (dtype CIR_BF16 / CIR_F16 names the payload only). */
int cir_transpose16(const void* src, void* dst, int rows, int cols, int64_t ld_src, int64_t ld_dst, int batch, int64_t s_src, int64_t s_dst,
                    int dtype, void* stream);
/* `count` transposes in one launch (ABI v10): matrix i - rows_i x cols_i, contiguous - at element offset off_i of src is written transposed at
 * the same offset of dst.  table (DEVICE memory, 4 int64 per matrix): {off_i, rows_i, cols_i, index of its first 32 x 32 tile}; total_tiles =
 * the sum of the tile counts.  The trainer transposes every trained weight of its flat 16-bit parameter buffer with it, once per step. */
int cir_transpose16_multi(const void* src, void* dst, const int64_t* table, int count, int64_t total_tiles, int dtype, void* stream);
/* C[z] = alpha * op(A[z]) * op(B[z]) (+ C[z] if accumulate): op(A) (M,K) from A stored (M,K) [trans_a 0] or (K,M) [1]; op(B) (K,N)
 * from B stored (K,N) [trans_b 0] or (N,K) [1].  Any extents, fp32 accumulate.  Two batch levels: z = z1 * nb2 + z2 with a stride per
 * level and operand (element units; 0 broadcasts), nb1 * nb2 <= 65535.  in_dtype CIR_BF16 / CIR_F16: MFMA kernel (64 x 64 tiles,
 * edges predicated, 16-byte loads when base / ld / strides allow, 2-byte loads otherwise), out_dtype = in_dtype or CIR_F32;
 * in_dtype CIR_F32: plain kernel, fp32 out.  Used for the un-fused attention of the training pass (Q K^T, P V and their four
 * adjoints per (candidate or triplet, head)) and, with trans_a = 1 and a batch over row chunks, for the weight gradients.
 * accumulate: 0 C = ..., 1 C += ... (read-modify-write: batch items must own their C), 2 (ABI v10; MFMA kernel, fp32 C) atomic C += ... -
 * batch items may share one C (stride 0): the row chunks of a weight gradient sum straight into dW, in no fixed order. */
int cir_bmm(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int trans_a, int trans_b,
            int nb1, int nb2, int64_t sA1, int64_t sA2, int64_t sB1, int64_t sB2, int64_t sC1, int64_t sC2, float alpha, int accumulate,
            int in_dtype, int out_dtype, void* stream);
/* Weight gradient of a dense layer (ABI v10; train_wgrad.hip): dw (N, K) fp32 (row stride ldw) += dy^T x, dy (rows, N) and x (rows, K) 16-bit
 * row-major (strides ldy, ldx; multiples of 8) read AS STORED - LDS-DMA staging, transposing LDS reads, 128 x 128 output tiles, the rows split
 * over `splits` workgroups per tile (0: chosen from the shape) that add their partial tiles into dw atomically (no fixed summation order).
 * N % 128 == 0 and K % 128 == 0 (CIR_ESHAPE otherwise: use cir_bmm with trans_a); any row count (a tail below 64 rows runs on cir_bmm's kernel).
 * Adjoint of nn.Linear's weight (nlvr_encoder.py:150-168, 250-264, 383-409 under stage2_train.py:216's backward). */
int cir_wgrad(const void* dy, int64_t ldy, const void* x, int64_t ldx, float* dw, int64_t ldw, int64_t rows, int N, int K, int splits, int in_dtype,
              void* stream);
/* Up to 16 weight gradients in ONE launch: dw_i += dy_i^T x_i.  The 13 weight gradients of one two-branch BertLayer together have ~940 output
 * tiles - enough to fill the chip with every tile's row sum formed by one workgroup (plain adds, no atomics; a 768 x 768 weight alone has 36
 * tiles and needs an 8 .. 16-way row split whose atomic traffic costs as much as the product).  Units are balanced by splitting the longer row
 * counts (the FFN's stacked 2R rows) to the shortest one's length - and, when the group fills the chip, halved once more (two workgroups' atomic
 * adds per tile: measured faster than the tail of a 2.4-round launch).  Same shape rules as cir_wgrad; splits 0 = automatic.
 * `problems` is a HOST array (the one exception to "every pointer is a device pointer": it is read during the call, its members dy / x / dw are
 * device pointers); count <= 16 (CIR_ESHAPE beyond: call again). */
typedef struct { const void* dy; int64_t ldy; const void* x; int64_t ldx; float* dw; int64_t ldw; int64_t rows; int N, K, splits; } cir_wgrad_desc;
int cir_wgrad_grouped(const cir_wgrad_desc* problems, int count, int in_dtype, void* stream);
/* P = softmax(S * scale + mask) per row (S fp32 (rows, cols); mask fp32 (cols) shared by each group of rows_per_mask rows, or
 * NULL), Pd = dropout(P, p_drop, seed); both 16-bit (dtype).  nlvr_encoder.py:183-207. */
int cir_softmax_dropout(const float* S, int64_t ld_s, const float* mask, int64_t rows_per_mask, int64_t ld_mask, void* P, void* Pd, int64_t ld_p,
                        int64_t rows, int cols, float scale, float p_drop, uint64_t seed, int dtype, void* stream);
/* dS = scale * P * (dP - sum_cols(dP * P)), dP = dropout-backward(dPd) with the forward's (p_drop, seed); dPd fp32, dS 16-bit. */
int cir_softmax_dropout_bwd(const void* P, int64_t ld_p, const float* dPd, int64_t ld_d, void* dS, int64_t ld_ds, int64_t rows, int cols,
                            float scale, float p_drop, uint64_t seed, int dtype, void* stream);
/* FUSED attention of the training pass (round 4): out = dropout(softmax(q k^T * scale + mask)) v per (group, head), head dimension 64, with
 * the log2-domain log-sum-exp of every row written to lse (G, H, Lq) - and its recomputing backward: dq / dk / dv (grad_dtype: CIR_F32, or the operand type - ABI v10 - so that the
 * projection's dgrad / wgrad products read them as they are) from q, k, v,
 * out, d_out and lse; the probabilities are recomputed tile by tile and the dropout mask regenerated from its counter (the pair hash of the
 * fused kernels, see below: row = (g * H + h) * Lq + query, col = key), so no score / probability tensor exists in memory.  Tensors are head views: element
 * (g, h, row, d) at base + g * x_sg + h * x_sh + row * x_sr + d; out and d_out share one layout; mask fp32 (G, Lk) additive or NULL;
 * dsum_scratch fp32 (G * H * Lq) receives rowsum(d_out * out) - with out taken from its optional fp32 twin out32 (written by the
 * forward, same layout): it then equals sum_j Pd_ij dPd_ij term for term and the row sums of dS vanish.  Replaces, for BertSelfAttention.forward in train() mode
 * (nlvr_encoder.py:140-222, dropout :207) and its adjoint, the cir_bmm / cir_softmax_dropout[_bwd] chain. */
int cir_attention_train_fwd(const void* q, int64_t q_sg, int64_t q_sh, int64_t q_sr, const void* k, int64_t k_sg, int64_t k_sh, int64_t k_sr,
                            const void* v, int64_t v_sg, int64_t v_sh, int64_t v_sr, const float* mask, void* out, int64_t o_sg, int64_t o_sh,
                            int64_t o_sr, float* out32, float* lse, int G, int H, int Lq, int Lk, float scale, float p_drop, uint64_t seed,
                            int dtype, void* stream);
int cir_attention_train_bwd(const void* q, int64_t q_sg, int64_t q_sh, int64_t q_sr, const void* k, int64_t k_sg, int64_t k_sh, int64_t k_sr,
                            const void* v, int64_t v_sg, int64_t v_sh, int64_t v_sr, const float* mask, const void* out, const void* d_out,
                            int64_t o_sg, int64_t o_sh, int64_t o_sr, const float* out32, const float* lse,
                            float* dsum_scratch, void* dq, int64_t dq_sg,
                            int64_t dq_sh, int64_t dq_sr, void* dk, int64_t dk_sg, int64_t dk_sh, int64_t dk_sr, void* dv, int64_t dv_sg,
                            int64_t dv_sh, int64_t dv_sr, int grad_dtype, int G, int H, int Lq, int Lk, float scale, float p_drop, uint64_t seed,
                            int dtype, void* stream);
/* LayerNorm backward from the saved fp32 input x of the forward: dx (written), dgamma / dbeta (fp32, ACCUMULATED atomically). */
int cir_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, int64_t rows, int cols,
                      float eps, void* stream);
/* FUSED row passes of the training step (ABI v10; train_fused.hip).  BertSelfOutput / BertOutput in train() mode
 * (nlvr_encoder.py:248-264, 399-409: dense -> dropout -> + residual -> LayerNorm, incl. the averaged two-branch merge :257-260):
 *   pre = dropout(alpha * (t0 + t1), p_drop, seed) + residual;  y = LayerNorm(pre; gamma, beta, eps)
 * t0 / t1 (t1 may be NULL) / residual fp32 (rows, cols) contiguous; pre (fp32, kept for the backward), y32 (fp32 or NULL) and y16 (16-bit in
 * dtype16, or NULL) are written.  cols % 4 == 0, cols <= 1024.
 * Dropout of the FUSED kernels (this one, its adjoint, cir_attention_train_*; ABI v10): element (row, col) is dropped iff its 16 random bits
 * are below round(p * 65536); the bits of the column pair (2j, 2j + 1) are the low / high half of
 *   hash32((uint32)seed + (uint32)(seed >> 32) * 0x85EBCA6B + (uint32)row * 0x9E3779B9 + (uint32)(row >> 32) * 0xC2B2AE35  ^  j),
 * hash32(x): x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15; x *= 0x846ca68b; x ^= x >> 16 - two 32-bit multiplies per pair instead of the three
 * 64-bit ones per element of the stand-alone kernels' splitmix64 (which made the hash ~10x the MFMA time of an attention tile). */
int cir_residual_layernorm_train(const float* t0, const float* t1, const float* residual, const float* gamma, const float* beta, float* pre,
                                 float* y32, void* y16, int64_t rows, int cols, float eps, float alpha, float p_drop, uint64_t seed, int dtype16,
                                 void* stream);
/* Its adjoint: dx = d pre from dy and the saved pre (x) as cir_layernorm_bwd (dx may be NULL when only the dense branch is wanted), dgamma / dbeta
 * ACCUMULATED; and, when dt16 is given, the gradient of the dense branch  dt = alpha * dropout'(dx + t_add)  (t_add fp32 or NULL: a second
 * LayerNorm's d pre over the SAME dense output, as behind the two-branch merge) written as the 16-bit operand (dtype16) of that layer's dgrad /
 * wgrad products, with its column sums ACCUMULATED into dbias and dbias2 (either may be NULL): the dense layer's bias gradient. */
int cir_layernorm_bwd_fused(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, const float* t_add,
                            void* dt16, float* dbias, float* dbias2, int64_t rows, int cols, float eps, float alpha, float p_drop, uint64_t seed,
                            int dtype16, void* stream);
/* 16-bit rows (rows, cols), row strides lda / ldz / ldo (multiples of 8), cols % 8 == 0.  mode 0: sums[c] += sum_r a[r][c] (bias gradient of a
 * dense layer whose output gradient is 16-bit: behind the attention adjoints);  mode 1: out = a * gelu'(z) (erf form) and, when sums is given,
 * sums[c] += sum_r out[r][c] - BertIntermediate's adjoint (nlvr_encoder.py:383-396) with the bias gradient in the same pass. */
int cir_rows16_colsum(const void* a, int64_t lda, const void* z, int64_t ldz, void* out, int64_t ldo, float* sums, int64_t rows, int cols, int mode,
                      int dtype, void* stream);
/* out[r] = a[r] + scale[r / rows_per_group] * b[r] over rows of `cols` fp32 values (a may be NULL: out = scale * b); out fp32 or 16-bit
 * (out_dtype).  DropPath - timm's per-sample stochastic depth around both branches of a ViT block (vit.py:98-109; blip_stage2.py:37 builds the
 * stage-II image encoder with drop_path_rate 0.1) - in the ViT fine-tuning pass: the forward adds the sample-scaled branch to the stream, the
 * backward writes the sample-scaled stream gradient as the 16-bit operand of the branch's adjoint products.  cols % 4 == 0. */
int cir_rows_scale_add(const float* a, const float* b, const float* scale, void* out, int64_t rows, int cols, int64_t rows_per_group, int out_dtype,
                       void* stream);
/* mode 0: out = gelu(z) (erf form, ACT2FN['gelu']); 1: out = dy * gelu'(z); 2: relu(z); 3: dy * (z > 0); 4: dropout(z, p_drop, seed);
 * 5: z + dy; 6: p_drop * z (scale).  z in z_dtype (CIR_F32 / CIR_BF16 / CIR_F16), dy fp32, out in out_dtype. */
int cir_eltwise(const void* z, int z_dtype, const float* dy, void* out, int out_dtype, int64_t n, int mode, float p_drop, uint64_t seed, void* stream);
/* out[c] += sum_r x[r][c] (bias gradients; fp32, atomics). */
int cir_colsum(const float* x, int64_t ld, float* out, int64_t rows, int cols, void* stream);
/* BertEmbeddings backward (nlvr_encoder.py:49-91): dword[ids[r]] += dy[r], dpos[r % L] += dy[r] (fp32, atomics). */
int cir_embed_bwd(const int64_t* ids, const float* dy, float* dword, float* dpos, int64_t rows, int L, int cols, void* stream);
/* torch.optim.AdamW step in place on fp32 parameter / moments (stage2_train.py:120-126 builds that optimizer). */
int cir_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                   int step, void* stream);
/* The same step with GradScaler.step's found_inf decision (stage2_train.py:215-218: `scaler.step(optimizer)` skips the update when the
 * unscaled gradients hold an inf / NaN) taken ON THE DEVICE - no host read between backward and the update (ABI 15).  `state` = 8 x 32 bit
 * of device memory, zero-initialised by the caller once: [0] found_inf of the step in flight (the caller clears it before the first check of
 * a step), [1] applied steps t, [2] skipped steps, [3] / [4] fp32 bias corrections 1 - beta^t of the step in flight.
 *   cir_grads_check     g[i] *= scale (scale != 1: GradScaler.unscale_) and state[0] |= any non-finite g[i]; one pass over g.
 *   cir_adamw_begin     after every check of the step: flag set -> state[2] += 1; else state[1] += 1 and the corrections for the new t.
 *   cir_adamw_step_dev  cir_adamw_step's update with t / corrections from `state`; a no-op while state[0] is set.  `p16` (optional):
 *                       the 16-bit operand copy of the updated parameters (dtype16 = CIR_F16 / CIR_BF16), written in the same pass. */
int cir_grads_check(float* g, int64_t n, float scale, int32_t* state, void* stream);
int cir_adamw_begin(int32_t* state, float beta1, float beta2, void* stream);
int cir_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                       const int32_t* state, void* p16, int dtype16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CIRRANK_H */
