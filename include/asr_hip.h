/* libasr_hip.so -- C ABI of the MI355X (gfx950) kernels behind the Transformer-ASR training step.
 *
 * The reference (gentaiscool/end2end-asr-pytorch) is pure Python: its "FFI" is the set of ATen calls its
 * modules make.  Each entry point below replaces one such call site (cited as file:line relative to the
 * reference tree) with a hand-written HIP kernel.  The host-side binding a maintainer would add is a
 * ctypes stub; ours lives in end2end-asr-pytorch_amd/asr_hip/lib.py.
 *
 * Conventions (every entry point)
 *   - returns ASR_OK (0) or a negative ASR_E* code; never throws, never allocates or frees device memory,
 *     never synchronises; all buffers are caller-owned device pointers; work is enqueued on `stream`.
 *   - `dtype` selects the storage type of activations / shadow weights: ASR_F32 (parity mode) or ASR_BF16
 *     (perf mode, bf16 in / fp32 accumulate).  Parameters, gradients of parameters, LayerNorm statistics,
 *     logits and losses are always fp32.  Token ids are int64, lengths int32, masks uint8 (1 = keep / 1 = masked
 *     as documented per argument).
 *   - leading dimensions / strides are in ELEMENTS.
 */
#ifndef ASR_HIP_H_
#define ASR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* asr_stream_t; /* == hipStream_t */

enum { ASR_OK = 0, ASR_EINVAL = -1, ASR_ELAUNCH = -2, ASR_EUNSUPPORTED = -3, ASR_ERUNTIME = -4 };
enum { ASR_F32 = 0, ASR_BF16 = 1 };
/* asr_attn_bwd `parts`: the backward is three kernels (delta = rowsum(dO*O); dQ; dK/dV).  dQ and dK/dV only depend on delta, so
 * a caller may launch them on two streams: parts = DELTA on the first, then DQ on one and DKV on the other.                     */
enum { ASR_ATTN_DELTA = 1, ASR_ATTN_DQ = 2, ASR_ATTN_DKV = 4, ASR_ATTN_ALL = 7 };
enum { ASR_GEMM_RELU = 1, ASR_GEMM_ACCUMULATE = 2 };
/* op ids for the built-in HIP-event profiler (asr_prof_*) */
enum {
  ASR_OP_GEMM = 0, ASR_OP_CONV_IGEMM = 1, ASR_OP_CONV_WGRAD = 2, ASR_OP_ATTN_FWD = 3, ASR_OP_ATTN_BWD = 4,
  ASR_OP_ADD_LN = 5, ASR_OP_CE = 6, ASR_OP_ADAM = 7, ASR_OP_CONV1 = 8, ASR_OP_POOL = 9, ASR_OP_LAYOUT = 10,
  ASR_OP_COUNT = 16
};

const char* asr_strerror(int code);
int asr_abi_version(void);   /* 4 since round 4: asr_gemm_nt_add_ln removed; the state block of asr_adam_noam_step is {seed counter, step, skipped flag, unused} */

/* ---- tuning / A-B switches.  The library reads no environment variables and has no other mutable global state than
 * this table and the profiling slots below: a switch (names: the ASR_* list of DESIGN.md section 4 without the prefix,
 * e.g. "C64", "GEMM_TILE", "ATTN_GENERIC") keeps its built-in default until set here.  ASR_EINVAL for an unknown name.
 * asr_clear_tuning(NULL) restores every default.                                                                */
int asr_set_tuning(const char* name, int64_t value);
int asr_clear_tuning(const char* name);

/* ---- profiling: bracket every launch of one op id with hipEvents on its own stream ------------------------- */
int asr_prof_enable(int op_id, int enable);           /* enable=1 starts a fresh capture for op_id            */
int asr_prof_collect(int op_id, double* total_ms, int64_t* launches); /* synchronises the recorded events      */

/* ---- dense contraction: nn.Linear / Conv1d(k=1) forward, dgrad, wgrad ----------------------------------------
 * C[M,N] (op)= alpha * sum_k A[m*lda+k] * B[n*ldb+k]  (+ bias[n]) (ReLU).  Both operands K-contiguous.
 * flags: ASR_GEMM_RELU, ASR_GEMM_ACCUMULATE (C += ...).  splits>1: split-K with fp32 atomics (0 = auto), requires
 * out_dtype F32 and ACCUMULATE.  Replaces common_layers.py:136-142 (FFN), :181-187 (Q/K/V), :197 (out proj),
 * transformer.py:172 (input_linear), :302 (output_linear) and their autograd backward.                        */
int asr_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias,
                const void* relu_mask /* optional (M,ldc) tensor of the input dtype: C = 0 where relu_mask <= 0 */,
                int M, int N, int K, float alpha, int flags, int splits, int in_dtype, int out_dtype,
                asr_stream_t stream);

/* Weight gradient without transposes: C[N,K] (fp32) += sum_m A[m*lda+n] * B[m*ldb+k]; colsum_acc[n] += sum_m A[m,n]
 * (bias gradient, optional).  A = dY (M,N), B = X (M,K) row-major as the forward produced them.  Needs 16-byte aligned
 * rows, else ASR_EUNSUPPORTED (callers fall back to asr_transpose + asr_gemm_nt); any M (a partial last stage is
 * zero-filled in LDS).  splits <= 0: automatic split over m.  workspace (fp32, >= asr_gemm_tn_workspace(...) elements,
 * optional): the m-slices write partial tiles there and a second kernel folds them into C; without it they meet in fp32
 * atomics on C (slower, and limited to 4 slices).  ldb < K is allowed: the rows of B are then overlapping windows of K
 * elements, ldb apart, over one longer buffer (a convolution along the contiguous axis without im2col: asr_hip/functions.py,
 * EmbCNNFn); the caller guarantees that (M - 1) * ldb + K rounded up to 8 elements are readable.                     */
int64_t asr_gemm_tn_workspace(int M, int N, int K, int splits, int dtype);
int asr_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* colsum_acc,
                float* workspace, int64_t workspace_floats, int M, int N, int K, int splits, int dtype,
                asr_stream_t stream);

/* The weight (and bias) gradients of up to 48 linear layers in ONE launch: dw[i] (N[i],K[i]) fp32 += dy[i][:, :N]^T x[i][:, :K],
 * db[i] (N[i]) += column sums of dy[i] (db[i] may be NULL), each over M[i] rows; bf16 operands with 16-byte aligned rows.  A weight
 * gradient is off the critical path of backward (models/common_layers.py:136-142,181-187 leave it to autograd's order): queued and
 * launched together the layers fill the chip where each alone is a latency-bound chain of 16 - 64 blocks.  Arrays are HOST
 * arrays of n entries (n <= 48; 32 until round 4); returns ASR_EUNSUPPORTED when a layout does not fit (the caller then uses asr_gemm_tn per layer). */
int asr_gemm_tn_grouped(int n, const void* const* dy, const int64_t* ld_dy, const void* const* x, const int64_t* ld_x,
                        float* const* dw, const int64_t* ld_dw, float* const* db, const int* M, const int* N, const int* K,
                        int dtype, asr_stream_t stream);
/* Data gradient without transposed weights: C[M,N] (op)= alpha * sum_k A[m*lda+k] * B[k*ldb+n] with B = W (K,N) in its
 * master layout (transposing LDS reads).  flags: ASR_GEMM_ACCUMULATE; relu_mask as in asr_gemm_nt.  Needs 16-byte aligned
 * rows; K is contracted in stages of 64 (bf16) / 32 (fp32): when K is not a multiple of that, lda must cover K rounded up to a
 * whole stage and the caller guarantees A[:, K:] == 0 there (B's rows are clamped), else ASR_EUNSUPPORTED.              */
int asr_gemm_nn(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* relu_mask, int M,
                int N, int K, float alpha, int flags, int in_dtype, int out_dtype, asr_stream_t stream);
/* The encoder input projection's data gradient WITH the second max-pool's backward in its epilogue (reference: the autograd of
 * models/asr/transformer.py:50-52, 74-76, 172 -- MaxPool2d, view / transpose, encoder.input_linear): the product
 * G (M, N) = A (M, K) . Bp (K, N), M = Bt * W2 rows (b, w2), N = H2 * C columns in (h2, c) order -- Bp is the projection weight with its
 * input columns PERMUTED from the model's (c, h2) order (asr_permute_cols_tcf) -- is never stored: every 16-byte piece (8 channels of
 * pooled pixel (h2, w2)) is rounded to bf16 and routed through its 8 selection bytes code_cl[((b W2 + w2) H2 + h2) C + c] (1 + k = window
 * position k in scan order (0, 0), (0, 1), (1, 0), (1, 1); 0 = none) into dy (Bt, 2 H2, 2 W2, C) NHWC, the gradient of the un-pooled
 * conv output: all four window positions are written (zeros where not selected), so dy needs no clearing.  Equals asr_gemm_nn into the
 * (b, w2, c, h2) layout followed by asr_maxpool_bwd_code bit for bit (same products, same rounding, then a selection).
 * bf16, K % 64 == 0, C % 8 == 0, 16-byte aligned rows, else ASR_EUNSUPPORTED.                                              */
int asr_gemm_nn_poolbwd(const void* A, int64_t lda, const void* Bp, int64_t ldb, const uint8_t* code_cl, void* dy, int M, int K, int H2,
                        int W2, int C, int dtype, asr_stream_t stream);
/* dst (rows, H2 * C) <- src (rows, C * H2): dst[r][h2 * C + c] = src[r][c * H2 + h2] (row strides ld_src / ld_dst elements): the
 * column permutation between the model's feature order c * F' + f (transformer.py:74-76) and the channel-last order of
 * asr_gemm_nn_poolbwd.  bf16 or fp32; C % 8 == 0 (bf16) / C % 4 == 0 (fp32); dst rows 16-byte aligned.                    */
int asr_permute_cols_tcf(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int C, int H2, int dtype,
                         asr_stream_t stream);
/* What asr_gemm_tn_grouped would do with these n problems (host only, nothing is launched): returns 1 = one workgroup per WHOLE block of
 * dW, dispatched longest first, no atomics (splits[i] > 1: a block longer than a CU's share of the launch is cut into that many
 * slices of its rows), 0 = the shared-block forms of round 3 (equal pieces / slices, fp32 atomics) because whole blocks would leave
 * CUs idle; < 0 = error.  The dispatch is played through on 256 CUs for a few slice lengths (csrc/gemm.hip tn_rot_plan).        */
int asr_gemm_tn_grouped_plan(int n, const int* M, const int* N, const int* K, int* splits);
/* The data gradient that IS the attention backward's dO (the output projection's, models/common_layers.py:190-198 under autograd),
 * with the softmax backward's row term from the same epilogue: C (M, N) bf16 = A (M, K) . B (K, N) as asr_gemm_nn (alpha = 1, no mask,
 * no +=, ldc = N), and rowdot[(b H + h) T + q] = sum_{d < 64} C[b T + q][64 h + d] * O[b T + q][64 h + d] with the ROUNDED C, H = N / 64,
 * O = O32 (fp32, the un-rounded attention output) when given, else O (bf16); both (M, N) contiguous.  This is asr_attn_bwd's `delta`
 * (then called without ASR_ATTN_DELTA): one dependent launch less per attention block.  bf16, N % 64 == 0, M % T == 0, else
 * ASR_EUNSUPPORTED (callers keep asr_gemm_nn + ASR_ATTN_DELTA).                                                          */
int asr_gemm_nn_rowdot(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, const void* O, const float* O32,
                       float* rowdot, int M, int N, int K, int T, int dtype, asr_stream_t stream);
/* A linear layer's whole backward in ONE launch (every nn.Linear / Conv1d(k=1) of models/common_layers.py:136-142,181-197 under
 * autograd): dx (M,K) (+)= dy (M,N) . w (N,K) [flags: ASR_GEMM_ACCUMULATE; relu_mask as in asr_gemm_nt] from the data-gradient
 * workgroups of asr_gemm_nn, and next to them in the same grid the weight-gradient workgroups: the partial 64 x 64 tiles of
 * dy^T . x over `splits` slices of the M rows go to workspace [split][tile][64][64] (asr_gemm_nn_tn_workspace floats), the bias
 * gradient db (N, may be NULL) += column sums of dy.  asr_tn_reduce_multi then folds the slices of up to any number of layers
 * into their dw (N,K; fp32; += semantics) in one launch per 48 layers, slices added in a fixed order.  Two launches on two streams
 * cost a fork and a join (5-10 us each on the critical path of a replayed graph) per layer.  bf16 only; 16-byte aligned rows;
 * ld_dy >= N rounded up to 64 with dy[:, N:] == 0 there (as asr_gemm_nn), else ASR_EUNSUPPORTED.  splits = 0: chosen by the library (asr_gemm_nn_tn_splits tells the number).
 * fold_ws != NULL: extra workgroups of the same launch fold the partial tiles an EARLIER asr_gemm_nn_tn call on this stream left
 * in fold_ws into fold_dw (fold_N x fold_K, row stride fold_ld, fold_splits slices) -- the HBM-bound second stage of layer i
 * rides under the MFMA-bound tiles of layer i+1 instead of being a launch of its own.                                         */
int asr_gemm_nn_tn_splits(int M, int splits);
int64_t asr_gemm_nn_tn_workspace(int M, int N, int K, int splits);
int asr_gemm_nn_tn(const void* dy, int64_t ld_dy, const void* w, int64_t ldw, const void* x, int64_t ldx, void* dx, int64_t ld_dx,
                   const void* relu_mask, float* db, float* workspace, int64_t workspace_floats, int M, int N, int K, int flags,
                   int splits, int dtype, const float* fold_ws, float* fold_dw, int64_t fold_ld, int fold_N, int fold_K,
                   int fold_splits, asr_stream_t stream);
int asr_tn_reduce_multi(const float* const* workspaces, float* const* dw, const int64_t* ld_dw, const int* N, const int* K,
                        const int* splits, int count, asr_stream_t stream);
/* dst[i] = (dtype) src[i] : the one-launch refresh of the flat compute-dtype weight shadow after an optimiser step  */
int asr_cast_flat(const float* src, void* dst, int64_t n, int dtype, asr_stream_t stream);
/* dst[i] = (float) src_bf16[i]: the way back from the bf16 wire format of the data-parallel gradient exchange (--grad-wire bf16:
 * asr_cast_flat into a staging buffer, all-reduce of the staging buffer, this).  Both pointers 16-byte aligned.                  */
int asr_widen_flat(const void* src_bf16, float* dst, int64_t n, asr_stream_t stream);

/* out[c*ld_out + r] = in[r*ld_in + c]   (operand preparation for dgrad / wgrad).  If colsum_acc != NULL also
 * colsum_acc[c] += sum_r in[r,c]  (the bias gradient, from the tile that is in LDS anyway).                    */
int asr_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols, float* colsum_acc,
                  int dtype, asr_stream_t stream);
/* fp32 (rows,cols; ld_src) -> copy (rows,cols; ld_dst) and transpose (cols,rows; ld_dst_t) in `dtype`; either dst
 * may be NULL.  Used for weight shadows and for fp32 gradients entering a bf16 backward.                        */
int asr_cast_weight(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, void* dst_t, int64_t ld_dst_t,
                    int rows, int cols, int dtype, asr_stream_t stream);
/* column sums: out[n] += sum_m X[m*ld+n]  (bias gradients)                                                     */
int asr_colsum_acc(const void* X, int64_t ld, int M, int N, float* out, int dtype, asr_stream_t stream);

/* ---- residual + dropout + LayerNorm (+ positional add) (+ row mask) -----------------------------------------
 * z = dropout(y) + residual ; out = (LN(z)*gamma+beta + post_add[row % post_period]) * row_keep[row]
 * y is overwritten with z (saved for backward).  residual/post_add/row_keep may be NULL.
 * Replaces common_layers.py:140-141, :197-198 (sub-layer epilogues), transformer.py:172-173 (input LN + PE)
 * and the `*= non_pad_mask` at transformer.py:198,201,536,540,543.                                            */
int asr_add_ln_fwd(void* y_z, const void* residual, const float* gamma, const float* beta, const float* post_add,
                   int post_period, const uint8_t* row_keep, void* out, float* mean, float* rstd, int M, int D,
                   float eps, float dropout_p, uint64_t seed, const uint64_t* seed_dev, int dtype, asr_stream_t stream);
/* d_res = LN'(dout*row_keep) ; d_y = d_res * dropmask/(1-p) (d_y may alias d_res when p == 0, or be NULL);
 * dgamma_acc/dbeta_acc (fp32, D) are accumulated into.  workspace (fp32, >= asr_add_ln_bwd_workspace(M, D) elements,
 * optional): per-block column partials for a two-stage reduction; without it the kernel falls back to 32-row blocks
 * and atomics on the 2*D gradient addresses.                                                                     */
int64_t asr_add_ln_bwd_workspace(int M, int D);
int asr_add_ln_bwd(const void* dout, const void* z, const float* mean, const float* rstd, const float* gamma,
                   const uint8_t* row_keep, void* d_res, void* d_y, float* dgamma_acc, float* dbeta_acc,
                   float* workspace, int64_t workspace_floats, int M, int D, float dropout_p, uint64_t seed,
                   const uint64_t* seed_dev, int dtype, asr_stream_t stream);
/* The same with the second stage postponed: asr_add_ln_bwd_partials leaves the per-block column partials in `workspace`
 * (required, asr_add_ln_bwd_workspace(M, D) floats, alive until the reduction), asr_ln_reduce_multi adds the partials of n layers
 * (rows[i] = that layer's M) into their dgamma / dbeta in ONE launch -- the parameter gradients are not on backward's critical
 * path, 21 seven-microsecond launches per step are (host arrays of n pointers; the kernel arguments carry them by value).   */
int asr_add_ln_bwd_partials(const void* dout, const void* z, const float* mean, const float* rstd, const float* gamma,
                            const uint8_t* row_keep, void* d_res, void* d_y, float* workspace, int64_t workspace_floats, int M,
                            int D, float dropout_p, uint64_t seed, const uint64_t* seed_dev, int dtype, asr_stream_t stream);
int asr_ln_reduce_multi(const float* const* workspaces, const int* rows, float* const* dgamma, float* const* dbeta, int n, int D,
                        asr_stream_t stream);

/* ---- fused multi-head attention core: softmax(mask(Q K^T * scale)) (dropout) V -------------------------------
 * Q (B,Tq,H,d) with element strides (q_sb, q_st) and head h at offset h*d; same for K,V (B,Tk,H,d), O (B,Tq,H,d).
 * Masks (True = masked, as reference): key index >= key_len[b] (NULL = none); key_pad[b*mask_sb + q*mask_sq + k] != 0
 * (NULL = none; mask_sb=Tk, mask_sq=0 for a per-key mask, mask_sb=Tq*Tk, mask_sq=Tk for a full (B,Tq,Tk) mask);
 * causal: key > query.  lse (B,H,Tq) fp32 saved for backward.  attn_out (H*B,Tq,Tk) fp32 optional (index h*B+b,
 * reference layout common_layers.py:185-190).  o32 (optional, NULL = none): fp32 copy of O with O's element strides,
 * for the backward's delta = rowsum(dO*O): with bf16 storage the rounding error of O does not cancel against
 * dP = dO.V^T and dominates dQ/dK when the value rows share a common component (measured 3-10x on dK).
 * Replaces common_layers.py:211-225 and the permutes at :185-195. */
int asr_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* o32, float* lse, float* attn_out, int B, int H,
                 int Tq, int Tk, int d, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb,
                 int64_t v_st, int64_t o_sb, int64_t o_st, const int32_t* key_len, const uint8_t* key_pad,
                 int64_t mask_sb, int64_t mask_sq, int causal, float scale, float dropout_p, uint64_t seed,
                 const uint64_t* seed_dev, int dtype, asr_stream_t stream);
/* delta (B,H,Tq) fp32 workspace.  dQ/dK/dV use the strides of Q/K/V.  o32: the forward's fp32 copy or NULL.    */
int asr_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const float* o32, const void* dO, const float* lse,
                 float* delta, void* dQ, void* dK, void* dV, int B, int H, int Tq, int Tk, int d, int64_t q_sb,
                 int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st, int64_t o_sb, int64_t o_st,
                 const int32_t* key_len, const uint8_t* key_pad, int64_t mask_sb, int64_t mask_sq, int causal,
                 float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev, int parts, int dtype, asr_stream_t stream);

/* ---- decoder input side ---------------------------------------------------------------------------------------
 * Decoder.preprocess (transformer.py:254-266) + masks (:282-286): strip PAD(0) anywhere, seq_in = [SOS]+y padded
 * with EOS to Td, seq_out = y+[EOS] padded with PAD; key_pad = (seq_in==EOS), row_keep = (seq_in!=EOS).
 * *overflow is set to 1 if any row needs more than Td slots (pad_list would raise, common_layers.py:21).       */
int asr_decoder_preprocess(const int64_t* tgt, int B, int L, int Td, int64_t* seq_in, int64_t* seq_out,
                           uint8_t* key_pad, uint8_t* row_keep, int32_t* overflow, asr_stream_t stream);
/* out = dropout(table[tok]*scale + pe[t])   (transformer.py:292-293); table/pe fp32                            */
int asr_embed_fwd(const int64_t* tok, const float* table, const float* pe, void* out, int B, int T, int D,
                  float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev, int dtype, asr_stream_t stream);
int asr_embed_bwd(const int64_t* tok, const void* dout, float* dtable_acc, int B, int T, int D, float scale,
                  float dropout_p, uint64_t seed, const uint64_t* seed_dev, int pad_id, int dtype, asr_stream_t stream);

/* ---- fp8 (OCP e4m3fn) projections of the Low-Rank Transformer variant (BASELINE configs[4]).
 * asr_quant_fp8: x (M,K) bf16/fp32 -> q (M, ldq >= K rounded up to 16; pad bytes zero) with one scale PER ROW:
 *   scale[m] = amax of row m / 448 (the dequantisation factor the GEMM multiplies in).  One launch.
 * asr_gemm_nt_fp8: C[m,n] = scale_a[m] * scale_b[n] * sum_k A[m,k] B[n,k] (+ bias[n]) (ReLU) on the K = 128 block-scaled
 *   fp8 MFMA with unit block scales (fp32 accumulators; measured 3e-5 of the largest output off the exact product
 *   of the decoded operands: the matrix core aligns a block's products before adding).  K multiple of 16.        */
int asr_quant_fp8(const void* x, int64_t ld, int M, int K, int dtype, uint8_t* q, int64_t ldq, float* scale, asr_stream_t stream);
int asr_gemm_nt_fp8(const uint8_t* A, int64_t lda, const float* scale_a, const uint8_t* B, int64_t ldb, const float* scale_b,
                    void* C, int64_t ldc, const float* bias, int M, int N, int K, int relu, int out_dtype, asr_stream_t stream);

/* ---- CTC loss (utils/metrics.py:133-154: F.log_softmax over the vocabulary + F.ctc_loss(reduction="mean"), blank 0,
 * zero_infinity=False) and its gradient w.r.t. the LOGITS (log-softmax included).  logits (B,T,ld) fp32, targets
 * (B,Lmax) int64 (row b holds target_lengths[b] labels), lengths int32 on the device.  loss[0] = mean_b(nll_b /
 * max(len_b,1)); an unreachable target gives +inf (the reference skips such a batch, trainer.py:87-90).  The workspace
 * (asr_ctc_workspace floats) carries the lattice from asr_ctc_fwd to asr_ctc_bwd.  dlogits (B,T,ldo) fp32, columns
 * >= V zeroed; grad_out: device scalar d(loss).                                                                  */
int64_t asr_ctc_workspace(int B, int T, int Lmax);
int asr_ctc_fwd(const float* logits, int64_t ld, const int64_t* targets, const int32_t* input_lengths,
                const int32_t* target_lengths, int B, int T, int V, int Lmax, int blank, float* workspace,
                int64_t workspace_floats, float* loss, asr_stream_t stream);
int asr_ctc_bwd(const float* logits, int64_t ld, const int64_t* targets, const int32_t* input_lengths,
                const int32_t* target_lengths, int B, int T, int V, int Lmax, int blank, const float* workspace,
                const float* grad_out, float* dlogits, int64_t ldo, asr_stream_t stream);

/* ---- incremental (KV-cached) decoding with the position on the device: one captured hipGraph serves all 300 steps of
 * the reference's greedy loop (models/asr/transformer.py:316-394).  state[0] = position t of the token being fed.
 * asr_decode_prepare(advance=0): pe_cur[0..D) = pe[t], key_len[0..B) = t+1;  (advance=1): state[0] = t+1.
 * asr_kv_append: k_src / v_src (B, ncols) rows (row stride src_ld) -> k_cache / v_cache (B, max_len, ncols) at row t.     */
int asr_decode_prepare(const float* pe, int D, float* pe_cur, int32_t* key_len, int B, int64_t* state, int advance,
                       asr_stream_t stream);
int asr_kv_append(const void* k_src, const void* v_src, int64_t src_ld, void* k_cache, void* v_cache, int B, int ncols,
                  int max_len, const int64_t* state, int dtype, asr_stream_t stream);

/* ---- the same greedy step in 34 launches instead of 62 (bf16; B <= 32 rows; csrc/decode.hip).  Replaces, per decoder layer of
 * models/asr/transformer.py:533-545 run on ONE position: the Q/K/V, output, feed-forward projections (common_layers.py:170-200,
 * :135-142) and -- as a prologue of the GEMM that consumes it -- the sub-layer epilogue LayerNorm(dropout(y) + residual)
 * (common_layers.py:140-141, :197-198) or the embedding + positional encoding of transformer.py:292-293.
 * asr_dec_gemm: out (B, ldo) = act(x W^T + bias), W (N, ldw) bf16, K % 64 == 0.  prologue 0: x = X (B, ldx) bf16;
 *   1: x = LN(Y + R) * gamma + beta (Y, R (B, K) bf16 contiguous, K <= 512; z = Y + R rounded to bf16 before the statistics,
 *      like asr_add_ln_fwd), x also stored to x_out (B, K) when given;  2: x = table[tok[b]] * scale + pe[state[0]] (fp32
 *      table / pe), stored to x_out.  out_dtype ASR_BF16 or ASR_F32.  ASR_EUNSUPPORTED outside these shapes.
 *   Fragment-major order (`layout` flags; every wave-wide operand load of the kernel is then 1 KB of consecutive bytes): a
 *   (rows, K) matrix is cut into blocks of 32 rows x 16 columns; block (r, s) holds, for lane = 32 * h + i (h = 0, 1; i < 32),
 *   the 8 elements [32 r + i][16 s + 8 h .. + 7] at element offset ((r * K / 16 + s) * 64 + lane) * 8.  Weights: rows padded
 *   with zeros to a multiple of 32.  Activations (<= 32 rows): r = 0, rows >= B are never used.
 * asr_dec_attn: one query row per sequence and head (dk = 64): q (B, ldq); keys / values (B, rows, H*64) with the given
 *   batch / row strides.  state != NULL: self attention at position t = state[0]: k_new / v_new (B, ld_new) are this
 *   position's rows -- stored to row t of both caches by this launch -- and the keys are rows 0..t; state == NULL: all
 *   `rows` keys (cross attention, no mask: transformer.py:336-350 passes none at decode time).  out (B, ldo) bf16, or
 *   fragment-major (out_frag != 0, B <= 32).
 * asr_dec_finish: tok[b] = argmax of logits row b (lowest index on ties), done[b] |= tok == eos, out[t * B + b] = tok
 *   (t = state[0] < max_len), then state[0] = t + 1 (by the last workgroup; *ticket must be 0 before the first call).      */
#define ASR_DEC_W_FRAG 1    /* W in fragment-major order (below)                                   */
#define ASR_DEC_X_FRAG 2    /* prologue 0: X in fragment-major order (written by a producer with ..._OUT_FRAG) */
#define ASR_DEC_OUT_FRAG 4  /* bf16 out in fragment-major order, N % 16 == 0                        */
int asr_dec_gemm(const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo, int B, int N, int K, int relu,
                 int out_dtype, int layout, int prologue, const void* X, int64_t ldx, const void* Y, const void* R, const float* gamma,
                 const float* beta, float eps, void* x_out, const int64_t* tok, const float* table, const float* pe,
                 float scale, const int64_t* state, asr_stream_t stream);
int asr_dec_attn(const void* q, int64_t ldq, const void* k_new, const void* v_new, int64_t ld_new, void* k_cache,
                 void* v_cache, int64_t cache_batch_stride, int64_t cache_row_stride, int rows, void* out, int64_t ldo,
                 int B, int H, int dk, float scale, int out_frag, const int64_t* state, asr_stream_t stream);
/* asr_dec_attn_fused: asr_dec_gemm(prologue 1 or 2) + asr_dec_attn in ONE launch: per (sequence, head) the sub-layer input row
 *   x = LN(Y + R) gamma + beta (tok == NULL) or table[tok] emb_scale + pe[state[0]] (self attention of the first layer), the
 *   head's projections from W ((3 H 64, D) = [query | key | value] rows for self_attention != 0, (H 64, D) query rows otherwise;
 *   row-major bf16, bias fp32 or NULL), then the attention of asr_dec_attn (self: key / value row appended at t = state[0]).
 *   x (B, D) is stored to x_out by the head-0 workgroups.  D % 64 == 0, D <= 512, dk = 64, rows <= 512.                    */
int asr_dec_attn_fused(const void* W, const float* bias, int D, int self_attention, const void* Y, const void* R,
                       const float* gamma, const float* beta, float eps, const int64_t* tok, const float* table, const float* pe,
                       float emb_scale, void* x_out, void* k_cache, void* v_cache, int64_t cache_batch_stride,
                       int64_t cache_row_stride, int rows, void* out, int64_t ldo, int B, int H, int dk, float scale, int out_frag,
                       const int64_t* state, asr_stream_t stream);
int asr_dec_finish(const float* logits, int64_t ld, int V, int64_t* tok, uint8_t* done, int64_t* out, int B, int max_len,
                   int eos, int64_t* state, int32_t* ticket, asr_stream_t stream);

/* ---- label-smoothed cross entropy + argmax + num_correct (utils/metrics.py:78-132, transformer.py:80) ---------
 * logits (M, ld) fp32.  sums[0] += sum of row losses over non-PAD rows, sums[1] += #non-PAD rows,
 * sums[2] += #(argmax == gold) over non-PAD rows.  argmax = lowest index among maxima.                        */
int asr_ce_fwd(const float* logits, int64_t ld, const int64_t* gold, int M, int V, float smoothing, int pad_id,
               float* row_lse, int64_t* argmax, float* sums, asr_stream_t stream);
/* The same with the statistics reproducible run to run and no zeroed destination (round 6): asr_ce_fwd_partials leaves the three sums of
 * every block of rows in partials (asr_ce_partial_blocks(M) x 3 floats, every slot written), asr_ce_finish adds them in a fixed order into
 * sums[0..2] (overwritten) and writes loss[0] = sums[0] / (den ? den[0] : sums[1]) -- utils/metrics.py:127-130's mean, asr_ratio included.
 * (asr_ce_fwd adds with fp32 atomics: equal up to their order; it stays for destinations that are summed over ranks.)                  */
int asr_ce_partial_blocks(int M);
int asr_ce_fwd_partials(const float* logits, int64_t ld, const int64_t* gold, int M, int V, float smoothing, int pad_id,
                        float* row_lse, int64_t* argmax, float* partials, asr_stream_t stream);
int asr_ce_finish(const float* partials, int nblocks, const float* den, float* sums, float* loss, asr_stream_t stream);
/* out[m] = lowest index of the row maximum (torch.topk(pred,1) at transformer.py:80, metrics.py:89)            */
int asr_argmax_rows(const float* logits, int64_t ld, int M, int V, int64_t* out, asr_stream_t stream);
/* HOST function (no device work): Levenshtein distances of n sequence pairs of int32 symbols (UTF-32 code points for CER, word
 * ids for WER; utils/metrics.py:48-76, trainer.py:62-75).  Pair p = a[a_off[p] .. a_off[p+1]) vs b[b_off[p] .. b_off[p+1]).   */
int asr_edit_distance_batch(const int32_t* a, const int64_t* a_off, const int32_t* b, const int64_t* b_off, int n, int32_t* out);

/* beam-search scoring (F.log_softmax + torch.topk per hypothesis, transformer.py:446-449): vals (M,k) = the k largest
 * log-probabilities of each row, best first, idx (M,k) their indices (lowest index first among equal values); k <= 16   */
int asr_logsoftmax_topk(const float* logits, int64_t ld, int M, int V, int k, float* vals, int64_t* idx, asr_stream_t stream);
/* dlogits[m,v] = coef * (softmax*sum_q - q), coef = *grad_out / *count (device scalars), 0 for PAD rows.
 * dlogits has leading dimension ldd >= V and dtype `out_dtype`; columns V..ldd-1 are zero-filled.              */
int asr_ce_bwd(const float* logits, int64_t ld, const int64_t* gold, const float* row_lse, int M, int V,
               float smoothing, int pad_id, const float* grad_out, const float* count, void* dlogits, int64_t ldd,
               int out_dtype, asr_stream_t stream);

/* ---- optimiser: Adam(beta 0.9/0.98, eps 1e-9) under the Noam schedule (utils/optimizer.py:15-32,
 * utils/functions.py:107).  One launch over the flat fp32 parameter/gradient/moment buffers.
 * grad_scale_dev: optional device scalar multiplied into g (gradient clipping coefficient), may be NULL.        */
int asr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float bias_corr1, float bias_corr2, const float* grad_scale_dev, asr_stream_t stream);
/* Graph-replayable variant: state = device uint64[4] {dropout seed counter, optimiser step t, skipped flag, unused}; asr_step_advance()
 * increments the first two (one launch at the top of every step, captured or not) -- except that t stays when the previous
 * asr_adam_noam_step cancelled its update (state[2], cleared here): a skipped batch does not consume a step number, like the
 * reference's `continue` in front of opt.step().  asr_adam_noam_step reads t from state[1] and
 * computes lr = max(min_lr, factor_ms * min(t^-0.5, t * warmup^-1.5)) (factor_ms = k_lr * model_size^-0.5) and Adam's
 * bias corrections on the device; *lr_out (optional) receives lr.  Every dropout kernel mixes state[0] into its seed
 * through its `seed_dev` argument (NULL = host seed only).  guard_dev (optional device scalar, e.g. the step's loss sum): when it
 * or the gradient scale is not finite the launch changes NOTHING -- parameters and both moments keep their values; state[2] is set; *lr_out is still written -- which is
 * the reference trainer's `if loss == inf: continue` (trainer/asr/trainer.py:102-104) for a step that is replayed from a graph
 * and cannot branch on the host.  shadow_bf16 (optional, n elements): receives the updated parameters rounded to bf16 -- the
 * compute-dtype copy the next step's GEMMs read -- so that no separate cast pass over the masters is needed.            */
int asr_step_advance(uint64_t* state, asr_stream_t stream);
int asr_adam_noam_step(float* p, const float* g, float* m, float* v, int64_t n, uint64_t* state, float beta1,
                       float beta2, float eps, float factor_ms, float warmup, float min_lr,
                       const float* grad_scale_dev, float* lr_out, const float* guard_dev, void* shadow_bf16,
                       asr_stream_t stream);
/* row_keep[b*T + t] = t < lengths[b]: the encoder's non_pad_mask (common_layers.py:33-38 via transformer.py:168)     */
int asr_length_mask(const int32_t* lengths, int B, int T, uint8_t* row_keep, asr_stream_t stream);
/* out[0] = num[0] / den[0]: the mean over non-PAD tokens (utils/metrics.py:127-130) from asr_ce_fwd's sums            */
int asr_ratio(const float* num, const float* den, float* out, asr_stream_t stream);
/* acc[0] += sum(g^2)  (clip_grad_norm_, trainer.py:108-109)                                                     */
int asr_sumsq_acc(const float* g, int64_t n, float* acc, asr_stream_t stream);
/* coef[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6))                                                          */
int asr_clip_coef(const float* sumsq, float max_norm, float* coef, asr_stream_t stream);
/* Data-parallel form (replaces nn.DataParallel's gather + one loss over the whole batch, utils/functions.py:154-160 with
 * utils/metrics.py:127-130): every rank back-propagates its UN-normalised loss sum, `denom[0]` is the all-reduced
 * non-PAD token count; coef[0] = s * min(1, max_norm / (s * sqrt(sumsq[0]) + 1e-6)) with s = 1 / max(denom[0], 1).
 * sumsq == NULL: no clipping (coef = s).  denom == NULL: s = 1 (== asr_clip_coef).                                   */
int asr_grad_coef(const float* sumsq, float max_norm, const float* denom, float* coef, asr_stream_t stream);

/* ---- vgg_cnn front end (transformer.py:42-53, :70-76) -----------------------------------------------------------
 * Activations are NHWC: (B, H=F, W=T, C), C contiguous, in `dtype`.                                             */
/* conv.0: 1->C0 3x3 pad 1 + ReLU from the raw fp32 spectrogram (B,1,F,T); weight (C0,1,3,3), bias fp32          */
int asr_conv1_fwd(const float* x, const float* w, const float* bias, void* y, int B, int H, int W, int C0,
                  int dtype, asr_stream_t stream);
/* dw (C0*9) += , db (C0) += ; dy already masked by ReLU.  bf16, C0 = 64: MFMA kernel with the pixel as contraction index
 * (x split into two bf16, conv1_wgrad_mfma.hip); otherwise the direct vector-ALU kernel.                            */
int asr_conv1_wgrad(const float* x, const void* dy, float* dw_acc, float* db_acc, int B, int H, int W, int C0,
                    int dtype, asr_stream_t stream);
/* master (Cout,Cin,3,3) fp32 -> wk (Cout,9,Cin) for forward and wd (Cin,9,Cout) tap-flipped for dgrad            */
int asr_conv_pack_weight(const float* w, void* wk, void* wd, int Cout, int Cin, int dtype, asr_stream_t stream);
/* the same for up to 8 weight tensors in ONE launch (arrays of n pointers / sizes in host memory): the conv stack's packs of a step */
int asr_conv_pack_weight_multi(int n, const float* const* w, void* const* wk, void* const* wd, const int* Cout, const int* Cin,
                               int dtype, asr_stream_t stream);
/* y = act(conv3x3_pad1(x; wk) + bias): relu=1 -> ReLU.  If mask_src != NULL: y *= (mask_src > 0) (dgrad through
 * the ReLU that produced this conv's input).  Cin, Cout multiples of 64, Cout <= 128.  bf16 64 -> 64 runs the persistent
 * register-resident-weights kernel of conv_c64.hip, bf16 with 128 input channels the weight-stationary kernel of
 * conv_ws.hip (round 5), everything else the generic implicit GEMM of conv.hip.                                    */
int asr_conv3x3_igemm(const void* x, const void* wk, const float* bias, const void* mask_src, void* y, int B,
                      int H, int W, int Cin, int Cout, int relu, int dtype, asr_stream_t stream);
/* ReLU masks of ONE BIT per element (round 5; transformer.py:48-52 and the autograd of its second ReLU: conv.5 -> ReLU -> conv.7).
 * A (B, H, W, 128) mask is an array of dwords [b][h / 4][w / 16][c / 32][lane], rows padded to a multiple of 8 and columns to a multiple
 * of 16 (asr_relu_bits_bytes = its size; -1 unless C = 128): byte r of the dword = row 4 (h / 4) + r, bit k = channel
 * 32 (c / 32) + 16 (g & 1) + 8 (g >> 1) + k of pixel column 16 (w / 16) + 8 (a & 1) + 2 (l & 3) + ((a >> 1 ^ a) & 1), where
 * l = lane & 15, a = l >> 2, g = lane >> 4 -- the kernels' own fragment order, so that the writer's store and the reader's loads are
 * 256 contiguous bytes per wave.  asr_conv3x3_igemm_bits is asr_conv3x3_igemm with exactly one of
 *   bits_out: also writes the mask (y > 0) of this launch's ReLU output          (bf16, Cin = 64, Cout = 128, relu = 1);
 *   bits_in : y *= bit, instead of asr_conv3x3_igemm's 16-bit mask_src tensor      (bf16, Cin = Cout = 128);
 * ASR_EUNSUPPORTED for any other shape / dtype (callers keep the mask_src form).  Pointers 4-byte aligned.  Results are bit for bit
 * those of asr_conv3x3_igemm with the bf16 mask.                                                                  */
int64_t asr_relu_bits_bytes(int B, int H, int W, int C);
int asr_conv3x3_igemm_bits(const void* x, const void* wk, const float* bias, const uint8_t* bits_in, void* y, uint8_t* bits_out,
                           int B, int H, int W, int Cin, int Cout, int relu, int dtype, asr_stream_t stream);
/* y = ReLU(conv3x3_pad1(x; wk) + bias) AND pool = 2x2/2 floor max-pool of y (B, H/2, W/2, Cout) from the same epilogue
 * (transformer.py:45-47: conv.2, ReLU, MaxPool2d): the pool no longer re-reads y.  ASR_EUNSUPPORTED unless bf16 and
 * Cin = Cout = 64 (the layer that has this shape in the model) -- callers then use asr_conv3x3_igemm + asr_maxpool_fwd.  */
int asr_conv3x3_relu_pool(const void* x, const void* wk, const float* bias, void* y, void* pool, int B, int H, int W,
                          int Cin, int Cout, int dtype, asr_stream_t stream);
/* 2x2/2 floor max-pool NHWC; if out_tcf != 0 writes (B, W/2, C, H/2) i.e. the encoder layout (B,T',C*F')        */
int asr_maxpool_fwd(const void* x, void* y, int B, int H, int W, int C, int out_tcf, int dtype, asr_stream_t stream);
/* dx = scatter of dy to the first maximum of each window, times (x > 0); dy layout per in_tcf                   */
int asr_maxpool_bwd(const void* x, const void* dy, void* dx, int B, int H, int W, int C, int in_tcf, int dtype,
                    asr_stream_t stream);
/* The same pooling (nn.MaxPool2d(2, stride=2) after a ReLU, transformer.py:46,52) with a SELECTION CODE instead of the pre-pool
 * activations in backward: the forward writes one byte per pooled element, in the pooled tensor's layout -- 0: maximum <= 0 (no
 * gradient: ReLU'), 1 + k: the first maximum is window position k in scan order -- and the backward reads dy and the codes only
 * (first pool of the VGG front end: 0.72 GB moved instead of 1.19 GB, and the 527 MB un-pooled activation need not be kept).
 * ASR_EUNSUPPORTED for layouts without whole 16-byte chunks (callers use the pair above).                          */
int asr_maxpool_fwd_code(const void* x, void* y, uint8_t* code, int B, int H, int W, int C, int out_tcf, int dtype, asr_stream_t stream);
int asr_maxpool_bwd_code(const uint8_t* code, const void* dy, void* dx, int B, int H, int W, int C, int in_tcf, int dtype,
                         asr_stream_t stream);
/* asr_conv3x3_relu_pool that writes the pooled output and its selection codes; y_or_null = NULL: the un-pooled output is not stored */
int asr_conv3x3_relu_pool_code(const void* x, const void* wk, const float* bias, void* y_or_null, void* pool, uint8_t* code, int B,
                               int H, int W, int Cin, int Cout, int dtype, asr_stream_t stream);
/* ---- the full-resolution level of vgg_cnn without its full-resolution activations (reference: models/asr/transformer.py:42-47 and
 * its autograd; csrc/conv_level0.hip).  bf16 activations, 64 channels.  src (B, H, W) fp32 = the front end's single input channel.
 * forward: pool (B, H/2, W/2, 64) = MaxPool2d(2,2)(ReLU(conv.2(ReLU(conv.0(src))))) and one selection byte per pooled element
 * (layout of pool: 0 = the maximum is 0, 1 + k = first maximum at window position k); conv.0's 64-channel output exists only as
 * halo patches in LDS.  w0 (64,1,3,3) / b0 fp32 masters, wk2 = conv.2's packed forward weights (asr_conv_pack_weight), b2 fp32.
 * Replaces asr_conv1_fwd + asr_conv3x3_relu_pool_code. */
int asr_vgg_level0_fwd(const float* src, const float* w0, const float* b0, const void* wk2, const float* b2, void* pool,
                       uint8_t* code, int B, int H, int W, asr_stream_t stream);
/* floats of workspace the two backward entry points need (the larger of the two) */
int64_t asr_vgg_level0_bwd_workspace(int B, int H, int W);
/* backward, data side: dpool (gradient of pool) is expanded through `code` inside the kernel, contracted with conv.2's tap-flipped
 * weights wd2 (asr_conv_pack_weight), masked with conv.0's ReLU mask RECOMPUTED from src, and reduced against the frames:
 * dw0 (64,1,3,3) += , db0 (64) += .  The gradient of conv.0's output is never stored (the input frames need none).
 * Replaces asr_maxpool_bwd_code + asr_conv3x3_igemm(mask) + asr_conv1_wgrad. */
int asr_vgg_level0_dgrad(const void* dpool, const uint8_t* code, const float* src, const float* w0, const float* b0, const void* wd2,
                         float* dw0, float* db0, float* workspace, int64_t workspace_floats, int B, int H, int W, asr_stream_t stream);
/* backward, weight side of conv.2: dw2 (64,64,3,3) += , db2 (64) += from ReLU(conv.0(src)) recomputed on halo patches and the
 * expanded dpool.  Replaces asr_maxpool_bwd_code + asr_conv3x3_wgrad_nhwc for this layer. */
int asr_vgg_level0_wgrad(const float* src, const float* w0, const float* b0, const void* dpool, const uint8_t* code, float* dw2,
                         float* db2, float* workspace, int64_t workspace_floats, int B, int H, int W, asr_stream_t stream);
/* conv.7 + ReLU + MaxPool2d + the (B, T', C F') view / transpose of transformer.py:50-52,74-76 from one epilogue: pool (B, W/2, Cout, H/2)
 * and its selection bytes (same layout); the un-pooled output is never stored.  bf16, Cout = 128, W a multiple of 16, H a multiple of
 * 8 with 128 input channels (the weight-stationary kernel of csrc/conv_ws.hip, round 5), of 16 otherwise.                          */
int asr_conv3x3_relu_pool_tcf_code(const void* x, const void* wk, const float* bias, void* pool, uint8_t* code, int B, int H, int W,
                                   int Cin, int Cout, int dtype, asr_stream_t stream);
/* The same with the selection bytes CHANNEL LAST, code_cl (B, W/2, H/2, Cout): the 128 bytes of a pooled pixel contiguous -- the form
 * asr_gemm_nn_poolbwd reads (round 6).  Served by the weight-stationary kernel only (bf16, 128 -> 128, H % 8 == 0, W % 16 == 0),
 * else ASR_EUNSUPPORTED (callers keep asr_conv3x3_relu_pool_tcf_code + asr_maxpool_bwd_code).                                       */
int asr_conv3x3_relu_pool_tcf_codecl(const void* x, const void* wk, const float* bias, void* pool, uint8_t* code_cl, int B, int H, int W,
                                     int Cin, int Cout, int dtype, asr_stream_t stream);
/* dW (Cout,Cin,3,3) += and db (Cout, optional) += straight from NHWC x (B,H,W,Cin) and dy (B,H,W,Cout): the transposed
 * MFMA operands are built in LDS with ds_read_b64_tr_b16, no planar copies (conv.hip; bf16 with a workspace: the LDS-DMA
 * pipelined kernel of conv_wgrad_dma.hip).                                                                      */
/* workspace (fp32, >= asr_conv3x3_wgrad_workspace(...) elements, optional): per-workgroup partial dW blocks for a two-stage
 * reduction; without it every workgroup adds its 36,864 partial sums with fp32 atomics (more than half of the kernel time) */
int64_t asr_conv3x3_wgrad_workspace(int B, int H, int W, int Cin, int Cout);
int asr_conv3x3_wgrad_nhwc(const void* x, const void* dy, float* dw_acc, float* db_acc, float* workspace,
                           int64_t workspace_floats, int B, int H, int W, int Cin, int Cout, int dtype, asr_stream_t stream);
/* The two stages separately: asr_conv3x3_wgrad_partials leaves the per-workgroup partial dW blocks in `workspace` (required; db is
 * accumulated as usual), asr_conv3x3_wgrad_reduce adds them into dw -- an HBM-bound pass that a caller with a second stream runs
 * next to the MFMA-bound data-gradient convolution that follows (asr_hip/functions.py: VGGFn.backward).                    */
int asr_conv3x3_wgrad_partials(const void* x, const void* dy, float* db, float* workspace, int64_t workspace_floats, int B, int H,
                               int W, int Cin, int Cout, int dtype, asr_stream_t stream);
int asr_conv3x3_wgrad_reduce(const float* workspace, float* dw, int B, int H, int W, int Cin, int Cout, asr_stream_t stream);

/* ---- emb_cnn front end (reference: models/asr/transformer.py:33-40: Conv2d(1,32,(41,11),(2,2),(0,10)) / BatchNorm2d /
 * Hardtanh(0,20) / Conv2d(32,32,(21,11),(2,1)) / BatchNorm2d / Hardtanh(0,20)), embcnn.hip.  The strided big-window
 * convolutions run as GEMMs (asr_gemm_nt / _tn / _nn) on im2col rows m = (b,oh,ow), columns k = (ky,kx,c).            */
/* col (rows_alloc, ld_col) <- patches of NHWC x (B,H,W,C); columns >= KH*KW*C and rows >= B*OH*OW are written as 0.  */
int asr_im2col(const void* x, void* col, int B, int H, int W, int C, int KH, int KW, int SH, int SW, int PH, int PW,
               int OH, int OW, int64_t ld_col, int64_t rows_alloc, int in_dtype, int out_dtype, asr_stream_t stream);
/* dx (B,H,W,C) <- sum of the dcol entries that cover each input pixel (the conv data gradient; gather, no atomics)  */
int asr_col2im(const void* dcol, void* dx, int B, int H, int W, int C, int KH, int KW, int SH, int SW, int PH, int PW,
               int OH, int OW, int64_t ld_col, int dtype, asr_stream_t stream);
/* The unit-time-stride big-window convolution (transformer.py:37, Conv2d(32,32,(21,11),(2,1))) after ONE dense product over the
 * single-time-step patches, Z (groups*Wg + KW - 1, ldz >= KW*Cout) fp32 with Z[r, kx*Cout + co] = patch(r) . W[co, kx]:
 * y[g*OW + j, co] = bias[co] + sum_kx Z[g*Wg + j + kx, kx*Cout + co], j < OW; columns Cout .. ldy-1 of y are written as 0.  */
int asr_window_sum(const float* Z, int64_t ldz, float* y, int64_t ldy, const float* bias, int64_t groups, int Wg, int OW,
                   int KW, int Cout, asr_stream_t stream);
/* nn.BatchNorm2d statistics over the rows of the fp32 conv output y (M, ldy), channel = column:
 * sums[0:C] += sum(y - center), sums[C:2C] += sum((y - center)^2); center may be NULL (two-pass mean / variance).     */
/* y_grid_w / y_grid_ow (all asr_bn_* calls): y_grid_ow > 0 says that y is the un-compacted output of a window GEMM (section emb_cnn
 * above): rows come in groups of y_grid_w of which the first y_grid_ow are convolution outputs, row m of the convolution is y row
 * (m / y_grid_ow) * y_grid_w + m % y_grid_ow (M < 2^31, M % y_grid_ow == 0); 0 / 0: y is compact.  asr_bn_act_bwd's dy_grid_* place
 * its OUTPUT rows the same way (the dense dy operand of the window data / weight gradients; the rows between groups are not written). */
int asr_bn_stats(const float* y, int64_t ldy, int64_t M, int C, const float* center, float* sums, int y_grid_w, int y_grid_ow,
                 asr_stream_t stream);
/* the same sums WITHOUT atomics: per-workgroup sums into the workspace partial (asr_bn_stats_blocks(M), 2C), then added in a
 * fixed order into sums (2C, overwritten): the batch statistics, and with them the forward pass, are reproducible bit for bit  */
int64_t asr_bn_stats_blocks(int64_t M);
int asr_bn_stats_partial(const float* y, int64_t ldy, int64_t M, int C, const float* center, float* partial, float* sums,
                         int y_grid_w, int y_grid_ow, asr_stream_t stream);
/* nn.BatchNorm2d's training-mode statistics in one call (four launches): mean, then the centred second moment, both summed in a
 * fixed order (partial: asr_bn_stats_blocks(M) x 2C floats), rstd = rsqrt(var + eps) and -- momentum >= 0 -- the module's buffers:
 * running_mean = (1 - momentum) running_mean + momentum mean, running_var likewise with the UNBIASED variance var M / (M - 1),
 * num_batches_tracked += 1 (transformer.py:35,38 = torch.nn.BatchNorm2d defaults eps 1e-5, momentum 0.1).                       */
int asr_bn_batch_stats(const float* y, int64_t ldy, int64_t M, int C, float* partial, float* mean, float* rstd, float eps,
                       float momentum, float* running_mean, float* running_var, int64_t* num_batches, int y_grid_w, int y_grid_ow,
                       asr_stream_t stream);
/* Length-masked forms (round 6).  Rows are (group, t), t = m % row_w the time step of the convolution output; only rows with
 * t < valid_w[0] (a DEVICE int, so that one captured graph serves every batch of its shape bucket) take part in the statistics
 * (count = (M / row_w) * valid_w[0]) and receive a gradient from asr_bn_act_bwd_v (the others get 0).  What is masked is the padding a
 * shape bucket adds BEHIND the batch as collated (trainer --graph-buckets): the reference's BatchNorm (transformer.py:35,38) runs over
 * the collate padding but never sees a bucket.  valid_w == NULL: the unmasked functions above.                                    */
int asr_bn_batch_stats_v(const float* y, int64_t ldy, int64_t M, int C, float* partial, float* mean, float* rstd, float eps,
                         float momentum, float* running_mean, float* running_var, int64_t* num_batches, int y_grid_w, int y_grid_ow,
                         const int* valid_w, int row_w, asr_stream_t stream);
/* out = clamp(gamma * (y - mean) * rstd + beta, lo, hi)   (BatchNorm2d + Hardtanh, transformer.py:35-36,38-39).
 * tH > 0: rows are (b,h,w) over (B,tH,tW) and out is the encoder input (B, tW, C*tH), feature c*tH + h (:74-76).     */
int asr_bn_act_fwd(const float* y, int64_t ldy, void* out, int64_t ldo, int64_t M, int C, const float* mean,
                   const float* rstd, const float* gamma, const float* beta, float lo, float hi, int tH, int tW,
                   int y_grid_w, int y_grid_ow, int dtype, asr_stream_t stream);
/* sums[0:C] += sum dz, sums[C:2C] += sum dz*xhat, dz = dout where lo < z < hi (= dbeta, dgamma of the BatchNorm)      */
int asr_bn_act_bwd_reduce(const void* dout, int64_t ldo, const float* y, int64_t ldy, int64_t M, int C, const float* mean,
                          const float* rstd, const float* gamma, const float* beta, float lo, float hi, int tH, int tW,
                          int y_grid_w, int y_grid_ow, float* sums, int dtype, asr_stream_t stream);
int asr_bn_act_bwd_reduce_v(const void* dout, int64_t ldo, const float* y, int64_t ldy, int64_t M, int C, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, float lo, float hi, int tH, int tW,
                            int y_grid_w, int y_grid_ow, float* sums, const int* valid_w, int row_w, int dtype, asr_stream_t stream);
/* dy (M, lddy) = gamma * rstd * (dz - sums[c]/M - xhat * sums[C+c]/M)   (training-mode BatchNorm backward)           */
int asr_bn_act_bwd(const void* dout, int64_t ldo, const float* y, int64_t ldy, void* dy, int64_t lddy, int64_t M, int C,
                   const float* mean, const float* rstd, const float* gamma, const float* beta, float lo, float hi,
                   int tH, int tW, int y_grid_w, int y_grid_ow, int dy_grid_w, int dy_grid_ow, const float* sums, int dtype,
                   asr_stream_t stream);
int asr_bn_act_bwd_v(const void* dout, int64_t ldo, const float* y, int64_t ldy, void* dy, int64_t lddy, int64_t M, int C,
                     const float* mean, const float* rstd, const float* gamma, const float* beta, float lo, float hi, int tH, int tW,
                     int y_grid_w, int y_grid_ow, int dy_grid_w, int dy_grid_ow, const float* sums, const int* valid_w, int row_w,
                     int dtype, asr_stream_t stream);

/* ---- spectrogram front end on the device (reference: SpectrogramParser.parse_audio, utils/data_loader.py:72-89) ---------
 * frames (B*Tmax, n_fft) fp32 <- windowed, centred (reflect padded) frames of the padded waveforms wav (B, wav_stride),
 * lengths (B) samples; frames past 1 + len/hop of an utterance are zero.  The DFT itself is asr_gemm_nt (fp32) against
 * the [cos | -sin] basis (2*(n_fft/2+1), n_fft).                                                                     */
int asr_stft_frames(const float* wav, int64_t wav_stride, const int32_t* lengths, const float* window, float* frames, int B,
                    int Tmax, int n_fft, int hop, asr_stream_t stream);
/* reim (B*Tmax, ld): [re(F) | im(F)] per frame -> spect (B, F, Tmax) = log1p(|.|), zero past each utterance's frames;
 * normalize != 0: (x - mean) / std per utterance (unbiased std).  sums / sqdev: zero-initialised fp32 (B) scratch.     */
int asr_spect_finish(const float* reim, int64_t ld, const int32_t* lengths, float* spect, float* sums, float* sqdev, int B,
                     int F, int Tmax, int hop, int normalize, asr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ASR_HIP_H_ */
