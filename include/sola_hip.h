/*
 * sola_hip.h — C ABI of libsola_hip.so: the MI355X (gfx950) implementation of SOLA's track-selection hot path.
 *
 * The reference (cvlab-kaist/SOLA) has no FFI layer: its boundary is the Python object interface used by
 * train.py / inference.py / evaluator.py.  The entry points below are what a binding for that
 * path binds; each cites the reference interface it replaces (paths relative to the reference root).
 * sola_amd/module.py, sola_amd/loss.py and sola_amd/seg_utils.py are the ctypes bindings that present the
 * reference's own Python interface on top of this ABI (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C types only; every pointer named *dev* is a device (HBM) pointer owned by the caller;
 *   - every function returns 0 on success or a negative SolaStatus; sola_last_error() gives the message of the
 *     last failure on the calling thread; nothing throws across the ABI;
 *   - all work is enqueued asynchronously on the given hipStream_t (passed as void*; NULL = default stream);
 *   - a context is re-entrant across contexts but not thread-safe on one context;
 *   - fp32 row-major tensors; activations are channels-last: object tokens [B,N,T,d], text tokens [B,L,D].
 */
#ifndef SOLA_HIP_H
#define SOLA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum SolaStatus {
    SOLA_OK = 0,
    SOLA_ERR_ARG = -1,       /* invalid argument / unsupported shape */
    SOLA_ERR_HIP = -2,       /* a HIP runtime call failed */
    SOLA_ERR_WEIGHT = -3,    /* unknown weight name, wrong size, or weights missing at forward time */
    SOLA_ERR_WORKSPACE = -4, /* workspace too small */
    SOLA_ERR_STATE = -5      /* call sequence error (e.g. backward without a saved forward) */
} SolaStatus;

/* configs["model"] of the reference (configs/mevis/default.yaml:3-13; consumed at module/module.py:59-63,76,19).
 * num_heads is hard-coded to 8 in the reference (module/module.py:13-15). */
typedef struct SolaConfig {
    int32_t object_token_dim;
    int32_t lang_token_dim;
    int32_t n_layers;
    int32_t max_temporal_length;
    int32_t n_negative;
    int32_t n_groups;        /* encoder GroupNorm groups */
    int32_t n_groups_module; /* alignment-layer GroupNorm groups */
    int32_t num_heads;
} SolaConfig;

typedef struct SolaCtx SolaCtx;

const char* sola_last_error(void);
const char* sola_version(void);

/* ---- context: replaces LanguageAlignedTrackSelectionModule.__init__ / .to(device) (module/module.py:55-110) ---- */
int sola_ctx_create(const SolaConfig* cfg, int device, SolaCtx** out);
int sola_ctx_destroy(SolaCtx* ctx);

/* Number of state_dict tensors and their names/sizes, in the reference's state_dict order
 * (84 entries for the default config: 83 parameters + the Fourier buffer: module/module.py:74-110, tools/attention.py:26-29). */
int sola_num_weights(const SolaCtx* ctx);
int sola_weight_info(const SolaCtx* ctx, int index, const char** name, int64_t* numel);

/* Borrow a device pointer for one state_dict tensor (fp32, contiguous).  Replaces load_state_dict / parameter
 * access (inference.py:33, train.py:46).  The pointer must stay valid until replaced or the ctx is destroyed. */
int sola_set_weight(SolaCtx* ctx, const char* name, const void* dev_ptr, int64_t numel);

/* Tell the context that weight VALUES changed in place (optimizer step): the standardised conv weights
 * (module/ws.py:9-13) are recomputed at the next forward.  recompute_every_forward=1 mirrors the reference,
 * which re-standardises on every call. */
int sola_weights_changed(SolaCtx* ctx);
int sola_set_ws_policy(SolaCtx* ctx, int recompute_every_forward);

/* Inference arithmetic of the dense contractions (convs and projections, 98 % of the FLOPs):
 *   0  exact f32: v_mfma_f32_32x32x2_f32 on f32 operands (default);
 *   1  split-f16: every operand value x is carried as (f16 hi, f16 lo) with hi + lo = x to 22 bits, in the same 4 bytes,
 *      and each product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with f32 accumulation - 3/16 of the f32
 *      matrix-pipe time (gfx950 has no xf32/TF32).  Softmax, GroupNorm statistics, score head and losses stay f32.
 *   2  16-bit activation STORAGE (sola_forward only; BASELINE configs C2 / C4 name bf16 / fp16 runs of this path): every
 *      activation between two kernels is a plain f16 (2 bytes per element), every product of the convs / projections /
 *      attention ONE f16 MFMA with f32 accumulation; softmax, GroupNorm statistics, biases, score head, losses stay f32.
 *      f16 rather than bf16: 11 significant bits for the same bytes, its range covered by the scales + guard below.  A
 *      reduced-precision mode with a heavy-tailed error (logits of magnitude ~10: 0.8 % rms, 99.5 % within 0.25, worst ~0.7;
 *      tests/test_gpu_f16.py states the bound), never the default; needs object_token_dim and lang_token_dim %% 64 == 0.
 *      sola_forward and sola_forward_ragged (round 3) both run it; a tripped range guard repeats the call in exact f32.
 * In training (sola_forward_train / sola_backward) precision 1 runs every GEMM of the step (forward, dX and dW of the
 * projections and convs) on split-f16 casts of the f32 activations / gradients, from 1024 token rows on; attention and
 * GroupNorm backward and everything saved for the backward stay f32.  Since round 3 the weight-gradient products dW = dY^T X
 * take PLAIN f16 casts (one MFMA per product; sola_tune "train_dw_f16" 0 = split pairs there too): a sum over >= 1024 token
 * rows averages the operand rounding out - median per-matrix error 1e-4 against the exact-f32 step, worst tensor and gradient
 * cosine unchanged (tests/test_gpu_backward.py::test_split_training_weight_gradient_products_on_f16_operands).  Precision 2 in training is MIXED precision (BASELINE
 * config C2): the same step with plain-f16 casts and ONE f16 MFMA per product (f32 accumulation, per-tensor power-of-two
 * scales); activations, statistics, softmax, saved tensors and master weights stay f32.  Reduced precision with a stated
 * tolerance: losses within 1 %, gradient cosine >= 0.95 against the exact-f32 step (tests/test_gpu_backward.py).
 *   3  TRAINING only (sola_forward_train[_ragged] / sola_backward[_ragged]; the inference entry points return SOLA_ERR_ARG):
 *      the mixed-precision step of precision 2 with BFLOAT16 GEMM operands (v_cvt_pk_bf16_f32 casts, v_mfma_f32_32x32x16_bf16,
 *      f32 accumulation) - the "bf16 training" BASELINE config C2 names.  8 significant bits instead of 11 for the same MFMA
 *      rate; the tolerance is stated in tests/test_gpu_backward.py next to the f16 one. */
int sola_set_precision(SolaCtx* ctx, int precision);
/* Range handling of precision 1 in sola_forward / sola_forward_ragged.  The split-f16 pairs keep 22 significant bits for
 * every value within 2^-16 of its tensor's largest, on top of a per-tensor power-of-two scale:
 *   - the caller's object tokens and text tokens and every projection weight matrix get a data-dependent scale found on the
 *     device (largest magnitude -> [2^13, 2^14)), undone in the consuming GEMM's epilogue: any input scale, weight outliers;
 *   - the activations between the stages use a fixed scale; every kernel that writes one checks the value against the f16
 *     range (non-finite or |v| >= 65000 sets a guard word), and when the weights change the rms each GroupNorm will emit,
 *     sqrt(mean(gamma^2 + beta^2)), is checked against [2^-6, 2^9].
 * With the guard enabled (default) the forward reads the guard words (a 4-byte copy and ONE stream synchronisation per
 * call) and, when one is set, repeats the call on the exact-f32 kernels in the same workspace, so the caller never sees a
 * result computed outside the format's range.  enable = 0 keeps the call fully asynchronous (also while the stream is
 * being captured into a graph, where the check is skipped automatically); the words can then be read with
 * sola_split_fallback_count: *count = calls repeated in f32 so far, *last_guard = guard bits of the last checked call
 * (bit 0: a value left the f16 range, bit 1: GroupNorm weights outside the covered magnitude).  Once bit 1 has been seen for the
 * current weights, further calls skip the split pass and run the exact-f32 kernels directly (one line on stderr says so) until a
 * weight changes.  The guard words, their pinned host copy and the precision switch of the repeat belong to the CONTEXT: a
 * context serves one stream / one host thread at a time (use one context per stream for concurrent calls). */
int sola_set_split_guard(SolaCtx* ctx, int enable);

/* Operand-cast arena of the reduced-precision TRAINING modes (no reference counterpart).  In precisions 1 / 2 / 3 the training forward
 * keeps the 16-bit operand casts of its GEMM inputs so that the backward's weight-gradient products read them instead of casting the
 * same activations again (sola_tune "train_x16_keep").  The memory is the CALLER's: `dev_ptr` (256-byte aligned, `bytes` long) is
 * borrowed until it is replaced (null = keep nothing; the backward then casts as before - results are bit-identical either way).
 * Replace it only between a backward and the next forward.  sola_x16_arena_info reports what the LAST training forward asked for in
 * total (`need_bytes`, whether it fitted or not), the capacity and the bytes in use, so a caller can grow its buffer between steps
 * (sola_amd/module.py keeps a torch tensor 1/8 above the largest need seen).  Round 4; before, the ctx allocated this arena itself. */
int sola_set_x16_arena(SolaCtx* ctx, void* dev_ptr, size_t bytes);
int sola_x16_arena_info(const SolaCtx* ctx, size_t* need_bytes, size_t* capacity_bytes, size_t* used_bytes);
int sola_split_fallback_count(const SolaCtx* ctx, int64_t* count, int32_t* last_guard);
/* f32 rows -> split-f16 rows (same bytes per element; K % 8 == 0); scale must be a power of two */
int sola_cast_sp16(const float* dev_in, int ld_in, float* dev_out, int ld_out, int64_t rows, int K, float scale, void* stream);
/* The same conversion with a data-dependent power-of-two scale, for operands whose magnitude the host does not know
 * (gradients: mostly below the f16 normal range).  dev_scal = 2 floats: [0] receives max|in| (float bits), [1] the
 * inverse of the scale applied (max|in| lands in [2^13, 2^14)); pass dev_scal + 1 as dev_out_scale below. */
int sola_cast_sp16_auto(const float* dev_in, int ld_in, float* dev_out, int ld_out, int64_t rows, int K, float* dev_scal, void* stream);
/* C = out_scale * (A W^T) + bias (+ R) with A [M,K], W [N,K] (and optionally R) in the split-f16 format; C is written
 * as f32, or as split-f16 pairs when c_is_split (N % 8 == 0) */
int sola_gemm_nt_split(const float* dev_a_sp, int lda, const float* dev_w_sp, const float* dev_bias, const float* dev_r,
                       int ldr, int r_is_split, float* dev_c, int ldc, int c_is_split, int M, int N, int K, float out_scale,
                       void* stream);
/* ... with a further result multiplier read from device memory (NULL = none) */
int sola_gemm_nt_split_scaled(const float* dev_a_sp, int lda, const float* dev_w_sp, const float* dev_bias, const float* dev_r,
                              int ldr, int r_is_split, float* dev_c, int ldc, int c_is_split, int M, int N, int K,
                              float out_scale, const float* dev_out_scale, void* stream);

/* Building blocks of precision 2 (16-bit activation storage), exposed for the parity tests: f32 rows -> _Float16 rows (fixed
 * power-of-two scale, or data-dependent with dev_scal as for sola_cast_sp16_auto); C = out_scale * (A W^T) + bias (+ R) on
 * _Float16 A [M,K] / W [N,K] / R with f32 accumulation, C written as _Float16 or f32 (K %% 64 == 0, lda %% 16 == 0; pitches in
 * elements); softmax(q k^T scale) v on _Float16 q / k / v / o with the addressing of sola_attention. */
int sola_cast_f16(const float* dev_in, int ld_in, void* dev_out, int ld_out, int64_t rows, int K, float scale, float* dev_scal, void* stream);
int sola_gemm_nt_f16(const void* dev_a_h, int lda, const void* dev_w_h, const float* dev_bias, const void* dev_r_h, int ldr,
                     void* dev_c, int ldc, int c_is_f16, int M, int N, int K, float out_scale, void* stream);
int sola_attention_f16(const void* dev_q, int ldq, const void* dev_k, int ldk, const void* dev_v, int ldv, void* dev_o, int ldo,
                       int G, int H, int head_dim, int Sq, int Sk, int inner, int64_t q_outer, int64_t q_inner, int64_t q_row_stride,
                       int64_t k_outer, int64_t k_inner, int64_t k_row_stride, float scale, void* stream);

/* Building blocks of the bf16 training step's 16-bit storage (precision 3, round 6), exposed for the parity tests.  BFLOAT16 matrices are
 * rows of 2-byte values, pitches in values.
 *   sola_gemm_nt_bf16: C = A W^T + bias (+ R) on bfloat16 A [M,K] / W [N,K], f32 accumulation; C and R f32 or (c_is_bf16 / r_is_bf16) bfloat16.
 *   sola_attention_bf16: softmax(q k^T scale) v on bfloat16 q / k / v (sola_attention's addressing, more than 16 queries or keys); the
 *     output as f32 rows (dev_o) and / or bfloat16 rows (dev_o_bf16), the log-sum-exp optional.
 *   sola_attention_backward_bf16: sola_attention_backward_ws on bfloat16 q / k / v with the gradients written as bfloat16 rows; dev_dq_scratch
 *     = an f32 [q rows, ld_dq] matrix (units of more keys than one key group accumulate dQ there).  One-pass kernel shapes and sequences
 *     of <= 4 steps.  dev_dout: f32 rows, or (dout_is_bf16, with the bf16 products on: sola_tune "attn_bwd_bf16_mfma") bfloat16 rows of pitch ldo;
 *     dev_o likewise (o_is_bf16, only together with dout_is_bf16). */
int sola_gemm_nt_bf16(const void* dev_a, int lda, const void* dev_w, const float* dev_bias, const void* dev_r, int ldr, int r_is_bf16,
                      void* dev_c, int ldc, int c_is_bf16, int M, int N, int K, void* stream);
int sola_attention_bf16(const void* dev_q, int ldq, const void* dev_k, int ldk, const void* dev_v, int ldv, float* dev_o, void* dev_o_bf16, int ldo,
                        int G, int H, int head_dim, int Sq, int Sk, int inner, int64_t q_outer, int64_t q_inner, int64_t q_row_stride,
                        int64_t k_outer, int64_t k_inner, int64_t k_row_stride, float scale, float* dev_lse, void* stream);
int sola_attention_backward_bf16(const void* dev_q, int ldq, const void* dev_k, int ldk, const void* dev_v, int ldv, const void* dev_o, int o_is_bf16,
                                 const void* dev_dout, int dout_is_bf16, int ldo, const float* dev_lse, void* dev_dq_bf16, void* dev_dk_bf16, void* dev_dv_bf16,
                                 int ld_dq, int ld_dk, int ld_dv, float* dev_dq_scratch, float* dev_dvec, int G, int H, int head_dim, int Sq, int Sk,
                                 int inner, int64_t q_outer, int64_t q_inner, int64_t q_row_stride, int64_t k_outer, int64_t k_inner,
                                 int64_t k_row_stride, float scale, int64_t q_rows, float* dev_scratch, size_t scratch_floats, void* stream);

/* ---- forward: replaces LanguageAlignedTrackSelectionModule.forward (module/module.py:130-162) ------------------ */
size_t sola_workspace_bytes(const SolaCtx* ctx, int B, int N, int T, int L);
int sola_forward(SolaCtx* ctx,
                 const float* dev_object_tokens, /* [B,N,T,object_token_dim] */
                 const float* dev_lang_tokens,   /* [B,L,lang_token_dim]     */
                 int B, int N, int T, int L,
                 float* dev_score_map,           /* [B,N]                    */
                 float* dev_score_tokens,        /* [B,N,lang_token_dim]     */
                 void* dev_workspace, size_t workspace_bytes, void* stream);

/* ---- ragged batches: many (video, expression) samples of different shapes in one pass ---------------------------------
 * The reference scores ONE sample per call (configs/mevis/default.yaml:37,42,47 batch_size 1; inference.py:44-58,
 * evaluator.py:88-112) with per-sample N tracks, T frames, L text tokens.  sola_forward_ragged takes the token rows of all
 * samples concatenated - no padding: no GroupNorm statistic or softmax sees a token that is not the sample's own - and
 * separates the VIDEOS (object sets) from the SAMPLES that refer to them: everything that does not depend on the text (the
 * motion encoder and layer 0's inter-object and motion sub-blocks, 57 % of a sample's FLOPs at the headline shape) is
 * computed once per video and shared by all its expressions, which inference.py:44-58 recomputes per expression.
 *   dev_object_tokens [sum_v N_v*T_v, object_token_dim]   video-major, then track, then frame (= [N_v,T_v,d] blocks)
 *   dev_lang_tokens   [sum_i L_i, lang_token_dim]          sample-major
 *   dev_score_map     [sum_i N_v(i)]                       sample-major, then track
 *   dev_score_tokens  [sum_i N_v(i), lang_token_dim]
 * All descriptor arrays are HOST arrays, read before the call returns.  n_samples == n_videos with sample_video[i] == i is
 * the plain ragged batch.  Results equal per-sample sola_forward calls up to f32 summation order. */
typedef struct SolaRaggedBatch {
    int32_t n_videos;
    const int32_t* video_tracks;     /* [n_videos]  N_v >= 1 */
    const int32_t* video_frames;     /* [n_videos]  T_v >= 1 */
    int32_t n_samples;
    const int32_t* sample_video;     /* [n_samples] index of the sample's video */
    const int32_t* sample_text_len;  /* [n_samples] L_i >= 1 */
} SolaRaggedBatch;
size_t sola_ragged_workspace_bytes(const SolaCtx* ctx, const SolaRaggedBatch* batch);
int sola_forward_ragged(SolaCtx* ctx, const float* dev_object_tokens, const float* dev_lang_tokens, const SolaRaggedBatch* batch,
                        float* dev_score_map, float* dev_score_tokens, void* dev_workspace, size_t workspace_bytes, void* stream);
/* sola_loss for a ragged batch: sample b owns the tracks dev_track_offsets[b] .. dev_track_offsets[b+1] (DEVICE int32,
 * n_samples + 1 entries) of the concatenated score_map / score_tokens / labels; dev_pos_tokens is [n_samples, D].  Every
 * sample gets its own means - what train.py:98-113 / evaluator.py compute at the reference's batch size of 1 - in
 * dev_loss3 [n_samples, 3] = {total, bce, alignment}.  Scratch: 3 * total_tracks floats. */
int sola_loss_ragged(const float* dev_score_map, const float* dev_score_tokens, const float* dev_labels,
                     const float* dev_pos_tokens, const float* dev_neg_tokens, int64_t neg_batch_stride, int n_samples,
                     const int32_t* dev_track_offsets, int max_tracks, int64_t total_tracks, int D, int n_neg,
                     float positive_weight, float temperature, float alignment_weight, float* dev_loss3,
                     int32_t* dev_neg_argmax, void* dev_scratch, size_t scratch_bytes, void* stream);

/* Location of a named intermediate of the LAST forward inside the workspace (parity tests / debugging).
 * Names: conv0..conv5 (pre-norm encoder conv outputs [B*N*T_l, C_l]), pe [T',D], lang [B*W,D],
 * l<i>_obj, l<i>_motion, l<i>_o2l (post-GroupNorm activations [B*N*T', D]). */
int sola_workspace_tap(const SolaCtx* ctx, const char* name, size_t* byte_offset, int64_t* rows, int64_t* cols);

/* ---- losses: replaces train.py:98-113 (weighted BCE + AlignmentLoss.forward, tools/loss.py:14-58) --------------
 * dev_neg_tokens is [B,n_neg,D] with neg_batch_stride = n_neg*D, or one [n_neg,D] table shared by every sample with
 * neg_batch_stride = 0 (train.py:92 repeats negative_token.weight over the batch).
 * dev_loss3 receives {total, bce, alignment}; dev_neg_argmax (optional, int32 [B,N]) the hardest-negative index.
 * Scratch: 3*B*N floats. */
int sola_loss(const float* dev_score_map, const float* dev_score_tokens,
              const float* dev_labels,     /* [B,N] {0,1} */
              const float* dev_pos_tokens, /* [B,1,D]     */
              const float* dev_neg_tokens, int64_t neg_batch_stride,
              int B, int N, int D, int n_neg,
              float positive_weight, float temperature, float alignment_weight,
              float* dev_loss3, int32_t* dev_neg_argmax,
              void* dev_scratch, size_t scratch_bytes, void* stream);

/* ---- selection: replaces inference.py:59-60 (sigmoid, strict > threshold) ------------------------------------- */
int sola_select(const float* dev_score_map, int64_t n, float threshold, float* dev_prob, float* dev_pred, void* stream);

/* ---- per-stage entry points (each is one kernel family; used by the parity tests and by sola_forward) ---------- */
/* module/ws.py:9-13: w [cout,cin,k] -> standardised, re-laid-out [cout, k*cin] (k-major) */
int sola_ws_standardize(const float* dev_w, int cout, int cin, int k, float* dev_out, void* stream);
/* C[M,N] = A[M,K] * W[N,K]^T + bias[N] (+ R[M,N]); F.linear (tools/attention.py:63-65,73). K % 4 == 0. */
int sola_gemm_nt(const float* dev_a, int lda, const float* dev_w, const float* dev_bias, const float* dev_r, int ldr,
                 float* dev_c, int ldc, int M, int N, int K, void* stream);
/* C[M,N] = A[M,K] * Wcat[K,N] (+ R[M,N]), Wcat = up to three row-major matrices of w_rows rows each stacked along K (dev_w1 / dev_w2 NULL
 * when K == w_rows / 2 w_rows): the input gradient of F.linear - dX = dY W, d[x] = [dq|dk|dv] [Wq;Wk;Wv] (tools/attention.py:63-65 backward)
 * - with the weights read where they lie, no transposed copy.  The few-row exact-f32 shape only (one sample per optimizer step,
 * train.py:116-125): M <= 2048, N % 32 == 0, K % 128 == 0, w_rows % 32 == 0, 16-byte aligned rows; SOLA_ERR_ARG otherwise.  Bit-identical
 * to sola_gemm_nt on the transposed copy. */
int sola_gemm_nn(const float* dev_a, int lda, const float* dev_w0, const float* dev_w1, const float* dev_w2, int w_rows, const float* dev_r,
                 int ldr, float* dev_c, int ldc, int M, int N, int K, void* stream);
/* channels-last conv1d along T (module/ws.py:14-22): x [R,T_in,cin], w_std [cout,k*cin] -> y [R,T_out,cout] */
int sola_conv1d_cl(const float* dev_x, const float* dev_wstd, const float* dev_bias, float* dev_y,
                   int R, int T_in, int cin, int cout, int k, int stride, int pad, void* stream);
/* GroupNorm over token sets (nn.GroupNorm; module/module.py:76,34,43,49): for instance i the tokens are rows
 * (i / inner) * outer_stride + (i % inner) * inner_stride + j * tok_stride, j < ntok, of an [*, C] matrix.
 * y = gn(x) (optionally LeakyReLU(slope)); if dev_pe != NULL also y2 = y + pe[(i % inner)] (module.py:38). */
int sola_group_norm(const float* dev_x, float* dev_y, float* dev_y2, const float* dev_pe,
                    const float* dev_gamma, const float* dev_beta,
                    int n_inst, int inner, int64_t outer_stride, int64_t inner_stride, int64_t tok_stride,
                    int ntok, int C, int groups, float eps, float leaky_slope, int apply_leaky, void* stream);
/* softmax(q k^T * scale) v for G groups x H heads (tools/attention.py:66-72).  Row r of group g lives at matrix row
 * (g / inner) * outer + (g % inner) * inner_stride + r * row_stride (q and o share addressing; k and v share). */
int sola_attention(const float* dev_q, int ldq, const float* dev_k, int ldk, const float* dev_v, int ldv,
                   float* dev_o, int ldo, int G, int H, int head_dim, int Sq, int Sk, int inner,
                   int64_t q_outer, int64_t q_inner, int64_t q_row_stride,
                   int64_t k_outer, int64_t k_inner, int64_t k_row_stride, float scale,
                   float* dev_lse /* optional [q rows, H]: log-sum-exp, needed by the backward */, void* stream);
/* module/module.py:112-128: pe [t_len, D] */
int sola_pos_encoding(const float* dev_gauss, int D, int t_len, int max_temporal_length, float* dev_pe, void* stream);

/* ---- training: what loss.backward() does to this path under autograd (train.py:116-117) ----------------------------
 * sola_forward_train = sola_forward that keeps every attention's q/k/v/output/residual and the softmax log-sum-exp in
 * the workspace (sola_train_workspace_bytes); the workspace and dev_object_tokens must stay untouched until
 * sola_backward has run.  sola_backward consumes d(score_map), d(score_tokens) and OVERWRITES the gradient buffer
 * registered with sola_set_grad for each of the 83 parameters (same names and sizes as sola_set_weight; the Fourier
 * buffer has no gradient).  Gradients flowing into negative_token.weight through the loss's neg_tokens argument come
 * from sola_loss_backward (d_neg) and are added by the caller, exactly as autograd does for train.py:92. */
/* Dropout of the NEXT sola_forward_train (and of the sola_backward that follows it): p_encoder after every encoder
 * LeakyReLU (nn.Dropout(p=dropout_p), module/module.py:78-94), p_attention on the attention probabilities
 * (tools/attention.py:12,71, hard-coded 0.1 in the reference).  Masks are a pure function of (seed, element index), so
 * nothing is stored; a given seed reproduces the same masks.  p = 0 disables (eval-mode numerics). */
int sola_set_dropout(SolaCtx* ctx, float p_encoder, float p_attention, uint64_t seed);
/* The same mask generator for the per-stage entry points sola_group_norm[_backward] / sola_attention[_backward]
 * (thread-local; used by the parity tests). */
int sola_set_stage_dropout(float p, uint64_t seed);
size_t sola_train_workspace_bytes(const SolaCtx* ctx, int B, int N, int T, int L);
size_t sola_backward_workspace_bytes(const SolaCtx* ctx, int B, int N, int T, int L);
int sola_set_grad(SolaCtx* ctx, const char* name, void* dev_ptr, int64_t numel);
int sola_forward_train(SolaCtx* ctx, const float* dev_object_tokens, const float* dev_lang_tokens,
                       int B, int N, int T, int L, float* dev_score_map, float* dev_score_tokens,
                       void* dev_workspace, size_t workspace_bytes, void* stream);
int sola_backward(SolaCtx* ctx, const float* dev_d_score_map, const float* dev_d_score_tokens,
                  const void* dev_forward_workspace, void* dev_scratch, size_t scratch_bytes, void* stream);
/* ---- one optimizer step's device work in ONE call (round 5) ----------------------------------------------------------------------
 * The body of the reference's training loop - forward, weighted BCE + alignment loss, loss.backward(), get_grad_norm_dict(), gradient
 * clipping (train.py:62-125 at its batch size of one sample, configs/mevis/default.yaml:37) - enqueued from C++ on `stream`: the ~110
 * launches of a one-sample step cost 1.6-2.4 ms of host time when driven call by call from Python, more than their GPU time.
 * Same kernels, arguments and order as sola_forward_train + sola_loss + sola_loss_backward + sola_backward + sola_grad_sqnorms +
 * sola_grad_clip called one by one (bit-identical gradients).  The optimizer update is the caller's: the gradients are where
 * sola_set_grad bound them.
 *   sola_train_step_bind: the parameters in the order module/module.py:164-199 walks them (names[i] of group group[i]: encoder, every
 *     layer, negative tokens) - the gradient-norm reduction takes them in that order.  Call once per context (after sola_set_grad).
 *   dev_loss3 [3] = {total, bce, alignment} (train.py:98-113); dev_grad_sq [n_groups + 1] doubles = the groups' sums of squares of the
 *     UNclipped gradients and their total (module/module.py:164-199 reports the square roots); max_grad_norm <= 0: no clipping.
 *   dev_labels [B, N], dev_pos_tokens [B, 1, D]; the negative tokens are the context's own `negative_token.weight` (train.py:92), whose
 *     gradient receives both of its contributions (through the network and straight from the alignment loss).
 *   Workspaces (caller-owned): sola_train_workspace_bytes / sola_backward_workspace_bytes / sola_train_step_workspace_bytes. */
int sola_train_step_bind(SolaCtx* ctx, const char* const* names, const int32_t* group, int n, int n_groups);
size_t sola_train_step_workspace_bytes(const SolaCtx* ctx, int B, int N);
int sola_train_step(SolaCtx* ctx, const float* dev_object_tokens, const float* dev_lang_tokens, int B, int N, int T, int L,
                    const float* dev_labels, const float* dev_pos_tokens, float positive_weight, float temperature, float alignment_weight,
                    float max_grad_norm, float* dev_score_map, float* dev_score_tokens, float* dev_loss3, double* dev_grad_sq,
                    void* dev_train_workspace, size_t train_workspace_bytes, void* dev_backward_scratch, size_t backward_scratch_bytes,
                    void* dev_step_workspace, size_t step_workspace_bytes, void* stream);
/* Gradient clipping + the AdamW update (train.py:121-125) in ONE multi-tensor launch, with torch.optim.AdamW(fused=True)'s own arithmetic
 * (ATen/native/cuda/fused_adam_utils.cuh: double scalars, float tensors, the same expressions): bit-identical parameters and moments.
 *   sola_adamw_bind: the optimizer's state tensors per parameter name (exp_avg, exp_avg_sq as float arrays of the parameter's size; step =
 *     torch's per-parameter device float counter, or null).  Parameters and gradients are the context's own bindings
 *     (sola_set_weight / sola_set_grad): the update writes the caller's parameter storage.  Bind again when a pointer changes: a
 *     sola_set_weight / sola_set_grad with a NEW pointer unbinds the optimizer and sola_adamw_step fails until the next bind.
 *   sola_adamw_step: `step` = 0: this update's number is read from the bound step tensors ON THE DEVICE (counter + 1) and the kernel
 *     advances them, exactly as torch's own step() does - the two may be mixed on one optimizer, and a load_state_dict into the same
 *     storage is picked up (needs step tensors); `step` >= 1: an explicit number (1, 2, ...).  max_grad_norm > 0 scales every gradient by min(1, max_norm / (sqrt(*dev_total_sq)
 *     + 1e-6)) first (torch.nn.utils.clip_grad_norm_), decided on the device; write_back_grads != 0 leaves the scaled gradients in the
 *     gradient tensors as clip_grad_norm_ does, 0 leaves them unclipped (an eighth less traffic; train.py:121-125 never reads them again). */
int sola_adamw_bind(SolaCtx* ctx, const char* const* names, void* const* dev_exp_avg, void* const* dev_exp_avg_sq, void* const* dev_step, int n);
int sola_adamw_step(SolaCtx* ctx, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, const double* dev_total_sq,
                    float max_grad_norm, int write_back_grads, void* stream);
/* ---- ragged training step: many (video, expression) samples of DIFFERENT shapes per optimizer step -------------------------
 * The reference trains at batch size 1 (configs/mevis/default.yaml:37; train.py:62-137: one forward, one backward, one AdamW
 * step per sample) because every sample has its own N tracks, T frames and L text tokens (dataloader.py:119-163,187-199).
 * sola_forward_train_ragged / sola_backward_ragged run the same step over a SolaRaggedBatch: token rows concatenated without
 * padding (layouts as sola_forward_ragged), every GEMM of the forward and backward over all rows at once, every shape-dependent
 * kernel (conv windows and their transposes, GroupNorm and its backward, the three attentions and their backward, score head)
 * reading per-unit tables built on the device.  One sample per video, in order (n_samples == n_videos, sample_video[i] == i):
 * under training-mode dropout each sample has its own masks, so nothing is shared between the expressions of a video.
 * The parameter gradients sola_backward_ragged leaves are the SUM over the samples of what one-sample calls produce for the
 * same upstream d(score_map) / d(score_tokens) (weighting per sample is the caller's, through those: sola_loss_backward_ragged
 * takes one upstream triple per sample).  The workspace holds the unit tables: hand the SAME workspace to sola_backward_ragged. */
size_t sola_train_ragged_workspace_bytes(const SolaCtx* ctx, const SolaRaggedBatch* batch);
size_t sola_backward_ragged_workspace_bytes(const SolaCtx* ctx, const SolaRaggedBatch* batch);
int sola_forward_train_ragged(SolaCtx* ctx, const float* dev_object_tokens, const float* dev_lang_tokens, const SolaRaggedBatch* batch,
                              float* dev_score_map, float* dev_score_tokens, void* dev_workspace, size_t workspace_bytes, void* stream);
int sola_backward_ragged(SolaCtx* ctx, const float* dev_d_score_map, const float* dev_d_score_tokens,
                         const void* dev_forward_workspace, void* dev_scratch, size_t scratch_bytes, void* stream);
/* Backward of sola_loss_ragged: dev_g3 [n_samples, 3] = upstream gradients of every sample's {total, bce, alignment} (the mean
 * of the per-sample totals - train.py:113 averaged over the batch - is g3[b] = {1/n_samples, 0, 0}).  Outputs as
 * sola_loss_backward over the concatenated tracks; d_neg [n_samples, n_neg, D], or [n_neg, D] summed over the samples when
 * neg_batch_stride == 0.  Scratch: total_tracks*n_neg floats (+ n_samples*n_neg*D for shared negatives). */
int sola_loss_backward_ragged(const float* dev_score_map, const float* dev_score_tokens, const float* dev_labels,
                              const float* dev_pos_tokens, const float* dev_neg_tokens, int64_t neg_batch_stride, int n_samples,
                              const int32_t* dev_track_offsets, int max_tracks, int64_t total_tracks, int D, int n_neg,
                              float positive_weight, float temperature, float alignment_weight, const float* dev_g3,
                              float* dev_d_score_map, float* dev_d_score_tokens, float* dev_d_neg, void* dev_scratch,
                              size_t scratch_bytes, void* stream);
/* Gradient buckets for multi-GPU training (train.py under torchrun: one RCCL all-reduce of the gradient per step, the only
 * collective of the path).  sola_backward finishes the parameters' gradients in a fixed order: alignment layer n-1, ...,
 * layer 1, then layer 0 together with negative_token.weight (which collects contributions from every layer), then the
 * encoder - sola_grad_bucket_count() = n_layers + 1 buckets, sola_grad_bucket_of(name) gives a parameter's bucket.  A caller
 * that lays its gradient buffers out bucket by bucket in ONE flat allocation can all-reduce each bucket in place, and
 * sola_backward_wait_bucket(ctx, k, stream) makes `stream` wait (device-side, hipStreamWaitEvent) for the moment bucket k
 * is final, so its all-reduce overlaps the rest of the backward.  Valid after sola_backward until the next one. */
int sola_grad_bucket_count(const SolaCtx* ctx);
int sola_grad_bucket_of(const SolaCtx* ctx, const char* name);
int sola_backward_wait_bucket(SolaCtx* ctx, int bucket, void* stream);
/* Backward of sola_loss.  dev_g3 = upstream gradients of {total, bce, alignment} (3 floats on the device).
 * Writes d(score_map) [B,N], d(score_tokens) [B,N,D] and (optional) d(neg_tokens) [B,n_neg,D].  Scratch: B*N*n_neg floats. */
int sola_loss_backward(const float* dev_score_map, const float* dev_score_tokens, const float* dev_labels,
                       const float* dev_pos_tokens, const float* dev_neg_tokens, int64_t neg_batch_stride,
                       int B, int N, int D, int n_neg, float positive_weight, float temperature, float alignment_weight,
                       const float* dev_g3, float* dev_d_score_map, float* dev_d_score_tokens, float* dev_d_neg,
                       void* dev_scratch, size_t scratch_bytes, void* stream);
/* Gradient statistics of the training step.  sola_grad_sqnorms: out[g] = sum over the tensors of group g of |grad|^2
 * for g < n_groups and out[n_groups] = their total (dev_out holds n_groups + 1 doubles)
 * (replaces get_grad_norm_dict, module/module.py:164-199, which syncs the host once per parameter); the pointer /
 * numel / group arrays are HOST arrays of n <= 128 entries.  sola_grad_clip scales every tensor in place by
 * min(1, max_norm / (sqrt(*dev_total_sq) + 1e-6)) (torch.nn.utils.clip_grad_norm_, train.py:121-122) without a host sync. */
size_t sola_grad_sqnorms_scratch_bytes(int n, const int64_t* numel);
int sola_grad_sqnorms(const float* const* dev_grads, const int64_t* numel, const int32_t* group, int n, int n_groups,
                      double* dev_out, void* dev_scratch, size_t scratch_bytes, void* stream);
int sola_grad_clip(float* const* dev_grads, const int64_t* numel, int n, const double* dev_total_sq, float max_norm,
                   void* stream);
/* per-stage backward entry points (parity tests) */
int sola_ws_backward(const float* dev_w, const float* dev_dwstd, int cout, int cin, int k, float* dev_dw, void* stream);
/* The same attention with q, k, v given as split-f16 rows: every product runs as three v_mfma_f32_16x16x16_f16 with f32
 * accumulation instead of the exact-f32 MFMA (the inference fast path; sequences longer than 16 steps only). */
int sola_attention_split(const float* dev_q_sp, int ldq, const float* dev_k_sp, int ldk, const float* dev_v_sp, int ldv,
                         float* dev_o, int ldo, int o_is_split, int G, int H, int head_dim, int Sq, int Sk, int inner,
                         int64_t q_outer, int64_t q_inner, int64_t q_row_stride,
                         int64_t k_outer, int64_t k_inner, int64_t k_row_stride, float scale, float* dev_lse, void* stream);
/* C[N,K] = A[M,N]^T * B[M,K] (weight gradient), optional bias_grad[N] = column sums of A */
size_t sola_gemm_tn_scratch_bytes(int M, int N, int K);
int sola_gemm_tn(const float* dev_a, int lda, const float* dev_b, int ldb, float* dev_c, float* dev_bias_grad,
                 int M, int N, int K, void* dev_scratch, size_t scratch_bytes, void* stream);
/* The same weight gradient on the split-f16 MFMA path (training precision 1): both operands are written transposed in
 * the split-f16 format (dY with a power-of-two scale found on the device), the product is the NT split GEMM with the
 * reduction cut into ranges, partial sums folded in a fixed order.  N % 8 == 0, K % 8 == 0, M >= 64. */
size_t sola_gemm_tn_split_scratch_bytes(int M, int N, int K);
int sola_gemm_tn_split(const float* dev_a, int lda, const float* dev_b, int ldb, float* dev_c, int M, int N, int K,
                       void* dev_scratch, size_t scratch_bytes, void* stream);
/* The same weight gradient on plain 16-bit operands (fmt 1 = f16, 2 = bfloat16: training with 16-bit GEMM operands, and the dW
 * products of the default split-f16 step), ONE MFMA per product.  With N % 256 == 0 and K % 256 == 0 no transposed copy is made: the
 * operands are cast row-major and the kernel transposes between LDS and the matrix pipe (ds_read_b64_tr_b16; sola_tune "train_tn_tr"
 * 0 = the transposed-copy route for every shape).  Scratch as for sola_gemm_tn_split. */
int sola_gemm_tn_f16(const float* dev_a, int lda, const float* dev_b, int ldb, float* dev_c, int M, int N, int K, int fmt,
                     void* dev_scratch, size_t scratch_bytes, void* stream);
/* dW_std[cout][k*cin] of the channels-last conv (module/ws.py:14-22) on 16-bit operands: dy [R*T_out][cout], x [R*T_in][cin].  With
 * cout % 256 == 0 and cin % 256 == 0 the conv input is cast ONCE, row-major, and the kernel gathers the taps (implicit im2col) in its
 * LDS-DMA addresses; else one transposing cast per tap.  Scratch: sola_gemm_tn_split_scratch_bytes(R*T_out, cout, k*cin). */
int sola_conv1d_cl_wgrad_f16(const float* dev_x, const float* dev_dy, float* dev_dwstd, int R, int T_in, int cin, int cout, int k,
                             int stride, int pad, int fmt, void* dev_scratch, size_t scratch_bytes, void* stream);
int sola_conv1d_cl_backward(const float* dev_x, const float* dev_wstd, const float* dev_dy, float* dev_dx,
                            float* dev_dwstd, float* dev_dbias, int R, int T_in, int cin, int cout, int k, int stride,
                            int pad, void* dev_scratch, size_t scratch_bytes, void* stream);
/* ... on the split-f16 MFMA path (what sola_backward runs under precision 1); cout % 128 == 0, cin % 8 == 0, R*T_out >= 64 */
size_t sola_conv1d_cl_backward_split_scratch_bytes(int R, int T_in, int cin, int cout, int k, int stride, int pad);
int sola_conv1d_cl_backward_split(const float* dev_x, const float* dev_wstd, const float* dev_dy, float* dev_dx,
                                  float* dev_dwstd, float* dev_dbias, int R, int T_in, int cin, int cout, int k, int stride,
                                  int pad, void* dev_scratch, size_t scratch_bytes, void* stream);
int sola_group_norm_backward(const float* dev_x, const float* dev_dy, const float* dev_dy2, const float* dev_gamma,
                             const float* dev_beta, float* dev_dx, float* dev_dgamma, float* dev_dbeta,
                             int n_inst, int inner, int64_t outer_stride, int64_t inner_stride, int64_t tok_stride,
                             int ntok, int C, int groups, float eps, float leaky_slope, int apply_leaky,
                             void* dev_scratch, size_t scratch_bytes, void* stream);
int sola_attention_backward(const float* dev_q, int ldq, const float* dev_k, int ldk, const float* dev_v, int ldv,
                            const float* dev_o, const float* dev_dout, int ldo, const float* dev_lse,
                            float* dev_dq, float* dev_dk, float* dev_dv, float* dev_dvec /* [q rows, H] scratch */,
                            int G, int H, int head_dim, int Sq, int Sk, int inner,
                            int64_t q_outer, int64_t q_inner, int64_t q_row_stride,
                            int64_t k_outer, int64_t k_inner, int64_t k_row_stride, float scale, void* stream);
/* ... with scratch for the one-pass kernel's long query ranges (round 3).  Units of at most 128 queries and 128 keys take the
 * one-pass kernel either way (one block per (unit, head): S / dP / dS formed once, q, k, v, o, dO read once; sola_tune
 * "attn_bwd_fused" 0 = the two-pass kernels).  Longer query ranges against <= 64 keys (object -> language) take it when
 * dev_scratch holds sola_attention_backward_scratch_floats(q rows, G, H, Sk) floats and the units' rows are consecutive
 * (inner == 1 and q_row_stride == 1): 256-query chunks, one block each, dK / dV partial sums added in chunk order. */
size_t sola_attention_backward_scratch_floats(int64_t q_rows, int G, int H, int Sk);
int sola_attention_backward_ws(const float* dev_q, int ldq, const float* dev_k, int ldk, const float* dev_v, int ldv,
                               const float* dev_o, const float* dev_dout, int ldo, const float* dev_lse,
                               float* dev_dq, float* dev_dk, float* dev_dv, float* dev_dvec, int G, int H, int head_dim, int Sq, int Sk,
                               int inner, int64_t q_outer, int64_t q_inner, int64_t q_row_stride, int64_t k_outer, int64_t k_inner,
                               int64_t k_row_stride, float scale, int64_t q_rows, float* dev_scratch, size_t scratch_floats, void* stream);

/* ---- mask IoU: replaces track_generation/seg_utils.py:128-142 (compute_mask_iou), :109-125 (compute_masklet_iou)
 * and the prompt-mask nearest resize of generate_tokens_grid.py:269-272 ------------------------------------------
 * Masks are packed to 1 bit/pixel (words of 32 pixels, row-major over the H*W comparison grid).
 * elem_type: 0 = uint8, 1 = float32; a pixel is set when its value != 0.  If (h,w) != (H,W) the source is
 * resampled with ATen's nearest rule while packing.  dev_area receives the per-mask popcount (int64). */
int64_t sola_mask_words(int H, int W);
int sola_mask_pack(const void* dev_masks, int elem_type, int n, int h, int w, int H, int W,
                   uint32_t* dev_bits, int64_t* dev_area, void* stream);
/* inter[p,r] = popcount(A[a_index(p,r)] & B[r]), union = area_A + area_B - inter.
 * dev_a_frame (optional, int32 [R]): A holds P masklets of T frames (A index = p*T + frame[r]), which is the
 * "pred masklet at the prompt's frame" gather of generate_tokens_grid.py:269.  T=1, NULL = plain P x R matrix. */
int sola_mask_pair_counts(const uint32_t* dev_a_bits, const int64_t* dev_a_area, int P, int T,
                          const uint32_t* dev_b_bits, const int64_t* dev_b_area, int R,
                          const int32_t* dev_a_frame, int64_t words,
                          int64_t* dev_inter, int64_t* dev_union, void* stream);
/* One call for the common case: A [P,H,W], B [R,h,w] -> inter/union [P,R] (scratch: (P+R)*(words*4+8) bytes). */
size_t sola_mask_iou_scratch_bytes(int P, int R, int H, int W);
int sola_mask_iou_matrix(const void* dev_a, const void* dev_b, int elem_type, int P, int R, int H, int W, int h, int w,
                         int64_t* dev_inter, int64_t* dev_union, void* dev_scratch, size_t scratch_bytes, void* stream);

/* ---- masklet resampling / decoding on the same packed representation ----------------------
 * sola_mask_bilinear_pack replaces track_generation/seg_utils.py:145-160 (reshape_masklet): bilinear resample
 * (align_corners=False, ATen's source-index rule) of n {0,1} masks [n,h,w] to H x W, `> 0.5`, bit-pack, area.
 * elem_type 0 = uint8 (non-zero counts as 1.0), 1 = float32 (any values; the fp32 arithmetic order is ATen's),
 * 2 = float32 tracker logits, binarised as (v > 0) while reading — the `(out_mask_logits > 0.0).float()` of
 * generate_tokens_grid.py:215-222 folded in.  W <= 4096. */
int sola_mask_bilinear_pack(const void* dev_masks, int elem_type, int n, int h, int w, int H, int W,
                            uint32_t* dev_bits, int64_t* dev_area, void* stream);
/* packed bits -> {0,1} images [n,H,W] (elem_type 0 = uint8, 1 = float32): the tensor reshape_masklet returns. */
int sola_mask_unpack(const uint32_t* dev_bits, int n, int H, int W, void* dev_out, int elem_type, void* stream);
/* COCO run-length masks -> OR of K masks per frame, replaces dataloader.py:305-369 (rle_masklet_decode + np.logical_or
 * in get_sam2_masklet / get_gt_masklet).  Mask (frame f, slot k) owns the runs dev_off[f*K+k] .. dev_off[f*K+k+1] of
 * dev_cum, the inclusive prefix sums of its run lengths (uint32, column-major positions as in pycocotools); an empty
 * range is an absent / unselected mask.  dev_out [n_frames,h,w] uint8 row-major and/or dev_bits + dev_area. */
int sola_rle_fill_or(const uint32_t* dev_cum, const int64_t* dev_off, int n_frames, int K, int h, int w,
                     uint8_t* dev_out, uint32_t* dev_bits, int64_t* dev_area, void* stream);

/* Host-only helper: COCO compressed RLE string (pycocotools rleFrString format) -> inclusive prefix sums of its run
 * lengths in host_cum[0..cap).  Returns the number of runs, or a negative status (malformed string, more than cap runs,
 * or runs covering more than `limit` pixels; limit < 0 disables that check). */
int64_t sola_rle_string_to_cum(const char* str, int64_t len, uint32_t* host_cum, int64_t cap, int64_t limit);

/* ---- in-library kernel timing (HIP events on the launch stream; used by bench.py's roofline object) ------------ */
enum { SOLA_PROF_GEMM = 0,      /* gemm_nt_f32_kernel<128,128> */
       SOLA_PROF_ATTN = 1,      /* attn_fwd_f32_kernel */
       SOLA_PROF_NORM = 2,      /* group_norm_kernel */
       SOLA_PROF_WS = 3,        /* ws_standardize_kernel */
       SOLA_PROF_HEAD = 4,      /* score head / loss / select */
       SOLA_PROF_MISC = 5,      /* pos-encoding, lang concat */
       SOLA_PROF_IOU_PACK = 6,  /* mask_pack_kernel */
       SOLA_PROF_IOU_PAIR = 7,  /* mask_pair_kernel */
       SOLA_PROF_GEMM_SMALL = 8, /* gemm_nt_f32_kernel<64,64> (small grids) */
       SOLA_PROF_GEMM_TN = 9,   /* gemm_tn_f32_kernel (weight gradients) */
       SOLA_PROF_ATTN_BWD = 10, /* attn_bwd_* kernels */
       SOLA_PROF_GEMM_SPLIT = 11, /* split-f16 GEMMs other than the 256x256 shape (128x128 glds blocks, 64x64 small grids) */
       SOLA_PROF_GEMM_SPLIT256 = 12, /* gemm_nt_split_glds_persist_kernel<*> (and the one-tile <4,2,4,*> kernel): 256x256 blocks, split-f16, 3 x f16 MFMA */
       SOLA_PROF_GEMM_SPLIT256_GN = 13, /* the same kernel with GroupNorm + LeakyReLU applied in the epilogue (encoder conv0-2): its time includes the norm */
       SOLA_PROF_NCAT = 14 };
/* Kernel-schedule switches for within-process A/B measurements and tests (e.g. "train_bf16_store", "iou_fused", "attn_split_min_keys",
 * "infer_f32_rows", "train_split_min_rows").  Not part of the drop-in boundary: the defaults are the shipped path.  The catalogue of keys,
 * values and what each was measured to do lives in docs/tune_keys.md; an unknown key returns SOLA_ERR_ARG.  Except under the *_ablate keys
 * (measurement only), results are identical across variants up to f32 summation order. */
int sola_tune(const char* key, int value);
/* 1 when the library was built with EXPERIMENTS=1 (make -C sola_amd/csrc EXPERIMENTS=1): the closed experiments' kernels and their
 * sola_tune keys (gemm_pp, gemm_nw4, gemm_k16, gemm_stagger, gemm_order, gemm_trace, gemm_ld, gemm_gn_fuse, gemm_ablate, attn_bwd_ablate,
 * attn_reg_minw, attn_res_splitm, gemm_f32p_ablate) exist only there; the default library rejects those keys. */
int sola_has_experiments(void);
/* Self-test of the library's cross-lane primitives on the current device: wave sums / maxima on v_permlane*_swap + DPP against the
 * ds_bpermute butterfly, every lane of every step bit for bit (synchronises the stream).  SOLA_OK or SOLA_ERR_STATE + sola_last_error(). */
int sola_selftest(void* stream);

/* Measurement only (no reference counterpart): with sola_tune "gemm_trace" 1 the plain persistent split-f16 GEMM (no conv, no
 * residual, f32 output) runs an instrumented instantiation that records, per (block, wave), the cycles spent at the k-tile wait +
 * barrier, in the k-loops and in the epilogues, and real-time stamps per tile.  Copies the record of the last traced launch to
 * `host` (device-synchronising); returns the bytes written, 0 if nothing was traced, < 0 on error.  Layout: tools/gemm_trace.py. */
long long sola_gemm_trace_read(void* host, long long bytes);
/* In-library kernel timing: two HIP events on the launch stream around every kernel launch.  enable 0 = off, 1 = every category,
 * a larger value = only the categories c with bit (c + 1) set (each timed launch costs two event records, ~1.5 us of stream time
 * each: 2.8 % of the headline step with every category on, tools/prof_overhead.py). */
int sola_profile_enable(int enable);
/* Synchronises the recorded events and returns, per category: launches, total milliseconds, algorithmic flops,
 * algorithmic bytes accumulated since the last reset. Arrays have SOLA_PROF_NCAT entries. */
int sola_profile_read(int64_t* launches, double* ms, double* flops, double* bytes, int reset);

#ifdef __cplusplus
}
#endif
#endif /* SOLA_HIP_H */
