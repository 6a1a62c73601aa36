// Internal launcher interface shared by api.cpp and the kernel translation units.
#pragma once
#include "common.h"

// ---- GEMM (gemm.hip): C = A * W^T + bias (+R), f32 MFMA ------------------------------------------------------
struct GemmProblem {
    const float* A;     // [M, K] rows (plain) or the conv source x [R*T_in, Cin]
    const float* W;     // [N, K]
    const float* bias;  // [N] or null
    const float* R;     // [M, N] residual or null
    float* C;           // [M, N]
    const float* scale_dev;  // arith 1, optional: this problem's own result multiplier in device memory (the inverse of the
                             // data-dependent power-of-two scale its W operand was cast with); null = 1
};
struct GemmDesc {
    GemmProblem p[3];
    int nprob;  // 1..3 problems sharing every dimension (blockIdx.z selects)
    int M, N, K;
    int lda, ldr, ldc;  // ldw == K
    // implicit im2col for the k=3 channels-last conv: row m = (r, t_out); k = kk*Cin + ci
    int conv;  // 0: plain; 1: gather window; 2: transposed-conv gather
    int T_in, T_out, stride, pad, Cin;
    // conv == 1, optional: ragged batches (sequences of different lengths in one launch).  Output row m reads its window from
    // source row rowmap[m].x (the row of tap 0, may lie outside the sequence) and bit kk of rowmap[m].y says whether tap kk
    // is inside the sequence (else it reads zeros).  T_in / T_out / stride / pad are then unused.
    const int2* rowmap;
    int arith;        // 0: f32 operands (exact f32 MFMA); 1: split-f16 operands (3 x f16 MFMA, f32 accumulate);
                      // 2: plain f16 operands (the 16-bit storage mode: ONE f16 MFMA per product, f32 accumulate).  A and W
                      //    then hold _Float16 rows and K, lda, Cin count halfs (K % 64 == 0, lda % 16 == 0, Cin % 64 == 0);
                      //    direct-to-LDS kernels only
    float out_scale;  // arith 1: multiplier undoing the power-of-two weight pre-scale (0 = 1)
    int ksplit;  // arith 1, direct-to-LDS kernels: > 1 cuts the reduction into that many ranges, one work item each (few output
                 // tiles, long K: weight gradients); needs splitk_ws of nprob * ksplit * M * N floats and K % (32 * ksplit) == 0
    const float* out_scale_dev;  // arith 1, optional: a further multiplier read from device memory (the inverse of the
                                 // data-dependent scale launch_cast_sp16_auto gave the A operand)
    int r_sp16;       // arith 1: the residual R is split-f16
    int c_sp16;       // arith 1: write C as split-f16 pairs (N % 8 == 0), e.g. q/k/v for the split attention kernel
    int r_f16, c_f16; // arith 2: R / C are _Float16 matrices (ldr / ldc count halfs then; N % 4 == 0, pitches % 4 == 0)
    int bf16;         // arith 2: A and W hold bfloat16 rows instead of f16 ones (v_mfma_f32_32x32x16_bf16; f32 C / R only): training with
                      // bf16 GEMM operands (BASELINE config C2)
    const float* bias_scale_dev;  // direct-to-LDS kernels, optional: the bias is multiplied by this device scalar (an output kept in
                                  // the units of a scaled input: conv0 of the 16-bit storage mode)
    // optional scratch for the two-pass split-K of small grids (fewer 64x64 tiles than CUs): S partial sums per problem,
    // [nprob][S][M][N] f32, reduced in a fixed order (deterministic).  Null = never split.
    float* splitk_ws;
    size_t splitk_bytes;
    // arith 1 with c_sp16, optional: device word that gets bit 0 set when a value written as split-f16 is not finite or
    // beyond the f16 range (|v| >= 65000): the inference forward then repeats the call on the exact-f32 kernels
    int* guard;
    // arith 1, optional (check gemm_gn_fusable first): GroupNorm + LeakyReLU of the output applied in the epilogue - instances of
    // gn_tokens (4 / 8 / 16) consecutive rows x 64 channels - and C written as split-f16 pairs (c_sp16 must be 1)
    const float *gn_gamma, *gn_beta;
    int gn_tokens;
    float gn_eps, gn_slope;
    // arith 0, few-row shape only (gemm_nn_supported): the weights in their ROW-MAJOR [K][N] form instead of W[N][K] - the reduction index
    // runs over the rows of up to three matrices of w_nn_rows rows each, stacked (K = their total; pitch N; p[].W unused): a dX GEMM reads
    // the layer's weight matrices where they lie, no transposed copy.  One problem.  0 = off.
    const float* w_nn[3];
    int w_nn_rows;
};
bool gemm_nn_supported(const GemmDesc& d);  // the launch takes the w_nn form (else launch_gemm fails)
int launch_gemm(const GemmDesc& d, hipStream_t s);
bool gemm_gn_fusable(const GemmDesc& d, int channels_per_group, int tokens);

// ---- weight-gradient GEMM + helpers (gemm_tn.hip) ----------------------------------------------------------------
struct GemmTnDesc {
    const float* A;   // dY [M, N]
    const float* B;   // X [M, K] (or conv source when conv = 1)
    float* C;         // [N, K]
    float* bias_grad; // optional [N] = column sums of A
    int M, N, K, lda, ldb;
    int conv, T_in, T_out, stride, pad, Cin;
    float* scratch;
    size_t scratch_bytes;
    const int2* rowmap;  // conv = 1, optional, ragged batches: as GemmDesc::rowmap (T_in / T_out / stride / pad are then unused)
};
// Grouped exact-f32 form (few-sample steps): up to 32 problems dW_j = A_j^T B_j (+ bias gradients), each reduced over all of its own M_j
// rows by the blocks of its 128 x 128 tiles - one launch, no partial sums (gemm_tn.hip)
struct GemmTnGroupDesc {
    struct Prob {
        const float* A;         // dY [M, N]
        const float* B;         // X [M, K], or the channels-last conv input when conv = 1 (K = k * Cin: implicit im2col, as GemmTnDesc)
        float* C;               // [N, K]
        float* bias_grad;       // optional [N]
        const int2* rowmap;     // conv = 1, ragged batches
        int M, N, K, lda, ldb;
        int conv, T_in, T_out, stride, pad, Cin;
    } p[32];
    int nprob;
};
int launch_gemm_tn_group(const GemmTnGroupDesc& d, hipStream_t s);
// Split-f16 version (gemm_tn_split.hip): up to three problems dW_j = A_j^T B_j sharing M, N, K and the pitches
struct GemmTnSplitDesc {
    const float* A[3];  // dY_j [M, N], pitch lda (column slices of one buffer are fine)
    const float* B[3];  // X_j [M, K], pitch ldb; equal pointers are transposed once
    float* C[3];        // dW_j [N, K], contiguous
    int nprob, M, N, K, lda, ldb;
    int conv, T_in, T_out, stride, pad, Cin;  // conv = 1: B is the channels-last conv input [R*T_in, Cin], K = k*Cin (implicit im2col)
    float* scal;  // optional device pair with scal[0] = max|A| over all problems already computed (else found here)
    float* scal_b;  // optional device pair with scal_b[0] = max|B| already computed: B is cast with that power-of-two scale too
                    // (activations whose magnitude the caller does not control: the object tokens); null = B is cast unscaled
    int pure;     // 1: plain f16 transposed operands and ONE f16 MFMA per product (training with f16 GEMM operands), 2: the same in
                  // bfloat16, else split-f16
    // optional: the transposing cast of A also writes the ROW-MAJOR cast of A (same scale, the format of launch_cast_f16_scaled /
    // launch_cast_sp16_scaled) - the operand of the dX GEMM that consumes the same gradient matrix - so dY is read once for both.
    // a_rm addresses the value (row 0, column of A[0]); problem j lands at column A[j] - A[0]; a_rm_ld = row pitch in values.
    // Needs N % 64 == 0 and column offsets % 8 == 0 (else ignored: check gemm_tn_split_writes_rm()).
    float* a_rm;
    int a_rm_ld;
    int rm_split;  // pure != 0 only: the row-major copy stays split-f16 (the dX GEMM of a split-f16 step whose dW products run on plain f16)
    int a_rm_ready;  // pure 16-bit, rm_split == 0: a_rm already holds the cast of every problem (launch_cast_bf16_colsum): no cast here
    float* scratch;
    size_t scratch_bytes;
    const int2* rowmap;  // conv = 1, optional, ragged batches: as GemmDesc::rowmap
    long long B_rows;    // conv = 1 with rowmap: rows of the conv input B (the geometry gives M / T_out * T_in otherwise); 0 = unknown
    const void* B16[3];  // optional, pure != 0: problem j's B already cast row-major in the operand format ([M][K], or the conv input
                         // [rows_in][Cin]), unscaled - the training forward's own operand cast (SolaCtx::x16); used by the row-major route
                         // when every problem has one
    int b16_split;       // ... as SPLIT-f16 pairs (the split-f16 step's forward operands; pure == 1): the kernel takes their hi halves
};
// in [rows][cols] f32 -> out [cols][ld_out] split-f16 (rows rows..ld_out zero-filled; ld_out % 128 == 0); scal: optional
// device pair as for launch_cast_sp16_auto with scal[0] = max|in| already there
int launch_cast_sp16_t(const float* in, int ld_in, float* out, long long ld_out, int rows, int cols, float* scal, hipStream_t s);
// dx[(r, ti)][ci] = sum over taps of z[(r, to)][kk*cin + ci] (the scatter of a transposed conv, as a gather)
int launch_col2im(const float* z, float* dx, long long R, int T_in, int T_out, int cin, int k, int stride, int pad, hipStream_t s, int z_bf16 = 0);  // z_bf16: z is a bfloat16 matrix (round 6)
// ragged batches: input row i has imap[i] = (first output row of its sequence, T_out, step ti inside the sequence, -)
int launch_col2im_ragged(const float* z, float* dx, long long rows_in, const int4* imap, int cin, int k, int stride, int pad, hipStream_t s, int z_bf16 = 0);
bool gemm_tn_split_supported(int M, int N, int K);
bool gemm_tn_split_writes_rm(const GemmTnSplitDesc& d);
size_t gemm_tn_split_scratch_bytes(int M, int N, int K, int nprob);
int launch_gemm_tn_split(const GemmTnSplitDesc& d, hipStream_t s);
size_t gemm_tn_scratch_bytes(int M, int N, int K);
int launch_gemm_tn(const GemmTnDesc& d, hipStream_t s);
// dW_j[n][k] = sum_m A_j[m][n] B_j[m][k] on ROW-MAJOR 16-bit operands (gemm_glds.hip's gemm_tn_tr_kernel: transposing LDS reads,
// no transposed copies): raw partial sums of `ksplit` row ranges of `kper` 64-row k-tiles into part[(j * ksplit + range)][N][K]
struct GemmTnTrDesc {
    const void* A[3];  // 16-bit rows [M][lda], 16-byte aligned
    const void* B[3];  // 16-bit rows [M][ldb]
    int nprob, M, N, K;
    long long lda, ldb;  // row pitch in VALUES, multiples of 8
    int a_split, b_split;  // the operand is a split-f16 row-major matrix ([hi8 | lo8] blocks, 4 bytes per value; f16 only): its hi halves
                           // are the plain f16 cast of the same values and are what the kernel fetches
    int bf16;
    int ksplit, kper;    // gemm_tn_tr_geometry()
    float* part;
    // conv = 1: B is the channels-last conv input [rows_in][Cin] as 16-bit rows, K = k * Cin (implicit im2col, Cin % 256 == 0); the
    // geometry or the ragged rowmap as in GemmDesc
    int conv, Cin, T_in, T_out, stride, pad;
    const int2* rowmap;
    int part_bf16 = 0;  // round 6 (bf16 operands): the split-K partial sums are written as bfloat16 slabs (same element order) - launch_splitk_reduce_bf16 folds them
};
int launch_splitk_reduce_bf16(const void* part, int ksplit, int nprob, float* const* C, int M, int N, int ldc, const float* out_scale_dev,
                              const float* scale_dev, hipStream_t s);
bool gemm_tn_tr_supported(int M, int N, int K, long long lda, long long ldb);
void gemm_tn_tr_geometry(int M, int N, int K, int nprob, int max_ranges, int& ksplit, int& kper);
int launch_gemm_tn_tr(const GemmTnTrDesc& d, hipStream_t s);
// C_j = out_scale_dev * scale_dev_j * (sum of the ksplit partial sums, in index order): gemm.hip's second split-K pass on its own
int launch_splitk_reduce(const float* part, int ksplit, int nprob, float* const* C, int M, int N, int ldc, const float* out_scale_dev,
                         const float* scale_dev, hipStream_t s);

int launch_transpose(const float* in, float* out, int rows, int cols, int ldi, int ldo, int col_off, hipStream_t s);
// out row c (pitch ldo values), column col_off + r = scale * in[r][c], written as a GEMM operand: fmt 0 = split-f16 pairs, 1 = f16, 2 = bf16
int launch_transpose_cast(const float* in, void* out, int rows, int cols, int ldi, int ldo, int col_off, float scale, int fmt, hipStream_t s);
size_t colsum_scratch_bytes(int segments, int seg_rows, int cols);
int launch_colsum(const float* in, float* out, int segments, int seg_rows, int cols, int ld, float scale, int accumulate,
                  float* scratch, size_t scratch_bytes, hipStream_t s);
// out0 = column sums of in0, out1 = column sums of in1 (both [seg_rows][ld] views of `cols` columns) in the launches of one call
int launch_colsum_pair(const float* in0, const float* in1, float* out0, float* out1, int seg_rows, int cols, int ld, float* scratch,
                       size_t scratch_bytes, hipStream_t s);
// up to 16 such pairs over <= 256 rows each in one launch (the few-sample backward's GroupNorm parameter gradients, deferred); bit-identical to launch_colsum_pair
struct ColsumPairGroupDesc {
    const float* in0[16];
    const float* in1[16];
    float* out0[16];
    float* out1[16];
    int rows[16], cols[16];
    int n;
};
int launch_colsum_pair_group(const ColsumPairGroupDesc& d, hipStream_t s);
// up to 48 single column sums out[j][c] = sum_r in[j][r * ld + c] in one launch (the arithmetic of launch_colsum without scratch: same bits)
struct ColsumJobsDesc {
    const float* in[48];
    float* out[48];
    int rows[48], cols[48], ld[48];
    int n;
};
int launch_colsum_jobs(const ColsumJobsDesc& d, hipStream_t s);

// ---- attention core (attn.hip) ---------------------------------------------------------------------------------
struct AttnDesc {
    const float *q, *k, *v;
    float* o;
    int ldq, ldk, ldv, ldo;
    int G, H, DH, Sq, Sk, inner;
    long long q_outer, q_inner, q_rs;
    long long k_outer, k_inner, k_rs;
    float scale;
    float* lse;  // optional [q rows][H]
    DropoutCfg drop;  // on the probabilities (tools/attention.py:71)
    int o_sp16;       // output as split-f16 pairs
    int in_sp16;      // q, k, v are split-f16 rows (written by a GEMM with c_sp16); not for sequences of <= 16 steps
    int in_bf16 = 0;  // round 6: q, k, v are BFLOAT16 rows (ldq / ldk / ldv in values; a GEMM with bf16 + c_f16 wrote them) - the training
                      // step's bf16 storage; check attention_in_bf16_supported().  o may then be null: only the o_cast copy is written
    int* guard;       // o_sp16, optional: range guard word (see GemmDesc::guard)
    // optional, ragged batches: group g attends q_units[g] = (first row, row stride, Sq_g, -) over k_units[g] = (first row,
    // row stride, Sk_g, -); Sq / Sk are then the LARGEST lengths (they select the kernel shape and size the LDS)
    const int4 *q_units, *k_units;
    // f32 q / k / v, but the caller's arithmetic is the split-f16 mode's: QK^T and PV may be evaluated as hi*hi + hi*lo + lo*hi
    // on f16 MFMAs (22-bit products, f32 accumulate and softmax) instead of the exact-f32 MFMA (attn_simple.hip)
    int split_math;
    // training forward, optional (round 4): the operand cast of the output that the out-projection GEMM behind this launch takes, written by
    // the attention kernel next to the f32 output (the cast launch read the f32 output back: 62 us per attention on the ragged mix).
    // o_cast_fmt 1: split-f16 pairs (rows of ldo floats) + optionally o_side = the hi halves once more as plain f16 rows (ldo halfs; the
    // backward's dW operand); 2 / 3: plain f16 / bfloat16 rows (ldo halfs).  Unscaled, bit-identical to launch_cast_sp16 / launch_cast_f16.
    // Only some shapes write it: the launch that did sets *o_cast_done (else the caller casts as before).
    // optional (round 5, the few-keys shape of attn_res.hip only - check attention_shared_keys_supported): the keys j >= k_private of EVERY
    // group are shared rows k_shared_row + (j - k_private) of k / v instead of rows of the group's own range - the object -> language
    // attention's 32 negative tokens (module/module.py:146-147), whose key / value projections are the same for every sample
    int k_private = 0;
    long long k_shared_row = 0;
    void* o_cast = nullptr;
    void* o_side = nullptr;
    int o_cast_fmt = 0;
    bool* o_cast_done = nullptr;
};
int launch_attention(const AttnDesc& d, hipStream_t s);
bool attention_in_bf16_supported(const AttnDesc& d);   // attn_simple.hip's f32-MFMA kernel on bfloat16 rows
bool attention_bf16_mfma_supported(const AttnDesc& d);  // attn_f16.hip's bf16-MFMA training kernel (the default where it applies)
bool attention_shared_keys_supported(const AttnDesc& d);  // d.k_private / k_shared_row can be honoured (the launch then takes attn_res.hip's shape)

struct AttnBwdDesc {
    const float *q, *k, *v, *o, *dout, *lse;
    float *dq, *dk, *dv, *dvec;
    int ldq, ldk, ldv, ldo;
    int ld_dq, ld_dk, ld_dv;
    int G, H, DH, Sq, Sk, inner;
    long long q_outer, q_inner, q_rs;
    long long k_outer, k_inner, k_rs;
    float scale;
    DropoutCfg drop;
    // optional, ragged batches: as AttnDesc::q_units / k_units (per-group (first row, row stride, length, -)); Sq / Sk are then
    // the LARGEST lengths
    const int4 *q_units, *k_units;
    // optional scratch of the one-pass kernel for long query ranges against <= 64 keys (object -> language): a unit's queries are
    // cut into chunks of 256, each chunk's dK / dV partial sums land here and are added in order.  part_floats >=
    // attention_bwd_part_floats(part_rows, G, H, Sk), part_rows = rows of the q matrix (the units' rows: stride 1, in unit order).
    // Without it such launches take the two-pass kernels.
    float* part = nullptr;
    size_t part_floats = 0;
    long long part_rows = 0;
    // round 6, the training step's bf16 storage (check attention_bwd_bf16_supported): q, k, v are BFLOAT16 rows (ldq / ldk / ldv in values)
    // and the gradients leave as bfloat16 rows dq16 / dk16 / dv16 (pitches ld_dq / ld_dk / ld_dv, in values); dq (f32, same pitch) is
    // scratch for units whose keys exceed one key group
    int io_bf16 = 0;
    void *dq16 = nullptr, *dk16 = nullptr, *dv16 = nullptr;
    // ... and dout is a BFLOAT16 matrix (pitch ldo, in values; ldo % 8 == 0): with io_bf16 and the bf16 products only
    // (attention_bwd_dout_bf16_enabled())
    int dout_bf16 = 0;
    int o_bf16 = 0;  // ... and o as well (with dout_bf16): the forward kept only the bfloat16 output rows
};
bool attention_bwd_dout_bf16_enabled();
int launch_attention_bwd(const AttnBwdDesc& d, hipStream_t s);
bool attention_bwd_bf16_supported(const AttnBwdDesc& d);
size_t attention_bwd_part_floats(long long q_rows, int G, int H, int Sk);

// ---- backward of the elementwise / reduction stages (bwd.hip) -------------------------------------------------
struct GroupNormBwdDesc {
    const float* x;    // pre-norm input (saved)
    const float* dy;   // gradient wrt y
    const float* dy2;  // optional second gradient stream (the x+pe side output), added to dy
    const float *gamma, *beta;
    float* dx;
    float* dgamma_part;  // [n_inst][C] per-instance partial sums
    float* dbeta_part;   // [n_inst][C]
    int n_inst, inner;
    long long outer_stride, inner_stride, tok_stride;
    int ntok, C, groups;
    float eps, slope;
    int leaky;
    DropoutCfg drop;
    // optional, ragged batches: instance i covers units[i] = (first row, row stride, token count, -) (GroupNormDesc::units);
    // ntok is then the LARGEST token count (it selects the kernel shape)
    const int4* units;
    void* dx16 = nullptr;  // optional (round 6, bf16 steps): dx once more as bfloat16 rows, same pitch (C values)
    int x_bf16 = 0;        // round 6: x is a bfloat16 matrix (the bf16 step's pre-norm rows), same pitch
    // optional (round 6): (mean, rstd) of every (instance, group) unit as the forward computed them (GroupNormDesc::stats_out) - the
    // three-pass kernel of the largest units then skips its two statistics walks over x (the register shapes recompute: x is in registers)
    const void* stats_in = nullptr;
    int dy2_bf16 = 0;  // round 6: dy2 is a bfloat16 matrix (same pitch)
};
int launch_group_norm_bwd(const GroupNormBwdDesc& d, hipStream_t s);
bool group_norm_bwd_dy2_bf16_supported(int ntok, int C, int groups);  // the shapes whose kernels read a bfloat16 dy2 (all but the 1024-thread register shape)
struct WsBwdLayer {
    const float* w;      // [cout, cin, k] original weights
    const float* dwstd;  // [cout, k*cin] gradient wrt the standardised weights (GEMM layout)
    float* dw;           // [cout, cin, k]
    int cout, cin, k;
};
int launch_ws_backward(const WsBwdLayer* layers, int n_layers, hipStream_t s);
struct HeadBwdDesc {
    const float *x, *lbar, *d_score, *d_tok;
    float* dx;          // [B, N, Tp, D]
    float* dlbar_part;  // [B*N, D]
    int B, N, Tp, D;
    const int4* units;  // optional, ragged batches: as HeadDesc::units (B*N = number of tracks, Tp = the largest T')
};
int launch_score_head_bwd(const HeadBwdDesc& d, hipStream_t s);
struct LossBwdDesc {
    const float *score_map, *score_tokens, *labels, *pos, *neg;
    long long neg_batch_stride;
    int B, N, D, n_neg;
    float pos_w, temp_scale, align_w;
    const float* g3;  // upstream gradients of {total, bce, alignment} (device, 3 floats)
    float* d_score;   // [B*N]
    float* d_tok;     // [B*N, D]
    float* coef;      // [B*N, n_neg] scratch
    float* d_neg;     // [B, n_neg, D]
    // optional, ragged batches: sample b owns the tracks trk_off[b] .. trk_off[b + 1] (device, B + 1 entries) of the concatenated
    // arrays (total_tracks of them), every mean runs over the sample's own tracks and g3 is [B][3] (one upstream triple per sample)
    const int32_t* trk_off;
    long long total_tracks;
};
int launch_loss_bwd(const LossBwdDesc& d, hipStream_t s);
// d_negw[m] = sum_b (d_lang[b, L+m] + dlbar[b] / W) (+ d_neg_align[b, m]); also used to fold dlbar into d_lang rows
// units (optional, ragged batches): sample b has units[b] = (-, L_b, first row of its text ++ negatives in d_lang, W_b)
int launch_neg_token_grad(const float* d_lang, const float* dlbar, const float* d_neg_align, float* d_negw, int B, int L,
                          int n_neg, int D, hipStream_t s, const int4* units = nullptr);
// out[seg] = sum of the rows off[seg] .. off[seg + 1] of in [*, cols] (per-sample sums over a ragged batch's tracks)
int launch_segsum_rows(const float* in, float* out, const int32_t* off, int segments, int cols, hipStream_t s);

// ---- multi-tensor gradient statistics (optim.hip) ---------------------------------------------------------------
size_t mt_sqnorm_scratch_bytes(int n, const long long* numel);
int launch_mt_sqnorm(const float* const* ptrs, const long long* numel, const int* group, int n, int n_groups, double* out,
                     void* scratch, size_t scratch_bytes, hipStream_t s);
int launch_mt_clip(float* const* ptrs, const long long* numel, int n, const double* total_sq, float max_norm, hipStream_t s);
// clip + AdamW (torch's fused arithmetic) in one multi-tensor launch over a device-side table of (param, grad, exp_avg, exp_avg_sq, step) records
size_t mt_adam_entry_bytes();
void mt_adam_entry_fill(void* host_entry, float* p, float* g, float* m, float* v, float* step, long long numel, int first_block);
int mt_adam_blocks(long long numel);
int launch_mt_clip_adamw(const void* tab_dev, int n, int blocks, double bytes, const double* total_sq, float max_norm, double lr, double beta1, double beta2,
                         double eps, double weight_decay, float step, int write_back, int* ticket, hipStream_t s);

// ---- normalisation / elementwise (norm.hip) --------------------------------------------------------------------
struct WsLayer {
    const float* w;  // [cout, cin, k]
    float* out;      // [cout, k*cin]
    int cout, cin, k;
};
int launch_ws_standardize(const WsLayer* layers, int n_layers, hipStream_t s);

struct GroupNormDesc {
    const float* x;
    float* y;
    float* y2;        // optional y + pe
    const float* pe;  // [inner, C] when y2 != null
    const float *gamma, *beta;
    int n_inst, inner;
    long long outer_stride, inner_stride, tok_stride;
    int ntok, C, groups;
    float eps, slope;
    int leaky;
    DropoutCfg drop;
    int out_sp16;  // y / y2 written as split-f16 pairs
    int* guard;    // out_sp16, optional: range guard word (see GemmDesc::guard)
    // optional, ragged batches: instance i covers the tokens units[i] = (first row, row stride, token count, pe row) instead
    // of the strided pattern above; ntok is then the LARGEST token count (it selects the kernel shape)
    const int4* units;
    int in_f16, out_f16;  // 16-bit storage mode: x / (y, y2) are _Float16 matrices (statistics stay f32); in_f16 == 2: x is a BFLOAT16 matrix (round 6)
    const float* in_scale_dev;  // optional: x is multiplied by this device scalar while it is read (an input stored in scaled units)
    // optional scratch of the sliced shape (units beyond the register shapes): 8 bytes per (unit, 256-token slice); the forward
    // orchestrators pass a piece of the caller's workspace (concurrent calls on one device must not share it); null = a
    // per-device buffer owned by the library (the stage entry point: one stream at a time)
    void* slice_ws;
    size_t slice_ws_bytes;
    // optional, f32 outputs only (training forward with 16-bit GEMM operands): besides y / y2 the kernel also writes their casts in
    // the format the NEXT GEMM takes as its operand, unscaled - cast_fmt 2: plain f16, 3: bfloat16 - so the operand cast launch (a
    // read of y and a write of the cast) disappears.  Same row pitch C as y.  (Split-f16 pairs were tried too: the pair shuffles and
    // 16 more bytes per lane cost the norm what the cast launch saved - 34.2 vs 34.3 ms per ragged step.)
    void* y_cast = nullptr;
    void* y2_cast = nullptr;
    int cast_fmt = 0;
    // optional (round 6, training forward of the largest units): where the sliced shape leaves (mean, rstd) of every (instance, group) unit
    // - [n_inst * groups] pairs - for the backward (GroupNormBwdDesc::stats_in); *stats_written (host) is set to 1 when the launch took
    // that shape and wrote them, left alone otherwise
    void* stats_out = nullptr;
    int* stats_written = nullptr;
};
int launch_group_norm(const GroupNormDesc& d, hipStream_t s);
// side16 (optional): also the plain f16 cast of the same values, contiguous [rows][K] halfs (= the hi halves)
int launch_cast_sp16(const float* in, int ld_in, float* out, int ld_out, long long rows, int K, float scale, hipStream_t s, void* side16 = nullptr);
// Same conversion with a data-dependent power-of-two scale (gradients: their magnitude is not known on the host and
// mostly below the f16 normal range).  scal[0] receives max|in| (as float bits), the cast maps it into [2^13, 2^14) and
// writes the inverse scale to scal[1] for the GEMM's out_scale_dev.  scal = 2 device floats.
// scal[0] = max(scal[0], max|in|) as float bits (the caller zeroes the slot); the first half of launch_cast_sp16_auto
int launch_amax_accumulate(const float* in, int ld_in, long long rows, int K, float* scal, hipStream_t s);
// amax into scal[0] (accumulating; the caller zeroes the pair) AND the column sums of every 64-row slab into
// part[(rows + 63) / 64][K]: one read of a gradient matrix for its scale and its bias gradients
int launch_amax_colsum(const float* in, int ld_in, long long rows, int K, float* scal, float* part, hipStream_t s);
// bf16 mode: one read leaves the unscaled row-major bf16 cast [rows][ld_out] + amax_colsum's slab sums; scal <- {2^13, 1} (scale 1)
int launch_cast_bf16_colsum(const float* in, int ld_in, void* out, int ld_out, long long rows, int K, float* scal, float* part, hipStream_t s);
// the slab column sums alone, from a bfloat16 matrix its producer already wrote (round 6; same `part` layout and scale slot)
int launch_colsum_slabs_bf16(const void* in, int ld_in, long long rows, int K, float* scal, float* part, hipStream_t s);
// the cast half of launch_cast_sp16_auto: scal[0] already holds max|in|
int launch_cast_sp16_scaled(const float* in, int ld_in, float* out, int ld_out, long long rows, int K, float* scal, hipStream_t s);
int launch_cast_sp16_auto(const float* in, int ld_in, float* out, int ld_out, long long rows, int K, float* scal, hipStream_t s);
// The same for up to 64 equally shaped matrices in three launches (the projection weights): in[i] [rows][K] -> out[i], scale pair
// of matrix i at scal + 2 * i
int launch_cast_sp16_auto_multi(const float* const* in, float* const* out, int n, int rows, int K, float* scal, hipStream_t s);
// f32 rows -> plain _Float16 rows (16-bit storage mode).  scal != null: data-dependent power-of-two scale as above (scal[1]
// receives its inverse), else the fixed `scale`.  ld_out counts halfs.
// target_exp: the largest magnitude lands in [2^target_exp, 2^(target_exp+1)); scale_out (optional, device): receives the scale itself
// bf16 = 1 (every launcher of this group): the 16-bit values are bfloat16 (round to nearest even) instead of f16 - same bytes, same
// layouts; the GEMM that consumes them gets GemmDesc::bf16
int launch_cast_f16(const float* in, int ld_in, void* out, int ld_out, long long rows, int K, float scale, float* scal, hipStream_t s,
                    int target_exp = 13, float* scale_out = nullptr, int bf16 = 0);
int launch_cast_f16_auto_multi(const float* const* in, void* const* out, int n, int rows, int K, float* scal, hipStream_t s, int bf16 = 0);
int launch_cast_f16_scaled(const float* in, int ld_in, void* out, int ld_out, long long rows, int K, float* scal, hipStream_t s, int bf16 = 0);
// transposing cast to plain f16: in [rows][cols] f32 -> out [cols][ld_out halfs] (ld_out % 128 == 0, zero-filled past rows)
int launch_cast_f16_t(const float* in, int ld_in, void* out, long long ld_out, int rows, int cols, float* scal, hipStream_t s, int bf16 = 0);
// Weight-time range check of the split-f16 activations: a GroupNorm output has E[y^2] = gamma^2 + beta^2 per channel (its
// input is normalised), so the magnitude of every tensor the norms emit is known from the weights alone.  Sets bit 1 of
// *guard when the rms of any (gamma, beta) pair lies outside [2^-6, 2^9] - where the fixed-scale split-f16 activations would
// lose bits to f16 subnormals or come near the f16 overflow.
struct NormPair { const float *gamma, *beta; int C; };
int launch_norm_range_check(const NormPair* norms, int n, int* guard, hipStream_t s);
int launch_pos_encoding(const float* gauss, int D, int t_len, int max_len, float* pe, hipStream_t s);
// lang_cat[b] = [lang[b] (L rows); neg (n_neg rows)], lbar[b] = mean over the W rows
int launch_lang_concat(const float* lang, const float* neg, float* out, float* lbar, int B, int L, int n_neg, int D,
                       hipStream_t s);
// out = [B * L text rows | n_neg negative rows] (the negative tokens once, not per sample), lbar as above
int launch_lang_concat_shared(const float* lang, const float* neg, float* out, float* lbar, int B, int L, int n_neg, int D, hipStream_t s);
// ragged: sample b has units[b] = (first input row, L_b, first output row, L_b + n_neg)
int launch_lang_concat_ragged(const float* lang, const float* neg, float* out, float* lbar, int B, const int4* units, int n_neg,
                              int D, hipStream_t s);
// dst rows list[i].x .. + list[i].z  <-  src rows list[i].y .. (row_floats floats each); n entries
int launch_gather_rows(const float* src, float* dst, const int4* list, int n, int row_floats, long long total_rows, hipStream_t s);

// ---- score head + losses (head.hip) ----------------------------------------------------------------------------
struct HeadDesc {
    const float* x;     // [B, N, Tp, D]
    const float* lbar;  // [B, D]
    float* score_map;   // [B, N]
    float* score_tokens;  // [B, N, D]
    int B, N, Tp, D;
    // optional, ragged batches: B*N = number of tracks; track j has units[j] = (first row, -, T'_j, sample index) and
    // Tp is the largest T'
    const int4* units;
};
int launch_score_head(const HeadDesc& d, hipStream_t s);
struct LossDesc {
    const float *score_map, *score_tokens, *labels, *pos, *neg;
    long long neg_batch_stride;  // 0 when the same negatives serve every sample
    int B, N, D, n_neg;
    float pos_w, temp_scale, align_w;
    float* terms;  // [B*N, 3] scratch
    float* loss3;
    int32_t* neg_argmax;  // optional
    // optional, ragged batches: sample b owns the tracks trk_off[b] .. trk_off[b + 1] (device, B + 1 entries; N = the largest
    // count) and loss3 is [B][3]: every sample's own means, i.e. what the reference computes at its batch size of 1
    const int32_t* trk_off;
};
int launch_loss(const LossDesc& d, hipStream_t s);
int launch_select(const float* score, long long n, float thr, float* prob, float* pred, hipStream_t s);

// ---- mask IoU (iou.hip) ----------------------------------------------------------------------------------------
bool launch_mask_pack_pair(const void* a_masks, int P, uint32_t* a_bits, long long* a_area, const void* b_masks, int R, uint32_t* b_bits,
                           long long* b_area, size_t area_span_bytes, int H, int W, hipStream_t s, int* status);
int launch_mask_pack(const void* masks, int elem_type, int n, int h, int w, int H, int W, uint32_t* bits,
                     long long* area, hipStream_t s);
int launch_mask_pair(const uint32_t* a_bits, const long long* a_area, int P, int T, const uint32_t* b_bits,
                     const long long* b_area, int R, const int32_t* a_frame, long long words, long long* inter,
                     long long* uni, hipStream_t s);

// ---- masklet resampling / decoding (masklet.hip) -----------------------------------------------------------------
int launch_mask_bilinear_pack(const void* masks, int elem_type, int n, int h, int w, int H, int W, uint32_t* bits,
                              long long* area, hipStream_t s);
int launch_mask_unpack(const uint32_t* bits, int n, int H, int W, void* out, int elem_type, hipStream_t s);
int launch_rle_fill_or(const uint32_t* cum, const long long* off, int n_frames, int K, int h, int w, uint8_t* out,
                       uint32_t* bits, long long* area, hipStream_t s);
