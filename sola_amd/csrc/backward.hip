// Backward orchestration of the track-selection network (autograd of module/module.py:130-162): host code sequencing
// gemm.hip (dX via the NT kernel on transposed weights / the transposed-conv gather), gemm_tn.hip (dW, db),
// attn_bwd.hip, bwd.hip over the activations saved by sola_forward_impl(train = true).  Gradients of all 83
// parameters are written to the borrowed buffers registered with sola_set_grad (overwrite semantics).
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <initializer_list>
#include <string>
#include <vector>

#include "ragged.h"

extern int g_train_split_min_rows;
int g_bwd_dual_cast = 1;  // sola_tune "bwd_dual_cast": the transposing cast of a gradient matrix also writes its row-major cast (A/B)
void sola_set_bwd_dual_cast(int v) { g_bwd_dual_cast = v; }
int g_bwd_fused_bf16_cast = 1;  // sola_tune "bwd_fused_bf16_cast": bf16 storage - the gradient statistics pass is the cast as well (A/B)
void sola_set_bwd_fused_bf16_cast(int v) { g_bwd_fused_bf16_cast = v; }

// sola_tune "train_dw_f16" (default 1, round 3): in the split-f16 training step (precision 1) the weight-gradient products
// dW = dY^T X run on PLAIN f16 casts of dY and X (one MFMA per product, f32 accumulation) instead of split pairs (three); forward and
// dX keep the split pairs.  A weight gradient is a sum over all token rows (>= 1024 here) in which the 2^-11 operand roundings
// average out: measured against the exact-f32 step (tools/train_dw_f16_errors.py) the median per-matrix error goes 5e-6 -> 1.2e-4
// (64 samples; 1.1e-5 -> 2.3e-4 at 8), the worst tensor (1.9e-3 / 4.9e-3: the forward's softmax near-ties) and the cosine of the
// whole gradient (0.999999) do not move; the step is 14 % faster on the ragged mix.  0 = split pairs everywhere.
int g_train_dw_f16 = 1;
void sola_train_set_dw_f16(int v) { g_train_dw_f16 = v; }
// sola_tune "bwd_side_rows" (round 4): exact-f32 backward of at most this many token rows (the few-sample regime; the reference trains at
// batch size 1) runs its weight-gradient products dW = dY^T X - leaves of the graph, a third of the step's kernel time there - on a side
// stream beside the dX chain.  Kernels of 64-512 blocks leave most of the chip idle, so the two streams really overlap.  0 = off.
// Same kernels, same per-gradient order: results are bit-identical (tests/test_gpu_backward.py).  OFF by default: measured at one sample
// per step (tools/train_one_probe.py, profiles/r04_train_one_sample.txt) the step is bound by the HOST's enqueue rate as much as by the GPU
// (2.2 ms of host time per 2.7 ms step), and the ~60 event records / waits of the lane cost more host time (+0.5 ms) than the overlap saves.
int g_bwd_side_rows = 0;
// sola_tune "bwd_group_rows" (round 4): an exact-f32 backward of at most this many token rows DEFERS the weight-gradient products of its
// linear layers - every dY keeps a buffer of its own - and runs them all in ONE grouped launch behind the layers (gemm_tn.hip:
// gemm_tn_f32_group_kernel; 24 products + 48 slab reductions -> 1 launch), and transposes the weights its dX GEMMs read in one launch in
// front of them (36 -> 1).  At one sample per step (the reference's batch size) that is a third of the step's launches.  0 = off.
// Same products, a different (still fixed) summation order over the rows: deterministic, not bit-identical to the slab form.
int g_bwd_group_rows = 2048;

namespace {

struct Arena {
    char* base;
    size_t total = 0;
    std::unordered_map<std::string, size_t> off, bytes;
    size_t add(const std::string& name, size_t floats) {
        off[name] = total;
        bytes[name] = floats * sizeof(float);
        total += (floats * sizeof(float) + 255) & ~(size_t)255;
        return off[name];
    }
    float* get(const std::string& name) const { return reinterpret_cast<float*>(base + off.at(name)); }
};

// row counts the arena is sized by: a uniform (B, N, T, L) batch or the concatenated rows of a ragged one
struct BwdSizes {
    size_t M, BW, R, S, inst_bt;  // layer rows, text ++ negative rows, tracks, samples, inter-object GroupNorm instances (sum of T')
    size_t rows[7];               // token rows per encoder level (0 = the object tokens)
    bool rag;
};
BwdSizes sizes_of(const Plan& p) {
    BwdSizes z{};
    z.M = p.M; z.BW = (size_t)p.B * p.W; z.R = (size_t)p.B * p.N; z.S = p.B; z.inst_bt = (size_t)p.B * p.Tp;
    z.rows[0] = z.R * p.T;
    for (int i = 0; i < 6; ++i) z.rows[i + 1] = z.R * p.Tl[i];
    z.rag = false;
    return z;
}
template <class RagLike>  // RagShape (before the forward) or RagTables (after it): the same extent fields
BwdSizes sizes_of_ragged(const RagLike& r) {
    BwdSizes z{};
    z.M = (size_t)r.Ms; z.BW = (size_t)r.LW; z.R = (size_t)r.NT; z.S = (size_t)r.S; z.inst_bt = (size_t)r.sumTpS;
    for (int j = 0; j < 7; ++j) z.rows[j] = (size_t)r.rows[j];
    z.rag = true;
    return z;
}

Arena make_arena(const SolaCtx* c, const BwdSizes& z) {
    Arena a;
    a.base = nullptr;
    const size_t D = c->cfg.lang_token_dim, H = c->cfg.num_heads, M = z.M, BW = z.BW;
    const size_t R = z.R;
    a.add("g0", M * D);
    a.add("g1", M * D);
    a.add("e", M * D);       // d(x + pe)
    a.add("dres", M * D);
    a.add("dattn", M * D);
    a.add("dqkv", M * 3 * D);
    a.add("dlkv", BW * 2 * D);
    a.add("dlang", BW * D);
    a.add("dvec", M * H);
    // object -> language attention backward in one pass: dK / dV partial sums per 256-query chunk of a sample (attn_bwd.hip); sized
    // for the most keys a text can have against the 64 the chunked launch takes (0 floats otherwise: the two-pass kernels run)
    a.add("attn_part", attention_bwd_part_floats((long long)M, (int)z.S, (int)H, 64));
    a.add("dlbar_part", R * D);
    a.add("dlbar", z.S * D);
    // GroupNorm partials are [n_inst][C]; n_inst is B*N (per track), B*T' (per time step) or B
    size_t enc_max = 0, ws_total = 0, wt_max = 3 * D * D, tn_max = 0, inst_c_max = std::max(R, z.inst_bt) * D;
    for (int i = 0; i < 6; ++i) {
        const ConvGeom& g = c->conv[i];
        enc_max = std::max(enc_max, z.rows[i + 1] * (size_t)g.cout);
        enc_max = std::max(enc_max, z.rows[i] * (size_t)g.cin);
        ws_total += (size_t)g.cout * g.cin * g.k;
        wt_max = std::max(wt_max, (size_t)g.cout * g.cin * g.k);
        tn_max = std::max(tn_max, gemm_tn_scratch_bytes((int)z.rows[i + 1], g.cout, g.k * g.cin));
        inst_c_max = std::max(inst_c_max, R * (size_t)g.cout);
    }
    tn_max = std::max(tn_max, gemm_tn_scratch_bytes((int)M, (int)D, (int)D));
    tn_max = std::max(tn_max, gemm_tn_scratch_bytes((int)BW, (int)D, (int)D));
    a.add("enc0", enc_max);
    a.add("enc1", enc_max);
    a.add("dwstd", ws_total);
    a.add("wt", wt_max);
    a.add("tn", tn_max / sizeof(float) + 64);
    a.add("gpart", inst_c_max);
    a.add("bpart", inst_c_max);
    a.add("colsum", colsum_scratch_bytes(2, (int)std::max(M, std::max(R, z.inst_bt)), (int)D) / sizeof(float) + 64);
    // few-sample regime (the reference trains at batch size 1): the dX GEMMs have fewer 64x64 tiles than the chip has CUs; scratch
    // for their deterministic two-pass split-K, as the forward has (42-136 us per dX GEMM without it, 64 blocks on 256 CUs)
    if (M <= 8192) a.add("splitk", (size_t)8192 * 4096);
    const bool lowp = c->precision >= 1 && (long long)M >= g_train_split_min_rows;
    if (!lowp && g_bwd_group_rows > 0 && M <= (size_t)g_bwd_group_rows) {  // few-sample exact-f32 backward: deferred grouped dW (see g_bwd_group_rows)
        size_t enc_keep = 0;  // + the encoder stages' dY (GroupNorm backward outputs) for the deferred conv weight gradients
        for (int i = 0; i < 5; ++i) enc_keep += z.rows[i + 1] * (size_t)c->conv[i].cout + 64;
        a.add("dwkeep", (size_t)c->cfg.n_layers * (12 * M * D + 2 * BW * D + 7 * 64) + enc_keep);  // per layer: 3 dres + 3 dqkv (3D wide) + dlkv
        a.add("gnkeep", (size_t)(3 * c->cfg.n_layers + 5) * 2 * (inst_c_max + 64));     // private (dgamma, dbeta) partials of every GroupNorm backward: one grouped column-sum launch per bucket
    }
    {   // conv dX as one GEMM z = dY W (every tap's contribution) + a col2im gather: the split-f16 / f16 modes, and every ragged batch
        // (the f32 path's transposed-conv gather needs one sequence length)
        size_t zmax = 0;
        for (int i = 1; i < 6; ++i) zmax = std::max(zmax, z.rows[i + 1] * (size_t)c->conv[i].k * c->conv[i].cin);
        const bool few = !lowp && g_bwd_group_rows > 0 && M <= (size_t)g_bwd_group_rows;  // the few-sample backward takes this form too (weights read where they lie)
        if (lowp || z.rag || few) a.add("zcol", zmax);
    }
    if (lowp) {  // split-f16 / f16-operand dX GEMMs (same size gate as the training forward): casts of dY and of the transposed weights, the data-dependent scale
        a.add("dy_sp", std::max(M, BW) * 3 * D);
        a.add("wt_sp", wt_max);
        a.add("scal", 16 + 2 * (6 + 8 * (size_t)c->cfg.n_layers));  // 16 fixed slots + one pair per stats() call of a step (6 convs, <= 8 per layer)
        {   // per-64-row-slab column sums of a gradient matrix (launch_amax_colsum): bias gradients without a second read
            size_t cp = std::max((M / 64 + 1) * 3 * D, (BW / 64 + 1) * 2 * D);
            for (int i = 0; i < 6; ++i) cp = std::max(cp, (z.rows[i + 1] / 64 + 1) * (size_t)c->conv[i].cout);
            // round 6: every statistics pass of a step keeps its own slab table (the bias sums of a gradient bucket leave in one grouped
            // launch, flush_bias): per layer 3 out-projection tables, two [.., 3D] and one [.., D] for the q / k / v gradients, one text table
            size_t all = (size_t)c->cfg.n_layers * ((M / 64 + 1) * 10 * D + (BW / 64 + 1) * 2 * D + 7 * 64);
            for (int i = 0; i < 6; ++i) all += (z.rows[i + 1] / 64 + 1) * (size_t)c->conv[i].cout + 64;
            a.add("cpart", std::max(cp, all));
        }
        // split-f16 weight gradients of the projections (gemm_tn_split.hip): transposed operands + partial sums
        if (gemm_tn_split_supported((int)M, (int)D, (int)D)) {
            size_t need = gemm_tn_split_scratch_bytes((int)M, (int)D, (int)D, 3);
            if (gemm_tn_split_supported((int)BW, (int)D, (int)D)) need = std::max(need, gemm_tn_split_scratch_bytes((int)BW, (int)D, (int)D, 2));
            for (int i = 0; i < 6; ++i)
                if (gemm_tn_split_supported((int)z.rows[i + 1], c->conv[i].cout, c->conv[i].k * c->conv[i].cin))
                    need = std::max(need, gemm_tn_split_scratch_bytes((int)z.rows[i + 1], c->conv[i].cout, c->conv[i].k * c->conv[i].cin, 1));
            a.add("tns", need / sizeof(float) + 64);
        }
    }
    return a;
}

}  // namespace

size_t sola_backward_scratch_bytes(const SolaCtx* c, const Plan& p) { return make_arena(c, sizes_of(p)).total; }

extern "C" size_t sola_backward_workspace_bytes(const SolaCtx* c, int B, int N, int T, int L) {
    if (!c || B <= 0 || N <= 0 || T <= 0 || L < 1) return 0;
    return sola_backward_scratch_bytes(c, make_plan(c, B, N, T, L, true));
}

extern "C" size_t sola_backward_ragged_workspace_bytes(const SolaCtx* c, const SolaRaggedBatch* batch) {
    if (!c || !batch) return 0;
    try {
        RagShape r;
        if (rag_shape(c, batch, r) != SOLA_OK) return 0;
        return make_arena(c, sizes_of_ragged(r)).total;
    } catch (const std::exception& e) {
        sola_set_error("backward_ragged_workspace_bytes: %s", e.what());
        return 0;
    }
}

// want_rag: the caller is sola_backward_ragged (the last forward must have been sola_forward_train_ragged, in this workspace)
static int backward_impl(SolaCtx* c, const float* d_score_map, const float* d_score_tokens, const void* fwd_workspace,
                         void* scratch, size_t scratch_bytes, void* stream_, bool want_rag) {
    SOLA_ARG(c && d_score_map && d_score_tokens && fwd_workspace && scratch, "backward: null argument");
    const Plan& p = c->last;
    if (!p.train || p.M == 0) {
        sola_set_error("backward: the last forward on this context was not sola_forward_train%s", want_rag ? "_ragged" : "");
        return SOLA_ERR_STATE;
    }
    if (p.rag != want_rag) {
        sola_set_error("backward: the last training forward was %s - call %s", p.rag ? "ragged" : "uniform", p.rag ? "sola_backward_ragged" : "sola_backward");
        return SOLA_ERR_STATE;
    }
    if (p.rag && fwd_workspace != c->last_ws) {
        sola_set_error("backward_ragged: not the workspace of the last sola_forward_train_ragged (its unit tables live there)");
        return SOLA_ERR_STATE;
    }
    const RagTables* const rt = p.rag ? &c->last_rag : nullptr;
    for (const Weight& w : c->weights)
        if (!w.grad && w.name != "positional_encoding_gaussian_matrix") {
            sola_set_error("backward: no gradient buffer registered for '%s'", w.name.c_str());
            return SOLA_ERR_WEIGHT;
        }
    Arena ar = make_arena(c, rt ? sizes_of_ragged(*rt) : sizes_of(p));
    if (scratch_bytes < ar.total) {
        sola_set_error("backward: scratch %zu bytes < required %zu", scratch_bytes, ar.total);
        return SOLA_ERR_WORKSPACE;
    }
    SOLA_ARG((reinterpret_cast<uintptr_t>(scratch) & 255) == 0, "backward: scratch must be 256-byte aligned");
    ar.base = static_cast<char*>(scratch);
    hipStream_t s = as_stream(stream_);
    const char* fbase = static_cast<const char*>(fwd_workspace);
    auto fb = [&](const std::string& name) { return reinterpret_cast<const float*>(fbase + p.bufs.at(name).off); };
    auto W = [&](const std::string& name) { return ctx_weight(c, name); };
    auto G = [&](const std::string& name) { return ctx_grad(c, name); };
    const int B = p.B, N = p.N, Tp = p.Tp, M = p.M, Wn = p.W, L = p.L;
    const int D = c->cfg.lang_token_dim, H = c->cfg.num_heads, DH = D / H;
    const int R = rt ? rt->NT : B * N;                    // tracks
    const int BW = rt ? (int)rt->LW : B * Wn;             // text ++ negative rows
    const int n_bt = rt ? rt->sumTpS : B * Tp;            // inter-object units (one per sample and encoded step)
    const int max_rows_smp = rt ? rt->maxRowsSample : N * Tp;
    const float scale = 1.0f / sqrtf((float)DH);
    float* tn = ar.get("tn");
    const size_t tn_bytes = ar.total - ar.off.at("tn");  // upper bound; launch_gemm_tn checks its own need
    float* wt = ar.get("wt");
    float* const splitk_ws = ar.off.count("splitk") ? ar.get("splitk") : nullptr;
    const size_t splitk_bytes = splitk_ws ? (size_t)8192 * 4096 * sizeof(float) : 0;
    // split: every GEMM of the backward on f16 MFMAs - (hi, lo) operand pairs, three products (precision 1), or plain f16
    // operands, one product (precision 2: "pure"; mixed precision - activations, gradients and accumulation stay f32)
    const bool split = c->precision >= 1 && ar.off.count("dy_sp") != 0;
    const bool pure = c->precision >= 2;
    const int bf = c->precision == 3 ? 1 : 0;  // bfloat16 GEMM operands
    const int lowp_arith = pure ? 2 : 1;
    // sola_tune "train_dw_f16": in the split-f16 step the weight-gradient products dW = dY^T X take PLAIN f16 operands (one MFMA per
    // product; a sum over all token rows, where the operand rounding averages out) while forward and dX keep the split pairs
    const bool dw16 = !pure && g_train_dw_f16 != 0;
    auto cast_scaled = [&](const float* in, int ld, float* out, long long rows, int K, float* scal) -> int {
        return pure ? launch_cast_f16_scaled(in, ld, out, K, rows, K, scal, s, bf) : launch_cast_sp16_scaled(in, ld, out, K, rows, K, scal, s);
    };
    auto cast_auto = [&](const float* in, int ld, float* out, long long rows, int K, float* scal) -> int {
        return pure ? launch_cast_f16(in, ld, out, K, rows, K, 0.f, scal, s, 13, nullptr, bf) : launch_cast_sp16_auto(in, ld, out, K, rows, K, scal, s);
    };
    auto cast_fixed = [&](const float* in, int ld, float* out, long long rows, int K, float scale) -> int {
        return pure ? launch_cast_f16(in, ld, out, K, rows, K, scale, nullptr, s, 13, nullptr, bf) : launch_cast_sp16(in, ld, out, K, rows, K, scale, s);
    };

    // ---- side lane of the few-sample exact-f32 backward: dW products on c->side_stream.  dw_begin(): the side stream waits for everything
    //      the main stream has enqueued (the dY it reads); dw_end(slot): marks the side stream's progress for the buffer class `slot`
    //      (0 = dres, 1 = dqkv / dlkv, 2 = the encoder's dy, 3 = everything); wait_side(slot): the main stream waits for that mark before
    //      it overwrites the buffer.  `tn` (the dW kernels' slab scratch) is touched by the side stream only.
    const bool lane_on = !split && g_bwd_side_rows > 0 && M <= g_bwd_side_rows;
    if (lane_on && !c->side_stream) {
        SOLA_HIP(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        SOLA_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        for (hipEvent_t& e : c->ev_side) SOLA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    hipStream_t s2 = lane_on ? c->side_stream : s;
    // ---- few-sample exact-f32 backward: deferred, grouped weight gradients of the linear layers + one grouped weight transposition
    //      (g_bwd_group_rows).  keep(): the next private buffer for a gradient matrix a deferred product reads at the end.
    const bool group = !split && !lane_on && ar.off.count("dwkeep") != 0;
    float* keep_ptr = group ? ar.get("dwkeep") : nullptr;
    auto keep = [&](size_t floats) { float* q = keep_ptr; keep_ptr += (floats + 63) & ~(size_t)63; return q; };
    GemmTnGroupDesc gq{};
    auto flush_group = [&]() -> int {
        if (gq.nprob == 0) return SOLA_OK;
        SOLA_TRY(launch_gemm_tn_group(gq, s));
        gq.nprob = 0;
        return SOLA_OK;
    };
    // few-sample backward, round 5: the dX GEMMs read the weight matrices in their own row-major layout (GemmDesc::w_nn, the few-row
    // kernel's NN form: the same products in the same order as on a transposed copy) - the step's 49 weight transpositions (one grouped
    // launch, 264 MB of traffic) are gone.  transpose_into() records the matrices of the next product, grad_x() hands them over.
    const float* nn_w[3] = {nullptr, nullptr, nullptr};
    int nn_rows = 0;
    bool side_pending[4] = {false, false, false, false};
    auto dw_begin = [&]() -> int {
        if (!lane_on) return SOLA_OK;
        SOLA_HIP(hipEventRecord(c->ev_fork, s));
        SOLA_HIP(hipStreamWaitEvent(s2, c->ev_fork, 0));
        return SOLA_OK;
    };
    auto dw_end = [&](int slot) -> int {
        if (!lane_on) return SOLA_OK;
        SOLA_HIP(hipEventRecord(c->ev_side[slot], s2));
        side_pending[slot] = true;
        return SOLA_OK;
    };
    auto wait_side = [&](int slot) -> int {
        if (!lane_on || !side_pending[slot]) return SOLA_OK;
        SOLA_HIP(hipStreamWaitEvent(s, c->ev_side[slot], 0));
        side_pending[slot] = false;
        return SOLA_OK;
    };
    auto join_side = [&]() -> int {  // every gradient the side stream has been given is final for the main stream's next operation
        if (!lane_on) return SOLA_OK;
        SOLA_TRY(dw_end(3));
        SOLA_TRY(wait_side(3));
        side_pending[0] = side_pending[1] = side_pending[2] = false;
        return SOLA_OK;
    };

    // ---- helpers ----------------------------------------------------------------------------------------------
    // dW[N_out, K_in] = dY^T X, db = colsum(dY)
    auto grad_w = [&](const float* dY, int ldy, const float* X, int ldx, int rows, int n_out, int k_in, float* dW, float* db) -> int {
        GemmTnDesc d{};
        d.A = dY; d.B = X; d.C = dW; d.bias_grad = db; d.M = rows; d.N = n_out; d.K = k_in; d.lda = ldy; d.ldb = ldx;
        d.scratch = tn; d.scratch_bytes = tn_bytes;
        return launch_gemm_tn(d, s2);
    };
    // up to three weight gradients sharing rows / sizes / pitches (q, k, v of one attention): one launch on the split-f16
    // path (gemm_tn_split.hip) + the bias gradients as column sums, or the f32 kernel per problem
    // Split mode reads every gradient matrix ONCE for its scale (max|dY|, shared by the dW and dX GEMMs that consume it) and
    // its bias gradients (per-slab column sums, folded by a tiny second pass): stats() -> scale slot, bias_from_stats().
    float* cpart = split ? ar.get("cpart") : nullptr;  // the CURRENT statistics pass's slab table (bumped per pass: they all stay until flush_bias)
    float* cpart_next = cpart;
    const size_t cpart_bytes = split ? ar.bytes.at("cpart") : 0;
    int cpart_cols = 0, cpart_slabs = 0;
    ColsumJobsDesc bias_q{};  // deferred bias gradients of the current bucket
    int stat_calls = 0;
    const int scal_floats = split ? (int)(16 + 2 * (6 + 8 * (size_t)c->cfg.n_layers)) : 0;
    if (split) SOLA_HIP(hipMemsetAsync(ar.get("scal"), 0, scal_floats * sizeof(float), s));
    // bf16 storage: the pass is the CAST as well (launch_cast_bf16_colsum: bfloat16 needs no scale, so no max|x| in front of the
    // cast) - "dy_sp" then holds the row-major bf16 copy of [rows][cols] at pitch ld, and the dW / dX GEMMs of the matrix skip theirs
    const float* cast_src = nullptr;  // the matrix whose cast "dy_sp" holds (fused pass), its pitch and columns
    int cast_ld = 0, cast_cols = 0;
    const size_t dy_sp_halfs = split ? 2 * (size_t)std::max(M, BW) * 3 * D : 0;
    // round 6 (SolaCtx::qkv16): a site whose attention backward wrote dq / dk / dv as bfloat16 rows itself.  [M][3D] lands at the start of
    // "dy_sp" - exactly where the fused statistics pass would have put its cast, so everything downstream of it is unchanged - and the
    // object -> language site's [BW][2D] key / value gradients in the buffer's second half (kv_*: their own record, the dX GEMM of dq
    // still reads the first half when they are consumed)
    extern int g_train_bf16_store;
    const bool gn16 = split && pure && bf && g_train_bf16_store && g_bwd_fused_bf16_cast && g_bwd_dual_cast && D % 8 == 0;  // norms write their bf16 dx
    unsigned short* const dy16 = split ? reinterpret_cast<unsigned short*>(ar.get("dy_sp")) : nullptr;
    unsigned short* const kv16 = split ? dy16 + dy_sp_halfs / 2 : nullptr;
    const float* kv_src = nullptr;  // the [BW][2D] matrix whose bf16 rows kv16 holds
    // ready16: the producer already wrote dY's bfloat16 rows (same pitch) there - only the slab column sums are taken, from the 2-byte rows
    auto stats = [&](const float* dY, int ld, int rows, int cols, int slot, float** sc_out, bool may_cast = true, const void* ready16 = nullptr) -> int {
        *sc_out = nullptr;
        if (may_cast) cast_src = nullptr;
        if (!split || cols % 4 || ld % 4) return SOLA_OK;
        cpart = cpart_next;  // this pass's own table
        cpart_next += (((size_t)((rows + 63) / 64) * cols) + 63) & ~(size_t)63;
        SOLA_ARG((size_t)(cpart_next - ar.get("cpart")) * sizeof(float) <= cpart_bytes, "backward: the slab tables of the step's statistics passes exceed their arena");
        if (ready16) {
            (void)slot;
            SOLA_ARG(16 + 2 * (stat_calls + 1) <= scal_floats, "backward: more gradient-statistics passes (%d) than scale slots", stat_calls + 1);
            float* sc = ar.get("scal") + 16 + 2 * stat_calls++;
            SOLA_TRY(launch_colsum_slabs_bf16(ready16, ld, rows, cols, sc, cpart, s));
            if (ready16 == dy16) { cast_src = dY; cast_ld = ld; cast_cols = cols; }
            else kv_src = dY;
            cpart_cols = cols;
            cpart_slabs = (rows + 63) / 64;
            *sc_out = sc;
            return SOLA_OK;
        }
        // every call takes the next pair of the slots behind the 16 fixed ones: they were zeroed by ONE memset when the call began (a
        // memset of 8 bytes per gradient matrix was 20 launches of ~5 us per step)
        (void)slot;
        SOLA_ARG(16 + 2 * (stat_calls + 1) <= scal_floats, "backward: more gradient-statistics passes (%d) than scale slots", stat_calls + 1);
        float* sc = ar.get("scal") + 16 + 2 * stat_calls++;
        if (may_cast && pure && bf && g_bwd_fused_bf16_cast && g_bwd_dual_cast && ld % 8 == 0 && (size_t)rows * ld <= dy_sp_halfs) {
            SOLA_TRY(launch_cast_bf16_colsum(dY, ld, ar.get("dy_sp"), ld, rows, cols, sc, cpart, s));
            cast_src = dY; cast_ld = ld; cast_cols = cols;
        } else {
            SOLA_TRY(launch_amax_colsum(dY, ld, rows, cols, sc, cpart, s));
        }
        cpart_cols = cols;
        cpart_slabs = (rows + 63) / 64;
        *sc_out = sc;
        return SOLA_OK;
    };
    auto flush_bias = [&]() -> int {
        if (bias_q.n == 0) return SOLA_OK;
        SOLA_TRY(launch_colsum_jobs(bias_q, s));
        bias_q.n = 0;
        return SOLA_OK;
    };
    // deferred (round 6): the sums of a bucket's bias gradients leave in ONE launch in front of the bucket's event (flush_bias) - the same
    // per-column arithmetic as the per-gradient launch_colsum they replace
    auto bias_from_stats = [&](int col_off, int ncols, float* db) -> int {
        if (bias_q.n == 48) SOLA_TRY(flush_bias());
        const int e = bias_q.n++;
        bias_q.in[e] = cpart + col_off; bias_q.out[e] = db; bias_q.rows[e] = cpart_slabs; bias_q.cols[e] = ncols; bias_q.ld[e] = cpart_cols;
        return SOLA_OK;
    };
    struct WG { const float* dY; const float* X; float* dW; float* db; };
    // sc: scale slot from stats() over a matrix containing every dY of the call; db_done: the bias gradients were taken from it
    // dy_rm_done (optional, out): the call also left the ROW-MAJOR cast of the whole [rows][ldy] gradient matrix behind g[0].dY in
    // "dy_sp" (same scale): the dX GEMMs of the same matrix then skip their own cast (grad_x's `cast_done`)
    auto grad_w_many = [&](const WG* g, int n, int ldy, int ldx, int rows, int n_out, int k_in, float* sc = nullptr, bool db_done = false,
                           bool* dy_rm_done = nullptr) -> int {
        if (dy_rm_done) *dy_rm_done = false;
        if (group && n_out == D && k_in == D) {  // deferred: one grouped launch behind the layers (the dY buffers are private: keep())
            for (int j = 0; j < n; ++j) {
                if (gq.nprob == 32) SOLA_TRY(flush_group());
                GemmTnGroupDesc::Prob& q = gq.p[gq.nprob++];
                q = GemmTnGroupDesc::Prob{};
                q.A = g[j].dY; q.B = g[j].X; q.C = g[j].dW; q.bias_grad = db_done ? nullptr : g[j].db;
                q.M = rows; q.N = n_out; q.K = k_in; q.lda = ldy; q.ldb = ldx;
            }
            return SOLA_OK;
        }
        if (split && ar.off.count("tns") && gemm_tn_split_supported(rows, n_out, k_in)) {
            GemmTnSplitDesc d{};
            d.scal = sc; d.pure = pure ? 1 + bf : (dw16 ? 1 : 0); d.rm_split = dw16 ? 1 : 0;
            if (dy_rm_done && sc && g_bwd_dual_cast) {
                d.a_rm = ar.get("dy_sp"); d.a_rm_ld = ldy;
                d.a_rm_ready = cast_src && cast_src == g[0].dY && cast_ld == ldy && (g[n - 1].dY - g[0].dY) + n_out <= cast_cols;
            }
            if (kv_src && kv_src == g[0].dY && sc) {  // the attention backward's own bf16 rows of the key / value gradients
                d.a_rm = reinterpret_cast<float*>(kv16); d.a_rm_ld = ldy; d.a_rm_ready = 1;
            }
            d.nprob = n; d.M = rows; d.N = n_out; d.K = k_in; d.lda = ldy; d.ldb = ldx;
            for (int j = 0; j < n; ++j) {
                d.A[j] = g[j].dY; d.B[j] = g[j].X; d.C[j] = g[j].dW;
                if ((pure || dw16) && ldx == k_in) d.B16[j] = c->x16_find(g[j].X, k_in, pure ? 1 + bf : 1);
            }
            d.scratch = ar.get("tns"); d.scratch_bytes = ar.total - ar.off.at("tns");
            if (d.scratch_bytes >= gemm_tn_split_scratch_bytes(rows, n_out, k_in, n)) {
                if (dy_rm_done) *dy_rm_done = gemm_tn_split_writes_rm(d);
                SOLA_TRY(launch_gemm_tn_split(d, s));
                float* cs = ar.get("colsum");
                const size_t csb = ar.total - ar.off.at("colsum");  // upper bound; launch_colsum uses what it needs
                for (int j = 0; j < n && !db_done; ++j)
                    if (g[j].db) SOLA_TRY(launch_colsum(g[j].dY, g[j].db, 1, rows, n_out, ldy, 1.f, 0, cs, csb, s));
                return SOLA_OK;
            }
        }
        for (int j = 0; j < n; ++j) SOLA_TRY(grad_w(g[j].dY, ldy, g[j].X, ldx, rows, n_out, k_in, g[j].dW, g[j].db));
        return SOLA_OK;
    };
    bool wt_sp_ready = false;  // "wt_sp" already holds the cast of the transposed weights the next grad_x call needs (transpose_into)
    // dX[rows, k_in] = dY[rows, n_cat] * Wcat (+ R), where wt holds Wcat^T as [k_in][n_cat]
    // cast_done: "dy_sp" already holds the row-major cast of the [rows][ldy] matrix that dY - col_off starts (grad_w_many)
    auto grad_x = [&](const float* dY, int ldy, int rows, int n_cat, int k_in, const float* Radd, float* dX, float* sc = nullptr,
                      bool cast_done = false, int col_off = 0, bool c16 = false) -> int {
        GemmDesc d{};
        d.nprob = 1;
        d.p[0] = GemmProblem{dY, wt, nullptr, Radd, dX};
        d.M = rows; d.N = k_in; d.K = n_cat; d.lda = ldy; d.ldr = k_in; d.ldc = k_in;
        d.splitk_ws = splitk_ws; d.splitk_bytes = splitk_bytes;
        if (group) {
            SOLA_ARG(nn_rows > 0 && n_cat % nn_rows == 0 && n_cat / nn_rows <= 3, "backward: dX product over %d columns of %d-row weight matrices", n_cat, nn_rows);
            for (int j = 0; j < n_cat / nn_rows; ++j) d.w_nn[j] = nn_w[j];
            d.w_nn_rows = nn_rows;
            d.p[0].W = nullptr;
            if (!gemm_nn_supported(d)) {  // outside the few-row shape: the transposed copy, as the many-row path
                d.w_nn_rows = 0;
                for (int j = 0; j < n_cat / nn_rows; ++j) SOLA_TRY(launch_transpose(nn_w[j], wt, nn_rows, k_in, k_in, n_cat, j * nn_rows, s));
                d.p[0].W = wt;
            }
        }
        if (split && n_cat % (pure ? 64 : 32) == 0 && ldy % 4 == 0) {
            // dY is cast with a data-dependent power-of-two scale (gradients sit mostly below the f16 normal range), the
            // transposed weights with the fixed 2^6; the epilogue undoes both
            float* scal = sc ? sc : ar.get("scal");
            if (kv_src && kv_src == dY - col_off && sc) {
                d.p[0].A = reinterpret_cast<const float*>(kv16 + col_off);
                d.lda = ldy;
            } else if (cast_done && sc) {
                d.p[0].A = pure ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(ar.get("dy_sp")) + col_off) : ar.get("dy_sp") + col_off;
                d.lda = ldy;
            } else {
                cast_src = nullptr;  // "dy_sp" is rewritten
                if (sc) SOLA_TRY(cast_scaled(dY, ldy, ar.get("dy_sp"), rows, n_cat, scal));
                else SOLA_TRY(cast_auto(dY, ldy, ar.get("dy_sp"), rows, n_cat, scal));
                d.p[0].A = ar.get("dy_sp");
                d.lda = n_cat;
            }
            if (!wt_sp_ready) SOLA_TRY(cast_fixed(wt, n_cat, ar.get("wt_sp"), k_in, n_cat, kLinScale));  // else transpose_into wrote the operand itself
            wt_sp_ready = false;
            d.p[0].W = ar.get("wt_sp");
            d.arith = lowp_arith; d.bf16 = bf; d.out_scale = 1.f / kLinScale; d.out_scale_dev = scal + 1;
            if (c16) d.c_f16 = 1;  // dX as bfloat16 rows (pitch k_in values)
        } else {
            SOLA_ARG(!c16, "backward: a bf16 input gradient needs the 16-bit GEMM");
        }
        return launch_gemm(d, s);
    };
    auto transpose_into = [&](const float* w, int n_out, int k_in, int n_cat, int col_off) -> int {
        if (group) {  // no copy: grad_x reads the matrix where it lies
            SOLA_ARG(col_off % n_out == 0 && col_off / n_out < 3, "backward: weight block at column %d of %d-row matrices", col_off, n_out);
            nn_w[col_off / n_out] = w;
            nn_rows = n_out;
            return SOLA_OK;
        }
        if (split && n_cat % (pure ? 64 : 32) == 0) {  // grad_x's condition for the reduced-precision GEMM: the operand is written directly
            wt_sp_ready = true;
            return launch_transpose_cast(w, ar.get("wt_sp"), n_out, k_in, k_in, n_cat, col_off, kLinScale, pure ? 1 + bf : 0, s);
        }
        return launch_transpose(w, wt, n_out, k_in, k_in, n_cat, col_off, s);  // wt[k][col_off + n] = w[n][k]
    };
    ColsumPairGroupDesc gn_q{};
    float* gn_keep = (group && ar.off.count("gnkeep")) ? ar.get("gnkeep") : nullptr;
    // dx16 (round 6, bf16 steps): the norm's input gradient once more as bfloat16 rows - the consumer GEMMs' operand, written here instead of
    // by a statistics-and-cast pass that read the f32 matrix back (launch_cast_bf16_colsum); the f32 rows stay (residual stream)
    auto gn_bwd = [&](const float* xpre, const float* dy, const float* dy2, const std::string& wname, float* dx, int n_inst,
                      int inner, long long outer, long long inner_stride, long long tok_stride, int ntok, int C, int groups,
                      int leaky, const DropoutCfg* drop, const int4* units = nullptr, void* dx16 = nullptr, bool x_bf16 = false,
                      const void* stats_in = nullptr, bool dy2_16 = false) -> int {
        GroupNormBwdDesc d{};
        d.stats_in = stats_in;
        d.dy2_bf16 = dy2_16 ? 1 : 0;
        if (drop) d.drop = *drop;
        d.units = units;
        d.dx16 = dx16;
        d.x_bf16 = x_bf16 ? 1 : 0;
        d.x = xpre; d.dy = dy; d.dy2 = dy2; d.gamma = W(wname + ".weight"); d.beta = W(wname + ".bias"); d.dx = dx;
        d.dgamma_part = ar.get("gpart"); d.dbeta_part = ar.get("bpart");
        d.n_inst = n_inst; d.inner = inner; d.outer_stride = outer; d.inner_stride = inner_stride; d.tok_stride = tok_stride;
        d.ntok = ntok; d.C = C; d.groups = groups; d.eps = 1e-5f; d.slope = 0.01f; d.leaky = leaky;
        SOLA_TRY(wait_side(0));  // a GroupNorm backward writes dres (layers) or the encoder's next dy: the side stream's readers are done
        SOLA_TRY(wait_side(2));
        // round 5, few-sample backward: private partial buffers; the column sums of all norms of a bucket leave in ONE launch (flush_gn)
        const bool defer = group && n_inst <= 256 && gn_q.n < 16 && ar.off.count("gnkeep") != 0;
        if (defer) {
            d.dgamma_part = gn_keep; gn_keep += ((size_t)n_inst * C + 63) & ~(size_t)63;
            d.dbeta_part = gn_keep; gn_keep += ((size_t)n_inst * C + 63) & ~(size_t)63;
        }
        SOLA_TRY(launch_group_norm_bwd(d, s));
        if (defer) {
            const int e = gn_q.n++;
            gn_q.in0[e] = d.dgamma_part; gn_q.in1[e] = d.dbeta_part; gn_q.out0[e] = G(wname + ".weight"); gn_q.out1[e] = G(wname + ".bias");
            gn_q.rows[e] = n_inst; gn_q.cols[e] = C;
            return SOLA_OK;
        }
        float* cs = ar.get("colsum");
        const size_t csb = ar.total - ar.off.at("colsum");
        return launch_colsum_pair(ar.get("gpart"), ar.get("bpart"), G(wname + ".weight"), G(wname + ".bias"), n_inst, C, C, cs, csb, s);
    };
    auto flush_gn = [&]() -> int {
        if (gn_q.n == 0) return SOLA_OK;
        SOLA_TRY(launch_colsum_pair_group(gn_q, s));
        gn_q.n = 0;
        return SOLA_OK;
    };

    // ---- score head -------------------------------------------------------------------------------------------
    const std::string last = "l" + std::to_string(c->cfg.n_layers - 1) + "_o2l";
    float* gbuf[2] = {ar.get("g0"), ar.get("g1")};
    int cur = 0;
    // gradient-bucket events (ctx.h): created once, recorded as each bucket's last gradient kernel is enqueued
    if ((int)c->bucket_ev.size() != c->n_buckets()) {
        for (hipEvent_t e : c->bucket_ev) (void)hipEventDestroy(e);
        c->bucket_ev.assign(c->n_buckets(), nullptr);
        for (hipEvent_t& e : c->bucket_ev) SOLA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    c->bucket_recorded = false;
    {
        HeadBwdDesc d{fb(last), fb("lbar"), d_score_map, d_score_tokens, gbuf[cur], ar.get("dlbar_part"), B, N, Tp, D};
        if (rt) { d.B = 1; d.N = R; d.units = rt->u_strk; }
        SOLA_TRY(launch_score_head_bwd(d, s));
        if (rt) SOLA_TRY(launch_segsum_rows(ar.get("dlbar_part"), ar.get("dlbar"), rt->trk_off, B, D, s));
        else SOLA_TRY(launch_colsum(ar.get("dlbar_part"), ar.get("dlbar"), B, N, D, D, 1.f, 0, nullptr, 0, s));
    }
    float* dres = ar.get("dres");
    float* dattn = ar.get("dattn");
    float* dqkv = ar.get("dqkv");
    float* dlkv = ar.get("dlkv");
    float* dlang = ar.get("dlang");
    float* dvec = ar.get("dvec");
    float* egrad = ar.get("e");
    bool egrad_bf16 = false;  // the motion sub-block's backward left d(x_obj + pe) in `egrad` as bfloat16 rows (bf16 steps)
    bool dlang_init = false;

    for (int l = c->cfg.n_layers - 1; l >= 0; --l) {
        const std::string lp = "object_lang_align_layers." + std::to_string(l) + ".";
        const std::string ls = "l" + std::to_string(l);
        auto ab = [&](int a, const char* what) { return fb(abuf(true, l, kAttnShort[a], what)); };
        const float* xin = l == 0 ? fb("conv5") : fb("l" + std::to_string(l - 1) + "_o2l");
        // common tail of every sub-block: out_proj backward  (res = resid + attn * Wo^T + bo)
        // g16 (bf16 steps, train_bf16_store 3): d(attention output) leaves the GEMM as bfloat16 rows - the attention backward reads them
        auto out_proj_bwd = [&](int a, bool g16 = false) -> int {
            const std::string an = lp + kAttnLong[a];
            float* sc;
            SOLA_TRY(stats(dres, D, M, D, 2, &sc, true, gn16 ? dy16 : nullptr));  // gn16: the norm's backward in front of this wrote the bf16 rows
            if (sc) SOLA_TRY(bias_from_stats(0, D, G(an + ".out_proj.bias")));
            const WG wo[1] = {{dres, ab(a, "attn"), G(an + ".out_proj.weight"), G(an + ".out_proj.bias")}};
            bool rm;
            SOLA_TRY(dw_begin());
            SOLA_TRY(grad_w_many(wo, 1, D, D, M, D, D, sc, sc != nullptr, &rm));
            SOLA_TRY(dw_end(0));
            SOLA_TRY(transpose_into(W(an + ".out_proj.weight"), D, D, D, 0));
            return grad_x(dres, D, M, D, D, nullptr, dattn, sc, rm, 0, g16);
        };
        auto site_g16 = [&](int a) -> bool {  // the site's q / k / v are bf16 rows, its backward kernels take a bf16 dO, and the GEMM can write it
            return split && pure && bf && g_train_bf16_store >= 3 && (size_t)l * 3 + a < c->qkv16.size() && c->qkv16[(size_t)l * 3 + a] &&
                   attention_bwd_dout_bf16_enabled() && D % 8 == 0 && D % (pure ? 64 : 32) == 0;
        };

        // (iii) object -> language: x_o2l = GN2(x_mot + attn(q(x_mot), k(lang), v(lang)) Wo)
        {
            const std::string an = lp + "object2lang_attn";
            if (group) { dres = keep((size_t)M * D); dqkv = keep((size_t)M * 3 * D); dlkv = keep((size_t)BW * 2 * D); }
            SOLA_TRY(gn_bwd(ab(2, "res"), gbuf[cur], nullptr, lp + "norm.2", dres, B, 1, (long long)N * Tp, 0, 1, max_rows_smp, D,
                            c->cfg.n_groups_module, 0, nullptr, rt ? rt->u_smp : nullptr, gn16 ? dy16 : nullptr, (size_t)l * 3 + 2 < c->res16.size() && c->res16[(size_t)l * 3 + 2],
                            ((size_t)l < c->gn2_stats.size() && c->gn2_stats[l]) ? fb("l" + std::to_string(l) + "_gn2st") : nullptr));
            const bool g16_2 = site_g16(2);
            SOLA_ARG(g16_2 || !((size_t)l * 3 + 2 < c->attn_o16.size() && c->attn_o16[(size_t)l * 3 + 2]),
                     "backward: attention output %d of layer %d exists as bf16 rows only, but this backward takes the f32 rows (a switch changed between the forward and the backward)", 2, l);
            SOLA_TRY(out_proj_bwd(2, g16_2));
            AttnBwdDesc ad{ab(2, "q"), ab(2, "lk"), ab(2, "lv"), ab(2, "attn"), dattn, ab(2, "lse"),
                           dqkv, dlkv, dlkv + D, dvec, D, D, D, D, 3 * D, 2 * D, 2 * D,
                           B, H, DH, max_rows_smp, Wn, 1, (long long)N * Tp, 0, 1, (long long)Wn, 0, 1, scale};
            if (rt) { ad.q_units = rt->u_smp; ad.k_units = rt->u_langk; }
            ad.drop = c->attn_drop(l, 2);
            // a sample's N * T' query rows are consecutive and the samples follow each other: the chunked one-pass launch applies
            ad.part = ar.get("attn_part");
            ad.part_floats = attention_bwd_part_floats((long long)M, B, H, 64);
            ad.part_rows = (long long)M;
            const bool s16 = split && (size_t)l * 3 + 2 < c->qkv16.size() && c->qkv16[(size_t)l * 3 + 2];
            if (s16) { ad.io_bf16 = 1; ad.dq16 = dy16; ad.dk16 = kv16; ad.dv16 = kv16 + D; kv_src = nullptr; cast_src = nullptr; ad.dout_bf16 = g16_2 ? 1 : 0;
                if (g16_2 && (size_t)l * 3 + 2 < c->attn_o16.size() && c->attn_o16[(size_t)l * 3 + 2]) {
                    ad.o = static_cast<const float*>(c->x16_find(ab(2, "attn"), D, 2));
                    ad.o_bf16 = 1;
                    SOLA_ARG(ad.o, "backward: the forward kept only the bf16 rows of attention output %d of layer %d, and they are not in the arena", 2, l);
                } }
            SOLA_TRY(wait_side(1));  // dqkv / dlkv: the previous sub-block's weight gradients have read them
            SOLA_TRY(launch_attention_bwd(ad, s));
            const float* x_mot = fb(ls + "_motion");
            float *scq, *sckv;
            SOLA_TRY(stats(dqkv, 3 * D, M, D, 4, &scq, true, s16 ? dy16 : nullptr));
            if (scq) SOLA_TRY(bias_from_stats(0, D, G(an + ".q_proj.bias")));
            const WG wq[1] = {{dqkv, x_mot, G(an + ".q_proj.weight"), G(an + ".q_proj.bias")}};
            bool rmq;
            SOLA_TRY(dw_begin());
            SOLA_TRY(grad_w_many(wq, 1, 3 * D, D, M, D, D, scq, scq != nullptr, &rmq));
            SOLA_TRY(stats(dlkv, 2 * D, BW, 2 * D, 6, &sckv, false, s16 ? kv16 : nullptr));  // no fused cast: "dy_sp" holds dq's for the dX GEMM below
            if (sckv) {
                SOLA_TRY(bias_from_stats(0, D, G(an + ".k_proj.bias")));
                SOLA_TRY(bias_from_stats(D, D, G(an + ".v_proj.bias")));
            }
            const WG wkv[2] = {{dlkv, fb("lang"), G(an + ".k_proj.weight"), G(an + ".k_proj.bias")},
                               {dlkv + D, fb("lang"), G(an + ".v_proj.weight"), G(an + ".v_proj.bias")}};
            SOLA_TRY(grad_w_many(wkv, 2, 2 * D, D, BW, D, D, sckv, sckv != nullptr));
            SOLA_TRY(dw_end(1));
            SOLA_TRY(transpose_into(W(an + ".q_proj.weight"), D, D, D, 0));
            SOLA_TRY(grad_x(dqkv, 3 * D, M, D, D, dres, gbuf[1 - cur], scq, rmq));  // d x_mot = dres + dq Wq
            SOLA_TRY(transpose_into(W(an + ".k_proj.weight"), D, D, 2 * D, 0));
            SOLA_TRY(transpose_into(W(an + ".v_proj.weight"), D, D, 2 * D, D));
            SOLA_TRY(grad_x(dlkv, 2 * D, BW, 2 * D, D, dlang_init ? dlang : nullptr, dlang, sckv));  // accumulate over layers
            kv_src = nullptr;
            dlang_init = true;
            cur = 1 - cur;
        }
        // (ii) motion: x_mot = GN1(x_obj + attn(q(x_obj+pe), k(x_obj+pe), v(x_obj)) Wo)
        {
            const std::string an = lp + "motion_attn";
            if (group) { dres = keep((size_t)M * D); dqkv = keep((size_t)M * 3 * D); }
            SOLA_TRY(gn_bwd(ab(1, "res"), gbuf[cur], nullptr, lp + "norm.1", dres, R, 1, Tp, 0, 1, Tp, D,
                            c->cfg.n_groups_module, 0, nullptr, rt ? rt->u_strk : nullptr, gn16 ? dy16 : nullptr, (size_t)l * 3 + 1 < c->res16.size() && c->res16[(size_t)l * 3 + 1]));
            const bool g16_1 = site_g16(1);
            SOLA_ARG(g16_1 || !((size_t)l * 3 + 1 < c->attn_o16.size() && c->attn_o16[(size_t)l * 3 + 1]),
                     "backward: attention output %d of layer %d exists as bf16 rows only, but this backward takes the f32 rows (a switch changed between the forward and the backward)", 1, l);
            SOLA_TRY(out_proj_bwd(1, g16_1));
            AttnBwdDesc ad{ab(1, "q"), ab(1, "k"), ab(1, "v"), ab(1, "attn"), dattn, ab(1, "lse"),
                           dqkv, dqkv + D, dqkv + 2 * D, dvec, D, D, D, D, 3 * D, 3 * D, 3 * D,
                           R, H, DH, Tp, Tp, 1, (long long)Tp, 0, 1, (long long)Tp, 0, 1, scale};
            if (rt) ad.q_units = rt->u_strk;
            ad.drop = c->attn_drop(l, 1);
            const bool s16 = split && (size_t)l * 3 + 1 < c->qkv16.size() && c->qkv16[(size_t)l * 3 + 1];
            if (s16) { ad.io_bf16 = 1; ad.dq16 = dy16; ad.dk16 = dy16 + D; ad.dv16 = dy16 + 2 * D; cast_src = nullptr; ad.dout_bf16 = g16_1 ? 1 : 0;
                if (g16_1 && (size_t)l * 3 + 1 < c->attn_o16.size() && c->attn_o16[(size_t)l * 3 + 1]) {
                    ad.o = static_cast<const float*>(c->x16_find(ab(1, "attn"), D, 2));
                    ad.o_bf16 = 1;
                    SOLA_ARG(ad.o, "backward: the forward kept only the bf16 rows of attention output %d of layer %d, and they are not in the arena", 1, l);
                } }
            SOLA_TRY(wait_side(1));
            SOLA_TRY(launch_attention_bwd(ad, s));
            const float* x_pe = fb(ls + "_xpe");
            const float* x_obj = fb(ls + "_obj");
            float* sc3;
            SOLA_TRY(stats(dqkv, 3 * D, M, 3 * D, 4, &sc3, true, s16 ? dy16 : nullptr));
            if (sc3) {
                SOLA_TRY(bias_from_stats(0, D, G(an + ".q_proj.bias")));
                SOLA_TRY(bias_from_stats(D, D, G(an + ".k_proj.bias")));
                SOLA_TRY(bias_from_stats(2 * D, D, G(an + ".v_proj.bias")));
            }
            const WG w3[3] = {{dqkv, x_pe, G(an + ".q_proj.weight"), G(an + ".q_proj.bias")},
                              {dqkv + D, x_pe, G(an + ".k_proj.weight"), G(an + ".k_proj.bias")},
                              {dqkv + 2 * D, x_obj, G(an + ".v_proj.weight"), G(an + ".v_proj.bias")}};
            bool rm3;
            SOLA_TRY(dw_begin());
            SOLA_TRY(grad_w_many(w3, 3, 3 * D, D, M, D, D, sc3, sc3 != nullptr, &rm3));
            SOLA_TRY(dw_end(1));
            SOLA_TRY(transpose_into(W(an + ".q_proj.weight"), D, D, 2 * D, 0));
            SOLA_TRY(transpose_into(W(an + ".k_proj.weight"), D, D, 2 * D, D));
            // (bf16 steps, train_bf16_store 3: this branch gradient leaves its GEMM as bfloat16 rows; the inter-object norm's backward adds it as dy2)
            const bool e16 = split && pure && bf && g_train_bf16_store >= 3 && D % 8 == 0 && group_norm_bwd_dy2_bf16_supported(N, D, c->cfg.n_groups_module);
            egrad_bf16 = e16;
            SOLA_TRY(grad_x(dqkv, 3 * D, M, 2 * D, D, nullptr, egrad, sc3, rm3, 0, e16));  // d(x_obj + pe) = dq Wq + dk Wk
            SOLA_TRY(transpose_into(W(an + ".v_proj.weight"), D, D, D, 0));
            SOLA_TRY(grad_x(dqkv + 2 * D, 3 * D, M, D, D, dres, gbuf[1 - cur], sc3, rm3, 2 * D));  // d x_obj (direct) = dres + dv Wv
            cur = 1 - cur;
        }
        // (i) inter-object: x_obj = GN0(xin + attn(q,k,v(xin)) Wo); x_obj also feeds x_obj + pe
        {
            const std::string an = lp + "obj_attn";
            if (group) { dres = keep((size_t)M * D); dqkv = keep((size_t)M * 3 * D); }
            SOLA_TRY(gn_bwd(ab(0, "res"), gbuf[cur], egrad, lp + "norm.0", dres, n_bt, rt ? 1 : Tp, (long long)N * Tp, 1, Tp, N, D,
                            c->cfg.n_groups_module, 0, nullptr, rt ? rt->u_st : nullptr, gn16 ? dy16 : nullptr, (size_t)l * 3 + 0 < c->res16.size() && c->res16[(size_t)l * 3 + 0],
                            nullptr, egrad_bf16));
            const bool g16_0 = site_g16(0);
            SOLA_ARG(g16_0 || !((size_t)l * 3 + 0 < c->attn_o16.size() && c->attn_o16[(size_t)l * 3 + 0]),
                     "backward: attention output %d of layer %d exists as bf16 rows only, but this backward takes the f32 rows (a switch changed between the forward and the backward)", 0, l);
            SOLA_TRY(out_proj_bwd(0, g16_0));
            AttnBwdDesc ad{ab(0, "q"), ab(0, "k"), ab(0, "v"), ab(0, "attn"), dattn, ab(0, "lse"),
                           dqkv, dqkv + D, dqkv + 2 * D, dvec, D, D, D, D, 3 * D, 3 * D, 3 * D,
                           n_bt, H, DH, N, N, rt ? 1 : Tp, (long long)N * Tp, 1, Tp, (long long)N * Tp, 1, Tp, scale};
            if (rt) ad.q_units = rt->u_st;
            ad.drop = c->attn_drop(l, 0);
            const bool s16 = split && (size_t)l * 3 + 0 < c->qkv16.size() && c->qkv16[(size_t)l * 3 + 0];
            if (s16) { ad.io_bf16 = 1; ad.dq16 = dy16; ad.dk16 = dy16 + D; ad.dv16 = dy16 + 2 * D; cast_src = nullptr; ad.dout_bf16 = g16_0 ? 1 : 0;
                if (g16_0 && (size_t)l * 3 + 0 < c->attn_o16.size() && c->attn_o16[(size_t)l * 3 + 0]) {
                    ad.o = static_cast<const float*>(c->x16_find(ab(0, "attn"), D, 2));
                    ad.o_bf16 = 1;
                    SOLA_ARG(ad.o, "backward: the forward kept only the bf16 rows of attention output %d of layer %d, and they are not in the arena", 0, l);
                } }
            SOLA_TRY(wait_side(1));
            SOLA_TRY(launch_attention_bwd(ad, s));
            float* sc3;
            SOLA_TRY(stats(dqkv, 3 * D, M, 3 * D, 4, &sc3, true, s16 ? dy16 : nullptr));
            if (sc3) {
                SOLA_TRY(bias_from_stats(0, D, G(an + ".q_proj.bias")));
                SOLA_TRY(bias_from_stats(D, D, G(an + ".k_proj.bias")));
                SOLA_TRY(bias_from_stats(2 * D, D, G(an + ".v_proj.bias")));
            }
            const WG w3[3] = {{dqkv, xin, G(an + ".q_proj.weight"), G(an + ".q_proj.bias")},
                              {dqkv + D, xin, G(an + ".k_proj.weight"), G(an + ".k_proj.bias")},
                              {dqkv + 2 * D, xin, G(an + ".v_proj.weight"), G(an + ".v_proj.bias")}};
            bool rm3;
            SOLA_TRY(dw_begin());
            SOLA_TRY(grad_w_many(w3, 3, 3 * D, D, M, D, D, sc3, sc3 != nullptr, &rm3));
            SOLA_TRY(dw_end(1));
            SOLA_TRY(transpose_into(W(an + ".q_proj.weight"), D, D, 3 * D, 0));
            SOLA_TRY(transpose_into(W(an + ".k_proj.weight"), D, D, 3 * D, D));
            SOLA_TRY(transpose_into(W(an + ".v_proj.weight"), D, D, 3 * D, 2 * D));
            SOLA_TRY(grad_x(dqkv, 3 * D, M, 3 * D, D, dres, gbuf[1 - cur], sc3, rm3, 0));  // d xin = dres + [dq|dk|dv] [Wq;Wk;Wv]
            cur = 1 - cur;
        }
        if (l > 0) {
            SOLA_TRY(join_side());
            SOLA_TRY(flush_bias());  // the layer's deferred bias sums
            if (!group) SOLA_HIP(hipEventRecord(c->bucket_ev[c->cfg.n_layers - 1 - l], s));  // layer l's 30 gradients are final
        }
    }
    if (group) {  // the deferred weight gradients of every layer: one launch; all layer buckets are final behind it
        SOLA_TRY(flush_group());
        SOLA_TRY(flush_gn());  // the layers' norm parameter gradients
        for (int l = c->cfg.n_layers - 1; l > 0; --l) SOLA_HIP(hipEventRecord(c->bucket_ev[c->cfg.n_layers - 1 - l], s));
    }

    // ---- negative tokens: rows L.. of d(lang ++ neg) from the k/v projections + the mean-over-W of the score head
    SOLA_TRY(launch_neg_token_grad(dlang, ar.get("dlbar"), nullptr, G("negative_token.weight"), B, L, c->cfg.n_negative, D, s,
                                   rt ? rt->u_lang : nullptr));
    SOLA_TRY(join_side());
    SOLA_TRY(flush_bias());
    SOLA_HIP(hipEventRecord(c->bucket_ev[c->cfg.n_layers - 1], s));  // layer 0 + negative tokens

    // ---- encoder ----------------------------------------------------------------------------------------------
    const float* dy = gbuf[cur];  // gradient wrt conv5 output [R*T', D]
    float* enc[2] = {ar.get("enc0"), ar.get("enc1")};
    float* dwstd = ar.get("dwstd");
    size_t ws_off[6];
    {
        size_t o = 0;
        for (int i = 0; i < 6; ++i) {
            ws_off[i] = o;
            o += (size_t)c->conv[i].cout * c->conv[i].cin * c->conv[i].k;
        }
    }
    bool enc_dy16 = false;  // the current stage's dy also sits in "dy_sp" as bfloat16 rows (written by the norm backward that produced it)
    for (int i = 5; i >= 0; --i) {
        const ConvGeom& g = c->conv[i];
        const std::string cp = "short_motion_encoder." + std::to_string(kConvIdx[i]);
        const int t_in = i == 0 ? p.T : p.Tl[i - 1];
        const int rows = rt ? (int)rt->rows[i + 1] : R * p.Tl[i];
        const long long rows_in = rt ? rt->rows[i] : (long long)R * t_in;
        const int2* const rowmap = (rt && g.k > 1) ? rt->rowmap[i] : nullptr;
        const float* x_in = i == 0 ? nullptr : fb("act" + std::to_string(i - 1));
        // dW_std[cout][k*cin] = dY^T im2col(x_in), db
        float* scc = nullptr;  // scale slot of this layer's dY (shared by its dW and dX GEMMs)
        if (split && ar.off.count("tns") && gemm_tn_split_supported(rows, g.cout, g.k * g.cin) &&
            ar.total - ar.off.at("tns") >= gemm_tn_split_scratch_bytes(rows, g.cout, g.k * g.cin, 1)) {
            SOLA_TRY(stats(dy, g.cout, rows, g.cout, 8, &scc, true, enc_dy16 ? dy16 : nullptr));
            GemmTnSplitDesc d{};
            d.scal = scc; d.pure = pure ? 1 + bf : (dw16 ? 1 : 0); d.rm_split = dw16 ? 1 : 0;
            d.nprob = 1; d.A[0] = dy; d.B[0] = i == 0 ? c->last_obj : x_in; d.C[0] = dwstd + ws_off[i];
            d.M = rows; d.N = g.cout; d.K = g.k * g.cin; d.lda = g.cout; d.ldb = g.cin;
            d.conv = g.k > 1 ? 1 : 0; d.T_in = t_in; d.T_out = p.Tl[i]; d.stride = g.stride; d.pad = g.pad; d.Cin = g.cin;
            d.rowmap = rowmap; d.B_rows = rows_in;
            if (cast_src == dy && cast_ld == g.cout && cast_cols == g.cout) { d.a_rm = ar.get("dy_sp"); d.a_rm_ld = g.cout; d.a_rm_ready = 1; }
            if ((pure || dw16) && (i > 0 || (pure && bf))) d.B16[0] = c->x16_find(i == 0 ? c->last_obj : x_in, g.cin, pure ? 1 + bf : 1);
            d.scratch = ar.get("tns"); d.scratch_bytes = ar.total - ar.off.at("tns");
            SOLA_TRY(launch_gemm_tn_split(d, s));
            if (scc) SOLA_TRY(bias_from_stats(0, g.cout, G(cp + ".bias")));
            else SOLA_TRY(launch_colsum(dy, G(cp + ".bias"), 1, rows, g.cout, g.cout, 1.f, 0, ar.get("colsum"), ar.total - ar.off.at("colsum"), s));
        } else {
            GemmTnDesc d{};
            d.A = dy; d.B = i == 0 ? c->last_obj : x_in; d.C = dwstd + ws_off[i]; d.bias_grad = G(cp + ".bias");
            d.M = rows; d.N = g.cout; d.K = g.k * g.cin; d.lda = g.cout; d.ldb = g.cin;
            d.conv = g.k > 1 ? 1 : 0; d.T_in = t_in; d.T_out = p.Tl[i]; d.stride = g.stride; d.pad = g.pad; d.Cin = g.cin;
            d.rowmap = rowmap;
            d.scratch = tn; d.scratch_bytes = tn_bytes;
            if (group) {  // deferred with the other few-sample weight gradients: dy stays intact (keep() below)
                if (gq.nprob == 32) SOLA_TRY(flush_group());
                GemmTnGroupDesc::Prob& q = gq.p[gq.nprob++];
                q = GemmTnGroupDesc::Prob{};
                q.A = d.A; q.B = d.B; q.C = d.C; q.bias_grad = d.bias_grad; q.rowmap = d.rowmap;
                q.M = d.M; q.N = d.N; q.K = d.K; q.lda = d.lda; q.ldb = d.ldb;
                q.conv = d.conv; q.T_in = d.T_in; q.T_out = d.T_out; q.stride = d.stride; q.pad = d.pad; q.Cin = d.Cin;
            } else {
            SOLA_TRY(dw_begin());
            SOLA_TRY(launch_gemm_tn(d, s2));
            SOLA_TRY(dw_end(2));
            }
        }
        if (i == 0) break;
        // d act_{i-1}[(r, ti)][ci] = sum_{kk, co} dY[(r, to)][co] w_std[co][kk][ci]: the NT kernel with the transposed-conv
        // gather over dY and the weights re-laid-out to [cin][kk*cout + co]
        float* dact = enc[0];
        auto col2im = [&](const float* zc, int z16 = 0) -> int {
            if (rt) return launch_col2im_ragged(zc, dact, rows_in, rt->imap[i - 1], g.cin, g.k, g.stride, g.pad, s, z16);
            return launch_col2im(zc, dact, R, t_in, p.Tl[i], g.cin, g.k, g.stride, g.pad, s, z16);
        };
        // few-sample exact-f32 backward: z = dY W over the OUTPUT rows with the standardised weights read where the forward left them
        // ([cout][k*cin] row-major = the few-row kernel's NN form: no transposed copy), then the taps are gathered - uniform batches too
        // (their transposed-conv gather form multiplies twice the products for the stride-2 convs and ran on the 64x64 + split-K pair)
        GemmDesc zd{};
        if (group) {
            zd.nprob = 1;
            zd.p[0] = GemmProblem{dy, nullptr, nullptr, nullptr, g.k > 1 ? ar.get("zcol") : dact};
            zd.M = rows; zd.N = g.k * g.cin; zd.K = g.cout; zd.lda = g.cout; zd.ldc = g.k * g.cin;
            zd.w_nn[0] = c->ws_buf + c->ws_off[i]; zd.w_nn_rows = g.cout;
        }
        if (group && gemm_nn_supported(zd)) {
            SOLA_TRY(launch_gemm(zd, s));
            if (g.k > 1) SOLA_TRY(col2im(ar.get("zcol")));
        } else if (split && g.cout % 128 == 0 && g.cin % 8 == 0 && (size_t)rows * g.cout <= std::max((size_t)M, (size_t)BW) * 3 * D) {
            // split-f16: z[(r,to)][kk*cin+ci] = sum_co dY[(r,to)][co] w_std[co][kk*cin+ci] in ONE NT GEMM over the output steps (a
            // strided conv's gather form would multiply zeros for every skipped step), then the k taps are gathered into dX
            float* scal = scc ? scc : ar.get("scal");
            if (scc && cast_src == dy && cast_ld == g.cout && cast_cols == g.cout) {}  // the statistics pass left the cast (bf16)
            else if (scc) SOLA_TRY(cast_scaled(dy, g.cout, ar.get("dy_sp"), rows, g.cout, scal));
            else SOLA_TRY(cast_auto(dy, g.cout, ar.get("dy_sp"), rows, g.cout, scal));
            if (pure) SOLA_TRY(launch_cast_f16_t(c->ws_buf + c->ws_off[i], g.k * g.cin, ar.get("wt_sp"), g.cout, g.cout, g.k * g.cin, nullptr, s, bf));
            else SOLA_TRY(launch_cast_sp16_t(c->ws_buf + c->ws_off[i], g.k * g.cin, ar.get("wt_sp"), g.cout, g.cout, g.k * g.cin, nullptr, s));
            GemmDesc d{};
            d.nprob = 1;
            d.p[0] = GemmProblem{ar.get("dy_sp"), ar.get("wt_sp"), nullptr, nullptr, g.k > 1 ? ar.get("zcol") : dact};
            d.M = rows; d.N = g.k * g.cin; d.K = g.cout; d.lda = g.cout; d.ldc = g.k * g.cin;
            d.arith = lowp_arith; d.bf16 = bf; d.out_scale = 1.f; d.out_scale_dev = scal + 1;
            // bf16 steps (train_bf16_store): the per-tap contributions leave the GEMM as bfloat16 rows - the input gradient of a bf16 conv as
            // autocast computes it - and the gather sums the taps in f32: half the bytes written and read between the two launches
            const int z16 = (pure && bf && g_train_bf16_store >= 3 && g.k > 1 && (g.k * g.cin) % 8 == 0) ? 1 : 0;
            d.c_f16 = z16;
            SOLA_TRY(launch_gemm(d, s));
            if (g.k > 1) SOLA_TRY(col2im(ar.get("zcol"), z16));
        } else if (rt) {
            // ragged, exact f32: the same decomposition - z = dY W over the OUTPUT rows (a plain GEMM on the concatenated rows, half
            // the products of the gather form for the stride-2 convs), then the taps are gathered per sequence
            SOLA_TRY(launch_transpose(c->ws_buf + c->ws_off[i], wt, g.cout, g.k * g.cin, g.k * g.cin, g.cout, 0, s));  // wt [k*cin][cout]
            GemmDesc d{};
            d.nprob = 1;
            d.p[0] = GemmProblem{dy, wt, nullptr, nullptr, g.k > 1 ? ar.get("zcol") : dact};
            d.M = rows; d.N = g.k * g.cin; d.K = g.cout; d.lda = g.cout; d.ldc = g.k * g.cin;
            d.splitk_ws = splitk_ws; d.splitk_bytes = splitk_bytes;
            SOLA_TRY(launch_gemm(d, s));
            if (g.k > 1) SOLA_TRY(col2im(ar.get("zcol")));
        } else {
        for (int kk = 0; kk < g.k; ++kk)
            SOLA_TRY(launch_transpose(c->ws_buf + c->ws_off[i] + (size_t)kk * g.cin, wt, g.cout, g.cin, g.k * g.cin,
                                      g.k * g.cout, kk * g.cout, s));
        {
            GemmDesc d{};
            d.nprob = 1;
            d.p[0] = GemmProblem{dy, wt, nullptr, nullptr, dact};
            d.M = R * t_in; d.N = g.cin; d.K = g.k * g.cout; d.lda = g.cout; d.ldc = g.cin;
            d.conv = g.k > 1 ? 2 : 0; d.T_in = p.Tl[i]; d.T_out = t_in; d.stride = g.stride; d.pad = g.pad; d.Cin = g.cout;
            d.splitk_ws = splitk_ws; d.splitk_bytes = splitk_bytes;
            SOLA_TRY(launch_gemm(d, s));
        }
        }
        // GroupNorm + LeakyReLU backward of stage i-1
        const std::string np = "short_motion_encoder." + std::to_string(kNormIdx[i - 1]);
        const DropoutCfg edrop = c->enc_drop(i - 1);
        if (group) enc[1] = keep((size_t)rows_in * g.cin);  // conv i-1's dY: read again by its deferred weight gradient
        // the next stage's statistics pass finds the bf16 rows ready when they fit "dy_sp", the pitch is 16-byte aligned and that stage takes
        // the 16-bit dW route (the conditions of the statistics call at the top of the loop)
        enc_dy16 = gn16 && g.cin % 8 == 0 && (size_t)rows_in * g.cin <= dy_sp_halfs && ar.off.count("tns") &&
                   gemm_tn_split_supported((int)rows_in, g.cin, c->conv[i - 1].k * c->conv[i - 1].cin) &&
                   ar.total - ar.off.at("tns") >= gemm_tn_split_scratch_bytes((int)rows_in, g.cin, c->conv[i - 1].k * c->conv[i - 1].cin, 1);
        SOLA_TRY(gn_bwd(fb("conv" + std::to_string(i - 1)), dact, nullptr, np, enc[1], R, 1, t_in, 0, 1, t_in, g.cin,
                        c->cfg.n_groups, 1, &edrop, rt ? rt->u_lvl[i] : nullptr, enc_dy16 ? dy16 : nullptr));
        dy = enc[1];  // dact (enc[0]) is consumed; the next stage's dX may overwrite it, its GN backward overwrites enc[1]
    }
    // weight-standardisation backward for all six convs
    if (group) SOLA_TRY(flush_group());  // the encoder's deferred weight gradients
    SOLA_TRY(flush_gn());                // ... and its norms' parameter gradients
    SOLA_TRY(join_side());  // dwstd is the side stream's
    {
        WsBwdLayer layers[6];
        for (int i = 0; i < 6; ++i) {
            const std::string cp = "short_motion_encoder." + std::to_string(kConvIdx[i]);
            layers[i] = WsBwdLayer{W(cp + ".weight"), dwstd + ws_off[i], G(cp + ".weight"), c->conv[i].cout, c->conv[i].cin, c->conv[i].k};
        }
        SOLA_TRY(launch_ws_backward(layers, 6, s));
    }
    SOLA_TRY(flush_bias());
    SOLA_HIP(hipEventRecord(c->bucket_ev[c->cfg.n_layers], s));  // encoder
    c->bucket_recorded = true;
    return SOLA_OK;
}

extern "C" int sola_backward(SolaCtx* c, const float* d_score_map, const float* d_score_tokens, const void* fwd_workspace,
                             void* scratch, size_t scratch_bytes, void* stream_) {
    try {
        return backward_impl(c, d_score_map, d_score_tokens, fwd_workspace, scratch, scratch_bytes, stream_, false);
    } catch (const std::exception& e) {
        sola_set_error("backward: %s", e.what());
        return SOLA_ERR_ARG;
    }
}

extern "C" int sola_backward_ragged(SolaCtx* c, const float* d_score_map, const float* d_score_tokens, const void* fwd_workspace,
                                    void* scratch, size_t scratch_bytes, void* stream_) {
    try {
        return backward_impl(c, d_score_map, d_score_tokens, fwd_workspace, scratch, scratch_bytes, stream_, true);
    } catch (const std::exception& e) {
        sola_set_error("backward_ragged: %s", e.what());
        return SOLA_ERR_ARG;
    }
}

extern "C" int sola_grad_bucket_count(const SolaCtx* c) { return c ? c->n_buckets() : 0; }

// bucket of a parameter: layers in reverse order, then layer 0 together with the negative tokens, then the encoder
extern "C" int sola_grad_bucket_of(const SolaCtx* c, const char* name) {
    SOLA_ARG(c && name, "grad_bucket_of: null argument");
    const std::string n(name);
    if (c->index.find(n) == c->index.end() || n == "positional_encoding_gaussian_matrix") {
        sola_set_error("grad_bucket_of: '%s' is not a parameter", name);
        return SOLA_ERR_WEIGHT;
    }
    const std::string lp = "object_lang_align_layers.";
    if (n.compare(0, lp.size(), lp) == 0) return c->cfg.n_layers - 1 - atoi(n.c_str() + lp.size());
    if (n == "negative_token.weight") return c->cfg.n_layers - 1;
    return c->cfg.n_layers;
}

extern "C" int sola_backward_wait_bucket(SolaCtx* c, int bucket, void* stream_) {
    SOLA_ARG(c && bucket >= 0 && bucket < c->n_buckets(), "backward_wait_bucket: bucket %d out of range", bucket);
    if (!c->bucket_recorded) {
        sola_set_error("backward_wait_bucket: no completed sola_backward on this context");
        return SOLA_ERR_STATE;
    }
    SOLA_HIP(hipStreamWaitEvent(as_stream(stream_), c->bucket_ev[bucket], 0));
    return SOLA_OK;
}
