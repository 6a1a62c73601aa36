// Context, weight registry and workspace plan shared by api.hip / forward.hip / backward.hip.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"

struct ConvGeom {
    int cin, cout, k, stride, pad;
};
struct Weight {
    std::string name;
    int64_t numel;
    const float* ptr;  // borrowed parameter / buffer storage
    float* grad;       // borrowed gradient storage (training only)
};
struct Buf {
    size_t off;
    int64_t rows, cols;
};
struct Plan {
    int B = 0, N = 0, T = 0, L = 0, W = 0, Tp = 0, M = 0;
    int Tl[6] = {0};
    bool train = false;
    bool rag = false;  // ragged training batch: B = samples, N / T / L / W / Tl / Tp = the LARGEST extents, M = all layer rows
    std::unordered_map<std::string, Buf> bufs;
    size_t total = 0;
    size_t add(const std::string& name, int64_t rows, int64_t cols) {
        const size_t off = total;
        bufs[name] = Buf{off, rows, cols};
        total += (((size_t)rows * (size_t)cols * sizeof(float)) + 255) & ~(size_t)255;
        return off;
    }
};

// Ragged batches (ragged.h): device tables (pointers into the caller's workspace) + the host-side extents the launchers need.
// The context keeps the set of the last ragged training forward for its backward.
struct RagTables {
    const int4* u_lvl[7] = {nullptr};   // [NT] per encoder level: (first row, 1, T_level, video) of every track
    const int2* rowmap[5] = {nullptr};  // conv l, per OUTPUT row (level l + 1): (source row of tap 0, tap-validity bits)
    const int4* imap[5] = {nullptr};    // training only: conv l + 1's INPUT rows (level l + 1, l = 0..4): (first output row of the
                                        // row's sequence, T_out, step ti inside the sequence, -) - the col2im gather of the conv dX
    const int4* u_vt = nullptr;         // [sum T'_v]  (first row, T'_v, N_v, t')  inter-object units per video
    const int4* u_st = nullptr;         // [sum T'_i]  ... per sample
    const int4* u_strk = nullptr;       // [sum N_i]   (first row, 1, T', sample)  tracks of the samples
    const int4* u_smp = nullptr;        // [S]         (first row, 1, N*T', sample)
    const int4* u_lang = nullptr;       // [S]         (first input text row, L, first lang_cat row, W)
    const int4* u_langk = nullptr;      // [S]         (first lang_cat row, 1, W, 0)
    const int4* u_gather = nullptr;     // [S]         (first sample row, first video row at T', rows, 0)
    const int32_t* trk_off = nullptr;   // [S + 1]     first track of every sample
    // host-side extents
    int V = 0, S = 0, NT = 0, sumNS = 0, sumTpV = 0, sumTpS = 0;
    int maxN = 0, maxT[7] = {0}, maxW = 0, maxRowsSample = 0;
    long long rows[7] = {0}, Mv = 0, Ms = 0, LW = 0, Lin = 0;
    bool identity = false;
};

struct SolaRagStage;  // pinned staging ring of the ragged forward's descriptor upload (forward_ragged.hip)
void sola_rag_stage_free(SolaRagStage* st);

struct SolaCtx {
    SolaConfig cfg;
    int device;
    ConvGeom conv[6];
    std::vector<Weight> weights;
    std::unordered_map<std::string, int> index;
    float* ws_buf = nullptr;  // standardised conv weights [cout][k*cin], all six layers, ctx-owned
    size_t ws_off[6];
    bool ws_dirty = true;
    int ws16_fmt = 0;         // what ws16_buf holds for the CURRENT ws_buf: 0 nothing / stale, 1 split-f16 pairs, 2 f16, 3 bf16 (a forward in another
                              // arithmetic clears ws_dirty without writing this copy: round 4, few-row calls routed to exact f32)
    bool ws_every_forward = true;
    Plan last;                // plan of the last forward (taps, backward)
    const float* last_obj = nullptr;  // input of the last training forward (conv0's weight gradient reads it)
    RagTables last_rag;               // last.rag: the unit tables that forward left in its workspace
    const void* last_ws = nullptr;    // workspace of the last forward (sola_backward_ragged checks it gets the same one)
    // round 6, bf16 training step: attention site (layer * 3 + which) of the last training forward keeps its q / k / v as BFLOAT16 rows
    // (written by the projection GEMM's epilogue, read by the attention forward and backward; the backward then writes dq / dk / dv as
    // bfloat16 too - the dW / dX GEMMs' operand, no f32 copy, no cast pass).  Decided per site by the shapes the kernels take.
    std::vector<char> qkv16;
    // ... and sub-block (layer * 3 + which) keeps its PRE-NORM rows (out-projection + residual, the GroupNorm's input) as bfloat16: the
    // out-projection GEMM reads the residual from the sub-block input's bf16 operand copy and writes bf16, the GroupNorm forward and
    // backward read bf16 (sola_tune "train_bf16_store" 2)
    std::vector<char> res16;
    std::vector<char> attn_o16;   // per (layer, site): the last training forward wrote that attention's output as bfloat16 rows ONLY (arena slot keyed by the f32 buffer)
    std::vector<char> gn2_stats;  // per layer: the last training forward left the object->language norm's (mean, rstd) in "l<l>_gn2st" (forward.hip)
    // inference precision: 0 = exact f32 MFMA; 1 = split-f16 operands, 3 x f16 MFMA with f32 accumulation (cast.hip).
    // ctx-owned split-f16 copies of the weights: standardised conv weights (same offsets as ws_buf) and the
    // 12 * n_layers linear weights pre-scaled by 64 (index (layer * 3 + attn) * 4 + proj, D*D floats each).
    int precision = 0;
    bool lin16_dirty = true;  // split copies of the projection weights are stale
    float* ws16_buf = nullptr;
    float* lin16_buf = nullptr;
    // 16-bit operand modes, training: the row-major casts of the forward's GEMM inputs are KEPT (one arena slot each instead of two
    // shared buffers) and listed by their f32 source, so that the backward's weight-gradient products take them as their X operand
    // instead of casting the same activations again (sola_tune "train_x16_keep").  The arena grows between steps to the last need.
    struct X16Entry { const float* src; const void* p16; int cols; int fmt; };
    std::vector<X16Entry> x16;
    char* x16_arena = nullptr;  // BORROWED from the caller (sola_set_x16_arena); null = nothing is kept, the backward casts as before
    size_t x16_cap = 0, x16_used = 0, x16_need = 0;  // x16_need: what the last training forward asked for in total (fitting or not)
    void* x16_alloc(size_t bytes) {  // null = no room this step (the caller uses its shared buffer and lists nothing)
        bytes = (bytes + 255) & ~(size_t)255;
        x16_need += bytes;
        if (x16_used + bytes > x16_cap) return nullptr;
        void* r = x16_arena + x16_used;
        x16_used += bytes;
        return r;
    }
    const void* x16_find(const float* src, int cols, int fmt) const {
        for (const X16Entry& e : x16)
            if (e.src == src && e.cols == cols && e.fmt == fmt) return e.p16;
        return nullptr;
    }
    // Range handling of the split-f16 mode (device floats, ctx-owned): (max|x| bits, 1/scale) pairs of the data-dependent
    // power-of-two scales - [0] object tokens, [1] text ++ negative tokens, [2 + i] projection weight i (same index as
    // lin16_buf) - followed by the guard words: guard[0] is cleared by every inference forward and gets bit 0 from any kernel
    // that had to write a non-finite or out-of-f16-range value as a split-f16 pair; guard[1] is recomputed when the weights
    // change (bit 1: a GroupNorm's (gamma, beta) put its output outside the magnitude the fixed activation scale covers).
    // Either sends the call to the exact-f32 kernels (api.hip: sola_forward).
    float* scal_buf = nullptr;
    int* guard = nullptr;
    int* guard_host = nullptr;      // pinned, 2 ints
    bool split_guard = true;        // sola_set_split_guard
    long long split_fallbacks = 0;  // calls that were repeated in exact f32
    bool weight_range_bad = false;  // the weight-time guard bit was seen set for the current weights: precision 1 goes straight to f32
    SolaRagStage* rag_stage = nullptr;
    // Gradient buckets of sola_backward, in the order their gradients become final: layer n-1, ..., layer 1, layer 0 (+ the
    // negative tokens, whose gradient collects contributions from every layer), encoder.  An event is recorded on the
    // backward's stream when a bucket is complete, so the caller can start that bucket's all-reduce on another stream
    // while the rest of the backward still runs (sola_backward_wait_bucket).
    std::vector<hipEvent_t> bucket_ev;
    // few-sample backward (exact f32, round 4): the weight-gradient products run on a side stream beside the dX chain (backward.hip)
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_side[4] = {nullptr, nullptr, nullptr, nullptr};
    bool bucket_recorded = false;
    int n_buckets() const { return cfg.n_layers + 1; }
    float* scal_pair(int i) const { return scal_buf + 2 * i; }
    float* scal_extra(int i) const { return scal_buf + 2 * (2 + (size_t)cfg.n_layers * 12) + 2 + i; }  // 4 spare device floats behind the guard words
    const float* lin_inv_scale(int layer, int attn, int proj) const { return scal_buf + 2 * (2 + (layer * 3 + attn) * 4 + proj) + 1; }
    // sola_train_step (train_step.hip): the gradient tensors in the order the norm reduction takes them (group by group, as
    // module/module.py:164-199 walks them), resolved by name once (sola_train_step_bind)
    struct StepBinding {
        std::vector<int> widx;
        std::vector<int32_t> group;
        std::vector<long long> numel;
        int n_groups = 0;
        size_t sq_scratch = 0;
    } step;
    // sola_adamw_bind / sola_adamw_step (train_step.hip): device table of (param, grad, exp_avg, exp_avg_sq, step) records, ctx-owned
    void* adam_tab = nullptr;
    int adam_n = 0, adam_cap = 0, adam_blocks = 0;  // adam_n = 0: not bound (also after a weight / gradient pointer changed: bind again)
    bool adam_has_steps = false;                    // the bound optimizer carries device step tensors (step 0 = "read them")
    int* adam_ticket = nullptr;                     // the update kernel's last-block ticket (device int, zero between launches)
    double adam_bytes = 0.0;
    // dropout of the training forward (module/module.py:78-94 p = dropout_p; tools/attention.py:12,71 p = 0.1);
    // the seed used by the last sola_forward_train is kept for sola_backward
    float p_drop_encoder = 0.f, p_drop_attention = 0.f;
    unsigned long long drop_seed = 0;
    DropoutCfg enc_drop(int conv_idx) const { return make_dropout(p_drop_encoder, drop_seed, 1u + (unsigned)conv_idx); }
    DropoutCfg attn_drop(int layer, int a) const { return make_dropout(p_drop_attention, drop_seed, 100u + 3u * (unsigned)layer + (unsigned)a); }
};

static const int kConvIdx[6] = {0, 4, 8, 12, 16, 20};
static const int kNormIdx[5] = {1, 5, 9, 13, 17};
static const char* const kAttnShort[3] = {"obj", "mot", "o2l"};
static const char* const kAttnLong[3] = {"obj_attn", "motion_attn", "object2lang_attn"};
constexpr float kLinScale = 64.f;  // backward only (transposed weight casts): linear weights are U(-1/32, 1/32)-sized
// (re)builds the split-f16 copies of the projection weights with a per-matrix power-of-two scale and runs the weight-time
// range check; no-op unless the copies are stale
int sola_refresh_lin16(SolaCtx* c, hipStream_t s);

inline const float* ctx_weight(const SolaCtx* c, const std::string& name) {
    auto it = c->index.find(name);
    return it == c->index.end() ? nullptr : c->weights[it->second].ptr;
}
inline float* ctx_grad(const SolaCtx* c, const std::string& name) {
    auto it = c->index.find(name);
    return it == c->index.end() ? nullptr : c->weights[it->second].grad;
}

// Name of a per-attention buffer: shared scratch in inference, one per (layer, attention) when saving for backward.
inline std::string abuf(bool train, int layer, const char* attn, const char* what) {
    if (!train) return what;
    return "l" + std::to_string(layer) + "_" + attn + "_" + what;
}

struct RagShape;
Plan make_plan(const SolaCtx* c, int B, int N, int T, int L, bool train);
Plan make_plan_ragged(const SolaCtx* c, const RagShape& r, bool train);
int sola_forward_impl(SolaCtx* c, const float* obj, const float* lang, int B, int N, int T, int L, float* score_map,
                      float* score_tokens, void* workspace, size_t ws_bytes, hipStream_t s, bool train, const RagShape* rs = nullptr);
int sola_forward_fast_impl(SolaCtx* c, const float* obj, const float* lang, int B, int N, int T, int L, float* score_map,
                           float* score_tokens, void* workspace, size_t ws_bytes, hipStream_t s);
int sola_forward_f16_impl(SolaCtx* c, const float* obj, const float* lang, int B, int N, int T, int L, float* score_map,
                          float* score_tokens, void* workspace, size_t ws_bytes, hipStream_t s);
size_t sola_backward_scratch_bytes(const SolaCtx* c, const Plan& p);
size_t sola_ragged_workspace_bytes_impl(const SolaCtx* c, const SolaRaggedBatch* b, int precision);
int sola_forward_ragged_impl(SolaCtx* c, const float* obj, const float* lang, const SolaRaggedBatch* batch, float* score_map,
                             float* score_tokens, void* workspace, size_t ws_bytes, hipStream_t s);
