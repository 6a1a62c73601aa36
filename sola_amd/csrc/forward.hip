// Forward orchestration of LanguageAlignedTrackSelectionModule.forward (module/module.py:130-162): host code that
// sequences the kernels of gemm.hip / attn.hip / norm.hip / head.hip over one workspace arena.  With train = true every
// attention keeps its own q/k/v/output/residual buffers plus the softmax log-sum-exp, which backward.hip consumes.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "ragged.h"

int g_train_gn_cast = 1;  // sola_tune "train_gn_cast": 1 = a GroupNorm of the training forward also writes the operand cast of the GEMM behind it
void sola_train_set_gn_cast(int v) { g_train_gn_cast = v; }
extern int g_train_dw_f16;
int g_train_attn_cast = 1;  // sola_tune "train_attn_cast": 1 = f16 / bf16 operand steps: the training forward's attention kernels also write the out-projection's operand cast; 2 = the split-f16 step too
void sola_train_set_attn_cast(int v) { g_train_attn_cast = v; }
int g_train_bf16_store = 3;  // sola_tune "train_bf16_store" (round 6): 1 = bf16 steps keep q / k / v and the backward's gradient operands as bfloat16 rows where the kernels take
                             // them; 2 = also the pre-norm rows of the layers (out-projection + residual -> GroupNorm); 3 (default) = also the conv input gradients' per-tap
                             // contributions between their GEMM and the gather; 0 = f32 storage (the round-5 step)
int g_train_gn_stats = 1;  // sola_tune "train_gn_stats": 1 = the training forward keeps the sliced GroupNorm shape's (mean, rstd) for the backward (A/B)
void sola_train_set_gn_stats(int v) { g_train_gn_stats = v; }
int g_train_x16_keep = 1;  // sola_tune "train_x16_keep": 1 = 16-bit operand modes keep the forward's operand casts for the backward's dW products (ctx.h)
int g_train_split_min_rows = 1024;  // sola_tune "train_split_min_rows": training takes the split-f16 GEMMs from this many token rows on
void sola_set_train_split_min_rows(int v) { g_train_split_min_rows = v; }

// Workspace layout shared by the uniform and the ragged plans: only the row counts differ.
namespace {
struct PlanSizes {
    int64_t rows0;      // object-token rows (level 0)
    int64_t rows[6];    // token rows behind each conv
    int64_t M, BW;      // layer rows, text ++ negative rows
    int64_t S;          // samples
    int64_t gn_slots;   // floats of the sliced GroupNorm shape's scratch
    size_t tables;      // ragged: bytes of the descriptor / unit-table region (0 = uniform batch)
};
void plan_fill(Plan& p, const SolaCtx* c, const PlanSizes& z, bool train) {
    const int D = c->cfg.lang_token_dim, H = c->cfg.num_heads;
    if (z.tables) p.add("tables", (int64_t)((z.tables + 3) / 4), 1);
    for (int i = 0; i < 6; ++i) {
        p.add("conv" + std::to_string(i), z.rows[i], c->conv[i].cout);
        if (i < 5) p.add("act" + std::to_string(i), z.rows[i], c->conv[i].cout);
    }
    // single-sample regime: scratch for the two-pass split-K of the GEMMs whose grid is smaller than the chip (gemm.hip)
    if (z.M <= 8192) p.add("splitk", 8192, 4096);
    p.add("pe", p.Tp, D);
    // slots of the sliced GroupNorm shape (8 bytes per (unit, slice); units of more than 256 tokens at 128 channels per group):
    // bounded by rows / 11 entries over every norm of the path
    p.add("gn_slots", z.gn_slots, 1);
    p.add("lang", z.BW, D);
    p.add("lbar", z.S, D);
    if (!train && c->precision >= 1) {  // split-f16 (or f16) copies of the three f32-born GEMM inputs (forward_fast.hip, forward_f16.hip)
        p.add("obj_sp", z.rows0, c->cfg.object_token_dim);
        p.add("conv5_sp", z.M, D);
        p.add("lang_sp", z.BW, D);
    }
    if (train && c->precision >= 1 && z.M >= g_train_split_min_rows) {
        // split-f16 training forward: every GEMM input is cast into one of two scratch buffers right before its launch
        // (the f32 activations stay where the backward reads them)
        int64_t amax = std::max<int64_t>(z.rows0 * c->cfg.object_token_dim, std::max<int64_t>(z.M * D, z.BW * D));
        for (int i = 0; i < 5; ++i) amax = std::max<int64_t>(amax, z.rows[i] * c->conv[i].cout);
        p.add("sp_a", amax, 1);
        p.add("sp_b", z.M, D);
    }
    const int n_sets = train ? c->cfg.n_layers : 1;
    for (int l = 0; l < n_sets; ++l)
        for (int a = 0; a < (train ? 3 : 1); ++a) {
            p.add(abuf(train, l, kAttnShort[a], "q"), z.M, D);
            if (!train || a < 2) {
                p.add(abuf(train, l, kAttnShort[a], "k"), z.M, D);
                p.add(abuf(train, l, kAttnShort[a], "v"), z.M, D);
            }
            p.add(abuf(train, l, kAttnShort[a], "attn"), z.M, D);
            p.add(abuf(train, l, kAttnShort[a], "res"), z.M, D);
            if (train) p.add(abuf(train, l, kAttnShort[a], "lse"), z.M, H);
            if (!train || a == 2) {
                p.add(abuf(train, l, kAttnShort[a], "lk"), z.BW, D);
                p.add(abuf(train, l, kAttnShort[a], "lv"), z.BW, D);
            }
        }
    for (int l = 0; l < c->cfg.n_layers; ++l) {
        // training: (mean, rstd) of the object->language norm's (sample, group) units where the sliced shape computes them - the backward's
        // three-pass kernel of those units then walks x twice instead of four times (GroupNormDesc::stats_out)
        if (train) p.add("l" + std::to_string(l) + "_gn2st", z.S * c->cfg.n_groups_module, 2);
        p.add("l" + std::to_string(l) + "_obj", z.M, D);
        p.add("l" + std::to_string(l) + "_xpe", z.M, D);
        p.add("l" + std::to_string(l) + "_motion", z.M, D);
        p.add("l" + std::to_string(l) + "_o2l", z.M, D);
    }
}
}  // namespace

Plan make_plan(const SolaCtx* c, int B, int N, int T, int L, bool train) {
    Plan p;
    p.B = B; p.N = N; p.T = T; p.L = L; p.train = train;
    p.W = L + c->cfg.n_negative;
    int t = T;
    for (int i = 0; i < 6; ++i) {
        t = (t + 2 * c->conv[i].pad - c->conv[i].k) / c->conv[i].stride + 1;
        p.Tl[i] = t;
    }
    p.Tp = p.Tl[5];
    p.M = B * N * p.Tp;
    const int64_t R = (int64_t)B * N;
    PlanSizes z{};
    z.rows0 = R * T;
    for (int i = 0; i < 6; ++i) z.rows[i] = R * p.Tl[i];
    z.M = p.M; z.BW = (int64_t)B * p.W; z.S = B;
    z.gn_slots = std::max<int64_t>(R * T, (int64_t)p.M) / 4 + 1024;
    plan_fill(p, c, z, train);
    return p;
}

// Plan of a ragged TRAINING batch (sola_forward_train_ragged): the same buffers over the concatenated rows, plus the table
// region.  B = samples; N, T, L, W, Tl, Tp hold the LARGEST extents (they only select kernel shapes); M = all layer rows.
Plan make_plan_ragged(const SolaCtx* c, const RagShape& r, bool train) {
    Plan p;
    p.rag = true;
    p.B = r.S; p.N = r.maxN; p.T = r.maxT[0]; p.L = r.maxW - c->cfg.n_negative; p.train = train;
    p.W = r.maxW;
    for (int i = 0; i < 6; ++i) p.Tl[i] = r.maxT[i + 1];
    p.Tp = r.maxT[6];
    p.M = (int)r.Ms;
    PlanSizes z{};
    z.rows0 = r.rows[0];
    for (int i = 0; i < 6; ++i) z.rows[i] = r.rows[i + 1];
    z.M = r.Ms; z.BW = r.LW; z.S = r.S;
    z.gn_slots = (int64_t)(rag_gn_slots_bytes(r) / 4 + 64);
    z.tables = rag_tables_bytes(r, train);
    plan_fill(p, c, z, train);
    return p;
}

// rs != null: a ragged TRAINING batch (sola_forward_train_ragged) - the token rows of all samples concatenated, every shape-
// dependent kernel reads the unit tables built here (ragged.h); B, N, T, L are then ignored.
int sola_forward_impl(SolaCtx* c, const float* obj, const float* lang, int B, int N, int T, int L, float* score_map,
                      float* score_tokens, void* workspace, size_t ws_bytes, hipStream_t s, bool train, const RagShape* rs) {
    SOLA_ARG(c && obj && lang && score_map && score_tokens && workspace, "forward: null argument");
    SOLA_ARG(c->precision != 3 || train, "forward: precision 3 (bf16 GEMM operands) is a TRAINING mode; inference runs precision 0, 1 or 2");
    if (rs) SOLA_ARG(train && rs->identity, "forward: a ragged training batch has one sample per video (sample_video[i] == i)");
    else SOLA_ARG(B > 0 && N > 0 && T > 0 && L >= 1, "forward: bad sizes B=%d N=%d T=%d L=%d", B, N, T, L);
    for (const Weight& w : c->weights)
        if (!w.ptr) {
            sola_set_error("forward: weight '%s' has not been set", w.name.c_str());
            return SOLA_ERR_WEIGHT;
        }
    Plan p = rs ? make_plan_ragged(c, *rs, train) : make_plan(c, B, N, T, L, train);
    if (ws_bytes < p.total) {
        sola_set_error("forward: workspace %zu bytes < required %zu", ws_bytes, p.total);
        return SOLA_ERR_WORKSPACE;
    }
    SOLA_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "forward: workspace must be 256-byte aligned");
    char* base = static_cast<char*>(workspace);
    auto buf = [&](const std::string& name) { return reinterpret_cast<float*>(base + p.bufs.at(name).off); };
    float* const splitk_ws = p.bufs.count("splitk") ? buf("splitk") : nullptr;
    const size_t splitk_bytes = splitk_ws ? (size_t)p.bufs.at("splitk").rows * p.bufs.at("splitk").cols * sizeof(float) : 0;
    auto W = [&](const std::string& name) { return ctx_weight(c, name); };
    const int D = c->cfg.lang_token_dim, H = c->cfg.num_heads, DH = D / H;
    const int Tp = p.Tp, M = p.M, Wn = p.W;
    if (rs) { B = rs->S; N = rs->maxN; T = rs->maxT[0]; }
    const int R = rs ? rs->NT : B * N;
    RagTables rt;  // ragged: unit tables in the workspace (they stay valid for sola_backward_ragged)
    if (rs) SOLA_TRY(rag_build_tables(c, *rs, base + p.bufs.at("tables").off, true, &rt, s));
    auto level_rows = [&](int lvl) -> long long { return rs ? rs->rows[lvl] : (long long)R * (lvl == 0 ? T : p.Tl[lvl - 1]); };
    const long long text_rows = rs ? rs->LW : (long long)B * Wn;
    // training forward in the split-f16 mode (sola_set_precision 1): the same GEMM kernels as forward_fast.hip on casts of
    // the f32 activations, everything the backward reads stays f32
    // (from ~1024 token rows on: below that the step is launch-bound and the extra cast launches cost more than the GEMMs gain)
    // precision 2 ("pure"): the same structure with plain f16 GEMM operands, ONE MFMA per product - mixed-precision training:
    // activations, statistics, softmax, accumulation and everything the backward reads stay f32
    const bool pure = c->precision >= 2;
    const int bf = c->precision == 3 ? 1 : 0;  // precision 3: the 16-bit GEMM operands are bfloat16 (training only; BASELINE config C2)
    const bool split = train && c->precision >= 1 && D % (pure ? 64 : 32) == 0 && c->cfg.object_token_dim % (pure ? 64 : 32) == 0 && p.M >= g_train_split_min_rows;
    const int lowp_arith = pure ? 2 : 1;
    auto cast_auto = [&](const float* in, int ld, float* out, long long rows, int K, float* scal) -> int {
        return pure ? launch_cast_f16(in, ld, out, K, rows, K, 0.f, scal, s, 13, nullptr, bf) : launch_cast_sp16_auto(in, ld, out, K, rows, K, scal, s);
    };
    auto cast_fixed = [&](const float* in, int ld, float* out, long long rows, int K, float scale, void* side16 = nullptr) -> int {
        return pure ? launch_cast_f16(in, ld, out, K, rows, K, scale, nullptr, s, 13, nullptr, bf) : launch_cast_sp16(in, ld, out, K, rows, K, scale, s, side16);
    };

    // a1: weight standardisation (module/ws.py:9-13), every forward like the reference unless the policy says cached
    if (c->ws_dirty || c->ws_every_forward || train) {
        WsLayer layers[6];
        for (int i = 0; i < 6; ++i) {
            const std::string nm = "short_motion_encoder." + std::to_string(kConvIdx[i]) + ".weight";
            layers[i] = WsLayer{W(nm), c->ws_buf + c->ws_off[i], c->conv[i].cout, c->conv[i].cin, c->conv[i].k};
        }
        SOLA_TRY(launch_ws_standardize(layers, 6, s));
        c->ws16_fmt = split ? (pure ? 2 + bf : 1) : 0;  // ws_buf is new: ws16_buf is current only if it is rewritten below
        if (split)
            for (int i = 0; i < 6; ++i) {
                const int kc = c->conv[i].k * c->conv[i].cin;
                float* dst = pure ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(c->ws16_buf) + c->ws_off[i]) : c->ws16_buf + c->ws_off[i];
                SOLA_TRY(cast_fixed(c->ws_buf + c->ws_off[i], kc, dst, c->conv[i].cout, kc, 1.f));
            }
        c->ws_dirty = false;
    }
    auto lin_idx = [&](const std::string& attn, int proj) -> int {  // attn = "object_lang_align_layers.<l>.<name>"
        const int l = attn[attn.find('.') + 1] - '0';
        int a3 = 0;
        for (int a = 0; a < 3; ++a)
            if (attn.size() >= strlen(kAttnLong[a]) && attn.compare(attn.size() - strlen(kAttnLong[a]), std::string::npos, kAttnLong[a]) == 0) a3 = a;
        return (l * 3 + a3) * 4 + proj;
    };
    auto lin16 = [&](const std::string& attn, int proj) -> const float* {  // split pairs: D*D floats per matrix; plain f16: D*D halfs
        return pure ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(c->lin16_buf) + (size_t)lin_idx(attn, proj) * D * D)
                    : c->lin16_buf + (size_t)lin_idx(attn, proj) * D * D;
    };
    auto lin_inv = [&](const std::string& attn, int proj) -> const float* { return c->scal_buf + 2 * (2 + lin_idx(attn, proj)) + 1; };
    if (split) SOLA_TRY(sola_refresh_lin16(c, s));
    float* const sp_a = split ? buf("sp_a") : nullptr;
    float* const sp_b = split ? buf("sp_b") : nullptr;
    // 16-bit operand modes: every fixed-scale operand cast gets its own slot of the ctx's arena and is listed by its f32 source - the
    // backward's dW products read it instead of casting the activation again (ctx.h; data-dependent scales keep the shared buffers)
    // The split-f16 step ("f16x3") keeps nothing of its split pairs (fetching their hi halves with every other 16-byte chunk slowed the
    // dW kernel's DMA by what the saved casts had cost); its operand casts instead write the hi halves ONCE MORE as plain f16 rows
    // into the arena (side16 below): 2 more bytes per value written here, 6 fewer moved by the backward's X cast.
    const bool keep16 = split && pure && g_train_x16_keep != 0;
    const bool side16_on = split && !pure && g_train_x16_keep != 0 && g_train_dw_f16 != 0;
    c->x16.clear();
    if (keep16 || side16_on) {
        // the arena is the CALLER's (sola_set_x16_arena, round 4: an allocation of its own - half of the free device memory at most, grown with a
        // stream sync and a device-wide free inside the training forward - was memory torch's allocator could neither see nor reclaim);
        // what a forward asked for in total is read with sola_x16_arena_info; what does not fit is cast again by the backward as before
        c->x16_used = 0;
        c->x16_need = 0;
    } else {
        c->x16_used = 0;
        c->x16_need = 0;
    }
    auto side16 = [&](const float* src, long long rows, int cols) -> void* {  // f16x3: where the plain-f16 side copy of `src` goes
        if (!side16_on) return nullptr;
        void* q = c->x16_alloc((size_t)rows * cols * 2);
        if (q) c->x16.push_back(SolaCtx::X16Entry{src, q, cols, 1});
        return q;
    };
    auto slot16 = [&](int which, const float* src, long long rows, int cols) -> float* {  // the operand cast of `src` goes here
        if (keep16)
            if (void* q = c->x16_alloc((size_t)rows * cols * 2)) {
                c->x16.push_back(SolaCtx::X16Entry{src, q, cols, 1 + bf});
                return static_cast<float*>(q);
            }
        return which ? sp_b : sp_a;
    };

    // a2: encoder (module/module.py:74-96,137-140)
    // Operand casts written by the PRODUCING GroupNorm (GroupNormDesc::y_cast; sola_tune "train_gn_cast" 0 = separate cast launches):
    // the cast of a GEMM input was a launch of its own reading the f32 activation back - 24 of them per step
    const int gn_fmt = (split && pure && g_train_gn_cast) ? 2 + bf : 0;  // 16-bit operand modes (split pairs: no gain, kernels.h)
    const float* pc_src[2] = {nullptr, nullptr};  // activations whose operand cast already sits in pc_dst (consumed by the next GEMM)
    float* pc_dst[2] = {nullptr, nullptr};
    const float* x = obj;
    int t_in = T;
    for (int i = 0; i < 6; ++i) {
        const ConvGeom& g = c->conv[i];
        const std::string cp = "short_motion_encoder." + std::to_string(kConvIdx[i]);
        GemmDesc gd{};
        gd.nprob = 1;
        gd.p[0] = GemmProblem{x, c->ws_buf + c->ws_off[i], W(cp + ".bias"), nullptr, buf("conv" + std::to_string(i))};
        gd.M = (int)level_rows(i + 1); gd.N = g.cout; gd.K = g.k * g.cin;
        gd.lda = g.cin; gd.ldr = 0; gd.ldc = g.cout;
        gd.conv = g.k > 1 ? 1 : 0;
        gd.T_in = t_in; gd.T_out = p.Tl[i]; gd.stride = g.stride; gd.pad = g.pad; gd.Cin = g.cin;
        if (rs && gd.conv) { gd.rowmap = rt.rowmap[i]; gd.T_in = 1; gd.T_out = 1; }
        gd.splitk_ws = splitk_ws; gd.splitk_bytes = splitk_bytes;
        if (split && g.cin % (pure ? 64 : 32) == 0) {
            if (i == 0 && !(pure && bf)) {  // the caller's tokens: data-dependent power-of-two scale (forward_fast.hip); bfloat16 has f32's range -
                                            // its cast is the fixed-scale one below, kept for the backward's dW product like every other operand
                SOLA_TRY(cast_auto(x, g.cin, sp_a, level_rows(i), g.cin, c->scal_pair(0)));
                gd.out_scale_dev = c->scal_pair(0) + 1;
                gd.p[0].A = sp_a;
            } else if (pc_src[0] != x) {
                float* dst = slot16(0, x, level_rows(i), g.cin);
                SOLA_TRY(cast_fixed(x, g.cin, dst, level_rows(i), g.cin, 1.f, side16(x, level_rows(i), g.cin)));
                gd.p[0].A = dst;
            } else {
                gd.p[0].A = pc_dst[0];
            }
            pc_src[0] = nullptr;
            gd.p[0].W = pure ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(c->ws16_buf) + c->ws_off[i]) : c->ws16_buf + c->ws_off[i];
            gd.arith = lowp_arith; gd.out_scale = 1.f; gd.bf16 = bf;
        }
        SOLA_TRY(launch_gemm(gd, s));
        if (i < 5) {
            const std::string np = "short_motion_encoder." + std::to_string(kNormIdx[i]);
            GroupNormDesc nd{};
            nd.slice_ws = buf("gn_slots"); nd.slice_ws_bytes = (size_t)p.bufs.at("gn_slots").rows * p.bufs.at("gn_slots").cols * sizeof(float);
            nd.x = buf("conv" + std::to_string(i)); nd.y = buf("act" + std::to_string(i)); nd.y2 = nullptr; nd.pe = nullptr;
            nd.gamma = W(np + ".weight"); nd.beta = W(np + ".bias");
            nd.n_inst = R; nd.inner = 1; nd.outer_stride = p.Tl[i]; nd.inner_stride = 0; nd.tok_stride = 1;
            nd.ntok = p.Tl[i]; nd.C = g.cout; nd.groups = c->cfg.n_groups; nd.eps = 1e-5f; nd.slope = 0.01f; nd.leaky = 1;
            if (rs) nd.units = rt.u_lvl[i + 1];
            if (train) nd.drop = c->enc_drop(i);
            if (gn_fmt && c->conv[i + 1].cin % 64 == 0) {  // conv i+1 takes the cast from here
                float* dst = slot16(0, nd.y, level_rows(i + 1), g.cout);
                nd.y_cast = dst; nd.cast_fmt = gn_fmt;
                pc_src[0] = nd.y; pc_dst[0] = dst;
            }
            SOLA_TRY(launch_group_norm(nd, s));
            x = buf("act" + std::to_string(i));
        } else {
            x = buf("conv5");
        }
        t_in = p.Tl[i];
    }

    // a3: positional table; text tokens ++ negative tokens and their mean (module/module.py:143-147)
    SOLA_TRY(launch_pos_encoding(W("positional_encoding_gaussian_matrix"), D, Tp, c->cfg.max_temporal_length, buf("pe"), s));
    if (rs) SOLA_TRY(launch_lang_concat_ragged(lang, W("negative_token.weight"), buf("lang"), buf("lbar"), B, rt.u_lang, c->cfg.n_negative, D, s));
    else SOLA_TRY(launch_lang_concat(lang, W("negative_token.weight"), buf("lang"), buf("lbar"), B, L, c->cfg.n_negative, D, s));

    // a5: alignment layers (module/module.py:22-52)
    const float scale = 1.0f / sqrtf((float)DH);
    // c16: the outputs are written as BFLOAT16 rows (bf16 steps, sites whose attention kernels read them: site16 below)
    auto linear3 = [&](const float* a0, const float* a1, const float* a2, const std::string& attn, int nprob, int rows,
                       float* o0, float* o1, float* o2, int first_proj, float* a_scal = nullptr, bool c16 = false) -> int {
        static const char* pn[3] = {"q_proj", "k_proj", "v_proj"};
        const float* as[3] = {a0, a1, a2};
        float* os[3] = {o0, o1, o2};
        GemmDesc gd{};
        gd.nprob = nprob;
        for (int j = 0; j < nprob; ++j)
            gd.p[j] = GemmProblem{as[j], W(attn + "." + pn[first_proj + j] + ".weight"),
                                  W(attn + "." + pn[first_proj + j] + ".bias"), nullptr, os[j]};
        gd.M = rows; gd.N = D; gd.K = D; gd.lda = D; gd.ldr = 0; gd.ldc = D;
        gd.splitk_ws = splitk_ws; gd.splitk_bytes = splitk_bytes;
        if (split) {
            float* dst[2] = {sp_a, sp_b};
            const float* src[2] = {nullptr, nullptr};
            int nsrc = 0;
            for (int j = 0; j < nprob; ++j) {
                int at = -1;
                for (int e = 0; e < nsrc; ++e)
                    if (src[e] == as[j]) at = e;
                if (at < 0) {
                    at = nsrc++;
                    src[at] = as[j];
                    if (pc_src[at] == as[j]) dst[at] = pc_dst[at];  // the producing GroupNorm wrote this operand
                    else if (a_scal) SOLA_TRY(cast_auto(as[j], D, dst[at], rows, D, a_scal));
                    else {
                        dst[at] = slot16(at, as[j], rows, D);
                        SOLA_TRY(cast_fixed(as[j], D, dst[at], rows, D, 1.f, side16(as[j], rows, D)));
                    }
                }
                gd.p[j].A = dst[at];
                gd.p[j].W = lin16(attn, first_proj + j);
                gd.p[j].scale_dev = lin_inv(attn, first_proj + j);
            }
            gd.arith = lowp_arith; gd.out_scale = 1.f; gd.bf16 = bf;
            if (a_scal) gd.out_scale_dev = a_scal + 1;
            if (c16) gd.c_f16 = 1;
            pc_src[0] = pc_src[1] = nullptr;
        }
        return launch_gemm(gd, s);
    };
    // bf16 steps: does attention site `ad` keep q / k / v (forward) and dq / dk / dv (backward) as bfloat16 rows?  Both kernels must take
    // the shape (attn_simple.hip's training instantiation; attn_bwd.hip's one-pass kernel)
    c->qkv16.assign((size_t)c->cfg.n_layers * 3, 0);
    auto site16 = [&](const AttnDesc& ad, bool o2l) -> bool {
        if (!(train && split && pure && bf && g_train_bf16_store)) return false;
        // the backward's dW products of these gradients must take the row-major 16-bit kernel (no f32 copy exists to transpose)
        if (!gemm_tn_tr_supported(M, D, D, D, D) || (o2l && !gemm_tn_tr_supported((int)text_rows, D, D, D, D))) return false;
        AttnDesc f = ad;
        f.in_bf16 = 1;
        f.o_cast = reinterpret_cast<void*>(1); f.o_cast_fmt = 3;  // (what attention() below hands the launch)
        if (!attention_in_bf16_supported(f) && !attention_bf16_mfma_supported(f)) return false;
        AttnBwdDesc b{};
        b.ldq = b.ldk = b.ldv = b.ldo = D;
        b.ld_dq = 3 * D; b.ld_dk = b.ld_dv = o2l ? 2 * D : 3 * D;
        b.G = ad.G; b.H = ad.H; b.DH = ad.DH; b.Sq = ad.Sq; b.Sk = ad.Sk; b.inner = ad.inner;
        b.q_outer = ad.q_outer; b.q_inner = ad.q_inner; b.q_rs = ad.q_rs; b.k_outer = ad.k_outer; b.k_inner = ad.k_inner; b.k_rs = ad.k_rs;
        b.q_units = ad.q_units; b.k_units = ad.k_units;
        if (o2l) {  // what backward.hip hands the chunked launch
            b.part = reinterpret_cast<float*>(1);
            b.part_floats = attention_bwd_part_floats((long long)M, B, H, 64);
            b.part_rows = (long long)M;
        }
        return attention_bwd_bf16_supported(b);
    };
    // The attention output's operand cast for the out-projection is written by the attention kernel itself where its shape can
    // (AttnDesc::o_cast, round 4; sola_tune "train_attn_cast" 0 = always the separate cast launch)
    const float* ac_src = nullptr;
    float* ac_dst = nullptr;
    void* ac_side = nullptr;
    bool ac_done = false;
    int o16_site = -1;  // set by the layer loop in front of a site's attention() call: index l * 3 + a of c->attn_o16
    c->attn_o16.assign((size_t)c->cfg.n_layers * 3, 0);
    auto attention = [&](AttnDesc& ad) -> int {
        ac_src = nullptr; ac_done = false;
        // measured on the 64-sample ragged mix (same box): f16 operands 22.67 -> 22.55 ms per step (attention +0.10 ms, cast launches -0.26 ms);
        // split pairs + side copy (10 instead of 4 bytes per element out of a latency-bound kernel) 31.76 -> 31.78 ms: no gain, so the
        // split-f16 step keeps its cast launches unless the key is 2
        if (split && (pure ? g_train_attn_cast != 0 : g_train_attn_cast == 2) && ad.ldo == D) {
            ac_src = ad.o;
            ac_dst = slot16(0, ad.o, M, D);
            ac_side = side16(ad.o, M, D);
            ad.o_cast = ac_dst; ad.o_side = ac_side; ad.o_cast_fmt = pure ? 2 + bf : 1; ad.o_cast_done = &ac_done;
            // bf16 steps (train_bf16_store 3): where the bf16-MFMA kernel writes the out-projection's operand rows itself and they have an arena
            // slot of their own, the f32 rows are not written at all - the out-projection, its weight gradient and the attention backward
            // (D = dO . O) read the bfloat16 rows (4 of the launch's 12 bytes per element)
            if (o16_site >= 0 && ad.in_bf16 && g_train_bf16_store >= 3 && ac_dst != sp_a && ac_dst != sp_b && ad.o_cast_fmt == 3 &&
                attention_bf16_mfma_supported(ad) && attention_bwd_dout_bf16_enabled()) {
                ad.o = nullptr;
                c->attn_o16[(size_t)o16_site] = 1;
            }
        }
        o16_site = -1;
        return launch_attention(ad, s);
    };
    // r16 != null (bf16 steps, sola_tune "train_bf16_store" 2): the residual is read from the sub-block input's bfloat16 operand copy and
    // the pre-norm rows are written as bfloat16 - 4 instead of 8 bytes per element through the epilogue of a K = 1024 GEMM that its f32
    // epilogue traffic bounds (profiles/r06_train_ragged_bf16.txt)
    auto out_proj = [&](const std::string& attn, const float* ao, const float* resid, float* res, const void* r16 = nullptr) -> int {
        GemmDesc gd{};
        gd.nprob = 1;
        gd.p[0] = GemmProblem{ao, W(attn + ".out_proj.weight"), W(attn + ".out_proj.bias"), resid, res};
        gd.M = M; gd.N = D; gd.K = D; gd.lda = D; gd.ldr = D; gd.ldc = D;
        gd.splitk_ws = splitk_ws; gd.splitk_bytes = splitk_bytes;
        if (split && r16) { gd.p[0].R = static_cast<const float*>(r16); gd.r_f16 = 1; gd.c_f16 = 1; }
        if (split) {
            float* dst;
            if (ac_src == ao) {  // attention() reserved the operand's place; the attention kernel wrote it unless its shape cannot
                dst = ac_dst;
                if (!ac_done) SOLA_TRY(cast_fixed(ao, D, dst, M, D, 1.f, ac_side));
            } else {
                dst = slot16(0, ao, M, D);
                SOLA_TRY(cast_fixed(ao, D, dst, M, D, 1.f, side16(ao, M, D)));
            }
            ac_src = nullptr;
            gd.p[0].A = dst; gd.p[0].W = lin16(attn, 3); gd.p[0].scale_dev = lin_inv(attn, 3);
            gd.arith = lowp_arith; gd.out_scale = 1.f; gd.bf16 = bf;
        }
        return launch_gemm(gd, s);
    };
    // next_src0 / next_src1: the FIRST / SECOND distinct input of the projection launch that follows (linear3 casts them into sp_a / sp_b in
    // that order): when one of them is this norm's y or y2, the norm writes the cast itself
    int gn_layer = 0;  // the layer whose norms the loop below is launching
    auto gn = [&](const std::string& lp, int idx, const float* res, float* y, float* y2, int n_inst, int inner,
                  long long outer, long long inner_stride, long long tok_stride, int ntok, const int4* units = nullptr,
                  const float* next_src0 = nullptr, const float* next_src1 = nullptr, bool res_bf16 = false) -> int {
        GroupNormDesc nd{};
        nd.in_f16 = res_bf16 ? 2 : 0;
        if (gn_fmt) {
            const float* nxt[2] = {next_src0, next_src1};
            const long long y_rows = std::max<long long>(M, text_rows);  // the norm's token rows, bounded from above
            for (int e = 0; e < 2; ++e) {
                if (!nxt[e]) continue;
                if (nxt[e] == y) { float* dst = slot16(e, y, y_rows, D); nd.y_cast = dst; pc_src[e] = y; pc_dst[e] = dst; }
                else if (nxt[e] == y2) { float* dst = slot16(e, y2, y_rows, D); nd.y2_cast = dst; pc_src[e] = y2; pc_dst[e] = dst; }
            }
            nd.cast_fmt = gn_fmt;
        }
            nd.slice_ws = buf("gn_slots"); nd.slice_ws_bytes = (size_t)p.bufs.at("gn_slots").rows * p.bufs.at("gn_slots").cols * sizeof(float);
        nd.x = res; nd.y = y; nd.y2 = y2; nd.pe = y2 ? buf("pe") : nullptr;
        nd.gamma = W(lp + "norm." + std::to_string(idx) + ".weight");
        nd.beta = W(lp + "norm." + std::to_string(idx) + ".bias");
        nd.n_inst = n_inst; nd.inner = inner; nd.outer_stride = outer; nd.inner_stride = inner_stride;
        nd.tok_stride = tok_stride; nd.ntok = ntok; nd.C = D; nd.groups = c->cfg.n_groups_module;
        nd.eps = 1e-5f; nd.slope = 0.f; nd.leaky = 0;
        nd.units = units;
        if (train && idx == 2 && g_train_gn_stats) {
            const int l = gn_layer;
            int wrote = 0;
            nd.stats_out = buf("l" + std::to_string(l) + "_gn2st");
            nd.stats_written = &wrote;
            SOLA_TRY(launch_group_norm(nd, s));
            if ((size_t)l < c->gn2_stats.size()) c->gn2_stats[l] = (char)wrote;
            return SOLA_OK;
        }
        return launch_group_norm(nd, s);
    };
    c->gn2_stats.assign((size_t)c->cfg.n_layers, 0);
    c->res16.assign((size_t)c->cfg.n_layers * 3, 0);
    // the bf16 operand copy of a sub-block's input, if it sits in the kept-operand arena (a shared cast buffer is overwritten before the
    // out-projection runs) - then the sub-block's pre-norm rows are bfloat16 (out_proj / gn above)
    auto resid16 = [&](const float* resid) -> const void* {
        if (!(train && split && pure && bf && g_train_bf16_store >= 2 && keep16 && D % 8 == 0)) return nullptr;
        return c->x16_find(resid, D, 1 + bf);
    };
    const float* xin = buf("conv5");
    for (int l = 0; l < c->cfg.n_layers; ++l) {
        const std::string lp = "object_lang_align_layers." + std::to_string(l) + ".";
        gn_layer = l;
        const std::string ls = "l" + std::to_string(l);
        auto ab = [&](int a, const char* what) { return buf(abuf(train, l, kAttnShort[a], what)); };
        auto lse = [&](int a) -> float* { return train ? ab(a, "lse") : nullptr; };
        float* x_obj = buf(ls + "_obj");
        float* x_pe = buf(ls + "_xpe");
        float* x_mot = buf(ls + "_motion");
        float* x_o2l = buf(ls + "_o2l");
        // (i) inter-object attention over the N tracks of each (b, t'): module.py:31-35
        {
            AttnDesc ad{ab(0, "q"), ab(0, "k"), ab(0, "v"), ab(0, "attn"), D, D, D, D, B * Tp, H, DH, N, N, Tp,
                        (long long)N * Tp, 1, Tp, (long long)N * Tp, 1, Tp, scale, lse(0)};
            if (rs) { ad.G = rt.sumTpS; ad.inner = 1; ad.q_units = rt.u_st; }
            if (train) ad.drop = c->attn_drop(l, 0);
            const bool s16 = site16(ad, false);
            SOLA_TRY(linear3(xin, xin, xin, lp + "obj_attn", 3, M, ab(0, "q"), ab(0, "k"), ab(0, "v"), 0, nullptr, s16));
            ad.in_bf16 = s16 ? 1 : 0;
            c->qkv16[(size_t)l * 3 + 0] = s16;
            o16_site = l * 3 + 0;
            SOLA_TRY(attention(ad));
        }
        const void* r16_0 = resid16(xin);
        c->res16[(size_t)l * 3 + 0] = r16_0 != nullptr;
        SOLA_TRY(out_proj(lp + "obj_attn", ab(0, "attn"), xin, ab(0, "res"), r16_0));
        if (rs) SOLA_TRY(gn(lp, 0, ab(0, "res"), x_obj, x_pe, rt.sumTpS, 1, 0, 0, 1, N, rt.u_st, x_pe, x_obj, r16_0 != nullptr));
        else SOLA_TRY(gn(lp, 0, ab(0, "res"), x_obj, x_pe, B * Tp, Tp, (long long)N * Tp, 1, Tp, N, nullptr, x_pe, x_obj, r16_0 != nullptr));
        // (ii) motion attention over T' per track, PE on q and k only: module.py:38-43
        {
            AttnDesc ad{ab(1, "q"), ab(1, "k"), ab(1, "v"), ab(1, "attn"), D, D, D, D, B * N, H, DH, Tp, Tp, 1,
                        (long long)Tp, 0, 1, (long long)Tp, 0, 1, scale, lse(1)};
            if (rs) { ad.G = rt.sumNS; ad.q_units = rt.u_strk; }
            if (train) ad.drop = c->attn_drop(l, 1);
            const bool s16 = site16(ad, false);
            SOLA_TRY(linear3(x_pe, x_pe, x_obj, lp + "motion_attn", 3, M, ab(1, "q"), ab(1, "k"), ab(1, "v"), 0, nullptr, s16));
            ad.in_bf16 = s16 ? 1 : 0;
            c->qkv16[(size_t)l * 3 + 1] = s16;
            o16_site = l * 3 + 1;
            SOLA_TRY(attention(ad));
        }
        const void* r16_1 = resid16(x_obj);
        c->res16[(size_t)l * 3 + 1] = r16_1 != nullptr;
        SOLA_TRY(out_proj(lp + "motion_attn", ab(1, "attn"), x_obj, ab(1, "res"), r16_1));
        if (rs) SOLA_TRY(gn(lp, 1, ab(1, "res"), x_mot, nullptr, rt.sumNS, 1, 0, 0, 1, Tp, rt.u_strk, x_mot, nullptr, r16_1 != nullptr));
        else SOLA_TRY(gn(lp, 1, ab(1, "res"), x_mot, nullptr, B * N, 1, Tp, 0, 1, Tp, nullptr, x_mot, nullptr, r16_1 != nullptr));
        // (iii) object -> language cross attention: module.py:46-50
        {
            AttnDesc ad{ab(2, "q"), ab(2, "lk"), ab(2, "lv"), ab(2, "attn"), D, D, D, D, B, H, DH, N * Tp, Wn, 1,
                        (long long)N * Tp, 0, 1, (long long)Wn, 0, 1, scale, lse(2)};
            if (rs) { ad.Sq = rt.maxRowsSample; ad.q_units = rt.u_smp; ad.k_units = rt.u_langk; }
            if (train) ad.drop = c->attn_drop(l, 2);
            const bool s16 = site16(ad, true);
            SOLA_TRY(linear3(x_mot, nullptr, nullptr, lp + "object2lang_attn", 1, M, ab(2, "q"), nullptr, nullptr, 0, nullptr, s16));
            SOLA_TRY(linear3(buf("lang"), buf("lang"), nullptr, lp + "object2lang_attn", 2, (int)text_rows, ab(2, "lk"), ab(2, "lv"), nullptr, 1,
                             split ? c->scal_pair(1) : nullptr, s16));
            ad.in_bf16 = s16 ? 1 : 0;
            c->qkv16[(size_t)l * 3 + 2] = s16;
            o16_site = l * 3 + 2;
            SOLA_TRY(attention(ad));
        }
        const void* r16_2 = resid16(x_mot);
        c->res16[(size_t)l * 3 + 2] = r16_2 != nullptr;
        SOLA_TRY(out_proj(lp + "object2lang_attn", ab(2, "attn"), x_mot, ab(2, "res"), r16_2));
        // the next layer's q / k / v projections read x_o2l: its norm writes their operand (nothing reads it behind the last layer)
        const float* nxt_in = l + 1 < c->cfg.n_layers ? x_o2l : nullptr;
        if (rs) SOLA_TRY(gn(lp, 2, ab(2, "res"), x_o2l, nullptr, B, 1, 0, 0, 1, rt.maxRowsSample, rt.u_smp, nxt_in, nullptr, r16_2 != nullptr));
        else SOLA_TRY(gn(lp, 2, ab(2, "res"), x_o2l, nullptr, B, 1, (long long)N * Tp, 0, 1, N * Tp, nullptr, nxt_in, nullptr, r16_2 != nullptr));
        xin = x_o2l;
    }

    // a6: score head (module/module.py:152-160)
    HeadDesc hd{xin, buf("lbar"), score_map, score_tokens, B, N, Tp, D};
    if (rs) { hd.B = 1; hd.N = rt.sumNS; hd.units = rt.u_strk; }
    SOLA_TRY(launch_score_head(hd, s));
    c->last = p;
    c->last_rag = rt;
    c->last_ws = workspace;
    c->last_obj = train ? obj : nullptr;
    return SOLA_OK;
}
