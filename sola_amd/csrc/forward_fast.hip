// Inference forward in the split-f16 precision mode (sola_set_precision(ctx, 1)).
//
// Same network, same kernels and the same single [B,N,T',D] layout as forward.hip; what changes is the arithmetic of
// the dense contractions (98 % of the FLOPs).  f32 MFMA runs at 1/16 of the f16 MFMA rate on gfx950 and there is no
// xf32, so every GEMM operand is kept as an (f16 hi, f16 lo) pair in the 4 bytes of the f32 it replaces and each
// product is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with f32 accumulation (gemm.hip ARITH 1):
// ~22-bit products, 3/16 of the matrix-pipe time, identical bytes.  The producers emit the split format directly -
// GroupNorm epilogue, attention epilogue - so only three tensors are converted by a separate pass (the input tokens,
// the conv5 output that layer 0 consumes, the text tokens).  Softmax, GroupNorm statistics, the score head and the
// losses stay in f32.  Parity: the same golden-vector tests and 1e-3 bound as the f32 mode (tests/test_gpu_fast.py).
#include <math.h>

#include <algorithm>

#include "ctx.h"

// Split-f16 copies of the 12 * n_layers projection weights, each with its own power-of-two scale (max|w| -> [2^13, 2^14),
// found on the device: a trained matrix may be far from the U(-1/32, 1/32) of the default init, and a few outliers must not
// push the rest into f16 subnormals - the pair format keeps 22 bits for everything within 2^-16 of the largest entry), plus
// the weight-time range check of the activations the GroupNorms will emit (kernels.h: launch_norm_range_check).
int sola_refresh_lin16(SolaCtx* c, hipStream_t s) {
    if (!c->lin16_dirty) return SOLA_OK;
    SOLA_ARG(c->lin16_buf && c->scal_buf, "split-f16 weights requested before sola_set_precision(ctx, 1)");
    static const char* pn[4] = {"q_proj", "k_proj", "v_proj", "out_proj"};
    const int D = c->cfg.lang_token_dim;
    std::vector<const float*> in;
    std::vector<float*> out;
    for (int l = 0; l < c->cfg.n_layers; ++l)
        for (int a = 0; a < 3; ++a)
            for (int j = 0; j < 4; ++j) {
                const std::string nm = "object_lang_align_layers." + std::to_string(l) + "." + kAttnLong[a] + "." + pn[j] + ".weight";
                const float* w = ctx_weight(c, nm);
                if (!w) {
                    sola_set_error("forward: weight '%s' has not been set", nm.c_str());
                    return SOLA_ERR_WEIGHT;
                }
                in.push_back(w);
                out.push_back(c->lin16_buf + ((size_t)(l * 3 + a) * 4 + j) * D * D);
            }
    if (c->precision >= 2) {  // 16-bit storage mode / 16-bit GEMM operands: plain f16 (precision 3: bfloat16) copies, same per-matrix
        std::vector<void*> outh;  // scales, packed in the first half of the buffer
        for (size_t i = 0; i < out.size(); ++i) outh.push_back(reinterpret_cast<_Float16*>(c->lin16_buf) + i * (size_t)D * D);
        SOLA_TRY(launch_cast_f16_auto_multi(in.data(), outh.data(), (int)in.size(), D, D, c->scal_pair(2), s, c->precision == 3 ? 1 : 0));
    } else {
        SOLA_TRY(launch_cast_sp16_auto_multi(in.data(), out.data(), (int)in.size(), D, D, c->scal_pair(2), s));
    }
    std::vector<NormPair> norms;
    for (int i = 0; i < 5; ++i) {
        const std::string np = "short_motion_encoder." + std::to_string(kNormIdx[i]);
        norms.push_back(NormPair{ctx_weight(c, np + ".weight"), ctx_weight(c, np + ".bias"), c->conv[i].cout});
    }
    for (int l = 0; l < c->cfg.n_layers; ++l)
        for (int j = 0; j < 3; ++j) {
            const std::string np = "object_lang_align_layers." + std::to_string(l) + ".norm." + std::to_string(j);
            norms.push_back(NormPair{ctx_weight(c, np + ".weight"), ctx_weight(c, np + ".bias"), D});
        }
    for (const NormPair& n : norms)
        if (!n.gamma || !n.beta) {
            sola_set_error("forward: a GroupNorm weight has not been set");
            return SOLA_ERR_WEIGHT;
        }
    SOLA_HIP(hipMemsetAsync(c->guard + 1, 0, sizeof(int), s));
    for (size_t i0 = 0; i0 < norms.size(); i0 += 32)
        SOLA_TRY(launch_norm_range_check(norms.data() + i0, (int)std::min<size_t>(32, norms.size() - i0), c->guard + 1, s));
    c->lin16_dirty = false;
    return SOLA_OK;
}

// sola_tune "attn_split_min_keys": units with more keys than this take the split-f16 MFMA attention on q/k/v the projection GEMMs
// wrote as split pairs.  Rounds 1-2: 64.  Round 3: behind the one-pass / register-only f32 shapes (attn_simple.hip, attn_reg.hip)
// attn.hip's kernel for split inputs lost at 80 keys (401 vs 280 us per launch in the bench) and tied at 128, so the threshold went
// to 128; with the high-occupancy shape for split inputs (attn_fwd_spin_kernel: 262 vs 362 us at 128 tracks, 270 vs 285 at 80, 216
// vs 256 at 64) the attention wins from 64 keys on, but the q/k/v GEMM pays ~70 us per launch for the split-pair epilogue: net
// gain at 128 tracks (14.41 -> 14.30 ms per step, attention 0.43 -> 0.49 of the HBM peak), net loss at 80 and 64.  96.
int g_attn_split_min_keys = 96;
void sola_attn_set_split_min_keys(int v) { g_attn_split_min_keys = v; }

int g_lang_shared_neg = 1;  // sola_tune "lang_shared_neg": 0 = the negative tokens repeated per sample through the text-side projections (A/B)

int sola_forward_fast_impl(SolaCtx* c, const float* obj, const float* lang, int B, int N, int T, int L, float* score_map,
                           float* score_tokens, void* workspace, size_t ws_bytes, hipStream_t s) {
    SOLA_ARG(c && obj && lang && score_map && score_tokens && workspace, "forward: null argument");
    SOLA_ARG(B > 0 && N > 0 && T > 0 && L >= 1, "forward: bad sizes B=%d N=%d T=%d L=%d", B, N, T, L);
    SOLA_ARG(c->cfg.object_token_dim % 8 == 0 && (c->cfg.lang_token_dim / c->cfg.n_groups_module) % 8 == 0 &&
                 (2 * c->cfg.object_token_dim / c->cfg.n_groups) % 8 == 0 && (c->cfg.lang_token_dim / c->cfg.n_groups) % 8 == 0,
             "split-f16 mode needs channel counts per GroupNorm group that are multiples of 8");
    for (const Weight& w : c->weights)
        if (!w.ptr) {
            sola_set_error("forward: weight '%s' has not been set", w.name.c_str());
            return SOLA_ERR_WEIGHT;
        }
    Plan p = make_plan(c, B, N, T, L, false);
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
    const int R = B * N;
    auto lin16 = [&](int layer, int attn, int proj) { return c->lin16_buf + ((size_t)(layer * 3 + attn) * 4 + proj) * D * D; };

    // ---- weights: standardise the conv weights (module/ws.py:9-13) and refresh the split-f16 copies
    if (c->ws_dirty || c->ws_every_forward || c->ws16_fmt != 1) {
        WsLayer layers[6];
        for (int i = 0; i < 6; ++i) {
            const std::string nm = "short_motion_encoder." + std::to_string(kConvIdx[i]) + ".weight";
            layers[i] = WsLayer{W(nm), c->ws_buf + c->ws_off[i], c->conv[i].cout, c->conv[i].cin, c->conv[i].k};
        }
        SOLA_TRY(launch_ws_standardize(layers, 6, s));
        for (int i = 0; i < 6; ++i) {
            const int kc = c->conv[i].k * c->conv[i].cin;
            SOLA_TRY(launch_cast_sp16(c->ws_buf + c->ws_off[i], kc, c->ws16_buf + c->ws_off[i], kc, c->conv[i].cout, kc, 1.f, s));
        }
        c->ws_dirty = false;
        c->ws16_fmt = 1;
    }
    // The projection weights are used as they are by the reference (no per-forward transform), so their split copies
    // are refreshed only when a weight pointer or value changed (sola_set_weight / sola_weights_changed).
    SOLA_TRY(sola_refresh_lin16(c, s));
    SOLA_HIP(hipMemsetAsync(c->guard, 0, sizeof(int), s));  // per-call range guard word (ctx.h)

    // ---- encoder: split-f16 activations between the stages, f32 conv outputs into GroupNorm.
    // The caller's tokens come with an unknown magnitude (SAM2 memory-attention features here, anything elsewhere): their
    // largest entry is mapped into [2^13, 2^14) by a power of two found on the device and conv0's epilogue undoes it.
    SOLA_TRY(launch_cast_sp16_auto(obj, c->cfg.object_token_dim, buf("obj_sp"), c->cfg.object_token_dim, (long long)R * T,
                                   c->cfg.object_token_dim, c->scal_pair(0), s));
    const float* x = buf("obj_sp");
    int t_in = T;
    for (int i = 0; i < 6; ++i) {
        const ConvGeom& g = c->conv[i];
        const std::string cp = "short_motion_encoder." + std::to_string(kConvIdx[i]);
        GemmDesc gd{};
        gd.nprob = 1;
        // conv5 (no norm behind it) feeds layer 0 both as the projections' A operand and as the first residual: its
        // epilogue writes the split-f16 form directly, no f32 copy and no cast pass
        const bool last_conv = i == 5 && g.cout % 8 == 0;
        gd.p[0] = GemmProblem{x, c->ws16_buf + c->ws_off[i], W(cp + ".bias"), nullptr, last_conv ? buf("conv5_sp") : buf("conv" + std::to_string(i))};
        gd.c_sp16 = last_conv ? 1 : 0;
        gd.M = R * p.Tl[i]; gd.N = g.cout; gd.K = g.k * g.cin;
        gd.lda = g.cin; gd.ldr = 0; gd.ldc = g.cout;
        gd.conv = g.k > 1 ? 1 : 0;
        gd.T_in = t_in; gd.T_out = p.Tl[i]; gd.stride = g.stride; gd.pad = g.pad; gd.Cin = g.cin;
        gd.arith = 1; gd.out_scale = 1.f;
        if (i == 0) gd.out_scale_dev = c->scal_pair(0) + 1;
        gd.guard = c->guard;
        gd.splitk_ws = splitk_ws; gd.splitk_bytes = splitk_bytes;
        // conv0-2 at GPU-filling batches: the norm behind the conv (64 channels per group, 16 / 8 / 4 tokens per instance) is
        // applied in the GEMM's epilogue and the activation written as split-f16 pairs directly (gemm_glds.hip, GNF)
        bool fused_norm = false;
        if (i < 5) {
            GemmDesc probe = gd;
            probe.c_sp16 = 1;
            if (gemm_gn_fusable(probe, g.cout / c->cfg.n_groups, p.Tl[i])) {
                const std::string np = "short_motion_encoder." + std::to_string(kNormIdx[i]);
                gd.c_sp16 = 1;
                gd.p[0].C = buf("act" + std::to_string(i));
                gd.gn_gamma = W(np + ".weight"); gd.gn_beta = W(np + ".bias");
                gd.gn_tokens = p.Tl[i]; gd.gn_eps = 1e-5f; gd.gn_slope = 0.01f;
                fused_norm = true;
            }
        }
        SOLA_TRY(launch_gemm(gd, s));
        if (fused_norm) {
            x = buf("act" + std::to_string(i));
        } else if (i < 5) {
            const std::string np = "short_motion_encoder." + std::to_string(kNormIdx[i]);
            GroupNormDesc nd{};
            nd.slice_ws = buf("gn_slots"); nd.slice_ws_bytes = (size_t)p.bufs.at("gn_slots").rows * p.bufs.at("gn_slots").cols * sizeof(float);
            nd.x = buf("conv" + std::to_string(i)); nd.y = buf("act" + std::to_string(i)); nd.y2 = nullptr; nd.pe = nullptr;
            nd.gamma = W(np + ".weight"); nd.beta = W(np + ".bias");
            nd.n_inst = R; nd.inner = 1; nd.outer_stride = p.Tl[i]; nd.inner_stride = 0; nd.tok_stride = 1;
            nd.ntok = p.Tl[i]; nd.C = g.cout; nd.groups = c->cfg.n_groups; nd.eps = 1e-5f; nd.slope = 0.01f; nd.leaky = 1;
            nd.out_sp16 = 1;
            nd.guard = c->guard;
            SOLA_TRY(launch_group_norm(nd, s));
            x = buf("act" + std::to_string(i));
        }
        t_in = p.Tl[i];
    }
    const bool conv5_split = c->conv[5].cout % 8 == 0;
    if (!conv5_split) SOLA_TRY(launch_cast_sp16(buf("conv5"), D, buf("conv5_sp"), D, M, D, 1.f, s));

    SOLA_TRY(launch_pos_encoding(W("positional_encoding_gaussian_matrix"), D, Tp, c->cfg.max_temporal_length, buf("pe"), s));
    // round 5 (sola_tune "lang_shared_neg", default 1): the negative tokens' key / value rows are the same for every sample - project them
    // once (B * L + n_neg rows through the two text-side GEMMs of a layer instead of B * (L + n_neg)) and let the object -> language
    // attention read the shared rows behind each sample's L own ones.  Same products per row, same bits - from 1024 text rows on: below
    // that the projection's 64x64 kernel splits K by the size of its grid, and the two forms would sum in different orders (nothing to gain there).
    bool shared_neg = false;
    if (g_lang_shared_neg && c->cfg.n_negative > 0 && Wn <= g_attn_split_min_keys && (long long)B * L + c->cfg.n_negative >= 1024) {
        AttnDesc probe{};
        probe.G = B; probe.H = H; probe.DH = DH; probe.Sq = N * Tp; probe.Sk = Wn; probe.inner = 1; probe.k_private = L;
        shared_neg = attention_shared_keys_supported(probe);
    }
    const long long lang_rows = shared_neg ? (long long)B * L + c->cfg.n_negative : (long long)B * Wn;
    if (shared_neg) SOLA_TRY(launch_lang_concat_shared(lang, W("negative_token.weight"), buf("lang"), buf("lbar"), B, L, c->cfg.n_negative, D, s));
    else SOLA_TRY(launch_lang_concat(lang, W("negative_token.weight"), buf("lang"), buf("lbar"), B, L, c->cfg.n_negative, D, s));
    SOLA_TRY(launch_cast_sp16_auto(buf("lang"), D, buf("lang_sp"), D, lang_rows, D, c->scal_pair(1), s));

    const float scale = 1.0f / sqrtf((float)DH);
    auto linear3 = [&](const float* a0, const float* a1, const float* a2, int layer, int attn, int nprob, int rows, float* o0,
                       float* o1, float* o2, int first_proj, int out_sp16, const float* a_inv_scale = nullptr) -> int {
        static const char* pn[3] = {"q_proj", "k_proj", "v_proj"};
        const std::string an = "object_lang_align_layers." + std::to_string(layer) + "." + kAttnLong[attn];
        const float* as[3] = {a0, a1, a2};
        float* os[3] = {o0, o1, o2};
        GemmDesc gd{};
        gd.nprob = nprob;
        for (int j = 0; j < nprob; ++j)
            gd.p[j] = GemmProblem{as[j], lin16(layer, attn, first_proj + j), W(an + "." + pn[first_proj + j] + ".bias"), nullptr, os[j],
                                  c->lin_inv_scale(layer, attn, first_proj + j)};
        gd.M = rows; gd.N = D; gd.K = D; gd.lda = D; gd.ldr = 0; gd.ldc = D;
        gd.arith = 1; gd.out_scale = 1.f; gd.out_scale_dev = a_inv_scale; gd.c_sp16 = out_sp16; gd.guard = c->guard;
        gd.splitk_ws = splitk_ws; gd.splitk_bytes = splitk_bytes;
        return launch_gemm(gd, s);
    };
    auto out_proj = [&](int layer, int attn, const float* resid, int resid_sp16) -> int {
        const std::string an = "object_lang_align_layers." + std::to_string(layer) + "." + kAttnLong[attn];
        GemmDesc gd{};
        gd.nprob = 1;
        gd.p[0] = GemmProblem{buf("attn"), lin16(layer, attn, 3), W(an + ".out_proj.bias"), resid, buf("res"), c->lin_inv_scale(layer, attn, 3)};
        gd.M = M; gd.N = D; gd.K = D; gd.lda = D; gd.ldr = D; gd.ldc = D;
        gd.arith = 1; gd.out_scale = 1.f; gd.r_sp16 = resid_sp16;
        gd.splitk_ws = splitk_ws; gd.splitk_bytes = splitk_bytes;
        return launch_gemm(gd, s);
    };
    auto gn = [&](const std::string& lp, int idx, float* y, float* y2, int sp16, int n_inst, int inner, long long outer,
                  long long inner_stride, long long tok_stride, int ntok) -> int {
        GroupNormDesc nd{};
            nd.slice_ws = buf("gn_slots"); nd.slice_ws_bytes = (size_t)p.bufs.at("gn_slots").rows * p.bufs.at("gn_slots").cols * sizeof(float);
        nd.x = buf("res"); nd.y = y; nd.y2 = y2; nd.pe = y2 ? buf("pe") : nullptr;
        nd.gamma = W(lp + "norm." + std::to_string(idx) + ".weight");
        nd.beta = W(lp + "norm." + std::to_string(idx) + ".bias");
        nd.n_inst = n_inst; nd.inner = inner; nd.outer_stride = outer; nd.inner_stride = inner_stride;
        nd.tok_stride = tok_stride; nd.ntok = ntok; nd.C = D; nd.groups = c->cfg.n_groups_module;
        nd.eps = 1e-5f; nd.slope = 0.f; nd.leaky = 0; nd.out_sp16 = sp16; nd.guard = c->guard;
        return launch_group_norm(nd, s);
    };
    auto attention = [&](const float* q, const float* k, const float* v, int G, int Sq, int Sk, int inner, long long qo,
                         long long qi, long long qr, long long ko, long long ki, long long kr, int in_sp16, int k_private = 0,
                         long long k_shared_row = 0) -> int {
        AttnDesc ad{q, k, v, buf("attn"), D, D, D, D, G, H, DH, Sq, Sk, inner, qo, qi, qr, ko, ki, kr, scale, nullptr};
        ad.k_private = k_private;
        ad.k_shared_row = k_shared_row;
        ad.o_sp16 = 1;
        ad.in_sp16 = in_sp16;
        ad.guard = c->guard;
        ad.split_math = 1;
        return launch_attention(ad, s);
    };
    // q/k/v leave the projection GEMM already split when the attention that reads them runs the split-f16 MFMA shape.
    // Measured (tools/attn_probe.py): with <= 64 keys per unit the exact-f32 MFMA kernel is as fast or faster (the kernel
    // is then bound by latency and LDS traffic, not by the matrix pipe); with 65..128 keys the split shape wins by 14 %.
    const int obj_sp = (N > g_attn_split_min_keys && N > 16 && DH % 16 == 0) ? 1 : 0;
    const int o2l_sp = (Wn > g_attn_split_min_keys && DH % 16 == 0) ? 1 : 0;

    const float* xin = buf("conv5_sp");  // split-f16 A operand of the layer
    const float* xres = conv5_split ? buf("conv5_sp") : buf("conv5");  // residual of the first sub-block
    int xres_sp = conv5_split ? 1 : 0;
    for (int l = 0; l < c->cfg.n_layers; ++l) {
        const std::string lp = "object_lang_align_layers." + std::to_string(l) + ".";
        const std::string ls = "l" + std::to_string(l);
        const bool last = l + 1 == c->cfg.n_layers;
        float *q = buf("q"), *k = buf("k"), *v = buf("v");
        float* x_obj = buf(ls + "_obj");
        float* x_pe = buf(ls + "_xpe");
        float* x_mot = buf(ls + "_motion");
        float* x_o2l = buf(ls + "_o2l");
        // (i) inter-object attention (module.py:31-35)
        SOLA_TRY(linear3(xin, xin, xin, l, 0, 3, M, q, k, v, 0, obj_sp));
        SOLA_TRY(attention(q, k, v, B * Tp, N, N, Tp, (long long)N * Tp, 1, Tp, (long long)N * Tp, 1, Tp, obj_sp));
        SOLA_TRY(out_proj(l, 0, xres, xres_sp));
        SOLA_TRY(gn(lp, 0, x_obj, x_pe, 1, B * Tp, Tp, (long long)N * Tp, 1, Tp, N));
        // (ii) motion attention (module.py:38-43)
        const int mot_sp = (Tp > g_attn_split_min_keys && Tp > 16 && DH % 16 == 0) ? 1 : 0;
        SOLA_TRY(linear3(x_pe, x_pe, x_obj, l, 1, 3, M, q, k, v, 0, mot_sp));
        SOLA_TRY(attention(q, k, v, B * N, Tp, Tp, 1, (long long)Tp, 0, 1, (long long)Tp, 0, 1, mot_sp));
        SOLA_TRY(out_proj(l, 1, x_obj, 1));
        SOLA_TRY(gn(lp, 1, x_mot, nullptr, 1, B * N, 1, Tp, 0, 1, Tp));
        // (iii) object -> language attention (module.py:46-50)
        SOLA_TRY(linear3(x_mot, nullptr, nullptr, l, 2, 1, M, q, nullptr, nullptr, 0, o2l_sp));
        SOLA_TRY(linear3(buf("lang_sp"), buf("lang_sp"), nullptr, l, 2, 2, (int)lang_rows, buf("lk"), buf("lv"), nullptr, 1, o2l_sp, c->scal_pair(1) + 1));
        if (shared_neg) SOLA_TRY(attention(q, buf("lk"), buf("lv"), B, N * Tp, Wn, 1, (long long)N * Tp, 0, 1, (long long)L, 0, 1, o2l_sp, L, (long long)B * L));
        else SOLA_TRY(attention(q, buf("lk"), buf("lv"), B, N * Tp, Wn, 1, (long long)N * Tp, 0, 1, (long long)Wn, 0, 1, o2l_sp));
        SOLA_TRY(out_proj(l, 2, x_mot, 1));
        SOLA_TRY(gn(lp, 2, x_o2l, nullptr, last ? 0 : 1, B, 1, (long long)N * Tp, 0, 1, N * Tp));  // the score head reads f32
        xin = x_o2l;
        xres = x_o2l;
        xres_sp = 1;
    }
    HeadDesc hd{xin, buf("lbar"), score_map, score_tokens, B, N, Tp, D};
    SOLA_TRY(launch_score_head(hd, s));
    c->last = p;
    c->last_obj = nullptr;
    return SOLA_OK;
}
