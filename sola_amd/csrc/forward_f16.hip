// Inference forward in the 16-bit storage mode (sola_set_precision(ctx, 2)): BASELINE configs C2 / C4 name bf16 / fp16 runs of
// this path (the reference's only mixed-precision site is track_generation/generate_tokens_grid.py:84-88).
//
// Same network, same single [B,N,T',D] layout and the same orchestration as forward_fast.hip; what changes is the storage:
// every activation between two kernels is a plain _Float16 (2 bytes per element instead of 4), so the HBM-bound kernels
// (GroupNorm, attention, GEMM epilogues) move half the bytes, and every product of the dense contractions is ONE f16 MFMA
// with f32 accumulation instead of three.  Softmax, GroupNorm statistics, biases, the score head and the losses stay f32.
// f16, not bf16: 11 significant bits instead of 8 for the same bytes; its narrow exponent range is covered by the
// machinery of the split-f16 mode - device-side power-of-two scales for the caller's tokens and every weight matrix, range
// guard words on everything written, exact-f32 repeat of the call when one is set (include/sola_hip.h).
// Parity: a REDUCED-precision mode with a stated tolerance (tests/test_gpu_f16.py: 3e-2 on logits of magnitude ~10, track
// decisions compared away from the threshold), reported beside the f32-class modes, never as the headline number.
#include <math.h>

#include <algorithm>

#include "ctx.h"

int launch_attention_f16(const AttnDesc& d, hipStream_t s);

int sola_forward_f16_impl(SolaCtx* c, const float* obj, const float* lang, int B, int N, int T, int L, float* score_map,
                          float* score_tokens, void* workspace, size_t ws_bytes, hipStream_t s) {
    SOLA_ARG(c && obj && lang && score_map && score_tokens && workspace, "forward: null argument");
    SOLA_ARG(B > 0 && N > 0 && T > 0 && L >= 1, "forward: bad sizes B=%d N=%d T=%d L=%d", B, N, T, L);
    SOLA_ARG(c->cfg.object_token_dim % 64 == 0 && c->cfg.lang_token_dim % 64 == 0,
             "16-bit storage mode needs object_token_dim and lang_token_dim to be multiples of 64");
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
    // buffers are the f32 plan's (sized for 4 bytes per element); this mode stores halfs in their first half
    auto buf = [&](const std::string& name) { return reinterpret_cast<float*>(base + p.bufs.at(name).off); };
    auto W = [&](const std::string& name) { return ctx_weight(c, name); };
    const int D = c->cfg.lang_token_dim, H = c->cfg.num_heads, DH = D / H, d_in = c->cfg.object_token_dim;
    const int Tp = p.Tp, M = p.M, Wn = p.W;
    const int R = B * N;
    _Float16* const ws16 = reinterpret_cast<_Float16*>(c->ws16_buf);
    _Float16* const lin16 = reinterpret_cast<_Float16*>(c->lin16_buf);
    auto linw = [&](int layer, int attn, int proj) { return reinterpret_cast<const float*>(lin16 + ((size_t)(layer * 3 + attn) * 4 + proj) * D * D); };

    if (c->ws_dirty || c->ws_every_forward || c->ws16_fmt != 2) {
        WsLayer layers[6];
        for (int i = 0; i < 6; ++i) {
            const std::string nm = "short_motion_encoder." + std::to_string(kConvIdx[i]) + ".weight";
            layers[i] = WsLayer{W(nm), c->ws_buf + c->ws_off[i], c->conv[i].cout, c->conv[i].cin, c->conv[i].k};
        }
        SOLA_TRY(launch_ws_standardize(layers, 6, s));
        for (int i = 0; i < 6; ++i) {  // standardised rows are unit-variance by construction: fixed scale 1
            const int kc = c->conv[i].k * c->conv[i].cin;
            SOLA_TRY(launch_cast_f16(c->ws_buf + c->ws_off[i], kc, ws16 + c->ws_off[i], kc, c->conv[i].cout, kc, 1.f, nullptr, s));
        }
        c->ws_dirty = false;
        c->ws16_fmt = 2;
    }
    SOLA_TRY(sola_refresh_lin16(c, s));
    SOLA_HIP(hipMemsetAsync(c->guard, 0, sizeof(int), s));

    // ---- encoder
    // The caller's tokens: largest magnitude -> [2^6, 2^7).  conv0's output STAYS in those scaled units (its standardised
    // weights have a gain of sqrt(k*cin) = 28: tokens at 1e3 would leave the f16 range if the scale were undone here, tokens
    // at 1e-5 would sink into subnormals): the bias is multiplied by the scale in conv0's epilogue and the first GroupNorm
    // multiplies by the inverse while it reads, so its statistics (and eps) see the true values.
    // The scale is capped at 2^8: conv0's bias rides along multiplied by it (tokens of 1e-5 would ask for 2^21 and carry a
    // bias of 0.03 to 75 000); below the cap the smallest token entries lose some of their 11 bits to f16 subnormals.
    SOLA_TRY(launch_cast_f16(obj, d_in, buf("obj_sp"), d_in, (long long)R * T, d_in, 256.f, c->scal_pair(0), s, 6, c->scal_extra(0)));
    const float* x = buf("obj_sp");
    int t_in = T;
    for (int i = 0; i < 6; ++i) {
        const ConvGeom& g = c->conv[i];
        const std::string cp = "short_motion_encoder." + std::to_string(kConvIdx[i]);
        GemmDesc gd{};
        gd.nprob = 1;
        gd.p[0] = GemmProblem{x, reinterpret_cast<const float*>(ws16 + c->ws_off[i]), W(cp + ".bias"), nullptr,
                              i == 5 ? buf("conv5_sp") : buf("conv" + std::to_string(i))};
        gd.M = R * p.Tl[i]; gd.N = g.cout; gd.K = g.k * g.cin;
        gd.lda = g.cin; gd.ldr = 0; gd.ldc = g.cout;
        gd.conv = g.k > 1 ? 1 : 0;
        gd.T_in = t_in; gd.T_out = p.Tl[i]; gd.stride = g.stride; gd.pad = g.pad; gd.Cin = g.cin;
        gd.arith = 2; gd.out_scale = 1.f; gd.c_f16 = 1; gd.guard = c->guard;
        if (i == 0) gd.bias_scale_dev = c->scal_extra(0);
        SOLA_TRY(launch_gemm(gd, s));
        if (i < 5) {
            const std::string np = "short_motion_encoder." + std::to_string(kNormIdx[i]);
            GroupNormDesc nd{};
            nd.slice_ws = buf("gn_slots"); nd.slice_ws_bytes = (size_t)p.bufs.at("gn_slots").rows * p.bufs.at("gn_slots").cols * sizeof(float);
            nd.x = buf("conv" + std::to_string(i)); nd.y = buf("act" + std::to_string(i));
            nd.gamma = W(np + ".weight"); nd.beta = W(np + ".bias");
            nd.n_inst = R; nd.inner = 1; nd.outer_stride = p.Tl[i]; nd.inner_stride = 0; nd.tok_stride = 1;
            nd.ntok = p.Tl[i]; nd.C = g.cout; nd.groups = c->cfg.n_groups; nd.eps = 1e-5f; nd.slope = 0.01f; nd.leaky = 1;
            nd.in_f16 = 1; nd.out_f16 = 1; nd.guard = c->guard;
            if (i == 0) nd.in_scale_dev = c->scal_pair(0) + 1;
            SOLA_TRY(launch_group_norm(nd, s));
            x = buf("act" + std::to_string(i));
        }
        t_in = p.Tl[i];
    }
    SOLA_TRY(launch_pos_encoding(W("positional_encoding_gaussian_matrix"), D, Tp, c->cfg.max_temporal_length, buf("pe"), s));
    SOLA_TRY(launch_lang_concat(lang, W("negative_token.weight"), buf("lang"), buf("lbar"), B, L, c->cfg.n_negative, D, s));
    SOLA_TRY(launch_cast_f16(buf("lang"), D, buf("lang_sp"), D, (long long)B * Wn, D, 1.f, c->scal_pair(1), s, 6));

    const float scale = 1.0f / sqrtf((float)DH);
    auto linear3 = [&](const float* a0, const float* a1, const float* a2, int layer, int attn, int nprob, int rows, float* o0,
                       float* o1, float* o2, int first_proj, const float* a_inv_scale = nullptr) -> int {
        static const char* pn[3] = {"q_proj", "k_proj", "v_proj"};
        const std::string an = "object_lang_align_layers." + std::to_string(layer) + "." + kAttnLong[attn];
        const float* as[3] = {a0, a1, a2};
        float* os[3] = {o0, o1, o2};
        GemmDesc gd{};
        gd.nprob = nprob;
        for (int j = 0; j < nprob; ++j)
            gd.p[j] = GemmProblem{as[j], linw(layer, attn, first_proj + j), W(an + "." + pn[first_proj + j] + ".bias"), nullptr, os[j],
                                  c->lin_inv_scale(layer, attn, first_proj + j)};
        gd.M = rows; gd.N = D; gd.K = D; gd.lda = D; gd.ldr = 0; gd.ldc = D;
        gd.arith = 2; gd.out_scale = 1.f; gd.out_scale_dev = a_inv_scale; gd.c_f16 = 1; gd.guard = c->guard;
        return launch_gemm(gd, s);
    };
    auto out_proj = [&](int layer, int attn, const float* resid) -> int {
        const std::string an = "object_lang_align_layers." + std::to_string(layer) + "." + kAttnLong[attn];
        GemmDesc gd{};
        gd.nprob = 1;
        gd.p[0] = GemmProblem{buf("attn"), linw(layer, attn, 3), W(an + ".out_proj.bias"), resid, buf("res"), c->lin_inv_scale(layer, attn, 3)};
        gd.M = M; gd.N = D; gd.K = D; gd.lda = D; gd.ldr = D; gd.ldc = D;
        gd.arith = 2; gd.out_scale = 1.f; gd.r_f16 = 1; gd.c_f16 = 1; gd.guard = c->guard;
        return launch_gemm(gd, s);
    };
    auto gn = [&](const std::string& lp, int idx, float* y, float* y2, int f16_out, int n_inst, int inner, long long outer,
                  long long inner_stride, long long tok_stride, int ntok) -> int {
        GroupNormDesc nd{};
            nd.slice_ws = buf("gn_slots"); nd.slice_ws_bytes = (size_t)p.bufs.at("gn_slots").rows * p.bufs.at("gn_slots").cols * sizeof(float);
        nd.x = buf("res"); nd.y = y; nd.y2 = y2; nd.pe = y2 ? buf("pe") : nullptr;
        nd.gamma = W(lp + "norm." + std::to_string(idx) + ".weight");
        nd.beta = W(lp + "norm." + std::to_string(idx) + ".bias");
        nd.n_inst = n_inst; nd.inner = inner; nd.outer_stride = outer; nd.inner_stride = inner_stride;
        nd.tok_stride = tok_stride; nd.ntok = ntok; nd.C = D; nd.groups = c->cfg.n_groups_module;
        nd.eps = 1e-5f; nd.slope = 0.f; nd.leaky = 0; nd.in_f16 = 1; nd.out_f16 = f16_out; nd.guard = c->guard;
        return launch_group_norm(nd, s);
    };
    auto attention = [&](const float* q, const float* k, const float* v, int G, int Sq, int Sk, int inner, long long qo,
                         long long qi, long long qr, long long ko, long long ki, long long kr) -> int {
        AttnDesc ad{q, k, v, buf("attn"), D, D, D, D, G, H, DH, Sq, Sk, inner, qo, qi, qr, ko, ki, kr, scale, nullptr};
        ad.guard = c->guard;
        return launch_attention_f16(ad, s);
    };

    const float* xin = buf("conv5_sp");
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
        SOLA_TRY(linear3(xin, xin, xin, l, 0, 3, M, q, k, v, 0));
        SOLA_TRY(attention(q, k, v, B * Tp, N, N, Tp, (long long)N * Tp, 1, Tp, (long long)N * Tp, 1, Tp));
        SOLA_TRY(out_proj(l, 0, xin));
        SOLA_TRY(gn(lp, 0, x_obj, x_pe, 1, B * Tp, Tp, (long long)N * Tp, 1, Tp, N));
        // (ii) motion attention (module.py:38-43)
        SOLA_TRY(linear3(x_pe, x_pe, x_obj, l, 1, 3, M, q, k, v, 0));
        SOLA_TRY(attention(q, k, v, B * N, Tp, Tp, 1, (long long)Tp, 0, 1, (long long)Tp, 0, 1));
        SOLA_TRY(out_proj(l, 1, x_obj));
        SOLA_TRY(gn(lp, 1, x_mot, nullptr, 1, B * N, 1, Tp, 0, 1, Tp));
        // (iii) object -> language attention (module.py:46-50)
        SOLA_TRY(linear3(x_mot, nullptr, nullptr, l, 2, 1, M, q, nullptr, nullptr, 0));
        SOLA_TRY(linear3(buf("lang_sp"), buf("lang_sp"), nullptr, l, 2, 2, B * Wn, buf("lk"), buf("lv"), nullptr, 1, c->scal_pair(1) + 1));
        SOLA_TRY(attention(q, buf("lk"), buf("lv"), B, N * Tp, Wn, 1, (long long)N * Tp, 0, 1, (long long)Wn, 0, 1));
        SOLA_TRY(out_proj(l, 2, x_mot));
        SOLA_TRY(gn(lp, 2, x_o2l, nullptr, last ? 0 : 1, B, 1, (long long)N * Tp, 0, 1, N * Tp));  // the score head reads f32
        xin = x_o2l;
    }
    HeadDesc hd{xin, buf("lbar"), score_map, score_tokens, B, N, Tp, D};
    SOLA_TRY(launch_score_head(hd, s));
    c->last = p;
    c->last_obj = nullptr;
    return SOLA_OK;
}
