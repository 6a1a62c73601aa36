// C ABI of libsola_hip.so (see include/sola_hip.h): context, weight registry, forward orchestration, event profiler.
// Host orchestration only - every numeric step is a HIP kernel in gemm.hip / attn.hip / norm.hip / head.hip / iou.hip.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"

// ---------------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void sola_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* sola_last_error(void) { return g_err; }
extern "C" const char* sola_version(void) { return "sola_hip 0.1 (gfx950, f32 MFMA)"; }

// ---------------------------------------------------------------------------------------------------------------
// event profiler: start/stop HIP events on the launch stream around every kernel launch while enabled
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct ProfRec {
    int cat;
    hipEvent_t e0, e1;
    double flops, bytes;
};
struct Profiler {
    std::mutex mu;
    bool enabled = false;
    std::vector<ProfRec> pending;
    std::vector<hipEvent_t> pool;
    int64_t launches[SOLA_PROF_NCAT] = {0};
    double ms[SOLA_PROF_NCAT] = {0}, flops[SOLA_PROF_NCAT] = {0}, bytes[SOLA_PROF_NCAT] = {0};
    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    void drain() {
        for (auto& r : pending) {
            float t = 0.f;
            if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
                launches[r.cat] += 1;
                ms[r.cat] += t;
                flops[r.cat] += r.flops;
                bytes[r.cat] += r.bytes;
            }
            pool.push_back(r.e0);
            pool.push_back(r.e1);
        }
        pending.clear();
    }
};
Profiler g_prof;
}  // namespace

SolaProfScope::SolaProfScope(int cat_, hipStream_t stream_, double flops, double bytes)
    : cat(cat_), stream(stream_), on(false), slot(-1) {
    if (!g_prof.enabled) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    ProfRec r{cat, g_prof.get(), g_prof.get(), flops, bytes};
    if (!r.e0 || !r.e1) return;
    if (hipEventRecord(r.e0, stream) != hipSuccess) return;
    g_prof.pending.push_back(r);
    slot = (int)g_prof.pending.size() - 1;
    on = true;
}
SolaProfScope::~SolaProfScope() {
    if (!on) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (slot < (int)g_prof.pending.size()) (void)hipEventRecord(g_prof.pending[slot].e1, stream);
    if (g_prof.pending.size() > 200000) g_prof.drain();
}
extern "C" int sola_profile_enable(int enable) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.enabled = enable != 0;
    return SOLA_OK;
}
extern "C" int sola_profile_read(int64_t* launches, double* ms, double* flops, double* bytes, int reset) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.drain();
    for (int i = 0; i < SOLA_PROF_NCAT; ++i) {
        if (launches) launches[i] = g_prof.launches[i];
        if (ms) ms[i] = g_prof.ms[i];
        if (flops) flops[i] = g_prof.flops[i];
        if (bytes) bytes[i] = g_prof.bytes[i];
        if (reset) {
            g_prof.launches[i] = 0;
            g_prof.ms[i] = g_prof.flops[i] = g_prof.bytes[i] = 0;
        }
    }
    return SOLA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct ConvGeom {
    int cin, cout, k, stride, pad;
};
struct Weight {
    std::string name;
    int64_t numel;
    const float* ptr;
};
struct Buf {
    size_t off;
    int64_t rows, cols;
};
struct Plan {
    int B = 0, N = 0, T = 0, L = 0, W = 0, Tp = 0, M = 0;
    int Tl[6] = {0};
    std::unordered_map<std::string, Buf> bufs;
    size_t total = 0;
    size_t add(const std::string& name, int64_t rows, int64_t cols) {
        const size_t off = total;
        bufs[name] = Buf{off, rows, cols};
        total += (((size_t)rows * (size_t)cols * sizeof(float)) + 255) & ~(size_t)255;
        return off;
    }
};
}  // namespace

struct SolaCtx {
    SolaConfig cfg;
    int device;
    ConvGeom conv[6];
    std::vector<Weight> weights;
    std::unordered_map<std::string, int> index;
    float* ws_buf = nullptr;  // standardised conv weights [cout][k*cin], all six layers, ctx-owned
    size_t ws_off[6];
    bool ws_dirty = true;
    bool ws_every_forward = true;
    Plan last;
};

namespace {
const int kConvIdx[6] = {0, 4, 8, 12, 16, 20};
const int kNormIdx[5] = {1, 5, 9, 13, 17};

void add_weight(SolaCtx* c, const std::string& name, int64_t numel) {
    c->index[name] = (int)c->weights.size();
    c->weights.push_back(Weight{name, numel, nullptr});
}

const float* W(const SolaCtx* c, const std::string& name) {
    auto it = c->index.find(name);
    return it == c->index.end() ? nullptr : c->weights[it->second].ptr;
}

Plan make_plan(const SolaCtx* c, int B, int N, int T, int L) {
    Plan p;
    p.B = B; p.N = N; p.T = T; p.L = L;
    p.W = L + c->cfg.n_negative;
    int t = T;
    for (int i = 0; i < 6; ++i) {
        t = (t + 2 * c->conv[i].pad - c->conv[i].k) / c->conv[i].stride + 1;
        p.Tl[i] = t;
    }
    p.Tp = p.Tl[5];
    p.M = B * N * p.Tp;
    const int D = c->cfg.lang_token_dim;
    const int64_t R = (int64_t)B * N;
    for (int i = 0; i < 6; ++i) {
        p.add("conv" + std::to_string(i), R * p.Tl[i], c->conv[i].cout);
        if (i < 5) p.add("act" + std::to_string(i), R * p.Tl[i], c->conv[i].cout);
    }
    p.add("pe", p.Tp, D);
    p.add("lang", (int64_t)B * p.W, D);
    p.add("lbar", B, D);
    p.add("q", p.M, D);
    p.add("k", p.M, D);
    p.add("v", p.M, D);
    p.add("lk", (int64_t)B * p.W, D);
    p.add("lv", (int64_t)B * p.W, D);
    p.add("attn", p.M, D);
    p.add("res", p.M, D);
    p.add("xpe", p.M, D);
    for (int l = 0; l < c->cfg.n_layers; ++l) {
        p.add("l" + std::to_string(l) + "_obj", p.M, D);
        p.add("l" + std::to_string(l) + "_motion", p.M, D);
        p.add("l" + std::to_string(l) + "_o2l", p.M, D);
    }
    p.add("loss_terms", (int64_t)B * N, 4);
    return p;
}
}  // namespace

extern "C" int sola_ctx_create(const SolaConfig* cfg, int device, SolaCtx** out) {
    SOLA_ARG(cfg && out, "ctx_create: null argument");
    SOLA_ARG(cfg->num_heads > 0 && cfg->lang_token_dim % cfg->num_heads == 0, "ctx_create: lang_token_dim %% num_heads != 0");
    const int dh = cfg->lang_token_dim / cfg->num_heads;
    SOLA_ARG(dh == 16 || dh == 32 || dh == 64 || dh == 128, "ctx_create: head_dim %d unsupported (16/32/64/128)", dh);
    SOLA_ARG(cfg->object_token_dim % 4 == 0 && cfg->object_token_dim > 0, "ctx_create: object_token_dim must be a multiple of 4");
    SOLA_ARG(cfg->n_layers >= 1 && cfg->n_negative >= 1 && cfg->n_groups >= 1 && cfg->n_groups_module >= 1, "ctx_create: bad config");
    SOLA_ARG((2 * cfg->object_token_dim) % (4 * cfg->n_groups) == 0 && cfg->lang_token_dim % (4 * cfg->n_groups) == 0 &&
                 cfg->lang_token_dim % (4 * cfg->n_groups_module) == 0,
             "ctx_create: channels per GroupNorm group must be a multiple of 4");
    SOLA_HIP(hipSetDevice(device));
    SolaCtx* c = new SolaCtx();
    c->cfg = *cfg;
    c->device = device;
    const int d = cfg->object_token_dim, h = 2 * d, D = cfg->lang_token_dim;
    const ConvGeom geo[6] = {{d, h, 3, 2, 1}, {h, h, 3, 2, 1}, {h, h, 3, 2, 1}, {h, D, 3, 1, 1}, {D, D, 3, 1, 1}, {D, D, 1, 1, 0}};
    // state_dict order of the reference: own buffer, encoder, layers, embedding
    add_weight(c, "positional_encoding_gaussian_matrix", D / 2);
    size_t ws_total = 0;
    for (int i = 0; i < 6; ++i) {
        c->conv[i] = geo[i];
        const std::string p = "short_motion_encoder." + std::to_string(kConvIdx[i]);
        add_weight(c, p + ".weight", (int64_t)geo[i].cout * geo[i].cin * geo[i].k);
        add_weight(c, p + ".bias", geo[i].cout);
        if (i < 5) {
            const std::string n = "short_motion_encoder." + std::to_string(kNormIdx[i]);
            add_weight(c, n + ".weight", geo[i].cout);
            add_weight(c, n + ".bias", geo[i].cout);
        }
        c->ws_off[i] = ws_total;
        ws_total += (size_t)geo[i].cout * geo[i].cin * geo[i].k;
    }
    static const char* attn_names[3] = {"obj_attn", "motion_attn", "object2lang_attn"};
    static const char* proj_names[4] = {"q_proj", "k_proj", "v_proj", "out_proj"};
    for (int l = 0; l < cfg->n_layers; ++l) {
        const std::string p = "object_lang_align_layers." + std::to_string(l) + ".";
        for (int a = 0; a < 3; ++a)
            for (int j = 0; j < 4; ++j) {
                add_weight(c, p + attn_names[a] + "." + proj_names[j] + ".weight", (int64_t)D * D);
                add_weight(c, p + attn_names[a] + "." + proj_names[j] + ".bias", D);
            }
        for (int j = 0; j < 3; ++j) {
            add_weight(c, p + "norm." + std::to_string(j) + ".weight", D);
            add_weight(c, p + "norm." + std::to_string(j) + ".bias", D);
        }
    }
    add_weight(c, "negative_token.weight", (int64_t)cfg->n_negative * D);
    hipError_t e = hipMalloc(&c->ws_buf, ws_total * sizeof(float));
    if (e != hipSuccess) {
        sola_set_error("ctx_create: hipMalloc(%zu) failed: %s", ws_total * sizeof(float), hipGetErrorString(e));
        delete c;
        return SOLA_ERR_HIP;
    }
    *out = c;
    return SOLA_OK;
}

extern "C" int sola_ctx_destroy(SolaCtx* c) {
    if (!c) return SOLA_OK;
    if (c->ws_buf) (void)hipFree(c->ws_buf);
    delete c;
    return SOLA_OK;
}

extern "C" int sola_num_weights(const SolaCtx* c) { return c ? (int)c->weights.size() : 0; }

extern "C" int sola_weight_info(const SolaCtx* c, int index, const char** name, int64_t* numel) {
    SOLA_ARG(c && index >= 0 && index < (int)c->weights.size(), "weight_info: index %d out of range", index);
    if (name) *name = c->weights[index].name.c_str();
    if (numel) *numel = c->weights[index].numel;
    return SOLA_OK;
}

extern "C" int sola_set_weight(SolaCtx* c, const char* name, const void* dev_ptr, int64_t numel) {
    SOLA_ARG(c && name && dev_ptr, "set_weight: null argument");
    auto it = c->index.find(name);
    if (it == c->index.end()) {
        sola_set_error("set_weight: unknown state_dict key '%s'", name);
        return SOLA_ERR_WEIGHT;
    }
    Weight& w = c->weights[it->second];
    if (w.numel != numel) {
        sola_set_error("set_weight: '%s' has %lld elements, expected %lld", name, (long long)numel, (long long)w.numel);
        return SOLA_ERR_WEIGHT;
    }
    if ((reinterpret_cast<uintptr_t>(dev_ptr) & 15) != 0) {
        sola_set_error("set_weight: '%s' must be 16-byte aligned", name);
        return SOLA_ERR_WEIGHT;
    }
    w.ptr = static_cast<const float*>(dev_ptr);
    c->ws_dirty = true;
    return SOLA_OK;
}

extern "C" int sola_weights_changed(SolaCtx* c) {
    SOLA_ARG(c, "weights_changed: null ctx");
    c->ws_dirty = true;
    return SOLA_OK;
}

extern "C" int sola_set_ws_policy(SolaCtx* c, int every) {
    SOLA_ARG(c, "set_ws_policy: null ctx");
    c->ws_every_forward = every != 0;
    return SOLA_OK;
}

extern "C" size_t sola_workspace_bytes(const SolaCtx* c, int B, int N, int T, int L) {
    if (!c || B <= 0 || N <= 0 || T <= 0 || L < 0) return 0;
    return make_plan(c, B, N, T, L).total;
}

extern "C" int sola_workspace_tap(const SolaCtx* c, const char* name, size_t* off, int64_t* rows, int64_t* cols) {
    SOLA_ARG(c && name, "workspace_tap: null argument");
    auto it = c->last.bufs.find(name);
    SOLA_ARG(it != c->last.bufs.end(), "workspace_tap: no buffer named '%s' (run sola_forward first)", name);
    if (off) *off = it->second.off;
    if (rows) *rows = it->second.rows;
    if (cols) *cols = it->second.cols;
    return SOLA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
extern "C" int sola_forward(SolaCtx* c, const float* obj, const float* lang, int B, int N, int T, int L,
                            float* score_map, float* score_tokens, void* workspace, size_t ws_bytes, void* stream_) {
    SOLA_ARG(c && obj && lang && score_map && score_tokens && workspace, "forward: null argument");
    SOLA_ARG(B > 0 && N > 0 && T > 0 && L >= 1, "forward: bad sizes B=%d N=%d T=%d L=%d", B, N, T, L);
    for (const Weight& w : c->weights)
        if (!w.ptr) {
            sola_set_error("forward: weight '%s' has not been set", w.name.c_str());
            return SOLA_ERR_WEIGHT;
        }
    hipStream_t s = as_stream(stream_);
    Plan p = make_plan(c, B, N, T, L);
    if (ws_bytes < p.total) {
        sola_set_error("forward: workspace %zu bytes < required %zu", ws_bytes, p.total);
        return SOLA_ERR_WORKSPACE;
    }
    SOLA_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "forward: workspace must be 256-byte aligned");
    char* base = static_cast<char*>(workspace);
    auto buf = [&](const std::string& name) { return reinterpret_cast<float*>(base + p.bufs.at(name).off); };
    const int D = c->cfg.lang_token_dim, H = c->cfg.num_heads, DH = D / H;
    const int Tp = p.Tp, M = p.M, Wn = p.W;
    const int R = B * N;

    // a1: weight standardisation (module/ws.py:9-13), every forward like the reference unless the policy says cached
    if (c->ws_dirty || c->ws_every_forward) {
        WsLayer layers[6];
        for (int i = 0; i < 6; ++i) {
            const std::string nm = "short_motion_encoder." + std::to_string(kConvIdx[i]) + ".weight";
            layers[i] = WsLayer{W(c, nm), c->ws_buf + c->ws_off[i], c->conv[i].cout, c->conv[i].cin, c->conv[i].k};
        }
        SOLA_TRY(launch_ws_standardize(layers, 6, s));
        c->ws_dirty = false;
    }

    // a2: encoder (module/module.py:74-96,137-140)
    const float* x = obj;
    int t_in = T;
    for (int i = 0; i < 6; ++i) {
        const ConvGeom& g = c->conv[i];
        const std::string cp = "short_motion_encoder." + std::to_string(kConvIdx[i]);
        GemmDesc gd{};
        gd.nprob = 1;
        gd.p[0] = GemmProblem{x, c->ws_buf + c->ws_off[i], W(c, cp + ".bias"), nullptr, buf("conv" + std::to_string(i))};
        gd.M = R * p.Tl[i]; gd.N = g.cout; gd.K = g.k * g.cin;
        gd.lda = g.cin; gd.ldr = 0; gd.ldc = g.cout;
        gd.conv = g.k > 1 ? 1 : 0;
        gd.T_in = t_in; gd.T_out = p.Tl[i]; gd.stride = g.stride; gd.pad = g.pad; gd.Cin = g.cin;
        SOLA_TRY(launch_gemm(gd, s));
        if (i < 5) {
            const std::string np = "short_motion_encoder." + std::to_string(kNormIdx[i]);
            GroupNormDesc nd{};
            nd.x = buf("conv" + std::to_string(i)); nd.y = buf("act" + std::to_string(i)); nd.y2 = nullptr; nd.pe = nullptr;
            nd.gamma = W(c, np + ".weight"); nd.beta = W(c, np + ".bias");
            nd.n_inst = R; nd.inner = 1; nd.outer_stride = p.Tl[i]; nd.inner_stride = 0; nd.tok_stride = 1;
            nd.ntok = p.Tl[i]; nd.C = g.cout; nd.groups = c->cfg.n_groups; nd.eps = 1e-5f; nd.slope = 0.01f; nd.leaky = 1;
            SOLA_TRY(launch_group_norm(nd, s));
            x = buf("act" + std::to_string(i));
        } else {
            x = buf("conv5");
        }
        t_in = p.Tl[i];
    }

    // a3: positional table; text tokens ++ negative tokens and their mean (module/module.py:143-147)
    SOLA_TRY(launch_pos_encoding(W(c, "positional_encoding_gaussian_matrix"), D, Tp, c->cfg.max_temporal_length, buf("pe"), s));
    SOLA_TRY(launch_lang_concat(lang, W(c, "negative_token.weight"), buf("lang"), buf("lbar"), B, L, c->cfg.n_negative, D, s));

    // a5: alignment layers (module/module.py:22-52)
    const float scale = 1.0f / sqrtf((float)DH);
    auto linear3 = [&](const float* a0, const float* a1, const float* a2, const std::string& attn, int nprob, int rows,
                       float* o0, float* o1, float* o2, int first_proj) -> int {
        static const char* pn[3] = {"q_proj", "k_proj", "v_proj"};
        const float* as[3] = {a0, a1, a2};
        float* os[3] = {o0, o1, o2};
        GemmDesc gd{};
        gd.nprob = nprob;
        for (int j = 0; j < nprob; ++j)
            gd.p[j] = GemmProblem{as[j], W(c, attn + "." + pn[first_proj + j] + ".weight"),
                                  W(c, attn + "." + pn[first_proj + j] + ".bias"), nullptr, os[j]};
        gd.M = rows; gd.N = D; gd.K = D; gd.lda = D; gd.ldr = 0; gd.ldc = D;
        return launch_gemm(gd, s);
    };
    auto out_proj = [&](const std::string& attn, const float* resid) -> int {
        GemmDesc gd{};
        gd.nprob = 1;
        gd.p[0] = GemmProblem{buf("attn"), W(c, attn + ".out_proj.weight"), W(c, attn + ".out_proj.bias"), resid, buf("res")};
        gd.M = M; gd.N = D; gd.K = D; gd.lda = D; gd.ldr = D; gd.ldc = D;
        return launch_gemm(gd, s);
    };
    auto gn = [&](const std::string& lp, int idx, float* y, float* y2, int n_inst, int inner, long long outer,
                  long long inner_stride, long long tok_stride, int ntok) -> int {
        GroupNormDesc nd{};
        nd.x = buf("res"); nd.y = y; nd.y2 = y2; nd.pe = y2 ? buf("pe") : nullptr;
        nd.gamma = W(c, lp + "norm." + std::to_string(idx) + ".weight");
        nd.beta = W(c, lp + "norm." + std::to_string(idx) + ".bias");
        nd.n_inst = n_inst; nd.inner = inner; nd.outer_stride = outer; nd.inner_stride = inner_stride;
        nd.tok_stride = tok_stride; nd.ntok = ntok; nd.C = D; nd.groups = c->cfg.n_groups_module;
        nd.eps = 1e-5f; nd.slope = 0.f; nd.leaky = 0;
        return launch_group_norm(nd, s);
    };
    const float* xin = buf("conv5");
    for (int l = 0; l < c->cfg.n_layers; ++l) {
        const std::string lp = "object_lang_align_layers." + std::to_string(l) + ".";
        const std::string ls = "l" + std::to_string(l);
        float *q = buf("q"), *k = buf("k"), *v = buf("v");
        // (i) inter-object attention over the N tracks of each (b, t'): module.py:31-35
        SOLA_TRY(linear3(xin, xin, xin, lp + "obj_attn", 3, M, q, k, v, 0));
        {
            AttnDesc ad{q, k, v, buf("attn"), D, D, D, D, B * Tp, H, DH, N, N, Tp,
                        (long long)N * Tp, 1, Tp, (long long)N * Tp, 1, Tp, scale};
            SOLA_TRY(launch_attention(ad, s));
        }
        SOLA_TRY(out_proj(lp + "obj_attn", xin));
        SOLA_TRY(gn(lp, 0, buf(ls + "_obj"), buf("xpe"), B * Tp, Tp, (long long)N * Tp, 1, Tp, N));
        // (ii) motion attention over T' per track, PE on q and k only: module.py:38-43
        SOLA_TRY(linear3(buf("xpe"), buf("xpe"), buf(ls + "_obj"), lp + "motion_attn", 3, M, q, k, v, 0));
        {
            AttnDesc ad{q, k, v, buf("attn"), D, D, D, D, B * N, H, DH, Tp, Tp, 1,
                        (long long)Tp, 0, 1, (long long)Tp, 0, 1, scale};
            SOLA_TRY(launch_attention(ad, s));
        }
        SOLA_TRY(out_proj(lp + "motion_attn", buf(ls + "_obj")));
        SOLA_TRY(gn(lp, 1, buf(ls + "_motion"), nullptr, B * N, 1, Tp, 0, 1, Tp));
        // (iii) object -> language cross attention: module.py:46-50
        SOLA_TRY(linear3(buf(ls + "_motion"), nullptr, nullptr, lp + "object2lang_attn", 1, M, q, nullptr, nullptr, 0));
        SOLA_TRY(linear3(buf("lang"), buf("lang"), nullptr, lp + "object2lang_attn", 2, B * Wn, buf("lk"), buf("lv"), nullptr, 1));
        {
            AttnDesc ad{q, buf("lk"), buf("lv"), buf("attn"), D, D, D, D, B, H, DH, N * Tp, Wn, 1,
                        (long long)N * Tp, 0, 1, (long long)Wn, 0, 1, scale};
            SOLA_TRY(launch_attention(ad, s));
        }
        SOLA_TRY(out_proj(lp + "object2lang_attn", buf(ls + "_motion")));
        SOLA_TRY(gn(lp, 2, buf(ls + "_o2l"), nullptr, B, 1, (long long)N * Tp, 0, 1, N * Tp));
        xin = buf(ls + "_o2l");
    }

    // a6: score head (module/module.py:152-160)
    HeadDesc hd{xin, buf("lbar"), score_map, score_tokens, B, N, Tp, D};
    SOLA_TRY(launch_score_head(hd, s));
    c->last = p;
    return SOLA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// losses / selection
// ---------------------------------------------------------------------------------------------------------------
extern "C" int sola_loss(const float* score_map, const float* score_tokens, const float* labels, const float* pos,
                         const float* neg, int64_t neg_batch_stride, int B, int N, int D, int n_neg, float pw,
                         float temperature, float aw, float* loss3, int32_t* neg_argmax, void* scratch,
                         size_t scratch_bytes, void* stream_) {
    SOLA_ARG(score_map && score_tokens && labels && pos && neg && loss3 && scratch, "loss: null argument");
    SOLA_ARG(B > 0 && N > 0 && D > 0 && n_neg > 0, "loss: bad sizes");
    const size_t need = (size_t)B * N * 3 * sizeof(float);
    if (scratch_bytes < need) {
        sola_set_error("loss: scratch %zu bytes < required %zu", scratch_bytes, need);
        return SOLA_ERR_WORKSPACE;
    }
    LossDesc d{};
    d.score_map = score_map; d.score_tokens = score_tokens; d.labels = labels; d.pos = pos;
    d.neg = neg; d.neg_batch_stride = neg_batch_stride;
    d.B = B; d.N = N; d.D = D; d.n_neg = n_neg;
    d.pos_w = pw; d.temp_scale = expf(temperature); d.align_w = aw;  // exp(temperature): tools/loss.py:32
    d.terms = static_cast<float*>(scratch); d.loss3 = loss3; d.neg_argmax = neg_argmax;
    return launch_loss(d, as_stream(stream_));
}

extern "C" int sola_select(const float* score, int64_t n, float thr, float* prob, float* pred, void* stream_) {
    SOLA_ARG(score && (prob || pred), "select: null argument");
    return launch_select(score, n, thr, prob, pred, as_stream(stream_));
}

// ---------------------------------------------------------------------------------------------------------------
// per-stage entry points
// ---------------------------------------------------------------------------------------------------------------
extern "C" int sola_ws_standardize(const float* w, int cout, int cin, int k, float* out, void* stream_) {
    SOLA_ARG(w && out && cout > 0, "ws_standardize: bad argument");
    WsLayer l{w, out, cout, cin, k};
    return launch_ws_standardize(&l, 1, as_stream(stream_));
}

extern "C" int sola_gemm_nt(const float* a, int lda, const float* w, const float* bias, const float* r, int ldr,
                            float* cmat, int ldc, int M, int N, int K, void* stream_) {
    SOLA_ARG(a && w && cmat, "gemm_nt: null argument");
    GemmDesc gd{};
    gd.nprob = 1;
    gd.p[0] = GemmProblem{a, w, bias, r, cmat};
    gd.M = M; gd.N = N; gd.K = K; gd.lda = lda; gd.ldr = ldr; gd.ldc = ldc;
    return launch_gemm(gd, as_stream(stream_));
}

extern "C" int sola_conv1d_cl(const float* x, const float* wstd, const float* bias, float* y, int R, int T_in, int cin,
                              int cout, int k, int stride, int pad, void* stream_) {
    SOLA_ARG(x && wstd && y && R > 0 && T_in > 0 && k >= 1 && stride >= 1, "conv1d_cl: bad argument");
    const int T_out = (T_in + 2 * pad - k) / stride + 1;
    SOLA_ARG(T_out > 0, "conv1d_cl: empty output");
    GemmDesc gd{};
    gd.nprob = 1;
    gd.p[0] = GemmProblem{x, wstd, bias, nullptr, y};
    gd.M = R * T_out; gd.N = cout; gd.K = k * cin; gd.lda = cin; gd.ldc = cout;
    gd.conv = (k > 1 || stride > 1 || pad > 0) ? 1 : 0;
    gd.T_in = T_in; gd.T_out = T_out; gd.stride = stride; gd.pad = pad; gd.Cin = cin;
    return launch_gemm(gd, as_stream(stream_));
}

extern "C" int sola_group_norm(const float* x, float* y, float* y2, const float* pe, const float* gamma,
                               const float* beta, int n_inst, int inner, int64_t outer_stride, int64_t inner_stride,
                               int64_t tok_stride, int ntok, int C, int groups, float eps, float slope, int leaky,
                               void* stream_) {
    SOLA_ARG(x && y && gamma && beta, "group_norm: null argument");
    GroupNormDesc d{x, y, y2, pe, gamma, beta, n_inst, inner, outer_stride, inner_stride, tok_stride, ntok, C, groups, eps, slope, leaky};
    return launch_group_norm(d, as_stream(stream_));
}

extern "C" int sola_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                              int G, int H, int head_dim, int Sq, int Sk, int inner, int64_t q_outer, int64_t q_inner,
                              int64_t q_rs, int64_t k_outer, int64_t k_inner, int64_t k_rs, float scale, void* stream_) {
    SOLA_ARG(q && k && v && o, "attention: null argument");
    AttnDesc d{q, k, v, o, ldq, ldk, ldv, ldo, G, H, head_dim, Sq, Sk, inner, q_outer, q_inner, q_rs, k_outer, k_inner, k_rs, scale};
    return launch_attention(d, as_stream(stream_));
}

extern "C" int sola_pos_encoding(const float* gauss, int D, int t_len, int max_len, float* pe, void* stream_) {
    SOLA_ARG(gauss && pe, "pos_encoding: null argument");
    return launch_pos_encoding(gauss, D, t_len, max_len, pe, as_stream(stream_));
}

// ---------------------------------------------------------------------------------------------------------------
// mask IoU
// ---------------------------------------------------------------------------------------------------------------
extern "C" int64_t sola_mask_words(int H, int W) { return ((int64_t)H * W + 31) / 32; }

extern "C" int sola_mask_pack(const void* masks, int elem_type, int n, int h, int w, int H, int W, uint32_t* bits,
                              int64_t* area, void* stream_) {
    SOLA_ARG(masks && bits && area, "mask_pack: null argument");
    return launch_mask_pack(masks, elem_type, n, h, w, H, W, bits, reinterpret_cast<long long*>(area), as_stream(stream_));
}

extern "C" int sola_mask_pair_counts(const uint32_t* a_bits, const int64_t* a_area, int P, int T, const uint32_t* b_bits,
                                     const int64_t* b_area, int R, const int32_t* a_frame, int64_t words,
                                     int64_t* inter, int64_t* uni, void* stream_) {
    SOLA_ARG(a_bits && a_area && b_bits && b_area && inter && uni, "mask_pair_counts: null argument");
    return launch_mask_pair(a_bits, reinterpret_cast<const long long*>(a_area), P, T, b_bits,
                            reinterpret_cast<const long long*>(b_area), R, a_frame, words,
                            reinterpret_cast<long long*>(inter), reinterpret_cast<long long*>(uni), as_stream(stream_));
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t sola_mask_iou_scratch_bytes(int P, int R, int H, int W) {
    const size_t words = (size_t)sola_mask_words(H, W);
    return align256((size_t)P * words * 4) + align256((size_t)R * words * 4) + align256((size_t)P * 8) + align256((size_t)R * 8);
}

extern "C" int sola_mask_iou_matrix(const void* a, const void* b, int elem_type, int P, int R, int H, int W, int h, int w,
                                    int64_t* inter, int64_t* uni, void* scratch, size_t scratch_bytes, void* stream_) {
    SOLA_ARG(a && b && inter && uni && scratch, "mask_iou_matrix: null argument");
    SOLA_ARG(P > 0 && R > 0, "mask_iou_matrix: bad sizes");
    if (scratch_bytes < sola_mask_iou_scratch_bytes(P, R, H, W)) {
        sola_set_error("mask_iou_matrix: scratch %zu bytes < required %zu", scratch_bytes, sola_mask_iou_scratch_bytes(P, R, H, W));
        return SOLA_ERR_WORKSPACE;
    }
    SOLA_ARG((reinterpret_cast<uintptr_t>(scratch) & 255) == 0, "mask_iou_matrix: scratch must be 256-byte aligned");
    const size_t words = (size_t)sola_mask_words(H, W);
    char* base = static_cast<char*>(scratch);
    uint32_t* abits = reinterpret_cast<uint32_t*>(base);
    uint32_t* bbits = reinterpret_cast<uint32_t*>(base + align256((size_t)P * words * 4));
    long long* aarea = reinterpret_cast<long long*>(base + align256((size_t)P * words * 4) + align256((size_t)R * words * 4));
    long long* barea = reinterpret_cast<long long*>(reinterpret_cast<char*>(aarea) + align256((size_t)P * 8));
    hipStream_t s = as_stream(stream_);
    SOLA_TRY(launch_mask_pack(a, elem_type, P, H, W, H, W, abits, aarea, s));
    SOLA_TRY(launch_mask_pack(b, elem_type, R, h, w, H, W, bbits, barea, s));
    return launch_mask_pair(abits, aarea, P, 1, bbits, barea, R, nullptr, (long long)words,
                            reinterpret_cast<long long*>(inter), reinterpret_cast<long long*>(uni), s);
}
