// C ABI of libsola_hip.so (see include/sola_hip.h): context + weight registry, thin wrappers over the orchestration in
// forward.hip / backward.hip and over the kernel launchers, and the HIP-event profiler.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <mutex>

#include "ragged.h"

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
extern "C" const char* sola_version(void) { return "sola_hip 0.6 (gfx950; f32, range-guarded split-f16 and f16-operand MFMA forward and backward, ragged batches, mask IoU + masklet rows)"; }

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
    unsigned mask = ~0u;  // categories that are timed while enabled (sola_profile_enable)
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
    if (!g_prof.enabled || !((g_prof.mask >> cat) & 1u)) return;
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
    g_prof.mask = (enable == 0 || enable == 1) ? ~0u : ((unsigned)enable >> 1);  // > 1: bit (c + 1) set = category c is timed
    // events are only recycled when the counters are read, so a timed region of K unsynchronised steps needs two per kernel
    // launch per step: create them here, outside anybody's timed region (hipEventCreate costs 3-20 us a piece)
    while (enable && g_prof.pool.size() < 8192) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) break;
        g_prof.pool.push_back(e);
    }
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
static void add_weight(SolaCtx* c, const std::string& name, int64_t numel) {
    c->index[name] = (int)c->weights.size();
    c->weights.push_back(Weight{name, numel, nullptr, nullptr});
}

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
    static const char* proj_names[4] = {"q_proj", "k_proj", "v_proj", "out_proj"};
    for (int l = 0; l < cfg->n_layers; ++l) {
        const std::string p = "object_lang_align_layers." + std::to_string(l) + ".";
        for (int a = 0; a < 3; ++a)
            for (int j = 0; j < 4; ++j) {
                add_weight(c, p + kAttnLong[a] + "." + proj_names[j] + ".weight", (int64_t)D * D);
                add_weight(c, p + kAttnLong[a] + "." + proj_names[j] + ".bias", D);
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
    if (c->ws16_buf) (void)hipFree(c->ws16_buf);
    if (c->lin16_buf) (void)hipFree(c->lin16_buf);
    if (c->scal_buf) (void)hipFree(c->scal_buf);
    if (c->guard_host) (void)hipHostFree(c->guard_host);
    sola_rag_stage_free(c->rag_stage);
    for (hipEvent_t e : c->bucket_ev) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (hipEvent_t e : c->ev_side)
        if (e) (void)hipEventDestroy(e);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->adam_tab) (void)hipFree(c->adam_tab);
    if (c->adam_ticket) (void)hipFree(c->adam_ticket);
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

static int find_weight(SolaCtx* c, const char* what, const char* name, const void* ptr, int64_t numel, Weight** out) {
    SOLA_ARG(c && name && ptr, "%s: null argument", what);
    auto it = c->index.find(name);
    if (it == c->index.end()) {
        sola_set_error("%s: unknown state_dict key '%s'", what, name);
        return SOLA_ERR_WEIGHT;
    }
    Weight& w = c->weights[it->second];
    if (w.numel != numel) {
        sola_set_error("%s: '%s' has %lld elements, expected %lld", what, name, (long long)numel, (long long)w.numel);
        return SOLA_ERR_WEIGHT;
    }
    if ((reinterpret_cast<uintptr_t>(ptr) & 15) != 0) {
        sola_set_error("%s: '%s' must be 16-byte aligned", what, name);
        return SOLA_ERR_WEIGHT;
    }
    *out = &w;
    return SOLA_OK;
}

extern "C" int sola_set_weight(SolaCtx* c, const char* name, const void* dev_ptr, int64_t numel) {
    Weight* w = nullptr;
    SOLA_TRY(find_weight(c, "set_weight", name, dev_ptr, numel, &w));
    if (w->ptr != static_cast<const float*>(dev_ptr)) c->adam_n = 0;  // the update table holds the old pointer: sola_adamw_step asks for a new bind
    w->ptr = static_cast<const float*>(dev_ptr);
    c->ws_dirty = true;
    c->lin16_dirty = true;
    return SOLA_OK;
}

extern "C" int sola_set_grad(SolaCtx* c, const char* name, void* dev_ptr, int64_t numel) {
    Weight* w = nullptr;
    SOLA_TRY(find_weight(c, "set_grad", name, dev_ptr, numel, &w));
    if (w->grad != static_cast<float*>(dev_ptr)) c->adam_n = 0;  // as in sola_set_weight
    w->grad = static_cast<float*>(dev_ptr);
    return SOLA_OK;
}

extern "C" int sola_weights_changed(SolaCtx* c) {
    SOLA_ARG(c, "weights_changed: null ctx");
    c->ws_dirty = true;
    c->lin16_dirty = true;
    return SOLA_OK;
}

extern "C" int sola_set_dropout(SolaCtx* c, float p_encoder, float p_attention, uint64_t seed) {
    SOLA_ARG(c && p_encoder >= 0.f && p_encoder < 1.f && p_attention >= 0.f && p_attention < 1.f, "set_dropout: bad argument");
    c->p_drop_encoder = p_encoder;
    c->p_drop_attention = p_attention;
    c->drop_seed = seed;
    return SOLA_OK;
}

// dropout applied by the per-stage entry points sola_group_norm[_backward] / sola_attention[_backward] (tests)
static thread_local DropoutCfg g_stage_drop = {0u, 0u, 0xFFFFFFFFu, 1.f, 0};
extern "C" int sola_set_stage_dropout(float p, uint64_t seed) {
    SOLA_ARG(p >= 0.f && p < 1.f, "set_stage_dropout: p=%f", (double)p);
    g_stage_drop = make_dropout(p, seed, 0);
    return SOLA_OK;
}

void sola_gemm_set_variant(int v);
void sola_gemm_set_glds(int v);
void sola_gemm_set_splitk(int v);
void sola_gemm_set_splitk_max(int v);
void sola_gemm_set_small_rows(int v);
void sola_gemm_set_small_nw8(int v);
void sola_gemm_set_f32_nw8(int v);
void sola_gemm_set_f32_persist(int v);
void sola_gemm_tn_set_persist(int v);
#ifdef SOLA_EXPERIMENTS
extern int g_gemm_f32p_ablate;
#endif
void sola_gemm_tn_set_nw8(int v);
extern int g_gemm_nw4, g_gemm_pp, g_gemm_k16, g_train_tn_tr, g_train_x16_keep, g_train_attn_cast;
extern int g_gemm_stagger, g_gemm_order, g_gemm_trace, g_gemm_ld;
extern int g_bwd_side_rows, g_bwd_group_rows, g_lang_shared_neg;
extern int g_infer_f32_rows;
void sola_gemm_set_ablate(int v);
void sola_gemm_set_persist(int v);
void sola_set_train_split_min_rows(int v);
void sola_gemm_set_glds_force(int v);
void sola_gn_set_variant(int v);
void sola_bilinear_set_staged(int v);
void sola_attn_set_variant(int v);
void sola_attn_set_target_blocks(int v);
void sola_iou_set_fused(int v);
void sola_attn_set_split_min_keys(int v);
void sola_attn_set_bwd_fused(int v);
void sola_attn_set_bwd_ablate(int v);
void sola_attn_set_bwd_bf16_mfma(int v);
void sola_train_set_gn_stats(int v);
void sola_gemm_set_slack_stagger(int v);
void sola_iou_set_packed(int v);
void sola_attn_set_f16_small(int v);
void sola_gn_set_h8(int v);
void sola_attn_set_spin(int v);
void sola_train_set_dw_f16(int v);
void sola_train_set_gn_cast(int v);
void sola_attn_set_splitm(int v);
void sola_attn_set_reg(int v);
void sola_attn_set_reg_minw(int v);
void sola_attn_set_res(int v);
void sola_attn_set_res_tiles(int v);
void sola_attn_set_res_shape(int v);
void sola_attn_set_res_splitm(int v);
void sola_attn_set_bwd_small(int v);
void sola_attn_set_bwd_blk(int v);
void sola_attn_set_bwd_rag_wave(int v);
void sola_attn_set_simple_train(int v);
void sola_gn_set_bwd_reg(int v);
void sola_gn_set_slices(int v);
void sola_gn_set_wide(int v);
void sola_gemm_set_gn_fuse(int v);
void sola_set_bwd_dual_cast(int v);
void sola_set_bwd_fused_bf16_cast(int v);
void sola_attn_set_ring(int v);
void sola_attn_set_f16_qpb(int v);
void sola_attn_set_ring_blocks(int v);
void sola_attn_set_ring_remap(int v);
void sola_attn_set_ring_ablate(int v);
void sola_attn_set_simple_remap(int v);
void sola_iou_set_shape(int v);
extern int g_train_bf16_store;
void sola_attn_set_bf16_mfma(int v);
void sola_attn_set_simple_db(int v);
void sola_pack_set_resample_lds(int v);
static int g_stage_split_math = 0;
// ---- self-test of the cross-lane primitives (common.h): wave_sum / wave_max run on v_permlane*_swap + DPP; every lane of every butterfly
// step must equal the ds_bpermute (__shfl_xor) form bit for bit - a compiler that folds a swap's two results (seen with the builtin form,
// common.h) or a wrong rotation would show here, not as a tolerance drift three kernels later
__global__ void selftest_wave_kernel(const float* __restrict__ in, int* __restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = in[i];
    float v = x, m = x;
    int wrong = 0;
    float w = x;
#pragma unroll
    for (int o = 32, s = 0; o > 0; o >>= 1, ++s) {
        v += __shfl_xor(v, o, 64);
        m = fmaxf(m, __shfl_xor(m, o, 64));
        w = s == 0 ? sum_xor32(w) : s == 1 ? sum_xor16(w) : s == 2 ? sum_xor8(w) : s == 3 ? sum_xor4(w) : s == 4 ? sum_xor2(w) : sum_xor1(w);
        wrong |= __float_as_uint(w) != __float_as_uint(v) ? 1 << s : 0;
    }
    wrong |= __float_as_uint(wave_sum(x)) != __float_as_uint(v) ? 64 : 0;
    wrong |= __float_as_uint(wave_max(x)) != __float_as_uint(m) ? 128 : 0;
    if (wrong) atomicOr(bad, wrong);
}
extern "C" int sola_selftest(void* stream_) {
    hipStream_t s = as_stream(stream_);
    constexpr int n = 64 * 1024;
    std::vector<float> h(n);
    unsigned r = 12345u;
    for (int i = 0; i < n; ++i) {
        r = r * 1664525u + 1013904223u;
        h[i] = ((int)(r >> 8) % 20001 - 10000) / 37.0f * ((i % 7) ? 1.f : 1e-3f);
    }
    float* d_in = nullptr;
    int* d_bad = nullptr;
    SOLA_HIP(hipMalloc(&d_in, n * sizeof(float)));
    if (hipMalloc(&d_bad, sizeof(int)) != hipSuccess) { (void)hipFree(d_in); sola_set_error("selftest: hipMalloc"); return SOLA_ERR_HIP; }
    int bad = 0, rc = SOLA_OK;
    if (hipMemcpyAsync(d_in, h.data(), n * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess || hipMemsetAsync(d_bad, 0, sizeof(int), s) != hipSuccess) rc = SOLA_ERR_HIP;
    if (rc == SOLA_OK) {
        hipLaunchKernelGGL(selftest_wave_kernel, dim3(n / 256), dim3(256), 0, s, d_in, d_bad);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = SOLA_ERR_HIP;
    }
    (void)hipFree(d_in);
    (void)hipFree(d_bad);
    if (rc != SOLA_OK) { sola_set_error("selftest: HIP error"); return rc; }
    if (bad) {
        sola_set_error("selftest: the vector-ALU wave reductions differ from the __shfl_xor butterfly (mask 0x%x: bits 0-5 = steps 32..1, 64 = wave_sum, 128 = wave_max)", bad);
        return SOLA_ERR_STATE;
    }
    return SOLA_OK;
}

extern "C" int sola_has_experiments(void) {
#ifdef SOLA_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int sola_tune(const char* key, int value) {
    SOLA_ARG(key, "tune: null key");
    if (!strcmp(key, "attn_stage_split_math")) { g_stage_split_math = value != 0; return SOLA_OK; }
    if (!strcmp(key, "gemm_variant")) { sola_gemm_set_variant(value); return SOLA_OK; }
    if (!strcmp(key, "gemm_glds")) { sola_gemm_set_glds(value); return SOLA_OK; }
    if (!strcmp(key, "bwd_side_rows")) { g_bwd_side_rows = value; return SOLA_OK; }
    if (!strcmp(key, "bwd_group_rows")) { g_bwd_group_rows = value; return SOLA_OK; }
    if (!strcmp(key, "lang_shared_neg")) { g_lang_shared_neg = value; return SOLA_OK; }
    if (!strcmp(key, "train_tn_tr")) { g_train_tn_tr = value; return SOLA_OK; }
    if (!strcmp(key, "train_x16_keep")) { g_train_x16_keep = value; return SOLA_OK; }
    if (!strcmp(key, "train_attn_cast")) { g_train_attn_cast = value; return SOLA_OK; }
    if (!strcmp(key, "gemm_splitk")) { sola_gemm_set_splitk(value); return SOLA_OK; }
    if (!strcmp(key, "gemm_splitk_max")) { sola_gemm_set_splitk_max(value); return SOLA_OK; }
    if (!strcmp(key, "gemm_small_rows")) { sola_gemm_set_small_rows(value); return SOLA_OK; }
    if (!strcmp(key, "infer_f32_rows")) { g_infer_f32_rows = value; return SOLA_OK; }
    if (!strcmp(key, "gemm_small_nw8")) { sola_gemm_set_small_nw8(value); return SOLA_OK; }
    if (!strcmp(key, "gemm_f32_nw8")) { sola_gemm_set_f32_nw8(value); return SOLA_OK; }
    if (!strcmp(key, "gemm_f32_persist")) { sola_gemm_set_f32_persist(value); return SOLA_OK; }
#ifdef SOLA_EXPERIMENTS
    if (!strcmp(key, "gemm_f32p_ablate")) { g_gemm_f32p_ablate = value; return SOLA_OK; }
#endif
    if (!strcmp(key, "gemm_tn_nw8")) { sola_gemm_tn_set_nw8(value); return SOLA_OK; }
    if (!strcmp(key, "gemm_tn_persist")) { sola_gemm_tn_set_persist(value); return SOLA_OK; }
    if (!strcmp(key, "gemm_glds_force")) { sola_gemm_set_glds_force(value); return SOLA_OK; }
    if (!strcmp(key, "train_split_min_rows")) { sola_set_train_split_min_rows(value); return SOLA_OK; }
    if (!strcmp(key, "gemm_persist")) { sola_gemm_set_persist(value); return SOLA_OK; }
    if (!strcmp(key, "gn_variant")) { sola_gn_set_variant(value); return SOLA_OK; }
    if (!strcmp(key, "bilinear_staged")) { sola_bilinear_set_staged(value); return SOLA_OK; }
    if (!strcmp(key, "attn_variant")) { sola_attn_set_variant(value); return SOLA_OK; }
    if (!strcmp(key, "attn_target_blocks")) { sola_attn_set_target_blocks(value); return SOLA_OK; }
    if (!strcmp(key, "iou_fused")) { sola_iou_set_fused(value); return SOLA_OK; }
    if (!strcmp(key, "iou_shape")) { sola_iou_set_shape(value); return SOLA_OK; }
    if (!strcmp(key, "train_bf16_store")) { g_train_bf16_store = value; return SOLA_OK; }
    if (!strcmp(key, "attn_bf16_mfma")) { sola_attn_set_bf16_mfma(value); return SOLA_OK; }
    if (!strcmp(key, "attn_bwd_bf16_mfma")) { sola_attn_set_bwd_bf16_mfma(value); return SOLA_OK; }
    if (!strcmp(key, "train_gn_stats")) { sola_train_set_gn_stats(value); return SOLA_OK; }
    if (!strcmp(key, "gemm_slack_stagger")) { sola_gemm_set_slack_stagger(value); return SOLA_OK; }
    if (!strcmp(key, "iou_packed")) { sola_iou_set_packed(value); return SOLA_OK; }
    if (!strcmp(key, "attn_split_min_keys")) { sola_attn_set_split_min_keys(value); return SOLA_OK; }
    if (!strcmp(key, "attn_splitm")) { sola_attn_set_splitm(value); return SOLA_OK; }
    if (!strcmp(key, "attn_reg")) { sola_attn_set_reg(value); return SOLA_OK; }
    if (!strcmp(key, "attn_res")) { sola_attn_set_res(value); return SOLA_OK; }
    if (!strcmp(key, "attn_res_tiles")) { sola_attn_set_res_tiles(value); return SOLA_OK; }
    if (!strcmp(key, "attn_res_shape")) { sola_attn_set_res_shape(value); return SOLA_OK; }
    if (!strcmp(key, "attn_bwd_small")) { sola_attn_set_bwd_small(value); return SOLA_OK; }
    if (!strcmp(key, "attn_bwd_blk")) { sola_attn_set_bwd_blk(value); return SOLA_OK; }
    if (!strcmp(key, "attn_bwd_rag_wave")) { sola_attn_set_bwd_rag_wave(value); return SOLA_OK; }
    if (!strcmp(key, "attn_bwd_fused")) { sola_attn_set_bwd_fused(value); return SOLA_OK; }
    if (!strcmp(key, "attn_f16_small")) { sola_attn_set_f16_small(value); return SOLA_OK; }
    if (!strcmp(key, "gn_h8")) { sola_gn_set_h8(value); return SOLA_OK; }
    if (!strcmp(key, "attn_spin")) { sola_attn_set_spin(value); return SOLA_OK; }
    if (!strcmp(key, "train_dw_f16")) { sola_train_set_dw_f16(value); return SOLA_OK; }
    if (!strcmp(key, "train_gn_cast")) { sola_train_set_gn_cast(value); return SOLA_OK; }
    if (!strcmp(key, "attn_simple_train")) { sola_attn_set_simple_train(value); return SOLA_OK; }
    if (!strcmp(key, "gn_bwd_reg")) { sola_gn_set_bwd_reg(value); return SOLA_OK; }
    if (!strcmp(key, "gn_slices")) { sola_gn_set_slices(value); return SOLA_OK; }
    if (!strcmp(key, "gn_wide")) { sola_gn_set_wide(value); return SOLA_OK; }
    if (!strcmp(key, "bwd_dual_cast")) { sola_set_bwd_dual_cast(value); return SOLA_OK; }
    if (!strcmp(key, "attn_f16_qpb")) { sola_attn_set_f16_qpb(value); return SOLA_OK; }
    if (!strcmp(key, "bwd_fused_bf16_cast")) { sola_set_bwd_fused_bf16_cast(value); return SOLA_OK; }
    if (!strcmp(key, "attn_simple_remap")) { sola_attn_set_simple_remap(value); return SOLA_OK; }
    if (!strcmp(key, "attn_simple_db")) { sola_attn_set_simple_db(value); return SOLA_OK; }
    if (!strcmp(key, "pack_resample_lds")) { sola_pack_set_resample_lds(value); return SOLA_OK; }
#ifdef SOLA_EXPERIMENTS  // closed experiments and measurement switches: EXPERIMENTS=1 builds only (make -C sola_amd/csrc EXPERIMENTS=1)
    if (!strcmp(key, "gemm_pp")) { g_gemm_pp = value; return SOLA_OK; }
    if (!strcmp(key, "gemm_nw4")) { g_gemm_nw4 = value; return SOLA_OK; }
    if (!strcmp(key, "gemm_k16")) { g_gemm_k16 = value; return SOLA_OK; }
    if (!strcmp(key, "gemm_stagger")) { g_gemm_stagger = value; return SOLA_OK; }
    if (!strcmp(key, "gemm_order")) { g_gemm_order = value; return SOLA_OK; }
    if (!strcmp(key, "gemm_trace")) { g_gemm_trace = value; return SOLA_OK; }
    if (!strcmp(key, "gemm_ld")) { g_gemm_ld = value; return SOLA_OK; }
    if (!strcmp(key, "gemm_ablate")) { sola_gemm_set_ablate(value); return SOLA_OK; }
    if (!strcmp(key, "attn_reg_minw")) { sola_attn_set_reg_minw(value); return SOLA_OK; }
    if (!strcmp(key, "attn_res_splitm")) { sola_attn_set_res_splitm(value); return SOLA_OK; }
    if (!strcmp(key, "attn_ring")) { sola_attn_set_ring(value); return SOLA_OK; }
    if (!strcmp(key, "attn_ring_blocks")) { sola_attn_set_ring_blocks(value); return SOLA_OK; }
    if (!strcmp(key, "attn_ring_remap")) { sola_attn_set_ring_remap(value); return SOLA_OK; }
    if (!strcmp(key, "attn_ring_ablate")) { sola_attn_set_ring_ablate(value); return SOLA_OK; }
    if (!strcmp(key, "attn_bwd_ablate")) { sola_attn_set_bwd_ablate(value); return SOLA_OK; }
    if (!strcmp(key, "gemm_gn_fuse")) { sola_gemm_set_gn_fuse(value); return SOLA_OK; }
#endif
    sola_set_error("tune: unknown key '%s'%s", key, sola_has_experiments() ? "" : " (experiment keys exist in EXPERIMENTS=1 builds only)");
    return SOLA_ERR_ARG;
}

extern "C" int sola_set_ws_policy(SolaCtx* c, int every) {
    SOLA_ARG(c, "set_ws_policy: null ctx");
    c->ws_every_forward = every != 0;
    return SOLA_OK;
}

extern "C" size_t sola_workspace_bytes(const SolaCtx* c, int B, int N, int T, int L) {
    if (!c || B <= 0 || N <= 0 || T <= 0 || L < 0) return 0;
    return make_plan(c, B, N, T, L, false).total;
}

extern "C" size_t sola_train_workspace_bytes(const SolaCtx* c, int B, int N, int T, int L) {
    if (!c || B <= 0 || N <= 0 || T <= 0 || L < 0) return 0;
    return make_plan(c, B, N, T, L, true).total;
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

extern "C" int sola_set_precision(SolaCtx* c, int precision) {
    SOLA_ARG(c && precision >= 0 && precision <= 3, "set_precision: 0 (f32), 1 (split-f16), 2 (f16 storage / f16 GEMM operands) or 3 (bf16 GEMM operands, training)");
    if (precision >= 1 && !c->lin16_buf) {
        const size_t D = c->cfg.lang_token_dim;
        size_t ws_total = 0;
        for (int i = 0; i < 6; ++i) ws_total += (size_t)c->conv[i].cout * c->conv[i].cin * c->conv[i].k;
        SOLA_HIP(hipSetDevice(c->device));
        SOLA_HIP(hipMalloc(&c->ws16_buf, ws_total * sizeof(float)));
        SOLA_HIP(hipMalloc(&c->lin16_buf, (size_t)c->cfg.n_layers * 12 * D * D * sizeof(float)));
        const size_t n_pairs = 2 + (size_t)c->cfg.n_layers * 12;
        SOLA_HIP(hipMalloc(&c->scal_buf, (2 * n_pairs + 2 + 4) * sizeof(float)));
        SOLA_HIP(hipMemset(c->scal_buf, 0, (2 * n_pairs + 2 + 4) * sizeof(float)));
        c->guard = reinterpret_cast<int*>(c->scal_buf + 2 * n_pairs);
        SOLA_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->guard_host), 2 * sizeof(int), hipHostMallocDefault));
        c->guard_host[0] = c->guard_host[1] = 0;
    }
    c->precision = precision;
    c->ws_dirty = true;
    c->lin16_dirty = true;
    return SOLA_OK;
}

extern "C" int sola_cast_sp16(const float* in, int ld_in, float* out, int ld_out, int64_t rows, int K, float scale, void* stream_) {
    return launch_cast_sp16(in, ld_in, out, ld_out, rows, K, scale, as_stream(stream_));
}

// split-f16 GEMM entry point (tests): a_sp [M,K] and w_sp [N,K] in the split format, f32 result
extern "C" int sola_gemm_nt_split(const float* a_sp, int lda, const float* w_sp, const float* bias, const float* r, int ldr,
                                  int r_sp16, float* cmat, int ldc, int c_sp16, int M, int N, int K, float out_scale, void* stream_) {
    SOLA_ARG(a_sp && w_sp && cmat, "gemm_nt_split: null argument");
    GemmDesc gd{};
    gd.nprob = 1;
    gd.p[0] = GemmProblem{a_sp, w_sp, bias, r, cmat};
    gd.M = M; gd.N = N; gd.K = K; gd.lda = lda; gd.ldr = ldr; gd.ldc = ldc;
    gd.arith = 1; gd.out_scale = out_scale; gd.r_sp16 = r_sp16; gd.c_sp16 = c_sp16;
    return launch_gemm(gd, as_stream(stream_));
}

// ---- 16-bit storage mode building blocks (tests) ----------------------------------------------------------------
int launch_attention_f16(const AttnDesc& d, hipStream_t s);
extern "C" int sola_cast_f16(const float* in, int ld_in, void* out, int ld_out, int64_t rows, int K, float scale, float* dev_scal, void* stream_) {
    return launch_cast_f16(in, ld_in, out, ld_out, rows, K, scale, dev_scal, as_stream(stream_));
}
extern "C" int sola_gemm_nt_f16(const void* a_h, int lda, const void* w_h, const float* bias, const void* r_h, int ldr, void* cmat, int ldc,
                                int c_is_f16, int M, int N, int K, float out_scale, void* stream_) {
    SOLA_ARG(a_h && w_h && cmat, "gemm_nt_f16: null argument");
    GemmDesc gd{};
    gd.nprob = 1;
    gd.p[0] = GemmProblem{static_cast<const float*>(a_h), static_cast<const float*>(w_h), bias, static_cast<const float*>(r_h), static_cast<float*>(cmat)};
    gd.M = M; gd.N = N; gd.K = K; gd.lda = lda; gd.ldr = ldr; gd.ldc = ldc;
    gd.arith = 2; gd.out_scale = out_scale; gd.r_f16 = r_h ? 1 : 0; gd.c_f16 = c_is_f16;
    return launch_gemm(gd, as_stream(stream_));
}
extern "C" int sola_attention_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int G, int H,
                                  int head_dim, int Sq, int Sk, int inner, int64_t q_outer, int64_t q_inner, int64_t q_rs,
                                  int64_t k_outer, int64_t k_inner, int64_t k_rs, float scale, void* stream_) {
    SOLA_ARG(q && k && v && o, "attention_f16: null argument");
    AttnDesc d{static_cast<const float*>(q), static_cast<const float*>(k), static_cast<const float*>(v), static_cast<float*>(o), ldq, ldk, ldv, ldo,
               G, H, head_dim, Sq, Sk, inner, q_outer, q_inner, q_rs, k_outer, k_inner, k_rs, scale, nullptr};
    return launch_attention_f16(d, as_stream(stream_));
}

// ---- building blocks of the bf16 training step's 16-bit storage (round 6), exposed for the parity tests --------------------------------
extern "C" int sola_gemm_nt_bf16(const void* a_b, int lda, const void* w_b, const float* bias, const void* r, int ldr, int r_is_bf16, void* cmat, int ldc,
                                 int c_is_bf16, int M, int N, int K, void* stream_) {
    SOLA_ARG(a_b && w_b && cmat, "gemm_nt_bf16: null argument");
    GemmDesc gd{};
    gd.nprob = 1;
    gd.p[0] = GemmProblem{static_cast<const float*>(a_b), static_cast<const float*>(w_b), bias, static_cast<const float*>(r), static_cast<float*>(cmat)};
    gd.M = M; gd.N = N; gd.K = K; gd.lda = lda; gd.ldr = ldr; gd.ldc = ldc;
    gd.arith = 2; gd.bf16 = 1; gd.out_scale = 1.f; gd.r_f16 = (r && r_is_bf16) ? 1 : 0; gd.c_f16 = c_is_bf16;
    return launch_gemm(gd, as_stream(stream_));
}
extern "C" int sola_attention_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, float* o, void* o_bf16, int ldo, int G, int H,
                                   int head_dim, int Sq, int Sk, int inner, int64_t q_outer, int64_t q_inner, int64_t q_rs,
                                   int64_t k_outer, int64_t k_inner, int64_t k_rs, float scale, float* lse, void* stream_) {
    SOLA_ARG(q && k && v && (o || o_bf16), "attention_bf16: null argument");
    AttnDesc d{static_cast<const float*>(q), static_cast<const float*>(k), static_cast<const float*>(v), o, ldq, ldk, ldv, ldo,
               G, H, head_dim, Sq, Sk, inner, q_outer, q_inner, q_rs, k_outer, k_inner, k_rs, scale, lse};
    d.drop = g_stage_drop;
    d.in_bf16 = 1;
    bool done = false;
    if (o_bf16) { d.o_cast = o_bf16; d.o_cast_fmt = 3; d.o_cast_done = &done; }
    SOLA_TRY(launch_attention(d, as_stream(stream_)));
    SOLA_ARG(!o_bf16 || done, "attention_bf16: this shape does not write the bf16 output");
    return SOLA_OK;
}
extern "C" int sola_attention_backward_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o, int o_bf16, const void* dout, int dout_bf16, int ldo,
                                            const float* lse, void* dq16, void* dk16, void* dv16, int ld_dq, int ld_dk, int ld_dv, float* dq_scratch,
                                            float* dvec, int G, int H, int head_dim, int Sq, int Sk, int inner, int64_t q_outer, int64_t q_inner,
                                            int64_t q_rs, int64_t k_outer, int64_t k_inner, int64_t k_rs, float scale, int64_t q_rows, float* scratch,
                                            size_t scratch_floats, void* stream_) {
    SOLA_ARG(q && k && v && o && dout && lse && dq16 && dk16 && dv16 && dq_scratch && dvec, "attention_backward_bf16: null argument");
    AttnBwdDesc d{static_cast<const float*>(q), static_cast<const float*>(k), static_cast<const float*>(v), static_cast<const float*>(o), static_cast<const float*>(dout), lse, dq_scratch, nullptr, nullptr, dvec,
                  ldq, ldk, ldv, ldo, ld_dq, ld_dk, ld_dv, G, H, head_dim, Sq, Sk, inner, q_outer, q_inner, q_rs, k_outer, k_inner, k_rs, scale};
    d.drop = g_stage_drop;
    d.part = scratch; d.part_floats = scratch_floats; d.part_rows = q_rows;
    d.io_bf16 = 1; d.dq16 = dq16; d.dk16 = dk16; d.dv16 = dv16;
    d.dout_bf16 = dout_bf16 ? 1 : 0;
    d.o_bf16 = o_bf16 ? 1 : 0;
    return launch_attention_bwd(d, as_stream(stream_));
}

extern "C" int sola_cast_sp16_auto(const float* in, int ld_in, float* out, int ld_out, int64_t rows, int K, float* scal, void* stream_) {
    return launch_cast_sp16_auto(in, ld_in, out, ld_out, rows, K, scal, as_stream(stream_));
}

extern "C" int sola_gemm_nt_split_scaled(const float* a_sp, int lda, const float* w_sp, const float* bias, const float* r, int ldr,
                                         int r_sp16, float* cmat, int ldc, int c_sp16, int M, int N, int K, float out_scale,
                                         const float* out_scale_dev, void* stream_) {
    SOLA_ARG(a_sp && w_sp && cmat, "gemm_nt_split_scaled: null argument");
    GemmDesc gd{};
    gd.nprob = 1;
    gd.p[0] = GemmProblem{a_sp, w_sp, bias, r, cmat};
    gd.M = M; gd.N = N; gd.K = K; gd.lda = lda; gd.ldr = ldr; gd.ldc = ldc;
    gd.arith = 1; gd.out_scale = out_scale; gd.out_scale_dev = out_scale_dev; gd.r_sp16 = r_sp16; gd.c_sp16 = c_sp16;
    return launch_gemm(gd, as_stream(stream_));
}

// Reads the guard words the split-f16 forward left (ctx.h) and reports whether the call must be repeated in exact f32.
// Synchronises the stream (4 bytes back to the host), so it is skipped while the stream is being captured into a graph.
int sola_split_guard_tripped(SolaCtx* c, hipStream_t s, bool* tripped) {
    *tripped = false;
    if (!c->split_guard || !c->guard) return SOLA_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return SOLA_OK;
    SOLA_HIP(hipMemcpyAsync(c->guard_host, c->guard, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    SOLA_HIP(hipStreamSynchronize(s));
    *tripped = (c->guard_host[0] | c->guard_host[1]) != 0;
    if (!c->guard_host[1]) c->weight_range_bad = false;  // the current weights are fine (again)
    if (c->guard_host[1] && !c->weight_range_bad) {
        // the WEIGHTS put a GroupNorm's output outside the split-f16 pairs' range: every call would run the split pass and then
        // the exact-f32 repeat (2x the cost) until the weights change - remember it and go straight to f32 from the next call on
        c->weight_range_bad = true;
        fprintf(stderr, "sola_hip: GroupNorm weights outside the range of the split-f16 activations (rms of (gamma, beta) not in "
                        "[2^-6, 2^9]): precision 1 calls run on the exact-f32 kernels until the weights change\n");
    }
    return SOLA_OK;
}
// precision 1 only: the weight-time guard bit is known to be set for the current weights -> skip the split pass
static bool split_known_out_of_range(const SolaCtx* c) { return c->precision == 1 && c->split_guard && c->weight_range_bad && !c->lin16_dirty; }

// sola_tune "infer_f32_rows" (round 4): a precision-1 inference call over at most this many object-token rows (one sample per call: the
// reference's inference.py / evaluator batch size) runs the exact-f32 kernels.  Since the few-row GEMM shape (gemm.hip) they are FASTER there
// than the split-f16 pass - whose casts, guard read-back (a stream synchronisation per call) and 64x64 split-K GEMMs are fixed costs
// - and exact: one sample of the headline shape 0.61 -> 0.45 ms per call (tools/infer_one_probe.py).  0 = never.
int g_infer_f32_rows = 4096;
// a temporary switch of the ctx's arithmetic for one call: restored on every exit, exceptions included (a throwing plan left the
// ctx on the exact-f32 kernels for good, silently)
struct PrecScope {
    SolaCtx* c;
    int saved;
    PrecScope(SolaCtx* ctx, int p) : c(ctx), saved(ctx->precision) { c->precision = p; }
    ~PrecScope() { c->precision = saved; }
    PrecScope(const PrecScope&) = delete;
    PrecScope& operator=(const PrecScope&) = delete;
};
static bool few_rows_f32(const SolaCtx* c, long long rows0) { return c->precision == 1 && g_infer_f32_rows > 0 && rows0 <= g_infer_f32_rows; }

extern "C" int sola_forward(SolaCtx* c, const float* obj, const float* lang, int B, int N, int T, int L, float* score_map,
                            float* score_tokens, void* workspace, size_t ws_bytes, void* stream_) {
    hipStream_t s = as_stream(stream_);
    SOLA_ARG(!c || c->precision != 3, "forward: precision 3 (bf16 GEMM operands) is a TRAINING mode; inference runs precision 0, 1 or 2");
    if (c && B > 0 && N > 0 && T > 0 && few_rows_f32(c, (long long)B * N * T)) {
        PrecScope f32(c, 0);
        return sola_forward_impl(c, obj, lang, B, N, T, L, score_map, score_tokens, workspace, ws_bytes, s, false);
    }
    if (c && split_known_out_of_range(c)) {
        c->split_fallbacks += 1;
        PrecScope f32(c, 0);
        return sola_forward_impl(c, obj, lang, B, N, T, L, score_map, score_tokens, workspace, ws_bytes, s, false);
    }
    if (c && c->precision >= 1) {
        const int prec = c->precision;
        if (prec == 2) SOLA_TRY(sola_forward_f16_impl(c, obj, lang, B, N, T, L, score_map, score_tokens, workspace, ws_bytes, s));
        else SOLA_TRY(sola_forward_fast_impl(c, obj, lang, B, N, T, L, score_map, score_tokens, workspace, ws_bytes, s));
        bool tripped = false;
        SOLA_TRY(sola_split_guard_tripped(c, s, &tripped));
        if (!tripped) return SOLA_OK;
        // a value left the range the split-f16 pairs cover (or the weights say it would): same call, exact-f32 kernels.
        // The f32 plan is a subset of the split plan, so the workspace fits.
        c->split_fallbacks += 1;
        PrecScope f32(c, 0);
        return sola_forward_impl(c, obj, lang, B, N, T, L, score_map, score_tokens, workspace, ws_bytes, s, false);
    }
    return sola_forward_impl(c, obj, lang, B, N, T, L, score_map, score_tokens, workspace, ws_bytes, s, false);
}

extern "C" size_t sola_ragged_workspace_bytes(const SolaCtx* c, const SolaRaggedBatch* batch) {
    if (!c || !batch) return 0;
    // the exact-f32 repeat of a guarded split-f16 call runs in the same arena: size it for the larger of the two plans
    try {
        size_t n = sola_ragged_workspace_bytes_impl(c, batch, c->precision);
        if (c->precision >= 1) n = std::max(n, sola_ragged_workspace_bytes_impl(c, batch, 0));  // room for the guard's exact-f32 repeat
        return n;
    } catch (const std::exception& e) {
        sola_set_error("ragged_workspace_bytes: %s", e.what());
        return 0;
    }
}

extern "C" int sola_forward_ragged(SolaCtx* c, const float* obj, const float* lang, const SolaRaggedBatch* batch, float* score_map,
                                   float* score_tokens, void* workspace, size_t ws_bytes, void* stream_) {
    hipStream_t s = as_stream(stream_);
    SOLA_ARG(!c || c->precision != 3, "forward_ragged: precision 3 (bf16 GEMM operands) is a TRAINING mode; inference runs precision 0, 1 or 2");
    try {  // the host-side plan allocates; nothing may throw across the ABI
        if (c && batch && c->precision == 1 && batch->n_videos > 0 && batch->video_tracks && batch->video_frames) {
            long long rows0 = 0;
            for (int v = 0; v < batch->n_videos; ++v) rows0 += (long long)batch->video_tracks[v] * batch->video_frames[v];
            if (few_rows_f32(c, rows0)) {  // see g_infer_f32_rows
                PrecScope f32(c, 0);
                return sola_forward_ragged_impl(c, obj, lang, batch, score_map, score_tokens, workspace, ws_bytes, s);
            }
        }
        if (c && split_known_out_of_range(c)) {
            c->split_fallbacks += 1;
            PrecScope f32(c, 0);
            return sola_forward_ragged_impl(c, obj, lang, batch, score_map, score_tokens, workspace, ws_bytes, s);
        }
        SOLA_TRY(sola_forward_ragged_impl(c, obj, lang, batch, score_map, score_tokens, workspace, ws_bytes, s));
        if (c->precision < 1) return SOLA_OK;
        bool tripped = false;
        SOLA_TRY(sola_split_guard_tripped(c, s, &tripped));
        if (!tripped) return SOLA_OK;
        c->split_fallbacks += 1;
        PrecScope f32(c, 0);
        return sola_forward_ragged_impl(c, obj, lang, batch, score_map, score_tokens, workspace, ws_bytes, s);
    } catch (const std::exception& e) {
        sola_set_error("forward_ragged: %s", e.what());
        return SOLA_ERR_ARG;
    }
}

extern "C" int sola_loss_ragged(const float* score_map, const float* score_tokens, const float* labels, const float* pos,
                                const float* neg, int64_t neg_batch_stride, int n_samples, const int32_t* dev_track_offsets,
                                int max_tracks, int64_t total_tracks, int D, int n_neg, float pw, float temperature, float aw,
                                float* loss3, int32_t* neg_argmax, void* scratch, size_t scratch_bytes, void* stream_) {
    SOLA_ARG(score_map && score_tokens && labels && pos && neg && loss3 && scratch && dev_track_offsets, "loss_ragged: null argument");
    SOLA_ARG(n_samples > 0 && max_tracks > 0 && total_tracks > 0 && D > 0 && n_neg > 0, "loss_ragged: bad sizes");
    const size_t need = (size_t)total_tracks * 3 * sizeof(float);
    if (scratch_bytes < need) {
        sola_set_error("loss_ragged: scratch %zu bytes < required %zu", scratch_bytes, need);
        return SOLA_ERR_WORKSPACE;
    }
    LossDesc d{};
    d.score_map = score_map; d.score_tokens = score_tokens; d.labels = labels; d.pos = pos;
    d.neg = neg; d.neg_batch_stride = neg_batch_stride;
    d.B = n_samples; d.N = max_tracks; d.D = D; d.n_neg = n_neg;
    d.pos_w = pw; d.temp_scale = expf(temperature); d.align_w = aw;
    d.terms = static_cast<float*>(scratch); d.loss3 = loss3; d.neg_argmax = neg_argmax;
    d.trk_off = dev_track_offsets;
    return launch_loss(d, as_stream(stream_));
}

extern "C" int sola_set_split_guard(SolaCtx* c, int enable) {
    SOLA_ARG(c, "set_split_guard: null ctx");
    c->split_guard = enable != 0;
    return SOLA_OK;
}

extern "C" int sola_set_x16_arena(SolaCtx* c, void* dev_ptr, size_t bytes) {
    SOLA_ARG(c, "set_x16_arena: null ctx");
    SOLA_ARG((reinterpret_cast<uintptr_t>(dev_ptr) & 255) == 0, "set_x16_arena: the arena must be 256-byte aligned");
    c->x16.clear();  // casts listed by an earlier forward lived in the previous arena
    c->x16_arena = static_cast<char*>(dev_ptr);
    c->x16_cap = dev_ptr ? bytes : 0;
    c->x16_used = 0;
    if (!dev_ptr) c->x16_need = 0;  // a released arena: the next forward states its need afresh
    return SOLA_OK;
}

extern "C" int sola_x16_arena_info(const SolaCtx* c, size_t* need_bytes, size_t* capacity_bytes, size_t* used_bytes) {
    SOLA_ARG(c, "x16_arena_info: null ctx");
    if (need_bytes) *need_bytes = c->x16_need;
    if (capacity_bytes) *capacity_bytes = c->x16_cap;
    if (used_bytes) *used_bytes = c->x16_used;
    return SOLA_OK;
}

extern "C" int sola_split_fallback_count(const SolaCtx* c, int64_t* count, int32_t* last_guard) {
    SOLA_ARG(c && count, "split_fallback_count: null argument");
    *count = c->split_fallbacks;
    if (last_guard) *last_guard = c->guard_host ? (c->guard_host[0] | c->guard_host[1]) : 0;
    return SOLA_OK;
}

extern "C" int sola_forward_train(SolaCtx* c, const float* obj, const float* lang, int B, int N, int T, int L,
                                  float* score_map, float* score_tokens, void* workspace, size_t ws_bytes, void* stream_) {
    return sola_forward_impl(c, obj, lang, B, N, T, L, score_map, score_tokens, workspace, ws_bytes, as_stream(stream_), true);
}

// ---- ragged training step: many samples of different (N, T, L) per optimizer step (train.py:62-137 runs ONE per step) --------
extern "C" size_t sola_train_ragged_workspace_bytes(const SolaCtx* c, const SolaRaggedBatch* batch) {
    if (!c || !batch) return 0;
    try {
        RagShape r;
        if (rag_shape(c, batch, r) != SOLA_OK) return 0;
        return make_plan_ragged(c, r, true).total;
    } catch (const std::exception& e) {
        sola_set_error("train_ragged_workspace_bytes: %s", e.what());
        return 0;
    }
}

extern "C" int sola_forward_train_ragged(SolaCtx* c, const float* obj, const float* lang, const SolaRaggedBatch* batch,
                                         float* score_map, float* score_tokens, void* workspace, size_t ws_bytes, void* stream_) {
    SOLA_ARG(c && batch, "forward_train_ragged: null argument");
    try {  // the host-side plan allocates; nothing may throw across the ABI
        RagShape r;
        SOLA_TRY(rag_shape(c, batch, r));
        SOLA_ARG(r.identity, "forward_train_ragged: one sample per video, in order (n_samples == n_videos, sample_video[i] == i): "
                             "every training sample runs its own encoder pass under its own dropout masks");
        return sola_forward_impl(c, obj, lang, 0, 0, 0, 0, score_map, score_tokens, workspace, ws_bytes, as_stream(stream_), true, &r);
    } catch (const std::exception& e) {
        sola_set_error("forward_train_ragged: %s", e.what());
        return SOLA_ERR_ARG;
    }
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

extern "C" int sola_loss_backward(const float* score_map, const float* score_tokens, const float* labels, const float* pos,
                                  const float* neg, int64_t neg_batch_stride, int B, int N, int D, int n_neg, float pw,
                                  float temperature, float aw, const float* g3, float* d_score_map, float* d_score_tokens,
                                  float* d_neg, void* scratch, size_t scratch_bytes, void* stream_) {
    SOLA_ARG(score_map && score_tokens && labels && pos && neg && g3 && d_score_map && d_score_tokens && scratch,
             "loss_backward: null argument");
    SOLA_ARG(B > 0 && N > 0 && D > 0 && n_neg > 0, "loss_backward: bad sizes");
    // shared negatives (stride 0): the per-sample gradients are staged in the scratch and summed over b into d_neg [n_neg,D]
    const bool shared = neg_batch_stride == 0 && d_neg != nullptr;
    const size_t coef_floats = ((size_t)B * N * n_neg + 63) & ~(size_t)63;
    const size_t need = (coef_floats + (shared ? (size_t)B * n_neg * D : 0)) * sizeof(float);
    if (scratch_bytes < need) {
        sola_set_error("loss_backward: scratch %zu bytes < required %zu", scratch_bytes, need);
        return SOLA_ERR_WORKSPACE;
    }
    float* stage = static_cast<float*>(scratch) + coef_floats;
    LossBwdDesc d{};
    d.score_map = score_map; d.score_tokens = score_tokens; d.labels = labels; d.pos = pos; d.neg = neg;
    d.neg_batch_stride = neg_batch_stride; d.B = B; d.N = N; d.D = D; d.n_neg = n_neg;
    d.pos_w = pw; d.temp_scale = expf(temperature); d.align_w = aw; d.g3 = g3;
    d.d_score = d_score_map; d.d_tok = d_score_tokens; d.coef = static_cast<float*>(scratch);
    d.d_neg = shared ? stage : d_neg;
    SOLA_TRY(launch_loss_bwd(d, as_stream(stream_)));
    if (shared) return launch_neg_token_grad(nullptr, nullptr, stage, d_neg, B, 0, n_neg, D, as_stream(stream_));
    return SOLA_OK;
}

extern "C" int sola_loss_backward_ragged(const float* score_map, const float* score_tokens, const float* labels, const float* pos,
                                         const float* neg, int64_t neg_batch_stride, int n_samples, const int32_t* dev_track_offsets,
                                         int max_tracks, int64_t total_tracks, int D, int n_neg, float pw, float temperature, float aw,
                                         const float* g3, float* d_score_map, float* d_score_tokens, float* d_neg, void* scratch,
                                         size_t scratch_bytes, void* stream_) {
    SOLA_ARG(score_map && score_tokens && labels && pos && neg && g3 && d_score_map && d_score_tokens && scratch && dev_track_offsets,
             "loss_backward_ragged: null argument");
    SOLA_ARG(n_samples > 0 && max_tracks > 0 && total_tracks > 0 && D > 0 && n_neg > 0, "loss_backward_ragged: bad sizes");
    const bool shared = neg_batch_stride == 0 && d_neg != nullptr;
    const size_t coef_floats = ((size_t)total_tracks * n_neg + 63) & ~(size_t)63;
    const size_t need = (coef_floats + (shared ? (size_t)n_samples * n_neg * D : 0)) * sizeof(float);
    if (scratch_bytes < need) {
        sola_set_error("loss_backward_ragged: scratch %zu bytes < required %zu", scratch_bytes, need);
        return SOLA_ERR_WORKSPACE;
    }
    float* stage = static_cast<float*>(scratch) + coef_floats;
    LossBwdDesc d{};
    d.score_map = score_map; d.score_tokens = score_tokens; d.labels = labels; d.pos = pos; d.neg = neg;
    d.neg_batch_stride = neg_batch_stride; d.B = n_samples; d.N = max_tracks; d.D = D; d.n_neg = n_neg;
    d.pos_w = pw; d.temp_scale = expf(temperature); d.align_w = aw; d.g3 = g3;
    d.d_score = d_score_map; d.d_tok = d_score_tokens; d.coef = static_cast<float*>(scratch);
    d.d_neg = shared ? stage : d_neg;
    d.trk_off = dev_track_offsets; d.total_tracks = total_tracks;
    SOLA_TRY(launch_loss_bwd(d, as_stream(stream_)));
    if (shared) return launch_neg_token_grad(nullptr, nullptr, stage, d_neg, n_samples, 0, n_neg, D, as_stream(stream_));
    return SOLA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// gradient norms / clipping (module/module.py:164-199, train.py:120-122)
// ---------------------------------------------------------------------------------------------------------------
extern "C" size_t sola_grad_sqnorms_scratch_bytes(int n, const int64_t* numel) {
    if (n <= 0 || !numel) return 0;
    return mt_sqnorm_scratch_bytes(n, reinterpret_cast<const long long*>(numel));
}

extern "C" int sola_grad_sqnorms(const float* const* dev_grads, const int64_t* numel, const int32_t* group, int n,
                                 int n_groups, double* dev_out, void* scratch, size_t scratch_bytes, void* stream_) {
    SOLA_ARG(dev_grads && numel && group && dev_out && scratch, "grad_sqnorms: null argument");
    return launch_mt_sqnorm(dev_grads, reinterpret_cast<const long long*>(numel), group, n, n_groups, dev_out, scratch,
                            scratch_bytes, as_stream(stream_));
}

extern "C" int sola_grad_clip(float* const* dev_grads, const int64_t* numel, int n, const double* dev_total_sq,
                              float max_norm, void* stream_) {
    SOLA_ARG(dev_grads && numel && dev_total_sq, "grad_clip: null argument");
    return launch_mt_clip(dev_grads, reinterpret_cast<const long long*>(numel), n, dev_total_sq, max_norm, as_stream(stream_));
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

extern "C" int sola_ws_backward(const float* w, const float* dwstd, int cout, int cin, int k, float* dw, void* stream_) {
    SOLA_ARG(w && dwstd && dw && cout > 0, "ws_backward: bad argument");
    WsBwdLayer l{w, dwstd, dw, cout, cin, k};
    return launch_ws_backward(&l, 1, as_stream(stream_));
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

extern "C" int sola_gemm_nn(const float* a, int lda, const float* w0, const float* w1, const float* w2, int w_rows, const float* r, int ldr,
                            float* cmat, int ldc, int M, int N, int K, void* stream_) {
    SOLA_ARG(a && w0 && cmat && w_rows > 0, "gemm_nn: null argument");
    GemmDesc gd{};
    gd.nprob = 1;
    gd.p[0] = GemmProblem{a, nullptr, nullptr, r, cmat};
    gd.M = M; gd.N = N; gd.K = K; gd.lda = lda; gd.ldr = ldr; gd.ldc = ldc;
    gd.w_nn[0] = w0; gd.w_nn[1] = w1; gd.w_nn[2] = w2; gd.w_nn_rows = w_rows;
    return launch_gemm(gd, as_stream(stream_));
}

extern "C" int sola_gemm_tn_f16(const float* a, int lda, const float* b, int ldb, float* cmat, int M, int N, int K, int fmt, void* scratch,
                                size_t scratch_bytes, void* stream_) {
    SOLA_ARG(a && b && cmat && scratch && (fmt == 1 || fmt == 2), "gemm_tn_f16: null argument or fmt %d", fmt);
    GemmTnSplitDesc d{};
    d.A[0] = a; d.B[0] = b; d.C[0] = cmat; d.nprob = 1; d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.pure = fmt;
    d.scratch = static_cast<float*>(scratch); d.scratch_bytes = scratch_bytes;
    return launch_gemm_tn_split(d, as_stream(stream_));
}
extern "C" int sola_conv1d_cl_wgrad_f16(const float* x, const float* dy, float* dwstd, int R, int T_in, int cin, int cout, int k, int stride,
                                        int pad, int fmt, void* scratch, size_t scratch_bytes, void* stream_) {
    SOLA_ARG(x && dy && dwstd && scratch && (fmt == 1 || fmt == 2) && R > 0 && T_in > 0 && k >= 1 && k <= 8 && stride >= 1, "conv1d_cl_wgrad_f16: bad argument");
    const int T_out = (T_in + 2 * pad - k) / stride + 1;
    SOLA_ARG(T_out > 0 && gemm_tn_split_supported(R * T_out, cout, k * cin), "conv1d_cl_wgrad_f16: R*T_out=%d cout=%d k*cin=%d", R * T_out, cout, k * cin);
    GemmTnSplitDesc d{};
    d.nprob = 1; d.A[0] = dy; d.B[0] = x; d.C[0] = dwstd; d.M = R * T_out; d.N = cout; d.K = k * cin; d.lda = cout; d.ldb = cin; d.pure = fmt;
    d.conv = (k > 1 || stride > 1 || pad > 0) ? 1 : 0; d.T_in = T_in; d.T_out = T_out; d.stride = stride; d.pad = pad; d.Cin = cin;
    d.scratch = static_cast<float*>(scratch); d.scratch_bytes = scratch_bytes;
    return launch_gemm_tn_split(d, as_stream(stream_));
}
extern "C" size_t sola_gemm_tn_scratch_bytes(int M, int N, int K) { return gemm_tn_scratch_bytes(M, N, K); }
extern "C" size_t sola_gemm_tn_split_scratch_bytes(int M, int N, int K) { return gemm_tn_split_supported(M, N, K) ? gemm_tn_split_scratch_bytes(M, N, K, 1) : 0; }
extern "C" int sola_gemm_tn_split(const float* a, int lda, const float* b, int ldb, float* cmat, int M, int N, int K, void* scratch,
                                  size_t scratch_bytes, void* stream_) {
    SOLA_ARG(a && b && cmat && scratch, "gemm_tn_split: null argument");
    GemmTnSplitDesc d{};
    d.A[0] = a; d.B[0] = b; d.C[0] = cmat; d.nprob = 1; d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb;
    d.scratch = static_cast<float*>(scratch); d.scratch_bytes = scratch_bytes;
    return launch_gemm_tn_split(d, as_stream(stream_));
}

extern "C" int sola_gemm_tn(const float* a, int lda, const float* b, int ldb, float* cmat, float* bias_grad, int M, int N,
                            int K, void* scratch, size_t scratch_bytes, void* stream_) {
    SOLA_ARG(a && b && cmat && scratch, "gemm_tn: null argument");
    GemmTnDesc d{};
    d.A = a; d.B = b; d.C = cmat; d.bias_grad = bias_grad; d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb;
    d.scratch = static_cast<float*>(scratch); d.scratch_bytes = scratch_bytes;
    return launch_gemm_tn(d, as_stream(stream_));
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

// Backward of sola_conv1d_cl: dx [R,T_in,cin] (optional), dwstd [cout,k*cin], dbias [cout].
// Scratch: max(sola_gemm_tn_scratch_bytes(R*T_out, cout, k*cin), 4*cout*k*cin) bytes.
extern "C" int sola_conv1d_cl_backward(const float* x, const float* wstd, const float* dy, float* dx, float* dwstd,
                                       float* dbias, int R, int T_in, int cin, int cout, int k, int stride, int pad,
                                       void* scratch, size_t scratch_bytes, void* stream_) {
    SOLA_ARG(x && wstd && dy && dwstd && scratch && R > 0 && T_in > 0 && k >= 1 && stride >= 1, "conv1d_cl_backward: bad argument");
    const int T_out = (T_in + 2 * pad - k) / stride + 1;
    hipStream_t s = as_stream(stream_);
    const bool gather = (k > 1 || stride > 1 || pad > 0);
    GemmTnDesc d{};
    d.A = dy; d.B = x; d.C = dwstd; d.bias_grad = dbias; d.M = R * T_out; d.N = cout; d.K = k * cin; d.lda = cout; d.ldb = cin;
    d.conv = gather ? 1 : 0; d.T_in = T_in; d.T_out = T_out; d.stride = stride; d.pad = pad; d.Cin = cin;
    d.scratch = static_cast<float*>(scratch); d.scratch_bytes = scratch_bytes;
    SOLA_TRY(launch_gemm_tn(d, s));
    if (!dx) return SOLA_OK;
    const size_t need = (size_t)cout * k * cin * sizeof(float);
    if (scratch_bytes < need) {
        sola_set_error("conv1d_cl_backward: scratch %zu < %zu", scratch_bytes, need);
        return SOLA_ERR_WORKSPACE;
    }
    float* wt = static_cast<float*>(scratch);  // [cin][kk*cout + co]
    for (int kk = 0; kk < k; ++kk)
        SOLA_TRY(launch_transpose(wstd + (size_t)kk * cin, wt, cout, cin, k * cin, k * cout, kk * cout, s));
    GemmDesc gd{};
    gd.nprob = 1;
    gd.p[0] = GemmProblem{dy, wt, nullptr, nullptr, dx};
    gd.M = R * T_in; gd.N = cin; gd.K = k * cout; gd.lda = cout; gd.ldc = cin;
    gd.conv = gather ? 2 : 0; gd.T_in = T_out; gd.T_out = T_in; gd.stride = stride; gd.pad = pad; gd.Cin = cout;
    return launch_gemm(gd, s);
}

// The same backward on the split-f16 MFMA path (what sola_backward runs under precision 1): dwstd through
// gemm_tn_split.hip with per-tap transposing im2col casts, dx as ONE GEMM over the output steps + a col2im gather.
// Needs cout % 128 == 0, cin % 8 == 0, R*T_out >= 64.
static size_t conv_bwd_split_layout(int R, int T_in, int cin, int cout, int k, int stride, int pad, size_t off[5]) {
    const int T_out = (T_in + 2 * pad - k) / stride + 1;
    const size_t M = (size_t)R * T_out;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t o = 0;
    off[0] = o; o += up(gemm_tn_split_scratch_bytes((int)M, cout, k * cin, 1));
    off[1] = o; o += up(M * cout * sizeof(float));                      // dy_sp
    off[2] = o; o += up((size_t)k * cin * cout * sizeof(float));        // wt_sp
    off[3] = o; o += up(M * (size_t)k * cin * sizeof(float));           // z
    off[4] = o; o += 256;                                               // scal
    return o;
}
extern "C" size_t sola_conv1d_cl_backward_split_scratch_bytes(int R, int T_in, int cin, int cout, int k, int stride, int pad) {
    size_t off[5];
    const int T_out = (T_in + 2 * pad - k) / stride + 1;
    if (R <= 0 || T_out <= 0 || cout % 128 || cin % 8 || !gemm_tn_split_supported(R * T_out, cout, k * cin)) return 0;
    return conv_bwd_split_layout(R, T_in, cin, cout, k, stride, pad, off);
}
extern "C" int sola_conv1d_cl_backward_split(const float* x, const float* wstd, const float* dy, float* dx, float* dwstd,
                                             float* dbias, int R, int T_in, int cin, int cout, int k, int stride, int pad,
                                             void* scratch, size_t scratch_bytes, void* stream_) {
    SOLA_ARG(x && wstd && dy && dwstd && scratch && R > 0 && T_in > 0 && k >= 1 && stride >= 1, "conv1d_cl_backward_split: bad argument");
    const size_t need = sola_conv1d_cl_backward_split_scratch_bytes(R, T_in, cin, cout, k, stride, pad);
    SOLA_ARG(need > 0, "conv1d_cl_backward_split: needs cout %% 128 == 0, cin %% 8 == 0, R*T_out >= 64 (cout=%d cin=%d)", cout, cin);
    if (scratch_bytes < need) {
        sola_set_error("conv1d_cl_backward_split: scratch %zu < %zu", scratch_bytes, need);
        return SOLA_ERR_WORKSPACE;
    }
    size_t off[5];
    conv_bwd_split_layout(R, T_in, cin, cout, k, stride, pad, off);
    char* base = static_cast<char*>(scratch);
    const int T_out = (T_in + 2 * pad - k) / stride + 1, M = R * T_out;
    hipStream_t s = as_stream(stream_);
    GemmTnSplitDesc d{};
    d.nprob = 1; d.A[0] = dy; d.B[0] = x; d.C[0] = dwstd; d.M = M; d.N = cout; d.K = k * cin; d.lda = cout; d.ldb = cin;
    d.conv = (k > 1 || stride > 1 || pad > 0) ? 1 : 0; d.T_in = T_in; d.T_out = T_out; d.stride = stride; d.pad = pad; d.Cin = cin;
    d.scratch = reinterpret_cast<float*>(base + off[0]); d.scratch_bytes = off[1] - off[0];
    SOLA_TRY(launch_gemm_tn_split(d, s));
    if (dbias) SOLA_TRY(launch_colsum(dy, dbias, 1, M, cout, cout, 1.f, 0, nullptr, 0, s));
    if (!dx) return SOLA_OK;
    float* dy_sp = reinterpret_cast<float*>(base + off[1]);
    float* wt_sp = reinterpret_cast<float*>(base + off[2]);
    float* z = reinterpret_cast<float*>(base + off[3]);
    float* scal = reinterpret_cast<float*>(base + off[4]);
    SOLA_TRY(launch_cast_sp16_auto(dy, cout, dy_sp, cout, M, cout, scal, s));
    SOLA_TRY(launch_cast_sp16_t(wstd, k * cin, wt_sp, cout, cout, k * cin, nullptr, s));
    GemmDesc gd{};
    gd.nprob = 1;
    const bool gather = d.conv != 0;
    gd.p[0] = GemmProblem{dy_sp, wt_sp, nullptr, nullptr, gather ? z : dx};
    gd.M = M; gd.N = k * cin; gd.K = cout; gd.lda = cout; gd.ldc = k * cin;
    gd.arith = 1; gd.out_scale = 1.f; gd.out_scale_dev = scal + 1;
    SOLA_TRY(launch_gemm(gd, s));
    if (gather) SOLA_TRY(launch_col2im(z, dx, R, T_in, T_out, cin, k, stride, pad, s));
    return SOLA_OK;
}

extern "C" int sola_group_norm(const float* x, float* y, float* y2, const float* pe, const float* gamma,
                               const float* beta, int n_inst, int inner, int64_t outer_stride, int64_t inner_stride,
                               int64_t tok_stride, int ntok, int C, int groups, float eps, float slope, int leaky,
                               void* stream_) {
    SOLA_ARG(x && y && gamma && beta, "group_norm: null argument");
    GroupNormDesc d{x, y, y2, pe, gamma, beta, n_inst, inner, outer_stride, inner_stride, tok_stride, ntok, C, groups, eps, slope, leaky};
    d.drop = g_stage_drop;
    return launch_group_norm(d, as_stream(stream_));
}

// dgamma / dbeta [C]; scratch: 2 * n_inst * C floats
extern "C" int sola_group_norm_backward(const float* x, const float* dy, const float* dy2, const float* gamma,
                                        const float* beta, float* dx, float* dgamma, float* dbeta, int n_inst, int inner,
                                        int64_t outer_stride, int64_t inner_stride, int64_t tok_stride, int ntok, int C,
                                        int groups, float eps, float slope, int leaky, void* scratch, size_t scratch_bytes,
                                        void* stream_) {
    SOLA_ARG(x && dy && gamma && beta && dx && dgamma && dbeta && scratch, "group_norm_backward: null argument");
    const size_t need = (size_t)2 * n_inst * C * sizeof(float);
    if (scratch_bytes < need) {
        sola_set_error("group_norm_backward: scratch %zu < %zu", scratch_bytes, need);
        return SOLA_ERR_WORKSPACE;
    }
    float* gp = static_cast<float*>(scratch);
    float* bp = gp + (size_t)n_inst * C;
    GroupNormBwdDesc d{x, dy, dy2, gamma, beta, dx, gp, bp, n_inst, inner, outer_stride, inner_stride, tok_stride, ntok, C, groups, eps, slope, leaky};
    d.drop = g_stage_drop;
    hipStream_t s = as_stream(stream_);
    SOLA_TRY(launch_group_norm_bwd(d, s));
    SOLA_TRY(launch_colsum(gp, dgamma, 1, n_inst, C, C, 1.f, 0, nullptr, 0, s));
    return launch_colsum(bp, dbeta, 1, n_inst, C, C, 1.f, 0, nullptr, 0, s);
}

extern "C" int sola_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                              int G, int H, int head_dim, int Sq, int Sk, int inner, int64_t q_outer, int64_t q_inner,
                              int64_t q_rs, int64_t k_outer, int64_t k_inner, int64_t k_rs, float scale, float* lse,
                              void* stream_) {
    SOLA_ARG(q && k && v && o, "attention: null argument");
    AttnDesc d{q, k, v, o, ldq, ldk, ldv, ldo, G, H, head_dim, Sq, Sk, inner, q_outer, q_inner, q_rs, k_outer, k_inner, k_rs, scale, lse};
    d.drop = g_stage_drop;
    d.split_math = g_stage_split_math;  // sola_tune "attn_stage_split_math" (tests): the split precision mode's arithmetic
    return launch_attention(d, as_stream(stream_));
}

// q, k, v given as split-f16 rows (sola_cast_sp16 or a split GEMM with c_is_split): the split-f16 MFMA shape of the kernel
extern "C" int sola_attention_split(const float* q_sp, int ldq, const float* k_sp, int ldk, const float* v_sp, int ldv, float* o,
                                    int ldo, int o_is_split, int G, int H, int head_dim, int Sq, int Sk, int inner,
                                    int64_t q_outer, int64_t q_inner, int64_t q_rs, int64_t k_outer, int64_t k_inner, int64_t k_rs,
                                    float scale, float* lse, void* stream_) {
    SOLA_ARG(q_sp && k_sp && v_sp && o, "attention_split: null argument");
    AttnDesc d{q_sp, k_sp, v_sp, o, ldq, ldk, ldv, ldo, G, H, head_dim, Sq, Sk, inner, q_outer, q_inner, q_rs, k_outer, k_inner, k_rs, scale, lse};
    d.drop = g_stage_drop;
    d.o_sp16 = o_is_split;
    d.in_sp16 = 1;
    return launch_attention(d, as_stream(stream_));
}

// dq/dk/dv share the pitches of q/k/v; dvec scratch: (q rows) * H floats
extern "C" int sola_attention_backward(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                       const float* o, const float* dout, int ldo, const float* lse, float* dq, float* dk,
                                       float* dv, float* dvec, int G, int H, int head_dim, int Sq, int Sk, int inner,
                                       int64_t q_outer, int64_t q_inner, int64_t q_rs, int64_t k_outer, int64_t k_inner,
                                       int64_t k_rs, float scale, void* stream_) {
    SOLA_ARG(q && k && v && o && dout && lse && dq && dk && dv && dvec, "attention_backward: null argument");
    AttnBwdDesc d{q, k, v, o, dout, lse, dq, dk, dv, dvec, ldq, ldk, ldv, ldo, ldq, ldk, ldv, G, H, head_dim, Sq, Sk, inner,
                  q_outer, q_inner, q_rs, k_outer, k_inner, k_rs, scale};
    d.drop = g_stage_drop;
    return launch_attention_bwd(d, as_stream(stream_));
}

extern "C" size_t sola_attention_backward_scratch_floats(int64_t q_rows, int G, int H, int Sk) {
    return attention_bwd_part_floats(q_rows, G, H, Sk);
}
extern "C" int sola_attention_backward_ws(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                          const float* dout, int ldo, const float* lse, float* dq, float* dk, float* dv, float* dvec, int G,
                                          int H, int head_dim, int Sq, int Sk, int inner, int64_t q_outer, int64_t q_inner, int64_t q_rs,
                                          int64_t k_outer, int64_t k_inner, int64_t k_rs, float scale, int64_t q_rows, float* scratch,
                                          size_t scratch_floats, void* stream_) {
    SOLA_ARG(q && k && v && o && dout && lse && dq && dk && dv && dvec, "attention_backward_ws: null argument");
    AttnBwdDesc d{q, k, v, o, dout, lse, dq, dk, dv, dvec, ldq, ldk, ldv, ldo, ldq, ldk, ldv, G, H, head_dim, Sq, Sk, inner,
                  q_outer, q_inner, q_rs, k_outer, k_inner, k_rs, scale};
    d.drop = g_stage_drop;
    d.part = scratch; d.part_floats = scratch_floats; d.part_rows = q_rows;
    return launch_attention_bwd(d, as_stream(stream_));
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

extern "C" int sola_mask_bilinear_pack(const void* masks, int elem_type, int n, int h, int w, int H, int W, uint32_t* bits,
                                       int64_t* area, void* stream_) {
    SOLA_ARG(masks && bits && area, "mask_bilinear_pack: null argument");
    return launch_mask_bilinear_pack(masks, elem_type, n, h, w, H, W, bits, reinterpret_cast<long long*>(area), as_stream(stream_));
}

extern "C" int sola_mask_unpack(const uint32_t* bits, int n, int H, int W, void* out, int elem_type, void* stream_) {
    SOLA_ARG(bits && out, "mask_unpack: null argument");
    return launch_mask_unpack(bits, n, H, W, out, elem_type, as_stream(stream_));
}

extern "C" int sola_rle_fill_or(const uint32_t* cum, const int64_t* off, int n_frames, int K, int h, int w, uint8_t* out,
                                uint32_t* bits, int64_t* area, void* stream_) {
    SOLA_ARG(off, "rle_fill_or: null offsets");
    return launch_rle_fill_or(cum, reinterpret_cast<const long long*>(off), n_frames, K, h, w, out, bits,
                              reinterpret_cast<long long*>(area), as_stream(stream_));
}

// Host helper (no GPU work): COCO compressed run-length string -> inclusive prefix sums of the run lengths, the form
// sola_rle_fill_or consumes.  pycocotools rleFrString: 5 data bits + continuation bit per char (offset 48), sign
// extension from bit 4 of the last char, runs from the 4th on stored as a delta to the run two places back.
extern "C" int64_t sola_rle_string_to_cum(const char* str, int64_t len, uint32_t* cum, int64_t cap, int64_t limit) {
    if (!str || !cum || len < 0 || cap < 0) {
        sola_set_error("rle_string_to_cum: bad arguments");
        return SOLA_ERR_ARG;
    }
    int64_t n = 0, p = 0;
    long long prev1 = 0, prev2 = 0;  // counts[n-1], counts[n-2]
    unsigned long long sum = 0;
    while (p < len) {
        long long x = 0;
        int k = 0;
        bool more = true;
        while (more) {
            if (p >= len || k > 12) {
                sola_set_error("rle_string_to_cum: truncated or over-long run at char %lld", (long long)p);
                return SOLA_ERR_ARG;
            }
            const long long c = (long long)(unsigned char)str[p] - 48;
            x |= (c & 0x1f) << (5 * k);
            more = (c & 0x20) != 0;
            ++p;
            ++k;
            if (!more && (c & 0x10)) x |= -1ll << (5 * k);
        }
        if (n > 2) x += prev2;
        if (x < 0 || n >= cap) {
            sola_set_error(x < 0 ? "rle_string_to_cum: negative run length" : "rle_string_to_cum: more runs than the output holds");
            return SOLA_ERR_ARG;
        }
        sum += (unsigned long long)x;
        if (limit >= 0 && sum > (unsigned long long)limit) {
            sola_set_error("rle_string_to_cum: runs cover %llu pixels, image has %lld", sum, (long long)limit);
            return SOLA_ERR_ARG;
        }
        cum[n++] = (uint32_t)sum;
        prev2 = prev1;
        prev1 = x;
    }
    return n;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
bool launch_mask_iou_fused(const void* am, const void* bm, int elem_type, int P, int R, int H, int W, int h, int w, long long* inter,
                           long long* uni, void* scratch, size_t scratch_bytes, hipStream_t s, int* status);

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
    {   // P <= 4 uint8 masks at the comparison resolution (the de-dup loop's calls): one memset + one kernel (iou.hip)
        int st = SOLA_OK;
        if (launch_mask_iou_fused(a, b, elem_type, P, R, H, W, h, w, reinterpret_cast<long long*>(inter), reinterpret_cast<long long*>(uni),
                                  scratch, scratch_bytes, as_stream(stream_), &st))
            return st;
    }
    const size_t words = (size_t)sola_mask_words(H, W);
    char* base = static_cast<char*>(scratch);
    uint32_t* abits = reinterpret_cast<uint32_t*>(base);
    uint32_t* bbits = reinterpret_cast<uint32_t*>(base + align256((size_t)P * words * 4));
    long long* aarea = reinterpret_cast<long long*>(base + align256((size_t)P * words * 4) + align256((size_t)R * words * 4));
    long long* barea = reinterpret_cast<long long*>(reinterpret_cast<char*>(aarea) + align256((size_t)P * 8));
    hipStream_t s = as_stream(stream_);
    int st = SOLA_OK;
    if (elem_type == 0 && h == H && w == W &&
        launch_mask_pack_pair(a, P, abits, aarea, b, R, bbits, barea, align256((size_t)P * 8) + (size_t)R * 8, H, W, s, &st)) {
        SOLA_TRY(st);  // one memset + one pack launch for both sets
    } else {
        SOLA_TRY(launch_mask_pack(a, elem_type, P, H, W, H, W, abits, aarea, s));
        SOLA_TRY(launch_mask_pack(b, elem_type, R, h, w, H, W, bbits, barea, s));
    }
    return launch_mask_pair(abits, aarea, P, 1, bbits, barea, R, nullptr, (long long)words,
                            reinterpret_cast<long long*>(inter), reinterpret_cast<long long*>(uni), s);
}
