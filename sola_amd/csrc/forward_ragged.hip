// Ragged inference forward (sola_forward_ragged): many (video, expression) samples of DIFFERENT shapes in one pass.
//
// The reference scores one sample per call (configs/mevis/default.yaml:37,42,47 batch_size 1; inference.py:44-58,
// evaluator.py:88-112): per-sample N tracks, T frames, L text tokens.  At one sample per launch the GPU runs 0.5-1.3 ms
// per call at a few percent of its rate; here the token rows of all samples are concatenated, the dense contractions run
// as ONE GEMM over all rows, and everything whose extent depends on the sample (conv windows along T, GroupNorm statistics,
// the three attentions, the score head) takes per-unit descriptors built on the device from the (N, T, L) arrays - no
// padding, so no statistic or softmax ever sees a token that is not the sample's own.
//
// Two levels:
//   videos  (object sets): N_v tracks x T_v frames of object tokens;
//   samples (video, expression): index of a video + L_i text tokens.
// Everything that does not depend on the text is computed once per VIDEO and shared by its expressions
// (inference.py:44-58 re-runs it per expression): the motion encoder and layer 0's inter-object and motion sub-blocks,
// 9.3 of the 16.4 GFLOP per sample at the headline shape.  The text enters at layer 0's object->language attention
// (module/module.py:46-50); from there on the rows are per sample (a row gather repeats the video's activations).
#include <math.h>
#include <string.h>

#include <algorithm>

#include "ragged.h"

int launch_attention_f16(const AttnDesc& d, hipStream_t s);

extern int g_attn_split_min_keys;  // forward_fast.hip

namespace {

// ---- device-side plan: unit tables from the compact per-video / per-sample descriptors ---------------------------
struct RagDev {
    // uploaded descriptor arrays (ints)
    const int *vN, *vT /* [7][V] */, *vRow0 /* [7][V+1] */, *vTrk0 /* [V+1] */, *vTp0 /* [V+1] */;
    const int *sVid, *sL, *sLin0, *sLrow0, *sTrk0, *sRow0, *sTp0;  // [S] / [S+1]
    int V, S, NT /* video tracks */, n_neg;
    int stride[5], pad[5], k[5];
    // tables to build
    int4* u_lvl[7];   // [NT] per level: (first row, 1, T_level, video)
    int2* rowmap[5];  // [rows of level l+1]
    int4* u_vt;       // [sum T'_v]  (first row, T'_v, N_v, t')
    int4* u_st;       // [sum T'_i]
    int4* u_strk;     // [sum N_i]   (first row, 1, T', sample)
    int4* u_smp;      // [S]         (first row, 1, N*T', sample)
    int4* u_lang;     // [S]         (first input text row, L, first lang_cat row, W)
    int4* u_langk;    // [S]         (first lang_cat row, 1, W, 0)
    int4* u_gather;   // [S]         (first sample row, first video row at T', rows, 0)
    int4* imap[5];    // training: [rows of level l + 1] (first output row of the sequence under conv l + 1, T_out, ti, 0); null = not built
};

// largest i in [0, n) with pre[i] <= x (pre ascending, pre[0] = 0)
__device__ __forceinline__ int seg_of(const int* pre, int n, int x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pre[mid] <= x) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// blockIdx.y = section: 0..6 level tables, 7..11 row maps, 12 (video, t') units, 13 (sample, t') units, 14 sample tracks, 15 samples,
// 16..20 (training) input-row maps of convs 1..5
__global__ __launch_bounds__(256) void ragged_plan_kernel(const RagDev p) {
    const int sec = blockIdx.y;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x;; i += (long long)gridDim.x * 256) {
        if (sec < 7) {
            if (i >= p.NT) return;
            const int v = seg_of(p.vTrk0, p.V, (int)i);
            const int T = p.vT[sec * p.V + v];
            p.u_lvl[sec][i] = make_int4(p.vRow0[sec * (p.V + 1) + v] + ((int)i - p.vTrk0[v]) * T, 1, T, v);
        } else if (sec < 12) {
            const int l = sec - 7;  // conv l: level l -> level l + 1
            const int* pre = p.vRow0 + (l + 1) * (p.V + 1);
            if (i >= pre[p.V]) return;
            const int v = seg_of(pre, p.V, (int)i);
            const int local = (int)i - pre[v];
            const int T_out = p.vT[(l + 1) * p.V + v], T_in = p.vT[l * p.V + v];
            const int n = local / T_out, to = local - n * T_out;
            const int t0 = to * p.stride[l] - p.pad[l];
            int bits = 0;
            for (int kk = 0; kk < p.k[l]; ++kk) bits |= ((unsigned)(t0 + kk) < (unsigned)T_in) ? (1 << kk) : 0;
            p.rowmap[l][i] = make_int2(p.vRow0[l * (p.V + 1) + v] + n * T_in + t0, bits);
        } else if (sec == 12) {
            if (i >= p.vTp0[p.V]) return;
            const int v = seg_of(p.vTp0, p.V, (int)i);
            const int t = (int)i - p.vTp0[v];
            p.u_vt[i] = make_int4(p.vRow0[6 * (p.V + 1) + v] + t, p.vT[6 * p.V + v], p.vN[v], t);
        } else if (sec == 13) {
            if (i >= p.sTp0[p.S]) return;
            const int sidx = seg_of(p.sTp0, p.S, (int)i);
            const int t = (int)i - p.sTp0[sidx];
            const int v = p.sVid[sidx];
            p.u_st[i] = make_int4(p.sRow0[sidx] + t, p.vT[6 * p.V + v], p.vN[v], t);
        } else if (sec == 14) {
            if (i >= p.sTrk0[p.S]) return;
            const int sidx = seg_of(p.sTrk0, p.S, (int)i);
            const int v = p.sVid[sidx];
            const int Tp = p.vT[6 * p.V + v];
            p.u_strk[i] = make_int4(p.sRow0[sidx] + ((int)i - p.sTrk0[sidx]) * Tp, 1, Tp, sidx);
        } else if (sec >= 16) {
            const int l = sec - 16 + 1;  // conv l reads level l and writes level l + 1: this is the table of its INPUT rows
            if (!p.imap[l - 1]) return;
            const int* pre = p.vRow0 + l * (p.V + 1);
            if (i >= pre[p.V]) return;
            const int v = seg_of(pre, p.V, (int)i);
            const int local = (int)i - pre[v];
            const int T_in = p.vT[l * p.V + v], T_out = p.vT[(l + 1) * p.V + v];
            const int n = local / T_in, ti = local - n * T_in;
            p.imap[l - 1][i] = make_int4(p.vRow0[(l + 1) * (p.V + 1) + v] + n * T_out, T_out, ti, 0);
        } else {
            if (i >= p.S) return;
            const int v = p.sVid[i];
            const int rows = p.vN[v] * p.vT[6 * p.V + v];
            const int W = p.sL[i] + p.n_neg;
            p.u_smp[i] = make_int4(p.sRow0[i], 1, rows, (int)i);
            p.u_lang[i] = make_int4(p.sLin0[i], p.sL[i], p.sLrow0[i], W);
            p.u_langk[i] = make_int4(p.sLrow0[i], 1, W, 0);
            p.u_gather[i] = make_int4(p.sRow0[i], p.vRow0[6 * (p.V + 1) + v], rows, 0);
        }
    }
}

}  // namespace

// ---- host-side shape bookkeeping ---------------------------------------------------------------------------------
int rag_shape(const SolaCtx* c, const SolaRaggedBatch* b, RagShape& r) {
    SOLA_ARG(b && b->n_videos > 0 && b->n_samples > 0 && b->video_tracks && b->video_frames && b->sample_video && b->sample_text_len,
             "ragged batch: null or empty descriptor");
    r.V = b->n_videos; r.S = b->n_samples;
    for (int j = 0; j < 7; ++j) { r.vT[j].resize(r.V); r.vRow0[j].assign(r.V + 1, 0); }
    r.vN.resize(r.V); r.vTrk0.assign(r.V + 1, 0); r.vTp0.assign(r.V + 1, 0);
    for (int v = 0; v < r.V; ++v) {
        const int N = b->video_tracks[v], T = b->video_frames[v];
        SOLA_ARG(N >= 1 && T >= 1, "ragged batch: video %d has N=%d T=%d", v, N, T);
        r.vN[v] = N;
        int t = T;
        r.vT[0][v] = t;
        for (int i = 0; i < 6; ++i) {
            t = (t + 2 * c->conv[i].pad - c->conv[i].k) / c->conv[i].stride + 1;
            SOLA_ARG(t >= 1, "ragged batch: video %d (T=%d) is shorter than the encoder's receptive field", v, T);
            r.vT[i + 1][v] = t;
        }
        for (int j = 0; j < 7; ++j) {
            const long long next = (long long)r.vRow0[j][v] + (long long)N * r.vT[j][v];
            SOLA_ARG(next < (1ll << 31), "ragged batch: more than 2^31 token rows");
            r.vRow0[j][v + 1] = (int)next;
            r.maxT[j] = std::max(r.maxT[j], r.vT[j][v]);
        }
        r.vTrk0[v + 1] = r.vTrk0[v] + N;
        r.vTp0[v + 1] = r.vTp0[v] + r.vT[6][v];
        r.maxN = std::max(r.maxN, N);
    }
    for (int j = 0; j < 7; ++j) r.rows[j] = r.vRow0[j][r.V];
    r.NT = r.vTrk0[r.V]; r.Mv = r.rows[6]; r.sumTpV = r.vTp0[r.V];
    r.sVid.resize(r.S); r.sL.resize(r.S);
    r.sLin0.assign(r.S + 1, 0); r.sLrow0.assign(r.S + 1, 0); r.sTrk0.assign(r.S + 1, 0); r.sRow0.assign(r.S + 1, 0); r.sTp0.assign(r.S + 1, 0);
    r.identity = r.S == r.V;
    for (int i = 0; i < r.S; ++i) {
        const int v = b->sample_video[i], L = b->sample_text_len[i];
        SOLA_ARG(v >= 0 && v < r.V && L >= 1, "ragged batch: sample %d has video=%d L=%d", i, v, L);
        r.sVid[i] = v; r.sL[i] = L;
        const int W = L + c->cfg.n_negative;
        const int rows = r.vN[v] * r.vT[6][v];
        const long long next = (long long)r.sRow0[i] + rows;
        SOLA_ARG(next < (1ll << 31), "ragged batch: more than 2^31 token rows");
        r.sLin0[i + 1] = r.sLin0[i] + L;
        r.sLrow0[i + 1] = r.sLrow0[i] + W;
        r.sTrk0[i + 1] = r.sTrk0[i] + r.vN[v];
        r.sRow0[i + 1] = (int)next;
        r.sTp0[i + 1] = r.sTp0[i] + r.vT[6][v];
        r.maxW = std::max(r.maxW, W);
        r.maxRowsSample = std::max(r.maxRowsSample, rows);
        if (v != i) r.identity = false;
    }
    r.Ms = r.sRow0[r.S]; r.LW = r.sLrow0[r.S]; r.Lin = r.sLin0[r.S];
    r.sumTpS = r.sTp0[r.S]; r.sumNS = r.sTrk0[r.S];
    return SOLA_OK;
}

namespace {

size_t blob_ints(const RagShape& r) { return (size_t)r.V * 8 + (size_t)(r.V + 1) * 9 + (size_t)r.S * 2 + (size_t)(r.S + 1) * 5; }

struct RagPlan {
    std::unordered_map<std::string, size_t> off;
    size_t total = 0;
    size_t add(const std::string& name, size_t bytes) {
        const size_t o = total;
        off[name] = o;
        total += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

}  // namespace

// slots of the sliced GroupNorm shape: a launch has (instances x 8 groups x slices of the LONGEST unit) blocks of 8 bytes; the
// object->language norm has one instance per sample (up to maxRowsSample tokens), the encoder norms one per track (up to
// maxT[1] tokens); slices are at least 128 tokens
size_t rag_gn_slots_bytes(const RagShape& r) {
    const size_t a = (size_t)r.S * ((size_t)r.maxRowsSample / 128 + 1);
    const size_t b = (size_t)r.NT * ((size_t)r.maxT[1] / 128 + 1);
    const size_t t = (size_t)std::max(r.sumTpV, r.sumTpS) * ((size_t)r.maxN / 128 + 1);  // inter-object norm: one instance per (sample, t')
    return 8 * 8 * std::max(a, std::max(b, t)) + 4096;
}

namespace {

RagPlan rag_plan(const SolaCtx* c, const RagShape& r, int precision) {  // `precision`: the arithmetic the plan is sized for (the ctx's, or 0 for the guard's exact-f32 repeat)
    RagPlan p;
    const size_t D = c->cfg.lang_token_dim, f = sizeof(float);
    const bool sp = precision >= 1;  // split-f16 or plain-f16 copies of the caller's tensors
    p.add("tables", rag_tables_bytes(r, false));
    for (int i = 0; i < 6; ++i) {
        p.add("conv" + std::to_string(i), (size_t)r.rows[i + 1] * c->conv[i].cout * f);
        if (i < 5) p.add("act" + std::to_string(i), (size_t)r.rows[i + 1] * c->conv[i].cout * f);
    }
    if (sp) p.add("obj_sp", (size_t)r.rows[0] * c->cfg.object_token_dim * f);
    const size_t Mmax = (size_t)std::max(r.Mv, r.Ms);
    if (Mmax <= 8192) p.add("splitk", (size_t)8192 * 4096 * f);
    p.add("pe", (size_t)r.maxT[6] * D * f);
    p.add("gn_slots", rag_gn_slots_bytes(r));  // sliced GroupNorm shape: 8 B per (unit, slice)
    p.add("lang", (size_t)r.LW * D * f);
    if (sp) p.add("lang_sp", (size_t)r.LW * D * f);
    p.add("lbar", (size_t)r.S * D * f);
    p.add("lk", (size_t)r.LW * D * f);
    p.add("lv", (size_t)r.LW * D * f);
    for (const char* nm : {"q", "k", "v", "attn", "res"}) p.add(nm, Mmax * D * f);
    p.add("v_obj", (size_t)r.Mv * D * f);
    p.add("v_xpe", (size_t)r.Mv * D * f);
    p.add("v_motion", (size_t)r.Mv * D * f);
    if (!r.identity) p.add("s_motion0", (size_t)r.Ms * D * f);
    p.add("s0_o2l", (size_t)r.Ms * D * f);
    for (int l = 1; l < c->cfg.n_layers; ++l) {
        const std::string ls = "s" + std::to_string(l);
        p.add(ls + "_obj", (size_t)r.Ms * D * f);
        p.add(ls + "_xpe", (size_t)r.Ms * D * f);
        p.add(ls + "_motion", (size_t)r.Ms * D * f);
        p.add(ls + "_o2l", (size_t)r.Ms * D * f);
    }
    return p;
}

// pinned staging for the descriptor upload: a small ring, a slot is reused only after the copy that read it has completed
struct RagStage {
    static constexpr int SLOTS = 4;
    int* host[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    bool used[SLOTS] = {false, false, false, false};
    size_t cap = 0;
    int next = 0;
};

}  // namespace

struct SolaRagStage : RagStage {};

void sola_rag_stage_free(SolaRagStage* st) {
    if (!st) return;
    for (int i = 0; i < RagStage::SLOTS; ++i) {
        if (st->host[i]) (void)hipHostFree(st->host[i]);
        if (st->ev[i]) (void)hipEventDestroy(st->ev[i]);
    }
    delete st;
}

static int stage_slot(SolaCtx* c, size_t ints, int** host, hipEvent_t* ev) {
    if (!c->rag_stage) c->rag_stage = new SolaRagStage();
    RagStage* st = c->rag_stage;
    if (ints > st->cap) {
        for (int i = 0; i < RagStage::SLOTS; ++i) {
            if (st->used[i]) SOLA_HIP(hipEventSynchronize(st->ev[i]));
            if (st->host[i]) SOLA_HIP(hipHostFree(st->host[i]));
            st->host[i] = nullptr;
            st->used[i] = false;
        }
        st->cap = std::max<size_t>(ints * 2, 16384);
        for (int i = 0; i < RagStage::SLOTS; ++i) SOLA_HIP(hipHostMalloc(reinterpret_cast<void**>(&st->host[i]), st->cap * sizeof(int), hipHostMallocDefault));
    }
    const int i = st->next;
    st->next = (st->next + 1) % RagStage::SLOTS;
    if (!st->ev[i]) SOLA_HIP(hipEventCreateWithFlags(&st->ev[i], hipEventDisableTiming));
    if (st->used[i]) SOLA_HIP(hipEventSynchronize(st->ev[i]));
    st->used[i] = true;
    *host = st->host[i];
    *ev = st->ev[i];
    return SOLA_OK;
}

namespace {
size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
// walks the table region in a fixed order; `visit(name index, bytes)` returns nothing, offsets accumulate
struct TableLayout {
    size_t blob, u_lvl[7], rowmap[5], imap[5], u_vt, u_st, u_strk, u_smp, u_lang, u_langk, u_gather, total;
};
TableLayout table_layout(const RagShape& r, bool train) {
    TableLayout t{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
    t.blob = take(blob_ints(r) * sizeof(int));
    for (int j = 0; j < 7; ++j) t.u_lvl[j] = take((size_t)r.NT * sizeof(int4));
    for (int l = 0; l < 5; ++l) t.rowmap[l] = take((size_t)r.rows[l + 1] * sizeof(int2));
    for (int l = 0; l < 5; ++l) t.imap[l] = train ? take((size_t)r.rows[l + 1] * sizeof(int4)) : 0;
    t.u_vt = take((size_t)r.sumTpV * sizeof(int4));
    t.u_st = take((size_t)r.sumTpS * sizeof(int4));
    t.u_strk = take((size_t)r.sumNS * sizeof(int4));
    t.u_smp = take((size_t)r.S * sizeof(int4));
    t.u_lang = take((size_t)r.S * sizeof(int4));
    t.u_langk = take((size_t)r.S * sizeof(int4));
    t.u_gather = take((size_t)r.S * sizeof(int4));
    t.total = o;
    return t;
}
}  // namespace

size_t rag_tables_bytes(const RagShape& r, bool train) { return table_layout(r, train).total; }

int rag_build_tables(SolaCtx* c, const RagShape& r, char* base, bool train, RagTables* out, hipStream_t s) {
    SOLA_ARG(c && base && out && (reinterpret_cast<uintptr_t>(base) & 255) == 0, "ragged tables: bad arguments");
    const TableLayout t = table_layout(r, train);
    int* host;
    hipEvent_t ev;
    SOLA_TRY(stage_slot(c, blob_ints(r), &host, &ev));
    int* w = host;
    int* const blob = reinterpret_cast<int*>(base + t.blob);
    auto put = [&](const std::vector<int>& v) { const int* at = blob + (w - host); memcpy(w, v.data(), v.size() * sizeof(int)); w += v.size(); return at; };
    RagDev dv{};
    dv.vN = put(r.vN);
    dv.vT = blob + (w - host);
    for (int j = 0; j < 7; ++j) put(r.vT[j]);
    dv.vRow0 = blob + (w - host);
    for (int j = 0; j < 7; ++j) put(r.vRow0[j]);
    dv.vTrk0 = put(r.vTrk0); dv.vTp0 = put(r.vTp0);
    dv.sVid = put(r.sVid); dv.sL = put(r.sL); dv.sLin0 = put(r.sLin0); dv.sLrow0 = put(r.sLrow0);
    dv.sTrk0 = put(r.sTrk0); dv.sRow0 = put(r.sRow0); dv.sTp0 = put(r.sTp0);
    SOLA_HIP(hipMemcpyAsync(blob, host, (size_t)(w - host) * sizeof(int), hipMemcpyHostToDevice, s));
    SOLA_HIP(hipEventRecord(ev, s));
    dv.V = r.V; dv.S = r.S; dv.NT = r.NT; dv.n_neg = c->cfg.n_negative;
    for (int l = 0; l < 5; ++l) { dv.stride[l] = c->conv[l].stride; dv.pad[l] = c->conv[l].pad; dv.k[l] = c->conv[l].k; }
    auto t4 = [&](size_t off) { return reinterpret_cast<int4*>(base + off); };
    for (int j = 0; j < 7; ++j) dv.u_lvl[j] = t4(t.u_lvl[j]);
    for (int l = 0; l < 5; ++l) dv.rowmap[l] = reinterpret_cast<int2*>(base + t.rowmap[l]);
    for (int l = 0; l < 5; ++l) dv.imap[l] = train ? t4(t.imap[l]) : nullptr;
    dv.u_vt = t4(t.u_vt); dv.u_st = t4(t.u_st); dv.u_strk = t4(t.u_strk); dv.u_smp = t4(t.u_smp);
    dv.u_lang = t4(t.u_lang); dv.u_langk = t4(t.u_langk); dv.u_gather = t4(t.u_gather);
    long long biggest = std::max<long long>(r.NT, r.rows[1]);
    biggest = std::max<long long>(biggest, std::max<long long>(r.sumNS, std::max(r.sumTpS, r.sumTpV)));
    const unsigned bx = (unsigned)std::min<long long>(2048, (biggest + 255) / 256);
    {
        SolaProfScope prof(SOLA_PROF_MISC, s, 0, 0);
        hipLaunchKernelGGL(ragged_plan_kernel, dim3(bx, train ? 21 : 16), dim3(256), 0, s, dv);
        SOLA_LAUNCH_CHECK();
    }
    RagTables& o = *out;
    for (int j = 0; j < 7; ++j) o.u_lvl[j] = dv.u_lvl[j];
    for (int l = 0; l < 5; ++l) { o.rowmap[l] = dv.rowmap[l]; o.imap[l] = dv.imap[l]; }
    o.u_vt = dv.u_vt; o.u_st = dv.u_st; o.u_strk = dv.u_strk; o.u_smp = dv.u_smp;
    o.u_lang = dv.u_lang; o.u_langk = dv.u_langk; o.u_gather = dv.u_gather;
    o.trk_off = dv.sTrk0;
    o.V = r.V; o.S = r.S; o.NT = r.NT; o.sumNS = r.sumNS; o.sumTpV = r.sumTpV; o.sumTpS = r.sumTpS;
    o.maxN = r.maxN; o.maxW = r.maxW; o.maxRowsSample = r.maxRowsSample;
    for (int j = 0; j < 7; ++j) { o.maxT[j] = r.maxT[j]; o.rows[j] = r.rows[j]; }
    o.Mv = r.Mv; o.Ms = r.Ms; o.LW = r.LW; o.Lin = r.Lin; o.identity = r.identity;
    return SOLA_OK;
}

size_t sola_ragged_workspace_bytes_impl(const SolaCtx* c, const SolaRaggedBatch* b, int precision) {
    RagShape r;
    if (!c || rag_shape(c, b, r) != SOLA_OK) return 0;
    return rag_plan(c, r, precision).total;
}

int sola_forward_ragged_impl(SolaCtx* c, const float* obj, const float* lang, const SolaRaggedBatch* batch, float* score_map,
                             float* score_tokens, void* workspace, size_t ws_bytes, hipStream_t s) {
    SOLA_ARG(c && obj && lang && batch && score_map && score_tokens && workspace, "forward_ragged: null argument");
    for (const Weight& w : c->weights)
        if (!w.ptr) {
            sola_set_error("forward_ragged: weight '%s' has not been set", w.name.c_str());
            return SOLA_ERR_WEIGHT;
        }
    const bool sp = c->precision == 1;
    // precision 2: 16-bit activation STORAGE (forward_f16.hip's arithmetic on the ragged layout): every activation between two
    // kernels a plain _Float16 in the first half of its f32-sized buffer, one f16 MFMA per product, f32 statistics / softmax
    const bool h16 = c->precision == 2;
    if (h16)
        SOLA_ARG(c->cfg.object_token_dim % 64 == 0 && c->cfg.lang_token_dim % 64 == 0,
                 "16-bit storage mode needs object_token_dim and lang_token_dim to be multiples of 64");
    if (sp)
        SOLA_ARG(c->cfg.object_token_dim % 8 == 0 && (c->cfg.lang_token_dim / c->cfg.n_groups_module) % 8 == 0 &&
                     (2 * c->cfg.object_token_dim / c->cfg.n_groups) % 8 == 0 && (c->cfg.lang_token_dim / c->cfg.n_groups) % 8 == 0,
                 "split-f16 mode needs channel counts per GroupNorm group that are multiples of 8");
    RagShape r;
    SOLA_TRY(rag_shape(c, batch, r));
    const RagPlan p = rag_plan(c, r, c->precision);
    if (ws_bytes < p.total) {
        sola_set_error("forward_ragged: workspace %zu bytes < required %zu", ws_bytes, p.total);
        return SOLA_ERR_WORKSPACE;
    }
    SOLA_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "forward_ragged: workspace must be 256-byte aligned");
    char* base = static_cast<char*>(workspace);
    auto raw = [&](const std::string& name) { return base + p.off.at(name); };
    auto buf = [&](const std::string& name) { return reinterpret_cast<float*>(raw(name)); };
    auto W = [&](const std::string& name) { return ctx_weight(c, name); };
    const int D = c->cfg.lang_token_dim, H = c->cfg.num_heads, DH = D / H, d_in = c->cfg.object_token_dim;
    const int S = r.S;
    float* const splitk_ws = p.off.count("splitk") ? buf("splitk") : nullptr;
    const size_t splitk_bytes = splitk_ws ? (size_t)8192 * 4096 * sizeof(float) : 0;
    const size_t gn_slots_bytes = rag_gn_slots_bytes(r);

    // ---- descriptors -> device, unit tables
    RagTables rt;
    SOLA_TRY(rag_build_tables(c, r, raw("tables"), false, &rt, s));
    auto tab4 = [&](const std::string& name) -> const int4* {
        if (name == "u_vt") return rt.u_vt;
        if (name == "u_st") return rt.u_st;
        if (name == "u_strk") return rt.u_strk;
        if (name == "u_smp") return rt.u_smp;
        if (name == "u_lang") return rt.u_lang;
        if (name == "u_langk") return rt.u_langk;
        if (name == "u_gather") return rt.u_gather;
        return rt.u_lvl[name.back() - '0'];  // "u_lvl<j>"
    };

    // ---- weights
    if (c->ws_dirty || c->ws_every_forward || (sp && c->ws16_fmt != 1) || (h16 && c->ws16_fmt != 2)) {
        WsLayer layers[6];
        for (int i = 0; i < 6; ++i) {
            const std::string nm = "short_motion_encoder." + std::to_string(kConvIdx[i]) + ".weight";
            layers[i] = WsLayer{W(nm), c->ws_buf + c->ws_off[i], c->conv[i].cout, c->conv[i].cin, c->conv[i].k};
        }
        SOLA_TRY(launch_ws_standardize(layers, 6, s));
        if (sp)
            for (int i = 0; i < 6; ++i) {
                const int kc = c->conv[i].k * c->conv[i].cin;
                SOLA_TRY(launch_cast_sp16(c->ws_buf + c->ws_off[i], kc, c->ws16_buf + c->ws_off[i], kc, c->conv[i].cout, kc, 1.f, s));
            }
        if (h16)
            for (int i = 0; i < 6; ++i) {
                const int kc = c->conv[i].k * c->conv[i].cin;
                SOLA_TRY(launch_cast_f16(c->ws_buf + c->ws_off[i], kc, reinterpret_cast<_Float16*>(c->ws16_buf) + c->ws_off[i], kc, c->conv[i].cout, kc, 1.f,
                                         nullptr, s));
            }
        c->ws_dirty = false;
        c->ws16_fmt = sp ? 1 : (h16 ? 2 : 0);
    }
    if (sp || h16) {
        SOLA_TRY(sola_refresh_lin16(c, s));
        SOLA_HIP(hipMemsetAsync(c->guard, 0, sizeof(int), s));
    }
    int* const guard = (sp || h16) ? c->guard : nullptr;
    auto ws_w = [&](int i) -> const float* {  // standardised conv weights in the mode's operand format
        if (h16) return reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(c->ws16_buf) + c->ws_off[i]);
        return sp ? c->ws16_buf + c->ws_off[i] : c->ws_buf + c->ws_off[i];
    };
    auto lin_w = [&](int layer, int attn, int proj) -> const float* {
        static const char* pn[4] = {"q_proj", "k_proj", "v_proj", "out_proj"};
        if (sp) return c->lin16_buf + ((size_t)(layer * 3 + attn) * 4 + proj) * D * D;
        if (h16) return reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(c->lin16_buf) + ((size_t)(layer * 3 + attn) * 4 + proj) * D * D);
        return W("object_lang_align_layers." + std::to_string(layer) + "." + kAttnLong[attn] + "." + pn[proj] + ".weight");
    };
    auto lin_b = [&](int layer, int attn, int proj) -> const float* {
        static const char* pn[4] = {"q_proj", "k_proj", "v_proj", "out_proj"};
        return W("object_lang_align_layers." + std::to_string(layer) + "." + kAttnLong[attn] + "." + pn[proj] + ".bias");
    };

    // ---- encoder over the videos' tracks (module/module.py:74-96,137-140)
    const float* x = obj;
    if (sp) {
        SOLA_TRY(launch_cast_sp16_auto(obj, d_in, buf("obj_sp"), d_in, r.rows[0], d_in, c->scal_pair(0), s));
        x = buf("obj_sp");
    }
    if (h16) {  // largest magnitude -> [2^6, 2^7), capped at 2^8; conv0's output stays in the scaled units (forward_f16.hip)
        SOLA_TRY(launch_cast_f16(obj, d_in, buf("obj_sp"), d_in, r.rows[0], d_in, 256.f, c->scal_pair(0), s, 6, c->scal_extra(0)));
        x = buf("obj_sp");
    }
    for (int i = 0; i < 6; ++i) {
        const ConvGeom& g = c->conv[i];
        const std::string cp = "short_motion_encoder." + std::to_string(kConvIdx[i]);
        const bool last_sp = sp && i == 5;  // conv5 feeds layer 0 as GEMM operand and residual: written as split-f16 pairs
        GemmDesc gd{};
        gd.nprob = 1;
        gd.p[0] = GemmProblem{x, ws_w(i), W(cp + ".bias"), nullptr, buf("conv" + std::to_string(i))};
        gd.M = (int)r.rows[i + 1]; gd.N = g.cout; gd.K = g.k * g.cin;
        gd.lda = g.cin; gd.ldr = 0; gd.ldc = g.cout;
        gd.conv = g.k > 1 ? 1 : 0;
        gd.Cin = g.cin; gd.stride = g.stride; gd.pad = g.pad; gd.T_in = 1; gd.T_out = 1;
        if (gd.conv) gd.rowmap = rt.rowmap[i];
        if (sp) {
            gd.arith = 1; gd.out_scale = 1.f; gd.c_sp16 = last_sp ? 1 : 0; gd.guard = guard;
            if (i == 0) gd.out_scale_dev = c->scal_pair(0) + 1;
        }
        if (h16) {
            gd.arith = 2; gd.out_scale = 1.f; gd.c_f16 = 1; gd.guard = guard;
            if (i == 0) gd.bias_scale_dev = c->scal_extra(0);
        }
        gd.splitk_ws = h16 ? nullptr : splitk_ws; gd.splitk_bytes = h16 ? 0 : splitk_bytes;
        SOLA_TRY(launch_gemm(gd, s));
        if (i < 5) {
            const std::string np = "short_motion_encoder." + std::to_string(kNormIdx[i]);
            GroupNormDesc nd{};
            nd.slice_ws = raw("gn_slots"); nd.slice_ws_bytes = gn_slots_bytes;
            nd.x = buf("conv" + std::to_string(i)); nd.y = buf("act" + std::to_string(i));
            nd.gamma = W(np + ".weight"); nd.beta = W(np + ".bias");
            nd.n_inst = r.NT; nd.inner = 1; nd.tok_stride = 1;
            nd.units = tab4("u_lvl" + std::to_string(i + 1));
            nd.ntok = r.maxT[i + 1]; nd.C = g.cout; nd.groups = c->cfg.n_groups; nd.eps = 1e-5f; nd.slope = 0.01f; nd.leaky = 1;
            nd.out_sp16 = sp ? 1 : 0; nd.guard = guard;
            if (h16) {
                nd.in_f16 = 1; nd.out_f16 = 1;
                if (i == 0) nd.in_scale_dev = c->scal_pair(0) + 1;
            }
            SOLA_TRY(launch_group_norm(nd, s));
            x = buf("act" + std::to_string(i));
        }
    }
    const int maxTp = r.maxT[6];
    SOLA_TRY(launch_pos_encoding(W("positional_encoding_gaussian_matrix"), D, maxTp, c->cfg.max_temporal_length, buf("pe"), s));
    SOLA_TRY(launch_lang_concat_ragged(lang, W("negative_token.weight"), buf("lang"), buf("lbar"), S, tab4("u_lang"), c->cfg.n_negative, D, s));
    const float* lang_in = buf("lang");
    if (sp) {
        SOLA_TRY(launch_cast_sp16_auto(buf("lang"), D, buf("lang_sp"), D, r.LW, D, c->scal_pair(1), s));
        lang_in = buf("lang_sp");
    }
    if (h16) {
        SOLA_TRY(launch_cast_f16(buf("lang"), D, buf("lang_sp"), D, r.LW, D, 1.f, c->scal_pair(1), s, 6));
        lang_in = buf("lang_sp");
    }

    const float scale = 1.0f / sqrtf((float)DH);
    auto linear3 = [&](const float* a0, const float* a1, const float* a2, int layer, int attn, int nprob, long long rows, float* o0, float* o1,
                       float* o2, int first_proj, int out_sp16, const float* a_inv_scale) -> int {
        const float* as[3] = {a0, a1, a2};
        float* os[3] = {o0, o1, o2};
        GemmDesc gd{};
        gd.nprob = nprob;
        for (int j = 0; j < nprob; ++j)
            gd.p[j] = GemmProblem{as[j], lin_w(layer, attn, first_proj + j), lin_b(layer, attn, first_proj + j), nullptr, os[j],
                                  (sp || h16) ? c->lin_inv_scale(layer, attn, first_proj + j) : nullptr};
        gd.M = (int)rows; gd.N = D; gd.K = D; gd.lda = D; gd.ldr = 0; gd.ldc = D;
        if (sp) { gd.arith = 1; gd.out_scale = 1.f; gd.out_scale_dev = a_inv_scale; gd.c_sp16 = out_sp16; gd.guard = guard; }
        if (h16) { gd.arith = 2; gd.out_scale = 1.f; gd.out_scale_dev = a_inv_scale; gd.c_f16 = 1; gd.guard = guard; }
        gd.splitk_ws = h16 ? nullptr : splitk_ws; gd.splitk_bytes = h16 ? 0 : splitk_bytes;
        return launch_gemm(gd, s);
    };
    auto out_proj = [&](int layer, int attn, long long rows, const float* resid, int resid_sp16) -> int {
        GemmDesc gd{};
        gd.nprob = 1;
        gd.p[0] = GemmProblem{buf("attn"), lin_w(layer, attn, 3), lin_b(layer, attn, 3), resid, buf("res"), (sp || h16) ? c->lin_inv_scale(layer, attn, 3) : nullptr};
        gd.M = (int)rows; gd.N = D; gd.K = D; gd.lda = D; gd.ldr = D; gd.ldc = D;
        if (sp) { gd.arith = 1; gd.out_scale = 1.f; gd.r_sp16 = resid_sp16; }
        if (h16) { gd.arith = 2; gd.out_scale = 1.f; gd.r_f16 = 1; gd.c_f16 = 1; gd.guard = guard; }
        gd.splitk_ws = h16 ? nullptr : splitk_ws; gd.splitk_bytes = h16 ? 0 : splitk_bytes;
        return launch_gemm(gd, s);
    };
    auto gn = [&](int layer, int idx, float* y, float* y2, int out_sp16, const int4* units, int n_inst, int max_tok, int out_f16 = 1) -> int {
        const std::string lp = "object_lang_align_layers." + std::to_string(layer) + ".norm." + std::to_string(idx);
        GroupNormDesc nd{};
            nd.slice_ws = raw("gn_slots"); nd.slice_ws_bytes = gn_slots_bytes;
        nd.x = buf("res"); nd.y = y; nd.y2 = y2; nd.pe = y2 ? buf("pe") : nullptr;
        nd.gamma = W(lp + ".weight"); nd.beta = W(lp + ".bias");
        nd.n_inst = n_inst; nd.inner = 1; nd.tok_stride = 1; nd.units = units; nd.ntok = max_tok;
        nd.C = D; nd.groups = c->cfg.n_groups_module; nd.eps = 1e-5f; nd.slope = 0.f; nd.leaky = 0;
        nd.out_sp16 = out_sp16; nd.guard = guard;
        if (h16) { nd.in_f16 = 1; nd.out_f16 = out_f16; }
        return launch_group_norm(nd, s);
    };
    auto attention = [&](const float* q, const float* k, const float* v, int G, const int4* qu, const int4* ku, int maxSq, int maxSk,
                         int in_sp16) -> int {
        AttnDesc ad{q, k, v, buf("attn"), D, D, D, D, G, H, DH, maxSq, maxSk, 1, 0, 0, 1, 0, 0, 1, scale, nullptr};
        ad.o_sp16 = sp ? 1 : 0;
        ad.in_sp16 = in_sp16;
        ad.guard = guard;
        ad.split_math = sp ? 1 : 0;
        ad.q_units = qu; ad.k_units = ku;
        if (h16) { ad.o_sp16 = 0; ad.in_sp16 = 0; ad.split_math = 0; return launch_attention_f16(ad, s); }
        return launch_attention(ad, s);
    };
    // q/k/v leave the projection already split where the attention runs the split-f16 MFMA shape (forward_fast.hip: units of
    // more than g_attn_split_min_keys keys; never the packed short-sequence shape)
    const int obj_in_sp = (sp && r.maxN > g_attn_split_min_keys && DH % 16 == 0) ? 1 : 0;
    const int mot_in_sp = (sp && maxTp > g_attn_split_min_keys && DH % 16 == 0) ? 1 : 0;
    const int o2l_in_sp = (sp && r.maxW > g_attn_split_min_keys && DH % 16 == 0) ? 1 : 0;
    const int spi = sp ? 1 : 0;
    float *q = buf("q"), *k = buf("k"), *v = buf("v");

    // ---- layer 0, text-independent half, once per VIDEO (module/module.py:31-43)
    {
        const float* xin = buf("conv5");
        const long long M = r.Mv;
        SOLA_TRY(linear3(xin, xin, xin, 0, 0, 3, M, q, k, v, 0, obj_in_sp, nullptr));
        SOLA_TRY(attention(q, k, v, r.sumTpV, tab4("u_vt"), nullptr, r.maxN, r.maxN, obj_in_sp));
        SOLA_TRY(out_proj(0, 0, M, xin, spi));
        SOLA_TRY(gn(0, 0, buf("v_obj"), buf("v_xpe"), spi, tab4("u_vt"), r.sumTpV, r.maxN));
        SOLA_TRY(linear3(buf("v_xpe"), buf("v_xpe"), buf("v_obj"), 0, 1, 3, M, q, k, v, 0, mot_in_sp, nullptr));
        SOLA_TRY(attention(q, k, v, r.NT, tab4("u_lvl6"), nullptr, maxTp, maxTp, mot_in_sp));
        SOLA_TRY(out_proj(0, 1, M, buf("v_obj"), spi));
        SOLA_TRY(gn(0, 1, buf("v_motion"), nullptr, spi, tab4("u_lvl6"), r.NT, maxTp));
    }
    // ---- from here on rows are per SAMPLE: repeat the video's activations for each of its expressions
    const float* x_mot = buf("v_motion");
    if (!r.identity) {
        SOLA_TRY(launch_gather_rows(buf("v_motion"), buf("s_motion0"), tab4("u_gather"), S, h16 ? D / 2 : D, r.Ms, s));  // f16 rows: D halfs
        x_mot = buf("s_motion0");
    }
    const long long M = r.Ms;
    const float* xin = nullptr;
    for (int l = 0; l < c->cfg.n_layers; ++l) {
        const std::string ls = "s" + std::to_string(l);
        const bool last = l + 1 == c->cfg.n_layers;
        if (l > 0) {
            float* x_obj = buf(ls + "_obj");
            float* x_pe = buf(ls + "_xpe");
            SOLA_TRY(linear3(xin, xin, xin, l, 0, 3, M, q, k, v, 0, obj_in_sp, nullptr));
            SOLA_TRY(attention(q, k, v, r.sumTpS, tab4("u_st"), nullptr, r.maxN, r.maxN, obj_in_sp));
            SOLA_TRY(out_proj(l, 0, M, xin, spi));
            SOLA_TRY(gn(l, 0, x_obj, x_pe, spi, tab4("u_st"), r.sumTpS, r.maxN));
            SOLA_TRY(linear3(x_pe, x_pe, x_obj, l, 1, 3, M, q, k, v, 0, mot_in_sp, nullptr));
            SOLA_TRY(attention(q, k, v, r.sumNS, tab4("u_strk"), nullptr, maxTp, maxTp, mot_in_sp));
            SOLA_TRY(out_proj(l, 1, M, x_obj, spi));
            SOLA_TRY(gn(l, 1, buf(ls + "_motion"), nullptr, spi, tab4("u_strk"), r.sumNS, maxTp));
            x_mot = buf(ls + "_motion");
        }
        // object -> language attention (module/module.py:46-50)
        float* x_o2l = buf(ls + "_o2l");
        SOLA_TRY(linear3(x_mot, nullptr, nullptr, l, 2, 1, M, q, nullptr, nullptr, 0, o2l_in_sp, nullptr));
        SOLA_TRY(linear3(lang_in, lang_in, nullptr, l, 2, 2, r.LW, buf("lk"), buf("lv"), nullptr, 1, o2l_in_sp, (sp || h16) ? c->scal_pair(1) + 1 : nullptr));
        SOLA_TRY(attention(q, buf("lk"), buf("lv"), S, tab4("u_smp"), tab4("u_langk"), r.maxRowsSample, r.maxW, o2l_in_sp));
        SOLA_TRY(out_proj(l, 2, M, x_mot, spi));
        SOLA_TRY(gn(l, 2, x_o2l, nullptr, (sp && !last) ? 1 : 0, tab4("u_smp"), S, r.maxRowsSample, last ? 0 : 1));  // the score head reads f32
        xin = x_o2l;
    }
    HeadDesc hd{xin, buf("lbar"), score_map, score_tokens, 1, r.sumNS, maxTp, D};
    hd.units = tab4("u_strk");
    SOLA_TRY(launch_score_head(hd, s));
    return SOLA_OK;
}
