// api.cpp -- the C ABI of libsdhip.so (include/sdhip.h): context, host-buffer wrappers
// around the device pipelines, whole-path driver (speakerDiarization(), sd.cpp:2937-3234),
// wav reader (wav.h:62-126) and the measurement hooks.
#include "common.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>

static std::string g_create_err;

int seg_prec(const sd_ctx* c) { return c->seg_precision >= 0 ? c->seg_precision : (c->ecapa_precision != 0 ? 3 : 0); }

extern "C" const char* sd_create_error(void) { return g_create_err.c_str(); }
extern "C" const char* sd_last_error(const sd_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" sd_ctx* sd_create(const char* seg_path, const char* emb_path, int device)
{
    g_create_err.clear();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_err = "no HIP device available (libsdhip has no CPU fallback)"; return nullptr; }
    if (device < 0 || device >= ndev) { g_create_err = "device id out of range"; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return nullptr; }
    sd_ctx* c = new sd_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_create_err = "hipStreamCreate failed"; delete c; return nullptr; }
    auto fail = [&](const std::string& m) { g_create_err = m; sd_destroy(c); return (sd_ctx*)nullptr; };
    const bool trace = getenv("SD_TRACE_CREATE") != nullptr;          // where the start-up time goes (tools/cold_start.py)
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    if (seg_path && seg_path[0]) {
        Pack p; std::string e;
        if (load_model_any(seg_path, 0, p, e)) return fail(e);
        const double t1 = now();
        if (build_seg_weights(c, p)) return fail(c->err);
        if (trace) fprintf(stderr, "sd_create: segmentation model read %.1f ms, layouts + upload %.1f ms\n", t1 - t0, now() - t1);
    }
    t0 = now();
    if (emb_path && emb_path[0]) {
        Pack p; std::string e;
        if (load_model_any(emb_path, 1, p, e)) return fail(e);
        const double t1 = now();
        if (build_ecapa_weights(c, p)) return fail(c->err);
        if (trace) fprintf(stderr, "sd_create: embedding model read %.1f ms, layouts + upload %.1f ms\n", t1 - t0, now() - t1);
    }
    c->err.clear();
    return c;
}

extern "C" void sd_destroy(sd_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    sd_flush_profile(c);
    (void)sd_comm_destroy(c);
    for (void* p : c->owned) (void)hipFree(p);
    for (auto& kv : c->ws) kv.second.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void sd_flush_profile(sd_ctx* c)
{
    for (auto& t : c->pending) {
        hipEvent_t e0 = std::get<1>(t), e1 = std::get<2>(t);
        float ms = 0;
        if (hipEventSynchronize(e1) == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
        KernelStat& s = c->stats[std::get<0>(t)];
        s.ms += ms; s.launches++; s.flops += std::get<3>(t); s.bytes += std::get<4>(t);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    c->pending.clear();
}

extern "C" int sd_kernel_stats(const sd_ctx* cc, const char* kernel, double* total_ms, int64_t* launches, double* flops, double* bytes)
{
    sd_ctx* c = const_cast<sd_ctx*>(cc);
    if (!c || !kernel) return SD_ERR_ARG;
    (void)hipStreamSynchronize(c->stream);
    sd_flush_profile(c);
    auto it = c->stats.find(kernel);
    KernelStat s; if (it != c->stats.end()) s = it->second;
    if (total_ms) *total_ms = s.ms; if (launches) *launches = s.launches; if (flops) *flops = s.flops; if (bytes) *bytes = s.bytes;
    return SD_OK;
}
// test hook: copy `bytes` of the named workspace (from `offset`) to the host -- intermediate activations for the precision diagnostics of tools/
extern "C" int sd_debug_read_ws(sd_ctx* c, const char* name, int64_t offset, void* h_out, int64_t bytes)
{
    if (!c || !name || !h_out || offset < 0 || bytes < 0) return SD_ERR_ARG;
    auto it = c->ws.find(name);
    if (it == c->ws.end() || !it->second.p) SD_FAIL(c, SD_ERR_ARG, "no workspace named %s", name);
    if ((size_t)(offset + bytes) > it->second.cap) SD_FAIL(c, SD_ERR_ARG, "workspace %s holds %zu bytes", name, it->second.cap);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(h_out, (const char*)it->second.p + offset, (size_t)bytes, hipMemcpyDeviceToHost));
    return SD_OK;
}
extern "C" void sd_reset_stats(sd_ctx* c) { if (!c) return; (void)hipStreamSynchronize(c->stream); sd_flush_profile(c); c->stats.clear(); }
extern "C" int sd_stage_ms(const sd_ctx* c, double* ms4) { if (!c || !ms4) return SD_ERR_ARG; for (int i = 0; i < 4; ++i) ms4[i] = c->stage_ms[i]; return SD_OK; }
extern "C" int sd_set_option(sd_ctx* c, const char* key, int64_t v)
{
    if (!c || !key) return SD_ERR_ARG;
    std::string k(key);
    if (k == "emb_batch_items") { c->emb_batch_items = v; c->emb_batch_explicit = true; }
    else if (k == "seg_batch_chunks") c->seg_batch_chunks = v;
    else if (k == "ws_limit_mb") g_ws_limit.store(v > 0 ? (size_t)v << 20 : 0);      // test hook, process-wide (sdhip_test.h)
    else if (k == "emb_batch_default") { c->emb_batch_items = 3072; c->emb_batch_explicit = false; c->embed_calls = v; }   // test hook: back to the implicit plan (v = calls already made)
    else if (k == "linkage_wgs") c->linkage_wgs = v;
    else if (k == "linkage_threads") c->linkage_threads = v;
    else if (k == "linkage_kernel") c->linkage_kernel = v;
    else if (k == "linkage_tie_kernel") c->linkage_tie_kernel = v;
    else if (k == "linkage_zero_phase") c->linkage_zero_phase = v;
    else if (k == "linkage_force_heap") c->linkage_force_heap = v != 0;
    else if (k == "linkage_hx_wide") c->linkage_hx_wide = v != 0;
    else if (k == "linkage_prefetch") c->linkage_prefetch = v != 0;
    else if (k == "linkage_one_xcd") c->linkage_one_xcd = v;
    else if (k == "linkage_square") c->linkage_square = v;
    else if (k == "skip_dead_rows") c->skip_dead_rows = v != 0;
    else if (k == "conv_h256") c->conv_h256 = v != 0;
    else if (k == "conv_glds") c->conv_glds = v != 0;
    else if (k == "conv_pp") c->conv_pp = v != 0;
    else if (k == "conv_glds_f32") c->conv_glds_f32 = v != 0;
    else if (k == "conv_mfma16") { if (v < 0 || v > 2) return SD_ERR_ARG; c->conv_mfma16 = (int)v; }      // 2: k_conv_gemm_g256 takes the 16x16x32 form on short contractions too (the reference the round-6 kernel is compared with bit for bit)
    else if (k == "conv_rot") { if (v < 0 || v > 3) return SD_ERR_ARG; c->conv_rot = (int)v; }
    else if (k == "conv_stagger") { if (v < 0 || v > 2) return SD_ERR_ARG; c->conv_stagger = (int)v; }
    else if (k == "conv_w256_f32") c->conv_w256_f32 = v != 0;
    else if (k == "conv_pn") c->conv_pn = (int)v;
    else if (k == "ecapa_ld_pad") { if (v < 0 || v > 1024 || ((int64_t)v & 7)) return SD_ERR_ARG; c->ecapa_ld_pad = (int)v; }
    else if (k == "seg_shared_conv0") c->seg_shared_conv0 = v != 0;
    else if (k == "seg_wide_ih") c->seg_wide_ih = v != 0;
    else if (k == "conv_w256_kmin") c->conv_w256_kmin = (int)v;
    else if (k == "conv_pn128") c->conv_pn128 = (int)v;
    else if (k == "seg_precision") { if (v != 0 && v != 3 && v != -1) SD_FAIL(c, SD_ERR_ARG, "seg_precision must be 0 (f32), 3 (split fp16 operands for the LSTM) or -1 (auto: 3 whenever ecapa_precision is not 0)"); c->seg_precision = (int)v; }
    else if (k == "ecapa_precision") { if (v < 0 || v > 3) SD_FAIL(c, SD_ERR_ARG, "ecapa_precision must be 0 (f32), 1 (fp16), 2 (fp16, hi + lo weight planes) or 3 (f32 tensors, split fp16 operands on the wide layers)");
                                        if (hipSetDevice(c->device) != hipSuccess) SD_FAIL(c, SD_ERR_HIP, "hipSetDevice failed");
                                        const int rcw = ensure_ecapa_mode_weights(c, (int)v); if (rcw) return rcw;      // fp16 weight forms: built on first use of the mode
                                        c->ecapa_precision = (int)v; }
    else if (k == "ecapa_f16_hp") { if (v != 0 && v != 1 && v != 3) SD_FAIL(c, SD_ERR_ARG, "ecapa_f16_hp must be 0, 1 (MFA output in f32) or 3 (+ the attention branch)"); c->ecapa_f16_hp = (int)v; }
    else if (k == "ecapa_keep_cat") c->ecapa_keep_cat = v != 0;
    else if (k == "diag_res2_single") c->diag_res2_single = v != 0;
    else if (k == "rank0_permille") c->rank0_permille = (int)v;
    else if (k == "virtual_world") c->virtual_world = (int)v;
    else if (k == "comm_timeout_ms") c->comm_timeout_ms = v;
    else if (k == "inject_fail_rank") c->inject_fail_rank = (int)v;
    else if (k == "constrained_assignment") c->constrained_assignment = v != 0;
    else if (k == "num_clusters") c->num_clusters = (int)v;
    else if (k == "min_clusters") c->min_clusters = (int)v;
    else if (k == "max_clusters") c->max_clusters = (int)v;
    else if (k == "profile") { c->profile = v != 0; c->profile_detail = v >= 2; }
    else SD_FAIL(c, SD_ERR_ARG, "unknown option %s", key);
    return SD_OK;
}

// ------------------------------------------------------------------ helpers
extern "C" int64_t sd_num_chunks(int64_t n, int64_t* last_len)
{
    int64_t i = 0, cnt = 0;
    if (n > SD_CHUNK) { cnt = (n - SD_CHUNK + SD_HOP - 1) / SD_HOP; i = cnt * SD_HOP; }   // while (i + window < n), sd.cpp:1419
    int64_t ll = 0;
    if (i + 1 < n) { ll = n - i; cnt++; }                                                  // sd.cpp:1457
    if (last_len) *last_len = ll;
    return cnt;
}

struct DevTmp {   // RAII device scratch for the host-pointer wrappers
    void* p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess ? 0 : 1; }
};
#define DTMP(ctx, var, bytes) DevTmp var; if (var.alloc(bytes)) SD_FAIL(ctx, SD_ERR_HIP, "hipMalloc(%zu) failed", (size_t)(bytes))
#define ENTER(ctx) do { if (!(ctx)) return SD_ERR_ARG; (ctx)->err.clear(); if (hipSetDevice((ctx)->device) != hipSuccess) SD_FAIL(ctx, SD_ERR_HIP, "hipSetDevice failed"); } while (0)

// ------------------------------------------------------------------ embedding path
extern "C" int sd_embed_dev(sd_ctx* c, const float* d_wav, int64_t n, const float* d_masks, int64_t items, int64_t first_item, float* d_emb)
{
    ENTER(c);
    if (!d_wav || !d_masks || !d_emb || n <= 0 || items < 0) SD_FAIL(c, SD_ERR_ARG, "sd_embed_dev: bad argument");
    int rc = run_embed(c, d_wav, n, d_masks, items, first_item, d_emb);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SD_OK;
}

extern "C" int sd_embed(sd_ctx* c, const float* h_wav, int64_t n, const float* h_masks, int64_t items, float* h_emb)
{
    ENTER(c);
    if (!h_wav || !h_masks || !h_emb || n <= 0 || items < 0) SD_FAIL(c, SD_ERR_ARG, "sd_embed: bad argument");
    DTMP(c, dw, n * sizeof(float)); DTMP(c, dm, items * SD_FRAMES * sizeof(float)); DTMP(c, de, items * SD_EMB_DIM * sizeof(float));
    HIPCHK(c, hipMemcpy(dw.p, h_wav, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dm.p, h_masks, items * SD_FRAMES * sizeof(float), hipMemcpyHostToDevice));
    int rc = run_embed(c, (const float*)dw.p, n, (const float*)dm.p, items, 0, (float*)de.p);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(h_emb, de.p, items * SD_EMB_DIM * sizeof(float), hipMemcpyDeviceToHost));
    return SD_OK;
}

extern "C" int sd_frontend(sd_ctx* c, const float* h_wav, int64_t n, const float* h_masks, int64_t items, float* h_feats, float* h_lens)
{
    ENTER(c);
    if (!h_wav || !h_masks || n <= 0 || items <= 0) SD_FAIL(c, SD_ERR_ARG, "sd_frontend: bad argument");
    DTMP(c, dw, n * sizeof(float)); DTMP(c, dm, items * SD_FRAMES * sizeof(float));
    DTMP(c, df, items * SD_TP * SD_FEAT_LD * sizeof(float)); DTMP(c, dl, items * sizeof(float));
    DTMP(c, dn, items * sizeof(int)); DTMP(c, dv, items * sizeof(int)); DTMP(c, dg, items * sizeof(int)); DTMP(c, dr, (items + 1) * sizeof(int));
    HIPCHK(c, hipMemcpy(dw.p, h_wav, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dm.p, h_masks, items * SD_FRAMES * sizeof(float), hipMemcpyHostToDevice));
    std::vector<int> ro((size_t)items + 1);
    for (int64_t i = 0; i <= items; ++i) ro[(size_t)i] = (int)(i * SD_TP);                  // every frame of every item (reference layout)
    HIPCHK(c, hipMemcpy(dr.p, ro.data(), ro.size() * sizeof(int), hipMemcpyHostToDevice));
    int rc = frontend_prepare(c, (const float*)dm.p, items, 0, (float*)dl.p, (int*)dn.p, (int*)dv.p, (int*)dg.p, false, nullptr, nullptr);
    if (rc) return rc;
    if ((rc = frontend_features(c, (const float*)dw.p, n, 0, items, false, (const int*)dn.p, (const int*)dr.p, (float*)df.p))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h_feats) {
        std::vector<float> tmp((size_t)items * SD_TP * SD_FEAT_LD);
        HIPCHK(c, hipMemcpy(tmp.data(), df.p, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < items; ++i)
            for (int t = 0; t < SD_T; ++t)
                memcpy(h_feats + ((size_t)i * SD_T + t) * SD_NMELS, &tmp[((size_t)i * SD_TP + t) * SD_FEAT_LD], SD_NMELS * sizeof(float));
    }
    if (h_lens) HIPCHK(c, hipMemcpy(h_lens, dl.p, items * sizeof(float), hipMemcpyDeviceToHost));
    return SD_OK;
}

extern "C" int sd_ecapa(sd_ctx* c, const float* h_feats, const float* h_lens, int64_t items, float* h_emb)
{
    ENTER(c);
    if (!h_feats || !h_lens || !h_emb || items <= 0) SD_FAIL(c, SD_ERR_ARG, "sd_ecapa: bad argument");
    std::vector<int> nv((size_t)items);
    EcapaRowPlan plan;
    for (int64_t i = 0; i < items; ++i) {
        float lt = h_lens[i] * (float)SD_T;
        int v = (int)ceilf(lt); if (v > SD_T) v = SD_T; if (v < 1) v = 1;
        nv[(size_t)i] = v;
    }
    DTMP(c, dv, items * sizeof(int)); DTMP(c, dr, EC_SPACES * (items + 1) * sizeof(int)); DTMP(c, de, items * SD_EMB_DIM * sizeof(float));
    int rc = ecapa_row_plan(c, nv.data(), items, plan, (int*)dr.p);
    if (rc) return rc;
    const std::vector<int>& rowoff = plan.off[0];
    const int64_t rows = rowoff[(size_t)items];
    std::vector<float> tmp((size_t)rows * SD_FEAT_LD, 0.0f);                                 // compact rows: the frames each item needs
    for (int64_t i = 0; i < items; ++i)
        for (int t = 0; t < rowoff[(size_t)i + 1] - rowoff[(size_t)i]; ++t)
            memcpy(&tmp[((size_t)rowoff[(size_t)i] + t) * SD_FEAT_LD], h_feats + ((size_t)i * SD_T + t) * SD_NMELS, SD_NMELS * sizeof(float));
    DTMP(c, df, tmp.size() * sizeof(float));
    HIPCHK(c, hipMemcpy(df.p, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dv.p, nv.data(), items * sizeof(int), hipMemcpyHostToDevice));
    int64_t nb = (c->emb_batch_items / 96) * 96; if (nb < 96) nb = 96;
    const int64_t cap_rows = nb * SD_TP;
    rc = ecapa_run_batches(c, [&]() -> int {
        for (int64_t a0 = 0; a0 < items;) {
            int64_t a1 = a0;
            while (a1 < items && a1 - a0 < ROWTAB_MAX_ITEMS && rowoff[(size_t)a1 + 1] - rowoff[(size_t)a0] <= cap_rows) ++a1;
            if (a1 == a0) a1 = a0 + 1;
            const int r = run_ecapa(c, (const float*)df.p, (const int*)dv.p, plan, a0, a1, (float*)de.p);
            if (r) return r;
            a0 = a1;
        }
        return SD_OK;
    });
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(h_emb, de.p, items * SD_EMB_DIM * sizeof(float), hipMemcpyDeviceToHost));
    return SD_OK;
}

// EmbeddingModel1::infer as the reference declares it (sd.cpp:1977-2040, calling _infer sd.cpp:1889-1970): B already compacted, zero-padded
// signals of 80000 samples and their relative lengths -> [B][192].  No NaN rule here (getEmbedding applies it to the result, sd.cpp:2479-2549);
// the STFT / fbank run over all 501 frames of every row as the reference's do, the network over the frames wav_lens leaves valid + its reach.
extern "C" int sd_embed_signals(sd_ctx* c, const float* h_signals, const float* h_wav_lens, int64_t B, float* h_emb)
{
    ENTER(c);
    if (!h_signals || !h_wav_lens || !h_emb || B <= 0) SD_FAIL(c, SD_ERR_ARG, "sd_embed_signals: bad argument");
    if (!c->ew.loaded) SD_FAIL(c, SD_ERR_MODEL, "embedding model not loaded");
    std::vector<int> nv((size_t)B), nn((size_t)B);
    for (int64_t i = 0; i < B; ++i) {
        if (!(h_wav_lens[i] > 0.0f) || h_wav_lens[i] > 1.0f) SD_FAIL(c, SD_ERR_ARG, "sd_embed_signals: wav_lens[%lld] = %g (relative length in (0, 1])", (long long)i, (double)h_wav_lens[i]);
        const float lt = h_wav_lens[i] * (float)SD_T;                       // float32 product as in torch (k_wav_lens)
        nn[(size_t)i] = (int)rintf(lt);                                      // torch.round, threeModel.py:358
        int v = (int)ceilf(lt); if (v > SD_T) v = SD_T; if (v < 1) v = 1;    // arange(L) < len * L
        nv[(size_t)i] = v;
        if (nn[(size_t)i] < 1) SD_FAIL(c, SD_ERR_ARG, "sd_embed_signals: wav_lens[%lld] = %g leaves no frame to normalise over", (long long)i, (double)h_wav_lens[i]);
    }
    const int64_t ns = B * (int64_t)SD_CHUNK;
    // persistent workspaces, not per-call allocations: the reference calls infer once per batch of 32 items (677 times per hour of audio)
    WS(c, float, dw_p, "sig_wav", ns + 512); WS(c, int, dv_p, "sig_nvalid", B); WS(c, int, dn_p, "sig_nnorm", B);
    WS(c, int, dr_p, "sig_rowoff", EC_SPACES * (B + 1)); WS(c, float, de_p, "sig_emb", B * SD_EMB_DIM);
    struct { void* p; } dw{dw_p}, dv{dv_p}, dn{dn_p}, dr{dr_p}, de{de_p};
    HIPCHK(c, hipMemcpyAsync(dw.p, h_signals, ns * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync((float*)dw.p + ns, 0, 512 * sizeof(float), c->stream));
    HIPCHK(c, hipMemcpyAsync(dv.p, nv.data(), B * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dn.p, nn.data(), B * sizeof(int), hipMemcpyHostToDevice, c->stream));
    EcapaRowPlan plan;
    int rc = ecapa_row_plan(c, nv.data(), B, plan, (int*)dr.p);          // (synchronises the stream: the host vectors above may go)
    if (rc) return rc;
    const std::vector<int>& rowoff = plan.off[0];
    WS(c, float, df_p, "sig_feats", (size_t)rowoff[(size_t)B] * SD_FEAT_LD);
    struct { void* p; } df{df_p};
    if ((rc = frontend_prepare_signals(c, B))) return rc;
    const int64_t origin = c->wav_origin;
    c->wav_origin = 0; c->fe_bill_samples = -1;
    rc = frontend_features(c, (const float*)dw.p, ns, 0, B, false, (const int*)dn.p, (const int*)dr.p, (float*)df.p, true);
    c->wav_origin = origin;
    if (rc) return rc;
    int64_t nb = (c->emb_batch_items / 96) * 96; if (nb < 96) nb = 96;
    if (nb > 768 && !c->emb_batch_explicit) nb = 768;                        // an operator-level call: the small arena
    const int64_t cap_rows = nb * SD_TP;
    rc = ecapa_run_batches(c, [&]() -> int {
        for (int64_t a0 = 0; a0 < B;) {
            int64_t a1 = a0;
            while (a1 < B && a1 - a0 < ROWTAB_MAX_ITEMS && rowoff[(size_t)a1 + 1] - rowoff[(size_t)a0] <= cap_rows) ++a1;
            if (a1 == a0) a1 = a0 + 1;
            const int r = run_ecapa(c, (const float*)df.p, (const int*)dv.p, plan, a0, a1, (float*)de.p);
            if (r) return r;
            a0 = a1;
        }
        return SD_OK;
    });
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(h_emb, de.p, B * SD_EMB_DIM * sizeof(float), hipMemcpyDeviceToHost));
    return SD_OK;
}

// ------------------------------------------------------------------ wav reader (a1)
// bytes of the data chunk that are really in the file.  A wav written to a pipe (ffmpeg, sox) announces 0xFFFFFFFF (or 0) because the
// writer could not seek back; the reference trusts the header (wav.h:92-97) and would "read" two billion samples from a short file.
// Deliberate deviation, input robustness only: what is announced beyond the end of the file is not read, and a size of
// 0xFFFFFFFF means "up to the end of the file"; so does 0, unless what follows the empty data chunk is a well-formed chain of RIFF
// sub-chunks (LIST, id3 ...) -- then the file really has no samples, as the reference reads it.
static uint32_t data_bytes_present(FILE* fp, uint32_t announced)
{
    const long cur = ftell(fp);
    if (cur < 0 || fseek(fp, 0, SEEK_END) != 0) return announced;
    const long end = ftell(fp);
    (void)fseek(fp, cur, SEEK_SET);
    if (end < cur) return announced;
    const uint64_t remaining = (uint64_t)(end - cur);
    const uint32_t to_end = (uint32_t)(remaining > 0xFFFFFFFEull ? 0xFFFFFFFEull : remaining);
    if (announced == 0xFFFFFFFFu || (uint64_t)announced > remaining) return to_end;
    if (announced == 0 && remaining > 0) {
        // 0 is what some stream writers leave -- but also what a well-formed file with an EMPTY data chunk followed by LIST / id3
        // metadata says (the reference reads 0 samples there).  Stream rule only when what follows is NOT a chain of RIFF sub-chunks
        // that ends exactly at the end of the file.
        long pos = cur;
        bool chain = true;
        while (pos < end) {
            unsigned char t8[8];
            if (end - pos < 8 || fseek(fp, pos, SEEK_SET) != 0 || fread(t8, 1, 8, fp) != 8) { chain = false; break; }
            for (int q = 0; q < 4; ++q) if (t8[q] < 0x20 || t8[q] > 0x7e) chain = false;
            uint32_t sz; memcpy(&sz, t8 + 4, 4);
            if (!chain || (uint64_t)sz > (uint64_t)(end - pos - 8)) { chain = false; break; }
            pos += 8 + (long)sz;
            if (pos < end && (sz & 1) && end - pos >= 1) pos += 1;      // RIFF pad byte
        }
        (void)fseek(fp, cur, SEEK_SET);
        return chain ? 0u : to_end;
    }
    return announced;
}

// The header walk both readers share (wav.h:62-90): the 44-byte canonical header, a longer fmt chunk (75-79), LIST / fact / ...
// sub-chunks in front of the data (85-90).  On success the file is positioned on the first data byte and *dsz holds the bytes of
// data really present.  Returns nullptr (file closed) when the file cannot be opened or ends inside its headers.
static FILE* wav_open_data(const char* path, uint32_t* sr, uint16_t* ch, uint16_t* bits, uint32_t* dsz)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) return nullptr;                          // reference ignores this (wav.h:60) and crashes later
    unsigned char h[44];
    if (fread(h, 1, 44, fp) != 44) { fclose(fp); return nullptr; }
    uint32_t fmt_size; char tag[4];
    memcpy(&fmt_size, h + 16, 4); memcpy(ch, h + 22, 2); memcpy(sr, h + 24, 4); memcpy(bits, h + 34, 2);
    memcpy(tag, h + 36, 4); memcpy(dsz, h + 40, 4);
    if (fmt_size < 16) { fclose(fp); return nullptr; }
    unsigned char t8[8];
    if (fmt_size > 16) {                               // wav.h:75-79
        fseek(fp, 44 - 8 + (long)fmt_size - 16, SEEK_SET);
        if (fread(t8, 1, 8, fp) != 8) { fclose(fp); return nullptr; }
        memcpy(tag, t8, 4); memcpy(dsz, t8 + 4, 4);
    }
    while (strncmp(tag, "data", 4) != 0) {             // wav.h:85-90 skip LIST/fact chunks
        fseek(fp, (long)*dsz, SEEK_CUR);
        if (fread(t8, 1, 8, fp) != 8) { fclose(fp); return nullptr; }
        memcpy(tag, t8, 4); memcpy(dsz, t8 + 4, 4);
    }
    *dsz = data_bytes_present(fp, *dsz);
    return fp;
}

extern "C" int sd_read_wav(const char* path, int16_t** pcm, int64_t* n, int32_t* sample_rate, int32_t* channels)
{
    if (!path || !pcm || !n) return SD_ERR_ARG;
    *pcm = nullptr; *n = 0;
    uint32_t sr, dsz; uint16_t ch, bits;
    FILE* fp = wav_open_data(path, &sr, &ch, &bits, &dsz);
    if (!fp) return SD_ERR_ARG;
    if (bits != 16) { fclose(fp); return SD_ERR_ARG; } // README.md:37: 16 kHz / mono / 16 bit only
    int64_t num = dsz / 2;
    int16_t* buf = (int16_t*)malloc((size_t)(num > 0 ? num : 1) * sizeof(int16_t));
    int64_t got = (int64_t)fread(buf, 2, (size_t)num, fp);
    for (int64_t i = got; i < num; ++i) buf[i] = 0;
    fclose(fp);
    *pcm = buf;
    *n = num / (ch ? ch : 1);                          // wav.h:97 (interleaved data read as mono)
    if (sample_rate) *sample_rate = (int32_t)sr;
    if (channels) *channels = ch;
    return SD_OK;
}
extern "C" void sd_free_pcm(int16_t* p) { free(p); }
extern "C" void sd_free_turns(sd_turn* t) { free(t); }
extern "C" int sd_format_turn(const sd_turn* t, char* buf, int cap)
{
    if (!t || !buf) return SD_ERR_ARG;
    snprintf(buf, (size_t)cap, "[%g -- %g] --> Speaker_%d", t->start, t->end, t->label);   // sd.cpp:3439, iostream default precision
    return SD_OK;
}

// ------------------------------------------------------------------ wav reader, all bit depths the reference reads (wav.h:99-122)
// 8-bit samples are read as signed char, 32-bit as int, every depth is then divided by 32768 (sd.cpp:2950) -- the
// reference's scaling, kept as is.  Returns malloc'd float samples (sd_free_wav).
extern "C" int sd_read_wav_f32(const char* path, float** wav, int64_t* n, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample)
{
    if (!path || !wav || !n) return SD_ERR_ARG;
    *wav = nullptr; *n = 0;
    uint32_t sr, dsz; uint16_t ch, bits;
    FILE* fp = wav_open_data(path, &sr, &ch, &bits, &dsz);
    if (!fp) return SD_ERR_ARG;
    if (bits != 8 && bits != 16 && bits != 32) { fclose(fp); return SD_ERR_ARG; }     // reference: exit(1), wav.h:119-121
    const int bps = bits / 8;
    const int64_t num = dsz / bps;
    std::vector<unsigned char> raw((size_t)(num > 0 ? num * bps : 1));
    const size_t got = fread(raw.data(), 1, (size_t)num * bps, fp);
    fclose(fp);
    if (got < (size_t)num * bps) memset(raw.data() + got, 0, (size_t)num * bps - got);
    float* out = (float*)malloc(sizeof(float) * (size_t)(num > 0 ? num : 1));
    for (int64_t i = 0; i < num; ++i) {
        float v;
        if (bits == 8) v = (float)(signed char)raw[(size_t)i];
        else if (bits == 16) { int16_t s; memcpy(&s, &raw[(size_t)i * 2], 2); v = (float)s; }
        else { int32_t s; memcpy(&s, &raw[(size_t)i * 4], 4); v = (float)s; }
        out[i] = (float)((double)(v * 1.0f) / 32768.0);
    }
    *wav = out;
    *n = num / (ch ? ch : 1);
    if (sample_rate) *sample_rate = (int32_t)sr;
    if (channels) *channels = ch;
    if (bits_per_sample) *bits_per_sample = bits;
    return SD_OK;
}
extern "C" void sd_free_wav(float* p) { free(p); }

// ------------------------------------------------------------------ output formats
// Relabelling.  The reference prints the raw cluster index (sd.cpp:3439).  pyannote.audio renames the clusters on the way
// out: SpeakerDiarization.apply maps `diarization.labels()` -- the labels that occur, sorted BY THEIR STRING -- onto
// SPEAKER_00, SPEAKER_01, ... (mode 1); mode 0 numbers the speakers in order of first appearance.
extern "C" int sd_relabel_turns_ex(sd_turn* turns, int64_t n, int mode)
{
    if (n < 0 || (n > 0 && !turns) || (mode != 0 && mode != 1)) return SD_ERR_ARG;
    std::vector<int> seen;
    for (int64_t i = 0; i < n; ++i) if (std::find(seen.begin(), seen.end(), turns[i].label) == seen.end()) seen.push_back(turns[i].label);
    if (mode == 1) std::sort(seen.begin(), seen.end(), [](int a, int b) { return std::to_string(a) < std::to_string(b); });
    for (int64_t i = 0; i < n; ++i) turns[i].label = (int32_t)(std::find(seen.begin(), seen.end(), turns[i].label) - seen.begin());
    return SD_OK;
}
extern "C" int sd_relabel_turns(sd_turn* turns, int64_t n) { return sd_relabel_turns_ex(turns, n, 1); }

extern "C" int sd_last_confidence(const sd_ctx* c, double* conf, int64_t cap, int64_t* n)
{
    if (!c) return SD_ERR_ARG;
    if (n) *n = (int64_t)c->last_conf.size();
    if (conf) for (int64_t i = 0; i < cap && i < (int64_t)c->last_conf.size(); ++i) conf[i] = c->last_conf[(size_t)i];
    return SD_OK;
}

// one "SPEAKER <uri> 1 <start> <duration> <NA> <NA> SPEAKER_<kk> <NA> <conf>" line per turn (pyannote's RTTM writer layout;
// the last field is RTTM's confidence column, <NA> without conf)
extern "C" int sd_write_rttm_ex(const char* path, const char* uri, const sd_turn* turns, int64_t n_turns, const double* conf)
{
    if (!path || (n_turns > 0 && !turns) || n_turns < 0) return SD_ERR_ARG;
    FILE* f = fopen(path, "w");
    if (!f) return SD_ERR_ARG;
    for (int64_t i = 0; i < n_turns; ++i) {
        fprintf(f, "SPEAKER %s 1 %.3f %.3f <NA> <NA> SPEAKER_%02d <NA> ", (uri && uri[0]) ? uri : "audio",
                turns[i].start, turns[i].end - turns[i].start, turns[i].label);
        if (conf && conf[i] == conf[i]) fprintf(f, "%.4f\n", conf[i]); else fprintf(f, "<NA>\n");
    }
    fclose(f);
    return SD_OK;
}
extern "C" int sd_write_rttm(const char* path, const char* uri, const sd_turn* turns, int64_t n_turns)
{
    return sd_write_rttm_ex(path, uri, turns, n_turns, nullptr);
}
