// ecapa.hip -- ECAPA-TDNN (C=1024, attention 128, 192-d) body of the embedding model.
// Replaces EmbeddingModel1::_infer's Ort::Session::Run on emd4.onnx (sd.cpp:1889-1970);
// architecture = speechbrain 0.5.14 ECAPA_TDNN as exported by embeddings/export3.py:151-190.
//
// All dense contractions run through conv_gemm.hip (f32 MFMA).  This file holds the
// HBM-bound glue kernels (masked SE mean, SE apply + residual, ASP statistics and the
// attentive softmax pooling) and the layer schedule.  Activations are channels-last
// [item][501][C] (no padding rows).
#include "common.h"
#include <algorithm>

// masked mean over the first nvalid[item] frames  -> out[item][C]   (SEBlock, lengths given)
__global__ void k_masked_mean(const float* __restrict__ x, int ld, const int* __restrict__ nvalid, float* __restrict__ out, int C)
{
    const int item = blockIdx.y, ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C) return;
    const int nv = nvalid[item];
    const float* p = x + (size_t)item * SD_TP * ld + ch;
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int t = 0;
    for (; t + 3 < nv; t += 4) { s0 += p[(size_t)t * ld]; s1 += p[(size_t)(t + 1) * ld]; s2 += p[(size_t)(t + 2) * ld]; s3 += p[(size_t)(t + 3) * ld]; }
    for (; t < nv; ++t) s0 += p[(size_t)t * ld];
    out[(size_t)item * C + ch] = ((s0 + s1) + (s2 + s3)) / (float)nv;
}

// y = gate[item][c] * t2 + residual   (SERes2NetBlock tail), float4 over channels
__global__ void k_se_apply(const float* __restrict__ t2, const float* __restrict__ gate, const float* __restrict__ res, int res_ld,
                           float* __restrict__ y, int y_ld, int C, int64_t rows)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4n = C / 4;
    if (idx >= rows * c4n) return;
    const int64_t row = idx / c4n;
    const int c = (int)(idx - row * c4n) * 4;
    const int64_t item = row / SD_TP;
    const float4 a = *(const float4*)(t2 + row * C + c);
    const float4 g = *(const float4*)(gate + item * C + c);
    const float4 r = *(const float4*)(res + row * res_ld + c);
    float4 o;
    o.x = g.x * a.x + r.x; o.y = g.y * a.y + r.y; o.z = g.z * a.z + r.z; o.w = g.w * a.w + r.w;
    *(float4*)(y + row * y_ld + c) = o;
}

// copy a channel slice [rows][w] between strided buffers (Res2Net first sub-band is identity)
__global__ void k_copy_slice(const float* __restrict__ src, int src_ld, float* __restrict__ dst, int dst_ld, int w, int64_t rows)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int w4 = w / 4;
    if (idx >= rows * w4) return;
    const int64_t row = idx / w4;
    const int c = (int)(idx - row * w4) * 4;
    *(float4*)(dst + row * dst_ld + c) = *(const float4*)(src + row * src_ld + c);
}

// ASP global-context statistics: mean / std over valid frames -> ms[item][2C]
// one pass (Welford) with 4 rows in flight per thread: the tensor is read once
__global__ void k_asp_stats(const float* __restrict__ x, int ld, const int* __restrict__ nvalid, float* __restrict__ ms, int C)
{
    const int item = blockIdx.y, ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C) return;
    const int nv = nvalid[item];
    const float* p = x + (size_t)item * SD_TP * ld + ch;
    float mean = 0.0f, m2 = 0.0f;
    int t = 0;
    for (; t + 3 < nv; t += 4) {
        const float v0 = p[(size_t)t * ld], v1 = p[(size_t)(t + 1) * ld], v2 = p[(size_t)(t + 2) * ld], v3 = p[(size_t)(t + 3) * ld];
        float d;
        d = v0 - mean; mean += d / (float)(t + 1); m2 += d * (v0 - mean);
        d = v1 - mean; mean += d / (float)(t + 2); m2 += d * (v1 - mean);
        d = v2 - mean; mean += d / (float)(t + 3); m2 += d * (v2 - mean);
        d = v3 - mean; mean += d / (float)(t + 4); m2 += d * (v3 - mean);
    }
    for (; t < nv; ++t) { const float v = p[(size_t)t * ld]; const float d = v - mean; mean += d / (float)(t + 1); m2 += d * (v - mean); }
    ms[(size_t)item * 2 * C + ch] = mean;
    ms[(size_t)item * 2 * C + C + ch] = sqrtf(fmaxf(m2 / (float)nv, 1e-12f));
}

// attentive statistics pooling: masked softmax over time of the logits, weighted mean/std of x.
// One pass over both tensors: online softmax (running max, rescaled weights) + weighted Welford update.
__global__ void k_asp_pool(const float* __restrict__ x, const float* __restrict__ logit, int ld, const int* __restrict__ nvalid,
                           float* __restrict__ pooled, int C)
{
    const int item = blockIdx.y, ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C) return;
    const int nv = nvalid[item];
    const float* px = x + (size_t)item * SD_TP * ld + ch;
    const float* pl = logit + (size_t)item * SD_TP * ld + ch;
    float mx = -INFINITY, W = 0.0f, mean = 0.0f, m2 = 0.0f;
    auto step = [&](float l, float v) {
        if (l > mx) { const float s = expf(mx - l); W *= s; m2 *= s; mx = l; }      // expf(-inf) = 0 on the first frame
        const float w = expf(l - mx);
        W += w;
        const float d = v - mean;
        mean += (w / W) * d;
        m2 += w * d * (v - mean);
    };
    int t = 0;
    for (; t + 3 < nv; t += 4) {
        const float l0 = pl[(size_t)t * ld], l1 = pl[(size_t)(t + 1) * ld], l2 = pl[(size_t)(t + 2) * ld], l3 = pl[(size_t)(t + 3) * ld];
        const float v0 = px[(size_t)t * ld], v1 = px[(size_t)(t + 1) * ld], v2 = px[(size_t)(t + 2) * ld], v3 = px[(size_t)(t + 3) * ld];
        step(l0, v0); step(l1, v1); step(l2, v2); step(l3, v3);
    }
    for (; t < nv; ++t) step(pl[(size_t)t * ld], px[(size_t)t * ld]);
    pooled[(size_t)item * 2 * C + ch] = mean;
    pooled[(size_t)item * 2 * C + C + ch] = sqrtf(fmaxf(m2 / W, 1e-12f));
}

// rows flagged too-short become NaN (sd.cpp:2541-2549)
__global__ void k_nan_rows(float* __restrict__ emb, const int* __restrict__ flags, int64_t items)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= items * SD_EMB_DIM) return;
    if (flags[idx / SD_EMB_DIM]) emb[idx] = __int_as_float(0x7fc00000);
}
// emb[item] = embedding of its compact slot, or NaN for items dropped before the network
__global__ void k_scatter_emb(const float* __restrict__ emb_c, const int* __restrict__ cidx, float* __restrict__ emb, int64_t items)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= items * SD_EMB_DIM) return;
    const int64_t item = idx / SD_EMB_DIM;
    const int a = cidx[item];
    emb[idx] = (a >= 0) ? emb_c[(size_t)a * SD_EMB_DIM + (idx - item * SD_EMB_DIM)] : __int_as_float(0x7fc00000);
}

static ConvArgs conv_args(const ConvLayer& L, const float* X, int x_ld, float* Y, int y_ld, int64_t M, bool per_item)
{
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.X = X; a.x_ld = x_ld; a.Y = Y; a.y_ld = y_ld; a.W = L.W; a.W16 = L.W16;
    a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.M = (int)M;
    if (per_item) { a.TpIn = a.TpOut = SD_TP; a.Tin = a.T = SD_T; }
    else { a.TpIn = a.TpOut = (int)M; a.Tin = a.T = (int)M; }
    a.Cin = L.CinPad; a.cin_real = L.Cin; a.Cout = L.Cout; a.KT = L.KT; a.dil = L.dil;
    a.pad_mode = 0;
    return a;
}

#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)

int run_ecapa(sd_ctx* c, const float* d_feats, const int* d_nvalid, const int* d_flags, int64_t items, float* d_emb,
              const int* h_nvalid)
{
    const EcapaWeights& E = c->ew;
    if (!E.loaded) SD_FAIL(c, SD_ERR_MODEL, "embedding model not loaded");
    if (items <= 0) return SD_OK;
    const int C = E.C, C3 = 3 * C;
    const int64_t M = items * SD_TP;
    if (M > 0x7fffffff / 2) SD_FAIL(c, SD_ERR_ARG, "ecapa batch too large");
    WS(c, float, x0, "ec_x0", M * C);
    WS(c, float, t1, "ec_t1", M * C);
    WS(c, float, rr, "ec_r", M * C);
    WS(c, float, t2, "ec_t2", M * C);
    WS(c, float, cat, "ec_cat", M * C3);
    WS(c, float, mfa, "ec_mfa", M * C3);
    WS(c, float, hid, "ec_hid", M * 128);
    WS(c, float, se_s, "ec_se_s", items * C);
    WS(c, float, se_h, "ec_se_h", items * 128);
    WS(c, float, se_g, "ec_se_g", items * C);
    WS(c, float, ms, "ec_ms", items * 2 * C3);
    WS(c, float, ib, "ec_ib", items * 128);
    WS(c, float, pooled, "ec_pooled", items * 2 * C3);
    int rc;
    hipStream_t st = c->stream;

    // ---- rows that matter.  Frames >= nvalid of an item are excluded from every statistic (SE mean, ASP), and a
    // frame t only sees frames within the network's receptive field: 2 (block0, k5) + 7*2*(2+3+4) = 128... per
    // side is the loose bound; the exact reach of one side is 2 + 7*(2 + 3 + 4) = 65 frames (each Res2Net block
    // chains 7 k3 convs of dilation 2 / 3 / 4; the 1x1 convs and the SE gate add none).  Row panels (128 rows)
    // that start at or beyond min(501, nvalid + 65) therefore cannot influence the embedding and are not
    // computed: they are simply absent from the per-XCD panel lists (panel p goes to list p % 8).
    const int* mlist = nullptr; const int* mcount = nullptr; int mlist_ld = 0; double rows_listed = 0;
    if (h_nvalid) {
        const int kReach = 65;
        std::vector<int> lists[8];
        const int64_t npanels = (M + 127) / 128;
        std::vector<char> live((size_t)npanels, 0);
        for (int64_t b = 0; b < items; ++b) {              // a panel is computed if it holds a needed row of any item it overlaps
            int need = h_nvalid[b] + kReach; if (need > SD_T) need = SD_T;
            if (need <= 0) continue;
            const int64_t r0 = b * SD_TP, r1 = r0 + need - 1;
            for (int64_t pnl = r0 / 128; pnl <= r1 / 128; ++pnl) live[(size_t)pnl] = 1;
            rows_listed += (double)need;
        }
        for (int64_t pnl = 0; pnl < npanels; ++pnl) if (live[(size_t)pnl]) lists[pnl & 7].push_back((int)pnl);
        size_t ld = 1;
        for (int x = 0; x < 8; ++x) if (lists[x].size() > ld) ld = lists[x].size();
        std::vector<int> flat(8 * ld, 0), cnt(8);
        for (int x = 0; x < 8; ++x) { cnt[x] = (int)lists[x].size(); std::copy(lists[x].begin(), lists[x].end(), flat.begin() + x * ld); }
        WS(c, int, d_mlist, "ec_mlist", 8 * ld);
        WS(c, int, d_mcount, "ec_mcount", 8);
        HIPCHK(c, hipMemcpyAsync(d_mlist, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(d_mcount, cnt.data(), 8 * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));                 // host vectors go out of scope
        mlist = d_mlist; mcount = d_mcount; mlist_ld = (int)ld;
    }
#define WITH_LIST(a) do { if (mlist) { (a).mlist = mlist; (a).mcount = mcount; (a).mlist_ld = mlist_ld; (a).rows_listed = rows_listed; } } while (0)

    // blocks[0]: TDNNBlock(80 -> C, k5)
    { ConvArgs a = conv_args(E.block0, d_feats, SD_FEAT_LD, x0, C, M, true); a.act1 = 1; WITH_LIST(a); if ((rc = launch_conv_gemm(c, a, "block0"))) return rc; }

    for (int b = 0; b < 3; ++b) {
        const auto& B = E.blk[b];
        const float* xin = (b == 0) ? x0 : cat + (size_t)(b - 1) * C;
        const int xin_ld = (b == 0) ? C : C3;
        { ConvArgs a = conv_args(B.tdnn1, xin, xin_ld, t1, C, M, true); a.act1 = 1; WITH_LIST(a); if ((rc = launch_conv_gemm(c, a, "tdnn1"))) return rc; }
        const int S = C / 8;
        hipLaunchKernelGGL(k_copy_slice, GRID1(M * (S / 4)), 0, st, t1, C, rr, C, S, M);
        KCHECK(c);
        for (int i = 1; i < 8; ++i) {
            ConvArgs a = conv_args(B.res[i - 1], t1 + i * S, C, rr + i * S, C, M, true);
            a.act1 = 1;
            if (i >= 2) { a.X2 = rr + (i - 1) * S; a.x2_ld = C; }
            WITH_LIST(a);
            if ((rc = launch_conv_gemm(c, a, "res2net"))) return rc;
        }
        { ConvArgs a = conv_args(B.tdnn2, rr, C, t2, C, M, true); a.act1 = 1; WITH_LIST(a); if ((rc = launch_conv_gemm(c, a, "tdnn2"))) return rc; }
        {
            ProfScope ps(c, "se_mean", 0, (double)items * SD_T * C * 4.0);
            hipLaunchKernelGGL(k_masked_mean, dim3((C + 255) / 256, (unsigned)items), dim3(256), 0, st, t2, C, d_nvalid, se_s, C);
            KCHECK(c);
        }
        { ConvArgs a = conv_args(B.se1, se_s, C, se_h, 128, items, false); a.act1 = 1; if ((rc = launch_conv_gemm(c, a, "se1"))) return rc; }
        { ConvArgs a = conv_args(B.se2, se_h, 128, se_g, C, items, false); a.act2 = 2; if ((rc = launch_conv_gemm(c, a, "se2"))) return rc; }
        {
            ProfScope ps(c, "se_apply", 0, (double)M * C * 12.0);
            hipLaunchKernelGGL(k_se_apply, GRID1(M * (C / 4)), 0, st, t2, se_g, xin, xin_ld, cat + (size_t)b * C, C3, C, M);
            KCHECK(c);
        }
    }
    // mfa: TDNNBlock(3C -> 3C, k1) over cat(x1,x2,x3)
    { ConvArgs a = conv_args(E.mfa, cat, C3, mfa, C3, M, true); a.act1 = 1; WITH_LIST(a); if ((rc = launch_conv_gemm(c, a, "mfa"))) return rc; }
    // ASP with global context: cat[x, mean, std] @ W == x @ Wx + (mean,std) @ Wms  (per-item bias)
    {
        ProfScope ps(c, "asp_stats", 0, (double)items * SD_T * C3 * 4.0);
        hipLaunchKernelGGL(k_asp_stats, dim3((C3 + 255) / 256, (unsigned)items), dim3(256), 0, st, mfa, C3, d_nvalid, ms, C3);
        KCHECK(c);
    }
    { ConvArgs a = conv_args(E.asp_tdnn_ms, ms, 2 * C3, ib, 128, items, false); if ((rc = launch_conv_gemm(c, a, "asp_ms"))) return rc; }
    { ConvArgs a = conv_args(E.asp_tdnn_x, mfa, C3, hid, 128, M, true); a.act1 = 1; a.act2 = 1; a.item_bias = ib; a.ib_ld = 128; WITH_LIST(a); if ((rc = launch_conv_gemm(c, a, "asp_tdnn"))) return rc; }
    float* logits = cat;   // cat is dead after mfa
    { ConvArgs a = conv_args(E.asp_conv, hid, 128, logits, C3, M, true); WITH_LIST(a); if ((rc = launch_conv_gemm(c, a, "asp_conv"))) return rc; }
    {
        ProfScope ps(c, "asp_pool", 0, (double)items * SD_T * C3 * 8.0);
        hipLaunchKernelGGL(k_asp_pool, dim3((C3 + 255) / 256, (unsigned)items), dim3(256), 0, st, mfa, logits, C3, d_nvalid, pooled, C3);
        KCHECK(c);
    }
    // asp_bn folded into fc
    { ConvArgs a = conv_args(E.fc, pooled, 2 * C3, d_emb, SD_EMB_DIM, items, false); if ((rc = launch_conv_gemm(c, a, "fc"))) return rc; }
    if (d_flags) {
        hipLaunchKernelGGL(k_nan_rows, GRID1(items * SD_EMB_DIM), 0, st, d_emb, d_flags, items);
        KCHECK(c);
    }
    return SD_OK;
}

int run_embed(sd_ctx* c, const float* d_wav, int64_t n, const float* d_masks, int64_t items, int64_t first_item, float* d_emb)
{
    if (items <= 0) return SD_OK;
    int64_t nb = c->emb_batch_items;
    nb = (nb / 96) * 96; if (nb < 96) nb = 96;
    int rc;
    // front end for the whole range first: items that are NaN by rule (sd.cpp:2479-2549) are dropped here, so the
    // network always runs on full batches of live items (the reference computes the dead ones and overwrites them)
    WS(c, float, feats, "emb_feats", items * SD_TP * SD_FEAT_LD);
    WS(c, float, lens, "emb_lens", items);
    WS(c, int, nnorm, "emb_nnorm", items);
    WS(c, int, nvalid, "emb_nvalid", items);
    WS(c, int, flags, "emb_flags", items);
    WS(c, int, cidx, "emb_cidx", items);
    WS(c, float, emb_c, "emb_compact", items * SD_EMB_DIM);
    int n_active = 0;
    if ((rc = run_frontend(c, d_wav, n, d_masks, items, first_item, feats, lens, nnorm, nvalid, flags, true, &n_active, cidx))) return rc;
    { KernelStat& ks = c->stats["items_live"]; ks.launches++; ks.flops += (double)n_active; ks.bytes += (double)items; }   // bench: live / all items
    std::vector<int> h_nvalid((size_t)(n_active > 0 ? n_active : 1));
    if (n_active > 0) {
        HIPCHK(c, hipMemcpyAsync(h_nvalid.data(), nvalid, (size_t)n_active * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    for (int64_t a0 = 0; a0 < n_active; a0 += nb) {
        const int64_t cnt = (n_active - a0 < nb) ? n_active - a0 : nb;
        if ((rc = run_ecapa(c, feats + (size_t)a0 * SD_TP * SD_FEAT_LD, nvalid + a0, nullptr, cnt, emb_c + (size_t)a0 * SD_EMB_DIM,
                            c->skip_dead_rows ? h_nvalid.data() + a0 : nullptr))) return rc;
    }
    hipLaunchKernelGGL(k_scatter_emb, GRID1(items * SD_EMB_DIM), 0, c->stream, emb_c, cidx, d_emb, items);
    KCHECK(c);
    return SD_OK;
}
