// weights.cpp -- ".sdw" weight-pack reader and conversion into device layouts.
// Replaces the model-loading half of OnnxModel::OnnxModel (onnx_model.cc:41-105):
// the reference hands the file to Ort::Session; here the tensors are re-laid out
// for the HIP kernels (conv weights [tap][Cout][CinPad], BatchNorm folded to
// scale/shift, LSTM biases pre-summed, sparse mel filters).
#include "common.h"
#include <cmath>

int load_pack(const char* path, Pack& out, std::string& err)
{
    FILE* f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open model file: ") + path; return SD_ERR_MODEL; }
    char magic[4];
    uint32_t cnt = 0;
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "SDW1", 4) != 0 || fread(&cnt, 4, 1, f) != 1) {
        fclose(f); err = std::string("not an SDW1 weight pack: ") + path; return SD_ERR_MODEL;
    }
    for (uint32_t i = 0; i < cnt; ++i) {
        uint16_t nl = 0; uint8_t nd = 0;
        if (fread(&nl, 2, 1, f) != 1) goto bad;
        {
            std::string name(nl, '\0');
            if (fread(&name[0], 1, nl, f) != nl) goto bad;
            if (fread(&nd, 1, 1, f) != 1) goto bad;
            PackTensor t;
            t.dims.resize(nd);
            if (nd && fread(t.dims.data(), 8, nd, f) != nd) goto bad;
            int64_t n = 1;
            for (auto d : t.dims) n *= d;
            if (n < 0 || n > (int64_t)1 << 32) goto bad;
            t.data.resize((size_t)n);
            if (n && fread(t.data.data(), 4, (size_t)n, f) != (size_t)n) goto bad;
            out[name] = std::move(t);
        }
    }
    fclose(f);
    return SD_OK;
bad:
    fclose(f);
    err = std::string("truncated weight pack: ") + path;
    return SD_ERR_MODEL;
}

static const PackTensor* need(sd_ctx* c, const Pack& p, const std::string& name)
{
    auto it = p.find(name);
    if (it == p.end()) { c->err = "weight pack lacks tensor '" + name + "'"; return nullptr; }
    return &it->second;
}

// weights live in a few large device blocks handed out by a bump allocator (one hipMalloc per 64 MB instead of one per tensor:
// ~250 tensors per context, and hipMalloc is what start-up spends its time in once the layouts are cheap); 256-byte aligned, 16 spare
// bytes behind every tensor (the kernels' 16-byte loads may touch the tail of a row)
void* weight_alloc(sd_ctx* c, size_t bytes)
{
    const size_t need = (bytes + 16 + 255) & ~(size_t)255;
    if (need > c->warena_left) {
        const size_t block = need > ((size_t)64 << 20) ? need : ((size_t)64 << 20);
        void* d = nullptr;
        if (hipMalloc(&d, block) != hipSuccess) { c->err = "hipMalloc(weights) failed"; return nullptr; }
        c->owned.push_back(d);
        c->warena_cur = (char*)d; c->warena_left = block;
    }
    void* out = c->warena_cur;
    c->warena_cur += need; c->warena_left -= need;
    return out;
}
template <class T> static T* upload(sd_ctx* c, const std::vector<T>& h)
{
    T* d = (T*)weight_alloc(c, h.size() * sizeof(T));
    if (!d) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { c->err = "hipMemcpy(weights) failed"; return nullptr; }
    return d;
}

// ecapa_precision = 3 (conv_gemm_h.hip P = 3, conv_gemm.hip PR = 3): hi / lo fp16 halves of W * 2^e, interleaved in groups of eight channels the
// way the kernels stage them -- per 32-channel chunk of a row [hi 0..7 | lo 0..7 | hi 8..15 | lo 8..15 | hi 16..23 | lo 16..23 | hi 24..31 | lo 24..31].
// 2^e puts the layer's largest weight into [2^13, 2^14), so that the lo half of every weight down to 2^-17 of it is a normal fp16 number
// (unscaled, the residue of a weight of 0.02 is a subnormal with 7 significant bits); the epilogue multiplies the accumulator by 2^-e.
static void pack_split_weights(const float* hw, int K, int Cout, int CinPad, int cin, _Float16* out, float* inv_scale)
{
    float wmax = 0.0f;
    for (size_t r = 0; r < (size_t)K * Cout; ++r)
        for (int i = 0; i < cin; ++i) { const float v = hw[r * CinPad + i]; if (std::isfinite(v)) wmax = fmaxf(wmax, fabsf(v)); }
    int e = 0;
    if (wmax > 0.0f) { (void)frexpf(wmax, &e); e = 14 - e; }
    const float sc = ldexpf(1.0f, e);
    *inv_scale = ldexpf(1.0f, -e);
    for (size_t q = 0; q < (size_t)2 * K * Cout * CinPad; ++q) out[q] = (_Float16)0.0f;
    for (int k = 0; k < K; ++k)
        for (int o = 0; o < Cout; ++o)
            for (int i = 0; i < cin; ++i) {
                const float v = hw[((size_t)k * Cout + o) * CinPad + i] * sc;
                const _Float16 hi = (_Float16)v;
                const size_t at = ((size_t)k * Cout + o) * 2 * CinPad + (size_t)(i / 32) * 64 + (size_t)((i % 32) / 8) * 16 + (i % 8);
                out[at] = hi;
                out[at + 8] = (_Float16)(v - (float)hi);
            }
}

// test hook (sdhip_test.h): the packing above on host buffers, no GPU
extern "C" int sd_test_pack_split_weights(const float* w, int K, int Cout, int CinPad, int cin, uint16_t* out_halves, float* inv_scale)
{
    if (!w || !out_halves || !inv_scale || K <= 0 || Cout <= 0 || CinPad <= 0 || CinPad % 32 || cin <= 0 || cin > CinPad) return SD_ERR_ARG;
    pack_split_weights(w, K, Cout, CinPad, cin, (_Float16*)out_halves, inv_scale);
    return SD_OK;
}

// PyTorch conv weight [Cout][Cin][K] (or linear [Cout][Cin]) -> [K][Cout][CinPad]
// optional input-channel slice [ci0, ci0+cin) of the source tensor
static int make_conv(sd_ctx* c, const Pack& p, const std::string& wname, const std::string& bname,
                     const std::string& bnprefix, int dil, ConvLayer& L, int ci0 = 0, int cin = -1,
                     const std::vector<float>* in_scale = nullptr, const std::vector<float>* in_shift = nullptr, bool want16 = false, int out_rot = 0)
{
    // out_rot: output channel o of the layer built here is channel (o + out_rot) mod Cout of the model's layer (a rotation of the
    // output channels: ecapa.hip stores tdnn1's sub-band 0 LAST, next to the Res2Net outputs, so that tdnn2 reads one contiguous row)
    const PackTensor* w = need(c, p, wname);
    if (!w) return SD_ERR_MODEL;
    if (w->dims.size() < 2 || w->dims.size() > 3) SD_FAIL(c, SD_ERR_MODEL, "tensor '%s' has rank %zu (conv / linear weights are [Cout][Cin] or [Cout][Cin][K])", wname.c_str(), w->dims.size());
    const int Cout = (int)w->dims[0];
    const int CinAll = (int)w->dims[1];
    const int K = w->dims.size() > 2 ? (int)w->dims[2] : 1;
    if (cin < 0) cin = CinAll - ci0;
    const int CinPad = (cin + 31) / 32 * 32;
    std::vector<float> hw((size_t)K * Cout * CinPad, 0.0f);
    std::vector<float> hb(Cout, 0.0f);
    bool has_bias = false;
    if (!bname.empty()) {
        const PackTensor* b = need(c, p, bname);
        if (!b) return SD_ERR_MODEL;
        for (int o = 0; o < Cout; ++o) hb[o] = b->data[(o + out_rot) % Cout];
        has_bias = true;
    }
    for (int o = 0; o < Cout; ++o)
        for (int i = 0; i < cin; ++i)
            for (int k = 0; k < K; ++k) {
                float v = w->data[((size_t)((o + out_rot) % Cout) * CinAll + (ci0 + i)) * K + k];
                if (in_scale) {           // fold an affine on the INPUT (BatchNorm before a 1x1 conv)
                    hb[o] += v * (*in_shift)[i];
                    v *= (*in_scale)[i];
                    has_bias = true;
                }
                hw[((size_t)k * Cout + o) * CinPad + i] = v;
            }
    L.W = upload(c, hw);
    if (!L.W) return SD_ERR_HIP;
    L.W16 = nullptr; L.W16x = nullptr;
    L.CinPad16 = (cin + 63) / 64 * 64;
    if (want16) c->ew.conv16.push_back(&L);          // fp16 forms: built on the GPU when a mode that reads them is selected (below)
    L.bias = nullptr;
    if (has_bias) { L.bias = upload(c, hb); if (!L.bias) return SD_ERR_HIP; }
    L.scale = L.shift = nullptr;
    if (!bnprefix.empty()) {
        const PackTensor *g = need(c, p, bnprefix + ".weight"), *be = need(c, p, bnprefix + ".bias"),
                         *mu = need(c, p, bnprefix + ".running_mean"), *var = need(c, p, bnprefix + ".running_var");
        if (!g || !be || !mu || !var) return SD_ERR_MODEL;
        std::vector<float> sc(Cout), sh(Cout);
        for (int o = 0; o < Cout; ++o) {
            const int so = (o + out_rot) % Cout;
            float s = g->data[so] / sqrtf(var->data[so] + 1e-5f);
            sc[o] = s; sh[o] = be->data[so] - mu->data[so] * s;
        }
        L.scale = upload(c, sc); L.shift = upload(c, sh);
        if (!L.scale || !L.shift) return SD_ERR_HIP;
    }
    L.Cin = cin; L.CinPad = CinPad; L.Cout = Cout; L.KT = K; L.dil = dil;
    return SD_OK;
}

int build_ecapa_weights(sd_ctx* c, const Pack& p)
{
    EcapaWeights& E = c->ew;
    // --- mel filterbank [201][80] -> per-mel contiguous non-zero range (exact zeros skipped)
    const PackTensor* mel = need(c, p, "fbank.matrix");
    if (!mel) return SD_ERR_MODEL;
    if (mel->dims.size() != 2 || mel->dims[0] != SD_NBINS || mel->dims[1] != SD_NMELS) SD_FAIL(c, SD_ERR_MODEL, "fbank.matrix must be [201][80]");
    std::vector<int> lo(SD_NMELS), cnt(SD_NMELS), off(SD_NMELS);
    std::vector<float> mw;
    for (int m = 0; m < SD_NMELS; ++m) {
        int a = SD_NBINS, b = -1;
        for (int k = 0; k < SD_NBINS; ++k) if (mel->data[(size_t)k * SD_NMELS + m] != 0.0f) { if (k < a) a = k; if (k > b) b = k; }
        if (b < a) { a = 0; b = -1; }
        lo[m] = a; cnt[m] = b - a + 1; off[m] = (int)mw.size();
        for (int k = a; k <= b; ++k) mw.push_back(mel->data[(size_t)k * SD_NMELS + m]);
    }
    if (mw.empty()) mw.push_back(0.0f);
    E.mel_nnz = (int)mw.size();
    E.mel_w = upload(c, mw); E.mel_lo = upload(c, lo); E.mel_cnt = upload(c, cnt); E.mel_off = upload(c, off);
    // --- STFT window (fp32 periodic Hamming, sd.cpp:2007) and fp64 twiddles
    std::vector<float> win(400);
    auto itw = p.find("stft.window");
    for (int n = 0; n < 400; ++n)
        // torch::hamming_window (periodic) evaluates arange * float(2 pi / N) -> cos -> * -0.46 -> + 0.54 in float32
        win[n] = (itw != p.end() && itw->second.data.size() == 400) ? itw->second.data[n]
                                                                      : cosf((float)n * (float)(2.0 * M_PI / 400.0)) * (-0.46f) + 0.54f;
    E.window = upload(c, win);
    std::vector<double> tc(400), ts(400);
    for (int k = 0; k < 400; ++k) { tc[k] = cos(2.0 * M_PI * k / 400.0); ts[k] = -sin(2.0 * M_PI * k / 400.0); }
    // exact values at the quarter points
    tc[0] = 1; ts[0] = 0; tc[100] = 0; ts[100] = -1; tc[200] = -1; ts[200] = 0; tc[300] = 0; ts[300] = 1;
    E.tw_cos = upload(c, tc); E.tw_nsin = upload(c, ts);
    if (!E.mel_w || !E.mel_lo || !E.mel_cnt || !E.mel_off || !E.window || !E.tw_cos || !E.tw_nsin) return SD_ERR_HIP;

    int rc;
    if ((rc = make_conv(c, p, "blocks.0.conv.weight", "blocks.0.conv.bias", "blocks.0.norm", 1, E.block0, 0, -1, nullptr, nullptr, true))) return rc;
    E.C = E.block0.Cout;
    const int dils[3] = {2, 3, 4};
    for (int b = 0; b < 3; ++b) {
        std::string pre = "blocks." + std::to_string(b + 1);
        auto& B = E.blk[b];
        B.dil = dils[b];
        if ((rc = make_conv(c, p, pre + ".tdnn1.conv.weight", pre + ".tdnn1.conv.bias", pre + ".tdnn1.norm", 1, B.tdnn1, 0, -1, nullptr, nullptr, true, E.C / 8))) return rc;      // sub-bands 1..7 first, sub-band 0 last
        for (int i = 0; i < 7; ++i) {
            std::string q = pre + ".res2net." + std::to_string(i);
            if ((rc = make_conv(c, p, q + ".conv.weight", q + ".conv.bias", q + ".norm", dils[b], B.res[i], 0, -1, nullptr, nullptr, true))) return rc;
        }
        if ((rc = make_conv(c, p, pre + ".tdnn2.conv.weight", pre + ".tdnn2.conv.bias", pre + ".tdnn2.norm", 1, B.tdnn2, 0, -1, nullptr, nullptr, true))) return rc;
        if ((rc = make_conv(c, p, pre + ".se.conv1.weight", pre + ".se.conv1.bias", "", 1, B.se1))) return rc;
        if ((rc = make_conv(c, p, pre + ".se.conv2.weight", pre + ".se.conv2.bias", "", 1, B.se2))) return rc;
    }
    if ((rc = make_conv(c, p, "mfa.conv.weight", "mfa.conv.bias", "mfa.norm", 1, E.mfa, 0, -1, nullptr, nullptr, true))) return rc;
    const int C3 = E.mfa.Cout;
    // ASP tdnn over cat[x, mean, std]: split into the x part and the (mean,std) part
    if ((rc = make_conv(c, p, "asp.tdnn.conv.weight", "asp.tdnn.conv.bias", "asp.tdnn.norm", 1, E.asp_tdnn_x, 0, C3, nullptr, nullptr, true))) return rc;
    if ((rc = make_conv(c, p, "asp.tdnn.conv.weight", "", "", 1, E.asp_tdnn_ms, C3, 2 * C3))) return rc;
    if ((rc = make_conv(c, p, "asp.conv.weight", "asp.conv.bias", "", 1, E.asp_conv, 0, -1, nullptr, nullptr, true))) return rc;
    // asp_bn (eval BatchNorm on the pooled vector) folded into fc
    {
        const PackTensor *g = need(c, p, "asp_bn.weight"), *be = need(c, p, "asp_bn.bias"),
                         *mu = need(c, p, "asp_bn.running_mean"), *var = need(c, p, "asp_bn.running_var");
        if (!g || !be || !mu || !var) return SD_ERR_MODEL;
        const int n = (int)g->data.size();
        std::vector<float> sc(n), sh(n);
        for (int i = 0; i < n; ++i) { float s = g->data[i] / sqrtf(var->data[i] + 1e-5f); sc[i] = s; sh[i] = be->data[i] - mu->data[i] * s; }
        if ((rc = make_conv(c, p, "fc.weight", "fc.bias", "", 1, E.fc, 0, -1, &sc, &sh))) return rc;
    }
    E.loaded = true;
    return SD_OK;
}

int build_seg_weights(sd_ctx* c, const Pack& p)
{
    SegWeights& S = c->sw;
    const PackTensor *ww = need(c, p, "sincnet.wav_norm.weight"), *wb = need(c, p, "sincnet.wav_norm.bias");
    if (!ww || !wb) return SD_ERR_MODEL;
    S.wn_w = ww->data[0]; S.wn_b = wb->data[0];
    // conv0: [80][1][251] -> GEMM weight [1][80][256] over taps (x_ld = 10 gives the stride-10 windows)
    {
        const PackTensor* w = need(c, p, "sincnet.conv0.weight");
        if (!w) return SD_ERR_MODEL;
        if (w->dims.size() != 3) SD_FAIL(c, SD_ERR_MODEL, "sincnet.conv0.weight must be [Cout][1][K]");
        const int Cout = (int)w->dims[0], K = (int)w->dims[2];
        if (K > 256) SD_FAIL(c, SD_ERR_MODEL, "sincnet.conv0 kernel %d > 256", K);
        std::vector<float> hw((size_t)Cout * 256, 0.0f);
        for (int o = 0; o < Cout; ++o) for (int k = 0; k < K; ++k) hw[(size_t)o * 256 + k] = w->data[(size_t)o * K + k];
        S.conv0.W = upload(c, hw);
        if (!S.conv0.W) return SD_ERR_HIP;
        std::vector<float> ws((size_t)Cout);
        for (int o = 0; o < Cout; ++o) { double a = 0.0; for (int k = 0; k < K; ++k) a += (double)w->data[(size_t)o * K + k]; ws[(size_t)o] = (float)a; }
        S.conv0_wsum = upload(c, ws);
        if (!S.conv0_wsum) return SD_ERR_HIP;
        S.conv0.Cin = K; S.conv0.CinPad = 256; S.conv0.Cout = Cout; S.conv0.KT = 1; S.conv0.dil = 1;
    }
    int rc;
    if ((rc = make_conv(c, p, "sincnet.conv1.weight", "sincnet.conv1.bias", "", 1, S.conv1))) return rc;
    if ((rc = make_conv(c, p, "sincnet.conv2.weight", "sincnet.conv2.bias", "", 1, S.conv2))) return rc;
    for (int i = 0; i < 3; ++i) {
        const PackTensor *g = need(c, p, "sincnet.norm" + std::to_string(i) + ".weight"), *b = need(c, p, "sincnet.norm" + std::to_string(i) + ".bias");
        if (!g || !b) return SD_ERR_MODEL;
        S.in_w[i] = upload(c, g->data); S.in_b[i] = upload(c, b->data);
        if (!S.in_w[i] || !S.in_b[i]) return SD_ERR_HIP;
    }
    for (int l = 0; l < 4; ++l) {
        const std::string sfx[2] = {"", "_reverse"};
        const PackTensor* wih[2]; const PackTensor* bih[2]; const PackTensor* bhh[2];
        for (int d = 0; d < 2; ++d) {
            wih[d] = need(c, p, "lstm.weight_ih_l" + std::to_string(l) + sfx[d]);
            bih[d] = need(c, p, "lstm.bias_ih_l" + std::to_string(l) + sfx[d]);
            bhh[d] = need(c, p, "lstm.bias_hh_l" + std::to_string(l) + sfx[d]);
            const PackTensor* whh = need(c, p, "lstm.weight_hh_l" + std::to_string(l) + sfx[d]);
            if (!wih[d] || !bih[d] || !bhh[d] || !whh) return SD_ERR_MODEL;
            S.lstm_hh[l][d] = upload(c, whh->data);
            if (!S.lstm_hh[l][d]) return SD_ERR_HIP;
            if (whh->data.size() == (size_t)512 * 128) {        // option seg_precision = 3: hi and lo planes of W_hh * 2^e (k_lstm_rec_x3)
                float wmax = 0.0f;
                for (float v : whh->data) if (std::isfinite(v)) wmax = fmaxf(wmax, fabsf(v));
                int e = 0;
                if (wmax > 0.0f) { (void)frexpf(wmax, &e); e = 14 - e; }
                const float sc = ldexpf(1.0f, e);
                std::vector<_Float16> hx((size_t)2 * 512 * 128);
                for (size_t q = 0; q < (size_t)512 * 128; ++q) {
                    const float v = whh->data[q] * sc;
                    const _Float16 hi = (_Float16)v;
                    hx[q] = hi; hx[(size_t)512 * 128 + q] = (_Float16)(v - (float)hi);
                }
                S.lstm_hh_x[l][d] = upload(c, hx);
                if (!S.lstm_hh_x[l][d]) return SD_ERR_HIP;
                S.lstm_hh_inv[l][d] = ldexpf(1.0f, -e);
            }
        }
        const int nin = (int)wih[0]->dims[1];
        const int pad = (nin + 31) / 32 * 32;
        std::vector<float> hw((size_t)1024 * pad, 0.0f), hb(1024);
        for (int d = 0; d < 2; ++d)
            for (int g = 0; g < 512; ++g) {
                for (int i = 0; i < nin; ++i) hw[((size_t)d * 512 + g) * pad + i] = wih[d]->data[(size_t)g * nin + i];
                hb[d * 512 + g] = bih[d]->data[g] + bhh[d]->data[g];
            }
        ConvLayer& L = S.lstm_ih[l];
        L.W = upload(c, hw); L.bias = upload(c, hb);
        if (!L.W || !L.bias) return SD_ERR_HIP;
        L.Cin = nin; L.CinPad = pad; L.Cout = 1024; L.KT = 1; L.dil = 1;
        {       // option seg_precision = 3: split weights of the input projection (the wide conv kernel's x3 form)
            std::vector<_Float16> hx((size_t)2 * 1024 * pad);
            pack_split_weights(hw.data(), 1, 1024, pad, nin, hx.data(), &L.w16x_inv);
            L.W16x = upload(c, hx);
            if (!L.W16x) return SD_ERR_HIP;
        }
    }
    if ((rc = make_conv(c, p, "linear.0.weight", "linear.0.bias", "", 1, S.lin0))) return rc;
    if ((rc = make_conv(c, p, "linear.1.weight", "linear.1.bias", "", 1, S.lin1))) return rc;
    const PackTensor *cw = need(c, p, "classifier.weight"), *cb = need(c, p, "classifier.bias");
    if (!cw || !cb) return SD_ERR_MODEL;
    S.cls_w = upload(c, cw->data); S.cls_b = upload(c, cb->data);
    if (!S.cls_w || !S.cls_b) return SD_ERR_HIP;
    S.loaded = true;
    return SD_OK;
}
