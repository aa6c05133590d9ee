// ecapa.hip -- ECAPA-TDNN (C=1024, attention 128, 192-d) body of the embedding model.
// Replaces EmbeddingModel1::_infer's Ort::Session::Run on emd4.onnx (sd.cpp:1889-1970);
// architecture = speechbrain 0.5.14 ECAPA_TDNN as exported by embeddings/export3.py:151-190.
//
// All dense contractions run through conv_gemm.hip (f32 MFMA).  This file holds the
// HBM-bound glue kernels (masked SE mean, SE apply + residual, ASP statistics and the
// attentive softmax pooling) and the layer schedule.  Activations are channels-last rows
// [compact row][C]: item after item, only the frames that can influence a valid output --
// need_i = min(501, nvalid_i + 65) rows of item i (frames >= nvalid are excluded from every
// statistic, and a frame only sees 65 frames to either side through the whole network), so a
// batch is one dense row space without dead panels; rowoff[i] is item i's first row.
#include "common.h"
#include <algorithm>

// masked mean over the first nvalid[item] frames  -> out[item][C]   (SEBlock, lengths given)
template <class T> __global__ void k_masked_mean(const T* __restrict__ x, int ld, const int* __restrict__ nvalid, const int* __restrict__ rowoff, int row_base,
                              float* __restrict__ out, int C)
{
    const int item = blockIdx.y, ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C) return;
    const int nv = nvalid[item];
    const T* p = x + (size_t)(rowoff[item] - row_base) * ld + ch;
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int t = 0;
    for (; t + 3 < nv; t += 4) { s0 += (float)p[(size_t)t * ld]; s1 += (float)p[(size_t)(t + 1) * ld]; s2 += (float)p[(size_t)(t + 2) * ld]; s3 += (float)p[(size_t)(t + 3) * ld]; }
    for (; t < nv; ++t) s0 += (float)p[(size_t)t * ld];
    out[(size_t)item * C + ch] = ((s0 + s1) + (s2 + s3)) / (float)nv;
}

// y = gate[item][c] * t2 + residual   (SERes2NetBlock tail), float4 over channels
template <class T> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
template <> struct Vec4<_Float16> { typedef _Float16 type __attribute__((ext_vector_type(4))); };
template <class T> __device__ __forceinline__ float4 ld4(const T* p)
{
    if constexpr (sizeof(T) == 4) return *(const float4*)p;
    else { const typename Vec4<_Float16>::type h = *(const typename Vec4<_Float16>::type*)p; return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]); }
}
template <class T> __device__ __forceinline__ void st4(T* p, float4 v)
{
    if constexpr (sizeof(T) == 4) *(float4*)p = v;
    else { typename Vec4<_Float16>::type h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; *(typename Vec4<_Float16>::type*)p = h; }
}
template <class T> __global__ void k_se_apply(const T* __restrict__ t2, const float* __restrict__ gate, const T* __restrict__ res, int res_ld,
                           T* __restrict__ y, int y_ld, int C, int64_t rows, const int2* __restrict__ rowtab)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4n = C / 4;
    if (idx >= rows * c4n) return;
    const int64_t row = idx / c4n;
    const int c = (int)(idx - row * c4n) * 4;
    const int64_t item = ROWTAB_ITEM(rowtab[row].y);
    const float4 a = ld4(t2 + row * C + c);
    const float4 g = *(const float4*)(gate + item * C + c);
    const float4 r = ld4(res + row * res_ld + c);
    float4 o;
    o.x = g.x * a.x + r.x; o.y = g.y * a.y + r.y; o.z = g.z * a.z + r.z; o.w = g.w * a.w + r.w;
    st4(y + row * y_ld + c, o);
}

// copy a channel slice [rows][w] between strided buffers (Res2Net first sub-band is identity)
template <class T> __global__ void k_copy_slice(const T* __restrict__ src, int src_ld, T* __restrict__ dst, int dst_ld, int w, int64_t rows)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int w4 = w / 4;
    if (idx >= rows * w4) return;
    const int64_t row = idx / w4;
    const int c = (int)(idx - row * w4) * 4;
    *(typename Vec4<T>::type*)(dst + row * dst_ld + c) = *(const typename Vec4<T>::type*)(src + row * src_ld + c);
}

// ASP global-context statistics: mean / std over valid frames -> ms[item][2C]
// one pass (Welford) with 4 rows in flight per thread: the tensor is read once
template <class T> __global__ void k_asp_stats(const T* __restrict__ x, int ld, const int* __restrict__ nvalid, const int* __restrict__ rowoff, int row_base,
                            float* __restrict__ ms, int C)
{
    const int item = blockIdx.y, ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C) return;
    const int nv = nvalid[item];
    const T* p = x + (size_t)(rowoff[item] - row_base) * ld + ch;
    float mean = 0.0f, m2 = 0.0f;
    int t = 0;
    for (; t + 3 < nv; t += 4) {
        const float v0 = (float)p[(size_t)t * ld], v1 = (float)p[(size_t)(t + 1) * ld], v2 = (float)p[(size_t)(t + 2) * ld], v3 = (float)p[(size_t)(t + 3) * ld];
        float d;
        d = v0 - mean; mean += d / (float)(t + 1); m2 += d * (v0 - mean);
        d = v1 - mean; mean += d / (float)(t + 2); m2 += d * (v1 - mean);
        d = v2 - mean; mean += d / (float)(t + 3); m2 += d * (v2 - mean);
        d = v3 - mean; mean += d / (float)(t + 4); m2 += d * (v3 - mean);
    }
    for (; t < nv; ++t) { const float v = (float)p[(size_t)t * ld]; const float d = v - mean; mean += d / (float)(t + 1); m2 += d * (v - mean); }
    ms[(size_t)item * 2 * C + ch] = mean;
    ms[(size_t)item * 2 * C + C + ch] = sqrtf(fmaxf(m2 / (float)nv, 1e-12f));
}

// attentive statistics pooling: masked softmax over time of the logits, weighted mean/std of x.
// One pass over both tensors: online softmax (running max, rescaled weights) + weighted Welford update.
template <class T> __global__ void k_asp_pool(const T* __restrict__ x, const float* __restrict__ logit, int ld, const int* __restrict__ nvalid,
                           const int* __restrict__ rowoff, int row_base, float* __restrict__ pooled, int C)
{
    const int item = blockIdx.y, ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C) return;
    const int nv = nvalid[item];
    const T* px = x + (size_t)(rowoff[item] - row_base) * ld + ch;
    const float* pl = logit + (size_t)(rowoff[item] - row_base) * ld + ch;
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
        const float l0 = (float)pl[(size_t)t * ld], l1 = (float)pl[(size_t)(t + 1) * ld], l2 = (float)pl[(size_t)(t + 2) * ld], l3 = (float)pl[(size_t)(t + 3) * ld];
        const float v0 = (float)px[(size_t)t * ld], v1 = (float)px[(size_t)(t + 1) * ld], v2 = (float)px[(size_t)(t + 2) * ld], v3 = (float)px[(size_t)(t + 3) * ld];
        step(l0, v0); step(l1, v1); step(l2, v2); step(l3, v3);
    }
    for (; t < nv; ++t) step((float)pl[(size_t)t * ld], (float)px[(size_t)t * ld]);
    pooled[(size_t)item * 2 * C + ch] = mean;
    pooled[(size_t)item * 2 * C + C + ch] = sqrtf(fmaxf(m2 / W, 1e-12f));
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

// rowtab[g] of every row g of the OUTPUT space of the batch (see ConvArgs): first row / last stored frame of the item in the
// INPUT space; one workgroup per item
__global__ void k_build_rowtab(const int* __restrict__ rowoff_out, int base_out, const int* __restrict__ rowoff_in, int base_in, int2* __restrict__ rowtab)
{
    const int item = blockIdx.x;
    const int r0 = rowoff_out[item] - base_out, n_out = rowoff_out[item + 1] - rowoff_out[item];
    const int i0 = rowoff_in[item] - base_in, n_in = rowoff_in[item + 1] - rowoff_in[item];
    for (int t = threadIdx.x; t < n_out; t += blockDim.x) rowtab[r0 + t] = make_int2(i0, t | ((n_in - 1) << 10) | (item << 20));
}

// fp16 feature rows for the fp16 mode: [rows][96] f32 -> [rows][128] halves (block0's K-step is 64 halves)
__global__ void k_feats_to_half(const float* __restrict__ f, _Float16* __restrict__ h, int64_t rows)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * 128) return;
    const int64_t row = idx >> 7; const int c = (int)(idx & 127);
    h[idx] = (_Float16)(c < SD_FEAT_LD ? f[row * SD_FEAT_LD + c] : 0.0f);
}

// prec 0: f32 MFMA, float buffers.  prec 1 (ecapa_precision): fp16 MFMA, _Float16 buffers (X / Y leading dimensions in elements)
static ConvArgs conv_args(const ConvLayer& L, const void* X, int x_ld, void* Y, int y_ld, int64_t M, bool per_item, int prec = 0)
{
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.X = (const float*)X; a.x_ld = x_ld; a.Y = (float*)Y; a.y_ld = y_ld; a.W = L.W; a.W16 = L.W16;
    a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.M = (int)M;
    if (per_item) { a.TpIn = a.TpOut = SD_TP; a.Tin = a.T = SD_T; }
    else { a.TpIn = a.TpOut = (int)M; a.Tin = a.T = (int)M; }
    a.Cin = prec ? L.CinPad16 : L.CinPad; a.cin_real = L.Cin; a.Cout = L.Cout; a.KT = L.KT; a.dil = L.dil;
    a.w_ld = a.Cin;
    a.pad_mode = 0; a.prec = prec;
    return a;
}

#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)

// One-sided reach of the network: 2 (block0, k5) + 7 * (2 + 3 + 4) (each Res2Net block chains 7 k3 convs of dilation 2 / 3 / 4;
// the 1x1 convs and the SE gate add none) = 65 frames.  Frames >= nvalid are excluded from every statistic (SE mean, ASP),
// so frames at or beyond min(501, nvalid + 65) cannot influence the embedding and are not stored at all.
int ecapa_need_rows(int nvalid, bool skip_dead_rows)
{
    if (!skip_dead_rows) return SD_T;
    const int need = nvalid + 65;
    return need > SD_T ? SD_T : (need < 1 ? 1 : need);
}

// d_feats: compact rows [rows][96] of `items` items; d_rowoff[items + 1] (first compact row of every item, in a row space that
// starts at row_base for this batch); d_nvalid[items].  Two row spaces: the WIDE one (need_i = nvalid_i + receptive-field margin)
// carries block0 and the three SE-Res2Net blocks; MFA is a 1x1 layer whose output is only ever read at frames < nvalid (ASP
// statistics and pooling), so it reads wide rows and writes the NARROW space (nvalid_i rows per item, d_rowoffN / row_baseN /
// rowsN), in which the ASP layers then run: 17 % fewer rows for half of the network's FLOPs on the planted hour.
template <class T>
static int run_ecapa_t(sd_ctx* c, const float* d_feats, const int* d_nvalid, const int* d_rowoff, int row_base, const int* d_rowoffN, int row_baseN,
                       int64_t items, int64_t rows, int64_t rowsN, float* d_emb)
{
    constexpr int P = sizeof(T) == 2 ? 1 : 0;                  // conv_gemm precision of the per-frame layers
    const EcapaWeights& E = c->ew;
    if (!E.loaded) SD_FAIL(c, SD_ERR_MODEL, "embedding model not loaded");
    if (items <= 0) return SD_OK;
    if (items > ROWTAB_MAX_ITEMS) SD_FAIL(c, SD_ERR_ARG, "ecapa batch of %lld items (limit %d)", (long long)items, ROWTAB_MAX_ITEMS);
    const int C = E.C, C3 = 3 * C;
    const int64_t M = rows;
    if (M > 0x7fffffff / 2) SD_FAIL(c, SD_ERR_ARG, "ecapa batch too large");
    WS(c, T, x0, "ec_x0", M * C);
    WS(c, T, t1, "ec_t1", M * C);
    WS(c, T, rr, "ec_r", M * C);
    WS(c, T, t2, "ec_t2", M * C);
    WS(c, T, cat, "ec_cat", M * C3);
    const int64_t MN = rowsN;
    WS(c, T, mfa, "ec_mfa", MN * C3);
    WS(c, T, hid, "ec_hid", MN * 128);
    WS(c, float, se_s, "ec_se_s", items * C);
    WS(c, float, se_h, "ec_se_h", items * 128);
    WS(c, float, se_g, "ec_se_g", items * C);
    WS(c, float, ms, "ec_ms", items * 2 * C3);
    WS(c, float, ib, "ec_ib", items * 128);
    WS(c, float, pooled, "ec_pooled", items * 2 * C3);
    WS(c, int2, rowtab, "ec_rowtab", M + 128);
    WS(c, int2, rowtab_nw, "ec_rowtab_nw", MN + 128);          // narrow rows -> wide input (MFA)
    WS(c, int2, rowtab_n, "ec_rowtab_n", MN + 128);            // narrow -> narrow (ASP layers)
    int rc;
    hipStream_t st = c->stream;
    hipLaunchKernelGGL(k_build_rowtab, dim3((unsigned)items), dim3(256), 0, st, d_rowoff, row_base, d_rowoff, row_base, rowtab);
    hipLaunchKernelGGL(k_build_rowtab, dim3((unsigned)items), dim3(256), 0, st, d_rowoffN, row_baseN, d_rowoff, row_base, rowtab_nw);
    hipLaunchKernelGGL(k_build_rowtab, dim3((unsigned)items), dim3(256), 0, st, d_rowoffN, row_baseN, d_rowoffN, row_baseN, rowtab_n);
    KCHECK(c);
#define WITH_LIST(a) do { (a).rowtab = rowtab; } while (0)

    // blocks[0]: TDNNBlock(80 -> C, k5)
    const void* f_in = d_feats; int f_ld = SD_FEAT_LD;
    if (P) {
        WS(c, _Float16, fh, "ec_feats16", M * 128);
        hipLaunchKernelGGL(k_feats_to_half, GRID1(M * 128), 0, st, d_feats, fh, M);
        KCHECK(c);
        f_in = fh; f_ld = 128;
    }
    { ConvArgs a = conv_args(E.block0, f_in, f_ld, x0, C, M, true, P); a.act1 = 1; WITH_LIST(a); if ((rc = launch_conv_gemm(c, a, "block0"))) return rc; }

    for (int b = 0; b < 3; ++b) {
        const auto& B = E.blk[b];
        const T* xin = (b == 0) ? x0 : cat + (size_t)(b - 1) * C;
        const int xin_ld = (b == 0) ? C : C3;
        { ConvArgs a = conv_args(B.tdnn1, xin, xin_ld, t1, C, M, true, P); a.act1 = 1; WITH_LIST(a); if ((rc = launch_conv_gemm(c, a, "tdnn1"))) return rc; }
        const int S = C / 8;
        hipLaunchKernelGGL(k_copy_slice<T>, GRID1(M * (S / 4)), 0, st, t1, C, rr, C, S, M);
        KCHECK(c);
        for (int i = 1; i < 8; ++i) {
            ConvArgs a = conv_args(B.res[i - 1], t1 + i * S, C, rr + i * S, C, M, true, P);
            a.act1 = 1;
            if (i >= 2) { a.X2 = (const float*)(rr + (i - 1) * S); a.x2_ld = C; }
            WITH_LIST(a);
            if ((rc = launch_conv_gemm(c, a, "res2net"))) return rc;
        }
        { ConvArgs a = conv_args(B.tdnn2, rr, C, t2, C, M, true, P); a.act1 = 1; WITH_LIST(a); if ((rc = launch_conv_gemm(c, a, "tdnn2"))) return rc; }
        {
            ProfScope ps(c, "se_mean", 0, (double)M * C * 4.0);
            hipLaunchKernelGGL(k_masked_mean<T>, dim3((C + 255) / 256, (unsigned)items), dim3(256), 0, st, t2, C, d_nvalid, d_rowoff, row_base, se_s, C);
            KCHECK(c);
        }
        { ConvArgs a = conv_args(B.se1, se_s, C, se_h, 128, items, false); a.act1 = 1; if ((rc = launch_conv_gemm(c, a, "se1"))) return rc; }
        { ConvArgs a = conv_args(B.se2, se_h, 128, se_g, C, items, false); a.act2 = 2; if ((rc = launch_conv_gemm(c, a, "se2"))) return rc; }
        {
            ProfScope ps(c, "se_apply", 0, (double)M * C * 12.0);
            hipLaunchKernelGGL(k_se_apply<T>, GRID1(M * (C / 4)), 0, st, t2, se_g, xin, xin_ld, cat + (size_t)b * C, C3, C, M, rowtab);
            KCHECK(c);
        }
    }
    // mfa: TDNNBlock(3C -> 3C, k1) over cat(x1,x2,x3)
    { ConvArgs a = conv_args(E.mfa, cat, C3, mfa, C3, MN, true, P); a.act1 = 1; a.rowtab = rowtab_nw; a.in_rows = (int)M; if ((rc = launch_conv_gemm(c, a, "mfa"))) return rc; }
    // ASP with global context: cat[x, mean, std] @ W == x @ Wx + (mean,std) @ Wms  (per-item bias)
    {
        ProfScope ps(c, "asp_stats", 0, (double)MN * C3 * 4.0);
        hipLaunchKernelGGL(k_asp_stats<T>, dim3((C3 + 255) / 256, (unsigned)items), dim3(256), 0, st, mfa, C3, d_nvalid, d_rowoffN, row_baseN, ms, C3);
        KCHECK(c);
    }
    { ConvArgs a = conv_args(E.asp_tdnn_ms, ms, 2 * C3, ib, 128, items, false); if ((rc = launch_conv_gemm(c, a, "asp_ms"))) return rc; }
    { ConvArgs a = conv_args(E.asp_tdnn_x, mfa, C3, hid, 128, MN, true, P); a.act1 = 1; a.act2 = 1; a.item_bias = ib; a.ib_ld = 128; a.rowtab = rowtab_n; if ((rc = launch_conv_gemm(c, a, "asp_tdnn"))) return rc; }
    // attention logits stay f32 in either mode (they feed an exp: fp16's 3 decimal digits at |logit| ~ 30 would be percents of a weight).
    // f32 mode: cat is dead after mfa and large enough; fp16 mode: its own buffer
    float* logits;
    if (P) { WS(c, float, lg, "ec_logits", MN * C3); logits = lg; } else logits = (float*)cat;
    { ConvArgs a = conv_args(E.asp_conv, hid, 128, logits, C3, MN, true, P); a.rowtab = rowtab_n; a.y_f32 = 1; if ((rc = launch_conv_gemm(c, a, "asp_conv"))) return rc; }
    {
        ProfScope ps(c, "asp_pool", 0, (double)MN * C3 * 8.0);
        hipLaunchKernelGGL(k_asp_pool<T>, dim3((C3 + 255) / 256, (unsigned)items), dim3(256), 0, st, mfa, logits, C3, d_nvalid, d_rowoffN, row_baseN, pooled, C3);
        KCHECK(c);
    }
    // asp_bn folded into fc
    { ConvArgs a = conv_args(E.fc, pooled, 2 * C3, d_emb, SD_EMB_DIM, items, false); if ((rc = launch_conv_gemm(c, a, "fc"))) return rc; }
    return SD_OK;
}

int run_ecapa(sd_ctx* c, const float* d_feats, const int* d_nvalid, const int* d_rowoff, int row_base, const int* d_rowoffN, int row_baseN,
              int64_t items, int64_t rows, int64_t rowsN, float* d_emb)
{
    if (c->ecapa_precision == 1) return run_ecapa_t<_Float16>(c, d_feats, d_nvalid, d_rowoff, row_base, d_rowoffN, row_baseN, items, rows, rowsN, d_emb);
    return run_ecapa_t<float>(c, d_feats, d_nvalid, d_rowoff, row_base, d_rowoffN, row_baseN, items, rows, rowsN, d_emb);
}

// host side of the compact row plan: need / rowoff of `n` items from their nvalid (uploaded to d_rowoff[n + 1])
// rowoff: wide space (need rows per item), rowoffN: narrow space (nvalid rows per item; = wide when nothing is skipped)
int ecapa_row_plan(sd_ctx* c, const int* h_nvalid, int64_t n, std::vector<int>& rowoff, std::vector<int>& rowoffN, int* d_rowoff)
{
    rowoff.assign((size_t)n + 1, 0); rowoffN.assign((size_t)n + 1, 0);
    int64_t acc = 0, accN = 0;
    for (int64_t i = 0; i < n; ++i) {
        rowoff[(size_t)i] = (int)acc; rowoffN[(size_t)i] = (int)accN;
        const int need = ecapa_need_rows(h_nvalid[i], c->skip_dead_rows);
        int nv = h_nvalid[i]; if (nv > need) nv = need; if (nv < 1) nv = 1;
        acc += need; accN += c->skip_dead_rows ? nv : need;
    }
    if (acc > 0x7fffffff / 4) SD_FAIL(c, SD_ERR_ARG, "embedding stage: %lld feature rows in one shard (limit %d)", (long long)acc, 0x7fffffff / 4);
    rowoff[(size_t)n] = (int)acc; rowoffN[(size_t)n] = (int)accN;
    HIPCHK(c, hipMemcpyAsync(d_rowoff, rowoff.data(), (size_t)(n + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_rowoff + n + 1, rowoffN.data(), (size_t)(n + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
    WS(c, float, lens, "emb_lens", items);
    WS(c, int, nnorm, "emb_nnorm", items);
    WS(c, int, nvalid, "emb_nvalid", items);
    WS(c, int, flags, "emb_flags", items);
    WS(c, int, cidx, "emb_cidx", items);
    WS(c, int, d_rowoff, "emb_rowoff", 2 * (items + 1));
    WS(c, float, emb_c, "emb_compact", items * SD_EMB_DIM);
    int n_active = 0;
    if ((rc = frontend_prepare(c, d_masks, items, first_item, lens, nnorm, nvalid, flags, true, &n_active, cidx))) return rc;
    { KernelStat& ks = c->stats["items_live"]; ks.launches++; ks.flops += (double)n_active; ks.bytes += (double)items; }   // bench: live / all items
    if (n_active > 0) {
        std::vector<int> h_nvalid((size_t)n_active), rowoff, rowoffN;
        HIPCHK(c, hipMemcpyAsync(h_nvalid.data(), nvalid, (size_t)n_active * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if ((rc = ecapa_row_plan(c, h_nvalid.data(), n_active, rowoff, rowoffN, d_rowoff))) return rc;
        const int* d_rowoffN = d_rowoff + n_active + 1;
        const int64_t rows_all = rowoff[(size_t)n_active];
        WS(c, float, feats, "emb_feats", rows_all * SD_FEAT_LD);
        if ((rc = frontend_features(c, d_wav, n, first_item, n_active, true, nnorm, d_rowoff, feats))) return rc;
        // batches by row budget: as many whole items as fit the activation workspaces of nb full-length items
        const int64_t cap_rows = nb * SD_TP;
        for (int64_t a0 = 0; a0 < n_active;) {
            int64_t a1 = a0;
            while (a1 < n_active && a1 - a0 < ROWTAB_MAX_ITEMS && rowoff[(size_t)a1 + 1] - rowoff[(size_t)a0] <= cap_rows) ++a1;
            if (a1 == a0) a1 = a0 + 1;
            const int base = rowoff[(size_t)a0], baseN = rowoffN[(size_t)a0];
            if ((rc = run_ecapa(c, feats + (size_t)base * SD_FEAT_LD, nvalid + a0, d_rowoff + a0, base, d_rowoffN + a0, baseN, a1 - a0,
                                rowoff[(size_t)a1] - base, rowoffN[(size_t)a1] - baseN, emb_c + (size_t)a0 * SD_EMB_DIM))) return rc;
            a0 = a1;
        }
    }
    hipLaunchKernelGGL(k_scatter_emb, GRID1(items * SD_EMB_DIM), 0, c->stream, emb_c, cidx, d_emb, items);
    KCHECK(c);
    return SD_OK;
}
