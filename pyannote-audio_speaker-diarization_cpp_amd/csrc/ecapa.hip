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
// rows = rows of t2's space; the residual and the output live in other (larger) spaces: their row of (item, t) is
// map[row].x + t of the space table (nullptr = the same space as t2)
template <class T> __global__ void k_se_apply(const T* __restrict__ t2, int t2_ld, const float* __restrict__ gate, const T* __restrict__ res, int res_ld,
                           T* __restrict__ y, int y_ld, int C, int64_t rows, const int2* __restrict__ rowtab,
                           const int2* __restrict__ res_map, const int2* __restrict__ out_map)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4n = C / 4;
    if (idx >= rows * c4n) return;
    const int64_t row = idx / c4n;
    const int c = (int)(idx - row * c4n) * 4;
    const int64_t item = ROWTAB_ITEM(rowtab[row].y);
    int64_t rrow = row, orow = row;
    if (res_map) { const int2 e = res_map[row]; rrow = (int64_t)e.x + ROWTAB_T(e.y); }
    if (out_map) { const int2 e = out_map[row]; orow = (int64_t)e.x + ROWTAB_T(e.y); }
    const float4 a = ld4(t2 + row * t2_ld + c);
    const float4 g = *(const float4*)(gate + item * C + c);
    const float4 r = ld4(res + rrow * res_ld + c);
    float4 o;
    o.x = g.x * a.x + r.x; o.y = g.y * a.y + r.y; o.z = g.z * a.z + r.z; o.w = g.w * a.w + r.w;
    st4(y + orow * y_ld + c, o);
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
        // (v_exp_f32 / v_rcp_f32 forms: the library's expf and the IEEE division made this kernel VALU-bound, two thirds of the chip's issue slots)
        if (l > mx) { const float s = __expf(mx - l); W *= s; m2 *= s; mx = l; }      // exp(-inf) = 0 on the first frame
        const float w = __expf(l - mx);
        W += w;
        const float d = v - mean;
        mean += (w * __builtin_amdgcn_rcpf(W)) * d;
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

// [rows][ld] f32 -> halves, same leading dimension (fp16 mode: the MFA output is kept in f32 for the pooling statistics and rounded once for the attention's MFMA)
__global__ void k_rows_to_half(const float* __restrict__ f, _Float16* __restrict__ h, int64_t n4)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    st4(h + idx * 4, *(const float4*)(f + idx * 4));
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
    const bool f16 = prec == 1 || prec == 2;
    a.Cin = f16 ? L.CinPad16 : L.CinPad; a.cin_real = L.Cin; a.Cout = L.Cout; a.KT = L.KT; a.dil = L.dil;
    a.w_ld = a.Cin;
    a.pad_mode = 0; a.prec = f16 ? 1 : 0;
    if (prec == 2) { a.kt_real = L.KT; a.KT = 2 * L.KT; }       // hi + lo weight planes (weights.cpp)
    if (prec == 3 && L.W16x) { a.prec = 3; a.W16x = L.W16x; a.acc_scale = L.w16x_inv; }      // f32 tensors, split operands on the fp16 MFMA (wide layers)
    return a;
}

#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)

// One-sided reach of the network: 2 (block0, k5) + 7 * (2 + 3 + 4) (each Res2Net block chains 7 k3 convs of dilation 2 / 3 / 4;
// the 1x1 convs and the SE gate add none) = 65 frames.  Frames >= nvalid are excluded from every statistic (SE mean, ASP),
// so frames at or beyond min(501, nvalid + 65) cannot influence the embedding and are not stored at all.  The margin shrinks
// on the way through the network: what block 3 hands to MFA is read at t < nvalid only, so its dilation-4 chain needs its input
// up to nvalid + 28, block 2's output is needed that far and its dilation-3 chain needs nvalid + 49, block 1's chain nvalid + 63.
static const int EC_MARGIN[EC_SPACES] = {65, 49, 28, 0};
static int ec_space_rows(int nvalid, bool skip_dead_rows, int s)
{
    if (!skip_dead_rows) return SD_T;
    const int need = nvalid + EC_MARGIN[s];
    return need > SD_T ? SD_T : (need < 1 ? 1 : need);
}
int ecapa_need_rows(int nvalid, bool skip_dead_rows) { return ec_space_rows(nvalid, skip_dead_rows, 0); }

// d_feats: space-0 rows [plan.off[0][n]][96] of all the plan's items; this call runs items [a0, a1).  Per block b (dilation 2, 3, 4):
//   tdnn1 and the Res2Net chain run in space b (block input: x0 in space 0, x1 / x2 in `cat`, which is stored in space 1),
//   tdnn2, the SE statistics and the gate run in space b + 1, the block output (gate * t2 + input) goes to `cat` (space 1 rows).
// MFA reads `cat` and writes space 3 (the valid frames), where the attentive pooling then runs.  A layer whose output space
// differs from its input space goes through a row table (k_build_rowtab): output row -> item's first input row, frame, last
// stored input frame.  On the planted hour: 4 % fewer rows for tdnn1 / Res2Net, 9 % for tdnn2, 17 % for MFA / ASP than space 0.
template <class T>
static int run_ecapa_t(sd_ctx* c, const float* d_feats_all, const int* d_nvalid_all, const EcapaRowPlan& plan, int64_t a0, int64_t a1, float* d_emb_all)
{
    // conv_gemm precision of the per-frame layers: 1 = fp16, 2 = fp16 with hi + lo weight planes, 3 = f32 tensors with split operands on the fp16 MFMA
    const int P = sizeof(T) == 2 ? (c->ecapa_precision == 2 ? 2 : 1) : (c->ecapa_precision == 3 ? 3 : 0);
    const EcapaWeights& E = c->ew;
    if (!E.loaded) SD_FAIL(c, SD_ERR_MODEL, "embedding model not loaded");
    const int64_t items = a1 - a0;
    if (items <= 0) return SD_OK;
    if (items > ROWTAB_MAX_ITEMS) SD_FAIL(c, SD_ERR_ARG, "ecapa batch of %lld items (limit %d)", (long long)items, ROWTAB_MAX_ITEMS);
    const int C = E.C, C3 = 3 * C;
    // leading dimensions of the activation buffers: rows of exactly 4 KB / 12 KB put the same 128-byte K-slice of all 256 rows of
    // a tile on the same few HBM channels; option ecapa_ld_pad (elements, multiple of 8) skews them
    const int LD = C + c->ecapa_ld_pad, LD3 = C3 + c->ecapa_ld_pad;
    int64_t R[EC_SPACES]; int rbase[EC_SPACES]; const int* ro[EC_SPACES];
    for (int sp = 0; sp < EC_SPACES; ++sp) {
        rbase[sp] = plan.off[sp][(size_t)a0];
        R[sp] = plan.off[sp][(size_t)a1] - rbase[sp];
        ro[sp] = plan.d_off + (size_t)sp * (plan.n + 1) + a0;
    }
    const float* d_feats = d_feats_all + (size_t)rbase[0] * SD_FEAT_LD;
    const int* d_nvalid = d_nvalid_all + a0;
    float* d_emb = d_emb_all + (size_t)a0 * SD_EMB_DIM;
    const int64_t M = R[0], MN = R[3];
    if (M > 0x7fffffff / 2) SD_FAIL(c, SD_ERR_ARG, "ecapa batch too large");
    WS(c, T, x0, "ec_x0", M * LD);
    // tdnn1 output and Res2Net chain of a block in ONE row: [t1 sub-bands 1..7 | t1 sub-band 0 | r1 .. r7] (tdnn1's output channels are
    // rotated by one sub-band at weight-build time, weights.cpp).  Res2Net conv i reads t1_i at (i-1)*S and r_(i-1) at C + (i-2)*S and
    // writes r_i at C + (i-1)*S; tdnn2's input -- sub-band 0 of tdnn1 as it is, then r1..r7 (the Res2Net block's identity branch) -- is the
    // K-contiguous slice at C - S: no copy of the identity sub-band.
    const int LDT = 2 * C - C / 8 + c->ecapa_ld_pad;
    WS(c, T, tr, "ec_tr", M * LDT);
    WS(c, T, t2, "ec_t2", R[1] * LD);
    WS(c, T, cat, "ec_cat", R[1] * LD3);
    WS(c, T, mfa, "ec_mfa", MN * LD3);
    WS(c, T, hid, "ec_hid", MN * 128);
    WS(c, float, se_s, "ec_se_s", items * C);
    WS(c, float, se_h, "ec_se_h", items * 128);
    WS(c, float, se_g, "ec_se_g", items * C);
    WS(c, float, ms, "ec_ms", items * 2 * C3);
    WS(c, float, ib, "ec_ib", items * 128);
    WS(c, float, pooled, "ec_pooled", items * 2 * C3);
    int rc;
    hipStream_t st = c->stream;
    // row tables [output space][input space], built on first use
    int2* tabs[EC_SPACES][EC_SPACES] = {};
    auto tab = [&](int so, int si) -> int2* {
        if (tabs[so][si]) return tabs[so][si];
        char key[32]; snprintf(key, sizeof(key), "ec_rowtab_%d%d", so, si);
        int2* p = ws_get<int2>(c, key, (size_t)R[so] + 128);
        if (!p) return nullptr;
        hipLaunchKernelGGL(k_build_rowtab, dim3((unsigned)items), dim3(256), 0, st, ro[so], rbase[so], ro[si], rbase[si], p);
        tabs[so][si] = p;
        return p;
    };
#define TAB(var, so, si) int2* var = tab(so, si); if (!var) SD_FAIL(c, SD_ERR_HIP, "hipMalloc of a row table failed")

    // blocks[0]: TDNNBlock(80 -> C, k5)
    const void* f_in = d_feats; int f_ld = SD_FEAT_LD;
    if (sizeof(T) == 2) {
        WS(c, _Float16, fh, "ec_feats16", M * 128);
        hipLaunchKernelGGL(k_feats_to_half, GRID1(M * 128), 0, st, d_feats, fh, M);
        KCHECK(c);
        f_in = fh; f_ld = 128;
    }
    { TAB(t00, 0, 0); ConvArgs a = conv_args(E.block0, f_in, f_ld, x0, LD, M, true, P); a.act1 = 1; a.rowtab = t00; if ((rc = launch_conv_gemm(c, a, "block0"))) return rc; }

    for (int b = 0; b < 3; ++b) {
        const auto& B = E.blk[b];
        const int cs = b, os = b + 1, is = b == 0 ? 0 : 1;       // chain space, output space, space the block input is stored in
        const int64_t Mc = R[cs], Mo = R[os];
        const T* xin = (b == 0) ? x0 : cat + (size_t)(b - 1) * C;
        const int xin_ld = (b == 0) ? LD : LD3;
        TAB(t_in, cs, is); TAB(t_cc, cs, cs); TAB(t_oc, os, cs);
        { ConvArgs a = conv_args(B.tdnn1, xin, xin_ld, tr, LDT, Mc, true, P); a.act1 = 1; a.rowtab = t_in; a.in_rows = (int)R[is]; if ((rc = launch_conv_gemm(c, a, "tdnn1"))) return rc; }
        const int S = C / 8;
        for (int i = 1; i < 8; ++i) {
            ConvArgs a = conv_args(B.res[i - 1], tr + (i - 1) * S, LDT, tr + C + (i - 1) * S, LDT, Mc, true, P);
            a.act1 = 1;
            if (i >= 2 && !c->diag_res2_single) { a.X2 = (const float*)(tr + C + (i - 2) * S); a.x2_ld = LDT; }
            a.rowtab = t_cc;
            if ((rc = launch_conv_gemm(c, a, "res2net"))) return rc;
        }
        { ConvArgs a = conv_args(B.tdnn2, tr + C - S, LDT, t2, LD, Mo, true, P); a.act1 = 1; a.rowtab = t_oc; a.in_rows = (int)Mc; if ((rc = launch_conv_gemm(c, a, "tdnn2"))) return rc; }
        {
            ProfScope ps(c, "se_mean", 0, (double)Mo * C * 4.0);
            hipLaunchKernelGGL(k_masked_mean<T>, dim3((C + 255) / 256, (unsigned)items), dim3(256), 0, st, t2, LD, d_nvalid, ro[os], rbase[os], se_s, C);
            KCHECK(c);
        }
        { ConvArgs a = conv_args(B.se1, se_s, C, se_h, 128, items, false); a.act1 = 1; if ((rc = launch_conv_gemm(c, a, "se1"))) return rc; }
        { ConvArgs a = conv_args(B.se2, se_h, 128, se_g, C, items, false); a.act2 = 2; if ((rc = launch_conv_gemm(c, a, "se2"))) return rc; }
        {
            // block output = gate * t2 + block input, stored in `cat` (space 1 rows) at the frames of space os
            const int2* res_map = nullptr; const int2* out_map = nullptr;
            if (os != is) { TAB(m, os, is); res_map = m; }
            if (os != 1) { TAB(m, os, 1); out_map = m; }
            ProfScope ps(c, "se_apply", 0, (double)Mo * C * 12.0);
            hipLaunchKernelGGL(k_se_apply<T>, GRID1(Mo * (C / 4)), 0, st, t2, LD, se_g, xin, xin_ld, cat + (size_t)b * C, LD3, C, Mo, t_oc, res_map, out_map);
            KCHECK(c);
        }
    }
    // mfa: TDNNBlock(3C -> 3C, k1) over cat(x1,x2,x3)
    TAB(t31, 3, 1); TAB(t33, 3, 3);
    // fp16 mode, option ecapa_f16_hp: bit 0 = the MFA output is stored in f32 (one rounding less in front of the pooling statistics, which
    // are differences of large sums) and rounded once for the attention's fp16 MFMA; bit 1 = the attention branch (asp_tdnn, asp_conv: 4 % of
    // the network's FLOPs, and the exponent of the softmax) on the f32 MFMA
    const int hp = sizeof(T) == 2 ? c->ecapa_f16_hp : 0;
    float* mfa32 = nullptr;
    if (hp & 1) { WS(c, float, m32, "ec_mfa32", MN * LD3); mfa32 = m32; }
    {
        ConvArgs a = conv_args(E.mfa, cat, LD3, mfa32 ? (void*)mfa32 : (void*)mfa, LD3, MN, true, P);
        a.act1 = 1; a.rowtab = t31; a.in_rows = (int)R[1]; a.y_f32 = mfa32 ? 1 : 0;
        if ((rc = launch_conv_gemm(c, a, "mfa"))) return rc;
    }
    if (mfa32 && !(hp & 2)) {
        hipLaunchKernelGGL(k_rows_to_half, GRID1(MN * LD3 / 4), 0, st, mfa32, (_Float16*)mfa, MN * LD3 / 4);
        KCHECK(c);
    }
    // ASP with global context: cat[x, mean, std] @ W == x @ Wx + (mean,std) @ Wms  (per-item bias)
    {
        ProfScope ps(c, "asp_stats", 0, (double)MN * C3 * 4.0);
        if (mfa32) hipLaunchKernelGGL(k_asp_stats<float>, dim3((C3 + 255) / 256, (unsigned)items), dim3(256), 0, st, mfa32, LD3, d_nvalid, ro[3], rbase[3], ms, C3);
        else hipLaunchKernelGGL(k_asp_stats<T>, dim3((C3 + 255) / 256, (unsigned)items), dim3(256), 0, st, mfa, LD3, d_nvalid, ro[3], rbase[3], ms, C3);
        KCHECK(c);
    }
    { ConvArgs a = conv_args(E.asp_tdnn_ms, ms, 2 * C3, ib, 128, items, false); if ((rc = launch_conv_gemm(c, a, "asp_ms"))) return rc; }
    // attention logits stay f32 in either mode (they feed an exp: fp16's 3 decimal digits at |logit| ~ 30 would be percents of a weight).
    // f32 mode: cat is dead after mfa and large enough; fp16 mode: its own buffer
    float* logits;
    if (sizeof(T) == 2 || c->ecapa_keep_cat) { WS(c, float, lg, "ec_logits", MN * LD3); logits = lg; } else logits = (float*)cat;
    if (hp & 2) {
        WS(c, float, hid32, "ec_hid32", MN * 128);
        { ConvArgs a = conv_args(E.asp_tdnn_x, mfa32, LD3, hid32, 128, MN, true, 0); a.act1 = 1; a.act2 = 1; a.item_bias = ib; a.ib_ld = 128; a.rowtab = t33; if ((rc = launch_conv_gemm(c, a, "asp_tdnn"))) return rc; }
        { ConvArgs a = conv_args(E.asp_conv, hid32, 128, logits, LD3, MN, true, 0); a.rowtab = t33; if ((rc = launch_conv_gemm(c, a, "asp_conv"))) return rc; }
    } else {
        { ConvArgs a = conv_args(E.asp_tdnn_x, mfa, LD3, hid, 128, MN, true, P); a.act1 = 1; a.act2 = 1; a.item_bias = ib; a.ib_ld = 128; a.rowtab = t33; if ((rc = launch_conv_gemm(c, a, "asp_tdnn"))) return rc; }
        { ConvArgs a = conv_args(E.asp_conv, hid, 128, logits, LD3, MN, true, P); a.rowtab = t33; a.y_f32 = 1; if ((rc = launch_conv_gemm(c, a, "asp_conv"))) return rc; }
    }
    {
        ProfScope ps(c, "asp_pool", 0, (double)MN * C3 * 8.0);
        if (mfa32) hipLaunchKernelGGL(k_asp_pool<float>, dim3((C3 + 255) / 256, (unsigned)items), dim3(256), 0, st, mfa32, logits, LD3, d_nvalid, ro[3], rbase[3], pooled, C3);
        else hipLaunchKernelGGL(k_asp_pool<T>, dim3((C3 + 255) / 256, (unsigned)items), dim3(256), 0, st, mfa, logits, LD3, d_nvalid, ro[3], rbase[3], pooled, C3);
        KCHECK(c);
    }
    // asp_bn folded into fc
    { ConvArgs a = conv_args(E.fc, pooled, 2 * C3, d_emb, SD_EMB_DIM, items, false); if ((rc = launch_conv_gemm(c, a, "fc"))) return rc; }
    return SD_OK;
}

// x3 mode splits f32 activations into fp16 halves: an activation beyond +-65 504 has no hi half (Inf, then NaN all the way to the embedding), where
// the f32 kernels would carry on.  Every x3 batch therefore raises a flag when one of its embeddings is not finite, and the caller
// (ecapa_run_batches) repeats the call on the f32 kernels: the mode is never worse than f32, and an overflow costs time, not the result.
__global__ void k_flag_nonfinite(const float* e, int64_t n, int* flag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !isfinite(e[i])) *flag = 1;
}

int run_ecapa(sd_ctx* c, const float* d_feats, const int* d_nvalid, const EcapaRowPlan& plan, int64_t a0, int64_t a1, float* d_emb)
{
    if (c->ecapa_precision == 1 || c->ecapa_precision == 2) return run_ecapa_t<_Float16>(c, d_feats, d_nvalid, plan, a0, a1, d_emb);
    const int rc = run_ecapa_t<float>(c, d_feats, d_nvalid, plan, a0, a1, d_emb);
    if (rc == SD_OK && c->ecapa_precision == 3 && a1 > a0) {
        WS(c, int, flag, "ec_x3_flag", 1);
        hipLaunchKernelGGL(k_flag_nonfinite, GRID1((a1 - a0) * SD_EMB_DIM), 0, c->stream, d_emb + (size_t)a0 * SD_EMB_DIM, (a1 - a0) * SD_EMB_DIM, flag);
        KCHECK(c);
    }
    return rc;
}

// runs `batches` (a sequence of run_ecapa calls over one set of items); in x3 mode it runs them again on the f32 kernels when the flag says that an
// embedding came out non-finite (one stream synchronisation per call of this function, x3 mode only)
int ecapa_run_batches(sd_ctx* c, const std::function<int()>& batches)
{
    if (c->ecapa_precision != 3) return batches();
    WS(c, int, flag, "ec_x3_flag", 1);
    HIPCHK(c, hipMemsetAsync(flag, 0, sizeof(int), c->stream));
    int rc = batches();
    if (rc) return rc;
    int h = 0;
    HIPCHK(c, hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!h) return SD_OK;
    c->stats["x3_overflow_fallbacks"].launches += 1;
    c->ecapa_precision = 0;
    rc = batches();
    c->ecapa_precision = 3;
    return rc;
}

// host side of the compact row plan: first row of every item in each space, from the items' nvalid (uploaded to d_off)
int ecapa_row_plan(sd_ctx* c, const int* h_nvalid, int64_t n, EcapaRowPlan& plan, int* d_off)
{
    plan.n = n; plan.d_off = d_off;
    for (int sp = 0; sp < EC_SPACES; ++sp) {
        std::vector<int>& off = plan.off[sp];
        off.assign((size_t)n + 1, 0);
        int64_t acc = 0;
        for (int64_t i = 0; i < n; ++i) { off[(size_t)i] = (int)acc; acc += ec_space_rows(h_nvalid[i], c->skip_dead_rows, sp); }
        if (acc > 0x7fffffff / 4) SD_FAIL(c, SD_ERR_ARG, "embedding stage: %lld feature rows in one shard (limit %d)", (long long)acc, 0x7fffffff / 4);
        off[(size_t)n] = (int)acc;
        HIPCHK(c, hipMemcpyAsync(d_off + (size_t)sp * (n + 1), off.data(), (size_t)(n + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SD_OK;
}

int run_embed(sd_ctx* c, const float* d_wav, int64_t n, const float* d_masks, int64_t items, int64_t first_item, float* d_emb)
{
    if (items <= 0) return SD_OK;
    int64_t nb = c->emb_batch_items;
    if (!c->emb_batch_explicit && c->embed_calls == 0 && nb > 768) nb = 768;      // first call of a context: the small arena (common.h)
    c->embed_calls++;
    nb = (nb / 96) * 96; if (nb < 96) nb = 96;
    int rc;
    // front end for the whole range first: items that are NaN by rule (sd.cpp:2479-2549) are dropped here, so the
    // network always runs on full batches of live items (the reference computes the dead ones and overwrites them)
    WS(c, float, lens, "emb_lens", items);
    WS(c, int, nnorm, "emb_nnorm", items);
    WS(c, int, nvalid, "emb_nvalid", items);
    WS(c, int, flags, "emb_flags", items);
    WS(c, int, cidx, "emb_cidx", items);
    WS(c, int, d_rowoff, "emb_rowoff", EC_SPACES * (items + 1));
    WS(c, float, emb_c, "emb_compact", items * SD_EMB_DIM);
    int n_active = 0;
    std::vector<int> h_nvalid;
    if ((rc = frontend_prepare(c, d_masks, items, first_item, lens, nnorm, nvalid, flags, true, &n_active, cidx, &h_nvalid))) return rc;
    { KernelStat& ks = c->stats["items_live"]; ks.launches++; ks.flops += (double)n_active; ks.bytes += (double)items; }   // bench: live / all items
    if (n_active > 0) {
        EcapaRowPlan plan;
        if ((rc = ecapa_row_plan(c, h_nvalid.data(), n_active, plan, d_rowoff))) return rc;
        const std::vector<int>& rowoff = plan.off[0];
        const int64_t rows_all = rowoff[(size_t)n_active];
        WS(c, float, feats, "emb_feats", rows_all * SD_FEAT_LD);
        c->fe_bill_samples = -1;
        if (c->profile) {               // bill the front end what it really reads and writes: selected samples of the live items, stored frames
            const int* d_counts = c->ws["fe_counts"].as<int>();
            std::vector<int> h_cnt((size_t)items), h_cidx((size_t)items);
            HIPCHK(c, hipMemcpyAsync(h_cnt.data(), d_counts, (size_t)items * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(h_cidx.data(), cidx, (size_t)items * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            int64_t tot = 0;
            for (int64_t i = 0; i < items; ++i) if (h_cidx[(size_t)i] >= 0) tot += h_cnt[(size_t)i];
            c->fe_bill_samples = tot; c->fe_bill_frames = rows_all;
        }
        rc = frontend_features(c, d_wav, n, first_item, n_active, true, nnorm, d_rowoff, feats);
        c->fe_bill_samples = -1;
        if (rc) return rc;
        // batches by row budget: whole items, at most what fits the activation workspaces of nb full-length items.  Within that the
        // boundary is placed where the wide-tile launches of the batch waste the least: a launch over M rows runs
        // ceil(ceil(M / 256) / 64) rounds of 64 row panels (8 XCDs x 8 panels per super-block) per group of column tiles, and the
        // last round is only as full as M happens to leave it; the four row spaces have different M, so the best compromise is
        // searched over the last few dozen items (results do not depend on the batching: every row's bits are placement-free).
        // A batch plan whose activation arena cannot be allocated (a second context on the GPU, an 8-h job beside the 80 GB distance matrix: the
        // default plan asks for ~57 GB at the 1-h size) is repeated with a smaller one -- 768 items (16 GB), then 96: every row's bits are
        // batch-independent (tests/test_planted.py), so only time is lost.  Any other failure is returned as it is.
        for (;;) {
        const int64_t cap_rows = nb * SD_TP;
        auto batch_efficiency = [&](int64_t a0, int64_t a1) -> double {
            // {row space, weight = K x groups of 4 column tiles} of the 256 x 256 launches: block0, tdnn1 / tdnn2 of the three blocks, MFA,
            // ASP conv; 16 384 rows (64 panels) per round.  (Adding the 128 x 128 launches -- 65 536 rows per round -- to the model measured
            // the same: their period is longer than the window searched.)
            static const int L[9][2] = {{0, 400}, {0, 1024}, {1, 1024}, {1, 1024}, {2, 1024}, {2, 1024}, {3, 1024}, {3, 3 * 3072}, {3, 2 * 128}};
            double ideal = 0.0, actual = 0.0;
            for (int l = 0; l < 9; ++l) {
                const int64_t M = plan.off[L[l][0]][(size_t)a1] - plan.off[L[l][0]][(size_t)a0];
                ideal += (double)L[l][1] * (double)M / 16384.0;
                actual += (double)L[l][1] * (double)((M + 16383) / 16384);
            }
            return actual > 0.0 ? ideal / actual : 0.0;
        };
        const int64_t n_batches = (rows_all + cap_rows - 1) / cap_rows;
        c->ws_failed.clear();
        rc = ecapa_run_batches(c, [&]() -> int {
        int rc = SD_OK;
        for (int64_t a0 = 0, k = 1; a0 < n_active; ++k) {
            int64_t a_max = a0;                                   // the most the workspaces take
            while (a_max < n_active && a_max - a0 < ROWTAB_MAX_ITEMS && rowoff[(size_t)a_max + 1] - rowoff[(size_t)a0] <= cap_rows) ++a_max;
            if (a_max == a0) a_max = a0 + 1;
            int64_t a1 = a_max;
            if (a_max < n_active && c->skip_dead_rows) {
                // aim at equal shares of what is left, then look for the best boundary among 160 items around the aim
                const int64_t left_batches = n_batches - k + 1 > 1 ? n_batches - k + 1 : 1;
                const int64_t aim_rows = rowoff[(size_t)a0] + (rows_all - rowoff[(size_t)a0] + left_batches - 1) / left_batches;
                int64_t aim = a0 + 1;
                while (aim < a_max && rowoff[(size_t)aim + 1] <= aim_rows) ++aim;
                if (aim + 80 < a_max) aim += 80; else aim = a_max;
                double best = -1.0;
                for (int64_t cand = aim; cand > a0 && cand + 160 > aim; --cand) {
                    const double e = batch_efficiency(a0, cand);
                    if (e > best) { best = e; a1 = cand; }
                }
            }
            if ((rc = run_ecapa(c, feats, nvalid, plan, a0, a1, emb_c))) return rc;
            a0 = a1;
        }
        return rc;
        });
        // out of memory in the activation arena (a workspace named ec_*, recorded by ws_get -- not recognised by the wording of the message): every
        // ec_* buffer is released and the batches are made smaller.  A failure of any other workspace is not retried
        if (rc == SD_ERR_HIP && nb > 96 && c->ws_failed.rfind("ec_", 0) == 0) {
            (void)hipStreamSynchronize(c->stream);
            (void)hipGetLastError();
            for (auto& kv : c->ws) if (kv.first.rfind("ec_", 0) == 0) kv.second.release();
            c->ws_failed.clear();
            nb = nb > 768 ? 768 : 96;
            c->stats["emb_arena_retries"].launches += 1;
            c->err.clear();
            continue;
        }
        break;
        }
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_scatter_emb, GRID1(items * SD_EMB_DIM), 0, c->stream, emb_c, cidx, d_emb, items);
    KCHECK(c);
    return SD_OK;
}
