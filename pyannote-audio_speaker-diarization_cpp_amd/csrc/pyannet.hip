// pyannet.hip -- PyanNet segmentation network (pyannote/segmentation@2022.07 as exported by
// segment/export2.py:16-53).  Replaces SegmentModel::slide + ::infer's Ort::Session::Run
// (sd.cpp:1352-1504): waveform [n] -> sigmoid speaker activities [chunks][293][3].
//
//   k_chunk_norm   InstanceNorm1d(1) over each 5 s chunk read straight out of the resident
//                  waveform (no [chunks][80000] framing copy of the input, sd.cpp:1426-1430)
//   conv_gemm      SincNet conv0 as a GEMM over stride-10 windows (x_ld = 10, K = 251 -> 256),
//                  conv1/conv2 (k=5, "valid"), LSTM input projections, linear layers (f32 MFMA)
//   k_pool_norm    |.| (stage 0) -> MaxPool1d(3) -> InstanceNorm1d(C, affine) -> LeakyReLU
//   k_lstm_rec     persistent bidirectional LSTM recurrence: one workgroup = 32 chunks x one
//                  direction; the 512x128 recurrent matrix lives in VGPRs as the stationary
//                  MFMA operand (128 regs/lane), h_{t-1} is the streamed operand from LDS,
//                  the cell update is lane-local in the MFMA accumulator layout
//   k_classifier   Linear(128->3) + sigmoid
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)

// ---------------------------------------------------------------- k_chunk_norm
// xn[chunk][j] = (x - mean) / sqrt(var + 1e-5) * w + b  for j < L, 0 beyond
__global__ __launch_bounds__(256) void k_chunk_norm(const float* __restrict__ wav, int64_t origin, int64_t first_chunk, int64_t hop, int L,
                                                    float w, float b, float* __restrict__ xn)
{
    __shared__ float red[256];
    const int ck = blockIdx.x, tid = threadIdx.x;
    const int64_t base = (first_chunk + ck) * hop - origin;          // wav[0] is sample `origin` of the recording; hop = SD_HOP (slide) or the row length (sd_segment_chunks)
    const float* x = wav + base;
    float s = 0.0f;
    for (int j = tid; j < L; j += 256) s += x[j];
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float mean = red[0] / (float)L;
    __syncthreads();
    float v = 0.0f;
    for (int j = tid; j < L; j += 256) { const float d = x[j] - mean; v += d * d; }
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float rstd = rsqrtf(red[0] / (float)L + 1e-5f);
    const float a = rstd * w, c = b - mean * rstd * w;
    float* y = xn + (size_t)ck * SD_CHUNK;
    for (int j = tid; j < SD_CHUNK; j += 256) y[j] = (j < L) ? x[j] * a + c : 0.0f;
}

// ---------------------------------------------------------------- k_chunk_stats (shared-conv0 path)
// The chunks overlap by 90 % and conv0 is linear, so it is applied ONCE to the raw waveform (row r = the 251-tap window at
// sample 10 r; chunk ck's frame f is row 800 ck + f) and the chunk's InstanceNorm1d(1) becomes an affine map of that output:
//   conv0((x - mean) * rstd * w + b)[f][ch] = a_ck * Y[800 ck + f][ch] + c_ck * sum_k W[ch][k],   a = rstd * w,  c = b - mean * rstd * w.
// This kernel only produces (a, c) per chunk -- same statistics, same summation order as k_chunk_norm.
__global__ __launch_bounds__(256) void k_chunk_stats(const float* __restrict__ wav, int64_t origin, int64_t first_chunk, int L,
                                                     float w, float b, float2* __restrict__ st)
{
    __shared__ float red[256];
    const int ck = blockIdx.x, tid = threadIdx.x;
    const int64_t base = (first_chunk + ck) * (int64_t)SD_HOP - origin;
    const float* x = wav + base;
    float s = 0.0f;
    for (int j = tid; j < L; j += 256) s += x[j];
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float mean = red[0] / (float)L;
    __syncthreads();
    float v = 0.0f;
    for (int j = tid; j < L; j += 256) { const float d = x[j] - mean; v += d * d; }
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float rstd = rsqrtf(red[0] / (float)L + 1e-5f);
    if (tid == 0) st[ck] = make_float2(rstd * w, b - mean * rstd * w);
}

// ---------------------------------------------------------------- k_pool_norm
// in  [chunk][Lc][C]  ->  out [chunk][Lp][Cpad],  Lp = Lc/3
// out = leaky_relu(instance_norm(maxpool3(abs?(in))))   per (chunk, channel) statistics
#define PN_T 960          // threads per chunk: Q = PN_T / C row groups per channel keep 12-16 loads per channel in flight
// SHARED (stage 0 of the shared-conv0 path): `in` is the conv of the raw waveform, chunk ck starts at row ck * chunk_rows and
// its values are a_ck * in + c_ck * wsum[channel] (k_chunk_stats)
template <int C, int CPAD, bool ABS, bool SHARED>
__global__ __launch_bounds__(PN_T) void k_pool_norm(const float* __restrict__ in, int Lc, int Lp, const float* __restrict__ gw,
                                                   const float* __restrict__ gb, float* __restrict__ out,
                                                   const float2* __restrict__ cst, const float* __restrict__ wsum, int chunk_rows)
{
    constexpr int Q = PN_T / C;
    __shared__ float red[Q][C];
    __shared__ float sa[C], sb[C];
    const int ck = blockIdx.x, tid = threadIdx.x;
    const int c = tid % C, ql = tid / C;
    const float* src = in + (SHARED ? (size_t)ck * chunk_rows * C : (size_t)ck * Lc * C);
    float* dst = out + (size_t)ck * Lp * CPAD;
    float ca = 1.0f, cc = 0.0f;
    if (SHARED) { const float2 e = cst[ck]; ca = e.x; cc = e.y * wsum[c]; }
    float s = 0.0f;
    for (int q = ql; q < Lp; q += Q) {
        float v0 = src[(size_t)(3 * q) * C + c], v1 = src[(size_t)(3 * q + 1) * C + c], v2 = src[(size_t)(3 * q + 2) * C + c];
        if (SHARED) { v0 = v0 * ca + cc; v1 = v1 * ca + cc; v2 = v2 * ca + cc; }
        if (ABS) { v0 = fabsf(v0); v1 = fabsf(v1); v2 = fabsf(v2); }
        const float m = fmaxf(v0, fmaxf(v1, v2));
        dst[(size_t)q * CPAD + c] = m;
        s += m;
    }
    red[ql][c] = s;
    __syncthreads();
    float tot = 0.0f;
    for (int k = 0; k < Q; ++k) tot += red[k][c];
    const float mean = tot / (float)Lp;
    __syncthreads();
    float v = 0.0f;
    for (int q = ql; q < Lp; q += Q) { const float d = dst[(size_t)q * CPAD + c] - mean; v += d * d; }
    red[ql][c] = v;
    __syncthreads();
    if (ql == 0) {
        float var = 0.0f;
        for (int k = 0; k < Q; ++k) var += red[k][c];
        const float rstd = rsqrtf(var / (float)Lp + 1e-5f);
        sa[c] = rstd * gw[c];
        sb[c] = gb[c] - mean * rstd * gw[c];
    }
    __syncthreads();
    const float a = sa[c], b = sb[c];
    for (int q = ql; q < Lp; q += Q) {
        float y = dst[(size_t)q * CPAD + c] * a + b;
        dst[(size_t)q * CPAD + c] = y > 0.0f ? y : 0.01f * y;
    }
    // zero the padding channels
    for (int idx = tid; idx < Lp * (CPAD - C); idx += PN_T) {
        const int q = idx / (CPAD - C), cc = C + idx % (CPAD - C);
        dst[(size_t)q * CPAD + cc] = 0.0f;
    }
}

// ---------------------------------------------------------------- k_lstm_rec
// gate non-linearities of the recurrence: 40 evaluations per lane and step, in a phase where every wave of the workgroup is past its MFMAs (nothing
// overlaps them).  v_exp_f32 / v_rcp_f32 forms (1-2 ulp each) instead of the library's division and tanhf (~35 instructions with branches):
//   sigm(x) = rcp(1 + e^-x);  tanh(x) = x (1 - x^2/3 + 2 x^4/15 - 17 x^6/315) for |x| < 0.18 (the exp form cancels there), else sign(x) (1 - 2 rcp(1 + e^(2|x|)))
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x)
{
    const float ax = fabsf(x), x2 = x * x;
    const float small = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * -0.053968254f)));
    const float big = copysignf(1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * ax)), x);
    return ax < 0.18f ? small : big;
}

#define HLD 132
__global__ __launch_bounds__(512) void k_lstm_rec(const float* __restrict__ G, const float* __restrict__ whh_f,
                                                  const float* __restrict__ whh_b, float* __restrict__ H, int B, int F)
{
    __shared__ __attribute__((aligned(16))) float hbuf[2][32 * HLD];
    const int dir = blockIdx.y, b0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int u0 = 16 * w;
    const float* whh = dir ? whh_b : whh_f;

    // stationary operand: A[tile][s] = Whh[gate*128 + unit][lh*64 + s]
    float A0[64], A1[64];
    {
        const int gsel = li >> 4, unit = u0 + (li & 15);
        const float* r0 = whh + ((size_t)((0 + gsel) * 128 + unit)) * 128 + lh * 64;   // tile 0: gates i (0), f (1)
        const float* r1 = whh + ((size_t)((2 + gsel) * 128 + unit)) * 128 + lh * 64;   // tile 1: gates g (2), o (3)
#pragma unroll
        for (int s = 0; s < 64; s += 4) {
            const float4 v0 = *(const float4*)(r0 + s), v1 = *(const float4*)(r1 + s);
            A0[s] = v0.x; A0[s + 1] = v0.y; A0[s + 2] = v0.z; A0[s + 3] = v0.w;
            A1[s] = v1.x; A1[s + 1] = v1.y; A1[s + 2] = v1.z; A1[s + 3] = v1.w;
        }
    }
    for (int i = tid; i < 2 * 32 * HLD; i += 512) (&hbuf[0][0])[i] = 0.0f;
    float cst[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) cst[r] = 0.0f;
    const int bj = (b0 + li < B) ? (b0 + li) : (B - 1);
    const bool live = (b0 + li) < B;
    const int ua = u0 + 4 * lh, ub = u0 + 8 + 4 * lh;      // the two 4-unit groups this lane owns
    __syncthreads();

    int cur = 0;
    for (int step = 0; step < F; ++step) {
        const int t = dir ? (F - 1 - step) : step;
        const float* g = G + ((size_t)bj * F + t) * 1024 + dir * 512;
        float4 gi[2], gf[2], gg[2], go[2];
        gi[0] = *(const float4*)(g + 0 * 128 + ua); gi[1] = *(const float4*)(g + 0 * 128 + ub);
        gf[0] = *(const float4*)(g + 1 * 128 + ua); gf[1] = *(const float4*)(g + 1 * 128 + ub);
        gg[0] = *(const float4*)(g + 2 * 128 + ua); gg[1] = *(const float4*)(g + 2 * 128 + ub);
        go[0] = *(const float4*)(g + 3 * 128 + ua); go[1] = *(const float4*)(g + 3 * 128 + ub);

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        const float* hb = &hbuf[cur][li * HLD + lh * 64];
#pragma unroll
        for (int s = 0; s < 64; s += 4) {
            const float4 bv = *(const float4*)(hb + s);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[s], bv.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[s], bv.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[s + 1], bv.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[s + 1], bv.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[s + 2], bv.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[s + 2], bv.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[s + 3], bv.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[s + 3], bv.w, acc1, 0, 0, 0);
        }
        const float giv[8] = {gi[0].x, gi[0].y, gi[0].z, gi[0].w, gi[1].x, gi[1].y, gi[1].z, gi[1].w};
        const float gfv[8] = {gf[0].x, gf[0].y, gf[0].z, gf[0].w, gf[1].x, gf[1].y, gf[1].z, gf[1].w};
        const float ggv[8] = {gg[0].x, gg[0].y, gg[0].z, gg[0].w, gg[1].x, gg[1].y, gg[1].z, gg[1].w};
        const float gov[8] = {go[0].x, go[0].y, go[0].z, go[0].w, go[1].x, go[1].y, go[1].z, go[1].w};
        float hv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            // accumulator rows: r -> gate A unit (r&3)+8*(r>>2)+4*lh ; r+8 -> gate B same unit
            const float ig = sigm(acc0[r] + giv[r]);
            const float fg = sigm(acc0[r + 8] + gfv[r]);
            const float gt = tanh_fast(acc1[r] + ggv[r]);
            const float og = sigm(acc1[r + 8] + gov[r]);
            cst[r] = fg * cst[r] + ig * gt;
            hv[r] = og * tanh_fast(cst[r]);
        }
        float* hn = &hbuf[cur ^ 1][li * HLD];
        *(float4*)(hn + ua) = make_float4(hv[0], hv[1], hv[2], hv[3]);
        *(float4*)(hn + ub) = make_float4(hv[4], hv[5], hv[6], hv[7]);
        if (live) {
            float* ho = H + ((size_t)(b0 + li) * F + t) * 256 + dir * 128;
            *(float4*)(ho + ua) = make_float4(hv[0], hv[1], hv[2], hv[3]);
            *(float4*)(ho + ub) = make_float4(hv[4], hv[5], hv[6], hv[7]);
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ---------------------------------------------------------------- k_lstm_rec_x3
// option seg_precision = 3: the same recurrence with both MFMA operands split into hi + lo fp16 halves (conv_gemm_h.hip's x3 arithmetic):
// W_hh * 2^e as two fp16 planes, stationary in the same 128 registers; h is kept in LDS as hi and lo halves (the lane that computes a
// unit writes both); per 16-wide k-block hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16, f32 accumulation, times 2^-e in front of the
// gates: 48 MFMAs of 32 cycles per wave and step instead of 128 of 64.
typedef _Float16 lhalf8 __attribute__((ext_vector_type(8)));
typedef _Float16 lhalf4 __attribute__((ext_vector_type(4)));
#define HXLD 264            // halves per batch row of the LDS image of h: 128 hi + 128 lo + 8 of padding (528 B: conflict-free 16-byte reads)
__global__ __launch_bounds__(512) void k_lstm_rec_x3(const float* __restrict__ G, const _Float16* __restrict__ whh_f, const _Float16* __restrict__ whh_b,
                                                     float inv_f, float inv_b, float* __restrict__ H, int B, int F)
{
    __shared__ __attribute__((aligned(16))) _Float16 hbuf[2][32 * HXLD];
    const int dir = blockIdx.y, b0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int u0 = 16 * w;
    const _Float16* whh = dir ? whh_b : whh_f;
    const float inv = dir ? inv_b : inv_f;

    // stationary operand: Wr[tile][plane][kb] = W[plane][(2 tile + gsel) * 128 + unit][16 kb + 8 lh .. + 7]
    lhalf8 Wr[2][2][8];
    {
        const int gsel = li >> 4, unit = u0 + (li & 15);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int kb = 0; kb < 8; ++kb)
                    Wr[t][pl][kb] = *(const lhalf8*)(whh + (size_t)pl * 512 * 128 + ((size_t)((2 * t + gsel) * 128 + unit)) * 128 + 16 * kb + 8 * lh);
    }
    for (int i = tid; i < 2 * 32 * HXLD; i += 512) (&hbuf[0][0])[i] = (_Float16)0.0f;
    float cst[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) cst[r] = 0.0f;
    const int bj = (b0 + li < B) ? (b0 + li) : (B - 1);
    const bool live = (b0 + li) < B;
    const int ua = u0 + 4 * lh, ub = u0 + 8 + 4 * lh;      // the two 4-unit groups this lane owns
    __syncthreads();

    // the input-projection rows of a step are fetched one step ahead: the MFMA phase (48 x 32 cycles) is shorter than a global load's latency
    float4 ni[2], nf[2], ng[2], no[2];
    auto fetch = [&](int step) {
        const int t = dir ? (F - 1 - step) : step;
        const float* g = G + ((size_t)bj * F + t) * 1024 + dir * 512;
        ni[0] = *(const float4*)(g + 0 * 128 + ua); ni[1] = *(const float4*)(g + 0 * 128 + ub);
        nf[0] = *(const float4*)(g + 1 * 128 + ua); nf[1] = *(const float4*)(g + 1 * 128 + ub);
        ng[0] = *(const float4*)(g + 2 * 128 + ua); ng[1] = *(const float4*)(g + 2 * 128 + ub);
        no[0] = *(const float4*)(g + 3 * 128 + ua); no[1] = *(const float4*)(g + 3 * 128 + ub);
    };
    fetch(0);
    int cur = 0;
    for (int step = 0; step < F; ++step) {
        const int t = dir ? (F - 1 - step) : step;
        float4 gi[2], gf[2], gg[2], go[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { gi[q] = ni[q]; gf[q] = nf[q]; gg[q] = ng[q]; go[q] = no[q]; }
        fetch(step + 1 < F ? step + 1 : step);

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        const _Float16* hb = &hbuf[cur][li * HXLD + 8 * lh];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const lhalf8 hh = *(const lhalf8*)(hb + 16 * kb);
            const lhalf8 hl = *(const lhalf8*)(hb + 128 + 16 * kb);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wr[0][0][kb], hh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wr[1][0][kb], hh, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wr[0][1][kb], hh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wr[1][1][kb], hh, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wr[0][0][kb], hl, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wr[1][0][kb], hl, acc1, 0, 0, 0);
        }
        const float giv[8] = {gi[0].x, gi[0].y, gi[0].z, gi[0].w, gi[1].x, gi[1].y, gi[1].z, gi[1].w};
        const float gfv[8] = {gf[0].x, gf[0].y, gf[0].z, gf[0].w, gf[1].x, gf[1].y, gf[1].z, gf[1].w};
        const float ggv[8] = {gg[0].x, gg[0].y, gg[0].z, gg[0].w, gg[1].x, gg[1].y, gg[1].z, gg[1].w};
        const float gov[8] = {go[0].x, go[0].y, go[0].z, go[0].w, go[1].x, go[1].y, go[1].z, go[1].w};
        float hv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float ig = sigm(acc0[r] * inv + giv[r]);
            const float fg = sigm(acc0[r + 8] * inv + gfv[r]);
            const float gt = tanh_fast(acc1[r] * inv + ggv[r]);
            const float og = sigm(acc1[r + 8] * inv + gov[r]);
            cst[r] = fg * cst[r] + ig * gt;
            hv[r] = og * tanh_fast(cst[r]);
        }
        _Float16* hn = &hbuf[cur ^ 1][li * HXLD];
        lhalf4 h0, l0, h1, l1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h0[e] = (_Float16)hv[e];     l0[e] = (_Float16)(hv[e] - (float)h0[e]);
            h1[e] = (_Float16)hv[4 + e]; l1[e] = (_Float16)(hv[4 + e] - (float)h1[e]);
        }
        *(lhalf4*)(hn + ua) = h0; *(lhalf4*)(hn + 128 + ua) = l0;
        *(lhalf4*)(hn + ub) = h1; *(lhalf4*)(hn + 128 + ub) = l1;
        if (live) {
            float* ho = H + ((size_t)(b0 + li) * F + t) * 256 + dir * 128;
            *(float4*)(ho + ua) = make_float4(hv[0], hv[1], hv[2], hv[3]);
            *(float4*)(ho + ub) = make_float4(hv[4], hv[5], hv[6], hv[7]);
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ---------------------------------------------------------------- k_classifier
// seg[chunk][f][k] = sigmoid(W[k] . y[chunk*F + f] + b[k]) for f < F; zero-padded to 293 frames (sd.cpp:1473-1479)
__global__ __launch_bounds__(256) void k_classifier(const float* __restrict__ y, const float* __restrict__ W, const float* __restrict__ bias,
                                                    float* __restrict__ seg, int64_t chunks, int F)
{
    __shared__ float w[3 * 128];
    for (int i = threadIdx.x; i < 384; i += 256) w[i] = W[i];
    __syncthreads();
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= chunks * SD_FRAMES) return;
    const int64_t ck = idx / SD_FRAMES;
    const int f = (int)(idx - ck * SD_FRAMES);
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
    if (f < F) {
        const float* x = y + ((size_t)ck * F + f) * 128;
        float a0 = bias[0], a1 = bias[1], a2 = bias[2];
        for (int k = 0; k < 128; k += 4) {
            const float4 v = *(const float4*)(x + k);
            a0 += v.x * w[k] + v.y * w[k + 1] + v.z * w[k + 2] + v.w * w[k + 3];
            a1 += v.x * w[128 + k] + v.y * w[129 + k] + v.z * w[130 + k] + v.w * w[131 + k];
            a2 += v.x * w[256 + k] + v.y * w[257 + k] + v.z * w[258 + k] + v.w * w[259 + k];
        }
        o0 = 1.0f / (1.0f + expf(-a0)); o1 = 1.0f / (1.0f + expf(-a1)); o2 = 1.0f / (1.0f + expf(-a2));
    }
    seg[idx * 3 + 0] = o0; seg[idx * 3 + 1] = o1; seg[idx * 3 + 2] = o2;
}

__global__ void k_zero_f32(float* p, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0f;
}

// identity row table for a dense [M][K] operand: the 256 x 256 kernel (conv_gemm_h.hip) addresses its rows through ConvArgs::rowtab, which
// packs the frame in 10 bits and the item in 12 -- rows are grouped into pseudo-items of 512 (up to 4095 * 512 rows)
__global__ void k_dense_rowtab(int2* __restrict__ rowtab, int64_t M)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < M) rowtab[g] = make_int2((int)((g >> 9) << 9), (int)(g & 511) | (511 << 10) | ((int)(g >> 9) << 20));
}

static ConvArgs gemm_args(const ConvLayer& L, const float* X, int x_ld, float* Y, int y_ld, int64_t M)
{
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.X = X; a.x_ld = x_ld; a.Y = Y; a.y_ld = y_ld; a.W = L.W; a.bias = L.bias;
    a.M = (int)M; a.TpIn = a.TpOut = (int)M; a.Tin = a.T = (int)M;
    a.Cin = L.CinPad; a.cin_real = L.Cin; a.Cout = L.Cout; a.KT = L.KT; a.dil = L.dil; a.pad_mode = 1;
    return a;
}

// one batch of `cnt` chunks that all have L samples
// hop = samples between the starts of consecutive chunks in d_wav: SD_HOP for the sliding window, the row length for separate rows
static int seg_batch(sd_ctx* c, const float* d_wav, int64_t n, int64_t first_chunk, int64_t cnt, int L, float* d_seg, int64_t hop = SD_HOP)
{
    const SegWeights& S = c->sw;
    hipStream_t st = c->stream;
    const int L0 = (L >= 251) ? (L - 251) / 10 + 1 : 0;
    const int P0 = L0 / 3, L1 = P0 - 4, P1 = L1 / 3, L2 = P1 - 4, P2 = L2 / 3;
    if (L0 <= 0 || P0 <= 0 || L1 <= 0 || P1 <= 0 || L2 <= 0 || P2 <= 0) {
        // too short for a single output frame: the reference pads the missing frames with zeros (sd.cpp:1473-1479)
        hipLaunchKernelGGL(k_zero_f32, GRID1(cnt * SD_FRAMES * 3), 0, st, d_seg, cnt * SD_FRAMES * 3);
        KCHECK(c);
        return SD_OK;
    }
    const int F = P2 > SD_FRAMES ? SD_FRAMES : P2;
    const int64_t CB = cnt;
    // the five largest buffers of this stage share their memory with the embedding stage's activation arena (ecapa.hip: the stages never overlap --
    // shard_infer runs them one after the other on one stream): 17 GB less to allocate on a context's first job, 16 GB less held afterwards
    WS(c, float, p0, "ec_mfa", CB * P0 * 96);
    WS(c, float, c1, "ec_tr", CB * L1 * 60);
    WS(c, float, p1, "sg_p1", CB * P1 * 64);
    WS(c, float, c2, "sg_c2", CB * L2 * 60);
    WS(c, float, p2, "sg_p2", CB * P2 * 64);
    WS(c, float, G, "ec_cat", CB * F * 1024);
    WS(c, float, Ha, "ec_t2", CB * F * 256);
    WS(c, float, Hb, "ec_x0", CB * F * 256);
    WS(c, float, y0, "sg_y0", CB * F * 128);
    WS(c, float, y1, "sg_y1", CB * F * 128);
    int rc;
    if (c->seg_shared_conv0 && c->wav_padded && hop == SD_HOP && S.conv0.Cout == 80 && S.conv0_wsum) {
        // conv0 once over the batch's stretch of the waveform: rows 0 .. 800 (CB - 1) + L0 (chunk ck's frames start at row 800 ck;
        // the last window ends 4 samples behind the last chunk, inside the waveform's padding), then the chunk's normalisation as an
        // affine map inside the pooling kernel.  10x fewer FLOPs and 10x less output than one conv per chunk.
        const int hop_rows = SD_HOP / 10;
        const int64_t MY = (int64_t)hop_rows * (CB - 1) + L0;
        WS(c, float2, cst, "sg_cst", CB);
        WS(c, float, c0s, "sg_c0s", MY * 80);
        hipLaunchKernelGGL(k_chunk_stats, dim3((unsigned)CB), dim3(256), 0, st, d_wav, c->wav_origin, first_chunk, L, S.wn_w, S.wn_b, cst);
        KCHECK(c);
        ConvArgs a; memset(&a, 0, sizeof(a));
        a.X = d_wav + (first_chunk * (int64_t)SD_HOP - c->wav_origin); a.x_ld = 10; a.W = S.conv0.W; a.Y = c0s; a.y_ld = 80;
        a.M = (int)MY; a.TpIn = a.TpOut = a.Tin = a.T = (int)MY;
        a.Cin = 256; a.cin_real = 251; a.Cout = 80; a.KT = 1; a.dil = 1; a.pad_mode = 1;
        if ((rc = launch_conv_narrow(c, a, "sinc0")) == 1) rc = launch_conv_gemm(c, a, "sinc0");
        if (rc) return rc;
        hipLaunchKernelGGL((k_pool_norm<80, 96, true, true>), dim3((unsigned)CB), dim3(PN_T), 0, st, c0s, L0, P0, S.in_w[0], S.in_b[0], p0, cst, S.conv0_wsum, hop_rows);
        KCHECK(c);
    } else {
        WS(c, float, xn, "sg_xn", CB * SD_CHUNK + 512);
        WS(c, float, c0, "sg_c0", CB * L0 * 80);
        hipLaunchKernelGGL(k_chunk_norm, dim3((unsigned)CB), dim3(256), 0, st, d_wav, c->wav_origin, first_chunk, hop, L, S.wn_w, S.wn_b, xn);
        KCHECK(c);
        {   // conv0: rows = output positions, row r reads xn[10 r .. 10 r + 256)
            ConvArgs a; memset(&a, 0, sizeof(a));
            a.X = xn; a.x_ld = 10; a.W = S.conv0.W; a.Y = c0; a.y_ld = 80;
            a.M = (int)(CB * L0); a.TpIn = SD_CHUNK / 10; a.TpOut = L0; a.Tin = SD_CHUNK / 10; a.T = L0;
            a.Cin = 256; a.cin_real = 251; a.Cout = 80; a.KT = 1; a.dil = 1; a.pad_mode = 1;
            if ((rc = launch_conv_narrow(c, a, "sinc0")) == 1) rc = launch_conv_gemm(c, a, "sinc0");
            if (rc) return rc;
        }
        hipLaunchKernelGGL((k_pool_norm<80, 96, true, false>), dim3((unsigned)CB), dim3(PN_T), 0, st, c0, L0, P0, S.in_w[0], S.in_b[0], p0, nullptr, nullptr, 0);
        KCHECK(c);
    }
    {
        ConvArgs a; memset(&a, 0, sizeof(a));
        a.X = p0; a.x_ld = 96; a.W = S.conv1.W; a.bias = S.conv1.bias; a.Y = c1; a.y_ld = 60;
        a.M = (int)(CB * L1); a.TpIn = P0; a.TpOut = L1; a.Tin = P0; a.T = L1;
        a.Cin = 96; a.cin_real = 80; a.Cout = 60; a.KT = 5; a.dil = 1; a.pad_mode = 1;
        if ((rc = launch_conv_narrow(c, a, "sinc1")) == 1) rc = launch_conv_gemm(c, a, "sinc1");
        if (rc) return rc;
    }
    hipLaunchKernelGGL((k_pool_norm<60, 64, false, false>), dim3((unsigned)CB), dim3(PN_T), 0, st, c1, L1, P1, S.in_w[1], S.in_b[1], p1, nullptr, nullptr, 0);
    KCHECK(c);
    {
        ConvArgs a; memset(&a, 0, sizeof(a));
        a.X = p1; a.x_ld = 64; a.W = S.conv2.W; a.bias = S.conv2.bias; a.Y = c2; a.y_ld = 60;
        a.M = (int)(CB * L2); a.TpIn = P1; a.TpOut = L2; a.Tin = P1; a.T = L2;
        a.Cin = 64; a.cin_real = 60; a.Cout = 60; a.KT = 5; a.dil = 1; a.pad_mode = 1;
        if ((rc = launch_conv_narrow(c, a, "sinc2")) == 1) rc = launch_conv_gemm(c, a, "sinc2");
        if (rc) return rc;
    }
    hipLaunchKernelGGL((k_pool_norm<60, 64, false, false>), dim3((unsigned)CB), dim3(PN_T), 0, st, c2, L2, P2, S.in_w[2], S.in_b[2], p2, nullptr, nullptr, 0);
    KCHECK(c);
    // if P2 > 293 (cannot happen for L <= 80000) only the first F frames would be used
    const float* lin = p2; int lin_ld = 64; int lin_rows_per_chunk = P2;
    float* hout = Ha;
    int2* dense_tab = nullptr;
    for (int l = 0; l < 4; ++l) {
        if (lin_rows_per_chunk != F) SD_FAIL(c, SD_ERR_ARG, "unexpected frame count %d", P2);
        {
            // the input projections of layers 1-3 (K = 256, 1024 outputs) on the 256 x 256 tile: a dense operand seen through an identity row table.
            // Same bits as the 128 x 128 form (K is summed in the same order in both); layer 0 (K = 64) is below that kernel's shortest contraction.
            ConvArgs a = gemm_args(S.lstm_ih[l], lin, lin_ld, G, 1024, CB * F);
            if (c->seg_wide_ih && l > 0 && CB * F <= (int64_t)ROWTAB_MAX_ITEMS * 512) {
                if (!dense_tab) {
                    dense_tab = ws_get<int2>(c, "seg_dense_rowtab", (size_t)(CB * F) + 512);
                    if (!dense_tab) SD_FAIL(c, SD_ERR_HIP, "hipMalloc of the dense row table failed");
                    hipLaunchKernelGGL(k_dense_rowtab, dim3((unsigned)((CB * F + 255) / 256)), dim3(256), 0, st, dense_tab, CB * F);
                    KCHECK(c);
                }
                a.rowtab = dense_tab; a.pad_mode = 0; a.Tin = 512; a.T = 512; a.TpIn = a.TpOut = 512; a.in_rows = (int)(CB * F);
                if (seg_prec(c) == 3 && S.lstm_ih[l].W16x) { a.prec = 3; a.W16x = S.lstm_ih[l].W16x; a.acc_scale = S.lstm_ih[l].w16x_inv; }   // split operands (conv_gemm_h.hip P = 3)
            }
            if ((rc = launch_conv_gemm(c, a, "lstm_ih"))) return rc;
        }
        {
            ProfScope ps(c, "lstm_rec", 2.0 * CB * F * 2 * 512 * 128, 0);
            if (seg_prec(c) == 3 && S.lstm_hh_x[l][0] && S.lstm_hh_x[l][1])
                hipLaunchKernelGGL(k_lstm_rec_x3, dim3((unsigned)((CB + 31) / 32), 2), dim3(512), 0, st, G, (const _Float16*)S.lstm_hh_x[l][0], (const _Float16*)S.lstm_hh_x[l][1],
                                   S.lstm_hh_inv[l][0], S.lstm_hh_inv[l][1], hout, (int)CB, F);
            else
                hipLaunchKernelGGL(k_lstm_rec, dim3((unsigned)((CB + 31) / 32), 2), dim3(512), 0, st, G, S.lstm_hh[l][0], S.lstm_hh[l][1], hout, (int)CB, F);
            KCHECK(c);
        }
        lin = hout; lin_ld = 256; lin_rows_per_chunk = F;
        hout = (hout == Ha) ? Hb : Ha;
    }
    { ConvArgs a = gemm_args(S.lin0, lin, 256, y0, 128, CB * F); a.act1 = 2; if ((rc = launch_conv_gemm(c, a, "lin0"))) return rc; }
    { ConvArgs a = gemm_args(S.lin1, y0, 128, y1, 128, CB * F); a.act1 = 2; if ((rc = launch_conv_gemm(c, a, "lin1"))) return rc; }
    hipLaunchKernelGGL(k_classifier, GRID1(CB * SD_FRAMES), 0, st, y1, S.cls_w, S.cls_b, d_seg, CB, F);
    KCHECK(c);
    return SD_OK;
}

// chunks [chunk_lo, chunk_hi) of the n-sample waveform -> d_seg [hi-lo][293][3]
int run_segment(sd_ctx* c, const float* d_wav, int64_t n, int64_t chunk_lo, int64_t chunk_hi, float* d_seg)
{
    if (!c->sw.loaded) SD_FAIL(c, SD_ERR_MODEL, "segmentation model not loaded");
    int64_t last_len = 0;
    const int64_t total = sd_num_chunks(n, &last_len);
    if (chunk_lo < 0 || chunk_hi > total || chunk_lo > chunk_hi) SD_FAIL(c, SD_ERR_ARG, "chunk range [%lld,%lld) outside [0,%lld)", (long long)chunk_lo, (long long)chunk_hi, (long long)total);
    // chunks < total-1 are full; the final ("last chunk" branch, sd.cpp:1457-1480) one may be shorter
    int64_t full_hi = chunk_hi;
    const bool has_short_tail = (chunk_hi == total) && (last_len > 0) && (last_len < SD_CHUNK);
    if (has_short_tail) full_hi = total - 1;
    int64_t cb = c->seg_batch_chunks;
    if (cb < 1) cb = 1;
    int rc;
    for (int64_t k = chunk_lo; k < full_hi; k += cb) {
        const int64_t cnt = (full_hi - k < cb) ? full_hi - k : cb;
        if ((rc = seg_batch(c, d_wav, n, k, cnt, SD_CHUNK, d_seg + (size_t)(k - chunk_lo) * SD_FRAMES * 3))) return rc;
    }
    if (has_short_tail && chunk_lo <= total - 1)
        if ((rc = seg_batch(c, d_wav, n, total - 1, 1, (int)last_len, d_seg + (size_t)(total - 1 - chunk_lo) * SD_FRAMES * 3))) return rc;
    return SD_OK;
}

// SegmentModel::infer as the reference declares it (sd.cpp:1352-1404): `rows` separate waveforms of T samples each, [rows][T] -> [rows][293][3]
// (frames beyond the *frames the network yields for T samples are zero, as slide() pads them, sd.cpp:1473-1479).  Same kernels as run_segment
// with the chunk stride T instead of SD_HOP (the shared-conv0 shortcut needs overlapping chunks and is off).
int run_segment_rows(sd_ctx* c, const float* d_rows, int64_t rows, int T, float* d_seg, int* frames)
{
    if (!c->sw.loaded) SD_FAIL(c, SD_ERR_MODEL, "segmentation model not loaded");
    if (T < 1 || T > SD_CHUNK) SD_FAIL(c, SD_ERR_ARG, "sd_segment_chunks: T = %d samples per row (1 .. %d)", T, SD_CHUNK);
    const int L0 = (T >= 251) ? (T - 251) / 10 + 1 : 0;
    const int P0 = L0 / 3, L1 = P0 - 4, P1 = L1 / 3, L2 = P1 - 4, P2 = L2 / 3;
    if (frames) *frames = (L0 <= 0 || P0 <= 0 || L1 <= 0 || P1 <= 0 || L2 <= 0 || P2 <= 0) ? 0 : (P2 > SD_FRAMES ? SD_FRAMES : P2);
    int64_t cb = c->seg_batch_chunks;
    if (cb < 1) cb = 1;
    const int64_t origin = c->wav_origin;
    c->wav_origin = 0;
    int rc = SD_OK;
    for (int64_t k = 0; k < rows && rc == SD_OK; k += cb)
        rc = seg_batch(c, d_rows, rows * (int64_t)T, k, rows - k < cb ? rows - k : cb, T, d_seg + (size_t)k * SD_FRAMES * 3, T);
    c->wav_origin = origin;
    return rc;
}
