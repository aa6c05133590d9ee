// conv_gemm.hip -- implicit-GEMM 1-D convolution / linear layer on the gfx950 f32 MFMA
// (v_mfma_f32_32x32x2_f32, exact f32, 64 FLOP/clk/SIMD).
//
// Replaces the dense contractions the reference delegates to onnxruntime inside
// emd4.onnx / segment2.onnx (sd.cpp:1947-1949, 1378-1380): every TDNN / Res2Net /
// SE / ASP 1x1 and dilated conv of ECAPA-TDNN, the SincNet convs, the LSTM input
// projections and the PyanNet linear layers.
//
// Layout: activations are channels-last [item][row(time)][channel] so that both
// MFMA operands are K-contiguous: A = X rows (time) x Cin, B = W[tap][Cout][Cin].
// A conv with KT taps is KT accumulated GEMMs whose A rows are shifted (and
// reflect-padded) by (tap - KT/2)*dilation rows.
//
// Tile: 128 rows x 128 cols x 32 K per step, 256 threads = 2x2 waves, each wave
// 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs).  LDS rows are padded to 36 floats:
// ds_read_b128 of 16 distinct rows then hits 16 distinct 4-bank slots (conflict
// free), and every row start stays 16-byte aligned.  Global->LDS goes through
// registers with the next step's loads issued before the current step's MFMAs
// (2 LDS buffers, one barrier per K step).  blockIdx is remapped so that the
// N tiles sharing one A row-panel run on one XCD (shared L2).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned float4 (x_ld may be 10)

#define BM 128
#define BN 128
#define BK 32
#define LDP 36

template <bool HAS_X2>
__global__ __launch_bounds__(256, 2) void k_conv_gemm(ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDP];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDP];

    // XCD-aware tile mapping: blocks b and b+8 share an XCD; give each XCD whole row panels
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int mt = (slot / a.n_tiles) * 8 + xcd;
    const int nt = slot % a.n_tiles;
    if (mt >= a.m_tiles) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int c4 = tid & 7, r0 = tid >> 3;

    // per-thread staging rows
    int item_base[4], tt[4];
    size_t wrow[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int g = m0 + r0 + 32 * p;
        if (g > a.M - 1) g = a.M - 1;
        int b = g / a.TpOut;
        int t = g - b * a.TpOut;
        if (t > a.T - 1) t = a.T - 1;
        item_base[p] = b * a.TpIn;
        tt[p] = t;
        int co = n0 + r0 + 32 * p;
        if (co > a.Cout - 1) co = a.Cout - 1;
        wrow[p] = (size_t)co * a.Cin;
    }
    const int kcs = a.Cin / BK;
    const int S = a.KT * kcs;
    const int half = a.KT / 2;

    f4u ra[4], rb[4];
    auto gload = [&](int s) {
        const int kk = s / kcs, kc = s - kk * kcs;
        const int coff = kc * BK + c4 * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int q;
            if (a.pad_mode == 0) {
                q = tt[p] + (kk - half) * a.dil;
                if (q < 0) q = -q;
                if (q >= a.Tin) q = 2 * (a.Tin - 1) - q;
                if (q < 0) q = 0;
            } else {
                q = tt[p] + kk * a.dil;
                if (q > a.Tin - 1) q = a.Tin - 1;
            }
            const size_t row = (size_t)(item_base[p] + q);
            ra[p] = *(const f4u*)(a.X + row * a.x_ld + coff);
            if (HAS_X2) {
                f4u v2 = *(const f4u*)(a.X2 + row * a.x2_ld + coff);
                ra[p] += v2;
            }
            rb[p] = *(const f4u*)(a.W + (size_t)kk * a.Cout * a.Cin + wrow[p] + coff);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *(float4*)&As[buf][(r0 + 32 * p) * LDP + c4 * 4] = make_float4(ra[p][0], ra[p][1], ra[p][2], ra[p][3]);
            *(float4*)&Bs[buf][(r0 + 32 * p) * LDP + c4 * 4] = make_float4(rb[p][0], rb[p][1], rb[p][2], rb[p][3]);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    gload(0);
    lstore(0);
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    const int aoff = (wr * 64 + li) * LDP + lh * 16;
    const int boff = (wc * 64 + li) * LDP + lh * 16;

    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        if (s + 1 < S) gload(s + 1);
        const float* Ab = &As[buf][aoff];
        const float* Bb = &Bs[buf][boff];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a0 = *(const float4*)(Ab + q * 4);
            const float4 a1 = *(const float4*)(Ab + 32 * LDP + q * 4);
            const float4 b0 = *(const float4*)(Bb + q * 4);
            const float4 b1 = *(const float4*)(Bb + 32 * LDP + q * 4);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv0[e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv1[e], acc[1][1], 0, 0, 0);
            }
        }
        if (s + 1 < S) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int g = m0 + row;
            if (g >= a.M) continue;
            const int b = g / a.TpOut;
            const int t = g - b * a.TpOut;
            const bool live = t < a.T;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = n0 + wc * 64 + j * 32 + li;
                if (co >= a.Cout) continue;
                float v = acc[i][j][r];
                if (a.bias) v += a.bias[co];
                if (a.item_bias) v += a.item_bias[(size_t)b * a.ib_ld + co];
                if (a.act1 == 1) v = v > 0.0f ? v : 0.0f;
                else if (a.act1 == 2) v = v > 0.0f ? v : 0.01f * v;
                if (a.scale) v = v * a.scale[co] + a.shift[co];
                if (a.act2 == 1) v = tanhf(v);
                else if (a.act2 == 2) v = 1.0f / (1.0f + expf(-v));
                if (a.R) v += a.R[(size_t)g * a.r_ld + co];
                a.Y[(size_t)g * a.y_ld + co] = live ? v : 0.0f;
            }
        }
    }
}

int launch_conv_gemm(sd_ctx* c, const ConvArgs& in, const char* tag)
{
    ConvArgs a = in;
    if (a.Cin % BK != 0) SD_FAIL(c, SD_ERR_ARG, "conv_gemm(%s): Cin=%d not a multiple of %d", tag, a.Cin, BK);
    if (a.M <= 0) return SD_OK;
    a.m_tiles = (a.M + BM - 1) / BM;
    a.n_tiles = (a.Cout + BN - 1) / BN;
    const int grid = ((a.m_tiles + 7) / 8) * 8 * a.n_tiles;
    // algorithmic work: valid rows only (T of every TpOut), un-padded input channels
    const double rows = (double)(a.M / a.TpOut) * a.T + (double)((a.M % a.TpOut) < a.T ? (a.M % a.TpOut) : a.T);
    const int cin = a.cin_real > 0 ? a.cin_real : a.Cin;
    const double flops = 2.0 * rows * a.Cout * cin * a.KT;
    const double bytes = 4.0 * (rows * cin * (a.X2 ? 2 : 1) + rows * a.Cout + (double)a.Cout * cin * a.KT);
    ProfScope ps(c, c->profile_detail ? std::string("conv_gemm:") + tag : std::string("conv_gemm"), flops, bytes);
    if (a.X2) hipLaunchKernelGGL(k_conv_gemm<true>, dim3(grid), dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL(k_conv_gemm<false>, dim3(grid), dim3(256), 0, c->stream, a);
    KCHECK(c);
    return SD_OK;
}
