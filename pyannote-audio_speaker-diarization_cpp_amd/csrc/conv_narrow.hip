// conv_narrow.hip -- the three SincNet convolutions of PyanNet (Cout = 80, 60, 60) on the f32 MFMA with a tile as wide as
// the layer: 128 rows x 32 NJ columns (NJ = 3 for 80 channels, 2 for 60), 4 waves of 32 rows x NJ column tiles.
// conv_gemm.hip's 128 x 128 tile spends 37 % (Cout 80) resp. 53 % (Cout 60) of its MFMAs on padding columns
// (profiles/r02_layer_profile.txt: 65 / 43 / 46 TFLOP/s on these layers).  Same operand layout and K order as conv_gemm.hip
// (channels-last rows, W [tap][Cout][CinPad], v_mfma_f32_32x32x2_f32 over k pairs (k, k + 16) of every 32-chunk, taps outermost),
// "valid" row map (src row = t + tap * dil), dense row spaces; one tile per workgroup, LDS double buffered, two workgroups per CU.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned float4 (x_ld = 10 for the strided first layer)

#define NLDP 36

template <int NJ>
__global__ __launch_bounds__(256, 2) void k_conv_narrow(ConvArgs a)
{
    constexpr int BNn = 32 * NJ;
    __shared__ __attribute__((aligned(16))) float As[2][128 * NLDP];
    __shared__ __attribute__((aligned(16))) float Bs[2][BNn * NLDP];
    const int tid = threadIdx.x, lane = tid & 63, wr = tid >> 6;
    const int c4 = tid & 7, r0 = tid >> 3;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 128;
    const int kcs = a.Cin / 32, S = a.KT * kcs;

    // rows of this tile: output row g -> item b, frame t; source row of tap kk = b * TpIn + min(t + kk * dil, Tin - 1)
    size_t rowbase[4]; int tt[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int g = m0 + r0 + 32 * p;
        if (g > a.M - 1) g = a.M - 1;
        const int b = g / a.TpOut;
        int t = g - b * a.TpOut;
        if (t > a.T - 1) t = a.T - 1;
        rowbase[p] = (size_t)b * a.TpIn; tt[p] = t;
    }
    f4u ra[4], rb[NJ];
    // the load stream never branches: the step after the last one re-reads the last one (its staging lands in the idle LDS buffer),
    // so every K-step is one basic block the scheduling pins below can work in
    auto gloadA = [&](int s) {
        const int kk = s / kcs, kc = s - kk * kcs;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int qr = tt[p] + kk * a.dil;
            if (qr > a.Tin - 1) qr = a.Tin - 1;
            ra[p] = *(const f4u*)(a.X + (rowbase[p] + qr) * a.x_ld + kc * 32 + c4 * 4);
        }
    };
    auto gloadB = [&](int s) {
        const int kk = s / kcs, kc = s - kk * kcs;
#pragma unroll
        for (int p = 0; p < NJ; ++p) {
            int co = r0 + 32 * p;
            if (co > a.Cout - 1) co = a.Cout - 1;                  // padding columns compute a copy of the last channel, never stored
            rb[p] = *(const f4u*)(a.W + ((size_t)kk * a.Cout + co) * a.w_ld + kc * 32 + c4 * 4);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) *(float4*)&As[buf][(r0 + 32 * p) * NLDP + c4 * 4] = make_float4(ra[p][0], ra[p][1], ra[p][2], ra[p][3]);
#pragma unroll
        for (int p = 0; p < NJ; ++p) *(float4*)&Bs[buf][(r0 + 32 * p) * NLDP + c4 * 4] = make_float4(rb[p][0], rb[p][1], rb[p][2], rb[p][3]);
    };
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // operand fragments of K-group qq (k = 16 lh + 4 qq .. +3 of the 32-chunk), double buffered in registers
    float4 fa[2], fb[2][NJ];
    auto lfrag = [&](int buf, int qq, int f) {
        fa[f] = *(const float4*)(&As[buf][(wr * 32 + li) * NLDP + lh * 16] + qq * 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[f][j] = *(const float4*)(&Bs[buf][li * NLDP + lh * 16] + j * 32 * NLDP + qq * 4);
    };
    auto mma = [&](int f) {
        const float av[4] = {fa[f].x, fa[f].y, fa[f].z, fa[f].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float bv = e == 0 ? fb[f][j].x : e == 1 ? fb[f][j].y : e == 2 ? fb[f][j].z : fb[f][j].w;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv, acc[j], 0, 0, 0);
            }
    };

    gloadA(0); gloadB(0);
    lstore(0);
    __syncthreads();
    lfrag(0, 0, 0);
    // one K-step = 4 K-groups of 4 NJ MFMAs.  Memory instructions are pinned one by one behind the first MFMAs of their group
    // (sched_group_barrier masks: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write), see conv_gemm_h.hip
#define N_PAIR(mask, n) do { _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(mask, 1, 0); } } while (0)
    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        const int sn = s + 1 < S ? s + 1 : S - 1;
        lfrag(buf, 1, 1);
        gloadA(sn);
        mma(0);
        N_PAIR(0x100, 1 + NJ); N_PAIR(0x020, 4);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NJ - 5 - NJ, 0);
        __builtin_amdgcn_sched_barrier(0);
        lfrag(buf, 2, 0);
        gloadB(sn);
        mma(1);
        N_PAIR(0x100, 1 + NJ); N_PAIR(0x020, NJ);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NJ - 1 - 2 * NJ, 0);
        __builtin_amdgcn_sched_barrier(0);
        lfrag(buf, 3, 1);
        mma(0);
        lstore(buf ^ 1);
        N_PAIR(0x100, 1 + NJ); N_PAIR(0x200, NJ == 3 ? 7 : 5);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        lfrag(buf ^ 1, 0, 0);
        mma(1);
        N_PAIR(0x100, 1 + NJ);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NJ - 1 - NJ, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: C layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); bias, then the 4 x 4 quad transpose of conv_gemm.hip
    const int lq = lane & 3;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cc = j * 32 + li;
        const float cb = (a.bias && cc < a.Cout) ? a.bias[cc] : 0.0f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = acc[j][4 * gq + e] + cb;
            float s0 = (lq & 1) ? x[0] : x[1];
            float s1 = (lq & 1) ? x[2] : x[3];
            float t0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s0), 0xB1, 0xF, 0xF, true));
            float t1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s1), 0xB1, 0xF, 0xF, true));
            if (lq & 1) { x[0] = t0; x[2] = t1; } else { x[1] = t0; x[3] = t1; }
            s0 = (lq & 2) ? x[0] : x[2];
            s1 = (lq & 2) ? x[1] : x[3];
            t0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s0), 0x4E, 0xF, 0xF, true));
            t1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s1), 0x4E, 0xF, 0xF, true));
            if (lq & 2) { x[0] = t0; x[1] = t1; } else { x[2] = t0; x[3] = t1; }
            const int g = m0 + wr * 32 + 8 * gq + 4 * lh + lq;
            const int co = j * 32 + (li & ~3);
            if (g < a.M && co < a.Cout) {
                const int b = g / a.TpOut, t = g - b * a.TpOut;
                if (t < a.T) *(float4*)(a.Y + (size_t)g * a.y_ld + co) = make_float4(x[0], x[1], x[2], x[3]);
            }
        }
    }
}

// returns 1 when the layer does not fit (the caller then uses conv_gemm.hip)
int launch_conv_narrow(sd_ctx* c, const ConvArgs& in, const char* tag)
{
    ConvArgs a = in;
    if (a.w_ld <= 0) a.w_ld = a.Cin;
    if (a.prec != 0 || a.rowtab || a.X2 || a.item_bias || a.R || a.scale || a.act1 || a.act2 || a.pad_mode != 1 || a.Cout > 96 || (a.Cout & 3) || (a.y_ld & 3) ||
        a.Cin % 32 != 0 || a.TpOut != a.T || a.M <= 0) return 1;
    const int grid = (a.M + 127) / 128;
    const int cin = a.cin_real > 0 ? a.cin_real : a.Cin;
    const double flops = 2.0 * (double)a.M * a.Cout * cin * a.KT;
    const double bytes = 4.0 * ((double)a.M * cin + (double)a.M * a.Cout + (double)a.Cout * cin * a.KT);
    ProfScope ps(c, c->profile_detail ? std::string("conv_gemm:") + tag : std::string("conv_gemm"), flops, bytes);
    ProfScope ps32(c, "conv_gemm_f32", flops, bytes);
    if (a.Cout > 64) hipLaunchKernelGGL((k_conv_narrow<3>), dim3(grid), dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL((k_conv_narrow<2>), dim3(grid), dim3(256), 0, c->stream, a);
    KCHECK(c);
    return SD_OK;
}
