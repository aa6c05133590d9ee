// conv_gemm_h.hip -- the fp16 mode's wide layers (Cout >= 256: block0, TDNN, MFA, ASP conv) on a 256 x 256 tile.
//
// conv_gemm.hip's fp16 instantiation keeps the f32 kernel's geometry: 128 x 128 tile, 4 waves of 64 x 64.  With
// v_mfma_f32_32x32x16_f16 a K-step of 64 halves is only 16 MFMAs = 512 cycles per wave, and every MFMA needs one
// ds_read_b128 of operand fragments: that kernel is bound by LDS traffic and by the L2 round trip of its load stream, not
// by the matrix pipe (profiles/r02_layer_profile.txt).  Here one workgroup of 8 waves owns 256 x 256 outputs and each wave
// 64 rows x 128 columns (2 x 4 MFMA tiles, 128 accumulator registers): 6 fragment reads feed 8 MFMAs (0.75 per MFMA
// instead of 1), every staged byte feeds twice as many MFMAs (half the LDS writes and half the L2 reads per FLOP), and a
// K-step is 32 MFMAs per wave between barriers.  LDS: 2 stages x (256 + 256) rows x 144 B = 144 KB, one workgroup per CU.
//
// Everything else is conv_gemm.hip's design: channels-last fp16 rows in the compact row space (rowtab), weights
// [tap][Cout][CinPad16] fp16, buffer loads with the K position as scalar offset, a load stream one K-step ahead that runs
// across tile boundaries, persistent PM x PN super-blocks per XCD, f32 epilogue (bias, ReLU / leaky, folded BN) with the
// 4 x 4 quad transpose and one rounding to fp16 at the store.  Only what the ECAPA layers need is implemented: compact row
// space, "same" reflect padding, no second input, no per-item bias, no tanh / sigmoid.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#define HM 256
#define HN 256
#define HLDP 36            // LDS row: 64 halves + 8 halves of padding = 36 floats = 144 B (conflict-free ds_read_b128, see conv_gemm.hip)

// P = 1: fp16 operands (v_mfma_f32_32x32x16_f16, one MFMA per tile and k-block of 16); P = 0: f32 operands
// (v_mfma_f32_32x32x2_f32 over the k pairs (k, k + 16) of a 32-chunk, four MFMAs per tile and K-group, conv_gemm.hip's order)
// P = 3 ("x3", option ecapa_precision = 3): f32 tensors in HBM, 22-bit operands on the fp16 matrix pipe.  Every f32 activation is split when it
// is staged, hi = fp16(a), lo = fp16(a - hi); the weights come split (and scaled by a power of two, so that the lo plane stays out of the
// fp16 subnormals) from weights.cpp: a K-step is 32 channels = a 128-byte LDS row [hi 0..31 | lo 0..31] on both sides, and every 16-channel
// block runs hi*hi + lo*hi + hi*lo (the lo*lo term, 2^-22 of the product, is dropped): 48 MFMAs of 32 cycles per wave and K-step where the
// f32 form needs 128 of 64 cycles, with the f32 form's bytes.  f32 accumulation, f32 epilogue (times 1 / weight scale), f32 stores.
template <int P>
__global__ __launch_bounds__(512) void k_conv_gemm_w256(ConvArgs a)
{
    constexpr bool H = P == 1, X = P == 3;
    constexpr int ES = H ? 2 : 4;                    // bytes per activation element; a K-step is 128 bytes of every row in all three forms
    constexpr int ESB = (H || X) ? 2 : 4;            // bytes per weight element
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const As0 = lds;                          // [2][HM * HLDP]
    float* const Bs0 = lds + 2 * HM * HLDP;          // [2][HN * HLDP]

    const int w = blockIdx.x, G = gridDim.x;         // G is a multiple of 8
    const int xcd = w & 7, wl = w >> 3, wpx = G >> 3;
    const int mx = (a.m_tiles - xcd + 7) >> 3;       // row panels of this XCD: m = xcd + 8 j
    // super-block shape (profiles/r02_layer_profile.txt): with 32 workgroups per XCD, 4 column tiles x 8 row panels beats 8 x 4 by a third
    // on the 3072 x 3072 layer (each W K-slice is shared by 8 workgroups instead of 4; f32 137 vs 103 TF, fp16 962 vs 778 TF)
    const int pnmax = a.sched > 0 ? a.sched : 4;
    const int PN = a.n_tiles < pnmax ? a.n_tiles : pnmax;
    const int PM = wpx / PN > 0 ? wpx / PN : 1;
    const int pm = wl / PN, pn = wl - pm * PN;
    if (pm >= PM) return;
    const int n_groups = (a.n_tiles + PN - 1) / PN, m_groups = (mx + PM - 1) / PM;
    const int sb_end = n_groups * m_groups;
    auto sb_valid = [&](int sb, int& j, int& nt) -> bool {
        const int mg = sb / n_groups, ng = sb - mg * n_groups;
        j = mg * PM + pm; nt = ng * PN + pn;
        return j < mx && nt < a.n_tiles;
    };
    auto next_sb = [&](int sb) -> int {
        int j, nt;
        for (++sb; sb < sb_end; ++sb) if (sb_valid(sb, j, nt)) return sb;
        return sb_end;
    };
    const int q0 = next_sb(-1);
    if (q0 >= sb_end) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;           // wave tile: rows wr * 64, columns wc * 128
    const int c4 = tid & 7, r0 = tid >> 3;           // loader: 16-byte chunk c4 of row r0 + 64 p
    const int li = lane & 31, lh = lane >> 5;

    const int kcs = a.Cin / (128 / ES);
    const int S = a.KT * kcs;
    const int ktr = a.kt_real > 0 ? a.kt_real : a.KT;      // taps that shift rows (split-weight mode: KT = 2 * ktr planes)
    const int half = ktr / 2;
    const size_t in_rows = (size_t)(a.in_rows > 0 ? a.in_rows : a.M);

    // activation loader.  f32 / fp16: 16-byte chunk c4 of rows r0 + 64 p, p < 4.  x3: the 32-byte pair c8 (channels 8 c8 .. + 7) of rows
    // r8 + 128 p, p < 2 -- eight channels per thread make one 16-byte LDS write of hi halves and one of lo halves
    constexpr int NPA = X ? 2 : 4;
    const int c8 = tid & 3, r8 = tid >> 2;
    auto a_row = [&](int p) { return X ? r8 + 128 * p : r0 + 64 * p; };
    int rrel[NPA], tt[NPA], nd[NPA];
    unsigned voA[NPA], voB[4];
    int2 pre[NPA]; int pre_base = 0;
    auto prefetch_tab = [&](int sb) {
        int j, nt;
        (void)sb_valid(sb, j, nt);
        const int m0 = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * HM);
        pre_base = a.rowtab[m0 < a.M ? m0 : a.M - 1].x;
#pragma unroll
        for (int p = 0; p < NPA; ++p) { int g = m0 + a_row(p); if (g > a.M - 1) g = a.M - 1; pre[p] = a.rowtab[g]; }
    };
    auto make_rsrc = [&](const void* base, size_t bytes) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t rA = make_rsrc(a.X, 0);
    const __amdgpu_buffer_rsrc_t rB = make_rsrc(X ? a.W16x : H ? a.W16 : (const void*)a.W, (size_t)a.KT * a.Cout * a.w_ld * ESB);
#pragma unroll
    for (int p = 0; p < 4; ++p) voB[p] = (unsigned)((r0 + 64 * p) * a.w_ld * ESB + c4 * 16);
    int l_q = q0, l_kk = 0, l_kc = 0, m0l = 0, n0l = 0;
    unsigned sK = 0, sB = 0;
    auto set_tile = [&](int sb) {
        int j, nt;
        (void)sb_valid(sb, j, nt);
        m0l = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * HM);
        n0l = __builtin_amdgcn_readfirstlane(nt * HN);
        const int base = __builtin_amdgcn_readfirstlane(pre_base);
#pragma unroll
        for (int p = 0; p < NPA; ++p) { rrel[p] = pre[p].x - base; tt[p] = ROWTAB_T(pre[p].y); nd[p] = ROWTAB_LAST(pre[p].y); }
        rA = make_rsrc((const char*)a.X + (size_t)base * a.x_ld * ES, (in_rows - base) * a.x_ld * ES);
    };
    auto set_tap = [&](int kk) {
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
            int qr = tt[p] + ((kk >= ktr ? kk - ktr : kk) - half) * a.dil;
            if (qr < 0) qr = -qr;
            if (qr >= a.Tin) qr = 2 * (a.Tin - 1) - qr;
            if (qr < 0) qr = 0;
            if (qr > nd[p]) qr = nd[p];
            voA[p] = (unsigned)(rrel[p] + qr) * (unsigned)a.x_ld * ES + (X ? c8 * 32 : c4 * 16);
        }
        // weight rows beyond Cout (a 256-wide tile over Cout = 1024 / 3072 never has any) are clamped by the descriptor
        sB = (unsigned)(((size_t)kk * a.Cout + n0l) * a.w_ld * ESB);
    };
    auto advance = [&]() {
        if (++l_kc < kcs) { sK += 128; return; }
        l_kc = 0; sK = 0;
        if (++l_kk == a.KT) {
            l_kk = 0;
            const int nq = next_sb(l_q);
            if (nq < sb_end) {
                l_q = nq; set_tile(l_q);
                const int nq2 = next_sb(l_q);
                if (nq2 < sb_end) prefetch_tab(nq2);
            }
        }
        set_tap(l_kk);
    };
    f4u ra[4], rb[4];
    auto gload = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if constexpr (X) ra[p] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(rA, voA[p >> 1] + (p & 1) * 16, sK, 0));
            else ra[p] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(rA, voA[p], sK, 0));
            rb[p] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(rB, voB[p], sB + sK, 0));
        }
    };
    auto gload_half = [&](int h) {                    // x3: the loader registers of restaging part h (row r8 + 128 h of A, two W rows)
#pragma unroll
        for (int p = 2 * h; p < 2 * h + 2; ++p) {
            ra[p] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(rA, voA[p >> 1] + (p & 1) * 16, sK, 0));
            rb[p] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(rB, voB[p], sB + sK, 0));
        }
    };
    auto lstore_part = [&](int buf, int p) {          // x3: row r8 + 128 p of A (the split), rows r0 + 64 (2 p), r0 + 64 (2 p + 1) of W
        float* A = As0 + buf * HM * HLDP; float* B = Bs0 + buf * HN * HLDP;
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = ra[2 * p + (e >> 2)][e & 3]; hi[e] = (_Float16)f; lo[e] = (_Float16)(f - (float)hi[e]); }
        // LDS row: [hi 0..7 | lo 0..7 | hi 8..15 | lo 8..15 | hi 16..23 | lo 16..23 | hi 24..31 | lo 24..31]: the 16-byte writes of the
        // eight lanes of an LDS cycle (two rows 144 bytes apart, four 32-byte pairs each) touch 32 different banks
        *(half8*)&A[(r8 + 128 * p) * HLDP + c8 * 8] = hi;
        *(half8*)&A[(r8 + 128 * p) * HLDP + c8 * 8 + 4] = lo;
#pragma unroll
        for (int h = 0; h < 2; ++h) { const int q = 2 * p + h; *(float4*)&B[(r0 + 64 * q) * HLDP + c4 * 4] = make_float4(rb[q][0], rb[q][1], rb[q][2], rb[q][3]); }
    };
    auto lstore = [&](int buf) {
        float* A = As0 + buf * HM * HLDP; float* B = Bs0 + buf * HN * HLDP;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if constexpr (X) {
                if (p < 2) lstore_part(buf, p);
                continue;
            } else
                *(float4*)&A[(r0 + 64 * p) * HLDP + c4 * 4] = make_float4(ra[p][0], ra[p][1], ra[p][2], ra[p][3]);
            *(float4*)&B[(r0 + 64 * p) * HLDP + c4 * 4] = make_float4(rb[p][0], rb[p][1], rb[p][2], rb[p][3]);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // one 16-byte fragment per lane, row and K-group kb (4 groups per K-step).  fp16: lane (li, lh) holds k = 16 kb + 8 lh .. +7 of
    // row li (float offset lh * 4 + kb * 8 of the 144-byte row); f32: k = 16 lh + 4 kb .. +3 (float offset lh * 16 + kb * 4)
    float4 ha[2][2], hb[2][4];
    // x3 fragment kb: 0 / 1 = hi halves of the step's channels 0..15 / 16..31, 2 / 3 = their lo halves (row layout: lstore_part)
    auto xoff = [&](int kb) { return ((kb & 1) * 2 + lh) * 8 + (kb >> 1) * 4; };
    auto afrag = [&](int buf, int kb, int fbuf) {
        const float* Ab = As0 + buf * HM * HLDP + (wr * 64 + li) * HLDP + xoff(kb);
        ha[fbuf][0] = *(const float4*)Ab;
        ha[fbuf][1] = *(const float4*)(Ab + 32 * HLDP);
    };
    auto bfrag = [&](int buf, int kb, int fbuf) {
        const float* Bb = Bs0 + buf * HN * HLDP + (wc * 128 + li) * HLDP + xoff(kb);
#pragma unroll
        for (int j = 0; j < 4; ++j) hb[fbuf][j] = *(const float4*)(Bb + j * 32 * HLDP);
    };
    auto xmma = [&](int fa, int fb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ha[fa][0]), __builtin_bit_cast(half8, hb[fb][j]), acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ha[fa][1]), __builtin_bit_cast(half8, hb[fb][j]), acc[1][j], 0, 0, 0);
        }
    };
    auto hfrag = [&](int buf, int kb, int fbuf) {
        const int ko = (H || X) ? lh * 4 + kb * 8 : lh * 16 + kb * 4;
        const float* Ab = As0 + buf * HM * HLDP + (wr * 64 + li) * HLDP + ko;
        const float* Bb = Bs0 + buf * HN * HLDP + (wc * 128 + li) * HLDP + ko;
        ha[fbuf][0] = *(const float4*)Ab;
        ha[fbuf][1] = *(const float4*)(Ab + 32 * HLDP);
#pragma unroll
        for (int j = 0; j < 4; ++j) hb[fbuf][j] = *(const float4*)(Bb + j * 32 * HLDP);
    };
    auto hmma = [&](int fbuf) {
        if constexpr (H) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ha[fbuf][0]), __builtin_bit_cast(half8, hb[fbuf][j]), acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ha[fbuf][1]), __builtin_bit_cast(half8, hb[fbuf][j]), acc[1][j], 0, 0, 0);
            }
        } else {
            const float a0[4] = {ha[fbuf][0].x, ha[fbuf][0].y, ha[fbuf][0].z, ha[fbuf][0].w}, a1[4] = {ha[fbuf][1].x, ha[fbuf][1].y, ha[fbuf][1].z, ha[fbuf][1].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float bv = e == 0 ? hb[fbuf][j].x : e == 1 ? hb[fbuf][j].y : e == 2 ? hb[fbuf][j].z : hb[fbuf][j].w;
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bv, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bv, acc[1][j], 0, 0, 0);
                }
        }
    };

    // prologue: stage step 0 of the first tile
    prefetch_tab(l_q);
    set_tile(l_q);
    { const int nq2 = next_sb(l_q); if (nq2 < sb_end) prefetch_tab(nq2); }
    set_tap(0);
    int m0c = m0l, n0c = n0l;
    gload();
    lstore(0);
    __syncthreads();
    advance();
    if constexpr (X) { gload(); advance(); afrag(0, 0, 0); bfrag(0, 0, 0); }       // x3: the loader registers hold step 1, the load stream stands at step 2
    else hfrag(0, 0, 0);

    int q = q0, s = 0, buf = 0;
    while (true) {
        // one K-step: fragments of k-block kb + 1 are read while the 8 MFMAs of kb run; the next step's global loads are issued
        // first and restaged to the other LDS buffer during k-block 2; one barrier per step
        // f32: a K-group is 32 MFMAs (2048 cycles).  Left alone the scheduler sinks the six fragment reads of the next group to the
        // end of the region -- one MFMA in front of their first use -- and the eight ds_writes to the last three MFMAs in front of
        // the barrier, so every group boundary and every barrier waits on LDS latency with an empty matrix pipe (13 % of the
        // cycles, profiles/pmc_conv_gemm_bench.json).  sched_group_barrier pins them behind the FIRST MFMAs of the region, one
        // memory instruction per MFMA (masks: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write).
#define W_PAIR(mask, n) do { _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(mask, 1, 0); } } while (0)
        if constexpr (X) {
            // x3: six groups of 8 MFMAs per K-step.  Fragment registers A0 / W0 hold hi block 0 of this step on entry; every group's operands
            // are read one group ahead into the registers the group before last released.  The load stream runs TWO steps ahead: on entry the
            // loader registers hold step s + 1 (restaged in groups 2 and 4, so that the LDS writes have landed long before the barrier behind
            // group 5), and each half of them is refilled with step s + 2 in the group after its restaging (five groups of latency budget).
            // [measured, tools/x3_ablate.sh, MFA layer: 168 ms with loads in group 1 and all restaging in groups 4-5; 142 ms without the
            // global loads, 115 ms without restaging and barrier = the MFMA + fragment-read floor]
            // SD_X3_ABLATE (tools/x3_ablate.sh, never in the product build): 1 = no global loads in the loop, 2 = no restaging / barrier, 3 = both
#ifndef SD_X3_ABLATE
#define SD_X3_ABLATE 0
#endif
            afrag(buf, 2, 1);                      // lo block 0 of A
            xmma(0, 0);                            // hi0 * hi0
            W_PAIR(0x100, 2); __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            __builtin_amdgcn_sched_barrier(0);
            bfrag(buf, 2, 1);                      // lo block 0 of W
            xmma(1, 0);                            // lo0 * hi0
            if (!(SD_X3_ABLATE & 2)) lstore_part(buf ^ 1, 0);
            // the split of a row's eight channels is ~28 VALU instructions, and there are four LDS writes: left alone they all sit behind the
            // group's last MFMA
            W_PAIR(0x100, 4);
#pragma unroll
            for (int i_ = 0; i_ < 3; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 10, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            afrag(buf, 1, 1); bfrag(buf, 1, 0);    // hi block 1 of both
            xmma(0, 1);                            // hi0 * lo0
            if (!(SD_X3_ABLATE & 1)) gload_half(0);
            W_PAIR(0x100, 6);
#pragma unroll
            for (int i_ = 0; i_ < 2; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 2, 0); }
            __builtin_amdgcn_sched_barrier(0);
            afrag(buf, 3, 0);                      // lo block 1 of A
            xmma(1, 0);                            // hi1 * hi1
            if (!(SD_X3_ABLATE & 2)) lstore_part(buf ^ 1, 1);
            W_PAIR(0x100, 2);
#pragma unroll
            for (int i_ = 0; i_ < 4; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 8, 0); }
#pragma unroll
            for (int i_ = 0; i_ < 2; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 2, 0); }
            __builtin_amdgcn_sched_barrier(0);
            bfrag(buf, 3, 1);                      // lo block 1 of W
            xmma(0, 0);                            // lo1 * hi1
            if (!(SD_X3_ABLATE & 1)) gload_half(1);
            W_PAIR(0x100, 4);
#pragma unroll
            for (int i_ = 0; i_ < 2; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 2, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(SD_X3_ABLATE & 2)) __syncthreads();
            afrag(buf ^ 1, 0, 0); bfrag(buf ^ 1, 0, 0);
            xmma(1, 1);                            // hi1 * lo1
            W_PAIR(0x100, 6); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
        hfrag(buf, 1, 1);
        gload();
        hmma(0);
        if constexpr (!H) {
            W_PAIR(0x100, 6);
#pragma unroll
            for (int i_ = 0; i_ < 8; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
        } else {
            // fp16: a K-group is 8 MFMAs of 32 cycles; same rule, the eight loads / stores ride behind the last two MFMAs
            W_PAIR(0x100, 6);
#pragma unroll
            for (int i_ = 0; i_ < 2; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 4, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
        hfrag(buf, 2, 0);
        hmma(1);
        W_PAIR(0x100, 6); __builtin_amdgcn_sched_group_barrier(0x008, H ? 2 : 26, 0);
        __builtin_amdgcn_sched_barrier(0);
        hfrag(buf, 3, 1);
        hmma(0);
        lstore(buf ^ 1);
        if constexpr (!H) { W_PAIR(0x100, 6); W_PAIR(0x200, 8); __builtin_amdgcn_sched_group_barrier(0x008, 18, 0); }
        else {
            W_PAIR(0x100, 6);
#pragma unroll
            for (int i_ = 0; i_ < 2; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 4, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        hfrag(buf ^ 1, 0, 0);
        hmma(1);
        W_PAIR(0x100, 6); __builtin_amdgcn_sched_group_barrier(0x008, H ? 2 : 26, 0);
        __builtin_amdgcn_sched_barrier(0);
        }
        advance();            // (x3: the load stream now stands two steps ahead of the step that starts next)

        if (s == S - 1) {
            // ---- epilogue.  C layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); see conv_gemm.hip
            const float slope = (a.act1 == 1) ? 0.0f : ((a.act1 == 2) ? 0.01f : 1.0f);
            if constexpr (!H) {
                const float as = X ? a.acc_scale : 1.0f;
                // f32 (and x3): every accumulator register is one output row for 32 consecutive columns across a half-wave: stored as it
                // lies, 128 contiguous bytes per row and instruction, no lane transposes (half the epilogue's VALU work).  The
                // descriptor starts at the tile's first row and ends at the batch's last one, so rows >= M are dropped by the range
                // check; the row term travels in the scalar offset.
                const int rows_left = a.M - m0c;
                const __amdgpu_buffer_rsrc_t rY = make_rsrc(a.Y + (size_t)m0c * a.y_ld, (size_t)(rows_left < HM ? rows_left : HM) * a.y_ld * 4);
                const unsigned ybytes = (unsigned)a.y_ld * 4u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cc = n0c + wc * 128 + j * 32 + li;
                    const float cb = a.bias ? a.bias[cc] : 0.0f;
                    const float cs = a.scale ? a.scale[cc] : 1.0f, ch = a.scale ? a.shift[cc] : 0.0f;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const unsigned vo = (unsigned)(wr * 64 + i * 32 + 4 * lh) * ybytes + (unsigned)cc * 4u;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = X ? acc[i][j][r] * as + cb : acc[i][j][r] + cb;
                            acc[i][j][r] = 0.0f;
                            v = fmaxf(v, v * slope);
                            v = v * cs + ch;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rY, vo, (unsigned)((r & 3) + 8 * (r >> 2)) * ybytes, 0);
                        }
                    }
                }
            } else {
            const int lq = lane & 3;
            _Float16* const Y = (_Float16*)a.Y;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = n0c + wc * 128 + j * 32 + li;
                const int ccl = cc < a.Cout ? cc : a.Cout - 1;
                const float cb = a.bias ? a.bias[ccl] : 0.0f;
                const float cs = a.scale ? a.scale[ccl] : 1.0f, ch = a.scale ? a.shift[ccl] : 0.0f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float x[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[i][j][4 * gq + e] + cb;
                            acc[i][j][4 * gq + e] = 0.0f;
                            v = fmaxf(v, v * slope);
                            x[e] = v * cs + ch;
                        }
                        // 4 x 4 transpose across the lane quad (two butterfly stages on DPP quad_perm)
                        float s0 = (lq & 1) ? x[0] : x[1];
                        float s1 = (lq & 1) ? x[2] : x[3];
                        float r0_ = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s0), 0xB1, 0xF, 0xF, true));
                        float r1_ = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s1), 0xB1, 0xF, 0xF, true));
                        if (lq & 1) { x[0] = r0_; x[2] = r1_; } else { x[1] = r0_; x[3] = r1_; }
                        s0 = (lq & 2) ? x[0] : x[2];
                        s1 = (lq & 2) ? x[1] : x[3];
                        r0_ = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s0), 0x4E, 0xF, 0xF, true));
                        r1_ = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s1), 0x4E, 0xF, 0xF, true));
                        if (lq & 2) { x[0] = r0_; x[1] = r1_; } else { x[2] = r0_; x[3] = r1_; }
                        const int g = m0c + wr * 64 + i * 32 + 8 * gq + 4 * lh + lq;
                        const int co = n0c + wc * 128 + j * 32 + (li & ~3);
                        if (g < a.M && co < a.Cout) {
                            if (a.y_f32) *(float4*)(a.Y + (size_t)g * a.y_ld + co) = make_float4(x[0], x[1], x[2], x[3]);
                            else {
                                const half4 hv = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
                                *(half4*)(Y + (size_t)g * a.y_ld + co) = hv;
                            }
                        }
                    }
                }
            }
            }
            q = next_sb(q);
            if (q >= sb_end) break;
            { int j_, nt_; (void)sb_valid(q, j_, nt_); m0c = __builtin_amdgcn_readfirstlane((xcd + 8 * j_) * HM); n0c = __builtin_amdgcn_readfirstlane(nt_ * HN); }
            s = 0;
        } else {
            ++s;
        }
        buf ^= 1;
    }
}

// returns 1 when the layer does not fit this kernel (the caller then uses conv_gemm.hip's 128 x 128 form)
int launch_conv_gemm_h256(sd_ctx* c, const ConvArgs& in, const char* tag)
{
    ConvArgs a = in;
    const bool h = a.prec == 1, x3 = a.prec == 3;
    if (!a.rowtab || (h && !a.W16) || (x3 && !a.W16x) || a.X2 || a.item_bias || a.R || a.act2 || a.pad_mode != 0 || a.Cout < 256 || (a.Cout % HN) != 0 || (a.y_ld & 3) ||
        a.Cin % (h ? 64 : 32) != 0 || (a.M < 8 * HM && !x3)) return 1;      // (x3 takes every batch size: the arithmetic of a layer must not depend on it)
    // fp16: the shortest contraction (ASP conv, K = 128) stays on the 128 x 128 form (measured); f32 takes every wide layer since the
    // K-groups are pinned (block0 K = 400: 92 -> 102 TF, ASP conv K = 128: 95 -> 105 TF)
    if (!x3 && (int64_t)a.Cin * (a.kt_real > 0 ? a.kt_real : a.KT) < (c->conv_w256_kmin > 0 ? c->conv_w256_kmin : (h ? 256 : 128))) return 1;
    const size_t lds_bytes = (size_t)2 * (HM + HN) * HLDP * sizeof(float);
    const unsigned dev_bit = 1u << (c->device & 31);
    if (!(g_attr_w256.load(std::memory_order_acquire) & dev_bit)) {        // once per device (a second thread that gets here meanwhile sets the same values)
        if (hipFuncSetAttribute((const void*)k_conv_gemm_w256<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv_gemm_w256<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv_gemm_w256<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 1; }
        g_attr_w256.fetch_or(dev_bit, std::memory_order_release);
    }
    if (a.w_ld <= 0) a.w_ld = a.Cin;
    if (x3) a.w_ld = 2 * a.Cin;              // halves: eight hi, eight lo, eight hi, ... (weights.cpp)
    a.m_tiles = (a.M + HM - 1) / HM;
    a.n_tiles = (a.Cout + HN - 1) / HN;
    a.sched = c->conv_pn;
    int grid = (c->num_cu / 8) * 8;
    if (grid < 8) grid = 8;
    const int lx_max = ((a.m_tiles + 7) / 8) * a.n_tiles;
    if (grid / 8 > lx_max) grid = lx_max * 8;
    const int cin = a.cin_real > 0 ? a.cin_real : a.Cin;
    const double flops = 2.0 * (double)a.M * a.Cout * cin * (a.kt_real > 0 ? a.kt_real : a.KT);
    const double bytes = (h ? 2.0 : 4.0) * ((double)a.M * cin + (double)a.M * a.Cout + (double)a.Cout * cin * a.KT);
    {
        ProfScope ps(c, c->profile_detail ? std::string("conv_gemm:") + tag : std::string("conv_gemm"), flops, bytes);
        ProfScope ps16(c, h ? "conv_gemm_f16" : x3 ? "conv_gemm_x3" : "conv_gemm_f32", flops, bytes);
        ProfScope psw(c, h ? "conv_w256_f16" : x3 ? "conv_w256_x3" : "conv_w256_f32", flops, bytes);         // this kernel alone (bench.py's roofline object)
        ProfScope pss(c, strcmp(tag, "lstm_ih") == 0 ? "conv_w256_seg" : "conv_w256_ecapa", flops, bytes);   // ... split by caller (PyanNet's K = 256 projections / the ECAPA layers)
        if (h) hipLaunchKernelGGL(k_conv_gemm_w256<1>, dim3(grid), dim3(512), lds_bytes, c->stream, a);
        else if (x3) hipLaunchKernelGGL(k_conv_gemm_w256<3>, dim3(grid), dim3(512), lds_bytes, c->stream, a);
        else hipLaunchKernelGGL(k_conv_gemm_w256<0>, dim3(grid), dim3(512), lds_bytes, c->stream, a);
    }
    if (hipGetLastError() != hipSuccess) SD_FAIL(c, SD_ERR_HIP, "k_conv_gemm_w256 launch failed (%s)", tag);
    return SD_OK;
}
