// conv_gemm_g.hip -- the fp16 mode's wide layers (Cout >= 256) with LDS-DMA staging (round 4).
//
// conv_gemm_h.hip's fp16 form stages its operands through registers (buffer_load -> VGPRs -> ds_write, one K-step ahead).  Here
// `buffer_load_dwordx4 ... lds` writes them straight into LDS: no staging registers, no ds_write instructions, no VALU, and the
// request for step s + 2 goes out the moment its stage is released (right behind the barrier inside step s) with a whole K-step to land.
// What that bought, measured (profiles/r04_g256_ablation.txt): nothing on the big layers (MFA 991 -> 1 000 TF), 5 - 10 % on block0 -- the
// loop is not bound by the issue -> land latency, as round 3 believed, but by the CU's vector-memory INGEST rate: 64 KB per K-step
// arrive at 16 - 17 bytes per clock and CU whichever way they are requested (the same loop without MFMAs takes the same 3 900 cycles
// per step; without the DMAs 2 257 = the matrix pipe 91 % busy).  With a 256 x 256 tile the fp16 pipe needs 32 B / clk / CU: the kernel
// runs at ~ 85 % of what its memory path allows.  It stays the default fp16 form (fewer registers and instructions, same bits).
//
// Geometry as before: 256 x 256 tile, K-step of 64 halves (128 bytes of every row), 8 waves of 64 x 128 (2 x 4 tiles of
// v_mfma_f32_32x32x16_f16, 128 accumulator registers), persistent super-blocks per XCD, the load stream runs across tile boundaries.
// LDS: 2 stages x (256 + 256) rows x 128 B = 128 KB (+ 32 KB of epilogue strips = the CU's 160 KB).  An LDS-DMA wave instruction
// writes 64 x 16 consecutive bytes = eight whole rows, so rows cannot be padded; bank conflicts of the fragment reads are removed by a
// swizzle instead: the 16-byte chunk c of row R lives at chunk position c ^ ((R >> 1) & 7).  Every 16-lane group of a ds_read_b128
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... MI355X_MICROARCH.md, LDS) then reads sixteen different (128-byte half, chunk) slots
// of the 256-byte bank row: conflict-free.  The swizzle is applied on the SOURCE side (the DMA's per-lane global address picks the
// logical chunk its LDS slot holds) and in the fragment address.
//
// Ordering (MI355X_MICROARCH.md item 7, cdna_hip_programming.md "Read a staged buffer one phase AFTER the wait that retires it"):
// a wave waits for its own DMAs (vmcnt) and for its outstanding fragment reads of the buffer about to be re-filled (lgkmcnt(0)),
// then joins a raw s_barrier; the next buffer is read, and the released one re-filled, only behind that barrier.
// Same arithmetic as the register-staged form: every output element sums the same products in the same order, so the two kernels
// give the same bits (tests/test_gpu_parity.py::test_ecapa_fp16_lds_dma_staged_kernel_gives_the_same_bits).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#define GM 256
#define GN 256
#define G_STAGE 65536          // bytes per stage: 256 A rows + 256 B rows of 128 B
#define G_BOFF 32768
#define G_STRIP 4096           // epilogue strip per wave: 16 rows x 128 channels of fp16 (128 + 32 KB = all of the CU's LDS)

#ifndef SD_G_ABLATE
#define SD_G_ABLATE 0          // diagnostic builds only: 1 no output stores, 2 no DMA in the loop, 4 no MFMA, 8 no A-operand DMA, 16 no W-operand DMA, 32 no fragment reads, 64 the DMAs of a K-step issued by four waves instead of eight
#endif
typedef __attribute__((address_space(3))) char lds_char;
typedef int v4i __attribute__((ext_vector_type(4)));

// One LDS-DMA wave instruction: 64 x 16 bytes from the buffer `rs` (per-lane byte offset `vo`, scalar byte offset `so`) to the 1 KB of LDS at
// `ldsaddr`.  Written as inline assembly on purpose: hipcc's waitcnt pass treats an LDS-DMA it knows about as a pending LDS store that ANY later
// ds_read may alias and puts `s_waitcnt vmcnt(0)` in front of the next fragment read -- which would serialise exactly the overlap this
// kernel exists for.  The waits that order these DMAs against the fragment reads are the explicit ones at the barrier (see the loop).
// (`s_nop 4`: the descriptor words often come straight from v_readfirstlane, and an SGPR written by the vector ALU needs five wait states before a
// vector-memory instruction reads it; hipcc counts them for its own instructions, not inside an asm string.)
__device__ __forceinline__ void lds_dma_b32(v4i rs, unsigned ldsaddr, unsigned vo)       // 64 x 4 bytes -> 256 bytes of LDS
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dword %1, %2, 0 offen lds" :: "s"(ldsaddr), "v"(vo), "s"(rs) : "memory");
}
__device__ __forceinline__ void lds_dma_b128(v4i rs, unsigned ldsaddr, unsigned vo, unsigned so)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(ldsaddr), "v"(vo), "s"(rs), "s"(so) : "memory");      // (m0 is reserved: the compiler does not keep values in it across statements)
}

// P = 1: fp16 tensors (v_mfma_f32_32x32x16_f16, a K-step is 64 halves).  P = 0: f32 tensors (v_mfma_f32_32x32x2_f32 over the k pairs (k, k + 16)
// of a 32-float K-step, conv_gemm.hip's order: same bits as the 128 x 128 kernel and as conv_gemm_h.hip's register-staged form).  Either way a
// K-step is 128 bytes of every row, so stages, swizzle, DMA pieces and the load stream are the same.
template <int P>
__global__ __launch_bounds__(512) void k_conv_gemm_g256(ConvArgs a)
{
    constexpr int ES = P == 0 ? 4 : 2;               // bytes per element of X, W, Y
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int w = blockIdx.x, G = gridDim.x;         // G is a multiple of 8
    const int xcd = w & 7, wl = w >> 3, wpx = G >> 3;
    const int mx = (a.m_tiles - xcd + 7) >> 3;       // row panels of this XCD: m = xcd + 8 j
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
    const int li = lane & 31, lh = lane >> 5;

    const int kcs = a.Cin / (128 / ES);
    const int S = a.KT * kcs;
    const int ktr = a.kt_real > 0 ? a.kt_real : a.KT;      // taps that shift rows (split-weight mode: KT = 2 * ktr planes)
    const int half = ktr / 2;
    const size_t in_rows = (size_t)(a.in_rows > 0 ? a.in_rows : a.M);

    // loader: wave `wid` fills rows [32 wid, 32 wid + 32) of both operands, eight rows per DMA instruction; lane l writes LDS bytes
    // [16 l, 16 l + 16) of the instruction's 1 KB: row Rp = 32 wid + 8 p + (l >> 3), chunk POSITION l & 7, which holds the logical
    // chunk (l & 7) ^ ((Rp >> 1) & 7)
    const int lrow = lane >> 3, pc = lane & 7;
    auto l_row = [&](int p) { return wid * 32 + p * 8 + lrow; };
    auto l_chunk = [&](int p) { return pc ^ ((((p & 1) << 2) + (lane >> 4)) & 7); };          // ((Rp >> 1) & 7) = 4 (p & 1) + (lane >> 4)
    // Activation rows (option conv_rot, same LDS image): piece p of wave `wid` covers rows 64 ((rot + p) & 3) + 8 wid + (l >> 3), so the
    // eight waves' p-th pieces together are one 64-row quarter of the tile, and the PN workgroups that share a row panel (same XCD, same
    // K-step) start with DIFFERENT quarters: each line of the panel is first requested by one of them and is on its way (or in the L2)
    // when the others ask -- a CU keeps ~16 KB of requests in flight, so what a K-step costs is the latency its lines see.  Measured on the
    // planted hour (profiles/r04_g256_request_order.txt): tdnn 749 -> 812 TF on one box, 883 -> 922 on another; MFA + 1 %.  [Requesting the own
    // quarter of step s + 3 one step early on top of it (into the idle strip area, only to warm the L2): no further gain.]
    const int rot = (a.stagger & 1) ? pn : 0;
    // bit 1: a wave's pieces 0, 1 (and 2, 3) lie 32 rows apart in ONE quarter, waves 0-3 and 4-7 take different quarters.  With the DMAs alone in the
    // loop (no MFMA, no fragment reads) a K-step of the tdnn layers takes 3 390 cycles in this order against 4 180 (MFA 2 910 / 3 400); the whole
    // kernel gains 2 - 5 % on tdnn (profiles/r04_g256_request_order.txt, section 6)
    const bool pairmap = (a.stagger & 2) != 0;
    auto qidx = [&](int p) { return pairmap ? (wid >> 2) + 2 * (p >> 1) : p; };
    auto rinq = [&](int p) { return pairmap ? 8 * (wid & 3) + 32 * (p & 1) : 8 * wid; };
    auto a_quarter = [&](int p) { return (rot + qidx(p)) & 3; };
    auto l_rowA = [&](int p) { return (a.stagger & 1) ? a_quarter(p) * 64 + rinq(p) + lrow : l_row(p); };
    auto l_chunkA = [&](int p) { return (a.stagger & 1) ? pc ^ ((((wid & 1) << 2) + (lane >> 4)) & 7) : l_chunk(p); };
    // the same for the weight rows among the PM workgroups that share a column panel (MFA's 6 MB column panel does not stay in a 4 MB L2: + 1.5 - 2 % there)
    const bool rotB = (a.stagger & 1) != 0;
    auto b_quarter = [&](int p) { return (pm + qidx(p)) & 3; };
    auto l_rowB = [&](int p) { return rotB ? b_quarter(p) * 64 + rinq(p) + lrow : l_row(p); };
    auto l_chunkB = [&](int p) { return rotB ? pc ^ ((((wid & 1) << 2) + (lane >> 4)) & 7) : l_chunk(p); };
    auto b_lds = [&](int p) { return rotB ? (unsigned)(b_quarter(p) * 8192 + rinq(p) * 128) : (unsigned)(wid * 4096 + p * 1024); };
    auto a_lds = [&](int p) { return (a.stagger & 1) ? (unsigned)(a_quarter(p) * 8192 + rinq(p) * 128) : (unsigned)(wid * 4096 + p * 1024); };
    unsigned a_off[4], b_off[4];          // wave-uniform LDS offsets of the pieces inside a stage (SGPRs)
#pragma unroll
    for (int p = 0; p < 4; ++p) { a_off[p] = __builtin_amdgcn_readfirstlane(a_lds(p)); b_off[p] = __builtin_amdgcn_readfirstlane((unsigned)G_BOFF + b_lds(p)); }
    int rrel[4], tt[4], nd[4];
    unsigned voA[4], voB[4];
    int2 pre[4]; int pre_base = 0;
    auto prefetch_tab = [&](int sb) {
        int j, nt;
        (void)sb_valid(sb, j, nt);
        const int m0 = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * GM);
        pre_base = a.rowtab[m0 < a.M ? m0 : a.M - 1].x;
#pragma unroll
        for (int p = 0; p < 4; ++p) { int g = m0 + l_rowA(p); if (g > a.M - 1) g = a.M - 1; pre[p] = a.rowtab[g]; }
    };
    // raw buffer resource {base[31:0], base[47:32] (stride 0), bytes, flags} -- the words __builtin_amdgcn_make_buffer_rsrc builds
    auto make_rsrc = [&](const void* base, size_t bytes) {
        const unsigned long long b = (unsigned long long)base;
        v4i r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffffu));
        r[2] = __builtin_amdgcn_readfirstlane((int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes));
        r[3] = 0x00020000;
        return r;
    };
    v4i rA = make_rsrc(a.X, 0);
    const v4i rB = make_rsrc(P == 0 ? (const void*)a.W : a.W16, (size_t)a.KT * a.Cout * a.w_ld * ES);
#pragma unroll
    for (int p = 0; p < 4; ++p) voB[p] = (unsigned)(l_rowB(p) * a.w_ld * ES + l_chunkB(p) * 16);
    int l_q = q0, l_kk = 0, l_kc = 0, m0l = 0, n0l = 0;
    unsigned sK = 0, sB = 0;
    auto set_tile = [&](int sb) {
        int j, nt;
        (void)sb_valid(sb, j, nt);
        m0l = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * GM);
        n0l = __builtin_amdgcn_readfirstlane(nt * GN);
        const int base = __builtin_amdgcn_readfirstlane(pre_base);
#pragma unroll
        for (int p = 0; p < 4; ++p) { rrel[p] = pre[p].x - base; tt[p] = ROWTAB_T(pre[p].y); nd[p] = ROWTAB_LAST(pre[p].y); }
        rA = make_rsrc((const char*)a.X + (size_t)base * a.x_ld * ES, (in_rows - base) * a.x_ld * ES);
    };
    auto set_tap = [&](int kk) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int qr = tt[p] + ((kk >= ktr ? kk - ktr : kk) - half) * a.dil;
            if (qr < 0) qr = -qr;
            if (qr >= a.Tin) qr = 2 * (a.Tin - 1) - qr;
            if (qr < 0) qr = 0;
            if (qr > nd[p]) qr = nd[p];
            voA[p] = (unsigned)(rrel[p] + qr) * (unsigned)a.x_ld * ES + (unsigned)l_chunkA(p) * 16;
        }
        sB = (unsigned)(((size_t)kk * a.Cout + n0l) * a.w_ld * ES);
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
    // eight LDS-DMA instructions per wave and K-step: four 1 KB pieces of A, four of B
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)lds;
    auto dma_piece = [&](int st, int p) {            // piece p of 8: A pieces 0..3, W pieces 4..7
        const unsigned stb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(st * G_STAGE));
        if (p < 4) lds_dma_b128(rA, stb + a_off[p], voA[p], sK);
        else lds_dma_b128(rB, stb + b_off[p - 4], voB[p - 4], sB + sK);
    };
    auto dma = [&](int st) {
        const unsigned stb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(st * G_STAGE));
        if (SD_G_ABLATE & (64 | 128 | 256)) {   // ablation (diagnostic builds only; meaningful with 4 + 32): the 64 DMAs of a K-step issued by 4 / 2 / 1 waves instead of eight
            constexpr int NW = (SD_G_ABLATE & 64) ? 4 : (SD_G_ABLATE & 128) ? 2 : 1;
            if (wid < NW) {
#pragma unroll
                for (int it = 0; it < 4 * (8 / NW); ++it) {
                    const int k = (SD_G_ABLATE & 512) ? it % (8 / NW) : it / 4, p = (SD_G_ABLATE & 512) ? it / (8 / NW) : it % 4;      // 512: piece-major order
                    lds_dma_b128(rA, stb + a_off[p] + k * NW * 1024, voA[p] + (unsigned)(k * NW * 8) * (unsigned)a.x_ld * ES, sK);
                    lds_dma_b128(rB, stb + b_off[p] + k * NW * 1024, voB[p] + (unsigned)(k * NW * 8) * (unsigned)a.w_ld * ES, sB + sK);
                }
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!(SD_G_ABLATE & 8)) lds_dma_b128(rA, stb + a_off[p], voA[p], sK);
            if (!(SD_G_ABLATE & 16)) lds_dma_b128(rB, stb + b_off[p], voB[p], sB + sK);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragments.  fp16: lane (li, lh) reads k = 16 kb + 8 lh .. + 7 (logical chunk 2 kb + lh) of row li of every 32-row block;
    // f32: k = 16 lh + 4 kb .. + 3 (logical chunk 4 lh + kb), the two k of each v_mfma_f32_32x32x2_f32 sixteen apart
    const int swz = (li >> 1) & 7;
    const char* const Afr = lds + (wr * 64 + li) * 128;
    const char* const Bfr = lds + G_BOFF + (wc * 128 + li) * 128;
    float4 ha[2][2], hb[2][4];
    // P = 2: v_mfma_f32_16x16x32_f16 (the chip holds 2.09 GHz on it against 1.79 on the 32x32x16 shape, tools/mfma_shape_clock.hip).  The wave's
    // 64 x 128 tile is 4 x 8 blocks of 16 x 16 (acc16, 128 registers); a K-step is two k-groups of 32 halves, each walked in two phases of
    // 4 x 4 blocks: phase kb = 2 g + h reads the weight fragments of blocks 4 h .. 4 h + 3 (and, for h = 0, the four activation fragments
    // of group g, kept for both phases).  Lane l reads k = 32 g + 8 (l >> 4) .. + 7 (logical chunk 4 g + (l >> 4)) of row l & 15 of a block;
    // with the stage's swizzle every 16-lane group of the ds_read_b128 covers the sixteen (128-byte half, chunk) slots once.
    f32x4 acc16[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc16[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int l15 = lane & 15, l4 = lane >> 4;
    const char* const Afr16 = lds + (wr * 64 + l15) * 128;
    const char* const Bfr16 = lds + G_BOFF + (wc * 128 + l15) * 128;
    float4 qa[2][4], qb[2][4];
    auto hfrag = [&](int st, int kb, int fbuf) {
        if (SD_G_ABLATE & 32) return;   // ablation (diagnostic builds only): no fragment reads, the matrix pipe runs on whatever the registers hold
        if constexpr (P == 2) {
            const int g = kb >> 1, h = kb & 1;
            const int co = ((4 * g + l4) ^ ((l15 >> 1) & 7)) * 16 + st * G_STAGE;
            if (h == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) qa[g][i] = *(const float4*)(Afr16 + co + i * 16 * 128);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) qb[fbuf][jj] = *(const float4*)(Bfr16 + co + (4 * h + jj) * 16 * 128);
            return;
        }
        const int co = ((P == 0 ? 4 * lh + kb : 2 * kb + lh) ^ swz) * 16 + st * G_STAGE;
        ha[fbuf][0] = *(const float4*)(Afr + co);
        ha[fbuf][1] = *(const float4*)(Afr + co + 32 * 128);
#pragma unroll
        for (int j = 0; j < 4; ++j) hb[fbuf][j] = *(const float4*)(Bfr + co + j * 32 * 128);
    };
    auto hmma = [&](int fbuf, int kb) {
        if constexpr (P == 2) {
            if (SD_G_ABLATE & 4) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { asm volatile("" :: "v"(qb[fbuf][jj].x), "v"(qb[fbuf][jj].w), "v"(qa[kb >> 1][jj].x), "v"(qa[kb >> 1][jj].w)); }
                return;
            }
            const int g = kb >> 1, h = kb & 1;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc16[i][4 * h + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, qa[g][i]), __builtin_bit_cast(half8, qb[fbuf][jj]), acc16[i][4 * h + jj], 0, 0, 0);
            return;
        }
        if (SD_G_ABLATE & 4) {          // ablation (diagnostic builds only): fragment reads without the matrix pipe
#pragma unroll
            for (int j = 0; j < 4; ++j) { asm volatile("" :: "v"(hb[fbuf][j].x), "v"(hb[fbuf][j].y), "v"(hb[fbuf][j].z), "v"(hb[fbuf][j].w)); }
            asm volatile("" :: "v"(ha[fbuf][0].x), "v"(ha[fbuf][0].w), "v"(ha[fbuf][1].x), "v"(ha[fbuf][1].w));
            return;
        }
        if constexpr (P == 0) {
            // f32: the WEIGHT fragment is the MFMA's first operand, so the accumulator tile comes out transposed -- register r of acc[i][j],
            // lane (li, lh) = activation row 32 i + li, output channel 32 j + 8 (r >> 2) + 4 lh + (r & 3): four registers are four
            // consecutive channels of one row = one 16-byte store.  Same products in the same order: same bits as the other operand order.
            const float a0[4] = {ha[fbuf][0].x, ha[fbuf][0].y, ha[fbuf][0].z, ha[fbuf][0].w}, a1[4] = {ha[fbuf][1].x, ha[fbuf][1].y, ha[fbuf][1].z, ha[fbuf][1].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float bv = e == 0 ? hb[fbuf][j].x : e == 1 ? hb[fbuf][j].y : e == 2 ? hb[fbuf][j].z : hb[fbuf][j].w;
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, a0[e], acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, a1[e], acc[1][j], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ha[fbuf][0]), __builtin_bit_cast(half8, hb[fbuf][j]), acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ha[fbuf][1]), __builtin_bit_cast(half8, hb[fbuf][j]), acc[1][j], 0, 0, 0);
            }
        }
    };
    // f32: the tile's 256 bias / BN scale / BN shift values go to LDS by LDS-DMA during the tile's LAST K-step (waves 0-3, 64 channels each, no
    // registers); that step's vmcnt(0) + barrier make them visible to the epilogue.  The area lies behind the stages (the fp16 form's strips).
    const v4i rPb = make_rsrc(a.bias, (size_t)a.Cout * 4), rPs = make_rsrc(a.scale, (size_t)a.Cout * 4), rPh = make_rsrc(a.shift, (size_t)a.Cout * 4);
    const float* const Ps = (const float*)(lds + 2 * G_STAGE);

    // prologue: steps 0 and 1 of the first tile are requested; step 0 has to be there
    prefetch_tab(l_q);
    set_tile(l_q);
    { const int nq2 = next_sb(l_q); if (nq2 < sb_end) prefetch_tab(nq2); }
    set_tap(0);
    int m0c = m0l, n0c = n0l;
    dma(0);
    advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    dma(1);
    advance();
    hfrag(0, 0, 0);

    int q = q0, s = 0, buf = 0;
#ifdef SD_G_STAMPS        // diagnostic build only (make libsdhip_gstamps.so): where a workgroup's cycles go -- K-loop, epilogue -- printed by one workgroup
    unsigned long long t_loop = 0, t_epi = 0, t_mark = __builtin_amdgcn_s_memtime(); int n_tiles_done = 0;
#endif
    while (true) {
#define W_PAIR(mask, n) do { _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(mask, 1, 0); } } while (0)
        // K-groups 0..2 of step s from stage `buf`; each group's fragments were read while the group before ran
        // MFMAs of a phase: the first carry the next phase's fragment reads (P = 2: four reads when the next phase keeps its activation fragments, eight otherwise)
        constexpr int NM = P == 0 ? 32 : P == 1 ? 8 : 16, NR_ODD = P == 2 ? 4 : 6, NR_EVEN = P == 2 ? 8 : 6;
        if constexpr (P == 0) {
            if (s == S - 1 && wid < 4) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(2 * G_STAGE + wid * 256));
                const unsigned vo = (unsigned)(n0c + wid * 64 + lane) * 4u;
                if (a.bias) lds_dma_b32(rPb, dst, vo);
                if (a.scale) { lds_dma_b32(rPs, dst + 1024, vo); lds_dma_b32(rPh, dst + 2048, vo); }
            }
        }
        hfrag(buf, 1, 1);
        hmma(0, 0);
        W_PAIR(0x100, NR_ODD); __builtin_amdgcn_sched_group_barrier(0x008, NM - NR_ODD, 0);
        __builtin_amdgcn_sched_barrier(0);
        hfrag(buf, 2, 0);
        hmma(1, 1);
        W_PAIR(0x100, NR_EVEN); __builtin_amdgcn_sched_group_barrier(0x008, NM - NR_EVEN, 0);
        __builtin_amdgcn_sched_barrier(0);
        hfrag(buf, 3, 1);
        hmma(0, 2);
        W_PAIR(0x100, NR_ODD); __builtin_amdgcn_sched_group_barrier(0x008, NM - NR_ODD, 0);
        __builtin_amdgcn_sched_barrier(0);
        // stage `buf` has been read out (this wave's last fragments of it are in registers once lgkmcnt reaches 0) and this wave's
        // share of step s + 1 has landed in the other stage (vmcnt(0)): behind the barrier that holds for every wave
        // (the counter retires in issue order.  Right behind an epilogue the wave's 16 output stores are YOUNGER than the eight DMAs
        // this barrier needs: vmcnt(16) lets them drain under the new tile's first K-step instead of in front of its first barrier)
        // (f32: 32 stores of 16 bytes per lane)
        if (s == 0 && q != q0 && P == 0) asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");
        else if (s == 0 && q != q0 && !a.y_f32) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P == 0) {
            // f32: a K-group is 32 MFMAs of 64 cycles; the eight DMA pieces go out one per four MFMAs instead of in a burst in front of them
            hfrag(buf ^ 1, 0, 0);
            const float a0[4] = {ha[1][0].x, ha[1][0].y, ha[1][0].z, ha[1][0].w}, a1[4] = {ha[1][1].x, ha[1][1].y, ha[1][1].z, ha[1][1].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float bv = e == 0 ? hb[1][j].x : e == 1 ? hb[1][j].y : e == 2 ? hb[1][j].z : hb[1][j].w;
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, a0[e], acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, a1[e], acc[1][j], 0, 0, 0);
                    if ((j & 1) == 1) { __builtin_amdgcn_sched_barrier(0); dma_piece(buf, e * 2 + (j >> 1)); __builtin_amdgcn_sched_barrier(0); }
                }
        } else {
            if (!(SD_G_ABLATE & 2)) dma(buf);            // step s + 2 into the stage just released
            hfrag(buf ^ 1, 0, 0);
            hmma(1, 3);
        }
        __builtin_amdgcn_sched_barrier(0);
        advance();

        if (s == S - 1) {
#ifdef SD_G_STAMPS
            { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_loop += t - t_mark; t_mark = t; }
#endif
            if constexpr (P == 0) {
            // ---- epilogue, f32.  Accumulator layout (weights first, see hmma): register r of acc[i][j], lane (li, lh) = row 32 i + li,
            // channel 32 j + 8 (r >> 2) + 4 lh + (r & 3): four registers = four consecutive channels of a row = one 16-byte store, 32
            // stores per wave.  [conv_gemm_h.hip stores every register as a row of 32 columns across a half-wave: 128 dword stores per
            // wave, more than the 6-bit vmcnt can skip, so the next tile's first LDS restaging waits for all of them; here the stores stay
            // outstanding under the next tile's first K-step (vmcnt(32) at its barrier).]  Parameters come from LDS (Ps, filled by DMA).
            // The descriptor starts at the tile's first row and ends at the batch's last one: rows >= M are dropped by the range check.
            const float slope = (a.act1 == 1) ? 0.0f : ((a.act1 == 2) ? 0.01f : 1.0f);
            const int rows_left = a.M - m0c;
            const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)(a.Y + (size_t)m0c * a.y_ld), 0,
                                                                               (unsigned)((size_t)(rows_left < GM ? rows_left : GM) * a.y_ld * 4), 0x00020000);
            const unsigned ybytes = (unsigned)a.y_ld * 4u;
            const bool hb_ = a.bias != nullptr, hs_ = a.scale != nullptr;
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int cl = wc * 128 + j * 32 + 8 * gq + 4 * lh;           // first of the lane's four channels, within the tile
                    const float4 cb = hb_ ? *(const float4*)(Ps + cl) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 cs = hs_ ? *(const float4*)(Ps + 256 + cl) : make_float4(1.f, 1.f, 1.f, 1.f);
                    const float4 ch = hs_ ? *(const float4*)(Ps + 512 + cl) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float cbv[4] = {cb.x, cb.y, cb.z, cb.w}, csv[4] = {cs.x, cs.y, cs.z, cs.w}, chv[4] = {ch.x, ch.y, ch.z, ch.w};
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[i][j][4 * gq + e] + cbv[e];
                            acc[i][j][4 * gq + e] = 0.0f;
                            v = fmaxf(v, v * slope);
                            o[e] = v * csv[e] + chv[e];
                        }
                        const unsigned vo = (unsigned)(wr * 64 + i * 32 + li) * ybytes + (unsigned)(n0c + cl) * 4u;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o), rY, vo, 0, 0);
                    }
                }
            }
            } else if constexpr (P == 2) {
            // ---- epilogue, 16 x 16 blocks.  C layout: register r of acc16[i][j], lane l = row 16 i + 4 (l >> 4) + r, column 16 j + (l & 15).
            // Same passage as the 32 x 32 form below: bias / activation / BatchNorm per value, a 4 x 4 transpose across the lane quad (four
            // consecutive columns of one row per lane), the wave's 4 KB strip sixteen rows (one i) at a time, whole rows back, 16-byte stores.
            const float slope = (a.act1 == 1) ? 0.0f : ((a.act1 == 2) ? 0.01f : 1.0f);
            const int lq = lane & 3;
            _Float16* const Y = (_Float16*)a.Y;
            char* const strip = lds + 2 * G_STAGE + wid * G_STRIP;               // [16 rows][256 bytes], 16-byte chunk c of row r at c ^ r
            float cb[8], cs[8], ch[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cc = n0c + wc * 128 + j * 16 + l15;
                cb[j] = a.bias ? a.bias[cc] : 0.0f;
                cs[j] = a.scale ? a.scale[cc] : 1.0f; ch[j] = a.scale ? a.shift[cc] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc16[i][j][e] + cb[j];
                        acc16[i][j][e] = 0.0f;
                        v = fmaxf(v, v * slope);
                        x[e] = v * cs[j] + ch[j];
                    }
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
                    // the lane now holds row 4 (l >> 4) + lq of the block, columns 16 j + (l15 & ~3) .. + 3
                    const int sr = 4 * l4 + lq;
                    if (a.y_f32) {
                        const int g = m0c + wr * 64 + i * 16 + sr;
                        const int co = n0c + wc * 128 + j * 16 + (l15 & ~3);
                        if (g < a.M) *(float4*)(a.Y + (size_t)g * a.y_ld + co) = make_float4(x[0], x[1], x[2], x[3]);
                    } else {
                        const half4 hv = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
                        *(half4*)(strip + sr * 256 + (((2 * j + (l15 >> 3)) ^ sr) * 16) + ((l15 >> 2) & 1) * 8) = hv;
                    }
                }
                if (!a.y_f32) {
                    const int c16 = lane & 15, rr = lane >> 4;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float4 v = *(const float4*)(strip + (4 * t + rr) * 256 + ((c16 ^ (4 * t + rr)) * 16));
                        const int g = m0c + wr * 64 + i * 16 + 4 * t + rr;
                        if (g < a.M) *(float4*)(Y + (size_t)g * a.y_ld + n0c + wc * 128 + c16 * 8) = v;
                    }
                }
            }
            } else {
            // ---- epilogue.  C layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
            // The wave's 64 x 128 outputs pass through a wave-private 4 KB LDS strip, sixteen rows at a time: written as the accumulators lie
            // after the 4 x 4 quad transpose (8 bytes per lane), read back as whole 256-byte rows and stored 16 bytes per lane -- 16 store
            // instructions per wave, each four complete rows (the register-staged kernel: 32 stores of eight 64-byte segments).
            // [Tried: the weight fragment as the MFMA's first operand, which hands every lane four consecutive channels of one row and
            // needs no cross-lane transposes -- but then a lane needs the bias / BatchNorm parameters of 64 channels instead of 4, and
            // fetching them inside the loop (3 dependent float4 loads per register group) cost more than the 3 VALU per value it saved.]
            const float slope = (a.act1 == 1) ? 0.0f : ((a.act1 == 2) ? 0.01f : 1.0f);
            const int lq = lane & 3;
            _Float16* const Y = (_Float16*)a.Y;
            char* const strip = lds + 2 * G_STAGE + wid * G_STRIP;               // [16 rows][256 bytes], 16-byte chunk c of row r at c ^ r
            float cb[4], cs[4], ch[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = n0c + wc * 128 + j * 32 + li;
                cb[j] = a.bias ? a.bias[cc] : 0.0f;
                cs[j] = a.scale ? a.scale[cc] : 1.0f; ch[j] = a.scale ? a.shift[cc] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {                                    // sixteen rows: gq = 2 hq, 2 hq + 1
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int gq = 2 * hq + g2;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float x[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = acc[i][j][4 * gq + e] + cb[j];
                                acc[i][j][4 * gq + e] = 0.0f;
                                v = fmaxf(v, v * slope);
                                x[e] = v * cs[j] + ch[j];
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
                            // the lane now holds row 8 g2 + 4 lh + lq of the strip, columns 32 j + (li & ~3) .. + 3
                            if (a.y_f32) {
                                const int g = m0c + wr * 64 + i * 32 + 8 * gq + 4 * lh + lq;
                                const int co = n0c + wc * 128 + j * 32 + (li & ~3);
                                if (g < a.M) *(float4*)(a.Y + (size_t)g * a.y_ld + co) = make_float4(x[0], x[1], x[2], x[3]);
                            } else {
                                const half4 hv = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
                                const int sr = 8 * g2 + 4 * lh + lq;
                                *(half4*)(strip + sr * 256 + (((j * 4 + (li >> 3)) ^ sr) * 16) + ((li >> 2) & 1) * 8) = hv;
                            }
                        }
                    }
                    if (!a.y_f32) {
                        // whole rows back: lane reads 16 bytes of row 4 t + (lane >> 4); the wave's LDS operations execute in order, so these
                        // reads see the writes above and the next strip's writes come behind them
                        const int c16 = lane & 15, rr = lane >> 4;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float4 v = *(const float4*)(strip + (4 * t + rr) * 256 + ((c16 ^ (4 * t + rr)) * 16));
                            const int g = m0c + wr * 64 + i * 32 + 16 * hq + 4 * t + rr;
                            if (g < a.M) *(float4*)(Y + (size_t)g * a.y_ld + n0c + wc * 128 + c16 * 8) = v;
                        }
                    }
                }
            }
            }
#ifdef SD_G_STAMPS
            { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_amdgcn_s_memtime(); t_epi += t - t_mark; t_mark = t; ++n_tiles_done; }
#endif
            q = next_sb(q);
            if (q >= sb_end) break;
            { int j_, nt_; (void)sb_valid(q, j_, nt_); m0c = __builtin_amdgcn_readfirstlane((xcd + 8 * j_) * GM); n0c = __builtin_amdgcn_readfirstlane(nt_ * GN); }
            s = 0;
        } else {
            ++s;
        }
        buf ^= 1;
    }
#ifdef SD_G_STAMPS
    if (blockIdx.x == 8 && tid == 0) printf("g256 M %d K %d N %d: %d tiles x %d steps; K-loop %llu cycles (%llu per step), epilogue %llu (%llu per tile = %.1f steps)\n", a.M, a.Cin * a.KT, a.Cout,
                                            n_tiles_done, S, t_loop, t_loop / (unsigned long long)(n_tiles_done * S), t_epi, t_epi / (unsigned long long)n_tiles_done,
                                            (double)t_epi / n_tiles_done / ((double)t_loop / (n_tiles_done * S)));
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the load stream runs past the last tile: nothing may be in flight when the LDS is given back
}

// returns 1 when the layer does not fit this kernel (the caller then uses the register-staged forms)
int launch_conv_gemm_g256(sd_ctx* c, const ConvArgs& in, const char* tag)
{
    ConvArgs a = in;
    const bool h = a.prec == 1;
    if ((a.prec != 1 && a.prec != 0) || !a.rowtab || (h && !a.W16) || (!h && !a.W) || a.X2 || a.item_bias || a.R || a.act2 || a.pad_mode != 0 || a.Cout < 256 || (a.Cout % GN) != 0 ||
        (a.y_ld & 3) || a.Cin % (h ? 64 : 32) != 0 || a.M < 8 * GM || (a.x_ld & (h ? 7 : 3)) || (!h && a.y_f32)) return 1;
    if ((int64_t)a.Cin * (a.kt_real > 0 ? a.kt_real : a.KT) < (c->conv_w256_kmin > 0 ? c->conv_w256_kmin : (h ? 256 : 128))) return 1;
    if (((size_t)a.X & 15) || ((size_t)(h ? a.W16 : (const void*)a.W) & 15)) return 1;
    const size_t lds_bytes = (size_t)2 * G_STAGE + 8 * G_STRIP;       // (f32: 3 KB of parameters where the fp16 form keeps its strips)
    const unsigned dev_bit = 1u << (c->device & 31);
    if (!(g_attr_g256.load(std::memory_order_acquire) & dev_bit)) {        // once per device (a second thread that gets here meanwhile sets the same values)
        if (hipFuncSetAttribute((const void*)k_conv_gemm_g256<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv_gemm_g256<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv_gemm_g256<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 1; }
        g_attr_g256.fetch_or(dev_bit, std::memory_order_release);
    }
    if (a.w_ld <= 0) a.w_ld = a.Cin;
    a.m_tiles = (a.M + GM - 1) / GM;
    a.n_tiles = (a.Cout + GN - 1) / GN;
    a.sched = c->conv_pn;
    a.stagger = c->conv_rot;
    int grid = (c->num_cu / 8) * 8;
    if (grid < 8) grid = 8;
    const int lx_max = ((a.m_tiles + 7) / 8) * a.n_tiles;
    if (grid / 8 > lx_max) grid = lx_max * 8;
    const int cin = a.cin_real > 0 ? a.cin_real : a.Cin;
    const double flops = 2.0 * (double)a.M * a.Cout * cin * (a.kt_real > 0 ? a.kt_real : a.KT);
    const double bytes = (h ? 2.0 : 4.0) * ((double)a.M * cin + (double)a.M * a.Cout + (double)a.Cout * cin * a.KT);
    {
        ProfScope ps(c, c->profile_detail ? std::string("conv_gemm:") + tag : std::string("conv_gemm"), flops, bytes);
        ProfScope ps16(c, h ? "conv_gemm_f16" : "conv_gemm_f32", flops, bytes);
        ProfScope psw(c, h ? "conv_w256_f16" : "conv_w256_f32", flops, bytes);         // this tile form alone (bench.py's roofline object)
        ProfScope pss(c, strcmp(tag, "lstm_ih") == 0 ? "conv_w256_seg" : "conv_w256_ecapa", flops, bytes);
        if (h && c->conv_mfma16 && ((int64_t)a.Cin * a.KT >= 1024 || c->conv_mfma16 == 2)) hipLaunchKernelGGL(k_conv_gemm_g256<2>, dim3(grid), dim3(512), lds_bytes, c->stream, a);
        else if (h) hipLaunchKernelGGL(k_conv_gemm_g256<1>         /* short contractions (block0, K = 640): 523 TF on this form against 486 */, dim3(grid), dim3(512), lds_bytes, c->stream, a);
        else hipLaunchKernelGGL(k_conv_gemm_g256<0>, dim3(grid), dim3(512), lds_bytes, c->stream, a);
    }
    if (hipGetLastError() != hipSuccess) SD_FAIL(c, SD_ERR_HIP, "k_conv_gemm_g256 launch failed (%s)", tag);
    return SD_OK;
}
