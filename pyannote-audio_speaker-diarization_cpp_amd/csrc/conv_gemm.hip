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
// registers (buffer loads: descriptor in SGPRs, K position as scalar offset): the
// next step's loads are issued during the first two MFMA groups of a step and
// written to the other LDS buffer during the third; one barrier per K step, placed
// before the fourth group, whose operands are already in registers.
//
// Persistent schedule: the grid is 2 workgroups per CU; the workgroups whose id is
// equal mod 8 (one XCD under round-robin dispatch, a speed assumption only) walk
// PM x PN super-blocks of tiles together (row panels m = xcd + 8j).  The K-step pipeline runs straight across tile
// boundaries: the first loads of tile i+1 are in flight while tile i's accumulators
// go through the epilogue, so short-K layers (Res2Net K=384, ASP K=128) do not pay
// a load-latency bubble per tile.
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned float4 (x_ld may be 10)

#define BM 128
#define BN 128
#define BK 32
#define LDP 36
#define SD_CONV_SCHED_DEFAULT 3

// DBG (micro-benchmark ablations only, never used by the pipeline): 1 = no epilogue stores,
// 2 = no global loads inside the K loop, 3 = no LDS restaging / barrier inside the K loop
// F16 (BASELINE.json configs[4], option "ecapa_precision" = 1): fp16 end to end.  X, X2, W16 and Y hold _Float16 (leading
// dimensions in elements), a K-step is 64 halves -- the same 128 bytes per row as 32 floats, so the load stream, the LDS tile
// (144-byte rows) and the K-step pipeline are byte for byte the f32 ones -- and the MFMA is v_mfma_f32_32x32x16_f16 (f32
// accumulation, f32 epilogue arithmetic, one rounding to fp16 at the store).  16x less MFMA time per K-step than f32.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// PR = 3 ("x3", option ecapa_precision = 3; the layers conv_gemm_h.hip's wide tile does not take: Res2Net, the attention's hidden layer): f32
// tensors, both operands split into hi + lo fp16 halves -- the activations (X + X2, added in f32) when they are staged, the weights by
// weights.cpp (W16x) -- and every 16-channel block runs hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16: 24 MFMAs of 32 cycles per wave and
// 32-channel K-step instead of 64 of 64 cycles.  LDS row and fragment offsets: conv_gemm_h.hip.
template <bool HAS_X2, int DBG, int PR>
__global__ __launch_bounds__(256, 2) void k_conv_gemm(ConvArgs a)
{
    constexpr bool F16 = PR == 1, X3 = PR == 3;
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDP];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDP];

    const int w = blockIdx.x, G = gridDim.x;                 // G is a multiple of 8
    const int xcd = w & 7, wl = w >> 3, wpx = G >> 3;
    // row panels owned by this XCD: m = xcd + 8 j
    const int mx = (a.m_tiles - xcd + 7) >> 3;
    // Super-block schedule: the wpx workgroups of an XCD work at the same time on a PM x PN block of tiles
    // (workgroup wl owns position (wl / PN, wl % PN) of every block).  They advance through K roughly in step,
    // so each A and W K-slice is pulled into the XCD's L2 once per block and shared by PN resp. PM workgroups
    // (measured before this order: 125 GB of fabric reads for the 3072x3072 layer against 4.9 GB algorithmic).
    const int pnmax = a.sched >= 100 ? a.sched - 100 : 8;          // (tuning: conv_pn128)
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
    auto next_sb = [&](int sb) -> int {         // next super-block in which this workgroup has a tile, or sb_end
        int j, nt;
        for (++sb; sb < sb_end; ++sb) if (sb_valid(sb, j, nt)) return sb;
        return sb_end;
    };
    const int q0 = next_sb(-1);
    if (q0 >= sb_end) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int c4 = tid & 7, r0 = tid >> 3;
    const int li = lane & 31, lh = lane >> 5;

    constexpr int ES = F16 ? 2 : 4;     // bytes per element of X / X2 / Y (and of W, except x3)
    constexpr int ESB = (F16 || X3) ? 2 : 4;
    // activation loader.  f32 / fp16: 16-byte chunk c4 of rows r0 + 32 p, p < 4.  x3: the 32-byte pair c8 of rows r8 + 64 p, p < 2
    constexpr int NPA = X3 ? 2 : 4;
    const int c8 = tid & 3, r8 = tid >> 2;
    auto a_row = [&](int p) { return X3 ? r8 + 64 * p : r0 + 32 * p; };
    constexpr int BKE = 128 / ES;       // elements per K-step: 128 bytes per row either way
    const int kcs = a.Cin / BKE;
    const int S = a.KT * kcs;
    const int ktr = a.kt_real > 0 ? a.kt_real : a.KT;      // taps that shift rows (split-weight mode: KT = 2 * ktr planes)
    const int half = ktr / 2;

    // ---- load stream state (runs one K-step ahead of the compute stream) ----
    // All global reads are buffer loads: a per-tile resource descriptor in SGPRs, a per-lane byte offset that only
    // changes with the tap, and the K position as the instruction's scalar offset -- moving to the next K-step is
    // one scalar add instead of twelve 64-bit vector pointer increments on the MFMA issue path.
    int rrel[NPA], tt[NPA], nd[NPA];    // per part: item offset (rows) relative to the tile's first item, clamped frame, last stored frame
    unsigned voA[NPA], voX[HAS_X2 ? NPA : 1], voB[4];
    const bool RT = a.rowtab != nullptr;                     // compact row space (see ConvArgs)
    const size_t in_rows = RT ? (size_t)(a.in_rows > 0 ? a.in_rows : a.M) : (size_t)((a.M + a.TpOut - 1) / a.TpOut) * a.TpIn;
    // row-table entries of the tile the load stream visits NEXT: fetched one tile ahead, so a tile switch never waits for them
    int2 pre[NPA]; int pre_base = 0;
    auto prefetch_tab = [&](int sb) {
        int j, nt;
        (void)sb_valid(sb, j, nt);
        const int m0 = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * BM);
        pre_base = a.rowtab[m0 < a.M ? m0 : a.M - 1].x;
#pragma unroll
        for (int p = 0; p < NPA; ++p) { int g = m0 + a_row(p); if (g > a.M - 1) g = a.M - 1; pre[p] = a.rowtab[g]; }
    };
    auto make_rsrc = [&](const void* base, size_t bytes) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t rA = make_rsrc(a.X, 0), rX = make_rsrc(a.X, 0);
    const __amdgpu_buffer_rsrc_t rB = make_rsrc(X3 ? a.W16x : F16 ? a.W16 : (const void*)a.W, (size_t)a.KT * a.Cout * a.w_ld * ESB);
#pragma unroll
    for (int p = 0; p < 4; ++p) voB[p] = (unsigned)((r0 + 32 * p) * a.w_ld * ESB + c4 * 16);
    int l_q = q0, l_kk = 0, l_kc = 0, m0l = 0, n0l = 0;
    // K always runs 0 .. Cin-1 in the same order for every tile: a row's result does not depend on where its tile sits
    // in the schedule (sharded and unsharded runs, full and dead-row-skipping runs stay bit-identical).  [Tried and
    // dropped: starting each workgroup of a super-block at a different K-chunk plus a per-tile rendezvous -- it lifts
    // the 3072x3072 layer's L2 hit rate from 27 % to 73 % (the L2 answers sharers that ask for a line at the same moment
    // with one miss each: fabric read requests == L2 misses) but not its speed, and it breaks that invariance.]
    unsigned sK = 0, sB = 0;            // scalar byte offsets: K position, start of this tap's / tile's W rows
    auto set_tile = [&](int sb) {
        int j, nt;
        (void)sb_valid(sb, j, nt);
        m0l = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * BM);   // wave-uniform: keeps the descriptors in SGPRs
        n0l = __builtin_amdgcn_readfirstlane(nt * BN);
        size_t row0;
        if (RT) {
            const int base = __builtin_amdgcn_readfirstlane(pre_base);
            row0 = (size_t)base;
#pragma unroll
            for (int p = 0; p < NPA; ++p) { rrel[p] = pre[p].x - base; tt[p] = ROWTAB_T(pre[p].y); nd[p] = ROWTAB_LAST(pre[p].y); }
        } else {
            const int b0 = m0l / a.TpOut;
            row0 = (size_t)b0 * a.TpIn;
#pragma unroll
            for (int p = 0; p < NPA; ++p) {
                int g = m0l + a_row(p);
                if (g > a.M - 1) g = a.M - 1;
                const int b = g / a.TpOut;
                int t = g - b * a.TpOut;
                if (t > a.T - 1) t = a.T - 1;
                rrel[p] = (b - b0) * a.TpIn;
                tt[p] = t; nd[p] = 0;
            }
        }
        rA = make_rsrc((const char*)a.X + row0 * a.x_ld * ES, (in_rows - row0) * a.x_ld * ES);
        if (HAS_X2) rX = make_rsrc((const char*)a.X2 + row0 * a.x2_ld * ES, (in_rows - row0) * a.x2_ld * ES);
    };
    auto set_tap = [&](int kk) {          // per-lane offsets of tap kk (reflect / valid row map)
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
            int qr;
            if (a.pad_mode == 0) {
                qr = tt[p] + ((kk >= ktr ? kk - ktr : kk) - half) * a.dil;
                if (qr < 0) qr = -qr;
                if (qr >= a.Tin) qr = 2 * (a.Tin - 1) - qr;
                if (qr < 0) qr = 0;
                if (RT && qr > nd[p]) qr = nd[p];
            } else {
                qr = tt[p] + kk * a.dil;
                if (qr > a.Tin - 1) qr = a.Tin - 1;
            }
            const unsigned row = (unsigned)(rrel[p] + qr);
            voA[p] = row * (unsigned)a.x_ld * ES + (X3 ? c8 * 32 : c4 * 16);
            if (HAS_X2) voX[p] = row * (unsigned)a.x2_ld * ES + (X3 ? c8 * 32 : c4 * 16);
        }
        sB = (unsigned)(((size_t)kk * a.Cout + n0l) * a.w_ld * ESB);
    };
    auto advance = [&]() {                // move the load stream to the next K-step
        if (++l_kc < kcs) { sK += 128; return; }
        l_kc = 0; sK = 0;
        if (++l_kk == a.KT) {
            l_kk = 0;
            const int nq = next_sb(l_q);
            if (nq < sb_end) {                                           // else: stay on the last tile (dummy loads)
                l_q = nq; set_tile(l_q);
                if (RT) { const int nq2 = next_sb(l_q); if (nq2 < sb_end) prefetch_tab(nq2); }
            }
        }
        set_tap(l_kk);
    };

    f4u ra[4], rb[4], rx[HAS_X2 ? 4 : 1];
    auto gload_part = [&](int p) {
        if constexpr (X3) {
            ra[p] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(rA, voA[p >> 1] + (p & 1) * 16, sK, 0));
            if (HAS_X2) rx[p] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(rX, voX[p >> 1] + (p & 1) * 16, sK, 0));
        } else {
            ra[p] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(rA, voA[p], sK, 0));
            if (HAS_X2) rx[p] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(rX, voX[p], sK, 0));
        }
        rb[p] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(rB, voB[p], sB + sK, 0));
    };
    auto lstore_x3 = [&](int buf, int p) {           // x3: row r8 + 64 p of A (X + X2 in f32, then the split), rows r0 + 32 (2 p), + 32 of W
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = ra[2 * p + (e >> 2)][e & 3];
            if (HAS_X2) f += rx[HAS_X2 ? 2 * p + (e >> 2) : 0][e & 3];
            hi[e] = (_Float16)f; lo[e] = (_Float16)(f - (float)hi[e]);
        }
        *(half8*)&As[buf][(r8 + 64 * p) * LDP + c8 * 8] = hi;
        *(half8*)&As[buf][(r8 + 64 * p) * LDP + c8 * 8 + 4] = lo;
#pragma unroll
        for (int h = 0; h < 2; ++h) { const int q = 2 * p + h; *(float4*)&Bs[buf][(r0 + 32 * q) * LDP + c4 * 4] = make_float4(rb[q][0], rb[q][1], rb[q][2], rb[q][3]); }
    };
    auto lstore = [&](int buf) {
        if constexpr (X3) { lstore_x3(buf, 0); lstore_x3(buf, 1); return; }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (HAS_X2) {                            // the add waits for the loads: keep it next to the LDS store
                if constexpr (F16) ra[p] = __builtin_bit_cast(f4u, __builtin_bit_cast(half8, ra[p]) + __builtin_bit_cast(half8, rx[p]));
                else ra[p] += rx[p];
            }
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

    const int aoff = (wr * 64 + li) * LDP + lh * 16;
    const int boff = (wc * 64 + li) * LDP + lh * 16;
    // Operand fragments are double-buffered in registers: the ds_reads of K-group qq+1 are issued before the
    // 16 MFMAs of group qq, so no MFMA waits on LDS latency (the scheduler is pinned with sched_barriers; left
    // alone it sinks the global loads to the end of the step, right in front of their ds_writes).
    float4 fa[2][2], fb[2][2];
    auto lfrag = [&](int buf, int qq, int fbuf) {
        const float* Ab = &As[buf][aoff];
        const float* Bb = &Bs[buf][boff];
        fa[fbuf][0] = *(const float4*)(Ab + qq * 4);
        fa[fbuf][1] = *(const float4*)(Ab + 32 * LDP + qq * 4);
        fb[fbuf][0] = *(const float4*)(Bb + qq * 4);
        fb[fbuf][1] = *(const float4*)(Bb + 32 * LDP + qq * 4);
    };
    auto mma16 = [&](int fbuf) {
        const float av0[4] = {fa[fbuf][0].x, fa[fbuf][0].y, fa[fbuf][0].z, fa[fbuf][0].w}, av1[4] = {fa[fbuf][1].x, fa[fbuf][1].y, fa[fbuf][1].z, fa[fbuf][1].w};
        const float bv0[4] = {fb[fbuf][0].x, fb[fbuf][0].y, fb[fbuf][0].z, fb[fbuf][0].w}, bv1[4] = {fb[fbuf][1].x, fb[fbuf][1].y, fb[fbuf][1].z, fb[fbuf][1].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv0[e], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv1[e], acc[1][1], 0, 0, 0);
        }
    };

    // fp16: one K-step = 4 k-blocks of 16 halves, 16 MFMAs; lane (li, lh) holds k = 16 kb + 8 lh .. +7 of row li.
    // LDS rows are the f32 tile's 144-byte rows (LDP floats): 64 halves + 8 halves of padding.
    half8 ha[2][2], hb[2][2];
    auto hfrag = [&](int buf, int kb, int fbuf) {
        const float* Ab = &As[buf][(wr * 64 + li) * LDP + lh * 4 + kb * 8];
        const float* Bb = &Bs[buf][(wc * 64 + li) * LDP + lh * 4 + kb * 8];
        ha[fbuf][0] = __builtin_bit_cast(half8, *(const float4*)Ab);
        ha[fbuf][1] = __builtin_bit_cast(half8, *(const float4*)(Ab + 32 * LDP));
        hb[fbuf][0] = __builtin_bit_cast(half8, *(const float4*)Bb);
        hb[fbuf][1] = __builtin_bit_cast(half8, *(const float4*)(Bb + 32 * LDP));
    };
    auto xoff = [&](int kb) { return ((kb & 1) * 2 + lh) * 8 + (kb >> 1) * 4; };     // x3 fragments: 0 / 1 = hi halves of channels 0..15 / 16..31, 2 / 3 = lo
    auto afrag = [&](int buf, int kb, int fbuf) {
        const float* Ab = &As[buf][(wr * 64 + li) * LDP + xoff(kb)];
        ha[fbuf][0] = __builtin_bit_cast(half8, *(const float4*)Ab);
        ha[fbuf][1] = __builtin_bit_cast(half8, *(const float4*)(Ab + 32 * LDP));
    };
    auto bfrag = [&](int buf, int kb, int fbuf) {
        const float* Bb = &Bs[buf][(wc * 64 + li) * LDP + xoff(kb)];
        hb[fbuf][0] = __builtin_bit_cast(half8, *(const float4*)Bb);
        hb[fbuf][1] = __builtin_bit_cast(half8, *(const float4*)(Bb + 32 * LDP));
    };
    auto xmma = [&](int fa_, int fb_) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[fa_][0], hb[fb_][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[fa_][0], hb[fb_][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[fa_][1], hb[fb_][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[fa_][1], hb[fb_][1], acc[1][1], 0, 0, 0);
    };
    auto hmma = [&](int fbuf) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[fbuf][0], hb[fbuf][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[fbuf][0], hb[fbuf][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[fbuf][1], hb[fbuf][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[fbuf][1], hb[fbuf][1], acc[1][1], 0, 0, 0);
    };

    // prologue: stage step 0 of the first tile
    if (RT) prefetch_tab(l_q);
    set_tile(l_q);
    if (RT) { const int nq2 = next_sb(l_q); if (nq2 < sb_end) prefetch_tab(nq2); }
    set_tap(0);
    int m0c = m0l, n0c = n0l;
#pragma unroll
    for (int p = 0; p < 4; ++p) gload_part(p);
    lstore(0);
    __syncthreads();
    advance();
    if constexpr (X3) { afrag(0, 0, 0); bfrag(0, 0, 0); } else if constexpr (F16) hfrag(0, 0, 0); else lfrag(0, 0, 0);
    if constexpr (F16) {
        // the two workgroups of a CU tend to run in lockstep (same tile length): both sit in their VALU-bound epilogues
        // together and both K loops fight for the MFMA pipe together.  Starting every second workgroup of an XCD half a
        // tile late lets one workgroup's epilogue overlap the other's MFMAs (a 16-MFMA K-step is only ~512 cycles here).
        if (a.sched == 3 && (wl & 1)) for (int i_ = 0; i_ < S * 4; ++i_) __builtin_amdgcn_s_sleep(1);
    }
    if constexpr (PR == 0) {
        // f32, short contractions (Res2Net: 12 K-steps of 4 096 cycles per tile, then an epilogue bound by its stores): the same lockstep.
        // a.stagger (tuning, option conv_stagger): 1 = every second workgroup of an XCD, 2 = the second half of an XCD's workgroups
        // start half a tile late
        if (a.stagger && S <= 16) {
            const bool late = a.stagger == 1 ? (wl & 1) : (wl >= (wpx >> 1));
            if (late) for (int i_ = 0; i_ < S / 4; ++i_) __builtin_amdgcn_s_sleep(127);
        }
    }

    int q = q0, s = 0, buf = 0;
    while (true) {
        const int cb = (DBG == 3) ? 0 : buf;
        if constexpr (X3) {
            // six groups of 4 MFMAs per K-step (conv_gemm_h.hip's order: hi0*hi0, lo0*hi0, hi0*lo0, hi1*hi1, lo1*hi1, hi1*lo1); every group's
            // operands are read one group ahead; loads in group 1, the split and the restaging in groups 4 and 5
            afrag(buf, 2, 1);
            gload_part(0); gload_part(1); gload_part(2); gload_part(3);
            xmma(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bfrag(buf, 2, 1);
            xmma(1, 0);
            __builtin_amdgcn_sched_barrier(0);
            afrag(buf, 1, 1); bfrag(buf, 1, 0);
            xmma(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            afrag(buf, 3, 0);
            xmma(1, 0);
            lstore_x3(buf ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            bfrag(buf, 3, 1);
            xmma(0, 0);
            lstore_x3(buf ^ 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            afrag(buf ^ 1, 0, 0); bfrag(buf ^ 1, 0, 0);
            xmma(1, 1);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (F16) {
            // same shape as the f32 step: fragments of k-block kb+1 are read while the MFMAs of kb run, the next step's
            // global loads are issued early, restaged to the other LDS buffer during k-block 2, one barrier per step
            hfrag(buf, 1, 1);
            gload_part(0); gload_part(1); gload_part(2); gload_part(3);
            hmma(0);
            __builtin_amdgcn_sched_barrier(0);
            hfrag(buf, 2, 0);
            hmma(1);
            __builtin_amdgcn_sched_barrier(0);
            hfrag(buf, 3, 1);
            hmma(0);
            lstore(buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            hfrag(buf ^ 1, 0, 0);
            hmma(1);
            __builtin_amdgcn_sched_barrier(0);
        } else {
        // one K step.  On entry fragment set 0 holds K-group 0 of this step.  Inside each scheduling region the
        // memory instructions are interleaved one by one with the MFMAs (sched_group_barrier: 0x008 MFMA, 0x020 VMEM
        // read, 0x100 DS read, 0x200 DS write): a bunch of 4-8 back-to-back VMEM/DS issues takes longer than the 64
        // cycles one MFMA keeps the pipe busy and leaves a bubble.
#define SGB_PAIR(mask, n) do { _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(mask, 1, 0); } } while (0)
        lfrag(cb, 1, 1);
        if (DBG < 2) { gload_part(0); gload_part(1); }
        mma16(0);
        SGB_PAIR(0x100, 4);
        if (HAS_X2) { SGB_PAIR(0x020, 6); __builtin_amdgcn_sched_group_barrier(0x008, 6, 0); }
        else {
#pragma unroll
            for (int i_ = 0; i_ < 4; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
        lfrag(cb, 2, 0);
        if (DBG < 2) { gload_part(2); gload_part(3); }
        mma16(1);
        SGB_PAIR(0x100, 4);
        if (HAS_X2) { SGB_PAIR(0x020, 6); __builtin_amdgcn_sched_group_barrier(0x008, 6, 0); }
        else {
#pragma unroll
            for (int i_ = 0; i_ < 4; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
        lfrag(cb, 3, 1);
        if (DBG < 3) lstore(buf ^ 1);        // restage: the next step's tile slices go to the other LDS buffer
        mma16(0);
        SGB_PAIR(0x100, 4);
        SGB_PAIR(0x200, 8);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        // after the barrier the next step's K-group 0 is fetched while this step's last 16 MFMAs (operands already
        // in registers) run
        if (DBG < 3) { __syncthreads(); lfrag(buf ^ 1, 0, 0); }
        else lfrag(0, 0, 0);
        mma16(1);
        SGB_PAIR(0x100, 4);
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        __builtin_amdgcn_sched_barrier(0);
        }
        const int m0n = m0l, n0n = n0l;           // origin of the tile the load stream is on
        advance();

        if (s == S - 1) {
            // ---- epilogue.  C layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
            // Per-column work (bias, activation, folded BN) happens while a lane still owns one column;
            // then each group of 4 registers (4 consecutive rows) is transposed inside its lane quad so
            // that a lane owns 4 consecutive columns of one row and stores one dwordx4 (16 wide stores
            // per lane and tile instead of 64 scalar ones).
            float cb_[2] = {0.0f, 0.0f}, cs_[2] = {1.0f, 1.0f}, ch_[2] = {0.0f, 0.0f};
            int cco[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                cco[j] = n0c + wc * 64 + j * 32 + li;
                const int cc = cco[j] < a.Cout ? cco[j] : a.Cout - 1;
                if (a.bias) cb_[j] = a.bias[cc];
                if (a.scale) { cs_[j] = a.scale[cc]; ch_[j] = a.shift[cc]; }
            }
            const float slope = (a.act1 == 1) ? 0.0f : ((a.act1 == 2) ? 0.01f : 1.0f);
            const float as = X3 ? a.acc_scale : 1.0f;        // x3: the weights were scaled by a power of two (weights.cpp)
            const int b0 = RT ? 0 : m0c / a.TpOut, t0 = RT ? 0 : m0c - b0 * a.TpOut;      // wave-uniform
            const bool fast_rows = a.TpOut >= BM;
            const bool wide = ((a.Cout | a.y_ld) & 3) == 0 && a.R == nullptr;
            const int lq = lane & 3;
            auto tile_out = [&](auto IBt, auto A2t) {
                constexpr bool IB = decltype(IBt)::value;
                constexpr int A2 = decltype(A2t)::value;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float x[2][4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int row = wr * 64 + i * 32 + e + 8 * gq + 4 * lh;
                            int b, t;
                            if (RT) { t = 0; b = 0; if (IB) { const int g = m0c + row; b = ROWTAB_ITEM(a.rowtab[g < a.M ? g : a.M - 1].y); } }
                            else if (fast_rows) { t = t0 + row; b = b0; if (t >= a.TpOut) { t -= a.TpOut; b += 1; } }
                            else { const int g = m0c + row; b = g / a.TpOut; t = g - b * a.TpOut; }
                            const bool live = RT || t < a.T;
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                float v = X3 ? acc[i][j][4 * gq + e] * as + cb_[j] : acc[i][j][4 * gq + e] + cb_[j];
                                acc[i][j][4 * gq + e] = 0.0f;
                                if (IB) v += a.item_bias[(size_t)b * a.ib_ld + (cco[j] < a.Cout ? cco[j] : a.Cout - 1)];
                                v = fmaxf(v, v * slope);                  // slope in [0, 1]: relu (0), leaky (0.01), identity (1)
                                v = v * cs_[j] + ch_[j];
                                if (A2 == 1) v = tanhf(v);
                                else if (A2 == 2) v = 1.0f / (1.0f + expf(-v));
                                x[j][e] = live ? v : 0.0f;
                            }
                        }
                        if (wide) {
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                // 4x4 transpose across the lane quad (two butterfly stages on DPP quad_perm)
                                float s0 = (lq & 1) ? x[j][0] : x[j][1];
                                float s1 = (lq & 1) ? x[j][2] : x[j][3];
                                float r0_ = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s0), 0xB1, 0xF, 0xF, true));
                                float r1_ = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s1), 0xB1, 0xF, 0xF, true));
                                if (lq & 1) { x[j][0] = r0_; x[j][2] = r1_; } else { x[j][1] = r0_; x[j][3] = r1_; }
                                s0 = (lq & 2) ? x[j][0] : x[j][2];
                                s1 = (lq & 2) ? x[j][1] : x[j][3];
                                r0_ = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s0), 0x4E, 0xF, 0xF, true));
                                r1_ = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s1), 0x4E, 0xF, 0xF, true));
                                if (lq & 2) { x[j][0] = r0_; x[j][1] = r1_; } else { x[j][2] = r0_; x[j][3] = r1_; }
                                const int g = m0c + wr * 64 + i * 32 + 8 * gq + 4 * lh + lq;
                                const int co = n0c + wc * 64 + j * 32 + (li & ~3);
                                if (g < a.M && co < a.Cout) {
                                    if (DBG == 1) { if (x[j][0] == 12345.678f) a.Y[0] = x[j][1]; }
                                    else if (F16 && !a.y_f32) {
                                        const half4 hv = {(_Float16)x[j][0], (_Float16)x[j][1], (_Float16)x[j][2], (_Float16)x[j][3]};
                                        *(half4*)((_Float16*)a.Y + (size_t)g * a.y_ld + co) = hv;
                                    } else *(float4*)(a.Y + (size_t)g * a.y_ld + co) = make_float4(x[j][0], x[j][1], x[j][2], x[j][3]);
                                }
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int g = m0c + wr * 64 + i * 32 + e + 8 * gq + 4 * lh;
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    if (g >= a.M || cco[j] >= a.Cout) continue;
                                    float v = x[j][e];
                                    if (a.R) v += a.R[(size_t)g * a.r_ld + cco[j]];
                                    if (F16 && !a.y_f32) ((_Float16*)a.Y)[(size_t)g * a.y_ld + cco[j]] = (_Float16)v;
                                    else a.Y[(size_t)g * a.y_ld + cco[j]] = v;
                                }
                            }
                        }
                    }
                }
            };
            using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
            // fast path (Res2Net, LSTM projections, linear layers): full column tiles, every row live, no per-item bias, no second
            // activation -- each accumulator register is one output row for 32 consecutive columns across a half-wave and is stored as it
            // lies (128 B per row and instruction, row term in the scalar offset, rows >= M dropped by the descriptor range): no lane
            // transposes, no per-row index arithmetic (conv_gemm_h.hip's epilogue).  Same values as the general path.
            if (DBG == 0 && wide && !a.item_bias && a.act2 == 0 && (a.Cout % BN) == 0 && (RT || a.T >= a.TpOut)) {
                const bool h16 = F16 && !a.y_f32;                    // fp16 mode: 2-byte elements (64 B per row and instruction)
                const unsigned es = h16 ? 2u : 4u;
                const int rows_left = a.M - m0c;
                const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)a.Y + (size_t)m0c * a.y_ld * es), 0,
                                                      (unsigned)((size_t)(rows_left < BM ? rows_left : BM) * a.y_ld * es), 0x00020000);
                const unsigned ybytes = (unsigned)a.y_ld * es;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const unsigned vo = (unsigned)(wr * 64 + i * 32 + 4 * lh) * ybytes + (unsigned)cco[j] * es;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = X3 ? acc[i][j][r] * as + cb_[j] : acc[i][j][r] + cb_[j];
                            acc[i][j][r] = 0.0f;
                            v = fmaxf(v, v * slope);
                            v = v * cs_[j] + ch_[j];
                            const unsigned so = (unsigned)((r & 3) + 8 * (r >> 2)) * ybytes;
                            if (h16) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), rY, vo, so, 0);
                            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rY, vo, so, 0);
                        }
                    }
                }
            } else
            if (a.item_bias) { if (a.act2 == 1) tile_out(std::true_type{}, T1{}); else if (a.act2 == 2) tile_out(std::true_type{}, T2{}); else tile_out(std::true_type{}, T0{}); }
            else { if (a.act2 == 1) tile_out(std::false_type{}, T1{}); else if (a.act2 == 2) tile_out(std::false_type{}, T2{}); else tile_out(std::false_type{}, T0{}); }
            q = next_sb(q);
            if (q >= sb_end) break;
            m0c = m0n; n0c = n0n; s = 0;
        } else {
            ++s;
        }
        buf ^= 1;
    }
}

// ---------------------------------------------------------------- k_skinny_gemm
// The per-utterance layers (SE bottleneck 1024->128->1024, ASP statistics bias 6144->128, fc 6144->192) have one row per
// item: a 128x128 tile grid keeps 6-48 workgroups busy for a K of up to 6144.  Here one workgroup owns a 32x32 output
// tile, its 4 waves split K and accumulate a 32x32 partial each with MFMA operands read straight from global memory;
// the partials are added in wave order through LDS (deterministic), wave 0 runs the epilogue.
__global__ __launch_bounds__(256) void k_skinny_gemm(ConvArgs a)
{
    __shared__ float part[3][16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    int row = m0 + li; if (row > a.M - 1) row = a.M - 1;
    int col = n0 + li; if (col > a.Cout - 1) col = a.Cout - 1;
    const float* pa = a.X + (size_t)row * a.x_ld + lh * 4;
    const float* pb = a.W + (size_t)col * a.w_ld + lh * 4;
    const int Kq = ((a.Cin / 4 + 7) / 8) * 8;
    const int k0 = w * Kq, k1 = (k0 + Kq < a.Cin) ? k0 + Kq : a.Cin;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    int k = k0;
    for (; k + 32 <= k1; k += 32) {                   // 4 x (8 k) per trip: 8 loads in flight per lane
        float4 av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { av[u] = *(const float4*)(pa + k + 8 * u); bv[u] = *(const float4*)(pb + k + 8 * u); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u].w, acc, 0, 0, 0);
        }
    }
    for (; k < k1; k += 8) {
        const float4 av = *(const float4*)(pa + k), bv = *(const float4*)(pb + k);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
    if (w > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[w - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (w != 0) return;
    const int co = n0 + li;
    const bool col_ok = co < a.Cout;
    const float cb = (a.bias && col_ok) ? a.bias[co] : 0.0f;
    const float cs = (a.scale && col_ok) ? a.scale[co] : 1.0f, ch = (a.scale && col_ok) ? a.shift[co] : 0.0f;
    const float slope = (a.act1 == 1) ? 0.0f : ((a.act1 == 2) ? 0.01f : 1.0f);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = ((acc[r] + part[0][r][lane]) + part[1][r][lane]) + part[2][r][lane];      // fixed order: waves 0,1,2,3
        v += cb;
        v = v > 0.0f ? v : v * slope;
        v = v * cs + ch;
        if (a.act2 == 1) v = tanhf(v);
        else if (a.act2 == 2) v = 1.0f / (1.0f + expf(-v));
        const int g = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;          // C layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        if (g < a.M && col_ok) a.Y[(size_t)g * a.y_ld + co] = v;
    }
}

static int conv_grid(sd_ctx* c, const ConvArgs& a)
{
    int g = 2 * c->num_cu;
    g = (g / 8) * 8;
    if (g < 8) g = 8;
    // no point launching more workgroups per XCD than the busiest XCD has tiles
    const int lx_max = ((a.m_tiles + 7) / 8) * a.n_tiles;
    if (g / 8 > lx_max) g = lx_max * 8;
    return g;
}

int launch_conv_gemm(sd_ctx* c, const ConvArgs& in, const char* tag)
{
    ConvArgs a = in;
    if (a.w_ld <= 0) a.w_ld = a.Cin;
    const bool f16 = a.prec == 1;
    if (f16 && !a.W16) SD_FAIL(c, SD_ERR_ARG, "conv_gemm(%s): fp16 mode without fp16 weights", tag);
    if (a.Cin % (f16 ? 64 : BK) != 0) SD_FAIL(c, SD_ERR_ARG, "conv_gemm(%s): Cin=%d not a multiple of %d", tag, a.Cin, f16 ? 64 : BK);
    if (a.M <= 0) return SD_OK;
    // LDS-DMA staged form (conv_gemm_g.hip): the default for fp16; for f32 it measured 5 % SLOWER than the register-staged, pinned kernel
    // (MFA 147 -> 140 TF, tdnn 140 -> 131: profiles/r04_g256_ablation.txt) and is only taken with option conv_glds_f32 = 1
    if (f16 && c->conv_h256 && c->conv_pp) { const int r = launch_conv_gemm_pp(c, a, tag); if (r != 1) return r; }
    if ((f16 && c->conv_h256 && c->conv_glds) || (!f16 && a.prec == 0 && c->conv_w256_f32 && c->conv_glds_f32)) { const int r = launch_conv_gemm_g256(c, a, tag); if (r != 1) return r; }
    if ((f16 && c->conv_h256) || (!f16 && c->conv_w256_f32) || a.prec == 3) { const int r = launch_conv_gemm_h256(c, a, tag); if (r != 1) return r; }
    const bool x3 = a.prec == 3 && a.W16x != nullptr;      // a layer the wide kernel does not take (X2, per-item bias, Cout = 128): the 128 x 128 form of the split
    if (a.prec == 3 && !x3) a.prec = 0;
    if (x3) a.w_ld = 2 * a.Cin;
    // (4 096 >= ROWTAB_MAX_ITEMS: the per-utterance layers of an ECAPA batch ALWAYS take this kernel.  Round 4 found the limit at 2 048 while a batch
    // could hold up to 4 095 short items: above it the 128 x 128 kernel took over, whose K order differs in the last bits, so a result depended on
    // how many items shared a batch -- tests/test_planted.py asserts batch-size independence bit for bit)
    if (a.prec == 0 && a.KT == 1 && a.M <= 4096 && a.TpIn == a.M && a.TpOut == a.M && a.T == a.M && a.Tin == a.M && !a.X2 && !a.item_bias && !a.R && !a.rowtab &&
        (a.x_ld & 3) == 0 && (a.w_ld & 3) == 0) {
        const int cinr = a.cin_real > 0 ? a.cin_real : a.Cin;
        ProfScope ps(c, c->profile_detail ? std::string("skinny_gemm:") + tag : std::string("skinny_gemm"), 2.0 * a.M * a.Cout * cinr,
                     4.0 * ((double)a.M * cinr + (double)a.M * a.Cout + (double)a.Cout * cinr));
        hipLaunchKernelGGL(k_skinny_gemm, dim3((a.M + 31) / 32, (a.Cout + 31) / 32), dim3(256), 0, c->stream, a);
        KCHECK(c);
        return SD_OK;
    }
    a.m_tiles = (a.M + BM - 1) / BM;
    a.n_tiles = (a.Cout + BN - 1) / BN;
    a.sched = c->conv_pn128 > 0 ? 100 + c->conv_pn128 : SD_CONV_SCHED_DEFAULT;
    a.stagger = c->conv_stagger;
    const int grid = conv_grid(c, a);
    // algorithmic work: valid rows only (T of every TpOut), un-padded input channels
    const double rows = a.rowtab ? (double)a.M
                                 : (double)(a.M / a.TpOut) * a.T + (double)((a.M % a.TpOut) < a.T ? (a.M % a.TpOut) : a.T);
    const int cin = a.cin_real > 0 ? a.cin_real : a.Cin;
    const int kt_alg = a.kt_real > 0 ? a.kt_real : a.KT;          // algorithmic taps (the lo planes of the split-weight mode are overhead, not work)
    const double flops = 2.0 * rows * a.Cout * cin * kt_alg;
    const double bytes = (f16 ? 2.0 : 4.0) * (rows * cin * (a.X2 ? 2 : 1) + rows * a.Cout + (double)a.Cout * cin * a.KT);
    ProfScope ps(c, c->profile_detail ? std::string("conv_gemm:") + tag : std::string("conv_gemm"), flops, bytes);
    ProfScope ps16(c, f16 ? "conv_gemm_f16" : x3 ? "conv_gemm_x3" : "conv_gemm_f32", flops, bytes);        // per precision (bench: roofline of the fp16 instantiations alone)
    if (f16) {
        if (a.X2) hipLaunchKernelGGL((k_conv_gemm<true, 0, 1>), dim3(grid), dim3(256), 0, c->stream, a);
        else hipLaunchKernelGGL((k_conv_gemm<false, 0, 1>), dim3(grid), dim3(256), 0, c->stream, a);
    } else if (x3) {
        if (a.X2) hipLaunchKernelGGL((k_conv_gemm<true, 0, 3>), dim3(grid), dim3(256), 0, c->stream, a);
        else hipLaunchKernelGGL((k_conv_gemm<false, 0, 3>), dim3(grid), dim3(256), 0, c->stream, a);
    } else if (a.X2) hipLaunchKernelGGL((k_conv_gemm<true, 0, 0>), dim3(grid), dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL((k_conv_gemm<false, 0, 0>), dim3(grid), dim3(256), 0, c->stream, a);
    KCHECK(c);
    return SD_OK;
}

// ---- measurement hook: time one conv_gemm shape on random-filled scratch buffers (tools/tune_conv.py)
__global__ void k_fill_rand(float* p, int64_t n, unsigned seed)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((float)(x & 0xffffff) / 8388608.0f - 1.0f) * 0.5f;
}

extern "C" int sd_bench_conv(sd_ctx* c, int64_t items, int Tp, int T, int Cin, int Cout, int KT, int dil, int has_x2, int dbg, int reps, double* ms_per_launch)
{
    if (!c || !ms_per_launch || items <= 0 || Cin % BK) return SD_ERR_ARG;
    c->err.clear();
    if (hipSetDevice(c->device) != hipSuccess) return SD_ERR_HIP;
    const int64_t M = items * Tp;
    const int pad = (dbg / 100) * 32;         // dbg = 100*pad_units + 10*(sched+1) + ablation: leading dimensions padded by pad floats
    dbg %= 100;
    const int xld = Cin + pad, wld = Cin + pad, yld = Cout + pad;
    WS(c, float, X, "bc_X", M * xld);
    WS(c, float, X2, "bc_X2", has_x2 ? M * xld : 16);
    WS(c, float, W, "bc_W", (int64_t)KT * Cout * wld);
    WS(c, float, Y, "bc_Y", M * yld);
    WS(c, float, B, "bc_B", 3 * Cout);
    hipLaunchKernelGGL(k_fill_rand, dim3((unsigned)((M * xld + 255) / 256)), dim3(256), 0, c->stream, X, M * xld, 1u);
    if (has_x2) hipLaunchKernelGGL(k_fill_rand, dim3((unsigned)((M * xld + 255) / 256)), dim3(256), 0, c->stream, X2, M * xld, 2u);
    hipLaunchKernelGGL(k_fill_rand, dim3((unsigned)(((int64_t)KT * Cout * wld + 255) / 256)), dim3(256), 0, c->stream, W, (int64_t)KT * Cout * wld, 3u);
    hipLaunchKernelGGL(k_fill_rand, dim3((unsigned)((3 * Cout + 255) / 256)), dim3(256), 0, c->stream, B, (int64_t)3 * Cout, 4u);
    ConvArgs a; memset(&a, 0, sizeof(a));
    a.X = X; a.x_ld = xld; a.X2 = has_x2 ? X2 : nullptr; a.x2_ld = xld; a.W = W; a.w_ld = wld; a.Y = Y; a.y_ld = yld;
    a.bias = B; a.scale = B + Cout; a.shift = B + 2 * Cout; a.act1 = 1;
    a.M = (int)M; a.TpIn = a.TpOut = Tp; a.Tin = a.T = T; a.Cin = Cin; a.Cout = Cout; a.KT = KT; a.dil = dil; a.pad_mode = 0;
    a.m_tiles = (a.M + BM - 1) / BM; a.n_tiles = (a.Cout + BN - 1) / BN;
    a.sched = (dbg >= 10) ? (dbg / 10 - 1) : SD_CONV_SCHED_DEFAULT;   // dbg = 10*(sched+1) + ablation
    dbg %= 10;
    const int grid = conv_grid(c, a);
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
#define LAUNCH_V(X2, D) hipLaunchKernelGGL((k_conv_gemm<X2, D, 0>), dim3(grid), dim3(256), 0, c->stream, a)
#ifdef SD_CONV_ABLATIONS      // make EXTRA=-DSD_CONV_ABLATIONS: 6 more instantiations of the kernel (minutes of compile time), tuning only
#define LAUNCH_S(X2) do { if (dbg == 0) LAUNCH_V(X2, 0); else if (dbg == 1) LAUNCH_V(X2, 1); else if (dbg == 2) LAUNCH_V(X2, 2); else LAUNCH_V(X2, 3); } while (0)
#else
    if (dbg != 0) SD_FAIL(c, SD_ERR_ARG, "sd_bench_conv: ablation %d needs a build with -DSD_CONV_ABLATIONS", dbg);
#define LAUNCH_S(X2) LAUNCH_V(X2, 0)
#endif
    for (int r = -2; r < reps; ++r) {
        if (r == 0) HIPCHK(c, hipEventRecord(e0, c->stream));
        if (has_x2) LAUNCH_S(true); else LAUNCH_S(false);
    }
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return SD_OK;
}
