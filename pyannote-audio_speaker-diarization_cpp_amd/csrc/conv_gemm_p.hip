// conv_gemm_p.hip -- the fp16 mode's wide layers (Cout >= 256, K >= 1 024) with a load stream that is never drained (round 6).
//
// conv_gemm_g.hip (round 4) stages a whole K-step per barrier and empties its LDS-DMA queue in front of every barrier: it runs at the rate at which a
// 64 KB burst lands.  Round 6 measured the pieces one by one in a stand-alone loop (tools/g256_lab.hip, profiles/r06_g256_lab.txt): a CU takes in
// 24 - 27 bytes per clock whichever way the bytes are requested (LDS-DMA or register loads, 4 or 8 waves issuing, 64 / 128 / 256 workgroups on the
// chip), i.e. 2 400 - 2 700 cycles for the 64 KB a 256 x 256 x 64 K-tile needs, against 2 048 for its MFMAs: the tile is bound by the CU's ingest,
// and what a kernel loses on top is the time in which nothing is being requested.  Two loop structures were built on that:
//   * "ping-pong" (k_pp2 in the lab; this file's first form, commit 3c2c29b): two wave groups half a phase apart, one multiplies while the other reads
//     fragments and issues DMAs, half-tiles released phase by phase: 3 184 cycles per K-tile = 2 624 (its DMAs' issue time) + 8 barriers of ~ 70;
//   * what is here (k_w8 in the lab): NO roles -- every wave interleaves its own fragment reads and LDS-DMAs with its MFMAs, the two waves of a SIMD
//     cover each other's DMA-issue stalls, two barriers per K-tile: 3 010 cycles per K-tile on MFA (1 220 TF in the lab against 1 142), 3 450 against
//     3 820 on tdnn.
//
//  * 256 x 256 tile, K-tile of 64 halves, 8 waves = 2 (row groups) x 4 (column quarters); a wave owns 128 x 64 outputs = 8 row blocks x 4 column blocks
//    of 16 x 16 (v_mfma_f32_16x16x32_f16, 128 accumulator registers), multiplied k-group by k-group (32 halves each): sub-group s = row blocks 2 s, 2 s + 1.
//  * LDS: 2 buffers x {A half 0, A half 1, B half 0, B half 1} x 128 rows x 128 B (unpadded, 16-byte chunk c of row R at c ^ ((R >> 1) & 7): conflict-free
//    ds_read_b128, as in conv_gemm_g.hip).  Buffer b = T & 1 holds K-tile T.  Per K-tile and wave: 64 MFMAs, 24 fragment reads, 8 LDS-DMAs (1 KB each):
//      phase X: 32 MFMAs on k-group 0 | the 12 fragment reads of k-group 1 (buffer b) | the 4 B pieces of K-tile T + 1 -> buffer b ^ 1, one behind every sub-group
//      lgkmcnt(0), barrier B1: every wave has read buffer b out
//      phase Y: 32 MFMAs on k-group 1 | the 4 A pieces of K-tile T + 2 -> buffer b, one behind every sub-group | in front of sub-group 3: vmcnt(3) (all of
//               K-tile T + 1 has landed: its A pieces went out a K-tile ago, its B pieces in phase X), barrier B2, the 12 reads of k-group 0 of T + 1
//    Eight DMAs per wave are in flight at any time, no wait ever empties the queue, the activation rows (HBM) have more than a K-tile to land, the weights
//    (L2) half of one.
//  * Output: the weight fragment is the MFMA's first operand, so a lane holds four consecutive channels of one row per 16 x 16 block; the loader
//    permutes which weight row goes to which LDS row (LDS row 16 j + rho of a wave's 32 channels holds channel 8 (rho >> 2) + 4 j + (rho & 3)), so
//    column blocks 2 nh, 2 nh + 1 together give a lane EIGHT consecutive channels = one 16-byte store, no cross-lane transposes, no LDS strip.
//  * Epilogue in 16 chunks (row block x column pair = 8 values per lane: bias, activation, BatchNorm, one store; a chunk zeroes what it has read), run
//    in front of the NEXT tile's first MFMAs (in a basic block of their own: in one block with MFMAs hipcc spills 217 registers); the load stream never
//    stops for a tile boundary.  Parameters of the wave's 64 channels come through a wave-private LDS area (three small LDS-DMAs per tile, two areas by
//    tile parity), the row table entries of the rows a wave stages through another (one LDS-DMA per tile, one tile ahead): no compiler-visible
//    vector-memory load sits in the loop, so hipcc's own vmcnt waits never drain the DMA queue.
//
// Same products in the same order as k_conv_gemm_g256<2> (K-groups of 32 ascending, one 16x16x32 MFMA per group and block), operands swapped: the two
// kernels are compared bit for bit (tests/test_gpu_parity.py::test_ecapa_fp16_ping_pong_kernel_gives_the_same_bits).
// Takes fp16 tensors only (prec 1), row-table layers without second input / per-item bias / residual / second activation; everything else stays
// with conv_gemm_g.hip / conv_gemm_h.hip.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;

#define P_HALF 16384           // a half-tile: 128 rows x 128 B
#define P_BUF 65536            // A half 0, A half 1, B half 0, B half 1
#define P_PAR (2 * P_BUF)      // parameters: 8 waves x 2 areas x 1 KB ([bias | scale | shift][16 groups of 4 channels])
#define P_TAB (P_PAR + 16384)  // row table entries: 8 waves x 2 slots x 256 B ([half][16 rows] int2)
#define P_LDS (P_TAB + 4096)

// One LDS-DMA wave instruction (inline assembly on purpose: see conv_gemm_g.hip -- hipcc would put vmcnt(0) in front of the next ds_read).
__device__ __forceinline__ void pp_dma_b128(v4i rs, unsigned ldsaddr, unsigned vo)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(ldsaddr), "v"(vo), "s"(rs) : "memory");
}
__device__ __forceinline__ v4i pp_rsrc(const void* base, size_t bytes)
{
    const unsigned long long b = (unsigned long long)base;
    v4i r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes));
    r[3] = 0x00020000;
    return r;
}

#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)
// Behind every 16-byte store of the epilogue: with an SGPR soffset (this kernel's stores have one for nh = 1) hipcc leaves NO wait state between the store
// and a VALU write of its first data register -- its hazard rule is gfx9's, where the store-data hazard needs an immediate soffset.  On gfx950 such a
// store now and then sent the overwritten value for lanes 12-15 of each row of 16: 4 x 4 patches of wrong outputs that moved from run to run, seen in an
// x3 form of this kernel that was measured and not kept (profiles/r06_px_attempt.txt).  Two wait states, fenced so that nothing is scheduled in between.
#define PP_STORE_PAD() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 1"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_BARRIER() do { PP_FENCE(); __builtin_amdgcn_s_barrier(); PP_FENCE(); } while (0)

// RELU: act1 == 1 known at compile time (every layer this kernel takes in ECAPA): the slope is an inline constant
template <bool RELU>
__global__ __launch_bounds__(512) void k_conv_gemm_pp(ConvArgs a)
{
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

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = wid >> 2, wc = wid & 3;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int kcs = a.Cin / 64;                      // K-tiles per tap
    const int S = a.KT * kcs;                        // K-tiles per output tile (>= 2)
    const int ktr = a.kt_real > 0 ? a.kt_real : a.KT;
    const int half = ktr / 2;
    const size_t in_rows = (size_t)(a.in_rows > 0 ? a.in_rows : a.M);
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)lds;
    const _Float16* const X = (const _Float16*)a.X;
    const _Float16* const W16 = (const _Float16*)a.W16;

    // ---- loader.  A half-tile = 16 wave instructions of 1 KB (8 rows x 128 B); wave `wid` fills LDS rows 16 wid + 8 p + (lane >> 3), p = 0, 1, of every
    // half-tile.  Lane l writes bytes [16 l, 16 l + 16) of its piece = row l >> 3, chunk POSITION l & 7, which holds the logical chunk (l & 7) ^ ((row >> 1) & 7)
    // (conflict-free ds_read_b128 of the fragments, as in conv_gemm_g.hip).
    const int prow = lane >> 3;
    unsigned chk[2], voB[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = wid * 16 + p * 8 + prow;
        chk[p] = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
        const int rho = r & 15;
        const int col = 32 * (wid >> 1) + 8 * (rho >> 2) + 4 * (wid & 1) + (rho & 3);      // the channel (within the half) whose weights that LDS row holds
        voB[p] = (unsigned)col * (unsigned)a.w_ld * 2u + chk[p];
    }
    const unsigned dstw = __builtin_amdgcn_readfirstlane((unsigned)(wid * 2048));
    const unsigned tab0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(P_TAB + wid * 512));
    const unsigned par0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(P_PAR + wid * 2048));
    const v4i rTab = pp_rsrc(a.rowtab, (size_t)a.M * 8);      // rows >= M read as {0, 0}: any valid address will do, their outputs are never stored

    // Two cursors walk the workgroup's stream of K-tiles (tile, tap kk, channel chunk kc): cB = the K-tile after the one being multiplied (its B pieces
    // go out in phase X), cA = the one after that (its A pieces go out in phase Y; it keeps the row offsets of the A rows this wave stages).
    struct Cur {
        int sb, kk, kc, m0, n0, base, tiles;         // base: first input row of the tile's first item (the buffer descriptor starts there)
        unsigned vo[4];                              // cA, per piece q = 2 h + p: byte offset of the row the lane stages, from the descriptor's base (cur_tap)
    };
    auto tile_of = [&](int sb, int& m0, int& n0) { int j, nt; (void)sb_valid(sb, j, nt); m0 = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * 256); n0 = __builtin_amdgcn_readfirstlane(nt * 256); };
    auto table_dma = [&](int m0, int slot) {         // this wave's 32 rows of tile m0: lanes 0-7 rows 16 wid .. + 15 of half 0 (two entries each), lanes 8-15 of half 1
        const unsigned dst = __builtin_amdgcn_readfirstlane(tab0 + (unsigned)(slot * 256));
        if (lane < 16) pp_dma_b128(rTab, dst, (unsigned)(m0 + 128 * (lane >> 3) + 16 * wid + 2 * (lane & 7)) * 8u);
    };
    auto cur_rows = [&](Cur& c) {                    // entries of the tile the cursor has just entered (its table slot was filled a tile ago)
        const int m0c = c.m0 < a.M ? c.m0 : a.M - 1;
        {   // rowtab[m0c].x by a scalar load written out by hand: hipcc takes the address for divergent and emits a global_load + vmcnt(0), which would drain the DMA queue
            const unsigned long long ad = (unsigned long long)(a.rowtab + m0c);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
            const unsigned long long ads = ((unsigned long long)hi << 32) | lo;
            int bs;
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bs) : "s"(ads) : "memory");
            c.base = bs;
        }
    };
    auto cur_tap = [&](Cur& c) {                     // the row table entries stay in the wave's LDS slot for the whole tile: re-read at every tap, not kept in registers
        const int2* const tb = (const int2*)(lds + P_TAB + wid * 512 + (c.tiles & 1) * 256);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int2 e = tb[(q >> 1) * 16 + (q & 1) * 8 + prow];
            int qr = ROWTAB_T(e.y) + ((c.kk >= ktr ? c.kk - ktr : c.kk) - half) * a.dil;
            const int nd = ROWTAB_LAST(e.y);
            if (qr < 0) qr = -qr;
            if (qr >= a.Tin) qr = 2 * (a.Tin - 1) - qr;
            if (qr < 0) qr = 0;
            if (qr > nd) qr = nd;
            c.vo[q] = (unsigned)(e.x - c.base + qr) * (unsigned)a.x_ld * 2u + chk[q & 1];
        }
    };
    // next K-tile; IS_A: the cursor that carries the activation rows.  Returns true when it has entered a new tile
    auto cur_adv = [&](Cur& c, bool is_a) -> bool {
        if (++c.kc < kcs) return false;
        c.kc = 0;
        if (++c.kk < a.KT) { if (is_a) cur_tap(c); return false; }
        c.kk = 0;
        const int nq = next_sb(c.sb);
        if (nq >= sb_end) { if (is_a) cur_tap(c); return false; }          // past the last tile: the stream re-reads the last tile (never multiplied)
        c.sb = nq; ++c.tiles;
        tile_of(nq, c.m0, c.n0);
        if (is_a) { cur_rows(c); cur_tap(c); }
        return true;
    };
    // piece q = 0 .. 3 of the wave's share of a K-tile's A (B): half q >> 1, piece q & 1
    auto dmaA = [&](const Cur& c, int buf, int q) {
        const v4i rs = pp_rsrc(X + ((size_t)c.base * a.x_ld + (size_t)c.kc * 64), (in_rows - (size_t)c.base) * a.x_ld * 2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * P_BUF + (q >> 1) * P_HALF + (q & 1) * 1024) + dstw);
        pp_dma_b128(rs, dst, c.vo[q]);
    };
    auto dmaB = [&](const Cur& c, int buf, int q) {
        const v4i rs = pp_rsrc(W16 + (((size_t)c.kk * a.Cout + c.n0 + 128 * (q >> 1)) * a.w_ld + (size_t)c.kc * 64), 0xffffffffull);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * P_BUF + (2 + (q >> 1)) * P_HALF + (q & 1) * 1024) + dstw);
        pp_dma_b128(rs, dst, voB[q & 1]);
    };
    // parameters of the wave's 64 channels: lane k < 16 of DMA `arr` fetches channels 128 (k >> 3) + 32 wc + 4 (k & 7) .. + 3 -> floats [arr][k][4] of the area
    const unsigned voP = (unsigned)(128 * ((lane & 15) >> 3) + 32 * wc + 4 * (lane & 7)) * 4u;
    auto stageP = [&](int n0, int par) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(par0 + (unsigned)(par * 1024));
        const v4i rb = pp_rsrc(a.bias + n0, (size_t)(a.Cout - n0) * 4), rc = pp_rsrc(a.scale + n0, (size_t)(a.Cout - n0) * 4), rh = pp_rsrc(a.shift + n0, (size_t)(a.Cout - n0) * 4);
        if (lane < 16) {
            pp_dma_b128(rb, dst, voP);
            pp_dma_b128(rc, dst + 256, voP);
            pp_dma_b128(rh, dst + 512, voP);
        }
    };

    // ---- fragments.  Lane l reads k = 32 ks + 8 (l >> 4) .. + 7 (logical chunk 4 ks + (l >> 4)) of row l & 15 of a 16-row block.
    const int sw = (l15 >> 1) & 7;
    const int c0 = (l4 ^ sw) * 16, c1 = ((4 + l4) ^ sw) * 16;
    const char* const Afr = lds + (64 * g + l15) * 128;                  // + mh * P_HALF + i * 2048 + c{ks}
    const char* const Bfr = lds + 2 * P_HALF + (32 * wc + l15) * 128;    // + nh * P_HALF + j * 2048 + c{ks}
    float4 fa[2][8], fb[2][4];          // [k-group][row block R = 4 mh + i], [k-group][column block C = 2 nh + j]
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto rdA = [&](int buf, int ks, int R) { fa[ks][R] = *(const float4*)(Afr + buf * P_BUF + (R >> 2) * P_HALF + (R & 3) * 2048 + (ks ? c1 : c0)); };
    auto rdB = [&](int buf, int ks, int C) { fb[ks][C] = *(const float4*)(Bfr + buf * P_BUF + (C >> 1) * P_HALF + (C & 1) * 2048 + (ks ? c1 : c0)); };
    auto rd_set = [&](int buf, int ks, int part) {         // the 12 reads of a k-group in four parts of three
        if (part == 0) { rdB(buf, ks, 0); rdB(buf, ks, 1); rdB(buf, ks, 2); }
        else if (part == 1) { rdB(buf, ks, 3); rdA(buf, ks, 0); rdA(buf, ks, 1); }
        else if (part == 2) { rdA(buf, ks, 2); rdA(buf, ks, 3); rdA(buf, ks, 4); }
        else { rdA(buf, ks, 5); rdA(buf, ks, 6); rdA(buf, ks, 7); }
    };
    auto mma_sg = [&](int ks, int sg) {                    // sub-group sg: row blocks 2 sg, 2 sg + 1
#pragma unroll
        for (int C = 0; C < 4; ++C)
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
                acc[2 * sg + r2][C] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fb[ks][C]), __builtin_bit_cast(half8, fa[ks][2 * sg + r2]), acc[2 * sg + r2][C], 0, 0, 0);
    };
    // one epilogue chunk: row block R (rows 128 (R >> 2) + 64 g + 16 (R & 3) + l15), channels 128 nh + 32 wc + 8 l4 .. + 7 of the tile behind `rY` (rows >= M fall outside the descriptor)
    const float slope = (a.act1 == 1) ? 0.0f : ((a.act1 == 2) ? 0.01f : 1.0f);
    const unsigned ybytes = (unsigned)a.y_ld * 2u;
    const unsigned voY = (unsigned)(64 * g + l15) * ybytes + (unsigned)(32 * wc + 8 * l4) * 2u;
    auto chunk = [&](int R, int nh, int par, __amdgpu_buffer_rsrc_t rY) {
        const float* const pw = (const float*)(lds + P_PAR + wid * 2048 + par * 1024) + (nh * 8 + 2 * l4) * 4;
        half8 hv;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {          // four channels at a time; the fences keep one half's twelve parameters live, not the next chunks' as well
            asm volatile("" ::: "memory");
            const float4 b = *(const float4*)(pw + 4 * hf), sc = *(const float4*)(pw + 64 + 4 * hf), sh = *(const float4*)(pw + 128 + 4 * hf);
            const float bb[4] = {b.x, b.y, b.z, b.w}, ss[4] = {sc.x, sc.y, sc.z, sc.w}, hh[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[R][2 * nh + hf][e] + bb[e];
                acc[R][2 * nh + hf][e] = 0.0f;      // (a first MFMA on C = 0 instead would save these 128 moves per tile -- hipcc then spills 363 registers)
                v = fmaxf(v, v * (RELU ? 0.0f : slope));      // (not max(v, 0): a NaN must stay a NaN -- the x3 mode's overflow check and every other kernel pass it on; v * 0 keeps it)
                hv[4 * hf + e] = (_Float16)(v * ss[e] + hh[e]);
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, hv), rY, voY + (unsigned)(128 * (R >> 2) + 16 * (R & 3)) * ybytes, 256u * nh, 0);
        PP_STORE_PAD();
    };
    _Float16* const Y = (_Float16*)a.Y;
    auto make_rY = [&](int m0, int n0) {
        const int rows_left = a.M - m0;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(Y + (size_t)m0 * a.y_ld + n0), 0, (unsigned)((size_t)(rows_left < 256 ? rows_left : 256) * a.y_ld * 2), 0x00020000);
    };


    // ---- prologue: the first tile's row table entries (and the second tile's, one tile ahead), K-tile 0 whole -> buffer 0, the A pieces of K-tile 1 -> buffer 1
    Cur cA; cA.sb = q0; cA.kk = 0; cA.kc = 0; cA.tiles = 0;
    tile_of(q0, cA.m0, cA.n0);
    table_dma(cA.m0, 0);
    { const int nq = next_sb(q0); if (nq < sb_end) { int m1, n1; tile_of(nq, m1, n1); table_dma(m1, 1); } }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    cur_rows(cA); cur_tap(cA);
    Cur cB = cA;
    const int m0_first = cA.m0, n0_first = cA.n0;
    // (the cursor that enters a new tile asks for the row table entries of the tile AFTER it)
    auto lead_adv = [&]() {
        if (cur_adv(cA, true)) { const int nq = next_sb(cA.sb); if (nq < sb_end) { int m1, n1; tile_of(nq, m1, n1); table_dma(m1, (cA.tiles + 1) & 1); } }
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) { dmaA(cA, 0, q); dmaB(cB, 0, q); }
    lead_adv();
    (void)cur_adv(cB, false);
#pragma unroll
    for (int q = 0; q < 4; ++q) dmaA(cA, 1, q);
    lead_adv();
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // K-tile 0 (and whatever row table went out behind it: older than the four pieces left in flight)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int part = 0; part < 4; ++part) rd_set(0, 0, part);

    int q = q0, t = 0, buf = 0, par = 0;
    bool have_prev = false;
    __amdgpu_buffer_rsrc_t rYc = make_rY(m0_first, n0_first), rYp = rYc;
    int n0c = n0_first;
    while (true) {
        // ---- phase X
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PP_FENCE();
        if (t == 0 && have_prev) {
            // the first K-tile of a tile: the 16 chunks of the tile before, in a block of their own
#pragma unroll
            for (int R = 0; R < 8; ++R) { chunk(R, 0, par ^ 1, rYp); chunk(R, 1, par ^ 1, rYp); }
        }
        PP_FENCE();
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            rd_set(buf, 1, sg);
            mma_sg(0, sg);
            PP_FENCE();
            dmaB(cB, buf ^ 1, sg);
            PP_FENCE();
        }
        (void)cur_adv(cB, false);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_BARRIER();
        // ---- phase Y
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            if (sg == 3) {
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                PP_BARRIER();
                if (t == 0) stageP(n0c, par);
                rd_set(buf ^ 1, 0, 0); rd_set(buf ^ 1, 0, 1); rd_set(buf ^ 1, 0, 2); rd_set(buf ^ 1, 0, 3);
            }
            mma_sg(1, sg);
            PP_FENCE();
            dmaA(cA, buf, sg);
            PP_FENCE();
        }
        lead_adv();
        buf ^= 1;
        if (t == S - 1) {
            q = next_sb(q);
            rYp = rYc; have_prev = true; par ^= 1;
            if (q >= sb_end) break;
            { int m0n; tile_of(q, m0n, n0c); rYc = make_rY(m0n, n0c); }
            t = 0;
        } else ++t;
    }
    // the last tile's chunks
#pragma unroll
    for (int R = 0; R < 8; ++R) { chunk(R, 0, par ^ 1, rYp); chunk(R, 1, par ^ 1, rYp); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the load stream runs past the last tile: nothing may be in flight when the LDS is given back
}


// returns 1 when the layer does not fit this kernel (the caller then uses conv_gemm_g.hip / conv_gemm_h.hip)
int launch_conv_gemm_pp(sd_ctx* c, const ConvArgs& in, const char* tag)
{
    ConvArgs a = in;
    if (a.prec != 1 || !a.rowtab || !a.W16 || a.X2 || a.item_bias || a.R || a.act2 || a.pad_mode != 0 || a.y_f32 || !a.bias || !a.scale || !a.shift ||
        a.Cout < 256 || (a.Cout % 256) != 0 || (a.y_ld & 7) || a.Cin % 64 != 0 || a.M < 8 * 256 || (a.x_ld & 7)) return 1;
    if (a.w_ld <= 0) a.w_ld = a.Cin;
    if ((a.w_ld & 7) || (int64_t)a.KT * (a.Cin / 64) < 2) return 1;
    // every layer this kernel takes gives the bits of k_conv_gemm_g256<2> (same products, same order) -- the two are compared bit for bit (tests/test_gpu_parity.py);
    // block0 (K = 5 x 128 = 640, ten K-tiles per tile) is taken too since the load stream is continuous (k_conv_gemm_g256 ran it on its 32x32x16 form: 490 TF)
    if ((int64_t)a.Cin * a.KT < 512 || !c->conv_mfma16) return 1;
    if (((size_t)a.X & 15) || ((size_t)a.W16 & 15) || ((size_t)a.Y & 15) || ((size_t)a.bias & 15) || ((size_t)a.scale & 15) || ((size_t)a.shift & 15) || ((size_t)a.rowtab & 15)) return 1;
    const unsigned dev_bit = 1u << (c->device & 31);
    if (!(g_attr_pp.load(std::memory_order_acquire) & dev_bit)) {
        if (hipFuncSetAttribute((const void*)k_conv_gemm_pp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv_gemm_pp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P_LDS) != hipSuccess) { (void)hipGetLastError(); return 1; }
        g_attr_pp.fetch_or(dev_bit, std::memory_order_release);
    }
    a.m_tiles = (a.M + 255) / 256;
    a.n_tiles = a.Cout / 256;
    a.sched = c->conv_pn;
    int grid = (c->num_cu / 8) * 8;
    if (grid < 8) grid = 8;
    const int lx_max = ((a.m_tiles + 7) / 8) * a.n_tiles;
    if (grid / 8 > lx_max) grid = lx_max * 8;
    const int cin = a.cin_real > 0 ? a.cin_real : a.Cin;
    const double flops = 2.0 * (double)a.M * a.Cout * cin * (a.kt_real > 0 ? a.kt_real : a.KT);
    const double bytes = 2.0 * ((double)a.M * cin + (double)a.M * a.Cout + (double)a.Cout * cin * a.KT);
    {
        ProfScope ps(c, c->profile_detail ? std::string("conv_gemm:") + tag : std::string("conv_gemm"), flops, bytes);
        ProfScope ps16(c, "conv_gemm_f16", flops, bytes);
        ProfScope psw(c, "conv_w256_f16", flops, bytes);
        ProfScope pss(c, "conv_w256_ecapa", flops, bytes);
        if (a.act1 == 1) hipLaunchKernelGGL(k_conv_gemm_pp<true>, dim3(grid), dim3(512), P_LDS, c->stream, a);
        else hipLaunchKernelGGL(k_conv_gemm_pp<false>, dim3(grid), dim3(512), P_LDS, c->stream, a);
    }
    if (hipGetLastError() != hipSuccess) SD_FAIL(c, SD_ERR_HIP, "k_conv_gemm_pp launch failed (%s)", tag);
    return SD_OK;
}
