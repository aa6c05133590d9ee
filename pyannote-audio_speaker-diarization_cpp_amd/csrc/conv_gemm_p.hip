// conv_gemm_p.hip -- the fp16 mode's wide layers (Cout >= 256) as a two-group "ping-pong" kernel (round 6).
//
// conv_gemm_g.hip (round 4) stages a whole K-step per barrier and drains its LDS-DMA queue in front of every barrier; it runs at the rate at which a
// 64 KB burst lands.  Round 6 measured the pieces one by one in a stand-alone loop (tools/g256_lab.hip, profiles/r06_g256_lab.txt): a CU takes in
// 24 - 27 bytes per clock whichever way the bytes are requested (LDS-DMA or register loads, 4 or 8 waves issuing, 64 / 128 / 256 workgroups on the
// chip), i.e. 2 400 - 2 700 cycles for the 64 KB a 256 x 256 x 64 K-tile needs, against 2 048 for its MFMAs -- the tile is bound by the CU's ingest,
// and what a kernel can lose on top is the time in which nothing is being requested.  This kernel keeps the queue fed all the time:
//
//  * 8 waves = 2 groups (row halves) x 4 (column quarters); a wave owns 128 x 64 outputs as four 64 x 32 quadrants (16 v_mfma_f32_16x16x32_f16 each).
//    A K-tile is four phases, one per quadrant; a phase = load segment | barrier | MFMA segment | barrier, and group 1 runs one barrier behind group 0:
//    on every SIMD one wave multiplies while its partner reads fragments and issues LDS-DMAs (cdna_hip_programming.md, "The 256^2 8-phase template";
//    MI355X_MICROARCH.md, "Two waves per SIMD").
//  * LDS: 2 buffers x {A half 0, A half 1, B half 0, B half 1} x 128 rows x 128 B.  A half h holds the tile rows BOTH groups read in the same phase
//    (group g's quadrant rows mh are tile rows 128 mh + 64 g .. + 63; B likewise by column quarter), so a half-tile is free two phases after its last
//    read and is refilled for the K-tile after next while this one is still being multiplied.  Every phase stages one half-tile (2 LDS-DMAs per wave)
//    and waits with vmcnt(6): three half-tiles (48 KB) are in flight per CU at any time and no wait empties the queue.
//      phase 0: reads B half 0 + A half 0, stages B half 1 of K-tile T + 1      phase 2: reads A half 1, stages A half 0 of T + 2
//      phase 1: reads B half 1,            stages A half 1 of T + 1             phase 3: (B half 0 stays in registers), stages B half 0 of T + 2
//    (a half-tile staged in phase P is waited for in phase P + 3 by every wave and first read in phase P + 4 or later, behind the barriers in between;
//    it is restaged no earlier than two phases after its last read: "Read a staged buffer one phase AFTER the wait that retires it").
//  * Output: the weight fragment is the MFMA's first operand, so a lane holds four consecutive channels of one row per 16 x 16 block; the loader
//    permutes which weight row goes to which LDS row (LDS row 16 j + rho of a wave's 32 channels holds channel 8 (rho >> 2) + 4 j + (rho & 3)), so
//    blocks j = 0, 1 together give a lane EIGHT consecutive channels = one 16-byte store, no cross-lane transposes, no LDS strip.
//  * Epilogue in 16 chunks (quadrant x 16 rows = 8 values per lane: bias, activation, BatchNorm, one store) that ride behind the MFMAs of the
//    segments around the tile boundary: quadrant (0,0) is final after phase 0 of the tile's last K-tile and goes out in its phases 1 - 2, (0,1) in
//    phases 2 - 3, (1,1) and (1,0) in phases 0 - 2 of the NEXT tile's first K-tile (a chunk zeroes what it has read).  The load stream never stops
//    for a tile boundary.  Parameters of the wave's 64 channels come through a wave-private LDS area (three small LDS-DMAs per tile, two areas by
//    tile parity), the row table entries of the rows a wave stages through another (one LDS-DMA per tile, one tile ahead): no compiler-visible
//    vector-memory load sits in the loop, so hipcc's own vmcnt waits never drain the DMA queue.
//
// Same products in the same order as k_conv_gemm_g256<2> (K-groups of 32 ascending, one 16x16x32 MFMA per group and block), operands swapped.
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
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(ldsaddr), "v"(vo), "s"(rs) : "memory");
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

    // A cursor walks the workgroup's stream of K-tiles: (tile, tap kk, channel chunk kc).  c1 = the K-tile after the one being multiplied (it stages
    // A half 1 and B half 1), c2 = the one after that (A half 0, B half 0).  Each keeps the row offsets of the A rows it stages.
    struct Cur {
        int sb, kk, kc, m0, n0, base, tiles;         // base: first input row of the tile's first item (the buffer descriptor starts there)
        int rr[2], yy[2];                            // per piece: input row of the item's frame 0 minus base; packed frame / last stored frame
        unsigned vo[2];
    };
    auto tile_of = [&](int sb, int& m0, int& n0) { int j, nt; (void)sb_valid(sb, j, nt); m0 = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * 256); n0 = __builtin_amdgcn_readfirstlane(nt * 256); };
    auto table_dma = [&](int m0, int slot) {         // this wave's 32 rows of tile m0: lanes 0-7 rows 16 wid .. + 15 of half 0 (two entries each), lanes 8-15 of half 1
        const unsigned dst = __builtin_amdgcn_readfirstlane(tab0 + (unsigned)(slot * 256));
        if (lane < 16) pp_dma_b128(rTab, dst, (unsigned)(m0 + 128 * (lane >> 3) + 16 * wid + 2 * (lane & 7)) * 8u);
    };
    auto cur_rows = [&](Cur& c, int h) {             // entries of the tile the cursor has just entered (its table slot was filled a tile ago)
        const int2* const tb = (const int2*)(lds + P_TAB + wid * 512 + (c.tiles & 1) * 256) + h * 16;
        const int m0c = c.m0 < a.M ? c.m0 : a.M - 1;
        {   // rowtab[m0c].x by a scalar load written out by hand: hipcc takes the address for divergent and emits a global_load + vmcnt(0), which would drain the DMA queue
            const unsigned long long ad = (unsigned long long)(a.rowtab + m0c);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
            const unsigned long long ads = ((unsigned long long)hi << 32) | lo;
            int bs;
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bs) : "s"(ads) : "memory");
            c.base = bs;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) { const int2 e = tb[p * 8 + prow]; c.rr[p] = e.x - c.base; c.yy[p] = e.y; }
    };
    auto cur_tap = [&](Cur& c) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            int qr = ROWTAB_T(c.yy[p]) + ((c.kk >= ktr ? c.kk - ktr : c.kk) - half) * a.dil;
            const int nd = ROWTAB_LAST(c.yy[p]);
            if (qr < 0) qr = -qr;
            if (qr >= a.Tin) qr = 2 * (a.Tin - 1) - qr;
            if (qr < 0) qr = 0;
            if (qr > nd) qr = nd;
            c.vo[p] = (unsigned)(c.rr[p] + qr) * (unsigned)a.x_ld * 2u + chk[p];
        }
    };
    // returns true when the cursor has entered a new tile
    auto cur_adv = [&](Cur& c, int h) -> bool {
        if (++c.kc < kcs) return false;
        c.kc = 0;
        if (++c.kk < a.KT) { cur_tap(c); return false; }
        c.kk = 0;
        const int nq = next_sb(c.sb);
        if (nq >= sb_end) { cur_tap(c); return false; }          // past the last tile: the stream re-reads the last tile (never multiplied)
        c.sb = nq; ++c.tiles;
        tile_of(nq, c.m0, c.n0);
        cur_rows(c, h);
        cur_tap(c);
        return true;
    };
    auto stageA = [&](const Cur& c, int h, int buf) {
        const v4i rs = pp_rsrc(X + ((size_t)c.base * a.x_ld + (size_t)c.kc * 64), (in_rows - (size_t)c.base) * a.x_ld * 2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * P_BUF + h * P_HALF) + dstw);
        pp_dma_b128(rs, dst, c.vo[0]);
        pp_dma_b128(rs, dst + 1024, c.vo[1]);
    };
    auto stageB = [&](const Cur& c, int h, int buf) {
        const v4i rs = pp_rsrc(W16 + (((size_t)c.kk * a.Cout + c.n0 + 128 * h) * a.w_ld + (size_t)c.kc * 64), 0xffffffffull);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * P_BUF + (2 + h) * P_HALF) + dstw);
        pp_dma_b128(rs, dst, voB[0]);
        pp_dma_b128(rs, dst + 1024, voB[1]);
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
    float4 fa[4][2], fb0[2][2], fb1[2][2];
    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int l = 0; l < 2; ++l) acc[i][j][k][l] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto readA = [&](int buf, int mh) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = *(const float4*)(Afr + buf * P_BUF + mh * P_HALF + i * 2048 + c0);
            fa[i][1] = *(const float4*)(Afr + buf * P_BUF + mh * P_HALF + i * 2048 + c1);
        }
    };
    auto readB = [&](int buf, int nh, float4 (&fb)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[j][0] = *(const float4*)(Bfr + buf * P_BUF + nh * P_HALF + j * 2048 + c0);
            fb[j][1] = *(const float4*)(Bfr + buf * P_BUF + nh * P_HALF + j * 2048 + c1);
        }
    };
    auto mma = [&](int mh, int nh, const float4 (&fb)[2][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[mh][nh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fb[j][ks]), __builtin_bit_cast(half8, fa[i][ks]), acc[mh][nh][i][j], 0, 0, 0);
    };
    // one epilogue chunk: rows 128 mh + 64 g + 16 i + l15, channels 128 nh + 32 wc + 8 l4 .. + 7 of the tile behind `rY` (rows >= M fall outside the descriptor)
    const float slope = (a.act1 == 1) ? 0.0f : ((a.act1 == 2) ? 0.01f : 1.0f);
    const unsigned ybytes = (unsigned)a.y_ld * 2u;
    const unsigned voY = (unsigned)(64 * g + l15) * ybytes + (unsigned)(32 * wc + 8 * l4) * 2u;
    auto chunk = [&](int mh, int nh, int i, int par, __amdgpu_buffer_rsrc_t rY) {
        const float* const pw = (const float*)(lds + P_PAR + wid * 2048 + par * 1024) + (nh * 8 + 2 * l4) * 4;
        half8 hv;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {          // four channels at a time; the fences keep one half's twelve parameters live, not the next chunks' as well
            asm volatile("" ::: "memory");
            const float4 b = *(const float4*)(pw + 4 * hf), sc = *(const float4*)(pw + 64 + 4 * hf), sh = *(const float4*)(pw + 128 + 4 * hf);
            const float bb[4] = {b.x, b.y, b.z, b.w}, ss[4] = {sc.x, sc.y, sc.z, sc.w}, hh[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[mh][nh][i][hf][e] + bb[e];
                acc[mh][nh][i][hf][e] = 0.0f;
                v = fmaxf(v, v * slope);
                hv[4 * hf + e] = (_Float16)(v * ss[e] + hh[e]);
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, hv), rY, voY + (unsigned)(128 * mh + 16 * i) * ybytes, 256u * nh, 0);
    };
    _Float16* const Y = (_Float16*)a.Y;
    auto make_rY = [&](int m0, int n0) {
        const int rows_left = a.M - m0;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(Y + (size_t)m0 * a.y_ld + n0), 0, (unsigned)((size_t)(rows_left < 256 ? rows_left : 256) * a.y_ld * 2), 0x00020000);
    };

#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_VM6() asm volatile("s_waitcnt vmcnt(6)" ::: "memory")
#define PP_MSEG_BEGIN() do { PP_BARRIER(); PP_LGKM0(); __builtin_amdgcn_s_setprio(1); } while (0)
#define PP_MSEG_END() do { __builtin_amdgcn_s_setprio(0); PP_BARRIER(); } while (0)

    // ---- prologue: the first tile's row table entries (and the second tile's, one tile ahead), K-tile 0 whole, A half 0 and B half 0 of K-tile 1
    Cur cc; cc.sb = q0; cc.kk = 0; cc.kc = 0; cc.tiles = 0;
    tile_of(q0, cc.m0, cc.n0);
    table_dma(cc.m0, 0);
    { const int nq = next_sb(q0); if (nq < sb_end) { int m1, n1; tile_of(nq, m1, n1); table_dma(m1, 1); } }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Cur c1_ = cc, c2_ = cc;
    cur_rows(c1_, 1); cur_tap(c1_);
    cur_rows(c2_, 0); cur_tap(c2_);
    stageA(c2_, 0, 0); stageB(c2_, 0, 0); stageB(c1_, 1, 0); stageA(c1_, 1, 0);
    // (a cursor that enters a new tile asks for the row table entries of the tile AFTER it: only c2, the leading one, does)
    auto lead_adv = [&]() {
        if (cur_adv(c2_, 0)) { const int nq = next_sb(c2_.sb); if (nq < sb_end) { int m1, n1; tile_of(nq, m1, n1); table_dma(m1, (c2_.tiles + 1) & 1); } }
    };
    lead_adv();
    stageA(c2_, 0, 1); stageB(c2_, 0, 1);
    (void)cur_adv(c1_, 1);
    lead_adv();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier behind

    int q = q0, t = 0, buf = 0, par = 0;
    bool have_prev = false;
    __amdgpu_buffer_rsrc_t rYc = make_rY(cc.m0, cc.n0), rYp = rYc;
    int n0c = cc.n0;
    while (true) {
        const bool last = t == S - 1, carry = t == 0 && have_prev;      // the tile's last K-tile carries chunks of quadrants (0,0), (0,1); its first K-tile those of the tile before
        // phase 0: quadrant (0, 0); stages B half 1 of the next K-tile
        readB(buf, 0, fb0); __builtin_amdgcn_sched_barrier(0); readA(buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        stageB(c1_, 1, buf ^ 1); PP_VM6();
        PP_MSEG_BEGIN();
        mma(0, 0, fb0);
        if (carry) { chunk(1, 1, 0, par ^ 1, rYp); chunk(1, 1, 1, par ^ 1, rYp); chunk(1, 1, 2, par ^ 1, rYp); }
        PP_MSEG_END();
        // phase 1: quadrant (0, 1); stages A half 1 of the next K-tile
        readB(buf, 1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        stageA(c1_, 1, buf ^ 1); PP_VM6();
        (void)cur_adv(c1_, 1);
        PP_MSEG_BEGIN();
        mma(0, 1, fb1);
        if (last) { chunk(0, 0, 0, par, rYc); chunk(0, 0, 1, par, rYc); }
        if (carry) { chunk(1, 1, 3, par ^ 1, rYp); chunk(1, 0, 0, par ^ 1, rYp); chunk(1, 0, 1, par ^ 1, rYp); }
        PP_MSEG_END();
        // phase 2: quadrant (1, 1); stages A half 0 of the K-tile after next (A half 0 of this one was last read in phase 0)
        readA(buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        stageA(c2_, 0, buf); PP_VM6();
        PP_MSEG_BEGIN();
        mma(1, 1, fb1);
        if (last) { chunk(0, 0, 2, par, rYc); chunk(0, 0, 3, par, rYc); chunk(0, 1, 0, par, rYc); }
        if (carry) { chunk(1, 0, 2, par ^ 1, rYp); chunk(1, 0, 3, par ^ 1, rYp); }
        PP_MSEG_END();
        // phase 3: quadrant (1, 0); stages B half 0 of the K-tile after next (and, in a tile's first K-tile, its parameters)
        if (t == 0) stageP(n0c, par);
        stageB(c2_, 0, buf); PP_VM6();
        lead_adv();
        PP_BARRIER(); __builtin_amdgcn_s_setprio(1);
        mma(1, 0, fb0);
        if (last) { chunk(0, 1, 1, par, rYc); chunk(0, 1, 2, par, rYc); chunk(0, 1, 3, par, rYc); }
        PP_MSEG_END();

        buf ^= 1;
        if (last) {
            q = next_sb(q);
            rYp = rYc; have_prev = true; par ^= 1;
            if (q >= sb_end) break;
            { int m0n; tile_of(q, m0n, n0c); rYc = make_rY(m0n, n0c); }
            t = 0;
        } else ++t;
    }
    // the last tile's quadrants (1,1) and (1,0)
#pragma unroll
    for (int i = 0; i < 4; ++i) chunk(1, 1, i, par ^ 1, rYp);
#pragma unroll
    for (int i = 0; i < 4; ++i) chunk(1, 0, i, par ^ 1, rYp);
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
    // contractions below 1 024 (block0: K = 640) stay with k_conv_gemm_g256<1>, as they do there (32x32x16 form): every layer this kernel takes is one that
    // k_conv_gemm_g256<2> would take, with the same products in the same order -- the two are compared bit for bit (tests/test_gpu_parity.py)
    if ((int64_t)a.Cin * a.KT < 1024 || !c->conv_mfma16) return 1;
    if (((size_t)a.X & 15) || ((size_t)a.W16 & 15) || ((size_t)a.Y & 15) || ((size_t)a.bias & 15) || ((size_t)a.scale & 15) || ((size_t)a.shift & 15) || ((size_t)a.rowtab & 15)) return 1;
    const unsigned dev_bit = 1u << (c->device & 31);
    if (!(g_attr_pp.load(std::memory_order_acquire) & dev_bit)) {
        if (hipFuncSetAttribute((const void*)k_conv_gemm_pp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P_LDS) != hipSuccess) { (void)hipGetLastError(); return 1; }
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
        hipLaunchKernelGGL(k_conv_gemm_pp, dim3(grid), dim3(512), P_LDS, c->stream, a);
    }
    if (hipGetLastError() != hipSuccess) SD_FAIL(c, SD_ERR_HIP, "k_conv_gemm_pp launch failed (%s)", tag);
    return SD_OK;
}
