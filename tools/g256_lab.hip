// g256_lab.hip -- bench + check of wide fp16 GEMM loop structures for k_conv_gemm_g256 (y[M][N] = x[M][K] . w[N][K]^T, f32 accumulation, fp16 out)
// on the shapes of ECAPA's wide layers, standalone (no library): the place where a K-loop structure is tried before it is carried into
// csrc/conv_gemm_g.hip.  Measurement and development only; nothing in the product path uses this file.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/g256_lab tools/g256_lab.hip ; tools/bin/g256_lab [rows] [variant mask]
//
// Variant PP ("ping-pong"): 256 x 256 tile, K-tile of 64 halves, 8 waves = 2 groups (rows) x 4 (columns); the two groups run half a phase apart
// (group 1 passes one extra barrier at the start), so that on every SIMD one wave issues MFMAs while its partner reads fragments and issues
// LDS-DMAs (cdna_hip_programming.md, "The 256^2 8-phase template"; MI355X_MICROARCH.md, "Two waves per SIMD").  A K-tile is four phases, one
// per 64 x 32 quadrant of the wave's 128 x 64 outputs (16 v_mfma_f32_16x16x32_f16 each); a phase = load segment | barrier | MFMA segment | barrier.
// LDS: 2 buffers x {A half 0, A half 1, B half 0, B half 1} x 128 rows x 128 B = 128 KB.  Half h of A holds the tile rows both groups consume in
// THE SAME phase (group g's quadrant rows mh are tile rows 128 mh + 64 g ..+63), so a half-tile is free again two phases after it was read and
// is re-filled for the K-tile after next while this K-tile is still being multiplied: every phase stages one half-tile (2 LDS-DMAs per wave)
// and waits with vmcnt(6) -- three half-tiles (48 KB) stay in flight per CU at all times, none of the waits drains the queue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;

struct Args {
    const _Float16* X; const _Float16* W; _Float16* Y;
    int M, N, K;          // M, N multiples of 256, K of 64
    int m_tiles, n_tiles;
    int pnmax;            // column tiles per super-block
    unsigned long long* clk;
    const float* P;       // [3][N]: bias, BatchNorm scale, shift (k_pp2)
    int skew;             // probe: workgroup (pm, pn) walks K from K-tile (pn * S / PN + pm * S / (PM * PN)) % S on (k_pp only)
};

#define HALF_BYTES 16384
#define BUF_BYTES 65536

__device__ __forceinline__ void lds_dma_b128(v4i rs, unsigned ldsaddr, unsigned vo, unsigned so)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(ldsaddr), "v"(vo), "s"(rs), "s"(so) : "memory");
}

__device__ __forceinline__ v4i make_rsrc(const void* base)
{
    const unsigned long long b = (unsigned long long)base;
    v4i r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffffu));
    r[2] = (int)0xffffffffu;
    r[3] = 0x00020000;
    return r;
}

// VAR bits (diagnostics): 1 no stagger between the groups, 2 every wait drains the queue (vmcnt(0)), 4 no MFMA, 8 no LDS-DMA in the loop, 16 no fragment reads
template <int VAR>
__global__ __launch_bounds__(512) void k_pp(Args a)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int w = blockIdx.x, G = gridDim.x;
    const int xcd = w & 7, wl = w >> 3, wpx = G >> 3;
    const int mx = (a.m_tiles - xcd + 7) >> 3;
    const int PN = a.n_tiles < a.pnmax ? a.n_tiles : a.pnmax;
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
    const int S = a.K / 64;
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)lds;
    const int kskew = a.skew ? (pn * S / PN + pm * S / (PM * PN)) % S : 0;

    // ---- loader.  A half-tile = 128 rows x 128 B = 16 wave instructions of 1 KB (8 rows); wave `wid` fills rows 16 wid + 8 p + (lane >> 3), p = 0, 1.
    // Lane l writes LDS bytes [16 l, 16 l + 16) of the piece = row l >> 3, chunk POSITION l & 7, which holds the logical chunk (l & 7) ^ ((row >> 1) & 7).
    unsigned voA[2], voB[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = wid * 16 + p * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((r >> 1) & 7);
        voA[p] = (unsigned)r * (unsigned)a.K * 2u + (unsigned)ch * 16u;
        voB[p] = voA[p];
    }
    const unsigned dstw = __builtin_amdgcn_readfirstlane((unsigned)(wid * 2048));
    // two cursors over the workgroup's stream of K-tiles: c1 = the K-tile after the one being multiplied, c2 = the one after that
    struct Cur { int sb; int t; int m0, n0; };
    auto cur_set = [&](Cur& c) { int j, nt; (void)sb_valid(c.sb, j, nt); c.m0 = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * 256); c.n0 = __builtin_amdgcn_readfirstlane(nt * 256); };
    auto cur_adv = [&](Cur& c) {
        if (++c.t < S) return;
        c.t = 0;
        const int nq = next_sb(c.sb);
        if (nq < sb_end) { c.sb = nq; cur_set(c); }          // past the last tile: the stream re-reads the last tile (never multiplied)
    };
    f32x4 sink[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { sink[i][0] = f32x4{0, 0, 0, 0}; sink[i][1] = f32x4{0, 0, 0, 0}; }
    auto stageA = [&](const Cur& c, int h, int buf) {
        const v4i rs = make_rsrc((const char*)a.X + ((size_t)(c.m0 + 128 * h) * a.K + (size_t)((c.t + kskew) % S) * 64) * 2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF_BYTES + h * HALF_BYTES) + dstw);
        if (VAR & 128) {       // the same bytes to REGISTERS (never used): what the CU takes in without the LDS write path
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(sink[h ? 1 : 2][0]) : "v"(voA[0]), "s"(rs) : "memory");      // (A half 1 goes out in phase 1, half 0 in phase 2)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(sink[h ? 1 : 2][1]) : "v"(voA[1]), "s"(rs) : "memory");
            return;
        }
        lds_dma_b128(rs, dst, voA[0], 0);
        lds_dma_b128(rs, dst + 1024, voA[1], 0);
    };
    auto stageB = [&](const Cur& c, int h, int buf) {
        const v4i rs = make_rsrc((const char*)a.W + ((size_t)(c.n0 + 128 * h) * a.K + (size_t)((c.t + kskew) % S) * 64) * 2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF_BYTES + (2 + h) * HALF_BYTES) + dstw);
        if (VAR & 128) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(sink[h ? 0 : 3][0]) : "v"(voB[0]), "s"(rs) : "memory");      // (B half 1 in phase 0, half 0 in phase 3)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(sink[h ? 0 : 3][1]) : "v"(voB[1]), "s"(rs) : "memory");
            return;
        }
        lds_dma_b128(rs, dst, voB[0], 0);
        lds_dma_b128(rs, dst + 1024, voB[1], 0);
    };

    // ---- fragments.  Lane l reads k = 32 ks + 8 (l >> 4) .. + 7 (logical chunk 4 ks + (l >> 4)) of row l & 15 of a 16-row block.
    const int sw = (l15 >> 1) & 7;
    const int c0 = (l4 ^ sw) * 16, c1 = ((4 + l4) ^ sw) * 16;
    const char* const Afr = lds + (64 * g + l15) * 128;                      // + mh * HALF_BYTES + i * 2048 + c{ks}
    const char* const Bfr = lds + 2 * HALF_BYTES + (32 * wc + l15) * 128;    // + nh * HALF_BYTES + j * 2048 + c{ks}
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
        if (VAR & 16) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = *(const float4*)(Afr + buf * BUF_BYTES + mh * HALF_BYTES + i * 2048 + c0);
            fa[i][1] = *(const float4*)(Afr + buf * BUF_BYTES + mh * HALF_BYTES + i * 2048 + c1);
        }
    };
    auto readB = [&](int buf, int nh, float4 (&fb)[2][2]) {
        if (VAR & 16) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[j][0] = *(const float4*)(Bfr + buf * BUF_BYTES + nh * HALF_BYTES + j * 2048 + c0);
            fb[j][1] = *(const float4*)(Bfr + buf * BUF_BYTES + nh * HALF_BYTES + j * 2048 + c1);
        }
    };
    auto mma = [&](int mh, int nh, const float4 (&fb)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!(VAR & 4)) acc[mh][nh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fb[j][ks]), __builtin_bit_cast(half8, fa[i][ks]), acc[mh][nh][i][j], 0, 0, 0);
        if (VAR & 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(fb[j][0].x), "v"(fb[j][0].w), "v"(fb[j][1].x), "v"(fb[j][1].w));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(fa[i][0].x), "v"(fa[i][0].w), "v"(fa[i][1].x), "v"(fa[i][1].w));
        }
        __builtin_amdgcn_s_setprio(0);
    };
#define SEG_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VM6() do { if (VAR & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); } while (0)

    // ---- prologue: K-tile 0 whole, A half 0 and B half 0 of K-tile 1
    Cur cc; cc.sb = q0; cc.t = 0; cur_set(cc);
    Cur c1_ = cc, c2_ = cc;
    stageA(cc, 0, 0); stageB(cc, 0, 0); stageB(cc, 1, 0); stageA(cc, 1, 0);
    cur_adv(c1_); c2_ = c1_;
    stageA(c1_, 0, 1); stageB(c1_, 0, 1);
    cur_adv(c2_);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g == 1 && !(VAR & 1)) __builtin_amdgcn_s_barrier();

    int q = q0, t = 0, buf = 0;
    int m0c = cc.m0, n0c = cc.n0;
    unsigned long long t_start = 0;
    if (a.clk) t_start = __builtin_amdgcn_s_memtime();
    while (true) {
        // phase 0: quadrant (0, 0); stages B half 1 of the next K-tile
        readB(buf, 0, fb0); __builtin_amdgcn_sched_barrier(0); readA(buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(VAR & 8)) stageB(c1_, 1, buf ^ 1); VM6();
        SEG_BARRIER(); LGKM0();
        mma(0, 0, fb0);
        SEG_BARRIER();
        // phase 1: quadrant (0, 1); stages A half 1 of the next K-tile
        readB(buf, 1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(VAR & 8)) stageA(c1_, 1, buf ^ 1); VM6();
        cur_adv(c1_);
        SEG_BARRIER(); LGKM0();
        mma(0, 1, fb1);
        SEG_BARRIER();
        // phase 2: quadrant (1, 1); stages A half 0 of the K-tile after next (A half 0 of this one was last read in phase 0)
        readA(buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(VAR & 8)) stageA(c2_, 0, buf); VM6();
        SEG_BARRIER(); LGKM0();
        mma(1, 1, fb1);
        SEG_BARRIER();
        // phase 3: quadrant (1, 0); stages B half 0 of the K-tile after next
        if (!(VAR & 8)) stageB(c2_, 0, buf); VM6();
        cur_adv(c2_);
        SEG_BARRIER();
        mma(1, 0, fb0);
        SEG_BARRIER();

        if (t == S - 1) {
            // ---- epilogue (lab form): register r of acc[mh][nh][i][j], lane l = row 128 mh + 64 g + 16 i + (l & 15), column 128 nh + 32 wc + 16 j + 4 (l >> 4) + r
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int row = m0c + 128 * mh + 64 * g + 16 * i + l15;
                            const int col = n0c + 128 * nh + 32 * wc + 16 * j + 4 * l4;
                            f32x4& v = acc[mh][nh][i][j];
                            const half4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                            *(half4*)(a.Y + (size_t)row * a.N + col) = hv;
                            v = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // lab form: the stores drain here (the product form counts them in the next phases' waits)
            q = next_sb(q);
            if (q >= sb_end) break;
            { int j_, nt_; (void)sb_valid(q, j_, nt_); m0c = __builtin_amdgcn_readfirstlane((xcd + 8 * j_) * 256); n0c = __builtin_amdgcn_readfirstlane(nt_ * 256); }
            t = 0;
        } else ++t;
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (VAR & 128) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(sink[i][0]), "v"(sink[i][1]));
    }
    if (a.clk && tid == 0) a.clk[blockIdx.x] = __builtin_amdgcn_s_memtime() - t_start;
}


// ---- k_pp2: the PP loop + what the product kernel needs around it:
//  * 16-byte output stores straight from the accumulators: LDS row 16 j + rho of a wave's 32-column group holds output column 8 (rho >> 2) + 4 j + (rho & 3)
//    (a permutation in the loader's SOURCE addresses only), so register r of blocks j = 0, 1 of lane (l15, l4) are columns 8 l4 .. 8 l4 + 7 of row l15;
//  * bias / ReLU / BatchNorm parameters of the wave's 64 columns through a wave-private LDS area (three small LDS-DMAs per tile, two areas by tile parity);
//  * the epilogue cut into 16 chunks (quadrant x 16 rows: 8 values per lane, one store) that ride in the MFMA segments around the tile boundary: quadrant
//    (0,0) is final after phase 0 of the tile's last K-tile and is written out during phases 1, 2 of that K-tile, ... quadrant (1,0) during phases 1, 2 of
//    the next tile's first K-tile; the first MFMAs of a quadrant in a new tile take C = 0.  NV = LDS-DMA instructions left in flight by every wait.
template <int NV, int VAR>
__global__ __launch_bounds__(512) void k_pp2(Args a)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int w = blockIdx.x, G = gridDim.x;
    const int xcd = w & 7, wl = w >> 3, wpx = G >> 3;
    const int mx = (a.m_tiles - xcd + 7) >> 3;
    const int PN = a.n_tiles < a.pnmax ? a.n_tiles : a.pnmax;
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
    const int S = a.K / 64;
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)lds;

    unsigned voA[2], voB[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = wid * 16 + p * 8 + (lane >> 3);                 // LDS row of the half-tile
        const int ch = (lane & 7) ^ ((r >> 1) & 7);
        voA[p] = (unsigned)r * (unsigned)a.K * 2u + (unsigned)ch * 16u;
        const int rho = r & 15;
        const int col = 32 * (wid >> 1) + 8 * (rho >> 2) + 4 * (wid & 1) + (rho & 3);     // output column (within the half) whose weights that LDS row holds
        voB[p] = (unsigned)col * (unsigned)a.K * 2u + (unsigned)ch * 16u;
    }
    const unsigned dstw = __builtin_amdgcn_readfirstlane((unsigned)(wid * 2048));
    struct Cur { int sb; int t; int m0, n0; };
    auto cur_set = [&](Cur& c) { int j, nt; (void)sb_valid(c.sb, j, nt); c.m0 = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * 256); c.n0 = __builtin_amdgcn_readfirstlane(nt * 256); };
    auto cur_adv = [&](Cur& c) {
        if (++c.t < S) return;
        c.t = 0;
        const int nq = next_sb(c.sb);
        if (nq < sb_end) { c.sb = nq; cur_set(c); }
    };
    auto stageA = [&](const Cur& c, int h, int buf, int pieces) {
        const v4i rs = make_rsrc((const char*)a.X + ((size_t)(c.m0 + 128 * h) * a.K + (size_t)c.t * 64) * 2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF_BYTES + h * HALF_BYTES) + dstw);
        if (pieces & 1) lds_dma_b128(rs, dst, voA[0], 0);
        if (pieces & 2) lds_dma_b128(rs, dst + 1024, voA[1], 0);
    };
    auto stageB = [&](const Cur& c, int h, int buf, int pieces) {
        const v4i rs = make_rsrc((const char*)a.W + ((size_t)(c.n0 + 128 * h) * a.K + (size_t)c.t * 64) * 2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF_BYTES + (2 + h) * HALF_BYTES) + dstw);
        if (pieces & 1) lds_dma_b128(rs, dst, voB[0], 0);
        if (pieces & 2) lds_dma_b128(rs, dst + 1024, voB[1], 0);
    };
    // parameters of the wave's 64 columns: lane k < 16 of DMA `arr` fetches columns 128 (k >> 3) + 32 wc + 4 (k & 7) .. + 3 -> floats [arr][k][4] of the area
    const unsigned par0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(2 * BUF_BYTES + wid * 2048));
    const unsigned voP = (unsigned)(128 * ((lane & 15) >> 3) + 32 * wc + 4 * (lane & 7)) * 4u;
    auto stageP = [&](int n0, int par) {
        if (lane < 16) {
#pragma unroll
            for (int arr = 0; arr < 3; ++arr) {
                const v4i rs = make_rsrc((const char*)(a.P + (size_t)arr * a.N + n0));
                lds_dma_b128(rs, par0 + (unsigned)(par * 1024 + arr * 256), voP, 0);
            }
        }
    };

    const int sw = (l15 >> 1) & 7;
    const int c0 = (l4 ^ sw) * 16, c1 = ((4 + l4) ^ sw) * 16;
    const char* const Afr = lds + (64 * g + l15) * 128;
    const char* const Bfr = lds + 2 * HALF_BYTES + (32 * wc + l15) * 128;
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
            fa[i][0] = *(const float4*)(Afr + buf * BUF_BYTES + mh * HALF_BYTES + i * 2048 + c0);
            fa[i][1] = *(const float4*)(Afr + buf * BUF_BYTES + mh * HALF_BYTES + i * 2048 + c1);
        }
    };
    auto readB = [&](int buf, int nh, float4 (&fb)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[j][0] = *(const float4*)(Bfr + buf * BUF_BYTES + nh * HALF_BYTES + j * 2048 + c0);
            fb[j][1] = *(const float4*)(Bfr + buf * BUF_BYTES + nh * HALF_BYTES + j * 2048 + c1);
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
    // one epilogue chunk: rows 128 mh + 64 g + 16 i + l15, columns 128 nh + 32 wc + 8 l4 .. + 7 of the tile behind `rY`
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const unsigned nrow16 = __builtin_amdgcn_readfirstlane((unsigned)a.N * 2u * 16u);
    const unsigned voY = (unsigned)(64 * g + l15) * (unsigned)a.N * 2u + (unsigned)(32 * wc + 8 * l4) * 2u;
    auto chunk = [&](int mh, int nh, int i, int par, __amdgpu_buffer_rsrc_t rY) {
        const float* const pw = (const float*)(lds + 2 * BUF_BYTES + wid * 2048 + par * 1024) + (nh * 8 + 2 * l4) * 4;
        half8 hv;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {          // four columns at a time; the fences keep one half's twelve parameters live, not the next chunks' as well (MFMAs and VALU still move across them)
            asm volatile("" ::: "memory");
            const float4 b = *(const float4*)(pw + 4 * hf), sc = *(const float4*)(pw + 64 + 4 * hf), sh = *(const float4*)(pw + 128 + 4 * hf);
            const float bb[4] = {b.x, b.y, b.z, b.w}, ss[4] = {sc.x, sc.y, sc.z, sc.w}, hh[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[mh][nh][i][hf][e] + bb[e];
                acc[mh][nh][i][hf][e] = 0.0f;
                v = fmaxf(v, 0.0f);
                hv[4 * hf + e] = (_Float16)(v * ss[e] + hh[e]);
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, hv), rY, voY, (unsigned)(8 * mh + i) * nrow16 + 256u * nh, 0);
    };
    auto make_rY = [&](int m0, int n0) { return __builtin_amdgcn_make_buffer_rsrc((void*)(a.Y + (size_t)m0 * a.N + n0), 0, 0x7fffffff, 0x00020000); };

#define VMN() do { if (NV == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (NV == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } while (0)
#define MSEG_BEGIN() do { SEG_BARRIER(); LGKM0(); __builtin_amdgcn_s_setprio(1); } while (0)
#define MSEG_END() do { __builtin_amdgcn_s_setprio(0); SEG_BARRIER(); } while (0)

    Cur cc; cc.sb = q0; cc.t = 0; cur_set(cc);
    Cur c1_ = cc, c2_ = cc;
    stageA(cc, 0, 0, 3); stageB(cc, 0, 0, 3); stageB(cc, 1, 0, 3); stageA(cc, 1, 0, 3);
    cur_adv(c1_); c2_ = c1_;
    stageA(c1_, 0, 1, 3); stageB(c1_, 0, 1, 3);
    cur_adv(c2_);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g == 1 && !(VAR & 1)) __builtin_amdgcn_s_barrier();

    int q = q0, t = 0, buf = 0, par = 0;
    bool have_prev = false;
    __amdgpu_buffer_rsrc_t rYc = make_rY(cc.m0, cc.n0), rYp = rYc;
    int n0c = cc.n0;
    unsigned long long t_start = 0;
    if (a.clk) t_start = __builtin_amdgcn_s_memtime();

    // VAR 512: in-kernel clocks (waves 0 and 4): per phase, cycles of [reads + DMA issue | vmcnt wait | barrier + lgkmcnt | MFMA issue | closing barrier]
    unsigned st[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) st[i][j] = 0;
    unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0;
#define STAMP(v) do { if (VAR & 512) v = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_FOLD(q_, tend) do { if (VAR & 512) { st[q_][0] += (unsigned)(tk1 - tk0); st[q_][1] += (unsigned)(tk2 - tk1); st[q_][2] += (unsigned)(tk3 - tk2); st[q_][3] += (unsigned)(tk4 - tk3); st[q_][4] += (unsigned)((tend) - tk4); } } while (0)
    // VAR 32: the phase's LDS-DMAs go out BEFORE its fragment reads (the memory pipe has work from the first cycle of the slot); 64: one of the two at the head of the
    // load segment, the other behind the MFMAs of the wave's own MFMA segment; 256: chunks in one basic block with the MFMAs (the scheduler interleaves them)
    constexpr int P_HEAD = (VAR & 64) ? 1 : (VAR & 32) ? 3 : 0, P_MID = (VAR & (32 | 64)) ? 0 : 3, P_TAIL = (VAR & 64) ? 2 : 0;
#define MSEG(MH, NH, FB, LASTC, CARRYC) do { \
        if (VAR & 256) { if (last) { mma(MH, NH, FB); LASTC; } else if (carry) { mma(MH, NH, FB); CARRYC; } else mma(MH, NH, FB); } \
        else { mma(MH, NH, FB); if (last) { LASTC; } if (carry) { CARRYC; } } } while (0)
    while (true) {
        const bool last = t == S - 1, carry = t == 0 && have_prev;      // the tile's last K-tile carries chunks of quadrants (0,0), (0,1); its first K-tile those of the tile before
        // phase 0: quadrant (0, 0); stages B half 1 of the next K-tile
        STAMP(tk0);
        if (P_HEAD) stageB(c1_, 1, buf ^ 1, P_HEAD);
        readB(buf, 0, fb0); __builtin_amdgcn_sched_barrier(0); readA(buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (P_MID) stageB(c1_, 1, buf ^ 1, P_MID);
        STAMP(tk1); VMN(); STAMP(tk2);
        MSEG_BEGIN(); STAMP(tk3);
        MSEG(0, 0, fb0, (void)0, (chunk(1, 1, 0, par ^ 1, rYp), chunk(1, 1, 1, par ^ 1, rYp), chunk(1, 1, 2, par ^ 1, rYp)));
        __builtin_amdgcn_s_setprio(0);
        if (P_TAIL) { __builtin_amdgcn_sched_barrier(0); stageB(c1_, 1, buf ^ 1, P_TAIL); }
        STAMP(tk4);
        SEG_BARRIER();
        // phase 1: quadrant (0, 1); stages A half 1 of the next K-tile
        { unsigned long long te = 0; STAMP(te); STAMP_FOLD(0, te); tk0 = te; }
        if (P_HEAD) stageA(c1_, 1, buf ^ 1, P_HEAD);
        readB(buf, 1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        if (P_MID) stageA(c1_, 1, buf ^ 1, P_MID);
        STAMP(tk1); VMN(); STAMP(tk2);
        Cur c1n = c1_; cur_adv(c1n);
        MSEG_BEGIN(); STAMP(tk3);
        MSEG(0, 1, fb1, (chunk(0, 0, 0, par, rYc), chunk(0, 0, 1, par, rYc)), (chunk(1, 1, 3, par ^ 1, rYp), chunk(1, 0, 0, par ^ 1, rYp), chunk(1, 0, 1, par ^ 1, rYp)));
        __builtin_amdgcn_s_setprio(0);
        if (P_TAIL) { __builtin_amdgcn_sched_barrier(0); stageA(c1_, 1, buf ^ 1, P_TAIL); }
        c1_ = c1n;
        STAMP(tk4);
        SEG_BARRIER();
        // phase 2: quadrant (1, 1); stages A half 0 of the K-tile after next
        { unsigned long long te = 0; STAMP(te); STAMP_FOLD(1, te); tk0 = te; }
        if (P_HEAD) stageA(c2_, 0, buf, P_HEAD);
        readA(buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (P_MID) stageA(c2_, 0, buf, P_MID);
        STAMP(tk1); VMN(); STAMP(tk2);
        MSEG_BEGIN(); STAMP(tk3);
        MSEG(1, 1, fb1, (chunk(0, 0, 2, par, rYc), chunk(0, 0, 3, par, rYc), chunk(0, 1, 0, par, rYc)), (chunk(1, 0, 2, par ^ 1, rYp), chunk(1, 0, 3, par ^ 1, rYp)));
        __builtin_amdgcn_s_setprio(0);
        if (P_TAIL) { __builtin_amdgcn_sched_barrier(0); stageA(c2_, 0, buf, P_TAIL); }
        STAMP(tk4);
        SEG_BARRIER();
        // phase 3: quadrant (1, 0); stages B half 0 of the K-tile after next (and, in a tile's first K-tile, its parameters)
        { unsigned long long te = 0; STAMP(te); STAMP_FOLD(2, te); tk0 = te; }
        if (t == 0) stageP(n0c, par);
        if (P_HEAD | P_MID) stageB(c2_, 0, buf, P_HEAD | P_MID);
        STAMP(tk1); VMN(); STAMP(tk2);
        Cur c2n = c2_; cur_adv(c2n);
        SEG_BARRIER(); __builtin_amdgcn_s_setprio(1); STAMP(tk3);
        MSEG(1, 0, fb0, (chunk(0, 1, 1, par, rYc), chunk(0, 1, 2, par, rYc), chunk(0, 1, 3, par, rYc)), (void)0);
        __builtin_amdgcn_s_setprio(0);
        if (P_TAIL) { __builtin_amdgcn_sched_barrier(0); stageB(c2_, 0, buf, P_TAIL); }
        c2_ = c2n;
        STAMP(tk4);
        SEG_BARRIER();
        { unsigned long long te = 0; STAMP(te); STAMP_FOLD(3, te); }

        buf ^= 1;
        if (last) {
            q = next_sb(q);
            rYp = rYc; have_prev = true; par ^= 1;
            if (q >= sb_end) break;
            { int j_, nt_; (void)sb_valid(q, j_, nt_); n0c = __builtin_amdgcn_readfirstlane(nt_ * 256); rYc = make_rY(__builtin_amdgcn_readfirstlane((xcd + 8 * j_) * 256), n0c); }
            t = 0;
        } else ++t;
    }
    // the last tile's quadrants (1,1) and (1,0)
#pragma unroll
    for (int i = 0; i < 4; ++i) chunk(1, 1, i, par ^ 1, rYp);
#pragma unroll
    for (int i = 0; i < 4; ++i) chunk(1, 0, i, par ^ 1, rYp);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.clk && tid == 0) a.clk[blockIdx.x] = __builtin_amdgcn_s_memtime() - t_start;
    if ((VAR & 512) && a.clk && (tid == 0 || tid == 256)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) a.clk[512 + (blockIdx.x * 2 + (tid >> 8)) * 20 + i * 5 + j] = st[i][j];
    }
}

template <int NV, int VAR>
__global__ __launch_bounds__(512) void k_pp3(Args a)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int w = blockIdx.x, G = gridDim.x;
    const int xcd = w & 7, wl = w >> 3, wpx = G >> 3;
    const int mx = (a.m_tiles - xcd + 7) >> 3;
    const int PN = a.n_tiles < a.pnmax ? a.n_tiles : a.pnmax;
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
    const int S = a.K / 64;
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)lds;

    unsigned voA[2], voB[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = wid * 16 + p * 8 + (lane >> 3);                 // LDS row of the half-tile
        const int ch = (lane & 7) ^ ((r >> 1) & 7);
        voA[p] = (unsigned)r * (unsigned)a.K * 2u + (unsigned)ch * 16u;
        const int rho = r & 15;
        const int col = 32 * (wid >> 1) + 8 * (rho >> 2) + 4 * (wid & 1) + (rho & 3);     // output column (within the half) whose weights that LDS row holds
        voB[p] = (unsigned)col * (unsigned)a.K * 2u + (unsigned)ch * 16u;
    }
    const unsigned dstw = __builtin_amdgcn_readfirstlane((unsigned)(wid * 2048));
    struct Cur { int sb; int t; int m0, n0; };
    auto cur_set = [&](Cur& c) { int j, nt; (void)sb_valid(c.sb, j, nt); c.m0 = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * 256); c.n0 = __builtin_amdgcn_readfirstlane(nt * 256); };
    auto cur_adv = [&](Cur& c) {
        if (++c.t < S) return;
        c.t = 0;
        const int nq = next_sb(c.sb);
        if (nq < sb_end) { c.sb = nq; cur_set(c); }
    };
    auto stageA = [&](const Cur& c, int h, int buf, int pieces) {
        const v4i rs = make_rsrc((const char*)a.X + ((size_t)(c.m0 + 128 * h) * a.K + (size_t)c.t * 64) * 2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF_BYTES + h * HALF_BYTES) + dstw);
        if (pieces & 1) lds_dma_b128(rs, dst, voA[0], 0);
        if (pieces & 2) lds_dma_b128(rs, dst + 1024, voA[1], 0);
    };
    auto stageB = [&](const Cur& c, int h, int buf, int pieces) {
        const v4i rs = make_rsrc((const char*)a.W + ((size_t)(c.n0 + 128 * h) * a.K + (size_t)c.t * 64) * 2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF_BYTES + (2 + h) * HALF_BYTES) + dstw);
        if (pieces & 1) lds_dma_b128(rs, dst, voB[0], 0);
        if (pieces & 2) lds_dma_b128(rs, dst + 1024, voB[1], 0);
    };
    // parameters of the wave's 64 columns: lane k < 16 of DMA `arr` fetches columns 128 (k >> 3) + 32 wc + 4 (k & 7) .. + 3 -> floats [arr][k][4] of the area
    const unsigned par0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(2 * BUF_BYTES + wid * 2048));
    const unsigned voP = (unsigned)(128 * ((lane & 15) >> 3) + 32 * wc + 4 * (lane & 7)) * 4u;
    auto stageP = [&](int n0, int par) {
        if (lane < 16) {
#pragma unroll
            for (int arr = 0; arr < 3; ++arr) {
                const v4i rs = make_rsrc((const char*)(a.P + (size_t)arr * a.N + n0));
                lds_dma_b128(rs, par0 + (unsigned)(par * 1024 + arr * 256), voP, 0);
            }
        }
    };

    const int sw = (l15 >> 1) & 7;
    const int c0 = (l4 ^ sw) * 16, c1 = ((4 + l4) ^ sw) * 16;
    const char* const Afr = lds + (64 * g + l15) * 128;
    const char* const Bfr = lds + 2 * HALF_BYTES + (32 * wc + l15) * 128;
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
            fa[i][0] = *(const float4*)(Afr + buf * BUF_BYTES + mh * HALF_BYTES + i * 2048 + c0);
            fa[i][1] = *(const float4*)(Afr + buf * BUF_BYTES + mh * HALF_BYTES + i * 2048 + c1);
        }
    };
    auto readB = [&](int buf, int nh, float4 (&fb)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[j][0] = *(const float4*)(Bfr + buf * BUF_BYTES + nh * HALF_BYTES + j * 2048 + c0);
            fb[j][1] = *(const float4*)(Bfr + buf * BUF_BYTES + nh * HALF_BYTES + j * 2048 + c1);
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
    // one epilogue chunk: rows 128 mh + 64 g + 16 i + l15, columns 128 nh + 32 wc + 8 l4 .. + 7 of the tile behind `rY`
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const unsigned nrow16 = __builtin_amdgcn_readfirstlane((unsigned)a.N * 2u * 16u);
    const unsigned voY = (unsigned)(64 * g + l15) * (unsigned)a.N * 2u + (unsigned)(32 * wc + 8 * l4) * 2u;
    auto chunk = [&](int mh, int nh, int i, int par, __amdgpu_buffer_rsrc_t rY) {
        const float* const pw = (const float*)(lds + 2 * BUF_BYTES + wid * 2048 + par * 1024) + (nh * 8 + 2 * l4) * 4;
        half8 hv;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {          // four columns at a time; the fences keep one half's twelve parameters live, not the next chunks' as well (MFMAs and VALU still move across them)
            asm volatile("" ::: "memory");
            const float4 b = *(const float4*)(pw + 4 * hf), sc = *(const float4*)(pw + 64 + 4 * hf), sh = *(const float4*)(pw + 128 + 4 * hf);
            const float bb[4] = {b.x, b.y, b.z, b.w}, ss[4] = {sc.x, sc.y, sc.z, sc.w}, hh[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[mh][nh][i][hf][e] + bb[e];
                acc[mh][nh][i][hf][e] = 0.0f;
                v = fmaxf(v, 0.0f);
                hv[4 * hf + e] = (_Float16)(v * ss[e] + hh[e]);
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, hv), rY, voY, (unsigned)(8 * mh + i) * nrow16 + 256u * nh, 0);
    };
    auto make_rY = [&](int m0, int n0) { return __builtin_amdgcn_make_buffer_rsrc((void*)(a.Y + (size_t)m0 * a.N + n0), 0, 0x7fffffff, 0x00020000); };

#define VMN() do { if (NV == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (NV == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } while (0)
#define MSEG_BEGIN() do { SEG_BARRIER(); LGKM0(); __builtin_amdgcn_s_setprio(1); } while (0)
#define MSEG_END() do { __builtin_amdgcn_s_setprio(0); SEG_BARRIER(); } while (0)

    Cur cc; cc.sb = q0; cc.t = 0; cur_set(cc);
    Cur c1_ = cc, c2_ = cc;
    stageA(cc, 0, 0, 3); stageB(cc, 0, 0, 3); stageB(cc, 1, 0, 3); stageA(cc, 1, 0, 3);
    cur_adv(c1_); c2_ = c1_;
    stageA(c1_, 0, 1, 3); stageB(c1_, 0, 1, 3); stageB(c1_, 1, 1, 3);
    cur_adv(c2_);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int q = q0, t = 0, buf = 0, par = 0;
    bool have_prev = false;
    __amdgpu_buffer_rsrc_t rYc = make_rY(cc.m0, cc.n0), rYp = rYc;
    int n0c = cc.n0;
    unsigned long long t_start = 0;
    if (a.clk) t_start = __builtin_amdgcn_s_memtime();

    // One barrier per phase.  Between two barriers group 1 runs [fragment reads + staging | MFMAs] and group 0 [MFMAs | reads for ITS next phase + staging]:
    // the matrix pipe passes from group 0 to group 1 in the middle of the interval without a barrier (a wave that comes early just shares the pipe).
    // Interval (T, q) stages: q = 0 A half 1 of K-tile T + 1, q = 1 A half 0 of T + 2, q = 2 B half 0 of T + 2, q = 3 B half 1 of T + 2; a half-tile staged in
    // interval s is waited for (vmcnt(NV)) in interval s + NV / 2 - 1 by both groups and read behind the barrier that follows.
#define STAGE_Q(Q) do { \
        if (Q == 0) { stageA(c1_, 1, buf ^ 1, 3); cur_adv(c1_); } \
        else if (Q == 1) stageA(c2_, 0, buf, 3); \
        else if (Q == 2) stageB(c2_, 0, buf, 3); \
        else { if (t == 0) stageP(n0c, par); stageB(c2_, 1, buf, 3); cur_adv(c2_); } \
        VMN(); } while (0)
#define READS_Q(Q, RB) do { \
        if (Q == 0) { readB(RB, 0, fb0); __builtin_amdgcn_sched_barrier(0); readA(RB, 0); } \
        else if (Q == 1) readB(RB, 1, fb1); \
        else if (Q == 2) readA(RB, 1); \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#define MMA_Q(Q) do { \
        __builtin_amdgcn_s_setprio(1); \
        if (Q == 0) { mma(0, 0, fb0); if (carry) { chunk(1, 1, 0, par ^ 1, rYp); chunk(1, 1, 1, par ^ 1, rYp); chunk(1, 1, 2, par ^ 1, rYp); } } \
        else if (Q == 1) { mma(0, 1, fb1); if (last) { chunk(0, 0, 0, par, rYc); chunk(0, 0, 1, par, rYc); } if (carry) { chunk(1, 1, 3, par ^ 1, rYp); chunk(1, 0, 0, par ^ 1, rYp); chunk(1, 0, 1, par ^ 1, rYp); } } \
        else if (Q == 2) { mma(1, 1, fb1); if (last) { chunk(0, 0, 2, par, rYc); chunk(0, 0, 3, par, rYc); chunk(0, 1, 0, par, rYc); } if (carry) { chunk(1, 0, 2, par ^ 1, rYp); chunk(1, 0, 3, par ^ 1, rYp); } } \
        else { mma(1, 0, fb0); if (last) { chunk(0, 1, 1, par, rYc); chunk(0, 1, 2, par, rYc); chunk(0, 1, 3, par, rYc); } } \
        __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); } while (0)
    if (g == 0) { READS_Q(0, 0); }                     // group 0 enters the loop with the fragments of its first phase requested
    while (true) {
        const bool last = t == S - 1, carry = t == 0 && have_prev;
        if (g == 0 && (VAR & 64)) {     // group 0: its LDS-DMAs behind the MFMAs, the reads last
            LGKM0(); MMA_Q(0); STAGE_Q(0); READS_Q(1, buf); SEG_BARRIER();
            LGKM0(); MMA_Q(1); STAGE_Q(1); READS_Q(2, buf); SEG_BARRIER();
            LGKM0(); MMA_Q(2); STAGE_Q(2); SEG_BARRIER();
            MMA_Q(3); STAGE_Q(3); READS_Q(0, buf ^ 1); SEG_BARRIER();
        } else if (g == 0) {
            LGKM0(); MMA_Q(0); READS_Q(1, buf); STAGE_Q(0); SEG_BARRIER();
            LGKM0(); MMA_Q(1); READS_Q(2, buf); STAGE_Q(1); SEG_BARRIER();
            LGKM0(); MMA_Q(2); STAGE_Q(2); SEG_BARRIER();
            MMA_Q(3); READS_Q(0, buf ^ 1); STAGE_Q(3); SEG_BARRIER();
        } else if (VAR & 32) {          // group 1: the LDS-DMAs before the fragment reads
            STAGE_Q(0); READS_Q(0, buf); LGKM0(); MMA_Q(0); SEG_BARRIER();
            STAGE_Q(1); READS_Q(1, buf); LGKM0(); MMA_Q(1); SEG_BARRIER();
            STAGE_Q(2); READS_Q(2, buf); LGKM0(); MMA_Q(2); SEG_BARRIER();
            STAGE_Q(3); MMA_Q(3); SEG_BARRIER();
        } else {
            READS_Q(0, buf); STAGE_Q(0); LGKM0(); MMA_Q(0); SEG_BARRIER();
            READS_Q(1, buf); STAGE_Q(1); LGKM0(); MMA_Q(1); SEG_BARRIER();
            READS_Q(2, buf); STAGE_Q(2); LGKM0(); MMA_Q(2); SEG_BARRIER();
            STAGE_Q(3); MMA_Q(3); SEG_BARRIER();
        }
        buf ^= 1;
        if (last) {
            q = next_sb(q);
            rYp = rYc; have_prev = true; par ^= 1;
            if (q >= sb_end) break;
            { int j_, nt_; (void)sb_valid(q, j_, nt_); n0c = __builtin_amdgcn_readfirstlane(nt_ * 256); rYc = make_rY(__builtin_amdgcn_readfirstlane((xcd + 8 * j_) * 256), n0c); }
            t = 0;
        } else ++t;
    }
    // the last tile's quadrants (1,1) and (1,0)
#pragma unroll
    for (int i = 0; i < 4; ++i) chunk(1, 1, i, par ^ 1, rYp);
#pragma unroll
    for (int i = 0; i < 4; ++i) chunk(1, 0, i, par ^ 1, rYp);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.clk && tid == 0) a.clk[blockIdx.x] = __builtin_amdgcn_s_memtime() - t_start;
}



// ---- k_w4: 4 waves (one per SIMD) x 128 x 128 outputs, accumulators 256 registers per lane; no ping-pong: a wave interleaves its own fragment reads and
// LDS-DMAs with its MFMAs, two barriers per K-tile.  Buffer b = T & 1 holds K-tile T ([A 256 rows | B 256 rows] x 128 B).  Per K-tile:
//   phase X: 64 MFMAs on k-group 0 | reads of k-group 1 (buffer b) | the 8 B pieces of K-tile T + 1 -> buffer b ^ 1 (groups 0 - 3, two each)
//   lgkmcnt(0), barrier B1: every wave has read buffer b out
//   phase Y: 64 MFMAs on k-group 1 | the 8 A pieces of K-tile T + 2 -> buffer b (groups 0 - 3) | vmcnt(8), barrier B2 behind group 5 | reads of k-group 0 of T + 1 (groups 6, 7)
template <int VAR>
__global__ __launch_bounds__(256) void k_w4(Args a)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int w = blockIdx.x, G = gridDim.x;
    const int xcd = w & 7, wl = w >> 3, wpx = G >> 3;
    const int mx = (a.m_tiles - xcd + 7) >> 3;
    const int PN = a.n_tiles < a.pnmax ? a.n_tiles : a.pnmax;
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
    const int wr = wid >> 1, wc = wid & 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int S = a.K / 64;
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)lds;

    // loader: wave `wid` stages LDS rows 64 wid + 8 p + (lane >> 3), p = 0 .. 7, of the A region and of the B region
    const int prow = lane >> 3;
    const unsigned ch0 = (unsigned)(((lane & 7) ^ ((prow >> 1) & 7)) * 16), ch1 = (unsigned)(((lane & 7) ^ (((8 + prow) >> 1) & 7)) * 16);      // chunk by row parity class: p even / odd
    // A: row r of the tile;  B: LDS row 32 grp + 16 jj + rho holds channel 32 grp + 8 (rho >> 2) + 4 jj + (rho & 3)
    unsigned voA[2], voB[2][2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        voA[pp] = (unsigned)(64 * wid + 8 * pp + prow) * (unsigned)a.K * 2u + (pp ? ch1 : ch0);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int rho = 8 * pp + prow;
            voB[jj][pp] = (unsigned)(64 * wid + 8 * (rho >> 2) + 4 * jj + (rho & 3)) * (unsigned)a.K * 2u + (pp ? ch1 : ch0);      // + 32 grp rows by the piece's scalar offset
        }
    }
    struct Cur { int sb; int t; int m0, n0; };
    auto cur_set = [&](Cur& c) { int j, nt; (void)sb_valid(c.sb, j, nt); c.m0 = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * 256); c.n0 = __builtin_amdgcn_readfirstlane(nt * 256); };
    auto cur_adv = [&](Cur& c) {
        if (++c.t < S) return;
        c.t = 0;
        const int nq = next_sb(c.sb);
        if (nq < sb_end) { c.sb = nq; cur_set(c); }
    };
    // piece p (0 .. 7) of the wave's 64 A rows: rows 8 p .. 8 p + 7
    auto dmaA = [&](const Cur& c, int buf, int p) {
        const v4i rs = make_rsrc((const char*)a.X + ((size_t)(c.m0 + 16 * (p >> 1)) * a.K + (size_t)c.t * 64) * 2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF_BYTES + (64 * wid + 8 * p) * 128));
        lds_dma_b128(rs, dst, voA[p & 1], 0);
    };
    // piece p of the wave's 64 B rows: LDS rows 8 p .. + 7 of the wave's share = group grp = p >> 2 (of the wave's two 32-row groups), jj = (p >> 1) & 1, half pp = p & 1
    auto dmaB = [&](const Cur& c, int buf, int p) {
        const v4i rs = make_rsrc((const char*)a.W + ((size_t)(c.n0 + 32 * (p >> 2)) * a.K + (size_t)c.t * 64) * 2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF_BYTES + 32768 + (64 * wid + 8 * p) * 128));
        lds_dma_b128(rs, dst, voB[(p >> 1) & 1][p & 1], 0);
    };

    const int sw = (l15 >> 1) & 7;
    const int c0 = (l4 ^ sw) * 16, c1 = ((4 + l4) ^ sw) * 16;
    const char* const Afr = lds + (128 * wr + l15) * 128;                 // + i * 2048 + c{ks}
    const char* const Bfr = lds + 32768 + (128 * wc + l15) * 128;         // + j * 2048 + c{ks}
    f32x4 fa[2][8], fb[2][8];
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto rdA = [&](int buf, int ks, int i) { fa[ks][i] = *(const f32x4*)(Afr + buf * BUF_BYTES + i * 2048 + (ks ? c1 : c0)); };
    auto rdB = [&](int buf, int ks, int j) { fb[ks][j] = *(const f32x4*)(Bfr + buf * BUF_BYTES + j * 2048 + (ks ? c1 : c0)); };
    // hipcc (ROCm 7.2) does not turn this into a spill-free kernel: with the builtin it keeps part of the 256 accumulator registers in VGPRs and some
    // fragments in AGPRs and spills 84 - 166 registers into the loop; with the MFMA written as inline assembly and its accumulator pinned to the AGPR
    // file ("+a") it spills 194 (its VGPR -> AGPR spill slots collide with the pinned accumulators).  The tiling needs hand-written assembly; the
    // kernel stays here as the record of the attempt (VERDICT r05 #1 asked for it) -- correct, slow.
    auto mma_row = [&](int ks, int i) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fb[ks][j]), __builtin_bit_cast(half8, fa[ks][i]), acc[i][j], 0, 0, 0);
    };
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)

    Cur cc; cc.sb = q0; cc.t = 0; cur_set(cc);
    Cur c1_ = cc, c2_ = cc;
    // prologue: K-tile 0 whole -> buffer 0, A pieces of K-tile 1 -> buffer 1
#pragma unroll
    for (int p = 0; p < 8; ++p) { dmaA(cc, 0, p); dmaB(cc, 0, p); }
    cur_adv(c1_); c2_ = c1_;
#pragma unroll
    for (int p = 0; p < 8; ++p) dmaA(c1_, 1, p);
    cur_adv(c2_);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) { rdA(0, 0, i); rdB(0, 0, i); }

    int q = q0, t = 0, buf = 0;
    int m0c = cc.m0, n0c = cc.n0;
    unsigned long long t_start = 0;
    if (a.clk) t_start = __builtin_amdgcn_s_memtime();
    while (true) {
        // ---- phase X
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); W4_FENCE();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            rdA(buf, 1, i); rdB(buf, 1, i);
            if (i < 4) { dmaB(c1_, buf ^ 1, 2 * i); dmaB(c1_, buf ^ 1, 2 * i + 1); }
            mma_row(0, i);
            W4_FENCE();
        }
        cur_adv(c1_);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W4_FENCE(); __builtin_amdgcn_s_barrier(); W4_FENCE();
        // ---- phase Y
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < 4) { dmaA(c2_, buf, 2 * i); dmaA(c2_, buf, 2 * i + 1); }
            if (i == 6) {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                W4_FENCE(); __builtin_amdgcn_s_barrier(); W4_FENCE();
            }
            if (i >= 6) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { rdA(buf ^ 1, 0, 4 * (i - 6) + r); rdB(buf ^ 1, 0, 4 * (i - 6) + r); }
            }
            mma_row(1, i);
            W4_FENCE();
        }
        cur_adv(c2_);
        buf ^= 1;
        if (t == S - 1) {
            // lab epilogue: all four waves at once, plain conversion (the product form would carry it in chunks as k_pp2 does)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    const int row = m0c + 128 * wr + 16 * i + l15;
                    const int col = n0c + 128 * wc + 32 * grp + 8 * l4;
                    half8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { hv[e] = (_Float16)acc[i][2 * grp + (e >> 2)][e & 3]; acc[i][2 * grp + (e >> 2)][e & 3] = 0.0f; }
                    *(half8*)(a.Y + (size_t)row * a.N + col) = hv;
                }
            q = next_sb(q);
            if (q >= sb_end) break;
            { int j_, nt_; (void)sb_valid(q, j_, nt_); m0c = __builtin_amdgcn_readfirstlane((xcd + 8 * j_) * 256); n0c = __builtin_amdgcn_readfirstlane(nt_ * 256); }
            t = 0;
        } else ++t;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.clk && tid == 0) a.clk[blockIdx.x] = __builtin_amdgcn_s_memtime() - t_start;
}


// ---- k_w8: 8 waves, NO ping-pong -- every wave interleaves its own fragment reads and LDS-DMAs with its MFMAs (the two waves of a SIMD cover each other's
// DMA-issue stalls), two barriers per K-tile, DMAs evenly spread and never drained.  Same wave tile and LDS image as k_pp2 (A half 0 / 1, B half 0 / 1 per
// buffer), MFMAs ks-major:
//   phase X: 32 MFMAs on k-group 0 (4 sub-groups of 8 = two row blocks x four column blocks) | the 12 fragment reads of k-group 1 | the 4 B pieces of K-tile T + 1 -> buffer b ^ 1
//   lgkmcnt(0), barrier B1: buffer b is read out
//   phase Y: 32 MFMAs on k-group 1 | the 4 A pieces of K-tile T + 2 -> buffer b | behind sub-group 2: vmcnt(3), barrier B2, the 12 reads of k-group 0 of T + 1
// The epilogue's 16 chunks run in the next tile's first phase X, those of a row block just before that block's first MFMAs.
template <int VAR>
__global__ __launch_bounds__(512) void k_w8(Args a)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int w = blockIdx.x, G = gridDim.x;
    const int xcd = w & 7, wl = w >> 3, wpx = G >> 3;
    const int mx = (a.m_tiles - xcd + 7) >> 3;
    const int PN = a.n_tiles < a.pnmax ? a.n_tiles : a.pnmax;
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
    const int S = a.K / 64;
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)lds;

    unsigned voA[2], voB[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = wid * 16 + p * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((r >> 1) & 7);
        voA[p] = (unsigned)r * (unsigned)a.K * 2u + (unsigned)ch * 16u;
        const int rho = r & 15;
        const int col = 32 * (wid >> 1) + 8 * (rho >> 2) + 4 * (wid & 1) + (rho & 3);
        voB[p] = (unsigned)col * (unsigned)a.K * 2u + (unsigned)ch * 16u;
    }
    const unsigned dstw = __builtin_amdgcn_readfirstlane((unsigned)(wid * 2048));
    struct Cur { int sb; int t; int m0, n0; };
    auto cur_set = [&](Cur& c) { int j, nt; (void)sb_valid(c.sb, j, nt); c.m0 = __builtin_amdgcn_readfirstlane((xcd + 8 * j) * 256); c.n0 = __builtin_amdgcn_readfirstlane(nt * 256); };
    auto cur_adv = [&](Cur& c) {
        if (++c.t < S) return;
        c.t = 0;
        const int nq = next_sb(c.sb);
        if (nq < sb_end) { c.sb = nq; cur_set(c); }
    };
    // piece q = 0 .. 3 of the wave's share of A (B): half q >> 1, piece q & 1
    auto dmaA = [&](const Cur& c, int buf, int q) {
        const int h = q >> 1;
        const v4i rs = make_rsrc((const char*)a.X + ((size_t)(c.m0 + 128 * h) * a.K + (size_t)c.t * 64) * 2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF_BYTES + h * HALF_BYTES + (q & 1) * 1024) + dstw);
        lds_dma_b128(rs, dst, voA[q & 1], 0);
    };
    auto dmaB = [&](const Cur& c, int buf, int q) {
        const int h = q >> 1;
        const v4i rs = make_rsrc((const char*)a.W + ((size_t)(c.n0 + 128 * h) * a.K + (size_t)c.t * 64) * 2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF_BYTES + (2 + h) * HALF_BYTES + (q & 1) * 1024) + dstw);
        lds_dma_b128(rs, dst, voB[q & 1], 0);
    };
    const unsigned par0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(2 * BUF_BYTES + wid * 2048));
    const unsigned voP = (unsigned)(128 * ((lane & 15) >> 3) + 32 * wc + 4 * (lane & 7)) * 4u;
    auto stageP = [&](int n0, int par) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(par0 + (unsigned)(par * 1024));
        const v4i r0 = make_rsrc((const char*)(a.P + n0)), r1 = make_rsrc((const char*)(a.P + (size_t)a.N + n0)), r2 = make_rsrc((const char*)(a.P + (size_t)2 * a.N + n0));
        if (lane < 16) { lds_dma_b128(r0, dst, voP, 0); lds_dma_b128(r1, dst + 256, voP, 0); lds_dma_b128(r2, dst + 512, voP, 0); }
    };

    const int sw = (l15 >> 1) & 7;
    const int c0 = (l4 ^ sw) * 16, c1 = ((4 + l4) ^ sw) * 16;
    const char* const Afr = lds + (64 * g + l15) * 128;                      // + mh * HALF_BYTES + i * 2048 + c{ks}
    const char* const Bfr = lds + 2 * HALF_BYTES + (32 * wc + l15) * 128;    // + nh * HALF_BYTES + j * 2048 + c{ks}
    float4 fa[2][8], fb[2][4];          // [ks][row block R = 4 mh + i], [ks][column block C = 2 nh + j]
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto rdA = [&](int buf, int ks, int R) { fa[ks][R] = *(const float4*)(Afr + buf * BUF_BYTES + (R >> 2) * HALF_BYTES + (R & 3) * 2048 + (ks ? c1 : c0)); };
    auto rdB = [&](int buf, int ks, int C) { fb[ks][C] = *(const float4*)(Bfr + buf * BUF_BYTES + (C >> 1) * HALF_BYTES + (C & 1) * 2048 + (ks ? c1 : c0)); };
    auto rd_set = [&](int buf, int ks, int part) {         // the 12 reads of a k-group in four parts of three
        if (part == 0) { rdB(buf, ks, 0); rdB(buf, ks, 1); rdB(buf, ks, 2); }
        else if (part == 1) { rdB(buf, ks, 3); rdA(buf, ks, 0); rdA(buf, ks, 1); }
        else if (part == 2) { rdA(buf, ks, 2); rdA(buf, ks, 3); rdA(buf, ks, 4); }
        else { rdA(buf, ks, 5); rdA(buf, ks, 6); rdA(buf, ks, 7); }
    };
    auto mma_sg = [&](int ks, int s) {                     // sub-group s: row blocks 2 s, 2 s + 1
#pragma unroll
        for (int C = 0; C < 4; ++C)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
                acc[2 * s + rr][C] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fb[ks][C]), __builtin_bit_cast(half8, fa[ks][2 * s + rr]), acc[2 * s + rr][C], 0, 0, 0);
    };
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const unsigned nrow16 = __builtin_amdgcn_readfirstlane((unsigned)a.N * 2u * 16u);
    const unsigned voY = (unsigned)(64 * g + l15) * (unsigned)a.N * 2u + (unsigned)(32 * wc + 8 * l4) * 2u;
    // chunk of row block R, column pair nh: rows 128 (R >> 2) + 64 g + 16 (R & 3) + l15, columns 128 nh + 32 wc + 8 l4 .. + 7
    auto chunk = [&](int R, int nh, int par, __amdgpu_buffer_rsrc_t rY) {
        const float* const pw = (const float*)(lds + 2 * BUF_BYTES + wid * 2048 + par * 1024) + (nh * 8 + 2 * l4) * 4;
        half8 hv;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            asm volatile("" ::: "memory");
            const float4 b = *(const float4*)(pw + 4 * hf), sc = *(const float4*)(pw + 64 + 4 * hf), sh = *(const float4*)(pw + 128 + 4 * hf);
            const float bb[4] = {b.x, b.y, b.z, b.w}, ss[4] = {sc.x, sc.y, sc.z, sc.w}, hh[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[R][2 * nh + hf][e] + bb[e];
                acc[R][2 * nh + hf][e] = 0.0f;
                v = fmaxf(v, 0.0f);
                hv[4 * hf + e] = (_Float16)(v * ss[e] + hh[e]);
            }
        }
        asm volatile("" ::: "memory");
        if (VAR & 4) { asm volatile("" :: "v"(hv)); return; }      // diagnostic: the epilogue without its stores
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, hv), rY, voY, (unsigned)(8 * (R >> 2) + (R & 3)) * nrow16 + 256u * nh, 0);
    };
    // VAR 16: chunk with whole-tuple reads and zeroing (can it share a basic block with MFMAs?)
    auto chunk3 = [&](int R, int nh, int par, __amdgpu_buffer_rsrc_t rY) {
        const float* const pw = (const float*)(lds + 2 * BUF_BYTES + wid * 2048 + par * 1024) + (nh * 8 + 2 * l4) * 4;
        const f32x4 x0 = acc[R][2 * nh], x1 = acc[R][2 * nh + 1];
        {   // zeros the compiler cannot fold into the next MFMA's C operand (an MFMA on an inline-constant C is not tied to its accumulator registers any more: 217 spills)
            f32x4 z0, z1;
            asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z0[0]), "=v"(z0[1]), "=v"(z0[2]), "=v"(z0[3]));
            asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z1[0]), "=v"(z1[1]), "=v"(z1[2]), "=v"(z1[3]));
            acc[R][2 * nh] = z0; acc[R][2 * nh + 1] = z1;
        }
        const float4 b0 = *(const float4*)(pw), b1 = *(const float4*)(pw + 4), s0 = *(const float4*)(pw + 64), s1 = *(const float4*)(pw + 68), h0 = *(const float4*)(pw + 128), h1 = *(const float4*)(pw + 132);
        half8 hv;
        hv[0] = (_Float16)(fmaxf(x0[0] + b0.x, 0.f) * s0.x + h0.x); hv[1] = (_Float16)(fmaxf(x0[1] + b0.y, 0.f) * s0.y + h0.y);
        hv[2] = (_Float16)(fmaxf(x0[2] + b0.z, 0.f) * s0.z + h0.z); hv[3] = (_Float16)(fmaxf(x0[3] + b0.w, 0.f) * s0.w + h0.w);
        hv[4] = (_Float16)(fmaxf(x1[0] + b1.x, 0.f) * s1.x + h1.x); hv[5] = (_Float16)(fmaxf(x1[1] + b1.y, 0.f) * s1.y + h1.y);
        hv[6] = (_Float16)(fmaxf(x1[2] + b1.z, 0.f) * s1.z + h1.z); hv[7] = (_Float16)(fmaxf(x1[3] + b1.w, 0.f) * s1.w + h1.w);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, hv), rY, voY, (unsigned)(8 * (R >> 2) + (R & 3)) * nrow16 + 256u * nh, 0);
    };
    // VAR 1: the same epilogue with the 24 parameters of a column pair read ONCE for the wave's eight row blocks, no fences (the chunk form reads them per chunk,
    // 96 ds_read_b128 per tile, and waits for them 32 times)
    auto epilogue2 = [&](int par, __amdgpu_buffer_rsrc_t rY) {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            const float* const pw = (const float*)(lds + 2 * BUF_BYTES + wid * 2048 + par * 1024) + (nh * 8 + 2 * l4) * 4;
            const float4 b0 = *(const float4*)(pw), b1 = *(const float4*)(pw + 4), s0 = *(const float4*)(pw + 64), s1 = *(const float4*)(pw + 68), h0 = *(const float4*)(pw + 128), h1 = *(const float4*)(pw + 132);
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, hh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int R = 0; R < 8; ++R) {
                half8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = acc[R][2 * nh + (e >> 2)][e & 3] + bb[e];
                    acc[R][2 * nh + (e >> 2)][e & 3] = 0.0f;
                    v = fmaxf(v, 0.0f);
                    hv[e] = (_Float16)(v * ss[e] + hh[e]);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, hv), rY, voY, (unsigned)(8 * (R >> 2) + (R & 3)) * nrow16 + 256u * nh, 0);
            }
        }
    };
    auto make_rY = [&](int m0, int n0) { return __builtin_amdgcn_make_buffer_rsrc((void*)(a.Y + (size_t)m0 * a.N + n0), 0, 0x7fffffff, 0x00020000); };
#define W8_FENCE() __builtin_amdgcn_sched_barrier(0)

    if (VAR & 2) {        // the workgroups start in 16 phases a.skew cycles apart: their epilogues (128 KB of stores each) no longer fall together
        const unsigned long long ts = __builtin_amdgcn_s_memtime(), wait = (unsigned long long)((wl * 5) % 16) * (unsigned long long)a.skew;
        while (__builtin_amdgcn_s_memtime() - ts < wait) __builtin_amdgcn_s_sleep(32);
    }
    Cur cc; cc.sb = q0; cc.t = 0; cur_set(cc);
    Cur c1_ = cc, c2_ = cc;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) { dmaA(cc, 0, qq); dmaB(cc, 0, qq); }
    cur_adv(c1_); c2_ = c1_;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) dmaA(c1_, 1, qq);
    cur_adv(c2_);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int part = 0; part < 4; ++part) rd_set(0, 0, part);

    int q = q0, t = 0, buf = 0, par = 0;
    bool have_prev = false;
    __amdgpu_buffer_rsrc_t rYc = make_rY(cc.m0, cc.n0), rYp = rYc;
    int n0c = cc.n0;
    unsigned long long t_start = 0;
    if (a.clk) t_start = __builtin_amdgcn_s_memtime();
    while (true) {
        const bool carry = t == 0 && have_prev;
        // ---- phase X
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); W8_FENCE();
        if (VAR & 32) {
            // ONE copy of phase X; the chunks of a sub-group's row blocks under a branch of their own in front of it
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (carry) { chunk(2 * s, 0, par ^ 1, rYp); chunk(2 * s, 1, par ^ 1, rYp); chunk(2 * s + 1, 0, par ^ 1, rYp); chunk(2 * s + 1, 1, par ^ 1, rYp); }
                W8_FENCE();
                rd_set(buf, 1, s);
                mma_sg(0, s);
                W8_FENCE();
                dmaB(c1_, buf ^ 1, s);
                W8_FENCE();
            }
            goto phase_x_done;
        }
        if (carry && (VAR & 16)) {
            // chunks of a row block in ONE block with the MFMAs, just in front of that block's first products (whole-tuple form)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                chunk3(2 * s, 0, par ^ 1, rYp); chunk3(2 * s, 1, par ^ 1, rYp); chunk3(2 * s + 1, 0, par ^ 1, rYp); chunk3(2 * s + 1, 1, par ^ 1, rYp);
                rd_set(buf, 1, s);
                mma_sg(0, s);
                W8_FENCE();
                dmaB(c1_, buf ^ 1, s);
                W8_FENCE();
            }
            goto phase_x_done;
        }
        if (carry) {
            // the first K-tile of a tile: the previous tile's 16 chunks in a block of their own (in one block with the MFMAs hipcc spills 217 registers)
            if (VAR & 1) epilogue2(par ^ 1, rYp);
            else {
#pragma unroll
                for (int R = 0; R < 8; ++R) { chunk(R, 0, par ^ 1, rYp); chunk(R, 1, par ^ 1, rYp); }
            }
        }
        W8_FENCE();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            rd_set(buf, 1, s);
            mma_sg(0, s);
            W8_FENCE();
            dmaB(c1_, buf ^ 1, s);
            W8_FENCE();
        }
    phase_x_done:
        cur_adv(c1_);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W8_FENCE(); __builtin_amdgcn_s_barrier(); W8_FENCE();
        // ---- phase Y
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s == 3) {
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                W8_FENCE(); __builtin_amdgcn_s_barrier(); W8_FENCE();
                if (t == 0) stageP(n0c, par);
                rd_set(buf ^ 1, 0, 0); rd_set(buf ^ 1, 0, 1); rd_set(buf ^ 1, 0, 2); rd_set(buf ^ 1, 0, 3);
            }
            mma_sg(1, s);
            W8_FENCE();
            dmaA(c2_, buf, s);
            W8_FENCE();
        }
        cur_adv(c2_);
        buf ^= 1;
        if (t == S - 1) {
            q = next_sb(q);
            rYp = rYc; have_prev = true; par ^= 1;
            if (q >= sb_end) break;
            { int j_, nt_; (void)sb_valid(q, j_, nt_); n0c = __builtin_amdgcn_readfirstlane(nt_ * 256); rYc = make_rY(__builtin_amdgcn_readfirstlane((xcd + 8 * j_) * 256), n0c); }
            t = 0;
        } else ++t;
    }
    if (VAR & 1) epilogue2(par ^ 1, rYp);
    else {
#pragma unroll
        for (int R = 0; R < 8; ++R) { chunk(R, 0, par ^ 1, rYp); chunk(R, 1, par ^ 1, rYp); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.clk && tid == 0) a.clk[blockIdx.x] = __builtin_amdgcn_s_memtime() - t_start;
}

// ---- reference: selected rows, f32 accumulation in k order (any order is within the check's tolerance)
__global__ void k_ref_rows(const _Float16* X, const _Float16* W, const int* rows, float* out, int N, int K, const float* P)
{
    const int r = rows[blockIdx.x];
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += (float)X[(size_t)r * K + k] * (float)W[(size_t)n * K + k];
        out[(size_t)blockIdx.x * N + n] = s;
        out[(size_t)(gridDim.x + blockIdx.x) * N + n] = fmaxf(s + P[n], 0.0f) * P[N + n] + P[2 * N + n];      // the k_pp2 epilogue
    }
}
__global__ void k_fill_f(float* p, size_t n, unsigned seed, float scale, float off)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = ((float)(x & 0xffffff) / 8388608.0f - 1.0f) * scale + off;
}
__global__ void k_fill(_Float16* p, size_t n, unsigned seed, float scale)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed + (unsigned)(i >> 32) * 40503u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = (_Float16)(((float)(x & 0xffffff) / 8388608.0f - 1.0f) * scale);
}
__global__ void k_gather_rows(const _Float16* Y, const int* rows, float* out, int N)
{
    const int r = rows[blockIdx.x];
    for (int n = threadIdx.x; n < N; n += blockDim.x) out[(size_t)blockIdx.x * N + n] = (float)Y[(size_t)r * N + n];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 860160;
    const int M = (rows / 256) * 256;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, rows %d\n", prop.name, cus, M);
    const int shapes[2][2] = {{1024, 1024}, {3072, 3072}};
    typedef void (*kern_t)(Args);
    struct V { const char* name; kern_t k; int check; int pn; int gdiv = 1; int skew = 0; };
    auto threads_of = [&](const V& v) { return strncmp(v.name, "W4", 2) == 0 ? 256 : 512; };          // check: 0 none, 1 plain product, 2 with the k_pp2 epilogue
    const V vars[] = {{"PP2 nv6", k_pp2<6, 0>, 2, 4}, {"PP2 dma first", k_pp2<6, 32>, 2, 4}, {"PP2 dma split", k_pp2<6, 64>, 2, 4}, 
                      {"PP3 nv6", k_pp3<6, 0>, 2, 4}, {"DMA only", k_pp<4 | 16>, 0, 4}, {"reg loads only", k_pp<4 | 16 | 128>, 0, 4},
                      {"W8", k_w8<0>, 2, 4}, {"W8 no stores", k_w8<4>, 0, 4}, {"W8 chunks split", k_w8<32>, 2, 4}};
    const int vmask = argc > 2 ? (int)strtol(argv[2], nullptr, 0) : 0x7fffffff;
    const int rounds = argc > 3 ? atoi(argv[3]) : 3;
    const size_t lds_bytes = 2 * BUF_BYTES + 8 * 2048;
    for (const V& v : vars) CK(hipFuncSetAttribute((const void*)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    for (int si = 0; si < 2; ++si) {
        const int K = shapes[si][0], N = shapes[si][1];
        _Float16 *X, *W, *Y;
        CK(hipMalloc(&X, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&Y, (size_t)M * N * 2));
        hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)M * K + 255) / 256)), dim3(256), 0, 0, X, (size_t)M * K, 1u, 1.0f);
        hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)N * K + 255) / 256)), dim3(256), 0, 0, W, (size_t)N * K, 2u, 0.05f);
        CK(hipMemset(Y, 0, (size_t)M * N * 2));
        unsigned long long* clk; CK(hipMalloc(&clk, 128 * 1024));
        const int NR = 64;
        std::vector<int> hr(NR);
        for (int i = 0; i < NR; ++i) hr[i] = (int)(((long long)i * 2654435761ll) % M);
        hr[0] = 0; hr[1] = M - 1; hr[2] = 255; hr[3] = 256; hr[4] = 127; hr[5] = 128; hr[6] = 64; hr[7] = 63;
        int* dr; float *o_ref, *o_got;
        CK(hipMalloc(&dr, NR * 4)); CK(hipMalloc(&o_ref, (size_t)2 * NR * N * 4)); CK(hipMalloc(&o_got, (size_t)NR * N * 4));
        CK(hipMemcpy(dr, hr.data(), NR * 4, hipMemcpyHostToDevice));
        float* P; CK(hipMalloc(&P, (size_t)3 * N * 4));
        hipLaunchKernelGGL(k_fill_f, dim3((N + 255) / 256), dim3(256), 0, 0, P, (size_t)N, 11u, 0.2f, 0.0f);
        hipLaunchKernelGGL(k_fill_f, dim3((N + 255) / 256), dim3(256), 0, 0, P + N, (size_t)N, 12u, 0.25f, 1.0f);
        hipLaunchKernelGGL(k_fill_f, dim3((N + 255) / 256), dim3(256), 0, 0, P + 2 * N, (size_t)N, 13u, 0.1f, 0.0f);
        hipLaunchKernelGGL(k_ref_rows, dim3(NR), dim3(256), 0, 0, X, W, dr, o_ref, N, K, P);
        std::vector<float> h_ref((size_t)2 * NR * N), h_got((size_t)NR * N);
        CK(hipMemcpy(h_ref.data(), o_ref, h_ref.size() * 4, hipMemcpyDeviceToHost));
        const int grid_full = (cus / 8) * 8;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int round = 0; round < rounds; ++round) {
        int vi = 0;
        for (const V& v : vars) {
            if (!((vmask >> vi++) & 1)) continue;
            Args a; memset(&a, 0, sizeof(a));
            a.X = X; a.W = W; a.Y = Y; a.M = M; a.N = N; a.K = K; a.m_tiles = M / 256; a.n_tiles = N / 256; a.pnmax = v.pn; a.P = P; a.skew = v.skew;
            const int grid = grid_full / v.gdiv;
            CK(hipMemset(clk, 0, 128 * 1024)); CK(hipMemset(Y, 0, (size_t)M * N * 2));
            const int reps = 4;
            for (int r = -1; r < reps; ++r) {
                if (r == 0) CK(hipEventRecord(e0, 0));
                a.clk = (r == reps - 1) ? clk : nullptr;
                hipLaunchKernelGGL(v.k, dim3(grid), dim3(threads_of(v)), lds_bytes, 0, a);
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            std::vector<unsigned long long> hc(grid); CK(hipMemcpy(hc.data(), clk, grid * 8, hipMemcpyDeviceToHost));
            std::sort(hc.begin(), hc.end());
            const double flops = 2.0 * M * (double)N * K;
            const double steps_per_wg = (double)(M / 256) * (N / 256) * (K / 64) / grid;    // (a reduced grid still walks the whole matrix)
            printf("%-18s K %4d N %4d: %8.3f ms  %7.1f TF  (%.3f of 2.5 PF); median workgroup %.0f cycles per K-tile (2048 = the matrix pipe)\n", v.name, K, N, ms, flops / ms / 1e9,
                   flops / ms / 1e9 / 2500.0, (double)hc[grid / 2] / steps_per_wg);
            if (strstr(v.name, "stamps")) {
                std::vector<unsigned long long> hs(16384); CK(hipMemcpy(hs.data(), clk, 128 * 1024, hipMemcpyDeviceToHost));
                const double kt = steps_per_wg;
                for (int gsel = 0; gsel < 2; ++gsel) {
                    printf("    group %d, cycles per K-tile by phase [reads+DMA issue | vmcnt wait | barrier+lgkmcnt | MFMA issue | closing barrier] (mean over workgroups)\n", gsel);
                    for (int ph = 0; ph < 4; ++ph) {
                        printf("      phase %d:", ph);
                        double tot = 0;
                        for (int j = 0; j < 5; ++j) { double sum = 0; for (int wg = 0; wg < grid; ++wg) sum += (double)hs[512 + (wg * 2 + gsel) * 20 + ph * 5 + j]; printf(" %7.0f", sum / grid / kt); tot += sum / grid / kt; }
                        printf("   = %.0f\n", tot);
                    }
                }
            }
            if (v.check && round == 0) {
                const float* ref = h_ref.data() + (v.check == 2 ? (size_t)NR * N : 0);
                hipLaunchKernelGGL(k_gather_rows, dim3(NR), dim3(256), 0, 0, Y, dr, o_got, N);
                CK(hipMemcpy(h_got.data(), o_got, h_got.size() * 4, hipMemcpyDeviceToHost));
                double maxerr = 0, maxref = 0; size_t bad = 0;
                for (size_t i = 0; i < h_got.size(); ++i) {
                    const double e = fabs((double)ref[i] - h_got[i]);
                    maxerr = std::max(maxerr, e); maxref = std::max(maxref, (double)fabs(ref[i]));
                    if (e > 2e-3 * fabs(ref[i]) + 2e-3) ++bad;
                }
                printf("    check: %d rows x %d columns, max |ref| %.3f, max error %.2e, %zu outside tolerance%s\n", NR, N, maxref, maxerr, bad, bad ? "  <-- WRONG" : "");
            }
            fflush(stdout);
        }
        }
        CK(hipFree(P));
        CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(Y)); CK(hipFree(clk)); CK(hipFree(dr)); CK(hipFree(o_ref)); CK(hipFree(o_got));
    }
    return 0;
}
