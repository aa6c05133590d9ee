// linkage_rg.hip -- k_linkage_rg: the cooperative centroid linkage (Clustering::linkage's fast_linkage, cl.cpp:289-406) with the
// per-column state in REGISTERS and everything a merge needs travelling in the slots.
//
// Same algorithm, same matrix layout and same exchange as k_linkage_mw<*, true> (cluster.hip: full N x N distance matrix, a workgroup
// owns a contiguous range of COLUMNS, row y is rewritten and not mirrored, the current copy of an entry lives in the row of the cluster
// that was a merge's survivor last -- index `ty` --, two-level lower bounds, cooperative refresh of stale rows, 8-byte tagged granules),
// so Z is the same bit for bit.  What differs is where a merge round spends its time (profiles/r05_linkage_*.txt):
//   * a thread keeps the bound / second bound / neighbour / freshness / size / ty of its (at most 4) columns in registers; the pass
//     reads no LDS and the candidate it publishes carries the sizes and ty of BOTH clusters of its pair.  The merge that follows
//     therefore starts from what the digest hands over: no per-workgroup size / ty arrays in global memory, no dependent loads
//     (k_linkage_mw's prefetch_pair: the size loads were waited for in front of the row loads -- the loop-invariant part of the
//     Lance-Williams formula is hoisted above the pass) and no global copies of the bounds at all;
//   * the "second pair at the merge height" flag rides in the block reduction, Z is written by its one thread without a barrier, the
//     slot is published by one store instruction (lane w stores word w): one workgroup barrier between the pass and the publish
//     instead of four;
//   * a row without an active column above it gets the exact bound +inf at once (columns are only ever removed) instead of a stale
//     bound that costs a refresh round later.
// Ties: as k_linkage_mw, a merge is taken only while the closest pair is unique; otherwise sync[5] is raised and run_linkage hands the
// job to the kernel that replays the reference's heap.
#include "common.h"
#include "linkage_dev.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_linkage_rg's fence-free slot exchange (sc1 loads / stores, 8-byte tagged granules) is written for gfx950 only"
#endif

#define RG_T_MAX 1024
#define RG_U 4            // columns per thread (register state): a workgroup owns at most 4 * blockDim.x columns
#define RG_KR 4           // stale rows refreshed per retry round
#define RG_GMAX 128       // workgroups
#define RG_SLOT 40        // granules between two slots
#define RG_CW 9           // words of a candidate: bound (2), row, neighbour, flags, size of row / of neighbour, ty of row / of neighbour
#define RG_QW 7           // words of a row partial: minimum (2), its column, second value (2), size and ty of that column's cluster
#define RG_MW 16          // merge round: candidate + NN(y) partial ("row x had a second pair at the merge height" is bit 2 of the candidate's flag word)
#define RG_ROWTIE 4

struct RCand { double v; int i, y, fl, szi, szy, tyi, tyy; };
struct RQ { double v; int i; double v2; int sz, ty; };

__device__ __forceinline__ RCand rc_none() { RCand c; c.v = INFINITY; c.i = -1; c.y = -1; c.fl = 0; c.szi = 0; c.szy = 0; c.tyi = -1; c.tyy = -1; return c; }
__device__ __forceinline__ RQ rq_none() { RQ q; q.v = INFINITY; q.i = -1; q.v2 = INFINITY; q.sz = 0; q.ty = -1; return q; }

// sequential accumulation of one row into a thread's running candidate (rules of cand_acc / cbetter, cluster.hip)
__device__ __forceinline__ void rc_acc(RCand& m, double v, int z, int y, int fl, int szi, int szy, int tyi, int tyy)
{
    if (m.i < 0 || v < m.v) { m.v = v; m.i = z; m.y = y; m.fl = fl; m.szi = szi; m.szy = szy; m.tyi = tyi; m.tyy = tyy; }
    else if (v == m.v) {
        const int tie = (v < INFINITY) ? CAND_TIE : 0;
        if (z < m.i) { m.fl = fl | tie | (m.fl & CAND_TIE); m.i = z; m.y = y; m.szi = szi; m.szy = szy; m.tyi = tyi; m.tyy = tyy; }
        else m.fl |= tie;
    }
}
__device__ __forceinline__ RCand rc_better(RCand a, RCand b)
{
    if (b.i < 0) return a;
    if (a.i < 0) return b;
    if (b.v < a.v) return b;
    if (b.v == a.v) {
        RCand r = (b.i < a.i) ? b : a;
        if (a.i != b.i && a.v < INFINITY) r.fl |= CAND_TIE | ((a.fl | b.fl) & CAND_TIE);
        return r;
    }
    return a;
}
// Wave minimum of doubles through 32-bit integer DPP steps.  A double's bits, sign-folded (negative: all bits flipped, else the sign bit
// set), order as unsigned integers exactly as the doubles do (no NaN here; -0 sorts below +0, which never meets it: distances are
// square roots of non-negative sums).  The high words go through six v_min_u32 DPP steps -- one instruction each, where v_min_f64 needs
// two DPP moves and a quarter-rate fp64 instruction per step --; only if several lanes share the smallest high word do the low words follow.
// Returns the minimum and the ballot of the lanes that hold exactly it.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = min(v, (unsigned)dppi<0xB1, 0xF>((int)v)); v = min(v, (unsigned)dppi<0x4E, 0xF>((int)v)); v = min(v, (unsigned)dppi<0x141, 0xF>((int)v));
    v = min(v, (unsigned)dppi<0x140, 0xF>((int)v)); v = min(v, (unsigned)dppi<0x142, 0xA>((int)v)); v = min(v, (unsigned)dppi<0x143, 0xC>((int)v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ double wave_argmin_d(double v, unsigned long long& mask)
{
    const unsigned h0 = (unsigned)__double2hiint(v), l0 = (unsigned)__double2loint(v);
    const unsigned sgn = (unsigned)((int)h0 >> 31);
    const unsigned hi = h0 ^ (sgn | 0x80000000u), lo = l0 ^ sgn;
    const unsigned mh = wave_min_u32(hi);
    const bool a = hi == mh;
    unsigned long long ma = __ballot(a);
    if (ma & (ma - 1)) { const unsigned ml = wave_min_u32(a ? lo : 0xffffffffu); ma = __ballot(a && lo == ml); }
    mask = ma;
    return readlane_d(v, __builtin_amdgcn_readfirstlane(__ffsll((long long)ma) - 1));
}
// wave arg-min: the minimum value, a ballot of the lanes that hold it, the winner's record by v_readlane (wave_min_c's rules)
__device__ __forceinline__ RCand wave_min_rc(RCand m)
{
    const bool has = m.i >= 0;
    RCand r = rc_none();
    if (__ballot(has) == 0ull) return r;
    unsigned long long mm;
    const double vmin = wave_argmin_d(has ? m.v : (double)INFINITY, mm);
    const bool at = has && ((mm >> (threadIdx.x & 63)) & 1ull);
    const unsigned long long mask = __ballot(at);
    if (mask == 0) return r;
    unsigned long long wm = mask;
    int extra = 0;
    if (mask & (mask - 1)) {
        const int ii = wave_min_i(at ? m.i : 0x7fffffff);
        wm = __ballot(at && m.i == ii);
        if (wm != mask && vmin < (double)INFINITY) extra = CAND_TIE;
    }
    const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)wm) - 1);
    r.v = vmin; r.i = __builtin_amdgcn_readlane(m.i, l); r.y = __builtin_amdgcn_readlane(m.y, l);
    r.fl = __builtin_amdgcn_readlane(m.fl, l) | extra;
    r.szi = __builtin_amdgcn_readlane(m.szi, l); r.szy = __builtin_amdgcn_readlane(m.szy, l);
    r.tyi = __builtin_amdgcn_readlane(m.tyi, l); r.tyy = __builtin_amdgcn_readlane(m.tyy, l);
    return r;
}
__device__ __forceinline__ void rq_acc(RQ& m, double v, int j, int sz, int ty)       // ascending j
{
    if (v < m.v) { m.v2 = m.v; m.v = v; m.i = j; m.sz = sz; m.ty = ty; }
    else if (v < m.v2) m.v2 = v;
}
__device__ __forceinline__ RQ rq_merge(RQ a, RQ b)
{
    if (b.i < 0) return a;
    if (a.i < 0) return b;
    const bool bw = b.v < a.v || (b.v == a.v && b.i < a.i);
    RQ r = bw ? b : a;
    r.v2 = fmin(bw ? a.v : b.v, fmin(a.v2, b.v2));
    return r;
}
__device__ __forceinline__ RQ wave_min_rq(RQ m)
{
    const bool has = m.i >= 0;
    RQ r = rq_none();
    if (__ballot(has) == 0ull) return r;
    unsigned long long mm;
    const double vmin = wave_argmin_d(has ? m.v : (double)INFINITY, mm);
    unsigned long long wm = __ballot(has && ((mm >> (threadIdx.x & 63)) & 1ull));
    if (wm == 0) return r;
    if (wm & (wm - 1)) {           // the same minimum in several lanes: the lowest column, as a sequential scan would find it
        const bool at = has && ((wm >> (threadIdx.x & 63)) & 1ull);
        const int ii = wave_min_i(at ? m.i : 0x7fffffff);
        wm = __ballot(at && m.i == ii);
    }
    const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)wm) - 1);
    r.v = vmin; r.i = __builtin_amdgcn_readlane(m.i, l);
    r.sz = __builtin_amdgcn_readlane(m.sz, l); r.ty = __builtin_amdgcn_readlane(m.ty, l);
    // every lane but the winner's contributes its own minimum, the winner's lane its second value
    const bool win = (int)(threadIdx.x & 63) == l;
    unsigned long long m2;
    r.v2 = wave_argmin_d(win ? m.v2 : (has ? m.v : (double)INFINITY), m2);
    return r;
}
__device__ __forceinline__ unsigned lo32(double v) { return (unsigned)(unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned hi32(double v) { return (unsigned)((unsigned long long)__double_as_longlong(v) >> 32); }
__device__ __forceinline__ unsigned rq_word(const RQ& q, int k)      // word k of a row partial
{
    return k == 0 ? lo32(q.v) : k == 1 ? hi32(q.v) : k == 2 ? (unsigned)q.i : k == 3 ? lo32(q.v2) : k == 4 ? hi32(q.v2) : k == 5 ? (unsigned)q.sz : (unsigned)q.ty;
}

#ifdef SD_LINKAGE_STAMPS
#ifndef SD_LINKAGE_STAMP_MASK
#define SD_LINKAGE_STAMP_MASK 0xffff
#endif
// (a stamp costs up to a few hundred ns -- it waits for the scalar / LDS queue: enable few at a time, -DSD_LINKAGE_STAMP_MASK=<bits>; an interval then runs from the previous ENABLED stamp)
// s_memtime (shader clock), not s_memrealtime: one s_memrealtime per merge round alone cost 1.7 us per round here.  Units: kilo-cycles / 10 in the output.
#define RSTAMP(i) do { if ((SD_LINKAGE_STAMP_MASK >> (i)) & 1) { const unsigned long long t_ = __builtin_readcyclecounter(); acc[i] += t_ - tS; tS = t_; } } while (0)
#else
#define RSTAMP(i) do { } while (0)
#endif

// TB = launch bound (256 / 512 / 1024 threads): the register budget follows it -- 128 VGPRs at 1024 threads spill part of the column state
template <bool ONEX, int TB, int UU>
__global__ __launch_bounds__(TB) void k_linkage_rg(double* D, int n, int* cid, const int* nb0, const double* md0, const double* md20, double* Z,
                                                         MwGran* gran /*[2][G][RG_SLOT], zeroed*/, unsigned* sync, int cap /*columns per workgroup + 1*/, int G, int helper,
                                                         int k0 /*merges already done*/, const int* sz0, const int* ty0 /*cluster size (0 = gone) / last rewrite of every column after k0 merges; null: a fresh start*/)
{
    extern __shared__ __attribute__((aligned(16))) int dyn_lds[];
    // what the cooperative row scans of a retry round need of a column they do not own in registers: last rewrite, cluster size, "gone"
    int* l_ty = dyn_lds;                              // [cap]
    int* l_sz = l_ty + cap;                           // [cap]
    unsigned char* l_dead = (unsigned char*)(l_sz + cap);   // [cap]
    __shared__ RQ sh_q[RG_T_MAX / 64];
    __shared__ RCand sh_c[RG_T_MAX / 64];
    __shared__ int sh_tie[RG_T_MAX / 64];
    __shared__ RQ s_part[RG_KR][RG_T_MAX / 64];
    __shared__ RQ s_row[RG_KR];
    __shared__ unsigned s_words[RG_GMAX][RG_SLOT + 1];   // this round's slots as received (+1: lane u reads word w of slot u)
    __shared__ RCand s_cand[RG_GMAX + 1];
    __shared__ int s_L[2][RG_KR], s_Lty[2][RG_KR], s_Lsz[2][RG_KR];
    __shared__ int s_nL[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // helper = 1: the last wave owns no columns and takes no part in the hand-offs; after every merge it requests the rows of the runner-up
    // neighbours of the new cluster for this workgroup's columns, so that they sit in the XCD's L2 when one of them becomes the next merge's
    // partner (on clustered data the new cluster is in 97 % of the next merges, and its partner was the 2nd or 3rd nearest one or two merges
    // before in 85 % of them: profiles/r05_linkage_*.txt).  Results do not depend on it.
    const int T = (int)blockDim.x - (helper ? 64 : 0), NW = (int)blockDim.x >> 6;
    const bool is_helper = helper && wv == NW - 1;
    int g = blockIdx.x;
    if constexpr (ONEX) {
        // 8 G workgroups were launched; the first G that find themselves on XCC 0 take part (rank = ticket), the others leave (k_linkage_mw)
        __shared__ int s_ticket;
        if (tid == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));       // HW_REG_XCC_ID[3:0]
            s_ticket = (xcc == 0) ? (int)atomicAdd(&sync[6], 1u) : -1;
        }
        __syncthreads();
        g = s_ticket;
        if (g < 0 || g >= G) return;
    }
    const int64_t N = n;
    const int colsB = cap - 1;
    const int z0 = g * colsB;
    const int nown = n - z0 < colsB ? (n - z0 > 0 ? n - z0 : 0) : colsB;
    const int nu = (nown + T - 1) / T;                       // register slots in use (uniform over the workgroup), <= UU
    const int zsafe = z0 < n ? z0 : 0;                       // a valid column for the loads of idle register slots
    unsigned bar = 0;
    int par = 0, lp = 0;
#ifdef SD_LINKAGE_STAMPS
    unsigned long long tS = __builtin_readcyclecounter(), acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned fix_lanes = 0, fix_waves = 0;
#endif

    // ---- per-column state, registers
    int zc[UU], c_nb[UU], c_fl[UU], c_ty[UU], c_sz[UU], c_nbsz[UU], c_nbty[UU];
    double c_md[UU], c_md2[UU];
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        const int p = tid + u * T;
        const int z = (!is_helper && p < nown) ? z0 + p : -1;
        zc[u] = z;
        const bool row = z >= 0 && z < n - 1;
        c_md[u] = row ? md0[z] : (double)INFINITY; c_md2[u] = row ? md20[z] : (double)INFINITY; c_nb[u] = row ? nb0[z] : -1;
        c_fl[u] = z >= 0 ? 1 : 2;                               // bit 0 fresh, bit 1 gone (or no column)
        c_ty[u] = -1; c_sz[u] = 1; c_nbsz[u] = 1; c_nbty[u] = -1;
        if (sz0 && z >= 0) {                                    // continuing behind k0 merges another kernel made (run_linkage: the heap replay took the duplicates)
            c_sz[u] = sz0[z]; c_ty[u] = ty0[z];
            if (c_sz[u] == 0) { c_fl[u] = 2; c_nb[u] = -1; c_md[u] = (double)INFINITY; c_md2[u] = (double)INFINITY; }
            if (c_nb[u] >= 0) { c_nbsz[u] = sz0[c_nb[u]]; c_nbty[u] = ty0[c_nb[u]]; }
        }
        if (z >= 0) { l_ty[p] = c_ty[u]; l_sz[p] = c_sz[u]; l_dead[p] = (c_fl[u] & 2) ? 1 : 0; }
    }
    if (tid == 0) { s_nL[0] = 0; s_nL[1] = 0; }
    __syncthreads();

    // receive round `bar` of every workgroup's slot (nw words each) into s_words; false on timeout
    auto consume = [&](int nw) -> bool {
        const MwGran* base = gran + (size_t)par * G * RG_SLOT;
        bool ok = true;
        const int total = G * nw;
        // a thread's granules are requested together and re-requested together until all of them carry this round's tag (one after the other each
        // would cost its own round trip to the L2: 3 in a row for some threads at 32 workgroups x 16 words and 256 threads)
        for (int i0 = is_helper ? total : tid; i0 < total; i0 += 4 * T) {
            const MwGran* p[4]; MwGran v[4]; int sl[4], wd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = i0 + j * T;
                const int ic = idx < total ? idx : i0;
                sl[j] = ic / nw; wd[j] = ic - sl[j] * nw;
                p[j] = base + (size_t)sl[j] * RG_SLOT + wd[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = LDG(p[j]);
            unsigned spins = 0;
            for (;;) {
                bool pend = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) pend |= (unsigned)(v[j] >> 32) != bar;
                if (!pend) break;
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int j = 0; j < 4; ++j) if ((unsigned)(v[j] >> 32) != bar) v[j] = LDG(p[j]);
                if (++spins > (1u << 24)) { sync[1] = 1; ok = false; break; }     // ~seconds: never in a healthy run
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (i0 + j * T < total) s_words[sl[j]][wd[j]] = (unsigned)v[j];
        }
        return __syncthreads_and(ok ? 1 : 0) != 0;
    };
    auto word_d = [&](int sl, int wd) -> double {
        return __longlong_as_double((long long)(((unsigned long long)s_words[sl][wd + 1] << 32) | s_words[sl][wd]));
    };
    auto slot_cand = [&](int u) -> RCand {
        RCand c; c.v = word_d(u, 0); c.i = (int)s_words[u][2]; c.y = (int)s_words[u][3]; c.fl = (int)s_words[u][4] & ~RG_ROWTIE;
        c.szi = (int)s_words[u][5]; c.szy = (int)s_words[u][6]; c.tyi = (int)s_words[u][7]; c.tyy = (int)s_words[u][8];
        return c;
    };
    auto slot_q = [&](int u, int base) -> RQ {
        RQ q; q.v = word_d(u, base); q.i = (int)s_words[u][base + 2]; q.v2 = word_d(u, base + 3); q.sz = (int)s_words[u][base + 5]; q.ty = (int)s_words[u][base + 6];
        return q;
    };
    // the workgroup's slot for the next round, one store instruction of wave 0 (lane w stores word w).  Every wave has drained its
    // stores to the distance matrix in front of the barrier that precedes this call: whoever sees the slot may read them.
    auto publish = [&](const RCand& m, const RQ& q, int row_tie, int nL, const RQ* rows /*LDS*/) {
        ++bar;
        if (wv == 0) {
            MwGran* sl = gran + ((size_t)par * G + g) * RG_SLOT;
            const MwGran tag = (MwGran)bar << 32;
            unsigned w = 0;
            bool on = false;
            if (lane < RG_CW) {
                on = true;
                w = lane == 0 ? lo32(m.v) : lane == 1 ? hi32(m.v) : lane == 2 ? (unsigned)m.i : lane == 3 ? (unsigned)m.y : lane == 4 ? (unsigned)(m.fl | (row_tie ? RG_ROWTIE : 0))
                  : lane == 5 ? (unsigned)m.szi : lane == 6 ? (unsigned)m.szy : lane == 7 ? (unsigned)m.tyi : (unsigned)m.tyy;
            } else if (nL == 0) {
                if (lane < RG_CW + RG_QW) { on = true; w = rq_word(q, lane - RG_CW); }
            } else if (lane < RG_CW + RG_QW * nL) {
                const int r = (lane - RG_CW) / RG_QW;
                const RQ t = rows[r];
                on = true; w = rq_word(t, lane - RG_CW - RG_QW * r);
            }
            if (on) STX<ONEX>(&sl[lane], tag | (MwGran)w);
        }
    };
    // after consume(): every WAVE folds the G slots itself -- global best, NN(y), "row x had a second pair"
    RCand d_best = rc_none(); RQ d_nn = rq_none(); int d_rowtie = 0;
    auto digest = [&](int nLprev, const int* Lprev, const int* Lprev_ty, const int* Lprev_sz, bool with_nn) {
        if (tid < G) s_cand[tid] = slot_cand(tid);          // kept for pick_stale (read there behind a barrier)
        if (nLprev > 0) {
            for (int r = wv; r < nLprev; r += NW) {          // refreshed rows: one wave folds the G partial minima of a row
                RQ a = rq_none();
                for (int u = lane; u < G; u += 64) a = rq_merge(a, slot_q(u, RG_CW + RG_QW * r));
                a = wave_min_rq(a);
                if (lane == 0) s_row[r] = a;
            }
            __syncthreads();
        }
        RCand b = rc_none();
        RQ a = rq_none();
        int rt = 0;
        for (int u = lane; u < G; u += 64) {
            b = rc_better(b, slot_cand(u));
            if (with_nn) a = rq_merge(a, slot_q(u, RG_CW));
            if (nLprev == 0) rt |= (int)s_words[u][4] & RG_ROWTIE;
        }
        if (lane < nLprev) {           // the rows refreshed in this round are exact now
            const RQ q = s_row[lane];
            RCand c; c.i = Lprev[lane]; c.y = q.i; c.v = (q.i < 0) ? (double)INFINITY : q.v; c.fl = 1;
            c.szi = Lprev_sz[lane]; c.tyi = Lprev_ty[lane]; c.szy = q.sz; c.tyy = q.ty;
            if (c.y >= 0) b = rc_better(b, c);
        }
        d_best = wave_min_rc(b);
        if (with_nn) d_nn = wave_min_rq(a);
        d_rowtie = (nLprev == 0 && __ballot(rt != 0) != 0ull) ? 1 : 0;
        if (nLprev > 0) {
            // the owner of a refreshed row takes it into its registers
            for (int r = 0; r < nLprev; ++r) {
                const int xr = Lprev[r];
                const RQ q = s_row[r];
#pragma unroll
                for (int u = 0; u < UU; ++u)
                    if (zc[u] == xr) {
                        c_nb[u] = q.i; c_md[u] = (q.i < 0) ? (double)INFINITY : q.v; c_md2[u] = (q.i < 0) ? (double)INFINITY : q.v2;
                        c_fl[u] = (c_fl[u] & 2) | 1; c_nbsz[u] = q.sz; c_nbty[u] = q.ty;
                    }
            }
        }
    };
    // local arg-min over the owned active rows, skipping the rows being refreshed this round
    auto local_argmin = [&](int nL, const int* L) -> RCand {
        int ex[RG_KR];
#pragma unroll
        for (int r = 0; r < RG_KR; ++r) ex[r] = r < nL ? L[r] : -1;
        RCand m = rc_none();
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            const int z = zc[u];
            bool skip = z < 0 || (c_fl[u] & 2) || z >= n - 1;
#pragma unroll
            for (int r = 0; r < RG_KR; ++r) skip |= (z == ex[r]);
            if (!skip) rc_acc(m, c_md[u], z, c_nb[u], c_fl[u] & 1, c_sz[u], c_nbsz[u], c_ty[u], c_nbty[u]);
        }
        m = wave_min_rc(m);
        __syncthreads();
        if (lane == 0) sh_c[wv] = m;
        __syncthreads();
        RCand r = rc_none();
        if (lane < NW) r = sh_c[lane];
        return wave_min_rc(r);
    };
    // this workgroup's share -- its own columns -- of the nL rows in L: wave tasks (row r, sub-slice s)
    auto scan_rows = [&](int nL, const int* L, const int* Lty, RQ* outv /*LDS [RG_KR]*/) {
        if (nL <= 0) return;
        const int S = NW >= nL ? NW / nL : 1;
        for (int t = wv; t < nL * S; t += NW) {
            const int r = t % nL, sidx = t / nL;
            const int x = L[r];
            const int txr = Lty[r];
            RQ q = rq_none();
            const int64_t jend = (int64_t)z0 + nown, step = (int64_t)S * 64;
            for (int64_t j0 = (x + 1 > z0 ? x + 1 : z0) + (int64_t)sidx * 64 + lane; j0 < jend; j0 += step * 4) {
                double v[4]; bool ok[4]; int sj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t j = j0 + u * step; const int64_t jc = j < jend ? j : jend - 1;
                    sj[u] = (int)(jc - z0);
                    ok[u] = j < jend && !l_dead[sj[u]];
                    v[u] = LDG(txr >= l_ty[sj[u]] ? &D[(int64_t)x * N + jc] : &D[jc * N + x]);      // entry {x, j} from the row written last
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (ok[u]) rq_acc(q, v[u], (int)(j0 + u * step), l_sz[sj[u]], l_ty[sj[u]]);
            }
            q = wave_min_rq(q);
            if (lane == 0) s_part[r][sidx] = q;
        }
        __syncthreads();
        if (tid < nL) {
            RQ q = s_part[tid][0];
            for (int k2 = 1; k2 < S; ++k2) q = rq_merge(q, s_part[tid][k2]);
            outv[tid] = q;
        }
        __syncthreads();
    };
    // the next refresh list: the RG_KR best stale candidates among s_cand[0..G) and `extra`; wave 0 extracts them, every
    // workgroup arrives at the same list (with the ty and the size of each listed row's cluster)
    MinIdx none; none.v = INFINITY; none.i = -1;
    auto pick_stale = [&](const RCand& extra, int slot) {
        __syncthreads();               // s_cand of this round complete
        if (wv == 0) {
            constexpr int CU = (RG_GMAX + 1 + 63) / 64;
            MinIdx c[CU]; int oty[CU], osz[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int idx = lane + 64 * u;
                c[u] = none; oty[u] = -1; osz[u] = 0;
                if (idx <= G) {
                    RCand o = extra;
                    if (idx < G) o = s_cand[idx];
                    if (o.i >= 0 && !(o.fl & 1) && o.v != INFINITY) { c[u].v = o.v; c[u].i = o.i; oty[u] = o.tyi; osz[u] = o.szi; }
                }
            }
            int nl = 0;
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const bool st = c[u].i >= 0;
                const unsigned long long mk = __ballot(st);
                const int at = nl + __popcll(mk & ((1ull << lane) - 1ull));
                if (st && at < RG_KR) { s_L[slot][at] = c[u].i; s_Lty[slot][at] = oty[u]; s_Lsz[slot][at] = osz[u]; }
                nl += __popcll(mk);
            }
            if (nl > RG_KR) {                       // more than RG_KR: the best by (bound, row)
                nl = 0;
                for (int r = 0; r < RG_KR; ++r) {
                    MinIdx bq = none;
#pragma unroll
                    for (int u = 0; u < CU; ++u) bq = better(bq, c[u]);
                    bq = wave_min(bq);
                    if (bq.i < 0) break;
#pragma unroll
                    for (int u = 0; u < CU; ++u)
                        if (c[u].i == bq.i) { s_L[slot][r] = bq.i; s_Lty[slot][r] = oty[u]; s_Lsz[slot][r] = osz[u]; c[u].i = -1; }
                    nl = r + 1;
                }
            }
            if (lane == 0) s_nL[slot] = nl;
        }
        __syncthreads();
    };
    RCand nocand = rc_none(); nocand.fl = 1;

    // ---- initial state: exact bounds from k_row_nn
    {
        const RCand m0 = local_argmin(0, s_L[0]);
        publish(m0, rq_none(), 0, 0, s_row);
    }
    if (!consume(RG_MW)) return;
    digest(0, s_L[0], s_Lty[0], s_Lsz[0], false);
    par ^= 1;
    RCand best = d_best;
    if (!((best.fl & 1) && best.y >= 0)) pick_stale(nocand, lp);

    for (int k = k0; k < n - 1; ++k) {
        // ---- lazy validation (cl.cpp:323-339): cooperative refresh of the best stale candidates
        for (int guard = 0; guard <= n - k; ++guard) {
            if ((best.fl & 1) && best.y >= 0) break;
            if (g == 0 && tid == 0) sync[2] += 1;            // diagnostic: retry rounds
            const int nL = s_nL[lp]; const int* L = s_L[lp];
            RSTAMP(5);
            scan_rows(nL, L, s_Lty[lp], s_row);
            RSTAMP(0);
            const RCand m = local_argmin(nL, L);
            publish(m, rq_none(), 0, nL, s_row);
            RSTAMP(1);
            if (!consume(nL > 0 ? RG_CW + RG_QW * nL : RG_MW)) return;
            RSTAMP(2);
            digest(nL, L, s_Lty[lp], s_Lsz[lp], false);
            par ^= 1;
            best = d_best;
            RSTAMP(3);
            lp ^= 1;
            if (!((best.fl & 1) && best.y >= 0)) pick_stale(nocand, lp);
            RSTAMP(4);
        }
        if (best.fl & CAND_TIE) {      // the closest pair is not unique: the heap decides (run_linkage); the height of the tie goes along
            if (g == 0 && tid == 0) { sync[5] = 1; sync[26] = lo32(best.v); sync[27] = hi32(best.v); }
            return;
        }
        // ---- merge (x, y) at height dist; everything about the pair came with the candidate
        const int x = best.i, y = best.y;
        const double dist = best.v;
        const int nx = best.szi, ny = best.szy, txm = best.tyi, tym = best.tyy;
        int cx_pre = 0, cy_pre = 0;
        if (g == 0 && tid == 0) { cx_pre = cid[x]; cy_pre = cid[y]; }      // dendrogram ids of the pair (this thread is their only reader and writer)
        auto write_Z = [&]() {
            if (tid == 0 && g == 0) {
                int ix = cx_pre, iy = cy_pre;
                if (ix > iy) { const int t = ix; ix = iy; iy = t; }
                Z[(size_t)k * 4 + 0] = (double)ix; Z[(size_t)k * 4 + 1] = (double)iy;
                Z[(size_t)k * 4 + 2] = dist;       Z[(size_t)k * 4 + 3] = (double)(nx + ny);
                cid[y] = n + k;
            }
        };
        if (k == n - 2) { write_Z(); break; }
        // ---- one pass over the owned active columns: Lance-Williams update + neighbour patches (cl.cpp:361-392), NN(y) partial
        // from the fresh distances (cl.cpp:395-404), next local arg-min
        RSTAMP(5);
        RQ q = rq_none();
        RCand m = rc_none();
        int row_tie = 0;
        double dzx[UU], dzy[UU]; bool act[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            act[u] = false;
            if (u < nu && !is_helper) {
                const int z = zc[u];
                act[u] = z >= 0 && !(c_fl[u] & 2) && z != x && z != y;
                const int zl = act[u] ? z : zsafe;
                // the row copies are requested at once; where a bystander's row was written after the pair's, the entry is fetched from there below
                dzx[u] = LDG(&D[(int64_t)x * N + zl]);
                dzy[u] = LDG(&D[(int64_t)y * N + zl]);
            }
        }
        RSTAMP(8);
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            if (u < nu && act[u]) {
#ifdef SD_LINKAGE_STAMPS
                { const bool fx = (txm < c_ty[u]) || (tym < c_ty[u]); fix_lanes += fx ? 1u : 0u; }
#endif
                if (txm < c_ty[u]) dzx[u] = LDG(&D[(int64_t)zc[u] * N + x]);        // z's row was written after x's: the current {z, x} is there
                if (tym < c_ty[u]) dzy[u] = LDG(&D[(int64_t)zc[u] * N + y]);
            }
        }
        RSTAMP(9);
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            if (!(u < nu && act[u])) continue;
            const int z = zc[u];
            const double nd = lw_centroid(dzx[u], dzy[u], dist, nx, ny);
            STX<ONEX>(&D[(int64_t)y * N + z], nd);
            if (z > x && dzx[u] == dist) row_tie = 1;         // row x had a second neighbour at exactly the merge height
            double mz = (z < n - 1) ? c_md[u] : (double)INFINITY;
            int nz = c_nb[u], fz = c_fl[u] & 1;
            if (z < y) {
                // row z's entries above the diagonal: x's is gone (if z < x), y's is nd now.  Invariants: mz <= every active entry of the row
                // (the reference's lower bound), m2 <= every active entry OTHER than the neighbour's (k_linkage_mw)
                const double m2 = fmax(c_md2[u], mz);
                if ((z < x && nz == x) || nz == y) {
                    nz = y;
                    if (nd <= m2) { mz = nd; fz = 1; } else { mz = m2; fz = 0; }
                    c_md[u] = mz; c_md2[u] = m2; c_nb[u] = y; c_fl[u] = fz; c_nbsz[u] = nx + ny; c_nbty[u] = k;
                } else if (nd < mz) {
                    c_md2[u] = mz;
                    nz = y; mz = nd; fz = 1; c_md[u] = nd; c_nb[u] = y; c_fl[u] = 1; c_nbsz[u] = nx + ny; c_nbty[u] = k;
                } else if (nd < m2) c_md2[u] = nd;
            } else rq_acc(q, nd, z, c_sz[u], c_ty[u]);
            if (z < n - 1) rc_acc(m, mz, z, nz, fz, c_sz[u], c_nbsz[u], c_ty[u], c_nbty[u]);
        }
        write_Z();
        RSTAMP(6);
        // ---- the two reductions and the flag through ONE LDS exchange; the stores of this wave have landed before its record is visible
        q = wave_min_rq(q);
        m = wave_min_rc(m);
        const int tie_w = __ballot(row_tie != 0) != 0ull ? 1 : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) { sh_q[wv] = q; sh_c[wv] = m; sh_tie[wv] = tie_w; }
        __syncthreads();
        if (NW <= 8) {            // few waves: every lane folds the records itself (LDS broadcast reads) -- shorter than two more DPP reductions
            q = sh_q[0]; m = sh_c[0]; row_tie = sh_tie[0];
            for (int w2 = 1; w2 < NW; ++w2) { q = rq_merge(q, sh_q[w2]); m = rc_better(m, sh_c[w2]); row_tie |= sh_tie[w2]; }
        } else {
            RQ rq = rq_none(); RCand rc = rc_none(); int rt = 0;
            if (lane < NW) { rq = sh_q[lane]; rc = sh_c[lane]; rt = sh_tie[lane]; }
            q = wave_min_rq(rq);
            m = wave_min_rc(rc);
            row_tie = __ballot(rt != 0) != 0ull ? 1 : 0;
        }
        RSTAMP(10);
        // ---- the pair's own columns: x is gone, y is the merged cluster, rewritten in this merge
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            if (zc[u] == x) { c_fl[u] |= 2; l_dead[tid + u * T] = 1; }
            if (zc[u] == y) { c_sz[u] = nx + ny; c_ty[u] = k; l_sz[tid + u * T] = nx + ny; l_ty[tid + u * T] = k; }
        }
        publish(m, q, row_tie, 0, s_row);
        RSTAMP(7);
        if (!consume(RG_MW)) return;
        RSTAMP(2);
        digest(0, s_L[lp], s_Lty[lp], s_Lsz[lp], true);
        RSTAMP(3);
        par ^= 1;
        if (d_rowtie) { if (g == 0 && tid == 0) { sync[5] = 1; sync[26] = lo32(dist); sync[27] = hi32(dist); } return; }     // (this merge's stores into row y are out, and so is row k of Z: run_linkage reads from Z's row 0 whether the matrix is still what linkage_prepare left)
        best = d_best;
        const RQ nn = d_nn;
        // row y: exact by construction (cl.cpp:395-404).  Without an active column above it the row has no pair left, now or later
        // (columns are only ever removed): exact bound +inf
        RCand cy = rc_none();
        if (y < n - 1) {
            if (nn.i >= 0) { cy.v = nn.v; cy.i = y; cy.y = nn.i; cy.fl = 1; cy.szi = nx + ny; cy.szy = nn.sz; cy.tyi = k; cy.tyy = nn.ty; }
#pragma unroll
            for (int u = 0; u < UU; ++u)
                if (zc[u] == y) {
                    c_nb[u] = nn.i; c_md[u] = nn.i >= 0 ? nn.v : (double)INFINITY; c_md2[u] = nn.i >= 0 ? nn.v2 : (double)INFINITY;
                    c_fl[u] = (c_fl[u] & 2) | 1; c_nbsz[u] = nn.sz; c_nbty[u] = nn.ty;
                }
            best = rc_better(best, cy);
        }
        if (is_helper) {
            // the two best local NN(y) candidates other than the winner (whose row the pass requests right now)
            double qv = INFINITY; int qi = -1;
            if (lane < G) { const RQ t = slot_q(lane, RG_CW); qv = t.v; qi = t.i; }
            if (qi == nn.i || qi < 0) qv = INFINITY;
            for (int e = 0; e < 2; ++e) {
                unsigned long long mm;
                const double v = wave_argmin_d(qv, mm);
                const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)mm) - 1);
                const int r = v < (double)INFINITY ? __builtin_amdgcn_readlane(qi, l) : -1;
                if (lane == l) qv = INFINITY;
                if (r >= 0) {
                    // requests whose data nobody waits for: the loads pull the lines into the L2, the register they land in is never read
                    const double* row = D + (int64_t)r * N + z0;
                    for (int j = lane; j < nown; j += 64) {
                        double dump;
                        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(dump) : "v"(row + j) : "memory");
                    }
                }
            }
        }
        lp ^= 1;
        if (!((best.fl & 1) && best.y >= 0)) pick_stale(cy, lp);
        RSTAMP(4);
    }

#ifdef SD_LINKAGE_STAMPS
    if (g == 0 && tid == 0) for (int i = 0; i < 16; ++i) sync[8 + i] = (unsigned)(acc[i] / 10000);   // 10 kilo-cycles
    if (tid == 0) for (int i = 0; i < 16; ++i) sync[32 + g * 16 + i] = (unsigned)(acc[i] / 10000);   // every workgroup's own view
    atomicAdd(&sync[24], fix_lanes);
#endif
}

// launcher: false = the geometry does not fit this kernel (the caller takes k_linkage_mw)
bool linkage_rg_fits(int64_t N, int G, int TH)
{
    if (G < 2 || G > RG_GMAX) return false;
    const int64_t colsB = (N + G - 1) / G;
    return colsB <= (int64_t)(TH <= 512 ? 8 : RG_U) * TH && TH <= RG_T_MAX;         // (8 columns per thread: the <= 512-thread forms only -- register budget)
}
hipError_t linkage_rg_launch(sd_ctx* c, bool onex, int G, int TH, double* D, int n, int* cid, const int* nb, const double* md, const double* md2,
                             double* Z, MwGran* gran, unsigned* sync, int cap, int helper, int k0, const int* sz0, const int* ty0)
{
    if (helper && (TH > 448 || G > 64)) helper = 0;         // (one slot per lane in the helper's fold; 16 waves at most)
    const int TT = TH + (helper ? 64 : 0);
    const size_t dyn = (((size_t)cap * 9) + 15) & ~(size_t)15;
    const bool wide = (int64_t)(cap - 1) > (int64_t)RG_U * TH;         // more than 4 columns per thread: the 8-column form
    const void* f = wide ? (TT <= 256 ? (onex ? (const void*)k_linkage_rg<true, 256, 8> : (const void*)k_linkage_rg<false, 256, 8>)
                                      : (onex ? (const void*)k_linkage_rg<true, 512, 8> : (const void*)k_linkage_rg<false, 512, 8>))
                  : TT <= 256 ? (onex ? (const void*)k_linkage_rg<true, 256, RG_U> : (const void*)k_linkage_rg<false, 256, RG_U>)
                  : TT <= 512 ? (onex ? (const void*)k_linkage_rg<true, 512, RG_U> : (const void*)k_linkage_rg<false, 512, RG_U>)
                              : (onex ? (const void*)k_linkage_rg<true, 1024, RG_U> : (const void*)k_linkage_rg<false, 1024, RG_U>);
    (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    (void)hipGetLastError();
    void* args[] = {&D, &n, &cid, &nb, &md, &md2, &Z, &gran, &sync, &cap, &G, &helper, &k0, &sz0, &ty0};
    // cooperative launch: all workgroups are resident together, or the launch is refused (they poll each other's slots)
    return hipLaunchCooperativeKernel(f, dim3(onex ? 8 * G : G), dim3(TT), args, dyn, c->stream);
}
int linkage_rg_slot_granules() { return RG_SLOT; }

// ---------------------------------------------------------------- tuning hook: what a merge round is made of
// A synthetic round with k_linkage_rg's geometry (G workgroups of T threads, one XCD when onex), its slot exchange and its memory
// pattern, whose parts can be switched on one by one (bits of `parts`): 1 = the two row loads per column (rows picked by a generator every
// workgroup runs alike), 2 = the Lance-Williams arithmetic, 4 = the row-y stores + drain, 8 = the workgroup's two reductions through LDS,
// 16 = publish + consume + digest (the all-to-all of slots).  tools/linkage_parts.py prints the time per round of each combination.
template <bool ONEX>
__global__ __launch_bounds__(256) void k_rg_parts(double* D, int n, MwGran* gran, unsigned* sync, int cap, int G, int rounds, int parts, double* sink)
{
    __shared__ RQ sh_q[4];
    __shared__ RCand sh_c[4];
    __shared__ unsigned s_words[RG_GMAX][RG_SLOT + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int T = blockDim.x, NW = T >> 6;
    int g = blockIdx.x;
    if constexpr (ONEX) {
        __shared__ int s_ticket;
        if (tid == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
            s_ticket = (xcc == 0) ? (int)atomicAdd(&sync[6], 1u) : -1;
        }
        __syncthreads();
        g = s_ticket;
        if (g < 0 || g >= G) return;
    }
    const int64_t N = n;
    const int colsB = cap - 1, z0 = g * colsB;
    const int nown = n - z0 < colsB ? (n - z0 > 0 ? n - z0 : 0) : colsB;
    const int nu = (nown + T - 1) / T;
    const int zsafe = z0 < n ? z0 : 0;
    unsigned bar = 0; int par = 0;
    unsigned long long rngs = 88172645463325252ull;
    double keep = 0.0;
    int y = n / 2;
    for (int r = 0; r < rounds; ++r) {
        rngs ^= rngs << 13; rngs ^= rngs >> 7; rngs ^= rngs << 17;
        const int x = (int)(rngs % (unsigned long long)n);                  // a cold row, as the chain's new partner is
        RQ q = rq_none(); RCand m = rc_none();
        double dzx[RG_U], dzy[RG_U];
#pragma unroll
        for (int u = 0; u < RG_U; ++u) {
            dzx[u] = 1.0 + u; dzy[u] = 2.0 + u;
            if (u < nu && (parts & 1)) {
                const int p = tid + u * T;
                const int zl = p < nown ? z0 + p : zsafe;
                dzx[u] = LDG(&D[(int64_t)x * N + zl]);
                dzy[u] = LDG(&D[(int64_t)y * N + zl]);
            }
        }
#pragma unroll
        for (int u = 0; u < RG_U; ++u) {
            if (!(u < nu)) continue;
            const int p = tid + u * T;
            if (p >= nown) continue;
            const int z = z0 + p;
            double nd = dzx[u] + dzy[u];
            if (parts & 2) nd = lw_centroid(dzx[u], dzy[u], 0.5 + 1e-3 * (r & 7), 3 + (r & 3), 5);
            if (parts & 4) STX<ONEX>(&D[(int64_t)y * N + z], dzy[u]);     // (the loaded value goes back: the matrix stays what it was)
            if (parts & 256) STX<ONEX>(&D[(int64_t)z * N + y], dzy[u]);   // the mirror of row y into column y: one scattered 8-byte store per column
            rq_acc(q, nd, z, 1, -1);
            rc_acc(m, nd + 1.0, z, z + 1, 1, 1, 1, -1, -1);
        }
        if (parts & 8) {
            q = wave_min_rq(q);
            m = wave_min_rc(m);
        }
        if (parts & 32) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) { sh_q[wv] = q; sh_c[wv] = m; }
            __syncthreads();
            q = sh_q[0]; m = sh_c[0];
            for (int w2 = 1; w2 < NW; ++w2) { q = rq_merge(q, sh_q[w2]); m = rc_better(m, sh_c[w2]); }
        }
        const int PW = (parts & 128) ? NW : 1;            // parts & 128: every WAVE publishes its own record (no LDS fold in front of the publish): G * NW slots
        if (parts & 16) {
            ++bar;
            if (parts & 128) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (wv == 0 || (parts & 128)) {
                MwGran* sl = gran + ((size_t)par * G * PW + (size_t)g * PW + ((parts & 128) ? wv : 0)) * RG_SLOT;
                const MwGran tag = (MwGran)bar << 32;
                unsigned w = 0; bool on = false;
                if (lane < RG_CW) { on = true; w = lane == 0 ? lo32(m.v) : lane == 1 ? hi32(m.v) : lane == 2 ? (unsigned)m.i : lane == 3 ? (unsigned)m.y : 1u; }
                else if (lane < RG_CW + RG_QW) { on = true; w = rq_word(q, lane - RG_CW); }
                if (on) STX<ONEX>(&sl[lane], tag | (MwGran)w);
            }
            const MwGran* base = gran + (size_t)par * G * PW * RG_SLOT;
            const int total = G * PW * RG_MW;
            bool ok = true;
            for (int i0 = tid; i0 < total; i0 += 4 * T) {
                const MwGran* p[4]; MwGran v[4]; int sl[4], wd[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int idx = i0 + j * T; const int ic = idx < total ? idx : i0; sl[j] = ic / RG_MW; wd[j] = ic - sl[j] * RG_MW; p[j] = base + (size_t)sl[j] * RG_SLOT + wd[j]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = LDG(p[j]);
                unsigned spins = 0;
                for (;;) {
                    bool pend = false;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pend |= (unsigned)(v[j] >> 32) != bar;
                    if (!pend) break;
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) if ((unsigned)(v[j] >> 32) != bar) v[j] = LDG(p[j]);
                    if (++spins > (1u << 22)) { sync[1] = 1; ok = false; break; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) if (i0 + j * T < total) s_words[sl[j]][wd[j]] = (unsigned)v[j];
            }
            if (!__syncthreads_and(ok ? 1 : 0)) return;
            if (parts & 64) { par ^= 1; keep += (double)s_words[lane % G][2]; continue; }       // the hand-off alone, without the digest
            RCand b = rc_none(); RQ a = rq_none();
            for (int u = lane; u < G * PW; u += 64) {
                RCand c; c.v = __longlong_as_double((long long)(((unsigned long long)s_words[u][1] << 32) | s_words[u][0])); c.i = (int)s_words[u][2]; c.y = (int)s_words[u][3]; c.fl = 1;
                c.szi = c.szy = 1; c.tyi = c.tyy = -1;
                b = rc_better(b, c);
                RQ qq; qq.v = __longlong_as_double((long long)(((unsigned long long)s_words[u][RG_CW + 1] << 32) | s_words[u][RG_CW])); qq.i = (int)s_words[u][RG_CW + 2];
                qq.v2 = __longlong_as_double((long long)(((unsigned long long)s_words[u][RG_CW + 4] << 32) | s_words[u][RG_CW + 3])); qq.sz = 1; qq.ty = -1;
                a = rq_merge(a, qq);
            }
            m = wave_min_rc(b);
            q = wave_min_rq(a);
            par ^= 1;
            if (m.i >= 0 && m.i < n) y = m.i;          // the next round's row y depends on the exchange, as the real one does
        }
        keep += q.v + m.v;
    }
    if (tid == 0) sink[g] = keep;
}
extern "C" int sd_bench_linkage_parts(sd_ctx* c, int64_t N, int G, int rounds, int parts, int onex, double* us_per_round)
{
    if (!c || !us_per_round || N < 64 || G < 2 || G > RG_GMAX || rounds < 1) return SD_ERR_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return SD_ERR_HIP;
    const int cap = (int)((N + G - 1) / G) + 1;
    if (cap - 1 > RG_U * 256) SD_FAIL(c, SD_ERR_ARG, "sd_bench_linkage_parts: %d columns per workgroup (limit %d)", cap - 1, RG_U * 256);
    WS(c, double, D, "cl_Dsq", (size_t)N * N);
    if ((parts & 128) && G * 4 > RG_GMAX) SD_FAIL(c, SD_ERR_ARG, "sd_bench_linkage_parts: per-wave slots need workgroups * 4 <= %d", RG_GMAX);
    WS(c, MwGran, gran, "cl_gran", (int64_t)2 * G * 4 * RG_SLOT);
    WS(c, unsigned, sync, "cl_sync", 32 + 16 * 256);
    WS(c, double, sink, "bb_scratch", 1 << 22);
    HIPCHK(c, hipMemsetAsync(gran, 0, (size_t)2 * G * 4 * RG_SLOT * sizeof(MwGran), c->stream));
    HIPCHK(c, hipMemsetAsync(sync, 0, (32 + 16 * 256) * sizeof(unsigned), c->stream));
    HIPCHK(c, hipMemsetAsync(D, 0x3f, (size_t)N * N * sizeof(double), c->stream));          // finite doubles
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    int n = (int)N, cap_i = cap;
    void* args[] = {&D, &n, &gran, &sync, &cap_i, &G, &rounds, &parts, &sink};
    HIPCHK(c, hipEventRecord(e0, c->stream));
    const void* f = onex ? (const void*)k_rg_parts<true> : (const void*)k_rg_parts<false>;
    HIPCHK(c, hipLaunchCooperativeKernel(f, dim3(onex ? 8 * G : G), dim3(256), args, 0, c->stream));
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    *us_per_round = ms * 1e3 / rounds;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    unsigned h[8];
    HIPCHK(c, hipMemcpy(h, sync, sizeof(h), hipMemcpyDeviceToHost));
    if (h[1]) SD_FAIL(c, SD_ERR_HIP, "sd_bench_linkage_parts: slot poll timed out");
    return SD_OK;
}
