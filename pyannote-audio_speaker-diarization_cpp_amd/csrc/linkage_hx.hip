// linkage_hx.hip -- k_linkage_hx: the reference's fast_linkage (cl.cpp:289-406) replayed EXACTLY -- indexed binary heap included
// (cl.cpp:28-119) -- with the O(n) parts of every step spread over G worker workgroups.
//
// Why: on data with exact ties (duplicated embeddings: looped audio, digital silence) the merge order among equal heights is whatever
// the reference's heap hands out first, and that depends on the whole history of its array; the cooperative kernels (k_linkage_rg /
// k_linkage_mw) take a merge only while the closest pair is unique and stop at the first tie.  Until round 5 the whole job then went
// to k_linkage_heap, ONE workgroup: 786 ms at N = 14 382 (the raw 1-h workload), 10x the tie-free time.  Here the heap is still replayed
// operation by operation by one thread (workgroup 0, the "master", which owns no columns), but
//   * the Lance-Williams update, the neighbour patches (cl.cpp:361-392) and the NN(y) search (cl.cpp:395-404) of a merge run on the
//     workers, each on its own contiguous range of columns of the full N x N matrix (k_linkage_rg's layout: row y is rewritten, not
//     mirrored; the current copy of an entry lives in the row of the cluster that was a merge's survivor last, index `ty`);
//   * the rows whose lower bound dropped (the reference's change_value calls inside the z loop, IN ASCENDING z) come back as one
//     ordered list per worker -- workers own ascending column ranges, so concatenating the lists in worker order is ascending z;
//   * a stale heap top (cl.cpp:329-338) is rescanned by all workers, each its own columns of that row;
//   * the top 8 191 heap entries (levels 0-12) live in the master's LDS, the rest in global memory.
// Hand-offs are k_linkage_rg's 8-byte {payload, tag} granules: one command record master -> workers, one reply record per worker.
// Z is bit-identical to the reference for every input (tests/test_gpu_parity.py: ties, lattices, duplicates).
#include "common.h"
#include "linkage_dev.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_linkage_hx's fence-free hand-offs (sc1 loads / stores, 8-byte tagged granules) are written for gfx950 only"
#endif

#define HX_T 256
#define HX_U 4            // columns per worker thread
#define HX_GMAX 128       // workers
#define HX_CMDW 10        // command: op, x, y, dist (2), nx, ny, tx, ty, k
#define HX_REPW 4         // reply: count of changed rows | NN partial: column, value (2)
#define HX_STAGE 2048     // changed rows staged in the master's LDS per pass
#define HX_LDS_HEAP 8191  // heap entries kept in LDS (levels 0 .. 12)
#define HX_OP_SCAN 1
#define HX_OP_MERGE 2
#define HX_OP_QUIT 3

// the reference's heap (cl.cpp:28-119) over two tiers of storage; every operation below is the reference's, swap by swap
// Storage: the values of the first `lc` entries (the top levels) in LDS, the rest in global memory; keys and positions in LDS as 16-bit
// numbers when n <= 65 535 (S16), else in global memory.  A sift moves ONE element along a path: it is held in registers and written once
// at its final place, the elements it passes are shifted by one level -- the array the reference's swap-by-swap loops leave behind
// (cl.cpp:44-78: every swap exchanges the moving element with the next one on the path), with a third of the memory traffic.
template <bool S16>
struct HxHeap {
    double* lv; double* gv; unsigned short* lk16; unsigned short* lp16; int* gk; int* gp; int lc; int size;
    __device__ __forceinline__ double V(int i) const { return i < lc ? lv[i] : gv[i]; }
    __device__ __forceinline__ void setV(int i, double v) { if (i < lc) lv[i] = v; else gv[i] = v; }
    __device__ __forceinline__ int K(int i) const { if constexpr (S16) return (int)lk16[i]; else return gk[i]; }
    __device__ __forceinline__ void setK(int i, int k) { if constexpr (S16) lk16[i] = (unsigned short)k; else gk[i] = k; }
    __device__ __forceinline__ int P(int key) const { if constexpr (S16) return (int)lp16[key]; else return gp[key]; }
    __device__ __forceinline__ void setP(int key, int i) { if constexpr (S16) lp16[key] = (unsigned short)i; else gp[key] = i; }
};
template <bool S16> __device__ __forceinline__ void hx_swap(HxHeap<S16>& h, int a, int b)                          // cl.cpp:70-78
{
    const double va = h.V(a), vb = h.V(b);
    h.setV(a, vb); h.setV(b, va);
    const int ka = h.K(a), kb = h.K(b);
    h.setK(a, kb); h.setK(b, ka);
    h.setP(ka, b); h.setP(kb, a);
}
template <bool S16> __device__ __forceinline__ void hx_down(HxHeap<S16>& h, int idx)                                // cl.cpp:53-68
{
    const double v = h.V(idx);
    const int k = h.K(idx);
    const int start = idx;
    int ch = 2 * idx + 1;
    while (ch < h.size) {
        double vc = h.V(ch);
        if (ch + 1 < h.size) { const double v1 = h.V(ch + 1); if (v1 < vc) { vc = v1; ch += 1; } }
        if (v > vc) { const int kc = h.K(ch); h.setV(idx, vc); h.setK(idx, kc); h.setP(kc, idx); idx = ch; ch = 2 * idx + 1; }
        else break;
    }
    if (idx != start) { h.setV(idx, v); h.setK(idx, k); h.setP(k, idx); }
}
template <bool S16> __device__ __forceinline__ void hx_up(HxHeap<S16>& h, int idx)                                  // cl.cpp:44-51
{
    const double v = h.V(idx);
    const int k = h.K(idx);
    const int start = idx;
    while (idx > 0) {
        const int par = (idx - 1) >> 1;
        const double vp = h.V(par);
        if (!(vp > v)) break;
        const int kp = h.K(par);
        h.setV(idx, vp); h.setK(idx, kp); h.setP(kp, idx);
        idx = par;
    }
    if (idx != start) { h.setV(idx, v); h.setK(idx, k); h.setP(k, idx); }
}
template <bool S16> __device__ __forceinline__ void hx_change(HxHeap<S16>& h, int key, double v)                    // cl.cpp:108-117
{
    const int idx = h.P(key);
    const double old = h.V(idx);
    h.setV(idx, v);
    if (v < old) hx_up(h, idx); else hx_down(h, idx);
}
// change_value for a value known to have DROPPED (the rows of cl.cpp:381-392: dist < min_dist[z]): the same as hx_change without reading the old value
template <bool S16> __device__ __forceinline__ void hx_decrease(HxHeap<S16>& h, int key, double v)
{
    const int idx = h.P(key);
    h.setV(idx, v);
    hx_up(h, idx);
}

template <bool ONEX, bool S16>
__global__ __launch_bounds__(HX_T) void k_linkage_hx(double* D, int n, int* cid, int* size, int* tyv, int* nb, double* md, double* Z,
                                                     double* g_hval, int* g_hkey, int* g_hpos, MwGran* cmd /*[16]*/, MwGran* rep /*[G][8]*/,
                                                     int* chg_z /*[G][cap]*/, double* chg_v /*[G][cap]*/, unsigned* sync, int cap, int G /*workers*/, int lc,
                                                     double stop_above /*stop in front of the first merge higher than this (inf = never); sync[3] = merges done*/)
{
    extern __shared__ __attribute__((aligned(16))) int dyn_lds[];     // master: heap tier [lc] doubles + [lc] ints
    __shared__ MinIdx sh[HX_T / 64];
    __shared__ int s_cnt[HX_U][HX_T / 64];
    __shared__ unsigned s_rep[HX_GMAX][HX_REPW];
    __shared__ int s_pre[HX_GMAX + 1];
    __shared__ int st_z[HX_STAGE];
    __shared__ double st_v[HX_STAGE];
    __shared__ int s_x, s_y, s_ok, s_tx, s_ty, s_nx, s_ny;
    __shared__ double s_d;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = HX_T / 64;
    int g = blockIdx.x;
    if constexpr (ONEX) {
        // 8 (G + 1) workgroups were launched; the first G + 1 that find themselves on XCC 0 take part (rank = ticket), the others leave
        __shared__ int s_ticket;
        if (tid == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));       // HW_REG_XCC_ID[3:0]
            s_ticket = (xcc == 0) ? (int)atomicAdd(&sync[6], 1u) : -1;
        }
        __syncthreads();
        g = s_ticket;
        if (g < 0 || g > G) return;
    }
    const int64_t N = n;
    unsigned seq = 0;                       // commands sent / received so far
    // one 8-byte granule = {payload word, tag}
    auto poll = [&](const MwGran* p, unsigned tag, bool& ok) -> unsigned {
        MwGran v = LDG(p);
        unsigned spins = 0;
        while ((unsigned)(v >> 32) != tag) {
            __builtin_amdgcn_s_sleep(1);
            v = LDG(p);
            if (++spins > (1u << 24)) { sync[1] = 1; ok = false; break; }     // ~seconds: never in a healthy run
        }
        return (unsigned)v;
    };

    if (g == 0) {
        // =============================================================== master: the reference's loop, heap included
        HxHeap<S16> h;
        h.lv = (double*)dyn_lds; h.gv = g_hval; h.lk16 = (unsigned short*)(h.lv + lc); h.lp16 = h.lk16 + n; h.gk = g_hkey; h.gp = g_hpos; h.lc = lc; h.size = n - 1;
        for (int i = tid; i < n - 1; i += HX_T) { h.setV(i, md[i]); h.setK(i, i); h.setP(i, i); }          // cl.cpp:80-91
        __syncthreads();
        if (tid == 0) for (int i = h.size / 2; i >= 0; --i) hx_down(h, i);                                  // cl.cpp:94
        __syncthreads();
        // command record: every thread knows the words (LDS), wave 0 stores them; the caller has drained the master's own stores (nb / md of
        // the rows it decided) before
        auto send = [&](int op, int x, int y, double d, int nx, int ny, int tx, int ty, int k) {
            ++seq;
            if (wv == 0 && lane < HX_CMDW) {
                const unsigned w = lane == 0 ? (unsigned)op : lane == 1 ? (unsigned)x : lane == 2 ? (unsigned)y : lane == 3 ? (unsigned)__double2loint(d) : lane == 4 ? (unsigned)__double2hiint(d)
                                 : lane == 5 ? (unsigned)nx : lane == 6 ? (unsigned)ny : lane == 7 ? (unsigned)tx : lane == 8 ? (unsigned)ty : (unsigned)k;
                STX<ONEX>(&cmd[lane], ((MwGran)seq << 32) | (MwGran)w);
            }
        };
        auto recv = [&]() -> bool {            // the G replies to command `seq` -> s_rep
            bool ok = true;
            for (int idx = tid; idx < G * HX_REPW; idx += HX_T) {
                const int sl = idx / HX_REPW, wd = idx - sl * HX_REPW;
                s_rep[sl][wd] = poll(&rep[(size_t)sl * 8 + wd], seq, ok);
            }
            return __syncthreads_and(ok ? 1 : 0) != 0;
        };
        auto fold_nn = [&]() -> MinIdx {       // lowest value, lowest column among equals: what a sequential scan in ascending j finds first
            MinIdx q; q.v = INFINITY; q.i = -1;
            for (int u = tid; u < G; u += HX_T) {
                MinIdx c; c.i = (int)s_rep[u][1];
                c.v = __hiloint2double((int)s_rep[u][3], (int)s_rep[u][2]);
                q = better(q, c);
            }
            q = wave_min(q);
            __syncthreads();
            if (lane == 0) sh[wv] = q;
            __syncthreads();
            MinIdx r = sh[0];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) r = better(r, sh[w2]);
            return r;
        };
        for (int k = 0; k < n - 1; ++k) {
            int x = 0, y = 0; double dist = 0.0;
            for (int guard = 0; guard < n - k; ++guard) {                                                // cl.cpp:323
                if (tid == 0) {
                    const int hx = h.K(0); const double hd = h.V(0); const int hy = LDG(&nb[hx]);        // get_min
                    const int tx = tyv[hx], ty = hy >= 0 ? tyv[hy] : -1;
                    s_x = hx; s_y = hy; s_d = hd; s_tx = tx; s_ty = ty;
                    s_ok = (hy >= 0) && (hd == LDG(tx >= ty ? &D[(int64_t)hx * N + hy] : &D[(int64_t)hy * N + hx]));   // cl.cpp:329
                }
                __syncthreads();
                x = s_x; y = s_y; dist = s_d;
                const int ok = s_ok, tx = s_tx;
                __syncthreads();
                if (ok) break;
                // stale candidate: row x's true nearest neighbour (cl.cpp:333-338), every worker scans its own columns
                if (g == 0 && tid == 0) sync[2] += 1;
                send(HX_OP_SCAN, x, 0, 0.0, 0, 0, tx, 0, k);
                if (!recv()) return;
                const MinIdx q = fold_nn();
                y = q.i; dist = (q.i < 0) ? (double)INFINITY : q.v;
                if (tid == 0) {
                    STX<ONEX>(&nb[x], y); STX<ONEX>(&md[x], dist);
                    hx_change(h, x, dist);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
            }
            if (y < 0) { if (tid == 0) { Z[(size_t)k * 4 + 3] = NAN; sync[1] = 1; } send(HX_OP_QUIT, 0, 0, 0.0, 0, 0, 0, 0, 0); return; }   // cannot happen while two clusters are active
            if (dist > stop_above) {                 // the caller continues from here with the cooperative kernel (run_linkage: duplicates merged, the rest is tie-free)
                if (tid == 0) sync[3] = (unsigned)k;
                send(HX_OP_QUIT, 0, 0, 0.0, 0, 0, 0, 0, 0);
                return;
            }
            if (tid == 0) { s_nx = size[x]; s_ny = size[y]; s_tx = tyv[x]; s_ty = tyv[y]; }
            __syncthreads();
            const int nx = s_nx, ny = s_ny, txm = s_tx, tym = s_ty;
            const bool last = (k == n - 2);
            if (!last) send(HX_OP_MERGE, x, y, dist, nx, ny, txm, tym, k);
            if (tid == 0) {
                hx_swap(h, 0, h.size - 1); h.size -= 1; hx_down(h, 0);                                   // remove_min, cl.cpp:101-105 (while the workers run their pass)
                int ix = cid[x], iy = cid[y];
                if (ix > iy) { const int t = ix; ix = iy; iy = t; }
                Z[(size_t)k * 4 + 0] = (double)ix; Z[(size_t)k * 4 + 1] = (double)iy;
                Z[(size_t)k * 4 + 2] = dist;       Z[(size_t)k * 4 + 3] = (double)(nx + ny);
                size[x] = 0; size[y] = nx + ny; cid[y] = n + k; tyv[y] = k;
            }
            if (last) break;
            if (!recv()) return;
            // change_value(z, D[z,y]) for the rows whose bound dropped, ascending z (cl.cpp:381-392): the workers' lists in worker order
            if (tid == 0) { int a = 0; for (int u = 0; u < G; ++u) { s_pre[u] = a; a += (int)s_rep[u][0]; } s_pre[G] = a; }
            __syncthreads();
            const int total = s_pre[G];
            for (int b0 = 0; b0 < total; b0 += HX_STAGE) {
                const int cnt = total - b0 < HX_STAGE ? total - b0 : HX_STAGE;
                for (int e = tid; e < cnt; e += HX_T) {
                    const int ge = b0 + e;
                    int lo = 0, hi = G - 1;                       // worker whose list holds entry ge: the last u with s_pre[u] <= ge
                    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pre[mid] <= ge) lo = mid; else hi = mid - 1; }
                    const size_t src = (size_t)lo * cap + (ge - s_pre[lo]);
                    st_z[e] = LDG(&chg_z[src]); st_v[e] = LDG(&chg_v[src]);
                }
                __syncthreads();
                // change_value for every staged row, in list order.  A decrease whose new value is still >= its parent's moves nothing (cl.cpp:44-51:
                // the sift-up loop ends at once); a parent's value only ever DROPS while the list is worked off, so an entry that passes that test against
                // the heap as it stands passes it whenever its turn comes, and a run of such entries can be written at once.  Wave 0 takes 64 entries at a
                // time: the lanes in front of the first entry that WOULD move write their values together, that entry is sifted by its lane alone,
                // and the rest of the chunk is looked at again (on clustered data most bounds drop by a little and stay where they are).
                if (wv == 0) {
                    int e0 = 0;
                    while (e0 < cnt) {
                        const int e = e0 + lane;
                        const bool valid = e < cnt;
                        int idx = 0; double v = 0.0; bool stay = false;
                        if (valid) {
                            idx = h.P(st_z[e]); v = st_v[e];
                            stay = idx == 0 || !(h.V((idx - 1) >> 1) > v);
                        }
                        const unsigned long long mv = __ballot(valid && !stay);
                        const int f = mv ? __builtin_ctzll(mv) : 64;               // first lane whose entry moves
                        if (valid && lane < f) h.setV(idx, v);
                        __builtin_amdgcn_wave_barrier();
                        if (f < 64) {
                            if (lane == f) hx_decrease(h, st_z[e], st_v[e]);
                            e0 += f + 1;
                        } else e0 += 64;
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this pass's heap writes are done before the next pass reads the heap
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                __syncthreads();
            }
            if (y < n - 1) {                                                                             // cl.cpp:395-404
                const MinIdx q = fold_nn();
                if (tid == 0 && q.i >= 0) { STX<ONEX>(&nb[y], q.i); STX<ONEX>(&md[y], q.v); hx_change(h, y, q.v); }
            }
            if (tid == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // nb / md of row y have landed before the next command is visible
            __syncthreads();
        }
        if (tid == 0) sync[3] = (unsigned)(n - 1);
        send(HX_OP_QUIT, 0, 0, 0.0, 0, 0, 0, 0, 0);
        return;
    }

    // =================================================================== workers
    const int wk = g - 1;
    const int colsB = cap;
    const int z0 = wk * colsB;
    const int nown = n - z0 < colsB ? (n - z0 > 0 ? n - z0 : 0) : colsB;
    const int nu = (nown + HX_T - 1) / HX_T;
    const int zsafe = z0 < n ? z0 : 0;
    int zc[HX_U], c_ty[HX_U]; bool c_dead[HX_U];
#pragma unroll
    for (int u = 0; u < HX_U; ++u) { const int p = tid + u * HX_T; zc[u] = p < nown ? z0 + p : -1; c_ty[u] = -1; c_dead[u] = zc[u] < 0; }
    auto reply = [&](unsigned w0, unsigned w1, double v) {
        if (wv == 0 && lane < HX_REPW) {
            const unsigned w = lane == 0 ? w0 : lane == 1 ? w1 : lane == 2 ? (unsigned)__double2loint(v) : (unsigned)__double2hiint(v);
            STX<ONEX>(&rep[(size_t)wk * 8 + lane], ((MwGran)seq << 32) | (MwGran)w);
        }
    };
    auto block_nn = [&](MinIdx q) -> MinIdx {
        q = wave_min(q);
        __syncthreads();
        if (lane == 0) sh[wv] = q;
        __syncthreads();
        MinIdx r = sh[0];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) r = better(r, sh[w2]);
        return r;
    };
    for (;;) {
        // ---- the next command: every wave polls the record itself (lane w takes word w)
        ++seq;
        bool ok = true;
        unsigned wdv = 0;
        if (lane < HX_CMDW) wdv = poll(&cmd[lane], seq, ok);
        if (__ballot(!ok) != 0ull) return;
        const int op = __builtin_amdgcn_readlane((int)wdv, 0), x = __builtin_amdgcn_readlane((int)wdv, 1), y = __builtin_amdgcn_readlane((int)wdv, 2);
        const double dist = __hiloint2double(__builtin_amdgcn_readlane((int)wdv, 4), __builtin_amdgcn_readlane((int)wdv, 3));
        const int nx = __builtin_amdgcn_readlane((int)wdv, 5), ny = __builtin_amdgcn_readlane((int)wdv, 6);
        const int txm = __builtin_amdgcn_readlane((int)wdv, 7), tym = __builtin_amdgcn_readlane((int)wdv, 8), k = __builtin_amdgcn_readlane((int)wdv, 9);
        if (op == HX_OP_QUIT) return;
        if (op == HX_OP_SCAN) {
            // find_min_dist (cl.cpp:259-276) over this worker's columns above x: first strictly smaller in ascending j
            MinIdx q; q.v = INFINITY; q.i = -1;
            double v[HX_U]; bool in[HX_U];
#pragma unroll
            for (int u = 0; u < HX_U; ++u) {
                in[u] = false; v[u] = 0.0;
                if (u < nu) {
                    const int j = zc[u];
                    in[u] = j > x && !c_dead[u];
                    const int jl = in[u] ? j : zsafe;
                    v[u] = LDG(txm >= c_ty[u] || !in[u] ? &D[(int64_t)x * N + jl] : &D[(int64_t)jl * N + x]);      // entry {x, j} from the row written last
                }
            }
#pragma unroll
            for (int u = 0; u < HX_U; ++u) if (u < nu && in[u] && v[u] < q.v) { q.v = v[u]; q.i = zc[u]; }
            q = block_nn(q);
            reply(0u, (unsigned)q.i, q.v);
            continue;
        }
        // ---- merge (x, y) at height dist: the z loop of cl.cpp:361-392 for this worker's columns
        MinIdx q; q.v = INFINITY; q.i = -1;
        double dzx[HX_U], dzy[HX_U], mdz[HX_U], ndv[HX_U]; int nbz[HX_U]; bool act[HX_U], chg[HX_U];
#pragma unroll
        for (int u = 0; u < HX_U; ++u) {
            act[u] = false; chg[u] = false; ndv[u] = 0.0;
            if (u < nu) {
                const int z = zc[u];
                act[u] = z >= 0 && !c_dead[u] && z != x && z != y;
                const int zl = act[u] ? z : zsafe;
                dzx[u] = LDG(&D[(int64_t)x * N + zl]);
                dzy[u] = LDG(&D[(int64_t)y * N + zl]);
                const int zr = zl < n - 1 ? zl : n - 2;
                nbz[u] = LDG(&nb[zr]); mdz[u] = LDG(&md[zr]);
            }
        }
#pragma unroll
        for (int u = 0; u < HX_U; ++u) {
            if (u < nu && act[u]) {
                if (txm < c_ty[u]) dzx[u] = LDG(&D[(int64_t)zc[u] * N + x]);        // z's row was written after x's: the current {z, x} is there
                if (tym < c_ty[u]) dzy[u] = LDG(&D[(int64_t)zc[u] * N + y]);
            }
        }
#pragma unroll
        for (int u = 0; u < HX_U; ++u) {
            if (!(u < nu && act[u])) continue;
            const int z = zc[u];
            const double nd = lw_centroid(dzx[u], dzy[u], dist, nx, ny);                                 // cl.cpp:367
            STX<ONEX>(&D[(int64_t)y * N + z], nd);
            if (z < x && nbz[u] == x) STX<ONEX>(&nb[z], y);                                             // cl.cpp:374-378
            if (z < y && nd < mdz[u]) { STX<ONEX>(&nb[z], y); STX<ONEX>(&md[z], nd); chg[u] = true; ndv[u] = nd; }   // cl.cpp:381-392
            if (z > y && nd < q.v) { q.v = nd; q.i = z; }                                                // cl.cpp:395-404 (ascending z per thread)
        }
#pragma unroll
        for (int u = 0; u < HX_U; ++u) {
            if (zc[u] == x && zc[u] >= 0) c_dead[u] = true;
            if (zc[u] == y && zc[u] >= 0) c_ty[u] = k;
        }
        // the changed rows in ascending z: column p = tid + u * T, so the order is u-major, then wave, then lane
        unsigned long long bm[HX_U];
#pragma unroll
        for (int u = 0; u < HX_U; ++u) { bm[u] = __ballot(chg[u]); if (lane == 0) s_cnt[u][wv] = __popcll(bm[u]); }
        q = block_nn(q);                      // (its barriers also publish s_cnt)
        int base[HX_U], totalc = 0;
#pragma unroll
        for (int u = 0; u < HX_U; ++u)
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) { if (w2 == wv) base[u] = totalc; totalc += s_cnt[u][w2]; }
#pragma unroll
        for (int u = 0; u < HX_U; ++u)
            if (chg[u]) {
                const size_t dst = (size_t)wk * cap + base[u] + __popcll(bm[u] & ((1ull << lane) - 1ull));
                STX<ONEX>(&chg_z[dst], zc[u]); STX<ONEX>(&chg_v[dst], ndv[u]);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // distances, neighbours, bounds and the list have landed before the reply is visible
        __syncthreads();
        reply((unsigned)totalc, (unsigned)q.i, q.v);
    }
}

// ---------------------------------------------------------------- launcher (run_linkage, cluster.hip)
bool linkage_hx_fits(int64_t N, int workers)
{
    if (workers < 1 || workers > HX_GMAX || N < 3) return false;
    return (N + workers - 1) / workers <= (int64_t)HX_U * HX_T;
}
// D: full N x N matrix, nb / md: exact nearest neighbours above each row (k_pdist_sq + k_row_nn); size = 1, cid = iota, tyv = -1
int linkage_hx_run(sd_ctx* c, bool onex, int workers, double* D, int64_t N, int* cid, int* size, int* tyv, int* nb, double* md, double* d_Z, bool* stopped,
                   double stop_above, int64_t* merges_done, bool* launched)
{
    *stopped = false;
    if (merges_done) *merges_done = 0;
    if (launched) *launched = false;
    const int G = workers;
    const int cap = (int)((N + G - 1) / G);
    // dynamic LDS: 16-bit keys and positions of all entries (n <= 65 535) + as many heap values as fit beside them, at most levels 0-12
    // (option linkage_hx_wide: the 32-bit key / position form, which jobs above 65 535 rows take, on any size -- tests)
    // The 16-bit form keeps 4 N bytes of keys and positions in LDS beside ~27 KB of static LDS: it is taken only while those leave room for at least
    // 1 024 heap values inside the budget (N <= 31 232); from there to 65 535 rows the request would exceed the CU's 160 KB and every cooperative launch
    // would be refused (a 4-hour job with ties fell through to the one-workgroup heap kernel) -- those sizes take the 32-bit form.
    const size_t budget = 130 * 1024;
    const bool s16 = N <= 65535 && !c->linkage_hx_wide && (size_t)4 * N + 8 * 1024 <= budget;
    const size_t kp_bytes = s16 ? (size_t)4 * N : 0;
    int64_t lc64 = kp_bytes < budget ? (int64_t)((budget - kp_bytes) / 8) : 0;
    if (lc64 > HX_LDS_HEAP) lc64 = HX_LDS_HEAP;
    if (lc64 > N - 1) lc64 = N - 1;
    const int lc = (int)lc64;
    WS(c, double, hv, "cl_hval", N);
    WS(c, int, hk, "cl_hkey", N);
    WS(c, int, hp, "cl_hpos", N);
    WS(c, MwGran, cmd, "hx_cmd", 16 + (int64_t)8 * G);
    WS(c, int, chz, "hx_chg_z", (int64_t)G * cap);
    WS(c, double, chv, "hx_chg_v", (int64_t)G * cap);
    WS(c, unsigned, sync, "cl_sync", 32 + 16 * 256);
    HIPCHK(c, hipMemsetAsync(cmd, 0, (size_t)(16 + 8 * G) * sizeof(MwGran), c->stream));
    HIPCHK(c, hipMemsetAsync(sync, 0, (32 + 16 * 256) * sizeof(unsigned), c->stream));
    MwGran* rep = cmd + 16;
    const size_t dyn = (((size_t)lc * 8 + kp_bytes) + 15) & ~(size_t)15;
    const void* f = onex ? (s16 ? (const void*)k_linkage_hx<true, true> : (const void*)k_linkage_hx<true, false>)
                         : (s16 ? (const void*)k_linkage_hx<false, true> : (const void*)k_linkage_hx<false, false>);
    (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    (void)hipGetLastError();
    int n_i = (int)N, cap_i = cap, G_i = G, lc_i = lc;
    void* args[] = {&D, &n_i, &cid, &size, &tyv, &nb, &md, &d_Z, &hv, &hk, &hp, &cmd, &rep, &chz, &chv, &sync, &cap_i, &G_i, &lc_i, &stop_above};
    hipError_t le;
    {
        ProfScope ps(c, "linkage_hx", 0, 24.0 * (double)N * (double)N);
        le = hipLaunchCooperativeKernel(f, dim3(onex ? 8 * (G + 1) : G + 1), dim3(HX_T), args, dyn, c->stream);
    }
    if (le != hipSuccess) { (void)hipGetLastError(); *stopped = true; return SD_OK; }
    if (launched) *launched = true;
    unsigned h[8] = {0};
    HIPCHK(c, hipMemcpyAsync(h, sync, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stats["linkage_hx_stale_scans"].flops += (double)h[2];
    if (h[1]) *stopped = true;                  // a hand-off timed out (or, one XCD: too few workgroups found themselves on XCC 0)
    if (merges_done) *merges_done = (int64_t)h[3];
    return SD_OK;
}
