// cluster.hip -- agglomerative clustering of the speaker embeddings on the GPU
// (compiled with -ffp-contract=off: every fp64 result is bit-identical to the reference's
// x86 build, which decides merge order and therefore speaker numbering).
//   a10 filter_embeddings                      sd.cpp:2214-2259
//   a11 Cluster::cluster (normalise, size split, small->large reassign, renumber)  sd.cpp:2300-2422
//   a12 Clustering::linkage = euclidean pdist + scipy-generic centroid linkage     cl.cpp:289-440
//   a13 Clustering::fcluster(criterion=distance)                                   cl.cpp:121-232, 442-457
//   a14 assign_embeddings (centroids, cosine cdist, argmax)                        sd.cpp:2119-2212
//
// Design: the condensed fp64 distance matrix (N(N-1)/2 doubles, 1.86 GB at N = 21 573) stays
// resident in HBM.  k_pdist builds it from LDS tiles with the reference's sequential
// per-pair summation order.  The N-1 dependent merges run in one of two persistent kernels:
// k_linkage_mw (N >= 1500) replaces the reference's binary heap by a parallel arg-min over the
// per-row lower bounds on G co-resident workgroups and takes a merge only while the closest pair
// is unique; k_linkage_heap (small N, and the fallback at the first exact tie) is one workgroup
// that replays the reference's heap operation by operation, so Z is bit-identical to the
// reference for every input, ties included.  fcluster is O(N) pointer chasing and runs on the host.
#include "common.h"
#include "linkage_dev.h"
#include <algorithm>
#include <cfloat>
#include <cmath>

// ---------------------------------------------------------------- row gather + L2 normalise (a10/a11)
// Xout[i] = X[tidx[i]] (un-normalised copy), Xn[i] = row / (double)(float)sqrt(sum x^2)
// Helper::L2Norm returns float (sd.cpp:332-340): the norm is rounded to f32 before the divide.
__global__ void k_gather_normalize(const double* __restrict__ X, const int* __restrict__ tidx, int64_t N, int d,
                                   double* __restrict__ Xout, double* __restrict__ Xn)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double* r = X + (size_t)tidx[i] * d;
    double s = 0.0;
    for (int q = 0; q < d; ++q) s += r[q] * r[q];
    const double nrm = (double)(float)sqrt(s);
    for (int q = 0; q < d; ++q) {
        const double v = r[q];
        if (Xout) Xout[(size_t)i * d + q] = v;
        Xn[(size_t)i * d + q] = (nrm != 0.0) ? v / nrm : v;
    }
}

// ---------------------------------------------------------------- k_pdist (cl.cpp:408-431)
#define PT 64
__global__ __launch_bounds__(256) void k_pdist(const double* __restrict__ X, int64_t N, int d, double* __restrict__ D)
{
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj < ti) return;
    __shared__ double Xi[PT][33], Xj[PT][33];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
    for (int q0 = 0; q0 < d; q0 += 32) {
        for (int e = tid; e < PT * 32; e += 256) {
            const int r = e >> 5, q = e & 31;
            int64_t gi = (int64_t)ti * PT + r; if (gi > N - 1) gi = N - 1;
            int64_t gj = (int64_t)tj * PT + r; if (gj > N - 1) gj = N - 1;
            const bool in = (q0 + q) < d;
            Xi[r][q] = in ? X[(size_t)gi * d + q0 + q] : 0.0;
            Xj[r][q] = in ? X[(size_t)gj * d + q0 + q] : 0.0;
        }
        __syncthreads();
        for (int q = 0; q < 32; ++q) {          // sequential in q: same summation order as the reference
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = Xi[ty + 16 * u][q]; b[u] = Xj[tx + 16 * u][q]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) { const double df = a[u] - b[v]; acc[u][v] += df * df; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t i = (int64_t)ti * PT + ty + 16 * u, j = (int64_t)tj * PT + tx + 16 * v;
            if (i < j && j < N) D[cidx(N, i, j)] = sqrt(acc[u][v]);
        }
}

// the same distances as the full N x N square (k_linkage_mw<*, true>): tile (ti, tj), ti <= tj, is computed once and written twice, the
// mirror through an LDS transpose so that both writes are row-contiguous.  D[i][j] and D[j][i] are the same bits; the diagonal is 0.
__global__ __launch_bounds__(256) void k_pdist_sq(const double* __restrict__ X, int64_t N, int d, double* __restrict__ D)
{
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj < ti) return;
    __shared__ double Xi[PT][33], Xj[PT][33];
    __shared__ double Tt[PT][PT + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
    for (int q0 = 0; q0 < d; q0 += 32) {
        for (int e = tid; e < PT * 32; e += 256) {
            const int r = e >> 5, q = e & 31;
            int64_t gi = (int64_t)ti * PT + r; if (gi > N - 1) gi = N - 1;
            int64_t gj = (int64_t)tj * PT + r; if (gj > N - 1) gj = N - 1;
            const bool in = (q0 + q) < d;
            Xi[r][q] = in ? X[(size_t)gi * d + q0 + q] : 0.0;
            Xj[r][q] = in ? X[(size_t)gj * d + q0 + q] : 0.0;
        }
        __syncthreads();
        for (int q = 0; q < 32; ++q) {          // sequential in q: same summation order as the reference (and as k_pdist)
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = Xi[ty + 16 * u][q]; b[u] = Xj[tx + 16 * u][q]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) { const double df = a[u] - b[v]; acc[u][v] += df * df; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int li = ty + 16 * u, lj = tx + 16 * v;
            const int64_t i = (int64_t)ti * PT + li, j = (int64_t)tj * PT + lj;
            const double dv = (i == j) ? 0.0 : sqrt(acc[u][v]);
            Tt[li][lj] = dv;
            if (i < N && j < N && (ti != tj || i <= j)) D[(size_t)i * N + j] = dv;
        }
    __syncthreads();
    for (int e = tid; e < PT * PT; e += 256) {
        const int lj = e >> 6, li = e & 63;                 // consecutive threads: consecutive i = consecutive addresses of row j
        const int64_t i = (int64_t)ti * PT + li, j = (int64_t)tj * PT + lj;
        if (i < N && j < N && i < j) D[(size_t)j * N + i] = Tt[li][lj];
    }
}

__global__ __launch_bounds__(256) void k_row_nn(const double* __restrict__ D, int64_t n, int* __restrict__ nb, double* __restrict__ md, double* __restrict__ md2, int square)
{
    const int lane = threadIdx.x & 63;
    const int64_t x = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (x >= n - 1) return;
    const double* row = D + (square ? x * n + x + 1 : cidx(n, x, x + 1));
    const int64_t cnt = n - 1 - x;
    Min2 m; m.v = INFINITY; m.i = -1; m.v2 = INFINITY;
    for (int64_t t = lane; t < cnt; t += 64) min2_acc(m, row[t], (int)(x + 1 + t));
    m = wave_min2(m);
    if (lane == 0) { nb[x] = m.i; md[x] = (m.i < 0) ? INFINITY : m.v; if (md2) md2[x] = (m.i < 0) ? INFINITY : m.v2; }
}

// the same for a square matrix some merges into the job (run_linkage: k_linkage_hx merged the duplicates, k_linkage_rg continues): only clusters still there
// (size != 0), and entry {x, j} from the row that was written last
__global__ __launch_bounds__(256) void k_row_nn_mid(const double* __restrict__ D, int64_t n, const int* __restrict__ size, const int* __restrict__ ty,
                                                    int* __restrict__ nb, double* __restrict__ md, double* __restrict__ md2)
{
    const int lane = threadIdx.x & 63;
    const int64_t x = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (x >= n - 1) return;
    Min2 m; m.v = INFINITY; m.i = -1; m.v2 = INFINITY;
    if (size[x] != 0) {
        const int tx = ty[x];
        for (int64_t j = x + 1 + lane; j < n; j += 64)
            if (size[j] != 0) min2_acc(m, tx >= ty[j] ? D[x * n + j] : D[j * n + x], (int)j);
    }
    m = wave_min2(m);
    if (lane == 0) { nb[x] = m.i; md[x] = (m.i < 0) ? INFINITY : m.v; md2[x] = (m.i < 0) ? INFINITY : m.v2; }
}


// nearest active neighbour above row x, scanned by `nthreads` threads with U loads in flight per thread
// (a plain strided loop keeps one load outstanding and is latency bound: ~1 us per element per thread)
template <int U>
__device__ __forceinline__ MinIdx scan_row_nn(const double* __restrict__ D, const int* __restrict__ size, int64_t N, int n, int x,
                                              int first, int stride)
{
    MinIdx q; q.v = INFINITY; q.i = -1;
    const double* row = D + cidx(N, x, (int64_t)x + 1) - (x + 1);       // row[j] = D[x, j]
    for (int j0 = x + 1 + first; j0 < n; j0 += stride * U) {
        double v[U]; int sz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * stride;
            const int jc = j < n ? j : n - 1;
            v[u] = row[jc]; sz[u] = size[jc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * stride;
            if (j < n && sz[u] != 0 && v[u] < q.v) { q.v = v[u]; q.i = j; }
        }
    }
    return q;
}

#define LT 1024
__device__ __forceinline__ MinIdx block_min(MinIdx m, MinIdx* sh)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    m = wave_min(m);
    __syncthreads();                 // sh may still be read from the previous use
    if (lane == 0) sh[w] = m;
    __syncthreads();
    MinIdx r = sh[0];
#pragma unroll
    for (int k = 1; k < LT / 64; ++k) r = better(r, sh[k]);
    return r;
}

// ---------------------------------------------------------------- k_linkage_heap : persistent single workgroup, the reference's heap included
// fast_linkage (cl.cpp:289-406) with its indexed binary min-heap (cl.cpp:28-119) kept bit for bit: thread 0 replays every
// Heap operation the reference performs, in the reference's order -- heapify (cl.cpp:94), get_min / change_value in the lazy
// validation loop (cl.cpp:323-339), remove_min (cl.cpp:340), change_value for the rows whose lower bound dropped IN ASCENDING z
// (cl.cpp:381-392), change_value for row y (cl.cpp:395-404) -- while the O(n) parts of a merge (Lance-Williams update, neighbour
// patches, nearest-neighbour scans) run on all threads.  Which of several rows with EXACTLY equal lower bounds the heap hands
// out first depends on the whole history of its array, so nothing short of replaying it reproduces the reference's merge
// order on data with ties (duplicate embeddings, lattice points); with it Z is bit-identical for any input.
// The heap (values / key_by_index / index_by_key) lives in LDS up to HEAP_LDS entries, in global memory above.
// The rows whose bound dropped are collected in an LDS bitmap and drained in ascending order by wave 0.
// (no __restrict__: every array here is written by one thread and re-read by others across barriers)
#define HEAP_LDS 2048
struct HeapRef { double* val; int* key; int* pos; int size; };
__device__ __forceinline__ void hp_swap(HeapRef& h, int a, int b)                          // cl.cpp:70-78
{
    const double va = h.val[a], vb = h.val[b];
    h.val[a] = vb; h.val[b] = va;
    const int ka = h.key[a], kb = h.key[b];
    h.key[a] = kb; h.key[b] = ka;
    h.pos[ka] = b; h.pos[kb] = a;
}
__device__ __forceinline__ void hp_down(HeapRef& h, int idx)                                // cl.cpp:53-68
{
    int ch = 2 * idx + 1;
    while (ch < h.size) {
        if (ch + 1 < h.size && h.val[ch + 1] < h.val[ch]) ch += 1;
        if (h.val[idx] > h.val[ch]) { hp_swap(h, idx, ch); idx = ch; ch = 2 * idx + 1; }
        else break;
    }
}
__device__ __forceinline__ void hp_up(HeapRef& h, int idx)                                  // cl.cpp:44-51
{
    int par = (idx - 1) >> 1;
    while (idx > 0 && h.val[par] > h.val[idx]) { hp_swap(h, idx, par); idx = par; par = (idx - 1) >> 1; }
}
__device__ __forceinline__ void hp_change(HeapRef& h, int key, double v)                    // cl.cpp:108-117
{
    const int idx = h.pos[key];
    const double old = h.val[idx];
    h.val[idx] = v;
    if (v < old) hp_up(h, idx); else hp_down(h, idx);
}

__global__ __launch_bounds__(LT) void k_linkage_heap(double* D, int n, int* size, int* cid, int* nb, double* md, double* Z,
                                                      double* g_hval, int* g_hkey, int* g_hpos)
{
    extern __shared__ unsigned changed[];                 // bitmap of the rows whose bound dropped in this merge
    __shared__ MinIdx sh[LT / 64];
    __shared__ int s_ok, s_x, s_y;
    __shared__ double s_dist;
    __shared__ double s_hv[HEAP_LDS];
    __shared__ int s_hk[HEAP_LDS], s_hp[HEAP_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t N = n;
    const bool in_lds = (n - 1) <= HEAP_LDS;
    HeapRef h;
    h.val = in_lds ? s_hv : g_hval; h.key = in_lds ? s_hk : g_hkey; h.pos = in_lds ? s_hp : g_hpos; h.size = n - 1;
    const int nwords = (n + 31) / 32;
    for (int i = tid; i < nwords; i += LT) changed[i] = 0u;
    for (int i = tid; i < n - 1; i += LT) { h.val[i] = md[i]; h.key[i] = i; h.pos[i] = i; }         // cl.cpp:80-91
    __syncthreads();
    if (tid == 0) for (int i = h.size / 2; i >= 0; --i) hp_down(h, i);                                // cl.cpp:94
    __syncthreads();
    for (int k = 0; k < n - 1; ++k) {
        int x = 0, y = 0; double dist = 0.0;
        for (int guard = 0; guard < n - k; ++guard) {                                                // cl.cpp:323
            if (tid == 0) {
                const int hx = h.key[0]; const double hd = h.val[0]; const int hy = nb[hx];          // get_min
                s_x = hx; s_y = hy; s_dist = hd;
                s_ok = (hy >= 0) && (hd == D[cidx(N, hx, hy)]);                                     // cl.cpp:329
            }
            __syncthreads();
            x = s_x; y = s_y; dist = s_dist;
            const int ok = s_ok;
            __syncthreads();
            if (ok) break;
            // stale candidate: row x's true nearest neighbour (cl.cpp:333-338)
            MinIdx q = scan_row_nn<4>(D, size, N, n, x, tid, LT);
            q = block_min(q, sh);
            y = q.i; dist = (q.i < 0) ? (double)INFINITY : q.v;
            if (tid == 0) { nb[x] = y; md[x] = dist; hp_change(h, x, dist); }
            __syncthreads();
        }
        if (tid == 0) { hp_swap(h, 0, h.size - 1); h.size -= 1; hp_down(h, 0); }                      // remove_min, cl.cpp:101-105
        if (y < 0) { if (tid == 0) Z[(size_t)k * 4 + 3] = NAN; return; }                             // cannot happen while two clusters are active
        const int nx = size[x], ny = size[y];
        __syncthreads();
        if (tid == 0) {
            int ix = cid[x], iy = cid[y];
            if (ix > iy) { const int t = ix; ix = iy; iy = t; }
            Z[(size_t)k * 4 + 0] = (double)ix; Z[(size_t)k * 4 + 1] = (double)iy;
            Z[(size_t)k * 4 + 2] = dist;       Z[(size_t)k * 4 + 3] = (double)(nx + ny);
            size[x] = 0; size[y] = nx + ny; cid[y] = n + k;
        }
        __syncthreads();
        for (int z0 = tid; z0 < n; z0 += LT * 4) {                   // 4 rows per thread: all their loads issued together
            double dzx[4], dzy[4], mdz[4]; int sz[4], nbz[4]; int64_t izy[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int z = z0 + u * LT;
                const int zc = (z < n && z != y) ? z : ((y > 0) ? 0 : 1);        // any valid row other than y
                izy[u] = cidx(N, zc, y);
                sz[u] = size[zc];
                dzx[u] = (zc == x) ? 0.0 : D[cidx(N, zc, x)];
                dzy[u] = D[izy[u]];
                const int zr = zc < n - 1 ? zc : n - 2;
                nbz[u] = nb[zr]; mdz[u] = md[zr];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int z = z0 + u * LT;
                if (z >= n || z == y || sz[u] == 0) continue;
                const double nd = lw_centroid(dzx[u], dzy[u], dist, nx, ny);           // cl.cpp:367
                D[izy[u]] = nd;
                if (z < x && nbz[u] == x) nb[z] = y;                                    // cl.cpp:374-378
                if (z < y && nd < mdz[u]) { nb[z] = y; md[z] = nd; atomicOr(&changed[z >> 5], 1u << (z & 31)); }   // cl.cpp:381-392
            }
        }
        __syncthreads();
        // change_value(z, D[z,y]) for the rows whose bound dropped, ascending z (cl.cpp:381-392): wave 0 walks the bitmap
        if (tid < 64) {
            for (int w0 = 0; w0 < nwords; w0 += 64) {
                const int wi = w0 + lane;
                const unsigned wd = wi < nwords ? changed[wi] : 0u;
                unsigned long long live = __ballot(wd != 0u);
                if (wd != 0u) changed[wi] = 0u;
                while (live) {
                    const int l = __builtin_ctzll(live);
                    live &= live - 1;
                    unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)wd, l);
                    while (bits) {
                        const int z = (w0 + l) * 32 + __builtin_ctz(bits);
                        bits &= bits - 1;
                        if (lane == 0) hp_change(h, z, md[z]);
                    }
                }
            }
        }
        __syncthreads();
        if (y < n - 1) {                                                              // cl.cpp:395-404
            MinIdx q = scan_row_nn<4>(D, size, N, n, y, tid, LT);
            q = block_min(q, sh);
            if (tid == 0 && q.i >= 0) { nb[y] = q.i; md[y] = q.v; hp_change(h, y, q.v); }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- k_linkage_mw : the same algorithm on G co-resident workgroups
// Rows are owned round-robin (row z belongs to workgroup z % G).  Per merge every workgroup
//   1. applies the size bookkeeping of the merge to its own view (identical stores from all workgroups),
//   2. runs the Lance-Williams update for its rows, folding the nearest-neighbour search of row y into the
//      same pass (the new D[z,y], z > y, are in registers: row y is never re-read),
//   3. computes its local arg-min of the lower bounds and publishes {NN(y) partial, arg-min} in a slot,
//   4. exchanges slots with the others: the slot is a set of tagged 8-byte granules {payload word, round number} that
//      the readers poll directly -- no counter, no flag, no fence (everything the workgroups hand each other travels in
//      agent-scope `sc1` loads / stores, see LDG / STG below; placement independent); every workgroup reduces the G
//      slots itself, so all take the same decision without a broadcast.
// A stale candidate (cl.cpp:329-338) costs one extra round (see the kernel's own header below).
// Used from N = 1500 up, where one CU's memory pipeline is the bottleneck.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_linkage_mw's fence-free slot exchange (sc1 write-through stores, sc1 loads, 8-byte tagged granules) is written for gfx950 only"
#endif
#define MWT 256
#define MWT_MAX 1024
// arg-min candidate that carries its neighbour and its flags along through the reductions.
// fresh bit 0: the bound is exact (== D[i, y]); bit 1 (CAND_TIE): some OTHER row holds exactly the same bound -- the case in
// which the reference's heap, not the value, decides who comes first (see k_linkage_heap)
struct Cand { double v; int i; int y; int fresh; };

__device__ __forceinline__ bool mw_barrier(unsigned* counter, unsigned target, unsigned* timeout_flag)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        (void)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 26)) { *timeout_flag = 1; ok = false; break; }      // ~seconds: never in a healthy run
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return ok;
}

__device__ __forceinline__ MinIdx block_min_t(MinIdx m, MinIdx* sh, int nwaves)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    m = wave_min(m);
    __syncthreads();
    if (lane == 0) sh[w] = m;
    __syncthreads();
    MinIdx r = sh[0];
    for (int k = 1; k < nwaves; ++k) r = better(r, sh[k]);
    return r;
}
__device__ __forceinline__ Cand cbetter(Cand a, Cand b)
{
    if (b.i < 0) return a;
    if (a.i < 0) return b;
    if (b.v < a.v) return b;
    if (b.v == a.v) {
        Cand r = (b.i < a.i) ? b : a;
        if (a.i != b.i && a.v < INFINITY) r.fresh |= CAND_TIE | ((a.fresh | b.fresh) & CAND_TIE);
        return r;
    }
    return a;
}
// sequential accumulation of one row into a thread's running candidate (same rules as cbetter)
__device__ __forceinline__ void cand_acc(Cand& m, double v, int z, int y, int fresh)
{
    if (m.i < 0 || v < m.v) { m.v = v; m.i = z; m.y = y; m.fresh = fresh; }
    else if (v == m.v) {
        const int tie = (v < INFINITY) ? CAND_TIE : 0;
        if (z < m.i) { m.i = z; m.y = y; m.fresh = fresh | tie | (m.fresh & CAND_TIE); }
        else m.fresh |= tie;
    }
}
// same two moves for candidates.  cbetter's tie rule survives: the winner is the lowest row among the lanes at the minimum, and
// CAND_TIE is raised when a DIFFERENT row sits at the same finite bound (flags inherited from the losers add nothing to that)
__device__ __forceinline__ Cand wave_min_c(Cand m)
{
    const bool has = m.i >= 0;
    const double vmin = wave_min_d(has ? m.v : (double)INFINITY);
    const bool at = has && m.v == vmin;
    const unsigned long long mask = __ballot(at);
    Cand r; r.v = INFINITY; r.i = -1; r.y = -1; r.fresh = 0;
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
    r.fresh = __builtin_amdgcn_readlane(m.fresh, l) | extra;
    return r;
}
__device__ __forceinline__ Cand block_min_c(Cand m, Cand* sh, int nwaves)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    m = wave_min_c(m);
    __syncthreads();
    if (lane == 0) sh[w] = m;
    __syncthreads();
    Cand r; r.v = INFINITY; r.i = -1; r.y = -1; r.fresh = 0;
    if (lane < nwaves) r = sh[lane];
    return wave_min_c(r);                // every wave folds the per-wave winners itself
}
// the two reductions of a merge round (NN(y) partial and local arg-min) through ONE LDS exchange
__device__ __forceinline__ void block_min_qc(Min2& q, Cand& m, Min2* shq, Cand* shc, int nwaves)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    q = wave_min2(q);
    m = wave_min_c(m);
    __syncthreads();
    if (lane == 0) { shq[w] = q; shc[w] = m; }
    __syncthreads();
    Min2 rq; rq.v = INFINITY; rq.i = -1; rq.v2 = INFINITY;
    Cand r; r.v = INFINITY; r.i = -1; r.y = -1; r.fresh = 0;
    if (lane < nwaves) { rq = shq[lane]; r = shc[lane]; }
    q = wave_min2(rq);
    m = wave_min_c(r);
}

// ---------------------------------------------------------------- k_linkage_mw : the cooperative kernel
//  * every workgroup keeps its owned ACTIVE rows as a compact list in LDS (swap-with-last removal), so the
//    Lance-Williams pass and the arg-min touch N-k rows at merge k instead of N (half the random HBM accesses);
//  * a stale candidate (cl.cpp:329-338) is not rescanned by its owner alone (one CU pulling a 1.4 MB row at
//    N = 172 773 takes ~60 us): all workgroups see the same published candidates, pick the same KR best stale rows
//    and each scans its share of the columns of every such row; the partial minima travel in the slots of the
//    one barrier the round needs anyway.  Refreshing up to KR near-top stale rows per round needs ~5x fewer rounds
//    than refreshing only the top one (measured 9.3 k vs 49 k rounds at N = 21 573).
#define KR 4
// A slot is SLOT_WORDS 8-byte granules {32-bit payload word, 32-bit round tag}: a reader that sees the tag of the round it
// waits for has the payload of that round (8-byte stores are single transactions), so publishing needs no separate
// "ready" flag and no counter -- the readers poll the granules themselves.  Words: 0-1 arg-min bound (double), 2 its row,
// 3 its neighbour, 4 freshness, 5-6 NN(y) partial (double), 7 its row; merge rounds: 8 "row x had a second pair at the merge height",
// 9-10 second value of the NN(y) partial; retry rounds: 8+5r.. refreshed-row partial r (minimum double, its row, second value double).
#define SLOT_WORDS 32

// SQ form (k_linkage_mw<*, true>): the distance matrix is the full N x N square and a workgroup owns a contiguous range of COLUMNS.
// Only rows are ever read or written in bulk: a merge (x, y) reads rows x and y and writes row y, every workgroup its own column range,
// coalesced (the condensed form touches one 64-byte line per entry for the half of the entries that lie in a column: 150 000 scattered
// transactions per merge at N = 100 000, tools/tlb_probe2.hip).  Row y is NOT mirrored into column y.  Instead every cluster carries the
// index ty of the last merge that rewrote its row (-1: never), and the entry {a, b} is read from the row of the cluster with the larger
// ty -- the one written last, which holds the current value; with equal ty (two clusters that never were a merge's y) both rows still
// hold the pdist value.  A merge that involves a cluster older than a bystander z reads that one entry from row z (scattered); on
// clustered data (one growing cluster per speaker swallowing singletons) that is a handful of entries per merge.
template <bool ONEX, bool SQ>
__global__ __launch_bounds__(MWT_MAX) void k_linkage_mw(double* D, int n, int* size_all, int* cid, int* nb, double* md, const double* md2_init,
                                                         double* Z, MwGran* gran /*[2][G][SLOT_WORDS], zeroed*/,
                                                         unsigned* sync, int cap /*owned rows per workgroup, upper bound*/, int G)
{
    extern __shared__ __attribute__((aligned(16))) int dyn_lds[];
    // per owned row, 32 B of LDS: the active list and, beside every entry, the row's lower bound / neighbour / freshness.  The
    // owner is the only reader of these in the hot loops (local arg-min, Lance-Williams pass), so they never leave the CU; the
    // global md / nb copies are still written (row y's old bound is read by everybody) but not read back by the owner.
    // l_md2 is the second level of the bound: every active entry of the row OTHER than the neighbour's is >= l_md2.  While the
    // neighbour's distance stays <= l_md2 the bound is exact whatever the merge did to it, and a row only goes stale (and costs a
    // retry round when it reaches the top) after it has lost BOTH levels; with the reference's single lower bound (cl.cpp:323-339) a
    // row went stale every time the distance to its neighbour grew -- about half the rows per merge on clustered data, 0.46 retry
    // rounds per merge on the planted hour.  Same merges: the arg-min of exact values does not depend on how the bounds are kept.
    double* l_md = (double*)dyn_lds;                 // [cap] bound of act[p]
    double* l_md2 = l_md + cap;                      // [cap] lower bound of the row's entries other than the neighbour's
    int* act = (int*)(l_md2 + cap);                  // [cap] owned active rows, unordered.  SQ: l_ty[s] = ty of owned column z0 + s
    int* pos = act + cap;                            // [cap] pos[z / G] = index of owned row z in act
    int* l_nb = pos + cap;                           // [cap] neighbour of act[p]
    unsigned char* l_fr = (unsigned char*)(l_nb + cap);   // [cap] freshness of act[p]
    __shared__ Min2 sh[MWT_MAX / 64];
    __shared__ Cand shc[MWT_MAX / 64];
    __shared__ Min2 s_part[KR][MWT_MAX / 64];
    __shared__ unsigned s_words[MWT][SLOT_WORDS + 1];  // this round's slots of all workgroups, as received (+1: lane u reads word w of slot u -- a 128-byte row stride would put all lanes on two banks)
    __shared__ Cand s_cand[MWT + 1];        // published local bests of the G <= 256 workgroups (+ row y)
    __shared__ Min2 s_row[KR];
    __shared__ int s_L[2][KR];
    __shared__ int s_nL[2];
    __shared__ int s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int T = blockDim.x, NW = T >> 6;
    int g = blockIdx.x;
    if constexpr (ONEX) {
        // 8 G workgroups were launched; the first G that find themselves on XCC 0 take part (rank = ticket), the others leave.
        // Under round-robin dispatch exactly the G workgroups with blockIdx % 8 == 0 qualify; under any other dispatch too few
        // may arrive and the participants run into the poll timeout -- run_linkage then repeats the job with the multi-XCD form.
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
    int* size = size_all + (size_t)g * n * (SQ ? 2 : 1);       // this workgroup's private copy of the cluster sizes (SQ: followed by its copy of ty)
    int* tyv = size + n;                                       // SQ only
    int* l_ty = act;                                           // SQ only
    const int colsB = cap - 1;                                 // SQ: columns per workgroup; z0 = first owned column, nown = how many exist
    const int z0 = SQ ? g * colsB : 0;
    const int nown = SQ ? (n - z0 < colsB ? (n - z0 > 0 ? n - z0 : 0) : colsB) : 0;
    auto own = [&](int z) -> bool { return SQ ? (z >= z0 && z < z0 + colsB) : ((z % G) == g); };
    auto slot = [&](int z) -> int { return SQ ? z - z0 : pos[z / G]; };
    unsigned bar = 0;
    int par = 0, lp = 0;             // slot parity, refresh-list parity
    // receive round `bar` of every workgroup's slot (nw words each) into s_words; returns false on timeout
    auto consume = [&](int nw) -> bool {
        const MwGran* base = gran + (size_t)par * G * SLOT_WORDS;
        bool ok = true;
        for (int idx = tid; idx < G * nw; idx += T) {
            const int sl = idx / nw, wd = idx - sl * nw;
            const MwGran* p = base + (size_t)sl * SLOT_WORDS + wd;
            MwGran v = LDG(p);
            unsigned spins = 0;
            while ((unsigned)(v >> 32) != bar) {
                __builtin_amdgcn_s_sleep(1);
                v = LDG(p);
                if (++spins > (1u << 24)) { sync[1] = 1; ok = false; break; }     // ~seconds: never in a healthy run
            }
            s_words[sl][wd] = (unsigned)v;
        }
        return __syncthreads_and(ok ? 1 : 0) != 0;
    };
    MinIdx none; none.v = INFINITY; none.i = -1;
#ifdef SD_LINKAGE_STAMPS
    unsigned long long tS = __builtin_amdgcn_s_memrealtime(), acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned fix_lanes = 0, fix_waves = 0;
#define STAMP2(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); acc[i] += t_ - tS; tS = t_; } while (0)
#else
#define STAMP2(i) do { } while (0)
#endif

    // local arg-min over the owned active rows, skipping the rows being refreshed this round
    auto local_argmin = [&](int nL, const int* L) -> Cand {
        int ex[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) ex[r] = r < nL ? L[r] : -1;
        Cand m; m.v = INFINITY; m.i = -1; m.y = -1; m.fresh = 0;
        const int cnt = SQ ? nown : s_cnt;
        for (int p0 = tid; p0 < cnt; p0 += T * 4) {
            double v[4]; int zz[4], ny_[4]; unsigned char fr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * T;
                const int pc = p < cnt ? p : 0;
                fr[u] = l_fr[pc];
                int z = p < cnt ? (SQ ? ((fr[u] & 2) ? -1 : z0 + pc) : act[pc]) : -1;
                if (z >= n - 1) z = -1;
                zz[u] = z;
                v[u] = l_md[pc]; ny_[u] = l_nb[pc];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int z = zz[u];
                bool skip = z < 0;
#pragma unroll
                for (int r = 0; r < KR; ++r) skip |= (z == ex[r]);
                if (!skip) cand_acc(m, v[u], z, ny_[u], fr[u] & 1);
            }
        }
        return block_min_c(m, shc, NW);
    };
    // this workgroup's share of the columns of the nL rows in L: wave tasks (row r, sub-slice s)
    auto scan_rows = [&](int nL, const int* L, Min2* outv /*LDS [KR]*/) {
        if (nL <= 0) return;
        const int S = NW >= nL ? NW / nL : 1;
        for (int t = wv; t < nL * S; t += NW) {
            const int r = t % nL, sidx = t / nL;
            const int x = L[r];
            Min2 q; q.v = INFINITY; q.i = -1; q.v2 = INFINITY;
            if constexpr (SQ) {
                // this workgroup's own columns above x: entry {x, j} from the row written last
                const int txr = tyv[x];
                const int64_t jend = (int64_t)z0 + nown, step = (int64_t)S * 64;
                for (int64_t j0 = (x + 1 > z0 ? x + 1 : z0) + (int64_t)sidx * 64 + lane; j0 < jend; j0 += step * 4) {
                    double v[4]; bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int64_t j = j0 + u * step; const int64_t jc = j < jend ? j : jend - 1;
                        const int sj = (int)(jc - z0);
                        ok[u] = j < jend && !(l_fr[sj] & 2);
                        v[u] = LDG(txr >= l_ty[sj] ? &D[(int64_t)x * N + jc] : &D[jc * N + x]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (ok[u]) min2_acc(q, v[u], (int)(j0 + u * step));
                }
                q = wave_min2(q);
                if (lane == 0) s_part[r][sidx] = q;
                continue;
            }
            const double* row = D + cidx(N, x, (int64_t)x + 1) - (x + 1);
            const int64_t step = (int64_t)G * S * 64;
            for (int64_t j0 = (int64_t)x + 1 + ((int64_t)g * S + sidx) * 64 + lane; j0 < n; j0 += step * 4) {
                double v[4]; int sz[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t j = j0 + u * step; const int64_t jc = j < n ? j : n - 1;
                    v[u] = LDG(&row[jc]); sz[u] = size[jc];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t j = j0 + u * step;
                    if (j < n && sz[u] != 0) min2_acc(q, v[u], (int)j);
                }
            }
            q = wave_min2(q);
            if (lane == 0) s_part[r][sidx] = q;
        }
        __syncthreads();
        if (tid < nL) {
            Min2 q = s_part[tid][0];
            for (int k2 = 1; k2 < S; ++k2) q = min2_merge(q, s_part[tid][k2]);
            outv[tid] = q;
        }
        __syncthreads();
    };
    // publish this workgroup's slot for the next round.  Every wave first drains its write-through stores (distance
    // matrix, bounds): whoever sees the slot may read them.
    auto publish = [&](Min2 q, Cand m, int nL, const Min2* rows, int row_tie = 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP2(12);
        __syncthreads();
        ++bar;
        MwGran* sl = gran + ((size_t)par * G + g) * SLOT_WORDS;
        const MwGran tag = (MwGran)bar << 32;
        if (tid == 0) {
            const unsigned long long av = (unsigned long long)__double_as_longlong(m.v), nv = (unsigned long long)__double_as_longlong(q.v);
            STX<ONEX>(&sl[0], tag | (unsigned)av); STX<ONEX>(&sl[1], tag | (unsigned)(av >> 32)); STX<ONEX>(&sl[2], tag | (unsigned)m.i);
            STX<ONEX>(&sl[3], tag | (unsigned)m.y); STX<ONEX>(&sl[4], tag | (unsigned)m.fresh);
            STX<ONEX>(&sl[5], tag | (unsigned)nv); STX<ONEX>(&sl[6], tag | (unsigned)(nv >> 32)); STX<ONEX>(&sl[7], tag | (unsigned)q.i);
            if (nL == 0) {          // rounds without refreshed rows: word 8 = "row x had a second pair at the merge height", 9-10 = second value of the NN(y) partial
                const unsigned long long sv = (unsigned long long)__double_as_longlong(q.v2);
                STX<ONEX>(&sl[8], tag | (unsigned)row_tie); STX<ONEX>(&sl[9], tag | (unsigned)sv); STX<ONEX>(&sl[10], tag | (unsigned)(sv >> 32));
            }
        }
        if (tid < nL) {
            const unsigned long long pv = (unsigned long long)__double_as_longlong(rows[tid].v), sv = (unsigned long long)__double_as_longlong(rows[tid].v2);
            MwGran* rw = sl + 8 + 5 * tid;
            STX<ONEX>(&rw[0], tag | (unsigned)pv); STX<ONEX>(&rw[1], tag | (unsigned)(pv >> 32)); STX<ONEX>(&rw[2], tag | (unsigned)rows[tid].i);
            STX<ONEX>(&rw[3], tag | (unsigned)sv); STX<ONEX>(&rw[4], tag | (unsigned)(sv >> 32));
        }
    };
    auto word_d = [&](int sl, int wd) -> double {
        return __longlong_as_double((long long)(((unsigned long long)s_words[sl][wd + 1] << 32) | s_words[sl][wd]));
    };
    // after consume(): every WAVE folds the G slots itself -- global best, NN(y), "row x had a second pair" -- so a merge round
    // needs no LDS broadcast and no workgroup barrier here; only a retry round (refreshed rows are folded one per wave) has two
    Cand d_best; Min2 d_nn; int d_rowtie = 0;
    const Min2 none2 = {INFINITY, -1, INFINITY};
    d_best.v = INFINITY; d_best.i = -1; d_best.y = -1; d_best.fresh = 0; d_nn = none2;
    auto digest = [&](int nLprev, const int* Lprev, int yrow, bool with_nn) {
        if (tid < G) {          // kept for pick_stale (read there behind a barrier)
            Cand c; c.v = word_d(tid, 0); c.i = (int)s_words[tid][2]; c.y = (int)s_words[tid][3]; c.fresh = (int)s_words[tid][4]; s_cand[tid] = c;
        }
        if (nLprev > 0) {
            for (int r = wv; r < nLprev; r += NW) {          // refreshed rows: one wave folds the G partial minima of a row
                Min2 a = none2;
                for (int u = lane; u < G; u += 64) { Min2 pq; pq.v = word_d(u, 8 + 5 * r); pq.i = (int)s_words[u][10 + 5 * r]; pq.v2 = word_d(u, 11 + 5 * r); a = min2_merge(a, pq); }
                a = wave_min2(a);
                if (lane == 0) s_row[r] = a;
            }
            __syncthreads();
        }
        Cand b; b.v = INFINITY; b.i = -1; b.y = -1; b.fresh = 0;
        Min2 a = none2;
        int rt = 0;
        for (int u = lane; u < G; u += 64) {
            Cand c; c.v = word_d(u, 0); c.i = (int)s_words[u][2]; c.y = (int)s_words[u][3]; c.fresh = (int)s_words[u][4];
            b = cbetter(b, c);
            if (with_nn) { Min2 pq; pq.v = word_d(u, 5); pq.i = (int)s_words[u][7]; pq.v2 = word_d(u, 9); a = min2_merge(a, pq); }
            if (nLprev == 0) rt |= (int)s_words[u][8];
        }
        if (lane < nLprev) {           // the rows refreshed in this round are exact now
            Cand c; c.i = Lprev[lane]; c.y = s_row[lane].i; c.v = (c.y < 0) ? INFINITY : s_row[lane].v; c.fresh = 1;
            if (c.y >= 0) b = cbetter(b, c);
        }
        d_best = wave_min_c(b);
        if (with_nn) d_nn = wave_min2(a);
        d_rowtie = (nLprev == 0 && __ballot(rt != 0) != 0ull) ? 1 : 0;
        if (nLprev > 0) {
            // owners store the refreshed rows (read back only by the owner's later arg-mins)
            if (tid < nLprev && own(Lprev[tid])) {
                const int x = Lprev[tid]; const Min2 q = s_row[tid];
                const double qv = (q.i < 0) ? (double)INFINITY : q.v;
                const int px = slot(x);
                l_nb[px] = q.i; l_md[px] = qv; l_md2[px] = (q.i < 0) ? (double)INFINITY : q.v2; l_fr[px] = 1;
                STX<ONEX>(&nb[x], q.i); STX<ONEX>(&md[x], qv);
            }
            __syncthreads();
        }
    };
    // the next refresh list: the KR best stale candidates among s_cand[0..G) and `extra` (row y after a merge); wave 0
    // extracts them one by one from registers, every workgroup arrives at the same list
    auto pick_stale = [&](Cand extra, int slot) {
        __syncthreads();               // s_cand of this round complete
        if (wv == 0) {
            MinIdx c[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int idx = lane + 64 * u;
                c[u] = none;
                if (idx <= G) {
                    Cand o = extra;
                    if (idx < G) o = s_cand[idx];
                    if (o.i >= 0 && !(o.fresh & 1) && o.v != INFINITY) { c[u].v = o.v; c[u].i = o.i; }
                }
            }
            // common case: at most KR stale candidates -> take them all (their order is irrelevant), no reduction needed
            int nl = 0;
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const bool st = c[u].i >= 0;
                const unsigned long long mk = __ballot(st);
                const int at = nl + __popcll(mk & ((1ull << lane) - 1ull));
                if (st && at < KR) s_L[slot][at] = c[u].i;
                nl += __popcll(mk);
            }
            if (nl > KR) {                       // more than KR: the KR best by (bound, row)
                nl = 0;
                for (int r = 0; r < KR; ++r) {
                    MinIdx bq = none;
#pragma unroll
                    for (int u = 0; u < 5; ++u) bq = better(bq, c[u]);
                    bq = wave_min(bq);
                    if (bq.i < 0) break;
                    if (lane == 0) s_L[slot][r] = bq.i;
#pragma unroll
                    for (int u = 0; u < 5; ++u) if (c[u].i == bq.i) c[u].i = -1;
                    nl = r + 1;
                }
            }
            if (lane == 0) s_nL[slot] = nl;
        }
        __syncthreads();
    };
    Cand nocand; nocand.v = INFINITY; nocand.i = -1; nocand.y = -1; nocand.fresh = 1;

    // ---- initial state: exact bounds from k_row_nn; owned rows g, g+G, ...
    int cnt0 = 0;
    if constexpr (SQ) {
        for (int i2 = tid; i2 < nown; i2 += T) {
            const int z = z0 + i2;
            l_ty[i2] = -1;
            l_md[i2] = z < n - 1 ? md[z] : (double)INFINITY; l_md2[i2] = z < n - 1 ? md2_init[z] : (double)INFINITY; l_nb[i2] = z < n - 1 ? nb[z] : -1; l_fr[i2] = 1;
        }
    } else
    for (int z = g + G * tid, i2 = tid; z < n; z += G * T, i2 += T) {
        act[i2] = z; pos[i2] = i2;
        l_md[i2] = z < n - 1 ? md[z] : (double)INFINITY; l_md2[i2] = z < n - 1 ? md2_init[z] : (double)INFINITY; l_nb[i2] = z < n - 1 ? nb[z] : -1; l_fr[i2] = 1;
    }
    if (tid == 0) { cnt0 = (n - g + G - 1) / G; if (cnt0 < 0) cnt0 = 0; s_cnt = cnt0; s_nL[0] = 0; s_nL[1] = 0; }
    __syncthreads();
    {
        Cand m0 = local_argmin(0, s_L[0]);
        publish(none2, m0, 0, s_row);
    }
    if (!consume(11)) return;
    digest(0, s_L[0], -1, false);
    par ^= 1;
    Cand best = d_best;
    if (!((best.fresh & 1) && best.y >= 0)) pick_stale(nocand, lp);
    int x = best.i, y = best.y; double dist = best.v; bool fresh = (best.fresh & 1) != 0;
    // cluster sizes of the pair about to merge, requested as soon as the pair is known (an L2 round trip off the merge's serial path)
    int nx_pre = 0, ny_pre = 0, cx_pre = 0, cy_pre = 0;            // (workgroup 0's first thread also needs the pair's dendrogram ids)
    int tx_pre = -1, ty_pre = -1;                                  // SQ: last merge that rewrote row x / row y
    auto prefetch_pair = [&]() {
        if (fresh && y >= 0) {
            nx_pre = size[x]; ny_pre = size[y];
            if constexpr (SQ) { tx_pre = tyv[x]; ty_pre = tyv[y]; }
            if (g == 0 && tid == 0) { cx_pre = cid[x]; cy_pre = cid[y]; }
        }
    };
    prefetch_pair();
    // a merge is taken from the arg-min only when its pair is the UNIQUE closest pair; otherwise the kernel stops and
    // run_linkage repeats the job with k_linkage_heap, which owns the reference's tie order
    auto tie_stop = [&](int flags) -> bool {
        if (!(flags & CAND_TIE)) return false;
        if (g == 0 && tid == 0) sync[5] = 1;
        return true;
    };

    for (int k = 0; k < n - 1; ++k) {
        // ---- lazy validation (cl.cpp:323-339): cooperative refresh of the KR best stale candidates per round
        for (int guard = 0; guard <= n - k; ++guard) {
            if (fresh && y >= 0) break;
            if (g == 0 && tid == 0) sync[2] += 1;            // diagnostic: retry rounds
            const int nL = s_nL[lp]; const int* L = s_L[lp];
            STAMP2(5);
            scan_rows(nL, L, s_row);
            STAMP2(0);
            Cand m = local_argmin(nL, L);
            publish(none2, m, nL, s_row);
            STAMP2(1);
            if (!consume(nL > 0 ? 8 + 5 * nL : 11)) return;
            STAMP2(2);
            digest(nL, L, -1, false);
            par ^= 1;
            best = d_best;
            STAMP2(3);
            lp ^= 1;
            if (!((best.fresh & 1) && best.y >= 0)) pick_stale(nocand, lp);
            STAMP2(4);
            x = best.i; dist = best.v; y = best.y; fresh = (best.fresh & 1) != 0;
            prefetch_pair();
        }
        if (tie_stop(best.fresh)) return;
        // ---- merge (x, y) at height dist
        const int nx = nx_pre, ny = ny_pre;
        // No workgroup barrier in front of the pass: the pair is in every thread's registers, the pass skips x and y by value, and x
        // leaves its owner's active list -- and the pair's sizes change in this workgroup's private size[] -- only behind the pass
        // (below): a slower wave may still be loading size[x] / size[y] in prefetch_pair when thread 0 gets here.
        auto write_Z = [&]() {                             // (behind the pass: thread 0 must not wait for the pair's ids before it issues its loads)
            if (tid == 0 && g == 0) {
                int ix = cx_pre, iy = cy_pre;
                if (ix > iy) { const int t = ix; ix = iy; iy = t; }
                Z[(size_t)k * 4 + 0] = (double)ix; Z[(size_t)k * 4 + 1] = (double)iy;
                Z[(size_t)k * 4 + 2] = dist;       Z[(size_t)k * 4 + 3] = (double)(nx + ny);
                cid[y] = n + k;
            }
        };
        if (k == n - 2) { write_Z(); break; }
        // ---- one pass over the owned active rows: Lance-Williams update + neighbour patches (cl.cpp:361-392),
        // NN(y) partial from the fresh distances (cl.cpp:395-404), next local arg-min
        STAMP2(5);
        Min2 q = none2;
        Cand m; m.v = INFINITY; m.i = -1; m.y = -1; m.fresh = 0;
        int row_tie = 0;
        const int cnt = SQ ? nown : s_cnt;
        const int txm = tx_pre, tym = ty_pre;             // SQ: ty of the pair before this merge
        int zdummy = 0;                                   // any valid row other than x and y (n >= 3 here)
        while (zdummy == x || zdummy == y) ++zdummy;
        if (SQ) { zdummy = z0; while ((zdummy == x || zdummy == y) && zdummy + 1 < z0 + (nown > 0 ? nown : 1)) ++zdummy; }   // (an own column: its l_ty slot exists)
        for (int p0 = tid; p0 < cnt; p0 += T * 4) {
            double dzx[4], dzy[4], mdz[4], md2z[4]; int zz[4], nbz[4], frz[4], pp[4]; int64_t izy[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * T;
                const int pc = p < cnt ? p : 0;
                pp[u] = pc;
                frz[u] = l_fr[pc];
                int z = p < cnt ? (SQ ? ((frz[u] & 2) ? -1 : z0 + pc) : act[pc]) : -1;
                if (z == y || z == x) z = -1;
                zz[u] = z;
                const int zc = z >= 0 ? z : zdummy;
                if constexpr (SQ) {
                    // the current value of {z, x} / {z, y} lives in the row of the cluster whose row was written last (row y is written
                    // below).  The row copies are requested at once, before the pair's ty have arrived (prefetch_pair's loads are still
                    // in flight): on clustered data they are the right ones for all but a handful of entries, fixed up below.
                    izy[u] = (int64_t)y * N + zc;
                    dzx[u] = LDG(&D[(int64_t)x * N + zc]);
                    dzy[u] = LDG(&D[izy[u]]);
                } else {
                    izy[u] = cidx(N, zc, y);
                    dzx[u] = LDG(&D[cidx(N, zc, x)]);
                    dzy[u] = LDG(&D[izy[u]]);
                }
                nbz[u] = l_nb[pc]; mdz[u] = l_md[pc]; md2z[u] = l_md2[pc];
            }
            STAMP2(8);
            if constexpr (SQ) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (zz[u] < 0) continue;
                    const int tz = l_ty[zz[u] - z0];
#ifdef SD_LINKAGE_STAMPS
                    { const bool fx = (txm < tz) || (tym < tz); const unsigned long long bm = __ballot(fx); fix_lanes += fx ? 1u : 0u; if (lane == 0 && bm) fix_waves += 1; }
#endif
                    if (txm < tz) dzx[u] = LDG(&D[(int64_t)zz[u] * N + x]);        // z's row was written after x's: the current {z, x} is there
                    if (tym < tz) dzy[u] = LDG(&D[(int64_t)zz[u] * N + y]);
                }
            }
            STAMP2(9);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int z = zz[u];
                if (z < 0) continue;
                const double nd = lw_centroid(dzx[u], dzy[u], dist, nx, ny);
                STX<ONEX>(&D[izy[u]], nd);
                if (z > x && dzx[u] == dist) row_tie = 1;         // row x had a second neighbour at exactly the merge height
                double mz = (z < n - 1) ? mdz[u] : INFINITY; int nz = nbz[u], fz = frz[u];
                if (z < y) {
                    // row z's entries above the diagonal: x's is gone (if z < x), y's is nd now.  Invariants: mz <= every active entry of
                    // the row (the reference's lower bound), m2 <= every active entry OTHER than the neighbour's.
                    const double m2 = fmax(md2z[u], mz);
                    if ((z < x && nz == x) || nz == y) {
                        // the neighbour's entry is the one that changed (or went away: y takes over).  Exact while nd is still the row minimum.
                        nz = y;
                        if (nd <= m2) { mz = nd; fz = 1; } else { mz = m2; fz = 0; }
                        l_md[pp[u]] = mz; l_md2[pp[u]] = m2; l_nb[pp[u]] = y; l_fr[pp[u]] = (unsigned char)fz; STX<ONEX>(&md[z], mz); STX<ONEX>(&nb[z], y);
                    } else if (nd < mz) {
                        // y becomes the neighbour; the old neighbour's entry (>= mz) joins the others, which are all >= mz
                        l_md2[pp[u]] = mz;
                        nz = y; mz = nd; fz = 1; l_md[pp[u]] = nd; l_nb[pp[u]] = y; l_fr[pp[u]] = 1; STX<ONEX>(&md[z], nd); STX<ONEX>(&nb[z], y);
                    } else if (nd < m2) l_md2[pp[u]] = nd;
                } else if (nd < q.v || (nd == q.v && z < q.i)) { q.v2 = q.v; q.v = nd; q.i = z; }
                else if (nd < q.v2) q.v2 = nd;
                if (z < n - 1) cand_acc(m, mz, z, nz, fz);
            }
        }
        STAMP2(6);
        block_min_qc(q, m, sh, shc, NW);
        STAMP2(10);
        row_tie = __syncthreads_or(row_tie);
        write_Z();
        if (tid == 0) { size[x] = 0; size[y] = nx + ny; }     // (every thread is past the pass and has used the old sizes)
        if constexpr (SQ) {
            if (tid == 0) {
                tyv[y] = k;                                   // row y is the current copy of every {y, z} from now on
                if (own(y)) l_ty[y - z0] = k;
                if (own(x)) l_fr[x - z0] = 2;                 // column x is gone
            }
        } else
        if (tid == 0 && (x % G) == g) {                       // owner drops x from its active list
            const int p = pos[x / G], c2 = s_cnt - 1, last = act[c2];
            act[p] = last; pos[last / G] = p; s_cnt = c2;
            l_md[p] = l_md[c2]; l_md2[p] = l_md2[c2]; l_nb[p] = l_nb[c2]; l_fr[p] = l_fr[c2];
        }
        STAMP2(11);
        publish(q, m, 0, s_row, row_tie);
        STAMP2(7);
        if (!consume(11)) return;
        STAMP2(2);
        digest(0, s_L[lp], y, true);
        STAMP2(3);
        par ^= 1;
        if (d_rowtie) { if (g == 0 && tid == 0) sync[5] = 1; return; }
        best = d_best;
        const Min2 nn = d_nn;
        // row y: exact by construction when it has an active neighbour above (cl.cpp:395-404), else its old (stale) bound
        Cand cy; cy.i = -1; cy.v = INFINITY; cy.y = -1; cy.fresh = 0;
        if (y < n - 1) {
            if (nn.i >= 0) {
                cy.v = nn.v; cy.i = y; cy.y = nn.i; cy.fresh = 1;
                // The next pass starts without a workgroup barrier, and a wave reads LDS behind its own writes: the values are written by the ONE wave whose
                // thread reads slot py in the pass (slot p belongs to thread p % T).  (Until round 5 lane 0 of EVERY wave wrote them: a wave that fell a
                // whole pass behind then overwrote the update the next merge had already made to the slot -- a wrong late merge in ~4 % of the runs of a
                // 2 200-row job with 16 waves per workgroup, none seen with 8 or 4; found by tools/linkage_fuzz.py, profiles/r05_linkage_fuzz.txt.)
                if (own(y)) {
                    const int py = slot(y);
                    if (lane == 0 && wv == ((py % T) >> 6)) { l_nb[py] = nn.i; l_md[py] = nn.v; l_md2[py] = nn.v2; l_fr[py] = 1; }
                    if (tid == 0) { STX<ONEX>(&nb[y], nn.i); STX<ONEX>(&md[y], nn.v); }
                }
            } else {
                cy.v = LDG(&md[y]); cy.i = y; cy.y = LDG(&nb[y]); cy.fresh = 0;
                if (own(y)) { const int py = slot(y); if (lane == 0 && wv == ((py % T) >> 6)) l_fr[py] = 0; }
            }
            best = cbetter(best, cy);
            __builtin_amdgcn_wave_barrier();       // keep the LDS stores above in front of the next pass's LDS loads in the instruction stream
        }
        lp ^= 1;
        if (!((best.fresh & 1) && best.y >= 0)) pick_stale(cy, lp);
        STAMP2(4);
        x = best.i; dist = best.v; y = best.y; fresh = (best.fresh & 1) != 0;
        prefetch_pair();
    }
#ifdef SD_LINKAGE_STAMPS
    if (g == 0 && tid == 0) for (int i = 0; i < 16; ++i) sync[8 + i] = (unsigned)(acc[i] / 100);   // microseconds
    atomicAdd(&sync[24], fix_lanes); if (lane == 0) atomicAdd(&sync[25], fix_waves);
#endif
}

// tuning hook: cost of the device-scope barrier alone (mode 0), or with each workgroup dirtying `dirty` doubles first
__global__ __launch_bounds__(MWT) void k_barrier_bench(unsigned* sync, int iters, double* scratch, int dirty)
{
    unsigned bar = 0;
    const int G = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < dirty; i += MWT) scratch[((size_t)blockIdx.x * 7919 + (size_t)i * 104729 + it) % (1 << 22)] = (double)it;
        ++bar;
        if (!mw_barrier(&sync[0], bar * (unsigned)G, &sync[1])) return;
    }
}
extern "C" int sd_bench_barrier(sd_ctx* c, int G, int iters, int dirty, double* us_per_barrier)
{
    if (!c || !us_per_barrier || G < 1 || G > c->num_cu) return SD_ERR_ARG;
    WS(c, unsigned, sync, "cl_sync", 4);
    WS(c, double, scratch, "bb_scratch", 1 << 22);
    HIPCHK(c, hipMemsetAsync(sync, 0, 4 * sizeof(unsigned), c->stream));
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    HIPCHK(c, hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL(k_barrier_bench, dim3(G), dim3(MWT), 0, c->stream, sync, iters, scratch, dirty);
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    *us_per_barrier = ms * 1e3 / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return SD_OK;
}

__global__ void k_fill_i32(int* p, int v, int64_t n, int iota)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = iota ? (int)i : v;
}

// X[N][d] (rows as given; Clustering::linkage does not normalise) -> Z[N-1][4] on the device
//   N < 1500 (or option linkage_wgs = 0): k_linkage_heap, the reference's algorithm heap included.
//   above: k_linkage_mw on G co-resident workgroups (cooperative launch: the runtime guarantees residency or refuses).  It
//   takes a merge from its parallel arg-min only while the closest pair is unique; at the first exact tie (duplicate
//   embeddings), on a refused launch or on a poll timeout the distance matrix is rebuilt and k_linkage_heap does the job.
static int linkage_prepare(sd_ctx* c, const double* d_X, int64_t N, int d, double* D, int* size, int* cid, int* nb, double* md, double* md2, bool square = false)
{
    const int64_t m = N * (N - 1) / 2;
    const int tiles = (int)((N + PT - 1) / PT);
    {
        ProfScope ps(c, "pdist", (double)m * d * 3.0, (double)m * 8.0 * (square ? 2 : 1) + (double)N * d * 8.0);
        if (square) hipLaunchKernelGGL(k_pdist_sq, dim3(tiles, tiles), dim3(256), 0, c->stream, d_X, N, d, D);
        else hipLaunchKernelGGL(k_pdist, dim3(tiles, tiles), dim3(256), 0, c->stream, d_X, N, d, D);
        KCHECK(c);
    }
    hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, size, 1, N, 0);
    hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, cid, 0, N, 1);
    KCHECK(c);
    {
        ProfScope ps(c, "row_nn", 0, (double)m * 8.0);
        hipLaunchKernelGGL(k_row_nn, dim3((unsigned)((N - 1 + 3) / 4)), dim3(256), 0, c->stream, D, N, nb, md, md2, square ? 1 : 0);
        KCHECK(c);
    }
    return SD_OK;
}

static int linkage_heap(sd_ctx* c, int64_t N, double* D, int* size, int* cid, int* nb, double* md, double* d_Z)
{
    double* hval = nullptr; int* hkey = nullptr; int* hpos = nullptr;
    if (N - 1 > HEAP_LDS) {
        WS(c, double, hv, "cl_hval", N);
        WS(c, int, hk, "cl_hkey", N);
        WS(c, int, hp, "cl_hpos", N);
        hval = hv; hkey = hk; hpos = hp;
    }
    ProfScope ps(c, "linkage_heap", 0, 24.0 * (double)N * (double)N);
    hipLaunchKernelGGL(k_linkage_heap, dim3(1), dim3(LT), (size_t)((N + 31) / 32) * sizeof(unsigned), c->stream, D, (int)N, size, cid, nb, md, d_Z, hval, hkey, hpos);
    KCHECK(c);
    return SD_OK;
}

// per-workgroup private copies for k_linkage_mw: [G][n] sizes (all 1); square form: [G][2 n] = sizes (1) followed by ty (-1)
__global__ void k_fill_size_ty(int* p, int64_t n, int64_t total, int with_ty)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) p[i] = (with_ty && ((i / n) & 1)) ? -1 : 1;
}

// linkage_rg.hip
bool linkage_rg_fits(int64_t N, int G, int TH);
hipError_t linkage_rg_launch(sd_ctx* c, bool onex, int G, int TH, double* D, int n, int* cid, const int* nb, const double* md, const double* md2,
                             double* Z, MwGran* gran, unsigned* sync, int cap, int helper, int k0 = 0, const int* sz0 = nullptr, const int* ty0 = nullptr);
int linkage_rg_slot_granules();
// linkage_hx.hip
bool linkage_hx_fits(int64_t N, int workers);
int linkage_hx_run(sd_ctx* c, bool onex, int workers, double* D, int64_t N, int* cid, int* size, int* tyv, int* nb, double* md, double* d_Z, bool* stopped,
                   double stop_above = (double)INFINITY, int64_t* merges_done = nullptr, bool* launched = nullptr);

int run_linkage(sd_ctx* c, const double* d_X, int64_t N, int d, double* d_Z)
{
    if (N < 2) return SD_OK;
    const int64_t m = N * (N - 1) / 2;
    if (N > 0x7fffffff / 4 || (double)m * 8.0 > 230e9) SD_FAIL(c, SD_ERR_ARG, "linkage: N=%lld needs a %.0f GB condensed matrix (limit 230 GB)", (long long)N, (double)m * 8e-9);
    WS(c, int, size, "cl_size", N);
    WS(c, int, cid, "cl_cid", N);
    WS(c, int, nb, "cl_nb", N);
    WS(c, double, md, "cl_md", N);
    WS(c, double, md2, "cl_md2", N);
    int rc;
    int G = (int)c->linkage_wgs;
    // auto geometry (measured on clustered data, profiles/r01_linkage_scaling.txt, r02_linkage_stamps.txt, r03_linkage_*.txt): up to N = 30 000
    // the one-XCD form with one workgroup on each of the XCD's 32 CUs; above, all XCDs
    // (r03, square form: one XCD wins up to N ~ 70 000 -- 384 ms against 443 at N = 50 158, 719 against 715 at N = 75 090 -- all XCDs above: 987 ms against 1 064 at
    // N = 100 174; all XCDs: 64 workgroups up to ~90 000 rows, 128 above)
    const bool auto_onex = G < 0 && c->linkage_one_xcd != 0 && c->num_cu >= 256 && N >= 1500 && N < 70000;
    if (auto_onex) G = 32;
    if (G < 0) G = N >= 90000 ? 128 : N >= 8000 ? 64 : N >= 1500 ? 32 : 0;
    if (G > c->num_cu) G = c->num_cu;
    int TH = (int)c->linkage_threads;
    // measured (r03, square form, N = 12 602, one XCD, three boxes): 32 x 256 75.2 / 82.7 / 76.8 ms, 32 x 512 84.4 / 77.2 / 85.0 ms -- the
    // boxes disagree, 256 wins on two of three; all XCDs 128 x 512 0.99 s at N = 100 174 (128 x 256: 1.10 s)
    // one XCD, larger N: 512 threads (N = 18 867: 121 ms against 126; N = 25 274: 170 against 179; N = 50 158: 384 against 465)
    if (TH <= 0) TH = auto_onex ? (N >= 16000 ? 512 : 256) : N >= 8000 ? 512 : 256;
    TH = TH >= 1024 ? 1024 : TH >= 512 ? 512 : TH >= 256 ? 256 : 128;      // (not below 128: the kernels fill their G <= 128 candidate slots with `tid < G`)
    if (G <= 1) {
        WS(c, double, D, "cl_D", m);
        if ((rc = linkage_prepare(c, d_X, N, d, D, size, cid, nb, md, md2))) return rc;
        return linkage_heap(c, N, D, size, cid, nb, md, d_Z);
    }
    // square form (full N x N matrix, row-only bulk accesses) while the square fits beside everything else; the condensed form above that
    bool square = c->linkage_square < 0 ? (double)N * (double)N * 8.0 <= 170e9 : c->linkage_square != 0;
    double* D = nullptr;
    if (square) {
        D = ws_get<double>(c, "cl_Dsq", (size_t)N * N);
        if (!D) { (void)hipGetLastError(); square = false; c->stats["linkage_square_alloc_failed"].launches += 1; }      // no room for the square: the condensed form needs half
    }
    if (!square) { WS(c, double, Dc, "cl_D", m); D = Dc; }
    // k_linkage_rg's own automatic geometry (r05, planted embeddings, 256 threads; profiles/r05_linkage_rg.txt): one XCD with 32 workgroups only while a thread
    // has one or two columns -- N = 12 602: 72.7 ms with 32 (one XCD), 72.6 with 64 (all XCDs); N = 18 867: 127.7 / 120.0; N = 25 274: 179.1 / 165.2 (128: 169.1;
    // k_linkage_mw 170.2); N = 50 158: 462.8 / 372.8 (128: 366.3; k_linkage_mw 381.8); N = 100 174: 128 workgroups 819 ms (64: 875; k_linkage_mw 996)
    if (c->linkage_wgs < 0 && square && c->linkage_kernel != 0 && N >= 1500) {
        G = N < 16000 ? 32 : N < 45000 ? 64 : 128;
        if (G > c->num_cu) G = c->num_cu;
    }
    if ((N + G - 1) / G > 3400) G = (int)((N + 3399) / 3400);      // active-row lists and bounds live in LDS: 32 B per owned row (109 KB + 43 KB static)
    if (G > c->num_cu || G > MWT) SD_FAIL(c, SD_ERR_ARG, "linkage: N=%lld needs %d cooperative workgroups", (long long)N, G);
    int cap = (int)((N + G - 1) / G) + 1;
    if ((rc = linkage_prepare(c, d_X, N, d, D, size, cid, nb, md, md2, square))) return rc;
    // one-XCD form while two workgroups per CU of one XCD (32 CUs) can hold the job; above, all XCDs' memory pipelines are worth more
    bool onex = c->linkage_one_xcd != 0 && G <= 32 && c->num_cu >= 256;
    // square form: k_linkage_rg (register state, sizes / ty in the slots) where its geometry fits -- at most 4 columns per thread -- else k_linkage_mw
    bool use_rg = square && c->linkage_kernel != 0;
    // measured (r05, planted embeddings): N = 12 602, one XCD, 32 workgroups: 256 threads 72.7 ms, 512 threads 81.2, 128 threads 76.6 (k_linkage_mw 76.7);
    // N = 100 174, all XCDs, 128 workgroups: 256 threads 819 ms, 512 threads 829 (k_linkage_mw 996); 64 x 512: 875
    if (use_rg && c->linkage_threads <= 0) TH = 256;
    if (use_rg && !linkage_rg_fits(N, G, TH)) {
        int t2 = TH;
        while (t2 < 1024 && !linkage_rg_fits(N, G, t2)) t2 *= 2;
        if (linkage_rg_fits(N, G, t2) && (c->linkage_threads <= 0 || c->linkage_kernel > 0)) TH = t2; else use_rg = false;
    }
    const int slot_gran = use_rg ? linkage_rg_slot_granules() : SLOT_WORDS;
    WS(c, MwGran, gran, "cl_gran", (int64_t)2 * G * slot_gran);
    HIPCHK(c, hipMemsetAsync(gran, 0, (size_t)2 * G * slot_gran * sizeof(MwGran), c->stream));
    int* size_all = nullptr;
    if (!use_rg) {
        const int64_t priv = (int64_t)G * N * (square ? 2 : 1);
        WS(c, int, sa, "cl_size_all", priv);
        size_all = sa;
        hipLaunchKernelGGL(k_fill_size_ty, dim3((unsigned)((priv + 255) / 256)), dim3(256), 0, c->stream, size_all, N, priv, square ? 1 : 0);
        KCHECK(c);
    }
    WS(c, unsigned, sync, "cl_sync", 32 + 16 * 256);
    HIPCHK(c, hipMemsetAsync(sync, 0, (32 + 16 * 256) * sizeof(unsigned), c->stream));
    const char* why = nullptr;
    bool launch_refused = false;
    if (c->linkage_force_heap) why = "forced (option linkage_force_heap)";
    else {
        ProfScope ps(c, "linkage", 0, 24.0 * (double)N * (double)N);
        int n_i = (int)N;
        void* args[] = {&D, &n_i, &size_all, &cid, &nb, &md, &md2, &d_Z, &gran, &sync, &cap, &G};
        // cooperative launch: all workgroups are resident together, or the launch is refused (they poll each other's slots)
        hipError_t le = hipErrorUnknown;
        const void* f_one = square ? (const void*)k_linkage_mw<true, true> : (const void*)k_linkage_mw<true, false>;
        const void* f_all = square ? (const void*)k_linkage_mw<false, true> : (const void*)k_linkage_mw<false, false>;
        if ((size_t)cap * 32 > 48 * 1024) {          // the row lists of a hand-set geometry may pass the default dynamic-LDS limit
            (void)hipFuncSetAttribute(f_one, hipFuncAttributeMaxDynamicSharedMemorySize, cap * 32);
            (void)hipFuncSetAttribute(f_all, hipFuncAttributeMaxDynamicSharedMemorySize, cap * 32);
            (void)hipGetLastError();
        }
        if (use_rg) {
            // "has the kernel written anything?" is read from Z itself: row 0 is zeroed here and written (size field >= 2) right behind the first merge's
            // stores into D, before any later exit (linkage_rg.hip: write_Z follows the column pass, the row-tie check comes behind it).  Round 5 kept a
            // merge counter in sync[28] instead; those two stores in the kernel's exit paths changed its register allocation and cost 9 ms per hour
            // (profiles/r06_linkage_ab.txt: 76.2 -> 85.7 ms on one box, one process, interleaved).
            HIPCHK(c, hipMemsetAsync(d_Z, 0, 4 * sizeof(double), c->stream));
            if (onex) {
                le = linkage_rg_launch(c, true, G, TH, D, n_i, cid, nb, md, md2, d_Z, gran, sync, cap, (int)c->linkage_prefetch);
                if (le != hipSuccess) { (void)hipGetLastError(); onex = false; }
            }
            if (!onex) le = linkage_rg_launch(c, false, G, TH, D, n_i, cid, nb, md, md2, d_Z, gran, sync, cap, (int)c->linkage_prefetch);
            c->stats["linkage_rg_launches"].launches += 1;
        } else {
        if (onex) {
            le = hipLaunchCooperativeKernel(f_one, dim3(8 * G), dim3(TH), args, (size_t)cap * 32, c->stream);
            if (le != hipSuccess) { (void)hipGetLastError(); onex = false; }
        }
        if (!onex) le = hipLaunchCooperativeKernel(f_all, dim3(G), dim3(TH), args, (size_t)cap * 32, c->stream);
        }
        if (le != hipSuccess) { (void)hipGetLastError(); why = "cooperative launch refused"; launch_refused = true; }
    }
    unsigned h[32] = {0};
    if (!why) {
        HIPCHK(c, hipMemcpyAsync(h, sync, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->stats["linkage_retry_rounds"].flops += (double)h[2];
#ifdef SD_LINKAGE_STAMPS
        fprintf(stderr, "linkage stamps (us): retry-scan %u retry-argmin %u barrier %u digest %u pick %u bookkeeping %u lw-compute+stores %u publish-stores %u | lw-issue %u lw-fixup %u block-min %u tie+Z %u drain %u | fix-up lanes %u waves %u\n",
                h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15], h[16], h[17], h[18], h[19], h[20], h[24], h[25]);
        if (use_rg) {          // every workgroup's view of the enabled intervals (min / mean / max over the workgroups, ms)
            std::vector<unsigned> hw((size_t)16 * G);
            HIPCHK(c, hipMemcpy(hw.data(), sync + 32, hw.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
            for (int i = 0; i < 16; ++i) {
                double mn = 1e30, mx = 0, sm = 0;
                for (int q = 0; q < G; ++q) { const double v = hw[(size_t)q * 16 + i] * 1e-2; mn = v < mn ? v : mn; mx = v > mx ? v : mx; sm += v; }
                if (mx > 0) fprintf(stderr, "  stamp %2d over %d workgroups: min %.2f mean %.2f max %.2f Mcycles (shader clock)\n", i, G, mn, sm / G, mx);
            }
        }
#endif
        if (h[1] && onex) {
            // too few workgroups found themselves on XCC 0 (dispatch not round-robin?): never again in this context; the
            // multi-XCD form does the job (the distance matrix is rebuilt by the recursive call)
            c->linkage_one_xcd = 0;
            c->stats["linkage_one_xcd_timeouts"].launches += 1;
            return run_linkage(c, d_X, N, d, d_Z);
        }
        if (h[1]) why = "slot poll timed out";
        else if (h[5]) { why = "exact tie"; c->stats["linkage_tie_fallbacks"].launches += 1; }
    }
    if (!why) return SD_OK;
    c->stats["linkage_fallbacks"].launches += 1;
    if (c->profile_detail) fprintf(stderr, "linkage: %s at N = %lld -> heap replay\n", why, (long long)N);
    // the reference's heap replayed with the row work spread over worker workgroups (k_linkage_hx) on the square matrix, where it fits
    // matrix, bounds and ids are still as linkage_prepare left them when nothing ran (forced replay, refused launch) or k_linkage_rg stopped in front of its
    // first merge (Z's row 0, zeroed before the launch, is still zero; with two pairs of duplicates: always): the replay starts from them, no second pdist
    bool untouched = c->linkage_force_heap != 0 || launch_refused;
    if (!untouched && use_rg && h[5] && !h[1]) {
        double z03 = 1.0;
        HIPCHK(c, hipMemcpy(&z03, d_Z + 3, sizeof(double), hipMemcpyDeviceToHost));
        untouched = z03 == 0.0;
    }
    if (square && c->linkage_tie_kernel != 0) {
        bool hx_onex = c->linkage_one_xcd != 0 && c->num_cu >= 256 && linkage_hx_fits(N, 31);
        int workers = hx_onex ? 31 : (N >= 60000 ? 127 : 63);
        if (c->linkage_tie_kernel > 1) { workers = (int)c->linkage_tie_kernel; hx_onex = hx_onex && workers <= 31; }
        if (workers + 1 > c->num_cu) workers = c->num_cu - 1;
        if (linkage_hx_fits(N, workers)) {
            WS(c, int, tyv, "cl_ty", N);
            // Ties at height 0 -- rows that occur twice, the usual reason for a tie -- end once the duplicates are merged.  The replay takes the merges at
            // height 0 only, and k_linkage_rg continues from that state (exact bounds of the rows still there from k_row_nn_mid); a later tie sends the whole
            // job through the replay, as before.  (Both kernels produce the reference's sequence: the replay always, k_linkage_rg while the closest pair is unique.)
            double tie_h = 0.0;
            { unsigned long long b = ((unsigned long long)h[27] << 32) | h[26]; memcpy(&tie_h, &b, 8); }
            if (use_rg && h[5] && !h[1] && tie_h == 0.0 && c->linkage_zero_phase != 0) {
                bool z_onex = hx_onex; int z_workers = workers;
                for (int attempt = 0; attempt < 2; ++attempt) {
                    if (!untouched && (rc = linkage_prepare(c, d_X, N, d, D, size, cid, nb, md, md2, true))) return rc;
                    untouched = false;
                    hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, tyv, -1, N, 0);
                    KCHECK(c);
                    bool stopped = false, launched = false; int64_t k_done = 0;
                    if ((rc = linkage_hx_run(c, z_onex, z_workers, D, N, cid, size, tyv, nb, md, d_Z, &stopped, 0.0, &k_done, &launched))) return rc;
                    if (!launched) untouched = true;      // a refused launch has not touched matrix, bounds or ids: the next attempt needs no second pdist
                    if (stopped) {
                        if (!z_onex) break;
                        z_onex = false; z_workers = N >= 60000 ? 127 : 63;
                        if (!linkage_hx_fits(N, z_workers)) break;
                        continue;
                    }
                    c->stats["linkage_zero_phase_merges"].flops += (double)k_done;
                    if (k_done >= N - 1) { c->stats["linkage_hx_jobs"].launches += 1; return SD_OK; }
                    if (k_done == 0) break;
                    {
                        ProfScope ps(c, "row_nn", 0, (double)m * 8.0);
                        hipLaunchKernelGGL(k_row_nn_mid, dim3((unsigned)((N - 1 + 3) / 4)), dim3(256), 0, c->stream, D, N, size, tyv, nb, md, md2);
                        KCHECK(c);
                    }
                    HIPCHK(c, hipMemsetAsync(gran, 0, (size_t)2 * G * slot_gran * sizeof(MwGran), c->stream));
                    HIPCHK(c, hipMemsetAsync(sync, 0, (32 + 16 * 256) * sizeof(unsigned), c->stream));
                    hipError_t le;
                    {
                        ProfScope ps(c, "linkage", 0, 24.0 * (double)N * (double)N);
                        le = linkage_rg_launch(c, onex, G, TH, D, (int)N, cid, nb, md, md2, d_Z, gran, sync, cap, (int)c->linkage_prefetch, (int)k_done, size, tyv);
                        c->stats["linkage_rg_launches"].launches += 1;
                    }
                    if (le != hipSuccess) { (void)hipGetLastError(); break; }
                    unsigned h2[32] = {0};
                    HIPCHK(c, hipMemcpyAsync(h2, sync, sizeof(h2), hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    if (!h2[1] && !h2[5]) { c->stats["linkage_zero_phase_jobs"].launches += 1; return SD_OK; }
                    if (c->profile_detail) fprintf(stderr, "linkage: zero phase (%lld merges), then %s -> whole replay\n", (long long)k_done, h2[5] ? "another tie" : "a poll timeout");
                    break;
                }
            }
            for (int attempt = 0; attempt < 2; ++attempt) {
                if (!untouched && (rc = linkage_prepare(c, d_X, N, d, D, size, cid, nb, md, md2, true))) return rc;
                untouched = false;
                hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, tyv, -1, N, 0);
                KCHECK(c);
                bool stopped = false, launched = false;
                if ((rc = linkage_hx_run(c, hx_onex, workers, D, N, cid, size, tyv, nb, md, d_Z, &stopped, (double)INFINITY, nullptr, &launched))) return rc;
                if (!launched) { untouched = true; c->stats["linkage_hx_refused"].launches += 1; }
                if (!stopped) { c->stats["linkage_hx_jobs"].launches += 1; return SD_OK; }
                if (!hx_onex) break;
                hx_onex = false;                 // one XCD refused or too few workgroups arrived there: all XCDs
                workers = N >= 60000 ? 127 : 63;
                if (!linkage_hx_fits(N, workers)) break;
            }
            c->stats["linkage_hx_failed"].launches += 1;
        }
    }
    // the heap kernel works on the condensed matrix; the square one (up to 170 GB) is given back first, so that the fallback -- exact
    // ties are its designed trigger -- also fits right below the auto-square limit (the stream is idle: it was synchronised above)
    { auto it = c->ws.find("cl_Dsq"); if (it != c->ws.end()) it->second.release(); }
    WS(c, double, Dh, "cl_D", m);
    if ((rc = linkage_prepare(c, d_X, N, d, Dh, size, cid, nb, md, md2))) return rc;
    return linkage_heap(c, N, Dh, size, cid, nb, md, d_Z);
}

// fcluster(criterion="distance"), cl.cpp:121-232 + 442-457.  Node ids grow with merge order, so the
// per-node max merge height is a forward pass; labels follow the reference's visiting order: internal
// left subtree, internal right subtree, then the leaf children of the node.
void fcluster_host(const std::vector<double>& Z, int64_t n, double cutoff, std::vector<int>& T)
{
    T.assign((size_t)n, 0);
    if (n == 1) { T[0] = 1; return; }
    if (n < 2) return;
    std::vector<double> MD((size_t)n - 1);
    for (int64_t k = 0; k < n - 1; ++k) {
        double mx = Z[k * 4 + 2];
        const int64_t lc = (int64_t)Z[k * 4 + 0], rc = (int64_t)Z[k * 4 + 1];
        if (lc >= n && MD[lc - n] > mx) mx = MD[lc - n];
        if (rc >= n && MD[rc - n] > mx) mx = MD[rc - n];
        MD[k] = mx;
    }
    struct Fr { int64_t node; int label; int state; };
    std::vector<Fr> st;
    st.push_back({2 * n - 2, 0, 0});
    int ncl = 0;
    while (!st.empty()) {
        Fr& f = st.back();
        const int64_t k = f.node - n;
        const int64_t lc = (int64_t)Z[k * 4 + 0], rc = (int64_t)Z[k * 4 + 1];
        if (f.state == 0) {
            if (f.label == 0 && MD[k] <= cutoff) f.label = ++ncl;
            f.state = 1;
            if (lc >= n) { const int lab = f.label; st.push_back({lc, lab, 0}); continue; }
        }
        if (f.state == 1) {
            f.state = 2;
            if (rc >= n) { const int lab = f.label; st.push_back({rc, lab, 0}); continue; }
        }
        if (lc < n) T[lc] = f.label ? f.label : ++ncl;
        if (rc < n) T[rc] = f.label ? f.label : ++ncl;
        st.pop_back();
    }
}

int run_cluster_labels(sd_ctx* c, const double* d_Xn, int64_t N, int d, double cutoff, std::vector<int>& labels1, std::vector<double>* Zout)
{
    labels1.assign((size_t)N, 0);
    if (N == 1) { labels1[0] = 1; return SD_OK; }
    if (N < 2) return SD_OK;
    WS(c, double, dZ, "cl_Z", (N - 1) * 4);
    int rc = run_linkage(c, d_Xn, N, d, dZ);
    if (rc) return rc;
    std::vector<double> Z((size_t)(N - 1) * 4);
    HIPCHK(c, hipMemcpyAsync(Z.data(), dZ, Z.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fcluster_host(Z, N, cutoff, labels1);
    if (Zout) Zout->swap(Z);
    return SD_OK;
}

// ---------------------------------------------------------------- cluster means (sequential in member order)
// members of cluster k are order[off[k] .. off[k+1]) in ascending row index: same summation order as
// Helper::calculateClusterMeans (sd.cpp:442-473) and the centroid loop of assign_embeddings (sd.cpp:2149-2167)
__global__ void k_cluster_means(const double* __restrict__ X, int d, const int* __restrict__ order, const int* __restrict__ off,
                                double* __restrict__ cen)
{
    const int k = blockIdx.x, q = threadIdx.x;
    if (q >= d) return;
    double s = 0.0;
    const int a = off[k], b = off[k + 1];
    int t = a;
    for (; t + 16 <= b; t += 16) {          // 16 rows in flight, added in member order (the sum itself stays sequential)
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = X[(size_t)order[t + u] * d + q];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; t < b; ++t) s += X[(size_t)order[t] * d + q];
    cen[(size_t)k * d + q] = s / (double)(b - a);
}

// cosine distance with the reference's sequential sums (sd.cpp:476-498); soft = 2 - d; argmax first-max-wins
#define ASSIGN_TILE 1024
__global__ __launch_bounds__(64) void k_assign(const double* __restrict__ E, int64_t M, int d, const double* __restrict__ cen, int K,
                                               int* __restrict__ hard, int* __restrict__ err, double* __restrict__ soft_out /*[M][K] or null*/,
                                               double* __restrict__ best_out /*[M] or null*/)
{
    // arg-max over the K clusters in tiles of ASSIGN_TILE scores (any number of clusters fits the fixed LDS tile); first maximum wins,
    // as Helper::argmax does (sd.cpp:293-316): tiles in ascending k, strict > inside and across tiles
    __shared__ double soft[ASSIGN_TILE];
    const int64_t row = blockIdx.x;
    const double* e = E + (size_t)row * d;
    int best = 0; double mv = -DBL_MAX, sb = NAN; bool first = true;
    for (int k0 = 0; k0 < K; k0 += ASSIGN_TILE) {
        const int kn = K - k0 < ASSIGN_TILE ? K - k0 : ASSIGN_TILE;
        for (int k = threadIdx.x; k < kn; k += 64) {
            const double* cc = cen + (size_t)(k0 + k) * d;
            double dot = 0.0, m1 = 0.0, m2 = 0.0;
            for (int i = 0; i < d; ++i) { dot += e[i] * cc[i]; m1 += e[i] * e[i]; m2 += cc[i] * cc[i]; }
            if (m1 == 0.0 || m2 == 0.0) { *err = 1; soft[k] = NAN; }
            else soft[k] = 2.0 - (1.0 - (dot / (sqrt(m1) * sqrt(m2))));
            if (soft_out) soft_out[(size_t)row * K + k0 + k] = soft[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (first) { sb = soft[0]; first = false; }             // best stays 0 when no score compares greater (NaN row -> cluster 0)
            for (int k = 0; k < kn; ++k) if (soft[k] > mv) { mv = soft[k]; best = k0 + k; sb = soft[k]; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        hard[row] = best;
        if (best_out) best_out[row] = sb;                          // NaN for rows without an embedding
    }
}

// ---------------------------------------------------------------- constrained_argmax (clustering/Clustering.py:81-94)
// hard[c] = linear_sum_assignment(soft[c] (3 x K), maximize=True): every local speaker of a chunk goes to a DIFFERENT cluster.
// The Python delegates to scipy.optimize.linear_sum_assignment (third-party, unpinned by the reference); its published
// algorithm -- Crouse's shortest-augmenting-path rectangular LSAP as implemented in scipy's rectangular_lsap.cpp (1.6 and
// later): rows in order, `remaining` columns filled in reverse, ties towards an unassigned column, transpose when there are
// fewer clusters than speakers -- is restated here step by step, because rows without an embedding become CONSTANT rows
// (nan_to_num with the global minimum) and which optimal assignment comes out of the many equal ones is decided by exactly
// those details.  nr <= nc <= LSAP_MAXC after the optional transpose.
#define LSAP_MAXC 64
__host__ __device__ inline void lsap_solve(int nr, int nc, const double* cost /*[nr][nc], minimised*/, int* col4row)
{
    double u[SD_SPEAKERS] = {0, 0, 0}, v[LSAP_MAXC], spc[LSAP_MAXC];
    int path[LSAP_MAXC], row4col[LSAP_MAXC], remaining[LSAP_MAXC];
    bool SR[SD_SPEAKERS], SC[LSAP_MAXC];
    for (int j = 0; j < nc; ++j) { v[j] = 0.0; path[j] = -1; row4col[j] = -1; }
    for (int i = 0; i < nr; ++i) col4row[i] = -1;
    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0.0;
        int num_remaining = nc;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        for (int i = 0; i < nr; ++i) SR[i] = false;
        for (int j = 0; j < nc; ++j) { SC[j] = false; spc[j] = INFINITY; }
        int sink = -1, i = cur;
        while (sink == -1) {
            int index = -1; double lowest = INFINITY;
            SR[i] = true;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = minVal + cost[i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            minVal = lowest;
            if (index < 0) return;                              // infeasible (cannot happen with finite costs)
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = true;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += minVal;
        for (int r = 0; r < nr; ++r) if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < nc; ++j) if (SC[j]) v[j] -= minVal - spc[j];
        int j = sink;
        while (true) {
            const int r = path[j];
            row4col[j] = r;
            const int t = col4row[r]; col4row[r] = j; j = t;
            if (r == cur) break;
        }
    }
}
__host__ __device__ inline void constrained_argmax_chunk(const double* soft /*[3][K]*/, int K, double fill, int* hard3)
{
    double cost[SD_SPEAKERS * LSAP_MAXC];
    int c4r[SD_SPEAKERS];
    hard3[0] = hard3[1] = hard3[2] = -2;
    if (K >= SD_SPEAKERS) {
        for (int s = 0; s < SD_SPEAKERS; ++s) for (int k = 0; k < K; ++k) { const double x = soft[s * K + k]; cost[s * K + k] = -((x != x) ? fill : x); }
        lsap_solve(SD_SPEAKERS, K, cost, c4r);
        for (int s = 0; s < SD_SPEAKERS; ++s) hard3[s] = c4r[s] >= 0 ? c4r[s] : -2;
    } else {                                                    // fewer clusters than speakers: scipy transposes
        for (int k = 0; k < K; ++k) for (int s = 0; s < SD_SPEAKERS; ++s) { const double x = soft[s * K + k]; cost[k * SD_SPEAKERS + s] = -((x != x) ? fill : x); }
        lsap_solve(K, SD_SPEAKERS, cost, c4r);
        for (int k = 0; k < K; ++k) if (c4r[k] >= 0) hard3[c4r[k]] = k;
    }
}
__global__ void k_constrained_argmax(const double* __restrict__ soft, int64_t chunks, int K, double fill, int* __restrict__ hard)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= chunks) return;
    int h[3];
    constrained_argmax_chunk(soft + (size_t)c * SD_SPEAKERS * K, K, fill, h);
    hard[c * 3] = h[0]; hard[c * 3 + 1] = h[1]; hard[c * 3 + 2] = h[2];
}

static double cos_dist_host(const double* a, const double* b, int d, bool* err)      // sd.cpp:476-498
{
    double dot = 0.0, m1 = 0.0, m2 = 0.0;
    for (int i = 0; i < d; ++i) { dot += a[i] * b[i]; m1 += a[i] * a[i]; m2 += b[i] * b[i]; }
    if (m1 == 0.0 || m2 == 0.0) { *err = true; return NAN; }
    return 1.0 - (dot / (sqrt(m1) * sqrt(m2)));
}

// group rows by label (ascending row order inside each group)
static void group_by_label(const std::vector<int>& lab, int nl, std::vector<int>& order, std::vector<int>& off)
{
    off.assign((size_t)nl + 1, 0);
    for (int v : lab) off[(size_t)v + 1]++;
    for (int k = 0; k < nl; ++k) off[(size_t)k + 1] += off[(size_t)k];
    order.resize(lab.size());
    std::vector<int> pos(off.begin(), off.end() - 1);
    for (size_t i = 0; i < lab.size(); ++i) order[(size_t)pos[(size_t)lab[i]]++] = (int)i;
}

// Constrained number of clusters -- the branch the reference leaves unimplemented (assert(false), sd.cpp:2368-2369);
// specification = the Python it was ported from, clustering/Clustering.py:352-399: re-cut the dendrogram by merge
// index, walking away from the tuned threshold, until the number of large clusters is (closest to) num_clusters.
static void constrained_recut(const std::vector<double>& Z, int64_t N, double threshold, size_t mcs, int num_clusters, std::vector<int>& lab)
{
    std::vector<double> Zi(Z);
    for (int64_t k = 0; k < N - 1; ++k) Zi[(size_t)k * 4 + 2] = (double)k;          // Clustering.py:353-354
    std::vector<int64_t> order((size_t)N - 1);
    for (int64_t k = 0; k < N - 1; ++k) order[(size_t)k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        return fabs(Z[(size_t)a * 4 + 2] - threshold) < fabs(Z[(size_t)b * 4 + 2] - threshold); });   // :362
    auto count_large = [&](const std::vector<int>& t1) {
        int mx = 0; for (int v : t1) if (v > mx) mx = v;
        std::vector<size_t> cnt((size_t)mx + 1, 0);
        for (int v : t1) cnt[(size_t)v]++;
        int nlarge = 0; for (size_t v = 1; v < cnt.size(); ++v) if (cnt[v] >= mcs) nlarge++;
        return nlarge;
    };
    int64_t best_iteration = N - 1; int best_large = 1;                               // :356-357
    std::vector<int> t1;
    bool exact = false;
    for (int64_t it : order) {
        if (Zi[(size_t)it * 4 + 3] < (double)mcs) continue;                           // :366-368
        fcluster_host(Zi, N, (double)it, t1);                                         // :371
        const int nlarge = count_large(t1);
        if (std::abs(nlarge - num_clusters) < std::abs(best_large - num_clusters)) { best_iteration = it; best_large = nlarge; }   // :377-381
        if (nlarge == num_clusters) { exact = true; break; }                          // :384-385
    }
    if (!exact) fcluster_host(Zi, N, (double)best_iteration, t1);                     // :388-391
    lab.resize((size_t)N);
    for (int64_t i = 0; i < N; ++i) lab[(size_t)i] = t1[(size_t)i] - 1;
}

// d_emb: [M][d] f64 (NaN rows = no embedding), M = chunks*3.  hard: [M]
int run_clustering(sd_ctx* c, const double* d_emb, int64_t M, int d, std::vector<int>& hard, int* Kout,
                   int num_clusters, int min_clusters, int max_clusters, std::vector<double>* soft_best)
{
    hard.assign((size_t)M, 0);
    if (soft_best) soft_best->assign((size_t)M, NAN);
    if (Kout) *Kout = 1;
    const bool dumping = !c->dump_dir.empty();
    c->stash.clustered = false;
    if (M <= 0) return SD_OK;
    // a10: rows whose first element is not NaN (sd.cpp:2224)
    std::vector<double> first((size_t)M);
    HIPCHK(c, hipMemcpy2DAsync(first.data(), sizeof(double), d_emb, (size_t)d * sizeof(double), sizeof(double), (size_t)M, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<int> tidx;
    for (int64_t i = 0; i < M; ++i) if (!std::isnan(first[(size_t)i])) tidx.push_back((int)i);
    const int64_t N = (int64_t)tidx.size();
    // set_num_clusters (sd.cpp:2261-2296; with num_clusters given, max = num_clusters as in Clustering.py:27-41 --
    // the port's `max_clusters == num_clusters;` is a no-op typo, sd.cpp:2278)
    const bool constrained = (num_clusters != -1) || (min_clusters != -1) || (max_clusters != -1);
    if (num_clusters != -1) { min_clusters = num_clusters; max_clusters = num_clusters; }
    if (min_clusters == -1) min_clusters = 1;
    if (max_clusters == -1) max_clusters = (int)N;
    min_clusters = std::max(1, std::min((int)N, min_clusters));
    max_clusters = std::max(1, std::min((int)N, max_clusters));
    if (min_clusters > max_clusters) min_clusters = max_clusters;
    if (min_clusters == max_clusters) num_clusters = min_clusters;
    if (N < 2 || max_clusters < 2) return SD_OK;                    // all zeros (sd.cpp:2081-2088)
    WS(c, int, d_tidx, "cl_tidx", N);
    WS(c, double, X, "cl_X", N * d);
    WS(c, double, Xn, "cl_Xn", N * d);
    HIPCHK(c, hipMemcpyAsync(d_tidx, tidx.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_gather_normalize, dim3((unsigned)((N + 127) / 128)), dim3(128), 0, c->stream, d_emb, d_tidx, N, d, X, Xn);
    KCHECK(c);
    // a12+a13 with the reference's float-typed threshold (sd.cpp:2049) promoted to double
    const float thr = 0.7153814381597874;
    std::vector<int> lab;
    std::vector<double> Zh;
    int rc = run_cluster_labels(c, Xn, N, d, (double)thr, lab, &Zh);
    if (rc) return rc;
    int nl = 0;
    for (auto& v : lab) { v -= 1; if (v + 1 > nl) nl = v + 1; }
    if (dumping) c->stash.clusters = lab;
    // a11: size split
    size_t mcs = std::min<size_t>(15, std::max<size_t>(1, (size_t)std::round(0.1 * (double)N)));     // sd.cpp:2308
    std::vector<int> order, off;
    group_by_label(lab, nl, order, off);
    if (constrained) {
        int nlarge0 = 0;
        for (int k = 0; k < nl; ++k) if ((size_t)(off[(size_t)k + 1] - off[(size_t)k]) >= mcs) nlarge0++;
        int target = (min_clusters == max_clusters) ? min_clusters : -1;
        if (nlarge0 < min_clusters) target = min_clusters;                                            // sd.cpp:2361-2366
        if (nlarge0 > max_clusters) target = max_clusters;
        if (target != -1) {
            constrained_recut(Zh, N, (double)thr, mcs, target, lab);
            nl = 0;
            for (int v : lab) if (v + 1 > nl) nl = v + 1;
            group_by_label(lab, nl, order, off);
        }
    }
    std::vector<int> large, small;
    for (int k = 0; k < nl; ++k) {
        const size_t cnt = (size_t)(off[(size_t)k + 1] - off[(size_t)k]);
        if (cnt == 0) continue;
        if (cnt >= mcs) large.push_back(k); else small.push_back(k);
    }
    if (large.empty()) {
        // reference: assert(false) in assert-enabled builds (sd.cpp:2368), all-zero labels otherwise (sd.cpp:2371-2375)
        std::fill(lab.begin(), lab.end(), 0);
        nl = 1;
    } else if (!small.empty()) {
        WS(c, int, d_order, "cl_order", N);
        WS(c, int, d_off, "cl_off", nl + 1);
        WS(c, double, d_cen, "cl_cen", (size_t)nl * d);
        HIPCHK(c, hipMemcpyAsync(d_order, order.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_off, off.data(), (size_t)(nl + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_cluster_means, dim3(nl), dim3(((d + 63) / 64) * 64), 0, c->stream, X, d, d_order, d_off, d_cen);   // means of UN-normalised rows (sd.cpp:2386)
        KCHECK(c);
        std::vector<double> cen((size_t)nl * d);
        HIPCHK(c, hipMemcpyAsync(cen.data(), d_cen, cen.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        bool err = false;
        std::vector<int> remap((size_t)nl);
        for (int k = 0; k < nl; ++k) remap[(size_t)k] = k;
        for (int sk : small) {
            float minVal = FLT_MAX; int best = -1;                                      // float accumulator, sd.cpp:2396
            for (size_t a = 0; a < large.size(); ++a) {
                const double dd = cos_dist_host(&cen[(size_t)large[a] * d], &cen[(size_t)sk * d], d, &err);
                if (dd < minVal) { minVal = (float)dd; best = (int)a; }
            }
            if (best >= 0) remap[(size_t)sk] = large[(size_t)best];
        }
        if (err) SD_FAIL(c, SD_ERR_NUMERIC, "zero-magnitude cluster centroid (reference throws, sd.cpp:493-495)");
        for (auto& v : lab) v = remap[(size_t)v];
        // findUniqueClusters: renumber 0..K-1 in sorted-id order (sd.cpp:519-548)
        std::vector<int> seen((size_t)nl, -1);
        for (int v : lab) seen[(size_t)v] = 0;
        int nk = 0;
        for (int k = 0; k < nl; ++k) if (seen[(size_t)k] == 0) seen[(size_t)k] = nk++;
        for (auto& v : lab) v = seen[(size_t)v];
        nl = nk;
    }
    // a14: centroids of the final clusters (un-normalised train rows), cosine cdist of ALL rows, argmax
    group_by_label(lab, nl, order, off);
    WS(c, int, d_order2, "cl_order", N);
    WS(c, int, d_off2, "cl_off", nl + 1);
    WS(c, double, d_cen2, "cl_cen", (size_t)nl * d);
    WS(c, int, d_hard, "cl_hard", M);
    WS(c, int, d_err, "cl_err", 4);
    HIPCHK(c, hipMemcpyAsync(d_order2, order.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_off2, off.data(), (size_t)(nl + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d_err, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(k_cluster_means, dim3(nl), dim3(((d + 63) / 64) * 64), 0, c->stream, X, d, d_order2, d_off2, d_cen2);
    KCHECK(c);
    const bool constrained_assign = c->constrained_assignment && (M % SD_SPEAKERS) == 0;
    // the full [M][K] score table only for the constrained assignment (it needs every score); the confidence needs the best score alone
    double* d_soft = nullptr; double* d_best = nullptr;
    if (constrained_assign || dumping) { WS(c, double, t_soft, "cl_soft", (size_t)M * nl); d_soft = t_soft; }
    if (constrained_assign || soft_best) { WS(c, double, t_best, "cl_best", M); d_best = t_best; }
    hipLaunchKernelGGL(k_assign, dim3((unsigned)M), dim3(64), 0, c->stream, d_emb, M, d, d_cen2, nl, d_hard, d_err, d_soft, d_best);
    KCHECK(c);
    int herr = 0;
    HIPCHK(c, hipMemcpyAsync(hard.data(), d_hard, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&herr, d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (soft_best) HIPCHK(c, hipMemcpyAsync(soft_best->data(), d_best, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (herr) SD_FAIL(c, SD_ERR_NUMERIC, "zero-magnitude embedding or centroid in assignment (reference throws, sd.cpp:493-495)");
    if (constrained_assign) {
        // Clustering.py:83: NaN (rows without an embedding) -> the smallest soft score of the whole recording
        std::vector<double> hs((size_t)M * nl);
        HIPCHK(c, hipMemcpy(hs.data(), d_soft, hs.size() * sizeof(double), hipMemcpyDeviceToHost));
        double fill = INFINITY;
        for (double x : hs) if (x == x && x < fill) fill = x;
        if (fill == INFINITY) fill = 0.0;
        const int64_t chunks = M / SD_SPEAKERS;
        if (nl <= LSAP_MAXC) {
            hipLaunchKernelGGL(k_constrained_argmax, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, c->stream, d_soft, chunks, nl, fill, d_hard);
            KCHECK(c);
            HIPCHK(c, hipMemcpyAsync(hard.data(), d_hard, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        } else SD_FAIL(c, SD_ERR_ARG, "constrained assignment supports up to %d clusters (found %d)", LSAP_MAXC, nl);
        if (soft_best)                                             // confidence follows the cluster actually assigned
            for (int64_t i = 0; i < M; ++i) (*soft_best)[(size_t)i] = hard[(size_t)i] >= 0 ? hs[(size_t)i * nl + hard[(size_t)i]] : NAN;
    }
    if (Kout) *Kout = nl;
    if (dumping) {
        StepStash& S = c->stash;
        S.clustered = true; S.N = N; S.K = nl; S.cluster_res = lab; S.hard_pre = hard;
        S.X.resize((size_t)N * d); S.Xn.resize((size_t)N * d); S.soft.resize((size_t)M * nl);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(S.X.data(), X, S.X.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(S.Xn.data(), Xn, S.Xn.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(S.soft.data(), d_soft, S.soft.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    return SD_OK;
}
