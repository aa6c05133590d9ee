// linkage_dev.h -- device helpers shared by the linkage kernels (cluster.hip: k_linkage_heap, k_linkage_mw; linkage_rg.hip: k_linkage_rg).
// Translation units that include this file are compiled with -ffp-contract=off (Makefile EXACT): fp64 results bit-identical to the reference's x86 build.
#pragma once
#include "common.h"
#include <cfloat>
#include <cmath>

__host__ __device__ __forceinline__ int64_t cidx(int64_t n, int64_t i, int64_t j)     // cl.cpp:236-242
{
    if (i < j) return n * i - (i * (i + 1) / 2) + (j - i - 1);
    return n * j - (j * (j + 1) / 2) + (i - j - 1);
}

// ---------------------------------------------------------------- nearest active neighbour above a row (cl.cpp:259-276)
struct MinIdx { double v; int i; };
__device__ __forceinline__ MinIdx better(MinIdx a, MinIdx b)
{
    // smaller value wins; equal values -> lower index (== "first strictly smaller" of a sequential scan); -1 = none
    if (b.i < 0) return a;
    if (a.i < 0) return b;
    if (b.v < a.v) return b;
    if (b.v == a.v && b.i < a.i) return b;
    return a;
}
// Wave-wide reductions with DPP lane moves (quad swaps, half-mirror, mirror, row broadcasts) instead of
// ds_bpermute shuffles: the dependent chain is ~10x shorter, and these reductions sit on the serial path of
// every merge.  All 64 lanes must be active.  The result (in lane 63 after the last step) is returned to all lanes.
template <int CTRL, int RM> __device__ __forceinline__ int dppi(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, RM, 0xF, false); }
template <int CTRL, int RM> __device__ __forceinline__ double dppd(double v)
{
    const int lo = dppi<CTRL, RM>(__double2loint(v)), hi = dppi<CTRL, RM>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Wave minimum in two moves instead of six (value, index) butterfly steps of ~12 dependent instructions each: the minimum VALUE
// first (v_min_f64 over DPP moves), then a ballot of the lanes that hold it.  One lane in the ballot (the rule on data without
// ties): its index is read with v_readlane.  Several: the lowest index among them, as `better` decides.  Same result as the
// butterfly for every input without NaN (distances are never NaN here).
__device__ __forceinline__ double wave_min_d(double v)
{
    v = fmin(v, dppd<0xB1, 0xF>(v)); v = fmin(v, dppd<0x4E, 0xF>(v)); v = fmin(v, dppd<0x141, 0xF>(v)); v = fmin(v, dppd<0x140, 0xF>(v));
    v = fmin(v, dppd<0x142, 0xA>(v)); v = fmin(v, dppd<0x143, 0xC>(v));
    return readlane_d(v, 63);
}
__device__ __forceinline__ int wave_min_i(int v)
{
    v = min(v, dppi<0xB1, 0xF>(v)); v = min(v, dppi<0x4E, 0xF>(v)); v = min(v, dppi<0x141, 0xF>(v)); v = min(v, dppi<0x140, 0xF>(v));
    v = min(v, dppi<0x142, 0xA>(v)); v = min(v, dppi<0x143, 0xC>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ MinIdx wave_min(MinIdx m)
{
    const bool has = m.i >= 0;
    const double vmin = wave_min_d(has ? m.v : (double)INFINITY);
    const bool at = has && m.v == vmin;
    const unsigned long long mask = __ballot(at);
    MinIdx r; r.v = INFINITY; r.i = -1;
    if (mask == 0) return r;
    r.v = vmin;
    if (mask & (mask - 1)) r.i = wave_min_i(at ? m.i : 0x7fffffff);
    else r.i = __builtin_amdgcn_readlane(m.i, __builtin_amdgcn_readfirstlane(__ffsll((long long)mask) - 1));
    return r;
}

// Two smallest values of a set: the minimum with its index (same rules as MinIdx) and the smallest value among all OTHER elements
// (equal to the minimum when it occurs twice).  The second value is what makes a row's bound robust: when the distance to its
// neighbour grows but stays below every other entry of the row, the bound is still exact (see k_linkage_mw).
struct Min2 { double v; int i; double v2; };
__device__ __forceinline__ void min2_acc(Min2& m, double v, int j)       // sequential scan in ascending j
{
    if (v < m.v) { m.v2 = m.v; m.v = v; m.i = j; }
    else if (v < m.v2) m.v2 = v;
}
__device__ __forceinline__ Min2 min2_merge(Min2 a, Min2 b)
{
    if (b.i < 0) return a;
    if (a.i < 0) return b;
    Min2 r;
    const bool bw = b.v < a.v || (b.v == a.v && b.i < a.i);
    r.v = bw ? b.v : a.v; r.i = bw ? b.i : a.i;
    r.v2 = fmin(bw ? a.v : b.v, fmin(a.v2, b.v2));
    return r;
}
__device__ __forceinline__ Min2 wave_min2(Min2 m)
{
    MinIdx q; q.v = m.v; q.i = m.i;
    const MinIdx w = wave_min(q);
    Min2 r; r.v = w.v; r.i = w.i; r.v2 = INFINITY;
    if (w.i < 0) return r;
    // every lane but the winner's contributes its own minimum, the winner's lane its second value
    const bool win = m.i == w.i && m.i >= 0;
    r.v2 = wave_min_d(win ? m.v2 : (m.i >= 0 ? m.v : (double)INFINITY));
    return r;
}

// Lance-Williams centroid update with the reference's operation order, cl.cpp:250-256
__device__ __forceinline__ double lw_centroid(double d_xi, double d_yi, double d_xy, int sx, int sy)
{
    return sqrt(((((double)sx * d_xi * d_xi) + ((double)sy * d_yi * d_yi)) -
                 ((double)(sx * sy) * d_xy * d_xy) / (double)(sx + sy)) / (double)(sx + sy));
}

// A slot is a set of 8-byte granules {32-bit payload word, 32-bit round tag}: a reader that sees the tag of the round it waits for has the
// payload of that round (8-byte stores are single transactions), so publishing needs no separate "ready" flag and no counter.
typedef unsigned long long MwGran;
// arg-min candidate flags: bit 0 = the bound is exact; bit 1 (CAND_TIE) = some OTHER row holds exactly the same bound -- the case in which the
// reference's heap, not the value, decides who comes first (k_linkage_heap)
#define CAND_TIE 2

// Cross-workgroup data of the cooperative kernel (distance matrix, bounds, neighbours, slots) is moved with agent-scope
// relaxed atomics only: on gfx950 these are `sc1` loads / stores (write-through past the per-XCD L2, loads that do not
// trust a non-coherent line).  Every wave drains its stores (`s_waitcnt vmcnt(0)`) before the workgroup publishes its
// slot, so no agent-scope release / acquire -- an L2 write-back and a full L2 invalidate per merge -- is needed, and
// each workgroup's private state (cluster sizes, freshness flags) stays cached.
template <class T> __device__ __forceinline__ T LDG(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void STG(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// One-XCD form (k_linkage_mw<true>): every participating workgroup sits on the SAME XCD (checked in the kernel from
// HW_REG_XCC_ID, not assumed), so that XCD's L2 is the point of coherence: stores stay plain -- they write through the CU's L1
// and KEEP the line in the L2 (an sc1 store drops it, and even a same-XCD reader then pays the cross-XCD round trip) -- while
// loads stay sc1 (bypass the reader's L1, served by the L2).  MI355X_MICROARCH.md, table of store / load flavours.
template <bool ONEX, class T> __device__ __forceinline__ void STX(T* p, T v)
{
    if constexpr (ONEX) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
