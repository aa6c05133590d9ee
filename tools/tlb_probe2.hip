// tools/tlb_probe2.hip -- memory cost of one Lance-Williams pass at N = 100 000 in two matrix layouts (G x 512 threads, every pass another pair):
//   condensed: 2 scattered 8-byte loads + 1 scattered 8-byte store per active row (one matrix row apart)         = k_linkage_mw today
//   square:    2 coalesced row reads + 1 coalesced row write + 1 scattered 8-byte mirror store per column        = full symmetric matrix
//   hipcc --offload-arch=gfx950 -O3 tools/tlb_probe2.hip -o /tmp/tlb_probe2 && /tmp/tlb_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <class T> __device__ __forceinline__ T LDG(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void STG(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_cond(double* D, size_t N, int passes, const unsigned* xs, const unsigned* ys, int drain)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    for (int p = 0; p < passes; ++p) {
        const size_t x = xs[p], y = ys[p];
        for (size_t z = t; z < N / 2; z += nt) {          // rows z < x, y: column accesses (the expensive half; N/2 rows of pitch N doubles inside 40 GB)
            const double a = LDG(&D[z * (N / 2) + x]), b = LDG(&D[z * (N / 2) + y]);
            STG(&D[z * (N / 2) + y], a * 0.5 + b * 0.5);
        }
        if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}
__global__ void k_sq(double* D, size_t N, int passes, const unsigned* xs, const unsigned* ys, int drain)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    for (int p = 0; p < passes; ++p) {
        const size_t x = xs[p], y = ys[p];
        for (size_t z = t; z < N / 2; z += nt) {          // same number of entries as k_cond
            const double a = LDG(&D[x * (N / 2) + z]), b = LDG(&D[y * (N / 2) + z]);
            const double nd = a * 0.5 + b * 0.5;
            STG(&D[y * (N / 2) + z], nd);
            STG(&D[z * (N / 2) + y], nd);
        }
        if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}
int main()
{
    const size_t N = 100000;
    const size_t bytes = (N / 2) * (N / 2) * 8 * 2;              // 40 GB
    double* D; if (hipMalloc(&D, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(D, 0, bytes);
    const int passes = 2000;
    std::vector<unsigned> xs(passes), ys(passes);
    for (int i = 0; i < passes; ++i) { xs[i] = rand() % 49000; ys[i] = rand() % 49000; }
    unsigned *dx, *dy; hipMalloc(&dx, passes * 4); hipMalloc(&dy, passes * 4);
    hipMemcpy(dx, xs.data(), passes * 4, hipMemcpyHostToDevice); hipMemcpy(dy, ys.data(), passes * 4, hipMemcpyHostToDevice);
    for (int drain = 0; drain < 2; ++drain)
        for (int G : {64, 128, 256})
            for (int sq = 0; sq < 2; ++sq) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                if (sq) hipLaunchKernelGGL(k_sq, dim3(G), dim3(512), 0, 0, D, N, 20, dx, dy, drain); else hipLaunchKernelGGL(k_cond, dim3(G), dim3(512), 0, 0, D, N, 20, dx, dy, drain);
                hipEventRecord(e0);
                if (sq) hipLaunchKernelGGL(k_sq, dim3(G), dim3(512), 0, 0, D, N, passes, dx, dy, drain); else hipLaunchKernelGGL(k_cond, dim3(G), dim3(512), 0, 0, D, N, passes, dx, dy, drain);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                printf("%s G=%3d drain=%d  %.2f us per pass over %zu entries\n", sq ? "square   " : "condensed", G, drain, ms * 1e3 / passes, N / 2);
            }
    return 0;
}
