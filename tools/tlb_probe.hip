// tools/tlb_probe.hip -- what one Lance-Williams pass of k_linkage_mw costs in memory latency: N scattered 8-byte loads by G x T threads,
//   (a) "column": one load per matrix row (stride = row pitch, 100 000 doubles -> every load in another 2 MB page of a 40 GB matrix),
//   (b) "tile":   the same number of loads, one per 2 KB (rows of a 256 x 256 tile: 64 distinct cache lines per wave, 391 pages per pass),
//   (c) "dense":  one load per cache line inside 6.4 MB.
// Every pass uses another column, so no pass finds its lines in a cache.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/tlb_probe.hip -o /tmp/tlb_probe && /tmp/tlb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_probe(const double* D, size_t stride, size_t n_loads, int passes, const unsigned* cols, double* sink, int sc1)
{
    double acc = 0;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    for (int p = 0; p < passes; ++p) {
        const size_t col = cols[p];
        for (size_t i = t; i < n_loads; i += nt) {
            const double* q = D + i * stride + col;
            acc += sc1 ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
        }
        __syncthreads();
    }
    if (acc == 12345.678) sink[0] = acc;
}
int main()
{
    const size_t N = 100000;
    const size_t bytes = N * N / 2 * 8;                          // the condensed matrix of the 8 h job: 40 GB
    double* D; if (hipMalloc(&D, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(D, 0, bytes);
    const int passes = 2000;
    std::vector<unsigned> cols(passes); for (auto& c : cols) c = (unsigned)(rand() % 200);
    unsigned* dc; hipMalloc(&dc, passes * 4); hipMemcpy(dc, cols.data(), passes * 4, hipMemcpyHostToDevice);
    double* sink; hipMalloc(&sink, 8);
    struct { const char* name; size_t stride; } pat[3] = {{"column (row pitch 50 000 doubles: one 2 MB page per load)", 50000}, {"tile (2 KB apart)", 256}, {"dense (64 B apart)", 8}};
    for (int sc1 = 0; sc1 < 2; ++sc1)
        for (int G : {32, 128})
            for (auto& pt : pat) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipLaunchKernelGGL(k_probe, dim3(G), dim3(512), 0, 0, D, pt.stride, N, 50, dc, sink, sc1);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_probe, dim3(G), dim3(512), 0, 0, D, pt.stride, N, passes, dc, sink, sc1);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("%s G=%3d %-62s %.2f us per pass of %zu loads\n", sc1 ? "sc1  " : "plain", G, pt.name, ms * 1e3 / passes, N);
            }
    return 0;
}
