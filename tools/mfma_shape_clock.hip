// mfma_shape_clock.hip -- what the chip delivers on bare MFMA loops of the two shapes per data type (random operands in registers, no memory
// traffic): FLOP/s and the in-kernel clock (s_memtime / s_memrealtime x 100 MHz).  MI355X_MICROARCH.md, DVFS give-back item 7.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_shape_clock tools/mfma_shape_clock.hip ; tools/bin/mfma_shape_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int SHAPE>   // 0: f32 32x32x2, 1: f32 16x16x4, 2: f16 32x32x16, 3: f16 16x16x32
__global__ __launch_bounds__(512) void k_loop(const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = src[(t * 16 + i) & 0xfffff]; b[i] = src[(t * 16 + 8 + i) & 0xfffff]; }
    half8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)a[i]; hb[i] = (_Float16)b[i]; }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    f32x4 d[16];
    for (int i = 0; i < 16; ++i) d[i] = f32x4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (SHAPE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 1) & 7], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 7], b[u], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 2) & 7], b[u], c3, 0, 0, 0);
            }
        } else if constexpr (SHAPE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) d[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + j) & 7], b[(u + 3 * j) & 7], d[j], 0, 0, 0);
        } else if constexpr (SHAPE == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, ha, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, hb, c3, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) d[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16((j & 1) ? ha : hb, (j & 2) ? ha : hb, d[j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + d[i][0] + d[i][1] + d[i][2] + d[i][3];
    out[t] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main()
{
    const int blocks = 256, threads[2] = {256, 512};
    std::vector<float> h(1 << 20);
    srand(7);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.0f - 1.0f;
    float *src, *out; unsigned long long* clk;
    hipMalloc(&src, h.size() * 4); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&clk, 512 * 16);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* names[4] = {"f32 32x32x2", "f32 16x16x4", "f16 32x32x16", "f16 16x16x32"};
    // flops per loop iteration per wave: shape 0: 32 MFMAs x 32*32*2*2; 1: 128 x 16*16*4*2; 2: 32 x 32*32*16*2; 3: 128 x 16*16*32*2
    const double fl[4] = {32.0 * 4096, 128.0 * 2048, 32.0 * 32768, 128.0 * 16384};
    for (int wv = 0; wv < 2; ++wv)
        for (int rep = 0; rep < 2; ++rep)
            for (int sh = 0; sh < 4; ++sh) {
                const int iters = sh < 2 ? 160000 : 640000;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                float ms = 0;
                for (int pass = 0; pass < 2; ++pass) {           // the first pass brings the chip to the clock it holds under this load
                    hipEventRecord(e0);
                    for (int l = 0; l < 3; ++l) {
                        if (sh == 0) hipLaunchKernelGGL(k_loop<0>, dim3(blocks), dim3(threads[wv]), 0, 0, src, out, clk, iters);
                        if (sh == 1) hipLaunchKernelGGL(k_loop<1>, dim3(blocks), dim3(threads[wv]), 0, 0, src, out, clk, iters);
                        if (sh == 2) hipLaunchKernelGGL(k_loop<2>, dim3(blocks), dim3(threads[wv]), 0, 0, src, out, clk, iters);
                        if (sh == 3) hipLaunchKernelGGL(k_loop<3>, dim3(blocks), dim3(threads[wv]), 0, 0, src, out, clk, iters);
                    }
                    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                }
                unsigned long long hc[4]; hipMemcpy(hc, clk + 2 * 100, 16, hipMemcpyDeviceToHost);
                const double waves = blocks * threads[wv] / 64.0;
                const double tf = 3.0 * iters * fl[sh] * waves / (ms * 1e-3) / 1e12;
                printf("%-13s %d waves/SIMD  %8.1f ms  %8.1f TFLOP/s   in-kernel clock %.2f GHz (cycles per loop iteration %.1f)\n", names[sh], threads[wv] / 256, ms, tf,
                       (double)hc[0] / (double)hc[1] * 0.1, (double)hc[0] / iters);
            }
    return 0;
}
