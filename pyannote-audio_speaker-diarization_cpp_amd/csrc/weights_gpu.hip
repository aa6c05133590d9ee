// weights_gpu.hip -- the fp16 weight forms of the fp16 / x3 modes, built ON THE GPU from the resident f32 weights the first time a mode that
// reads them is selected (sd_set_option "ecapa_precision"): the default f32 mode pays nothing for them at start-up (sd_create 581 -> 90 ms).
// Kept apart from weights.cpp, which stays host-only code: the model-file parsers are also built for the CPU with AddressSanitizer + UBSan
// (tools/sanitize/build.sh).
#include "common.h"
#include <cmath>

// ---- the fp16 forms of the per-frame ECAPA layers, derived from the resident f32 weights on the GPU (same bits as a host conversion:
// round-to-nearest-even to fp16, exact residue, exact power-of-two scale)
// planes [0, K): W rounded to fp16; planes [K, 2K): the rounding residue W - (float)hi, again in fp16 (mode ecapa_precision = 2 runs every
// tap twice, once against each plane: fp16 MFMA with 22-bit weights); rows padded to a multiple of 64 input channels
__global__ void k_w16_planes(const float* __restrict__ W, _Float16* __restrict__ W16, int K, int Cout, int CinPad, int CinPad16, int cin)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rows = (size_t)K * Cout;
    if (idx >= rows * (size_t)cin) return;
    const size_t r = idx / (size_t)cin; const int i = (int)(idx - r * (size_t)cin);
    const float v = W[r * CinPad + i];
    const _Float16 hi = (_Float16)v;
    W16[r * CinPad16 + i] = hi;
    W16[(rows + r) * CinPad16 + i] = (_Float16)(v - (float)hi);
}
__global__ void k_w_absmax(const float* __restrict__ W, size_t rows, int CinPad, int cin, unsigned* __restrict__ out)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.0f;
    if (idx < rows * (size_t)cin) {
        const size_t r = idx / (size_t)cin; const int i = (int)(idx - r * (size_t)cin);
        const float v = W[r * CinPad + i];
        if (isfinite(v)) m = fabsf(v);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(out, __float_as_uint(m));       // non-negative floats order like their bit patterns
}
// pack_split_weights on the GPU: hi / lo halves of W * sc in the staging layout of the x3 kernels
__global__ void k_w16x(const float* __restrict__ W, _Float16* __restrict__ out, int K, int Cout, int CinPad, int cin, float sc)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rows = (size_t)K * Cout;
    if (idx >= rows * (size_t)cin) return;
    const size_t r = idx / (size_t)cin; const int i = (int)(idx - r * (size_t)cin);
    const float v = W[r * CinPad + i] * sc;
    const _Float16 hi = (_Float16)v;
    const size_t at = r * 2 * (size_t)CinPad + (size_t)(i / 32) * 64 + (size_t)((i % 32) / 8) * 16 + (size_t)(i % 8);
    out[at] = hi;
    out[at + 8] = (_Float16)(v - (float)hi);
}

int ensure_ecapa_mode_weights(sd_ctx* c, int mode)
{
    EcapaWeights& E = c->ew;
    if (!E.loaded) return SD_OK;
    const bool need16 = (mode == 1 || mode == 2) && !E.have16, need16x = mode == 3 && !E.have16x;
    if (!need16 && !need16x) return SD_OK;
    unsigned* d_max = nullptr;
    if (need16x) { d_max = (unsigned*)weight_alloc(c, sizeof(unsigned)); if (!d_max) return SD_ERR_HIP; }
    for (ConvLayer* Lp : E.conv16) {
        ConvLayer& L = *Lp;
        const size_t rows = (size_t)L.KT * L.Cout, n = rows * (size_t)L.Cin;
        const dim3 grid((unsigned)((n + 255) / 256)), block(256);
        if (need16) {
            const size_t bytes = (size_t)2 * rows * L.CinPad16 * sizeof(_Float16);
            _Float16* d = (_Float16*)weight_alloc(c, bytes);
            if (!d) return SD_ERR_HIP;
            HIPCHK(c, hipMemsetAsync(d, 0, bytes, c->stream));
            hipLaunchKernelGGL(k_w16_planes, grid, block, 0, c->stream, L.W, d, L.KT, L.Cout, L.CinPad, L.CinPad16, L.Cin);
            KCHECK(c);
            L.W16 = d;
        }
        if (need16x) {
            // 2^e puts the layer's largest weight into [2^13, 2^14) (pack_split_weights above: same rule, same bits)
            HIPCHK(c, hipMemsetAsync(d_max, 0, sizeof(unsigned), c->stream));
            hipLaunchKernelGGL(k_w_absmax, grid, block, 0, c->stream, L.W, rows, L.CinPad, L.Cin, d_max);
            KCHECK(c);
            unsigned bits = 0;
            HIPCHK(c, hipMemcpyAsync(&bits, d_max, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            float wmax; memcpy(&wmax, &bits, 4);
            int e = 0;
            if (wmax > 0.0f) { (void)frexpf(wmax, &e); e = 14 - e; }
            const size_t bytes = (size_t)2 * rows * L.CinPad * sizeof(_Float16);
            _Float16* d = (_Float16*)weight_alloc(c, bytes);
            if (!d) return SD_ERR_HIP;
            HIPCHK(c, hipMemsetAsync(d, 0, bytes, c->stream));
            hipLaunchKernelGGL(k_w16x, grid, block, 0, c->stream, L.W, d, L.KT, L.Cout, L.CinPad, L.Cin, ldexpf(1.0f, e));
            KCHECK(c);
            L.W16x = d; L.w16x_inv = ldexpf(1.0f, -e);
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (need16) E.have16 = true;
    if (need16x) E.have16x = true;
    return SD_OK;
}

