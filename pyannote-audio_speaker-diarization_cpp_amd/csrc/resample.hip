// resample.hip -- SURVEY 8(f) row 2, the resample leg: Resampler::Resample (frontend/resampler.cc:19-36), i.e. libsamplerate 0.2.2's
// src_simple(SRC_SINC_BEST_QUALITY) with ratio out_sr / in_sr (pinned by pipeline/cmake/samplerate.cmake; the library itself is a
// FetchContent download and NOT in the reference checkout).
//
// What is kept of the reference: the call's contract -- mono float in, mono float out, ratio = out_sr / in_sr held in a FLOAT, output
// length (size_t)(in.size() * ratio) evaluated in float exactly as resampler.cc:21-22 does, output sample m taken at input time
// m * in_sr / out_sr with silence before the first and after the last input sample (src_simple: one-shot, end_of_input = 1).
// What cannot be kept: libsamplerate's arithmetic is a 340 239-entry coefficient table (src_sinc.c / high_qual_coeffs.h) that is
// third-party data absent here, so sample values are NOT comparable with it ("parity unpinned" for this leg).  The
// published algorithm is restated instead -- band-limited interpolation with a windowed sinc whose cutoff follows the LOWER of the
// two rates (J. O. Smith, "Digital Audio Resampling", the method libsamplerate's documentation names) -- in its exact polyphase
// form for the rational ratio L / M = out_sr / in_sr (no table interpolation error):
//     y[m] = sum_k x[k] * h(m * M - k * L),   h(u) = L * 2 fc * sinc(2 fc u) * kaiser_beta(u / half),  |u| <= half
//     fc = RS_CUTOFF / (2 * max(L, M))  (cycles per sample of the rate in_sr * L),  half = RS_ZEROS / (2 fc)
// with RS_ZEROS = 64 zero crossings per wing, RS_CUTOFF = 0.95 of the lower Nyquist frequency, Kaiser beta = 10.056 (100 dB).
// oracle/resample_oracle.py evaluates the same definition in float64 with numpy; scipy.signal.resample_poly applies the same taps
// independently.  One thread per output sample; the input span of a 256-output tile is staged in LDS (coalesced), the tap table
// [tap][phase] (built on the host in double, stored as float) stays in L2: a wave reads 64 phases of one tap row.
#include "common.h"
#include <cmath>

#define RS_ZEROS 64
#define RS_CUTOFF 0.95
#define RS_BETA 10.056
#define RS_BLOCK 256

static int64_t gcd64(int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; }

static double bessel_i0(double x)
{
    double s = 1.0, t = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 200; ++k) { t *= q / ((double)k * (double)k); s += t; if (t < 1e-17 * s) break; }
    return s;
}

struct ResamplePlan { int64_t L, M; int J; int64_t span; double half, fc; };

static int make_plan(int32_t in_sr, int32_t out_sr, ResamplePlan& p)
{
    if (in_sr <= 0 || out_sr <= 0) return 1;
    const int64_t g = gcd64(in_sr, out_sr);
    p.L = out_sr / g; p.M = in_sr / g;
    const double fmax = (double)(p.L > p.M ? p.L : p.M);
    p.fc = RS_CUTOFF / (2.0 * fmax);
    p.half = (double)RS_ZEROS / (2.0 * p.fc);
    p.J = (int)std::floor(p.half / (double)p.L) + 1;
    p.span = ((RS_BLOCK - 1) * p.M) / p.L + 2 * (int64_t)p.J + 3;      // input samples a tile of RS_BLOCK outputs can touch
    return 0;
}

extern "C" int64_t sd_resample_len(int64_t n, int32_t in_sr, int32_t out_sr)
{
    if (n < 0 || in_sr <= 0 || out_sr <= 0) return -1;
    const float ratio = (float)(1.0 * out_sr / in_sr);                  // resampler.cc:21 (float ratio)
    return (int64_t)((float)(uint64_t)n * ratio);                       // resampler.cc:22: size_t * float -> float -> size_t
}

// y[m], m in [0, n_out): thread per output; xs = input samples [base, base + span) of the tile, zero outside [0, n_in)
__global__ __launch_bounds__(RS_BLOCK) void k_resample(const float* __restrict__ x, int64_t n_in, float* __restrict__ y, int64_t n_out,
                                                        const float* __restrict__ taps /*[2J+1][L]*/, int L, int M, int J, int span)
{
    extern __shared__ float xs[];
    const int64_t m0 = (int64_t)blockIdx.x * RS_BLOCK;
    const int64_t base = (m0 * M) / L - J - 1;
    for (int i = threadIdx.x; i < span; i += RS_BLOCK) {
        const int64_t g = base + i;
        xs[i] = (g >= 0 && g < n_in) ? x[g] : 0.0f;
    }
    __syncthreads();
    const int64_t m = m0 + threadIdx.x;
    if (m >= n_out) return;
    const int64_t t = m * M;
    const int64_t i0 = t / L;
    const int ph = (int)(t - i0 * L);
    const float* xp = xs + (i0 - base);                                 // xp[-j] = x[i0 - j]
    const float* tp = taps + ph;
    float acc0 = 0.0f, acc1 = 0.0f;
    int j = -J;
    for (; j + 1 <= J; j += 2) {
        acc0 = fmaf(tp[(size_t)(j + J) * L], xp[-j], acc0);
        acc1 = fmaf(tp[(size_t)(j + J + 1) * L], xp[-j - 1], acc1);
    }
    if (j <= J) acc0 = fmaf(tp[(size_t)(j + J) * L], xp[-j], acc0);
    y[m] = acc0 + acc1;
}

// device-resident form used by sd_diarize_wav; d_out must hold sd_resample_len(n, in_sr, out_sr) floats
int resample_dev(sd_ctx* c, const float* d_in, int64_t n, int32_t in_sr, int32_t out_sr, float* d_out, int64_t n_out)
{
    ResamplePlan p;
    if (make_plan(in_sr, out_sr, p)) SD_FAIL(c, SD_ERR_ARG, "sd_resample: bad sample rate %d -> %d", in_sr, out_sr);
    const int64_t ntap = 2 * (int64_t)p.J + 1;
    if (ntap * p.L > (int64_t)64 << 20) SD_FAIL(c, SD_ERR_ARG, "sd_resample: %d -> %d Hz needs a %lld x %lld tap table (ratio %lld / %lld): unsupported rate pair",
                                                  in_sr, out_sr, (long long)ntap, (long long)p.L, (long long)p.L, (long long)p.M);
    if (p.span * (int64_t)sizeof(float) > 160 * 1024) SD_FAIL(c, SD_ERR_ARG, "sd_resample: %d -> %d Hz: a tile's input span (%lld samples) does not fit the LDS", in_sr, out_sr, (long long)p.span);
    if (n_out <= 0) return SD_OK;
    // tap table, double on the host, float on the device; cached per rate pair
    char key[64];
    snprintf(key, sizeof(key), "rs_taps_%d_%d", in_sr, out_sr);
    // "filled" is recorded only after the upload has completed: a failed allocation or copy leaves no entry behind that a later call would trust
    const bool have = c->rs_taps_filled.count(key) != 0;
    WS(c, float, d_taps, key, ntap * p.L);
    if (!have) {
        std::vector<float> h((size_t)(ntap * p.L));
        const double i0b = bessel_i0(RS_BETA);
        for (int64_t jj = 0; jj < ntap; ++jj)
            for (int64_t ph = 0; ph < p.L; ++ph) {
                const double u = (double)ph + (double)(jj - p.J) * (double)p.L;
                double v = 0.0;
                if (std::fabs(u) <= p.half) {
                    const double a = 2.0 * p.fc * u, r = u / p.half;
                    const double sinc = a == 0.0 ? 1.0 : std::sin(M_PI * a) / (M_PI * a);
                    v = (double)p.L * 2.0 * p.fc * sinc * bessel_i0(RS_BETA * std::sqrt(1.0 - r * r > 0 ? 1.0 - r * r : 0.0)) / i0b;
                }
                h[(size_t)(jj * p.L + ph)] = (float)v;
            }
        HIPCHK(c, hipMemcpyAsync(d_taps, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));                      // h goes out of scope
        c->rs_taps_filled.insert(key);
    }
    const size_t lds = (size_t)p.span * sizeof(float);
    if (lds > 64 * 1024) HIPCHK(c, hipFuncSetAttribute((const void*)k_resample, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t blocks = (n_out + RS_BLOCK - 1) / RS_BLOCK;
    {
        ProfScope ps(c, "resample", 2.0 * (double)ntap * (double)n_out, 4.0 * (double)(n + n_out));
        hipLaunchKernelGGL(k_resample, dim3((unsigned)blocks), dim3(RS_BLOCK), lds, c->stream, d_in, n, d_out, n_out, d_taps, (int)p.L, (int)p.M, p.J, (int)p.span);
    }
    KCHECK(c);
    return SD_OK;
}

extern "C" int sd_resample(sd_ctx* c, const float* wav, int64_t n, int32_t in_sr, int32_t out_sr, float* out, int64_t cap, int64_t* n_out)
{
    if (!c) return SD_ERR_ARG;
    c->err.clear();
    if (hipSetDevice(c->device) != hipSuccess) SD_FAIL(c, SD_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    if (!wav || n <= 0 || !n_out) SD_FAIL(c, SD_ERR_ARG, "sd_resample: bad argument");
    const int64_t no = sd_resample_len(n, in_sr, out_sr);
    if (no < 0) SD_FAIL(c, SD_ERR_ARG, "sd_resample: bad sample rate %d -> %d", in_sr, out_sr);
    *n_out = no;
    if (!out) return SD_OK;                                              // length query
    if (cap < no) SD_FAIL(c, SD_ERR_ARG, "sd_resample: output buffer holds %lld samples, %lld needed", (long long)cap, (long long)no);
    WS(c, float, d_in, "rs_in", n);
    WS(c, float, d_out, "rs_out", no > 0 ? no : 1);
    HIPCHK(c, hipMemcpyAsync(d_in, wav, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    int rc;
    if ((rc = resample_dev(c, d_in, n, in_sr, out_sr, d_out, no))) return rc;
    if (no > 0) HIPCHK(c, hipMemcpyAsync(out, d_out, (size_t)no * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    sd_flush_profile(c);
    return SD_OK;
}
