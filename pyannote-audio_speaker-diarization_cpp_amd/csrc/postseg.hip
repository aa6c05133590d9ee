// postseg.hip -- post-segmentation stages on the GPU (compiled with -ffp-contract=off):
//   a4  hysteresis binarisation            SegmentModel::binarize_swf/_ndarray, sd.cpp:1506-1639
//   a6  overlap cleaning + mask choice     Helper::cleanSegmentations sd.cpp:710-743, loop 3047-3078
//       + per-(chunk,speaker) active-frame counts for the inactive test, sd.cpp:3172-3191
//   a5  speaker counting                   speaker_count/trim/aggregate/np_rint, sd.cpp:1665-1782, 1167-1311, 260-272
// The reference allocates six [3c x 293] temporaries for a4 and walks c x 235 x 1 nested
// vectors for a5; here a4+a6 are one wave per chunk and a5 is a deterministic gather
// (each output frame reads the <= 9 chunks that cover it; no atomics).
#include "common.h"
#include <cfloat>
#include <cmath>

#define ONSET 0.4442333667381752      /* sd.cpp:1339 */

// ---------------------------------------------------------------- host scalar helpers (a5 geometry)
int sd_np_rint_host(double val)        // numpy rint as the reference restates it, sd.cpp:260-272
{
    const double sgn = val > 0 ? 1.0 : -1.0;
    if (fabs(val - (double)(int)val - 0.5 * sgn) < DBL_EPSILON) {
        const int tmp = (int)round(val);
        return (tmp % 2 == 0) ? tmp : tmp - (int)sgn;
    }
    return (int)round(val);
}
int64_t closest_frame_host(double w_start, double w_step, double w_dur, double t)   // sd.cpp:1084-1090
{
    double closest = (t - w_start - .5 * w_dur) / w_step;
    if (closest < 0.0) closest = 0.0;
    return (int64_t)sd_np_rint_host(closest);
}
static const double kFrameStep = 0.016875, kFrameDur = 0.016875;     // sd.cpp:2430-2431
int64_t count_frames_host(int64_t chunks)
{
    if (chunks <= 0) return 0;
    const double t_start = 0.0 + 0.1 * 5.0, t_dur = (1 - 0.1 - 0.1) * 5.0;     // trim(), sd.cpp:1776-1778
    const double target = t_start + t_dur + (double)(chunks - 1) * 0.5;         // sd.cpp:1232
    return closest_frame_host(t_start, kFrameStep, kFrameDur, target) + 1;
}
extern "C" int64_t sd_count_frames(int64_t chunks) { return count_frames_host(chunks); }

// ---------------------------------------------------------------- k_binarize_masks : one wave per chunk
__global__ __launch_bounds__(64) void k_binarize_masks(const float* __restrict__ seg, int64_t chunks, uint8_t* __restrict__ bin,
                                                       float* __restrict__ masks, int* __restrict__ nact)
{
    __shared__ uint8_t sb[SD_FRAMES * SD_SPEAKERS];
    const int64_t ck = blockIdx.x;
    const int lane = threadIdx.x;
    const float* s = seg + ck * SD_FRAMES * SD_SPEAKERS;
    if (lane < SD_SPEAKERS) {
        bool have = false, last = false;
        for (int f = 0; f < SD_FRAMES; ++f) {
            const double v = (double)s[f * SD_SPEAKERS + lane];
            bool b;
            if (fabs(v - ONSET) < DBL_EPSILON) b = have ? last : false;      // initial_state = false
            else { b = v > ONSET; have = true; last = b; }
            sb[f * SD_SPEAKERS + lane] = b ? 1 : 0;
        }
    }
    __syncthreads();
    int act[3] = {0, 0, 0}, cln[3] = {0, 0, 0};
    for (int f = lane; f < SD_FRAMES; f += 64) {
        const int b0 = sb[f * 3], b1 = sb[f * 3 + 1], b2 = sb[f * 3 + 2];
        const bool keep = (b0 + b1 + b2) < 2;                                 // sd.cpp:730
        act[0] += b0; act[1] += b1; act[2] += b2;
        if (keep) { cln[0] += b0; cln[1] += b1; cln[2] += b2; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int o = 32; o > 0; o >>= 1) { act[k] += __shfl_xor(act[k], o); cln[k] += __shfl_xor(cln[k], o); }
    // sd.cpp:3017: min_num_frames = ceil(293*640/(5*16000)) = 3 ; clean mask used iff its sum > 3
    const int min_num_frames = 3;
    for (int f = lane; f < SD_FRAMES; f += 64) {
        const int b[3] = {sb[f * 3], sb[f * 3 + 1], sb[f * 3 + 2]};
        const bool keep = (b[0] + b[1] + b[2]) < 2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (bin) bin[(ck * SD_FRAMES + f) * 3 + k] = (uint8_t)b[k];
            if (masks) {
                const bool use_clean = cln[k] > min_num_frames;
                masks[(ck * 3 + k) * SD_FRAMES + f] = (float)(use_clean ? (keep ? b[k] : 0) : b[k]);
            }
        }
    }
    if (nact && lane < 3) nact[ck * 3 + lane] = act[lane];
}

// ---------------------------------------------------------------- k_count : gather form of aggregate()
__device__ __forceinline__ int dev_np_rint(double val)
{
    const double sgn = val > 0 ? 1.0 : -1.0;
    if (fabs(val - (double)(int)val - 0.5 * sgn) < DBL_EPSILON) {
        const int tmp = (int)round(val);
        return (tmp % 2 == 0) ? tmp : tmp - (int)sgn;
    }
    return (int)round(val);
}

__global__ void k_count(const uint8_t* __restrict__ bin, const int* __restrict__ sfr, int64_t chunks, int nl, int Ft,
                        int32_t* __restrict__ count, int64_t n_count, double* __restrict__ avg)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_count) return;
    // chunk c covers frames [sfr[c], sfr[c]+Ft); sfr[c] = rint(c * 0.5/0.016875 - 0.5): bracket c generously
    const double per_chunk = 0.5 / 0.016875;
    int64_t lo = (int64_t)((double)(f - (Ft - 1)) / per_chunk) - 2; if (lo < 0) lo = 0;
    int64_t hi = (int64_t)((double)f / per_chunk) + 2; if (hi > chunks - 1) hi = chunks - 1;
    double sum = 0.0, cnt = 0.0;
    for (int64_t c = lo; c <= hi; ++c) {
        const int64_t j = f - sfr[c];
        if (j < 0 || j >= Ft) continue;
        const uint8_t* b = bin + (c * SD_FRAMES + (j + nl)) * SD_SPEAKERS;
        sum += (double)(b[0] + b[1] + b[2]);                                  // sd.cpp:1707-1712, 1260
        cnt += 1.0;
    }
    double v = sum / fmax(cnt, DBL_EPSILON);                                  // sd.cpp:1288
    if (cnt == 0.0) v = 0.0;                                                  // missing = 0.0, sd.cpp:1302-1305, 1720
    count[f] = dev_np_rint(v);                                                // sd.cpp:1734
    if (avg) avg[f] = v;                                                      // step dumps: count_data, the value in front of np_rint (sd.cpp:1723)
}

int run_postseg(sd_ctx* c, const float* d_seg, int64_t chunks, uint8_t* d_bin, float* d_masks, int* d_nact)
{
    if (chunks <= 0) return SD_OK;
    ProfScope ps(c, "binarize_masks", 0, (double)chunks * SD_FRAMES * 3 * (4.0 + 1.0 + 4.0));
    hipLaunchKernelGGL(k_binarize_masks, dim3((unsigned)chunks), dim3(64), 0, c->stream, d_seg, chunks, d_bin, d_masks, d_nact);
    KCHECK(c);
    return SD_OK;
}

int run_count(sd_ctx* c, const uint8_t* d_bin, int64_t chunks, int32_t* d_count, int64_t n_count, double* d_avg)
{
    if (chunks <= 0 || n_count <= 0) return SD_OK;
    const int nl = (int)floor((double)SD_FRAMES * 0.1);                        // sd.cpp:1755
    const int nr = (int)floor((double)SD_FRAMES * 0.1);
    const int Ft = SD_FRAMES - nl - nr;
    const double t_start = 0.0 + 0.1 * 5.0;
    std::vector<int> sfr((size_t)chunks);
    double start = t_start;
    for (int64_t i = 0; i < chunks; ++i) {                                     // sd.cpp:1248-1253
        sfr[(size_t)i] = (int)closest_frame_host(t_start, kFrameStep, kFrameDur, start);
        start += 0.5;
    }
    WS(c, int, d_sfr, "cnt_sfr", chunks);
    HIPCHK(c, hipMemcpyAsync(d_sfr, sfr.data(), (size_t)chunks * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));     // sfr is a stack-lifetime host vector
    ProfScope ps(c, "count", 0, (double)n_count * 4.0 + (double)chunks * SD_FRAMES * 3);
    hipLaunchKernelGGL(k_count, dim3((unsigned)((n_count + 255) / 256)), dim3(256), 0, c->stream, d_bin, d_sfr, chunks, nl, Ft, d_count, n_count, d_avg);
    KCHECK(c);
    return SD_OK;
}
