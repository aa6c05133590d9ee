// reconstruct.hip -- from hard clusters back to speaker turns (compiled with -ffp-contract=off):
//   a15 reconstruct / max_segmentation_cluster      sd.cpp:2767-2848
//   a16 to_diarization (aggregate skip_average, crop_segment, stable argsort top-count)  sd.cpp:2638-2764, 2567-2635
//   a17 to_annotation + Track::support + finalResult  sd.cpp:2852-2935, 911-941, 962-978   (host)
// The reference materialises a NaN-filled [chunks][293][K] double tensor and overlap-adds it
// with nested-vector loops; here each (frame, cluster) output gathers the <= 11 chunks that
// cover it straight from the raw f32 scores (deterministic chunk order, no atomics), and a second
// kernel picks the count[t] most active clusters per frame.
#include "common.h"
#include <algorithm>
#include <cfloat>
#include <cmath>

static const double kFrameStep = 0.016875, kFrameDur = 0.016875;      // sd.cpp:2430-2431

// act[f][k] = sum over chunks c covering frame f of max_{s: hard[c][s]==k} seg[c][f-sfr[c]][s]   (0.0 if none)
__global__ void k_activations(const float* __restrict__ seg, const int* __restrict__ hard, const int* __restrict__ sfr,
                              int64_t chunks, int K, double* __restrict__ act, int64_t nact)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nact * K) return;
    const int64_t f = idx / K;
    const int k = (int)(idx - f * K);
    const double per_chunk = 0.5 / 0.016875;
    int64_t lo = (int64_t)((double)(f - (SD_FRAMES - 1)) / per_chunk) - 2; if (lo < 0) lo = 0;
    int64_t hi = (int64_t)((double)f / per_chunk) + 2; if (hi > chunks - 1) hi = chunks - 1;
    double sum = 0.0;
    for (int64_t c = lo; c <= hi; ++c) {
        const int64_t j = f - sfr[c];
        if (j < 0 || j >= SD_FRAMES) continue;
        const int* h = hard + c * SD_SPEAKERS;
        const float* s = seg + (c * SD_FRAMES + j) * SD_SPEAKERS;
        float mx = -INFINITY; bool any = false;
#pragma unroll
        for (int q = 0; q < SD_SPEAKERS; ++q)
            if (h[q] == k) { any = true; mx = (mx < s[q]) ? s[q] : mx; }      // std::max semantics, sd.cpp:2779
        if (any) sum += (double)mx;                                           // NaN entries are masked out, sd.cpp:1197-1201
    }
    act[idx] = sum;                                                           // skip_average, missing = 0.0 (sd.cpp:2651)
}

// binary[i][k] = 1 for the count[i] clusters with the largest activation (stable order on ties), sd.cpp:2720-2759
__global__ void k_topk(const double* __restrict__ act, const int32_t* __restrict__ count, int64_t ar0, int64_t cr0,
                       int64_t rows, int64_t crow, int K, uint8_t* __restrict__ binary)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const double* a = act + (ar0 + i) * K;
    uint8_t* b = binary + i * K;
    for (int k = 0; k < K; ++k) b[k] = 0;
    if (i >= crow) return;
    int c = count[cr0 + i];
    if (c > K) c = K;                                                         // sd.cpp:2681
    for (int j = 0; j < c; ++j) {
        int best = -1; double bv = 0.0;
        for (int k = 0; k < K; ++k) {
            if (b[k]) continue;
            const double v = -1.0 * a[k];                                     // argsort of negated values, sd.cpp:2727
            if (best < 0 || v < bv) { best = k; bv = v; }
        }
        if (best >= 0) b[best] = 1;
    }
}

// SlidingWindow::operator[] (sd.cpp:1092-1115): walks from 0.0, may bail out on short audio
static double sw_index_start(double step, double dur, int64_t num_samples, int pos)
{
    const int window_size = (int)std::round(dur * 16000.0), step_size = (int)std::round(step * 16000.0);
    double start = 0.0; size_t cur = 0; int index = 0;
    while (true) {
        if (index == pos) return start;
        if (cur + (size_t)window_size >= (size_t)num_samples) break;
        start += step; cur += (size_t)step_size; index++;
    }
    return 0.0;
}

// index part of crop_segment (sd.cpp:2567-2618), float intermediates as in the reference
static void crop_range(double w_start, double w_step, double w_dur, int64_t w_ns, double f_start, double f_end,
                       int64_t n_rows, int64_t* r0, int64_t* r1, float* new_start)
{
    const float i_ = (float)((f_start - w_dur - w_start) / w_step);
    int rs = (int)std::ceil(i_);
    if (rs < 0) rs = 0;
    const float j_ = (float)((f_end - w_start) / w_step);
    const int re = (int)std::floor(j_) + 1;
    *new_start = (float)sw_index_start(w_step, w_dur, w_ns, rs);
    const size_t s = (size_t)(double)rs, e = (size_t)(double)re;
    if (s >= (size_t)n_rows) { *r0 = *r1 = 0; return; }
    *r0 = (int64_t)s;
    *r1 = (int64_t)std::min(e, (size_t)n_rows);
    if (*r1 < *r0) *r1 = *r0;
}

// a17 on the host: binary [rows][K] -> turns
static void to_annotation_host(const std::vector<uint8_t>& binary, int64_t rows, int K, double w_start, std::vector<sd_turn>& out)
{
    const double onset = 0.5, offset = 0.5;                                   // sd.cpp:3228-3230
    const float min_off_f = 0.5817029604921046;                               // float at the call site, sd.cpp:3210
    const double min_off = (double)min_off_f;
    out.clear();
    if (rows <= 0) return;
    std::vector<double> ts((size_t)rows);
    for (int64_t i = 0; i < rows; ++i) {
        const double s = w_start + (double)i * kFrameStep, e = s + kFrameDur;  // sd.cpp:2865-2867
        ts[(size_t)i] = (s + e) / 2;
    }
    std::vector<sd_turn> segs;
    for (int k = 0; k < K; ++k) {
        segs.clear();
        double start = ts[0];
        bool active = (double)binary[(size_t)k] > onset;
        for (int64_t j = 1; j < rows; ++j) {
            const double v = (double)binary[(size_t)j * K + k];
            if (active) {
                if (v < offset) { segs.push_back({start, ts[(size_t)j], k, 0}); start = ts[(size_t)j]; active = false; }
            } else if (v > onset) { start = ts[(size_t)j]; active = true; }
        }
        if (active) segs.push_back({start, ts[(size_t)rows - 1], k, 0});
        if (segs.empty()) continue;
        if (min_off > 0.0) {                                                  // Track::support, sd.cpp:911-941
            std::sort(segs.begin(), segs.end(), [](const sd_turn& a, const sd_turn& b) { return a.start < b.start; });
            std::vector<sd_turn> merged;
            sd_turn cur = segs[0];
            for (size_t i = 1; i < segs.size(); ++i) {
                const sd_turn& nx = segs[i];
                double gap;
                if (cur.start < nx.start) gap = (cur.end >= nx.start) ? 0.0 : nx.start - cur.end;      // Segment::gap, sd.cpp:831-855
                else gap = (cur.start <= nx.end) ? 0.0 : cur.start - nx.end;
                if (gap < min_off) { cur.start = std::min(cur.start, nx.start); cur.end = std::max(cur.end, nx.end); }
                else { merged.push_back(cur); cur = nx; }
            }
            merged.push_back(cur);
            segs.swap(merged);
        }
        // min_duration_on = 0.0 -> removeShort is never called (sd.cpp:2929)
        for (auto& s : segs) out.push_back(s);
    }
    // Annotation::finalResult: std::sort by start (sd.cpp:973) -- same algorithm, same input order as the reference
    std::sort(out.begin(), out.end(), [](const sd_turn& a, const sd_turn& b) { return a.start < b.start; });
}

// hard: host [chunks][3] with -2 for inactive local speakers (sd.cpp:3172-3191 applied by the caller)
int run_reconstruct(sd_ctx* c, const float* d_seg, const int* /*d_nact*/, const int* d_hard, const int32_t* d_count,
                    int64_t n_count, int64_t chunks, int64_t n_samples, int K, std::vector<sd_turn>& turns)
{
    turns.clear();
    c->stash.rows = 0;
    if (chunks <= 0 || K <= 0) return SD_OK;
    // activations grid: frames window (0.0, step, dur) (sd.cpp:1233), chunk windows (0.0, 0.5, 5.0)
    const double target = 0.0 + 5.0 + (double)(chunks - 1) * 0.5;
    const int64_t nact = closest_frame_host(0.0, kFrameStep, kFrameDur, target) + 1;
    std::vector<int> sfr((size_t)chunks);
    double start = 0.0;
    for (int64_t i = 0; i < chunks; ++i) { sfr[(size_t)i] = (int)closest_frame_host(0.0, kFrameStep, kFrameDur, start); start += 0.5; }
    WS(c, int, d_sfr, "rc_sfr", chunks);
    WS(c, double, d_act, "rc_act", nact * K);
    HIPCHK(c, hipMemcpyAsync(d_sfr, sfr.data(), (size_t)chunks * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {
        ProfScope ps(c, "activations", 0, (double)chunks * SD_FRAMES * 3 * 4.0 + (double)nact * K * 8.0);
        hipLaunchKernelGGL(k_activations, dim3((unsigned)((nact * K + 255) / 256)), dim3(256), 0, c->stream, d_seg, d_hard, d_sfr, chunks, K, d_act, nact);
        KCHECK(c);
    }
    // extents and their intersection, sd.cpp:2691-2706.  count window = (0.5, step, dur, num_samples 235)
    const double c_start_w = 0.0 + 0.1 * 5.0;
    const double a_end = 0.0 + (0 - .5) * kFrameStep + .5 * kFrameDur + (double)nact * kFrameStep;
    const double c_end = c_start_w + (0 - .5) * kFrameStep + .5 * kFrameDur + (double)n_count * kFrameStep;
    const double f0 = std::max(0.0, c_start_w), f1 = std::min(a_end, c_end);
    int64_t ar0, ar1, cr0, cr1; float astart, cstart;
    const int Ft = SD_FRAMES - 2 * (int)std::floor((double)SD_FRAMES * 0.1);
    crop_range(0.0, kFrameStep, kFrameDur, n_samples, f0, f1, nact, &ar0, &ar1, &astart);
    crop_range(c_start_w, kFrameStep, kFrameDur, Ft, f0, f1, n_count, &cr0, &cr1, &cstart);
    const int64_t rows = ar1 - ar0;
    int64_t crow = cr1 - cr0;
    if (rows <= 0) return SD_OK;                      // reference would index an empty vector here (sd.cpp:2735)
    { StepStash& S = c->stash; S.ar0 = ar0; S.cr0 = cr0; S.rows = rows; S.crow_all = crow; S.nact = nact; }      // step dumps
    if (crow > rows) crow = rows;                     // reference asserts cropped_count.size() <= rows (sd.cpp:2745)
    WS(c, uint8_t, d_binary, "rc_binary", rows * K);
    {
        ProfScope ps(c, "topk", 0, (double)rows * K * 9.0);
        hipLaunchKernelGGL(k_topk, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, d_act, d_count, ar0, cr0, rows, crow, K, d_binary);
        KCHECK(c);
    }
    std::vector<uint8_t> binary((size_t)rows * K);
    HIPCHK(c, hipMemcpyAsync(binary.data(), d_binary, binary.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    to_annotation_host(binary, rows, K, (double)astart, turns);
    return SD_OK;
}
