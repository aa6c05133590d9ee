// pipeline.cpp -- stage wrappers and the whole-path driver behind the C ABI.
// speakerDiarization() (sd.cpp:2937-3234) re-stated as: scale pcm -> segmentation -> post-seg ->
// embeddings -> [multi-GPU: all-gather here] -> count, clustering, reconstruction, annotation.
#include "common.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>

struct DevTmp {
    void* p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess ? 0 : 1; }
};
#define DTMP(ctx, var, bytes) DevTmp var; if (var.alloc(bytes)) SD_FAIL(ctx, SD_ERR_HIP, "hipMalloc(%zu) failed", (size_t)(bytes))
#define ENTER(ctx) do { if (!(ctx)) return SD_ERR_ARG; (ctx)->err.clear(); if (hipSetDevice((ctx)->device) != hipSuccess) SD_FAIL(ctx, SD_ERR_HIP, "hipSetDevice failed"); } while (0)
#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)

// a1 tail: input_wav[i] = sample * 1.0f / 32768.0 (sd.cpp:2948-2951); division by 2^15 is exact in f32
__global__ void k_pcm_to_f32(const int16_t* __restrict__ pcm, float* __restrict__ wav, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wav[i] = (float)pcm[i] * (1.0f / 32768.0f);
    else if (i < n + 512) wav[i] = 0.0f;          // padding: SincNet's 256-tap rows read 4 samples past the last window (zero weights, finite data)
}
// embeddings are widened to double before clustering (sd.cpp:2555)
__global__ void k_f32_to_f64(const float* __restrict__ a, double* __restrict__ b, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = (double)a[i];
}
__global__ void k_mark_inactive(int* __restrict__ hard, const int* __restrict__ nact, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && nact[i] == 0) hard[i] = -2;                                  // sd.cpp:3172-3191
}

// planted workload (sd_set_planted): rows that are not NaN by the reference's rule take the planted embedding.
// One 192-thread block per row: every thread reads element 0 before anyone overwrites it.
__global__ void k_plant_emb(float* __restrict__ emb, const float* __restrict__ planted)
{
    const size_t i = (size_t)blockIdx.x * SD_EMB_DIM + threadIdx.x;
    const float first = emb[(size_t)blockIdx.x * SD_EMB_DIM];
    const float v = planted[i];
    __syncthreads();
    if (first == first) emb[i] = v;
}

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------ a2+a3
extern "C" int sd_segment_dev(sd_ctx* c, const float* d_wav, int64_t n, float* d_out, int64_t chunks)
{
    ENTER(c);
    if (!d_wav || !d_out || n <= 1) SD_FAIL(c, SD_ERR_ARG, "sd_segment_dev: bad argument");
    if (chunks != sd_num_chunks(n, nullptr)) SD_FAIL(c, SD_ERR_ARG, "sd_segment_dev: chunks must equal sd_num_chunks(n)");
    c->wav_padded = false;                   // the caller's buffer: nothing is known about the bytes behind sample n
    int rc = run_segment(c, d_wav, n, 0, chunks, d_out);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SD_OK;
}

extern "C" int sd_segment(sd_ctx* c, const float* h_wav, int64_t n, float* h_out, int64_t* chunks)
{
    ENTER(c);
    if (!h_wav || !h_out || !chunks) SD_FAIL(c, SD_ERR_ARG, "sd_segment: bad argument");
    const int64_t nc = sd_num_chunks(n, nullptr);
    *chunks = nc;
    if (nc <= 0) SD_FAIL(c, SD_ERR_SHORT, "audio of %lld samples yields no chunk", (long long)n);
    DTMP(c, dw, (n + 512) * sizeof(float)); DTMP(c, ds, nc * SD_FRAMES * 3 * sizeof(float));
    HIPCHK(c, hipMemcpy(dw.p, h_wav, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset((float*)dw.p + n, 0, 512 * sizeof(float)));
    c->wav_padded = true;
    int rc = run_segment(c, (const float*)dw.p, n, 0, nc, (float*)ds.p);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(h_out, ds.p, nc * SD_FRAMES * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return SD_OK;
}

// SegmentModel::infer as the reference declares it (sd.cpp:1352-1404): [rows][T] separate waveforms -> [rows][293][3]; *frames = the
// frames the network yields for T samples (293 for T = 80000), the rest of each row's 293 is zero (slide()'s padding, sd.cpp:1473-1479)
extern "C" int sd_segment_chunks(sd_ctx* c, const float* h_chunks, int64_t rows, int64_t T, float* h_out, int32_t* frames)
{
    ENTER(c);
    if (!h_chunks || !h_out || rows <= 0 || T < 1 || T > SD_CHUNK) SD_FAIL(c, SD_ERR_ARG, "sd_segment_chunks: bad argument (rows >= 1, 1 <= T <= %d)", SD_CHUNK);
    // persistent workspaces, not per-call allocations: slide() calls infer once per batch of 32 chunks (225 times per hour of audio)
    WS(c, float, dw_p, "rows_wav", rows * T + 512); WS(c, float, ds_p, "rows_seg", rows * SD_FRAMES * 3);
    struct { void* p; } dw{dw_p}, ds{ds_p};
    HIPCHK(c, hipMemcpyAsync(dw.p, h_chunks, rows * T * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync((float*)dw.p + rows * T, 0, 512 * sizeof(float), c->stream));
    const bool padded = c->wav_padded;
    c->wav_padded = true;
    int fr = 0;
    const int rc = run_segment_rows(c, (const float*)dw.p, rows, (int)T, (float*)ds.p, &fr);
    c->wav_padded = padded;
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(h_out, ds.p, rows * SD_FRAMES * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (frames) *frames = fr;
    return SD_OK;
}

// ------------------------------------------------------------------ a4-a6
extern "C" int sd_postseg(sd_ctx* c, const float* h_seg, int64_t chunks, uint8_t* h_bin, float* h_masks,
                          int32_t* h_count, int64_t cap_count, int64_t* n_count)
{
    ENTER(c);
    if (!h_seg || chunks <= 0) SD_FAIL(c, SD_ERR_ARG, "sd_postseg: bad argument");
    const int64_t ne = chunks * SD_FRAMES * 3;
    const int64_t nf = count_frames_host(chunks);
    if (n_count) *n_count = nf;
    if (h_count && cap_count < nf) SD_FAIL(c, SD_ERR_ARG, "sd_postseg: count capacity %lld < %lld", (long long)cap_count, (long long)nf);
    DTMP(c, ds, ne * sizeof(float)); DTMP(c, db, ne); DTMP(c, dm, ne * sizeof(float)); DTMP(c, dn, chunks * 3 * sizeof(int)); DTMP(c, dc, nf * sizeof(int32_t));
    HIPCHK(c, hipMemcpy(ds.p, h_seg, ne * sizeof(float), hipMemcpyHostToDevice));
    int rc = run_postseg(c, (const float*)ds.p, chunks, (uint8_t*)db.p, (float*)dm.p, (int*)dn.p);
    if (rc) return rc;
    if ((rc = run_count(c, (const uint8_t*)db.p, chunks, (int32_t*)dc.p, nf))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h_bin) HIPCHK(c, hipMemcpy(h_bin, db.p, ne, hipMemcpyDeviceToHost));
    if (h_masks) HIPCHK(c, hipMemcpy(h_masks, dm.p, ne * sizeof(float), hipMemcpyDeviceToHost));
    if (h_count) HIPCHK(c, hipMemcpy(h_count, dc.p, nf * sizeof(int32_t), hipMemcpyDeviceToHost));
    return SD_OK;
}

// ------------------------------------------------------------------ a12-a14
extern "C" int sd_linkage(sd_ctx* c, const double* h_X, int64_t N, int d, double* h_Z)
{
    ENTER(c);
    if (!h_X || !h_Z || N < 2 || d <= 0) SD_FAIL(c, SD_ERR_ARG, "sd_linkage: need N >= 2");
    DTMP(c, dx, N * d * sizeof(double)); DTMP(c, dz, (N - 1) * 4 * sizeof(double));
    HIPCHK(c, hipMemcpy(dx.p, h_X, N * d * sizeof(double), hipMemcpyHostToDevice));
    int rc = run_linkage(c, (const double*)dx.p, N, d, (double*)dz.p);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(h_Z, dz.p, (N - 1) * 4 * sizeof(double), hipMemcpyDeviceToHost));
    return SD_OK;
}

extern "C" int sd_cluster(sd_ctx* c, const double* h_X, int64_t N, int d, double cutoff, int32_t* h_labels1)
{
    ENTER(c);
    if (!h_X || !h_labels1 || N < 1 || d <= 0) SD_FAIL(c, SD_ERR_ARG, "sd_cluster: bad argument");
    DTMP(c, dx, N * d * sizeof(double));
    HIPCHK(c, hipMemcpy(dx.p, h_X, N * d * sizeof(double), hipMemcpyHostToDevice));
    std::vector<int> lab;
    int rc = run_cluster_labels(c, (const double*)dx.p, N, d, cutoff, lab);
    if (rc) return rc;
    for (int64_t i = 0; i < N; ++i) h_labels1[i] = lab[(size_t)i];
    return SD_OK;
}

// Clustering::fcluster (cl.h:9-10, cl.cpp:442-457; criterion "distance"): Z [N-1][4] of N observations -> 1-based labels [N].  Host
// arithmetic only (O(N) pointer chasing over a dendrogram the caller already holds): no GPU work, c may be NULL.
extern "C" int sd_fcluster(sd_ctx* c, const double* h_Z, int64_t N, double cutoff, int32_t* h_labels1)
{
    if (c) c->err.clear();
    auto fail = [&](const char* m) { if (c) c->err = m; return (int)SD_ERR_ARG; };
    if (!h_labels1 || N < 1 || (N > 1 && !h_Z)) return fail("sd_fcluster: bad argument");
    // a dendrogram of N leaves: merge k joins two DIFFERENT nodes < N + k, each node at most once (the reference indexes arrays with them unchecked)
    std::vector<char> used((size_t)(2 * N - 1), 0);
    for (int64_t k = 0; k + 1 < N; ++k)
        for (int q = 0; q < 2; ++q) {
            const double v = h_Z[k * 4 + q];
            if (!(v >= 0.0) || v >= (double)(N + k) || v != std::floor(v) || used[(size_t)v]) return fail("sd_fcluster: Z is not a dendrogram of N observations");
            used[(size_t)v] = 1;
        }
    std::vector<double> Z(h_Z, h_Z + (size_t)(N > 1 ? N - 1 : 0) * 4);
    std::vector<int> T;
    fcluster_host(Z, N, cutoff, T);
    for (int64_t i = 0; i < N; ++i) h_labels1[i] = T[(size_t)i];
    return SD_OK;
}

extern "C" int sd_clustering_ex(sd_ctx* c, const double* h_emb, int64_t chunks, int d, int num_clusters, int min_clusters, int max_clusters,
                                int32_t* h_hard, int32_t* n_clusters);
extern "C" int sd_clustering(sd_ctx* c, const double* h_emb, int64_t chunks, int d, int32_t* h_hard, int32_t* n_clusters)
{
    return sd_clustering_ex(c, h_emb, chunks, d, -1, -1, -1, h_hard, n_clusters);
}
extern "C" int sd_clustering_ex(sd_ctx* c, const double* h_emb, int64_t chunks, int d, int num_clusters, int min_clusters, int max_clusters,
                                int32_t* h_hard, int32_t* n_clusters)
{
    ENTER(c);
    if (!h_emb || !h_hard || chunks <= 0 || d <= 0) SD_FAIL(c, SD_ERR_ARG, "sd_clustering: bad argument");
    const int64_t M = chunks * SD_SPEAKERS;
    DTMP(c, de, M * d * sizeof(double));
    HIPCHK(c, hipMemcpy(de.p, h_emb, M * d * sizeof(double), hipMemcpyHostToDevice));
    std::vector<int> hard; int K = 1;
    int rc = run_clustering(c, (const double*)de.p, M, d, hard, &K, num_clusters, min_clusters, max_clusters);
    if (rc) return rc;
    for (int64_t i = 0; i < M; ++i) h_hard[i] = hard[(size_t)i];
    if (n_clusters) *n_clusters = K;
    return SD_OK;
}

int turns_out(sd_ctx* c, const std::vector<sd_turn>& v, sd_turn** turns, int64_t* n_turns)
{
    sd_turn* t = (sd_turn*)malloc(sizeof(sd_turn) * (v.size() ? v.size() : 1));
    if (!t) SD_FAIL(c, SD_ERR_ARG, "out of host memory");
    for (size_t i = 0; i < v.size(); ++i) t[i] = v[i];
    *turns = t; *n_turns = (int64_t)v.size();
    return SD_OK;
}

// ------------------------------------------------------------------ a15-a17
extern "C" int sd_reconstruct(sd_ctx* c, const float* h_seg, const uint8_t* h_bin, const int32_t* h_hard, const int32_t* h_count,
                              int64_t n_count, int64_t chunks, int64_t n_samples, sd_turn** turns, int64_t* n_turns)
{
    ENTER(c);
    if (!h_seg || !h_bin || !h_hard || !h_count || !turns || !n_turns || chunks <= 0) SD_FAIL(c, SD_ERR_ARG, "sd_reconstruct: bad argument");
    const int64_t ne = chunks * SD_FRAMES * 3;
    std::vector<int> hard((size_t)chunks * 3);
    int K = 0;
    for (int64_t i = 0; i < chunks; ++i)
        for (int k = 0; k < 3; ++k) {
            int s = 0;
            for (int f = 0; f < SD_FRAMES; ++f) s += h_bin[(i * SD_FRAMES + f) * 3 + k];
            int h = h_hard[i * 3 + k];
            if (s == 0) h = -2;
            hard[(size_t)(i * 3 + k)] = h;
            if (h > K) K = h;
        }
    K += 1;                                                                   // sd.cpp:2803-2812
    DTMP(c, ds, ne * sizeof(float)); DTMP(c, dh, chunks * 3 * sizeof(int)); DTMP(c, dc, (n_count > 0 ? n_count : 1) * sizeof(int32_t));
    HIPCHK(c, hipMemcpy(ds.p, h_seg, ne * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dh.p, hard.data(), chunks * 3 * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dc.p, h_count, n_count * sizeof(int32_t), hipMemcpyHostToDevice));
    std::vector<sd_turn> v;
    int rc = run_reconstruct(c, (const float*)ds.p, nullptr, (const int*)dh.p, (const int32_t*)dc.p, n_count, chunks, n_samples, K, v);
    if (rc) return rc;
    return turns_out(c, v, turns, n_turns);
}

// ------------------------------------------------------------------ sharded inference + finalize
int pcm_to_wav(sd_ctx* c, const int16_t* d_pcm, int64_t n, float** d_wav)
{
    WS(c, float, w, "wav_f32", n + 512);
    hipLaunchKernelGGL(k_pcm_to_f32, GRID1(n + 512), 0, c->stream, d_pcm, w, n);
    KCHECK(c);
    c->wav_padded = true;
    *d_wav = w;
    return SD_OK;
}

int shard_infer(sd_ctx* c, const float* d_wav, int64_t n, int64_t lo, int64_t hi, float* d_seg, float* d_emb)
{
    const int64_t nc = hi - lo;
    c->stash.infer_items = nc > 0 ? nc * SD_SPEAKERS : 0;
    if (nc <= 0) return SD_OK;
    if ((lo * SD_SPEAKERS) % SD_EMB_BATCH != 0) SD_FAIL(c, SD_ERR_ARG, "shard start %lld must be a multiple of 32 chunks", (long long)lo);
    int rc;
    const double t0 = now_ms();
    if ((rc = run_segment(c, d_wav, n, lo, hi, d_seg))) return rc;
    // planted workload: chunks [pa, pb) of this shard take their scores / embeddings from the caller's buffers
    const int64_t pa = std::max(lo, c->planted_lo), pb = std::min(hi, c->planted_lo + c->planted_n);
    if (pb > pa && c->planted_scores)
        HIPCHK(c, hipMemcpyAsync(d_seg + (size_t)(pa - lo) * SD_FRAMES * 3, c->planted_scores + (size_t)(pa - c->planted_lo) * SD_FRAMES * 3,
                                 (size_t)(pb - pa) * SD_FRAMES * 3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    WS(c, float, d_masks, "sh_masks", nc * 3 * SD_FRAMES);
    if ((rc = run_postseg(c, d_seg, nc, nullptr, d_masks, nullptr))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const double t1 = now_ms();
    c->stage_ms[0] += t1 - t0;
    if ((rc = run_embed(c, d_wav, n, d_masks, nc * 3, lo * 3, d_emb))) return rc;
    if (pb > pa && c->planted_emb) {
        hipLaunchKernelGGL(k_plant_emb, dim3((unsigned)((pb - pa) * 3)), dim3(SD_EMB_DIM), 0, c->stream,
                           d_emb + (size_t)(pa - lo) * 3 * SD_EMB_DIM, c->planted_emb + (size_t)(pa - c->planted_lo) * 3 * SD_EMB_DIM);
        KCHECK(c);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stage_ms[1] += now_ms() - t1;
    return SD_OK;
}

int finalize(sd_ctx* c, const float* d_seg, const float* d_emb, int64_t chunks, int64_t n, std::vector<sd_turn>& v)
{
    int rc;
    const double t0 = now_ms();
    const int64_t M = chunks * SD_SPEAKERS;
    const int64_t nf = count_frames_host(chunks);
    WS(c, uint8_t, d_bin, "fin_bin", chunks * SD_FRAMES * 3);
    WS(c, int, d_nact, "fin_nact", M);
    WS(c, int32_t, d_count, "fin_count", nf);
    WS(c, double, d_e64, "fin_e64", M * SD_EMB_DIM);
    WS(c, int, d_hard, "fin_hard", M);
    if ((rc = run_postseg(c, d_seg, chunks, d_bin, nullptr, d_nact))) return rc;
    double* d_count_avg = nullptr;
    if (!c->dump_dir.empty()) { WS(c, double, t_avg, "fin_count_avg", nf); d_count_avg = t_avg; }
    if ((rc = run_count(c, d_bin, chunks, d_count, nf, d_count_avg))) return rc;
    hipLaunchKernelGGL(k_f32_to_f64, GRID1(M * SD_EMB_DIM), 0, c->stream, d_emb, d_e64, M * SD_EMB_DIM);
    KCHECK(c);
    std::vector<int> hard; int K = 1;
    std::vector<double> soft_best;
    if ((rc = run_clustering(c, d_e64, M, SD_EMB_DIM, hard, &K, c->num_clusters, c->min_clusters, c->max_clusters, &soft_best))) return rc;
    HIPCHK(c, hipMemcpyAsync(d_hard, hard.data(), (size_t)M * sizeof(int), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_mark_inactive, GRID1(M), 0, c->stream, d_hard, d_nact, M);
    KCHECK(c);
    HIPCHK(c, hipMemcpyAsync(hard.data(), d_hard, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int Kr = 0;
    for (int h : hard) if (h > Kr) Kr = h;
    Kr += 1;                                                                  // sd.cpp:2803-2812
    if ((rc = run_reconstruct(c, d_seg, d_nact, d_hard, d_count, nf, chunks, n, Kr, v))) return rc;
    if (!c->dump_dir.empty()) {
        rc = write_step_dumps(c, d_seg, d_emb, chunks, n, hard, Kr);
        c->stash.infer_items = 0;            // the per-batch files describe the inference that led to THIS finalize only
        if (rc) return rc;
    }
    // per-turn confidence: mean soft score (2 - cosine distance to the centroid, sd.cpp:2191-2207) of the
    // (chunk, local speaker) items assigned to the turn's cluster whose 5 s chunk [0.5 c, 0.5 c + 5) overlaps the turn
    { KernelStat& ks = c->stats["clusters_K"]; ks.launches++; ks.flops += (double)Kr; ks.bytes += (double)v.size(); }       // bench: K and turns per job
    c->last_conf.assign(v.size(), NAN);
    for (size_t t = 0; t < v.size(); ++t) {
        int64_t c0 = (int64_t)std::floor((v[t].start - 5.0) / 0.5), c1 = (int64_t)std::ceil(v[t].end / 0.5);
        if (c0 < 0) c0 = 0;
        if (c1 > chunks - 1) c1 = chunks - 1;
        double sum = 0.0; int64_t cnt = 0;
        for (int64_t ck = c0; ck <= c1; ++ck) {
            if (!(0.5 * (double)ck < v[t].end && 0.5 * (double)ck + 5.0 > v[t].start)) continue;
            for (int s = 0; s < SD_SPEAKERS; ++s) {
                const size_t i = (size_t)(ck * SD_SPEAKERS + s);
                if (hard[i] == v[t].label && soft_best[i] == soft_best[i]) { sum += soft_best[i]; cnt++; }
            }
        }
        if (cnt > 0) c->last_conf[t] = sum / (double)cnt;
    }
    c->stage_ms[2] += now_ms() - t0;
    return SD_OK;
}

extern "C" int sd_shard_infer_dev(sd_ctx* c, const int16_t* d_pcm_shard, int64_t first_sample, int64_t shard_samples, int64_t n,
                                  int64_t chunk_lo, int64_t chunk_hi, float* d_seg, float* d_emb)
{
    ENTER(c);
    if (!d_pcm_shard || !d_seg || !d_emb || n <= 1 || shard_samples <= 0 || first_sample < 0) SD_FAIL(c, SD_ERR_ARG, "sd_shard_infer_dev: bad argument");
    const int64_t need_lo = chunk_lo * SD_HOP;
    int64_t need_hi = (chunk_hi - 1) * SD_HOP + SD_CHUNK; if (need_hi > n) need_hi = n;
    if (chunk_hi > chunk_lo && (first_sample > need_lo || first_sample + shard_samples < need_hi))
        SD_FAIL(c, SD_ERR_ARG, "shard samples [%lld,%lld) do not cover chunks [%lld,%lld)", (long long)first_sample, (long long)(first_sample + shard_samples), (long long)chunk_lo, (long long)chunk_hi);
    float* w = nullptr;
    int rc0 = pcm_to_wav(c, d_pcm_shard, shard_samples, &w);
    if (rc0) return rc0;
    for (int i = 0; i < 4; ++i) c->stage_ms[i] = 0;
    c->wav_origin = first_sample;                 // kernels index the recording with absolute sample positions
    const int rc = shard_infer(c, w, n, chunk_lo, chunk_hi, d_seg, d_emb);
    c->wav_origin = 0;
    return rc;
}

extern "C" int sd_finalize_dev(sd_ctx* c, const float* d_seg, const float* d_emb, int64_t chunks, int64_t n, sd_turn** turns, int64_t* n_turns)
{
    ENTER(c);
    if (!d_seg || !d_emb || !turns || !n_turns || chunks <= 0) SD_FAIL(c, SD_ERR_ARG, "sd_finalize_dev: bad argument");
    std::vector<sd_turn> v;
    c->stage_ms[2] = 0;                      // per call: a rank that only finalizes never passes through sd_shard_infer_dev
    int rc = finalize(c, d_seg, d_emb, chunks, n, v);
    if (rc) return rc;
    return turns_out(c, v, turns, n_turns);
}

extern "C" int sd_set_planted(sd_ctx* c, const float* d_scores, const float* d_emb, int64_t chunk_lo, int64_t chunks)
{
    if (!c || chunk_lo < 0 || chunks < 0) return SD_ERR_ARG;
    c->planted_scores = chunks > 0 ? d_scores : nullptr;
    c->planted_emb = chunks > 0 ? d_emb : nullptr;
    c->planted_lo = chunk_lo; c->planted_n = (d_scores || d_emb) ? chunks : 0;
    return SD_OK;
}

extern "C" int sd_diarize_dev(sd_ctx* c, const int16_t* d_pcm, int64_t n, sd_turn** turns, int64_t* n_turns)
{
    ENTER(c);
    if (!d_pcm || !turns || !n_turns) SD_FAIL(c, SD_ERR_ARG, "sd_diarize_dev: bad argument");
    const double t0 = now_ms();
    const int64_t chunks = sd_num_chunks(n, nullptr);
    if (chunks <= 0) SD_FAIL(c, SD_ERR_SHORT, "audio of %lld samples yields no chunk", (long long)n);
    for (int i = 0; i < 4; ++i) c->stage_ms[i] = 0;
    float* d_wav = nullptr;
    int rc = pcm_to_wav(c, d_pcm, n, &d_wav);
    if (rc) return rc;
    WS(c, float, d_seg, "dz_seg", chunks * SD_FRAMES * 3);
    WS(c, float, d_emb, "dz_emb", chunks * 3 * SD_EMB_DIM);
    if ((rc = shard_infer(c, d_wav, n, 0, chunks, d_seg, d_emb))) return rc;
    std::vector<sd_turn> v;
    if ((rc = finalize(c, d_seg, d_emb, chunks, n, v))) return rc;
    c->stage_ms[3] = now_ms() - t0;
    if (getenv("SD_TRACE_WS")) {
        fprintf(stderr, "[sdhip] job %.1f ms (segmentation %.1f, embedding %.1f, finalize %.1f); workspace allocations so far in this process: %zu hipMalloc, %.2f GB, %.1f ms (+ %.1f ms hipFree)\n",
                c->stage_ms[3], c->stage_ms[0], c->stage_ms[1], c->stage_ms[2], g_ws_allocs.load(), (double)g_ws_alloc_bytes.load() / 1e9, (double)g_ws_alloc_us.load() * 1e-3, (double)g_ws_free_us.load() * 1e-3);
        for (const auto& kv : c->ws) if (kv.second.cap >= ((size_t)256 << 20)) fprintf(stderr, "[sdhip]   %-16s %8.2f GB\n", kv.first.c_str(), (double)kv.second.cap / 1e9);
    }
    return turns_out(c, v, turns, n_turns);
}

extern "C" int sd_diarize(sd_ctx* c, const int16_t* h_pcm, int64_t n, sd_turn** turns, int64_t* n_turns)
{
    ENTER(c);
    if (!h_pcm || n <= 0) SD_FAIL(c, SD_ERR_ARG, "sd_diarize: bad argument");
    DTMP(c, dp, n * sizeof(int16_t));
    HIPCHK(c, hipMemcpy(dp.p, h_pcm, n * sizeof(int16_t), hipMemcpyHostToDevice));
    return sd_diarize_dev(c, (const int16_t*)dp.p, n, turns, n_turns);
}

// ------------------------------------------------------------------ float-sample entry (8 / 32-bit wavs)
// `wav` holds samples already divided by 32768 exactly as the reference's loop does for every bit depth (sd.cpp:2948-2951)
extern "C" int sd_diarize_f32(sd_ctx* c, const float* h_wav, int64_t n, sd_turn** turns, int64_t* n_turns)
{
    ENTER(c);
    if (!h_wav || n <= 0 || !turns || !n_turns) SD_FAIL(c, SD_ERR_ARG, "sd_diarize_f32: bad argument");
    const double t0 = now_ms();
    const int64_t chunks = sd_num_chunks(n, nullptr);
    if (chunks <= 0) SD_FAIL(c, SD_ERR_SHORT, "audio of %lld samples yields no chunk", (long long)n);
    for (int i = 0; i < 4; ++i) c->stage_ms[i] = 0;
    WS(c, float, d_wav, "wav_f32", n + 512);
    HIPCHK(c, hipMemcpyAsync(d_wav, h_wav, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d_wav + n, 0, 512 * sizeof(float), c->stream));
    c->wav_padded = true;
    WS(c, float, d_seg, "dz_seg", chunks * SD_FRAMES * 3);
    WS(c, float, d_emb, "dz_emb", chunks * 3 * SD_EMB_DIM);
    int rc;
    if ((rc = shard_infer(c, d_wav, n, 0, chunks, d_seg, d_emb))) return rc;
    std::vector<sd_turn> v;
    if ((rc = finalize(c, d_seg, d_emb, chunks, n, v))) return rc;
    c->stage_ms[3] = now_ms() - t0;
    return turns_out(c, v, turns, n_turns);
}

// ------------------------------------------------------------------ wav file entry: reader + rate / channel handling + the path
int resample_dev(sd_ctx* c, const float* d_in, int64_t n, int32_t in_sr, int32_t out_sr, float* d_out, int64_t n_out);   // resample.hip

extern "C" int sd_diarize_wav(sd_ctx* c, const char* path, int flags, sd_turn** turns, int64_t* n_turns)
{
    ENTER(c);
    if (!path || !turns || !n_turns || (flags & ~(SD_WAV_RESAMPLE | SD_WAV_DOWNMIX | SD_WAV_ASSUME_16K))) SD_FAIL(c, SD_ERR_ARG, "sd_diarize_wav: bad argument");
    *turns = nullptr; *n_turns = 0;
    {   // the common case -- 16-bit, 16 kHz, nothing to mix -- stays int16 up to the GPU (k_pcm_to_f32 does the reference's / 32768 there)
        int16_t* pcm = nullptr; int64_t np = 0; int32_t sr16 = 0, ch16 = 0;
        if (sd_read_wav(path, &pcm, &np, &sr16, &ch16) == SD_OK) {
            struct FreePcm { int16_t* p; ~FreePcm() { sd_free_pcm(p); } } g{pcm};
            if ((sr16 == 16000 || (flags & SD_WAV_ASSUME_16K)) && !(ch16 > 1 && (flags & SD_WAV_DOWNMIX))) return sd_diarize(c, pcm, np, turns, n_turns);
        }
    }
    float* wav = nullptr; int64_t n = 0; int32_t sr = 0, ch = 0, bits = 0;
    if (sd_read_wav_f32(path, &wav, &n, &sr, &ch, &bits) != SD_OK) SD_FAIL(c, SD_ERR_ARG, "cannot read PCM wav: %s", path);
    struct Free { float* p; ~Free() { sd_free_wav(p); } } guard{wav};
    if (ch > 1 && (flags & SD_WAV_DOWNMIX)) {
        // the buffer holds all n * ch interleaved samples (the reference keeps the first n of them as "mono", wav.h:95-97)
        for (int64_t i = 0; i < n; ++i) {
            float s = 0.0f;
            for (int q = 0; q < ch; ++q) s += wav[i * ch + q];
            wav[i] = s / (float)ch;
        }
    }
    if (sr == 16000 || (flags & SD_WAV_ASSUME_16K)) return sd_diarize_f32(c, wav, n, turns, n_turns);      // ASSUME_16K: the reference's behaviour (sd.cpp:2940-2942)
    if (!(flags & SD_WAV_RESAMPLE))
        SD_FAIL(c, SD_ERR_ARG, "%s: sample rate %d Hz; the pipeline needs 16000 (README.md:37 of the reference, which would process the file as if it "
                               "were 16 kHz: SD_WAV_ASSUME_16K / --assume-16k does the same) -- pass SD_WAV_RESAMPLE / --resample", path, sr);
    if (n <= 0) SD_FAIL(c, SD_ERR_SHORT, "%s holds no samples", path);
    const double t0 = now_ms();
    const int64_t no = sd_resample_len(n, sr, 16000);
    if (no <= 0) SD_FAIL(c, SD_ERR_SHORT, "%s: %lld samples at %d Hz give no 16 kHz sample", path, (long long)n, sr);
    const int64_t chunks = sd_num_chunks(no, nullptr);
    if (chunks <= 0) SD_FAIL(c, SD_ERR_SHORT, "audio of %lld samples yields no chunk", (long long)no);
    for (int i = 0; i < 4; ++i) c->stage_ms[i] = 0;
    WS(c, float, d_in, "rs_in", n);
    WS(c, float, d_wav, "wav_f32", no + 512);
    HIPCHK(c, hipMemcpyAsync(d_in, wav, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d_wav + no, 0, 512 * sizeof(float), c->stream));
    int rc;
    if ((rc = resample_dev(c, d_in, n, sr, 16000, d_wav, no))) return rc;
    c->wav_padded = true;
    WS(c, float, d_seg, "dz_seg", chunks * SD_FRAMES * 3);
    WS(c, float, d_emb, "dz_emb", chunks * 3 * SD_EMB_DIM);
    if ((rc = shard_infer(c, d_wav, no, 0, chunks, d_seg, d_emb))) return rc;
    std::vector<sd_turn> v;
    if ((rc = finalize(c, d_seg, d_emb, chunks, no, v))) return rc;
    c->stage_ms[3] = now_ms() - t0;
    return turns_out(c, v, turns, n_turns);
}
