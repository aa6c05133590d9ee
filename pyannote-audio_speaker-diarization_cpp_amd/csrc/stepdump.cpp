// stepdump.cpp -- the step-dump harness: the counterpart of the reference's WRITE_DATA switch (debugWrite / debugWrite2d /
// debugWrite3d, sd.cpp:62-234, and their 40 call sites) for the items pipeline/script/verifyEveryStepResult.py:6-17 checks.
// sd_set_dump_dir(ctx, DIR) / `speakerDiarizer --dump-steps DIR` make the next whole-path call write DIR/cpp_<item>.txt in the
// reference's text format, so that the reference's verifier (which reads /tmp/cpp_<item>.txt against /tmp/py_<item>.txt) runs
// unchanged against this build with DIR = /tmp.
//
// Where the numbers come from.  Tensors the HIP path materialises are copied back from the GPU and written as they are:
//   segmentations, binarized_segmentations, count_data (k_count's average in front of np_rint), count, embeddings,
//   filtered_embeddings, norm_embeddings (k_gather_normalize), clusters (fcluster), clusterRes, soft_clusters and dist (k_assign's score
//   table), hard_clusters, to_diarization_activations (k_activations), discrete_diarization (k_topk), masks<n>, wav_lens<n>.
// The reference also dumps temporaries of ITS formulation that the fused kernels never hold -- the index arithmetic of
// binarize_ndarray (same_as, well_defined_idx, samples, on, initial_state, binarize_score, binary_ndarray), slices (trimmed,
// sum_trimmed, cropped_*), the per-chunk scatter buffers of aggregate (clustered_segmentations, scores / masks_in_aggregate,
// aggregated_mask, overlapping_chunk_count, aggregated_output) and the argsort that k_topk replaces with a selection
// (sorted_speakers).  Those are DERIVED here, on the host, from the GPU tensors above by the reference's definitions (each cited), so
// that the verifier finds every file; they are views of GPU results, not a second computation of them.  Two names the verifier lists
// are never written by the C++ either (final_wav_lens, signals: the file names already contain "/tmp/" and ".txt", sd.cpp:2513-2517 --
// SURVEY App. B #14), batch_masks has no writer in the reference; imasks<n> and batch_waveform<n> (15 - 25 MB of text per batch) are
// written only at level 2.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <numeric>
#include <sstream>

namespace {

// ---- the reference's three writers (sd.cpp:87-234), same stream operators, same special cases
struct Out {
    std::ofstream f;
    Out(const std::string& dir, const std::string& name) : f(dir + "/" + name + ".txt") {}
    bool ok() const { return (bool)f; }
};
template <class T> void put_plain(std::ofstream& f, T v) { f << v << ","; }
void put_bool(std::ofstream& f, bool b) { f << (b ? "True" : "False") << ","; }
// float / double with the NaN rule and the writeDecimalforZero rule of debugWrite2d<float> / debugWrite3d<float|double>
template <class T> void put_real(std::ofstream& f, T v, bool decimal_for_zero)
{
    if (std::isnan(v)) { f << "nan,"; return; }
    if (decimal_for_zero) {
        std::ostringstream oss;
        oss << std::setprecision(6) << v;
        std::string r = oss.str();
        if (r == "1") r = "1.0";
        if (r == "0") r = "0.0";
        f << r << ",";
    } else f << v << ",";
}
// debugWrite2d<double> has no NaN branch: it streams the value (sd.cpp:166-169); a NaN is streamed as "nan"
void put_double_2d(std::ofstream& f, double v) { if (std::isnan(v)) f << "nan,"; else f << v << ","; }

template <class T> std::vector<T> d2h(sd_ctx* c, const char* ws, size_t count)
{
    std::vector<T> h(count);
    auto it = c->ws.find(ws);
    if (it == c->ws.end() || !it->second.p || it->second.cap < count * sizeof(T)) { h.clear(); return h; }
    if (hipMemcpy(h.data(), it->second.p, count * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) h.clear();
    return h;
}

}  // namespace

int write_step_dumps(sd_ctx* c, const float* d_seg, const float* d_emb, int64_t chunks, int64_t n_samples, const std::vector<int>& hard_post, int K)
{
    const std::string& dir = c->dump_dir;
    const StepStash& S = c->stash;
    const int F = SD_FRAMES, SP = SD_SPEAKERS, D = SD_EMB_DIM;
    const int64_t M = chunks * SP;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<float> seg((size_t)chunks * F * SP), emb((size_t)M * D);
    HIPCHK(c, hipMemcpy(seg.data(), d_seg, seg.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(emb.data(), d_emb, emb.size() * sizeof(float), hipMemcpyDeviceToHost));
    const int64_t nf = count_frames_host(chunks);
    std::vector<uint8_t> bin = d2h<uint8_t>(c, "fin_bin", (size_t)chunks * F * SP);
    std::vector<int32_t> count = d2h<int32_t>(c, "fin_count", (size_t)nf);
    std::vector<double> count_avg = d2h<double>(c, "fin_count_avg", (size_t)nf);
    if (bin.empty() || count.empty() || count_avg.empty()) SD_FAIL(c, SD_ERR_ARG, "step dump: finalize workspaces are missing");
#define OPEN(var, name) Out var(dir, name); if (!var.ok()) SD_FAIL(c, SD_ERR_ARG, "step dump: cannot write %s/%s.txt", dir.c_str(), name)

    // ---- sd.cpp:2964  debugWrite3d<float>(segmentations)
    { OPEN(o, "cpp_segmentations");
      for (int64_t i = 0; i < chunks; ++i) for (int j = 0; j < F; ++j) { for (int k = 0; k < SP; ++k) put_real(o.f, seg[(size_t)((i * F + j) * SP + k)], false); o.f << "\n"; } }
    // ---- binarize_ndarray's temporaries (sd.cpp:1547-1637), rows = (chunk, speaker), cols = frames.  DERIVED from scores / k_binarize_masks' result
    {
        const int64_t R = chunks * SP;
        const double onset = 0.4442333667381752;
        OPEN(o_sc, "cpp_binarize_score"); OPEN(o_same, "cpp_same_as"); OPEN(o_on, "cpp_on"); OPEN(o_wd, "cpp_well_defined_idx");
        OPEN(o_init, "cpp_initial_state"); OPEN(o_smp, "cpp_samples"); OPEN(o_bn, "cpp_binary_ndarray");
        // well_defined_idx rows are padded with -1 to the longest row (Helper::wellDefinedIndex, sd.cpp:623-651)
        size_t max_idx = 0;
        std::vector<std::vector<int>> wd((size_t)R);
        for (int64_t r = 0; r < R; ++r) {
            const int64_t ck = r / SP; const int k = (int)(r % SP);
            for (int j = 0; j < F; ++j) {
                const double s = (double)seg[(size_t)((ck * F + j) * SP + k)];
                if (!(std::fabs(s - onset) < std::numeric_limits<double>::epsilon())) wd[(size_t)r].push_back(j);     // off_or_on, sd.cpp:1573-1580
            }
            max_idx = std::max(max_idx, wd[(size_t)r].size());
        }
        for (int64_t r = 0; r < R; ++r) {
            const int64_t ck = r / SP; const int k = (int)(r % SP);
            int run = 0;
            size_t w = 0;
            for (int j = 0; j < F; ++j) {
                const double s = (double)seg[(size_t)((ck * F + j) * SP + k)];
                put_double_2d(o_sc.f, s);
                const bool well = w < wd[(size_t)r].size() && wd[(size_t)r][w] == j;
                if (well) { ++run; ++w; }
                put_plain(o_same.f, run);                                       // cumulativeSum(off_or_on), sd.cpp:654-671
                put_bool(o_on.f, s > onset);                                    // sd.cpp:1560-1567
                put_bool(o_init.f, false);
                put_plain(o_smp.f, (int)r);
                put_bool(o_bn.f, bin[(size_t)((ck * F + j) * SP + k)] != 0);    // the GPU's decision (= numpy_where of the above)
            }
            for (size_t q = 0; q < max_idx; ++q) put_plain(o_wd.f, q < wd[(size_t)r].size() ? wd[(size_t)r][q] : -1);
            o_sc.f << "\n"; o_same.f << "\n"; o_on.f << "\n"; o_init.f << "\n"; o_smp.f << "\n"; o_bn.f << "\n"; o_wd.f << "\n";
        }
    }
    // ---- sd.cpp:3031-3032  clean (derived: Helper::cleanSegmentations 710-743 on the GPU's binarisation) and binarized
    { OPEN(o_c, "cpp_clean_segmentations"); OPEN(o_b, "cpp_binarized_segmentations");
      for (int64_t i = 0; i < chunks; ++i) for (int j = 0; j < F; ++j) {
          const uint8_t* b = &bin[(size_t)((i * F + j) * SP)];
          const bool keep = (b[0] + b[1] + b[2]) < 2;
          for (int k = 0; k < SP; ++k) { put_real(o_c.f, keep ? (double)b[k] : 0.0, true); put_real(o_b.f, (double)b[k], false); }
          o_c.f << "\n"; o_b.f << "\n"; } }
    // ---- speaker_count: trimmed, sum_trimmed (slices, sd.cpp:1742-1782, 1699-1714), count_data (k_count's average), count
    {
        const int nl = (int)std::floor((double)F * 0.1), Ft = F - 2 * nl;
        OPEN(o_t, "cpp_trimmed"); OPEN(o_s, "cpp_sum_trimmed"); OPEN(o_cd, "cpp_count_data"); OPEN(o_cnt, "cpp_count");
        for (int64_t i = 0; i < chunks; ++i) for (int j = 0; j < Ft; ++j) {
            const uint8_t* b = &bin[(size_t)((i * F + j + nl) * SP)];
            for (int k = 0; k < SP; ++k) put_real(o_t.f, (double)b[k], false);
            o_t.f << "\n";
            put_real(o_s.f, (double)(b[0] + b[1] + b[2]), false); o_s.f << "\n";
        }
        for (int64_t i = 0; i < nf; ++i) { put_double_2d(o_cd.f, count_avg[(size_t)i]); o_cd.f << "\n"; }
        for (int64_t i = 0; i < nf; ++i) put_plain(o_cnt.f, (int)count[(size_t)i]);             // debugWrite<int>: one line, no newline
    }
    // ---- sd.cpp:3149  embeddings [c][3][192] as double
    { OPEN(o, "cpp_embeddings");
      for (int64_t i = 0; i < M; ++i) { for (int q = 0; q < D; ++q) put_real(o.f, (double)emb[(size_t)(i * D + q)], false); o.f << "\n"; } }
    // ---- Cluster (sd.cpp:2074, 2330-2331, 2096, 2186, 2206, 3163); nothing of it exists when fewer than two rows are live (sd.cpp:2082-2090)
    std::vector<int> hard_pre = S.clustered ? S.hard_pre : std::vector<int>((size_t)M, 0);
    if (S.clustered) {
        const int64_t N = S.N; const int Kc = S.K;
        OPEN(o_f, "cpp_filtered_embeddings"); OPEN(o_n, "cpp_norm_embeddings"); OPEN(o_cl, "cpp_clusters"); OPEN(o_cr, "cpp_clusterRes");
        OPEN(o_d, "cpp_dist"); OPEN(o_sf, "cpp_soft_clusters");
        for (int64_t i = 0; i < N; ++i) {
            for (int q = 0; q < D; ++q) { put_double_2d(o_f.f, S.X[(size_t)(i * D + q)]); put_double_2d(o_n.f, S.Xn[(size_t)(i * D + q)]); }
            o_f.f << "\n"; o_n.f << "\n";
            put_plain(o_cl.f, S.clusters[(size_t)i]); put_plain(o_cr.f, S.cluster_res[(size_t)i]);
        }
        for (int64_t i = 0; i < M; ++i) {
            for (int k = 0; k < Kc; ++k) {
                const double sft = S.soft[(size_t)(i * Kc + k)];                 // k_assign: 2 - cosine distance (sd.cpp:2191-2203)
                put_double_2d(o_d.f, 2.0 - sft);
                put_real(o_sf.f, sft, true);
            }
            o_d.f << "\n"; o_sf.f << "\n";
        }
    }
    { OPEN(o, "cpp_hard_clusters");
      for (int64_t i = 0; i < chunks; ++i) { for (int k = 0; k < SP; ++k) put_plain(o.f, hard_pre[(size_t)(i * SP + k)]); o.f << "\n"; } }
    // ---- reconstruct / to_diarization (sd.cpp:2841, 1271-1275 at the second call site, 2654, 2716-2717, 2732, 3205)
    if (K > 0 && S.rows > 0) {
        std::vector<double> act = d2h<double>(c, "rc_act", (size_t)(S.nact * K));
        std::vector<uint8_t> binary = d2h<uint8_t>(c, "rc_binary", (size_t)(S.rows * K));
        if (act.empty() || binary.empty()) SD_FAIL(c, SD_ERR_ARG, "step dump: reconstruction workspaces are missing");
        OPEN(o_cs, "cpp_clustered_segmentations"); OPEN(o_ms, "cpp_masks_in_aggregate"); OPEN(o_ss, "cpp_scores_in_aggregate");
        std::vector<double> amask((size_t)(S.nact * K), 0.0), occ((size_t)(S.nact * K), 0.0);
        double start = 0.0;
        for (int64_t i = 0; i < chunks; ++i) {
            std::vector<char> has((size_t)K, 0);
            for (int s = 0; s < SP; ++s) { const int h = hard_post[(size_t)(i * SP + s)]; if (h >= 0 && h < K) has[(size_t)h] = 1; }
            const int64_t sf = closest_frame_host(0.0, 0.016875, 0.016875, start);                  // sd.cpp:1250
            start += 0.5;
            for (int j = 0; j < F; ++j) {
                for (int k = 0; k < K; ++k) {
                    double v = NAN;                                                                 // max over the local speakers mapped to k (sd.cpp:2767-2786, 2815-2838)
                    if (has[(size_t)k]) {
                        float mx = -INFINITY; bool any = false;
                        for (int s = 0; s < SP; ++s) if (hard_post[(size_t)(i * SP + s)] == k) { const float x = seg[(size_t)((i * F + j) * SP + s)]; if (!any || x > mx) mx = x; any = true; }
                        v = (double)mx;
                    }
                    put_real(o_cs.f, v, false);
                    const bool nn = !std::isnan(v);
                    put_real(o_ms.f, nn ? 1.0 : 0.0, false);                                        // sd.cpp:1183-1200
                    put_real(o_ss.f, nn ? v : 0.0, false);
                    if (nn && sf + j < S.nact) { occ[(size_t)((sf + j) * K + k)] += 1.0; amask[(size_t)((sf + j) * K + k)] = 1.0; }
                }
                o_cs.f << "\n"; o_ms.f << "\n"; o_ss.f << "\n";
            }
        }
        OPEN(o_ao, "cpp_aggregated_output"); OPEN(o_am, "cpp_aggregated_mask"); OPEN(o_oc, "cpp_overlapping_chunk_count"); OPEN(o_ta, "cpp_to_diarization_activations");
        for (int64_t t = 0; t < S.nact; ++t) {
            for (int k = 0; k < K; ++k) {
                const size_t q = (size_t)(t * K + k);
                put_double_2d(o_ao.f, act[q]);                   // written in front of the division / missing pass; skip_average and missing = 0 leave the sums
                put_double_2d(o_am.f, amask[q]); put_double_2d(o_oc.f, occ[q]); put_double_2d(o_ta.f, act[q]);
            }
            o_ao.f << "\n"; o_am.f << "\n"; o_oc.f << "\n"; o_ta.f << "\n";
        }
        OPEN(o_ca, "cpp_cropped_activations"); OPEN(o_cc, "cpp_cropped_count"); OPEN(o_so, "cpp_sorted_speakers"); OPEN(o_dd, "cpp_discrete_diarization");
        std::vector<int> idx((size_t)K);
        for (int64_t r = 0; r < S.rows; ++r) {
            const double* a = &act[(size_t)((S.ar0 + r) * K)];
            for (int k = 0; k < K; ++k) put_double_2d(o_ca.f, a[k]);
            o_ca.f << "\n";
            std::iota(idx.begin(), idx.end(), 0);                                                   // Helper::argsort of the negated row, stable (sd.cpp:2722-2727, 274-290)
            std::stable_sort(idx.begin(), idx.end(), [a](int i1, int i2) { return -a[i1] < -a[i2]; });
            for (int k = 0; k < K; ++k) put_plain(o_so.f, idx[(size_t)k]);
            o_so.f << "\n";
            for (int k = 0; k < K; ++k) put_double_2d(o_dd.f, (double)binary[(size_t)(r * K + k)]);
            o_dd.f << "\n";
        }
        for (int64_t r = 0; r < S.crow_all; ++r) {                                                  // converted_count clamps to the number of clusters (sd.cpp:2675-2682)
            const int v = (int)count[(size_t)(S.cr0 + r)];
            put_plain(o_cc.f, v > K ? K : v); o_cc.f << "\n";
        }
    }
    // ---- getEmbedding's per-batch files (sd.cpp:2453-2454, 2491, 2444): only when the masks / counts of the WHOLE recording are resident
    {
        std::vector<float> masks = d2h<float>(c, "sh_masks", (size_t)(M * F));
        std::vector<int> cnts = d2h<int>(c, "fe_counts", (size_t)M);
        if (c->stash.infer_items == M && !masks.empty() && !cnts.empty()) {
            std::vector<float> wav;
            if (c->dump_level >= 2) { wav = d2h<float>(c, "wav_f32", (size_t)n_samples); }
            // the reference's file counter only advances for batches that reach the model (`number++` sits behind the early return,
            // sd.cpp:2479-2519): a batch below min_num_samples leaves masks<n> / imasks<n> behind and the next batch overwrites them
            for (int64_t b0 = 0, nb = 0; b0 < M; b0 += SD_EMB_BATCH) {
                const int64_t b1 = std::min<int64_t>(M, b0 + SD_EMB_BATCH);
                const std::string sfx = std::to_string(nb);
                OPEN(o_m, ("cpp_masks" + sfx).c_str()); OPEN(o_w, ("cpp_wav_lens" + sfx).c_str());
                float mx = 0.0f;
                for (int64_t i = b0; i < b1; ++i) {
                    for (int j = 0; j < F; ++j) put_real(o_m.f, masks[(size_t)(i * F + j)], true);
                    o_m.f << "\n";
                    mx = std::max(mx, (float)cnts[(size_t)i]);
                }
                // the reference writes wav_lens (the selected sample COUNTS, sd.cpp:2466-2491) only for batches it sends to the model
                if (mx >= 640.0f) for (int64_t i = b0; i < b1; ++i) put_plain(o_w.f, (float)cnts[(size_t)i]);
                if (c->dump_level >= 2 && !wav.empty()) {
                    OPEN(o_i, ("cpp_imasks" + sfx).c_str()); OPEN(o_bw, ("cpp_batch_waveform" + sfx).c_str());
                    for (int64_t i = b0; i < b1; ++i) {
                        const int64_t s0 = (i / SP) * SD_HOP;
                        for (int64_t j = 0; j < SD_CHUNK; ++j) {
                            put_bool(o_i.f, masks[(size_t)(i * F + (j * F) / SD_CHUNK)] > 0.5f);    // Helper::interpolate, sd.cpp:746-767
                            put_real(o_bw.f, s0 + j < n_samples ? wav[(size_t)(s0 + j)] : 0.0f, false);   // SegmentModel::crop zero-pads, sd.cpp:1641-1662
                        }
                        o_i.f << "\n"; o_bw.f << "\n";
                    }
                }
                if (mx >= 640.0f) ++nb;
            }
        }
    }
#undef OPEN
    return SD_OK;
}

extern "C" int sd_set_dump_dir(sd_ctx* c, const char* dir, int level)
{
    if (!c) return SD_ERR_ARG;
    c->err.clear();
    if (level < 0 || level > 2) SD_FAIL(c, SD_ERR_ARG, "sd_set_dump_dir: level must be 0 (off), 1 or 2 (+ imasks / batch_waveform)");
    c->dump_dir = (dir && level > 0) ? dir : "";
    c->dump_level = c->dump_dir.empty() ? 0 : level;
    return SD_OK;
}
