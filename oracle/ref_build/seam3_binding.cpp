// seam3_binding.cpp -- the reference-side binding of INTEGRATION.md "Seam 3, literal form", kept compilable: the body a maintainer
// puts in place of EmbeddingModel1::infer (pipeline/src/speakerDiarizer.cpp:1977-2040; its torch::stft and the _infer /
// Ort::Session::Run it ends in, sd.cpp:1889-1970, both go) so that the reference's own getEmbedding() (sd.cpp:2436-2561:
// Helper::interpolate, padSequence, wav_lens, too_short / NaN rule) stays as it is.  Same parameter and return types as the reference's
// method ("input: batch size x waveform, wave lens; output: embedding").
// tests/test_abi.py compiles this file against include/sdhip.h; tests/test_next_rows.py builds it with -DSEAM_TEST_SHIM into a
// shared object and runs it on the GPU against sd_embed.  TEST INFRASTRUCTURE (never linked into libsdhip.so).
#include <vector>
#include <stdexcept>
#include "sdhip.h"

extern sd_ctx* g_ctx;          // created once by the host program (sd_create(segment_model, embedding_model, device))

// body of: std::vector<std::vector<float>> EmbeddingModel1::infer(const std::vector<std::vector<float>>& data, const std::vector<float>& lens)
std::vector<std::vector<float>> EmbeddingModel1_infer_on_sdhip(const std::vector<std::vector<float>>& data, const std::vector<float>& lens)
{
    const int64_t B = (int64_t)data.size();
    const size_t len = B ? data[0].size() : 0;                              // sd.cpp:1985: 80000 (getEmbedding pads every signal to num_samples)
    if (len != (size_t)SD_CHUNK || lens.size() != (size_t)B) throw std::runtime_error("EmbeddingModel1::infer: signals of 80000 samples and one length each");
    std::vector<float> x((size_t)B * len);
    for (int64_t i = 0; i < B; ++i) std::copy(data[(size_t)i].begin(), data[(size_t)i].end(), x.begin() + (size_t)i * len);
    std::vector<float> emb((size_t)B * SD_EMB_DIM);
    if (sd_embed_signals(g_ctx, x.data(), lens.data(), B, emb.data()) != SD_OK) throw std::runtime_error(sd_last_error(g_ctx));
    std::vector<std::vector<float>> res((size_t)B, std::vector<float>(SD_EMB_DIM));      // sd.cpp:1959-1967
    for (int64_t i = 0; i < B; ++i) res[(size_t)i].assign(emb.begin() + (size_t)i * SD_EMB_DIM, emb.begin() + (size_t)(i + 1) * SD_EMB_DIM);
    return res;
}

#ifdef SEAM_TEST_SHIM
sd_ctx* g_ctx = nullptr;
extern "C" int seam3_run(sd_ctx* ctx, const float* signals, const float* lens, long B, float* out /*[B][192]*/)
{
    g_ctx = ctx;
    std::vector<std::vector<float>> d((size_t)B, std::vector<float>((size_t)SD_CHUNK));
    for (long i = 0; i < B; ++i) d[(size_t)i].assign(signals + i * (long)SD_CHUNK, signals + (i + 1) * (long)SD_CHUNK);
    try {
        const auto res = EmbeddingModel1_infer_on_sdhip(d, std::vector<float>(lens, lens + B));
        for (size_t i = 0; i < res.size(); ++i) for (int q = 0; q < SD_EMB_DIM; ++q) out[i * SD_EMB_DIM + (size_t)q] = res[i][(size_t)q];
    } catch (const std::exception&) { return 1; }
    return 0;
}
#endif
