// seam2_binding.cpp -- the reference-side binding of INTEGRATION.md "Seam 2, literal form", kept compilable: the body a maintainer
// puts in place of SegmentModel::infer (pipeline/src/speakerDiarizer.cpp:1352-1404) so that the reference's own slide()
// (sd.cpp:1407-1504) keeps framing the waveform and only the Ort::Session::Run call is replaced by libsdhip.  Same parameter and
// return types as the reference's method ("input: batch size x samples count, output: batch size x 293 x 3").
// tests/test_abi.py compiles this file against include/sdhip.h; tests/test_next_rows.py builds it with -DSEAM_TEST_SHIM into a
// shared object and runs it on the GPU against sd_segment.  TEST INFRASTRUCTURE (never linked into libsdhip.so).
#include <vector>
#include <stdexcept>
#include "sdhip.h"

extern sd_ctx* g_ctx;          // created once by the host program (sd_create(segment_model, embedding_model, device))

// body of: std::vector<std::vector<std::vector<float>>> SegmentModel::infer(const std::vector<std::vector<float>>& waveform)
std::vector<std::vector<std::vector<float>>> SegmentModel_infer_on_sdhip(const std::vector<std::vector<float>>& waveform)
{
    const int64_t rows = (int64_t)waveform.size();
    const int64_t T = rows ? (int64_t)waveform[0].size() : 0;
    std::vector<float> audio((size_t)(rows * T));                         // sd.cpp:1356-1364 (the reference copies into its batch of 32)
    for (int64_t i = 0; i < rows; ++i) std::copy(waveform[(size_t)i].begin(), waveform[(size_t)i].end(), audio.begin() + i * T);
    std::vector<float> out((size_t)rows * SD_FRAMES * SD_SPEAKERS);
    int32_t frames = 0;
    if (sd_segment_chunks(g_ctx, audio.data(), rows, T, out.data(), &frames) != SD_OK) throw std::runtime_error(sd_last_error(g_ctx));
    // sd.cpp:1390-1401: res[len1 = waveform.size()][len2][len3]; len2 = the frames the model yields for T samples
    std::vector<std::vector<std::vector<float>>> res((size_t)rows, std::vector<std::vector<float>>((size_t)frames, std::vector<float>(SD_SPEAKERS)));
    for (int64_t i = 0; i < rows; ++i)
        for (int j = 0; j < frames; ++j)
            for (int k = 0; k < SD_SPEAKERS; ++k) res[(size_t)i][(size_t)j][(size_t)k] = out[((size_t)i * SD_FRAMES + j) * SD_SPEAKERS + k];
    return res;
}

#ifdef SEAM_TEST_SHIM
sd_ctx* g_ctx = nullptr;
extern "C" int seam2_run(sd_ctx* ctx, const float* chunks, long rows, long T, float* out /*[rows][293][3], zero beyond frames*/, int* frames)
{
    g_ctx = ctx;
    std::vector<std::vector<float>> w((size_t)rows, std::vector<float>((size_t)T));
    for (long i = 0; i < rows; ++i) w[(size_t)i].assign(chunks + i * T, chunks + (i + 1) * T);
    try {
        const auto res = SegmentModel_infer_on_sdhip(w);
        *frames = res.empty() ? 0 : (int)res[0].size();
        for (size_t i = 0; i < res.size(); ++i)
            for (size_t j = 0; j < res[i].size(); ++j)
                for (int k = 0; k < SD_SPEAKERS; ++k) out[(i * SD_FRAMES + j) * SD_SPEAKERS + k] = res[i][j][(size_t)k];
    } catch (const std::exception&) { return 1; }
    return 0;
}
#endif
