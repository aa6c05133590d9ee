// main.cpp -- the `speakerDiarizer` command line, same surface as the reference's main()
// (sd.cpp:3415-3442):   speakerDiarizer <segment model> <embedding model> <16 kHz mono 16-bit wav>
// prints the per-stage timings and, between two 52-dash rules, one line per turn:
//   [start -- end] --> Speaker_N
// Model files are .sdw weight packs (tools/make_weights.py); the whole path runs on GPU 0
// through the C ABI of libsdhip.so.
#include <cstdio>
#include <cstdlib>
#include <string>
#include "sdhip.h"

int main(int argc, char* argv[])
{
    if (argc < 4) {
        printf("program [segment model file] [embeding model file] [wave file]\n");   // sd.cpp:3423
        return 0;
    }
    float* wav = nullptr; int64_t n = 0; int32_t sr = 0, ch = 0, bits = 0;
    if (sd_read_wav_f32(argv[3], &wav, &n, &sr, &ch, &bits) != SD_OK) {       // 8 / 16 / 32-bit PCM like wav.h:99-122
        fprintf(stderr, "cannot read PCM wav: %s\n", argv[3]);
        return 1;
    }
    sd_ctx* ctx = sd_create(argv[1], argv[2], 0);
    if (!ctx) {
        fprintf(stderr, "sd_create failed: %s\n", sd_create_error());
        return 1;
    }
    sd_turn* turns = nullptr; int64_t nt = 0;
    int rc = sd_diarize_f32(ctx, wav, n, &turns, &nt);
    if (rc != SD_OK) {
        fprintf(stderr, "diarization failed (%d): %s\n", rc, sd_last_error(ctx));
        return 1;
    }
    double ms[4];
    sd_stage_ms(ctx, ms);
    printf("-----------\nSegmenations time: %lldms\n", (long long)ms[0]);      // labels of sd.cpp:3028, 3110, 3231
    printf("-----------\nEmbedding time: %lldms\n", (long long)ms[1]);
    printf("-----------\nClustering time: %lldms\n", (long long)ms[2]);
    printf("\n----Summary----\n-----------\nTime cost: %lldms\n", (long long)ms[3]);
    printf("----------------------------------------------------\n");
    char line[160];
    for (int64_t i = 0; i < nt; ++i) { sd_format_turn(&turns[i], line, sizeof(line)); printf("%s\n", line); }
    printf("----------------------------------------------------\n");
    if (argc >= 6 && std::string(argv[4]) == "--rttm") sd_write_rttm(argv[5], argv[3], turns, nt);   // optional extra: RTTM file
    sd_free_turns(turns);
    sd_free_wav(wav);
    sd_destroy(ctx);
    return 0;
}
