// main.cpp -- the `speakerDiarizer` command line, same surface as the reference's main()
// (sd.cpp:3415-3442):   speakerDiarizer <segment model> <embedding model> <16 kHz mono wav>
// prints the per-stage timings and, between two 52-dash rules, one line per turn:
//   [start -- end] --> Speaker_N
// Model files are the reference's ONNX files or .sdw weight packs (tools/make_weights.py); the
// whole path runs on the GPU through the C ABI of libsdhip.so.
//
// Extras after the three positional arguments (the reference has none):
//   --gpus N        shard the recording over N GPUs of this node: the launcher fork()s one
//                   process per GPU BEFORE anything touches HIP, rank 0 mints the RCCL rendezvous id and
//                   passes it to the others through pipes created before the fork; turns are printed by rank 0
//   --rttm FILE     also write the turns as RTTM
//   --precision P   f32 (default: f32 MFMA, the reference's ORT precision) | f16 (fp16 ECAPA layers, BASELINE configs[4]) | x3 (f32 tensors, both
//                   MFMA operands split into hi + lo fp16 halves in the ECAPA conv layers and PyanNet's LSTM: f32-grade results from the fp16 matrix
//                   pipe) = sd_set_option "ecapa_precision"; PyanNet's LSTM follows ("seg_precision" is left at auto for f16, set to 3 for x3: the same thing)
//   --resample      a wav whose sample rate is not 16 000 Hz is resampled on the GPU first (sd_resample; the dormant Resampler of the reference,
//                   frontend/resampler.cc:19-36).  WITHOUT it such a file is refused: the reference reads the rate and ignores it (sd.cpp:2940-2942),
//                   i.e. silently diarizes at the wrong speed
//   --assume-16k    the reference's own behaviour on such a file: the samples are processed as if they were 16 kHz whatever the header says
//                   (parity runs on off-rate files; SD_WAV_ASSUME_16K)
//   --downmix       average the channels of a multi-channel wav first (default: the reference's interleaved read, wav.h:95-97)
//   --dump-steps DIR [--dump-level 2]   the reference's WRITE_DATA switch: DIR/cpp_<item>.txt for the items of script/verifyEveryStepResult.py
//                   (sd_set_dump_dir; DIR = /tmp is what that script reads); single-GPU runs
//   --relabel       stdout / RTTM labels renumbered the way pyannote.audio names its output (the clusters that occur,
//                   sorted by their string, become 0, 1, ... = SPEAKER_00, SPEAKER_01, ...); default = raw cluster ids (sd.cpp:3439)
#include <cstdio>
#include <ctime>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <csignal>
#include <sys/wait.h>
#include <unistd.h>
#include "sdhip.h"

struct Args { const char* seg = nullptr; const char* emb = nullptr; const char* wav = nullptr; const char* rttm = nullptr; int gpus = 1; bool relabel = false; int precision = 0; int wav_flags = 0; const char* dump_dir = nullptr; int dump_level = 1; };

static void print_block(sd_ctx* ctx, sd_turn* turns, int64_t nt, const Args& a)
{
    double ms[4];
    sd_stage_ms(ctx, ms);
    printf("-----------\nSegmenations time: %lldms\n", (long long)ms[0]);      // labels of sd.cpp:3028, 3110, 3231
    printf("-----------\nEmbedding time: %lldms\n", (long long)ms[1]);
    printf("-----------\nClustering time: %lldms\n", (long long)ms[2]);
    printf("\n----Summary----\n-----------\nTime cost: %lldms\n", (long long)ms[3]);
    printf("----------------------------------------------------\n");
    if (a.relabel) sd_relabel_turns(turns, nt);
    char line[160];
    for (int64_t i = 0; i < nt; ++i) { sd_format_turn(&turns[i], line, sizeof(line)); printf("%s\n", line); }
    printf("----------------------------------------------------\n");
    if (a.rttm) sd_write_rttm(a.rttm, a.wav, turns, nt);
    fflush(stdout);
}

static double wall_ms()
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static const bool g_trace = getenv("SD_TRACE_CREATE") != nullptr;     // start-up breakdown on stderr (tools/cold_start.py)
static double g_t_main = 0;
#define TRACE(what) do { if (g_trace) fprintf(stderr, "cli: +%.1f ms %s\n", wall_ms() - g_t_main, what); } while (0)

static int run_single(const Args& a)
{
    TRACE("run_single");
    sd_ctx* ctx = sd_create(a.seg, a.emb, 0);
    if (!ctx) { fprintf(stderr, "sd_create failed: %s\n", sd_create_error()); return 1; }
    TRACE("sd_create done");
    if (a.precision && sd_set_option(ctx, "ecapa_precision", a.precision) != SD_OK) { fprintf(stderr, "%s\n", sd_last_error(ctx)); return 1; }
    if (a.precision == 3 && sd_set_option(ctx, "seg_precision", 3) != SD_OK) { fprintf(stderr, "%s\n", sd_last_error(ctx)); return 1; }
    if (a.dump_dir && sd_set_dump_dir(ctx, a.dump_dir, a.dump_level) != SD_OK) { fprintf(stderr, "%s\n", sd_last_error(ctx)); return 1; }
    sd_turn* turns = nullptr; int64_t nt = 0;
    const int rc = sd_diarize_wav(ctx, a.wav, a.wav_flags, &turns, &nt);      // 8 / 16 / 32-bit PCM like wav.h:99-122; rate and channels checked
    if (rc != SD_OK) { fprintf(stderr, "diarization failed (%d): %s\n", rc, sd_last_error(ctx)); return 1; }
    TRACE("sd_diarize_wav done");
    print_block(ctx, turns, nt, a);
    sd_free_turns(turns);
    sd_destroy(ctx);
    TRACE("sd_destroy done");
    return 0;
}

static bool read_all(int fd, void* buf, size_t n)
{
    size_t got = 0;
    while (got < n) { const ssize_t r = read(fd, (char*)buf + got, n - got); if (r <= 0) return false; got += (size_t)r; }
    return true;
}

// one rank of the sharded job.  id_rd = pipe this rank reads the rendezvous id from (ranks > 0), id_wr = the pipes rank 0 writes it to;
// ready_wr = pipe on which this rank (> 0) tells rank 0 that its wav and its context are in place, ready_rd = rank 0's ends.
// Rank 0 mints the id only after EVERY rank has reported ready: a rank that cannot start (no such GPU, unreadable wav, bad model)
// ends the job before anyone enters ncclCommInitRank, where a missing peer would mean waiting forever.
static int run_rank(const Args& a, int rank, int world, int id_rd, const std::vector<int>& id_wr, int ready_wr, const std::vector<int>& ready_rd)
{
    int16_t* pcm = nullptr; int64_t n = 0; int32_t sr = 0, ch = 0;
    if (sd_read_wav(a.wav, &pcm, &n, &sr, &ch) != SD_OK) {
        fprintf(stderr, "rank %d: cannot read 16-bit PCM wav: %s (--gpus needs 16-bit samples)\n", rank, a.wav);
        return 1;
    }
    if (sr != 16000 || a.wav_flags) {
        fprintf(stderr, "rank %d: %s: %d Hz%s; --gpus shards a 16 kHz mono 16-bit recording as it is (resample / downmix it first: --resample and --downmix "
                        "are single-GPU options)\n", rank, a.wav, sr, a.wav_flags ? ", --resample / --downmix given" : "");
        return 1;
    }
    sd_ctx* ctx = sd_create(a.seg, a.emb, rank);
    if (!ctx) { fprintf(stderr, "rank %d: sd_create failed: %s\n", rank, sd_create_error()); return 1; }
    if (a.precision && sd_set_option(ctx, "ecapa_precision", a.precision) != SD_OK) { fprintf(stderr, "rank %d: %s\n", rank, sd_last_error(ctx)); return 1; }
    if (a.precision == 3 && sd_set_option(ctx, "seg_precision", 3) != SD_OK) { fprintf(stderr, "rank %d: %s\n", rank, sd_last_error(ctx)); return 1; }
    unsigned char id[SD_COMM_ID_BYTES];
    if (rank == 0) {
        for (size_t q = 0; q < ready_rd.size(); ++q) {
            char ok = 0;
            if (!read_all(ready_rd[q], &ok, 1) || ok != 1) { fprintf(stderr, "rank 0: rank %zu did not come up; no job\n", q + 1); return 1; }
        }
        if (sd_comm_unique_id(id) != SD_OK) { fprintf(stderr, "rank 0: sd_comm_unique_id failed\n"); return 1; }
        for (int fd : id_wr) if (write(fd, id, sizeof(id)) != (ssize_t)sizeof(id)) { fprintf(stderr, "rank 0: cannot hand the rendezvous id over\n"); return 1; }
    } else {
        const char ok = 1;
        if (write(ready_wr, &ok, 1) != 1) { fprintf(stderr, "rank %d: rank 0 is gone\n", rank); return 1; }
        if (!read_all(id_rd, id, sizeof(id))) { fprintf(stderr, "rank %d: no rendezvous id from rank 0\n", rank); return 1; }
    }
    if (sd_comm_init(ctx, id, rank, world) != SD_OK) { fprintf(stderr, "rank %d: sd_comm_init failed: %s\n", rank, sd_last_error(ctx)); return 1; }
    std::vector<int64_t> ranges((size_t)world * 2);
    sd_shard_plan(n, world, -1, ranges.data(), nullptr);
    const int64_t lo = ranges[2 * (size_t)rank], hi = ranges[2 * (size_t)rank + 1];
    int64_t s0 = lo * SD_HOP, s1 = hi > lo ? (hi - 1) * SD_HOP + SD_CHUNK : s0;
    if (s1 > n) s1 = n;
    if (s0 > n) s0 = n;
    sd_turn* turns = nullptr; int64_t nt = 0;
    // collective failure: if any rank fails in its part, every rank gets a non-OK return here (comm.cpp) and exits 1
    const int rc = sd_diarize_sharded(ctx, pcm + s0, s0, s1 - s0, n, &turns, &nt);
    if (rc != SD_OK) { fprintf(stderr, "rank %d: diarization failed (%d): %s\n", rank, rc, sd_last_error(ctx)); return 1; }
    if (rank == 0) print_block(ctx, turns, nt, a);
    sd_free_turns(turns);
    sd_free_pcm(pcm);
    sd_destroy(ctx);
    return 0;
}

int main(int argc, char* argv[])
{
    g_t_main = wall_ms();
    Args a;
    std::vector<const char*> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string s(argv[i]);
        if (s == "--gpus" && i + 1 < argc) a.gpus = atoi(argv[++i]);
        else if (s == "--rttm" && i + 1 < argc) a.rttm = argv[++i];
        else if (s == "--relabel") a.relabel = true;
        else if (s == "--dump-steps" && i + 1 < argc) a.dump_dir = argv[++i];
        else if (s == "--dump-level" && i + 1 < argc) a.dump_level = atoi(argv[++i]);
        else if (s == "--resample") a.wav_flags |= SD_WAV_RESAMPLE;
        else if (s == "--downmix") a.wav_flags |= SD_WAV_DOWNMIX;
        else if (s == "--assume-16k") a.wav_flags |= SD_WAV_ASSUME_16K;
        else if (s == "--precision" && i + 1 < argc) {
            const std::string v(argv[++i]);
            if (v == "f32") a.precision = 0; else if (v == "f16") a.precision = 1; else if (v == "x3") a.precision = 3;
            else { fprintf(stderr, "--precision takes f32, f16 or x3\n"); return 2; }
        }
        else pos.push_back(argv[i]);
    }
    if (pos.size() < 3) {
        printf("program [segment model file] [embeding model file] [wave file]\n");   // sd.cpp:3423
        return 0;
    }
    a.seg = pos[0]; a.emb = pos[1]; a.wav = pos[2];
    if (a.gpus <= 1) return run_single(a);

    // ---- launcher: nothing below touches HIP in this process.  id pipes carry the rendezvous id from rank 0 to rank r,
    // ready pipes one byte from rank r to rank 0.
    const int world = a.gpus;
    std::vector<int> id_rd((size_t)world, -1), id_wr((size_t)world, -1), rdy_rd((size_t)world, -1), rdy_wr((size_t)world, -1);
    for (int r = 1; r < world; ++r) {
        int fd[2];
        if (pipe(fd) != 0) { perror("pipe"); return 1; }
        id_rd[(size_t)r] = fd[0]; id_wr[(size_t)r] = fd[1];
        if (pipe(fd) != 0) { perror("pipe"); return 1; }
        rdy_rd[(size_t)r] = fd[0]; rdy_wr[(size_t)r] = fd[1];
    }
    fflush(stdout); fflush(stderr);
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); for (pid_t k : kids) kill(k, SIGKILL); return 1; }
        if (pid == 0) {
            signal(SIGPIPE, SIG_IGN);           // a write to a dead rank's pipe is an error return, not a kill
            // keep only this rank's ends: a rank that dies closes its pipes, so nobody blocks on a read forever
            std::vector<int> my_id_wr, my_rdy_rd;
            for (int q = 1; q < world; ++q) {
                if (r == 0) { close(id_rd[(size_t)q]); close(rdy_wr[(size_t)q]); my_id_wr.push_back(id_wr[(size_t)q]); my_rdy_rd.push_back(rdy_rd[(size_t)q]); }
                else { close(id_wr[(size_t)q]); close(rdy_rd[(size_t)q]); if (q != r) { close(id_rd[(size_t)q]); close(rdy_wr[(size_t)q]); } }
            }
            const int rc = run_rank(a, r, world, r > 0 ? id_rd[(size_t)r] : -1, my_id_wr, r > 0 ? rdy_wr[(size_t)r] : -1, my_rdy_rd);
            fflush(stdout); fflush(stderr);
            _exit(rc);
        }
        kids.push_back(pid);
    }
    for (int r = 1; r < world; ++r) { close(id_rd[(size_t)r]); close(id_wr[(size_t)r]); close(rdy_rd[(size_t)r]); close(rdy_wr[(size_t)r]); }
    // reap in completion order; the first rank that ends badly ends the job: the others (which may be waiting for it inside RCCL)
    // are killed by their exact pids and reaped
    int worst = 0;
    size_t left = kids.size();
    while (left > 0) {
        int st = 0;
        const pid_t pid = waitpid(-1, &st, 0);
        if (pid < 0) { worst = 1; break; }
        size_t which = kids.size();
        for (size_t i = 0; i < kids.size(); ++i) if (kids[i] == pid) which = i;
        if (which == kids.size()) continue;
        kids[which] = -1; --left;
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
            worst = 1;
            fprintf(stderr, "launcher: rank %zu ended with %s %d; stopping the other ranks\n", which, WIFEXITED(st) ? "exit code" : "signal", WIFEXITED(st) ? WEXITSTATUS(st) : WTERMSIG(st));
            // a rank whose own failure is collective (comm.cpp) exits by itself within moments: give the others a short grace period
            // so that their reasons reach stderr, then kill what is left
            for (int spin = 0; spin < 100 && left > 0; ++spin) {
                int st2 = 0;
                const pid_t p2 = waitpid(-1, &st2, WNOHANG);
                if (p2 > 0) { for (size_t i = 0; i < kids.size(); ++i) if (kids[i] == p2) { kids[i] = -1; --left; } }
                else usleep(20000);
            }
            for (size_t i = 0; i < kids.size(); ++i) if (kids[i] > 0) kill(kids[i], SIGKILL);
            for (size_t i = 0; i < kids.size(); ++i) if (kids[i] > 0) { int st3 = 0; (void)waitpid(kids[i], &st3, 0); kids[i] = -1; }
            left = 0;
        }
    }
    return worst;
}
