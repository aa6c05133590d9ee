// comm.cpp -- the multi-GPU form of speakerDiarization() (sd.cpp:2937-3234) behind the C ABI.
// One process per GPU.  Chunks are independent up to clustering: every rank runs segmentation, post-segmentation and
// embeddings for a contiguous chunk range that starts on a multiple of 32 chunks (= 96 items = 3 of the reference's
// embedding batches, sd.cpp:3083), so each rank forms exactly the reference's batches.  The one exchange step is an
// RCCL all-gather (over xGMI) of the segmentation scores and the embeddings, issued on the library's own stream right
// behind the last embedding kernel; rank 0 then runs counting, clustering and reconstruction (centroid linkage is a
// single dependent chain and does not shard).  The other ranks return as soon as their part of the all-gather is
// queued, so consecutive jobs pipeline: rank 0's finalize of job k overlaps the others' inference of job k+1.
// Bootstrap: rank 0 calls sd_comm_unique_id() and the host program carries the 128 bytes to the other ranks by
// whatever channel it has (the CLI uses pipes it created before fork(), bench.py a torch.distributed store).
#include "common.h"
#include <rccl/rccl.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <unistd.h>

#define ENTER(ctx) do { if (!(ctx)) return SD_ERR_ARG; (ctx)->err.clear(); if (hipSetDevice((ctx)->device) != hipSuccess) SD_FAIL(ctx, SD_ERR_HIP, "hipSetDevice failed"); } while (0)
#define NCCLCHK(ctx, expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) SD_FAIL(ctx, SD_ERR_HIP, "%s failed: %s", #expr, ncclGetErrorString(_r)); } while (0)

static_assert(SD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "sdhip.h and rccl.h disagree on the size of the rendezvous id");

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// contiguous chunk ranges, every one starting on a multiple of 32 chunks.  rank0_permille < 0: equal shares; otherwise
// rank 0 (which also finalizes) infers that share of the chunks and the others split the rest evenly.
static int64_t plan_ranges(int64_t n_total, int world, int rank0_permille, std::vector<int64_t>& lo, std::vector<int64_t>& hi)
{
    const int64_t C = sd_num_chunks(n_total, nullptr);
    lo.assign((size_t)world, 0); hi.assign((size_t)world, 0);
    auto up32 = [](int64_t v) { return (v + 31) / 32 * 32; };
    if (rank0_permille < 0 || world == 1) {
        const int64_t per = std::max<int64_t>(32, up32((C + world - 1) / world));
        for (int r = 0; r < world; ++r) { lo[(size_t)r] = std::min(C, (int64_t)r * per); hi[(size_t)r] = std::min(C, (int64_t)(r + 1) * per); }
        return per;
    }
    if (rank0_permille > 1000) rank0_permille = 1000;
    const int64_t c0 = std::min(C, (int64_t)std::nearbyint((double)C * rank0_permille / 1000.0 / 32.0) * 32);
    const int64_t rest = C - c0;
    const int64_t per_r = up32((rest + world - 2) / (world - 1));
    lo[0] = 0; hi[0] = c0;
    for (int r = 1; r < world; ++r) { lo[(size_t)r] = std::min(C, c0 + (int64_t)(r - 1) * per_r); hi[(size_t)r] = std::min(C, c0 + (int64_t)r * per_r); }
    return std::max<int64_t>(std::max(c0, per_r), 32);
}

extern "C" int sd_shard_plan(int64_t n_total, int world, int rank0_permille, int64_t* ranges, int64_t* slot_chunks)
{
    if (world < 1 || !ranges || n_total < 0) return SD_ERR_ARG;
    std::vector<int64_t> lo, hi;
    const int64_t per = plan_ranges(n_total, world, rank0_permille, lo, hi);
    for (int r = 0; r < world; ++r) { ranges[2 * r] = lo[(size_t)r]; ranges[2 * r + 1] = hi[(size_t)r]; }
    if (slot_chunks) *slot_chunks = per;
    return SD_OK;
}

extern "C" int sd_comm_unique_id(void* id)
{
    if (!id) return SD_ERR_ARG;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return SD_ERR_HIP;       // needs a GPU: RCCL initialises HIP
    memcpy(id, &u, SD_COMM_ID_BYTES);
    return SD_OK;
}

static int wait_exchange(sd_ctx* c, const char* what);

extern "C" int sd_comm_destroy(sd_ctx* c)
{
    if (!c) return SD_ERR_ARG;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        // every sharded call waits for its own exchange, so the stream holds no collective here unless a call failed half-way
        if (wait_exchange(c, "draining the stream before ncclCommDestroy") == SD_OK && c->comm) (void)ncclCommDestroy((ncclComm_t)c->comm);
    }
    c->comm = nullptr; c->rank = 0; c->world = 1; c->job_seq = 0;
    return SD_OK;
}

extern "C" int sd_comm_init(sd_ctx* c, const void* id, int rank, int world)
{
    ENTER(c);
    if (!id || world < 1 || rank < 0 || rank >= world) SD_FAIL(c, SD_ERR_ARG, "sd_comm_init: bad rank %d / world %d", rank, world);
    if (c->comm) sd_comm_destroy(c);
    ncclUniqueId u;
    memcpy(&u, id, SD_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    // RCCL printf()s a version banner to stdout when the first communicator is created; a host program's stdout is its result
    // channel (the CLI's turn block, bench.py's JSON line), so the banner is sent to stderr
    fflush(stdout);
    const int saved = dup(1);
    if (saved >= 0) (void)dup2(2, 1);
    const ncclResult_t ir = ncclCommInitRank(&comm, world, u, rank);
    fflush(stdout);
    if (saved >= 0) { (void)dup2(saved, 1); close(saved); }
    NCCLCHK(c, ir);
    c->comm = comm; c->rank = rank; c->world = world; c->job_seq = 0;
    return SD_OK;
}

extern "C" int sd_comm_info(const sd_ctx* c, int* rank, int* world)
{
    if (!c) return SD_ERR_ARG;
    if (rank) *rank = c->rank;
    if (world) *world = c->comm ? c->world : 0;
    return SD_OK;
}

int shard_infer(sd_ctx* c, const float* d_wav, int64_t n, int64_t lo, int64_t hi, float* d_seg, float* d_emb);     // pipeline.cpp
int finalize(sd_ctx* c, const float* d_seg, const float* d_emb, int64_t chunks, int64_t n, std::vector<sd_turn>& v);
int turns_out(sd_ctx* c, const std::vector<sd_turn>& v, sd_turn** turns, int64_t* n_turns);
int pcm_to_wav(sd_ctx* c, const int16_t* d_pcm, int64_t n, float** d_wav);

// ---- failure handling of the collective step
// A rank that fails locally (bad samples, a HIP error in its inference) must not return before the all-gather: its peers would
// block in the collective forever.  Every rank therefore ALWAYS reaches the exchange and contributes a status record
// {its return code, job sequence number, n_total (two words)} that travels in the same RCCL group as the data; every rank then waits
// for the exchange with a deadline (hipEventQuery + ncclCommGetAsyncError, option "comm_timeout_ms") and reads all records, so one
// failing rank makes EVERY rank return an error for the SAME job.  A rank that disappears (crash, kill) is the deadline's case:
// the communicator is aborted (ncclCommAbort), the context has no communicator afterwards, SD_ERR_COMM is returned.
// Any non-OK return of a sharded call means the job group must be torn down (the peers' job counters no longer agree).
#define SD_STATUS_WORDS 4

static void comm_kill(sd_ctx* c)
{
    if (c->comm) (void)ncclCommAbort((ncclComm_t)c->comm);
    c->comm = nullptr; c->world = 1; c->rank = 0;
}

// waits until everything queued on the library's stream (the exchange included) has run, but never longer than the deadline
static int wait_exchange(sd_ctx* c, const char* what)
{
    hipEvent_t ev = nullptr;
    HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t er = hipEventRecord(ev, c->stream);
    if (er != hipSuccess) { (void)hipEventDestroy(ev); SD_FAIL(c, SD_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(er)); }
    const double t0 = now_ms();
    const double limit = c->comm_timeout_ms > 0 ? (double)c->comm_timeout_ms : 600000.0;
    int spins = 0;
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) {
            (void)hipEventDestroy(ev); comm_kill(c);
            SD_FAIL(c, SD_ERR_HIP, "%s: stream failed: %s", what, hipGetErrorString(q));
        }
        if (c->comm) {
            ncclResult_t ar = ncclSuccess;
            const ncclResult_t gr = ncclCommGetAsyncError((ncclComm_t)c->comm, &ar);
            if (gr != ncclSuccess || (ar != ncclSuccess && ar != ncclInProgress)) {
                const int rk = c->rank;
                (void)hipEventDestroy(ev); comm_kill(c);
                SD_FAIL(c, SD_ERR_COMM, "rank %d: %s: RCCL reports %s; communicator aborted", rk, what, ncclGetErrorString(gr != ncclSuccess ? gr : ar));
            }
        }
        const double waited = now_ms() - t0;
        if (waited > limit) {
            const int rk = c->rank, wd = c->world;
            (void)hipEventDestroy(ev); comm_kill(c);
            SD_FAIL(c, SD_ERR_COMM, "rank %d of %d: %s did not complete within %.0f ms (a peer rank never reached it); communicator aborted", rk, wd, what, limit);
        }
        if (++spins > 2000) usleep(waited > 50.0 ? 200 : 20);       // the first microseconds spin: the common case is an exchange already done
    }
    (void)hipEventDestroy(ev);
    return SD_OK;
}

extern "C" int sd_diarize_sharded_dev(sd_ctx* c, const int16_t* d_pcm_shard, int64_t first_sample, int64_t shard_samples, int64_t n,
                                      sd_turn** turns, int64_t* n_turns)
{
    ENTER(c);
    // argument errors every rank evaluates identically come first: they cannot split the group
    if (!c->comm) SD_FAIL(c, SD_ERR_ARG, "sd_diarize_sharded: no communicator (call sd_comm_init first)");
    if (!turns || !n_turns || n <= 1) SD_FAIL(c, SD_ERR_ARG, "sd_diarize_sharded: bad argument");
    *turns = nullptr; *n_turns = 0;
    const double t0 = now_ms();
    const int64_t C = sd_num_chunks(n, nullptr);
    if (C <= 0) SD_FAIL(c, SD_ERR_SHORT, "audio of %lld samples yields no chunk", (long long)n);
    // test mode (option "virtual_world" on a communicator of one rank): this rank plays every rank of a W-rank job in turn,
    // writing each shard into the slot the all-gather would put it in -- the plan and the slot assembly of an N-GPU job
    // run on a 1-GPU box (RCCL refuses two ranks on one device)
    const bool virt = c->world == 1 && c->virtual_world > 1;
    const int W = virt ? c->virtual_world : c->world;
    std::vector<int64_t> lo, hi;
    const int64_t per = plan_ranges(n, W, c->rank0_permille, lo, hi);
    const int64_t mylo = virt ? 0 : lo[(size_t)c->rank], myhi = virt ? C : hi[(size_t)c->rank];
    for (int i = 0; i < 4; ++i) c->stage_ms[i] = 0;
    const size_t seg_slot = (size_t)per * SD_FRAMES * 3, emb_slot = (size_t)per * 3 * SD_EMB_DIM;
    WS(c, float, s_seg, "mg_send_seg", seg_slot);
    WS(c, float, s_emb, "mg_send_emb", emb_slot);
    WS(c, float, g_seg, "mg_gather_seg", seg_slot * W);
    WS(c, float, g_emb, "mg_gather_emb", emb_slot * W);
    WS(c, int32_t, s_st, "mg_send_status", SD_STATUS_WORDS);
    WS(c, int32_t, g_st, "mg_gather_status", (size_t)SD_STATUS_WORDS * W);
    c->job_seq++;
    // ---- local part: from here on a failure is recorded, not returned, until the exchange has been queued
    std::vector<int32_t> st((size_t)SD_STATUS_WORDS * (size_t)W, 0);          // virt: one record per played rank; else record 0 = mine
    auto record = [&](int slot, int code) {
        int32_t* r = &st[(size_t)slot * SD_STATUS_WORDS];
        r[0] = code; r[1] = (int32_t)(c->job_seq & 0x7fffffff); r[2] = (int32_t)(n & 0xffffffffll); r[3] = (int32_t)(n >> 32);
    };
    int rc = SD_OK;
    std::string my_err;
    if (myhi > mylo) {
        const int64_t need_lo = mylo * SD_HOP;
        int64_t need_hi = (myhi - 1) * SD_HOP + SD_CHUNK; if (need_hi > n) need_hi = n;
        float* w = nullptr;
        bool played = false;                          // virt: the per-rank loop was reached (every played rank then has its own record)
        auto local = [&]() -> int {
            if (!d_pcm_shard || first_sample < 0 || shard_samples < 0 || first_sample > need_lo || first_sample + shard_samples < need_hi)
                SD_FAIL(c, SD_ERR_ARG, "rank %d: samples [%lld,%lld) do not cover chunks [%lld,%lld)", c->rank, (long long)first_sample,
                        (long long)(first_sample + shard_samples), (long long)mylo, (long long)myhi);
            int r;
            if ((r = pcm_to_wav(c, d_pcm_shard, shard_samples, &w))) return r;
            c->wav_origin = first_sample;             // kernels index the recording with absolute sample positions
            r = SD_OK;
            if (!virt) { if (c->inject_fail_rank == c->rank) SD_FAIL(c, SD_ERR_ARG, "rank %d: injected failure (option inject_fail_rank)", c->rank);
                         r = shard_infer(c, w, n, mylo, myhi, s_seg, s_emb); }
            else for (int q = 0; q < W; ++q) {
                played = true;
                int rq = SD_OK;
                if (q == c->inject_fail_rank) rq = SD_ERR_ARG;            // the played rank q "fails": what the real rank 0 would see of it
                else if (hi[(size_t)q] > lo[(size_t)q] && !r) rq = shard_infer(c, w, n, lo[(size_t)q], hi[(size_t)q], g_seg + (size_t)q * seg_slot, g_emb + (size_t)q * emb_slot);
                record(q, rq);
                if (rq && !r && q != c->inject_fail_rank) r = rq;         // a real failure of the one physical rank
            }
            return r;
        };
        rc = local();
        c->wav_origin = 0;
        if (rc) my_err = c->err;
        // a failure in front of the loop (samples that do not cover the chunks, pcm_to_wav) is every played rank's failure: without a
        // record the digest would read sequence number 0 and report "rank 0 is in another job" instead of the real code and message
        if (virt && rc && !played) for (int q = 0; q < W; ++q) record(q, rc);
    } else if (virt) for (int q = 0; q < W; ++q) record(q, SD_OK);
    if (!virt) record(0, rc);
    // ---- the exchange: always reached
    int xrc = SD_OK;
    {
        auto exchange = [&]() -> int {
            if (virt) {
                HIPCHK(c, hipMemcpyAsync(g_st, st.data(), st.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                return SD_OK;
            }
            HIPCHK(c, hipMemcpyAsync(s_st, st.data(), SD_STATUS_WORDS * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            ProfScope ps(c, "rccl_all_gather", 0, (double)(seg_slot + emb_slot) * sizeof(float) * c->world);
            NCCLCHK(c, ncclGroupStart());
            NCCLCHK(c, ncclAllGather(s_st, g_st, SD_STATUS_WORDS, ncclInt32, (ncclComm_t)c->comm, c->stream));
            NCCLCHK(c, ncclAllGather(s_seg, g_seg, seg_slot, ncclFloat, (ncclComm_t)c->comm, c->stream));
            NCCLCHK(c, ncclAllGather(s_emb, g_emb, emb_slot, ncclFloat, (ncclComm_t)c->comm, c->stream));
            NCCLCHK(c, ncclGroupEnd());
            return SD_OK;
        };
        xrc = exchange();
        if (xrc) {                                   // could not even queue the exchange: the peers will run into their deadline
            const std::string e = c->err; comm_kill(c);
            SD_FAIL(c, xrc, "%s; communicator aborted", e.c_str());
        }
    }
    if ((xrc = wait_exchange(c, "all-gather of scores and embeddings"))) return xrc;
    std::vector<int32_t> all((size_t)SD_STATUS_WORDS * (size_t)W);
    HIPCHK(c, hipMemcpy(all.data(), g_st, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int r = 0; r < W; ++r) {
        const int32_t* q = &all[(size_t)r * SD_STATUS_WORDS];
        const int64_t n_r = (int64_t)(uint32_t)q[2] | ((int64_t)q[3] << 32);
        if (q[0] != SD_OK) {
            const int code = q[0] > 0 && q[0] <= SD_ERR_COMM ? q[0] : SD_ERR_HIP;
            if (!my_err.empty() && (virt || r == c->rank)) SD_FAIL(c, code, "%s", my_err.c_str());
            SD_FAIL(c, code, "rank %d of %d failed with code %d in its part of the job (its own sd_last_error has the reason); no turns", r, W, q[0]);
        }
        if (q[1] != (int32_t)(c->job_seq & 0x7fffffff) || n_r != n)
            SD_FAIL(c, SD_ERR_COMM, "rank %d is in another job (its call #%d on %lld samples, this rank's call #%d on %lld samples): the ranks must make the same "
                    "sequence of sd_diarize_sharded calls", r, (int)q[1], (long long)n_r, (int)(c->job_seq & 0x7fffffff), (long long)n);
    }
    if (c->rank != 0) {
        // the send buffers are reused by the next job on the same stream: nothing else to wait for here.  Rank 0's finalize of this job
        // overlaps this rank's inference of the next one.
        c->stage_ms[3] = now_ms() - t0;
        return SD_OK;
    }
    // rank 0: bring the shards into chunk order (slot r holds chunks [lo_r, hi_r)); equal shares are already in place
    const float* f_seg = g_seg; const float* f_emb = g_emb;
    bool in_place = true;
    for (int r = 0; r < W; ++r) if (hi[(size_t)r] > lo[(size_t)r] && lo[(size_t)r] != (int64_t)r * per) in_place = false;
    if (!in_place) {
        WS(c, float, a_seg, "mg_asm_seg", (size_t)C * SD_FRAMES * 3);
        WS(c, float, a_emb, "mg_asm_emb", (size_t)C * 3 * SD_EMB_DIM);
        for (int r = 0; r < W; ++r) {
            const int64_t cnt = hi[(size_t)r] - lo[(size_t)r];
            if (cnt <= 0) continue;
            HIPCHK(c, hipMemcpyAsync(a_seg + (size_t)lo[(size_t)r] * SD_FRAMES * 3, g_seg + (size_t)r * seg_slot, (size_t)cnt * SD_FRAMES * 3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(a_emb + (size_t)lo[(size_t)r] * 3 * SD_EMB_DIM, g_emb + (size_t)r * emb_slot, (size_t)cnt * 3 * SD_EMB_DIM * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        }
        f_seg = a_seg; f_emb = a_emb;
    }
    std::vector<sd_turn> v;
    if ((rc = finalize(c, f_seg, f_emb, C, n, v))) return rc;
    c->stage_ms[3] = now_ms() - t0;
    return turns_out(c, v, turns, n_turns);
}

extern "C" int sd_diarize_sharded(sd_ctx* c, const int16_t* h_pcm_shard, int64_t first_sample, int64_t shard_samples, int64_t n,
                                  sd_turn** turns, int64_t* n_turns)
{
    ENTER(c);
    if (shard_samples < 0 || (shard_samples > 0 && !h_pcm_shard)) SD_FAIL(c, SD_ERR_ARG, "sd_diarize_sharded: bad argument");
    WS(c, int16_t, d_pcm, "mg_pcm", shard_samples + 16);
    if (shard_samples > 0) HIPCHK(c, hipMemcpyAsync(d_pcm, h_pcm_shard, (size_t)shard_samples * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
    return sd_diarize_sharded_dev(c, d_pcm, first_sample, shard_samples, n, turns, n_turns);
}
