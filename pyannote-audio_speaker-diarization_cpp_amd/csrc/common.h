// common.h -- internal declarations shared by the HIP translation units of libsdhip.so
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <functional>
#include <vector>
#include "../../include/sdhip_test.h"

#define SD_T 501          // STFT frames per item (1 + 80000/160), sd.cpp:1980-2008
#define SD_TP 501         // rows per item in activation buffers (= T: no padding rows; a 128-row tile may span two items)
#define SD_NBINS 201
#define SD_NMELS 80
#define SD_FEAT_LD 96     // mel channels padded to a multiple of 32

struct KernelStat { double ms = 0; int64_t launches = 0; double flops = 0, bytes = 0; };

// SD_TRACE_WS=1 (diagnostic): what the workspace allocations of a job cost -- hipMalloc / hipFree time and bytes, printed by sd_diarize*
// (process-wide, touched by every context's allocations: atomics)
inline std::atomic<long long> g_ws_alloc_us{0}, g_ws_free_us{0};
inline std::atomic<size_t> g_ws_alloc_bytes{0}, g_ws_allocs{0};
inline std::atomic<size_t> g_ws_limit{0};          // test hook (option ws_limit_mb): a workspace request above this many bytes fails like an exhausted GPU; 0 = off
// per-device "dynamic-LDS attribute set" masks of the wide conv kernels (the attribute belongs to the device's code object, the launch to a context)
inline std::atomic<unsigned> g_attr_w256{0}, g_attr_g256{0}, g_attr_pp{0};
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        const auto t0 = std::chrono::steady_clock::now();
        if (p) (void)hipFree(p);
        const auto t1 = std::chrono::steady_clock::now();
        p = nullptr; cap = 0;
        size_t want = bytes + (bytes >> 3) + 256;
        const size_t lim = g_ws_limit.load(std::memory_order_relaxed);
        if ((lim && bytes > lim) || hipMalloc(&p, want) != hipSuccess) { p = nullptr; (void)hipGetLastError(); return 1; }      // (the failure must not surface again at the next kernel-launch check)
        const auto t2 = std::chrono::steady_clock::now();
        g_ws_free_us += (long long)std::chrono::duration<double, std::micro>(t1 - t0).count();
        g_ws_alloc_us += (long long)std::chrono::duration<double, std::micro>(t2 - t1).count();
        g_ws_alloc_bytes += want; ++g_ws_allocs;
        cap = want; return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

// a conv / linear layer in device layout
struct ConvLayer {
    float* W = nullptr;       // [KT][Cout][CinPad] (Cin contiguous)
    void* W16 = nullptr;      // the same weights rounded to fp16 (ECAPA conv layers only; option ecapa_precision = 1)
    float* bias = nullptr;    // [Cout] or null
    float* scale = nullptr;   // folded BatchNorm (applied after act1) or null
    float* shift = nullptr;
    int Cin = 0, CinPad = 0, Cout = 0, KT = 1, dil = 1;
    int CinPad16 = 0;         // row length of W16 (multiple of 64)
    void* W16x = nullptr;     // option ecapa_precision = 3: [KT][Cout][CinPad / 8][8 hi | 8 lo] fp16 halves of W * w16x_scale (a power of two)
    float w16x_inv = 1.0f;    // 1 / w16x_scale
};

struct ConvArgs {
    const float* X; const float* X2; const float* W; float* Y;
    const void* W16;       // optional fp16 copy of W (same layout)
    const float* bias; const float* scale; const float* shift; const float* item_bias; const float* R;
    int x_ld, x2_ld, y_ld, r_ld, ib_ld;
    int w_ld;              // floats between consecutive output-channel rows of W (0 = Cin)
    int M;                 // total output rows
    int TpIn, TpOut;       // rows per item in input / output buffers
    int Tin, T;            // valid rows per item in input / output
    int Cin, Cout, KT, dil;
    int cin_real;          // un-padded input channels (algorithmic FLOP accounting only; 0 = Cin)
    int pad_mode;          // 0 = "same" with reflect padding, 1 = "valid" (src row = t + kk*dil)
    int act1, act2;        // act1: 0 none 1 relu 2 leaky(0.01); act2 (after BN): 0 none 1 tanh 2 sigmoid
    int m_tiles, n_tiles;
    int sched;             // persistent-schedule variant (set by the launcher)
    // compact row space (ECAPA): the buffers hold, item after item, only the frames that can influence a valid output
    // (need_i = min(501, nvalid_i + receptive field) rows of item i).  rowtab[g] of compact row g:
    //   .x = first compact row of g's item,  .y = frame t | (need - 1) << 10 | item << 20
    // g enumerates the OUTPUT rows; .x and the "last stored frame" refer to the INPUT buffer, which may live in a wider row
    // space of `in_rows` rows (0 = the same space, M rows): the 1x1 MFA layer reads the rows with their receptive-field margin
    // and writes only the frames < nvalid.  TpIn / TpOut / T are unused; a tap that reaches beyond the item's last stored
    // frame reads that last frame instead (it can only feed frames that are themselves beyond nvalid).  null = dense mapping.
    const int2* rowtab;
    int in_rows;
    int y_f32;             // fp16 mode: Y is float all the same (the attention logits feed an exp)
    int kt_real;           // fp16 split-weight mode: KT counts 2 * kt_real weight planes (hi, then lo) and tap kk shifts the rows like tap kk % kt_real; 0 = KT
    int prec;              // 0 = f32 (X, X2, W, Y are float), 1 = fp16 end to end (X, X2, W16, Y are _Float16; option ecapa_precision),
                           // 3 = f32 tensors, split fp16 operands on the matrix pipe (W16x; wide layers only, conv_gemm_h.hip)
    const void* W16x;      // prec 3: split weights (ConvLayer::W16x)
    float acc_scale;       // prec 3: 1 / weight scale, applied to the accumulator in the epilogue
    int stagger;           // 128 x 128 f32 kernel, short contractions: 0 = all workgroups start together, 1 / 2 = half of them start half a tile late (conv_gemm.hip)
};
#define ROWTAB_T(y) ((y) & 1023)
#define ROWTAB_LAST(y) (((y) >> 10) & 1023)
#define ROWTAB_ITEM(y) ((int)((unsigned)(y) >> 20))
#define ROWTAB_MAX_ITEMS 4095

struct EcapaWeights {
    bool loaded = false;
    float* mel_w = nullptr; int* mel_lo = nullptr; int* mel_cnt = nullptr; int* mel_off = nullptr; int mel_nnz = 0;
    float* window = nullptr;          // [400] f32 periodic Hamming
    double* tw_cos = nullptr;         // [400] cos(2 pi k/400)
    double* tw_nsin = nullptr;        // [400] -sin(2 pi k/400)
    ConvLayer block0;
    struct SERes { ConvLayer tdnn1, res[7], tdnn2, se1, se2; int dil; } blk[3];
    ConvLayer mfa, asp_tdnn_x, asp_tdnn_ms, asp_conv, fc;
    int C = 1024;
    // the fp16 copies of the per-frame conv layers are built ON THE GPU from the f32 weights the first time a mode that reads them is
    // selected (weights.cpp: ensure_ecapa_mode_weights): the default f32 mode pays nothing for them at start-up
    std::vector<ConvLayer*> conv16;   // layers that have fp16 forms
    bool have16 = false, have16x = false;
};

struct SegWeights {
    bool loaded = false;
    float wn_w = 1, wn_b = 0;          // InstanceNorm1d(1) affine on the waveform
    ConvLayer conv0, conv1, conv2;     // sincnet convs (conv0: Cin = 256 padded taps, x_ld = 10)
    float* conv0_wsum = nullptr;       // [80] sum of conv0's taps per filter (the shared-conv0 path folds the chunk normalisation into an affine map)
    float* in_w[3] = {nullptr, nullptr, nullptr};   // InstanceNorm affine per stage
    float* in_b[3] = {nullptr, nullptr, nullptr};
    ConvLayer lstm_ih[4];              // [1024][in] both directions stacked, bias = b_ih + b_hh
    float* lstm_hh[4][2] = {};         // [512][128] per layer, direction (PyTorch gate order i,f,g,o)
    void* lstm_hh_x[4][2] = {};        // option seg_precision = 3: [hi | lo][512][128] fp16 halves of W_hh * 2^e; lstm_hh_inv = 2^-e
    float lstm_hh_inv[4][2] = {};
    ConvLayer lin0, lin1;
    float* cls_w = nullptr; float* cls_b = nullptr;   // [3][128], [3]
};

// results kept for the step dumps (stepdump.cpp); filled only while dump_dir is set
struct StepStash {
    bool clustered = false;                    // run_clustering went past the "fewer than two rows" exit
    int64_t N = 0; int K = 0;
    std::vector<int> clusters, cluster_res, hard_pre;   // fcluster labels - 1, labels after the small -> large pass, [M] assignment before the inactive rule
    std::vector<double> X, Xn, soft;           // [N][d] filtered rows, the same normalised, [M][K] scores 2 - cosine distance
    int64_t ar0 = 0, cr0 = 0, rows = 0, crow_all = 0, nact = 0;   // to_diarization's crop ranges
    int64_t infer_items = 0;                   // items of the last shard_infer call (masks / counts workspaces describe them)
};

struct sd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    EcapaWeights ew;
    SegWeights sw;
    std::vector<void*> owned;                  // device allocations freed in sd_destroy
    char* warena_cur = nullptr; size_t warena_left = 0;   // weights.cpp: bump allocator over 64 MB device blocks
    std::map<std::string, DevBuf> ws;          // named workspaces
    std::set<std::string> rs_taps_filled;      // resample.hip: tap tables whose upload has completed
    std::map<std::string, KernelStat> stats;
    bool profile = false;
    bool profile_detail = false;               // also bracket conv_gemm per layer tag (option profile=2)
    std::vector<std::tuple<std::string, hipEvent_t, hipEvent_t, double, double>> pending;
    double stage_ms[4] = {0, 0, 0, 0};
    int64_t emb_batch_items = 3072;            // multiple of 96; row budget of a batch = this many full-length items (a batch also holds at most 4 095 items): ~57 GB of activation workspaces at the 1-h size, 1 % faster than 768 (fewer partly filled tile rounds and launch tails)
    bool emb_batch_explicit = false;            // set through sd_set_option: then it holds from the first call on
    int64_t embed_calls = 0;                    // run_embed calls of this context: the FIRST one plans 768-item batches (16 GB) unless the option was set --
                                                // a one-shot CLI user does not pay ~0.5 s of first-touch hipMalloc for a 57 GB arena; a context that sees more work grows it
    int64_t seg_batch_chunks = 4096;           // 128 LSTM workgroups per direction: one full wave of CUs
    int num_clusters = -1, min_clusters = -1, max_clusters = -1;   // optional constraints for the whole-path entry points
    int ecapa_precision = 0;                    // 0 = f32 MFMA (default, the measured configuration), 1 = fp16 MFMA with f32 accumulation, 2 = the same with hi + lo fp16 weight planes, 3 = f32 tensors, hi + lo split of BOTH operands on the fp16 MFMA (wide layers; the others stay f32)
    int seg_precision = -1;                     // -1 = auto: 3 whenever ecapa_precision != 0 (an fp16-pipe mode was asked for), else 0 (round 6; seg_prec() below); 0 = f32 MFMA; 3 = PyanNet's LSTM (input projections of layers 1-3 and the recurrence) with both MFMA operands split into hi + lo fp16 halves
    bool diag_res2_single = false;              // TIMING diagnostics only (results are garbage): the Res2Net convolutions read t1_i alone, without r_(i-1) -- what a pre-added input would save at most
    bool ecapa_keep_cat = false;                // diagnostics: f32 mode keeps the block outputs (the logits get their own buffer)
    int ecapa_f16_hp = 0;                       // fp16 mode: bit 0 = MFA output / pooling inputs in f32, bit 1 = attention branch on the f32 MFMA
    bool conv_w256_f32 = true;                  // f32: the same 256 x 256 kernel for the wide, long-K ECAPA layers (TDNN, MFA)
    int conv_pn128 = 0;                         // 128 x 128 kernel: column tiles per super-block (0 = 8); tuning
    int ecapa_ld_pad = 0;                       // elements added to the leading dimensions of the ECAPA activation buffers (multiple of 8)
    bool wav_padded = false;                    // the waveform buffer in use was allocated by the library with 512 zeroed floats behind the samples
    bool seg_wide_ih = true;                    // LSTM input projections of layers 1-3 on the 256 x 256 tile (identity row table); tuning
    bool seg_shared_conv0 = true;               // SincNet conv0 once over the waveform instead of once per (90 % overlapping) chunk
    int conv_w256_kmin = 0;                     // 256 x 256 kernel: shortest contraction Cin * KT it takes (0 = built-in: 1024 f32, 256 fp16); tuning
    int conv_pn = 0;                            // 256 x 256 kernel: column tiles per super-block (0 = 8); tuning
    bool conv_h256 = true;                      // fp16 mode: 256 x 256 tile kernel for the wide layers (conv_gemm_h.hip)
    int conv_mfma16 = 1;                        // fp16 LDS-DMA wide kernel on v_mfma_f32_16x16x32_f16 (higher clock under load) instead of 32x32x16; conv_gemm_g.hip
    int conv_rot = 3;                           // LDS-DMA wide kernel: the workgroups that share a row panel request its quarters in rotated order (conv_gemm_g.hip); tuning
    int conv_stagger = 0;                       // 128 x 128 f32 kernel: start half of the workgroups half a tile late (0 off, 1 odd, 2 upper half); tuning
    bool conv_glds_f32 = false;                 // f32: the LDS-DMA staged form of the wide tile (conv_gemm_g.hip, P = 0): bit-identical, measured slower; A-B only
    bool conv_pp = true;                        // fp16 mode: the never-drained kernel for the wide layers (conv_gemm_p.hip, round 6); 0 = conv_gemm_g.hip; A-B
    bool conv_glds = true;                      // fp16 mode: ... staged by LDS-DMA (conv_gemm_g.hip) instead of through registers; tuning / A-B
    bool skip_dead_rows = true;                 // ECAPA: skip row panels beyond nvalid + receptive field
    int64_t linkage_wgs = -1;                  // -1 auto, 0/1 single workgroup, else cooperative workgroups
    int64_t linkage_square = -1;               // -1 auto (full N x N distance matrix while it fits), 0 condensed, 1 square
    int64_t linkage_one_xcd = 1;               // 1 = k_linkage_mw<true> (all workgroups on one XCD) when G <= 64
    int64_t linkage_zero_phase = 1;            // a tie at height 0 (duplicate rows): the heap replay takes the merges at height 0, k_linkage_rg the rest (0 = whole replay)
    int64_t linkage_tie_kernel = 1;            // what finishes a job with exact ties: 1 = k_linkage_hx (heap replay, row work on worker workgroups; > 1 = that many workers), 0 = k_linkage_heap (one workgroup)
    int64_t linkage_prefetch = 0;              // k_linkage_rg: 1 = a helper wave per workgroup requests the rows of the runner-up neighbours ahead of time
    bool linkage_hx_wide = false;              // test: k_linkage_hx with 32-bit heap keys / positions in global memory (the form of N > 65 535) on any size
    bool linkage_force_heap = false;           // test / measurement: skip the cooperative kernel, go straight to the heap replay
    int64_t linkage_kernel = -1;               // -1 auto / 1: k_linkage_rg (linkage_rg.hip) for the square form where its geometry fits; 0: k_linkage_mw
    int64_t linkage_threads = 0;               // 0 auto (256, or 1024 for N >= 60000), else 256 / 512 / 1024 threads per cooperative workgroup
    int num_cu = 256;
    bool constrained_assignment = false;        // Clustering.py:81-94 (one cluster per local speaker of a chunk)
    std::vector<double> last_conf;               // per-turn confidence of the last finalize (sd_last_confidence)
    int64_t fe_bill_samples = -1, fe_bill_frames = 0;   // profiling: selected samples / stored frames of the next k_stft_fbank launch (-1 = unknown)
    int64_t wav_origin = 0;                     // recording position of d_wav[0] for the current call (sharded entry points hold a slice)
    void* comm = nullptr;                       // ncclComm_t (comm.cpp), null = single GPU
    int rank = 0, world = 1;
    int virtual_world = 0;                      // test mode of sd_diarize_sharded on one rank (comm.cpp)
    int64_t comm_timeout_ms = 600000;            // deadline of the exchange step of a sharded job (a peer that never arrives); then ncclCommAbort
    int64_t job_seq = 0;                        // sharded jobs since sd_comm_init (travels in the status record: ranks in different jobs are detected)
    int inject_fail_rank = -1;                  // test hook: this rank (a played rank under virtual_world) reports SD_ERR_ARG instead of inferring
    int rank0_permille = -1;                    // share of the chunks rank 0 infers itself (it also finalizes); -1 = equal shares
    std::string dump_dir;                       // sd_set_dump_dir: the next finalize writes the reference's WRITE_DATA items there
    int dump_level = 0;
    StepStash stash;
    std::string ws_failed;                      // ws_get: name of the workspace whose allocation failed last ("" = none); the embedding stage's retry reads it
    const float* planted_scores = nullptr;      // sd_set_planted: measurement / test hook
    const float* planted_emb = nullptr;
    int64_t planted_lo = 0, planted_n = 0;
};

#define SD_FAIL(ctx, code, ...) do { char _b[512]; snprintf(_b, sizeof(_b), __VA_ARGS__); (ctx)->err = _b; return (code); } while (0)
#define HIPCHK(ctx, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { SD_FAIL(ctx, SD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); } } while (0)
#define KCHECK(ctx) HIPCHK(ctx, hipGetLastError())

// workspace helper
template <class T> inline T* ws_get(sd_ctx* c, const char* name, size_t count) {
    DevBuf& b = c->ws[name];
    if (b.reserve(count * sizeof(T))) { c->ws_failed = name; return nullptr; }
    return b.as<T>();
}
#define WS(ctx, T, var, name, count) T* var = ws_get<T>(ctx, name, (size_t)(count)); if (!var) SD_FAIL(ctx, SD_ERR_HIP, "hipMalloc of workspace %s (%zu bytes) failed", name, (size_t)(count) * sizeof(T))

// profiling bracket: records events around one launch when ctx->profile is set
struct ProfScope {
    sd_ctx* c; std::string name; hipEvent_t e0 = nullptr, e1 = nullptr; double flops, bytes;
    ProfScope(sd_ctx* ctx, const std::string& n, double fl = 0, double by = 0) : c(ctx), name(n), flops(fl), bytes(by) {
        if (c->profile) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, c->stream); }
    }
    ~ProfScope() {
        if (c->profile) { (void)hipEventRecord(e1, c->stream); c->pending.emplace_back(name, e0, e1, flops, bytes); }
        else { KernelStat& s = c->stats[name]; s.launches++; s.flops += flops; s.bytes += bytes; }
    }
};
void sd_flush_profile(sd_ctx* c);   // api.cpp: resolves pending event pairs into stats

// ---- conv_gemm.hip
int launch_conv_gemm(sd_ctx* c, const ConvArgs& a, const char* tag);
// ---- conv_gemm_h.hip (fp16 mode, Cout >= 256: 256 x 256 tile; returns 1 = not applicable)
int launch_conv_gemm_h256(sd_ctx* c, const ConvArgs& a, const char* tag);
// ---- conv_gemm_g.hip (fp16 mode, Cout >= 256: the same tile with LDS-DMA staging; returns 1 = not applicable)
int launch_conv_gemm_g256(sd_ctx* c, const ConvArgs& a, const char* tag);
// ---- conv_gemm_p.hip (fp16 mode, Cout % 256 == 0, K >= 512, M >= 2 048: the never-drained form of round 6; returns 1 = not applicable)
int launch_conv_gemm_pp(sd_ctx* c, const ConvArgs& a, const char* tag);
// ---- api.cpp: the precision PyanNet's LSTM runs in: the option, or -- left at auto -- the split-operand form whenever the caller selected an fp16-pipe
// mode for ECAPA (scores within 1e-6 of the f32 path's, identical turns: tests/test_gpu_parity.py, tests/test_planted.py; 86 -> 58 ms per hour)
int seg_prec(const sd_ctx* c);
// ---- conv_narrow.hip (f32, Cout <= 96, "valid" convs of SincNet: tile as wide as the layer; returns 1 = not applicable)
int launch_conv_narrow(sd_ctx* c, const ConvArgs& a, const char* tag);
// ---- weights.cpp
struct PackTensor { std::vector<int64_t> dims; std::vector<float> data; };
typedef std::map<std::string, PackTensor> Pack;
int load_pack(const char* path, Pack& out, std::string& err);
int load_model_any(const char* path, int kind, Pack& out, std::string& err);   // onnx_reader.cpp: .sdw pack or .onnx (kind 0 seg, 1 emb)
int build_ecapa_weights(sd_ctx* c, const Pack& p);
int build_seg_weights(sd_ctx* c, const Pack& p);
void* weight_alloc(sd_ctx* c, size_t bytes);                      // weights.cpp: bump allocator over 64 MB device blocks (freed by sd_destroy)
int ensure_ecapa_mode_weights(sd_ctx* c, int ecapa_precision);   // weights_gpu.hip: builds W16 (modes 1, 2) / W16x (mode 3) on first use
// ---- frontend.hip
int frontend_prepare(sd_ctx* c, const float* d_masks, int64_t items, int64_t first_item, float* d_wav_lens, int* d_nnorm, int* d_nvalid,
                     int* d_flags, bool compact, int* h_n_active, int* d_cidx, std::vector<int>* h_nvalid = nullptr);
int frontend_features(sd_ctx* c, const float* d_wav, int64_t n, int64_t first_item, int64_t run_items, bool compact, const int* d_nnorm,
                      const int* d_rowoff, float* d_feats /*[rowoff[run_items]][96]*/, bool sig_mode = false);
int frontend_prepare_signals(sd_ctx* c, int64_t items);       // sd_embed_signals: identity gather tables for [items][80000] signal rows
// ---- ecapa.hip
// Compact row spaces of the embedding network: space s stores the frames t < min(501, nvalid + EC_MARGIN[s]) of every item.
// A layer is only computed where a later valid frame can see it (ecapa.hip): block0 / block 1 run in space 0, block 2 in space 1,
// block 3 in space 2, MFA and the attentive pooling in space 3 (the valid frames alone).
#define EC_SPACES 4
struct EcapaRowPlan {
    int64_t n = 0;
    std::vector<int> off[EC_SPACES];      // host prefix sums [n + 1]
    const int* d_off = nullptr;           // device copy [EC_SPACES][n + 1]
};
int ecapa_need_rows(int nvalid, bool skip_dead_rows);             // rows of an item in space 0
int ecapa_row_plan(sd_ctx* c, const int* h_nvalid, int64_t n, EcapaRowPlan& plan, int* d_off /*[EC_SPACES][n + 1]*/);
// items [a0, a1) of the plan; d_feats = space-0 rows of ALL the plan's items
int run_ecapa(sd_ctx* c, const float* d_feats, const int* d_nvalid /*[n]*/, const EcapaRowPlan& plan, int64_t a0, int64_t a1, float* d_emb /*[n][192]*/);
int ecapa_run_batches(sd_ctx* c, const std::function<int()>& batches);      // x3 mode: repeats the batches on the f32 kernels if an embedding came out non-finite
int run_embed(sd_ctx* c, const float* d_wav, int64_t n, const float* d_masks, int64_t items, int64_t first_item, float* d_emb);
// ---- pyannet.hip
int run_segment(sd_ctx* c, const float* d_wav, int64_t n, int64_t chunk_lo, int64_t chunk_hi, float* d_seg);
int run_segment_rows(sd_ctx* c, const float* d_rows, int64_t rows, int T, float* d_seg, int* frames);   // SegmentModel::infer as declared (sd.cpp:1352)
// ---- postseg.hip
int run_postseg(sd_ctx* c, const float* d_seg, int64_t chunks, uint8_t* d_bin, float* d_masks, int* d_nact);
int run_count(sd_ctx* c, const uint8_t* d_bin, int64_t chunks, int32_t* d_count, int64_t n_count, double* d_avg = nullptr);
// ---- stepdump.cpp
int write_step_dumps(sd_ctx* c, const float* d_seg, const float* d_emb, int64_t chunks, int64_t n_samples, const std::vector<int>& hard_post, int K);
int64_t count_frames_host(int64_t chunks);
int sd_np_rint_host(double v);
int64_t closest_frame_host(double w_start, double w_step, double w_dur, double t);
// ---- cluster.hip
int run_linkage(sd_ctx* c, const double* d_Xn, int64_t N, int d, double* d_Z);
int run_cluster_labels(sd_ctx* c, const double* d_Xn, int64_t N, int d, double cutoff, std::vector<int>& labels1, std::vector<double>* Zout = nullptr);
int run_clustering(sd_ctx* c, const double* d_emb /*[M][d] f64*/, int64_t M, int d, std::vector<int>& hard, int* K,
                   int num_clusters = -1, int min_clusters = -1, int max_clusters = -1, std::vector<double>* soft_best = nullptr);
void fcluster_host(const std::vector<double>& Z, int64_t n, double cutoff, std::vector<int>& T);
// ---- reconstruct.hip
int run_reconstruct(sd_ctx* c, const float* d_seg, const int* d_nact, const int* d_hard, const int32_t* d_count,
                    int64_t n_count, int64_t chunks, int64_t n_samples, int K, std::vector<sd_turn>& turns);
