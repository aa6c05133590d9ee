/*
 * sdhip.h -- C ABI of libsdhip.so, the MI355X (gfx950) implementation of the
 * speaker-diarization hot path of leohuang2013/pyannote-audio_speaker-diarization_cpp.
 *
 * Every entry point names the reference interface it replaces ("sd.cpp" =
 * pipeline/src/speakerDiarizer.cpp, "cl.h/.cpp" = pipeline/src/clustering/).
 * Plain pointers and sizes only; no C++/torch types; no exceptions cross the
 * boundary.  All functions return SD_OK (0) or an SD_ERR_* code; the message is
 * available from sd_last_error().  A context is bound to one GPU and is not
 * thread-safe (same as the reference's static Ort::Env, onnx_model.cc:21-24).
 *
 * Pointer naming: h_* = host memory, d_* = device (HBM) memory of the ctx's GPU.
 */
#ifndef SDHIP_H
#define SDHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sd_ctx sd_ctx;

/* one speaker turn; replaces Annotation::Result (sd.cpp:866-876) */
typedef struct sd_turn { double start, end; int32_t label; int32_t _pad; } sd_turn;

enum {
    SD_OK = 0,
    SD_ERR_ARG = 1,      /* bad argument                                          */
    SD_ERR_HIP = 2,      /* HIP runtime failure / no GPU                          */
    SD_ERR_MODEL = 3,    /* weight file missing or malformed (reference: Ort::Exception) */
    SD_ERR_SHORT = 4,    /* audio too short for one segmentation frame (reference: UB, sd.cpp:2997) */
    SD_ERR_NUMERIC = 5,  /* zero-norm centroid (reference throws, sd.cpp:493-495) */
    SD_ERR_COMM = 6      /* multi-GPU exchange failed or timed out; the communicator has been aborted */
};

/* fixed geometry of the reference */
#define SD_SAMPLE_RATE 16000
#define SD_CHUNK 80000      /* 5.0 s, sd.cpp:1335,1411 */
#define SD_HOP 8000         /* 0.5 s, sd.cpp:1336,1412 */
#define SD_FRAMES 293       /* sd.cpp:1415 */
#define SD_SPEAKERS 3       /* sd.cpp:1351 */
#define SD_EMB_DIM 192      /* sd.cpp:2484 */
#define SD_EMB_BATCH 32     /* sd.cpp:2429 */

/* ---- context ------------------------------------------------------------
 * replaces OnnxModel::OnnxModel(path) x2 (onnx_model.cc:41-105) + SegmentModel /
 * EmbeddingModel1 construction (sd.cpp:2958, 3043).  Model files are either the
 * reference's ONNX files (segment2.onnx / emd4.onnx as written by segment/export2.py
 * and embeddings/export3.py; weights are pulled out of the graph, no ONNX runtime)
 * or ".sdw" weight packs (tools/make_weights.py).  Either path may be NULL when only
 * the other network (or only clustering) is used. */
sd_ctx* sd_create(const char* seg_model_path, const char* emb_model_path, int device_id);
void sd_destroy(sd_ctx*);
/* host-only: parse an ONNX model (kind 0 = segmentation, 1 = embedding) and write the .sdw pack */
int sd_convert_onnx(const char* onnx_path, int kind, const char* out_sdw_path);
const char* sd_convert_error(void);
const char* sd_last_error(const sd_ctx*);     /* "" when no error */
const char* sd_create_error(void);            /* reason for the last NULL from sd_create */

/* ---- a2: chunk rule of SegmentModel::slide (sd.cpp:1419, 1457) ---------- */
int64_t sd_num_chunks(int64_t n_samples, int64_t* last_chunk_len);

/* ---- a2+a3: SegmentModel::slide + ::infer (sd.cpp:1352-1504) ------------
 * wav: n float samples already scaled to [-1,1).  out: [chunks][293][3] f32. */
int sd_segment(sd_ctx*, const float* h_wav, int64_t n, float* h_out, int64_t* chunks);
int sd_segment_dev(sd_ctx*, const float* d_wav, int64_t n, float* d_out, int64_t chunks);

/* ---- a3 alone: SegmentModel::infer exactly as the reference declares it (sd.cpp:1352-1404: "input: batch size x samples count,
 * output: batch size x 293 x 3") for a host that keeps the reference's own slide() (sd.cpp:1407-1504) and swaps only the model call.
 * h_chunks [rows][T] separate waveforms (T = 80000 in slide(); a shorter last chunk is what sd.cpp:1457-1480 passes), any number of rows
 * (the reference fills its fixed batch of 32, sd.cpp:1356-1364).  h_out [rows][293][3]; *frames (may be NULL) = the frames the network
 * yields for T samples, 293 for T = 80000; frames beyond it are zero (slide()'s padding, sd.cpp:1473-1479).  Bit-identical to sd_segment
 * on the same samples. */
int sd_segment_chunks(sd_ctx*, const float* h_chunks, int64_t rows, int64_t T, float* h_out, int32_t* frames);

/* ---- a4-a6: binarize_swf, speaker_count, cleanSegmentations + mask choice
 * (sd.cpp:1506-1639, 1665-1738, 710-743, 3047-3078).
 * seg [chunks][293][3] -> binarized u8 [chunks][293][3], masks f32 [chunks*3][293],
 * count i32 [*n_count] (capacity cap_count).  Any output pointer may be NULL. */
int sd_postseg(sd_ctx*, const float* h_seg, int64_t chunks, uint8_t* h_bin,
               float* h_masks, int32_t* h_count, int64_t cap_count, int64_t* n_count);
int64_t sd_count_frames(int64_t chunks);      /* frames speaker_count produces */

/* ---- a6-a9: crop + getEmbedding + EmbeddingModel1::infer (sd.cpp:1641-1662,
 * 2436-2561, 1977-2040, 1889-1970).  Item i = (chunk i/3, local speaker i%3)
 * reads wav[chunk*8000 .. +80000) (zero padded) with mask row i.  Batches of 32
 * consecutive items share max_len exactly as the reference's batches do.
 * out: [items][192] f32, NaN rows for too-short items. */
int sd_embed(sd_ctx*, const float* h_wav, int64_t n, const float* h_masks,
             int64_t items, float* h_emb);
int sd_embed_dev(sd_ctx*, const float* d_wav, int64_t n, const float* d_masks,
                 int64_t items, int64_t first_item, float* d_emb);

/* ---- a8+a9 alone: EmbeddingModel1::infer exactly as the reference declares it (sd.cpp:1977-2040 "input: batch size x waveform, wave
 * lens; output: embedding", with _infer sd.cpp:1889-1970) for a host that keeps the reference's own getEmbedding() (sd.cpp:2436-2561:
 * interpolate, padSequence, wav_lens, NaN rule) and swaps only the model call.  h_signals [B][80000] compacted zero-padded signals,
 * h_wav_lens [B] relative lengths in (0, 1] (sd.cpp:2499-2510), any B (the reference pads its batch to 32 with lens 1.0, sd.cpp:1898-1899).
 * h_emb [B][192]; no NaN rule here -- getEmbedding applies it to what infer returns (sd.cpp:2540-2556).  Rows that sd_embed computes
 * come out bit-identical. */
int sd_embed_signals(sd_ctx*, const float* h_signals, const float* h_wav_lens, int64_t B, float* h_emb);

/* ---- a8 + head of a9 alone (operator seam for parity tests): compaction +
 * STFT + power + mel + dB + mean-norm.  signals-level inputs as for sd_embed;
 * out feats [items][501][80] f32, wav_lens [items] f32 (reference layout). */
int sd_frontend(sd_ctx*, const float* h_wav, int64_t n, const float* h_masks,
                int64_t items, float* h_feats, float* h_wav_lens);
/* ECAPA-TDNN body alone: feats [items][501][80], wav_lens [items] -> [items][192] */
int sd_ecapa(sd_ctx*, const float* h_feats, const float* h_wav_lens, int64_t items, float* h_emb);

/* ---- a12: Clustering::linkage (cl.h:8, cl.cpp:417-440): X[N][d] f64 -> Z[N-1][4] */
int sd_linkage(sd_ctx*, const double* h_X, int64_t N, int d, double* h_Z);
/* ---- a12+a13: Clustering::cluster (cl.h:7, cl.cpp:459-468): 1-based labels */
int sd_cluster(sd_ctx*, const double* h_X, int64_t N, int d, double cutoff, int32_t* h_labels1);
/* ---- a13 alone: Clustering::fcluster (cl.h:9-10, cl.cpp:442-457; criterion "distance", cl.cpp:121-232): Z [N-1][4] of N observations
 * -> 1-based labels [N], numbered as the reference numbers them.  Host arithmetic only: no GPU work, the context may be NULL.  A Z that
 * is not a dendrogram of N observations (the reference indexes with its entries unchecked) returns SD_ERR_ARG. */
int sd_fcluster(sd_ctx*, const double* h_Z, int64_t N, double cutoff, int32_t* h_labels1);
/* ---- a10+a11+a14: Cluster::clustering (sd.cpp:2063-2116): emb [chunks][3][d]
 * f64 with NaN rows -> hard clusters i32 [chunks][3]; *n_clusters = K */
int sd_clustering(sd_ctx*, const double* h_emb, int64_t chunks, int d, int32_t* h_hard, int32_t* n_clusters);
/* same with the num_clusters / min_clusters / max_clusters constraints (-1 = unset) that the reference leaves
 * unimplemented (assert(false), sd.cpp:2368-2369); semantics of clustering/Clustering.py:21-43, 352-399 */
int sd_clustering_ex(sd_ctx*, const double* h_emb, int64_t chunks, int d, int num_clusters, int min_clusters,
                     int max_clusters, int32_t* h_hard, int32_t* n_clusters);

/* ---- a15-a17: inactive mask, reconstruct, to_diarization, to_annotation
 * (sd.cpp:3172-3191, 2789-2848, 2638-2764, 2852-2935).  Returns malloc'd turns
 * sorted by start (Annotation::finalResult, sd.cpp:962-978). */
int sd_reconstruct(sd_ctx*, const float* h_seg, const uint8_t* h_bin, const int32_t* h_hard,
                   const int32_t* h_count, int64_t n_count, int64_t chunks, int64_t n_samples,
                   sd_turn** turns, int64_t* n_turns);

/* ---- whole path: speakerDiarization() (sd.cpp:2937-3234) -----------------
 * pcm: 16-bit mono 16 kHz samples as WavReader yields them (wav.h:107-111);
 * scaling by 1/32768 (sd.cpp:2950) happens on the GPU. */
int sd_diarize(sd_ctx*, const int16_t* h_pcm, int64_t n, sd_turn** turns, int64_t* n_turns);
int sd_diarize_dev(sd_ctx*, const int16_t* d_pcm, int64_t n, sd_turn** turns, int64_t* n_turns);
void sd_free_turns(sd_turn*);

/* ---- the two halves of the multi-GPU path as separate calls (a host that brings its own collective): ranks run infer on
 * their contiguous chunk range (multiple of 32 chunks), exchange d_seg / d_emb, then any rank finalizes. */
/* d_pcm_shard holds samples [first_sample, first_sample + shard_samples) of the n_total-sample
 * recording and must cover [chunk_lo*8000, min(n_total, (chunk_hi-1)*8000 + 80000)). */
int sd_shard_infer_dev(sd_ctx*, const int16_t* d_pcm_shard, int64_t first_sample, int64_t shard_samples,
                       int64_t n_total, int64_t chunk_lo, int64_t chunk_hi,
                       float* d_seg /*[hi-lo][293][3]*/, float* d_emb /*[(hi-lo)*3][192]*/);
int sd_finalize_dev(sd_ctx*, const float* d_seg, const float* d_emb, int64_t chunks, int64_t n,
                    sd_turn** turns, int64_t* n_turns);

/* ---- the same path on several GPUs under the boundary (comm.cpp): one process per GPU, RCCL all-gather of scores and
 * embeddings over xGMI on the library's stream, clustering on rank 0.  Replaces speakerDiarization() (sd.cpp:2937-3234)
 * for long recordings; the reference has no counterpart (single device, onnx_model.cc:21-71).
 * Bootstrap: rank 0 calls sd_comm_unique_id (needs a GPU) and the host program hands the SD_COMM_ID_BYTES bytes to every
 * rank; then every rank calls sd_comm_init with the same id (collective, like ncclCommInitRank).  */
#define SD_COMM_ID_BYTES 128
int sd_comm_unique_id(void* id /*[SD_COMM_ID_BYTES]*/);
int sd_comm_init(sd_ctx*, const void* id, int rank, int world);
int sd_comm_destroy(sd_ctx*);
int sd_comm_info(const sd_ctx*, int* rank, int* world /* 0 = no communicator */);
/* host-only: chunk range [ranges[2r], ranges[2r+1]) of every rank r for an n_total-sample recording -- contiguous, each
 * starting on a multiple of 32 chunks (= 3 reference embedding batches) -- and the slot size (chunks) of the padded
 * all-gather.  rank0_permille = share of the chunks rank 0 infers itself (it also finalizes), -1 = equal shares; the
 * library uses option "rank0_permille" for the same plan.  Rank r needs samples
 * [lo*8000, min(n_total, (hi-1)*8000 + 80000)). */
int sd_shard_plan(int64_t n_total, int world, int rank0_permille, int64_t* ranges /*[world][2]*/, int64_t* slot_chunks);
/* collective: every rank passes the samples of its range (h_/d_pcm_shard[0] is sample first_sample of the recording).
 * Rank 0 receives the turns; the other ranks return *n_turns = 0 once the exchange has completed (rank 0's clustering of this
 * job then overlaps their inference of the next one).  Failure is collective: a rank that fails in its part still joins the
 * exchange with a status record, and EVERY rank returns an error for that job; a rank that never arrives (crash) makes the
 * others return SD_ERR_COMM after "comm_timeout_ms" with the communicator aborted.  After any non-OK return the job group must be
 * torn down (sd_comm_destroy / exit) -- the ranks' job counters no longer agree. */
int sd_diarize_sharded(sd_ctx*, const int16_t* h_pcm_shard, int64_t first_sample, int64_t shard_samples, int64_t n_total,
                       sd_turn** turns, int64_t* n_turns);
int sd_diarize_sharded_dev(sd_ctx*, const int16_t* d_pcm_shard, int64_t first_sample, int64_t shard_samples, int64_t n_total,
                           sd_turn** turns, int64_t* n_turns);

/* ---- a1: wav::WavReader::Open (wav.h:62-126).  Returns malloc'd pcm (free with
 * sd_free_pcm); only 16-bit PCM is accepted (README.md:37), channels are read
 * interleaved-as-mono exactly like the reference (wav.h:95-97). */
int sd_read_wav(const char* path, int16_t** pcm, int64_t* n, int32_t* sample_rate, int32_t* channels);
void sd_free_pcm(int16_t*);
/* every bit depth WavReader reads (8 / 16 / 32, wav.h:99-122), as float samples already divided by 32768
 * (sd.cpp:2948-2951); free with sd_free_wav.  sd_diarize_f32 is sd_diarize for such samples. */
int sd_read_wav_f32(const char* path, float** wav, int64_t* n, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample);
void sd_free_wav(float*);
int sd_diarize_f32(sd_ctx*, const float* h_wav, int64_t n, sd_turn** turns, int64_t* n_turns);
/* ---- the resample leg of SURVEY 8(f) row 2: Resampler::Resample (frontend/resampler.cc:19-36 = libsamplerate 0.2.2 src_simple,
 * SRC_SINC_BEST_QUALITY; dormant in the reference, which reads `sample_rate` and ignores it, sd.cpp:2940-2942).  Mono float samples at
 * in_sr -> out_sr on the GPU (k_resample: exact polyphase band-limited sinc interpolation, 64 zero crossings per wing, cutoff 0.95 of
 * the lower Nyquist frequency, Kaiser 100 dB; csrc/resample.hip states what is and is not comparable with libsamplerate).
 * sd_resample_len = the reference's output length, (size_t)(n * ratio) with both factors in float (resampler.cc:21-22).
 * out == NULL: only *n_out is set. */
int64_t sd_resample_len(int64_t n, int32_t in_sr, int32_t out_sr);
int sd_resample(sd_ctx*, const float* wav, int64_t n, int32_t in_sr, int32_t out_sr, float* out, int64_t cap, int64_t* n_out);
/* ---- the head of speakerDiarization() (sd.cpp:2937-2951: WavReader, / 32768) in one call, with the input checks the reference
 * lacks (README.md:37 asks for 16 kHz / mono / 16-bit and nothing validates it).  flags = 0: a file whose sample rate is not 16 000 is
 * REFUSED with SD_ERR_ARG (the reference would process it as if it were 16 kHz); channels are read interleaved as the reference
 * does (wav.h:95-97).  SD_WAV_RESAMPLE: other rates go through sd_resample first.  SD_WAV_DOWNMIX: channels are averaged first.
 * SD_WAV_ASSUME_16K: drop-in parity with the reference on off-rate files -- the rate in the header is ignored as the reference ignores it. */
/* ---- the reference's WRITE_DATA switch (debugWrite / debugWrite2d / debugWrite3d, sd.cpp:62-234 and their call sites): with a directory
 * set, every following whole-path call (sd_diarize*, sd_finalize_dev) writes DIR/cpp_<item>.txt for the items of
 * pipeline/script/verifyEveryStepResult.py:6-17 in the reference's text format -- DIR = "/tmp" is what that script reads.  level 1: all
 * items but the two 15 - 25 MB-per-batch ones; level 2: + imasks<n>, batch_waveform<n>; level 0 / dir NULL: off.  csrc/stepdump.cpp says
 * which files are GPU tensors written as they are and which are views derived from them. */
int sd_set_dump_dir(sd_ctx*, const char* dir, int level);
#define SD_WAV_RESAMPLE 1
#define SD_WAV_DOWNMIX 2
#define SD_WAV_ASSUME_16K 4   /* the reference's behaviour on a file whose rate is not 16 000: process the samples as if it were (sd.cpp:2940-2942) */
int sd_diarize_wav(sd_ctx*, const char* path, int flags, sd_turn** turns, int64_t* n_turns);
/* ---- output formats (the reference prints raw cluster ids to stdout only, sd.cpp:3433-3441) */
/* RTTM file of the turns ("SPEAKER <uri> 1 <start> <dur> <NA> <NA> SPEAKER_kk <NA> <conf|NA>") */
int sd_write_rttm(const char* path, const char* uri, const sd_turn* turns, int64_t n_turns);
int sd_write_rttm_ex(const char* path, const char* uri, const sd_turn* turns, int64_t n_turns, const double* conf /* or NULL */);
/* renumber the labels in place: mode 1 (sd_relabel_turns) = pyannote.audio's SPEAKER_00.. convention (the labels that occur,
 * sorted by their decimal string, pyannote.core Annotation.labels()); mode 0 = order of first appearance */
int sd_relabel_turns(sd_turn* turns, int64_t n_turns);
int sd_relabel_turns_ex(sd_turn* turns, int64_t n_turns, int mode);
/* per-turn confidence of the turns the last sd_diarize* / sd_finalize_dev of this ctx returned (same order): mean soft score
 * (2 - cosine distance to the cluster centroid, soft_clusters of sd.cpp:2191-2207; range 0..2) of the (chunk, local speaker)
 * items assigned to the turn's cluster whose chunk overlaps the turn; NaN when no such item has an embedding */
int sd_last_confidence(const sd_ctx*, double* conf, int64_t cap, int64_t* n);

/* ---- a18: the reference's output line (sd.cpp:3439) */
int sd_format_turn(const sd_turn* t, char* buf, int cap);

/* ---- the reference's four stage timers (sd.cpp:53-60; labels of sd.cpp:3028, 3110, 3231, 3434): wall ms of the last
 * sd_diarize* call, [0]=segmentation [1]=embedding [2]=clustering [3]=total */
int sd_stage_ms(const sd_ctx*, double* ms4);
/* ---- options.  Product keys (the knobs clustering/Clustering.py and the multi-GPU path expose; the reference hard-codes them):
 * "num_clusters", "min_clusters", "max_clusters" (-1 = unset; Clustering.py:21-43), "constrained_assignment" (1 = constrained_argmax of
 * Clustering.py:81-94: the local speakers of a chunk go to different clusters; applies to sd_clustering* and the whole path),
 * "ecapa_precision" (0 = f32 MFMA = the reference's ORT precision (default); 1 = fp16 weights and activations on the fp16 MFMA with f32
 *   accumulation; 2 = the same with hi + lo fp16 weight planes; 3 = f32 tensors, both MFMA operands split into hi + lo fp16 halves, three
 *   products per multiply-add: f32-grade embeddings (<= 1e-7 cosine distance to mode 0) at about half of mode 0's time; a batch whose activations
 *   leave fp16's range is detected and repeated on the f32 kernels),
 *   Mode 0 is f32 storage, f32 MFMA products and f32 accumulation; its transcendentals are the hardware forms, not libm: tanh / sigmoid of
 *   the LSTM gates through v_exp_f32 + v_rcp_f32, the softmax of the attentive pooling through v_exp_f32 / v_rcp_f32 (each <= 1 ulp of
 *   the f32 result, i.e. ~1e-7 relative -- three orders below the parity tolerance rtol 1e-3 / atol 1e-4, and below what a different
 *   summation order already moves); "f32 = the reference's precision" means that class of result, not libm-bit-identical.
 * "seg_precision" (-1 = auto (default): 3 whenever "ecapa_precision" is not 0 -- a caller who asked for an fp16-pipe mode gets it in both networks --
 *   else 0; 0 = f32 MFMA; 3 = the same operand split for PyanNet's LSTM: input projections of layers 1-3 and the recurrence; scores within 2e-6 of
 *   mode 0, identical turns on the planted hour and on the reference's 1-min wav),
 * "rank0_permille" (sd_diarize_sharded: share of the chunks rank 0 infers itself, -1 = equal),
 * "comm_timeout_ms" (deadline of the exchange step of a sharded job, default 600 000).
 * Test and tuning keys are listed in sdhip_test.h.  An unknown key returns SD_ERR_ARG. */
int sd_set_option(sd_ctx*, const char* key, int64_t value);

#ifdef __cplusplus
}
#endif
#endif
