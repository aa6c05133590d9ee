/*
 * sdhip_test.h -- test, measurement and tuning hooks of libsdhip.so.
 *
 * Nothing in this header replaces a reference interface: these entry points exist for tests/, bench.py and tools/ only (the
 * drop-in boundary is sdhip.h, whose every entry cites the reference code it stands in for).  The symbols are exported by the
 * same library; a host program that only diarizes never needs this file.
 */
#ifndef SDHIP_TEST_H
#define SDHIP_TEST_H
#include "sdhip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- planted workload (measurement / test hook, SURVEY 8d: "with synthetic weights force a deterministic activity
 * pattern for stage >= a4 (override sigmoid outputs from the schedule) so N is controlled").  With seeded random
 * weights PyanNet and ECAPA do not follow the talkers, so every stage after them would only ever see one degenerate
 * case (K = 1, one turn).  After this call sd_diarize* / sd_shard_infer_dev still run both networks at full cost, then
 * replace the segmentation scores of chunks [chunk_lo, chunk_lo + chunks) by d_scores [chunks][293][3] before
 * post-segmentation, and the embedding rows of those chunks that are not NaN by the reference's own rule
 * (sd.cpp:2479-2549) by d_emb [chunks*3][192] before the all-gather / clustering.  Either pointer may be NULL; the
 * buffers stay owned by the caller and must outlive the calls; chunks = 0 removes the hook.  Never set by the CLI. */
int sd_set_planted(sd_ctx*, const float* d_scores, const float* d_emb, int64_t chunk_lo, int64_t chunks);

/* ---- measurement hooks (bench.py): GPU time of named kernels measured with hipEvents on the library's own stream
 * (option "profile" = 1), with the algorithmic FLOPs / bytes the launcher bills them. */
int sd_kernel_stats(const sd_ctx*, const char* kernel, double* total_ms, int64_t* launches, double* flops, double* bytes);
void sd_reset_stats(sd_ctx*);
/* copy `bytes` of the library's named device workspace (from byte `offset`) to the host: intermediate activations for the precision
 * diagnostics under tools/ ("ec_x0", "ec_cat", "ec_mfa", "ec_pooled", ...) */
int sd_debug_read_ws(sd_ctx*, const char* name, int64_t offset, void* h_out, int64_t bytes);
/* host only: the split-weight packing of option ecapa_precision = 3 (weights.cpp) on host buffers.  w = [K][Cout][CinPad] floats (CinPad a
 * multiple of 32, channels >= cin ignored), out_halves = 2 * K * Cout * CinPad fp16 bit patterns: per 32-channel chunk of a row
 * [hi 0..7 | lo 0..7 | hi 8..15 | ...] of w * 2^e; *inv_scale = 2^-e */
int sd_test_pack_split_weights(const float* w, int K, int Cout, int CinPad, int cin, uint16_t* out_halves, float* inv_scale);
/* test / tuning keys of sd_set_option (defaults are the measured optimum; results do not depend on the tuning keys):
 * "profile", "emb_batch_items", "seg_batch_chunks", "linkage_wgs" (-1 auto, 0 one workgroup), "linkage_threads", "linkage_one_xcd",
 * "skip_dead_rows", "virtual_world" (test mode: a communicator of ONE rank plays all W ranks of the plan in turn, slot by slot, so plan +
 * slot assembly + status exchange of a W-GPU job run on a 1-GPU box), "inject_fail_rank" (test: that rank -- a played rank under
 * virtual_world -- reports SD_ERR_ARG instead of inferring; every rank must then return an error for the job),
 * "conv_h256" / "conv_w256_f32" (256 x 256 tile for the wide ECAPA layers in fp16 / f32), "conv_glds" (fp16: that tile staged by LDS-DMA,
 * conv_gemm_g.hip, instead of through registers; same bits; default 1), "conv_glds_f32" (the same for f32: same bits, measured 5 % slower, default 0), "conv_rot" (LDS-DMA kernel, bit 0: the workgroups that
 * share a row / column panel request its four quarters in rotated order; bit 1: a wave's DMA pieces pair up inside one quarter; same bits; default 3), "conv_mfma16" (that kernel on
 * v_mfma_f32_16x16x32_f16 instead of 32x32x16: the chip holds a higher clock on it; default 1; 2 = on short contractions (block0) too: the form the round-6
 * kernel is compared with bit for bit), "conv_pp" (fp16: the wide layers with K >= 512 and M >= 2 048 on the never-drained kernel of round 6, conv_gemm_p.hip;
 * same bits as "conv_glds"; default 1), "conv_stagger" (128 x 128 f32 kernel: half of
 * the workgroups start half a tile late; 0 off (default, no effect measured), 1 / 2), "conv_w256_kmin" (shortest contraction that tile
 * takes), "conv_pn" / "conv_pn128" (column tiles per super-block), "ecapa_ld_pad" (elements added to the activation rows, multiple of 8),
 * "seg_shared_conv0" (1 = SincNet's first convolution once over the waveform instead of once per overlapping chunk), "seg_wide_ih" (1 = LSTM input
 * projections of layers 1-3 on the 256 x 256 tile), "linkage_square" (-1 auto, 0 condensed, 1 full N x N distance matrix), "ecapa_f16_hp" /
 * "ecapa_keep_cat" (precision diagnostics of tools/diag_fp16_layers.py), "linkage_kernel" (-1 auto / 1: k_linkage_rg for the square matrix where its geometry fits,
 * 0: k_linkage_mw), "linkage_tie_kernel" (what finishes a job with exact ties: 1 = k_linkage_hx, n > 1 = with n worker workgroups, 0 = k_linkage_heap),
 * "diag_res2_single" (TIMING ONLY, results are garbage: the Res2Net convolutions read one input stream instead of two -- the upper bound of what a pre-added
 * input could save, profiles/r05_res2net_single_stream.txt),
 * "linkage_zero_phase" (1, default: after a tie at height 0 the replay takes the merges at height 0 only and k_linkage_rg the rest; 0: whole replay),
 * "linkage_force_heap" (1 = skip the cooperative kernel: the heap replay on tie-free data), "linkage_hx_wide" (1 = k_linkage_hx's 32-bit key / position form, which
 * jobs above 65 535 rows take, on any size), "linkage_prefetch" (1 = k_linkage_rg's helper wave; measured: no gain), "ws_limit_mb" (test: PROCESS-WIDE, a workspace request above this many
 * MB fails as on an exhausted GPU; 0 = off) and "emb_batch_default" (test: forget an explicit "emb_batch_items"; value = embedding calls the
 * context pretends to have made, 0 = the next one plans the small first-job arena). */
/* environment (diagnostic): SD_TRACE_CREATE=1 prints where sd_create's time goes; SD_TRACE_WS=1 makes sd_diarize_dev print the job's stage times and what the
 * process's workspace hipMalloc / hipFree calls have cost so far (count, GB, ms) with every workspace of 256 MB or more. */
/* tuning hooks (tools/): time one conv_gemm shape on scratch data (dbg selects an ablation); time a grid barrier */
int sd_bench_barrier(sd_ctx*, int workgroups, int iters, int dirty_doubles, double* us_per_barrier);
/* what a merge round of k_linkage_rg is made of (linkage_rg.hip: k_rg_parts): a synthetic round with its geometry and memory pattern, parts
 * switched on by bits: 1 row loads, 2 Lance-Williams arithmetic, 4 row stores, 8 workgroup reductions, 16 slot exchange + digest */
int sd_bench_linkage_parts(sd_ctx*, int64_t N, int workgroups, int rounds, int parts, int one_xcd, double* us_per_round);
int sd_bench_conv(sd_ctx*, int64_t items, int Tp, int T, int Cin, int Cout, int KT, int dil, int has_x2, int dbg, int reps, double* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif
