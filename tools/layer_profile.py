#!/usr/bin/env python3
"""Per-kernel / per-layer HIP-event profile (option profile = 2) of one 1 h diarization: tools/layer_profile.py"""
import os, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip, synth, weightpack as nn
tmp = tempfile.mkdtemp()
nn.save_pack(tmp + "/s.sdw", nn.synth_segmentation_weights(4321)); nn.save_pack(tmp + "/e.sdw", nn.synth_embedding_weights(4322))
d = sdhip.Diarizer(tmp + "/s.sdw", tmp + "/e.sdw", 0)
pcm = synth.make_pcm(3600, seed=1234)
d.diarize(pcm)
d.set_option("profile", 2); d.reset_stats()
d.diarize(pcm)
names = ["chunk_norm", "pool_norm", "lstm_rec", "classifier", "stft_mel", "fbank_norm", "masked_mean", "se_mean", "se_apply", "copy_slice", "asp_stats", "asp_pool", "pdist", "linkage", "row_nn",
         "cluster_means", "assign", "mask_prefix", "wav_lens", "compact_active", "nan_rows", "scatter_emb", "binarize_masks", "count", "activations", "topk"]
tags = ["sinc0", "sinc1", "sinc2", "lstm_ih", "lin0", "lin1", "block0", "tdnn1", "tdnn2", "res2net", "se1", "se2", "mfa", "asp_tdnn", "asp_tdnn_ms", "asp_conv", "fc", "blk_tdnn1", "blk_tdnn2", "res", "asp_ms"]
tot = 0
for n in names + ["conv_gemm:" + t for t in tags] + ["skinny_gemm:" + t for t in tags]:
    s = d.kernel_stats(n)
    if s["launches"]:
        tot += s["ms"]
        print("%-28s %5d launches %9.2f ms  %7.1f TF" % (n, s["launches"], s["ms"], s["flops"] / max(s["ms"], 1e-9) / 1e9))
print("sum %.1f ms; stages" % tot, d.stage_ms())
