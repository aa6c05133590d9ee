#!/usr/bin/env python3
"""Per-kernel / per-layer HIP-event profile (option profile = 2) of one 1 h diarization:
tools/layer_profile.py [planted|raw] [skip_dead_rows 0|1] [f32|f16]"""
import os, sys, tempfile, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip, synth, weightpack as nn
workload = sys.argv[1] if len(sys.argv) > 1 else "planted"
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
tmp = tempfile.mkdtemp()
nn.save_pack(tmp + "/s.sdw", nn.synth_segmentation_weights(4321)); nn.save_pack(tmp + "/e.sdw", nn.synth_embedding_weights(4322))
d = sdhip.Diarizer(tmp + "/s.sdw", tmp + "/e.sdw", 0)
d.set_option("skip_dead_rows", skip)
d.set_option("ecapa_precision", 1 if prec == "f16" else 0)
for kv in filter(None, os.environ.get("OPTS", "").split(",")):
    k, v = kv.split("="); d.set_option(k, int(v)); print("option", k, v)
pcm = synth.make_pcm(3600, seed=1234)
n = len(pcm)
dev = torch.device("cuda", 0)
d_pcm = torch.from_numpy(pcm).to(dev)
if workload == "planted":
    nc = synth.num_chunks(n)
    sc, asg = synth.planted_scores(synth.with_duets(synth.schedule(3600, 1234)), n, 0, nc)
    d_sc, d_pe = torch.from_numpy(sc).to(dev), torch.from_numpy(synth.planted_embeddings(asg)).to(dev)
    d.set_planted(d_sc.data_ptr(), d_pe.data_ptr(), 0, nc)
torch.cuda.synchronize()
d.diarize_dev(d_pcm.data_ptr(), n)
d.set_option("profile", 2); d.reset_stats()
d.diarize_dev(d_pcm.data_ptr(), n)
names = ["chunk_norm", "pool_norm", "lstm_rec", "classifier", "stft_mel", "fbank_norm", "masked_mean", "se_mean", "se_apply", "copy_slice", "asp_stats", "asp_pool", "pdist", "linkage", "linkage_heap", "row_nn",
         "cluster_means", "assign", "mask_prefix", "wav_lens", "compact_active", "scatter_emb", "binarize_masks", "count", "activations", "topk", "conv_w256_x3", "conv_w256_f16", "x3_overflow_fallbacks"]
tags = ["sinc0", "sinc1", "sinc2", "lstm_ih", "lin0", "lin1", "block0", "tdnn1", "tdnn2", "res2net", "se1", "se2", "mfa", "asp_tdnn", "asp_tdnn_ms", "asp_conv", "fc", "asp_ms"]
tot = 0
print("workload %s, skip_dead_rows %d, precision %s" % (workload, skip, prec))
for nm in names + ["conv_gemm:" + t for t in tags] + ["skinny_gemm:" + t for t in tags]:
    s = d.kernel_stats(nm)
    if s["launches"]:
        tot += s["ms"]
        print("%-28s %5d launches %9.2f ms  %7.1f TF  %8.1f GB/s" % (nm, s["launches"], s["ms"], s["flops"] / max(s["ms"], 1e-9) / 1e9, s["bytes"] / max(s["ms"], 1e-9) / 1e6))
print("sum %.1f ms; stages" % tot, d.stage_ms())
