#!/usr/bin/env python3
"""ecapa_precision = 3 (f32 tensors, split fp16 operands on the wide layers) against the f32 path: cosine distances of the real embeddings
of the planted 10-min set, NaN rows, and the time of the embedding call.  tools/x3_check.py [seconds]"""
import os, sys, tempfile, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sdhip, synth, weightpack as nn
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
tmp = tempfile.mkdtemp()
nn.save_pack(tmp + "/s.sdw", nn.synth_segmentation_weights()); nn.save_pack(tmp + "/e.sdw", nn.synth_embedding_weights())
d = sdhip.Diarizer(tmp + "/s.sdw", tmp + "/e.sdw", 0)
pcm = synth.make_pcm(seconds, seed=1234)
n = len(pcm)
nc = synth.num_chunks(n)
scores, asg = synth.planted_scores(synth.with_duets(synth.schedule(seconds, 1234)), n, 0, nc)
wav = pcm.astype(np.float32) / np.float32(32768.0)
from test_planted import nan_rule
b, masks, counts, bad = nan_rule(scores)
res = {}
for mode in (0, 3, 2, 1):
    d.set_option("ecapa_precision", mode)
    d.embed(wav, masks)
    t0 = time.perf_counter()
    e = d.embed(wav, masks)
    res[mode] = (e.astype(np.float64), (time.perf_counter() - t0) * 1e3)
c = res[0][0]
live = ~np.isnan(c[:, 0])
print("items %d live %d; f32 embed call %.1f ms" % (len(c), live.sum(), res[0][1]))
for mode in (3, 2, 1):
    a = res[mode][0]
    same_nan = np.array_equal(np.isnan(a[:, 0]), ~live)
    cd = 1 - (a[live] * c[live]).sum(1) / np.linalg.norm(a[live], axis=1) / np.linalg.norm(c[live], axis=1)
    rel = np.linalg.norm(a[live] - c[live], axis=1) / np.linalg.norm(c[live], axis=1)
    print("mode %d: %.1f ms; cosine distance to f32: median %.3g q99 %.3g max %.3g above 1e-3: %d; relative L2 median %.3g max %.3g; same NaN rows %s"
          % (mode, res[mode][1], np.median(cd), np.quantile(cd, 0.99), cd.max(), (cd > 1e-3).sum(), np.median(rel), rel.max(), same_nan))
