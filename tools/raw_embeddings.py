#!/usr/bin/env python3
"""Embeddings of the RAW 1 h workload (random-weight networks on synthetic audio): tools/raw_embeddings.py [hours] [out.npy].
Saves the [chunks*3][192] f32 embeddings (NaN rows included) so that the tie structure clustering meets on them (duplicated
rows: looped audio / digital silence) can be studied on the CPU, and prints the exact-duplicate statistics."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip, synth
import weightpack as nn
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "raw_emb_%gh.npy" % hours)
tmp = tempfile.mkdtemp(prefix="sdw_")
nn.save_pack(os.path.join(tmp, "segment.sdw"), nn.synth_segmentation_weights(4321))
nn.save_pack(os.path.join(tmp, "embedding.sdw"), nn.synth_embedding_weights(4322))
d = sdhip.Diarizer(os.path.join(tmp, "segment.sdw"), os.path.join(tmp, "embedding.sdw"), 0)
pcm = synth.make_pcm(hours * 3600, seed=1234)
wav = pcm.astype(np.float32) / 32768.0
seg = d.segment(wav)
from oracle import orc
masks = orc.select_masks(orc.binarize(seg))
emb = d.embed(wav, masks)
os.makedirs(os.path.dirname(out), exist_ok=True)
np.save(out, emb.astype(np.float32))
live = ~np.isnan(emb).any(1)
X = emb[live]
u, inv, cnt = np.unique(X.view(np.uint8).reshape(len(X), -1), axis=0, return_inverse=True, return_counts=True)
print("items %d live %d distinct %d rows-in-duplicate-groups %d largest group %d" % (len(emb), live.sum(), len(u), int(cnt[cnt > 1].sum()), int(cnt.max())))
