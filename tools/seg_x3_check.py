import os, sys, tempfile, numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip, synth, weightpack as nn
tmp = tempfile.mkdtemp()
nn.save_pack(tmp + "/s.sdw", nn.synth_segmentation_weights()); nn.save_pack(tmp + "/e.sdw", nn.synth_embedding_weights())
d = sdhip.Diarizer(tmp + "/s.sdw", tmp + "/e.sdw", 0)
pcm = synth.make_pcm(120.0, seed=5)
wav = pcm.astype(np.float32) / np.float32(32768.0)
s0 = d.segment(wav)
d.set_option("seg_precision", 3)
s3 = d.segment(wav)
d.set_option("seg_precision", 0)
print("chunks", s0.shape, "max abs diff", np.abs(s3 - s0).max(), "mean", np.abs(s3 - s0).mean(), "equal", np.array_equal(s3, s0), "range", s0.min(), s0.max())
