"""Mints pyannote-audio_speaker-diarization_cpp_amd/calibrated_bn_4322.npz: the BatchNorm tensors of the calibrated seeded ECAPA pack
(oracle/nn_oracle.calibrate_embedding_weights: one calibration batch through the torch oracle).  Data only: 31 x (weight, bias,
running_mean, running_var).  tests/test_planted.py re-derives them and compares."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")]
from oracle import nn_oracle as nn  # noqa: E402

seed = 4322
w = nn.calibrate_embedding_weights(seed)
w0 = nn.synth_embedding_weights(seed)
out = {k: v for k, v in w.items() if not np.array_equal(v, w0[k])}
assert all((".norm." in k or k.startswith("asp_bn.")) for k in out), sorted(out)[:5]
path = os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd", "calibrated_bn_%d.npz" % seed)
np.savez_compressed(path, **out)
print("wrote %s: %d tensors, %d bytes" % (path, len(out), os.path.getsize(path)))
