"""debug: run-to-run identity of the ecapa stage in a precision mode: tools/x3_batch_debug.py [mode] [conv_rot]"""
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/pyannote-audio_speaker-diarization_cpp_amd")
import sdhip, weightpack as nn, tempfile
tmp = tempfile.mkdtemp()
nn.save_pack(tmp + "/s.sdw", nn.synth_segmentation_weights(4321)); nn.save_pack(tmp + "/e.sdw", nn.synth_embedding_weights(4322))
d = sdhip.Diarizer(tmp + "/s.sdw", tmp + "/e.sdw", 0)
rng = np.random.default_rng(41)
n = 2100
lens = np.full(n, 0.06, np.float32); lens[::7] = 0.2
feats = (3.0 * rng.standard_normal((n, 501, 80))).astype(np.float32)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rot = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d.set_option("ecapa_precision", mode); d.set_option("conv_rot", rot)
kmin = int(sys.argv[3]) if len(sys.argv) > 3 else 0
d.set_option("conv_w256_kmin", kmin); print("kmin", kmin)
print("ecapa_precision", mode, "conv_rot", rot)
for bi in (3072, 96):
    d.set_option("emb_batch_items", bi)
    r = [d.ecapa(feats, lens) for _ in range(5)]
    print("batch", bi, "equal to run 0:", [bool(np.array_equal(x, r[0])) for x in r], "consecutive equal:", [bool(np.array_equal(r[i], r[i + 1])) for i in range(4)],
          "items differing run3 vs run4:", np.flatnonzero((r[3] != r[4]).any(1))[:12])
