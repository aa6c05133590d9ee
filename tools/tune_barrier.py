import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
d = sdhip.Diarizer(None, None)
for G in (2, 8, 32, 64, 128, 256):
    print("G=%3d barrier %.2f us | +256 dirty doubles/WG %.2f us | +2048 %.2f us" % (G, d.bench_barrier(G, 3000, 0), d.bench_barrier(G, 3000, 256), d.bench_barrier(G, 3000, 2048)), flush=True)
