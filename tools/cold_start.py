"""Where the cold start goes: process start -> sd_create -> PCM upload -> first job, against the warm job.
Usage (GPU box): python tools/cold_start.py [seconds]"""
import os, sys, time
t_proc = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")]
import numpy as np
import tempfile
import sdhip, synth, weightpack as wp

sec = float(sys.argv[1]) if len(sys.argv) > 1 else 3600.0
tmp = tempfile.mkdtemp()
wp.save_pack(os.path.join(tmp, "s.sdw"), wp.synth_segmentation_weights(4321))
wp.save_pack(os.path.join(tmp, "e.sdw"), wp.synth_embedding_weights(4322))
pcm = synth.make_pcm(sec, seed=1234)
import torch
t0 = time.perf_counter()
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t1 = time.perf_counter()
print("torch/hip runtime init %.1f ms" % ((t1 - t0) * 1e3))
d = sdhip.Diarizer(os.path.join(tmp, "s.sdw"), os.path.join(tmp, "e.sdw"), 0)
t2 = time.perf_counter()
print("sd_create %.1f ms" % ((t2 - t1) * 1e3))
for k in range(3):
    ta = time.perf_counter()
    turns = d.diarize(pcm)
    tb = time.perf_counter()
    print("job %d (sd_diarize, host PCM): %.1f ms, stages %s, %d turns" % (k, (tb - ta) * 1e3, [round(x, 1) for x in d.stage_ms()], len(turns)))
d.close()
