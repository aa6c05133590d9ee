#!/usr/bin/env python3
"""VERDICT r04 #8, measured: can finalize(k) run under inference(k + 1) on ONE GPU?  tools/pipelined_probe.py [hours] [jobs]
Two contexts of one process (own streams): context A runs the inference half of the planted job (sd_shard_infer_dev: both networks, outputs replaced by the
planted ones), context B the finalize half (sd_finalize_dev: count, clustering with the cooperative linkage, reconstruction) of the PREVIOUS job from a
second pair of buffers.  Timed: `jobs` jobs in series on one context (the bench's step), then pipelined from two host threads (ctypes releases the GIL).
Prints both rates and whether the pipelined jobs' turns equal the serial ones'."""
import os, sys, time, threading, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip, synth, weightpack as nn
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tmp = tempfile.mkdtemp()
nn.save_pack(tmp + "/s.sdw", nn.synth_segmentation_weights(4321)); nn.save_pack(tmp + "/e.sdw", nn.synth_embedding_weights(4322))
sec = hours * 3600; n = int(sec * 16000)
pcm = synth.make_pcm(sec, seed=1234)
C_, _ = sdhip.num_chunks(n)
sc, asg = synth.planted_scores(synth.with_duets(synth.schedule(sec, 1234)), n, 0, C_)
pe = synth.planted_embeddings(asg)
dev = torch.device("cuda", 0)
d_pcm = torch.from_numpy(pcm).to(dev); d_sc = torch.from_numpy(sc).to(dev); d_pe = torch.from_numpy(pe).to(dev)
A = sdhip.Diarizer(tmp + "/s.sdw", tmp + "/e.sdw", 0)
B = sdhip.Diarizer(None, None, 0)
A.set_planted(d_sc.data_ptr(), d_pe.data_ptr(), 0, C_)
seg = [torch.zeros((C_, 293, 3), dtype=torch.float32, device=dev) for _ in range(2)]
emb = [torch.zeros((C_ * 3, 192), dtype=torch.float32, device=dev) for _ in range(2)]
def infer(k): A.shard_infer_dev(d_pcm.data_ptr(), 0, n, n, 0, C_, seg[k & 1].data_ptr(), emb[k & 1].data_ptr())
def fin(ctx, k): return ctx.finalize_dev(seg[k & 1].data_ptr(), emb[k & 1].data_ptr(), C_, n)
infer(0); t_ref = fin(A, 0); infer(1); fin(B, 1)                      # warm both contexts
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(jobs):
    infer(k); ts = fin(A, k)
torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / jobs
ok = ts == t_ref
# pipelined: the main thread infers job k + 1 while a worker thread finalizes job k on the other context
res = {}
def worker(k): res[k] = fin(B, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
infer(0)
th = None
for k in range(jobs):
    if th is not None: th.join()
    th = threading.Thread(target=worker, args=(k,)); th.start()
    if k + 1 < jobs: infer(k + 1)
th.join()
torch.cuda.synchronize()
piped = (time.perf_counter() - t0) / jobs
same = all(res[k] == t_ref for k in range(jobs))
print("%g h planted job, %d jobs: serial %.1f ms per job (%.0fx real time), finalize of job k under inference of job k + 1: %.1f ms per job (%.0fx) = %.3fx the serial rate; "
      "turns equal to the serial job's: %s (serial %s), %d turns" % (hours, jobs, serial * 1e3, sec / serial, piped * 1e3, sec / piped, serial / piped, same, ok, len(t_ref)))
