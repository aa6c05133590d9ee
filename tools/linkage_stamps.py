#!/usr/bin/env python3
"""Linkage kernel on the embeddings of the planted 1 h workload: tools/linkage_stamps.py [hours] -- kernel time per (G, T);
with SDHIP_LIB pointing at a -DSD_LINKAGE_STAMPS build the library prints the in-kernel phase split to stderr."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip, synth
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
combos = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]] or [(32, 256), (32, 512), (64, 256), (64, 512), (16, 512), (32, 1024)]
sec = hours * 3600
n = int(sec * 16000)
nc = synth.num_chunks(n)
sc, asg = synth.planted_scores(synth.with_duets(synth.schedule(sec, 1234)), n, 0, nc)
emb = synth.planted_embeddings(asg).astype(np.float64)
live = (sc > 0.4442333667381752).sum(1).reshape(-1) > 12          # rough stand-in for the reference's too-short rule
X = emb[live]
X /= np.linalg.norm(X, axis=1, keepdims=True)
N = len(X)
d = sdhip.Diarizer(None, None)
d.set_option("profile", 1)
Z0 = None
if os.environ.get("REF"):          # REF=1: the single-workgroup kernel that replays the reference's heap (bit-identical to the oracle) is the yardstick
    d.set_option("linkage_wgs", 0)
    t0 = time.time(); Z0 = d.linkage(X); print("k_linkage_heap reference: %.1f s" % (time.time() - t0), flush=True)
onex = int(os.environ.get("ONEX", "1"))
d.set_option("linkage_one_xcd", onex)
sq = int(os.environ.get("SQ", "-1"))
d.set_option("linkage_square", sq)
print("linkage_one_xcd", onex, "linkage_square", sq)
for G, T in combos:
    d.set_option("linkage_wgs", G); d.set_option("linkage_threads", T)
    d.reset_stats()
    Z = d.linkage(X)
    st = d.kernel_stats("linkage")
    rr = d.kernel_stats("linkage_retry_rounds")["flops"]
    if Z0 is None: Z0 = Z
    print("one-XCD timeouts", d.kernel_stats("linkage_one_xcd_timeouts")["launches"], end="  ")
    print("N=%d G=%3d T=%4d linkage %.1f ms (%.2f us/merge) retry rounds %d same %s" % (N, G, T, st["ms"], st["ms"] * 1e3 / (N - 1), rr, np.array_equal(Z, Z0)), flush=True)
