#!/usr/bin/env python3
"""Time / verify the linkage kernels: tests/tune_linkage.py [N]  (scratch tuner, not collected by pytest; checks Z against the oracle)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
from oracle import orc
d = sdhip.Diarizer(None, None)
rng = np.random.default_rng(0)
def blobs(N):
    cen = rng.standard_normal((4, 192))
    X = cen[rng.integers(0, 4, N)] + 0.6 * rng.standard_normal((N, 192))
    return X / np.linalg.norm(X, axis=1, keepdims=True)
# correctness of the cooperative kernel against the oracle
for N in (2, 3, 65, 300, 2000):
    X = blobs(N)
    _, Zr = orc.ahc(X, 0.7)
    for G, TH, AL in ((0, 0, 2), (2, 256, 2), (7, 512, 2), (64, 256, 2), (64, 1024, 2), (200, 256, 2), (7, 256, 1)):
        d.set_option("linkage_wgs", G)
        d.set_option("linkage_threads", TH)
        Z = d.linkage(X)
        print("N=%d G=%d T=%d algo=%d bit-equal %s" % (N, G, TH, AL, np.array_equal(Z, Zr)), flush=True)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 21573
X = blobs(N)
d.set_option("profile", 1)
Z0 = None
for G, TH, AL in ((0, 0, 2), (64, 256, 1), (32, 256, 2), (64, 256, 2), (64, 512, 2), (64, 1024, 2), (128, 256, 2)):
    d.set_option("linkage_wgs", G)
    d.set_option("linkage_threads", TH)
    d.reset_stats()
    t = time.time(); Z = d.linkage(X); t1 = time.time() - t
    st = d.kernel_stats("linkage")
    if Z0 is None: Z0 = Z
    rr = d.kernel_stats("linkage_retry_rounds")["flops"]
    print("   flag-conservative exits:", d.kernel_stats("linkage_flag_conservative")["flops"])
    print("N=%d G=%3d T=%d algo=%d linkage kernel %.1f ms (%.2f us/merge) wall %.2f s  same-as-G0 %s retry_rounds %d" % (N, G, TH, AL, st["ms"], st["ms"] * 1e3 / (N - 1), t1, np.array_equal(Z, Z0), rr), flush=True)
