#!/usr/bin/env python3
"""Cooperative linkage at the 8 h / 8 GPU size (N = 172 773): time + dendrogram validity.  tools/big_linkage.py [N] [G,...] [threads,...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 172773
Gs = [int(g) for g in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 128]
ALs = [int(g) for g in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2]
THs = [int(g) for g in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
d = sdhip.Diarizer(None, None)
rng = np.random.default_rng(0)
cen = rng.standard_normal((6, 192))
lab = rng.integers(0, 6, N)
X = cen[lab] + 0.6 * rng.standard_normal((N, 192))
X /= np.linalg.norm(X, axis=1, keepdims=True)
d.set_option("profile", 1)
for G, TH, AL in [(g, t, a) for a in ALs for g in Gs for t in THs]:
    d.set_option("linkage_wgs", G)
    d.set_option("linkage_threads", TH)
    d.reset_stats()
    t = time.time(); Z = d.linkage(X); t1 = time.time() - t
    st, pd = d.kernel_stats("linkage"), d.kernel_stats("pdist")
    ids = np.concatenate([Z[:, 0], Z[:, 1]]).astype(np.int64)
    ok = np.array_equal(np.sort(ids), np.arange(2 * N - 2)) and Z[-1, 3] == N and (Z[:, 0] < Z[:, 1]).all()
    print("N=%d G=%d T=%d algo=%d: pdist %.1f ms, linkage %.1f ms (%.2f us/merge), wall %.2f s, retry rounds %d, dendrogram valid %s" %
          (N, G, TH, AL, pd["ms"], st["ms"], st["ms"] * 1e3 / (N - 1), t1, d.kernel_stats("linkage_retry_rounds")["flops"], ok), flush=True)
