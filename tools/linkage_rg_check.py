#!/usr/bin/env python3
"""k_linkage_rg against k_linkage_mw and the oracle: tools/linkage_rg_check.py -- small bit-exactness cases first (clustered, unclustered
= retry-heavy, ties = must stop and fall back), then the planted hour: time per (kernel, G, T)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip, synth
from oracle import orc
d = sdhip.Diarizer(None, None)
d.set_option("profile", 1)
def blobs(rng, N, dd=192, k=4, s=0.6):
    cen = rng.standard_normal((k, dd))
    X = cen[rng.integers(0, k, N)] + s * rng.standard_normal((N, dd))
    return X / np.linalg.norm(X, axis=1, keepdims=True)
ok_all = True
def run(X, G, T, kern, sq=1):
    d.set_option("linkage_wgs", G); d.set_option("linkage_threads", T); d.set_option("linkage_kernel", min(kern, 1)); d.set_option("linkage_square", sq)
    d.set_option("linkage_prefetch", 1 if kern == 2 else 0)          # kernel code 2 = rg with the helper (prefetch) wave
    d.reset_stats()
    Z = d.linkage(X)
    return Z, d.kernel_stats("linkage")["ms"], d.kernel_stats("linkage_retry_rounds")["flops"], d.kernel_stats("linkage_fallbacks")["launches"], d.kernel_stats("linkage_rg_launches")["launches"]
if os.environ.get("SMALL", "1") == "1":
    cases = []
    rng = np.random.default_rng(1)
    for N, G, T in ((3, 2, 256), (65, 7, 512), (300, 64, 256), (300, 100, 128), (2000, 16, 256), (2000, 32, 128), (5000, 64, 512), (4000, 5, 256)):
        cases.append(("blobs", blobs(rng, N), G, T))
    for N, dd, G, T in ((2500, 3, 32, 256), (4000, 2, 16, 512), (3000, 8, 64, 256), (6000, 3, 32, 512)):
        cases.append(("uniform%dd" % dd, np.random.default_rng(31 * N + dd).random((N, dd)), G, T))
    Y = blobs(rng, 2400); Y[rng.integers(0, 2400, 300)] = Y[rng.integers(0, 2400, 300)]
    cases.append(("duplicates", Y, 16, 256))
    g = np.stack(np.meshgrid(np.arange(8.0), np.arange(8.0), np.arange(8.0)), -1).reshape(-1, 3)
    cases.append(("lattice", g[rng.permutation(len(g))], 16, 256))
    for name, X, G, T in cases:
        _, Zr = orc.ahc(X, orc.THRESH_F32)
        Z1, ms1, rr1, fb1, rg1 = run(X, G, T, 1)
        Z0, ms0, rr0, fb0, rg0 = run(X, G, T, 0)
        ok = np.array_equal(Z1, Zr) and np.array_equal(Z0, Zr)
        ok_all &= ok
        print("%-11s N=%5d G=%3d T=%4d  rg: equal %s %.2f ms retry %d fallbacks %d (rg launches %d) | mw: equal %s %.2f ms retry %d fallbacks %d" % (
            name, len(X), G, T, np.array_equal(Z1, Zr), ms1, rr1, fb1, rg1, np.array_equal(Z0, Zr), ms0, rr0, fb0), flush=True)
hours = float(os.environ.get("HOURS", "1"))
if hours > 0:
    sec = hours * 3600
    n = int(sec * 16000)
    nc = synth.num_chunks(n)
    sc, asg = synth.planted_scores(synth.with_duets(synth.schedule(sec, 1234)), n, 0, nc)
    emb = synth.planted_embeddings(asg).astype(np.float64)
    live = (sc > 0.4442333667381752).sum(1).reshape(-1) > 12
    X = emb[live]; X /= np.linalg.norm(X, axis=1, keepdims=True)
    Zref = None
    combos = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(0, 32, 256), (1, 32, 256), (1, 32, 128), (1, 32, 512), (1, 16, 512), (1, 16, 256), (1, 24, 256), (1, 64, 256)]
    for kern, G, T in combos:
        Z, ms, rr, fb, rg = run(X, G, T, kern, -1)
        if Zref is None: Zref = Z
        same = np.array_equal(Z, Zref); ok_all &= same
        print("planted %gh N=%d kernel %s G=%3d T=%4d: %.1f ms (%.2f us/merge) retry %d fallbacks %d same %s" % (hours, len(X), ("rg+pf" if kern == 2 else "rg") if rg else "mw", G, T, ms, ms * 1e3 / (len(X) - 1), rr, fb, same), flush=True)
print("ALL OK" if ok_all else "MISMATCH")
