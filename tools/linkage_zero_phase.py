#!/usr/bin/env python3
"""Duplicate rows: the heap replay takes the merges at height 0, k_linkage_rg the rest (option linkage_zero_phase, run_linkage) -- against the oracle, with
the whole-replay time beside it.  Cases: clustered rows with 1 % / 5 % / 30 % duplicates at several sizes, triples, and the raw 1-h embeddings
(oracle/_ref/raw_emb_1h.npy if present)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
from oracle import orc
d = sdhip.Diarizer(None, None)
d.set_option("profile", 1)
KEYS = ("linkage", "linkage_hx", "linkage_heap", "pdist", "row_nn", "linkage_zero_phase_jobs", "linkage_zero_phase_merges", "linkage_hx_jobs", "linkage_rg_launches")
def blobs(rng, N, dd=192, k=4, s=0.6):
    cen = rng.standard_normal((k, dd))
    X = cen[rng.integers(0, k, N)] + s * rng.standard_normal((N, dd))
    return X / np.linalg.norm(X, axis=1, keepdims=True)
def run(X, zero_phase):
    d.set_option("linkage_zero_phase", zero_phase)
    d.linkage(X)                                     # (buffers)
    d.reset_stats()
    t = time.time(); Z = d.linkage(X); wall = time.time() - t
    return Z, wall, {k: d.kernel_stats(k) for k in KEYS}
rng = np.random.default_rng(21)
cases = []
for N, frac in ((1600, 0.05), (3000, 0.3), (6000, 0.01), (12989, 0.05), (12989, 0.01), (12989, 0.3), (25000, 0.05)):
    X = blobs(rng, N); m = int(N * frac)
    X[rng.integers(0, N, m)] = X[rng.integers(0, N, m)]
    cases.append(("dups %4.0f%%" % (100 * frac), X))
X = blobs(rng, 5000); X[100:200] = X[7]; X[300:310] = X[8]                       # one row a hundred times, another ten times
cases.append(("multiples", X))
p = os.path.join(ROOT, "oracle", "_ref", "raw_emb_1h.npy")
if os.path.exists(p) and os.environ.get("RAW", "1") == "1":
    emb = np.load(p).astype(np.float64)
    X = emb[~np.isnan(emb).any(1)]
    X /= np.linalg.norm(X, axis=1, keepdims=True).astype(np.float32).astype(np.float64)
    cases.append(("raw 1 h", X))
ok_all = True
for name, X in cases:
    if len(X) > int(os.environ.get("MAXN", "30000")): continue
    t = time.time(); D = orc.pdist(X); Zr = orc.linkage_centroid(D, len(X)); del D; t_or = time.time() - t
    for zp in (1, 0):
        Z, wall, st = run(X, zp)
        ok = np.array_equal(Z, Zr); ok_all &= ok
        print("%-12s N=%5d zero phase %d: equal %s  replay %.1f ms + cooperative %.1f ms (%d launches; zero-phase jobs %d, merges at height 0: %d), pdist %.1f ms, row_nn %.2f ms, wall %.1f ms  (oracle %.1f s)" % (
              name, len(X), zp, ok, st["linkage_hx"]["ms"], st["linkage"]["ms"], st["linkage_rg_launches"]["launches"], st["linkage_zero_phase_jobs"]["launches"],
              st["linkage_zero_phase_merges"]["flops"], st["pdist"]["ms"], st["row_nn"]["ms"], wall * 1e3, t_or), flush=True)
if os.environ.get("BIG", "0") == "1":        # the 8-h size: no oracle on the host (80 GB matrix); the two routes -- replay + cooperative kernel, whole replay -- against each other
    N = 100174
    X = blobs(rng, N, k=6); m = N // 100
    X[rng.integers(0, N, m)] = X[rng.integers(0, N, m)]
    out = []
    for zp in (1, 0):
        Z, wall, st = run(X, zp); out.append(Z)
        print("8 h, 1 %% duplicates N=%d zero phase %d: replay %.1f ms + cooperative %.1f ms (zero-phase jobs %d, merges at height 0: %d), pdist %.1f ms, wall %.1f ms" % (
              N, zp, st["linkage_hx"]["ms"], st["linkage"]["ms"], st["linkage_zero_phase_jobs"]["launches"], st["linkage_zero_phase_merges"]["flops"], st["pdist"]["ms"], wall * 1e3), flush=True)
    ok = np.array_equal(out[0], out[1]); ok_all &= ok
    print("8 h: both routes give the same Z: %s" % ok)
print("ALL OK" if ok_all else "MISMATCH")
