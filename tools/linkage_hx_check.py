#!/usr/bin/env python3
"""k_linkage_hx (heap replay with worker workgroups) against the oracle: ties (duplicates, lattice), tie-free data forced through it,
then the raw 1-h embeddings (oracle/_ref/raw_emb_1h.npy if present: 14 382 live rows, nearly all of them in duplicate pairs)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
from oracle import orc
d = sdhip.Diarizer(None, None)
d.set_option("profile", 1)
def blobs(rng, N, dd=192, k=4, s=0.6):
    cen = rng.standard_normal((k, dd))
    X = cen[rng.integers(0, k, N)] + s * rng.standard_normal((N, dd))
    return X / np.linalg.norm(X, axis=1, keepdims=True)
def run(X, tie_kernel, force):
    d.set_option("linkage_tie_kernel", tie_kernel); d.set_option("linkage_force_heap", force)
    d.reset_stats()
    t = time.time(); Z = d.linkage(X); wall = time.time() - t
    return Z, wall, {k: d.kernel_stats(k) for k in ("linkage", "linkage_hx", "linkage_heap", "linkage_hx_jobs", "linkage_hx_failed", "linkage_hx_stale_scans", "pdist")}
rng = np.random.default_rng(9)
cases = []
X = blobs(rng, 200); X[7] = X[3]; X[19] = X[3]; X[30] = X[11]; X[150] = X[149]
cases.append(("dup200", X, 0))
g = np.stack(np.meshgrid(np.arange(8.0), np.arange(8.0), np.arange(8.0)), -1).reshape(-1, 3)
cases.append(("lattice", g[rng.permutation(len(g))], 0))
Y = blobs(rng, 2400); Y[rng.integers(0, 2400, 300)] = Y[rng.integers(0, 2400, 300)]
cases.append(("bigdup", Y, 0))
cases.append(("blobs3000", blobs(rng, 3000), 1))
cases.append(("uniform3d", np.random.default_rng(5).random((4000, 3)), 1))
Y = blobs(rng, 6000); Y[rng.integers(0, 6000, 2000)] = Y[rng.integers(0, 6000, 2000)]
cases.append(("dup6000", Y, 0))
ok_all = True
for name, X, force in cases:
    _, Zr = orc.ahc(X, orc.THRESH_F32)
    for workers in ((1, 5, 31) if len(X) <= 4000 else (31, 63)):
        if (len(X) + workers - 1) // workers > 1024: continue
        Z, wall, st = run(X, workers if workers > 1 else 1, force)
        ok = np.array_equal(Z, Zr); ok_all &= ok
        print("%-10s N=%5d workers %3s: equal %s  hx %.1f ms (jobs %d failed %d stale scans %d) heap %.1f ms" % (name, len(X), workers, ok, st["linkage_hx"]["ms"], st["linkage_hx_jobs"]["launches"],
              st["linkage_hx_failed"]["launches"], st["linkage_hx_stale_scans"]["flops"], st["linkage_heap"]["ms"]), flush=True)
p = os.path.join(ROOT, "oracle", "_ref", "raw_emb_1h.npy")
if os.path.exists(p) and os.environ.get("RAW", "1") == "1":
    emb = np.load(p).astype(np.float64)
    X = emb[~np.isnan(emb).any(1)]
    X /= np.linalg.norm(X, axis=1, keepdims=True).astype(np.float32).astype(np.float64)
    t = time.time(); D = orc.pdist(X); Zr = orc.linkage_centroid(D, len(X)); del D
    print("oracle %.1f s" % (time.time() - t), flush=True)
    for tk in (1, 0):
        Z, wall, st = run(X, tk, 0)
        ok = np.array_equal(Z, Zr); ok_all &= ok
        print("raw 1h N=%d tie kernel %s: equal %s  cooperative %.2f ms, hx %.1f ms (stale scans %d), heap %.1f ms, pdist %.1f ms, wall %.2f s" % (len(X), "hx" if tk else "heap", ok,
              st["linkage"]["ms"], st["linkage_hx"]["ms"], st["linkage_hx_stale_scans"]["flops"], st["linkage_heap"]["ms"], st["pdist"]["ms"], wall), flush=True)
print("ALL OK" if ok_all else "MISMATCH")
