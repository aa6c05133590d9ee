#!/usr/bin/env python3
"""Differential fuzz of the linkage kernels against the C oracle: tools/linkage_fuzz.py [seconds] [seed].
Random sizes (3 .. 3500), dimensions, data families (clustered, uniform, lattice = ties everywhere, duplicated rows, duplicates on a lattice / in clusters, collinear, one far outlier,
tiny scale, huge scale), workgroup counts / thread counts, kernel choice (auto, k_linkage_mw, forced heap replay with 1 .. 63 workers); Z must be
array_equal to the oracle's every time.  Prints one line per failure and a summary."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
from oracle import orc
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
d = sdhip.Diarizer(None, None)
def make(kind, N, dd):
    if kind == "blobs":
        k = int(rng.integers(1, 7)); cen = rng.standard_normal((k, dd))
        X = cen[rng.integers(0, k, N)] + rng.choice([0.05, 0.6, 2.0]) * rng.standard_normal((N, dd))
    elif kind == "uniform":
        X = rng.random((N, dd))
    elif kind == "lattice":
        side = int(np.ceil(N ** (1.0 / min(dd, 3))))
        g = np.stack(np.meshgrid(*[np.arange(float(side))] * min(dd, 3)), -1).reshape(-1, min(dd, 3))
        X = np.zeros((N, dd)); X[:, :min(dd, 3)] = g[rng.permutation(len(g))[:N]]
    elif kind == "dups":
        X = rng.standard_normal((N, dd)); m = max(1, N // int(rng.integers(2, 20)))
        X[rng.integers(0, N, m)] = X[rng.integers(0, N, m)]
    elif kind == "dups+lattice":                      # merges at height 0, then ties at height 1: the zero phase, then the whole replay
        side = int(np.ceil(N ** (1.0 / min(dd, 3))))
        g = np.stack(np.meshgrid(*[np.arange(float(side))] * min(dd, 3)), -1).reshape(-1, min(dd, 3))
        X = np.zeros((N, dd)); X[:, :min(dd, 3)] = g[rng.permutation(len(g))[:N]]
        m = max(1, N // int(rng.integers(5, 40))); X[rng.integers(0, N, m)] = X[rng.integers(0, N, m)]
    elif kind == "dups+blobs":                        # clustered rows with copies: the zero phase, then the cooperative kernel
        k = int(rng.integers(1, 7)); cen = rng.standard_normal((k, dd))
        X = cen[rng.integers(0, k, N)] + 0.6 * rng.standard_normal((N, dd)); m = max(1, N // int(rng.integers(2, 40)))
        X[rng.integers(0, N, m)] = X[rng.integers(0, N, m)]
    elif kind == "collinear":
        X = np.outer(rng.random(N), rng.standard_normal(dd))
    elif kind == "outlier":
        X = rng.standard_normal((N, dd)); X[int(rng.integers(0, N))] += 1e3
    elif kind == "tiny":
        X = 1e-150 * rng.standard_normal((N, dd))
    else:
        X = 1e120 * rng.standard_normal((N, dd))
    return np.ascontiguousarray(X, np.float64)
kinds = ["blobs", "uniform", "lattice", "dups", "dups+lattice", "dups+blobs", "collinear", "outlier", "tiny", "huge"]
t0 = time.time(); runs = 0; fails = 0; by = {}
while time.time() - t0 < budget:
    kind = kinds[int(rng.integers(0, len(kinds)))]
    N = int(rng.choice([3, 4, 7, 33, 100, 257, 800, 1499, 1500, 1501, 2200, 3500], p=[.04, .04, .04, .06, .1, .1, .12, .06, .08, .08, .14, .14]))
    dd = int(rng.choice([1, 2, 3, 8, 192]))
    X = make(kind, N, dd)
    mode = int(rng.integers(0, 4))                    # 0 auto, 1 cooperative with a random geometry, 2 k_linkage_mw, 3 forced heap replay
    G = int(rng.choice([2, 3, 5, 16, 31, 32, 64, 100])) if mode in (1, 2) else -1
    T = int(rng.choice([128, 256, 512, 1024])) if mode in (1, 2) else 0
    if N < 1500 and mode == 0: pass
    d.set_option("linkage_wgs", G); d.set_option("linkage_threads", T)
    d.set_option("linkage_kernel", 0 if mode == 2 else -1)
    d.set_option("linkage_force_heap", 1 if mode == 3 else 0)
    tk = int(rng.choice([1, 3, 31, 63])) if mode == 3 or rng.random() < 0.5 else 1
    sq = int(rng.choice([-1, 1, 0]))
    d.set_option("linkage_tie_kernel", tk)
    d.set_option("linkage_hx_wide", int(rng.random() < 0.3))
    d.set_option("linkage_square", sq)
    d.set_option("linkage_zero_phase", int(rng.random() < 0.8))
    if mode == 3: d.set_option("linkage_wgs", 16)     # (a cooperative geometry, so that the forced replay is k_linkage_hx also below N = 1500)
    try:
        Z = d.linkage(X)
        _, Zr = orc.ahc(X, orc.THRESH_F32)
        ok = np.array_equal(Z, Zr, equal_nan=True)
    except Exception as e:
        ok = False; print("EXCEPTION", kind, N, dd, mode, G, T, repr(e)[:200], flush=True)
    runs += 1; by[kind] = by.get(kind, 0) + 1
    if not ok:
        fails += 1
        rows = np.where((Z != Zr).any(1))[0] if Z.shape == Zr.shape else []
        again = 0
        for _ in range(20):
            again += 0 if np.array_equal(d.linkage(X), Zr, equal_nan=True) else 1
        print("MISMATCH kind=%s N=%d d=%d mode=%d G=%d T=%d square=%d tie_kernel=%d run %d: %d rows differ, first %s got %s want %s; the same job again: %d of 20 differ; stats rg %d hx %d tie-fallbacks %d fallbacks %d timeouts %d" % (
              kind, N, dd, mode, G, T, sq, tk, runs, len(rows), rows[:1], Z[rows[0]] if len(rows) else None, Zr[rows[0]] if len(rows) else None, again,
              d.kernel_stats("linkage_rg_launches")["launches"], d.kernel_stats("linkage_hx_jobs")["launches"], d.kernel_stats("linkage_tie_fallbacks")["launches"],
              d.kernel_stats("linkage_fallbacks")["launches"], d.kernel_stats("linkage_one_xcd_timeouts")["launches"]), flush=True)
for k in ("linkage_rg_launches", "linkage_zero_phase_jobs", "linkage_hx_jobs", "linkage_tie_fallbacks", "linkage_fallbacks", "linkage_hx_failed", "linkage_one_xcd_timeouts"):
    print(k, d.kernel_stats(k)["launches"])
print("runs %d failures %d by family %s (%.0f s)" % (runs, fails, by, time.time() - t0))
sys.exit(1 if fails else 0)
