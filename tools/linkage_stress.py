#!/usr/bin/env python3
"""tools/linkage_stress.py kind N d G T reps [kernel] [seed]: the same linkage job `reps` times; reports every run whose Z differs from the oracle's and where."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
from oracle import orc
kind, N, dd, G, T, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
kern = int(sys.argv[7]) if len(sys.argv) > 7 else -1
seed = int(sys.argv[8]) if len(sys.argv) > 8 else 0
rng = np.random.default_rng(seed)
if kind == "uniform":
    X = rng.random((N, dd))
elif kind == "blobs":
    k = 4; cen = rng.standard_normal((k, dd)); X = cen[rng.integers(0, k, N)] + 0.6 * rng.standard_normal((N, dd))
elif kind == "dups":          # clustered rows, 5 % of them copies of other rows: the zero phase (replay for the merges at height 0, then k_linkage_rg)
    k = 4; cen = rng.standard_normal((k, dd)); X = cen[rng.integers(0, k, N)] + 0.6 * rng.standard_normal((N, dd))
    X[rng.integers(0, N, N // 20)] = X[rng.integers(0, N, N // 20)]
else:
    X = rng.standard_normal((N, dd))
_, Zr = orc.ahc(X, orc.THRESH_F32)
d = sdhip.Diarizer(None, None)
d.set_option("linkage_wgs", G); d.set_option("linkage_threads", T); d.set_option("linkage_kernel", kern)
for opt in os.environ.get("OPTS", "").split(","):
    if opt: k, v = opt.split("="); d.set_option(k, int(v))
bad = 0
for r in range(reps):
    Z = d.linkage(X)
    if not np.array_equal(Z, Zr):
        bad += 1
        rows = np.where((Z != Zr).any(1))[0]
        k0 = rows[0]
        print("run %d: %d rows differ, first %d: got %s want %s" % (r, len(rows), k0, Z[k0], Zr[k0]), flush=True)
print("%s N=%d d=%d G=%d T=%d kernel %d: %d of %d runs differ; retry rounds %d rg launches %d fallbacks %d one-xcd timeouts %d" % (kind, N, dd, G, T, kern, bad, reps,
      d.kernel_stats("linkage_retry_rounds")["flops"], d.kernel_stats("linkage_rg_launches")["launches"], d.kernel_stats("linkage_fallbacks")["launches"], d.kernel_stats("linkage_one_xcd_timeouts")["launches"]))
