#!/usr/bin/env python3
"""Differential fuzz of the non-neural stages against the C oracle: tools/stage_fuzz.py [seconds] [seed].
Random score tensors (smooth, noisy, saturated, pinned to the binarisation threshold, silent chunks), random embeddings (clustered, NaN rows, duplicated
rows, too few live rows), random chunk counts and sample counts: sd_postseg (binarised, masks, count), sd_clustering (labels, K), sd_reconstruct (turns,
order included) must equal orc.binarize / select_masks / speaker_count / clustering / reconstruct + to_annotation every time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
from oracle import orc
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
d = sdhip.Diarizer(None, None)
def scores(c, kind):
    if kind == 0:
        s = rng.random((c, 293, 3))
    elif kind == 1:                                   # smooth talk spurts
        t = np.cumsum(rng.standard_normal((c, 293, 3)) * 0.15, axis=1); s = 1 / (1 + np.exp(-t))
    elif kind == 2:                                   # saturated
        s = (rng.random((c, 293, 3)) > rng.random()).astype(np.float64)
    elif kind == 3:                                   # at and around the threshold
        s = np.float32(orc.ONSET) + rng.choice([-1, 0, 1], (c, 293, 3)) * np.float32(6e-8)
    else:
        s = rng.random((c, 293, 3)) * (rng.random((c, 1, 3)) > 0.5)      # whole local speakers silent
    return np.ascontiguousarray(s, np.float32)
t0 = time.time(); runs = 0; fails = 0
while time.time() - t0 < budget:
    c = int(rng.choice([1, 2, 3, 10, 11, 40, 150, 400]))
    sc = scores(c, int(rng.integers(0, 5)))
    ok = True; what = ""
    try:
        b, m, cnt = d.postseg(sc)
        b_ref = orc.binarize(sc)
        cnt_ref, win, ft = orc.speaker_count(b_ref)
        if not (np.array_equal(b.astype(np.float64), b_ref) and np.array_equal(m, orc.select_masks(b_ref)) and np.array_equal(cnt, cnt_ref)): ok = False; what += " postseg"
        # clustering on random embeddings
        dd = 192
        k = int(rng.integers(1, 6)); cen = rng.standard_normal((k, dd))
        emb = cen[rng.integers(0, k, c * 3)] + rng.choice([0.05, 0.6, 1.5]) * rng.standard_normal((c * 3, dd))
        emb[rng.random(c * 3) < rng.choice([0.0, 0.3, 0.9, 1.0])] = np.nan
        if c * 3 > 4 and rng.random() < 0.4:
            q = rng.integers(0, c * 3, max(1, c // 2)); emb[q] = emb[rng.integers(0, c * 3, len(q))]      # duplicated rows: exact ties
        emb = emb.reshape(c, 3, dd)
        hard, K = d.clustering(emb)
        h_ref, _, _ = orc.clustering(emb)
        if not np.array_equal(hard, h_ref): ok = False; what += " clustering"
        n_s = 80000 + 8000 * (c - 1) - int(rng.integers(0, 7000)) if c > 1 else int(rng.integers(30000, 80001))
        turns = d.reconstruct(sc, b, hard, cnt, n_s)
        binr, st = orc.reconstruct(sc, orc.mark_inactive(b_ref, h_ref), cnt_ref, win, ft, n_s)
        if turns != orc.to_annotation(binr, st): ok = False; what += " reconstruct"
    except Exception as e:
        ok = False; what += " EXCEPTION " + repr(e)[:160]
    runs += 1
    if not ok:
        fails += 1
        print("MISMATCH c=%d%s" % (c, what), flush=True)
print("runs %d failures %d (%.0f s)" % (runs, fails, time.time() - t0))
sys.exit(1 if fails else 0)
