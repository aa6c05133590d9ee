#!/usr/bin/env python3
"""Stress of the finalize stage against the oracle on planted inputs of several seeds / durations (the GPU tests pin seed 1234 at
10 min and 1 h): tools/stress_finalize.py [seeds] [seconds,...] -- turns bit for bit (order included) and the dendrogram Z."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sdhip, synth
from oracle import orc, pipeline_oracle
from test_planted import planted_case, nan_rule

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
durations = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [300.0, 900.0, 1800.0]
d = sdhip.Diarizer(None, None)
dev = torch.device("cuda", 0)
bad_cases = 0
for seconds in durations:
    for seed in range(100, 100 + seeds):
        pcm, scores, assign, emb = planted_case(seconds, seed, outlier_every=97 + seed % 50)
        n, nc = len(pcm), scores.shape[0]
        _, _, _, bad = nan_rule(scores)
        e32 = emb.copy(); e32[bad] = np.nan
        d_seg, d_emb = torch.from_numpy(scores).to(dev), torch.from_numpy(e32).to(dev)
        torch.cuda.synchronize()
        t0 = time.time()
        turns = d.finalize_dev(d_seg.data_ptr(), d_emb.data_ptr(), nc, n)
        t1 = time.time()
        t_ref, info = pipeline_oracle.diarize_ref(pcm, None, None, seg_override=scores, emb_override=e32.astype(np.float64), return_all=True)
        X = e32[~bad].astype(np.float64)
        Xn = X / np.sqrt((X * X).sum(1)).astype(np.float32).astype(np.float64)[:, None]
        _, Z = orc.ahc(Xn, orc.THRESH_F32)
        okz = np.array_equal(d.linkage(Xn), Z)
        ok = turns == t_ref
        bad_cases += (not ok) or (not okz)
        print("%6.0f s seed %d: N=%5d K=%d turns=%4d  gpu finalize %.0f ms  turns %s  Z %s" % (seconds, seed, len(X), info["K"], len(turns), (t1 - t0) * 1e3,
              "equal" if ok else "DIFFER", "equal" if okz else "DIFFER"), flush=True)
print("mismatching cases:", bad_cases)
sys.exit(1 if bad_cases else 0)
