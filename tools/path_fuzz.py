#!/usr/bin/env python3
"""Whole-path differential fuzz: tools/path_fuzz.py [seconds] [seed].  Random recording lengths around every edge of the chunk rule (sd.cpp:1419, 1457:
below one frame, one chunk exactly, one sample more, a tail of 1 .. 79 999 samples ...), random synthetic audio incl. digital silence and a looped
second; sd_diarize's turns must equal the oracle pipeline's with the GPU's own network outputs injected (every non-neural stage bit for bit, order
included), and the two networks must hold the parity tolerance against the torch oracle on the same samples."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip, synth
from oracle import nn_oracle as nn, orc, pipeline_oracle
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp(prefix="sdw_")
ws, we = nn.synth_segmentation_weights(), nn.synth_embedding_weights()
nn.save_pack(os.path.join(tmp, "s.sdw"), ws); nn.save_pack(os.path.join(tmp, "e.sdw"), we)
d = sdhip.Diarizer(os.path.join(tmp, "s.sdw"), os.path.join(tmp, "e.sdw"), 0)
base = synth.make_pcm(60.0, seed=int(rng.integers(0, 1000)))
t0 = time.time(); runs = 0; fails = 0
edges = [300, 2710, 2711, 16000, 79999, 80000, 80001, 80002, 87999, 88000, 88001, 96000, 159999, 160000, 160001]
while time.time() - t0 < budget:
    n = int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(2800, 400000))
    kind = int(rng.integers(0, 4))
    off = int(rng.integers(0, len(base) - n)) if n < len(base) else 0
    pcm = base[off:off + n].copy()
    if kind == 1: pcm[:] = 0                                                   # digital silence
    if kind == 2 and n > 32000: pcm = np.resize(pcm[:16000], n)                # one second looped: duplicated embeddings, exact ties
    if kind == 3: pcm = (pcm.astype(np.int32) * 3).clip(-32768, 32767).astype(np.int16)
    what = ""
    try:
        try:
            turns = d.diarize(pcm)
        except sdhip.SdError as e:
            nc, _ = orc.num_chunks(n)
            turns = None
            if nc > 0 and e.code != 4: what += " error %d on %d chunks" % (e.code, nc)
        if turns is not None:
            wav = pcm.astype(np.float32) / np.float32(32768.0)
            seg = d.segment(wav)
            masks = orc.select_masks(orc.binarize(seg))
            emb = d.embed(wav, masks)
            t1 = pipeline_oracle.diarize_ref(pcm, ws, we, seg_override=seg, emb_override=emb)
            if turns != t1: what += " turns(%d vs %d)" % (len(turns), len(t1))
            if runs % 4 == 0 and n <= 200000:                                   # the networks against torch on the same samples (slow: a quarter of the runs)
                nc, last = orc.num_chunks(n)
                full = nc - 1 if 0 < last < 80000 else nc
                if full > 0:
                    ref = nn.PyanNetOracle(ws)(np.stack([wav[i * 8000:i * 8000 + 80000] for i in range(full)])).numpy()
                    if not np.allclose(seg[:full], ref, rtol=1e-3, atol=1e-4): what += " segmentation"
    except Exception as e:
        what += " EXCEPTION " + repr(e)[:200]
    runs += 1
    if what:
        fails += 1
        print("MISMATCH n=%d kind=%d%s" % (n, kind, what), flush=True)
print("runs %d failures %d (%.0f s); tie fallbacks %d" % (runs, fails, time.time() - t0, d.kernel_stats("linkage_tie_fallbacks")["launches"]))
sys.exit(1 if fails else 0)
