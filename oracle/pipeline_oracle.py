"""Whole-path oracle: speakerDiarization() (sd.cpp:2937-3234) restated on top of the C oracle
(oracle/sd_oracle.c) and the torch networks (oracle/nn_oracle.py).  TEST INFRASTRUCTURE ONLY --
used by tests/, smoke() and bench.py's cpu_baseline leg."""
import numpy as np

from . import nn_oracle as nn
from . import orc


def diarize_ref(pcm, ws, we, seg_override=None, emb_override=None, return_all=False, planted=None, wav=None):
    """pcm int16 [n] -> list of (start, end, label) sorted by start.
    seg_override / emb_override let a test inject the GPU's network outputs so the non-neural
    stages can be compared bit-for-bit.  wav = float samples already divided by 32768 (pcm is then ignored).  planted = (scores, emb) mirrors sd_set_planted: both networks run (so a timing
    includes them), then their outputs are replaced exactly as the library does it (NaN rows by rule are kept)."""
    if wav is None:
        wav = (pcm.astype(np.float32) * np.float32(1.0)) / np.float32(32768.0)  # sd.cpp:2950
    else:
        wav = np.ascontiguousarray(wav, np.float32)                             # 8 / 32-bit files: samples as the reader + sd.cpp:2950 leave them
    n = len(wav)
    nc, last_len = orc.num_chunks(n)
    if nc == 0:
        return []
    if seg_override is None:
        net = nn.PyanNetOracle(ws)
        seg = np.zeros((nc, orc.FRAMES, 3), np.float32)
        full = nc - 1 if (0 < last_len < orc.WINDOW) else nc
        for b0 in range(0, full, 32):                                           # sd.cpp:1431 batches of 32
            b1 = min(full, b0 + 32)
            x = np.stack([wav[i * orc.STEP:i * orc.STEP + orc.WINDOW] for i in range(b0, b1)])
            seg[b0:b1] = net(x).numpy()
        if full < nc:                                                           # last, shorter chunk (sd.cpp:1457-1480)
            y = net(wav[None, full * orc.STEP:]).numpy()[0]
            seg[full, :y.shape[0]] = y[:orc.FRAMES]
    else:
        seg = np.asarray(seg_override, np.float32)
    if planted is not None and planted[0] is not None:
        seg = np.asarray(planted[0], np.float32)
    binar = orc.binarize(seg)
    count, cwin, ft = orc.speaker_count(binar)
    masks = orc.select_masks(binar)
    items = nc * 3
    if emb_override is None:
        ecapa = nn.EcapaOracle(we)
        emb = np.zeros((items, 192), np.float64)
        for b0 in range(0, items, orc.EMB_BATCH):                               # sd.cpp:3083 batches of 32
            b1 = min(items, b0 + orc.EMB_BATCH)
            sigs = np.zeros((b1 - b0, orc.WINDOW), np.float32)
            cnts = np.zeros(b1 - b0, np.int64)
            for i in range(b0, b1):
                ch = orc.crop(wav, (i // 3) * orc.STEP)
                sigs[i - b0], cnts[i - b0] = orc.mask_compact(ch, masks[i])
            lens, too_short, all_nan = orc.wav_lens(cnts)
            if all_nan:
                emb[b0:b1] = np.nan
                continue
            st = nn.stft_ref(sigs, we.get("stft.window"))
            feats = nn.fbank_norm_ref(st, lens, we["fbank.matrix"])
            e = ecapa(feats, lens).numpy().astype(np.float64)
            e[too_short] = np.nan
            emb[b0:b1] = e
    else:
        emb = np.asarray(emb_override, np.float64)
    if planted is not None and planted[1] is not None:
        live = ~np.isnan(emb[:, 0])
        emb = emb.copy()
        emb[live] = np.asarray(planted[1], np.float64)[live]
    hard, K, _ = orc.clustering(emb.reshape(nc, 3, 192))
    hard = orc.mark_inactive(binar, hard)
    binary, start = orc.reconstruct(seg, hard, count, cwin, ft, n)
    turns = orc.to_annotation(binary, start)
    if return_all:
        return turns, dict(seg=seg, binarized=binar, count=count, masks=masks, emb=emb, hard=hard, K=K,
                           binary=binary, start=start)
    return turns
