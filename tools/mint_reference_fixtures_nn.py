#!/usr/bin/env python3
"""Mints golden vectors for the numpy / torch glue AROUND the two networks from the REFERENCE'S OWN Python:

  /root/reference/segment/mysegment.py     binarize_ndarray (:356-419), embedding_mask (:150-208)
  /root/reference/embeddings/threeModel.py MySTFT (:7-66), MyNormalization (:292-396)

Neither file can be imported as a module here (their first lines import pyannote.audio / speechbrain, which this container does not
have), but the four definitions above use numpy, itertools and torch only.  The script therefore parses each file with `ast`, takes
exactly those FunctionDef / ClassDef nodes AS THEY STAND in the reference tree and executes them in a namespace holding the modules they
name: what runs is the reference's code, unedited; nothing of it is written anywhere.  Runs in the build container only; what it writes
-- tests/golden/ref_nn_glue.npz + .sha256 -- is data: inputs and the outputs the reference functions gave for them.
tests/test_reference_nn_glue.py checks the oracle (CPU) and the HIP path (GPU) against it.

    python tools/mint_reference_fixtures_nn.py          # rewrites the fixture; deterministic (seeded)

What is minted:
  bin_*    binarize_ndarray(scores[(c k), f], onset = offset = 0.5, initial_state = False)  <-> binarize_ndarray, sd.cpp:1565-1639 (a4);
           scores with many values at and next to the threshold (0.5 is a float32 number: the "neither on nor off" branch fires)
  pin_*    the same at the pipeline's threshold 0.4442333667381752 (sd.cpp:1339), the call sd_postseg makes
  em_*     embedding_mask(waveforms[32, 1, 80000], masks[32, 293]) with min_num_samples = 640 (sd.cpp:2479-2549: F.interpolate(nearest)
           of the 293-frame mask to 80 000 samples, > 0.5, compaction, wav_lens = count / max, too-short items -> 1.0, a batch whose longest
           item is too short -> all NaN)  <-> a7.  Two batches of 32 items cut from one recording the way the pipeline cuts them
           (item i = local speaker i % 3 of chunk i // 3, chunk hop 8 000 samples) + one batch that is too short as a whole
  mc_*     the mask choice of a6 (clean mask if it keeps more than ceil(293 * 640 / 80000) frames, else the full mask; sd.cpp:2430-2470).  In
           the reference's Python this rule stands in mysegment.py:forward (:436-485) BEHIND an early `return` (the author's test hook), so it
           cannot be reached by a call: the script takes those lines as they stand, puts `def _(self, waveform, sample_rate,
           binary_segmentations):` in front and `return list(iter_waveform_and_mask())` behind (the only two lines that are not the
           reference's), and runs them on the reference's own SlidingWindowFeature / SlidingWindow / crop
  crop_*   crop(waveform, 16000, Segment(t, t + 5), duration = 5.0, mode = "pad") (:226-260, with downmix_and_resample :261-291 at the native rate)
           <-> the chunk windows of a7 (sd.cpp:2567-2635): start sample floor(t * 16000), zero padding past the end of the recording
  stft_*   MySTFT(16000)(x) on float32 signals (the Python's arithmetic; the C++ runs the same torch::stft in fp64, sd.cpp:1980-2036) <-> a8
  norm_*   MyNormalization()(x, lengths) (sentence mean normalisation over the first round(len * T) frames) <-> the last step of a9's front end
"""
import ast
import hashlib
import itertools
import os
import sys
import types
from typing import Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pad_sequence

SEG = "/root/reference/segment/mysegment.py"
EMB = "/root/reference/embeddings/threeModel.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "ref_nn_glue.npz")


def take(path, names, ns):
    """execute the top-level or class-level definitions `names` of the reference file `path`, unedited, in namespace ns"""
    tree = ast.parse(open(path).read(), path)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names and node.name not in found:
            found[node.name] = node
    missing = [n for n in names if n not in found]
    if missing:
        raise SystemExit("%s no longer defines %s" % (path, missing))
    for n in names:
        mod = ast.Module(body=[found[n]], type_ignores=[])
        exec(compile(mod, path, "exec"), ns)
    return [ns[n] for n in names]


def l_is_batches(line):
    return line.strip().startswith("batches = batchify(")


def tw_for_mc(wav):
    return torch.from_numpy(wav)[None]


def main():
    if not (os.path.exists(SEG) and os.path.exists(EMB)):
        raise SystemExit("reference tree absent: fixtures can only be minted in the build container")
    class Numpy1:                 # the reference was written for numpy 1.x, where np.NAN is an alias of np.nan (removed in numpy 2); everything else is numpy
        NAN = np.nan

        def __getattr__(self, name):
            return getattr(np, name)
    ns = {"np": Numpy1(), "torch": torch, "F": F, "itertools": itertools, "pad_sequence": pad_sequence, "Optional": Optional, "Union": Union, "Tuple": Tuple}
    import math
    sys.path.insert(0, os.path.dirname(SEG))
    import utils as ref_utils                            # the reference's own module (numpy only): Segment
    ns.update({"math": math, "Segment": ref_utils.Segment, "torchaudio": None})
    binarize_ndarray, embedding_mask, crop, downmix_and_resample = take(SEG, ["binarize_ndarray", "embedding_mask", "crop", "downmix_and_resample"], ns)
    MySTFT, MyNormalization = take(EMB, ["MySTFT", "MyNormalization"], {"torch": torch})
    rng = np.random.default_rng(20261004)
    out = {}

    # ---- binarize (a4)
    c, Fr, K = 48, 293, 3
    s = rng.uniform(0.0, 1.0, (c, Fr, K)).astype(np.float32)
    near = rng.random((c, Fr, K))
    s[near < 0.15] = np.float32(0.5)                                               # exactly the threshold: neither on nor off
    s[(near >= 0.15) & (near < 0.25)] = np.nextafter(np.float32(0.5), np.float32(1.0))
    s[(near >= 0.25) & (near < 0.35)] = np.nextafter(np.float32(0.5), np.float32(0.0))
    s[3] = np.float32(0.5)                                                         # a chunk that never leaves the initial state
    s[4, :, 1] = np.float32(0.9)
    s[5, :7, :] = np.float32(0.5)                                                  # undefined until frame 7
    flat = np.ascontiguousarray(s.transpose(0, 2, 1).reshape(c * K, Fr))            # 'c f k -> (c k) f', as the Python's wrapper does
    b = binarize_ndarray(None, flat, onset=0.5, offset=None, initial_state=False)
    out["bin_scores"] = s
    out["bin_expected"] = np.ascontiguousarray(np.asarray(b).reshape(c, K, Fr).transpose(0, 2, 1)).astype(np.uint8)
    # the same at the pipeline's own threshold (sd.cpp:1339, m_diarization_segmentation_threashold): this is what sd_postseg computes.  The threshold
    # is not a float32 number, so no float32 score equals it in the C++'s double comparison (sd.cpp:1582-1597); numpy compares in float32, where
    # a score equal to float32(threshold) is neither on nor off: those frames are marked (pin_tie) -- the one regime where the two differ
    ONSET = 0.4442333667381752
    s2 = rng.uniform(0.0, 1.0, (c, Fr, K)).astype(np.float32)
    near = rng.random((c, Fr, K))
    t32 = np.float32(ONSET)
    s2[near < 0.05] = t32
    s2[(near >= 0.05) & (near < 0.15)] = np.nextafter(t32, np.float32(1.0))
    s2[(near >= 0.15) & (near < 0.25)] = np.nextafter(t32, np.float32(0.0))
    flat2 = np.ascontiguousarray(s2.transpose(0, 2, 1).reshape(c * K, Fr))
    b2 = binarize_ndarray(None, flat2, onset=ONSET, offset=None, initial_state=False)
    out["pin_onset"] = np.array([ONSET])
    out["pin_scores"] = s2
    out["pin_expected"] = np.ascontiguousarray(np.asarray(b2).reshape(c, K, Fr).transpose(0, 2, 1)).astype(np.uint8)
    out["pin_tie"] = (s2 == t32)

    # ---- embedding_mask (a7)
    chunks = 22
    n = (chunks - 1) * 8000 + 80000
    pcm = np.clip(np.round(0.3 * 32768.0 * rng.standard_normal(n)), -32768, 32767).astype(np.int16)
    wav = pcm.astype(np.float32) / np.float32(32768.0)                             # the pipeline's own conversion (a1)
    masks = np.zeros((3 * chunks, Fr), np.float32)
    for i in range(3 * chunks):
        kind = i % 11
        if kind == 0:
            masks[i] = 1.0
        elif kind == 1:
            pass                                                                   # silent local speaker
        elif kind == 2:
            masks[i, 100:102] = 1.0                                                # 2 frames = 546 samples < 640: too short
        elif kind == 3:
            masks[i, 100:103] = 1.0                                                # 3 frames = 819 samples
        elif kind == 4:
            masks[i, ::2] = 1.0                                                    # every frame boundary of the nearest map
        else:
            a, bb = sorted(rng.integers(0, Fr + 1, 2))
            masks[i, a:bb] = 1.0
            masks[i, rng.integers(0, Fr, 5)] = 1.0
    self_ = types.SimpleNamespace(min_num_samples=640, dimension=192)
    sig_all, len_all, cnt_all = [], [], []
    for b0 in (0, 32):
        w = torch.stack([torch.from_numpy(wav[(i // 3) * 8000:(i // 3) * 8000 + 80000].copy()) for i in range(b0, b0 + 32)]).unsqueeze(1)
        m = torch.from_numpy(masks[b0:b0 + 32].copy())
        signals, wav_lens = embedding_mask(self_, w, m)
        im = F.interpolate(m.unsqueeze(1), size=80000, mode="nearest").squeeze(1) > 0.5
        cnt_all.append(im.sum(1).numpy().astype(np.int64))
        sg = np.zeros((32, 80000), np.float32)
        sg[:, :signals.shape[1]] = signals.numpy()
        sig_all.append(sg)
        len_all.append(wav_lens.numpy().astype(np.float32))
    out["em_pcm"] = pcm                                                            # wav = pcm / 32768 in float32
    out["em_masks"] = masks[:64]
    out["em_counts"] = np.concatenate(cnt_all)
    out["em_wav_lens"] = np.concatenate(len_all)
    sig = np.concatenate(sig_all)
    # the compacted signals themselves: 64 x 80 000 floats would be 20 MB; their content is wav at the selected indices, so the selected
    # INDEX lists pin them: first / last selected sample and a CRC of the index list, plus the signal values at 64 probe positions
    probe = rng.integers(0, 80000, 64)
    out["em_probe"] = probe.astype(np.int64)
    out["em_signal_probe"] = sig[:, probe]
    out["em_signal_sum"] = sig.astype(np.float64).sum(1)
    short = torch.zeros(32, Fr)
    short[:, 10:12] = 1.0
    r = embedding_mask(self_, torch.zeros(32, 1, 80000), short)
    out["em_all_short_is_nan"] = np.array([isinstance(r, np.ndarray) and bool(np.isnan(r).all()) and r.shape == (32, 192)])

    # ---- mask choice (a6)
    lines = open(SEG).read().splitlines(keepends=True)
    i0 = next(i for i, l in enumerate(lines) if l.strip().startswith("duration = binary_segmentations.sliding_window.duration"))
    i1 = next(i for i in range(i0, len(lines)) if l_is_batches(lines[i]))
    body = "".join(l[4:] if l.startswith("    ") else l for l in lines[i0:i1])
    src = "def _mask_choice(self, waveform, sample_rate, binary_segmentations):\n" + body + "    return list(iter_waveform_and_mask())\n"
    ns.update({"SlidingWindowFeature": ref_utils.SlidingWindowFeature, "SlidingWindow": ref_utils.SlidingWindow})
    exec(compile(src, SEG, "exec"), ns)
    mself = types.SimpleNamespace(sample_rate=16000, min_num_samples=640)
    mself.downmix_and_resample = types.MethodType(downmix_and_resample, mself)
    mself.crop = types.MethodType(crop, mself)
    mc_chunks = 22
    bz = np.zeros((mc_chunks, Fr, 3), np.float64)
    for ci in range(mc_chunks):
        for k in range(3):
            kind = (3 * ci + k) % 7
            if kind == 0:
                bz[ci, :, k] = 1.0
            elif kind == 1:
                a = int(rng.integers(0, Fr - 8)); bz[ci, a:a + int(rng.integers(1, 8)), k] = 1.0      # 1..7 frames: around the 3-frame limit
            elif kind == 2:
                pass
            else:
                a, bb = sorted(rng.integers(0, Fr + 1, 2)); bz[ci, a:bb, k] = 1.0
    bz[5, :, :] = 0.0
    bz[5, 10:14, 0] = 1.0; bz[5, 11:14, 1] = 1.0          # clean mask keeps exactly 1 / 0 frames
    bz[6, :, :] = 0.0
    bz[6, 10:20, 0] = 1.0; bz[6, 16:30, 1] = 1.0          # clean masks keep 6 / 10 frames (> 3)
    bz[7, :, :] = 0.0
    bz[7, 10:15, 0] = 1.0; bz[7, 13:40, 1] = 1.0          # speaker 0 keeps exactly 3 clean frames: NOT more than 3 -> full mask
    swf = ref_utils.SlidingWindowFeature(bz, ref_utils.SlidingWindow(start=0.0, duration=5.0, step=0.5))
    got = ns["_mask_choice"](mself, tw_for_mc(wav), 16000, swf)
    assert len(got) == 3 * mc_chunks
    out["mc_binarized"] = bz.astype(np.uint8)
    out["mc_used_masks"] = np.stack([m[0].numpy() for _, m in got]).astype(np.float32)
    out["mc_wave_sum"] = np.array([float(w.double().sum()) for w, _ in got])

    # ---- crop (chunk windows of a7)
    cself = types.SimpleNamespace(sample_rate=16000)
    cself.downmix_and_resample = types.MethodType(downmix_and_resample, cself)
    tw = torch.from_numpy(wav)[None]
    starts = np.concatenate([np.arange(0, 31) * 0.5, [0.25, 10.03125, 11.2, 15.4999]])
    ch = []
    for t in starts:
        data, sr = crop(cself, tw, 16000, ref_utils.Segment(float(t), float(t) + 5.0), duration=5.0, mode="pad")
        assert sr == 16000 and data.shape == (1, 80000)
        ch.append(data[0].numpy())
    ch = np.stack(ch)
    out["crop_starts"] = starts
    out["crop_head"] = ch[:, :8]
    out["crop_tail"] = ch[:, -8:]
    out["crop_sum"] = ch.astype(np.float64).sum(1)
    out["crop_zeros_at_end"] = np.array([int(80000 - np.max(np.nonzero(r)[0]) - 1) if r.any() else 80000 for r in ch], np.int64)

    # ---- STFT (a8)
    st = MySTFT(16000)
    sine = np.round(8192.0 * np.sin(2 * np.pi * 440.0 * np.arange(80000) / 16000.0)).astype(np.int16)
    x = np.stack([wav[:80000], sine.astype(np.float32) / np.float32(32768.0),
                  np.concatenate([wav[5000:9000], np.zeros(76000, np.float32)])]).astype(np.float32)
    y = st(torch.from_numpy(x)).numpy()                                            # [3, 501, 201, 2] float32
    assert y.shape == (3, 501, 201, 2)
    frames = np.array([0, 1, 2, 3, 24, 25, 26, 250, 497, 498, 499, 500])
    out["stft_sine_pcm"] = sine                                                    # x = [wav[:80000], sine / 32768, wav[5000:9000] + zeros]
    out["stft_frames"] = frames
    out["stft_expected"] = y[:, frames]
    out["stft_power_sum"] = (y.astype(np.float64) ** 2).sum((2, 3))               # every frame's energy
    out["stft_window"] = st.window.numpy()

    # ---- sentence mean normalisation
    xn = (20.0 * rng.standard_normal((6, 501, 12)) - 30.0).astype(np.float32)      # (the statistics are per channel: 12 of the 80 are enough)
    lens = np.array([1.0, 0.7, 0.333, 0.05, 0.998, 0.5009980], np.float32)
    yn = MyNormalization().forward(torch.from_numpy(xn.copy()), torch.from_numpy(lens)).numpy()
    out["norm_x"] = xn
    out["norm_lens"] = lens
    out["norm_expected"] = yn

    np.savez_compressed(OUT, **out)
    h = hashlib.sha256(open(OUT, "rb").read()).hexdigest()
    open(OUT.replace(".npz", ".sha256"), "w").write("%s  ref_nn_glue.npz\n" % h)
    print("wrote %s (%d bytes), sha256 %s" % (OUT, os.path.getsize(OUT), h))
    for k, v in out.items():
        print("  %-22s %s %s" % (k, v.dtype, v.shape))


if __name__ == "__main__":
    main()
