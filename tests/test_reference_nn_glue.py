"""The numpy / torch glue around the two networks pinned on the REFERENCE'S OWN Python: tests/golden/ref_nn_glue.npz was minted by
tools/mint_reference_fixtures_nn.py from the unedited definitions of binarize_ndarray / embedding_mask
(/root/reference/segment/mysegment.py:356-419, 150-208) and MySTFT / MyNormalization (/root/reference/embeddings/threeModel.py:7-66,
292-396) -- the Python the C++ at sd.cpp:746-767, 2479-2549, 1980-2036 and the front half of emd4.onnx were derived from.
CPU: the C / torch oracle against the vectors; GPU: sd_postseg and sd_frontend of libsdhip against the same vectors.

Where the C++ deliberately differs from the Python, the test says what and asserts the regime:
  STFT   the Python runs torch.stft in float32, the C++ the same call in fp64 and narrows (sd.cpp:1980-2036): equal to float32 rounding
         (<= 2e-6 of the frame's largest bin), every frame's energy to 1e-5 relative
"""
import hashlib
import os

import numpy as np
import pytest
import torch

import sdhip
from oracle import nn_oracle as nn, orc


@pytest.fixture(scope="module")
def gold(golden_dir):
    path = os.path.join(golden_dir, "ref_nn_glue.npz")
    want = open(os.path.join(golden_dir, "ref_nn_glue.sha256")).read().split()[0]
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == want, "fixture and manifest disagree: re-mint both"
    return np.load(path)


def test_binarize_against_reference_python(gold):
    """a4: hysteresis thresholding with onset = offset = 0.5 and initial state off: a frame is on iff the last frame that was not exactly
    0.5 was above it -- including scores one ulp either side of the threshold, chunks that never leave the initial state, a late start"""
    s, want = gold["bin_scores"], gold["bin_expected"]
    got = orc.binarize(s, 0.5)
    assert np.array_equal(got.astype(np.uint8), want)
    assert (s == np.float32(0.5)).mean() > 0.1 and 0.2 < want.mean() < 0.8 and want[3].sum() == 0 and want[4, :, 1].all()


def _pin_case(gold):
    """the pipeline's threshold (sd.cpp:1339) is not a float32 number.  The C++ compares in double (sd.cpp:1582-1597), so a float32 score is always
    on or off; numpy compares in float32, where a score equal to float32(threshold) is neither and copies the previous state.  On those frames
    (pin_tie) the C++ semantics are asserted -- float32(threshold) lies BELOW the threshold, so they are off; everywhere else the two
    agree frame by frame (no state is ever carried past a defined frame)"""
    s, want, tie = gold["pin_scores"], gold["pin_expected"], gold["pin_tie"]
    onset = float(gold["pin_onset"][0])
    assert onset == orc.ONSET and float(np.float32(onset)) < onset
    return s, want, tie


def test_binarize_at_the_pipeline_threshold_against_reference_python(gold):
    s, want, tie = _pin_case(gold)
    got = orc.binarize(s).astype(np.uint8)
    assert np.array_equal(got[~tie], want[~tie])
    assert (got[tie] == 0).all() and tie.sum() > 1000 and want[tie].sum() > 100      # (the Python has some of them on: the copied state)


def _em_case(gold):
    wav = gold["em_pcm"].astype(np.float32) / np.float32(32768.0)
    return wav, gold["em_masks"]


def test_mask_interpolation_compaction_and_wav_lens_against_reference_python(gold):
    """a7: F.interpolate(mode="nearest") of a 293-frame mask onto 80 000 samples, > 0.5, compaction of the selected samples, wav_lens =
    count / batch maximum, too-short items (< 640 samples) -> 1.0 (mysegment.py:150-208 <-> Helper::interpolate sd.cpp:746-767 and the
    loop of sd.cpp:2479-2549)"""
    wav, masks = _em_case(gold)
    counts = np.zeros(64, np.int64)
    sums = np.zeros(64)
    probe = gold["em_probe"]
    pv = np.zeros((64, len(probe)), np.float32)
    for i in range(64):
        sig, n = orc.mask_compact(orc.crop(wav, (i // 3) * 8000), masks[i])
        counts[i] = n
        full = np.zeros(80000, np.float32)
        full[:n] = sig[:n]
        sums[i] = full.astype(np.float64).sum()
        pv[i] = full[probe]
    assert np.array_equal(counts, gold["em_counts"])                 # the nearest-neighbour map selects exactly the reference's samples ...
    assert np.array_equal(pv, gold["em_signal_probe"]) and np.array_equal(sums, gold["em_signal_sum"])     # ... in the reference's order
    for b0 in (0, 32):
        lens, ts, an = orc.wav_lens(counts[b0:b0 + 32])
        assert not an and np.array_equal(lens, gold["em_wav_lens"][b0:b0 + 32])
        assert np.array_equal(ts, counts[b0:b0 + 32] < 640) and ts.sum() >= 4
    # a batch whose longest item is below min_num_samples: the Python returns an all-NaN block, the oracle raises the flag the C++ keeps (sd.cpp:2479)
    assert bool(gold["em_all_short_is_nan"][0])
    _, ts, an = orc.wav_lens(np.full(32, 546, np.int64))
    assert an and ts.all()


def test_mask_choice_against_reference_python(gold):
    """a6: per (chunk, local speaker) the clean mask -- frames where fewer than two speakers are active -- if it keeps MORE than
    ceil(293 * 640 / 80000) = 3 frames, else the full mask (mysegment.py:436-485 <-> sd.cpp:2430-2470), in the order the embedding batches are
    filled (chunk-major, speaker-minor); the chunk windows handed over with the masks are the crops of a7"""
    bz = gold["mc_binarized"].astype(np.float64)
    want = gold["mc_used_masks"]
    got = orc.select_masks(bz)
    assert np.array_equal(got, want)
    clean = bz * (bz.sum(axis=2, keepdims=True) < 2)
    cm = clean.transpose(0, 2, 1).reshape(-1, 293)
    fm = bz.transpose(0, 2, 1).reshape(-1, 293)
    took_clean = (want == cm).all(1) & (cm != fm).any(1)
    took_full = (want == fm).all(1) & (cm != fm).any(1)
    assert took_clean.sum() >= 10 and took_full.sum() >= 3            # both branches decided somewhere
    assert took_full[7 * 3 + 0] and cm[7 * 3 + 0].sum() == 3          # exactly 3 clean frames is not "more than 3"
    wav, _ = _em_case(gold)
    for i in range(0, len(want), 3):
        assert orc.crop(wav, (i // 3) * 8000).astype(np.float64).sum() == gold["mc_wave_sum"][i]


def test_chunk_crop_against_reference_python(gold):
    """the 5-second window of chunk k: samples floor(t * 16000) ... + 80 000, zeros past the end of the recording (mysegment.py:226-260, mode
    "pad") <-> the crop in front of a7 (sd.cpp:2567-2635); chunk starts k * 0.5 s and a few off-grid starts"""
    wav, _ = _em_case(gold)
    st = gold["crop_starts"]
    for i, t in enumerate(st):
        c = orc.crop(wav, int(np.floor(t * 16000)))
        assert np.array_equal(c[:8], gold["crop_head"][i]) and np.array_equal(c[-8:], gold["crop_tail"][i])
        assert c.astype(np.float64).sum() == gold["crop_sum"][i]
        z = int(80000 - np.max(np.nonzero(c)[0]) - 1) if c.any() else 80000
        assert z == gold["crop_zeros_at_end"][i]
    assert (gold["crop_zeros_at_end"] > 0).sum() >= 8 and (gold["crop_zeros_at_end"] == 0).sum() >= 8


def _stft_inputs(gold):
    wav, _ = _em_case(gold)
    return np.stack([wav[:80000], gold["stft_sine_pcm"].astype(np.float32) / np.float32(32768.0),
                     np.concatenate([wav[5000:9000], np.zeros(76000, np.float32)])]).astype(np.float32)


def test_stft_against_reference_python(gold):
    """a8: MySTFT(16000) = torch.stft(n_fft 400, hop 160, win 400, periodic Hamming, center, constant padding, onesided) -> transpose(2, 1):
    window, frame count, frame alignment at both edges (frames 0-3 and 497-500 see the zero padding), bin order, re / im layout"""
    x = _stft_inputs(gold)
    assert np.array_equal(torch.hamming_window(400).numpy(), gold["stft_window"])        # the window the oracle and the weight packs use
    y = nn.stft_ref(x).numpy()
    assert y.shape == (3, 501, 201, 2)
    fr = gold["stft_frames"]
    want = gold["stft_expected"]
    scale = np.abs(want).max(axis=(2, 3), keepdims=True)
    assert (np.abs(y[:, fr] - want) <= 2e-6 * scale + 1e-9).all()
    e = (y.astype(np.float64) ** 2).sum((2, 3))
    np.testing.assert_allclose(e, gold["stft_power_sum"], rtol=1e-5, atol=1e-12)


def test_sentence_mean_normalisation_against_reference_python(gold):
    """the last step of emd4.onnx's front end (threeModel.py:333-369): subtract the mean over the first round(len * T) frames, no variance"""
    x, lens, want = gold["norm_x"], gold["norm_lens"], gold["norm_expected"]
    got = nn.sentence_mean_norm(torch.from_numpy(x.copy()), lens).numpy()
    assert np.array_equal(got, want)
    n = [int(torch.round(torch.tensor(l) * 501).long()) for l in lens]
    assert n == [501, 351, 167, 25, 500, 251]


@pytest.mark.gpu
def test_hip_postseg_binarize_against_reference_python(diarizer, gold):
    s, want, tie = _pin_case(gold)
    nb, masks, count = diarizer.postseg(s)
    assert np.array_equal(nb[~tie], want[~tie]) and (nb[tie] == 0).all()


@pytest.mark.gpu
def test_hip_postseg_mask_choice_against_reference_python(diarizer, gold):
    """sd_postseg on scores that binarise to the reference's a6 patterns (0.9 where active, 0.1 where not): its masks are the reference's choice"""
    bz = gold["mc_binarized"]
    scores = np.where(bz > 0, np.float32(0.9), np.float32(0.1)).astype(np.float32)
    nb, masks, count = diarizer.postseg(scores)
    assert np.array_equal(nb, bz) and np.array_equal(masks, gold["mc_used_masks"])


@pytest.mark.gpu
def test_hip_frontend_lengths_against_reference_python(diarizer, gold):
    """sd_frontend on the reference's embedding_mask cases: the relative lengths it hands to the network are the reference's wav_lens, item
    by item and batch by batch (a too-short item gets 1.0); the features of an item depend on exactly the selected samples, so an item and
    a copy of it whose UNselected samples are replaced by noise give the same bits"""
    wav, masks = _em_case(gold)
    feats, lens = diarizer.frontend(wav[:(21 * 8000 + 80000)], masks)
    counts = gold["em_counts"]
    want = gold["em_wav_lens"]
    live = counts >= 640
    assert np.array_equal(lens[live], want[live])
    assert np.array_equal(lens[~live], np.ones((~live).sum(), np.float32)) and np.array_equal(want[~live], np.ones((~live).sum(), np.float32))
    # selection: poison every sample no mask of its chunk selects; the features of single-chunk items must not move
    per_frame = (np.arange(80000, dtype=np.int64) * 293) // 80000
    used = np.zeros(len(wav), bool)
    for i in range(64):
        sel = masks[i][per_frame] > 0.5
        used[(i // 3) * 8000:(i // 3) * 8000 + 80000] |= sel
    poisoned = wav.copy()
    rng = np.random.default_rng(1)
    # item 0 is a full mask (everything of chunk 0 is used); take the items of the LAST chunk, whose unselected samples beyond the previous
    # chunks' reach nobody else uses
    tail = slice(20 * 8000 + 72000, 21 * 8000 + 80000)
    unused_tail = ~used[tail]
    assert unused_tail.sum() > 1000
    poisoned[tail][unused_tail] = rng.standard_normal(int(unused_tail.sum())).astype(np.float32)
    assert not np.array_equal(poisoned, wav)
    f2, l2 = diarizer.frontend(poisoned[:(21 * 8000 + 80000)], masks)
    assert np.array_equal(l2, lens) and np.array_equal(f2, feats)
