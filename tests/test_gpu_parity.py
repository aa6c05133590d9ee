"""GPU parity tests: every stage of the HIP path, called through the C ABI, against the oracle.
Integer / index / fp64 stages must be bit-exact; the fp32 networks use the reference's own
tolerance rtol=1e-3, atol=1e-4 (pipeline/script/verifyEveryStepResult.py:119-124) and the
north-star bar "embedding cosine distance within 1e-3"."""
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import nn_oracle as nn
from oracle import orc, pipeline_oracle
import sdhip

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-3, 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tkey(t):
    return (round(t[0], 9), t[2])


def test_native_library_is_the_one_running(diarizer):
    import sdhip
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(sdhip.LIB_PATH) in maps


# ------------------------------------------------------------------ a2 + a3
@pytest.mark.parametrize("n", [80000 + 8000 * 3, 80000 + 8000 * 2 + 3000, 80000, 47011, 80001])
def test_segmentation_parity(diarizer, weights, n):
    rng = np.random.default_rng(n)
    wav = (0.1 * rng.standard_normal(n)).astype(np.float32) * (1 + np.sin(np.arange(n) / 2000.0)).astype(np.float32)
    seg = diarizer.segment(wav)
    nc, last = orc.num_chunks(n)
    assert seg.shape == (nc, 293, 3)
    net = nn.PyanNetOracle(weights[2])
    for i in range(nc):
        y = net(wav[None, i * 8000:i * 8000 + 80000]).numpy()[0]
        ref = np.zeros((293, 3), np.float32)
        ref[:y.shape[0]] = y                        # short last chunk is zero padded (sd.cpp:1473-1479)
        np.testing.assert_allclose(seg[i], ref, rtol=RTOL, atol=ATOL)


def test_segmentation_with_split_lstm_operands_holds_the_f32_bars(diarizer, weights):
    """option seg_precision = 3: PyanNet's LSTM -- the input projections of layers 1-3 (conv_gemm_h.hip's x3 form) and the recurrence
    (k_lstm_rec_x3) -- with both MFMA operands split into hi + lo fp16 halves.  Same bars as the f32 path against the torch oracle, scores
    within 2e-6 of the f32 path's (293 recurrent steps deep), through full chunks, a short last chunk, a loud stretch with a DC offset and
    a nearly silent one; leaving the mode restores the f32 bits; the 8 s recording that yields a single short chunk works too."""
    n = 80000 + 8000 * 5 + 3000
    rng = np.random.default_rng(91)
    wav = (0.05 * rng.standard_normal(n)).astype(np.float32) * (1 + np.sin(np.arange(n) / 2000.0)).astype(np.float32)
    wav[:30000] = wav[:30000] * 8 + 0.2
    wav[70000:95000] *= 1e-3
    s32 = diarizer.segment(wav)
    diarizer.set_option("seg_precision", 3)
    try:
        sx = diarizer.segment(wav)
        short = diarizer.segment(wav[:47011])
    finally:
        diarizer.set_option("seg_precision", -1)
    assert not np.array_equal(sx, s32) and np.abs(sx - s32).max() <= 2e-6, np.abs(sx - s32).max()
    nc, last = orc.num_chunks(n)
    net = nn.PyanNetOracle(weights[2])
    refs = np.zeros((nc, 293, 3), np.float32)
    for i in range(nc):
        y = net(wav[None, i * 8000:i * 8000 + 80000]).numpy()[0]
        refs[i, :y.shape[0]] = y                    # short last chunk is zero padded (sd.cpp:1473-1479)
    np.testing.assert_allclose(sx, refs, rtol=RTOL, atol=ATOL)
    # as close to the oracle as the f32 path is (both differ from it by summation order and by the gates' exp / rcp forms)
    assert np.abs(sx - refs).max() <= 2 * max(np.abs(s32 - refs).max(), 1e-6), (np.abs(sx - refs).max(), np.abs(s32 - refs).max())
    assert np.array_equal(diarizer.segment(wav), s32)
    assert short.shape == (1, 293, 3) and np.abs(short - diarizer.segment(wav[:47011])).max() <= 2e-6


def test_option_keys_are_validated(diarizer):
    """sd_set_option: an unknown key and an out-of-range precision are refused with SD_ERR_ARG and a message; the context stays usable"""
    import sdhip
    for key, val, frag in (("no_such_key", 1, "no_such_key"), ("ecapa_precision", 4, "ecapa_precision must be"), ("ecapa_precision", -1, "ecapa_precision must be"),
                           ("seg_precision", 1, "seg_precision must be"), ("seg_precision", 2, "seg_precision must be")):
        with pytest.raises(sdhip.SdError) as e:
            diarizer.set_option(key, val)
        assert e.value.code == 1 and frag in str(e.value), (key, val, str(e.value))
    for key, val in (("ecapa_precision", 3), ("ecapa_precision", 0), ("seg_precision", 3), ("seg_precision", 0), ("seg_precision", -1)):
        diarizer.set_option(key, val)


def test_segmentation_precision_follows_the_embedding_mode_when_left_at_auto(diarizer, golden_dir):
    """seg_precision = -1 (the default since round 6): PyanNet's LSTM takes the split-operand form whenever an fp16-pipe mode is selected for ECAPA
    (ecapa_precision 1, 2 or 3) and the f32 MFMA otherwise -- bit for bit the scores of the explicit settings -- and the recording of BASELINE configs[0]
    (multi-speaker_1min.wav) comes out with the same turns in fp16 mode whichever form the segmentation ran in."""
    import sdhip
    rng = np.random.default_rng(92)
    n = 80000 + 8000 * 7 + 1234
    wav = (0.05 * rng.standard_normal(n)).astype(np.float32) * (1 + np.sin(np.arange(n) / 1500.0)).astype(np.float32)
    try:
        diarizer.set_option("seg_precision", 0); s_f32 = diarizer.segment(wav)
        diarizer.set_option("seg_precision", 3); s_x3 = diarizer.segment(wav)
        diarizer.set_option("seg_precision", -1)
        assert np.array_equal(diarizer.segment(wav), s_f32)
        for mode in (1, 2, 3):
            diarizer.set_option("ecapa_precision", mode)
            assert np.array_equal(diarizer.segment(wav), s_x3), mode
        diarizer.set_option("ecapa_precision", 0)
        assert np.array_equal(diarizer.segment(wav), s_f32)
        assert not np.array_equal(s_f32, s_x3) and np.abs(s_f32 - s_x3).max() <= 2e-6
        pcm = sdhip.read_wav(os.path.join(golden_dir, "multi-speaker_1min.wav"))[0]
        diarizer.set_option("ecapa_precision", 1)
        t_auto = diarizer.diarize(pcm)
        diarizer.set_option("seg_precision", 0)
        t_f32seg = diarizer.diarize(pcm)
        assert len(t_auto) > 0 and t_auto == t_f32seg
    finally:
        diarizer.set_option("seg_precision", -1)
        diarizer.set_option("ecapa_precision", 0)


def test_segmentation_shared_conv0_equals_per_chunk_conv0(diarizer, weights):
    """SincNet conv0 applied once to the raw waveform + the chunk normalisation as an affine map (pyannet.hip, k_chunk_stats)
    against one conv per normalised chunk: same scores to rounding, also with a DC offset, a loud and a nearly silent stretch
    and a short last chunk; both within the oracle tolerance."""
    n = 80000 + 8000 * 4 + 5000
    rng = np.random.default_rng(77)
    wav = (0.05 * rng.standard_normal(n)).astype(np.float32)
    wav[:40000] = wav[:40000] * 8 + 0.3                 # loud, with a DC offset
    wav[60000:90000] *= 1e-3                            # nearly silent
    diarizer.set_option("seg_shared_conv0", 0)
    per_chunk = diarizer.segment(wav)
    diarizer.set_option("seg_shared_conv0", 1)
    shared = diarizer.segment(wav)
    assert np.abs(shared - per_chunk).max() <= 2e-5
    nc, last = orc.num_chunks(n)
    net = nn.PyanNetOracle(weights[2])
    for i in range(nc):
        y = net(wav[None, i * 8000:i * 8000 + 80000]).numpy()[0]
        ref = np.zeros((293, 3), np.float32)
        ref[:y.shape[0]] = y
        np.testing.assert_allclose(shared[i], ref, rtol=RTOL, atol=ATOL)


def test_segmentation_shared_conv0_with_a_dc_offset_on_near_silence(diarizer, weights):
    """the hard case for the shared first convolution: a recording with a DC offset whose chunks are nearly silent -- the chunk
    normalisation then multiplies by rstd up to 316 and the DC term of conv0(x) has to cancel against mean * sum(W).  Scores of the
    shared form against the per-chunk form and against the oracle, and the binarised decisions (hysteresis at 0.4442) of the two forms"""
    n = 80000 + 8000 * 6
    rng = np.random.default_rng(78)
    wav = (2e-4 * rng.standard_normal(n) + 0.05).astype(np.float32)             # DC / sigma = 250
    wav[30000:50000] += (0.02 * rng.standard_normal(20000)).astype(np.float32)     # a burst, so that the chunks differ
    wav[100000:] = (1e-3 * rng.standard_normal(n - 100000) - 0.2).astype(np.float32)   # another offset, DC / sigma = 200
    diarizer.set_option("seg_shared_conv0", 0)
    per_chunk = diarizer.segment(wav)
    diarizer.set_option("seg_shared_conv0", 1)
    shared = diarizer.segment(wav)
    nc, last = orc.num_chunks(n)
    net = nn.PyanNetOracle(weights[2])
    ref = np.stack([net(wav[None, i * 8000:i * 8000 + 80000]).numpy()[0] for i in range(nc)])
    print("shared vs per-chunk %.2e, shared vs oracle %.2e, per-chunk vs oracle %.2e" % (np.abs(shared - per_chunk).max(), np.abs(shared - ref).max(), np.abs(per_chunk - ref).max()))
    np.testing.assert_allclose(shared, ref, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(per_chunk, ref, rtol=RTOL, atol=ATOL)
    assert np.abs(shared - per_chunk).max() <= 5e-5


def test_segmentation_too_short_for_one_frame_is_zero(diarizer):
    # a (single) chunk too short for one output frame: frames are zero padded like sd.cpp:1473-1479
    wav = np.random.default_rng(0).standard_normal(200).astype(np.float32) * 0.1
    seg = diarizer.segment(wav)
    assert seg.shape == (1, 293, 3) and not seg.any()


# ------------------------------------------------------------------ a4 - a6
def test_postseg_bit_exact(diarizer):
    rng = np.random.default_rng(1)
    c = 150
    sc = rng.random((c, 293, 3)).astype(np.float32)
    sc[:, :, 2] *= 0.5
    sc[5] = 0.1                                   # inactive chunk
    sc[6, :, 0] = 0.9; sc[6, :, 1] = 0.9          # permanent overlap: clean masks empty
    sc[7, :, :] = 0.9                             # 3 speakers always on
    sc[8, :5, 1] = 0.9; sc[8, 5:, 1] = 0.1        # 5 frames only
    sc[9] = np.float32(orc.ONSET)                 # nearest float to the threshold
    b, m, cnt = diarizer.postseg(sc)
    b_ref = orc.binarize(sc)
    assert np.array_equal(b.astype(np.float64), b_ref)
    assert np.array_equal(m, orc.select_masks(b_ref))
    cnt_ref, _, _ = orc.speaker_count(b_ref)
    assert np.array_equal(cnt, cnt_ref)


# ------------------------------------------------------------------ a7 - a9
def _wav_and_masks(rng, items):
    n = 8000 * ((items + 2) // 3 - 1) + 80000
    wav = (0.2 * rng.standard_normal(n)).astype(np.float32) * (1 + np.sin(np.arange(n) / 3000.0)).astype(np.float32)
    masks = (rng.random((items, 293)) > 0.4).astype(np.float32)
    return wav, masks


def _oracle_signals(wav, masks):
    items = masks.shape[0]
    sigs = np.zeros((items, 80000), np.float32)
    cnts = np.zeros(items, np.int64)
    for i in range(items):
        sigs[i], cnts[i] = orc.mask_compact(orc.crop(wav, (i // 3) * 8000), masks[i])
    lens = np.zeros(items, np.float32)
    bad = np.zeros(items, bool)
    for b0 in range(0, items, 32):
        l, ts, an = orc.wav_lens(cnts[b0:b0 + 32])
        lens[b0:b0 + 32] = l
        bad[b0:b0 + 32] = ts | an
    return sigs, lens, bad


def test_frontend_parity(diarizer, weights):
    rng = np.random.default_rng(2)
    wav, masks = _wav_and_masks(rng, 70)
    masks[3] = 0; masks[3, :2] = 1                 # 546 samples < 640: too short
    masks[7] = 1                                   # everything selected
    masks[9] = 0                                   # nothing selected
    masks[11, ::2] = 0; masks[11, 1::2] = 1        # alternating frames: many short runs
    f_gpu, l_gpu = diarizer.frontend(wav, masks)
    sigs, lens, bad = _oracle_signals(wav, masks)
    assert np.array_equal(l_gpu, lens)             # wav_lens is "same file content" in the reference harness
    st = nn.stft_ref(sigs, weights[3]["stft.window"])
    f_ref = nn.fbank_norm_ref(st, lens, weights[3]["fbank.matrix"]).numpy()
    np.testing.assert_allclose(f_gpu[~bad], f_ref[~bad], rtol=RTOL, atol=ATOL)


def test_ecapa_parity(diarizer, weights):
    rng = np.random.default_rng(3)
    feats = (3.0 * rng.standard_normal((6, 501, 80))).astype(np.float32)
    lens = np.array([1.0, 0.7311, 0.25, 0.5, 0.9991, 0.008], np.float32)
    e_gpu = diarizer.ecapa(feats, lens)
    e_ref = nn.EcapaOracle(weights[3])(feats, lens).numpy()
    np.testing.assert_allclose(e_gpu, e_ref, rtol=RTOL, atol=ATOL)


def test_ecapa_dead_row_skipping_is_invisible(diarizer, weights):
    """row panels beyond nvalid + 65 frames (the network's one-sided receptive field) are not computed
    (DESIGN.md section 2): short items must give the same embedding with and without the skip, and a poisoned
    workspace (NaN left behind by a previous batch) must not leak into them"""
    rng = np.random.default_rng(13)
    lens = np.array([1.0, 0.008, 0.25, 0.5, 0.12, 0.37, 0.62, 0.87, 0.999, 0.3], np.float32)
    feats = (3.0 * rng.standard_normal((len(lens), 501, 80))).astype(np.float32)
    poison = np.full((len(lens), 501, 80), np.nan, np.float32)
    diarizer.set_option("skip_dead_rows", 0)
    diarizer.ecapa(poison, np.ones(len(lens), np.float32))           # fill every activation buffer with NaN
    e_full = diarizer.ecapa(feats, lens)
    diarizer.set_option("skip_dead_rows", 1)
    diarizer.ecapa(poison, np.ones(len(lens), np.float32))
    e_skip = diarizer.ecapa(feats, lens)
    assert np.isfinite(e_skip).all()
    assert np.array_equal(e_full, e_skip)
    e_ref = nn.EcapaOracle(weights[3])(feats, lens).numpy()
    np.testing.assert_allclose(e_skip, e_ref, rtol=RTOL, atol=ATOL)


def test_ecapa_fp16_mfma_within_reference_tolerance(diarizer, weights):
    """BASELINE.json configs[4]: ECAPA conv layers on the fp16 MFMA (fp16 weights, activations rounded to fp16 on the
    way into LDS, f32 accumulation) against the f32 oracle.  Bar: embedding cosine distance <= 1e-3 (north star /
    verifyEveryStepResult.py); the element-wise tolerance of the f32 path does not apply to a 10-bit mantissa."""
    rng = np.random.default_rng(21)
    lens = np.array([1.0, 0.5, 0.25, 0.9, 0.7, 0.33, 1.0, 0.6], np.float32)
    feats = (3.0 * rng.standard_normal((len(lens), 501, 80))).astype(np.float32)
    e32 = diarizer.ecapa(feats, lens)
    diarizer.set_option("ecapa_precision", 1)
    try:
        e16 = diarizer.ecapa(feats, lens)
    finally:
        diarizer.set_option("ecapa_precision", 0)
    e_ref = nn.EcapaOracle(weights[3])(feats, lens).numpy().astype(np.float64)
    assert np.isfinite(e16).all() and not np.array_equal(e16, e32)      # the fp16 kernels really ran
    g = e16.astype(np.float64)
    cos = (g * e_ref).sum(1) / np.linalg.norm(g, axis=1) / np.linalg.norm(e_ref, axis=1)
    assert (1 - cos).max() < 1e-3, (1 - cos).max()
    rel = np.linalg.norm(g - e_ref, axis=1) / np.linalg.norm(e_ref, axis=1)
    assert rel.max() < 2e-2, rel.max()
    assert np.array_equal(diarizer.ecapa(feats, lens), e32)                 # and the f32 path is back, bit for bit


def test_ecapa_x3_split_operands_on_the_fp16_mfma_hold_the_f32_bars(diarizer, weights):
    """option ecapa_precision = 3: f32 tensors, every ECAPA conv layer with both operands split into hi + lo fp16 halves on
    v_mfma_f32_32x32x16_f16 (conv_gemm_h.hip P = 3, conv_gemm.hip PR = 3).  22-bit operands, f32 accumulation: the mode has to pass the bars
    of the F32 path -- element-wise RTOL / ATOL against the torch oracle and a cosine distance three orders below the north-star 1e-3 --
    on full and partial lengths, through the wide tile (block0, TDNN, MFA, attention logits) and the 128 x 128 tile (Res2Net with its
    second input, the attention's hidden layer with per-item bias and tanh); the result does not depend on the batch it is computed in
    (the wide tile takes every batch size in this mode), and leaving the mode restores the f32 bits."""
    rng = np.random.default_rng(31)
    lens = np.array([1.0, 0.5, 0.25, 0.9, 0.7, 0.33, 1.0, 0.6, 0.8, 0.45, 0.05, 1.0], np.float32)
    feats = (3.0 * rng.standard_normal((len(lens), 501, 80))).astype(np.float32)
    feats[3] *= 30.0                                       # loud item: activations of a few thousand stay far inside fp16's range
    feats[5] *= 1e-3                                       # quiet item: the lo halves of small activations are fp16 subnormals
    e32 = diarizer.ecapa(feats, lens)
    diarizer.set_option("ecapa_precision", 3)
    try:
        ex = diarizer.ecapa(feats, lens)
        ex_one = np.concatenate([diarizer.ecapa(feats[i:i + 1], lens[i:i + 1]) for i in (0, 2, 10)])
        ex_pair = diarizer.ecapa(feats[4:9], lens[4:9])
    finally:
        diarizer.set_option("ecapa_precision", 0)
    assert np.isfinite(ex).all() and not np.array_equal(ex, e32)                  # the split kernels really ran
    assert np.array_equal(ex_one, ex[[0, 2, 10]]) and np.array_equal(ex_pair, ex[4:9])
    e_ref = nn.EcapaOracle(weights[3])(feats, lens).numpy()
    np.testing.assert_allclose(ex, e_ref, rtol=RTOL, atol=ATOL * np.abs(e_ref).max())
    g, r = ex.astype(np.float64), e_ref.astype(np.float64)
    cd = 1 - (g * r).sum(1) / np.linalg.norm(g, axis=1) / np.linalg.norm(r, axis=1)
    g32 = e32.astype(np.float64)
    cd32 = 1 - (g32 * r).sum(1) / np.linalg.norm(g32, axis=1) / np.linalg.norm(r, axis=1)
    assert cd.max() < 1e-6 and cd32.max() < 1e-6, (cd.max(), cd32.max())
    # as close to the oracle as the f32 MFMA path is (both differ from it by summation order; the split adds 2^-22 per product)
    rel = np.linalg.norm(g - r, axis=1) / np.linalg.norm(r, axis=1)
    rel32 = np.linalg.norm(g32 - r, axis=1) / np.linalg.norm(r, axis=1)
    assert rel.max() < 3 * max(rel32.max(), 1e-5), (rel.max(), rel32.max())
    assert np.array_equal(diarizer.ecapa(feats, lens), e32)
    # overflow guard: activations beyond fp16's range have no hi half (Inf -> NaN embeddings); the mode notices and repeats the call on the
    # f32 kernels -- the f32 path's bits, one fallback counted, and the mode stays selected for the next call
    big = feats.copy()
    big[1] *= 3.0e4
    e32_big = diarizer.ecapa(big, lens)
    assert np.isfinite(e32_big).all()
    diarizer.set_option("ecapa_precision", 3)
    try:
        diarizer.reset_stats()
        ex_big = diarizer.ecapa(big, lens)
        assert diarizer.kernel_stats("x3_overflow_fallbacks")["launches"] == 1
        assert np.array_equal(ex_big, e32_big)
        again = diarizer.ecapa(feats, lens)
        assert np.array_equal(again, ex) and diarizer.kernel_stats("x3_overflow_fallbacks")["launches"] == 1
    finally:
        diarizer.set_option("ecapa_precision", 0)


def test_ecapa_x3_holds_for_tiny_and_huge_weights(weights, tmp_path):
    """x3 mode scales every layer's weights by a power of two before the split, so that their lo halves are normal fp16 numbers whatever the
    layer's magnitude (weights.cpp).  Packs whose MFA / tdnn1 / Res2Net / block0 / tdnn2 layers are rescaled by 3e-5 ... 2e4 (conv weights and
    bias by s, the BatchNorm behind them by s and s^2, so that the activations keep their size) are valid models of their own: on each of them
    the x3 embeddings must sit on the f32 path's, as they do for the original pack.  (Without the per-layer scale the lo halves of weights of
    1e-6 would all be zero: fp16-mode accuracy.)"""
    import sdhip
    we = weights[3]
    rng = np.random.default_rng(41)
    lens = np.array([1.0, 0.6, 0.3, 0.85], np.float32)
    feats = (3.0 * rng.standard_normal((len(lens), 501, 80))).astype(np.float32)

    def scaled(pack, layer, s):
        w = dict(pack)
        w[layer + ".conv.weight"] = (pack[layer + ".conv.weight"].astype(np.float64) * s).astype(np.float32)
        w[layer + ".conv.bias"] = (pack[layer + ".conv.bias"].astype(np.float64) * s).astype(np.float32)
        w[layer + ".norm.running_mean"] = (pack[layer + ".norm.running_mean"].astype(np.float64) * s).astype(np.float32)
        w[layer + ".norm.running_var"] = np.maximum(pack[layer + ".norm.running_var"].astype(np.float64) * s * s, 1e-30).astype(np.float32)
        return w

    def run(pack, mode, k=[0]):
        ep = str(tmp_path / ("e%d.sdw" % k[0])); k[0] += 1
        nn.save_pack(ep, pack)
        d = sdhip.Diarizer(weights[0], ep)
        try:
            d.set_option("ecapa_precision", mode)
            return d.ecapa(feats, lens).astype(np.float64)
        finally:
            d.close()

    def cosd(a, b):
        return 1 - (a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)

    packs = {"original": we,
             "large": scaled(scaled(scaled(we, "mfa", 1.9e4), "blocks.1.tdnn1", 2.0 ** 14), "blocks.2.res2net.3", 3.0e2),
             "small": scaled(scaled(scaled(scaled(we, "mfa", 2.0e-2), "blocks.1.tdnn1", 6.0e-3), "blocks.3.tdnn2", 7.3e-3), "blocks.0", 1.5e-2),
             # weights of 1e-6: BatchNorm's eps dominates the variance, so the layer's output shrinks to ~1 % -- a different but valid model
             "tiny": scaled(we, "mfa", 3.1e-5)}
    for name, p in packs.items():
        ex, ef = run(p, 3), run(p, 0)
        assert np.isfinite(ex).all() and np.isfinite(ef).all() and not np.array_equal(ex, ef), name
        assert cosd(ex, ef).max() < 1e-6, (name, cosd(ex, ef).max())
        rel = np.linalg.norm(ex - ef, axis=1) / np.linalg.norm(ef, axis=1)
        assert rel.max() < 1e-3, (name, rel.max())


def test_ecapa_fp16_wide_tile_kernel_gives_the_same_bits(diarizer):
    """fp16 mode: the 256 x 256 kernel of the wide layers (conv_gemm_h.hip) and the 128 x 128 kernel feed the same k-blocks to the
    same MFMA in the same order: embeddings must be bit-identical, whichever kernel a batch size selects"""
    rng = np.random.default_rng(23)
    lens = np.array([1.0, 0.5, 0.25, 0.9, 0.7, 0.33, 1.0, 0.6, 0.8, 0.45], np.float32)
    feats = (3.0 * rng.standard_normal((len(lens), 501, 80))).astype(np.float32)
    diarizer.set_option("ecapa_precision", 1)
    try:
        diarizer.set_option("conv_h256", 1)
        e_wide = diarizer.ecapa(feats, lens)
        diarizer.set_option("conv_h256", 0)
        e_128 = diarizer.ecapa(feats, lens)
    finally:
        diarizer.set_option("conv_h256", 1)
        diarizer.set_option("ecapa_precision", 0)
    assert np.isfinite(e_wide).all() and np.array_equal(e_wide, e_128)


CONV_ROT_DEFAULT = 3


def test_ecapa_fp16_lds_dma_staged_kernel_gives_the_same_bits(diarizer):
    """fp16 mode: the LDS-DMA staged 256 x 256 kernel (conv_gemm_g.hip: buffer_load ... lds into a swizzled, unpadded LDS image, two
    K-steps in flight) in its 32x32x16 form (conv_mfma16 = 0) against the register-staged one (conv_gemm_h.hip): same k-blocks into the
    same MFMA chain in the same order -- bit-identical embeddings on ragged items (row tables, reflect padding, clamped taps, partly
    filled last tiles), run twice (a DMA that lands late would show as a run-to-run difference).  The default form runs the same stages
    through v_mfma_f32_16x16x32_f16 (32 k per instruction instead of 16: the f32 accumulation is grouped differently, so its bits are its
    own): bit-identical to itself across runs and across the request orders of conv_rot, and within rounding of the 32x32x16 form."""
    rng = np.random.default_rng(31)
    lens = np.array([1.0, 0.5, 0.25, 0.9, 0.7, 0.33, 1.0, 0.6, 0.8, 0.45, 0.12, 1.0, 0.77, 0.05, 0.95, 0.5, 0.61, 1.0, 0.29, 0.83], np.float32)
    feats = (3.0 * rng.standard_normal((len(lens), 501, 80))).astype(np.float32)
    diarizer.set_option("ecapa_precision", 1)
    try:
        diarizer.set_option("conv_glds", 1)
        diarizer.set_option("conv_mfma16", 0)
        e_dma = diarizer.ecapa(feats, lens)
        e_dma2 = diarizer.ecapa(feats, lens)
        e_rot = {0: [], 1: []}
        for m16 in (0, 1):
            diarizer.set_option("conv_mfma16", m16)
            for rot in (0, 1, 3):          # request order of a panel's quarters / which rows a wave's pieces cover: the same LDS image
                diarizer.set_option("conv_rot", rot)
                e_rot[m16] += [diarizer.ecapa(feats, lens), diarizer.ecapa(feats, lens)]
        diarizer.set_option("conv_glds", 0)
        e_reg = diarizer.ecapa(feats, lens)
    finally:
        diarizer.set_option("conv_glds", 1)
        diarizer.set_option("conv_mfma16", 1)
        diarizer.set_option("conv_rot", CONV_ROT_DEFAULT)
        diarizer.set_option("ecapa_precision", 0)
    assert np.isfinite(e_dma).all() and np.array_equal(e_dma, e_reg) and np.array_equal(e_dma, e_dma2)
    for e in e_rot[0]:
        assert np.array_equal(e, e_reg)
    e16 = e_rot[1][0]
    for e in e_rot[1]:
        assert np.array_equal(e, e16)
    assert np.isfinite(e16).all()
    cosd = 1.0 - (e16 * e_reg).sum(1) / np.sqrt((e16 * e16).sum(1) * (e_reg * e_reg).sum(1))
    print("16x16x32 against 32x32x16: max cosine distance %.2e, max |diff| / max |e| %.2e" % (cosd.max(), np.abs(e16 - e_reg).max() / np.abs(e_reg).max()))
    assert cosd.max() < 1e-5 and np.abs(e16 - e_reg).max() < 2e-3 * np.abs(e_reg).max()


def test_ecapa_fp16_ping_pong_kernel_gives_the_same_bits(diarizer):
    """fp16 mode: the round-6 kernel of the wide layers (conv_gemm_p.hip: a load stream that is never drained, every wave interleaves its
    fragment reads and LDS-DMAs with its MFMAs, weights as the MFMA's first operand, 16-byte stores straight from the accumulators, the epilogue
    in chunks in front of the next tile's first MFMAs) against the kernel it replaces (conv_gemm_g.hip, 16x16x32 form): the same products in the
    same order, so the embeddings must be bit-identical -- on ragged items (row tables with cross-space layers, clamped frames, a partly filled
    last tile, several tiles per workgroup), run three times (a half-tile read before its DMA has landed, or overwritten before its last read,
    or a store whose data register was rewritten too early, shows as a difference)."""
    rng = np.random.default_rng(37)
    lens = np.array([1.0, 0.5, 0.25, 0.9, 0.7, 0.33, 1.0, 0.6, 0.8, 0.45, 0.12, 1.0, 0.77, 0.05, 0.95, 0.5, 0.61, 1.0, 0.29, 0.83] * 3, np.float32)
    feats = (3.0 * rng.standard_normal((len(lens), 501, 80))).astype(np.float32)
    diarizer.set_option("ecapa_precision", 1)
    try:
        diarizer.set_option("conv_pp", 1)
        e_pp = [diarizer.ecapa(feats, lens) for _ in range(3)]
        diarizer.set_option("conv_pp", 0)
        diarizer.set_option("conv_mfma16", 2)          # the 16x16x32 form on block0's short contraction too, as the round-6 kernel runs it
        e_g = diarizer.ecapa(feats, lens)
    finally:
        diarizer.set_option("conv_pp", 1)
        diarizer.set_option("conv_mfma16", 1)
        diarizer.set_option("ecapa_precision", 0)
    assert np.isfinite(e_g).all()
    for e in e_pp:
        assert np.isfinite(e).all()
        bad = np.flatnonzero((e != e_g).any(axis=1))
        assert bad.size == 0, ("items that differ", bad[:10], np.abs(e - e_g).max())


@pytest.mark.parametrize("n_items,seed", [(7, 1), (9, 2), (23, 3), (131, 4)])
def test_ecapa_fp16_wide_kernel_same_bits_over_batch_geometries(diarizer, n_items, seed):
    """the same comparison over batch geometries the hour does not produce: just above the kernel's 2 048-row minimum (9 - 10 row panels: one or two per
    XCD, most workgroups of the grid without a tile), a handful of panels per XCD with a ragged last one, and ~ 170 panels (several super-block rounds per
    workgroup, the last round partly filled).  Two runs each."""
    rng = np.random.default_rng(100 + seed)
    lens = rng.choice(np.array([1.0, 0.97, 0.5, 0.26, 0.81, 0.07, 0.64], np.float32), size=n_items).astype(np.float32)
    lens[:5] = 1.0          # >= 2 500 rows: the kernel takes the layers (M >= 2 048)
    feats = (3.0 * rng.standard_normal((n_items, 501, 80))).astype(np.float32)
    diarizer.set_option("ecapa_precision", 1)
    try:
        diarizer.set_option("conv_pp", 1)
        e_pp = [diarizer.ecapa(feats, lens) for _ in range(2)]
        diarizer.set_option("conv_pp", 0)
        diarizer.set_option("conv_mfma16", 2)
        e_g = diarizer.ecapa(feats, lens)
    finally:
        diarizer.set_option("conv_pp", 1)
        diarizer.set_option("conv_mfma16", 1)
        diarizer.set_option("ecapa_precision", 0)
    assert np.isfinite(e_g).all()
    for e in e_pp:
        assert np.array_equal(e, e_g), ("items that differ", np.flatnonzero((e != e_g).any(axis=1))[:10], np.abs(e - e_g).max())


def test_ecapa_bits_do_not_depend_on_how_many_items_share_a_batch(diarizer):
    """2 100 short items: one batch under the default row budget (more than 2 048 items: the per-utterance layers -- SE, ASP bias, fc -- must
    still take the kernel they take in small batches) against batches of 96: bit-identical embeddings in f32, x3 and fp16 mode.  (Round 4:
    above 2 048 items per batch those layers fell to the 128 x 128 kernel, whose K order differs in the last bits.)"""
    rng = np.random.default_rng(41)
    n = 2100
    lens = np.full(n, 0.06, np.float32)
    lens[::7] = 0.2
    feats = (3.0 * rng.standard_normal((n, 501, 80))).astype(np.float32)
    try:
        for mode in (0, 3, 1):
            diarizer.set_option("ecapa_precision", mode)
            diarizer.set_option("emb_batch_items", 96)
            e_small = diarizer.ecapa(feats, lens)
            diarizer.set_option("emb_batch_items", 3072)
            e_big = diarizer.ecapa(feats, lens)
            assert np.isfinite(e_big).all() and np.array_equal(e_big, e_small), mode
    finally:
        diarizer.set_option("emb_batch_items", 3072)
        diarizer.set_option("ecapa_precision", 0)


def test_ecapa_f32_lds_dma_staged_kernel_gives_the_same_bits(diarizer):
    """f32: the LDS-DMA staged form of the wide tile (conv_gemm_g.hip P = 0: weights as the MFMA's first operand -> transposed accumulators,
    16-byte output stores that drain under the next tile, parameters through LDS) against the default register-staged kernel: bit-identical
    embeddings, twice.  (It is not the default: 5 % slower on the planted hour.)"""
    rng = np.random.default_rng(37)
    lens = np.array([1.0, 0.5, 0.25, 0.9, 0.7, 0.33, 1.0, 0.6, 0.8, 0.45, 0.12, 1.0, 0.77, 0.05, 0.95, 0.5, 0.61, 1.0, 0.29, 0.83], np.float32)
    feats = (3.0 * rng.standard_normal((len(lens), 501, 80))).astype(np.float32)
    e_reg = diarizer.ecapa(feats, lens)
    diarizer.set_option("conv_glds_f32", 1)
    try:
        e_dma = diarizer.ecapa(feats, lens)
        e_dma2 = diarizer.ecapa(feats, lens)
    finally:
        diarizer.set_option("conv_glds_f32", 0)
    assert np.isfinite(e_dma).all() and np.array_equal(e_dma, e_reg) and np.array_equal(e_dma, e_dma2)


def test_ecapa_f32_wide_tile_kernel_gives_the_same_bits(diarizer):
    """f32: the 256 x 256 kernel (TDNN, MFA) sums K in conv_gemm.hip's order: bit-identical embeddings"""
    rng = np.random.default_rng(29)
    lens = np.array([1.0, 0.5, 0.25, 0.9, 0.7, 0.33, 1.0, 0.6, 0.8, 0.45], np.float32)
    feats = (3.0 * rng.standard_normal((len(lens), 501, 80))).astype(np.float32)
    try:
        diarizer.set_option("conv_w256_f32", 1)
        e_wide = diarizer.ecapa(feats, lens)
        diarizer.set_option("conv_w256_f32", 0)
        e_128 = diarizer.ecapa(feats, lens)
    finally:
        diarizer.set_option("conv_w256_f32", 1)
    assert np.isfinite(e_wide).all() and np.array_equal(e_wide, e_128)


def test_embed_parity(diarizer, weights):
    rng = np.random.default_rng(4)
    wav, masks = _wav_and_masks(rng, 100)
    masks[3] = 0; masks[3, :2] = 1
    masks[64:96] = 0; masks[64:96, :1] = 1         # a whole reference batch below min_num_samples -> all NaN (sd.cpp:2479)
    e_gpu = diarizer.embed(wav, masks)
    sigs, lens, bad = _oracle_signals(wav, masks)
    assert bad[64:96].all() and bad[3]
    assert np.array_equal(np.isnan(e_gpu[:, 0]), bad)
    st = nn.stft_ref(sigs, weights[3]["stft.window"])
    ok = ~bad
    f32 = nn.fbank_norm_ref(st, lens, weights[3]["fbank.matrix"])
    e_ref = nn.EcapaOracle(weights[3])(f32, lens).numpy()[ok]
    f64 = nn.fbank_norm_ref(st, lens, weights[3]["fbank.matrix"], torch.float64)
    e_ref64 = nn.EcapaOracle(weights[3], torch.float64)(f64, lens).numpy()[ok]
    g = e_gpu[ok].astype(np.float64)
    # north-star bar: cosine distance to the reference embedding within 1e-3
    cos = (g * e_ref).sum(1) / np.linalg.norm(g, axis=1) / np.linalg.norm(e_ref, axis=1)
    assert (1 - cos).max() < 1e-3
    # reference tolerance, atol scaled to the embedding magnitude (the synthetic net's outputs are O(100))
    np.testing.assert_allclose(g, e_ref, rtol=RTOL, atol=ATOL * np.abs(e_ref).max())
    # and no less accurate than the fp32 oracle itself is w.r.t. an fp64 evaluation
    assert np.abs(g - e_ref64).max() <= 8 * np.abs(e_ref - e_ref64).max() + 1e-6


# ------------------------------------------------------------------ a12 - a14
def _blobs(rng, N, d=192, k=4, s=0.6):
    cen = rng.standard_normal((k, d))
    X = cen[rng.integers(0, k, N)] + s * rng.standard_normal((N, d))
    return X / np.linalg.norm(X, axis=1, keepdims=True)


@pytest.mark.parametrize("N,d", [(2, 192), (3, 192), (12, 2), (65, 192), (300, 192), (2000, 192), (700, 7)])
def test_linkage_bit_exact(diarizer, N, d):
    rng = np.random.default_rng(N + d)
    X = _blobs(rng, N, d) if d > 2 else np.array([[0, 0], [0, 1], [1, 0], [0, 4], [0, 3], [1, 4], [4, 0], [3, 0], [4, 1], [4, 4], [3, 4], [4, 3]], float)
    cutoff = orc.THRESH_F32 if d > 2 else 1.1
    T_ref, Z_ref = orc.ahc(X, cutoff)
    Z = diarizer.linkage(X)
    # bit-identical dendrogram -- also on the 12-point grid of cluster.cpp:8-13, which is all exact ties (unit distances):
    # the merge order among equal heights is the reference heap's (k_linkage_heap)
    assert np.array_equal(Z, Z_ref)
    assert np.array_equal(diarizer.cluster(X, cutoff), T_ref)


@pytest.mark.parametrize("N,G,T", [(3, 2, 256), (65, 7, 512), (300, 64, 256), (300, 200, 1024), (2000, 16, 256), (2000, 32, 256), (5000, 64, 512), (5000, 2, 256), (3000, 3, 128)])
def test_cooperative_linkage_bit_exact(diarizer, N, G, T):
    """the cooperative kernels (k_linkage_rg: per-column state in registers, 4 or -- (5000, 2, 256), (3000, 3, 128) -- 8 columns per thread;
    k_linkage_mw where the geometry does not fit it, e.g. 200 workgroups) give the oracle's dendrogram bit for bit for any workgroup
    count / size"""
    rng = np.random.default_rng(1000 + N + G)
    X = _blobs(rng, N)
    _, Z_ref = orc.ahc(X, orc.THRESH_F32)
    diarizer.set_option("linkage_wgs", G)
    diarizer.set_option("linkage_threads", T)
    try:
        Z = diarizer.linkage(X)
    finally:
        diarizer.set_option("linkage_wgs", -1)
        diarizer.set_option("linkage_threads", 0)
    assert np.array_equal(Z, Z_ref)


def _tie_sets():
    rng = np.random.default_rng(9)
    X = _blobs(rng, 200)
    X[7] = X[3]; X[19] = X[3]; X[30] = X[11]; X[150] = X[149]               # duplicate embeddings (looped audio, digital silence)
    yield "duplicates", X
    g = np.stack(np.meshgrid(np.arange(8.0), np.arange(8.0), np.arange(8.0)), -1).reshape(-1, 3)   # 512 lattice points: almost every merge is a tie
    yield "lattice", g[rng.permutation(len(g))]
    Y = _blobs(rng, 2400)
    Y[rng.integers(0, 2400, 300)] = Y[rng.integers(0, 2400, 300)]            # ~300 duplicated rows in a set large enough for the cooperative kernel
    yield "big_duplicates", Y


@pytest.mark.parametrize("G", [-1, 0, 16])
def test_linkage_with_exact_ties_is_bit_identical(diarizer, G):
    """exact ties are resolved by the reference's heap (clustering.cpp:28-119, 323-404), not by value: Z must still be
    array_equal.  G = 0 forces k_linkage_heap (heap in global memory above 2048 rows), G = 16 forces the cooperative kernel,
    which has to notice the first tie and hand the job over."""
    for name, X in _tie_sets():
        T_ref, Z_ref = orc.ahc(X, orc.THRESH_F32)
        fb0 = diarizer.kernel_stats("linkage_tie_fallbacks")["launches"]
        diarizer.set_option("linkage_wgs", G)
        try:
            Z = diarizer.linkage(X)
            T = diarizer.cluster(X, orc.THRESH_F32)
        finally:
            diarizer.set_option("linkage_wgs", -1)
        assert np.array_equal(Z, Z_ref), (name, G)
        assert np.array_equal(T, T_ref), (name, G)
        used_mw = (G == 16 and len(X) > 16) or (G == -1 and len(X) >= 1500)
        assert (diarizer.kernel_stats("linkage_tie_fallbacks")["launches"] - fb0 == 2) == used_mw, (name, G)


@pytest.mark.parametrize("workers", [5, 31, 63])
def test_heap_replay_with_worker_workgroups_is_bit_identical(diarizer, workers):
    """k_linkage_hx (linkage_hx.hip): the reference's loop, heap included, on one thread, the row work of every step on `workers` workgroups
    (31 = one XCD, 63 = all XCDs).  Forced onto tie-free data (clustered: many bounds drop per merge, ordered change lists; uniform: stale heap
    tops rescanned cooperatively) and run where it is meant to run -- duplicated rows, the cooperative kernel stops at the first tie --
    the dendrogram is the oracle's bit for bit"""
    rng = np.random.default_rng(400 + workers)
    Y = _blobs(rng, 2600); Y[rng.integers(0, 2600, 400)] = Y[rng.integers(0, 2600, 400)]
    cases = [("blobs", _blobs(rng, 2500), 1), ("uniform", rng.random((3000, 3)), 1), ("duplicates", Y, 0)]
    for name, X, force in cases:
        _, Z_ref = orc.ahc(X, orc.THRESH_F32)
        j0 = diarizer.kernel_stats("linkage_hx_jobs")["launches"]
        diarizer.set_option("linkage_tie_kernel", workers)
        diarizer.set_option("linkage_force_heap", force)
        diarizer.set_option("linkage_hx_wide", 1 if workers == 5 else 0)          # (5 workers: also the 32-bit key / position form of jobs above 65 535 rows)
        diarizer.set_option("linkage_zero_phase", 0)                              # (the whole job in the replay, not only its merges at height 0)
        try:
            Z = diarizer.linkage(X)
        finally:
            diarizer.set_option("linkage_tie_kernel", 1)
            diarizer.set_option("linkage_force_heap", 0)
            diarizer.set_option("linkage_hx_wide", 0)
            diarizer.set_option("linkage_zero_phase", 1)
        assert np.array_equal(Z, Z_ref), (name, workers)
        assert diarizer.kernel_stats("linkage_hx_jobs")["launches"] == j0 + 1, (name, workers)      # it was this kernel, not the one-workgroup fallback
    assert diarizer.kernel_stats("linkage_hx_stale_scans")["flops"] > 0


@pytest.mark.parametrize("zero_phase", [1, 0])
def test_one_hour_sized_set_with_duplicated_rows_against_the_reference_clustering(diarizer, zero_phase):
    """12 989 rows (the live items of the planted hour), 5 % of them exact copies of other rows (looped audio, digital
    silence): the cooperative kernel stops at the first tie -- at height 0 -- and k_linkage_hx replays the reference's heap for the merges at
    height 0 (about 600), after which no two rows coincide and k_linkage_rg takes the other 12 000 merges (zero_phase = 0: the replay does
    the whole job); Z / the labels are those of the REFERENCE's own compiled clustering.cpp (oracle/_ref/libref_clustering.so; the C oracle
    where that is absent) bit for bit"""
    rng = np.random.default_rng(12989)
    N = 12989
    X = _blobs(rng, N, k=5)
    dup = rng.choice(N, N // 20, replace=False)
    X[dup] = X[rng.integers(0, N, len(dup))]
    R = orc.ref()
    if R is not None:
        Z_ref = np.zeros((N - 1, 4)); T_ref = np.zeros(N, np.int32)
        R.ref_linkage(np.ascontiguousarray(X), N, X.shape[1], Z_ref)
        R.ref_fcluster(Z_ref, N, orc.THRESH_F32, T_ref)
    else:
        T_ref, Z_ref = orc.ahc(X, orc.THRESH_F32)
    f0 = diarizer.kernel_stats("linkage_tie_fallbacks")["launches"]
    j0 = diarizer.kernel_stats("linkage_hx_jobs")["launches"]
    diarizer.reset_stats()
    diarizer.set_option("profile", 1)
    diarizer.set_option("linkage_zero_phase", zero_phase)
    try:
        Z = diarizer.linkage(X)
        ms = diarizer.kernel_stats("linkage_hx")["ms"] + diarizer.kernel_stats("linkage")["ms"]
    finally:
        diarizer.set_option("profile", 0)
        diarizer.set_option("linkage_zero_phase", 1)
    assert np.array_equal(Z, Z_ref)
    assert np.array_equal(sdhip.fcluster(Z, orc.THRESH_F32), T_ref)
    assert diarizer.kernel_stats("linkage_zero_phase_jobs" if zero_phase else "linkage_hx_jobs")["launches"] >= 1
    print("12 989 rows with 5 %% duplicates, zero phase %d: replay + cooperative kernel %.1f ms" % (zero_phase, ms))
    # measured: 84 ms with the zero phase (replay 8.7 ms for ~600 merges, k_linkage_rg 75 ms for the rest -- the tie-free job takes 73 ms);
    # 290 ms for the whole replay (clustered rows are its slow regime: in the reference's semantics ~1.5 stale heap tops are rescanned per
    # merge and many bounds drop per merge, each a hand-off round or a heap operation of the one master thread); k_linkage_heap: ~700 ms
    assert ms < (250.0 if zero_phase else 1200.0)


def test_duplicates_and_a_later_tie_end_in_the_whole_replay(diarizer):
    """run_linkage's zero phase: duplicated rows on a lattice -- the merges at height 0 go through the replay, k_linkage_rg continues and meets the
    lattice's ties at height 1, and the whole job is replayed: still the oracle's dendrogram bit for bit"""
    rng = np.random.default_rng(1728)
    g = np.stack(np.meshgrid(np.arange(12.0), np.arange(12.0), np.arange(12.0)), -1).reshape(-1, 3)
    X = g[rng.permutation(len(g))]
    X[rng.choice(len(X), 100, replace=False)] = X[rng.integers(0, len(X), 100)]
    _, Z_ref = orc.ahc(X, orc.THRESH_F32)
    z0 = diarizer.kernel_stats("linkage_zero_phase_jobs")["launches"]
    m0 = diarizer.kernel_stats("linkage_zero_phase_merges")["flops"]
    j0 = diarizer.kernel_stats("linkage_hx_jobs")["launches"]
    Z = diarizer.linkage(X)
    assert np.array_equal(Z, Z_ref)
    assert diarizer.kernel_stats("linkage_zero_phase_merges")["flops"] > m0          # the zero phase ran ...
    assert diarizer.kernel_stats("linkage_zero_phase_jobs")["launches"] == z0        # ... did not finish the job ...
    assert diarizer.kernel_stats("linkage_hx_jobs")["launches"] == j0 + 1            # ... and the whole replay did


def test_row_tie_at_the_first_merge_rebuilds_the_matrix(diarizer):
    """rows 1 and 2 lie at the same distance on either side of row 0, closer than anything else: the first merge's row has two neighbours at the merge
    height, which k_linkage_rg notices only after that merge's stores went out -- so the replay must not start from the matrix as it stands (run_linkage
    keeps linkage_prepare's matrix only when the cooperative kernel stopped in FRONT of its first merge; found by tools/linkage_fuzz.py)"""
    rng = np.random.default_rng(31)
    X = _blobs(rng, 1800)
    X[1] = X[0]; X[2] = X[0]
    X[1, 0] += 2.0 ** -12; X[2, 0] -= 2.0 ** -12
    assert np.linalg.norm(X[1] - X[0]) == np.linalg.norm(X[2] - X[0])
    _, Z_ref = orc.ahc(X, orc.THRESH_F32)
    f0 = diarizer.kernel_stats("linkage_tie_fallbacks")["launches"]
    for G in (-1, 16):
        diarizer.set_option("linkage_wgs", G)
        try:
            Z = diarizer.linkage(X)
        finally:
            diarizer.set_option("linkage_wgs", -1)
        assert np.array_equal(Z, Z_ref), G
    assert diarizer.kernel_stats("linkage_tie_fallbacks")["launches"] == f0 + 2


def test_heap_linkage_with_global_heap_is_bit_identical(diarizer):
    """k_linkage_heap with its heap in global memory (N - 1 > 2048 entries), tie-free data"""
    X = _blobs(np.random.default_rng(77), 3000)
    _, Z_ref = orc.ahc(X, orc.THRESH_F32)
    diarizer.set_option("linkage_wgs", 0)
    try:
        Z = diarizer.linkage(X)
    finally:
        diarizer.set_option("linkage_wgs", -1)
    assert np.array_equal(Z, Z_ref)


@pytest.mark.parametrize("square", [1, 0, 2])
@pytest.mark.parametrize("N,d,G,T", [(2500, 3, 32, 256), (4000, 2, 16, 512), (3000, 8, 64, 256), (6000, 3, -1, 0)])
def test_cooperative_linkage_on_unclustered_low_dimensional_data(diarizer, N, d, G, T, square):
    """uniform points in 2 / 3 / 8 dimensions: the opposite regime of the speaker embeddings.  Merges are balanced (many small clusters grow
    side by side), so rows go stale and are refreshed cooperatively all the time, the two-level bounds lose both levels regularly, and in
    the square form most merges involve a cluster OLDER than many bystanders: the entries are then read from the bystanders' rows (the
    last-rewrite index rule) instead of the pair's.  Dendrogram bit-identical to the oracle in both matrix layouts; the kernel must not
    have fallen back to the heap kernel (the data has no exact ties)."""
    rng = np.random.default_rng(31 * N + d)
    X = rng.random((N, d))
    _, Z_ref = orc.ahc(X, orc.THRESH_F32)
    fb0 = diarizer.kernel_stats("linkage_fallbacks")["launches"]
    rr0 = diarizer.kernel_stats("linkage_retry_rounds")["flops"]
    rg0 = diarizer.kernel_stats("linkage_rg_launches")["launches"]
    diarizer.set_option("linkage_wgs", G)
    diarizer.set_option("linkage_threads", T)
    diarizer.set_option("linkage_square", 1 if square else 0)          # 1: square matrix, k_linkage_rg; 2: square matrix, k_linkage_mw; 0: condensed, k_linkage_mw
    diarizer.set_option("linkage_kernel", 0 if square == 2 else -1)
    try:
        Z = diarizer.linkage(X)
    finally:
        diarizer.set_option("linkage_wgs", -1)
        diarizer.set_option("linkage_threads", 0)
        diarizer.set_option("linkage_square", -1)
        diarizer.set_option("linkage_kernel", -1)
    assert np.array_equal(Z, Z_ref)
    assert (diarizer.kernel_stats("linkage_rg_launches")["launches"] - rg0 == 1) == (square == 1)
    assert diarizer.kernel_stats("linkage_fallbacks")["launches"] == fb0
    assert diarizer.kernel_stats("linkage_retry_rounds")["flops"] > rr0          # stale candidates did reach the top here


def test_tie_free_data_never_leaves_the_cooperative_kernel(diarizer):
    fb0 = diarizer.kernel_stats("linkage_fallbacks")["launches"]
    X = _blobs(np.random.default_rng(5), 4000)
    _, Z_ref = orc.ahc(X, orc.THRESH_F32)
    assert np.array_equal(diarizer.linkage(X), Z_ref)
    assert diarizer.kernel_stats("linkage_fallbacks")["launches"] == fb0


def test_clustering_parity(diarizer):
    rng = np.random.default_rng(10)
    for trial, (c, k, s, pnan) in enumerate([(200, 5, 0.5, 0.2), (120, 3, 0.4, 0.15), (40, 2, 0.3, 0.0), (300, 8, 0.7, 0.3)]):
        cen = rng.standard_normal((k, 192)) * 2
        lab = rng.integers(0, k, (c, 3))
        lab[rng.random((c, 3)) < 0.03] = k - 1                       # a rare (small) cluster
        emb = (cen[lab] + s * rng.standard_normal((c, 3, 192))).astype(np.float32).astype(np.float64)
        emb[rng.random((c, 3)) < pnan] = np.nan
        h, K = diarizer.clustering(emb)
        h_ref, K_ref, _ = orc.clustering(emb)
        assert K == K_ref and np.array_equal(h, h_ref), trial
    # fewer than two embeddings -> all zeros (sd.cpp:2081-2088)
    e1 = np.full((4, 3, 192), np.nan); e1[2, 1] = 1.0
    h1, _ = diarizer.clustering(e1)
    assert not h1.any()
    # no cluster reaches min_cluster_size -> all zeros (sd.cpp:2371-2375)
    e2 = np.random.default_rng(1).standard_normal((10, 3, 192))
    h2, _ = diarizer.clustering(e2)
    h2_ref, _, _ = orc.clustering(e2)
    assert np.array_equal(h2, h2_ref)


def test_clustering_with_more_clusters_than_one_assignment_tile(diarizer):
    """K = 1 100 clusters (a recording with that many voices does not exist; a long one with hundreds does): k_assign's arg-max runs over
    the clusters in tiles of 1 024 scores, first maximum wins across tiles as Helper::argmax does (sd.cpp:293-316).  32-dimensional rows,
    which sd_clustering takes as they are, keep the oracle's O(N^2 d) pdist short."""
    rng = np.random.default_rng(123)
    K, per, d = 1100, 16, 32
    cen = rng.standard_normal((K, d))
    cen /= np.linalg.norm(cen, axis=1, keepdims=True)
    lab = np.repeat(np.arange(K), per)
    rng.shuffle(lab)
    c = (len(lab) + 2) // 3
    emb = np.full((c * 3, d), np.nan)
    emb[:len(lab)] = cen[lab] + 0.02 * rng.standard_normal((len(lab), d))
    emb = emb.astype(np.float32).astype(np.float64).reshape(c, 3, d)
    h, Kg = diarizer.clustering(emb)
    h_ref, K_ref, _ = orc.clustering(emb)
    assert Kg == K_ref == K
    assert np.array_equal(h, h_ref)
    assert len(np.unique(h.reshape(-1)[:len(lab)])) == K


# ------------------------------------------------------------------ a15 - a17
@pytest.mark.parametrize("c,kmax", [(30, 3), (12, 1), (77, 6), (1, 2)])
def test_reconstruct_parity(diarizer, c, kmax):
    rng = np.random.default_rng(c)
    n_s = 80000 + 8000 * (c - 1)
    sc = rng.random((c, 293, 3)).astype(np.float32)
    sc[min(3, c - 1), :, 1] = 0.01                      # an inactive local speaker -> -2
    b, _, cnt = diarizer.postseg(sc)
    hard = rng.integers(0, kmax, (c, 3)).astype(np.int32)
    turns = diarizer.reconstruct(sc, b, hard, cnt, n_s)
    b_ref = orc.binarize(sc)
    cnt_ref, win, ft = orc.speaker_count(b_ref)
    binr, st = orc.reconstruct(sc, orc.mark_inactive(b_ref, hard), cnt_ref, win, ft, n_s)
    t_ref = orc.to_annotation(binr, st)
    assert turns == t_ref


# ------------------------------------------------------------------ whole path
def test_whole_path_against_oracle(diarizer, weights):
    import synth
    pcm = synth.make_pcm(33.3, seed=11)
    turns = diarizer.diarize(pcm)
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    nc, _ = orc.num_chunks(len(pcm))
    seg = diarizer.segment(wav)
    masks = orc.select_masks(orc.binarize(seg))
    emb = diarizer.embed(wav, masks)
    # (1) with the GPU's network outputs injected, every non-neural stage must agree bit for bit
    t1 = pipeline_oracle.diarize_ref(pcm, weights[2], weights[3], seg_override=seg, emb_override=emb)
    assert turns == t1                                  # same turns in the same order (finalResult's std::sort included)
    # (2) against the full oracle (torch networks): same speakers, boundaries within +-1 frame (north star)
    t2 = pipeline_oracle.diarize_ref(pcm, weights[2], weights[3])
    assert len(t2) == len(turns)
    for a, b in zip(sorted(turns, key=tkey), sorted(t2, key=tkey)):
        assert a[2] == b[2] and abs(a[0] - b[0]) <= 0.016875 + 1e-9 and abs(a[1] - b[1]) <= 0.016875 + 1e-9


@pytest.mark.parametrize("n", [79999, 80000, 80001, 87999, 88000, 88001, 151999, 168000])
def test_whole_path_chunk_rule_edges(diarizer, weights, n):
    """lengths around the chunk rule (strict `<` loop + last-chunk branch, sd.cpp:1419, 1457): a recording shorter than
    one window, exactly one window, one sample more, and both sides of the next hop; the non-neural stages must agree bit
    for bit with the oracle fed the GPU's network outputs, the full oracle within +-1 frame"""
    import synth
    pcm = synth.make_pcm(n / 16000.0 + 0.01, seed=31)[:n]
    assert len(pcm) == n
    turns = diarizer.diarize(pcm)
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    seg = diarizer.segment(wav)
    assert seg.shape[0] == orc.num_chunks(n)[0]
    masks = orc.select_masks(orc.binarize(seg))
    emb = diarizer.embed(wav, masks)
    t1 = pipeline_oracle.diarize_ref(pcm, weights[2], weights[3], seg_override=seg, emb_override=emb)
    assert turns == t1                                  # same turns in the same order (finalResult's std::sort included)
    t2 = pipeline_oracle.diarize_ref(pcm, weights[2], weights[3])
    assert len(t2) == len(turns)
    for a, b in zip(sorted(turns, key=tkey), sorted(t2, key=tkey)):
        assert a[2] == b[2] and abs(a[0] - b[0]) <= 0.016875 + 1e-9 and abs(a[1] - b[1]) <= 0.016875 + 1e-9


@pytest.mark.parametrize("kind", ["silence", "full_scale_square", "dc", "one_sample"])
def test_whole_path_degenerate_audio(diarizer, weights, kind):
    """inputs the reference would not survive gracefully (no active frame anywhere: fewer than two embeddings, sd.cpp:2081-2088)
    or that sit at the edge of the int16 range: same turns as the oracle, no error, no NaN leaking into the output"""
    n = 16000 * 12
    if kind == "silence":
        pcm = np.zeros(n, np.int16)
    elif kind == "full_scale_square":
        pcm = np.where((np.arange(n) // 40) % 2 == 0, 32767, -32768).astype(np.int16)
    elif kind == "dc":
        pcm = np.full(n, 12345, np.int16)
    else:
        pcm = np.zeros(n, np.int16); pcm[n // 2] = -32768
    turns = diarizer.diarize(pcm)
    assert all(np.isfinite(t[0]) and np.isfinite(t[1]) and t[0] < t[1] for t in turns)
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    seg = diarizer.segment(wav)
    assert np.isfinite(seg).all()
    masks = orc.select_masks(orc.binarize(seg))
    emb = diarizer.embed(wav, masks)
    t1 = pipeline_oracle.diarize_ref(pcm, weights[2], weights[3], seg_override=seg, emb_override=emb)
    assert turns == t1                                  # same turns in the same order (finalResult's std::sort included)


def test_sharded_equals_unsharded_and_is_idempotent(diarizer):
    """multi-GPU split exercised on one GPU: two 32-aligned shards == one shot"""
    import sdhip
    import synth
    pcm = synth.make_pcm(60.0, seed=12)
    n = len(pcm)
    C, _ = sdhip.num_chunks(n)
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm).to(dev)
    seg = torch.zeros((C, 293, 3), dtype=torch.float32, device=dev)
    emb = torch.zeros((C * 3, 192), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    whole = diarizer.diarize_dev(d_pcm.data_ptr(), n)
    assert whole == diarizer.diarize_dev(d_pcm.data_ptr(), n)
    per, ranges = sdhip.plan_shards(n, 2)
    assert per % 32 == 0 and ranges[0][1] == ranges[1][0]
    for lo, hi in ranges:
        s0, s1 = sdhip.shard_sample_range(lo, hi, n)
        shard = d_pcm[s0:s1].contiguous()
        torch.cuda.synchronize()
        diarizer.shard_infer_dev(shard.data_ptr(), s0, s1 - s0, n, lo, hi, seg[lo:].data_ptr(), emb[lo * 3:].data_ptr())
    assert diarizer.finalize_dev(seg.data_ptr(), emb.data_ptr(), C, n) == whole


def test_cli_matches_api(diarizer, weights, golden_dir):
    import sdhip
    wav = os.path.join(golden_dir, "multi-speaker_1min.wav")          # BASELINE.json configs[0] input
    exe = os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd", "speakerDiarizer")
    out = subprocess.run([exe, weights[0], weights[1], wav], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    rule = "-" * 52
    i0 = lines.index(rule)
    i1 = lines.index(rule, i0 + 1)                                   # block between the two rules (sd.cpp:3435-3441)
    pcm, sr, ch = sdhip.read_wav(wav)
    turns = diarizer.diarize(pcm)
    assert lines[i0 + 1:i1] == [sdhip.format_turn(t) for t in turns]
    for lab in ("Segmenations time", "Embedding time", "Clustering time", "Time cost"):
        assert any(l.startswith(lab) for l in lines)
    # --precision x3 (f32 tensors, split fp16 MFMA operands): the same turns from the command line; an unknown precision is refused
    outx = subprocess.run([exe, weights[0], weights[1], wav, "--precision", "x3"], capture_output=True, text=True, timeout=600)
    assert outx.returncode == 0, outx.stderr
    lx = outx.stdout.splitlines()
    j0 = lx.index(rule)
    assert lx[j0 + 1:lx.index(rule, j0 + 1)] == lines[i0 + 1:i1]
    bad = subprocess.run([exe, weights[0], weights[1], wav, "--precision", "bf16"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2 and "f32, f16 or x3" in bad.stderr


# ------------------------------------------------------------------ full-size, size-independent properties
def test_full_size_linkage_properties(diarizer):
    """N = 21 573 (1 h of audio, BASELINE.json configs[2]): dendrogram validity + planted partition"""
    N = 21573
    rng = np.random.default_rng(0)
    cen = rng.standard_normal((4, 192))
    lab = rng.integers(0, 4, N)
    X = cen[lab] + 0.6 * rng.standard_normal((N, 192))
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Z = diarizer.linkage(X)
    ids = np.concatenate([Z[:, 0], Z[:, 1]]).astype(np.int64)
    assert np.array_equal(np.sort(ids), np.arange(2 * N - 2))         # every node merged exactly once
    assert (Z[:, 0] < Z[:, 1]).all() and Z[-1, 3] == N and (Z[:, 2] >= 0).all()
    size = np.ones(2 * N - 1)
    for k in range(N - 1):
        size[N + k] = size[int(Z[k, 0])] + size[int(Z[k, 1])]
    assert np.array_equal(size[N:], Z[:, 3])
    T = orc.fcluster_distance(Z, orc.THRESH_F32)                       # oracle's fcluster on the GPU dendrogram
    assert np.array_equal(T, diarizer.cluster(X, orc.THRESH_F32))
    assert T.max() == 4
    for k in range(4):
        assert len(set(T[lab == k].tolist())) == 1
    # first merge is the globally closest pair
    sub = X[:3000]
    from scipy.spatial.distance import pdist
    assert Z[0, 2] <= pdist(sub).min() + 1e-15


def test_full_size_one_hour_whole_path_properties(diarizer):
    """BASELINE.json configs[2] (1 h synthetic, the bench workload) through the C ABI: the run is deterministic, a 4-way
    chunk-range split (the 4-GPU plan, exercised on one GPU) reproduces it turn for turn, the stage outputs have the
    reference's geometry, and the rows the reference would overwrite with NaN are exactly the NaN rows."""
    import sdhip
    import synth
    pcm = synth.make_pcm(3600.0, seed=1234)
    n = len(pcm)
    C, _ = sdhip.num_chunks(n)
    assert C == 7191                                                   # SURVEY 8 size table
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm).to(dev)
    torch.cuda.synchronize()
    whole = diarizer.diarize_dev(d_pcm.data_ptr(), n)
    assert whole == diarizer.diarize_dev(d_pcm.data_ptr(), n)
    starts = [t[0] for t in whole]
    assert starts == sorted(starts)                                    # Annotation::finalResult order (sd.cpp:962-978)
    assert all(0.0 <= t[0] < t[1] <= 3600.0 + 0.02 for t in whole)
    seg = torch.zeros((C, 293, 3), dtype=torch.float32, device=dev)
    emb = torch.zeros((C * 3, 192), dtype=torch.float32, device=dev)
    per, ranges = sdhip.plan_shards(n, 4)
    assert per % 32 == 0
    for lo, hi in ranges:
        s0, s1 = sdhip.shard_sample_range(lo, hi, n)
        shard = d_pcm[s0:s1].contiguous()
        torch.cuda.synchronize()
        diarizer.shard_infer_dev(shard.data_ptr(), s0, s1 - s0, n, lo, hi, seg[lo:].data_ptr(), emb[lo * 3:].data_ptr())
    assert diarizer.finalize_dev(seg.data_ptr(), emb.data_ptr(), C, n) == whole
    # NaN rows == the reference's "too short" rule evaluated by the oracle on the same scores (sd.cpp:2479-2549)
    seg_h = seg.cpu().numpy()
    assert ((seg_h >= 0) & (seg_h <= 1)).all()
    b = orc.binarize(seg_h)
    masks = orc.select_masks(b)
    idx = (np.arange(80000, dtype=np.int64) * 293) // 80000            # Helper::interpolate, sd.cpp:746-767
    per_frame = np.bincount(idx, minlength=293)                        # samples each mask frame selects
    counts = ((masks > 0.5) * per_frame[None, :]).sum(1).astype(np.int64)
    bad = np.zeros(3 * C, bool)
    for b0 in range(0, 3 * C, 32):
        _, ts, an = orc.wav_lens(counts[b0:b0 + 32])
        bad[b0:b0 + 32] = ts | an
    assert np.array_equal(np.isnan(emb.cpu().numpy()[:, 0]), bad)
    cnt, _, _ = orc.speaker_count(b)
    assert len(cnt) == int(sdhip.lib().sd_count_frames(C))
