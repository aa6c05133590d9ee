"""Planted multi-speaker workload: segmentation scores derived from the synthetic turn
schedule (4 talkers, overlaps, silence) and one planted embedding per (chunk, local speaker) item,
NaN rows by the reference's own rule.  With them every stage after the two networks sees a
NON-degenerate case -- clean-mask branch, partial-length items, K >= 4 clusters, small -> large
re-assignment, top-count selection among several clusters, gap merging -- and the turns that come
out of the C ABI are compared bit for bit, order included, with the oracle at the 10-min
(BASELINE.json configs[1]) and 1-h (configs[2]) sizes."""
import numpy as np
import pytest

import synth
from oracle import nn_oracle as nn
from oracle import orc, pipeline_oracle

RTOL, ATOL = 1e-3, 1e-4


def planted_case(seconds, seed, outlier_every=211):
    pcm = synth.make_pcm(seconds, seed)
    n = len(pcm)
    turns = synth.with_duets(synth.schedule(seconds, seed))
    nc = synth.num_chunks(n)
    scores, assign = synth.planted_scores(turns, n, 0, nc)
    emb = synth.planted_embeddings(assign, outlier_every=outlier_every)
    return pcm, scores, assign, emb


def nan_rule(scores):
    """rows the reference overwrites with NaN (sd.cpp:2479-2549), from the oracle's masks of these scores"""
    b = orc.binarize(scores)
    masks = orc.select_masks(b)
    per_frame = np.bincount((np.arange(80000, dtype=np.int64) * 293) // 80000, minlength=293)     # Helper::interpolate, sd.cpp:746-767
    counts = ((masks > 0.5) * per_frame[None, :]).sum(1).astype(np.int64)
    bad = np.zeros(len(counts), bool)
    for b0 in range(0, len(counts), 32):
        _, ts, an = orc.wav_lens(counts[b0:b0 + 32])
        bad[b0:b0 + 32] = ts | an
    return b, masks, counts, bad


# ------------------------------------------------------------------ CPU: the planted workload itself
def test_planted_workload_is_shard_invariant_and_non_degenerate():
    pcm, scores, assign, emb = planted_case(600.0, 1234)
    n = len(pcm)
    nc = scores.shape[0]
    assert nc == 1191 == orc.num_chunks(n)[0]                              # SURVEY 8 size table, configs[1]
    turns = synth.with_duets(synth.schedule(600.0, 1234))
    s2, a2 = synth.planted_scores(turns, n, 64, 160)
    assert np.array_equal(scores[64:160], s2) and np.array_equal(assign[64:160], a2)
    assert np.array_equal(emb[192:480], synth.planted_embeddings(a2, chunk_lo=64, outlier_every=211))
    b, masks, counts, bad = nan_rule(scores)
    live = ~bad
    assert 0.3 < live.mean() < 0.8                                         # dead third speakers, silence
    assert ((counts[live] < 80000).mean() > 0.5)                           # most live items are partial length
    clean_used = (masks.reshape(nc, 3, 293) != b.transpose(0, 2, 1)).any(2)
    assert clean_used.sum() > 20                                           # overlap frames removed from some masks (sd.cpp:3071)
    e = emb.astype(np.float64)
    e[bad] = np.nan
    t, info = pipeline_oracle.diarize_ref(pcm, None, None, seg_override=scores, emb_override=e, return_all=True)
    assert info["K"] == 4 and len(t) > 60
    assert len({x[2] for x in t}) == 4
    assert info["count"].max() >= 2                                        # overlapped speech survives counting
    starts = [x[0] for x in t]
    assert len(starts) > len(set(starts))                                  # equal-start turns: finalResult's std::sort order matters


@pytest.mark.gpu
@pytest.mark.parametrize("seconds,seed", [(600.0, 1234), (3600.0, 1234)])
def test_finalize_from_planted_inputs_matches_oracle_bit_for_bit(diarizer, seconds, seed):
    """sd_finalize_dev alone (count, a10-a14 clustering, a15-a17 reconstruction) at configs[1] / configs[2] size"""
    import torch
    pcm, scores, assign, emb = planted_case(seconds, seed)
    n = len(pcm)
    nc = scores.shape[0]
    _, _, _, bad = nan_rule(scores)
    e32 = emb.copy()
    e32[bad] = np.nan
    dev = torch.device("cuda", 0)
    d_seg = torch.from_numpy(scores).to(dev)
    d_emb = torch.from_numpy(e32).to(dev)
    torch.cuda.synchronize()
    turns = diarizer.finalize_dev(d_seg.data_ptr(), d_emb.data_ptr(), nc, n)
    t_ref, info = pipeline_oracle.diarize_ref(pcm, None, None, seg_override=scores, emb_override=e32.astype(np.float64), return_all=True)
    assert turns == t_ref                                                  # same turns, same order (std::sort of finalResult included)
    assert info["K"] >= 4 and len(turns) >= (60 if seconds < 1000 else 400)
    # the interesting branches really fired
    X = e32[~bad].astype(np.float64)
    N = len(X)
    Xn = X / np.sqrt((X * X).sum(1)).astype(np.float32).astype(np.float64)[:, None]      # float norm, sd.cpp:332-357
    T, Z = orc.ahc(Xn, orc.THRESH_F32)
    mcs = min(15, max(1, int(round(0.1 * N))))
    sizes = np.bincount(T)[1:]
    assert (sizes < mcs).any() and (sizes >= mcs).sum() == info["K"]       # small clusters existed and were re-assigned (sd.cpp:2377-2412)
    assert len(orc.to_annotation(info["binary"], info["start"], min_off=0.0)) > len(t_ref)   # Track::support merged gaps (sd.cpp:911-941)
    assert (info["hard"] == -2).any() and info["count"].max() >= 2
    # a12 at this size: dendrogram bit-identical (N ~ 2 000 at 10 min, ~ 12 000 at 1 h)
    assert N > (1500 if seconds < 1000 else 9000)
    assert np.array_equal(diarizer.linkage(Xn), Z)
    h, K = diarizer.clustering(e32.astype(np.float64).reshape(nc, 3, 192))
    assert K == info["K"] and np.array_equal(orc.mark_inactive(info["binarized"], h), info["hard"])


@pytest.mark.gpu
def test_whole_path_with_planted_scores_and_embeddings(diarizer):
    """sd_diarize_dev end to end at the 10-min size: both networks run, their outputs are replaced by the planted ones
    (sd_set_planted), everything after them must give the oracle's turns"""
    import torch
    pcm, scores, assign, emb = planted_case(600.0, 1234)
    n, nc = len(pcm), scores.shape[0]
    _, _, _, bad = nan_rule(scores)
    e = emb.astype(np.float64)
    e[bad] = np.nan
    t_ref, info = pipeline_oracle.diarize_ref(pcm, None, None, seg_override=scores, emb_override=e, return_all=True)
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm).to(dev)
    d_sc, d_em = torch.from_numpy(scores).to(dev), torch.from_numpy(emb).to(dev)
    torch.cuda.synchronize()
    diarizer.set_planted(d_sc.data_ptr(), d_em.data_ptr(), 0, nc)
    try:
        turns = diarizer.diarize_dev(d_pcm.data_ptr(), n)
        # the 2-GPU plan on one GPU: shards see their part of the planted buffers
        import sdhip
        seg = torch.zeros((nc, 293, 3), dtype=torch.float32, device=dev)
        em = torch.zeros((nc * 3, 192), dtype=torch.float32, device=dev)
        per, ranges = sdhip.plan_shards(n, 2)
        for lo, hi in ranges:
            s0, s1 = sdhip.shard_sample_range(lo, hi, n)
            shard = d_pcm[s0:s1].contiguous()
            torch.cuda.synchronize()
            diarizer.shard_infer_dev(shard.data_ptr(), s0, s1 - s0, n, lo, hi, seg[lo:].data_ptr(), em[lo * 3:].data_ptr())
        sharded = diarizer.finalize_dev(seg.data_ptr(), em.data_ptr(), nc, n)
    finally:
        diarizer.set_planted(0, 0, 0, 0)
    assert info["K"] == 4 and len(t_ref) > 60
    assert turns == t_ref and sharded == t_ref
    assert np.array_equal(np.isnan(em.cpu().numpy()[:, 0]), bad)
    assert np.array_equal(seg.cpu().numpy(), scores)


@pytest.mark.gpu
def test_whole_path_with_planted_scores_and_real_embeddings(diarizer, weights):
    """planted scores only: the masks are partial (clean-mask branch, nvalid < 501, dead-row panel lists non-trivial) and the
    real ECAPA embeddings of those items go through clustering; the oracle is fed the GPU's embeddings, so every
    non-neural stage must agree bit for bit; a sample of 96 items is checked against the torch oracle"""
    import torch
    seconds = 600.0
    pcm = synth.make_pcm(seconds, 77)
    n = len(pcm)
    nc = synth.num_chunks(n)
    scores, _ = synth.planted_scores(synth.schedule(seconds, 77), n, 0, nc)
    b, masks, counts, bad = nan_rule(scores)
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    emb = diarizer.embed(wav, masks)
    assert np.array_equal(np.isnan(emb[:, 0]), bad)
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm).to(dev)
    d_sc = torch.from_numpy(scores).to(dev)
    torch.cuda.synchronize()
    diarizer.set_planted(d_sc.data_ptr(), 0, 0, nc)
    try:
        turns = diarizer.diarize_dev(d_pcm.data_ptr(), n)
    finally:
        diarizer.set_planted(0, 0, 0, 0)
    t_ref, info = pipeline_oracle.diarize_ref(pcm, weights[2], weights[3], seg_override=scores, emb_override=emb.astype(np.float64), return_all=True)
    assert turns == t_ref and len(turns) >= 5        # (the random-weight ECAPA separates items by length, not by talker: few clusters)
    # embed-level: 3 whole reference batches (96 items) of partial-length items against the torch oracle
    c0 = 320                                                               # multiple of 32 chunks -> item 960 starts a batch
    sub_wav = wav[c0 * 8000:(c0 + 31) * 8000 + 80000]
    sub_masks = masks[3 * c0:3 * c0 + 96]
    e_gpu = diarizer.embed(sub_wav, sub_masks)
    assert np.array_equal(e_gpu, emb[3 * c0:3 * c0 + 96], equal_nan=True)   # batch placement does not change a row's bits
    # ... nor does the batch size: with room for 96 full-length items per batch the same 960 items go through ~5 batches whose
    # boundaries are placed by the wide-tile search of run_embed (ecapa.hip)
    big_masks = masks[3 * c0:3 * c0 + 960]
    big_wav = wav[c0 * 8000:(c0 + 319) * 8000 + 80000]
    diarizer.set_option("emb_batch_items", 96)
    try:
        e_small_batches = diarizer.embed(big_wav, big_masks)
    finally:
        diarizer.set_option("emb_batch_items", 3072)
    assert np.array_equal(e_small_batches, emb[3 * c0:3 * c0 + 960], equal_nan=True)
    sigs = np.zeros((96, 80000), np.float32)
    cnts = np.zeros(96, np.int64)
    for i in range(96):
        sigs[i], cnts[i] = orc.mask_compact(orc.crop(sub_wav, (i // 3) * 8000), sub_masks[i])
    assert np.array_equal(cnts, counts[3 * c0:3 * c0 + 96])
    lens = np.zeros(96, np.float32)
    sbad = np.zeros(96, bool)
    for b0 in range(0, 96, 32):
        l, ts, an = orc.wav_lens(cnts[b0:b0 + 32])
        lens[b0:b0 + 32] = l
        sbad[b0:b0 + 32] = ts | an
    ok = ~sbad
    assert ok.sum() >= 30 and (lens[ok] < 0.999).sum() >= 20               # partial lengths: nvalid < 501
    st = nn.stft_ref(sigs[ok], weights[3]["stft.window"])
    f = nn.fbank_norm_ref(st, lens[ok], weights[3]["fbank.matrix"])
    e_ref = nn.EcapaOracle(weights[3])(f, lens[ok]).numpy()
    g = e_gpu[ok].astype(np.float64)
    cos = (g * e_ref).sum(1) / np.linalg.norm(g, axis=1) / np.linalg.norm(e_ref, axis=1)
    assert (1 - cos).max() < 1e-3
    np.testing.assert_allclose(g, e_ref, rtol=RTOL, atol=ATOL * np.abs(e_ref).max())


@pytest.mark.gpu
def test_fp16_mode_embeddings_at_scale_stay_within_the_north_star_tolerance(diarizer):
    """BASELINE.json configs[4] at the 10-min size: the real embeddings of the planted masks (partial lengths, compact rows, narrow
    MFA space, 256 x 256 and 128 x 128 fp16 kernels) in fp16 mode against the f32 path of the UNQUANTISED model.  What separates the two
    is, almost entirely, the rounding of the WEIGHTS to fp16 (test_fp16_weight_rounding_alone... shows 1.9e-3 for item 2 509 in exact
    arithmetic; the kernels and the fp16 activations add < 3e-4, test_fp16_mode_against_the_f32_reference_of_the_same_fp16_weights):
    a weight error is the same linear map in every frame, does not average out in the SE statistics, and the seeded random network's
    saturated SE gates amplify it (tools/diag_fp16_layers.py, profiles/r03_fp16_error_by_layer.txt).  So the distribution is asserted:
    median <= 1e-4, 99th percentile <= 1e-3 (the north-star bar), at most 1 % of the items above it, maximum <= 5e-3; the same rows NaN;
    identical turns, with the real embeddings and with the planted ones of the bench.  Mode 2 (hi + lo fp16 weight planes, 22-bit
    weights on the fp16 MFMA) removes the weight term: median <= 1e-5, 99th percentile <= 5e-4, at most 5 items above 1e-3."""
    import torch
    seconds = 600.0
    pcm, scores, assign, emb_planted = planted_case(seconds, 1234)
    n, nc = len(pcm), scores.shape[0]
    b, masks, counts, bad = nan_rule(scores)
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    e32 = diarizer.embed(wav, masks)
    diarizer.set_option("ecapa_precision", 2)
    try:
        e16x2 = diarizer.embed(wav, masks)
        diarizer.set_option("ecapa_precision", 1)
        e16 = diarizer.embed(wav, masks)
        dev = torch.device("cuda", 0)
        d_pcm = torch.from_numpy(pcm).to(dev)
        d_sc, d_em = torch.from_numpy(scores).to(dev), torch.from_numpy(emb_planted).to(dev)
        torch.cuda.synchronize()
        diarizer.set_planted(d_sc.data_ptr(), d_em.data_ptr(), 0, nc)
        turns16 = diarizer.diarize_dev(d_pcm.data_ptr(), n)
        diarizer.set_planted(d_sc.data_ptr(), 0, 0, nc)
        real16 = diarizer.diarize_dev(d_pcm.data_ptr(), n)
        diarizer.set_option("ecapa_precision", 0)
        real32 = diarizer.diarize_dev(d_pcm.data_ptr(), n)
        diarizer.set_planted(d_sc.data_ptr(), d_em.data_ptr(), 0, nc)
        turns32 = diarizer.diarize_dev(d_pcm.data_ptr(), n)
    finally:
        diarizer.set_planted(0, 0, 0, 0)
        diarizer.set_option("ecapa_precision", 0)
    assert np.array_equal(np.isnan(e16[:, 0]), bad) and np.array_equal(np.isnan(e32[:, 0]), bad)
    live = ~bad
    assert live.sum() > 1500 and not np.array_equal(e16[live], e32[live])
    a, c = e16[live].astype(np.float64), e32[live].astype(np.float64)
    cos = (a * c).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(c, axis=1)
    cd = 1 - cos
    assert np.median(cd) <= 1e-4 and np.quantile(cd, 0.99) <= 1e-3 and cd.max() <= 5e-3, (np.median(cd), np.quantile(cd, 0.99), cd.max())
    assert (cd > 1e-3).mean() <= 0.01
    a2 = e16x2[live].astype(np.float64)
    cd2 = 1 - (a2 * c).sum(1) / np.linalg.norm(a2, axis=1) / np.linalg.norm(c, axis=1)
    assert np.array_equal(np.isnan(e16x2[:, 0]), bad)
    assert np.median(cd2) <= 1e-5 and np.quantile(cd2, 0.99) <= 5e-4 and (cd2 > 1e-3).sum() <= 5 and cd2.max() <= 2e-3, (np.median(cd2), np.quantile(cd2, 0.99), cd2.max())
    rel = np.linalg.norm(a - c, axis=1) / np.linalg.norm(c, axis=1)
    assert np.median(rel) <= 1e-2 and rel.max() <= 1e-1, (np.median(rel), rel.max())
    assert turns16 == turns32 and len(turns16) > 60
    assert real16 == real32 and len(real16) >= 5


def test_fp16_weight_rounding_on_the_calibrated_pack_is_harmless():
    """CPU, exact f32 arithmetic, the experiment of the test above on the CALIBRATED seeded pack (same conv weights; the 31 BatchNorms carry
    statistics learnt from one calibration batch, so SE pre-activations are O(1) and the gates unsaturated, as in a trained ECAPA):
    rounding the conv weights to fp16 moves the same 32 items by <= 1e-4 -- two orders below what it does to the plain pack.  The plain
    pack's 1.9e-3 is a property of its saturated gates, not of fp16 weights."""
    pcm, scores, assign, emb_planted = planted_case(600.0, 1234)
    b, masks, counts, bad = nan_rule(scores)
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    wc = nn.calibrated_embedding_weights()                                  # package data (weightpack.py + calibrated_bn_4322.npz)
    w0 = nn.synth_embedding_weights()
    assert all(np.array_equal(wc[k], w0[k]) for k in w0 if not (k.endswith("running_mean") or k.endswith("running_var") or ".norm." in k or k.startswith("asp_bn")))
    fresh = nn.calibrate_embedding_weights()                                # the stored statistics are what the calibration computes
    for k in w0:
        np.testing.assert_allclose(wc[k], fresh[k], rtol=1e-4, atol=1e-5, err_msg=k)
    sig = np.zeros((32, 80000), np.float32)
    cn = np.zeros(32, np.int64)
    for j, i in enumerate(range(2496, 2528)):
        sig[j], cn[j] = orc.mask_compact(orc.crop(wav, (i // 3) * 8000), masks[i])
    lens, ts, an = orc.wav_lens(cn)
    feats = nn.fbank_norm_ref(nn.stft_ref(sig, wc.get("stft.window")), lens, wc["fbank.matrix"])
    cd = _cosd(nn.EcapaOracle(_fp16_weights(wc))(feats, lens).numpy(), nn.EcapaOracle(wc)(feats, lens).numpy())
    assert cd[~(ts | an)].max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("seconds", [600.0, 3600.0])
def test_fp16_mode_holds_1e_3_on_every_item_of_the_calibrated_pack(diarizer_calibrated, weights_calibrated, seconds):
    """BASELINE.json configs[4], the tolerance check proper (north star: embedding cosine distances within 1e-3): on the calibrated seeded
    pack `ecapa_precision = 1` (fp16 weights AND activations on the fp16 MFMA, f32 accumulation) stays within 1e-3 of the f32 path on
    EVERY live item of the planted 10-min (2 130 items) and 1-h (12 989 items) recordings -- maximum, not a quantile -- with the same
    NaN rows and the same turns; and the f32 path itself is the torch oracle's (96 items checked element-wise)."""
    import torch
    d = diarizer_calibrated
    pcm, scores, assign, emb_planted = planted_case(seconds, 1234)
    n, nc = len(pcm), scores.shape[0]
    b, masks, counts, bad = nan_rule(scores)
    dev = torch.device("cuda", 0)
    d_pcm, d_sc = torch.from_numpy(pcm).to(dev), torch.from_numpy(scores).to(dev)
    torch.cuda.synchronize()
    d.set_planted(d_sc.data_ptr(), 0, 0, nc)                                # scores planted (the masks), embeddings real
    try:
        t32 = d.diarize_dev(d_pcm.data_ptr(), n)
        e32 = d.read_ws("dz_emb", np.float32, nc * 3 * 192).reshape(-1, 192).astype(np.float64)
        d.set_option("ecapa_precision", 1)
        t16 = d.diarize_dev(d_pcm.data_ptr(), n)
        e16 = d.read_ws("dz_emb", np.float32, nc * 3 * 192).reshape(-1, 192).astype(np.float64)
    finally:
        d.set_option("ecapa_precision", 0)
        d.set_planted(0, 0, 0, 0)
    assert np.array_equal(np.isnan(e32[:, 0]), bad) and np.array_equal(np.isnan(e16[:, 0]), bad)
    live = ~bad
    cd = _cosd(e16[live], e32[live])
    assert live.sum() >= (2000 if seconds < 1000 else 12000) and not np.array_equal(e16[live], e32[live])
    assert cd.max() <= 1e-3, (cd.max(), np.quantile(cd, 0.99), np.median(cd), int((cd > 1e-3).sum()))
    assert t16 == t32
    if seconds < 1000:
        # the f32 path of this pack against the torch oracle: three reference batches of partial-length items
        wav = pcm.astype(np.float32) / np.float32(32768.0)
        wc = weights_calibrated[3]
        for b0 in (2496, 960, 3232):
            idx = np.arange(b0, b0 + 32)
            sig = np.zeros((32, 80000), np.float32)
            cn = np.zeros(32, np.int64)
            for j, i in enumerate(idx):
                sig[j], cn[j] = orc.mask_compact(orc.crop(wav, (i // 3) * 8000), masks[i])
            lens, ts, an = orc.wav_lens(cn)
            ok = ~(ts | an)
            if an or ok.sum() == 0:
                continue
            f = nn.fbank_norm_ref(nn.stft_ref(sig[ok], wc.get("stft.window")), lens[ok], wc["fbank.matrix"])
            e_ref = nn.EcapaOracle(wc)(f, lens[ok]).numpy().astype(np.float64)
            g = e32[idx][ok]
            assert _cosd(g, e_ref).max() < 1e-3
            np.testing.assert_allclose(g, e_ref, rtol=RTOL, atol=ATOL * np.abs(e_ref).max())


def _fp16_weights(w):
    """the weights the fp16 mode multiplies with: every per-frame conv layer rounded to fp16 (SE and fc stay f32, csrc/weights.cpp)"""
    out = dict(w)
    for k, v in w.items():
        if k.endswith("conv.weight") and ".se." not in k and not k.startswith("fc"):
            out[k] = np.asarray(v, np.float32).astype(np.float16).astype(np.float32)
    return out


def _cosd(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return 1 - (a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)


@pytest.mark.gpu
def test_x3_mode_at_scale_is_f32_grade(diarizer):
    """ecapa_precision = 3 at the 10-min size (2 130 live items, partial lengths, compact rows, every batch shape run_embed picks): the real
    embeddings within 1e-6 (cosine distance) of the f32 path on EVERY item -- where fp16 mode leaves 11 items above 1e-3 and the hi + lo
    weight planes of mode 2 still 3 -- the same rows NaN, identical turns from the real embeddings and from the planted ones, and the
    same bits whatever the batch size."""
    import torch
    seconds = 600.0
    pcm, scores, assign, emb_planted = planted_case(seconds, 1234)
    n, nc = len(pcm), scores.shape[0]
    b, masks, counts, bad = nan_rule(scores)
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    e32 = diarizer.embed(wav, masks)
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm).to(dev)
    d_sc, d_em = torch.from_numpy(scores).to(dev), torch.from_numpy(emb_planted).to(dev)
    torch.cuda.synchronize()
    diarizer.set_option("ecapa_precision", 3)
    try:
        ex = diarizer.embed(wav, masks)
        diarizer.set_option("emb_batch_items", 96)
        ex_small = diarizer.embed(wav[:(319 * 8000 + 80000)], masks[:960])
        diarizer.set_option("emb_batch_items", 3072)
        diarizer.set_planted(d_sc.data_ptr(), 0, 0, nc)
        real_x = diarizer.diarize_dev(d_pcm.data_ptr(), n)
        diarizer.set_planted(d_sc.data_ptr(), d_em.data_ptr(), 0, nc)
        turns_x = diarizer.diarize_dev(d_pcm.data_ptr(), n)
        diarizer.set_option("ecapa_precision", 0)
        turns_32 = diarizer.diarize_dev(d_pcm.data_ptr(), n)
        diarizer.set_planted(d_sc.data_ptr(), 0, 0, nc)
        real_32 = diarizer.diarize_dev(d_pcm.data_ptr(), n)
    finally:
        diarizer.set_planted(0, 0, 0, 0)
        diarizer.set_option("emb_batch_items", 3072)
        diarizer.set_option("ecapa_precision", 0)
    assert np.array_equal(np.isnan(ex[:, 0]), bad) and np.array_equal(np.isnan(e32[:, 0]), bad)
    live = ~bad
    assert live.sum() > 1500 and not np.array_equal(ex[live], e32[live])
    cd = _cosd(ex[live], e32[live])
    assert cd.max() < 1e-6 and np.median(cd) < 1e-9, (cd.max(), np.median(cd))
    assert np.array_equal(ex_small, ex[:960], equal_nan=True)
    assert turns_x == turns_32 and len(turns_x) > 50
    assert real_x == real_32 and len(real_x) >= 5


def test_fp16_weight_rounding_alone_exceeds_the_bar_on_the_seeded_model():
    """CPU, exact f32 arithmetic: rounding the conv weights of the seeded synthetic ECAPA to fp16 -- nothing else -- moves item 2 509 of
    the planted 10-min set by a cosine distance of 1.9e-3.  BASELINE configs[4] ("fp16 ECAPA-TDNN weights") therefore cannot hold 1e-3
    against the unquantised f32 model on every item, whatever the kernels do; the kernels' own share is checked against the f32
    reference of the SAME fp16 weights (next test)."""
    pcm, scores, assign, emb_planted = planted_case(600.0, 1234)
    b, masks, counts, bad = nan_rule(scores)
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    we = nn.synth_embedding_weights()
    items = [i for i in range(2496, 2528)]
    sig = np.zeros((32, 80000), np.float32)
    cn = np.zeros(32, np.int64)
    for j, i in enumerate(items):
        sig[j], cn[j] = orc.mask_compact(orc.crop(wav, (i // 3) * 8000), masks[i])
    lens, ts, an = orc.wav_lens(cn)
    assert np.array_equal(ts | an, bad[2496:2528])
    feats = nn.fbank_norm_ref(nn.stft_ref(sig, we.get("stft.window")), lens, we["fbank.matrix"])
    e0 = nn.EcapaOracle(we)(feats, lens).numpy()
    eq = nn.EcapaOracle(_fp16_weights(we))(feats, lens).numpy()
    cd = _cosd(eq, e0)
    live = ~(ts | an)
    assert 1.5e-3 < cd[13] == cd[live].max() < 2.5e-3                       # item 2 509
    assert np.median(cd[live]) < 1e-4


@pytest.mark.gpu
def test_fp16_mode_against_the_f32_reference_of_the_same_fp16_weights(diarizer, weights):
    """BASELINE configs[4], the tolerance check proper: the fp16 HIP path (fp16 weights, fp16 activations, f32 accumulation) against
    the torch f32 oracle running the SAME fp16-rounded weights, on the three 32-item batches of the planted 10-min set that hold the
    worst items of the mode: every embedding within 1e-3 (measured <= 2.7e-4).  Beside it, on the same items: the f32 HIP path against
    the unquantised oracle (<= 1e-6), and the distance the weight rounding alone creates (the oracle against itself: > 1e-3 on the worst
    item of every batch) -- the part of the distance to the unquantised model that no kernel can remove."""
    pcm, scores, assign, emb_planted = planted_case(600.0, 1234)
    b, masks, counts, bad = nan_rule(scores)
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    we = weights[3]
    wq = _fp16_weights(we)
    feats, lens = diarizer.frontend(wav, masks)
    n_items = 0
    for w0 in (2509, 3240, 3315):
        b0 = (w0 // 32) * 32
        idx = np.array([i for i in range(b0, b0 + 32) if not bad[i]])
        f, l = np.ascontiguousarray(feats[idx]), np.ascontiguousarray(lens[idx])
        e_exact = nn.EcapaOracle(we)(f, l).numpy()
        e_q = nn.EcapaOracle(wq)(f, l).numpy()
        e32 = diarizer.ecapa(f, l)
        try:
            diarizer.set_option("ecapa_precision", 1)
            e16 = diarizer.ecapa(f, l)
            diarizer.set_option("ecapa_precision", 2)
            e16x2 = diarizer.ecapa(f, l)
        finally:
            diarizer.set_option("ecapa_precision", 0)
        assert _cosd(e32, e_exact).max() < 1e-6
        assert _cosd(e16, e_q).max() < 1e-3                                   # the north-star bar, against the f32 reference of the model that runs
        assert _cosd(e_q, e_exact).max() > 1e-3                               # ... which the weight rounding alone puts beyond 1e-3 of the unquantised one
        assert _cosd(e16x2, e_exact).max() < 2e-3 and np.median(_cosd(e16x2, e_exact)) < 2e-5
        n_items += len(idx)
    assert n_items >= 64


@pytest.mark.gpu
def test_eight_hour_job_as_eight_rank_plan_gives_the_single_call_turns(diarizer):
    """BASELINE.json configs[3] on one GPU: 8 h of audio (57 591 chunks, ~100 000 live items) through sd_diarize_dev, and the same job as
    the 8-rank plan of sd_diarize_sharded_dev (`virtual_world` = 8: every rank's 32-aligned chunk range inferred into the slot the RCCL
    all-gather would put it in, status records, assembly with a reduced rank-0 share, finalize): identical turns, order included.
    The audio is one synthetic hour repeated (its content cannot reach the output: the networks' outputs are replaced by the planted
    ones); every hour has its own schedule, so the clustering sees the full 8-hour problem."""
    import torch
    import sdhip
    hour = synth.make_pcm(3600.0, 1234)
    pcm = np.tile(hour, 8)
    n = len(pcm)
    nc = synth.num_chunks(n)
    assert nc == 57591                                                     # SURVEY 8 size table, configs[3]
    sched = []
    for h in range(8):
        for (s, e, k, ov) in synth.with_duets(synth.schedule(3600.0, 1234 + h)):
            sched.append((s + h * len(hour), e + h * len(hour), k, ov))
    scores, assign = synth.planted_scores(sched, n, 0, nc)
    emb = synth.planted_embeddings(assign)
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm).to(dev)
    d_sc, d_em = torch.from_numpy(scores).to(dev), torch.from_numpy(emb).to(dev)
    torch.cuda.synchronize()
    diarizer.set_planted(d_sc.data_ptr(), d_em.data_ptr(), 0, nc)
    diarizer.comm_init(sdhip.comm_unique_id(), 0, 1)
    try:
        diarizer.reset_stats()
        whole = diarizer.diarize_dev(d_pcm.data_ptr(), n)
        live = diarizer.kernel_stats("items_live")
        diarizer.set_option("virtual_world", 8)
        diarizer.set_option("rank0_permille", 60)
        plan = diarizer.diarize_sharded_dev(d_pcm.data_ptr(), 0, n, n)
    finally:
        diarizer.set_option("virtual_world", 0)
        diarizer.set_option("rank0_permille", -1)
        diarizer.comm_destroy()
        diarizer.set_planted(0, 0, 0, 0)
    assert len(whole) > 4000 and len({t[2] for t in whole}) == 4
    assert plan == whole
    assert live["flops"] / max(live["launches"], 1) > 90000                # N of the AHC: ~100 000 live embeddings
