"""The glue stages pinned on the REFERENCE'S OWN C++ (round 4).  oracle/_ref/libref_glue.so is every line range of
/root/reference/pipeline/src/speakerDiarizer.cpp that does not touch onnxruntime / libtorch -- Helper, Segment, Annotation,
SlidingWindow, PipelineHelper::aggregate, SegmentModel's binarize_swf / speaker_count / trim / crop, Cluster, crop_segment,
to_diarization, reconstruct, to_annotation, and the three model-free blocks of speakerDiarization() (mask choice, wav_lens rule,
inactive speakers) -- compiled UNEDITED where the file lies (oracle/ref_build/make_glue_tu.sh, oracle/Makefile) behind a C shim of
ours that only calls it.  Nothing of it is in this repository; the .so is git-ignored and travels to the GPU box like our own.

CPU (-m "not gpu"): the C oracle (oracle/sd_oracle.c) against the reference's compiled code, stage by stage and end to end, in
the regimes where the C++ and the Python it was ported from DIFFER (the C++ wins) -- un-normalised centroids in the
small -> large re-assignment (#6, sets on which a label really moves), float thresholds (#7), padding to 80000 with
wav_len = len / max_len (#1), closest_frame clamp (#4), float crop indices (#11), unstable std::sort of equal-start turns.
GPU (-m gpu): the HIP path's stage entry points and sd_finalize_dev against the same compiled reference: the turns of the planted
10-min and 1-h recordings must be the reference's own, order included."""
import os

import numpy as np
import pytest

import synth
from oracle import orc

pytestmark = pytest.mark.skipif(orc.refglue() is None, reason="oracle/_ref/libref_glue.so absent (built where /root/reference exists)")


@pytest.fixture(scope="module")
def G():
    return orc.RefGlue()


def rng_scores(rng, c, onset_hits=True):
    """segmentation-like scores: smooth-ish activity per speaker, values on both sides of the threshold, some exactly at
    float32(onset) (the C++ compares in double: simply 'off', App. B / test_reference_nn_glue)"""
    t = np.linspace(0, 1, 293)[None, :, None]
    ph = rng.uniform(0, 6.28, (c, 1, 3))
    fr = rng.uniform(1, 9, (c, 1, 3))
    s = 0.5 + 0.5 * np.sin(6.28 * fr * t + ph) * rng.uniform(0.2, 1.0, (c, 1, 3))
    s += rng.normal(0, 0.03, s.shape)
    s = np.clip(s, 0, 1).astype(np.float32)
    s[rng.random(s.shape) < 0.02] = 0.0
    if onset_hits:
        s[rng.random(s.shape) < 0.01] = np.float32(orc.ONSET)
    dead = rng.random((c, 3)) < 0.25
    s[np.broadcast_to(dead[:, None, :], s.shape)] *= np.float32(0.1)          # inactive local speakers
    return s


# ---------------------------------------------------------------------------------------------------------------- helpers
def test_np_rint_and_closest_frame_are_the_reference_build(G, golden_dir):
    """the reference's fixture through the reference's own compiled np_rint / closest_frame: this also proves the build reads
    abs() as the floating overload (with ::abs(int) np_rint(1.3) would be 0, see oracle/Makefile)"""
    rows = [ln.strip().split(",") for ln in open(os.path.join(golden_dir, "closest_frame.txt")) if ln.strip()]
    assert len(rows) == 10000
    t = 0.0
    for f, _ in rows:                                                                   # the reference's test(), sd.cpp:3236-3277
        assert G.closest_frame(t) == int(f)
        t += 0.5
    assert G.np_rint(1.3) == 1 and G.np_rint(2.5) == 2 and G.np_rint(3.5) == 4 and G.np_rint(-0.5) == 0
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-50, 5000, 20000), np.arange(0, 400) + 0.5, -np.arange(0, 50) - 0.5])
    for x in xs:
        assert orc.np_rint(x) == G.np_rint(x)
    ts = np.concatenate([rng.uniform(-1, 4000, 20000), 0.016875 * (np.arange(3000) + 0.5) + 0.0084375])
    for st, step, dur in ((0.0, 0.016875, 0.016875), (0.5, 0.016875, 0.016875), (0.0, 0.5, 5.0)):
        for t in ts[::7]:
            assert orc.closest_frame(t, st, step, dur) == G.closest_frame(t, st, step, dur)


def test_argsort_argmax_cdist(G):
    rng = np.random.default_rng(1)
    for _ in range(200):
        v = rng.integers(0, 4, rng.integers(1, 9)).astype(np.float64) * 0.25            # many ties: the sort must be stable
        assert np.array_equal(G.argsort(v), np.argsort(v, kind="stable"))
    soft = rng.normal(size=(50, 3, 5))
    soft[3, 1] = np.nan                                                                 # NaN row -> 0 (sd.cpp:293-316)
    soft[4, 2, 1] = soft[4, 2, 3] = 9.0                                                 # first maximum wins
    am = G.argmax(soft)
    assert am[3, 1] == 0 and am[4, 2] == 1
    ok = ~np.isnan(soft).any(2)
    assert np.array_equal(am[ok], np.argmax(soft, 2)[ok])
    A, B = rng.normal(size=(6, 192)) * rng.uniform(0.1, 9, (6, 1)), rng.normal(size=(4, 192))
    d = G.cosine_cdist(A, B)
    from scipy.spatial.distance import cdist
    assert np.allclose(d, cdist(A, B, "cosine"), rtol=0, atol=1e-14)


# ------------------------------------------------------------------------------------------------------- a4, a6, a7: masks
def test_binarize_clean_and_mask_choice(G):
    rng = np.random.default_rng(2)
    seg = rng_scores(rng, 40)
    b_ref = G.binarize(seg)
    b = orc.binarize(seg)
    assert np.array_equal(b, b_ref) and 0.2 < b.mean() < 0.8
    assert np.array_equal(orc.select_masks(b), G.select_masks(b))
    # the 'exactly min_num_frames clean frames' edge (sum > 3, strict: sd.cpp:3071) and its neighbours
    bb = np.zeros((6, 293, 3))
    for i, k in enumerate((2, 3, 4, 5, 0, 293)):
        bb[i, 10:10 + k, 0] = 1.0                    # speaker 0 alone on k frames
        bb[i, 100:160, 0] = 1.0
        bb[i, 100:160, 1] = 1.0                      # overlapped with speaker 1 elsewhere
    m_ref = G.select_masks(bb)
    assert np.array_equal(orc.select_masks(bb), m_ref)
    assert m_ref[0 * 3 + 0].sum() == 62 and m_ref[1 * 3 + 0].sum() == 63 and m_ref[2 * 3 + 0].sum() == 4   # full, full, clean
    cl = G.clean_segmentations(b)
    two = b.sum(2) >= 2
    assert (cl[two] == 0).all() and np.array_equal(cl[~two], b[~two])


def test_crop_and_embedding_inputs(G):
    rng = np.random.default_rng(3)
    n = 8000 * 20 + 3217
    wav = rng.normal(0, 0.1, n).astype(np.float32)
    nc, _ = orc.num_chunks(n)
    for i in (0, 1, nc - 2, nc - 1):
        assert np.array_equal(orc.crop(wav, i * 8000), G.crop(wav, i * 8000))
    assert (orc.crop(wav, (nc - 1) * 8000)[-100:] == 0).all()                           # zero padding past the end
    seg = rng_scores(rng, nc, onset_hits=False)
    b = orc.binarize(seg)
    masks = orc.select_masks(b)
    masks[5] = 0
    masks[5, 7:9] = 1                                                                   # 2 frames = 546 samples < 640: too short
    items = list(range(0, 32))
    wavs = np.stack([orc.crop(wav, (i // 3) * 8000) for i in items])
    sig_ref, lens_ref, ts_ref, an_ref = G.embedding_inputs(wavs, masks[items])
    assert not an_ref and ts_ref[5] and lens_ref[5] == 1.0
    cnts = np.zeros(32, np.int64)
    for j, i in enumerate(items):
        sig, cnts[j] = orc.mask_compact(wavs[j], masks[i])
        assert np.array_equal(sig, sig_ref[j])                                          # compaction + zero fill to 80000 (App. B #1)
    lens, ts, an = orc.wav_lens(cnts)
    assert np.array_equal(lens, lens_ref) and np.array_equal(ts, ts_ref) and an == an_ref
    assert (lens < 1).any() and lens.max() == 1.0
    # a batch in which nobody reaches 640 samples: the reference returns NaN rows without calling the model
    short = np.zeros((4, 293), np.float32)
    short[:, 3] = 1.0
    _, _, _, an2 = G.embedding_inputs(wavs[:4], short)
    assert an2 and orc.wav_lens(np.full(4, 274, np.int64))[2]


# ------------------------------------------------------------------------------------------------- a5: count (aggregate #1)
@pytest.mark.parametrize("c,tail", [(1, 0), (2, 4000), (7, 123), (33, 7999), (109, 0), (400, 1)])
def test_speaker_count_overlap_add(G, c, tail):
    """SegmentModel::speaker_count -> trim -> PipelineHelper::aggregate(missing = 0) -> np_rint (sd.cpp:1665-1782, 1167-1311):
    the overlap-ADD AND AVERAGE of the reference itself, not only its frame indexing"""
    rng = np.random.default_rng(100 + c)
    b = orc.binarize(rng_scores(rng, c))
    n = (c - 1) * 8000 + 80000 + tail
    cnt_ref, win_ref = G.speaker_count(b, n)
    cnt, win, ft = orc.speaker_count(b)
    assert np.array_equal(cnt, cnt_ref)
    assert np.array_equal(win, win_ref[:3]) and ft == win_ref[3] == 235
    if c >= 7:
        assert cnt.max() >= 2
    # the averaged values themselves (before np_rint): exact equality of the doubles
    trimmed = b[:, 29:264, :].sum(2, keepdims=True)
    a_ref = G.aggregate(trimmed, 0.5, 0.5, 4.0, 235, missing=0.0, skip_average=False)
    a = orc.aggregate(trimmed, 0.5, 0.5, 4.0, missing=0.0, skip_average=False)
    assert np.array_equal(a, a_ref) and len(a) == len(cnt)
    if c > 20:
        frac = a - np.floor(a)
        assert ((frac > 0) & (frac < 1)).any()                                          # real averages, not only integers


def test_aggregate_second_call_site_with_nan_entries(G):
    """to_diarization's call (sd.cpp:2646-2651): skip_average = true, missing = 0, NaN where a cluster has no local speaker"""
    rng = np.random.default_rng(5)
    c, K = 60, 5
    x = rng.random((c, 293, K))
    x[rng.random((c, 1, K)).repeat(293, 1) < 0.5] = np.nan
    x[:, :, 4] = np.nan                                                                 # a cluster nobody maps to: stays `missing`
    a_ref = G.aggregate(x, 0.0, 0.5, 5.0, 80000 + (c - 1) * 8000, missing=0.0, skip_average=True)
    a = orc.aggregate(x, 0.0, 0.5, 5.0, missing=0.0, skip_average=True)
    assert np.array_equal(a, a_ref) and (a[:, 4] == 0).all() and a.max() > 3
    a_ref = G.aggregate(x, 0.0, 0.5, 5.0, 80000 + (c - 1) * 8000, skip_average=False)   # default missing = NaN
    a = orc.aggregate(x, 0.0, 0.5, 5.0, skip_average=False)
    assert np.array_equal(np.isnan(a), np.isnan(a_ref)) and np.isnan(a[:, 4]).all()
    assert np.array_equal(np.nan_to_num(a), np.nan_to_num(a_ref))


# ------------------------------------------------------------------------------------ a10-a14: Cluster (App. B #6 regime)
def loose_set(seed, n_big=(60, 45, 38), n_small=(4, 3, 5, 2), d=192):
    """NOT well separated, and with row norms spread over 25x: large clusters around three centres, small clusters BETWEEN
    centres.  A small cluster goes to the large one with the nearest centroid (cosine); the C++ takes centroids of the RAW rows
    (sd.cpp:2386-2390), the Python of the normalised ones (Clustering.py:319-321, 410-422) -- with norms this uneven the
    two centroids of a cluster point in visibly different directions."""
    rng = np.random.default_rng(seed)
    cen = rng.normal(size=(3, d))
    cen /= np.linalg.norm(cen, axis=1, keepdims=True)
    rows = []
    for k, nb in enumerate(n_big):
        dirs = cen[k] + 0.035 * rng.normal(size=(nb, d))
        # direction drifts WITH the norm: the raw mean leans to the loud rows, the normalised mean does not
        drift = rng.normal(size=d) / np.sqrt(d)
        scale = np.exp(rng.uniform(np.log(0.2), np.log(5.0), nb))
        dirs = dirs + 0.22 * (np.log(scale) / np.log(5.0))[:, None] * drift
        rows.append(dirs / np.linalg.norm(dirs, axis=1, keepdims=True) * scale[:, None])
    for j, ns in enumerate(n_small):
        a, b = rng.choice(3, 2, replace=False)
        w = rng.uniform(0.42, 0.58)
        mid = w * cen[a] + (1 - w) * cen[b] + 0.6 * rng.normal(size=d) / np.sqrt(d)
        pts = mid + 0.01 * rng.normal(size=(ns, d))
        rows.append(pts * np.exp(rng.uniform(np.log(0.2), np.log(5.0), ns))[:, None])
    X = np.concatenate(rows)
    return X[rng.permutation(len(X))]


def python_rule_labels(X):
    """what the PYTHON would answer (centroids of the normalised rows), on top of the same dendrogram cut"""
    Xn = X / np.sqrt((X * X).sum(1)).astype(np.float32).astype(np.float64)[:, None]
    T, _ = orc.ahc(Xn, orc.THRESH_F32)
    T = T - 1
    N = len(X)
    mcs = min(15, max(1, int(round(0.1 * N))))
    ids, cnt = np.unique(T, return_counts=True)
    large, small = ids[cnt >= mcs], ids[cnt < mcs]
    if len(large) == 0 or len(small) == 0:
        return None
    lc = np.stack([Xn[T == k].mean(0) for k in large])
    sc = np.stack([Xn[T == k].mean(0) for k in small])
    from scipy.spatial.distance import cdist
    to = cdist(lc, sc, "cosine").argmin(0)
    out = T.copy()
    for s, l in zip(small, to):
        out[T == s] = large[l]
    return np.unique(out, return_inverse=True)[1]


def test_cluster_on_loose_sets_where_the_cpp_rule_moves_labels(G):
    moved = 0
    for seed in range(40):
        X = loose_set(seed)
        lab_ref, K_ref = G.cluster_embeddings(X)
        lab, K = orc.cluster_embeddings(X)
        assert K == K_ref and np.array_equal(lab, lab_ref), "seed %d" % seed
        py = python_rule_labels(X)
        if py is not None and not np.array_equal(py, lab_ref):
            moved += 1
    assert moved >= 3, "the sets must exercise App. B #6: only %d of 40 answers differ from the Python's rule" % moved


def test_clustering_with_nan_rows_and_assignment(G):
    """Cluster::clustering (sd.cpp:2063-2116): filter (NaN test on element 0 only, App. B #9), cluster, centroids of raw rows,
    cosine cdist of ALL rows, argmax of 2 - d (NaN row -> 0)"""
    for seed in (3, 11, 19):
        X = loose_set(seed)
        rng = np.random.default_rng(seed)
        c = -(-len(X) // 3) + 6
        emb = np.full((c * 3, 192), np.nan)
        slots = rng.permutation(c * 3)[:len(X)]
        emb[np.sort(slots)] = X
        emb = emb.reshape(c, 3, 192)
        hard_ref = G.clustering(emb)
        hard, K, _ = orc.clustering(emb)
        assert np.array_equal(hard, hard_ref) and K == hard_ref.max() + 1 and K >= 2
        assert (hard_ref[np.isnan(emb[:, :, 0])] == 0).all()
    # fewer than two embeddings: everything is cluster 0 (sd.cpp:2082-2090)
    e1 = np.full((4, 3, 192), np.nan)
    e1[2, 1] = 1.0
    assert np.array_equal(G.clustering(e1), orc.clustering(e1)[0]) and (G.clustering(e1) == 0).all()


# ---------------------------------------------------------------------------------------- a15-a17: reconstruct, annotate
def random_job(seed, c, K=4, tail=0):
    rng = np.random.default_rng(seed)
    seg = rng_scores(rng, c)
    b = orc.binarize(seg)
    hard = rng.integers(0, K, (c, 3)).astype(np.int32)
    same = rng.random(c) < 0.3
    hard[same, 1] = hard[same, 0]                                                        # two local speakers in one cluster: max
    hard = orc.mark_inactive(b, hard)
    n = (c - 1) * 8000 + 80000 + tail
    return seg, b, hard, n


@pytest.mark.parametrize("c,tail", [(1, 0), (3, 17), (12, 4000), (109, 0), (350, 7999)])
def test_reconstruct_and_to_annotation(G, c, tail):
    seg, b, hard, n = random_job(40 + c, c, tail=tail)
    assert np.array_equal(G.mark_inactive(b, np.abs(hard)), orc.mark_inactive(b, np.abs(hard)))
    cnt_ref, win_ref = G.speaker_count(b, n)
    cnt, win, ft = orc.speaker_count(b)
    bin_ref, fr = G.reconstruct(seg, hard, cnt_ref, win_ref, n)
    binr, st = orc.reconstruct(seg, hard, cnt, win, ft, n)
    assert binr.shape == bin_ref.shape and np.array_equal(binr, bin_ref)
    assert st == fr[0] and fr[1] == fr[2] == orc.FRAME_STEP
    if c >= 12:
        assert (binr.sum(1) >= 2).any()                                                  # top-count selection among several clusters
    t_ref = G.to_annotation(bin_ref, fr[0])
    t = orc.to_annotation(binr, st)
    assert t == t_ref and len(t) >= 1                                                    # same turns, same ORDER (std::sort, unstable)


def test_to_annotation_state_machine_and_support(G):
    """onset = offset = 0.5 on {0, 1} rows, gap merging with the float collar (sd.cpp:2852-2935, 911-941)"""
    rng = np.random.default_rng(7)
    for trial in range(30):
        rows, K = int(rng.integers(2, 900)), int(rng.integers(1, 6))
        b = np.zeros((rows, K))
        for k in range(K):
            t = 0
            while t < rows:
                on = int(rng.integers(1, 60))
                off = int(rng.choice([1, 2, 10, 33, 34, 35, 36, 80]))                    # 34.47 frames = the collar
                b[t:t + on, k] = 1.0
                t += on + off
        b[:, K - 1] = b[:, 0] if trial % 3 == 0 else b[:, K - 1]                         # equal-start turns of two labels
        if trial % 5 == 0:
            b[-1, 0] = 1.0
            b[-2, 0] = 0.0                                                               # a run that starts on the last row: kept by the C++
        st = float(np.float32(rng.uniform(0, 3)))
        t_ref = G.to_annotation(b, st)
        assert orc.to_annotation(b, st) == t_ref
        if trial % 3 == 0 and K > 1:
            starts = [x[0] for x in t_ref]
            assert len(starts) > len(set(starts))                                        # equal-start turns: the ORDER std::sort leaves is tested
    segs = sorted((float(a), float(a + d)) for a, d in zip(rng.uniform(0, 50, 200), rng.uniform(0.01, 1.2, 200)))
    assert orc.support(segs) == G.support(segs)
    assert orc.support(segs, 0.0) == G.support(segs, 0.0)


# ------------------------------------------------------------------------------------------------------------ end to end
def planted(seconds, seed=1234):
    pcm = synth.make_pcm(seconds, seed)
    n = len(pcm)
    turns = synth.with_duets(synth.schedule(seconds, seed))
    nc = synth.num_chunks(n)
    scores, assign = synth.planted_scores(turns, n, 0, nc)
    emb = synth.planted_embeddings(assign, outlier_every=211)
    b = orc.binarize(scores)
    masks = orc.select_masks(b)
    per_frame = np.bincount((np.arange(80000, dtype=np.int64) * 293) // 80000, minlength=293)
    counts = ((masks > 0.5) * per_frame[None, :]).sum(1).astype(np.int64)
    bad = np.zeros(len(counts), bool)
    for b0 in range(0, len(counts), 32):
        _, ts, an = orc.wav_lens(counts[b0:b0 + 32])
        bad[b0:b0 + 32] = ts | an
    e32 = emb.copy()
    e32[bad] = np.nan
    return pcm, scores, e32, n, nc


def test_reference_finalize_equals_oracle_on_the_planted_10_min(G):
    """everything behind the two networks, run by the reference's own compiled code in speakerDiarization()'s order"""
    from oracle import pipeline_oracle
    pcm, scores, e32, n, nc = planted(600.0)
    t_ref, K_ref = G.finalize(scores, e32.astype(np.float64), n)
    t, info = pipeline_oracle.diarize_ref(pcm, None, None, seg_override=scores, emb_override=e32.astype(np.float64), return_all=True)
    assert t == t_ref and K_ref == info["K"] == 4 and len(t) > 100
    # and on a job whose embeddings are NOT well separated (loose sets above, shuffled over the items)
    for seed in (5, 23):
        X = loose_set(seed)
        c = 70
        seg = rng_scores(np.random.default_rng(seed), c, onset_hits=False)
        b = orc.binarize(seg)
        e = np.full((c * 3, 192), np.nan)
        live = np.flatnonzero((b.sum(1) > 0).reshape(-1))                                    # items (chunk, speaker) with activity
        e[live[:len(X)]] = X[:len(live)]
        n2 = (c - 1) * 8000 + 80000
        t_ref, K_ref = G.finalize(seg, e, n2)
        hard, K, _ = orc.clustering(e.reshape(c, 3, 192))
        hard = orc.mark_inactive(b, hard)
        cnt, win, ft = orc.speaker_count(b)
        binr, st = orc.reconstruct(seg, hard, cnt, win, ft, n2)
        assert orc.to_annotation(binr, st) == t_ref and K == K_ref >= 2


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("seconds", [600.0, 3600.0])
def test_hip_finalize_gives_the_reference_cpp_turns(diarizer, G, seconds):
    """sd_finalize_dev at configs[1] / configs[2] size against the reference's OWN Cluster::clustering -> reconstruct ->
    to_diarization -> to_annotation -> finalResult (43 s of host time at 1 h; the HIP path takes 80 ms)"""
    import torch
    pcm, scores, e32, n, nc = planted(seconds)
    dev = torch.device("cuda", 0)
    d_seg, d_emb = torch.from_numpy(scores).to(dev), torch.from_numpy(e32).to(dev)
    torch.cuda.synchronize()
    turns = diarizer.finalize_dev(d_seg.data_ptr(), d_emb.data_ptr(), nc, n)
    t_ref, K_ref = G.finalize(scores, e32.astype(np.float64), n)
    assert turns == t_ref and K_ref == 4 and len(turns) >= (100 if seconds < 1000 else 600)


@pytest.mark.gpu
def test_hip_stage_entry_points_against_the_reference_cpp(diarizer, G):
    """sd_postseg (a4-a6), sd_clustering_ex (a10-a14) on loose sets, sd_reconstruct (a15-a17) against the compiled reference"""
    rng = np.random.default_rng(77)
    c = 120
    seg = rng_scores(rng, c)
    n = (c - 1) * 8000 + 80000 + 555
    nb, masks, count = diarizer.postseg(seg)
    b_ref = G.binarize(seg)
    assert np.array_equal(nb.astype(np.float64), b_ref)
    assert np.array_equal(masks, G.select_masks(b_ref))
    cnt_ref, win_ref = G.speaker_count(b_ref, n)
    assert np.array_equal(count, cnt_ref)
    moved = 0
    for seed in range(12):
        X = loose_set(seed)
        cc = -(-len(X) // 3)
        emb = np.full((cc * 3, 192), np.nan)
        emb[:len(X)] = X
        hard, K = diarizer.clustering(emb.reshape(cc, 3, 192))
        hard_ref = G.clustering(emb.reshape(cc, 3, 192))
        assert np.array_equal(hard, hard_ref) and K == hard_ref.max() + 1
        py = python_rule_labels(X)
        moved += int(py is not None and not np.array_equal(py, G.cluster_embeddings(X)[0]))
    assert moved >= 1
    seg, b, hard, n = random_job(9, 150, tail=321)
    cnt_ref, win_ref = G.speaker_count(b, n)
    bin_ref, fr = G.reconstruct(seg, hard, cnt_ref, win_ref, n)
    t_ref = G.to_annotation(bin_ref, fr[0])
    assert diarizer.reconstruct(seg, b.astype(np.uint8), hard, cnt_ref, n) == t_ref and len(t_ref) > 20
