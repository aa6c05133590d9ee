"""CPU tests that pin the oracle (oracle/sd_oracle.c, oracle/nn_oracle.py) against the reference's own
fixtures, the reference's clustering.cpp built in place (oracle/_ref) and scipy / numpy / torch."""
import os

import numpy as np
import pytest

from oracle import orc


def test_closest_frame_fixture(golden_dir):
    # reference fixture pipeline/src/test/closest_frame.txt + test() sd.cpp:3236-3277
    rows = [ln.strip().split(",") for ln in open(os.path.join(golden_dir, "closest_frame.txt")) if ln.strip()]
    assert len(rows) == 10000
    t = 0.0
    for f, ts in rows:
        assert orc.closest_frame(t) == int(f)
        assert abs(float(ts) - t) < 1e-3
        t += 0.5
    # the fixture is np.rint of the clamped expression
    i = np.arange(10000)
    exp = np.rint(np.maximum((0.5 * i - 0.0084375) / 0.016875, 0)).astype(int)
    assert [int(r[0]) for r in rows] == exp.tolist()


def test_np_rint_half_even():
    for v, e in [(0.5, 0), (1.5, 2), (2.5, 2), (3.5, 4), (-0.5, 0), (-1.5, -2), (-2.5, -2), (1.2, 1), (3.6, 4), (0.0, 0)]:
        assert orc.np_rint(v) == e


def test_chunk_rule():
    # sd.cpp:1419 (strict <) and 1457 (i+1<n); SURVEY 8 table
    assert orc.num_chunks(944000) == (109, 80000)
    assert orc.num_chunks(9600000) == (1191, 80000)
    assert orc.num_chunks(57600000) == (7191, 80000)
    assert orc.num_chunks(80000) == (1, 80000)
    assert orc.num_chunks(80001) == (2, 72001)
    assert orc.num_chunks(100000) == (4, 76000)
    assert orc.num_chunks(1) == (0, 0)
    assert orc.num_chunks(2) == (1, 2)


def _blobs(rng, N, d=192, k=4, s=0.6):
    cen = rng.standard_normal((k, d))
    X = cen[rng.integers(0, k, N)] + s * rng.standard_normal((N, d))
    return X / np.linalg.norm(X, axis=1, keepdims=True)


@pytest.mark.parametrize("N", [2, 3, 12, 300, 1200])
def test_linkage_vs_scipy_and_reference(N):
    from scipy.cluster.hierarchy import fcluster, linkage
    rng = np.random.default_rng(N)
    X = _blobs(rng, N)
    T, Z = orc.ahc(X, orc.THRESH_F32)
    Zs = linkage(X, "centroid", "euclidean")
    assert np.array_equal(Z, Zs)                       # bit-identical dendrogram
    assert np.array_equal(T, fcluster(Zs, orc.THRESH_F32, "distance"))
    R = orc.ref()
    if R is not None:                                  # the reference's own clustering.cpp (oracle/_ref)
        Zr = np.zeros((N - 1, 4))
        R.ref_linkage(X, N, X.shape[1], Zr)
        Tr = np.zeros(N, np.int32)
        R.ref_cluster(X, N, X.shape[1], orc.THRESH_F32, Tr)
        assert np.array_equal(Z, Zr)
        assert np.array_equal(T, Tr)


def test_toy_points_of_reference_driver():
    # pipeline/src/clustering/cluster.cpp:8-13, cutoff 1.1 (answer minted with scipy)
    from scipy.cluster.hierarchy import fcluster, linkage
    P = np.array([[0, 0], [0, 1], [1, 0], [0, 4], [0, 3], [1, 4], [4, 0], [3, 0], [4, 1], [4, 4], [3, 4], [4, 3]], float)
    T, _ = orc.ahc(P, 1.1)
    assert T.tolist() == fcluster(linkage(P, "centroid"), 1.1, "distance").tolist() == [5, 5, 6, 7, 7, 8, 1, 1, 2, 3, 3, 4]


def test_linkage_with_duplicates_matches_reference_heap_order():
    # exact ties (duplicate rows): the oracle keeps the reference's heap, so it must still match scipy
    from scipy.cluster.hierarchy import linkage
    rng = np.random.default_rng(5)
    X = _blobs(rng, 40, d=8)
    X[7] = X[3]; X[19] = X[3]; X[30] = X[11]
    _, Z = orc.ahc(X, 0.5)
    assert np.array_equal(Z, linkage(X, "centroid", "euclidean"))


def test_tie_heavy_sets_match_the_reference_build():
    """the sets the GPU tie tests use (tests/test_gpu_parity.py::_tie_sets), pinned on the reference's own clustering.cpp:
    lattice points (almost every merge is an exact tie) and duplicated rows"""
    R = orc.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (no /root/reference)")
    rng = np.random.default_rng(9)
    g = np.stack(np.meshgrid(np.arange(8.0), np.arange(8.0), np.arange(8.0)), -1).reshape(-1, 3)
    Y = _blobs(rng, 900, d=16)
    Y[rng.integers(0, 900, 120)] = Y[rng.integers(0, 900, 120)]
    for X in (g[rng.permutation(len(g))], Y):
        X = np.ascontiguousarray(X)
        N = len(X)
        T, Z = orc.ahc(X, orc.THRESH_F32)
        Zr = np.zeros((N - 1, 4))
        R.ref_linkage(X, N, X.shape[1], Zr)
        Tr = np.zeros(N, np.int32)
        R.ref_cluster(X, N, X.shape[1], orc.THRESH_F32, Tr)
        assert np.array_equal(Z, Zr) and np.array_equal(T, Tr)
        assert len(np.unique(Z[:, 2])) < N - 1                      # the set really has tied merge heights


def test_binarize_semantics():
    rng = np.random.default_rng(1)
    seg = rng.random((7, 293, 3)).astype(np.float32)
    b = orc.binarize(seg)
    assert np.array_equal(b, (seg.astype(np.float64) > orc.ONSET).astype(np.float64))
    # exact-onset frames copy the previous decided frame (sd.cpp:1595, 693-702): exercise with onset=0.5
    s2 = np.full((1, 6, 1), 0.5, np.float32)
    s2[0, 1, 0] = 0.9
    s2[0, 4, 0] = 0.1
    assert orc.binarize(s2, onset=0.5)[0, :, 0].tolist() == [0, 1, 1, 1, 0, 0]
    assert orc.binarize(s2, onset=0.5, initial_state=True)[0, :, 0].tolist() == [1, 1, 1, 1, 0, 0]
    # no float32 value is within DBL_EPSILON of the reference onset => for f32 scores binarize == (score > onset)
    near = np.float32(orc.ONSET)
    assert abs(float(near) - orc.ONSET) > 1e-12


def test_aggregate_against_numpy():
    rng = np.random.default_rng(2)
    c, F, K = 9, 293, 2
    sc = rng.random((c, F, K))
    sc[3, 100:150, 1] = np.nan
    out = orc.aggregate(sc, 0.0, 0.5, 5.0, missing=0.0, skip_average=False)
    nf = orc.closest_frame(5.0 + (c - 1) * 0.5) + 1
    assert out.shape == (nf, K)
    acc = np.zeros((nf, K)); cnt = np.zeros((nf, K))
    for i in range(c):
        s0 = orc.closest_frame(0.5 * i)
        v = sc[i]
        m = ~np.isnan(v)
        acc[s0:s0 + F] += np.where(m, v, 0.0)
        cnt[s0:s0 + F] += m
    exp = np.where(cnt > 0, acc / np.maximum(cnt, np.finfo(float).eps), 0.0)
    assert np.allclose(out, exp, rtol=0, atol=1e-12)


def test_speaker_count_geometry():
    rng = np.random.default_rng(3)
    b = (rng.random((109, 293, 3)) > 0.5).astype(np.float64)
    cnt, win, ft = orc.speaker_count(b)
    assert ft == 235 and win[0] == 0.5 and abs(win[1] - 0.016875) < 1e-15
    assert cnt.min() >= 0 and cnt.max() <= 3
    # constant input -> constant count
    cnt1, _, _ = orc.speaker_count(np.ones((20, 293, 3)))
    assert set(cnt1.tolist()) <= {0, 3} and (cnt1 == 3).sum() > 0.9 * len(cnt1)   # uncovered tail frames -> missing = 0


def test_masks_and_compaction():
    rng = np.random.default_rng(4)
    b = (rng.random((5, 293, 3)) > 0.6).astype(np.float64)
    b[1, :, 0] = 1; b[1, :, 1] = 1          # permanent overlap: clean mask of speakers 0/1 is empty -> full mask used
    m = orc.select_masks(b)
    assert m.shape == (15, 293)
    assert np.array_equal(m[3], b[1, :, 0].astype(np.float32))
    tot = b.sum(-1)
    clean0 = np.where(tot[0] < 2, b[0, :, 0], 0)
    exp0 = clean0 if clean0.sum() > 3 else b[0, :, 0]
    assert np.array_equal(m[0], exp0.astype(np.float32))
    chunk = rng.standard_normal(80000).astype(np.float32)
    sig, n = orc.mask_compact(chunk, m[0])
    idx = (np.arange(80000) * 293 // 80000)
    sel = m[0][idx] > 0.5
    assert n == sel.sum()
    assert np.array_equal(sig[:n], chunk[sel]) and not sig[n:].any()
    lens, ts, allnan = orc.wav_lens(np.array([80000, 40000, 639, 640], np.int64))
    assert lens.tolist() == [1.0, 0.5, 1.0, np.float32(640 / 80000)] and ts.tolist() == [False, False, True, False] and not allnan
    assert orc.wav_lens(np.array([100, 639], np.int64))[2]


def test_clustering_small_to_large_and_nan_rows():
    rng = np.random.default_rng(6)
    c = 120
    cen = rng.standard_normal((3, 192)) * 2
    lab = rng.integers(0, 3, (c, 3))
    emb = cen[lab] + 0.4 * rng.standard_normal((c, 3, 192))
    emb[rng.random((c, 3)) < 0.15] = np.nan
    hard, K, tl = orc.clustering(emb)
    assert K == 3 and hard.min() == 0 and hard.max() == 2
    ok = ~np.isnan(emb[:, :, 0])
    # same partition as the generating labels
    for k in range(3):
        assert len(set(hard[ok & (lab == k)].tolist())) == 1
    assert (hard[~ok] == 0).all()                      # NaN rows argmax to 0 (sd.cpp:293-316)
    # < 2 embeddings -> all zeros (sd.cpp:2081)
    e1 = np.full((4, 3, 192), np.nan); e1[2, 1] = 1.0
    h1, K1, _ = orc.clustering(e1)
    assert K1 == 0 and not h1.any()


def test_to_annotation_and_format():
    b = np.zeros((200, 2))
    b[10:50, 0] = 1
    b[60:90, 0] = 1          # gap of 10 frames = 0.169 s < min_duration_off -> merged
    b[120:130, 0] = 1        # gap of 30 frames = 0.506 s < 0.5817 -> merged too
    b[0:5, 1] = 1
    b[190:200, 1] = 1        # active until the end
    st = float(np.float32(29 * 0.016875))
    turns = orc.to_annotation(b, st)
    assert len(turns) == 3
    mid = lambda i: st + i * 0.016875 + 0.016875 / 2
    t0 = [t for t in turns if t[2] == 0][0]
    assert abs(t0[0] - mid(10)) < 1e-12 and abs(t0[1] - mid(130)) < 1e-12
    assert turns == sorted(turns, key=lambda t: t[0])
    assert orc.format_turn((5.222812345, 17.74406789, 3)) == "[5.22281 -- 17.7441] --> Speaker_3"   # README.md:44


def test_wav_reader(golden_dir, tmp_path):
    import struct
    pcm = (np.arange(-50, 50) * 300).astype(np.int16)
    # header with an extra LIST chunk before data (wav.h:82-90)
    body = b"LIST" + struct.pack("<I", 4) + b"abcd" + b"data" + struct.pack("<I", pcm.nbytes) + pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16)
    p = tmp_path / "t.wav"
    p.write_bytes(hdr + body)
    w, sr, ch, bits = orc.read_wav(str(p))
    assert sr == 16000 and ch == 1 and bits == 16
    assert np.array_equal(w, pcm.astype(np.float32) / 32768.0)
    w1, sr1, _, _ = orc.read_wav(os.path.join(golden_dir, "multi-speaker_1min.wav"))
    assert len(w1) == 944000 and sr1 == 16000


def _write_wav(path, samples, bits, channels=1, extra_chunk=False, fmt_extra=0):
    import struct
    dt = {8: np.int8, 16: np.int16, 32: np.int32}[bits]
    raw = np.asarray(samples, dt).tobytes()
    body = b""
    if extra_chunk:
        body += b"LIST" + struct.pack("<I", 6) + b"abcdef"
    body += b"data" + struct.pack("<I", len(raw)) + raw
    fmt = struct.pack("<HHIIHH", 1, channels, 16000, 16000 * channels * bits // 8, channels * bits // 8, bits) + b"\0" * fmt_extra
    hdr = b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + len(body)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    with open(path, "wb") as f:
        f.write(hdr + body)


@pytest.mark.parametrize("bits,channels,extra,fmt_extra", [(16, 1, False, 0), (16, 1, True, 0), (16, 2, False, 0), (8, 1, False, 0),
                                                           (32, 1, True, 0), (16, 1, False, 2), (16, 1, True, 2)])
def test_wav_readers_against_the_reference_reader(tmp_path, bits, channels, extra, fmt_extra):
    """a1 pinned on the reference's OWN WavReader (frontend/wav.h compiled in place, oracle/_ref/libref_wav.so): bit depths
    8 / 16 / 32, an extra sub-chunk before "data", a fmt chunk longer than 16 bytes, interleaved stereo read as one stream
    (wav.h:95-97).  The oracle's and the library's readers must return exactly data()/32768 (sd.cpp:2948-2951)."""
    rng = np.random.default_rng(bits + 7 * channels + extra)
    lim = {8: 127, 16: 32767, 32: 2 ** 31 - 1}[bits]
    smp = rng.integers(-lim - 1, lim + 1, 1000 * channels)
    p = str(tmp_path / "r.wav")
    _write_wav(p, smp, bits, channels, extra, fmt_extra)
    ref = orc.ref_wav_read(p)
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    raw, sr, ch, b = ref
    assert (sr, ch, b) == (16000, channels, bits) and len(raw) == len(smp)
    assert np.array_equal(raw, smp.astype(np.float32))                  # what WavReader::data() holds
    w, sr2, ch2, b2 = orc.read_wav(p)
    assert (sr2, ch2, b2) == (sr, ch, b)
    import sdhip
    wl, srl, chl, bl = sdhip.read_wav_f32(p)                            # host-only entry point of the library
    assert (srl, chl, bl) == (sr, ch, b)
    expect = raw / np.float32(32768.0)
    # the oracle returns num_samples (per channel) values, the library all interleaved values: compare the common prefix
    assert np.array_equal(w, expect[:len(w)]) and np.array_equal(wl[:len(expect)], expect[:len(wl)])


def test_stft_oracle_against_direct_dft():
    import torch
    from oracle import nn_oracle as nn
    rng = np.random.default_rng(7)
    x = rng.standard_normal((1, 80000)).astype(np.float32)
    st = nn.stft_ref(x).numpy()[0]                   # [501,201,2]
    assert st.shape == (501, 201, 2)
    win = torch.hamming_window(400).numpy().astype(np.float64)
    xp = np.concatenate([np.zeros(200), x[0].astype(np.float64), np.zeros(200)])
    for t in (0, 1, 250, 500):
        fr = xp[160 * t:160 * t + 400] * win
        k = np.arange(201)[:, None] * np.arange(400)[None, :]
        re = (fr[None, :] * np.cos(2 * np.pi * k / 400)).sum(1)
        im = -(fr[None, :] * np.sin(2 * np.pi * k / 400)).sum(1)
        assert np.allclose(st[t, :, 0], re, atol=2e-5) and np.allclose(st[t, :, 1], im, atol=2e-5)


def test_nn_oracle_shapes_and_param_counts():
    from oracle import nn_oracle as nn
    ws, we = nn.synth_segmentation_weights(), nn.synth_embedding_weights()
    assert abs(sum(v.size for v in ws.values()) - 1.47e6) < 0.05e6
    ecapa = sum(v.size for k, v in we.items() if not k.startswith(("fbank", "stft")))
    assert abs(ecapa - 20.8e6) < 0.1e6
    y = nn.PyanNetOracle(ws)(np.zeros((1, 80000), np.float32))
    assert tuple(y.shape) == (1, 293, 3)                                    # sd.cpp:1350-1351
    y2 = nn.PyanNetOracle(ws)(np.zeros((1, 43000), np.float32))
    assert y2.shape[1] == ((((43000 - 251) // 10 + 1) // 3 - 4) // 3 - 4) // 3
