"""SURVEY 8(f) "next" rows: f2 wav reader robustness (8/16/32-bit, extra RIFF chunks), f3 constrained
clustering (the branch the reference asserts on; spec = clustering/Clustering.py:21-43, 352-399), f4 RTTM."""
import os
import struct

import numpy as np
import pytest

import sdhip
from oracle import orc


def _wav_bytes(samples, bits, channels=1, extra_chunk=True, fmt_extra=0):
    if bits == 8:
        data = samples.astype(np.int8).tobytes()
    elif bits == 16:
        data = samples.astype(np.int16).tobytes()
    else:
        data = samples.astype(np.int32).tobytes()
    body = b""
    if extra_chunk:
        body += b"LIST" + struct.pack("<I", 6) + b"abcdef"
    body += b"data" + struct.pack("<I", len(data)) + data
    fmt = struct.pack("<HHIIHH", 1, channels, 16000, 16000 * channels * bits // 8, channels * bits // 8, bits) + b"\0" * fmt_extra
    return b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<I", 16 + fmt_extra) + fmt + body


@pytest.mark.parametrize("bits,fmt_extra", [(8, 0), (16, 0), (32, 0), (16, 2), (32, 24)])
def test_wav_reader_bit_depths_match_oracle(tmp_path, bits, fmt_extra):
    rng = np.random.default_rng(bits)
    hi = {8: 127, 16: 32767, 32: 2 ** 31 - 1}[bits]
    s = rng.integers(-hi, hi, 1000)
    p = tmp_path / "t.wav"
    p.write_bytes(_wav_bytes(s, bits, fmt_extra=fmt_extra))
    w, sr, ch, b = sdhip.read_wav_f32(str(p))
    w_ref, sr_ref, ch_ref, b_ref = orc.read_wav(str(p))                 # oracle restatement of wav.h:62-126
    assert (sr, ch, b) == (sr_ref, ch_ref, b_ref) == (16000, 1, bits)
    assert np.array_equal(w, w_ref)
    assert np.array_equal(w, (s.astype(np.float32) * np.float32(1.0) / 32768.0).astype(np.float32))   # sd.cpp:2950


def test_wav_reader_multichannel_is_read_interleaved_like_the_reference(tmp_path):
    s = np.arange(200) * 100                         # 2 channels interleaved: wav.h:95-97 keeps num_data / channels samples
    p = tmp_path / "st.wav"
    p.write_bytes(_wav_bytes(s, 16, channels=2))
    w, sr, ch, b = sdhip.read_wav_f32(str(p))
    w_ref, _, _, _ = orc.read_wav(str(p))
    assert ch == 2 and len(w) == 100 and np.array_equal(w, w_ref)


def test_wav_reader_errors(tmp_path):
    p = tmp_path / "bad.wav"
    p.write_bytes(_wav_bytes(np.zeros(10), 16)[:30])
    with pytest.raises(sdhip.SdError):
        sdhip.read_wav_f32(str(p))
    with pytest.raises(sdhip.SdError):
        sdhip.read_wav_f32(str(tmp_path / "nope.wav"))


def test_rttm_writer(tmp_path):
    turns = [(5.222812345, 17.74406789, 3), (17.8116, 25.2197, 0)]
    p = tmp_path / "o.rttm"
    sdhip.write_rttm(str(p), "multi-speaker_1min", turns)
    lines = p.read_text().splitlines()
    assert lines[0] == "SPEAKER multi-speaker_1min 1 5.223 12.521 <NA> <NA> SPEAKER_03 <NA> <NA>"
    assert lines[1].split()[3:5] == ["17.812", "7.408"] and lines[1].split()[7] == "SPEAKER_00"


# ------------------------------------------------------------------ f3: constrained clustering
def _py_spec(emb, num_clusters=None, min_clusters=None, max_clusters=None, threshold=orc.THRESH_F32, mcs_cfg=15):
    """scipy transcription of clustering/Clustering.py:21-43 + 278-428 (with the C++ port's un-normalised
    centroids and float32 L2 norm, SURVEY App. B #6 and sd.cpp:332)"""
    from scipy.cluster.hierarchy import fcluster, linkage
    from scipy.spatial.distance import cdist
    N = len(emb)
    min_clusters = num_clusters or min_clusters or 1
    min_clusters = max(1, min(N, min_clusters))
    max_clusters = num_clusters or max_clusters or N
    max_clusters = max(1, min(N, max_clusters))
    if min_clusters > max_clusters:
        min_clusters = max_clusters
    if min_clusters == max_clusters:
        num_clusters = min_clusters
    mcs = min(mcs_cfg, max(1, round(0.1 * N)))
    nrm = np.sqrt((emb * emb).sum(1)).astype(np.float32).astype(np.float64)
    Z = linkage(emb / nrm[:, None], method="centroid", metric="euclidean")
    clusters = fcluster(Z, threshold, criterion="distance") - 1
    cu, cc = np.unique(clusters, return_counts=True)
    large = cu[cc >= mcs]
    nlarge = len(large)
    if nlarge < min_clusters:
        num_clusters = min_clusters
    elif nlarge > max_clusters:
        num_clusters = max_clusters
    if num_clusters is not None:
        _Z = np.copy(Z)
        _Z[:, 2] = np.arange(N - 1)
        best_it, best_nl = N - 1, 1
        for it in np.argsort(np.abs(Z[:, 2] - threshold), kind="stable"):
            if _Z[it, 3] < mcs:
                continue
            clusters = fcluster(_Z, it, criterion="distance") - 1
            cu, cc = np.unique(clusters, return_counts=True)
            large = cu[cc >= mcs]
            nlarge = len(large)
            if abs(nlarge - num_clusters) < abs(best_nl - num_clusters):
                best_it, best_nl = it, nlarge
            if nlarge == num_clusters:
                break
        if best_nl != num_clusters:
            clusters = fcluster(_Z, best_it, criterion="distance") - 1
            cu, cc = np.unique(clusters, return_counts=True)
            large = cu[cc >= mcs]
            nlarge = len(large)
    if nlarge == 0:
        clusters[:] = 0
        return clusters
    small = cu[cc < mcs]
    if len(small) == 0:
        return clusters
    lc = np.vstack([emb[clusters == k].mean(0) for k in large])
    sc = np.vstack([emb[clusters == k].mean(0) for k in small])
    for sk, lk in enumerate(np.argmin(cdist(lc, sc, metric="cosine"), axis=0)):
        clusters[clusters == small[sk]] = large[lk]
    return np.unique(clusters, return_inverse=True)[1]


CONSTRAINTS = [{}, {"num_clusters": 2}, {"num_clusters": 3}, {"num_clusters": 8}, {"min_clusters": 7}, {"max_clusters": 2},
               {"min_clusters": 2, "max_clusters": 4}, {"num_clusters": 1}]


def _data(seed=0, N=400, k=5, s=0.9):
    rng = np.random.default_rng(seed)
    cen = rng.standard_normal((k, 192)) * 2
    return cen[rng.integers(0, k, N)] + s * rng.standard_normal((N, 192))


@pytest.mark.parametrize("kw", CONSTRAINTS)
def test_oracle_constrained_clustering_follows_python_spec(kw):
    X = _data()
    lab, K = orc.cluster_embeddings_ex(X, **kw)
    if kw.get("num_clusters") == 1 or kw.get("max_clusters") == 1:
        return                                       # max_clusters < 2 short-circuits before cluster() (sd.cpp:2081)
    exp = _py_spec(X.copy(), **kw)
    assert np.array_equal(lab, exp) and K == exp.max() + 1


@pytest.mark.gpu
@pytest.mark.parametrize("kw", CONSTRAINTS)
def test_gpu_constrained_clustering_matches_oracle(diarizer, kw):
    rng = np.random.default_rng(3)
    X = _data(seed=4, N=450)
    emb = X.astype(np.float32).astype(np.float64).reshape(150, 3, 192)
    emb[rng.random((150, 3)) < 0.1] = np.nan
    h, K = diarizer.clustering(emb, **kw)
    h_ref, K_ref, _ = orc.clustering(emb, **kw)
    assert np.array_equal(h, h_ref)


@pytest.mark.gpu
def test_gpu_diarize_f32_equals_int16_path_and_cli_rttm(diarizer, weights, tmp_path):
    import subprocess
    import synth
    pcm = synth.make_pcm(22.0, seed=9)
    p = tmp_path / "a.wav"
    p.write_bytes(_wav_bytes(pcm, 16, extra_chunk=True))
    w, sr, ch, bits = sdhip.read_wav_f32(str(p))
    assert diarizer.diarize_f32(w) == diarizer.diarize(pcm)
    # 32-bit file holding the same values: the reference scales every depth by 1/32768, so results are identical
    p32 = tmp_path / "a32.wav"
    p32.write_bytes(_wav_bytes(pcm.astype(np.int64), 32))
    w32, _, _, b32 = sdhip.read_wav_f32(str(p32))
    assert b32 == 32 and np.array_equal(w32, w)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyannote-audio_speaker-diarization_cpp_amd", "speakerDiarizer")
    rttm = tmp_path / "o.rttm"
    out = subprocess.run([exe, weights[0], weights[1], str(p32), "--rttm", str(rttm)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    turns = diarizer.diarize(pcm)
    assert len(rttm.read_text().splitlines()) == len(turns)
