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


def test_wav_reader_streamed_and_truncated_headers(tmp_path):
    """a wav written to a pipe (ffmpeg -f wav -, sox) cannot seek back to fill in its sizes and announces 0xFFFFFFFF (or 0) for the data
    chunk; a file cut off in transit announces more than it holds.  The reference trusts the header (wav.h:92-97); the library reads what is
    there (deliberate deviation, input robustness) -- and a well-formed file is read exactly as before"""
    s = (np.arange(3000) % 700 - 350).astype(np.int64) * 40
    good = _wav_bytes(s, 16)
    at = good.index(b"data") + 4                                           # the data chunk's size field (an extra LIST chunk sits in front)
    want = (s.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    for name, data, n_expect in (("stream_ff.wav", good[:at] + struct.pack("<I", 0xFFFFFFFF) + good[at + 4:], 3000),
                                 ("stream_0.wav", good[:at] + struct.pack("<I", 0) + good[at + 4:], 3000),
                                 ("cut.wav", good[:at + 4 + 2 * 1234], 1234),
                                 ("good.wav", good, 3000)):
        p = tmp_path / name
        p.write_bytes(data)
        w, sr, ch, b = sdhip.read_wav_f32(str(p))
        assert (sr, ch, b, len(w)) == (16000, 1, 16, n_expect), name
        assert np.array_equal(w, want[:n_expect]), name
        pcm, sr2, ch2 = sdhip.read_wav(str(p))
        assert len(pcm) == n_expect and np.array_equal(pcm, s[:n_expect].astype(np.int16)), name


def test_wav_reader_empty_data_chunk_followed_by_metadata_has_no_samples(tmp_path):
    """a well-formed file whose data chunk is genuinely EMPTY and is followed by LIST / id3 sub-chunks: announced size 0 is then not the
    stream writers' "unknown" -- the metadata must not be decoded as audio (round-3 advisor); the reference's own reader says 0 samples"""
    head = _wav_bytes(np.zeros(0), 16, extra_chunk=False)                   # ... "data" + size 0
    assert head.endswith(b"data" + struct.pack("<I", 0))
    tail = b"LIST" + struct.pack("<I", 11) + b"INFOabcdefg" + b"\0" + b"id3 " + struct.pack("<I", 4) + b"wxyz"
    p = tmp_path / "empty_then_list.wav"
    p.write_bytes(head + tail)
    w, sr, ch, b = sdhip.read_wav_f32(str(p))
    assert len(w) == 0 and (sr, ch, b) == (16000, 1, 16)
    assert len(sdhip.read_wav(str(p))[0]) == 0
    r = orc.ref_wav_read(str(p))
    if r is not None:
        assert len(r[0]) == 0
    # the same size field in front of raw samples (a stream writer's file) is still read to the end
    s = (np.arange(500) * 37 % 9000 - 4500).astype(np.int64)
    p2 = tmp_path / "stream_0.wav"
    p2.write_bytes(head + s.astype(np.int16).tobytes())
    assert np.array_equal(sdhip.read_wav(str(p2))[0], s.astype(np.int16))


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


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [8, 32])
def test_gpu_diarize_f32_on_8_and_32_bit_files_against_the_oracle(diarizer, weights, tmp_path, bits):
    """f2 on the GPU against the ORACLE (not against sd_diarize): an 8-bit and a genuinely 32-bit file are read by the REFERENCE's own
    WavReader (oracle/_ref/libref_wav.so; the C restatement where _ref is absent), scaled as sd.cpp:2950 does, and fed to
    pipeline_oracle; sd_read_wav_f32 must deliver the same samples and sd_diarize_f32 the oracle's turns -- bit for bit behind the
    networks, within +-1 frame against the oracle's own torch networks (north star)"""
    import synth
    from oracle import pipeline_oracle
    pcm = synth.make_pcm(24.0, seed=13)
    if bits == 8:
        s = (pcm.astype(np.int64) >> 8).clip(-127, 127)                     # signed char, as wav.h:103-106 reads it
    else:
        s = pcm.astype(np.int64) * 37 + 11                                  # beyond the int16 range: +-1.2e6 -> samples up to +-37 after / 32768
    p = tmp_path / ("a%d.wav" % bits)
    p.write_bytes(_wav_bytes(s, bits, extra_chunk=True, fmt_extra=2 if bits == 32 else 0))
    r = orc.ref_wav_read(str(p))
    raw = r[0] if r is not None else None
    w_ref = (raw * np.float32(1.0) / np.float32(32768.0)).astype(np.float32) if raw is not None else orc.read_wav(str(p))[0]
    w, sr, ch, b = sdhip.read_wav_f32(str(p))
    assert b == bits and np.array_equal(w, w_ref) and np.abs(w).max() > (5 if bits == 32 else 0.0005)
    turns = diarizer.diarize_f32(w)
    seg = diarizer.segment(w_ref)
    masks = orc.select_masks(orc.binarize(seg))
    emb = diarizer.embed(w_ref, masks)
    t1 = pipeline_oracle.diarize_ref(None, weights[2], weights[3], seg_override=seg, emb_override=emb, wav=w_ref)
    assert turns == t1 and len(turns) >= 1
    t2 = pipeline_oracle.diarize_ref(None, weights[2], weights[3], wav=w_ref)
    assert len(t2) == len(turns)
    key = lambda t: (t[2], t[0])
    for a, c in zip(sorted(turns, key=key), sorted(t2, key=key)):
        assert a[2] == c[2] and abs(a[0] - c[0]) <= 0.016875 + 1e-9 and abs(a[1] - c[1]) <= 0.016875 + 1e-9
    assert diarizer.diarize_wav(p) == turns                                 # the file entry point reads the same samples


# ------------------------------------------------------------------ f4: relabelling, confidence, RTTM confidence column
def test_relabel_conventions():
    turns = [(0.5, 2.0, 10), (1.0, 3.0, 2), (3.5, 4.0, 2), (4.0, 9.0, 0), (9.5, 10.0, 10)]
    # pyannote: Annotation.labels() sorts the labels that occur by their STRING ("0" < "10" < "2") -> SPEAKER_00, 01, 02
    assert [t[2] for t in sdhip.relabel_turns(turns, "pyannote")] == [1, 2, 2, 0, 1]
    assert [t[2] for t in sdhip.relabel_turns(turns, "first")] == [0, 1, 1, 2, 0]
    assert sdhip.relabel_turns([], "first") == []
    r = sdhip.relabel_turns(turns, "first")
    assert [(a[0], a[1]) for a in r] == [(a[0], a[1]) for a in turns]


def test_rttm_confidence_column(tmp_path):
    turns = [(5.222812345, 17.74406789, 3), (17.8116, 25.2197, 0)]
    p = tmp_path / "c.rttm"
    sdhip.write_rttm(str(p), "rec", turns, conf=[1.87654321, float("nan")])
    lines = p.read_text().splitlines()
    assert lines[0] == "SPEAKER rec 1 5.223 12.521 <NA> <NA> SPEAKER_03 <NA> 1.8765"
    assert lines[1].endswith("SPEAKER_00 <NA> <NA>")


def test_cli_multi_gpu_launcher_fails_cleanly_without_devices(golden_dir):
    """speakerDiarizer --gpus 2: the launcher forks one process per GPU before anything touches HIP and hands the RCCL
    rendezvous id over through pipes.  Without (enough) GPUs every rank must say why and the launcher must return 1
    promptly -- no rank may block on a pipe whose writer died."""
    import subprocess
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyannote-audio_speaker-diarization_cpp_amd", "speakerDiarizer")
    out = subprocess.run([exe, "--gpus", "2", "/nonexistent/seg.onnx", "/nonexistent/emb.onnx", os.path.join(golden_dir, "multi-speaker_1min.wav")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 1
    assert "rank 0" in out.stderr and "rank 1" in out.stderr and "Speaker_" not in out.stdout


@pytest.mark.gpu
def test_cli_multi_gpu_launcher_ends_promptly_when_one_rank_cannot_start(weights, golden_dir):
    """`speakerDiarizer --gpus 2` on a ONE-GPU box: rank 0 gets its GPU, loads the models and the wav; rank 1 has no device.  Without the
    ready handshake rank 0 would mint the rendezvous id and wait in ncclCommInitRank for a peer that never comes: now rank 0
    learns from the closed ready pipe that rank 1 did not come up, and the launcher, reaping in completion order, returns 1 at once."""
    import subprocess
    import time
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one GPU")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyannote-audio_speaker-diarization_cpp_amd", "speakerDiarizer")
    t0 = time.time()
    out = subprocess.run([exe, "--gpus", "2", weights[0], weights[1], os.path.join(golden_dir, "multi-speaker_1min.wav")], capture_output=True, text=True, timeout=120)
    dt = time.time() - t0
    assert out.returncode == 1 and dt < 60, (out.returncode, dt, out.stderr[-500:])
    assert "rank 1: sd_create failed" in out.stderr and "rank 1 did not come up" in out.stderr
    assert "Speaker_" not in out.stdout


def _run_bench(args, timeout, env_extra=None):
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra or {})
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    return out, time.time() - t0


def test_bench_self_launches_n_ranks_and_fails_loudly_when_the_gpus_are_not_there():
    """`python bench.py --gpus 2` without a launcher must not quietly measure one GPU: it starts 2 ranks itself (before any GPU call),
    and on a box with fewer than 2 GPUs every rank says why, no result line is printed, the exit code is non-zero -- well within a minute
    (a rank that waits for a dead peer in the rendezvous is stopped by the launcher)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    out, dt = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--hours-per-gpu", "0.02"], 120)
    assert out.returncode != 0 and dt < 60, (out.returncode, dt)
    assert "{" not in out.stdout                                            # no result line
    assert "rank 1" in out.stderr and "GPU" in out.stderr and "no result line" in out.stderr
    if torch.cuda.device_count() == 0:
        assert "[rank 0]" in out.stderr and "[rank 1]" in out.stderr


def test_bench_refuses_a_world_size_that_is_not_the_one_asked_for():
    out, _ = _run_bench(["--gpus", "2"], 120, {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "refusing" in out.stderr and "{" not in out.stdout
    out, _ = _run_bench(["--gpus", "1"], 120, {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "refusing" in out.stderr and "{" not in out.stdout


@pytest.mark.gpu
def test_bench_gpus_2_on_a_one_gpu_box_fails_within_a_minute():
    """the same on the GPU box: rank 0 gets its GPU and waits in the rendezvous, rank 1 has no device; the launcher ends the job"""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one GPU")
    out, dt = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--hours-per-gpu", "0.02"], 150)
    assert out.returncode != 0 and dt < 60, (out.returncode, dt, out.stderr[-2000:])
    assert "{" not in out.stdout and "[rank 1]" in out.stderr and "LOCAL_RANK 1 but this node has 1 GPU" in out.stderr


# ------------------------------------------------------------------ f3: constrained_argmax (Clustering.py:81-94)
def test_oracle_linear_sum_assignment_is_scipys():
    """the dependency behind constrained_argmax is scipy.optimize.linear_sum_assignment; the oracle restates its algorithm
    (rectangular_lsap.cpp) and must pick the same one of several optimal assignments: ties, constant rows (NaN rows after
    nan_to_num), fewer clusters than speakers (transposed problem)"""
    from scipy.optimize import linear_sum_assignment as lsa
    rng = np.random.default_rng(0)
    for trial in range(2000):
        nc = int(rng.integers(1, 9))
        mode = trial % 4
        if mode == 0:
            cost = rng.standard_normal((3, nc))
        elif mode == 1:
            cost = rng.integers(0, 3, (3, nc)).astype(float)
        elif mode == 2:
            cost = rng.standard_normal((3, nc))
            cost[rng.integers(0, 3)] = cost.min()
            if rng.random() < 0.5:
                cost[rng.integers(0, 3)] = cost.min()
        else:
            cost = np.zeros((3, nc))
        for mx in (True, False):
            r1, c1 = lsa(cost, maximize=mx)
            r2, c2 = orc.linear_sum_assignment(cost, mx)
            assert np.array_equal(r1, r2) and np.array_equal(c1, c2), (trial, mx)


def _py_constrained_argmax(soft):
    from scipy.optimize import linear_sum_assignment as lsa
    s = np.nan_to_num(soft, nan=np.nanmin(soft))                            # Clustering.py:83
    hard = -2 * np.ones(s.shape[:2], np.int32)
    for c, cost in enumerate(s):
        for a, b in zip(*lsa(cost, maximize=True)):
            hard[c, a] = b
    return hard


def _planted_emb(c, k, pnan, seed, s=0.5):
    rng = np.random.default_rng(seed)
    cen = rng.standard_normal((k, 192)) * 2
    emb = (cen[rng.integers(0, k, (c, 3))] + s * rng.standard_normal((c, 3, 192))).astype(np.float32).astype(np.float64)
    emb[rng.random((c, 3)) < pnan] = np.nan
    return emb


@pytest.mark.parametrize("c,k,pnan", [(200, 5, 0.3), (150, 2, 0.2), (90, 3, 0.0), (120, 8, 0.5)])
def test_oracle_constrained_assignment_follows_python_spec(c, k, pnan):
    emb = _planted_emb(c, k, pnan, seed=c + k)
    hard, K, soft = orc.clustering_full(emb, constrained=True)
    assert K == k and np.array_equal(hard, _py_constrained_argmax(soft))
    per_chunk_distinct = all(len({x for x in row if x >= 0}) == sum(x >= 0 for x in row) for row in hard.tolist())
    assert per_chunk_distinct and ((hard == -2).any() == (k < 3))
    h0, K0, _ = orc.clustering(emb)
    assert not np.array_equal(h0, hard)                                     # the plain argmax does put two local speakers in one cluster


@pytest.mark.gpu
@pytest.mark.parametrize("c,k,pnan", [(200, 5, 0.3), (150, 2, 0.2), (90, 3, 0.0), (120, 8, 0.5)])
def test_gpu_constrained_assignment_matches_oracle(diarizer, c, k, pnan):
    emb = _planted_emb(c, k, pnan, seed=c + k)
    diarizer.set_option("constrained_assignment", 1)
    try:
        h, K = diarizer.clustering(emb)
    finally:
        diarizer.set_option("constrained_assignment", 0)
    h_ref, K_ref, _ = orc.clustering_full(emb, constrained=True)
    assert K == K_ref and np.array_equal(h, h_ref)
    assert np.array_equal(diarizer.clustering(emb)[0], orc.clustering(emb)[0])       # and the default is back


@pytest.mark.gpu
def test_gpu_confidence_and_relabelled_rttm(diarizer, tmp_path):
    """per-turn confidence = mean soft score (sd.cpp:2191-2207) of the items of the turn's cluster whose chunk overlaps it"""
    import torch
    import synth
    sec = 300.0
    pcm = synth.make_pcm(sec, 21)
    n = len(pcm)
    nc = synth.num_chunks(n)
    scores, assign = synth.planted_scores(synth.with_duets(synth.schedule(sec, 21)), n, 0, nc)
    emb = synth.planted_embeddings(assign, outlier_every=97)
    from test_planted import nan_rule
    binar, _, _, bad = nan_rule(scores)
    emb[bad] = np.nan
    dev = torch.device("cuda", 0)
    d_seg, d_emb = torch.from_numpy(scores).to(dev), torch.from_numpy(emb).to(dev)
    torch.cuda.synchronize()
    turns = diarizer.finalize_dev(d_seg.data_ptr(), d_emb.data_ptr(), nc, n)
    conf = diarizer.last_confidence()
    assert len(conf) == len(turns) > 20
    hard, K, soft = orc.clustering_full(emb.astype(np.float64).reshape(nc, 3, 192))
    hard = orc.mark_inactive(binar, hard)
    exp = np.full(len(turns), np.nan)
    for t, (s, e, k) in enumerate(turns):
        vals = [soft[c, sp, k] for c in range(nc) for sp in range(3)
                if hard[c, sp] == k and 0.5 * c < e and 0.5 * c + 5.0 > s and not np.isnan(soft[c, sp, k])]
        if vals:
            exp[t] = np.mean(vals)
    assert np.isfinite(exp).all() and (exp > 1.0).all()
    np.testing.assert_allclose(conf, exp, rtol=1e-12, atol=0)
    p = tmp_path / "r.rttm"
    rel = sdhip.relabel_turns(turns, "first")
    sdhip.write_rttm(str(p), "rec", rel, conf=conf)
    lines = p.read_text().splitlines()
    assert len(lines) == len(turns) and lines[0].split()[7] == "SPEAKER_00" and float(lines[0].split()[9]) == round(conf[0], 4)


# ------------------------------------------------------------------ 8e under the boundary: RCCL communicator inside libsdhip
@pytest.mark.gpu
def test_sharded_entry_point_through_rccl_world_of_one(diarizer):
    """sd_comm_unique_id / sd_comm_init / sd_diarize_sharded_dev on one rank: the RCCL all-gather runs on the library's
    stream and the result equals sd_diarize_dev (the N-rank form cannot run on a 1-GPU box: RCCL refuses two ranks on one device)"""
    import torch
    import synth
    pcm = synth.make_pcm(60.0, seed=12)
    n = len(pcm)
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm).to(dev)
    torch.cuda.synchronize()
    whole = diarizer.diarize_dev(d_pcm.data_ptr(), n)
    assert diarizer.comm_info() == (0, 0)
    diarizer.comm_init(sdhip.comm_unique_id(), 0, 1)
    try:
        assert diarizer.comm_info() == (0, 1)
        assert diarizer.diarize_sharded_dev(d_pcm.data_ptr(), 0, n, n) == whole
        assert diarizer.diarize_sharded(pcm, 0, n) == whole
        assert diarizer.kernel_stats("rccl_all_gather")["launches"] >= 2
        # the plan and slot assembly of 2- / 3- / 8-rank jobs (equal shares, reduced and zero rank-0 share), played by this one rank
        for W, pm in [(2, -1), (3, 100), (3, 0), (8, 30)]:
            diarizer.set_option("virtual_world", W)
            diarizer.set_option("rank0_permille", pm)
            try:
                assert diarizer.diarize_sharded_dev(d_pcm.data_ptr(), 0, n, n) == whole, (W, pm)
            finally:
                diarizer.set_option("virtual_world", 0)
                diarizer.set_option("rank0_permille", -1)
        # the same in x3 mode (ecapa_precision = 3): a rank's embeddings do not depend on which rank computed them or in which batch
        diarizer.set_option("ecapa_precision", 3)
        try:
            whole_x3 = diarizer.diarize_dev(d_pcm.data_ptr(), n)
            for W, pm in [(2, -1), (8, 30)]:
                diarizer.set_option("virtual_world", W)
                diarizer.set_option("rank0_permille", pm)
                assert diarizer.diarize_sharded_dev(d_pcm.data_ptr(), 0, n, n) == whole_x3, (W, pm)
        finally:
            diarizer.set_option("virtual_world", 0)
            diarizer.set_option("rank0_permille", -1)
            diarizer.set_option("ecapa_precision", 0)
        with pytest.raises(sdhip.SdError) as e:
            diarizer.diarize_sharded_dev(d_pcm.data_ptr(), 8000, n - 8000, n)       # samples do not cover the rank's chunks
        assert "do not cover" in str(e.value)
        # a failure in one rank's part is collective (comm.cpp): the failing rank still joins the exchange with a status record and
        # every rank returns an error for the job -- here rank 1 of a 3-rank plan "fails" and rank 0 (this process) must report it
        diarizer.set_option("virtual_world", 3)
        diarizer.set_option("inject_fail_rank", 1)
        try:
            with pytest.raises(sdhip.SdError) as e:
                diarizer.diarize_sharded_dev(d_pcm.data_ptr(), 0, n, n)
            assert e.value.code == 1 and "rank 1 of 3 failed" in str(e.value)
        finally:
            diarizer.set_option("virtual_world", 0)
        # ... and the real exchange with the real rank failing: the all-gather still runs, the error is this rank's own
        diarizer.set_option("inject_fail_rank", 0)
        try:
            before = diarizer.kernel_stats("rccl_all_gather")["launches"]
            with pytest.raises(sdhip.SdError) as e:
                diarizer.diarize_sharded_dev(d_pcm.data_ptr(), 0, n, n)
            assert "injected failure" in str(e.value)
            assert diarizer.kernel_stats("rccl_all_gather")["launches"] == before + 1
        finally:
            diarizer.set_option("inject_fail_rank", -1)
        assert diarizer.diarize_sharded_dev(d_pcm.data_ptr(), 0, n, n) == whole     # the communicator of this single rank is still in step
    finally:
        diarizer.comm_destroy()
    assert diarizer.comm_info() == (0, 0)
    with pytest.raises(sdhip.SdError):
        diarizer.diarize_sharded_dev(d_pcm.data_ptr(), 0, n, n)                     # no communicator


@pytest.mark.gpu
def test_bench_line_contract_on_a_short_job():
    """the driver's contract for bench.py, on a 3-minute job: ONE JSON line with metric / value / unit / n_gpus / steps / warmup / ms_per_step /
    higher_is_better / scaling / vs_baseline / dtype / data / config.workload, a `roofline` object for the dominant kernel measured with HIP
    events on the library's stream and a `cpu_baseline` object timed on this box's host cores; plus this round's additions"""
    import json
    out, _ = _run_bench(["--steps", "2", "--warmup", "1", "--hours-per-gpu", "0.05", "--cpu-seconds", "10", "--fp16-steps", "1", "--x3-steps", "1"], 600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.stdout.strip().endswith(lines[0])
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - j["config"]["audio_seconds"] / (j["ms_per_step"] / 1e3)) < 0.01 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0
    assert r["avg_launch_ms"] > 0 and r["launches_per_step"] > 0
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["reference_clustering"] is None or cb["reference_clustering"].get("same_labels") is True          # oracle/_ref travels with the snapshot
    assert j["value_host_pcm"]["same_turns"] is True and j["value_cold"]["ms"] > j["ms_per_step"]
    f = j["fp16"]
    assert f["same_turns_as_f32"] is True and f["roofline"]["peak"] == 2500.0 and f["cosine_distance_to_f32_embeddings"]["same_nan_rows"] is True
    assert f["cosine_distance_to_f32_embeddings"]["q99"] < 2e-3
    x = j["x3"]
    assert x["same_turns_as_f32"] is True and x["roofline"]["peak"] == 2500.0 and x["cosine_distance_to_f32_embeddings"]["same_nan_rows"] is True
    assert x["cosine_distance_to_f32_embeddings"]["max"] < 1e-6 and x["cosine_distance_to_f32_embeddings"]["above_1e-3"] == 0
    assert abs(x["roofline"]["achieved"] - 3 * x["roofline"]["achieved_algorithmic"]) < 0.5 and x["value"] > j["value"]


@pytest.mark.gpu
def test_bench_strong_scaling_leg_through_rccl_world_of_one():
    """`bench.py --gpus N` (N > 1) also times ONE hour sharded over the N ranks (`strong_scaling_reading`).  No multi-GPU box exists here, so the leg
    is run the only way it can be with the real library: `--force-dist` on one GPU -- RCCL communicator of one rank, a SECOND recording length (the
    leg's shorter job) through the same context and communicator, planted outputs re-pointed and restored.  Both jobs must give their known turns."""
    import json
    out, _ = _run_bench(["--force-dist", "--hours-per-gpu", "0.1", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--fp16-steps", "0", "--x3-steps", "0",
                         "--strong-steps", "2"], 900)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    sr = j["strong_scaling_reading"]
    assert j["rccl_ranks"] == 1 and sr["scaling"] == "strong" and sr["steps"] == 2 and sr["value"] > 0
    assert sr["chunks"] == j["config"]["chunks"] and sr["turns"] == j["config"]["turns"] > 0          # (at <= 1 h per GPU the leg's recording is the job's own)
    assert "multi_gpu_note" not in j


@pytest.mark.gpu
def test_cli_cold_start_on_a_one_hour_wav_by_process_wall(weights, tmp_path):
    """what a one-shot user of `speakerDiarizer` sees (the reference's surface is a one-shot CLI, sd.cpp:3415-3442): process wall of the CLI
    on a 1-h 16-bit wav against the job time the CLI itself reports ("Time cost", the reference's own timer line, sd.cpp:3434).  Everything
    that is not the job -- process start, HIP runtime initialisation, sd_create (weights -> HBM; the fp16 weight forms are built only when
    their mode is selected), wav read (115 MB), printing, teardown -- is reported, best of three runs (measured 541 ms on a quiet box; round 3: ~1 100 ms, 580 of them
    in sd_create); the assertion is on sd_create's own share."""
    import re
    import subprocess
    import time
    import synth
    pcm = synth.make_pcm(3600.0, seed=1234)
    p = tmp_path / "hour.wav"
    data = pcm.tobytes()
    fmt = struct.pack("<HHIIHH", 1, 1, 16000, 32000, 2, 16)
    p.write_bytes(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(data)) + data)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyannote-audio_speaker-diarization_cpp_amd", "speakerDiarizer")
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        out = subprocess.run([exe, weights[0], weights[1], str(p)], capture_output=True, text=True, timeout=600, env=dict(os.environ, SD_TRACE_CREATE="1"))
        wall = (time.perf_counter() - t0) * 1e3
        assert out.returncode == 0, out.stderr
        job = float(re.search(r"Time cost: (\d+)ms", out.stdout).group(1))
        if best is None or wall - job < best[0]:
            best = (wall - job, wall, job, out.stderr)
    print("cli wall %.0f ms, job %.0f ms, start-up + teardown %.0f ms\n%s" % (best[1], best[2], best[0], best[3]))
    assert out.stdout.count("--> Speaker_") >= 1
    # the gate is on what the library itself controls -- sd_create's own breakdown (SD_TRACE_CREATE: model read + layouts + upload, both
    # networks) -- not on the process wall, which also holds HIP runtime initialisation and a 115 MB file read and moves by a second between
    # boxes for identical calls (round 4, commit 8128be2); the wall is printed and only warned about
    parts = [float(v) for v in re.findall(r"(\d+\.\d+) ms", best[3])]
    assert len(parts) >= 4 and sum(parts[:4]) < 700.0, best[3]
    if best[0] >= 900.0:
        import warnings
        warnings.warn("speakerDiarizer start-up + teardown %.0f ms on this box (541 ms on a quiet one)" % best[0])


# ------------------------------------------------------------------ the two literal model seams (SURVEY 8b seams 2 / 3 as the reference declares them)
def _build_seam_shim(tmp_path, seam):
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "pyannote-audio_speaker-diarization_cpp_amd")
    so = str(tmp_path / ("lib%s.so" % seam))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-include", "algorithm", "-include", "cstdint", "-DSEAM_TEST_SHIM", "-fPIC", "-shared", "-I", os.path.join(root, "include"),
                           os.path.join(root, "oracle", "ref_build", seam + "_binding.cpp"), "-o", so, "-L", pkg, "-lsdhip", "-Wl,-rpath," + pkg])
    sdhip.lib()                                   # libsdhip.so is in the process before the shim resolves against it
    return C.CDLL(so)


@pytest.mark.gpu
def test_gpu_segment_chunks_is_the_model_call_of_sd_segment(diarizer, weights, tmp_path):
    """sd_segment_chunks = SegmentModel::infer as declared (sd.cpp:1352-1404): the reference's own slide() framing (sd.cpp:1419-1480, oracle crop)
    fed row by row gives sd_segment's scores -- bit for bit against the per-chunk form of the first convolution (option seg_shared_conv0 = 0),
    to 1e-5 against the default (one convolution over the overlapping chunks, the chunk normalisation folded behind it) -- and the torch
    oracle's within the parity tolerance; a shorter row yields the frames the network has for it, zero behind them; the reference-side
    binding (oracle/ref_build/seam2_binding.cpp, the reference's vector-of-vector types) returns the same numbers"""
    import ctypes as C
    import synth
    from oracle import nn_oracle as nn
    pcm = synth.make_pcm(21.3, seed=11)
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    nc, last = orc.num_chunks(len(wav))
    rows = np.stack([orc.crop(wav, i * 8000) for i in range(nc - 1)])                  # the full chunks, as slide() batches them
    out, fr = diarizer.segment_chunks(rows)
    assert fr == 293 and out.shape == (nc - 1, 293, 3)
    seg_default = diarizer.segment(wav)
    diarizer.set_option("seg_shared_conv0", 0)
    try:
        seg_plain = diarizer.segment(wav)
    finally:
        diarizer.set_option("seg_shared_conv0", 1)
    assert np.array_equal(out, seg_plain[:nc - 1])
    assert np.abs(out - seg_default[:nc - 1]).max() < 1e-5
    ref = nn.PyanNetOracle(weights[2])(rows).numpy()
    assert np.allclose(out, ref, rtol=1e-3, atol=1e-4)
    # the last, shorter chunk as slide() passes it (sd.cpp:1457-1480): its own length, fewer frames, zeros behind them
    tail = wav[(nc - 1) * 8000:][None, :]
    assert tail.shape[1] == last and last < 80000
    o2, fr2 = diarizer.segment_chunks(tail)
    ref2 = nn.PyanNetOracle(weights[2])(tail).numpy()
    assert fr2 == ref2.shape[1] and 0 < fr2 < 293
    assert np.allclose(o2[:, :fr2], ref2, rtol=1e-3, atol=1e-4) and not o2[:, fr2:].any()
    assert np.array_equal(o2[0], seg_plain[nc - 1])
    with pytest.raises(sdhip.SdError):
        diarizer.segment_chunks(np.zeros((2, 80001), np.float32))
    # the reference-side binding
    S = _build_seam_shim(tmp_path, "seam2")
    S.seam2_run.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.POINTER(C.c_int)]
    ob = np.zeros_like(out); frb = C.c_int(0)
    assert S.seam2_run(diarizer._h, rows.ctypes.data, rows.shape[0], rows.shape[1], ob.ctypes.data, C.byref(frb)) == 0
    assert frb.value == 293 and np.array_equal(ob, out)


@pytest.mark.gpu
def test_gpu_embed_signals_is_the_model_call_of_sd_embed(diarizer, weights, golden_dir, tmp_path):
    """sd_embed_signals = EmbeddingModel1::infer as declared (sd.cpp:1977-2040): fed the signals / wav_lens the reference's getEmbedding builds
    (sd.cpp:2436-2510; here the oracle's restatement, on the em_* masks minted from the reference's own Python) it returns sd_embed's rows
    bit for bit wherever sd_embed computes one (the NaN rule is getEmbedding's, not infer's), the torch oracle's within the parity
    tolerance, and so does the reference-side binding (oracle/ref_build/seam3_binding.cpp)"""
    import ctypes as C
    from oracle import nn_oracle as nn
    gold = np.load(os.path.join(golden_dir, "ref_nn_glue.npz"))
    wav = gold["em_pcm"].astype(np.float32) / np.float32(32768.0)
    masks = gold["em_masks"]
    items = masks.shape[0]
    wav = wav[:((items + 2) // 3 - 1) * 8000 + 80000]
    e_path = diarizer.embed(wav, masks)
    sigs = np.zeros((items, 80000), np.float32); cnts = np.zeros(items, np.int64)
    for i in range(items):
        sigs[i], cnts[i] = orc.mask_compact(orc.crop(wav, (i // 3) * 8000), masks[i])
    assert np.array_equal(cnts, gold["em_counts"][:items])
    lens = np.zeros(items, np.float32); bad = np.zeros(items, bool)
    for b0 in range(0, items, 32):
        l, ts, an = orc.wav_lens(cnts[b0:b0 + 32])
        lens[b0:b0 + 32] = l; bad[b0:b0 + 32] = ts | an
    assert np.array_equal(np.isnan(e_path[:, 0]), bad) and 0 < bad.sum() < items
    e_sig = diarizer.embed_signals(sigs, lens)
    assert np.isfinite(e_sig).all()                                      # infer itself has no NaN rule
    assert np.array_equal(e_sig[~bad], e_path[~bad])
    e_ref = nn.embed_ref(sigs, lens, weights[3]).numpy()
    g = e_sig.astype(np.float64)
    cos = (g * e_ref).sum(1) / np.linalg.norm(g, axis=1) / np.linalg.norm(e_ref, axis=1)
    assert (1 - cos).max() < 1e-3
    # element-wise on the rows getEmbedding keeps (the too-short ones are a few hundred samples of signal in 80 000 zeros: their embeddings are
    # computed -- infer has no NaN rule -- but are numerically touchy and thrown away by the caller; the cosine bound above covers them)
    # (atol 2e-4 of the embedding scale: the fixture's masks are the reference's edge cases -- items a few frames above the 640-sample rule --
    # whose embeddings sit further from the torch evaluation than test_embed_parity's random masks do; sd_embed's own parity is asserted there,
    # bit-identity with it above)
    np.testing.assert_allclose(g[~bad], e_ref[~bad], rtol=1e-3, atol=2e-4 * np.abs(e_ref).max())
    # a signal that is NOT silent behind its stated length (nothing getEmbedding produces, but the declared interface allows it): the dB ceiling
    # runs over all 501 frames, as the reference's does
    loud = sigs[:4].copy(); ll = np.full(4, 0.25, np.float32)
    loud[:, 30000:] = np.random.default_rng(3).standard_normal((4, 50000)).astype(np.float32)
    e_l = diarizer.embed_signals(loud, ll)
    r_l = nn.embed_ref(loud, ll, weights[3]).numpy()
    np.testing.assert_allclose(e_l, r_l, rtol=1e-3, atol=1e-4 * np.abs(r_l).max())
    with pytest.raises(sdhip.SdError):
        diarizer.embed_signals(sigs[:2], np.array([0.0, 1.0], np.float32))
    S = _build_seam_shim(tmp_path, "seam3")
    S.seam3_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    eb = np.zeros_like(e_sig)
    assert S.seam3_run(diarizer._h, sigs.ctypes.data, lens.ctypes.data, items, eb.ctypes.data) == 0
    assert np.array_equal(eb, e_sig)


@pytest.mark.gpu
def test_gpu_embedding_arena_falls_back_to_a_smaller_batch_plan(weights):
    """a context's first embedding call plans 768-item batches, every later one 3 072 (a ~57 GB arena at the 1-h size).  When that
    arena cannot be allocated (second context on the GPU, 8-h job beside the distance matrix) the call must not fail: the plan is repeated
    with 768, then 96 items, and -- the rows being batch-independent -- returns the same bits.  Fresh context, NO explicit batch option,
    three calls: implicit small plan, implicit large plan under an allocation limit that only the smaller plan passes, large plan unlimited"""
    rng = np.random.default_rng(21)
    items = 1080
    n = 8000 * (items // 3 - 1) + 80000
    wav = (0.2 * rng.standard_normal(n)).astype(np.float32)
    masks = (rng.random((items, 293)) > 0.3).astype(np.float32)
    d = sdhip.Diarizer(weights[0], weights[1])
    try:
        e1 = d.embed(wav, masks)                                  # first call of the context: 768-item plan
        assert d.kernel_stats("emb_arena_retries")["launches"] == 0
        d.set_option("ws_limit_mb", 5500)                         # the 1 080-item batch needs a 6.6 GB buffer ([rows][3072] f32), a 768-item batch 4.7 GB
        try:
            e2 = d.embed(wav, masks)                              # second call: 3 072-item plan -> one batch of 1 080 -> refused -> 768
        finally:
            d.set_option("ws_limit_mb", 0)
        assert d.kernel_stats("emb_arena_retries")["launches"] == 1
        e3 = d.embed(wav, masks)                                  # the large plan, nothing in its way
        assert d.kernel_stats("emb_arena_retries")["launches"] == 1
        assert np.isfinite(e1).any() and np.array_equal(e1, e2, equal_nan=True) and np.array_equal(e1, e3, equal_nan=True)
    finally:
        d.close()
    # a context that cannot even allocate the smallest plan: a clean error, not a crash; and the next context is unharmed
    d2 = sdhip.Diarizer(weights[0], weights[1])
    try:
        d2.set_option("ws_limit_mb", 200)
        try:
            with pytest.raises(sdhip.SdError):
                d2.embed(wav, masks)
        finally:
            d2.set_option("ws_limit_mb", 0)
        assert np.array_equal(d2.embed(wav, masks), e1, equal_nan=True)
    finally:
        d2.close()
