"""The glue stages pinned on the REFERENCE'S OWN Python: tests/golden/ref_glue.npz was minted by tools/mint_reference_fixtures.py
from /root/reference/segment/utils.py (Segment, SlidingWindow -- the pyannote.core classes the C++ at sd.cpp:802-861, 1029-1138,
2567-2635 was ported from).  CPU: the C oracle (and the host-only entry points of libsdhip) against the vectors; GPU: the HIP path's
sd_reconstruct driven through the same gap-and-merge cases.

Regimes where the C++ deliberately deviates from the Python, and what the tests do there:
  #4   closest_frame clamps negative indices to 0 (sd.cpp:1086-1089)          -> asserted: the oracle returns 0 where Python is < 0
  #11  SlidingWindow::operator[] accumulates `start += step` instead of start + i * step (sd.cpp:1092-1115): for the 0.016875 s frame
       grid the two differ in the last bits for some i; the pipeline only uses operator[](29) narrowed to float (sd.cpp:2590)
       -> asserted bit-equal at the index the pipeline uses and after the float narrowing everywhere; the count of double-precision
       differences is printed, not asserted
  --   a run of activity that STARTS on the last row is an empty Segment: pyannote's Timeline drops it, the C++ keeps it -> not minted
"""
import hashlib
import os

import numpy as np
import pytest

import sdhip
from oracle import orc


@pytest.fixture(scope="module")
def gold(golden_dir):
    path = os.path.join(golden_dir, "ref_glue.npz")
    want = open(os.path.join(golden_dir, "ref_glue.sha256")).read().split()[0]
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == want, "fixture and manifest disagree: re-mint both"
    return np.load(path)


@pytest.mark.parametrize("name", ["chunks", "frames", "count"])
def test_closest_frame_against_reference_python(gold, name):
    st, step, dur = gold["cf_%s_win" % name]
    t, idx = gold["cf_%s_t" % name], gold["cf_%s_idx" % name]
    got = np.array([orc.closest_frame(float(x), st, step, dur) for x in t], np.int64)
    neg = idx < 0
    assert np.array_equal(got[~neg], idx[~neg])
    assert (got[neg] == 0).all()                                            # App. B #4: the C++ clamps
    assert len(t) > 240000 and (~neg).sum() > 200000
    if name == "chunks":
        assert neg.sum() > 0                                                # t < 2.25 s rounds to a negative chunk index in Python


def test_count_frames_against_reference_python(gold):
    """frames of the speaker count for c chunks (sd.cpp:1232): oracle and libsdhip's host-only sd_count_frames"""
    L = sdhip.lib()
    for c, nf in zip(gold["nf_chunks"], gold["nf_frames"]):
        assert orc.closest_frame(4.5 + (int(c) - 1) * 0.5, 0.5) + 1 == nf
        assert L.sd_count_frames(int(c)) == nf
    b = np.zeros((7, 293, 3))
    assert len(orc.speaker_count(b)[0]) == gold["nf_frames"][list(gold["nf_chunks"]).index(7)]


def test_range_to_segment_against_reference_python(gold):
    """the extents to_diarization intersects (sd.cpp:2691-2706) = SlidingWindow.range_to_segment(0, n)"""
    for name, st in (("frames", 0.0), ("count", 0.5)):
        for n, s, e in zip(gold["r2s_%s_n" % name], gold["r2s_%s_start" % name], gold["r2s_%s_end" % name]):
            assert orc.range_to_segment(0, int(n), st) == (s, e)
    for i0, n, s, e in zip(gold["r2s_gen_i0"], gold["r2s_gen_n"], gold["r2s_gen_start"], gold["r2s_gen_end"]):
        assert orc.range_to_segment(int(i0), int(n)) == (s, e)


def test_window_getitem_against_reference_python(gold):
    # chunk window: 0.5 * i is exact either way
    i, s = gold["gi_chunks_i"], gold["gi_chunks_start"]
    small = i < 4000
    got = np.array([orc.window_start(int(k), 0.5, 5.0) for k in i[small]])
    assert np.array_equal(got, s[small])
    # frame window: the C++ accumulates (App. B #11)
    i, s = gold["gi_frames_i"], gold["gi_frames_start"]
    small = i < 4000
    got = np.array([orc.window_start(int(k)) for k in i[small]])
    first = int(gold["sup_first_row"][0])
    # the one index the pipeline asks for (sd.cpp:2590), narrowed to float as the C++ does: 0.48937499999999967 (accumulated) and
    # 0.489375 (multiplied) are the same float, 0.4893749952316284
    assert first == 29 and np.float32(got[first]) == np.float32(s[first]) == np.float32(gold["sup_window_start"][0])
    # (elsewhere the two agree to one float ulp)
    rel = np.abs(got.astype(np.float32) - s[small].astype(np.float32)) / np.maximum(np.abs(s[small]), 1e-30).astype(np.float32)
    assert rel.max() <= 1.2e-7
    print("operator[] accumulated vs multiplied start: %d of %d differ in double precision, %d after the float narrowing"
          % ((got != s[small]).sum(), small.sum(), (got.astype(np.float32) != s[small].astype(np.float32)).sum()))


def _cases(gold):
    po, eo = gold["sup_pattern_off"], gold["sup_expected_off"]
    for k in range(len(po) - 1):
        yield (k, gold["sup_pattern"][po[k]:po[k + 1]], gold["sup_expected_start"][eo[k]:eo[k + 1]], gold["sup_expected_end"][eo[k]:eo[k + 1]],
               int(gold["sup_chunks"][k]))


def test_gap_and_merge_against_reference_segment_ops(gold):
    """Track::support / Segment::gap / ::merge (sd.cpp:831-860, 911-941) and the frame-middle timestamps of to_annotation
    (sd.cpp:2865-2867) against Segment.__xor__ / __or__ and SlidingWindow.__getitem__(i).middle of the reference's utils.py"""
    start = float(gold["sup_window_start"][0])
    collar = float(gold["sup_collar"][0])
    assert collar == orc.MIN_OFF_F32
    merged_any = 0
    for k, pat, es, ee, _ in _cases(gold):
        t = orc.to_annotation(pat.astype(np.float64)[:, None], start)
        assert [x[0] for x in t] == list(es) and [x[1] for x in t] == list(ee), k
        merged_any += int(gold["sup_raw_segments"][k]) - len(es)
        # the support step alone on the unmerged segments
        raw = orc.to_annotation(pat.astype(np.float64)[:, None], start, min_off=0.0)
        assert len(raw) == int(gold["sup_raw_segments"][k])
        assert orc.support([(a, b) for a, b, _ in raw], collar) == list(zip(es, ee))
        # the way the GPU test drives the case -- K = 1 and the speaker count carrying the pattern through reconstruct -- on the oracle
        c = int(gold["sup_chunks"][k])
        if c and k % 16 == 1:
            seg = np.full((c, 293, 3), 0.9, np.float32)
            cnt, win, ft = orc.speaker_count(np.ones((c, 293, 3)))
            assert len(cnt) == len(pat) - 1
            binr, st = orc.reconstruct(seg, np.zeros((c, 3), np.int32), pat[:len(cnt)].astype(np.int32), win, ft, 80000 + (c - 1) * 8000)
            assert st == start and np.array_equal(binr[:, 0], pat.astype(np.float64))
            assert [(a, b) for a, b, _ in orc.to_annotation(binr, st)] == list(zip(es, ee))
    assert merged_any > 2000


@pytest.mark.gpu
def test_hip_reconstruct_through_the_reference_gap_and_merge_cases(diarizer, gold):
    """the same cases through libsdhip's sd_reconstruct (a15-a17): one cluster, activity everywhere, and the speaker COUNT carries the
    pattern -- to_diarization keeps the count[t] best clusters per frame (sd.cpp:2681-2740), so with K = 1 the discrete diarization is
    the pattern itself; the turns must be the segments the reference's Segment / SlidingWindow classes give"""
    done = 0
    for k, pat, es, ee, c in _cases(gold):
        if c == 0 or k % 4:
            continue
        seg = np.full((c, 293, 3), 0.9, np.float32)
        binz = np.ones((c, 293, 3), np.uint8)
        hard = np.zeros((c, 3), np.int32)
        n_count = int(sdhip.lib().sd_count_frames(c))
        assert n_count == len(pat) - 1
        count = pat[:n_count].astype(np.int32)
        turns = diarizer.reconstruct(seg, binz, hard, count, 80000 + (c - 1) * 8000)
        assert [t[2] for t in turns] == [0] * len(es)
        assert [t[0] for t in turns] == list(es) and [t[1] for t in turns] == list(ee), (k, c)
        done += 1
    assert done > 60
