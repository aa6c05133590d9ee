#!/usr/bin/env python3
"""Mints golden vectors for the glue stages from the REFERENCE'S OWN Python (`/root/reference/segment/utils.py`, the pyannote.core
classes the C++ at sd.cpp:802-861, 1029-1138, 2567-2635 was ported from).  Runs in the build container only (the reference tree
does not travel); what it writes -- `tests/golden/ref_glue.npz` + `tests/golden/ref_glue.sha256` -- is data: inputs and the outputs
the reference classes gave for them.  tests/test_reference_glue.py checks the oracle (CPU) and the HIP path (GPU) against it.

    python tools/mint_reference_fixtures.py            # rewrites the fixture; deterministic (seeded)

What is minted (each array family is one call site of the pipeline):
  cf_*     SlidingWindow.closest_frame (utils.py:409-425)  <-> SlidingWindow::closest_frame, sd.cpp:1084-1090
           on the pipeline's three windows: chunks (0 / 0.5 / 5.0), frames (0 / 0.016875 / 0.016875), trimmed count window
           (0.5 / 0.016875 / 0.016875), at the arguments aggregate() makes (chunk starts k * 0.5, sd.cpp:1251; frame targets, :1232)
           and at random times.  Negative results are kept with their flag: the C++ clamps them to 0.
  nf_*     frames speaker_count produces for c chunks = closest_frame(0.5 + 4.0 + (c - 1) * 0.5) + 1 on the count window (sd.cpp:1232)
  r2s_*    SlidingWindow.range_to_segment(0, n) (utils.py:497-540) <-> the extents of to_diarization, sd.cpp:2691-2706
  gi_*     SlidingWindow.__getitem__(i) (utils.py:560-583) start / end / middle <-> SlidingWindow::operator[] (sd.cpp:1092-1115, which
           accumulates `start += step`: App. B #11) and the frame-middle timestamps of to_annotation (sd.cpp:2865-2867)
  sup_*    Segment.__xor__ (gap) / __or__ (union) / __bool__ / duration (utils.py:73-99, 194-250) driven by the loop of pyannote.core's
           Timeline.support(collar) <-> Track::support + Segment::gap / ::merge, sd.cpp:831-860, 911-941, with
           collar = float32(0.5817029604921046) as the C++ call site holds it (sd.cpp:3210).  Cases are binary activity patterns on the
           frame grid (so that the HIP path can be driven through sd_reconstruct with the same patterns): the segments of a pattern
           are (middle[a], middle[b]) for every run of ones [a, b) (b = first inactive frame; a run that reaches the last row ends at
           the last row's middle -- the state machine of to_annotation, sd.cpp:2880-2921).
"""
import hashlib
import os
import sys

import numpy as np

REF = "/root/reference/segment"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "ref_glue.npz")

FRAME = 0.016875                       # sd.cpp:2430-2431
COLLAR = float(np.float32(0.5817029604921046))


def main():
    if not os.path.exists(os.path.join(REF, "utils.py")):
        raise SystemExit("reference tree absent: fixtures can only be minted in the build container")
    sys.path.insert(0, REF)
    import utils as ref                                  # the reference's own module (numpy only)
    ref.Segment.set_precision()                          # defines SEGMENT_PRECISION = 1e-6 (utils.py:48-71)
    rng = np.random.default_rng(20261003)
    out = {}

    # ---- closest_frame
    wins = {"chunks": (0.0, 0.5, 5.0), "frames": (0.0, FRAME, FRAME), "count": (0.5, FRAME, FRAME)}
    for name, (st, step, dur) in wins.items():
        w = ref.SlidingWindow(duration=dur, step=step, start=st)
        t = np.concatenate([np.arange(0, 60000) * 0.5,                                   # chunk starts (8 h = 57 591 chunks)
                            0.5 + np.arange(0, 60000) * 0.5,                              # trimmed chunk starts
                            5.0 + np.arange(0, 60000) * 0.5, 4.5 + np.arange(0, 60000) * 0.5,   # frame targets
                            rng.uniform(0.0, 30000.0, 4000), rng.uniform(0.0, 2.0, 1000),
                            (np.arange(0, 2000) + 0.5) * step + st + 0.5 * dur])           # exact .5 ties of the rounding
        idx = np.array([w.closest_frame(float(x)) for x in t], np.int64)
        out["cf_%s_win" % name] = np.array([st, step, dur])
        out["cf_%s_t" % name] = t
        out["cf_%s_idx" % name] = idx

    # ---- frames of the speaker count for c chunks
    wc = ref.SlidingWindow(duration=FRAME, step=FRAME, start=0.5)
    cs = np.concatenate([np.arange(1, 3000), rng.integers(3000, 60000, 3000)]).astype(np.int64)
    out["nf_chunks"] = cs
    out["nf_frames"] = np.array([wc.closest_frame(0.5 + 4.0 + (int(c) - 1) * 0.5) + 1 for c in cs], np.int64)

    # ---- range_to_segment(0, n): the extents to_diarization intersects
    ns = np.concatenate([np.arange(1, 2000), rng.integers(2000, 1800000, 3000)]).astype(np.int64)
    for name in ("frames", "count"):
        st, step, dur = wins[name]
        w = ref.SlidingWindow(duration=dur, step=step, start=st)
        segs = [w.range_to_segment(0, int(n)) for n in ns]
        out["r2s_%s_n" % name] = ns
        out["r2s_%s_start" % name] = np.array([s.start for s in segs])
        out["r2s_%s_end" % name] = np.array([s.end for s in segs])
    # general i0 as well (the class's own formula for i0 > 0)
    i0s = rng.integers(1, 100000, 2000).astype(np.int64)
    n2 = rng.integers(1, 100000, 2000).astype(np.int64)
    w = ref.SlidingWindow(duration=FRAME, step=FRAME, start=0.0)
    segs = [w.range_to_segment(int(a), int(b)) for a, b in zip(i0s, n2)]
    out["r2s_gen_i0"], out["r2s_gen_n"] = i0s, n2
    out["r2s_gen_start"] = np.array([s.start for s in segs])
    out["r2s_gen_end"] = np.array([s.end for s in segs])

    # ---- __getitem__
    ii = np.concatenate([np.arange(0, 4000), rng.integers(4000, 1800000, 2000)]).astype(np.int64)
    for name in ("chunks", "frames"):
        st, step, dur = wins[name]
        w = ref.SlidingWindow(duration=dur, step=step, start=st)
        segs = [w[int(i)] for i in ii]
        out["gi_%s_i" % name] = ii
        out["gi_%s_start" % name] = np.array([s.start for s in segs])
        out["gi_%s_end" % name] = np.array([s.end for s in segs])
        out["gi_%s_middle" % name] = np.array([s.middle for s in segs])

    # ---- support(collar) on activity patterns of the frame grid
    # window start of the discrete diarization = float32(frames[29].start): to_diarization crops to the count window, which starts
    # at 0.5 s: ceil((0.5 - 0.016875 - 0) / 0.016875) = 29 (sd.cpp:2577-2590)
    w = ref.SlidingWindow(duration=FRAME, step=FRAME, start=0.0)
    first = int(np.ceil((0.5 - FRAME - 0.0) / FRAME))
    start32 = float(np.float32(w[first].start))
    out["sup_first_row"] = np.array([first], np.int64)
    out["sup_window_start"] = np.array([start32])
    wa = ref.SlidingWindow(duration=FRAME, step=FRAME, start=start32)
    collar_frames = COLLAR / FRAME                                   # 34.47: gaps of 30..40 rows straddle the collar
    # rows of the discrete diarization of a c-chunk recording: to_diarization crops activations (window 0 / FRAME / FRAME, nact rows) and
    # the speaker count (window 0.5 / FRAME / FRAME, ncount rows) to the intersection of their extents with pyannote's "loose" crop
    # (i = ceil((focus.start - duration - start) / step), j = floor((focus.end - start) / step), rows [i, j + 1)), which the C++ evaluates in
    # float (sd.cpp:2577-2587).  The count is one row shorter than the activations' crop, so the last row is never active there.
    wcnt = ref.SlidingWindow(duration=FRAME, step=FRAME, start=0.5)

    def rows_for(c):
        nact = w.closest_frame(5.0 + (c - 1) * 0.5) + 1
        ncount = wcnt.closest_frame(4.5 + (c - 1) * 0.5) + 1
        a, cc = w.range_to_segment(0, nact), wcnt.range_to_segment(0, ncount)
        f0, f1 = max(a.start, cc.start), min(a.end, cc.end)
        ar0 = max(0, int(np.ceil(np.float32((f0 - FRAME - 0.0) / FRAME))))
        ar1 = min(nact, int(np.floor(np.float32((f1 - 0.0) / FRAME))) + 1)
        cr1 = min(ncount, int(np.floor(np.float32((f1 - 0.5) / FRAME))) + 1)
        assert ar0 == first
        return ar1 - ar0, cr1, ncount

    pats, offs, exp_s, exp_e, exp_off, raw_cnt, case_chunks = [], [0], [], [], [0], [], []
    for case in range(400):
        drivable = case % 7 != 0                                     # 6 of 7 cases can be driven through the whole reconstruction (GPU test)
        if drivable:
            c = int(rng.integers(1, 171))
            rows, crow, ncount = rows_for(c)
            assert crow == ncount == rows - 1
        else:
            c = 0
            rows = int(rng.integers(200, 3000))
        case_chunks.append(c)
        pat = np.zeros(rows, np.uint8)
        pos = int(rng.integers(0, 40))
        while pos < rows:
            on = int(rng.integers(1, 200)) if rng.random() < 0.8 else int(rng.integers(1, 4))
            pat[pos:pos + on] = 1
            r = rng.random()
            if r < 0.55:
                gap = int(rng.integers(30, 41))                       # around the collar
            elif r < 0.75:
                gap = int(rng.integers(1, 30))
            else:
                gap = int(rng.integers(41, 400))
            pos += on + gap
        if not drivable:
            pat[-int(rng.integers(1, 50)):] = 1                       # active through the last row
        else:
            pat[rows - 1] = 0                                         # (no count for the last row)
        if case % 11 == 0:
            pat[:int(rng.integers(1, 50))] = 1                        # active from row 0
        if pat[-1] and not pat[-2]:
            pat[-1] = 0            # a run that STARTS on the last row is an empty segment: pyannote's Timeline drops it, the C++ keeps it (excluded regime)
        mids = [wa[i].middle for i in range(rows)]
        # runs of ones -> segments (state machine of to_annotation with onset = offset = 0.5 on a 0/1 pattern)
        segs, a = [], None
        for i in range(rows):
            if pat[i] and a is None:
                a = i
            elif not pat[i] and a is not None:
                segs.append(ref.Segment(mids[a], mids[i]))
                a = None
        if a is not None:
            segs.append(ref.Segment(mids[a], mids[rows - 1]))
        raw_cnt.append(len(segs))
        # pyannote.core Timeline.support(collar) / support_iter: merge while the gap is empty or shorter than the collar
        merged = []
        if segs:
            segs = sorted(segs)
            new = segs[0]
            for s in segs[1:]:
                possible_gap = s ^ new                                 # utils.py:224-250
                if not possible_gap or possible_gap.duration < COLLAR:
                    new = new | s                                      # utils.py:194-222
                else:
                    merged.append(new)
                    new = s
            merged.append(new)
        pats.append(pat)
        offs.append(offs[-1] + rows)
        exp_s += [m.start for m in merged]
        exp_e += [m.end for m in merged]
        exp_off.append(exp_off[-1] + len(merged))
    out["sup_pattern"] = np.concatenate(pats)
    out["sup_pattern_off"] = np.array(offs, np.int64)
    out["sup_expected_start"] = np.array(exp_s)
    out["sup_expected_end"] = np.array(exp_e)
    out["sup_expected_off"] = np.array(exp_off, np.int64)
    out["sup_raw_segments"] = np.array(raw_cnt, np.int64)
    out["sup_collar"] = np.array([COLLAR])
    out["sup_chunks"] = np.array(case_chunks, np.int64)            # chunks of the recording whose reconstruction has exactly this many rows (0 = CPU-only case)

    np.savez_compressed(OUT, **out)
    h = hashlib.sha256(open(OUT, "rb").read()).hexdigest()
    with open(OUT.replace(".npz", ".sha256"), "w") as f:
        f.write("%s  ref_glue.npz\n# python tools/mint_reference_fixtures.py  (imports /root/reference/segment/utils.py; numpy %s)\n" % (h, np.__version__))
    print("wrote %s (%d arrays, %d bytes, sha256 %s)" % (OUT, len(out), os.path.getsize(OUT), h[:16]))
    print("collar / frame = %.3f rows; merged %d of %d raw segments" % (collar_frames, sum(raw_cnt) - len(exp_s), sum(raw_cnt)))


if __name__ == "__main__":
    main()
