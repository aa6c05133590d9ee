"""Deterministic synthetic 16 kHz mono speech-like audio: K=4 harmonic talkers
(f0 110/150/190/230 Hz, 20 harmonics, talker-specific 3-formant envelope) + -30 dB white noise,
turn lengths U(1,10) s, 10 % of boundaries overlapped by 0.5 s, 5 % silence.  int16 PCM.
Content does not change the neural cost; it only shapes the clustering input.

With seeded random weights the two networks do not follow the talkers (PyanNet says "speakers 0 and 1
always on", the ECAPA embedding separates items by length, not by voice), so the stages after them
would only ever see one degenerate case.  SURVEY 8(d) therefore prescribes a *planted* workload:
`planted_scores` derives the segmentation scores a trained PyanNet would emit from the very turn
schedule the audio was synthesised from (local speakers in order of first appearance inside each 5 s
chunk), `planted_embeddings` gives every (chunk, local speaker) item its talker's centroid + noise.
The library installs them with sd_set_planted(): the networks still run at full cost, their outputs
are replaced, and post-segmentation, masking, clustering and reconstruction see 4 talkers with
overlaps, silence, partial-length items and hundreds of turns."""
import numpy as np

SR = 16000
CHUNK, HOP, FRAMES, SPEAKERS, EMB_DIM = 80000, 8000, 293, 3, 192
_F0 = (110.0, 150.0, 190.0, 230.0)
_FORMANTS = ((700, 1200, 2600), (400, 2000, 2800), (300, 900, 2300), (550, 1700, 2500))


def _templates():
    # every f0 is a multiple of 10 Hz -> all talkers are periodic in 1600 samples
    t = np.arange(1600) / SR
    out = np.zeros((4, 1600), np.float32)
    for k in range(4):
        x = np.zeros(1600)
        for h in range(1, 21):
            f = h * _F0[k]
            env = sum(np.exp(-0.5 * ((f - fc) / 150.0) ** 2) for fc in _FORMANTS[k]) + 0.05
            x += env * np.sin(2 * np.pi * f * t + 0.37 * h * (k + 1))
        out[k] = (x / np.abs(x).max()).astype(np.float32)
    return out


def schedule(seconds, seed=1234, limit=None):
    """turn-taking schedule of the first `limit` samples (default: all) of the `seconds`-long signal:
    list of (start_sample, end_sample, talker or -1 for silence, overlapping previous talker or -1).
    The draws are sequential, so a prefix of the signal needs only a prefix of them."""
    n_full = int(round(seconds * SR))
    n = n_full if limit is None else max(0, min(int(limit), n_full))
    rng = np.random.default_rng(seed)
    out, pos, prev = [], 0, -1
    while pos < n:
        ln = int(rng.uniform(1.0, 10.0) * SR)
        end = min(n, pos + ln)
        if rng.random() < 0.05:
            k = -1
        else:
            k = int(rng.integers(0, 4))
            if k == prev:
                k = (k + 1) % 4
        ov = -1
        if prev >= 0 and k >= 0 and rng.random() < 0.10:
            ov = prev                                   # previous talker keeps going for 0.5 s
        out.append((pos, end, k, ov))
        prev, pos = k, end
    return out


def rasterize(turns, lo, hi):
    """per-sample primary / overlapping talker (-1 = none) of samples [lo, hi)"""
    who = np.full(hi - lo, -1, np.int8)
    who2 = np.full(hi - lo, -1, np.int8)
    for s, e, k, ov in turns:
        if e <= lo or s >= hi:
            continue
        who[max(s, lo) - lo:min(e, hi) - lo] = k
        if ov >= 0:
            o1 = min(hi, s + SR // 2)
            if o1 > max(s, lo):
                who2[max(s, lo) - lo:o1 - lo] = ov
    return who, who2


def make_pcm(seconds, seed=1234, limit=None):
    """-> int16 [seconds*16000]; with `limit` only the first `limit` samples are synthesised (they are
    identical to the prefix of the full signal: the schedule and the noise use separate streams)"""
    n_full = int(round(seconds * SR))
    n = n_full if limit is None else max(0, min(int(limit), n_full))
    rng_noise = np.random.default_rng(seed + 1000003)
    who, who2 = rasterize(schedule(seconds, seed, limit), 0, n)
    tpl = _templates()
    ph = np.arange(n) % 1600
    x = np.zeros(n, np.float32)
    for k in range(4):
        m = who == k
        x[m] += 0.25 * tpl[k][ph[m]]
        m2 = who2 == k
        x[m2] += 0.25 * tpl[k][ph[m2]]
    # slow amplitude modulation (syllable rate) and -30 dB noise
    x *= (0.75 + 0.25 * np.sin(2 * np.pi * 4.0 * np.arange(n, dtype=np.float32) / SR)).astype(np.float32)
    x += (0.25 * 10 ** (-30 / 20.0)) * rng_noise.standard_normal(n, dtype=np.float32)
    return np.clip(np.rint(x * 32768.0), -32768, 32767).astype(np.int16)


def num_chunks(n):
    """chunk rule of SegmentModel::slide (sd.cpp:1419, 1457)"""
    i = cnt = 0
    if n > CHUNK:
        cnt = (n - CHUNK + HOP - 1) // HOP
        i = cnt * HOP
    if i + 1 < n:
        cnt += 1
    return cnt


def _hash01(idx, seed):
    """counter-based uniform [0,1) (independent of how the recording is sharded)"""
    x = idx.astype(np.uint64) + np.uint64((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)      # wraps mod 2^64
    x ^= x >> np.uint64(33); x = x * np.uint64(0xFF51AFD7ED558CCD)
    x ^= x >> np.uint64(33); x = x * np.uint64(0xC4CEB9FE1A85EC53)
    x ^= x >> np.uint64(33)
    return (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def with_duets(turns):
    """the schedule as the planted scores see it: the first turn and every turn that follows a silence open as a duet
    (a second talker speaks along for the first 0.5 s), so that two speakers' turns START on the same frame --
    the equal-key case of Annotation::finalResult's std::sort (sd.cpp:962-978)"""
    out, prev = [], -1
    for s, e, k, ov in turns:
        if k >= 0 and prev < 0 and ov < 0:
            ov = (k + 2) % 4
        out.append((s, e, k, ov))
        prev = k
    return out


def planted_scores(turns, n_total, chunk_lo, chunk_hi, seed=99, pause=(48000, 3200)):
    """segmentation scores a trained PyanNet would give for chunks [chunk_lo, chunk_hi) of the recording
    whose schedule is `turns` (recordings made of several make_pcm pieces: concatenate their shifted turns).
    -> scores f32 [chunks][293][3] (active ~0.9, inactive ~0.05, both jittered; frames past the end of the
    recording are 0 like the reference's zero padding, sd.cpp:1473-1479), assign int8 [chunks][3] = talker
    of each local speaker (-1 = unused).  Local speakers are numbered by first appearance inside the chunk;
    a 4th talker inside one 5 s chunk is dropped.  pause = (period, length) in samples: everybody is silent
    for `length` samples every `period` (0.2 s every 3 s): breathing pauses shorter than min_duration_off
    that Track::support has to bridge (sd.cpp:911-941)."""
    nc = chunk_hi - chunk_lo
    if nc <= 0:
        return np.zeros((0, FRAMES, SPEAKERS), np.float32), np.zeros((0, SPEAKERS), np.int8)
    s_lo = chunk_lo * HOP
    s_hi = min(n_total, (chunk_hi - 1) * HOP + CHUNK)
    who, who2 = rasterize(turns, s_lo, s_hi)
    mid = ((np.arange(FRAMES) * 2 + 1) * CHUNK) // (2 * FRAMES)                # middle sample of each frame
    S = (np.arange(chunk_lo, chunk_hi, dtype=np.int64)[:, None] * HOP + mid[None, :])     # [nc][293] absolute
    inside = S < n_total
    Sl = np.where(inside, S, s_lo) - s_lo
    if pause is not None:
        inside_p = inside & ((S % pause[0]) >= pause[1])
    else:
        inside_p = inside
    a1 = np.where(inside_p, who[Sl], -1)
    a2 = np.where(inside_p, who2[Sl], -1)
    big = 10 * FRAMES
    first = np.full((nc, 4), big, np.int64)
    fidx = np.arange(FRAMES)[None, :]
    for k in range(4):
        f1 = np.where(a1 == k, 2 * fidx, big).min(1)
        f2 = np.where(a2 == k, 2 * fidx + 1, big).min(1)
        first[:, k] = np.minimum(f1, f2)
    order = np.argsort(first, axis=1, kind="stable")                           # talkers by first appearance
    assign = np.full((nc, SPEAKERS), -1, np.int8)
    active = np.zeros((nc, FRAMES, SPEAKERS), bool)
    rows = np.arange(nc)
    for loc in range(SPEAKERS):
        t = order[:, loc]
        used = first[rows, t] < big
        assign[used, loc] = t[used]
        active[:, :, loc] = used[:, None] & ((a1 == t[:, None]) | (a2 == t[:, None]))
    gidx = (np.arange(chunk_lo, chunk_hi, dtype=np.int64)[:, None, None] * FRAMES + np.arange(FRAMES)[None, :, None]) * SPEAKERS \
        + np.arange(SPEAKERS)[None, None, :]
    u = _hash01(gidx, seed)
    scores = np.where(active, 0.85 + 0.1 * u, 0.02 + 0.1 * u)
    scores = np.where(inside[:, :, None], scores, 0.0)
    return scores.astype(np.float32), assign


def planted_embeddings(assign, chunk_lo=0, seed=7, sigma=0.6, outlier_every=0):
    """one embedding per (chunk, local speaker): centre of the item's talker (4 centres ~ N(0, I), the clustering
    micro-bench recipe of SURVEY 8d) + sigma * N(0, I), f32 [chunks*3][192]; unused local speakers get talker 0's
    centre (their rows are NaN by the reference's too-short rule anyway and are never read).  outlier_every > 0
    turns every such item into a far outlier of its own (small clusters that Cluster::cluster must re-assign,
    sd.cpp:2377-2412)."""
    nc = assign.shape[0]
    cen = np.random.default_rng(seed).standard_normal((4, EMB_DIM))
    t = np.maximum(assign.reshape(-1).astype(np.int64), 0)
    item = np.arange(chunk_lo * SPEAKERS, (chunk_lo + nc) * SPEAKERS, dtype=np.int64)
    # Box-Muller on counter-based uniforms: rows do not depend on the shard they are generated in
    idx = item[:, None] * EMB_DIM + np.arange(EMB_DIM)[None, :]
    u1 = np.maximum(_hash01(idx, seed + 1), 1e-12)
    u2 = _hash01(idx, seed + 2)
    noise = np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * np.pi * u2)
    e = cen[t] + sigma * noise
    if outlier_every > 0:
        out = (item % outlier_every) == (outlier_every - 1)
        e[out] = 3.0 * noise[out] + 0.2 * cen[t[out]]
    return e.astype(np.float32)
