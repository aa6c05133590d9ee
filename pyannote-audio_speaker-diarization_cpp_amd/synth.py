"""Deterministic synthetic 16 kHz mono speech-like audio: K=4 harmonic talkers
(f0 110/150/190/230 Hz, 20 harmonics, talker-specific 3-formant envelope) + -30 dB white noise,
turn lengths U(1,10) s, 10 % of boundaries overlapped by 0.5 s, 5 % silence.  int16 PCM.
Content does not change the neural cost; it only shapes the clustering input."""
import numpy as np

SR = 16000
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


def make_pcm(seconds, seed=1234, limit=None):
    """-> int16 [seconds*16000]; with `limit` only the first `limit` samples are synthesised (they are
    identical to the prefix of the full signal: the schedule and the noise use separate streams)"""
    n_full = int(round(seconds * SR))
    n = n_full if limit is None else max(0, min(int(limit), n_full))
    rng = np.random.default_rng(seed)
    rng_noise = np.random.default_rng(seed + 1000003)
    who = np.full(n, -1, np.int8)          # primary talker per sample (-1 = silence)
    who2 = np.full(n, -1, np.int8)         # overlapping second talker
    pos, prev = 0, -1
    while pos < n:            # schedule draws are sequential, so a prefix needs only a prefix of them
        ln = int(rng.uniform(1.0, 10.0) * SR)
        end = min(n, pos + ln)
        if rng.random() < 0.05:
            k = -1
        else:
            k = int(rng.integers(0, 4))
            if k == prev:
                k = (k + 1) % 4
        who[pos:end] = k
        if prev >= 0 and k >= 0 and rng.random() < 0.10:
            who2[pos:min(n, pos + SR // 2)] = prev      # previous talker keeps going for 0.5 s
        prev, pos = k, end
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
