"""CPU oracle of the resample leg (csrc/resample.hip).  TEST INFRASTRUCTURE ONLY.

The reference's Resampler::Resample (frontend/resampler.cc:19-36) is libsamplerate 0.2.2 `src_simple(SRC_SINC_BEST_QUALITY)`; the
library is a cmake FetchContent download (pipeline/cmake/samplerate.cmake) and NOT in the checkout, and its arithmetic is a 340 239-entry
coefficient table -- third-party data, absent: **parity unpinned** for the sample values of this leg.  What the reference's own lines
pin is restated exactly: the float ratio and the output length `(size_t)(in.size() * ratio)` (resampler.cc:21-22).  The values follow the
algorithm libsamplerate documents (band-limited interpolation, J. O. Smith) in exact polyphase form, float64:

    L / M = out_sr / in_sr reduced;  fc = CUTOFF / (2 max(L, M));  half = ZEROS / (2 fc)
    h(u) = L * 2 fc * sinc(2 fc u) * I0(BETA * sqrt(1 - (u / half)^2)) / I0(BETA),  |u| <= half, else 0
    y[m] = sum_k x[k] * h(m * M - k * L),   x[k] = 0 outside [0, n)

The same three constants as csrc/resample.hip.  tests/test_resample.py checks this file against scipy.signal.resample_poly applying
the same taps, against analytic tones, and the HIP kernel against this file."""
from math import gcd

import numpy as np

ZEROS = 64
CUTOFF = 0.95
BETA = 10.056


def out_len(n, in_sr, out_sr):
    """resampler.cc:21-22: `float ratio = 1.0 * out_sr / in_sr; out_wav->resize(in_wav.size() * ratio);` (size_t * float -> float)"""
    ratio = np.float32(1.0 * out_sr / in_sr)
    return int(np.float32(n) * ratio)


def plan(in_sr, out_sr):
    g = gcd(in_sr, out_sr)
    L, M = out_sr // g, in_sr // g
    fc = CUTOFF / (2.0 * max(L, M))
    half = ZEROS / (2.0 * fc)
    J = int(np.floor(half / L)) + 1
    return L, M, fc, half, J


def h(u, L, fc, half):
    u = np.asarray(u, np.float64)
    r2 = 1.0 - (u / half) ** 2
    w = np.where(np.abs(u) <= half, np.i0(BETA * np.sqrt(np.maximum(r2, 0.0))) / np.i0(BETA), 0.0)
    return L * 2.0 * fc * np.sinc(2.0 * fc * u) * w


def resample(x, in_sr, out_sr=16000):
    x = np.asarray(x, np.float64)
    n = len(x)
    L, M, fc, half, J = plan(in_sr, out_sr)
    no = out_len(n, in_sr, out_sr)
    m = np.arange(no, dtype=np.int64)
    t = m * M
    i0 = t // L
    ph = (t - i0 * L).astype(np.float64)
    xp = np.concatenate([np.zeros(J + 1), x, np.zeros(J + 2 + max(0, int(i0.max(initial=0)) + 1 - n))])
    y = np.zeros(no)
    for j in range(-J, J + 1):
        y += h(ph + j * L, L, fc, half) * xp[i0 - j + J + 1]
    return y
