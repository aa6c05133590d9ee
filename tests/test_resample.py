"""SURVEY 8(f) row 2, the resample leg (Resampler::Resample, frontend/resampler.cc:19-36): CPU -- the oracle (oracle/resample_oracle.py)
against the reference's own length rule, against scipy.signal.resample_poly applying the same taps, and against analytic tones;
GPU -- k_resample (sd_resample) against the oracle at the common rates and one awkward one, and the file entry point sd_diarize_wav:
a 44.1 kHz / 8 kHz / stereo file is refused or handled, never silently diarized at the wrong speed."""
import struct

import numpy as np
import pytest

import sdhip
from oracle import resample_oracle as ro

TOL = 3e-6          # f32 taps and f32 accumulation of up to ~1 600 products against the float64 oracle, relative to the signal's peak


def test_output_length_is_the_reference_rule():
    # resampler.cc:21-22 in float: the product n * ratio is rounded to 24 bits before the truncation
    assert ro.out_len(441000, 44100, 16000) == int(np.float32(441000) * np.float32(16000 / 44100))
    assert ro.out_len(80000, 8000, 16000) == 160000 and ro.out_len(48000 * 7, 48000, 16000) == 16000 * 7
    L = sdhip.lib()
    rng = np.random.default_rng(0)
    for sr in (8000, 11025, 22050, 32000, 44100, 48000, 96000, 44101):
        for n in list(rng.integers(1, 1 << 27, 40)) + [1, 2, 3, 100000]:
            assert L.sd_resample_len(int(n), sr, 16000) == ro.out_len(int(n), sr, 16000)
    assert L.sd_resample_len(100, 0, 16000) == -1


@pytest.mark.parametrize("sr", [8000, 22050, 44100, 48000])
def test_oracle_against_scipy_polyphase_with_the_same_taps(sr):
    from scipy.signal import resample_poly
    rng = np.random.default_rng(sr)
    x = rng.normal(size=sr // 4)
    L, M, fc, half, J = ro.plan(sr, 16000)
    H = int(np.floor(half))
    taps = ro.h(np.arange(-H, H + 1), L, fc, half)
    y_sp = resample_poly(x, L, M, window=taps / L)                   # scipy multiplies its window by `up`
    y = ro.resample(x, sr)
    k = min(len(y), len(y_sp))
    assert k >= len(x) * 16000 // sr - 1
    assert np.abs(y[:k] - y_sp[:k]).max() < 1e-12 * max(1.0, np.abs(y).max())


@pytest.mark.parametrize("sr", [8000, 44100, 48000])
def test_oracle_passes_tones_below_and_rejects_tones_above_the_new_nyquist(sr):
    n = sr
    t_in, t_out = np.arange(n) / sr, np.arange(ro.out_len(n, sr, 16000)) / 16000.0
    edge = 2000                                                      # outputs near the ends see the zero padding
    f_pass = 0.8 * min(sr, 16000) / 2                                # 0.8 of the lower Nyquist frequency
    y = ro.resample(np.sin(2 * np.pi * f_pass * t_in), sr)
    err = y[edge:-edge] - np.sin(2 * np.pi * f_pass * t_out[edge:-edge])
    assert 20 * np.log10(np.abs(err).max()) < -90
    if sr > 16000:
        f_stop = 8000 * 1.06                                         # just above the new Nyquist frequency: would alias to 7 520 Hz
        y = ro.resample(np.sin(2 * np.pi * f_stop * t_in), sr)
        assert 20 * np.log10(np.abs(y[edge:-edge]).max()) < -90


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("sr,seconds", [(8000, 7.3), (11025, 5.0), (22050, 5.0), (32000, 4.0), (44100, 11.7), (48000, 6.0), (96000, 2.0), (44101, 1.5)])
def test_hip_resampler_against_the_oracle(diarizer, sr, seconds):
    rng = np.random.default_rng(sr)
    n = int(sr * seconds)
    t = np.arange(n) / sr
    x = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 3100 * t + 1) + 0.05 * rng.normal(size=n)).astype(np.float32)
    x[n // 2:n // 2 + 50] = 0.9                                      # a step: every phase of the filter is exercised with non-smooth input
    y = diarizer.resample(x, sr)
    y_ref = ro.resample(x.astype(np.float64), sr)
    assert len(y) == len(y_ref) == ro.out_len(n, sr, 16000)
    assert np.abs(y - y_ref).max() < TOL * max(1.0, np.abs(y_ref).max()), np.abs(y - y_ref).max()
    assert np.abs(y_ref).max() > 0.5


def _wav(path, samples, sr, channels=1):
    data = np.asarray(samples, np.int16).tobytes()
    fmt = struct.pack("<HHIIHH", 1, channels, sr, sr * channels * 2, channels * 2, 16)
    path.write_bytes(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(data)) + data)


@pytest.mark.gpu
def test_wav_entry_point_checks_rate_and_channels(diarizer, weights, tmp_path):
    """sd_diarize_wav: 16 kHz mono = sd_diarize; another rate is REFUSED unless SD_WAV_RESAMPLE, and then it is sd_diarize_f32 of the
    resampled samples; stereo is read interleaved like the reference unless SD_WAV_DOWNMIX"""
    import subprocess, os
    import synth
    pcm = synth.make_pcm(21.0, seed=3)
    _wav(tmp_path / "m16.wav", pcm, 16000)
    t16 = diarizer.diarize(pcm)
    assert diarizer.diarize_wav(tmp_path / "m16.wav") == t16 and len(t16) >= 1
    # the same content at 44.1 kHz (made with the oracle's interpolator)
    up = ro.resample(pcm.astype(np.float64), 16000, 44100)
    pcm44 = np.clip(np.rint(up), -32768, 32767).astype(np.int16)
    _wav(tmp_path / "m44.wav", pcm44, 44100)
    with pytest.raises(sdhip.SdError) as e:
        diarizer.diarize_wav(tmp_path / "m44.wav")
    assert e.value.code == 1 and "44100" in str(e.value) and "resample" in str(e.value)
    got = diarizer.diarize_wav(tmp_path / "m44.wav", resample=True)
    w = diarizer.resample(pcm44.astype(np.float32) / np.float32(32768.0), 44100)
    assert got == diarizer.diarize_f32(w)                            # the entry point = resample + the f32 path, nothing else
    # stereo: default = the reference's interleaved read (wav.h:95-97); --downmix averages
    st = np.stack([pcm, pcm], 1).reshape(-1)
    _wav(tmp_path / "st.wav", st, 16000, channels=2)
    inter = diarizer.diarize_wav(tmp_path / "st.wav")
    assert inter == diarizer.diarize_f32(st[:len(pcm)].astype(np.float32) / np.float32(32768.0))
    assert diarizer.diarize_wav(tmp_path / "st.wav", downmix=True) == t16
    # the CLI: refusal is an error exit with the reason, --resample works
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyannote-audio_speaker-diarization_cpp_amd", "speakerDiarizer")
    out = subprocess.run([exe, weights[0], weights[1], str(tmp_path / "m44.wav")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 1 and "44100" in out.stderr
    out = subprocess.run([exe, weights[0], weights[1], str(tmp_path / "m44.wav"), "--resample"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.count("--> Speaker_") == len(got)
    # SD_WAV_ASSUME_16K / --assume-16k: the reference's own behaviour on such a file (sample_rate read and ignored, sd.cpp:2940-2942): the 44.1 kHz samples
    # are diarized as if they were 16 kHz -- exactly sd_diarize of those samples
    as16 = diarizer.diarize_wav(tmp_path / "m44.wav", assume_16k=True)
    assert as16 == diarizer.diarize(pcm44) and as16 != got
    out = subprocess.run([exe, weights[0], weights[1], str(tmp_path / "m44.wav"), "--assume-16k"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.count("--> Speaker_") == len(as16)
