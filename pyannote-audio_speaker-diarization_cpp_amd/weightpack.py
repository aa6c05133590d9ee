"""weightpack.py -- the '.sdw' weight-pack container and the seeded synthetic weights the bench,
smoke test and tools use when the reference's ONNX blobs (segment2.onnx / emd4.onnx, missing from
the checkout) are not supplied.  Pure data: tensor names/shapes follow the upstream checkpoints the
reference's exporters load (segment/export2.py:17-21, embeddings/export3.py:158-159); the mel matrix
is the constant speechbrain's Filterbank folds into emd4.onnx (embeddings/threeModel.py:181-227).
csrc/weights.cpp reads this container; csrc/onnx_reader.cpp writes the same container from ONNX.
"""
import struct
import numpy as np
import torch

N_FFT, HOP, WIN = 400, 160, 400
N_BINS = 201
N_MELS = 80
T_FRAMES = 501
EMB_DIM = 192


# ---------------------------------------------------------------- weight pack
def save_pack(path, tensors):
    """'.sdw' pack: b'SDW1', u32 count, then per tensor: u16 name_len, name,
    u8 ndim, i64 dims[ndim], f32 data (no padding)."""
    with open(path, "wb") as f:
        f.write(b"SDW1")
        f.write(struct.pack("<I", len(tensors)))
        for name, arr in tensors.items():
            a = np.ascontiguousarray(arr, np.float32)
            nb = name.encode()
            f.write(struct.pack("<H", len(nb)))
            f.write(nb)
            f.write(struct.pack("<B", a.ndim))
            f.write(struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(a.tobytes())


def load_pack(path):
    out = {}
    with open(path, "rb") as f:
        assert f.read(4) == b"SDW1"
        (cnt,) = struct.unpack("<I", f.read(4))
        for _ in range(cnt):
            (nl,) = struct.unpack("<H", f.read(2))
            name = f.read(nl).decode()
            (nd,) = struct.unpack("<B", f.read(1))
            dims = struct.unpack("<%dq" % nd, f.read(8 * nd))
            n = int(np.prod(dims)) if nd else 1
            out[name] = np.frombuffer(f.read(4 * n), np.float32).reshape(dims).copy()
    return out


# ---------------------------------------------------------------- mel matrix
def mel_matrix(n_mels=N_MELS, n_fft=N_FFT, sr=16000, f_min=0.0, f_max=8000.0):
    """speechbrain 0.5.14 Filterbank(triangular, freeze) matrix [n_stft, n_mels]
    (what gets constant-folded into emd4.onnx)."""
    def to_mel(hz):
        return 2595.0 * np.log10(1.0 + hz / 700.0)

    def to_hz(mel):
        return 700.0 * (10.0 ** (mel / 2595.0) - 1.0)

    mel = torch.linspace(float(to_mel(f_min)), float(to_mel(f_max)), n_mels + 2)
    hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    band = hz[1:] - hz[:-1]
    band = band[:-1]
    f_central = hz[1:-1]
    n_stft = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sr // 2, n_stft)
    fc = f_central.repeat(n_stft, 1).transpose(0, 1)
    bd = band.repeat(n_stft, 1).transpose(0, 1)
    slope = (all_freqs.repeat(n_mels, 1) - fc) / bd
    left, right = slope + 1.0, -slope + 1.0
    fb = torch.max(torch.zeros(1), torch.min(left, right))
    return fb.transpose(0, 1).contiguous().numpy().astype(np.float32)


# ---------------------------------------------------------------- synthetic weights
def _conv_w(rng, co, ci, k, gain=1.0):
    return (rng.standard_normal((co, ci, k)) * (gain / np.sqrt(ci * k))).astype(np.float32)


def _bn(rng, c, prefix, out):
    out[prefix + ".weight"] = rng.uniform(0.6, 1.4, c).astype(np.float32)
    out[prefix + ".bias"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
    out[prefix + ".running_mean"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
    out[prefix + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)


def synth_segmentation_weights(seed=4321):
    rng = np.random.default_rng(seed)
    w = {}
    w["sincnet.wav_norm.weight"] = np.array([1.3], np.float32)
    w["sincnet.wav_norm.bias"] = np.array([0.02], np.float32)
    w["sincnet.conv0.weight"] = _conv_w(rng, 80, 1, 251, 4.0)
    w["sincnet.norm0.weight"] = rng.uniform(0.6, 1.4, 80).astype(np.float32)
    w["sincnet.norm0.bias"] = (0.1 * rng.standard_normal(80)).astype(np.float32)
    w["sincnet.conv1.weight"] = _conv_w(rng, 60, 80, 5, 1.5)
    w["sincnet.conv1.bias"] = (0.1 * rng.standard_normal(60)).astype(np.float32)
    w["sincnet.norm1.weight"] = rng.uniform(0.6, 1.4, 60).astype(np.float32)
    w["sincnet.norm1.bias"] = (0.1 * rng.standard_normal(60)).astype(np.float32)
    w["sincnet.conv2.weight"] = _conv_w(rng, 60, 60, 5, 1.5)
    w["sincnet.conv2.bias"] = (0.1 * rng.standard_normal(60)).astype(np.float32)
    w["sincnet.norm2.weight"] = rng.uniform(0.6, 1.4, 60).astype(np.float32)
    w["sincnet.norm2.bias"] = (0.1 * rng.standard_normal(60)).astype(np.float32)
    for layer in range(4):
        nin = 60 if layer == 0 else 256
        for sfx in ("", "_reverse"):
            w["lstm.weight_ih_l%d%s" % (layer, sfx)] = (rng.standard_normal((512, nin)) * (1.2 / np.sqrt(nin))).astype(np.float32)
            w["lstm.weight_hh_l%d%s" % (layer, sfx)] = (rng.standard_normal((512, 128)) * (1.2 / np.sqrt(128))).astype(np.float32)
            w["lstm.bias_ih_l%d%s" % (layer, sfx)] = (0.1 * rng.standard_normal(512)).astype(np.float32)
            w["lstm.bias_hh_l%d%s" % (layer, sfx)] = (0.1 * rng.standard_normal(512)).astype(np.float32)
    w["linear.0.weight"] = (rng.standard_normal((128, 256)) * (1.5 / np.sqrt(256))).astype(np.float32)
    w["linear.0.bias"] = (0.1 * rng.standard_normal(128)).astype(np.float32)
    w["linear.1.weight"] = (rng.standard_normal((128, 128)) * (1.5 / np.sqrt(128))).astype(np.float32)
    w["linear.1.bias"] = (0.1 * rng.standard_normal(128)).astype(np.float32)
    w["classifier.weight"] = (rng.standard_normal((3, 128)) * (6.0 / np.sqrt(128))).astype(np.float32)
    w["classifier.bias"] = np.array([-0.3, -0.8, -1.5], np.float32)
    return w


def synth_embedding_weights(seed=4322, C=1024):
    rng = np.random.default_rng(seed)
    w = {}
    w["fbank.matrix"] = mel_matrix()
    w["stft.window"] = torch.hamming_window(WIN).numpy().astype(np.float32)
    w["blocks.0.conv.weight"] = _conv_w(rng, C, 80, 5, 0.4)
    w["blocks.0.conv.bias"] = (0.1 * rng.standard_normal(C)).astype(np.float32)
    _bn(rng, C, "blocks.0.norm", w)
    S = C // 8
    for b, _dil in ((1, 2), (2, 3), (3, 4)):
        p = "blocks.%d" % b
        w[p + ".tdnn1.conv.weight"] = _conv_w(rng, C, C, 1, 1.4)
        w[p + ".tdnn1.conv.bias"] = (0.1 * rng.standard_normal(C)).astype(np.float32)
        _bn(rng, C, p + ".tdnn1.norm", w)
        for i in range(7):
            q = p + ".res2net.%d" % i
            w[q + ".conv.weight"] = _conv_w(rng, S, S, 3, 1.4)
            w[q + ".conv.bias"] = (0.1 * rng.standard_normal(S)).astype(np.float32)
            _bn(rng, S, q + ".norm", w)
        w[p + ".tdnn2.conv.weight"] = _conv_w(rng, C, C, 1, 1.4)
        w[p + ".tdnn2.conv.bias"] = (0.1 * rng.standard_normal(C)).astype(np.float32)
        _bn(rng, C, p + ".tdnn2.norm", w)
        w[p + ".se.conv1.weight"] = _conv_w(rng, 128, C, 1, 1.4)
        w[p + ".se.conv1.bias"] = (0.1 * rng.standard_normal(128)).astype(np.float32)
        w[p + ".se.conv2.weight"] = _conv_w(rng, C, 128, 1, 1.4)
        w[p + ".se.conv2.bias"] = (0.1 * rng.standard_normal(C)).astype(np.float32)
    w["mfa.conv.weight"] = _conv_w(rng, 3 * C, 3 * C, 1, 1.4)
    w["mfa.conv.bias"] = (0.1 * rng.standard_normal(3 * C)).astype(np.float32)
    _bn(rng, 3 * C, "mfa.norm", w)
    w["asp.tdnn.conv.weight"] = _conv_w(rng, 128, 9 * C, 1, 1.4)
    w["asp.tdnn.conv.bias"] = (0.1 * rng.standard_normal(128)).astype(np.float32)
    _bn(rng, 128, "asp.tdnn.norm", w)
    w["asp.conv.weight"] = _conv_w(rng, 3 * C, 128, 1, 3.0)
    w["asp.conv.bias"] = (0.1 * rng.standard_normal(3 * C)).astype(np.float32)
    _bn(rng, 6 * C, "asp_bn", w)
    w["fc.weight"] = _conv_w(rng, EMB_DIM, 6 * C, 1, 1.0)
    w["fc.bias"] = (0.1 * rng.standard_normal(EMB_DIM)).astype(np.float32)
    return w


def calibrated_embedding_weights(seed=4322):
    """synth_embedding_weights(seed) with the 31 BatchNorms' running_mean / running_var (gamma = 1, beta = 0) learnt from one calibration batch
    -- unsaturated SE gates, the regime of a trained ECAPA; the pack BASELINE configs[4]'s tolerance is asserted on.  The statistics are
    package data (calibrated_bn_<seed>.npz, minted by tools/mint_calibrated_bn.py through the torch oracle; a CPU test re-derives them)."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "calibrated_bn_%d.npz" % seed)
    if not os.path.exists(path):
        raise RuntimeError("no calibrated BatchNorm statistics for seed %d: run tools/mint_calibrated_bn.py" % seed)
    w = dict(synth_embedding_weights(seed))
    z = np.load(path)
    for k in z.files:
        assert k in w and w[k].shape == z[k].shape, k
        w[k] = z[k].astype(np.float32)
    return w
