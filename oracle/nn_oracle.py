"""PyTorch-CPU restatement of the two neural networks and of the embedding front
end on the reference hot path.  TEST INFRASTRUCTURE ONLY (floating-point kernels
=> a torch fp32/fp64 reference is the oracle; tolerance rtol 1e-3 / atol 1e-4,
pipeline/script/verifyEveryStepResult.py:119-124).

PARITY UNPINNED: the reference runs these stages as opaque ONNX blobs that are
missing from the checkout (.MISSING_LARGE_BLOBS).  The architectures restated
here are the published upstream ones the exporters load:
  * segment/export2.py:17-21   pyannote/segmentation@2022.07 (PyanNet:
    SincNet stride 10 -> 4x biLSTM(128) -> 2x Linear(128) -> Linear(3) -> sigmoid),
    pyannote.audio 2.1.x (version not pinned by the reference).
  * embeddings/export3.py:158-159 + embeddings/threeModel.py:140-232
    speechbrain==0.5.14 spkrec-ecapa-voxceleb: spectral_magnitude -> Filterbank(80)
    -> MyNormalization (mean only) -> ECAPA_TDNN(C=1024, att 128, lin 192).
  * STFT: sd.cpp:1980-2013 (torch::stft fp64, fp32 periodic Hamming window,
    center, zero pad, onesided), transposed to [B,501,201,2] and cast to f32.
Weights are seeded synthetic (no checkpoints travel); the same weight pack file
feeds the HIP library so both sides use identical numbers.
"""
import struct
import numpy as np
import torch
import torch.nn.functional as F

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


def _t(w, name, dtype=torch.float32):
    return torch.from_numpy(np.asarray(w[name])).to(dtype)


# ---------------------------------------------------------------- PyanNet
class PyanNetOracle:
    """[B,1,T] -> [B,frames,3] sigmoid activities (sd.cpp:1350-1351 contract)."""

    def __init__(self, w, dtype=torch.float32):
        self.w = w
        self.dt = dtype
        self.lstm = torch.nn.LSTM(60, 128, num_layers=4, bidirectional=True, batch_first=True).to(dtype)
        sd = {}
        for k in self.lstm.state_dict().keys():
            sd[k] = _t(w, "lstm." + k, dtype)
        self.lstm.load_state_dict(sd)
        self.lstm.eval()

    def sincnet(self, x):
        w, dt = self.w, self.dt
        x = F.instance_norm(x, weight=_t(w, "sincnet.wav_norm.weight", dt), bias=_t(w, "sincnet.wav_norm.bias", dt), eps=1e-5)
        x = F.conv1d(x, _t(w, "sincnet.conv0.weight", dt), None, stride=10)
        x = torch.abs(x)
        x = F.max_pool1d(x, 3, 3)
        x = F.leaky_relu(F.instance_norm(x, weight=_t(w, "sincnet.norm0.weight", dt), bias=_t(w, "sincnet.norm0.bias", dt), eps=1e-5))
        for i in (1, 2):
            x = F.conv1d(x, _t(w, "sincnet.conv%d.weight" % i, dt), _t(w, "sincnet.conv%d.bias" % i, dt))
            x = F.max_pool1d(x, 3, 3)
            x = F.leaky_relu(F.instance_norm(x, weight=_t(w, "sincnet.norm%d.weight" % i, dt), bias=_t(w, "sincnet.norm%d.bias" % i, dt), eps=1e-5))
        return x  # [B,60,frames]

    @torch.no_grad()
    def __call__(self, wav, return_intermediate=False):
        w, dt = self.w, self.dt
        x = torch.as_tensor(wav).to(dt)
        if x.dim() == 2:
            x = x[:, None, :]
        feat = self.sincnet(x).transpose(1, 2).contiguous()     # [B,frames,60]
        h, _ = self.lstm(feat)
        y = F.leaky_relu(F.linear(h, _t(w, "linear.0.weight", dt), _t(w, "linear.0.bias", dt)))
        y = F.leaky_relu(F.linear(y, _t(w, "linear.1.weight", dt), _t(w, "linear.1.bias", dt)))
        y = torch.sigmoid(F.linear(y, _t(w, "classifier.weight", dt), _t(w, "classifier.bias", dt)))
        if return_intermediate:
            return y, feat, h
        return y


# ---------------------------------------------------------------- front end
@torch.no_grad()
def stft_ref(signals, window=None):
    """sd.cpp:1980-2036: fp64 STFT -> transpose(2,1) -> float32 [B,501,201,2]"""
    x = torch.as_tensor(signals, dtype=torch.float32).to(torch.float64)
    win = torch.hamming_window(WIN) if window is None else torch.as_tensor(window, dtype=torch.float32)
    y = torch.stft(x, N_FFT, HOP, WIN, win.to(torch.float64), center=True, pad_mode="constant",
                   normalized=False, onesided=True, return_complex=True)
    y = torch.view_as_real(y)                     # [B,201,T,2]
    return y.transpose(2, 1).to(torch.float32).contiguous()


@torch.no_grad()
def fbank_norm_ref(stft_out, wav_lens, mel, dtype=torch.float32):
    """threeModel.py:212-220, 333-369: power -> mel -> dB(top 80) -> mean-norm"""
    s = stft_out.to(dtype)
    power = s.pow(2).sum(-1)                                     # [B,T,201]
    fb = torch.matmul(power, torch.as_tensor(mel).to(dtype))      # [B,T,80]
    x_db = 10.0 * torch.log10(torch.clamp(fb, min=1e-10))
    mx = x_db.amax(dim=(-2, -1)) - 80.0
    x_db = torch.max(x_db, mx.view(-1, 1, 1))
    Tn = x_db.shape[1]
    lens = torch.as_tensor(wav_lens, dtype=torch.float32)
    out = x_db.clone()
    for i in range(x_db.shape[0]):
        n = int(torch.round(lens[i] * Tn).long())
        mean = x_db[i, 0:n].mean(dim=0)
        out[i] = x_db[i] - mean
    return out


# ---------------------------------------------------------------- ECAPA-TDNN
class EcapaOracle:
    def __init__(self, w, dtype=torch.float32):
        self.w = w
        self.dt = dtype

    def _conv(self, x, p, dil=1):
        W = _t(self.w, p + ".weight", self.dt)
        b = _t(self.w, p + ".bias", self.dt)
        k = W.shape[2]
        pad = dil * (k - 1) // 2
        if pad > 0:
            x = F.pad(x, (pad, pad), mode="reflect")
        return F.conv1d(x, W, b, dilation=dil)

    def _bn(self, x, p):
        w = self.w
        return F.batch_norm(x, _t(w, p + ".running_mean", self.dt), _t(w, p + ".running_var", self.dt),
                            _t(w, p + ".weight", self.dt), _t(w, p + ".bias", self.dt), False, 0.0, 1e-5)

    def _tdnn(self, x, p, dil=1):
        return self._bn(F.relu(self._conv(x, p + ".conv", dil)), p + ".norm")

    @staticmethod
    def _mask(lengths, L, dtype):
        # speechbrain length_to_mask(lengths*L, max_len=L): arange(L) < len
        return (torch.arange(L)[None, :] < (lengths * L)[:, None]).to(dtype)[:, None, :]

    def _se_res2net(self, x, p, dil, lengths):
        res = x
        x = self._tdnn(x, p + ".tdnn1")
        ys = []
        y_i = None
        for i, x_i in enumerate(torch.chunk(x, 8, dim=1)):
            if i == 0:
                y_i = x_i
            elif i == 1:
                y_i = self._tdnn(x_i, p + ".res2net.%d" % (i - 1), dil)
            else:
                y_i = self._tdnn(x_i + y_i, p + ".res2net.%d" % (i - 1), dil)
            ys.append(y_i)
        x = torch.cat(ys, dim=1)
        x = self._tdnn(x, p + ".tdnn2")
        L = x.shape[-1]
        mask = self._mask(lengths, L, x.dtype)
        total = mask.sum(dim=2, keepdim=True)
        s = (x * mask).sum(dim=2, keepdim=True) / total
        s = F.relu(self._conv(s, p + ".se.conv1"))
        s = torch.sigmoid(self._conv(s, p + ".se.conv2"))
        return s * x + res

    @torch.no_grad()
    def __call__(self, feats, wav_lens, return_intermediate=False):
        """feats [B,T,80] (normalised log-mel), wav_lens [B] -> [B,192]"""
        dt = self.dt
        x = torch.as_tensor(feats).to(dt).transpose(1, 2)
        lengths = torch.as_tensor(wav_lens, dtype=torch.float32).to(dt)
        inter = {}
        xl = []
        x = self._tdnn(x, "blocks.0", 1)
        xl.append(x)
        for b, dil in ((1, 2), (2, 3), (3, 4)):
            x = self._se_res2net(x, "blocks.%d" % b, dil, lengths)
            xl.append(x)
        inter["blocks"] = xl
        x = torch.cat(xl[1:], dim=1)
        x = self._tdnn(x, "mfa")
        inter["mfa"] = x
        L = x.shape[-1]
        mask = self._mask(lengths, L, dt)
        eps = 1e-12

        def stats(x, m):
            mean = (m * x).sum(2)
            std = torch.sqrt((m * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(eps))
            return mean, std

        total = mask.sum(dim=2, keepdim=True)
        mean, std = stats(x, mask / total)
        attn = torch.cat([x, mean.unsqueeze(2).repeat(1, 1, L), std.unsqueeze(2).repeat(1, 1, L)], dim=1)
        attn = self._conv(torch.tanh(self._tdnn(attn, "asp.tdnn")), "asp.conv")
        attn = attn.masked_fill(mask == 0, float("-inf"))
        attn = F.softmax(attn, dim=2)
        mean, std = stats(x, attn)
        pooled = torch.cat((mean, std), dim=1).unsqueeze(2)
        inter["pooled"] = pooled
        pooled = self._bn(pooled, "asp_bn")
        out = self._conv(pooled, "fc").squeeze(2)
        if return_intermediate:
            return out, inter
        return out


@torch.no_grad()
def embed_ref(signals, wav_lens, w, dtype=torch.float32):
    """EmbeddingModel1::infer (sd.cpp:1977-2040) on already compacted signals."""
    st = stft_ref(signals, w.get("stft.window"))
    feats = fbank_norm_ref(st, wav_lens, w["fbank.matrix"], dtype)
    return EcapaOracle(w, dtype)(feats, wav_lens)
