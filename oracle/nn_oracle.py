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
What the reference's own Python holds of these stages IS pinned (tests/test_reference_nn_glue.py, fixtures minted from
embeddings/threeModel.py): stft_ref against MySTFT (to float32 rounding: the Python runs it in float32, the C++ in fp64)
and sentence_mean_norm against MyNormalization (bit for bit).
Weights are seeded synthetic (no checkpoints travel); the same weight pack file
feeds the HIP library so both sides use identical numbers.
"""
import os
import sys
import numpy as np
import torch
import torch.nn.functional as F

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyannote-audio_speaker-diarization_cpp_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
# container + seeded weights are plain data shared with the product side; only the arithmetic below is the oracle
from weightpack import (N_FFT, HOP, WIN, N_BINS, N_MELS, T_FRAMES, EMB_DIM, save_pack, load_pack, mel_matrix,  # noqa: E402,F401
                        synth_segmentation_weights, synth_embedding_weights, calibrated_embedding_weights)



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
    return sentence_mean_norm(x_db, wav_lens)


def sentence_mean_norm(x, wav_lens):
    """threeModel.py:333-369 (MyNormalization, norm_type "sentence", mean only): subtract the mean over the first round(len * T) frames;
    pinned on the reference's own class by tests/test_reference_nn_glue.py"""
    Tn = x.shape[1]
    lens = torch.as_tensor(wav_lens, dtype=torch.float32)
    out = x.clone()
    for i in range(x.shape[0]):
        n = int(torch.round(lens[i] * Tn).long())
        mean = x[i, 0:n].mean(dim=0)
        out[i] = x[i] - mean
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


# ---------------------------------------------------------------- calibrated seeded pack (BASELINE configs[4]'s tolerance check)
class _CalibratingEcapa(EcapaOracle):
    """EcapaOracle whose BatchNorms LEARN their running statistics from the batch that passes through (per channel, over the batch and
    the valid frames), the way training leaves them in a real speechbrain ECAPA (embeddings/threeModel.py:140-232 loads such a model)."""

    def __init__(self, w, lengths):
        super().__init__(w)
        self.len = torch.as_tensor(lengths, dtype=torch.float32)

    def _bn(self, x, p):
        L = x.shape[-1]
        if L > 1:
            m = self._mask(self.len, L, x.dtype)                                     # [B,1,L]
            n = m.sum() + 1e-9
            mean = (x * m).sum(dim=(0, 2)) / n
            var = (((x - mean[None, :, None]) ** 2) * m).sum(dim=(0, 2)) / n
        else:
            mean, var = x.mean(dim=(0, 2)), x.var(dim=(0, 2), unbiased=False)
        self.w[p + ".running_mean"] = mean.numpy().astype(np.float32)
        self.w[p + ".running_var"] = np.maximum(var.numpy(), 1e-6).astype(np.float32)
        self.w[p + ".weight"] = np.ones_like(self.w[p + ".weight"])
        self.w[p + ".bias"] = np.zeros_like(self.w[p + ".bias"])
        return super()._bn(x, p)


@torch.no_grad()
def calibrate_embedding_weights(seed=4322, items=16, audio_seed=777):
    """COMPUTES the calibration (tools/mint_calibrated_bn.py stores its BatchNorm tensors as package data, weightpack.calibrated_embedding_weights
    loads them; tests/test_planted.py checks the stored ones against a fresh run).  The seeded synthetic ECAPA with every BatchNorm's running_mean / running_var set from ONE calibration batch (synthetic audio, full
    and partial-length items), gamma = 1, beta = 0: post-BN activations are zero-mean / unit-variance per channel, so the SE gates'
    pre-activations are O(1) and the gates unsaturated -- the regime of a trained model, where the plain seeded pack (BN = identity on
    log-mel inputs of order 30) has pre-activations of order 100 and gates pinned at 0 / 1 (profiles/r03_fp16_error_by_layer.txt).
    Same conv weights as synth_embedding_weights(seed): only the 31 BatchNorms change."""
    import synth
    w = {k: np.array(v, copy=True) for k, v in synth_embedding_weights(seed).items()}
    pcm = synth.make_pcm(5.0 * items / 2 + 6.0, seed=audio_seed)
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    rng = np.random.default_rng(audio_seed)
    sig = np.zeros((items, 80000), np.float32)
    lens = np.ones(items, np.float32)
    for i in range(items):
        s0 = int(rng.integers(0, len(wav) - 80000))
        n = 80000 if i % 2 == 0 else int(rng.integers(16000, 80000))                # half the items are partial: zero tail, wav_len < 1
        sig[i, :n] = wav[s0:s0 + n]
        lens[i] = n / 80000.0
    feats = fbank_norm_ref(stft_ref(sig, w.get("stft.window")), lens, w["fbank.matrix"])
    _CalibratingEcapa(w, lens)(feats, lens)
    return w
