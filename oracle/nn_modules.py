"""torch.nn.Module forms of the two networks, used ONLY to export ONNX test files with the same
exporter the reference uses (torch.onnx.export, opset 17; segment/export2.py:42-52 and
embeddings/export3.py:177-189), so the library's ONNX reader can be tested without the missing
blobs.  TEST INFRASTRUCTURE ONLY."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class PyanNetModule(nn.Module):
    def __init__(self, w):
        super().__init__()
        t = lambda k: torch.from_numpy(np.asarray(w[k]))
        self.wav_norm = nn.InstanceNorm1d(1, affine=True)
        self.conv0 = nn.Conv1d(1, 80, 251, stride=10, bias=False)
        self.conv1 = nn.Conv1d(80, 60, 5)
        self.conv2 = nn.Conv1d(60, 60, 5)
        self.norm0 = nn.InstanceNorm1d(80, affine=True)
        self.norm1 = nn.InstanceNorm1d(60, affine=True)
        self.norm2 = nn.InstanceNorm1d(60, affine=True)
        self.lstm = nn.LSTM(60, 128, num_layers=4, bidirectional=True, batch_first=True)
        self.lin0 = nn.Linear(256, 128)
        self.lin1 = nn.Linear(128, 128)
        self.cls = nn.Linear(128, 3)
        with torch.no_grad():
            self.wav_norm.weight.copy_(t("sincnet.wav_norm.weight")); self.wav_norm.bias.copy_(t("sincnet.wav_norm.bias"))
            self.conv0.weight.copy_(t("sincnet.conv0.weight"))
            for i, (c, nrm) in enumerate(((self.conv0, self.norm0), (self.conv1, self.norm1), (self.conv2, self.norm2))):
                if i:
                    c.weight.copy_(t("sincnet.conv%d.weight" % i)); c.bias.copy_(t("sincnet.conv%d.bias" % i))
                nrm.weight.copy_(t("sincnet.norm%d.weight" % i)); nrm.bias.copy_(t("sincnet.norm%d.bias" % i))
            self.lstm.load_state_dict({k: t("lstm." + k) for k in self.lstm.state_dict().keys()})
            self.lin0.weight.copy_(t("linear.0.weight")); self.lin0.bias.copy_(t("linear.0.bias"))
            self.lin1.weight.copy_(t("linear.1.weight")); self.lin1.bias.copy_(t("linear.1.bias"))
            self.cls.weight.copy_(t("classifier.weight")); self.cls.bias.copy_(t("classifier.bias"))

    def forward(self, signal):
        x = self.wav_norm(signal)
        x = F.leaky_relu(self.norm0(F.max_pool1d(torch.abs(self.conv0(x)), 3, 3)))
        x = F.leaky_relu(self.norm1(F.max_pool1d(self.conv1(x), 3, 3)))
        x = F.leaky_relu(self.norm2(F.max_pool1d(self.conv2(x), 3, 3)))
        h, _ = self.lstm(x.transpose(1, 2))
        y = F.leaky_relu(self.lin0(h))
        y = F.leaky_relu(self.lin1(y))
        return torch.sigmoid(self.cls(y))


class _TDNN(nn.Module):
    def __init__(self, w, p, dil):
        super().__init__()
        W = torch.from_numpy(np.asarray(w[p + ".conv.weight"]))
        self.k, self.dil = W.shape[2], dil
        self.conv = nn.Conv1d(W.shape[1], W.shape[0], self.k, dilation=dil)
        self.norm = nn.BatchNorm1d(W.shape[0])
        with torch.no_grad():
            self.conv.weight.copy_(W); self.conv.bias.copy_(torch.from_numpy(np.asarray(w[p + ".conv.bias"])))
            for a, b in (("weight", "weight"), ("bias", "bias"), ("running_mean", "running_mean"), ("running_var", "running_var")):
                getattr(self.norm, a).copy_(torch.from_numpy(np.asarray(w[p + ".norm." + b])))

    def forward(self, x):
        pad = self.dil * (self.k - 1) // 2
        if pad:
            x = F.pad(x, (pad, pad), mode="reflect")
        return self.norm(F.relu(self.conv(x)))


def _conv1x1(w, p):
    W = torch.from_numpy(np.asarray(w[p + ".weight"]))
    c = nn.Conv1d(W.shape[1], W.shape[0], W.shape[2])
    with torch.no_grad():
        c.weight.copy_(W); c.bias.copy_(torch.from_numpy(np.asarray(w[p + ".bias"])))
    return c


class EmbeddingModule(nn.Module):
    """MyEmbedding0 (embeddings/threeModel.py:140-232): STFT output + wav_lens -> embedding"""

    def __init__(self, w):
        super().__init__()
        self.register_buffer("mel", torch.from_numpy(np.asarray(w["fbank.matrix"])))
        self.block0 = _TDNN(w, "blocks.0", 1)
        self.blk = nn.ModuleList()
        for b, dil in ((1, 2), (2, 3), (3, 4)):
            p = "blocks.%d" % b
            m = nn.ModuleDict(dict(tdnn1=_TDNN(w, p + ".tdnn1", 1),
                                   res=nn.ModuleList([_TDNN(w, p + ".res2net.%d" % i, dil) for i in range(7)]),
                                   tdnn2=_TDNN(w, p + ".tdnn2", 1), se1=_conv1x1(w, p + ".se.conv1"), se2=_conv1x1(w, p + ".se.conv2")))
            self.blk.append(m)
        self.mfa = _TDNN(w, "mfa", 1)
        self.asp_tdnn = _TDNN(w, "asp.tdnn", 1)
        self.asp_conv = _conv1x1(w, "asp.conv")
        self.asp_bn = nn.BatchNorm1d(self.mel.shape[1] and w["asp_bn.weight"].shape[0])
        with torch.no_grad():
            for a in ("weight", "bias", "running_mean", "running_var"):
                getattr(self.asp_bn, a).copy_(torch.from_numpy(np.asarray(w["asp_bn." + a])))
        self.fc = _conv1x1(w, "fc")

    @staticmethod
    def _mask(lengths, L):
        return (torch.arange(L)[None, :] < (lengths * L)[:, None]).float()[:, None, :]

    def forward(self, feats, wav_lens):
        power = feats.pow(2).sum(-1)
        fb = torch.matmul(power, self.mel)
        x_db = 10.0 * torch.log10(torch.clamp(fb, min=1e-10))
        x_db = torch.max(x_db, (x_db.amax(dim=(-2, -1)) - 80.0).view(-1, 1, 1))
        T = x_db.size(1)
        rows = []
        for i in range(x_db.size(0)):                 # unrolled at export like MyNormalization (threeModel.py:352-366)
            n = torch.round(wav_lens[i] * T).long()
            rows.append(x_db[i] - x_db[i, 0:n].mean(dim=0))
        x = torch.stack(rows).transpose(1, 2)
        xl = []
        x = self.block0(x)
        for m in self.blk:
            res = x
            x = m["tdnn1"](x)
            ys, y = [], None
            for i, xi in enumerate(torch.chunk(x, 8, dim=1)):
                y = xi if i == 0 else (m["res"][i - 1](xi) if i == 1 else m["res"][i - 1](xi + y))
                ys.append(y)
            x = m["tdnn2"](torch.cat(ys, dim=1))
            L = x.shape[-1]
            mask = self._mask(wav_lens, L)
            s = (x * mask).sum(dim=2, keepdim=True) / mask.sum(dim=2, keepdim=True)
            x = torch.sigmoid(m["se2"](F.relu(m["se1"](s)))) * x + res
            xl.append(x)
        x = self.mfa(torch.cat(xl, dim=1))
        L = x.shape[-1]
        mask = self._mask(wav_lens, L)
        mw = mask / mask.sum(dim=2, keepdim=True)
        mean = (mw * x).sum(2)
        std = torch.sqrt((mw * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(1e-12))
        attn = torch.cat([x, mean.unsqueeze(2).repeat(1, 1, L), std.unsqueeze(2).repeat(1, 1, L)], dim=1)
        attn = self.asp_conv(torch.tanh(self.asp_tdnn(attn)))
        attn = F.softmax(attn.masked_fill(mask == 0, float("-inf")), dim=2)
        mean = (attn * x).sum(2)
        std = torch.sqrt((attn * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(1e-12))
        pooled = self.asp_bn(torch.cat((mean, std), dim=1).unsqueeze(2))
        return self.fc(pooled).transpose(1, 2)


def export_onnx(module, args, path, input_names, output_names, dynamic_axes=None):
    """torch.onnx.export (TorchScript exporter, opset 17) without the `onnx` python package: the
    only use of that package is a post-pass for custom onnxscript functions, which these graphs lack."""
    import warnings
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    orig = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(module.eval(), args, path, opset_version=17, do_constant_folding=True,
                              input_names=input_names, output_names=output_names, dynamic_axes=dynamic_axes, dynamo=False)
    finally:
        onnx_proto_utils._add_onnxscript_fn = orig
