"""The reference CLI takes ONNX files (sd.cpp:3428-3430).  libsdhip's reader is tested on graphs written by
the same exporter the reference uses (torch.onnx.export, opset 17) from nn.Module forms of the two networks:
every tensor it extracts must equal the weights that went in (LSTM gates re-ordered iofc -> ifgo)."""
import os
import re

import numpy as np
import pytest
import torch

import sdhip
from oracle import nn_modules as nm
from oracle import nn_oracle as nn


@pytest.fixture(scope="module")
def onnx_files(tmp_path_factory, weights):
    d = tmp_path_factory.mktemp("onnx")
    seg_p, emb_p = str(d / "segment2.onnx"), str(d / "emd4.onnx")
    nm.export_onnx(nm.PyanNetModule(weights[2]), (torch.randn(2, 1, 80000) * 0.1,), seg_p, ["signal"], ["segments"],
                   {"signal": {0: "B", 2: "T"}})                                 # segment/export2.py:42-52
    st = nn.stft_ref((0.1 * np.random.default_rng(0).standard_normal((2, 80000))).astype(np.float32))
    nm.export_onnx(nm.EmbeddingModule(weights[3]), (st, torch.tensor([1.0, 0.7])), emb_p, ["feats", "wav_lens"], ["embedding"])
    return seg_p, emb_p                                                          # embeddings/export3.py:177-189


def test_segmentation_onnx_to_pack(onnx_files, weights, tmp_path):
    out = str(tmp_path / "seg.sdw")
    sdhip.convert_onnx(onnx_files[0], "segmentation", out)
    p = nn.load_pack(out)
    assert set(p) == set(weights[2])
    for k, v in weights[2].items():
        assert np.array_equal(p[k], v), k


def test_embedding_onnx_to_pack(onnx_files, weights, tmp_path):
    out = str(tmp_path / "emb.sdw")
    sdhip.convert_onnx(onnx_files[1], "embedding", out)
    p = nn.load_pack(out)
    assert set(p) == set(weights[3]) - {"stft.window"}        # the STFT lives outside emd4.onnx (sd.cpp:1997-2008)
    for k in p:
        assert np.array_equal(p[k], weights[3][k]), k


def test_reader_rejects_wrong_or_broken_files(onnx_files, tmp_path):
    with pytest.raises(sdhip.SdError) as e:
        sdhip.convert_onnx(onnx_files[1], "segmentation", str(tmp_path / "x.sdw"))
    assert "expected 4 InstanceNormalization" in str(e.value)
    with pytest.raises(sdhip.SdError):
        sdhip.convert_onnx(onnx_files[0], "embedding", str(tmp_path / "x.sdw"))
    junk = tmp_path / "junk.onnx"
    junk.write_bytes(b"\x00\x01\x02 this is not a protobuf" * 10)
    with pytest.raises(sdhip.SdError):
        sdhip.convert_onnx(str(junk), "segmentation", str(tmp_path / "x.sdw"))
    trunc = tmp_path / "trunc.onnx"
    trunc.write_bytes(open(onnx_files[0], "rb").read()[:100000])
    with pytest.raises(sdhip.SdError):
        sdhip.convert_onnx(str(trunc), "segmentation", str(tmp_path / "x.sdw"))
    with pytest.raises(sdhip.SdError):
        sdhip.convert_onnx(str(tmp_path / "missing.onnx"), "segmentation", str(tmp_path / "x.sdw"))


@pytest.mark.gpu
def test_onnx_models_give_identical_results(onnx_files, diarizer):
    """sd_create on the .onnx files == sd_create on the .sdw packs (bit for bit for the segmentation net; the
    embedding path differs only by the Hamming window, which is not part of emd4.onnx and is re-computed)"""
    import synth
    d2 = sdhip.Diarizer(onnx_files[0], onnx_files[1])
    pcm = synth.make_pcm(21.0, seed=3)
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    assert np.array_equal(d2.segment(wav), diarizer.segment(wav))
    masks = (np.random.default_rng(1).random((33, 293)) > 0.3).astype(np.float32)
    np.testing.assert_allclose(d2.embed(wav, masks), diarizer.embed(wav, masks), rtol=1e-3, atol=1e-4 * 300, equal_nan=True)
    assert d2.diarize(pcm) == diarizer.diarize(pcm)
    # ... and the ONNX-loaded context against the ORACLE (not against another libsdhip context): the weights that came out of the graphs
    # drive the HIP path to the oracle's numbers
    from oracle import orc
    nc, _ = orc.num_chunks(len(wav))
    ws, we = nn.synth_segmentation_weights(), nn.synth_embedding_weights()
    seg_ref = nn.PyanNetOracle(ws)(np.stack([orc.crop(wav, i * 8000) for i in range(nc)])).numpy()
    assert np.allclose(d2.segment(wav), seg_ref, rtol=1e-3, atol=1e-4)
    feats, lens = d2.frontend(wav, masks[:6])
    e_ref = nn.EcapaOracle(we)(feats, lens).numpy()
    e_hip = d2.ecapa(feats, lens)
    cos = (e_hip.astype(np.float64) * e_ref).sum(1) / np.linalg.norm(e_hip.astype(np.float64), axis=1) / np.linalg.norm(e_ref, axis=1)
    assert (1 - cos).max() < 1e-5
    d2.close()


# ------------------------------------------------------------------ exporter variants the real blobs may use
def _export(module, args, path, input_names, output_names, dynamic_axes=None, **kw):
    import warnings
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    orig = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(module.eval(), args, path, opset_version=17, input_names=input_names, output_names=output_names,
                              dynamic_axes=dynamic_axes, dynamo=False, **kw)
    finally:
        onnx_proto_utils._add_onnxscript_fn = orig


# --- a 40-line protobuf re-writer (the `onnx` package is not installed): enough to move initializers into Constant nodes
def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _fields(buf):
    i, n = 0, len(buf)
    while i < n:
        tag = 0; sh = 0
        while True:
            c = buf[i]; i += 1
            tag |= (c & 0x7F) << sh; sh += 7
            if not c & 0x80:
                break
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            j = i
            while buf[j] & 0x80:
                j += 1
            yield f, wt, bytes(buf[i:j + 1]); i = j + 1
        elif wt == 2:
            ln = 0; sh = 0
            while True:
                c = buf[i]; i += 1
                ln |= (c & 0x7F) << sh; sh += 7
                if not c & 0x80:
                    break
            yield f, wt, bytes(buf[i:i + ln]); i += ln
        elif wt == 5:
            yield f, wt, bytes(buf[i:i + 4]); i += 4
        elif wt == 1:
            yield f, wt, bytes(buf[i:i + 8]); i += 8
        else:
            raise ValueError(wt)


def _emit(f, wt, payload):
    head = _varint((f << 3) | wt)
    return head + (_varint(len(payload)) + payload if wt == 2 else payload)


def _initializers_to_constant_nodes(model_bytes, external=False):
    """ModelProto -> the same model with every initializer turned into a Constant node (or, external=True, with the raw
    data of every initializer replaced by an external-data reference)"""
    out = bytearray()
    for f, wt, p in _fields(model_bytes):
        if f != 7:
            out += _emit(f, wt, p); continue
        g_nodes, g_rest = bytearray(), bytearray()
        for gf, gwt, gp in _fields(p):
            if gf == 5:                                       # TensorProto initializer
                name = [x for ff, _, x in _fields(gp) if ff == 8][0]
                if external:
                    t = b"".join(_emit(ff, ww, x) for ff, ww, x in _fields(gp) if ff not in (4, 9, 10))
                    t += _emit(14, 0, _varint(1)) + _emit(13, 2, _emit(1, 2, b"location") + _emit(2, 2, b"weights.bin"))
                    g_rest += _emit(5, 2, t)
                else:
                    attr = _emit(1, 2, b"value") + _emit(5, 2, gp) + _emit(20, 0, _varint(4))      # AttributeProto{name, t, type=TENSOR}
                    g_nodes += _emit(1, 2, _emit(2, 2, name) + _emit(4, 2, b"Constant") + _emit(5, 2, attr))
            else:
                g_rest += _emit(gf, gwt, gp)
        out += _emit(7, 2, bytes(g_nodes) + bytes(g_rest))
    return bytes(out)


@pytest.fixture(scope="module")
def seg_args(weights):
    return nm.PyanNetModule(weights[2]), (torch.randn(2, 1, 80000) * 0.1,)


@pytest.mark.parametrize("variant", ["no_constant_folding", "initializers_as_inputs", "constant_nodes"])
def test_segmentation_reader_on_exporter_variants(seg_args, weights, tmp_path, variant):
    """the real segment2.onnx may have been exported with constant folding off, with initializers kept as graph inputs, or
    post-processed so that weights sit in Constant nodes: the extracted tensors must be the same bits in every case"""
    path = str(tmp_path / "v.onnx")
    kw = {"do_constant_folding": variant != "no_constant_folding"}
    if variant == "initializers_as_inputs":
        kw["keep_initializers_as_inputs"] = True
    _export(seg_args[0], seg_args[1], path, ["signal"], ["segments"], {"signal": {0: "B", 2: "T"}}, **kw)
    if variant == "constant_nodes":
        data = open(path, "rb").read()
        open(path, "wb").write(_initializers_to_constant_nodes(data))
    out = str(tmp_path / "v.sdw")
    sdhip.convert_onnx(path, "segmentation", out)
    p = nn.load_pack(out)
    assert set(p) == set(weights[2])
    for k, v in weights[2].items():
        assert np.array_equal(p[k], v), (variant, k)


def test_reader_refuses_external_data_with_a_reason(onnx_files, tmp_path):
    ext = tmp_path / "ext.onnx"
    ext.write_bytes(_initializers_to_constant_nodes(open(onnx_files[0], "rb").read(), external=True))
    with pytest.raises(sdhip.SdError) as e:
        sdhip.convert_onnx(str(ext), "segmentation", str(tmp_path / "x.sdw"))
    assert "external" in str(e.value)


def test_embedding_reader_on_constant_node_variant(onnx_files, weights, tmp_path):
    path = tmp_path / "c.onnx"
    path.write_bytes(_initializers_to_constant_nodes(open(onnx_files[1], "rb").read()))
    out = str(tmp_path / "c.sdw")
    sdhip.convert_onnx(str(path), "embedding", out)
    p = nn.load_pack(out)
    for k in p:
        assert np.array_equal(p[k], weights[3][k]), k


# ------------------------------------------------------------------ a module tree nested and named like speechbrain 0.5.14's (embeddings/threeModel.py:181-227)
class _SBConv1d(torch.nn.Module):
    """speechbrain.nnet.CNN.Conv1d (skip_transpose=True, padding="same", padding_mode="reflect"): explicit F.pad, then `self.conv`"""
    def __init__(self, cin, cout, k, dil=1):
        super().__init__()
        self.k, self.dil = k, dil
        self.conv = torch.nn.Conv1d(cin, cout, k, dilation=dil, padding=0)

    def forward(self, x):
        pad = self.dil * (self.k - 1) // 2
        if pad:
            x = torch.nn.functional.pad(x, (pad, pad), mode="reflect")
        return self.conv(x)


class _SBBatchNorm1d(torch.nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = torch.nn.BatchNorm1d(c)

    def forward(self, x):
        return self.norm(x)


class _SBTDNN(torch.nn.Module):
    def __init__(self, cin, cout, k, dil):
        super().__init__()
        self.conv, self.activation, self.norm = _SBConv1d(cin, cout, k, dil), torch.nn.ReLU(), _SBBatchNorm1d(cout)

    def forward(self, x):
        return self.norm(self.activation(self.conv(x)))


class _SBRes2Net(torch.nn.Module):
    def __init__(self, c, dil):
        super().__init__()
        self.blocks = torch.nn.ModuleList([_SBTDNN(c // 8, c // 8, 3, dil) for _ in range(7)])

    def forward(self, x):
        y, ys = None, []
        for i, xi in enumerate(torch.chunk(x, 8, dim=1)):
            y = xi if i == 0 else (self.blocks[i - 1](xi) if i == 1 else self.blocks[i - 1](xi + y))
            ys.append(y)
        return torch.cat(ys, dim=1)


class _SBSE(torch.nn.Module):
    def __init__(self, c, s):
        super().__init__()
        self.conv1, self.relu, self.conv2, self.sigmoid = _SBConv1d(c, s, 1), torch.nn.ReLU(), _SBConv1d(s, c, 1), torch.nn.Sigmoid()

    def forward(self, x, mask):
        s = (x * mask).sum(dim=2, keepdim=True) / mask.sum(dim=2, keepdim=True)
        return self.sigmoid(self.conv2(self.relu(self.conv1(s)))) * x


class _SBSERes2Net(torch.nn.Module):
    def __init__(self, c, dil):
        super().__init__()
        self.tdnn1, self.res2net_block, self.tdnn2, self.se_block = _SBTDNN(c, c, 1, 1), _SBRes2Net(c, dil), _SBTDNN(c, c, 1, 1), _SBSE(c, 128)

    def forward(self, x, mask):
        return self.se_block(self.tdnn2(self.res2net_block(self.tdnn1(x))), mask) + x


class _SBASP(torch.nn.Module):
    def __init__(self, c):
        super().__init__()
        self.tdnn, self.tanh, self.conv = _SBTDNN(3 * c, 128, 1, 1), torch.nn.Tanh(), _SBConv1d(128, c, 1)

    def forward(self, x, mask):
        L = x.shape[-1]
        mw = mask / mask.sum(dim=2, keepdim=True)
        mean = (mw * x).sum(2)
        std = torch.sqrt((mw * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(1e-12))
        attn = torch.cat([x, mean.unsqueeze(2).repeat(1, 1, L), std.unsqueeze(2).repeat(1, 1, L)], dim=1)
        attn = torch.nn.functional.softmax(self.conv(self.tanh(self.tdnn(attn))).masked_fill(mask == 0, float("-inf")), dim=2)
        mean = (attn * x).sum(2)
        std = torch.sqrt((attn * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(1e-12))
        return torch.cat((mean, std), dim=1).unsqueeze(2)


class _SBEcapa(torch.nn.Module):
    def __init__(self, C=1024):
        super().__init__()
        self.blocks = torch.nn.ModuleList([_SBTDNN(80, C, 5, 1)] + [_SBSERes2Net(C, d) for d in (2, 3, 4)])
        self.mfa, self.asp, self.asp_bn, self.fc = _SBTDNN(3 * C, 3 * C, 1, 1), _SBASP(3 * C), _SBBatchNorm1d(6 * C), _SBConv1d(6 * C, 192, 1)

    def forward(self, x, lengths):
        x = x.transpose(1, 2)
        L = x.shape[-1]
        mask = (torch.arange(L)[None, :] < (lengths * L)[:, None]).float()[:, None, :]
        xl = []
        for i, layer in enumerate(self.blocks):
            x = layer(x) if i == 0 else layer(x, mask)
            xl.append(x)
        x = self.mfa(torch.cat(xl[1:], dim=1))
        return self.fc(self.asp_bn(self.asp(x, mask))).transpose(1, 2)


class _SBEncoder(torch.nn.Module):
    """the exported object of embeddings/export3.py: spectral_magnitude -> Filterbank -> MyNormalization -> mods.embedding_model"""
    def __init__(self, w):
        super().__init__()
        self.register_buffer("mel", torch.from_numpy(np.asarray(w["fbank.matrix"])))
        self.mods = torch.nn.ModuleDict(dict(embedding_model=_SBEcapa()))
        sd = {}
        for k, v in w.items():
            if k in ("fbank.matrix", "stft.window"):
                continue
            parts = k.split(".")
            leaf = parts[-1]
            body = ".".join(parts[:-1]).replace(".res2net.", ".res2net_block.blocks.").replace(".se.", ".se_block.")
            inner = "norm" if (body.endswith(".norm") or body == "asp_bn") else "conv"
            sd["mods.embedding_model.%s.%s.%s" % (body, inner, leaf)] = torch.from_numpy(np.asarray(v))
        own = self.state_dict()
        missing = [k for k in own if k not in sd and not k.endswith("num_batches_tracked") and k != "mel"]
        assert not missing, missing[:5]
        self.load_state_dict({**{k: v for k, v in own.items() if k not in sd}, **sd})

    def forward(self, feats, wav_lens):
        fb = torch.matmul(feats.pow(2).sum(-1), self.mel)
        x_db = 10.0 * torch.log10(torch.clamp(fb, min=1e-10))
        x_db = torch.max(x_db, (x_db.amax(dim=(-2, -1)) - 80.0).view(-1, 1, 1))
        T = x_db.size(1)
        rows = []
        for i in range(x_db.size(0)):
            n = torch.round(wav_lens[i] * T).long()
            rows.append(x_db[i] - x_db[i, 0:n].mean(dim=0))
        return self.mods["embedding_model"](torch.stack(rows), wav_lens)


def test_embedding_reader_on_a_speechbrain_shaped_module_tree(weights, tmp_path):
    """the reader takes the layers in execution order by operator type, never by name: a module tree nested and named the way
    speechbrain 0.5.14 builds ECAPA_TDNN (Conv1d / BatchNorm1d wrappers with an inner `.conv` / `.norm`, explicit reflect Pad in front
    of every k > 1 convolution, `mods.embedding_model.blocks.N.res2net_block.blocks.M`, `se_block`, `asp`) must give the same pack"""
    m = _SBEncoder(weights[3])
    st = nn.stft_ref((0.1 * np.random.default_rng(0).standard_normal((2, 80000))).astype(np.float32))
    ref = nm.EmbeddingModule(weights[3]).eval()
    with torch.no_grad():
        assert torch.equal(m.eval()(st, torch.tensor([1.0, 0.7])), ref(st, torch.tensor([1.0, 0.7])))
    path = str(tmp_path / "sb.onnx")
    _export(m, (st, torch.tensor([1.0, 0.7])), path, ["feats", "wav_lens"], ["embedding"], do_constant_folding=True)
    raw = open(path, "rb").read()
    assert b"mods.embedding_model.blocks.1.res2net_block.blocks.0.conv.conv.weight" in raw or b"onnx::Conv" in raw
    out = str(tmp_path / "sb.sdw")
    sdhip.convert_onnx(path, "embedding", out)
    p = nn.load_pack(out)
    assert set(p) == set(weights[3]) - {"stft.window"}
    for k in p:
        assert np.array_equal(p[k], weights[3][k]), k


# ------------------------------------------------------------------ malformed constant-folding inputs (header: "malformed models are reported ... nothing is guessed")
def _tensor(name, dims, floats=None, ints=None):
    t = b"".join(_emit(1, 0, _varint(d)) for d in dims)
    if floats is not None:
        t += _emit(2, 0, _varint(1)) + _emit(9, 2, np.asarray(floats, np.float32).tobytes())
    else:
        t += _emit(2, 0, _varint(7)) + _emit(9, 2, np.asarray(ints, np.int64).tobytes())
    return t + _emit(8, 2, name)


def _node(op, ins, outs, attrs=b""):
    return b"".join(_emit(1, 2, i) for i in ins) + b"".join(_emit(2, 2, o) for o in outs) + _emit(4, 2, op) + attrs


def _ints_attr(name, vals):
    return _emit(5, 2, _emit(1, 2, name) + b"".join(_emit(8, 0, _varint(v & 0xFFFFFFFFFFFFFFFF)) for v in vals) + _emit(20, 0, _varint(7)))


def _model(nodes, inits):
    g = b"".join(_emit(1, 2, n) for n in nodes) + _emit(2, 2, b"g") + b"".join(_emit(5, 2, t) for t in inits)
    return _emit(1, 0, _varint(8)) + _emit(7, 2, g)


@pytest.mark.parametrize("case", ["transpose_perm_out_of_range", "transpose_perm_duplicate", "unsqueeze_duplicate_axes", "concat_short_second_input",
                                  "numel_overflow", "reshape_two_minus_ones", "squeeze_axis_out_of_range"])
def test_constant_folder_survives_malformed_nodes(tmp_path, case):
    """graphs whose weight-shuffling nodes are malformed (perm outside [0, R), duplicated axes, an input holding fewer elements than its
    dims say, dims whose product overflows): the folder must skip them -- no out-of-bounds read, no giant allocation -- and the
    conversion must end with SD_ERR_MODEL and a reason"""
    x = _tensor(b"x", [2, 3], floats=np.arange(6))
    nodes, inits = [], [x]
    if case == "transpose_perm_out_of_range":
        nodes = [_node(b"Transpose", [b"x"], [b"y"], _ints_attr(b"perm", [0, 5]))]
    elif case == "transpose_perm_duplicate":
        nodes = [_node(b"Transpose", [b"x"], [b"y"], _ints_attr(b"perm", [1, 1]))]
    elif case == "unsqueeze_duplicate_axes":
        inits.append(_tensor(b"ax", [3], ints=[0, 0, 1]))
        nodes = [_node(b"Unsqueeze", [b"x", b"ax"], [b"y"])]
    elif case == "concat_short_second_input":
        inits.append(_tensor(b"z", [2, 300000], floats=np.arange(4)))           # says 600 000 elements, holds 4
        nodes = [_node(b"Concat", [b"x", b"z"], [b"y"], _emit(5, 2, _emit(1, 2, b"axis") + _emit(3, 0, _varint(1)) + _emit(20, 0, _varint(2))))]
    elif case == "numel_overflow":
        inits = [_tensor(b"x", [1 << 40, 1 << 40], floats=np.arange(6))]
        nodes = [_node(b"Identity", [b"x"], [b"y"])]
    elif case == "reshape_two_minus_ones":
        inits.append(_tensor(b"sh", [2], ints=[-1, -1]))
        nodes = [_node(b"Reshape", [b"x", b"sh"], [b"y"])]
    elif case == "squeeze_axis_out_of_range":
        inits.append(_tensor(b"ax", [1], ints=[7]))
        nodes = [_node(b"Squeeze", [b"x", b"ax"], [b"y"])]
    path = tmp_path / (case + ".onnx")
    path.write_bytes(_model(nodes, inits))
    for kind in ("segmentation", "embedding"):
        with pytest.raises(sdhip.SdError) as e:
            sdhip.convert_onnx(str(path), kind, str(tmp_path / "x.sdw"))
        assert e.value.code == 3 and ("expected" in str(e.value) or "no MatMul" in str(e.value))        # SD_ERR_MODEL with a reason


# ------------------------------------------------------------------ a PARAMETRISED first convolution: the real segment2.onnx most likely has one
class _ParamSincFB(torch.nn.Module):
    """asteroid_filterbanks.ParamSincFB as pyannote.audio's SincNet uses it (80 filters of 251 taps computed from 2 x 40 learnable
    frequencies): cos / sin band-pass pairs, Hamming half-window, flip for the right half"""
    def __init__(self, n_filters=80, kernel_size=251, sample_rate=16000.0, min_low_hz=50.0, min_band_hz=50.0):
        super().__init__()
        self.half, self.cutoff, self.kernel_size = kernel_size // 2, n_filters // 2, kernel_size
        self.min_low_hz, self.min_band_hz, self.sample_rate = min_low_hz, min_band_hz, sample_rate
        mel = np.linspace(2595 * np.log10(1 + 30 / 700), 2595 * np.log10(1 + (sample_rate / 2 - (min_low_hz + min_band_hz)) / 700), self.cutoff + 1)
        hz = 700 * (10 ** (mel / 2595) - 1)
        self.low_hz_ = torch.nn.Parameter(torch.from_numpy(hz[:-1]).view(-1, 1).float())
        self.band_hz_ = torch.nn.Parameter(torch.from_numpy(np.diff(hz)).view(-1, 1).float())
        n_lin = torch.linspace(0, kernel_size / 2 - 1, steps=self.half)
        self.register_buffer("window_", 0.54 - 0.46 * torch.cos(2 * np.pi * n_lin / kernel_size))
        self.register_buffer("n_", 2 * np.pi * (torch.arange(-self.half, 0.0).view(1, -1) / sample_rate))

    def _make(self, low, high, kind):
        band = (high - low)[:, 0]
        ft_low, ft_high = torch.matmul(low, self.n_), torch.matmul(high, self.n_)
        if kind == "cos":
            left = ((torch.sin(ft_high) - torch.sin(ft_low)) / (self.n_ / 2)) * self.window_
            center, right = 2 * band.view(-1, 1), torch.flip(left, dims=[1])
        else:
            left = ((torch.cos(ft_low) - torch.cos(ft_high)) / (self.n_ / 2)) * self.window_
            center, right = torch.zeros_like(band.view(-1, 1)), -torch.flip(left, dims=[1])
        return (torch.cat([left, center, right], dim=1) / (2 * band[:, None])).view(self.cutoff, 1, self.kernel_size)

    def filters(self):
        low = self.min_low_hz + torch.abs(self.low_hz_)
        high = torch.clamp(low + self.min_band_hz + torch.abs(self.band_hz_), self.min_low_hz, self.sample_rate / 2)
        return torch.cat([self._make(low, high, "cos"), self._make(low, high, "sin")], dim=0)


class _SincPyanNet(nm.PyanNetModule):
    def __init__(self, w):
        super().__init__(w)
        self.fb = _ParamSincFB()

    def forward(self, signal):
        F = torch.nn.functional
        x = F.conv1d(self.wav_norm(signal), self.fb.filters(), stride=10)
        x = F.leaky_relu(self.norm0(F.max_pool1d(torch.abs(x), 3, 3)))
        x = F.leaky_relu(self.norm1(F.max_pool1d(self.conv1(x), 3, 3)))
        x = F.leaky_relu(self.norm2(F.max_pool1d(self.conv2(x), 3, 3)))
        h, _ = self.lstm(x.transpose(1, 2))
        return torch.sigmoid(self.cls(F.leaky_relu(self.lin1(F.leaky_relu(self.lin0(h))))))


@pytest.mark.parametrize("fold", [True, False])
def test_segmentation_reader_evaluates_a_parametrised_sinc_filter_bank(weights, tmp_path, fold):
    """pyannote.audio's SincNet does not store its first convolution's 80 x 251 weights: it computes them in forward() from 2 x 40
    learnable frequencies (asteroid ParamSincFB).  torch.onnx.export cannot fold that sub-graph -- Abs, Clip, MatMul, Sin, Cos, a flip
    (Slice with step -1), ConstantOfShape, Neg are not on the TorchScript exporter's folding list -- so the reference's segment2.onnx
    (segment/export2.py:42-52) in all likelihood carries it in front of the Conv's weight input, with or without `do_constant_folding`.
    The reader evaluates constant sub-graphs the way a runtime would (float32, numpy broadcasting) and must arrive at torch's filters;
    the filter bank's own MatMuls must not be mistaken for the network's three linear layers."""
    m = _SincPyanNet(weights[2]).eval()
    path = str(tmp_path / "sinc.onnx")
    _export(m, (torch.randn(2, 1, 80000) * 0.1,), path, ["signal"], ["segments"], {"signal": {0: "B", 2: "T"}}, do_constant_folding=fold)
    raw = open(path, "rb").read()
    for op in (b"Sin", b"Cos", b"Clip", b"MatMul"):
        assert op in raw                                                     # the exporter did leave the sub-graph in the file
    out = str(tmp_path / "sinc.sdw")
    sdhip.convert_onnx(path, "segmentation", out)
    p = nn.load_pack(out)
    ref = m.fb.filters().detach().numpy()
    got = p["sincnet.conv0.weight"]
    assert got.shape == ref.shape == (80, 1, 251)
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()                # libm sin / cos against torch's: a few ulp of the largest tap
    for k, v in weights[2].items():                                          # everything else: the bits that went in
        if k != "sincnet.conv0.weight":
            assert np.array_equal(p[k], v), k


class _SBFilterbankEncoder(nm.EmbeddingModule):
    """the embedding exporter's front end with speechbrain 0.5.14's Filterbank as embeddings/threeModel.py:181-227 uses it: constructed
    inside forward(), it builds the triangular [201, 80] mel matrix from `f_central` / `band` with repeat, sub, div, min, max, transpose
    on every call instead of holding it as a buffer"""
    def __init__(self, w):
        super().__init__(w)
        sr, n_fft, n_mels, f_min, f_max = 16000, 400, 80, 0.0, 8000.0
        to_mel = lambda hz: 2595 * np.log10(1 + hz / 700)
        mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
        hz = 700 * (10 ** (mel / 2595) - 1)
        self.register_buffer("band", (hz[1:] - hz[:-1])[:-1])
        self.register_buffer("f_central", hz[1:-1])
        all_freqs = torch.linspace(0, sr // 2, n_fft // 2 + 1)
        self.register_buffer("all_freqs_mat", all_freqs.repeat(n_mels, 1))
        self.n_stft = n_fft // 2 + 1

    def sb_matrix(self):
        f_central_mat = self.f_central.repeat(self.n_stft, 1).transpose(0, 1)
        band_mat = self.band.repeat(self.n_stft, 1).transpose(0, 1)
        slope = (self.all_freqs_mat - f_central_mat) / band_mat
        left_side, right_side = slope + 1.0, -slope + 1.0
        return torch.max(torch.zeros(1), torch.min(left_side, right_side)).transpose(0, 1)

    def forward(self, feats, wav_lens):
        self.mel = self.sb_matrix()
        return super().forward(feats, wav_lens)


@pytest.mark.parametrize("fold", [True, False])
def test_embedding_reader_evaluates_a_filterbank_built_in_forward(weights, tmp_path, fold):
    """the [201, 80] mel matrix the reader looks for may not be an initializer at all: speechbrain's Filterbank, built inside the exported
    forward(), computes it from 80 centre frequencies with Tile / Sub / Div / Min / Max / Transpose.  Evaluated as a constant sub-graph it
    must be the matrix torch computes, and the rest of the pack the bits that went in."""
    w = dict(weights[3])
    m = _SBFilterbankEncoder(w).eval()
    st = nn.stft_ref((0.1 * np.random.default_rng(0).standard_normal((2, 80000))).astype(np.float32))
    path = str(tmp_path / "sbfb.onnx")
    _export(m, (st, torch.tensor([1.0, 0.7])), path, ["feats", "wav_lens"], ["embedding"], do_constant_folding=fold)
    if not fold:
        assert b"Tile" in open(path, "rb").read() or b"Expand" in open(path, "rb").read()      # the repeat() is in the file
    out = str(tmp_path / "sbfb.sdw")
    sdhip.convert_onnx(path, "embedding", out)
    p = nn.load_pack(out)
    ref = m.sb_matrix().numpy()
    assert p["fbank.matrix"].shape == (201, 80) and np.array_equal(p["fbank.matrix"], ref)
    for k in p:
        if k != "fbank.matrix":
            assert np.array_equal(p[k], weights[3][k]), k


def test_onnx_reader_under_address_and_ub_sanitizers(onnx_files, weights, tmp_path):
    """the code that parses untrusted model files (protobuf reader, constant folder, layer extraction, weight-pack reader), built for
    the CPU with AddressSanitizer + UBSan (tools/sanitize/build.sh) and run over: the malformed weight-shuffling graphs above, the two
    valid exports, truncations of a valid export at 60 lengths, 300 single-byte corruptions of it, and 200 corruptions of an export whose first
    convolution is a parametrised filter bank (the arithmetic constant evaluator).  No sanitizer report, no crash;
    the valid files convert."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "onnx_convert_asan")
    b = subprocess.run(["bash", os.path.join(root, "tools", "sanitize", "build.sh"), exe], capture_output=True, text=True, timeout=600)
    if b.returncode != 0:
        pytest.skip("sanitizer build not available here: " + b.stderr[-300:])
    x = _tensor(b"x", [2, 3], floats=np.arange(6))
    graphs = {
        "perm_oob": ([_node(b"Transpose", [b"x"], [b"y"], _ints_attr(b"perm", [0, 5]))], [x]),
        "perm_dup": ([_node(b"Transpose", [b"x"], [b"y"], _ints_attr(b"perm", [1, 1]))], [x]),
        "unsq_dup": ([_node(b"Unsqueeze", [b"x", b"ax"], [b"y"])], [x, _tensor(b"ax", [3], ints=[0, 0, 1])]),
        "concat_short": ([_node(b"Concat", [b"x", b"z"], [b"y"], _emit(5, 2, _emit(1, 2, b"axis") + _emit(3, 0, _varint(1)) + _emit(20, 0, _varint(2))))],
                         [x, _tensor(b"z", [2, 300000], floats=np.arange(4))]),
        "overflow": ([_node(b"Identity", [b"x"], [b"y"])], [_tensor(b"x", [1 << 40, 1 << 40], floats=np.arange(6))]),
        "reshape2": ([_node(b"Reshape", [b"x", b"sh"], [b"y"])], [x, _tensor(b"sh", [2], ints=[-1, -1])]),
        "squeeze_oob": ([_node(b"Squeeze", [b"x", b"ax"], [b"y"])], [x, _tensor(b"ax", [1], ints=[7])]),
        "slice_oob": ([_node(b"Slice", [b"x", b"st", b"en", b"ax"], [b"y"])],
                      [x, _tensor(b"st", [1], ints=[-100]), _tensor(b"en", [1], ints=[1 << 40]), _tensor(b"ax", [1], ints=[9])]),
        # round-3 advisor findings in the arithmetic evaluator: Range on an EMPTY (dims = {0}) operand, Gemm with a rank-3 C and with a C
        # that does not broadcast to [M, N]
        "range_empty": ([_node(b"Range", [b"e", b"lim", b"dl"], [b"y"])], [_tensor(b"e", [0], ints=[]), _tensor(b"lim", [], ints=[5]), _tensor(b"dl", [], ints=[1])]),
        "range_empty_f": ([_node(b"Range", [b"st", b"lim", b"e"], [b"y"])], [_tensor(b"st", [], floats=[0.0]), _tensor(b"lim", [], floats=[5.0]), _tensor(b"e", [0], floats=[])]),
        "gemm_c_rank3": ([_node(b"Gemm", [b"x", b"w", b"c"], [b"y"])], [x, _tensor(b"w", [3, 2], floats=np.arange(6)), _tensor(b"c", [1, 1, 2], floats=np.arange(2))]),
        "gemm_c_no_bcast": ([_node(b"Gemm", [b"x", b"w", b"c"], [b"y"])], [x, _tensor(b"w", [3, 3], floats=np.arange(9)), _tensor(b"c", [2], floats=np.arange(2))]),
        "gemm_c_ok": ([_node(b"Gemm", [b"x", b"w", b"c"], [b"y"])], [x, _tensor(b"w", [3, 3], floats=np.arange(9)), _tensor(b"c", [3], floats=np.arange(3))]),
    }
    paths = []
    for k, (n, i) in graphs.items():
        p = tmp_path / (k + ".onnx")
        p.write_bytes(_model(n, i))
        paths.append(str(p))
    good = open(onnx_files[0], "rb").read()
    rng = np.random.default_rng(5)
    for j, ln in enumerate(sorted(set(int(v) for v in np.linspace(1, len(good) - 1, 60)))):
        p = tmp_path / ("trunc%02d.onnx" % j)
        p.write_bytes(good[:ln])
        paths.append(str(p))
    for j in range(300):
        bad = bytearray(good)
        pos = int(rng.integers(0, min(len(bad), 60000)))       # the graph structure sits in front of the big raw_data blobs
        bad[pos] = int(rng.integers(0, 256))
        p = tmp_path / ("flip%03d.onnx" % j)
        p.write_bytes(bytes(bad))
        paths.append(str(p))
    # the arithmetic evaluator: a graph with a parametrised sinc filter bank in front of its first convolution, 200 corruptions of its node region
    sinc = str(tmp_path / "sinc_good.onnx")
    _export(_SincPyanNet(weights[2]).eval(), (torch.randn(2, 1, 80000) * 0.1,), sinc, ["signal"], ["segments"], {"signal": {0: "B", 2: "T"}}, do_constant_folding=False)
    sgood = open(sinc, "rb").read()
    for j in range(200):
        bad = bytearray(sgood)
        pos = int(rng.integers(0, min(len(bad), 30000)))
        bad[pos] = int(rng.integers(0, 256))
        p = tmp_path / ("sflip%03d.onnx" % j)
        p.write_bytes(bytes(bad))
        paths.append(str(p))
    paths += [sinc, onnx_files[0], onnx_files[1]]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe] + paths, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    assert "Sanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    assert len(re.findall(r" kind [01] -> \d", out.stdout)) == 2 * len(paths)               # (a corrupted tensor name may put a newline into a message)
    assert lines[-6].endswith("kind 0 -> 0 ") and lines[-4].endswith("kind 0 -> 0 ") and lines[-1].endswith("kind 1 -> 0 ")   # sinc and plain segment2.onnx as segmentation, emd4.onnx as embedding


@pytest.mark.gpu
def test_segmentation_from_a_parametrised_sinc_export_matches_the_exported_module(weights, tmp_path):
    """end to end through the operator seam: sd_create on an ONNX file whose first convolution is a parametrised sinc filter bank (the
    reader evaluates the filters), sd_segment on the GPU, against the very torch module that was exported"""
    import synth
    m = _SincPyanNet(weights[2]).eval()
    path = str(tmp_path / "sinc.onnx")
    _export(m, (torch.randn(2, 1, 80000) * 0.1,), path, ["signal"], ["segments"], {"signal": {0: "B", 2: "T"}}, do_constant_folding=True)
    d = sdhip.Diarizer(path, None)
    try:
        pcm = synth.make_pcm(12.0, seed=4)
        wav = pcm.astype(np.float32) / np.float32(32768.0)
        seg = d.segment(wav)
        from oracle import orc
        nc, _ = orc.num_chunks(len(wav))
        with torch.no_grad():
            ref = m(torch.from_numpy(np.stack([orc.crop(wav, i * 8000) for i in range(nc)]))[:, None, :]).numpy()
        assert seg.shape == ref.shape and np.allclose(seg, ref, rtol=1e-3, atol=1e-4), np.abs(seg - ref).max()
    finally:
        d.close()
