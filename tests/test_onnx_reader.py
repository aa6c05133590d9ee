"""The reference CLI takes ONNX files (sd.cpp:3428-3430).  libsdhip's reader is tested on graphs written by
the same exporter the reference uses (torch.onnx.export, opset 17) from nn.Module forms of the two networks:
every tensor it extracts must equal the weights that went in (LSTM gates re-ordered iofc -> ifgo)."""
import os

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
