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
