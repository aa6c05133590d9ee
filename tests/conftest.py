import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def weights(tmp_path_factory):
    """seeded synthetic weight packs (no checkpoints travel); returns (seg_path, emb_path, seg_dict, emb_dict)"""
    from oracle import nn_oracle as nn
    d = tmp_path_factory.mktemp("sdw")
    ws = nn.synth_segmentation_weights()
    we = nn.synth_embedding_weights()
    sp, ep = str(d / "segment.sdw"), str(d / "embedding.sdw")
    nn.save_pack(sp, ws)
    nn.save_pack(ep, we)
    return sp, ep, ws, we


@pytest.fixture(scope="session")
def diarizer(weights):
    import sdhip
    d = sdhip.Diarizer(weights[0], weights[1])     # raises loudly when the HIP library / GPU is missing
    yield d
    d.close()


@pytest.fixture(scope="session")
def weights_calibrated(tmp_path_factory, weights):
    """the same seeded networks with the ECAPA BatchNorm statistics learnt from one calibration batch (nn_oracle.calibrated_embedding_weights):
    unsaturated SE gates, the regime of a trained model -- the pack BASELINE configs[4]'s tolerance is asserted on"""
    from oracle import nn_oracle as nn
    d = tmp_path_factory.mktemp("sdw_cal")
    wc = nn.calibrated_embedding_weights()
    ep = str(d / "embedding_cal.sdw")
    nn.save_pack(ep, wc)
    return weights[0], ep, weights[2], wc


@pytest.fixture(scope="session")
def diarizer_calibrated(weights_calibrated):
    import sdhip
    d = sdhip.Diarizer(weights_calibrated[0], weights_calibrated[1])
    yield d
    d.close()
