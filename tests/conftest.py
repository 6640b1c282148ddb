import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / at round end)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def synth_weights():
    """Seed-0 synthetic checkpoint (same arrays the golden fixtures were generated from)."""
    from streamspeech_amd import synth
    from streamspeech_amd.config import ModelConfig, VocoderConfig
    cfg, vcfg = ModelConfig(), VocoderConfig()
    return cfg, vcfg, synth.make_model_state_dict(0, cfg), synth.make_vocoder_state_dict(0, vcfg)


@pytest.fixture(scope="session")
def hip_model(synth_weights):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from streamspeech_amd.engine import HipModel
    import numpy as np
    cfg, vcfg, sd, vsd = synth_weights
    g = np.load(os.path.join(ROOT, "tests", "golden", "gcmvn_fr-en.npz"))
    return HipModel(sd, cfg, cmvn_mean=g["mean"], cmvn_std=g["std"])


@pytest.fixture(scope="session")
def hip_vocoder(synth_weights):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from streamspeech_amd.engine import HipVocoder
    cfg, vcfg, sd, vsd = synth_weights
    return HipVocoder(vsd, vcfg)
