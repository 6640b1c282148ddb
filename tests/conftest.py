import os
import sys

import pytest

# The CPU suite runs tiny torch ops (oracle agents, reference modules): torch's default of one thread per core
# oversubscribes the box (51 min / 335 CPU-minutes on 8 cores).  Cap BEFORE torch spins up its pools;
# SS_TEST_THREADS overrides.
_NT = os.environ.get("SS_TEST_THREADS", "4")
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_k, _NT)

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # as the package sets it (streamspeech_amd/__init__.py), before anything starts the HIP runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / at round end)")
    config.addinivalue_line("markers", "slow: long CPU cases (live reference-agent re-runs); deselect with -m 'not gpu and not slow'")
    try:
        import torch
        torch.set_num_threads(int(_NT))
    except Exception:  # noqa: BLE001
        pass
    if not config.option.durations:
        config.option.durations = 10          # always print the ten slowest tests
    # no test may hang the suite: with pytest-timeout present (it is in this image) a test that passes SS_TEST_TIMEOUT seconds (default
    # 900 on a GPU box, 300 without one; the slowest GPU test takes ~2 min, the slowest CPU test ~15 s) fails instead
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        try:
            import torch
            has_gpu = torch.cuda.is_available()
        except Exception:  # noqa: BLE001
            has_gpu = False
        config.option.timeout = float(os.environ.get("SS_TEST_TIMEOUT", "900" if has_gpu else "300"))
        # "thread": a watchdog thread dumps every thread's stack and ends the process -- the default (SIGALRM) cannot interrupt a test
        # that is stuck inside a C call (seen twice in round 6: the CPU suite spinning in native code past the 900 s, no traceback)
        if not getattr(config.option, "timeout_method", None) or config.option.timeout_method == "signal":
            config.option.timeout_method = "thread"


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
