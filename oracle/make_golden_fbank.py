"""Writes tests/golden/kaldi_fbank_hf.npz: log-mel filterbank features of seeded waveforms computed by a THIRD-PARTY
Kaldi-compliance implementation -- `transformers.SeamlessM4TFeatureExtractor._extract_fbank_features` (transformers 5.15.0 in
this image: `spectrogram(x * 2**15, povey window, 400 / 160 / 512, power 2, center=False, preemphasis 0.97,
remove_dc_offset, mel_filter_bank(257, 80, 20 Hz .. Nyquist, mel_scale="kaldi", triangularize_in_mel_space=True),
log with floor float32 eps)` -- the numpy path that package ships "to mimic Kaldi" when torchaudio is absent, and which its
own test-suite checks against `torchaudio.compliance.kaldi.fbank`).  These are exactly the arguments the reference passes to
torchaudio (fairseq/fairseq/data/audio/audio_utils.py:241-247: 80 bins, 16 kHz, everything else Kaldi's default; x 2^15 at
fairseq/examples/speech_to_text/data_utils.py:73-98).  torchaudio itself is not installable here (no index), so this is the
strongest pin available for SURVEY.md §8 row a1: an independently written and independently maintained implementation of the
same published algorithm.   python -m oracle.make_golden_fbank      TEST INFRASTRUCTURE."""
import os

import numpy as np

from streamspeech_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "kaldi_fbank_hf.npz")

CASES = {"noise_1s": ("noise", 11, 16000), "noise_short": ("noise", 12, 4000), "noise_ragged": ("noise", 13, 48017),
         "tonal_2s": ("tonal", 14, 32000), "quiet_1s": ("quiet", 15, 16000), "one_frame": ("noise", 16, 400)}


def waveform(kind, seed, n):
    if kind == "noise":
        return synth.synth_pcm(seed, n)
    t = np.arange(n) / 16000.0
    noise = synth.normal(seed, f"fbank_golden/{n}", (n,), 1.0)
    if kind == "tonal":      # decaying harmonic + a steady partial + a little noise: energy concentrated in few mel bins
        return (0.3 * np.sin(2 * np.pi * 220.0 * t) * np.exp(-t) + 0.1 * np.sin(2 * np.pi * 1900.0 * t + 0.5) + 0.01 * noise).astype(np.float32)
    return (1e-4 * noise).astype(np.float32)      # near-silence: exercises the log floor region


def hf_fbank(x):
    """transformers' Kaldi-compliance log-mel features [frames, 80] of a float waveform in [-1, 1)."""
    # oracle/ref_agent.py parks name-only `torchaudio` / `soundfile` stubs in sys.modules for the reference agent files; transformers probes
    # torchaudio with importlib.util.find_spec (lazily, at import and at call time), which chokes on a module without
    # __spec__ -- the stubs are hidden for the duration
    import sys
    probed = ("torchaudio", "soundfile", "librosa", "torchcodec", "omegaconf", "yt_dlp", "textgrid")   # what transformers looks for / the loaders stub
    hidden = {k: sys.modules.pop(k) for k in list(sys.modules)
              if k.split(".")[0] in probed and getattr(sys.modules[k], "__spec__", None) is None}
    try:
        from transformers import SeamlessM4TFeatureExtractor
        fe = SeamlessM4TFeatureExtractor(feature_size=80, sampling_rate=16000, num_mel_bins=80, stride=1)
        return np.asarray(fe._extract_fbank_features(np.asarray(x, np.float32)), np.float32)
    finally:
        sys.modules.update(hidden)


def main():
    import transformers
    out = {"transformers_version": np.array(transformers.__version__)}
    for name, (kind, seed, n) in CASES.items():
        x = waveform(kind, seed, n)
        out[name] = hf_fbank(x)
        print(name, out[name].shape, float(out[name].mean()))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
