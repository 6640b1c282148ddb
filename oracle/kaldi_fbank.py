"""CPU ORACLE (test infrastructure): Kaldi-compatible log-mel filterbank + global CMVN.

The arithmetic lives in ``torchaudio.compliance.kaldi.fbank`` (third party, pinned only as
torchaudio>=0.8.0 by the reference's fairseq/setup.py:190; README env PyTorch 2.0.1 =>
torchaudio 2.0.x), which is NOT vendored under /root/reference and not installable in this
image.  This restates the published Kaldi/torchaudio algorithm for the arguments the reference
passes (fairseq/data/audio/audio_utils.py:241-247: num_mel_bins=80, sample_frequency=16000,
everything else default) -- SURVEY.md Appendix C.
PINNED (round 3) against an independent third-party implementation of the same algorithm with the
same arguments: transformers' SeamlessM4TFeatureExtractor numpy path ("to mimic Kaldi", checked
upstream against torchaudio.compliance.kaldi.fbank) -- oracle/make_golden_fbank.py ->
tests/golden/kaldi_fbank_hf.npz, tests/test_oracle_golden.py (max 1.5e-4, RMS 1-3e-6 on log-mel
values ~20).  NOT pinned against torchaudio's own code: that needs the package.
The reference call sites are agent/speech_to_speech.streamspeech.agent.py:66-98
(OnlineFeatureExtractor) and fairseq/examples/speech_to_text/data_utils.py:73-98 (x 2^15).
"""
import math

import numpy as np

SAMPLE_RATE = 16000
WIN = 400          # 25 ms
SHIFT = 160        # 10 ms
NFFT = 512         # round_to_power_of_two
NMEL = 80
LOW_FREQ = 20.0
HIGH_FREQ = 8000.0  # high_freq = 0 -> Nyquist
PREEMPH = 0.97
EPS = np.float32(1.1920928955078125e-07)  # torch.finfo(float32).eps, the log floor


def mel_scale(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_banks() -> np.ndarray:
    """[80, 257] float32 triangular weights (last column, the Nyquist bin, is zero)."""
    nbins = NFFT // 2
    fft_bin_width = SAMPLE_RATE / NFFT
    mlow, mhigh = mel_scale(LOW_FREQ), mel_scale(HIGH_FREQ)
    delta = (mhigh - mlow) / (NMEL + 1)
    b = np.arange(NMEL, dtype=np.float64)[:, None]
    left = mlow + b * delta
    center = left + delta
    right = center + delta
    mel = mel_scale(fft_bin_width * np.arange(nbins, dtype=np.float64))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    w = np.maximum(0.0, np.minimum(up, down))
    return np.pad(w, ((0, 0), (0, 1))).astype(np.float32)


def povey_window() -> np.ndarray:
    i = np.arange(WIN, dtype=np.float64)
    return ((0.5 - 0.5 * np.cos(2 * math.pi * i / (WIN - 1))) ** 0.85).astype(np.float32)


def num_frames(n_samples: int) -> int:
    """snip_edges=True framing."""
    return 0 if n_samples < WIN else 1 + (n_samples - WIN) // SHIFT


def fbank(waveform_i16_scale: np.ndarray) -> np.ndarray:
    """waveform: float32 mono 16 kHz already multiplied by 2^15 -> float32 [T, 80] log-mel."""
    x = np.asarray(waveform_i16_scale, dtype=np.float32)
    m = num_frames(x.shape[0])
    if m == 0:
        return np.zeros((0, NMEL), np.float32)
    idx = np.arange(m)[:, None] * SHIFT + np.arange(WIN)[None, :]
    fr = x[idx]                                                  # [m, 400]
    fr = fr - fr.mean(axis=1, keepdims=True, dtype=np.float32)   # remove_dc_offset
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)       # replicate-pad left
    fr = fr - np.float32(PREEMPH) * prev
    fr = fr * povey_window()[None, :]
    fr = np.pad(fr, ((0, 0), (0, NFFT - WIN)))
    spec = np.fft.rfft(fr.astype(np.float64), axis=1)
    power = (spec.real ** 2 + spec.imag ** 2).astype(np.float32)  # [m, 257]
    mel = power @ mel_banks().T
    return np.log(np.maximum(mel, EPS)).astype(np.float32)


def global_cmvn(feat: np.ndarray, mean: np.ndarray, std: np.ndarray) -> np.ndarray:
    """agent/speech_to_speech.streamspeech.agent.py:89-98: (x - mean) / std."""
    return ((feat - mean.astype(np.float32)) / std.astype(np.float32)).astype(np.float32)


def online_features(samples_16k: np.ndarray, mean: np.ndarray, std: np.ndarray) -> np.ndarray:
    """OnlineFeatureExtractor.__call__ on already-16 kHz float PCM in [-1, 1]
    (agent :66-87; the sox 48k->16k resample is outside the parity contract):
    keep floor((n - 240)/160) frames worth of samples, scale by 2^15, fbank, CMVN."""
    n = len(samples_16k)
    nfr = int(math.floor((n - (WIN - SHIFT)) / SHIFT))
    eff = int(nfr * SHIFT + (WIN - SHIFT)) if nfr > 0 else 0
    x = np.asarray(samples_16k[:eff], np.float32) * np.float32(2 ** 15)
    return global_cmvn(fbank(x), mean, std)
