"""TEST INFRASTRUCTURE (oracle): rational polyphase resampler, numpy restatement.

The reference resamples 48 kHz input to 16 kHz with libsox `rate` through torchaudio
(fairseq/data/audio/audio_utils.py:53-62, called from agent/speech_to_speech.streamspeech.agent.py:66-98
via convert_waveform).  Neither sox nor torchaudio is vendored or installed, and the
north-star parity contract starts at the fbank input, so the resampler is restated as the textbook
zero-phase polyphase FIR with the published design of scipy.signal.resample_poly (scipy 1.15:
half_len = 10*max(up,down), cutoff 1/max(up,down), Kaiser beta 5, gain up) and PINNED against
scipy.signal.resample_poly itself in tests/test_frontend_cpu.py.  PARITY VS SOX: UNPINNED.
"""
import math

import numpy as np


def design_filter(up: int, down: int) -> np.ndarray:
    """firwin(2*half_len+1, 1/max_rate, window=('kaiser', 5.0)) * up, in float64."""
    max_rate = max(up, down)
    half_len = 10 * max_rate
    n = np.arange(2 * half_len + 1, dtype=np.float64) - half_len
    fc = 1.0 / max_rate
    h = fc * np.sinc(fc * n) * np.kaiser(2 * half_len + 1, 5.0)
    h /= h.sum()                       # unit DC gain (firwin scale=True)
    return h * up


def resample_poly_ref(x: np.ndarray, up: int, down: int, h: np.ndarray = None) -> np.ndarray:
    """y[k] = sum_m x[m] * h[half_len + k*down - m*up], k < ceil(len(x)*up/down)."""
    g = math.gcd(up, down)
    up, down = up // g, down // g
    if up == down == 1:
        return np.asarray(x, np.float32).copy()
    h = design_filter(up, down) if h is None else h
    half = (len(h) - 1) // 2
    x = np.asarray(x, np.float64)
    n_out = -(-len(x) * up // down)
    y = np.zeros(n_out, np.float64)
    for k in range(n_out):
        c = k * down
        m_lo = max(0, -(-(c - half) // up))
        m_hi = min(len(x) - 1, (c + half) // up)
        if m_hi >= m_lo:
            m = np.arange(m_lo, m_hi + 1)
            y[k] = np.dot(x[m], h[half + c - m * up])
    return y.astype(np.float32)
