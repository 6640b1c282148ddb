"""OnlineFeatureExtractor of the agent (reference agent/speech_to_speech.streamspeech.agent.py:43-98)
over the fused fbank+CMVN HIP kernel.

The reference resamples the whole 48 kHz sample history to 16 kHz with sox ("rate", via
torchaudio.sox_effects, fairseq/data/audio/audio_utils.py:53-62) -- third-party arithmetic outside
the parity contract (BASELINE.json: "on the same fbank input").  Here a polyphase
FIR (scipy.signal.resample_poly) stands in on the host when the source is not already 16 kHz.
"""
import math

import numpy as np
import torch

SHIFT_SIZE, WINDOW_SIZE, ORG_SAMPLE_RATE, SAMPLE_RATE, FEATURE_DIM = 10, 25, 48000, 16000, 80


class OnlineFeatureExtractor:
    def __init__(self, args, engine):
        self.shift_size = args.shift_size
        self.window_size = args.window_size
        assert self.window_size >= self.shift_size
        self.sample_rate = args.sample_rate
        self.feature_dim = args.feature_dim
        self.num_samples_per_shift = int(self.shift_size * self.sample_rate / 1000)
        self.num_samples_per_window = int(self.window_size * self.sample_rate / 1000)
        self.len_ms_to_samples = lambda x: x * self.sample_rate / 1000
        self.engine = engine

    def clear_cache(self):
        pass

    def __call__(self, new_samples, sr=None):
        sr = sr or self.sample_rate
        samples = new_samples
        num_frames = math.floor(
            (len(samples) - self.len_ms_to_samples(self.window_size - self.shift_size)) / self.num_samples_per_shift)
        if num_frames <= 0:
            return torch.empty((0, self.feature_dim), device=self.engine.device)
        effective = int(num_frames * self.len_ms_to_samples(self.shift_size)
                        + self.len_ms_to_samples(self.window_size - self.shift_size))
        x = np.asarray(samples[:effective], dtype=np.float32)
        if sr != SAMPLE_RATE:
            from scipy.signal import resample_poly
            g = math.gcd(int(sr), SAMPLE_RATE)
            x = resample_poly(x, SAMPLE_RATE // g, int(sr) // g).astype(np.float32)
        pcm = torch.from_numpy(x).to(self.engine.device)
        return self.engine.fbank_cmvn(pcm, 32768.0)
