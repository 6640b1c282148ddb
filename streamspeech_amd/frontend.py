"""OnlineFeatureExtractor of the agent (reference agent/speech_to_speech.streamspeech.agent.py:43-98)
over the fused fbank+CMVN HIP kernel, plus the waveform front-end.

The reference resamples the whole 48 kHz sample history to 16 kHz with sox ("rate", via
torchaudio.sox_effects, fairseq/data/audio/audio_utils.py:53-62) -- third-party arithmetic outside
the parity contract (BASELINE.json: "on the same fbank input").  Here a zero-phase
polyphase FIR with the published design of scipy.signal.resample_poly runs on the device
(ss_resample, csrc/fbank.hip) when the source is not already 16 kHz.
"""
import array
import math
import wave

import numpy as np
import torch


def design_filter(up: int, down: int) -> np.ndarray:
    """Low-pass of the polyphase resampler, float64 [2*10*max(up,down)+1]: windowed sinc, cutoff
    1/max(up,down) of Nyquist, Kaiser beta 5, unit DC gain, times `up` (the design
    scipy.signal.resample_poly documents; `up`/`down` in lowest terms)."""
    max_rate = max(up, down)
    half_len = 10 * max_rate
    n = np.arange(2 * half_len + 1, dtype=np.float64) - half_len
    fc = 1.0 / max_rate
    h = fc * np.sinc(fc * n) * np.kaiser(2 * half_len + 1, 5.0)
    return h / h.sum() * up


def unsettled_fbank_frames(sr_in: int, sr_out: int = 16000, shift_samples: int = 160) -> int:
    """How many of the newest fbank frames may still change when more audio arrives: the resampler's output samples whose
    FIR window reaches past the end of the received input (zero-padded there) are recomputed on the next call -- that edge
    is half_len / down output samples of design_filter's low-pass -- and every fbank frame that overlaps it is unsettled.
    0 when nothing is resampled.  (ss_encoder_stream_set_tail: such frames must not be cached as final.)"""
    import math
    g = math.gcd(int(sr_in), int(sr_out))
    up, down = int(sr_out) // g, int(sr_in) // g
    if up == down:
        return 0
    half_len = (len(design_filter(up, down)) - 1) // 2
    edge = math.ceil(half_len / down)                       # output samples that still see zero padding
    return max(1, math.ceil(edge / shift_samples))


def read_wav(path: str):
    """PCM WAV (8/16/32-bit integer) -> (float32 mono samples in [-1, 1), sample rate): the `list[float]`
    the SimulEval dataloader hands the agent (SimulEval/simuleval/data/dataloader/s2t_dataloader.py).
    MP3 (example/wavs/*.mp3) needs a decoder this image does not have; convert to WAV first."""
    if str(path).lower().endswith(".mp3"):
        raise IOError("no MP3 decoder is available here; convert %s to PCM WAV" % path)
    with wave.open(str(path), "rb") as w:
        sr, nch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, "<i4").astype(np.float32) / 2147483648.0
    elif sw == 1:
        x = (np.frombuffer(raw, "u1").astype(np.float32) - 128.0) / 128.0
    else:
        raise IOError("unsupported WAV sample width %d" % sw)
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)          # convert_waveform(to_mono=True): channel mean
    return x, sr


def write_wav(path: str, samples, sr: int = 16000):
    """float samples in [-1, 1] -> 16-bit PCM WAV (generate_waveform_from_code.py dumps soundfile PCM_16)."""
    x = np.clip(np.asarray(samples, np.float32), -1.0, 1.0)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(int(sr))
        w.writeframes(np.round(x * 32767.0).astype("<i2").tobytes())

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
        self._np = np.zeros(0, np.float32)
        self._buf = np.zeros(0, np.float32)      # backing store of _np (grows by doubling: appending a segment does not copy the history)
        self._dev = None
        self._n_dev = 0
        self._fb = None                          # fbank rows of the cached history (16-kHz sources on the HIP engine; see __call__)
        self._n_fb = 0
        self._src_id = None

    def _samples(self, samples, n):
        """float32 array of samples[:n].  SimulEval hands the WHOLE sample history as a Python list at every policy() call
        (states.source only grows within an utterance); converting 15 s of floats costs ~10 ms per call, so only the new
        tail is converted and the rest comes from the cache.  The cache belongs to ONE source list: it is dropped when
        another list object comes in (AgentStates.reset() starts a new list, SimulEval/simuleval/agents/states.py), when
        the history shrank, or when a spot check of eight cached values (compared as float32, the cache's type) fails.
        The agents also call clear_cache() in reset(); a caller that reuses an extractor for unrelated audio must do the same."""
        c = getattr(self, "_np", None)
        if c is None:
            self.clear_cache()
            c = self._np
        k = len(c)
        ok = k <= n and (k == 0 or getattr(self, "_src_id", None) == id(samples))
        if ok and k:
            for i in {0, k - 1, k // 2, k // 3, k // 5, (2 * k) // 3, (4 * k) // 5, k // 7}:
                if np.float32(samples[i]) != c[i]:
                    ok = False
                    break
        if not ok:
            self.clear_cache()
            c, k = self._np, 0
        self._src_id = id(samples)
        if n > k:
            # Python floats are doubles: array('d') takes the list in one C loop (1.4x faster than np.asarray(list)), the cast to float32
            # rounds exactly as np.asarray(list, float32) does
            new = np.frombuffer(array.array("d", samples[k:n]), dtype=np.float64)
            buf = getattr(self, "_buf", None)
            if buf is None or len(buf) < n or (k and buf.ctypes.data != c.ctypes.data):
                buf = np.empty(max(2 * n, 1 << 15), np.float32)
                buf[:k] = c
                self._buf = buf
            buf[k:n] = new
            c = buf[:n]
            self._np = c
        return c[:n]

    def __call__(self, new_samples, sr=None):
        sr = sr or self.sample_rate
        samples = new_samples
        num_frames = math.floor(
            (len(samples) - self.len_ms_to_samples(self.window_size - self.shift_size)) / self.num_samples_per_shift)
        if num_frames <= 0:
            return torch.empty((0, self.feature_dim), device=self.engine.device)
        effective = int(num_frames * self.len_ms_to_samples(self.shift_size)
                        + self.len_ms_to_samples(self.window_size - self.shift_size))
        x = self._samples(samples, effective)
        # device copy of the history: only the new samples cross PCIe
        dev = self.engine.device
        if getattr(self, "_dev", None) is None or self._dev.numel() < effective or self._n_dev > effective:
            cap = max(2 * effective, 1 << 16)
            buf = torch.empty((cap,), dtype=torch.float32, device=dev)
            if getattr(self, "_dev", None) is not None and 0 < self._n_dev <= effective:
                buf[: self._n_dev] = self._dev[: self._n_dev]
            else:
                self._n_dev = 0
                self._n_fb = 0
            self._dev = buf
        if effective > self._n_dev:
            self._dev[self._n_dev:effective] = torch.from_numpy(x[self._n_dev:effective]).to(dev)
            self._n_dev = effective
        pcm = self._dev[:effective]
        if sr != SAMPLE_RATE:
            pcm = self.engine.resample(pcm, int(sr), SAMPLE_RATE)
            return self.engine.fbank_cmvn(pcm, 32768.0)
        if not hasattr(self.engine, "lib"):              # the CPU oracle engine: as the reference, everything every time
            return self.engine.fbank_cmvn(pcm, 32768.0)
        # A fbank row is a function of ITS 400 samples only (one workgroup per 25-ms frame, global CMVN): rows of the cached history stay
        # as they are and only the new frames are computed, into a buffer that grows by doubling.  Same bits as the full call
        # (tests/test_stages_gpu.py); the reference recomputes all frames per policy() call (agent :66-98).
        nf = int(num_frames)
        k = self._n_fb if self._fb is not None else 0
        if k > nf:
            k = 0
        if self._fb is None or self._fb.shape[0] < nf:
            fb = torch.empty((max(2 * nf, 512), self.feature_dim), dtype=torch.float32, device=dev)
            if k:
                fb[:k] = self._fb[:k]
            self._fb = fb
        if nf > k:
            self.engine.fbank_cmvn(self._dev[k * self.num_samples_per_shift:effective], 32768.0, out=self._fb[k:nf])
        self._n_fb = nf
        return self._fb[:nf]
