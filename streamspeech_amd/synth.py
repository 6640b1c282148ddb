"""Deterministic synthetic checkpoints in the reference's state-dict key layout.

No released ``streamspeech.*.pt`` / vocoder ``g_00500000`` exists in this environment, so
benchmarks and parity tests run on seeded random weights of the real architecture
(SURVEY.md §8d).  Keys follow the fairseq state dict the reference loads
(``fairseq/checkpoint_utils.py:288`` -> ``model.load_state_dict``; vocoder
``agent/tts/vocoder.py:36-45`` reads ``state["generator"]`` with weight-norm
``weight_g``/``weight_v`` pairs).

The generator is a counter-based hash (splitmix64 -> Box-Muller) so the same
``(seed, key)`` gives bit-identical float32 arrays on any machine / numpy version; golden
fixtures under ``tests/golden`` therefore only need to store reference *outputs*.
"""
import hashlib
from typing import Dict

import numpy as np

from .config import ModelConfig, VocoderConfig

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def _key_seed(seed: int, name: str) -> np.uint64:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.uint64(int.from_bytes(h[:8], "little"))


def uniform01(seed: int, name: str, n: int) -> np.ndarray:
    """n float64 uniforms in (0,1), a pure function of (seed, name, index)."""
    base = _key_seed(seed, name)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(base + idx * np.uint64(0xD1342543DE82EF95))
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def normal(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    u1 = uniform01(seed, name + "/u1", m)
    u2 = uniform01(seed, name + "/u2", m)
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]
    return (z * std + mean).astype(np.float32).reshape(shape)


def uniform(seed: int, name: str, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    return (uniform01(seed, name, n) * (hi - lo) + lo).astype(np.float32).reshape(shape)


class _Builder:
    def __init__(self, seed: int):
        self.seed = seed
        self.sd: Dict[str, np.ndarray] = {}

    def linear(self, name, out_f, in_f, bias=True, gain=1.0):
        self.sd[name + ".weight"] = normal(self.seed, name + ".weight", (out_f, in_f), gain / np.sqrt(in_f))
        if bias:
            self.sd[name + ".bias"] = normal(self.seed, name + ".bias", (out_f,), 0.02)

    def conv(self, name, out_c, in_c, k, bias=True, gain=1.0):
        self.sd[name + ".weight"] = normal(
            self.seed, name + ".weight", (out_c, in_c, k), gain / np.sqrt(in_c * k))
        if bias:
            self.sd[name + ".bias"] = normal(self.seed, name + ".bias", (out_c,), 0.02)

    def layer_norm(self, name, dim):
        self.sd[name + ".weight"] = normal(self.seed, name + ".weight", (dim,), 0.1, 1.0)
        self.sd[name + ".bias"] = normal(self.seed, name + ".bias", (dim,), 0.05)

    def fairseq_mha(self, name, dim, kdim=None):
        kdim = kdim or dim
        self.linear(name + ".q_proj", dim, dim)
        self.linear(name + ".k_proj", dim, kdim)
        self.linear(name + ".v_proj", dim, kdim)
        self.linear(name + ".out_proj", dim, dim)


def make_model_state_dict(seed: int = 0, cfg: ModelConfig = None) -> Dict[str, np.ndarray]:
    """Synthetic ``state["model"]`` of a ``streamspeech`` checkpoint."""
    cfg = cfg or ModelConfig()
    b = _Builder(seed)
    d, f = cfg.enc_dim, cfg.enc_ffn
    # Conv1dSubsampler (reference chunk_unity/modules/convolution.py:48-59)
    b.conv("encoder.subsample.conv_layers.0", cfg.conv_channels, cfg.input_feat, cfg.conv_kernel, gain=1.4)
    b.conv("encoder.subsample.conv_layers.1", 2 * d, cfg.conv_channels // 2, cfg.conv_kernel, gain=1.4)
    b.linear("encoder.linear", d, d, gain=1.0 / 16.0)  # input is scaled by sqrt(256) first
    for i in range(cfg.enc_layers):
        p = f"encoder.conformer_layers.{i}"
        for ffn in ("ffn1", "ffn2"):
            b.layer_norm(f"{p}.{ffn}.layer_norm", d)
            b.linear(f"{p}.{ffn}.w_1", f, d)
            b.linear(f"{p}.{ffn}.w_2", d, f)
        b.layer_norm(f"{p}.self_attn_layer_norm", d)
        for nm in ("linear_q", "linear_k", "linear_v", "linear_out"):
            b.linear(f"{p}.self_attn.{nm}", d, d)
        b.linear(f"{p}.self_attn.linear_pos", d, d, bias=False)
        b.sd[f"{p}.self_attn.pos_bias_u"] = normal(seed, f"{p}.pos_bias_u", (cfg.enc_heads, cfg.head_dim), 0.1)
        b.sd[f"{p}.self_attn.pos_bias_v"] = normal(seed, f"{p}.pos_bias_v", (cfg.enc_heads, cfg.head_dim), 0.1)
        b.layer_norm(f"{p}.conv_module.layer_norm", d)
        b.conv(f"{p}.conv_module.pointwise_conv1", 2 * d, d, 1, bias=False, gain=1.4)
        b.conv(f"{p}.conv_module.depthwise_conv", d, 1, cfg.dw_kernel, bias=False)
        bn = f"{p}.conv_module.batch_norm"
        b.sd[bn + ".weight"] = normal(seed, bn + ".weight", (d,), 0.1, 1.0)
        b.sd[bn + ".bias"] = normal(seed, bn + ".bias", (d,), 0.05)
        b.sd[bn + ".running_mean"] = normal(seed, bn + ".running_mean", (d,), 0.1)
        b.sd[bn + ".running_var"] = uniform(seed, bn + ".running_var", (d,), 0.5, 1.5)
        b.conv(f"{p}.conv_module.pointwise_conv2", d, d, 1, bias=False)
        b.layer_norm(f"{p}.final_layer_norm", d)
    # CTC heads (reference fairseq/models/speech_to_speech/modules/ctc_decoder.py:11-18)
    b.linear("source_unigram_decoder.proj", cfg.src_vocab, d, gain=4.0)
    b.linear("ctc_target_unigram_decoder.proj", cfg.tgt_vocab, d, gain=4.0)
    # MT decoder (4L pre-LN, tied in/out embedding)
    D, F = cfg.dec_dim, cfg.dec_ffn
    emb = normal(seed, "target_unigram_decoder.embed_tokens.weight", (cfg.tgt_vocab, D), D ** -0.5)
    emb[cfg.pad] = 0.0
    b.sd["target_unigram_decoder.embed_tokens.weight"] = emb
    b.sd["target_unigram_decoder.output_projection.weight"] = emb  # tied (same array)
    for i in range(cfg.mt_layers):
        p = f"target_unigram_decoder.layers.{i}"
        b.fairseq_mha(p + ".self_attn", D)
        b.layer_norm(p + ".self_attn_layer_norm", D)
        b.fairseq_mha(p + ".encoder_attn", D, kdim=d)
        b.layer_norm(p + ".encoder_attn_layer_norm", D)
        b.linear(p + ".fc1", F, D)
        b.linear(p + ".fc2", D, F)
        b.layer_norm(p + ".final_layer_norm", D)
    b.layer_norm("target_unigram_decoder.layer_norm", D)
    # T2U encoder (no embeddings; reference ctc_unity/modules/transformer_encoder.py:15-30)
    for i in range(cfg.t2u_layers):
        p = f"synthesizer_encoder.layers.{i}"
        b.fairseq_mha(p + ".self_attn", D)
        b.layer_norm(p + ".self_attn_layer_norm", D)
        b.linear(p + ".fc1", F, D)
        b.linear(p + ".fc2", D, F)
        b.layer_norm(p + ".final_layer_norm", D)
    b.layer_norm("synthesizer_encoder.layer_norm", D)
    # NAR CTC unit decoder (tied output projection to 1005)
    uemb = normal(seed, "decoder.embed_tokens.weight", (cfg.unit_vocab, D), 4.0 * D ** -0.5)
    uemb[cfg.pad] = 0.0
    b.sd["decoder.embed_tokens.weight"] = uemb
    b.sd["decoder.output_projection.weight"] = uemb
    for i in range(cfg.unit_layers):
        p = f"decoder.layers.{i}"
        b.fairseq_mha(p + ".self_attn", D)
        b.layer_norm(p + ".self_attn_layer_norm", D)
        b.fairseq_mha(p + ".encoder_attn", D)
        b.layer_norm(p + ".encoder_attn_layer_norm", D)
        b.linear(p + ".fc1", F, D)
        b.linear(p + ".fc2", D, F)
        b.layer_norm(p + ".final_layer_norm", D)
    b.layer_norm("decoder.layer_norm", D)
    return b.sd


def make_vocoder_state_dict(seed: int = 0, cfg: VocoderConfig = None) -> Dict[str, np.ndarray]:
    """Synthetic ``state["generator"]`` of the unit HiFi-GAN (weight-norm g/v pairs kept, the
    loader folds them exactly as ``remove_weight_norm`` does, reference hifigan.py:172-179)."""
    cfg = cfg or VocoderConfig()
    sd: Dict[str, np.ndarray] = {}

    def wn_conv(name, shape, g_mean, transposed=False):
        # weight_norm(dim=0): g has one entry per index of dim 0 (Cout for Conv1d, Cin for ConvTranspose1d)
        sd[name + ".weight_v"] = normal(seed, name + ".weight_v", shape, 1.0)
        sd[name + ".weight_g"] = normal(seed, name + ".weight_g", (shape[0], 1, 1), 0.05 * g_mean, g_mean)
        nb = shape[1] if transposed else shape[0]
        sd[name + ".bias"] = normal(seed, name + ".bias", (nb,), 0.02)

    E, H = cfg.embedding_dim, cfg.dur_hidden
    sd["dict.weight"] = normal(seed, "dict.weight", (cfg.num_embeddings, E), 1.0)
    k = cfg.dur_kernel
    sd["dur_predictor.conv1.0.weight"] = normal(seed, "dur.conv1.w", (H, E, k), 1.4 / np.sqrt(E * k))
    sd["dur_predictor.conv1.0.bias"] = normal(seed, "dur.conv1.b", (H,), 0.02)
    sd["dur_predictor.ln1.weight"] = normal(seed, "dur.ln1.w", (H,), 0.1, 1.0)
    sd["dur_predictor.ln1.bias"] = normal(seed, "dur.ln1.b", (H,), 0.05)
    sd["dur_predictor.conv2.0.weight"] = normal(seed, "dur.conv2.w", (H, H, k), 1.4 / np.sqrt(H * k))
    sd["dur_predictor.conv2.0.bias"] = normal(seed, "dur.conv2.b", (H,), 0.02)
    sd["dur_predictor.ln2.weight"] = normal(seed, "dur.ln2.w", (H,), 0.1, 1.0)
    sd["dur_predictor.ln2.bias"] = normal(seed, "dur.ln2.b", (H,), 0.05)
    sd["dur_predictor.proj.weight"] = normal(seed, "dur.proj.w", (1, H), 0.6 / np.sqrt(H))
    sd["dur_predictor.proj.bias"] = np.full((1,), 0.75, np.float32)

    C0 = cfg.upsample_initial_channel
    wn_conv("conv_pre", (C0, cfg.model_in_dim, 7), 1.0)
    for i, (u, ku) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        cin, cout = C0 // (2 ** i), C0 // (2 ** (i + 1))
        # ConvTranspose1d weight is [Cin, Cout, k]; weight_norm(dim=0) -> g is [Cin,1,1]
        wn_conv(f"ups.{i}", (cin, cout, ku), 1.6 * np.sqrt(u / 2.0), transposed=True)
        nk = len(cfg.resblock_kernel_sizes)
        for j, (kr, dil) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            for di in range(len(dil)):
                wn_conv(f"resblocks.{i * nk + j}.convs1.{di}", (cout, cout, kr), 1.0)
                wn_conv(f"resblocks.{i * nk + j}.convs2.{di}", (cout, cout, kr), 0.6)
    wn_conv("conv_post", (1, C0 // (2 ** len(cfg.upsample_rates)), 7), 0.25)
    return sd


def synth_fbank(seed: int, n_frames: int, feat: int = 80) -> np.ndarray:
    """Post-CMVN fbank statistics N(0,1) (SURVEY.md §8d synthetic inputs)."""
    return normal(seed, f"fbank/{n_frames}", (n_frames, feat), 1.0)


def synth_pcm(seed: int, n_samples: int) -> np.ndarray:
    """16 kHz PCM, N(0, 0.05^2) clipped to +-1."""
    return np.clip(normal(seed, f"pcm/{n_samples}", (n_samples,), 0.05), -1.0, 1.0)


def synth_durations(seed: int, n: int) -> np.ndarray:
    """Utterance durations clip(LogNormal(ln 4.5, 0.45), 1, 15) seconds."""
    z = normal(seed, f"durations/{n}", (n,), 0.45, float(np.log(4.5))).astype(np.float64)
    return np.clip(np.exp(z), 1.0, 15.0)
