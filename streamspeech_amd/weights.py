"""fairseq state dict  ->  packed FP32 weight blob for the HIP library.

Ingests the key layout of the reference checkpoints (SURVEY.md §8b "Checkpoint ingest",
Appendix A) and produces ``(names, offsets, numels, blob)`` for ``ss_model_create`` /
``ss_vocoder_create``.  Layout transforms done ONCE here (all exact in FP32):

* Conv1d ``[Cout, Cin, k]`` -> tap-major ``[Cout, k*Cin]`` (implicit-GEMM K order).
* GLU producers (subsampler convs, conformer pointwise_conv1): output channels re-ordered in
  blocks of 16 value rows followed by their 16 gate rows, so value and gate of a channel land in
  the same lane of adjacent MFMA tiles (csrc/gemm.hip epilogue).
* q/k/v projections stacked into one ``[3D, D]`` matrix; fairseq-MHA q rows (and bias) are
  pre-multiplied by head_dim^-0.5 = 0.125 (a power of two: same bits as ``q *= scaling``,
  reference ctc_unity/modules/multihead_attention.py:563).
* ``encoder.linear`` weight pre-multiplied by sqrt(256) = 16 (``x = embed_scale * x`` before the
  Linear, reference chunk_unity/models/s2t_conformer.py:127,139).
* depthwise conv ``[C,1,k]`` -> ``[k, C]``.
* ConvTranspose1d ``[Cin, Cout, k]`` (stride s, padding (k-s)/2) -> 3-tap polyphase conv
  ``[s*Cout, 3*Cin]``: out[q*s+r] = sum_{jj<3} in[q+jj-1] . W[(1-jj)*s + r + p]  (zero where the
  tap index falls outside [0,k)).
* weight-norm pairs folded with ``torch._weight_norm`` -- the op ``remove_weight_norm`` uses
  (reference fairseq/models/text_to_speech/hifigan.py:172-179).
* sinusoid tables built with the same torch ops as the reference
  (fairseq/modules/positional_encoding.py:94-111, sinusoidal_positional_embedding.py:43-64).
"""
import math
from typing import Dict, List, Tuple

import numpy as np
import torch

from .config import ModelConfig, VocoderConfig


def _t(x) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x.detach().float().cpu()
    return torch.from_numpy(np.ascontiguousarray(x)).float()


def glu_interleave(w: torch.Tensor) -> torch.Tensor:
    """rows [value(H) ; gate(H)] -> blocks of [16 value | 16 gate]."""
    H = w.shape[0] // 2
    assert H % 16 == 0
    val = w[:H].reshape(H // 16, 16, *w.shape[1:])
    gate = w[H:].reshape(H // 16, 16, *w.shape[1:])
    return torch.cat([val, gate], dim=1).reshape(w.shape)


def conv_tap_major(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, k] -> [Cout, k*Cin]."""
    return w.permute(0, 2, 1).contiguous().reshape(w.shape[0], -1)


def convT_polyphase(w: torch.Tensor, b: torch.Tensor, stride: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """ConvTranspose1d weight [Cin, Cout, k] -> ([s*Cout, 3*Cin], bias [s*Cout])."""
    cin, cout, k = w.shape
    assert (k - stride) % 2 == 0
    p = (k - stride) // 2
    out = torch.zeros(stride, cout, 3, cin)
    for r in range(stride):
        for jj in range(3):
            j = (1 - jj) * stride + r + p
            if 0 <= j < k:
                out[r, :, jj, :] = w[:, :, j].t()
    # 3 taps must cover every kernel index exactly once per phase
    return out.reshape(stride * cout, 3 * cin).contiguous(), b.repeat(stride).contiguous()


def rel_pos_table(tmax: int, d: int) -> torch.Tensor:
    """[2*tmax-1, d]; row m <-> relative offset tmax-1-m (positional_encoding.py:94-111)."""
    pe_positive = torch.zeros(tmax, d)
    pe_negative = torch.zeros(tmax, d)
    position = torch.arange(0, tmax, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe_positive[:, 0::2] = torch.sin(position * div_term)
    pe_positive[:, 1::2] = torch.cos(position * div_term)
    pe_negative[:, 0::2] = torch.sin(-1 * position * div_term)
    pe_negative[:, 1::2] = torch.cos(-1 * position * div_term)
    return torch.cat([torch.flip(pe_positive, [0]), pe_negative[1:]], dim=0)


def sinusoid_table(n: int, dim: int, padding_idx: int) -> torch.Tensor:
    """sinusoidal_positional_embedding.py:43-64."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(n, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(n, -1)
    e[padding_idx, :] = 0
    return e


def povey_window() -> torch.Tensor:
    i = np.arange(400, dtype=np.float64)
    return torch.from_numpy(((0.5 - 0.5 * np.cos(2 * math.pi * i / 399)) ** 0.85).astype(np.float32))


def mel_banks() -> torch.Tensor:
    """[80, 257] Kaldi triangular mel weights, 20 Hz .. 8 kHz."""
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    mlow, mhigh = mel(20.0), mel(8000.0)
    delta = (mhigh - mlow) / 81
    b = np.arange(80, dtype=np.float64)[:, None]
    left = mlow + b * delta
    center, right = left + delta, left + 2 * delta
    m = mel(31.25 * np.arange(256, dtype=np.float64))[None, :]
    w = np.maximum(0.0, np.minimum((m - left) / (center - left), (right - m) / (right - center)))
    return torch.from_numpy(np.pad(w, ((0, 0), (0, 1))).astype(np.float32))


class Packer:
    def __init__(self):
        self.names: List[str] = []
        self.tensors: List[torch.Tensor] = []

    def add(self, name: str, t: torch.Tensor):
        assert name not in self.names, name
        self.names.append(name)
        self.tensors.append(t.contiguous().float().reshape(-1))

    def finish(self):
        offsets, numels, off = [], [], 0
        for t in self.tensors:
            offsets.append(off)
            numels.append(t.numel())
            off += (t.numel() + 63) // 64 * 64       # 256-byte aligned slots (float4 loads)
        blob = torch.zeros(off, dtype=torch.float32)
        for o, t in zip(offsets, self.tensors):
            blob[o:o + t.numel()] = t
        return self.names, offsets, numels, blob


def _ln(pk, name, sd, key):
    pk.add(name + ".g", _t(sd[key + ".weight"]))
    pk.add(name + ".b", _t(sd[key + ".bias"]))


def _lin(pk, name, sd, key, bias=True, scale=1.0):
    pk.add(name + ".w", _t(sd[key + ".weight"]) * scale)
    if bias:
        pk.add(name + ".b", _t(sd[key + ".bias"]) * scale)


def _fairseq_layers(pk: Packer, sd, src: str, dst: str, n: int, cross: bool):
    for l in range(n):
        p, q = f"{src}.layers.{l}", f"{dst}.L{l}"
        _ln(pk, q + ".self.ln", sd, p + ".self_attn_layer_norm")
        w = torch.cat([_t(sd[p + ".self_attn.q_proj.weight"]) * 0.125, _t(sd[p + ".self_attn.k_proj.weight"]),
                       _t(sd[p + ".self_attn.v_proj.weight"])], 0)
        b = torch.cat([_t(sd[p + ".self_attn.q_proj.bias"]) * 0.125, _t(sd[p + ".self_attn.k_proj.bias"]),
                       _t(sd[p + ".self_attn.v_proj.bias"])], 0)
        pk.add(q + ".self.qkv.w", w)
        pk.add(q + ".self.qkv.b", b)
        _lin(pk, q + ".self.out", sd, p + ".self_attn.out_proj")
        if cross:
            _ln(pk, q + ".cross.ln", sd, p + ".encoder_attn_layer_norm")
            _lin(pk, q + ".cross.q", sd, p + ".encoder_attn.q_proj", scale=0.125)
            pk.add(q + ".cross.kv.w", torch.cat([_t(sd[p + ".encoder_attn.k_proj.weight"]),
                                                 _t(sd[p + ".encoder_attn.v_proj.weight"])], 0))
            pk.add(q + ".cross.kv.b", torch.cat([_t(sd[p + ".encoder_attn.k_proj.bias"]),
                                                 _t(sd[p + ".encoder_attn.v_proj.bias"])], 0))
            _lin(pk, q + ".cross.out", sd, p + ".encoder_attn.out_proj")
        _ln(pk, q + ".ffn.ln", sd, p + ".final_layer_norm")
        _lin(pk, q + ".fc1", sd, p + ".fc1")
        _lin(pk, q + ".fc2", sd, p + ".fc2")


def pack_model(sd: Dict, cfg: ModelConfig, cmvn_mean=None, cmvn_std=None, max_rel_pos: int = 2048,
               max_tgt_pos: int = 1026):
    """state["model"] of a ``streamspeech`` checkpoint -> (names, offsets, numels, blob)."""
    assert cfg.head_dim == 64 and cfg.dec_dim // cfg.dec_heads == 64
    pk = Packer()
    d = cfg.enc_dim
    for i in range(2):
        k = f"encoder.subsample.conv_layers.{i}"
        pk.add(f"enc.sub{i}.w", conv_tap_major(glu_interleave(_t(sd[k + ".weight"]))))
        pk.add(f"enc.sub{i}.b", glu_interleave(_t(sd[k + ".bias"])))
    pk.add("enc.linear.w", _t(sd["encoder.linear.weight"]) * math.sqrt(d))
    pk.add("enc.linear.b", _t(sd["encoder.linear.bias"]))
    pk.add("enc.pos_table", rel_pos_table(max_rel_pos, d))
    pk.add("enc.pos_w", torch.cat([_t(sd[f"encoder.conformer_layers.{l}.self_attn.linear_pos.weight"])
                                   for l in range(cfg.enc_layers)], 0))
    for l in range(cfg.enc_layers):
        p, q = f"encoder.conformer_layers.{l}", f"enc.L{l}"
        for ffn in ("ffn1", "ffn2"):
            _ln(pk, f"{q}.{ffn}.ln", sd, f"{p}.{ffn}.layer_norm")
            _lin(pk, f"{q}.{ffn}.w1", sd, f"{p}.{ffn}.w_1")
            _lin(pk, f"{q}.{ffn}.w2", sd, f"{p}.{ffn}.w_2")
        _ln(pk, q + ".attn.ln", sd, p + ".self_attn_layer_norm")
        pk.add(q + ".attn.qkv.w", torch.cat([_t(sd[f"{p}.self_attn.linear_{n}.weight"]) for n in "qkv"], 0))
        pk.add(q + ".attn.qkv.b", torch.cat([_t(sd[f"{p}.self_attn.linear_{n}.bias"]) for n in "qkv"], 0))
        _lin(pk, q + ".attn.out", sd, p + ".self_attn.linear_out")
        pk.add(q + ".attn.u", _t(sd[p + ".self_attn.pos_bias_u"]))
        pk.add(q + ".attn.v", _t(sd[p + ".self_attn.pos_bias_v"]))
        _ln(pk, q + ".conv.ln", sd, p + ".conv_module.layer_norm")
        pk.add(q + ".conv.pw1.w", glu_interleave(_t(sd[p + ".conv_module.pointwise_conv1.weight"])[:, :, 0]))
        pk.add(q + ".conv.dw.wt", _t(sd[p + ".conv_module.depthwise_conv.weight"])[:, 0, :].t())
        bn = p + ".conv_module.batch_norm"
        pk.add(q + ".conv.bn.mean", _t(sd[bn + ".running_mean"]))
        pk.add(q + ".conv.bn.var", _t(sd[bn + ".running_var"]))
        pk.add(q + ".conv.bn.g", _t(sd[bn + ".weight"]))
        pk.add(q + ".conv.bn.b", _t(sd[bn + ".bias"]))
        pk.add(q + ".conv.pw2.w", _t(sd[p + ".conv_module.pointwise_conv2.weight"])[:, :, 0])
        _ln(pk, q + ".final_ln", sd, p + ".final_layer_norm")
    _lin(pk, "ctc.asr", sd, "source_unigram_decoder.proj")
    _lin(pk, "ctc.st", sd, "ctc_target_unigram_decoder.proj")
    # front-end constants
    pk.add("fe.window", povey_window())
    pk.add("fe.melw", mel_banks())
    pk.add("fe.cmvn_mean", _t(cmvn_mean) if cmvn_mean is not None else torch.zeros(80))
    pk.add("fe.cmvn_std", _t(cmvn_std) if cmvn_std is not None else torch.ones(80))
    # MT decoder
    D = cfg.dec_dim
    pk.add("mt.emb", _t(sd["target_unigram_decoder.embed_tokens.weight"]))
    pk.add("mt.pos_table", sinusoid_table(max_tgt_pos, D, cfg.pad))
    _fairseq_layers(pk, sd, "target_unigram_decoder", "mt", cfg.mt_layers, True)
    _ln(pk, "mt.ln", sd, "target_unigram_decoder.layer_norm")
    _fairseq_layers(pk, sd, "synthesizer_encoder", "t2u", cfg.t2u_layers, False)
    _ln(pk, "t2u.ln", sd, "synthesizer_encoder.layer_norm")
    _fairseq_layers(pk, sd, "decoder", "unit", cfg.unit_layers, True)
    _ln(pk, "unit.ln", sd, "decoder.layer_norm")
    out_key = "decoder.output_projection.weight" if "decoder.output_projection.weight" in sd \
        else "decoder.embed_tokens.weight"
    pk.add("unit.out.w", _t(sd[out_key]))
    pk.add("unit.pos_row", sinusoid_table(cfg.pad + 2, D, cfg.pad)[cfg.pad + 1])
    return pk.finish()


def fold_weight_norm(sd: Dict, name: str) -> torch.Tensor:
    if name + ".weight" in sd:
        return _t(sd[name + ".weight"])
    return torch._weight_norm(_t(sd[name + ".weight_v"]), _t(sd[name + ".weight_g"]), 0)


def pack_vocoder(vsd: Dict, vcfg: VocoderConfig):
    """state["generator"] of the unit HiFi-GAN -> (names, offsets, numels, blob)."""
    pk = Packer()
    pk.add("voc.dict", _t(vsd["dict.weight"]))
    pk.add("voc.dur.conv1.w", conv_tap_major(_t(vsd["dur_predictor.conv1.0.weight"])))
    pk.add("voc.dur.conv1.b", _t(vsd["dur_predictor.conv1.0.bias"]))
    _ln(pk, "voc.dur.ln1", vsd, "dur_predictor.ln1")
    pk.add("voc.dur.conv2.w", conv_tap_major(_t(vsd["dur_predictor.conv2.0.weight"])))
    pk.add("voc.dur.conv2.b", _t(vsd["dur_predictor.conv2.0.bias"]))
    _ln(pk, "voc.dur.ln2", vsd, "dur_predictor.ln2")
    pk.add("voc.dur.proj.w", _t(vsd["dur_predictor.proj.weight"]))
    pk.add("voc.dur.proj.b", _t(vsd["dur_predictor.proj.bias"]))
    pk.add("voc.pre.w", conv_tap_major(fold_weight_norm(vsd, "conv_pre")))
    pk.add("voc.pre.b", _t(vsd["conv_pre.bias"]))
    nk = len(vcfg.resblock_kernel_sizes)
    for i, u in enumerate(vcfg.upsample_rates):
        w, b = convT_polyphase(fold_weight_norm(vsd, f"ups.{i}"), _t(vsd[f"ups.{i}.bias"]), u)
        pk.add(f"voc.up{i}.w", w)
        pk.add(f"voc.up{i}.b", b)
        for j in range(nk):
            for dd in range(3):
                for c in ("1", "2"):
                    src = f"resblocks.{i * nk + j}.convs{c}.{dd}"
                    pk.add(f"voc.rb{i * nk + j}.c{c}.{dd}.w", conv_tap_major(fold_weight_norm(vsd, src)))
                    pk.add(f"voc.rb{i * nk + j}.c{c}.{dd}.b", _t(vsd[src + ".bias"]))
    pk.add("voc.post.w", conv_tap_major(fold_weight_norm(vsd, "conv_post")).reshape(-1))
    pk.add("voc.post.b", _t(vsd["conv_post.bias"]))
    return pk.finish()
