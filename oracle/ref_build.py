"""Assemble the reference's OWN modules (loaded by oracle/ref_loader.py) in the released
architecture and fill them with a synthetic state dict.  TEST INFRASTRUCTURE; needs
/root/reference, so it only runs in the build container (fixture generation + CPU pin tests).
"""
from argparse import Namespace
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from . import ref_loader


class _Dict:
    """Stand-in for fairseq Dictionary: bos=0 pad=1 eos=2 unk=3 (fairseq/data/dictionary.py)."""

    def __init__(self, n, blank_index=None):
        self.n = n
        self.pad_index, self.eos_index, self.unk_index, self.bos_index = 1, 2, 3, 0
        self.blank_index = blank_index

    def __len__(self):
        return self.n

    def pad(self):
        return 1

    def eos(self):
        return 2

    def unk(self):
        return 3

    def bos(self):
        return 0


def _load(module: nn.Module, sd, prefix: str, strict=True):
    sub = {k[len(prefix):]: torch.from_numpy(np.ascontiguousarray(v)).clone()
           for k, v in sd.items() if k.startswith(prefix)}
    own = module.state_dict()
    # non-persistent / bookkeeping buffers the synthetic dict does not carry
    for k in list(own.keys()):
        if k not in sub and (k.endswith("num_batches_tracked") or k.endswith("version")
                             or k.endswith("_float_tensor")):
            sub[k] = own[k]
    missing, unexpected = module.load_state_dict(sub, strict=False)
    if strict:
        assert not missing and not unexpected, (missing, unexpected)
    module.eval()
    return module


def encoder_args(cfg, chunk_size):
    return Namespace(
        encoder_freezing_updates=0, encoder_embed_dim=cfg.enc_dim, no_scale_embedding=False,
        conv_version="s2t_transformer", chunk_size=chunk_size,
        input_feat_per_channel=cfg.input_feat, input_channels=1, conv_channels=cfg.conv_channels,
        conv_kernel_sizes=f"{cfg.conv_kernel},{cfg.conv_kernel}", pos_enc_type="rel_pos",
        max_source_positions=cfg.max_source_positions, dropout=0.1,
        encoder_ffn_embed_dim=cfg.enc_ffn, encoder_attention_heads=cfg.enc_heads,
        depthwise_conv_kernel_size=cfg.dw_kernel, attn_type="espnet", fp16=False,
        encoder_layers=cfg.enc_layers, uni_encoder=False)


def build_encoder(sd, cfg, attn_chunk, conv_chunk):
    """ChunkS2TConformerEncoder with the chunk sizes the agent imposes
    (agent/speech_to_speech.streamspeech.agent.py:395-413)."""
    R = ref_loader.load()
    enc = R.ChunkS2TConformerEncoder(encoder_args(cfg, 8))
    _load(enc, sd, "encoder.")
    enc.chunk_size = attn_chunk
    for conv in enc.subsample.conv_layers:
        conv.chunk_size = conv_chunk
    for layer in enc.conformer_layers:
        layer.conv_module.depthwise_conv.chunk_size = conv_chunk
    return enc


def decoder_args(cfg, layers, enc_dim):
    return Namespace(
        decoder_embed_dim=cfg.dec_dim, decoder_ffn_embed_dim=cfg.dec_ffn, decoder_layers=layers,
        decoder_attention_heads=cfg.dec_heads, decoder_normalize_before=True,
        encoder_embed_dim=enc_dim, activation_fn="relu", share_decoder_input_output_embed=True,
        max_target_positions=cfg.max_target_positions if layers == cfg.unit_layers else 1024,
        n_frames_per_step=1, ctc_upsample_rate=cfg.ctc_upsample, dropout=0.0)


def build_mt_decoder(sd, cfg):
    R = ref_loader.load()
    emb = nn.Embedding(cfg.tgt_vocab, cfg.dec_dim, padding_idx=cfg.pad)
    dec = R.TransformerDecoder(decoder_args(cfg, cfg.mt_layers, cfg.enc_dim), _Dict(cfg.tgt_vocab), emb)
    return _load(dec, sd, "target_unigram_decoder.")


def build_t2u_encoder(sd, cfg, uni=False):
    R = ref_loader.load()
    args = Namespace(encoder_layers=cfg.t2u_layers, encoder_embed_dim=cfg.dec_dim,
                     encoder_ffn_embed_dim=cfg.dec_ffn, encoder_attention_heads=cfg.dec_heads,
                     encoder_normalize_before=True, activation_fn="relu", uni_encoder=uni, dropout=0.0)
    return _load(R.UniTransformerEncoderNoEmb(args), sd, "synthesizer_encoder.")


def build_unit_decoder(sd, cfg):
    R = ref_loader.load()
    emb = R.StackedEmbedding(cfg.unit_vocab, cfg.dec_dim, cfg.pad, num_stacked=1)
    dec = R.CTCTransformerUnitDecoder(decoder_args(cfg, cfg.unit_layers, cfg.dec_dim),
                                      _Dict(cfg.unit_vocab, cfg.unit_blank), emb)
    return _load(dec, sd, "decoder.")


def build_ctc_head(sd, cfg, name):
    R = ref_loader.load()
    n = cfg.src_vocab if name == "source_unigram" else cfg.tgt_vocab
    return _load(R.CTCDecoder(_Dict(n), cfg.enc_dim), sd, f"{name}_decoder.")


def build_vocoder(vsd, vcfg):
    """CodeGenerator + remove_weight_norm, as CodeHiFiGANVocoderWithDur.__init__ does
    (agent/tts/vocoder.py:36-45)."""
    R = ref_loader.load()
    gen = R.CodeGenerator(vcfg.as_dict())
    _load(gen, vsd, "")
    gen.remove_weight_norm()
    return gen
