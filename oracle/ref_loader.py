"""Loads the reference's own hot-path module FILES from /root/reference for oracle pinning.

TEST INFRASTRUCTURE.  ``import fairseq`` fails in this image (no omegaconf/hydra/bitarray), but
every hot-path module file is pure torch; this loader installs a minimal ``sys.modules`` stub
tree for the few fairseq symbols those files import and then executes the reference files
*where they lie* (nothing is copied into the repo).  Used only by ``oracle/make_golden.py``
(fixture generation, in the build container) and by CPU tests that are skipped when
/root/reference is absent (it does not exist on the GPU box).

Stubbed symbols and what they stand for (all eval-mode equivalents):
  fairseq.modules.LayerNorm            -> torch.nn.LayerNorm(eps=1e-5)  (fairseq/modules/layer_norm.py:28-33, non-apex branch)
  fairseq.modules.FairseqDropout       -> identity in eval
  fairseq.modules.quant_noise          -> identity (p = 0)
  fairseq.utils.get_activation_fn      -> swish: nn.SiLU, relu: F.relu (fairseq/utils.py)
  fairseq.utils.softmax                -> F.softmax(dtype=float32)
  fairseq.models.transformer.TransformerConfig.from_namespace -> nested namespace view
"""
import importlib.util
import math
import os
import sys
import types
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

REF = os.environ.get("STREAMSPEECH_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "researches", "ctc_unity"))


class _AutoStub(types.ModuleType):
    """A module whose unknown attributes are inert nn.Module subclasses (so ``class X(Stub)`` works)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (nn.Module,), {"__init__": lambda self, *a, **k: nn.Module.__init__(self)})
        setattr(self, name, cls)
        return cls


def _mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None or not isinstance(m, _AutoStub):
        m = _AutoStub(name)
        m.__path__ = []  # behave as a package
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(_mod(parent), child, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _load_file(modname, relpath):
    path = os.path.join(REF, relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    if "." in modname:
        parent, child = modname.rsplit(".", 1)
        setattr(_mod(parent), child, m)
    spec.loader.exec_module(m)
    return m


def _identity_decorator(*a, **k):
    def deco(x):
        return x
    return deco


class _FairseqDropout(nn.Module):
    def __init__(self, p=0.0, module_name=None):
        super().__init__()
        self.p = p
        self.apply_during_inference = False

    def forward(self, x, inplace=False):
        return F.dropout(x, self.p, self.training) if self.training else x


def _layer_norm(normalized_shape, eps=1e-5, elementwise_affine=True, export=False):
    return nn.LayerNorm(normalized_shape, eps, elementwise_affine)


def _get_activation_fn(activation):
    if activation == "swish":
        return nn.SiLU
    if activation == "relu":
        return F.relu
    raise RuntimeError(activation)


def _make_positions(tensor, padding_idx, onnx_trace=False):
    # arithmetic of fairseq/utils.py:256-266
    mask = tensor.ne(padding_idx).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + padding_idx


def _fill_with_neg_inf(t):
    return t.float().fill_(float("-inf")).type_as(t)


def _linear(in_features, out_features, bias=True):
    m = nn.Linear(in_features, out_features, bias)
    nn.init.xavier_uniform_(m.weight)
    if bias:
        nn.init.constant_(m.bias, 0.0)
    return m


class _TransformerConfig:
    """Nested read view over a flat argparse-style namespace (fairseq TransformerConfig.from_namespace)."""

    @staticmethod
    def from_namespace(args):
        if hasattr(args, "decoder") and hasattr(args, "quant_noise"):
            return args  # already a nested config (fairseq returns cfg unchanged)
        g = lambda k, d=None: getattr(args, k, d)
        dec = SimpleNamespace(
            embed_dim=g("decoder_embed_dim"), ffn_embed_dim=g("decoder_ffn_embed_dim"),
            layers=g("decoder_layers"), attention_heads=g("decoder_attention_heads"),
            normalize_before=g("decoder_normalize_before", True), learned_pos=False,
            layerdrop=0.0, output_dim=g("decoder_embed_dim"), input_dim=g("decoder_embed_dim"),
            xformers_att_config=None)
        enc = SimpleNamespace(
            embed_dim=g("encoder_embed_dim"), ffn_embed_dim=g("encoder_ffn_embed_dim"),
            layers=g("encoder_layers"), attention_heads=g("encoder_attention_heads"),
            normalize_before=g("encoder_normalize_before", True), learned_pos=False,
            layerdrop=0.0, xformers_att_config=None)
        return SimpleNamespace(
            decoder=dec, encoder=enc, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
            relu_dropout=0.0, activation_fn=g("activation_fn", "relu"),
            quant_noise=SimpleNamespace(pq=0, pq_block_size=8, scalar=0),
            share_decoder_input_output_embed=g("share_decoder_input_output_embed", True),
            max_target_positions=g("max_target_positions", 1024),
            max_source_positions=g("max_source_positions", 6000),
            no_scale_embedding=False, adaptive_input=False, no_token_positional_embeddings=False,
            layernorm_embedding=False, cross_self_attention=False, export=False,
            no_decoder_final_norm=False, tie_adaptive_weights=False, adaptive_softmax_cutoff=None,
            checkpoint_activations=False, offload_activations=False, min_params_to_wrap=int(1e8),
            scale_fc=False, scale_attn=False, scale_heads=False, scale_resids=False,
            no_cross_attention=False, base_layers=0)


_LOADED = {}


def load():
    """Install the stubs, execute the reference files, return a namespace of reference classes."""
    if _LOADED:
        return SimpleNamespace(**_LOADED)
    if not available():
        raise FileNotFoundError(f"reference tree not found at {REF}")

    utils = _mod("fairseq.utils",
                 get_activation_fn=_get_activation_fn,
                 softmax=lambda x, dim, onnx_trace=False: F.softmax(x, dim=dim, dtype=torch.float32),
                 log_softmax=lambda x, dim, onnx_trace=False: F.log_softmax(x, dim=dim, dtype=torch.float32),
                 fill_with_neg_inf=_fill_with_neg_inf, make_positions=_make_positions,
                 item=lambda t: t.item() if hasattr(t, "item") else t,
                 eval_str_dict=lambda x, type=dict: None if x is None else x,
                 safe_getattr=lambda obj, k, default=None: getattr(obj, k, default),
                 safe_hasattr=lambda obj, k: getattr(obj, k, None) is not None)
    _mod("fairseq", utils=utils, checkpoint_utils=_mod("fairseq.checkpoint_utils"))
    inc = _load_file("fairseq.incremental_decoding_utils", "fairseq/fairseq/incremental_decoding_utils.py")

    # the reference's own base classes (pure torch): FairseqDecoder.get_normalized_probs is what the
    # generators call (fairseq/models/fairseq_decoder.py:59-90), FairseqEncoder.forward_torchscript what
    # EnsembleModel.forward_encoder calls (fairseq/models/fairseq_encoder.py)
    fm = _mod("fairseq.models", register_model=_identity_decorator, register_model_architecture=_identity_decorator)
    fdec = _load_file("fairseq.models.fairseq_decoder", "fairseq/fairseq/models/fairseq_decoder.py")
    fm.FairseqDecoder = fdec.FairseqDecoder
    fenc = _load_file("fairseq.models.fairseq_encoder", "fairseq/fairseq/models/fairseq_encoder.py")
    fm.FairseqEncoder = fenc.FairseqEncoder
    fincd = _load_file("fairseq.models.fairseq_incremental_decoder",
                       "fairseq/fairseq/models/fairseq_incremental_decoder.py")
    FairseqIncrementalDecoder = fincd.FairseqIncrementalDecoder
    fm.FairseqIncrementalDecoder = FairseqIncrementalDecoder
    if "omegaconf" not in sys.modules:          # fairseq_model.py imports the name only
        om = types.ModuleType("omegaconf")
        om.DictConfig = type("DictConfig", (), {})
        sys.modules["omegaconf"] = om
    _mod("fairseq.data", Dictionary=type("Dictionary", (), {}))
    _mod("fairseq.dataclass")
    _mod("fairseq.dataclass.utils", convert_namespace_to_omegaconf=lambda a: a,
         gen_parser_from_dataclass=lambda *a, **k: None)
    fmodel = _load_file("fairseq.models.fairseq_model", "fairseq/fairseq/models/fairseq_model.py")
    for name in ("BaseFairseqModel", "FairseqEncoderDecoderModel", "FairseqEncoderModel", "FairseqLanguageModel"):
        setattr(fm, name, getattr(fmodel, name))
    _mod("fairseq.models.transformer", TransformerConfig=_TransformerConfig, Linear=_linear)
    _mod("fairseq.models.speech_to_text")
    _mod("fairseq.models.text_to_speech")
    _mod("fairseq.models.text_to_speech.hub_interface")
    _mod("fairseq.models.text_to_speech.tacotron2")
    _mod("fairseq.models.speech_to_speech")
    _mod("fairseq.models.speech_to_speech.modules")
    _mod("fairseq.data")
    def lengths_to_padding_mask(lens):
        # arithmetic of fairseq/data/data_utils.py lengths_to_padding_mask
        bsz, max_lens = lens.size(0), torch.max(lens).item()
        mask = torch.arange(max_lens).to(lens.device).view(1, max_lens)
        return mask.expand(bsz, -1) >= lens.view(bsz, 1).expand(-1, max_lens)
    _mod("fairseq.data.data_utils", lengths_to_padding_mask=lengths_to_padding_mask)
    _mod("fairseq.distributed", fsdp_wrap=lambda m, **k: m)
    _mod("fairseq.modules.fairseq_dropout", FairseqDropout=_FairseqDropout)
    _mod("fairseq.modules.quant_noise", quant_noise=lambda m, p=0, block_size=8: m)
    _mod("fairseq.modules.checkpoint_activations", checkpoint_wrapper=lambda m, **k: m)
    fmods = _mod("fairseq.modules", LayerNorm=_layer_norm, FairseqDropout=_FairseqDropout)

    rot = _load_file("fairseq.modules.rotary_positional_embedding",
                     "fairseq/fairseq/modules/rotary_positional_embedding.py")
    posenc = _load_file("fairseq.modules.positional_encoding", "fairseq/fairseq/modules/positional_encoding.py")
    sinpos = _load_file("fairseq.modules.sinusoidal_positional_embedding",
                        "fairseq/fairseq/modules/sinusoidal_positional_embedding.py")
    fmods.RelPositionalEncoding = posenc.RelPositionalEncoding
    fmods.SinusoidalPositionalEmbedding = sinpos.SinusoidalPositionalEmbedding

    def PositionalEmbedding(num_embeddings, embedding_dim, padding_idx, learned=False):
        # fairseq/modules/positional_embedding.py: sinusoidal branch
        assert not learned
        return sinpos.SinusoidalPositionalEmbedding(
            embedding_dim, padding_idx, init_size=num_embeddings + padding_idx + 1)
    fmods.PositionalEmbedding = PositionalEmbedding

    # S2TTransformerEncoder: ChunkS2TConformerEncoder borrows its reorder_encoder_out (s2t_conformer.py:217)
    fmods.TransformerEncoderLayer = type("TransformerEncoderLayer", (nn.Module,), {})
    sys.modules["fairseq.models.transformer"].Embedding = nn.Embedding
    _mod("fairseq.models.speech_to_text.hub_interface")
    _mod("fairseq.models.speech_to_text.modules")
    _load_file("fairseq.models.speech_to_text.modules.convolution",
               "fairseq/fairseq/models/speech_to_text/modules/convolution.py")
    _load_file("fairseq.models.speech_to_text.s2t_transformer",
               "fairseq/fairseq/models/speech_to_text/s2t_transformer.py")
    # research packages (user-dir style imports: uni_unity / chunk_unity / ctc_unity)
    for pkg in ("uni_unity", "uni_unity.modules", "chunk_unity", "chunk_unity.modules",
                "ctc_unity", "ctc_unity.modules"):
        _mod(pkg)
    _mod("uni_unity.modules.multihead_attention")  # only imported by name in conformer_layer.py
    espnet = _load_file("uni_unity.modules.espnet_multihead_attention",
                        "researches/uni_unity/modules/espnet_multihead_attention.py")
    ccc = _load_file("chunk_unity.modules.chunk_causal_conv1d",
                     "researches/chunk_unity/modules/chunk_causal_conv1d.py")
    convm = _load_file("chunk_unity.modules.convolution", "researches/chunk_unity/modules/convolution.py")
    conf = _load_file("chunk_unity.modules.conformer_layer", "researches/chunk_unity/modules/conformer_layer.py")
    _mod("chunk_unity.models")
    s2tc = _load_file("chunk_unity.models.s2t_conformer", "researches/chunk_unity/models/s2t_conformer.py")
    mha = _load_file("ctc_unity.modules.multihead_attention", "researches/ctc_unity/modules/multihead_attention.py")
    tl = _load_file("ctc_unity.modules.transformer_layer", "researches/ctc_unity/modules/transformer_layer.py")
    getattr(sys.modules["ctc_unity.modules"], "transformer_layer")
    td = _load_file("ctc_unity.modules.transformer_decoder", "researches/ctc_unity/modules/transformer_decoder.py")
    te = _load_file("ctc_unity.modules.transformer_encoder", "researches/ctc_unity/modules/transformer_encoder.py")
    ctcdec = _load_file("fairseq.models.speech_to_speech.modules.ctc_decoder",
                        "fairseq/fairseq/models/speech_to_speech/modules/ctc_decoder.py")
    stk = _load_file("fairseq.models.speech_to_speech.modules.stacked_embedding",
                     "fairseq/fairseq/models/speech_to_speech/modules/stacked_embedding.py")
    ud = _load_file("ctc_unity.modules.ctc_transformer_unit_decoder",
                    "researches/ctc_unity/modules/ctc_transformer_unit_decoder.py")
    hifi = _load_file("fairseq.models.text_to_speech.hifigan", "fairseq/fairseq/models/text_to_speech/hifigan.py")
    fs2 = _load_file("fairseq.models.text_to_speech.fastspeech2",
                     "fairseq/fairseq/models/text_to_speech/fastspeech2.py")
    _mod("agent"); _mod("agent.tts")
    chg = _load_file("agent.tts.codehifigan", "agent/tts/codehifigan.py")

    _LOADED.update(
        RelPositionalEncoding=posenc.RelPositionalEncoding,
        ESPNETMultiHeadedAttention=espnet.ESPNETMultiHeadedAttention,
        RelPositionMultiHeadedAttention=espnet.RelPositionMultiHeadedAttention,
        ChunkCausalConv1d=ccc.ChunkCausalConv1d,
        Conv1dSubsampler=convm.Conv1dSubsampler,
        ChunkConformerEncoderLayer=conf.ChunkConformerEncoderLayer,
        ChunkS2TConformerEncoder=s2tc.ChunkS2TConformerEncoder,
        MultiheadAttention=mha.MultiheadAttention,
        TransformerEncoderLayer=tl.TransformerEncoderLayer,
        TransformerDecoderLayer=tl.TransformerDecoderLayer,
        TransformerDecoder=td.TransformerDecoder,
        UniTransformerEncoderNoEmb=te.UniTransformerEncoderNoEmb,
        CTCDecoder=ctcdec.CTCDecoder,
        StackedEmbedding=stk.StackedEmbedding,
        CTCTransformerUnitDecoder=ud.CTCTransformerUnitDecoder,
        Generator=hifi.Generator,
        CodeGenerator=chg.CodeGenerator,
        VariancePredictor=fs2.VariancePredictor,
        SinusoidalPositionalEmbedding=sinpos.SinusoidalPositionalEmbedding,
    )
    return SimpleNamespace(**_LOADED)
