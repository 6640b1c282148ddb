"""fairseq-registry surface of the S2ST path, backed by the HIP library.

Keeps the names and call surface the reference agent touches (SURVEY.md §8b "Module call surface"):
model class ``StreamSpeechModel`` registered as model/arch ``streamspeech``
(researches/ctc_unity/models/streamspeech_model.py:57,418), task ``speech_to_speech_ctc``
(researches/ctc_unity/tasks/speech_to_speech_ctc.py:11) and vocoder
``CodeHiFiGANVocoderWithDur`` (agent/tts/vocoder.py:30).  Sub-modules are thin objects whose
``__call__`` forwards to one C-ABI stage; the attributes the agent mutates
(``encoder.chunk_size``, ``conv.chunk_size``, agent :395-413) are plain Python attributes read at
call time.  Utterances are processed one at a time (B = 1 semantics).
"""
import json
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from .config import ModelConfig, VocoderConfig
from .engine import HipModel, HipVocoder

MODEL_REGISTRY: Dict[str, type] = {}
ARCH_MODEL_REGISTRY: Dict[str, type] = {}
TASK_REGISTRY: Dict[str, type] = {}


def register_model(name):
    """fairseq.models.register_model (fairseq/models/__init__.py:109) for this package."""
    def deco(cls):
        MODEL_REGISTRY[name] = cls
        return cls
    return deco


def register_model_architecture(model_name, arch_name):
    def deco(fn):
        ARCH_MODEL_REGISTRY[arch_name] = MODEL_REGISTRY[model_name]
        return fn
    return deco


def register_task(name):
    def deco(cls):
        TASK_REGISTRY[name] = cls
        return cls
    return deco


class _ChunkHolder:
    """Stands in for a ChunkCausalConv1d whose ``chunk_size`` the agent overwrites."""

    def __init__(self, chunk_size=999999):
        self.chunk_size = chunk_size


class _ConvModuleView:
    def __init__(self):
        self.depthwise_conv = _ChunkHolder()


class _LayerView:
    def __init__(self):
        self.conv_module = _ConvModuleView()


class _SubsampleView:
    def __init__(self):
        self.conv_layers = [_ChunkHolder(), _ChunkHolder()]


class HipChunkConformerEncoder:
    """``model.encoder``: ChunkS2SConformerEncoder surface (chunk_unity/models/s2s_conformer.py:37-62)."""

    def __init__(self, hip: HipModel):
        self.hip = hip
        self.chunk_size = 999999
        self.chunk = True
        # streaming: reuse the rows that are already final instead of re-encoding all audio at every
        # policy() call; the agent switches it on and calls reset_stream() per utterance
        self.incremental = False
        self.subsample = _SubsampleView()
        self.conformer_layers = [_LayerView() for _ in range(hip.cfg.enc_layers)]

    def _conv_chunk(self) -> int:
        cs = {c.chunk_size for c in self.subsample.conv_layers}
        cs |= {l.conv_module.depthwise_conv.chunk_size for l in self.conformer_layers}
        if len(cs) != 1:
            raise ValueError("all ChunkCausalConv1d chunk sizes must agree (the agent sets them together)")
        return cs.pop()

    def __call__(self, src_tokens: torch.Tensor, src_lengths: torch.Tensor = None, **kw):
        assert src_tokens.dim() == 3 and src_tokens.size(0) == 1, "one utterance per call (B = 1)"
        fb = src_tokens[0].to(self.hip.device, torch.float32).contiguous()
        if self.incremental and hasattr(self.hip, "encoder_stream_forward"):
            out = self.hip.encoder_stream_forward(fb, self.chunk_size, self._conv_chunk())
        else:
            out = self.hip.encoder_forward(fb, self.chunk_size, self._conv_chunk())
        return {"encoder_out": [out.unsqueeze(1)], "encoder_padding_mask": [], "encoder_embedding": [],
                "encoder_states": [], "src_tokens": [], "src_lengths": []}

    forward = __call__

    def reset_stream(self):
        if hasattr(self.hip, "encoder_stream_reset"):
            self.hip.encoder_stream_reset()


class HipCTCDecoder:
    """``{source,ctc_target}_unigram_decoder``: Linear head (speech_to_speech/modules/ctc_decoder.py:11-18)."""

    def __init__(self, hip: HipModel, head: int):
        self.hip, self.head = hip, head

    def __call__(self, enc_out: torch.Tensor, **kw):
        x = enc_out[:, 0] if enc_out.dim() == 3 else enc_out
        _, _, _, logits = self.hip.ctc_greedy(self.head, x.contiguous(), want_logits=True)
        return {"encoder_out": logits.unsqueeze(1)}


class HipMTDecoder:
    """``target_unigram_decoder``: first-pass text decoder (features_only surface, agent :638-642)."""

    padding_idx = 1

    def __init__(self, hip: HipModel):
        self.hip = hip

    def __call__(self, prev_output_tokens: torch.Tensor, encoder_out=None, features_only=True, **kw):
        assert features_only, "logits are produced inside the greedy search (ss_mt_append)"
        assert prev_output_tokens.size(0) == 1
        enc = encoder_out["encoder_out"][0]
        self.hip.mt_begin(enc[:, 0].contiguous() if enc.dim() == 3 else enc)
        feats, _ = self.hip.mt_append(prev_output_tokens[0].tolist(), 0, False, False, want_next=False)
        return feats.unsqueeze(0), {"attn": [None], "inner_states": []}


class HipT2UAndUnitDecoder:
    """``synthesizer_encoder`` + ``decoder`` fused into one C-ABI stage (agent :661-689)."""

    def __init__(self, hip: HipModel, uni_encoder: bool):
        self.hip, self.uni = hip, uni_encoder

    def units(self, mt_feats: torch.Tensor, mask_eos=False):
        return self.hip.t2u_units(mt_feats, t2u_causal=self.uni, mask_eos=mask_eos)


@register_model("streamspeech")
class StreamSpeechModel:
    """Registry name ``streamspeech``; built from a fairseq checkpoint's ``state['model']``."""

    def __init__(self, state_dict, cfg: ModelConfig = None, device="cuda:0", cmvn=None, uni_encoder=False):
        cfg = cfg or ModelConfig()
        mean, std = (cmvn["mean"], cmvn["std"]) if cmvn is not None else (None, None)
        self._wire(HipModel(state_dict, cfg, device=device, cmvn_mean=mean, cmvn_std=std), uni_encoder)

    @classmethod
    def from_engine(cls, engine, uni_encoder=False):
        """Wrap an already-built engine (anything with the HipModel method set)."""
        self = cls.__new__(cls)
        self._wire(engine, uni_encoder)
        return self

    def _wire(self, engine, uni_encoder):
        self.hip = engine
        self.cfg = cfg = engine.cfg
        self.uni_encoder = uni_encoder
        self.mt_task_name = "target_unigram"
        self.encoder = HipChunkConformerEncoder(self.hip)
        self.source_unigram_decoder = HipCTCDecoder(self.hip, 0)
        self.ctc_target_unigram_decoder = HipCTCDecoder(self.hip, 1)
        self.target_unigram_decoder = HipMTDecoder(self.hip)
        self.t2u = HipT2UAndUnitDecoder(self.hip, uni_encoder)
        self.t2u_augmented_cross_attn = False

    @classmethod
    def build_model(cls, args, task=None):
        """fairseq ``build_model(args, task)`` (streamspeech_model.py:182-258): args carries
        ``model_path`` (a fairseq .pt or ``synthetic:<seed>``)."""
        sd, uni = load_model_state(getattr(args, "model_path"))
        return cls(sd, device=getattr(args, "device_str", "cuda:0"), cmvn=getattr(args, "global_cmvn", None),
                   uni_encoder=uni or getattr(args, "uni_encoder", False))

    def eval(self):
        return self

    def cuda(self):
        return self

    def share_memory(self):
        return self

    def max_decoder_positions(self):
        return self.cfg.max_target_positions

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        return self.hip.normalized_probs(net_output[0], log_probs)       # ss_log_softmax: no model math in torch


@register_model_architecture("streamspeech", "streamspeech")
def streamspeech_architecture(args):
    for k, v in ModelConfig().__dict__.items():
        if not hasattr(args, k):
            setattr(args, k, v)


@register_task("speech_to_speech_ctc")
class SpeechToSpeechCTCTask:
    """Task name kept for ``--task speech_to_speech_ctc``; it only owns the dictionaries here."""

    def __init__(self, dicts: Dict[str, "Dictionary"]):
        self.dicts = dicts
        self.target_dictionary = dicts["tgt"]
        self.multitask_tasks = {k: v for k, v in dicts.items() if k != "tgt"}


class Dictionary:
    """Minimal fairseq Dictionary (fairseq/data/dictionary.py): '<s> <pad> </s> <unk>' + symbols."""

    def __init__(self, symbols: List[str], extra: Optional[List[str]] = None):
        self.symbols = ["<s>", "<pad>", "</s>", "<unk>"] + list(symbols) + list(extra or [])
        self.bos_index, self.pad_index, self.eos_index, self.unk_index = 0, 1, 2, 3
        self.blank_index = self.symbols.index("<blank>") if "<blank>" in self.symbols else None

    @classmethod
    def load(cls, path, extra=None):
        with open(path, encoding="utf-8") as f:
            return cls([ln.rsplit(" ", 1)[0] for ln in f.read().splitlines() if ln.strip()], extra)

    @classmethod
    def units(cls, n=1000):
        # SpeechToSpeechCTCTask adds "<blank>" after the unit symbols (tasks/speech_to_speech_ctc.py:14-19)
        return cls([str(i) for i in range(n)], ["<blank>"])

    @classmethod
    def placeholder(cls, n):
        return cls([f"▁w{i}" for i in range(n - 4)])

    def __len__(self):
        return len(self.symbols)

    def __getitem__(self, i):
        return self.symbols[int(i)]

    def pad(self):
        return self.pad_index

    def eos(self):
        return self.eos_index

    def unk(self):
        return self.unk_index

    def bos(self):
        return self.bos_index


def load_model_state(path: str):
    """-> (state dict, uni_encoder flag).  ``synthetic:<seed>`` gives the seeded random checkpoint."""
    if path.startswith("synthetic"):
        from . import synth
        seed = int(path.split(":")[1]) if ":" in path else 0
        return synth.make_model_state_dict(seed), False
    if not os.path.exists(path):
        raise IOError("Model file not found: {}".format(path))   # agent :357-358
    state = torch.load(path, map_location="cpu", weights_only=False)
    sd = state["model"] if "model" in state else state
    uni = False
    try:
        uni = bool(getattr(state["cfg"]["model"], "uni_encoder", False))
    except Exception:  # noqa: BLE001
        pass
    return {k: v for k, v in sd.items()}, uni


@register_model("CodeHiFiGANVocoderWithDur")
class CodeHiFiGANVocoderWithDur:
    """agent/tts/vocoder.py:30-60: ``vocoder({"code": LongTensor[1,K]}, dur_prediction) -> (wav[S], dur[1,K])``."""

    def __init__(self, checkpoint_path: str, model_cfg: Dict = None, fp16: bool = False, device="cuda:0"):
        assert not fp16, "the HIP path is FP32 (parity with the reference CPU path)"
        vcfg = vocoder_config_from_json(model_cfg) if model_cfg else VocoderConfig()
        if checkpoint_path.startswith("synthetic"):
            from . import synth
            seed = int(checkpoint_path.split(":")[1]) if ":" in checkpoint_path else 0
            gsd = synth.make_vocoder_state_dict(seed, vcfg)
        else:
            if not os.path.exists(checkpoint_path):
                raise IOError("Vocoder file not found: {}".format(checkpoint_path))
            gsd = torch.load(checkpoint_path, map_location="cpu", weights_only=False)["generator"]
        self.hip = HipVocoder(gsd, vcfg, device=device)

    def cuda(self):
        return self

    def __call__(self, x: Dict[str, torch.Tensor], dur_prediction=False):
        assert "code" in x
        code = x["code"]
        code = code[code >= 0].view(-1)                      # remove invalid code (vocoder.py:52-54)
        wav, dur = self.hip.forward(code.to(torch.int32), dur_prediction)
        return wav, dur.view(1, -1).long()

    forward = __call__


def vocoder_config_from_json(d: Dict) -> VocoderConfig:
    dp = d.get("dur_predictor_params") or {}
    return VocoderConfig(
        num_embeddings=d.get("num_embeddings", 1000), embedding_dim=d.get("embedding_dim", 128),
        model_in_dim=d.get("model_in_dim", 128), upsample_rates=tuple(d["upsample_rates"]),
        upsample_kernel_sizes=tuple(d["upsample_kernel_sizes"]),
        upsample_initial_channel=d["upsample_initial_channel"],
        resblock_kernel_sizes=tuple(d["resblock_kernel_sizes"]),
        resblock_dilation_sizes=tuple(tuple(x) for x in d["resblock_dilation_sizes"]),
        dur_hidden=dp.get("var_pred_hidden_dim", 128), dur_kernel=dp.get("var_pred_kernel_size", 3),
        code_hop_size=d.get("code_hop_size", 320))


def register_with_fairseq() -> bool:
    """Put the HIP-backed classes into a REAL fairseq's registries (fairseq/models/__init__.py:109,161 register_model /
    register_model_architecture; fairseq/tasks/__init__.py register_task) when fairseq is importable; the package-local
    registries above are filled either way.  fairseq insists on its own base classes, so thin subclasses are made here:
    the model wrapper is a BaseFairseqModel whose build_model returns the StreamSpeechModel facade, the task a
    LegacyFairseqTask that owns the dictionaries.  Returns False when fairseq is absent or already holds the names."""
    try:
        from fairseq.models import BaseFairseqModel, register_model as f_register_model
        from fairseq.models import register_model_architecture as f_register_arch
        from fairseq.tasks import LegacyFairseqTask, register_task as f_register_task
    except Exception:  # noqa: BLE001  (fairseq absent or not importable in this environment)
        return False
    try:
        @f_register_model("streamspeech")
        class FairseqStreamSpeechModel(BaseFairseqModel):
            """fairseq-facing shell: ``build_model(args, task)`` hands back the HIP-backed facade, which carries the
            attribute surface the agent and the generators touch."""

            @staticmethod
            def add_args(parser):
                parser.add_argument("--uni-encoder", action="store_true", default=False)

            @classmethod
            def build_model(cls, args, task):
                return StreamSpeechModel.build_model(args, task)

        @f_register_arch("streamspeech", "streamspeech")
        def _arch(args):
            streamspeech_architecture(args)

        @f_register_task("speech_to_speech_ctc")
        class FairseqSpeechToSpeechCTCTask(LegacyFairseqTask):
            def __init__(self, args, dicts=None):
                super().__init__(args)
                self.impl = SpeechToSpeechCTCTask(dicts or {"tgt": Dictionary.units(1000)})

            @classmethod
            def setup_task(cls, args, **kw):
                return cls(args)

            @property
            def target_dictionary(self):
                return self.impl.target_dictionary

            @property
            def multitask_tasks(self):
                return self.impl.multitask_tasks

        @f_register_model("CodeHiFiGANVocoderWithDur")
        class FairseqCodeHiFiGANVocoderWithDur(BaseFairseqModel):
            def __new__(cls, checkpoint_path, model_cfg=None, fp16=False, **kw):
                return CodeHiFiGANVocoderWithDur(checkpoint_path, model_cfg, fp16, **kw)
    except ValueError:      # "Cannot register duplicate model/task": the reference's user dir was imported first
        return False
    return True
