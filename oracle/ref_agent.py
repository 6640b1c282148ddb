"""Runs the reference's OWN generator classes and SimulEval agents on CPU, for pinning the host side.

TEST INFRASTRUCTURE (needs /root/reference; used by oracle/make_golden_agent.py to write
tests/golden/agent_*.npz and by CPU tests that are skipped when the reference is absent).

What is executed from /root/reference, unmodified and where it lies:
  agent/ctc_decoder.py, agent/ctc_generator.py, agent/sequence_generator.py,
  fairseq/fairseq/sequence_generator.py, fairseq/fairseq/search.py, fairseq/fairseq/ngram_repeat_block.py,
  fairseq/fairseq/token_generation_constraints.py, fairseq/fairseq/models/fairseq_{model,encoder,decoder,
  incremental_decoder}.py, fairseq/fairseq/data/audio/audio_utils.py, fairseq/examples/speech_to_text/data_utils.py,
  agent/tts/vocoder.py, agent/tts/codehifigan.py, SimulEval/simuleval/agents/{agent,states,actions}.py,
  SimulEval/simuleval/data/segments.py, and the three agent files
  agent/speech_to_speech.streamspeech.agent.py, agent/speech_to_text.{s2tt,asr}.streamspeech.agent.py
  (their __init__, load_model_vocab, reset and policy run as written).

What is substituted (and why it does not touch what is being pinned):
  * fairseq's control plane -- checkpoint_utils.load_checkpoint_to_cpu / load_model_ensemble, tasks.setup_task,
    utils.import_user_module -- hands the agent a model assembled from the reference's own module classes
    (oracle/ref_build.py) filled with the synthetic state dict, and synthetic dictionaries.  The model wrapper is
    a FairseqEncoderDecoderModel (the reference's class) with the attributes StreamSpeechModel.build_model sets
    (researches/ctc_unity/models/streamspeech_model.py:182-258); StreamSpeechModel itself needs the whole fairseq
    model zoo to import.
  * torchaudio (absent): ``torchaudio.compliance.kaldi.fbank`` -> oracle/kaldi_fbank.py (the third-party arithmetic
    SURVEY.md §8c lists as unpinned), ``torchaudio.sox_effects.apply_effects_tensor(["rate", ...])`` ->
    oracle/resample.py (scipy-pinned polyphase; sox itself is outside the parity contract).
  * omegaconf / soundfile / yt_dlp-dependent SimulEval modules: name-only stubs.
"""
import argparse
import json
import os
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np
import torch

from . import kaldi_fbank as K
from . import ref_build, ref_loader
from .ref_loader import _load_file, _mod

_STATE = {}


class SynthDictionary:
    """fairseq Dictionary surface the agents touch (fairseq/data/dictionary.py): bos=0 pad=1 eos=2 unk=3, then
    n-4 symbols.  Text dictionaries get SentencePiece-looking symbols, two out of three word-initial
    ("▁..."), so the whole-word path (agent :540-574) has word boundaries to find; the unit dictionary is
    '0'..'999' + '<blank>' as speech_to_speech_ctc builds it (tasks/speech_to_speech_ctc.py)."""

    def __init__(self, n, kind="text", tag="w"):
        self.symbols = ["<s>", "<pad>", "</s>", "<unk>"]
        if kind == "unit":
            self.symbols += [str(i) for i in range(n - 5)] + ["<blank>"]
            self.blank_index = n - 1
        else:
            self.symbols += [("" if i % 3 == 0 else "▁") + f"{tag}{i}" for i in range(n - 4)]
        assert len(self.symbols) == n
        self.bos_index, self.pad_index, self.eos_index, self.unk_index = 0, 1, 2, 3

    def __len__(self):
        return len(self.symbols)

    def __getitem__(self, i):
        i = int(i)
        return self.symbols[i] if i < len(self.symbols) else "<unk>"

    def bos(self):
        return 0

    def pad(self):
        return 1

    def eos(self):
        return 2

    def unk(self):
        return 3


def _install_stubs():
    if _STATE.get("stubs"):
        return
    R = ref_loader.load()
    fm = sys.modules["fairseq.models"]
    # ---- fairseq generic search machinery (real files) ----
    _load_file("fairseq.token_generation_constraints", "fairseq/fairseq/token_generation_constraints.py")
    _load_file("fairseq.search", "fairseq/fairseq/search.py")
    _load_file("fairseq.ngram_repeat_block", "fairseq/fairseq/ngram_repeat_block.py")
    _load_file("fairseq.sequence_generator", "fairseq/fairseq/sequence_generator.py")
    # ---- audio front-end: reference files over stubbed torchaudio / soundfile ----
    from . import resample as RS

    def kaldi_fbank(waveform, num_mel_bins=80, sample_frequency=16000.0, **kw):
        assert num_mel_bins == 80 and int(sample_frequency) == 16000 and not kw
        x = waveform.detach().cpu().numpy().astype(np.float32)
        assert x.ndim == 2 and x.shape[0] == 1
        return torch.from_numpy(K.fbank(x[0]))

    def apply_effects_tensor(waveform, sample_rate, effects):
        out, sr = waveform, sample_rate
        for eff in effects:
            if eff[0] == "rate":
                to = int(eff[1])
                x = out.detach().cpu().numpy().astype(np.float32)
                out = torch.from_numpy(np.stack([RS.resample_poly_ref(c, to, sr) for c in x]))
                sr = to
            elif eff[0] == "channels":
                out = out.mean(0, keepdim=True)
            else:
                raise NotImplementedError(eff)
        return out, sr

    ta = types.ModuleType("torchaudio")
    ta.__path__ = []
    tac = types.ModuleType("torchaudio.compliance")
    tac.__path__ = []
    tak = types.ModuleType("torchaudio.compliance.kaldi")
    tak.fbank = kaldi_fbank
    tas = types.ModuleType("torchaudio.sox_effects")
    tas.apply_effects_tensor = apply_effects_tensor
    ta.compliance, tac.kaldi, ta.sox_effects = tac, tak, tas
    for m in (ta, tac, tak, tas):
        sys.modules.setdefault(m.__name__, m)
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
    _mod("fairseq.data.audio")
    _mod("fairseq.data.audio.waveform_transforms")
    _mod("fairseq.data.audio.feature_transforms",
         CompositeAudioFeatureTransform=SimpleNamespace(from_config_dict=lambda cfg: None))
    _load_file("fairseq.data.audio.audio_utils", "fairseq/fairseq/data/audio/audio_utils.py")
    _mod("examples")
    _mod("examples.speech_to_text")
    _load_file("examples.speech_to_text.data_utils", "fairseq/examples/speech_to_text/data_utils.py")
    # ---- names the agent files import but never call on this path ----
    for name in ("fairseq.data.audio.data_cfg", "fairseq.data.audio.speech_to_speech_dataset",
                 "fairseq.data.audio.speech_to_text_dataset", "fairseq.tasks", "fairseq.tasks.speech_to_text",
                 "fairseq.tasks.text_to_speech", "fairseq.file_io", "fairseq.options",
                 "fairseq.models.text_to_speech.hub_interface", "examples.speech_to_speech",
                 "examples.speech_to_speech.asr_bleu", "examples.speech_to_speech.asr_bleu.utils"):
        _mod(name)
    sys.modules["fairseq.tasks"].register_task = ref_loader._identity_decorator
    # ---- SimulEval: the agent base classes are pure Python ----
    _mod("simuleval")
    _mod("simuleval.data")
    _load_file("simuleval.data.segments", "SimulEval/simuleval/data/segments.py")
    _mod("simuleval.agents")
    acts = _load_file("simuleval.agents.actions", "SimulEval/simuleval/agents/actions.py")
    states = _load_file("simuleval.agents.states", "SimulEval/simuleval/agents/states.py")
    agent = _load_file("simuleval.agents.agent", "SimulEval/simuleval/agents/agent.py")
    sa = sys.modules["simuleval.agents"]
    for name in ("GenericAgent", "SpeechToTextAgent", "SpeechToSpeechAgent", "TextToSpeechAgent", "TextToTextAgent"):
        setattr(sa, name, getattr(agent, name))
    sa.AgentStates = states.AgentStates
    sa.Action, sa.ReadAction, sa.WriteAction = acts.Action, acts.ReadAction, acts.WriteAction

    def entrypoint(klass):            # SimulEval/simuleval/utils/__init__.py:10-12 (registers the class; no arithmetic)
        return klass
    _mod("simuleval.utils", entrypoint=entrypoint)
    # ---- the reference's generators and vocoder wrapper ----
    _mod("agent")
    _mod("agent.tts")
    _STATE["ctc_decoder"] = _load_file("agent.ctc_decoder", "agent/ctc_decoder.py")
    _STATE["ctc_generator"] = _load_file("agent.ctc_generator", "agent/ctc_generator.py")
    _STATE["sequence_generator"] = _load_file("agent.sequence_generator", "agent/sequence_generator.py")
    _STATE["vocoder"] = _load_file("agent.tts.vocoder", "agent/tts/vocoder.py")
    _STATE["R"] = R
    _STATE["stubs"] = True


def generators():
    """-> namespace(CTCDecoder, CTCSequenceGenerator, SequenceGenerator, BeamSearch): the reference's classes."""
    _install_stubs()
    return SimpleNamespace(
        CTCDecoder=_STATE["ctc_decoder"].CTCDecoder,
        CTCSequenceGenerator=_STATE["ctc_generator"].CTCSequenceGenerator,
        SequenceGenerator=_STATE["sequence_generator"].SequenceGenerator,
        BeamSearch=sys.modules["fairseq.search"].BeamSearch,
        CodeHiFiGANVocoderWithDur=_STATE["vocoder"].CodeHiFiGANVocoderWithDur)


def build_model(sd, cfg, uni_t2u=False, dicts=None):
    """The reference's modules wired as StreamSpeechModel.build_model wires them
    (researches/ctc_unity/models/streamspeech_model.py:182-258) inside the reference's
    FairseqEncoderDecoderModel."""
    _install_stubs()
    fm = sys.modules["fairseq.models"]
    dicts = dicts or make_dicts(cfg)
    enc = ref_build.build_encoder(sd, cfg, 999999, 999999)
    dec = ref_build.build_unit_decoder(sd, cfg)
    dec.dictionary = dicts["tgt"]
    model = fm.FairseqEncoderDecoderModel(enc, dec)
    model.t2u_augmented_cross_attn = False
    model.mt_task_name = "target_unigram"
    model.target_unigram_decoder = ref_build.build_mt_decoder(sd, cfg)
    model.source_unigram_decoder = ref_build.build_ctc_head(sd, cfg, "source_unigram")
    model.ctc_target_unigram_decoder = ref_build.build_ctc_head(sd, cfg, "ctc_target_unigram")
    model.synthesizer_encoder = ref_build.build_t2u_encoder(sd, cfg, uni=uni_t2u)
    model.eval()
    return model


def make_dicts(cfg):
    return {"tgt": SynthDictionary(cfg.unit_vocab, "unit"),
            "target_unigram": SynthDictionary(cfg.tgt_vocab, "text", "t"),
            "source_unigram": SynthDictionary(cfg.src_vocab, "text", "s"),
            "ctc_target_unigram": SynthDictionary(cfg.tgt_vocab, "text", "t")}


class _Cfg(dict):
    """state["cfg"] as load_model_vocab reads it: attribute and item access (agent :360-394)."""
    __getattr__ = dict.__getitem__


def _agent_env(sd, vsd, cfg, vcfg, workdir, cmvn_npz=None, uni_t2u=False):
    """Files + control-plane stubs the reference agents' __init__ / load_model_vocab read."""
    _install_stubs()
    dicts = make_dicts(cfg)
    model = build_model(sd, cfg, uni_t2u, dicts)
    os.makedirs(workdir, exist_ok=True)
    model_path = os.path.join(workdir, "synthetic_model.pt")
    open(model_path, "wb").close()                       # only os.path.exists() looks at it
    voc_path = os.path.join(workdir, "synthetic_vocoder.pt")
    torch.save({"generator": {k: torch.from_numpy(np.ascontiguousarray(v)).clone() for k, v in vsd.items()}}, voc_path)
    voc_cfg = os.path.join(workdir, "vocoder_config.json")
    with open(voc_cfg, "w") as f:
        json.dump(vcfg.as_dict(), f)
    with open(os.path.join(workdir, "config_gcmvn.yaml"), "w") as f:
        if cmvn_npz:
            f.write(f"global_cmvn:\n  stats_npz_path: {cmvn_npz}\n")
        else:
            f.write("input_feat_per_channel: 80\n")
    task = SimpleNamespace(target_dictionary=dicts["tgt"],
                           multitask_tasks={k: SimpleNamespace(tgt_dict=dicts[k], target_dictionary=dicts[k])
                                            for k in ("target_unigram", "source_unigram", "ctc_target_unigram")})
    state = _Cfg(cfg=_Cfg(common=_Cfg(user_dir=None), task=SimpleNamespace(),
                          common_eval=SimpleNamespace(model_overrides="{}"),
                          checkpoint=SimpleNamespace(checkpoint_suffix="", checkpoint_shard_count=1)))
    cu = sys.modules["fairseq.checkpoint_utils"]
    cu.load_checkpoint_to_cpu = lambda filename, *a, **k: state
    cu.load_model_ensemble = lambda filenames, **k: ([model], state["cfg"])
    cu.load_model_ensemble_and_task = lambda *a, **k: ([model], state["cfg"], task)
    sys.modules["fairseq.tasks"].setup_task = lambda task_args, **k: task
    fu = sys.modules["fairseq.utils"]
    fu.import_user_module = lambda *a, **k: None
    fu.split_paths = lambda p, separator=os.pathsep: p.split(separator)
    sys.modules["fairseq"].tasks = sys.modules["fairseq.tasks"]
    return SimpleNamespace(model=model, dicts=dicts, model_path=model_path, vocoder=voc_path, vocoder_cfg=voc_cfg,
                           data_bin=workdir, config_yaml="config_gcmvn.yaml")


_AGENT_FILES = {"s2st": "agent/speech_to_speech.streamspeech.agent.py",
                "s2tt": "agent/speech_to_text.s2tt.streamspeech.agent.py",
                "asr": "agent/speech_to_text.asr.streamspeech.agent.py"}


def agent_module(kind="s2st"):
    _install_stubs()
    key = "agent_mod_" + kind
    if key not in _STATE:
        _STATE[key] = _load_file("ref_streamspeech_agent_" + kind, _AGENT_FILES[kind])
    return _STATE[key]


def make_agent(sd, vsd, cfg, vcfg, segment_ms=320, sample_rate=16000, kind="s2st", cmvn_npz=None, workdir=None,
               uni_t2u=False, **over):
    """Instantiate the reference agent class (its own __init__) on CPU.  ``sample_rate`` is what SimulEval's
    dataloader would deliver; at 16000 the feature extractor's resampling default (the module constant
    ORG_SAMPLE_RATE = 48000 bound at :74) is re-bound to 16000 so that convert_waveform is the identity."""
    mod = agent_module(kind)
    workdir = workdir or tempfile.mkdtemp(prefix="ss_ref_agent_")
    env = _agent_env(sd, vsd, cfg, vcfg, workdir, cmvn_npz, uni_t2u)
    cls = {"s2st": "StreamSpeechS2STAgent", "s2tt": "StreamSpeechS2TTAgent", "asr": "StreamSpeechASRAgent"}[kind]
    klass = getattr(mod, cls)
    p = argparse.ArgumentParser()
    klass.add_args(p)
    argv = ["--model-path", env.model_path, "--data-bin", env.data_bin, "--config-yaml", env.config_yaml,
            "--sample-rate", str(sample_rate)]
    if kind == "s2st":
        argv += ["--vocoder", env.vocoder, "--vocoder-cfg", env.vocoder_cfg, "--dur-prediction"]
    args = p.parse_args(argv)
    args.source_segment_size = segment_ms                 # SimulEval dataloader option (data/dataloader/dataloader.py:93)
    args.device = "cpu"                                   # SimulEval option (options.py:157-159)
    for k, v in over.items():
        setattr(args, k, v)
    fe = mod.OnlineFeatureExtractor.__call__
    fe.__defaults__ = (int(sample_rate),)
    agent = klass(args)
    agent._ref_env = env
    return agent


def stream(agent, pcm, segment_ms=320, sr=16000):
    """SentenceLevelEvaluator's loop (SimulEval/simuleval/evaluator/evaluator.py:216-235): push one source segment,
    pop, until the source is finished.  Returns the per-call records."""
    seg_mod = sys.modules["simuleval.data.segments"]
    step = sr * segment_ms // 1000
    recs, pos = [], 0
    while True:
        chunk = pcm[pos:pos + step]
        pos += step
        finished = pos >= len(pcm)
        out = agent.pushpop(seg_mod.SpeechSegment(content=chunk.tolist(), sample_rate=sr, finished=finished))
        recs.append(out)
        if finished:
            return recs
