"""Host control flow of the SimulEval agent (policy gating, generators, registry surface) on CPU
with the oracle-backed engine: no GPU, no HIP compute."""
import argparse

import numpy as np
import pytest
import torch

from streamspeech_amd import synth
from streamspeech_amd.modules import (ARCH_MODEL_REGISTRY, MODEL_REGISTRY, TASK_REGISTRY, StreamSpeechModel)
from streamspeech_amd.simuleval_shim import ReadAction, SpeechSegment, WriteAction
from tests.oracle_engine import OracleEngine, OracleVocoder


def make_args(segment_ms=320, **over):
    from streamspeech_amd.agent import StreamSpeechS2STAgent
    p = argparse.ArgumentParser()
    StreamSpeechS2STAgent.add_args(p)
    a = p.parse_args(["--model-path", "synthetic:0", "--data-bin", "/nonexistent", "--vocoder", "synthetic:0",
                      "--dur-prediction", "--sample-rate", "16000"])
    a.source_segment_size = segment_ms
    a.device = "gpu"
    for k, v in over.items():
        setattr(a, k, v)
    return a


def stream(agent, pcm, segment_ms=320, sr=16000):
    """Mimics SentenceLevelEvaluator: send_source(chunk) -> pushpop -> collect (evaluator.py:216-235)."""
    step = sr * segment_ms // 1000
    out, actions = [], []
    pos = 0
    while True:
        chunk = pcm[pos:pos + step]
        pos += step
        finished = pos >= len(pcm)
        seg = agent.pushpop(SpeechSegment(content=chunk.tolist(), sample_rate=sr, finished=finished))
        actions.append("R" if seg.is_empty else "W")
        if not seg.is_empty:
            out.append(np.asarray(seg.content, np.float32))
        if finished:
            break
    return (np.concatenate(out) if out else np.zeros(0, np.float32)), actions


def test_registry_names():
    assert "streamspeech" in MODEL_REGISTRY and "streamspeech" in ARCH_MODEL_REGISTRY
    assert "CodeHiFiGANVocoderWithDur" in MODEL_REGISTRY and "speech_to_speech_ctc" in TASK_REGISTRY


def test_missing_model_raises_ioerror():
    from streamspeech_amd.modules import load_model_state
    with pytest.raises(IOError):
        load_model_state("/nonexistent/streamspeech.pt")


def test_agent_streaming_control_flow(synth_weights):
    from streamspeech_amd.agent import StreamSpeechS2STAgent
    cfg, vcfg, sd, vsd = synth_weights
    model = StreamSpeechModel.from_engine(OracleEngine(sd, cfg))
    agent = StreamSpeechS2STAgent(make_args(320), model=model, vocoder=OracleVocoder(vsd, vcfg))
    # the agent imposes the chunk sizes on the model (agent :395-413)
    assert model.encoder.chunk_size == 8 and model.encoder._conv_chunk() == 8
    pcm = synth.synth_pcm(3, 16000 * 2)
    wav, actions = stream(agent, pcm)
    assert len(actions) == 7                      # ceil(2 s / 320 ms) policy calls
    assert "W" in actions and len(wav) > 0 and len(wav) % 320 == 0
    assert np.isfinite(wav).all() and np.abs(wav).max() <= 1.0
    # a second utterance after reset() starts from scratch
    wav2, actions2 = stream(agent, pcm)
    assert actions2 == actions and np.array_equal(wav, wav2)


def test_policy_returns_action_types(synth_weights):
    from streamspeech_amd.agent import StreamSpeechS2STAgent
    cfg, vcfg, sd, vsd = synth_weights
    model = StreamSpeechModel.from_engine(OracleEngine(sd, cfg))
    agent = StreamSpeechS2STAgent(make_args(320), model=model, vocoder=OracleVocoder(vsd, vcfg))
    agent.states.source = synth.synth_pcm(3, 100).tolist()   # < one frame
    assert isinstance(agent.policy(), ReadAction)
    agent.states.source = synth.synth_pcm(3, 16000).tolist()
    agent.states.source_finished = True
    act = agent.policy()
    assert isinstance(act, WriteAction) and isinstance(act.content, SpeechSegment)
    assert act.content.sample_rate == 16000


def test_whole_word_mode_runs(synth_weights):
    """source_segment_size >= 640 ms: whole-word truncation + trailing <pad> position (agent :540-584)."""
    from streamspeech_amd.agent import StreamSpeechS2STAgent
    cfg, vcfg, sd, vsd = synth_weights
    model = StreamSpeechModel.from_engine(OracleEngine(sd, cfg))
    agent = StreamSpeechS2STAgent(make_args(640), model=model, vocoder=OracleVocoder(vsd, vcfg))
    assert agent.whole_word and model.encoder.chunk_size == 16 and model.encoder._conv_chunk() == 16
    wav, actions = stream(agent, synth.synth_pcm(4, 16000 * 2), segment_ms=640)
    assert "W" in actions and len(wav) % 320 == 0 and np.isfinite(wav).all()


def _stream_text(agent, pcm, segment_ms=320, sr=16000):
    step = sr * segment_ms // 1000
    out, pos = [], 0
    while True:
        chunk = pcm[pos:pos + step]
        pos += step
        finished = pos >= len(pcm)
        seg = agent.pushpop(SpeechSegment(content=chunk.tolist(), sample_rate=sr, finished=finished))
        if not seg.is_empty:
            out.append(seg.content)
        if finished:
            return out


def test_asr_and_s2tt_agents_are_prefixes_of_the_s2st_path(synth_weights):
    """agent/speech_to_text.{asr,s2tt}.streamspeech.agent.py surface over the same engine."""
    from streamspeech_amd.agent_text import StreamSpeechASRAgent, StreamSpeechS2TTAgent
    cfg, vcfg, sd, vsd = synth_weights
    eng = OracleEngine(sd, cfg)
    pcm = synth.synth_pcm(3, 16000 * 2)
    p = argparse.ArgumentParser()
    StreamSpeechASRAgent.add_args(p)
    a = p.parse_args(["--model-path", "synthetic:0", "--data-bin", "/nonexistent", "--sample-rate", "16000"])
    a.source_segment_size, a.device = 320, "gpu"
    asr = StreamSpeechASRAgent(a, model=StreamSpeechModel.from_engine(eng))
    words = _stream_text(asr, pcm)
    assert len(words) >= 1 and all(isinstance(w, str) for w in words)
    # fed as ONE finished segment the agent emits the offline CTC transcript of the whole utterance
    whole = _stream_text(asr, pcm, segment_ms=2000)
    fb = asr.feature_extractor.__class__(a, eng)(pcm.tolist())
    toks = eng.ctc_greedy(0, eng.encoder_forward(fb, 320 // 40, 8))[0]
    assert "".join(whole) == " ".join(asr.dict["source_unigram"][int(c)] for c in toks)
    s2tt = StreamSpeechS2TTAgent(a, model=StreamSpeechModel.from_engine(eng))
    outs = _stream_text(s2tt, pcm)
    assert len(outs) >= 1 and all(isinstance(w, str) for w in outs)


def test_incremental_vocoder_tail_equals_full_resynthesis(synth_weights):
    """§8f-1: synthesising only the last (new + receptive-field context) units emits the same speech as
    the reference's re-synthesis of all units at every write (agent :743-753)."""
    from streamspeech_amd.agent import synthesize_tail
    cfg, vcfg, sd, vsd = synth_weights
    rf = vcfg.receptive_field_frames()
    assert rf == 21
    units = [int(u) for u in synth.uniform(3, "inc_units", (140,), 0, 1000)]
    for dur_pred in (True, False):
        voc = OracleVocoder(vsd, vcfg)
        for upto, n_new in ((60, 7), (61, 1), (100, 30), (140, 12)):
            full, _ = synthesize_tail(voc, units[:upto], n_new, dur_pred, 0, rf)
            n_full = voc.call_lengths[-1]
            inc, _ = synthesize_tail(voc, units[:upto], n_new, dur_pred, rf + 8, rf)
            assert inc.shape == full.shape and inc.numel() > 0
            assert float(torch.sqrt(torch.mean((inc - full) ** 2))) < 1e-6
            assert voc.call_lengths[-1] == min(n_full, n_new + rf + 8)
        # a context shorter than the receptive field is detected and falls back to the full pass
        inc, _ = synthesize_tail(voc, units[:100], 5, False, 10, rf)
        assert voc.call_lengths[-1] == 100


def test_streaming_eval_latency_bookkeeping():
    """bench.py --mode streaming: SimulEval's speech-output timing model (evaluator/instance.py:349-366, 386-415;
    scorers/latency_scorer.py:540-587) on a scripted agent: delays are the source milliseconds received, playback
    intervals never overlap, RTF = end of the last interval / source length."""
    from streamspeech_amd import streaming_eval as SE
    from streamspeech_amd.simuleval_shim import EmptySegment

    class Scripted:
        """emits 0.5 s of speech after the 2nd and 4th chunk, 0.25 s after the last"""
        def __init__(self):
            self.n = 0

        def pushpop(self, seg):
            self.n += 1
            if self.n in (2, 4):
                return SpeechSegment(content=[0.0] * 8000, sample_rate=16000, finished=False)
            if seg.finished:
                return SpeechSegment(content=[0.0] * 4000, sample_rate=16000, finished=True)
            return EmptySegment()

    r = SE.run_utterance(Scripted(), np.zeros(16000 * 2, np.float32), 320, sync=False)    # 2 s = 7 chunks
    assert r["actions"] == "RWRWRRW" and r["writes"] == 3 and r["samples_out"] == 20000
    # delays 640, 1280, 2000 ms; durations 500, 500, 250 -> intervals [640,1140] [1280,1780] [2000,2250]
    assert abs(r["StartOffset"] - 640.0) < 1e-6 and abs(r["EndOffset"] - 250.0) < 1e-6
    assert abs(r["RTF"] - 2250.0 / 2000.0) < 1e-9
    assert r["RTF_CA"] >= r["RTF"] and r["StartOffset_CA"] >= r["StartOffset"]
    # overlapping playback is serialised: two 0.5-s segments 320 ms apart
    iv = SE._intervals([320.0, 640.0], [500.0, 500.0])
    assert iv == [(320.0, 500.0), (820.0, 500.0)]
    s = SE.summarize([r])
    assert s["utterances"] == 1 and s["policy_calls"] == 7 and s["RTF"] == round(2250.0 / 2000.0, 4)


def test_user_dir_registers_into_a_fairseq_registry(monkeypatch):
    """--user-dir streamspeech_amd/fairseq_user_dir: with a fairseq present, the names the reference's user dir
    registers (streamspeech_model.py:57,418; tasks/speech_to_speech_ctc.py:11; agent/tts/vocoder.py:30) land in ITS
    registries.  fairseq is not importable in this image, so a stand-in with fairseq's decorator contracts (same names,
    base-class check, duplicate refusal: fairseq/models/__init__.py:109-170, fairseq/tasks/__init__.py) is planted."""
    import importlib
    import sys
    import types

    class BaseFairseqModel:
        pass

    class LegacyFairseqTask:
        def __init__(self, args):
            self.args = args

    models, archs, tasks = {}, {}, {}

    def register_model(name):
        def deco(cls):
            if name in models:
                raise ValueError("Cannot register duplicate model ({})".format(name))
            if not issubclass(cls, BaseFairseqModel):
                raise ValueError("Model ({}: {}) must extend BaseFairseqModel".format(name, cls.__name__))
            models[name] = cls
            return cls
        return deco

    def register_model_architecture(model_name, arch_name):
        def deco(fn):
            if model_name not in models:
                raise ValueError("Cannot register model architecture for unknown model type ({})".format(model_name))
            archs[arch_name] = fn
            return fn
        return deco

    def register_task(name):
        def deco(cls):
            if name in tasks:
                raise ValueError("Cannot register duplicate task ({})".format(name))
            if not issubclass(cls, LegacyFairseqTask):
                raise ValueError("Task ({}: {}) must extend FairseqTask".format(name, cls.__name__))
            tasks[name] = cls
            return cls
        return deco

    fs = types.ModuleType("fairseq")
    fm = types.ModuleType("fairseq.models")
    ft = types.ModuleType("fairseq.tasks")
    fm.BaseFairseqModel, fm.register_model, fm.register_model_architecture = BaseFairseqModel, register_model, register_model_architecture
    ft.LegacyFairseqTask, ft.register_task = LegacyFairseqTask, register_task
    fs.models, fs.tasks = fm, ft
    for k, v in (("fairseq", fs), ("fairseq.models", fm), ("fairseq.tasks", ft)):
        monkeypatch.setitem(sys.modules, k, v)
    sys.modules.pop("streamspeech_amd.fairseq_user_dir", None)
    ud = importlib.import_module("streamspeech_amd.fairseq_user_dir")
    assert ud.REGISTERED is True
    assert set(models) == {"streamspeech", "CodeHiFiGANVocoderWithDur"} and "streamspeech" in archs and "speech_to_speech_ctc" in tasks
    task = tasks["speech_to_speech_ctc"].setup_task(argparse.Namespace())
    assert len(task.target_dictionary) == 1005 and task.target_dictionary.blank_index == 1004
    ns = argparse.Namespace()
    archs["streamspeech"](ns)
    assert ns.enc_layers == 12 and ns.ctc_upsample == 25
    # a second import attempt (the reference's user dir already there) is refused, not fatal
    from streamspeech_amd.modules import register_with_fairseq
    assert register_with_fairseq() is False
    sys.modules.pop("streamspeech_amd.fairseq_user_dir", None)
