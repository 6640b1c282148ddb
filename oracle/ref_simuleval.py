"""Loads SimulEval's own latency scorers and speech-output instance from /root/reference for pinning
streamspeech_amd/streaming_eval.py.  TEST INFRASTRUCTURE (fixture generation + CPU tests only).

`import simuleval` fails in this image (yt_dlp, soundfile, textgrid absent), but
SimulEval/simuleval/evaluator/instance.py and .../scorers/latency_scorer.py are plain Python: the two
files are executed where they lie with a private stub tree for the three imports they cannot satisfy
(`textgrid`: only used by the alignment scorers; `simuleval.data.dataloader`: type annotations;
`soundfile.write`: the wav dump inside SpeechOutputInstance.summarize, replaced by a no-op).
Nothing is copied into the repo."""
import importlib.util
import os
import sys
import types

REF = os.environ.get("STREAMSPEECH_REFERENCE", "/root/reference")
_SE = os.path.join(REF, "SimulEval", "simuleval")


def available() -> bool:
    return os.path.isfile(os.path.join(_SE, "evaluator", "scorers", "latency_scorer.py"))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load():
    """-> (instance module, latency_scorer module) of the reference, executed in place."""
    saved = {k: sys.modules.get(k) for k in ("textgrid", "simuleval", "simuleval.data", "simuleval.data.segments",
                                             "simuleval.data.dataloader", "simuleval.evaluator", "simuleval.evaluator.instance")}
    try:
        sys.modules["textgrid"] = types.ModuleType("textgrid")
        for pkg in ("simuleval", "simuleval.data", "simuleval.evaluator"):
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
        _load("simuleval.data.segments", os.path.join(_SE, "data", "segments.py"))
        dl = types.ModuleType("simuleval.data.dataloader")
        dl.SpeechToTextDataloader = dl.TextToTextDataloader = object
        sys.modules["simuleval.data.dataloader"] = dl
        inst = _load("simuleval.evaluator.instance", os.path.join(_SE, "evaluator", "instance.py"))
        inst.soundfile = types.SimpleNamespace(write=lambda *a, **k: None)
        inst.IS_IMPORT_SOUNDFILE = True
        sc = _load("simuleval.evaluator.scorers.latency_scorer", os.path.join(_SE, "evaluator", "scorers", "latency_scorer.py"))
        return inst, sc
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.modules.pop("simuleval.evaluator.scorers.latency_scorer", None)


def make_speech_instance(inst_mod, source_samples, sample_rate=16000):
    """A reference SpeechToSpeechInstance (= SpeechInputInstance + SpeechOutputInstance, instance.py:430) without a
    dataloader / audio file: `source_samples` zeros at `sample_rate`; send_source() / receive_prediction() / summarize()
    are the reference's own."""
    import argparse
    import tempfile
    ins = inst_mod.SpeechToSpeechInstance.__new__(inst_mod.SpeechToSpeechInstance)
    ins.index, ins.finish_prediction = 0, False
    ins.dataloader = types.SimpleNamespace(get_source_audio_path=lambda i: "synthetic")
    ins.reference, ins.source = None, [0.0] * int(source_samples)
    ins.reset()
    ins.args = argparse.Namespace(output=tempfile.mkdtemp(prefix="ss_simuleval_"), eval_latency_unit="word")
    ins.latency_unit = "word"
    ins.sample_rate_value, ins.sample_list, ins.source_finished_reading = sample_rate, None, False
    ins.audio_info = types.SimpleNamespace(samplerate=sample_rate)
    ins.prediction_time, ins.durations, ins.intervals, ins.target_sample_rate = 0, [], [], -1
    return ins
