"""streamspeech_amd/streaming_eval.py (the latency bookkeeping bench.py's streaming lines report) against SimulEval's
own classes: SpeechOutputInstance.receive_prediction / summarize (evaluator/instance.py:349-415) and the RTF /
StartOffset / EndOffset scorers (evaluator/scorers/latency_scorer.py:540-587), executed from /root/reference
(oracle/ref_simuleval.py) -- and against the committed fixture of their outputs, so the test also runs where the
reference is absent."""
import json
import os

import numpy as np
import pytest

from streamspeech_amd import streaming_eval as SE
from streamspeech_amd.simuleval_shim import EmptySegment, SpeechSegment

FIX = os.path.join(os.path.dirname(__file__), "golden", "simuleval_latency.json")
SR = 16000


class ScriptedAgent:
    """pushpop() returns a scripted sequence: None -> read (empty segment), n -> n output samples."""

    def __init__(self, script):
        self.script, self.i = list(script), 0

    def pushpop(self, seg):
        n = self.script[self.i]
        self.i += 1
        fin = bool(seg.finished)
        if n is None:
            return EmptySegment(finished=fin)
        return SpeechSegment(content=[0.0] * n, sample_rate=SR, finished=fin)


CASES = {
    # source samples, segment ms, script per pushpop call
    "steady": (16000 * 3 + 400, 320, [None, None, 3200, None, 6400, None, None, 1600, None, 9600]),
    "gaps_and_overlap": (16000 * 2, 320, [None, 16000, None, 320, None, None, 12800]),
    "single_write_at_end": (16000 * 1 + 7, 320, [None, None, None, 4800]),
}


def _ours(case):
    n, seg_ms, script = CASES[case]
    r = SE.run_utterance(ScriptedAgent(script), np.zeros(n, np.float32), seg_ms, sr=SR, sync=False)
    return {"RTF": r["RTF"], "StartOffset": r["StartOffset"], "EndOffset": r["EndOffset"], "source_ms": r["source_ms"],
            "writes": r["writes"], "calls": r["calls"]}


def _reference(case):
    from oracle import ref_simuleval as RS
    inst, sc = RS.load()
    n, seg_ms, script = CASES[case]
    ins = RS.make_speech_instance(inst, n, SR)
    for k in script:
        fin = ins.send_source(seg_ms).finished            # the reference's own source clock (instance.py:262-296)
        seg = (inst.EmptySegment(finished=fin) if k is None
               else inst.SpeechSegment(content=[0.0] * k, sample_rate=SR, finished=fin))
        ins.receive_prediction(seg)
    if not ins.intervals:
        ins.summarize()
    return {"RTF": sc.RTFScorer().compute(ins), "StartOffset": sc.StartOffsetScorer().compute(ins),
            "EndOffset": sc.EndOffsetScorer().compute(ins), "source_ms": ins.source_length,
            "writes": len(ins.delays), "calls": len(script)}


@pytest.mark.parametrize("case", sorted(CASES))
def test_matches_committed_reference_outputs(case):
    gold = json.load(open(FIX))[case]
    ours = _ours(case)
    for k, v in gold.items():
        assert ours[k] == pytest.approx(v, rel=1e-12, abs=1e-9), (case, k)


@pytest.mark.parametrize("case", sorted(CASES))
def test_matches_live_reference_scorers(case):
    from oracle import ref_simuleval as RS
    if not RS.available():
        pytest.skip("/root/reference not present")
    ref, ours = _reference(case), _ours(case)
    for k, v in ref.items():
        assert ours[k] == pytest.approx(v, rel=1e-12, abs=1e-9), (case, k)
    assert json.load(open(FIX))[case] == pytest.approx(ref)


def test_computation_aware_variants_only_add_compute_time():
    """The *_CA figures are this package's extension (elapsed-based playback intervals): SimulEval's RTFScorer reads
    ins.intervals, which summarize() builds from `delays` whatever `computation_aware` says (instance.py:349-366,
    latency_scorer.py:583-587), so its RTF has no computation-aware form for speech output.  Ours must never be below
    the non-CA figure and must coincide when compute time is zero."""
    n, seg_ms, script = CASES["steady"]
    r = SE.run_utterance(ScriptedAgent(script), np.zeros(n, np.float32), seg_ms, sr=SR, sync=False)
    assert r["RTF_CA"] >= r["RTF"] and r["StartOffset_CA"] >= r["StartOffset"] and r["EndOffset_CA"] >= r["EndOffset"]
    assert r["RTF_CA"] - r["RTF"] < 0.05     # a scripted agent computes nothing


if __name__ == "__main__":     # regenerate the fixture from the reference classes
    out = {c: _reference(c) for c in sorted(CASES)}
    json.dump(out, open(FIX, "w"), indent=1)
    print(json.dumps(out, indent=1))
