"""Helpers for the fixtures generated from the reference's own generator / agent classes
(oracle/make_golden_agent.py -> tests/golden/generators.npz, agent_traces.npz).  TEST ONLY."""
import argparse
import json
import os

import numpy as np

from streamspeech_amd import synth
from streamspeech_amd.modules import Dictionary

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def text_symbols(n, tag):
    """The synthetic dictionaries the fixtures were generated with (oracle/ref_agent.py SynthDictionary):
    two out of three subwords are word-initial."""
    return [("" if i % 3 == 0 else "▁") + f"{tag}{i}" for i in range(n - 4)]


def dictionaries(cfg):
    return {"tgt": Dictionary.units(1000),
            "target_unigram": Dictionary(text_symbols(cfg.tgt_vocab, "t")),
            "source_unigram": Dictionary(text_symbols(cfg.src_vocab, "s")),
            "ctc_target_unigram": Dictionary(text_symbols(cfg.tgt_vocab, "t"))}


def generators_gold():
    return np.load(os.path.join(GOLD, "generators.npz"))


def traces_gold():
    g = np.load(os.path.join(GOLD, "agent_traces.npz"))
    cases = {c[0]: dict(kind=c[1], segment_ms=c[2], sr=c[3], seed=c[4], seconds=c[5], over=c[6])
             for c in json.loads(str(g["cases"]))}
    return g, cases


def trace_pcm(seed, sr, seconds):
    """Same signal as oracle/make_golden_agent.py trace_pcm (seeded, a pure function of its arguments)."""
    n = int(round(sr * seconds))
    if sr == 16000:
        return synth.synth_pcm(seed, n)
    base = synth.synth_pcm(seed, n * 16000 // sr + 2)
    t = np.arange(n, dtype=np.float64) * (16000.0 / sr)
    i = np.floor(t).astype(np.int64)
    f = (t - i).astype(np.float32)
    return (base[i] * (1 - f) + base[i + 1] * f).astype(np.float32)


def agent_args(cls, segment_ms, sr, over=None, extra=()):
    p = argparse.ArgumentParser()
    cls.add_args(p)
    a = p.parse_args(["--model-path", "synthetic:0", "--data-bin", "/nonexistent", "--vocoder", "synthetic:0",
                      "--dur-prediction", "--sample-rate", str(sr), *extra])
    a.source_segment_size, a.device = segment_ms, "gpu"
    for k, v in (over or {}).items():
        setattr(a, k, v)
    return a


def set_dicts(agent, cfg):
    d = dictionaries(cfg)
    for k, v in d.items():
        agent.dict[k] = v
    for name, key in (("generator_mt", "target_unigram"), ("asr_ctc_generator", "source_unigram"),
                      ("st_ctc_generator", "ctc_target_unigram"), ("ctc_generator", "tgt")):
        if hasattr(agent, name):
            getattr(agent, name).tgt_dict = d[key]
    return agent


def run_case(agent, case):
    """SentenceLevelEvaluator's loop (SimulEval evaluator.py:216-235) -> per-call (is_write, content)."""
    from streamspeech_amd.simuleval_shim import SpeechSegment
    pcm = trace_pcm(case["seed"], case["sr"], case["seconds"])
    step = case["sr"] * case["segment_ms"] // 1000
    out, pos = [], 0
    while True:
        chunk = pcm[pos:pos + step]
        pos += step
        finished = pos >= len(pcm)
        seg = agent.pushpop(SpeechSegment(content=chunk.tolist(), sample_rate=case["sr"], finished=finished))
        out.append((not seg.is_empty, None if seg.is_empty else seg.content, bool(seg.finished)))
        if finished:
            return out


def check_s2st_trace(g, name, recs, rms_tol):
    want_actions = g[f"{name}/actions"].tolist()
    got_actions = [int(w) for w, _, _ in recs]
    assert got_actions == want_actions, (name, got_actions, want_actions)
    assert [f for _, _, f in recs] == g[f"{name}/finished"].tolist()
    lens = [0 if c is None else len(c) for _, c, _ in recs]
    assert lens == g[f"{name}/wav_len"].tolist(), (name, lens, g[f"{name}/wav_len"].tolist())
    wav = np.concatenate([np.asarray(c, np.float32) for _, c, _ in recs if c is not None] or [np.zeros(0, np.float32)])
    ref = g[f"{name}/wav"]
    rms = float(np.sqrt(np.mean((wav - ref) ** 2))) if len(ref) else 0.0
    assert rms < rms_tol, (name, rms)
    return rms


def check_text_trace(g, name, recs):
    want = json.loads(str(g[f"{name}/text"]))
    got = ["" if c is None else c for _, c, _ in recs]
    assert [int(w) for w, _, _ in recs] == g[f"{name}/actions"].tolist()
    assert got == want, (name, got, want)
