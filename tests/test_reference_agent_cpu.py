"""Host-side generators and SimulEval agents pinned to the REFERENCE'S OWN classes, on CPU:
  * tests/golden/generators.npz  <- agent/ctc_decoder.py, agent/ctc_generator.py, agent/sequence_generator.py
    (+ fairseq/fairseq/search.py) executed by oracle/make_golden_agent.py;
  * tests/golden/agent_traces.npz <- the reference agents' policy() driven chunk by chunk.
Here the repo's generators / agents run over the oracle-backed engine (no GPU); the GPU twins are in
tests/test_reference_agent_gpu.py.  With /root/reference present the reference agent is also run live on
inputs the fixtures do not contain."""
import numpy as np
import pytest
import torch

from oracle import ref_loader
from streamspeech_amd import synth
from streamspeech_amd.config import ModelConfig
from streamspeech_amd.generators import CTCDecoder, CTCSequenceGenerator, SequenceGenerator
from streamspeech_amd.modules import StreamSpeechModel
from tests import ref_fixtures as RF
from tests.oracle_engine import OracleEngine, OracleVocoder

import json
import os


@pytest.fixture(scope="module")
def cmvn(golden_dir):
    g = np.load(os.path.join(golden_dir, "gcmvn_fr-en.npz"))
    return g["mean"], g["std"]


def test_synthetic_dictionary_matches_fixture(synth_weights):
    cfg = synth_weights[0]
    g = RF.generators_gold()
    d = RF.dictionaries(cfg)["target_unigram"]
    assert [s.startswith("▁") for s in d.symbols] == g["dict/target_unigram_word_initial"].tolist()


def test_ctc_decoder_matches_reference_class(golden_dir, synth_weights):
    """a8: agent/ctc_decoder.py CTCDecoder.generate -- tokens, kept frame indices, prefix splice."""
    cfg, _, sd, _ = synth_weights
    g, ge = RF.generators_gold(), np.load(os.path.join(golden_dir, "encoder.npz"))
    eng, d = OracleEngine(sd, cfg), RF.dictionaries(cfg)
    for tag in ("offline", "c8"):
        enc = {"encoder_out": [torch.from_numpy(ge[f"enc_{tag}"])[:, None]]}
        for head, hid in (("source_unigram", 0), ("ctc_target_unigram", 1)):
            hyp = CTCDecoder(d[head], eng, hid).generate(enc, aux_task_name=head)[0][0]
            assert hyp["tokens"].tolist() == g[f"ctc/{head}_{tag}_tokens"].tolist()
            assert list(hyp["index"]) == g[f"ctc/{head}_{tag}_index"].tolist()
            assert hyp["org_tokens"].tolist() == g[f"ctc/{head}_{tag}_org"].tolist()
    enc = {"encoder_out": [torch.from_numpy(ge["enc_offline"])[:, None]]}
    hyp = CTCDecoder(d["source_unigram"], eng, 0).generate(enc, prefix=torch.from_numpy(g["ctc/prefix_in"]).long(),
                                                            aux_task_name="source_unigram")[0][0]
    assert hyp["tokens"].tolist() == g["ctc/prefix_tokens"].tolist()
    assert list(hyp["index"]) == g["ctc/prefix_index"].tolist()


def test_unit_generator_matches_reference_class(golden_dir, synth_weights):
    """a13: agent/ctc_generator.py CTCSequenceGenerator.generate on the reference T2U input."""
    cfg, _, sd, _ = synth_weights
    g, gd = RF.generators_gold(), np.load(os.path.join(golden_dir, "decoders.npz"))
    hyp = CTCSequenceGenerator(RF.dictionaries(cfg)["tgt"], OracleEngine(sd, cfg)).generate(
        torch.from_numpy(gd["mt_features"]))[0][0]
    assert hyp["tokens"].tolist() == g["unit/tokens"].tolist()
    assert hyp["org_tokens"].tolist() == g["unit/org"].tolist()


def mt_cases(g):
    for name in json.loads(str(g["mt/cases"])):
        yield name, json.loads(str(g[f"mt/{name}/args"])), g[f"mt/{name}/tokens"].tolist()


def run_mt_case(engine_factory, enc, cfg, args):
    """generate_decoder with the constructor / call arguments the reference class was given."""
    eos = cfg.eos if args["eos"] is None else args["eos"]
    c = ModelConfig(**{**cfg.__dict__, "eos": eos}) if eos != cfg.eos else cfg
    eng = engine_factory(c)
    d = RF.dictionaries(cfg)["target_unigram"]
    gen = SequenceGenerator(eng, d, beam_size=1, max_len_a=0, max_len_b=args["max_len_b"], max_len=0,
                            min_len=args["min_len"], eos=eos, use_incremental_states=False)
    pre = None if args["prefix"] is None else torch.tensor([args["prefix"]])
    out = gen.generate_decoder([{"encoder_out": [enc[:, None]]}], torch.zeros((1, 83, 80)), torch.tensor([83]), {"id": 1},
                               pre, None, None, aux_task_name="target_unigram", max_new_tokens=args["max_new_tokens"])
    return out[0][0]["tokens"].tolist()


def test_sequence_generator_matches_reference_class(golden_dir, synth_weights):
    """a9: agent/sequence_generator.py generate_decoder (beam 1 over fairseq BeamSearch): free run to the forced
    eos at max_len, prefix continuation with max_new_tokens, final call with a prefix, a stop token emitted
    right after the prefix, and min_len banning it."""
    cfg, _, sd, _ = synth_weights
    g, ge = RF.generators_gold(), np.load(os.path.join(golden_dir, "encoder.npz"))
    enc = torch.from_numpy(ge["enc_offline"])
    n = 0
    for name, args, want in mt_cases(g):
        got = run_mt_case(lambda c: OracleEngine(sd, c), enc, cfg, args)
        assert got == want, (name, got, want)
        n += 1
    assert n >= 7


def _make_agent(kind, case, engine, vocoder, cfg):
    from streamspeech_amd.agent import StreamSpeechS2STAgent
    from streamspeech_amd.agent_text import StreamSpeechASRAgent, StreamSpeechS2TTAgent
    cls = {"s2st": StreamSpeechS2STAgent, "s2tt": StreamSpeechS2TTAgent, "asr": StreamSpeechASRAgent}[kind]
    args = RF.agent_args(cls, case["segment_ms"], case["sr"], case["over"])
    model = StreamSpeechModel.from_engine(engine)
    agent = cls(args, model=model, vocoder=vocoder) if kind == "s2st" else cls(args, model=model)
    return RF.set_dicts(agent, cfg)


@pytest.mark.parametrize("name", ["s2st_320_a", "s2st_320_b", "s2st_320_k3", "s2st_640_a", "s2st_640_b", "s2st_960_a",
                                  "s2st_320_48k"])
def test_s2st_agent_trace_matches_reference_agent(synth_weights, cmvn, name):
    """a16: READ/WRITE trace, per-call sample counts and waveform of agent/speech_to_speech.streamspeech.agent.py
    policy() (reference modules on CPU).  640 / 960 ms = whole-word mode with non-final writes."""
    cfg, vcfg, sd, vsd = synth_weights
    g, cases = RF.traces_gold()
    agent = _make_agent("s2st", cases[name], OracleEngine(sd, cfg, *cmvn), OracleVocoder(vsd, vcfg), cfg)
    recs = RF.run_case(agent, cases[name])
    RF.check_s2st_trace(g, name, recs, 1e-5)


@pytest.mark.parametrize("name", ["s2tt_320_a", "s2tt_640_a", "asr_320_a"])
def test_text_agent_trace_matches_reference_agent(synth_weights, cmvn, name):
    """f2: text increments of agent/speech_to_text.{s2tt,asr}.streamspeech.agent.py policy()."""
    cfg, vcfg, sd, vsd = synth_weights
    g, cases = RF.traces_gold()
    agent = _make_agent(cases[name]["kind"], cases[name], OracleEngine(sd, cfg, *cmvn), None, cfg)
    RF.check_text_trace(g, name, RF.run_case(agent, cases[name]))


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not present")
@pytest.mark.parametrize("segment_ms,seed,seconds", [(320, 41, 2.2), (640, 28, 4.0)])
def test_live_reference_agent(synth_weights, cmvn, golden_dir, segment_ms, seed, seconds):
    """The reference agent run live (not from fixtures) on another utterance, against the repo's agent."""
    from oracle import ref_agent
    cfg, vcfg, sd, vsd = synth_weights
    case = dict(kind="s2st", segment_ms=segment_ms, sr=16000, seed=seed, seconds=seconds, over={})
    with torch.no_grad():
        ref = ref_agent.make_agent(sd, vsd, cfg, vcfg, segment_ms, 16000,
                                   cmvn_npz=os.path.join(golden_dir, "gcmvn_fr-en.npz"))
        want = ref_agent.stream(ref, RF.trace_pcm(seed, 16000, seconds), segment_ms, 16000)
    agent = _make_agent("s2st", case, OracleEngine(sd, cfg, *cmvn), OracleVocoder(vsd, vcfg), cfg)
    got = RF.run_case(agent, case)
    assert [w for w, _, _ in got] == [not r.is_empty for r in want]
    a = np.concatenate([np.asarray(c, np.float32) for _, c, _ in got if c is not None])
    b = np.concatenate([np.asarray(r.content, np.float32) for r in want if not r.is_empty])
    assert a.shape == b.shape and float(np.sqrt(np.mean((a - b) ** 2))) < 1e-5
