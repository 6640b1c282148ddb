"""Writes tests/golden/generators.npz and tests/golden/agent_traces.npz from the reference's OWN generator
classes and SimulEval agents, executed on CPU in place (oracle/ref_agent.py).  TEST INFRASTRUCTURE; run in
the build container (needs /root/reference):  python -m oracle.make_golden_agent

generators.npz -- outputs of agent/ctc_decoder.py CTCDecoder.generate, agent/ctc_generator.py
CTCSequenceGenerator.generate and agent/sequence_generator.py SequenceGenerator.generate_decoder
(fairseq/fairseq/search.py BeamSearch underneath) on the encoder / decoder goldens of make_golden.py.
agent_traces.npz -- per-policy() records of agent/speech_to_speech.streamspeech.agent.py (READ/WRITE,
emitted waveform, unit history) and of the S2TT / ASR agents (text increments) on seeded PCM.
"""
import json
import os
import sys

import numpy as np
import torch

from streamspeech_amd import synth
from streamspeech_amd.config import ModelConfig, VocoderConfig

from . import ref_agent

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# (name, kind, segment ms, sample rate, pcm seed, seconds, agent-arg overrides)
TRACE_CASES = [
    ("s2st_320_a", "s2st", 320, 16000, 3, 2.0, {}),
    ("s2st_320_b", "s2st", 320, 16000, 9, 5.0, {}),
    ("s2st_320_k3", "s2st", 320, 16000, 11, 3.0, {"lagging_k1": 3, "stride_n": 2}),
    ("s2st_640_a", "s2st", 640, 16000, 4, 3.0, {}),
    ("s2st_640_b", "s2st", 640, 16000, 22, 4.0, {}),         # whole-word mode with non-final WRITEs
    ("s2st_960_a", "s2st", 960, 16000, 29, 4.0, {}),
    ("s2st_320_48k", "s2st", 320, 48000, 13, 1.6, {}),       # resampled front-end (sox -> scipy-pinned polyphase)
    ("s2tt_320_a", "s2tt", 320, 16000, 3, 2.0, {}),
    ("s2tt_640_a", "s2tt", 640, 16000, 4, 3.0, {}),
    ("asr_320_a", "asr", 320, 16000, 3, 2.0, {}),
]


def trace_pcm(seed, sr, seconds):
    n = int(round(sr * seconds))
    if sr == 16000:
        return synth.synth_pcm(seed, n)
    # band-limited content at the higher rate: 16 kHz noise linearly interpolated (any deterministic signal will do)
    base = synth.synth_pcm(seed, n * 16000 // sr + 2)
    t = np.arange(n, dtype=np.float64) * (16000.0 / sr)
    i = np.floor(t).astype(np.int64)
    f = (t - i).astype(np.float32)
    return (base[i] * (1 - f) + base[i + 1] * f).astype(np.float32)


def _spy(agent, units_hist):
    orig = agent.policy

    def policy():                          # zero-parameter like the reference's (SimulEval inspects the signature)
        act = orig()
        units_hist.append(list(agent.unit) if agent.unit is not None else [])
        return act
    return policy


def run_traces(sd, vsd, cfg, vcfg, cmvn_npz):
    fix, summary = {}, {}
    for name, kind, seg, sr, seed, secs, over in TRACE_CASES:
        agent = ref_agent.make_agent(sd, vsd, cfg, vcfg, seg, sr, kind=kind, cmvn_npz=cmvn_npz, **over)
        pcm = trace_pcm(seed, sr, secs)
        units_hist = []
        if kind == "s2st":
            # record the unit history the agent keeps (self.unit) after each call: wrap policy, do not touch it
            agent.policy = _spy(agent, units_hist)
        recs = ref_agent.stream(agent, pcm, seg, sr)
        actions = np.array([0 if r.is_empty else 1 for r in recs], np.int8)
        fix[f"{name}/actions"] = actions
        fix[f"{name}/finished"] = np.array([bool(r.finished) for r in recs], np.bool_)
        if kind == "s2st":
            lens = [0 if r.is_empty else len(r.content) for r in recs]
            fix[f"{name}/wav_len"] = np.array(lens, np.int32)
            wav = [np.asarray(r.content, np.float32) for r in recs if not r.is_empty]
            fix[f"{name}/wav"] = np.concatenate(wav) if wav else np.zeros(0, np.float32)
            # self.unit is cleared by the final reset(): keep the longest history seen
            fix[f"{name}/units"] = np.array(max(units_hist, key=len) if units_hist else [], np.int32)
            fix[f"{name}/n_units"] = np.array([len(u) for u in units_hist], np.int32)
            summary[name] = {"actions": "".join("RW"[a] for a in actions), "samples": int(sum(lens)),
                             "units": int(len(fix[f"{name}/units"]))}
        else:
            texts = ["" if r.is_empty else r.content for r in recs]
            fix[f"{name}/text"] = np.array(json.dumps(texts))
            summary[name] = {"actions": "".join("RW"[a] for a in actions), "text": "".join(texts)[:60]}
        print(name, summary[name], flush=True)
    fix["cases"] = np.array(json.dumps([[n, k, s, r, sd_, sec, o] for n, k, s, r, sd_, sec, o in TRACE_CASES]))
    np.savez_compressed(os.path.join(OUT, "agent_traces.npz"), **fix)
    return summary


def run_generators(sd, cfg):
    G = ref_agent.generators()
    dicts = ref_agent.make_dicts(cfg)
    model = ref_agent.build_model(sd, cfg, dicts=dicts)
    ge = np.load(os.path.join(OUT, "encoder.npz"))
    gd = np.load(os.path.join(OUT, "decoders.npz"))
    fix, summary = {}, {}
    # ---- a8: CTCDecoder.generate (agent/ctc_decoder.py:39-111) on the reference encoder outputs ----
    for tag in ("offline", "c8"):
        enc_out = {"encoder_out": [torch.from_numpy(ge[f"enc_{tag}"])[:, None]], "encoder_padding_mask": []}
        for head in ("source_unigram", "ctc_target_unigram"):
            hyp = G.CTCDecoder(dicts[head], [model]).generate(enc_out, aux_task_name=head)[0][0]
            fix[f"ctc/{head}_{tag}_tokens"] = hyp["tokens"].numpy().astype(np.int32)
            fix[f"ctc/{head}_{tag}_index"] = np.array(hyp["index"], np.int32)
            fix[f"ctc/{head}_{tag}_org"] = hyp["org_tokens"].numpy().astype(np.int32)
            # the re-derived fixtures of make_golden.py must agree with the class
            assert fix[f"ctc/{head}_{tag}_tokens"].tolist() == ge[f"{head}_{tag}_tokens"].tolist()
            assert fix[f"ctc/{head}_{tag}_index"].tolist() == ge[f"{head}_{tag}_index"].tolist()
            assert fix[f"ctc/{head}_{tag}_org"].tolist() == ge[f"{head}_{tag}_raw"].tolist()
    # prefix splice (agent/ctc_decoder.py:90-92)
    enc_out = {"encoder_out": [torch.from_numpy(ge["enc_offline"])[:, None]], "encoder_padding_mask": []}
    pre = torch.tensor([[7, 7, 0, 9]])
    hyp = G.CTCDecoder(dicts["source_unigram"], [model]).generate(enc_out, prefix=pre, aux_task_name="source_unigram")[0][0]
    fix["ctc/prefix_in"] = pre.numpy().astype(np.int32)
    fix["ctc/prefix_tokens"] = hyp["tokens"].numpy().astype(np.int32)
    fix["ctc/prefix_index"] = np.array(hyp["index"], np.int32)
    # ---- a13: CTCSequenceGenerator.generate (agent/ctc_generator.py:40-123) ----
    t2u_out = {"encoder_out": [torch.from_numpy(gd["t2u_out"])[:, None]], "encoder_padding_mask": []}
    hyp = G.CTCSequenceGenerator(dicts["tgt"], [model]).generate(t2u_out)[0][0]
    fix["unit/tokens"] = hyp["tokens"].numpy().astype(np.int32)
    fix["unit/org"] = hyp["org_tokens"].numpy().astype(np.int32)
    assert fix["unit/org"].tolist() == gd["unit_raw"].tolist()
    # ---- a9: SequenceGenerator.generate_decoder (agent/sequence_generator.py:165-582) ----
    enc = torch.from_numpy(ge["enc_offline"])
    enc_outs = [{"encoder_out": [enc[:, None]], "encoder_padding_mask": [], "encoder_embedding": [],
                 "encoder_states": [], "src_tokens": [], "src_lengths": []}]
    src_tokens = torch.zeros((1, int(ge["T"]), 80))
    src_lengths = torch.tensor([int(ge["T"])])

    def gen(prefix, max_new, eos=None, max_len_b=100, min_len=1):
        d = dicts["target_unigram"]
        g = G.SequenceGenerator([model], d, beam_size=1, max_len_a=0, max_len_b=max_len_b, max_len=0, min_len=min_len,
                                normalize_scores=True, len_penalty=1.0, unk_penalty=0.0, temperature=1.0,
                                match_source_len=False, no_repeat_ngram_size=0, search_strategy=G.BeamSearch(d),
                                eos=d.eos() if eos is None else eos, symbols_to_strip_from_output=None,
                                use_incremental_states=False)
        pre = None if prefix is None else torch.tensor([prefix])
        out = g.generate_decoder(enc_outs, src_tokens, src_lengths, {"id": 1}, pre, None, None,
                                 aux_task_name="target_unigram", max_new_tokens=max_new)
        return out[0][0]["tokens"].tolist()

    cases = []
    toks9 = gd["mt_tokens_in"].tolist()[1:]
    cases.append(("free_b12", None, -1, None, 12, 1))          # no prefix, max_len = max_len_b: forced eos at step 12
    cases.append(("prefix9_new5", toks9, 5, None, 100, 1))     # streaming continuation (agent :520-538)
    cases.append(("new1", None, 1, None, 100, 1))              # first streaming call with one CTC token
    cases.append(("prefix9_new1", toks9, 1, None, 100, 1))
    cases.append(("prefix3_final", toks9[:3], -1, None, 20, 1))  # source finished: max_new_tokens = -1 with a prefix
    # early eos: the search's stop token is a constructor argument (agent :162-180 passes tgt_dict_mt.eos());
    # random weights never emit </s>, so pick as "eos" a token the model does emit after the start
    # (a search over 400 candidate start tokens found none that comes back unprompted); a prefix that ends in the
    # stop token makes the repeat-prone random model emit it again at the first new step -> stop right after the prefix
    cases.append(("stop_after_prefix", toks9[:2] + [3013], -1, 3013, 30, 1))
    # min_len: eos is banned while step < min_len (sequence_generator.py:382-385; steps count from the prefix length)
    cases.append(("stop_after_prefix_min6", toks9[:2] + [3013], -1, 3013, 30, 6))
    for name, prefix, max_new, eos, mlb, minl in cases:
        out = gen(prefix, max_new, eos, mlb, minl)
        fix[f"mt/{name}/tokens"] = np.array(out, np.int32)
        fix[f"mt/{name}/args"] = np.array(json.dumps({"prefix": prefix, "max_new_tokens": max_new, "eos": eos,
                                                      "max_len_b": mlb, "min_len": minl}))
        summary["mt_" + name] = out
        print("mt", name, out, flush=True)
    assert fix["mt/prefix9_new5/tokens"].tolist() == gd["mt_greedy_prefix9_new5"].tolist()
    fix["mt/cases"] = np.array(json.dumps([c[0] for c in cases]))
    # the word-boundary table the whole-word agent path consults
    fix["dict/target_unigram_word_initial"] = np.array([s.startswith("▁") for s in dicts["target_unigram"].symbols])
    np.savez_compressed(os.path.join(OUT, "generators.npz"), **fix)
    return summary


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.manual_seed(0)
    cfg, vcfg = ModelConfig(), VocoderConfig()
    sd = synth.make_model_state_dict(0, cfg)
    vsd = synth.make_vocoder_state_dict(0, vcfg)
    with torch.no_grad():
        s1 = run_generators(sd, cfg)
        s2 = run_traces(sd, vsd, cfg, vcfg, os.path.join(OUT, "gcmvn_fr-en.npz"))
    with open(os.path.join(OUT, "SUMMARY_agent.json"), "w") as f:
        json.dump({"generators": s1, "traces": s2, "torch": torch.__version__}, f, indent=1)


if __name__ == "__main__":
    main()
