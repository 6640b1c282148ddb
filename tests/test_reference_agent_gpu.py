"""GPU twins of tests/test_reference_agent_cpu.py: the repo's generators and agents over the HIP engine against
fixtures produced by the REFERENCE'S OWN generator classes and SimulEval agents (oracle/make_golden_agent.py)."""
import os

import numpy as np
import pytest
import torch

from tests import ref_fixtures as RF

pytestmark = pytest.mark.gpu

WAV_RMS_TOL = 1e-3


class HipVocSurface:
    """CodeHiFiGANVocoderWithDur call surface over the shared fixture handle."""

    def __init__(self, hv):
        self.hip = hv

    def __call__(self, x, dur_prediction=False):
        from streamspeech_amd.modules import CodeHiFiGANVocoderWithDur
        return CodeHiFiGANVocoderWithDur.__call__(self, x, dur_prediction)


def test_ctc_generators_match_reference_classes(hip_model, golden_dir, synth_weights):
    """a8 / a13 on HIP: CTCDecoder.generate and CTCSequenceGenerator.generate of the reference."""
    from streamspeech_amd.generators import CTCDecoder, CTCSequenceGenerator
    cfg = synth_weights[0]
    g, ge = RF.generators_gold(), np.load(os.path.join(golden_dir, "encoder.npz"))
    gd = np.load(os.path.join(golden_dir, "decoders.npz"))
    d = RF.dictionaries(cfg)
    for tag in ("offline", "c8"):
        enc = {"encoder_out": [torch.from_numpy(ge[f"enc_{tag}"]).cuda()[:, None]]}
        for head, hid in (("source_unigram", 0), ("ctc_target_unigram", 1)):
            hyp = CTCDecoder(d[head], hip_model, hid).generate(enc, aux_task_name=head)[0][0]
            assert hyp["tokens"].tolist() == g[f"ctc/{head}_{tag}_tokens"].tolist()
            assert list(hyp["index"]) == g[f"ctc/{head}_{tag}_index"].tolist()
            assert hyp["org_tokens"].tolist() == g[f"ctc/{head}_{tag}_org"].tolist()
    enc = {"encoder_out": [torch.from_numpy(ge["enc_offline"]).cuda()[:, None]]}
    hyp = CTCDecoder(d["source_unigram"], hip_model, 0).generate(enc, prefix=torch.from_numpy(g["ctc/prefix_in"]).long(),
                                                                 aux_task_name="source_unigram")[0][0]
    assert hyp["tokens"].tolist() == g["ctc/prefix_tokens"].tolist() and list(hyp["index"]) == g["ctc/prefix_index"].tolist()
    hyp = CTCSequenceGenerator(d["tgt"], hip_model).generate(torch.from_numpy(gd["mt_features"]).cuda())[0][0]
    assert hyp["tokens"].tolist() == g["unit/tokens"].tolist() and hyp["org_tokens"].tolist() == g["unit/org"].tolist()


@pytest.mark.parametrize("api", ["ss_mt_greedy", "ss_mt_append"])
def test_sequence_generator_matches_reference_class(hip_model, golden_dir, synth_weights, api):
    """a9 on HIP: every generate_decoder case of the reference class, through the one-call search and through the
    step API (the generator falls back to it when the engine has no mt_greedy)."""
    from streamspeech_amd.engine import HipModel
    from tests.test_reference_agent_cpu import mt_cases, run_mt_case
    cfg = synth_weights[0]
    g, ge = RF.generators_gold(), np.load(os.path.join(golden_dir, "encoder.npz"))
    enc = torch.from_numpy(ge["enc_offline"]).cuda()
    made = {}

    class StepOnly:                      # the same handle without the one-call search
        def __init__(self, m):
            self._m = m

        def __getattr__(self, k):
            if k == "mt_greedy":
                raise AttributeError(k)
            return getattr(self._m, k)

    def factory(c):
        if c.eos not in made:            # another stop token = another ss_config over the SAME packed weights
            made[c.eos] = hip_model if c.eos == cfg.eos else HipModel(None, c, device=str(hip_model.device), _share=hip_model._packed)
        return made[c.eos] if api == "ss_mt_greedy" else StepOnly(made[c.eos])

    for name, args, want in mt_cases(g):
        got = run_mt_case(factory, enc, cfg, args)
        assert got == want, (api, name, got, want)


@pytest.mark.parametrize("incremental", [True, False])
@pytest.mark.parametrize("name", ["s2st_320_a", "s2st_320_b", "s2st_320_k3", "s2st_640_a", "s2st_640_b", "s2st_960_a",
                                  "s2st_320_48k"])
def test_s2st_agent_trace_matches_reference_agent(hip_model, hip_vocoder, synth_weights, name, incremental):
    """a16 on HIP: READ/WRITE trace, per-call sample counts and waveform of the reference agent's policy() --
    with the incremental encoder + receptive-field vocoder tail (the default) and with the reference's full
    recompute.  640 / 960 ms: whole-word mode with non-final writes; 48k: resampling front-end."""
    from streamspeech_amd.agent import StreamSpeechS2STAgent
    from streamspeech_amd.modules import StreamSpeechModel
    cfg = synth_weights[0]
    g, cases = RF.traces_gold()
    case = cases[name]
    over = dict(case["over"])
    if not incremental:
        over.update(full_recompute_encoder=True, vocoder_context_units=0)
    args = RF.agent_args(StreamSpeechS2STAgent, case["segment_ms"], case["sr"], over)
    agent = RF.set_dicts(StreamSpeechS2STAgent(args, model=StreamSpeechModel.from_engine(hip_model),
                                               vocoder=HipVocSurface(hip_vocoder)), cfg)
    RF.check_s2st_trace(g, name, RF.run_case(agent, case), WAV_RMS_TOL)
    hip_model.encoder_stream_set_tail(0)


@pytest.mark.parametrize("name", ["s2tt_320_a", "s2tt_640_a", "asr_320_a"])
def test_text_agent_trace_matches_reference_agent(hip_model, synth_weights, name):
    """f2 on HIP: text increments of the reference S2TT / ASR agents."""
    from streamspeech_amd.agent_text import StreamSpeechASRAgent, StreamSpeechS2TTAgent
    from streamspeech_amd.modules import StreamSpeechModel
    cfg = synth_weights[0]
    g, cases = RF.traces_gold()
    case = cases[name]
    cls = StreamSpeechS2TTAgent if case["kind"] == "s2tt" else StreamSpeechASRAgent
    agent = RF.set_dicts(cls(RF.agent_args(cls, case["segment_ms"], case["sr"], case["over"]),
                             model=StreamSpeechModel.from_engine(hip_model)), cfg)
    RF.check_text_trace(g, name, RF.run_case(agent, case))
