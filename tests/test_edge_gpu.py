"""Edge cases of the hot path on the GPU: empty / minimal / maximal / ragged inputs and the error
conventions of the C ABI (codes -> StreamSpeechHipError), as the reference's call sites exercise them."""
import ctypes as C

import numpy as np
import pytest
import torch

from streamspeech_amd import lib as L
from streamspeech_amd import synth

pytestmark = pytest.mark.gpu


def test_too_short_audio_gives_zero_frames_and_agent_reads(hip_model, hip_vocoder, synth_weights):
    from streamspeech_amd.agent import StreamSpeechS2STAgent
    from streamspeech_amd.modules import CodeHiFiGANVocoderWithDur, StreamSpeechModel
    from streamspeech_amd.simuleval_shim import SpeechSegment
    from tests.test_agent_cpu import make_args
    assert hip_model.lib.ss_fbank_num_frames(399) == 0 and hip_model.lib.ss_fbank_num_frames(400) == 1
    assert hip_model.fbank_cmvn(torch.zeros(100, device=hip_model.device)).shape == (0, 80)

    class Surf:
        def __init__(self, hv):
            self.hip = hv
        __call__ = CodeHiFiGANVocoderWithDur.__call__

    agent = StreamSpeechS2STAgent(make_args(320), model=StreamSpeechModel.from_engine(hip_model), vocoder=Surf(hip_vocoder))
    seg = agent.pushpop(SpeechSegment(content=[0.0] * 200, sample_rate=16000, finished=False))
    assert seg.is_empty                                   # ReadAction: not even one fbank frame yet
    seg = agent.pushpop(SpeechSegment(content=[], sample_rate=16000, finished=True))
    assert seg.finished                                   # an utterance that never reached one frame still finishes


@pytest.mark.parametrize("T", [7, 9, 10, 11, 12, 13])
def test_minimal_encoder_lengths(hip_model, synth_weights, T):
    """T' = ((T-1)//2+1 -1)//2+1 down to a single frame (SURVEY lens formula)."""
    from oracle import streamspeech_oracle as O
    cfg, _, sd, _ = synth_weights
    fb = torch.from_numpy(synth.synth_fbank(3, T))
    got = hip_model.encoder_forward(fb.to(hip_model.device), 8, 8).cpu()
    ref = O.encoder_forward(O.SD(sd), fb, cfg, 8, 8)
    assert got.shape == ref.shape and got.shape[0] == hip_model.lib.ss_encoder_out_len(T)
    assert (got - ref).abs().max() < 5e-4
    toks = hip_model.ctc_greedy(0, got.to(hip_model.device))[0]
    assert toks == O.ctc_head(O.SD(sd), ref, "source_unigram", cfg)[0]


def test_capacity_errors_are_reported_not_overrun(hip_model):
    cap = 2048                                            # max_rel_pos of the fixture model
    T = 4 * (cap + 8)
    fb = torch.zeros((T, 80), device=hip_model.device)
    with pytest.raises(L.StreamSpeechHipError, match="capacity|CAPACITY"):
        hip_model.encoder_forward(fb)
    enc = hip_model.encoder_forward(torch.from_numpy(synth.synth_fbank(1, 60)).to(hip_model.device))
    with pytest.raises(L.StreamSpeechHipError):
        hip_model.mt_greedy(enc, [], 5000, 1)             # beyond max_target_positions


def test_longest_supported_utterance_offline_and_chunked(hip_model, synth_weights):
    """T' = 2048 = the positional-table capacity (82 s of audio), chunked and offline, vs the oracle on the
    last frames only (the oracle takes ~10 s for the full thing, so compare a property: chunked rows of the first
    chunks equal the same rows computed on a prefix)."""
    T = 4 * 2048 - 1
    fb = torch.from_numpy(synth.synth_fbank(5, T)).to(hip_model.device)
    full = hip_model.encoder_forward(fb, 8, 8)
    assert full.shape == (2048, 256) and torch.isfinite(full).all()
    pre = hip_model.encoder_forward(fb[:640].contiguous(), 8, 8)
    n_final = (640 // 32 - 1) * 8
    assert (full[:n_final] - pre[:n_final]).abs().max() < 5e-5      # chunk-causality: the future cannot reach them
    off = hip_model.encoder_forward(fb)
    assert torch.isfinite(off).all() and (off - full).abs().max() > 1e-3   # offline really sees the future


def test_single_unit_and_extreme_ragged_vocoder_batch(hip_vocoder, synth_weights):
    from oracle import streamspeech_oracle as O
    _, vcfg, _, vsd = synth_weights
    w1, d1 = hip_vocoder.forward(torch.tensor([7], dtype=torch.int32, device="cuda:0"), True)
    rw, rd = O.vocoder_forward(O.SD(vsd), [7], vcfg, True)
    assert d1.cpu().tolist() == rd.view(-1).tolist() and w1.numel() == rw.numel()
    assert float(torch.sqrt(torch.mean((w1.cpu() - rw) ** 2))) < 1e-3
    codes = [[5], [int(c) for c in synth.uniform(1, "edge/long", (400,), 0, 1000)], [9, 9]]
    wavs, dur, K = hip_vocoder.batch_forward(codes, True)
    for b, c in enumerate(codes):
        one, _ = hip_vocoder.forward(torch.tensor(c, dtype=torch.int32, device="cuda:0"), True)
        assert wavs[b].shape == one.shape
        assert float(torch.sqrt(torch.mean((wavs[b] - one) ** 2))) < 1e-5


def test_one_frame_utterances_inside_a_pack_at_winograd_scale(hip_vocoder, synth_weights):
    """Segments SHORTER than a Winograd pair's reach (1, 2, 3 frames: 5 / 10 / 15 rows at the 256-channel stage, where a dilation-5 pair is
    rows t and t + 5) inside a pack big enough for every stage to take its Winograd slab kernel (>= 32768 rows at the 256-channel stage):
    the masked pair rows, the zero padding at both utterance edges inside one block and the neighbours' rows must not leak -- each
    utterance against its own single-utterance forward (direct-form kernels) and the shortest ones against the CPU oracle."""
    from oracle import streamspeech_oracle as O
    from streamspeech_amd import lib as L
    lib = L.load()
    _, vcfg, _, vsd = synth_weights
    sizes = [1, 700, 2, 1400, 3, 900, 1, 1200, 5, 800, 1000, 37]
    codes = [[int(c) for c in synth.uniform(6, f"edge/w{i}", (k,), 0, 1000)] for i, k in enumerate(sizes)]
    durs = [[1 + (j % 3 == 2) for j in range(len(c))] for c in codes]
    assert sum(sum(d) for d in durs) * 5 >= 32768

    def launches(name):
        for c in range(lib.ss_prof_num_classes()):
            if lib.ss_prof_class_name(c).decode() == name:
                n = C.c_int64()
                lib.ss_prof_totals(c, None, None, C.byref(n))
                return n.value
        raise KeyError(name)

    before = {k: launches(k) for k in ("conv_c256w<256,128>", "conv_c128w<256,128>", "conv_c64w<256,64>")}
    wavs, dur, K = hip_vocoder.batch_forward(codes, True, forced_dur=durs)
    for k, n0 in before.items():
        assert launches(k) > n0, f"{k} must have taken the pack's ResBlock convs"
    for b, c in enumerate(codes):
        fd = torch.tensor(durs[b], dtype=torch.int32, device="cuda:0")
        one, _ = hip_vocoder.forward(torch.tensor(c, dtype=torch.int32, device="cuda:0"), True, forced_dur=fd)
        assert wavs[b].shape == one.shape == (320 * sum(durs[b]),)
        assert torch.isfinite(wavs[b]).all()
        rms = float(torch.sqrt(torch.mean((wavs[b] - one) ** 2)))
        assert rms < 1e-5, f"utterance {b} ({len(c)} units): packed vs alone rms {rms}"
        if len(c) <= 5:
            rw, _ = O.vocoder_forward(vsd, c, vcfg, True, forced_dur=durs[b])
            assert float(torch.sqrt(torch.mean((wavs[b].cpu() - rw) ** 2))) < 1e-3


def test_ragged_batch_one_second_next_to_fifteen(hip_model):
    lens = [16000 * 15, 16000 * 1, 410, 16000 * 7]
    pcm = [torch.from_numpy(synth.synth_pcm(20 + i, n)).to(hip_model.device) for i, n in enumerate(lens)]
    feat, T = hip_model.batch_fbank_cmvn(torch.cat(pcm), lens)
    enc, Tp = hip_model.batch_encoder_forward(feat, T)
    off = 0
    for b, p in enumerate(pcm):
        one = hip_model.encoder_forward(hip_model.fbank_cmvn(p))
        assert one.shape[0] == Tp[b]
        assert (enc[off:off + Tp[b]] - one).abs().max() < 5e-5
        off += Tp[b]


def test_missing_weight_slot_is_an_error(synth_weights):
    from streamspeech_amd.config import ModelConfig
    from streamspeech_amd.engine import HipModel
    cfg, _, sd, _ = synth_weights
    bad = {k: v for k, v in sd.items() if not k.startswith("encoder.conformer_layers.3.ffn1.w_1")}
    with pytest.raises((L.StreamSpeechHipError, KeyError)):
        HipModel(bad, ModelConfig())


@pytest.mark.parametrize("persistent", [0, 64])
def test_nan_encoder_states_and_bad_ids_never_index_out_of_the_tables(hip_model, persistent):
    """Broken inputs must not become out-of-range device indices: an encoder output full of NaN turns every logit into NaN ->
    -inf (agent/sequence_generator.py:350), so each step returns the first unmasked column like an all -inf row; the chained
    token stays inside the dictionary in both forms of the decode step.  Ids from the host are range-checked like
    nn.Embedding does."""
    m = hip_model.new_context()
    m.set_persistent_mt_step(persistent)
    enc = torch.full((20, m.cfg.enc_dim), float("nan"), device=m.device)
    toks, feats = m.mt_greedy(enc, [11], 9, 1)
    assert len(toks) == 9 and toks[-1] == m.cfg.eos and all(0 <= t < m.cfg.tgt_vocab for t in toks)
    assert toks[:-1] == [0] * 8                             # first unmasked column
    good = hip_model.encoder_forward(torch.from_numpy(synth.synth_fbank(2, 80)).to(m.device))
    with pytest.raises(L.StreamSpeechHipError):
        m.mt_greedy(good, [m.cfg.tgt_vocab], 6, 1)
    m.mt_begin(good)
    with pytest.raises(IndexError):
        m.mt_append([m.cfg.eos, -3], 0, False, False)
    assert m.lib.ss_mt_get_persistent(m.h) == persistent    # NaNs are not time-outs: no fall-back was taken


def test_unit_ids_outside_the_vocoder_codebook(hip_vocoder):
    """Host ids are range-checked like the reference's nn.Embedding (codehifigan.py:56-70); ids already on the device cannot be
    checked without a sync, so the gather kernel itself never reads outside the table (they take row 0)."""
    n = hip_vocoder.cfg.num_embeddings
    with pytest.raises(IndexError):
        hip_vocoder.forward([5, n, 7])
    with pytest.raises(IndexError):
        hip_vocoder.batch_forward([[1, 2], [3, -1]])
    ok, _ = hip_vocoder.forward([0, 7, 9], forced_dur=[1, 2, 1])
    dev_ids = torch.tensor([n + 12345, 7, 9], dtype=torch.int32, device=hip_vocoder.device)
    got, _ = hip_vocoder.forward(dev_ids, forced_dur=[1, 2, 1])
    assert torch.equal(got, ok)


def test_empty_members_of_a_batch_are_refused(hip_model, hip_vocoder):
    """The ragged-batch calls take per-utterance lengths from the host: a zero-length member (no encoder rows, no decoder states,
    no units) is an argument error, like the single-utterance calls (Tp <= 0, K <= 0), not a zero-row segment for the kernels."""
    enc = hip_model.encoder_forward(torch.from_numpy(synth.synth_fbank(5, 90)).to(hip_model.device))
    with pytest.raises(L.StreamSpeechHipError):
        hip_model.batch_mt_greedy(enc, [enc.shape[0], 0], [4, 4])
    feats = torch.zeros((2, 6, hip_model.cfg.dec_dim), device=hip_model.device)
    with pytest.raises(L.StreamSpeechHipError):
        hip_model.batch_t2u_units(feats, [3, 0])
    with pytest.raises(L.StreamSpeechHipError):
        hip_vocoder.batch_forward([[1, 2, 3], []], forced_dur=[[1, 1, 1], []])


@pytest.mark.gpu
def test_log_softmax_kernel_and_lprobs_glue(hip_model):
    """no model math in torch on a product path -- `lprobs` / get_normalized_probs go through ss_log_softmax.
    Checked against torch's float64 log_softmax; pad / unk are -inf AFTER the normalisation (agent/ctc_decoder.py:52-60)."""
    import torch
    from streamspeech_amd.generators import CTCDecoder
    from streamspeech_amd.modules import StreamSpeechModel
    from tests import ref_fixtures as RF
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(37, 6000, generator=g) * 4).cuda()
    ref = torch.log_softmax(x.double().cpu(), -1)
    got = hip_model.normalized_probs(x, True, 1, 3).cpu()
    assert torch.isinf(got[:, 1]).all() and torch.isinf(got[:, 3]).all() and (got[:, 1] < 0).all()
    keep = [i for i in range(6000) if i not in (1, 3)]
    assert (got[:, keep].double() - ref[:, keep]).abs().max() < 1e-5          # f32 on values down to -30: 1 ulp = 2e-6
    p = hip_model.normalized_probs(x, False).cpu()
    assert (p.double() - ref.exp()).abs().max() < 1e-6 and (p.sum(-1) - 1).abs().max() < 1e-5
    # the agent-facing glue: CTCDecoder.generate(..., want_lprobs=True) and model.get_normalized_probs
    cfg = hip_model.cfg
    enc = hip_model.encoder_forward(torch.randn(83, 80, generator=g).cuda())
    d = RF.dictionaries(cfg)["source_unigram"]
    hyp = CTCDecoder(d, hip_model, 0).generate({"encoder_out": [enc[:, None]]}, aux_task_name="source_unigram", want_lprobs=True)[0][0]
    _, _, _, logits = hip_model.ctc_greedy(0, enc, want_logits=True)
    want = torch.log_softmax(logits.double().cpu(), -1)
    want[:, [d.pad(), d.unk()]] = float("-inf")
    lp = hyp["lprobs"][0].cpu().double()
    fin = torch.isfinite(want)
    assert torch.equal(torch.isfinite(lp), fin) and (lp[fin] - want[fin]).abs().max() < 1e-5
    m = StreamSpeechModel.from_engine(hip_model)
    assert (m.get_normalized_probs([logits], True).cpu().double() - torch.log_softmax(logits.double().cpu(), -1)).abs().max() < 1e-5
