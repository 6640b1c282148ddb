"""Persistent layer launches of the incremental streaming encoder (csrc/enc_step.hip): on a scratch set whose
persistent forms are on, a streaming call with <= 48 rows to compute runs every Conformer layer as two persistent launches around
the attention kernel instead of eleven launches.  Must stay equal to the full recompute the reference does per chunk
(agent/speech_to_speech.streamspeech.agent.py:425-435) and to the launch-per-op form of the same call; rows reported final never
change; a call with more rows (the first call of a long prefix) takes the launch-per-op form."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from streamspeech_amd import synth
    from streamspeech_amd.config import ModelConfig
    from streamspeech_amd.engine import HipModel
    cfg = ModelConfig()
    return HipModel(synth.make_model_state_dict(0, cfg), cfg)


def _drive(m, fb_all, ac, cc, Ts):
    m.encoder_stream_reset()
    outs, finals = [], []
    for T in Ts:
        outs.append(m.encoder_stream_forward(fb_all[:T].contiguous(), ac, cc).clone())
        finals.append(m.stream_stats[0])
    m.encoder_stream_reset()
    return outs, finals


@pytest.mark.parametrize("ac,cc,step", [(8, 8, 32), (16, 16, 32), (8, 8, 45), (24, 16, 64), (16, 16, 128)])   # 8 ... 40 rows per call: the 1-, 2- and 3-row-tile kernels
def test_persistent_layer_launches_equal_launch_per_op_and_full_recompute(model, ac, cc, step):
    from streamspeech_amd import synth
    lib = model.lib
    fb_all = torch.from_numpy(synth.synth_fbank(43, 620)).to(model.device)
    Ts = list(range(40, 620, step)) + [620]
    model.set_persistent_mt_step(0)
    n0 = lib.ss_debug_enc_step_launches()
    ref, ref_final = _drive(model, fb_all, ac, cc, Ts)
    assert lib.ss_debug_enc_step_launches() == n0                          # launch per op
    model.set_persistent_mt_step(64)
    got, got_final = _drive(model, fb_all, ac, cc, Ts)
    used = lib.ss_debug_enc_step_launches() - n0
    assert used >= 2 * model.cfg.enc_layers * 3 and used % (2 * model.cfg.enc_layers) == 0, used   # two per layer and call with <= 48 rows to compute
    assert got_final == ref_final
    worst = 0.0
    for T, a, b in zip(Ts, got, ref):
        assert a.shape == b.shape
        worst = max(worst, (a - b).abs().max().item())
        full = model.encoder_forward(fb_all[:T].contiguous(), ac, cc)
        assert (a - full).abs().max().item() < 5e-5, T
    assert worst < 2e-5, worst
    for i in range(1, len(got)):                                           # final rows are served unchanged
        nf = got_final[i - 1]
        assert torch.equal(got[i][:nf], got[i - 1][:nf])
    assert model.ctc_greedy(0, got[-1])[0] == model.ctc_greedy(0, ref[-1])[0]
    assert lib.ss_debug_sk_errors() == 0


def test_calls_with_more_than_48_rows_take_the_launch_per_op_form(model):
    from streamspeech_amd import synth
    lib = model.lib
    model.set_persistent_mt_step(64)
    fb = torch.from_numpy(synth.synth_fbank(44, 400)).to(model.device)       # 100 encoder rows at once: nothing is final yet
    model.encoder_stream_reset()
    n0 = lib.ss_debug_enc_step_launches()
    out = model.encoder_stream_forward(fb, 8, 8)
    assert lib.ss_debug_enc_step_launches() == n0
    assert (out - model.encoder_forward(fb, 8, 8)).abs().max().item() < 5e-5
    model.encoder_stream_reset()


@pytest.mark.parametrize("persistent", [0, 64])
@pytest.mark.parametrize("ac,cc,step", [(8, 8, 32), (16, 16, 64), (24, 16, 150)])
def test_few_queries_attention_kernel_equals_the_tile_kernel_on_streaming_calls(model, persistent, ac, cc, step):
    """attention_relpos_q16_kernel (<= 48 query rows at offset q0 over ALL keys so far: one workgroup per 16-query tile, 64-key tile and
    head, key tiles merged by the last arrival) against the 64-query tile kernel on the same calls: prefixes up to 12 s, so 1 ... 5 key
    tiles, in the launch-per-op and the persistent form; repeated runs reproduce their bits (arrival counters back at zero)."""
    from streamspeech_amd import synth
    lib = model.lib
    fb_all = torch.from_numpy(synth.synth_fbank(47, 1200)).to(model.device)
    Ts = list(range(40, 1200, step)) + [1200]
    model.set_persistent_mt_step(persistent)
    try:
        lib.ss_debug_attention_q16(0)
        ref, ref_final = _drive(model, fb_all, ac, cc, Ts)
        lib.ss_debug_attention_q16(1)
        got, got_final = _drive(model, fb_all, ac, cc, Ts)
        again, _ = _drive(model, fb_all, ac, cc, Ts)
    finally:
        lib.ss_debug_attention_q16(1)
        model.set_persistent_mt_step(64)
    assert got_final == ref_final
    for a, b, c in zip(got, ref, again):
        assert (a - b).abs().max().item() < 2e-5
        assert torch.equal(a, c)
    full = model.encoder_forward(fb_all, ac, cc)
    assert (got[-1] - full).abs().max().item() < 5e-5
    assert lib.ss_debug_sk_errors() == 0


def test_deferred_time_out_check_and_the_repeat_protocol(model):
    """The agents' engine path: with ``ctc_speculate`` on a persistent context, encoder_stream_forward queues both CTC heads behind the
    layers and synchronises once (ss_encoder_stream_set_deferred / ss_encoder_stream_status).  Same encoder rows and CTC answers as the
    synchronous form; an injected time-out (ss_debug_enc_step_inject_timeout) makes the status call ask for a repeat, the repeated call
    runs one launch per op, the failed call's rows are not final, and the outputs are still those of the reference run."""
    from streamspeech_amd import synth
    lib = model.lib
    fb_all = torch.from_numpy(synth.synth_fbank(53, 700)).to(model.device)
    Ts = list(range(40, 700, 32)) + [700]
    model.set_persistent_mt_step(64)
    model.ctc_speculate = False
    ref, ref_final = _drive(model, fb_all, 8, 8, Ts)
    ref_ctc = [(model.ctc_greedy(0, o)[0], model.ctc_greedy(1, o)[0]) for o in ref]
    try:
        model.ctc_speculate = True
        model.encoder_stream_reset()
        n0 = lib.ss_debug_enc_step_launches()
        for i, T in enumerate(Ts):
            if i == 7:
                lib.ss_debug_enc_step_inject_timeout(model.h)
            out = model.encoder_stream_forward(fb_all[:T].contiguous(), 8, 8)
            assert model.stream_stats[0] == ref_final[i]
            assert (out - ref[i]).abs().max().item() < 2e-5
            if i != 7:
                assert model._ctc_both is not None or i > 7 or out.shape[0] == 0
            a0, a1 = model.ctc_greedy(0, out)[0], model.ctc_greedy(1, out)[0]
            assert (a0, a1) == ref_ctc[i], i
            if i == 6:
                used = lib.ss_debug_enc_step_launches() - n0
                assert used > 0                                          # the persistent form ran until the injected time-out
            if i == 7:
                used7 = lib.ss_debug_enc_step_launches()
        assert lib.ss_debug_enc_step_launches() == used7                 # after it: one launch per op
    finally:
        model.ctc_speculate = False
        model._ctc_both = None
        model.set_persistent_mt_step(64)                                 # re-arms the persistent layer launches
        model.encoder_stream_reset()
    n1 = lib.ss_debug_enc_step_launches()
    _drive(model, fb_all, 8, 8, Ts[:4])
    assert lib.ss_debug_enc_step_launches() > n1
    assert lib.ss_debug_sk_errors() == 0                                 # the injected time-out is not a time-out of the process
