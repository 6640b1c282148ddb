"""The persistent MT decode step (csrc/mt_step.hip, ss_mt_set_persistent) against the launch-per-op step it replaces:
identical greedy token ids, decoder states within float noise, on prefixes / forced eos / min_len, alone and with several
contexts decoding at once next to a full-chip vocoder batch (the exchange between workgroups must hold under uneven load:
cdna_hip_programming.md Guideline 16 "test every hand-off under uneven load"); every bounded wait must have been met."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _search(model, enc, prefix, max_new, min_len=1):
    max_len = len(prefix) + max_new
    toks, feats = model.mt_greedy(enc, prefix, max_len, min_len)
    return toks, feats.clone()


@pytest.mark.parametrize("wgs", [64, 128, 256])
def test_persistent_step_equals_launch_per_op_step(hip_model, wgs):
    from streamspeech_amd import synth
    lib = hip_model.lib
    err0 = lib.ss_debug_sk_errors()
    cases = [(131, [], 12, 1), (435, [17, 4021, 99], 9, 1), (57, [], 30, 5), (23, [5999], 1, 1)]
    encs = [hip_model.encoder_forward(torch.from_numpy(synth.synth_fbank(60 + i, T)).cuda()) for i, (T, _, _, _) in enumerate(cases)]
    try:
        for (T, prefix, max_new, min_len), enc in zip(cases, encs):
            hip_model.set_persistent_mt_step(0)
            ref_t, ref_f = _search(hip_model, enc, prefix, max_new, min_len)
            hip_model.set_persistent_mt_step(wgs)
            got_t, got_f = _search(hip_model, enc, prefix, max_new, min_len)
            assert got_t == ref_t, (T, prefix, got_t, ref_t)
            assert got_f.shape == ref_f.shape and (got_f - ref_f).abs().max().item() < 5e-5
            # second run of the persistent form: bit-reproducible (fixed summation orders, epoch-tagged exchange)
            again_t, again_f = _search(hip_model, enc, prefix, max_new, min_len)
            assert again_t == got_t and torch.equal(again_f, got_f)
    finally:
        hip_model.set_persistent_mt_step(0)
    assert lib.ss_debug_sk_errors() == err0, "a bounded wait of the persistent step timed out"


def test_persistent_steps_of_several_contexts_under_load(hip_model, hip_vocoder):
    """Four contexts decode concurrently with the persistent step (64 workgroups each) while a fifth stream runs batch-scale
    vocoder passes (full-chip stream-K kernels): tokens equal the serial launch-per-op result, no time-out."""
    from streamspeech_amd import synth
    lib = hip_model.lib
    err0 = lib.ss_debug_sk_errors()
    dev = hip_model.device
    Ts = [131, 260, 77, 401]
    fbs = [torch.from_numpy(synth.synth_fbank(80 + i, T)).to(dev) for i, T in enumerate(Ts)]
    ctxs = [hip_model.new_context() for _ in Ts]
    serial = []
    for m, fb in zip(ctxs, fbs):
        enc = m.encoder_forward(fb)
        serial.append((_search(m, enc, [], 24)[0], enc))
    codes = [[int(c) for c in synth.uniform(21, f"mtp/{i}", (150,), 0, 1000)] for i in range(24)]
    durs = [[1 + (j % 3 == 1) for j in range(150)] for _ in range(24)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(len(Ts) + 1)]
    results, errors = [None] * len(Ts), []
    stop = threading.Event()
    bar = threading.Barrier(len(Ts) + 1)

    def decoder(i):
        try:
            m = ctxs[i]
            m.set_persistent_mt_step(64)
            with torch.cuda.stream(streams[i]):
                bar.wait()
                out = []
                for _ in range(6):
                    out.append(_search(m, serial[i][1], [], 24)[0])
                streams[i].synchronize()
            results[i] = out
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            bar.abort()
        finally:
            ctxs[i].set_persistent_mt_step(0)

    def load():
        try:
            with torch.cuda.stream(streams[-1]):
                bar.wait()
                while not stop.is_set():
                    hip_vocoder.batch_forward(codes, True, forced_dur=durs)
                    streams[-1].synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=decoder, args=(i,)) for i in range(len(Ts))] + [threading.Thread(target=load)]
    for t in th:
        t.start()
    for t in th[:-1]:
        t.join()
    stop.set()
    th[-1].join()
    torch.cuda.synchronize()
    if errors:
        raise errors[0]
    for i in range(len(Ts)):
        for out in results[i]:
            assert out == serial[i][0], f"context {i}"
    assert lib.ss_debug_sk_errors() == err0, "a bounded wait timed out under load"


def test_timed_out_persistent_step_falls_back_to_launch_per_op(hip_model):
    """A launch whose bounded waits timed out publishes -1 instead of a token; ss_mt_greedy (one C call) and the engine's
    mt_append (the step-by-step form of generators.py) then repeat with one launch per op, loudly, and the context stays on that
    form.  The time-out is injected (ss_debug_mt_inject_timeout); tokens and states must equal the launch-per-op search."""
    import warnings
    from streamspeech_amd import synth
    lib = hip_model.lib
    err0 = lib.ss_debug_sk_errors()
    m = hip_model.new_context()
    enc = m.encoder_forward(torch.from_numpy(synth.synth_fbank(91, 211)).cuda())
    ref_t, ref_f = _search(m, enc, [7, 4242], 14)
    # (1) the search in one C call
    m.set_persistent_mt_step(64)
    assert lib.ss_mt_get_persistent(m.h) == 64
    assert lib.ss_debug_mt_inject_timeout(m.h) == 0
    got_t, got_f = _search(m, enc, [7, 4242], 14)
    assert got_t == ref_t and torch.equal(got_f, ref_f)
    assert lib.ss_mt_get_persistent(m.h) == 0 and m.persistent_mt == 0
    # (2) step by step through mt_append, time-out on the third generated token
    m.set_persistent_mt_step(64)
    m.mt_begin(enc)
    feats, nxt = m.mt_append([m.cfg.eos, 7, 4242], 0, False, False)
    toks, rows = [nxt], [feats]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        step = 3                                             # the loop of generators.py (start = 2 prefix tokens, max_len = 16)
        while nxt != m.cfg.eos and step <= 16:
            if step == 5:
                lib.ss_debug_mt_inject_timeout(m.h)
            feats, nxt = m.mt_append([toks[-1]], step, False, step >= 16)
            rows.append(feats)
            toks.append(nxt)
            step += 1
    assert any("persistent MT decode step timed out" in str(x.message) for x in w)
    assert lib.ss_mt_get_persistent(m.h) == 0
    assert toks == ref_t
    got = torch.cat(rows)
    assert got.shape == ref_f.shape and (got - ref_f).abs().max().item() < 5e-5
    assert lib.ss_debug_sk_errors() == err0            # the injection does not touch the real counter


def test_single_persistent_step_computes_a_fed_eos_like_the_launch_per_op_form(hip_model):
    """the early exit on a fed </s> belongs to ss_mt_greedy's search loop only; ss_mt_append with the persistent step on must
    compute the step it is fed (cache row, state row, next token) exactly as its launch-per-op form does."""
    from streamspeech_amd import synth
    m = hip_model.new_context()
    enc = m.encoder_forward(torch.from_numpy(synth.synth_fbank(92, 171)).cuda())
    out = {}
    for wg in (0, 64):
        m.set_persistent_mt_step(wg)
        m.mt_begin(enc)
        m.mt_append([m.cfg.eos, 7, 4242], 0, False, False)
        feats, nxt = m.mt_append([m.cfg.eos], 3, False, False)          # </s> fed at position 3
        feats2, nxt2 = m.mt_append([11], 4, False, False)               # and the search goes on over that cache row
        out[wg] = (feats.cpu(), nxt, feats2.cpu(), nxt2)
    m.set_persistent_mt_step(0)
    assert out[64][1] == out[0][1] and out[64][3] == out[0][3]
    assert torch.isfinite(out[64][0]).all() and (out[64][0] - out[0][0]).abs().max().item() < 5e-5
    assert (out[64][2] - out[0][2]).abs().max().item() < 5e-5
