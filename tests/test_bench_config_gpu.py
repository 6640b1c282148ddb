"""Parity at the configuration that is benchmarked.

bench.py's timed step is streamspeech_amd/workload.py::run_batch on the ragged batches of
workload.bench_plan (64 CVSS-C-shaped utterances per batch, length-bucketed, natural kernel dispatch -- no
forced tiles), 8 batches in flight on 8 HIP streams / contexts.  Here exactly that runs, and every utterance
of the longest (up to 15 s), the shortest (1 s) and a middle batch is checked against the CPU oracle:
identical ASR / ST ids and frame indices, identical MT ids, identical raw unit argmax at every one of the
U = 25 (N+1) positions and identical collapsed units, durations as forced, waveform RMS <= 1e-3
(reference: fairseq/models/text_to_speech/hifigan.py:154-170, ctc_transformer_unit_decoder.py:153-260).
The oracle is fed the HIP fbank (north star: 'on the same fbank input'); the fbank itself is checked
against the oracle's Kaldi restatement; rows whose arg-max the ORACLE's own float32 logits leave undecided (top-1 / top-2
gap < 5e-5, at most 3 of ~100 k rows) are the one stated exception, see NEAR_TIE below (log-mel values: RMS 1e-4 (observed 4e-6), max abs 2e-2
and at most 1e-5 of the values past 1e-3 -- float32 cancellation in near-empty bins of the noise input, where d(log e) =
de / e, reaches 7.0e-3 in single values of the 7.7 M compared here)."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

WAV_RMS_TOL = 1e-3
# Arg-max rows the float32 arithmetic does not decide.  The HIP path and the CPU oracle sum the same float32 products in
# different orders, so their logits differ by up to ~2e-5 (measured: tests/test_margin_gpu.py; the oracle's own logits move
# by as much between CPU thread counts).  With seeded RANDOM weights the 6000-way / 1005-way CTC rows include near ties
# that no trained model has: ~3e-5 of the rows have a top-1 / top-2 gap below that difference, and this test compares
# ~100 k rows.  Such a row may pick the oracle's SECOND choice -- only if the oracle's own gap there is below NEAR_TIE --
# and at most MAX_NEAR_TIES rows in the whole test may do so (every one is printed).  Everything else stays bit-identical;
# MT ids (gaps >= 0.1) are compared strictly.
NEAR_TIE = 5e-5
MAX_NEAR_TIES = 3


def _argmax_rows(tag, stage, hip_raw, ref_raw, ref_logits, masked, near):
    """Raw arg-max ids identical, except near ties of the oracle's own logits (see NEAR_TIE)."""
    hip_raw, ref_raw = list(hip_raw), list(ref_raw)
    assert len(hip_raw) == len(ref_raw), f"{tag}: {stage} row count"
    if hip_raw == ref_raw:
        return
    x = torch.as_tensor(np.asarray(ref_logits)).double().clone()
    x[:, masked] = float("-inf")
    for t, (a, b) in enumerate(zip(hip_raw, ref_raw)):
        if a == b:
            continue
        top = torch.topk(x[t], 2)
        gap = float(top.values[0] - top.values[1])
        assert int(top.indices[0]) == b and int(top.indices[1]) == a and gap < NEAR_TIE, \
            f"{tag}: {stage} row {t}: HIP {a}, oracle {b} (oracle top-2 {top.indices.tolist()}, gap {gap:.3e})"
        near.append(f"{tag}: {stage} row {t}: HIP {a} / oracle {b}, oracle gap {gap:.2e}")


def _oracle_utterance(O, osd, ovsd, cfg, vcfg, fb, u, workload, hip_unit_raw=None):
    enc = O.encoder_forward(osd, fb, cfg)
    asr = O.ctc_head(osd, enc, "source_unigram", cfg)
    st = O.ctc_head(osd, enc, "ctc_target_unigram", cfg)
    toks = O.mt_greedy(osd, enc, cfg, max_new_tokens=u.n_mt)
    body = toks[:-1] if toks and toks[-1] == cfg.eos else toks
    feats = O.mt_decoder_features(osd, [cfg.eos] + body, enc, cfg)
    logits = O.unit_decoder_logits(osd, O.t2u_encoder(osd, feats, cfg), cfg)
    unit_toks, raw = O.unit_ctc_generate(logits, cfg)       # unit ids (0..999) and raw argmax over the unit vocabulary
    if hip_unit_raw is not None and list(hip_unit_raw) != list(raw):     # only near ties get past _argmax_rows (checked by the caller)
        toks_h, _ = O.ctc_collapse(list(hip_unit_raw), cfg.unit_blank, cfg.pad)
        toks_h = toks_h[:-1] if toks_h and toks_h[-1] == cfg.eos else toks_h
        unit_toks = [t - 4 for t in toks_h if t not in (0, cfg.eos)]
    codes = workload.resize_units(unit_toks, u.n_units, u.idx)
    wav, dur = O.vocoder_forward(ovsd, codes, vcfg, True, forced_dur=u.durations)
    return {"asr": asr, "st": st, "mt": toks, "raw": raw, "unit_logits": logits, "codes": codes, "wav": wav, "dur": dur}


def test_timed_step_matches_oracle_on_8_streams(hip_model, hip_vocoder, synth_weights):
    from oracle import kaldi_fbank as K
    from oracle import streamspeech_oracle as O
    from streamspeech_amd import synth, workload
    cfg, vcfg, sd, vsd = synth_weights
    osd, ovsd = O.SD(sd), O.SD(vsd)
    dev = hip_model.device
    mine, groups = workload.bench_plan(16, 64)            # the default bench: 16 steps x 64 utterances
    assert len(groups) == 16 and all(len(g) == 64 for g in groups)
    secs = [[mine[i].seconds for i in g] for g in groups]
    assert max(secs[0]) == 15.0 and min(secs[-1]) < 1.2    # both ends of the length distribution (clipped to [1, 15] s) are in
    checked = [0, len(groups) // 2, len(groups) - 1]
    others = [2, 5, 9, 11, 13]                            # in flight at the same time (not oracle-checked)
    sel = checked + others
    S = len(sel)
    assert S == 8
    # model.hip's CMVN for this test = identity, as in bench.py (HipModel(sd, cfg) without statistics)
    from streamspeech_amd.engine import HipModel
    model0 = HipModel(sd, cfg, device=str(dev))
    ctxs = [(model0, hip_vocoder)] + [(model0.new_context(), hip_vocoder.new_context()) for _ in range(S - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    packs = [torch.cat([torch.from_numpy(synth.synth_pcm(1234 + mine[i].idx, mine[i].n_samples)) for i in groups[g]]).to(dev)
             for g in sel]
    torch.cuda.synchronize()
    results, errors = [None] * S, []
    bar = threading.Barrier(S)

    def worker(wi):
        try:
            m, v = ctxs[wi]
            utts = [mine[i] for i in groups[sel[wi]]]
            with torch.cuda.stream(streams[wi]):
                bar.wait()
                results[wi] = workload.run_batch(m, v, packs[wi], utts, detail=True)
                streams[wi].synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            try:
                bar.abort()
            except Exception:  # noqa: BLE001
                pass

    th = [threading.Thread(target=worker, args=(i,)) for i in range(S)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    if errors:
        raise errors[0]
    assert int(hip_model.lib.ss_debug_sk_errors()) == 0, "stream-K bounded wait timed out"

    torch.set_num_threads(min(32, torch.get_num_threads()))
    worst = {"fbank": 0.0, "rms": 0.0}
    near = []
    fb_sq, fb_n, fb_far = 0.0, 0, 0
    n_units_total = n_pos_total = 0
    with torch.inference_mode():
        for wi in range(len(checked)):
            utts = [mine[i] for i in groups[sel[wi]]]
            r = results[wi]
            fb_all = r["fbank"].cpu()
            off_f = off_s = 0
            for b, u in enumerate(utts):
                fb = fb_all[off_f:off_f + r["T"][b]].numpy()
                off_f += r["T"][b]
                pcm = synth.synth_pcm(1234 + u.idx, u.n_samples)
                off_s += u.n_samples
                ref_fb = K.fbank(pcm * np.float32(32768.0))
                assert ref_fb.shape == fb.shape
                worst["fbank"] = max(worst["fbank"], float(np.abs(ref_fb - fb).max()))
                fb_far += int((np.abs(ref_fb - fb) > 1e-3).sum())
                fb_sq += float(((ref_fb - fb).astype(np.float64) ** 2).sum())
                fb_n += fb.size
                ref = _oracle_utterance(O, osd, ovsd, cfg, vcfg, fb, u, workload, hip_unit_raw=r["unit_raw"][b])
                tag = f"batch {sel[wi]} utt {u.idx} ({u.seconds:.2f} s)"
                for head, name in (("asr", "ASR"), ("st", "ST")):
                    _argmax_rows(tag, name + " CTC", r[head][b][2], ref[head][2], ref[head][3], [cfg.pad, cfg.unk], near)
                    ids, index = O.ctc_collapse(list(r[head][b][2]), 0, cfg.pad)     # the collapse itself, on the rows as decided
                    assert list(r[head][b][0]) == ids and list(r[head][b][1]) == index, f"{tag}: {name} ids / frame index"
                    if list(r[head][b][2]) == list(ref[head][2]):
                        assert list(r[head][b][0]) == list(ref[head][0]) and list(r[head][b][1]) == list(ref[head][1]), tag
                assert r["mt"][b] == ref["mt"], tag + ": MT ids"
                _argmax_rows(tag, "unit CTC", r["unit_raw"][b], ref["raw"], ref["unit_logits"], [cfg.pad, cfg.unk], near)
                assert r["codes"][b] == ref["codes"], tag + ": units fed to the vocoder"
                n_pos_total += len(ref["raw"])
                n_units_total += len(ref["codes"])
                wav = r["wavs"][b].cpu()
                assert wav.numel() == ref["wav"].numel() == 320 * sum(u.durations), tag
                rms = float(torch.sqrt(torch.mean((wav - ref["wav"]) ** 2)))
                worst["rms"] = max(worst["rms"], rms)
                assert rms < WAV_RMS_TOL, f"{tag}: waveform rms {rms}"
            durs = r["dur"].cpu().tolist()
            assert durs == [d for u in utts for d in u.durations]
    for line in near:
        print("near tie (oracle gap < %.0e): %s" % (NEAR_TIE, line))
    assert len(near) <= MAX_NEAR_TIES, near
    fb_rms = (fb_sq / fb_n) ** 0.5
    assert worst["fbank"] < 2e-2 and fb_rms < 1e-4 and fb_far <= 1e-5 * fb_n, (worst, fb_rms, fb_far, fb_n)
    if worst["rms"] >= 1e-4:     # observed ~1e-6: report a regression that the north-star bar (1e-3) would let through
        import warnings
        warnings.warn(f"bench-config waveform RMS {worst['rms']:.2e} is past the tight bar 1e-4 (north-star bar 1e-3 still met)")
    print(f"bench-config parity: {64 * len(checked)} utterances, {n_pos_total} unit positions, {n_units_total} vocoder units, "
          f"{len(near)} near-tie rows, worst fbank err {worst['fbank']:.2e} (rms {fb_rms:.2e}, {fb_far} of {fb_n} values past 1e-3), worst wav rms {worst['rms']:.2e}")
