"""is an arg-max of the packed-batch path a function of the utterance alone?

The reference decodes ONE utterance per call (agent/speech_to_speech.streamspeech.agent.py:425-478), so an id can never depend on
what an utterance is batched with.  Per arg-max stage (ASR CTC, ST CTC, MT greedy, unit CTC), for packs of 64 and 128 taken from the
bench plan's longest, a middle and the shortest length bucket:

* `alone` -- every utterance of the pack as a pack of ONE through the same ss_batch_* calls -- must give BIT-IDENTICAL dense
  logits / decoder states (max_logit_diff == 0.0): the ragged-batch arithmetic is pack-invariant by construction (one accumulator
  chain per GEMM output element whatever the row count, whole-tile fused FFN, fixed LayerNorm / attention / decode forms);
* `single` -- the single-utterance entry points the agents use (latency kernels: GEMV and small-M split-K forms, key-split
  attention, the persistent MT step) -- computes the same B = 1 arithmetic in ANOTHER float32 order, exactly as the CPU oracle does:
  logits within 5e-5, arg-max identical on every row whose top-1 / top-2 margin exceeds twice its own logit difference (a flip there
  would be a bug, not rounding); rows below that (seeded random weights make 6000-way margins of 1e-5 on a few of ~18 k rows; median
  margin 0.05-0.5) are counted and printed -- they are the rows tests/test_bench_config_gpu.py would adjudicate in float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _margins(logits, masked):
    x = logits.double().clone()
    x[:, masked] = float("-inf")
    top = torch.topk(x, 2, dim=1).values
    return (top[:, 0] - top[:, 1]), x.argmax(1)


def _stage(name, single, packed, masked, report):
    assert single.shape == packed.shape, name
    diff = (single.double() - packed.double()).abs().max(dim=1).values
    mg, am_s = _margins(single, masked)
    _, am_p = _margins(packed, masked)
    risk = mg < 2.0 * diff                       # float32 order alone could decide these rows either way
    report[name] = {"rows": int(single.shape[0]), "max_logit_diff_vs_single_entry_points": float(diff.max()),
                    "min_margin": float(mg.min()), "median_margin": float(mg.median()),
                    "rows_with_margin_below_2x_own_diff": int(risk.sum()), "of_them_flipped": int((am_s != am_p)[risk].sum()),
                    "rows_with_margin_below_100x_max_diff": int((mg < 100.0 * diff.max()).sum())}
    assert float(diff.max()) < 5e-5, f"{name}: {report[name]}"
    assert torch.equal(am_s[~risk], am_p[~risk]), f"{name}: arg-max differs on a row with a decidable margin: {report[name]}"
    assert int(risk.sum()) <= max(3, single.shape[0] // 2000), f"{name}: too many undecidable rows for rounding alone: {report[name]}"


def _pack_stages(m, utts, pcms):
    """The calls of workload.run_batch up to the unit logits."""
    feat, T = m.batch_fbank_cmvn(torch.cat(pcms), [u.n_samples for u in utts])
    enc, Tp = m.batch_encoder_forward(feat, T)
    m.batch_ctc_greedy(0, enc, Tp)
    asr = m.last_logits().cpu()
    m.batch_ctc_greedy(1, enc, Tp)
    st = m.last_logits().cpu()
    toks, feats, n = m.batch_mt_greedy(enc, Tp, [u.n_mt for u in utts])
    m.batch_t2u_units(feats, n)
    unit = m.last_logits().cpu()
    mt = torch.cat([feats[b, : utts[b].n_mt + 1].cpu() for b in range(len(utts))])
    return asr, st, mt, unit, [list(t) for t in toks]


@pytest.mark.parametrize("bucket", ["longest", "middle", "shortest"])
@pytest.mark.parametrize("pack", [64, 128])
def test_pack_vs_single_utterance_argmax_margins(hip_model, synth_weights, pack, bucket):
    from streamspeech_amd import synth, workload
    from streamspeech_amd.pipeline import mt_greedy
    cfg, vcfg, sd, vsd = synth_weights
    m = hip_model
    assert m.pack_invariant()
    order = sorted(workload.make_utterances(16 * pack), key=lambda u: -u.seconds)        # the bench plan's length-sorted set
    k = {"longest": 0, "middle": 8, "shortest": 15}[bucket]
    utts = order[k * pack:(k + 1) * pack]
    pcms = [torch.from_numpy(synth.synth_pcm(1234 + u.idx, u.n_samples)).cuda() for u in utts]
    emb = torch.from_numpy(np.asarray(sd["target_unigram_decoder.embed_tokens.weight"])).double()

    # ---- the ragged pack (the calls of workload.run_batch) ----
    p_asr, p_st, p_mt, p_unit, p_tok = _pack_stages(m, utts, pcms)

    # ---- every utterance as a pack of one: bit-identical, every stage ----
    alone = [_pack_stages(m, [u], [pcm]) for u, pcm in zip(utts, pcms)]
    bit = {}
    for i, name in enumerate(("asr_ctc", "st_ctc", "mt_states", "unit_ctc")):
        a = torch.cat([x[i] for x in alone])
        p = (p_asr, p_st, p_mt, p_unit)[i]
        assert a.shape == p.shape
        bit[name] = float((a.double() - p.double()).abs().max())
        assert torch.equal(a, p), f"{name}: logits of an utterance alone and in a pack of {pack} differ by {bit[name]:.3e} -- not pack-invariant"
    assert [x[4][0] for x in alone] == p_tok

    # ---- single-utterance entry points (the agents' path: other kernels, same B = 1 arithmetic) ----
    s_asr, s_st, s_mt, s_unit, s_tok = [], [], [], [], []
    for u, pcm in zip(utts, pcms):
        enc = m.encoder_forward(m.fbank_cmvn(pcm))
        s_asr.append(m.ctc_greedy(0, enc, want_logits=True)[3].cpu())
        s_st.append(m.ctc_greedy(1, enc, want_logits=True)[3].cpu())
        toks, feats = mt_greedy(m, enc, max_new_tokens=u.n_mt)
        assert len(toks) == u.n_mt + 1 and toks[-1] == cfg.eos
        s_tok.append(toks)
        s_mt.append(feats[: u.n_mt + 1].cpu())
        s_unit.append(m.t2u_units(feats[: u.n_mt + 1], want_logits=True)[2].cpu())
    assert p_tok == s_tok

    report = {}
    _stage("asr_ctc", torch.cat(s_asr), p_asr, [cfg.pad, cfg.unk], report)
    _stage("st_ctc", torch.cat(s_st), p_st, [cfg.pad, cfg.unk], report)
    _stage("unit_ctc", torch.cat(s_unit), p_unit, [cfg.pad, cfg.unk], report)
    # MT greedy: next-token logits = decoder state . E^T (tied projection); the last row of every utterance is the forced </s>
    keep = torch.cat([torch.arange(u.n_mt + 1) < u.n_mt for u in utts])
    lm_s = (torch.cat(s_mt).double() @ emb.T)[keep]
    lm_p = (p_mt.double() @ emb.T)[keep]
    _stage("mt_greedy", lm_s.float(), lm_p.float(), [cfg.pad, cfg.eos], report)
    print(f"pack of {pack}, {bucket} bucket ({utts[-1].seconds:.2f}-{utts[0].seconds:.2f} s): alone-vs-pack max_logit_diff = {bit} (bitwise);"
          f" single-entry-point margins: {report}")
