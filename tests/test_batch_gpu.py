"""Ragged-batch stages (BASELINE.json configs[3]/[4]): B utterances of different lengths packed
together must give, per utterance, exactly what the single-utterance path gives (identical ids;
floats within summation-order noise) -- there is no padding, so no batch effect may leak."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LENS = [83, 435, 9, 257, 31, 612]   # fbank frames per utterance (ragged, incl. tiny)


def _fbanks():
    from streamspeech_amd import synth
    return [synth.synth_fbank(40 + i, T) for i, T in enumerate(LENS)]


@pytest.mark.parametrize("ac,cc", [(999999, 999999), (8, 8)])
def test_batch_encoder_and_ctc_match_single(hip_model, ac, cc):
    fbs = _fbanks()
    packed = torch.from_numpy(np.concatenate(fbs)).cuda()
    enc, Tp = hip_model.batch_encoder_forward(packed, LENS, ac, cc)
    ctc = [hip_model.batch_ctc_greedy(h, enc, Tp) for h in (0, 1)]
    off = 0
    for b, fb in enumerate(fbs):
        one = hip_model.encoder_forward(torch.from_numpy(fb).cuda(), ac, cc)
        assert one.shape[0] == Tp[b]
        err = (enc[off:off + Tp[b]] - one).abs().max().item()
        assert err < 2e-4, f"utt {b}: {err}"
        for h in (0, 1):
            toks, idx, _, _ = hip_model.ctc_greedy(h, one)
            assert ctc[h][b][0] == toks and ctc[h][b][1] == idx
        off += Tp[b]


def test_batch_fbank_matches_single(hip_model):
    from streamspeech_amd import synth
    ns = [16000, 4000, 23017, 400]
    pcms = [synth.synth_pcm(70 + i, n) for i, n in enumerate(ns)]
    feat, T = hip_model.batch_fbank_cmvn(torch.from_numpy(np.concatenate(pcms)).cuda(), ns)
    off = 0
    for p, t in zip(pcms, T):
        one = hip_model.fbank_cmvn(torch.from_numpy(p).cuda())
        assert one.shape[0] == t and torch.equal(feat[off:off + t], one)
        off += t


def test_batch_mt_t2u_vocoder_match_single(hip_model, hip_vocoder, synth_weights):
    from streamspeech_amd import synth
    from streamspeech_amd.pipeline import mt_greedy, units_from_tokens
    cfg = hip_model.cfg
    fbs = _fbanks()[:4]
    lens = LENS[:4]
    max_new = [7, 12, 3, 9]
    enc, Tp = hip_model.batch_encoder_forward(torch.from_numpy(np.concatenate(fbs)).cuda(), lens)
    toks, feats, n = hip_model.batch_mt_greedy(enc, Tp, max_new)
    units_b = hip_model.batch_t2u_units(feats, n)
    off = 0
    all_units = []
    for b in range(4):
        e1 = enc[off:off + Tp[b]].contiguous()
        t1, f1 = mt_greedy(hip_model, e1, max_new_tokens=max_new[b])
        assert toks[b] == t1, (b, toks[b], t1)
        assert n[b] == f1.shape[0]
        assert (feats[b, :n[b]] - f1).abs().max().item() < 2e-4
        u1, _, _ = hip_model.t2u_units(f1)
        assert units_b[b] == u1
        all_units.append(units_from_tokens(u1, cfg))
        off += Tp[b]
    codes = [u if len(u) > 0 else [1, 2, 3] for u in all_units]
    wavs, dur, K = hip_vocoder.batch_forward(codes, dur_prediction=True)
    dur_h = dur.cpu().tolist()
    o = 0
    for b, c in enumerate(codes):
        w1, d1 = hip_vocoder.forward(c, True)
        assert dur_h[o:o + K[b]] == d1.cpu().tolist()
        assert wavs[b].shape == w1.shape
        rms = float(torch.sqrt(torch.mean((wavs[b] - w1) ** 2)))
        assert rms < 1e-5, f"utt {b}: rms {rms}"
        o += K[b]


def test_batch_vocoder_forced_durations_vs_oracle(hip_vocoder, synth_weights):
    from oracle import streamspeech_oracle as O
    from streamspeech_amd import synth
    _, vcfg, _, vsd = synth_weights
    codes = [[int(c) for c in synth.uniform(5, f"bv/{i}", (k,), 0, 1000)] for i, k in enumerate((30, 7, 55))]
    durs = [[1 + (j % 3 == 2) for j in range(len(c))] for c in codes]
    wavs, dur, K = hip_vocoder.batch_forward(codes, True, forced_dur=durs)
    for b in range(3):
        rw, rd = O.vocoder_forward(vsd, codes[b], vcfg, True, forced_dur=durs[b])
        assert wavs[b].numel() == rw.numel()
        rms = float(torch.sqrt(torch.mean((wavs[b].cpu() - rw) ** 2)))
        assert rms < 1e-3, f"rms {rms}"


def test_batch_vocoder_on_stream_k_kernels_vs_oracle(hip_vocoder, synth_weights):
    """Same ragged batch with every eligible conv forced through the persistent stream-K kernel
    (segment edges inside 128-row tiles, tiles shared by several workgroups)."""
    from oracle import streamspeech_oracle as O
    from streamspeech_amd import lib as L, synth
    lib = L.load()
    _, vcfg, _, vsd = synth_weights
    codes = [[int(c) for c in synth.uniform(5, f"bv/{i}", (k,), 0, 1000)] for i, k in enumerate((30, 7, 55, 1))]
    durs = [[1 + (j % 3 == 2) for j in range(len(c))] for c in codes]
    lib.ss_debug_force_tile(1, 0, 0)
    try:
        wavs, dur, K = hip_vocoder.batch_forward(codes, True, forced_dur=durs)
        single, _ = hip_vocoder.forward(torch.tensor(codes[2], dtype=torch.int32, device="cuda:0"), True,
                                        forced_dur=torch.tensor(durs[2], dtype=torch.int32, device="cuda:0"))
    finally:
        lib.ss_debug_force_tile(0, 0, 0)
    assert lib.ss_debug_sk_errors() == 0
    for b in range(4):
        rw, rd = O.vocoder_forward(vsd, codes[b], vcfg, True, forced_dur=durs[b])
        assert wavs[b].numel() == rw.numel()
        rms = float(torch.sqrt(torch.mean((wavs[b].cpu() - rw) ** 2)))
        assert rms < 1e-3, f"utt {b}: rms {rms}"
    assert float(torch.sqrt(torch.mean((single.cpu() - wavs[2].cpu()) ** 2))) < 1e-5


def test_offline_driver_writes_fairseq_generate_format(hip_model, hip_vocoder, tmp_path):
    """§8f-4: generate-<subset>.log/.txt in the reference's line format, the files pred.offline-s2st.sh
    cuts out of them, <n>_pred.wav dumps; hypotheses equal the single-utterance path."""
    from streamspeech_amd import frontend, offline, synth
    from streamspeech_amd.modules import Dictionary
    cfg = hip_model.cfg
    dicts = {k: Dictionary.placeholder(n) for k, n in (("source_unigram", cfg.src_vocab), ("ctc_target_unigram", cfg.tgt_vocab),
                                                       ("target_unigram", cfg.tgt_vocab))}
    secs = [1.3, 2.9, 0.8, 2.1, 1.7]
    items = [(10 + i, torch.from_numpy(synth.synth_pcm(70 + i, int(16000 * s))).to(hip_model.device)) for i, s in enumerate(secs)]
    hyps = offline.generate(hip_model, hip_vocoder, items, dicts, str(tmp_path), "test", batch_size=3, max_len_a_mt=0.0,
                            max_len_b_mt=12, dur_prediction=True, dump_wav=True)
    assert sorted(hyps) == [10, 11, 12, 13, 14]
    log = (tmp_path / "generate-test.log").read_text().splitlines()
    assert sorted(ln.split("\t")[0] for ln in log) == sorted(f"{p}-{i}" for p in "ASD" for i in range(10, 15))
    res = (tmp_path / "generate-test.txt").read_text().splitlines()
    units = {}
    for ln in res:
        tag, score, u = ln.split("\t")
        assert tag[:2] in ("H-", "D-") and float(score) == 0.0
        units[int(tag[2:])] = [int(x) for x in u.split()]
    unit_file = (tmp_path / "generate-test.unit").read_text().splitlines()
    assert [[int(x) for x in ln.split()] for ln in unit_file] == [units[i] for i in range(10, 15)]
    assert len((tmp_path / "generate-test.asr").read_text().splitlines()) == 5
    # same hypotheses as the one-utterance-at-a-time path, wav dump = that path's waveform as PCM16
    for n, (sid, pcm) in enumerate(items[:3]):
        fb = hip_model.fbank_cmvn(pcm)
        enc = hip_model.encoder_forward(fb)
        from streamspeech_amd.pipeline import mt_greedy, units_from_tokens
        toks, feats = mt_greedy(hip_model, enc, max_new_tokens=12)
        u_toks, _, _ = hip_model.t2u_units(feats[: len(toks)], mask_eos=True)
        assert units_from_tokens(u_toks, cfg) == hyps[sid]["units"]
        if hyps[sid]["units"]:
            y, sr = frontend.read_wav(tmp_path / "pred_wav" / f"{n}_pred.wav")
            w = hyps[sid]["wav"].cpu().numpy()
            assert sr == 16000 and y.shape == w.shape and np.abs(np.clip(w, -1, 1) - y).max() < 1e-4



def test_multilingual_models_coexist_with_mixed_length_batches(synth_weights, golden_dir):
    """BASELINE.json configs[4] shape: fr/es/de checkpoints (same architecture, different weights and
    CMVN statistics -- configs/{fr,es,de}-en/gcmvn.npz) live side by side on one GPU and serve
    mixed-length ragged batches; every utterance equals its one-at-a-time result."""
    import os
    from streamspeech_amd import synth
    from streamspeech_amd.config import ModelConfig
    from streamspeech_amd.engine import HipModel
    cfg = ModelConfig()
    models = {}
    for seed, lang in enumerate(("fr", "es", "de")):
        g = np.load(os.path.join(golden_dir, f"gcmvn_{lang}-en.npz"))
        models[lang] = HipModel(synth.make_model_state_dict(seed, cfg), cfg, cmvn_mean=g["mean"], cmvn_std=g["std"])
    lens = [16000 * 6, 16000 * 1 + 37, 16000 * 3, 16000 * 11]
    pcm = [torch.from_numpy(synth.synth_pcm(50 + i, n)).to("cuda:0") for i, n in enumerate(lens)]
    outs = {}
    for lang, m in models.items():
        feat, T = m.batch_fbank_cmvn(torch.cat(pcm), lens)
        enc, Tp = m.batch_encoder_forward(feat, T)
        asr = m.batch_ctc_greedy(0, enc, Tp)
        off = 0
        for b, p in enumerate(pcm):
            one = m.encoder_forward(m.fbank_cmvn(p))
            assert (enc[off:off + Tp[b]] - one).abs().max() < 5e-5
            assert asr[b][0] == m.ctc_greedy(0, one)[0]
            off += Tp[b]
        outs[lang] = enc
    assert (outs["fr"] - outs["es"]).abs().max() > 1e-2 and (outs["es"] - outs["de"]).abs().max() > 1e-2


def test_offline_driver_is_deterministic_and_shards_cover_the_set(hip_model, hip_vocoder, tmp_path):
    """configs[3] shape in miniature: a few hundred utterances through length-bucketed ragged batches.
    Two runs give bit-identical hypotheses (fixed summation orders everywhere, stream-K included);
    the two shards of a 2-way split (fairseq --num-shards/--shard-id) cover every id exactly once."""
    from streamspeech_amd import offline, synth
    from streamspeech_amd.modules import Dictionary
    cfg = hip_model.cfg
    dicts = {k: Dictionary.placeholder(n) for k, n in (("source_unigram", cfg.src_vocab), ("ctc_target_unigram", cfg.tgt_vocab),
                                                       ("target_unigram", cfg.tgt_vocab))}
    durs = synth.synth_durations(77, 192)
    items = [(i, torch.from_numpy(synth.synth_pcm(300 + i, int(16000 * min(float(d), 4.0)))).to(hip_model.device))
             for i, d in enumerate(durs)]
    kw = dict(batch_size=32, max_len_a_mt=0.0, max_len_b_mt=6, dur_prediction=True, dump_wav=False)
    a = offline.generate(hip_model, hip_vocoder, items, dicts, str(tmp_path / "a"), "test", **kw)
    b = offline.generate(hip_model, hip_vocoder, items, dicts, str(tmp_path / "b"), "test", **kw)
    assert sorted(a) == list(range(192))
    for i in a:
        assert a[i]["units"] == b[i]["units"] and a[i]["asr"] == b[i]["asr"] and a[i]["mt"] == b[i]["mt"]
    assert (tmp_path / "a" / "generate-test.unit").read_text() == (tmp_path / "b" / "generate-test.unit").read_text()
    s0 = offline.generate(hip_model, hip_vocoder, items[0::2], dicts, str(tmp_path / "s0"), "test.shard0", **kw)
    s1 = offline.generate(hip_model, hip_vocoder, items[1::2], dicts, str(tmp_path / "s1"), "test.shard1", **kw)
    assert sorted(list(s0) + list(s1)) == list(range(192))
    assert hip_model.lib.ss_debug_sk_errors() == 0


def test_concurrent_contexts_on_separate_streams_are_bitwise_reproducible(hip_model, hip_vocoder):
    """bench.py's serving mode: several contexts (own scratch / stream-K workspace, shared weights) run
    ragged batches at the same time from worker threads.  Every result equals the one computed alone."""
    import threading
    from streamspeech_amd import lib as L, synth
    lib = L.load()
    codes = [[[int(c) for c in synth.uniform(9, f"cc/{w}/{i}", (k,), 0, 1000)] for i, k in enumerate((120, 35, 260, 77))] for w in range(3)]
    durs = [[[1 + ((i + j) % 3 == 0) for j in range(len(c))] for i, c in enumerate(cw)] for cw in codes]
    lib.ss_debug_force_tile(1, 0, 0)          # every eligible conv through stream-K: the fix-up paths run concurrently
    try:
        alone = []
        for w in range(3):
            wavs, _, _ = hip_vocoder.batch_forward(codes[w], True, forced_dur=durs[w])
            alone.append([x.clone() for x in wavs])
        ctxs = [hip_vocoder.new_context() for _ in range(3)]
        streams = [torch.cuda.Stream() for _ in range(3)]
        got, errs = [None] * 3, []

        def work(w):
            try:
                torch.cuda.set_device(0)
                with torch.cuda.stream(streams[w]):
                    for _ in range(4):
                        wavs, _, _ = ctxs[w].batch_forward(codes[w], True, forced_dur=durs[w])
                    streams[w].synchronize()
                    got[w] = [x.clone() for x in wavs]
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=work, args=(w,)) for w in range(3)]
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
    finally:
        lib.ss_debug_force_tile(0, 0, 0)
    assert not errs, errs
    for w in range(3):
        for a, b in zip(alone[w], got[w]):
            assert torch.equal(a, b)
    assert lib.ss_debug_sk_errors() == 0


def test_fused_resblock_pairs_equal_the_two_launch_form_bitwise(hip_vocoder, synth_weights):
    """conv_pair_kernel (narrow stages: conv1 -> leaky-ReLU -> conv2 -> +x with the intermediate in LDS) performs
    the same per-element arithmetic as the two separate launches: waveforms are bit-identical, single and ragged."""
    from streamspeech_amd import lib as L, synth
    lib = L.load()
    codes = [[int(c) for c in synth.uniform(13, f"fp/{i}", (k,), 0, 1000)] for i, k in enumerate((90, 33, 150))]
    durs = [[1 + (j % 4 == 1) for j in range(len(c))] for c in codes]
    lib.ss_debug_conv_c32(4)                   # direct-form kernels on both legs (see the ResBlock test below)
    try:
        fused, _, _ = hip_vocoder.batch_forward(codes, True, forced_dur=durs)
        f1, _ = hip_vocoder.forward(torch.tensor(codes[2], dtype=torch.int32, device="cuda:0"), True,
                                    forced_dur=torch.tensor(durs[2], dtype=torch.int32, device="cuda:0"))
        fused = [w.clone() for w in fused]; f1 = f1.clone()
        lib.ss_debug_force_tile(3, 0, 0)
        try:
            plain, _, _ = hip_vocoder.batch_forward(codes, True, forced_dur=durs)
            p1, _ = hip_vocoder.forward(torch.tensor(codes[2], dtype=torch.int32, device="cuda:0"), True,
                                        forced_dur=torch.tensor(durs[2], dtype=torch.int32, device="cuda:0"))
        finally:
            lib.ss_debug_force_tile(0, 0, 0)
    finally:
        lib.ss_debug_conv_c32(5)
    for a, b in zip(fused, plain):
        assert torch.equal(a, b)
    assert torch.equal(f1, p1)


def test_fused_narrow_resblocks_equal_the_multi_launch_form_bitwise(hip_vocoder):
    """resblock_fused_kernel (resblock.hip: the three (dilated conv, conv, residual) pairs of a C = 32 / 16 ResBlock in one
    persistent launch, raw residual stream in registers, halo rows recomputed) performs the same per-element arithmetic as
    the conv_pair / conv_slab launches it replaces (hifigan.py:95-102,159-165): waveforms are bit-identical -- ragged batch
    with a one-frame utterance and a long one (several row blocks per utterance, block edges inside and at utterance
    edges), and the single-utterance entry point."""
    from streamspeech_amd import lib as L, synth
    lib = L.load()
    codes = [[int(c) for c in synth.uniform(17, f"frb/{i}", (k,), 0, 1000)] for i, k in enumerate((120, 1, 33, 2, 260, 7))]
    durs = [[1 + (j % 3 == 1) for j in range(len(c))] for c in codes]
    one = lambda i: hip_vocoder.forward(torch.tensor(codes[i], dtype=torch.int32, device="cuda:0"), True,   # noqa: E731
                                        forced_dur=torch.tensor(durs[i], dtype=torch.int32, device="cuda:0"))[0].clone()
    # (the comparison is between the DIRECT-form kernels: the Winograd form of the 32-channel per-conv launches -- conv_c64w.hip, same
    #  float32 error but another chain of roundings -- is switched off for both legs; the multi-launch leg would otherwise take it
    #  for the k = 3 / 7 convs that the fused launch computes in direct form)
    lib.ss_debug_conv_c32(4)
    try:
        fused = [w.clone() for w in hip_vocoder.batch_forward(codes, True, forced_dur=durs)[0]]
        f_single = [one(i) for i in (1, 4)]
        lib.ss_debug_force_tile(6, 0, 0)           # narrow-stage ResBlocks as pair / two-launch kernels
        try:
            plain = [w.clone() for w in hip_vocoder.batch_forward(codes, True, forced_dur=durs)[0]]
            p_single = [one(i) for i in (1, 4)]
        finally:
            lib.ss_debug_force_tile(0, 0, 0)
    finally:
        lib.ss_debug_conv_c32(5)
    for i, (a, b) in enumerate(zip(fused, plain)):
        assert a.numel() == 320 * sum(durs[i]) and torch.isfinite(a).all()
        assert torch.equal(a, b), f"utterance {i}: max diff {(a - b).abs().max().item()}"
    # single-utterance entry point: the long one takes the same slab kernels in the multi-launch form -> bitwise; the
    # one-frame utterance (160 / 320 rows per narrow stage) runs there on the LDS-tiled conv_gemm kernel (M < 2048), whose
    # summation tree differs -> float noise only
    assert torch.equal(f_single[1], p_single[1])
    assert (f_single[0] - p_single[0]).abs().max().item() < 2e-5
    assert (f_single[1] - fused[4]).abs().max().item() < 2e-5   # ragged pack vs single-utterance entry point (other tile kernels upstream)
