"""Pins the CPU oracle: (a) the reference's own known-answer tests, (b) the committed fixtures
produced by the reference's modules (oracle/make_golden.py), (c) when /root/reference is
present, the reference modules live."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import kaldi_fbank as K
from oracle import ref_loader
from oracle import streamspeech_oracle as O
from streamspeech_amd import synth


def _gold(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_kat_rel_positional_encoding(golden_dir):
    """fairseq/tests/test_positional_encoding.py:17-59."""
    g = _gold(golden_dir, "kat_espnet.npz")
    assert np.allclose(O.rel_pos_table(4, 2).numpy(), g["expected_pe_len4"], atol=1e-4)
    assert np.allclose(O.rel_pos_table(3, 2).numpy(), g["expected_pos_T3"], atol=1e-4)


def test_kat_rel_shift(golden_dir):
    """fairseq/tests/test_espnet_multihead_attention.py:99-118."""
    g = _gold(golden_dir, "kat_espnet.npz")
    x = torch.from_numpy(g["sample_x"])[0]            # [1,3,5]
    out = O.rel_shift_closed(x)
    assert np.allclose(out.numpy(), g["expected_rel_shift"][0], atol=1e-4)


def test_kat_relpos_forward(golden_dir):
    """fairseq/tests/test_espnet_multihead_attention.py:120-147.  That test feeds a [B=1, 2T-1, C]
    pos tensor where the module expects time-major, so linear_pos sees 5 'batches' of one
    position; each batch b then equals attention with a single positional row.  With one
    position rel_shift degenerates to the identity on a length-1 axis and the BD term is constant
    over keys (softmax-invariant), so every batch must reproduce plain (q+u).k attention."""
    g = _gold(golden_dir, "kat_espnet.npz")
    sd = {"a." + k[4:]: g[k] for k in g.files if k.startswith("mha.")}
    x = torch.from_numpy(g["sample"])[:, 0]           # [3,2]
    s = O.SD(sd)
    q = O.linear(x, s, "a.linear_q") + s["a.pos_bias_u"][0]
    k = O.linear(x, s, "a.linear_k")
    v = O.linear(x, s, "a.linear_v")
    att = torch.softmax(q @ k.t() / np.sqrt(2.0), -1) @ v
    out = O.linear(att, s, "a.linear_out")
    expect = g["expected_forward"].reshape(5, 3, 2)
    for b in range(5):
        assert np.allclose(out.numpy(), expect[b], atol=1e-4)
    assert np.allclose(g["ref_forward"], g["expected_forward"], atol=1e-4)


def test_chunk_causal_conv_golden(golden_dir):
    g = _gold(golden_dir, "chunk_causal_conv.npz")
    for name, (cin, cout, k, stride, groups) in {"sub": (12, 10, 5, 2, 1), "dw": (16, 16, 31, 1, 16)}.items():
        wshape = (cout, cin // groups, k)
        w = torch.from_numpy(synth.normal(0, f"cc/{name}/w", wshape, 0.3))
        b = torch.from_numpy(synth.normal(0, f"cc/{name}/b", (cout,), 0.1))
        for cs in (8, 16, 999999):
            for L in (37, 64, 5):
                x = torch.from_numpy(synth.normal(0, f"cc/{name}/x/{L}", (cin, L), 1.0))
                y = O.chunk_causal_conv1d(x, w, b, stride, cs, groups)
                assert np.abs(y.numpy() - g[f"{name}_cs{cs}_L{L}"]).max() < 1e-5


@pytest.mark.parametrize("tag,ac,cc", [("offline", 999999, 999999), ("c8", 8, 8), ("c24", 24, 16)])
def test_encoder_golden(golden_dir, synth_weights, tag, ac, cc):
    cfg, _, sd, _ = synth_weights
    g = _gold(golden_dir, "encoder.npz")
    out = O.encoder_forward(sd, synth.synth_fbank(0, int(g["T"])), cfg, ac, cc)
    assert np.abs(out.numpy() - g[f"enc_{tag}"]).max() < 1e-4
    if tag in ("offline", "c8"):
        for head in ("source_unigram", "ctc_target_unigram"):
            toks, idx, raw, _ = O.ctc_head(sd, torch.from_numpy(g[f"enc_{tag}"]), head, cfg)
            assert raw == g[f"{head}_{tag}_raw"].tolist()
            assert toks == g[f"{head}_{tag}_tokens"].tolist() and idx == g[f"{head}_{tag}_index"].tolist()


def test_decoders_golden(golden_dir, synth_weights):
    cfg, _, sd, _ = synth_weights
    ge, gd = _gold(golden_dir, "encoder.npz"), _gold(golden_dir, "decoders.npz")
    enc = torch.from_numpy(ge["enc_offline"])
    toks = gd["mt_tokens_in"].tolist()
    feats = O.mt_decoder_features(sd, toks, enc, cfg)
    assert np.abs(feats.numpy() - gd["mt_features"]).max() < 1e-4
    assert O.mt_greedy(sd, enc, cfg, prefix=toks[1:], max_new_tokens=5) == gd["mt_greedy_prefix9_new5"].tolist()
    t2u = O.t2u_encoder(sd, torch.from_numpy(gd["mt_features"]), cfg)
    assert np.abs(t2u.numpy() - gd["t2u_out"]).max() < 1e-4
    t2u_uni = O.t2u_encoder(sd, torch.from_numpy(gd["mt_features"]), cfg, causal=True)
    assert np.abs(t2u_uni.numpy() - gd["t2u_out_uni"]).max() < 1e-4
    logits = O.unit_decoder_logits(sd, torch.from_numpy(gd["t2u_out"]), cfg)
    assert np.abs(logits[:8].numpy() - gd["unit_logits_first8"]).max() < 1e-3
    units, raw = O.unit_ctc_generate(logits, cfg)
    assert raw == gd["unit_raw"].tolist() and units == gd["units"].tolist()


def test_vocoder_golden(golden_dir, synth_weights):
    _, vcfg, _, vsd = synth_weights
    g = _gold(golden_dir, "vocoder.npz")
    wav, dur = O.vocoder_forward(vsd, g["codes"].tolist(), vcfg, True)
    assert dur.tolist() == g["dur"].tolist()
    assert np.abs(wav.numpy() - g["wav"]).max() < 1e-4
    wav1, _ = O.vocoder_forward(vsd, g["codes"].tolist(), vcfg, False)
    assert np.abs(wav1.numpy() - g["wav_nodur"]).max() < 1e-4


def test_unit_decoder_position_quirk(synth_weights):
    """a T2U state whose first feature equals padding_idx (1.0) exactly gets the
    zero positional row."""
    cfg, _, sd, _ = synth_weights
    x = torch.from_numpy(synth.normal(3, "quirk", (3, cfg.dec_dim), 1.0)).clone()
    a = O.unit_decoder_logits(sd, x, cfg)
    x2 = x.clone()
    x2[1, 0] = 1.0
    b = O.unit_decoder_logits(sd, x2, cfg)
    assert not torch.allclose(a[25:50], b[25:50])


FBANK_MAX_TOL, FBANK_RMS_TOL = 5e-4, 1e-5      # log-mel values ~20; observed max 1.5e-4, RMS 1-3e-6 (float32 FFT / log rounding)


def test_fbank_oracle_matches_third_party_kaldi_implementation(golden_dir):
    """the Kaldi fbank restatement against the committed outputs of an independent implementation of the
    same algorithm with the reference's arguments -- transformers' SeamlessM4TFeatureExtractor numpy path, "to mimic Kaldi"
    (oracle/make_golden_fbank.py) -- on noise, tonal, near-silent, ragged-length and one-frame inputs."""
    from oracle import make_golden_fbank as MG
    g = _gold(golden_dir, "kaldi_fbank_hf.npz")
    for name, (kind, seed, n) in MG.CASES.items():
        x = MG.waveform(kind, seed, n)
        mine = K.fbank(x * np.float32(32768.0))
        ref = g[name]
        assert mine.shape == ref.shape == (K.num_frames(n), 80), name
        err = np.abs(mine - ref)
        assert err.max() < FBANK_MAX_TOL and np.sqrt(np.mean(err.astype(np.float64) ** 2)) < FBANK_RMS_TOL, (name, err.max())


def test_fbank_oracle_matches_live_transformers_extractor():
    """The same, live, on a waveform that is not in the fixture (skipped where transformers is not importable)."""
    from oracle import make_golden_fbank as MG
    x = MG.waveform("tonal", 99, 23456)
    try:
        ref = MG.hf_fbank(x)       # imports transformers with the reference loader's torchaudio stub hidden
    except ImportError:
        pytest.skip("transformers not importable")
    mine = K.fbank(x * np.float32(32768.0))
    assert mine.shape == ref.shape and np.abs(mine - ref).max() < FBANK_MAX_TOL


def test_fbank_oracle_properties(golden_dir):
    """Kaldi fbank restatement -- structural checks (frame count, filterbank shape, a tone lands in its bin)."""
    g = _gold(golden_dir, "gcmvn_fr-en.npz")
    assert K.num_frames(399) == 0 and K.num_frames(400) == 1 and K.num_frames(16000) == 98
    mb = K.mel_banks()
    assert mb.shape == (80, 257) and (mb >= 0).all() and mb[:, 256].max() == 0 and (mb.sum(1) > 0).all()
    pcm = synth.synth_pcm(0, 16000)
    f = K.online_features(pcm, g["mean"], g["std"])
    assert f.shape == (98, 80) and np.isfinite(f).all()
    # a pure tone lights up the expected mel bin
    t = np.arange(16000) / 16000.0
    tone = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32) * 32768
    fb = K.fbank(tone)
    centers = 700.0 * (np.exp((K.mel_scale(20.0) + (np.arange(80) + 1) * (K.mel_scale(8000.0) - K.mel_scale(20.0)) / 81) / 1127.0) - 1)
    assert abs(centers[int(fb.mean(0).argmax())] - 1000.0) < 80.0


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not present")
def test_oracle_matches_live_reference_modules(synth_weights):
    """Second pin, live: a different seed / length than the fixtures."""
    from oracle import ref_build
    cfg, vcfg, sd, vsd = synth_weights
    T = 57
    fb = synth.synth_fbank(21, T)
    with torch.no_grad():
        enc = ref_build.build_encoder(sd, cfg, 8, 8)
        ref = enc(torch.from_numpy(fb)[None], torch.tensor([T]))["encoder_out"][0][:, 0]
        mine = O.encoder_forward(sd, fb, cfg, 8, 8)
        assert (ref - mine).abs().max() < 1e-4
        voc = ref_build.build_vocoder(vsd, vcfg)
        codes = [3, 999, 17, 17, 250, 0]
        wav, dur = voc(code=torch.tensor([codes]), dur_prediction=True)
        mw, md = O.vocoder_forward(vsd, codes, vcfg, True)
        assert dur.view(-1).tolist() == md.tolist() and (wav.squeeze() - mw).abs().max() < 1e-4


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not present")
def test_float64_oracle_mode_matches_the_reference_modules_in_double(synth_weights):
    """The adjudicator of near-tie arg-max rows (oracle/adjudicate.py) is the oracle in float64: pin THAT mode too, against the
    reference's own encoder and CTC head modules cast to double (same float32 weights), on an input the fixtures do not hold."""
    from oracle import ref_build
    cfg, vcfg, sd, vsd = synth_weights
    T = 83
    fb = synth.synth_fbank(33, T)
    with torch.no_grad():
        enc = ref_build.build_encoder(sd, cfg, 999999, 999999).double()
        ref = enc(torch.from_numpy(fb).double()[None], torch.tensor([T]))["encoder_out"][0][:, 0]
        sd64 = O.SD(sd, dtype=torch.float64)
        mine = O.encoder_forward(sd64, fb, cfg)
        assert ref.dtype == mine.dtype == torch.float64
        err = float((ref - mine).abs().max())
        assert err < 1e-10, err                    # observed 6e-15: the same arithmetic, operation by operation
        head = ref_build.build_ctc_head(sd, cfg, "source_unigram").double()
        logits_ref = head(ref[:, None, :])["encoder_out"][:, 0]
        logits = O.ctc_head(sd64, mine, "source_unigram", cfg)[3]
        assert float((logits_ref - logits).abs().max()) < 1e-9
        # and float32 sits ~2^-20 x max|logit| from it: the size of the bar the adjudication uses
        l32 = O.ctc_head(O.SD(sd), O.encoder_forward(O.SD(sd), fb, cfg), "source_unigram", cfg)[3]
        d = float((l32.double() - logits).abs().max())
        scale = float(logits.abs().max())
        assert 0.1 * scale * 2.0 ** -20 < d < 16 * scale * 2.0 ** -20, (d, scale)
        # the unit-decoder side of the adjudicator (MT decoder states -> T2U encoder -> CTC unit decoder) in double as well.
        # fairseq's MultiheadAttention rounds its softmax to float32 whatever the module dtype (fairseq/utils.py:514-518), so the
        # reference modules in double still carry ~2e-7 of float32 there; the adjudicator must not, so for this comparison the
        # cast is lifted from the reference (observed without the lift: 2.3e-7 on the MT states; with it: below 1e-12).
        import fairseq.utils as fu
        import torch.nn.functional as F
        toks = [cfg.eos] + [int(t) for t in synth.uniform(33, "f64/mt_tokens", (7,), 4, cfg.tgt_vocab)]
        keep = fu.softmax
        fu.softmax = lambda x, dim, onnx_trace=False: F.softmax(x, dim=dim)
        try:
            mt = ref_build.build_mt_decoder(sd, cfg).double()
            feats_ref = mt(torch.tensor([toks]), encoder_out={"encoder_out": [ref[:, None]], "encoder_padding_mask": []}, features_only=True)[0][0]
            t2u_ref = ref_build.build_t2u_encoder(sd, cfg).double()(feats_ref[:, None], None)["encoder_out"][0][:, 0]
            ul_ref, _ = ref_build.build_unit_decoder(sd, cfg).double()(None, encoder_out={"encoder_out": [t2u_ref[:, None]], "encoder_padding_mask": []})
        finally:
            fu.softmax = keep
        feats = O.mt_decoder_features(sd64, toks, mine, cfg)
        assert feats.dtype == torch.float64 and float((feats_ref - feats).abs().max()) < 1e-9
        t2u = O.t2u_encoder(sd64, feats, cfg)
        assert float((t2u_ref - t2u).abs().max()) < 1e-9
        ul = O.unit_decoder_logits(sd64, t2u, cfg)
        assert ul.dtype == torch.float64 and float((ul_ref[0] - ul).abs().max()) < 1e-8
