"""Waveform front-end: resampler oracle pinned against scipy.signal.resample_poly,
the product filter design equals the oracle's, WAV read/write round trip."""
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.resample import design_filter as oracle_design, resample_poly_ref
from streamspeech_amd import frontend, synth


@pytest.mark.parametrize("sr_in", [48000, 44100, 22050, 8000, 16000])
def test_resampler_oracle_matches_scipy(sr_in):
    from scipy.signal import resample_poly
    x = synth.synth_pcm(3, 4801)
    g = math.gcd(16000, sr_in)
    want = resample_poly(x.astype(np.float64), 16000 // g, sr_in // g)
    got = resample_poly_ref(x, 16000, sr_in)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 5e-7


def test_product_filter_design_is_the_oracle_design():
    for up, down in ((1, 3), (160, 441), (2, 1)):
        assert np.array_equal(frontend.design_filter(up, down), oracle_design(up, down))
    h = frontend.design_filter(1, 3)
    assert len(h) == 61 and abs(h.sum() - 1.0) < 1e-12 and np.allclose(h, h[::-1])


def test_resampler_properties():
    # linearity, and a 1 kHz tone at 48 kHz stays a 1 kHz tone of the same amplitude at 16 kHz
    a, b = synth.synth_pcm(1, 3000), synth.synth_pcm(2, 3000)
    lhs = resample_poly_ref(2.0 * a + b, 16000, 48000)
    rhs = 2.0 * resample_poly_ref(a, 16000, 48000) + resample_poly_ref(b, 16000, 48000)
    assert np.abs(lhs - rhs).max() < 1e-6
    t = np.arange(48000) / 48000.0
    y = resample_poly_ref(np.sin(2 * np.pi * 1000 * t).astype(np.float32), 16000, 48000)
    ref = np.sin(2 * np.pi * 1000 * np.arange(16000) / 16000.0)
    assert np.abs(y[200:-200] - ref[200:-200]).max() < 2e-3
    assert resample_poly_ref(a[:0], 16000, 48000).shape == (0,)


def test_wav_round_trip(tmp_path):
    x = synth.synth_pcm(5, 1234)
    p = tmp_path / "a.wav"
    frontend.write_wav(p, x, 16000)
    y, sr = frontend.read_wav(p)
    assert sr == 16000 and y.shape == x.shape and np.abs(x - y).max() <= 1e-4
    with pytest.raises(IOError):
        frontend.read_wav("example.mp3")


def test_ordered_batches_are_length_sorted_and_bounded():
    from streamspeech_amd.offline import detok, ordered_batches
    b = ordered_batches([5, 50, 7, 49, 48, 6], 2)
    assert b == [[1, 3], [4, 2], [5, 0]]
    b = ordered_batches([10, 10, 10, 10], 8, max_tokens=25)
    assert b == [[0, 1], [2, 3]]
    assert detok(["▁he", "llo", "▁wor", "ld", "</s>"]) == "hello world"


def test_offline_driver_on_cpu_doubles(tmp_path):
    """The offline driver's host logic (length-sorted batches, line formats, cut files, wav dumps) over the
    oracle-backed engine: no GPU needed; hypotheses equal the oracle's one-utterance pipeline."""
    import torch
    from oracle import streamspeech_oracle as O
    from streamspeech_amd import offline
    from streamspeech_amd.config import ModelConfig, VocoderConfig
    from streamspeech_amd.modules import Dictionary
    from tests.oracle_engine import OracleEngine, OracleVocoder
    cfg, vcfg = ModelConfig(), VocoderConfig()
    sd, vsd = synth.make_model_state_dict(0, cfg), synth.make_vocoder_state_dict(0, vcfg)
    eng, voc = OracleEngine(sd, cfg), OracleVocoder(vsd, vcfg)
    dicts = {k: Dictionary.placeholder(n) for k, n in (("source_unigram", cfg.src_vocab), ("ctc_target_unigram", cfg.tgt_vocab),
                                                       ("target_unigram", cfg.tgt_vocab))}
    items = [(7 + i, torch.from_numpy(synth.synth_pcm(90 + i, int(16000 * s)))) for i, s in enumerate((0.9, 1.6, 0.7))]
    items.append((10, torch.zeros(123)))                            # shorter than one fbank window: empty hypothesis
    hyps = offline.generate(eng, voc, items, dicts, str(tmp_path), "dev", batch_size=2, max_len_a_mt=0.0, max_len_b_mt=4,
                            dur_prediction=True, dump_wav=True)
    assert sorted(hyps) == [7, 8, 9, 10] and hyps[10]["units"] == []
    log = (tmp_path / "generate-dev.log").read_text().splitlines()
    assert len(log) == 12 and log[3].startswith("A-8\t")           # after the empty one, the longest utterance comes first
    res = (tmp_path / "generate-dev.txt").read_text().splitlines()
    assert [ln.split("\t")[0][:2] for ln in res] == ["H-", "D-"] * 4
    assert len((tmp_path / "generate-dev.unit").read_text().splitlines()) == 4
    for sid, pcm in items[:3]:                                          # same hypotheses as the oracle's single-utterance path
        fb = eng.fbank_cmvn(pcm)
        enc = O.encoder_forward(O.SD(sd), fb, cfg)
        toks = O.mt_greedy(O.SD(sd), enc, cfg, max_new_tokens=4)
        assert hyps[sid]["mt"] == offline.detok([dicts["target_unigram"][t] for t in toks if t != cfg.eos])
    n_wav = len(list((tmp_path / "pred_wav").glob("*_pred.wav")))
    assert n_wav == sum(1 for h in hyps.values() if h["units"])
    # The reference's own post-processing, verbatim (researches/ctc_unity/test_scripts/pred.offline-s2st.sh:31, 37, 42-44:
    # the grep / sort / cut / sed lines that feed sacrebleu and generate_waveform_from_code.py), run over the driver's
    # generate-<subset>.log / .txt: it must cut out exactly the files the driver wrote itself.
    import subprocess
    od, sp = str(tmp_path), "dev"
    lines = {
        "asr": f"grep '^A-' {od}/generate-{sp}.log | sort -t'-' -k2,2n | cut -f2",
        "tgt": f"grep '^D-' {od}/generate-{sp}.log | sort -t'-' -k2,2n | cut -f2",
        "unit": f"grep \"^D\\-\" {od}/generate-{sp}.txt | sed 's/^D-//ig' | sort -nk1 | cut -f3",
    }
    for ext, cmd in lines.items():
        got = subprocess.run(["bash", "-c", cmd], capture_output=True, text=True, check=True).stdout
        assert got == (tmp_path / f"generate-{sp}.{ext}").read_text(), ext
    assert (tmp_path / f"generate-{sp}.unit").read_text().splitlines() == [" ".join(str(u) for u in hyps[i]["units"]) for i in sorted(hyps)]


def test_sample_history_cache_appends_without_copying_and_detects_another_source():
    """OnlineFeatureExtractor._samples: the float32 view of SimulEval's growing sample list -- only the new tail is converted (array('d')
    then a cast: the rounding of np.asarray(list, float32)), the backing store grows by doubling, earlier views stay valid, and another
    list object, a shrunk history or changed contents start a fresh cache."""
    import types
    args = types.SimpleNamespace(shift_size=10, window_size=25, sample_rate=16000, feature_dim=80)
    fe = frontend.OnlineFeatureExtractor(args, engine=None)
    rng = np.random.default_rng(5)
    hist, views = [], []
    for step in (5120, 5120, 333, 70000, 5120, 1, 40000):
        hist.extend((rng.standard_normal(step) * 0.3).tolist())
        got = fe._samples(hist, len(hist))
        assert got.dtype == np.float32 and np.array_equal(got, np.asarray(hist, dtype=np.float32))
        views.append((got, len(hist)))
    for v, n in views:                                   # views handed out earlier still show their samples
        assert np.array_equal(v, np.asarray(hist[:n], dtype=np.float32))
    buf_before = fe._buf
    hist.extend([0.25] * 100)
    fe._samples(hist, len(hist))
    assert fe._buf is buf_before                         # room left: appended in place
    other = list(hist)                                   # same contents, another list object (a new utterance's states.source)
    other[10] = 0.5
    assert np.array_equal(fe._samples(other, 3000), np.asarray(other[:3000], dtype=np.float32))
    del other[2000:]                                     # the history shrank
    assert np.array_equal(fe._samples(other, 2000), np.asarray(other, dtype=np.float32))
    other.extend([0.125] * 50)
    other[5] = -0.75                                     # contents changed under the cache: the spot check may or may not see index 5 ...
    fe.clear_cache()                                     # ... which is why the agents clear the cache in reset()
    assert np.array_equal(fe._samples(other, len(other)), np.asarray(other, dtype=np.float32))
