"""oracle/adjudicate.py and the float64 mode of the oracle (test infrastructure of tests/test_bench_config_gpu.py and of bench.py's
cpu_baseline.oracle_check): the rule lets a row through only if float32 cannot decide it."""
import numpy as np
import pytest
import torch


def _row(vals):
    x = torch.full((1, 8), -5.0, dtype=torch.float64)
    for k, v in vals.items():
        x[0, k] = v
    return x


def test_float64_oracle_is_the_same_arithmetic_in_double(synth_weights):
    from oracle import streamspeech_oracle as O
    from streamspeech_amd import synth
    cfg, vcfg, sd, vsd = synth_weights
    fb = synth.synth_fbank(5, 131)
    with torch.inference_mode():
        e32 = O.encoder_forward(O.SD(sd), fb, cfg)
        e64 = O.encoder_forward(O.SD(sd, dtype=torch.float64), fb, cfg)
        assert e32.dtype == torch.float32 and e64.dtype == torch.float64
        assert float((e32.double() - e64).abs().max()) < 5e-5
        a32 = O.ctc_head(O.SD(sd), e32, "source_unigram", cfg)
        a64 = O.ctc_head(O.SD(sd, dtype=torch.float64), e64, "source_unigram", cfg)
        scale = float(a64[3].abs().max())
        assert float((a32[3].double() - a64[3]).abs().max()) < scale * 2.0 ** -17      # float32 sits ~2^-20 x max|logit| from float64
        # the default (float32) path is bit-for-bit what it was: a second float32 view gives identical tensors
        assert torch.equal(e32, O.encoder_forward(O.SD(sd, dtype=torch.float32), fb, cfg))


def test_adjudication_passes_only_rows_float32_cannot_decide():
    from oracle import adjudicate as J
    scale = 10.0
    L64 = _row({0: scale, 1: scale - 2e-6})                   # float64 gap 2e-6 < 2^-20 x 10 = 9.5e-6
    L32 = L64.clone(); L32[0, 0] += 3e-6                       # float32 oracle: id 0
    Lh = L64.clone(); Lh[0, 1] += 2.5e-6                         # HIP: id 1
    lines = J.adjudicate("t", [(0, 1, 0)], L64, L32, Lh, [7])
    assert len(lines) == 1 and "float64 gap 2.00e-06" in lines[0]
    with pytest.raises(AssertionError, match="float32 decides"):       # the same exchange with a gap float32 resolves
        big = _row({0: scale, 1: scale - 1e-4})
        J.adjudicate("t", [(0, 1, 0)], big, big.clone(), big.clone(), [7])
    with pytest.raises(AssertionError, match="top-2 exchange"):        # HIP picked something that is not the float64 runner-up
        J.adjudicate("t", [(0, 2, 0)], L64, L32, Lh, [7])
    with pytest.raises(AssertionError, match="too far"):               # a near tie does not excuse wrong logits
        far = Lh.clone(); far[0, 3] += 1e-3
        J.adjudicate("t", [(0, 1, 0)], L64, L32, far, [7])
    with pytest.raises(AssertionError, match="twice as far"):          # float32-grade in absolute terms, but 3x the oracle's distance on the row
        worse = Lh.clone(); worse[0, 3] += 1.0e-5                      # oracle row distance 3e-6, HIP 1e-5 (absolute bar 3.8e-5)
        J.adjudicate("t", [(0, 1, 0)], L64, L32, worse, [7])
    with pytest.raises(AssertionError, match="utterance farther"):     # the row is fine, the rest of the utterance is not
        L64b, L32b, Lhb = (torch.cat([x, _row({0: 1.0})]) for x in (L64, L32, Lh))
        L32b[1, 2] += 1e-6; Lhb[1, 2] += 3e-6; Lhb[1, 3] -= 3e-6; Lhb[1, 4] += 3e-6
        J.adjudicate("t", [(0, 1, 0)], L64b, L32b, Lhb, [7])
    assert J.differing_rows([1, 2, 3], [1, 5, 3]) == [(1, 2, 5)]
    with pytest.raises(AssertionError):
        J.differing_rows([1, 2], [1, 2, 3])


def test_masked_columns_do_not_take_part():
    from oracle import adjudicate as J
    L64 = _row({0: 10.0, 1: 10.0 - 2e-6, 7: 50.0})             # column 7 (pad / unk) is masked: it neither wins nor sets the scale
    L32 = L64.clone(); L32[0, 0] += 3e-6
    Lh = L64.clone(); Lh[0, 1] += 2.5e-6
    assert len(J.adjudicate("t", [(0, 1, 0)], L64, L32, Lh, [7])) == 1
    assert isinstance(np.asarray(L64), np.ndarray)
