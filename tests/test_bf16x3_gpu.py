"""Opt-in split-bf16 (3 bf16 MFMAs per k-slice) contraction of the C >= 64 vocoder convs (ss_vocoder_set_bf16x3,
conv_sk2_kernel<BN, LRELU, X3 = true>): NEVER the default path.  Bars: waveform RMS <= 1e-3 against
the FP32 CPU oracle, durations identical, and the default handle stays exact f32 (bit-identical to before the switch)."""
import ctypes as C

import pytest
import torch

from test_ops_gpu import lib, rnd, run_conv_gemm  # noqa: F401  (fixture + helpers)

pytestmark = pytest.mark.gpu
WAV_RMS_TOL = 1e-3


def _class_launches(lib, name):
    for c in range(lib.ss_prof_num_classes()):
        if lib.ss_prof_class_name(c).decode() == name:
            fl, by, n = C.c_double(), C.c_double(), C.c_int64()
            lib.ss_prof_totals(c, C.byref(fl), C.byref(by), C.byref(n))
            return int(n.value)
    raise KeyError(name)


@pytest.mark.parametrize("M,N,Cin,taps,dil,in_act", [(5000, 256, 256, 11, 5, 0), (9000, 64, 64, 7, 3, 0), (3000, 128, 128, 3, 1, 3),
                                                     (20011, 128, 128, 3, 1, 0), (257, 128, 64, 1, 1, 3), (5001, 64, 64, 11, 1, 3)])
def test_split_bf16_conv_close_to_f32(lib, M, N, Cin, taps, dil, in_act):
    """Same launch through the f32 kernel (force code 4) and its split-bf16 twin (code 5): operands carry 16 significant
    bits, so the relative RMS difference is ~3e-6 (bar 2e-5) -- and not zero, i.e. the twin really ran; deterministic."""
    from streamspeech_amd.weights import conv_tap_major
    A = rnd(M, Cin, seed=21)
    W = rnd(N, Cin, taps, seed=22, scale=(Cin * taps) ** -0.5)
    b, R = rnd(N, seed=23, scale=0.1), rnd(M, N, seed=24)
    Wp = conv_tap_major(W) if taps > 1 else W.view(N, Cin)
    kw = dict(taps=taps, dil=dil, pad=dil * (taps - 1) // 2, in_act=in_act, slope=0.1, R=R)
    n0 = _class_launches(lib, "conv_sk2_bf16x3<256,128,32>")
    try:
        lib.ss_debug_force_tile(4, 0, 0)
        ref = run_conv_gemm(lib, A, Wp, b, M, N, Cin, **kw)
        lib.ss_debug_force_tile(5, 0, 0)
        got = run_conv_gemm(lib, A, Wp, b, M, N, Cin, **kw)
        got2 = run_conv_gemm(lib, A, Wp, b, M, N, Cin, **kw)
    finally:
        lib.ss_debug_force_tile(0, 0, 0)
    assert _class_launches(lib, "conv_sk2_bf16x3<256,128,32>") == n0 + 2
    assert lib.ss_debug_sk_errors() == 0
    assert torch.isfinite(got).all() and torch.equal(got, got2)
    rel = float(((got - ref).double().pow(2).mean() / ref.double().pow(2).mean()).sqrt())
    assert 0.0 < rel < 2e-5, rel
    assert float((got - ref).abs().max()) < 1e-3


def test_bf16x3_vocoder_batch_vs_oracle_and_f32(hip_vocoder, synth_weights):
    """A ragged batch big enough for the stream-K dispatch (8 utterances, 60..170 units): the split-bf16 handle gives the
    same durations as the f32 handle and the oracle, its waveforms are within 1e-3 RMS of the CPU oracle and within 1e-4
    of the f32 handle (but not identical: the bf16 kernels really ran); switching it off restores bit-exact f32."""
    from oracle import streamspeech_oracle as O
    from streamspeech_amd import synth
    _, vcfg, _, vsd = synth_weights
    lens = [60, 170, 95, 130, 150, 77, 110, 165]
    codes = [[int(c) for c in synth.uniform(5, f"x3_codes_{i}", (n,), 0, 1000)] for i, n in enumerate(lens)]
    f32 = hip_vocoder.new_context()
    x3 = hip_vocoder.new_context()
    x3.set_bf16x3(True)
    lib_ = x3.lib
    n0 = _class_launches(lib_, "conv_sk2_bf16x3<256,128,32>")
    w_ref, d_ref, _ = f32.batch_forward(codes, True)
    w_x3, d_x3, _ = x3.batch_forward(codes, True)
    torch.cuda.synchronize()
    assert _class_launches(lib_, "conv_sk2_bf16x3<256,128,32>") > n0, "the batch must be large enough to reach conv_sk2"
    assert d_x3.cpu().tolist() == d_ref.cpu().tolist()
    for a, b in zip(w_x3, w_ref):
        assert a.shape == b.shape
        rms = float(torch.sqrt(torch.mean((a - b) ** 2)))
        assert 0.0 < rms < 1e-4, rms
    off = 0
    for i in (1, 5):                                      # two utterances against the CPU oracle
        off = sum(lens[:i])
        rw, rd = O.vocoder_forward(vsd, codes[i], vcfg, True)
        assert d_x3.cpu().tolist()[off: off + lens[i]] == rd.tolist()
        rms = float(torch.sqrt(torch.mean((w_x3[i].cpu() - rw) ** 2)))
        assert rms < WAV_RMS_TOL, f"rms {rms}"
    x3.set_bf16x3(False)
    w_back, _, _ = x3.batch_forward(codes, True)
    for a, b in zip(w_back, w_ref):
        assert torch.equal(a, b), "bf16x3 off must be the exact f32 path again"
