"""GPU parity of the fused Conformer feed-forward kernel (csrc/ffn.hip, what ss_batch_encoder_forward launches twice per layer
on packed batches) against a torch float64 restatement of FeedForwardModule.forward + the 0.5-residual wiring + final_layer_norm
(researches/chunk_unity/modules/conformer_layer.py:152-164, 254-312).  Exact-f32 MFMA: differences are summation order only."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

D = 256


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from streamspeech_amd import lib as L
    return L.load()


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def params(Fh, seed=0):
    return dict(ln_g=1 + rnd(D, seed=seed + 1, scale=0.1), ln_b=rnd(D, seed=seed + 2, scale=0.1),
                W1=rnd(Fh, D, seed=seed + 3, scale=D ** -0.5), b1=rnd(Fh, seed=seed + 4, scale=0.1),
                W2=rnd(D, Fh, seed=seed + 5, scale=Fh ** -0.5), b2=rnd(D, seed=seed + 6, scale=0.1),
                ln2_g=1 + rnd(D, seed=seed + 7, scale=0.1), ln2_b=rnd(D, seed=seed + 8, scale=0.1))


def reference(x, p, alpha, ln2):
    x = x.double()
    q = {k: v.double() for k, v in p.items()}
    h = F.layer_norm(x, (D,), q["ln_g"], q["ln_b"], 1e-5)
    y = x + alpha * (F.linear(F.silu(F.linear(h, q["W1"], q["b1"])), q["W2"], q["b2"]))
    return F.layer_norm(y, (D,), q["ln2_g"], q["ln2_b"], 1e-5) if ln2 else y


def run(lib, x, p, alpha=0.5, ln2=False, inplace=True, ldx=D):
    from streamspeech_amd import lib as L
    M, Fh = x.shape[0], p["W1"].shape[0]
    dp = {k: v.contiguous().cuda() for k, v in p.items()}
    dx = torch.zeros(M, ldx).cuda()
    dx[:, :D] = x.cuda()
    dy = dx if inplace else torch.full((M, ldx), float("nan")).cuda()
    L.check(lib.ss_op_ffn_fused(S(), P(dx), ldx, P(dy), ldx, P(dp["ln_g"]), P(dp["ln_b"]), P(dp["W1"]), P(dp["b1"]), P(dp["W2"]),
                                P(dp["b2"]), alpha, P(dp["ln2_g"]) if ln2 else None, P(dp["ln2_b"]) if ln2 else None, M, D, Fh),
            "ss_op_ffn_fused")
    torch.cuda.synchronize()
    return dy[:, :D].cpu()


@pytest.mark.parametrize("wm", [3, 4])
@pytest.mark.parametrize("M,Fh,grid,ln2", [(1, 2048, 0, False), (47, 2048, 0, True), (48, 2048, 0, False), (49, 2048, 5, True),
                                           (64, 2048, 1, False), (65, 64, 0, True), (300, 128, 0, False), (300, 2048, 7, True),
                                           (1000, 2048, 0, False), (4200, 2048, 0, True), (4200, 2048, 256, False),
                                           (4200, 2048, 37, True), (12000, 2048, 0, True), (20001, 2048, 0, False)])
def test_ffn_fused_vs_float64(lib, wm, M, Fh, grid, ln2):
    """Every geometry class: a tile owned by one workgroup (no hand-off), tiles split over 2..8 workgroups, workgroups that
    span several tiles (grid < tiles), ragged last tile, both tile heights, with and without the trailing LayerNorm."""
    p = params(Fh, seed=M % 97)
    x = rnd(M, D, seed=M + 1)
    assert lib.ss_debug_ffn(grid, wm, -1) == 0
    try:
        got = run(lib, x, p, ln2=ln2)
        got2 = run(lib, x, p, ln2=ln2, inplace=False, ldx=D + 8)
    finally:
        lib.ss_debug_ffn(0, 0, -1)
    ref = reference(x, p, 0.5, ln2)
    assert torch.isfinite(got).all()
    err = (got.double() - ref).abs().max().item()
    assert err < 2e-5, f"max err {err}"
    assert torch.equal(got, got2)          # in place == out of place, padded leading dimension: bit-identical


def test_ffn_fused_is_bit_reproducible_under_concurrent_load(lib):
    """The hand-off is wait-free and sums in workgroup order: the same launch repeated next to another stream's full-chip
    kernels gives the same bits every time, and the arrival counters are back at zero after every launch."""
    M, Fh = 4200, 2048
    p = params(Fh, seed=3)
    x = rnd(M, D, seed=9)
    first = run(lib, x, p, ln2=True)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    for it in range(40):
        with torch.cuda.stream(side):
            for _ in range(3):
                a = (a @ a) * 1e-3
        again = run(lib, x, p, ln2=True)
        assert torch.equal(first, again), f"launch {it} differs"
    torch.cuda.synchronize()
    ref = reference(x, p, 0.5, True)
    assert (first.double() - ref).abs().max() < 2e-5


def test_batch_encoder_with_and_without_ffn_fusion(hip_model):
    """The packed-batch encoder three ways: the default pack-invariant form (whole-tile fused FFN, one-chain GEMMs), the round-4
    routes (stream-K fused FFN: ss_model_set_pack_invariant(0)) and those with two GEMM launches + LayerNorm per FFN
    (ss_debug_ffn(.., enable = 0)): same rows to summation-order accuracy, identical CTC ids."""
    from streamspeech_amd import lib as L, synth
    lib = L.load()
    T = [1203, 900, 777, 640, 512, 300, 150, 83] * 4
    fb = torch.cat([torch.from_numpy(synth.synth_fbank(50 + i, t)) for i, t in enumerate(T)]).cuda()

    def run():
        enc, Tp = hip_model.batch_encoder_forward(fb, T)
        return enc.clone(), list(Tp), [hip_model.batch_ctc_greedy(h, enc, Tp) for h in (0, 1)]

    assert hip_model.pack_invariant()
    enc_c, Tp, ids_c = run()
    hip_model.set_pack_invariant(False)
    try:
        enc_f, Tp1, ids_f = run()
        assert lib.ss_debug_ffn(0, 0, 0) == 0
        try:
            enc_u, Tp2, ids_u = run()
        finally:
            lib.ss_debug_ffn(0, 0, 1)
    finally:
        hip_model.set_pack_invariant(True)
    assert Tp == Tp1 == Tp2 and sum(Tp) >= 1000
    for name, e in (("stream-K fused", enc_f), ("pack-invariant", enc_c)):
        err = (e - enc_u).abs().max().item()
        assert err < 5e-5, f"{name} vs two-launch encoder rows differ by {err}"
    assert ids_f == ids_u == ids_c
    n = L.load().ss_prof_num_classes()
    names = [L.load().ss_prof_class_name(c).decode() for c in range(n)]
    assert "ffn_fused<256,2048>" in names
