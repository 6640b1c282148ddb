"""GPU parity of single HIP kernels (through the C ABI's ss_op_* entry points) against plain
torch-fp32 CPU references of the same op.  Tolerances are absolute on O(1)-scaled data; the
kernels accumulate in exact FP32 (v_mfma_f32_16x16x4_f32 / v_fma_f32), so differences are
summation-order only.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-4


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


def run_conv_gemm(lib, A, Wp, bias, M, N, Cin, taps=1, dil=1, stride=1, pad=0, in_len=None, chunk=0,
                  in_act=0, slope=0.1, act=0, alpha=1.0, div=0.0, glu=0, R=None, R2=None, out_cols=None):
    from streamspeech_amd import lib as L
    dev = "cuda:0"
    dA, dW = A.contiguous().to(dev), Wp.contiguous().to(dev)
    db = None if bias is None else bias.to(dev)
    dR = None if R is None else R.contiguous().to(dev)
    dR2 = None if R2 is None else R2.contiguous().to(dev)
    oc = out_cols or (N // 2 if glu else N)
    dC = torch.full((M, oc), float("nan"), device=dev)
    L.check(lib.ss_op_conv_gemm(S(), P(dA), A.shape[1], P(dW), P(db), P(dR), oc, P(dR2), oc, P(dC), oc, M, N, Cin,
                                taps, dil, stride, pad, in_len if in_len is not None else A.shape[0], chunk,
                                in_act, slope, act, alpha, div, glu), "ss_op_conv_gemm")
    torch.cuda.synchronize()
    return dC.cpu()


@pytest.mark.parametrize("M,N,K", [(125, 2048, 256), (125, 256, 2048), (1, 6000, 512), (7, 512, 512),
                                   (33, 1005, 512), (500, 768, 256), (3000, 512, 256), (20000, 128, 128),
                                   (40000, 32, 32), (80000, 16, 16), (19, 1, 128)])
def test_linear_shapes(lib, M, N, K):
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
    got = run_conv_gemm(lib, A, W, b, M, N, K)
    ref = F.linear(A, W, b)
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() < TOL, f"max err {(got - ref).abs().max()}"


@pytest.mark.parametrize("act", [1, 2])
def test_linear_epilogues(lib, act):
    M, N, K = 125, 2048, 256
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
    R = rnd(M, N, seed=4)
    got = run_conv_gemm(lib, A, W, b, M, N, K, act=act, alpha=0.5, R=R)
    y = F.linear(A, W, b)
    y = F.silu(y) if act == 1 else F.relu(y)
    ref = y * 0.5 + R
    assert (got - ref).abs().max() < TOL


@pytest.mark.parametrize("cin,cout,T,chunk", [(80, 1024, 83, 0), (80, 1024, 83, 8), (512, 512, 42, 16),
                                              (512, 512, 11, 8), (80, 1024, 435, 0)])
def test_subsampler_conv_glu(lib, cin, cout, T, chunk):
    """stride-2 k5 chunk-causal conv + GLU vs the oracle's closed form."""
    from oracle import streamspeech_oracle as O
    from streamspeech_amd.weights import conv_tap_major, glu_interleave
    x = rnd(T, cin, seed=5)
    w = rnd(cout, cin, 5, seed=6, scale=(cin * 5) ** -0.5)
    b = rnd(cout, seed=7, scale=0.1)
    ref = F.glu(O.chunk_causal_conv1d(x.t().contiguous(), w, b, 2, chunk if chunk else 999999), dim=0).t()
    M = ref.shape[0]
    got = run_conv_gemm(lib, x, conv_tap_major(glu_interleave(w)), glu_interleave(b), M, cout, cin, taps=5, stride=2,
                        pad=2, in_len=T, chunk=chunk, glu=1)
    assert (got - ref).abs().max() < TOL, f"{(got - ref).abs().max()}"


@pytest.mark.parametrize("M", [1, 15, 16, 40, 131, 192, 193, 700])
def test_pointwise_glu_linear(lib, M):
    """The conformer conv module's pointwise conv 1 + GLU (conformer_layer.py:94-119: 256 -> 512, no bias in the checkpoint; a
    bias here to cover the epilogue): the small-M kernel's GLU form (M <= 192 rows: single utterances, streaming tail rows)
    and the tile kernel's (beyond) against torch."""
    from streamspeech_amd.weights import conv_tap_major, glu_interleave
    cin, cout = 256, 512
    x = rnd(M, cin, seed=205)
    w = rnd(cout, cin, 1, seed=206, scale=cin ** -0.5)
    b = rnd(cout, seed=207, scale=0.1)
    ref = F.glu(x @ w[:, :, 0].t() + b, dim=1)
    got = run_conv_gemm(lib, x, conv_tap_major(glu_interleave(w)), glu_interleave(b), M, cout, cin, taps=1, glu=1)
    assert got.shape == ref.shape == (M, cout // 2)
    assert (got - ref).abs().max() < TOL, f"{(got - ref).abs().max()}"


@pytest.mark.parametrize("C,k,dil,T", [(256, 11, 5, 300), (128, 7, 3, 1000), (64, 3, 1, 700), (32, 7, 3, 900),
                                       (16, 11, 5, 2000), (16, 3, 1, 257), (512, 7, 1, 50)])
def test_resblock_conv(lib, C, k, dil, T):
    """leaky_relu -> dilated conv (+ residual, + MRF accumulate and mean)."""
    from streamspeech_amd.weights import conv_tap_major
    x = rnd(T, C, seed=8)
    w = rnd(C, C, k, seed=9, scale=(C * k) ** -0.5)
    b = rnd(C, seed=10, scale=0.1)
    R, R2 = rnd(T, C, seed=11), rnd(T, C, seed=12)
    y = F.conv1d(F.leaky_relu(x.t()[None], 0.1), w, b, dilation=dil, padding=dil * (k - 1) // 2)[0].t()
    got = run_conv_gemm(lib, x, conv_tap_major(w), b, T, C, C, taps=k, dil=dil, pad=dil * (k - 1) // 2, in_act=3)
    assert (got - y).abs().max() < TOL
    got = run_conv_gemm(lib, x, conv_tap_major(w), b, T, C, C, taps=k, dil=dil, pad=dil * (k - 1) // 2, in_act=3,
                        R=R, R2=R2, div=3.0)
    ref = (R2 + (y + R)) / 3
    assert (got - ref).abs().max() < TOL


@pytest.mark.parametrize("cin,k,s,T", [(512, 11, 5, 40), (256, 8, 4, 200), (128, 8, 4, 300), (64, 4, 2, 500),
                                       (32, 4, 2, 777)])
def test_conv_transpose_polyphase(lib, cin, k, s, T):
    from streamspeech_amd.weights import convT_polyphase
    cout = cin // 2
    x = rnd(T, cin, seed=13)
    w = rnd(cin, cout, k, seed=14, scale=(cin * k / s) ** -0.5)
    b = rnd(cout, seed=15, scale=0.1)
    ref = F.conv_transpose1d(F.leaky_relu(x.t()[None], 0.1), w, b, stride=s, padding=(k - s) // 2)[0].t()
    wp, bp = convT_polyphase(w, b, s)
    got = run_conv_gemm(lib, x, wp, bp, T, s * cout, cin, taps=3, pad=1, in_act=3).reshape(T * s, cout)
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < TOL


@pytest.mark.parametrize("D", [128, 256, 512])
def test_layernorm(lib, D):
    from streamspeech_amd import lib as L
    M = 131
    x, g, b = rnd(M, D, seed=16) * 3 + 1, rnd(D, seed=17) * 0.1 + 1, rnd(D, seed=18) * 0.1
    dx, dg, db = x.cuda(), g.cuda(), b.cuda()
    dy = torch.empty_like(dx)
    L.check(lib.ss_op_layernorm(S(), P(dx), D, P(dy), D, P(dg), P(db), M, D, 1e-5), "ln")
    ref = F.layer_norm(x, (D,), g, b, 1e-5)
    assert (dy.cpu() - ref).abs().max() < 2e-5


def _attn_ref(q, k, v, H, scale, causal, chunk, P=None, u=None, vb=None):
    Tq, Tk = q.shape[0], k.shape[0]
    qh = q.view(Tq, H, 64).transpose(0, 1)
    kh = k.view(Tk, H, 64).transpose(0, 1)
    vh = v.view(Tk, H, 64).transpose(0, 1)
    if P is not None:
        ph = P.view(-1, H, 64).transpose(0, 1)
        ac = torch.matmul(qh + u.view(H, 1, 64), kh.transpose(1, 2))
        bd = torch.matmul(qh + vb.view(H, 1, 64), ph.transpose(1, 2))
        i = torch.arange(Tq)[:, None]
        j = torch.arange(Tk)[None, :]
        bd = torch.gather(bd, 2, (j - i + Tk - 1).expand(H, Tq, Tk))
        s = (ac + bd) * scale
    else:
        s = torch.matmul(qh, kh.transpose(1, 2)) * scale
    i = torch.arange(Tq)[:, None]
    j = torch.arange(Tk)[None, :]
    mask = torch.zeros(Tq, Tk, dtype=torch.bool)
    if causal:
        mask |= j > i + (Tk - Tq)
    if chunk > 0:
        mask |= j >= (i // chunk + 1) * chunk
    s = s.masked_fill(mask[None], float("-inf"))
    return torch.matmul(torch.softmax(s, -1), vh).transpose(0, 1).reshape(Tq, H * 64)


@pytest.mark.parametrize("T,chunk", [(21, 0), (21, 8), (125, 0), (125, 16), (200, 8), (64, 0), (65, 24), (1, 0), (16, 8), (17, 16), (33, 8), (48, 0), (48, 24)])   # <= 48 rows: attention_relpos_q16_kernel
def test_relpos_attention(lib, T, chunk):
    from streamspeech_amd import lib as L
    H = 4
    qkv = rnd(T, 3 * 256, seed=19)
    Pt = rnd(2 * T - 1, 3 * 256, seed=20)     # padded row stride to exercise ldp
    u, vb = rnd(256, seed=21) * 0.3, rnd(256, seed=22) * 0.3
    dqkv, dP, du, dv = qkv.cuda(), Pt.cuda(), u.cuda(), vb.cuda()
    out = torch.full((T, 256), float("nan"), device="cuda")
    L.check(lib.ss_op_attention(S(), P(dqkv), 768, C.c_void_p(dqkv.data_ptr() + 256 * 4), 768,
                                C.c_void_p(dqkv.data_ptr() + 512 * 4), 768, P(out), 256, T, T, H, 0.125, 0, chunk,
                                C.c_void_p(dP.data_ptr() + 256 * 4), 768, P(du), P(dv)), "attn")
    ref = _attn_ref(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], H, 0.125, False, chunk,
                    Pt[:, 256:512].contiguous(), u, vb)
    err = (out.cpu() - ref).abs().max()
    assert err < 5e-5, f"{err}"


@pytest.mark.parametrize("T,chunk", [(65, 0), (200, 16), (449, 0), (700, 24), (1000, 0)])
def test_relpos_attention_key_split_equals_the_serial_key_walk(lib, T, chunk):
    """Single-utterance rel-pos attention: the key-split form (one workgroup per (query tile, key group, head), last arrival
    merges in group order) against the serial walk over the key tiles, for the launch heuristic and forced group sizes;
    repeated launches reuse the same counters (they must be back at zero) and reproduce their bits."""
    from streamspeech_amd import lib as L
    H = 4
    qkv = rnd(T, 3 * 256, seed=119)
    Pt = rnd(2 * T - 1, 256, seed=120)
    u, vb = rnd(256, seed=121) * 0.3, rnd(256, seed=122) * 0.3
    dqkv, dP, du, dv = qkv.cuda(), Pt.cuda(), u.cuda(), vb.cuda()

    def run():
        out = torch.full((T, 256), float("nan"), device="cuda")
        L.check(lib.ss_op_attention(S(), P(dqkv), 768, C.c_void_p(dqkv.data_ptr() + 256 * 4), 768,
                                    C.c_void_p(dqkv.data_ptr() + 512 * 4), 768, P(out), 256, T, T, H, 0.125, 0, chunk,
                                    P(dP), 256, P(du), P(dv)), "attn")
        return out
    try:
        lib.ss_debug_attention_split(-1)
        serial = run()
        ref = _attn_ref(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], H, 0.125, False, chunk, Pt, u, vb)
        assert (serial.cpu() - ref).abs().max() < 5e-5
        for mode in (0, 1, 3):
            lib.ss_debug_attention_split(mode)
            a = run()
            b = run()
            assert torch.equal(a, b), f"split mode {mode}: not reproducible / counters not reset"
            err = (a - serial).abs().max().item()
            assert err < 3e-6, f"split mode {mode}: {err}"
    finally:
        lib.ss_debug_attention_split(0)


@pytest.mark.parametrize("Tq,Tk,causal", [(130, 130, 1), (1, 37, 1), (50, 21, 0), (525, 525, 1), (9, 9, 0), (3, 40, 1)])
def test_plain_attention(lib, Tq, Tk, causal):
    from streamspeech_amd import lib as L
    H = 8
    q, k, v = rnd(Tq, 512, seed=23) * 0.3, rnd(Tk, 512, seed=24), rnd(Tk, 512, seed=25)
    dq, dk, dv = q.cuda(), k.cuda(), v.cuda()
    out = torch.full((Tq, 512), float("nan"), device="cuda")
    L.check(lib.ss_op_attention(S(), P(dq), 512, P(dk), 512, P(dv), 512, P(out), 512, Tq, Tk, H, 1.0, causal, 0,
                                None, 0, None, None), "attn")
    ref = _attn_ref(q, k, v, H, 1.0, bool(causal), 0)
    err = (out.cpu() - ref).abs().max()
    assert err < 5e-5, f"{err}"


@pytest.mark.parametrize("T,chunk", [(21, 0), (21, 8), (125, 16), (125, 0), (7, 8)])
def test_dwconv_bn_silu(lib, T, chunk):
    from oracle import streamspeech_oracle as O
    from streamspeech_amd import lib as L
    Cc, K = 256, 31
    x, w = rnd(T, Cc, seed=26), rnd(Cc, 1, K, seed=27, scale=K ** -0.5)
    mean, var = rnd(Cc, seed=28) * 0.1, torch.rand(Cc, generator=torch.Generator().manual_seed(29)) + 0.5
    g, b = rnd(Cc, seed=30) * 0.1 + 1, rnd(Cc, seed=31) * 0.1
    y = O.chunk_causal_conv1d(x.t().contiguous(), w, None, 1, chunk if chunk else 999999, groups=Cc).t()
    ref = F.silu((y - mean) / torch.sqrt(var + 1e-5) * g + b)
    dx, dwt = x.cuda(), w[:, 0, :].t().contiguous().cuda()
    dm, dvar, dg, db = mean.cuda(), var.cuda(), g.cuda(), b.cuda()
    out = torch.empty_like(dx)
    L.check(lib.ss_op_dwconv_bn_silu(S(), P(dx), Cc, P(out), Cc, P(dwt), K, P(dm), P(dvar), P(dg), P(db), 1e-5, T, Cc,
                                     chunk), "dw")
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max() < 2e-5


# ---- persistent stream-K conv kernel (conv_sk.hip) ----------------------------------------------
def _conv_ref(A, W, b, taps, dil, slope=None):
    x = A if slope is None else F.leaky_relu(A, slope)
    N, Cin = W.shape[0], A.shape[1]
    return F.conv1d(x.t()[None], W.view(N, Cin, taps), b, dilation=dil, padding=dil * (taps - 1) // 2)[0].t()


@pytest.mark.parametrize("M,N,Cin,taps,dil,G", [
    (1000, 256, 256, 11, 5, 0), (1000, 256, 256, 11, 5, 37), (1000, 256, 256, 3, 1, 512), (777, 128, 128, 7, 3, 200),
    (5001, 64, 64, 11, 1, 0), (5001, 64, 64, 3, 5, 97), (300, 1280, 512, 3, 1, 0), (129, 128, 64, 1, 1, 3),
    (40, 64, 32, 1, 1, 1), (9000, 256, 128, 7, 3, -8)])
def test_stream_k_conv_matches_torch(lib, M, N, Cin, taps, dil, G):
    """Every fix-up shape: tiles split over 2..many workgroups, ranges inside one tile, G = 1 (no split),
    ragged last M tile; leaky-ReLU input, bias, both residuals and the MRF division in the epilogue."""
    from streamspeech_amd.weights import conv_tap_major
    A = rnd(M, Cin, seed=11)
    W = rnd(N, Cin, taps, seed=12, scale=(Cin * taps) ** -0.5)
    b, R, R2 = rnd(N, seed=13, scale=0.1), rnd(M, N, seed=14), rnd(M, N, seed=15)
    Wp = conv_tap_major(W) if taps > 1 else W.view(N, Cin)
    lib.ss_debug_force_tile(1, 8 if G < 0 else 0, max(G, 0))     # G = -8: XCD tile grouping on
    try:
        got = run_conv_gemm(lib, A, Wp, b, M, N, Cin, taps=taps, dil=dil, pad=dil * (taps - 1) // 2, in_act=3, slope=0.1,
                            div=3.0, R=R, R2=R2)
        got2 = run_conv_gemm(lib, A, Wp, b, M, N, Cin, taps=taps, dil=dil, pad=dil * (taps - 1) // 2, in_act=3, slope=0.1,
                             div=3.0, R=R, R2=R2)
        plain = run_conv_gemm(lib, A, Wp, None, M, N, Cin, taps=taps, dil=dil, pad=dil * (taps - 1) // 2)
    finally:
        lib.ss_debug_force_tile(0, 0, 0)
    assert lib.ss_debug_sk_errors() == 0
    ref = (R2 + (_conv_ref(A, W.reshape(N, -1) if taps == 1 else W, b, taps, dil, 0.1) + R)) / 3.0
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() < TOL, f"max err {(got - ref).abs().max()}"
    assert torch.equal(got, got2), "stream-K must be deterministic run to run"
    assert (plain - _conv_ref(A, W.reshape(N, -1) if taps == 1 else W, None, taps, dil)).abs().max() < TOL



# ---- second-generation stream-K kernel (conv_sk2.hip): 256 x 128 tiles, wait-free two-contributor hand-off ----
@pytest.mark.parametrize("M,N,Cin,taps,dil,G", [
    (1000, 256, 256, 11, 5, 0), (1000, 256, 256, 11, 5, 7), (1000, 256, 256, 11, 5, 100), (1000, 256, 256, 3, 1, 8),
    (777, 128, 128, 7, 3, 3), (777, 128, 128, 7, 3, 41), (9000, 256, 128, 7, 3, 0), (9000, 256, 128, 7, 3, 71),
    (300, 1280, 512, 3, 1, 0), (257, 128, 64, 1, 1, 2), (40, 128, 32, 1, 1, 1), (20011, 128, 128, 3, 1, 0),
    (5000, 512, 256, 3, 1, 33), (3000, 256, 2048, 1, 1, 0), (5001, 64, 64, 11, 1, 0), (5001, 64, 64, 3, 5, 97),
    (9000, 64, 128, 7, 3, 0), (700, 192, 64, 3, 1, 5)])
def test_stream_k2_conv_matches_torch(lib, M, N, Cin, taps, dil, G):
    """Every hand-off shape of the 2nd-generation kernel: tiles split over 2..many workgroups, ranges inside one tile,
    G = 1 (no split), ragged last M tile, 1..88 k-steps per part (staging pipeline running across part boundaries,
    one-step parts); leaky-ReLU input, bias, both residuals and the MRF division in the epilogue."""
    from streamspeech_amd.weights import conv_tap_major
    A = rnd(M, Cin, seed=11)
    W = rnd(N, Cin, taps, seed=12, scale=(Cin * taps) ** -0.5)
    b, R, R2 = rnd(N, seed=13, scale=0.1), rnd(M, N, seed=14), rnd(M, N, seed=15)
    Wp = conv_tap_major(W) if taps > 1 else W.view(N, Cin)
    lib.ss_debug_force_tile(4, 0, G)
    try:
        got = run_conv_gemm(lib, A, Wp, b, M, N, Cin, taps=taps, dil=dil, pad=dil * (taps - 1) // 2, in_act=3, slope=0.1,
                            div=3.0, R=R, R2=R2)
        got2 = run_conv_gemm(lib, A, Wp, b, M, N, Cin, taps=taps, dil=dil, pad=dil * (taps - 1) // 2, in_act=3, slope=0.1,
                             div=3.0, R=R, R2=R2)
        plain = run_conv_gemm(lib, A, Wp, None, M, N, Cin, taps=taps, dil=dil, pad=dil * (taps - 1) // 2)
    finally:
        lib.ss_debug_force_tile(0, 0, 0)
    assert lib.ss_debug_sk_errors() == 0
    ref = (R2 + (_conv_ref(A, W.reshape(N, -1) if taps == 1 else W, b, taps, dil, 0.1) + R)) / 3.0
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() < TOL, f"max err {(got - ref).abs().max()}"
    assert torch.equal(got, got2), "stream-K must be deterministic run to run"
    assert (plain - _conv_ref(A, W.reshape(N, -1) if taps == 1 else W, None, taps, dil)).abs().max() < TOL


def test_stream_k2_handoff_is_reproducible_under_load(lib):
    """Workspace slots and hand-off words are reused by every launch (partials travel as sc1 stores/loads past the
    non-coherent per-XCD L2s; the words carry a launch epoch).  Alternate two different problems of the same shape
    150 times on one stream while a second stream keeps the chip busy with another stream-K conv: every launch must
    reproduce its own first result bit for bit, whoever arrives first at each shared tile."""
    from streamspeech_amd.weights import conv_tap_major
    M, N, Cin, taps, dil = 8200, 256, 128, 7, 3
    pad = dil * (taps - 1) // 2
    W = rnd(N, Cin, taps, seed=41, scale=(Cin * taps) ** -0.5)
    Wp = conv_tap_major(W).contiguous().cuda()
    A = [rnd(M, Cin, seed=42 + i).cuda() for i in range(2)]
    out = [torch.empty(M, N, device="cuda") for _ in range(2)]
    side = torch.cuda.Stream()
    sA, sO = rnd(3000, Cin, seed=50).cuda(), torch.empty(3000, N, device="cuda")
    first = [None, None]
    torch.cuda.synchronize()
    lib.ss_debug_force_tile(4, 0, 0)
    try:
        for it in range(150):
            i = it & 1
            out[i].fill_(float("nan"))
            with torch.cuda.stream(side):        # uneven background load on another stream (own workspace)
                assert lib.ss_op_conv_gemm(S(), P(sA), Cin, P(Wp), None, None, N, None, N, P(sO), N, 3000, N, Cin, taps, dil, 1, pad,
                                           3000, 0, 0, 0.1, 0, 1.0, 0.0, 0) == 0
            assert lib.ss_op_conv_gemm(S(), P(A[i]), Cin, P(Wp), None, None, N, None, N, P(out[i]), N, M, N, Cin, taps, dil, 1, pad,
                                       M, 0, 0, 0.1, 0, 1.0, 0.0, 0) == 0
            if first[i] is None:
                torch.cuda.synchronize()
                first[i] = out[i].clone()
            elif it % 10 < 2 or it > 140:
                torch.cuda.synchronize()
                assert torch.equal(out[i], first[i]), f"launch {it} differs"
    finally:
        lib.ss_debug_force_tile(0, 0, 0)
    torch.cuda.synchronize()
    assert lib.ss_debug_sk_errors() == 0
    ref = _conv_ref(A[0].cpu(), W, None, taps, dil)
    assert (first[0].cpu() - ref).abs().max() < TOL


# ---- slab conv for the narrow vocoder stages (conv_slab.hip) -------------------------------------
@pytest.mark.parametrize("M,C,N,taps,dil,lrelu", [
    (5000, 32, 32, 11, 5, True), (5000, 32, 32, 3, 1, False), (4099, 16, 16, 11, 5, True), (2048, 16, 16, 7, 3, True),
    (3001, 32, 32, 3, 1, True), (2500, 16, 16, 3, 5, False), (70000, 16, 16, 7, 1, True)])
def test_slab_conv_matches_torch(lib, M, C, N, taps, dil, lrelu):
    """Default dispatch routes these (C, N in {16,32}, same-length) to conv_slab; all epilogue options on."""
    from streamspeech_amd.weights import conv_tap_major
    A = rnd(M, C, seed=21)
    W = rnd(N, C, taps, seed=22, scale=(C * taps) ** -0.5)
    b, R, R2 = rnd(N, seed=23, scale=0.1), rnd(M, N, seed=24), rnd(M, N, seed=25)
    Wp = conv_tap_major(W)
    pad = dil * (taps - 1) // 2
    got = run_conv_gemm(lib, A, Wp, b, M, N, C, taps=taps, dil=dil, pad=pad, in_act=3 if lrelu else 0, slope=0.1,
                        div=3.0, R=R, R2=R2)
    ref = (R2 + (_conv_ref(A, W, b, taps, dil, 0.1 if lrelu else None) + R)) / 3.0
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() < TOL, f"max err {(got - ref).abs().max()}"
    lib.ss_debug_force_tile(2, 0, 0)       # same call on the LDS-tiled kernel: results agree to rounding
    try:
        old = run_conv_gemm(lib, A, Wp, b, M, N, C, taps=taps, dil=dil, pad=pad, in_act=3 if lrelu else 0, slope=0.1,
                            div=3.0, R=R, R2=R2)
    finally:
        lib.ss_debug_force_tile(0, 0, 0)
    assert (got - old).abs().max() < 1e-5
    plain = run_conv_gemm(lib, A, Wp, None, M, N, C, taps=taps, dil=dil, pad=pad, act=3)   # LRELU epilogue, no bias
    assert (plain - F.leaky_relu(_conv_ref(A, W, None, taps, dil), 0.1)).abs().max() < TOL


@pytest.mark.parametrize("M,N,K,act", [(1, 512, 512, 0), (1, 2048, 512, 2), (1, 512, 2048, 0), (1, 6000, 512, 0), (1, 1536, 512, 0),
                                       (2, 512, 2048, 1), (4, 1005, 512, 0), (3, 512, 256, 0), (1, 512, 768, 0)])
def test_gemv_decode_shapes(lib, M, N, K, act):
    """M <= 4 plain linears route to the GEMV kernel (1 / 2 / 4 waves per column); residual + activation fused."""
    A, W, b, R = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=K ** -0.5), rnd(N, seed=33, scale=0.1), rnd(M, N, seed=34)
    got = run_conv_gemm(lib, A, W, b, M, N, K, act=act, alpha=0.5, R=R)
    y = F.linear(A, W, b)
    y = F.silu(y) if act == 1 else F.relu(y) if act == 2 else y
    ref = 0.5 * y + R
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() < TOL, f"max err {(got - ref).abs().max()}"


def test_stream_k_fixup_never_reads_stale_partials(lib):
    """The fix-up workspace slots are reused by every launch (partials travel as sc1 stores/loads past the
    non-coherent per-XCD L2s).  Alternate two different problems of the same shape 150 times on one stream:
    every launch must reproduce its own first result bit for bit."""
    from streamspeech_amd.weights import conv_tap_major
    M, N, Cin, taps, dil = 4100, 256, 128, 7, 3
    pad = dil * (taps - 1) // 2
    W = rnd(N, Cin, taps, seed=41, scale=(Cin * taps) ** -0.5)
    Wp = conv_tap_major(W).contiguous().cuda()
    A = [rnd(M, Cin, seed=42 + i).cuda() for i in range(2)]
    out = [torch.empty(M, N, device="cuda") for _ in range(2)]
    first = [None, None]
    lib.ss_debug_force_tile(1, 0, 0)
    try:
        for it in range(150):
            i = it & 1
            out[i].fill_(float("nan"))
            L_check = lib.ss_op_conv_gemm(S(), P(A[i]), Cin, P(Wp), None, None, N, None, N, P(out[i]), N, M, N, Cin, taps, dil, 1, pad,
                                          M, 0, 0, 0.1, 0, 1.0, 0.0, 0)
            assert L_check == 0
            if first[i] is None:
                torch.cuda.synchronize()
                first[i] = out[i].clone()
            elif it % 10 < 2 or it > 140:
                torch.cuda.synchronize()
                assert torch.equal(out[i], first[i]), f"launch {it} differs"
    finally:
        lib.ss_debug_force_tile(0, 0, 0)
    torch.cuda.synchronize()
    assert lib.ss_debug_sk_errors() == 0
    ref = _conv_ref(A[0].cpu(), W, None, taps, dil)
    assert (first[0].cpu() - ref).abs().max() < TOL


@pytest.mark.parametrize("C_", [64, 32, 16])
@pytest.mark.parametrize("taps,dil,M,lrelu,extras", [(3, 1, 70000, True, True), (3, 5, 66000, True, False), (7, 3, 70077, True, True),
                                                      (11, 5, 66001, True, True), (11, 1, 70000, False, False), (3, 1, 65536, False, True),
                                                      (2, 1, 66000, True, False), (7, 5, 70001, True, True), (11, 3, 69999, True, False),
                                                      (7, 1, 66003, False, True)])
def test_conv_c64_slab_kernel(lib, C_, taps, dil, M, lrelu, extras):
    """csrc/conv_c64.hip (the 64-channel vocoder stage of a packed batch: input slab once into LDS with the leaky-ReLU applied on
    the way, weight fragments streamed from L2): dilated "same" convs with bias / residual / MRF accumulate / mean against torch,
    and against the stream-K kernel it replaces (ss_debug_conv_c64(0)); both block heights (k = 11 at dilation 5 -> 192 rows)."""
    from streamspeech_amd.weights import conv_tap_major
    C = C_
    if C == 16:
        if taps == 2:
            pytest.skip("conv_c16 keeps the weight matrix in registers: k = 3 / 7 / 11 only")
        M = 2 * M                # its dispatch threshold is 131072 rows
    cls = {64: "conv_c64<256,64>", 32: "conv_c32<256,32>", 16: "conv_c16<256,16>"}[C]
    dbg = {64: lib.ss_debug_conv_c64, 32: lib.ss_debug_conv_c32, 16: lib.ss_debug_conv_c16}[C]
    x = rnd(M, C, seed=31)
    w = rnd(C, C, taps, seed=32, scale=(C * taps) ** -0.5)
    b = rnd(C, seed=33, scale=0.1)
    R, R2 = (rnd(M, C, seed=34), rnd(M, C, seed=35)) if extras else (None, None)
    pad = dil * (taps - 1) // 2
    xin = F.leaky_relu(x, 0.1) if lrelu else x
    xin_p = F.pad(xin.t()[None].double(), (pad, dil * (taps - 1) - pad))          # even tap counts: the extra row goes to the right
    y = F.conv1d(xin_p, w.double(), b.double(), dilation=dil)[0].t()
    ref = ((R2.double() + (y + R.double())) / 3.0) if extras else y
    kw = dict(taps=taps, dil=dil, pad=pad, in_act=3 if lrelu else 0, slope=0.1, R=R, R2=R2, div=3.0 if extras else 0.0)
    if C in (64, 32):
        dbg(4)                    # the direct form first; the Winograd form of the same launch (conv_c64w.hip) below
    n0 = _class_launches_ops(lib, cls)
    got = run_conv_gemm(lib, x, conv_tap_major(w), b, M, C, C, **kw)
    assert _class_launches_ops(lib, cls) == n0 + 1, "the slab kernel must have taken the launch"
    assert torch.isfinite(got).all()
    assert (got.double() - ref).abs().max() < TOL, f"max err {(got.double() - ref).abs().max()}"
    dbg(0)
    try:
        old = run_conv_gemm(lib, x, conv_tap_major(w), b, M, C, C, **kw)
    finally:
        dbg(1)
    assert _class_launches_ops(lib, cls) == n0 + 1
    assert (got - old).abs().max() < 2e-5
    if C in (64, 32):
        # Winograd F(2,3) on the dilation lattice: every k >= 3 conv whose slab fits two per CU (not k = 11 at dilation 5); pairs (t, t + d)
        # of 256- / 252- / 240-row blocks, ragged last block, residual / MRF epilogue on both rows of a pair.  32 channels: every k >= 3.
        takes = taps >= 3 and (C == 32 or (not (taps == 11 and dil == 5) and not (taps == 7 and dil == 5)))      # (64 channels, k = 7 at dilation 5: measured slower, stays direct)
        wcls = "conv_c64w<256,64>" if C == 64 else "conv_c32w<256,32>"
        dbg(5)
        nw = _class_launches_ops(lib, wcls)
        gotw = run_conv_gemm(lib, x, conv_tap_major(w), b, M, C, C, **kw)
        assert _class_launches_ops(lib, wcls) == nw + (1 if takes else 0)
        assert _class_launches_ops(lib, cls) == n0 + (1 if takes else 2)
        assert torch.isfinite(gotw).all()
        assert (gotw.double() - ref).abs().max() < TOL, f"max err {(gotw.double() - ref).abs().max()}"
        assert (gotw - got).abs().max() < 2e-5
        e_dir = ((got.double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()
        e_win = ((gotw.double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()
        assert e_win < 2.0 * e_dir + 1e-8, (float(e_dir), float(e_win))     # as close to float64 as the direct form (tools/winograd_error.py)


@pytest.mark.parametrize("taps,dil,M,extras", [(3, 1, 66000, True), (3, 5, 70001, False), (7, 3, 65536, True), (7, 5, 66003, True),
                                               (11, 1, 70000, False), (11, 3, 66001, True), (11, 5, 69999, True)])
def test_conv_c128_winograd_kernel(lib, taps, dil, M, extras):
    """csrc/conv_c64w.hip at 128 channels (the 128-channel vocoder stage of a packed batch: one 8-wave workgroup per CU, two column sets on one
    slab, Winograd F(2,3) on the dilation lattice, input leaky-ReLU while staging, optional pre-activated twin output): against torch
    float64, against the stream-K kernel it replaces (ss_debug_conv_c64(6)), and as close to float64 as that direct form."""
    from streamspeech_amd.weights import conv_tap_major
    C = 128
    x = rnd(M, C, seed=41)
    w = rnd(C, C, taps, seed=42, scale=(C * taps) ** -0.5)
    b = rnd(C, seed=43, scale=0.1)
    R, R2 = (rnd(M, C, seed=44), rnd(M, C, seed=45)) if extras else (None, None)
    pad = dil * (taps - 1) // 2
    xin = F.leaky_relu(x, 0.1)
    y = F.conv1d(F.pad(xin.t()[None].double(), (pad, pad)), w.double(), b.double(), dilation=dil)[0].t()
    ref = ((R2.double() + (y + R.double())) / 3.0) if extras else y
    kw = dict(taps=taps, dil=dil, pad=pad, in_act=3, slope=0.1, R=R, R2=R2, div=3.0 if extras else 0.0)
    lib.ss_debug_conv_c64(7)
    n0 = _class_launches_ops(lib, "conv_c128w<256,128>")
    got = run_conv_gemm(lib, x, conv_tap_major(w), b, M, C, C, **kw)
    assert _class_launches_ops(lib, "conv_c128w<256,128>") == n0 + 1, "the Winograd slab kernel must have taken the launch"
    assert torch.isfinite(got).all()
    assert (got.double() - ref).abs().max() < TOL, f"max err {(got.double() - ref).abs().max()}"
    lib.ss_debug_conv_c64(6)
    try:      # the stream-K kernel reads a pre-activated input
        old = run_conv_gemm(lib, xin.contiguous(), conv_tap_major(w), b, M, C, C, **dict(kw, in_act=0))
    finally:
        lib.ss_debug_conv_c64(7)
    assert _class_launches_ops(lib, "conv_c128w<256,128>") == n0 + 1
    assert (got - old).abs().max() < 2e-5
    e_dir = ((old.double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()
    e_win = ((got.double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()
    assert e_win < 2.0 * e_dir + 1e-8, (float(e_dir), float(e_win))


@pytest.mark.parametrize("taps,dil,M,extras", [(3, 1, 40000, True), (3, 5, 33001, False), (7, 3, 32768, True), (7, 5, 36003, True),
                                               (11, 1, 35000, False), (11, 3, 33001, True), (11, 5, 34999, True)])
def test_conv_c256_winograd_kernel(lib, taps, dil, M, extras):
    """csrc/conv_c64w.hip at 256 channels (round 5: the 256-channel vocoder stage of a packed batch -- the 128-channel form over two slab
    phases of 128 input channels and two column halves per block, workgroup pairs on one XCD): against torch float64, against the stream-K
    kernel it replaces (ss_debug_conv_c64(8)), and as close to float64 as that direct form; the pre-activated twin is exercised by the
    model-level tests (the up-conv that leaves the stage reads it)."""
    from streamspeech_amd.weights import conv_tap_major
    C = 256
    x = rnd(M, C, seed=41)
    w = rnd(C, C, taps, seed=42, scale=(C * taps) ** -0.5)
    b = rnd(C, seed=43, scale=0.1)
    R, R2 = (rnd(M, C, seed=44), rnd(M, C, seed=45)) if extras else (None, None)
    pad = dil * (taps - 1) // 2
    xin = F.leaky_relu(x, 0.1)
    y = F.conv1d(F.pad(xin.t()[None].double(), (pad, pad)), w.double(), b.double(), dilation=dil)[0].t()
    ref = ((R2.double() + (y + R.double())) / 3.0) if extras else y
    kw = dict(taps=taps, dil=dil, pad=pad, in_act=3, slope=0.1, R=R, R2=R2, div=3.0 if extras else 0.0)
    lib.ss_debug_conv_c64(9)
    n0 = _class_launches_ops(lib, "conv_c256w<256,128>")
    got = run_conv_gemm(lib, x, conv_tap_major(w), b, M, C, C, **kw)
    assert _class_launches_ops(lib, "conv_c256w<256,128>") == n0 + 1, "the Winograd slab kernel must have taken the launch"
    assert torch.isfinite(got).all()
    assert (got.double() - ref).abs().max() < TOL, f"max err {(got.double() - ref).abs().max()}"
    lib.ss_debug_conv_c64(8)
    try:      # the stream-K kernel reads a pre-activated input
        old = run_conv_gemm(lib, xin.contiguous(), conv_tap_major(w), b, M, C, C, **dict(kw, in_act=0))
    finally:
        lib.ss_debug_conv_c64(9)
    assert _class_launches_ops(lib, "conv_c256w<256,128>") == n0 + 1
    assert (got - old).abs().max() < 2e-5
    e_dir = ((old.double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()
    e_win = ((got.double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()
    assert e_win < 2.0 * e_dir + 1e-8, (float(e_dir), float(e_win))


def _class_launches_ops(lib, name):
    for c in range(lib.ss_prof_num_classes()):
        if lib.ss_prof_class_name(c).decode() == name:
            n = C.c_int64()
            lib.ss_prof_totals(c, None, None, C.byref(n))
            return n.value
    raise KeyError(name)
