"""GPU parity of the row-tile linear kernel (csrc/rtlin.hip): every K = 256 linear of more than 192 rows -- encoder QKV /
attention output / pointwise convs / input projection, CTC heads, MT cross K|V -- against torch float64, across epilogues, the
LayerNorm prologue, GLU, ragged last tiles and forced grids (units are independent: any grid must give the same bits), and against
the LDS-tiled kernel it replaces (ss_debug_rtlin(0, 0))."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
K = 256


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


def run(lib, x, W, b, ln=None, R=None, act=0, alpha=1.0, glu=0, inplace_r=False):
    from streamspeech_amd import lib as L
    M, N = x.shape[0], W.shape[0]
    oc = N // 2 if glu else N
    dx, dW = x.contiguous().cuda(), W.contiguous().cuda()
    db = None if b is None else b.cuda()
    dR = None if R is None else R.contiguous().cuda()
    dC = dR if inplace_r else torch.full((M, oc), float("nan")).cuda()
    if ln is None:
        rc = lib.ss_op_conv_gemm(S(), P(dx), K, P(dW), P(db), P(dR), oc, None, 0, P(dC), oc, M, N, K, 1, 1, 1, 0, M, 0, 0, 0.1, act,
                                 alpha, 0.0, glu)
    else:
        dg, dbt = ln[0].cuda(), ln[1].cuda()
        rc = lib.ss_op_ln_linear(S(), P(dx), K, P(dg), P(dbt), P(dW), P(db), P(dR), oc, P(dC), oc, M, N, K, act, alpha, glu)
    L.check(rc, "linear")
    torch.cuda.synchronize()
    return dC.cpu()


def reference(x, W, b, ln=None, R=None, act=0, alpha=1.0, glu=0):
    x, W = x.double(), W.double()
    if ln is not None:
        x = F.layer_norm(x, (K,), ln[0].double(), ln[1].double(), 1e-5)
    y = F.linear(x, W, None if b is None else b.double())
    if glu:
        n = W.shape[0]
        y = y.view(-1, n // 32, 2, 16)
        return (y[:, :, 0] * torch.sigmoid(y[:, :, 1])).reshape(-1, n // 2)      # [16 value | 16 gate] blocks (pack-time interleave)
    y = F.silu(y) if act == 1 else F.relu(y) if act == 2 else y
    y = y * alpha
    return y if R is None else y + R.double()


def class_launches(lib, name):
    for c in range(lib.ss_prof_num_classes()):
        if lib.ss_prof_class_name(c).decode() == name:
            n = C.c_int64()
            lib.ss_prof_totals(c, None, None, C.byref(n))
            return n.value
    raise KeyError(name)


@pytest.mark.parametrize("M,N,bias,ln,res,act,alpha,glu,grid", [
    (193, 768, True, True, False, 0, 1.0, 0, 60), (240, 256, True, False, True, 0, 1.0, 0, 20), (700, 512, False, True, False, 0, 1.0, 1, 256),
    (700, 512, True, False, False, 0, 1.0, 1, 3), (4200, 768, True, True, False, 0, 1.0, 0, 0), (4200, 256, True, False, True, 0, 0.5, 0, 700),
    (4200, 2048, True, True, False, 1, 1.0, 0, 0), (4200, 1024, True, False, False, 2, 1.0, 0, 97), (4173, 6000, True, False, False, 0, 1.0, 0, 0),
    (12001, 16, False, False, False, 0, 1.0, 0, 1), (12001, 512, False, True, False, 0, 1.0, 1, 0), (333, 6000, False, False, False, 0, 1.0, 0, 0)])
def test_row_tile_linear_vs_float64(lib, M, N, bias, ln, res, act, alpha, glu, grid):
    x = rnd(M, K, seed=M + N)
    W = rnd(N, K, seed=N + 1, scale=K ** -0.5)
    b = rnd(N, seed=N + 2, scale=0.1) if bias else None
    lnp = (1 + rnd(K, seed=5, scale=0.1), rnd(K, seed=6, scale=0.1)) if ln else None
    R = rnd(M, N, seed=7) if res else None
    n0 = class_launches(lib, "rt_linear<48,256>")
    assert lib.ss_debug_rtlin(grid, 1) == 0
    try:
        got = run(lib, x, W, b, lnp, R, act, alpha, glu, inplace_r=res)
        assert lib.ss_debug_rtlin(5 if grid != 5 else 11, 1) == 0
        again = run(lib, x, W, b, lnp, R, act, alpha, glu)
    finally:
        lib.ss_debug_rtlin(0, 1)
    assert class_launches(lib, "rt_linear<48,256>") == n0 + 2, "the row-tile kernel must have taken both launches"
    ref = reference(x, W, b, lnp, R, act, alpha, glu)
    assert got.shape == ref.shape and torch.isfinite(got).all()
    err = (got.double() - ref).abs().max().item()
    assert err < 2e-5, f"max err {err}"
    assert torch.equal(got, again), "units are independent: the grid must not change a bit"


def test_row_tile_linear_vs_the_tiled_kernel(lib):
    """Same launches on the LDS-tiled kernel (ss_debug_rtlin(0, 0)): agreement to summation order."""
    M, N = 4500, 768
    x, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
    R = rnd(M, N, seed=4)
    new = run(lib, x, W, b, None, R, 1, 0.5)
    n0 = class_launches(lib, "rt_linear<48,256>")
    lib.ss_debug_rtlin(0, 0)
    try:
        old = run(lib, x, W, b, None, R, 1, 0.5)
    finally:
        lib.ss_debug_rtlin(0, 1)
    assert class_launches(lib, "rt_linear<48,256>") == n0
    assert (new - old).abs().max() < 1e-5
