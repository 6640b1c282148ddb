"""pack-invariant arithmetic upstream of every arg-max.

The reference decodes one utterance per call (agent/speech_to_speech.streamspeech.agent.py:425-478; agent/ctc_decoder.py:39-111),
so an utterance's ids cannot depend on what it is batched with.  The ragged-batch twins (ss_batch_*) therefore compute a packed
utterance with a summation order that is a function of that utterance alone.  Two levels:

* op level: the kernels the pack-invariant routes may choose between (LDS-tiled kernel without split-K, row-tile kernels at
  K = 256 and at K = 512 ... through 256-wide k-blocks; the whole-tile fused FFN at every tile height; the fixed small-M form of
  the decode rows) give a row the SAME BITS whatever the row count, i.e. whichever of them the row count selects -- and (round 6)
  sit no farther from float64 than torch's CPU sgemm: the chain is cut every 256 k, block sums added in ascending order;
* model level: logits of every arg-max stage of an utterance are bit-identical alone (a pack of one), in a pack of 8 and in
  another pack at another position.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from test_ops_gpu import P, S, lib, rnd, run_conv_gemm  # noqa: E402,F401  (lib: fixture)


@pytest.fixture()
def canon(lib):
    def set_mode(mode):
        assert lib.ss_debug_canon(mode) == 0
    yield set_mode
    lib.ss_debug_canon(0)


def _ln_linear(lib, X, g, b, W, bias, M, N, K, act=0, glu=0):
    from streamspeech_amd import lib as L
    dev = "cuda:0"
    dX, dg, db, dW = X[:M].contiguous().to(dev), g.to(dev), b.to(dev), W.contiguous().to(dev)
    dbias = None if bias is None else bias.to(dev)
    oc = N // 2 if glu else N
    dC = torch.full((M, oc), float("nan"), device=dev)
    L.check(lib.ss_op_ln_linear(S(), P(dX), K, P(dg), P(db), P(dW), P(dbias), None, 0, P(dC), oc, M, N, K, act, 1.0, glu), "ss_op_ln_linear")
    torch.cuda.synchronize()
    return dC.cpu()


@pytest.mark.parametrize("N,K,act,Ms", [
    (768, 256, 0, (9000, 300, 48, 5)),        # QKV: row-tile kernel at 9000 rows, LDS-tiled kernel below
    (256, 256, 0, (9000, 700, 17)),           # attention output / pointwise conv 2: 32x64 tiles, 32x32 tiles, 16-row tile
    (6000, 256, 0, (2500, 200, 3)),           # CTC vocabulary head
    (1536, 512, 0, (28000, 1100, 30)),        # unit-decoder QKV: stream-K on whole tiles at 28000 rows, LDS tiles below
    (2048, 512, 2, (28000, 1100, 30)),        # fc1 + ReLU
    (512, 2048, 0, (28000, 1100, 30)),        # fc2
    (1005, 512, 0, (5000, 100, 9)),           # unit vocabulary head (N % 16 != 0)
])
def test_one_chain_kernels_give_a_row_the_same_bits_at_every_row_count(lib, canon, N, K, act, Ms):
    canon(1)
    Mmax = max(Ms)
    A, W, b = rnd(Mmax, K, seed=31), rnd(N, K, seed=32, scale=K ** -0.5), rnd(N, seed=33, scale=0.1)
    R = rnd(Mmax, N, seed=34)
    full = run_conv_gemm(lib, A, W, b, Mmax, N, K, act=act, alpha=0.5, R=R)
    ref = F.linear(A[:64].double(), W.double(), b.double())
    ref = (F.relu(ref) if act == 2 else ref) * 0.5 + R[:64].double()
    assert (full[:64].double() - ref).abs().max() < 2e-4
    for M in Ms[1:]:
        for off in (0, Mmax - M):                                   # the same rows at another position of the launch
            part = run_conv_gemm(lib, A[off:off + M], W, b, M, N, K, act=act, alpha=0.5, R=R[off:off + M])
            assert torch.equal(part, full[off:off + M]), f"N={N} K={K}: rows differ between M={Mmax} and M={M} (offset {off})"


@pytest.mark.parametrize("N,K,M", [(512, 2048, 20000), (512, 2048, 300), (2048, 512, 20000), (256, 256, 4000), (768, 256, 9000)])
def test_blocked_chain_is_as_close_to_float64_as_torch_cpu_sgemm(lib, canon, N, K, M):
    """one sequential chain over K = 2048 sat 2.8x farther from float64 than the oracle's BLAS.  The pack-invariant
    routes now cut the chain every 256 k: RMS error against float64 at most 1.1x that of torch's CPU float32 matmul on the same
    operands (and a row's bits still do not depend on M: the test above)."""
    canon(1)
    A, W = rnd(M, K, seed=71), rnd(N, K, seed=72, scale=K ** -0.5)
    out = run_conv_gemm(lib, A, W, None, M, N, K)[:256].double()
    ref = A[:256].double() @ W.double().t()
    cpu = (A[:256] @ W.t()).double()
    e_hip, e_cpu = float(((out - ref) ** 2).mean().sqrt()), float(((cpu - ref) ** 2).mean().sqrt())
    print(f"N={N} K={K} M={M}: rms vs float64: HIP {e_hip:.3e}, torch CPU {e_cpu:.3e}, ratio {e_hip / e_cpu:.3f}")
    assert e_hip <= 1.1 * e_cpu


def test_subsampler_conv_blocked_chain_accuracy(lib, canon):
    """The encoder's second subsampling conv (Conv1dSubsampler, fairseq/models/speech_to_text/s2t_transformer.py: 1024 -> 512, k = 5,
    stride 2): K = 5120, the longest contraction upstream of the arg-maxes."""
    canon(1)
    Tin, Cin, N, taps = 1500, 1024, 512, 5
    A, W = rnd(Tin, Cin, seed=73), rnd(N, taps * Cin, seed=74, scale=(taps * Cin) ** -0.5)
    M = (Tin + 2 * 2 - taps) // 2 + 1
    out = run_conv_gemm(lib, A, W, None, M, N, Cin, taps=taps, stride=2, pad=2, in_len=Tin).double()
    x = A.double().t().unsqueeze(0)                                        # [1, Cin, T]
    w = W.double().view(N, taps, Cin).permute(0, 2, 1).contiguous()       # tap-major [N][k][Cin] -> [N, Cin, k]
    ref = F.conv1d(x, w, stride=2, padding=2)[0].t()
    cpu = F.conv1d(x.float(), w.float(), stride=2, padding=2)[0].t().double()
    e_hip, e_cpu = float(((out - ref) ** 2).mean().sqrt()), float(((cpu - ref) ** 2).mean().sqrt())
    print(f"subsampler conv2: rms vs float64: HIP {e_hip:.3e}, torch CPU {e_cpu:.3e}, ratio {e_hip / e_cpu:.3f}")
    assert e_hip <= 1.1 * e_cpu


@pytest.mark.parametrize("N,glu", [(768, 0), (512, 1)])
def test_layernorm_linear_pack_invariant(lib, canon, N, glu):
    from streamspeech_amd.weights import glu_interleave
    canon(1)
    K, Mmax = 256, 9000
    X, g, b = rnd(Mmax, K, seed=41), 1 + 0.1 * rnd(K, seed=42), 0.1 * rnd(K, seed=43)
    W, bias = rnd(N, K, seed=44, scale=K ** -0.5), rnd(N, seed=45, scale=0.1)
    if glu:
        W, bias = glu_interleave(W), glu_interleave(bias)
    full = _ln_linear(lib, X, g, b, W, bias, Mmax, N, K, glu=glu)
    for M in (700, 100, 3):
        part = _ln_linear(lib, X, g, b, W, bias, M, N, K, glu=glu)
        assert torch.equal(part, full[:M]), f"LayerNorm + linear N={N}: rows differ between M={Mmax} and M={M}"


def _ffn(lib, X, prm, M, ln2):
    from streamspeech_amd import lib as L
    dev = "cuda:0"
    d = [t.to(dev) for t in prm]
    dX = X[:M].contiguous().to(dev)
    dY = torch.full_like(dX, float("nan"))
    L.check(lib.ss_op_ffn_fused(S(), P(dX), 256, P(dY), 256, P(d[0]), P(d[1]), P(d[2]), P(d[3]), P(d[4]), P(d[5]), 0.5,
                                P(d[6]) if ln2 else None, P(d[7]) if ln2 else None, M, 256, 2048), "ss_op_ffn_fused")
    torch.cuda.synchronize()
    return dY.cpu()


@pytest.mark.parametrize("ln2", [False, True])
def test_fused_ffn_whole_tile_form_is_pack_invariant(lib, canon, ln2):
    """Every tile height (16 .. 64 rows), every grid and every row count give a row the same bits; the result is the FFN."""
    canon(1)
    D, Fd, Mmax = 256, 2048, 8000
    X = rnd(Mmax, D, seed=51)
    prm = [1 + 0.1 * rnd(D, seed=52), 0.1 * rnd(D, seed=53), rnd(Fd, D, seed=54, scale=D ** -0.5), rnd(Fd, seed=55, scale=0.1),
           rnd(D, Fd, seed=56, scale=Fd ** -0.5), rnd(D, seed=57, scale=0.1), 1 + 0.1 * rnd(D, seed=58), 0.1 * rnd(D, seed=59)]
    full = _ffn(lib, X, prm, Mmax, ln2)
    x64 = X[:64].double()
    h = F.layer_norm(x64, (D,), prm[0].double(), prm[1].double(), 1e-5)
    y = x64 + 0.5 * (F.silu(h @ prm[2].double().t() + prm[3].double()) @ prm[4].double().t() + prm[5].double())
    if ln2:
        y = F.layer_norm(y, (D,), prm[6].double(), prm[7].double(), 1e-5)
    assert (full[:64].double() - y).abs().max() < 2e-4
    try:
        for wm in (1, 2, 3, 4):
            for M, grid in ((Mmax, 0), (Mmax, 37), (1000, 0), (50, 0), (1, 0)):
                assert lib.ss_debug_ffn(grid, wm, -1) == 0
                part = _ffn(lib, X, prm, M, ln2)
                assert torch.equal(part, full[:M]), f"tile height {16 * wm}, M={M}, grid={grid}: rows differ"
    finally:
        lib.ss_debug_ffn(0, 0, -1)


@pytest.mark.parametrize("N,K,act", [(1536, 512, 0), (512, 2048, 0), (2048, 512, 2), (6000, 512, 0)])
def test_decode_rows_fixed_small_m_form(lib, canon, N, K, act):
    """The lock-step MT decode: one row per utterance; a row's bits must not depend on how many utterances decode with it."""
    canon(2)
    A, W, b = rnd(128, K, seed=61), rnd(N, K, seed=62, scale=K ** -0.5), rnd(N, seed=63, scale=0.1)
    full = run_conv_gemm(lib, A, W, b, 128, N, K, act=act)
    ref = F.linear(A.double(), W.double(), b.double())
    assert (full.double() - (F.relu(ref) if act == 2 else ref)).abs().max() < 2e-4
    for M in (64, 32, 5, 4, 1):
        part = run_conv_gemm(lib, A[:M], W, b, M, N, K, act=act)
        assert torch.equal(part, full[:M]), f"N={N} K={K}: decode rows differ between 128 and {M} utterances"


def _stages(m, utts, pcms):
    """The calls of workload.run_batch up to the unit logits; returns per-stage dense logits / states, split per utterance."""
    feat, T = m.batch_fbank_cmvn(torch.cat(pcms), [u.n_samples for u in utts])
    enc, Tp = m.batch_encoder_forward(feat, T)
    m.batch_ctc_greedy(0, enc, Tp)
    asr = m.last_logits().cpu()
    m.batch_ctc_greedy(1, enc, Tp)
    st = m.last_logits().cpu()
    toks, feats, n = m.batch_mt_greedy(enc, Tp, [u.n_mt for u in utts])
    m.batch_t2u_units(feats, n)
    unit = m.last_logits().cpu()
    up = m.cfg.ctc_upsample
    out, o1, o2 = [], 0, 0
    for b, u in enumerate(utts):
        out.append({"enc": enc[o1:o1 + Tp[b]].cpu(), "asr": asr[o1:o1 + Tp[b]], "st": st[o1:o1 + Tp[b]], "tok": list(toks[b]),
                    "mt": feats[b, :n[b]].cpu(), "unit": unit[o2:o2 + n[b] * up]})
        o1 += Tp[b]; o2 += n[b] * up
    return out


def test_every_argmax_stage_is_bitwise_pack_invariant(hip_model):
    from streamspeech_amd import synth, workload
    m = hip_model
    assert m.pack_invariant()
    allu = sorted(workload.make_utterances(64), key=lambda u: u.seconds)
    utts = [allu[0], allu[5], allu[20], allu[31], allu[40], allu[50], allu[60], allu[63]]        # 1 s ... 15 s
    pcms = [torch.from_numpy(synth.synth_pcm(1234 + u.idx, u.n_samples)).cuda() for u in utts]
    pack = _stages(m, utts, pcms)
    perm = [5, 2, 7, 0, 3]                                              # another pack: other companions, other positions
    other = _stages(m, [utts[i] for i in perm], [pcms[i] for i in perm])
    for j, i in enumerate(perm):
        for k in ("enc", "asr", "st", "mt", "unit"):
            assert torch.equal(other[j][k], pack[i][k]), f"utterance {i}: {k} differs between two packs"
        assert other[j]["tok"] == pack[i]["tok"]
    for i in (0, 3, 7):                                                 # alone
        alone = _stages(m, [utts[i]], [pcms[i]])[0]
        for k in ("enc", "asr", "st", "mt", "unit"):
            assert torch.equal(alone[k], pack[i][k]), f"utterance {i}: {k} differs between a pack of one and a pack of eight"
        assert alone["tok"] == pack[i]["tok"]


def test_predicted_durations_do_not_depend_on_the_pack(hip_vocoder):
    """a14 (agent/tts/codehifigan.py:56-95): durations are integers, round(exp(log-duration) - 1) clamped to >= 1 -- like an id, a
    duration must not depend on what an utterance is batched with.  The batched duration predictor's convs are one-chain launches."""
    import numpy as np
    rng = np.random.default_rng(7)
    codes = [[int(x) for x in rng.integers(0, 1000, size=n)] for n in (180, 37, 1, 555, 64, 90, 12, 300)]
    _, dur, _ = hip_vocoder.batch_forward(codes, dur_prediction=True)
    dur = dur.cpu().tolist()
    off = np.cumsum([0] + [len(c) for c in codes])
    for b in (0, 2, 3, 6):
        _, d1, _ = hip_vocoder.batch_forward([codes[b]], dur_prediction=True)
        assert d1.cpu().tolist() == dur[off[b]:off[b + 1]], f"utterance {b}: predicted durations differ alone vs in a pack of eight"
    perm = [3, 0, 7]
    _, d2, _ = hip_vocoder.batch_forward([codes[i] for i in perm], dur_prediction=True)
    d2 = d2.cpu().tolist()
    o2 = np.cumsum([0] + [len(codes[i]) for i in perm])
    for j, i in enumerate(perm):
        assert d2[o2[j]:o2[j + 1]] == dur[off[i]:off[i + 1]]
