"""CPU checks of the host logic: weight-layout transforms (GLU interleave, tap-major conv,
ConvTranspose polyphase, q-scale folding) against the oracle through a torch emulation of the
conv-GEMM kernel's index semantics, plus the C-ABI library load / symbol export check.
No GPU compute is issued here.
"""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import streamspeech_oracle as O
from streamspeech_amd import synth, weights as W
from streamspeech_amd.config import ModelConfig, VocoderConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def emulate_conv_gemm(A, Wp, bias, M, N, Cin, taps=1, dil=1, stride=1, pad=0, chunk=0, in_slope=None, glu=False):
    """Index-for-index torch emulation of csrc/gemm.hip (same K order j*Cin+c, same zero rules,
    same GLU block mapping)."""
    L = A.shape[0]
    m = torch.arange(M)
    cols = []
    for j in range(taps):
        rin = m * stride + j * dil - pad
        ok = (rin >= 0) & (rin < L)
        if chunk > 0:
            ok &= rin < ((m * stride) // chunk + 1) * chunk
        a = A[rin.clamp(0, L - 1)] * ok[:, None]
        if in_slope is not None:
            a = torch.where(a > 0, a, a * in_slope)
        cols.append(a)
    acc = torch.cat(cols, 1) @ Wp.reshape(N, taps * Cin).t()
    if bias is not None:
        acc = acc + bias
    if glu:
        blk = acc.reshape(M, N // 32, 2, 16)
        return (blk[:, :, 0] * torch.sigmoid(blk[:, :, 1])).reshape(M, N // 2)
    return acc


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("chunk", [0, 8, 16])
def test_glu_interleave_and_tap_major(chunk):
    cin, cout, T = 80, 64, 53
    x, w, b = rnd(T, cin, seed=1), rnd(cout, cin, 5, seed=2, scale=0.05), rnd(cout, seed=3, scale=0.1)
    ref = F.glu(O.chunk_causal_conv1d(x.t().contiguous(), w, b, 2, chunk if chunk else 999999), dim=0).t()
    got = emulate_conv_gemm(x, W.conv_tap_major(W.glu_interleave(w)), W.glu_interleave(b), ref.shape[0], cout, cin,
                            taps=5, stride=2, pad=2, chunk=chunk, glu=True)
    assert (got - ref).abs().max() < 1e-5


@pytest.mark.parametrize("k,s", [(11, 5), (8, 4), (4, 2)])
def test_conv_transpose_polyphase(k, s):
    cin, cout, T = 32, 16, 41
    x, w, b = rnd(T, cin, seed=4), rnd(cin, cout, k, seed=5, scale=0.1), rnd(cout, seed=6, scale=0.1)
    ref = F.conv_transpose1d(F.leaky_relu(x.t()[None], 0.1), w, b, stride=s, padding=(k - s) // 2)[0].t()
    wp, bp = W.convT_polyphase(w, b, s)
    # every kernel tap must appear exactly once in the packed matrix
    assert torch.isclose(wp.abs().sum(), w.abs().sum(), rtol=1e-5)
    got = emulate_conv_gemm(x, wp, bp, T, s * cout, cin, taps=3, pad=1, in_slope=0.1).reshape(T * s, cout)
    assert got.shape == ref.shape and (got - ref).abs().max() < 1e-5


def test_pack_model_slots_and_transforms():
    cfg = ModelConfig()
    sd = synth.make_model_state_dict(0, cfg)
    names, offs, nums, blob = W.pack_model(sd, cfg, max_rel_pos=64)
    assert len(set(names)) == len(names)
    slot = {n: blob[o:o + k] for n, o, k in zip(names, offs, nums)}
    assert all(o % 64 == 0 for o in offs)
    # encoder.linear carries the sqrt(256) embed scale exactly
    assert torch.equal(slot["enc.linear.w"].view(256, 256), torch.from_numpy(sd["encoder.linear.weight"]) * 16.0)
    # q rows of the stacked fairseq projections are pre-scaled by 64^-0.5 (power of two: exact)
    q = torch.from_numpy(sd["decoder.layers.0.self_attn.q_proj.weight"])
    assert torch.equal(slot["unit.L0.self.qkv.w"].view(1536, 512)[:512], q * 0.125)
    # projected-position table slice used at run time equals the oracle's table for that T
    T = 21
    tab = slot["enc.pos_table"].view(2 * 64 - 1, 256)
    assert torch.equal(tab[64 - T: 64 - T + 2 * T - 1], O.rel_pos_table(T, 256))
    # sinusoid row added in the unit decoder (H2 quirk): row padding_idx + 1
    assert torch.equal(slot["unit.pos_row"], O.sinusoid_table(8, 512, 1)[2])
    assert torch.equal(slot["mt.pos_table"].view(-1, 512)[:50], O.sinusoid_table(1026, 512, 1)[:50])


def test_pack_vocoder_weight_norm_fold():
    vcfg = VocoderConfig()
    vsd = synth.make_vocoder_state_dict(0, vcfg)
    names, offs, nums, blob = W.pack_vocoder(vsd, vcfg)
    slot = {n: blob[o:o + k] for n, o, k in zip(names, offs, nums)}
    w = O.fold_weight_norm(O.SD(vsd), "resblocks.4.convs1.2")       # [128,128,7]
    assert torch.allclose(slot["voc.rb4.c1.2.w"].view(128, 7, 128), w.permute(0, 2, 1), atol=1e-7)
    assert slot["voc.up0.w"].numel() == 5 * 256 * 3 * 512
    assert slot["voc.post.w"].numel() == 7 * 16


def test_c_abi_library_exports_every_declared_symbol():
    """The C-ABI .so must load and export every function include/streamspeech_hip.h declares."""
    from streamspeech_amd import lib as L
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    header = open(os.path.join(ROOT, "include", "streamspeech_hip.h")).read()
    declared = set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(L.SIGNATURES.keys()), declared ^ set(L.SIGNATURES.keys())
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.ss_abi_version() == 2
    assert lib.ss_fbank_num_frames(16000) == 98
    assert lib.ss_encoder_out_len(83) == 21


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from streamspeech_amd import lib as L
    from streamspeech_amd.engine import HipModel
    with pytest.raises(L.StreamSpeechHipError):
        HipModel({}, ModelConfig())


def test_nothing_but_the_checkers_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's CPU-baseline legs may import it -- the product
    package, the tools and the job scripts must not (a product path that routes through the oracle would void every parity claim)."""
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b|from\s+\.+oracle\b)", re.M)
    offenders = []
    for top in ("streamspeech_amd", "tools", "fairseq_user_dir"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith(".py") and pat.search(open(os.path.join(dirpath, f)).read()):
                    offenders.append(os.path.relpath(os.path.join(dirpath, f), ROOT))
    assert not offenders, offenders
    # bench.py: every oracle import sits inside a function of the CPU-baseline legs
    src = open(os.path.join(ROOT, "bench.py")).read()
    allowed = ("cpu_baseline", "oracle_check", "streaming_measure")
    func = None
    for line in src.splitlines():
        m = re.match(r"def\s+(\w+)", line)
        if m:
            func = m.group(1)
        if pat.search(line):
            assert line.startswith(" ") and func in allowed, (func, line)
    # __graft_entry__: only smoke()
    func = None
    for line in open(os.path.join(ROOT, "__graft_entry__.py")).read().splitlines():
        m = re.match(r"def\s+(\w+)", line)
        if m:
            func = m.group(1)
        if pat.search(line):
            assert func == "smoke", (func, line)


def test_debug_force_tile_rejects_unknown_codes():
    """an unrecognised `bm` used to be stored as a forced tile and silently disabled the small-M / GEMV kernels."""
    from streamspeech_amd import lib as L
    lib = L.load()
    try:
        for bad in (7, 6000, -1, 16):
            assert lib.ss_debug_force_tile(bad, 0, 0) == 2          # SS_ERR_ARG
        for ok in (1, 2, 3, 4, 5, 6, 32, 64, 128, 0):
            assert lib.ss_debug_force_tile(ok, 0, 0) == 0
    finally:
        lib.ss_debug_force_tile(0, 0, 0)
