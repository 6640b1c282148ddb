#!/usr/bin/env python
"""Per-op distance from float64: the HIP kernel vs torch's CPU float32 evaluation of the same op on the same operands (random data of
the encoder's shapes).  Looks for the op that makes HIP encoder logits sit farther from float64 than the oracle's (VERDICT r5 #2;
tests/diagnostics/chain_ablation_cpu.py showed the linears' summation order is NOT it).  Test infrastructure (imports oracle/ and tests/)."""
import ctypes as C
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from streamspeech_amd import lib as L  # noqa: E402
from test_ops_gpu import P, S, _attn_ref, rnd, run_conv_gemm  # noqa: E402

lib = L.load()
res = {}


def report(name, hip, f32, f64):
    eh = float(((hip.double() - f64) ** 2).mean().sqrt())
    ec = float(((f32.double() - f64) ** 2).mean().sqrt())
    res[name] = {"rms_hip": eh, "rms_torch_cpu": ec, "ratio": round(eh / ec, 3), "max_hip": float((hip.double() - f64).abs().max()),
                 "max_torch_cpu": float((f32.double() - f64).abs().max())}
    print(name, res[name], file=sys.stderr, flush=True)


def main():
    torch.set_num_threads(8)
    lib.ss_debug_canon(1)
    # ---- rel-pos attention (encoder self-attention, offline: every key visible) ----
    for T, chunk in ((300, 0), (300, 8), (800, 0)):
        H = 4
        qkv, Pt = rnd(T, 768, seed=19), rnd(2 * T - 1, 256, seed=20)
        u, vb = rnd(256, seed=21) * 0.3, rnd(256, seed=22) * 0.3
        dqkv, dP, du, dv = qkv.cuda(), Pt.cuda(), u.cuda(), vb.cuda()
        out = torch.full((T, 256), float("nan"), device="cuda")
        L.check(lib.ss_op_attention(S(), P(dqkv), 768, C.c_void_p(dqkv.data_ptr() + 1024), 768, C.c_void_p(dqkv.data_ptr() + 2048), 768,
                                    P(out), 256, T, T, H, 0.125, 0, chunk, P(dP), 256, P(du), P(dv)), "attn")
        a = (qkv[:, :256], qkv[:, 256:512], qkv[:, 512:])
        report(f"relpos_attention T={T} chunk={chunk}", out.cpu(), _attn_ref(*a, H, 0.125, False, chunk, Pt, u, vb),
               _attn_ref(*(x.double() for x in a), H, 0.125, False, chunk, Pt.double(), u.double(), vb.double()))
    # ---- plain attention (decoders) ----
    Tq = Tk = 500
    q, k, v = rnd(Tq, 512, seed=23) * 0.3, rnd(Tk, 512, seed=24), rnd(Tk, 512, seed=25)
    dq, dk, dv = q.cuda(), k.cuda(), v.cuda()
    out = torch.full((Tq, 512), float("nan"), device="cuda")
    L.check(lib.ss_op_attention(S(), P(dq), 512, P(dk), 512, P(dv), 512, P(out), 512, Tq, Tk, 8, 1.0, 0, 0, None, 0, None, None), "attn")
    report("plain_attention 500x500", out.cpu(), _attn_ref(q, k, v, 8, 1.0, False, 0), _attn_ref(q.double(), k.double(), v.double(), 8, 1.0, False, 0))
    # ---- depthwise conv + BatchNorm(eval) + SiLU ----
    from oracle import streamspeech_oracle as O
    T, Cc, K = 400, 256, 31
    x, w = rnd(T, Cc, seed=26), rnd(Cc, 1, K, seed=27, scale=K ** -0.5)
    mean, var = rnd(Cc, seed=28) * 0.1, torch.rand(Cc, generator=torch.Generator().manual_seed(29)) + 0.5
    g, b = rnd(Cc, seed=30) * 0.1 + 1, rnd(Cc, seed=31) * 0.1

    def dw_ref(x, w, mean, var, g, b):
        y = O.chunk_causal_conv1d(x.t().contiguous(), w, None, 1, 999999, groups=Cc).t()
        return F.silu(F.batch_norm(y, mean, var, g, b, False, 0.0, 1e-5))
    dx, dwt = x.cuda(), w[:, 0, :].t().contiguous().cuda()
    dm, dvar, dg, db = mean.cuda(), var.cuda(), g.cuda(), b.cuda()
    out = torch.empty_like(dx)
    L.check(lib.ss_op_dwconv_bn_silu(S(), P(dx), Cc, P(out), Cc, P(dwt), K, P(dm), P(dvar), P(dg), P(db), 1e-5, T, Cc, 0), "dw")
    torch.cuda.synchronize()
    report("dwconv_bn_silu", out.cpu(), dw_ref(x, w, mean, var, g, b), dw_ref(*(t.double() for t in (x, w, mean, var, g, b))))
    # ---- LayerNorm ----
    M, D = 1000, 256
    x, g, b = rnd(M, D, seed=16) * 3 + 1, rnd(D, seed=17) * 0.1 + 1, rnd(D, seed=18) * 0.1
    dx, dg, db = x.cuda(), g.cuda(), b.cuda()
    dy = torch.empty_like(dx)
    L.check(lib.ss_op_layernorm(S(), P(dx), D, P(dy), D, P(dg), P(db), M, D, 1e-5), "ln")
    torch.cuda.synchronize()
    report("layernorm", dy.cpu(), F.layer_norm(x, (D,), g, b, 1e-5), F.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-5))
    # ---- fused FFN (canon: whole tiles), with and without the trailing LayerNorm ----
    from test_ffn_gpu import params, reference, run
    for ln2 in (False, True):
        M = 4000
        x = rnd(M, 256, seed=50)
        p = params(2048, seed=60)
        hip = run(lib, x, p, 0.5, ln2, inplace=False)
        h = F.layer_norm(x, (256,), p["ln_g"], p["ln_b"], 1e-5)
        y = x + 0.5 * F.linear(F.silu(F.linear(h, p["W1"], p["b1"])), p["W2"], p["b2"])
        y = F.layer_norm(y, (256,), p["ln2_g"], p["ln2_b"], 1e-5) if ln2 else y
        report(f"ffn_fused ln2={ln2}", hip, y, reference(x, p, 0.5, ln2))
    # ---- LN + pointwise conv 1 + GLU; LN + QKV; CTC head ----
    from streamspeech_amd.weights import conv_tap_major, glu_interleave
    from test_pack_invariance_gpu import _ln_linear
    M, K = 4000, 256
    X, g, b = rnd(M, K, seed=41), 1 + 0.1 * rnd(K, seed=42), 0.1 * rnd(K, seed=43)
    W, bias = rnd(512, K, seed=44, scale=K ** -0.5), rnd(512, seed=45, scale=0.1)
    hip = _ln_linear(lib, X, g, b, glu_interleave(W), glu_interleave(bias), M, 512, K, glu=1)
    report("ln + pw1 + GLU", hip, F.glu(F.linear(F.layer_norm(X, (K,), g, b, 1e-5), W, bias), dim=1),
           F.glu(F.linear(F.layer_norm(X.double(), (K,), g.double(), b.double(), 1e-5), W.double(), bias.double()), dim=1))
    W, bias = rnd(6000, K, seed=46, scale=K ** -0.5) * 4, rnd(6000, seed=47, scale=0.1)
    hip = _ln_linear(lib, X, g, b, W, bias, M, 6000, K)
    report("ln + CTC head (logit scale ~4)", hip, F.linear(F.layer_norm(X, (K,), g, b, 1e-5), W, bias),
           F.linear(F.layer_norm(X.double(), (K,), g.double(), b.double(), 1e-5), W.double(), bias.double()))
    # ---- subsampler convs (stride 2, k = 5, GLU) ----
    for cin, cout, T in ((80, 1024, 1200), (512, 512, 600)):
        x = rnd(T, cin, seed=5)
        w = rnd(cout, cin, 5, seed=6, scale=(cin * 5) ** -0.5)
        b = rnd(cout, seed=7, scale=0.1)
        f32 = F.glu(O.chunk_causal_conv1d(x.t().contiguous(), w, b, 2, 999999), dim=0).t()
        f64 = F.glu(O.chunk_causal_conv1d(x.double().t().contiguous(), w.double(), b.double(), 2, 999999), dim=0).t()
        hip = run_conv_gemm(lib, x, conv_tap_major(glu_interleave(w)), glu_interleave(b), f32.shape[0], cout, cin, taps=5, stride=2, pad=2, in_len=T, glu=1)
        report(f"subsampler conv {cin}->{cout} + GLU", hip, f32, f64)
    lib.ss_debug_canon(0)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
