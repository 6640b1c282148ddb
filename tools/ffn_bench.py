"""Fused Conformer FFN (csrc/ffn.hip) against the two-launch form it replaces (LayerNorm launch + 256 -> 2048 GEMM with SiLU +
2048 -> 256 GEMM with the 0.5-residual epilogue, as ss_batch_encoder_forward issued them in round 3), at packed row counts of the
bench's length buckets.  Prints us per FFN and algorithmic TFLOP/s (4 M D F FLOP).
    python tools/ffn_bench.py [rows ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import lib as L          # noqa: E402

D, F = 256, 2048


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def main():
    rows = [int(a) for a in sys.argv[1:]] or [800, 1600, 2400, 3300, 4200, 6000, 8000, 12000]
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc     # noqa: E731
    ln_g, ln_b = 1 + rn(D, sc=0.1), rn(D, sc=0.1)
    W1, b1, W2, b2 = rn(F, D, sc=D ** -0.5), rn(F, sc=0.1), rn(D, F, sc=F ** -0.5), rn(D, sc=0.1)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    print("rows | two-launch us (TF/s) | fused 48-row tiles us (TF/s) | fused 64-row tiles us (TF/s) | fused + trailing LN us")
    for M in rows:
        x0 = rn(M, D)
        x, h, ff = x0.clone(), torch.empty(M, D, device="cuda"), torch.empty(M, F, device="cuda")
        fl = 4.0 * M * D * F

        def two():
            assert lib.ss_op_layernorm(s, P(x), D, P(h), D, P(ln_g), P(ln_b), M, D, C.c_float(1e-5)) == 0
            assert lib.ss_op_conv_gemm(s, P(h), D, P(W1), P(b1), None, 0, None, 0, P(ff), F, M, F, D, 1, 1, 1, 0, M, 0, 0, 0.1, 1, 1.0, 0.0, 0) == 0
            assert lib.ss_op_conv_gemm(s, P(ff), F, P(W2), P(b2), P(x), D, None, 0, P(x), D, M, D, F, 1, 1, 1, 0, M, 0, 0, 0.1, 0, 0.5, 0.0, 0) == 0

        def fused(ln2=False):
            assert lib.ss_op_ffn_fused(s, P(x), D, P(x), D, P(ln_g), P(ln_b), P(W1), P(b1), P(W2), P(b2), 0.5,
                                       P(ln_g) if ln2 else None, P(ln_b) if ln2 else None, M, D, F) == 0

        t2 = timed(two)
        out = [f"{M:6d} | {t2:7.1f} ({fl / t2 * 1e-6:5.1f})"]
        for wm in (3, 4):
            lib.ss_debug_ffn(0, wm, -1)
            x.copy_(x0)
            t = timed(fused)
            out.append(f"{t:7.1f} ({fl / t * 1e-6:5.1f})")
        lib.ss_debug_ffn(0, 0, -1)
        out.append(f"{timed(lambda: fused(True)):7.1f}")
        print(" | ".join(out), flush=True)


if __name__ == "__main__":
    main()
