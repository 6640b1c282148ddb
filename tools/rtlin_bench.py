"""Row-tile linear kernel (csrc/rtlin.hip) against the LDS-tiled kernel it replaces, on the K = 256 linears of the packed-batch
encoder / heads: us per launch and algorithmic TFLOP/s, for 1 / 2 / 3 workgroups per CU.
    python tools/rtlin_bench.py [rows ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import lib as L          # noqa: E402

K = 256
# name, N, LayerNorm prologue, residual, GLU
SHAPES = [("qkv", 768, True, False, 0), ("attn_out", 256, False, True, 0), ("pw1_glu", 512, True, False, 1), ("pw2", 256, False, True, 0),
          ("cross_kv", 1024, False, False, 0), ("ctc_head", 6000, False, False, 0)]


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def main():
    rows = [int(a) for a in sys.argv[1:]] or [800, 2400, 4200, 8000, 12000]
    lib = L.load()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc     # noqa: E731
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ln_g, ln_b = 1 + rn(K, sc=0.1), rn(K, sc=0.1)

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

    print("rows shape | tiled kernel (+ LayerNorm launch) us (TF/s) | row-tile 1 / 2 / 3 workgroups per CU us (TF/s)")
    for M in rows:
        x, h = rn(M, K), torch.empty(M, K, device="cuda")
        for name, N, ln, res, glu in SHAPES:
            W, b = rn(N, K, sc=K ** -0.5), rn(N, sc=0.1)
            oc = N // 2 if glu else N
            out = rn(M, oc)
            fl = 2.0 * M * N * K

            def tiled():
                src = x
                if ln:
                    assert lib.ss_op_layernorm(s, P(x), K, P(h), K, P(ln_g), P(ln_b), M, K, C.c_float(1e-5)) == 0
                    src = h
                assert lib.ss_op_conv_gemm(s, P(src), K, P(W), P(b), P(out) if res else None, oc, None, 0, P(out), oc, M, N, K, 1, 1, 1, 0, M,
                                           0, 0, 0.1, 0, 1.0, 0.0, glu) == 0

            def rowtile():
                if ln:
                    assert lib.ss_op_ln_linear(s, P(x), K, P(ln_g), P(ln_b), P(W), P(b), P(out) if res else None, oc, P(out), oc, M, N, K, 0, 1.0, glu) == 0
                else:
                    assert lib.ss_op_conv_gemm(s, P(x), K, P(W), P(b), P(out) if res else None, oc, None, 0, P(out), oc, M, N, K, 1, 1, 1, 0, M,
                                               0, 0, 0.1, 0, 1.0, 0.0, glu) == 0

            lib.ss_debug_rtlin(0, 0)
            t0 = timed(tiled)
            cols = [f"{t0:7.1f} ({fl / t0 * 1e-6:5.1f})"]
            for per in (1, 2, 3):
                lib.ss_debug_rtlin(per * cus, 1)
                t = timed(rowtile)
                cols.append(f"{t:6.1f} ({fl / t * 1e-6:5.1f})")
            lib.ss_debug_rtlin(0, 1)
            print(f"{M:6d} {name:9s} | " + " | ".join(cols), flush=True)


if __name__ == "__main__":
    main()
