"""Narrow-stage conv kernels (conv_slab.hip) on batch-32-scale shapes: per-shape time, TFLOP/s and operand TB/s
(HIP events around 10 launches), result checksum for A/B runs.  python tools/slab_bench.py [lib.so ...]"""
import ctypes as C
import os
import subprocess
import sys

# name, rows, C, taps, dil, residual operands (0: none, 1: R, 2: R + R2 + /3)
SHAPES = [("s3 k11 d5 R", 1556000, 32, 11, 5, 1), ("s3 k11 d1 RR2", 1556000, 32, 11, 1, 2), ("s3 k7 d3 R", 1556000, 32, 7, 3, 1),
          ("s3 k7 d1 in", 1556000, 32, 7, 1, 0), ("s3 k3 d1 in", 1556000, 32, 3, 1, 0),
          ("s4 k11 d5 R", 3112000, 16, 11, 5, 1), ("s4 k11 d1 RR2", 3112000, 16, 11, 1, 2), ("s4 k7 d3 R", 3112000, 16, 7, 3, 1),
          ("s4 k7 d1 in", 3112000, 16, 7, 1, 0)]


def run_one():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from streamspeech_amd import lib as L
    lib = L.load()
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    print("library:", os.environ.get("SS_HIP_LIB", "default"))
    print("%-14s %9s %8s %8s %14s" % ("shape", "us", "TF", "TB/s", "checksum"))
    tot = 0.0
    for name, M, Cc_, taps, dil, nres in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        A = torch.randn(M, Cc_, device="cuda", generator=g)
        W = torch.randn(Cc_, taps * Cc_, device="cuda", generator=g) * (taps * Cc_) ** -0.5
        b = torch.randn(Cc_, device="cuda", generator=g)
        R = torch.randn(M, Cc_, device="cuda", generator=g) if nres >= 1 else None
        R2 = torch.randn(M, Cc_, device="cuda", generator=g) if nres >= 2 else None
        out = torch.empty(M, Cc_, device="cuda")
        pad = dil * (taps - 1) // 2
        args = (s, P(A), Cc_, P(W), P(b), P(R), Cc_, P(R2), Cc_, P(out), Cc_, M, Cc_, Cc_, taps, dil, 1, pad, M, 0, 3, 0.1, 0, 1.0,
                3.0 if nres >= 2 else 0.0, 0)
        for _ in range(2):
            assert lib.ss_op_conv_gemm(*args) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            lib.ss_op_conv_gemm(*args)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        gf = 2.0 * M * Cc_ * taps * Cc_ / 1e9
        gb = 4.0 * M * Cc_ * (2 + nres) / 1e9
        tot += us
        print("%-14s %9.1f %8.1f %8.2f %14.6e" % (name, us, gf / (us * 1e-6) / 1e3, gb / (us * 1e-6) / 1e3,
                                                  float(out.double().sum())), flush=True)
    print("sum of shapes: %.1f us" % tot, flush=True)


if __name__ == "__main__":
    if os.environ.get("SLAB_CHILD"):
        run_one()
    else:
        for libpath in [None] + sys.argv[1:]:
            env = dict(os.environ, SLAB_CHILD="1")
            if libpath:
                env["SS_HIP_LIB"] = os.path.abspath(libpath)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
