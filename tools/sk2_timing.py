"""Where a conv_sk2 launch spends its cycles, per workgroup (diagnostic build -DK2_TIMING=1: thread 0 of every workgroup
accumulates s_memtime cycles per phase).  python tools/sk2_timing.py   with SS_HIP_LIB=tools/libss_k2timing.so
(SS_EXTRA_FLAGS="-DK2_DIAGNOSTIC_BUILD -DK2_TIMING=1" SS_BUILD_DIR=build/k2timing SS_OUT_LIB=../../tools/libss_k2timing.so bash streamspeech_amd/csrc/build.sh)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import lib as L  # noqa: E402

SHAPES = [("stage0 k11", 38900, 256, 256, 11, 5), ("stage0 k3", 38900, 256, 256, 3, 1), ("stage1 k7", 155600, 128, 128, 7, 1),
          ("stage1 k3", 155600, 128, 128, 3, 3), ("stage2 k11", 622400, 64, 64, 11, 5), ("stage2 k7", 622400, 64, 64, 7, 3),
          ("stage2 k3", 622400, 64, 64, 3, 1), ("unit fc2", 14400, 512, 2048, 1, 1)]


def main():
    lib = L.load()
    raw = C.CDLL(os.environ.get("SS_HIP_LIB", ""))
    raw.ss_debug_sk2_timing.restype = C.c_int
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nseg = int(os.environ.get("SK2_NSEG", "0"))
    print("%-11s %4s %8s %6s | %% of workgroup cycles: %6s %6s %6s %7s %6s %5s | %7s %8s %9s %9s" % (
        "shape", "WGs", "event us", "GHz", "prolog", "setup", "k-loop", "handoff", "epilog", "idle", "steps", "cyc/step", "MFMA/step", "loop eff"))
    for name, M, N, Cin, taps, dil in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        A = torch.randn(M, Cin, device="cuda", generator=g)
        W = torch.randn(N, taps * Cin, device="cuda", generator=g) * (taps * Cin) ** -0.5
        b = torch.randn(N, device="cuda", generator=g)
        R = torch.randn(M, N, device="cuda", generator=g)
        out = torch.empty(M, N, device="cuda")
        pad = dil * (taps - 1) // 2
        args = (s, P(A), Cin, P(W), P(b), P(R), N, None, N, P(out), N, M, N, Cin, taps, dil, 1, pad, M, 0, 0, 0.1, 0, 1.0, 0.0, 0)
        lib.ss_debug_force_tile(4, 0, 0)
        for _ in range(3):
            assert lib.ss_op_conv_gemm(*args) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert lib.ss_op_conv_gemm(*args) == 0
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        lib.ss_debug_force_tile(0, 0, 0)
        buf = (C.c_ulonglong * (512 * 8))()
        G = raw.ss_debug_sk2_timing(buf, 512)
        assert G > 0
        d = np.frombuffer(buf, dtype=np.uint64)[: G * 8].reshape(G, 8).astype(np.float64)
        d = d[d[:, 7] > 0]                                  # workgroups that had work
        tot = d[:, 7] - d[:, 6]                             # cycles each workgroup lived (s_memtime is per XCD: only differences inside one workgroup mean anything)
        span = tot.max()
        ghz = span / (us * 1e3)                             # s_memtime counts shader cycles: the clock this launch actually ran at
        busy = d[:, :5].sum(axis=1)
        ph = 100.0 * d[:, :5].mean(axis=0) / span
        idle = 100.0 * (1.0 - busy.mean() / span)
        cyc = d[:, 2].sum() / d[:, 5].sum()
        BN = 128 if N % 128 == 0 else 64
        mfma = 32.0 * (256 if BN == 128 else 128)           # cycles the step's MFMAs need on one SIMD (32 per v_mfma_f32_16x16x4_f32)
        print("%-11s %4d %8.1f %6.2f |                         %6.1f %6.1f %6.1f %7.1f %6.1f %5.1f | %7.1f %8.0f %9.0f %9.3f" % (
            name, len(d), us, ghz, ph[0], ph[1], ph[2], ph[3], ph[4], idle, d[:, 5].mean(), cyc, mfma, mfma / cyc), flush=True)


if __name__ == "__main__":
    main()
