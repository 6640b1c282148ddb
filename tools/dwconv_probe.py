"""Kernel duration of the depthwise-conv + BN + SiLU launch at streaming / single-utterance row counts (run under rocprofv3
--kernel-trace --stats, or read the HIP-event averages it prints)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import lib as L  # noqa: E402

lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
Cc, K = 256, 31
for T in (8, 24, 131, 375, 4200):
    x = torch.randn(T, Cc, device="cuda")
    w = torch.randn(K, Cc, device="cuda")
    m, v, g, b = (torch.randn(Cc, device="cuda") for _ in range(4))
    v = v.abs() + 0.5
    y = torch.empty_like(x)
    args = (S(), P(x), Cc, P(y), Cc, P(w), K, P(m), P(v), P(g), P(b), 1e-5, T, Cc, 8)
    for _ in range(5):
        lib.ss_op_dwconv_bn_silu(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        lib.ss_op_dwconv_bn_silu(*args)
    e1.record()
    torch.cuda.synchronize()
    print(f"T = {T:5d}: {e0.elapsed_time(e1) * 5:.2f} us per back-to-back launch")
