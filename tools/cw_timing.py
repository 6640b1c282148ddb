"""Where a Winograd slab-kernel launch (csrc/conv_c64w.hip) spends its cycles, per workgroup: diagnostic build -DCW_TIMING=1 (thread 0
of every workgroup accumulates s_memtime cycles for slab staging / contraction / epilogue / wait at block start).
  SS_EXTRA_FLAGS="-DCW_TIMING=1" SS_BUILD_DIR=build/cwt SS_OUT_LIB=../../tools/bin/libss_cwt.so bash streamspeech_amd/csrc/build.sh
  SS_HIP_LIB=tools/bin/libss_cwt.so python tools/cw_timing.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import lib as L  # noqa: E402

SHAPES = [(256, 144000), (128, 288000), (64, 576000)]


def main():
    lib = L.load()
    raw = C.CDLL(os.environ["SS_HIP_LIB"])
    raw.ss_debug_cw_timing.restype = C.c_int
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())   # noqa: E731
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    print("ch taps dil conv | event us  GHz | % of the workgroup's cycles: staging contraction epilogue wait | blocks/wg | contraction cycles per block / MFMA cycles per block")
    for CH, M in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        x, R, out = torch.randn(M, CH, device="cuda", generator=g), torch.randn(M, CH, device="cuda", generator=g), torch.empty(M, CH, device="cuda")
        for taps in (3, 7, 11):
            for dil, conv2 in ((1, False), (5, False), (1, True)):
                if CH == 64 and taps >= 7 and dil == 5:
                    continue
                W = torch.randn(CH, taps * CH, device="cuda", generator=g) * (taps * CH) ** -0.5
                b = torch.randn(CH, device="cuda", generator=g) * 0.1
                pad = dil * (taps - 1) // 2
                args = (s, P(x), CH, P(W), P(b), P(R) if conv2 else None, CH, None, 0, P(out), CH, M, CH, CH, taps, dil, 1, pad, M, 0, 3, 0.1, 0, 1.0, 0.0, 0)
                for _ in range(3):
                    assert lib.ss_op_conv_gemm(*args) == 0
                torch.cuda.synchronize()
                buf = (C.c_ulonglong * (1024 * 8))()
                raw.ss_debug_cw_timing(buf, 1024)          # clear
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert lib.ss_op_conv_gemm(*args) == 0
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3
                raw.ss_debug_cw_timing(buf, 1024)
                d = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
                d = d[d[:, 7] > 0]
                life = d[:, 6] - d[:, 5]
                span = life.max()
                ph = 100.0 * d[:, :4].mean(axis=0) / span
                groups = (taps + 2) // 3
                cs = 128 if CH == 256 else CH
                prods = 4 * (taps // 3) + (taps % 3 + 1 if taps % 3 else 0)                     # products per channel block: 4 per full tap group, 2 / 3 for a one- / two-tap tail
                mfma = prods * (cs // 16) * (CH // cs) * 32 * (32 if CH >= 64 else 16)           # sub-steps x MFMAs x 32 cycles, per wave and block
                per_blk = d[:, 1].sum() / d[:, 4].sum()
                print(f"{CH:3d} {taps:4d} {dil:3d} {'c2+R' if conv2 else 'c1  '} | {us:8.1f} {span / (us * 1e3):5.2f} | {ph[0]:6.1f} {ph[1]:6.1f} {ph[2]:6.1f} {ph[3]:6.1f} | "
                      f"{d[:, 4].mean():5.1f} | {per_blk:9.0f} / {mfma:8d} = {per_blk / mfma:5.2f} (x2 waves per SIMD: {per_blk / (2 * mfma):4.2f})", flush=True)


if __name__ == "__main__":
    main()
