"""The 64-channel vocoder-stage convs of a packed batch on the slab kernel with streamed weights (csrc/conv_c64.hip, input
leaky-ReLU applied while staging) against the stream-K kernel they ran on through round 3 (conv_sk2<64>: pre-activated input, and
for the second conv of a pair a pre-activated twin output): us per launch and algorithmic TFLOP/s.
    python tools/c64_bench.py [rows]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import lib as L          # noqa: E402

CH = int(os.environ.get("C64_BENCH_CHANNELS", "64"))      # 64 (conv_c64.hip) or 32 (conv_c32.hip)


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else {256: 144000, 128: 288000, 64: 576000, 32: 1152000, 16: 2304000}[CH]
    lib = L.load()
    dbg = {256: lambda on: lib.ss_debug_conv_c64(9 if on else 8), 128: lambda on: lib.ss_debug_conv_c64(7 if on else 6), 64: lib.ss_debug_conv_c64, 32: lib.ss_debug_conv_c32, 16: lib.ss_debug_conv_c16}[CH]
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc     # noqa: E731
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x, R, out = rn(M, CH), rn(M, CH), torch.empty(M, CH, device="cuda")

    def timed(fn, reps=20):
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

    print(f"{M} rows x {CH} channels; conv1 = dilated conv of lrelu(x) + bias; conv2 = conv of lrelu(h) + bias + residual")
    print("taps dil | conv1: stream-K (pre-activated in, LRELU epilogue) | slab | conv2: stream-K (+ twin out) | slab   [us (TF/s)]")
    for taps in (3, 7, 11):
        for dil in (1, 3, 5):
            W, b = rn(CH, taps * CH, sc=(taps * CH) ** -0.5), rn(CH, sc=0.1)
            fl = 2.0 * M * CH * CH * taps
            pad = dil * (taps - 1) // 2
            cols = []
            for conv2 in (False, True):
                for slab in (False, True):
                    dbg(1 if slab else 0)
                    # stream-K form: input already activated by the producer (in_act 0), conv1 applies the LRELU epilogue; the slab
                    # form reads raw rows (in_act 3) and writes raw rows
                    in_act = 3 if slab else 0
                    act = 0 if (slab or conv2) else 3
                    d = dil if not conv2 else 1
                    pd = pad if not conv2 else (taps - 1) // 2

                    def run():
                        assert lib.ss_op_conv_gemm(s, P(x), CH, P(W), P(b), P(R) if conv2 else None, CH, None, 0, P(out), CH, M, CH, CH, taps, d, 1, pd,
                                                   M, 0, in_act, 0.1, act, 1.0, 0.0, 0) == 0

                    t = timed(run)
                    cols.append(f"{t:7.1f} ({fl / t * 1e-6:5.1f})")
            print(f"{taps:4d} {dil:3d} | " + " | ".join(cols), flush=True)
    dbg(1)
    print("(the stream-K conv2 of the pipeline additionally writes the pre-activated twin of its output: +1 tensor pass not timed here)")


if __name__ == "__main__":
    main()
