"""rt_linear_kb (csrc/rtlin.hip: K = 512 ... linears of the packed decoders, 256-wide k-slices through LDS, 64-wide summation
blocks) at the bench's shapes: us per launch and algorithmic TFLOP/s for the grid / units-per-wave settings, next to the
blocked-chain LDS-tiled kernel (same bits) and the round-5 stream-K kernel (one chain over K).
    SS_RTLIN_KB_UW=0|1|2|4 python tools/rtlin_kb_bench.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import lib as L          # noqa: E402

# name, rows, N, K, act (2 = ReLU), residual
SHAPES = [("unit fc1 + relu", 45759, 2048, 512, 2, False), ("unit fc2 + residual", 60532, 512, 2048, 0, True), ("qkv", 53197, 1536, 512, 0, False),
          ("attn out + residual", 60532, 512, 512, 0, True), ("cross kv", 4000, 1024, 512, 0, False), ("fc2 short pack", 9000, 512, 2048, 0, True), ("fc2 very short", 2500, 512, 2048, 0, True),
          ("fc1 short pack", 9000, 2048, 512, 2, False)]


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def main():
    lib = L.load()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc     # noqa: E731
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def timed(fn, reps=30):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    print(f"SS_RTLIN_KB_UW={os.environ.get('SS_RTLIN_KB_UW', '0 (default: by unit count)')}")
    print("shape | rt_linear_kb: grid = heuristic / 1 / 2 / 3 per CU  us (TF/s) | LDS tiles, blocked chain | conv_sk2 (one chain, non-canon)")
    for name, M, N, K, act, res in SHAPES:
        x, W, b = rn(M, K), rn(N, K, sc=K ** -0.5), rn(N, sc=0.1)
        R, out = (rn(M, N) if res else None), torch.empty(M, N, device="cuda")
        fl = 2.0 * M * N * K

        def run():
            assert lib.ss_op_conv_gemm(s, P(x), K, P(W), P(b), P(R), N, None, 0, P(out), N, M, N, K, 1, 1, 1, 0, M, 0, 0, 0.1, act, 1.0, 0.0, 0) == 0
        cols = []
        lib.ss_debug_canon(1)
        for grid in (0, cus, 2 * cus, 3 * cus):
            lib.ss_debug_rtlin(grid, 1)
            t = timed(run)
            cols.append(f"{t:7.1f} ({fl / t / 1e6:5.1f})")
        lib.ss_debug_rtlin(0, 0)                 # row-tile kernels off: the blocked-chain LDS tiles
        t = timed(run)
        tiles = f"{t:7.1f} ({fl / t / 1e6:5.1f})"
        lib.ss_debug_rtlin(0, 1)
        lib.ss_debug_canon(0)
        t = timed(run)
        sk2 = f"{t:7.1f} ({fl / t / 1e6:5.1f})"
        print(f"{name:22s} M={M:6d} N={N:4d} K={K:4d} | " + " / ".join(cols) + f" | {tiles} | {sk2}")


if __name__ == "__main__":
    main()
