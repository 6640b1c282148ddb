"""The pack-invariant (whole-tile) form of the fused Conformer FFN (csrc/ffn.hip, canon != 0) at every tile height against the stream-K
form of round 4, over the packed row counts of the bench's length buckets (64 utterances x 25 ... 375 encoder rows): us per FFN.
The tile height does not change a row's bits, so the launcher may pick the fastest one per row count: this table sets its cost model
(SS_FFN_COST).    python tools/ffn_canon_bench.py [rows ...]"""
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
    rows = [int(a) for a in sys.argv[1:]] or [1600, 2400, 3300, 4200, 5000, 6000, 7000, 8000, 9000, 10000, 12000, 12500, 14000, 16000, 20000, 24000]
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

    print("rows | stream-K form (48-row tiles) | whole-tile form at 16 / 32 / 48 / 64 rows per tile | the launcher's own choice      [us]")
    for M in rows:
        x = rn(M, D)

        def fused():
            assert lib.ss_op_ffn_fused(s, P(x), D, P(x), D, P(ln_g), P(ln_b), P(W1), P(b1), P(W2), P(b2), 0.5, None, None, M, D, F) == 0

        lib.ss_debug_canon(0)
        lib.ss_debug_ffn(0, 3, -1)
        out = [f"{M:6d} | {timed(fused):7.1f}"]
        lib.ss_debug_canon(1)
        ts = []
        for wm in (1, 2, 3, 4):
            lib.ss_debug_ffn(0, wm, -1)
            ts.append(timed(fused))
        out.append(" ".join(f"{t:7.1f}" for t in ts))
        lib.ss_debug_ffn(0, 0, -1)
        out.append(f"{timed(fused):7.1f}  (best {min(ts):7.1f} at {16 * (1 + ts.index(min(ts)))})")
        print(" | ".join(out), flush=True)
    lib.ss_debug_canon(0)


if __name__ == "__main__":
    main()
