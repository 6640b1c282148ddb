"""Timing-only ablation of the conv-GEMM main loop (tools/libss_ablate.so, built with -DSS_ABLATE).
bits: 1 no global loads, 2 no MFMA, 4 no LDS store, 8 no barrier."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import lib as L
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libss_ablate.so"))
for name, (res, args) in L.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())

def bench(M, N, Cin, taps, reps=10):
    A = torch.randn(M, Cin, device="cuda"); W = torch.randn(N, taps * Cin, device="cuda") * 0.02
    b = torch.randn(N, device="cuda"); Cc = torch.empty(M, N, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (s, P(A), Cin, P(W), P(b), None, N, None, N, P(Cc), N, M, N, Cin, taps, 1, 1, (taps - 1) // 2, M, 0,
            3 if taps > 1 else 0, 0.1, 0, 1.0, 0.0, 0)
    for _ in range(3): lib.ss_op_conv_gemm(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.ss_op_conv_gemm(*args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

if len(sys.argv) > 1 and sys.argv[1] == "sk":
    # stream-K kernel: 1 no glds in the loop, 2 no ds_read/MFMA, 8 no waitcnt/barrier, 16 no fix-up, 32 no fix-up/epilogue
    shapes = [("stage0 k11", 18000, 256, 256, 11), ("stage0 k3", 18000, 256, 256, 3), ("stage2 k11", 288000, 64, 64, 11),
              ("unit fc2", 6800, 512, 2048, 1)]
    masks = [0, 1, 2, 8, 16, 32, 1 | 8, 1 | 8 | 32, 2 | 32, 1 | 2 | 8 | 32]
    print("%-12s %-6s" % ("shape", "G") + "".join("%9s" % f"m{m}" for m in masks))
    for name, M, N, Cin, taps in shapes:
        for G in (0, 256):
            lib.ss_debug_force_tile(1, 0, G)
            line = "%-12s %-6d" % (name, G)
            for m in masks:
                lib.ss_debug_set_ablate(m)
                line += "%9.1f" % bench(M, N, Cin, taps, reps=5)
            print(line, flush=True)
    sys.exit(0)

shapes = [("stage2 k11", 288000, 64, 64, 11), ("stage0 k11", 18000, 256, 256, 11), ("unit fc2", 6800, 512, 2048, 1)]
tiles = [(64, 64, 11), (32, 64, 11)]
masks = [0, 1, 2, 4, 8, 1 | 4, 1 | 4 | 8, 2 | 4 | 8, 1 | 2, 1 | 2 | 4 | 8]
print("%-12s %-9s" % ("shape", "tile") + "".join("%9s" % f"m{m}" for m in masks))
for name, M, N, Cin, taps in shapes:
    for bm, bn, ks in tiles:
        lib.ss_debug_force_tile(bm, bn, ks)
        line = "%-12s %-9s" % (name, f"{bm}x{bn}/{ks}")
        for m in masks:
            lib.ss_debug_set_ablate(m)
            line += "%9.1f" % bench(M, N, Cin, taps, reps=5)
        print(line, flush=True)
