"""Per-shape timing of the conv-GEMM kernel on the vocoder / encoder / decoder layer shapes
(HIP events around 20 launches each).  Run on the GPU box: python tools/conv_bench.py"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import lib as L  # noqa: E402

lib = L.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())


def bench(name, M, N, Cin, taps=1, dil=1, reps=20):
    A = torch.randn(M, Cin, device="cuda")
    W = torch.randn(N, taps * Cin, device="cuda") * (taps * Cin) ** -0.5
    b = torch.randn(N, device="cuda")
    Cc = torch.empty(M, N, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pad = dil * (taps - 1) // 2
    args = (s, P(A), Cin, P(W), P(b), None, N, None, N, P(Cc), N, M, N, Cin, taps, dil, 1, pad, M, 0, 3 if taps > 1 else 0,
            0.1, 0, 1.0, 0.0, 0)
    for _ in range(3):
        lib.ss_op_conv_gemm(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.ss_op_conv_gemm(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gf = 2.0 * M * N * taps * Cin / 1e9
    return {"name": name, "M": M, "N": N, "K": taps * Cin, "us": round(us, 2), "gflop": round(gf, 3),
            "tflops": round(gf / (us * 1e-6) / 1e3, 2)}


def sweep():
    """tile sweep on representative shapes (N > 32, Cin % 32 == 0)."""
    shapes = [("stage0 k11", 1125, 256, 256, 11), ("stage0 k3", 1125, 256, 256, 3), ("stage1 k11", 4500, 128, 128, 11),
              ("stage1 k3", 4500, 128, 128, 3), ("stage2 k11", 18000, 64, 64, 11), ("stage2 k3", 18000, 64, 64, 3),
              ("up0", 225, 1280, 512, 3), ("unit fc2", 425, 512, 2048, 1), ("unit fc1", 425, 2048, 512, 1),
              ("unit qkv", 425, 1536, 512, 1), ("sub1", 112, 512, 512, 5)]
    tiles = [(64, 64, 11), (64, 64, 12), (64, 64, 13), (64, 64, 23), (32, 64, 11), (32, 64, 12), (32, 64, 13), (32, 64, 14),
             (32, 64, 23), (32, 64, 43), (32, 32, 11), (32, 32, 12), (32, 32, 13), (32, 32, 14), (32, 32, 23), (32, 32, 43)]
    print("%-12s" % "shape" + "".join("%9s" % f"{a}x{b}/{k}" for a, b, k in tiles))
    for name, M, N, Cin, taps in shapes:
        line = "%-12s" % name
        for bm, bn, ks in tiles:
            lib.ss_debug_force_tile(bm, bn, ks)
            r = bench(name, M, N, Cin, taps, 1, reps=10)
            line += "%9.1f" % r["us"]
        print(line, flush=True)
    lib.ss_debug_force_tile(0, 0, 0)


def sweep_big():
    """batch-scale shapes (16 utterances packed)."""
    shapes = [("enc ffn1", 1800, 2048, 256, 1), ("enc ffn2", 1800, 256, 2048, 1), ("enc qkv", 1800, 768, 256, 1),
              ("ctc head", 1800, 6000, 256, 1), ("unit fc1", 6800, 2048, 512, 1), ("unit fc2", 6800, 512, 2048, 1),
              ("stage0 k11", 18000, 256, 256, 11), ("stage0 k3", 18000, 256, 256, 3), ("stage1 k11", 72000, 128, 128, 11),
              ("stage1 k3", 72000, 128, 128, 3), ("stage2 k11", 288000, 64, 64, 11), ("stage2 k3", 288000, 64, 64, 3),
              ("up0", 3600, 1280, 512, 3), ("up1", 18000, 512, 256, 3)]
    tiles = [(128, 128, 11), (128, 64, 11), (64, 64, 11), (64, 64, 12), (32, 64, 11), (32, 32, 11)]
    print("%-12s" % "shape" + "".join("%11s" % f"{a}x{b}/{k}" for a, b, k in tiles) + "   GFLOP")
    for name, M, N, Cin, taps in shapes:
        line = "%-12s" % name
        for bm, bn, ks in tiles:
            lib.ss_debug_force_tile(bm, bn, ks)
            r = bench(name, M, N, Cin, taps, 1, reps=5)
            line += "%11.1f" % r["us"]
        print(line + "   %.2f" % r["gflop"], flush=True)
    lib.ss_debug_force_tile(0, 0, 0)


def sweep_sk():
    """persistent stream-K kernel (force code bm=1, ks = grid) against the 32x64 kernel, batch and
    single-utterance scale."""
    shapes = [("stage0 k11 b32", 36000, 256, 256, 11), ("stage0 k11 b16", 18000, 256, 256, 11), ("stage0 k3 b16", 18000, 256, 256, 3),
              ("stage1 k11 b16", 72000, 128, 128, 11), ("stage1 k3 b16", 72000, 128, 128, 3), ("stage2 k11 b16", 288000, 64, 64, 11),
              ("stage2 k3 b16", 288000, 64, 64, 3), ("up0 b16", 3600, 1280, 512, 3), ("up1 b16", 18000, 512, 256, 3),
              ("up2 b16", 72000, 256, 128, 3), ("stage0 k11 b4", 4500, 256, 256, 11), ("stage0 k11 b1", 1125, 256, 256, 11),
              ("stage1 k7 b1", 4500, 128, 128, 7), ("stage2 k3 b1", 18000, 64, 64, 3), ("up0 b1", 225, 1280, 512, 3),
              ("unit fc1 b16", 6800, 2048, 512, 1), ("unit fc2 b16", 6800, 512, 2048, 1)]
    cfgs = [("heur", 0, 0, 0), ("32x64", 32, 64, 11), ("sk auto", 1, 0, 0), ("sk XCDgrp", 1, 8, 0), ("sk 256", 1, 0, 256)]
    print("%-16s" % "shape" + "".join("%10s" % c[0] for c in cfgs) + "   GFLOP   best TF")
    for name, M, N, Cin, taps in shapes:
        line, best = "%-16s" % name, 1e9
        for _, bm, bn, ks in cfgs:
            lib.ss_debug_force_tile(bm, bn, ks)
            r = bench(name, M, N, Cin, taps, 1, reps=5)
            line += "%10.1f" % r["us"]
            best = min(best, r["us"])
        print(line + "   %6.2f  %6.1f" % (r["gflop"], r["gflop"] / best * 1e3), flush=True)
    lib.ss_debug_force_tile(0, 0, 0)
    print("sk errors:", lib.ss_debug_sk_errors())


def sweep_enc():
    """encoder / decoder linear shapes at batch-32 scale (M = 3900 encoder rows, 14400 unit-decoder rows)."""
    shapes = [("enc ffn1", 3900, 2048, 256, 1), ("enc ffn2", 3900, 256, 2048, 1), ("enc qkv", 3900, 768, 256, 1),
              ("enc out", 3900, 256, 256, 1), ("ctc head", 3900, 6000, 256, 1), ("unit qkv", 14400, 1536, 512, 1),
              ("unit out", 14400, 512, 512, 1), ("unit fc1", 14400, 2048, 512, 1), ("unit fc2", 14400, 512, 2048, 1),
              ("unit head", 14400, 1005, 512, 1)]
    cfgs = [("heur", 0, 0, 0), ("32x64", 32, 64, 11), ("64x64", 64, 64, 11), ("128x64", 128, 64, 11), ("sk auto", 1, 0, 0)]
    print("%-12s" % "shape" + "".join("%10s" % c[0] for c in cfgs) + "   GFLOP   best TF")
    for name, M, N, Cin, taps in shapes:
        line, best = "%-12s" % name, 1e9
        for _, bm, bn, ks in cfgs:
            lib.ss_debug_force_tile(bm, bn, ks)
            r = bench(name, M, N, Cin, taps, 1, reps=5)
            line += "%10.1f" % r["us"]
            best = min(best, r["us"])
        print(line + "   %6.2f  %6.1f" % (r["gflop"], r["gflop"] / best * 1e3), flush=True)
    lib.ss_debug_force_tile(0, 0, 0)


def sweep_slab():
    """narrow vocoder stages: slab kernel (default dispatch) vs the LDS-tiled kernels (force code 2)."""
    shapes = [("stage3 k11 b16", 576000, 32, 32, 11), ("stage3 k7 b16", 576000, 32, 32, 7), ("stage3 k3 b16", 576000, 32, 32, 3),
              ("stage4 k11 b16", 1152000, 16, 16, 11), ("stage4 k7 b16", 1152000, 16, 16, 7), ("stage4 k3 b16", 1152000, 16, 16, 3),
              ("up4 b16", 576000, 32, 32, 3), ("stage3 k11 b1", 36000, 32, 32, 11), ("stage4 k3 b1", 72000, 16, 16, 3)]
    print("%-16s%10s%10s   GFLOP  slab TF  slab GB/s (in+out)" % ("shape", "slab", "tiled"))
    for name, M, N, Cin, taps in shapes:
        lib.ss_debug_force_tile(0, 0, 0)
        a = bench(name, M, N, Cin, taps, 1, reps=5)
        lib.ss_debug_force_tile(2, 0, 0)
        b = bench(name, M, N, Cin, taps, 1, reps=5)
        print("%-16s%10.1f%10.1f   %5.2f   %6.1f   %7.0f" % (name, a["us"], b["us"], a["gflop"], a["tflops"],
                                                            4.0 * M * (N + Cin) / (a["us"] * 1e-6) / 1e9), flush=True)
    lib.ss_debug_force_tile(0, 0, 0)


def sweep_b1():
    """single-utterance encoder / decoder linears (M = 131 and 201 rows, just past the small-M kernel's row limit): the
    32x32 tile kernel with k-split KS and register prefetch depth PD; run once more with SS_SMALLM_MAX_ROWS=256 in the
    environment to see the small-M kernel ("heur" column) on the same shapes."""
    shapes = []
    for M in (131, 201):
        shapes += [(f"ffn1 M={M}", M, 2048, 256, 1), (f"ffn2 M={M}", M, 256, 2048, 1), (f"qkv M={M}", M, 768, 256, 1),
                   (f"out M={M}", M, 256, 256, 1)]
    shapes += [("unit fc1 M=500", 500, 2048, 512, 1), ("unit fc2 M=500", 500, 512, 2048, 1)]
    cfgs = [("heur", 0, 0, 0), ("32x32/11", 32, 32, 11), ("32x32/14", 32, 32, 14), ("32x32/23", 32, 32, 23), ("32x32/43", 32, 32, 43),
            ("32x64/11", 32, 64, 11), ("32x64/14", 32, 64, 14), ("32x64/43", 32, 64, 43)]
    print("%-16s" % "shape" + "".join("%10s" % c[0] for c in cfgs) + "   GFLOP")
    for name, M, N, Cin, taps in shapes:
        line = "%-16s" % name
        for _, bm, bn, ks in cfgs:
            lib.ss_debug_force_tile(bm, bn, ks)
            r = bench(name, M, N, Cin, taps, 1, reps=50)
            line += "%10.1f" % r["us"]
        print(line + "   %6.3f" % r["gflop"], flush=True)
    lib.ss_debug_force_tile(0, 0, 0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "b1":
        return sweep_b1()
    if len(sys.argv) > 1 and sys.argv[1] == "enc":
        return sweep_enc()
    if len(sys.argv) > 1 and sys.argv[1] == "slab":
        return sweep_slab()
    if len(sys.argv) > 1 and sys.argv[1] == "sk":
        return sweep_sk()
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        return sweep()
    if len(sys.argv) > 1 and sys.argv[1] == "sweep_big":
        return sweep_big()
    F = 225  # frames of a 4.5 s utterance
    rows = []
    T, Cc = F, 512
    for i, (u, ku) in enumerate(zip((5, 4, 4, 2, 2), (11, 8, 8, 4, 4))):
        rows.append(bench(f"up{i} (3-tap polyphase)", T, u * Cc // 2, Cc, 3))
        T, Cc = T * u, Cc // 2
        for k in (3, 7, 11):
            rows.append(bench(f"stage{i} resconv k{k} d1", T, Cc, Cc, k, 1))
            rows.append(bench(f"stage{i} resconv k{k} d5", T, Cc, Cc, k, 5))
    for name, M, N, K in [("enc ffn1", 113, 2048, 256), ("enc ffn2", 113, 256, 2048), ("enc qkv", 113, 768, 256),
                          ("ctc head", 113, 6000, 256), ("unit fc1", 425, 2048, 512), ("unit fc2", 425, 512, 2048),
                          ("unit qkv", 425, 1536, 512), ("unit head", 425, 1005, 512), ("mt fc1 M=1", 1, 2048, 512),
                          ("mt fc2 M=1", 1, 512, 2048), ("mt out M=1", 1, 512, 512), ("mt logits M=1", 1, 6000, 512)]:
        rows.append(bench(name, M, N, K))
    for r in rows:
        print(f"{r['name']:28s} M={r['M']:6d} N={r['N']:5d} K={r['K']:5d}  {r['us']:8.2f} us  {r['tflops']:7.2f} TFLOP/s")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/conv_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
