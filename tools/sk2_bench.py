"""Second-generation stream-K kernel (conv_sk2.hip) against the first (conv_sk.hip) on batch-32-scale shapes:
per-shape time and TFLOP/s (HIP events around 10 launches), max |difference| between the two kernels' outputs and
run-to-run bit equality.  python tools/sk2_bench.py [lib.so ...]  (extra libraries = tuning builds, run in turn)"""
import ctypes as C
import os
import subprocess
import sys

SHAPES = [("stage0 k11", 38900, 256, 256, 11, 5), ("stage0 k7", 38900, 256, 256, 7, 3), ("stage0 k3", 38900, 256, 256, 3, 1),
          ("stage1 k11", 155600, 128, 128, 11, 5), ("stage1 k7", 155600, 128, 128, 7, 1), ("stage1 k3", 155600, 128, 128, 3, 3),
          ("up1", 38900, 512, 256, 3, 1), ("up2", 155600, 256, 128, 3, 1), ("up0", 7780, 1280, 512, 3, 1),
          ("conv_pre", 7780, 512, 128, 7, 1), ("unit fc1", 14400, 2048, 512, 1, 1), ("unit fc2", 14400, 512, 2048, 1, 1),
          ("unit qkv", 14400, 1536, 512, 1, 1), ("enc ffn1", 3900, 2048, 256, 1, 1), ("enc ffn2", 3900, 256, 2048, 1, 1),
          ("stage2 k11", 622400, 64, 64, 11, 5), ("stage2 k7", 622400, 64, 64, 7, 3), ("stage2 k3", 622400, 64, 64, 3, 1),
          ("up3", 622400, 64, 64, 3, 1),
          ("enc qkv", 3900, 768, 256, 1, 1), ("enc out", 3900, 256, 256, 1, 1), ("enc pw1", 3900, 512, 256, 1, 1),
          ("ctc head", 3900, 6016, 256, 1, 1), ("unit out", 14400, 512, 512, 1, 1), ("unit head", 14400, 1024, 512, 1, 1),
          ("sub conv1", 7800, 512, 512, 5, 1), ("short s0 k11", 12000, 256, 256, 11, 5), ("short s1 k3", 48000, 128, 128, 3, 1),
          ("short s0 k3", 12000, 256, 256, 3, 1), ("long s0 k7", 70000, 256, 256, 7, 3)]


def run_one():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from streamspeech_amd import lib as L
    lib = L.load()
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    print("library:", os.environ.get("SS_HIP_LIB", "default"), "| in-loop leaky-ReLU" if os.environ.get("SK2_LRELU") else "")
    print("%-12s %9s %9s %9s | %8s %8s %8s | %9s %5s" % ("shape", "32x64 us", "sk us", "sk2 us", "32x64 TF", "sk TF", "sk2 TF", "max|d|", "det"))
    tot = {"sk": 0.0, "sk2": 0.0}
    only = [x.strip() for x in os.environ.get("SK2_SHAPES", "").split(",") if x.strip()]
    for name, M, N, Cin, taps, dil in SHAPES:
        if only and name not in only:
            continue
        g = torch.Generator(device="cuda").manual_seed(1)
        A = torch.randn(M, Cin, device="cuda", generator=g)
        W = torch.randn(N, taps * Cin, device="cuda", generator=g) * (taps * Cin) ** -0.5
        b = torch.randn(N, device="cuda", generator=g)
        R = torch.randn(M, N, device="cuda", generator=g)
        outs, times = {}, {}
        pad = dil * (taps - 1) // 2
        x3 = bool(os.environ.get("SK2_X3"))          # SK2_X3=1: the first column is the split-bf16 variant of conv_sk2 instead of the 32x64 kernel
        for key, code in (("t", (5, 0, 0) if x3 else (32, 64, 11)), ("sk", (1, 0, 0)), ("sk2", (4, 0, 0))):
            lib.ss_debug_force_tile(*code)
            Cc = torch.empty(M, N, device="cuda")
            in_act = 3 if os.environ.get("SK2_LRELU") else 0      # leaky-ReLU on the conv input inside the k-loop
            args = (s, P(A), Cin, P(W), P(b), P(R), N, None, N, P(Cc), N, M, N, Cin, taps, dil, 1, pad, M, 0, in_act, 0.1, 0, 1.0, 0.0, 0)
            for _ in range(2):
                assert lib.ss_op_conv_gemm(*args) == 0
            torch.cuda.synchronize()
            first = Cc.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                lib.ss_op_conv_gemm(*args)
            e1.record()
            torch.cuda.synchronize()
            times[key] = e0.elapsed_time(e1) * 1e3 / reps
            outs[key] = (Cc, bool(torch.equal(Cc, first)))
        lib.ss_debug_force_tile(0, 0, 0)
        gf = 2.0 * M * N * taps * Cin / 1e9
        d = float((outs["sk2"][0] - outs["t"][0]).abs().max())
        if x3:                                          # relative RMS difference of the split-bf16 result against the f32 one
            d = float(((outs["sk2"][0] - outs["t"][0]).double().pow(2).mean() / outs["sk2"][0].double().pow(2).mean()).sqrt())
        tf = {k: gf / (times[k] * 1e-6) / 1e3 for k in times}
        tot["sk"] += times["sk"]; tot["sk2"] += times["sk2"]
        print("%-12s %9.1f %9.1f %9.1f | %8.1f %8.1f %8.1f | %9.2e %5s" % (name, times["t"], times["sk"], times["sk2"], tf["t"], tf["sk"],
                                                                          tf["sk2"], d, outs["sk2"][1]), flush=True)
    print("sum of shapes: sk %.1f us, sk2 %.1f us; bounded-wait time-outs: %d" % (tot["sk"], tot["sk2"], lib.ss_debug_sk_errors()), flush=True)


if __name__ == "__main__":
    if os.environ.get("SK2_CHILD"):
        run_one()
    else:
        for libpath in [None] + sys.argv[1:]:
            env = dict(os.environ, SK2_CHILD="1")
            if libpath:
                env["SS_HIP_LIB"] = os.path.abspath(libpath)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
