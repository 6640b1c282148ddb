"""A/B of the narrow-stage ResBlock kernels on the batch-32 vocoder workload of bench.py: fused per-ResBlock launches
(resblock.hip, default) vs the pair / two-launch form (ss_debug_force_tile(6)).  Prints whole-vocoder time per batch, the
narrow-stage kernel classes' event time / launches / TFLOP/s, and checks the waveforms are bit-identical.
    python tools/resblock_bench.py [n_utterances]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import lib as L, synth, workload          # noqa: E402
from streamspeech_amd.config import VocoderConfig               # noqa: E402
from streamspeech_amd.engine import HipVocoder                  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    lib = L.load()
    vcfg = VocoderConfig()
    voc = HipVocoder(synth.make_vocoder_state_dict(0, vcfg), vcfg)
    utts = sorted(workload.make_utterances(4 * n), key=lambda u: -u.seconds)[n:2 * n]     # a middle-of-the-distribution batch
    codes = [[int(c) for c in synth.uniform(3, f"rbb/{u.idx}", (u.n_units,), 0, 1000)] for u in utts]
    durs = [u.durations for u in utts]
    frames = sum(sum(d) for d in durs)
    print(f"{n} utterances, {sum(u.seconds for u in utts):.1f} s of audio, {frames} frames")
    outs = {}
    for name, mode in (("fused resblocks", 0), ("pair / two-launch", 6), ("fused resblocks (again)", 0)):
        lib.ss_debug_force_tile(mode, 0, 0)
        for _ in range(2):
            w = voc.batch_forward(codes, True, forced_dur=durs)[0]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            w = voc.batch_forward(codes, True, forced_dur=durs)[0]
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        outs[name] = torch.cat([x.clone() for x in w])
        # per-class event time of the narrow-stage kernels (classes 16 / 17)
        lib.ss_prof_reset()
        lib.ss_prof_enable((1 << 16) | (1 << 17))
        voc.batch_forward(codes, True, forced_dur=durs)
        torch.cuda.synchronize()
        lib.ss_prof_enable(0)
        line = f"{name:26s} vocoder {ms:8.3f} ms/batch"
        for cls in (16, 17):
            t, fl, nl, by = C.c_double(), C.c_double(), C.c_int64(), C.c_double()
            lib.ss_prof_read(cls, C.byref(t), C.byref(fl), C.byref(nl), C.byref(by))
            if nl.value:
                line += f" | {lib.ss_prof_class_name(cls).decode()}: {t.value:7.3f} ms in {nl.value} launches, {fl.value / t.value / 1e9:6.1f} TF"
        lib.ss_prof_reset()
        print(line, flush=True)
    lib.ss_debug_force_tile(0, 0, 0)
    a, b = outs["fused resblocks"], outs["pair / two-launch"]
    print("bit-identical:", bool(torch.equal(a, b)), "max abs diff", float((a - b).abs().max()), "finite", bool(torch.isfinite(a).all()))


if __name__ == "__main__":
    main()
