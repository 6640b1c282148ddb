"""Streaming (BASELINE.json configs[2], 320 ms chunks) cost of the per-chunk encoder call:
full recompute over all audio so far (reference semantics, agent :425-435) vs the incremental entry
point (ss_encoder_stream_forward), and of the per-write vocoder call: all units vs receptive-field tail.
Run on the GPU box: python tools/stream_bench.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import synth  # noqa: E402
from streamspeech_amd.agent import synthesize_tail  # noqa: E402
from streamspeech_amd.config import ModelConfig, VocoderConfig  # noqa: E402
from streamspeech_amd.engine import HipModel, HipVocoder  # noqa: E402
from streamspeech_amd.modules import CodeHiFiGANVocoderWithDur  # noqa: E402


def main():
    cfg, vcfg = ModelConfig(), VocoderConfig()
    m = HipModel(synth.make_model_state_dict(0, cfg), cfg)
    v = HipVocoder(synth.make_vocoder_state_dict(0, vcfg), vcfg)
    out = {}
    for seconds in (5.0, 15.0):
        T_all = int(seconds * 100) - 2
        fb = torch.from_numpy(synth.synth_fbank(7, T_all)).cuda()
        prefixes = [min(T_all, 30 + 32 * i) for i in range(0, (T_all - 30) // 32 + 2)]
        for mode in ("full", "incremental"):
            for rep in range(3):                       # last repetition is the one reported
                m.encoder_stream_reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for T in prefixes:
                    x = fb[:T].contiguous()
                    (m.encoder_forward if mode == "full" else m.encoder_stream_forward)(x, 8, 8)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            out[f"encoder_{mode}_{seconds:g}s_ms_per_chunk"] = round(1e3 * dt / len(prefixes), 3)

    class Surf:
        def __init__(self, hv):
            self.hip = hv
        __call__ = CodeHiFiGANVocoderWithDur.__call__

    voc, rf = Surf(v), vcfg.receptive_field_frames()
    units = [int(u) for u in synth.uniform(3, "sb_units", (37 * 15,), 0, 1000)]
    for ctx, name in ((0, "full"), (rf + 8, "tail")):
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for upto in range(12, len(units) + 1, 12):     # ~12 new units per 320 ms chunk
                synthesize_tail(voc, units[:upto], 12, True, ctx, rf)
                n += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out[f"vocoder_{name}_15s_ms_per_write"] = round(1e3 * dt / n, 3)
    print(json.dumps(out, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/stream_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
