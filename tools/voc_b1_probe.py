"""Single-utterance vocoder forward under forced tile configurations of the LDS-tiled conv kernel (ss_debug_force_tile:
bm, bn, ks * 10 + pd) -- which tile / k-split / register prefetch depth suits the ~2-GFLOP convs of one utterance."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import synth, workload  # noqa: E402
from streamspeech_amd.config import VocoderConfig  # noqa: E402
from streamspeech_amd.engine import HipVocoder  # noqa: E402

vcfg = VocoderConfig()
voc = HipVocoder(synth.make_vocoder_state_dict(0, vcfg), vcfg)
utts = workload.make_utterances(64)
for want in (2.5, 5.26, 8.0):
    u = min(utts, key=lambda x: abs(x.seconds - want))
    units = [int(c) for c in synth.uniform(7, f"vp/{u.idx}", (u.n_units,), 0, 1000)]
    line = f"{u.seconds:5.2f} s ({u.n_units} units):"
    for name, cfg in (("default", (0, 0, 0)), ("32x32/11", (32, 32, 11)), ("32x32/13", (32, 32, 13)), ("32x32/21", (32, 32, 21)),
                      ("32x32/23", (32, 32, 23)), ("32x64/11", (32, 64, 11)), ("32x64/13", (32, 64, 13)), ("32x64/23", (32, 64, 23)),
                      ("64x64/11", (64, 64, 11)), ("64x64/12", (64, 64, 12)), ("64x64/23", (64, 64, 23))):
        voc.lib.ss_debug_force_tile(*cfg)
        for _ in range(3):
            voc.forward(units, dur_prediction=True, forced_dur=u.durations)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(15):
            voc.forward(units, dur_prediction=True, forced_dur=u.durations)
        torch.cuda.synchronize()
        line += f"  {name} {(time.perf_counter() - t0) / 15 * 1e3:.3f}"
    voc.lib.ss_debug_force_tile(0, 0, 0)
    print(line, flush=True)
