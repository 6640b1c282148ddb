"""One utterance at a time (configs[1] read literally): run the median utterance N times; meant to be run under
`rocprofv3 --kernel-trace --stats` so that the per-kernel averages of the launch-bound chain can be read off
(tools/jobs/r03_b1_prof.sh)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from streamspeech_amd import synth, workload  # noqa: E402
from streamspeech_amd.config import ModelConfig, VocoderConfig  # noqa: E402
from streamspeech_amd.engine import HipModel, HipVocoder  # noqa: E402

cfg, vcfg = ModelConfig(), VocoderConfig()
model = HipModel(synth.make_model_state_dict(0, cfg), cfg)
voc = HipVocoder(synth.make_vocoder_state_dict(0, vcfg), vcfg)
model.set_persistent_mt_step(int(os.environ.get("SS_B1_MT_WGS", "64")))
utts = workload.make_utterances(64)
want = float(os.environ.get("SS_B1_SECONDS", "0"))
u = min(utts, key=lambda x: abs(x.seconds - want)) if want > 0 else sorted(utts, key=lambda x: x.seconds)[len(utts) // 2]
pcm = torch.from_numpy(synth.synth_pcm(1234 + u.idx, u.n_samples)).cuda()
for _ in range(3):
    bench.run_utterance(model, voc, pcm, u)
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t0 = time.perf_counter()
for _ in range(N):
    bench.run_utterance(model, voc, pcm, u)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"utterance {u.seconds:.2f} s: {dt * 1e3:.3f} ms per run_utterance ({u.seconds / dt:.0f}x real time), N = {N}")
