"""Debug driver: inject a time-out into the persistent MT step and watch the fall-back (stderr visible)."""
import sys
import torch
sys.path.insert(0, ".")
from streamspeech_amd import synth
from streamspeech_amd.config import ModelConfig
from streamspeech_amd.engine import HipModel

cfg = ModelConfig()
m = HipModel(synth.make_model_state_dict(0, cfg), cfg)
enc = m.encoder_forward(torch.from_numpy(synth.synth_fbank(91, 211)).cuda())
ref_t, ref_f = m.mt_greedy(enc, [7, 4242], 16, 1)
print("ref", ref_t, flush=True)
m.set_persistent_mt_step(64)
got_t, got_f = m.mt_greedy(enc, [7, 4242], 16, 1)
print("persistent", got_t == ref_t, flush=True)
m.lib.ss_debug_mt_inject_timeout(m.h)
print("injected", flush=True)
got_t, got_f = m.mt_greedy(enc, [7, 4242], 16, 1)
torch.cuda.synchronize()
print("after", got_t == ref_t, m.lib.ss_mt_get_persistent(m.h), flush=True)
