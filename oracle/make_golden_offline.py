"""Writes tests/golden/offline_generator.json: the reference's OFFLINE generator (oracle/ref_offline.py -- its own
CTCMultiDecoderSequenceGenerator / CTCSequenceGenerator / CTCDecoder / unity SequenceGenerator classes executed from
/root/reference) on seeded synthetic utterances, B = 1 samples: the captured A-/S-/D- stdout lines, the
generate-<subset>.txt lines (T-/H-/D-/P-), the unit sequences.  Run in the build container:
    python -m oracle.make_golden_offline
TEST INFRASTRUCTURE; the committed JSON is what travels to the GPU box."""
import json
import os

import numpy as np
import torch

from streamspeech_amd import synth
from streamspeech_amd.config import ModelConfig

from . import kaldi_fbank as K
from . import ref_offline as RO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "offline_generator.json")

# (sample id, PCM seed, samples at 16 kHz); ids deliberately unordered and sparse like a sharded test set
SAMPLES = [(17, 301, 14400), (3, 302, 27200), (42, 303, 41600), (8, 304, 54400)]
GROUPS = {"short_search": {"max_len_b_mt": 10, "ids": [17, 3, 42, 8]},
          "default_search": {"max_len_b_mt": 200, "ids": [3]}}      # the task default 0 / 200 (speech_to_speech_ctc.py:39-40)


def sample_pcm(seed, n):
    return synth.synth_pcm(seed, n)


def sample_targets(sid):
    """Reference units for the T- line (only for some samples, as in a test set without targets for all)."""
    if sid in (3, 8):
        return [int(u) for u in synth.uniform(sid, "offline_targets", (9 + sid,), 0, 1000)]
    return None


def main():
    cfg = ModelConfig()
    sd = synth.make_model_state_dict(0, cfg)
    g = np.load(os.path.join(ROOT, "tests", "golden", "gcmvn_fr-en.npz"))
    torch.manual_seed(0)
    out = {"note": "outputs of the reference generator classes (oracle/ref_offline.py); regenerate with python -m oracle.make_golden_offline",
           "samples": [{"id": i, "pcm_seed": s, "n_samples": n} for i, s, n in SAMPLES], "groups": {}}
    for name, gcfg in GROUPS.items():
        gen, _, dicts = RO.build_generator(sd, cfg, max_len_b_mt=gcfg["max_len_b_mt"])
        recs = {}
        for sid, seed, n in SAMPLES:
            if sid not in gcfg["ids"]:
                continue
            fb = K.global_cmvn(K.fbank(sample_pcm(seed, n) * np.float32(32768.0)), g["mean"], g["std"])
            r = RO.run_sample(gen, dicts, sid, fb, target_units=sample_targets(sid))
            recs[str(sid)] = {"log": r["log"], "result": r["result"], "units": r["units"], "score": r["score"],
                              "n_positions": len(r["positional_scores"])}
            print(name, sid, r["log"][2][:60], len(r["units"]), r["score"])
        out["groups"][name] = {"max_len_b_mt": gcfg["max_len_b_mt"], "hypotheses": recs}
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
