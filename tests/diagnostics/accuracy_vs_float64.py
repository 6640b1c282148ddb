#!/usr/bin/env python
"""How far the HIP logits and the float32 CPU oracle's logits sit from the float64 evaluation of the same arithmetic, per arg-max
stage, on utterances of the bench workload (HIP must be at least as close as the oracle).

  python tests/diagnostics/accuracy_vs_float64.py [n_utterances=6] > profiles/rNN_accuracy_vs_float64.json

Per utterance and stage (asr / st / unit): RMS(HIP - f64), RMS(oracle - f64), their ratio, worst-row max errors.  The HIP side is the
utterance ALONE through the ss_batch_* calls (pack-invariant arithmetic: the bits it has in any pack).  Test infrastructure: imports oracle/."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle import adjudicate as J  # noqa: E402
from oracle import streamspeech_oracle as O  # noqa: E402
from streamspeech_amd import synth, workload  # noqa: E402
from streamspeech_amd.config import ModelConfig  # noqa: E402
from streamspeech_amd.engine import HipModel  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = ModelConfig()
    sd = synth.make_model_state_dict(0, cfg)
    model = HipModel(sd, cfg, device="cuda:0")
    osd = O.SD(sd)
    utts = sorted(workload.make_utterances(64), key=lambda u: u.seconds)
    pick = [utts[int(i * (len(utts) - 1) / max(1, n - 1))] for i in range(n)]
    keep = np.ones(0)
    rows, tot = [], {k: [0.0, 0.0, 0] for k in ("asr", "st", "unit")}
    with torch.inference_mode():
        for u in pick:
            pcm = torch.from_numpy(synth.synth_pcm(1234 + u.idx, u.n_samples)).to("cuda:0")
            h = bench.hip_stage_logits(model, u, pcm)
            enc = O.encoder_forward(osd, h["fbank"], cfg)
            ref = {"asr": O.ctc_head(osd, enc, "source_unigram", cfg)[3], "st": O.ctc_head(osd, enc, "ctc_target_unigram", cfg)[3]}
            toks = O.mt_greedy(osd, enc, cfg, max_new_tokens=u.n_mt)
            body = toks[:-1] if toks and toks[-1] == cfg.eos else toks
            ref["unit"] = O.unit_decoder_logits(osd, O.t2u_encoder(osd, O.mt_decoder_features(osd, [cfg.eos] + body, enc, cfg), cfg), cfg)
            rec = {"utterance": u.idx, "seconds": round(u.seconds, 2), "mt_ids_equal": toks == h["mt"]}
            # intermediate tensors: encoder output rows and MT decoder states (what the T2U encoder is fed)
            sd64 = O.SD(sd, dtype=torch.float64)
            enc64 = O.encoder_forward(sd64, h["fbank"], cfg)
            mt64 = O.mt_decoder_features(sd64, [cfg.eos] + body, enc64, cfg)
            mt32 = O.mt_decoder_features(osd, [cfg.eos] + body, enc, cfg)
            for name, a, b32, b64 in (("enc", h["enc"], enc, enc64), ("mt_states", h["mt_states"], mt32, mt64)):
                n_ = min(a.shape[0], b64.shape[0])
                eh = float(((a[:n_].double() - b64[:n_]) ** 2).mean().sqrt())
                eo = float(((b32[:n_].double() - b64[:n_]) ** 2).mean().sqrt())
                rec[name] = {"rows": int(n_), "rms_hip": eh, "rms_oracle": eo, "ratio": round(eh / eo, 3)}
            for stage in ("asr", "st", "unit"):
                L64 = J.float64_logits(sd, cfg, h["fbank"], stage, toks).double()
                keep = torch.ones(L64.shape[1], dtype=torch.bool)
                keep[[cfg.pad, cfg.unk]] = False
                eh = (h[stage][1].double() - L64)[:, keep]
                eo = (torch.as_tensor(np.asarray(ref[stage])).double() - L64)[:, keep]
                rh, ro = float((eh ** 2).mean().sqrt()), float((eo ** 2).mean().sqrt())
                rec[stage] = {"rows": int(L64.shape[0]), "rms_hip": rh, "rms_oracle": ro, "ratio": round(rh / ro, 3),
                              "worst_row_hip": float(eh.abs().max()), "worst_row_oracle": float(eo.abs().max()),
                              "max_abs_logit": float(L64[:, keep].abs().max())}
                tot[stage][0] += float((eh ** 2).sum())
                tot[stage][1] += float((eo ** 2).sum())
                tot[stage][2] += eh.numel()
            rows.append(rec)
            print(json.dumps(rec), file=sys.stderr, flush=True)
    summary = {s: {"rms_hip": (a / c) ** 0.5, "rms_oracle": (b / c) ** 0.5, "ratio": round((a / b) ** 0.5, 3)} for s, (a, b, c) in tot.items()}
    print(json.dumps({"what": "RMS distance of float32 logits from the float64 evaluation (same weights and fbank), HIP vs the torch CPU oracle",
                      "threads": torch.get_num_threads(), "summary": summary, "utterances": rows}, indent=1))


if __name__ == "__main__":
    main()
