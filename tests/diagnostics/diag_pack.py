"""Diagnosis: for the packs the bench-config test checks, find every utterance whose ASR / ST CTC ids differ between the
ragged pack, the single-utterance HIP entry points and the CPU oracle, and print the top-1 / top-2 margin at the frames
that differ (a near tie of the seeded random weights vs a defect).  Usage: python tests/diagnostics/diag_pack.py [steps] [batch]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import streamspeech_oracle as O  # noqa: E402  (diagnosis tool, not the product path)
from streamspeech_amd import synth, workload  # noqa: E402
from streamspeech_amd.config import ModelConfig, VocoderConfig  # noqa: E402
from streamspeech_amd.engine import HipModel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg, vcfg = ModelConfig(), VocoderConfig()
sd = synth.make_model_state_dict(0, cfg)
osd = O.SD(sd)
m = HipModel(sd, cfg, device="cuda:0")
mine, groups = workload.bench_plan(steps, batch)
torch.set_num_threads(32)
for g in (0, len(groups) // 2, len(groups) - 1):
    utts = [mine[i] for i in groups[g]]
    pcms = [torch.from_numpy(synth.synth_pcm(1234 + u.idx, u.n_samples)).cuda() for u in utts]
    feat, T = m.batch_fbank_cmvn(torch.cat(pcms), [u.n_samples for u in utts])
    enc, Tp = m.batch_encoder_forward(feat, T)
    heads = {}
    for h, name in ((0, "source_unigram"), (1, "ctc_target_unigram")):
        out = m.batch_ctc_greedy(h, enc, Tp, return_raw=True)
        heads[h] = (out, m.last_logits().cpu())
    off_f = off_p = 0
    for b, u in enumerate(utts):
        fb = feat[off_f:off_f + T[b]]
        off_f += T[b]
        enc1 = m.encoder_forward(fb)
        with torch.inference_mode():
            oenc = O.encoder_forward(osd, fb.cpu().numpy(), cfg)
        for h, name in ((0, "source_unigram"), (1, "ctc_target_unigram")):
            out, plog = heads[h]
            single = m.ctc_greedy(h, enc1, want_logits=True)
            with torch.inference_mode():
                ref = O.ctc_head(osd, oenc, name, cfg)
            praw, sraw, oraw = list(out[b][2]), list(single[2]), list(ref[2])
            if praw != oraw or sraw != oraw:
                lp = plog[off_p:off_p + Tp[b]].double()
                ol = ref[3].double().clone()
                ol[:, [cfg.pad, cfg.unk]] = float("-inf")
                for t in range(len(oraw)):
                    if praw[t] != oraw[t] or sraw[t] != oraw[t]:
                        top = torch.topk(ol[t], 2)
                        print(f"pack {g} utt {u.idx} ({u.seconds:.2f}s) head {h} frame {t}: pack {praw[t]} single {sraw[t]} oracle {oraw[t]}; "
                              f"oracle top2 {top.indices.tolist()} margin {float(top.values[0] - top.values[1]):.3e}; "
                              f"pack-oracle logit diff {float((lp[t] - ref[3][t].double()).abs().max()):.3e}; "
                              f"single-oracle {float((single[3][t].cpu().double() - ref[3][t].double()).abs().max()):.3e}")
        off_p += Tp[b]
    print(f"pack {g}: {len(utts)} utterances compared")
