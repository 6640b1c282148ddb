"""Where one utterance's latency goes (BASELINE.json configs[1] read literally: batch 1, one stream): per stage wall time with
a device synchronisation after each stage, launches per stage from the library's census, for a short / median / long
utterance of the bench workload.   python tools/latency_breakdown.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import lib as L, synth, workload                   # noqa: E402
from streamspeech_amd.config import ModelConfig, VocoderConfig           # noqa: E402
from streamspeech_amd.engine import HipModel, HipVocoder                 # noqa: E402
from streamspeech_amd.pipeline import mt_greedy, units_from_tokens       # noqa: E402


def launches(lib):
    tot = 0
    for c in range(lib.ss_prof_num_classes()):
        n = C.c_int64()
        lib.ss_prof_totals(c, None, None, C.byref(n))
        tot += n.value
    return tot


def main():
    lib = L.load()
    cfg, vcfg = ModelConfig(), VocoderConfig()
    model = HipModel(synth.make_model_state_dict(0, cfg), cfg)
    voc = HipVocoder(synth.make_vocoder_state_dict(0, vcfg), vcfg)
    utts = sorted(workload.make_utterances(64), key=lambda u: u.seconds)
    for u in (utts[6], utts[32], utts[58]):
        pcm = torch.from_numpy(synth.synth_pcm(1234 + u.idx, u.n_samples)).cuda()
        rows = {}

        def run(record):
            def stage(name, fn):
                torch.cuda.synchronize()
                n0, t0 = launches(lib), time.perf_counter()
                out = fn()
                torch.cuda.synchronize()
                if record:
                    rows.setdefault(name, []).append((1e3 * (time.perf_counter() - t0), launches(lib) - n0))
                return out
            feat = stage("a1 fbank+cmvn", lambda: model.fbank_cmvn(pcm))
            enc = stage("a2-a7 encoder", lambda: model.encoder_forward(feat))
            stage("a8 ctc heads x2", lambda: (model.ctc_greedy(0, enc), model.ctc_greedy(1, enc)))
            toks, feats = stage("a9-a10 mt greedy", lambda: mt_greedy(model, enc, max_new_tokens=u.n_mt))
            n_in = len(toks) if toks[-1] != cfg.eos else len(toks) - 1
            unit_toks = stage("a11-a13 t2u+unit+ctc", lambda: model.t2u_units(feats[: n_in + 1]))[0]
            units = workload.resize_units(units_from_tokens(unit_toks, cfg), u.n_units, u.idx)
            stage("a14-a15 vocoder", lambda: voc.forward(units, dur_prediction=True, forced_dur=u.durations))

        for i in range(7):
            run(i >= 2)
        # and without the per-stage synchronisations
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run(False)
        torch.cuda.synchronize()
        e2e = 1e3 * (time.perf_counter() - t0) / 5
        print(f"utterance {u.seconds:.2f} s (T = {u.n_samples // 160}, N_mt = {u.n_mt}, K = {u.n_units}):")
        tot = 0.0
        for k, v in rows.items():
            ms = sorted(x[0] for x in v)[len(v) // 2]
            tot += ms
            print(f"  {k:24s} {ms:7.3f} ms  {v[0][1]:5d} GEMM-class launches")
        print(f"  {'sum of stages':24s} {tot:7.3f} ms;  end to end with per-stage syncs inside run(): {e2e:7.3f} ms -> {u.seconds / e2e * 1e3:.0f}x real time")


if __name__ == "__main__":
    main()
