"""Calibration of bench.py's `cpu_baseline.kind = "port"`: the CPU oracle (oracle/streamspeech_oracle.py, what the GPU box
times, because /root/reference does not exist there) against the REFERENCE's own modules (oracle/ref_build.py: the classes of
/root/reference executed in place) on the same inputs, threads and protocol (2 warm-ups, 5 timed passes, median per stage)
-- BASELINE.md §3 / SURVEY.md §8d.  Runs in the build container only.
    python tests/diagnostics/cpu_port_vs_reference.py profiles/r03_cpu_port_vs_reference.json"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kaldi_fbank as K                                  # noqa: E402
from oracle import ref_agent, ref_build                              # noqa: E402
from oracle import streamspeech_oracle as O                          # noqa: E402
from streamspeech_amd import synth, workload                         # noqa: E402
from streamspeech_amd.config import ModelConfig, VocoderConfig       # noqa: E402


def med(fn, warm=2, reps=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    cfg, vcfg = ModelConfig(), VocoderConfig()
    sd, vsd = synth.make_model_state_dict(0, cfg), synth.make_vocoder_state_dict(0, vcfg)
    osd, ovsd = O.SD(sd), O.SD(vsd)
    torch.set_grad_enabled(False)
    nthreads = torch.get_num_threads()
    # reference modules (their own forward code) + the reference's generator classes
    gens = ref_agent.generators()
    dicts = ref_agent.make_dicts(cfg)
    model = ref_agent.build_model(sd, cfg, False, dicts)
    voc = ref_build.build_vocoder(vsd, vcfg)
    asr_gen = gens.CTCDecoder(dicts["source_unigram"], [model])
    st_gen = gens.CTCDecoder(dicts["ctc_target_unigram"], [model])
    unit_gen = gens.CTCSequenceGenerator(dicts["tgt"], [model])

    utts = sorted(workload.make_utterances(64), key=lambda u: u.seconds)
    rows = []
    for u in (utts[8], utts[32], utts[56]):
        pcm = synth.synth_pcm(1234 + u.idx, u.n_samples)
        fb = K.global_cmvn(K.fbank(pcm * np.float32(32768.0)), np.zeros(80, np.float32), np.ones(80, np.float32))
        fbt = torch.from_numpy(fb)
        rec = {"seconds": round(u.seconds, 2), "port_ms": {}, "reference_ms": {}}
        with torch.inference_mode():
            # ---- encoder ----
            t_p, enc_p = med(lambda: O.encoder_forward(osd, fb, cfg))
            t_r, enc_r = med(lambda: model.encoder(fbt.unsqueeze(0), torch.tensor([fb.shape[0]])))
            rec["port_ms"]["a2_a7_encoder"], rec["reference_ms"]["a2_a7_encoder"] = 1e3 * t_p, 1e3 * t_r
            assert float((enc_p - enc_r["encoder_out"][0][:, 0]).abs().max()) < 1e-3
            # ---- CTC heads ----
            t_p, _ = med(lambda: (O.ctc_head(osd, enc_p, "source_unigram", cfg), O.ctc_head(osd, enc_p, "ctc_target_unigram", cfg)))
            t_r, _ = med(lambda: (asr_gen.generate(enc_r, aux_task_name="source_unigram"), st_gen.generate(enc_r, aux_task_name="ctc_target_unigram")))
            rec["port_ms"]["a8_ctc_heads"], rec["reference_ms"]["a8_ctc_heads"] = 1e3 * t_p, 1e3 * t_r
            # ---- MT greedy (no KV cache in either: the agent's generator runs with use_incremental_states=False) + features pass ----
            mt_gen = gens.SequenceGenerator([model], dicts["target_unigram"], beam_size=1, max_len_a=0, max_len_b=100, max_len=0, min_len=1,
                                            search_strategy=gens.BeamSearch(dicts["target_unigram"]), eos=2, use_incremental_states=False)

            def ref_mt():
                fin = mt_gen.generate_decoder([enc_r], fbt.unsqueeze(0), torch.tensor([fb.shape[0]]), {"id": 1}, None, None, None,
                                              aux_task_name="target_unigram", max_new_tokens=u.n_mt)
                toks = fin[0][0]["tokens"]
                toks = toks[:-1] if toks[-1] == 2 else toks
                prev = torch.cat([torch.tensor([2]), toks]).unsqueeze(0)
                return toks, model.target_unigram_decoder(prev, encoder_out=enc_r, features_only=True)[0]

            def port_mt():
                toks = O.mt_greedy(osd, enc_p, cfg, max_new_tokens=u.n_mt)
                toks = toks[:-1] if toks[-1] == cfg.eos else toks
                return toks, O.mt_decoder_features(osd, [cfg.eos] + toks, enc_p, cfg)
            t_p, (tk_p, f_p) = med(port_mt, 1, 3)
            t_r, (tk_r, f_r) = med(ref_mt, 1, 3)
            assert list(tk_p) == tk_r.tolist()
            rec["port_ms"]["a9_a10_mt_greedy_and_features"], rec["reference_ms"]["a9_a10_mt_greedy_and_features"] = 1e3 * t_p, 1e3 * t_r
            # ---- T2U encoder + unit decoder + CTC search ----
            def port_t2u():
                logits = O.unit_decoder_logits(osd, O.t2u_encoder(osd, f_p, cfg), cfg)
                return O.unit_ctc_generate(logits, cfg)

            def ref_t2u():
                t2u = model.synthesizer_encoder(f_r.transpose(0, 1), None)
                return unit_gen.generate(t2u)
            t_p, _ = med(port_t2u)
            t_r, _ = med(ref_t2u)
            rec["port_ms"]["a11_a13_t2u_unit_decoder_ctc"], rec["reference_ms"]["a11_a13_t2u_unit_decoder_ctc"] = 1e3 * t_p, 1e3 * t_r
            # ---- vocoder (same unit count and durations as the workload pins; duration predictor runs in both) ----
            units = [int(x) for x in synth.uniform(5, f"cal_units/{u.idx}", (u.n_units,), 0, 1000)]
            code = torch.tensor(units).view(1, -1)
            t_p, _ = med(lambda: O.vocoder_forward(ovsd, units, vcfg, True), 1, 3)
            t_r, _ = med(lambda: voc(code=code, dur_prediction=True), 1, 3)
            rec["port_ms"]["a14_a15_vocoder_dur_predicted"], rec["reference_ms"]["a14_a15_vocoder_dur_predicted"] = 1e3 * t_p, 1e3 * t_r
        rec["port_ms"] = {k: round(v, 2) for k, v in rec["port_ms"].items()}
        rec["reference_ms"] = {k: round(v, 2) for k, v in rec["reference_ms"].items()}
        sp, sr = sum(rec["port_ms"].values()), sum(rec["reference_ms"].values())
        rec["port_total_ms"], rec["reference_total_ms"], rec["reference_over_port"] = round(sp, 1), round(sr, 1), round(sr / sp, 3)
        print(json.dumps(rec), flush=True)
        rows.append(rec)
    tot_p = sum(r["port_total_ms"] for r in rows)
    tot_r = sum(r["reference_total_ms"] for r in rows)
    out = {"what": "CPU oracle (bench.py cpu_baseline kind 'port') vs the reference's own module classes, same inputs / threads / protocol, "
                   "build container (no GPU)", "threads": nthreads, "host_cpus": os.cpu_count(), "protocol": "2 warm-ups + 5 timed passes (1 + 3 for the MT search and the vocoder), median",
           "utterances": rows, "reference_time_over_port_time": round(tot_r / tot_p, 3),
           "reading": "a ratio > 1 means the reference modules are SLOWER than the port on the same cores, i.e. the port's RTFx overstates the reference's by that factor "
                      "(and understates the GPU / CPU ratio)"}
    if out_path:
        json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "utterances"}))


if __name__ == "__main__":
    main()
