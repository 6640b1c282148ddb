#!/usr/bin/env python
"""bench.py -- offline S2ST (BASELINE.json configs[1]) real-time factor on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU, utterance-level data parallel over RCCL.  Either launched by torch.distributed.run
   (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment -- what the driver does), or typed as above with no
   launcher: the command then re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
   --master-addr 127.0.0.1 --master-port <free>` with the same arguments, rank 0 prints the one JSON line, the exit code
   is the launcher's.  `--dry-plan` runs the plan + barrier + reductions of that path on CPU over gloo, no GPU.)

A "step" is one pass of the whole HIP hot path over one ragged batch of --batch (128) synthetic
utterances, each with B = 1 arithmetic (streamspeech_amd/workload.py): PCM already in HBM ->
fbank+CMVN -> chunk-Conformer -> CTC x2 -> AR MT greedy decode -> T2U + NAR unit decoder -> CTC
collapse -> unit HiFi-GAN -> waveform in HBM.  The default K = 16 steps is twice BASELINE.json's
1024-utterance set, four batches on each of the 4 default streams (a scratch set each; 8 streams measure +0.9 % at twice the HBM,
profiles/r06_stream_sweep.txt).  Packs: 32 in rounds 1-3, 64 in round 4, 128 since round 5 (ids do not depend on the pack --
tests/test_pack_invariance_gpu.py); with --batch 1 a step is one utterance through the single-utterance entry points.
stdout carries ONE line under 6 KB (compact_line); the full result goes to the sidecar bench_detail.json (SS_BENCH_DETAIL) and to stderr.
value = total audio seconds / wall seconds over all ranks (RTFx; higher is better); the line also
carries utterances/sec, the roofline of the dominant kernel (HIP events recorded on the launch
stream inside the timed region) and the CPU oracle timed on this box's host cores (rank 0, N=1).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # kernel arguments in device memory (streamspeech_amd/__init__.py): must precede the first HIP call

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from streamspeech_amd import lib as L                      # noqa: E402
from streamspeech_amd import dp, synth, workload           # noqa: E402
from streamspeech_amd.config import ModelConfig, VocoderConfig  # noqa: E402
from streamspeech_amd.engine import HipModel, HipVocoder   # noqa: E402
from streamspeech_amd.pipeline import mt_greedy, units_from_tokens  # noqa: E402

METRIC = "real-time factor (RTFx = audio seconds / wall seconds) + utterances/sec, offline S2ST fr-en"
WORKLOAD = ("offline S2ST fr-en, B=1 semantics per utterance (ragged no-padding batches), synthetic CVSS-C-shaped utterances "
            "(LogNormal(ln 4.5 s, 0.45) clipped to [1,15] s, seed 1234), full fbank+encoder+CTC+AR-MT+T2U+NAR-unit+vocoder HIP path, "
            "random-init weights of the streamspeech.offline.fr-en architecture")
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0          # same guide, "HBM3E peak BW" (spec; 6.29 TB/s measured copy)


def run_utterance(model, voc, pcm, utt):
    """The timed hot path for one utterance; everything stays in HBM except the id lists the
    agent API itself hands to the host."""
    cfg = model.cfg
    feat = model.fbank_cmvn(pcm)
    enc = model.encoder_forward(feat)
    asr, _, _, _ = model.ctc_greedy(0, enc)
    st, _, _, _ = model.ctc_greedy(1, enc)
    toks, feats = mt_greedy(model, enc, max_new_tokens=utt.n_mt)
    n_in = len(toks) if toks[-1] != cfg.eos else len(toks) - 1
    unit_toks, _, _ = model.t2u_units(feats[: n_in + 1])
    units = workload.resize_units(units_from_tokens(unit_toks, cfg), utt.n_units, utt.idx)
    wav, dur = voc.forward(units, dur_prediction=True, forced_dur=utt.durations)
    return wav, len(asr), len(st), len(toks)


run_batch = workload.run_batch      # the timed step; tests/test_bench_config_gpu.py checks this very function against the oracle


def _pick_threads(O, osd, ovsd, cfg, vcfg, probe_seconds):
    """torch's default (all cores) is far from the best setting for these small ops on a many-core host: probe a few
    thread counts on a short encoder + vocoder pass and keep the fastest."""
    ncores = os.cpu_count() or 1
    best_t, best_dt = torch.get_num_threads(), None
    with torch.inference_mode():
        fb0 = synth.synth_fbank(1, int(probe_seconds * 100) - 2)
        for nt in sorted({min(ncores, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(nt)
            O.encoder_forward(osd, fb0, cfg, n_layers=2)
            t0 = time.perf_counter()
            O.encoder_forward(osd, fb0, cfg, n_layers=4)
            O.vocoder_forward(ovsd, list(range(40)), vcfg, False)
            dt = time.perf_counter() - t0
            if best_dt is None or dt < best_dt:
                best_t, best_dt = nt, dt
    torch.set_num_threads(best_t)
    return best_t


CPU_STAGES = ("a1_fbank_cmvn", "a2_a7_encoder", "a8_ctc_heads", "a9_a10_mt_greedy_and_features", "a11_a13_t2u_unit_decoder_ctc",
              "a14_a15_vocoder")


def cpu_baseline(sd, vsd, cfg, vcfg, utts, reps=5, warm=2, budget_s=30.0, hip_model=None, dev=None):
    """The CPU oracle (torch fp32, the box's host cores) on a bounded sample of the same workload, timed as BASELINE.md
    §3 / SURVEY.md §8d prescribe: per utterance `warm` untimed passes, then `reps` timed passes, MEDIAN per stage
    (a1 ... a15) and end to end.  The sample is three utterances spread over the length distribution (short / median /
    long), cut short if the budget runs out (the entry says what was measured)."""
    from oracle import kaldi_fbank as K
    from oracle import streamspeech_oracle as O
    osd, ovsd = O.SD(sd), O.SD(vsd)
    g_mean, g_std = np.zeros(80, np.float32), np.ones(80, np.float32)
    by_len = sorted(utts[:64], key=lambda u: u.seconds)
    sample = [by_len[len(by_len) // 8], by_len[len(by_len) // 2], by_len[(7 * len(by_len)) // 8]]
    nthreads = _pick_threads(O, osd, ovsd, cfg, vcfg, sample[1].seconds)

    def one_pass(u, pcm):
        st = {}
        t = time.perf_counter()
        fb = K.global_cmvn(K.fbank(pcm * np.float32(32768.0)), g_mean, g_std)
        t, st["a1_fbank_cmvn"] = time.perf_counter(), time.perf_counter() - t
        enc = O.encoder_forward(osd, fb, cfg)
        t, st["a2_a7_encoder"] = time.perf_counter(), time.perf_counter() - t
        O.ctc_head(osd, enc, "source_unigram", cfg)
        O.ctc_head(osd, enc, "ctc_target_unigram", cfg)
        t, st["a8_ctc_heads"] = time.perf_counter(), time.perf_counter() - t
        toks = O.mt_greedy(osd, enc, cfg, max_new_tokens=u.n_mt)
        if toks[-1] == cfg.eos:
            toks = toks[:-1]
        feats = O.mt_decoder_features(osd, [cfg.eos] + toks, enc, cfg)
        t, st["a9_a10_mt_greedy_and_features"] = time.perf_counter(), time.perf_counter() - t
        logits = O.unit_decoder_logits(osd, O.t2u_encoder(osd, feats, cfg), cfg)
        units, _ = O.unit_ctc_generate(logits, cfg)
        units = workload.resize_units(units, u.n_units, u.idx)
        t, st["a11_a13_t2u_unit_decoder_ctc"] = time.perf_counter(), time.perf_counter() - t
        O.vocoder_forward(ovsd, units, vcfg, True, forced_dur=u.durations)
        st["a14_a15_vocoder"] = time.perf_counter() - t
        st["end_to_end"] = sum(st.values())
        return st

    t_begin = time.perf_counter()
    audio, wall, per_utt = 0.0, 0.0, []
    stage_tot = {k: 0.0 for k in CPU_STAGES}
    with torch.inference_mode():
        for u in sample:
            pcm = synth.synth_pcm(1234 + u.idx, u.n_samples)
            n_warm = warm if time.perf_counter() - t_begin < budget_s else 0
            for _ in range(n_warm):
                one_pass(u, pcm)
            runs = []
            for r in range(reps):
                runs.append(one_pass(u, pcm))
                if time.perf_counter() - t_begin > budget_s and len(runs) >= 3:
                    break
            med = {k: float(np.median([x[k] for x in runs])) for k in runs[0]}
            audio += u.seconds
            wall += med["end_to_end"]
            for k in CPU_STAGES:
                stage_tot[k] += med[k]
            per_utt.append({"seconds": round(u.seconds, 2), "warmups": n_warm, "reps": len(runs),
                            "median_ms": round(1e3 * med["end_to_end"], 1), "rtfx": round(u.seconds / med["end_to_end"], 2)})
            if time.perf_counter() - t_begin > budget_s:
                break
    check = oracle_check(sd, cfg, hip_model, sample[:len(per_utt)], dev) if hip_model is not None else None
    return {"value": round(audio / wall, 3), "unit": "x real-time (audio s / wall s)", "utterances_per_sec": round(len(per_utt) / wall, 3),
            "cores": nthreads, "host_cpus": os.cpu_count(), "kind": "port", "oracle_check": check,
            "sample": f"{len(per_utt)} utterances of the same synthetic workload (short / median / long: "
                      + ", ".join(f"{p['seconds']} s" for p in per_utt) + f"; {audio:.1f} s of audio), each {warm} warm-ups then "
                      f"{reps} timed passes, median per stage and end to end",
            "per_utterance": per_utt,
            "stage_ms_per_audio_second": {k: round(1e3 * v / audio, 2) for k, v in stage_tot.items()},
            "port_vs_reference_modules": "profiles/r03_cpu_port_vs_reference.json (this container, same inputs and threads)"}


def hip_stage_logits(model, u, pcm_dev):
    """One utterance ALONE through the ss_batch_* calls (pack-invariant arithmetic: the bits it has in any pack): raw arg-max ids
    and dense logits of every arg-max stage."""
    feat, T = model.batch_fbank_cmvn(pcm_dev, [u.n_samples])
    enc, Tp = model.batch_encoder_forward(feat, T)
    out = {"fbank": feat.cpu().numpy()}
    for h, k in ((0, "asr"), (1, "st")):
        raw = model.batch_ctc_greedy(h, enc, Tp, return_raw=True)[0][2]
        out[k] = (list(raw), model.last_logits().cpu())
    toks, feats, n = model.batch_mt_greedy(enc, Tp, [u.n_mt])
    out["mt"] = list(toks[0])
    raw = model.batch_t2u_units(feats, n, return_raw=True)[1][0]
    out["unit"] = (list(raw), model.last_logits().cpu())
    out["enc"], out["mt_states"] = enc.cpu(), feats[0, :n[0]].cpu()
    return out


def pack_invariance_check(model, work, n_alone=3):
    """Inside the bench process, on the timed packs themselves: utterances of the shortest and of a middle pack, alone vs in
    their pack -- encoder rows, both CTC heads' logits, MT decoder states and unit logits must be BIT-identical (the reference
    decodes one utterance per call: agent/speech_to_speech.streamspeech.agent.py:425-478)."""
    checked, ok = 0, True
    for us, pk in (work[-1], work[len(work) // 2]):
        feat, T = model.batch_fbank_cmvn(pk, [u.n_samples for u in us])
        enc, Tp = model.batch_encoder_forward(feat, T)
        model.batch_ctc_greedy(0, enc, Tp)
        asr = model.last_logits().cpu()
        model.batch_ctc_greedy(1, enc, Tp)
        st = model.last_logits().cpu()
        toks, feats, n = model.batch_mt_greedy(enc, Tp, [u.n_mt for u in us])
        model.batch_t2u_units(feats, n)
        unit = model.last_logits().cpu()
        enc_c, feats_c = enc.cpu(), feats.cpu()
        o1 = np.cumsum([0] + list(Tp))
        o2 = np.cumsum([0] + [x * model.cfg.ctc_upsample for x in n])
        ps = np.cumsum([0] + [u.n_samples for u in us])
        for b in sorted({0, len(us) // 2, len(us) - 1})[:n_alone]:
            a = hip_stage_logits(model, us[b], pk[ps[b]:ps[b + 1]])
            same = (torch.equal(a["enc"], enc_c[o1[b]:o1[b + 1]]) and torch.equal(a["asr"][1], asr[o1[b]:o1[b + 1]]) and
                    torch.equal(a["st"][1], st[o1[b]:o1[b + 1]]) and torch.equal(a["mt_states"], feats_c[b, :n[b]]) and
                    torch.equal(a["unit"][1], unit[o2[b]:o2[b + 1]]) and a["mt"] == list(toks[b]))
            ok, checked = ok and same, checked + 1
    return {"utterances_checked": checked, "packs": 2, "stages": ["encoder rows", "ASR CTC logits", "ST CTC logits", "MT decoder states",
                                                                  "unit logits"],
            "alone_equals_in_pack_bitwise": bool(ok), "pack_invariant_context": bool(model.pack_invariant())}


def oracle_check(sd, cfg, model, sample, dev):
    """cpu_baseline leg only: the ids of the three CPU-baseline utterances, HIP (alone through the ss_batch_* calls = the bits they
    have in any pack) against the float32 oracle fed the HIP fbank; strict, a differing row must pass the float64 adjudication of
    oracle/adjudicate.py (and is then counted as a near-tie row)."""
    from oracle import adjudicate as J
    from oracle import streamspeech_oracle as O
    osd = O.SD(sd)
    rows_total, lines, identical = 0, [], True
    with torch.inference_mode():
        for u in sample:
            pcm = torch.from_numpy(synth.synth_pcm(1234 + u.idx, u.n_samples)).to(dev)
            h = hip_stage_logits(model, u, pcm)
            enc = O.encoder_forward(osd, h["fbank"], cfg)
            ref = {"asr": O.ctc_head(osd, enc, "source_unigram", cfg), "st": O.ctc_head(osd, enc, "ctc_target_unigram", cfg)}
            toks = O.mt_greedy(osd, enc, cfg, max_new_tokens=u.n_mt)
            if toks != h["mt"]:
                raise RuntimeError(f"utterance {u.idx}: MT ids differ from the oracle: {h['mt']} vs {toks}")
            body = toks[:-1] if toks and toks[-1] == cfg.eos else toks
            logits = O.unit_decoder_logits(osd, O.t2u_encoder(osd, O.mt_decoder_features(osd, [cfg.eos] + body, enc, cfg), cfg), cfg)
            ref_unit_raw = O.unit_ctc_generate(logits, cfg)[1]
            for stage, ref_raw, ref_logits in (("asr", ref["asr"][2], ref["asr"][3]), ("st", ref["st"][2], ref["st"][3]),
                                               ("unit", ref_unit_raw, logits)):
                rows_total += len(ref_raw)
                rows = J.differing_rows(h[stage][0], ref_raw)
                if rows:
                    identical = False
                    L64 = J.float64_logits(sd, cfg, h["fbank"], stage, toks)
                    lines += J.adjudicate(f"utt {u.idx} ({u.seconds:.2f} s) {stage}", rows, L64, ref_logits, h[stage][1], [cfg.pad, cfg.unk])
    return {"utterances": len(sample), "argmax_rows_compared": rows_total, "ids_identical_to_float32_oracle": identical,
            "near_tie_rows": len(lines), "adjudicated_in_float64": lines,
            "rule": "strict ids; a differing row must be a float64 top-2 exchange with gap < 2^-20 x max|logit| (oracle/adjudicate.py); "
                    "the measured configuration at full size (320 utterances of 8 packs of 128 in flight, ~250 k rows): tests/test_bench_config_gpu.py"}


def census(lib):
    """Launches and algorithmic work per tile class over the WHOLE process (warm-ups, latency passes, timed region,
    replay): divide a rocprofv3 kernel-stats / PMC profile of this command by these to recompute the roofline."""
    out = {}
    for c in range(lib.ss_prof_num_classes()):
        fl, by, n = C.c_double(), C.c_double(), C.c_int64()
        lib.ss_prof_totals(c, C.byref(fl), C.byref(by), C.byref(n))
        if n.value:
            out[lib.ss_prof_class_name(c).decode()] = {"launches": int(n.value), "algo_tflop": round(fl.value / 1e12, 4),
                                                       "algo_gbytes": round(by.value / 1e9, 3)}
    return out


def _agent_args(segment_ms, **over):
    from streamspeech_amd.agent import StreamSpeechS2STAgent
    p = argparse.ArgumentParser()
    StreamSpeechS2STAgent.add_args(p)
    a = p.parse_args(["--model-path", "synthetic:0", "--data-bin", "/nonexistent", "--vocoder", "synthetic:0", "--dur-prediction",
                      "--sample-rate", "16000"])
    a.source_segment_size, a.device = segment_ms, "gpu"
    for k, v in over.items():
        setattr(a, k, v)
    return a


def streaming_measure(model, voc, lib, cfg, segment_ms=320, n_utts=12, cpu_sd=None, cpu_utts=0, long_seconds=(), configurations=None):
    """BASELINE.json configs[2]: the drop-in SimulEval agent (streamspeech_amd/agent.py) fed `segment_ms` chunks of synthetic
    CVSS-C-shaped utterances, one at a time -- with the incremental encoder + receptive-field vocoder tail (default)
    and with the reference's full recompute at every policy() call (agent :425-435, 686-689, 748-751).  With `cpu_sd`
    = (sd, vsd, vcfg) the SAME agent class runs over the CPU oracle engine (oracle/engine.py: full recompute per call,
    the reference's semantics) on the first `cpu_utts` of the same utterances: the CPU baseline of this config.
    `long_seconds`: the incremental-state claim of SURVEY.md §8f-1 measured where it can matter -- see long_prefix_sweep.
    Random weights: the READ/WRITE pattern and the MT / unit lengths are whatever the random model emits (CTC heads fire on
    most frames, the unit decoder collapses to few units), so per-call costs are indicative, not CVSS-C statistics."""
    from streamspeech_amd import streaming_eval as SE
    from streamspeech_amd.agent import StreamSpeechS2STAgent
    from streamspeech_amd.modules import CodeHiFiGANVocoderWithDur, StreamSpeechModel

    class VocSurface:
        def __init__(self, hv):
            self.hip = hv
        __call__ = CodeHiFiGANVocoderWithDur.__call__

    # utterances up to 9 s: the random model's CTC heads fire more often than a trained model's (up to ~5 subwords/s), and
    # the agent's first-pass search is capped at max_len_b = 100 subwords (agent :162-180)
    cap_s = 9.0 if segment_ms < 640 else 5.0     # whole-word mode (>= 640 ms) commits one more subword per call
    utts = [u for u in workload.make_utterances(8 * n_utts + 8) if u.seconds <= cap_s][: 2 * n_utts + 1]
    pcms = [synth.synth_pcm(1234 + u.idx, u.n_samples) for u in utts]
    out = {"metric": "simultaneous S2ST fr-en, wait-k agent policy() loop, batch 1 (BASELINE.json configs[2])", "mode": "streaming",
           "n_gpus": 1, "dtype": "f32", "data": "synthetic", "segment_ms": segment_ms,
           "config": {"workload": f"{n_utts} synthetic CVSS-C-shaped utterances (LogNormal(ln 4.5 s, 0.45) clipped to [1,15] s, <= {cap_s:.0f} s kept), "
                                  f"{segment_ms}-ms source segments at 16 kHz, StreamSpeechS2STAgent.policy() per segment, "
                                  "random-init weights of the streamspeech.simultaneous.fr-en architecture; the source-finished call's "
                                  "first-pass search pinned to 8 more subwords (random weights never emit </s>: `incremental_search_to_cap` is that case)"},
           "scorer": "streamspeech_amd/streaming_eval.py: restatement of SimulEval's SpeechOutputInstance timing + RTF/StartOffset/EndOffset "
                     "scorers (SimulEval itself cannot import here: yt_dlp / soundfile / textgrid absent; the reference files do not exist on the "
                     "GPU box); pinned against the reference's RTFScorer / StartOffsetScorer / EndOffsetScorer classes loaded from "
                     "/root/reference by tests/test_streaming_eval_cpu.py"}

    def census_launches():
        tot = 0
        for c in range(lib.ss_prof_num_classes()):
            n = C.c_int64()
            lib.ss_prof_totals(c, None, None, C.byref(n))
            tot += n.value
        return tot

    def pin_final_search(agent, remainder=8):
        """The source-finished policy() call lets the first-pass search run to </s> (agent :520-533, max_len_b = 100).  A trained model
        emits </s> a few subwords after the committed prefix; seeded random weights never do, so that call would decode to the 100-token
        cap -- 100 x 145 us that say nothing about the path.  The workload pins the remainder of the final
        search to `remainder` subwords, like the offline workload pins N; the to-the-cap case stays as its own row."""
        gen = agent.generator_mt
        plain = gen.generate_decoder

        def generate_decoder(*a, max_new_tokens=-1, **kw):
            return plain(*a, max_new_tokens=remainder if max_new_tokens == -1 else max_new_tokens, **kw)
        gen.generate_decoder = generate_decoder

    kept_ids = None
    # the agent's default: incremental encoder + tail vocoder + the MT decode step as one persistent launch (mt_step.hip);
    # then the reference's full recompute per call, the default with the launch-per-op decode step (A/B of mt_step.hip), and the
    # default with the final search left to run to the reference's 100-token cap (what random weights do without the pin)
    for name, over in (("incremental", {}), ("full_recompute", {"full_recompute_encoder": True, "vocoder_context_units": 0}),
                       ("incremental_launch_per_op_mt", {"mt_step_workgroups": 0}), ("incremental_search_to_cap", {})):
        if configurations is not None and name not in configurations:
            continue
        agent = StreamSpeechS2STAgent(_agent_args(segment_ms, **over), model=StreamSpeechModel.from_engine(model), vocoder=VocSurface(voc))
        if name != "incremental_search_to_cap":
            pin_final_search(agent)
        SE.run_utterance(agent, pcms[0], segment_ms)                      # warm-up utterance
        n0 = census_launches()
        runs, skipped, ids = [], 0, []
        for j, pcm in enumerate(pcms[1:], 1):
            if len(runs) >= n_utts:
                break
            try:
                runs.append(SE.run_utterance(agent, pcm, segment_ms))
                ids.append(j)
            except IndexError:      # the random model kept committing subwords past the first-pass cap of 100: the reference
                skipped += 1        # agent fails the same way (fairseq sequence_generator: no hypothesis can be finalized)
                agent.reset()
        n1 = census_launches()
        summ = SE.summarize(runs)
        summ["utterances_skipped_prefix_over_cap"] = skipped
        summ["gemm_class_launches_per_policy_call"] = round((n1 - n0) / max(1, summ["policy_calls"]), 1)
        summ["actions_first_utterance"] = runs[0]["actions"]
        out[name] = summ
        kept_ids = kept_ids or ids
    if hasattr(model, "set_persistent_mt_step"):
        model.set_persistent_mt_step(0)             # the agents switched it on for this (shared) context
    out["value"] = out["incremental"]["rtfx_compute"]
    out["unit"] = "x real-time (audio s / policy() compute s, one utterance at a time)"
    out["higher_is_better"] = True
    if "full_recompute" in out:
        out["incremental_speedup_over_full_recompute"] = round(out["full_recompute"]["compute_s"] / out["incremental"]["compute_s"], 3)

    if cpu_sd is not None and cpu_utts > 0:
        # the reference's per-chunk full recompute on the host cores: same agent class, oracle engine (kind "port")
        from oracle.engine import OracleEngine, OracleVocoder
        sd, vsd, vcfg = cpu_sd
        a = _agent_args(segment_ms, full_recompute_encoder=True, vocoder_context_units=0)
        agent = StreamSpeechS2STAgent(a, model=StreamSpeechModel.from_engine(OracleEngine(sd, cfg)), vocoder=OracleVocoder(vsd, vcfg))
        pin_final_search(agent)
        if torch.get_num_threads() > 16:
            torch.set_num_threads(16)         # torch's default (all cores of a 128-core host) is ~20x slower on these small ops (see _pick_threads)
        runs = []
        t_begin = time.perf_counter()
        with torch.inference_mode():
            for j in kept_ids[:cpu_utts]:
                runs.append(SE.run_utterance(agent, pcms[j], segment_ms, sync=False))
                if time.perf_counter() - t_begin > 20.0:
                    break
        cs = SE.summarize(runs)
        out["cpu_baseline"] = {"value": cs["rtfx_compute"], "unit": out["unit"], "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"the first {len(runs)} of the same utterances ({cs['audio_s']} s of audio), the same agent class over the CPU "
                                         "oracle engine with the reference's full recompute at every policy() call, one pass",
                               "ms_per_policy_call_mean": cs["ms_per_policy_call_mean"], "ms_per_policy_call_p95": cs["ms_per_policy_call_p95"],
                               "RTF": cs["RTF"], "RTF_CA": cs["RTF_CA"], "StartOffset_CA_ms": cs["StartOffset_CA_ms"], "EndOffset_CA_ms": cs["EndOffset_CA_ms"]}
    if long_seconds:
        out["long_prefix_sweep"] = long_prefix_sweep(model, voc, cfg, segment_ms, long_seconds)
    return out


def long_prefix_sweep(model, voc, cfg, segment_ms, seconds):
    """SURVEY.md §8f-1 measured where it can matter: for a `d`-second source, the model-side work of every policy() call that
    depends on the audio received so far -- fbank + encoder (+ both CTC heads) over the growing prefix, and the vocoder over the
    growing unit sequence -- with the incremental state (ss_encoder_stream_forward: only non-final rows; vocoder: the last
    new + receptive-field units) and as the reference's full recompute (O(n^2) over the utterance).  The agent loop itself is
    not used here: with random weights its first-pass search overruns max_len_b = 100 on sources this long (the reference's
    generator raises too), so the sweep drives the same entry points directly, one call per `segment_ms` of new audio."""
    from streamspeech_amd.agent import synthesize_tail
    from streamspeech_amd.modules import CodeHiFiGANVocoderWithDur

    class VocSurface:
        def __init__(self, hv):
            self.hip = hv
        __call__ = CodeHiFiGANVocoderWithDur.__call__

    vs = VocSurface(voc)
    chunk = max(1, segment_ms // 40)
    conv_chunk = 16 if chunk >= 16 else 8
    step = 16 * segment_ms
    rf = voc.cfg.receptive_field_frames()
    res = []
    for d in seconds:
        n = int(d * 16000)
        pcm = torch.from_numpy(synth.synth_pcm(4242, n)).to(model.device)
        units_all = [int(x) for x in synth.uniform(7, f"sweep_units/{d}", (int(30 * d),), 0, 1000)]   # ~30 units per source second so far (CVSS-C: ~37)
        row = {"source_s": d, "policy_calls": -(-n // step)}
        for mode in ("incremental", "full_recompute"):
            for rep in range(2):                      # first pass warms shapes
                model.encoder_stream_reset()
                torch.cuda.synchronize()
                t_enc = t_voc = 0.0
                pos, k_prev = 0, 0
                while pos < n:
                    pos = min(n, pos + step)
                    t0 = time.perf_counter()
                    fb = model.fbank_cmvn(pcm[:pos])
                    if fb.shape[0] > 0:
                        enc = (model.encoder_stream_forward(fb, chunk, conv_chunk) if mode == "incremental"
                               else model.encoder_forward(fb, chunk, conv_chunk))
                        model.ctc_greedy(0, enc)
                        model.ctc_greedy(1, enc)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    k = min(len(units_all), int(30 * pos / 16000))
                    if k > k_prev:
                        synthesize_tail(vs, units_all[:k], k - k_prev, True, (rf + 8) if mode == "incremental" else 0, rf)
                        k_prev = k
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    t_enc += t1 - t0
                    t_voc += t2 - t1
            row[mode] = {"encoder_side_ms_total": round(1e3 * t_enc, 2), "vocoder_side_ms_total": round(1e3 * t_voc, 2),
                         "ms_per_call_mean": round(1e3 * (t_enc + t_voc) / row["policy_calls"], 3)}
        fi, ff = row["incremental"], row["full_recompute"]
        row["speedup_encoder_side"] = round(ff["encoder_side_ms_total"] / fi["encoder_side_ms_total"], 3)
        row["speedup_vocoder_side"] = round(ff["vocoder_side_ms_total"] / fi["vocoder_side_ms_total"], 3)
        row["speedup_total"] = round((ff["encoder_side_ms_total"] + ff["vocoder_side_ms_total"])
                                     / (fi["encoder_side_ms_total"] + fi["vocoder_side_ms_total"]), 3)
        res.append(row)
    return res


def streaming_mode(args, model, voc, lib, cfg, sd, vsd, vcfg):
    out = streaming_measure(model, voc, lib, cfg, args.segment_ms, args.utterances,
                            cpu_sd=None if args.no_cpu_baseline else (sd, vsd, vcfg), cpu_utts=4, long_seconds=(15, 30))
    emit_streaming(out)


def compact_streaming_line(out, detail=None):
    """--mode streaming: the compact stdout line (< LINE_BYTE_CAP) of a streaming_measure() result."""
    line = {k: out.get(k) for k in ("metric", "mode", "value", "unit", "higher_is_better", "n_gpus", "dtype", "data", "segment_ms",
                                    "incremental_speedup_over_full_recompute")}
    line["config"] = {"workload": out["config"]["workload"]}
    for name in ("incremental", "full_recompute", "incremental_launch_per_op_mt", "incremental_search_to_cap"):
        v = out.get(name)
        if v:
            line[name] = {**{k: v[k] for k in ("rtfx_compute", "utterances", "policy_calls", "ms_per_policy_call_mean", "ms_per_policy_call_p95",
                                               "gemm_class_launches_per_policy_call", "RTF_CA", "StartOffset_CA_ms", "EndOffset_CA_ms") if k in v},
                          "ms_per_read_call_mean": v["calls_by_kind"]["ms_per_read_call_mean"],
                          "ms_per_write_call_mean": v["calls_by_kind"]["ms_per_write_call_mean"],
                          "ms_per_source_finished_call_mean": v["calls_by_kind"]["ms_per_source_finished_call_mean"]}
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = None if not cb else {k: cb[k] for k in ("value", "unit", "cores", "kind", "ms_per_policy_call_mean", "RTF_CA")}
    if out.get("long_prefix_sweep"):
        line["long_prefix_speedup_total"] = {str(r["source_s"]): r["speedup_total"] for r in out["long_prefix_sweep"]}
    line["detail"] = detail
    return line


def emit_streaming(out):
    """--mode streaming: the full report to the sidecar / stderr, the compact line to stdout."""
    path = detail_path()
    try:
        json.dump(out, open(path, "w"), indent=1)
    except OSError:
        path = None
    print("bench.py full result (indented on purpose: no stderr line can be taken for the bench line):\n" + json.dumps(out, indent=1), file=sys.stderr, flush=True)
    _emit(compact_streaming_line(out, path and os.path.relpath(path, ROOT)))


def _pmc_file():
    """Newest committed PMC summary (profiles/rNN_pmc_traffic*.json, written by tools/pmc_traffic.py)."""
    import glob
    names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")))
    return names[-1] if names else None


PMC_FILE = _pmc_file()


_REAL_STDOUT = None


def _claim_stdout():
    """The bench line is the ONLY thing this command may put on stdout.  Libraries underneath print there from C / C++ (RCCL's
    version banner sits in the C stdio buffer until the process exits and would land AFTER the line; "[Gloo] Rank 0 is connected"),
    so for the whole run file descriptor 1 points at stderr and the line goes to a private duplicate of the original stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def _emit(obj):
    line = json.dumps(obj)
    if _REAL_STDOUT is not None:
        _REAL_STDOUT.write(line + "\n")
        _REAL_STDOUT.flush()
    else:
        print(line, flush=True)


LINE_BYTE_CAP = 6000      # the driver could not parse a 21-KB line; the contract line stays under 6 KB, the rest is a sidecar
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_issued", "launches", "avg_launch_us", "algo_gflop_per_launch",
                 "algo_mbytes_per_launch", "traffic", "traffic_over_algorithmic", "mfma_util_pct")


def _pick(d, keys):
    return None if not d else {k: d[k] for k in keys if k in d}


def compact_line(full):
    """The ONE stdout line of the bench contract, built from the full result dict: the contract keys, `roofline` and `cpu_baseline` as
    plain numbers, and one number per optional leg.  Everything else (dispatch table, process census, traffic detail, the streaming
    reports, notes) lives in the sidecar (`detail_path()`) and on stderr.  Never longer than LINE_BYTE_CAP bytes: optional keys are
    dropped (and named in `dropped`) before the cap is passed -- tests/test_bench_line_cpu.py holds the bound for N = 1 and N = 8."""
    cfgd = full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "utterances_per_sec",
                                     "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfgd[k] for k in ("workload", "utterances_per_ragged_batch", "concurrent_streams_per_gpu", "utterances_per_gpu",
                                           "audio_seconds_per_gpu", "parallelism") if k in cfgd}
    if full.get("dry_plan"):
        line["dry_plan"] = True
        for k in ("steps_per_gpu", "utterances_per_step", "planned_audio_seconds", "planned_utterances", "stand_in_wall_max_s", "self_launched"):
            line[k] = full.get(k)
    r = full.get("roofline")
    if r:
        rr = _pick(r, ROOFLINE_KEYS)
        td = r.get("traffic_detail") or {}
        rr.setdefault("mfma_util_pct", td.get("mfma_util_pct"))
        rr["traffic_same_kernel_sources"] = td.get("same_kernel_sources")
        rr["measured_on"] = "one-stream replay of the timed batches" if "replay" in str(r.get("measured_on", "")) else "timed region"
        line["roofline"] = rr
    else:
        line["roofline"] = None
    fam = full.get("roofline_family")
    if fam:
        line["roofline_family"] = _pick(fam, ("achieved", "frac", "frac_issued", "launches"))
    cb = full.get("cpu_baseline")
    if cb:
        oc = cb.get("oracle_check") or {}
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": "x real-time", "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": f"{len(cb.get('per_utterance') or [])} utterances (short / median / long) of the same workload, "
                                          "median of timed passes",
                                "oracle_check": _pick(oc, ("utterances", "argmax_rows_compared", "ids_identical_to_float32_oracle", "near_tie_rows"))}
    else:
        line["cpu_baseline"] = None
    line["near_tie_rows"] = full.get("near_tie_rows")
    pi = full.get("pack_invariance")
    line["pack_invariance"] = None if not pi else {"alone_equals_in_pack_bitwise": pi.get("alone_equals_in_pack_bitwise")}
    line["per_rank"] = [{k: p[k] for k in ("rank", "wall_s", "audio_s", "utterances")} for p in (full.get("per_rank") or [])]
    rc = full.get("rccl") or full.get("comm")
    line["rccl"] = None if not rc else ({"error": str(rc["error"])[:120]} if "error" in rc else _pick(rc, ("backend", "world", "results_ok", "barrier_us")))
    for k in ("streaming_320ms", "multilingual", "soak", "bf16x3"):
        v = full.get(k)
        if v:
            line[k] = v.get("value") if not v.get("skipped") else None
    st = full.get("streaming_320ms")
    if st and st.get("incremental"):
        line["streaming_320ms_read_call_ms"] = (st["incremental"].get("calls_by_kind") or {}).get("ms_per_read_call_mean")
        line["streaming_320ms_launches_per_call"] = st["incremental"].get("gemm_class_launches_per_policy_call")
    hb = full.get("hbm")
    if hb:
        line["hbm_in_use_gb"] = hb.get("in_use_after_the_timed_region_gb")
    line["stream_k_spin_timeouts"] = full.get("stream_k_spin_timeouts")
    line["detail"] = full.get("detail_file")
    dropped = []
    for k in ("detail", "hbm_in_use_gb", "streaming_320ms_launches_per_call", "streaming_320ms_read_call_ms", "bf16x3", "roofline_family",
              "stream_k_spin_timeouts", "soak", "multilingual", "streaming_320ms", "pack_invariance", "per_rank"):
        if len(json.dumps(line)) < LINE_BYTE_CAP:
            break
        if k in line:
            del line[k]
            dropped.append(k)
            line["dropped"] = dropped
    return line


def detail_path():
    return os.environ.get("SS_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")


def emit_result(full):
    """Sidecar + stderr get the full result; stdout gets compact_line(full)."""
    path = detail_path()
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        full["detail_file"] = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError as e:
        full["detail_file"] = None
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    print("bench.py full result (indented on purpose: no stderr line can be taken for the bench line):\n" + json.dumps(full, indent=1), file=sys.stderr, flush=True)
    _emit(compact_line(full))


class _stdout_to_stderr:
    """The bench line is the ONLY thing this command may print on stdout: communicator start-up banners of the libraries
    underneath ("[Gloo] Rank 0 is connected ...") are sent to stderr (fd-level, so C++ prints are covered)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)          # C stdio buffers of the libraries (the banner must not surface later on the restored fd)
        except Exception:  # noqa: BLE001
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` typed without a launcher (fairseq's own entry does the same: distributed_utils.call_main
    spawns one process per device, fairseq/fairseq/distributed/utils.py:344-380): run the same command line under
    torch.distributed.run, one rank per GPU, and leave with its exit code (non-zero if any rank failed)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this host driver (RCCL needs it)
    env["SS_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def comm_probe(dist, device, reps=20):
    """The collectives of the utterance-DP path (one barrier before the timed region, MAX / SUM all-reduce and
    one all-gather of three doubles after it), timed on an initialised process group -- untimed for `value`."""
    def sync():
        if str(device).startswith("cuda"):
            torch.cuda.synchronize()

    def timed(fn):
        fn()
        sync()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        sync()
        return round(1e6 * (time.perf_counter() - t) / reps, 2)

    world = dist.get_world_size()
    t3 = torch.tensor([1.0 + dist.get_rank(), 2.0, 3.0], dtype=torch.float64, device=device)
    outs = [torch.zeros_like(t3) for _ in range(world)]
    res = {"backend": dist.get_backend(), "world": world,
           "barrier_us": timed(dist.barrier),
           "all_reduce_max_3xf64_us": timed(lambda: dist.all_reduce(t3.clone(), op=dist.ReduceOp.MAX)),
           "all_reduce_sum_3xf64_us": timed(lambda: dist.all_reduce(t3.clone(), op=dist.ReduceOp.SUM)),
           "all_gather_3xf64_us": timed(lambda: dist.all_gather(outs, t3))}
    mx, sm = t3.clone(), t3.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    dist.all_gather(outs, t3)
    sync()
    res["results_ok"] = bool(float(mx[0]) == float(world) and float(sm[1]) == 2.0 * world
                             and [float(o[0]) for o in outs] == [1.0 + r for r in range(world)])
    return res


def rccl_probe_main():
    """`bench.py --rccl-probe`: initialise the nccl (= RCCL) backend on this GPU as a world of one and run the DP path's
    collectives; printed as one JSON object.  bench.py runs this as a SUBPROCESS with a time-out after its measurement at
    N = 1 (a communicator that fails or hangs on some box must not cost the bench line), SS_FORCE_DIST=1 runs the same
    inside the bench process instead, around the timed region."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            timeout=datetime.timedelta(seconds=60), device_id=torch.device("cuda", 0))
    dist.barrier()
    torch.cuda.synchronize()
    init_ms = 1e3 * (time.perf_counter() - t0)
    out = comm_probe(dist, "cuda:0")
    out["init_plus_first_barrier_ms"] = round(init_ms, 1)
    try:
        out["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        pass
    dist.destroy_process_group()
    _emit(out)


def rccl_probe_subprocess(timeout_s=120):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--rccl-probe"], env=env, capture_output=True, text=True,
                           timeout=timeout_s)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                out = json.loads(line)
                out["how"] = "subprocess `bench.py --rccl-probe` on the same GPU after the measurement (world of one)"
                return out
        return {"error": f"rc {r.returncode}: " + (r.stderr.strip().splitlines() or ["no output"])[-1][:300]}
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout_s} s"}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def dry_plan(args, rank, world):
    """`--dry-plan`: the N-rank control path with no GPU -- every rank forms its share of the plan (workload.bench_plan, the
    function the real run uses), meets the barrier, and the (wall, audio seconds, utterances) statistics go through the same
    MAX / SUM all-reduce + all-gather over gloo (tests/test_dp_gloo.py runs `bench.py --gpus 2 --dry-plan` as typed)."""
    dist = None
    if world > 1:
        import torch.distributed as dist
        import datetime
        with _stdout_to_stderr():
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=180))    # a rendezvous that cannot complete fails, it does not wait 30 min
            dist.barrier()
    Ksteps, Bsz = max(1, args.steps), max(1, args.batch)
    mine, groups = workload.bench_plan(Ksteps, Bsz, rank, world, bucket=not args.no_length_bucketing, warm=3, strong=args.scaling == "strong")
    timed_ids = [i for g in groups for i in g]
    audio = sum(mine[i].seconds for i in timed_ids)
    if os.environ.get("SS_BENCH_DRY_FAIL_RANK") == str(rank):     # tests: a dying rank must fail the whole command
        raise SystemExit(3)
    if dist is not None:
        dist.barrier()
    wall = 1.0 + 0.001 * rank          # a stand-in wall time: the MAX must pick the last rank's
    placement = dp.pin_rank_to_gpu_numa(rank, world, pci_bus_id=os.environ.get("SS_BENCH_DRY_PCI", "0000:00:00.0"), apply=False)
    per_rank = dp.gather_per_rank(dist, wall, audio, float(len(timed_ids)), placement=placement)
    wall_max, audio_tot, nutt = dp.reduce_stats(dist, wall, audio, float(len(timed_ids)))
    comm = comm_probe(dist, "cpu", reps=5) if dist is not None else None
    if rank == 0:
        full = {"metric": METRIC, "dry_plan": True, "value": None, "unit": "x real-time", "n_gpus": world, "steps": Ksteps, "warmup": max(0, args.warmup),
                "ms_per_step": None, "utterances_per_sec": None, "higher_is_better": True, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "scaling": args.scaling, "steps_per_gpu": len(groups), "utterances_per_step": Bsz,
                "config": {"workload": WORKLOAD, "utterances_per_ragged_batch": Bsz, "parallelism": f"utterance-dp{world}"},
                "planned_audio_seconds": round(audio_tot, 2), "planned_utterances": int(nutt),
                "stand_in_wall_max_s": wall_max, "per_rank": per_rank, "comm": comm,
                "self_launched": bool(os.environ.get("SS_BENCH_SELF_LAUNCHED"))}
        if os.environ.get("SS_BENCH_DRY_DETAIL") == "1":
            emit_result(full)                   # tests: the sidecar path of the real run
        else:
            _emit(compact_line(full))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16, help="timed steps per GPU (one step = one ragged batch of --batch utterances)")
    ap.add_argument("--warmup", type=int, default=3, help="untimed warm-up steps per GPU before the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bf16x3-line", action="store_true", help="also run the optional split-bf16 vocoder-conv leg (same batches; never the headline)")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket the dominant kernel with HIP events")
    ap.add_argument("--bracket-ab", action="store_true", help="also run the event-bracket on / off A/B of the timed region (five more region passes)")
    ap.add_argument("--full", action="store_true",
                    help="every optional leg at its long form (bracket A/B, bf16x3, 8-stream batch-1 pass, launch-per-op latency A/B, streaming: three "
                         "agent configurations + CPU leg + long-prefix sweep); the default keeps the command under about a minute")
    ap.add_argument("--no-multilingual", action="store_true", help="skip the configs[4] sub-object (fr/es/de weight sets resident together)")
    ap.add_argument("--no-streaming-line", action="store_true", help="skip the configs[2] sub-object (320-ms agent policy() loop + its CPU baseline)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): --steps batches PER GPU; strong: --steps batches in TOTAL (configs[3] literally: "
                         "steps x batch utterances / N per GPU)")
    ap.add_argument("--streams", type=int, default=4,
                    help="concurrent HIP streams per GPU, a scratch set each (4: within 1 %% of 8 at half the HBM -- profiles/r06_stream_sweep.txt)")
    ap.add_argument("--no-latency-pass", action="store_true", help="skip the single-stream latency measurement")
    ap.add_argument("--no-soak", action="store_true", help="skip the >= 100-step sustained passes after the timed region")
    ap.add_argument("--batch", type=int, default=128,
                    help="utterances packed per ragged batch (1 = the one-utterance-at-a-time entry points)")
    ap.add_argument("--mode", choices=("offline", "streaming"), default="offline",
                    help="offline: the headline metric (default, what the driver runs); streaming: BASELINE.json configs[2] -- the "
                         "SimulEval agent's policy() loop on 320-ms chunks, one utterance at a time")
    ap.add_argument("--segment-ms", type=int, default=320, help="--mode streaming: source segment size")
    ap.add_argument("--utterances", type=int, default=12, help="--mode streaming: utterances per configuration")
    ap.add_argument("--no-length-bucketing", action="store_true",
                    help="form ragged batches in arrival order instead of sorted by source length")
    ap.add_argument("--dry-plan", action="store_true",
                    help="no GPU: form every rank's plan and run the barrier + statistics reductions over gloo (the N > 1 control path)")
    ap.add_argument("--rccl-probe", action="store_true", help="internal: world-of-one RCCL initialisation + the DP collectives, one JSON object")
    ap.add_argument("--no-rccl-probe", action="store_true", help="skip the RCCL probe of the N = 1 line")
    args = ap.parse_args()
    if args.full:
        args.bracket_ab = args.bf16x3_line = True
    args.no_bracket_ab, args.no_bf16x3_line = not args.bracket_ab, not args.bf16x3_line
    t_process = time.perf_counter()
    legs = {}                                  # seconds per leg of this command (sidecar)
    if args.rccl_probe:
        _claim_stdout()
        return rccl_probe_main()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))        # typed without a launcher: one rank per GPU under torch.distributed.run
    _claim_stdout()                              # from here on stdout carries the one JSON line and nothing else
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; running with the launcher's world size", file=sys.stderr)
    if args.dry_plan:
        return dry_plan(args, rank, world)
    # one process per GPU; the modulo only matters for the 2-ranks-on-1-GPU smoke test of this code path
    # (SS_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2)
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    placement = dp.pin_rank_to_gpu_numa(local_rank, world)      # before any launching thread exists (children inherit the mask)
    dist = None
    force_dist = world == 1 and os.environ.get("SS_FORCE_DIST") == "1"   # a world of one still goes through RCCL: init, barrier, reductions
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("SS_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        kw = {}
        if "MASTER_ADDR" not in os.environ:                           # SS_FORCE_DIST=1 typed without a launcher
            kw = {"init_method": f"tcp://127.0.0.1:{_free_port()}", "rank": 0, "world_size": 1}
        with _stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **kw)
            else:
                dist.init_process_group(backend, **kw)
            dist.barrier()          # the communicator is built lazily: pay for it (and its banners) here, not at the timed barrier

    cfg, vcfg = ModelConfig(), VocoderConfig()
    sd = synth.make_model_state_dict(0, cfg)
    vsd = synth.make_vocoder_state_dict(0, vcfg)
    dev = f"cuda:{local_rank}"
    model = HipModel(sd, cfg, device=dev)
    voc = HipVocoder(vsd, vcfg, device=dev)
    lib = L.load()
    legs["setup_weights_and_contexts"] = round(time.perf_counter() - t_process, 2)
    if args.mode == "streaming":
        return streaming_mode(args, model, voc, lib, cfg, sd, vsd, vcfg)

    Ksteps, Wsteps, Bsz = max(1, args.steps), max(0, args.warmup), max(1, args.batch)
    strong = args.scaling == "strong"
    Wn = 3                                   # single-utterance warm-ups (first-touch of every code path)
    mine, groups = workload.bench_plan(Ksteps, Bsz, rank, world, bucket=not args.no_length_bucketing, warm=Wn, strong=strong)
    K = sum(len(g) for g in groups)          # timed utterances on this rank
    steps_here = len(groups)                 # timed steps on this rank (= --steps unless --scaling strong)
    Kpool = len(mine) - Wn                   # distinct synthetic utterances on this rank (cycled beyond 4096)
    pcms = [torch.from_numpy(synth.synth_pcm(1234 + u.idx, u.n_samples)).to(dev) for u in mine]
    timed_ids = [i for g in groups for i in g]
    torch.cuda.synchronize()

    for u, p in zip(mine[:Wn], pcms[:Wn]):
        run_utterance(model, voc, p, u)
    torch.cuda.synchronize()

    # dominant kernel classes: decided from an untimed profiled pass over one utterance
    dom, dom_conv = None, None
    if not args.no_prof:   # untimed; uses the first utterances of this rank whatever --warmup is
        ncls = lib.ss_prof_num_classes()
        lib.ss_prof_reset()
        lib.ss_prof_enable((1 << ncls) - 1)
        if args.batch > 1:   # classify on the shapes the timed region will run
            nb = min(args.batch, len(mine))
            run_batch(model, voc, torch.cat(pcms[:nb]), mine[:nb])
        else:
            run_utterance(model, voc, pcms[0], mine[0])
        torch.cuda.synchronize()
        times = []
        for c in range(ncls):
            ms, fl, n, by = C.c_double(), C.c_double(), C.c_int64(), C.c_double()
            lib.ss_prof_read(c, C.byref(ms), C.byref(fl), C.byref(n), C.byref(by))
            times.append(ms.value)
        dom = max(range(ncls), key=lambda c: times[c])
        others = [c for c in range(ncls) if c != dom and times[c] > 0 and lib.ss_prof_class_name(c).decode().startswith(("conv_", "resblock_fused", "ffn_fused", "rt_linear"))]
        dom_conv = max(others, key=lambda c: times[c]) if others else None
        lib.ss_prof_enable(0)
        lib.ss_prof_reset()

    # single-stream latency pass (untimed for `value`; reported as latency_ms_single_stream): one utterance at a time on the
    # primary context, whose MT decode step is the persistent one-launch form by default (engine.HipModel; the batched /
    # multi-stream paths of this file never use it) -- and the same utterances with the launch-per-op step for comparison
    pmt_default = int(getattr(model, "persistent_mt", 0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nlat = 1 if args.no_latency_pass else min(Kpool, 8)
    lat_idx = list(range(Wn, Wn + Kpool))[::max(1, Kpool // nlat)][:nlat]   # evenly spread over the length-sorted pool
    for i in lat_idx:
        run_utterance(model, voc, pcms[i], mine[i])
    torch.cuda.synchronize()
    single_ms = 1e3 * (time.perf_counter() - t0) / nlat
    single_rtfx = sum(mine[i].seconds for i in lat_idx) / (single_ms * 1e-3 * nlat)
    pmt_after = int(getattr(model, "persistent_mt", 0))      # 0 if the persistent step timed out and the context fell back
    single_ms_lpo = None
    if args.full and not args.no_latency_pass and hasattr(model, "set_persistent_mt_step"):
        model.set_persistent_mt_step(0)
        run_utterance(model, voc, pcms[lat_idx[0]], mine[lat_idx[0]])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in lat_idx:
            run_utterance(model, voc, pcms[i], mine[i])
        torch.cuda.synchronize()
        single_ms_lpo = 1e3 * (time.perf_counter() - t0) / nlat
    if hasattr(model, "set_persistent_mt_step"):
        model.set_persistent_mt_step(0)       # from here on this context runs next to seven others: launch-per-op (see HipModel.new_context)

    # S concurrent utterance streams: worker threads (ctypes releases the GIL inside the C ABI),
    # each with its own HIP stream and its own scratch/KV-cache context over the shared weights
    S = max(1, args.streams)
    legs["warmups_and_latency_pass"] = round(time.perf_counter() - t_process - legs["setup_weights_and_contexts"], 2)
    t_leg = time.perf_counter()
    import threading
    hbm_free0, hbm_total = torch.cuda.mem_get_info(dev)     # before any scratch context has grown (weights + the packed PCM are resident)
    # one scratch set per stream (activations, KV caches, stream-K state: everything a call mutates), shared by the model and the vocoder
    # handle of the stream -- and by the other languages' handles in the configs[4] leg
    from streamspeech_amd.engine import Scratch
    scratches = [Scratch(dev) for _ in range(S)]
    model.bind_scratch(scratches[0])
    voc.bind_scratch(scratches[0])
    ctxs = [(model, voc)] + [(model.new_context(scratch=scratches[wi]), voc.new_context(scratch=scratches[wi])) for wi in range(1, S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    # one work item per timed step: the utterances of the ragged batch + their packed PCM (built before the timed region)
    if Bsz == 1:
        work = [(mine[g[0]], pcms[g[0]]) for g in groups]
    else:
        work = [([mine[i] for i in g], torch.cat([pcms[i] for i in g])) for g in groups]
    longest = max(range(len(mine)), key=lambda i: mine[i].seconds)
    next_idx = [0]
    lock = threading.Lock()
    ready_evt = threading.Barrier(S + 1)   # every context warmed up (nothing of the warm-up may be profiled or timed)
    start_evt = threading.Barrier(S + 1)   # released at t0, after the counters were reset and switched on
    samples = [0] * S
    errors = []

    def worker(wi):
        try:
            torch.cuda.set_device(local_rank)
            m, v = ctxs[wi]
            with torch.cuda.stream(streams[wi]):
                run_utterance(m, v, pcms[longest], mine[longest])   # warm this context at the largest shapes
                if Bsz > 1 and work:
                    big = max(work, key=lambda w: w[1].numel())
                    run_batch(m, v, big[1], big[0])
                streams[wi].synchronize()
                ready_evt.wait()
                start_evt.wait()
                while True:
                    with lock:
                        i = next_idx[0]
                        next_idx[0] += 1
                    if i >= len(work):
                        break
                    if Bsz == 1:
                        u, p = work[i]
                        wav, _, _, _ = run_utterance(m, v, p, u)
                        samples[wi] += wav.numel()
                    else:
                        us, pk = work[i]
                        wavs, _, _, _ = run_batch(m, v, pk, us)
                        samples[wi] += sum(w.numel() for w in wavs)
                streams[wi].synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            for ev in (ready_evt, start_evt):
                try:
                    ev.abort()
                except Exception:  # noqa: BLE001
                    pass

    for wb in work[:Wsteps]:                 # W untimed warm-up steps (on top of the per-context warm-up inside worker())
        if Bsz == 1:
            run_utterance(model, voc, wb[1], wb[0])
        else:
            run_batch(model, voc, wb[1], wb[0])
    torch.cuda.synchronize()
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(S)]
    for t in threads:
        t.start()
    ready_evt.wait()                         # all per-context warm-ups are done and synchronised
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if dom is not None:                      # counters cover exactly the timed region
        lib.ss_prof_reset()
        lib.ss_prof_enable((1 << dom) | ((1 << dom_conv) if dom_conv is not None else 0))
    sk_err0 = int(lib.ss_debug_sk_errors())
    start_evt.wait()
    t0 = time.perf_counter()
    t0_mono_ns = time.monotonic_ns()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    t1_mono_ns = time.monotonic_ns()
    legs["context_warmups_and_timed_region"] = round(time.perf_counter() - t_leg, 2)
    t_leg = time.perf_counter()
    lib.ss_prof_enable(0)
    if errors:
        raise errors[0]

    hbm_free1, _ = torch.cuda.mem_get_info(dev)
    hbm = {"total_gb": round(hbm_total / 2 ** 30, 1), "in_use_after_the_timed_region_gb": round((hbm_total - hbm_free1) / 2 ** 30, 1),
           "scratch_sets": S, "scratch_set_gb": [round(sc.bytes() / 2 ** 30, 2) for sc in scratches],
           "note": "a scratch set (ss_scratch: one per stream, model + vocoder side) keeps the activations of the largest pack it has run until "
                   "ss_scratch_trim; ss_scratch_set_cap bounds it"}
    audio = sum(mine[i].seconds for i in timed_ids)
    per_rank = dp.gather_per_rank(dist, wall, audio, float(K), device=dev, placement=placement)
    wall, audio, nutt = dp.reduce_stats(dist, wall, audio, float(K), device=dev)
    rccl = None
    if dist is not None:                     # the DP path's collectives on the live communicator (untimed for `value`)
        rccl = comm_probe(dist, dev if dist.get_backend() == "nccl" else "cpu")
        rccl["how"] = ("this process group: the barriers around the timed region and the MAX / SUM / all-gather of the statistics above went "
                       "through it" + ("; SS_FORCE_DIST=1 (world of one)" if force_dist else ""))

    def read_class(c):
        ms, fl, n, by = C.c_double(), C.c_double(), C.c_int64(), C.c_double()
        lib.ss_prof_read(c, C.byref(ms), C.byref(fl), C.byref(n), C.byref(by))
        return ms.value, fl.value, n.value, by.value

    def roofline_of(c):
        ms, fl, n, by = read_class(c)
        if n == 0 or ms <= 0:
            return None
        name = lib.ss_prof_class_name(c).decode()
        traffic, traffic_detail = None, None
        try:   # PMC pass is a separate rocprofv3 run (tools/pmc_traffic.py -> profiles/); per-launch bytes with the guide's gfx950 correction
            pm = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE))) if PMC_FILE else {}
            kv = pm.get("classes", {}).get(name)
            if kv and kv.get("join_ok", True) is False:
                kv = None                                  # counter pass and census disagree on the launch count: no figure
            if kv:
                traffic = round(kv["hbm_mbytes_per_launch_corrected"] * 1e6)          # HBM-side bytes per launch (counters)
                from tools.pmc_traffic import csrc_file_sha16, csrc_sha16
                then, now = pm.get("csrc_files_sha16") or {}, csrc_file_sha16()
                changed = sorted(f for f in set(then) | set(now) if then.get(f) != now.get(f)) if then else None
                kfile = ("conv_c64w.hip" if name.startswith(("conv_c64w", "conv_c128w", "conv_c256w", "conv_c32w")) else
                         name.split("<")[0].replace("conv_gemm", "gemm").replace("smallm_gemm", "gemm").replace("rt_linear", "rtlin").replace("ffn_fused", "ffn")
                         .replace("resblock_fused", "resblock") + ".hip")
                dominant = (kfile, "gemm.hpp", "gemm.hip", "common.hpp", "dispatch.hpp")   # the class's kernel, its argument block, dispatch + workspace
                traffic_detail = {"measured_by": "a separate rocprofv3 --pmc run of an earlier process (NOT this run)",
                                  "pmc_run_kernel_sources_sha16": pm.get("csrc_sha16"), "this_build_kernel_sources_sha16": csrc_sha16(),
                                  "same_kernel_sources": pm.get("csrc_sha16") == csrc_sha16(),
                                  "files_changed_since_pmc_run": changed,
                                  "dominant_kernel_sources_unchanged": (None if changed is None else not any(f in changed for f in dominant)),
                                  "mbytes_per_launch": kv["hbm_mbytes_per_launch_corrected"],
                                  "algorithmic_mbytes_per_launch_of_the_pmc_run": kv.get("algo_mbytes_per_launch"),
                                  "traffic_over_algorithmic": kv.get("traffic_over_algorithmic"),
                                  "mfma_util_pct": kv.get("mfma_util_pct"),
                                  "source": f"profiles/{PMC_FILE}: " + pm.get("note", "")}
        except Exception:  # noqa: BLE001
            pass
        common = {"kernel": name, "launches": int(n), "avg_launch_us": round(1e3 * ms / n, 2),
                  "algo_gflop_per_launch": round(fl / n / 1e9, 4), "algo_mbytes_per_launch": round(by / n / 1e6, 3),
                  "traffic": traffic,
                  # `traffic` comes from a separate PMC run with its OWN launch mix: divide it by THAT run's algorithmic bytes (next
                  # two keys), never by algo_mbytes_per_launch of this run
                  "traffic_over_algorithmic": None if not traffic_detail else traffic_detail["traffic_over_algorithmic"],
                  "algorithmic_mbytes_per_launch_of_the_pmc_run": None if not traffic_detail else traffic_detail["algorithmic_mbytes_per_launch_of_the_pmc_run"],
                  "traffic_detail": traffic_detail, "kernel_time_over_wall": round(ms * 1e-3 / wall, 3)}
        if name.startswith("smallm"):
            # M <= 128 projections / M = 1 decode GEMVs stream their weights once: HBM-side roofline
            ach = by / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4), **common}
        ach = fl / (ms * 1e-3) / 1e12
        if name.startswith(("conv_c64w", "conv_c128w", "conv_c256w", "conv_c32w")):
            # Winograd F(2,3) forms (csrc/conv_c64w.hip, 32 ... 256 channels): the census counts the conv's ALGORITHMIC (direct-form) FLOPs, as for
            # every class; the kernel issues only (4 (k div 3) + [0, 2, 3][k mod 3]) / (2 k) of them as MFMAs (4/6 at k = 3, 10/14 at k = 7, 15/22 at
            # k = 11), so `frac` can pass 1 -- it is not the matrix cores' busy fraction; `frac_issued` (the library's count of issued MFMA FLOPs) is
            iss = C.c_double()
            lib.ss_prof_read_issued(c, C.byref(iss))
            common["issued_mfma_tflops"] = round(iss.value / (ms * 1e-3) / 1e12, 3)
            common["frac_issued"] = round(iss.value / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
            common["note"] = ("Winograd F(2,3) with F(2,1) / F(2,2) tail groups: `frac` = algorithmic (direct-form) FLOPs over time against the FP32-MFMA peak; "
                              "the MFMAs issued are 4/6 (k = 3), 10/14 (k = 7), 15/22 (k = 11) of them = `frac_issued`, the matrix cores' busy fraction")
        return {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), **common}

    in_region = roofline_of(dom) if dom is not None else None
    in_region_conv = roofline_of(dom_conv) if dom_conv is not None and dom_conv != dom else None
    sk_err_timed = int(lib.ss_debug_sk_errors()) - sk_err0
    if sk_err_timed:
        raise RuntimeError(f"{sk_err_timed} stream-K bounded waits timed out inside the timed region")

    # Roofline of the dominant kernel.  With S > 1 the timed region keeps S streams in flight, so a launch bracketed by
    # events there shares the chip with the other streams' kernels (and queues behind them): that bracket measures
    # scheduling, not the kernel.  The per-launch figure is therefore taken on the SAME batches replayed on ONE stream
    # (HIP events on the launch stream, every launch of the class; untimed for `value`) -- with --streams 1 the replay is
    # skipped and the timed region itself is the measurement.  `in_region` keeps what the S-stream region gave:
    # the event-bracket sums and the class's algorithmic FLOPs over the region's wall time.
    def region_stats(r):
        if not r:
            return None
        tot = r["algo_gflop_per_launch"] * r["launches"]
        return {"launches": r["launches"], "avg_event_bracket_us": r["avg_launch_us"],
                "algo_tflop_total": round(tot / 1e3, 3), "algo_tflops_over_wall": round(tot / 1e3 / wall, 3),
                "event_bracket_time_over_wall": r["kernel_time_over_wall"], "concurrent_streams": S}

    roofline, roofline_conv = in_region, in_region_conv
    roofline_family, dispatch = None, None
    WINO = [c for c in range(lib.ss_prof_num_classes()) if lib.ss_prof_class_name(c).decode().startswith(("conv_c64w", "conv_c128w", "conv_c32w", "conv_c256w"))]
    if rank == 0 and dom is not None and S > 1 and work and not os.environ.get("SS_BENCH_NO_REPLAY"):   # (ranks > 0 run no optional leg)   # (tools/jobs/*trace*: keep the trace to the timed region)
        lib.ss_prof_reset()
        mask = (1 << dom) | ((1 << dom_conv) if dom_conv is not None else 0)
        for c in WINO:
            mask |= 1 << c
        lib.ss_prof_enable(mask)
        lib.ss_prof_shape_log(1)
        t_rep = time.perf_counter()
        for wk in work:
            if Bsz == 1:
                run_utterance(model, voc, wk[1], wk[0])
            else:
                run_batch(model, voc, wk[1], wk[0])
        torch.cuda.synchronize()
        rep_wall = time.perf_counter() - t_rep
        lib.ss_prof_enable(0)
        lib.ss_prof_shape_log(0)
        # ---- which kernel class took which conv / linear shape in this run (the replay = the timed batches on one stream) ----
        need = lib.ss_prof_shape_dump(None, 0)
        buf = C.create_string_buffer(need + 16)
        lib.ss_prof_shape_dump(buf, need + 16)
        rows = [ln.split() for ln in buf.value.decode().splitlines()[1:]]
        tot_fl = sum(float(r[7]) * int(r[5]) for r in rows) or 1.0
        rows.sort(key=lambda r: -float(r[7]) * int(r[5]))
        dispatch = {"columns": ["kernel class", "N", "taps", "Cin", "operands (1 R, 2 R2, 4 twin out, 8 in-act, 16 out-act, 32 GLU, 64 ragged)",
                                "launches", "mean rows", "GFLOP per launch", "share of algorithmic FLOPs"],
                    "shapes": [[lib.ss_prof_class_name(int(r[0])).decode(), int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5]), int(float(r[6])),
                                float(r[7]), round(float(r[7]) * int(r[5]) / tot_fl, 4)] for r in rows[:48]],
                    "shapes_total": len(rows), "covered_by": "the timed batches replayed on one stream (same dispatch as the timed region)"}
        # ---- the Winograd F(2,3) family as ONE line: algorithmic fraction and the fraction of the matrix cores' issue rate ----
        fam_ms = fam_fl = fam_iss = 0.0
        fam_n, members = 0, {}
        for c in WINO:
            ms_c, fl_c, n_c, by_c = read_class(c)
            if n_c:
                iss = C.c_double()
                lib.ss_prof_read_issued(c, C.byref(iss))
                fam_ms, fam_fl, fam_iss, fam_n = fam_ms + ms_c, fam_fl + fl_c, fam_iss + iss.value, fam_n + n_c
                members[lib.ss_prof_class_name(c).decode()] = {"launches": int(n_c), "ms": round(ms_c, 2), "algorithmic_tflops": round(fl_c / ms_c / 1e9, 2),
                                                               "issued_tflops": round(iss.value / ms_c / 1e9, 2)}
        if fam_n:
            roofline_family = {"family": "conv_c64w_kernel<LRELU, DIL, CH> (Winograd F(2,3) on the dilation lattice; CH = 32 / 64 / 128 / 256)",
                               "bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "launches": int(fam_n),
                               "achieved": round(fam_fl / fam_ms / 1e9, 3), "frac": round(fam_fl / fam_ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4),
                               "issued_mfma_tflops": round(fam_iss / fam_ms / 1e9, 3),
                               "frac_issued": round(fam_iss / fam_ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4),
                               "kernel_time_over_replay_wall": round(fam_ms * 1e-3 / rep_wall, 3), "members": members,
                               "note": "`frac` counts the convs' ALGORITHMIC (direct-form) FLOPs; `frac_issued` counts the MFMAs the kernels issue "
                                       "(4 ceil(k/3) / (2 k) of them) -- the matrix cores' busy fraction; HIP events on the one-stream replay"}
        wall_keep, wall = wall, rep_wall
        roofline = roofline_of(dom)
        roofline_conv = roofline_of(dom_conv) if dom_conv is not None and dom_conv != dom else None
        wall = wall_keep
        for r, reg in ((roofline, in_region), (roofline_conv, in_region_conv)):
            if r:
                r["measured_on"] = (f"the {len(work)} timed batches replayed on one stream right after the timed region "
                                    f"(replay wall {rep_wall * 1e3:.1f} ms); HIP events on the launch stream around every launch")
                r["in_region"] = region_stats(reg)
    elif roofline:
        roofline["measured_on"] = "the timed region itself (one stream); HIP events on the launch stream around every launch"

    legs["one_stream_replay_for_the_roofline"] = round(time.perf_counter() - t_leg, 2)
    t_leg = time.perf_counter()

    # Optional SECOND line, never the headline (`value` above is the exact-f32 path): the same timed batches on the same S
    # streams with the C >= 64 vocoder convs contracted by three bf16 MFMAs per k-slice on operands split into
    # bf16(x) + bf16(x - bf16(x)) (ss_vocoder_set_bf16x3; f32 accumulation, everything else -- every argmax stage, the
    # duration predictor, the narrow vocoder stages -- stays f32).  Untimed for `value`; tests/test_bf16x3_gpu.py holds
    # its parity bars (durations identical, waveform RMS <= 1e-3 vs the FP32 oracle).
    def region_pass(pick=None):
        """All timed batches once more over the same S streams (untimed for `value`); pick(wi, i) -> (model, vocoder)
        handles for work item i on worker wi (default: the worker's own fr-en handles)."""
        nxt, errs = [0], []

        def w2(wi):
            try:
                torch.cuda.set_device(local_rank)
                with torch.cuda.stream(streams[wi]):
                    while True:
                        with lock:
                            i = nxt[0]
                            nxt[0] += 1
                        if i >= len(work):
                            break
                        m, v = ctxs[wi] if pick is None else pick(wi, i)
                        run_batch(m, v, work[i][1], work[i][0])
                    streams[wi].synchronize()
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=w2, args=(i,)) for i in range(S)]
        torch.cuda.synchronize()
        t_p = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t_p
        if errs:
            raise errs[0]
        return dt

    # What the HIP-event brackets (two classes, enabled inside the timed region so that `in_region` exists) cost the headline:
    # the same region once more with the brackets on and off (VERDICT r2 "quantify once").
    bracket_ab = None
    if world == 1 and dom is not None and work and Bsz > 1 and S > 1 and not args.no_bracket_ab:
        mask = (1 << dom) | ((1 << dom_conv) if dom_conv is not None else 0)
        region_pass()
        lib.ss_prof_enable(mask)
        on = min(region_pass(), region_pass())
        lib.ss_prof_enable(0)
        off = min(region_pass(), region_pass())
        lib.ss_prof_reset()
        bracket_ab = {"region_ms_with_event_brackets": round(1e3 * on, 2), "region_ms_without": round(1e3 * off, 2),
                      "with_over_without": round(on / off, 4), "note": "best of two region passes each, right after the timed region"}

    # Sustained figure inside the driver's own line (the timed region is ~1 s): the same region repeated until >= 100
    # more steps have run, one wall clock around all of them (untimed for `value`).
    soak = None
    try:
        if world == 1 and Bsz > 1 and S > 1 and work and not args.no_soak:
            reps = max(1, -(-(100 if args.full else 48) // len(work)))
            dts = [region_pass() for _ in range(reps)]
            soak = {"steps": reps * len(work), "value": round(reps * audio / sum(dts), 2), "unit": "x real-time", "ms_per_step": round(1e3 * sum(dts) / (reps * len(work)), 3),
                    "utterances": int(reps * nutt), "gpu_seconds": round(sum(dts), 3), "per_pass_value": [round(audio / d, 1) for d in dts],
                    "stream_k_spin_timeouts": int(lib.ss_debug_sk_errors()),
                    "note": f"the timed region's {len(work)} batches x {reps} passes on the same {S} streams right after the timed region"}
    except Exception as e:  # noqa: BLE001  (an optional leg never costs the line)
        soak = {"value": None, "skipped": f"soak: {type(e).__name__}: {e}"[:200]}

    legs["bracket_ab_and_soak"] = round(time.perf_counter() - t_leg, 2)
    t_leg = time.perf_counter()
    bf16x3_line = None
    try:
        if world == 1 and Bsz > 1 and work and not args.no_bf16x3_line:
            wav_f32 = [w.clone() for w in run_batch(model, voc, work[0][1], work[0][0])[0]]
            for _, v in ctxs:
                v.set_bf16x3(True)
            try:
                wav_x3 = run_batch(model, voc, work[0][1], work[0][0])[0]
                num = sum(float(((a - b).double() ** 2).sum()) for a, b in zip(wav_x3, wav_f32))
                den = sum(float((b.double() ** 2).sum()) for b in wav_f32)
                n_s = sum(b.numel() for b in wav_f32)
                region_pass()                                   # warm (first launches of the bf16 kernels on every context)
                dt3 = min(region_pass(), region_pass())
            finally:
                for _, v in ctxs:
                    v.set_bf16x3(False)
            bf16x3_line = {"value": round(audio / dt3, 2), "unit": "x real-time (audio s / wall s)", "utterances_per_sec": round(nutt / dt3, 3),
                           "ms_per_step": round(1e3 * dt3 / max(1, len(work)), 3), "dtype": "bf16x3 in the C >= 64 vocoder convs, f32 everywhere else",
                           "wav_rms_vs_f32_path": round((num / max(n_s, 1)) ** 0.5, 9), "wav_rel_rms_vs_f32_path": round((num / max(den, 1e-30)) ** 0.5, 9),
                           "note": "optional second line, NOT the headline: same timed batches and streams, best of two passes after one warm "
                                   "pass; durations and unit ids are identical by construction (only the vocoder's generator convs change)"}
    except Exception as e:  # noqa: BLE001  (an optional leg never costs the line)
        bf16x3_line = {"value": None, "skipped": f"bf16x3: {type(e).__name__}: {e}"[:200]}

    # BASELINE.json configs[4]: fr-en + es-en + de-en weight sets resident together (3 x (model + vocoder)), the SAME timed
    # batches dealt round-robin over the languages inside the same S-stream region (untimed for `value`); parity of exactly
    # this arrangement against the oracle: tests/test_multilingual_gpu.py.
    multilingual = None
    # Scratch sets are objects of their own since round 6 (ss_scratch_*): the es / de weight handles of a stream are bound to THAT stream's
    # scratch set, so any language runs on any stream with S scratch sets (round 5: a scratch set per (language, stream) pair, which no
    # longer fitted at packs of 128 and forced one language per stream).
    try:
        if world == 1 and Bsz > 1 and work and not args.no_multilingual:
            golden = os.path.join(ROOT, "tests", "golden")
            order = ("fr", "es", "de")
            per_lang = {"fr": {wi: ctxs[wi] for wi in range(S)}}
            weights_mb = 4e-6 * (model.blob.numel() + voc.blob.numel())
            hbm_before = torch.cuda.mem_get_info(dev)[0]
            for seed, lang in ((1, "es"), (2, "de")):
                g = np.load(os.path.join(golden, f"gcmvn_{lang}-en.npz"))      # configs/{es,de}-en/gcmvn.npz of the reference
                m = HipModel(synth.make_model_state_dict(seed, cfg), cfg, device=dev, cmvn_mean=g["mean"], cmvn_std=g["std"], scratch=scratches[0])
                v = HipVocoder(synth.make_vocoder_state_dict(seed, vcfg), vcfg, device=dev, scratch=scratches[0])
                weights_mb += 4e-6 * (m.blob.numel() + v.blob.numel())
                per_lang[lang] = {wi: ((m, v) if wi == 0 else (m.new_context(scratch=scratches[wi]), v.new_context(scratch=scratches[wi]))) for wi in range(S)}
            pick = lambda wi, i: per_lang[order[i % 3]][wi]   # noqa: E731   batch i is of language i % 3, on whichever stream takes it
            region_pass(pick)                                  # warm pass
            n_best = 2 if args.full else 1
            dt_ml = min(region_pass(pick) for _ in range(n_best))
            dt_1 = min(region_pass() for _ in range(n_best))  # the single-language set through the same queues, same moment
            multilingual = {"value": round(audio / dt_ml, 2), "unit": "x real-time (audio s / wall s)", "utterances_per_sec": round(nutt / dt_ml, 3),
                            "ms_per_step": round(1e3 * dt_ml / max(1, len(work)), 3), "languages": list(order),
                            "weights_mb": round(weights_mb, 1), "weight_handles": 3 * S, "scratch_sets": S,
                            "hbm_added_by_two_more_languages_gb": round((hbm_before - torch.cuda.mem_get_info(dev)[0]) / 2 ** 30, 2),
                            "single_language_same_method": {"value": round(audio / dt_1, 2), "ms_per_step": round(1e3 * dt_1 / max(1, len(work)), 3)},
                            "multilingual_over_single": round(dt_1 / dt_ml, 4),
                            "note": "BASELINE.json configs[4]: three weight sets (seeds 0/1/2 of the same architecture, es/de with the reference's "
                                    "gcmvn statistics) resident together; batch i is of language i % 3 and runs on whichever stream takes it -- every "
                                    f"stream's scratch set serves all three languages; best of {n_best} pass(es) after one warm pass"}
            del per_lang["es"], per_lang["de"]
            torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001  (an optional leg never costs the line)
        multilingual = {"value": None, "skipped": f"multilingual: {type(e).__name__}: {e}"[:200]}

    legs["bf16x3_and_multilingual"] = round(time.perf_counter() - t_leg, 2)
    t_leg = time.perf_counter()
    b1_rtfx = b1_ups = None
    if world == 1 and args.full and not args.no_latency_pass:
        S1, n1 = 8, min(Kpool, 64)
        sel = list(range(Wn, Wn + Kpool))[::max(1, Kpool // n1)][:n1]
        ctx1 = ctxs + [(model.new_context(), voc.new_context()) for _ in range(max(0, S1 - len(ctxs)))]
        str1 = [torch.cuda.Stream(device=dev) for _ in range(S1)]
        nxt, lk, bar = [0], threading.Lock(), threading.Barrier(S1 + 1)

        def w1(wi):
            torch.cuda.set_device(local_rank)
            m, v = ctx1[wi]
            with torch.cuda.stream(str1[wi]):
                run_utterance(m, v, pcms[longest], mine[longest])
                str1[wi].synchronize()
                bar.wait()
                while True:
                    with lk:
                        j = nxt[0]
                        nxt[0] += 1
                    if j >= len(sel):
                        break
                    run_utterance(m, v, pcms[sel[j]], mine[sel[j]])
                str1[wi].synchronize()

        th = [threading.Thread(target=w1, args=(i,)) for i in range(S1)]
        for t in th:
            t.start()
        bar.wait()
        t1 = time.perf_counter()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        w1s = time.perf_counter() - t1
        b1_rtfx = sum(mine[i].seconds for i in sel) / w1s
        b1_ups = len(sel) / w1s

    if rank == 0:
        out = {
            "metric": METRIC,
            "value": round(audio / wall, 2), "unit": "x real-time",
            "utterances_per_sec": round(nutt / wall, 3),
            "n_gpus": world, "steps": Ksteps, "warmup": Wsteps, "ms_per_step": round(1e3 * wall / max(1, steps_here), 3),
            "steps_per_gpu": steps_here,
            "utterances_per_step": Bsz, "ms_per_utterance": round(1e3 * wall / max(1, K), 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD,
                       "audio_seconds_per_gpu": round(sum(mine[i].seconds for i in timed_ids), 2), "utterances_per_gpu": K,
                       "lengths_pinned": "data-dependent lengths are pinned by the workload (SURVEY.md §8d): MT search forced to "
                                         "N = ceil(3.5 d) subwords + </s> (checked per batch inside the timed step), collapsed unit "
                                         "sequence cyclically resized to K = ceil(37 d), durations forced to the (1,1,2) cycle",
                       "length_bucketed_batches": (not args.no_length_bucketing) and Bsz > 1, "utterances_per_ragged_batch": Bsz, "concurrent_streams_per_gpu": S,
                       "parallelism": f"utterance-dp{world}"},
            "latency_ms_single_stream": round(single_ms, 3), "rtfx_single_stream": round(single_rtfx, 2),
            "latency_ms_single_stream_launch_per_op_mt_step": None if single_ms_lpo is None else round(single_ms_lpo, 3),
            "single_stream_mt_step": {"persistent_workgroups_default": pmt_default, "after_the_pass": pmt_after},
            "batch1_8streams": None if b1_rtfx is None else {"rtfx": round(b1_rtfx, 1), "utterances_per_sec": round(b1_ups, 1),
                                                             "note": "one utterance per call (no ragged packs), 8 concurrent streams, 64 utterances"},
            "pack_invariance": pack_invariance_check(model, work) if (Bsz > 1 and len(work) >= 2) else None,
            "stream_k_spin_timeouts": int(lib.ss_debug_sk_errors()),   # must be 0 (bounded waits of the stream-K fix-up); also asserted for the timed region
            "roofline": roofline,
            "roofline_family": roofline_family,
            "soak": soak,
            "hbm": hbm,
            "dispatch": dispatch,
            "bf16x3": bf16x3_line,
            "multilingual": multilingual,
            "event_bracket_perturbation": bracket_ab,
            "roofline_second_kernel": roofline_conv,
            "process_census": census(lib),
            "per_rank": per_rank,
            "host_placement": {**placement, "rule": "ranks of an N > 1 run are pinned to the CPUs local to their GPU's NUMA node (dp.pin_rank_to_gpu_numa)"},
            "multi_gpu_note": ("no N > 1 RCCL run has been made by the builder (no multi-GPU node in the build environment): the N > 1 path is "
                               "covered by gloo tests (tests/test_dp_gloo.py, incl. `--gpus 8 --dry-plan`) and by RCCL as a world of one (`rccl`)"),
            "rccl": rccl if (rccl is not None or args.no_rccl_probe) else rccl_probe_subprocess(),
            "self_launched": bool(os.environ.get("SS_BENCH_SELF_LAUNCHED")),
            "timed_region_monotonic_ns": [t0_mono_ns, t1_mono_ns],   # tools/trace_gaps.py: window of a rocprofv3 kernel trace
        }
        legs["pack_invariance_census_rccl_probe"] = round(time.perf_counter() - t_leg, 2)
        t_leg = time.perf_counter()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, vsd, cfg, vcfg, workload.make_utterances(Wn + Kpool + 1)[Wn:], hip_model=model, dev=dev,
                                               reps=5, warm=2, budget_s=30.0 if args.full else 25.0)     # ~10 s of CPU work on the GPU box's host
            out["near_tie_rows"] = out["cpu_baseline"]["oracle_check"]["near_tie_rows"]
        else:
            out["cpu_baseline"] = None
            out["near_tie_rows"] = None
        legs["cpu_baseline_and_oracle_check"] = round(time.perf_counter() - t_leg, 2)
        t_leg = time.perf_counter()
        out["streaming_320ms"] = None
        if world == 1 and not args.no_streaming_line:
            # BASELINE.json configs[2] inside the driver's line: the agent's policy() loop on 320-ms segments with the incremental state
            # (--full: also the reference's full recompute, the launch-per-op A/B, the CPU baseline on the same utterances and the
            # long-source sweep -- `bench.py --mode streaming` prints those as its own line).  A failing optional leg never costs the line.
            try:
                st = streaming_measure(model, voc, lib, cfg, 320, args.utterances if args.full else min(args.utterances, 6),
                                       cpu_sd=(sd, vsd, vcfg) if (args.full and not args.no_cpu_baseline) else None,
                                       cpu_utts=3, long_seconds=(15, 30) if args.full else (),
                                       configurations=None if args.full else ("incremental",))
                for k in ("metric", "mode", "n_gpus", "dtype", "data", "higher_is_better"):
                    st.pop(k, None)
                out["streaming_320ms"] = st
            except Exception as e:  # noqa: BLE001
                out["streaming_320ms"] = {"value": None, "skipped": f"{type(e).__name__}: {e}"[:200]}
        legs["streaming_320ms"] = round(time.perf_counter() - t_leg, 2)
        legs["total_since_argument_parsing"] = round(time.perf_counter() - t_process, 2)
        out["leg_seconds"] = legs
        emit_result(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
