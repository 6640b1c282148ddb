#!/usr/bin/env python
"""bench.py -- offline S2ST (BASELINE.json configs[1]) real-time factor on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, utterance-level data parallel)

A "step" is one synthetic utterance, batch 1, through the whole HIP hot path
(streamspeech_amd/workload.py): PCM already in HBM -> fbank+CMVN -> chunk-Conformer -> CTC x2 ->
AR MT greedy decode -> T2U + NAR unit decoder -> CTC collapse -> unit HiFi-GAN -> waveform in HBM.
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

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from streamspeech_amd import lib as L                      # noqa: E402
from streamspeech_amd import synth, workload               # noqa: E402
from streamspeech_amd.config import ModelConfig, VocoderConfig  # noqa: E402
from streamspeech_amd.engine import HipModel, HipVocoder   # noqa: E402
from streamspeech_amd.pipeline import mt_greedy, units_from_tokens  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"


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


def cpu_baseline(sd, vsd, cfg, vcfg, utts, budget_s=25.0):
    """The CPU oracle (torch fp32, all host cores) on a bounded sample of the same workload."""
    from oracle import kaldi_fbank as K
    from oracle import streamspeech_oracle as O
    osd, ovsd = O.SD(sd), O.SD(vsd)
    g_mean, g_std = np.zeros(80, np.float32), np.ones(80, np.float32)
    audio, wall, n = 0.0, 0.0, 0
    with torch.inference_mode():
        for i, u in enumerate(utts):
            pcm = synth.synth_pcm(1234 + u.idx, u.n_samples)
            t0 = time.perf_counter()
            fb = K.global_cmvn(K.fbank(pcm * np.float32(32768.0)), g_mean, g_std)
            enc = O.encoder_forward(osd, fb, cfg)
            O.ctc_head(osd, enc, "source_unigram", cfg)
            O.ctc_head(osd, enc, "ctc_target_unigram", cfg)
            toks = O.mt_greedy(osd, enc, cfg, max_new_tokens=u.n_mt)
            if toks[-1] == cfg.eos:
                toks = toks[:-1]
            feats = O.mt_decoder_features(osd, [cfg.eos] + toks, enc, cfg)
            logits = O.unit_decoder_logits(osd, O.t2u_encoder(osd, feats, cfg), cfg)
            units, _ = O.unit_ctc_generate(logits, cfg)
            units = workload.resize_units(units, u.n_units, u.idx)
            O.vocoder_forward(ovsd, units, vcfg, True, forced_dur=u.durations)
            dt = time.perf_counter() - t0
            if i == 0:
                continue  # warm-up utterance (thread pools, allocator)
            audio += u.seconds
            wall += dt
            n += 1
            if wall > budget_s:
                break
    return {"value": audio / wall, "unit": "x real-time (audio s / wall s)", "utterances_per_sec": n / wall,
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} utterances ({audio:.1f} s of audio) of the same synthetic workload, after 1 warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed utterances per GPU")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket the dominant kernel with HIP events")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg, vcfg = ModelConfig(), VocoderConfig()
    sd = synth.make_model_state_dict(0, cfg)
    vsd = synth.make_vocoder_state_dict(0, vcfg)
    dev = f"cuda:{local_rank}"
    model = HipModel(sd, cfg, device=dev)
    voc = HipVocoder(vsd, vcfg, device=dev)
    lib = L.load()

    K, Wn = args.steps, args.warmup
    all_utts = workload.make_utterances((K + Wn) * world)
    mine = workload.shard(all_utts, rank, world)            # weak scaling: K + W utterances per rank
    pcms = [torch.from_numpy(synth.synth_pcm(1234 + u.idx, u.n_samples)).to(dev) for u in mine]
    torch.cuda.synchronize()

    for u, p in zip(mine[:Wn], pcms[:Wn]):
        run_utterance(model, voc, p, u)
    torch.cuda.synchronize()

    # dominant kernel class: decided from an untimed profiled pass over one utterance
    dom = None
    if not args.no_prof and Wn > 0:
        ncls = lib.ss_prof_num_classes()
        lib.ss_prof_reset()
        lib.ss_prof_enable((1 << ncls) - 1)
        run_utterance(model, voc, pcms[0], mine[0])
        torch.cuda.synchronize()
        best = -1.0
        for c in range(ncls):
            ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
            lib.ss_prof_read(c, C.byref(ms), C.byref(fl), C.byref(n))
            if ms.value > best:
                best, dom = ms.value, c
        lib.ss_prof_enable(0)
        lib.ss_prof_reset()

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if dom is not None:
        lib.ss_prof_enable(1 << dom)
    t0 = time.perf_counter()
    samples_out = 0
    for u, p in zip(mine[Wn:Wn + K], pcms[Wn:Wn + K]):
        wav, _, _, _ = run_utterance(model, voc, p, u)
        samples_out += wav.numel()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    lib.ss_prof_enable(0)

    audio = sum(u.seconds for u in mine[Wn:Wn + K])
    stats = torch.tensor([wall, audio, float(K)], dtype=torch.float64, device=dev)
    if dist is not None:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        wall, audio, nutt = float(mx[0]), float(stats[1]), float(stats[2])
    else:
        nutt = float(K)

    roofline = None
    if dom is not None:
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        lib.ss_prof_read(dom, C.byref(ms), C.byref(fl), C.byref(n))
        if n.value > 0 and ms.value > 0:
            ach = fl.value / (ms.value * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": lib.ss_prof_class_name(dom).decode(), "achieved": round(ach, 3),
                        "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                        "traffic": None, "launches": int(n.value), "avg_launch_us": round(1e3 * ms.value / n.value, 2),
                        "algo_gflop_per_launch": round(fl.value / n.value / 1e9, 4),
                        "share_of_wall": round(ms.value * 1e-3 / wall, 3)}

    if rank == 0:
        out = {
            "metric": "real-time factor (RTFx = audio seconds / wall seconds) + utterances/sec, offline S2ST fr-en",
            "value": round(audio / wall, 2), "unit": "x real-time",
            "utterances_per_sec": round(nutt / wall, 3),
            "n_gpus": world, "steps": K, "warmup": Wn, "ms_per_step": round(1e3 * wall / K, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "offline S2ST fr-en, batch=1 per GPU, synthetic CVSS-C-shaped utterances "
                                   "(LogNormal(ln 4.5 s, 0.45) clipped to [1,15] s, seed 1234), full "
                                   "fbank+encoder+CTC+AR-MT+T2U+NAR-unit+vocoder HIP path, random-init weights "
                                   "of the streamspeech.offline.fr-en architecture",
                       "audio_seconds_per_gpu": round(sum(u.seconds for u in mine[Wn:Wn + K]), 2),
                       "parallelism": f"utterance-dp{world}"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, vsd, cfg, vcfg, all_utts[Wn:Wn + K + 1])
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
