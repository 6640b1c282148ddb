"""Synthetic CVSS-C-shaped workload for benchmarks: no dataset / checkpoint is
reachable, so utterance shapes are synthesised and weights are seeded random (synth.py).

One benchmark "step" = one offline S2ST utterance, batch 1 (BASELINE.json configs[1]):
  PCM (d seconds, already in HBM) -> fbank+CMVN -> 12-layer chunk-Conformer -> ASR/ST CTC greedy
  -> autoregressive MT greedy decode of N = ceil(3.5 d) subwords (KV cache, forced eos at N)
  -> T2U encoder -> NAR unit decoder over U = 25 (N+1) positions -> CTC collapse
  -> unit HiFi-GAN on K = ceil(37 d) units with durations cycling (1,1,2)  => F ~ 50 d frames,
     S = 320 F samples.
Random weights never emit eos and give an arbitrary unit count, so N is fixed by max_new_tokens
and the collapsed unit sequence is cyclically resized to K (every kernel still runs on real
shapes; only the data-dependent lengths are pinned to what a real utterance of d seconds has).
"""
import math
from dataclasses import dataclass
from typing import List

import numpy as np

from . import synth


@dataclass
class Utterance:
    idx: int
    seconds: float
    n_samples: int       # 16 kHz PCM samples
    n_mt: int            # MT subwords to generate
    n_units: int         # vocoder input units
    durations: List[int]


def make_utterances(n: int, seed: int = 1234) -> List[Utterance]:
    durs = synth.synth_durations(seed, n)
    out = []
    for i, d in enumerate(durs):
        ns = int(round(d * 16000))
        K = int(math.ceil(37 * d))
        pattern = [1, 1, 2]
        out.append(Utterance(i, float(d), ns, int(math.ceil(3.5 * d)), K, [pattern[j % 3] for j in range(K)]))
    return out


def resize_units(units: List[int], K: int, seed_idx: int = 0) -> List[int]:
    """Cyclically extend / truncate the (random-weight) unit sequence to K entries."""
    if not units:
        units = [int(u) for u in synth.uniform(99, f"fallback_units/{seed_idx}", (K,), 0, 1000)]
    reps = (K + len(units) - 1) // len(units)
    return (units * reps)[:K]


def shard(items, rank: int, world: int):
    """Utterance-level data parallel: round-robin deal."""
    return items[rank::world]


def run_batch(model, voc, pcm_packed, utts, detail: bool = False):
    """The benchmarked hot path for one ragged batch of utterances (bench.py's timed step and the parity
    test at the benchmarked configuration call THIS function): PCM in HBM -> fbank+CMVN -> chunk-Conformer
    -> CTC x2 -> lock-step AR MT greedy (n_mt tokens, forced eos) -> T2U + NAR unit decoder + CTC collapse
    -> unit sequence resized to K -> unit HiFi-GAN with the workload's durations.  Each utterance keeps its
    B = 1 arithmetic (streamspeech_amd/csrc/batch.hip ss_batch_*), and since round 5 PACK-INVARIANTLY so: every stage upstream of
    an arg-max sums an utterance's products in an order that is a function of that utterance alone (one accumulator chain per GEMM
    output element whatever the packed row count, whole-tile fused FFN, fixed LayerNorm / attention / decode forms:
    ss_model_set_pack_invariant, default on), so its logits are bit-identical alone, in this pack and in any other
    (tests/test_pack_invariance_gpu.py, tests/test_margin_gpu.py); tests/test_bench_config_gpu.py and tests/test_multilingual_gpu.py
    hold the ids of 320 packed utterances (packs of 128, the measured configuration) identical to the B = 1 oracle (a row the float32 oracle itself cannot decide is
    adjudicated in float64, oracle/adjudicate.py).  The vocoder (float output) keeps its stream-K / Winograd forms.  With detail=True every intermediate the
    parity test compares is returned as well (raw argmax ids, features); the launches are the same."""
    from .pipeline import units_from_tokens
    cfg = model.cfg
    feat, T = model.batch_fbank_cmvn(pcm_packed, [u.n_samples for u in utts])
    enc, Tp = model.batch_encoder_forward(feat, T)
    asr = model.batch_ctc_greedy(0, enc, Tp, return_raw=detail)
    st = model.batch_ctc_greedy(1, enc, Tp, return_raw=detail)
    toks, feats, n = model.batch_mt_greedy(enc, Tp, [u.n_mt for u in utts])
    # the workload pins the data-dependent lengths: every search must have run to its forced </s>
    for b, u in enumerate(utts):
        if len(toks[b]) != u.n_mt + 1 or toks[b][-1] != cfg.eos:
            raise RuntimeError(f"utterance {u.idx}: MT search returned {len(toks[b])} tokens, expected {u.n_mt} + </s>")
    unit_out = model.batch_t2u_units(feats, n, return_raw=detail)
    unit_toks, unit_raw = unit_out if detail else (unit_out, None)
    codes = [resize_units(units_from_tokens(t, cfg), u.n_units, u.idx) for t, u in zip(unit_toks, utts)]
    wavs, dur, _ = voc.batch_forward(codes, dur_prediction=True, forced_dur=[u.durations for u in utts])
    if not detail:
        return wavs, asr, st, toks
    return {"wavs": wavs, "asr": asr, "st": st, "mt": toks, "fbank": feat, "T": T, "Tp": Tp, "unit_toks": unit_toks,
            "unit_raw": unit_raw, "codes": codes, "dur": dur, "n_feats": n}


def bench_plan(steps: int, batch: int, rank: int = 0, world: int = 1, bucket: bool = True, warm: int = 3,
               pool_cap: int = 4096, strong: bool = False):
    """The utterance set and ragged-batch grouping of `bench.py --steps steps --batch batch` on one rank.
    -> (mine, groups): `mine` = this rank's utterances (the `warm` single-utterance warm-ups first), `groups`
    = the timed steps as lists of indices into `mine`, in dispatch order.  Weak scaling (default): every rank gets
    `steps` batches; strong scaling (BASELINE.json configs[3] read literally: ONE set of steps x batch utterances
    split over the ranks): the set is dealt round-robin, so a rank times ~steps / world batches.  Either way the pool
    is length-sorted before the deal so audio seconds balance too.  Like fairseq-generate
    (dataset.ordered_indices() sorts by source length before batch_by_size) batches are formed from length-sorted
    utterances, longest first (LPT over the concurrent streams)."""
    K = steps * batch
    if strong:
        pool_total = min(K, pool_cap * world)
        timed_all = make_utterances(pool_total)               # the SAME set whatever the world size
        warm_all = make_utterances(warm * world, seed=4321)
        timed_all = sorted(timed_all, key=lambda u: -u.seconds)
        share = shard(timed_all, rank, world)
        mine = shard(warm_all, rank, world) + share
        n_mine = len(range(rank, K, world))          # this rank's part of the K timed utterances
        ids = [warm + (i % max(1, len(share))) for i in range(n_mine)] if share else []
    else:
        pool = min(K, pool_cap)
        all_utts = make_utterances((pool + warm) * world)
        warm_all, timed_all = all_utts[:warm * world], all_utts[warm * world:]
        timed_all = sorted(timed_all, key=lambda u: -u.seconds)
        mine = shard(warm_all, rank, world) + shard(timed_all, rank, world)
        ids = [warm + (i % pool) for i in range(K)]
    if batch > 1 and bucket:
        ids = sorted(ids, key=lambda i: -mine[i].n_samples)
    groups = [ids[g0:g0 + batch] for g0 in range(0, len(ids), batch)]
    return mine, groups
