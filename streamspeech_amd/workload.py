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
