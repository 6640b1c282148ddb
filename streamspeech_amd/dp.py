"""Utterance-level data parallelism over the GPUs of one node: utterances are
independent, every rank holds a full weight replica, and the ONLY communication is one barrier
before the timed region plus one tiny all-reduce of (wall, audio seconds, utterances) after it.
Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests."""
from typing import List, Sequence, Tuple

import torch


def shard(items: Sequence, rank: int, world: int) -> List:
    """Round-robin deal; with a length-sorted list this balances audio seconds per rank."""
    return list(items[rank::world])


def balanced_shards(durations: Sequence[float], world: int) -> List[List[int]]:
    """Greedy longest-processing-time assignment of utterance indices to ranks."""
    order = sorted(range(len(durations)), key=lambda i: -durations[i])
    loads = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: loads[k])
        out[r].append(i)
        loads[r] += durations[i]
    return out


def reduce_stats(dist, wall: float, audio_s: float, n_utts: float, device="cpu") -> Tuple[float, float, float]:
    """(max wall over ranks, total audio seconds, total utterances)."""
    if dist is None or not dist.is_initialized():       # an initialised world of one still runs the collectives (SS_FORCE_DIST=1)
        return wall, audio_s, n_utts
    t = torch.tensor([wall, audio_s, n_utts], dtype=torch.float64, device=device)
    mx = t.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(mx[0]), float(t[1]), float(t[2])


def gather_per_rank(dist, wall: float, audio_s: float, n_utts: float, device="cpu") -> List[dict]:
    """Per-rank (wall, audio seconds, utterances) on every rank, so that load imbalance is visible in the bench line
    (one more 3-double all-gather after the timed region; the scaling loss of this path is imbalance, not communication)."""
    if dist is None or not dist.is_initialized():
        return [{"rank": 0, "wall_s": round(wall, 5), "audio_s": round(audio_s, 2), "utterances": int(n_utts)}]
    world = dist.get_world_size()
    t = torch.tensor([wall, audio_s, n_utts], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [{"rank": r, "wall_s": round(float(o[0]), 5), "audio_s": round(float(o[1]), 2), "utterances": int(o[2])}
            for r, o in enumerate(out)]
