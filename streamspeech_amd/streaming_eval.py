"""Sentence-level streaming evaluation loop + latency bookkeeping of SimulEval, for bench.py --mode streaming
(BASELINE.json configs[2]: simultaneous S2ST, 320 ms chunks, batch 1).

Mirrors SimulEval/simuleval/evaluator/evaluator.py:216-235 (send one source segment, pushpop, until the target is
finished) and the speech-output instance's timing model (evaluator/instance.py:315-319, 349-366, 386-415):
  delay_i   = source milliseconds received when prediction i came out                       (non computation-aware)
  elapsed_i = delay_i + wall-clock milliseconds since the first policy() call                (computation-aware)
  intervals = [start_i, duration_i] with start_i = max(previous end, delay_i)                (playback never overlaps)
  RTF       = end of the last interval / source length  (scorers/latency_scorer.py:574-587; *_CA uses elapsed_i)
  StartOffset = delay_0, EndOffset = last end - source length                               (:540-571)
Only what the S2ST hot path needs: no dataloader, no file output."""
import time
from typing import Dict, List

import numpy as np
import torch

from .simuleval_shim import SpeechSegment


def _intervals(marks: List[float], durations: List[float]):
    out, prev_end = [], None
    for m, d in zip(marks, durations):
        start = m if prev_end is None else max(prev_end, m)
        out.append((start, d))
        prev_end = start + d
    return out


def run_utterance(agent, pcm: np.ndarray, segment_ms: int = 320, sr: int = 16000, sync=True) -> Dict:
    """Feed `pcm` to the agent chunk by chunk; time every pushpop (policy() + host glue, device synchronised)."""
    step = sr * segment_ms // 1000
    src_ms_total = 1000.0 * len(pcm) / sr
    pos, src_ms = 0, 0.0
    delays, elapsed, durations, call_ms, actions = [], [], [], [], []
    samples_out = 0
    wall0 = None
    busy = 0.0
    while True:
        chunk = pcm[pos:pos + step]
        pos += step
        finished = pos >= len(pcm)
        src_ms = min(src_ms_total, 1000.0 * pos / sr)
        if wall0 is None:
            wall0 = time.perf_counter()
        t0 = time.perf_counter()
        seg = agent.pushpop(SpeechSegment(content=chunk.tolist(), sample_rate=sr, finished=finished))
        if sync and torch.cuda.is_available():
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        call_ms.append(1e3 * (t1 - t0))
        busy += t1 - t0
        actions.append("R" if seg.is_empty else "W")
        if not seg.is_empty and len(seg.content) > 0:
            durations.append(1000.0 * len(seg.content) / seg.sample_rate)
            delays.append(src_ms)
            # computation-aware: the source clock does not advance while the system computes, so elapsed = source time
            # + all compute time so far (instance.py:318-319 with the evaluator feeding as fast as the agent consumes)
            elapsed.append(src_ms + 1e3 * busy)
            samples_out += len(seg.content)
        if finished:
            break
    iv, iv_ca = _intervals(delays, durations), _intervals(elapsed, durations)
    end = (iv[-1][0] + iv[-1][1]) if iv else src_ms_total
    end_ca = (iv_ca[-1][0] + iv_ca[-1][1]) if iv_ca else src_ms_total + 1e3 * busy
    return {"source_ms": src_ms_total, "compute_ms": 1e3 * busy, "calls": len(call_ms), "call_ms": call_ms,
            "actions": "".join(actions), "writes": len(delays), "samples_out": samples_out,
            "RTF": end / src_ms_total, "RTF_CA": end_ca / src_ms_total,
            "StartOffset": delays[0] if delays else src_ms_total, "StartOffset_CA": elapsed[0] if elapsed else src_ms_total + 1e3 * busy,
            "EndOffset": end - src_ms_total, "EndOffset_CA": end_ca - src_ms_total}


def summarize(runs: List[Dict]) -> Dict:
    tot_src = sum(r["source_ms"] for r in runs)
    tot_cmp = sum(r["compute_ms"] for r in runs)
    calls = [c for r in runs for c in r["call_ms"]]
    mean = lambda k: float(np.mean([r[k] for r in runs]))
    # where the time goes: calls that READ (encoder prefix + two CTC heads + the gate), calls that WRITE before the source
    # ends (+ MT continuation, T2U, unit decoder, vocoder tail) and the source-finished call (MT search to </s> or to the cap)
    reads = [c for r in runs for c, a in zip(r["call_ms"][:-1], r["actions"][:-1]) if a == "R"]
    writes = [c for r in runs for c, a in zip(r["call_ms"][:-1], r["actions"][:-1]) if a == "W"]
    finals = [r["call_ms"][-1] for r in runs]
    by_kind = {"read_calls": len(reads), "ms_per_read_call_mean": round(float(np.mean(reads)), 3) if reads else None,
               "ms_per_read_call_p95": round(float(np.percentile(reads, 95)), 3) if reads else None,
               "write_calls_before_source_end": len(writes), "ms_per_write_call_mean": round(float(np.mean(writes)), 3) if writes else None,
               "source_finished_calls": len(finals), "ms_per_source_finished_call_mean": round(float(np.mean(finals)), 3)}
    return {"calls_by_kind": by_kind, "utterances": len(runs), "audio_s": round(tot_src / 1e3, 2), "compute_s": round(tot_cmp / 1e3, 4),
            "rtfx_compute": round(tot_src / tot_cmp, 2), "ms_per_utterance": round(tot_cmp / len(runs), 3),
            "policy_calls": len(calls), "ms_per_policy_call_mean": round(float(np.mean(calls)), 3),
            "ms_per_policy_call_p95": round(float(np.percentile(calls, 95)), 3), "ms_per_policy_call_max": round(float(np.max(calls)), 3),
            "writes_per_utterance": round(mean("writes"), 2),
            "RTF": round(mean("RTF"), 4), "RTF_CA": round(mean("RTF_CA"), 4),
            "StartOffset_ms": round(mean("StartOffset"), 1), "StartOffset_CA_ms": round(mean("StartOffset_CA"), 1),
            "EndOffset_ms": round(mean("EndOffset"), 1), "EndOffset_CA_ms": round(mean("EndOffset_CA"), 1)}
