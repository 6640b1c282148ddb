"""Host control flow of the S2ST path over the HIP stages (one utterance, B = 1 semantics).

Mirrors the source-finished branch of ``StreamSpeechS2STAgent.policy`` (reference
agent/speech_to_speech.streamspeech.agent.py:433-753) and the beam-1 loop of
``SequenceGenerator.generate_decoder`` (agent/sequence_generator.py:165-582).
"""
from typing import Dict, List, Optional

import torch

from .engine import HipModel, HipVocoder


def mt_greedy(model: HipModel, enc_out: torch.Tensor, prefix: Optional[List[int]] = None,
              max_new_tokens: int = -1, max_len_b: int = 100, begin: bool = True):
    """Greedy (beam 1) first-pass decoding with KV cache.

    Returns (tokens including the final eos if produced, features [len(fed), 512]) where the
    features cover every fed position ([eos] + prefix + generated-but-last), i.e. exactly what
    the reference recomputes with ``mt_decoder(prev_output_tokens_mt, features_only=True)``
    (agent :638-642) once the trailing eos is stripped.
    """
    cfg = model.cfg
    prefix = list(prefix or [])
    start = len(prefix)
    if max_new_tokens == -1:
        # generator_mt: max_len_a=0, max_len_b=100, max_len = model.max_decoder_positions() (agent :162-180)
        max_len = min(max_len_b, cfg.max_target_positions - 1)
    else:
        max_len = start + max_new_tokens
    if hasattr(model, "mt_greedy"):
        out, feats = model.mt_greedy(enc_out, prefix, max_len, 1)
        return prefix + out, feats
    if begin:
        model.mt_begin(enc_out)
    feats_all = []
    fed = [cfg.eos] + prefix
    feats, nxt = model.mt_append(fed, 0, ban_eos=(start < 1), force_eos=(start >= max_len))
    feats_all.append(feats)
    out = [nxt]
    step = start + 1
    while nxt != cfg.eos and step <= max_len:
        feats, nxt = model.mt_append([out[-1]], step, ban_eos=False, force_eos=(step >= max_len))
        feats_all.append(feats)
        out.append(nxt)
        step += 1
    return prefix + out, torch.cat(feats_all, 0)


def units_from_tokens(tokens: List[int], cfg) -> List[int]:
    """Dictionary walk of the agent (:708-717): symbols are '<s> <pad> </s> <unk> 0..999 <blank>';
    the trailing eos is dropped, bos/eos map to '' and vanish, unit = id - 4."""
    toks = list(tokens)
    if toks and toks[-1] == cfg.eos:
        toks = toks[:-1]
    return [t - 4 for t in toks if t not in (0, cfg.eos)]


def offline_s2st(model: HipModel, vocoder: HipVocoder, fbank: torch.Tensor, attn_chunk: int = 999999,
                 conv_chunk: int = 999999, forced_mt_tokens: Optional[List[int]] = None,
                 t2u_causal: bool = False, dur_prediction: bool = True) -> Dict:
    """fbank [T,80] on the device -> dict(asr, st, mt, units, dur, wav)  (BASELINE.json configs[1])."""
    cfg = model.cfg
    enc = model.encoder_forward(fbank, attn_chunk, conv_chunk)
    asr, asr_idx, _, _ = model.ctc_greedy(0, enc)
    st, st_idx, _, _ = model.ctc_greedy(1, enc)
    model.mt_begin(enc)
    if forced_mt_tokens is None:
        toks, feats = mt_greedy(model, enc, begin=False)
        if toks and toks[-1] == cfg.eos:
            toks = toks[:-1]
        feats = feats[: len(toks) + 1]
    else:
        toks = list(forced_mt_tokens)
        feats, _ = model.mt_append([cfg.eos] + toks, 0, False, False, want_next=False)
    unit_toks, _, _ = model.t2u_units(feats, t2u_causal=t2u_causal)
    units = units_from_tokens(unit_toks, cfg)
    out = {"enc": enc, "asr": asr, "st": st, "mt": toks, "units": units, "asr_index": asr_idx, "st_index": st_idx}
    if units:
        wav, dur = vocoder.forward(units, dur_prediction)
        out["wav"], out["dur"] = wav, dur
    return out


def ctc_collapse_host(ids: List[int], blank: int, pad: int):
    """Host twin of the device CTC collapse (agent/ctc_decoder.py:66-88), used only when a caller
    splices a prefix into the raw argmax sequence."""
    toks, index = [], []
    for i, v in enumerate(ids):
        if (i == 0 or v != ids[i - 1]) and v != blank and v != pad:
            toks.append(v)
            index.append(i)
    return toks, index
