"""Generators the agent instantiates, same names / argument meaning / return structure as the
reference's ``agent/ctc_decoder.py``, ``agent/ctc_generator.py`` and ``agent/sequence_generator.py``,
over the HIP stages.  ``engine`` is anything with the HipModel method set (tests substitute a
CPU-oracle adapter to exercise the host control flow without a GPU).
"""
from typing import Dict, List, Optional

import torch


class CTCDecoder:
    """agent/ctc_decoder.py:30-111 -- ASR / ST CTC greedy search (blank = index 0)."""

    def __init__(self, tgt_dict, engine, head: int):
        self.tgt_dict, self.engine, self.head = tgt_dict, engine, head
        self.pad, self.eos, self.unk = tgt_dict.pad(), tgt_dict.eos(), tgt_dict.unk()

    @torch.no_grad()
    def generate(self, encoder_out, prefix=None, aux_task_name=None, want_lprobs=False, **kw):
        enc = encoder_out["encoder_out"][0]
        enc = enc[:, 0] if enc.dim() == 3 else enc
        toks, index, raw, logits = self.engine.ctc_greedy(self.head, enc.contiguous(), want_logits=want_lprobs)
        if prefix is not None:  # agent/ctc_decoder.py:90-92
            pre = [int(t) for t in prefix.view(-1).tolist()]
            merged = pre + raw.tolist()[len(pre):]
            from .pipeline import ctc_collapse_host
            toks, index = ctc_collapse_host(merged, 0, self.pad)
            raw = torch.tensor(merged, dtype=torch.int32)
        lprobs = None
        if logits is not None:
            # model.get_normalized_probs + "never select pad, unk" (agent/ctc_decoder.py:52-60), on the engine (ss_log_softmax)
            lprobs = self.engine.normalized_probs(logits, True, self.pad, self.unk).unsqueeze(0)
        return [[{"tokens": torch.tensor(toks, dtype=torch.long), "org_tokens": raw, "lprobs": lprobs,
                  "index": index, "attn": None, "alignment": None}]]


class CTCSequenceGenerator:
    """agent/ctc_generator.py:26-123 -- NAR unit search over the T2U+unit-decoder stage
    (blank = tgt_dict.blank_index = 1004)."""

    def __init__(self, tgt_dict, engine, use_incremental_states=False, t2u_causal=False, mask_eos=False):
        assert not use_incremental_states
        self.tgt_dict, self.engine = tgt_dict, engine
        self.t2u_causal, self.mask_eos = t2u_causal, mask_eos
        self.incremental_states = None

    def reset_incremental_states(self):
        self.incremental_states = None

    @torch.no_grad()
    def generate(self, mt_features: torch.Tensor, prefix=None, n_tail_pad: int = 0, **kw):
        """mt_features [n,512] = the MT decoder states the reference feeds to synthesizer_encoder
        (agent :652-689; T2U encoder and unit decoder run inside one C-ABI stage)."""
        assert prefix is None, "tgt_units_indices is never set by the reference agent (SURVEY.md App. B)"
        toks, raw, _ = self.engine.t2u_units(mt_features, t2u_causal=self.t2u_causal, mask_eos=self.mask_eos,
                                             n_tail_pad=n_tail_pad)
        return [[{"tokens": torch.tensor(toks, dtype=torch.long), "org_tokens": raw, "attn": None, "alignment": None}]]


class SequenceGenerator:
    """agent/sequence_generator.py:165-582 at beam_size = 1: greedy first-pass text
    decoding with prefix and ``max_new_tokens``.  Keeps the reference's token-buffer semantics
    ([eos, prefix...], forced eos at max_len, eos banned below min_len) but feeds only NEW positions
    to the decoder thanks to the KV cache (per-position results are identical)."""

    def __init__(self, engine, tgt_dict, beam_size=1, max_len_a=0, max_len_b=200, max_len=0, min_len=1,
                 eos=None, use_incremental_states=False, **kw):
        assert beam_size == 1, "the StreamSpeech agent always searches with beam 1"
        self.engine, self.tgt_dict = engine, tgt_dict
        self.max_len_a, self.max_len_b, self.min_len = max_len_a, max_len_b, min_len
        self.max_len = max_len or engine.cfg.max_target_positions
        self.eos = tgt_dict.eos() if eos is None else eos
        self.pad = tgt_dict.pad()
        self.incremental_states = None
        self.use_incremental_states = use_incremental_states

    def reset_incremental_states(self):
        self.incremental_states = None

    @torch.no_grad()
    def generate_decoder(self, encoder_outs, src_tokens, src_lengths, sample=None, prefix_tokens=None,
                         constraints=None, bos_token=None, aux_task_name="", encoder_outs_aug=None,
                         max_new_tokens=-1, **kw):
        enc = encoder_outs[0]["encoder_out"][0]
        enc = enc[:, 0] if enc.dim() == 3 else enc
        src_len = src_tokens.size(1)
        prefix = [] if prefix_tokens is None else [int(t) for t in prefix_tokens.view(-1).tolist()]
        start = len(prefix)
        if max_new_tokens == -1:
            max_len = min(int(self.max_len_a * src_len + self.max_len_b), self.max_len - 1)
        else:
            max_len = start + max_new_tokens
        assert self.min_len <= max_len, "min_len cannot be larger than max_len, please adjust these!"
        if start > max_len:
            # the reference's step loop `for step in range(start, max_len + 1)` (agent/sequence_generator.py:340) is empty
            # then, nothing is finalized and the agent's finalized_mt[0][0] raises IndexError: same error here
            raise IndexError(f"prefix of {start} tokens is longer than max_len = {max_len}: no hypothesis can be finalized")
        eng = self.engine
        if hasattr(eng, "mt_greedy"):
            out, feats = eng.mt_greedy(enc.contiguous(), prefix, max_len, self.min_len)
            return [[{"tokens": torch.tensor(prefix + out, dtype=torch.long), "features": feats, "score": None,
                      "attention": None, "alignment": None, "positional_scores": None}]]
        eng.mt_begin(enc.contiguous())
        feats_all = []
        feats, nxt = eng.mt_append([self.eos] + prefix, 0, ban_eos=(start < self.min_len), force_eos=(start >= max_len))
        feats_all.append(feats)
        out = [nxt]
        step = start + 1
        while nxt != self.eos and step <= max_len:
            feats, nxt = eng.mt_append([out[-1]], step, ban_eos=(step < self.min_len), force_eos=(step >= max_len))
            feats_all.append(feats)
            out.append(nxt)
            step += 1
        tokens = torch.tensor(prefix + out, dtype=torch.long)
        # features for positions [eos, prefix..., generated minus the last]: what the reference
        # recomputes via mt_decoder(prev_output_tokens_mt, features_only=True) (agent :638-642)
        return [[{"tokens": tokens, "features": torch.cat(feats_all, 0), "score": None, "attention": None,
                  "alignment": None, "positional_scores": None}]]
