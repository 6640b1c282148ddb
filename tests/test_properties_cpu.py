"""Property tests (hypothesis) of the host logic and of two facts the streaming path relies on:
the generator's receptive field and the resampler filter design."""
import os
import sys

import numpy as np
import torch
from hypothesis import given, settings, strategies as st

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import streamspeech_oracle as O
from streamspeech_amd import dp, frontend, synth
from streamspeech_amd.config import ModelConfig, VocoderConfig
from streamspeech_amd.offline import ordered_batches
from streamspeech_amd.pipeline import ctc_collapse_host, units_from_tokens


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(1, 10000), min_size=1, max_size=200), st.integers(1, 40), st.integers(0, 60000))
def test_ordered_batches_partition_sorted_bounded(lengths, bs, max_tokens):
    batches = ordered_batches(lengths, bs, max_tokens)
    flat = [i for b in batches for i in b]
    assert sorted(flat) == list(range(len(lengths)))                       # every utterance exactly once
    assert all(1 <= len(b) <= bs for b in batches)
    seq = [lengths[i] for i in flat]
    assert seq == sorted(seq, reverse=True)                                # longest first, within and across batches
    if max_tokens > 0:
        for b in batches:
            assert len(b) == 1 or len(b) * lengths[b[0]] <= max_tokens


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(0, 6), max_size=120), st.integers(0, 6), st.integers(0, 6))
def test_ctc_collapse_host_matches_oracle(ids, blank, pad):
    toks, idx = ctc_collapse_host(ids, blank, pad)
    otoks, oidx = O.ctc_collapse(ids, blank, pad)
    assert toks == otoks and idx == oidx
    assert all(ids[i] == t for t, i in zip(toks, idx)) and blank not in toks and pad not in toks
    assert all(a != b or j > i + 1 for (a, i), (b, j) in zip(zip(toks, idx), list(zip(toks, idx))[1:]) if j == i + 1)


@settings(max_examples=100, deadline=None)
@given(st.lists(st.integers(0, 1004), max_size=60))
def test_units_from_tokens(tokens):
    cfg = ModelConfig()
    units = units_from_tokens(tokens, cfg)
    body = tokens[:-1] if tokens and tokens[-1] == cfg.eos else tokens
    assert units == [t - 4 for t in body if t not in (0, cfg.eos)]


@settings(max_examples=100, deadline=None)
@given(st.lists(st.floats(0.5, 20.0), min_size=1, max_size=300), st.integers(1, 8))
def test_balanced_shards_cover_and_balance(durs, world):
    shards = dp.balanced_shards(durs, world)
    assert sorted(i for s in shards for i in s) == list(range(len(durs)))
    loads = [sum(durs[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(durs) + 1e-9                     # greedy LPT bound


@settings(max_examples=30, deadline=None)
@given(st.integers(1, 12), st.integers(1, 12))
def test_resampler_filter_design(up, down):
    import math
    g = math.gcd(up, down)
    up, down = up // g, down // g
    h = frontend.design_filter(up, down)
    assert len(h) == 2 * 10 * max(up, down) + 1 and np.allclose(h, h[::-1])
    assert abs(h.sum() - up) < 1e-9                                         # DC gain `up`: unit gain after decimation


def test_vocoder_receptive_field_bound_is_tight_enough():
    """`VocoderConfig.receptive_field_frames()` (21) is what the tail-only synthesis relies on: a change at input
    frame p must not reach any sample of frames > p + rf (bound holds), and does reach frame p + rf - 3 (bound not
    grossly loose).  Checked on the oracle generator with seeded weights."""
    vcfg = VocoderConfig()
    rf = vcfg.receptive_field_frames()
    sd = O.SD(synth.make_vocoder_state_dict(0, vcfg))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(vcfg.model_in_dim, 70, generator=g)
    y0 = O.hifigan_generator(sd, x, vcfg).reshape(-1)
    p = 35
    x2 = x.clone()
    x2[:, p] += 1.0
    y1 = O.hifigan_generator(sd, x2, vcfg).reshape(-1)
    diff = (y1 - y0).abs().reshape(70, 320).amax(dim=1)                     # per-frame max change
    assert float(diff[p + rf + 1:].max()) == 0.0 and float(diff[: p - rf].max()) == 0.0
    assert float(diff[p + rf - 3]) > 0.0 and float(diff[p - rf + 3]) > 0.0
