"""CPU ORACLE for the StreamSpeech S2ST hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain torch-fp32 CPU restatement of the reference's forward pass for one utterance
(B = 1 semantics), written from the reference's algorithm and cited
function by function (paths relative to /root/reference).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file;
the product path (``streamspeech_amd``) never does and fails loudly without its HIP library.

Pinning status (see DESIGN.md "Oracle"):
  * rel-pos attention / rel_shift / RelPositionalEncoding: pinned against the reference's own
    known-answer tests (fairseq/tests/test_espnet_multihead_attention.py:99-147,
    fairseq/tests/test_positional_encoding.py:42-59) in tests/test_oracle_golden.py.
  * every other stage: pinned against outputs of the reference modules themselves, loaded
    file-by-file from /root/reference (oracle/ref_loader.py) and stored as fixtures under
    tests/golden/ by oracle/make_golden.py.
  * Kaldi fbank (torchaudio.compliance.kaldi.fbank, third-party, absent here): restated from
    the published algorithm in oracle/kaldi_fbank.py -- PARITY UNPINNED.

All tensors are float32 on CPU; time-major 2-D ``[T, C]`` (the reference's ``T x B x C`` with
B = 1 squeezed).  ``sd`` is a state dict with the fairseq key names (numpy or torch values).
"""
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5
BN_EPS = 1e-5


def _t(x) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x
    return torch.from_numpy(np.ascontiguousarray(x))


class SD:
    """Read-only view of a state dict that hands out torch tensors (zero-copy from numpy)."""

    def __init__(self, sd, dtype=torch.float32):
        # dtype = torch.float64: the SAME float32 weights and tables, every operation carried out in double precision -- the
        # adjudicator of arg-max rows whose float32 top-1 / top-2 gap is inside float32 rounding (tests/test_bench_config_gpu.py);
        # float32 (default) is the reference's arithmetic and what every fixture was generated with
        self.sd = sd
        self.dtype = dtype
        self._cache = {}

    def __getitem__(self, k) -> torch.Tensor:
        v = self._cache.get(k)
        if v is None:
            v = _t(self.sd[k]).to(self.dtype)
            self._cache[k] = v
        return v

    def __contains__(self, k):
        return k in self.sd


def layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def linear(x, sd: SD, name: str, bias=True):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"] if bias else None)


# ----------------------------------------------------------------------------------------
# a2  Conv1dSubsampler / ChunkCausalConv1d
# ----------------------------------------------------------------------------------------
def chunk_causal_conv1d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor],
                        stride: int, chunk: int, groups: int = 1) -> torch.Tensor:
    """x [Cin, L] -> [Cout, Lo].  Closed form of reference
    chunk_unity/modules/chunk_causal_conv1d.py:39-68: output o is the
    ordinary 'same'-padded conv (pad = k//2) of an input in which every sample at position
    >= (c+1)*chunk is zeroed, c = (o*stride)//chunk; with chunk >= 999 (or <= 0) it is exactly
    F.conv1d(padding=k//2) (the reference's else-branch, :63-66)."""
    k = w.shape[-1]
    pad = k // 2
    L = x.shape[-1]
    Lo = (L + 2 * pad - k) // stride + 1
    if not (0 < chunk < 999):
        return F.conv1d(x[None], w, b, stride=stride, padding=pad, groups=groups)[0]
    o = torch.arange(Lo)
    j = torch.arange(k)
    idx = o[:, None] * stride + j[None, :] - pad                      # [Lo, k] input position
    limit = ((o * stride) // chunk + 1) * chunk                       # first invisible position
    valid = (idx >= 0) & (idx < L) & (idx < limit[:, None])
    cols = x[:, idx.clamp(0, L - 1)] * valid[None].to(x.dtype)        # [Cin, Lo, k]
    if groups == 1:
        out = torch.einsum("clk,ock->ol", cols, w)
    else:  # depthwise: groups == Cin == Cout, w [C,1,k]
        assert groups == x.shape[0] == w.shape[0] and w.shape[1] == 1
        out = torch.einsum("clk,ck->cl", cols, w[:, 0])
    if b is not None:
        out = out + b[:, None]
    return out


def subsample(sd: SD, fbank: torch.Tensor, conv_chunk: int) -> torch.Tensor:
    """fbank [T,80] -> [T',256].  Reference chunk_unity/modules/convolution.py:81-89:
    two (stride-2 k=5 ChunkCausalConv1d -> GLU over channels)."""
    x = fbank.t().contiguous()                                        # [80, T]
    for i in range(2):
        p = f"encoder.subsample.conv_layers.{i}"
        x = chunk_causal_conv1d(x, sd[p + ".weight"], sd[p + ".bias"], 2, conv_chunk)
        x = F.glu(x, dim=0)
    return x.t().contiguous()


def subsample_out_len(T: int) -> int:
    """convolution.py:75-79: ((L-1)/2+1) floor, twice."""
    for _ in range(2):
        T = int(math.floor((T - 1) / 2 + 1))
    return T


# ----------------------------------------------------------------------------------------
# a3  positional table / chunk mask
# ----------------------------------------------------------------------------------------
def rel_pos_table(T: int, d: int) -> torch.Tensor:
    """[2T-1, d]; row m encodes relative offset T-1-m with interleaved sin/cos.
    Reference fairseq/modules/positional_encoding.py:83-129 (flip(pe_positive) ++ pe_negative[1:],
    then the centre slice taken in forward())."""
    rel = torch.arange(T - 1, -T, -1, dtype=torch.float32)[:, None]   # T-1 ... -(T-1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(2 * T - 1, d)
    pe[:, 0::2] = torch.sin(rel * div)
    pe[:, 1::2] = torch.cos(rel * div)
    return pe


def chunk_mask(T: int, chunk: int) -> Optional[torch.Tensor]:
    """bool [T,T], True = masked.  Reference s2t_conformer.py:195-213: key j is hidden from
    query i iff j >= (i//c+1)*c."""
    c = max(chunk, 1)
    i = torch.arange(T)[:, None]
    j = torch.arange(T)[None, :]
    return j >= (i // c + 1) * c


# ----------------------------------------------------------------------------------------
# a4-a7  conformer layer
# ----------------------------------------------------------------------------------------
def ffn_module(sd: SD, p: str, x):
    """conformer_layer.py:152-164: LN -> w_1 -> SiLU -> w_2."""
    h = layer_norm(x, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"])
    h = F.silu(linear(h, sd, p + ".w_1"))
    return linear(h, sd, p + ".w_2")


def rel_shift_closed(bd_raw: torch.Tensor) -> torch.Tensor:
    """[H,T,2T-1] -> [H,T,T]: out[i,j] = raw[i, j-i+T-1] (espnet_multihead_attention.py:133-152,
    closed form SURVEY.md §7.3 H4)."""
    H, T, _ = bd_raw.shape
    i = torch.arange(T)[:, None]
    j = torch.arange(T)[None, :]
    idx = (j - i + T - 1).expand(H, T, T)
    return torch.gather(bd_raw, 2, idx)


def relpos_mha(sd: SD, p: str, x, pos, mask: Optional[torch.Tensor], heads: int):
    """espnet_multihead_attention.py:154-209 (+ :61-87) for B = 1.  x [T,d], pos [2T-1,d]."""
    T, d = x.shape
    dk = d // heads
    q = linear(x, sd, p + ".linear_q").view(T, heads, dk)
    k = linear(x, sd, p + ".linear_k").view(T, heads, dk).transpose(0, 1)      # [H,T,dk]
    v = linear(x, sd, p + ".linear_v").view(T, heads, dk).transpose(0, 1)
    pp = F.linear(pos, sd[p + ".linear_pos.weight"]).view(-1, heads, dk).transpose(0, 1)  # [H,2T-1,dk]
    qu = (q + sd[p + ".pos_bias_u"]).transpose(0, 1)                           # [H,T,dk]
    qv = (q + sd[p + ".pos_bias_v"]).transpose(0, 1)
    ac = torch.matmul(qu, k.transpose(-2, -1))
    bd = rel_shift_closed(torch.matmul(qv, pp.transpose(-2, -1)))
    scores = (ac + bd) / math.sqrt(dk)
    if mask is not None:
        scores = scores.masked_fill(mask[None], float("-inf"))
    attn = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(attn, v).transpose(0, 1).reshape(T, d)
    return linear(ctx, sd, p + ".linear_out")


def conv_module(sd: SD, p: str, x, conv_chunk: int):
    """conformer_layer.py:94-119: LN -> pw1 -> GLU -> chunk-causal depthwise -> BN(eval) -> SiLU -> pw2."""
    h = layer_norm(x, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"])
    h = F.linear(h, sd[p + ".pointwise_conv1.weight"][:, :, 0])                # [T,2d]
    h = F.glu(h, dim=-1)
    h = chunk_causal_conv1d(h.t().contiguous(), sd[p + ".depthwise_conv.weight"], None, 1,
                            conv_chunk, groups=h.shape[-1]).t()
    bn = p + ".batch_norm"
    h = (h - sd[bn + ".running_mean"]) / torch.sqrt(sd[bn + ".running_var"] + BN_EPS) \
        * sd[bn + ".weight"] + sd[bn + ".bias"]
    h = F.silu(h)
    return F.linear(h, sd[p + ".pointwise_conv2.weight"][:, :, 0])


def conformer_layer(sd: SD, p: str, x, pos, mask, heads: int, conv_chunk: int):
    """conformer_layer.py:254-312."""
    x = ffn_module(sd, p + ".ffn1", x) * 0.5 + x
    h = layer_norm(x, sd[p + ".self_attn_layer_norm.weight"], sd[p + ".self_attn_layer_norm.bias"])
    x = relpos_mha(sd, p + ".self_attn", h, pos, mask, heads) + x
    x = x + conv_module(sd, p + ".conv_module", x, conv_chunk)
    x = ffn_module(sd, p + ".ffn2", x) * 0.5 + x
    return layer_norm(x, sd[p + ".final_layer_norm.weight"], sd[p + ".final_layer_norm.bias"])


def encoder_forward(sd, fbank, cfg, attn_chunk: int = 999999, conv_chunk: int = 999999,
                    n_layers: Optional[int] = None) -> torch.Tensor:
    """fbank [T,80] -> encoder_out [T',256].  Reference chunk_unity/models/s2t_conformer.py:111-163.
    attn_chunk = encoder.chunk_size, conv_chunk = the ChunkCausalConv1d chunk the agent sets
    (agent/speech_to_speech.streamspeech.agent.py:395-413)."""
    sd = sd if isinstance(sd, SD) else SD(sd)
    fbank = _t(fbank).to(sd.dtype)
    x = subsample(sd, fbank, conv_chunk)
    T, d = x.shape
    x = math.sqrt(d) * x
    pos = rel_pos_table(T, d).to(sd.dtype)
    x = linear(x, sd, "encoder.linear")
    mask = chunk_mask(T, attn_chunk) if attn_chunk < T else None
    L = cfg.enc_layers if n_layers is None else n_layers
    for i in range(L):
        x = conformer_layer(sd, f"encoder.conformer_layers.{i}", x, pos, mask, cfg.enc_heads, conv_chunk)
    return x


# ----------------------------------------------------------------------------------------
# a8  CTC heads (ASR / ST)
# ----------------------------------------------------------------------------------------
def ctc_collapse(ids: List[int], blank: int, pad: int) -> Tuple[List[int], List[int]]:
    """agent/ctc_decoder.py:66-88: dedup consecutive, drop blank and pad; also the kept frame index."""
    toks, index = [], []
    for i, v in enumerate(ids):
        if i == 0 or v != ids[i - 1]:
            if v != blank and v != pad:
                toks.append(v)
                index.append(i)
    return toks, index


def ctc_head(sd, enc_out, name: str, cfg):
    """agent/ctc_decoder.py:39-111 + fairseq ctc_decoder.py:11-18: Linear -> log_softmax ->
    pad/unk = -inf -> argmax -> collapse with blank = 0.  Returns (tokens, index, raw argmax, logits)."""
    sd = sd if isinstance(sd, SD) else SD(sd)
    logits = linear(_t(enc_out).to(sd.dtype), sd, f"{name}_decoder.proj")
    lp = F.log_softmax(logits, dim=-1)
    lp[:, cfg.pad] = -math.inf
    lp[:, cfg.unk] = -math.inf
    raw = lp.argmax(dim=-1).tolist()
    toks, index = ctc_collapse(raw, 0, cfg.pad)
    return toks, index, raw, logits


# ----------------------------------------------------------------------------------------
# fairseq MHA / transformer layers (a9-a12)
# ----------------------------------------------------------------------------------------
def sinusoid_table(n: int, dim: int, padding_idx: int) -> torch.Tensor:
    """fairseq/modules/sinusoidal_positional_embedding.py:43-64 (sin || cos, pad row zero)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(n, dtype=torch.float)[:, None] * e[None, :]
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1)
    e[padding_idx] = 0
    return e


def fairseq_mha(sd: SD, p: str, q_in, kv_in, heads: int, causal: bool = False, k_mask_tail: int = 0):
    """ctc_unity/modules/multihead_attention.py:544-573,673-760 for B = 1, no padding:
    q = (W_q x + b_q) * d_h^-0.5, additive triu(-inf,1) mask when causal, fp32 softmax, bmm, out_proj."""
    Tq, D = q_in.shape
    Tk = kv_in.shape[0]
    dh = D // heads
    q = (linear(q_in, sd, p + ".q_proj") * dh ** -0.5).view(Tq, heads, dh).transpose(0, 1)
    k = linear(kv_in, sd, p + ".k_proj").view(Tk, heads, dh).transpose(0, 1)
    v = linear(kv_in, sd, p + ".v_proj").view(Tk, heads, dh).transpose(0, 1)
    w = torch.bmm(q, k.transpose(1, 2))
    if causal:
        w = w + torch.triu(torch.full((Tq, Tk), float("-inf")), 1)[None]
    if k_mask_tail > 0:   # key_padding_mask on trailing <pad> keys (multihead_attention.py:700-718)
        w[:, :, Tk - k_mask_tail:] = float("-inf")
    w = torch.softmax(w, dim=-1)
    ctx = torch.bmm(w, v).transpose(0, 1).reshape(Tq, D)
    return linear(ctx, sd, p + ".out_proj")


def decoder_layer(sd: SD, p: str, x, enc, heads: int, self_tail: int = 0, cross_tail: int = 0):
    """ctc_unity/modules/transformer_layer.py:388-551 (pre-LN, ReLU, causal self-attn + cross-attn)."""
    h = layer_norm(x, sd[p + ".self_attn_layer_norm.weight"], sd[p + ".self_attn_layer_norm.bias"])
    x = x + fairseq_mha(sd, p + ".self_attn", h, h, heads, causal=True, k_mask_tail=self_tail)
    h = layer_norm(x, sd[p + ".encoder_attn_layer_norm.weight"], sd[p + ".encoder_attn_layer_norm.bias"])
    x = x + fairseq_mha(sd, p + ".encoder_attn", h, enc, heads, k_mask_tail=cross_tail)
    h = layer_norm(x, sd[p + ".final_layer_norm.weight"], sd[p + ".final_layer_norm.bias"])
    h = linear(F.relu(linear(h, sd, p + ".fc1")), sd, p + ".fc2")
    return x + h


def encoder_layer(sd: SD, p: str, x, heads: int, causal: bool, tail: int = 0):
    """ctc_unity/modules/transformer_layer.py:165-230 (pre-LN encoder layer, ReLU)."""
    h = layer_norm(x, sd[p + ".self_attn_layer_norm.weight"], sd[p + ".self_attn_layer_norm.bias"])
    x = x + fairseq_mha(sd, p + ".self_attn", h, h, heads, causal=causal, k_mask_tail=tail)
    h = layer_norm(x, sd[p + ".final_layer_norm.weight"], sd[p + ".final_layer_norm.bias"])
    h = linear(F.relu(linear(h, sd, p + ".fc1")), sd, p + ".fc2")
    return x + h


def mt_decoder_features(sd, tokens: List[int], enc_out, cfg) -> torch.Tensor:
    """target_unigram_decoder(prev_output_tokens, encoder_out, features_only=True)[0] for B = 1:
    ctc_unity/modules/transformer_decoder.py:257-403.  tokens = [eos, t1, ...] -> [n, 512]
    (post final LayerNorm).  Embedding = sqrt(512)*E[tok] + sinusoid(pos), positions from
    padding_idx+1 = 2 (fairseq/utils.py:256-266)."""
    sd = sd if isinstance(sd, SD) else SD(sd)
    enc_out = _t(enc_out).to(sd.dtype)
    p = "target_unigram_decoder"
    D = cfg.dec_dim
    tok = torch.tensor(tokens, dtype=torch.long)
    n = tok.numel()
    mask = tok.ne(cfg.pad).int()
    positions = torch.cumsum(mask, 0) * mask + cfg.pad
    table = sinusoid_table(cfg.pad + 1 + max(n, 1024), D, cfg.pad)
    x = math.sqrt(D) * F.embedding(tok, sd[p + ".embed_tokens.weight"]) + table[positions]
    n_pad = int((tok == cfg.pad).sum())      # trailing <pad> (whole-word mode, agent :576-584)
    assert n_pad == 0 or bool((tok[-n_pad:] == cfg.pad).all())
    for i in range(cfg.mt_layers):
        x = decoder_layer(sd, f"{p}.layers.{i}", x, enc_out, cfg.dec_heads, self_tail=n_pad)
    return layer_norm(x, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"])


def mt_greedy(sd, enc_out, cfg, prefix: Optional[List[int]] = None, max_new_tokens: int = -1,
              max_len_b: int = 100) -> List[int]:
    """agent/sequence_generator.py:165-582 at beam 1: returns generated
    tokens INCLUDING the final eos (caller strips it, agent :535-538).  No KV cache: every step
    re-runs the decoder on the whole prefix, exactly like the reference (use_incremental_states=False)."""
    sd = sd if isinstance(sd, SD) else SD(sd)
    prefix = list(prefix or [])
    start = len(prefix)
    if max_new_tokens == -1:
        # generator_mt: max_len_a=0, max_len_b=100, max_len = model.max_decoder_positions() = 1200
        # (agent :162-180, sequence_generator.py:204-214)
        max_len = min(max_len_b, cfg.max_target_positions - 1)
    else:
        max_len = start + max_new_tokens
    tokens = [cfg.eos] + prefix
    E = sd["target_unigram_decoder.output_projection.weight"]
    for step in range(start, max_len + 1):
        feats = mt_decoder_features(sd, tokens, enc_out, cfg)
        lp = F.log_softmax(F.linear(feats[-1], E), dim=-1)
        lp[lp != lp] = -math.inf
        lp[cfg.pad] = -math.inf
        if step >= max_len:
            lp[: cfg.eos] = -math.inf
            lp[cfg.eos + 1:] = -math.inf
        elif step < 1:  # min_len = 1
            lp[cfg.eos] = -math.inf
        nxt = int(lp.argmax())
        tokens.append(nxt)
        if nxt == cfg.eos:
            break
    return tokens[1:]


def t2u_encoder(sd, x, cfg, causal: bool = False, n_tail_pad: int = 0) -> torch.Tensor:
    """synthesizer_encoder: ctc_unity/modules/transformer_encoder.py:32-77 (2 pre-LN layers + LN);
    causal iff --uni-encoder (simultaneous checkpoints)."""
    sd = sd if isinstance(sd, SD) else SD(sd)
    x = _t(x).to(sd.dtype)
    for i in range(cfg.t2u_layers):
        x = encoder_layer(sd, f"synthesizer_encoder.layers.{i}", x, cfg.dec_heads, causal, n_tail_pad)
    return layer_norm(x, sd["synthesizer_encoder.layer_norm.weight"], sd["synthesizer_encoder.layer_norm.bias"])


def unit_decoder_logits(sd, t2u_out, cfg, n_tail_pad: int = 0) -> torch.Tensor:
    """CTCTransformerUnitDecoder.forward for B = 1: ctc_transformer_unit_decoder.py:153-260.
    Each T2U state is repeated ctc_upsample times; the positional term is the reference's quirk
    (SURVEY.md H2): embed_positions(x[:, :, 0]) treats [U, B] floats as [bsz, seqlen], so for
    B = 1 every position gets sinusoid row padding_idx+1 (=2) -- or the zero pad row where the
    feature value equals padding_idx (1.0) exactly."""
    sd = sd if isinstance(sd, SD) else SD(sd)
    t2u_out = _t(t2u_out).to(sd.dtype)
    n, D = t2u_out.shape
    x = t2u_out[:, None, :].repeat(1, cfg.ctc_upsample, 1).reshape(n * cfg.ctc_upsample, D)
    table = sinusoid_table(cfg.pad + 1 + 1024, D, cfg.pad)
    first = x[:, 0]
    posidx = torch.where(first.ne(float(cfg.pad)), torch.tensor(cfg.pad + 1), torch.tensor(cfg.pad))
    x = x + table[posidx]
    for i in range(cfg.unit_layers):
        x = decoder_layer(sd, f"decoder.layers.{i}", x, t2u_out, cfg.dec_heads,
                          self_tail=n_tail_pad * cfg.ctc_upsample, cross_tail=n_tail_pad)
    x = layer_norm(x, sd["decoder.layer_norm.weight"], sd["decoder.layer_norm.bias"])
    return F.linear(x, sd["decoder.output_projection.weight"])


def unit_ctc_generate(logits, cfg, mask_eos: bool = False) -> Tuple[List[int], List[int]]:
    """agent/ctc_generator.py:40-123 (offline twin researches/ctc_unity/ctc_generator.py:58 also
    masks eos): log_softmax -> pad/unk(-/eos) = -inf -> argmax -> dedup -> drop blank(1004)/pad.
    Then the agent's dictionary walk (agent :708-717): symbols are '<s> <pad> </s> <unk> 0..999
    <blank>', so unit = id - 4 and bos/eos vanish.  Returns (units, raw argmax)."""
    lp = F.log_softmax(_t(logits), dim=-1)
    lp[:, cfg.pad] = -math.inf
    lp[:, cfg.unk] = -math.inf
    if mask_eos:
        lp[:, cfg.eos] = -math.inf
    raw = lp.argmax(dim=-1).tolist()
    toks, _ = ctc_collapse(raw, cfg.unit_blank, cfg.pad)
    if toks and toks[-1] == cfg.eos:
        toks = toks[:-1]
    units = [t - 4 for t in toks if t not in (0, cfg.eos)]
    return units, raw


# ----------------------------------------------------------------------------------------
# a14-a15  unit HiFi-GAN vocoder with duration prediction
# ----------------------------------------------------------------------------------------
def fold_weight_norm(sd: SD, name: str) -> torch.Tensor:
    """w = g * v / ||v||, norm over all dims but 0 (torch.nn.utils.weight_norm dim=0;
    reference hifigan.py:172-179 remove_weight_norm)."""
    if name + ".weight" in sd:
        return sd[name + ".weight"]
    v, g = sd[name + ".weight_v"], sd[name + ".weight_g"]
    nrm = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return g * v / nrm


def duration_predict(sd: SD, emb: torch.Tensor) -> torch.Tensor:
    """emb [K,128] -> int64 dur [K].  fastspeech2.py:117-151 + codehifigan.py:59-66:
    conv k3 -> ReLU -> LN -> conv k3 -> ReLU -> LN -> Linear -> clamp(round(exp(.) - 1), min=1)."""
    p = "dur_predictor"
    x = emb.t()[None]
    x = F.relu(F.conv1d(x, sd[p + ".conv1.0.weight"], sd[p + ".conv1.0.bias"], padding=1))[0].t()
    x = layer_norm(x, sd[p + ".ln1.weight"], sd[p + ".ln1.bias"])
    x = F.relu(F.conv1d(x.t()[None], sd[p + ".conv2.0.weight"], sd[p + ".conv2.0.bias"], padding=1))[0].t()
    x = layer_norm(x, sd[p + ".ln2.weight"], sd[p + ".ln2.bias"])
    log_dur = F.linear(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"])[:, 0]
    return torch.clamp(torch.round(torch.exp(log_dur) - 1).long(), min=1)


def hifigan_generator(sd: SD, x: torch.Tensor, vcfg) -> torch.Tensor:
    """x [128, F] -> wav [320 F].  hifigan.py:154-170 and ResBlock.forward :95-102."""
    x = F.conv1d(x[None], fold_weight_norm(sd, "conv_pre"), sd["conv_pre.bias"], padding=3)
    nk = len(vcfg.resblock_kernel_sizes)
    for i, (u, ku) in enumerate(zip(vcfg.upsample_rates, vcfg.upsample_kernel_sizes)):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, fold_weight_norm(sd, f"ups.{i}"), sd[f"ups.{i}.bias"],
                               stride=u, padding=(ku - u) // 2)
        xs = None
        for j, (kr, dils) in enumerate(zip(vcfg.resblock_kernel_sizes, vcfg.resblock_dilation_sizes)):
            r = x
            for di, dil in enumerate(dils):
                p = f"resblocks.{i * nk + j}"
                xt = F.leaky_relu(r, 0.1)
                xt = F.conv1d(xt, fold_weight_norm(sd, f"{p}.convs1.{di}"), sd[f"{p}.convs1.{di}.bias"],
                              dilation=dil, padding=(kr * dil - dil) // 2)
                xt = F.leaky_relu(xt, 0.1)
                xt = F.conv1d(xt, fold_weight_norm(sd, f"{p}.convs2.{di}"), sd[f"{p}.convs2.{di}.bias"],
                              padding=(kr - 1) // 2)
                r = xt + r
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)  # default slope 0.01 (hifigan.py:166)
    x = F.conv1d(x, fold_weight_norm(sd, "conv_post"), sd["conv_post.bias"], padding=3)
    return torch.tanh(x)[0, 0]


def vocoder_forward(vsd, units: List[int], vcfg, dur_prediction: bool = True,
                    forced_dur: Optional[List[int]] = None):
    """CodeHiFiGANVocoderWithDur.forward (agent/tts/vocoder.py:48-60) + CodeGenerator.forward
    (agent/tts/codehifigan.py:56-95) for one utterance -> (wav [320*sum(dur)], dur [K])."""
    vsd = vsd if isinstance(vsd, SD) else SD(vsd)
    code = torch.tensor([u for u in units if u >= 0], dtype=torch.long)
    emb = F.embedding(code, vsd["dict.weight"])                       # [K,128]
    if forced_dur is not None:
        dur = torch.tensor(forced_dur, dtype=torch.long)
    elif dur_prediction:
        dur = duration_predict(vsd, emb)
    else:
        dur = torch.ones(code.numel(), dtype=torch.long)
    x = torch.repeat_interleave(emb, dur, dim=0).t().contiguous()     # [128,F]
    return hifigan_generator(vsd, x, vcfg), dur


# ----------------------------------------------------------------------------------------
# whole offline utterance (BASELINE.json configs[1]) -- what bench.py's cpu_baseline times
# ----------------------------------------------------------------------------------------
def offline_s2st(sd, vsd, fbank, cfg, vcfg, attn_chunk=999999, conv_chunk=999999,
                 forced_mt_tokens: Optional[List[int]] = None, t2u_causal: bool = False):
    """fbank [T,80] -> dict(asr, st, mt, units, dur, wav).  Mirrors the source-finished branch of
    StreamSpeechS2STAgent.policy (agent :433-753).  ``forced_mt_tokens`` teacher-forces the MT
    hypothesis (random weights never emit eos)."""
    sd = sd if isinstance(sd, SD) else SD(sd)
    vsd = vsd if isinstance(vsd, SD) else SD(vsd)
    enc = encoder_forward(sd, fbank, cfg, attn_chunk, conv_chunk)
    asr, asr_idx, _, _ = ctc_head(sd, enc, "source_unigram", cfg)
    st, st_idx, _, _ = ctc_head(sd, enc, "ctc_target_unigram", cfg)
    if forced_mt_tokens is None:
        mt = mt_greedy(sd, enc, cfg)
        if mt and mt[-1] == cfg.eos:
            mt = mt[:-1]
    else:
        mt = list(forced_mt_tokens)
    feats = mt_decoder_features(sd, [cfg.eos] + mt, enc, cfg)
    t2u = t2u_encoder(sd, feats, cfg, causal=t2u_causal)
    logits = unit_decoder_logits(sd, t2u, cfg)
    units, _ = unit_ctc_generate(logits, cfg)
    out = {"enc": enc, "asr": asr, "st": st, "mt": mt, "units": units, "unit_logits": logits}
    if len(units) > 0:
        wav, dur = vocoder_forward(vsd, units, vcfg, True)
        out["wav"], out["dur"] = wav, dur
    return out
