"""Python host side of the HIP library: owns device memory (torch tensors), hands raw pointers
to the C ABI.  Torch is plumbing here (allocation, streams, H2D/D2H copies) -- every FLOP of the
S2ST path runs in libstreamspeech_hip.so.
"""
import ctypes as C
import os
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import lib as L
from .config import ModelConfig, VocoderConfig
from .weights import pack_model, pack_vocoder


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise L.StreamSpeechHipError(
            "streamspeech_amd needs an AMD GPU (torch.cuda.is_available() is False); "
            "there is no CPU fallback for the product path")
    return torch.device(device)


def _slots(names, offsets, numels):  # noqa: E302
    n = len(names)
    c_names = (C.c_char_p * n)(*[s.encode() for s in names])
    c_off = (C.c_int64 * n)(*offsets)
    c_num = (C.c_int64 * n)(*numels)
    return c_names, c_off, c_num, n


def _i32(vals):
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])


class BatchMixin:
    """Ragged-batch calls on a HipModel (B utterances packed along the row axis, no padding)."""

    def batch_fbank_cmvn(self, pcm_packed: torch.Tensor, n_samples: List[int], pcm_scale: float = 32768.0):
        B = len(n_samples)
        starts = np.concatenate([[0], np.cumsum(n_samples)[:-1]]).astype(np.int64)
        T = [self.lib.ss_fbank_num_frames(int(n)) for n in n_samples]
        feat = torch.empty((sum(T), 80), dtype=torch.float32, device=self.device)
        hT = (C.c_int32 * B)()
        L.check(self.lib.ss_batch_fbank_cmvn(self.h, _stream(), B, _ptr(pcm_packed), (C.c_int64 * B)(*starts.tolist()),
                                             _i32(n_samples), pcm_scale, _ptr(feat), hT), "ss_batch_fbank_cmvn")
        return feat, list(hT)

    def batch_encoder_forward(self, fbank_packed: torch.Tensor, T: List[int], attn_chunk=999999, conv_chunk=999999):
        B = len(T)
        Tp = [self.lib.ss_encoder_out_len(int(t)) for t in T]
        out = torch.empty((sum(Tp), self.cfg.enc_dim), dtype=torch.float32, device=self.device)
        hTp = (C.c_int32 * B)()
        L.check(self.lib.ss_batch_encoder_forward(self.h, _stream(), B, _ptr(fbank_packed), _i32(T),
                                                  int(min(attn_chunk, 1 << 30)), int(min(conv_chunk, 1 << 30)),
                                                  _ptr(out), hTp), "ss_batch_encoder_forward")
        return out, list(hTp)

    def batch_ctc_greedy(self, head: int, enc_packed: torch.Tensor, Tp: List[int], return_raw: bool = False):
        """-> per-utterance (tokens, frame index) lists (+ the raw per-frame argmax with return_raw)."""
        B, tot = len(Tp), sum(Tp)
        ibuf = torch.empty((3 * tot + B,), dtype=torch.int32, device=self.device)
        raw, toks, idx, cnt = ibuf[:tot], ibuf[tot:2 * tot], ibuf[2 * tot:3 * tot], ibuf[3 * tot:]
        L.check(self.lib.ss_batch_ctc_greedy(self.h, _stream(), head, B, _ptr(enc_packed), _i32(Tp), _ptr(raw),
                                             _ptr(toks), _ptr(idx), _ptr(cnt)), "ss_batch_ctc_greedy")
        host = ibuf.cpu().numpy()
        out, off = [], 0
        for b in range(B):
            n = int(host[3 * tot + b])
            rec = (host[tot + off: tot + off + n].tolist(), host[2 * tot + off: 2 * tot + off + n].tolist())
            out.append(rec + (host[off: off + Tp[b]].tolist(),) if return_raw else rec)
            off += Tp[b]
        return out

    def batch_mt_greedy(self, enc_packed: torch.Tensor, Tp: List[int], max_len: List[int], min_len: int = 1):
        """-> (list of token lists incl. final eos, feats [B, Lcap, D], n_feats list)."""
        B = len(Tp)
        Lmax = max(max_len)
        rows, stride = Lmax + 2, Lmax + 1
        feats = torch.empty((B, rows, self.cfg.dec_dim), dtype=torch.float32, device=self.device)
        out = (C.c_int32 * (B * stride))()
        n_out = (C.c_int32 * B)()
        L.check(self.lib.ss_batch_mt_greedy(self.h, _stream(), B, _ptr(enc_packed), _i32(Tp), _i32(max_len), min_len, out,
                                            stride, n_out, _ptr(feats), rows), "ss_batch_mt_greedy")
        toks = [list(out[b * stride: b * stride + n_out[b]]) for b in range(B)]
        return toks, feats, list(n_out)

    def last_logits(self) -> torch.Tensor:
        """Dense logits [rows, cols] of this context's last batch_ctc_greedy / batch_t2u_units call (test hook: arg-max margins)."""
        rows, cols = C.c_int(0), C.c_int(0)
        L.check(self.lib.ss_debug_last_logits(self.h, _stream(), None, 0, C.byref(rows), C.byref(cols)), "ss_debug_last_logits")
        out = torch.empty((rows.value, cols.value), dtype=torch.float32, device=self.device)
        L.check(self.lib.ss_debug_last_logits(self.h, _stream(), _ptr(out), out.numel(), C.byref(rows), C.byref(cols)), "ss_debug_last_logits")
        return out

    def batch_t2u_units(self, feats: torch.Tensor, n_rows: List[int], t2u_causal=False, mask_eos=False,
                        return_raw: bool = False):
        """feats [B, rows, D] (rows of utterance b used: n_rows[b]) -> list of collapsed unit-vocab token lists
        (with return_raw: (collapsed lists, raw per-position argmax lists))."""
        B, rows = feats.shape[0], feats.shape[1]
        up = self.cfg.ctc_upsample
        U = sum(n_rows) * up
        ibuf = torch.empty((2 * U + B,), dtype=torch.int32, device=self.device)
        raw, toks, cnt = ibuf[:U], ibuf[U:2 * U], ibuf[2 * U:]
        L.check(self.lib.ss_batch_t2u_units(self.h, _stream(), B, _ptr(feats), rows, _i32(n_rows), int(t2u_causal),
                                            int(mask_eos), _ptr(raw), _ptr(toks), _ptr(cnt)), "ss_batch_t2u_units")
        host = ibuf.cpu().numpy()
        out, raws, off = [], [], 0
        for b in range(B):
            k = int(host[2 * U + b])
            out.append(host[U + off: U + off + k].tolist())
            raws.append(host[off: off + n_rows[b] * up].tolist())
            off += n_rows[b] * up
        return (out, raws) if return_raw else out


class Scratch:
    """One scratch set (ss_scratch: activations, KV caches, stream-K hand-off state, streaming-encoder state) -- everything a call
    mutates.  One per concurrent stream; weight handles of any language (HipModel / HipVocoder) are bound to it with
    `handle.bind_scratch(scratch)` or at construction (`scratch=`).  Driven by one host thread at a time."""

    def __init__(self, device="cuda:0", cap_bytes: int = 0):
        self.lib = L.load()
        self.device = _require_gpu(device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.ss_scratch_create(C.byref(h)), "ss_scratch_create")
        self.h = h
        if cap_bytes:
            self.set_cap(cap_bytes)

    def set_cap(self, max_bytes: int):
        L.check(self.lib.ss_scratch_set_cap(self.h, int(max_bytes)), "ss_scratch_set_cap")

    def trim(self, keep_bytes: int = 0):
        """Synchronises the device and releases the re-sizable buffers, largest first, until at most keep_bytes are held."""
        with torch.cuda.device(self.device):
            L.check(self.lib.ss_scratch_trim(self.h, int(keep_bytes)), "ss_scratch_trim")

    def bytes(self) -> int:
        return int(self.lib.ss_scratch_bytes(self.h))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ss_scratch_destroy(self.h)
                self.h = None
        except Exception:
            pass


class HipModel(BatchMixin):
    """ss_model handle + packed weights (StreamSpeechModel replacement)."""

    def __init__(self, state_dict, cfg: ModelConfig = None, device="cuda:0", cmvn_mean=None, cmvn_std=None,
                 max_rel_pos: int = 2048, max_tgt_pos: int = 1026, _share=None, scratch: "Scratch" = None):
        self.lib = L.load()
        self.cfg = cfg or ModelConfig()
        self.device = _require_gpu(device)
        if _share is not None:      # another execution context over the same (read-only) weight blob
            names, offsets, numels, self.blob = _share
        else:
            names, offsets, numels, blob = pack_model(state_dict, self.cfg, cmvn_mean, cmvn_std, max_rel_pos, max_tgt_pos)
            self.blob = blob.to(self.device)
        self._packed = (names, offsets, numels, self.blob)
        self._dims = (max_rel_pos, max_tgt_pos)
        c = self.cfg
        self.c_cfg = L.SSConfig(
            c.input_feat, c.conv_channels, c.conv_kernel, c.enc_dim, c.enc_ffn, c.enc_heads, c.enc_layers,
            c.dw_kernel, c.src_vocab, c.tgt_vocab, c.mt_layers, c.dec_dim, c.dec_ffn, c.dec_heads, c.t2u_layers,
            c.unit_layers, c.unit_vocab, c.ctc_upsample, c.pad, c.eos, c.unk, max_rel_pos, max_tgt_pos)
        cn, co, cm, n = _slots(names, offsets, numels)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.ss_model_create(C.byref(self.c_cfg), _ptr(self.blob), self.blob.numel(), cn, co, cm, n,
                                             C.byref(h)), "ss_model_create")
        self.h = h
        self.scratch = None                 # None: the handle's own scratch set (made by ss_model_create)
        if scratch is not None:
            self.bind_scratch(scratch)
        self.max_tgt_pos = max_tgt_pos
        # MT decode step of single-utterance searches (ss_mt_greedy / ss_mt_append with one token): the C ABI's default is the
        # launch-per-op form (SS_MT_PERSISTENT overrides it); a PRIMARY context -- the one an agent / the offline driver / a
        # one-utterance-at-a-time caller decodes on, alone on its device -- uses the persistent one-launch step (mt_step.hip:
        # 125 vs 189 us per token).  Contexts made by new_context() exist for concurrency and keep the launch-per-op form; a
        # time-out of the persistent step (its workgroups were not all resident) falls back to it for good and says so.
        self.persistent_mt = int(self.lib.ss_mt_get_persistent(self.h))
        if _share is None and scratch is None and "SS_MT_PERSISTENT" not in os.environ:     # (a handle made ON a shared scratch set is a concurrent one)
            self.set_persistent_mt_step(64)

    def new_context(self, scratch: "Scratch" = None) -> "HipModel":
        """Another ss_model handle borrowing the same weights -- on its own scratch set, or on `scratch` (a set shared with the
        handles of other languages on the same stream): one per concurrent utterance stream.  (Concurrent contexts start with the
        launch-per-op MT decode step: the persistent step's workgroups must all be resident, which only a context that decodes
        alone on the device can count on.)"""
        return HipModel(None, self.cfg, device=str(self.device), max_rel_pos=self._dims[0],
                        max_tgt_pos=self._dims[1], _share=self._packed, scratch=scratch)

    def bind_scratch(self, scratch: "Scratch"):
        """Run this handle on `scratch` from now on (between stateful sequences only: mt_begin ... mt_append and the streaming
        encoder keep their state in the scratch set)."""
        with torch.cuda.device(self.device):
            L.check(self.lib.ss_model_bind_scratch(self.h, scratch.h), "ss_model_bind_scratch")
        self.scratch = scratch
        if hasattr(self, "persistent_mt"):      # the MT decode-step form is a setting of the scratch set (its granule region lives there)
            self.persistent_mt = int(self.lib.ss_mt_get_persistent(self.h))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ss_model_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- waveform front-end (§8f-3) --------------------------------------------------------
    def resample(self, pcm: torch.Tensor, sr_in: int, sr_out: int = 16000) -> torch.Tensor:
        """float32 [n] on the device at sr_in -> [ceil(n*sr_out/sr_in)] at sr_out (polyphase FIR kernel)."""
        import math
        from .frontend import design_filter
        g = math.gcd(int(sr_in), int(sr_out))
        up, down = int(sr_out) // g, int(sr_in) // g
        if up == down:
            return pcm
        key = (up, down)
        cache = self.__dict__.setdefault("_resample_taps", {})
        if key not in cache:
            cache[key] = torch.from_numpy(design_filter(up, down).astype(np.float32)).to(self.device)
        taps = cache[key]
        n_in = pcm.numel()
        n_out = -(-n_in * up // down)
        out = torch.empty((n_out,), dtype=torch.float32, device=self.device)
        L.check(self.lib.ss_resample(_stream(), _ptr(pcm), n_in, up, down, _ptr(taps), (taps.numel() - 1) // 2,
                                     _ptr(out), n_out), "ss_resample")
        return out

    # ---- a1 -------------------------------------------------------------------------------
    def fbank_cmvn(self, pcm16k: torch.Tensor, pcm_scale: float = 32768.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """pcm16k: float32 [n] on the device -> [T, 80] (into ``out`` -- T contiguous rows -- when given)."""
        n = pcm16k.numel()
        T = self.lib.ss_fbank_num_frames(n)
        if out is not None:
            assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (T, 80)
        feat = out if out is not None else torch.empty((T, 80), dtype=torch.float32, device=self.device)
        nf = C.c_int(0)
        L.check(self.lib.ss_fbank_cmvn(self.h, _stream(), _ptr(pcm16k), n, pcm_scale, _ptr(feat), C.byref(nf)),
                "ss_fbank_cmvn")
        return feat

    # ---- a2-a7 ----------------------------------------------------------------------------
    def encoder_out_len(self, T: int) -> int:
        return self.lib.ss_encoder_out_len(T)

    def encoder_forward(self, fbank: torch.Tensor, attn_chunk: int = 999999, conv_chunk: int = 999999) -> torch.Tensor:
        """fbank [T,80] (device, contiguous) -> [T',256]."""
        assert fbank.is_cuda and fbank.dtype == torch.float32 and fbank.is_contiguous()
        T = fbank.shape[0]
        Tp = self.lib.ss_encoder_out_len(T)
        out = torch.empty((Tp, self.cfg.enc_dim), dtype=torch.float32, device=self.device)
        L.check(self.lib.ss_encoder_forward(self.h, _stream(), _ptr(fbank), T, int(min(attn_chunk, 1 << 30)),
                                            int(min(conv_chunk, 1 << 30)), _ptr(out)), "ss_encoder_forward")
        self._last_enc, self._ctc_stash, self._ctc_both = (out.data_ptr(), Tp), None, None
        return out

    def encoder_stream_reset(self):
        """Forget the incremental-encoder cache (new utterance)."""
        L.check(self.lib.ss_encoder_stream_reset(self.h), "ss_encoder_stream_reset")
        self.stream_stats = (0, 0)

    def encoder_stream_set_tail(self, unsettled_fbank_frames: int):
        """Newest fbank frames that may still change on the next call (1 when the front-end resamples, else 0)."""
        L.check(self.lib.ss_encoder_stream_set_tail(self.h, int(unsettled_fbank_frames)), "ss_encoder_stream_set_tail")

    def encoder_stream_forward(self, fbank: torch.Tensor, attn_chunk: int, conv_chunk: int) -> torch.Tensor:
        """Incremental twin of :meth:`encoder_forward` for streaming: same input (fbank
        of all audio so far) and output, but only the rows that are not final yet are recomputed.
        ``self.stream_stats`` = (rows final after the call, rows recomputed by it)."""
        assert fbank.is_cuda and fbank.dtype == torch.float32 and fbank.is_contiguous()
        T = fbank.shape[0]
        Tp = self.lib.ss_encoder_out_len(T)
        out = torch.empty((Tp, self.cfg.enc_dim), dtype=torch.float32, device=self.device)
        nf, nc = C.c_int32(0), C.c_int32(0)

        def forward():
            L.check(self.lib.ss_encoder_stream_forward(self.h, _stream(), _ptr(fbank), T, int(min(attn_chunk, 1 << 30)),
                                                       int(min(conv_chunk, 1 << 30)), _ptr(out), C.byref(nf), C.byref(nc)),
                    "ss_encoder_stream_forward")
        self._ctc_stash = None
        if self.ctc_speculate and self.persistent_mt > 0 and Tp > 0 and not os.environ.get("SS_NO_CTC_DEFER"):
            # The agents' policy() reads both CTC heads of every encoder output: queue them behind the layers and synchronise ONCE --
            # the encoder's own time-out check is deferred to ss_encoder_stream_status (include/streamspeech_hip.h), so the device
            # runs from the last layer straight into the heads while the host is on its way back (~40 us of a ~1-ms call).
            L.check(self.lib.ss_encoder_stream_set_deferred(self.h, 1), "ss_encoder_stream_set_deferred")
            try:
                forward()
                ibuf = torch.empty((2 * (3 * Tp + 1),), dtype=torch.int32, device=self.device)
                for hd in (0, 1):
                    b = ibuf[hd * (3 * Tp + 1):(hd + 1) * (3 * Tp + 1)]
                    L.check(self.lib.ss_ctc_greedy(self.h, _stream(), hd, _ptr(out), Tp, _ptr(b[:Tp]), _ptr(b[Tp:2 * Tp]), _ptr(b[2 * Tp:3 * Tp]),
                                                   _ptr(b[3 * Tp:]), None), "ss_ctc_greedy")
                host = ibuf.cpu()
                rep = C.c_int32(0)
                L.check(self.lib.ss_encoder_stream_status(self.h, _stream(), C.byref(rep)), "ss_encoder_stream_status")
            finally:
                L.check(self.lib.ss_encoder_stream_set_deferred(self.h, 0), "ss_encoder_stream_set_deferred")
            if rep.value:            # a persistent launch timed out: the call again (one launch per op now), heads on request
                forward()
            else:
                self._ctc_both = ((out.data_ptr(), Tp), {0: host[:3 * Tp + 1], 1: host[3 * Tp + 1:]})
                self.stream_stats = (nf.value, nc.value)
                self._last_enc = (out.data_ptr(), Tp)
                return out
        else:
            forward()
        self._ctc_both = None
        self.stream_stats = (nf.value, nc.value)
        self._last_enc = (out.data_ptr(), Tp)
        return out

    # ---- a8 -------------------------------------------------------------------------------
    # Both CTC heads behind ONE host round trip (off unless a caller sets ``ctc_speculate``; the agents do: policy() always asks for the
    # source head and then the target head of the same encoder output -- reference agent :437-452 -- and each answer used to cost a
    # synchronising device-to-host copy, ~35 us of an ~1-ms call).  The first request for a head of the tensor this engine produced last
    # also runs the OTHER head and parks its host-side answer; the second request takes it.  The parked answer dies with the next
    # encoder call and is handed out once, for the same (pointer, rows) only; a caller that rewrites the encoder output in place between
    # its two requests must leave the switch off.
    ctc_speculate = False

    def _ctc_unpack(self, host, Tp):
        n = int(host[3 * Tp])
        return host[Tp:Tp + n].tolist(), host[2 * Tp:2 * Tp + n].tolist(), host[:Tp], None

    def ctc_greedy(self, head: int, enc_out: torch.Tensor, want_logits: bool = False):
        """-> (tokens list, frame index list, raw argmax tensor, logits or None)."""
        Tp = enc_out.shape[0]
        key = (enc_out.data_ptr(), Tp)
        both = getattr(self, "_ctc_both", None)
        if both is not None and not want_logits and both[0] == key and head in both[1]:
            return self._ctc_unpack(both[1].pop(head), Tp)              # computed behind the streaming encoder call (handed out once)
        stash = getattr(self, "_ctc_stash", None)
        if stash is not None:
            self._ctc_stash = None
            if not want_logits and stash[0] == key and stash[1] == head:
                return self._ctc_unpack(stash[2], Tp)
        if self.ctc_speculate and not want_logits and key == getattr(self, "_last_enc", None):
            ibuf = torch.empty((2 * (3 * Tp + 1),), dtype=torch.int32, device=self.device)
            for k, hd in enumerate((head, 1 - head)):
                b = ibuf[k * (3 * Tp + 1):(k + 1) * (3 * Tp + 1)]
                L.check(self.lib.ss_ctc_greedy(self.h, _stream(), hd, _ptr(enc_out), Tp, _ptr(b[:Tp]), _ptr(b[Tp:2 * Tp]), _ptr(b[2 * Tp:3 * Tp]),
                                               _ptr(b[3 * Tp:]), None), "ss_ctc_greedy")
            host = ibuf.cpu()
            self._ctc_stash = (key, 1 - head, host[3 * Tp + 1:])
            return self._ctc_unpack(host[:3 * Tp + 1], Tp)
        V = self.cfg.src_vocab if head == 0 else self.cfg.tgt_vocab
        ibuf = torch.empty((3 * Tp + 1,), dtype=torch.int32, device=self.device)
        raw, toks, idx, cnt = ibuf[:Tp], ibuf[Tp:2 * Tp], ibuf[2 * Tp:3 * Tp], ibuf[3 * Tp:]
        logits = torch.empty((Tp, V), dtype=torch.float32, device=self.device) if want_logits else None
        L.check(self.lib.ss_ctc_greedy(self.h, _stream(), head, _ptr(enc_out), Tp, _ptr(raw), _ptr(toks), _ptr(idx),
                                       _ptr(cnt), _ptr(logits)), "ss_ctc_greedy")
        host = ibuf.cpu()
        n = int(host[3 * Tp])
        return host[Tp:Tp + n].tolist(), host[2 * Tp:2 * Tp + n].tolist(), host[:Tp], logits

    # ---- a9-a10 ---------------------------------------------------------------------------
    def mt_begin(self, enc_out: torch.Tensor):
        self._mt_enc = enc_out  # keep alive
        L.check(self.lib.ss_mt_begin(self.h, _stream(), _ptr(enc_out), enc_out.shape[0]), "ss_mt_begin")

    def mt_append(self, tokens: List[int], pos0: int, ban_eos: bool, force_eos: bool,
                  want_feats: bool = True, want_next: bool = True, n_tail_pad: int = 0) -> Tuple[Optional[torch.Tensor], Optional[int]]:
        n = len(tokens)
        if any(not 0 <= int(t) < self.cfg.tgt_vocab for t in tokens):      # nn.Embedding raises the same in the reference
            raise IndexError(f"token id outside the target dictionary of {self.cfg.tgt_vocab} entries: {tokens}")
        tok = torch.tensor(tokens, dtype=torch.int32).to(self.device)
        feats = torch.empty((n, self.cfg.dec_dim), dtype=torch.float32, device=self.device) if want_feats else None
        nxt = torch.empty((1,), dtype=torch.int32, device=self.device) if want_next else None
        L.check(self.lib.ss_mt_append(self.h, _stream(), _ptr(tok), n, pos0, int(ban_eos), int(force_eos),
                                      _ptr(feats), _ptr(nxt), n_tail_pad), "ss_mt_append")
        nx = int(nxt.item()) if want_next else None
        if nx is not None and nx < 0:
            # csrc/mt_step.hip: a bounded wait of the persistent decode step timed out (its workgroups were not all resident).
            # Loud, then the same call again with one launch per op -- it rewrites the same cache row and features.
            import warnings
            warnings.warn("persistent MT decode step timed out; this context falls back to one launch per op", RuntimeWarning)
            self.set_persistent_mt_step(0)
            return self.mt_append(tokens, pos0, ban_eos, force_eos, want_feats, want_next, n_tail_pad)
        return feats, nx

    def mt_greedy(self, enc_out: torch.Tensor, prefix: List[int], max_len: int, min_len: int = 1):
        """Beam-1 search in one C call -> (tokens after the prefix incl. final eos, feats [n_fed, D])."""
        self._mt_enc = enc_out
        n_pre = len(prefix)
        cap = max_len + 2
        feats = torch.empty((cap, self.cfg.dec_dim), dtype=torch.float32, device=self.device)
        c_pre = (C.c_int32 * max(n_pre, 1))(*prefix)
        c_out = (C.c_int32 * (max_len + 2 - n_pre))()
        n_out, n_feats = C.c_int(0), C.c_int(0)
        L.check(self.lib.ss_mt_greedy(self.h, _stream(), _ptr(enc_out), enc_out.shape[0], c_pre, n_pre, max_len,
                                      min_len, c_out, C.byref(n_out), _ptr(feats), C.byref(n_feats)), "ss_mt_greedy")
        if getattr(self, "persistent_mt", 0):                # a time-out inside makes the library fall back (and say so on stderr)
            self.persistent_mt = int(self.lib.ss_mt_get_persistent(self.h))
        return list(c_out[: n_out.value]), feats[: n_feats.value]

    def set_persistent_mt_step(self, workgroups: int = 64):
        """One persistent launch per MT decode step (0 restores the launch-per-op form); see ss_mt_set_persistent."""
        L.check(self.lib.ss_mt_set_persistent(self.h, int(workgroups)), "ss_mt_set_persistent")
        self.persistent_mt = int(workgroups)

    def set_pack_invariant(self, on: bool = True):
        """Pack-invariant arithmetic of the batch_* calls of this context (default on): see ss_model_set_pack_invariant."""
        L.check(self.lib.ss_model_set_pack_invariant(self.h, int(bool(on))), "ss_model_set_pack_invariant")

    def pack_invariant(self) -> bool:
        return bool(self.lib.ss_model_get_pack_invariant(self.h))

    def mt_truncate(self, length: int):
        L.check(self.lib.ss_mt_truncate(self.h, length), "ss_mt_truncate")

    # ---- a11-a13 --------------------------------------------------------------------------
    def t2u_units(self, mt_feats: torch.Tensor, t2u_causal: bool = False, mask_eos: bool = False,
                  want_logits: bool = False, n_tail_pad: int = 0):
        """mt_feats [n,512] -> (collapsed unit-vocab tokens list, raw argmax tensor(host), logits or None)."""
        n = mt_feats.shape[0]
        U = n * self.cfg.ctc_upsample
        ibuf = torch.empty((2 * U + 1,), dtype=torch.int32, device=self.device)
        raw, toks, cnt = ibuf[:U], ibuf[U:2 * U], ibuf[2 * U:]
        logits = torch.empty((U, self.cfg.unit_vocab), dtype=torch.float32, device=self.device) if want_logits else None
        L.check(self.lib.ss_t2u_units(self.h, _stream(), _ptr(mt_feats.contiguous()), n, int(t2u_causal),
                                      int(mask_eos), _ptr(raw), _ptr(toks), _ptr(cnt), _ptr(logits), n_tail_pad), "ss_t2u_units")
        host = ibuf.cpu()
        k = int(host[2 * U])
        return host[U:U + k].tolist(), host[:U], logits


    def normalized_probs(self, logits: torch.Tensor, log_probs: bool = True, mask0: int = -1, mask1: int = -1) -> torch.Tensor:
        """model.get_normalized_probs on the device (ss_log_softmax): (log-)softmax over the last axis of dense logits, then ids
        mask0 / mask1 set to -inf (0).  Glue for callers that ask for `lprobs`; the greedy searches never need it."""
        x = logits.to(self.device, torch.float32).contiguous()
        V = x.shape[-1]
        rows = x.numel() // V
        out = torch.empty_like(x)
        L.check(self.lib.ss_log_softmax(_stream(), _ptr(x), rows, V, int(mask0), int(mask1), int(not log_probs), _ptr(out)), "ss_log_softmax")
        return out

    def unit_scores(self, mt_feats: torch.Tensor, t2u_causal: bool = False):
        """Per-position maximum log-probability (natural log, pad / unk / eos masked after the softmax) of the unit
        decoder over the 25 n positions -- the offline search's positional scores (researches/ctc_unity/ctc_generator.py:
        55-63).  -> float32 [U] on the host.  Offline driver with --scores only; not on the timed path."""
        _, _, logits = self.t2u_units(mt_feats, t2u_causal=t2u_causal, mask_eos=True, want_logits=True)
        U, V = logits.shape
        out = torch.empty((U,), dtype=torch.float32, device=self.device)
        c = self.cfg
        L.check(self.lib.ss_row_max_logprob(_stream(), _ptr(logits), U, V, c.pad, c.unk, c.eos, _ptr(out)), "ss_row_max_logprob")
        return out.cpu()


class HipVocoder:
    """ss_vocoder handle (CodeHiFiGANVocoderWithDur replacement)."""

    def __init__(self, generator_state_dict, cfg: VocoderConfig = None, device="cuda:0", _share=None, scratch: "Scratch" = None):
        self.lib = L.load()
        self.cfg = cfg or VocoderConfig()
        self.device = _require_gpu(device)
        if _share is not None:
            names, offsets, numels, self.blob = _share
        else:
            names, offsets, numels, blob = pack_vocoder(generator_state_dict, self.cfg)
            self.blob = blob.to(self.device)
        self._packed = (names, offsets, numels, self.blob)
        c = self.cfg
        cc = L.SSVocoderConfig()
        cc.num_embeddings, cc.embedding_dim, cc.model_in_dim = c.num_embeddings, c.embedding_dim, c.model_in_dim
        cc.upsample_initial_channel = c.upsample_initial_channel
        cc.n_up = len(c.upsample_rates)
        for i, (u, k) in enumerate(zip(c.upsample_rates, c.upsample_kernel_sizes)):
            cc.upsample_rates[i] = u
            cc.upsample_kernel_sizes[i] = k
        cc.n_res = len(c.resblock_kernel_sizes)
        for j, (k, dil) in enumerate(zip(c.resblock_kernel_sizes, c.resblock_dilation_sizes)):
            cc.resblock_kernel_sizes[j] = k
            for t, dv in enumerate(dil):
                cc.resblock_dilations[j][t] = dv
        cc.dur_hidden, cc.dur_kernel = c.dur_hidden, c.dur_kernel
        self.c_cfg = cc
        cn, co, cm, n = _slots(names, offsets, numels)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.ss_vocoder_create(C.byref(cc), _ptr(self.blob), self.blob.numel(), cn, co, cm, n,
                                               C.byref(h)), "ss_vocoder_create")
        self.h = h
        self.scratch = None
        if scratch is not None:
            self.bind_scratch(scratch)
        self.hop = int(np.prod(c.upsample_rates))
        self.max_dur = 64

    def bind_scratch(self, scratch: "Scratch"):
        L.check(self.lib.ss_vocoder_bind_scratch(self.h, scratch.h), "ss_vocoder_bind_scratch")
        self.scratch = scratch

    def new_context(self, scratch: "Scratch" = None) -> "HipVocoder":
        v = HipVocoder(None, self.cfg, device=str(self.device), _share=self._packed, scratch=scratch)
        if getattr(self, "bf16x3", False):
            v.set_bf16x3(True)
        return v

    def set_bf16x3(self, on: bool):
        """Opt-in split-bf16 (3 bf16 MFMAs per k-slice) contraction for the C >= 64 generator convs of this handle; the
        default (off) is exact f32, the reference's arithmetic."""
        L.check(self.lib.ss_vocoder_set_bf16x3(self.h, int(bool(on))), "ss_vocoder_set_bf16x3")
        self.bf16x3 = bool(on)

    def _check_units(self, ids: torch.Tensor):
        """nn.Embedding(num_embeddings, ...) of the reference's CodeGenerator raises IndexError on such ids (codehifigan.py:56-70)."""
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.cfg.num_embeddings):
            raise IndexError(f"unit id outside the vocoder's {self.cfg.num_embeddings} codes")

    def batch_forward(self, codes: List[List[int]], dur_prediction=True, forced_dur: Optional[List[List[int]]] = None):
        """-> (list of wav tensors (views into one packed buffer), list of dur lists)."""
        B = len(codes)
        K = [len(c) for c in codes]
        flat = torch.tensor([u for c in codes for u in c], dtype=torch.int32)
        self._check_units(flat)
        flat = flat.to(self.device)
        fd = None
        if forced_dur is not None:
            fd = torch.tensor([d for ds in forced_dur for d in ds], dtype=torch.int32).to(self.device)
            cap = int(sum(sum(ds) for ds in forced_dur)) * self.hop
        else:
            cap = sum(K) * (self.max_dur if dur_prediction else 1) * self.hop
        wav = torch.empty((cap,), dtype=torch.float32, device=self.device)
        dur = torch.empty((sum(K),), dtype=torch.int32, device=self.device)
        st, ns = (C.c_int64 * B)(), (C.c_int64 * B)()
        L.check(self.lib.ss_batch_vocoder_forward(self.h, _stream(), B, _ptr(flat), _i32(K), int(dur_prediction), _ptr(fd),
                                                  _ptr(wav), cap, _ptr(dur), st, ns), "ss_batch_vocoder_forward")
        wavs = [wav[st[b]: st[b] + ns[b]] for b in range(B)]
        return wavs, dur, K

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ss_vocoder_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def forward(self, codes, dur_prediction: bool = True, forced_dur=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """codes: list/array/tensor of unit ids -> (wav [S] float32 device, dur [K] int32 device)."""
        codes_t = torch.as_tensor(codes, dtype=torch.int32).reshape(-1)
        if not codes_t.is_cuda:
            self._check_units(codes_t)                      # device-resident ids are guarded by the gather kernel instead
        codes_t = codes_t.to(self.device)
        K = codes_t.numel()
        fd = None if forced_dur is None else torch.as_tensor(forced_dur, dtype=torch.int32).reshape(-1).to(self.device)
        cap = K * self.max_dur * self.hop
        wav = torch.empty((cap,), dtype=torch.float32, device=self.device)
        dur = torch.empty((K,), dtype=torch.int32, device=self.device)
        ns = C.c_int64(0)
        rc = self.lib.ss_vocoder_forward(self.h, _stream(), _ptr(codes_t), K, int(dur_prediction), _ptr(fd),
                                         _ptr(wav), cap, _ptr(dur), C.byref(ns))
        L.check(rc, "ss_vocoder_forward")
        return wav[: ns.value], dur
