"""CPU stand-ins with the HipModel / HipVocoder method set, backed by the oracle.  TEST INFRASTRUCTURE
(like everything under oracle/): lets the agent's host control flow run without a GPU, gives the
streaming GPU test its reference, and is what bench.py's cpu_baseline legs time (the reference agent's
per-chunk full recompute, agent/speech_to_speech.streamspeech.agent.py:425-435,686-689,748-751, on the
host cores).  The product path never imports it."""
import numpy as np
import torch

from oracle import kaldi_fbank as K
from oracle import streamspeech_oracle as O


class OracleEngine:
    def __init__(self, sd, cfg, cmvn_mean=None, cmvn_std=None):
        self.sd, self.cfg = O.SD(sd), cfg
        self.device = torch.device("cpu")
        self.mean = np.zeros(80, np.float32) if cmvn_mean is None else np.asarray(cmvn_mean, np.float32)
        self.std = np.ones(80, np.float32) if cmvn_std is None else np.asarray(cmvn_std, np.float32)
        self._tokens, self._enc = [], None

    def resample(self, pcm, sr_in, sr_out=16000):
        from oracle.resample import resample_poly_ref
        return torch.from_numpy(resample_poly_ref(pcm.cpu().numpy(), sr_out, sr_in))

    def fbank_cmvn(self, pcm, pcm_scale=32768.0):
        x = pcm.cpu().numpy().astype(np.float32) * np.float32(pcm_scale)
        return torch.from_numpy(K.global_cmvn(K.fbank(x), self.mean, self.std))

    def encoder_forward(self, fbank, attn_chunk=999999, conv_chunk=999999):
        return O.encoder_forward(self.sd, fbank.cpu(), self.cfg, attn_chunk, conv_chunk)

    def ctc_greedy(self, head, enc_out, want_logits=False):
        name = "source_unigram" if head == 0 else "ctc_target_unigram"
        toks, idx, raw, logits = O.ctc_head(self.sd, enc_out.cpu(), name, self.cfg)
        return toks, idx, torch.tensor(raw, dtype=torch.int32), (logits if want_logits else None)

    def normalized_probs(self, logits, log_probs=True, mask0=-1, mask1=-1):
        """fairseq_model.py:60-77 get_normalized_probs (+ agent/ctc_decoder.py:58-60 pad / unk masking)."""
        x = logits.float()
        out = torch.log_softmax(x, -1) if log_probs else torch.softmax(x, -1)
        for m in (mask0, mask1):
            if m >= 0:
                out[..., m] = float("-inf") if log_probs else 0.0
        return out

    def mt_begin(self, enc_out):
        self._enc, self._tokens = enc_out.cpu(), []

    def mt_append(self, tokens, pos0, ban_eos, force_eos, want_feats=True, want_next=True, n_tail_pad=0):
        self._tokens = self._tokens[:pos0] + list(tokens)
        feats = O.mt_decoder_features(self.sd, self._tokens, self._enc, self.cfg)
        nxt = None
        if want_next:
            E = self.sd["target_unigram_decoder.output_projection.weight"]
            lp = torch.log_softmax(torch.nn.functional.linear(feats[-1], E), -1)
            lp[lp != lp] = float("-inf")
            lp[self.cfg.pad] = float("-inf")
            if force_eos:
                nxt = self.cfg.eos
            else:
                if ban_eos:
                    lp[self.cfg.eos] = float("-inf")
                nxt = int(lp.argmax())
        return (feats[pos0:] if want_feats else None), nxt

    def mt_truncate(self, length):
        self._tokens = self._tokens[:length]

    def t2u_units(self, mt_feats, t2u_causal=False, mask_eos=False, want_logits=False, n_tail_pad=0):
        t2u = O.t2u_encoder(self.sd, mt_feats.cpu(), self.cfg, causal=t2u_causal, n_tail_pad=n_tail_pad)
        logits = O.unit_decoder_logits(self.sd, t2u, self.cfg, n_tail_pad=n_tail_pad)
        lp = torch.log_softmax(logits, -1)
        lp[:, self.cfg.pad] = float("-inf")
        lp[:, self.cfg.unk] = float("-inf")
        if mask_eos:
            lp[:, self.cfg.eos] = float("-inf")
        raw = lp.argmax(-1).tolist()
        toks, _ = O.ctc_collapse(raw, self.cfg.unit_blank, self.cfg.pad)
        return toks, torch.tensor(raw, dtype=torch.int32), (logits if want_logits else None)


    def unit_scores(self, mt_feats, t2u_causal=False):
        """researches/ctc_unity/ctc_generator.py:55-63: log_softmax, pad / unk / eos -> -inf, max over the vocabulary."""
        t2u = O.t2u_encoder(self.sd, mt_feats.cpu(), self.cfg, causal=t2u_causal)
        lp = torch.log_softmax(O.unit_decoder_logits(self.sd, t2u, self.cfg).float(), -1)
        lp[:, [self.cfg.pad, self.cfg.unk, self.cfg.eos]] = float("-inf")
        return lp.max(-1).values

    # ---- ragged-batch method set (one utterance after the other: the B = 1 arithmetic is the definition) ----
    def batch_fbank_cmvn(self, pcm_packed, n_samples, pcm_scale=32768.0):
        feats, T, off = [], [], 0
        for n in n_samples:
            f = self.fbank_cmvn(pcm_packed[off:off + n], pcm_scale)
            feats.append(f); T.append(f.shape[0]); off += n
        return torch.cat(feats), T

    def batch_encoder_forward(self, fbank_packed, T, attn_chunk=999999, conv_chunk=999999):
        outs, off = [], 0
        for t in T:
            outs.append(self.encoder_forward(fbank_packed[off:off + t], attn_chunk, conv_chunk)); off += t
        return torch.cat(outs), [o.shape[0] for o in outs]

    def batch_ctc_greedy(self, head, enc_packed, Tp):
        out, off = [], 0
        for tp in Tp:
            toks, idx, _, _ = self.ctc_greedy(head, enc_packed[off:off + tp]); off += tp
            out.append((toks, idx))
        return out

    def batch_mt_greedy(self, enc_packed, Tp, max_len, min_len=1):
        toks, feats, n, off = [], [], [], 0
        for tp, ml in zip(Tp, max_len):
            t = O.mt_greedy(self.sd, enc_packed[off:off + tp], self.cfg, max_new_tokens=ml); off += tp
            f = O.mt_decoder_features(self.sd, [self.cfg.eos] + [x for x in t if x != self.cfg.eos], enc_packed[off - tp:off], self.cfg)
            toks.append(t); feats.append(f); n.append(f.shape[0])
        rows = max(n)
        packed = torch.zeros((len(Tp), rows, self.cfg.dec_dim))
        for b, f in enumerate(feats):
            packed[b, :f.shape[0]] = f
        return toks, packed, n

    def batch_t2u_units(self, feats, n_rows, t2u_causal=False, mask_eos=False):
        return [self.t2u_units(feats[b][:n], t2u_causal, mask_eos)[0] for b, n in enumerate(n_rows)]


class OracleVocoder:
    def __init__(self, vsd, vcfg):
        self.vsd, self.vcfg = O.SD(vsd), vcfg
        self.cfg = vcfg
        self.call_lengths = []

    def __call__(self, x, dur_prediction=False):
        code = x["code"]
        code = code[code >= 0].view(-1).tolist()
        self.call_lengths.append(len(code))
        wav, dur = O.vocoder_forward(self.vsd, code, self.vcfg, dur_prediction)
        return wav, dur.view(1, -1)

    def batch_forward(self, codes, dur_prediction=True, forced_dur=None):
        wavs, durs = [], []
        for b, c in enumerate(codes):
            w, d = O.vocoder_forward(self.vsd, list(c), self.vcfg, dur_prediction,
                                     forced_dur=None if forced_dur is None else forced_dur[b])
            wavs.append(w); durs.append(d.view(-1).tolist())
        return wavs, durs, [len(c) for c in codes]
