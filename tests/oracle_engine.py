"""CPU stand-ins with the HipModel / HipVocoder method set, backed by the oracle.  TEST ONLY: lets
the agent's host control flow run without a GPU and gives the streaming GPU test its reference."""
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
