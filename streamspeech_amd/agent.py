"""StreamSpeechS2STAgent -- drop-in for the reference SimulEval agent
(agent/speech_to_speech.streamspeech.agent.py:101-770) with the model math on MI355X.

Same class name, ``@entrypoint``, ``add_args`` flags, ``reset`` / ``policy`` contract
(``ReadAction`` | ``WriteAction(SpeechSegment(content=list[float], sample_rate=16000, finished))``)
and the same read/write gating; every tensor op of the reference policy is replaced by a C-ABI
stage (fbank -> encoder -> CTC x2 -> MT greedy -> T2U + unit decoder -> vocoder).  Like the
reference it recomputes the whole utterance-so-far on every call (incremental state
is the next step, not parity-affecting).
"""
import json
import os
from pathlib import Path

import numpy as np
import torch
import yaml

from .frontend import FEATURE_DIM, ORG_SAMPLE_RATE, SAMPLE_RATE, SHIFT_SIZE, WINDOW_SIZE, OnlineFeatureExtractor
from .generators import CTCDecoder, CTCSequenceGenerator, SequenceGenerator
from .modules import CodeHiFiGANVocoderWithDur, Dictionary, StreamSpeechModel, load_model_state
from .simuleval_shim import ReadAction, SpeechSegment, SpeechToSpeechAgent, WriteAction, entrypoint

DEFAULT_EOS = 2


def _detok(symbols):
    text = "".join(symbols)
    for a, b in (("_", " "), ("▁", " "), ("<unk>", " "), ("<s>", ""), ("</s>", "")):
        text = text.replace(a, b)
    return text[1:] if len(text) > 0 and text[0] == " " else text


def synthesize_tail(vocoder, unit, n_new, dur_prediction, ctx=0, rf=None):
    """Speech of the last ``n_new`` units of ``unit`` (agent :743-753: vocoder over all units, keep
    ``wav[-dur[-n_new:].sum()*320:]``).  With ``ctx`` > 0 only the last ``n_new + ctx`` units are
    synthesised: identical tail, because the frames further left than the
    generator's receptive field ``rf`` cannot reach it.  Returns (tail, wav of what was synthesised)."""
    wav = dur = None
    if ctx > 0 and len(unit) > n_new + ctx:
        x = {"code": torch.tensor(unit[-(n_new + ctx):], dtype=torch.long).view(1, -1)}
        wav, dur = vocoder(x, dur_prediction)
        # frames left of the new units whose durations are exact (the 2 left-most context units see a
        # truncated duration-predictor window) must cover the generator's receptive field
        if int(dur[:, 2:ctx].sum()) < rf + 2:
            wav = None
    if wav is None:
        x = {"code": torch.tensor(unit, dtype=torch.long).view(1, -1)}
        wav, dur = vocoder(x, dur_prediction)
    return wav[-int(dur[:, -n_new:].sum()) * 320:], wav


@entrypoint
class StreamSpeechS2STAgent(SpeechToSpeechAgent):
    """Simultaneous speech-to-speech translation agent for StreamSpeech on the HIP backend."""

    def __init__(self, args, model=None, vocoder=None):
        super().__init__(args)
        self.eos = DEFAULT_EOS
        self.args = args
        self.gpu = True
        self.device = getattr(args, "device_str", "cuda:0")
        self.load_model_vocab(args, model)
        self.max_len = args.max_len
        self.force_finish = args.force_finish
        torch.set_grad_enabled(False)

        eng = self.model.hip if hasattr(self.model, "hip") else self.model
        self.engine = eng
        if hasattr(eng, "set_persistent_mt_step"):       # HIP engine: the MT decode step as one persistent launch (mt_step.hip)
            eng.set_persistent_mt_step(int(getattr(args, "mt_step_workgroups", 64)))
        tgt_dict_mt = self.dict[self.model.mt_task_name]
        tgt_dict = self.dict["tgt"]
        uni = getattr(self.model, "uni_encoder", False)
        self.ctc_generator = CTCSequenceGenerator(tgt_dict, eng, use_incremental_states=False, t2u_causal=uni)
        self.asr_ctc_generator = CTCDecoder(self.dict["source_unigram"], eng, 0)
        self.st_ctc_generator = CTCDecoder(self.dict["ctc_target_unigram"], eng, 1)
        if hasattr(eng, "set_persistent_mt_step"):
            eng.ctc_speculate = True                     # policy() asks for both CTC heads of every encoder output: one host round trip (engine.ctc_greedy)
        # generator_mt of the reference: beam 1, max_len_a=0, max_len_b=100, min_len=1 (agent :162-180)
        self.generator_mt = SequenceGenerator(eng, tgt_dict_mt, beam_size=1, max_len_a=0, max_len_b=100, max_len=0,
                                              min_len=1, eos=tgt_dict_mt.eos(), use_incremental_states=False)
        if vocoder is not None:
            self.vocoder = vocoder
        else:
            vcfg = None
            if args.vocoder_cfg and os.path.exists(args.vocoder_cfg):
                with open(args.vocoder_cfg) as f:
                    vcfg = json.load(f)
            self.vocoder = CodeHiFiGANVocoderWithDur(args.vocoder, vcfg, device=self.device)
        self.dur_prediction = args.dur_prediction
        # Incremental synthesis: the reference re-synthesises ALL units at every write
        # and keeps the tail (agent :743-753).  The generator has a finite receptive field, so the same
        # tail comes out of the last (new + context) units only; context = receptive field + the +-2
        # units the duration predictor looks at + slack.  0 restores the full re-synthesis.
        vcfg = getattr(getattr(self.vocoder, "hip", self.vocoder), "cfg", None)
        rf = vcfg.receptive_field_frames() if vcfg is not None and hasattr(vcfg, "receptive_field_frames") else None
        want = getattr(args, "vocoder_context_units", -1)
        self.vocoder_rf = rf
        self.vocoder_ctx = 0 if (rf is None or want == 0) else (want if want > 0 else rf + 8)
        self.lagging_k1, self.lagging_k2 = args.lagging_k1, args.lagging_k2
        self.segment_size = args.segment_size
        self.stride_n, self.stride_n2 = args.stride_n, args.stride_n2
        self.unit_per_subword = args.unit_per_subword
        if args.extra_output_dir is not None:
            self.asr_file = Path(args.extra_output_dir + "/asr.txt")
            self.st_file = Path(args.extra_output_dir + "/st.txt")
            self.unit_file = Path(args.extra_output_dir + "/unit.txt")
            self.quiet = False
        else:
            self.quiet = True
        self.output_asr_translation = args.output_asr_translation
        self.whole_word = args.source_segment_size >= 640
        self.reset()

    @staticmethod
    def add_args(parser):
        a = parser.add_argument
        a("--model-path", type=str, required=True, help="path to your pretrained model (or synthetic:<seed>)")
        a("--data-bin", type=str, required=True, help="Path of data binary")
        a("--config-yaml", type=str, default=None, help="Path to config yaml file")
        a("--multitask-config-yaml", type=str, default=None, help="Path to config yaml file")
        a("--global-stats", type=str, default=None, help="Path to json file containing cmvn stats")
        a("--tgt-splitter-type", type=str, default="SentencePiece", help="Subword splitter type for target text")
        a("--tgt-splitter-path", type=str, default=None, help="Subword splitter model path for target text")
        a("--user-dir", type=str, default="researches/ctc_unity", help="User directory for model")
        a("--agent-dir", type=str, default="agent", help="User directory for agents")
        a("--max-len", type=int, default=200, help="Max length of translation")
        a("--force-finish", default=False, action="store_true", help="Force the model to finish the hypothsis")
        a("--shift-size", type=int, default=SHIFT_SIZE, help="Shift size of feature extraction window.")
        a("--window-size", type=int, default=WINDOW_SIZE, help="Window size of feature extraction window.")
        a("--sample-rate", type=int, default=ORG_SAMPLE_RATE, help="Sample rate")
        a("--feature-dim", type=int, default=FEATURE_DIM, help="Acoustic feature dimension.")
        a("--vocoder", type=str, required=True, help="path to the CodeHiFiGAN vocoder (or synthetic:<seed>)")
        a("--vocoder-cfg", type=str, required=False, default=None, help="path to the CodeHiFiGAN vocoder config")
        a("--dur-prediction", action="store_true", help="enable duration prediction (for reduced/unique code sequences)")
        a("--lagging-k1", type=int, default=0, help="lagging number")
        a("--lagging-k2", type=int, default=0, help="lagging number")
        a("--segment-size", type=int, default=320, help="segment-size")
        a("--stride-n", type=int, default=1, help="lagging number")
        a("--stride-n2", type=int, default=1, help="lagging number")
        a("--unit-per-subword", type=int, default=15, help="lagging number")
        a("--full-recompute-encoder", default=False, action="store_true",
          help="re-encode all received audio at every policy() call like the reference (default: reuse final rows)")
        a("--vocoder-context-units", type=int, default=-1,
          help="left-context units re-synthesised with each new unit tail (-1: receptive field + 8, 0: all units like the reference)")
        a("--mt-step-workgroups", type=int, default=64,
          help="first-pass text decoder: one persistent launch per decode step on this many workgroups (64 | 128 | 256; the agent "
               "decodes one utterance at a time, which is what that form needs); 0: one launch per op")
        a("--extra-output-dir", type=str, default=None, help="extra output dir")
        a("--output-asr-translation", type=bool, default=False, help="extra output dir")

    def reset(self):
        self.src_seg_num = 0
        self.tgt_subwords_indices = None
        self.src_ctc_indices = None
        self.src_ctc_prefix_length = 0
        self.tgt_ctc_prefix_length = 0
        self.tgt_units_indices = None
        self.prev_output_tokens_mt = None
        self.tgt_text = []
        self.mt_decoder_out = None
        self.unit = None
        self.wav = []
        self.post_transcription = ""
        self.unfinished_wav = None
        self.states.reset()
        enc = getattr(getattr(self, "model", None), "encoder", None)
        if enc is not None and hasattr(enc, "reset_stream"):
            enc.reset_stream()                     # incremental encoder cache: one utterance at a time
        fe = getattr(self, "feature_extractor", None)
        if fe is not None:
            fe.clear_cache()                       # converted sample history of the previous utterance
        try:
            self.generator_mt.reset_incremental_states()
            self.ctc_generator.reset_incremental_states()
        except Exception:  # noqa: BLE001
            pass

    def load_model_vocab(self, args, model=None):
        """agent :355-420: checkpoint -> model, CMVN stats, dictionaries, chunk sizes."""
        args.global_cmvn = None
        config = {}
        if args.config_yaml is not None:
            ypath = os.path.join(args.data_bin, args.config_yaml)
            if os.path.exists(ypath):
                with open(ypath, "r") as f:
                    config = yaml.load(f, Loader=yaml.BaseLoader) or {}
                if "global_cmvn" in config:
                    npz = config["global_cmvn"]["stats_npz_path"]
                    if not os.path.exists(npz):  # reference YAMLs hold absolute paths of the authors' machine
                        npz = os.path.join(args.data_bin, os.path.basename(npz))
                    args.global_cmvn = np.load(npz)
        if args.global_cmvn is None and getattr(args, "global_stats", None):
            args.global_cmvn = np.load(args.global_stats)
        if model is None:
            sd, uni = load_model_state(args.model_path)     # raises IOError("Model file not found") like the agent
            model = StreamSpeechModel(sd, device=self.device, cmvn=args.global_cmvn, uni_encoder=uni)
        self.model = model
        self.models = [model]
        eng = model.hip if hasattr(model, "hip") else model
        self.feature_extractor = OnlineFeatureExtractor(args, eng)

        chunk_size = args.source_segment_size // 40
        model.encoder.chunk_size = chunk_size
        conv_chunk = 16 if chunk_size >= 16 else 8
        for conv in model.encoder.subsample.conv_layers:
            conv.chunk_size = conv_chunk
        for layer in model.encoder.conformer_layers:
            layer.conv_module.depthwise_conv.chunk_size = conv_chunk
        if hasattr(model.encoder, "incremental"):
            model.encoder.incremental = not getattr(args, "full_recompute_encoder", False)
            # a resampled source re-computes its last ~10 output samples when more audio arrives (zero-padded
            # FIR edge), so the newest fbank frame is not settled: the incremental encoder must not cache rows
            # that can see it
            if hasattr(eng, "encoder_stream_set_tail"):
                from .frontend import unsettled_fbank_frames
                eng.encoder_stream_set_tail(unsettled_fbank_frames(int(args.sample_rate), SAMPLE_RATE,
                                                                   int(args.shift_size * SAMPLE_RATE / 1000)))

        # dictionaries: target units + the three multitask text dictionaries
        self.dict = {"tgt": Dictionary.units(1000)}
        mt_cfg = {}
        if args.multitask_config_yaml is not None:
            mpath = os.path.join(args.data_bin, args.multitask_config_yaml)
            if os.path.exists(mpath):
                with open(mpath) as f:
                    mt_cfg = yaml.load(f, Loader=yaml.BaseLoader) or {}
        cfg = eng.cfg
        for name, n in (("target_unigram", cfg.tgt_vocab), ("source_unigram", cfg.src_vocab),
                        ("ctc_target_unigram", cfg.tgt_vocab)):
            path = (mt_cfg.get(name) or {}).get("dict")
            if path and not os.path.exists(path):
                path = os.path.join(args.data_bin, *Path(path).parts[-2:])
            self.dict[name] = Dictionary.load(path) if path and os.path.exists(path) else Dictionary.placeholder(n)

    @torch.inference_mode()
    def policy(self):
        feature = self.feature_extractor(self.states.source)
        if feature.size(0) == 0 and not self.states.source_finished:
            return ReadAction()
        if feature.size(0) < 1:
            return self._finish_empty() if self.states.source_finished else ReadAction()

        src_indices = feature.unsqueeze(0)
        src_lengths = torch.tensor([feature.size(0)]).long()
        model = self.model
        encoder_out = model.encoder(src_indices, src_lengths)
        self.encoder_outs = [encoder_out]

        # ASR / ST CTC heads (agent :437-478)
        finalized_asr = self.asr_ctc_generator.generate(encoder_out, aux_task_name="source_unigram")
        src_ctc_indices = finalized_asr[0][0]["tokens"].int()
        if (self.states.source_finished and not self.quiet) or self.output_asr_translation:
            text = _detok([self.dict["source_unigram"][c] for c in src_ctc_indices])
            if self.states.source_finished and not self.quiet:
                with open(self.asr_file, "a") as f:
                    print(text, file=f)
            if self.output_asr_translation:
                print("Streaming ASR:", text)
        finalized_st = self.st_ctc_generator.generate(encoder_out, aux_task_name="ctc_target_unigram")
        tgt_ctc_indices = finalized_st[0][0]["tokens"].int()

        # read/write gate on the CTC token counts (agent :480-512)
        if not self.states.source_finished:
            src_ctc_prefix_length = src_ctc_indices.size(-1)
            tgt_ctc_prefix_length = tgt_ctc_indices.size(-1)
            self.src_ctc_indices = src_ctc_indices
            if (src_ctc_prefix_length < self.src_ctc_prefix_length + self.stride_n
                    or tgt_ctc_prefix_length < self.tgt_ctc_prefix_length + self.stride_n):
                return ReadAction()
            self.src_ctc_prefix_length = max(src_ctc_prefix_length, self.src_ctc_prefix_length)
            self.tgt_ctc_prefix_length = max(tgt_ctc_prefix_length, self.tgt_ctc_prefix_length)
            subword_tokens = ((tgt_ctc_prefix_length - self.lagging_k1) // self.stride_n) * self.stride_n
            if self.whole_word:
                subword_tokens += 1
            new_subword_tokens = (subword_tokens - self.tgt_subwords_indices.size(-1)
                                  if self.tgt_subwords_indices is not None else subword_tokens)
            if new_subword_tokens < 1:
                return ReadAction()
        else:
            self.src_ctc_indices = src_ctc_indices
            new_subword_tokens = -1
        new_subword_tokens = int(new_subword_tokens)

        # 1. MT decoder: greedy continuation of the committed prefix (agent :520-538)
        finalized_mt = self.generator_mt.generate_decoder(
            self.encoder_outs, src_indices, src_lengths, {"id": 1}, self.tgt_subwords_indices, None, None,
            aux_task_name=model.mt_task_name, max_new_tokens=new_subword_tokens)
        hyp = finalized_mt[0][0]
        if hyp["tokens"][-1] == 2:
            tgt_subwords_indices = hyp["tokens"][:-1].unsqueeze(0)
        else:
            tgt_subwords_indices = hyp["tokens"].unsqueeze(0)

        if self.whole_word:  # agent :540-574 (the KV-cache surgery there is dead code: no incremental states)
            if not self.states.source_finished:
                j = 999999
                for j in range(tgt_subwords_indices.size(-1) - 1, -1, -1):
                    if self.generator_mt.tgt_dict[tgt_subwords_indices[0][j]].startswith("▁"):
                        break
                tgt_subwords_indices = tgt_subwords_indices[:, :j]
                hyp["tokens"] = hyp["tokens"][:j]
                if j == 0:
                    return ReadAction()

        # agent :576-591: max_tgt_len = len(tokens) (+1 in whole-word mode), row = [eos, tokens-without-eos, <pad>...].
        # A non-final whole-word hypothesis was cut to [:j] above and lost its eos, so its row has NO pad; a
        # final one keeps the eos and gets exactly one trailing <pad> position.
        max_tgt_len = len(hyp["tokens"]) + (1 if self.whole_word else 0)
        tmp = hyp["tokens"].int()
        if len(tmp) > 0 and tmp[-1] == self.generator_mt.eos:
            tmp = tmp[:-1]
        n_tail_pad = max_tgt_len - (len(tmp) + 1)
        assert n_tail_pad in (0, 1), "hypothesis without eos outside whole-word mode (the reference fails here too)"
        prev_output_tokens_mt = torch.full((1, max_tgt_len), self.model.target_unigram_decoder.padding_idx
                                           if hasattr(self.model, "target_unigram_decoder") else 1, dtype=torch.int32)
        prev_output_tokens_mt[0, 0] = self.generator_mt.eos
        prev_output_tokens_mt[0, 1:len(tmp) + 1] = tmp
        if (self.states.source_finished and not self.quiet) or self.output_asr_translation:
            text = _detok([self.generator_mt.tgt_dict[c] for c in tmp])
            if self.states.source_finished and not self.quiet:
                with open(self.st_file, "a") as f:
                    print(text, file=f)
            if self.output_asr_translation:
                print("Simultaneous translation:", text)

        if self.tgt_subwords_indices is not None and torch.equal(self.tgt_subwords_indices, tgt_subwords_indices):
            if not self.states.source_finished:
                return ReadAction()
            return self._finish_empty()
        self.tgt_subwords_indices = tgt_subwords_indices

        if not self.states.source_finished and self.prev_output_tokens_mt is not None:
            if (torch.equal(self.prev_output_tokens_mt, prev_output_tokens_mt)
                    or prev_output_tokens_mt.size(-1) <= self.prev_output_tokens_mt.size(-1)):
                return ReadAction()
        self.prev_output_tokens_mt = prev_output_tokens_mt

        # MT decoder states of [eos, tokens...] = the features the greedy pass already produced
        # (causal: a prefix slice is exact); the reference recomputes them (agent :638-651).
        mt_feats = hyp["features"][: len(tmp) + 1]
        if n_tail_pad:
            # whole-word mode: prev_output_tokens_mt carries one trailing <pad> (agent :576-584); the
            # reference runs that position through the MT decoder, T2U encoder and unit decoder with
            # key-padding masks, and its 25 unit positions are decoded like any other
            eng = self.engine
            eng.mt_truncate(len(tmp) + 1)
            pad_feat, _ = eng.mt_append([int(prev_output_tokens_mt[0, -1])], len(tmp) + 1, False, False,
                                        want_next=False, n_tail_pad=1)
            mt_feats = torch.cat((mt_feats, pad_feat), 0)
        self.mt_decoder_out = mt_feats

        # 2+3. T2U encoder + CTC unit decoder + CTC search (agent :661-689)
        finalized = self.ctc_generator.generate(mt_feats, prefix=self.tgt_units_indices, n_tail_pad=n_tail_pad)
        if len(finalized[0][0]["tokens"]) == 0:
            if not self.states.source_finished:
                return ReadAction()
            return self._finish_empty()
        tmp_u = finalized[0][0]["tokens"].int()
        if tmp_u[-1] == self.eos:
            tmp_u = tmp_u[:-1]
        unit = []
        for c in tmp_u:
            u = self.dict["tgt"][c].replace("<s>", "").replace("</s>", "")
            if u != "":
                unit.append(int(u))
        if self.states.source_finished and not self.quiet:
            with open(self.unit_file, "a") as f:
                print(" ".join(str(_) for _ in unit), file=f)
        cur_unit = unit if self.unit is None else unit[len(self.unit):]
        if len(unit) < 1 or len(cur_unit) < 1:
            if not self.states.source_finished:
                return ReadAction()
            return self._finish_empty()

        # 4. vocoder over ALL units so far; emit the tail that belongs to the new units (agent :743-753)
        new_wav, wav = synthesize_tail(self.vocoder, unit, len(cur_unit), self.dur_prediction, self.vocoder_ctx,
                                       self.vocoder_rf)
        if self.unfinished_wav is not None and len(self.unfinished_wav) > 0:
            new_wav = torch.cat((self.unfinished_wav, new_wav), dim=0)
        self.wav = wav
        self.unit = unit

        if self.states.source_finished and new_subword_tokens == -1:
            self.states.target_finished = True
            # self.reset() in the reference (agent :759-761) also clears the states; kept
            self.reset()
        return WriteAction(
            SpeechSegment(content=new_wav.tolist(), sample_rate=SAMPLE_RATE, finished=self.states.source_finished),
            finished=self.states.target_finished)

    def _finish_empty(self):
        return WriteAction(
            SpeechSegment(content=(self.unfinished_wav.tolist() if self.unfinished_wav is not None else []),
                          sample_rate=SAMPLE_RATE, finished=True),
            finished=True)
