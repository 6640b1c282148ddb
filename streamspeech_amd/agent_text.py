"""Streaming-ASR and simultaneous-S2TT agents of StreamSpeech on the HIP backend: strict prefixes of
the S2ST path -- same encoder + CTC heads (+ MT greedy search), text out.

Reference: agent/speech_to_text.asr.streamspeech.agent.py:385-433 and
agent/speech_to_text.s2tt.streamspeech.agent.py:381-545 (same flags as the S2ST agent minus the
vocoder ones)."""
import torch

from .agent import StreamSpeechS2STAgent, _detok
from .frontend import OnlineFeatureExtractor  # noqa: F401  (re-exported for parity with the reference files)
from .generators import CTCDecoder, SequenceGenerator
from .simuleval_shim import ReadAction, SpeechToTextAgent, WriteAction, entrypoint


def _add_text_args(parser):
    StreamSpeechS2STAgent.add_args(parser)
    for act in parser._actions:           # the text agents have no vocoder
        if "--vocoder" in act.option_strings:
            act.required = False


class _TextAgentBase(SpeechToTextAgent):
    def __init__(self, args, model=None):
        super().__init__(args)
        self.args = args
        self.device = getattr(args, "device_str", "cuda:0")
        # checkpoint / CMVN / dictionaries / chunk sizes exactly as the S2ST agent (agent :355-420)
        StreamSpeechS2STAgent.load_model_vocab(self, args, model)
        torch.set_grad_enabled(False)
        eng = self.model.hip if hasattr(self.model, "hip") else self.model
        self.engine = eng
        if hasattr(eng, "set_persistent_mt_step"):       # HIP engine: the MT decode step as one persistent launch (mt_step.hip)
            eng.set_persistent_mt_step(int(getattr(args, "mt_step_workgroups", 64)))
        self.asr_ctc_generator = CTCDecoder(self.dict["source_unigram"], eng, 0)
        self.st_ctc_generator = CTCDecoder(self.dict["ctc_target_unigram"], eng, 1)
        if hasattr(eng, "set_persistent_mt_step"):       # both CTC heads per call only in the translation agent (engine.ctc_greedy)
            eng.ctc_speculate = type(self).__name__ == "StreamSpeechS2TTAgent"
        tgt_dict_mt = self.dict[self.model.mt_task_name]
        # the text agents search with max_len_a=1, max_len_b=200 (s2tt agent :161-180) -- NOT the S2ST agent's 0 / 100:
        # a final hypothesis may run to (fbank frames + 200) subwords
        self.generator_mt = SequenceGenerator(eng, tgt_dict_mt, beam_size=1, max_len_a=1, max_len_b=200, max_len=0,
                                              min_len=1, eos=tgt_dict_mt.eos(), use_incremental_states=False)
        self.lagging_k1, self.stride_n = args.lagging_k1, args.stride_n
        self.quiet = args.extra_output_dir is None
        if not self.quiet:
            from pathlib import Path
            self.asr_file = Path(args.extra_output_dir + "/asr.txt")
            self.st_file = Path(args.extra_output_dir + "/st.txt")
        self.reset()

    add_args = staticmethod(_add_text_args)

    def reset(self):
        self.tgt_subwords_indices = None
        self.src_ctc_prefix_length = 0
        self.tgt_ctc_prefix_length = 0
        self.asr_text = ""
        self.tgt_text = ""
        self.states.reset()
        enc = getattr(getattr(self, "model", None), "encoder", None)
        if enc is not None and hasattr(enc, "reset_stream"):
            enc.reset_stream()                     # incremental encoder cache: one utterance at a time
        fe = getattr(self, "feature_extractor", None)
        if fe is not None:
            fe.clear_cache()                       # converted sample history of the previous utterance

    def _encode(self):
        feature = self.feature_extractor(self.states.source)
        if feature.size(0) == 0:
            return None, None, None
        src_indices = feature.unsqueeze(0)
        return self.model.encoder(src_indices, None), src_indices, torch.tensor([feature.size(0)]).long()


@entrypoint
class StreamSpeechASRAgent(_TextAgentBase):
    """Streaming ASR: encoder + source_unigram CTC greedy, emits the newly confirmed subwords."""

    @torch.inference_mode()
    def policy(self):
        enc, _, _ = self._encode()
        if enc is None:
            return WriteAction("", finished=True) if self.states.source_finished else ReadAction()
        hyp = self.asr_ctc_generator.generate(enc, aux_task_name="source_unigram")[0][0]
        tokens = [self.dict["source_unigram"][c] for c in hyp["tokens"].int()]
        if self.states.source_finished and not self.quiet:
            with open(self.asr_file, "a") as f:
                print(_detok(tokens), file=f)
        text = " ".join(tokens)
        new_text = text[len(self.asr_text):]
        self.asr_text = text
        if self.states.source_finished:
            self.states.target_finished = True
            self.reset()
        return WriteAction(new_text, finished=self.states.target_finished)


@entrypoint
class StreamSpeechS2TTAgent(_TextAgentBase):
    """Simultaneous speech-to-text translation: encoder + both CTC heads (read/write gate) + MT greedy."""

    @torch.inference_mode()
    def policy(self):
        enc, src_indices, src_lengths = self._encode()
        if enc is None:
            return WriteAction("", finished=True) if self.states.source_finished else ReadAction()
        src_ctc = self.asr_ctc_generator.generate(enc, aux_task_name="source_unigram")[0][0]["tokens"].int()
        tgt_ctc = self.st_ctc_generator.generate(enc, aux_task_name="ctc_target_unigram")[0][0]["tokens"].int()
        if not self.states.source_finished:
            ns, nt = src_ctc.size(-1), tgt_ctc.size(-1)
            if ns < self.src_ctc_prefix_length + self.stride_n or nt < self.tgt_ctc_prefix_length + self.stride_n:
                return ReadAction()
            self.src_ctc_prefix_length = max(ns, self.src_ctc_prefix_length)
            self.tgt_ctc_prefix_length = max(nt, self.tgt_ctc_prefix_length)
            subword_tokens = ((nt - self.lagging_k1) // self.stride_n) * self.stride_n
            new_subword_tokens = (subword_tokens - self.tgt_subwords_indices.size(-1)
                                  if self.tgt_subwords_indices is not None else subword_tokens)
            if new_subword_tokens < 1:
                return ReadAction()
        else:
            new_subword_tokens = -1
        hyp = self.generator_mt.generate_decoder([enc], src_indices, src_lengths, {"id": 1}, self.tgt_subwords_indices,
                                                 None, None, aux_task_name=self.model.mt_task_name,
                                                 max_new_tokens=int(new_subword_tokens))[0][0]
        toks = hyp["tokens"]
        tgt_subwords_indices = (toks[:-1] if toks[-1] == 2 else toks).unsqueeze(0)
        tokens = [self.generator_mt.tgt_dict[c] for c in tgt_subwords_indices[0]]
        if self.states.source_finished and not self.quiet:
            with open(self.st_file, "a") as f:
                print(_detok(tokens), file=f)
        if self.tgt_subwords_indices is not None and torch.equal(self.tgt_subwords_indices, tgt_subwords_indices):
            return WriteAction("", finished=True) if self.states.source_finished else ReadAction()
        self.tgt_subwords_indices = tgt_subwords_indices
        text = " ".join(tokens)
        new_text = text[len(self.tgt_text):]
        self.tgt_text = text
        if self.states.source_finished and new_subword_tokens == -1:
            self.states.target_finished = True
            self.reset()
        return WriteAction(new_text, finished=self.states.target_finished)
