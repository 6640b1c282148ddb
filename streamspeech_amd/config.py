"""Hyper-parameters of the StreamSpeech S2ST hot path.

Values follow the released ``streamspeech.{offline,simultaneous}.*`` checkpoints:
reference ``researches/ctc_unity/train_scripts/train.offline-s2st.sh`` (arch flags),
``configs/fr-en/config_mtl_asr_st_ctcst.yaml:1-11`` (MT decoder) and
``researches/ctc_unity/models/streamspeech_model.py:418-430`` (arch defaults).
The vocoder block is the public unit-HiFi-GAN config (``config.json`` is not in the
reference tree; structure from ``fairseq/models/text_to_speech/hifigan.py:111-152``).
"""
from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass
class ModelConfig:
    # front-end / encoder (chunk conformer)
    input_feat: int = 80
    conv_channels: int = 1024          # Conv1dSubsampler mid channels
    conv_kernel: int = 5
    enc_dim: int = 256
    enc_ffn: int = 2048
    enc_heads: int = 4
    enc_layers: int = 12
    dw_kernel: int = 31
    max_source_positions: int = 6000
    # CTC heads
    src_vocab: int = 6000              # source_unigram
    tgt_vocab: int = 6000              # target_unigram / ctc_target_unigram
    # MT decoder
    mt_layers: int = 4
    dec_dim: int = 512
    dec_ffn: int = 2048
    dec_heads: int = 8
    # T2U encoder + unit decoder
    t2u_layers: int = 2
    unit_layers: int = 2
    unit_vocab: int = 1005             # 1000 units + 4 specials + <blank>
    ctc_upsample: int = 25
    max_target_positions: int = 1200
    # dictionary conventions (fairseq Dictionary): bos=0 pad=1 eos=2 unk=3
    pad: int = 1
    eos: int = 2
    unk: int = 3

    @property
    def head_dim(self) -> int:
        return self.enc_dim // self.enc_heads

    @property
    def unit_blank(self) -> int:
        # SpeechToSpeechCTCTask appends "<blank>" (reference tasks/speech_to_speech_ctc.py:14-19)
        return self.unit_vocab - 1


@dataclass
class VocoderConfig:
    num_embeddings: int = 1000
    embedding_dim: int = 128
    model_in_dim: int = 128
    upsample_rates: Tuple[int, ...] = (5, 4, 4, 2, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (11, 8, 8, 4, 4)
    upsample_initial_channel: int = 512
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    dur_hidden: int = 128
    dur_kernel: int = 3
    code_hop_size: int = 320

    def channels(self, stage: int) -> int:
        return self.upsample_initial_channel // (2 ** (stage + 1))

    def receptive_field_frames(self) -> int:
        """One-sided receptive field of the generator in input frames (hifigan.py:154-170): how many
        frames to the left of frame f can influence the samples of frame f.  Walked from the output
        back to the input: conv_post k7, per stage the widest resblock (sum over its dilated conv1 +
        plain conv2 pairs), the transposed conv (a 3-tap polyphase filter on the stage input), conv_pre k7."""
        r = 3
        for i in reversed(range(len(self.upsample_rates))):
            r += max(sum((k - 1) // 2 * (d + 1) for d in dil)
                     for k, dil in zip(self.resblock_kernel_sizes, self.resblock_dilation_sizes))
            r = -(-r // self.upsample_rates[i]) + 1
        return r + 3

    def as_dict(self):
        """The JSON the reference ``CodeHiFiGANVocoderWithDur`` is constructed from."""
        return {
            "num_embeddings": self.num_embeddings,
            "embedding_dim": self.embedding_dim,
            "model_in_dim": self.model_in_dim,
            "upsample_rates": list(self.upsample_rates),
            "upsample_kernel_sizes": list(self.upsample_kernel_sizes),
            "upsample_initial_channel": self.upsample_initial_channel,
            "resblock_kernel_sizes": list(self.resblock_kernel_sizes),
            "resblock_dilation_sizes": [list(d) for d in self.resblock_dilation_sizes],
            "dur_predictor_params": {
                "encoder_embed_dim": self.embedding_dim,
                "var_pred_hidden_dim": self.dur_hidden,
                "var_pred_kernel_size": self.dur_kernel,
                "var_pred_dropout": 0.5,
            },
            "code_hop_size": self.code_hop_size,
        }
