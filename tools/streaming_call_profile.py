"""One synthetic utterance through the drop-in agent's policy() loop on 320-ms segments (incremental encoder state), for a rocprofv3
kernel-stats profile of the streaming path:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x -- python tools/streaming_call_profile.py [seconds=6]
Prints the number of policy() calls by kind so that the per-kernel totals can be divided."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from streamspeech_amd import streaming_eval as SE, synth  # noqa: E402
from streamspeech_amd.agent import StreamSpeechS2STAgent  # noqa: E402
from streamspeech_amd.config import ModelConfig, VocoderConfig  # noqa: E402
from streamspeech_amd.engine import HipModel, HipVocoder  # noqa: E402
from streamspeech_amd.modules import CodeHiFiGANVocoderWithDur, StreamSpeechModel  # noqa: E402


class VocSurface:
    def __init__(self, hv):
        self.hip = hv
    __call__ = CodeHiFiGANVocoderWithDur.__call__


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    cfg, vcfg = ModelConfig(), VocoderConfig()
    model = HipModel(synth.make_model_state_dict(0, cfg), cfg)
    voc = HipVocoder(synth.make_vocoder_state_dict(0, vcfg), vcfg)
    agent = StreamSpeechS2STAgent(bench._agent_args(320), model=StreamSpeechModel.from_engine(model), vocoder=VocSurface(voc))
    pcm = synth.synth_pcm(4321, int(secs * 16000))
    SE.run_utterance(agent, pcm, 320)          # warm
    torch.cuda.synchronize()
    r = SE.run_utterance(agent, pcm, 320)
    print("actions", r["actions"], "calls", r["calls"], "call_ms", [round(c, 2) for c in r["call_ms"]], file=sys.stderr)


if __name__ == "__main__":
    main()
