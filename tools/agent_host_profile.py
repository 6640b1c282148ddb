import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tools'))
import torch
import bench
from streamspeech_amd import streaming_eval as SE, synth
from streamspeech_amd.agent import StreamSpeechS2STAgent
from streamspeech_amd.config import ModelConfig, VocoderConfig
from streamspeech_amd.engine import HipModel, HipVocoder
from streamspeech_amd.modules import CodeHiFiGANVocoderWithDur, StreamSpeechModel
class VocSurface:
    def __init__(self, hv): self.hip = hv
    __call__ = CodeHiFiGANVocoderWithDur.__call__
cfg, vcfg = ModelConfig(), VocoderConfig()
model = HipModel(synth.make_model_state_dict(0, cfg), cfg)
voc = HipVocoder(synth.make_vocoder_state_dict(0, vcfg), vcfg)
agent = StreamSpeechS2STAgent(bench._agent_args(320), model=StreamSpeechModel.from_engine(model), vocoder=VocSurface(voc))
pcm = synth.synth_pcm(4321, int(6 * 16000))
for _ in range(3): SE.run_utterance(agent, pcm, 320)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): r = SE.run_utterance(agent, pcm, 320)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(35); print(s.getvalue()[:9000])
print(r['calls'], [round(c, 2) for c in r['call_ms']])
