"""fairseq ``--user-dir`` entry point: ``fairseq-generate ... --user-dir streamspeech_amd/fairseq_user_dir`` (or
``--user-dir`` in the SimulEval agent) imports this package (fairseq/fairseq/utils.py:464-511), which registers the
HIP-backed classes under the names the reference's own user dir registers -- model + architecture ``streamspeech``
(researches/ctc_unity/models/streamspeech_model.py:57,418), task ``speech_to_speech_ctc``
(researches/ctc_unity/tasks/speech_to_speech_ctc.py:11), vocoder ``CodeHiFiGANVocoderWithDur`` (agent/tts/vocoder.py:30)
-- into fairseq's registries.  Use it INSTEAD of researches/ctc_unity: fairseq refuses duplicate names."""
from streamspeech_amd.modules import register_with_fairseq

REGISTERED = register_with_fairseq()
