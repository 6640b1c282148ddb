"""`simuleval --agent <this file>`: same file name as the reference agent (agent/speech_to_text.asr.streamspeech.agent.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from streamspeech_amd.agent_text import StreamSpeechASRAgent  # noqa: E402,F401
