"""`simuleval --agent <this file>`: same file name as the reference agent
(agent/speech_to_speech.streamspeech.agent.py); the @entrypoint class lives in streamspeech_amd.agent."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from streamspeech_amd.agent import StreamSpeechS2STAgent  # noqa: E402,F401
