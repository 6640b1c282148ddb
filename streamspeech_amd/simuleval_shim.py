"""SimulEval agent API surface (``simuleval.agents`` / ``simuleval.data.segments`` / ``@entrypoint``).

If SimulEval is importable its own classes are re-exported, so the agent plugs straight into
``simuleval --agent ...``.  In this image SimulEval's package import fails (missing yt_dlp /
pydub / soundfile), so the same public surface is provided here with identical
names, fields and method behaviour (reference SimulEval/simuleval/agents/agent.py:18-196,
agents/actions.py:12-60, agents/states.py:10-70, data/segments.py:11-56, utils/__init__.py:10-12).
"""
import json
from dataclasses import dataclass, field
from inspect import signature
from typing import List, Optional, Union

try:  # pragma: no cover - depends on the environment
    from simuleval.agents import SpeechToSpeechAgent, SpeechToTextAgent  # type: ignore
    from simuleval.agents.actions import Action, ReadAction, WriteAction  # type: ignore
    from simuleval.agents.states import AgentStates  # type: ignore
    from simuleval.data.segments import EmptySegment, Segment, SpeechSegment, TextSegment  # type: ignore
    from simuleval.utils import entrypoint  # type: ignore
    HAVE_SIMULEVAL = True
except Exception:  # noqa: BLE001
    HAVE_SIMULEVAL = False

    @dataclass
    class Segment:
        index: int = 0
        content: list = field(default_factory=list)
        finished: bool = False
        is_empty: bool = False
        data_type: str = None

        def json(self) -> str:
            return json.dumps(dict(self.__dict__))

        @classmethod
        def from_json(cls, json_string: str):
            return cls(**json.loads(json_string))

    @dataclass
    class EmptySegment(Segment):
        is_empty: bool = True

    @dataclass
    class TextSegment(Segment):
        content: str = ""
        data_type: str = "text"

    @dataclass
    class SpeechSegment(Segment):
        sample_rate: int = -1
        data_type: str = "speech"

    class Action:
        def is_read(self) -> bool:
            raise NotImplementedError

    class ReadAction(Action):
        def is_read(self) -> bool:
            return True

        def __repr__(self):
            return "ReadAction()"

    @dataclass
    class WriteAction(Action):
        content: Union[str, List[float], Segment]
        finished: bool

        def is_read(self) -> bool:
            return False

    class AgentStates:
        def __init__(self) -> None:
            self.reset()

        def reset(self) -> None:
            self.source = []
            self.target = []
            self.source_finished = False
            self.target_finished = False
            self.source_sample_rate = 0
            self.target_sample_rate = 0

        def update_source(self, segment: Segment):
            self.source_finished = segment.finished
            if isinstance(segment, EmptySegment):
                return
            if isinstance(segment, TextSegment):
                self.source.append(segment.content)
            elif isinstance(segment, SpeechSegment):
                self.source += segment.content
                self.source_sample_rate = segment.sample_rate
            else:
                raise NotImplementedError

        def update_target(self, segment: Segment):
            self.target_finished = segment.finished
            if not self.target_finished:
                if isinstance(segment, EmptySegment):
                    return
                if isinstance(segment, TextSegment):
                    self.target.append(segment.content)
                elif isinstance(segment, SpeechSegment):
                    self.target += segment.content
                    self.target_sample_rate = segment.sample_rate
                else:
                    raise NotImplementedError

    _SEGMENT_TYPE = {"text": TextSegment, "speech": SpeechSegment}

    class GenericAgent:
        source_type = None
        target_type = None

        def __init__(self, args=None) -> None:
            if args is not None:
                self.args = args
            assert self.source_type and self.target_type
            self.device = "cpu"
            self.states = self.build_states()
            self.reset()

        def build_states(self):
            return AgentStates()

        def reset(self) -> None:
            self.states.reset()

        def policy(self, states=None):
            raise NotImplementedError

        def push(self, source_segment, states=None) -> None:
            (states or self.states).update_source(source_segment)

        def pop(self, states=None):
            is_stateless = len(signature(self.policy).parameters) != 0
            if not is_stateless and states:
                raise RuntimeError("Feeding states to stateful agents.")
            if states is None:
                states = self.states
            if states.target_finished:
                return EmptySegment(finished=True)
            action = self.policy(states) if is_stateless else self.policy()
            if not isinstance(action, Action):
                raise RuntimeError(f"The return value of {self.policy.__qualname__} is not an Action instance")
            if action.is_read():
                return EmptySegment()
            if isinstance(action.content, Segment):
                return action.content
            segment = _SEGMENT_TYPE[self.target_type](index=0, content=action.content, finished=action.finished)
            states.update_target(segment)
            return segment

        def pushpop(self, segment, states=None):
            self.push(segment, states)
            return self.pop(states)

        @staticmethod
        def add_args(parser):
            pass

        @classmethod
        def from_args(cls, args):
            return cls(args)

        def to(self, device, *a, **k):
            pass

    class SpeechToSpeechAgent(GenericAgent):
        source_type = "speech"
        target_type = "speech"

    class SpeechToTextAgent(GenericAgent):
        source_type = "speech"
        target_type = "text"

    EVALUATION_SYSTEM_LIST = []

    def entrypoint(klass):
        EVALUATION_SYSTEM_LIST.append(klass)
        return klass
