"""MI355X-native drop-in for StreamSpeech's S2ST hot path.

Importing the package asks the HIP runtime to keep kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1, unless the variable is
already set): every kernel's first instructions are scalar loads of its arguments, and from host-coherent memory that round trip costs
~2 us per launch -- nothing for the packed path (+0.5 %), 8 % of a streaming policy() call (24 persistent layer launches + 40 small
kernels).  The variable is read when the runtime starts, i.e. at the process's first HIP call: import this package (or
bench.py) before touching torch.cuda."""
import os as _os

_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
