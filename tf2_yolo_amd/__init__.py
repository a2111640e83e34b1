"""MI355X-native YOLO training / detection path.

Importing the package sets GPU_MAX_HW_QUEUES=8 unless the caller chose a value: the ROCm runtime maps HIP streams onto
4 hardware queues by default, and a data-parallel rank has more live streams than that (compute, filter gradients,
the reducer's communication stream, RCCL's own) -- with 4 queues the two COMPUTE streams of the training step landed on
one queue and ran one after the other: 34.2 ms per step instead of 30.1 (profiles/r04_b_dp_readiness.json; 30.5 with 8
queues). The variable is read when the HIP runtime initialises, i.e. at the first device call of the process, so it
must be in the environment before that: import this package (or set the variable) before touching torch.cuda.
"""
import os as _os

_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
