"""MI355X-native YOLO training / detection path.

Importing the package sets GPU_MAX_HW_QUEUES=8 unless the caller chose a value: the ROCm runtime maps HIP streams onto
4 hardware queues by default, and a data-parallel rank has more live streams than that (compute, filter gradients,
the reducer's communication stream, RCCL's own) -- with 4 queues the two COMPUTE streams of the training step landed on
one queue and ran one after the other: 34.2 ms per step instead of 30.1 (profiles/r04_b_dp_readiness.json; 30.5 with 8
queues). The variable is read when the HIP runtime initialises, i.e. at the first device call of the process, so it
must be in the environment before that: import this package (or set the variable) before touching torch.cuda.
"""
import os as _os

if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    import sys as _sys
    _t = _sys.modules.get("torch")
    if _t is not None and _t.cuda.is_initialized():    # too late for this process: say so instead of silently doing nothing
        import warnings as _w
        _w.warn("tf2_yolo_amd: the HIP runtime was initialised before this import, so GPU_MAX_HW_QUEUES=8 does not apply to "
                "this process (the training step's two compute streams may share a hardware queue: ~13 % slower steps). "
                "Import tf2_yolo_amd, or set GPU_MAX_HW_QUEUES, before the first torch.cuda call.")
