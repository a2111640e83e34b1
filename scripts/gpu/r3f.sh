#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
python -m pytest tests/test_gpu_keras_shell.py tests/test_gpu_dp.py -x -q > $O/tests.log 2>&1; echo "tests rc $?"; grep -v "frame #" $O/tests.log | tail -12
YOLO_STEP_MODE=tape python scripts/host_enqueue.py 10 2>&1 | grep -v amdgpu.ids | tee $O/host_tape.log
