#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O
python scripts/step_determinism.py 96 4 2>&1 | grep -v amdgpu.ids | tee $O/det.log
python -m pytest "tests/test_gpu_keras_shell.py::test_captured_step_is_bit_identical_to_eager_steps" tests/test_gpu_dp.py tests/test_gpu_decode_nms.py -x -q > $O/tests_a.log 2>&1; echo "tests_a rc $?"; grep -v "frame #" $O/tests_a.log | tail -30
