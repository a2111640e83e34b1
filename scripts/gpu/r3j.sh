#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_loss.py tests/test_gpu_kats.py tests/test_gpu_keras_shell.py -x -q --durations=5 > $O/tests.log 2>&1; echo "tests rc $?"; grep -v "frame #" $O/tests.log | tail -14
