#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_loss.py -x -q --durations=5 > $O/tests.log 2>&1; echo "tests rc $?"; grep -v "frame #" $O/tests.log | tail -14
python scripts/grad_excess.py 4 608 1 $O/ge_v4_608_1_final.json 2>&1 | grep -v amdgpu | tail -12
